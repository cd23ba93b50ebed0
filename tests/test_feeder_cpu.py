"""The bench harness's feeder (csrc/bpsw_feeder.cpp, libbpsw_synth.so) without a GPU: T native threads go round the items
`repeats` times with no barrier between the rounds, every item exactly `repeats` times, never two calls on one item at once
(an item's result buffers are shared by its repeats).  The C ABI is passed in as function pointers, here a Python callback."""
import ctypes as C
import threading
import time

from bpsw_hip.feeder import FeedItem
from bpsw_hip.synth import _load as load_synth


def test_feeder_rounds_without_barrier_and_without_overlap_on_an_item():
    syn = load_synth()
    syn.bpsw_feeder_create.restype = C.c_void_p
    syn.bpsw_feeder_create.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    syn.bpsw_feeder_run_repeats.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    syn.bpsw_feeder_destroy.argtypes = [C.c_void_p]
    syn.bpsw_feeder_destroy.restype = None
    n_threads, n_items, repeats = 6, 5, 40
    lock = threading.Lock()
    in_flight, calls, overlaps, spans = set(), [0] * n_items, [0], []

    @C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t)
    def fake_extend(ctx, wire, nbytes, out, out_len):
        item = int(nbytes)            # the item's index travels as its "size"
        with lock:
            if item in in_flight:
                overlaps[0] += 1
            in_flight.add(item)
            calls[item] += 1
            rnd = calls[item]
        t0 = time.perf_counter()
        time.sleep(0.0005 * (1 + item % 3))   # items of different cost: the threads drift apart
        with lock:
            in_flight.discard(item)
            spans.append((rnd, t0, time.perf_counter()))
        return 0

    ctxs = (C.c_void_p * n_threads)(*[C.c_void_p(t + 1) for t in range(n_threads)])
    items = (FeedItem * n_items)()
    for i in range(n_items):
        items[i].kind, items[i].in_bytes = 0, i
    h = syn.bpsw_feeder_create(n_threads, ctxs, C.cast(fake_extend, C.c_void_p), None, None, 0, None, 0)
    assert h
    try:
        assert syn.bpsw_feeder_run_repeats(h, items, n_items, repeats) == 0
        assert calls == [repeats] * n_items and overlaps[0] == 0
        # no barrier: a call of round r+1 starts while a (slower) call of round r is still running -- six threads, five items
        last_end = {}
        for rnd, t0, t1 in spans:
            last_end[rnd] = max(last_end.get(rnd, 0.0), t1)
        assert any(t0 < last_end[rnd - 1] for rnd, t0, t1 in spans if rnd > 1)
        assert syn.bpsw_feeder_run_repeats(h, items, n_items, 1) == 0      # the feeder can be run again
        assert calls == [repeats + 1] * n_items
        assert all(it.rc == 0 and it.ms > 0 for it in items)
    finally:
        syn.bpsw_feeder_destroy(h)
