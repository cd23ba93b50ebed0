"""ctypes face of csrc/bpsw_feeder.cpp (libbpsw_synth.so): T native host threads, one context each, that make the blocking
host-buffer calls of the two boundaries (bpsw_extend_batch, bpsw_matesw_group).  Harness for bench.py and the tests."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import ALNREG_DTYPE, BpswError, Context, Opt, load_library, RESCUE_C
from .synth import _load as _load_synth


class FeedItem(C.Structure):
    _fields_ = [("kind", C.c_int32), ("rc", C.c_int32), ("inp", C.c_void_p), ("in_bytes", C.c_size_t), ("out", C.c_void_p),
                ("out2", C.c_void_p), ("out_cap", C.c_int64), ("out_total", C.c_int64), ("ms", C.c_double), ("cpu_ms", C.c_double)]


class Feeder:
    """`n_threads` native threads over `n_threads` contexts of one device."""

    def __init__(self, n_threads: int, device: int, opt: Opt, mode: int = RESCUE_C, cpus=None):
        self.lib = load_library()
        self.syn = _load_synth()
        self.syn.bpsw_feeder_create.restype = C.c_void_p
        self.syn.bpsw_feeder_create.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        self.syn.bpsw_feeder_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        self.syn.bpsw_feeder_run_repeats.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        self.syn.bpsw_feeder_destroy.argtypes = [C.c_void_p]
        self.syn.bpsw_feeder_destroy.restype = None
        self.ctxs = [Context(device) for _ in range(n_threads)]
        self.opt = opt
        self._ctx_arr = (C.c_void_p * n_threads)(*[c.h for c in self.ctxs])
        self._cpus = np.ascontiguousarray(cpus, np.int32) if cpus is not None and len(cpus) else None
        self.h = self.syn.bpsw_feeder_create(n_threads, self._ctx_arr, C.cast(self.lib.bpsw_extend_batch, C.c_void_p),
                                             C.cast(self.lib.bpsw_matesw_group, C.c_void_p), C.byref(opt), mode,
                                             self._cpus.ctypes.data if self._cpus is not None else None,
                                             0 if self._cpus is None else int(self._cpus.size))
        if not self.h:
            raise BpswError("bpsw_feeder_create failed")
        self.n_threads = n_threads
        self.stage_commit = False

    def use_stage_commit(self, on: bool = True) -> None:
        """extension items through bpsw_extend_stage / bpsw_extend_commit, the way the JNI shim's swExtendFPGAJNI makes the call (the wire
        bytes go straight into the pinned staging block, the results are read where the kernel wrote them)"""
        self.syn.bpsw_feeder_use_stage_commit.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        self.syn.bpsw_feeder_use_stage_commit.restype = None
        self.syn.bpsw_feeder_use_stage_commit(self.h, C.cast(self.lib.bpsw_extend_stage, C.c_void_p) if on else None,
                                              C.cast(self.lib.bpsw_extend_commit, C.c_void_p) if on else None)
        self.stage_commit = bool(on)

    def run(self, items, repeats: int = 1) -> None:
        """items: a ctypes array of FeedItem (make_items); every item is run `repeats` times -- the threads go round the items
        without a barrier between the rounds, never two calls on one item at a time -- the call returns when all are done"""
        rc = self.syn.bpsw_feeder_run_repeats(self.h, items, len(items), int(repeats))
        if rc != 0:
            raise BpswError(f"a feeder call failed with code {rc}")

    def stats_sum(self):
        """sum of bpsw_get_stats over the contexts"""
        tot = {}
        for c in self.ctxs:
            s = c.stats()
            for f, _ in s._fields_:
                tot[f] = tot.get(f, 0) + getattr(s, f)
        return tot

    def reset_stats(self):
        for c in self.ctxs:
            self.lib.bpsw_reset_stats(c.h)

    def close(self):
        if self.h:
            self.syn.bpsw_feeder_destroy(self.h)
            self.h = None
        for c in self.ctxs:
            c.close()
        self.ctxs = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_items(wires, ext_outs, groups, group_structs, grp_cnts, grp_regs, interleave: bool = True):
    """One FeedItem per wire batch (out: int16[10 n]) and per rescue group (out_cnt int32[2G], out_regs ALNREG[cap]).
    interleave: spread the extension batches evenly between the groups so that both kernels share the device."""
    ne, ng = len(wires), len(groups)
    order = []
    if interleave and ne and ng:
        e = g = 0
        while e < ne or g < ng:  # Bresenham-style merge by fractional position
            if g >= ng or (e < ne and (e + 0.5) / ne <= (g + 0.5) / ng):
                order.append((0, e)); e += 1
            else:
                order.append((1, g)); g += 1
    else:
        order = [(0, i) for i in range(ne)] + [(1, i) for i in range(ng)]
    items = (FeedItem * len(order))()
    for at, (kind, i) in enumerate(order):
        it = items[at]
        it.kind = kind
        if kind == 0:
            it.inp, it.in_bytes = wires[i].ctypes.data, wires[i].size
            it.out, it.out_cap = ext_outs[i].ctypes.data, ext_outs[i].size
        else:
            it.inp = C.addressof(group_structs[i])
            it.out, it.out2, it.out_cap = grp_cnts[i].ctypes.data, grp_regs[i].ctypes.data, grp_regs[i].shape[0]
    return items, order
