"""CPU-side checks of the drop-in boundary: the library loads, exports every symbol include/bpsw.h and the
reference's JNI declare, and the host-side packer is byte-identical with the oracle's restatement of the
Scala packer (MemChainToAlignBatched.scala:76-172).  No compute calls (no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import bpsw_hip
from bpsw_hip import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = bpsw_hip.load_library()
    header = open(os.path.join(ROOT, "include", "bpsw.h")).read()
    declared = set(re.findall(r"\b(bpsw_[a-z0-9_]+)\s*\(", header))
    declared -= {"bpsw_ctx"}
    assert declared == set(bpsw_hip.ABI_SYMBOLS), declared ^ set(bpsw_hip.ABI_SYMBOLS)
    for name in sorted(declared) + bpsw_hip.JNI_SYMBOLS:
        assert hasattr(lib, name), f"{name} not exported by libbPSW_hip.so"


def test_struct_layouts_match_the_reference_records():
    assert C.sizeof(bpsw_hip.AlnReg) == 64  # mem_alnreg_t, native/bwamem.h:49-61
    assert C.sizeof(bpsw_hip.PeStat) == 32  # mem_pestat_t, native/bwamem.h:65-69
    o = bpsw_hip.default_opt()
    assert (o.a, o.b, o.o_del, o.e_del, o.o_ins, o.e_ins) == (1, 4, 6, 1, 6, 1)
    assert (o.pen_unpaired, o.pen_clip5, o.pen_clip3, o.w, o.zdrop, o.min_seed_len, o.max_matesw) == (17, 5, 5, 100, 100, 19, 100)
    mat = np.array(list(o.mat), np.int8).reshape(5, 5)
    assert mat[0, 0] == 1 and mat[0, 1] == -4 and mat[4, 4] == -1 and mat[2, 4] == -1  # bwaFillScmat


def test_no_device_is_an_error_not_a_fallback():
    lib = bpsw_hip.load_library()
    if lib.bpsw_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(bpsw_hip.BpswError):
        bpsw_hip.Context(0)
    with pytest.raises(bpsw_hip.BpswError):      # the tail pool's workers are contexts too: no device, no pool
        bpsw_hip.TailPool(0, workers=2)


@pytest.mark.parametrize("read_len,n", [(100, 300), (150, 500), (250, 200)])
def test_host_packer_matches_scala_packer_restatement(orc, read_len, n):
    soa = synth.ext_tasks(n, read_len=read_len, seed=1000 + read_len)
    assert soa.n > 0
    ours = bpsw_hip.wire_pack(soa)
    theirs = orc.wire_pack_soa(soa)
    assert ours.size == theirs.size and ours.size % 4 == 0
    assert np.array_equal(ours, theirs)
    # header fields (MemChainToAlignBatched.scala:78-85)
    assert list(ours[:7]) == [6, 1, 6, 1, 5, 5, 100] and int(np.frombuffer(ours[8:12].tobytes(), "<i4")[0]) == soa.n


def test_packer_rejects_small_buffer():
    soa = synth.ext_tasks(20, seed=5)
    lib = bpsw_hip.load_library()
    st = soa.as_struct()
    buf = np.zeros(16, np.uint8)
    used = C.c_size_t(0)
    rc = lib.bpsw_wire_pack(C.byref(st), buf.ctypes.data_as(C.c_void_p), buf.size, C.byref(used))
    assert rc == -3 and used.value == lib.bpsw_wire_size(C.byref(st))


def test_coordinate_batch_layout_against_the_format_1_batch():
    """wire format 2 (include/bpsw.h): every field a format-1 record has is at the same place, the seed length sits in the
    16-bit idx slot, the seed's rBeg follows as int64, and the nibble area holds leftQs + rightQs only -- checked against the
    format-1 bytes of the same tasks (which test_host_packer_matches_scala_packer_restatement pins on the Scala packer)."""
    l_pac = 60_013
    pac, bases = synth.random_pac(l_pac, seed=5)
    chains = synth.read_chains(40, bases, l_pac, read_len=100, seed=6)
    co, by = synth.coord_ext_tasks(chains, bases, seed=7)
    w1, w2 = bpsw_hip.wire_pack(by), bpsw_hip.wire_coords_pack(co)
    n = co.n
    assert list(w2[:7]) == list(w1[:7]) and w2[7] == 2 and w1[7] == 0
    assert np.array_equal(w2[8:12], w1[8:12])

    def nibbles(wire, word_off, count):
        words = np.frombuffer(wire.tobytes(), "<u4")
        return [int((words[word_off + k // 8] >> (4 * (7 - k % 8))) & 15) for k in range(count)]
    for t in range(n):
        a, b = 32 + 32 * t, 32 + 40 * t
        assert np.array_equal(w2[b: b + 8], w1[a: a + 8])              # four lengths
        assert np.array_equal(w2[b + 12: b + 18], w1[a + 12: a + 18])  # regScore, qBeg, h0
        assert np.array_equal(w2[b + 20: b + 32], w1[a + 20: a + 32])  # maxIns / maxDel shorts, idx
        assert int(np.frombuffer(w2[b + 18: b + 20].tobytes(), "<i2")[0]) == int(co.seed_len[t])
        assert int(np.frombuffer(w2[b + 32: b + 40].tobytes(), "<i8")[0]) == int(co.seed_rbeg[t])
        lq, rq = int(co.left_qlen[t]), int(co.right_qlen[t])
        o1 = int(np.frombuffer(w1[a + 8: a + 12].tobytes(), "<i4")[0])
        o2 = int(np.frombuffer(w2[b + 8: b + 12].tobytes(), "<i4")[0])
        assert o2 >= (32 + 40 * n) // 4
        assert nibbles(w2, o2, lq + rq) == nibbles(w1, o1, lq + rq)
    assert w2.size == 32 + 40 * n + 4 * sum(((int(co.left_qlen[t]) + int(co.right_qlen[t]) + 1) // 2 + 3) // 4 for t in range(n))
