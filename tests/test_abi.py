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
