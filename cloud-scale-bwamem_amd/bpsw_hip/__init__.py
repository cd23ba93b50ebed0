"""ctypes harness over the C ABI of libbPSW_hip.so (include/bpsw.h).

The product is the shared library; this package only lets the JVM-free tests and bench.py drive the
same entry points the JNI shim calls.  It never computes alignments itself and has no fallback: if the
HIP library is missing or no gfx950 device is usable, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(_HERE)
# BPSW_LIB: load another build of the same library (kernel experiments: tools/build_variant.sh)
LIB_PATH = os.environ.get("BPSW_LIB") or os.path.join(PKG_ROOT, "lib", "libbPSW_hip.so")
SYNTH_PATH = os.path.join(PKG_ROOT, "lib", "libbpsw_synth.so")

BPSW_OK = 0
ZDROP_SCALA, ZDROP_BWA = 0, 1
RESCUE_C, RESCUE_SCALA = 0, 1
KSW_XBYTE, KSW_XSTOP, KSW_XSUBO, KSW_XSTART = 0x10000, 0x20000, 0x40000, 0x80000

# every symbol include/bpsw.h declares (tests check the built library exports all of them)
ABI_SYMBOLS = [
    "bpsw_device_count", "bpsw_create", "bpsw_destroy", "bpsw_device_of", "bpsw_device_slots", "bpsw_device_for_partition", "bpsw_last_error", "bpsw_version",
    "bpsw_set_ext_scoring", "bpsw_set_ext_shortcuts", "bpsw_extend_batch", "bpsw_extend_stage", "bpsw_extend_commit", "bpsw_extend_batch_classify", "bpsw_extend_batch_device", "bpsw_wire_size", "bpsw_wire_pack", "bpsw_wire_coords_size", "bpsw_wire_coords_pack",
    "bpsw_opt_default", "bpsw_swalign2_batch", "bpsw_swalign2_batch_device", "bpsw_matesw_group", "bpsw_global_batch",
    "bpsw_get_stats", "bpsw_reset_stats", "bpsw_last_kernel_ms", "bpsw_ring_stats", "bpsw_sw_batches_in_flight", "bpsw_ring_integrity",
    "bpsw_ref_load", "bpsw_ref_unload", "bpsw_ref_length", "bpsw_ref_fetch", "bpsw_chain2aln_batch",
    "bpsw_tail_opt_default", "bpsw_bns_load", "bpsw_reg2aln_batch", "bpsw_sam_pe_batch", "bpsw_worker2_batch", "bpsw_last_tail_times",
    "bpsw_tail_pool_create", "bpsw_tail_pool_destroy", "bpsw_tail_pool_submit", "bpsw_tail_pool_wait", "bpsw_tail_pool_workers",
    "bpsw_mark_primary_se", "bpsw_approx_mapq_se", "bpsw_mem_pair", "bpsw_sort_dedup", "bpsw_pe_stat",
]
JNI_SYMBOLS = [
    "Java_cs_ucla_edu_bwaspark_jni_SWExtendFPGAJNI_swExtendFPGAJNI",
    "Java_cs_ucla_edu_bwaspark_jni_MateSWJNI_mateSWJNI",
    "Java_cs_ucla_edu_bwaspark_jni_HelloWorld_helloWorld",
    "Java_cs_ucla_edu_bwaspark_jni_MateSWJNI_loadPacJNI",   # new entry for SURVEY.md 8f.2
    "Java_cs_ucla_edu_bwaspark_jni_SWExtendFPGAJNI_chainToAlnJNI",   # new entry for SURVEY.md 8f.3
    "Java_cs_ucla_edu_bwaspark_jni_MateSWJNI_loadBnsJNI",            # new entries for SURVEY.md 8f.1 / 8f.4
    "Java_cs_ucla_edu_bwaspark_jni_MateSWJNI_samPeTailJNI",
    "Java_cs_ucla_edu_bwaspark_jni_MateSWJNI_samPeTailSubmitJNI",   # the same call in two halves (bpsw_tail_pool_*, round 5)
    "Java_cs_ucla_edu_bwaspark_jni_MateSWJNI_samPeTailCollectJNI",
    "Java_cs_ucla_edu_bwaspark_jni_MateSWJNI_samPeTailCancelJNI",    # drop a handle that will not be collected (round 6)
    "Java_cs_ucla_edu_bwaspark_jni_MateSWJNI_mateSWFlatJNI",         # boundary 1 with primitive arrays (round 4)
]


class BpswError(RuntimeError):
    pass


class AlnReg(C.Structure):  # bpsw_alnreg_t
    _fields_ = [("rb", C.c_int64), ("re", C.c_int64), ("qb", C.c_int32), ("qe", C.c_int32), ("score", C.c_int32),
                ("truesc", C.c_int32), ("sub", C.c_int32), ("csub", C.c_int32), ("sub_n", C.c_int32), ("w", C.c_int32),
                ("seedcov", C.c_int32), ("secondary", C.c_int32), ("hash", C.c_uint64)]


ALNREG_DTYPE = np.dtype([("rb", "<i8"), ("re", "<i8"), ("qb", "<i4"), ("qe", "<i4"), ("score", "<i4"),
                         ("truesc", "<i4"), ("sub", "<i4"), ("csub", "<i4"), ("sub_n", "<i4"), ("w", "<i4"),
                         ("seedcov", "<i4"), ("secondary", "<i4"), ("hash", "<u8")])
assert ALNREG_DTYPE.itemsize == 64 and C.sizeof(AlnReg) == 64


class PeStat(C.Structure):  # bpsw_pestat_t
    _fields_ = [("low", C.c_int32), ("high", C.c_int32), ("failed", C.c_int32), ("pad_", C.c_int32),
                ("avg", C.c_double), ("std", C.c_double)]


class Opt(C.Structure):  # bpsw_opt_t
    _fields_ = [(n, C.c_int32) for n in ("a", "b", "o_del", "e_del", "o_ins", "e_ins", "pen_unpaired", "pen_clip5",
                                         "pen_clip3", "w", "zdrop", "T", "flag", "min_seed_len", "max_ins",
                                         "max_matesw")] + [("mask_level_redun", C.c_float), ("mat", C.c_int8 * 25),
                                                           ("pad_", C.c_int8 * 3)]


class ExtTasks(C.Structure):  # bpsw_ext_tasks_t
    _fields_ = [("n", C.c_int32)] + [(n, C.c_int32) for n in ("o_del", "e_del", "o_ins", "e_ins", "pen_clip5",
                                                                "pen_clip3", "w", "mat_max")] + \
               [(n, C.c_void_p) for n in ("left_qlen", "left_rlen", "right_qlen", "right_rlen", "left_q_off",
                                          "left_r_off", "right_q_off", "right_r_off", "reg_score", "q_beg", "h0", "idx",
                                          "pool")]


class ExtCoordTasks(C.Structure):  # bpsw_ext_coord_tasks_t
    _fields_ = [("n", C.c_int32)] + [(n, C.c_int32) for n in ("o_del", "e_del", "o_ins", "e_ins", "pen_clip5",
                                                                "pen_clip3", "w", "mat_max")] + \
               [(n, C.c_void_p) for n in ("left_qlen", "left_rlen", "right_qlen", "right_rlen", "left_q_off", "right_q_off",
                                          "reg_score", "q_beg", "h0", "idx", "seed_len", "seed_rbeg", "pool")]


class SwJobs(C.Structure):  # bpsw_sw_jobs_t
    _fields_ = [("n", C.c_int32), ("xtra", C.c_int32)] + \
               [(n, C.c_void_p) for n in ("q_len", "t_len", "q_off", "t_off", "q_rev", "q_pool", "t_pool")] + \
               [("q_pool_bytes", C.c_size_t), ("t_pool_bytes", C.c_size_t)]


class GlobalJobs(C.Structure):  # bpsw_global_jobs_t
    _fields_ = [("n", C.c_int32), ("max_cigar", C.c_int32)] + \
               [(n, C.c_void_p) for n in ("q_len", "t_len", "w", "q_off", "t_off", "q_pool", "t_pool")] + \
               [("q_pool_bytes", C.c_size_t), ("t_pool_bytes", C.c_size_t)]


class Chains(C.Structure):  # bpsw_chains_t
    _fields_ = [("n_reads", C.c_int32), ("read_len", C.c_void_p), ("read_off", C.c_void_p), ("read_pool", C.c_void_p),
                ("read_pool_bytes", C.c_size_t), ("chain_cnt", C.c_void_p), ("seed_cnt", C.c_void_p), ("seed_rbeg", C.c_void_p),
                ("seed_qbeg", C.c_void_p), ("seed_len", C.c_void_p)]


C2A_SORT_DEDUP, C2A_DEDUP_SCALA = 1, 2


class RescueGroup(C.Structure):  # bpsw_rescue_group_t
    _fields_ = [("group_size", C.c_int32), ("l_pac", C.c_int64), ("pes", PeStat * 4), ("seq_len", C.c_void_p),
                ("seq_off", C.c_void_p), ("seq_pool", C.c_void_p), ("seq_pool_bytes", C.c_size_t),
                ("reg_cnt", C.c_void_p), ("regs", C.c_void_p), ("ref_cnt", C.c_void_p), ("ref_rb", C.c_void_p),
                ("ref_re", C.c_void_p), ("ref_len", C.c_void_p), ("ref_off", C.c_void_p), ("ref_pool", C.c_void_p),
                ("ref_pool_bytes", C.c_size_t)]


class Stats(C.Structure):  # bpsw_stats_t
    _fields_ = [(n, C.c_uint64) for n in ("ext_calls", "ext_tasks", "ext_wire_bytes", "sw_calls", "sw_jobs",
                                          "sw_speculated", "sw_replayed_rounds", "sw_wasted")] + \
               [(n, C.c_double) for n in ("ext_h2d_ms", "ext_kernel_ms", "ext_d2h_ms", "sw_h2d_ms", "sw_kernel_ms",
                                          "sw_d2h_ms", "sw_host_ms", "ext_host_in_ms", "ext_wait_ms", "ext_dev_ms",
                                          "ext_host_out_ms", "grp_plan_ms", "grp_pack_ms", "grp_wait_ms", "grp_dev_ms",
                                          "grp_replay_ms", "grp_out_ms")] + [("grp_calls", C.c_uint64), ("grp_pairs", C.c_uint64), ("ext_full_relaunches", C.c_uint64), ("sw_ring_calls", C.c_uint64), ("ext_ring_calls", C.c_uint64)]


_lib = None


def load_library(path: str | None = None) -> C.CDLL:
    """Load libbPSW_hip.so; fails loudly when it has not been built (no fallback of any kind)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise BpswError(f"{p} not found: build it with `make -C {PKG_ROOT}` (or __graft_entry__.build())")
    if os.environ.get("BPSW_NO_TORCH") != "1":
        # The PyTorch wheel bundles its own libamdhip64.so.7 / libhsa-runtime64; a process must initialise only
        # one HIP runtime.  Loading torch first makes this library bind to the runtime torch uses, so device
        # pointers and streams can be shared (bench.py, device-resident tests).  Under the JVM there is no
        # torch and the system ROCm runtime is used.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    lib = C.CDLL(p)
    lib.bpsw_last_error.restype = C.c_char_p
    lib.bpsw_version.restype = C.c_char_p
    lib.bpsw_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    lib.bpsw_destroy.argtypes = [C.c_void_p]
    lib.bpsw_destroy.restype = None
    lib.bpsw_device_of.argtypes = [C.c_void_p]
    lib.bpsw_set_ext_scoring.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    lib.bpsw_set_ext_shortcuts.argtypes = [C.c_void_p, C.c_int]
    lib.bpsw_extend_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    lib.bpsw_extend_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    lib.bpsw_wire_size.argtypes = [C.POINTER(ExtTasks)]
    lib.bpsw_wire_size.restype = C.c_size_t
    lib.bpsw_wire_pack.argtypes = [C.POINTER(ExtTasks), C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.bpsw_wire_coords_size.argtypes = [C.POINTER(ExtCoordTasks)]
    lib.bpsw_wire_coords_size.restype = C.c_size_t
    lib.bpsw_wire_coords_pack.argtypes = [C.POINTER(ExtCoordTasks), C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.bpsw_opt_default.argtypes = [C.POINTER(Opt)]
    lib.bpsw_opt_default.restype = None
    lib.bpsw_swalign2_batch.argtypes = [C.c_void_p, C.POINTER(Opt), C.POINTER(SwJobs), C.c_void_p]
    lib.bpsw_swalign2_batch_device.argtypes = [C.c_void_p, C.POINTER(Opt), C.POINTER(SwJobs), C.c_void_p, C.c_void_p]
    lib.bpsw_matesw_group.argtypes = [C.c_void_p, C.POINTER(Opt), C.POINTER(RescueGroup), C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    lib.bpsw_global_batch.argtypes = [C.c_void_p, C.POINTER(Opt), C.POINTER(GlobalJobs), C.c_void_p, C.c_void_p, C.c_void_p]
    lib.bpsw_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
    lib.bpsw_reset_stats.argtypes = [C.c_void_p]
    lib.bpsw_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.bpsw_ring_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    lib.bpsw_sw_batches_in_flight.argtypes = [C.c_int]
    lib.bpsw_ring_integrity.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.bpsw_chain2aln_batch.argtypes = [C.c_void_p, C.POINTER(Opt), C.POINTER(Chains), C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                         C.c_int64, C.POINTER(C.c_int64)]
    lib.bpsw_ref_load.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    lib.bpsw_ref_unload.argtypes = [C.c_void_p]
    lib.bpsw_ref_length.argtypes = [C.c_void_p]
    lib.bpsw_ref_length.restype = C.c_int64
    lib.bpsw_ref_fetch.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    lib.bpsw_tail_opt_default.argtypes = [C.c_void_p]
    lib.bpsw_tail_opt_default.restype = None
    lib.bpsw_bns_load.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_char_p]
    lib.bpsw_reg2aln_batch.argtypes = [C.c_void_p, C.POINTER(Opt), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.bpsw_sam_pe_batch.argtypes = [C.c_void_p, C.POINTER(Opt), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                      C.POINTER(C.c_size_t), C.c_void_p]
    lib.bpsw_worker2_batch.argtypes = [C.c_void_p, C.POINTER(Opt), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p,
                                       C.POINTER(C.c_size_t), C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    lib.bpsw_tail_pool_create.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    lib.bpsw_tail_pool_destroy.argtypes = [C.c_void_p]
    lib.bpsw_tail_pool_destroy.restype = None
    lib.bpsw_tail_pool_submit.argtypes = [C.c_void_p, C.POINTER(Opt), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    lib.bpsw_tail_pool_wait.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_size_t), C.POINTER(C.c_int64)]
    lib.bpsw_tail_pool_workers.argtypes = [C.c_void_p]
    lib.bpsw_mark_primary_se.argtypes = [C.POINTER(Opt), C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
    lib.bpsw_approx_mapq_se.argtypes = [C.POINTER(Opt), C.c_void_p, C.c_void_p]
    lib.bpsw_mem_pair.argtypes = [C.POINTER(Opt), C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64,
                                  C.c_void_p]
    lib.bpsw_sort_dedup.argtypes = [C.c_int32, C.c_void_p, C.c_float, C.c_int]
    lib.bpsw_pe_stat.argtypes = [C.POINTER(Opt), C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.bpsw_last_tail_times.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32), C.c_void_p]
    if path is None:
        _lib = lib
    return lib


def _ptr(a: np.ndarray | None):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _chk(lib, rc, what):
    if rc != BPSW_OK:
        raise BpswError(f"{what} failed ({rc}): {lib.bpsw_last_error().decode()}")


def default_opt() -> Opt:
    o = Opt()
    load_library().bpsw_opt_default(C.byref(o))
    return o


@dataclass
class ExtTaskSoA:
    """ExtParam fields (datatype/ExtensionParameters.scala:21-45) in struct-of-arrays form."""
    left_qlen: np.ndarray
    left_rlen: np.ndarray
    right_qlen: np.ndarray
    right_rlen: np.ndarray
    left_q_off: np.ndarray
    left_r_off: np.ndarray
    right_q_off: np.ndarray
    right_r_off: np.ndarray
    reg_score: np.ndarray
    q_beg: np.ndarray
    h0: np.ndarray
    idx: np.ndarray
    pool: np.ndarray
    o_del: int = 6
    e_del: int = 1
    o_ins: int = 6
    e_ins: int = 1
    pen_clip5: int = 5
    pen_clip3: int = 5
    w: int = 100
    mat_max: int = 1

    @property
    def n(self) -> int:
        return int(self.left_qlen.shape[0])

    def subset(self, sel) -> "ExtTaskSoA":
        kw = {k: getattr(self, k)[sel] for k in ("left_qlen", "left_rlen", "right_qlen", "right_rlen", "left_q_off",
                                                  "left_r_off", "right_q_off", "right_r_off", "reg_score", "q_beg",
                                                  "h0", "idx")}
        return ExtTaskSoA(pool=self.pool, o_del=self.o_del, e_del=self.e_del, o_ins=self.o_ins, e_ins=self.e_ins,
                          pen_clip5=self.pen_clip5, pen_clip3=self.pen_clip3, w=self.w, mat_max=self.mat_max,
                          **{k: np.ascontiguousarray(v) for k, v in kw.items()})

    def as_struct(self) -> ExtTasks:
        t = ExtTasks()
        t.n = self.n
        for f in ("o_del", "e_del", "o_ins", "e_ins", "pen_clip5", "pen_clip3", "w", "mat_max"):
            setattr(t, f, int(getattr(self, f)))
        for f in ("left_qlen", "left_rlen", "right_qlen", "right_rlen", "reg_score", "q_beg", "h0", "idx"):
            a = getattr(self, f)
            assert a.dtype == np.int32 and a.flags.c_contiguous
            setattr(t, f, a.ctypes.data)
        for f in ("left_q_off", "left_r_off", "right_q_off", "right_r_off"):
            a = getattr(self, f)
            assert a.dtype == np.int64 and a.flags.c_contiguous
            setattr(t, f, a.ctypes.data)
        assert self.pool.dtype == np.uint8
        t.pool = self.pool.ctypes.data
        return t


@dataclass
class ExtCoordTaskSoA:
    """bpsw_ext_coord_tasks_t: ExtParam without the target flanks, which the seed's coordinates name."""
    left_qlen: np.ndarray
    left_rlen: np.ndarray
    right_qlen: np.ndarray
    right_rlen: np.ndarray
    left_q_off: np.ndarray
    right_q_off: np.ndarray
    reg_score: np.ndarray
    q_beg: np.ndarray
    h0: np.ndarray
    idx: np.ndarray
    seed_len: np.ndarray
    seed_rbeg: np.ndarray
    pool: np.ndarray
    o_del: int = 6
    e_del: int = 1
    o_ins: int = 6
    e_ins: int = 1
    pen_clip5: int = 5
    pen_clip3: int = 5
    w: int = 100
    mat_max: int = 1

    @property
    def n(self) -> int:
        return int(self.left_qlen.shape[0])

    def as_struct(self) -> ExtCoordTasks:
        t = ExtCoordTasks()
        t.n = self.n
        for f in ("o_del", "e_del", "o_ins", "e_ins", "pen_clip5", "pen_clip3", "w", "mat_max"):
            setattr(t, f, int(getattr(self, f)))
        for f in ("left_qlen", "left_rlen", "right_qlen", "right_rlen", "reg_score", "q_beg", "h0", "idx", "seed_len"):
            a = getattr(self, f)
            assert a.dtype == np.int32 and a.flags.c_contiguous
            setattr(t, f, a.ctypes.data)
        for f in ("left_q_off", "right_q_off", "seed_rbeg"):
            a = getattr(self, f)
            assert a.dtype == np.int64 and a.flags.c_contiguous
            setattr(t, f, a.ctypes.data)
        assert self.pool.dtype == np.uint8
        t.pool = self.pool.ctypes.data
        return t


def wire_coords_pack(tasks: ExtCoordTaskSoA) -> np.ndarray:
    """A coordinate batch (wire format 2, include/bpsw.h) for bpsw_extend_batch / swExtendFPGAJNI."""
    lib = load_library()
    st = tasks.as_struct()
    size = lib.bpsw_wire_coords_size(C.byref(st))
    buf = np.zeros(size, dtype=np.uint8)
    used = C.c_size_t(0)
    _chk(lib, lib.bpsw_wire_coords_pack(C.byref(st), _ptr(buf), size, C.byref(used)), "bpsw_wire_coords_pack")
    return buf[: used.value]


def wire_pack(tasks: ExtTaskSoA) -> np.ndarray:
    """Mirror of runOnFPGAJNI's packing (MemChainToAlignBatched.scala:76-172) -> the JNI byte[]."""
    lib = load_library()
    st = tasks.as_struct()
    size = lib.bpsw_wire_size(C.byref(st))
    buf = np.zeros(size, dtype=np.uint8)
    used = C.c_size_t(0)
    _chk(lib, lib.bpsw_wire_pack(C.byref(st), _ptr(buf), size, C.byref(used)), "bpsw_wire_pack")
    return buf[: used.value]


class Context:
    """One bpsw_ctx_t: a device, a stream and its arenas."""

    def __init__(self, device: int = -1):
        self.lib = load_library()
        h = C.c_void_p()
        _chk(self.lib, self.lib.bpsw_create(device, C.byref(h)), "bpsw_create")
        self.h = h

    def close(self):
        if self.h:
            self.lib.bpsw_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def device(self) -> int:
        return self.lib.bpsw_device_of(self.h)

    def set_ext_scoring(self, mat=None, zdrop: int = 100, zdrop_mode: int = ZDROP_SCALA):
        m = None if mat is None else np.ascontiguousarray(mat, dtype=np.int8)
        _chk(self.lib, self.lib.bpsw_set_ext_scoring(self.h, _ptr(m), zdrop, zdrop_mode), "bpsw_set_ext_scoring")

    # boundary 2 ------------------------------------------------------------------------------
    def set_ext_shortcuts(self, mask: int = -1):
        _chk(self.lib, self.lib.bpsw_set_ext_shortcuts(self.h, C.c_int(mask)), "bpsw_set_ext_shortcuts")

    def extend_batch(self, wire: np.ndarray) -> np.ndarray:
        """swExtendFPGAJNI(n*10, wire) -> int16[10*n]"""
        wire = np.ascontiguousarray(wire, dtype=np.uint8)
        n = int(np.frombuffer(wire[8:12].tobytes(), dtype="<i4")[0]) if wire.size >= 12 else 0
        out = np.zeros(max(10 * n, 1), dtype=np.int16)
        _chk(self.lib, self.lib.bpsw_extend_batch(self.h, _ptr(wire), wire.size, _ptr(out), out.size), "bpsw_extend_batch")
        return out[: 10 * n]

    def extend_batch_classify(self, wire: np.ndarray):
        """extend_batch + per task and side how the result was produced (0 empty side, 1 exact shortcut, 2 DP swept)"""
        wire = np.ascontiguousarray(wire, dtype=np.uint8)
        n = int(np.frombuffer(wire[8:12].tobytes(), dtype="<i4")[0]) if wire.size >= 12 else 0
        out = np.zeros(max(10 * n, 1), dtype=np.int16)
        how = np.zeros((max(n, 1), 2), dtype=np.uint8)
        self.lib.bpsw_extend_batch_classify.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        _chk(self.lib, self.lib.bpsw_extend_batch_classify(self.h, _ptr(wire), wire.size, _ptr(out), out.size, _ptr(how)),
             "bpsw_extend_batch_classify")
        return out[: 10 * n], how[:n]

    def extend_batch_device(self, d_wire_ptr: int, wire_bytes: int, n_tasks: int, d_out_ptr: int, stream: int = 0):
        _chk(self.lib, self.lib.bpsw_extend_batch_device(self.h, C.c_void_p(d_wire_ptr), wire_bytes, n_tasks,
                                                        C.c_void_p(d_out_ptr), C.c_void_p(stream)),
             "bpsw_extend_batch_device")

    # boundary 1 ------------------------------------------------------------------------------
    def swalign2_batch(self, opt: Opt, xtra: int, q_len, t_len, q_off, t_off, q_rev, q_pool, t_pool) -> np.ndarray:
        """t_pool=None: t_off/t_len name windows of the reference loaded with ref_load (doubled coordinates)"""
        j = SwJobs()
        arrs = dict(q_len=np.ascontiguousarray(q_len, np.int32), t_len=np.ascontiguousarray(t_len, np.int32),
                    q_off=np.ascontiguousarray(q_off, np.int64), t_off=np.ascontiguousarray(t_off, np.int64),
                    q_rev=np.ascontiguousarray(q_rev, np.uint8), q_pool=np.ascontiguousarray(q_pool, np.uint8))
        if t_pool is not None:
            arrs["t_pool"] = np.ascontiguousarray(t_pool, np.uint8)
        j.n = int(arrs["q_len"].shape[0])
        j.xtra = int(xtra)
        for k, a in arrs.items():
            setattr(j, k, a.ctypes.data)
        j.q_pool_bytes = arrs["q_pool"].size
        j.t_pool_bytes = arrs["t_pool"].size if t_pool is not None else 0
        out = np.zeros((max(j.n, 1), 7), dtype=np.int32)
        _chk(self.lib, self.lib.bpsw_swalign2_batch(self.h, C.byref(opt), C.byref(j), _ptr(out)), "bpsw_swalign2_batch")
        return out[: j.n]

    def swalign2_batch_device(self, opt: Opt, jobs: SwJobs, d_out_ptr: int, stream: int = 0):
        _chk(self.lib, self.lib.bpsw_swalign2_batch_device(self.h, C.byref(opt), C.byref(jobs), C.c_void_p(d_out_ptr),
                                                          C.c_void_p(stream)), "bpsw_swalign2_batch_device")

    def global_batch(self, opt: Opt, q_len, t_len, w, q_off, t_off, q_pool, t_pool, max_cigar: int = 64):
        """SWGlobal (SWUtil.scala:233-397) for a batch of jobs -> (score[n], list of cigar arrays len<<4|op)"""
        j = GlobalJobs()
        arrs = dict(q_len=np.ascontiguousarray(q_len, np.int32), t_len=np.ascontiguousarray(t_len, np.int32),
                    w=np.ascontiguousarray(w, np.int32), q_off=np.ascontiguousarray(q_off, np.int64),
                    t_off=np.ascontiguousarray(t_off, np.int64), q_pool=np.ascontiguousarray(q_pool, np.uint8),
                    t_pool=np.ascontiguousarray(t_pool, np.uint8))
        j.n, j.max_cigar = int(arrs["q_len"].shape[0]), int(max_cigar)
        for k, a in arrs.items():
            setattr(j, k, a.ctypes.data)
        j.q_pool_bytes, j.t_pool_bytes = arrs["q_pool"].size, arrs["t_pool"].size
        score = np.zeros(max(j.n, 1), np.int32)
        ncig = np.zeros(max(j.n, 1), np.int32)
        cig = np.zeros((max(j.n, 1), max_cigar), np.uint32)
        _chk(self.lib, self.lib.bpsw_global_batch(self.h, C.byref(opt), C.byref(j), _ptr(score), _ptr(ncig), _ptr(cig)),
             "bpsw_global_batch")
        return score[: j.n], ncig[: j.n], cig[: j.n]

    def matesw_group(self, opt: Opt, g: "RescueGroupSoA", mode: int = RESCUE_C):
        st = g.as_struct()
        out_cnt = np.zeros(2 * g.group_size, dtype=np.int32)
        cap = int(g.regs.shape[0] + 4 * g.ref_rb.shape[0] + 16)
        out = np.empty(cap, dtype=ALNREG_DTYPE)   # the library writes the first `total` records; zeroing 10 MB per call was most of the call
        total = C.c_int64(0)
        _chk(self.lib, self.lib.bpsw_matesw_group(self.h, C.byref(opt), C.byref(st), mode, _ptr(out_cnt), _ptr(out), cap,
                                                 C.byref(total)), "bpsw_matesw_group")
        return out_cnt, out[: total.value]

    # reference resident on the device --------------------------------------------
    def ref_load(self, pac: np.ndarray, l_pac: int):
        pac = np.ascontiguousarray(pac, np.uint8)
        if pac.size < (int(l_pac) + 3) // 4:
            raise BpswError("pac shorter than (l_pac+3)/4 bytes")
        _chk(self.lib, self.lib.bpsw_ref_load(self.h, _ptr(pac), int(l_pac)), "bpsw_ref_load")

    def ref_unload(self):
        _chk(self.lib, self.lib.bpsw_ref_unload(self.h), "bpsw_ref_unload")

    def ref_length(self) -> int:
        return int(self.lib.bpsw_ref_length(self.h))

    def ref_fetch(self, beg, end):
        """bnsGetSeq for n windows -> (list of uint8 arrays, lengths)"""
        beg = np.ascontiguousarray(beg, np.int64)
        end = np.ascontiguousarray(end, np.int64)
        n = int(beg.shape[0])
        span = np.abs(end - beg) + 16
        off = np.zeros(n, np.int64)
        if n > 1:
            off[1:] = np.cumsum((span[:-1] + 15) & ~15)
        total = int(off[-1] + span[-1]) if n else 0
        pool = np.zeros(max(total, 16), np.uint8)
        lens = np.zeros(max(n, 1), np.int64)
        _chk(self.lib, self.lib.bpsw_ref_fetch(self.h, n, _ptr(beg), _ptr(end), _ptr(pool), pool.size, _ptr(off), _ptr(lens)),
             "bpsw_ref_fetch")
        return [pool[off[i]: off[i] + lens[i]].copy() for i in range(n)], lens[:n]

    # memChainToAlnBatched on the device ------------------------------------------
    def chain2aln_batch(self, opt: Opt, b: "ChainBatchSoA", zdrop_mode: int = ZDROP_SCALA, flags: int = 0):
        st = Chains()
        st.n_reads = b.n_reads
        for f, dt in (("read_len", np.int32), ("read_off", np.int64), ("read_pool", np.uint8), ("chain_cnt", np.int32),
                      ("seed_cnt", np.int32), ("seed_rbeg", np.int64), ("seed_qbeg", np.int32), ("seed_len", np.int32)):
            a = getattr(b, f)
            assert a.dtype == dt and a.flags.c_contiguous, f
            setattr(st, f, a.ctypes.data)
        st.read_pool_bytes = b.read_pool.size
        cap = int(b.seed_len.shape[0]) + 8
        out_cnt = np.zeros(max(b.n_reads, 1), np.int32)
        out = np.empty(cap, dtype=ALNREG_DTYPE)   # the library writes the first `total` records
        total = C.c_int64(0)
        _chk(self.lib, self.lib.bpsw_chain2aln_batch(self.h, C.byref(opt), C.byref(st), zdrop_mode, flags, _ptr(out_cnt), _ptr(out), cap,
                                                    C.byref(total)), "bpsw_chain2aln_batch")
        return out_cnt[: b.n_reads], out[: total.value]

    def ring_stats(self):
        """(epochs, submitted, carried) of the device's submission ring (include/bpsw.h: bpsw_ring_stats)"""
        e, s, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        _chk(self.lib, self.lib.bpsw_ring_stats(self.h, C.byref(e), C.byref(s), C.byref(c), None, None), "bpsw_ring_stats")
        return int(e.value), int(s.value), int(c.value)

    def ring_integrity(self):
        """(on, records checked, faults) of the rings' integrity tripwire, process-wide (include/bpsw.h: bpsw_ring_integrity)"""
        a, b = C.c_uint64(0), C.c_uint64(0)
        on = self.lib.bpsw_ring_integrity(C.byref(a), C.byref(b))
        return bool(on), int(a.value), int(b.value)

    def sw_batches_in_flight(self):
        """gauge: SW batches of any context in their device phase on this context's device (include/bpsw.h)"""
        return int(self.lib.bpsw_sw_batches_in_flight(self.lib.bpsw_device_of(self.h)))

    def ring_epoch_times(self):
        """(summed duration in ms, count) of the device's ring epochs that are over: each the resident kernel's launch as a kernel trace times it"""
        ms, n = C.c_double(0.0), C.c_uint64(0)
        _chk(self.lib, self.lib.bpsw_ring_stats(self.h, None, None, None, C.byref(ms), C.byref(n)), "bpsw_ring_stats")
        return float(ms.value), int(n.value)

    def stats(self) -> Stats:
        s = Stats()
        _chk(self.lib, self.lib.bpsw_get_stats(self.h, C.byref(s)), "bpsw_get_stats")
        return s

    def last_kernel_ms(self):
        a, b = C.c_float(0), C.c_float(0)
        _chk(self.lib, self.lib.bpsw_last_kernel_ms(self.h, C.byref(a), C.byref(b)), "bpsw_last_kernel_ms")
        return a.value, b.value


@dataclass
class ChainBatchSoA:
    """Reads and their seed chains, the arguments of memChainToAlnBatched (MemChainToAlignBatched.scala:380-391), flat."""
    l_pac: int
    read_len: np.ndarray   # int32 [n]
    read_off: np.ndarray   # int64 [n]
    read_pool: np.ndarray  # uint8, codes 0..4
    chain_cnt: np.ndarray  # int32 [n]
    seed_cnt: np.ndarray   # int32 [sum chain_cnt]
    seed_rbeg: np.ndarray  # int64 [sum seed_cnt]
    seed_qbeg: np.ndarray  # int32
    seed_len: np.ndarray   # int32

    @property
    def n_reads(self) -> int:
        return int(self.read_len.shape[0])

    def slice(self, lo: int, hi: int) -> "ChainBatchSoA":
        """reads lo..hi-1 with their chains and seeds (the read pool is shared, offsets stay valid)"""
        c0, c1 = int(self.chain_cnt[:lo].sum()), int(self.chain_cnt[:hi].sum())
        s0, s1 = int(self.seed_cnt[:c0].sum()), int(self.seed_cnt[:c1].sum())
        cp = np.ascontiguousarray
        return ChainBatchSoA(l_pac=self.l_pac, read_len=cp(self.read_len[lo:hi]), read_off=cp(self.read_off[lo:hi]),
                             read_pool=self.read_pool, chain_cnt=cp(self.chain_cnt[lo:hi]), seed_cnt=cp(self.seed_cnt[c0:c1]),
                             seed_rbeg=cp(self.seed_rbeg[s0:s1]), seed_qbeg=cp(self.seed_qbeg[s0:s1]), seed_len=cp(self.seed_len[s0:s1]))


@dataclass
class RescueGroupSoA:
    """The arguments of MateSWJNI.mateSWJNI (MemSamPe.scala:2091-2092) flattened; see include/bpsw.h."""
    group_size: int
    l_pac: int
    pes: list  # 4 x (low, high, failed, avg, std)
    seq_len: np.ndarray
    seq_off: np.ndarray
    seq_pool: np.ndarray
    reg_cnt: np.ndarray
    regs: np.ndarray  # ALNREG_DTYPE
    ref_cnt: np.ndarray
    ref_rb: np.ndarray
    ref_re: np.ndarray
    ref_len: np.ndarray
    ref_off: np.ndarray
    ref_pool: np.ndarray | None  # None: windows are (ref_rb, ref_re) coordinates of the reference loaded with ref_load

    def as_struct(self) -> RescueGroup:
        g = RescueGroup()
        g.group_size = self.group_size
        g.l_pac = self.l_pac
        for r in range(4):
            lo, hi, failed, avg, std = self.pes[r]
            g.pes[r].low, g.pes[r].high, g.pes[r].failed, g.pes[r].avg, g.pes[r].std = int(lo), int(hi), int(failed), float(avg), float(std)
        for f, dt in (("seq_len", np.int32), ("seq_off", np.int64), ("seq_pool", np.uint8), ("reg_cnt", np.int32),
                      ("ref_cnt", np.int32), ("ref_rb", np.int64), ("ref_re", np.int64), ("ref_len", np.int64),
                      ("ref_off", np.int64), ("ref_pool", np.uint8)):
            a = getattr(self, f)
            if a is None and f in ("ref_pool", "ref_len", "ref_off"):
                continue  # coordinate mode
            assert a.dtype == dt and a.flags.c_contiguous, f
            setattr(g, f, a.ctypes.data)
        assert self.regs.dtype == ALNREG_DTYPE
        g.regs = self.regs.ctypes.data
        g.seq_pool_bytes = self.seq_pool.size
        g.ref_pool_bytes = self.ref_pool.size if self.ref_pool is not None else 0
        return g


@dataclass
class TailGroupSoA:
    """What worker2's tail (memSamPeGroupRest, MemSamPe.scala:1390-1612) works on, flat: a group of pairs with the region
    lists of both ends after the rescue, the reads with names and qualities, and the contig table of the reference."""
    group_size: int
    l_pac: int
    id0: int               # pair id of the first pair (the reference hashes id + k)
    pes: list              # 4 x (low, high, failed, avg, std)
    read_len: np.ndarray   # int32 [2G], (pair, end) order
    read_off: np.ndarray   # int64 [2G]
    read_pool: np.ndarray  # uint8 codes 0..4
    qual_pool: np.ndarray | None  # uint8 ASCII, same offsets as read_pool
    name_off: np.ndarray   # int64 [G+1]
    name_pool: np.ndarray  # uint8
    reg_cnt: np.ndarray    # int32 [2G]
    regs: np.ndarray       # ALNREG_DTYPE, (pair, end, j) order
    ann_off: np.ndarray    # int64 [n_seqs]   bntann1_t.offset
    ann_len: np.ndarray    # int32 [n_seqs]   bntann1_t.len
    ann_name_off: np.ndarray  # int64 [n_seqs+1]
    ann_name_pool: np.ndarray  # uint8


def make_tail_group(chain_batch, names, qual_pool, pes, reg_cnt, regs, ann_off, ann_len, ann_names, id0=0) -> TailGroupSoA:
    n = chain_batch.n_reads
    assert n % 2 == 0 and len(names) == n // 2

    def pool(strs):
        off = np.zeros(len(strs) + 1, np.int64)
        bs = [s.encode() for s in strs]
        off[1:] = np.cumsum([len(b) for b in bs])
        return off, np.frombuffer(b"".join(bs) + b"\0", np.uint8).copy()

    name_off, name_pool = pool(names)
    a_off, a_pool = pool(ann_names)
    cp = np.ascontiguousarray
    return TailGroupSoA(group_size=n // 2, l_pac=int(chain_batch.l_pac), id0=int(id0), pes=list(pes), read_len=cp(chain_batch.read_len),
                        read_off=cp(chain_batch.read_off), read_pool=cp(chain_batch.read_pool),
                        qual_pool=None if qual_pool is None else cp(qual_pool, np.uint8), name_off=name_off, name_pool=name_pool,
                        reg_cnt=cp(reg_cnt, np.int32), regs=cp(regs), ann_off=cp(ann_off, np.int64), ann_len=cp(ann_len, np.int32),
                        ann_name_off=a_off, ann_name_pool=a_pool)


# ---------------------------------------------------------------------------------------------------------------------
# worker2's tail: bpsw_bns_load, bpsw_reg2aln_batch, bpsw_sam_pe_batch
TAIL_SCALA, TAIL_C = 0, 1
MEM_F_NOPAIRING, MEM_F_ALL, MEM_F_NO_MULTI = 0x4, 0x8, 0x10
ALN_OK, ALN_XREF, ALN_NOCIGAR, ALN_OVERFLOW = 0, 1, 2, 3
ALN_DTYPE = np.dtype([("pos", "<i8"), ("rid", "<i4"), ("flag", "<i4"), ("is_rev", "<i4"), ("mapq", "<i4"), ("NM", "<i4"),
                      ("n_cigar", "<i4"), ("score", "<i4"), ("sub", "<i4"), ("md_len", "<i4"), ("status", "<i4")])


class TailOpt(C.Structure):  # bpsw_tail_opt_t
    _fields_ = [("mask_level", C.c_float), ("mapq_coef_len", C.c_float), ("mapq_coef_fac", C.c_int32), ("flavour", C.c_int32),
                ("rg_id", C.c_char * 64)]


class Reg2AlnJobs(C.Structure):  # bpsw_reg2aln_jobs_t
    _fields_ = [("n", C.c_int32), ("max_cigar", C.c_int32), ("max_md", C.c_int32), ("read_len", C.c_void_p), ("read_off", C.c_void_p),
                ("read_pool", C.c_void_p), ("read_pool_bytes", C.c_size_t), ("regs", C.c_void_p)]


class Pairs(C.Structure):  # bpsw_pairs_t
    _fields_ = [("group_size", C.c_int32), ("id0", C.c_int64), ("pes", PeStat * 4), ("read_len", C.c_void_p), ("read_off", C.c_void_p),
                ("read_pool", C.c_void_p), ("qual_pool", C.c_void_p), ("read_pool_bytes", C.c_size_t), ("name_off", C.c_void_p),
                ("name_pool", C.c_void_p), ("reg_cnt", C.c_void_p), ("regs", C.c_void_p)]


def default_tail_opt(flavour: int = TAIL_SCALA) -> TailOpt:
    t = TailOpt()
    load_library().bpsw_tail_opt_default(C.byref(t))
    t.flavour = flavour
    return t


def _ctx_bns_load(self, ann_off, ann_len, names=None):
    off = np.ascontiguousarray(ann_off, np.int64)
    ln = np.ascontiguousarray(ann_len, np.int32)
    blob = None if names is None else b"".join(n.encode() + b"\0" for n in names)
    _chk(self.lib, self.lib.bpsw_bns_load(self.h, int(off.shape[0]), _ptr(off), _ptr(ln), blob), "bpsw_bns_load")


def _ctx_reg2aln_batch(self, opt: Opt, topt: TailOpt, read_len, read_off, read_pool, regs, max_cigar: int = 64, max_md: int = 256):
    """memRegToAln for n (read, region) jobs -> (alns[n] ALN_DTYPE, cigar[n, max_cigar], md[n, max_md])"""
    rl = np.ascontiguousarray(read_len, np.int32); ro = np.ascontiguousarray(read_off, np.int64)
    rp = np.ascontiguousarray(read_pool, np.uint8); rg = np.ascontiguousarray(regs)
    assert rg.dtype == ALNREG_DTYPE
    j = Reg2AlnJobs()
    j.n, j.max_cigar, j.max_md = int(rg.shape[0]), int(max_cigar), int(max_md)
    j.read_len, j.read_off, j.read_pool, j.read_pool_bytes, j.regs = rl.ctypes.data, ro.ctypes.data, rp.ctypes.data, rp.size, rg.ctypes.data
    alns = np.zeros(max(j.n, 1), ALN_DTYPE)
    cig = np.zeros((max(j.n, 1), max_cigar), np.uint32)
    md = np.zeros((max(j.n, 1), max_md), np.uint8)
    _chk(self.lib, self.lib.bpsw_reg2aln_batch(self.h, C.byref(opt), C.byref(topt), C.byref(j), _ptr(alns), _ptr(cig), _ptr(md)),
         "bpsw_reg2aln_batch")
    return alns[: j.n], cig[: j.n], md[: j.n]


def _pairs_struct(g):
    st = Pairs()
    st.group_size, st.id0 = g.group_size, g.id0
    for r in range(4):
        lo, hi, failed, avg, std = g.pes[r]
        st.pes[r].low, st.pes[r].high, st.pes[r].failed, st.pes[r].avg, st.pes[r].std = int(lo), int(hi), int(failed), float(avg), float(std)
    keep = []
    for f, dt in (("read_len", np.int32), ("read_off", np.int64), ("read_pool", np.uint8), ("name_off", np.int64),
                  ("name_pool", np.uint8), ("reg_cnt", np.int32)):
        a = np.ascontiguousarray(getattr(g, f), dt)
        keep.append(a)
        setattr(st, f, a.ctypes.data)
    if g.qual_pool is not None:
        q = np.ascontiguousarray(g.qual_pool, np.uint8)
        keep.append(q)
        st.qual_pool = q.ctypes.data
    st.read_pool_bytes = int(np.asarray(g.read_pool).size)
    regs = np.ascontiguousarray(g.regs)
    assert regs.dtype == ALNREG_DTYPE
    st.regs = regs.ctypes.data
    keep.append(regs)
    return st, keep, regs


def _ctx_sam_pe_batch(self, opt: Opt, topt: TailOpt, g: "TailGroupSoA"):
    """memSamPeGroupRest -> (list of 2G SAM texts (bytes), regions as the tail leaves them)"""
    st, keep, regs = _pairs_struct(g)
    off = np.zeros(2 * g.group_size + 1, np.int64)
    out_regs = np.zeros(max(regs.shape[0], 1), ALNREG_DTYPE)
    need = C.c_size_t(0)
    cap = 1024 * max(1, 2 * g.group_size)
    while True:
        buf = np.empty(cap, np.uint8)   # the library writes off[-1] bytes
        rc = self.lib.bpsw_sam_pe_batch(self.h, C.byref(opt), C.byref(topt), C.byref(st), _ptr(buf), cap, _ptr(off), C.byref(need),
                                        _ptr(out_regs))
        if rc == -3 and need.value > cap:   # BPSW_ERR_CAPACITY
            cap = int(need.value) + 64
            continue
        _chk(self.lib, rc, "bpsw_sam_pe_batch")
        break
    text = buf[: int(off[-1])].tobytes()
    return [text[int(off[i]):int(off[i + 1])] for i in range(2 * g.group_size)], out_regs[: regs.shape[0]]


def _ctx_worker2_batch(self, opt: Opt, topt: TailOpt, g: "TailGroupSoA", rescue_mode: int = RESCUE_C):
    """rescue + tail for a group whose g.regs are the lists BEFORE the rescue -> (SAM texts, reg_cnt[2G], regs after)"""
    st, keep, regs = _pairs_struct(g)
    off = np.zeros(2 * g.group_size + 1, np.int64)
    cnt = np.zeros(2 * g.group_size, np.int32)
    need, total = C.c_size_t(0), C.c_int64(0)
    cap, rcap = 1024 * max(1, 2 * g.group_size), int(regs.shape[0]) + 8 * g.group_size + 64
    while True:
        buf = np.empty(cap, np.uint8)
        out_regs = np.empty(rcap, ALNREG_DTYPE)
        rc = self.lib.bpsw_worker2_batch(self.h, C.byref(opt), C.byref(topt), C.byref(st), rescue_mode, _ptr(buf), cap, _ptr(off),
                                         C.byref(need), _ptr(cnt), _ptr(out_regs), rcap, C.byref(total))
        if rc == -3 and (need.value > cap or total.value > rcap):
            cap, rcap = max(cap, int(need.value) + 64), max(rcap, int(total.value) + 16)
            continue
        _chk(self.lib, rc, "bpsw_worker2_batch")
        break
    text = buf[: int(off[-1])].tobytes()
    return [text[int(off[i]):int(off[i + 1])] for i in range(2 * g.group_size)], cnt, out_regs[: total.value]


def _ctx_last_tail_kernel(self):
    """(kernel ms, jobs) of the most recent tail call"""
    ms, n = C.c_float(0), C.c_int32(0)
    _chk(self.lib, self.lib.bpsw_last_tail_times(self.h, C.byref(ms), C.byref(n), None), "bpsw_last_tail_times")
    return ms.value, n.value


def _ctx_last_tail_host_ms(self):
    """(plan, device round trip, emit) ms of the most recent bpsw_sam_pe_batch"""
    h = (C.c_double * 3)()
    _chk(self.lib, self.lib.bpsw_last_tail_times(self.h, None, None, h), "bpsw_last_tail_times")
    return h[0], h[1], h[2]


TAIL_POOL_TAIL_ONLY = -1


class TailPool:
    """bpsw_tail_pool_*: the tail of worker2 on library-owned worker threads.  submit() only enqueues (the arrays of `g` and the output
    buffers are kept alive by the ticket object until collect()); collect() blocks for that group and returns what the direct call
    would have: (SAM texts, regions as the tail leaves them) for TAIL_POOL_TAIL_ONLY, (texts, reg_cnt, regs) for a rescue mode."""

    class Ticket:
        __slots__ = ("id", "mode", "g", "st", "keep", "buf", "off", "cnt", "out_regs", "n_regs_in", "cap", "rcap", "opt", "topt")

    def __init__(self, device: int = 0, workers: int = 8):
        self.lib = load_library()
        h = C.c_void_p()
        _chk(self.lib, self.lib.bpsw_tail_pool_create(int(device), int(workers), C.byref(h)), "bpsw_tail_pool_create")
        self.h = h

    @property
    def workers(self) -> int:
        return int(self.lib.bpsw_tail_pool_workers(self.h))

    def _enqueue(self, t):
        tid = C.c_int64(0)
        _chk(self.lib, self.lib.bpsw_tail_pool_submit(self.h, C.byref(t.opt), C.byref(t.topt), C.byref(t.st), t.mode, _ptr(t.buf), t.cap,
                                                      _ptr(t.off), _ptr(t.cnt), _ptr(t.out_regs), t.rcap, C.byref(tid)),
             "bpsw_tail_pool_submit")
        t.id = tid.value

    def submit(self, opt: Opt, topt: "TailOpt", g: "TailGroupSoA", rescue_mode: int = TAIL_POOL_TAIL_ONLY, text_cap: int = 0):
        t = TailPool.Ticket()
        t.mode, t.g, t.opt, t.topt = int(rescue_mode), g, opt, topt
        t.st, t.keep, regs = _pairs_struct(g)
        t.n_regs_in = int(regs.shape[0])
        t.off = np.zeros(2 * g.group_size + 1, np.int64)
        t.cnt = np.zeros(max(2 * g.group_size, 1), np.int32)
        t.cap = int(text_cap) if text_cap > 0 else 1024 * max(1, 2 * g.group_size)
        t.rcap = max(t.n_regs_in, 1) if t.mode == TAIL_POOL_TAIL_ONLY else t.n_regs_in + 8 * g.group_size + 64
        t.buf = np.empty(t.cap, np.uint8)
        t.out_regs = np.zeros(t.rcap, ALNREG_DTYPE)
        self._enqueue(t)
        return t

    def collect(self, t):
        need, total = C.c_size_t(0), C.c_int64(0)
        while True:
            rc = self.lib.bpsw_tail_pool_wait(self.h, t.id, C.byref(need), C.byref(total))
            if rc == -3 and (need.value > t.cap or total.value > t.rcap):   # BPSW_ERR_CAPACITY: once more with what it asked for
                t.cap, t.rcap = max(t.cap, int(need.value) + 64), max(t.rcap, int(total.value) + 16)
                t.buf = np.empty(t.cap, np.uint8)
                t.out_regs = np.zeros(t.rcap, ALNREG_DTYPE)
                self._enqueue(t)
                continue
            _chk(self.lib, rc, "bpsw_tail_pool_wait")
            break
        G = t.g.group_size
        text = t.buf[: int(t.off[-1])].tobytes()
        texts = [text[int(t.off[i]):int(t.off[i + 1])] for i in range(2 * G)]
        if t.mode == TAIL_POOL_TAIL_ONLY:
            return texts, t.out_regs[: t.n_regs_in]
        return texts, t.cnt[: 2 * G], t.out_regs[: total.value]

    def close(self):
        if self.h:
            self.lib.bpsw_tail_pool_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001 - interpreter shutdown
            pass


Context.bns_load = _ctx_bns_load
Context.reg2aln_batch = _ctx_reg2aln_batch
Context.sam_pe_batch = _ctx_sam_pe_batch
Context.worker2_batch = _ctx_worker2_batch
Context.last_tail_kernel = _ctx_last_tail_kernel
Context.last_tail_host_ms = _ctx_last_tail_host_ms


# ---- host-only pieces (no device): usable and testable on a CPU-only box -------------------------------------------------
def _pes_struct(pes):
    a = (PeStat * 4)()
    for r in range(4):
        lo, hi, failed, avg, std = pes[r]
        a[r].low, a[r].high, a[r].failed, a[r].avg, a[r].std = int(lo), int(hi), int(failed), float(avg), float(std)
    return a


def mark_primary_se(opt: Opt, topt: TailOpt, regs: np.ndarray, rid: int) -> np.ndarray:
    a = np.ascontiguousarray(regs.copy())
    lib = load_library()
    _chk(lib, lib.bpsw_mark_primary_se(C.byref(opt), C.byref(topt), int(a.shape[0]), _ptr(a), int(rid)), "bpsw_mark_primary_se")
    return a


def approx_mapq_se(opt: Opt, topt: TailOpt, reg) -> int:
    a = np.ascontiguousarray(np.array([reg], ALNREG_DTYPE))
    return int(load_library().bpsw_approx_mapq_se(C.byref(opt), C.byref(topt), _ptr(a)))


def mem_pair(opt: Opt, topt: TailOpt, l_pac: int, pes, regs0: np.ndarray, regs1: np.ndarray, pid: int):
    a0, a1 = np.ascontiguousarray(regs0), np.ascontiguousarray(regs1)
    out = np.zeros(5, np.int32)
    lib = load_library()
    _chk(lib, lib.bpsw_mem_pair(C.byref(opt), C.byref(topt), int(l_pac), _pes_struct(pes), int(a0.shape[0]), _ptr(a0), int(a1.shape[0]),
                                _ptr(a1), int(pid), _ptr(out)), "bpsw_mem_pair")
    return int(out[0]), int(out[1]), int(out[2]), (int(out[3]), int(out[4]))


def sort_dedup(regs: np.ndarray, mask_level_redun: float = 0.95, mode: int = RESCUE_C) -> np.ndarray:
    a = np.ascontiguousarray(regs.copy())
    n = load_library().bpsw_sort_dedup(int(a.shape[0]), _ptr(a), mask_level_redun, mode)
    if n < 0:
        raise BpswError("bpsw_sort_dedup failed")
    return a[:n]


def pe_stat(opt: Opt, topt: TailOpt, l_pac: int, reg_cnt: np.ndarray, regs: np.ndarray):
    """memPeStat -> 4 x (low, high, failed, avg, std)"""
    rc = np.ascontiguousarray(reg_cnt, np.int32); rg = np.ascontiguousarray(regs)
    pes = (PeStat * 4)()
    lib = load_library()
    _chk(lib, lib.bpsw_pe_stat(C.byref(opt), C.byref(topt), int(l_pac), int(rc.shape[0] // 2), _ptr(rc), _ptr(rg), pes), "bpsw_pe_stat")
    return [(int(pes[r].low), int(pes[r].high), int(pes[r].failed), float(pes[r].avg), float(pes[r].std)) for r in range(4)]
