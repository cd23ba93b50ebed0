"""ctypes bindings for the parity oracle -- TEST INFRASTRUCTURE ONLY.

`Oracle`  : oracle/_build/libbpsw_oracle.so, our C restatement of the Scala SW path (bpsw_oracle.c).
`Ref`     : oracle/_ref/libbwaref.so, the reference's own C sources compiled in place (oracle/Makefile
            `make ref`); present only where /root/reference was available at build time or the prebuilt
            .so travelled with the snapshot.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "_build", "libbpsw_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libbwaref.so")

ZDROP_SCALA, ZDROP_BWA = 0, 1
RESCUE_C, RESCUE_SCALA = 0, 1
KSW_XBYTE, KSW_XSTOP, KSW_XSUBO, KSW_XSTART = 0x10000, 0x20000, 0x40000, 0x80000

ALNREG_DTYPE = np.dtype([("rb", "<i8"), ("re", "<i8"), ("qb", "<i4"), ("qe", "<i4"), ("score", "<i4"),
                         ("truesc", "<i4"), ("sub", "<i4"), ("csub", "<i4"), ("sub_n", "<i4"), ("w", "<i4"),
                         ("seedcov", "<i4"), ("secondary", "<i4"), ("hash", "<u8")])


def build(ref: bool = True):
    subprocess.run(["make", "-C", HERE, "-s"], check=True)
    if ref:
        subprocess.run(["make", "-C", HERE, "-s", "ref"], check=True)


class ExtParam(C.Structure):
    _fields_ = [("left_qlen", C.c_int32), ("left_rlen", C.c_int32), ("right_qlen", C.c_int32), ("right_rlen", C.c_int32),
                ("left_qs", C.c_void_p), ("left_rs", C.c_void_p), ("right_qs", C.c_void_p), ("right_rs", C.c_void_p)] + \
               [(n, C.c_int32) for n in ("w", "o_del", "e_del", "o_ins", "e_ins", "pen_clip5", "pen_clip3", "zdrop", "h0",
                                         "reg_score", "q_beg", "idx")] + [("mat", C.c_void_p)]


class ExtRet(C.Structure):
    _fields_ = [("q_beg", C.c_int32), ("q_end", C.c_int32), ("r_beg", C.c_int64), ("r_end", C.c_int64),
                ("score", C.c_int32), ("true_score", C.c_int32), ("width", C.c_int32), ("idx", C.c_int32)]


class PeStat(C.Structure):
    _fields_ = [("low", C.c_int32), ("high", C.c_int32), ("failed", C.c_int32), ("pad_", C.c_int32),
                ("avg", C.c_double), ("std", C.c_double)]


class Opt(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("a", "b", "o_del", "e_del", "o_ins", "e_ins", "pen_unpaired", "pen_clip5",
                                         "pen_clip3", "w", "zdrop", "T", "flag", "min_seed_len", "max_ins",
                                         "max_matesw")] + [("mask_level_redun", C.c_float), ("mat", C.c_int8 * 25),
                                                           ("pad_", C.c_int8 * 3)]


def _vp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _get_seq(fn, l_pac, pac, beg, end):
    pac = np.ascontiguousarray(pac, np.uint8)
    cap = int(abs(int(end) - int(beg))) + 8
    out = np.zeros(cap, np.uint8)
    n = fn(C.c_int64(l_pac), _vp(pac), C.c_int64(beg), C.c_int64(end), _vp(out), C.c_int64(cap))
    assert n >= 0
    return out[:n].copy()


def default_mat(a: int = 1, b: int = 4) -> np.ndarray:
    m = np.full((5, 5), -1, dtype=np.int8)
    for i in range(4):
        for j in range(4):
            m[i, j] = a if i == j else -b
    return m.reshape(25)


class Oracle:
    def __init__(self, path: str = ORACLE_SO):
        if not os.path.exists(path):
            build(ref=False)
        self.lib = C.CDLL(path)
        self.lib.orc_wire_size.restype = C.c_size_t
        self.lib.orc_wire_pack.restype = C.c_size_t
        self.lib.orc_matesw_group.restype = C.c_int64
        self.lib.orc_sw_global.restype = C.c_int
        self.lib.orc_bns_get_seq.restype = C.c_int64

    def default_opt(self) -> Opt:
        o = Opt()
        self.lib.orc_opt_default(C.byref(o))
        return o

    def sw_extend(self, query, target, mat, o_del, e_del, o_ins, e_ins, w, end_bonus, zdrop, h0, zdrop_mode=ZDROP_SCALA):
        q = np.ascontiguousarray(query, np.uint8)
        t = np.ascontiguousarray(target, np.uint8)
        m = np.ascontiguousarray(mat, np.int8)
        out = np.zeros(6, np.int32)
        cells = C.c_int64(0)
        self.lib.orc_sw_extend(C.c_int(q.size), _vp(q), C.c_int(t.size), _vp(t), 5, _vp(m), o_del, e_del, o_ins, e_ins,
                               w, end_bonus, zdrop, h0, zdrop_mode, _vp(out), C.byref(cells))
        return out, cells.value

    def wire_extend(self, wire, mat=None, zdrop=100, zdrop_mode=ZDROP_SCALA):
        """decode a boundary-2 batch and run extension() per task -> (int16[10n], cells)"""
        wire = np.ascontiguousarray(wire, np.uint8)
        m = np.ascontiguousarray(default_mat() if mat is None else mat, np.int8)
        n = int(np.frombuffer(wire[8:12].tobytes(), "<i4")[0])
        out = np.zeros(max(10 * n, 1), np.int16)
        cells = C.c_int64(0)
        rc = self.lib.orc_wire_extend(_vp(wire), C.c_size_t(wire.size), _vp(m), zdrop, zdrop_mode, _vp(out), C.byref(cells))
        if rc < 0:
            raise ValueError("malformed wire batch")
        return out[: 10 * n], cells.value

    def wire_extend_sides(self, wire, mat=None, zdrop=100, zdrop_mode=ZDROP_SCALA):
        """wire_extend that also reports the DP cells of every side -> (int16[10n], cells, int64[n, 2])"""
        wire = np.ascontiguousarray(wire, np.uint8)
        m = np.ascontiguousarray(default_mat() if mat is None else mat, np.int8)
        n = int(np.frombuffer(wire[8:12].tobytes(), "<i4")[0])
        out = np.zeros(max(10 * n, 1), np.int16)
        sides = np.zeros((max(n, 1), 2), np.int64)
        cells = C.c_int64(0)
        rc = self.lib.orc_wire_extend_sides(_vp(wire), C.c_size_t(wire.size), _vp(m), zdrop, zdrop_mode, _vp(out), C.byref(cells), _vp(sides))
        if rc < 0:
            raise ValueError("malformed wire batch")
        return out[: 10 * n], cells.value, sides[:n]

    def wire_pack_soa(self, soa, mat=None):
        """pack an ExtTaskSoA through the ORACLE's packer (MemChainToAlignBatched.scala:76-172)"""
        m = np.ascontiguousarray(default_mat() if mat is None else mat, np.int8)
        n = soa.n
        arr = (ExtParam * n)()
        base = soa.pool.ctypes.data
        for i in range(n):
            p = arr[i]
            p.left_qlen, p.left_rlen = int(soa.left_qlen[i]), int(soa.left_rlen[i])
            p.right_qlen, p.right_rlen = int(soa.right_qlen[i]), int(soa.right_rlen[i])
            p.left_qs, p.left_rs = base + int(soa.left_q_off[i]), base + int(soa.left_r_off[i])
            p.right_qs, p.right_rs = base + int(soa.right_q_off[i]), base + int(soa.right_r_off[i])
            p.w, p.o_del, p.e_del, p.o_ins, p.e_ins = soa.w, soa.o_del, soa.e_del, soa.o_ins, soa.e_ins
            p.pen_clip5, p.pen_clip3, p.zdrop = soa.pen_clip5, soa.pen_clip3, 100
            p.h0, p.reg_score, p.q_beg, p.idx = int(soa.h0[i]), int(soa.reg_score[i]), int(soa.q_beg[i]), int(soa.idx[i])
            p.mat = m.ctypes.data
        size = self.lib.orc_wire_size(n, arr)
        buf = np.zeros(size, np.uint8)
        got = self.lib.orc_wire_pack(n, arr, _vp(buf), C.c_size_t(size))
        assert got == size
        return buf

    def sw_align2(self, query, target, opt: Opt, xtra, two_pass=True):
        q = np.ascontiguousarray(query, np.uint8)
        t = np.ascontiguousarray(target, np.uint8)
        out = np.zeros(7, np.int32)
        cells = C.c_int64(0)
        fn = self.lib.orc_sw_align2 if two_pass else self.lib.orc_sw_align
        fn(C.c_int(q.size), _vp(q), C.c_int(t.size), _vp(t), 5, C.byref(opt, Opt.mat.offset), opt.a, opt.b, opt.o_del,
           opt.e_del, opt.o_ins, opt.e_ins, C.c_int(xtra), _vp(out), C.byref(cells))
        return out, cells.value

    def sw_align2_jobs(self, opt: Opt, xtra, q_len, t_len, q_off, t_off, q_rev, q_pool, t_pool):
        n = len(q_len)
        out = np.zeros((n, 7), np.int32)
        cells = 0
        for i in range(n):
            q = q_pool[q_off[i]: q_off[i] + q_len[i]]
            if q_rev[i]:
                q = np.where(q[::-1] < 4, 3 - q[::-1], 4).astype(np.uint8)  # MemSamPe.scala:1175-1184
            o, c = self.sw_align2(q, t_pool[t_off[i]: t_off[i] + t_len[i]], opt, xtra)
            out[i] = o
            cells += c
        return out, cells

    def bns_get_seq(self, l_pac, pac, beg, end):
        """bnsGetSeq (util/BNTSeqUtil.scala:37-79) -> uint8 bases (empty when the window bridges the strands)"""
        return _get_seq(self.lib.orc_bns_get_seq, l_pac, pac, beg, end)

    def chain2aln_batch(self, opt: Opt, pac, b, zdrop_mode=ZDROP_SCALA):
        """memChainToAlnBatched (MemChainToAlignBatched.scala:380-616) per read -> (cnt[n], regs, n_ext, cells)"""
        pac = np.ascontiguousarray(pac, np.uint8)
        n = int(b.read_len.shape[0])
        cap = int(b.seed_len.shape[0]) + 8
        out_cnt = np.zeros(max(n, 1), np.int32)
        out = np.zeros(cap, ALNREG_DTYPE)
        n_ext, cells = C.c_int64(0), C.c_int64(0)
        self.lib.orc_chain2aln_batch.restype = C.c_int64
        tot = self.lib.orc_chain2aln_batch(C.byref(opt), C.c_int(zdrop_mode), C.c_int64(b.l_pac), _vp(pac), C.c_int(n), _vp(b.read_len),
                                           _vp(b.read_off), _vp(b.read_pool), _vp(b.chain_cnt), _vp(b.seed_cnt), _vp(b.seed_rbeg),
                                           _vp(b.seed_qbeg), _vp(b.seed_len), _vp(out_cnt), _vp(out), C.c_int64(cap),
                                           C.byref(n_ext), C.byref(cells))
        assert tot >= 0
        return out_cnt[:n], out[:tot], n_ext.value, cells.value

    def sw_global(self, query, target, mat, o_del, e_del, o_ins, e_ins, w):
        q = np.ascontiguousarray(query, np.uint8)
        t = np.ascontiguousarray(target, np.uint8)
        m = np.ascontiguousarray(mat, np.int8)
        cap = q.size + t.size + 4
        cig = np.zeros(cap, np.uint32)
        nc = C.c_int(0)
        score = self.lib.orc_sw_global(C.c_int(q.size), _vp(q), C.c_int(t.size), _vp(t), 5, _vp(m), o_del, e_del, o_ins,
                                       e_ins, w, C.byref(nc), _vp(cig), cap)
        return score, cig[: nc.value].copy()

    def sort_dedup(self, regs: np.ndarray, mask_level_redun=0.95, mode=RESCUE_C) -> np.ndarray:
        a = np.ascontiguousarray(regs.copy())
        n = self.lib.orc_sort_dedup(C.c_int(a.shape[0]), _vp(a), C.c_float(mask_level_redun), mode)
        return a[:n]

    def matesw_group(self, opt: Opt, g, mode=RESCUE_C):
        """g: bpsw_hip.RescueGroupSoA (same flat layout)."""
        pes = (PeStat * 4)()
        for r in range(4):
            pes[r].low, pes[r].high, pes[r].failed, pes[r].avg, pes[r].std = g.pes[r]
        out_cnt = np.zeros(2 * g.group_size, np.int32)
        cap = int(g.regs.shape[0] + 4 * g.ref_rb.shape[0] + 16)
        out = np.zeros(cap, ALNREG_DTYPE)
        n_sw, cells = C.c_int64(0), C.c_int64(0)
        total = self.lib.orc_matesw_group(C.byref(opt), C.c_int64(g.l_pac), pes, g.group_size, _vp(g.seq_len), _vp(g.seq_off),
                                          _vp(g.seq_pool), _vp(g.reg_cnt), _vp(g.regs), _vp(g.ref_cnt), _vp(g.ref_rb),
                                          _vp(g.ref_re), _vp(g.ref_len), _vp(g.ref_off), _vp(g.ref_pool), mode,
                                          _vp(out_cnt), _vp(out), C.c_int64(cap), C.byref(n_sw), C.byref(cells))
        assert total >= 0
        return out_cnt, out[:total], n_sw.value, cells.value


class KswrT(C.Structure):  # kswr_t, native/ksw.h
    _fields_ = [("score", C.c_int), ("te", C.c_int), ("qe", C.c_int), ("score2", C.c_int), ("te2", C.c_int),
                ("tb", C.c_int), ("qb", C.c_int)]


class Ref:
    """The reference's own C (bwa-0.7.8 as vendored under src/main/native), compiled in place."""

    def __init__(self, path: str = REF_SO):
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = C.CDLL(path)
        self.lib.ksw_align2.restype = KswrT
        self.lib.ref_group_matesw_flat.restype = C.c_int64
        if hasattr(self.lib, "ref_bns_get_seq"):
            self.lib.ref_bns_get_seq.restype = C.c_int64
        self.libc = C.CDLL(None)
        self.libc.free.argtypes = [C.c_void_p]

    @staticmethod
    def available() -> bool:
        return os.path.exists(REF_SO)

    def bns_get_seq(self, l_pac, pac, beg, end):
        """bns_get_seq (native/bntseq.c:355-376)"""
        return _get_seq(self.lib.ref_bns_get_seq, l_pac, pac, beg, end)

    def chain2aln_batch(self, opt: Opt, pac, b):
        """mem_chain2aln (native/bwamem.c:552-672) per chain of every read -> (cnt[n], regs)"""
        ints = np.array([opt.a, opt.b, opt.o_del, opt.e_del, opt.o_ins, opt.e_ins, opt.pen_unpaired, opt.pen_clip5, opt.pen_clip3,
                         opt.w, opt.zdrop, opt.T, opt.flag, opt.min_seed_len, opt.max_ins, opt.max_matesw], np.int32)
        mat = np.array(list(opt.mat), np.int8)
        pac = np.ascontiguousarray(pac, np.uint8)
        n = int(b.read_len.shape[0])
        cap = int(b.seed_len.shape[0]) + 8
        out_cnt = np.zeros(max(n, 1), np.int32)
        out = np.zeros(cap, ALNREG_DTYPE)
        self.lib.ref_chain2aln_batch.restype = C.c_int64
        tot = self.lib.ref_chain2aln_batch(_vp(ints), _vp(mat), C.c_int64(b.l_pac), _vp(pac), C.c_int(n), _vp(b.read_len),
                                           _vp(b.read_off), _vp(b.read_pool), _vp(b.chain_cnt), _vp(b.seed_cnt), _vp(b.seed_rbeg),
                                           _vp(b.seed_qbeg), _vp(b.seed_len), _vp(out_cnt), _vp(out), C.c_int64(cap))
        assert tot >= 0
        return out_cnt[:n], out[:tot]

    def ksw_extend2(self, query, target, mat, o_del, e_del, o_ins, e_ins, w, end_bonus, zdrop, h0):
        q = np.ascontiguousarray(query, np.uint8)
        t = np.ascontiguousarray(target, np.uint8)
        m = np.ascontiguousarray(mat, np.int8)
        qle, tle, gtle, gscore, max_off = (C.c_int(0) for _ in range(5))
        score = self.lib.ksw_extend2(C.c_int(q.size), _vp(q), C.c_int(t.size), _vp(t), 5, _vp(m), o_del, e_del, o_ins,
                                     e_ins, w, end_bonus, zdrop, h0, C.byref(qle), C.byref(tle), C.byref(gtle),
                                     C.byref(gscore), C.byref(max_off))
        return np.array([score, qle.value, tle.value, gtle.value, gscore.value, max_off.value], np.int32)

    def ksw_align2(self, query, target, mat, o_del, e_del, o_ins, e_ins, xtra):
        q = np.ascontiguousarray(query, np.uint8).copy()  # reversed in place and restored
        t = np.ascontiguousarray(target, np.uint8).copy()
        m = np.ascontiguousarray(mat, np.int8)
        r = self.lib.ksw_align2(C.c_int(q.size), _vp(q), C.c_int(t.size), _vp(t), 5, _vp(m), o_del, e_del, o_ins, e_ins,
                                C.c_int(xtra), None)
        return np.array([r.score, r.te, r.qe, r.score2, r.te2, r.tb, r.qb], np.int32)

    def ksw_global2(self, query, target, mat, o_del, e_del, o_ins, e_ins, w):
        q = np.ascontiguousarray(query, np.uint8)
        t = np.ascontiguousarray(target, np.uint8)
        m = np.ascontiguousarray(mat, np.int8)
        nc = C.c_int(0)
        cig = C.POINTER(C.c_uint32)()
        score = self.lib.ksw_global2(C.c_int(q.size), _vp(q), C.c_int(t.size), _vp(t), 5, _vp(m), o_del, e_del, o_ins,
                                     e_ins, w, C.byref(nc), C.byref(cig))
        out = np.array([cig[i] for i in range(nc.value)], np.uint32)
        self.libc.free(C.cast(cig, C.c_void_p))
        return score, out

    def sort_dedup(self, regs: np.ndarray, mask_level_redun=0.95) -> np.ndarray:
        a = np.ascontiguousarray(regs.copy())
        n = self.lib.mem_sort_and_dedup(C.c_int(a.shape[0]), _vp(a), C.c_float(mask_level_redun))
        return a[:n]

    def extend_batch(self, soa, mat, zdrop=100):
        """extension() control (MemChainToAlignBatched.scala:789-883) around the reference ksw_extend2, looped in C"""
        m = np.ascontiguousarray(mat, np.int8)
        out = np.zeros((soa.n, 7), np.int32)
        self.lib.ref_extend_batch(C.c_int(soa.n), _vp(soa.left_qlen), _vp(soa.left_rlen), _vp(soa.right_qlen), _vp(soa.right_rlen),
                                  _vp(soa.left_q_off), _vp(soa.left_r_off), _vp(soa.right_q_off), _vp(soa.right_r_off),
                                  _vp(soa.reg_score), _vp(soa.q_beg), _vp(soa.h0), _vp(soa.pool), _vp(m), soa.o_del, soa.e_del,
                                  soa.o_ins, soa.e_ins, soa.w, soa.pen_clip5, soa.pen_clip3, zdrop, _vp(out))
        return out

    def align2_batch(self, mat, o_del, e_del, o_ins, e_ins, xtra, q_len, t_len, q_off, t_off, q_rev, q_pool, t_pool):
        """ksw_align2 (the SSE2 kernels jniNative.so runs) over SoA jobs, looped in C"""
        m = np.ascontiguousarray(mat, np.int8)
        n = len(q_len)
        out = np.zeros((n, 7), np.int32)
        a = [np.ascontiguousarray(x, dt) for x, dt in ((q_len, np.int32), (t_len, np.int32), (q_off, np.int64), (t_off, np.int64),
                                                      (q_rev, np.uint8), (q_pool, np.uint8), (t_pool, np.uint8))]
        self.lib.ref_align2_batch(C.c_int(n), *[_vp(x) for x in a], _vp(m), o_del, e_del, o_ins, e_ins, C.c_int(xtra), _vp(out))
        return out

    def matesw_group(self, opt: Opt, g):
        ints = np.array([opt.a, opt.b, opt.o_del, opt.e_del, opt.o_ins, opt.e_ins, opt.pen_unpaired, opt.pen_clip5,
                         opt.pen_clip3, opt.w, opt.zdrop, opt.T, opt.flag, opt.min_seed_len, opt.max_ins,
                         opt.max_matesw], np.int32)
        mat = np.array(list(opt.mat), np.int8)
        pes = (PeStat * 4)()
        for r in range(4):
            pes[r].low, pes[r].high, pes[r].failed, pes[r].avg, pes[r].std = g.pes[r]
        out_cnt = np.zeros(2 * g.group_size, np.int32)
        cap = int(g.regs.shape[0] + 4 * g.ref_rb.shape[0] + 16)
        out = np.zeros(cap, ALNREG_DTYPE)
        total = self.lib.ref_group_matesw_flat(_vp(ints), C.c_float(opt.mask_level_redun), _vp(mat), C.c_int64(g.l_pac), pes,
                                               g.group_size, _vp(g.seq_len), _vp(g.seq_off), _vp(g.seq_pool), _vp(g.reg_cnt),
                                               _vp(g.regs), _vp(g.ref_cnt), _vp(g.ref_rb), _vp(g.ref_re), _vp(g.ref_len),
                                               _vp(g.ref_off), _vp(g.ref_pool), _vp(out_cnt), _vp(out), C.c_int64(cap))
        assert total >= 0
        return out_cnt, out[:total]


# ---------------------------------------------------------------------------------------------------------------------
# worker2's tail (bpsw_oracle_tail.c / the tail shims of ref_shim.c)
TAIL_SCALA, TAIL_C = 0, 1
MEM_F_NOPAIRING, MEM_F_ALL, MEM_F_NO_MULTI, MEM_F_NO_RESCUE = 0x4, 0x8, 0x10, 0x20

ALN_DTYPE = np.dtype([("pos", "<i8"), ("rid", "<i4"), ("flag", "<i4"), ("is_rev", "<i4"), ("mapq", "<i4"), ("NM", "<i4"),
                      ("n_cigar", "<i4"), ("score", "<i4"), ("sub", "<i4"), ("md_len", "<i4"), ("status", "<i4")])


class TailOpt(C.Structure):
    _fields_ = [("mask_level", C.c_float), ("mapq_coef_len", C.c_float), ("mapq_coef_fac", C.c_int32), ("pad_", C.c_int32),
                ("rg_id", C.c_char * 64)]


def _ints_of(opt: Opt):
    return np.array([opt.a, opt.b, opt.o_del, opt.e_del, opt.o_ins, opt.e_ins, opt.pen_unpaired, opt.pen_clip5, opt.pen_clip3,
                     opt.w, opt.zdrop, opt.T, opt.flag, opt.min_seed_len, opt.max_ins, opt.max_matesw], np.int32)


def _pes_of(pes_list):
    pes = (PeStat * 4)()
    for r in range(4):
        pes[r].low, pes[r].high, pes[r].failed, pes[r].avg, pes[r].std = pes_list[r]
    return pes


def _split_text(text: bytes, off: np.ndarray):
    return [text[int(off[i]):int(off[i + 1])] for i in range(len(off) - 1)]


def _orc_default_tail_opt(self) -> TailOpt:
    t = TailOpt()
    self.lib.orc_tail_opt_default(C.byref(t))
    return t


def _orc_mark_primary(self, opt, topt, regs, rid, flavour=TAIL_SCALA):
    a = np.ascontiguousarray(regs.copy())
    self.lib.orc_mark_primary_se(C.byref(opt), C.byref(topt), C.c_int(a.shape[0]), _vp(a), C.c_int64(rid), C.c_int(flavour))
    return a


def _orc_approx_mapq(self, opt, topt, reg, flavour=TAIL_SCALA):
    a = np.ascontiguousarray(np.array([reg], ALNREG_DTYPE))
    return int(self.lib.orc_approx_mapq_se(C.byref(opt), C.byref(topt), _vp(a), C.c_int(flavour)))


def _orc_mem_pair(self, opt, l_pac, pes, a0, a1, pid, flavour=TAIL_SCALA):
    a0 = np.ascontiguousarray(a0); a1 = np.ascontiguousarray(a1)
    sub, n_sub = C.c_int(0), C.c_int(0)
    z = (C.c_int * 2)(-1, -1)
    ret = self.lib.orc_mem_pair(C.byref(opt), C.c_int64(l_pac), _pes_of(pes), C.c_int(a0.shape[0]), _vp(a0), C.c_int(a1.shape[0]),
                                _vp(a1), C.c_int64(pid), C.c_int(flavour), C.byref(sub), C.byref(n_sub), z)
    return int(ret), sub.value, n_sub.value, (z[0], z[1])


def _orc_reg2aln_batch(self, opt, topt, pac, l_pac, ann_off, ann_len, read_len, read_off, read_pool, regs, flavour=TAIL_SCALA,
                       cigar_cap=64, md_cap=256):
    """memRegToAln (MemRegToADAMSAM.scala:172-313) for n (read, region) jobs -> (alns[n], cigar[n, cigar_cap], md[n, md_cap])"""
    pac = np.ascontiguousarray(pac, np.uint8)
    ann_off = np.ascontiguousarray(ann_off, np.int64); ann_len = np.ascontiguousarray(ann_len, np.int32)
    regs = np.ascontiguousarray(regs)
    n = int(regs.shape[0])
    alns = np.zeros(n, ALN_DTYPE)
    cig = np.zeros((n, cigar_cap), np.uint32)
    md = np.zeros((n, md_cap), np.uint8)
    for j in range(n):
        q = np.ascontiguousarray(read_pool[int(read_off[j]):int(read_off[j]) + int(read_len[j])])
        self.lib.orc_reg2aln(C.byref(opt), C.byref(topt), C.c_int(ann_off.shape[0]), _vp(ann_off), _vp(ann_len), C.c_int64(l_pac),
                             _vp(pac), C.c_int(q.size), _vp(q), C.c_void_p(regs.ctypes.data + j * regs.itemsize), C.c_int(flavour),
                             C.c_void_p(alns.ctypes.data + j * alns.itemsize), C.c_void_p(cig.ctypes.data + 4 * j * cigar_cap),
                             C.c_int(cigar_cap), C.c_void_p(md.ctypes.data + j * md_cap), C.c_int(md_cap))
    return alns, cig, md


def _orc_sam_pe_batch(self, opt, topt, pac, g, flavour=TAIL_SCALA):
    """memSamPeGroupRest (MemSamPe.scala:1390-1612) -> (list of 2G SAM texts, regs as the tail leaves them, n_reg2aln)"""
    pac = np.ascontiguousarray(pac, np.uint8)
    regs = np.ascontiguousarray(g.regs.copy())
    cap = 4096 * max(1, 2 * g.group_size)
    off = np.zeros(2 * g.group_size + 1, np.int64)
    nra = C.c_int64(0)
    self.lib.orc_sam_pe_batch.restype = C.c_int64
    while True:
        buf = np.zeros(cap, np.uint8)
        regs = np.ascontiguousarray(g.regs.copy())
        tot = self.lib.orc_sam_pe_batch(C.byref(opt), C.byref(topt), C.c_int(g.ann_off.shape[0]), _vp(g.ann_off), _vp(g.ann_len),
                                        _vp(g.ann_name_off), _vp(g.ann_name_pool), C.c_int64(g.l_pac), _vp(pac), _pes_of(g.pes),
                                        C.c_int(g.group_size), C.c_int64(g.id0), _vp(g.read_len), _vp(g.read_off), _vp(g.read_pool),
                                        _vp(g.qual_pool), _vp(g.name_off), _vp(g.name_pool), _vp(g.reg_cnt), _vp(regs),
                                        C.c_int(flavour), _vp(buf), C.c_int64(cap), _vp(off), C.byref(nra))
        if tot >= 0:
            break
        cap = -tot + 64
    return _split_text(buf[:tot].tobytes(), off), regs, nra.value


Oracle.default_tail_opt = _orc_default_tail_opt
Oracle.mark_primary = _orc_mark_primary
Oracle.approx_mapq = _orc_approx_mapq
Oracle.mem_pair = _orc_mem_pair
Oracle.reg2aln_batch = _orc_reg2aln_batch
Oracle.sam_pe_batch = _orc_sam_pe_batch


def _tail3(topt: TailOpt):
    return np.array([topt.mask_level, topt.mapq_coef_len, float(topt.mapq_coef_fac)], np.float32)


def _ref_mark_primary(self, opt, topt, regs, rid):
    a = np.ascontiguousarray(regs.copy())
    self.lib.ref_mark_primary_se(_vp(_ints_of(opt)), _vp(np.array(list(opt.mat), np.int8)), _vp(_tail3(topt)), C.c_int(a.shape[0]),
                                 _vp(a), C.c_int64(rid))
    return a


def _ref_approx_mapq(self, opt, topt, reg):
    a = np.ascontiguousarray(np.array([reg], ALNREG_DTYPE))
    return int(self.lib.ref_approx_mapq_se(_vp(_ints_of(opt)), _vp(np.array(list(opt.mat), np.int8)), _vp(_tail3(topt)), _vp(a)))


def _ref_mem_pair(self, opt, l_pac, pes, a0, a1, pid):
    a0 = np.ascontiguousarray(a0); a1 = np.ascontiguousarray(a1)
    out = np.zeros(5, np.int32)
    self.lib.ref_mem_pair(_vp(_ints_of(opt)), _vp(np.array(list(opt.mat), np.int8)), C.c_int64(l_pac), _pes_of(pes),
                          C.c_int(a0.shape[0]), _vp(a0), C.c_int(a1.shape[0]), _vp(a1), C.c_int(pid), _vp(out))
    return int(out[0]), int(out[1]), int(out[2]), (int(out[3]), int(out[4]))


def _ref_reg2aln_batch(self, opt, topt, pac, l_pac, ann_off, ann_len, read_len, read_off, read_pool, regs, cigar_cap=64, md_cap=256):
    """mem_reg2aln (native/bwamem.c:949-1021) -> (alns[n] as ALN_DTYPE with status 0, cigar, md)"""
    pac = np.ascontiguousarray(pac, np.uint8)
    ann_off = np.ascontiguousarray(ann_off, np.int64); ann_len = np.ascontiguousarray(ann_len, np.int32)
    regs = np.ascontiguousarray(regs)
    n = int(regs.shape[0])
    raw = np.zeros((n, 10), np.int64)
    cig = np.zeros((n, cigar_cap), np.uint32)
    md = np.zeros((n, md_cap), np.uint8)
    rl = np.ascontiguousarray(read_len, np.int32); ro = np.ascontiguousarray(read_off, np.int64)
    rp = np.ascontiguousarray(read_pool, np.uint8)
    self.lib.ref_reg2aln_batch(_vp(_ints_of(opt)), _vp(np.array(list(opt.mat), np.int8)), _vp(_tail3(topt)), C.c_int64(l_pac), _vp(pac),
                               C.c_int(ann_off.shape[0]), _vp(ann_off), _vp(ann_len), C.c_int(n), _vp(rl), _vp(ro), _vp(rp), _vp(regs),
                               _vp(raw), _vp(cig), C.c_int(cigar_cap), _vp(md), C.c_int(md_cap))
    alns = np.zeros(n, ALN_DTYPE)
    for k, f in enumerate(("pos", "rid", "flag", "is_rev", "mapq", "NM", "n_cigar", "score", "sub", "md_len")):
        alns[f] = raw[:, k]
    return alns, cig, md


def _ref_sam_pe_batch(self, opt, topt, pac, g, no_rescue=True):
    """mem_sam_pe (native/bwamem_pair.c:361-453) per pair -> list of 2G SAM texts (topt.rg_id -> the reference's bwa_rg_id)"""
    pac = np.ascontiguousarray(pac, np.uint8)
    self.lib.ref_set_rg_id(C.c_char_p(bytes(topt.rg_id)))
    ints = _ints_of(opt)
    if no_rescue:
        ints[12] |= MEM_F_NO_RESCUE
    cap = 4096 * max(1, 2 * g.group_size)
    off = np.zeros(2 * g.group_size + 1, np.int64)
    self.lib.ref_sam_pe_batch.restype = C.c_int64
    while True:
        buf = np.zeros(cap, np.uint8)
        tot = self.lib.ref_sam_pe_batch(_vp(ints), _vp(np.array(list(opt.mat), np.int8)), _vp(_tail3(topt)), C.c_int64(g.l_pac), _vp(pac),
                                        C.c_int(g.ann_off.shape[0]), _vp(g.ann_off), _vp(g.ann_len), _vp(g.ann_name_off),
                                        _vp(g.ann_name_pool), _pes_of(g.pes), C.c_int(g.group_size), C.c_int64(g.id0), _vp(g.read_len),
                                        _vp(g.read_off), _vp(g.read_pool), _vp(g.qual_pool), _vp(g.name_off), _vp(g.name_pool),
                                        _vp(g.reg_cnt), _vp(np.ascontiguousarray(g.regs)), _vp(buf), C.c_int64(cap), _vp(off))
        if tot >= 0:
            break
        cap = -tot + 64
    return _split_text(buf[:tot].tobytes(), off)


Ref.mark_primary = _ref_mark_primary
Ref.approx_mapq = _ref_approx_mapq
Ref.mem_pair = _ref_mem_pair
Ref.reg2aln_batch = _ref_reg2aln_batch
Ref.sam_pe_batch = _ref_sam_pe_batch


def _pes_list(pes):
    return [(int(pes[r].low), int(pes[r].high), int(pes[r].failed), float(pes[r].avg), float(pes[r].std)) for r in range(4)]


def _orc_pe_stat(self, opt, topt, l_pac, reg_cnt, regs, flavour=TAIL_SCALA):
    """memPeStat (MemSamPe.scala:117-260) -> 4 x (low, high, failed, avg, std)"""
    rc = np.ascontiguousarray(reg_cnt, np.int32); rg = np.ascontiguousarray(regs)
    pes = (PeStat * 4)()
    self.lib.orc_pe_stat(C.byref(opt), C.byref(topt), C.c_int64(l_pac), C.c_int(rc.shape[0] // 2), _vp(rc), _vp(rg), C.c_int(flavour), pes)
    return _pes_list(pes)


def _ref_pestat(self, opt, topt, l_pac, reg_cnt, regs):
    """mem_pestat (native/bwamem_pair.c:50-112)"""
    rc = np.ascontiguousarray(reg_cnt, np.int32); rg = np.ascontiguousarray(regs)
    pes = (PeStat * 4)()
    self.lib.ref_pestat(_vp(_ints_of(opt)), _vp(np.array(list(opt.mat), np.int8)), _vp(_tail3(topt)), C.c_int64(l_pac),
                        C.c_int(rc.shape[0] // 2), _vp(rc), _vp(rg), pes)
    return _pes_list(pes)


Oracle.pe_stat = _orc_pe_stat
Ref.pestat = _ref_pestat
