"""Seeded synthetic workloads (ctypes over libbpsw_synth.so, csrc/bpsw_synth.cpp).

Concretises the BASELINE.json configs the way SURVEY.md 8(d) lays out; inputs only, never measured.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import SYNTH_PATH, ExtTaskSoA, BpswError

CONFIG_SEED_BASE = 0xB5A30000  # seed = 0xB5A3_0000 + config#


class _ExtCfg(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_reads", C.c_int32), ("read_len", C.c_int32), ("sub_rate", C.c_double),
                ("indel_rate", C.c_double), ("n_rate", C.c_double), ("tail_frac", C.c_double),
                ("tail_sub_rate", C.c_double), ("tail_indel_rate", C.c_double)] + \
               [(n, C.c_int32) for n in ("a", "o_del", "e_del", "o_ins", "e_ins", "w", "min_seed_len", "second_seed")]


class _SwCfg(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_jobs", C.c_int32), ("read_len", C.c_int32), ("win_min", C.c_int32),
                ("win_max", C.c_int32), ("sub_rate", C.c_double), ("indel_rate", C.c_double), ("n_rate", C.c_double),
                ("unrelated_frac", C.c_double), ("decoy_frac", C.c_double), ("rev_frac", C.c_double)]


_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(SYNTH_PATH):
            raise BpswError(f"{SYNTH_PATH} not found: run `make -C {os.path.dirname(os.path.dirname(SYNTH_PATH))}`")
        _lib = C.CDLL(SYNTH_PATH)
        _lib.bpsw_synth_ext_tasks.restype = C.c_int
        _lib.bpsw_synth_sw_jobs.restype = C.c_int
    return _lib


def ext_tasks(n_reads: int, read_len: int = 150, sub_rate: float = 0.01, indel_rate: float = 0.001,
              n_rate: float = 0.001, tail_frac: float = 0.0, tail_sub_rate: float = 0.2, tail_indel_rate: float = 0.02,
              seed: int = CONFIG_SEED_BASE + 3, second_seed: bool = True) -> ExtTaskSoA:
    """Extension tasks for `n_reads` synthetic reads (default = config-3 error model, 2x150 bp)."""
    lib = _load()
    cfg = _ExtCfg(seed=seed, n_reads=n_reads, read_len=read_len, sub_rate=sub_rate, indel_rate=indel_rate,
                  n_rate=n_rate, tail_frac=tail_frac, tail_sub_rate=tail_sub_rate, tail_indel_rate=tail_indel_rate,
                  a=1, o_del=6, e_del=1, o_ins=6, e_ins=1, w=100, min_seed_len=19, second_seed=int(second_seed))
    cap = 2 * n_reads + 1
    i32 = lambda: np.zeros(cap, dtype=np.int32)
    i64 = lambda: np.zeros(cap, dtype=np.int64)
    f = dict(left_qlen=i32(), left_rlen=i32(), right_qlen=i32(), right_rlen=i32(), left_q_off=i64(), left_r_off=i64(),
             right_q_off=i64(), right_r_off=i64(), reg_score=i32(), q_beg=i32(), h0=i32(), idx=i32())
    pool_cap = int(cap) * (4 * read_len + 16)
    pool = np.zeros(pool_cap, dtype=np.uint8)
    used = C.c_size_t(0)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    n = lib.bpsw_synth_ext_tasks(C.byref(cfg), vp(f["left_qlen"]), vp(f["left_rlen"]), vp(f["right_qlen"]),
                                 vp(f["right_rlen"]), vp(f["left_q_off"]), vp(f["left_r_off"]), vp(f["right_q_off"]),
                                 vp(f["right_r_off"]), vp(f["reg_score"]), vp(f["q_beg"]), vp(f["h0"]), vp(f["idx"]),
                                 vp(pool), C.c_size_t(pool_cap), C.byref(used))
    if n < 0:
        raise BpswError("synthetic pool too small")
    return ExtTaskSoA(pool=pool[: max(used.value, 1)].copy(), **{k: v[:n].copy() for k, v in f.items()})


def hash_pac(l_pac: int, seed: int = CONFIG_SEED_BASE + 77) -> np.ndarray:
    """A 2-bit .pac of l_pac i.i.d. bases from a counter hash (any window can be regenerated from its coordinates);
    chr21-sized references take a fraction of a second"""
    lib = _load()
    pac = np.zeros((l_pac + 3) // 4, np.uint8)
    lib.bpsw_synth_hash_pac(C.c_int64(l_pac), C.c_uint64(seed), pac.ctypes.data_as(C.c_void_p))
    return pac


def ext_tasks_ref(n_reads: int, pac: np.ndarray, l_pac: int, read_len: int = 150, sub_rate: float = 0.01, indel_rate: float = 0.001,
                  n_rate: float = 0.001, tail_frac: float = 0.0, tail_sub_rate: float = 0.2, tail_indel_rate: float = 0.02,
                  seed: int = CONFIG_SEED_BASE + 3, second_seed: bool = True):
    """ext_tasks with the reads drawn from the reference `pac`: returns (byte tasks, coordinate tasks) of the SAME seeds -- what the
    Scala driver ships today (ExtTaskSoA) and what a coordinate batch ships instead (ExtCoordTaskSoA, wire format 2)"""
    from . import ExtCoordTaskSoA
    lib = _load()
    lib.bpsw_synth_ext_tasks_ref.restype = C.c_int
    cfg = _ExtCfg(seed=seed, n_reads=n_reads, read_len=read_len, sub_rate=sub_rate, indel_rate=indel_rate,
                  n_rate=n_rate, tail_frac=tail_frac, tail_sub_rate=tail_sub_rate, tail_indel_rate=tail_indel_rate,
                  a=1, o_del=6, e_del=1, o_ins=6, e_ins=1, w=100, min_seed_len=19, second_seed=int(second_seed))
    cap = 2 * n_reads + 1
    i32 = lambda: np.zeros(cap, dtype=np.int32)  # noqa: E731
    i64 = lambda: np.zeros(cap, dtype=np.int64)  # noqa: E731
    f = dict(left_qlen=i32(), left_rlen=i32(), right_qlen=i32(), right_rlen=i32(), left_q_off=i64(), left_r_off=i64(),
             right_q_off=i64(), right_r_off=i64(), reg_score=i32(), q_beg=i32(), h0=i32(), idx=i32())
    srb, sln = i64(), i32()
    pool_cap = int(cap) * (4 * read_len + 16)
    pool = np.zeros(pool_cap, dtype=np.uint8)
    used = C.c_size_t(0)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    pac = np.ascontiguousarray(pac, np.uint8)
    n = lib.bpsw_synth_ext_tasks_ref(C.byref(cfg), vp(pac), C.c_int64(l_pac), vp(f["left_qlen"]), vp(f["left_rlen"]), vp(f["right_qlen"]),
                                     vp(f["right_rlen"]), vp(f["left_q_off"]), vp(f["left_r_off"]), vp(f["right_q_off"]),
                                     vp(f["right_r_off"]), vp(f["reg_score"]), vp(f["q_beg"]), vp(f["h0"]), vp(f["idx"]), vp(srb), vp(sln),
                                     vp(pool), C.c_size_t(pool_cap), C.byref(used))
    if n < 0:
        raise BpswError("synthetic pool too small (or the reference shorter than a read's window)")
    poolv = pool[: max(used.value, 1)].copy()
    by = ExtTaskSoA(pool=poolv, **{k: v[:n].copy() for k, v in f.items()})
    co = ExtCoordTaskSoA(pool=poolv, seed_len=sln[:n].copy(), seed_rbeg=srb[:n].copy(),
                         **{k: f[k][:n].copy() for k in ("left_qlen", "left_rlen", "right_qlen", "right_rlen", "left_q_off", "right_q_off",
                                                         "reg_score", "q_beg", "h0", "idx")})
    return by, co


def sw_jobs(n_jobs: int, read_len: int = 150, win_min: int = 450, win_max: int = 750, sub_rate: float = 0.02,
            indel_rate: float = 0.002, n_rate: float = 0.001, unrelated_frac: float = 0.25, decoy_frac: float = 0.5,
            rev_frac: float = 0.5, seed: int = CONFIG_SEED_BASE + 3):
    """SWAlign2 jobs (mate vs rescue window).  Returns dict of arrays for Context.swalign2_batch."""
    lib = _load()
    cfg = _SwCfg(seed=seed, n_jobs=n_jobs, read_len=read_len, win_min=win_min, win_max=win_max, sub_rate=sub_rate,
                 indel_rate=indel_rate, n_rate=n_rate, unrelated_frac=unrelated_frac, decoy_frac=decoy_frac,
                 rev_frac=rev_frac)
    q_len = np.zeros(n_jobs, np.int32)
    t_len = np.zeros(n_jobs, np.int32)
    q_off = np.zeros(n_jobs, np.int64)
    t_off = np.zeros(n_jobs, np.int64)
    q_rev = np.zeros(n_jobs, np.uint8)
    q_cap = n_jobs * (read_len + 16) + 16
    t_cap = n_jobs * (read_len + win_max + 16) + 16
    q_pool = np.zeros(q_cap, np.uint8)
    t_pool = np.zeros(t_cap, np.uint8)
    qu, tu = C.c_size_t(0), C.c_size_t(0)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    n = lib.bpsw_synth_sw_jobs(C.byref(cfg), vp(q_len), vp(t_len), vp(q_off), vp(t_off), vp(q_rev), vp(q_pool),
                               C.c_size_t(q_cap), vp(t_pool), C.c_size_t(t_cap), C.byref(qu), C.byref(tu))
    if n < 0:
        raise BpswError("synthetic pool too small")
    return dict(q_len=q_len, t_len=t_len, q_off=q_off, t_off=t_off, q_rev=q_rev, q_pool=q_pool[: max(qu.value, 16)].copy(),
                t_pool=t_pool[: max(tu.value, 16)].copy())


# ---------------------------------------------------------------------------------------------------
# Boundary-1 groups: what memSamPeGroupJNIPrepare (MemSamPe.scala:1895-2000) hands to MateSWJNI.mateSWJNI
# ---------------------------------------------------------------------------------------------------
def _revcomp(a: np.ndarray) -> np.ndarray:
    r = a[::-1]
    return np.where(r < 4, 3 - r, 4).astype(np.uint8)


def _mutate(rng, seq: np.ndarray, sub: float, indel: float) -> np.ndarray:
    out = []
    i = 0
    L = len(seq)
    while len(out) < L and i < L:
        u = rng.random()
        if u < indel / 2:
            out.append(int(rng.integers(0, 4)))
        elif u < indel:
            i += 1
        elif u < indel + sub:
            out.append(int((seq[i] + 1 + rng.integers(0, 3)) & 3))
            i += 1
        else:
            out.append(int(seq[i]))
            i += 1
    while len(out) < L:
        out.append(int(rng.integers(0, 4)))
    return np.array(out[:L], dtype=np.uint8)


def random_pac(l_pac: int, seed: int = CONFIG_SEED_BASE + 77):
    """A 2-bit packed i.i.d. reference in BWA's .pac layout: base k = pac[k>>2] >> ((~k&3)<<1) & 3"""
    rng = np.random.default_rng(seed)
    bases = rng.integers(0, 4, l_pac).astype(np.uint8)
    padded = np.zeros(((l_pac + 3) // 4) * 4, np.uint8)
    padded[:l_pac] = bases
    q = padded.reshape(-1, 4)
    pac = ((q[:, 0] << 6) | (q[:, 1] << 4) | (q[:, 2] << 2) | q[:, 3]).astype(np.uint8)
    return pac, bases


class PacBases:
    """bases[a:b] of the forward strand, unpacked on demand from a 2-bit .pac (for references too long to unpack whole)"""

    def __init__(self, pac: np.ndarray, l_pac: int):
        self.pac, self.l_pac = pac, l_pac

    def __len__(self):
        return self.l_pac

    def __getitem__(self, sl):
        a, b, step = sl.indices(self.l_pac)
        assert step == 1
        if b <= a:
            return np.zeros(0, np.uint8)
        k = np.arange(a, b, dtype=np.int64)
        return ((self.pac[k >> 2] >> ((~k & 3) << 1)) & 3).astype(np.uint8)


def window_bases(bases: np.ndarray, l_pac: int, rb: int, re: int) -> np.ndarray:
    """bnsGetSeq on unpacked forward bases (data generation only): empty when the window bridges the strands"""
    rb, re = max(rb, 0), min(re, 2 * l_pac)
    if re <= rb or not (rb >= l_pac or re <= l_pac):
        return np.zeros(0, np.uint8)
    if rb >= l_pac:
        return (3 - bases[2 * l_pac - re: 2 * l_pac - rb])[::-1].astype(np.uint8)
    return bases[rb:re].copy()


def rescue_group(n_pairs: int, read_len: int = 150, seed: int = CONFIG_SEED_BASE + 3, l_pac: int = 46_709_983,
                 p_resc: float = 0.10, all_orientations: bool = False, sub_rate: float = 0.02, indel_rate: float = 0.002,
                 p_multi_anchor: float = 0.10, p_wrong_mate: float = 0.05, max_matesw: int = 100, pen_unpaired: int = 17,
                 ref_bases: np.ndarray | None = None, positions=None, p_decoy_anchor: float = 0.0):
    """Synthetic pair-end group (FR library, insert ~ N(400, 50^2)) in the flat layout of include/bpsw.h.

    Each pair has an anchor on one end; with probability p_resc the mate has no consistent hit, so the
    reference's mem_matesw would run SWAlign2 on the rescue window (which contains the mutated mate).
    With ref_bases (the unpacked forward strand of a reference of length l_pac -- anything that answers ref_bases[a:b], e.g.
    PacBases) reads and windows are cut from that reference, so the same group can be submitted with bytes or, after ref_load,
    with coordinates only.  positions: forward start of the first len(positions) pairs (the rest are drawn at random).
    p_decoy_anchor: an end gets, in front of its true hit, a better-scoring hit at an unrelated locus -- the rescue from it finds
    nothing, so the true hit's rescue is still needed afterwards (the replay's second round, csrc/bpsw_rescue.cpp).
    """
    from . import RescueGroupSoA, ALNREG_DTYPE
    rng = np.random.default_rng(seed)
    L = read_len
    avg, std = 400.0, 50.0
    low, high = int(avg - 4 * std), int(avg + 4 * std)  # 200 .. 600
    if all_orientations:  # exercise the non-reversed orientations too (never the case for Illumina FR data)
        pes = [(low, high, 0, avg, std), (low, high, 0, avg, std), (low - 50, high - 50, 0, avg, std), (low, high + 40, 0, avg, std)]
    else:
        pes = [(0, 0, 1, 0.0, 0.0), (low, high, 0, avg, std), (0, 0, 1, 0.0, 0.0), (0, 0, 1, 0.0, 0.0)]

    seq_len, seq_off, seq_chunks = [], [], []
    reg_cnt, regs, ref_cnt = [], [], []
    ref_rb, ref_re, ref_len, ref_off, ref_chunks = [], [], [], [], []
    seq_at = 0
    ref_at = 0

    def add_seq(s):
        nonlocal seq_at
        seq_len.append(len(s)); seq_off.append(seq_at); seq_chunks.append(s)
        pad = (-len(s)) % 16
        if pad:
            seq_chunks.append(np.zeros(pad, np.uint8))
        seq_at += len(s) + pad

    def mk_reg(rb, score, qb=0, qe=None):
        qe = L if qe is None else qe
        return (rb, rb + (qe - qb), qb, qe, score, score, 0, 0, 0, 100, (qe - qb) // 2, -1, int(rng.integers(0, 2 ** 62)))

    def window_for(anchor_rb, r, mate_len):  # getAlnRegRefJNI, MemSamPe.scala:1810-1878
        lo, hi, failed = pes[r][0], pes[r][1], pes[r][2]
        if failed:
            return -1, -1
        is_rev = (r >> 1) != (r & 1)
        is_larger = (r >> 1) == 0
        if not is_rev:
            rb = anchor_rb + lo if is_larger else anchor_rb - hi
            re = (anchor_rb + hi if is_larger else anchor_rb - lo) + mate_len
        else:
            rb = (anchor_rb + lo if is_larger else anchor_rb - hi) - mate_len
            re = anchor_rb + hi if is_larger else anchor_rb - lo
        return max(rb, 0), min(re, 2 * l_pac)

    for k in range(n_pairs):
        P = int(rng.integers(2000, l_pac - 3000))
        if positions is not None and k < len(positions):
            P = int(positions[k])
        ins = int(np.clip(rng.normal(avg, std), low + 20, high - 20))
        clean = [rng.integers(0, 4, L).astype(np.uint8), rng.integers(0, 4, L).astype(np.uint8)]  # forward-strand loci
        if ref_bases is not None:
            clean = [ref_bases[P:P + L].copy(), ref_bases[P + ins - L:P + ins].copy()]
        reads = [_mutate(rng, clean[0], sub_rate, indel_rate), _revcomp(_mutate(rng, clean[1], sub_rate, indel_rate))]
        true_rb = [P, 2 * l_pac - (P + ins)]  # end 0 forward at P; end 1 on the reverse strand
        u = rng.random()
        have = [True, True]
        if u < p_resc:
            have[int(rng.integers(0, 2))] = False
        ends_regs = [[], []]
        for i in range(2):
            if have[i]:
                ends_regs[i].append(mk_reg(true_rb[i], L - int(rng.integers(0, 8))))
                if rng.random() < p_multi_anchor:  # a near-duplicate anchor a few bases away (overlapping hit)
                    ends_regs[i].append(mk_reg(true_rb[i] + int(rng.integers(1, 4)), ends_regs[i][0][4] - int(rng.integers(0, pen_unpaired))))
                if p_decoy_anchor > 0 and rng.random() < p_decoy_anchor:  # a decoy that sorts in front of the true hit (no draw when off: the streams of the golden fixtures stay as they were)
                    ends_regs[i].append(mk_reg(int(rng.integers(0, 2 * l_pac - L)), ends_regs[i][0][4] + 1 + int(rng.integers(0, 5))))
                if rng.random() < 0.15:  # a weak, far-away secondary hit below the anchor threshold
                    ends_regs[i].append(mk_reg(int(rng.integers(0, 2 * l_pac - L)), ends_regs[i][0][4] - pen_unpaired - 5, 10, L - 20))
            elif rng.random() < p_wrong_mate:
                ends_regs[i].append(mk_reg(int(rng.integers(0, 2 * l_pac - L)), L // 2, 0, L // 2 + 10))
            ends_regs[i].sort(key=lambda t: -t[4])
        for i in range(2):
            add_seq(reads[i])
        for i in range(2):
            rl = ends_regs[i]
            reg_cnt.append(len(rl)); regs.extend(rl)
            anchors = [t for t in rl if t[4] >= rl[0][4] - pen_unpaired][:max_matesw] if rl else []
            ref_cnt.append(len(anchors))
            mate = 1 - i
            for a in anchors:
                for r in range(4):
                    rb, re = window_for(a[0], r, L)
                    if rb < 0 and re < 0:
                        ref_rb.append(-1); ref_re.append(-1); ref_len.append(0); ref_off.append(0)
                        continue
                    n = max(re - rb, 0)
                    if ref_bases is not None:
                        w = window_bases(ref_bases, l_pac, rb, re)
                        n = len(w)  # 0 when the window bridges the strands (then len != re - rb: the window is skipped)
                    else:
                        w = rng.integers(0, 4, n).astype(np.uint8)
                    # where the mate truly lies in this window (only for the FR orientation of a true anchor)
                    if ref_bases is None and r == 1 and abs(a[0] - true_rb[i]) < 8 and n >= L:
                        off = (true_rb[i] + ins - L) - rb if i == 0 else (true_rb[i] + ins - L) - rb
                        # window is on the anchor's strand; the mate lies `ins - L` past the anchor start
                        off = (a[0] - rb) + (ins - L) + (true_rb[i] - a[0])
                        if 0 <= off <= n - L:
                            w[off:off + L] = clean[mate] if i == 0 else _revcomp(clean[mate])
                    ref_rb.append(rb); ref_re.append(re); ref_len.append(n); ref_off.append(ref_at)
                    ref_chunks.append(w)
                    pad = (-n) % 16
                    if pad:
                        ref_chunks.append(np.zeros(pad, np.uint8))
                    ref_at += n + pad

    regs_arr = np.array(regs, dtype=ALNREG_DTYPE) if regs else np.zeros(0, ALNREG_DTYPE)
    cat = lambda ch: np.concatenate(ch) if ch else np.zeros(16, np.uint8)
    return RescueGroupSoA(group_size=n_pairs, l_pac=l_pac, pes=pes, seq_len=np.array(seq_len, np.int32),
                          seq_off=np.array(seq_off, np.int64), seq_pool=cat(seq_chunks), reg_cnt=np.array(reg_cnt, np.int32),
                          regs=regs_arr, ref_cnt=np.array(ref_cnt, np.int32), ref_rb=np.array(ref_rb, np.int64),
                          ref_re=np.array(ref_re, np.int64), ref_len=np.array(ref_len, np.int64),
                          ref_off=np.array(ref_off, np.int64), ref_pool=cat(ref_chunks))


def read_chains(n_reads: int, ref_bases: np.ndarray, l_pac: int, read_len: int = 150, sub_rate: float = 0.01,
                indel_rate: float = 0.001, tail_frac: float = 0.0, min_seed: int = 19, p_subseed: float = 0.3,
                p_shifted: float = 0.15, p_decoy: float = 0.3, seed: int = CONFIG_SEED_BASE + 9, positions=None):
    """Reads with their seed chains, as memChainToAlnBatched (MemChainToAlignBatched.scala:380-616) receives them.

    A read is a mutated substring of either strand of the reference; its chain holds the exact-match runs >= min_seed
    (a surrogate for the SMEM seeds), sometimes a contained sub-seed (skipped by testExtension), sometimes a
    seed on a neighbouring diagonal (kept alive by checkOverlapping), and sometimes a second chain at an unrelated locus.
    """
    from . import ChainBatchSoA
    rng = np.random.default_rng(seed)
    L = read_len
    read_len_a, read_off, pools = [], [], []
    chain_cnt, seed_cnt, s_rb, s_qb, s_len = [], [], [], [], []
    at = 0
    for r in range(n_reads):
        rev = rng.random() < 0.5
        lo, hi = (l_pac + 600, 2 * l_pac - L - 600) if rev else (600, l_pac - L - 600)
        rb0 = int(rng.integers(lo, hi))
        if positions is not None and r < len(positions):   # a start in the doubled coordinate space, given by the caller
            rb0 = int(positions[r])
        seg = window_bases(ref_bases, l_pac, rb0, rb0 + L + 60)
        es, ei = sub_rate, indel_rate
        if tail_frac > 0 and rng.random() < tail_frac:
            es, ei = 0.2, 0.05
        read, pos = [], []          # pos[i] = index in seg the read base was copied from, -1 otherwise
        i = 0
        while len(read) < L and i < len(seg):
            u = rng.random()
            if u < ei / 2:          # insertion
                read.append(int(rng.integers(0, 4))); pos.append(-1)
            elif u < ei:            # deletion
                i += 1
            elif u < ei + es:       # substitution
                read.append(int((seg[i] + 1 + rng.integers(0, 3)) & 3)); pos.append(-1); i += 1
            else:
                read.append(int(seg[i])); pos.append(i); i += 1
        while len(read) < L:
            read.append(int(rng.integers(0, 4))); pos.append(-1)
        read = np.array(read, np.uint8)
        if rng.random() < 0.02:
            read[int(rng.integers(0, L))] = 4   # an N
            pos[int(np.argmax(read == 4))] = -1
        seeds = []
        q = 0
        while q < L:
            if pos[q] < 0:
                q += 1
                continue
            e = q
            while e + 1 < L and pos[e + 1] == pos[e] + 1:
                e += 1
            if e + 1 - q >= min_seed:
                seeds.append((rb0 + pos[q], q, e + 1 - q))
            q = e + 1
        chains = []
        if seeds:
            ch = list(seeds)
            if rng.random() < p_subseed:
                rb, qb, ln = ch[int(rng.integers(0, len(ch)))]
                if ln >= min_seed + 8:
                    d = int(rng.integers(1, ln - min_seed))
                    ch.append((rb + d, qb + d, int(rng.integers(min_seed, ln - d + 1))))
            if rng.random() < p_shifted:
                rb, qb, ln = ch[int(rng.integers(0, len(ch)))]
                sh = int(rng.integers(1, 6)) * (1 if rng.random() < 0.5 else -1)
                cut = int(rng.integers(0, max(1, ln // 3)))
                if ln - cut >= min_seed:
                    ch.append((rb + sh + cut, qb + cut, ln - cut))
            ch.sort(key=lambda t: (t[1], t[0]))
            chains.append(ch)
        if rng.random() < p_decoy:
            lo2, hi2 = (l_pac + 600, 2 * l_pac - L - 600) if rng.random() < 0.5 else (600, l_pac - L - 600)
            rbd = int(rng.integers(lo2, hi2))
            qb = int(rng.integers(0, L - 30))
            ch = [(rbd + qb, qb, int(rng.integers(min_seed, 30)))]
            if rng.random() < 0.3 and qb + 60 < L:
                ch.append((rbd + qb + 40 + int(rng.integers(-2, 3)), qb + 40, int(rng.integers(min_seed, 20 + 1))))
            chains.append(ch)
        read_len_a.append(L); read_off.append(at); pools.append(read)
        pad = (-L) % 16
        if pad:
            pools.append(np.zeros(pad, np.uint8))
        at += L + pad
        chain_cnt.append(len(chains))
        for ch in chains:
            seed_cnt.append(len(ch))
            for rb, qb, ln in ch:
                s_rb.append(rb); s_qb.append(qb); s_len.append(ln)
    return ChainBatchSoA(l_pac=l_pac, read_len=np.array(read_len_a, np.int32), read_off=np.array(read_off, np.int64),
                         read_pool=np.concatenate(pools) if pools else np.zeros(16, np.uint8),
                         chain_cnt=np.array(chain_cnt, np.int32), seed_cnt=np.array(seed_cnt, np.int32),
                         seed_rbeg=np.array(s_rb, np.int64), seed_qbeg=np.array(s_qb, np.int32), seed_len=np.array(s_len, np.int32))


def coord_ext_tasks(chains, ref_bases: np.ndarray, a: int = 1, w: int = 100, seed: int = CONFIG_SEED_BASE + 10, pad_max: int = 40):
    """Extension tasks for the seeds of `read_chains` output, twice: as a coordinate batch (ExtCoordTaskSoA: query flanks +
    the seed's coordinates) and as the byte tasks the Scala driver builds for the same seeds (ExtTaskSoA: leftRs / rightRs
    copied out of bnsGetSeq's window, MemChainToAlignBatched.scala:500-545).  The window around a seed is the query flank plus
    calMaxGap-like slack of random size, clipped to the seed's strand as getMaxSpan does (:654-677)."""
    from . import ExtCoordTaskSoA, ExtTaskSoA
    rng = np.random.default_rng(seed)
    l_pac = int(chains.l_pac)
    lq_a, lr_a, rq_a, rr_a, lqo, rqo, lro, rro, sc_a, qb_a, h0_a, idx_a, sl_a, srb_a = ([] for _ in range(14))
    pool, at = [], 0

    def add(seq):
        nonlocal at
        off = at
        pool.append(np.asarray(seq, np.uint8)); at += len(seq)
        return off
    s_at = 0
    ch_at = 0
    for r in range(len(chains.read_len)):
        L = int(chains.read_len[r])
        read = chains.read_pool[int(chains.read_off[r]): int(chains.read_off[r]) + L]
        for _ in range(int(chains.chain_cnt[r])):
            ns = int(chains.seed_cnt[ch_at]); ch_at += 1
            for k in range(ns):
                rb, qb, ln = int(chains.seed_rbeg[s_at + k]), int(chains.seed_qbeg[s_at + k]), int(chains.seed_len[s_at + k])
                lo_s, hi_s = (0, l_pac) if rb < l_pac else (l_pac, 2 * l_pac)
                lq, rq = qb, L - (qb + ln)
                lr = min(lq + int(rng.integers(0, pad_max + 1)), rb - lo_s) if lq else 0
                rr = min(rq + int(rng.integers(0, pad_max + 1)), hi_s - (rb + ln)) if rq else 0
                if rng.random() < 0.05:
                    lr = min(lr, max(0, lq - 3))   # a target flank shorter than the query flank
                lwin = window_bases(ref_bases, l_pac, rb - lr, rb)[::-1] if lr else np.zeros(0, np.uint8)
                rwin = window_bases(ref_bases, l_pac, rb + ln, rb + ln + rr) if rr else np.zeros(0, np.uint8)
                lq_a.append(lq); lr_a.append(lr); rq_a.append(rq); rr_a.append(rr)
                lqo.append(add(read[:qb][::-1])); rqo.append(add(read[qb + ln:]))
                lro.append(add(lwin)); rro.append(add(rwin))
                sc_a.append(ln * a); qb_a.append(qb); h0_a.append(ln * a); idx_a.append(len(idx_a) & 0x7fff)
                sl_a.append(ln); srb_a.append(rb)
            s_at += ns
    i32 = lambda v: np.array(v, np.int32)  # noqa: E731
    i64 = lambda v: np.array(v, np.int64)  # noqa: E731
    poolv = np.concatenate(pool + [np.zeros(16, np.uint8)])
    co = ExtCoordTaskSoA(left_qlen=i32(lq_a), left_rlen=i32(lr_a), right_qlen=i32(rq_a), right_rlen=i32(rr_a), left_q_off=i64(lqo),
                         right_q_off=i64(rqo), reg_score=i32(sc_a), q_beg=i32(qb_a), h0=i32(h0_a), idx=i32(idx_a), seed_len=i32(sl_a),
                         seed_rbeg=i64(srb_a), pool=poolv, w=w, mat_max=a)
    by = ExtTaskSoA(left_qlen=i32(lq_a), left_rlen=i32(lr_a), right_qlen=i32(rq_a), right_rlen=i32(rr_a), left_q_off=i64(lqo),
                    left_r_off=i64(lro), right_q_off=i64(rqo), right_r_off=i64(rro), reg_score=i32(sc_a), q_beg=i32(qb_a), h0=i32(h0_a),
                    idx=i32(idx_a), pool=poolv, w=w, mat_max=a)
    return co, by


# ---------------------------------------------------------------------------------------------------------------------
# worker2's tail: pairs with their seed chains over a multi-contig reference
def contig_reference(contig_lens, seed: int = CONFIG_SEED_BASE + 78, dup_len: int = 1500):
    """A multi-contig 2-bit reference: (pac, bases, ann_off, ann_len, ann_names, dups).  Each contig but the first also
    carries a near-copy (0.5 % substitutions) of a `dup_len` stretch of contig 0, so reads drawn from those stretches have
    more than one hit (sub / sub_n / secondary / the pairing choice of memPair all come alive)."""
    l_pac = int(sum(contig_lens))
    pac, bases = random_pac(l_pac, seed)
    rng = np.random.default_rng(seed + 1)
    off = np.concatenate([[0], np.cumsum(contig_lens)[:-1]]).astype(np.int64)
    dups = []  # (src, dst, len)
    for c in range(1, len(contig_lens)):
        if contig_lens[0] < 4 * dup_len or contig_lens[c] < 4 * dup_len:
            continue
        src = int(rng.integers(dup_len, contig_lens[0] - 2 * dup_len))
        dst = int(off[c] + rng.integers(dup_len, contig_lens[c] - 2 * dup_len))
        seg = bases[src:src + dup_len].copy()
        mut = rng.random(dup_len) < 0.005
        seg[mut] = (seg[mut] + 1 + rng.integers(0, 3, int(mut.sum()))) & 3
        bases[dst:dst + dup_len] = seg
        dups.append((src, dst, dup_len))
    padded = np.zeros(((l_pac + 3) // 4) * 4, np.uint8)
    padded[:l_pac] = bases
    q = padded.reshape(-1, 4)
    pac = ((q[:, 0] << 6) | (q[:, 1] << 4) | (q[:, 2] << 2) | q[:, 3]).astype(np.uint8)
    names = [f"ctg{c + 1}" for c in range(len(contig_lens))]
    return pac, bases, off, np.array(contig_lens, np.int32), names, dups


def _seeded_read(rng, ref_bases, l_pac, rb0, L, es, ei, min_seed):
    """A read copied from doubled-coordinate position rb0 with errors; returns (read, exact-match seeds, pos map)"""
    seg = window_bases(ref_bases, l_pac, rb0, rb0 + L + 60)
    read, pos = [], []
    i = 0
    while len(read) < L and i < len(seg):
        u = rng.random()
        if u < ei / 2:
            read.append(int(rng.integers(0, 4))); pos.append(-1)
        elif u < ei:
            i += 1
        elif u < ei + es:
            read.append(int((seg[i] + 1 + rng.integers(0, 3)) & 3)); pos.append(-1); i += 1
        else:
            read.append(int(seg[i])); pos.append(i); i += 1
    while len(read) < L:
        read.append(int(rng.integers(0, 4))); pos.append(-1)
    read = np.array(read, np.uint8)
    seeds = []
    q = 0
    while q < L:
        if pos[q] < 0:
            q += 1
            continue
        e = q
        while e + 1 < L and pos[e + 1] == pos[e] + 1:
            e += 1
        if e + 1 - q >= min_seed:
            seeds.append((rb0 + pos[q], q, e + 1 - q))
        q = e + 1
    return read, seeds


def tail_pairs(n_pairs: int, ref_bases: np.ndarray, ann_off, ann_len, dups=(), read_len: int = 150, sub_rate: float = 0.01,
               indel_rate: float = 0.002, p_unmappable: float = 0.04, p_far: float = 0.05, p_span: float = 0.03,
               p_dup: float = 0.15, p_decoy: float = 0.15, p_clip: float = 0.1, p_hard: float = 0.0, min_seed: int = 19,
               seed: int = CONFIG_SEED_BASE + 11):
    """FR pairs (insert ~ N(400, 50^2)) with the seed chains of both ends, plus names and qualities: what worker2's tail
    needs once chains have been turned into regions (memChainToAln + memSortAndDedup) and the rescue has run.
    Returns (ChainBatchSoA over the 2n reads in (pair, end) order, names list, qual_pool aligned with read_pool, pes)."""
    from . import ChainBatchSoA
    rng = np.random.default_rng(seed)
    L = read_len
    l_pac = int(ann_off[-1] + ann_len[-1])
    avg, std = 400.0, 50.0
    low, high = int(avg - 4 * std), int(avg + 4 * std)
    pes = [(0, 0, 1, 0.0, 0.0), (low, high, 0, avg, std), (0, 0, 1, 0.0, 0.0), (0, 0, 1, 0.0, 0.0)]
    read_len_a, read_off, pools, quals = [], [], [], []
    chain_cnt, seed_cnt, s_rb, s_qb, s_len = [], [], [], [], []
    names = []
    at = 0

    def add_read(read, chains):
        nonlocal at
        read_len_a.append(L); read_off.append(at); pools.append(read)
        quals.append((33 + rng.integers(2, 41, L)).astype(np.uint8))
        pad = (-L) % 16
        if pad:
            pools.append(np.zeros(pad, np.uint8)); quals.append(np.full(pad, 33, np.uint8))
        at += L + pad
        chain_cnt.append(len(chains))
        for ch in chains:
            seed_cnt.append(len(ch))
            for rb, qb, ln in ch:
                s_rb.append(rb); s_qb.append(qb); s_len.append(ln)

    for k in range(n_pairs):
        names.append(f"pair{seed & 0xffff}_{k}")
        ins = int(np.clip(rng.normal(avg, std), low + 20, high - 20))
        u = rng.random()
        if u < p_span and len(ann_off) > 1:      # the fragment straddles a contig boundary (bwaFixXref2)
            c = int(rng.integers(1, len(ann_off)))
            P = int(ann_off[c]) - int(rng.integers(10, ins - 10))
        elif u < p_span + p_dup and dups:        # inside a duplicated stretch: two candidate loci
            src, dst, dl = dups[int(rng.integers(0, len(dups)))]
            P = (src if rng.random() < 0.5 else dst) + int(rng.integers(0, max(1, dl - ins)))
        else:
            c = int(rng.integers(0, len(ann_off)))
            P = int(ann_off[c]) + int(rng.integers(0, max(1, int(ann_len[c]) - ins)))
        P = max(0, min(P, l_pac - ins - 70))
        P2 = P
        if rng.random() < p_far:                 # improper pair: the mate comes from somewhere else
            P2 = int(rng.integers(0, l_pac - ins - 70))
        fwd_rb = P                               # forward-strand end starts at P
        rev_rb = 2 * l_pac - (P2 + ins)          # reverse-strand end, doubled coordinates
        ends = [fwd_rb, rev_rb] if rng.random() < 0.5 else [rev_rb, fwd_rb]
        for i in range(2):
            if rng.random() < p_unmappable:
                add_read(rng.integers(0, 4, L).astype(np.uint8), [])
                continue
            es, ei = (0.09, 0.01) if rng.random() < p_hard else (sub_rate, indel_rate)   # p_hard: too many errors for a seed
            read, seeds = _seeded_read(rng, ref_bases, l_pac, ends[i], L, es, ei, min_seed)
            if rng.random() < p_clip:            # an unrelated tail: soft clipping on one side
                cut = int(rng.integers(15, 50))
                if rng.random() < 0.5:
                    read[:cut] = rng.integers(0, 4, cut)
                    seeds = [(rb + max(0, cut - qb), max(qb, cut), ln - max(0, cut - qb)) for rb, qb, ln in seeds if qb + ln - cut >= min_seed]
                else:
                    read[L - cut:] = rng.integers(0, 4, cut)
                    seeds = [(rb, qb, min(ln, L - cut - qb)) for rb, qb, ln in seeds if min(ln, L - cut - qb) >= min_seed]
            if rng.random() < 0.02:              # an N: no seed may cover it
                pn = int(rng.integers(0, L))
                read[pn] = 4
                seeds = [(rb, qb, ln) for rb, qb, ln in seeds if not (qb <= pn < qb + ln)]
            chains = [sorted(seeds, key=lambda t: (t[1], t[0]))] if seeds else []
            # the other copy of a duplicated stretch: seeds that are exact there too
            for src, dst, dl in dups:
                for a, b in ((src, dst), (dst, src)):
                    fpos = ends[i] if ends[i] < l_pac else 2 * l_pac - ends[i] - L
                    if a <= fpos and fpos + L <= a + dl and seeds:
                        shift = b - a
                        alt = []
                        for rb, qb, ln in seeds:
                            rb2 = rb + shift if rb < l_pac else rb - shift
                            w1 = window_bases(ref_bases, l_pac, rb2, rb2 + ln)
                            if len(w1) == ln and np.array_equal(w1, read[qb:qb + ln]):
                                alt.append((rb2, qb, ln))
                        if alt:
                            chains.append(alt)
            if rng.random() < p_decoy:
                lo2, hi2 = (l_pac + 600, 2 * l_pac - L - 600) if rng.random() < 0.5 else (600, l_pac - L - 600)
                rbd = int(rng.integers(lo2, hi2))
                qb = int(rng.integers(0, L - 30))
                chains.append([(rbd + qb, qb, int(rng.integers(min_seed, 30)))])
            add_read(read, chains)
    b = ChainBatchSoA(l_pac=l_pac, read_len=np.array(read_len_a, np.int32), read_off=np.array(read_off, np.int64),
                      read_pool=np.concatenate(pools), chain_cnt=np.array(chain_cnt, np.int32),
                      seed_cnt=np.array(seed_cnt, np.int32), seed_rbeg=np.array(s_rb, np.int64),
                      seed_qbeg=np.array(s_qb, np.int32), seed_len=np.array(s_len, np.int32))
    return b, names, np.concatenate(quals), pes


class _GroupCfg(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_pairs", C.c_int32), ("read_len", C.c_int32), ("l_pac", C.c_int64),
                ("p_resc", C.c_double), ("sub_rate", C.c_double), ("indel_rate", C.c_double), ("p_multi_anchor", C.c_double),
                ("p_wrong_mate", C.c_double), ("max_matesw", C.c_int32), ("pen_unpaired", C.c_int32), ("low", C.c_int32),
                ("high", C.c_int32), ("avg", C.c_double), ("std", C.c_double)]


def rescue_group_fast(n_pairs: int, read_len: int = 150, seed: int = CONFIG_SEED_BASE + 3, l_pac: int = 46_709_983,
                      p_resc: float = 0.10, sub_rate: float = 0.02, indel_rate: float = 0.002, p_multi_anchor: float = 0.10,
                      p_wrong_mate: float = 0.05, max_matesw: int = 100, pen_unpaired: int = 17):
    """rescue_group()'s model (FR library, windows shipped as bytes, no backing reference) generated by libbpsw_synth.so:
    a million pairs in seconds, for bench.py.  Same layout, different random stream."""
    from . import RescueGroupSoA, ALNREG_DTYPE
    lib = _load()
    lib.bpsw_synth_rescue_group.restype = C.c_int
    avg, std = 400.0, 50.0
    low, high = int(avg - 4 * std), int(avg + 4 * std)
    cfg = _GroupCfg(seed=seed, n_pairs=n_pairs, read_len=read_len, l_pac=l_pac, p_resc=p_resc, sub_rate=sub_rate,
                    indel_rate=indel_rate, p_multi_anchor=p_multi_anchor, p_wrong_mate=p_wrong_mate, max_matesw=max_matesw,
                    pen_unpaired=pen_unpaired, low=low, high=high, avg=avg, std=std)
    n = n_pairs
    Lp = (read_len + 15) & ~15
    wp = (high - low + 2 * read_len + 15) & ~15
    seq_len, seq_off = np.zeros(2 * n, np.int32), np.zeros(2 * n, np.int64)
    seq_pool = np.empty(max(2 * n * Lp, 16), np.uint8)
    reg_cnt, ref_cnt = np.zeros(2 * n, np.int32), np.zeros(2 * n, np.int32)
    regs = np.empty(6 * n + 1, ALNREG_DTYPE)
    rows_cap = 4 * n + 1
    ref_rb, ref_re = np.empty(4 * rows_cap, np.int64), np.empty(4 * rows_cap, np.int64)
    ref_len, ref_off = np.empty(4 * rows_cap, np.int64), np.empty(4 * rows_cap, np.int64)
    ref_pool = np.empty(max(rows_cap * wp, 16), np.uint8)
    counts = np.zeros(4, np.int64)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = lib.bpsw_synth_rescue_group(C.byref(cfg), vp(seq_len), vp(seq_off), vp(seq_pool), C.c_size_t(seq_pool.size), vp(reg_cnt),
                                     vp(regs), C.c_size_t(regs.shape[0]), vp(ref_cnt), vp(ref_rb), vp(ref_re), vp(ref_len), vp(ref_off),
                                     C.c_size_t(rows_cap), vp(ref_pool), C.c_size_t(ref_pool.size), vp(counts))
    if rc != 0:
        raise BpswError("synthetic rescue group: pool too small")
    nreg, nrow, sb, rb_ = (int(v) for v in counts)
    pes = [(0, 0, 1, 0.0, 0.0), (low, high, 0, avg, std), (0, 0, 1, 0.0, 0.0), (0, 0, 1, 0.0, 0.0)]
    return RescueGroupSoA(group_size=n, l_pac=l_pac, pes=pes, seq_len=seq_len, seq_off=seq_off, seq_pool=seq_pool[:max(sb, 16)].copy(),
                          reg_cnt=reg_cnt, regs=regs[:nreg].copy(), ref_cnt=ref_cnt, ref_rb=ref_rb[:4 * nrow].copy(),
                          ref_re=ref_re[:4 * nrow].copy(), ref_len=ref_len[:4 * nrow].copy(), ref_off=ref_off[:4 * nrow].copy(),
                          ref_pool=ref_pool[:max(rb_, 16)].copy())
