#!/usr/bin/env python3
"""Generates the committed golden vectors from the REFERENCE'S OWN C (oracle/_ref/libbwaref.so = the files
under /root/reference/src/main/native compiled in place by oracle/Makefile).  Run in the build container:

    make -C oracle ref && python tests/golden/make_golden.py

Each .npz holds inputs and the reference's outputs; nothing here is produced by our oracle or kernels.
Known caveat recorded per file: ksw_align2's score2/te2 come from the SSE2 striped kernel and are NOT the
true-DP values the Scala text computes; tests compare them only where `b8_safe` is set.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "cloud-scale-bwamem_amd"))
import pyoracle as po  # noqa: E402
from bpsw_hip import synth  # noqa: E402

ref = po.Ref()
mat = po.default_mat()
rng = np.random.default_rng(20261003)
# usage: make_golden.py [--check] [stem ...]
#   stems: the files to (re)write, e.g. `make_golden.py bns_get_seq`; default all
#   --check: write nothing; regenerate every file in memory and fail (exit 1) when an array differs from the committed one --
#            "a committed script generated them" stays true only as long as this passes (tests/test_golden.py runs it)
CHECK = "--check" in sys.argv[1:]
ONLY = set(a for a in sys.argv[1:] if not a.startswith("--"))
_savez = np.savez_compressed
_mismatch = []


def _filtered_savez(path, **kw):
    stem = os.path.splitext(os.path.basename(path))[0]
    if ONLY and stem not in ONLY:
        return
    if CHECK:
        if not os.path.exists(path):
            _mismatch.append(f"{stem}: not committed")
            return
        old = np.load(path)
        for k, v in kw.items():
            v = np.asarray(v)
            if k not in old.files or old[k].shape != v.shape or old[k].dtype != v.dtype or not np.array_equal(old[k], v):
                _mismatch.append(f"{stem}.{k}")
        for k in old.files:
            if k not in kw:
                _mismatch.append(f"{stem}.{k}: no longer generated")
        print("checked", os.path.basename(path))
        return
    _savez(path, **kw)
    print("wrote", os.path.basename(path))


np.savez_compressed = _filtered_savez


def mutate(seq, sub, indel):
    out, i = [], 0
    while i < len(seq):
        u = rng.random()
        if u < indel / 2:
            out.append(int(rng.integers(0, 4)))
        elif u < indel:
            i += 1
        elif u < indel + sub:
            out.append(int((seq[i] + 1 + rng.integers(0, 3)) & 3)); i += 1
        else:
            out.append(int(seq[i])); i += 1
    return np.array(out, np.uint8)


def pack(seqs):
    off = np.zeros(len(seqs) + 1, np.int64)
    for i, s in enumerate(seqs):
        off[i + 1] = off[i] + len(s)
    return off, (np.concatenate(seqs) if seqs else np.zeros(0, np.uint8)).astype(np.uint8)


# ---- ksw_extend2 (== SWExtend under the BWA z-drop parse) ---------------------------------------
qs, ts, par, outs = [], [], [], []
for n in range(600):
    ql = int(rng.integers(1, 232))
    err = [0.01, 0.05, 0.10, 0.20][n % 4]
    tl = ql + int(rng.integers(0, max(ql, 2)))
    t = rng.integers(0, 4, tl).astype(np.uint8)
    q = mutate(t, err, err / 5)[:ql]
    if len(q) < ql:
        q = np.concatenate([q, rng.integers(0, 4, ql - len(q)).astype(np.uint8)])
    if n % 7 == 0:  # a good prefix followed by unrelated sequence: the z-drop / early-stop regime
        cut = ql // 2
        q[cut:] = rng.integers(0, 4, ql - cut)
    if n % 50 == 0:
        q[rng.integers(0, ql)] = 4
    w = [100, 200, 5, 30][n % 4]
    h0 = int(rng.integers(19, 120))
    zdrop = [100, 100, 20, 0][(n // 4) % 4]
    bonus = 5
    qs.append(q); ts.append(t); par.append((w, bonus, zdrop, h0))
    outs.append(ref.ksw_extend2(q, t, mat, 6, 1, 6, 1, w, bonus, zdrop, h0))
qo, qp = pack(qs); to, tp = pack(ts)
np.savez_compressed(os.path.join(HERE, "ksw_extend2.npz"), q_off=qo, q_pool=qp, t_off=to, t_pool=tp,
                    params=np.array(par, np.int32), out=np.array(outs, np.int32),
                    note="ksw_extend2(native/ksw.c:379-476), scoring 1/-4/6/1/6/1, mat N=-1; out=[score,qle,tle,gtle,gscore,max_off]")

# ---- ksw_align2 (== SWAlign2 for score,te,qe,tb,qb) -----------------------------------------------
qs, ts, outs, safe = [], [], [], []
XTRA = po.KSW_XSUBO | po.KSW_XSTART | po.KSW_XBYTE | 19
for n in range(160):
    L = [100, 150, 150, 250][n % 4]
    wl = L + int(rng.integers(300, 700))
    t = rng.integers(0, 4, wl).astype(np.uint8)
    pos = int(rng.integers(0, wl - L - 20))
    q = mutate(t[pos:pos + L + 20], [0.02, 0.10][n % 2], 0.004)[:L]
    if len(q) < L:
        q = np.concatenate([q, rng.integers(0, 4, L - len(q)).astype(np.uint8)])
    if n % 5 == 0:
        q = rng.integers(0, 4, L).astype(np.uint8)  # unrelated
    if n % 3 == 0:  # decoy partial copy
        dl = int(rng.integers(25, 80)); dp = int(rng.integers(0, wl - dl))
        t[dp:dp + dl] = q[:dl]
    xtra = XTRA if L * 1 < 250 else (XTRA & ~po.KSW_XBYTE)
    qs.append(q); ts.append(t)
    outs.append(ref.ksw_align2(q, t, mat, 6, 1, 6, 1, xtra))
qo, qp = pack(qs); to, tp = pack(ts)
np.savez_compressed(os.path.join(HERE, "ksw_align2.npz"), q_off=qo, q_pool=qp, t_off=to, t_pool=tp,
                    out=np.array(outs, np.int32),
                    note="ksw_align2(native/ksw.c:342-364) xtra=XSUBO|XSTART|XBYTE(if L<250)|19; out=[score,te,qe,score2,te2,tb,qb]; "
                         "score2/te2 are SSE2-padded values (SURVEY B8), compare only score,te,qe,tb,qb")

# ---- ksw_global2 (== SWGlobal) -----------------------------------------------------------------------
qs, ts, ws, scores, cigs = [], [], [], [], []
for n in range(200):
    ql = int(rng.integers(20, 251))
    q = rng.integers(0, 4, ql).astype(np.uint8)
    t = mutate(q, [0.0, 0.05, 0.12, 0.19][n % 4], [0.0, 0.01, 0.03, 0.05][n % 4])
    if len(t) == 0:
        t = q[:1].copy()
    w = abs(len(t) - ql) + 3 + int(rng.integers(0, 40))
    sc, cg = ref.ksw_global2(q, t, mat, 6, 1, 6, 1, w)
    qs.append(q); ts.append(t); ws.append(w); scores.append(sc); cigs.append(cg)
qo, qp = pack(qs); to, tp = pack(ts)
co = np.zeros(len(cigs) + 1, np.int64)
for i, c in enumerate(cigs):
    co[i + 1] = co[i] + len(c)
np.savez_compressed(os.path.join(HERE, "ksw_global2.npz"), q_off=qo, q_pool=qp, t_off=to, t_pool=tp, w=np.array(ws, np.int32),
                    score=np.array(scores, np.int32), cig_off=co, cig_pool=np.concatenate(cigs).astype(np.uint32),
                    note="ksw_global2(native/ksw.c:501-584); cigar = len<<4|op, op 0=M 1=I 2=D")

# ---- mem_sort_and_dedup ---------------------------------------------------------------------------------
ins, outs_, in_off, out_off = [], [], [0], [0]
for n in range(150):
    k = int(rng.integers(0, 40)) if n % 10 else int(rng.integers(17, 80))
    regs = np.zeros(k, po.ALNREG_DTYPE)
    base = int(rng.integers(1000, 5000))
    for i in range(k):
        rb = base + int(rng.integers(0, 400)) if rng.random() < 0.8 else int(rng.integers(0, 10 ** 6))
        ln = int(rng.integers(30, 151))
        qb = int(rng.integers(0, 20))
        regs[i] = (rb, rb + ln + int(rng.integers(-3, 4)), qb, qb + ln, int(rng.integers(19, 151)), 0, 0, 0, 0, 100, ln // 2, -1, int(rng.integers(0, 2 ** 62)))
    if k > 3 and n % 3 == 0:  # exact duplicates and equal-re ties
        regs[1] = regs[0]
        regs[2]["re"] = regs[0]["re"]
    out = ref.sort_dedup(regs, 0.95)
    ins.append(regs); outs_.append(out); in_off.append(in_off[-1] + k); out_off.append(out_off[-1] + len(out))
np.savez_compressed(os.path.join(HERE, "mem_sort_and_dedup.npz"), in_off=np.array(in_off, np.int64), out_off=np.array(out_off, np.int64),
                    regs_in=np.concatenate(ins), regs_out=np.concatenate(outs_),
                    note="mem_sort_and_dedup(native/bwamem.c:394-435), mask_level_redun=0.95")

# ---- mem_group_matesw (the whole boundary-1 computation of jniNative.so) ------------------------------
import ctypes as C  # noqa: E402
orc_opt = po.Opt()
ints = np.zeros(16, np.int32); mlr = C.c_float(0); m25 = np.zeros(25, np.int8)
ref.lib.ref_opt_default(ints.ctypes.data_as(C.c_void_p), C.byref(mlr), m25.ctypes.data_as(C.c_void_p))
for i, name in enumerate(["a", "b", "o_del", "e_del", "o_ins", "e_ins", "pen_unpaired", "pen_clip5", "pen_clip3", "w", "zdrop", "T",
                          "flag", "min_seed_len", "max_ins", "max_matesw"]):
    setattr(orc_opt, name, int(ints[i]))
orc_opt.mask_level_redun = mlr.value
for k in range(25):
    orc_opt.mat[k] = int(m25[k])
# ("250": 2x250 bp mates at 8 % / 2 % error, a quarter of the pairs to rescue -- BASELINE.json configs[4]'s half of boundary 1: l_ms * a >= 250, so the
# reference leaves KSW_XBYTE off, native/bwamem_pair.c:179 / MemSamPe.scala:1187-1190, and its windows are 250 bases wider)
for tag, allo, n, p, kw in (("fr", False, 48, 0.4, {}), ("all4", True, 24, 0.5, {}),
                            ("250", False, 40, 0.4, dict(read_len=250, sub_rate=0.08, indel_rate=0.02, p_decoy_anchor=0.2)),
                            ("250_all4", True, 24, 0.5, dict(read_len=250, sub_rate=0.08, indel_rate=0.02))):
    g = synth.rescue_group(n, seed=4242 + n + 7 * len(kw), p_resc=p, all_orientations=allo, **kw)
    cnt, regs = ref.matesw_group(orc_opt, g)
    np.savez_compressed(os.path.join(HERE, f"mem_group_matesw_{tag}.npz"), group_size=g.group_size, l_pac=g.l_pac,
                        pes=np.array(g.pes, np.float64), seq_len=g.seq_len, seq_off=g.seq_off, seq_pool=g.seq_pool, reg_cnt=g.reg_cnt,
                        regs=g.regs, ref_cnt=g.ref_cnt, ref_rb=g.ref_rb, ref_re=g.ref_re, ref_len=g.ref_len, ref_off=g.ref_off,
                        ref_pool=g.ref_pool, out_cnt=cnt, out_regs=regs, opt_ints=ints, opt_mat=m25, opt_mask=np.float32(mlr.value),
                        note="mem_group_matesw(native/bwamem_pair.c:115-156) with mem_opt_init() defaults; csub inherits B8")
print("golden vectors written:", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))


# ---- bns_get_seq (== bnsGetSeq, util/BNTSeqUtil.scala:37-79): windows of a 2-bit reference -------------------------
rng2 = np.random.default_rng(20261004)
l_pac = 50_021
pac = rng2.integers(0, 256, (l_pac + 3) // 4, dtype=np.uint8)
begs, ends, seqs = [], [], []
for t in range(400):
    kind = t % 8
    L = int(rng2.integers(0, 900))
    if kind == 0:    # forward strand
        b = int(rng2.integers(0, l_pac - L)); e = b + L
    elif kind == 1:  # reverse strand
        b = int(rng2.integers(l_pac, 2 * l_pac - L)); e = b + L
    elif kind == 2:  # bridging the strands -> nothing
        b = l_pac - 1 - int(rng2.integers(0, 300)); e = l_pac + 1 + int(rng2.integers(0, 300))
    elif kind == 3:  # swapped ends
        e = int(rng2.integers(0, l_pac - L)); b = e + L
    elif kind == 4:  # clamped at 0
        b = -int(rng2.integers(1, 200)); e = int(rng2.integers(0, 600))
    elif kind == 5:  # clamped at 2*l_pac
        b = 2 * l_pac - int(rng2.integers(0, 600)); e = 2 * l_pac + int(rng2.integers(1, 200))
    elif kind == 6:  # touching the strand boundary from either side
        b = l_pac - L if t % 16 == 6 else l_pac; e = b + L
    else:            # anywhere
        b = int(rng2.integers(0, 2 * l_pac - L)); e = b + L
    begs.append(b); ends.append(e); seqs.append(ref.bns_get_seq(l_pac, pac, b, e))
so, sp = pack(seqs)
np.savez_compressed(os.path.join(HERE, "bns_get_seq.npz"), l_pac=l_pac, pac=pac, beg=np.array(begs, np.int64),
                    end=np.array(ends, np.int64), seq_off=so, seq_pool=sp)


# ---- mem_chain2aln (== memChainToAlnBatched per read, MemChainToAlignBatched.scala:380-616; BWA z-drop parse) -------
l_pac3 = 60_013
pac3, bases3 = synth.random_pac(l_pac3, seed=20261005)
cb = synth.read_chains(400, bases3, l_pac3, read_len=150, sub_rate=0.03, indel_rate=0.006, tail_frac=0.08, seed=20261006)
opt3 = po.Oracle().default_opt()
cnt3, regs3 = ref.chain2aln_batch(opt3, pac3, cb)
np.savez_compressed(os.path.join(HERE, "mem_chain2aln.npz"), l_pac=l_pac3, pac=pac3, read_len=cb.read_len, read_off=cb.read_off,
                    read_pool=cb.read_pool, chain_cnt=cb.chain_cnt, seed_cnt=cb.seed_cnt, seed_rbeg=cb.seed_rbeg,
                    seed_qbeg=cb.seed_qbeg, seed_len=cb.seed_len, out_cnt=cnt3, out_regs=regs3)

# ---- worker2's tail: mem_reg2aln and mem_sam_pe (== memRegToAln / memSamPeGroupRest, C flavour) --------------------
# Regions come from the reference's own mem_chain2aln + mem_sort_and_dedup on synthetic FR pairs over a four-contig
# reference with duplicated stretches; mem_sam_pe runs with MEM_F_NO_RESCUE (the rescue is boundary 1's business).
import bpsw_hip  # noqa: E402

contigs = [60_000, 40_000, 25_000, 35_000]
pac4, bases4, ann_off4, ann_len4, ann_names4, dups4 = synth.contig_reference(contigs, seed=20261007)
opt4 = po.Oracle().default_opt()
topt4 = po.Oracle().default_tail_opt()


def regions_of(batch, opt):
    cnt, regs = ref.chain2aln_batch(opt, pac4, batch)
    out_cnt, out, at = [], [], 0
    for c in cnt:
        r = ref.sort_dedup(regs[at:at + c]) if c else regs[0:0]
        at += c
        out_cnt.append(len(r)); out.append(r)
    return np.array(out_cnt, np.int32), np.concatenate(out)


for stem, n_pairs, es, ei, flag, seed, rg_id in (("mem_sam_pe", 160, 0.02, 0.006, 0, 20261008, b""),
                                                 ("mem_sam_pe_all", 90, 0.03, 0.01, po.MEM_F_ALL, 20261009, b""),
                                                 ("mem_sam_pe_rg", 60, 0.03, 0.01, po.MEM_F_ALL, 20261011, b"lane7.A")):  # -R "@RG\tID:lane7.A\t..."
    tb, names, quals, pes = synth.tail_pairs(n_pairs, bases4, ann_off4, ann_len4, dups4, sub_rate=es, indel_rate=ei, seed=seed)
    rc, rg = regions_of(tb, opt4)
    g4 = bpsw_hip.make_tail_group(tb, names, quals, pes, rc, rg, ann_off4, ann_len4, ann_names4, id0=4242)
    o = po.Oracle().default_opt()
    o.flag = flag
    topt4.rg_id = rg_id
    texts = ref.sam_pe_batch(o, topt4, pac4, g4)
    topt4.rg_id = b""
    t_off = np.zeros(len(texts) + 1, np.int64)
    t_off[1:] = np.cumsum([len(t) for t in texts])
    np.savez_compressed(os.path.join(HERE, stem + ".npz"), l_pac=g4.l_pac, pac=pac4, id0=g4.id0, flag=flag, pes=np.array(pes, np.float64),
                        read_len=g4.read_len, read_off=g4.read_off, read_pool=g4.read_pool, qual_pool=g4.qual_pool,
                        name_off=g4.name_off, name_pool=g4.name_pool, reg_cnt=g4.reg_cnt, regs=g4.regs, ann_off=g4.ann_off,
                        ann_len=g4.ann_len, ann_name_off=g4.ann_name_off, ann_name_pool=g4.ann_name_pool,
                        text=np.frombuffer(b"".join(texts), np.uint8), text_off=t_off, **({"rg_id": np.frombuffer(rg_id, np.uint8)} if rg_id else {}))

# mem_reg2aln on every region of 250 pairs (primary and secondary alike), plus the unmapped record
tb, names, quals, pes = synth.tail_pairs(250, bases4, ann_off4, ann_len4, dups4, sub_rate=0.03, indel_rate=0.01, p_span=0.08, seed=20261010)
rc, rg = regions_of(tb, opt4)
rg = rg.copy()
jl, jo = [], []
for r in range(tb.n_reads):
    for _ in range(int(rc[r])):
        jl.append(int(tb.read_len[r])); jo.append(int(tb.read_off[r]))
at = 0
for r in range(tb.n_reads):        # what mem_mark_primary_se would leave: give some regions a parent and sub scores
    c = int(rc[r])
    rg[at:at + c] = ref.mark_primary(opt4, topt4, rg[at:at + c], 2 * r)
    at += c
jl.append(int(tb.read_len[0])); jo.append(int(tb.read_off[0]))
unm = np.zeros(1, rg.dtype)
unm["rb"] = -1; unm["re"] = -1
rg = np.concatenate([rg, unm])
alns, cig, md = ref.reg2aln_batch(opt4, topt4, pac4, int(sum(contigs)), ann_off4, ann_len4, np.array(jl, np.int32), np.array(jo, np.int64),
                                  tb.read_pool, rg, cigar_cap=32, md_cap=160)
assert int(alns["n_cigar"].max()) <= 32 and int(alns["md_len"].max()) <= 160
np.savez_compressed(os.path.join(HERE, "mem_reg2aln.npz"), l_pac=int(sum(contigs)), pac=pac4, ann_off=ann_off4, ann_len=ann_len4,
                    read_len=np.array(jl, np.int32), read_off=np.array(jo, np.int64), read_pool=tb.read_pool, regs=rg, alns=alns,
                    cigar=cig, md=md)

# ---- whole extension tasks through the reference's ksw_extend2 (ref_extend_batch, oracle/ref_shim.c: the extension() control of
# MemChainToAlignBatched.scala:789-883 around native/ksw.c:379-476): two-sided tasks with their band retries, for the kernels' own
# row loops to be held against reference OUTPUTS (tests/test_golden_gpu.py) -- 150 and 250 bp, 1-20 % error, bands 5 / 70 / 100 / 127.
# (own generator: the shared one above must keep its sequence for the other files)
sets, fields = [], ("left_qlen", "left_rlen", "right_qlen", "right_rlen", "left_q_off", "left_r_off", "right_q_off", "right_r_off",
                    "reg_score", "q_beg", "h0", "idx")
ext_kw = {}
for si, (rl, sub, indel, w, zd) in enumerate(((150, 0.01, 0.001, 100, 100), (150, 0.05, 0.01, 100, 100), (150, 0.10, 0.02, 70, 100),
                                               (150, 0.20, 0.02, 5, 100), (150, 0.02, 0.05, 5, 100), (250, 0.08, 0.02, 100, 100),
                                               (250, 0.20, 0.02, 127, 100), (250, 0.05, 0.05, 5, 20), (250, 0.03, 0.004, 127, 0))):
    soa = synth.ext_tasks(90, read_len=rl, sub_rate=sub, indel_rate=indel, n_rate=0.002, seed=synth.CONFIG_SEED_BASE + 4000 + si)
    soa.w = w
    out = ref.extend_batch(soa, mat, zdrop=zd)
    for f in fields:
        ext_kw[f"s{si}_{f}"] = getattr(soa, f)
    ext_kw[f"s{si}_pool"] = soa.pool
    ext_kw[f"s{si}_out"] = out
    sets.append((rl, int(round(sub * 1000)), int(round(indel * 1000)), w, zd, soa.n))
# one more set, built by hand: right-side flanks of ~200 bases with ONE indel of 60-95 bases behind 40 matching ones -- at band 100 an
# alignment that ends 75 or more columns off the diagonal makes extension() run ksw_extend2 again with the band doubled to 200
import bpsw_hip  # noqa: E402
rg2 = np.random.default_rng(20261004)
pool, fl = [], {k: [] for k in fields}
for k in range(48):
    core = rg2.integers(0, 4, 200).astype(np.uint8)
    gap = int(rg2.integers(60, 96))
    extra = rg2.integers(0, 4, gap).astype(np.uint8)
    if k % 2 == 0:   # the target has `gap` bases more (a deletion in the read)
        q, t = core, np.concatenate([core[:40], extra, core[40:]])
    else:            # the read has them (an insertion)
        q, t = np.concatenate([core[:40], extra, core[40:200 - gap if gap < 90 else 110]]), core
    t = np.concatenate([t, rg2.integers(0, 4, 60).astype(np.uint8)])
    for j in rg2.integers(0, len(q), 2):
        q[j] = (q[j] + 1) & 3
    fl["left_qlen"].append(0); fl["left_rlen"].append(0); fl["left_q_off"].append(0); fl["left_r_off"].append(0)
    fl["right_qlen"].append(len(q)); fl["right_rlen"].append(len(t))
    fl["right_q_off"].append(len(pool)); pool.extend(q.tolist())
    fl["right_r_off"].append(len(pool)); pool.extend(t.tolist())
    h = int(rg2.integers(90, 140))   # (the alignment must survive the gap: its score at the gap exceeds the gap's cost)
    fl["reg_score"].append(h); fl["h0"].append(h); fl["q_beg"].append(0); fl["idx"].append(k)
si = len(sets)
soa = bpsw_hip.ExtTaskSoA(pool=np.array(pool + [0] * 16, np.uint8), **{k: np.array(v, np.int64 if k.endswith("_off") else np.int32) for k, v in fl.items()})
soa.w = 100
out = ref.extend_batch(soa, mat, zdrop=100)
for f in fields:
    ext_kw[f"s{si}_{f}"] = getattr(soa, f)
ext_kw[f"s{si}_pool"] = soa.pool
ext_kw[f"s{si}_out"] = out
sets.append((200, 10, 0, 100, 100, soa.n))
np.savez_compressed(os.path.join(HERE, "ref_extension_tasks.npz"), sets=np.array(sets, np.int32), **ext_kw,
                    note="ref_extend_batch: per set s<i>_*: ExtTaskSoA fields + out[n,7] = qBeg,qEnd,rBeg,rEnd,score,trueScore,width; "
                         "sets rows = read_len, sub permille, indel permille, w, zdrop, n; scoring 1/-4/6/1/6/1, clip 5/5, BWA z-drop parse")

if CHECK:
    if _mismatch:
        print("golden fixtures differ from what this script generates now:", ", ".join(_mismatch))
        sys.exit(1)
    print("all committed fixtures regenerate bit-identically")
