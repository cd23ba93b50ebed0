#!/usr/bin/env python3
"""Where and when the wavefronts of a bench run were resident (GPU box; needs the diagnostics build of the library:
   tools/build_variant.sh waves -DBPSW_DIAG_WAVES).

   python tools/wave_placement.py OUT_PREFIX [bench args]      e.g.  gpurun_out/r04_waves --config 3 --steps 2 --warmup 1

Runs bench.main() in this process against lib_exp/libbPSW_hip_waves.so, then reads the kernels' wave logs (csrc/bpsw_diag_waves.h:
start / end on the 100 MHz clock, HW_ID, XCC_ID, kernel kind, a tag of the calling context) and prints
 - the average number of resident waves per CU over the middle of the logged interval, CU by CU (is the load spread evenly?),
 - per kernel kind: wave lifetime, and per launch (waves of one context between two idle gaps) the launch's span, the spread
   of its waves' start times and the share of span x waves its waves were resident for."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("BPSW_LIB", os.path.join(ROOT, "cloud-scale-bwamem_amd", "lib_exp", "libbPSW_hip_waves.so"))
for p in ("cloud-scale-bwamem_amd", ""):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402

prefix = sys.argv[1]
sys.argv = ["bench.py"] + sys.argv[2:] + ["--no-cpu-baseline", "--no-extras"]
import bench  # noqa: E402
import bpsw_hip  # noqa: E402

bench.main()
lib = bpsw_hip.load_library()
logs, last = [], []
for name in ("ext", "sw"):
    path = f"{prefix}_{name}.bin"
    n = getattr(lib, "bpsw_diag_dump_waves_" + name)(path.encode())
    print(f"# {name}: {n} waves logged", file=sys.stderr)
    if n > 0:
        logs.append(np.fromfile(path, np.uint32).reshape(-1, 8))
        last.append(len(logs[-1]))
    if os.path.exists(path):
        os.remove(path)
a = np.concatenate(logs)
t0 = a[:, 0].astype(np.int64); t1 = a[:, 1].astype(np.int64)
t1 = np.where(t1 < t0, t1 + (1 << 32), t1)    # the low word wrapped between start and end
# unwrap the starts around the median (a run is far shorter than the 43 s period of the low word)
med = np.median(t0)
wrap = t0 < med - (1 << 31)
t0 = np.where(wrap, t0 + (1 << 32), t0); t1 = np.where(wrap, t1 + (1 << 32), t1)
hw = a[:, 2]; w3 = a[:, 3]
x_last, x_last_dur, x_n, x_max = (a[:, 4 + i].astype(np.int64) for i in range(4))   # ext_kernel: the wave's last task, its ticks, tasks swept, the longest
xcc = (w3 & 0xf).astype(np.int64); kind = ((w3 >> 4) & 0xf).astype(np.int64); tag = (w3 >> 8).astype(np.int64)
cu = ((hw >> 8) & 0xff).astype(np.int64)     # CU_ID, SH_ID, SE_ID
simd = ((hw >> 4) & 3).astype(np.int64)
# the logs fill up front and stop when full: the window ends where the first of them ends
ends, at = [], 0
for n in last:
    ends.append(np.percentile(t0[at:at + n], 99)); at += n
hi = min(ends) - 100000   # 1 ms
lo = t0.min() + 0.4 * (hi - t0.min())
print(f"window {1e-5 * (hi - lo):.2f} ms of {1e-5 * (t1.max() - t0.min()):.2f} ms logged; {len(a)} waves")
ov = np.clip(np.minimum(t1, hi) - np.maximum(t0, lo), 0, None)
key = xcc * 256 + cu
res = np.bincount(key, weights=ov, minlength=8 * 256) / (hi - lo)
used = np.nonzero(np.bincount(key, minlength=8 * 256))[0]
r = res[used]
print(f"CUs seen {len(used)}; resident waves per CU: mean {r.mean():.2f} min {r.min():.2f} p10 {np.percentile(r, 10):.2f} median {np.median(r):.2f} p90 {np.percentile(r, 90):.2f} max {r.max():.2f}")
for x in range(8):
    m = used[(used // 256) == x]
    print(f"  XCC {x}: {len(m)} CUs, mean {res[m].mean():.2f}; by CU " + " ".join(f"{v:.1f}" for v in res[m]))
# the same per SIMD (is a CU's load on one SIMD?)
skey = key * 4 + simd
sres = np.bincount(skey, weights=ov, minlength=8 * 256 * 4) / (hi - lo)
sused = np.nonzero(np.bincount(skey, minlength=8 * 256 * 4))[0]
print(f"resident waves per SIMD: mean {sres[sused].mean():.2f} p10 {np.percentile(sres[sused], 10):.2f} median {np.median(sres[sused]):.2f} p90 {np.percentile(sres[sused], 90):.2f} max {sres[sused].max():.2f}")
# time-resolved: share of the window in which a SIMD holds 0, 1, 2, ... waves (sampled)
samples = np.linspace(lo, hi, 400)
occ = np.zeros(12)
for s in samples:
    live = (t0 <= s) & (t1 > s)
    c = np.bincount(skey[live], minlength=8 * 256 * 4)[sused]
    occ += np.bincount(np.minimum(c, 11), minlength=12)
occ /= occ.sum()
print("share of SIMD-time with k resident waves, k = 0..11+: " + " ".join(f"{v:.3f}" for v in occ))

# rescue kernel: what the waves' job pairs looked like, and how a wave's lifetime follows from it
sw = kind == 3
if sw.any():
    tl0, tl1 = x_last[sw] & 0xffff, x_last[sw] >> 16
    r0, r1 = x_last_dur[sw] & 0xffff, x_last_dur[sw] >> 16
    p0, p1 = x_n[sw] & 0xffff, x_n[sw] >> 16
    life = (t1 - t0)[sw] * 1e-2
    both = (tl0 > 0) & (tl1 > 0)
    pc = lambda v: " ".join(f"{np.percentile(v, q):.0f}" for q in (1, 10, 50, 90, 99))
    print(f"rescue job pairs: window rows p1/10/50/90/99: {pc(np.concatenate([tl0[tl0 > 0], tl1[tl1 > 0]]))}; rows the first pass swept: {pc(np.concatenate([r0[tl0 > 0], r1[tl1 > 0]]))}; "
          f"rows of the second pass (0 = none): {pc(np.concatenate([p0[tl0 > 0], p1[tl1 > 0]]))}; jobs with a second pass {np.mean(np.concatenate([p0[tl0 > 0], p1[tl1 > 0]]) > 0):.3f}")
    steps = np.maximum(r0, r1) + np.maximum(p0, p1)      # the pair advances together: a pass takes as many steps as its longer job
    ideal = (r0 + r1 + p0 + p1) / 2.0
    print(f"   steps of a pair (max of the two jobs per pass) mean {steps.mean():.0f}, half the sum of its jobs' rows {ideal.mean():.0f}: pairing costs {steps.mean() / ideal.mean() - 1:.3f}; "
          f"lifetime us per step {np.median(life[steps > 0] / steps[steps > 0]):.3f}; corr(lifetime, steps) {np.corrcoef(life, steps)[0, 1]:.3f}")
    print(f"   steps p1/10/50/90/99: {pc(steps)}; lifetime us: {pc(life)}")

names = {0: "ext_kernel<.,0> (full)", 1: "ext_kernel<.,1> (short)", 3: "swp_kernel"}
for k in sorted(set(kind.tolist())):
    sel = kind == k
    life = (t1 - t0)[sel] * 1e-2   # us
    print(f"{names.get(k, k)}: {sel.sum()} waves, lifetime us mean {life.mean():.1f} median {np.median(life):.1f} p90 {np.percentile(life, 90):.1f} max {life.max():.1f}")
    spans, spreads, fills, nw = [], [], [], []
    start_off, end_off, xcc_first, xcc_spread, late = [], [], [], [], []
    for tg in set(tag[sel].tolist()):
        m = sel & (tag == tg)
        o = np.argsort(t0[m]); s0 = t0[m][o]; s1 = t1[m][o]; sx = xcc[m][o]
        e_last, e_dur, e_n, e_max = x_last[m][o], x_last_dur[m][o], x_n[m][o], x_max[m][o]
        start = 0; cur_end = s1[0]
        for i in range(1, len(s0) + 1):
            if i == len(s0) or s0[i] > cur_end:      # nothing of this context was resident: the next launch
                b0, b1 = s0[start:i], s1[start:i]
                span = b1.max() - b0.min()
                if span > 0 and lo <= b0.min() <= hi:
                    spans.append(span * 1e-2); spreads.append((b0.max() - b0.min()) * 1e-2)
                    fills.append((b1 - b0).sum() / (span * len(b0))); nw.append(len(b0))
                    if len(spans) % 8 == 0:   # a sample of the launches: when their waves start / end within the launch, and XCC by XCC
                        start_off.append((b0 - b0.min()) * 1e-2); end_off.append((b1.max() - b1) * 1e-2)
                        bx = sx[start:i]
                        if k == 1:   # the five waves that leave last: (us before the launch's end, last task, its us, tasks swept, longest us, lifetime us)
                            for j in np.argsort(b1)[-5:]:
                                late.append(((b1.max() - b1[j]) * 1e-2, int(e_last[start + j]), e_dur[start + j] * 1e-2, int(e_n[start + j]), e_max[start + j] * 1e-2, (b1[j] - b0[j]) * 1e-2))
                        first = np.array([(b0[bx == x].min() - b0.min()) * 1e-2 for x in range(8) if (bx == x).any()])
                        xcc_first.append(first.max())
                        xcc_spread.append(np.mean([(b0[bx == x].max() - b0[bx == x].min()) * 1e-2 for x in range(8) if (bx == x).any()]))
                start = i
                if i < len(s0):
                    cur_end = s1[i]
            else:
                cur_end = max(cur_end, s1[i])
    if spans:
        print(f"   launches {len(spans)}: waves/launch {np.mean(nw):.0f}; span us mean {np.mean(spans):.1f} median {np.median(spans):.1f}; "
              f"start spread us mean {np.mean(spreads):.1f} median {np.median(spreads):.1f} p90 {np.percentile(spreads, 90):.1f}; "
              f"resident share of span x waves mean {np.mean(fills):.3f}")
        so, eo = np.concatenate(start_off), np.concatenate(end_off)
        pc = lambda v: " ".join(f"{np.percentile(v, q):.0f}" for q in (10, 25, 50, 75, 90, 99))
        print(f"   wave start after the launch's first wave, us p10/25/50/75/90/99: {pc(so)};  wave end before the launch's last, us: {pc(eo)}")
        if late:
            L = np.array(late)
            print(f"   the five last waves of a launch: end {L[:, 0].mean():.0f} us before the launch's (mean); their last task took {L[:, 2].mean():.0f} us (median {np.median(L[:, 2]):.0f}, p90 {np.percentile(L[:, 2], 90):.0f}); tasks swept {L[:, 3].mean():.1f}; lifetime {L[:, 5].mean():.0f} us")
            print("   the very last wave of each sampled launch (its last task, us of it, tasks swept, longest us, lifetime us): " + "; ".join(f"{int(r[1])} {r[2]:.0f} {int(r[3])} {r[4]:.0f} {r[5]:.0f}" for r in L[4::5][:12]))
            lm = x_max[sel] * 1e-2; ln = x_n[sel]
            print(f"   all waves: tasks swept mean {ln.mean():.1f} p10 {np.percentile(ln, 10):.0f} p90 {np.percentile(ln, 90):.0f}; longest task of a wave us p50 {np.percentile(lm, 50):.0f} p90 {np.percentile(lm, 90):.0f} p99 {np.percentile(lm, 99):.0f} p99.9 {np.percentile(lm, 99.9):.0f} max {lm.max():.0f}")
        print(f"   the last XCC's first wave starts {np.mean(xcc_first):.1f} us after the launch's first (mean); spread of starts within one XCC {np.mean(xcc_spread):.1f} us (mean)")
