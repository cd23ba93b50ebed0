"""Exhaustive check of the extension kernel -- and in particular of its exact shortcuts (closed form for near-exact flanks,
single-gap certificate, two gap opens, start-gap form, tail-row bound: csrc/bpsw_extend_core.h), which resolve most flanks of a
2x150 bp batch without running the DP and one of which was wrong once while the random suite stayed green -- against the
oracle's full DP (oracle/bpsw_oracle.c: orc_sw_extend == SWUtil.scala:61-230, orc_extension == MemChainToAlignBatched.scala:
789-883) on EVERY short flank:

  query 1..7 bases over a two-letter alphabet (first base fixed: the matrices are symmetric under relabelling), target
  1..qLen+3 bases, all of them; for queries up to 5 bases also every variant with an N at one query or one target position:
  298 078 flanks, each once as the left and once as the right side of a seed;
  x seed score h0 in {1, 5, 19, 40}
  x four scorings of the family the shortcuts accept (match 1, everything else <= -1, gap extension 1) with gap opens 6, 1, 3, 2
  x band width w in {1, 2, 100}
  x z-drop (100, Scala parse), (3, Scala parse), (3, BWA parse)
  = 86 M task runs with every shortcut on, and a sixth of that again with each shortcut level taken away in turn
  (bpsw_set_ext_shortcuts), so that a flank is seen by whichever form claims it at every level.
Short flanks over two letters are where shifted diagonals match all the time, i.e. where the shortcuts' exclusion tests work
hardest."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import bpsw_hip
import pyoracle as po

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import exhaustive_flanks as ef  # noqa: E402

pytestmark = pytest.mark.gpu


def _mat(b, nscore):
    m = po.default_mat(1, b).copy()
    for k in range(5):
        m[4 * 5 + k] = nscore; m[k * 5 + 4] = nscore
    return m


SCORINGS = [(po.default_mat(), 6), (_mat(1, -1), 1), (_mat(2, -3), 3), (_mat(6, -1), 2)]   # (matrix, gap open); gap extension 1
ZDROPS = [(100, po.ZDROP_SCALA), (3, po.ZDROP_SCALA), (3, po.ZDROP_BWA)]
LEVELS_FULL = 63
LEVELS_REDUCED = [0, 1, 1 | 16, 1 | 2 | 16, 1 | 2 | 4 | 16, 31, 1 | 32, 1 | 2 | 4 | 16 | 32]   # pure DP; closed form only; + tail bound; + certificate; + two gap opens (no start-gap form); all forms in ext_kernel alone; the sift kernel at two levels


def test_every_short_flank_at_every_shortcut_level():
    F = ef.enumerate_flanks(max_q=7, extra_t=3, n_variants_up_to=5)
    assert F[1].shape[0] == 298_078
    orc_pool = ThreadPoolExecutor(8)
    ctxs = {m: bpsw_hip.Context(0) for m in [LEVELS_FULL] + LEVELS_REDUCED}
    for m, c in ctxs.items():
        c.set_ext_shortcuts(m)
    runs = bad = 0
    first_bad = None
    for si, (mat, o) in enumerate(SCORINGS):
        for w in (1, 2, 100):
            work = []   # (wire, n) per (h0, side)
            for h0 in (1, 5, 19, 40):
                for left in (False, True):
                    soa = ef.flank_tasks(*F, h0=h0, left=left)
                    soa.o_del = soa.o_ins = o
                    soa.e_del = soa.e_ins = 1
                    soa.w = w
                    work.append((bpsw_hip.wire_pack(soa), soa.n, h0, left))
            for zdrop, zmode in ZDROPS:
                # the other shortcut levels see a sixth of the settings: one z-drop setting per (scoring, band) in rotation
                levels = [LEVELS_FULL] + (LEVELS_REDUCED if (si + w + zdrop + zmode) % 3 == 0 and w != 2 else [])
                wants = list(orc_pool.map(lambda x: orc_pool_oracle(x[0], mat, zdrop, zmode), work))
                for m in levels:
                    ctxs[m].set_ext_scoring(mat, zdrop, zmode)
                    for (wire, n, h0, left), want in zip(work, wants):
                        got = ctxs[m].extend_batch(wire).reshape(-1, 10)
                        diff = np.nonzero((got != want).any(axis=1))[0]
                        runs += n
                        if diff.size:
                            bad += int(diff.size)
                            if first_bad is None:
                                first_bad = (f"scoring {si} gap open {o} w {w} zdrop {zdrop} parse {zmode} shortcuts {m} h0 {h0} "
                                             f"{'left' if left else 'right'} task {int(diff[0])}: got {got[diff[0]]} want {want[diff[0]]}")
    for c in ctxs.values():
        c.close()
    assert bad == 0, f"{bad} of {runs} task runs differ; first: {first_bad}"
    assert runs > 195_000_000


_tls_orc = {}


def orc_pool_oracle(wire, mat, zdrop, zmode):
    import threading
    k = threading.get_ident()
    if k not in _tls_orc:
        _tls_orc[k] = po.Oracle()
    want, _ = _tls_orc[k].wire_extend(wire, mat, zdrop, zmode)
    return np.asarray(want).reshape(-1, 10)
