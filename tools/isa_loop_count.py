#!/usr/bin/env python3
"""Static instruction census of the innermost loops of a kernel in a --save-temps .s file: for every innermost loop (label with
'This Inner Loop Header') of functions whose name contains PATTERN, the number of vector / scalar / branch / other instructions
between the header and the last back edge.  A proxy for the per-row cost of the extension sweeps while iterating on the source
without a GPU (the counters of tools/pmc_rowcost.sh are the truth).  Usage: tools/isa_loop_count.py FILE.s [PATTERN]"""
import re
import sys

fn = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "ext_kernel"
lines = open(fn).read().split("\n")
func = None
i = 0
out = []
while i < len(lines):
    ln = lines[i]
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        func = m.group(1)
    if func and pat in func and "This Inner Loop Header" in ln:
        k = i
        while k > 0 and not lines[k].startswith(".LBB"):
            k -= 1
        label = lines[k].split(":")[0].strip()
        # find the last branch back to this label before the next function end
        j, last = i + 1, None
        while j < len(lines) and not lines[j].startswith("\t.end_amdhsa_kernel") and ".Lfunc_end" not in lines[j]:
            if re.search(r"s_c?branch\w*\s+" + re.escape(label) + r"\b", lines[j]):
                last = j
            j += 1
        if last:
            body = [l.strip() for l in lines[i + 1:last + 1] if l.startswith("\t") and not l.strip().startswith(";") and not l.strip().startswith(".")]
            v = sum(1 for l in body if l.startswith("v_") or l.startswith("ds_") or l.startswith("global_") or l.startswith("flat_") or l.startswith("scratch_"))
            br = sum(1 for l in body if l.startswith("s_cbranch") or l.startswith("s_branch"))
            misc = sum(1 for l in body if l.startswith("s_nop") or l.startswith("s_waitcnt"))
            s = sum(1 for l in body if l.startswith("s_")) - br - misc
            dpp = sum(1 for l in body if "row_" in l or "wave_shr" in l)
            out.append((func[:60], label, i + 1, last + 1, len(body), v, s, br, misc, dpp))
    i += 1
for o in out:
    print("%s %s lines %d-%d: total %d  vector %d  scalar %d  branch %d  nop/wait %d  (dpp %d)" % o)
