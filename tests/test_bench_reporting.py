"""bench.py's reporting path on CPU: the roofline block's cross-check against the committed rocprofv3 summaries.

Round 5's line carried `avg_launch_ms_kernel_trace: null` and `dominant_by: issued wave-instructions` in every run: the trace loop used a
name before its assignment and a blanket `except Exception` hid the UnboundLocalError.  The parsing is now three small functions
(bench.trace_figures, bench.pick_dominant, bench.load_committed_json) that only forgive a MISSING file; this test runs them on the
summaries the repository commits for bench.PROFILE_TAG."""
import json
import os

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROFILES = os.path.join(ROOT, "profiles")


def _csv():
    return os.path.join(PROFILES, f"{bench.PROFILE_TAG}_kernel_stats_bench.csv")


def test_the_committed_kernel_trace_of_the_profile_tag_exists():
    assert os.path.isfile(_csv()), f"bench.PROFILE_TAG = {bench.PROFILE_TAG!r} names no committed kernel trace"


def test_trace_figures_are_not_null_on_the_committed_trace():
    avg, share = bench.trace_figures(_csv())
    # every kernel family of the configs[2] step has an average launch duration and a share of kernel time
    assert avg.get("extend") and avg["extend"] > 0
    assert avg.get("swalign2_resident") and avg["swalign2_resident"] > 0
    assert avg.get("swalign2_resident_share_of_kernel_time") is not None
    assert share.get("extend") and 0 < share["extend"] < 1
    assert share.get("swalign2_resident") and 0 < share["swalign2_resident"] < 1
    # a row that is neither an extension kernel of format 1 nor the resident kernel (ext_kernel<true, 1>, reg2aln_kernel ...) is skipped, not fatal
    names = open(_csv()).read()
    assert "ext_kernel<true" in names or "reg2aln_kernel" in names


def test_the_dominant_kernel_is_named_by_the_trace_share():
    avg, share = bench.trace_figures(_csv())
    dom, by = bench.pick_dominant(share, 10, 20, 1.0, 2.0, bench.PROFILE_TAG)
    assert by.startswith("share of summed kernel time in profiles/")
    sw = share.get("swalign2_resident", 0.0) + share.get("swalign2", 0.0)
    assert dom == ("extend" if share["extend"] >= sw else "swalign2")
    # the fallbacks, in their order
    assert bench.pick_dominant({}, 10, 20, 5.0, 1.0, "x") == ("swalign2", "issued wave-instructions (profiles/pmc_issue.json x launches)")
    assert bench.pick_dominant({}, None, None, 5.0, 1.0, "x")[0] == "extend"


def test_only_a_missing_file_is_forgiven(tmp_path):
    assert bench.trace_figures(str(tmp_path / "absent.csv")) == ({}, {})
    bad = tmp_path / "bad.csv"
    bad.write_text("Name,Calls\n\"ext_kernel<false, 1>\",3\n")      # a column the parser needs is gone
    with pytest.raises(KeyError):
        bench.trace_figures(str(bad))
    assert bench.load_committed_json("no_such_file.json") == {}


def test_committed_counter_files_parse_and_carry_what_the_line_quotes():
    issue = bench.load_committed_json("pmc_issue.json")
    traffic = bench.load_committed_json("pmc_traffic.json")
    assert (issue.get("extend_per_call") or issue.get("extend")) and issue.get("swalign2")
    for k in ("valu", "salu"):
        assert (issue.get("extend_per_call") or issue.get("extend"))[k] > 0 and issue["swalign2"][k] > 0
    assert traffic.get("extend") and traffic.get("swalign2")
    sa = bench.load_committed_json(f"{bench.PROFILE_TAG}_sq_activity.json")
    assert "extend" in sa and "swalign2" in sa
    json.dumps([issue, traffic, sa])


def test_the_committed_cpu_sweep_is_there_and_monotone():
    """host.reads_per_s_vs_cpus of the default line comes from profiles/<tag>_cpu_sweep.json (python bench.py --cpu-sweep measures it live)"""
    sw = bench.load_committed_json(f"{bench.PROFILE_TAG}_cpu_sweep.json")
    pts = sw["reads_per_s_vs_cpus"]
    rates = [pts[k]["reads_per_s"] for k in sorted(pts, key=int)]
    assert len(rates) >= 3 and all(b >= a for a, b in zip(rates, rates[1:])), rates
    assert sw["unconfined"]["reads_per_s"] > 0
