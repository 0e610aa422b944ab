#!/usr/bin/env python
"""Summarise rocprofv3 rocpd databases (gpurun_out/prof_<tag>/) into a small text report for profiles/."""
import json
import os
import sqlite3
import sys
from collections import defaultdict


def short(name):
    # "void afq::(anonymous namespace)::k_atac_parse(afq::AtacArgs)": the namespace's parenthesis is not the argument list's
    n = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("afq::", "")
    return n.strip() or name


def main(tag):
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", f"prof_{tag}")
    out = []
    db = sqlite3.connect(os.path.join(root, "stats", "stats_results.db"))
    cmd = "python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
    if os.path.exists(os.path.join(root, "command.txt")):
        cmd = open(os.path.join(root, "command.txt")).read().strip().replace(os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/", "")
        cmd = " ".join(w.split("/")[-1] if w.endswith("bench.py") else w for w in cmd.split())
    out.append(f"# rocprofv3 --kernel-trace --stats  (tag {tag}; command: {cmd})")
    out.append(f"{'kernel':28s} {'calls':>6s} {'total_us':>12s} {'avg_us':>12s} {'pct':>7s}")
    for name, calls, tot, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        out.append(f"{short(name):28s} {calls:6d} {tot/1:12.1f} {avg:12.1f} {pct:7.2f}")
    try:
        b = json.load(open(os.path.join(root, "bench_stats.json")))
        out.append("")
        out.append("bench.py under the stats pass: ms_per_step=%s value=%s M reads/s" % (b["ms_per_step"], b["value"]))
        out.append("bench.py HIP-event kernel ms/step: " + json.dumps(b["roofline"]["all_kernels_ms_per_step"]))
    except Exception as e:  # noqa
        out.append(f"(no bench json: {e})")
    for sub in ("pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write", "pmc_tcc"):
        p = os.path.join(root, sub, "pmc_results.db")
        if not os.path.exists(p):
            continue
        db = sqlite3.connect(p)
        acc = defaultdict(lambda: defaultdict(float))
        nd = defaultdict(set)
        for kn, cn, val, did in db.execute("select kernel_name,counter_name,value,dispatch_id from counters_collection"):
            acc[short(kn)][cn] += val
            nd[short(kn)].add(did)
        out.append("")
        out.append(f"# --pmc pass {sub}: per-launch average of each counter (sum over dispatches / #dispatches)")
        for k in acc:
            n = max(1, len(nd[k]))
            out.append(f"{k:28s} launches={n:3d} " + " ".join(f"{c}={v/n:.4g}" for c, v in sorted(acc[k].items())))
    print("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r1")
