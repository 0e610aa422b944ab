#!/usr/bin/env python
"""HBM-side bytes per launch from the FETCH_SIZE / WRITE_SIZE passes of run_prof.sh -> profiles/<round>_traffic.json
(read by bench.py for roofline.traffic).  Usage: python profiles/traffic.py <gpurun tag> <output name> [output dir]"""
import json
import os
import sqlite3
import sys
from collections import defaultdict


def short(name):
    n = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("afq::", "").split("<")[0].strip()
    return n or name


def per_launch(db_path, counter, totals=None):
    db = sqlite3.connect(db_path)
    tot, disp = defaultdict(float), defaultdict(set)
    for kn, cn, val, did in db.execute("select kernel_name,counter_name,value,dispatch_id from counters_collection"):
        if cn == counter:
            tot[short(kn)] += val
            disp[short(kn)].add(did)
    if totals is not None:
        for k in tot:
            totals[k] = (tot[k], len(disp[k]))
    return {k: tot[k] / max(1, len(disp[k])) for k in tot}


def main(tag, rnd, outdir=None):
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.join(os.path.dirname(here), "gpurun_out", f"prof_{tag}")
    ft, wt = {}, {}
    f = per_launch(os.path.join(root, "pmc_fetch", "pmc_results.db"), "FETCH_SIZE", ft)
    w = per_launch(os.path.join(root, "pmc_write", "pmc_results.db"), "WRITE_SIZE", wt)
    cmd = open(os.path.join(root, "command.txt")).read().strip() if os.path.exists(os.path.join(root, "command.txt")) else ""
    words = cmd.split()
    steps_prof = (int(words[words.index("--steps") + 1]) + int(words[words.index("--warmup") + 1])) if "--steps" in words and "--warmup" in words else None
    cmd_short = " ".join(x.split("/")[-1] if x.endswith("bench.py") else x for x in words)
    out = {"_about": f"HBM-side bytes from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tag {tag}, command: {cmd_short}). "
                     "Per kernel: per dispatch (bytes_per_launch_*) and over the whole profiled run (bytes_total_*: steps_profiled steps, warm-up included; "
                     "bench.py divides by steps and by the launches of the timer that brackets the kernel). The counters report KB. "
                     "MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half the bytes of coalesced streaming reads; calibrated on k_hist, "
                     "which reads exactly 8 B per key (see DESIGN.md): traffic = 2 x FETCH_SIZE + WRITE_SIZE (bytes_per_launch_fetch_doubled) "
                     "is what bench.py reports as roofline.traffic.",
           "steps_profiled": steps_prof, "kernels": {}}
    for k in f:
        fk, wk = f[k], w.get(k, 0.0)
        out["kernels"][k] = {"FETCH_SIZE_KB_per_launch": round(fk, 1), "WRITE_SIZE_KB_per_launch": round(wk, 1),
                             "bytes_per_launch_raw": int((fk + wk) * 1024), "bytes_per_launch_fetch_doubled": int((2 * fk + wk) * 1024),
                             "dispatches": ft[k][1], "bytes_total_fetch_doubled": int((2 * ft[k][0] + wt.get(k, (0.0, 0))[0]) * 1024)}
    with open(os.path.join(outdir or here, f"{rnd}_traffic.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    for k in ("k_hist", "k_resolve", "k_decode_recs", "k_scatter"):
        if k in out["kernels"]:
            print(k, out["kernels"][k])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
