#!/usr/bin/env python
"""Text timeline of the LAST step of a bench run from a rocprofv3 rocpd database (--kernel-trace --memory-copy-trace):
every kernel and memory copy with its queue / stream, start (us from the step's first kernel) and duration, plus what the step's
wall is made of: time covered by kernels, by copies only, by nothing.  Usage: python profiles/timeline.py trace_results.db"""
import sqlite3
import sys


def short(n):
    return n.split("(")[0].replace("void ", "").replace("afq::", "")


def main(path):
    db = sqlite3.connect(path)
    ev = []
    for name, s, e, q, st in db.execute("select name,start,end,queue_id,stream_id from kernels"):
        ev.append((s, e, "K", short(name), f"q{q}/s{st}"))
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    if "memory_copies" in tabs:
        cols = [r[1] for r in db.execute("pragma table_info(memory_copies)")]
        szc = "size" if "size" in cols else None
        q = f"select name,start,end,{szc or '0'} from memory_copies"
        for name, s, e, sz in db.execute(q):
            ev.append((s, e, "C", f"{name} {sz / 1e6:.2f} MB", ""))
    ev.sort()
    # the last step: from the last k_gather_headers / first kernel after the largest idle gap near the end
    starts = [i for i, x in enumerate(ev) if x[3].startswith("k_gather_headers") or x[3].startswith("k_atac_parse")]
    i0 = starts[-1] if starts else 0
    # (a step's planning precedes its first kernel by host time only; its uploads come right before)
    while i0 > 0 and ev[i0 - 1][2] == "C" and "HOST_TO_DEVICE" in ev[i0 - 1][3] and ev[i0][0] - ev[i0 - 1][1] < 300000:
        i0 -= 1
    step = ev[i0:]
    t0 = step[0][0]
    print(f"# last step: {len(step)} events, wall {(max(x[1] for x in step) - t0) / 1e3:.1f} us")
    print(f"{'start_us':>10s} {'dur_us':>9s} {'what':1s} {'queue':8s} name")
    for s, e, k, n, q in step:
        print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:9.1f} {k} {q:8s} {n}")
    # coverage
    pts = sorted(set([x[0] for x in step] + [x[1] for x in step]))
    kern = cop = idle = 0
    for a, b in zip(pts, pts[1:]):
        hk = any(x[0] <= a and x[1] >= b and x[2] == "K" for x in step)
        hc = any(x[0] <= a and x[1] >= b and x[2] == "C" for x in step)
        if hk:
            kern += b - a
        elif hc:
            cop += b - a
        else:
            idle += b - a
    print(f"# covered by kernels {kern / 1e3:.1f} us, by copies only {cop / 1e3:.1f} us, by nothing {idle / 1e3:.1f} us")
    tot = {}
    for s, e, k, n, q in step:
        key = n.split(" ")[0] if k == "C" else n.split("<")[0]
        a = tot.setdefault((k, key), [0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e3
    for (k, n), (c, d) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print(f"# {k} {n:40s} x{c:4d} {d:10.1f} us")


if __name__ == "__main__":
    main(sys.argv[1])
