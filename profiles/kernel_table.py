#!/usr/bin/env python
"""Markdown table of a leg's kernels from the committed evidence: profiles/<tag>_rocprof.txt (rocprofv3 --kernel-trace --stats of
`bench.py --steps 3 --warmup 1`: four steps) and profiles/rNN_resource_usage.txt of the same round (the tag's first three letters).
Usage: python profiles/kernel_table.py r06_configs2 [min_pct]"""
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def resources(tag):
    out = {}
    p = os.path.join(HERE, tag[:3] + "_resource_usage.txt")
    if not os.path.exists(p):
        return out
    for ln in open(p):
        if "|" not in ln:
            continue
        name, rest = ln.split("|", 1)
        g = lambda k: (re.search(k + r": (\d+)", rest) or [None, "?"])[1]
        out[name.strip()] = (g("VGPRs"), g("VGPRs Spill"), g("SGPRs Spill"), g("LDS Size"), g("Occupancy"))
    return out


def main(tag, min_pct=0.8):
    res = resources(tag)
    rows = []
    for ln in open(os.path.join(HERE, f"{tag}_rocprof.txt")):
        if ln.startswith("#") or ln.startswith("kernel") or not ln.strip():
            if rows:
                break
            continue
        m = re.match(r"(.+?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", ln)
        if not m:
            continue
        name, calls, tot, avg, pct = m.group(1).strip(), int(m.group(2)), float(m.group(3)), float(m.group(4)), float(m.group(5))
        if pct < min_pct or name.startswith("k_synth"):
            continue
        r = res.get(name, ("?",) * 5)
        lds = f"{int(r[3]) / 1024:.1f} KiB" if r[3] != "?" else "?"
        rows.append(f"| `{name}` | {calls // 4 if calls % 4 == 0 else calls / 4:g} | {avg:.0f} | {tot / 4e3:.2f} | {pct:.1f} | {r[0]} ({r[1]} / {r[2]}) | {lds} | {r[4]} |")
    print("| kernel | launches / step | us / launch | ms / step | % of GPU time | VGPRs (spilled V / S) | LDS / workgroup | waves / SIMD |")
    print("|---|---|---|---|---|---|---|---|")
    print("\n".join(rows))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.8)
