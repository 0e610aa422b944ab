#!/usr/bin/env python
"""Write a self-contained directory with which anyone holding an `alevin-fry` 0.18 binary can pin this repository's
oracle (and with it the device path) against the reference itself, in one command.  CPU only; no GPU needed.

The reference cannot be built in this project's image (Rust; no cargo, crates not vendored), so the oracle is pinned by the
reference's own structural known-answers and by hand-derived vectors only.  This script closes the loop
from the other side:

    python tests/make_reference_pin.py make  PINDIR [--resolution cr-like] [--usa] [--seed 61]
        PINDIR/in/                 generate_permit_list.json, collate.json, map.collated.rad (written by this repo's RAD writer:
                                   this also puts the writer in front of libradicl's reader), t2g.tsv
        PINDIR/expected_counts.tsv barcode <tab> gene <tab> count, from the oracle (reference arithmetic), count > 0 only
        PINDIR/RUN.sh              the alevin-fry command line

    sh PINDIR/RUN.sh              (on a box with alevin-fry 0.18: writes PINDIR/ref_out)

    python tests/make_reference_pin.py compare PINDIR
        joins PINDIR/ref_out/alevin/quants_mat.{mtx,_rows.txt,_cols.txt} with expected_counts.tsv on (barcode, gene) - the
        method of the reference's scripts/testing/compare_counts.py, without pyroe - and reports: entries only on one side,
        entries that differ at all, entries beyond 1e-4 relative.  cr-like and trivial must agree exactly; for the parsimony
        resolutions see DESIGN.md section 5 (the cover's tie-break is a HashSet walk in the reference: unpinnable), for the
        -em resolutions the f32 summation order (a HashMap walk: 1e-4 is the bar).
"""
import argparse
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

NT = "ACGT"


def bc_string(code, length):
    """2-bit code, first base most significant (convert.rs:75-90)."""
    return "".join(NT[(int(code) >> (2 * (length - 1 - i))) & 3] for i in range(length))


def make(args):
    pkg = importlib.import_module("alevin-fry_amd")
    import oracle as ora

    s = pkg.synth.synth(args.seed, [4000, 1500, 600, 260, 251, 250, 120, 100, 99, 60, 7], num_genes=150, txp_per_gene=3, usa=args.usa,
                        dup=0.5, cross=0.3, umi_err=0.02, max_extra_na=4)
    b, off = s.encode()
    names = [f"T{t}" for t in range(len(s.tid_to_gid))]
    if s.usa:
        rows = [(names[t], f"G{g >> 1}", "S" if g % 2 == 0 else "U") for t, g in enumerate(s.tid_to_gid.tolist())]
    else:
        rows = [(names[t], f"G{g}") for t, g in enumerate(s.tid_to_gid.tolist())]
    os.makedirs(args.dir, exist_ok=True)
    tg = pkg.rad.write_quant_input_dir(os.path.join(args.dir, "in"), np.asarray(b).tobytes(), len(off), names, rows, cblen=16, ulen=s.umi_len)
    cfg = pkg.WorkerConfig.for_resolution(args.resolution, usa_mode=s.usa, num_genes=s.num_genes, num_rows=s.num_rows)
    res = ora.quant(cfg, s.tid_to_gid, b, off)   # the reference's arithmetic (em_arith="reference")
    if s.usa:   # quants_mat_cols.txt in USA mode: names, then name-U, then name-A (quant.rs:1791-1809)
        G = s.num_rows // 3
        col = [f"G{g}" for g in range(G)] + [f"G{g}-U" for g in range(G)] + [f"G{g}-A" for g in range(G)]
    else:
        col = [f"G{g}" for g in range(s.num_genes)]
    n = 0
    with open(os.path.join(args.dir, "expected_counts.tsv"), "w") as f:
        for i in range(res.n_cells):
            g, v = res.row(i)
            bc = bc_string(res.bc[i], 16)
            for gi, vi in zip(g.tolist(), v.tolist()):
                f.write(f"{bc}\t{col[gi]}\t{vi!r}\n")
                n += 1
    with open(os.path.join(args.dir, "RUN.sh"), "w") as f:
        f.write("#!/bin/sh\n# alevin-fry 0.18.x; run from anywhere\n"
                f"D=$(dirname \"$0\")\n${{ALEVIN_FRY_BIN:-alevin-fry}} quant -i \"$D/in\" -m \"$D/in/{os.path.basename(tg)}\" -o \"$D/ref_out\" "
                f"-r {args.resolution} -t 4 --use-mtx\n")
    print(f"{args.dir}: {res.n_cells} cells, {n} expected (barcode, gene) entries, resolution {args.resolution}, usa={bool(s.usa)}; now run RUN.sh where alevin-fry is")


def load_mtx_dir(d):
    rows = open(os.path.join(d, "quants_mat_rows.txt")).read().split()
    cols = open(os.path.join(d, "quants_mat_cols.txt")).read().split()
    body = [l for l in open(os.path.join(d, "quants_mat.mtx")).read().splitlines() if l and not l.startswith("%")]
    return {(rows[int(r) - 1], cols[int(c) - 1]): float(v) for r, c, v in (l.split() for l in body[1:])}


def compare(args):
    want = {}
    for l in open(os.path.join(args.dir, "expected_counts.tsv")):
        bc, g, v = l.rstrip("\n").split("\t")
        want[(bc, g)] = float(v)
    got = load_mtx_dir(os.path.join(args.dir, args.out, "alevin"))
    keys = set(want) | set(got)
    only_ref = sum(1 for k in keys if k not in want)
    only_ours = sum(1 for k in keys if k not in got)
    differ = sum(1 for k in keys if want.get(k, 0.0) != got.get(k, 0.0))
    beyond = sum(1 for k in keys if abs(want.get(k, 0.0) - got.get(k, 0.0)) > 1e-4 * max(want.get(k, 0.0), got.get(k, 0.0)))
    print(f"entries {len(keys)}: only in the reference's output {only_ref}, only expected {only_ours}, differing {differ}, beyond 1e-4 relative {beyond}")
    return 0 if beyond == 0 else 1


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)
    m = sub.add_parser("make")
    m.add_argument("dir")
    m.add_argument("--resolution", default="cr-like")
    m.add_argument("--usa", action="store_true")
    m.add_argument("--seed", type=int, default=61)
    c = sub.add_parser("compare")
    c.add_argument("dir")
    c.add_argument("--out", default="ref_out", help="the alevin-fry output directory inside PINDIR (default ref_out)")
    a = ap.parse_args()
    sys.exit(make(a) if a.cmd == "make" else compare(a))
