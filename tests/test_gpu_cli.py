"""GPU end-to-end test of the drop-in surface: `afquant quant` (same flags and directory protocol as
`alevin-fry quant`) on a synthetic collated-RAD directory; outputs are joined on (barcode string, gene name)
exactly as scripts/testing/compare_counts.py of the reference does, against the oracle's rows."""
import json
import os
import subprocess

import numpy as np
import pytest

from util import ROOT, cfg_for, pkg

pytestmark = pytest.mark.gpu
rad = pkg.rad
synth = pkg.synth
CLI = os.path.join(ROOT, "alevin-fry_amd", "csrc", "afquant")


def make_dir(tmp, s, compressed):
    b, off = s.encode()
    G = s.num_rows // 3 if s.usa else s.num_genes
    tpg = (len(s.tid_to_gid) - (G if s.usa else 0)) // G
    names = [f"T{t}" for t in range(len(s.tid_to_gid))]
    rows = []
    for t, gid in enumerate(s.tid_to_gid.tolist()):
        if s.usa:
            rows.append((names[t], f"G{gid >> 1}", "S" if gid % 2 == 0 else "U"))
        else:
            rows.append((names[t], f"G{gid}"))
    tg = rad.write_quant_input_dir(str(tmp), np.asarray(b).tobytes(), len(off), names, rows, cblen=16, ulen=s.umi_len,
                                   compressed=compressed)
    return tg, b, off


def read_outputs(out):
    rows = open(os.path.join(out, "alevin", "quants_mat_rows.txt")).read().split()
    cols = open(os.path.join(out, "alevin", "quants_mat_cols.txt")).read().split()
    lines = open(os.path.join(out, "alevin", "quants_mat.mtx")).read().splitlines()
    assert lines[0].startswith("%%MatrixMarket matrix coordinate real general")
    body = [l for l in lines if not l.startswith("%")]
    nr, nc, nnz = map(int, body[0].split())
    trip = {}
    for l in body[1:]:
        r, c, v = l.split()
        trip[(rows[int(r) - 1], cols[int(c) - 1])] = float(v)
    assert len(trip) == nnz and nr == len(rows) and nc == len(cols)
    feat = [l.split("\t") for l in open(os.path.join(out, "featureDump.txt")).read().splitlines()]
    meta = json.load(open(os.path.join(out, "quant.json")))
    return rows, cols, trip, feat, meta


@pytest.mark.parametrize("res,usa,compressed,sa", [("cr-like", False, False, None), ("parsimony-em", True, True, None),
                                                  ("cr-like-em", True, False, None), ("cr-like", True, False, "prefer-ambig"),
                                                  ("cr-like", False, True, "prefer-ambig")])
def test_afquant_cli_matches_oracle(tmp_path, oracle, res, usa, compressed, sa):
    s = synth.synth(51, [4000, 1500, 600, 260, 120, 60, 7], num_genes=150, txp_per_gene=3, usa=usa, dup=0.5, cross=0.3, umi_err=0.02)
    tg, b, off = make_dir(tmp_path / "in", s, compressed)
    out = str(tmp_path / "out")
    # unmapped reads per corrected barcode (count:u64, then key:u64 value:u32 pairs): cells 0, 2 and a barcode not in the file
    unm = {int(s.cell_bc[0]): 1234, int(s.cell_bc[2]): 7, 0xFFFFFFF0: 99} if res == "cr-like" else {}
    if unm:
        with open(tmp_path / "in" / "unmapped_bc_count_collated.bin", "wb") as f:
            f.write(len(unm).to_bytes(8, "little") + b"".join(k.to_bytes(8, "little") + v.to_bytes(4, "little") for k, v in unm.items()))
    elif res == "cr-like-em":
        (tmp_path / "in" / "unmapped_bc_count_collated.bin").write_bytes(b"\x01\x02\x03")  # unreadable = no unmapped reads
    r = subprocess.run([CLI, "quant", "-i", str(tmp_path / "in"), "-m", tg, "-o", out, "-r", res, "-t", "4"] + (["--sa-model", sa] if sa else []),
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rows, cols, trip, feat, meta = read_outputs(out)
    want = oracle.quant(cfg_for(s, res, **({"sa_model": sa} if sa else {})), s.tid_to_gid, b, off)
    if sa:  # the hidden switch only lives in USA mode (quant.rs:1456-1469); it also turns the tiny-cell path off there
        assert ("SplicedAmbiguityModel will be ignored" in r.stderr) == (not usa)
        assert meta["quant_options"]["sa_model"] == ("PreferAmbiguity" if usa else "WinnerTakeAll")
        assert (meta["num_tiny_cell_resolved"] == 0) == usa
    G = s.num_rows // 3 if usa else s.num_genes
    gname = [f"G{i}" for i in range(G)]
    colname = gname + [g + "-U" for g in gname] + [g + "-A" for g in gname] if usa else gname
    assert cols == colname
    exp = {}
    for i in range(want.n_cells):
        bcs = rad.int_to_seq(int(want.bc[i]), 16)
        g, v = want.row(i)
        for a, x in zip(g.tolist(), v.tolist()):
            exp[(bcs, colname[a])] = x
    assert set(trip) == set(exp)
    for k in exp:  # MTX values are the shortest round-trip text of the f32: parse back to the same f32
        assert np.float32(trip[k]) == np.float32(exp[k]), k
    assert rows == [rad.int_to_seq(int(x), 16) for x in want.bc]
    # featureDump: header, one row per cell, the reference's columns (src/quant.rs:1609-1612, 1248-1260)
    assert feat[0] == ["CB", "CorrectedReads", "MappedReads", "DeduplicatedReads", "MappingRate", "DedupRate", "MeanByMax", "NumGenesExpressed", "NumGenesOverMean"]
    for i in range(want.n_cells):
        st = want.cell_stats(i)
        f = feat[1 + i]
        nu = unm.get(int(want.bc[i]), 0)
        assert f[0] == rows[i] and int(f[1]) == int(want.nrec[i]) + nu and int(f[2]) == int(want.nrec[i])
        assert int(f[7]) == st["num_expr"] and int(f[8]) == st["num_genes_over_mean"]
        assert np.float32(float(f[3])) == np.float32(st["sum_umi"])
        assert np.float32(float(f[4])) == np.float32(want.nrec[i]) / np.float32(int(want.nrec[i]) + nu)   # quant.rs:1185-1187
        assert np.float32(float(f[5])) == np.float32(st["dedup_rate"])
    assert meta["resolution_strategy"] in ("CellRangerLike", "ParsimonyEm", "CellRangerLikeEm") and meta["usa_mode"] == usa
    assert meta["num_quantified_cells"] == want.n_cells and meta["num_genes"] == s.num_rows
    assert meta["tiny_cell_resolved_cell_numbers"] == [i for i in range(want.n_cells) if want.flags[i] & 1]
    assert meta["empty_resolved_cell_numbers"] == [i for i in range(want.n_cells) if want.flags[i] & 4]


@pytest.mark.parametrize("res,usa", [("cr-like", True), ("parsimony-em", False), ("parsimony", True)])
def test_afquant_cli_dump_eqclasses(tmp_path, oracle, res, usa):
    """-d (write_eqc_counts, quant.rs:229-355): geqc_counts.mtx (cells x classes) + gene_eqclass.txt.gz; class ids are
    arbitrary in the reference (hash + completion order), so the comparison is on (cell, gene set, count)."""
    import gzip

    s = synth.synth(53, [3000, 800, 260, 120, 60, 7], num_genes=120, txp_per_gene=2, usa=usa, dup=0.5, cross=0.3, umi_err=0.02)
    tg, b, off = make_dir(tmp_path / "in", s, False)
    out = str(tmp_path / "out")
    r = subprocess.run([CLI, "quant", "-i", str(tmp_path / "in"), "-m", tg, "-o", out, "-r", res, "-d", "-t", "2"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rows, cols, trip, feat, meta = read_outputs(out)
    want = oracle.quant(cfg_for(s, res, dump_eq=True), s.tid_to_gid, b, off)
    for k, v in trip.items():   # the count matrix is the one without -d
        pass
    assert meta["dump_eq"] is True and meta["quant_options"]["dump_eq"] is True
    lines = gzip.open(os.path.join(out, "alevin", "gene_eqclass.txt.gz"), "rt").read().split("\n")
    assert int(lines[0]) == s.num_rows
    n_cls = int(lines[1])
    cls = {}
    for ln in lines[2:2 + n_cls]:
        t = ln.split("\t")
        cls[int(t[-1])] = tuple(int(x) for x in t[:-1])
    assert sorted(cls) == list(range(n_cls)) and len(set(cls.values())) == n_cls and lines[2 + n_cls:] == [""]
    with open(os.path.join(out, "alevin", "geqc_counts.mtx")) as f:
        hdr = f.readline(); f.readline()
        nr, nc, nz = (int(x) for x in f.readline().split())
        ent = [ln.split() for ln in f.read().splitlines()]
    assert hdr.startswith("%%MatrixMarket matrix coordinate real general") and (nr, nc, nz) == (want.n_cells, n_cls, len(ent))
    got = {}
    for rr, cc, vv in ent:
        got.setdefault(int(rr) - 1, []).append((cls[int(cc) - 1], int(float(vv))))
    uo = s.num_rows // 3
    def out_label(lab):   # USA: gene ids -> S/U/A columns as the writer prints them (quant.rs:284-335)
        if not usa:
            return tuple(lab)
        o, k = [], 0
        while k < len(lab):
            g = lab[k]
            if k + 1 < len(lab) and lab[k + 1] >> 1 == g >> 1:
                o.append((g >> 1) + 2 * uo); k += 2
            else:
                o.append((g >> 1) + uo if g & 1 else g >> 1); k += 1
        return tuple(o)
    for i in range(want.n_cells):
        exp = sorted((out_label(lab), c) for lab, c in want.eqclasses.cell(i))
        assert sorted(got.get(i, [])) == exp, (res, i)


@pytest.mark.parametrize("summary_stat", [True, False])
def test_afquant_cli_bootstraps(tmp_path, oracle, summary_stat):
    """-b N [--summary-stat] (quant.rs:127-210, 1850-1877): alevin/bootstraps_mean.mtx and bootstraps_var.mtx, cells x columns,
    non-zero entries only; the draws are reproducible here (--boot-seed), so the files equal the oracle's numbers."""
    s = synth.synth(54, [2500, 700, 150, 40], num_genes=90, txp_per_gene=2, usa=False, dup=0.5, cross=0.35, umi_err=0.02)
    tg, b, off = make_dir(tmp_path / "in", s, False)
    out = str(tmp_path / "out")
    cmd = [CLI, "quant", "-i", str(tmp_path / "in"), "-m", tg, "-o", out, "-r", "cr-like-em", "-b", "9", "--boot-seed", "77"] + (["--summary-stat"] if summary_stat else [])
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert ("Full per-replicate bootstrap output is not yet supported" in r.stderr) == (not summary_stat)
    rows, cols, trip, feat, meta = read_outputs(out)
    assert meta["quant_options"]["num_bootstraps"] == 9 and meta["quant_options"]["summary_stat"] is summary_stat
    want = oracle.quant(cfg_for(s, "cr-like-em", num_bootstraps=9, summary_stat=summary_stat, boot_seed=77), s.tid_to_gid, b, off)

    def read_mtx(name):
        with open(os.path.join(out, "alevin", name)) as f:
            assert f.readline().startswith("%%MatrixMarket matrix coordinate real general")
            f.readline()
            nr, nc, nz = (int(x) for x in f.readline().split())
            ent = {(int(a) - 1, int(c) - 1): np.float32(v) for a, c, v in (ln.split() for ln in f.read().splitlines())}
        assert (nr, nc, nz) == (want.n_cells, s.num_rows, len(ent))
        return ent

    for name, getter in (("bootstraps_mean.mtx", want.bootstraps.mean), ("bootstraps_var.mtx", want.bootstraps.var)):
        ent = read_mtx(name)
        exp = {}
        for i in range(want.n_cells):
            c, v = getter(i)
            for a, x in zip(c.tolist(), v.tolist()):
                exp[(i, a)] = np.float32(x)
        assert ent == exp, name
    assert not any(i == 3 for i, _ in read_mtx("bootstraps_mean.mtx"))   # the 40-read cell took the tiny path: no bootstraps
    # -b is for the -em resolutions (main.rs:713-728); --summary-stat needs -b (main.rs:307); -d with trivial is refused (main.rs:705-711)
    for extra, res in ((["-b", "4"], "cr-like"), (["--summary-stat"], "cr-like-em"), (["-d"], "trivial")):
        r = subprocess.run([CLI, "quant", "-i", str(tmp_path / "in"), "-m", tg, "-o", out, "-r", res] + extra, capture_output=True, text=True)
        assert r.returncode != 0


@pytest.mark.parametrize("usa", [False, True])
def test_afquant_cli_infer(tmp_path, oracle, usa):
    """`afquant infer` (src/infer.rs) on the files `afquant quant -d` wrote: per row the oracle's em_optimize_subset
    restatement over the row's classes in column order, bit for bit; rows / cols files carried over; --quant-subset."""
    import gzip

    s = synth.synth(55, [2000, 600, 150, 20], num_genes=80, txp_per_gene=2, usa=usa, dup=0.5, cross=0.4, umi_err=0.02)
    tg, b, off = make_dir(tmp_path / "in", s, False)
    out = str(tmp_path / "out")
    r = subprocess.run([CLI, "quant", "-i", str(tmp_path / "in"), "-m", tg, "-o", out, "-r", "cr-like-em", "-d"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    al = os.path.join(out, "alevin")
    lines = gzip.open(os.path.join(al, "gene_eqclass.txt.gz"), "rt").read().split("\n")
    n_cls = int(lines[1])
    labels = [None] * n_cls
    for ln in lines[2:2 + n_cls]:
        t = [int(x) for x in ln.split()]
        labels[t[-1]] = t[:-1]
    rows_in = open(os.path.join(al, "quants_mat_rows.txt")).read().split()
    cells = {i: [] for i in range(len(rows_in))}
    with open(os.path.join(al, "geqc_counts.mtx")) as f:
        f.readline(); f.readline(); f.readline()
        for ln in f.read().splitlines():
            a, c, v = ln.split()
            cells[int(a) - 1].append((int(c) - 1, int(round(float(v)))))
    keep = [0, 2]
    sub = tmp_path / "subset.txt"
    sub.write_text("\n".join(rows_in[i] for i in keep) + "\n")
    for extra, sel in (([], list(range(len(rows_in)))), (["--quant-subset", str(sub)], keep)):
        out2 = str(tmp_path / ("inf" + str(len(sel))))
        r = subprocess.run([CLI, "infer", "-c", os.path.join(al, "geqc_counts.mtx"), "-e", os.path.join(al, "gene_eqclass.txt.gz"), "-o", out2, "-t", "2"]
                           + (["--usa"] if usa else []) + extra, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert open(os.path.join(out2, "quants_mat_rows.txt")).read().split() == [rows_in[i] for i in sel]
        assert open(os.path.join(out2, "quants_mat_cols.txt")).read() == open(os.path.join(al, "quants_mat_cols.txt")).read()
        with open(os.path.join(out2, "quants_mat.mtx")) as f:
            assert f.readline().startswith("%%MatrixMarket matrix coordinate real general")
            f.readline()
            nr, nc, nz = (int(x) for x in f.readline().split())
            ent = {(int(a) - 1, int(c) - 1): np.float32(v) for a, c, v in (ln.split() for ln in f.read().splitlines())}
        assert (nr, nc, nz) == (len(sel), s.num_rows, len(ent))
        exp = {}
        for ri, i in enumerate(sel):
            row = sorted(cells[i])
            if not row:
                continue
            alphas, _ = oracle.em([labels[e] for e, _ in row], [c for _, c in row], s.num_rows,
                                  usa_offsets=(s.num_rows // 3, 2 * s.num_rows // 3) if usa else None, dense=0)
            for c in np.flatnonzero(alphas > 0):
                exp[(ri, int(c))] = np.float32(alphas[c])
        assert ent == exp


def test_afquant_cli_quant_subset_and_flag_errors(tmp_path, oracle):
    s = synth.synth(52, [900, 500, 300, 100], num_genes=80, dup=0.4)
    tg, b, off = make_dir(tmp_path / "in", s, False)
    keep = [1, 3]
    sub = tmp_path / "subset.txt"
    sub.write_text("\n".join(rad.int_to_seq(int(s.cell_bc[i]), 16) for i in keep) + "\n")
    out = str(tmp_path / "out")
    r = subprocess.run([CLI, "quant", "-i", str(tmp_path / "in"), "-m", tg, "-o", out, "-r", "cr-like", "--quant-subset", str(sub)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rows, cols, trip, feat, meta = read_outputs(out)
    assert rows == [rad.int_to_seq(int(s.cell_bc[i]), 16) for i in keep] and meta["num_quantified_cells"] == 2
    want = oracle.quant(cfg_for(s, "cr-like"), s.tid_to_gid, b, off[keep])
    assert abs(sum(trip.values()) - float(want.val.sum())) < 1e-3
    # cr-like does not take --umi-edit-dist 1 (src/main.rs:674-688)
    for extra in (["--umi-edit-dist", "1"],):
        r = subprocess.run([CLI, "quant", "-i", str(tmp_path / "in"), "-m", tg, "-o", out, "-r", "cr-like"] + extra, capture_output=True, text=True)
        assert r.returncode != 0 and "afquant quant failed" in r.stderr
    os.remove(tmp_path / "in" / "generate_permit_list.json")
    r = subprocess.run([CLI, "quant", "-i", str(tmp_path / "in"), "-m", tg, "-o", out, "-r", "cr-like"], capture_output=True, text=True)
    assert r.returncode != 0 and "generate_permit_list.json" in r.stderr


@pytest.mark.parametrize("usa", [False, True])
def test_reference_pin_directory_against_our_own_cli(tmp_path, usa):
    """tests/make_reference_pin.py writes the directory with which an alevin-fry 0.18 binary pins the oracle; here `afquant quant`
    stands where alevin-fry would: its output files are joined with the expected counts by the script's own `compare`
    (barcode strings, USA column names, MatrixMarket indices all have to line up)."""
    import subprocess
    import sys

    script = os.path.join(ROOT, "tests", "make_reference_pin.py")
    d = tmp_path / "pin"
    r = subprocess.run([sys.executable, script, "make", str(d)] + (["--usa"] if usa else []), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([CLI, "quant", "-i", str(d / "in"), "-m", str(d / "in" / "t2g.tsv"), "-o", str(d / "ref_out"), "-r", "cr-like", "-t", "4"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([sys.executable, script, "compare", str(d)], capture_output=True, text=True)
    assert r.returncode == 0 and "differing 0," in r.stdout and "only expected 0," in r.stdout, r.stdout + r.stderr
