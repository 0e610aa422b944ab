"""GPU parity tests (bit-exact): HIP cr-like path through the C ABI vs the CPU oracle."""
import numpy as np
import pytest

from util import assert_same_result, cfg_for, load_golden, pkg, rows_of

pytestmark = pytest.mark.gpu
rad = pkg.rad
synth = pkg.synth


def run_both(oracle, cfg, t2g, b, off):
    q = pkg.Quantifier(cfg, t2g)
    try:
        got = q.quant_chunks(b, off)
        st = q.batch_stats()
    finally:
        q.close()
    want = oracle.quant(cfg, t2g, b, off)
    return got, want, st


@pytest.mark.parametrize("small_thresh", [0, 100])
def test_hand_cases(oracle, small_thresh):
    for case in load_golden("crlike_hand_cases.json")["cases"]:
        cells = [(c["bc"], [(u, r) for u, r in c["reads"]]) for c in case["cells"]]
        b, off = rad.encode_cells(cells, 4, 4)
        cfg = pkg.WorkerConfig.for_resolution("cr-like", usa_mode=case["usa"], num_genes=case["num_genes"],
                                              num_rows=case["num_rows"], small_thresh=small_thresh)
        got, want, _ = run_both(oracle, cfg, np.asarray(case["t2g"], np.uint32), b, off)
        for c, g in zip(case["cells"], rows_of(got)):
            assert [[int(a), int(v)] for a, v in g] == c["expected"], (case["name"], c["bc"], c["why"])
        assert_same_result(got, want, what=case["name"])


def test_prefer_ambig_hand_cases(oracle):
    """Hidden `--sa-model prefer-ambig` (pugutils.rs:505-641) against the hand-derived counts and the oracle."""
    for case in load_golden("prefer_ambig_hand_cases.json")["cases"]:
        cells = [(c["bc"], [(u, r) for u, r in c["reads"]]) for c in case["cells"]]
        b, off = rad.encode_cells(cells, 4, 4)
        t2g = np.asarray(case["t2g"], np.uint32)
        kw = dict(usa_mode=case["usa"], num_genes=case["num_genes"], num_rows=case["num_rows"])
        got, want, _ = run_both(oracle, pkg.WorkerConfig.for_resolution("cr-like", sa_model="prefer-ambig", **kw), t2g, b, off)
        for c, g in zip(case["cells"], rows_of(got)):
            assert [[int(a), int(v)] for a, v in g] == c["expected"], (case["name"], c["bc"], c["why"])
        assert_same_result(got, want, what=case["name"])
        got, want, _ = run_both(oracle, pkg.WorkerConfig.for_resolution("cr-like", **kw), t2g, b, off)
        for c, g in zip(case["cells"], rows_of(got)):
            assert [[int(a), int(v)] for a, v in g] == c["expected_wta"], (case["name"], c["bc"], "winner-take-all")


@pytest.mark.parametrize("resolution", ["cr-like", "cr-like-em", "parsimony", "trivial"])
def test_prefer_ambig_synthetic(oracle, resolution):
    """prefer-ambig on USA data with S/U-ambiguous molecules: tiny cells lose their fast path (quant.rs:794) and go
    through -r's own strategy; cells from a few reads to multi-bucket sizes; result differs from winner-take-all."""
    s = synth.synth(23, [5, 40, 99, 100, 250, 251, 900, 5000, 60000], num_genes=300, usa=True, dup=0.6, max_extra_na=6)
    b, off = s.encode()
    got, want, _ = run_both(oracle, cfg_for(s, resolution, sa_model="prefer-ambig"), s.tid_to_gid, b, off)
    assert_same_result(got, want, what=resolution)
    assert not (got.flags & pkg._abi.CELL_TINY_PATH).any()
    if resolution.startswith("cr-like"):
        wta, _, _ = run_both(oracle, cfg_for(s, resolution), s.tid_to_gid, b, off)
        assert not (np.array_equal(wta.gene, got.gene) and np.array_equal(wta.val, got.val))
    # outside USA mode the switch is ignored (quant.rs:1456-1469)
    s2 = synth.synth(24, [50, 3000], num_genes=100)
    b2, off2 = s2.encode()
    got2, want2, _ = run_both(oracle, cfg_for(s2, "cr-like", sa_model="prefer-ambig"), s2.tid_to_gid, b2, off2)
    assert_same_result(got2, want2)
    assert bool(got2.flags[0] & pkg._abi.CELL_TINY_PATH)


@pytest.mark.parametrize("usa", [False, True])
@pytest.mark.parametrize("resolution", ["cr-like-em", "parsimony-em", "parsimony-gene-em"])
def test_dump_eqclasses(oracle, resolution, usa):
    """cfg.dump_eq (-d, quant.rs:1282-1307): per cell the same set of (gene-level label, molecules) as the oracle's
    gene_eqc; tiny-path cells report none; the counts are untouched by the flag."""
    s = synth.synth(31, [3, 60, 99, 100, 400, 3000, 30000], num_genes=200, usa=usa, dup=0.5, cross=0.3, max_extra_na=5, umi_err=0.02)
    b, off = s.encode()
    got, want, _ = run_both(oracle, cfg_for(s, resolution, dump_eq=True), s.tid_to_gid, b, off)
    assert_same_result(got, want, what=resolution)
    plain, _, _ = run_both(oracle, cfg_for(s, resolution), s.tid_to_gid, b, off)
    assert_same_result(got, plain, what="dump_eq changes nothing")
    assert not hasattr(plain, "eqclasses")
    n_cls = 0
    for i in range(got.n_cells):
        g, w = got.eqclasses.cell(i), want.eqclasses.cell(i)
        assert g == w, (resolution, usa, i, len(g), len(w))
        assert (len(g) == 0) == bool(got.flags[i] & pkg._abi.CELL_TINY_PATH)
        n_cls += len(g)
    assert n_cls > 100 and any(len(lab) > 1 for lab, _ in got.eqclasses.cell(6))
    # the plain resolutions do not keep the classes on the device: refused (the host front-end runs the -em sibling)
    q = pkg.Quantifier(cfg_for(s, resolution[:-3], dump_eq=True), s.tid_to_gid)
    try:
        with pytest.raises(pkg.AfqError) as e:
            q.quant_chunks(b, off)
        assert e.value.code == pkg._abi.AFQ_ERR_UNSUPPORTED
    finally:
        q.close()


@pytest.mark.parametrize("summary_stat", [True, False])
@pytest.mark.parametrize("resolution,usa", [("cr-like-em", False), ("cr-like-em", True), ("parsimony-em", True), ("parsimony-gene-em", False)])
def test_bootstraps(oracle, resolution, usa, summary_stat):
    """-b (em.rs:585-690, multinomial.rs, quant.rs:157-210): per non-tiny cell the mean / variance over the replicates.
    The reference's generator is unseeded; with the draws restated on Philox (same streams both sides) the device
    equals the oracle bit for bit, and the summaries behave like a bootstrap (means near the point estimate, total
    mass = the cell's molecules)."""
    s = synth.synth(37, [5, 99, 100, 700, 9000, 40000], num_genes=300, usa=usa, dup=0.5, cross=0.3, max_extra_na=5, umi_err=0.02)
    b, off = s.encode()
    cfg = cfg_for(s, resolution, num_bootstraps=12, summary_stat=summary_stat, boot_seed=0xC0FFEE1234)
    q = pkg.Quantifier(cfg, s.tid_to_gid)
    try:
        got = q.quant_chunks(b, off, first_cell_index=1000)
        # the same cells handed over as two batches: the draws hang off the cell index, not the batch
        g1 = q.quant_chunks(b, off[:3], first_cell_index=1000)
        g2 = q.quant_chunks(b, off[3:], first_cell_index=1003)
    finally:
        q.close()
    want = oracle.quant(cfg, s.tid_to_gid, b, off, first_cell_index=1000)
    assert_same_result(got, want, what=resolution)
    gb, wb = got.bootstraps, want.bootstraps
    for name in ("mean_ptr", "mean_col", "var_ptr", "var_col"):
        assert np.array_equal(getattr(gb, name), getattr(wb, name)), name
    assert np.array_equal(gb.mean_val.view(np.uint32), wb.mean_val.view(np.uint32))
    assert np.array_equal(gb.var_val.view(np.uint32), wb.var_val.view(np.uint32))
    for i in range(got.n_cells):
        part, j = (g1, i) if i < 3 else (g2, i - 3)
        assert np.array_equal(part.bootstraps.mean(j)[1], gb.mean(i)[1]) and np.array_equal(part.bootstraps.var(j)[0], gb.var(i)[0])
        mc, mv = gb.mean(i)
        if got.flags[i] & pkg._abi.CELL_TINY_PATH:
            assert len(mc) == 0 and len(gb.var(i)[0]) == 0
            continue
        assert len(mc) > 0 and np.all(np.diff(mc.astype(np.int64)) > 0)
        # every replicate redistributes the cell's molecules: the means add up to the point estimate's total (floored alphas aside)
        _, v = got.row(i)
        assert abs(float(mv.sum()) - float(v.sum())) <= 0.02 * float(v.sum()) + 1.0
    assert float(gb.var_val.min()) >= -1e-3 and float(gb.var_val.max()) > 0
    # another seed: other draws
    w2 = oracle.quant(cfg_for(s, resolution, num_bootstraps=12, summary_stat=summary_stat, boot_seed=1), s.tid_to_gid, b, off, first_cell_index=1000)
    assert not np.array_equal(w2.bootstraps.mean_val, wb.mean_val)


@pytest.mark.parametrize("usa", [False, True])
def test_infer(oracle, usa):
    """`alevin-fry infer` (src/infer.rs): one EM per row of an equivalence-class count matrix, classes in column order,
    informative start, USA offsets when asked.  Input built the way `quant -d` writes it (labels as output columns, class
    ids in order of first appearance); expected = the oracle's em_optimize_subset restatement on the same rows, bit for bit;
    and the abundances land near quant's own cr-like-em estimate of the same cells."""
    s = synth.synth(43, [8, 150, 2500, 30000], num_genes=400, usa=usa, dup=0.5, cross=0.4, max_extra_na=5, umi_err=0.02)
    b, off = s.encode()
    ref = oracle.quant(cfg_for(s, "cr-like-em", small_thresh=0, dump_eq=True), s.tid_to_gid, b, off)
    uo = s.num_rows // 3
    def col_label(lab):   # gene ids -> output columns, as write_eqc_counts prints them (quant.rs:284-335)
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
    ids, cells = {}, []
    for i in range(ref.n_cells):
        row = []
        for lab, cnt in ref.eqclasses.cell(i):
            row.append((ids.setdefault(col_label(lab), len(ids)), cnt))
        cells.append(sorted(row))
    eq_labels = [list(l) for l, _ in sorted(ids.items(), key=lambda kv: kv[1])]
    cells.append([])   # a cell with no classes: an empty row
    q = pkg.Quantifier(cfg_for(s), s.tid_to_gid)
    try:
        got = q.infer(eq_labels, cells, s.num_rows, usa_mode=usa)
    finally:
        q.close()
    assert got.n_cells == len(cells) and len(got.row(len(cells) - 1)[0]) == 0
    for i, row in enumerate(cells[:-1]):
        alphas, _ = oracle.em([eq_labels[e] for e, _ in row], [c for _, c in row], s.num_rows,
                              usa_offsets=(uo, 2 * s.num_rows // 3) if usa else None, dense=0)
        nz = np.flatnonzero(alphas > 0)
        g, v = got.row(i)
        assert np.array_equal(g, nz.astype(np.uint32)), i
        assert np.array_equal(v.view(np.uint32), alphas[nz].astype(np.float32).view(np.uint32)), i
        # near quant's own estimate (another schedule and class order: only entries well above the 0.01 floor are comparable)
        rg, rv = ref.row(i)
        est = dict(zip(rg.tolist(), rv.tolist()))
        big = [(c, x) for c, x in zip(g.tolist(), v.tolist()) if x >= 0.9]
        assert len(big) > 0 and all(abs(est.get(c, 0.0) - x) <= 0.05 * x + 0.1 for c, x in big), i


def test_bootstraps_large_cell(oracle):
    """A cell with more classes and more expressed genes than the bootstrap kernel keeps in LDS (working arrays in its scratch)."""
    s = synth.synth(38, [150000], num_genes=30000, usa=False, dup=0.3, cross=0.5, max_extra_na=6, zipf=0.3)
    b, off = s.encode()
    cfg = cfg_for(s, "cr-like-em", num_bootstraps=3, summary_stat=True, boot_seed=5, dump_eq=True)
    got, want, _ = run_both(oracle, cfg, s.tid_to_gid, b, off)
    assert_same_result(got, want)
    cls = got.eqclasses.cell(0)
    n_genes = len({g for lab, _ in cls for g in lab})
    assert len(cls) > 11264 and n_genes > 11264, (len(cls), n_genes)
    gb, wb = got.bootstraps, want.bootstraps
    assert np.array_equal(gb.mean_col, wb.mean_col) and np.array_equal(gb.var_col, wb.var_col)
    assert np.array_equal(gb.mean_val.view(np.uint32), wb.mean_val.view(np.uint32))
    assert np.array_equal(gb.var_val.view(np.uint32), wb.var_val.view(np.uint32))


@pytest.mark.parametrize("bw,uw", [(1, 1), (2, 2), (8, 8), (2, 4), (4, 2), (8, 4), (1, 8)])
def test_field_widths(oracle, bw, uw):
    """Unaligned record layouts take the byte-granular walk."""
    case = load_golden("crlike_hand_cases.json")["cases"][0]
    cells = [(c["bc"], [(u, r) for u, r in c["reads"]]) for c in case["cells"]]
    b, off = rad.encode_cells(cells, bw, uw)
    cfg = pkg.WorkerConfig.for_resolution("cr-like", num_genes=4, num_rows=4, bc_bytes=bw, umi_bytes=uw)
    got, want, _ = run_both(oracle, cfg, np.asarray(case["t2g"], np.uint32), b, off)
    assert_same_result(got, want)
    for c, g in zip(case["cells"], rows_of(got)):
        assert [[int(a), int(v)] for a, v in g] == c["expected"]


@pytest.mark.parametrize("usa", [False, True])
def test_config1_plumbing(oracle, usa):
    """BASELINE config 1: 1k cells x 50 reads (all cells take the tiny path in the reference)."""
    s = synth.synth(1, [50] * 1000, num_genes=1000, usa=usa)
    b, off = s.encode()
    got, want, st = run_both(oracle, cfg_for(s), s.tid_to_gid, b, off)
    assert_same_result(got, want)
    assert st["n_records"] == 50_000 and st["n_overflow_buckets"] == 0
    assert (got.flags & pkg._abi.CELL_TINY_PATH).all()


@pytest.mark.parametrize("usa", [False, True])
def test_multi_bucket_cells(oracle, usa):
    """Cells from 1 read to 60k reads: single-bucket LDS finish, multi-bucket dense rows, ragged sizes."""
    sizes = [60000, 33000, 9000, 5000, 2100, 1025, 513, 512, 300, 251, 250, 100, 99, 64, 63, 2, 1]
    s = synth.synth(2, sizes, num_genes=3000, usa=usa, dup=0.45, zipf=0.6, max_extra_na=20)
    b, off = s.encode()
    got, want, st = run_both(oracle, cfg_for(s), s.tid_to_gid, b, off)
    assert_same_result(got, want)
    assert st["n_buckets"] > len(sizes)


@pytest.mark.parametrize("n_same", [1500, 7000, 12000])
def test_overflow_buckets(oracle, n_same):
    """One UMI carried by n_same reads lands in one bucket: 1500 -> over the 2-wave cap (LDS mid path),
    12000 -> beyond LDS reach (in-place sort in global scratch).  Must still resolve exactly."""
    s = synth.synth(3, [n_same + 2000, 400], num_genes=500, dup=0.3)
    umi = s.umi.copy()
    umi[:n_same] = 0x123456
    s.umi = umi
    b, off = s.encode()
    got, want, st = run_both(oracle, cfg_for(s), s.tid_to_gid, b, off)
    assert st["n_overflow_buckets"] >= 1
    assert_same_result(got, want)


def test_degenerate_records(oracle):
    """na == 0 records, reads with > 8 distinct genes, duplicate-heavy UMIs, max-width ref lists."""
    t2g = np.arange(40, dtype=np.uint32)
    cells = [
        (5, [(1, []), (2, [3]), (2, []), (7, list(range(0, 30))), (7, list(range(0, 30))), (7, [4])]),
        (6, [(9, list(range(5, 17))), (9, [5])] + [(100 + i, [i % 40]) for i in range(300)]),
        (7, [(1, [])]),
    ]
    b, off = rad.encode_cells(cells, 4, 4)
    cfg = pkg.WorkerConfig.for_resolution("cr-like", num_genes=40, num_rows=40)
    got, want, _ = run_both(oracle, cfg, t2g, b, off)
    assert_same_result(got, want)
    assert rows_of(got)[0] == [(3, 1.0), (4, 1.0)]  # UMI7: gene4 has 3 votes, genes 0..29 two each
    assert got.flags[2] & pkg._abi.CELL_EMPTY


def test_device_resident_submit_and_order_independence(oracle):
    """afq_submit_device on a torch buffer; shuffling reads inside cells leaves counts unchanged."""
    import torch

    s = synth.synth(4, [3000, 800, 120], num_genes=400, dup=0.4)
    b, off = s.encode()
    cfg = cfg_for(s)
    want = oracle.quant(cfg, s.tid_to_gid, b, off)
    d = torch.from_numpy(np.asarray(b).copy()).to("cuda:0")
    q = pkg.Quantifier(cfg, s.tid_to_gid)
    try:
        q.submit_device(d.data_ptr(), d.numel(), off)
        got = q.collect()
        assert_same_result(got, want)
        # permute the reads of each cell
        rng = np.random.default_rng(0)
        cells = []
        for i in range(len(off)):
            bc, reads = rad.decode_chunk(bytes(b), int(off[i]), 4, 4)
            rng.shuffle(reads)
            cells.append((bc, reads))
        b2, off2 = rad.encode_cells(cells, 4, 4)
        got2 = q.quant_chunks(b2, off2)
        assert_same_result(got2, want)
    finally:
        q.close()


def test_bad_input_is_reported_not_crashed():
    s = synth.synth(5, [200, 200], num_genes=50)
    b, off = s.encode()
    cfg = cfg_for(s)
    q = pkg.Quantifier(cfg, s.tid_to_gid)
    try:
        bad = np.asarray(b).copy()
        bad[8 + int(off[1])] = 77  # first record's na in cell 1 no longer tiles the chunk
        with pytest.raises(pkg.AfqError) as e:
            q.quant_chunks(bad, off)
        assert e.value.code == pkg._abi.AFQ_ERR_BAD_INPUT and "cell 1" in str(e.value)
        # the context stays usable
        q.quant_chunks(b, off)
    finally:
        q.close()
    q = pkg.Quantifier(cfg, s.tid_to_gid[: len(s.tid_to_gid) // 2])  # ref ids now out of range
    try:
        with pytest.raises(pkg.AfqError) as e:
            q.quant_chunks(b, off)
        assert e.value.code == pkg._abi.AFQ_ERR_BAD_INPUT
    finally:
        q.close()


def test_unsupported_requests_fail_loudly():
    """What the device path does not implement is refused with AFQ_ERR_UNSUPPORTED, never approximated."""
    s = synth.synth(6, [50], num_genes=20)
    cells = [(1, [(5, [0])])]
    b, off = rad.encode_cells(cells, 4, 4)
    # -d keeps gene-level classes, which only the -em resolutions hold on the device (afq_quantify runs the sibling)
    q = pkg.Quantifier(pkg.WorkerConfig.for_resolution("parsimony", num_genes=20, num_rows=20, dump_eq=True), s.tid_to_gid)
    try:
        with pytest.raises(pkg.AfqError) as e:
            q.quant_chunks(b, off)
        assert e.value.code == pkg._abi.AFQ_ERR_UNSUPPORTED
    finally:
        q.close()


@pytest.mark.parametrize("usa", [False, True])
@pytest.mark.parametrize("small_thresh", [100, 0])
def test_trivial_resolution(oracle, usa, small_thresh):
    """`trivial` (pugutils.rs:852-911): single-gene reads only, distinct UMIs per gene; cells under
    --small-thresh still take the cr-like tiny path (quant.rs:794-845)."""
    sizes = [30000, 4000, 700, 260, 120, 99, 40, 3]
    s = synth.synth(12, sizes, num_genes=800, usa=usa, dup=0.5, zipf=0.8, max_extra_na=10)
    b, off = s.encode()
    cfg = cfg_for(s, "trivial", small_thresh=small_thresh)
    got, want, _ = run_both(oracle, cfg, s.tid_to_gid, b, off)
    assert_same_result(got, want)
    # trivial really differs from cr-like on this input
    other = oracle.quant(cfg_for(s, "cr-like", small_thresh=small_thresh), s.tid_to_gid, b, off)
    assert not np.array_equal(other.val, want.val) or not np.array_equal(other.gene, want.gene)


def test_atac_dedup(oracle):
    """afq_atac_dedup vs the oracle (atac/deduplicate.rs:199-237): config-5-like fragments, 20 % exact duplicates,
    an empty cell, a cell of one fragment, u16 count wrap-around."""
    rng = np.random.default_rng(5)
    sizes = [3000, 0, 1, 777, 2500, 64, 65, 1024, 1025, 70000]
    refs, starts, lens = [], [], []
    for n in sizes:
        r = rng.integers(0, 25, n).astype(np.uint32)
        st = rng.integers(0, 1_000_000, n).astype(np.uint32)
        ln = np.clip(np.round(np.exp(rng.normal(5.0, 0.6, n))), 30, 2500).astype(np.uint16)
        d = rng.random(n) < 0.2
        src = rng.integers(0, max(n, 1), n)
        r, st, ln = np.where(d, r[src], r), np.where(d, st[src], st), np.where(d, ln[src], ln)
        refs.append(r); starts.append(st); lens.append(ln)
    # last cell: one fragment repeated 66000 times -> count wraps to 66000 & 0xFFFF like `count as u16`
    refs[-1][:66000] = 3; starts[-1][:66000] = 12345; lens[-1][:66000] = 150
    ref, start, flen = np.concatenate(refs), np.concatenate(starts), np.concatenate(lens)
    ptr = np.concatenate(([0], np.cumsum(sizes))).astype(np.uint64)
    want = oracle.atac_dedup(ref, start, flen, ptr)
    q = pkg.Quantifier(pkg.WorkerConfig.for_resolution("cr-like", num_genes=1, num_rows=1), np.zeros(1, np.uint32))
    try:
        got = q.atac_dedup(ref, start, flen, ptr)
    finally:
        q.close()
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    assert (66000 & 0xFFFF) in got[4][int(got[0][-2]):].tolist()
    # reference ids that do not fit the packed 64-bit key (>= 65536: contig-level assemblies) take the 16-byte-record
    # kernel; and a second call on the same context reuses its device buffers
    ref2 = ref.copy()
    ref2[::7] += 70000
    want2 = oracle.atac_dedup(ref2, start, flen, ptr)
    q = pkg.Quantifier(pkg.WorkerConfig.for_resolution("cr-like", num_genes=1, num_rows=1), np.zeros(1, np.uint32))
    try:
        got2 = q.atac_dedup(ref2, start, flen, ptr)
        got1 = q.atac_dedup(ref, start, flen, ptr)
    finally:
        q.close()
    for g, w in zip(got2, want2):
        assert np.array_equal(g, w)
    for g, w in zip(got1, want):
        assert np.array_equal(g, w)


@pytest.mark.parametrize("usa", [False, True])
@pytest.mark.parametrize("init_uniform", [False, True])
def test_crlike_em(oracle, usa, init_uniform):
    """cr-like-em: winner-take-all ties kept as gene-level classes + per-cell EM (em.rs).  The device runs the
    same f32 operation sequence as the oracle (canonical class order), so the comparison is bitwise; the
    north-star tolerance for EM resolutions is 1e-4 relative."""
    sizes = [20000, 6000, 1500, 700, 260, 250, 120, 99, 40, 3]
    s = synth.synth(14, sizes, num_genes=300, usa=usa, dup=0.5, zipf=0.5, cross=0.7, max_extra_na=6)
    b, off = s.encode()
    cfg = cfg_for(s, "cr-like-em", em_init_uniform=init_uniform)
    got, want, _ = run_both(oracle, cfg, s.tid_to_gid, b, off)
    assert np.array_equal(got.cell_ptr, want.cell_ptr) and np.array_equal(got.gene, want.gene)
    np.testing.assert_allclose(got.val, want.val, rtol=1e-4, atol=0)
    assert_same_result(got, want)  # and in fact bit-identical
    # the EM really ran: some counts are fractional, and mass is above plain cr-like
    plain = oracle.quant(cfg_for(s, "cr-like"), s.tid_to_gid, b, off)
    assert (got.val != np.round(got.val)).any() and got.val.sum() > plain.val.sum()


def test_crlike_em_overflow_paths(oracle):
    """EM labels emitted from the mid-size and global-scratch bucket paths."""
    for n_same in (1500, 12000):
        s = synth.synth(15, [n_same + 3000, 500], num_genes=120, dup=0.3, cross=0.8)
        umi = s.umi.copy()
        umi[:n_same] = 0x00F0F0
        s.umi = umi
        b, off = s.encode()
        got, want, st = run_both(oracle, cfg_for(s, "cr-like-em"), s.tid_to_gid, b, off)
        assert st["n_overflow_buckets"] >= 1
        assert_same_result(got, want)


def test_many_ranges_pipeline(oracle, monkeypatch):
    """A batch cut into many ranges (forced by AFQ_TEST_RANGE_BYTES) goes through the two-buffer-set pipeline
    and yields the same rows in the same cell order."""
    monkeypatch.setenv("AFQ_TEST_RANGE_BYTES", "200000")
    sizes = [5000, 3000, 2500, 2000, 1500, 1200, 900, 600, 300, 120, 60, 20, 5, 1]
    for res in ("cr-like", "cr-like-em", "parsimony-em"):
        s = synth.synth(41, sizes, num_genes=200, txp_per_gene=3, dup=0.5, cross=0.4, umi_err=0.02)
        b, off = s.encode()
        got, want, st = run_both(oracle, cfg_for(s, res), s.tid_to_gid, b, off)
        assert_same_result(got, want, what=res)
    # the per-cell extras (-d classes, -b summaries) are appended range by range too
    cfg = cfg_for(s, "parsimony-em", dump_eq=True, num_bootstraps=3, summary_stat=True, boot_seed=9)
    got, want, _ = run_both(oracle, cfg, s.tid_to_gid, b, off)
    assert_same_result(got, want)
    for i in range(got.n_cells):
        assert got.eqclasses.cell(i) == want.eqclasses.cell(i), i
        assert np.array_equal(got.bootstraps.mean(i)[0], want.bootstraps.mean(i)[0]), i
        assert np.array_equal(got.bootstraps.mean(i)[1].view(np.uint32), want.bootstraps.mean(i)[1].view(np.uint32)), i
        assert np.array_equal(got.bootstraps.var(i)[1].view(np.uint32), want.bootstraps.var(i)[1].view(np.uint32)), i


@pytest.mark.parametrize("ranges", ["one", "many"])
@pytest.mark.parametrize("resolution", ["cr-like", "trivial", "parsimony", "cr-like-em"])
def test_context_reused_across_batches(oracle, monkeypatch, resolution, ranges):
    """One context, batch after batch (what a host streaming a file does, and every bench step after the first).  The rows of a
    range are compacted behind its kernels into the slot's row buffers as the previous batches left them (k_row_ptr + k_compact;
    csrc/afq_api.cpp: run_range / finish_range): a first batch finds none (the host compacts after allocating), a repeat fits, a
    larger batch does not fit and is compacted again once the buffers have grown, a smaller one fits with room to spare.  Same
    rows as the oracle every time - in both orders of sizes, with one range per batch and with many (two slots taking turns)."""
    if ranges == "many":
        monkeypatch.setenv("AFQ_TEST_RANGE_BYTES", "150000")
    s = synth.synth(52, [6000, 5000, 3000, 2500, 1500, 800, 300, 90, 20, 3], num_genes=400, txp_per_gene=2, dup=0.4, cross=0.3, umi_err=0.02)
    b, off = s.encode()
    n = len(off)
    cfg = cfg_for(s, resolution)

    def batch(a, e):
        lo, hi = int(off[a]), (int(off[e]) if e < n else len(b))
        return b[lo:hi], np.asarray(off[a:e], np.uint64) - np.uint64(lo)

    growing = [(n - 2, n), (n // 2, n), (n // 2, n), (0, n), (0, n), (2, n)]   # tiny, larger, repeat, whole, repeat, a little less
    for cuts in (growing, growing[::-1]):
        q = pkg.Quantifier(cfg, s.tid_to_gid)
        try:
            for a, e in cuts:
                bb, oo = batch(a, e)
                assert_same_result(q.quant_chunks(bb, oo), oracle.quant(cfg, s.tid_to_gid, bb, oo), what=f"{resolution} cells [{a}, {e})")
        finally:
            q.close()


@pytest.mark.parametrize("usa", [False, True])
@pytest.mark.parametrize("pad_reads", [0, 3000])
def test_tie_shapes_and_wide_umis(oracle, usa, pad_reads):
    """Every way a UMI's (gene, reads) counters can look: 1-6 genes per UMI, ties of 2 and 3 winners, spliced /
    unspliced siblings, plus UMIs that do not fit 32 bits (8-byte UMI field) including 0xFFFFFFFF itself.
    pad_reads > 0 pushes the cell over one bucket."""
    rng = np.random.default_rng(11)
    n_txp = 64
    t2g = (np.arange(n_txp, dtype=np.uint32) % 32) if not usa else np.arange(n_txp, dtype=np.uint32) % 32
    num_genes = 32
    reads = []
    umi = 1000
    for n_genes in range(1, 7):
        for shape in range(40):
            umi += 7
            txps = rng.choice(n_txp // 2, size=n_genes, replace=False)
            if usa and shape % 3 == 0 and n_genes >= 2:
                txps[1] = txps[0] ^ 1  # spliced/unspliced sibling pair (gene ids 2g, 2g+1)
            counts = rng.integers(1, 4, size=n_genes)
            if shape % 2 == 0:
                counts[:] = counts[0]  # all tied
            for t, c in zip(txps, counts):
                reads += [(umi, [int(t)])] * int(c)
            if shape % 5 == 0:
                reads.append((umi, sorted(int(x) for x in txps)))  # one multi-mapping read over all of them
    for wide in (0xFFFFFFFF, 0xFFFFFFFE, 0x100000005, 0xABC12345678):
        reads += [(wide, [3]), (wide, [3]), (wide, [9])]
    for _ in range(pad_reads):
        reads.append((int(rng.integers(1 << 20, 1 << 24)), [int(rng.integers(0, n_txp))]))
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    cells = [(5, reads), (6, reads[: len(reads) // 3])]
    b, off = rad.encode_cells(cells, 4, 8)
    cfg = pkg.WorkerConfig.for_resolution("cr-like", usa_mode=usa, num_genes=num_genes, num_rows=(num_genes // 2) * 3 if usa else num_genes,
                                          bc_bytes=4, umi_bytes=8, small_thresh=0)
    got, want, _ = run_both(oracle, cfg, t2g.astype(np.uint32), b, off)
    assert_same_result(got, want)
    assert got.val.sum() > 0


def set_decoder(monkeypatch, decoder):
    """recs: one lane per record; keys: one lane per dword, a record's repeated genes found by look-back compares and a scan;
    keys-hash: one lane per dword, repeated genes found through an LDS hash table (AFQ_TEST_DECODE_DEDUP=hash)."""
    monkeypatch.setenv("AFQ_TEST_DECODE", "keys" if decoder == "keys-hash" else decoder)
    monkeypatch.setenv("AFQ_TEST_DECODE_DEDUP", "hash" if decoder == "keys-hash" else "scan")   # (hash is the default)


@pytest.mark.parametrize("decoder", ["keys", "keys-hash"])
@pytest.mark.parametrize("usa", [False, True])
def test_many_gene_reads_both_dedups(oracle, monkeypatch, decoder, usa):
    """Reads of many alignments that repeat genes (the bench's label-length tail: geometric extra refs on gene families, up to
    64 per read): which alignment of a read is the first to name its gene decides the read's keys.  Cells from a handful of
    reads to many buckets; the lane-per-dword decoder with both ways of finding the repeats, against the oracle."""
    import importlib

    sn = importlib.import_module("alevin-fry_amd.synth_native")
    set_decoder(monkeypatch, decoder)
    d = sn.generate(seed=13 + usa, n_cells=90, median_reads=2500.0, sigma=1.4, num_genes=300, txp_per_gene=4, usa=usa, umi_err=0.02,
                    tail=0.75, tail_max=64, family=8)
    for resolution in ("cr-like", "cr-like-em"):
        cfg = pkg.WorkerConfig.for_resolution(resolution, usa_mode=usa, num_genes=d.num_genes, num_rows=d.num_rows, umi_len=12)
        got, want, st = run_both(oracle, cfg, d.tid_to_gid, d.data, d.chunk_off)
        assert_same_result(got, want, what=f"{resolution} usa={usa} {decoder}")
        assert st["n_fallback_cells"] == 0


@pytest.mark.parametrize("decoder", ["recs", "keys", "keys-hash"])
@pytest.mark.parametrize("resolution", ["cr-like", "trivial"])
def test_long_and_straddling_records(oracle, monkeypatch, decoder, resolution):
    """Records of every awkward length for the walk-free decoders: 0 alignments, just around the inline limits
    (3, 4), around the 64-dword halo, around a whole 256-dword slab, and thousands of alignments (a record that
    spans many slabs, so later slabs start in the middle of it).  Alignments repeat genes on purpose.  Both
    decoders (one lane per record / one lane per dword, the latter with both ways of finding a record's repeated genes) must
    agree with the oracle bit for bit."""
    set_decoder(monkeypatch, decoder)
    rng = np.random.default_rng(5)
    n_txp, n_genes = 6000, 700
    t2g = (rng.permutation(n_txp) % n_genes).astype(np.uint32)
    lens = [0, 1, 2, 3, 4, 5, 8, 9, 59, 60, 61, 62, 63, 64, 65, 66, 250, 251, 252, 253, 254, 255, 256, 257, 258, 319, 320, 321,
            1000, 5000]
    cells = []
    for ci in range(3):
        reads = []
        umi = 50
        for rep in range(3):
            for n in rng.permutation(lens):
                umi += int(rng.integers(1, 3))
                refs = sorted(int(x) for x in rng.choice(n_txp, size=int(n), replace=False)) if n else []
                reads.append((umi, refs))
                for _ in range(int(rng.integers(0, 4))):  # short records in between, some sharing the UMI
                    reads.append((umi + int(rng.integers(0, 2)), sorted(int(x) for x in rng.choice(n_txp, size=int(rng.integers(1, 4)), replace=False))))
        cells.append((0x5A5A0000 + ci, reads if ci else reads[: len(reads) // 2]))  # a barcode no UMI / ref / count collides with
    b, off = rad.encode_cells(cells, 4, 4)
    cfg = pkg.WorkerConfig.for_resolution(resolution, num_genes=n_genes, num_rows=n_genes, small_thresh=0)
    got, want, st = run_both(oracle, cfg, t2g, b, off)
    assert_same_result(got, want)
    assert st["n_fallback_cells"] == 0  # the walk-free proof held; nothing went through the sequential walk
    assert got.val.sum() > 0


@pytest.mark.parametrize("resolution", ["cr-like", "cr-like-em", "parsimony-em", "trivial"])
def test_empty_and_single_cell_batches(oracle, resolution):
    """A batch with no cells gives an empty result (no error); a one-read cell and a one-cell batch work in every mode."""
    s = synth.synth(9, [1, 700, 3], num_genes=60, dup=0.3)
    b, off = s.encode()
    cfg = cfg_for(s, resolution, small_thresh=0 if resolution != "cr-like" else 100)
    q = pkg.Quantifier(cfg, s.tid_to_gid)
    try:
        none = q.quant_chunks(b, off[:0])
        assert none.n_cells == 0 and len(none.gene) == 0 and list(none.cell_ptr) == [0]
        for sel in ([0], [1], [2], [0, 1, 2]):
            got = q.quant_chunks(b, off[sel])
            want = oracle.quant(cfg, s.tid_to_gid, b, off[sel])
            assert_same_result(got, want, what=f"{resolution} cells {sel}")
        again = q.quant_chunks(b, off[:0])   # and an empty batch after real ones
        assert again.n_cells == 0
    finally:
        q.close()


@pytest.mark.parametrize("usa", [False, True])
def test_giant_cell_beyond_the_lds_histogram(oracle, usa):
    """A 600 k-read cell gets 4096 buckets - more than the LDS histogram / multisplit hold (2048) - so its tiles
    take the straight-to-global paths of k_hist and k_scatter; a small neighbour rides along."""
    s = synth.synth(21, [600000, 50], num_genes=3000, usa=usa, dup=0.4, zipf=0.8)
    b, off = s.encode()
    got, want, st = run_both(oracle, cfg_for(s), s.tid_to_gid, b, off)
    assert st["n_buckets"] >= 4096
    assert_same_result(got, want)


@pytest.mark.parametrize("env", [{"AFQ_TEST_FIXED_SLABS": "0"}, {"AFQ_TEST_SLAB_CAP": "8"}, {"AFQ_TEST_SLAB_CAP": "200"}, {}])
@pytest.mark.parametrize("res,usa", [("cr-like", False), ("cr-like", True), ("cr-like-em", True)])
def test_bucket_placement_routes_agree(oracle, monkeypatch, env, res, usa):
    """Keys of a multi-bucket cell go to fixed-capacity bucket slabs without a counting pass; a bucket that outgrows its
    slab sends its cell through the exact placement (scan of the counts the cursors already hold).  The exact route
    alone (AFQ_TEST_FIXED_SLABS=0), slabs so small that every cell overflows (8), slabs that only the heavy-UMI cell
    overflows (200), and the default: all against the oracle."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    sizes = [60000, 20000, 9000, 5000, 3000, 1200, 700, 300, 90, 12]
    s = synth.synth(77, sizes, num_genes=600, txp_per_gene=3, usa=usa, dup=0.5, zipf=0.7, cross=0.3, umi_err=0.02, max_extra_na=6)
    # one UMI with hundreds of reads over a few genes in the second cell: a bucket far above the mean
    r0 = int(s.cell_nrec[0])
    s.umi[r0:r0 + 900] = s.umi[r0]
    b, off = s.encode()
    got, want, st = run_both(oracle, cfg_for(s, res), s.tid_to_gid, b, off)
    assert_same_result(got, want, what=str(env))
    assert got.val.sum() > 0
