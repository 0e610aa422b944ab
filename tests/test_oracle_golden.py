"""CPU tests: the oracle against the hand-derived / reference-derived known answers."""
import numpy as np
import pytest

from util import load_golden, pkg, rows_of

rad = pkg.rad


@pytest.mark.parametrize("small_thresh", [0, 100])
@pytest.mark.parametrize("route", [0, 1, 2])
def test_crlike_hand_cases(oracle, small_thresh, route):
    """Tiny path, from-reads route and EqMap route all give the hand-derived counts."""
    for case in load_golden("crlike_hand_cases.json")["cases"]:
        cells = [(c["bc"], [(u, r) for u, r in c["reads"]]) for c in case["cells"]]
        b, off = rad.encode_cells(cells, 4, 4)
        cfg = pkg.WorkerConfig.for_resolution("cr-like", usa_mode=case["usa"], num_genes=case["num_genes"],
                                              num_rows=case["num_rows"], small_thresh=small_thresh)
        res = oracle.quant(cfg, np.asarray(case["t2g"], np.uint32), b, off, force_route=route)
        got = rows_of(res)
        for c, g in zip(case["cells"], got):
            assert [[int(a), int(b_)] for a, b_ in g] == c["expected"], (case["name"], c["bc"], c["why"])
        for i, c in enumerate(case["cells"]):
            assert int(res.bc[i]) == c["bc"] and int(res.nrec[i]) == len(c["reads"])
            tiny = bool(res.flags[i] & pkg._abi.CELL_TINY_PATH)
            assert tiny == (route == 0 and len(c["reads"]) < small_thresh)
            assert bool(res.flags[i] & pkg._abi.CELL_EMPTY) == (len(c["expected"]) == 0)


@pytest.mark.parametrize("route", [0, 1, 2])
def test_prefer_ambig_hand_cases(oracle, route):
    """`--sa-model prefer-ambig` (pugutils.rs:505-641): reads of a UMI are tallied per gene, both splicing states together;
    the tiny-cell path is off (quant.rs:794), both remaining routes give the hand-derived counts, and the same cell
    under winner-take-all gives the contrasting ones."""
    for case in load_golden("prefer_ambig_hand_cases.json")["cases"]:
        cells = [(c["bc"], [(u, r) for u, r in c["reads"]]) for c in case["cells"]]
        b, off = rad.encode_cells(cells, 4, 4)
        t2g = np.asarray(case["t2g"], np.uint32)
        kw = dict(usa_mode=case["usa"], num_genes=case["num_genes"], num_rows=case["num_rows"])
        res = oracle.quant(pkg.WorkerConfig.for_resolution("cr-like", sa_model="prefer-ambig", **kw), t2g, b, off, force_route=route)
        wta = oracle.quant(pkg.WorkerConfig.for_resolution("cr-like", **kw), t2g, b, off, force_route=route)
        for c, g, w in zip(case["cells"], rows_of(res), rows_of(wta)):
            assert [[int(a), int(v)] for a, v in g] == c["expected"], (case["name"], c["bc"], c["why"])
            assert [[int(a), int(v)] for a, v in w] == c["expected_wta"], (case["name"], c["bc"], "winner-take-all")
        assert not (res.flags & pkg._abi.CELL_TINY_PATH).any()
    # outside USA mode the switch is ignored (quant.rs:1456-1469)
    case = load_golden("crlike_hand_cases.json")["cases"][0]
    cells = [(c["bc"], [(u, r) for u, r in c["reads"]]) for c in case["cells"]]
    b, off = rad.encode_cells(cells, 4, 4)
    res = oracle.quant(pkg.WorkerConfig.for_resolution("cr-like", sa_model="prefer-ambig", num_genes=4, num_rows=4),
                       np.asarray(case["t2g"], np.uint32), b, off, force_route=route)
    for c, g in zip(case["cells"], rows_of(res)):
        assert [[int(a), int(v)] for a, v in g] == c["expected"]


def test_philox_known_answers(oracle):
    """The bootstrap draws run on Philox4x32-10; the oracle's restatement against the known-answer vectors published
    with the Random123 library (kat_vectors: philox4x32 10)."""
    assert oracle.philox4x32_10([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert oracle.philox4x32_10([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert oracle.philox4x32_10([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_bootstrap_oracle_properties(oracle):
    """-b restatement (em.rs:585-690): every replicate redistributes the cell's molecules, so the means add up to them;
    the mean tracks the point estimate, the variance is that of a multinomial share (~ mean for small shares); classes with
    one gene only (no EM) have exactly multinomial moments; tiny-path cells have no bootstraps; --summary-stat switches the
    variance formula (population vs n-1), not the mean."""
    s = pkg.synth.synth(41, [60, 4000], num_genes=40, dup=0.4, cross=0.3, max_extra_na=3)
    b, off = s.encode()
    kw = dict(num_genes=s.num_genes, num_rows=s.num_rows, num_bootstraps=200, boot_seed=3)
    r1 = oracle.quant(pkg.WorkerConfig.for_resolution("cr-like-em", summary_stat=True, **kw), s.tid_to_gid, b, off)
    r0 = oracle.quant(pkg.WorkerConfig.for_resolution("cr-like-em", summary_stat=False, **kw), s.tid_to_gid, b, off)
    assert len(r1.bootstraps.mean(0)[0]) == 0 and bool(r1.flags[0] & pkg._abi.CELL_TINY_PATH)
    g, v = r1.row(1)
    mc, mv = r1.bootstraps.mean(1)
    vc, vv = r1.bootstraps.var(1)
    n_mol = float(v.sum())
    assert abs(float(mv.sum()) - n_mol) < 1e-3 * n_mol
    est = dict(zip(g.tolist(), v.tolist()))
    big = [(c, m) for c, m in zip(mc.tolist(), mv.tolist()) if est.get(c, 0) > 50]
    assert len(big) > 10 and all(abs(m - est[c]) < 0.15 * est[c] for c, m in big)
    var = dict(zip(vc.tolist(), vv.tolist()))
    assert all(0.4 * m < var[c] < 2.5 * m for c, m in big)
    m0c, m0v = r0.bootstraps.mean(1)
    assert np.array_equal(m0c, mc) and np.allclose(m0v, mv, rtol=1e-5)
    v0 = dict(zip(*[x.tolist() for x in r0.bootstraps.var(1)]))
    assert all(abs(v0[c] * 199.0 / 200.0 - var[c]) <= 2e-3 * var[c] + 1e-3 for c, _ in big)
    with pytest.raises(Exception):   # main.rs:713-728
        oracle.quant(pkg.WorkerConfig.for_resolution("cr-like", summary_stat=True, **kw), s.tid_to_gid, b, off)


@pytest.mark.parametrize("bw,uw", [(1, 1), (2, 2), (4, 4), (8, 8), (2, 4), (4, 2), (8, 4)])
def test_field_widths(oracle, bw, uw):
    """Record field widths 1/2/4/8 bytes (src/convert.rs:323-344) decode identically."""
    case = load_golden("crlike_hand_cases.json")["cases"][0]
    cells = [(c["bc"], [(u, r) for u, r in c["reads"]]) for c in case["cells"]]
    b, off = rad.encode_cells(cells, bw, uw)
    cfg = pkg.WorkerConfig.for_resolution("cr-like", num_genes=4, num_rows=4, bc_bytes=bw, umi_bytes=uw)
    res = oracle.quant(cfg, np.asarray(case["t2g"], np.uint32), b, off)
    for c, g in zip(case["cells"], rows_of(res)):
        assert [[int(a), int(b_)] for a, b_ in g] == c["expected"]


def test_three_crlike_routes_agree_on_synthetic(oracle):
    """quant.rs:794-880: tiny / <=250 / EqMap routes are the same function of the cell."""
    for usa in (False, True):
        s = pkg.synth.synth(7, [40, 90, 200, 260, 600, 1500], num_genes=60, usa=usa, dup=0.5, max_extra_na=12)
        b, off = s.encode()
        cfg = pkg.WorkerConfig.for_resolution("cr-like", usa_mode=usa, num_genes=s.num_genes, num_rows=s.num_rows)
        r0 = oracle.quant(cfg, s.tid_to_gid, b, off, force_route=0)
        r1 = oracle.quant(cfg, s.tid_to_gid, b, off, force_route=1)
        r2 = oracle.quant(cfg, s.tid_to_gid, b, off, force_route=2)
        for r in (r1, r2):
            assert np.array_equal(r0.cell_ptr, r.cell_ptr) and np.array_equal(r0.gene, r.gene) and np.array_equal(r0.val, r.val)
        assert r0.val.sum() > 0


def test_oracle_rejects_corrupt_chunks(oracle):
    case = load_golden("crlike_hand_cases.json")["cases"][0]
    cells = [(c["bc"], [(u, r) for u, r in c["reads"]]) for c in case["cells"]]
    b, off = rad.encode_cells(cells, 4, 4)
    cfg = pkg.WorkerConfig.for_resolution("cr-like", num_genes=4, num_rows=4)
    t2g = np.asarray(case["t2g"], np.uint32)
    bad = bytearray(b)
    bad[8] = 200  # first record's na
    with pytest.raises(oracle.OracleError):
        oracle.quant(cfg, t2g, bytes(bad), off)
    with pytest.raises(oracle.OracleError):  # ref id out of range
        oracle.quant(cfg, t2g[:3], b, off)


def test_em_known_answers(oracle):
    """The reference's own EM unit tests (src/em.rs:1176-1215 and the two before): the sparse-support
    EM equals `dense_reference` bit for bit, for Uniform/Informative init, non-USA and USA, and at the
    0.01 output threshold where the result must contain an exact 0.0."""
    eq = [[0], [1], [0, 1], [1, 2], [2, 3, 4]]
    cases = [[], [(0, 7)], [(0, 20), (1, 4), (2, 8), (3, 1), (4, 2)]]
    for uni in (True, False):
        for cd in cases:
            labels = [eq[i] for i, _ in cd]
            counts = [c for _, c in cd]
            sp, _ = oracle.em(labels, counts, 8, init_uniform=uni, dense=0)
            de, _ = oracle.em(labels, counts, 8, init_uniform=uni, dense=2)
            assert np.array_equal(sp.view(np.uint32), de.view(np.uint32))
    eq = [[0, 1], [3, 4], [6, 7], [0, 4, 8], [2]]
    cases = [[(0, 5)], [(1, 5)], [(2, 5)], [(0, 3), (1, 4), (2, 5), (3, 7), (4, 2)]]
    for uni in (True, False):
        for cd in cases:
            labels = [eq[i] for i, _ in cd]
            counts = [c for _, c in cd]
            sp, _ = oracle.em(labels, counts, 9, init_uniform=uni, usa_offsets=(3, 6), dense=0)
            de, _ = oracle.em(labels, counts, 9, init_uniform=uni, usa_offsets=(3, 6), dense=2)
            assert np.array_equal(sp.view(np.uint32), de.view(np.uint32))
            assert abs(float(sp.sum()) - sum(counts)) < 0.05 * sum(counts) + 0.1
    labels = [[0], [0, 1], [1, 2], [2, 3, 4]]
    counts = [10000, 1, 1, 1]
    sp, _ = oracle.em(labels, counts, 6, dense=0)
    de, _ = oracle.em(labels, counts, 6, dense=2)
    assert np.array_equal(sp.view(np.uint32), de.view(np.uint32))
    assert (sp == 0.0).any(), "threshold case must contain an exact 0.0 (em.rs:1176-1215)"
    # only_unique returns the singleton counts untouched (em.rs:499-514)
    u, _ = oracle.em([[0], [1], [0, 1]], [3, 4, 5], 4, only_unique=True, dense=1)
    assert u.tolist() == [3.0, 4.0, 0.0, 0.0]


def test_em_by_hand(oracle):
    """Two classes {0}:2, {0,1}:2, informative init a0=(2+.5)e-3, a1=.5e-3 (em.rs:519-531).
    Update (em.rs:458-485): a1' = 2*a1/(a0+a1), a0' = 4 - a1'.  a1 = 1/3, 1/6, ... halves each round.
    Round 6: a1'=.0104 > .01 and |delta| = .0104 > .01 -> not converged; round 7: a1'=.0052 <= .01 is
    not checked and |delta a0| = .0052 -> converged after 7 rounds (em.rs:538-565).  The 0.01 output
    floor (em.rs:568-572) then zeroes a1, leaving a0 = 4 - (1/3)/64."""
    a, it = oracle.em([[0], [0, 1]], [2, 2], 2, dense=1)
    assert it == 7
    assert a[1] == 0.0 and abs(float(a[0]) - (4.0 - (1.0 / 3.0) / 64.0)) < 1e-5


def test_has_edge_boundaries(oracle):
    """pugutils.rs:76-99.  Returns 0 none, 1 x->y, 2 y->x, 3 bidirected."""
    A = 0b000000
    B = 0b000001  # one base differs
    C = 0b000101  # two bases differ from A
    assert oracle.has_edge(A, 5, B, 2) == 1  # 5 > 2*2-1
    assert oracle.has_edge(A, 2, B, 5) == 2
    assert oracle.has_edge(A, 3, B, 2) == 3  # 3 > 3 false, 2 > 5 false
    assert oracle.has_edge(A, 4, B, 2) == 1  # 4 > 3
    assert oracle.has_edge(A, 1, A, 9) == 3  # identical UMIs
    assert oracle.has_edge(A, 1, C, 1) == 0
    assert oracle.has_edge(A, 5, B, 2, exact=True) == 0
    assert oracle.has_edge(0b11, 1, 0b00, 1) == 3  # both bits of one base differ: still distance 1


def test_atac_dedup_known_answer(oracle):
    """atac/deduplicate.rs:199-237: sort by (chr,start,frag_len), run-length count."""
    ref = [1, 0, 1, 0, 1]
    start = [10, 5, 10, 5, 10]
    flen = [100, 50, 100, 60, 100]
    ptr, r, s, f, c = oracle.atac_dedup(ref, start, flen, [0, 5])
    assert ptr.tolist() == [0, 3]
    assert list(zip(r.tolist(), s.tolist(), f.tolist(), c.tolist())) == [(0, 5, 50, 1), (0, 5, 60, 1), (1, 10, 100, 3)]


def test_pug_hand_cases(oracle):
    """Hand-derived parsimony known-answers (tests/golden/pug_hand_cases.json)."""
    d = load_golden("pug_hand_cases.json")
    cells = [(c["bc"], [(u, r) for u, r in c["reads"]]) for c in d["cells"]]
    b, off = rad.encode_cells(cells, 4, 4)
    t2g = np.asarray(d["t2g"], np.uint32)
    cfg = pkg.WorkerConfig.for_resolution("parsimony", num_genes=d["num_genes"], num_rows=d["num_genes"], small_thresh=0)
    for c, g in zip(d["cells"], rows_of(oracle.quant(cfg, t2g, b, off))):
        assert [[int(a), int(v)] for a, v in g] == c["expected"], (c["bc"], c["why"])
    cr = pkg.WorkerConfig.for_resolution("cr-like", num_genes=d["num_genes"], num_rows=d["num_genes"], small_thresh=0)
    for c, g in zip(d["cells"], rows_of(oracle.quant(cr, t2g, b, off))):
        if "crlike" in c:
            assert [[int(a), int(v)] for a, v in g] == c["crlike"]
    # the reference's structural known-answer (tests/multi_barcode_integration.rs:1404-1556): on cells whose reads all
    # hit two genes with distinct UMIs, cr-like mass is 0 while parsimony-em --small-thresh 0 keeps the mass
    case = load_golden("crlike_hand_cases.json")["cases"][3]
    cells = [(c["bc"], [(u, r) for u, r in c["reads"]]) for c in case["cells"]][:1]
    b, off = rad.encode_cells(cells, 4, 4)
    t2g = np.asarray(case["t2g"], np.uint32)
    pem = pkg.WorkerConfig.for_resolution("parsimony-em", num_genes=10, num_rows=10, small_thresh=0)
    r_em = oracle.quant(pem, t2g, b, off)
    r_tiny = oracle.quant(pkg.WorkerConfig.for_resolution("parsimony-em", num_genes=10, num_rows=10), t2g, b, off)
    # (consecutive integer UMIs are one base apart and neighbouring labels overlap, so the PUG merges some molecules:
    #  the reference asserts only mass > 0 = fast-path mass, and so do we)
    assert float(r_em.val.sum()) > 0.0 and float(r_tiny.val.sum()) == 0.0
    assert (r_tiny.flags & pkg._abi.CELL_TINY_PATH).all() and not (r_em.flags & pkg._abi.CELL_TINY_PATH).any()


def _atac_reference_cells(num_cells=16, good=5, unmapped=2, multimapped=1):
    """The synthetic scATAC input of the reference's own test (tests/atac_integration.rs:150-224): per cell `good` properly
    paired unique fragments (type 4, length 120, spread-out starts), `unmapped` records without alignments, `multimapped`
    records with two alignments placed beyond position 500 000."""
    cells = []
    for ci in range(num_cells):
        bc = ((ci + 1) * 2654435761) & 0xFFFFFFFF
        recs = [[(r % 2, 4, 1000 + ci * 977 + r * 131, 120)] for r in range(good)]
        recs += [[] for _ in range(unmapped)]
        recs += [[(0, 4, 500000 + ci * 13 + r * 7, 110), (1, 4, 500000 + ci * 13 + r * 7 + 250, 115)] for r in range(multimapped)]
        cells.append((bc, recs))
    return cells


def test_atac_dedup_from_rad_reference_vector(oracle):
    """atac/deduplicate.rs:199-237 from the records themselves: the reference's structural vector - 16 cells x 5 good
    fragments => 80 distinct fragments, every count >= 1, no multi-mapping record among them
    (tests/atac_integration.rs:531-607) - plus the counters the reference logs."""
    from util import pkg

    cells = _atac_reference_cells()
    b, off = pkg.rad.encode_atac_cells(cells)
    ptr, bc, ref, start, flen, cnt, st = oracle.atac_dedup_rad(b, off)
    assert int(ptr[-1]) == 16 * 5 and (cnt >= 1).all() and (start < 500000).all() and set(ref.tolist()) <= {0, 1}
    assert (flen == 120).all() and [int(x) for x in bc] == [c[0] for c in cells]
    assert st == dict(n_records=16 * 8, n_multimapped=16, n_not_mapped_pair=32, n_deduplicated=0, n_long_fragments=0)
    # duplicates, a long fragment, a one-alignment record that is not a proper pair (type 1), an 8-byte barcode
    recs = [[(3, 4, 50, 200)], [(3, 4, 50, 200)], [(1, 4, 9, 2000)], [(3, 1, 50, 200)], [(0, 4, 7, 30)], [(3, 4, 50, 199)]]
    b, off = pkg.rad.encode_atac_cells([(0x1122334455667788, recs)], bc_bytes=8)
    ptr, bc, ref, start, flen, cnt, st = oracle.atac_dedup_rad(b, off, bc_bytes=8)
    assert list(zip(ref.tolist(), start.tolist(), flen.tolist(), cnt.tolist())) == [(0, 7, 30, 1), (1, 9, 2000, 1), (3, 50, 199, 1), (3, 50, 200, 2)]
    assert int(bc[0]) == 0x1122334455667788 and st["n_not_mapped_pair"] == 1 and st["n_deduplicated"] == 1 and st["n_long_fragments"] == 1
    with pytest.raises(oracle.OracleError):   # records that do not tile their chunk
        oracle.atac_dedup_rad(b[:4] + (7).to_bytes(4, "little") + b[8:], off, bc_bytes=8)   # header says 7 records, there are 6


@pytest.mark.parametrize("bcb", [1, 2, 4, 8])
def test_atac_dedup_from_rad_is_layout_independent(oracle, bcb):
    """The same cells encoded with every barcode width, and with the chunks shifted to each byte alignment, must
    de-duplicate to the same fragments (the oracle is what the device's aligned-dword parse is compared with)."""
    from util import pkg

    rng = np.random.default_rng(3)
    cells = []
    for ci, n in enumerate([1, 4, 33, 200]):
        recs = []
        for _ in range(n):
            k = int(rng.choice([0, 1, 1, 1, 2]))
            recs.append([(int(rng.integers(0, 5)), int(rng.choice([4, 4, 1])), int(rng.integers(0, 50)), int(rng.integers(30, 2100))) for _ in range(k)])
        cells.append((9 + ci, recs))
    base = None
    b, off = pkg.rad.encode_atac_cells(cells, bc_bytes=bcb)
    for pad in range(4):
        data = bytes(pad) + b
        got = oracle.atac_dedup_rad(data, np.asarray(off, np.uint64) + np.uint64(pad), bc_bytes=bcb)
        key = [x.tolist() for x in got[:6]] + [got[6]]
        if base is None:
            base = key
        assert key == base, (bcb, pad)
    # ... and to what the 4-byte encoding gives
    b4, off4 = pkg.rad.encode_atac_cells(cells, bc_bytes=4)
    ref = oracle.atac_dedup_rad(b4, off4, bc_bytes=4)
    assert [x.tolist() for x in ref[:6]] + [ref[6]] == base


def test_tie_free_components_and_em_order_switches(oracle):
    """The two measurement switches of the oracle (bench.py reports them for configs[2]): covering every component with the
    scan reversed changes nothing in components that met no tie (pugutils.rs:1090-1146: only the choice between equally
    large arborescences depends on the scan), and summing the EM's classes in a shuffled order (em.rs:464 walks a
    HashMap) moves counts by rounding only; with both switches off the result is the canonical one, bit for bit."""
    import numpy as np

    pkg = __import__("importlib").import_module("alevin-fry_amd")
    s = pkg.synth.synth(31, [6000, 1500, 700, 260], num_genes=250, txp_per_gene=3, usa=True, dup=0.55, zipf=0.5, cross=0.3,
                        umi_err=0.03, max_extra_na=5)
    b, off = s.encode()
    cfg = pkg.WorkerConfig.for_resolution("parsimony-em", usa_mode=True, num_genes=s.num_genes, num_rows=s.num_rows)
    base = oracle.quant(cfg, s.tid_to_gid, b, off)
    chk, ps = oracle.quant(cfg, s.tid_to_gid, b, off, want_pug_stats=True, check_tie_free=True)
    assert np.array_equal(chk.val.view(np.uint32), base.val.view(np.uint32)) and np.array_equal(chk.gene, base.gene)
    assert ps[:, 1].sum() > 0, "the input is meant to have ties"
    assert ps[:, 4].sum() == 0, "a component without a tie event changed with the scan order"
    moved = 0
    for seed in (1, 2):
        o = oracle.quant(cfg, s.tid_to_gid, b, off, em_order_seed=seed)
        assert np.array_equal(o.cell_ptr, base.cell_ptr) and np.array_equal(o.gene, base.gene)
        np.testing.assert_allclose(o.val, base.val, rtol=1e-5, atol=0)
        moved += int((o.val.view(np.uint32) != base.val.view(np.uint32)).sum())
    assert moved > 0, "a shuffled summation order should move some last bits"
    again = oracle.quant(cfg, s.tid_to_gid, b, off)
    assert np.array_equal(again.val.view(np.uint32), base.val.view(np.uint32))

