"""GPU parity tests for the parsimony (PUG) path through the C ABI vs the CPU oracle."""
import numpy as np
import pytest

from util import assert_same_result, cfg_for, load_golden, pkg, rows_of

pytestmark = pytest.mark.gpu
rad = pkg.rad
synth = pkg.synth


@pytest.fixture(autouse=True, params=["phase-kernels", "one-workgroup", "handed-back", "cover-1024", "graph-per-cell", "graph-per-cell-1024", "graph-per-cell-ties-set-aside"])
def pug_route(request, monkeypatch):
    """Every test of this module runs seven times: through the partition-parallel phase kernels with the range-wide flat graph
    build (csrc/afq_pug2.hip + csrc/afq_pugflat.hip, the default; a cell with a component of more than 64 vertices is routed to
    the per-cell graph kernel from there), with every parsimony cell sent to the one-workgroup kernel (csrc/afq_pug.hip), with the
    phase kernels' partition capacity cut to 24 reads so that most cells start on the first route and are handed back to the
    second, with every cell of 300 reads or more given the 1024-thread instances of the cover / tie kernels (by default: cells
    of 15 000 reads), and three times with the per-cell graph kernel of rounds 3-5 for EVERY cell (AFQ_TEST_P2_GRAPH=cell): as
    it decides itself, with its 1024-thread instance from 300 reads, and with every cell covering its components in slot order
    and setting the tied ones aside for k_p2_tied (what the flat build always does)."""
    if request.param == "one-workgroup":
        monkeypatch.setenv("AFQ_TEST_PUG_ROUTE", "mono")
    elif request.param == "handed-back":
        monkeypatch.setenv("AFQ_TEST_P2_PART_CAP", "24")
    elif request.param == "cover-1024":
        monkeypatch.setenv("AFQ_TEST_P2_BIG_READS", "300")
    elif request.param == "graph-per-cell":
        monkeypatch.setenv("AFQ_TEST_P2_GRAPH", "cell")
    elif request.param == "graph-per-cell-1024":
        monkeypatch.setenv("AFQ_TEST_P2_GRAPH", "cell")
        monkeypatch.setenv("AFQ_TEST_P2_BIG_READS", "300")
        monkeypatch.setenv("AFQ_TEST_P2_DEFER_MIN", "0")
    elif request.param == "graph-per-cell-ties-set-aside":
        monkeypatch.setenv("AFQ_TEST_P2_GRAPH", "cell")
        monkeypatch.setenv("AFQ_TEST_P2_DEFER_MIN", "0")
    return request.param


def run_both(oracle, cfg, t2g, b, off):
    q = pkg.Quantifier(cfg, t2g)
    try:
        got = q.quant_chunks(b, off)
    finally:
        q.close()
    return got, oracle.quant(cfg, t2g, b, off)


def test_pug_hand_cases(oracle):
    d = load_golden("pug_hand_cases.json")
    cells = [(c["bc"], [(u, r) for u, r in c["reads"]]) for c in d["cells"]]
    b, off = rad.encode_cells(cells, 4, 4)
    t2g = np.asarray(d["t2g"], np.uint32)
    cfg = pkg.WorkerConfig.for_resolution("parsimony", num_genes=d["num_genes"], num_rows=d["num_genes"], small_thresh=0)
    got, want = run_both(oracle, cfg, t2g, b, off)
    for c, g in zip(d["cells"], rows_of(got)):
        assert [[int(a), int(v)] for a, v in g] == c["expected"], (c["bc"], c["why"])
    assert_same_result(got, want)


@pytest.mark.parametrize("usa", [False, True])
@pytest.mark.parametrize("res", ["parsimony", "parsimony-em", "parsimony-gene", "parsimony-gene-em"])
def test_parsimony_synthetic(oracle, usa, res):
    """Ragged cells incl. tiny ones (cr-like fast path), 3 % UMI errors so the PUG has real components."""
    sizes = [12000, 5000, 1500, 700, 260, 250, 120, 99, 40, 3]
    s = synth.synth(31, sizes, num_genes=250, txp_per_gene=3, usa=usa, dup=0.55, zipf=0.5, cross=0.3, umi_err=0.03, max_extra_na=5)
    b, off = s.encode()
    cfg = cfg_for(s, res)
    got, want = run_both(oracle, cfg, s.tid_to_gid, b, off)
    assert np.array_equal(got.cell_ptr, want.cell_ptr) and np.array_equal(got.gene, want.gene)
    np.testing.assert_allclose(got.val, want.val, rtol=1e-4, atol=0)  # the north-star tolerance for EM resolutions
    assert_same_result(got, want)  # and in fact bit-identical
    plain = oracle.quant(cfg_for(s, "cr-like"), s.tid_to_gid, b, off)
    assert not np.array_equal(plain.val, want.val) or not np.array_equal(plain.gene, want.gene)


def test_parsimony_exact_umi_and_small_thresh(oracle):
    """--umi-edit-dist 0 (only identical UMIs induce edges) and --small-thresh 0."""
    s = synth.synth(32, [3000, 500, 60], num_genes=120, txp_per_gene=3, dup=0.5, cross=0.3, umi_err=0.05)
    b, off = s.encode()
    for kw in (dict(pug_exact_umi=True), dict(small_thresh=0), dict(pug_exact_umi=True, small_thresh=0)):
        cfg = cfg_for(s, "parsimony", **kw)
        got, want = run_both(oracle, cfg, s.tid_to_gid, b, off)
        assert_same_result(got, want, what=str(kw))


def test_large_component_fallback_sets_alt_flag(oracle):
    """Components above --large-graph-thresh are resolved cr-like and the cell is flagged (pugutils.rs:1055-1072)."""
    s = synth.synth(33, [2500, 800], num_genes=60, txp_per_gene=2, dup=0.6, cross=0.2, umi_err=0.25)
    b, off = s.encode()
    cfg = cfg_for(s, "parsimony", large_graph_thresh=3)
    got, want = run_both(oracle, cfg, s.tid_to_gid, b, off)
    assert (want.flags & pkg._abi.CELL_ALT_RES).any()
    assert_same_result(got, want)


def test_components_beyond_one_wave(oracle):
    """All 256 UMIs over 4 positions on one transcript form one 256-vertex component (each vertex has 12
    neighbours); a second, overlapping class splits the cover.  Exercises the multi-word cover."""
    reads = []
    for u in range(256):
        umi = (u & 3) | (((u >> 2) & 3) << 4) | (((u >> 4) & 3) << 10) | (((u >> 6) & 3) << 20)
        reads.append((umi, [0] if u % 3 else [0, 2]))
        if u % 5 == 0:
            reads.append((umi, [0]))
    reads += [(0x555555, [4]), (0x555554, [4]), (0x555554, [4]), (0x555554, [4])]
    cells = [(77, reads), (78, reads[:130])]
    b, off = rad.encode_cells(cells, 4, 4)
    t2g = np.asarray([0, 0, 1, 1, 2, 2], np.uint32)
    for res in ("parsimony", "parsimony-em", "parsimony-gene"):
        cfg = pkg.WorkerConfig.for_resolution(res, num_genes=3, num_rows=3, small_thresh=0)
        got, want = run_both(oracle, cfg, t2g, b, off)
        assert_same_result(got, want, what=res)


def _quant(cfg, t2g, b, off):
    q = pkg.Quantifier(cfg, t2g)
    try:
        return q.quant_chunks(b, off)
    finally:
        q.close()


@pytest.mark.parametrize("res,usa", [("parsimony", False), ("parsimony-em", True)])
def test_neighbour_search_routes_agree(oracle, monkeypatch, res, usa):
    """The neighbour search runs out of an LDS hash table, the vertices cut into 4^k partitions by the low bases of their
    UMIs (csrc/afq_pug.hip, phase 4); cells whose partitions would not fit take the older route through a hash table in
    global memory.  Cell sizes for 1, 4, 16 and 64 partitions: both routes against each other, the smaller cells against
    the oracle too."""
    sizes = [150000, 30000, 9000, 2000, 400]
    s = synth.synth(41, sizes, num_genes=2000, txp_per_gene=3, usa=usa, dup=0.3, cross=0.3, umi_err=0.04)
    b, off = s.encode()
    cfg = cfg_for(s, res)
    fast = _quant(cfg, s.tid_to_gid, b, off)
    monkeypatch.setenv("AFQ_TEST_PUG_GLOBAL_ROUTE", "1")
    slow = _quant(cfg, s.tid_to_gid, b, off)
    monkeypatch.delenv("AFQ_TEST_PUG_GLOBAL_ROUTE")
    assert_same_result(fast, slow, what="LDS route vs global route")
    want = oracle.quant(cfg, s.tid_to_gid, b, off[1:], n_threads=4)
    for j in range(want.n_cells):
        g0, v0 = fast.row(j + 1)
        g1, v1 = want.row(j)
        assert np.array_equal(g0, g1) and np.array_equal(v0.view(np.uint32), v1.view(np.uint32)), f"cell {j + 1}"


def test_skewed_umis_fall_back_to_the_global_route(oracle):
    """UMIs that all share their low bases land in ONE partition; above the table's capacity the cell silently takes the
    global-memory route - same rows.  (Low bases fixed = e.g. a UMI read with a constant primer tail.)"""
    rng = np.random.default_rng(8)
    n = 24000
    umi = (rng.integers(0, 1 << 16, n).astype(np.int64) << 8) | 0x5A      # 4 constant low bases, 8 random ones
    reads = [(int(u), [int(t)] if t % 3 else [int(t), int(t) + 1]) for u, t in zip(umi, rng.integers(0, 300, n))]
    b, off = rad.encode_cells([(9, reads)], 4, 4)
    t2g = (np.arange(302) // 3).astype(np.uint32)
    cfg = pkg.WorkerConfig.for_resolution("parsimony", num_genes=101, num_rows=101, umi_len=12)
    assert_same_result(_quant(cfg, t2g, b, off), oracle.quant(cfg, t2g, b, off))


@pytest.mark.parametrize("decoder", ["recs", "keys"])
@pytest.mark.parametrize("res", ["parsimony", "parsimony-gene"])
def test_pug_batches_through_both_decoders(oracle, monkeypatch, decoder, res):
    """A parsimony batch turns records into reads (label key, UMI, offset) in the decode: the lane-per-record kernel when
    records are short (AFQ_TEST_DECODE=recs, the planner's choice for 10x data), the older per-record walk otherwise
    (AFQ_TEST_DECODE=keys).  Records of every awkward length (around the three inline refs, up to the 64-dword halo), repeated genes, tiny cells that take the cr-like rule next to PUG cells: both against the oracle."""
    monkeypatch.setenv("AFQ_TEST_DECODE", decoder)
    rng = np.random.default_rng(77)
    n_txp, n_genes = 900, 300
    t2g = (rng.permutation(n_txp) % n_genes).astype(np.uint32)
    lens = [1, 2, 3, 4, 5, 8, 9, 30, 59, 60, 61]   # (few UMIs: at gene level the cell is one component above --large-graph-thresh,
    # resolved per UMI, with ties among more genes than the device carries per molecule - dropped, as any tie is without an EM)
    cells = []
    for ci in range(4):
        reads = []
        for rep in range(40 if ci < 3 else 1):
            for n in rng.permutation(lens):
                umi = int(rng.integers(0, 1 << 10))   # few UMIs: neighbours and repeats
                refs = sorted(int(x) for x in rng.choice(n_txp, size=int(n), replace=False)) if n else []
                reads.append((umi, refs))
                for _ in range(int(rng.integers(0, 6))):
                    reads.append((int(rng.integers(0, 1 << 10)), sorted(int(x) for x in rng.choice(n_txp, size=int(rng.integers(1, 4)), replace=False))))
        cells.append((0x5A5A0000 + ci, reads))
    b, off = rad.encode_cells(cells, 4, 4)
    cfg = pkg.WorkerConfig.for_resolution(res, num_genes=n_genes, num_rows=n_genes, small_thresh=100)
    got, want = run_both(oracle, cfg, t2g, b, off)
    assert_same_result(got, want)
    assert got.val.sum() > 0


def _cells_of(s, bc_of):
    """A SynthRad as the (bc, [(umi, refs)]) lists rad.encode_cells takes (any field widths)."""
    cells, r, w = [], 0, 0
    for ci, n in enumerate(s.cell_nrec):
        reads = []
        for _ in range(int(n)):
            na = int(s.na[r])
            reads.append((int(s.umi[r]), [int(x) for x in s.refs[w:w + na]]))
            r += 1; w += na
        cells.append((bc_of(ci), reads))
    return cells


@pytest.mark.parametrize("bw,uw,umi_len", [(4, 2, 8), (2, 2, 8), (2, 1, 4), (1, 2, 7), (8, 2, 8), (2, 8, 12)])
@pytest.mark.parametrize("res,usa", [("parsimony", False), ("parsimony-em", True), ("cr-like", False)])
def test_narrow_fields_are_widened_on_the_device(oracle, bw, uw, umi_len, res, usa):
    """Barcode / UMI fields of 1 or 2 bytes (UMIs of up to 8 nt: Drop-seq, CEL-Seq2, inDrop) make records that are not
    dword aligned.  The batch is rewritten on the device with 4-byte fields (k_widen) and then takes the ordinary path -
    parsimony included, which used to refuse such input.  Against the oracle reading the original bytes."""
    s = synth.synth(90 + bw + uw, [2500, 600, 150, 120, 40, 7], num_genes=300, txp_per_gene=3, usa=usa, umi_len=umi_len, dup=0.4, cross=0.3,
                    umi_err=0.03, max_extra_na=6)
    cells = _cells_of(s, lambda ci: 3 + 5 * ci)   # barcodes that fit one byte
    b, off = rad.encode_cells(cells, bw, uw)
    cfg = cfg_for(s, res, bc_bytes=bw, umi_bytes=uw, umi_len=umi_len)
    got, want = run_both(oracle, cfg, s.tid_to_gid, b, off)
    assert_same_result(got, want)
    assert got.val.sum() > 0
    assert list(got.bc) == [3 + 5 * ci for ci in range(len(cells))]


def test_parsimony_over_chunks_at_odd_offsets(oracle):
    """4-byte fields, but the caller's chunks sit at offsets that differ mod 4 (so no single shift aligns them): the
    batch is repacked on the device by the same kernel (nothing widened) instead of being refused."""
    s = synth.synth(97, [1800, 300, 45], num_genes=200, txp_per_gene=3, dup=0.4, cross=0.3, umi_err=0.03)
    b, off = s.encode()
    b = np.asarray(b, np.uint8)
    ends = list(off[1:]) + [len(b)]
    parts, offs, pos = [], [], 0
    for i, (a, e) in enumerate(zip(off, ends)):
        pad = i + 1   # 1, 2, 3 bytes of padding in front of the chunks
        parts.append(np.zeros(pad, np.uint8)); pos += pad
        offs.append(pos)
        parts.append(b[int(a):int(e)]); pos += int(e) - int(a)
    b2 = np.concatenate(parts)
    cfg = cfg_for(s, "parsimony")
    got, want = run_both(oracle, cfg, s.tid_to_gid, b2, np.asarray(offs, np.uint64))
    assert_same_result(got, want)
    assert got.val.sum() > 0


@pytest.mark.parametrize("res", ["parsimony", "cr-like"])
def test_widened_batch_cut_into_many_ranges(oracle, monkeypatch, res):
    """The widened copy is made range by range (each range's kernels wait for ITS bytes when the input is piped over
    PCIe): force a dozen ranges and the pipelined upload on a 2-byte-UMI batch."""
    monkeypatch.setenv("AFQ_TEST_RANGE_BYTES", str(1 << 20))
    sizes = [900, 700, 650, 600, 500, 450, 400, 300, 250, 200, 150, 120, 110, 90, 60, 30, 8, 2]
    s = synth.synth(123, sizes, num_genes=300, txp_per_gene=3, umi_len=8, dup=0.4, cross=0.3, umi_err=0.03, max_extra_na=6)
    cells = _cells_of(s, lambda ci: 100000 + 7 * ci)
    b, off = rad.encode_cells(cells, 4, 2)
    cfg = cfg_for(s, res, bc_bytes=4, umi_bytes=2, umi_len=8)
    q = pkg.Quantifier(cfg, s.tid_to_gid)
    try:
        got = q.quant_chunks(b, off)
        st = q.batch_stats()
    finally:
        q.close()
    want = oracle.quant(cfg, s.tid_to_gid, b, off)
    assert_same_result(got, want)
    assert st["n_records"] == sum(sizes) and st["n_fallback_cells"] == 0, st


def _umi_neighbours(u, length):
    """All UMIs one base away from u (2 bits per base)."""
    out = []
    for pos in range(length):
        b = (u >> (2 * pos)) & 3
        for d in (1, 2, 3):
            out.append((u & ~(3 << (2 * pos))) | (((b + d) & 3) << (2 * pos)))
    return out


@pytest.mark.parametrize("usa", [False, True])
@pytest.mark.parametrize("res", ["parsimony-em", "parsimony-gene-em", "parsimony"])
def test_classes_of_more_than_64_genes(oracle, res, usa):
    """A read that hits a large gene family gives a molecule whose gene label has more entries than the device carries in
    registers (64).  Without an EM such a molecule is dropped like any multi-gene one; with an EM it is a class of the
    cell and has to come out whole - from every place a molecule is emitted: lone vertices, two-vertex components, the
    eight-to-a-wave cover (3..8 vertices), the wave cover (9..64), the workgroup cover (> 64) and, in the last cell,
    the winner-take-all fallback above --large-graph-thresh with its tie sets."""
    L = 12
    n_genes = 220
    t2g, num_gene_ids, num_rows = synth.make_t2g(n_genes, 2, usa)
    n_spliced = 2 * n_genes
    rng = np.random.default_rng(5 + usa)
    wide1 = sorted(int(2 * g + rng.integers(0, 2)) for g in rng.choice(n_genes, 90, replace=False))          # 90 genes
    wide2 = sorted(set(wide1[:80]) | {int(2 * g) for g in rng.choice(n_genes, 30, replace=False)})            # overlaps wide1 in >= 80
    if usa:   # a few unspliced refs too: S/U pairs of one gene inside a wide class
        wide1 = sorted(set(wide1) | {n_spliced + int(g) for g in rng.choice(n_genes, 10, replace=False)})
    short = lambda: sorted(int(x) for x in rng.choice(n_spliced, size=int(rng.integers(1, 4)), replace=False))
    def far_umis(k):
        return [int(x) for x in rng.choice(1 << (2 * L), size=k, replace=False)]
    def star(center, n_nb, lab, reads_center=6):
        r = [(center, lab)] * reads_center
        for u in _umi_neighbours(center, L)[:n_nb]:
            r.append((u, lab))
        return r
    cells = []
    # lone vertices and pairs
    reads = [(u, wide1) for u in far_umis(5)] + [(u, wide2) for u in far_umis(3)]
    c = far_umis(2)
    reads += star(c[0], 1, wide1) + [(c[1], wide1), (_umi_neighbours(c[1], L)[7], wide2)]
    reads += [(u, short()) for u in far_umis(150)]
    cells.append((11, reads))
    # 3..8 and 9..64 vertices
    c = far_umis(4)
    reads = star(c[0], 4, wide1) + star(c[1], 6, wide2) + star(c[2], 20, wide1) + star(c[3], 30, wide2, reads_center=1)
    reads += [(u, short()) for u in far_umis(150)]
    cells.append((12, reads))
    # > 64 vertices: a centre, its 36 neighbours and the neighbours of two of them
    c = far_umis(1)[0]
    nb = _umi_neighbours(c, L)
    us = {c, *nb, *_umi_neighbours(nb[0], L), *_umi_neighbours(nb[20], L)}
    reads = [(u, wide1) for u in sorted(us)] + [(c, wide1)] * 5 + [(u, short()) for u in far_umis(150)]
    cells.append((13, reads))
    b, off = rad.encode_cells(cells, 4, 4)
    cfg = pkg.WorkerConfig.for_resolution(res, usa_mode=usa, num_genes=num_gene_ids, num_rows=num_rows, small_thresh=0)
    got, want = run_both(oracle, cfg, t2g, b, off)
    assert_same_result(got, want)
    assert got.val.sum() > 0
    # the same cells with every component above three vertices sent down the winner-take-all fallback: ties among > 64 genes
    cfg2 = pkg.WorkerConfig.for_resolution(res, usa_mode=usa, num_genes=num_gene_ids, num_rows=num_rows, small_thresh=0, large_graph_thresh=3)
    got2, want2 = run_both(oracle, cfg2, t2g, b, off)
    assert_same_result(got2, want2)
    assert (got2.flags & pkg._abi.CELL_ALT_RES).any()


@pytest.mark.parametrize("res", ["parsimony", "parsimony-gene-em"])
def test_label_hash_collision_is_rehashed_not_refused(oracle, monkeypatch, pug_route, res):
    """Labels of three or more ids are keyed by a 62-bit hash (labels of one or two carry the ids themselves).  With the
    first try's hashes cut to 3 bits, different labels of a cell share keys for certain: the device notices (equal keys,
    different lists), and the range is decoded again under another hash function - the rows are the oracle's, nothing is
    refused (the reference keys by the list itself: eq_class.rs:859-903)."""
    s = synth.synth(41, [3000, 800, 300, 150], num_genes=60, txp_per_gene=4, dup=0.4, cross=0.6, umi_err=0.03, max_extra_na=6)
    b, off = s.encode()
    cfg = cfg_for(s, res, small_thresh=0)
    want = oracle.quant(cfg, s.tid_to_gid, b, off)
    monkeypatch.setenv("AFQ_TEST_LABEL_HASH_BITS", "3")
    q = pkg.Quantifier(cfg, s.tid_to_gid)
    try:
        got = q.quant_chunks(b, off)
        assert q.label_rehash_count() >= 1, "the cut hashes were meant to collide"
    finally:
        q.close()
    assert_same_result(got, want, what=res)


@pytest.mark.parametrize("res", ["parsimony", "parsimony-em"])
def test_a_graph_that_outgrows_the_pool_is_run_again_not_refused(oracle, monkeypatch, pug_route, res):
    """The per-cell graphs live in a pool sized by the range's reads.  Short UMIs (7 nt: 16 384 of them for 120 000 reads) give
    every vertex many same-UMI and one-base neighbours, and the cell's pairs, components and match lists outgrow a pool planned
    for sparse graphs: the range is run again with four times the pool (afq_pool_regrow_count) instead of ending in
    AFQ_ERR_OOM - found by tests/extended_fuzz.py; the reference allocates per graph (pugutils.rs:65-267)."""
    # (120 000 reads: the pair list alone is several times the 2 words per read that AFQ_TEST_POOL_WORDS=12 leaves the pool)
    s = synth.synth(5012, [900, 120000, 300], num_genes=17, txp_per_gene=3, usa=True, dup=0.5, cross=0.9, umi_err=0.02, max_extra_na=6, umi_len=7)
    b, off = s.encode()
    cfg = cfg_for(s, res, small_thresh=0)
    want = oracle.quant(cfg, s.tid_to_gid, b, off)
    monkeypatch.setenv("AFQ_TEST_POOL_WORDS", "12")
    q = pkg.Quantifier(cfg, s.tid_to_gid)
    try:
        got = q.quant_chunks(b, off)
        assert q.pool_regrow_count() >= 1, "the small first pool was meant to run out"
    finally:
        q.close()
    assert_same_result(got, want, what=res)


def test_parsimony_cell_of_more_than_2_pow_20_reads(oracle, pug_route):
    """The reference has no limit on a cell's reads (quant.rs:733-757).  The one-workgroup kernel numbers a cell's vertices
    in 20 bits and refuses cells of 2^20 reads or more; the phase kernels take them (up to 2^22): 1.15 M reads of one cell,
    8 192 UMI partitions, a class table of a few MiB out of the pool - rows bit-exact against the oracle."""
    if pug_route not in ("phase-kernels", "graph-per-cell"):
        pytest.skip("the one-workgroup kernel refuses cells of 2^20 reads or more (AFQ_ERR_UNSUPPORTED), by design")
    import importlib

    sn = importlib.import_module("alevin-fry_amd.synth_native")
    d = sn.generate(seed=12, n_cells=1, median_reads=1.15e6, sigma=0.0, num_genes=36601, ref_count=199138, umi_err=0.01)
    assert d.cell_nrec[0] > (1 << 20)
    cfg = pkg.WorkerConfig.for_resolution("parsimony", num_genes=d.num_genes, num_rows=d.num_rows, umi_len=12)
    q = pkg.Quantifier(cfg, d.tid_to_gid)
    try:
        got = q.quant_chunks(d.data, d.chunk_off)
    finally:
        q.close()
    want = oracle.quant(cfg, d.tid_to_gid, d.data, d.chunk_off)
    assert_same_result(got, want)


@pytest.mark.parametrize("thresh", [4096, 1000])
@pytest.mark.parametrize("res,usa", [("parsimony", False), ("parsimony-em", True)])
def test_components_of_65_to_4096_vertices_stay_with_the_phase_kernels(oracle, monkeypatch, pug_route, res, usa, thresh):
    """Short UMIs in a cell of a few thousand reads chain hundreds of vertices into one component.  Up to 4096 vertices (and
    --large-graph-thresh) the phase kernels cover such a component themselves, a workgroup to it (cover_big in
    csrc/afq_pug_common.h); until round 4 its cell went back to the one-workgroup kernel.  afq_mono_cell_count says which
    kernel had the cells; AFQ_TEST_P2_MAX_COMP=64 brings the old routing back - same rows (pugutils.rs:1004-1200).  Under the default
    --large-graph-thresh of 1000 the two largest components are resolved winner-take-all instead (pugutils.rs:916-982) - by the
    phase kernels as well (cover_large in csrc/afq_pug2.hip), and the cells are flagged."""
    # (components of 2499, 1117, 132, 70 vertices without the USA labels, of 2552, 1119, 192 with them)
    s = synth.synth(6107, [3000, 1500, 700, 300, 90], num_genes=4, txp_per_gene=3, usa=usa, dup=0.35, cross=0.5, umi_err=0.05, max_extra_na=5, umi_len=5)
    b, off = s.encode()
    cfg = cfg_for(s, res, small_thresh=0, large_graph_thresh=thresh)
    want = oracle.quant(cfg, s.tid_to_gid, b, off)
    assert (np.asarray(want.flags[:2]) != 0).all() == (thresh == 1000), "the two big cells take the fallback under the default threshold only"

    def run():
        q = pkg.Quantifier(cfg, s.tid_to_gid)
        try:
            return q.quant_chunks(b, off), q.mono_cell_count()
        finally:
            q.close()

    got, n_mono = run()
    assert_same_result(got, want, what=res)
    if pug_route in ("phase-kernels", "cover-1024", "graph-per-cell", "graph-per-cell-1024"):
        assert n_mono == 0, "no cell should have needed the one-workgroup kernel"
        monkeypatch.setenv("AFQ_TEST_P2_MAX_COMP", "64")
        got64, n_mono64 = run()
        assert n_mono64 >= 1, "the cells were meant to hold components of more than 64 vertices"
        assert_same_result(got64, want, what=res + ", components over 64 vertices handed back")
    elif pug_route == "one-workgroup":
        assert n_mono == 5


@pytest.mark.parametrize("res", ["parsimony", "parsimony-em"])
def test_more_pairs_than_reads_stay_with_the_phase_kernels(oracle, pug_route, res):
    """Reads of one molecule that hit different members of a gene family: one UMI under a dozen labels that all overlap - every
    two of them are a pair (66 pairs for 12 reads), and the partition's pair list outgrows its own slots, one per read.  The
    search then takes the list's slots out of the pool in a second pass (k_p2_search_over); until round 4 the whole cell went to
    the one-workgroup kernel (has_edge at distance 0: pugutils.rs:76-99)."""
    rng = np.random.default_rng(991)
    n_txp = 64                                  # eight families of eight transcripts, a gene per transcript pair
    t2g = (np.arange(n_txp) // 2).astype(np.uint32)
    cells = []
    for ci, n_umi in enumerate([400, 120, 30, 5]):
        reads = []
        umis = rng.choice(1 << 20, size=n_umi, replace=False)
        for u in umis:
            fam = int(rng.integers(0, 8)) * 8
            anchor = fam + int(rng.integers(0, 8))
            for _ in range(int(rng.integers(6, 14))):
                extra = rng.choice(8, size=int(rng.integers(0, 5)), replace=False)
                lab = sorted({anchor} | {fam + int(e) for e in extra})
                umi = int(u)
                if rng.random() < 0.1:
                    umi ^= 1 << (2 * int(rng.integers(0, 10)))   # a one-base neighbour now and then
                reads.append((umi, lab))
        order = rng.permutation(len(reads))
        cells.append((1000 + ci, [reads[i] for i in order]))
    b, off = rad.encode_cells(cells, 4, 4)
    cfg = pkg.WorkerConfig.for_resolution(res, num_genes=n_txp // 2, num_rows=n_txp // 2, small_thresh=0)
    want = oracle.quant(cfg, t2g, b, off)
    q = pkg.Quantifier(cfg, t2g)
    try:
        got = q.quant_chunks(b, off)
        n_mono = q.mono_cell_count()
    finally:
        q.close()
    assert_same_result(got, want, what=res)
    if pug_route in ("phase-kernels", "cover-1024", "graph-per-cell", "graph-per-cell-1024"):
        assert n_mono == 0, "no cell should have needed the one-workgroup kernel"


def _grid_component(n, t, hi):
    """n vertices (UMI, label, reads) of ONE component: the first n points of the 4 x 4 x 4 x 4 grid over four UMI bases (point i is
    one base away from the point with its highest non-zero digit cleared, so every prefix is connected); labels of 1, 2, 6 and 2
    refs that all hold transcript t (every 1-Hamming pair is an edge candidate), 1..3 reads (edges in one direction only)."""
    out = []
    for i in range(n):
        umi = hi | (i & 3) | (((i >> 2) & 3) << 4) | (((i >> 4) & 3) << 10) | (((i >> 6) & 3) << 20)
        lab = ([t], [t, t + 2], [t, t + 1, t + 2, t + 3, t + 4, t + 5], [t, t + 4])[i % 4]
        out.append((umi, lab, 1 + (i * 7) % 3))
    return out


@pytest.mark.parametrize("thresh", [0, 60, 8])
@pytest.mark.parametrize("res,usa", [("parsimony", False), ("parsimony-em", True)])
def test_component_sizes_at_every_boundary_of_the_flat_build(oracle, res, usa, thresh):
    """csrc/afq_pugflat.hip sorts a range's components by size into the covers' lists: 2 vertices (k_pc_pairs), 3..4 (a lane each,
    k_pc_lane4 - the ones with a label of more than four refs go to its eight-lane list), 5..8 (k_pc_tiny8), 9..64 (k_pc_mid, a wave
    each), more than 64 (the whole cell to the per-cell kernels) and, above --large-graph-thresh, the winner-take-all fallback.
    One component of every size on both sides of each boundary, alone in a small cell, all of them in one cell, and all of them in a
    cell of three tiles whose records are shuffled (class first appearance = the tie-break order); --large-graph-thresh at its
    default, at 60 (the 63..100-vertex components fall back) and at 8."""
    rng = np.random.default_rng(77)
    sizes = [2, 3, 4, 5, 8, 9, 16, 63, 64, 65, 100]
    comps = [_grid_component(n, 8 * k, (k + 1) << 26 if k % 2 else 0) for k, n in enumerate(sizes)]   # (a transcript block of its own per component: no edges between them)
    n_t = 8 * len(sizes) + 64
    def recs_of(cs, filler=0):
        r = [(u, lab) for c in cs for u, lab, reads in c for _ in range(reads)]
        r += [(int(rng.integers(0, 1 << 24)), [8 * len(sizes) + int(rng.integers(0, 64))]) for _ in range(filler)]
        return [r[j] for j in rng.permutation(len(r))]
    cells = [(100 + k, recs_of([c])) for k, c in enumerate(comps)]
    cells.append((300, recs_of(comps)))
    cells.append((301, recs_of(comps[:9])))           # (no component beyond 64 vertices: the cell stays with the flat build)
    cells.append((302, recs_of(comps, filler=9000)))
    cells.append((303, recs_of(comps[:9], filler=9000)))
    b, off = rad.encode_cells(cells, 4, 4)
    G = n_t // 2
    t2g = (np.arange(n_t, dtype=np.uint32) // 2 * 2 + (np.arange(n_t, dtype=np.uint32) & 1)) if usa else np.arange(n_t, dtype=np.uint32) // 2
    kw = dict(large_graph_thresh=thresh) if thresh else {}
    cfg = pkg.WorkerConfig.for_resolution(res, usa_mode=usa, num_genes=2 * G if usa else G, num_rows=3 * G if usa else G, small_thresh=0, **kw)
    got, want = run_both(oracle, cfg, t2g, b, off)
    if thresh:
        assert (want.flags & pkg._abi.CELL_ALT_RES).any()
    assert_same_result(got, want, what=f"{res} usa={usa} thresh={thresh}")


@pytest.mark.parametrize("res", ["parsimony", "parsimony-em"])
def test_edge_directions_from_read_counts_beyond_the_flag_bytes_127(oracle, res):
    """The flat build's cover records carry (UMI, reads) and the covers decide has_edge themselves (x -> y one base apart iff
    reads(y) < 2 reads(x), pugutils.rs:88-97); k_p2_part hands the reads over in seven bits of the vertex's flag byte, 127 standing
    for "127 or more: ask the count array".  Paths a - b - c with reads (2 m, m, 2 m) need two molecules (b reaches neither end),
    (2 m - 1, m, 2 m - 1) one; m on both sides of 127 and of 64, beside a 300-read vertex with 1000- and 599-read neighbours."""
    cells, k = [], 0
    for m, big in [(63, 126), (63, 125), (64, 128), (64, 127), (126, 252), (126, 251), (127, 254), (127, 253), (128, 256), (128, 255), (200, 400), (200, 399), (300, 1000), (300, 599)]:
        base = (k + 1) << 8
        a, c = base ^ 1, base ^ (1 << 2)
        recs = [(a, [2 * k])] * big + [(base, [2 * k, 2 * k + 1])] * m + [(c, [2 * k])] * big
        cells.append((500 + k, recs))
        k += 1
    cells.append((600, [r for _, rs in cells for r in rs]))   # all of them in one cell
    b, off = rad.encode_cells(cells, 4, 4)
    n_t = 2 * k
    cfg = pkg.WorkerConfig.for_resolution(res, num_genes=n_t, num_rows=n_t, small_thresh=0)
    got, want = run_both(oracle, cfg, np.arange(n_t, dtype=np.uint32), b, off)
    per_cell = [sum(v for _, v in row) for row in rows_of(want)]
    assert per_cell[0] != per_cell[1] and per_cell[6] != per_cell[7] and per_cell[12] != per_cell[13]   # (the direction rule decides the molecule count)
    assert_same_result(got, want, what=res)


@pytest.mark.parametrize("ref_base", [0, 600000])
@pytest.mark.parametrize("res", ["parsimony", "parsimony-em"])
def test_candidate_pairs_whose_labels_share_no_ref_are_cleared(oracle, res, ref_base):
    """k_p2_search (late round 6) writes a CANDIDATE pair for every two vertices whose UMIs are as has_edge wants them and whose
    19-bit label signatures share a bit - ref t sets bit t % 19 - and k_p2_check (a thread per candidate) compares the labels and
    clears the candidates that share no ref (pugutils.rs:187-204).  Cells made of such near-misses: refs 19 apart under UMIs one base
    apart or equal, one / two / five refs a label (inline and hashed keys), beside true edges; a vertex all of whose candidates are
    cleared must come out as the lone molecule it is, one with a true and a false candidate as a component of two.
    ref_base 600000: the same with ref ids beyond 19 bits (a transcriptome of more than half a million refs)."""
    rng = np.random.default_rng(5)
    cells = []
    for k in range(40):
        recs = []
        for q in range(60):
            u = int(rng.integers(0, 1 << 24))
            t = int(rng.integers(0, 19))
            kind = q % 6
            if kind == 0:     # one base apart, single refs 19 apart: a candidate, no edge
                recs += [(u, [t])] * 2 + [(u ^ 1, [t + 19])]
            elif kind == 1:   # the same UMI under two labels 19 apart
                recs += [(u, [t]), (u, [t + 38])]
            elif kind == 2:   # two refs a label, signatures overlap in both bits, no ref in common
                recs += [(u, [t, t + 57])] * 2 + [(u ^ (2 << 6), [t + 19, t + 38])]
            elif kind == 3:   # hashed labels (five refs) that share signature bits only
                recs += [(u, [t, t + 19, t + 38, t + 57, t + 76])] + [(u ^ (3 << 10), [t + 95, t + 114, t + 133, t + 152, t + 171])] * 2
            elif kind == 4:   # a true edge beside a false candidate at the same vertex
                recs += [(u, [t])] * 3 + [(u ^ 1, [t])] + [(u ^ (1 << 4), [t + 19])]
            else:             # hashed against inline: one shared ref / none
                recs += [(u, [t, t + 19, t + 38])] + [(u ^ 2, [t + 19])] + [(u ^ (1 << 8), [t + 57])]
        recs = [(u, [ref_base + t for t in lab]) for u, lab in (recs[j] for j in rng.permutation(len(recs)))]
        cells.append((900 + k, recs))
    b, off = rad.encode_cells(cells, 4, 4)
    n_g = 200
    t2g = np.zeros(ref_base + n_g, dtype=np.uint32)
    t2g[ref_base:] = np.arange(n_g, dtype=np.uint32)
    cfg = pkg.WorkerConfig.for_resolution(res, num_genes=n_g, num_rows=n_g, small_thresh=0)
    got, want = run_both(oracle, cfg, t2g, b, off)
    assert_same_result(got, want, what=f"{res} ref_base={ref_base}")
