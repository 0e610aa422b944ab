"""Full-size, size-independent properties (BASELINE configs[1] shape: 11 000 cells, ~4e8 reads).  The oracle cannot
run this size inside a unit test (bench.py's cpu_baseline leg does compare every cell), so these check what must
hold at any size: cells are independent (any way of splitting the batch gives the same rows), the result does
not depend on how the library cuts ranges or which walk-free decoder it picks, a second run is identical, rows
are sorted and positive, per-cell mass never exceeds the cell's reads, and a cell's row does not depend on its
neighbours (a sample of cells re-quantified alone - and against the oracle)."""
import hashlib

import numpy as np
import pytest

from util import assert_em_within_the_reference_envelope, assert_same_result, pkg

pytestmark = pytest.mark.gpu
sn = pytest.importorskip("importlib").import_module("alevin-fry_amd.synth_native")


def _digest(r):
    h = hashlib.sha256()
    for a in (r.cell_ptr, r.gene, r.val.view(np.uint32), r.bc, r.nrec, r.flags):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


@pytest.fixture(scope="module")
def big():
    rad = sn.generate(seed=2, n_cells=11000, median_reads=30000.0, sigma=0.6, num_genes=36601, txp_per_gene=5)
    cfg = pkg.WorkerConfig.for_resolution("cr-like", num_genes=rad.num_genes, num_rows=rad.num_rows)
    q = pkg.Quantifier(cfg, rad.tid_to_gid)
    whole = q.quant_chunks(rad.data, rad.chunk_off)
    yield rad, cfg, q, whole
    q.close()


def test_rows_are_well_formed(big):
    rad, cfg, q, r = big
    assert r.n_cells == 11000 and np.array_equal(r.nrec, rad.cell_nrec)
    ptr = r.cell_ptr.astype(np.int64)
    assert ptr[0] == 0 and (np.diff(ptr) >= 0).all() and ptr[-1] == len(r.gene) == len(r.val)
    assert (r.val > 0).all() and (r.val == np.floor(r.val)).all()  # cr-like counts are whole molecules
    nonempty = np.diff(ptr) > 0
    first = np.zeros(len(r.gene), bool)
    first[ptr[:-1][nonempty]] = True            # first entry of every non-empty row
    assert (np.diff(r.gene.astype(np.int64))[~first[1:]] > 0).all()  # columns strictly ascending inside a row
    assert (r.gene < rad.num_rows).all()
    mass = np.add.reduceat(r.val.astype(np.float64), ptr[:-1][nonempty])
    assert (mass <= rad.cell_nrec[nonempty]).all()


def test_second_run_is_identical(big):
    rad, cfg, q, r = big
    assert _digest(q.quant_chunks(rad.data, rad.chunk_off)) == _digest(r)


def test_any_split_of_the_batch_gives_the_same_rows(big):
    rad, cfg, q, r = big
    cuts = [0, 17, 4000, 4001, 9500, 11000]  # ragged pieces, incl. a one-cell piece
    parts = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        parts.append(q.quant_chunks(rad.data, rad.chunk_off[a:b], first_cell_index=a))
    import importlib

    got = importlib.import_module("alevin-fry_amd.shard").concat_results(parts)
    assert_same_result(got, r)


def test_range_planning_and_decoder_choice_do_not_matter(big, monkeypatch):
    rad, cfg, q, r = big
    monkeypatch.setenv("AFQ_TEST_RANGE_BYTES", "7e8")   # ~15 ranges instead of 5
    monkeypatch.setenv("AFQ_TEST_DECODE", "keys")       # lane-per-dword decoder instead of lane-per-record
    assert _digest(q.quant_chunks(rad.data, rad.chunk_off)) == _digest(r)


def test_sampled_cells_alone_and_against_the_oracle(big, oracle):
    rad, cfg, q, r = big
    idx = np.concatenate(([0, 1, 2], np.arange(50, 11000, 997), [10998, 10999]))
    alone = q.quant_chunks(rad.data, rad.chunk_off[idx])
    want = oracle.quant(cfg, rad.tid_to_gid, rad.data, rad.chunk_off[idx], n_threads=16)
    assert_same_result(alone, want)
    for j, ci in enumerate(idx):
        g0, v0 = r.row(int(ci))
        g1, v1 = alone.row(j)
        assert np.array_equal(g0, g1) and np.array_equal(v0, v1), f"cell {ci} depends on its neighbours"


@pytest.fixture(scope="module")
def big_pug():
    rad = sn.generate(seed=7, n_cells=1500, median_reads=30000.0, sigma=0.6, num_genes=36601, txp_per_gene=5, usa=True, umi_err=0.01)
    cfg = pkg.WorkerConfig.for_resolution("parsimony-em", usa_mode=True, num_genes=rad.num_genes, num_rows=rad.num_rows, umi_len=12)
    q = pkg.Quantifier(cfg, rad.tid_to_gid)
    whole = q.quant_chunks(rad.data, rad.chunk_off)
    yield rad, cfg, q, whole
    q.close()


def test_parsimony_em_is_deterministic_and_split_invariant(big_pug):
    """BASELINE configs[2] shape (USA, parsimony-em) on 1 500 PBMC-sized cells: the parsimony kernel's atomics
    (edge lists, candidate lists) must not leak scheduling order into the counts."""
    import importlib

    rad, cfg, q, r = big_pug
    assert _digest(q.quant_chunks(rad.data, rad.chunk_off)) == _digest(r)
    parts = [q.quant_chunks(rad.data, rad.chunk_off[a:b], first_cell_index=a) for a, b in ((0, 3), (3, 700), (700, 1500))]
    assert_same_result(importlib.import_module("alevin-fry_amd.shard").concat_results(parts), r)
    ptr = r.cell_ptr.astype(np.int64)
    nonempty = np.diff(ptr) > 0
    mass = np.add.reduceat(r.val.astype(np.float64), ptr[:-1][nonempty])
    assert (r.val > 0).all() and (mass <= rad.cell_nrec[nonempty] * (1 + 1e-6)).all()


def test_parsimony_em_sampled_cells_against_the_oracle(big_pug, oracle):
    rad, cfg, q, r = big_pug
    import os

    idx = np.arange(3, 1500, 7)   # 214 cells spread over the whole size range (largest-first order), multi-threaded oracle
    want = oracle.quant(cfg, rad.tid_to_gid, rad.data, rad.chunk_off[idx], n_threads=os.cpu_count() or 8)
    for j, ci in enumerate(idx):
        g0, v0 = r.row(int(ci))
        g1, v1 = want.row(j)
        assert np.array_equal(g0, g1), f"cell {ci}: columns differ"
        np.testing.assert_allclose(v0, v1, rtol=1e-4, atol=0, err_msg=f"cell {ci}")  # the tolerance north_star states for EM
        assert np.array_equal(v0.view(np.uint32), v1.view(np.uint32)), f"cell {ci}: not bit-identical to the oracle's canonical order"


def test_parsimony_em_sampled_cells_against_the_reference_arithmetic(big_pug, oracle_module):
    """What counts toward north_star's 1e-4: the same 214 PBMC-sized cells against the oracle in the REFERENCE's arithmetic
    (f32 sums in canonical class order - not the fixed-point restatement of the device, which the test above uses and which is a
    self-check only).  Floor crossings and entries beyond 1e-4 are counted and held to what the reference's own (unpinned)
    summation order produces on these very cells: three shuffled class orders of the oracle, measured here."""
    rad, cfg, q, r = big_pug
    idx = np.arange(3, 1500, 7)
    out = assert_em_within_the_reference_envelope(None, oracle_module, cfg, rad.tid_to_gid, rad.data, rad.chunk_off[idx],
                                                  rows_of_got=[r.row(int(ci)) for ci in idx], what="configs[2]-shaped cells")
    assert out["entries"] > 100000
    print("parsimony-em at size vs the reference arithmetic:", out)


def test_tailed_parsimony_em_sample_against_the_reference_arithmetic(oracle_module):
    """The same on the label-tail model (labels of up to 64 refs over gene families: many more multi-gene classes per cell,
    which is where a fixed-point sum and an f32 sum could drift apart)."""
    rad = sn.generate(seed=9, n_cells=360, median_reads=30000.0, sigma=0.6, num_genes=36601, txp_per_gene=5, usa=True, umi_err=0.01,
                      tail=0.65, tail_max=64, family=8)
    cfg = pkg.WorkerConfig.for_resolution("parsimony-em", usa_mode=True, num_genes=rad.num_genes, num_rows=rad.num_rows, umi_len=12)
    q = pkg.Quantifier(cfg, rad.tid_to_gid)
    try:
        r = q.quant_chunks(rad.data, rad.chunk_off)
    finally:
        q.close()
    idx = np.arange(2, 360, 5)
    out = assert_em_within_the_reference_envelope(None, oracle_module, cfg, rad.tid_to_gid, rad.data, rad.chunk_off[idx],
                                                  rows_of_got=[r.row(int(ci)) for ci in idx], what="tailed cells")
    assert out["entries"] > 30000
    print("tailed parsimony-em at size vs the reference arithmetic:", out)


def test_crlike_em_on_the_largest_cells_against_the_oracle(big_pug, oracle):
    """The EM rounds have three tiers by cell size (all state on chip / abundances + denominators in LDS / everything
    streamed); the 200 k-read cells at the head of the batch exercise the last two.  cr-like-em so that the oracle
    (no parsimony graph to build) finishes in seconds."""
    rad, _, _, _ = big_pug
    cfg = pkg.WorkerConfig.for_resolution("cr-like-em", usa_mode=True, num_genes=rad.num_genes, num_rows=rad.num_rows)
    idx = np.array([0, 1, 5, 60, 300, 900])
    q = pkg.Quantifier(cfg, rad.tid_to_gid)
    try:
        got = q.quant_chunks(rad.data, rad.chunk_off[idx])
    finally:
        q.close()
    want = oracle.quant(cfg, rad.tid_to_gid, rad.data, rad.chunk_off[idx], n_threads=8)
    assert (np.diff(got.cell_ptr.astype(np.int64)) > 0).all()
    assert_same_result(got, want)


def test_configs3_shard_sample_against_oracle(oracle):
    """BASELINE configs[3] at the size one GPU holds of it: 125 000 cells x ~2*10^4 reads (2.5 G reads, 43.7 GB) generated
    in HBM exactly as bench.py's configs3 leg does, cr-like.  >= 500 cells spread over the shard bit-exact against the
    oracle, and the shard cut in two by shard.shard_ranges and quantified by two contexts == the one-context rows
    (cells are independent: src/quant.rs:880, 937, 967)."""
    import importlib
    import os

    shard = importlib.import_module("alevin-fry_amd.shard")
    n = 125000
    sigma = 0.6
    p = sn.params(seed=4, n_cells=n, median_reads=20000.0 / float(np.exp(sigma * sigma / 2)), sigma=sigma, num_genes=36601, ref_count=199138)
    sizes = sn.cell_sizes(p)
    rad = sn.generate_device(device=0, p=p, sizes=sizes, cell_range=(0, n))
    cfg = pkg.WorkerConfig.for_resolution("cr-like", num_genes=rad.num_genes, num_rows=rad.num_rows, umi_len=12)
    q = pkg.Quantifier(cfg, rad.tid_to_gid)
    q2 = None
    try:
        q.submit_device(rad.d_ptr, rad.n_bytes, rad.chunk_off, 0)
        whole = q.collect()
        assert whole.n_cells == n and np.array_equal(whole.nrec, rad.cell_nrec)
        idx = np.arange(11, n, 211)                      # 593 cells over the whole size range (largest first)
        assert len(idx) >= 500
        data, offs = rad.read_cells(idx)
        want = oracle.quant(cfg, rad.tid_to_gid, data, offs, n_threads=os.cpu_count() or 8)
        for j, ci in enumerate(idx):
            g0, v0 = whole.row(int(ci))
            g1, v1 = want.row(j)
            assert np.array_equal(g0, g1) and np.array_equal(v0.view(np.uint32), v1.view(np.uint32)), f"cell {ci} differs from the oracle"
        # two contexts over the byte-balanced halves == one context
        (a0, a1), (b0, b1) = shard.shard_ranges(rad.chunk_nbytes(), 2)
        assert a0 == 0 and a1 == b0 and b1 == n and 0 < a1 < n
        digest_whole = _digest(whole)
        whole = None
        q2 = pkg.Quantifier(cfg, rad.tid_to_gid)
        parts = []
        for qq, (c0, c1) in ((q, (a0, a1)), (q2, (b0, b1))):
            base = int(rad.chunk_off[c0])
            end = int(rad.chunk_off[c1]) if c1 < n else rad.n_bytes
            qq.submit_device(rad.d_ptr + base, end - base, rad.chunk_off[c0:c1] - np.uint64(base), c0)
            parts.append(qq.collect())
        assert _digest(shard.concat_results(parts)) == digest_whole
    finally:
        q.close()
        if q2 is not None:
            q2.close()
        rad.free()
