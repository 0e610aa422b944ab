"""GPU parity sweep: seeded random workloads x every resolution / mode switch, device vs oracle, bit for bit."""
import numpy as np
import pytest

from util import assert_same_result, pkg

synth = pkg.synth
pytestmark = pytest.mark.gpu

RES = ["trivial", "cr-like", "cr-like-em", "parsimony", "parsimony-em", "parsimony-gene", "parsimony-gene-em"]


@pytest.mark.parametrize("seed", range(36))
def test_random_workloads_match_the_oracle(oracle, seed, monkeypatch):
    """Cell sizes from one read to a few thousand (tiny path, single- and multi-bucket cells), few or many genes, heavy or
    light duplication, cross-gene multi-mappers, UMI errors, long reference lists; the switches drawn along: USA,
    --sa-model, --small-thresh 0, exact-UMI parsimony, uniform EM start, -d, -b (both summaries)."""
    rng = np.random.default_rng(1000 + seed)
    res = RES[seed % len(RES)]
    if res.startswith("parsimony") and (seed // len(RES)) % 2:   # every other parsimony workload: the per-cell graph kernel of rounds 3-5 for every cell (by default the range-wide flat build, csrc/afq_pugflat.hip), tied components set aside in every cell (k_p2_tied)
        monkeypatch.setenv("AFQ_TEST_P2_GRAPH", "cell")
        monkeypatch.setenv("AFQ_TEST_P2_DEFER_MIN", "0")
    usa = bool(rng.integers(0, 2))
    sizes = [int(x) for x in rng.choice([1, 2, 7, 40, 99, 100, 101, 250, 251, 600, 1500, 4000], size=int(rng.integers(3, 9)))]
    if seed % 5 == 0:
        sizes.append(int(rng.integers(8000, 20000)))
    s = synth.synth(2000 + seed, sizes, num_genes=int(rng.choice([3, 17, 120, 900])), txp_per_gene=int(rng.integers(1, 4)), usa=usa,
                    dup=float(rng.choice([0.0, 0.3, 0.7, 0.95])), cross=float(rng.choice([0.0, 0.3, 0.9])),
                    umi_err=float(rng.choice([0.0, 0.02, 0.2])), max_extra_na=int(rng.choice([0, 3, 12])),
                    zipf=float(rng.choice([0.0, 0.8])), umi_len=int(rng.choice([6, 12])))
    b, off = s.encode()
    em = res.endswith("em")
    kw = dict(small_thresh=int(rng.choice([0, 100])), em_init_uniform=bool(em and rng.integers(0, 2)))
    if usa and rng.integers(0, 2):
        kw["sa_model"] = "prefer-ambig"
    if res.startswith("parsimony") and rng.integers(0, 3) == 0:
        kw["pug_exact_umi"] = True
    if res.startswith("parsimony") and rng.integers(0, 3) == 0:
        kw["large_graph_thresh"] = int(rng.choice([2, 5, 50]))
    if em and rng.integers(0, 2):
        kw["dump_eq"] = True
    if em and rng.integers(0, 2):
        kw.update(num_bootstraps=int(rng.integers(1, 6)), summary_stat=bool(rng.integers(0, 2)), boot_seed=int(rng.integers(0, 2**40)))
    cfg = pkg.WorkerConfig.for_resolution(res, usa_mode=usa, num_genes=s.num_genes, num_rows=s.num_rows, umi_len=s.umi_len if rng.integers(0, 2) else 0, **kw)
    q = pkg.Quantifier(cfg, s.tid_to_gid)
    try:
        got = q.quant_chunks(b, off, first_cell_index=seed)
    finally:
        q.close()
    want = oracle.quant(cfg, s.tid_to_gid, b, off, first_cell_index=seed)
    assert_same_result(got, want, what=f"{res} usa={usa} {kw}")
    if kw.get("dump_eq"):
        for i in range(got.n_cells):
            assert got.eqclasses.cell(i) == want.eqclasses.cell(i), (res, i)
    if kw.get("num_bootstraps"):
        gb, wb = got.bootstraps, want.bootstraps
        assert np.array_equal(gb.mean_ptr, wb.mean_ptr) and np.array_equal(gb.mean_col, wb.mean_col) and np.array_equal(gb.var_col, wb.var_col)
        assert np.array_equal(gb.mean_val.view(np.uint32), wb.mean_val.view(np.uint32))
        assert np.array_equal(gb.var_val.view(np.uint32), wb.var_val.view(np.uint32))
