"""One-off extended parity sweep of the parsimony phase kernels (not collected by pytest: run it on a GPU box,
`python tests/extended_fuzz.py [n_seeds] [first_seed]`).  Bigger cells than tests/test_gpu_fuzz.py (several UMI partitions,
foreign-partition probes, pool-resident class tables), skewed and short UMIs, long labels; device rows == oracle rows."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from util import assert_same_result, pkg  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import oracle as ora  # noqa: E402

synth = pkg.synth


def one_tailed(seed):
    """Third family (seeds from 2000; from 3000: parsimony with AFQ_TEST_P2_LONE_COOP=2): the bench's label-tail model out of the native generator - reads of up to 64 alignments on
    gene families - through a decoder picked per seed (the planner's choice, lane per record, lane per dword with either way of
    finding a record's repeated genes) and every resolution."""
    import importlib

    sn = importlib.import_module("alevin-fry_amd.synth_native")
    rng = np.random.default_rng(99000 + seed)
    res = ["cr-like", "cr-like-em", "trivial", "parsimony", "parsimony-em", "cr-like", "cr-like-em"][seed % 7]
    usa = bool(rng.integers(0, 2))
    dec = [None, "recs", "keys", "keys", "keys"][int(rng.integers(0, 5))]
    dedup = ["hash", "scan"][int(rng.integers(0, 2))]
    coop = str(int(rng.integers(0, 3)))   # k_p2_lone: labels over four refs by their lane in scratch memory (0), by the wave (1), 5..8 refs by the lane in registers and 9..64 by the wave (2)
    if seed >= 3000:   # fourth family: parsimony only, the lone-vertex kernel's per-lane route for labels of 5..8 refs
        res, coop = ["parsimony", "parsimony-em"][seed % 2], "2"
    for k, v in (("AFQ_TEST_DECODE", dec), ("AFQ_TEST_DECODE_DEDUP", dedup), ("AFQ_TEST_P2_LONE_COOP", coop)):
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    d = sn.generate(seed=seed, n_cells=int(rng.choice([8, 40, 150])), median_reads=float(rng.choice([300.0, 2500.0, 9000.0])), sigma=float(rng.choice([0.5, 1.3])),
                    num_genes=int(rng.choice([40, 400, 3000])), txp_per_gene=int(rng.integers(1, 6)), usa=usa, umi_err=float(rng.choice([0.0, 0.02])),
                    tail=float(rng.choice([0.5, 0.65, 0.8, 0.9])), tail_max=int(rng.choice([8, 64])), family=int(rng.choice([4, 8, 16])))
    kw = dict(small_thresh=int(rng.choice([0, 100])))
    if usa and rng.integers(0, 2):
        kw["sa_model"] = "prefer-ambig"
    cfg = pkg.WorkerConfig.for_resolution(res, usa_mode=usa, num_genes=d.num_genes, num_rows=d.num_rows, umi_len=12, **kw)
    q = pkg.Quantifier(cfg, d.tid_to_gid)
    try:
        got = q.quant_chunks(d.data, d.chunk_off)
        rehash = q.label_rehash_count()
    finally:
        q.close()
        for k in ("AFQ_TEST_DECODE", "AFQ_TEST_DECODE_DEDUP", "AFQ_TEST_P2_LONE_COOP"):
            os.environ.pop(k, None)
    want = ora.quant(cfg, d.tid_to_gid, d.data, d.chunk_off, n_threads=os.cpu_count() or 1, em_arith="reference" if os.environ.get("AFQ_EM_ORDER") == "canonical" else "fixed")
    assert_same_result(got, want, what=f"seed {seed} {res} usa={usa} decoder={dec} dedup={dedup} lone_coop={coop} {kw} cells={len(d.chunk_off)}")
    return int(d.n_reads), rehash


def one(seed):
    if seed >= 2000:
        return one_tailed(seed)
    rng = np.random.default_rng(77000 + seed)
    res = ["parsimony", "parsimony-em"][seed % 2]
    if seed >= 1000:   # second family: every resolution (gene-level parsimony = the one-workgroup kernel, the cr-like routes, EM)
        res = ["trivial", "cr-like", "cr-like-em", "parsimony", "parsimony-em", "parsimony-gene", "parsimony-gene-em"][seed % 7]
    usa = bool(rng.integers(0, 2))
    sizes = [int(x) for x in rng.choice([1, 30, 300, 900, 2500, 6000, 12000], size=int(rng.integers(2, 6)))]
    sizes.append(int(rng.choice([15000, 30000, 45000, 70000])))
    if seed % 7 == 0:
        sizes.append(int(rng.integers(90000, 130000)))
    s = synth.synth(5000 + seed, sizes, num_genes=int(rng.choice([17, 300, 3000])), txp_per_gene=int(rng.integers(1, 5)), usa=usa,
                    dup=float(rng.choice([0.2, 0.5, 0.8])), cross=float(rng.choice([0.0, 0.3, 0.9])),
                    umi_err=float(rng.choice([0.0, 0.02, 0.1])), max_extra_na=int(rng.choice([0, 2, 6, 20])),
                    zipf=float(rng.choice([0.0, 0.8, 1.1])), umi_len=int(rng.choice([7, 8, 10, 12])))
    b, off = s.encode()
    kw = dict(small_thresh=int(rng.choice([0, 100])))
    if seed % 3 == 1:   # every third workload: tied components set aside in every cell (k_p2_tied), not only in those whose classes outgrow the LDS table
        os.environ["AFQ_TEST_P2_DEFER_MIN"] = "0"
        os.environ["AFQ_TEST_P2_GRAPH"] = "cell"   # ... through the per-cell graph kernel (by default: the range-wide flat build)
    else:
        os.environ.pop("AFQ_TEST_P2_DEFER_MIN", None)
        os.environ.pop("AFQ_TEST_P2_GRAPH", None)
    if rng.integers(0, 4) == 0:
        kw["pug_exact_umi"] = True
    if rng.integers(0, 4) == 0:
        kw["large_graph_thresh"] = int(rng.choice([5, 40, 200]))
    if usa and rng.integers(0, 2):
        kw["sa_model"] = "prefer-ambig"
    if res.endswith("em") and rng.integers(0, 2):
        kw["em_init_uniform"] = True
    cfg = pkg.WorkerConfig.for_resolution(res, usa_mode=usa, num_genes=s.num_genes, num_rows=s.num_rows, umi_len=s.umi_len if rng.integers(0, 2) else 0, **kw)
    q = pkg.Quantifier(cfg, s.tid_to_gid)
    try:
        got = q.quant_chunks(b, off)
        rehash = q.label_rehash_count()
        one.mono += q.mono_cell_count()
        one.regrow += q.pool_regrow_count()
    finally:
        q.close()
    # (EM resolutions: the oracle in the device's order-free fixed-point arithmetic - bit-identical by construction)
    want = ora.quant(cfg, s.tid_to_gid, b, off, n_threads=os.cpu_count() or 1, em_arith="reference" if os.environ.get("AFQ_EM_ORDER") == "canonical" else "fixed")
    assert_same_result(got, want, what=f"seed {seed} {res} usa={usa} {kw} sizes={sizes}")
    return sum(sizes), rehash


one.mono = one.regrow = 0


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad, reads, t0 = 0, 0, time.time()
    for seed in range(first, first + n):
        try:
            r, rh = one(seed)
            reads += r
            if rh:
                print(f"seed {seed}: {rh} range(s) re-keyed", flush=True)
        except Exception as e:   # noqa: BLE001
            bad += 1
            print(f"seed {seed} FAILED: {type(e).__name__}: {str(e)[:300]}", flush=True)
    print(f"extended fuzz: {n} workloads, {reads} reads, {bad} failures, {one.mono} cells through the one-workgroup kernel, {one.regrow} pool re-grows, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)
