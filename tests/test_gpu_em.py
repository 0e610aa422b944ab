"""The device EM (csrc/afq_em2.hip: order-free fixed-point sums) against the oracle in BOTH arithmetics.

north_star: EM resolutions "within 1e-4 relative".  The device does not replay the reference's f32 additions (their order is
a HashMap walk, em.rs:464 - the reference itself does not repeat it run to run); it accumulates each round's shares as
integers.  What is required of it:
  * against the oracle's restatement of the reference's arithmetic (canonical class order): the same non-zero entries
    (the 0.01 output floor, em.rs:568-572, cuts the same ones) and every value within 1e-4 relative;
  * against the oracle's restatement of the fixed-point arithmetic (em_update_fixed): bit-identical;
  * AFQ_EM_ORDER=canonical: the sequential f32 kernels of rounds 1-3, bit-identical to the reference arithmetic."""
import numpy as np
import pytest

from util import ROOT, assert_same_result, cfg_for, pkg

pytestmark = pytest.mark.gpu
synth = pkg.synth

EM_RES = ["cr-like-em", "parsimony-em", "parsimony-gene-em"]


def _device(cfg, s, b, off):
    q = pkg.Quantifier(cfg, s.tid_to_gid)
    try:
        return q.quant_chunks(b, off)
    finally:
        q.close()


def _workload(usa, seed=5, sizes=None, **kw):
    sizes = sizes or [30000, 9000, 4000, 1500, 700, 260, 250, 120, 99, 40, 3]
    args = dict(num_genes=400, txp_per_gene=3, usa=usa, dup=0.5, zipf=0.6, cross=0.4, umi_err=0.02, max_extra_na=5)
    args.update(kw)
    s = synth.synth(seed, sizes, **args)
    return s, *s.encode()


def assert_within_tolerance(got, want, what=""):
    """Same rows (= the same entries survive the output floor), values within north_star's 1e-4."""
    assert np.array_equal(got.cell_ptr, want.cell_ptr), what + " rows differ in length"
    assert np.array_equal(got.gene, want.gene), what + " different entries are non-zero"
    np.testing.assert_allclose(got.val, want.val, rtol=1e-4, atol=0, err_msg=what)


@pytest.mark.parametrize("usa", [False, True])
@pytest.mark.parametrize("res", EM_RES)
def test_em_default_within_1e4_of_the_reference_arithmetic(oracle_module, res, usa):
    s, b, off = _workload(usa)
    cfg = cfg_for(s, res)
    got = _device(cfg, s, b, off)
    want = oracle_module.quant(cfg, s.tid_to_gid, b, off, em_arith="reference")
    assert_within_tolerance(got, want, f"{res} usa={usa}")
    assert np.array_equal(got.flags, want.flags) and np.array_equal(got.bc, want.bc)
    assert not np.array_equal(got.val.view(np.uint32), want.val.view(np.uint32))   # (it IS another arithmetic: some low bits differ)


@pytest.mark.parametrize("usa", [False, True])
@pytest.mark.parametrize("res", EM_RES)
def test_em_default_is_bit_identical_to_the_fixed_point_oracle(oracle_module, res, usa):
    s, b, off = _workload(usa, seed=6)
    cfg = cfg_for(s, res)
    assert_same_result(_device(cfg, s, b, off), oracle_module.quant(cfg, s.tid_to_gid, b, off, em_arith="fixed"), what=f"{res} usa={usa}")


@pytest.mark.parametrize("usa", [False, True])
@pytest.mark.parametrize("res", EM_RES)
def test_em_canonical_order_is_bit_identical_to_the_reference_arithmetic(oracle_module, monkeypatch, res, usa):
    monkeypatch.setenv("AFQ_EM_ORDER", "canonical")
    s, b, off = _workload(usa, seed=7)
    cfg = cfg_for(s, res)
    assert_same_result(_device(cfg, s, b, off), oracle_module.quant(cfg, s.tid_to_gid, b, off, em_arith="reference"), what=f"{res} usa={usa}")


@pytest.mark.parametrize("init_uniform", [False, True])
def test_em_init_uniform(oracle_module, init_uniform):
    s, b, off = _workload(True, seed=8)
    cfg = cfg_for(s, "cr-like-em", em_init_uniform=init_uniform)
    assert_same_result(_device(cfg, s, b, off), oracle_module.quant(cfg, s.tid_to_gid, b, off, em_arith="fixed"))


@pytest.mark.parametrize("tier", [1, 2, 3, 4, "4-wide-ids"])
@pytest.mark.parametrize("usa", [False, True])
def test_every_instance_of_the_rounds_kernel_agrees(oracle_module, monkeypatch, tier, usa):
    """The rounds kernel has five instances by cell size (all in LDS at 256 / 512 / 1024 threads, lists streamed, hot entries in
    LDS and the rest in global memory); AFQ_TEST_EM2_MIN_TIER sends every cell to the given one or a larger one: the sums are
    integers, the rows must not move by a bit.  The streamed instances read 16-bit state ids (label words, both sibling links
    in one word) unless a cell has more than 65 534 of them; AFQ_TEST_EM2_WIDE_IDS takes small cells down that 32-bit route."""
    if tier == "4-wide-ids":
        monkeypatch.setenv("AFQ_TEST_EM2_WIDE_IDS", "1")
        tier = 4
    monkeypatch.setenv("AFQ_TEST_EM2_MIN_TIER", str(tier))
    s, b, off = _workload(usa, seed=9)
    cfg = cfg_for(s, "parsimony-em")
    assert_same_result(_device(cfg, s, b, off), oracle_module.quant(cfg, s.tid_to_gid, b, off, em_arith="fixed"), what=f"tier {tier}")


def test_em_long_labels_and_many_classes(oracle_module):
    """Labels of up to a dozen genes (the class pass keeps four label words in registers and walks the rest), a cell big
    enough for the 1024-thread instance, duplicated classes (the device does not merge equal labels: count * q is linear)."""
    s, b, off = _workload(True, seed=10, sizes=[120000, 50000, 800], num_genes=3000, cross=0.7, max_extra_na=12, dup=0.3)
    cfg = cfg_for(s, "cr-like-em")
    got = _device(cfg, s, b, off)
    assert_same_result(got, oracle_module.quant(cfg, s.tid_to_gid, b, off, em_arith="fixed", n_threads=4))
    assert_within_tolerance(got, oracle_module.quant(cfg, s.tid_to_gid, b, off, em_arith="reference", n_threads=4))


@pytest.mark.parametrize("tier", [0, 3, 4])
def test_em_labels_of_more_than_63_genes(oracle_module, monkeypatch, tier):
    """The streamed instances find a class's words from the runs of equal-length classes; labels of 63 words and more share one
    run whose offsets come from the offset table instead.  Reads with up to 90 further alignments: labels of every length from
    2 to beyond 63 in one cell, through the all-in-LDS instance and both streamed ones."""
    monkeypatch.setenv("AFQ_TEST_EM2_MIN_TIER", str(tier))
    s, b, off = _workload(False, seed=13, sizes=[20000, 3000, 150], num_genes=2000, cross=0.8, max_extra_na=90, dup=0.3)
    cfg = cfg_for(s, "cr-like-em")
    want = oracle_module.quant(cfg, s.tid_to_gid, b, off, em_arith="fixed", n_threads=4)
    assert_same_result(_device(cfg, s, b, off), want, what=f"tier {tier}")


def test_em_scratch_short_of_the_plan_is_sized_on_the_host(oracle_module, monkeypatch):
    """The EM follows a range's kernels on the device, in scratch set aside from an upper bound ahead of time; with next to
    nothing set aside (AFQ_TEST_EM2_SCRATCH_FRAC) the device-side plan does not fit, the kernels return at once and the host sizes the
    EM itself: same rows, and the counter says so."""
    s, b, off = _workload(True, seed=12)
    cfg = cfg_for(s, "parsimony-em")
    want = oracle_module.quant(cfg, s.tid_to_gid, b, off, em_arith="fixed")
    q = pkg.Quantifier(cfg, s.tid_to_gid)
    try:
        assert_same_result(q.quant_chunks(b, off), want)
        assert q.em_resize_count() == 0
    finally:
        q.close()
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests'); import numpy as np\n"
            "from util import pkg, cfg_for\n"
            "s = pkg.synth.synth(12, [30000, 9000, 4000, 1500, 700, 260, 250, 120, 99, 40, 3], num_genes=400, txp_per_gene=3, usa=True, dup=0.5, zipf=0.6, cross=0.4, umi_err=0.02, max_extra_na=5)\n"
            "b, off = s.encode(); q = pkg.Quantifier(cfg_for(s, 'parsimony-em'), s.tid_to_gid); r = q.quant_chunks(b, off)\n"
            "print(q.em_resize_count(), r.val.view(np.uint32).sum(dtype=np.uint64), len(r.gene))\n") % (ROOT, ROOT)
    import os
    env = dict(os.environ, AFQ_TEST_EM2_SCRATCH_FRAC="0.00001")   # (read once per process: a process of its own)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-500:]
    n_resized, vsum, nnz = out.stdout.split()[-3:]
    assert int(n_resized) >= 1 and int(nnz) == len(want.gene) and int(vsum) == int(want.val.view(np.uint32).sum(dtype=np.uint64))
