"""CPU: the oracle's two EM arithmetics against each other (no GPU needed).

`reference` = f32 additions in canonical class order (the restatement of em.rs:189-248, 458-485 that the reference's own unit
vectors pin, tests/test_oracle_golden.py); `fixed` = the order-free fixed-point accumulation the device computes
(afq_oracle.cpp em_update_fixed; csrc/afq_em2.hip).  north_star allows 1e-4 relative on EM counts: `fixed` must stay inside
it and must cut the same entries at the 0.01 output floor."""
import numpy as np
import pytest

from util import cfg_for, pkg

synth = pkg.synth


@pytest.mark.parametrize("usa", [False, True])
@pytest.mark.parametrize("res", ["cr-like-em", "parsimony-em"])
def test_fixed_point_em_within_1e4_of_reference_arithmetic(oracle, res, usa):
    sizes = [20000, 6000, 2500, 900, 300, 150, 60]
    s = synth.synth(21, sizes, num_genes=300, txp_per_gene=3, usa=usa, dup=0.5, zipf=0.6, cross=0.4, umi_err=0.02, max_extra_na=5)
    b, off = s.encode()
    cfg = cfg_for(s, res)
    ref, it_ref = oracle.quant(cfg, s.tid_to_gid, b, off, em_arith="reference", want_iters=True)
    fix, it_fix = oracle.quant(cfg, s.tid_to_gid, b, off, em_arith="fixed", want_iters=True)
    assert np.array_equal(ref.cell_ptr, fix.cell_ptr) and np.array_equal(ref.gene, fix.gene)   # the same entries survive the floor
    np.testing.assert_allclose(fix.val, ref.val, rtol=1e-4, atol=0)
    assert np.array_equal(it_ref, it_fix)          # and the same number of rounds
    assert it_ref.max() > 2                        # (the EM did iterate)
    assert not np.array_equal(ref.val.view(np.uint32), fix.val.view(np.uint32))


def test_fixed_point_em_does_not_depend_on_class_order(oracle):
    """Integer sums: shuffling the order in which the classes are visited (what the reference's HashMap does) cannot move a bit -
    under the reference arithmetic it does."""
    s = synth.synth(22, [15000, 4000, 900], num_genes=300, txp_per_gene=3, usa=True, dup=0.5, zipf=0.6, cross=0.4, umi_err=0.02, max_extra_na=5)
    b, off = s.encode()
    cfg = cfg_for(s, "cr-like-em")
    base = oracle.quant(cfg, s.tid_to_gid, b, off, em_arith="fixed")
    moved = False
    for seed in (1, 2, 3):
        o = oracle.quant(cfg, s.tid_to_gid, b, off, em_arith="fixed", em_order_seed=seed)
        assert np.array_equal(o.gene, base.gene) and np.array_equal(o.val.view(np.uint32), base.val.view(np.uint32))
        r = oracle.quant(cfg, s.tid_to_gid, b, off, em_arith="reference", em_order_seed=seed)
        moved |= not np.array_equal(r.val.view(np.uint32), oracle.quant(cfg, s.tid_to_gid, b, off).val.view(np.uint32))
    assert moved
