"""Shared helpers for the parity tests."""
import importlib
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
pkg = importlib.import_module("alevin-fry_amd")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def rows_of(res):
    """QuantResult -> list of [(col, val), ...] per cell."""
    out = []
    for i in range(res.n_cells):
        g, v = res.row(i)
        out.append(list(zip(g.tolist(), v.tolist())))
    return out


def assert_same_result(a, b, exact=True, rtol=0.0, what=""):
    """Bit-exact (integer work) or rtol comparison of two QuantResults, cell by cell."""
    assert a.n_cells == b.n_cells, what
    assert np.array_equal(a.bc, b.bc), what + " barcodes differ"
    assert np.array_equal(a.nrec, b.nrec), what + " nrec differ"
    assert np.array_equal(a.flags, b.flags), what + " flags differ"
    assert np.array_equal(a.cell_ptr, b.cell_ptr), what + f" cell_ptr differ (nnz {a.cell_ptr[-1]} vs {b.cell_ptr[-1]})"
    assert np.array_equal(a.gene, b.gene), what + " columns differ"
    if exact:
        assert np.array_equal(a.val.view(np.uint32), b.val.view(np.uint32)), what + " values differ (bitwise)"
    else:
        np.testing.assert_allclose(a.val, b.val, rtol=rtol, atol=0, err_msg=what)


def cfg_for(s, resolution="cr-like", **kw):
    return pkg.WorkerConfig.for_resolution(resolution, usa_mode=s.usa, num_genes=s.num_genes, num_rows=s.num_rows, **kw)


EM_FLOOR = 0.01   # em.rs:568-572: abundances below it leave the row


def em_row_differences(row_a, row_b):
    """Two EM rows (columns ascending, values above the 0.01 output floor of em.rs:568-572) entry by entry:
    (entries, entries of both rows beyond 1e-4 relative, entries only one row holds, those of them whose surviving value is more
    than 1e-4 above the floor, largest relative difference of the common entries).  An entry only one row holds crossed the floor;
    when the survivor is within 1e-4 of 0.01 the two abundances are as close as north_star asks wherever the other one lies
    in [0.01 (1 - 1e-4), 0.01) - a crossing further up is a real difference."""
    (g0, v0), (g1, v1) = row_a, row_b
    cols = np.union1d(g0, g1)
    a = np.zeros(len(cols), np.float64)
    b = np.zeros(len(cols), np.float64)
    a[np.searchsorted(cols, g0)] = v0
    b[np.searchsorted(cols, g1)] = v1
    both = (a > 0) & (b > 0)
    rel = np.abs(a[both] - b[both]) / np.maximum(a[both], b[both])
    one = (a == 0) != (b == 0)
    off = one & (np.maximum(a, b) > EM_FLOOR * (1 + 1e-4))
    return len(cols), int((rel > 1e-4).sum()), int(one.sum()), int(off.sum()), float(rel.max()) if len(rel) else 0.0


def assert_em_within_the_reference_envelope(got, oracle_module, cfg, tid_to_gid, data, offs, rows_of_got=None, n_threads=None, what=""):
    """north_star's bar for the EM resolutions, at any size: the device rows against the oracle IN THE REFERENCE'S ARITHMETIC
    (f32 sums, canonical class order; oracle/afq_oracle.cpp em_update <- src/em.rs:189-248, 455-533).  The reference itself
    sums its classes in a HashMap's order (em.rs:464), so two of its own runs can leave an entry on either side of the 0.01
    output floor, or - when a cell's round count changes with it - apart by more than 1e-4.  That envelope is MEASURED here, on
    the same cells: the oracle under three shuffled class orders against its canonical order.  The device must stay inside it:
    no more entries beyond 1e-4 and no more floor crossings away from the floor than the WORST SINGLE one of the reference's own
    reorderings produces (the maximum over the three shuffles, not their sum; 0 and 0 on every sample seen so far); crossings AT the floor (survivor within 1e-4 of 0.01) are within the tolerance and
    are reported.  Returns the counts."""
    n_threads = n_threads or os.cpu_count() or 8
    want = oracle_module.quant(cfg, tid_to_gid, data, offs, n_threads=n_threads, em_arith="reference")
    n = want.n_cells
    rows = rows_of_got if rows_of_got is not None else [got.row(j) for j in range(n)]

    def tally(rows_x):
        t = np.zeros(5)
        for j in range(n):
            e, far, cross, off, mx = em_row_differences(rows_x[j], want.row(j))
            t[:4] += (e, far, cross, off)
            t[4] = max(t[4], mx)
        return t

    dev = tally(rows)
    env = np.zeros(5)   # the LARGEST count any single reordering produced (not their sum: the device is one run, not three)
    for seed in (11, 12, 13):
        perm = oracle_module.quant(cfg, tid_to_gid, data, offs, n_threads=n_threads, em_arith="reference", em_order_seed=seed)
        env = np.maximum(env, tally([perm.row(j) for j in range(n)]))
    out = {"entries": int(dev[0]), "device_beyond_1e-4_rel": int(dev[1]), "device_floor_crossings": int(dev[2]),
           "device_floor_crossings_off_the_floor": int(dev[3]), "device_max_rel_diff": dev[4],
           "beyond_1e-4_allowed_by_shuffle_envelope": int(env[1]), "floor_crossings_of_the_shuffle_envelope": int(env[2]),
           "floor_crossings_off_the_floor_allowed_by_shuffle_envelope": int(env[3]), "envelope_max_rel_diff": env[4]}
    assert out["device_beyond_1e-4_rel"] <= out["beyond_1e-4_allowed_by_shuffle_envelope"], f"{what} {out}"
    assert out["device_floor_crossings_off_the_floor"] <= out["floor_crossings_off_the_floor_allowed_by_shuffle_envelope"], f"{what} {out}"
    return out
