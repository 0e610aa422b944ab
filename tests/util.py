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
