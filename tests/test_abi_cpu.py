"""CPU tests of the boundary: the C-ABI library loads and exports every symbol of include/afquant.h."""
import ctypes
import os
import re

import pytest

from util import ROOT, pkg


def test_header_symbols_match_export_list():
    hdr = open(os.path.join(ROOT, "include", "afquant.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)  # prototypes only, not prose
    declared = set(re.findall(r"\b(afq_[a-z_]+)\s*\(", hdr))
    assert declared == set(pkg._abi.EXPORTS)


def test_library_exports_every_symbol():
    path = pkg.afquant.LIB_PATH
    if not os.path.exists(path):
        import __graft_entry__ as ge  # noqa

        ge.build()
    lib = ctypes.CDLL(path)
    for name in pkg._abi.EXPORTS:
        assert hasattr(lib, name), name
    lib.afq_abi_version.restype = ctypes.c_int
    assert lib.afq_abi_version() == pkg._abi.AFQ_ABI_VERSION


def test_config_struct_layout():
    c = pkg.WorkerConfig.for_resolution("parsimony-em", num_genes=10, num_rows=10).to_c()
    assert ctypes.sizeof(c) == 20 * 4   # ABI version 2: + num_bootstraps, summary_stat, boot_seed (u64, 8-byte aligned at offset 72)
    assert pkg._abi.AfqConfig.boot_seed.offset == 72 and pkg._abi.AfqConfig.dump_eq.offset == 56
    assert c.resolution == 3 and c.large_graph_thresh == 1000 and c.pug_exact_umi == 0 and c.small_thresh == 100
    c = pkg.WorkerConfig.for_resolution("cr-like", num_genes=10, num_rows=10).to_c()
    assert c.resolution == 1 and c.large_graph_thresh == 0 and c.pug_exact_umi == 1


def test_create_rejects_bad_arguments_without_a_gpu():
    """Argument validation happens before any device call, so it is testable on CPU."""
    lib = pkg.load_library()
    import numpy as np

    t2g = np.zeros(4, np.uint32)
    h = ctypes.c_void_p()
    cfg = pkg.WorkerConfig.for_resolution("cr-like", num_genes=1, num_rows=1, bc_bytes=3).to_c()
    rc = lib.afq_create(ctypes.byref(cfg), t2g.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), 4, 0, ctypes.byref(h))
    assert rc == pkg._abi.AFQ_ERR_INVALID_ARG and b"bc_bytes" in lib.afq_last_error(None)
    cfg = pkg.WorkerConfig.for_resolution("cr-like", usa_mode=True, num_genes=4, num_rows=5).to_c()
    rc = lib.afq_create(ctypes.byref(cfg), t2g.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), 4, 0, ctypes.byref(h))
    assert rc == pkg._abi.AFQ_ERR_INVALID_ARG
    t2g[2] = 9
    cfg = pkg.WorkerConfig.for_resolution("cr-like", num_genes=4, num_rows=4).to_c()
    rc = lib.afq_create(ctypes.byref(cfg), t2g.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), 4, 0, ctypes.byref(h))
    assert rc == pkg._abi.AFQ_ERR_INVALID_ARG and b"tid_to_gid" in lib.afq_last_error(None)


def test_every_declared_function_of_every_header_is_exported():
    """include/*.h (boundary, host front-end, synthetic generator): each declared function is in the library."""
    import glob

    lib = ctypes.CDLL(pkg.afquant.LIB_PATH)
    n = 0
    for path in sorted(glob.glob(os.path.join(ROOT, "include", "*.h"))):
        hdr = open(path).read()
        hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
        hdr = re.sub(r"//[^\n]*", "", hdr)
        for name in set(re.findall(r"\b(afq_[a-z0-9_]+)\s*\(", hdr)):
            assert hasattr(lib, name), f"{os.path.basename(path)}: {name} is declared but not exported"
            n += 1
    assert n >= 20


def test_umi_len_hint_is_validated():
    lib = pkg.load_library()
    import numpy as np

    t2g = np.zeros(4, np.uint32)
    h = ctypes.c_void_p()
    cfg = pkg.WorkerConfig.for_resolution("parsimony", num_genes=4, num_rows=4, umi_bytes=2, umi_len=9).to_c()
    assert cfg.umi_len == 9
    # a UMI of 9 bases does not fit a 2-byte field: refused before any device work
    rc = lib.afq_create(ctypes.byref(cfg), t2g.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), 4, 0, ctypes.byref(h))
    assert rc == pkg._abi.AFQ_ERR_INVALID_ARG and b"umi_len" in lib.afq_last_error(None) and not h.value
