"""ctypes binding of the CPU oracle (oracle/libafq_oracle.so).

TEST INFRASTRUCTURE: import only from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Never from the product package.
"""
from __future__ import annotations

import ctypes as C
import importlib
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_pkg = importlib.import_module("alevin-fry_amd")
_abi = _pkg._abi
_afq = importlib.import_module("alevin-fry_amd.afquant")

LIB_PATH = os.path.join(_HERE, "libafq_oracle.so")
_lib = None


def build():
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        p = C.POINTER
        L.ora_quant.argtypes = [p(_abi.AfqConfig), p(C.c_uint32), C.c_uint32, C.c_void_p, C.c_size_t, p(C.c_uint64),
                                C.c_uint32, C.c_uint64, C.c_int, p(_abi.AfqResult)]
        L.ora_quant.restype = C.c_int
        L.ora_quant_mt.argtypes = [p(_abi.AfqConfig), p(C.c_uint32), C.c_uint32, C.c_void_p, C.c_size_t, p(C.c_uint64),
                                   C.c_uint32, C.c_uint64, C.c_uint32, p(_abi.AfqResult)]
        L.ora_quant_mt.restype = C.c_int
        L.ora_result_release.argtypes = [p(_abi.AfqResult)]
        L.ora_result_em_iters.argtypes = [p(_abi.AfqResult)]
        L.ora_result_em_iters.restype = p(C.c_uint32)
        L.ora_result_pug_stats.argtypes = [p(_abi.AfqResult)]
        L.ora_result_pug_stats.restype = p(C.c_uint32)
        L.ora_set_tie_break.argtypes = [C.c_int]
        L.ora_set_tie_break.restype = None
        L.ora_set_em_order.argtypes = [C.c_uint64]
        L.ora_set_em_order.restype = None
        L.ora_set_em_arith.argtypes = [C.c_int]
        L.ora_set_em_arith.restype = None
        L.ora_result_eqclasses.argtypes = [p(_abi.AfqResult), p(_abi.AfqEqclasses)]
        L.ora_result_eqclasses.restype = C.c_int
        L.ora_result_bootstraps.argtypes = [p(_abi.AfqResult), p(_abi.AfqBootstraps)]
        L.ora_result_bootstraps.restype = C.c_int
        L.ora_philox4x32_10.argtypes = [p(C.c_uint32), p(C.c_uint32), p(C.c_uint32)]
        L.ora_philox4x32_10.restype = None
        L.ora_last_error.restype = C.c_char_p
        L.ora_em.argtypes = [p(C.c_uint32), p(C.c_uint32), p(C.c_uint32), C.c_uint32, C.c_uint32, C.c_int, C.c_int,
                             C.c_int, C.c_uint32, C.c_uint32, C.c_int, p(C.c_float), p(C.c_uint32)]
        L.ora_em.restype = C.c_int
        L.ora_has_edge.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, C.c_int]
        L.ora_has_edge.restype = C.c_int
        L.ora_atac_dedup.argtypes = [p(C.c_uint32), p(C.c_uint32), p(C.c_uint16), p(C.c_uint64), C.c_uint32,
                                     p(C.c_uint64), p(C.c_uint32), p(C.c_uint32), p(C.c_uint16), p(C.c_uint16)]
        L.ora_atac_dedup.restype = C.c_int
        L.ora_atac_dedup_rad.argtypes = [C.c_void_p, C.c_size_t, p(C.c_uint64), C.c_uint32, C.c_uint32, p(C.c_uint64), p(C.c_uint64),
                                         p(C.c_uint32), p(C.c_uint32), p(C.c_uint16), p(C.c_uint16), p(C.c_uint64)]
        L.ora_atac_dedup_rad.restype = C.c_int
        _lib = L
    return _lib


class OracleError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"oracle error {code}: {msg}")
        self.code = code


def quant(cfg, tid_to_gid, chunk_bytes, chunk_off, first_cell_index=0, force_route=0, want_iters=False, n_threads=1, want_pug_stats=False,
          tie_break_descending=False, check_tie_free=False, em_order_seed=0, em_arith="reference"):
    """cfg: WorkerConfig.  Returns QuantResult (same container as the product).
    n_threads > 1 spreads cells over worker threads (reference dispatch only).
    want_pug_stats: also return u32[n_cells, 5] = molecules, tie events, components with a tie, molecules of those components,
        tie-free components whose cover differs under the reversed scan (counted with check_tie_free only; must be 0).
    tie_break_descending: scan the parsimony cover's candidates in descending vertex id (measures what hangs on the tie-break).
    em_order_seed: non-zero = the EM sums its classes in a shuffled order (what the reference's HashMap does, em.rs:464).
    em_arith: "reference" = the reference's f32 additions (canonical class order); "fixed" = the order-free fixed-point
        accumulation the device EM computes by default (afq_oracle.cpp em_update_fixed) - the device must match it bit for bit,
        and it must stay within north_star's 1e-4 of "reference"."""
    L = lib()
    L.ora_set_tie_break(1 if tie_break_descending else (2 if check_tie_free else 0))
    L.ora_set_em_order(int(em_order_seed))
    assert em_arith in ("reference", "fixed")
    L.ora_set_em_arith(1 if em_arith == "fixed" else 0)
    ccfg = cfg.to_c()
    t2g = np.ascontiguousarray(tid_to_gid, dtype=np.uint32)
    b = np.ascontiguousarray(np.frombuffer(chunk_bytes, dtype=np.uint8) if not isinstance(chunk_bytes, np.ndarray) else chunk_bytes)
    off = np.ascontiguousarray(chunk_off, dtype=np.uint64)
    res = _abi.AfqResult()
    if n_threads > 1:
        assert force_route == 0
        rc = L.ora_quant_mt(C.byref(ccfg), t2g.ctypes.data_as(C.POINTER(C.c_uint32)), len(t2g), b.ctypes.data_as(C.c_void_p),
                            b.nbytes, off.ctypes.data_as(C.POINTER(C.c_uint64)), len(off), first_cell_index, n_threads,
                            C.byref(res))
    else:
        rc = L.ora_quant(C.byref(ccfg), t2g.ctypes.data_as(C.POINTER(C.c_uint32)), len(t2g), b.ctypes.data_as(C.c_void_p),
                         b.nbytes, off.ctypes.data_as(C.POINTER(C.c_uint64)), len(off), first_cell_index, force_route,
                         C.byref(res))
    if rc != 0:
        raise OracleError(rc, L.ora_last_error().decode())
    try:
        out = _afq.result_from_c(res)
        if cfg.dump_eq:
            ec = _abi.AfqEqclasses()
            L.ora_result_eqclasses(C.byref(res), C.byref(ec))
            out.eqclasses = _afq.eqclasses_from_c(ec)
        if cfg.num_bootstraps:
            bs = _abi.AfqBootstraps()
            L.ora_result_bootstraps(C.byref(res), C.byref(bs))
            out.bootstraps = _afq.bootstraps_from_c(bs)
        if want_pug_stats:
            n = out.n_cells
            ps = np.ctypeslib.as_array(L.ora_result_pug_stats(C.byref(res)), shape=(n, 5)).copy() if n else np.zeros((0, 5), np.uint32)
            return out, ps
        if want_iters:
            n = out.n_cells
            it = np.ctypeslib.as_array(L.ora_result_em_iters(C.byref(res)), shape=(n,)).copy() if n else np.zeros(0, np.uint32)
            return out, it
        return out
    finally:
        L.ora_set_tie_break(0)
        L.ora_set_em_order(0)
        L.ora_set_em_arith(0)
        L.ora_result_release(C.byref(res))


def philox4x32_10(ctr, key):
    """One Philox4x32-10 block of the oracle's restatement (for the published known-answer vectors)."""
    c = (C.c_uint32 * 4)(*ctr); k = (C.c_uint32 * 2)(*key); o = (C.c_uint32 * 4)()
    lib().ora_philox4x32_10(c, k, o)
    return [int(x) for x in o]


def em(labels, counts, num_alphas, only_unique=False, init_uniform=False, usa_offsets=None, dense=0):
    """labels: list of label lists; counts: per-class counts.  Returns (alphas f32[num_alphas], iters)."""
    L = lib()
    flat = np.asarray([x for l in labels for x in l], dtype=np.uint32)
    start = np.zeros(len(labels) + 1, dtype=np.uint32)
    start[1:] = np.cumsum([len(l) for l in labels])
    cnt = np.asarray(counts, dtype=np.uint32)
    out = np.zeros(num_alphas, dtype=np.float32)
    iters = C.c_uint32(0)
    uo, ao = usa_offsets if usa_offsets else (0, 0)
    p32 = C.POINTER(C.c_uint32)
    L.ora_em(flat.ctypes.data_as(p32), start.ctypes.data_as(p32), cnt.ctypes.data_as(p32), len(labels), num_alphas,
             int(only_unique), int(init_uniform), int(usa_offsets is not None), uo, ao, int(dense),
             out.ctypes.data_as(C.POINTER(C.c_float)), C.byref(iters))
    return out, iters.value


def has_edge(xu, xc, yu, yc, exact=False):
    return lib().ora_has_edge(xu, xc, yu, yc, int(exact))


def atac_dedup(ref, start, frag_len, cell_ptr):
    L = lib()
    ref = np.ascontiguousarray(ref, np.uint32); start = np.ascontiguousarray(start, np.uint32)
    frag_len = np.ascontiguousarray(frag_len, np.uint16); cell_ptr = np.ascontiguousarray(cell_ptr, np.uint64)
    n = len(ref); nc = len(cell_ptr) - 1
    o_ptr = np.zeros(nc + 1, np.uint64); o_ref = np.zeros(n, np.uint32); o_start = np.zeros(n, np.uint32)
    o_len = np.zeros(n, np.uint16); o_cnt = np.zeros(n, np.uint16)
    a = lambda x, t: x.ctypes.data_as(C.POINTER(t))
    L.ora_atac_dedup(a(ref, C.c_uint32), a(start, C.c_uint32), a(frag_len, C.c_uint16), a(cell_ptr, C.c_uint64), nc,
                     a(o_ptr, C.c_uint64), a(o_ref, C.c_uint32), a(o_start, C.c_uint32), a(o_len, C.c_uint16), a(o_cnt, C.c_uint16))
    m = int(o_ptr[-1])
    return o_ptr, o_ref[:m], o_start[:m], o_len[:m], o_cnt[:m]


def atac_dedup_rad(chunk_bytes, chunk_off, bc_bytes=4):
    """deduplicate.rs:199-237 from collated scATAC chunks.  Returns (cell_ptr, bc, ref, start, frag_len, count, stats dict)."""
    L = lib()
    b = np.ascontiguousarray(np.frombuffer(chunk_bytes, dtype=np.uint8) if not isinstance(chunk_bytes, np.ndarray) else chunk_bytes)
    off = np.ascontiguousarray(chunk_off, dtype=np.uint64)
    nc = len(off)
    cap = max(1, b.nbytes // (4 + bc_bytes))
    o_ptr = np.zeros(nc + 1, np.uint64); o_bc = np.zeros(max(nc, 1), np.uint64)
    o_ref = np.zeros(cap, np.uint32); o_start = np.zeros(cap, np.uint32); o_len = np.zeros(cap, np.uint16); o_cnt = np.zeros(cap, np.uint16)
    st = np.zeros(5, np.uint64)
    a = lambda x, t: x.ctypes.data_as(C.POINTER(t))
    rc = L.ora_atac_dedup_rad(b.ctypes.data_as(C.c_void_p), b.nbytes, a(off, C.c_uint64), nc, bc_bytes, a(o_ptr, C.c_uint64), a(o_bc, C.c_uint64),
                              a(o_ref, C.c_uint32), a(o_start, C.c_uint32), a(o_len, C.c_uint16), a(o_cnt, C.c_uint16), a(st, C.c_uint64))
    if rc != 0:
        raise OracleError(rc, L.ora_last_error().decode())
    m = int(o_ptr[-1])
    stats = dict(n_records=int(st[0]), n_multimapped=int(st[1]), n_not_mapped_pair=int(st[2]), n_deduplicated=int(st[3]), n_long_fragments=int(st[4]))
    return o_ptr, o_bc[:nc], o_ref[:m], o_start[:m], o_len[:m], o_cnt[:m], stats
