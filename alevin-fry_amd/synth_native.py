"""ctypes binding of the multi-threaded C++ generator (csrc/afq_synth.cpp, include/afquant_synth.h).

Used by bench.py for the full-size PBMC-10k-like input.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

from .afquant import LIB_PATH


class SynthParams(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64),
        ("n_cells", C.c_uint32),
        ("min_reads", C.c_uint32),
        ("median_reads", C.c_double),
        ("sigma", C.c_double),
        ("num_genes", C.c_uint32),
        ("txp_per_gene", C.c_uint32),
        ("usa", C.c_uint32),
        ("umi_len", C.c_uint32),
        ("dup", C.c_double),
        ("p_na2", C.c_double),
        ("p_na3", C.c_double),
        ("cross", C.c_double),
        ("umi_err", C.c_double),
        ("zipf", C.c_double),
        ("p_unspliced", C.c_double),
        ("p_both", C.c_double),
        ("n_threads", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


@dataclass
class NativeRad:
    data: np.ndarray  # uint8, the collated chunks back to back
    chunk_off: np.ndarray  # uint64
    cell_nrec: np.ndarray  # uint32
    tid_to_gid: np.ndarray
    num_genes: int
    num_rows: int
    usa: bool
    n_reads: int


def generate(seed=2, n_cells=11000, median_reads=30000.0, sigma=0.6, num_genes=36601, txp_per_gene=5, usa=False,
             umi_len=12, dup=0.4, p_na2=0.2, p_na3=0.1, cross=0.5, umi_err=0.01, zipf=5.0, p_unspliced=0.35,
             p_both=0.08, min_reads=1, n_threads=0, pinned_out=None) -> NativeRad:
    lib = C.CDLL(LIB_PATH)
    P = C.POINTER
    lib.afq_synth_dims.argtypes = [P(SynthParams), P(C.c_uint32), P(C.c_uint32), P(C.c_uint32)]
    lib.afq_synth_dims.restype = None
    lib.afq_synth_t2g.argtypes = [P(SynthParams), P(C.c_uint32)]
    lib.afq_synth_t2g.restype = None
    lib.afq_synth_plan.argtypes = [P(SynthParams), P(C.c_uint32), P(C.c_uint64), P(C.c_uint64), P(C.c_uint64)]
    lib.afq_synth_plan.restype = C.c_int
    lib.afq_synth_fill.argtypes = [P(SynthParams), P(C.c_uint32), P(C.c_uint64), C.c_void_p, C.c_uint64]
    lib.afq_synth_fill.restype = C.c_int
    p = SynthParams(seed, n_cells, min_reads, median_reads, sigma, num_genes, txp_per_gene, int(usa), umi_len, dup,
                    p_na2, p_na3, cross, umi_err, zipf, p_unspliced, p_both, n_threads or (os.cpu_count() or 1), 0)
    rc, ng, nr = C.c_uint32(), C.c_uint32(), C.c_uint32()
    lib.afq_synth_dims(C.byref(p), C.byref(rc), C.byref(ng), C.byref(nr))
    t2g = np.zeros(rc.value, np.uint32)
    lib.afq_synth_t2g(C.byref(p), t2g.ctypes.data_as(P(C.c_uint32)))
    nrec = np.zeros(n_cells, np.uint32)
    off = np.zeros(n_cells, np.uint64)
    tb, tr = C.c_uint64(), C.c_uint64()
    r = lib.afq_synth_plan(C.byref(p), nrec.ctypes.data_as(P(C.c_uint32)), off.ctypes.data_as(P(C.c_uint64)),
                           C.byref(tb), C.byref(tr))
    if r != 0:
        raise RuntimeError(f"afq_synth_plan failed ({r})")
    data = np.empty(tb.value, np.uint8) if pinned_out is None else pinned_out(tb.value)
    r = lib.afq_synth_fill(C.byref(p), nrec.ctypes.data_as(P(C.c_uint32)), off.ctypes.data_as(P(C.c_uint64)),
                           data.ctypes.data_as(C.c_void_p), tb.value)
    if r != 0:
        raise RuntimeError(f"afq_synth_fill failed ({r})")
    return NativeRad(data, off, nrec, t2g, ng.value, nr.value, bool(usa), int(tr.value))
