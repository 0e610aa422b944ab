"""ctypes binding of the synthetic collated-RAD generator (csrc/afq_synth.hip, include/afquant_synth.h).

`generate` runs the multi-threaded host implementation, `generate_device` the gfx950 kernels over the same
integer record model (same bytes).  Both can produce any cell range of a data set, which is how a rank makes
its shard of configs[3] without a file.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

from .afquant import LIB_PATH, load_library


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
        ("pow_skew", C.c_double),
        ("p_unspliced", C.c_double),
        ("p_both", C.c_double),
        ("n_threads", C.c_uint32),
        ("ref_count", C.c_uint32),
        ("tail", C.c_double),
        ("tail_max", C.c_uint32),
        ("family", C.c_uint32),
    ]


@dataclass
class NativeRad:
    data: np.ndarray | None  # uint8, the collated chunks back to back (None: the bytes live on the device only)
    chunk_off: np.ndarray  # uint64
    cell_nrec: np.ndarray  # uint32
    tid_to_gid: np.ndarray
    num_genes: int
    num_rows: int
    usa: bool
    n_reads: int
    n_bytes: int = 0
    first_cell: int = 0
    d_ptr: int = 0  # device address of the bytes (generate_device)
    device: int = 0

    def chunk_nbytes(self):
        return np.diff(np.concatenate((self.chunk_off, np.array([self.n_bytes], np.uint64))).astype(np.int64))

    def read_cells(self, idx):
        """Host copy of the chunks of cells `idx` (device-resident data): bytes back to back + their offsets."""
        idx = np.asarray(idx, dtype=np.int64)
        nb = self.chunk_nbytes()[idx]
        off = np.concatenate(([0], np.cumsum(nb)[:-1])).astype(np.uint64) if len(idx) else np.zeros(0, np.uint64)
        if self.data is not None:
            out = np.concatenate([self.data[int(self.chunk_off[i]):int(self.chunk_off[i]) + int(n)] for i, n in zip(idx, nb)]) \
                if len(idx) else np.zeros(0, np.uint8)
            return out, off
        lib = _lib()
        out = np.empty(int(nb.sum()), np.uint8)
        for i, n, o in zip(idx, nb, off):
            rc = lib.afq_synth_device_read(self.device, C.c_void_p(self.d_ptr + int(self.chunk_off[i])), int(n),
                                           C.c_void_p(out.ctypes.data + int(o)))
            if rc:
                raise RuntimeError(f"afq_synth_device_read failed ({rc})")
        return out, off

    def to_host(self):
        if self.data is None:
            out = np.empty(self.n_bytes, np.uint8)
            rc = _lib().afq_synth_device_read(self.device, C.c_void_p(self.d_ptr), self.n_bytes, C.c_void_p(out.ctypes.data))
            if rc:
                raise RuntimeError(f"afq_synth_device_read failed ({rc})")
            self.data = out
        return self.data

    def free(self):
        if self.d_ptr:
            _lib().afq_synth_device_free(self.device, C.c_void_p(self.d_ptr))
            self.d_ptr = 0


_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        load_library()   # (one HIP runtime per process: torch's comes up first, see afquant.load_library)
        lib = C.CDLL(LIB_PATH)
        P = C.POINTER
        lib.afq_synth_dims.argtypes = [P(SynthParams), P(C.c_uint32), P(C.c_uint32), P(C.c_uint32)]
        lib.afq_synth_dims.restype = None
        lib.afq_synth_t2g.argtypes = [P(SynthParams), P(C.c_uint32)]
        lib.afq_synth_t2g.restype = None
        lib.afq_synth_cell_sizes.argtypes = [P(SynthParams), P(C.c_uint32)]
        lib.afq_synth_cell_sizes.restype = C.c_int
        lib.afq_synth_host_plan.argtypes = [P(SynthParams), C.c_uint64, C.c_uint32, P(C.c_uint32), P(C.c_uint64), P(C.c_uint64)]
        lib.afq_synth_host_plan.restype = C.c_int
        lib.afq_synth_host_fill.argtypes = [P(SynthParams), C.c_uint64, C.c_uint32, P(C.c_uint32), P(C.c_uint64), C.c_void_p, C.c_uint64]
        lib.afq_synth_host_fill.restype = C.c_int
        lib.afq_synth_device_generate.argtypes = [P(SynthParams), C.c_int, C.c_uint64, C.c_uint32, P(C.c_uint32), P(C.c_uint64),
                                                  P(C.c_uint64), P(C.c_void_p)]
        lib.afq_synth_device_generate.restype = C.c_int
        lib.afq_synth_device_free.argtypes = [C.c_int, C.c_void_p]
        lib.afq_synth_device_free.restype = None
        lib.afq_synth_device_read.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_void_p]
        lib.afq_synth_device_read.restype = C.c_int
        _LIB = lib
    return _LIB


def params(seed=2, n_cells=11000, median_reads=30000.0, sigma=0.6, num_genes=36601, txp_per_gene=5, usa=False,
           umi_len=12, dup=0.4, p_na2=0.2, p_na3=0.1, cross=0.5, umi_err=0.01, zipf=1.1, pow_skew=0.0, p_unspliced=0.35,
           p_both=0.08, min_reads=1, n_threads=0, ref_count=0, tail=0.0, tail_max=64, family=8) -> SynthParams:
    """Defaults = SURVEY §8(d) config 2 (PBMC-10k-like); pass ref_count=199138 for its transcriptome size.
    pow_skew=16 is the round-1 popularity (half of all molecules on gene 0), kept as a stress variant.
    tail > 0 is the label-length tail: a geometric run of further refs (P(one more) = tail, at most tail_max per record) on
    the genes of the read's gene family (blocks of `family` gene ids); 0.6 gives E[na] ~ 3 with labels of 5..30 refs."""
    return SynthParams(seed, n_cells, min_reads, median_reads, sigma, num_genes, txp_per_gene, int(usa), umi_len, dup,
                       p_na2, p_na3, cross, umi_err, zipf, pow_skew, p_unspliced, p_both, n_threads or (os.cpu_count() or 1), ref_count,
                       tail, tail_max, family)


def _dims(lib, p):
    rc, ng, nr = C.c_uint32(), C.c_uint32(), C.c_uint32()
    lib.afq_synth_dims(C.byref(p), C.byref(rc), C.byref(ng), C.byref(nr))
    t2g = np.zeros(rc.value, np.uint32)
    lib.afq_synth_t2g(C.byref(p), t2g.ctypes.data_as(C.POINTER(C.c_uint32)))
    return t2g, ng.value, nr.value


def cell_sizes(p: SynthParams) -> np.ndarray:
    """Reads per cell of the whole data set, descending."""
    nrec = np.zeros(p.n_cells, np.uint32)
    r = _lib().afq_synth_cell_sizes(C.byref(p), nrec.ctypes.data_as(C.POINTER(C.c_uint32)))
    if r:
        raise RuntimeError(f"afq_synth_cell_sizes failed ({r})")
    return nrec


def _range(p, cell_range, sizes):
    nrec_all = cell_sizes(p) if sizes is None else sizes
    c0, c1 = (0, p.n_cells) if cell_range is None else cell_range
    return int(c0), np.ascontiguousarray(nrec_all[c0:c1])


def generate(cell_range=None, sizes=None, **kw) -> NativeRad:
    """Host generator.  cell_range = (c0, c1) of the data set's cells (default: all)."""
    lib = _lib()
    P = C.POINTER
    p = kw.pop("p", None) or params(**kw)
    t2g, ng, nr = _dims(lib, p)
    c0, nrec = _range(p, cell_range, sizes)
    n = len(nrec)
    off = np.zeros(n, np.uint64)
    tb = C.c_uint64()
    r = lib.afq_synth_host_plan(C.byref(p), c0, n, nrec.ctypes.data_as(P(C.c_uint32)), off.ctypes.data_as(P(C.c_uint64)), C.byref(tb))
    if r != 0:
        raise RuntimeError(f"afq_synth_host_plan failed ({r})")
    data = np.empty(tb.value, np.uint8)
    r = lib.afq_synth_host_fill(C.byref(p), c0, n, nrec.ctypes.data_as(P(C.c_uint32)), off.ctypes.data_as(P(C.c_uint64)),
                                data.ctypes.data_as(C.c_void_p), tb.value)
    if r != 0:
        raise RuntimeError(f"afq_synth_host_fill failed ({r})")
    return NativeRad(data, off, nrec, t2g, ng, nr, bool(p.usa), int(nrec.astype(np.int64).sum()), int(tb.value), c0)


def generate_device(device=0, cell_range=None, sizes=None, **kw) -> NativeRad:
    """Device generator: the same bytes as `generate`, written straight into HBM of `device` (NativeRad.d_ptr)."""
    lib = _lib()
    P = C.POINTER
    p = kw.pop("p", None) or params(**kw)
    t2g, ng, nr = _dims(lib, p)
    c0, nrec = _range(p, cell_range, sizes)
    n = len(nrec)
    off = np.zeros(n, np.uint64)
    tb = C.c_uint64()
    dp = C.c_void_p()
    r = lib.afq_synth_device_generate(C.byref(p), device, c0, n, nrec.ctypes.data_as(P(C.c_uint32)), off.ctypes.data_as(P(C.c_uint64)),
                                      C.byref(tb), C.byref(dp))
    if r != 0:
        raise RuntimeError(f"afq_synth_device_generate failed ({r})")
    return NativeRad(None, off, nrec, t2g, ng, nr, bool(p.usa), int(nrec.astype(np.int64).sum()), int(tb.value), c0, int(dp.value or 0), device)


class SynthAtacParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_cells", C.c_uint32), ("frags_per_cell", C.c_uint32), ("n_refs", C.c_uint32), ("ref_len", C.c_uint32),
                ("p_dup", C.c_double), ("p_multi", C.c_double), ("p_unmapped", C.c_double), ("flen_mu", C.c_double), ("flen_sigma", C.c_double),
                ("n_threads", C.c_uint32), ("reserved", C.c_uint32)]


def generate_atac(seed=5, n_cells=10000, frags_per_cell=20000, n_refs=25, ref_len=150_000_000, p_dup=0.2, p_multi=0.05, p_unmapped=0.05,
                  flen_mu=5.2, flen_sigma=0.6, n_threads=0):
    """the chunks of a collated scATAC RAD (u32 barcodes).  Returns (bytes uint8, chunk_off uint64)."""
    lib = _lib()
    lib.afq_synth_atac.argtypes = [C.POINTER(SynthAtacParams), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p]
    lib.afq_synth_atac.restype = C.c_int
    p = SynthAtacParams(seed, n_cells, frags_per_cell, n_refs, ref_len, p_dup, p_multi, p_unmapped, flen_mu, flen_sigma, n_threads or (os.cpu_count() or 1), 0)
    off = np.zeros(n_cells, np.uint64)
    tb = C.c_uint64()
    rc = lib.afq_synth_atac(C.byref(p), off.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(tb), None)
    if rc:
        raise RuntimeError(f"afq_synth_atac failed ({rc})")
    data = np.empty(tb.value, np.uint8)
    rc = lib.afq_synth_atac(C.byref(p), off.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(tb), data.ctypes.data_as(C.c_void_p))
    if rc:
        raise RuntimeError(f"afq_synth_atac failed ({rc})")
    return data, off
