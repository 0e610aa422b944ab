"""Host-side mirror of the reference's per-cell quant interface, over the C ABI.

The reference is compiled Rust whose hot path has no plugin/FFI surface: it is
called from `run_worker_thread` (src/quant.rs:659-1325) and configured by
`WorkerConfig` (src/quant.rs:398-416).  `WorkerConfig` below carries the same
fields with the same meaning; `Quantifier.quant_chunks` takes collated chunk
bytes (one chunk = one cell) and returns the rows the worker would have emitted.

There is no CPU fallback: if csrc/libafquant.so (hand-written HIP for gfx950) is
missing or no device is usable, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

from . import _abi
from ._abi import AfqBatchStats, AfqBootstraps, AfqConfig, AfqEqclasses, AfqKernelTime, AfqResult, RESOLUTIONS

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AFQ_LIB_PATH") or os.path.join(_HERE, "csrc", "libafquant.so")   # (AFQ_LIB_PATH: an instrumented build, profiles/ only)


class AfqError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"afquant error {code}: {msg}")
        self.code = code


@dataclass
class WorkerConfig:
    """Field-for-field WorkerConfig (src/quant.rs:398-416) + record field widths."""

    resolution: str = "cr-like"
    usa_mode: bool = False
    num_genes: int = 0  # gene-id space of tid_to_gid (USA: 2*G)
    num_rows: int = 0  # output columns (USA: 3*G)
    small_thresh: int = 100  # tiny_cell_thresh, DEFAULT_SMALL_CELL_FAST_THRESHOLD quant.rs:449
    large_graph_thresh: int = 0  # main.rs:332-341: 1000 for parsimony*, else 0
    pug_exact_umi: bool = True  # main.rs:652-703: umi-edit-dist 0 unless parsimony*
    sa_model: str = "winner-take-all"
    em_init_uniform: bool = False
    bc_bytes: int = 4
    umi_bytes: int = 4
    umi_len: int = 0  # UMI length in bases if known (RAD file tag ulen); 0 = unknown
    dump_eq: bool = False  # -d: keep each cell's gene-level equivalence classes (QuantResult.eqclasses; -em resolutions)
    num_bootstraps: int = 0  # -b: bootstrap replicates per non-tiny cell (QuantResult.bootstraps; -em resolutions)
    summary_stat: bool = False  # --summary-stat: population variance of the replicates instead of the n-1 sample variance
    boot_seed: int = 0  # key of the counter-based generator behind the bootstrap draws
    profile: bool = False

    @staticmethod
    def for_resolution(resolution: str, **kw) -> "WorkerConfig":
        """Apply the CLI's conditional defaults (src/main.rs:320-341, 652-703)."""
        pars = resolution.startswith("parsimony")
        d = dict(resolution=resolution, large_graph_thresh=1000 if pars else 0, pug_exact_umi=not pars)
        d.update(kw)
        return WorkerConfig(**d)

    def to_c(self) -> AfqConfig:
        if self.resolution not in RESOLUTIONS:
            raise ValueError(f"unknown resolution {self.resolution!r}")
        c = AfqConfig()
        c.abi_version = _abi.AFQ_ABI_VERSION
        c.resolution = RESOLUTIONS[self.resolution]
        c.sa_model = {"winner-take-all": 0, "prefer-ambig": 1}[self.sa_model]
        c.usa_mode = int(self.usa_mode)
        c.num_genes = self.num_genes
        c.num_rows = self.num_rows
        c.small_thresh = self.small_thresh
        c.large_graph_thresh = self.large_graph_thresh
        c.pug_exact_umi = int(self.pug_exact_umi)
        c.em_init_uniform = int(self.em_init_uniform)
        c.bc_bytes = self.bc_bytes
        c.umi_bytes = self.umi_bytes
        c.profile = int(self.profile)
        c.umi_len = int(self.umi_len)
        c.dump_eq = 1 if self.dump_eq else 0
        c.num_bootstraps = int(self.num_bootstraps)
        c.summary_stat = 1 if self.summary_stat else 0
        c.boot_seed = int(self.boot_seed)
        return c


@dataclass
class QuantResult:
    """Rows for a run of cells: CSR of (output column, f32 count) + per-cell scalars."""

    first_cell_index: int
    cell_ptr: np.ndarray
    gene: np.ndarray
    val: np.ndarray
    bc: np.ndarray
    nrec: np.ndarray
    flags: np.ndarray

    @property
    def n_cells(self) -> int:
        return len(self.bc)

    def row(self, i: int):
        a, b = int(self.cell_ptr[i]), int(self.cell_ptr[i + 1])
        return self.gene[a:b], self.val[a:b]

    def cell_stats(self, i: int):
        """sum/max/num_expr/mean_by_max/num_genes_over_mean as src/quant.rs:1150-1196."""
        _, v = self.row(i)
        s = np.float32(0.0)
        for x in v:  # f32 sum in gene-index order
            s = np.float32(s + x)
        mx = np.float32(v.max()) if len(v) else np.float32(0)
        n = len(v)
        with np.errstate(divide="ignore", invalid="ignore"):
            mean = np.float32(s) / np.float32(n)
            over = int((v > mean).sum())
            return dict(sum_umi=float(s), max_umi=float(mx), num_expr=n, mean_by_max=float(mean / mx),
                        num_genes_over_mean=over, dedup_rate=float(s / np.float32(self.nrec[i])))


@dataclass
class EqClasses:
    """-d: per-cell gene-level equivalence classes (label = ascending gene ids, count = molecules), CSR over CSR."""

    cell_ptr: np.ndarray
    label_ptr: np.ndarray
    labels: np.ndarray
    count: np.ndarray

    def cell(self, i: int):
        """Classes of cell i as a sorted list of (label tuple, count) - the reference's order is a hash map's."""
        a, b = int(self.cell_ptr[i]), int(self.cell_ptr[i + 1])
        lp = self.label_ptr
        return sorted((tuple(int(x) for x in self.labels[int(lp[k]):int(lp[k + 1])]), int(self.count[k])) for k in range(a, b))


@dataclass
class Bootstraps:
    """-b: per-cell non-zero bootstrap means and variances (two CSR matrices; columns = gene ids of gene_eqc's labels)."""

    mean_ptr: np.ndarray
    mean_col: np.ndarray
    mean_val: np.ndarray
    var_ptr: np.ndarray
    var_col: np.ndarray
    var_val: np.ndarray

    def mean(self, i: int):
        a, b = int(self.mean_ptr[i]), int(self.mean_ptr[i + 1])
        return self.mean_col[a:b], self.mean_val[a:b]

    def var(self, i: int):
        a, b = int(self.var_ptr[i]), int(self.var_ptr[i + 1])
        return self.var_col[a:b], self.var_val[a:b]


def bootstraps_from_c(bs) -> Bootstraps:
    def arr(ptr, count, dt):
        return np.ctypeslib.as_array(ptr, shape=(count,)).astype(dt, copy=True) if count else np.zeros(0, dtype=dt)

    n = int(bs.n_cells)
    mp, vp = arr(bs.mean_ptr, n + 1, np.uint64), arr(bs.var_ptr, n + 1, np.uint64)
    nm, nv = int(mp[-1]), int(vp[-1])
    return Bootstraps(mp, arr(bs.mean_col, nm, np.uint32), arr(bs.mean_val, nm, np.float32), vp, arr(bs.var_col, nv, np.uint32),
                      arr(bs.var_val, nv, np.float32))


def eqclasses_from_c(ec) -> EqClasses:
    def arr(ptr, count, dt):
        return np.ctypeslib.as_array(ptr, shape=(count,)).astype(dt, copy=True) if count else np.zeros(0, dtype=dt)

    n, k, w = int(ec.n_cells), int(ec.n_classes), int(ec.n_words)
    return EqClasses(arr(ec.cell_ptr, n + 1, np.uint64), arr(ec.label_ptr, k + 1, np.uint64), arr(ec.labels, w, np.uint32),
                     arr(ec.count, k, np.uint32))


class _ResultOwner:
    """Keeps a library-owned afq_result alive while numpy views point into it."""

    def __init__(self, release, res: AfqResult):
        self._release, self._res = release, res

    def __del__(self):
        try:
            self._release(C.byref(self._res))
        except Exception:
            pass


def result_from_c(res: AfqResult, owner=None) -> QuantResult:
    """With `owner` the big CSR arrays are zero-copy views of the library's pinned buffers
    (the owner releases them when the QuantResult is garbage collected); without, everything is copied."""
    n, nnz = int(res.n_cells), int(res.nnz)

    def arr(ptr, count, dt, view=False):
        if count == 0:
            return np.zeros(0, dtype=dt)
        a = np.ctypeslib.as_array(ptr, shape=(count,))
        return a if view else a.astype(dt, copy=True)

    z = owner is not None
    out = QuantResult(int(res.first_cell_index), arr(res.cell_ptr, n + 1, np.uint64), arr(res.gene, nnz, np.uint32, z),
                      arr(res.val, nnz, np.float32, z), arr(res.bc, n, np.uint64), arr(res.nrec, n, np.uint32),
                      arr(res.flags, n, np.uint8))
    out._owner = owner
    return out


_lib = None


def configure_runtime():
    """Runtime settings the path wants, set while they can still take effect: GPU_FORCE_BLIT_COPY_SIZE=0 (device<->host copies on
    the DMA engines, not as blit kernels on the compute queue next to the path's own kernels) is read by the HIP runtime when it
    comes up, so it is set here only when NO HIP runtime is mapped into the process yet (torch not imported, libamdhip64 not
    loaded) and the host has not said otherwise.  A host whose runtime is already up keeps its own setting; afq_create notes a
    missing one on stderr.  Returns True when the setting is in place."""
    import sys

    if "GPU_FORCE_BLIT_COPY_SIZE" in os.environ:
        return os.environ["GPU_FORCE_BLIT_COPY_SIZE"] == "0"
    mapped = "torch" in sys.modules
    if not mapped:
        try:
            with open("/proc/self/maps") as f:
                mapped = "libamdhip64" in f.read()
        except OSError:
            mapped = False
    if mapped:
        return False
    os.environ["GPU_FORCE_BLIT_COPY_SIZE"] = "0"
    return True


def load_library(path: str = LIB_PATH):
    """dlopen the HIP library and declare prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    # libafquant.so needs libamdhip64.so.7.  PyTorch-ROCm bundles its own copy under the same
    # soname; a process must hold exactly one HIP/HSA runtime, so when torch is installed it is
    # imported first and the library binds to the runtime torch already mapped (a torch-free
    # host gets /opt/rocm's).  See DESIGN.md "one HIP runtime per process".
    # (GPU_FORCE_BLIT_COPY_SIZE=0 - copies on the DMA engines, not as blit kernels next to the path's own - must be in the environment
    #  before the HIP runtime comes up: bench.py, the afquant CLI and tests/conftest.py set it; configure_runtime() sets it for a Python
    #  host that imports this binding before any HIP runtime is mapped and has not chosen a value itself.  afq_create says so once on
    #  stderr when it is missing.  INTEGRATION.md "runtime settings")
    configure_runtime()
    try:
        import torch  # noqa: F401

        if torch.cuda.is_available():
            torch.cuda.init()   # torch brings its HIP runtime up first; initialising it AFTER this library has used HIP finds no GPU
    except ImportError:
        pass
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} not found: the HIP extension is not built (run __graft_entry__.build()). "
            "There is no CPU fallback for the quant hot path."
        )
    lib = C.CDLL(path)
    p = C.POINTER
    lib.afq_create.argtypes = [p(AfqConfig), p(C.c_uint32), C.c_uint32, C.c_int, p(C.c_void_p)]
    lib.afq_create.restype = C.c_int
    lib.afq_destroy.argtypes = [C.c_void_p]
    lib.afq_destroy.restype = None
    lib.afq_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, p(C.c_uint64), C.c_uint32, C.c_uint64]
    lib.afq_submit.restype = C.c_int
    lib.afq_submit_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, p(C.c_uint64), C.c_uint32, C.c_uint64]
    lib.afq_submit_device.restype = C.c_int
    lib.afq_collect.argtypes = [C.c_void_p, p(AfqResult)]
    lib.afq_collect.restype = C.c_int
    lib.afq_result_release.argtypes = [p(AfqResult)]
    lib.afq_result_release.restype = None
    lib.afq_result_eqclasses.argtypes = [p(AfqResult), p(AfqEqclasses)]
    lib.afq_result_eqclasses.restype = C.c_int
    lib.afq_result_bootstraps.argtypes = [p(AfqResult), p(AfqBootstraps)]
    lib.afq_result_bootstraps.restype = C.c_int
    lib.afq_infer.argtypes = [C.c_void_p, p(C.c_uint32), p(C.c_uint64), C.c_uint32, p(C.c_uint64), p(C.c_uint32), p(C.c_uint32), C.c_uint32,
                              C.c_uint32, C.c_uint32, p(AfqResult)]
    lib.afq_infer.restype = C.c_int
    lib.afq_atac_dedup.argtypes = [C.c_void_p, p(C.c_uint32), p(C.c_uint32), p(C.c_uint16), p(C.c_uint64), C.c_uint32,
                                   p(p(C.c_uint64)), p(p(C.c_uint32)), p(p(C.c_uint32)), p(p(C.c_uint16)), p(p(C.c_uint16))]
    lib.afq_atac_dedup.restype = C.c_int
    lib.afq_atac_dedup_rad.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, p(C.c_uint64), C.c_uint32, C.c_uint32, C.c_int, p(p(C.c_uint64)),
                                       p(p(C.c_uint64)), p(p(C.c_uint32)), p(p(C.c_uint32)), p(p(C.c_uint16)), p(p(C.c_uint16)), p(_abi.AfqAtacStats)]
    lib.afq_atac_dedup_rad.restype = C.c_int
    lib.afq_free.argtypes = [C.c_void_p]
    lib.afq_free.restype = None
    lib.afq_get_kernel_times.argtypes = [C.c_void_p, p(AfqKernelTime), C.c_uint32]
    lib.afq_get_kernel_times.restype = C.c_int
    lib.afq_get_batch_stats.argtypes = [C.c_void_p, p(AfqBatchStats)]
    lib.afq_get_batch_stats.restype = C.c_int
    lib.afq_last_error.argtypes = [C.c_void_p]
    lib.afq_last_error.restype = C.c_char_p
    lib.afq_abi_version.argtypes = []
    lib.afq_abi_version.restype = C.c_int
    if lib.afq_abi_version() != _abi.AFQ_ABI_VERSION:
        raise RuntimeError("libafquant.so ABI version mismatch")
    _lib = lib
    return lib


class _Held(np.ndarray):
    """ndarray view that keeps the owner of its memory alive."""


class Quantifier:
    """One per device: config + tid_to_gid resident on the GPU (afq_ctx)."""

    def __init__(self, cfg: WorkerConfig, tid_to_gid: np.ndarray, device: int = 0):
        self.lib = load_library()
        self.cfg = cfg
        self._t2g = np.ascontiguousarray(tid_to_gid, dtype=np.uint32)
        ccfg = cfg.to_c()
        h = C.c_void_p()
        rc = self.lib.afq_create(C.byref(ccfg), self._t2g.ctypes.data_as(C.POINTER(C.c_uint32)), len(self._t2g),
                                 device, C.byref(h))
        if rc != 0:
            raise AfqError(rc, (self.lib.afq_last_error(None) or b"").decode())
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self.lib.afq_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != 0:
            raise AfqError(rc, (self.lib.afq_last_error(self._h) or b"").decode())

    def submit(self, chunk_bytes, chunk_off, first_cell_index: int = 0):
        b = np.ascontiguousarray(np.frombuffer(chunk_bytes, dtype=np.uint8) if not isinstance(chunk_bytes, np.ndarray) else chunk_bytes)
        off = np.ascontiguousarray(chunk_off, dtype=np.uint64)
        self._keep = (b, off)
        self._check(self.lib.afq_submit(self._h, b.ctypes.data_as(C.c_void_p), b.nbytes,
                                        off.ctypes.data_as(C.POINTER(C.c_uint64)), len(off), first_cell_index))

    def submit_ptr(self, h_ptr: int, n_bytes: int, chunk_off, first_cell_index: int = 0):
        """afq_submit from a raw host address (e.g. pinned memory the caller owns)."""
        off = np.ascontiguousarray(chunk_off, dtype=np.uint64)
        self._keep = (off,)
        self._check(self.lib.afq_submit(self._h, C.c_void_p(h_ptr), n_bytes,
                                        off.ctypes.data_as(C.POINTER(C.c_uint64)), len(off), first_cell_index))

    def submit_device(self, d_ptr: int, n_bytes: int, chunk_off, first_cell_index: int = 0):
        off = np.ascontiguousarray(chunk_off, dtype=np.uint64)
        self._keep = (off,)
        self._check(self.lib.afq_submit_device(self._h, C.c_void_p(d_ptr), n_bytes,
                                               off.ctypes.data_as(C.POINTER(C.c_uint64)), len(off), first_cell_index))

    def collect(self) -> QuantResult:
        res = AfqResult()
        self._check(self.lib.afq_collect(self._h, C.byref(res)))
        out = result_from_c(res, owner=_ResultOwner(self.lib.afq_result_release, res))
        if self.cfg.dump_eq:
            ec = AfqEqclasses()
            self._check(self.lib.afq_result_eqclasses(C.byref(res), C.byref(ec)))
            out.eqclasses = eqclasses_from_c(ec)
        if self.cfg.num_bootstraps:
            bs = AfqBootstraps()
            self._check(self.lib.afq_result_bootstraps(C.byref(res), C.byref(bs)))
            out.bootstraps = bootstraps_from_c(bs)
        return out

    def infer(self, eq_labels, cell_classes, num_alphas: int, usa_mode: bool = False) -> QuantResult:
        """`alevin-fry infer` (src/infer.rs): eq_labels = list of label lists (global classes); cell_classes = per cell a list of
        (class id, count) with ascending class ids.  Returns the non-zero abundances per cell."""
        lab = np.asarray([x for l in eq_labels for x in l], dtype=np.uint32)
        lp = np.zeros(len(eq_labels) + 1, dtype=np.uint64)
        lp[1:] = np.cumsum([len(l) for l in eq_labels])
        cp = np.zeros(len(cell_classes) + 1, dtype=np.uint64)
        cp[1:] = np.cumsum([len(c) for c in cell_classes])
        ce = np.asarray([e for c in cell_classes for e, _ in c], dtype=np.uint32)
        cc = np.asarray([n for c in cell_classes for _, n in c], dtype=np.uint32)
        u32, u64 = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
        res = AfqResult()
        self._check(self.lib.afq_infer(self._h, lab.ctypes.data_as(u32), lp.ctypes.data_as(u64), len(eq_labels), cp.ctypes.data_as(u64),
                                       ce.ctypes.data_as(u32), cc.ctypes.data_as(u32), len(cell_classes), int(num_alphas), 1 if usa_mode else 0,
                                       C.byref(res)))
        return result_from_c(res, owner=_ResultOwner(self.lib.afq_result_release, res))

    def quant_chunks(self, chunk_bytes, chunk_off, first_cell_index: int = 0) -> QuantResult:
        self.submit(chunk_bytes, chunk_off, first_cell_index)
        return self.collect()

    def kernel_times(self):
        buf = (AfqKernelTime * 64)()
        n = self.lib.afq_get_kernel_times(self._h, buf, 64)
        if n < 0:
            self._check(n)
        return {buf[i].name.decode(): (buf[i].ms, buf[i].launches) for i in range(n)}

    def label_rehash_count(self) -> int:
        """Ranges of cells decoded again under another label-hash function after a collision (parsimony; see afquant.h)."""
        self.lib.afq_label_rehash_count.restype = C.c_uint64
        self.lib.afq_label_rehash_count.argtypes = [C.c_void_p]
        return int(self.lib.afq_label_rehash_count(self._h))

    def pool_regrow_count(self) -> int:
        """Ranges of cells run again with four times the parsimony pool because a cell's graph outgrew it (see afquant.h)."""
        self.lib.afq_pool_regrow_count.restype = C.c_uint64
        self.lib.afq_pool_regrow_count.argtypes = [C.c_void_p]
        return int(self.lib.afq_pool_regrow_count(self._h))

    def mono_cell_count(self) -> int:
        """Parsimony cells resolved by the one-workgroup kernel (afq_mono_cell_count)."""
        self.lib.afq_mono_cell_count.restype = C.c_uint64
        self.lib.afq_mono_cell_count.argtypes = [C.c_void_p]
        return int(self.lib.afq_mono_cell_count(self._h))

    def em_resize_count(self) -> int:
        """Ranges whose EM scratch had to be sized on the host (the device-side plan did not fit what was set aside)."""
        self.lib.afq_em_resize_count.restype = C.c_uint64
        self.lib.afq_em_resize_count.argtypes = [C.c_void_p]
        return int(self.lib.afq_em_resize_count(self._h))

    def batch_stats(self) -> dict:
        s = AfqBatchStats()
        self._check(self.lib.afq_get_batch_stats(self._h, C.byref(s)))
        return {k: int(getattr(s, k)) for k, _ in s._fields_}

    def atac_dedup(self, ref, start, frag_len, cell_ptr):
        ref = np.ascontiguousarray(ref, np.uint32)
        start = np.ascontiguousarray(start, np.uint32)
        frag_len = np.ascontiguousarray(frag_len, np.uint16)
        cell_ptr = np.ascontiguousarray(cell_ptr, np.uint64)
        n_cells = len(cell_ptr) - 1
        o_ptr, o_ref, o_start = C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint32)()
        o_len, o_cnt = C.POINTER(C.c_uint16)(), C.POINTER(C.c_uint16)()
        self._check(self.lib.afq_atac_dedup(
            self._h, ref.ctypes.data_as(C.POINTER(C.c_uint32)), start.ctypes.data_as(C.POINTER(C.c_uint32)),
            frag_len.ctypes.data_as(C.POINTER(C.c_uint16)), cell_ptr.ctypes.data_as(C.POINTER(C.c_uint64)), n_cells,
            C.byref(o_ptr), C.byref(o_ref), C.byref(o_start), C.byref(o_len), C.byref(o_cnt)))
        try:
            ptr = np.ctypeslib.as_array(o_ptr, shape=(n_cells + 1,)).copy()
            n = int(ptr[-1])
            mk = lambda p_, dt: (np.ctypeslib.as_array(p_, shape=(n,)).astype(dt, copy=True) if n else np.zeros(0, dt))
            return ptr, mk(o_ref, np.uint32), mk(o_start, np.uint32), mk(o_len, np.uint16), mk(o_cnt, np.uint16)
        finally:
            for q in (o_ptr, o_ref, o_start, o_len, o_cnt):
                self.lib.afq_free(q)

    def atac_dedup_rad(self, chunk_bytes, chunk_off, bc_bytes: int = 4, d_ptr: int = 0, n_bytes: int = 0, copy: bool = True):
        """afq_atac_dedup_rad: collated scATAC chunks in (host bytes, or d_ptr/n_bytes for bytes already on the device),
        (cell_ptr, bc, ref, start, frag_len, count, stats dict) out."""
        off = np.ascontiguousarray(chunk_off, dtype=np.uint64)
        n_cells = len(off)
        if d_ptr:
            ptr, nb, on_dev = C.c_void_p(d_ptr), n_bytes, 1
        else:
            b = np.ascontiguousarray(np.frombuffer(chunk_bytes, dtype=np.uint8) if not isinstance(chunk_bytes, np.ndarray) else chunk_bytes)
            ptr, nb, on_dev = b.ctypes.data_as(C.c_void_p), b.nbytes, 0
        o_ptr, o_bc, o_ref, o_start = C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint32)()
        o_len, o_cnt = C.POINTER(C.c_uint16)(), C.POINTER(C.c_uint16)()
        st = _abi.AfqAtacStats()
        self._check(self.lib.afq_atac_dedup_rad(self._h, ptr, nb, off.ctypes.data_as(C.POINTER(C.c_uint64)), n_cells, bc_bytes, on_dev,
                                                C.byref(o_ptr), C.byref(o_bc), C.byref(o_ref), C.byref(o_start), C.byref(o_len), C.byref(o_cnt), C.byref(st)))
        cp = np.ctypeslib.as_array(o_ptr, shape=(n_cells + 1,)).copy()
        n = int(cp[-1])
        stats = {k: int(getattr(st, k)) for k, _ in st._fields_}
        if not copy:   # views over the library's (pinned) arrays, handed back to it when the last view dies
            import weakref

            lib, ptrs = self.lib, (o_ptr, o_bc, o_ref, o_start, o_len, o_cnt)

            class _Own:
                pass

            own = _Own()
            weakref.finalize(own, lambda: [lib.afq_free(q) for q in ptrs])

            def view(p_, k):
                a = np.ctypeslib.as_array(p_, shape=(max(k, 1),))[:k]
                a.flags.writeable = False
                keep = np.ndarray.__new__(_Held, a.shape, a.dtype, a, 0, a.strides)
                keep._own = own
                return keep

            return cp, view(o_bc, n_cells), view(o_ref, n), view(o_start, n), view(o_len, n), view(o_cnt, n), stats
        try:
            mk = lambda p_, dt, k: (np.ctypeslib.as_array(p_, shape=(k,)).astype(dt, copy=True) if k else np.zeros(0, dt))
            return cp, mk(o_bc, np.uint64, n_cells), mk(o_ref, np.uint32, n), mk(o_start, np.uint32, n), mk(o_len, np.uint16, n), mk(o_cnt, np.uint16, n), stats
        finally:
            for q in (o_ptr, o_bc, o_ref, o_start, o_len, o_cnt):
                self.lib.afq_free(q)
