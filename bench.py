#!/usr/bin/env python
"""bench.py — M reads/s through the `quant` hot path on N MI355X GPUs (one process per GPU).

A "step" is one pass of the whole hot path (afq_submit_device + afq_collect: decode -> bucket ->
resolve -> extract -> CSR on the host) over one batch of synthetic collated RAD that is already
resident in HBM.  The input is produced on the device itself by the Philox generator of
csrc/afq_synth.hip (same bytes as its host twin, include/afquant_synth.h).

  python bench.py --gpus N [--steps K --warmup W]

With N > 1 and no torchrun environment the script spawns its own N ranks (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_*), one per GPU, RCCL for the timing barrier and the scalar reductions;
launched under torchrun it just joins as a rank.  Headline line = BASELINE.json configs[1]
(PBMC-10k-like, cr-like), one such sample PER RANK (weak scaling: cells are independent, ranks share
nothing on the data path).  The same JSON line carries further legs under "also":
  configs2  configs[2]: USA, parsimony-em on the same cells (N = 1)
  configs1_tail / configs2_tail   the same two configurations under the label-length tail model (--na-model tail: E[na] ~ 3,
            labels of 5..30 refs on gene families), oracle-sampled, with the time per input byte against the plain model (N = 1)
  configs3  configs[3]: the ~10^6-cell x 2*10^4-read data set, cells range-sharded over the ranks
            (each rank generates and quantifies ITS byte-balanced range: 125 000 cells per GPU; every rank checks a
            sample of its shard against the oracle)
  atac      configs[4]: scATAC fragment de-duplication from collated-RAD bytes, 10^4 cells x 2*10^4 records per GPU
  e2e       the crossing included: input starts in pinned host memory (N = 1)
  cli       `afquant quant` wall on the same input written as a collated RAD directory (N = 1)
  reference the real `alevin-fry quant`, only if a binary is on the box ($ALEVIN_FRY_BIN / PATH)
See DESIGN.md §6 for how roofline / cpu_baseline are built.
"""
import argparse
import importlib
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# (before the HIP runtime comes up with `import torch`: device<->host copies on the DMA engines, never as blit kernels on the compute
#  queue - the rows of a range come back while the next range's kernels run; INTEGRATION.md "runtime settings")
os.environ.setdefault("GPU_FORCE_BLIT_COPY_SIZE", "0")

METRIC = "M reads/s through quant (PUG dedup+eq-class) at 1/2/4/8 GPUs; cells/s"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)   # (one of a process's first three steps can take 7 ms longer than the others - seen in half the runs, whatever the library's switches: profiles/history/run_r04af.sh)
    ap.add_argument("--workload", default="configs1", choices=["configs1", "configs2", "configs3", "atac"],
                    help="what the headline line measures (default configs1 = the configuration the metric is quoted on)")
    ap.add_argument("--cells", type=int, default=11000, help="cells per GPU of the PBMC-10k-like sample (configs1/2)")
    ap.add_argument("--median-reads", type=float, default=30000.0)
    ap.add_argument("--sigma", type=float, default=0.6)
    ap.add_argument("--genes", type=int, default=36601)
    ap.add_argument("--ref-count", type=int, default=199138, help="spliced transcripts (SURVEY §8d config 2)")
    ap.add_argument("--popularity", default="zipf1.1", choices=["zipf1.1", "pow16"],
                    help="gene popularity: Zipf(1.1) as SURVEY §8d specifies, or the round-1 stress variant floor(G*x^16)")
    ap.add_argument("--na-model", default="plain", choices=["plain", "tail"],
                    help="alignments per read: plain = 1..3 (SURVEY 8d), tail = a geometric run of further refs on the gene's family "
                         "(E[na] ~ 3, labels of 5..30 refs, molecules of more than four genes)")
    ap.add_argument("--usa", action="store_true")
    ap.add_argument("--resolution", default=None, help="override the workload's resolution")
    ap.add_argument("--umi-err", type=float, default=0.01)
    ap.add_argument("--bootstraps", type=int, default=0)
    ap.add_argument("--c3-cells", type=int, default=125000, help="configs3: cells per GPU (x8 = the 10^6-cell data set)")
    ap.add_argument("--c3-mean-reads", type=float, default=20000.0)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="configs3: weak = c3-cells per GPU, strong = c3-cells in total, sharded over the ranks")
    ap.add_argument("--frags-per-cell", type=int, default=20000, help="atac")
    ap.add_argument("--atac-cells", type=int, default=10000, help="atac: cells per GPU (configs[4]: 10^4 x 2*10^4 records)")
    ap.add_argument("--also", default="auto", help="comma list of extra legs (configs2,configs1_tail,configs2_tail,configs3,atac,e2e,cli,cli_sz,cli_pug,reference), 'auto' or 'none'")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU baseline budget (rank 0, N=1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL) by default; gloo + --share-gpu exercises the N>1 logic on a 1-GPU box")
    ap.add_argument("--share-gpu", action="store_true", help="testing only: every rank uses cuda:0")
    ap.add_argument("--plan-only", action="store_true",
                    help="no kernels: every rank prints the device bytes each leg would hold against the free memory of its GPU, the host its pinned "
                         "total against MemAvailable; exit code 0 when everything fits, 1 otherwise (the first thing to run on an 8-GPU node)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------
def spawn_ranks(args):
    """--gpus N without a launcher: start the N ranks ourselves (what `torch.distributed.run` would do)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    # A rank that dies (no such GPU, hipSetDevice failed, out of memory) must not leave the others waiting at a barrier for the
    # process group's timeout: the first non-zero exit ends the job - the others are stopped by their exact pids.
    rc = 0
    alive = list(procs)
    while alive:
        time.sleep(0.2)
        for p in list(alive):
            r = p.poll()
            if r is None:
                continue
            alive.remove(p)
            if r != 0 and rc == 0:
                rc = r
                print(f"[bench] rank {procs.index(p)} exited with code {r}: stopping the other {len(alive)} rank(s)", file=sys.stderr, flush=True)
                for q in alive:
                    q.terminate()
    sys.exit(rc)


class Dist:
    """Rank bookkeeping + the only collectives the path uses: a barrier and MAX/SUM of a few scalars."""

    def __init__(self, args, torch):
        self.torch = torch
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
        self.backend = args.dist_backend
        ndev = torch.cuda.device_count()
        if self.local_rank >= ndev:
            raise SystemExit(f"rank {self.rank}: no GPU {self.local_rank} on this box ({ndev} visible); --share-gpu shares cuda:0 for testing")
        self.dev = torch.device("cuda", self.local_rank)
        try:
            torch.cuda.set_device(self.local_rank)
            torch.cuda.mem_get_info(self.local_rank)   # (the first call that really talks to the device)
        except Exception as e:   # one clear line instead of a barrier the other ranks would hang in
            raise SystemExit(f"rank {self.rank}: GPU {self.local_rank} cannot be used ({type(e).__name__}: {str(e)[:200]})")
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.backend == "nccl":
                dist.init_process_group(backend="nccl", rank=self.rank, world_size=self.world, device_id=self.dev)
            else:
                dist.init_process_group(backend=self.backend, rank=self.rank, world_size=self.world)
            self.dist = dist
        self.rdev = self.dev if self.backend == "nccl" else torch.device("cpu")

    def sync(self):
        self.torch.cuda.synchronize(self.dev)
        if self.dist:
            self.dist.barrier()

    def reduce(self, vals, op):
        if not self.dist:
            return [float(v) for v in vals]
        t = self.torch.tensor([float(v) for v in vals], dtype=self.torch.float64, device=self.rdev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX if op == "max" else self.dist.ReduceOp.SUM)
        return [float(x) for x in t.cpu()]

    def close(self):
        if self.dist:
            self.dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------
TAIL_P = 0.65   # P(one more ref) of the tail model: E[na] ~ 3


def synth_kw(args, usa, tail=None):
    kw = dict(median_reads=args.median_reads, sigma=args.sigma, num_genes=args.genes, usa=usa, umi_err=args.umi_err,
              ref_count=args.ref_count if args.ref_count >= args.genes else 0)
    if (args.na_model == "tail") if tail is None else tail:
        kw.update(tail=TAIL_P, tail_max=64, family=8)
    if args.popularity == "pow16":
        kw.update(zipf=0.0, pow_skew=16.0)
    return kw


def timed_steps(D, q, rad, steps, warmup, first_cell=0):
    """W untimed + exactly K timed steps, barrier + device sync on both sides; returns (elapsed max over ranks, kernel times, last rows)."""
    res = None

    def step():
        nonlocal res
        res = None   # the host has consumed the previous rows: their pinned buffers go back to the library's pool
        q.submit_device(rad.d_ptr, rad.n_bytes, rad.chunk_off, first_cell)
        res = q.collect()

    for _ in range(warmup):
        step()
    D.sync()
    t0 = time.perf_counter()
    ktimes = {}
    trace = os.environ.get("AFQ_BENCH_STEP_TIMES")   # (measurement scripts: every step's own wall time on stderr)
    for _ in range(steps):
        ts = time.perf_counter()
        step()
        if trace:
            print(f"[bench] step {1e3 * (time.perf_counter() - ts):.3f} ms", file=sys.stderr)
        for k, (ms, n) in q.kernel_times().items():   # HIP events on the library's own streams
            a = ktimes.setdefault(k, [0.0, 0])
            a[0] += ms
            a[1] += n
    D.torch.cuda.synchronize(D.dev)
    elapsed = time.perf_counter() - t0
    D.local_elapsed = elapsed   # this rank's own time (the imbalance report compares the ranks' times, not their bytes)
    if D.dist:
        D.dist.barrier()
    elapsed = D.reduce([elapsed], "max")[0]
    return elapsed, ktimes, res


def rows_crc(res):
    """CRC-32 of the last step's rows (row offsets, columns, values, barcodes): one number to compare runs of the same workload under different switches."""
    import zlib

    import numpy as np

    c = 0
    for a in (res.cell_ptr, res.gene, res.val, res.bc):
        c = zlib.crc32(np.ascontiguousarray(a).view(np.uint8), c)
    return c


def sanity(res, rad):
    import numpy as np

    assert res.n_cells == len(rad.cell_nrec) and (np.diff(res.cell_ptr.astype(np.int64)) >= 0).all()
    assert (res.val > 0).all() and float(res.val.sum(dtype=np.float64)) <= rad.n_reads * (1 + 1e-6)
    assert np.array_equal(res.nrec, rad.cell_nrec)


# timers of the library that bracket several kernels: their bytes add up (the decode timer brackets whichever decoder ran)
BRACKETS = {"k_em": ("k_em2_plan", "k_em2_setup", "k_em2_rounds", "k_em2_rounds_hybrid", "k_em", "k_em_rounds"),
            "k_p2_split": ("k_p2_hist", "k_p2_scan", "k_p2_scatter"), "k_p2_search": ("k_p2_search", "k_p2_search_over", "k_p2_check"),
            "k_p2_graph": ("k_pf_count", "k_pf_scan1", "k_pf_scan2", "k_pf_number", "k_pf_union", "k_pf_root", "k_pf_cats", "k_pf_pscan1", "k_pf_pscan2", "k_pf_move", "k_pf_cells",
                           "k_pf_tiles", "k_pf_alloc", "k_pf_place", "k_pc_pairs", "k_pc_lane4", "k_pc_tiny8", "k_pc_mid", "k_pc_resume", "k_pc_finish",
                           "k_p2_graph", "k_p2_cover", "k_p2_tied"), "k_p2_part": ("k_p2_part",), "k_p2_lone": ("k_pl_lone", "k_p2_lone"),
            # (since late round 4 the 5 us kernels around the large ones are timed with them - one HIP event between two timed kernels
            #  instead of two per bracket, csrc/afq_api.cpp: TimerChain - so the decode bracket also holds the proof's fix-up decode,
            #  the scatter bracket k_fix_slabs, the resolve bracket k_resolve_mid / k_resolve_big)
            "k_decode_par": ("k_slab_setup", "k_decode_recs", "k_decode_keys", "k_decode_par", "k_verify_cells", "k_decode"),
            "k_scatter": ("k_scatter", "k_fix_slabs"), "k_resolve": ("k_bucket_desc", "k_resolve", "k_resolve_mid", "k_resolve_big"),
            "k_compact": ("k_row_ptr", "k_compact", "k_compact_em"),
            "k_atac_dedup": ("k_atac_dedup64", "k_atac_dedup"), "k_atac_parse": ("k_atac_parse",)}


def traffic_of(tag, name, launches_per_step):
    """HBM bytes per launch of the timer `name` from the committed PMC passes of the leg `tag` (profiles/rNN[_tag]_traffic.json,
    written by profiles/traffic.py from separate FETCH_SIZE / WRITE_SIZE runs of this very command): 2 x FETCH_SIZE + WRITE_SIZE
    of every kernel the timer brackets, summed over the profiled steps, per launch of the bracket.  None when the leg has no pass."""
    pdir = os.path.join(ROOT, "profiles")
    if tag is None or not os.path.isdir(pdir):
        return None
    suffix = f"_{tag}_traffic.json" if tag else "_traffic.json"
    files = sorted(f for f in os.listdir(pdir) if f.endswith(suffix) and (tag or f.count("_") == 1))
    if not files:
        return None
    d = json.load(open(os.path.join(pdir, files[-1])))
    kern, steps_prof = d["kernels"], d.get("steps_profiled")
    parts = [k for k in BRACKETS.get(name, (name,)) if k in kern]
    if not parts:
        return None
    if steps_prof and all("bytes_total_fetch_doubled" in kern[k] for k in parts):
        return int(sum(kern[k]["bytes_total_fetch_doubled"] for k in parts) / (steps_prof * launches_per_step))
    return sum(kern[k]["bytes_per_launch_fetch_doubled"] for k in parts)   # (files of rounds 1-3: per dispatch of each kernel)


def dominant_member(tag, timer):
    """The kernel, AS ROCPROF NAMES IT, that takes the most time among the kernels the library's timer `timer` brackets, from the
    committed `rocprofv3 --kernel-trace --stats` summary of the leg `tag` (profiles/rNN[_tag]_rocprof.txt).  A reader can then find
    `roofline.kernel` in that file as it stands.  Without a committed summary: the timer's first member, by its plain name."""
    import re

    members = BRACKETS.get(timer, (timer,))
    pdir = os.path.join(ROOT, "profiles")
    if tag is not None and os.path.isdir(pdir):
        suffix = f"_{tag}_rocprof.txt" if tag else "_rocprof.txt"
        files = sorted(f for f in os.listdir(pdir) if re.match(r"r\d\d", f) and f.endswith(suffix) and (tag or f.count("_") == 1))
        if files:
            best, best_us = None, -1.0
            for ln in open(os.path.join(pdir, files[-1])):
                m = re.match(r"^(\S.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", ln)
                if not m:
                    continue
                nm, tot = m.group(1), float(m.group(3))
                if any(nm == k or nm.startswith(k + "<") for k in members) and tot > best_us:
                    best, best_us = nm, tot
            if best:
                return best, files[-1]
    return members[0], None


def roofline_of(ktimes, alg_bytes, steps, traffic_tag, ms_per_step=None):
    """Dominant kernel = largest summed HIP-event time; it is charged with the path's algorithmic bytes of one launch
    (every record once, the chunk headers, 8 B per emitted non-zero).  frac_step = the whole step's algorithmic
    bytes over the step's wall time, as a fraction of HBM peak."""
    if not ktimes:
        return None
    name, (ms_tot, launches) = max(ktimes.items(), key=lambda kv: kv[1][0])
    avg_ms = ms_tot / max(1, launches)
    lps = launches / steps
    achieved = alg_bytes / lps / (avg_ms * 1e-3) / 1e9
    kernels_ms = sum(v[0] for v in ktimes.values()) / steps
    kname, kfile = dominant_member(traffic_tag, name)
    # "kernel": the dominant kernel as the committed rocprof summary names it; "bracket": the library's HIP-event timer the duration
    # comes from (one event between two timed kernels: a timer brackets the large kernel and the 5 us kernels around it) and its members
    return {"bound": "hbm", "kernel": kname, "bracket": {"timer": name, "members": list(BRACKETS.get(name, (name,))), "named_from": kfile}, "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s",
            "frac": round(achieved / 8000.0, 5), "traffic": traffic_of(traffic_tag, name, lps), "avg_launch_ms": round(avg_ms, 4), "launches_per_step": lps,
            "alg_bytes_per_step": alg_bytes,
            "frac_step": round(alg_bytes / (ms_per_step * 1e-3) / 1e9 / 8000.0, 5) if ms_per_step else None,
            "path_achieved_GBps": round(alg_bytes / (kernels_ms * 1e-3) / 1e9, 2) if kernels_ms else None,   # all kernels together
            "frac_path": round(alg_bytes / (kernels_ms * 1e-3) / 1e9 / 8000.0, 5) if kernels_ms else None,    # ... as a fraction of HBM peak: the path's own figure
            "all_kernels_ms_per_step": {k: round(v[0] / steps, 4) for k, v in ktimes.items()}}


def legs_summary(legs):
    """{leg: [value, ms_per_step, roofline.frac, roofline.frac_step]} for the legs that were timed, {leg: "error: ..."} /
    {leg: "skipped: ..."} for the others: what a reader who only has the tail of the line needs of every leg."""
    s = {}
    for name, o in legs.items():
        if not isinstance(o, dict):
            s[name] = None
        elif "error" in o:
            s[name] = "error: " + str(o["error"])[:80]
        elif "skipped" in o:
            s[name] = "skipped: " + str(o["skipped"])[:60]
        else:
            r = o.get("roofline") or {}
            s[name] = [o.get("value"), o.get("ms_per_step", o.get("wall_s")), r.get("frac"), r.get("frac_step")]
    return s


# ---------------------------------------------------------------------------------------------------------
# Where the submitting thread runs.  The host side of a step - planning, the pinned arena, reading the packed status, the row
# pointers - talks to the device over PCIe; from the CPUs of the socket the GPU does not hang off, a configs[1] step was
# 15.9 ms instead of 13.5 (taskset on a two-socket box).  Every rank therefore binds itself to the CPUs of its device's NUMA
# node (what `afquant --devices` does for its worker threads, csrc/afq_host.cpp), and goes back to all CPUs for the oracle legs.
_AFFINITY = {"all": None, "node": None, "id": None}


def bind_to_device_node(torch, dev_index):
    if not hasattr(os, "sched_setaffinity") or os.environ.get("AFQ_BENCH_NO_BIND"):
        return None
    try:
        _AFFINITY["all"] = os.sched_getaffinity(0)
        pr = torch.cuda.get_device_properties(dev_index)
        bus = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        node = int(open(f"/sys/bus/pci/devices/{bus}/numa_node").read())
        if node < 0:
            return None
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= _AFFINITY["all"]
        if cpus:
            os.sched_setaffinity(0, cpus)
            _AFFINITY["node"] = cpus
            _AFFINITY["id"] = node
            return node
    except Exception:
        pass
    return None


class all_cpus:
    """The oracle's worker threads inherit the affinity of the thread that starts them: all CPUs for the CPU legs."""

    def __enter__(self):
        if _AFFINITY["node"]:
            os.sched_setaffinity(0, _AFFINITY["all"])

    def __exit__(self, *exc):
        if _AFFINITY["node"]:
            os.sched_setaffinity(0, _AFFINITY["node"])


EM_FLOOR = 0.01   # em.rs:568-572


def em_row_diff(g0, v0, g1, v1):
    """Two EM rows entry by entry: (entries, common entries beyond 1e-4 relative, entries only one row holds, those of them whose
    survivor is more than 1e-4 above the 0.01 output floor, largest relative difference of the common entries) - tests/util.py has the same."""
    import numpy as np

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


def cpu_leg(*a, **kw):
    with all_cpus():
        return _cpu_leg(*a, **kw)


def _cpu_leg(cfg, rad, res, budget_s, min_cells=0, tie_stats=False, tol=None, n_threads=None, round_cells=None):
    """The oracle (a C++ port, NOT the Rust binary) on a bounded sample of the same cells, one worker thread per host core;
    the sample doubles as a full-size parity check: GPU rows == oracle rows (bit for bit unless tol is given)."""
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as ora

    ora.lib()
    n = len(rad.cell_nrec)
    ncores = n_threads or os.cpu_count() or 1
    k = max(1, n // (round_cells or max(64, 4 * ncores, min_cells)))   # every k-th cell, so the size mix matches the workload
    done_reads, t_cpu, ncell, start = 0, 0.0, 0, 0
    ties = np.zeros(5, np.int64)
    tie_cells = diff_cells = diff_entries = diff_entries_tol = tot_entries = 0
    em_rel, em_flips, em_entries, em_runs = [], 0, 0, 0
    # EM resolutions: the device sums a round's shares in order-free fixed point (csrc/afq_em2.hip).  The timed oracle run is the
    # reference's arithmetic (f32 sums, canonical class order): rows compared entry by entry at north_star's 1e-4 and on which
    # entries are non-zero, the differences reported and GATED round by round (below); every round's cells are also run through the
    # oracle's restatement of the fixed-point arithmetic and compared bit for bit (untimed).
    is_em = cfg.resolution.endswith("-em") and os.environ.get("AFQ_EM_ORDER") != "canonical"
    arith = {"entries": 0, "beyond_1e-4_rel": 0, "across_the_0.01_floor": 0, "across_the_floor_and_more_than_1e-4_above_it": 0, "max_rel_diff": 0.0,
             "cells_bit_identical_to_fixed_point_oracle": 0}
    first = [0, 0, 0, 0]   # all rounds' cells, device against the reference arithmetic: entries, beyond 1e-4, floor crossings, crossings off the floor
    env = [0, 0, 0, 0]     # the same cells, round by round the worst single one of the oracle's three shuffled class orders against its canonical order
    while start < k and (t_cpu < budget_s or ncell < min_cells):
        idx = np.arange(start, n, k)
        start += 1
        data, offs = rad.read_cells(idx)
        tb = time.perf_counter()
        want = ora.quant(cfg, rad.tid_to_gid, data, offs, n_threads=ncores)   # (the TIMED run: no diagnostic mode - the tie statistics below are a run of their own)
        t_cpu += time.perf_counter() - tb
        ps = None
        if tie_stats:   # untimed: tie events per component and the tie-free check (every tie-free component covered a second time, descending)
            diag, ps = ora.quant(cfg, rad.tid_to_gid, data, offs, n_threads=ncores, want_pug_stats=True, check_tie_free=True)
            for j in range(len(idx)):   # (the diagnostic mode changes no row)
                assert np.array_equal(diag.row(j)[0], want.row(j)[0]) and np.array_equal(diag.row(j)[1].view(np.uint32), want.row(j)[1].view(np.uint32))
        done_reads += int(rad.cell_nrec[idx].sum())
        ncell += len(idx)
        fixed = ora.quant(cfg, rad.tid_to_gid, data, offs, n_threads=ncores, em_arith="fixed") if is_em else None   # (every round, untimed)
        rnd = [0, 0, 0, 0]   # this round's cells, device against the reference arithmetic: entries, beyond 1e-4, floor crossings, crossings off the floor
        for j, ci in enumerate(idx):
            g0, v0 = res.row(int(ci))
            g1, v1 = want.row(j)
            if is_em:
                if fixed is not None:
                    g2, v2 = fixed.row(j)
                    assert np.array_equal(g0, g2) and np.array_equal(v0.view(np.uint32), v2.view(np.uint32)), f"GPU / fixed-point oracle mismatch on cell {ci}"
                    arith["cells_bit_identical_to_fixed_point_oracle"] += 1
                e_, far_, cross_, off_, mx_ = em_row_diff(g0, v0, g1, v1)
                arith["entries"] += e_
                arith["across_the_0.01_floor"] += cross_
                arith["across_the_floor_and_more_than_1e-4_above_it"] += off_
                arith["beyond_1e-4_rel"] += far_
                arith["max_rel_diff"] = max(arith["max_rel_diff"], mx_)
                rnd[0] += e_; rnd[1] += far_; rnd[2] += cross_; rnd[3] += off_
                continue
            ok = np.array_equal(g0, g1) and (np.array_equal(v0.view(np.uint32), v1.view(np.uint32)) if tol is None
                                             else np.allclose(v0, v1, rtol=tol, atol=0))
            assert ok, f"GPU/oracle mismatch on cell {ci}"
        if is_em:   # the reference's own envelope on these very cells, EVERY round: three shuffled class orders (em.rs:464 walks a HashMap);
            # what the device is allowed is the WORST SINGLE shuffle's count (the device is one run, not three), round by round
            worst = [0, 0, 0, 0]
            for seed in (11, 12, 13):
                perm = ora.quant(cfg, rad.tid_to_gid, data, offs, n_threads=ncores, em_order_seed=seed)
                em_runs += 1
                one = [0, 0, 0, 0]
                for j in range(len(idx)):
                    g1, v1 = want.row(j)
                    g2, v2 = perm.row(j)
                    e_, far_, cross_, off_, mx_ = em_row_diff(g2, v2, g1, v1)
                    one[0] += e_; one[1] += far_; one[2] += cross_; one[3] += off_
                    em_entries += e_
                    em_flips += cross_
                    em_rel.append(np.array([mx_]))
                worst = [max(a, b) for a, b in zip(worst, one)]
            assert rnd[1] <= worst[1] and rnd[3] <= worst[3], f"EM rows leave the reference's own envelope in round {start}: device {rnd}, worst single shuffle {worst}"
            for i in range(4):
                first[i] += rnd[i]; env[i] += worst[i]
        if tie_stats:
            ties += ps.sum(0).astype(np.int64)
            tie_cells += int((ps[:, 1] > 0).sum())
            other = ora.quant(cfg, rad.tid_to_gid, data, offs, n_threads=ncores, tie_break_descending=True)
            for j in range(len(idx)):
                g1, v1 = want.row(j)
                g2, v2 = other.row(j)
                cols = np.union1d(g1, g2)
                a = np.zeros(len(cols), np.float64)
                b = np.zeros(len(cols), np.float64)
                a[np.searchsorted(cols, g1)] = v1
                b[np.searchsorted(cols, g2)] = v2
                d = a != b
                tot_entries += len(cols)
                diff_cells += bool(d.any())
                diff_entries += int(d.sum())
                diff_entries_tol += int((np.abs(a - b) > 1e-4 * np.maximum(a, b)).sum())
    out = {"value": round(done_reads / t_cpu / 1e6, 4), "unit": "M reads/s", "cores": ncores, "kind": "port",
           "sample": f"{ncell} of {n} cells (every {k}-th, in rounds), {done_reads} reads, {t_cpu:.1f} s, C++ restatement (oracle/) with one "
                     f"worker thread per host core popping whole cells off a shared queue, input already in RAM; rows compared "
                     + ("entry by entry with the GPU's (em_arithmetic)" if is_em else f"{'bit-exact' if tol is None else f'within {tol:g} rel'} with the GPU's")}
    if is_em:
        arith["what"] = ("GPU rows (order-free fixed-point EM sums) against the oracle in the reference's f32 arithmetic, canonical class order: "
                         "north_star allows 1e-4 relative; every round's cells also bit for bit against the oracle's fixed-point restatement")
        out["em_arithmetic"] = arith
        arith["gated_rounds"] = start
        arith["beyond_1e-4_allowed_by_shuffle_envelope"] = env[1]
        arith["floor_crossings_allowed_by_shuffle_envelope"] = env[3]
        arith["floor_crossings_of_the_shuffle_envelope"] = env[2]
        arith["gate"] = ("asserted in EVERY round of sampled cells: the device may have no more entries beyond 1e-4, and no more floor crossings whose survivor is more "
                         "than 1e-4 above 0.01, than the WORST SINGLE one of the oracle's three shuffled class orders produces on the same cells (the maximum over "
                         "the shuffles, not their sum; a crossing AT the floor is within the tolerance); every round's cells are also compared bit for bit with "
                         "the oracle's fixed-point restatement (a self-check)")
        assert first[0] == arith["entries"] and first[1] <= env[1] and first[3] <= env[3], f"EM rows leave the reference's own envelope: {arith}"
    if tie_stats:   # SURVEY §7 hard part 1: how much of the result hangs on the cover's (unpinned) tie-break
        out["parsimony_ties"] = {
            "cells": ncell, "molecules": int(ties[0]), "cells_with_a_tie": tie_cells, "tie_events": int(ties[1]),
            "molecules_in_tied_components": int(ties[3]),
            "tie_free_components_differing": int(ties[4]),   # components without a tie event whose cover changes with the scan order: must be 0
            "vs_descending_tie_break": {"cells_differing": diff_cells, "entries_differing": diff_entries,
                                        "entries_differing_beyond_1e-4_rel": diff_entries_tol, "entries": tot_entries}}
    if em_runs:
        rel = np.concatenate(em_rel) if em_rel else np.zeros(1)
        out["em_order_sensitivity"] = {
            "what": "the oracle's EM with its classes summed in 3 shuffled orders (the reference walks a HashMap, em.rs:464) against the canonical order, every round's cells",
            "shuffles": em_runs, "entries": em_entries, "max_rel_diff": float(rel.max()),
            "entries_beyond_1e-4_rel": env[1], "entries_across_the_0.01_floor": em_flips}
    return out


def line(D, args, workload, value, elapsed, steps, warmup, cfgd, extra):
    return {"metric": METRIC, "value": round(value, 3), "unit": "M reads/s", "n_gpus": D.world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(1e3 * elapsed / steps, 3), "higher_is_better": True,
            "scaling": "strong" if (workload == "configs3" and args.scaling == "strong") else "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic", "config": cfgd, **extra}


# ---------------------------------------------------------------------------------------------------------
def run_pbmc(D, args, pkg, sn, usa, resolution, steps, warmup, cpu_seconds, name, min_cells=0, tie_stats=False, tail=None, round_cells=None):
    """configs[1] / configs[2]: one PBMC-10k-like sample per rank, generated in HBM."""
    import numpy as np

    t0 = time.time()
    tailed = (args.na_model == "tail") if tail is None else tail
    rad = sn.generate_device(device=D.local_rank, seed=2 + D.rank, n_cells=args.cells, **synth_kw(args, usa, tail))
    t_gen = time.time() - t0
    cfg = pkg.WorkerConfig.for_resolution(resolution, usa_mode=rad.usa, num_genes=rad.num_genes, num_rows=rad.num_rows, profile=True,
                                          umi_len=12, num_bootstraps=args.bootstraps, summary_stat=True)   # umi_len: the RAD header's `ulen`
    q = pkg.Quantifier(cfg, rad.tid_to_gid, device=D.local_rank)
    try:
        elapsed, ktimes, res = timed_steps(D, q, rad, steps, warmup)
        total_reads, total_cells = D.reduce([rad.n_reads, args.cells], "sum")
        if D.rank != 0:
            return None, rad, q
        st = q.batch_stats()
        sanity(res, rad)
        nnz = int(res.cell_ptr[-1])
        alg = float(st["input_bytes"]) + 8.0 * nnz
        sizes_default = (args.cells, args.median_reads, args.sigma, args.genes, args.ref_count, args.popularity, args.umi_err) == \
            (11000, 30000.0, 0.6, 36601, 199138, "zipf1.1", 0.01)
        ttag = None   # which committed counter passes belong to this workload
        if sizes_default and not usa and resolution == "cr-like":
            ttag = "configs1_tail" if tailed else ""
        elif sizes_default and usa and resolution == "parsimony-em":
            ttag = "configs2_tail" if tailed else "configs2"
        roof = roofline_of(ktimes, alg, steps, ttag, 1e3 * elapsed / steps)
        cpu = None
        if D.world == 1 and not args.no_cpu_baseline and cpu_seconds > 0:
            cpu = cpu_leg(cfg, rad, res, cpu_seconds, min_cells=min_cells, tie_stats=tie_stats, round_cells=round_cells)
        cfgd = {"workload": f"{name}: PBMC-10k-like 10x-v3 collated RAD, {resolution}, per GPU: {args.cells} cells, log-normal reads/cell "
                            f"median {args.median_reads:g} sigma {args.sigma:g}, {args.genes} genes / {len(rad.tid_to_gid)} transcripts, "
                            f"gene popularity {args.popularity}" + (", USA" if usa else "") +
                            (f", label-length tail (geometric extra refs p={TAIL_P} on gene families of 8, <= 64 refs per read)" if tailed else "") +
                            "; generated in HBM (Philox)",
                "mean_refs_per_read": round((st["input_bytes"] - 8.0 * args.cells) / max(1, rad.n_reads) / 4.0 - 3.0, 3),
                "reads_per_gpu": rad.n_reads, "input_bytes_per_gpu": st["input_bytes"], "resolution": resolution,
                **({"bootstraps": args.bootstraps} if args.bootstraps else {}),
                "sharding": f"{D.world} x independent cell shards, no data-path collective",
                "host_thread": (f"bound to the CPUs of NUMA node {_AFFINITY['id']} (the device's)" if _AFFINITY["id"] is not None else "not bound")}
        out = line(D, args, name, total_reads * steps / elapsed / 1e6, elapsed, steps, warmup, cfgd,
                   {"cells_per_s": round(total_cells * steps / elapsed, 1), "nnz": nnz, "keys": st["n_keys"],
                    "overflow_buckets": st["n_overflow_buckets"],
                    "retries": {"label_rehashes": q.label_rehash_count(), "pool_regrows": q.pool_regrow_count(), "em_resizes": q.em_resize_count(),
                                "cells_through_the_one_workgroup_kernel": q.mono_cell_count(),
                                "what": "ranges run again under another label hash / with a larger parsimony pool, EMs sized on the host after all - since the context was made (warm-up included)"},
                    "gen_seconds": round(t_gen, 2),
                    **({"rows_crc32": rows_crc(res)} if os.environ.get("AFQ_BENCH_CRC") else {}),   # (measurement scripts: the same rows under every switch)
                    "roofline": roof, "cpu_baseline": cpu})
        return out, rad, q
    except BaseException:
        q.close()
        rad.free()
        raise


def run_configs3(D, args, pkg, sn, steps, warmup):
    """configs[3]: ~10^6 cells x 2*10^4 reads, cr-like, cells range-sharded over the ranks.  The data set is one Philox
    stream over GLOBAL cell indices; every rank makes its own byte-balanced range of it directly in HBM."""
    import numpy as np

    shard = importlib.import_module("alevin-fry_amd.shard")
    n_total = args.c3_cells * (D.world if args.scaling == "weak" else 1)
    sigma = 0.6
    median = args.c3_mean_reads / float(np.exp(sigma * sigma / 2))   # log-normal with the asked-for MEAN
    kw = synth_kw(args, False)
    kw.update(median_reads=median, sigma=sigma)
    p = sn.params(seed=4, n_cells=n_total, **kw)
    sizes = sn.cell_sizes(p)
    # records are 12 + 4*na bytes with the same na mix in every cell: balancing reads balances bytes (the ranges a host
    # would cut from the chunk table of a file, shard.shard_ranges on chunk bytes)
    c0, c1 = shard.shard_ranges(sizes.astype(np.float64) * 17.5 + 8, D.world)[D.rank]
    t0 = time.time()
    rad = sn.generate_device(device=D.local_rank, p=p, sizes=sizes, cell_range=(c0, c1))
    D.torch.cuda.synchronize(D.dev)
    t_gen = time.time() - t0
    cfg = pkg.WorkerConfig.for_resolution("cr-like", num_genes=rad.num_genes, num_rows=rad.num_rows, profile=True, umi_len=12)
    q = pkg.Quantifier(cfg, rad.tid_to_gid, device=D.local_rank)
    try:
        elapsed, ktimes, res = timed_steps(D, q, rad, steps, warmup, first_cell=c0)
        sanity(res, rad)
        assert res.first_cell_index == c0
        st = q.batch_stats()
        nnz = int(res.cell_ptr[-1])
        tot_reads, tot_cells, tot_bytes, tot_nnz = D.reduce([rad.n_reads, c1 - c0, st["input_bytes"], nnz], "sum")
        max_bytes = D.reduce([st["input_bytes"]], "max")[0]
        t_max, t_sum = D.reduce([D.local_elapsed], "max")[0], D.reduce([D.local_elapsed], "sum")[0]
        # parity at the shard's size: the oracle on every k-th cell of THIS rank's shard (each rank checks its own), rows
        # compared bit for bit; rank 0's timing is the reported cpu_baseline
        cpu = None
        if not args.no_cpu_baseline and args.cpu_seconds > 0:
            cpu = cpu_leg(cfg, rad, res, min(args.cpu_seconds, 8.0), min_cells=500,
                          n_threads=max(1, (os.cpu_count() or 1) // D.world))
            if D.world > 1:
                cpu["sample"] += f"; every one of the {D.world} ranks checked its own shard this way ({cpu['cores']} threads each)"
        checked = D.reduce([1.0 if cpu else 0.0], "sum")[0]
        if D.rank != 0:
            return None
        if cpu:
            cpu["ranks_checked"] = int(checked)
        alg = float(st["input_bytes"]) + 8.0 * nnz
        cfgd = {"workload": f"configs[3]: synthetic 10x-v3 collated RAD, {int(tot_cells)} cells x ~{args.c3_mean_reads:g} reads/cell (log-normal, "
                            f"sigma {sigma:g}, largest first), cr-like, cells range-sharded by bytes over {D.world} GPU(s) "
                            f"({'the full 10^6-cell set is 8 such shards' if args.scaling == 'weak' else 'fixed total'}); generated in HBM shard by shard (Philox, global cell index)",
                "cells": int(tot_cells), "reads": int(tot_reads), "input_bytes": int(tot_bytes), "rank0_cells": [int(c0), int(c1)],
                "imbalance_max_over_mean_bytes": round(max_bytes / (tot_bytes / D.world), 4),
                "imbalance_max_over_mean_time": round(t_max / (t_sum / D.world), 4), "resolution": "cr-like",
                "sharding": f"{D.world} contiguous cell ranges, no data-path collective"}
        return line(D, args, "configs3", tot_reads * steps / elapsed / 1e6, elapsed, steps, warmup, cfgd,
                    {"cells_per_s": round(tot_cells * steps / elapsed, 1), "nnz": int(tot_nnz), "gen_seconds": round(t_gen, 2),
                     "retries": {"label_rehashes": q.label_rehash_count(), "pool_regrows": q.pool_regrow_count(), "em_resizes": q.em_resize_count()},
                     "roofline": roofline_of(ktimes, alg, steps, "configs3" if (args.c3_cells, args.c3_mean_reads, args.genes, args.ref_count) == (125000, 20000.0, 36601, 199138) else None,
                                             1e3 * elapsed / steps), "cpu_baseline": cpu})
    finally:
        q.close()
        rad.free()


def run_e2e(D, args, pkg, sn, rad, q, steps):
    """The crossing included: the input starts in pinned host memory, afq_submit brings it over range by range while the
    earlier ranges already run (csrc/afq_api.cpp)."""
    torch = D.torch
    host = torch.empty(rad.n_bytes, dtype=torch.uint8, pin_memory=True)
    rc = sn._lib().afq_synth_device_read(D.local_rank, rad.d_ptr, rad.n_bytes, host.data_ptr())
    assert rc == 0
    res = None
    for _ in range(1):
        q.submit_ptr(host.data_ptr(), rad.n_bytes, rad.chunk_off)
        res = q.collect()
    torch.cuda.synchronize(D.dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        res = None
        q.submit_ptr(host.data_ptr(), rad.n_bytes, rad.chunk_off)
        res = q.collect()
    torch.cuda.synchronize(D.dev)
    dt = (time.perf_counter() - t0) / steps
    sanity(res, rad)
    return {"what": "afq_submit from pinned host memory + afq_collect: H2D of the RAD bytes inside the step, overlapped with the kernels",
            "value": round(rad.n_reads / dt / 1e6, 3), "unit": "M reads/s", "ms_per_step": round(dt * 1e3, 3),
            "h2d_GBps_effective": round(rad.n_bytes / dt / 1e9, 2), "steps": steps}, host


def run_e2e_ranks(D, args, pkg, sn, rad, q, steps):
    """The e2e leg at N > 1: every rank brings its own sample over from its own pinned host buffer at the same time (the
    ranks' PCIe links and the host's memory controllers are shared resources a one-rank run does not see).  Every rank makes
    the same collective calls whatever happens to it; a failing rank turns the leg into an error on all of them."""
    torch = D.torch
    err, host, res, dt = None, None, None, 0.0
    try:
        host = torch.empty(rad.n_bytes, dtype=torch.uint8, pin_memory=True)
        assert sn._lib().afq_synth_device_read(D.local_rank, rad.d_ptr, rad.n_bytes, host.data_ptr()) == 0
        q.submit_ptr(host.data_ptr(), rad.n_bytes, rad.chunk_off)
        res = q.collect()
    except Exception as e:
        err = e
    D.sync()
    t0 = time.perf_counter()
    try:
        if err is None:
            for _ in range(steps):
                res = None
                q.submit_ptr(host.data_ptr(), rad.n_bytes, rad.chunk_off)
                res = q.collect()
            torch.cuda.synchronize(D.dev)
            dt = (time.perf_counter() - t0) / steps
            sanity(res, rad)
    except Exception as e:
        err = e
    D.sync()
    mx = D.reduce([dt, 1.0 if err is not None else 0.0], "max")
    tot = D.reduce([rad.n_reads, rad.n_bytes, dt], "sum")
    if mx[1] > 0:
        raise RuntimeError(f"e2e leg failed on a rank ({type(err).__name__ if err else 'another rank'}: {err})")
    return {"what": "afq_submit from pinned host memory + afq_collect on every rank at once: H2D of each rank's RAD bytes inside the step",
            "value": round(tot[0] / mx[0] / 1e6, 3), "unit": "M reads/s", "ms_per_step": round(mx[0] * 1e3, 3),
            "h2d_GBps_effective_total": round(tot[1] / mx[0] / 1e9, 2), "n_gpus": D.world,
            "imbalance_max_over_mean_time": round(mx[0] / (tot[2] / D.world), 3), "steps": steps}


def write_rad_dir(pkg, rad, host_bytes, path, n_cells=None, compressed=False):
    names = [f"t{i}" for i in range(len(rad.tid_to_gid))]
    if rad.usa:
        rows = [(names[i], f"g{int(g) >> 1}", "U" if int(g) & 1 else "S") for i, g in enumerate(rad.tid_to_gid)]
    else:
        rows = [(names[i], f"g{int(g)}") for i, g in enumerate(rad.tid_to_gid)]
    return pkg.rad.write_quant_input_dir(path, host_bytes, len(rad.cell_nrec) if n_cells is None else n_cells, names, rows, cblen=16, ulen=12,
                                         compressed=compressed)


def run_cli(pkg, rad, host_np, workdir, resolution="cr-like", n_cells=None, compressed=False, sub="in"):
    """`afquant quant` on the same cells written as a collated-RAD directory: process start to the last output file.
    n_cells: only the first n_cells chunks (host_np holds exactly their bytes); compressed: map.collated.rad.sz (snappy frames)."""
    d = os.path.join(workdir, sub)
    o = os.path.join(workdir, sub + "_out")
    t0 = time.time()
    tg = write_rad_dir(pkg, rad, host_np, d, n_cells=n_cells, compressed=compressed)
    t_write = time.time() - t0
    n_reads = rad.n_reads if n_cells is None else int(rad.cell_nrec[:n_cells].astype("int64").sum())
    exe = os.path.join(ROOT, "alevin-fry_amd", "csrc", "afquant")
    nt = str(os.cpu_count() or 1)
    best = None
    os.sync()            # the 6.9 GB just written are dirty pages: without this their write-back competes with the first runs' reads
    for _ in range(3):   # best of three: page cache warm, as a pipeline that has just written the file would find it
        shutil.rmtree(o, ignore_errors=True)
        t0 = time.perf_counter()
        with all_cpus():   # (a process of its own: it places its threads itself)
            p = subprocess.run([exe, "quant", "-i", d, "-m", tg, "-o", o, "-r", resolution, "-t", nt], capture_output=True, text=True)
        dt = time.perf_counter() - t0
        if p.returncode != 0:
            return {"error": p.stderr[-400:]}, d, tg
        if os.environ.get("AFQ_HOST_TIMING"):
            sys.stderr.write(p.stderr)
        best = dt if best is None else min(best, dt)
    return {"what": f"afquant quant -r {resolution} -t {nt}: wall from process start to the last output file, map.collated.rad{'.sz (snappy frames, undone by the host threads)' if compressed else ''} in the page cache (best of 3)"
                    + (f"; the first {n_cells} cells of the sample" if n_cells is not None else ""),
            "wall_s": round(best, 3), "value": round(n_reads / best / 1e6, 3), "unit": "M reads/s", "reads": n_reads,
            "rad_bytes": int(len(host_np)), "write_input_s": round(t_write, 1)}, d, tg


def run_reference_binary(rad_dir, tg, rad, workdir, res_rows=None):
    """If the real alevin-fry is on this box: time its quant on the same directory and diff the counts keyed by
    (barcode, gene) (scripts/testing/compare_counts.py of the reference does the same through pyroe)."""
    exe = os.environ.get("ALEVIN_FRY_BIN") or shutil.which("alevin-fry")
    if not exe:
        return None
    import numpy as np

    o = os.path.join(workdir, "ref_out")
    nt = os.cpu_count() or 1
    t0 = time.perf_counter()
    with all_cpus():
        p = subprocess.run([exe, "quant", "-i", rad_dir, "-m", tg, "-o", o, "-r", "cr-like", "-t", str(nt), "--use-mtx"], capture_output=True, text=True)
    dt = time.perf_counter() - t0
    if p.returncode != 0:
        return {"error": (p.stderr or p.stdout)[-400:]}
    out = {"value": round(rad.n_reads / dt / 1e6, 3), "unit": "M reads/s", "cores": nt, "kind": "reference",
           "sample": f"`{exe} quant -r cr-like -t {nt}` on the whole input, wall {dt:.1f} s (reads the RAD from the page cache, writes MTX)"}
    try:
        ours = os.path.join(workdir, "out", "alevin")
        theirs = os.path.join(o, "alevin")

        def load(dirp):
            rows = [l.strip() for l in open(os.path.join(dirp, "quants_mat_rows.txt"))]
            cols = [l.strip() for l in open(os.path.join(dirp, "quants_mat_cols.txt"))]
            m = np.loadtxt(os.path.join(dirp, "quants_mat.mtx"), comments="%", skiprows=0)
            m = m[1:]   # size line
            return {(rows[int(r) - 1], cols[int(c) - 1]): float(v) for r, c, v in m}
        a, b = load(ours), load(theirs)
        keys = set(a) | set(b)
        nd = sum(1 for k in keys if a.get(k, 0.0) != b.get(k, 0.0))
        out["count_diff"] = {"entries": len(keys), "differing": nd}
    except Exception as e:   # the diff is a bonus; the timing stands
        out["count_diff"] = {"error": repr(e)[:200]}
    return out


# ---------------------------------------------------------------------------------------------------------
def plan_only(D, args):
    """bench.py --gpus N --plan-only: what every leg of this command would hold on this rank's GPU and in pinned host memory,
    against what is there - without generating a byte or launching a kernel.  The input sizes are exact (the generator's own
    planner, afq_synth_host_plan, over the cells this rank would make); the library's side mirrors plan_ranges (csrc/afq_api.cpp):
    it sizes a range to 40 % of the memory that is free once the input is resident and keeps two range buffer sets alive, so a leg
    fits when its input leaves room for its largest cell's buffers twice over - the figure printed is the library's need for the
    WHOLE leg (what it would take if memory were no object) and what it will actually take.  Every rank prints one JSON line; rank 0
    adds the host's pinned total against MemAvailable.  Returns the exit code: 0 = every leg fits on every rank."""
    import ctypes as C

    import numpy as np

    sn = importlib.import_module("alevin-fry_amd.synth_native")
    shard = importlib.import_module("alevin-fry_amd.shard")
    free_b, total_b = D.torch.cuda.mem_get_info(D.local_rank)
    lib = sn._lib()

    def planned(p, sizes, c0, c1):
        nrec = np.ascontiguousarray(sizes[c0:c1])
        off = np.zeros(len(nrec), np.uint64)
        tb = C.c_uint64()
        r = lib.afq_synth_host_plan(C.byref(p), c0, len(nrec), nrec.ctypes.data_as(C.POINTER(C.c_uint32)), off.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(tb))
        if r:
            raise RuntimeError(f"afq_synth_host_plan failed ({r})")
        reads = int(nrec.astype(np.int64).sum())
        n_ref = (int(tb.value) - 8 * len(nrec) - 12 * reads) // 4     # records are 12 header bytes + 4 per alignment
        return int(tb.value), reads, n_ref, int(nrec.max()) if len(nrec) else 0

    def lib_need(reads, n_ref, usa, resolution):   # plan_ranges' `need`, summed over the leg's cells
        em = resolution.endswith("-em")
        pug = resolution.startswith("parsimony")
        nd = ((24.0 + 40.0 * (3 if usa else 1)) if em else 16.0) * n_ref
        if pug:
            nd += 148.0 * reads
        return nd * 1.1   # (+ bucket tables and slabs: a few per cent)

    legs = {}
    names = [args.workload] + ([] if args.also == "none" else (["configs2", "configs1_tail", "configs2_tail", "configs3", "atac", "e2e"] if D.world == 1 else ["e2e", "configs3", "atac"]))
    pinned = 0
    for name in dict.fromkeys(names):
        if name == "atac":
            inp = int(args.atac_cells) * int(args.frags_per_cell) * 19   # ~19 bytes per record of the scATAC chunks
            need = 2.0 * inp
            legs[name] = {"input_bytes": inp, "reads": int(args.atac_cells) * int(args.frags_per_cell), "library_need_bytes": int(need)}
            continue
        if name == "e2e":
            base = legs.get("configs1")
            if base:
                legs[name] = {"input_bytes": 0, "reads": base["reads"], "library_need_bytes": base["library_need_bytes"], "pinned_host_bytes": base["input_bytes"]}
                pinned = base["input_bytes"]
            continue
        if name == "configs3":
            n_total = args.c3_cells * (D.world if args.scaling == "weak" else 1)
            sigma = 0.6
            kw = synth_kw(args, False)
            kw.update(median_reads=args.c3_mean_reads / float(np.exp(sigma * sigma / 2)), sigma=sigma)
            p = sn.params(seed=4, n_cells=n_total, **kw)
            sizes = sn.cell_sizes(p)
            c0, c1 = shard.shard_ranges(sizes.astype(np.float64) * 17.5 + 8, D.world)[D.rank]
            usa, res = False, "cr-like"
        else:
            tail = name.endswith("_tail") or (name == args.workload and args.na_model == "tail")
            usa = name.startswith("configs2") or (name == args.workload and args.usa)
            res = "parsimony-em" if name.startswith("configs2") else (args.resolution or "cr-like")
            p = sn.params(seed=2 + D.rank, n_cells=args.cells, **synth_kw(args, usa, tail))
            sizes = sn.cell_sizes(p)
            c0, c1 = 0, args.cells
        inp, reads, n_ref, largest = planned(p, sizes, c0, c1)
        legs[name] = {"input_bytes": inp, "reads": reads, "library_need_bytes": int(lib_need(reads, n_ref, usa, res)), "resolution": res}
    ok = True
    for name, leg in legs.items():
        room = free_b - leg["input_bytes"]
        takes = min(leg["library_need_bytes"], int(0.8 * max(0, room)))          # two range buffer sets of <= 40 % of what is free each
        leg["resident_peak_bytes"] = leg["input_bytes"] + takes
        leg["ranges_at_least"] = max(1, int(np.ceil(leg["library_need_bytes"] / max(1.0, 0.4 * room)))) if room > 0 else None
        leg["fits"] = bool(room > 0 and leg["library_need_bytes"] / max(1, leg.get("reads", 1)) * 300000 < 0.4 * room)   # a 300 k-read cell's buffers fit one set
        ok = ok and leg["fits"]
    out = {"plan_only": True, "rank": D.rank, "world": D.world, "device": D.local_rank, "device_free_bytes": int(free_b), "device_total_bytes": int(total_b),
           "legs": legs, "fits": ok}
    pinned_all = D.reduce([pinned + 0.3e9], "sum")[0]   # + every rank's pinned result pool
    all_ok = D.reduce([1.0 if ok else 0.0], "sum")[0] == D.world
    if D.rank == 0:
        avail = None
        try:
            for ln in open("/proc/meminfo"):
                if ln.startswith("MemAvailable:"):
                    avail = int(ln.split()[1]) * 1024
        except OSError:
            pass
        out["host"] = {"pinned_bytes_all_ranks": int(pinned_all), "mem_available_bytes": avail, "fits": bool(avail is None or pinned_all < 0.8 * avail)}
        all_ok = all_ok and out["host"]["fits"]
    print(json.dumps(out), flush=True)
    D.close()
    return 0 if all_ok else 1


# ---------------------------------------------------------------------------------------------------------
def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args)
    import numpy as np  # noqa: F401
    import torch  # (imported before libafquant.so: one HIP runtime per process)

    D = Dist(args, torch)
    if args.plan_only:
        sys.exit(plan_only(D, args))
    numa_node = bind_to_device_node(torch, D.local_rank)
    pkg = importlib.import_module("alevin-fry_amd")
    sn = importlib.import_module("alevin-fry_amd.synth_native")
    if args.workload == "atac":
        return bench_atac(args, pkg, D)
    also = args.also.split(",") if args.also not in ("auto", "none") else \
        ([] if args.also == "none" else (["configs2", "configs1_tail", "configs2_tail", "configs3", "atac", "e2e", "cli", "cli_sz", "cli_pug", "reference"] if D.world == 1 else ["e2e", "configs3", "atac"]))
    also = [a for a in also if a and a != args.workload]
    legs = {}
    out = None
    rad = q = None
    try:
        if args.workload == "configs3":
            out = run_configs3(D, args, pkg, sn, args.steps, args.warmup)
        else:
            usa = args.usa or args.workload == "configs2"
            resolution = args.resolution or ("parsimony-em" if args.workload == "configs2" else "cr-like")
            name = "configs[2]" if (usa and resolution == "parsimony-em") else "configs[1]"
            out, rad, q = run_pbmc(D, args, pkg, sn, usa, resolution, args.steps, args.warmup, args.cpu_seconds, name,
                                   min_cells=200 if name == "configs[2]" else 0, tie_stats=name == "configs[2]")

        def leg(name, fn):
            try:
                legs[name] = fn()
            except Exception as e:   # an extra leg must never take the headline line down with it
                legs[name] = {"error": f"{type(e).__name__}: {e}"[:300]}

        host = None
        workdir = None
        if rad is not None and D.world == 1 and args.workload == "configs1" and not args.usa and (args.resolution in (None, "cr-like")):
            if "e2e" in also:
                def f():
                    nonlocal host
                    r, host = run_e2e(D, args, pkg, sn, rad, q, max(2, args.steps))
                    return r
                leg("e2e", f)
            if "cli" in also or "reference" in also:
                workdir = tempfile.mkdtemp(prefix="afq_bench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
                host_np = host.numpy() if host is not None else rad.to_host()
                cli_state = {}

                def f():
                    r, d, tg = run_cli(pkg, rad, host_np, workdir)
                    cli_state.update(d=d, tg=tg)
                    return r
                leg("cli", f)
                if "cli_sz" in also:   # the compressed front end (src/quant.rs:373-395): the first tenth of the sample as snappy frames
                    def fz():
                        nz = max(1, len(rad.cell_nrec) // 10)
                        nbz = int(rad.chunk_off[nz]) if nz < len(rad.cell_nrec) else rad.n_bytes
                        r, _, _ = run_cli(pkg, rad, host_np[:nbz], workdir, n_cells=nz, compressed=True, sub="in_sz")
                        shutil.rmtree(os.path.join(workdir, "in_sz"), ignore_errors=True)
                        return r
                    leg("cli_sz", fz)
                if "reference" in also and cli_state:
                    leg("reference", lambda: run_reference_binary(cli_state["d"], cli_state["tg"], rad, workdir))
                    if legs.get("reference") is None:
                        legs["reference"] = {"skipped": "no alevin-fry binary on this box ($ALEVIN_FRY_BIN / PATH); cpu_baseline is the C++ port"}
                    elif "value" in legs["reference"] and out is not None:
                        out["cpu_baseline_reference"] = legs["reference"]
        if rad is not None and D.world > 1 and "e2e" in also and args.workload == "configs1" and not args.usa and (args.resolution in (None, "cr-like")):
            r_e2e = None
            try:
                r_e2e = run_e2e_ranks(D, args, pkg, sn, rad, q, max(2, min(5, args.steps)))
            except Exception as e:   # (raised on every rank alike: the collectives inside have all been made)
                r_e2e = {"error": f"{type(e).__name__}: {e}"[:300]}
            if D.rank == 0:
                legs["e2e"] = r_e2e
        if q is not None:
            q.close()
            q = None
        if rad is not None:
            rad.free()
            rad = None
        host = None
        if workdir:
            shutil.rmtree(workdir, ignore_errors=True)
        if "configs2" in also and D.world == 1:
            def f():
                o2, r2, q2 = run_pbmc(D, args, pkg, sn, True, "parsimony-em", max(1, min(5, args.steps)), 1,
                                      min(args.cpu_seconds, 6.0), "configs[2]", min_cells=200, tie_stats=True)
                q2.close()
                if "cli_pug" in also and o2:   # the front end on the USA sample: afquant quant -r parsimony-em
                    wd = tempfile.mkdtemp(prefix="afq_bench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
                    try:
                        rc, _, _ = run_cli(pkg, r2, r2.to_host(), wd, resolution="parsimony-em")
                        legs["cli_pug"] = rc
                    except Exception as e:
                        legs["cli_pug"] = {"error": f"{type(e).__name__}: {e}"[:300]}
                    finally:
                        shutil.rmtree(wd, ignore_errors=True)
                r2.free()
                return o2
            leg("configs2", f)
        for tleg, tusa, tres in (("configs1_tail", False, "cr-like"), ("configs2_tail", True, "parsimony-em")):
            if tleg in also and D.world == 1:
                def f(tusa=tusa, tres=tres, tleg=tleg):
                    o2, r2, q2 = run_pbmc(D, args, pkg, sn, tusa, tres, max(1, min(5, args.steps)), 1, min(args.cpu_seconds, 4.0),
                                          "configs[2]" if tusa else "configs[1]", min_cells=100, tail=True,
                                          round_cells=256 if tusa else None)   # the oracle is ~8x slower on tailed parsimony cells
                    q2.close()
                    r2.free()
                    base = out if not tusa else legs.get("configs2")
                    if o2 and base and "ms_per_step" in base:   # time per input byte against the plain model of the same configuration
                        o2["slowdown_per_input_byte_vs_plain"] = round((o2["ms_per_step"] / o2["config"]["input_bytes_per_gpu"]) /
                                                                       (base["ms_per_step"] / base["config"]["input_bytes_per_gpu"]), 3)
                    return o2
                leg(tleg, f)
        if "configs3" in also:
            r3 = None
            try:
                r3 = run_configs3(D, args, pkg, sn, max(1, min(2, args.steps)), 1)
            except Exception as e:
                if D.world > 1:
                    raise   # all ranks take part in its collectives: no way to carry on half-way
                r3 = {"error": f"{type(e).__name__}: {e}"[:300]}
            if D.rank == 0:
                legs["configs3"] = r3
        if "atac" in also:
            r4 = None
            try:
                r4 = run_atac(args, pkg, D, max(1, min(3, args.steps)), 1)
            except Exception as e:
                if D.world > 1:
                    raise
                r4 = {"error": f"{type(e).__name__}: {e}"[:300]}
            if D.rank == 0:
                legs["atac"] = r4
        if D.rank == 0 and out is not None:
            if legs:
                out["also"] = legs
                # The line is ~20 KB and a log keeps its tail: the last key is a few hundred bytes with every leg's
                # [value (M reads/s, atac: M fragments/s), ms per step, roofline.frac, roofline.frac_step] - or its error / skip reason.
                out["legs"] = legs_summary(legs)
            out["runtime_settings"] = {"GPU_FORCE_BLIT_COPY_SIZE": os.environ.get("GPU_FORCE_BLIT_COPY_SIZE"),
                                       "what": "0 = device<->host copies on the DMA engines; set by this script before the HIP runtime came up unless the caller's environment said otherwise"}
            if legs:
                out["legs"] = out.pop("legs")   # (stays the last key)
            print(json.dumps(out), flush=True)
    finally:
        if q is not None:
            q.close()
        if rad is not None:
            rad.free()
        D.close()


def run_atac(args, pkg, D, steps, warmup):
    """BASELINE configs[4]: scATAC fragment / barcode de-duplication from collated-RAD bytes (afq_atac_dedup_rad): the record
    walk with the na == 1 && type == 4 filter, the per-cell sort and the run-length count all on the device.  The chunk
    bytes are resident in HBM when the timed region starts (uploaded once); a step ends with the distinct fragments on the
    host.  The kernels' own times come from the library's HIP-event timers.  Returns the leg's JSON object (rank 0)."""
    import numpy as np

    sn = importlib.import_module("alevin-fry_amd.synth_native")
    torch = D.torch
    n_cells = args.atac_cells
    per = args.frags_per_cell
    t0 = time.time()
    data, off = sn.generate_atac(seed=5 + D.rank, n_cells=n_cells, frags_per_cell=per)
    t_gen = time.time() - t0
    n = n_cells * per
    d_bytes = torch.from_numpy(data).to(D.dev)
    cfg = pkg.WorkerConfig.for_resolution("cr-like", num_genes=1, num_rows=1, profile=True)
    q = pkg.Quantifier(cfg, np.zeros(1, np.uint32), device=D.local_rank)
    res = None

    def step():
        nonlocal res
        res = None   # (the previous step's arrays go back to the library's pinned pool)
        res = q.atac_dedup_rad(None, off, d_ptr=d_bytes.data_ptr(), n_bytes=len(data), copy=False)

    try:
        for _ in range(warmup):
            step()
        D.sync()
        t0 = time.perf_counter()
        kt = {}
        for _ in range(steps):
            step()
            for k, (ms, nl) in q.kernel_times().items():
                a = kt.setdefault(k, [0.0, 0])
                a[0] += ms
                a[1] += nl
        torch.cuda.synchronize(D.dev)
        elapsed = time.perf_counter() - t0
        if D.dist:
            D.dist.barrier()
        elapsed = D.reduce([elapsed], "max")[0]
        total = D.reduce([float(n)], "sum")[0]
        if D.rank != 0:
            return None
        distinct = int(res[0][-1])
        alg = float(len(data)) + 12.0 * distinct   # every record byte once; (ref, start, len, count) per distinct fragment out
        cpu = None
        if D.world == 1 and not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle as ora
            from concurrent.futures import ThreadPoolExecutor

            # the oracle on the first k cells, one slice of cells per host core (ctypes drops the GIL inside the call); every field
            # of every distinct fragment compared with the GPU's
            ncores = os.cpu_count() or 1
            k = max(1, min(n_cells, int(40e6 // per)))
            bounds = np.linspace(0, k, min(ncores, k) + 1).astype(np.int64)
            ends = [int(off[c]) if c < n_cells else len(data) for c in bounds]

            def part(i):
                c0, c1 = int(bounds[i]), int(bounds[i + 1])
                return ora.atac_dedup_rad(data[int(off[c0]):ends[i + 1]], off[c0:c1] - off[c0])
            tb = time.perf_counter()
            with all_cpus(), ThreadPoolExecutor(max_workers=ncores) as ex:
                parts = list(ex.map(part, range(len(bounds) - 1)))
            tc = time.perf_counter() - tb
            pos = 0
            for i, want in enumerate(parts):
                c0, c1 = int(bounds[i]), int(bounds[i + 1])
                nfr = int(want[0][-1])
                assert np.array_equal(want[0] + pos, res[0][c0:c1 + 1]), "GPU/oracle mismatch (fragments per cell)"
                for f_ in (2, 3, 4, 5):   # ref, start, frag_len, count
                    assert np.array_equal(want[f_], res[f_][pos:pos + nfr]), f"GPU/oracle mismatch (field {f_})"
                assert np.array_equal(want[1], res[1][c0:c1]), "GPU/oracle mismatch (barcodes)"
                pos += nfr
            cpu = {"value": round(k * per / tc / 1e6, 3), "unit": "M fragments/s", "cores": ncores, "kind": "port",
                   "sample": f"first {k} cells ({k * per} records), {tc:.1f} s, C++ restatement (oracle/), one slice of cells per host core; "
                             f"cell_ptr, barcode, ref, start, frag_len and count of every distinct fragment compared with the GPU's"}
        roof = roofline_of(kt, alg, steps, "atac" if (n_cells, per) == (10000, 20000) else None, 1e3 * elapsed / steps)
        return {
            "metric": "M fragments/s through atac dedup (fragment/barcode dedup path)", "value": round(total * steps / elapsed / 1e6, 3),
            "unit": "M fragments/s", "n_gpus": D.world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(1e3 * elapsed / steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"configs[4]: scATAC dedup from collated-RAD bytes, per GPU: {n_cells} cells x {per} records (20 % exact duplicates, "
                                   f"5 % multi-mapped, 5 % unmapped, 25 chromosomes); bytes resident in HBM, distinct fragments back on the host",
                       "records_per_gpu": n, "input_bytes_per_gpu": int(len(data)), "distinct": distinct, "stats": res[6]},
            "gen_seconds": round(t_gen, 1), "roofline": roof, "cpu_baseline": cpu}
    finally:
        res = None
        q.close()
        del d_bytes


def bench_atac(args, pkg, D):
    out = run_atac(args, pkg, D, args.steps, args.warmup)
    if D.rank == 0:
        print(json.dumps(out), flush=True)
    D.close()


if __name__ == "__main__":
    main()
