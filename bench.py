#!/usr/bin/env python
"""bench.py — M reads/s through the `quant` hot path (cr-like) on N MI355X GPUs.

A "step" is one pass of the whole hot path (afq_submit_device + afq_collect: decode ->
bucket -> resolve -> extract -> CSR on the host) over one batch of synthetic collated RAD
that is already resident in HBM.  Workload at every N: BASELINE.json configs[1] — a
PBMC-10k-like 10x-v3 collated RAD (11 000 cells, log-normal reads/cell with median 3e4,
36 601 genes, cr-like), one such shard PER RANK (weak scaling: cells are independent, so
ranks share nothing on the data path; the only collectives are the timing barrier and a
max/sum of scalars).

One JSON line on rank 0; see DESIGN.md §Measurement for how roofline/cpu_baseline are built.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402  (imported before libafquant.so: one HIP runtime per process)
import torch.distributed as dist  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cells", type=int, default=11000)
    ap.add_argument("--median-reads", type=float, default=30000.0)
    ap.add_argument("--sigma", type=float, default=0.6)
    ap.add_argument("--genes", type=int, default=36601)
    ap.add_argument("--usa", action="store_true")
    ap.add_argument("--resolution", default="cr-like",
                    help="default cr-like = configs[1] (the headline line); parsimony-em with --usa = configs[2]")
    ap.add_argument("--umi-err", type=float, default=0.01)
    ap.add_argument("--bootstraps", type=int, default=0, help="-b: bootstrap replicates per cell (-em resolutions; extra measurement, not the default)")
    ap.add_argument("--atac", action="store_true", help="configs[4]: scATAC fragment dedup (extra line, not the default)")
    ap.add_argument("--frags-per-cell", type=int, default=20000)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU baseline budget (rank 0, N=1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL) by default; gloo + --share-gpu exercises the N>1 logic on a 1-GPU box")
    ap.add_argument("--share-gpu", action="store_true", help="testing only: every rank uses cuda:0")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.share_gpu:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.dist_backend, rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rdev = dev if args.dist_backend == "nccl" else torch.device("cpu")  # where the scalar reductions live

    pkg = importlib.import_module("alevin-fry_amd")
    sn = importlib.import_module("alevin-fry_amd.synth_native")
    if args.atac:
        return bench_atac(args, pkg, rank, world, local_rank, dev)

    # ---- synthetic input (config 2), one shard per rank, then resident in HBM -------------
    t0 = time.time()
    rad = sn.generate(seed=2 + rank, n_cells=args.cells, median_reads=args.median_reads, sigma=args.sigma,
                      num_genes=args.genes, txp_per_gene=5, usa=args.usa, umi_err=args.umi_err)
    t_gen = time.time() - t0
    d_bytes = torch.from_numpy(rad.data).to(dev)
    cfg = pkg.WorkerConfig.for_resolution(args.resolution, usa_mode=rad.usa, num_genes=rad.num_genes,
                                          num_rows=rad.num_rows, profile=True, umi_len=12, num_bootstraps=args.bootstraps, summary_stat=True)  # umi_len: what the RAD header's `ulen` tag says
    q = pkg.Quantifier(cfg, rad.tid_to_gid, device=local_rank)

    res = None

    def step():
        # the host has consumed the previous batch's rows: hand its pinned buffers back to the library's
        # pool before the next batch needs them (otherwise the pool pins a second 0.3 GB set, ~25 ms once)
        nonlocal res
        res = None
        q.submit_device(d_bytes.data_ptr(), d_bytes.numel(), rad.chunk_off)
        res = q.collect()
        return res

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    ktimes = {}
    for _ in range(args.steps):
        step()
        for k, (ms, n) in q.kernel_times().items():  # HIP events on the library's own stream
            a = ktimes.setdefault(k, [0.0, 0])
            a[0] += ms
            a[1] += n
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        r = torch.tensor([float(rad.n_reads), float(args.cells)], dtype=torch.float64, device=rdev)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        total_reads, total_cells = float(r[0].item()), float(r[1].item())
    else:
        total_reads, total_cells = float(rad.n_reads), float(args.cells)

    if rank != 0:
        q.close()
        if world > 1:
            dist.destroy_process_group()
        return

    st = q.batch_stats()
    ms_per_step = 1e3 * elapsed / args.steps
    value = total_reads * args.steps / elapsed / 1e6

    # ---- size-independent sanity on the full-size result --------------------------------
    nnz = int(res.cell_ptr[-1])
    assert res.n_cells == args.cells and (np.diff(res.cell_ptr.astype(np.int64)) >= 0).all()
    assert (res.val > 0).all() and float(res.val.sum()) <= rad.n_reads * (1 + 1e-6)
    assert np.array_equal(res.nrec, rad.cell_nrec)

    # ---- roofline of the dominant kernel -------------------------------------------------
    # algorithmic bytes of one pass: every record read once (12+4*na), the chunk
    # headers, and 8 B per emitted non-zero.  The dominant kernel is charged with all of them:
    # it is the share of the path's wall time that decides the path's achieved bandwidth.
    alg_bytes = float(st["input_bytes"]) + 8.0 * nnz
    dom = max(ktimes.items(), key=lambda kv: kv[1][0]) if ktimes else None
    roofline = None
    if dom:
        name, (ms_tot, launches) = dom
        avg_ms = ms_tot / max(1, launches)
        launches_per_step = launches / args.steps
        achieved = alg_bytes / launches_per_step / (avg_ms * 1e-3) / 1e9
        # HBM bytes per launch of that kernel from the committed PMC passes (profiles/*_traffic.json: FETCH_SIZE doubled
        # per the gfx950 note + WRITE_SIZE); only meaningful for the default workload the profile was taken on
        traffic = None
        default_wl = (args.cells, args.median_reads, args.sigma, args.genes, args.usa, args.resolution) == \
            (11000, 30000.0, 0.6, 36601, False, "cr-like")
        tfiles = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic.json")) \
            if os.path.isdir(os.path.join(ROOT, "profiles")) else []
        if default_wl and tfiles:
            kern = json.load(open(os.path.join(ROOT, "profiles", tfiles[-1])))["kernels"]
            alias = {"k_decode_par": ("k_decode_recs", "k_decode_keys", "k_decode_par")}
            for cand in alias.get(name, (name,)):
                if cand in kern:
                    traffic = kern[cand]["bytes_per_launch_fetch_doubled"]
                    break
        roofline = {"bound": "hbm", "kernel": name, "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s",
                    "frac": round(achieved / 8000.0, 5), "traffic": traffic,
                    "avg_launch_ms": round(avg_ms, 4), "launches_per_step": launches_per_step,
                    "alg_bytes_per_step": alg_bytes,
                    "all_kernels_ms_per_step": {k: round(v[0] / args.steps, 4) for k, v in ktimes.items()}}

    # ---- CPU baseline: the oracle (a port, not the Rust binary) on a bounded sample ---------
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle as ora

        ora.lib()
        # sample = every k-th cell so the size mix matches the workload; grow until the budget is used
        order = np.arange(args.cells)
        ncores = os.cpu_count() or 1
        k = max(1, args.cells // max(64, 4 * ncores))
        done_reads, t_cpu, ncell = 0, 0.0, 0
        start = 0
        while t_cpu < args.cpu_seconds and start < k:
            idx = order[start::k]
            start += 1
            offs = rad.chunk_off[idx]
            tb = time.perf_counter()
            want = ora.quant(cfg, rad.tid_to_gid, rad.data, offs, n_threads=ncores)
            t_cpu += time.perf_counter() - tb
            done_reads += int(rad.cell_nrec[idx].sum())
            ncell += len(idx)
            # the sample doubles as a full-size parity check: GPU rows == oracle rows, bit for bit
            for j, ci in enumerate(idx):
                g0, v0 = res.row(int(ci))
                g1, v1 = want.row(j)
                assert np.array_equal(g0, g1) and np.array_equal(v0.view(np.uint32), v1.view(np.uint32)), \
                    f"GPU/oracle mismatch on cell {ci}"
        cpu = {"value": round(done_reads / t_cpu / 1e6, 4), "unit": "M reads/s", "cores": ncores, "kind": "port",
               "sample": f"{ncell} of {args.cells} cells (every {k}-th, in rounds), {done_reads} reads, {t_cpu:.1f} s, "
                         f"C++ restatement (oracle/) with one worker thread per host core over whole cells, input "
                         f"already parsed from RAM; rows compared bit-exact with the GPU's"}

    out = {
        "metric": "M reads/s through quant (PUG dedup+eq-class) at 1/2/4/8 GPUs; cells/s",
        "value": round(value, 3),
        "unit": "M reads/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {"workload": ("configs[2]" if (args.usa and args.resolution == "parsimony-em") else "configs[1]") +
                               f": PBMC-10k-like 10x-v3 collated RAD, {args.resolution}, per GPU: "
                               f"{args.cells} cells, log-normal reads/cell median {args.median_reads:g} sigma {args.sigma:g}, "
                               f"{args.genes} genes" + (", USA" if args.usa else ""),
                   "reads_per_gpu": rad.n_reads, "input_bytes_per_gpu": st["input_bytes"], "resolution": args.resolution, **({"bootstraps": args.bootstraps} if args.bootstraps else {}),
                   "sharding": f"{world} x independent cell shards, no data-path collective"},
        "cells_per_s": round(total_cells * args.steps / elapsed, 1),
        "nnz": nnz,
        "keys": st["n_keys"],
        "overflow_buckets": st["n_overflow_buckets"],
        "gen_seconds": round(t_gen, 1),
        "roofline": roofline,
        "cpu_baseline": cpu,
    }
    print(json.dumps(out))
    q.close()
    if world > 1:
        dist.destroy_process_group()


def bench_atac(args, pkg, rank, world, local_rank, dev):
    """BASELINE configs[4]: per-cell ATAC fragment de-duplication (afq_atac_dedup).  The boundary takes and returns host
    arrays (the reference's deduplicate reads them from the sorted RAD), so a step includes both PCIe crossings; the kernel's
    own time comes from the library's HIP-event timers."""
    import ctypes as C

    n_cells = args.cells if args.cells != 11000 else 10000
    per = args.frags_per_cell
    rng = np.random.default_rng(5 + rank)
    n = n_cells * per
    ref = rng.integers(0, 25, n, dtype=np.uint32)
    start = rng.integers(0, 150_000_000, n, dtype=np.uint32)
    flen = np.clip(rng.lognormal(5.2, 0.6, n), 30, 2500).astype(np.uint16)
    idx = np.arange(n)
    src = np.where((rng.random(n) < 0.2) & (idx % per != 0), idx - 1, idx)  # 20 % exact duplicates
    ref, start, flen = ref[src], start[src], flen[src]
    cell_ptr = np.arange(n_cells + 1, dtype=np.uint64) * per
    cfg = pkg.WorkerConfig.for_resolution("cr-like", num_genes=1, num_rows=1, profile=True)
    q = pkg.Quantifier(cfg, np.zeros(1, np.uint32), device=local_rank)
    outs = [C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint16)(), C.POINTER(C.c_uint16)()]

    def step():
        rc = q.lib.afq_atac_dedup(q._h, ref.ctypes.data_as(C.POINTER(C.c_uint32)), start.ctypes.data_as(C.POINTER(C.c_uint32)),
                                  flen.ctypes.data_as(C.POINTER(C.c_uint16)), cell_ptr.ctypes.data_as(C.POINTER(C.c_uint64)), n_cells,
                                  *[C.byref(o) for o in outs])
        assert rc == 0, q.lib.afq_last_error(q._h)
        distinct = int(outs[0][n_cells])
        for o in outs:
            q.lib.afq_free(o)
        return distinct

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    kms, kl, distinct = 0.0, 0, 0
    for _ in range(args.steps):
        distinct = step()
        ms, nl = q.kernel_times().get("k_atac_dedup", (0.0, 0))
        kms += ms
        kl += nl
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    total = float(n)
    if world > 1:
        dist.barrier()
        rdev = dev if args.dist_backend == "nccl" else torch.device("cpu")
        t = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        r = torch.tensor([total], dtype=torch.float64, device=rdev)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        total = float(r.item())
    if rank == 0:
        alg = 10.0 * n + 12.0 * distinct   # ref u32 + start u32 + len u16 in; (ref, start, len, count) per distinct fragment out
        avg_ms = kms / max(1, kl)
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle as ora

            k = max(1, min(n_cells, int(20e6 // per)))
            tb = time.perf_counter()
            ora.atac_dedup(ref[: k * per], start[: k * per], flen[: k * per], cell_ptr[: k + 1])
            tc = time.perf_counter() - tb
            cpu = {"value": round(k * per / tc / 1e6, 3), "unit": "M fragments/s", "cores": 1, "kind": "port",
                   "sample": f"first {k} cells ({k * per} fragments), {tc:.1f} s, single-thread C++ restatement (oracle/)"}
        print(json.dumps({
            "metric": "M fragments/s through atac dedup (fragment/barcode dedup path)", "value": round(total * args.steps / elapsed / 1e6, 3),
            "unit": "M fragments/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"configs[4]: scATAC dedup, per GPU: {n_cells} cells x {per} fragments, 25 chromosomes, 20 % exact duplicates; host arrays in, host arrays out (both PCIe crossings inside the step)",
                       "fragments_per_gpu": n, "distinct": distinct},
            "roofline": {"bound": "hbm", "kernel": "k_atac_dedup64", "achieved": round(alg / (avg_ms * 1e-3) / 1e9, 2) if avg_ms else None,
                         "peak": 8000.0, "unit": "GB/s", "frac": round(alg / (avg_ms * 1e-3) / 1e9 / 8000.0, 5) if avg_ms else None,
                         "traffic": None, "avg_launch_ms": round(avg_ms, 4), "alg_bytes_per_step": alg},
            "cpu_baseline": cpu}))
    q.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
