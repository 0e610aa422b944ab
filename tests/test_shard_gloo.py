"""CPU tests of the N>1 host logic: byte-balanced cell-range sharding + host-side gather, run as a
2-process gloo job.  Each rank quantifies its shard with the oracle standing in for the GPU (this is a
test of the sharding/gather logic, the device path has its own -m gpu tests)."""
import os
import subprocess
import sys

import numpy as np

from util import ROOT, pkg

shard = __import__("importlib").import_module("alevin-fry_amd.shard")


def test_shard_ranges_cover_and_balance():
    rng = np.random.default_rng(0)
    nb = np.sort(rng.integers(100, 100000, 1000))[::-1]
    for world in (1, 2, 3, 8):
        r = shard.shard_ranges(nb, world)
        assert r[0][0] == 0 and r[-1][1] == len(nb)
        assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
        loads = [nb[a:b].sum() for a, b in r]
        assert max(loads) <= 1.15 * nb.sum() / world + nb.max()
    assert shard.shard_ranges([5, 5], 4)[-1][1] == 2  # fewer cells than ranks


WORKER = r'''
import importlib, os, sys
import numpy as np
import torch.distributed as dist
sys.path.insert(0, os.environ["AFQ_ROOT"]); sys.path.insert(0, os.path.join(os.environ["AFQ_ROOT"], "oracle"))
pkg = importlib.import_module("alevin-fry_amd"); shard = importlib.import_module("alevin-fry_amd.shard")
import oracle as ora
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
s = pkg.synth.synth(21, [900, 700, 400, 300, 120, 80, 33, 5], num_genes=150, dup=0.4)
b, off = s.encode()
b = np.asarray(b)
nbytes = np.diff(np.concatenate((off, [len(b)]))).astype(np.int64)
cfg = pkg.WorkerConfig.for_resolution("cr-like", num_genes=s.num_genes, num_rows=s.num_rows)
c0, c1 = shard.shard_ranges(nbytes, world)[rank]
local = ora.quant(cfg, s.tid_to_gid, b, off[c0:c1], first_cell_index=c0) if c1 > c0 else None
full = shard.gather_results(local, dist, dst=0)
if rank == 0:
    want = ora.quant(cfg, s.tid_to_gid, b, off)
    assert full.n_cells == want.n_cells
    for f in ("cell_ptr", "gene", "val", "bc", "nrec", "flags"):
        assert np.array_equal(getattr(full, f), getattr(want, f)), f
    print("GLOO_SHARD_OK", c0, c1)
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_gloo_shard_and_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, AFQ_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29591", WORLD_SIZE="2")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "GLOO_SHARD_OK" in outs[0]
