"""BASELINE configs[4]-shaped ATAC dedup: n_cells x frags/cell, 20 % exact duplicates; times afq_atac_dedup end to end
(host arrays in, host arrays out) and the oracle on a sample."""
import sys, importlib, time, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
pkg = importlib.import_module("alevin-fry_amd")
import oracle as ora
n_cells = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
per = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
rng = np.random.default_rng(5)
n = n_cells * per
ref = rng.integers(0, 25, n, dtype=np.uint32)
start = rng.integers(0, 150_000_000, n, dtype=np.uint32)
flen = np.clip(rng.lognormal(5.2, 0.6, n), 30, 2500).astype(np.uint16)
dup = rng.random(n) < 0.2            # copy the previous fragment of the same cell
idx = np.arange(n); src = np.where(dup & (idx % per != 0), idx - 1, idx)
ref, start, flen = ref[src], start[src], flen[src]
cell_ptr = (np.arange(n_cells + 1, dtype=np.uint64) * per)
cfg = pkg.WorkerConfig.for_resolution("cr-like", num_genes=1, num_rows=1)
q = pkg.Quantifier(cfg, np.zeros(1, np.uint32))
for it in range(3):
    t = time.perf_counter(); out = q.atac_dedup(ref, start, flen, cell_ptr); dt = time.perf_counter() - t
    print(f"run {it}: {n/1e6:.0f} M fragments in {dt*1e3:.1f} ms -> {n/dt/1e6:.0f} M fragments/s (host in, host out); {int(out[0][-1])/1e6:.1f} M distinct")
k = 200
t = time.perf_counter(); want = ora.atac_dedup(ref[:k * per], start[:k * per], flen[:k * per], cell_ptr[:k + 1]); dt = time.perf_counter() - t
print(f"oracle: {k*per/dt/1e6:.1f} M fragments/s (1 thread)")
e = int(out[0][k])
assert np.array_equal(out[0][:k + 1], want[0]) and all(np.array_equal(a[:e], b) for a, b in zip(out[1:], want[1:])), "mismatch"
print("first", k, "cells bit-exact vs oracle")
q.close()
