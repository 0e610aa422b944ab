import importlib, sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo/tests')
pkg = importlib.import_module("alevin-fry_amd"); import oracle as ora
from util import cfg_for
synth = pkg.synth
sizes = [20000, 6000, 1500, 700, 260, 250, 120, 99, 40, 3]
s = synth.synth(14, sizes, num_genes=300, usa=False, dup=0.5, zipf=0.5, cross=0.7, max_extra_na=6)
b, off = s.encode()
cfg = cfg_for(s, "cr-like-em")
q = pkg.Quantifier(cfg, s.tid_to_gid); got = q.quant_chunks(b, off); q.close()
want, iters = ora.quant(cfg, s.tid_to_gid, b, off, want_iters=True)
print("iters", iters)
for i in range(got.n_cells):
    g0,v0 = got.row(i); g1,v1 = want.row(i)
    if not np.array_equal(g0,g1) or not np.array_equal(v0,v1):
        d0 = dict(zip(g0.tolist(), v0.tolist())); d1 = dict(zip(g1.tolist(), v1.tolist()))
        keys = sorted(set(d0)|set(d1))
        bad = [(k, d0.get(k), d1.get(k)) for k in keys if d0.get(k)!=d1.get(k)]
        print("cell", i, "nrec", got.nrec[i], "ndiff", len(bad), bad[:6], "sum", v0.sum(), v1.sum())
