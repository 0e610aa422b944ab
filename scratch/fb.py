import importlib, sys, numpy as np, torch
sys.path.insert(0,'/root/repo')
pkg = importlib.import_module("alevin-fry_amd"); sn = importlib.import_module("alevin-fry_amd.synth_native")
rad = sn.generate(seed=3, n_cells=2000, median_reads=30000)
cfg = pkg.WorkerConfig.for_resolution('cr-like', num_genes=rad.num_genes, num_rows=rad.num_rows, profile=True)
d = torch.from_numpy(rad.data).to('cuda:0')
q = pkg.Quantifier(cfg, rad.tid_to_gid)
q.submit_device(d.data_ptr(), d.numel(), rad.chunk_off); r=q.collect()
print(q.batch_stats(), {k:round(v[0],2) for k,v in q.kernel_times().items()})
