import importlib, sys, time, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
pkg = importlib.import_module("alevin-fry_amd"); sn = importlib.import_module("alevin-fry_amd.synth_native")
import oracle as ora
ncell = int(sys.argv[1]) if len(sys.argv)>1 else 1000
for usa, res in ((False,'parsimony'),(True,'parsimony-em')):
    rad = sn.generate(seed=3, n_cells=ncell, median_reads=30000, usa=usa, umi_err=0.01)
    cfg = pkg.WorkerConfig.for_resolution(res, usa_mode=usa, num_genes=rad.num_genes, num_rows=rad.num_rows, profile=True)
    d = torch.from_numpy(rad.data).to('cuda:0')
    q = pkg.Quantifier(cfg, rad.tid_to_gid)
    for it in range(2):
        t=time.perf_counter(); q.submit_device(d.data_ptr(), d.numel(), rad.chunk_off); r=q.collect(); dt=time.perf_counter()-t
    print(res, 'usa',usa, 'cells',ncell,'reads',rad.n_reads,'ms',dt*1e3,'Mreads/s',rad.n_reads/dt/1e6, {k:round(v[0],2) for k,v in q.kernel_times().items()})
    idx = np.arange(0, ncell, max(1,ncell//40))
    t=time.perf_counter(); want = ora.quant(cfg, rad.tid_to_gid, rad.data, rad.chunk_off[idx], n_threads=64); dt=time.perf_counter()-t
    nbad=0; maxrel=0
    for j,ci in enumerate(idx):
        g0,v0=r.row(int(ci)); g1,v1=want.row(j)
        if not (np.array_equal(g0,g1) and np.array_equal(v0,v1)):
            nbad+=1
            if np.array_equal(g0,g1): maxrel=max(maxrel, float(np.max(np.abs(v0-v1)/v1)))
    print('  oracle', len(idx),'cells', rad.cell_nrec[idx].sum()/dt/1e6,'Mreads/s (64 thr)  mismatching cells',nbad,'maxrel',maxrel)
    q.close()
