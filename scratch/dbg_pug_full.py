import sys, importlib, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
pkg = importlib.import_module("alevin-fry_amd"); sn = importlib.import_module("alevin-fry_amd.synth_native")
import oracle as ora
rad = sn.generate(seed=7, n_cells=1500, median_reads=30000.0, sigma=0.6, num_genes=36601, txp_per_gene=5, usa=True, umi_err=0.01)
for ul in (12, 0):
    cfg = pkg.WorkerConfig.for_resolution("parsimony-em", usa_mode=True, num_genes=rad.num_genes, num_rows=rad.num_rows, umi_len=ul)
    q = pkg.Quantifier(cfg, rad.tid_to_gid)
    r = q.quant_chunks(rad.data, rad.chunk_off)
    nnz = np.diff(r.cell_ptr.astype(np.int64))
    print("umi_len", ul, "cells", r.n_cells, "empty rows", int((nnz == 0).sum()), "first empties", np.where(nnz == 0)[0][:10], "flags of empties", r.flags[nnz == 0][:10], "nrec of empties", r.nrec[nnz==0][:10])
    idx = np.array([400])
    alone = q.quant_chunks(rad.data, rad.chunk_off[idx])
    print("  cell 400 alone nnz", len(alone.gene), "in batch", nnz[400], "nrec", r.nrec[400])
    q.close()
