import sys, importlib, time, os, subprocess, numpy as np
sys.path.insert(0,'/root/repo')
pkg = importlib.import_module("alevin-fry_amd"); sn = importlib.import_module("alevin-fry_amd.synth_native"); rad = pkg.rad
ncell = int(sys.argv[1]) if len(sys.argv) > 1 else 11000
r = sn.generate(seed=2, n_cells=ncell, median_reads=30000.0, sigma=0.6, num_genes=36601, txp_per_gene=5)
d = "/tmp/e2e_in"; o = "/tmp/e2e_out"
names = [f"t{i}" for i in range(len(r.tid_to_gid))]
rows = [(names[i], f"g{int(r.tid_to_gid[i])}") for i in range(len(names))]
t = time.time()
tg = rad.write_quant_input_dir(d, r.data, ncell, names, rows, cblen=16, ulen=12)
print("wrote input dir: %.1f s, %.2f GB" % (time.time() - t, r.data.nbytes / 1e9), flush=True)
for res in ("cr-like",):
    t = time.time()
    p = subprocess.run(["/root/repo/alevin-fry_amd/csrc/afquant", "quant", "-i", d, "-m", tg, "-o", o, "-r", res, "-t", "16"], capture_output=True, text=True)
    dt = time.time() - t
    print(res, "rc", p.returncode, "wall %.2f s" % dt, "-> %.1f M reads/s end to end" % (r.n_reads / dt / 1e6))
    print(p.stderr[-1500:])
    print({f: os.path.getsize(os.path.join(o, "alevin", f)) for f in os.listdir(os.path.join(o, "alevin"))})

