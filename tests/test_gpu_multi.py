"""GPU tests of what round 2 added around the hot path: the device generator (== its host twin, byte for byte), the
range-by-range input upload of afq_submit, the multi-device front-end (`afquant quant --devices`, two contexts sharing one
device), `bench.py --gpus 2` spawning its own ranks, multi-barcode (10x Flex) input, and the reference-binary hook."""
import importlib
import json
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

from util import ROOT, assert_same_result, cfg_for, pkg

pytestmark = pytest.mark.gpu
rad = pkg.rad
synth = pkg.synth
sn = importlib.import_module("alevin-fry_amd.synth_native")
shard = importlib.import_module("alevin-fry_amd.shard")
CLI = os.path.join(ROOT, "alevin-fry_amd", "csrc", "afquant")


@pytest.mark.parametrize("kw", [dict(n_cells=300, median_reads=900.0, sigma=1.0, num_genes=500, ref_count=1733),
                                dict(n_cells=40, median_reads=30000.0, ref_count=199138),
                                dict(n_cells=200, median_reads=2000.0, num_genes=300, txp_per_gene=3, usa=True, umi_err=0.05),
                                dict(n_cells=500, median_reads=40.0, sigma=0.3, num_genes=100, pow_skew=16.0, zipf=0.0, umi_len=10),
                                dict(n_cells=120, median_reads=2500.0, num_genes=400, txp_per_gene=4, usa=True, tail=0.65, family=8),
                                dict(n_cells=60, median_reads=4000.0, num_genes=300, ref_count=1500, tail=0.8, tail_max=64, family=16)])
def test_device_generator_writes_the_host_generators_bytes(kw):
    """csrc/afq_synth.hip: the gfx950 kernels and the host loop run the same integer record model (Philox4x32-10 words
    against 32-bit thresholds), so the bytes must agree exactly - for the whole set and for a range of it."""
    h = sn.generate(seed=7, **kw)
    d = sn.generate_device(device=0, seed=7, **kw)
    try:
        assert d.n_bytes == h.n_bytes and np.array_equal(d.chunk_off, h.chunk_off) and np.array_equal(d.cell_nrec, h.cell_nrec)
        assert np.array_equal(d.to_host(), h.data)
    finally:
        d.free()
    n = kw["n_cells"]
    c0, c1 = n // 3, (2 * n) // 3
    dr = sn.generate_device(device=0, seed=7, cell_range=(c0, c1), **kw)
    try:
        want, _ = h.read_cells(np.arange(c0, c1))
        assert np.array_equal(dr.to_host(), want)   # a rank's shard does not depend on what the others make
    finally:
        dr.free()
    # structure: headers tile the buffer, barcodes are distinct, refs ascending and in range
    w = h.data.view(np.uint32)
    bcs = set()
    for c in range(n):
        o = int(h.chunk_off[c]) // 4
        assert w[o + 1] == h.cell_nrec[c] and w[o] == (h.chunk_off[c + 1] - h.chunk_off[c] if c + 1 < n else h.n_bytes - h.chunk_off[c])
        bcs.add(int(w[o + 3]))
    assert len(bcs) == n
    na = int(w[int(h.chunk_off[0]) // 4 + 2])
    refs = w[int(h.chunk_off[0]) // 4 + 5: int(h.chunk_off[0]) // 4 + 5 + na] & 0x7FFFFFFF
    assert 1 <= na <= (64 if kw.get("tail") else 3) and (np.diff(refs.astype(np.int64)) > 0).all() and refs.max() < len(h.tid_to_gid)


@pytest.mark.parametrize("tail", [0.0, 0.7])
def test_generated_workload_quantifies_like_the_oracle(oracle, monkeypatch, tail):
    """The generator's output through the device path and through the oracle (USA, parsimony-em): the bench's input is
    an ordinary collated RAD as far as both are concerned - with the label-length tail too (labels of up to dozens of refs
    on gene families: the long-record paths of the decoders, hashed label keys, molecules of more than four genes).  On the
    tailed input the parsimony resolutions also run with each way the lone-vertex kernel has of resolving a label of more than
    four refs (AFQ_TEST_P2_LONE_COOP: by its lane in scratch memory, by the wave, 5..8 refs by the lane in registers; the default picks
    by range)."""
    d = sn.generate_device(device=0, seed=3, n_cells=60, median_reads=3000.0, num_genes=400, txp_per_gene=4, usa=True, umi_err=0.03,
                           tail=tail, family=8)
    try:
        host = d.to_host()
        for res in ("cr-like", "parsimony-em", "parsimony", "cr-like-em"):
            cfg = pkg.WorkerConfig.for_resolution(res, usa_mode=True, num_genes=d.num_genes, num_rows=d.num_rows, umi_len=12)
            want = oracle.quant(cfg, d.tid_to_gid, host, d.chunk_off, n_threads=4)
            for lone in ((None, "0", "1", "2") if tail and res.startswith("parsimony") else (None,)):
                with monkeypatch.context() as mp:
                    if lone is not None:
                        mp.setenv("AFQ_TEST_P2_LONE_COOP", lone)
                    q = pkg.Quantifier(cfg, d.tid_to_gid, device=0)
                    try:
                        q.submit_device(d.d_ptr, d.n_bytes, d.chunk_off)
                        got = q.collect()
                    finally:
                        q.close()
                assert_same_result(got, want, what=f"{res} AFQ_TEST_P2_LONE_COOP={lone}")
    finally:
        d.free()


@pytest.mark.parametrize("pinned", [False, True])
@pytest.mark.parametrize("res", ["cr-like", "parsimony"])
def test_piped_upload_matches_resident_input(monkeypatch, pinned, res):
    """afq_submit brings the input over range by range while earlier ranges run (csrc/afq_api.cpp); forced here onto a
    small batch by capping the per-range memory.  Pageable source = staging thread, pinned source = straight DMA.
    Chunk offsets are deliberately not dword-aligned in the caller's buffer (a RAD prelude has any length)."""
    import torch

    s = synth.synth(77, [6000, 5000, 3000, 2500, 900, 700, 400, 300, 120, 80, 33, 5] * 3, num_genes=200, dup=0.4, umi_err=0.02)
    b, off = s.encode()
    b = np.concatenate((np.zeros(3, np.uint8), np.asarray(b)))   # shift: offsets become 3 mod 4
    off = off + 3
    cfg = cfg_for(s, res)
    q = pkg.Quantifier(cfg, s.tid_to_gid, device=0)
    try:
        monkeypatch.setenv("AFQ_TEST_NO_H2D_PIPELINE", "1")
        want = q.quant_chunks(b, off)
        monkeypatch.delenv("AFQ_TEST_NO_H2D_PIPELINE")
        monkeypatch.setenv("AFQ_TEST_RANGE_BYTES", str(400_000))
        if pinned:
            t = torch.empty(len(b), dtype=torch.uint8, pin_memory=True)
            t.numpy()[:] = b
            q.submit_ptr(t.data_ptr(), len(b), off)
            got = q.collect()
        else:
            got = q.quant_chunks(b, off)
        assert_same_result(got, want, what=f"piped upload, pinned={pinned}")
        got2 = q.quant_chunks(b, off)   # and again on the same context (events / staging pieces are reused)
        assert_same_result(got2, want)
    finally:
        q.close()


def test_two_contexts_on_one_device_equal_one_context(oracle):
    """§8e: cells are independent, so two contexts over byte-balanced contiguous ranges (here both on device 0, running
    concurrently from two host threads) must reproduce the single-context rows bit for bit."""
    import threading

    s = synth.synth(91, [5000, 4000, 2600, 1200, 900, 700, 400, 300, 120, 80, 33, 5, 3000, 60], num_genes=250, usa=True, dup=0.5, umi_err=0.02)
    b, off = s.encode()
    b = np.asarray(b)
    nbytes = np.diff(np.concatenate((off, [len(b)]))).astype(np.int64)
    for res in ("cr-like", "parsimony-em"):
        cfg = cfg_for(s, res)
        q = pkg.Quantifier(cfg, s.tid_to_gid, device=0)
        whole = q.quant_chunks(b, off)
        q.close()
        ranges = shard.shard_ranges(nbytes, 2)
        parts = [None, None]
        errs = []

        def work(r):
            try:
                c0, c1 = ranges[r]
                qq = pkg.Quantifier(cfg, s.tid_to_gid, device=0)
                try:
                    parts[r] = qq.quant_chunks(b, off[c0:c1], first_cell_index=c0)
                finally:
                    qq.close()
            except Exception as e:  # noqa: BLE001
                errs.append(e)

        th = [threading.Thread(target=work, args=(r,)) for r in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        assert parts[1].first_cell_index == ranges[1][0]
        assert_same_result(shard.concat_results(parts), whole, what=res)


def _make_dir(tmp, s):
    b, off = s.encode()
    names = [f"T{t}" for t in range(len(s.tid_to_gid))]
    rows = [(names[t], f"G{g >> 1}", "S" if g % 2 == 0 else "U") if s.usa else (names[t], f"G{g}") for t, g in enumerate(s.tid_to_gid.tolist())]
    return rad.write_quant_input_dir(str(tmp), np.asarray(b).tobytes(), len(off), names, rows, cblen=16, ulen=s.umi_len), b, off


@pytest.mark.parametrize("res,extra", [("cr-like", []), ("parsimony-em", ["-d"]), ("cr-like-em", ["-b", "4", "--summary-stat"])])
def test_cli_devices_output_is_independent_of_the_device_count(tmp_path, res, extra):
    """`afquant quant --devices 0,0,0` (three contexts + host threads popping batches of cells off one queue, rows gathered in
    cell order, the -d dictionary filled in cell order) writes the same files as `--device 0`."""
    s = synth.synth(52, [4000, 2500, 1500, 600, 260, 120, 60, 7, 900, 30], num_genes=150, txp_per_gene=3, usa=True, dup=0.5, cross=0.3, umi_err=0.02)
    tg, _, _ = _make_dir(tmp_path / "in", s)
    outs = []
    for name, dev in (("one", ["--device", "0"]), ("three", ["--devices", "0,0,0"])):
        o = tmp_path / name
        env = dict(os.environ, AFQ_TEST_QUEUE_BATCH_BYTES="30000") if name == "three" else None   # (a queue of several batches even on this small input)
        r = subprocess.run([CLI, "quant", "-i", str(tmp_path / "in"), "-m", tg, "-o", str(o), "-r", res, "-t", "4"] + dev + extra, capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr
        outs.append(o)
    files = ["alevin/quants_mat.mtx", "alevin/quants_mat_rows.txt", "alevin/quants_mat_cols.txt", "featureDump.txt"]
    if "-d" in extra:
        files += ["alevin/geqc_counts.mtx"]
    if "-b" in extra:
        files += ["alevin/bootstraps_mean.mtx", "alevin/bootstraps_var.mtx"]
    for f in files:
        assert (outs[0] / f).read_bytes() == (outs[1] / f).read_bytes(), f
    if "-d" in extra:
        import gzip

        assert gzip.open(outs[0] / "alevin/gene_eqclass.txt.gz").read() == gzip.open(outs[1] / "alevin/gene_eqclass.txt.gz").read()
    a, b = (json.load(open(o / "quant.json")) for o in outs)
    for k in ("num_quantified_cells", "total_records", "alt_resolved_cell_numbers", "empty_resolved_cell_numbers", "tiny_cell_resolved_cell_numbers"):
        assert a[k] == b[k], k
    dv = json.load(open(outs[1] / "afquant_devices.json"))
    assert dv["batches"] >= 3 and sum(d["cells"] for d in dv["devices"]) == 10 and all(d["batches"] >= 1 for d in dv["devices"])
    r = subprocess.run([CLI, "quant", "-i", str(tmp_path / "in"), "-m", tg, "-o", str(tmp_path / "bad"), "-r", res, "--devices", "0,99"], capture_output=True, text=True,
                       env=dict(os.environ, AFQ_TEST_QUEUE_BATCH_BYTES="30000"))
    assert r.returncode != 0 and "device 99" in r.stderr


def test_cli_device_queue_balances_a_largest_first_file(tmp_path):
    """Dynamic distribution over devices (the reference's workers pop chunks off a queue, quant.rs:1553-1575): a collated file
    is ordered largest cells first and a parsimony cell's cost grows faster than its bytes, so contiguous byte-balanced cuts
    would give one device all the expensive cells.  Three contexts on cuda:0 pop fixed-byte batches instead; their busy times
    must come out within 1.35x of each other (1.00-1.15x on most boxes, 1.19x seen once; one thread's first allocations can add 0.04 s of a
    0.22 s run - contiguous cuts are off by more than 2x) and the files must be those of one device."""
    d = sn.generate(seed=9, n_cells=600, median_reads=12000.0, sigma=0.8, num_genes=2000, txp_per_gene=4, usa=True, umi_err=0.02)
    names = [f"T{t}" for t in range(len(d.tid_to_gid))]
    rows = [(names[t], f"G{g >> 1}", "S" if g % 2 == 0 else "U") for t, g in enumerate(d.tid_to_gid.tolist())]
    tg = rad.write_quant_input_dir(str(tmp_path / "in"), d.data.tobytes(), len(d.chunk_off), names, rows, cblen=16, ulen=12)
    outs = []
    for name, dev, env in (("one", ["--device", "0"], None), ("three", ["--devices", "0,0,0"], dict(os.environ, AFQ_TEST_QUEUE_BATCH_BYTES=str(d.n_bytes // 90)))):
        o = tmp_path / name
        r = subprocess.run([CLI, "quant", "-i", str(tmp_path / "in"), "-m", tg, "-o", str(o), "-r", "parsimony-em", "-t", "8"] + dev, capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr
        outs.append(o)
    for f in ("alevin/quants_mat.mtx", "alevin/quants_mat_rows.txt", "alevin/quants_mat_cols.txt", "featureDump.txt"):
        assert (outs[0] / f).read_bytes() == (outs[1] / f).read_bytes(), f
    dv = json.load(open(outs[1] / "afquant_devices.json"))
    busy = [x["busy_s"] for x in dv["devices"]]
    assert dv["batches"] >= 60 and len(busy) == 3 and sum(x["cells"] for x in dv["devices"]) == 600
    assert max(busy) <= 1.35 * min(busy), dv


def test_quant_subset_sizes_the_matrix_by_the_subset(tmp_path):
    """--quant-subset: rows for the cells found, but the MTX row dimension and num_quantified_cells = the subset's size
    (quant.rs:1529, 1836, 1918), also when a listed barcode is not in the file."""
    s = synth.synth(53, [500, 300, 120, 60], num_genes=50, dup=0.3)
    tg, _, _ = _make_dir(tmp_path / "in", s)
    keep = [rad.int_to_seq(int(s.cell_bc[1]), 16), rad.int_to_seq(int(s.cell_bc[3]), 16), "ACGTACGTACGTACGT"]
    (tmp_path / "subset.txt").write_text("\n".join(keep) + "\n")
    o = tmp_path / "out"
    r = subprocess.run([CLI, "quant", "-i", str(tmp_path / "in"), "-m", tg, "-o", str(o), "-r", "cr-like", "--quant-subset", str(tmp_path / "subset.txt")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (o / "alevin/quants_mat_rows.txt").read_text().split() == keep[:2]
    size = [l for l in (o / "alevin/quants_mat.mtx").read_text().splitlines() if not l.startswith("%")][0].split()
    assert int(size[0]) == 3 and json.load(open(o / "quant.json"))["num_quantified_cells"] == 3


@pytest.mark.parametrize("w0", [4, 2, 1])
def test_multi_barcode_flex_quant(tmp_path, oracle, w0):
    """(w0 = bytes of b0: a RAD writer picks the narrowest integer per barcode length, so an 8-nt sample barcode next to a
    16-nt cell barcode is u16 + u32 - a 6-byte pair the library takes as a split barcode field, afq_config.bc_split.)
    10x Flex (KnownRecordType::RnaShortMultiBC): records carry (b0 = sample index after collation, b1 = cell barcode,
    u); rows are labelled sample_cell and featureDump has the sample_name column (src/quant.rs:1217-1262, 1354-1373).
    Structure as the reference's tests build it (tests/multi_barcode_integration.rs:721-1050): samples x cells x 8 reads,
    shared cell barcodes across samples; plus the counts against the oracle reading the same bytes with an 8-byte key."""
    n_samples, cells_per_sample, G = 3, 4, 10
    names = [f"gene_{i}" for i in range(G)]
    rng = np.random.default_rng(5)
    cells = []
    for si in range(n_samples):          # collation leaves the samples contiguous, b0 = sample ordinal
        for ci in range(cells_per_sample):
            cell_bc = (ci * 2654435761) & 0xFFFFFFFF   # the same cell barcodes in every sample
            nrec = 8 if ci else 300
            reads = []
            for r in range(nrec):
                umi = int(rng.integers(0, 1 << 24)) if ci == 0 else ((si * 100000 + ci * 100 + r) * 2654435761) & 0xFFFFFF
                refs = [r % G] if ci else sorted({int(rng.integers(0, G)) for _ in range(int(rng.integers(1, 3)))})
                reads.append((umi, refs))
            cells.append(((cell_bc << 32) | si, reads))
    b, off = rad.encode_cells(cells, bc_bytes=8, umi_bytes=4)   # what the oracle reads: the pair as one 8-byte key
    bf, _ = rad.encode_cells([(((bc >> 32) << (8 * w0)) | (bc & 0xFFFFFFFF), reads) for bc, reads in cells], bc_bytes=w0 + 4, umi_bytes=4)
    d = tmp_path / "in"
    pre = rad.rad_prelude_multi_bc(names, len(cells), 8, 16, 12, b0_bytes=w0)
    tg = rad.write_quant_input_dir(str(d), bytes(bf), len(cells), names, [(n, n) for n in names], prelude=pre)
    groups = [(0xAA + i, None if i == 1 else f"sample_{'abc'[i]}", i * cells_per_sample, cells_per_sample, 0) for i in range(n_samples)]
    (d / "collation_manifest.bin").write_bytes(rad.collation_manifest(groups))
    for res in ("trivial", "cr-like", "parsimony"):
        o = tmp_path / f"out_{res}"
        r = subprocess.run([CLI, "quant", "-i", str(d), "-m", tg, "-o", str(o), "-r", res, "--small-thresh", "0"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        rows = (o / "alevin/quants_mat_rows.txt").read_text().split()
        sname = ["sample_a", f"{0xAB:x}", "sample_c"]   # an unnamed sample goes by its key in hex
        assert rows == [f"{sname[i // cells_per_sample]}_{rad.int_to_seq(((i % cells_per_sample) * 2654435761) & 0xFFFFFFFF, 16)}" for i in range(len(cells))]
        assert len(set(rows)) == len(rows)
        feat = [l.split("\t") for l in (o / "featureDump.txt").read_text().splitlines()]
        assert {len(f) for f in feat} == {10} and feat[0][1] == "sample_name" and len(feat) == len(cells) + 1
        assert [f[1] for f in feat[1:]] == [sname[i // cells_per_sample] for i in range(len(cells))]
        cfg = pkg.WorkerConfig.for_resolution(res, num_genes=G, num_rows=G, bc_bytes=8, umi_bytes=4, small_thresh=0)
        want = oracle.quant(cfg, np.arange(G, dtype=np.uint32), b, off)
        body = [l for l in (o / "alevin/quants_mat.mtx").read_text().splitlines() if not l.startswith("%")]
        got = {(int(r) - 1, int(c) - 1): float(v) for r, c, v in (l.split() for l in body[1:])}
        exp = {(i, int(g)): float(v) for i in range(want.n_cells) for g, v in zip(*want.row(i))}
        assert got == exp, res
        if res != "parsimony":   # 8 distinct UMIs on genes 0..7: one molecule each under any strategy (tests/multi_barcode_integration.rs:163-199)
            assert all(got.get((i, g), 0.0) == 1.0 for i in range(len(cells)) if i % cells_per_sample for g in range(8))


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher environment starts two ranks itself, and the line says n_gpus 2 (here
    both on cuda:0 over gloo, tiny workloads; the driver's run uses one GPU per rank and RCCL)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--dist-backend", "gloo", "--steps", "2", "--warmup", "1",
           "--cells", "300", "--median-reads", "2000", "--c3-cells", "2000", "--c3-mean-reads", "300", "--also", "e2e,configs3"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["config"]["workload"].startswith("configs[1]")
    c3 = line["also"]["configs3"]
    assert c3["n_gpus"] == 2 and c3["config"]["cells"] == 4000 and c3["value"] > 0
    assert 0.9 < c3["config"]["imbalance_max_over_mean_bytes"] < 1.3
    e2e = line["also"]["e2e"]   # every rank from its own pinned host buffer at once
    assert "error" not in e2e and e2e["n_gpus"] == 2 and e2e["value"] > 0 and e2e["imbalance_max_over_mean_time"] >= 1.0


def test_bench_plan_only_prints_every_ranks_memory_plan():
    """`bench.py --gpus N --plan-only` (the first thing to run on an 8-GPU node): no kernel, every rank one JSON line with the bytes
    each leg would hold against its GPU's free memory, rank 0 the host's pinned total against MemAvailable; exit code 0 = it fits."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--dist-backend", "gloo", "--plan-only"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert sorted(l["rank"] for l in lines) == [0, 1] and all(l["plan_only"] and l["fits"] for l in lines)
    l0 = [l for l in lines if l["rank"] == 0][0]
    assert l0["host"]["fits"] and l0["host"]["pinned_bytes_all_ranks"] > 2 * 6e9          # two ranks' e2e buffers
    c1, c3 = l0["legs"]["configs1"], l0["legs"]["configs3"]
    assert 6.5e9 < c1["input_bytes"] < 7.3e9 and 3.5e8 < c1["reads"] < 4.5e8              # the headline's sample, to the byte the generator would make
    assert 4.0e10 < c3["input_bytes"] < 4.8e10 and c3["resident_peak_bytes"] < l0["device_total_bytes"]
    # a rank without a device ends the job with one line instead of a barrier the others hang in
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--plan-only"]   # (no --share-gpu: rank 1 asks for cuda:1)
    import torch
    if torch.cuda.device_count() == 1:
        r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode != 0 and "no GPU 1" in r.stderr and "stopping the other" in r.stderr


def test_bench_eight_ranks_bookkeeping_on_one_gpu():
    """The driver's `--gpus 8` shape on the one GPU a box has: eight ranks (gloo, all on cuda:0), the configs[3] data set cut into
    eight byte-balanced ranges with every rank generating and checking its own, the e2e and atac legs' collectives made by all
    eight.  What it cannot show is eight devices' throughput; what it does show is that no rank's bookkeeping is off."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--share-gpu", "--dist-backend", "gloo", "--steps", "2", "--warmup", "1",
           "--cells", "200", "--median-reads", "1500", "--c3-cells", "4000", "--c3-mean-reads", "300", "--atac-cells", "60", "--frags-per-cell", "400",
           "--cpu-seconds", "1", "--also", "e2e,configs3,atac"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 8 and line["value"] > 0 and line["scaling"] == "weak"
    c3 = line["also"]["configs3"]
    assert "error" not in c3 and c3["n_gpus"] == 8 and c3["config"]["cells"] == 8 * 4000 and c3["cpu_baseline"]["ranks_checked"] == 8
    assert 0.9 < c3["config"]["imbalance_max_over_mean_bytes"] < 1.3
    assert line["also"]["e2e"]["n_gpus"] == 8 and "error" not in line["also"]["e2e"]
    assert line["also"]["atac"]["n_gpus"] == 8 and "error" not in line["also"]["atac"]


def test_bench_single_gpu_legs_small():
    """The default legs on a small workload: configs[2] with its tie report, configs[3], the PCIe-inclusive figure, the CLI wall."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--cells", "400", "--median-reads", "3000",
           "--c3-cells", "3000", "--c3-mean-reads", "400", "--cpu-seconds", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["roofline"]["frac"] > 0 and line["cpu_baseline"]["value"] > 0
    also = line["also"]
    for k in ("configs2", "configs3", "e2e", "cli"):
        assert k in also and "error" not in also[k], (k, also.get(k))
    assert also["configs2"]["cpu_baseline"]["parsimony_ties"]["molecules"] > 0
    assert also["e2e"]["value"] > 0 and also["cli"]["wall_s"] > 0


def test_reference_binary_hook(tmp_path, oracle):
    """If the real alevin-fry is on this box ($ALEVIN_FRY_BIN or PATH): run its quant on a directory written by our RAD
    writer and compare counts keyed by (barcode, gene) with the device path - the only check there is against the Rust
    binary itself (and of the RAD codec against libradicl's reader).  Skipped when there is no binary."""
    exe = os.environ.get("ALEVIN_FRY_BIN") or shutil.which("alevin-fry")
    if not exe:
        pytest.skip("no alevin-fry binary on this box")
    s = synth.synth(61, [4000, 1500, 600, 260, 120, 60, 7], num_genes=150, txp_per_gene=3, dup=0.5, cross=0.3, umi_err=0.02)
    tg, b, off = _make_dir(tmp_path / "in", s)
    ref_out, our_out = tmp_path / "ref", tmp_path / "ours"
    r = subprocess.run([exe, "quant", "-i", str(tmp_path / "in"), "-m", tg, "-o", str(ref_out), "-r", "cr-like", "-t", "4", "--use-mtx"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([CLI, "quant", "-i", str(tmp_path / "in"), "-m", tg, "-o", str(our_out), "-r", "cr-like"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr

    def load(o):
        rows = (o / "alevin/quants_mat_rows.txt").read_text().split()
        cols = (o / "alevin/quants_mat_cols.txt").read_text().split()
        body = [l for l in (o / "alevin/quants_mat.mtx").read_text().splitlines() if not l.startswith("%")]
        return {(rows[int(r) - 1], cols[int(c) - 1]): float(v) for r, c, v in (l.split() for l in body[1:])}

    assert load(ref_out) == load(our_out)   # cr-like is integer-exact and order-independent


def test_submit_reader_pulls_the_bytes_through_a_callback(monkeypatch):
    """afq_submit_reader: the library asks for byte ranges of the chunk stream (several threads, pinned destinations) instead
    of being handed a buffer - what the CLI front-end does with pread on the collated file.  Same rows as afq_submit, with
    one range and with many, and a failing reader is an error, not a hang."""
    import ctypes as C

    s = synth.synth(78, [6000, 5000, 3000, 2500, 900, 700, 400, 300, 120, 80, 33, 5] * 2, num_genes=200, dup=0.4, umi_err=0.02)
    b, off = s.encode()
    b = np.ascontiguousarray(np.asarray(b))
    cfg = cfg_for(s, "cr-like")
    q = pkg.Quantifier(cfg, s.tid_to_gid, device=0)
    READ = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_size_t)
    calls = []

    def reader(user, offset, dst, n):
        calls.append((offset, n))
        C.memmove(dst, b.ctypes.data + offset, n)
        return 0

    cb = READ(reader)
    q.lib.afq_submit_reader.argtypes = [C.c_void_p, READ, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_uint32, C.c_uint64]
    q.lib.afq_submit_reader.restype = C.c_int
    hdr = np.ascontiguousarray(np.stack([b.view(np.uint32)[off // 4], b.view(np.uint32)[off // 4 + 1]], axis=1).astype(np.uint32))
    o64 = np.ascontiguousarray(off, dtype=np.uint64)
    try:
        want = q.quant_chunks(b, off)
        for rng in (None, "300000"):
            if rng:
                monkeypatch.setenv("AFQ_TEST_RANGE_BYTES", rng)
            rc = q.lib.afq_submit_reader(q._h, cb, None, b.nbytes, o64.ctypes.data_as(C.POINTER(C.c_uint64)), hdr.ctypes.data_as(C.POINTER(C.c_uint32)), len(off), 0)
            assert rc == 0, q.lib.afq_last_error(q._h)
            assert_same_result(q.collect(), want, what=f"reader, ranges {rng}")
        assert sum(n for _, n in calls) >= 2 * b.nbytes
        bad = READ(lambda user, offset, dst, n: -1)
        rc = q.lib.afq_submit_reader(q._h, bad, None, b.nbytes, o64.ctypes.data_as(C.POINTER(C.c_uint64)), hdr.ctypes.data_as(C.POINTER(C.c_uint32)), len(off), 0)
        assert rc == pkg._abi.AFQ_ERR_BAD_INPUT
        assert_same_result(q.quant_chunks(b, off), want)   # the context is usable afterwards
    finally:
        q.close()
