"""GPU tests of configs[4] from RAD bytes: afq_atac_dedup_rad (record walk + filter + sort + count on the device) and the
`afquant atac deduplicate` sub-command against the oracle's restatement of src/atac/deduplicate.rs:199-237, 37-66."""
import importlib
import os
import subprocess

import numpy as np
import pytest

from test_oracle_golden import _atac_reference_cells
from util import ROOT, pkg

pytestmark = pytest.mark.gpu
rad = pkg.rad
sn = importlib.import_module("alevin-fry_amd.synth_native")
CLI = os.path.join(ROOT, "alevin-fry_amd", "csrc", "afquant")


def _q():
    cfg = pkg.WorkerConfig.for_resolution("cr-like", num_genes=1, num_rows=1)
    return pkg.Quantifier(cfg, np.zeros(1, np.uint32), device=0)


def _same(got, want):
    for a, b, name in zip(got[:6], want[:6], ("cell_ptr", "bc", "ref", "start", "frag_len", "count")):
        assert np.array_equal(a, b), name
    for k, v in want[6].items():
        assert got[6][k] == v, k


def test_atac_from_rad_reference_vector(oracle):
    """16 cells x (5 good, 2 unmapped, 1 multi-mapped) => 80 fragments (tests/atac_integration.rs:531-607), every field width."""
    for bcb in (4, 8, 2):
        cells = _atac_reference_cells()
        if bcb == 2:
            cells = [(bc & 0xFFFF, r) for bc, r in cells]
        b, off = rad.encode_atac_cells(cells, bc_bytes=bcb)
        q = _q()
        try:
            got = q.atac_dedup_rad(b, off, bc_bytes=bcb)
        finally:
            q.close()
        _same(got, oracle.atac_dedup_rad(b, off, bc_bytes=bcb))
        assert int(got[0][-1]) == 80 and got[6]["n_fallback_cells"] == 0


@pytest.mark.parametrize("piped", [False, True, "run-list-overflows-midway", "no-run-list"])
def test_atac_from_rad_matches_oracle_on_generated_cells(oracle, monkeypatch, piped):
    """config-5-like cells (20 % exact duplicates, 5 % multi-mapped, 5 % unmapped, log-normal lengths incl. >= 2000),
    host bytes and device-resident bytes; cells of 1 .. 40 000 records and an empty-record chunk.  piped: the batch goes
    through in six ranges whose results cross PCIe under the later ranges' kernels (what inputs over 128 MB do); the ref column
    of a range is written on the host from the device's list of (first row, length, ref) runs - or copied like the other
    columns from the range on in which that list overflows (forced here: after 400 runs; from the first one)."""
    if piped:
        monkeypatch.setenv("AFQ_TEST_ATAC_PIPE_BYTES", "1")
    if piped == "run-list-overflows-midway":
        monkeypatch.setenv("AFQ_TEST_ATAC_RUN_CAP", "400")
    if piped == "no-run-list":
        monkeypatch.setenv("AFQ_TEST_ATAC_RUN_CAP", "0")
    d, off = sn.generate_atac(seed=9, n_cells=300, frags_per_cell=3000, flen_sigma=1.2)
    big, boff = sn.generate_atac(seed=10, n_cells=2, frags_per_cell=40000)
    tiny_b, tiny_off = rad.encode_atac_cells([(7, [[(2, 4, 10, 100)]]), (8, [[]]), (9, [[(0, 4, 1, 1)], [(0, 4, 1, 1)]])])
    data = np.concatenate((d, big, np.frombuffer(tiny_b, np.uint8)))
    offs = np.concatenate((off, boff + len(d), tiny_off + len(d) + len(big)))
    want = oracle.atac_dedup_rad(data, offs)
    q = _q()
    try:
        _same(q.atac_dedup_rad(data, offs), want)
        import torch

        t = torch.from_numpy(data).cuda()
        _same(q.atac_dedup_rad(None, offs, d_ptr=t.data_ptr(), n_bytes=len(data)), want)
        _same(q.atac_dedup_rad(data, offs), want)   # buffers reused
    finally:
        q.close()
    assert want[6]["n_long_fragments"] > 0 and want[6]["n_multimapped"] > 0


def test_atac_walk_free_parse_falls_back_when_a_field_spells_the_barcode(oracle):
    """A start position equal to the cell's barcode makes a false candidate record start: the proof fails, the cell is
    walked record by record, the result is the same; a corrupt chunk is reported, not misread."""
    bc = 0x00ABCDEF
    recs = [[(1, 4, 1000 + i, 100)] for i in range(50)]
    recs[17] = [(1, 4, bc, 100)]          # start_pos bytes == barcode bytes, 4 bytes into the alignment
    recs[30] = [(bc, 4, 5, 50)]           # ... and as a reference id
    b, off = rad.encode_atac_cells([(bc, recs), (5, [[(0, 4, 3, 3)]] * 4)])
    q = _q()
    try:
        got = q.atac_dedup_rad(b, off)
        _same(got, oracle.atac_dedup_rad(b, off))
        bad = bytearray(b)
        bad[4:8] = (60).to_bytes(4, "little")   # the header claims 60 records
        with pytest.raises(pkg.AfqError) as e:
            q.atac_dedup_rad(bytes(bad), off)
        assert e.value.code == pkg._abi.AFQ_ERR_BAD_INPUT
    finally:
        q.close()


@pytest.mark.parametrize("rev,compressed", [(True, False), (False, True)])
def test_afquant_atac_deduplicate_writes_the_reference_bed(tmp_path, oracle, rev, compressed):
    """map.bed = chr name, start, start + frag_len, barcode (reverse-complemented with the default -d rc), count; fragments
    of >= 2000 bases are left out (src/atac/deduplicate.rs:37-66).  Compared as a multiset: the reference writes cells in
    worker-completion order."""
    import json

    d, off = sn.generate_atac(seed=12, n_cells=40, frags_per_cell=500, n_refs=3, ref_len=100000, flen_sigma=1.5)
    names = ["chr1", "chr2", "chrX"]
    pre = rad.rad_prelude_atac(names, [100000] * 3, len(off))
    os.makedirs(tmp_path / "in")
    (tmp_path / "in" / "generate_permit_list.json").write_text(json.dumps({"velo_mode": False}))
    (tmp_path / "in" / "collate.json").write_text(json.dumps({"compressed_output": compressed}))
    body = pre + d.tobytes()
    if compressed:
        (tmp_path / "in" / "map.collated.rad.sz").write_bytes(rad.snappy_frame_encode(body))
    else:
        (tmp_path / "in" / "map.collated.rad").write_bytes(body)
    r = subprocess.run([CLI, "atac", "deduplicate", "-i", str(tmp_path / "in"), "-t", "3"] + ([] if rev else ["-d", "fw"]), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    ptr, bc, ref, start, flen, cnt, st = oracle.atac_dedup_rad(d, off)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    exp = []
    for i in range(len(off)):
        s = rad.int_to_seq(int(bc[i]), 16)
        if rev:
            s = "".join(comp[ch] for ch in reversed(s))
        for k in range(int(ptr[i]), int(ptr[i + 1])):
            if flen[k] < 2000:
                exp.append(f"{names[ref[k]]}\t{start[k]}\t{int(start[k]) + int(flen[k])}\t{s}\t{cnt[k]}")
    got = (tmp_path / "in" / "map.bed").read_text().splitlines()
    assert sorted(got) == sorted(exp) and len(got) == int(ptr[-1]) - st["n_long_fragments"]
    assert f"greater than 1 mapping {st['n_multimapped']}" in r.stderr and f"not mapped pairs {st['n_not_mapped_pair']}" in r.stderr
    # a RAD whose alignment tags are not the scATAC four is refused
    (tmp_path / "in" / "map.collated.rad").write_bytes(rad.rad_prelude(names, 1, 16, 12) + d.tobytes())
    (tmp_path / "in" / "collate.json").write_text(json.dumps({"compressed_output": False}))
    r = subprocess.run([CLI, "atac", "deduplicate", "-i", str(tmp_path / "in")], capture_output=True, text=True)
    assert r.returncode != 0 and "scATAC RAD" in r.stderr


@pytest.mark.parametrize("bcb", [1, 2, 4, 8])
def test_atac_parse_at_every_alignment_and_width(oracle, bcb):
    """The parse reads the chunk as aligned dwords and rebuilds every byte position's fields with funnel shifts: chunks
    starting at each of the four byte alignments (records are 4 + bc + 11 na bytes, so successive chunks land on all of
    them; a pad in front shifts the lot), every barcode width, cells from one record to a few thousand."""
    rng = np.random.default_rng(40 + bcb)
    cells = []
    for ci, n in enumerate([1, 2, 3, 5, 17, 64, 65, 255, 256, 257, 1000, 3000]):
        recs = []
        for _ in range(n):
            k = int(rng.choice([0, 1, 1, 1, 1, 2, 3]))
            recs.append([(int(rng.integers(0, 25)), int(rng.choice([4, 4, 4, 1, 2])), int(rng.integers(0, 1 << 27)), int(rng.integers(30, 2500)))
                         for _ in range(k)])
        cells.append(((37 + 11 * ci) & ((1 << (8 * bcb)) - 1) if bcb < 8 else 0x1122334400000000 + ci, recs))
    b, off = rad.encode_atac_cells(cells, bc_bytes=bcb)
    for pad in range(4):
        data = np.concatenate((np.zeros(pad, np.uint8), np.frombuffer(b, np.uint8)))
        offs = np.asarray(off, np.uint64) + np.uint64(pad)
        want = oracle.atac_dedup_rad(data, offs, bc_bytes=bcb)
        q = _q()
        try:
            got = q.atac_dedup_rad(data, offs, bc_bytes=bcb)
        finally:
            q.close()
        _same(got, want)
        assert got[6]["n_records"] == sum(len(r) for _, r in cells)


@pytest.mark.parametrize("bcb", [1, 2, 4])
def test_atac_last_record_in_a_partial_dword_needs_no_fallback(oracle, bcb):
    """A buffer that ends in a partial dword whose last record has no alignments (4 + bc bytes): the walk-free parse takes
    the buffer's last bytes one by one, so the proof holds and no cell goes to the sequential walk (round-2 advice)."""
    hits = 0
    for k in range(1, 6):
        recs = [[(3, 4, 1000 * i + 7, 120 + i)] for i in range(k)] + [[]]
        b, off = rad.encode_atac_cells([(5, [[(1, 4, 50, 200)]] * 3), (9, recs)], bc_bytes=bcb)
        data = np.frombuffer(b, np.uint8)
        if len(data) % 4 == 0:
            continue
        hits += 1
        q = _q()
        try:
            got = q.atac_dedup_rad(data, np.asarray(off, np.uint64), bc_bytes=bcb)
        finally:
            q.close()
        _same(got, oracle.atac_dedup_rad(data, np.asarray(off, np.uint64), bc_bytes=bcb))
        assert got[6]["n_fallback_cells"] == 0
    assert hits >= 2
