"""CPU tests of the host front-end pieces in libafquant.so (no GPU calls): Rust-style float formatting,
the snappy frame decoder and the RAD prelude parser (include/afquant_host.h)."""
import ctypes as C
import math
import os

import numpy as np
import pytest

from util import ROOT, pkg

rad = pkg.rad


@pytest.fixture(scope="module")
def lib():
    L = pkg.load_library()
    L.afq_format_f32.argtypes = [C.c_float, C.c_char_p, C.c_size_t]
    L.afq_format_f32.restype = C.c_int
    L.afq_snappy_frame_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    L.afq_snappy_frame_decode.restype = C.c_int64
    L.afq_host_last_error.restype = C.c_char_p

    class RadInfo(C.Structure):
        _fields_ = [("ref_count", C.c_uint64), ("num_chunks", C.c_uint64), ("first_chunk_off", C.c_uint64),
                    ("is_paired", C.c_uint32), ("cblen", C.c_uint32), ("ulen", C.c_uint32), ("bc_bytes", C.c_uint32),
                    ("umi_bytes", C.c_uint32)]

    L.RadInfo = RadInfo
    L.afq_rad_parse_prelude.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(RadInfo)]
    L.afq_rad_parse_prelude.restype = C.c_int
    return L


def fmt(lib, v):
    b = C.create_string_buffer(64)
    n = lib.afq_format_f32(v, b, 64)
    return b.value.decode()[:n]


def test_rust_display_of_f32(lib):
    """featureDump.txt and the MTX values are written with Rust's `{}` (src/quant.rs:1231-1262; sprs):
    shortest digits that round-trip as f32, positional notation, NaN/inf spelled the Rust way."""
    known = {1.0: "1", 0.5: "0.5", 0.1: "0.1", 1e-7: "0.0000001", 16777216.0: "16777216", 2.5: "2.5",
             1.5e10: "15000000000", 0.0: "0", 3.0: "3", 0.33333334: "0.33333334", 123.456: "123.456"}
    for v, s in known.items():
        assert fmt(lib, v) == s, (v, fmt(lib, v))
    assert fmt(lib, float("nan")) == "NaN" and fmt(lib, float("inf")) == "inf" and fmt(lib, -float("inf")) == "-inf"
    rng = np.random.default_rng(0)
    for x in np.concatenate((rng.random(200), rng.random(200) * 1e4, rng.integers(1, 5000, 200))).astype(np.float32):
        s = fmt(lib, float(x))
        assert "e" not in s and np.float32(float(s)) == x  # round-trips, never an exponent


def test_snappy_frame_decode(lib):
    rng = np.random.default_rng(1)
    data = bytes(rng.integers(0, 256, 200_000, dtype=np.uint8)) + b"abcdabcdabcd" * 1000
    enc = rad.snappy_frame_encode(data, chunk=50_000)
    out = C.create_string_buffer(len(data))
    n = lib.afq_snappy_frame_decode(enc, len(enc), out, len(data))
    assert n == len(data) and out.raw == data
    # a hand-assembled compressed block with copies: "abcd" literal, copy(len 8, off 4), copy-2byte(len 5, off 12)
    block = bytes([17]) + bytes([3 << 2]) + b"abcd" + bytes([((8 - 4) << 2) | 1, 4]) + bytes([((5 - 1) << 2) | 2, 12, 0])
    want = b"abcd" + b"abcdabcd" + b"abcda"
    crc = rad._crc32c(want)
    m = (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF
    body = m.to_bytes(4, "little") + block
    frame = b"\xff\x06\x00\x00sNaPpY" + b"\x00" + len(body).to_bytes(3, "little") + body
    out = C.create_string_buffer(64)
    assert lib.afq_snappy_frame_decode(frame, len(frame), out, 64) == len(want) and out.raw[: len(want)] == want
    bad = bytearray(enc)
    bad[20] ^= 0xFF  # payload corruption -> CRC mismatch
    assert lib.afq_snappy_frame_decode(bytes(bad), len(bad), None, 0) < 0
    # blocks compressed by Google's snappy (through pyarrow): RAD-like data is full of back-references (the barcode
    # repeats in every record), overlapping copies (runs) included; many chunks, so the threaded path runs too
    pa = pytest.importorskip("pyarrow")
    if not pa.Codec.is_available("snappy"):
        pytest.skip("pyarrow without snappy")
    rec = np.zeros((300_000, 4), np.uint32)
    rec[:, 0] = 1; rec[:, 1] = 0x5A5A1234; rec[:, 2] = rng.integers(0, 1 << 24, len(rec)); rec[:, 3] = rng.integers(0, 5000, len(rec)) | 0x80000000
    data2 = rec.tobytes() + b"\x00" * 70_000 + bytes(rng.integers(0, 4, 100_000, dtype=np.uint8))
    enc2 = rad.snappy_frame_encode(data2, chunk=65_536)
    assert len(enc2) < 0.8 * len(data2)  # it did compress (i.e. the real codec ran)
    out2 = C.create_string_buffer(len(data2))
    assert lib.afq_snappy_frame_decode(enc2, len(enc2), out2, len(data2)) == len(data2) and out2.raw == data2


def test_rad_prelude_roundtrip(lib):
    names = [f"ENST{i:08d}.{i % 7}" for i in range(50)]
    pre = rad.rad_prelude(names, 123, 16, 12)
    info = lib.RadInfo()
    assert lib.afq_rad_parse_prelude(pre + b"\x00" * 16, len(pre) + 16, C.byref(info)) == 0
    assert (info.ref_count, info.num_chunks, info.first_chunk_off) == (50, 123, len(pre))
    assert (info.cblen, info.ulen, info.bc_bytes, info.umi_bytes, info.is_paired) == (16, 12, 4, 4, 0)
    pre = rad.rad_prelude(names[:3], 1, 20, 10, bc_bytes=8, umi_bytes=4)
    assert lib.afq_rad_parse_prelude(pre, len(pre), C.byref(info)) == 0 and (info.bc_bytes, info.umi_bytes) == (8, 4)
    assert lib.afq_rad_parse_prelude(pre[:20], 20, C.byref(info)) < 0  # truncated


def _quantify(lib, in_dir, tg, out_dir, resolution="cr-like"):
    class Opts(C.Structure):
        _fields_ = [("input_dir", C.c_char_p), ("tg_map", C.c_char_p), ("output_dir", C.c_char_p), ("resolution", C.c_char_p),
                    ("filter_list", C.c_char_p), ("cmdline", C.c_char_p), ("num_threads", C.c_uint32), ("small_thresh", C.c_uint32),
                    ("umi_edit_dist", C.c_int32), ("large_graph_thresh", C.c_int32), ("init_uniform", C.c_uint32), ("dump_eq", C.c_uint32),
                    ("num_bootstraps", C.c_uint32), ("device", C.c_uint32), ("batch_bytes", C.c_uint64), ("sa_model", C.c_uint32),
                    ("summary_stat", C.c_uint32), ("boot_seed", C.c_uint64), ("devices", C.POINTER(C.c_int32)), ("n_devices", C.c_uint32),
                    ("reserved", C.c_uint32)]

    o = Opts(str(in_dir).encode(), str(tg).encode(), str(out_dir).encode(), resolution.encode(), None, b"test", 1, 100, -1, -1, 0, 0, 0, 0, 0, 0, 0, 0,
             None, 0, 0)
    lib.afq_quantify.argtypes = [C.POINTER(Opts)]
    lib.afq_quantify.restype = C.c_int
    rc = lib.afq_quantify(C.byref(o))
    return rc, lib.afq_host_last_error().decode()


def test_quantify_refuses_record_layouts_it_would_misread(lib, tmp_path):
    """The device decoders walk `na, b, u, na x u32` records (src/convert.rs:124-144); a RAD file whose read- or
    alignment-level tags say otherwise (multi-barcode data, extra per-read tags) is refused before anything reaches the
    device - never decoded on a guess.  All of these fail ahead of the first device call, so they run without a GPU."""
    import os

    s = pkg.synth.synth(9, [30, 20], num_genes=5)
    b, off = s.encode()
    names = [f"t{i}" for i in range(len(s.tid_to_gid))]
    rows = [(names[i], f"g{int(s.tid_to_gid[i])}") for i in range(len(names))]
    good = tmp_path / "good"
    tg = rad.write_quant_input_dir(str(good), b, len(off), names, rows)
    pre = rad.rad_prelude(names, len(off), 16, 12)
    raw = open(good / "map.collated.rad", "rb").read()
    assert raw[:len(pre)] == pre

    def variant(name, new_prelude):
        d = tmp_path / name
        os.makedirs(d)
        for f in ("generate_permit_list.json", "collate.json", "t2g.tsv"):
            (d / f).write_bytes((good / f).read_bytes())
        (d / "map.collated.rad").write_bytes(new_prelude + raw[len(pre):])
        return d

    # an extra read-level tag after (b, u): one more u32 per record that the walk would take for alignment words
    extra_read = pre.replace((2).to_bytes(2, "little") + b"\x01\x00b\x03" + b"\x01\x00u\x03",
                             (3).to_bytes(2, "little") + b"\x01\x00b\x03" + b"\x01\x00u\x03" + b"\x01\x00x\x03")
    assert extra_read != pre
    rc, msg = _quantify(lib, variant("extra_read", extra_read), tg, tmp_path / "o1")
    assert rc == pkg._abi.AFQ_ERR_UNSUPPORTED and "read-level tags" in msg
    # a second alignment-level tag
    tag = (1).to_bytes(2, "little") + (20).to_bytes(2, "little") + b"compressed_ori_refid" + b"\x03"
    assert tag in pre
    extra_aln = pre.replace(tag, (2).to_bytes(2, "little") + tag[2:] + b"\x01\x00p\x03")
    rc, msg = _quantify(lib, variant("extra_aln", extra_aln), tg, tmp_path / "o2")
    assert rc == pkg._abi.AFQ_ERR_UNSUPPORTED and "alignment-level tags" in msg
    # a chunk that holds no record (the reference panics on it, quant.rs:756) / a truncated last chunk
    empty = raw[:len(pre)] + (8).to_bytes(4, "little") + (0).to_bytes(4, "little") + raw[len(pre):]
    d = variant("empty_chunk", pre)
    (d / "map.collated.rad").write_bytes(empty)
    rc, msg = _quantify(lib, d, tg, tmp_path / "o2b")
    assert rc == pkg._abi.AFQ_ERR_BAD_INPUT and "holds no record" in msg
    # multi-barcode (Flex) files: three barcode levels / unequal widths are refused, and so is a collation manifest
    # that does not tile as the layout afq_host.cpp documents
    mb = rad.rad_prelude_multi_bc(names, len(off), 8, 16, 12)
    d = variant("mb_manifest", mb)
    (d / "collation_manifest.bin").write_bytes(b"\x01\x02\x03")
    rc, msg = _quantify(lib, d, tg, tmp_path / "o2c")
    assert rc == pkg._abi.AFQ_ERR_UNSUPPORTED and "collation_manifest.bin" in msg
    rc, msg = _quantify(lib, variant("mb_three", mb.replace(b"\x02\x00\x08\x00\x10\x00", b"\x03\x00\x08\x00\x10\x00")), tg, tmp_path / "o2d")
    assert rc == pkg._abi.AFQ_ERR_UNSUPPORTED and "two barcode levels" in msg
    # no generate_permit_list.json (src/main.rs:733-734); unknown resolution; -b with a plain resolution (main.rs:713-728)
    os.remove(good / "generate_permit_list.json")
    rc, msg = _quantify(lib, good, tg, tmp_path / "o3")
    assert rc != 0 and "generate_permit_list.json" in msg
    rc, msg = _quantify(lib, good, tg, tmp_path / "o4", resolution="full")
    assert rc == pkg._abi.AFQ_ERR_INVALID_ARG


def test_infer_files_input_errors_are_reported(lib, tmp_path):
    """`afquant infer` (src/infer.rs:61-111): malformed inputs are reported, not crashed on; all of these fail before the
    first device call."""
    import gzip

    class IOpts(C.Structure):
        _fields_ = [("count_mat", C.c_char_p), ("eq_labels", C.c_char_p), ("output_dir", C.c_char_p), ("filter_list", C.c_char_p),
                    ("usa_mode", C.c_uint32), ("num_threads", C.c_uint32), ("device", C.c_uint32), ("reserved", C.c_uint32)]

    lib.afq_infer_files.argtypes = [C.POINTER(IOpts)]
    lib.afq_infer_files.restype = C.c_int

    def run(mtx_text, eq_text, rows="ACGT\nTTTT\n", cols="g0\ng1\ng2\n"):
        d = tmp_path / f"case{run.n}"
        run.n += 1
        d.mkdir()
        (d / "geqc_counts.mtx").write_text(mtx_text)
        with gzip.open(d / "gene_eqclass.txt.gz", "wt") as f:
            f.write(eq_text)
        if rows is not None:
            (d / "quants_mat_rows.txt").write_text(rows)
        if cols is not None:
            (d / "quants_mat_cols.txt").write_text(cols)
        o = IOpts(str(d / "geqc_counts.mtx").encode(), str(d / "gene_eqclass.txt.gz").encode(), str(d / "out").encode(), None, 0, 1, 0, 0)
        rc = lib.afq_infer_files(C.byref(o))
        return rc, lib.afq_host_last_error().decode()

    run.n = 0
    good_mtx = "%%MatrixMarket matrix coordinate real general\n% written by sprs\n2 2 2\n1 1 3\n2 2 1\n"
    good_eq = "3\n2\n0\t1\t0\n2\t1\n"
    rc, msg = run("not a matrix\n", good_eq)
    assert rc == pkg._abi.AFQ_ERR_BAD_INPUT and "MatrixMarket" in msg
    rc, msg = run(good_mtx.replace("coordinate real", "array real"), good_eq)
    assert rc == pkg._abi.AFQ_ERR_BAD_INPUT
    rc, msg = run(good_mtx.replace("2 2 1\n", "2 9 1\n"), good_eq)          # column beyond the declared size
    assert rc == pkg._abi.AFQ_ERR_BAD_INPUT and "bad entry" in msg
    rc, msg = run(good_mtx, "3\n2\n0\t1\t7\n")                                # class id beyond the declared count
    assert rc == pkg._abi.AFQ_ERR_BAD_INPUT and "out of range" in msg
    rc, msg = run(good_mtx.replace("2 2 2", "2 5 2"), good_eq)               # more columns than classes
    assert rc == pkg._abi.AFQ_ERR_BAD_INPUT and "more columns" in msg
    rc, msg = run(good_mtx, good_eq, rows=None)                               # barcodes live next to the matrix (infer.rs:113-133)
    assert rc == pkg._abi.AFQ_ERR_BAD_INPUT and "quants_mat_rows.txt" in msg
    rc, msg = run(good_mtx, good_eq, cols=None)
    assert rc == pkg._abi.AFQ_ERR_BAD_INPUT and "column (gene) names" in msg


def test_tail_model_of_the_host_generator():
    """The label-length tail of the workload generator (csrc/afq_synth_model.h, host twin): records stay well formed (refs
    ascending and distinct, inside the reference), the mean and the maximum are what DESIGN 3.5 quotes (E[na] ~ 3.1 at
    p = 0.65, never over 64), and the extra refs are drawn per MOLECULE: among reads that share a UMI and their first ref, the
    share whose lists are equal is the plain model's (whose one to three refs already differ from read to read) - drawn per
    read, as a first version did, hardly two reads of a molecule would carry the same list."""
    import importlib

    sn = importlib.import_module("alevin-fry_amd.synth_native")
    kw = dict(seed=5, n_cells=30, median_reads=6000, sigma=0.4, num_genes=36601, ref_count=199138)

    def scan(r, cap):
        w = r.data.view(np.uint32)
        nas, same, diff = [], 0, 0
        for c in range(30):
            o = int(r.chunk_off[c]) // 4
            nrec, p = int(w[o + 1]), o + 2
            seen = {}
            for _ in range(nrec):
                na = int(w[p])
                refs = w[p + 3:p + 3 + na] & 0x7FFFFFFF
                assert 1 <= na <= cap and (np.diff(refs.astype(np.int64)) > 0).all() and int(refs.max()) < 199138
                nas.append(na)
                key, lab = (int(w[p + 2]), int(refs[0])), tuple(refs.tolist())
                if key in seen:
                    same += seen[key] == lab
                    diff += seen[key] != lab
                else:
                    seen[key] = lab
                p += 3 + na
        return np.asarray(nas), same, diff

    tailed, plain = sn.generate(tail=0.65, **kw), sn.generate(**kw)
    nas, same, diff = scan(tailed, 64)
    nas0, same0, diff0 = scan(plain, 3)
    assert 2.8 < nas.mean() < 3.5 and nas.max() > 12 and (nas >= 5).mean() > 0.15 and 1.2 < nas0.mean() < 1.6
    assert plain.n_reads == tailed.n_reads and len(plain.data) < len(tailed.data)   # the same reads, longer records
    assert same0 > 1000 and same > 0.8 * same0, (same, diff, same0, diff0)


def test_reference_pin_script_round_trip(tmp_path):
    """tests/make_reference_pin.py: `make` writes the input directory + the oracle's counts keyed by (barcode, gene); `compare`
    joins an alevin-fry output directory with them.  Here the "output directory" is written from the expected counts themselves
    (MatrixMarket, rows and columns shuffled as the reference's are: completion order) - the join must come out clean, and a
    changed count must be seen."""
    import random
    import subprocess
    import sys

    script = os.path.join(ROOT, "tests", "make_reference_pin.py")
    d = tmp_path / "pin"
    r = subprocess.run([sys.executable, script, "make", str(d), "--usa", "--resolution", "cr-like"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (d / "in" / "map.collated.rad").exists() and (d / "RUN.sh").exists()
    ent = [l.split("\t") for l in (d / "expected_counts.tsv").read_text().splitlines()]
    assert len(ent) > 100
    rows = sorted({e[0] for e in ent}); cols = sorted({e[1] for e in ent})
    random.Random(3).shuffle(rows); random.Random(4).shuffle(cols)
    out = d / "ref_out" / "alevin"
    out.mkdir(parents=True)
    (out / "quants_mat_rows.txt").write_text("\n".join(rows) + "\n")
    (out / "quants_mat_cols.txt").write_text("\n".join(cols) + "\n")
    ri = {b: i + 1 for i, b in enumerate(rows)}; ci = {g: i + 1 for i, g in enumerate(cols)}
    body = [f"{ri[b]} {ci[g]} {float(v)}" for b, g, v in ent]
    (out / "quants_mat.mtx").write_text("%%MatrixMarket matrix coordinate real general\n" + f"{len(rows)} {len(cols)} {len(body)}\n" + "\n".join(body) + "\n")
    r = subprocess.run([sys.executable, script, "compare", str(d)], capture_output=True, text=True)
    assert r.returncode == 0 and "differing 0," in r.stdout, r.stdout + r.stderr
    body[5] = " ".join(body[5].split()[:2] + ["7777.0"])
    (out / "quants_mat.mtx").write_text("%%MatrixMarket matrix coordinate real general\n" + f"{len(rows)} {len(cols)} {len(body)}\n" + "\n".join(body) + "\n")
    r = subprocess.run([sys.executable, script, "compare", str(d)], capture_output=True, text=True)
    assert r.returncode == 1 and "differing 1," in r.stdout


def test_molecule8_column_is_the_general_rule_on_host(tmp_path):
    """csrc/afq_pug2.hip: molecule8_column (a label of 5..8 refs resolved in eight registers: a 19-exchange sorting network, repeats
    to padding, a second pass) must give what genes_of + molecule_column_n (afq_pug_common.h) give - column, EM class words and
    descriptors, error codes.  The functions' own source text is compiled for the host with a stand-in PugCtx and run on 1.6 M
    random labels (small gene spaces: many repeats and spliced / unspliced siblings), USA and not, EM and not."""
    import shutil
    import subprocess

    if not shutil.which("g++"):
        pytest.skip("no g++")
    src = open(os.path.join(ROOT, "alevin-fry_amd", "csrc", "afq_pug2.hip")).read()
    com = open(os.path.join(ROOT, "alevin-fry_amd", "csrc", "afq_pug_common.h")).read()

    def grab(text, start, end):
        a = text.index(start)
        return text[a:text.index(end, a)]

    parts = [grab(com, "template <typename GetRef>\n__device__ __forceinline__ uint32_t genes_of(", "// One resolved molecule with gene label"),
             grab(com, "__device__ __forceinline__ uint32_t molecule_column_n(", "__device__ __forceinline__ void emit_molecule("),
             grab(src, "__device__ __forceinline__ void sort8", "__device__ __forceinline__ uint32_t molecule8_column"),
             grab(src, "__device__ __forceinline__ uint32_t molecule8_column", "// L8: labels of 5..8 refs by their own lane")]
    host = r'''#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <random>
#define __device__
#define __forceinline__ inline
using std::min; using std::max;
constexpr uint32_t kMaxGenesPerLabel = 64, kErrPugLimit = 7, kErrSlotRange = 9;
struct PugCtx { const uint32_t* t2g; uint32_t gene_level, usa, em, num_rows, uo, ao, lab_cap; uint32_t* labw; uint32_t* labd; uint32_t* s_cnt; };
static uint32_t atomicAdd(uint32_t* p, uint32_t v) { uint32_t o = *p; *p += v; return o; }
''' + "".join(parts) + r'''
int main() {
    std::mt19937 rng(7);
    long n = 0, bad = 0;
    for (int usa = 0; usa < 2; ++usa) for (int em = 0; em < 2; ++em) for (int it = 0; it < 400000; ++it) {
        const uint32_t G = usa ? 2 * (3 + rng() % 6) : 4 + rng() % 8;
        const uint32_t rows = usa ? (G / 2) * 3 - (it % 5 == 0 ? 2 : 0) : G - (it % 5 == 0 ? 1 : 0);   // (now and then a column out of range: the error path)
        uint32_t la[64], da[64], lb[64], db[64], ca[4] = {0, 0, 0, 0}, cb[4] = {0, 0, 0, 0};
        PugCtx A{nullptr, 1, (uint32_t)usa, (uint32_t)em, rows, ((G / 2) * 3) / 3, 2 * (((G / 2) * 3) / 3), (uint32_t)(it % 7 == 0 ? 3 : 40), la, da, ca};
        PugCtx B = A; B.labw = lb; B.labd = db; B.s_cnt = cb;
        const uint32_t len = 5 + rng() % 4;
        uint32_t refs[8], g8[8];
        for (uint32_t q = 0; q < 8; ++q) { refs[q] = rng() % G; g8[q] = q < len ? refs[q] : 0xFFFFFFFFu; }
        uint32_t g[kMaxGenesPerLabel];
        const uint32_t ng = genes_of(A, len, [&](uint32_t j) { return refs[j]; }, g);
        const uint32_t want = molecule_column_n(A, g, ng);
        const uint32_t got = molecule8_column(B, g8);
        bool same = want == got && ca[1] == cb[1] && ca[2] == cb[2] && ca[3] == cb[3];
        for (uint32_t k = 0; same && k < ca[1] && k < 40; ++k) same = la[k] == lb[k];
        for (uint32_t k = 0; same && k < 2 * ca[2] && k < 40; ++k) same = da[k] == db[k];
        ++n; if (!same) { if (bad < 5) printf("MISMATCH usa=%d em=%d len=%u want=%u got=%u\n", usa, em, len, want, got); ++bad; }
    }
    printf("%ld cases, %ld mismatches\n", n, bad);
    return bad != 0;
}
'''
    (tmp_path / "t.cpp").write_text(host)
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", str(tmp_path / "t"), str(tmp_path / "t.cpp")], check=True, capture_output=True)
    r = subprocess.run([str(tmp_path / "t")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 mismatches" in r.stdout


def test_crlike_rule_functions_are_the_oracles_on_host(tmp_path):
    """csrc/afq_kernels.hip: col_from_pairs (a UMI's three in-slot (gene, reads) counters -> column) and col_from_candidates (the
    same rule table on order-free aggregates, any number of genes) are the device's statement of the winner-take-all rule with
    the USA spliced / unspliced / ambiguous table (quant.rs:557-605, utils.rs:688-753).  Their own source text is compiled for the
    host and asked every UMI of one to five genes out of six gene ids with one to three reads each, USA and not; the oracle
    (crlike_walk + extract_counts in oracle/afq_oracle.cpp) is asked the same UMIs, one per cell.  No GPU."""
    import itertools
    import shutil
    import subprocess
    import sys

    if not shutil.which("g++"):
        pytest.skip("no g++")
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as ora

    cases = []   # (usa, ((gene, reads), ...))
    for usa in (0, 1):
        for k in range(1, 6):
            for genes in itertools.combinations(range(6), k):
                for counts in itertools.product((1, 2, 3) if k <= 3 else (1, 2), repeat=k):
                    cases.append((usa, tuple(zip(genes, counts))))
    # the oracle: every case a cell of its own holding one UMI; a read = one alignment to transcript g (tid_to_gid = identity)
    want = []
    for usa in (0, 1):
        sub = [c for c in cases if c[0] == usa]
        cells = [(100 + i, [(7, [g]) for g, n in gc for _ in range(n)]) for i, (_, gc) in enumerate(sub)]
        b, off = rad.encode_cells(cells, 4, 4)
        cfg = pkg.WorkerConfig.for_resolution("cr-like", usa_mode=bool(usa), num_genes=6, num_rows=9 if usa else 6, small_thresh=0)
        res = ora.quant(cfg, np.arange(6, dtype=np.uint32), b, off)
        for i in range(len(sub)):
            g, v = res.row(i)
            assert len(g) <= 1 and all(x == 1.0 for x in v)
            want.append(int(g[0]) if len(g) else 0xFFFFFFFF)
    src = open(os.path.join(ROOT, "alevin-fry_amd", "csrc", "afq_kernels.hip")).read()

    def grab(start, end):
        a = src.index(start)
        return src[a:src.index(end, a)]

    host = r'''#include <cstdint>
#include <cstdio>
#include <algorithm>
#define __device__
#define __forceinline__ inline
using std::max;
struct ResolveCfg { uint32_t usa, num_rows, uo, ao, mode, pa, sort_only; };
static inline bool is_spliced(uint32_t g) { return (g & 1u) == 0; }
static inline bool same_gene(uint32_t a, uint32_t b) { return (a & ~1u) == (b & ~1u); }
constexpr uint32_t kNoCol = 0xFFFFFFFFu;
''' + grab("__device__ __forceinline__ uint32_t col_from_pairs(", "// A UMI seen with more than kHtPairs genes parks") + \
        grab("template <int N>\n__device__ __forceinline__ uint32_t col_from_candidates(", "// cr-like-em: a UMI whose winners are one output column") + r'''
int main() {
    unsigned usa, k;
    while (scanf("%u %u", &usa, &k) == 2) {
        uint32_t g[8], c[8];
        for (unsigned i = 0; i < k; ++i) if (scanf("%u %u", &g[i], &c[i]) != 2) return 2;
        const ResolveCfg rc{usa, usa ? 9u : 6u, 3u, 6u, 0u, 0u, 0u};
        uint32_t a = kNoCol;
        if (k <= 3) {
            uint32_t p[3] = {0, 0, 0};   // (an unused counter: 0 reads)
            for (unsigned i = 0; i < k; ++i) p[(i + g[0]) % 3] = (g[i] << 12) | c[i];   // the counters in any slot order
            a = col_from_pairs(p[0], p[1], p[2], rc);
        }
        uint32_t cg[8], cc[8];   // (the merge's register arrays: candidates in any order, a count of 0 = no candidate)
        for (unsigned i = 0; i < 8; ++i) { cg[i] = kNoCol; cc[i] = 0; }
        for (unsigned i = 0; i < k; ++i) { cg[(3 * i + 1) % 8] = g[k - 1 - i]; cc[(3 * i + 1) % 8] = c[k - 1 - i]; }
        const uint32_t b = col_from_candidates(cg, cc, rc);
        printf("%u %u\n", a, b);
    }
    return 0;
}
'''
    (tmp_path / "t.cpp").write_text(host)
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", str(tmp_path / "t"), str(tmp_path / "t.cpp")], check=True, capture_output=True)
    text = "".join(f"{usa} {len(gc)} " + " ".join(f"{g} {n}" for g, n in gc) + "\n" for usa, gc in cases)
    r = subprocess.run([str(tmp_path / "t")], input=text, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = [tuple(int(x) for x in line.split()) for line in r.stdout.splitlines()]
    assert len(got) == len(cases) == len(want)
    for (usa, gc), (a, b), w in zip(cases, got, want):
        assert b == w, ("col_from_candidates", usa, gc, b, w)
        if len(gc) <= 3:
            assert a == w, ("col_from_pairs", usa, gc, a, w)


def test_parsimony_molecule_rules_are_the_oracles_on_host(tmp_path):
    """csrc/afq_pug_common.h: how a resolved molecule's label becomes a count - genes_of4 + molecule4_column (labels of up to four
    refs, in registers; molecule2_column for one or two genes) and genes_of + molecule_column_n (any label) - against the oracle's
    `parsimony` on cells of ONE read (one vertex, no edge: the molecule is the read's label; quant.rs:974-1024, utils.rs:688-753).
    Every label of one to six of nine transcripts (two spliced transcripts per gene and, in USA mode, the genes' unspliced ones),
    USA and not.  The functions' own source text, compiled for the host.  No GPU."""
    import itertools
    import shutil
    import subprocess
    import sys

    if not shutil.which("g++"):
        pytest.skip("no g++")
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as ora

    t2g = {0: [0, 0, 1, 1, 2, 2, 3, 4, 5], 1: [0, 0, 2, 2, 4, 4, 1, 3, 5]}   # non-USA: six genes; USA: gene ids 2g (spliced) / 2g + 1 (unspliced)
    labels = [lab for k in range(1, 7) for lab in itertools.combinations(range(9), k)]
    want = {}
    for usa in (0, 1):
        cells = [(500 + i, [(3, list(lab))]) for i, lab in enumerate(labels)]
        b, off = rad.encode_cells(cells, 4, 4)
        cfg = pkg.WorkerConfig.for_resolution("parsimony", usa_mode=bool(usa), num_genes=6, num_rows=9 if usa else 6, small_thresh=0)
        res = ora.quant(cfg, np.asarray(t2g[usa], np.uint32), b, off)
        for i, lab in enumerate(labels):
            g, v = res.row(i)
            assert len(g) <= 1 and all(x == 1.0 for x in v)
            want[(usa, lab)] = int(g[0]) if len(g) else 0xFFFFFFFF
    com = open(os.path.join(ROOT, "alevin-fry_amd", "csrc", "afq_pug_common.h")).read()

    def grab(start, end):
        a = com.index(start)
        return com[a:com.index(end, a)]

    host = r'''#include <cstdint>
#include <cstdio>
#define __device__
#define __forceinline__ inline
constexpr uint32_t kMaxGenesPerLabel = 64, kErrPugLimit = 7, kErrSlotRange = 9;
struct PugCtx { const uint32_t* t2g; uint32_t gene_level, usa, em, num_rows, uo, ao, lab_cap; uint32_t* labw; uint32_t* labd; uint32_t* s_cnt; };
static uint32_t atomicAdd(uint32_t* p, uint32_t v) { uint32_t o = *p; *p += v; return o; }
''' + grab("template <typename GetRef>\n__device__ __forceinline__ uint32_t genes_of(", "// One resolved molecule with gene label") + \
        grab("__device__ __forceinline__ uint32_t molecule_column_n(", "__device__ __forceinline__ void emit_molecule(") + \
        grab("__device__ __forceinline__ uint32_t molecule2_column(", "// Up to four refs -> their distinct gene ids") + \
        grab("__device__ __forceinline__ uint32_t genes_of4(", "// Append one column per lane that has one.") + r'''
int main() {
    static const uint32_t T2G[2][9] = {{0, 0, 1, 1, 2, 2, 3, 4, 5}, {0, 0, 2, 2, 4, 4, 1, 3, 5}};
    unsigned usa, k;
    while (scanf("%u %u", &usa, &k) == 2) {
        uint32_t t[8];
        for (unsigned i = 0; i < k; ++i) if (scanf("%u", &t[i]) != 1) return 2;
        uint32_t cnt[4] = {0, 0, 0, 0}, lw[64], ld[64];
        PugCtx C{T2G[usa], 0, usa, 0, usa ? 9u : 6u, 3u, 6u, 64, lw, ld, cnt};
        uint32_t g[kMaxGenesPerLabel];
        const uint32_t ng = genes_of(C, k, [&](uint32_t j) { return t[j]; }, g);
        const uint32_t a = molecule_column_n(C, g, ng);
        uint32_t b = 0xFFFFFFFEu;   // (no answer: the label has more than four refs)
        if (k <= 4) {
            uint32_t g4[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
            for (unsigned i = 0; i < k; ++i) g4[i] = t[k - 1 - i];   // (any order)
            bool cls = false;
            const uint32_t n4 = genes_of4(C, g4, k);
            b = molecule4_column(C, g4, n4, cls);
            if (cls) return 3;   // (classes are the EM's)
        }
        printf("%u %u %u\n", a, b, cnt[3]);
    }
    return 0;
}
'''
    (tmp_path / "t.cpp").write_text(host)
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", str(tmp_path / "t"), str(tmp_path / "t.cpp")], check=True, capture_output=True)
    order = [(usa, lab) for usa in (0, 1) for lab in labels]
    text = "".join(f"{usa} {len(lab)} " + " ".join(str(x) for x in lab) + "\n" for usa, lab in order)
    r = subprocess.run([str(tmp_path / "t")], input=text, capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stderr)
    got = [tuple(int(x) for x in line.split()) for line in r.stdout.splitlines()]
    assert len(got) == len(order)
    for key, (a, b, err) in zip(order, got):
        assert err == 0, key
        assert a == want[key], ("molecule_column_n", key, a, want[key])
        if len(key[1]) <= 4:
            assert b == want[key], ("molecule4_column", key, b, want[key])


def test_em_class_lookup_by_runs_is_the_offset_table_on_host(tmp_path):
    """csrc/afq_em2.hip: the streamed EM instances do not read a class's offset from the offset table every round - the set-up
    lays the classes out longest label first, equal lengths in one run, and `LocateByRuns` computes (first word, length) of
    class c from the run list with a cursor (labels of 63 words and more share a run that keeps its offsets).  The struct's own
    source text, compiled for the host, against the offset table on random label-length mixes: every class, walked the way the
    kernel's threads walk them (ascending, strides of 1024 and 2048 from any start), and past the end.  No GPU."""
    import shutil
    import subprocess

    if not shutil.which("g++"):
        pytest.skip("no g++")
    src = open(os.path.join(ROOT, "alevin-fry_amd", "csrc", "afq_em2.hip")).read()
    a = src.index("struct LocateByRuns {")
    struct_text = src[a:src.index("};", a) + 2]
    host = r'''#include <cstdint>
#include <cstdio>
#include <vector>
#define __device__
#define __forceinline__ inline
struct uint4 { uint32_t x, y, z, w; };
''' + struct_text + r'''
int main() {
    unsigned n_runs, K;
    if (scanf("%u %u", &n_runs, &K) != 2) return 2;
    std::vector<uint4> runs(64, uint4{0xFFFFFFFFu, 0, 0, 0xFFFFFFFFu});   // (what the kernel's LDS holds behind the list is not read: c < K)
    for (unsigned r = 0; r < n_runs; ++r) if (scanf("%u %u %u %u", &runs[r].x, &runs[r].y, &runs[r].z, &runs[r].w) != 4) return 2;
    std::vector<uint32_t> coff(K + 1);
    for (unsigned c = 0; c <= K; ++c) if (scanf("%u", &coff[c]) != 1) return 2;
    unsigned long long bad = 0, seen = 0;
    for (unsigned stride : {1024u, 2048u, 1u})
        for (unsigned start = 0; start < stride && start < K; start += (stride == 1u ? 1u : 37u)) {
            LocateByRuns loc(runs.data(), coff.data());
            for (unsigned c = start; c < K + 3 * stride; c += stride) {
                uint32_t o0 = 123, n = 456;
                loc(c, c < K, o0, n);
                if (c < K) { ++seen; if (o0 != coff[c] || n != coff[c + 1] - coff[c]) ++bad; }
                else if (o0 != 0 || n != 0) ++bad;
            }
        }
    printf("%llu %llu\n", seen, bad);
    return 0;
}
'''
    (tmp_path / "t.cpp").write_text(host)
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", str(tmp_path / "t"), str(tmp_path / "t.cpp")], check=True, capture_output=True)
    rng = np.random.default_rng(5)
    for case in range(6):
        # label lengths: mostly 2..6, a tail to 62, sometimes a few of 63 and more (one run, lengths as they come)
        n_cls = int(rng.integers(1, 9000))
        lens = np.minimum(2 + rng.geometric(0.45, n_cls) - 1, 62)
        if case % 2:
            lens[rng.integers(0, n_cls, 5)] = rng.integers(63, 200, 5)
        if case == 4:
            lens[:] = 3   # one run
        hist = np.bincount(np.minimum(lens, 63), minlength=64)
        # the set-up's layout (k_em2_setup step 1): runs from the longest labels down; the 63+ run first, its offsets ascending
        order = np.argsort(-np.minimum(lens, 63), kind="stable")
        sl = lens[order]
        coff = np.concatenate(([0], np.cumsum(sl))).astype(np.uint32)
        runs, cpos, wpos = [], 0, 0
        for b in range(63, -1, -1):
            if hist[b]:
                runs.append((cpos, wpos, 0 if b == 63 else b, cpos + int(hist[b])))
            wpos += int(sl[cpos:cpos + hist[b]].sum())
            cpos += int(hist[b])
        text = f"{len(runs)} {n_cls}\n" + "\n".join(" ".join(map(str, r)) for r in runs) + "\n" + " ".join(map(str, coff)) + "\n"
        r = subprocess.run([str(tmp_path / "t")], input=text, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        seen, bad = map(int, r.stdout.split())
        assert seen >= n_cls and bad == 0, (case, n_cls, seen, bad)


def test_sort_unique_in_row_on_host(tmp_path):
    """csrc/afq_pug_common.h: the covers sort a staged label's gene ids where they lie - in place in a lane's LDS row, by insertion,
    the sorted prefix never longer than the part already read.  The function's own source text on the host: every row of up to
    16 slots over a small alphabet with holes (0xFFFFFFFF), against sorted(set()).  No GPU."""
    import shutil
    import subprocess

    if not shutil.which("g++"):
        pytest.skip("no g++")
    src = open(os.path.join(ROOT, "alevin-fry_amd", "csrc", "afq_pug_common.h")).read()
    a = src.index("__device__ __forceinline__ uint32_t sort_unique_in_row(")
    fn = src[a:src.index("\n}\n", a) + 3]
    host = r'''#include <cstdint>
#include <cstdio>
#define __device__
#define __forceinline__ inline
''' + fn + r'''
int main() {
    unsigned n;
    while (scanf("%u", &n) == 1) {
        uint32_t row[16];
        for (unsigned i = 0; i < 16; ++i) row[i] = 0xDEADBEEFu;
        for (unsigned i = 0; i < n; ++i) if (scanf("%u", &row[i]) != 1) return 2;
        const uint32_t k = sort_unique_in_row(row, n);
        printf("%u", k);
        for (unsigned i = 0; i < k; ++i) printf(" %u", row[i]);
        printf("\n");
    }
    return 0;
}
'''
    (tmp_path / "t.cpp").write_text(host)
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", str(tmp_path / "t"), str(tmp_path / "t.cpp")], check=True, capture_output=True)
    rng = np.random.default_rng(6)
    rows = []
    for _ in range(4000):
        n = int(rng.integers(0, 17))
        v = rng.integers(0, 9, n).astype(np.uint64) * 1000 + 7
        v[rng.random(n) < 0.3] = 0xFFFFFFFF
        rows.append([int(x) for x in v])
    text = "".join(f"{len(r)} " + " ".join(map(str, r)) + "\n" for r in rows)
    out = subprocess.run([str(tmp_path / "t")], input=text, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    got = [[int(x) for x in ln.split()] for ln in out.stdout.splitlines()]
    assert len(got) == len(rows)
    for r, g in zip(rows, got):
        want = sorted(set(x for x in r if x != 0xFFFFFFFF))
        assert g[0] == len(want) and g[1:] == want, (r, g, want)


def test_atac_ref_runs_rebuild_the_column_on_host(tmp_path):
    """csrc/afq_kernels.hip, k_atac_compact: the ref column of the scATAC rows does not cross PCIe - a cell's rows are sorted by ref
    first, the kernel lists the column's runs (the row that starts one finds its end by bisection) and the host writes the column
    from the list (afq_api.cpp: std::fill_n per run).  The kernel's own source text run thread by thread on the host (blockIdx /
    threadIdx as plain variables), then the host's fill: the column must come back, on cells of no row, one row, one ref, a ref
    per row and the usual few dozen runs, with source and destination offsets that differ (the compaction).  A list that is too
    short loses runs and nothing else (the library then copies the column).  No GPU."""
    import shutil
    import subprocess

    if not shutil.which("g++"):
        pytest.skip("no g++")
    src = open(os.path.join(ROOT, "alevin-fry_amd", "csrc", "afq_kernels.hip")).read()
    a = src.index("__global__ __launch_bounds__(256) void k_atac_compact(")
    fn = src[a:src.index("\n}\n", a) + 3]
    host = r'''#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>
#define __global__
#define __launch_bounds__(x)
#define __restrict__
struct uint4 { uint32_t x, y, z, w; };
static inline uint4 make_uint4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return uint4{a, b, c, d}; }
static struct { unsigned x; } blockIdx, threadIdx;
template <typename T> static inline T atomicAdd(T* p, T v) { const T o = *p; *p += v; return o; }
static inline uint32_t __shfl_xor(uint32_t v, int) { return 0 * v; }
''' + fn + r'''
int main() {
    unsigned n_cells, cap;
    if (scanf("%u %u", &n_cells, &cap) != 2) return 2;
    std::vector<uint64_t> cell_ptr(n_cells + 1), out_ptr(n_cells + 1);
    std::vector<uint32_t> cnt(n_cells);
    uint64_t in_total = 0, out_total = 0;
    for (unsigned c = 0; c < n_cells; ++c) {
        unsigned gap;
        if (scanf("%u %u", &cnt[c], &gap) != 2) return 2;
        cell_ptr[c] = in_total; out_ptr[c] = out_total;
        in_total += cnt[c] + gap; out_total += cnt[c];   // (the input keeps a cell's capacity, the output is dense)
    }
    cell_ptr[n_cells] = in_total; out_ptr[n_cells] = out_total;
    std::vector<uint32_t> i_ref(in_total + 1, 0xABCDu), i_start(in_total + 1, 7), o_ref(out_total + 1), o_start(out_total + 1);
    std::vector<uint16_t> i_flen(in_total + 1, 100), i_cnt(in_total + 1, 1), o_flen(out_total + 1), o_cnt(out_total + 1);
    for (unsigned c = 0; c < n_cells; ++c)
        for (unsigned i = 0; i < cnt[c]; ++i) if (scanf("%u", &i_ref[cell_ptr[c] + i]) != 1) return 2;
    std::vector<uint4> runs(cap ? cap : 1);
    uint32_t ctr = 0;
    for (unsigned c = 0; c < n_cells; ++c)
        for (unsigned t = 0; t < 256; ++t) {
            blockIdx.x = c; threadIdx.x = t;
            k_atac_compact(cell_ptr.data(), out_ptr.data(), i_ref.data(), i_start.data(), i_flen.data(), i_cnt.data(), o_ref.data(), o_start.data(),
                           o_flen.data(), o_cnt.data(), nullptr, runs.data(), &ctr, cap);
        }
    // the host's side (afq_api.cpp): the column from the list, into an array that holds something else
    std::vector<uint32_t> col(out_total + 1, 0xFFFFFFFFu);
    const uint32_t have = std::min(ctr, cap);
    for (uint32_t i = 0; i < have; ++i) { const uint4 q = runs[i]; std::fill_n(col.begin() + ((((uint64_t)q.y) << 32) | q.x), (size_t)q.z, q.w); }
    uint64_t wrong = 0, unset = 0;
    for (uint64_t i = 0; i < out_total; ++i) { if (col[i] == 0xFFFFFFFFu) ++unset; else if (col[i] != o_ref[i]) ++wrong; }
    printf("%u %llu %llu %llu\n", ctr, (unsigned long long)out_total, (unsigned long long)wrong, (unsigned long long)unset);
    return 0;
}
'''
    (tmp_path / "t.cpp").write_text(host)
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", str(tmp_path / "t"), str(tmp_path / "t.cpp")], check=True, capture_output=True)
    rng = np.random.default_rng(7)
    cells = [np.zeros(0, np.int64), np.array([5]), np.full(700, 3), np.arange(600)]   # no row, one row, one ref, a ref per row
    for _ in range(40):
        n = int(rng.integers(1, 3000))
        cells.append(np.sort(rng.integers(0, int(rng.integers(1, 40)), n)))
    true_runs = sum(int(1 + np.count_nonzero(np.diff(c))) for c in cells if len(c))
    for cap in (1 << 20, true_runs, true_runs // 2):
        text = f"{len(cells)} {cap}\n" + "".join(f"{len(c)} {int(rng.integers(0, 50))}\n" for c in cells) + " ".join(" ".join(map(str, c)) for c in cells) + "\n"
        r = subprocess.run([str(tmp_path / "t")], input=text, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        ctr, total, wrong, unset = map(int, r.stdout.split())
        assert ctr == true_runs and total == sum(len(c) for c in cells) and wrong == 0
        assert (unset == 0) == (cap >= true_runs)   # (every row written exactly when the list held every run)


def test_lane_per_component_cover_is_the_oracles_on_host(tmp_path):
    """csrc/afq_pug_common.h: the cover of a component of up to four vertices in ONE lane's registers (Lane4: lane4_load with the
    edges worked out from the records' (UMI, reads) by umi_edge, lane4_rounds, lane4_permute - what k_pc_lane4 / k_pc_resume of
    csrc/afq_pugflat.hip run since round 6) against the oracle's `parsimony` (pugutils.rs:76-99, 1048-1261) on cells that hold 2..4
    vertices with labels of 1..4 refs, UMIs equal / one base / two bases apart, 1..4 reads each, USA and not, --umi-edit-dist 0 and
    1.  Both ways the device takes: (a) the records in the reference's vertex order, ties broken by position (kCoverOrdered);
    (b) the records in ANY order, a component set aside at its first tie (kCoverDefer), its uncovered vertices renumbered into the
    reference's order (lane4_permute) and resumed (kCoverResume).  The functions' own source text, compiled for the host; a "wave"
    of one lane.  No GPU."""
    import shutil
    import subprocess
    import sys
    from collections import Counter

    if not shutil.which("g++"):
        pytest.skip("no g++")
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as ora

    t2g = {0: [0, 0, 1, 1, 2, 2, 3, 4, 5], 1: [0, 0, 2, 2, 4, 4, 1, 3, 5]}
    rng = np.random.default_rng(606)
    cases = []   # (usa, exact, [(label, umi, reads)] in "any order", refpos per vertex)
    want = []
    for usa in (0, 1):
        for exact in (0, 1):
            sub, cells = [], []
            for i in range(12000 if not exact else 3000):
                n = int(rng.integers(2, 5))
                base = int(rng.integers(0, 1 << 16))
                verts = []
                while len(verts) < n:
                    k = int(rng.integers(1, 5))
                    lab = tuple(sorted(int(x) for x in rng.choice(9, size=k, replace=False)))
                    u = base
                    for _ in range(int(rng.choice([0, 0, 1, 1, 2]))):
                        u ^= int(rng.integers(1, 4)) << (2 * int(rng.integers(0, 8)))
                    if any(v[0] == lab and v[1] == u for v in verts):
                        continue
                    verts.append((lab, u, int(rng.choice([1, 1, 2, 3, 4]))))
                recs = [(u, list(lab)) for lab, u, r in verts for _ in range(r)]
                recs = [recs[j] for j in rng.permutation(len(recs))]
                first = {}
                for j, (u, lab) in enumerate(recs):
                    first.setdefault(tuple(lab), j)
                order = sorted(range(n), key=lambda v: (first[verts[v][0]], verts[v][1]))
                refpos = [order.index(v) for v in range(n)]
                sub.append((usa, exact, verts, refpos))
                cells.append((1000 + i, recs))
            b, off = rad.encode_cells(cells, 4, 4)
            cfg = pkg.WorkerConfig.for_resolution("parsimony", usa_mode=bool(usa), num_genes=6, num_rows=9 if usa else 6, small_thresh=0,
                                                  pug_exact_umi=bool(exact))
            res = ora.quant(cfg, np.asarray(t2g[usa], np.uint32), b, off)
            for i in range(len(sub)):
                g, v = res.row(i)
                assert all(float(x).is_integer() for x in v)
                want.append(Counter({int(a): int(x) for a, x in zip(g, v)}))
            cases += sub
    com = open(os.path.join(ROOT, "alevin-fry_amd", "csrc", "afq_pug_common.h")).read()

    def grab(start, end):
        a = com.index(start)
        return com[a:com.index(end, a)]

    host = r'''#include <cstdint>
#include <cstdio>
#include <algorithm>
#define __device__
#define __forceinline__ inline
struct uint4 { uint32_t x, y, z, w; };
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
struct DevStatus;
constexpr uint32_t kMaxGenesPerLabel = 64, kErrPugLimit = 7, kErrSlotRange = 9;
constexpr int kCoverOrdered = 0, kCoverDefer = 1, kCoverResume = 2;
// a wave of one lane
static inline uint32_t lane_id() { return 0; }
static inline uint64_t __ballot(bool b) { return b ? 1ull : 0ull; }
static inline bool __any(bool b) { return b; }
static inline int __popc(uint32_t x) { return __builtin_popcount(x); }
static inline int __popcll(uint64_t x) { return __builtin_popcountll(x); }
#define __builtin_amdgcn_readlane(x, l) (x)
static uint32_t atomicAdd(uint32_t* p, uint32_t v) { uint32_t o = *p; *p += v; return o; }
''' + grab("struct PugCtx {", "struct Lab {") + \
        grab("__device__ __forceinline__ uint32_t molecule2_column(", "// A class of more than kMaxGenesPerLabel genes for the EM") + \
        grab("struct Lane4 {", "// ---- components of 9..64 vertices") + r'''
int main() {
    static const uint32_t T2G[2][9] = {{0, 0, 1, 1, 2, 2, 3, 4, 5}, {0, 0, 2, 2, 4, 4, 1, 3, 5}};
    unsigned usa, exact, n;
    while (scanf("%u %u %u", &usa, &exact, &n) == 3) {
        uint4 rec[8], ord[8];
        uint32_t refpos[4] = {0, 1, 2, 3};
        for (unsigned v = 0; v < n; ++v) {
            unsigned ln, r[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, um, rc;
            if (scanf("%u", &ln) != 1) return 2;
            for (unsigned j = 0; j < ln; ++j) if (scanf("%u", &r[j]) != 1) return 2;
            if (scanf("%u %u %u", &um, &rc, &refpos[v]) != 3) return 2;
            rec[2 * v] = make_uint4(0, ln, r[0], r[1]); rec[2 * v + 1] = make_uint4(r[2], r[3], um, rc);
        }
        for (unsigned v = 0; v < n; ++v) { ord[2 * refpos[v]] = rec[2 * v]; ord[2 * refpos[v] + 1] = rec[2 * v + 1]; }
        for (int way = 0; way < 2; ++way) {
            uint32_t cnt[4] = {0, 0, 0, 0}, cols[16], lw[16], ld[16], tied_cnt = 0, tied[8];
            PugCtx C{};
            C.t2g = T2G[usa]; C.usa = usa; C.num_rows = usa ? 9u : 6u; C.uo = 3; C.ao = 6; C.exact_umi = exact;
            C.cols = cols; C.cols_cap = 16; C.labw = lw; C.labd = ld; C.lab_cap = 16; C.s_cnt = cnt; C.adj_umi = 1;
            if (way == 0) cover_lane4<kCoverOrdered>(C, ord, 0, n, (1u << n) - 1u, 0);
            else {
                Lane4 L;
                lane4_load(C, rec, 0, n, L);
                lane4_rounds<kCoverDefer>(C, L, (1u << n) - 1u, 5, &tied_cnt, tied);
                if (tied_cnt) {   // set aside: what k_pc_resume does with the entry
                    if (tied_cnt != 1 || tied[0] != 5) return 4;
                    uint32_t uc = tied[1], rank[4];
                    uint64_t key[4];
                    for (unsigned v = 0; v < 4; ++v) key[v] = v >= n ? ~0ull : ((uc >> v) & 1u) ? (uint64_t)refpos[v] : (1ull << 63) | v;
                    for (unsigned v = 0; v < 4; ++v) { rank[v] = 0; for (unsigned u = 0; u < 4; ++u) rank[v] += (u != v && key[u] < key[v]) ? 1u : 0u; if (v >= n) rank[v] = v; }
                    lane4_permute(L, rank, uc);
                    lane4_rounds<kCoverResume>(C, L, uc, 5, nullptr, nullptr);
                }
            }
            if (cnt[3]) return 3;
            std::sort(cols, cols + cnt[0]);
            printf("%u %u", tied_cnt, cnt[0]);
            for (uint32_t k = 0; k < cnt[0]; ++k) printf(" %u", cols[k]);
            printf("\n");
        }
    }
    return 0;
}
'''
    (tmp_path / "t.cpp").write_text(host)
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", str(tmp_path / "t"), str(tmp_path / "t.cpp")], check=True, capture_output=True)
    text = "".join(f"{usa} {exact} {len(verts)} " + " ".join(f"{len(lab)} " + " ".join(str(x) for x in lab) + f" {u} {r} {p}" for (lab, u, r), p in zip(verts, refpos)) + "\n"
                   for usa, exact, verts, refpos in cases)
    r = subprocess.run([str(tmp_path / "t")], input=text, capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stderr)
    lines = [[int(x) for x in line.split()] for line in r.stdout.splitlines()]
    assert len(lines) == 2 * len(cases)
    set_aside = 0
    for i, (case, w) in enumerate(zip(cases, want)):
        for way in (0, 1):
            tied, ncol, *cols = lines[2 * i + way]
            assert ncol == len(cols)
            assert Counter(cols) == w, (("ordered", "set aside + resumed")[way], case, cols, w)
        set_aside += lines[2 * i + 1][0]
    assert set_aside > len(cases) // 50   # the tie path was walked (a few percent of these components meet a tie)


def test_label_overlap_by_key_is_set_intersection_on_host(tmp_path):
    """csrc/afq_p2_shared.h + afq_pug_common.h: `klab` / `klab_overlap` - do two labels share a ref (pugutils.rs:187-204)? - is the
    whole of k_p2_check's decision about a candidate pair (csrc/afq_pugflat.hip, round 6) and of the covers' edge test.  A label of
    one or two refs sits in its 64-bit key (tag 1, 2), a longer one in the chunk behind its record header (tag 3: the refs ascending,
    the orientation bit still on).  The functions' own source text, compiled for the host, against a set intersection on 400 000
    random pairs of labels of 1..9 refs out of 12.  No GPU."""
    import shutil
    import subprocess

    if not shutil.which("g++"):
        pytest.skip("no g++")
    com = open(os.path.join(ROOT, "alevin-fry_amd", "csrc", "afq_pug_common.h")).read()
    sh = open(os.path.join(ROOT, "alevin-fry_amd", "csrc", "afq_p2_shared.h")).read()

    def grab(text, start, end):
        a = text.index(start)
        return text[a:text.index(end, a)]

    host = r'''#include <cstdint>
#include <cstdio>
#include <random>
#include <set>
#include <vector>
#include <algorithm>
#define __device__
#define __forceinline__ inline
struct PugCtx { const uint32_t* W; uint32_t HW; };
''' + grab(com, "struct Lab {", "__device__ __forceinline__ bool lab_equal(") + grab(sh, "struct KLab {", "__device__ __forceinline__ PugCtx make_ctx(") + r'''
int main() {
    std::mt19937 rng(11);
    const uint32_t HW = 3;
    long n = 0, bad = 0, yes = 0;
    for (int it = 0; it < 400000; ++it) {
        std::vector<uint32_t> W(4, 0xDEADBEEFu);
        std::set<uint32_t> refs[2];
        uint64_t key[2];
        uint32_t off[2] = {0, 0};
        for (int s = 0; s < 2; ++s) {
            const uint32_t len = 1 + rng() % 9;
            while (refs[s].size() < len) refs[s].insert(rng() % 12 + (it % 3 == 0 ? 1000000u : 0u));
            std::vector<uint32_t> v(refs[s].begin(), refs[s].end());
            if (len == 1) key[s] = (1ull << 62) | v[0];
            else if (len == 2) key[s] = (2ull << 62) | ((uint64_t)v[0] << 31) | v[1];
            else {
                key[s] = (3ull << 62) | (rng() & 0xFFFFFFu);   // (the hash itself is never looked at here)
                off[s] = (uint32_t)W.size();
                W.push_back(len); W.push_back(0x11111111u); W.push_back(0x22222222u);   // the record header: na, barcode, UMI
                for (uint32_t r : v) W.push_back(r | ((rng() & 1u) << 31));              // refs ascending, orientation bit on or off
                W.push_back(0x7FFFFFFFu);
            }
        }
        bool want = false;
        for (uint32_t r : refs[0]) want = want || refs[1].count(r);
        const bool got = klab_overlap(klab(W.data(), HW, key[0], off[0]), klab(W.data(), HW, key[1], off[1]));
        const bool rev = klab_overlap(klab(W.data(), HW, key[1], off[1]), klab(W.data(), HW, key[0], off[0]));
        ++n; yes += want;
        if (got != want || rev != want) { if (bad < 5) printf("MISMATCH it=%d want=%d got=%d rev=%d\n", it, (int)want, (int)got, (int)rev); ++bad; }
    }
    printf("%ld pairs (%ld share a ref), %ld mismatches\n", n, yes, bad);
    return bad != 0 || yes < n / 10 || yes > 9 * n / 10;
}
'''
    (tmp_path / "t.cpp").write_text(host)
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", str(tmp_path / "t"), str(tmp_path / "t.cpp")], check=True, capture_output=True)
    r = subprocess.run([str(tmp_path / "t")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 mismatches" in r.stdout
