"""CPU tests of the host front-end pieces in libafquant.so (no GPU calls): Rust-style float formatting,
the snappy frame decoder and the RAD prelude parser (include/afquant_host.h)."""
import ctypes as C
import math

import numpy as np
import pytest

from util import pkg

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
