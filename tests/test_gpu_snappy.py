"""The device snappy-frame decoder (csrc/afq_snappy.hip; include/afquant.h: afq_snappy_decode_device) against the bytes that
were compressed: blocks from Google's snappy (through pyarrow: back-references of every form it emits), hand-made blocks for
the element forms it rarely emits (1-byte and 4-byte offsets, run-length copies that overlap their own output, the one- to
three-byte literal lengths), uncompressed chunks, chunk sizes from one byte to the format's 65 536, destinations at every
byte alignment, and malformed blocks, which must be refused with the chunk's number.  (The reference reads map.collated.rad.sz
through snap::read::FrameDecoder, src/quant.rs:373-395.)"""
import importlib

import numpy as np
import pytest

from util import pkg

af = importlib.import_module("alevin-fry_amd.afquant")
rad = pkg.rad


def _mask(crc):
    return (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _uvarint(v):
    o = bytearray()
    while True:
        if v < 0x80:
            o.append(v)
            return bytes(o)
        o.append((v & 0x7F) | 0x80)
        v >>= 7


def frame_stream(chunks):
    """chunks: (kind, payload, data): 'raw' = uncompressed chunk; 'blk' = compressed chunk whose snappy block is `payload`
    (decoding to `data`)."""
    out = bytearray(b"\xff\x06\x00\x00sNaPpY")
    for kind, payload, data in chunks:
        body = _mask(rad._crc32c(data)).to_bytes(4, "little") + (data if kind == "raw" else payload)
        out += (b"\x01" if kind == "raw" else b"\x00") + len(body).to_bytes(3, "little") + body
    return bytes(out)


def py_raw_decompress(blk):
    """snappy's raw format, element by element (the rules of the host decoder, csrc/afq_host.cpp)."""
    p, ulen, shift = 0, 0, 0
    while True:
        v = blk[p]
        p += 1
        ulen |= (v & 0x7F) << shift
        if not v & 0x80:
            break
        shift += 7
    out = bytearray()
    while p < len(blk):
        tag = blk[p]
        p += 1
        t = tag & 3
        if t == 0:
            ln = (tag >> 2) + 1
            if ln > 60:
                nb = ln - 60
                ln = int.from_bytes(blk[p:p + nb], "little") + 1
                p += nb
            out += blk[p:p + ln]
            p += ln
        else:
            if t == 1:
                ln, off = ((tag >> 2) & 7) + 4, ((tag >> 5) << 8) | blk[p]
                p += 1
            elif t == 2:
                ln, off = (tag >> 2) + 1, int.from_bytes(blk[p:p + 2], "little")
                p += 2
            else:
                ln, off = (tag >> 2) + 1, int.from_bytes(blk[p:p + 4], "little")
                p += 4
            assert 0 < off <= len(out)
            for i in range(ln):
                out.append(out[len(out) - off])
    assert len(out) == ulen
    return bytes(out)


def lit(b):
    n = len(b) - 1
    if n < 60:
        return bytes([n << 2]) + b
    nb = 1 if n < 256 else 2 if n < 65536 else 3
    return bytes([(59 + nb) << 2]) + n.to_bytes(nb, "little") + b


def cp1(ln, off):
    assert 4 <= ln <= 11 and off < 2048
    return bytes([((ln - 4) << 2) | 1 | ((off >> 8) << 5), off & 0xFF])


def cp2(ln, off):
    assert 1 <= ln <= 64 and off < 65536
    return bytes([((ln - 1) << 2) | 2]) + off.to_bytes(2, "little")


def cp4(ln, off):
    assert 1 <= ln <= 64
    return bytes([((ln - 1) << 2) | 3]) + off.to_bytes(4, "little")


def handmade_blocks():
    rng = np.random.default_rng(3)
    rnd = lambda n: bytes(rng.integers(0, 256, n, dtype=np.uint8))   # noqa: E731
    bodies = [
        lit(b"abcd") + cp1(8, 4) + cp1(11, 3) + cp1(4, 1),                                  # copies that overlap their own output
        lit(b"x") + cp2(64, 1) + cp2(64, 1) + cp2(37, 65) + lit(b"yz") + cp2(64, 2),          # run-length of one byte, then of two
        lit(rnd(60)) + lit(rnd(61)) + lit(rnd(255)) + lit(rnd(256)) + lit(rnd(257)) + cp2(64, 700) + cp4(64, 889) + cp4(5, 1),
        lit(rnd(3000)) + b"".join(cp2(int(rng.integers(1, 65)), int(rng.integers(1, 3000))) for _ in range(400)),
        lit(rnd(1)),
        lit(rnd(65536)),                                                                    # the three-byte literal length, a full chunk
        lit(rnd(7)) + b"".join(cp1(int(rng.integers(4, 12)), int(rng.integers(1, 8))) for _ in range(2000)),
    ]
    out = []
    for body in bodies:
        data = _decode_body(body)   # (the block's first bytes are the length of what it decodes to)
        out.append(("blk", _uvarint(len(data)) + body, data))
    return out


def _decode_body(body):
    """The elements of a block that has no length prefix yet."""
    out, p = bytearray(), 0
    while p < len(body):
        tag = body[p]
        p += 1
        t = tag & 3
        if t == 0:
            ln = (tag >> 2) + 1
            if ln > 60:
                nb = ln - 60
                ln = int.from_bytes(body[p:p + nb], "little") + 1
                p += nb
            out += body[p:p + ln]
            p += ln
        else:
            if t == 1:
                ln, off = ((tag >> 2) & 7) + 4, ((tag >> 5) << 8) | body[p]
                p += 1
            elif t == 2:
                ln, off = (tag >> 2) + 1, int.from_bytes(body[p:p + 2], "little")
                p += 2
            else:
                ln, off = (tag >> 2) + 1, int.from_bytes(body[p:p + 4], "little")
                p += 4
            assert 0 < off <= len(out), (off, len(out))
            for _ in range(ln):
                out.append(out[len(out) - off])
    return bytes(out)


def test_python_reference_decoder_against_snappy():
    """(CPU) the element-by-element decoder the hand-made blocks are checked with decodes what Google's snappy wrote."""
    pa = pytest.importorskip("pyarrow")
    if not pa.Codec.is_available("snappy"):
        pytest.skip("pyarrow without snappy")
    codec = pa.Codec("snappy")
    rng = np.random.default_rng(9)
    for data in (bytes(rng.integers(0, 3, 50000, dtype=np.uint8)), b"ab" * 20000, bytes(rng.integers(0, 256, 70, dtype=np.uint8)) * 500):
        assert py_raw_decompress(codec.compress(data, asbytes=True)) == data
    for kind, blk, data in handmade_blocks():
        assert py_raw_decompress(blk) == data and len(data) <= 65536
        assert bytes(codec.decompress(blk, decompressed_size=len(data), asbytes=True)) == data   # and snappy reads the hand-made ones


@pytest.mark.gpu
def test_device_decoder_on_handmade_blocks_and_every_alignment():
    blocks = handmade_blocks()
    rng = np.random.default_rng(4)
    for lead in range(5):   # a raw chunk of `lead` bytes in front shifts every later destination by one byte
        chunks = ([("raw", b"", bytes(rng.integers(0, 256, lead, dtype=np.uint8)))] if lead else []) + blocks + [("raw", b"", b"tail-bytes")]
        want = b"".join(d for _, _, d in chunks)
        got = af.snappy_decode_device(frame_stream(chunks))
        assert got.tobytes() == want, lead


@pytest.mark.gpu
def test_device_decoder_on_snappy_compressed_rad_bytes():
    """A collated RAD (barcodes repeat record after record: back-references at short offsets) and other data through Google's
    snappy, in chunks of 1 ... 65536 bytes, every chunk compressed."""
    pa = pytest.importorskip("pyarrow")
    if not pa.Codec.is_available("snappy"):
        pytest.skip("pyarrow without snappy")
    codec = pa.Codec("snappy")
    s = pkg.synth.synth(77, [3000, 1200, 40, 9000], num_genes=200, txp_per_gene=3, dup=0.5, max_extra_na=8)
    b, _ = s.encode()
    rng = np.random.default_rng(6)
    inputs = [bytes(np.asarray(b).tobytes()), bytes(rng.integers(0, 2, 300000, dtype=np.uint8)), b"\x00" * 200000,
              bytes(rng.integers(0, 256, 100000, dtype=np.uint8)), bytes(range(256)) * 700]
    for data in inputs:
        for size in (65536, 60000, 4097, 1):
            piece = data if size > 1 else data[:300]
            chunks = [("blk", codec.compress(piece[i:i + size], asbytes=True), piece[i:i + size]) for i in range(0, len(piece), size)]
            got = af.snappy_decode_device(frame_stream(chunks))
            assert got.tobytes() == piece, size
    # the alternating writer of alevin-fry_amd/rad.py (what the [compressed] CLI tests feed the host decoder)
    assert af.snappy_decode_device(rad.snappy_frame_encode(inputs[0])).tobytes() == inputs[0]


@pytest.mark.gpu
def test_device_decoder_refuses_malformed_blocks():
    good = handmade_blocks()
    bad_blocks = [
        _uvarint(10) + lit(b"abcd") + cp1(8, 4),          # decodes to 12 bytes, announces 10
        _uvarint(12) + lit(b"abcd") + cp1(8, 5),          # offset beyond what has been written
        _uvarint(12) + lit(b"abcd") + cp2(8, 0),          # offset 0
        _uvarint(300) + bytes([61 << 2, 0x2B]),           # literal length cut off
        _uvarint(40) + lit(b"abcd"),                      # ends early
        _uvarint(8) + bytes([7 << 2]) + b"abcd",          # literal longer than the block
    ]
    for k, blk in enumerate(bad_blocks):
        chunks = good[:2] + [("blk", blk, b"?" * 4)] + good[2:3]
        stream = frame_stream(chunks)
        with pytest.raises(af.AfqError) as e:
            af.snappy_decode_device(stream)
        assert e.value.code == -2 and "chunk 2" in str(e.value), (k, str(e.value))
