"""Collated-RAD chunk codec (host side, numpy).

Wire format as witnessed in the reference (paths relative to /root/reference):
  chunk   = nbytes:u32 (includes this 8-byte header), nrec:u32, records   src/convert.rs:473-481
  record  = na:u32, bc:<u8|u16|u32|u64>, umi:<u8|u16|u32|u64>, na x u32    src/convert.rs:124-144
            alignment word = orientation<<31 | ref_id                      src/convert.rs:443-445
  widths  = by sequence length 1-4 / 5-8 / 9-16 / 17-32 nt                 src/convert.rs:323-344
  2-bit   = A0 C1 G2 T3, first base most significant                       src/convert.rs:75-90
In a collated file one chunk holds all records of one corrected cell barcode.
"""
from __future__ import annotations

import numpy as np

_NT = "ACGT"


def width_for_len(n_nt: int) -> int:
    """Bytes of the integer type the reference picks for an n_nt-long barcode/UMI."""
    if 1 <= n_nt <= 4:
        return 1
    if n_nt <= 8:
        return 2
    if n_nt <= 16:
        return 4
    if n_nt <= 32:
        return 8
    raise ValueError("cannot encode a sequence longer than 32 nt")


def seq_to_int(s: str) -> int:
    v = 0
    for ch in s:
        v = (v << 2) | (0 if ch == "N" else _NT.index(ch))
    return v


def int_to_seq(v: int, n_nt: int) -> str:
    return "".join(_NT[(v >> (2 * (n_nt - 1 - i))) & 3] for i in range(n_nt))


def encode_cells(cells, bc_bytes: int = 4, umi_bytes: int = 4, fw_bit: bool = True):
    """cells: list of (bc, [(umi, [ref, ...]), ...]).  Returns (bytes, chunk_off[u64]).

    Generic (any field width) pure-python encoder for small hand-written cases.
    """
    out = bytearray()
    offs = []
    for bc, reads in cells:
        offs.append(len(out))
        body = bytearray()
        for umi, refs in reads:
            body += int(len(refs)).to_bytes(4, "little")
            body += int(bc).to_bytes(bc_bytes, "little")
            body += int(umi).to_bytes(umi_bytes, "little")
            for r in refs:
                w = int(r) | (0x80000000 if fw_bit else 0)
                body += w.to_bytes(4, "little")
        out += (len(body) + 8).to_bytes(4, "little") + len(reads).to_bytes(4, "little") + body
    return bytes(out), np.asarray(offs, dtype=np.uint64)


def encode_cells_np(cell_nrec, cell_bc, umi, na, refs, fw_bits=None):
    """Vectorised encoder for the 10x-v3 layout (u32 barcode, u32 UMI).

    cell_nrec[n_cells], cell_bc[n_cells]; per read umi[], na[]; refs[] = concatenated,
    sorted-ascending ref ids.  Returns (uint8 array, chunk_off[u64]).
    """
    cell_nrec = np.asarray(cell_nrec, dtype=np.int64)
    na = np.asarray(na, dtype=np.int64)
    n_reads = int(cell_nrec.sum())
    assert n_reads == len(na) == len(umi)
    n_cells = len(cell_nrec)
    cell_of_read = np.repeat(np.arange(n_cells), cell_nrec)
    rec_words = 3 + na
    # word offset of each record: preceding record words + 2 header words per started cell
    rec_off = np.concatenate(([0], np.cumsum(rec_words)[:-1])) + 2 * (cell_of_read + 1)
    total_words = int(rec_words.sum()) + 2 * n_cells
    w = np.zeros(total_words, dtype=np.uint32)
    # chunk headers
    first_read = np.concatenate(([0], np.cumsum(cell_nrec)[:-1]))
    cell_words = np.add.reduceat(rec_words, first_read[cell_nrec > 0]) if n_reads else np.zeros(0, np.int64)
    cw = np.zeros(n_cells, dtype=np.int64)
    cw[cell_nrec > 0] = cell_words
    chunk_word_off = np.concatenate(([0], np.cumsum(cw + 2)[:-1]))
    w[chunk_word_off] = ((cw + 2) * 4).astype(np.uint32)
    w[chunk_word_off + 1] = cell_nrec.astype(np.uint32)
    w[rec_off] = na.astype(np.uint32)
    w[rec_off + 1] = np.asarray(cell_bc, dtype=np.uint64)[cell_of_read].astype(np.uint32)
    w[rec_off + 2] = np.asarray(umi, dtype=np.uint64).astype(np.uint32)
    ref_start = np.concatenate(([0], np.cumsum(na)[:-1]))
    # position of each ref word
    read_of_ref = np.repeat(np.arange(n_reads), na)
    within = np.arange(len(refs)) - ref_start[read_of_ref]
    rw = np.asarray(refs, dtype=np.uint32).copy()
    if fw_bits is None:
        rw |= np.uint32(0x80000000)
    else:
        rw |= np.asarray(fw_bits, dtype=np.uint32) << np.uint32(31)
    w[rec_off[read_of_ref] + 3 + within] = rw
    return w.view(np.uint8), (chunk_word_off * 4).astype(np.uint64)


def decode_chunk(buf: bytes, off: int, bc_bytes: int, umi_bytes: int):
    """Returns (bc, [(umi, [ref...])...]) for the chunk starting at off."""
    nbytes = int.from_bytes(buf[off : off + 4], "little")
    nrec = int.from_bytes(buf[off + 4 : off + 8], "little")
    p = off + 8
    reads = []
    bc0 = None
    for _ in range(nrec):
        na = int.from_bytes(buf[p : p + 4], "little")
        bc = int.from_bytes(buf[p + 4 : p + 4 + bc_bytes], "little")
        umi = int.from_bytes(buf[p + 4 + bc_bytes : p + 4 + bc_bytes + umi_bytes], "little")
        p += 4 + bc_bytes + umi_bytes
        refs = [int.from_bytes(buf[p + 4 * j : p + 4 * j + 4], "little") & 0x7FFFFFFF for j in range(na)]
        p += 4 * na
        if bc0 is None:
            bc0 = bc
        reads.append((umi, refs))
    assert p == off + nbytes, "chunk nbytes does not match its records"
    return bc0, reads


def chunk_offsets(buf, start: int = 0):
    """Hop the nbytes headers of back-to-back chunks (what the quant producer does)."""
    b = np.frombuffer(buf, dtype=np.uint8)
    offs = []
    p = start
    while p + 8 <= len(b):
        nbytes = int(b[p : p + 4].view(np.uint32)[0])
        if nbytes < 8:
            raise ValueError("corrupt chunk header")
        offs.append(p)
        p += nbytes
    if p != len(b):
        raise ValueError("trailing bytes after last chunk")
    return np.asarray(offs, dtype=np.uint64)


# ---------------------------------------------------------------------------
# whole-file writer (test fixtures / synthetic inputs for the host front-end)
_INT_TYPE_ID = {1: 1, 2: 2, 4: 3, 8: 4}  # bytes -> RadType id (U8..U64)


def rad_prelude(ref_names, num_chunks, cblen, ulen, bc_bytes=4, umi_bytes=4, is_paired=False) -> bytes:
    """Header + the three tag sections + file-tag values of a single-barcode scRNA RAD file
    (order of writes in src/convert.rs:254-369)."""
    out = bytearray()
    out += bytes([1 if is_paired else 0])
    out += len(ref_names).to_bytes(8, "little")
    for n in ref_names:
        b = n.encode()
        out += len(b).to_bytes(2, "little") + b
    out += int(num_chunks).to_bytes(8, "little")

    def tag(name, type_id):
        b = name.encode()
        return len(b).to_bytes(2, "little") + b + bytes([type_id])

    out += (2).to_bytes(2, "little") + tag("cblen", 2) + tag("ulen", 2)  # file tags (u16, u16)
    out += (2).to_bytes(2, "little") + tag("b", _INT_TYPE_ID[bc_bytes]) + tag("u", _INT_TYPE_ID[umi_bytes])  # read tags
    out += (1).to_bytes(2, "little") + tag("compressed_ori_refid", 3)  # alignment tags
    out += int(cblen).to_bytes(2, "little") + int(ulen).to_bytes(2, "little")  # file tag values
    return bytes(out)


def rad_prelude_multi_bc(ref_names, num_chunks, b0len, b1len, ulen, b_bytes=4, umi_bytes=4, b0_bytes=None) -> bytes:
    """Prelude of a two-level multi-barcode (10x Flex) RAD file as the reference's tests build it
    (tests/multi_barcode_integration.rs:58-130): file tags num_barcodes, b0len, b1len, ulen (u16) and known_rad_type
    (string); read tags b0, b1, u; one u32 alignment tag."""
    out = bytearray()
    out += bytes([0])
    out += len(ref_names).to_bytes(8, "little")
    for n in ref_names:
        b = n.encode()
        out += len(b).to_bytes(2, "little") + b
    out += int(num_chunks).to_bytes(8, "little")

    def tag(name, type_id):
        b = name.encode()
        return len(b).to_bytes(2, "little") + b + bytes([type_id])

    out += (5).to_bytes(2, "little") + tag("num_barcodes", 2) + tag("b0len", 2) + tag("b1len", 2) + tag("ulen", 2) + tag("known_rad_type", 8)
    out += (3).to_bytes(2, "little") + tag("b0", _INT_TYPE_ID[b0_bytes or b_bytes]) + tag("b1", _INT_TYPE_ID[b_bytes]) + tag("u", _INT_TYPE_ID[umi_bytes])
    out += (1).to_bytes(2, "little") + tag("compressed_ori_refid", 3)
    kind = b"sc_rna_multi_bc"
    out += (2).to_bytes(2, "little") + int(b0len).to_bytes(2, "little") + int(b1len).to_bytes(2, "little") + int(ulen).to_bytes(2, "little")
    out += len(kind).to_bytes(2, "little") + kind
    return bytes(out)


def rad_prelude_atac(ref_names, ref_lengths, num_chunks, cblen=16, bc_bytes=4) -> bytes:
    """Prelude of a scATAC RAD file as piscem writes it and the reference's tests build it (tests/atac_integration.rs:75-131):
    paired flag set; file tags cblen (u16), known_rad_type (string), ref_lengths (array of u32, u32 length); read tag b;
    alignment tags ref:u32, type:u8, start_pos:u32, frag_len:u16."""
    out = bytearray()
    out += bytes([1])
    out += len(ref_names).to_bytes(8, "little")
    for n in ref_names:
        b = n.encode()
        out += len(b).to_bytes(2, "little") + b
    out += int(num_chunks).to_bytes(8, "little")

    def tag(name, type_id, extra=b""):
        b = name.encode()
        return len(b).to_bytes(2, "little") + b + bytes([type_id]) + extra

    out += (3).to_bytes(2, "little") + tag("cblen", 2) + tag("known_rad_type", 8) + tag("ref_lengths", 7, bytes([3, 3]))
    out += (1).to_bytes(2, "little") + tag("b", _INT_TYPE_ID[bc_bytes])
    out += (4).to_bytes(2, "little") + tag("ref", 3) + tag("type", 1) + tag("start_pos", 3) + tag("frag_len", 2)
    kind = b"sc_atac"
    out += int(cblen).to_bytes(2, "little") + len(kind).to_bytes(2, "little") + kind
    out += len(ref_lengths).to_bytes(4, "little") + b"".join(int(x).to_bytes(4, "little") for x in ref_lengths)
    return bytes(out)


def encode_atac_cells(cells, bc_bytes: int = 4):
    """cells: list of (bc, [[(ref, type, start, frag_len), ...] per record]).  Returns (bytes, chunk_off[u64])."""
    out = bytearray()
    offs = []
    for bc, recs in cells:
        offs.append(len(out))
        body = bytearray()
        for alns in recs:
            body += len(alns).to_bytes(4, "little") + int(bc).to_bytes(bc_bytes, "little")
            for ref, ty, start, fl in alns:
                body += int(ref).to_bytes(4, "little") + bytes([ty]) + int(start).to_bytes(4, "little") + int(fl).to_bytes(2, "little")
        out += (len(body) + 8).to_bytes(4, "little") + len(recs).to_bytes(4, "little") + body
    return bytes(out), np.asarray(offs, dtype=np.uint64)


def collation_manifest(groups, level_names=("sample", "cell")) -> bytes:
    """collation_manifest.bin in the layout csrc/afq_host.cpp reads (bincode of libradicl's CollationManifest - a
    restatement, libradicl's source is not in the reference tree).  groups: (key, name or None, chunk_start, num_chunks,
    num_records)."""
    out = bytearray()
    out += len(level_names).to_bytes(8, "little")
    for n in level_names:
        b = n.encode()
        out += len(b).to_bytes(8, "little") + b
    out += len(groups).to_bytes(8, "little")
    for key, name, start, nch, nrec in groups:
        out += int(key).to_bytes(8, "little")
        if name is None:
            out += b"\x00"
        else:
            b = name.encode()
            out += b"\x01" + len(b).to_bytes(8, "little") + b
        out += int(start).to_bytes(8, "little") + int(nch).to_bytes(8, "little") + int(nrec).to_bytes(8, "little")
    return bytes(out)


def _crc32c(data: bytes) -> int:
    tbl = getattr(_crc32c, "_t", None)
    if tbl is None:
        tbl = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            tbl.append(c)
        _crc32c._t = tbl
    c = 0xFFFFFFFF
    for b in data:
        c = tbl[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def snappy_frame_encode(data: bytes, chunk: int = 60000, compress_literal: bool = True, real: bool = True) -> bytes:
    """Snappy *frame format* writer: stream identifier + one chunk per `chunk` bytes, alternating uncompressed
    chunks (type 0x01) and compressed chunks (type 0x00).  With `real` (and pyarrow importable) the compressed
    blocks come from Google's snappy through pyarrow - back-references and all, i.e. what
    `snap::write::FrameEncoder` (src/collate.rs:550-554) puts in a real file; otherwise they are literal-only blocks."""
    codec = None
    if real:
        try:
            import pyarrow as pa

            codec = pa.Codec("snappy") if pa.Codec.is_available("snappy") else None
        except Exception:  # noqa
            codec = None
    out = bytearray(b"\xff\x06\x00\x00sNaPpY")
    for k, i in enumerate(range(0, len(data), chunk)):
        piece = data[i : i + chunk]
        crc = _crc32c(piece)
        m = (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF
        if codec is not None and compress_literal and k % 2 == 1:
            body = m.to_bytes(4, "little") + codec.compress(bytes(piece), asbytes=True)
            out += b"\x00" + len(body).to_bytes(3, "little") + body
        elif compress_literal and k % 2 == 1:
            n = len(piece)
            blk = bytearray()
            v = n
            while True:  # uvarint of the uncompressed length
                if v < 0x80:
                    blk.append(v)
                    break
                blk.append((v & 0x7F) | 0x80)
                v >>= 7
            ln = n - 1
            if ln < 60:
                blk.append(ln << 2)
            elif ln < 256:
                blk += bytes([60 << 2, ln])
            elif ln < 65536:
                blk += bytes([61 << 2]) + ln.to_bytes(2, "little")
            else:
                blk += bytes([62 << 2]) + ln.to_bytes(3, "little")
            blk += piece
            body = m.to_bytes(4, "little") + bytes(blk)
            out += b"\x00" + len(body).to_bytes(3, "little") + body
        else:
            body = m.to_bytes(4, "little") + piece
            out += b"\x01" + len(body).to_bytes(3, "little") + body
    return bytes(out)


def write_quant_input_dir(path, chunk_bytes, n_chunks, ref_names, t2g_rows, cblen=16, ulen=12, bc_bytes=4, umi_bytes=4,
                          compressed=False, prelude=None):
    """Lay out what `alevin-fry quant -i` expects: generate_permit_list.json, collate.json,
    map.collated.rad[.sz], plus the tg-map next to it.  t2g_rows: list of tab-separated row tuples."""
    import json
    import os

    os.makedirs(path, exist_ok=True)
    if prelude is None:
        prelude = rad_prelude(ref_names, n_chunks, cblen, ulen, bc_bytes, umi_bytes)
    with open(os.path.join(path, "generate_permit_list.json"), "w") as f:
        json.dump({"velo_mode": False, "expected_ori": "fw"}, f)
    with open(os.path.join(path, "collate.json"), "w") as f:
        json.dump({"cmd": "synthetic", "version_str": "0.18.0", "compressed_output": bool(compressed)}, f)
    if compressed:
        with open(os.path.join(path, "map.collated.rad.sz"), "wb") as f:
            f.write(snappy_frame_encode(prelude + bytes(chunk_bytes)))
    else:
        with open(os.path.join(path, "map.collated.rad"), "wb") as f:   # (no concatenated copy: inputs run to gigabytes)
            f.write(prelude)
            f.write(memoryview(chunk_bytes))
    tg = os.path.join(path, "t2g.tsv")
    with open(tg, "w") as f:
        for row in t2g_rows:
            f.write("\t".join(row) + "\n")
    return tg
