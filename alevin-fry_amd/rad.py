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
