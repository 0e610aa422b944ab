"""Seeded synthetic collated-RAD generators (numpy) for tests and small benches.

Distributions follow SURVEY.md §8(d): distinct splitmix barcodes, 12-mer UMIs with
forced duplicates (reads drawn from a per-cell pool of molecules), na in {1,2,3}
with p = 0.7/0.2/0.1, half of the multi-ref reads crossing genes, a small rate of
1-mismatch UMI errors, refs sorted ascending and duplicate-free (the mapper
guarantee the reference relies on, src/pugutils.rs:375, 1183).
The full-size bench input is produced by the C++ generator in csrc/afq_synth.cpp
with the same model.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import rad


def splitmix64(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64)
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


@dataclass
class SynthRad:
    cell_nrec: np.ndarray  # [n_cells] i64
    cell_bc: np.ndarray  # [n_cells] u64
    umi: np.ndarray  # [n_reads] u64
    na: np.ndarray  # [n_reads] i64
    refs: np.ndarray  # [sum na] u32, ascending within a read
    tid_to_gid: np.ndarray  # [ref_count] u32
    num_genes: int  # gene-id space of tid_to_gid (USA: 2*G)
    num_rows: int  # output columns (USA: 3*G)
    usa: bool
    umi_len: int

    def encode(self):
        return rad.encode_cells_np(self.cell_nrec, self.cell_bc, self.umi, self.na, self.refs)

    @property
    def ref_count(self) -> int:
        return len(self.tid_to_gid)


def make_t2g(num_genes: int, txp_per_gene: int, usa: bool):
    """Non-USA: gene g owns tids [g*tpg, (g+1)*tpg).  USA (splici-like): spliced txps
    first (gid 2g), then one unspliced/intron txp per gene (gid 2g+1) — ids as
    assigned by parse_tg_map for a 3-column map (src/utils.rs:506-539)."""
    if not usa:
        t2g = (np.arange(num_genes * txp_per_gene) // txp_per_gene).astype(np.uint32)
        return t2g, num_genes, num_genes
    s = np.arange(num_genes * txp_per_gene) // txp_per_gene
    t2g = np.concatenate((2 * s, 2 * np.arange(num_genes) + 1)).astype(np.uint32)
    return t2g, 2 * num_genes, 3 * num_genes


def synth(seed: int, cell_nrec, num_genes: int = 1000, txp_per_gene: int = 2, umi_len: int = 12,
          dup: float = 0.3, p_na=(0.7, 0.2, 0.1), cross: float = 0.5, umi_err: float = 0.01,
          usa: bool = False, p_unspliced: float = 0.35, p_both: float = 0.08, zipf: float = 0.0,
          max_extra_na: int = 0) -> SynthRad:
    rng = np.random.default_rng(seed)
    cell_nrec = np.asarray(cell_nrec, dtype=np.int64)
    n_cells = len(cell_nrec)
    n_reads = int(cell_nrec.sum())
    t2g, gid_space, num_rows = make_t2g(num_genes, txp_per_gene, usa)
    n_spliced_txp = num_genes * txp_per_gene
    cell_bc = splitmix64(np.arange(n_cells, dtype=np.uint64) + np.uint64(seed) * np.uint64(1000003)) & np.uint64(0xFFFFFFFF)
    # force distinct barcodes
    _, first = np.unique(cell_bc, return_index=True)
    if len(first) != n_cells:
        cell_bc = (cell_bc + np.arange(n_cells, dtype=np.uint64) * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)
    cell_of_read = np.repeat(np.arange(n_cells), cell_nrec)
    # molecule pool per cell: a read picks molecule m in [0, n_mol(cell))
    n_mol = np.maximum(1, np.round(cell_nrec * (1.0 - dup)).astype(np.int64))
    mol = (rng.random(n_reads) * n_mol[cell_of_read]).astype(np.int64)
    mol_key = splitmix64((cell_of_read.astype(np.uint64) << np.uint64(32)) ^ mol.astype(np.uint64) ^ (np.uint64(seed) << np.uint64(50)))
    umi_mask = np.uint64((1 << (2 * umi_len)) - 1)
    umi = mol_key & umi_mask
    # gene of the molecule (optionally Zipf-skewed popularity)
    gsel = splitmix64(mol_key ^ np.uint64(0xABCDEF))
    if zipf > 0:
        u = (gsel >> np.uint64(11)).astype(np.float64) / float(1 << 53)
        gene = np.minimum(num_genes - 1, np.floor(num_genes * u ** (1.0 + zipf * 3))).astype(np.int64)
    else:
        gene = (gsel % np.uint64(num_genes)).astype(np.int64)
    # per-read alignment count
    r = rng.random(n_reads)
    na = np.where(r < p_na[0], 1, np.where(r < p_na[0] + p_na[1], 2, 3)).astype(np.int64)
    if max_extra_na > 0:  # a few heavily multi-mapping reads
        heavy = rng.random(n_reads) < 0.002
        na = np.where(heavy, rng.integers(4, 4 + max_extra_na, n_reads), na)
    max_na = int(na.max()) if n_reads else 1
    big = np.uint32(0xFFFFFFFF)
    cand = np.full((n_reads, max_na), big, dtype=np.uint32)
    # first ref: a txp of the molecule's gene
    j0 = rng.integers(0, txp_per_gene, n_reads)
    if usa:
        status = rng.random(n_reads)  # per read: U / both / S
        is_u = status < p_unspliced
        is_both = (status >= p_unspliced) & (status < p_unspliced + p_both)
        first = np.where(is_u, n_spliced_txp + gene, gene * txp_per_gene + j0)
    else:
        is_both = np.zeros(n_reads, dtype=bool)
        first = gene * txp_per_gene + j0
    cand[:, 0] = first.astype(np.uint32)
    for k in range(1, max_na):
        has = na > k
        crosses = rng.random(n_reads) < cross
        other_gene = rng.integers(0, num_genes, n_reads)
        g_k = np.where(crosses, other_gene, gene)
        t_k = g_k * txp_per_gene + rng.integers(0, txp_per_gene, n_reads)
        if usa:
            to_u = rng.random(n_reads) < p_unspliced
            t_k = np.where(to_u, n_spliced_txp + g_k, t_k)
            if k == 1:
                t_k = np.where(is_both, n_spliced_txp + gene, t_k)
        cand[:, k] = np.where(has, t_k, big).astype(np.uint32)
    if usa:  # reads flagged "both" always carry S and U of their gene
        need = is_both & (na < 2)
        na = np.where(need, 2, na)
        if max_na < 2:
            cand = np.concatenate((cand, np.full((n_reads, 1), big, np.uint32)), axis=1)
            max_na = 2
        cand[:, 1] = np.where(is_both, (n_spliced_txp + gene).astype(np.uint32), cand[:, 1])
    cand.sort(axis=1)
    # drop duplicates within a read
    dupm = np.zeros_like(cand, dtype=bool)
    dupm[:, 1:] = cand[:, 1:] == cand[:, :-1]
    keep = (cand != big) & ~dupm
    na = keep.sum(axis=1).astype(np.int64)
    refs = cand[keep]
    # UMI sequencing errors: flip one base
    err = rng.random(n_reads) < umi_err
    pos = rng.integers(0, umi_len, n_reads).astype(np.uint64)
    delta = rng.integers(1, 4, n_reads).astype(np.uint64)
    base = (umi >> (np.uint64(2) * pos)) & np.uint64(3)
    newb = (base + delta) & np.uint64(3)
    umi_e = (umi & ~(np.uint64(3) << (np.uint64(2) * pos))) | (newb << (np.uint64(2) * pos))
    umi = np.where(err, umi_e, umi)
    return SynthRad(cell_nrec, cell_bc, umi, na, refs.astype(np.uint32), t2g, gid_space, num_rows, usa, umi_len)


def lognormal_cell_sizes(seed: int, n_cells: int, median: float, sigma: float, min_reads: int = 1):
    rng = np.random.default_rng(seed)
    v = np.maximum(min_reads, np.round(np.exp(rng.normal(np.log(median), sigma, n_cells)))).astype(np.int64)
    return np.sort(v)[::-1].copy()  # collate orders large cells first (src/collate.rs:272-274)
