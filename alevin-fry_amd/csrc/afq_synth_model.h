// afq_synth_model.h — the record model of the synthetic collated-RAD generator, shared by its host and device
// implementations (afq_synth.hip).  Bench / test tooling: nothing on the quant path includes this.
//
// Counter-based and all-integer: every draw is a word of a Philox4x32-10 block (Salmon et al., SC'11) keyed by the
// seed and counted by (read or molecule index, cell index, stream), and every probability is a 32-bit threshold, so
// the host and the gfx950 kernels produce the same bytes for the same parameters, cells can be generated in any order
// and on any rank, and the data set does not depend on how it is sharded.
#pragma once
#include <stdint.h>

namespace afq_synth {

struct Model {
    uint32_t k0, k1;           // Philox key (seed)
    uint32_t num_genes;        // G
    uint32_t tpg, tx_big;      // genes [0, tx_big) have tpg+1 spliced transcripts, the others tpg
    uint32_t n_spliced;        // G*tpg + tx_big
    uint32_t usa;
    uint32_t umi_len, umi_mask;
    uint32_t mol_q;            // (1 - dup) * 2^32: molecules per read of a cell
    uint32_t thr_na3, thr_na23;            // P(na = 3), P(na >= 2)
    uint32_t thr_cross, thr_umi_err;       // P(an extra ref sits on another gene), P(1-base UMI error)
    uint32_t thr_unspl, thr_unspl_both;    // USA: P(unspliced), P(unspliced or spliced+unspliced)
    uint32_t bc_salt;
    // label-length tail (na_model "tail"): after the one to three refs above, further refs - the same for every read of a
    // molecule - are added while a draw stays under
    // thr_tail (a geometric run, at most tail_max refs in all), each on a gene of the read's gene FAMILY - the block of
    // `family` consecutive gene ids its gene sits in - so that labels of 5..30 refs over more than four genes occur, as they
    // do against a transcriptome with paralogues; thr_tail = 0: the plain model, byte for byte what it was
    uint32_t thr_tail, tail_max, family;
    const uint32_t* alias_thr;  // [G] Walker alias table of the gene popularity: threshold / other gene
    const uint32_t* alias_idx;
};

__host__ __device__ inline void philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t (&o)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

// uniform integer below n from one 32-bit word
__host__ __device__ inline uint32_t below(uint32_t w, uint32_t n) { return (uint32_t)(((uint64_t)w * n) >> 32); }

__host__ __device__ inline uint32_t first_txp(const Model& m, uint32_t g) {
    return g < m.tx_big ? g * (m.tpg + 1) : m.tx_big * (m.tpg + 1) + (g - m.tx_big) * m.tpg;
}
__host__ __device__ inline uint32_t num_txp(const Model& m, uint32_t g) { return m.tpg + (g < m.tx_big ? 1u : 0u); }

// molecules of a cell with nrec reads
__host__ __device__ inline uint32_t num_molecules(const Model& m, uint32_t nrec) {
    const uint32_t n = (uint32_t)(((uint64_t)nrec * m.mol_q + 0x80000000ull) >> 32);
    return n ? n : 1u;
}

// 32-bit barcode of a cell: a bijection of the cell index, so barcodes are distinct (cells < 2^32)
__host__ __device__ inline uint32_t barcode(const Model& m, uint64_t cell) {
    uint32_t x = (uint32_t)cell + m.bc_salt;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}

constexpr uint32_t kMaxRefs = 64;   // refs of one record at most (tail model)

// One record: refs[0..n) ascending and distinct (n <= 3, or <= tail_max under the tail model), its UMI.  Returns n.
// FULL = false computes only what the record's size depends on (no UMI).
template <bool FULL>
__host__ __device__ inline uint32_t record(const Model& m, uint64_t cell, uint32_t read, uint32_t n_mol, uint32_t (&refs)[kMaxRefs], uint32_t& umi) {
    const uint32_t cl = (uint32_t)cell, ch = (uint32_t)(cell >> 32);
    uint32_t w[4], q[4];
    philox(read, cl, ch, 0u, m.k0, m.k1, w);   // w0 molecule, w1 na, w2 transcript, w3 splicing state
    const uint32_t mol = below(w[0], n_mol);
    philox(mol, cl, ch, 0x80000000u, m.k0, m.k1, q);   // q0 UMI, q1/q2 gene (alias draw)
    const uint32_t gi = below(q[1], m.num_genes);
    const uint32_t gene = q[2] < m.alias_thr[gi] ? gi : m.alias_idx[gi];
    uint32_t na = w[1] < m.thr_na3 ? 3u : (w[1] < m.thr_na23 ? 2u : 1u);
    uint32_t n = 0;
    bool both = false;
    refs[0] = refs[1] = refs[2] = 0;
    const uint32_t spl = first_txp(m, gene) + below(w[2], num_txp(m, gene));
    if (m.usa) {
        if (w[3] < m.thr_unspl) refs[n++] = m.n_spliced + gene;
        else {
            refs[n++] = spl;
            if (w[3] < m.thr_unspl_both) { both = true; refs[n++] = m.n_spliced + gene; if (na < 2) na = 2; }
        }
    } else refs[n++] = spl;
    for (uint32_t k = both ? 2u : 1u; k < na; ++k) {
        uint32_t e[4];
        philox(read, cl, ch, 1u + k, m.k0, m.k1, e);   // e0 other gene?, e1 which, e2 transcript, e3 unspliced?
        const uint32_t g = e[0] < m.thr_cross ? below(e[1], m.num_genes) : gene;
        uint32_t t = first_txp(m, g) + below(e[2], num_txp(m, g));
        if (m.usa && e[3] < m.thr_unspl) t = m.n_spliced + g;
        refs[n++] = t;
    }
    // ascending, distinct (n <= 3)
    if (n > 1 && refs[0] > refs[1]) { const uint32_t t = refs[0]; refs[0] = refs[1]; refs[1] = t; }
    if (n > 2) {
        if (refs[1] > refs[2]) { const uint32_t t = refs[1]; refs[1] = refs[2]; refs[2] = t; }
        if (refs[0] > refs[1]) { const uint32_t t = refs[0]; refs[0] = refs[1]; refs[1] = t; }
        if (refs[1] == refs[2]) n = 2;
    }
    if (n > 1 && refs[0] == refs[1]) { refs[1] = refs[2]; --n; }
    if (m.thr_tail) {   // the tail: more refs on the gene's family, kept ascending and distinct by insertion
        const uint32_t fam0 = (gene / m.family) * m.family;
        const uint32_t fam_n = fam0 + m.family <= m.num_genes ? m.family : m.num_genes - fam0;
        for (uint32_t k = 0; n < m.tail_max; ++k) {
            uint32_t e[4];
            philox(mol, cl, ch, 0x80000100u + k, m.k0, m.k1, e);   // e0 one more?, e1 which gene of the family, e2 transcript, e3 unspliced?
            // (drawn per MOLECULE: the reads of a molecule come off the same end of the same transcript and map to the same
            //  set of paralogues; per-read draws gave every read of a molecule a label of its own)
            if (e[0] >= m.thr_tail) break;
            const uint32_t g = fam0 + below(e[1], fam_n);
            uint32_t t = first_txp(m, g) + below(e[2], num_txp(m, g));
            if (m.usa && e[3] < m.thr_unspl) t = m.n_spliced + g;
            uint32_t at = n;
            while (at > 0 && refs[at - 1] > t) --at;
            if (at > 0 && refs[at - 1] == t) continue;
            for (uint32_t q = n; q > at; --q) refs[q] = refs[q - 1];
            refs[at] = t;
            ++n;
        }
    }
    if (FULL) {
        uint32_t u = q[0] & m.umi_mask;
        uint32_t x[4];
        philox(read, cl, ch, 1u, m.k0, m.k1, x);   // x0 UMI error?, x1 position, x2 substitution
        if (x[0] < m.thr_umi_err) {
            const uint32_t pos = below(x[1], m.umi_len), delta = 1u + below(x[2], 3u);
            const uint32_t b = (u >> (2 * pos)) & 3u;
            u = (u & ~(3u << (2 * pos))) | (((b + delta) & 3u) << (2 * pos));
        }
        umi = u;
    }
    return n;
}

}  // namespace afq_synth
