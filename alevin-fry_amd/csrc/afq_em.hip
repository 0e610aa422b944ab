// afq_em.hip - per-cell EM over gene-level equivalence classes on gfx950 (src/em.rs of the reference):
//   k_em          setup: classes, support, inverted index, compact active set
//   k_em_rounds   the rounds, with the round state on chip
//   k_compact_em  EM output pairs -> CSR
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "afq_common.h"
#include "afq_kernels.h"
#include "afq_prims.h"

namespace afq {
// ---------------------------------------------------------------------------
// Per-cell EM over the gene-level equivalence classes (src/em.rs).  One workgroup per cell.
//   inputs : the cell's single-label counts = sorted (column,count) pairs (k_cell_hist / k_resolve),
//            and the labels of its ambiguous molecules (label area, written by resolve)
//   steps  : group identical labels into classes (lexicographic order), rewrite USA labels to S/U/A
//            slots (utils.rs:865-925), build the support (labels, and in USA their sibling statuses,
//            em.rs:87-113), then iterate.  One round = (A) thread per class: denominator in label order;
//            (B) thread per support entry: the single-label count first, then the contributions of the
//            classes containing it in class order - the same f32 operation sequence as the sequential
//            loop of em_update (em.rs:189-248, 458-485) under the oracle's canonical class order, so
//            results are bit-identical to the oracle; (C) convergence vote.
//   schedule: non-USA = em_optimize (em.rs:536-572); USA = em_optimize_subset_impl with the extra
//            round after zeroing < 0.01 (em.rs:391-451).
// All arrays live in a per-cell global scratch slice (L2 resident); sizes are tiny next to the decode.
struct EmCfg {
    uint32_t usa, num_alphas, uo, ao, init_uniform;
};
constexpr int kEmNT = 512;   // threads per cell in k_em (256 and 1024 both measured slower on configs[2]: 178 / 181 vs 177 ms per step)
constexpr float kMinOutputAlpha = 0.01f, kAlphaCheckCutoff = 1e-2f, kRelDiffTol = 1e-2f;
constexpr uint32_t kMinIter = 2, kMaxIter = 100;


// Per-cell EM scratch (u32 words; mirrored by em_scratch_words).  Filled by k_em (setup), consumed by k_em_rounds.
struct EmScratch {
    uint2* out; uint64_t* inv_pairs; uint32_t *order, *cls_first, *cls_cnt, *cls_woff, *cls_w, *cls_sidx; float* inv;
    uint32_t *support, *sib1, *sib2, *ucnt; float *a_in, *a_out; uint32_t *slot_off, *aid; uint4 *ent, *lw3;
    uint32_t *act_col, *memb;
};
__device__ __forceinline__ EmScratch em_carve(uint32_t* scratch, uint64_t off, uint32_t nU, uint32_t W, uint32_t M, uint32_t capS) {
    EmScratch e;
    uint32_t* p = scratch + off;
    e.out = reinterpret_cast<uint2*>(p); p += 2 * (capS + 1);           // (column, f32 bits); 8-byte aligned by construction
    e.inv_pairs = reinterpret_cast<uint64_t*>(p); p += 2 * (W + 1);     // (support idx << 32 | class)
    e.order = p; p += M + 1;       // molecule indices sorted by label
    e.cls_first = p; p += M + 1;   // class -> position in `order` of its first molecule
    e.cls_cnt = p; p += M + 1;
    e.cls_woff = p; p += M + 2;    // class -> offset of its EM label in cls_w
    e.cls_w = p; p += W + 1;       // EM labels (slots)
    e.cls_sidx = p; p += W + 1;    // ... as support indices
    e.inv = reinterpret_cast<float*>(p); p += M + 1;
    e.support = p; p += capS + 1;
    e.sib1 = p; p += capS + 1;
    e.sib2 = p; p += capS + 1;
    e.ucnt = p; p += capS + 1;
    e.a_in = reinterpret_cast<float*>(p); p += capS + 2;
    e.a_out = reinterpret_cast<float*>(p); p += capS + 2;
    e.slot_off = p; p += capS + 2;
    e.aid = p; p += capS + 1;          // support idx -> active idx
    p += (4 - ((p - scratch) & 3)) & 3;        // 16-byte records below (slices start 16-byte aligned)
    e.ent = reinterpret_cast<uint4*>(p); p += 4 * (nU + W + 2);   // per active entry: count, sibling ids, first membership
    e.lw3 = reinterpret_cast<uint4*>(p); p += 4 * (W + 1);        // per label word: its entry and the entry's siblings
    e.act_col = p; p += nU + W + 2;
    e.memb = p; p += W + 1;            // class ids of the memberships, entry-major
    return e;
}

__global__ __launch_bounds__(kEmNT) void k_em(const CellMeta* __restrict__ meta, const uint32_t* __restrict__ nnz_unique,
                                             const uint64_t* __restrict__ keys0, const uint64_t* __restrict__ keys1,
                                             const uint32_t* __restrict__ lab, const uint32_t* __restrict__ lab_cnt,
                                             const uint64_t* __restrict__ em_off, uint32_t* __restrict__ scratch,
                                             uint32_t* __restrict__ out_nnz, uint4* __restrict__ em_hdr, const uint32_t* __restrict__ em_order, EmCfg cfg) {
    __shared__ uint32_t s_ws[kEmNT / 64];
    __shared__ __attribute__((aligned(16))) uint32_t s_tile[8192];  // 32 KiB sort tile
    const uint32_t cell = em_order[blockIdx.x];  // largest cells first (the host sorts: input order is arbitrary in real data)
#ifdef AFQ_EM_TIMING
    __shared__ unsigned long long tmark[12];
#define EM_MARK(i) do { __syncthreads(); if (threadIdx.x == 0 && (blockIdx.x % 1000) == 7) tmark[i] = wall_clock64(); } while (0)
#else
#define EM_MARK(i) do {} while (0)
#endif
    const CellMeta m = meta[cell];
    const uint32_t nU = nnz_unique[cell];
    const uint2* U = reinterpret_cast<const uint2*>(((m.lg_nb || mode_is_pug(m.mode)) ? keys1 : keys0) + m.key_off);
    const uint32_t W = lab_cnt[2 * cell], M = lab_cnt[2 * cell + 1];
    const uint32_t* lw = lab + 2 * m.key_off;
    const uint32_t* ld = lw + m.n_ref + 1;
    const uint32_t mult = cfg.usa ? 3u : 1u;
    const uint32_t capS = (nU + W) * mult;
    const EmScratch sc = em_carve(scratch, em_off[cell], nU, W, M, capS);
    uint2* out = sc.out; uint64_t* inv_pairs = sc.inv_pairs; uint32_t* order = sc.order; uint32_t* cls_first = sc.cls_first;
    uint32_t* cls_cnt = sc.cls_cnt; uint32_t* cls_woff = sc.cls_woff; uint32_t* cls_w = sc.cls_w; uint32_t* cls_sidx = sc.cls_sidx;
    uint32_t* support = sc.support; uint32_t* sib1 = sc.sib1; uint32_t* sib2 = sc.sib2; uint32_t* ucnt = sc.ucnt;
    uint32_t* slot_off = sc.slot_off; uint32_t* aid = sc.aid; uint4* ent = sc.ent; uint4* lw3 = sc.lw3;
    uint32_t* act_col = sc.act_col; uint32_t* memb = sc.memb;
    if (M == 0) {  // no multi-label class: the counts are the single-label counts (em.rs:339-341, 499-514)
        for (uint32_t i = threadIdx.x; i < nU; i += kEmNT) out[i] = make_uint2(U[i].x, __float_as_uint((float)U[i].y));
        if (threadIdx.x == 0) { out_nnz[cell] = nU; em_hdr[cell] = make_uint4(0u, 0u, 0u, 1u); }
        return;
    }
    auto lab_gt = [&](uint32_t a, uint32_t b) {  // lexicographic a > b on the gene-level labels
        const uint32_t oa = ld[2 * a], na = ld[2 * a + 1], ob = ld[2 * b], nb = ld[2 * b + 1];
        const uint32_t nm = na < nb ? na : nb;
        for (uint32_t i = 0; i < nm; ++i) {
            const uint32_t x = lw[oa + i], y = lw[ob + i];
            if (x != y) return x > y;
        }
        return na > nb;
    };
    auto lab_ne = [&](uint32_t a, uint32_t b) { return lab_gt(a, b) || lab_gt(b, a); };
    EM_MARK(0);
    // 1. classes = runs of equal labels in lexicographic order
    // The sort runs on 16-byte records out of LDS: a 63-bit key holding the label's first three genes (+1, a missing
    // gene is 0, so key order IS the lexicographic order with shorter labels first) and the molecule index; only labels
    // that tie on three genes and are longer than that fall back to the pointer-chasing comparison.
    struct LabKey { uint64_t key; uint32_t idx, len; };
    LabKey* lk = reinterpret_cast<LabKey*>(inv_pairs);  // 4 words per molecule; inv_pairs holds 2(W+1) >= 4M+2 words (every label has >= 2 genes)
    for (uint32_t i = threadIdx.x; i < M; i += kEmNT) {
        const uint32_t o = ld[2 * i], n = ld[2 * i + 1];
        uint64_t key = (uint64_t)(lw[o] + 1u) << 42;
        if (n > 1) key |= (uint64_t)(lw[o + 1] + 1u) << 21;
        if (n > 2) key |= (uint64_t)(lw[o + 2] + 1u);
        lk[i] = LabKey{key, i, n};
    }
    __syncthreads();
    auto lk_gt = [&](const LabKey& a, const LabKey& b) {
        if (a.key != b.key) return a.key > b.key;
        if (a.len <= 3 && b.len <= 3) return false;  // same three-or-fewer genes: the same label
        return lab_gt(a.idx, b.idx);
    };
    tiled_bitonic_sort_by<kEmNT, 2048>(lk, M, lk_gt, reinterpret_cast<LabKey*>(s_tile));
    for (uint32_t i = threadIdx.x; i < M; i += kEmNT) order[i] = lk[i].idx;
    __syncthreads();
    uint32_t K = 0;
    for (uint32_t base = 0; base < M; base += kEmNT) {
        const uint32_t i = base + threadIdx.x;
        bool head = i < M;
        if (head && i > 0) {
            const LabKey a = lk[i], b = lk[i - 1];
            head = a.key != b.key || ((a.len > 3 || b.len > 3) && lab_ne(a.idx, b.idx));
        }
        const uint32_t h = head;
        uint32_t tot;
        const uint32_t ex = block_excl_scan<kEmNT>(h, s_ws, tot);
        if (h) cls_first[K + ex] = i;
        K += tot;
    }
    __syncthreads();
    EM_MARK(1);
    // 2. EM label of each class: length, then contents
    auto em_label = [&](uint32_t c, uint32_t* dst) -> uint32_t {  // returns the length; writes when dst != null
        const uint32_t mol = order[cls_first[c]];
        const uint32_t o = ld[2 * mol], n = ld[2 * mol + 1];
        uint32_t w = 0;
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t gn = lw[o + i];
            uint32_t idx = gn;
            if (cfg.usa) {
                idx = gn >> 1;
                if (is_spliced(gn)) {
                    if (i + 1 < n && same_gene(gn, lw[o + i + 1])) { idx += cfg.ao; ++i; }
                } else idx += cfg.uo;
            }
            if (dst) dst[w] = idx;
            ++w;
        }
        return w;
    };
    uint32_t Wc = 0;
    for (uint32_t base = 0; base < K; base += kEmNT) {
        const uint32_t c = base + threadIdx.x;
        const uint32_t len = c < K ? em_label(c, nullptr) : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan<kEmNT>(len, s_ws, tot);
        if (c < K) {
            cls_woff[c] = Wc + ex;
            cls_cnt[c] = (c + 1 < K ? cls_first[c + 1] : M) - cls_first[c];
        }
        Wc += tot;
    }
    if (threadIdx.x == 0) cls_woff[K] = Wc;
    __syncthreads();
    for (uint32_t c = threadIdx.x; c < K; c += kEmNT) em_label(c, cls_w + cls_woff[c]);
    __syncthreads();
    EM_MARK(2);
    // 3. support = single-label columns + label slots (+ USA sibling statuses), sorted, distinct.
    // When one bit per output column fits the LDS tile next to its rank table (num_alphas <= 131072: every gene-level
    // matrix in practice), the support is a bitmap: mark, prefix-popcount, and "index of column x in the support" is
    // two LDS reads instead of a sort of 3(nU + W) values and a binary search per lookup.
    const uint32_t nwb = (cfg.num_alphas + 31) >> 5;
    const bool bm = 2 * nwb <= 8192;
    uint32_t* bm_bits = s_tile;
    uint32_t* bm_rank = s_tile + nwb;
    uint32_t S = 0;
    if (bm) {
        for (uint32_t i = threadIdx.x; i < nwb; i += kEmNT) bm_bits[i] = 0;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < nU + Wc; i += kEmNT) {
            const uint32_t x = i < nU ? U[i].x : cls_w[i - nU];
            atomicOr(&bm_bits[x >> 5], 1u << (x & 31));
            if (cfg.usa) {
                uint32_t s1, s2;
                if (x >= cfg.ao) { s1 = x - cfg.uo; s2 = x - cfg.ao; }
                else if (x >= cfg.uo) { s1 = x + cfg.uo; s2 = x - cfg.uo; }
                else { s1 = x + cfg.ao; s2 = x + cfg.uo; }
                atomicOr(&bm_bits[s1 >> 5], 1u << (s1 & 31));
                atomicOr(&bm_bits[s2 >> 5], 1u << (s2 & 31));
            }
        }
        __syncthreads();
        for (uint32_t base = 0; base < nwb; base += kEmNT) {
            const uint32_t w = base + threadIdx.x;
            const uint32_t c = w < nwb ? (uint32_t)__popc(bm_bits[w]) : 0u;
            uint32_t tot;
            const uint32_t ex = block_excl_scan<kEmNT>(c, s_ws, tot);
            if (w < nwb) bm_rank[w] = S + ex;
            S += tot;
        }
        __syncthreads();
        for (uint32_t w = threadIdx.x; w < nwb; w += kEmNT) {
            uint32_t b = bm_bits[w], o = bm_rank[w];
            for (; b; b &= b - 1) support[o++] = (w << 5) + (uint32_t)__builtin_ctz(b);
        }
        __syncthreads();
    } else {
        uint32_t nC = 0;
        {
            const uint32_t nsrc = nU + Wc;
            for (uint32_t i = threadIdx.x; i < nsrc; i += kEmNT) {
                const uint32_t x = i < nU ? U[i].x : cls_w[i - nU];
                support[i * mult] = x;
                if (cfg.usa) {
                    uint32_t s1, s2;
                    if (x >= cfg.ao) { s1 = x - cfg.uo; s2 = x - cfg.ao; }
                    else if (x >= cfg.uo) { s1 = x + cfg.uo; s2 = x - cfg.uo; }
                    else { s1 = x + cfg.ao; s2 = x + cfg.uo; }
                    support[i * mult + 1] = s1;
                    support[i * mult + 2] = s2;
                }
            }
            nC = nsrc * mult;
        }
        __syncthreads();
        tiled_bitonic_sort_by<kEmNT, 8192>(support, nC, [](uint32_t a, uint32_t b) { return a > b; }, s_tile);
        for (uint32_t base = 0; base < nC; base += kEmNT) {  // in-place unique: position S+ex <= i, so reads stay ahead of writes
            const uint32_t i = base + threadIdx.x;
            const uint32_t v = i < nC ? support[i] : 0u;
            const uint32_t h = (i < nC) && (i == 0 || v != support[i - 1]);
            uint32_t tot;
            const uint32_t ex = block_excl_scan<kEmNT>(h, s_ws, tot);
            __syncthreads();
            if (h) support[S + ex] = v;
            S += tot;
            __syncthreads();
        }
    }
    auto sup_index = [&](uint32_t x) -> uint32_t {  // position of column x in the support (x is in it)
        if (bm) return bm_rank[x >> 5] + (uint32_t)__popc(bm_bits[x >> 5] & ((1u << (x & 31)) - 1u));
        return lower_bound_u32(support, S, x);
    };
    // NOTE on the USA support: the reference marks, for a label x, x and its siblings so that reads of
    // get_abundance_for are reset every round (em.rs:351-356).  Marking both siblings for every status is a
    // superset of em.rs:101-109 (which marks exactly the statuses get_abundance_for reads); the extra entries
    // hold 0 throughout and never change a sum.
    for (uint32_t s = threadIdx.x; s < S; s += kEmNT) {
        ucnt[s] = 0;
        sib1[s] = 0xFFFFFFFFu; sib2[s] = 0xFFFFFFFFu;
        if (cfg.usa) {
            const uint32_t x = support[s];
            if (x >= cfg.ao) { sib1[s] = sup_index(x - cfg.uo); sib2[s] = sup_index(x - cfg.ao); }
            else if (x >= cfg.uo) sib1[s] = sup_index(x + cfg.uo);
            else sib1[s] = sup_index(x + cfg.ao);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nU; i += kEmNT) ucnt[sup_index(U[i].x)] = U[i].y;
    for (uint32_t c = threadIdx.x; c < K; c += kEmNT)
        for (uint32_t w = cls_woff[c]; w < cls_woff[c + 1]; ++w) {
            const uint32_t s = sup_index(cls_w[w]);
            cls_sidx[w] = s;
            inv_pairs[w] = ((uint64_t)s << 32) | c;
        }
    __syncthreads();
    EM_MARK(3);
    // 4. inverted index: for every support entry the classes containing it, ascending class
    tiled_bitonic_sort_by<kEmNT, 4096>(inv_pairs, Wc, [](uint64_t a, uint64_t b) { return a > b; }, reinterpret_cast<uint64_t*>(s_tile));
    EM_MARK(4);
    // slot_off[s] = first pair with support idx >= s: count the memberships per entry, exclusive scan
    for (uint32_t s = threadIdx.x; s <= S; s += kEmNT) slot_off[s] = 0;
    __syncthreads();
    for (uint32_t q = threadIdx.x; q < Wc; q += kEmNT) atomicAdd(&slot_off[(uint32_t)(inv_pairs[q] >> 32)], 1u);
    __syncthreads();
    // (the scans below take eight consecutive elements per thread and trip: the support of a PBMC-sized USA cell has
    // 25 000 entries, and a block scan per 256 of them was a hundred latency-bound round trips through two barriers each)
    constexpr uint32_t kSc = 8;
    {
        uint32_t carry = 0;
        for (uint32_t base = 0; base <= S; base += kSc * kEmNT) {
            const uint32_t s0 = base + kSc * threadIdx.x;
            uint32_t v[kSc], sum = 0;
#pragma unroll
            for (uint32_t j = 0; j < kSc; ++j) v[j] = s0 + j <= S ? slot_off[s0 + j] : 0u;
#pragma unroll
            for (uint32_t j = 0; j < kSc; ++j) { const uint32_t t = v[j]; v[j] = sum; sum += t; }
            uint32_t tot;
            const uint32_t ex = block_excl_scan<kEmNT>(sum, s_ws, tot);
#pragma unroll
            for (uint32_t j = 0; j < kSc; ++j) if (s0 + j <= S) slot_off[s0 + j] = carry + ex + v[j];
            carry += tot;
        }
    }
    __syncthreads();
    EM_MARK(5);
    // 4b. The rounds only ever change entries that have a single-label count or sit in some class label
    // ("active"); every other support entry (the USA sibling statuses marked for em.rs:351-356) is produced
    // as 0 by each round.  Compact the active entries and express everything the rounds touch in active ids:
    // per entry one 16-byte record, per label word one, the memberships as plain class ids.  Two extra slots
    // stand for "an inactive sibling" (the initial value in round 1, 0 afterwards) and "no sibling" (0; adding
    // +0.0f to a non-negative float is exact, so one three-term formula serves every status).
    uint32_t A = 0;
    for (uint32_t base = 0; base < S; base += kSc * kEmNT) {
        const uint32_t s0 = base + kSc * threadIdx.x;
        uint32_t uc[kSc], so[kSc + 1], sum = 0, hm = 0;
#pragma unroll
        for (uint32_t j = 0; j < kSc; ++j) uc[j] = s0 + j < S ? ucnt[s0 + j] : 0u;
#pragma unroll
        for (uint32_t j = 0; j <= kSc; ++j) so[j] = s0 + j <= S ? slot_off[s0 + j] : 0u;
#pragma unroll
        for (uint32_t j = 0; j < kSc; ++j) { const bool h = s0 + j < S && (uc[j] != 0 || so[j + 1] > so[j]); hm |= (uint32_t)h << j; sum += h; }
        uint32_t tot;
        const uint32_t ex = block_excl_scan<kEmNT>(sum, s_ws, tot);
        uint32_t run = A + ex;
#pragma unroll
        for (uint32_t j = 0; j < kSc; ++j) if (s0 + j < S) { const bool h = (hm >> j) & 1u; aid[s0 + j] = h ? run : 0xFFFFFFFFu; run += h; }
        A += tot;
    }
    __syncthreads();
    const uint32_t Z0 = A, Z1 = A + 1;
    auto amap = [&](uint32_t x) -> uint32_t {
        if (x == 0xFFFFFFFFu) return Z1;
        const uint32_t a = aid[x];
        return a == 0xFFFFFFFFu ? Z0 : a;
    };
    // (four entries / label words per thread and trip, each level of the dependent gathers issued for all four together)
    for (uint32_t s0 = threadIdx.x; s0 < S; s0 += 4 * kEmNT) {
        uint32_t a[4], u[4], x1[4], x2[4], so[4], sp[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const uint32_t s2 = s0 + j * kEmNT; a[j] = s2 < S ? aid[s2] : 0xFFFFFFFFu; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t s2 = s0 + j * kEmNT;
            const bool on = a[j] != 0xFFFFFFFFu;
            u[j] = on ? ucnt[s2] : 0u; x1[j] = on ? sib1[s2] : 0xFFFFFFFFu; x2[j] = on ? sib2[s2] : 0xFFFFFFFFu;
            so[j] = on ? slot_off[s2] : 0u; sp[j] = on ? support[s2] : 0u;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { x1[j] = amap(x1[j]); x2[j] = amap(x2[j]); }
#pragma unroll
        for (int j = 0; j < 4; ++j) if (a[j] != 0xFFFFFFFFu) { ent[a[j]] = make_uint4(u[j], x1[j], x2[j], so[j]); act_col[a[j]] = sp[j]; }
    }
    if (threadIdx.x == 0) ent[A] = make_uint4(0u, Z1, Z1, Wc);
    for (uint32_t w0 = threadIdx.x; w0 < Wc; w0 += 4 * kEmNT) {
        uint32_t s2[4], a0[4], y1[4], y2[4], mb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const uint32_t w = w0 + j * kEmNT; s2[j] = w < Wc ? cls_sidx[w] : 0u; mb[j] = w < Wc ? (uint32_t)inv_pairs[w] : 0u; }
#pragma unroll
        for (int j = 0; j < 4; ++j) { const bool on = w0 + j * kEmNT < Wc; a0[j] = on ? aid[s2[j]] : 0u; y1[j] = on ? sib1[s2[j]] : 0xFFFFFFFFu; y2[j] = on ? sib2[s2[j]] : 0xFFFFFFFFu; }
#pragma unroll
        for (int j = 0; j < 4; ++j) { y1[j] = amap(y1[j]); y2[j] = amap(y2[j]); }
#pragma unroll
        for (int j = 0; j < 4; ++j) { const uint32_t w = w0 + j * kEmNT; if (w < Wc) { lw3[w] = make_uint4(a0[j], y1[j], y2[j], 0u); memb[w] = mb[j]; } }
    }
    EM_MARK(6);
    if (threadIdx.x == 0) em_hdr[cell] = make_uint4(A, K, Wc, 0u);  // the rounds run in k_em_rounds
#ifdef AFQ_EM_TIMING
    if (threadIdx.x == 0 && (blockIdx.x % 1000) == 7) { printf("em setup nrec=%u nU=%u M=%u K=%u S=%u Wc=%u A=%u:", m.nrec, nU, M, K, S, Wc, A); for (int i = 1; i <= 6; ++i) printf(" p%d=%.3fms", i, (double)(tmark[i] - tmark[i - 1]) / 1e5); printf("\n"); }
#endif
}

// acc + sum over q in [q0, q1), in that order, of (iv(q) >= 0 ? ab * iv(q) : 0) - by the whole wave: the loads
// of 64 memberships go out together, the additions stay one after the other (float addition is not associative
// and the order is the parity contract).  An entry that sits in hundreds of classes (a highly expressed gene)
// otherwise makes its one thread walk hundreds of dependent loads per round while the wave waits.
// All 64 lanes must call it with the same arguments; every lane returns the result.
constexpr uint32_t kEmHeavy = 8;   // memberships above which an entry is summed by the wave
template <typename InvAt>
__device__ __forceinline__ float wave_ordered_sum(float acc, float ab, uint32_t q0, uint32_t q1, InvAt&& inv_at) {
    const uint32_t lane = lane_id();
    for (uint32_t base = q0; base < q1; base += 64) {
        const uint32_t q = base + lane;
        const float iv = q < q1 ? inv_at(q) : -1.0f;
        // every lane forms its own term; a skipped term is +0.0f, which leaves a non-negative sum bit for bit
        // unchanged, so the chain below needs no branches: 64 dependent adds fed by constant-lane reads
        const float term = iv >= 0.0f ? ab * iv : 0.0f;
        const uint32_t tb = __float_as_uint(term);
#pragma unroll
        for (int i = 0; i < 64; ++i) acc += __uint_as_float(__builtin_amdgcn_readlane(tb, i));
    }
    return acc;
}
__device__ __forceinline__ uint32_t bcast_u32(uint32_t v, uint32_t src_lane) { return __builtin_amdgcn_readlane(v, (int)src_lane); }
__device__ __forceinline__ float bcast_f32(float v, uint32_t src_lane) { return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), (int)src_lane)); }

// ---------------------------------------------------------------------------
// The EM rounds.  k_em left, per cell, the compact structures of section 4b in global scratch; the rounds
// themselves used to stream them from L2/HBM every round (~1.5 MB of cache lines per round and cell - with a
// thousand cells in flight that is the memory system's full throughput, for 20-100 rounds).  Here one
// 1024-thread workgroup takes a cell and keeps everything the rounds touch ON CHIP: abundances, 1/denominators,
// class offsets/counts, label words and memberships (16-bit ids) in LDS, the per-entry records in registers
// (kEmPer entries per thread).  A round is then LDS traffic and three barriers.  Cells too big for that (more than
// kEmPer x 1024 active entries or classes, or > 144 KiB of LDS) run the same arithmetic out of global memory.
// Arithmetic and its order are unchanged (bit-identical to the oracle).
constexpr int kEmRNT = 1024;
constexpr uint32_t kEmPer = 10;   // entries per thread in registers (ten cost the same registers and spills as eight: cells of up to 10 240 entries)
constexpr uint32_t kEmLdsWords = 36 * 1024;
__global__ __launch_bounds__(kEmRNT) void k_em_rounds(const CellMeta* __restrict__ meta, const uint32_t* __restrict__ nnz_unique,
                                                     const uint32_t* __restrict__ lab_cnt, const uint64_t* __restrict__ em_off,
                                                     uint32_t* __restrict__ scratch, uint32_t* __restrict__ out_nnz,
                                                     const uint4* __restrict__ em_hdr, const uint32_t* __restrict__ em_order, EmCfg cfg) {
    __shared__ uint32_t s_ws[kEmRNT / 64];
    __shared__ uint32_t s_flag[2];
    __shared__ __attribute__((aligned(16))) uint32_t s_mem[kEmLdsWords];
    const uint32_t cell = em_order[blockIdx.x];
#ifdef AFQ_EM_TIMING
    __shared__ unsigned long long tm2[6];
#define EM2_MARK(i) do { __syncthreads(); if (threadIdx.x == 0 && (blockIdx.x % 1000) == 7) tm2[i] = wall_clock64(); } while (0)
    __shared__ unsigned long long tph[5];
    unsigned long long tph_t = 0;
    if (threadIdx.x < 5) tph[threadIdx.x] = 0;
#define EM2_PH0() do { tph_t = wall_clock64(); } while (0)
#define EM2_PH(i) do { if (threadIdx.x == 0) { const unsigned long long n_ = wall_clock64(); tph[i] += n_ - tph_t; tph_t = n_; } } while (0)
#else
#define EM2_MARK(i) do {} while (0)
#define EM2_PH0() do {} while (0)
#define EM2_PH(i) do {} while (0)
#endif
    const uint4 hdr = em_hdr[cell];
    if (hdr.w) return;  // no multi-label class: k_em already wrote the row
    const uint32_t A = hdr.x, K = hdr.y, Wc = hdr.z;
    const uint32_t nU = nnz_unique[cell], W = lab_cnt[2 * cell], M = lab_cnt[2 * cell + 1];
    const uint32_t capS = (nU + W) * (cfg.usa ? 3u : 1u);
    const EmScratch sc = em_carve(scratch, em_off[cell], nU, W, M, capS);
    uint2* out = sc.out;
    const uint4* ent = sc.ent;
    const uint4* lw3 = sc.lw3;
    const uint32_t* memb = sc.memb;
    const uint32_t* cls_woff = sc.cls_woff;
    const uint32_t* cls_cnt = sc.cls_cnt;
    const uint32_t* act_col = sc.act_col;
    const uint32_t Z0 = A, Z1 = A + 1;
    const uint32_t tid = threadIdx.x;
    const uint32_t lw_words = (3 * Wc + 1) / 2, mb_words = (Wc + 1) / 2;
    const uint32_t need = (A + 2) + K + (K + 1) + K + lw_words + mb_words;
    const bool fits = need <= kEmLdsWords && A <= kEmPer * kEmRNT && K <= kEmPer * kEmRNT;
    uint32_t nout = 0;
    [[maybe_unused]] uint32_t it_dbg = 0;
    EM2_MARK(0);
    if (fits) {
        float* vin = reinterpret_cast<float*>(s_mem);
        float* inv = vin + (A + 2);
        uint32_t* woff = reinterpret_cast<uint32_t*>(inv + K);
        uint32_t* cnt = woff + (K + 1);
        uint16_t* lw16 = reinterpret_cast<uint16_t*>(cnt + K);
        uint16_t* mb16 = reinterpret_cast<uint16_t*>(cnt + K + lw_words);
        for (uint32_t c = tid; c <= K; c += kEmRNT) woff[c] = cls_woff[c];
        for (uint32_t c = tid; c < K; c += kEmRNT) cnt[c] = cls_cnt[c];
        for (uint32_t w = tid; w < Wc; w += kEmRNT) {
            const uint4 l = lw3[w];
            lw16[3 * w] = (uint16_t)l.x; lw16[3 * w + 1] = (uint16_t)l.y; lw16[3 * w + 2] = (uint16_t)l.z;
            mb16[w] = (uint16_t)memb[w];
        }
        uint32_t e_cnt[kEmPer], e_sib[kEmPer], e_q0[kEmPer], e_q1[kEmPer];
        float acc[kEmPer];
        const float uni = 1.0f / (float)cfg.num_alphas;
        if (tid == 0) { s_flag[0] = 0; s_flag[1] = 0; }
        __syncthreads();
        uint32_t n_hv_mine = 0;
#pragma unroll
        for (uint32_t j = 0; j < kEmPer; ++j) {
            const uint32_t a = tid + j * kEmRNT;
            e_cnt[j] = 0; e_sib[j] = 0; e_q0[j] = 0; e_q1[j] = 0; acc[j] = 0.0f;
            if (a < A) {
                const uint4 e = ent[a];
                e_cnt[j] = e.x; e_sib[j] = e.y | (e.z << 16); e_q0[j] = e.w; e_q1[j] = ent[a + 1].w;
                vin[a] = cfg.init_uniform ? uni : ((float)e.x + 0.5f) * 1e-3f;
                n_hv_mine += e_q1[j] - e_q0[j] > kEmHeavy;
            }
        }
        // Entries that sit in many classes (highly expressed genes: hundreds of memberships) are summed by a whole wave each,
        // the additions one after the other in class order (wave_ordered_sum).  Those entries have the lowest ids, i.e. they
        // all belong to the threads of wave 0 - left there, one wave walks every long chain of the cell while fifteen wait
        // at the barrier.  They go on a list in LDS instead (6 words each) and the waves take them in turn.
        if (n_hv_mine) atomicAdd(&s_flag[1], n_hv_mine);
        if (tid == 0) { vin[Z0] = cfg.init_uniform ? uni : ((float)0u + 0.5f) * 1e-3f; vin[Z1] = 0.0f; }
        __syncthreads();
        const uint32_t NH = s_flag[1];
        const bool hv_list = NH > 0 && need + 6 * NH <= kEmLdsWords;
        uint32_t* hv = s_mem + need;   // {entry, count, siblings, q0, q1, result} per listed entry
        uint32_t hv_mask = 0;          // bit j: my entry j is on the list (its owner leaves it alone)
        if (hv_list) {
#pragma unroll
            for (uint32_t j = 0; j < kEmPer; ++j) {
                const uint32_t a = tid + j * kEmRNT;
                if (a < A && e_q1[j] - e_q0[j] > kEmHeavy) {
                    const uint32_t i = atomicAdd(&s_flag[0], 1u);
                    hv[6 * i] = a; hv[6 * i + 1] = e_cnt[j]; hv[6 * i + 2] = e_sib[j]; hv[6 * i + 3] = e_q0[j]; hv[6 * i + 4] = e_q1[j];
                    hv_mask |= 1u << j;
                }
            }
        }
        __syncthreads();
        if (hv_list && NH <= (uint32_t)kEmRNT) {   // longest chains first: a wave that draws the longest last would hold the barrier alone
            uint32_t r[5] = {0, 0, 0, 0, 0}, rank = 0;
            if (tid < NH) {
#pragma unroll
                for (int x = 0; x < 5; ++x) r[x] = hv[6 * tid + x];
                const uint32_t dg = r[4] - r[3];
                for (uint32_t o = 0; o < NH; ++o) { const uint32_t d2 = hv[6 * o + 4] - hv[6 * o + 3]; rank += d2 > dg || (d2 == dg && o < tid); }
            }
            __syncthreads();
            if (tid < NH) {
#pragma unroll
                for (int x = 0; x < 5; ++x) hv[6 * rank + x] = r[x];
            }
            __syncthreads();
        }
        EM2_MARK(1);
        uint32_t it = 0;
        bool conv = true, last_round = false;
        EM2_PH0();
        while (it < kMinIter || (it < kMaxIter && !conv) || last_round) {
            // (A) per class: denominator in label order (get_abundance_for, em.rs:167-187)
#pragma unroll
            for (uint32_t j = 0; j < kEmPer; ++j) {
                const uint32_t c = tid + j * kEmRNT;
                if (c < K) {
                    float denom = 0.0f;
                    const uint32_t we = woff[c + 1];
                    for (uint32_t w = woff[c]; w < we; ++w)
                        denom += (vin[lw16[3 * w + 1]] + vin[lw16[3 * w + 2]]) + vin[lw16[3 * w]];
                    inv[c] = denom > 0.0f ? (float)cnt[c] / denom : -1.0f;
                }
            }
            if (tid == 0) { s_flag[0] = 0; s_flag[1] = 0; }   // [1]: next listed entry (the waves draw them as they come free)
            __syncthreads();
            EM2_PH(0);
            // (B) per active entry: single-label count, then class contributions in class order
            bool bad = false;
#pragma unroll
            for (uint32_t j = 0; j < kEmPer; ++j) {
                const uint32_t a = tid + j * kEmRNT;
                const bool valid = a < A && !((hv_mask >> j) & 1u);
                const bool heavy = valid && e_q1[j] - e_q0[j] > kEmHeavy;
                float x = 0.0f, old = 0.0f, ab = 0.0f;
                if (valid) {
                    if (e_cnt[j]) x += (float)e_cnt[j];
                    old = vin[a];
                    ab = (vin[e_sib[j] & 0xFFFFu] + vin[e_sib[j] >> 16]) + old;
                    if (!heavy)
                        for (uint32_t q = e_q0[j]; q < e_q1[j]; ++q) {
                            const float iv = inv[mb16[q]];
                            if (iv >= 0.0f) x += ab * iv;
                        }
                }
                for (uint64_t hm = __ballot(heavy); hm; hm &= hm - 1) {
                    const uint32_t L = (uint32_t)__builtin_ctzll(hm);
                    const float r = wave_ordered_sum(bcast_f32(x, L), bcast_f32(ab, L), bcast_u32(e_q0[j], L), bcast_u32(e_q1[j], L),
                                                     [&](uint32_t q) { return inv[mb16[q]]; });
                    if (lane_id() == L) x = r;
                }
                if (valid) {
                    acc[j] = x;
                    if (x > kAlphaCheckCutoff && fabsf(old - x) > kRelDiffTol) bad = true;
                }
            }
            EM2_PH(1);
            if (hv_list)
                for (;;) {   // the listed entries, a wave each: drawn from a counter, so a wave that met a long chain takes fewer
                    uint32_t i = 0;
                    if (lane_id() == 0) i = atomicAdd(&s_flag[1], 1u);
                    i = __builtin_amdgcn_readfirstlane(i);
                    if (i >= NH) break;
                    const uint32_t a = hv[6 * i], hc = hv[6 * i + 1], hs = hv[6 * i + 2];
                    const float old = vin[a];
                    const float ab = (vin[hs & 0xFFFFu] + vin[hs >> 16]) + old;
                    const float x = wave_ordered_sum(hc ? (float)hc : 0.0f, ab, hv[6 * i + 3], hv[6 * i + 4], [&](uint32_t q) { return inv[mb16[q]]; });
                    if (lane_id() == 0) hv[6 * i + 5] = __float_as_uint(x);
                    if (x > kAlphaCheckCutoff && fabsf(old - x) > kRelDiffTol) bad = true;
                }
            EM2_PH(2);
            if (bad) s_flag[0] = 1;
            __syncthreads();  // every read of the old abundances is done
            EM2_PH(3);
            conv = s_flag[0] == 0;
#pragma unroll
            for (uint32_t j = 0; j < kEmPer; ++j) {
                const uint32_t a = tid + j * kEmRNT;
                if (a < A && !((hv_mask >> j) & 1u)) vin[a] = acc[j];
            }
            if (hv_list) for (uint32_t i = tid; i < NH; i += kEmRNT) vin[hv[6 * i]] = __uint_as_float(hv[6 * i + 5]);
            if (tid == 0) vin[Z0] = 0.0f;  // inactive entries come out of every round as 0
            ++it;
            __syncthreads();
            EM2_PH(4);
            if (cfg.usa) {
                if (last_round) break;
                if (it >= kMinIter && conv) {
#pragma unroll
                    for (uint32_t j = 0; j < kEmPer; ++j) {
                        const uint32_t a = tid + j * kEmRNT;
                        if (a < A && vin[a] < kMinOutputAlpha) vin[a] = 0.0f;
                    }
                    last_round = true;
                    __syncthreads();
                }
            }
        }
        it_dbg = it;
#ifdef AFQ_EM_TIMING
        if (tid == 0 && (blockIdx.x % 1000) == 7) {
            uint32_t mx = 0; for (uint32_t i = 0; i < (hv_list ? NH : 0u); ++i) mx = max(mx, hv[6 * i + 4] - hv[6 * i + 3]);
            printf("em fits A=%u K=%u NH=%u list=%d maxdeg=%u it=%u: A=%.3f B=%.3f list(wave0)=%.3f wait=%.3f wb=%.3f ms\n", A, K, NH, (int)hv_list, mx, it,
                   (double)tph[0]/1e5, (double)tph[1]/1e5, (double)tph[2]/1e5, (double)tph[3]/1e5, (double)tph[4]/1e5);
        }
#endif
        EM2_MARK(2);
        // floor and emit the non-zero alphas in column order (active ids ascend with the column)
        for (uint32_t base = 0; base < A; base += kEmRNT) {
            const uint32_t a = base + tid;
            float v = a < A ? vin[a] : 0.0f;
            if (v < kMinOutputAlpha) v = 0.0f;
            const uint32_t h = v > 0.0f;
            uint32_t tot;
            const uint32_t ex = block_excl_scan<kEmRNT>(h, s_ws, tot);
            if (h) out[nout + ex] = make_uint2(act_col[a], __float_as_uint(v));
            nout += tot;
        }
    } else {
        // bigger cells: the two randomly accessed arrays (abundances, 1/denominators) still live in LDS when they
        // fit; the entry / label-word / membership records are streamed, coalesced, from global memory
        const bool mid = (A + 2) + K <= kEmLdsWords;
        float* vin = mid ? reinterpret_cast<float*>(s_mem) : sc.a_in;
        float* vout = sc.a_out;
        float* inv = mid ? reinterpret_cast<float*>(s_mem) + (A + 2) : sc.inv;
        // ... and the memberships (class ids, 16 bits each) too when there is room: an entry's walk over its classes is then
        // LDS reads only - out of global memory it is one dependent ~2 us load per class, round after round
        const bool mb_lds = mid && K <= 65536u && (A + 2) + K + mb_words <= kEmLdsWords;
        uint16_t* mb16 = reinterpret_cast<uint16_t*>(s_mem + (A + 2) + K);
        if (mb_lds) for (uint32_t w = threadIdx.x; w < Wc; w += kEmRNT) mb16[w] = (uint16_t)memb[w];
        auto memb_at = [&](uint32_t q) -> uint32_t { return mb_lds ? (uint32_t)mb16[q] : memb[q]; };
        // the entries summed by a whole wave (see the on-chip tier): on a list in LDS, taken by the waves in turn
        const uint32_t hv_base = (A + 2) + K + (mb_lds ? mb_words : 0u);
        uint32_t* hv = s_mem + hv_base;   // {entry, count, siblings, q0, q1}
        if (threadIdx.x == 0) { s_flag[0] = 0; s_flag[1] = 0; }
        __syncthreads();
        if (mid) {
            uint32_t mine = 0;
            for (uint32_t a = threadIdx.x; a < A; a += kEmRNT) mine += ent[a + 1].w - ent[a].w > 2 + kEmHeavy;
            if (mine) atomicAdd(&s_flag[1], mine);
        }
        __syncthreads();
        const uint32_t NH = s_flag[1];
        const bool hv_list = mid && NH > 0 && hv_base + 5 * NH <= kEmLdsWords;
        if (hv_list)
            for (uint32_t a = threadIdx.x; a < A; a += kEmRNT) {
                const uint4 e = ent[a];
                const uint32_t q1 = ent[a + 1].w;
                if (q1 - e.w > 2 + kEmHeavy) {
                    const uint32_t i = atomicAdd(&s_flag[0], 1u);
                    hv[5 * i] = a; hv[5 * i + 1] = e.x; hv[5 * i + 2] = e.y | (e.z << 16); hv[5 * i + 3] = e.w; hv[5 * i + 4] = q1;
                }
            }
        EM2_MARK(1);
        const float uni = 1.0f / (float)cfg.num_alphas;
        for (uint32_t a = threadIdx.x; a < A; a += kEmRNT) vin[a] = cfg.init_uniform ? uni : ((float)ent[a].x + 0.5f) * 1e-3f;
        if (threadIdx.x == 0) { vin[Z0] = cfg.init_uniform ? uni : ((float)0u + 0.5f) * 1e-3f; vin[Z1] = 0.0f; }
        __syncthreads();
        uint32_t it = 0;
        bool conv = true, last_round = false;
        while (it < kMinIter || (it < kMaxIter && !conv) || last_round) {
            // (A) per class: denominator in label order (get_abundance_for, em.rs:167-187)
            // four classes per thread per trip, their loads issued together: the rounds are chains of dependent
            // L2 round trips, and a thread walking its classes one at a time has only one chain in flight
            for (uint32_t c0 = threadIdx.x; c0 < K; c0 += 4 * kEmRNT) {
                uint32_t wb[4], we[4], cn[4];
                uint4 l0[4], l1[4], l2[4];
    #pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t c = c0 + j * kEmRNT;
                    const bool ok = c < K;
                    wb[j] = ok ? cls_woff[c] : 0u;
                    we[j] = ok ? cls_woff[c + 1] : 0u;
                    cn[j] = ok ? cls_cnt[c] : 0u;
                }
    #pragma unroll
                for (int j = 0; j < 4; ++j) {
                    l0[j] = lw3[wb[j] < we[j] ? wb[j] : 0u];
                    l1[j] = lw3[wb[j] + 1 < we[j] ? wb[j] + 1 : 0u];
                    l2[j] = lw3[wb[j] + 2 < we[j] ? wb[j] + 2 : 0u];
                }
    #pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t c = c0 + j * kEmRNT;
                    if (c >= K) continue;
                    float denom = 0.0f;
                    if (wb[j] < we[j]) denom += (vin[l0[j].y] + vin[l0[j].z]) + vin[l0[j].x];
                    if (wb[j] + 1 < we[j]) denom += (vin[l1[j].y] + vin[l1[j].z]) + vin[l1[j].x];
                    if (wb[j] + 2 < we[j]) denom += (vin[l2[j].y] + vin[l2[j].z]) + vin[l2[j].x];
                    for (uint32_t w = wb[j] + 3; w < we[j]; ++w) {
                        const uint4 l = lw3[w];
                        denom += (vin[l.y] + vin[l.z]) + vin[l.x];
                    }
                    inv[c] = denom > 0.0f ? (float)cn[j] / denom : -1.0f;
                }
            }
            if (threadIdx.x == 0) { s_flag[0] = 0; s_flag[1] = 0; }
            __syncthreads();
            // (B) per active entry: single-label count, then class contributions in class order
            bool bad = false;
            // four entries of this thread (a0, a0 + 1024, ...): new abundance in out[j] (ok bit j set) - called with a wave-uniform
            // a0 - lane: the heavy-entry sums need every lane
            auto four_entries = [&](uint32_t a0, float (&out)[4]) -> uint32_t {
                uint32_t okm = 0;
                uint4 e[4];
                uint32_t qe[4], m0[4], m1[4];
                float i0[4], i1[4];
    #pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t a = a0 + j * kEmRNT;
                    e[j] = ent[a < A ? a : A];         // ent[A] is the sentinel record
                    qe[j] = ent[a < A ? a + 1 : A].w;
                }
    #pragma unroll
                for (int j = 0; j < 4; ++j) {
                    m0[j] = memb_at(e[j].w < qe[j] ? e[j].w : 0u);
                    m1[j] = memb_at(e[j].w + 1 < qe[j] ? e[j].w + 1 : 0u);
                }
    #pragma unroll
                for (int j = 0; j < 4; ++j) { i0[j] = inv[m0[j]]; i1[j] = inv[m1[j]]; }
    #pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t a = a0 + j * kEmRNT;
                    const bool valid = a < A && !(hv_list && qe[j] - e[j].w > 2 + kEmHeavy);   // (listed entries: below)
                    const bool heavy = valid && qe[j] - e[j].w > 2 + kEmHeavy;
                    float acc = 0.0f, old = 0.0f, ab = 0.0f;
                    if (valid) {
                        if (e[j].x) acc += (float)e[j].x;
                        old = vin[a];
                        ab = (vin[e[j].y] + vin[e[j].z]) + old;
                        if (e[j].w < qe[j] && i0[j] >= 0.0f) acc += ab * i0[j];
                        if (e[j].w + 1 < qe[j] && i1[j] >= 0.0f) acc += ab * i1[j];
                        if (!heavy)
                            for (uint32_t q = e[j].w + 2; q < qe[j]; ++q) {
                                const float iv = inv[memb_at(q)];
                                if (iv >= 0.0f) acc += ab * iv;
                            }
                    }
                    for (uint64_t hm = __ballot(heavy); hm; hm &= hm - 1) {
                        const uint32_t L = (uint32_t)__builtin_ctzll(hm);
                        const float r = wave_ordered_sum(bcast_f32(acc, L), bcast_f32(ab, L), bcast_u32(e[j].w, L) + 2, bcast_u32(qe[j], L),
                                                         [&](uint32_t q) { return inv[memb_at(q)]; });
                        if (lane_id() == L) acc = r;
                    }
                    out[j] = acc;
                    if (valid) {
                        okm |= 1u << j;
                        if (acc > kAlphaCheckCutoff && fabsf(old - acc) > kRelDiffTol) bad = true;
                    }
                }
                return okm;
            };
            // Up to 16 384 entries the new abundances wait for the barrier in registers (sixteen per thread) instead of
            // going out to global memory and back.
            const bool nv_regs = A <= 16u * kEmRNT;
            float nv[16];
            uint32_t nv_ok = 0;
            if (nv_regs) {
    #pragma unroll
                for (int t = 0; t < 16; ++t) nv[t] = 0.0f;
                uint32_t trip = 0;
                for (uint32_t a0 = threadIdx.x; a0 - lane_id() < A; a0 += 4 * kEmRNT, ++trip) {   // (one copy of the body; the trip picks the registers)
                    float o4[4];
                    const uint32_t okm = four_entries(a0, o4);
    #pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (trip == (uint32_t)t) {
    #pragma unroll
                            for (int j2 = 0; j2 < 4; ++j2) nv[4 * t + j2] = o4[j2];
                        }
                    nv_ok |= okm << (4 * trip);
                }
            } else {
                for (uint32_t a0 = threadIdx.x; a0 - lane_id() < A; a0 += 4 * kEmRNT) {
                    float o4[4];
                    const uint32_t okm = four_entries(a0, o4);
    #pragma unroll
                    for (int j2 = 0; j2 < 4; ++j2) if ((okm >> j2) & 1u) vout[a0 + j2 * kEmRNT] = o4[j2];
                }
            }
            if (hv_list)
                for (;;) {
                    uint32_t i = 0;
                    if (lane_id() == 0) i = atomicAdd(&s_flag[1], 1u);
                    i = __builtin_amdgcn_readfirstlane(i);
                    if (i >= NH) break;
                    const uint32_t a = hv[5 * i], hc = hv[5 * i + 1], hs = hv[5 * i + 2];
                    const float old = vin[a];
                    const float ab = (vin[hs & 0xFFFFu] + vin[hs >> 16]) + old;
                    const float x = wave_ordered_sum(hc ? (float)hc : 0.0f, ab, hv[5 * i + 3], hv[5 * i + 4], [&](uint32_t q) { return inv[memb_at(q)]; });
                    if (lane_id() == 0) vout[a] = x;
                    if (x > kAlphaCheckCutoff && fabsf(old - x) > kRelDiffTol) bad = true;
                }
            if (bad) s_flag[0] = 1;
            __syncthreads();
            conv = s_flag[0] == 0;
            if (nv_regs) {
    #pragma unroll
                for (int t = 0; t < 16; ++t) if ((nv_ok >> t) & 1u) vin[threadIdx.x + (uint32_t)(t >> 2) * 4u * kEmRNT + (uint32_t)(t & 3) * kEmRNT] = nv[t];
                if (hv_list) for (uint32_t i = threadIdx.x; i < NH; i += kEmRNT) vin[hv[5 * i]] = vout[hv[5 * i]];
            } else
                for (uint32_t a = threadIdx.x; a < A; a += kEmRNT) vin[a] = vout[a];
            if (threadIdx.x == 0) vin[Z0] = 0.0f;  // inactive entries come out of every round as 0
            ++it;
            __syncthreads();
            if (cfg.usa) {
                if (last_round) break;
                if (it >= kMinIter && conv) {
                    for (uint32_t a = threadIdx.x; a < A; a += kEmRNT) if (vin[a] < kMinOutputAlpha) vin[a] = 0.0f;
                    last_round = true;
                    __syncthreads();
                }
            }
        }
        it_dbg = it + (mb_lds ? 1000u : 0u) + (mid ? 10000u : 0u);
        EM2_MARK(2);
        // 6. floor and emit the non-zero alphas in column order (active ids ascend with the column)
        for (uint32_t base = 0; base < A; base += kEmRNT) {
            const uint32_t a = base + threadIdx.x;
            float v = a < A ? vin[a] : 0.0f;
            if (v < kMinOutputAlpha) v = 0.0f;
            const uint32_t h = v > 0.0f;
            uint32_t tot;
            const uint32_t ex = block_excl_scan<kEmRNT>(h, s_ws, tot);
            if (h) out[nout + ex] = make_uint2(act_col[a], __float_as_uint(v));
            nout += tot;
        }
    }
    if (threadIdx.x == 0) out_nnz[cell] = nout;
    EM2_MARK(3);
#ifdef AFQ_EM_TIMING
    if (threadIdx.x == 0 && (blockIdx.x % 1000) == 7) printf("em rounds cell nrec=%u A=%u K=%u Wc=%u need=%u fits=%d it=%u: load=%.3f rounds=%.3f out=%.3f total=%.3f ms\n", meta[cell].nrec, A, K, Wc, need, (int)fits, it_dbg, (double)(tm2[1]-tm2[0])/1e5, (double)(tm2[2]-tm2[1])/1e5, (double)(tm2[3]-tm2[2])/1e5, (double)(tm2[3]-tm2[0])/1e5);
#endif
}

// EM output pairs -> final CSR
__global__ __launch_bounds__(256) void k_compact_em(uint32_t n_cells, const uint64_t* __restrict__ em_off,
                                                   const uint32_t* __restrict__ scratch, const uint32_t* __restrict__ nnz,
                                                   const uint64_t* __restrict__ cell_ptr, uint32_t* __restrict__ gene,
                                                   float* __restrict__ val) {
    const uint32_t cell = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (cell >= n_cells) return;
    const uint2* src = reinterpret_cast<const uint2*>(scratch + em_off[cell]);
    const uint32_t n = nnz[cell];
    const uint64_t o = cell_ptr[cell];
    for (uint32_t i = lane_id(); i < n; i += 64) {
        const uint2 p = src[i];
        gene[o + i] = p.x;
        val[o + i] = __uint_as_float(p.y);
    }
}

// `-d` / --dump-eqclasses: the cell's gene-level equivalence classes, i.e. what the reference reads off `gene_eqc`
// after resolution (quant.rs:1282-1307).  They are all still on the device once k_em has run: a single-label
// molecule was counted as an output column, which names its label uniquely (non-USA column g <- [g]; USA S column
// g <- [2g], U <- [2g+1], A <- [2g, 2g+1]: utils.rs:865-925 read backwards), and every other label is a class of
// k_em's step 1 (label of its first molecule, multiplicity cls_cnt).  Cells that took the tiny-cell path never
// touch gene_eqc (quant.rs:794-845) and report no classes.  One wave per cell; pass 1 sizes, pass 2 fills.
__device__ __forceinline__ bool cell_has_eqclasses(uint32_t mode) {
    return mode == kModeCrLikeEm || mode == kModePugEm || mode == kModePugGeneEm;
}
__global__ __launch_bounds__(256) void k_eqc_dump(uint32_t n_cells, const CellMeta* __restrict__ meta,
                                                 const uint32_t* __restrict__ nnz_unique, const uint64_t* __restrict__ keys0,
                                                 const uint64_t* __restrict__ keys1, const uint32_t* __restrict__ lab,
                                                 const uint32_t* __restrict__ lab_cnt, const uint64_t* __restrict__ em_off,
                                                 const uint32_t* __restrict__ scratch, const uint4* __restrict__ em_hdr, EmCfg cfg,
                                                 uint32_t* __restrict__ n_cls, uint32_t* __restrict__ n_words,
                                                 const uint64_t* __restrict__ cls_ptr, const uint64_t* __restrict__ word_ptr,
                                                 uint32_t* __restrict__ o_len, uint32_t* __restrict__ o_count,
                                                 uint32_t* __restrict__ o_labels) {
    const uint32_t cell = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (cell >= n_cells) return;
    const uint32_t lane = lane_id();
    const CellMeta m = meta[cell];
    const bool fill = cls_ptr != nullptr;
    if (!cell_has_eqclasses(m.mode)) { if (!fill && lane == 0) { n_cls[cell] = 0; n_words[cell] = 0; } return; }
    const uint32_t nU = nnz_unique[cell];
    const uint2* U = reinterpret_cast<const uint2*>(((m.lg_nb || mode_is_pug(m.mode)) ? keys1 : keys0) + m.key_off);
    const uint32_t W = lab_cnt[2 * cell], M = lab_cnt[2 * cell + 1];
    const uint32_t K = M ? em_hdr[cell].y : 0u;
    const uint32_t* lw = lab + 2 * m.key_off;
    const uint32_t* ld = lw + m.n_ref + 1;
    const EmScratch sc = em_carve(const_cast<uint32_t*>(scratch), em_off[cell], nU, W, M, (nU + W) * (cfg.usa ? 3u : 1u));
    auto ulen = [&](uint32_t col) -> uint32_t { return (cfg.usa && col >= cfg.ao) ? 2u : 1u; };
    if (!fill) {
        uint32_t words = 0;
        for (uint32_t i = lane; i < nU; i += 64) words += ulen(U[i].x);
        for (uint32_t c = lane; c < K; c += 64) words += ld[2 * sc.order[sc.cls_first[c]] + 1];
        for (int d = 32; d; d >>= 1) words += __shfl_xor(words, d);
        if (lane == 0) { n_cls[cell] = nU + K; n_words[cell] = words; }
        return;
    }
    const uint64_t c0 = cls_ptr[cell];
    uint32_t wo = 0;   // words written so far (wave-uniform)
    for (uint32_t base = 0; base < nU + K; base += 64) {
        const uint32_t i = base + lane;
        uint32_t len = 0, cnt = 0, col = 0, mol = 0;
        if (i < nU) { col = U[i].x; cnt = U[i].y; len = ulen(col); }
        else if (i < nU + K) { const uint32_t c = i - nU; mol = sc.order[sc.cls_first[c]]; len = ld[2 * mol + 1]; cnt = sc.cls_cnt[c]; }
        uint32_t incl = len;
        for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(incl, d); if ((int)lane >= d) incl += t; }
        const uint64_t o = word_ptr[cell] + wo + (incl - len);
        if (i < nU) {
            if (!cfg.usa) o_labels[o] = col;
            else if (col >= cfg.ao) { o_labels[o] = 2 * (col - cfg.ao); o_labels[o + 1] = 2 * (col - cfg.ao) + 1; }
            else if (col >= cfg.uo) o_labels[o] = 2 * (col - cfg.uo) + 1;
            else o_labels[o] = 2 * col;
        } else if (i < nU + K) {
            const uint32_t src = ld[2 * mol];
            for (uint32_t j = 0; j < len; ++j) o_labels[o + j] = lw[src + j];
        }
        if (i < nU + K) { o_len[c0 + i] = len; o_count[c0 + i] = cnt; }
        wo += __shfl(incl, 63);
    }
}

// ---------------------------------------------------------------------------
// -b / --num-bootstraps (run_bootstrap_subset_with_scratch, em.rs:585-690; Multinomial, multinomial.rs:9-49; summaries
// quant.rs:157-210).  Input: the cell's gene-level classes as k_eqc_dump left them (the canonical class order).  Per
// replicate the class counts are redrawn - N = sum of counts draws, each a uniform integer below N located in the
// cumulative counts - and re-estimated by the EM of em_optimize_subset_impl with usa_offsets = None (labels are gene_eqc's
// gene ids as they are, also in USA mode: quant.rs:1028-1038) from a random start.  The reference's generator is an
// unseeded ThreadRng; here Philox4x32-10 keyed by the seed and counted by (cell index, replicate, draw), the streams
// being those of the oracle's header, so that the two agree bit for bit.  f32 order: a class's denominator adds its
// label in label order, an alpha adds its classes' shares in class order (pairs (entry, class) sorted) - the order of the
// sequential loop (em.rs:189-218).  One workgroup per cell.
constexpr int kBootNT = 1024;
constexpr uint32_t kBootLds = 11264;  // classes / support entries whose working arrays live in LDS (132 KiB)
constexpr uint32_t kBootHeavy = 32;   // classes above which an entry is summed by its whole wave (one entry at a time)
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t (&o)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

struct BootCfg { uint32_t B, summary_stat; uint64_t seed, first_cell_index; uint32_t usa, uo, ao, pad; };

// per-cell scratch words for K classes with W label words
uint64_t boot_scratch_words(uint64_t K, uint64_t W, uint32_t B, bool summary_stat) {
    return (K + 1) + 3 * K + 3 * W + 2 * W + (W + 1) + 2 * W + (summary_stat ? 2 * W : (uint64_t)B * W) + 8;
}

// INFER = true is `alevin-fry infer` (src/infer.rs:31-426) on the same machinery: the classes are a cell's row of the
// equivalence-class count matrix in column (class id) order, the counts are taken as they are, the start is the
// informative one ((single-label count + 0.5) * 1e-3, em.rs:376-378), and with USA offsets a label's share goes by
// get_abundance_for (em.rs:167-187: S and U lean on the gene's A, A on all three) with the sibling statuses in the
// support (em.rs:95-110).  One EM per cell; o_mean carries the abundances.
template <bool INFER>
__global__ __launch_bounds__(kBootNT) void k_boot(const uint64_t* __restrict__ cls_ptr, const uint64_t* __restrict__ word_ptr,
                                                 const uint32_t* __restrict__ g_len, const uint32_t* __restrict__ g_cnt,
                                                 const uint32_t* __restrict__ g_lab, const uint64_t* __restrict__ scr_off,
                                                 uint32_t* __restrict__ scratch, BootCfg cfg, uint32_t* __restrict__ n_support,
                                                 uint32_t* __restrict__ o_col, float* __restrict__ o_mean, float* __restrict__ o_var) {
    __shared__ __attribute__((aligned(16))) uint32_t s_buf[3 * kBootLds + 2048];   // sort tile (8192 u32 / 4096 u64), then the EM arrays
    __shared__ uint32_t s_ws[kBootNT / 64];
    __shared__ uint32_t s_flag[2];
    const uint32_t cell = blockIdx.x, tid = threadIdx.x;
    const uint64_t c0 = cls_ptr[cell], w0 = word_ptr[cell];
    const uint32_t K = (uint32_t)(cls_ptr[cell + 1] - c0), W = (uint32_t)(word_ptr[cell + 1] - w0);
    if (K == 0) { if (tid == 0) n_support[cell] = 0; return; }   // tiny-path cell (or nothing resolved): no bootstraps
    const uint32_t* len = g_len + c0; const uint32_t* cnt0 = g_cnt + c0; const uint32_t* lab = g_lab + w0;
    const bool usa = INFER && cfg.usa;
    const uint32_t Ws = usa ? 3 * W : W;   // support capacity: the label words (+ two sibling statuses each)
    uint32_t* p = scratch + scr_off[cell];
    uint32_t* woff = p; p += K + 1;
    uint32_t* cum_g = p; p += K;
    uint32_t* cntb_g = p; p += K;
    float* inv_g = reinterpret_cast<float*>(p); p += K;
    uint32_t* sup = p; p += Ws;
    uint32_t* supc = p; p += Ws;
    uint32_t* widx = p; p += W;
    p += (p - scratch) & 1;   // 8-byte alignment
    uint64_t* pairs = reinterpret_cast<uint64_t*>(p); p += 2 * W;
    uint32_t* seg = p; p += Ws + 1;
    float* ain_g = reinterpret_cast<float*>(p); p += Ws;
    float* aout_g = reinterpret_cast<float*>(p); p += Ws;
    float* acc = reinterpret_cast<float*>(p);   // summary: sum[W], sq[W]; otherwise the replicates [B][S]
    uint32_t* sib_a = p;                         // INFER + USA (no sums then): support index of the sibling status(es)
    uint32_t* sib_b = p + Ws;
    // 1. label offsets, cumulative counts
    uint32_t wo = 0, N = 0, multi = 0;
    for (uint32_t base = 0; base < K; base += kBootNT) {
        const uint32_t k = base + tid;
        const uint32_t l = k < K ? len[k] : 0u, c = k < K ? cnt0[k] : 0u;
        uint32_t tl, tc;
        const uint32_t el = block_excl_scan<kBootNT>(l, s_ws, tl);
        const uint32_t ec = block_excl_scan<kBootNT>(c, s_ws, tc);
        if (k < K) { woff[k] = wo + el; cum_g[k] = N + ec + c; }
        multi |= l > 1;
        wo += tl; N += tc;
    }
    if (tid == 0) { woff[K] = wo; s_flag[0] = 0; }
    __syncthreads();
    if (multi) s_flag[0] = 1;
    // 2. possible support: the distinct gene ids, ascending
    for (uint32_t i = tid; i < W; i += kBootNT) {
        const uint32_t x = lab[i];
        sup[i] = x;
        if (usa) {   // prepare_support, em.rs:95-110
            const uint32_t a = x >= cfg.ao ? x - cfg.uo : (x >= cfg.uo ? x + cfg.uo : x + cfg.ao);
            sup[W + 2 * i] = a;
            sup[W + 2 * i + 1] = x >= cfg.ao ? x - cfg.ao : a;
        }
    }
    __syncthreads();
    const bool needs_em = s_flag[0] != 0;
    tiled_bitonic_sort_by<kBootNT, 8192>(sup, Ws, [](uint32_t a, uint32_t b) { return a > b; }, s_buf);
    uint32_t S = 0;
    for (uint32_t base = 0; base < Ws; base += kBootNT) {
        const uint32_t i = base + tid;
        const uint32_t h = i < Ws && (i == 0 || sup[i] != sup[i - 1]);
        uint32_t tot;
        const uint32_t ex = block_excl_scan<kBootNT>(h, s_ws, tot);
        if (h) supc[S + ex] = sup[i];
        S += tot;
    }
    __syncthreads();
    // 3. label word -> support index; (entry, class) pairs sorted: an entry's classes, ascending
    auto sup_index = [&](uint32_t g) -> uint32_t {
        uint32_t lo = 0, hi = S;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (supc[mid] < g) lo = mid + 1; else hi = mid; }
        return lo;
    };
    for (uint32_t i = tid; i < W; i += kBootNT) widx[i] = sup_index(lab[i]);
    if (usa)
        for (uint32_t s = tid; s < S; s += kBootNT) {   // entries that are labels have their siblings in the support; the others are never asked
            const uint32_t x = supc[s];
            uint32_t a = x >= cfg.ao ? x - cfg.uo : (x >= cfg.uo ? x + cfg.uo : x + cfg.ao);
            a = sup_index(a);
            sib_a[s] = a < S ? a : 0u;
            uint32_t b = 0xFFFFFFFFu;
            if (x >= cfg.ao) { b = sup_index(x - cfg.ao); b = b < S ? b : 0u; }
            sib_b[s] = b;
        }
    __syncthreads();
    for (uint32_t k = tid; k < K; k += kBootNT)
        for (uint32_t j = woff[k]; j < woff[k + 1]; ++j) pairs[j] = ((uint64_t)widx[j] << 32) | k;
    __syncthreads();
    tiled_bitonic_sort_by<kBootNT, 4096>(pairs, W, [](uint64_t a, uint64_t b) { return a > b; }, reinterpret_cast<uint64_t*>(s_buf));
    for (uint32_t s = tid; s <= S; s += kBootNT) {
        const uint64_t key = (uint64_t)s << 32;
        uint32_t lo = 0, hi = W;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (pairs[mid] < key) lo = mid + 1; else hi = mid; }
        seg[s] = lo;
    }
    uint32_t* pk = sup;   // dead since the dedupe: per sorted pair its class, bit 31 = single-label class
    for (uint32_t j = tid; j < W; j += kBootNT) {
        const uint32_t k = (uint32_t)pairs[j];
        pk[j] = k | ((woff[k + 1] - woff[k] == 1) ? 0x80000000u : 0u);
    }
    __syncthreads();
    // working arrays the rounds read at random (abundances, 1/denominators, counts): LDS when the cell fits, its scratch
    // otherwise; the new abundances are written and read back by the same thread only and stay in scratch
    const bool in_lds = K <= kBootLds && S <= kBootLds;
    uint32_t* cum = in_lds ? s_buf : cum_g;                                   // draws only; shares its space with inv
    float* inv = in_lds ? reinterpret_cast<float*>(s_buf) : inv_g;
    uint32_t* cntb = in_lds ? s_buf + kBootLds : cntb_g;
    float* ain = in_lds ? reinterpret_cast<float*>(s_buf + 2 * kBootLds) : ain_g;
    float* aout = aout_g;
    if (!INFER && cfg.summary_stat) for (uint32_t s = tid; s < 2 * S; s += kBootNT) acc[s] = 0.0f;
    // what a label contributes with: its own abundance, or (USA) the gene's as get_abundance_for adds it up
    auto abundance = [&](uint32_t s) -> float {
        if (!usa) return ain[s];
        const uint32_t b = sib_b[s];
        if (b != 0xFFFFFFFFu) return (ain[sib_a[s]] + ain[b]) + ain[s];   // ambiguous: unspliced + spliced + ambiguous
        return ain[sib_a[s]] + ain[s];                                     // unspliced / spliced: ambiguous + own
    };
    const uint64_t cell_index = cfg.first_cell_index + cell;
    const uint32_t k0 = (uint32_t)cfg.seed, k1 = (uint32_t)(cfg.seed >> 32), ci0 = (uint32_t)cell_index, ci1 = (uint32_t)(cell_index >> 32);
    for (uint32_t b = 0; b < cfg.B; ++b) {
        __syncthreads();
        // a. multinomial redraw of the class counts (infer: the counts as given)
        if (!INFER && in_lds) for (uint32_t k = tid; k < K; k += kBootNT) cum[k] = cum_g[k];
        for (uint32_t k = tid; k < K; k += kBootNT) cntb[k] = INFER ? cnt0[k] : 0u;
        __syncthreads();
        for (uint32_t q = tid; !INFER && q < (N + 3) / 4; q += kBootNT) {
            uint32_t w[4];
            philox4x32_10(q, b, ci0, ci1, k0, k1, w);
#pragma unroll
            for (uint32_t t = 0; t < 4; ++t) {
                if (4 * q + t < N) {
                    const uint32_t x = (uint32_t)(((uint64_t)w[t] * N) >> 32);
                    uint32_t lo = 0, hi = K;   // first class whose cumulative count exceeds x
                    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (cum[mid] <= x) lo = mid + 1; else hi = mid; }
                    atomicAdd(&cntb[lo], 1u);
                }
            }
        }
        __syncthreads();
        // b. EM from a random start (single-label counts only when nothing is ambiguous, em.rs:335-341)
        if (!needs_em) {
            for (uint32_t s = tid; s < S; s += kBootNT) {
                float a = 0.0f;
                for (uint32_t j = seg[s]; j < seg[s + 1]; ++j) a += (float)cntb[pk[j] & 0x7FFFFFFFu];
                ain[s] = a;
            }
        } else {
            if (INFER)
                for (uint32_t s = tid; s < S; s += kBootNT) {   // EmInitType::Informative, em.rs:376-378
                    float u = 0.0f;
                    for (uint32_t j = seg[s]; j < seg[s + 1]; ++j) if (pk[j] >> 31) u += (float)cntb[pk[j] & 0x7FFFFFFFu];
                    ain[s] = (u + 0.5f) * 1e-3f;
                }
            for (uint32_t q = tid; !INFER && q < (S + 3) / 4; q += kBootNT) {
                uint32_t w[4];
                philox4x32_10(q, b | 0x80000000u, ci0, ci1, k0, k1, w);
#pragma unroll
                for (uint32_t t = 0; t < 4; ++t) if (4 * q + t < S) ain[4 * q + t] = (float)(w[t] >> 8) * (1.0f / 16777216.0f) + 1e-5f;
            }
            uint32_t it = 0;
            bool conv = true, last_round = false;
            while (it < kMinIter || (it < kMaxIter && !conv) || last_round) {
                __syncthreads();
                if (tid == 0) s_flag[1] = 0;
                for (uint32_t k = tid; k < K; k += kBootNT) {   // (A) inv_denominator of the multi-label classes
                    const uint32_t a = woff[k], e = woff[k + 1];
                    float v = -1.0f;                              // single label, or denominator 0: no share
                    if (e - a > 1) {
                        float den = 0.0f;
                        for (uint32_t j = a; j < e; ++j) den += abundance(widx[j]);
                        if (den > 0.0f) v = (float)cntb[k] / den;
                    }
                    inv[k] = v;
                }
                __syncthreads();
                bool bad = false;
                for (uint32_t sb = 0; sb < S; sb += kBootNT) {   // (B) alphas_out, classes in order (wave-uniform trip count)
                    const uint32_t s = sb + tid;
                    const bool valid = s < S;
                    const float ai = valid ? ain[s] : 0.0f;
                    const uint32_t q0 = valid ? seg[s] : 0u, q1 = valid ? seg[s + 1] : 0u;
                    const float ab = (valid && q1 > q0) ? abundance(s) : 0.0f;   // the share's numerator (own abundance outside USA)
                    auto term_at = [&](float a, uint32_t q) -> float {   // a skipped share is +0.0f: leaves the non-negative sum unchanged
                        const uint32_t e = pk[q], k = e & 0x7FFFFFFFu;
                        if (e >> 31) return (float)cntb[k];
                        const float iv = inv[k];
                        return iv >= 0.0f ? a * iv : 0.0f;
                    };
                    const bool heavy = valid && q1 - q0 > kBootHeavy;
                    float o = 0.0f;
                    if (valid && !heavy) for (uint32_t q = q0; q < q1; ++q) o += term_at(ab, q);
                    // an entry in many classes (a highly expressed gene) is summed by its whole wave: the 64 loads go out
                    // together, the additions stay one after the other in class order
                    for (uint64_t hm = __ballot(heavy); hm; hm &= hm - 1) {
                        const uint32_t L = (uint32_t)__builtin_ctzll(hm);
                        const float a_l = bcast_f32(ab, L);
                        const uint32_t b0 = bcast_u32(q0, L), b1 = bcast_u32(q1, L);
                        float r = 0.0f;
                        for (uint32_t base = b0; base < b1; base += 64) {
                            const uint32_t q = base + lane_id();
                            const uint32_t tb = __float_as_uint(q < b1 ? term_at(a_l, q) : 0.0f);
#pragma unroll
                            for (int i = 0; i < 64; ++i) r += __uint_as_float(__builtin_amdgcn_readlane(tb, i));
                        }
                        if (lane_id() == L) o = r;
                    }
                    if (valid) {
                        if (o > kAlphaCheckCutoff && fabsf(ai - o) > kRelDiffTol) bad = true;
                        aout[s] = o;
                    }
                }
                if (bad) s_flag[1] = 1;
                __syncthreads();
                conv = s_flag[1] == 0;
                ++it;
                const bool floor_now = !last_round && it >= kMinIter && conv;
                for (uint32_t s = tid; s < S; s += kBootNT) { const float o = aout[s]; ain[s] = (floor_now && o < kMinOutputAlpha) ? 0.0f : o; }
                if (last_round) break;
                if (floor_now) last_round = true;
            }
            __syncthreads();
            for (uint32_t s = tid; s < S; s += kBootNT) if (ain[s] < kMinOutputAlpha) ain[s] = 0.0f;
        }
        __syncthreads();
        // c. running sums (em.rs:662-666) or the replicate itself
        for (uint32_t s = tid; !INFER && s < S; s += kBootNT) {
            const float a = ain[s];
            if (cfg.summary_stat) { acc[s] += a; acc[S + s] += a * a; }
            else acc[(uint64_t)b * S + s] = a;
        }
    }
    __syncthreads();
    if (INFER) {
        for (uint32_t s = tid; s < S; s += kBootNT) { o_col[w0 * (usa ? 3 : 1) + s] = supc[s]; o_mean[w0 * (usa ? 3 : 1) + s] = ain[s]; }
        if (tid == 0) n_support[cell] = S;
        return;
    }
    // mean / variance per support entry (em.rs:673-683 | quant.rs:185-210); the host keeps the non-zero ones
    const float n = (float)cfg.B;
    for (uint32_t s = tid; s < S; s += kBootNT) {
        float mean, var = 0.0f;
        if (cfg.summary_stat) {
            mean = acc[s] / n;
            var = (acc[S + s] / n) - (mean * mean);
        } else {
            float s1 = 0.0f;
            for (uint32_t b = 0; b < cfg.B; ++b) s1 += acc[(uint64_t)b * S + s];
            mean = s1 / n;
            if (mean != 0.0f) {
                float s2 = 0.0f;
                for (uint32_t b = 0; b < cfg.B; ++b) { const float d = acc[(uint64_t)b * S + s] - mean; s2 += d * d; }
                var = s2 / fmaxf(n - 1.0f, 1.0f);
            }
        }
        o_col[w0 + s] = supc[s]; o_mean[w0 + s] = mean; o_var[w0 + s] = var;
    }
    if (tid == 0) n_support[cell] = S;
}

__global__ __launch_bounds__(256) void k_boot_compact(uint32_t n_cells, const uint64_t* __restrict__ word_ptr, const uint64_t* __restrict__ sup_ptr,
                                                     const uint32_t* __restrict__ i_col, const float* __restrict__ i_mean, const float* __restrict__ i_var,
                                                     uint32_t* __restrict__ o_col, float* __restrict__ o_mean, float* __restrict__ o_var) {
    const uint32_t cell = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (cell >= n_cells) return;
    const uint64_t src = word_ptr[cell], dst = sup_ptr[cell];
    const uint32_t S = (uint32_t)(sup_ptr[cell + 1] - dst);
    for (uint32_t i = lane_id(); i < S; i += 64) { o_col[dst + i] = i_col[src + i]; o_mean[dst + i] = i_mean[src + i]; o_var[dst + i] = i_var[src + i]; }
}

void launch_boot_compact(hipStream_t s, uint32_t n_cells, const uint64_t* word_ptr, const uint64_t* sup_ptr, const uint32_t* i_col,
                         const float* i_mean, const float* i_var, uint32_t* o_col, float* o_mean, float* o_var) {
    if (!n_cells) return;
    AFQ_LAUNCH(k_boot_compact, (n_cells + 3) / 4, 256, s, n_cells, word_ptr, sup_ptr, i_col, i_mean, i_var, o_col, o_mean, o_var);
}

void launch_boot(hipStream_t s, uint32_t n_cells, const uint64_t* cls_ptr, const uint64_t* word_ptr, const uint32_t* len,
                 const uint32_t* cnt, const uint32_t* lab, const uint64_t* scr_off, uint32_t* scratch, uint32_t B, uint32_t summary_stat,
                 uint64_t seed, uint64_t first_cell_index, uint32_t* n_support, uint32_t* o_col, float* o_mean, float* o_var) {
    if (!n_cells) return;
    BootCfg cfg{B, summary_stat, seed, first_cell_index, 0u, 0u, 0u, 0u};
    AFQ_LAUNCH(k_boot<false>, n_cells, kBootNT, s, cls_ptr, word_ptr, len, cnt, lab, scr_off, scratch, cfg, n_support, o_col, o_mean, o_var);
}

// per-cell scratch words of the infer kernel for K classes with W label words
uint64_t infer_scratch_words(uint64_t K, uint64_t W, bool usa) {
    const uint64_t Ws = usa ? 3 * W : W;
    return (K + 1) + 3 * K + 2 * Ws + W + 2 * W + (Ws + 1) + 2 * Ws + 2 * Ws + 8;
}

void launch_infer(hipStream_t s, uint32_t n_cells, const uint64_t* cls_ptr, const uint64_t* word_ptr, const uint32_t* len,
                  const uint32_t* cnt, const uint32_t* lab, const uint64_t* scr_off, uint32_t* scratch, uint32_t usa, uint32_t num_alphas,
                  uint32_t* n_support, uint32_t* o_col, float* o_alpha) {
    if (!n_cells) return;
    BootCfg cfg{1u, 0u, 0ull, 0ull, usa, num_alphas / 3, (2 * num_alphas) / 3, 0u};   // usa_offsets of infer.rs:107-111
    AFQ_LAUNCH(k_boot<true>, n_cells, kBootNT, s, cls_ptr, word_ptr, len, cnt, lab, scr_off, scratch, cfg, n_support, o_col, o_alpha, o_alpha);
}

// words of per-cell EM scratch for nU single-label columns, W label words, M ambiguous molecules
uint64_t em_scratch_words(uint32_t nU, uint32_t W, uint32_t M, bool usa) {
    const uint64_t capS = ((uint64_t)nU + W) * (usa ? 3u : 1u);
    // mirrors the carve at the top of k_em
    uint64_t w = 2 * (capS + 1)            // out
                 + 2 * ((uint64_t)W + 1)   // inv_pairs
                 + 3 * ((uint64_t)M + 1)   // order, cls_first, cls_cnt
                 + ((uint64_t)M + 2)       // cls_woff
                 + 2 * ((uint64_t)W + 1)   // cls_w, cls_sidx
                 + ((uint64_t)M + 1)       // inv
                 + 4 * (capS + 1)          // support, sib1, sib2, ucnt
                 + 2 * (capS + 2)          // a_in, a_out
                 + (capS + 2)              // slot_off
                 + (capS + 1)              // aid
                 + 4 * ((uint64_t)nU + W + 2)  // ent
                 + 4 * ((uint64_t)W + 1)   // lw3
                 + ((uint64_t)nU + W + 2)  // act_col
                 + ((uint64_t)W + 1)       // memb
                 + 4;                      // alignment slack for the 16-byte records
    return (w + 3) & ~3ull;  // keep slices 16-byte aligned
}

void launch_em(hipStream_t s, const ResolveArgs& a, uint32_t n_cells, const uint64_t* em_off, uint32_t* scratch,
               uint32_t* out_nnz, void* em_hdr_v, const uint32_t* em_order, uint32_t num_alphas, uint32_t init_uniform, bool rounds) {
    uint4* em_hdr = reinterpret_cast<uint4*>(em_hdr_v);
    if (!n_cells) return;
    EmCfg cfg{a.usa, num_alphas, a.num_rows / 3, 2 * (a.num_rows / 3), init_uniform};
    AFQ_LAUNCH(k_em, n_cells, kEmNT, s, a.meta, a.nnz, a.keys0, a.keys1, a.lab, a.lab_cnt, em_off, scratch, out_nnz, em_hdr, em_order, cfg);
    if (rounds) AFQ_LAUNCH(k_em_rounds, n_cells, kEmRNT, s, a.meta, a.nnz, a.lab_cnt, em_off, scratch, out_nnz, em_hdr, em_order, cfg);
}

void launch_eqc_dump(hipStream_t s, const ResolveArgs& a, uint32_t n_cells, const uint64_t* em_off, const uint32_t* scratch,
                     const void* em_hdr, uint32_t num_alphas, uint32_t* n_cls, uint32_t* n_words, const uint64_t* cls_ptr,
                     const uint64_t* word_ptr, uint32_t* o_len, uint32_t* o_count, uint32_t* o_labels) {
    if (!n_cells) return;
    EmCfg cfg{a.usa, num_alphas, a.num_rows / 3, 2 * (a.num_rows / 3), 0u};
    AFQ_LAUNCH(k_eqc_dump, (n_cells + 3) / 4, 256, s, n_cells, a.meta, a.nnz, a.keys0, a.keys1, a.lab, a.lab_cnt, em_off, scratch,
               reinterpret_cast<const uint4*>(em_hdr), cfg, n_cls, n_words, cls_ptr, word_ptr, o_len, o_count, o_labels);
}

void launch_compact_em(hipStream_t s, uint32_t n_cells, const uint64_t* em_off, const uint32_t* scratch, const uint32_t* nnz,
                       const uint64_t* cell_ptr, uint32_t* gene, float* val) {
    if (!n_cells) return;
    AFQ_LAUNCH(k_compact_em, (n_cells + 3) / 4, 256, s, n_cells, em_off, scratch, nnz, cell_ptr, gene, val);
}

}  // namespace afq
