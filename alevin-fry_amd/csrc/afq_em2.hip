// afq_em2.hip - the per-cell EM (src/em.rs) with ORDER-FREE arithmetic, round 4.
//
// What the reference computes per round (em_update_subset[_usa], em.rs:189-248; em_update, em.rs:458-485): for every
// gene-level class with label l and count n: D = sum_{g in l} abundance(g); every g in l receives abundance(g) * (n / D);
// a single-label class adds n to its one entry.  The f32 additions into an entry happen in HashMap order (em.rs:464),
// i.e. in no order at all: run to run the reference's own low bits move (DESIGN.md section 5 measures how far).
//
// Here the shares are accumulated in 64-bit FIXED POINT, which makes the sum independent of any order:
//     r = 1.0f / D                       (f32, D summed in label order as the reference does)
//     q = (u64)((abundance(g) * r) * 2^F)  (f32 product, scaled by a power of two - exact -, truncated)
//     acc[g] += n * q                    (integers: associative, so classes may come in any order, from any lane, atomically)
//     alpha'[g] = (float)acc[g] * 2^-F   (acc starts at the entry's single-label count << F; one rounding)
// with F = min(40, 62 - bitlen(nrec)) so that acc < 2^63.  Two consequences shape the kernels:
//   * n * q is linear in n: a class of count n and n copies of it with count 1 add the same integer.  The device therefore
//     does NOT group equal labels (no sort, no hash table): every ambiguous molecule is its own class of count 1.  The oracle
//     (oracle/afq_oracle.cpp, ora_set_em_arith(1)) groups them as the reference does and gets the same bits.
//   * no inverted index entry -> classes, no serial chains for highly expressed genes: a round is ONE pass over the classes
//     (gather the label's abundances, D, r, scatter the shares with LDS 64-bit atomics) and one pass over the entries.
// Against the reference's f32 arithmetic in the oracle's canonical order the results differ like any reordering does
// (<= 1e-4 relative, north_star's tolerance; tests/test_gpu_em.py); against the oracle in the same arithmetic they are
// bit-identical.  AFQ_EM_ORDER=canonical keeps the round-1..3 kernels (afq_em.hip: sequential f32 in canonical class order).
//
// Only entries that sit in some class label ("live") change from round to round.  An entry with a single-label count that
// is in no label holds that count from round 1 on; it matters only as a USA sibling (get_abundance_for, em.rs:167-187) of a
// live entry ("passive": a constant after round 1) and in the output row.  The rounds run over live entries only.
//
//   k_em2_setup   per cell: EM labels (USA rewrite utils.rs:865-925) -> live set by bitmap + popcount ranks, sibling links,
//                 passive siblings, single-label counts, the merge ranks of the output row; picks the cell's tier
//   k_em2_rounds  per cell, five instances: everything in LDS at 256 / 512 / 1024 threads (38 / 78 / 157 KiB: four, two, one
//                 cell per CU), abundances + accumulators in LDS with the class lists streamed, everything in global memory
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#include "afq_common.h"
#include "afq_hooks.h"
#include "afq_kernels.h"
#include "afq_prims.h"

namespace afq {

namespace {
constexpr float kMinOutputAlpha2 = 0.01f, kAlphaCheckCutoff2 = 1e-2f, kRelDiffTol2 = 1e-2f;
constexpr uint32_t kMinIter2 = 2, kMaxIter2 = 100;
constexpr uint32_t kSibNone = 0xFFFFFFFFu;   // no such sibling status: adds +0.0f
constexpr uint32_t kSibZero = 0xFFFFFFFEu;   // a sibling status nothing maps to: the start value in round 1, 0 afterwards
constexpr uint32_t kSibPassive = 0x80000000u;   // | index into pas_val
constexpr int kSetupNT = 256;
#ifndef AFQ_EM2_PAIRS_PASS
#define AFQ_EM2_PAIRS_PASS 1
#endif
#ifndef AFQ_EM2_CPT_STREAM
#define AFQ_EM2_CPT_STREAM 4
#endif
constexpr int kCptS = AFQ_EM2_CPT_STREAM;

struct Em2Cfg { uint32_t usa, num_alphas, uo, ao, init_uniform, nwb, min_tier, force_wide; };

// per-cell scratch (u32 words); mirrored by em2_scratch_words
struct Em2Scratch {
    uint2* out; uint32_t* hdr; uint32_t *ent_col, *ent_ucnt, *ent_s1, *ent_s2, *ent_ub, *pas_val, *coff, *cw, *pu_col, *pu_cnt, *pu_lb;
    unsigned long long* g_acc; float *g_ab, *g_v; uint32_t *g_pre, *nid; uint16_t* cw16;
};
// header: 16 scalar words, then the class runs (up to 64 of {first class, first word, label length (0: the run of 63+, offsets in
// coff), one past the last class})
constexpr uint32_t kHdrWords = 16 + 4 * 64;
__host__ __device__ inline uint64_t em2_pas_cap(uint64_t nU, uint64_t W, bool usa) { return usa ? (nU < 2 * W ? nU : 2 * W) : 0; }
__device__ __forceinline__ Em2Scratch em2_carve(uint32_t* scratch, uint64_t off, uint32_t nU, uint32_t W, uint32_t M, bool usa) {
    Em2Scratch e;
    uint32_t* p = scratch + off;
    e.out = reinterpret_cast<uint2*>(p); p += 2 * ((uint64_t)nU + W);
    e.hdr = p; p += kHdrWords;
    e.g_acc = reinterpret_cast<unsigned long long*>(p); p += 2 * ((uint64_t)W + 1);
    e.ent_col = p; p += W;
    e.ent_ucnt = p; p += W;
    e.ent_s1 = p; p += usa ? W : 0;
    e.ent_s2 = p; p += usa ? W : 0;
    e.ent_ub = p; p += W;
    e.pas_val = p; p += em2_pas_cap(nU, W, usa);
    e.coff = p; p += M + 1;
    e.cw = p; p += W;
    e.pu_col = p; p += nU;
    e.pu_cnt = p; p += nU;
    e.pu_lb = p; p += nU;
    e.g_ab = reinterpret_cast<float*>(p); p += usa ? W : 0;
    e.g_v = reinterpret_cast<float*>(p); p += W + em2_pas_cap(nU, W, usa) + 2;
    e.g_pre = p; p += W + 1;
    e.nid = p; p += W;
    e.cw16 = reinterpret_cast<uint16_t*>(p); p += (W + 1) / 2 + 1;
    return e;
}
}  // namespace

__host__ __device__ inline uint64_t em2_words(uint32_t nU, uint32_t W, uint32_t M, bool usa) {
    const uint64_t pc = em2_pas_cap(nU, W, usa);
    uint64_t w = 2 * ((uint64_t)nU + W) + kHdrWords + 2 * ((uint64_t)W + 1) + 3 * (uint64_t)W + (usa ? 2 * (uint64_t)W : 0) + pc + ((uint64_t)M + 1) + W +
                 3 * (uint64_t)nU + (usa ? W : 0) + ((uint64_t)W + pc + 2) + ((uint64_t)W + 1) + W + (((uint64_t)W + 1) / 2 + 1);
    return (w + 3) & ~3ull;   // slices stay 16-byte aligned
}
uint64_t em2_scratch_words(uint32_t nU, uint32_t W, uint32_t M, bool usa) { return em2_words(nU, W, M, usa); }

// header words
enum { H_L = 0, H_P, H_K, H_WC, H_NPU, H_FBITS, H_TIER, H_NHOT, H_NRUN, H_NARROW, H_C4, H_C2 };
constexpr uint32_t kTierNone = 7;   // no multi-label class: the row is the single-label counts (em.rs:339-341, 499-514)
// LDS words of the three all-in-LDS instances (4 x 38 KiB, 2 x 78 KiB, 157 KiB: next to the few static words they fit a CU's 160 KiB)
constexpr uint32_t kT0Words = 9728, kT1Words = 19968, kT2Words = 40192;

__device__ __forceinline__ uint32_t em2_fbits(uint32_t nrec) {
    const uint32_t bl = 32u - (uint32_t)__builtin_clz(nrec | 1u);
    const uint32_t f = 62u - bl;
    return f < 40u ? f : 40u;
}
// LDS words of the all-in-LDS layout: acc u64[L] | ab f32[L] (USA) | v f32[L+P+2] | coff u16[K+1] | cw u16[Wc]
__host__ __device__ inline uint32_t em2_lds_core_words(uint32_t L, uint32_t P, bool usa) { return 2 * L + (usa ? L : 0) + (L + P + 2); }
// entries whose state the largest instance keeps in LDS when a cell does not fit whole (acc u64 + ab f32 (USA) + v f32 each)
__host__ __device__ inline uint32_t em2_hot_cap(bool usa) { return kT2Words / (usa ? 4u : 3u); }
__host__ __device__ inline uint32_t em2_lds_all_words(uint32_t L, uint32_t P, uint32_t K, uint32_t Wc, bool usa) {
    return em2_lds_core_words(L, P, usa) + (K + 2) / 2 + (Wc + 1) / 2 + 2;
}

__device__ __forceinline__ uint32_t em2_map_sib(uint32_t s, uint32_t L, uint32_t P) {
    if (s == kSibNone) return L + P + 1;
    if (s == kSibZero) return L + P;
    if (s & kSibPassive) return L + (s & 0x7FFFFFFFu);
    return s;
}

// The barrier between phases that hand each other data through GLOBAL memory: a wave first waits for its own stores to be
// acknowledged (stores count in vmcnt on gfx9), then goes to the barrier; a word the other waves change with atomics (L2) is
// read back with an atomic as well (afq_pug2.hip has the measurements behind both rules).
__device__ __forceinline__ void em2_gsync() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }

// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kSetupNT) void k_em2_setup(const CellMeta* __restrict__ meta, const uint32_t* __restrict__ nnz_unique,
                                                        const uint64_t* __restrict__ keys0, const uint64_t* __restrict__ keys1,
                                                        const uint32_t* __restrict__ lab, const uint32_t* __restrict__ lab_cnt,
                                                        const uint64_t* __restrict__ em_off, uint32_t* __restrict__ scratch,
                                                        uint32_t* __restrict__ out_nnz, const uint32_t* __restrict__ em_order,
                                                        uint32_t* __restrict__ tiers /* [8] counters, then 5 lists of n_cells */, uint32_t n_cells,
                                                        Em2Cfg cfg) {
    extern __shared__ uint32_t s_bm[];   // bits[nwb], rank[nwb]
    __shared__ uint32_t s_ws[kSetupNT / 64];
    __shared__ uint32_t s_P;
    __shared__ uint32_t s_hist[64], s_pick[3], s_len_cur[64], s_len_cbase[64], s_len_wbase[64], s_long_words, s_len_cur63w;
    constexpr int NT = kSetupNT;
    const uint32_t tid = threadIdx.x;
    if (tiers[7]) return;   // (the scratch the host set aside is too small: it sizes the EM itself)
    const uint32_t cell = em_order[blockIdx.x];
    const CellMeta m = meta[cell];
    const uint32_t nU = nnz_unique[cell];
    const uint2* U = reinterpret_cast<const uint2*>(((m.lg_nb || mode_is_pug(m.mode)) ? keys1 : keys0) + m.key_off);
    const uint32_t W = lab_cnt[2 * cell], M = lab_cnt[2 * cell + 1];
    const uint32_t* lw = lab + 2 * m.key_off;
    const uint32_t* ld = lw + m.n_ref + 1;
    const bool usa = cfg.usa != 0;
    const Em2Scratch sc = em2_carve(scratch, em_off[cell], nU, W, M, usa);
    if (M == 0) {
        for (uint32_t i = tid; i < nU; i += NT) sc.out[i] = make_uint2(U[i].x, __float_as_uint((float)U[i].y));
        if (tid == 0) { out_nnz[cell] = nU; sc.hdr[H_TIER] = kTierNone; }
        return;
    }
    uint32_t* bits = s_bm;
    uint32_t* rank = s_bm + cfg.nwb;
    for (uint32_t i = tid; i < cfg.nwb; i += NT) bits[i] = 0;
    if (tid == 0) s_P = 0;
    // 1. EM label of every ambiguous molecule (extract_usa_eqmap, utils.rs:865-925: S alone -> g>>1, U alone -> uo + (g>>1),
    //    an adjacent S,U pair -> ao + (g>>1)); offsets by a scan of the lengths
    auto em_label = [&](uint32_t mol, uint32_t* dst) -> uint32_t {   // returns the length; writes (and marks the live bitmap) when dst != null
        const uint32_t o = ld[2 * mol], n = ld[2 * mol + 1];
        uint32_t w = 0;
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t gn = lw[o + i];
            uint32_t idx = gn;
            if (usa) {
                idx = gn >> 1;
                if (is_spliced(gn)) {
                    if (i + 1 < n && same_gene(gn, lw[o + i + 1])) { idx += cfg.ao; ++i; }
                } else idx += cfg.uo;
            }
            if (dst) { dst[w] = idx; atomicOr(&bits[idx >> 5], 1u << (idx & 31)); }
            ++w;
        }
        return w;
    };
    // The classes are laid out LONGEST LABEL FIRST (their order is free: every sum over classes is an integer sum).  A thread of
    // the rounds takes one class and walks its label; with the classes as they come, a wave walks as far as its longest label
    // (gene families: labels of 5, 10, 30 genes next to the usual pairs - the rounds' class pass then ran at the pace of the
    // longest of 64 labels, every step a dependent load).  Sorted by length a wave's labels are the same length.  Counting sort:
    // classes of one length sit in one run, so a class's words are at run base + rank x length and no per-class scan is needed
    // (lengths of 63 and more share a run and take their words from a cursor).
    if (tid < 64) { s_hist[tid] = 0; s_len_cur[tid] = 0; }
    if (tid == 0) { s_long_words = 0; s_len_cur63w = 0; }
    __syncthreads();
    for (uint32_t i = tid; i < M; i += NT) {
        const uint32_t len = usa ? em_label(i, nullptr) : ld[2 * i + 1];
        atomicAdd(&s_hist[len < 63u ? len : 63u], 1u);
        if (len >= 63u) atomicAdd(&s_long_words, len);
    }
    __syncthreads();
    if (tid == 0) {   // runs from the longest labels down: class positions and word offsets
        uint32_t cpos = 0, wpos = 0;
        for (int b = 63; b >= 0; --b) {
            s_len_cbase[b] = cpos; s_len_wbase[b] = wpos;
            cpos += s_hist[b];
            wpos += b == 63 ? s_long_words : (uint32_t)b * s_hist[b];
        }
        s_long_words = wpos;   // (now: all label words)
        uint32_t r = 0;   // the runs as the rounds read them: a class's words are at first word + (class - first class) x length
        for (int b = 63; b >= 0; --b)
            if (s_hist[b]) {
                uint32_t* q = sc.hdr + 16 + 4 * r++;
                q[0] = s_len_cbase[b]; q[1] = s_len_wbase[b]; q[2] = b == 63 ? 0u : (uint32_t)b; q[3] = s_len_cbase[b] + s_hist[b];
            }
        sc.hdr[H_NRUN] = r;
        sc.hdr[H_C4] = s_len_cbase[4];   // classes before this one have labels of more than four words
        sc.hdr[H_C2] = s_len_cbase[2];   // ... of more than two
    }
    __syncthreads();
    const uint32_t Wc = s_long_words;
    for (uint32_t i = tid; i < M; i += NT) {
        const uint32_t len = usa ? em_label(i, nullptr) : ld[2 * i + 1];
        const uint32_t b = len < 63u ? len : 63u;
        const uint32_t rk = atomicAdd(&s_len_cur[b], 1u);
        const uint32_t off = s_len_wbase[b] + (b < 63u ? rk * len : atomicAdd(&s_len_cur63w, len));
        sc.coff[s_len_cbase[b] + rk] = off;
        em_label(i, sc.cw + off);
    }
    if (tid == 0) sc.coff[M] = Wc;
    __syncthreads();
    if (s_hist[63]) {   // the 63+ run took its words in arrival order: its offsets must ascend with the class index (coff[c + 1] ends class c)
        // (rare: re-number the run's classes by offset - a serial pass of one thread over a handful of classes)
        if (tid == 0) {
            const uint32_t c0 = s_len_cbase[63], n63 = s_hist[63];
            for (uint32_t a = 1; a < n63; ++a) {   // insertion sort by offset
                const uint32_t v = sc.coff[c0 + a];
                uint32_t j = a;
                for (; j > 0 && sc.coff[c0 + j - 1] > v; --j) sc.coff[c0 + j] = sc.coff[c0 + j - 1];
                sc.coff[c0 + j] = v;
            }
        }
        __syncthreads();
    }
    // 2. live entries = distinct label slots: prefix popcount over the bitmap (ids ascend with the column)
    uint32_t L = 0;
    constexpr uint32_t kSc = 8;
    for (uint32_t base = 0; base < cfg.nwb; base += kSc * NT) {
        const uint32_t w0 = base + kSc * tid;
        uint32_t v[kSc], sum = 0;
#pragma unroll
        for (uint32_t j = 0; j < kSc; ++j) { v[j] = w0 + j < cfg.nwb ? (uint32_t)__popc(bits[w0 + j]) : 0u; }
#pragma unroll
        for (uint32_t j = 0; j < kSc; ++j) { const uint32_t t = v[j]; v[j] = sum; sum += t; }
        uint32_t tot;
        const uint32_t ex = block_excl_scan<NT>(sum, s_ws, tot);
#pragma unroll
        for (uint32_t j = 0; j < kSc; ++j) if (w0 + j < cfg.nwb) rank[w0 + j] = L + ex + v[j];
        L += tot;
    }
    __syncthreads();
    auto is_live = [&](uint32_t x) -> bool { return (bits[x >> 5] >> (x & 31)) & 1u; };
    auto rank_of = [&](uint32_t x) -> uint32_t { return rank[x >> 5] + (uint32_t)__popc(bits[x >> 5] & ((1u << (x & 31)) - 1u)); };
    // 3. per live entry: column, sibling statuses (get_abundance_for, em.rs:167-187: S and U lean on the gene's A; A on U and S)
    for (uint32_t w = tid; w < cfg.nwb; w += NT) {
        uint32_t b = bits[w], e = rank[w];
        for (; b; b &= b - 1, ++e) {
            const uint32_t x = (w << 5) + (uint32_t)__builtin_ctz(b);
            sc.ent_col[e] = x; sc.ent_ucnt[e] = 0; sc.ent_ub[e] = 0;
            if (usa) {
                uint32_t c1, c2 = kSibNone;
                if (x >= cfg.ao) { c1 = x - cfg.uo; c2 = x - cfg.ao; }
                else if (x >= cfg.uo) c1 = x + cfg.uo;
                else c1 = x + cfg.ao;
                sc.ent_s1[e] = is_live(c1) ? rank_of(c1) : kSibZero;
                sc.ent_s2[e] = c2 == kSibNone ? kSibNone : (is_live(c2) ? rank_of(c2) : kSibZero);
            }
        }
    }
    __syncthreads();
    // 4. the single-label counts: of a live entry -> its count; otherwise the column goes straight to the output row (and, USA,
    //    becomes a passive sibling of the live entries that read it)
    uint32_t nPU = 0;
    for (uint32_t base = 0; base < nU; base += NT) {
        const uint32_t i = base + tid;
        uint32_t col = 0, cnt = 0, flag = 0;
        if (i < nU) {
            const uint2 u = U[i];
            col = u.x; cnt = u.y;
            if (is_live(col)) sc.ent_ucnt[rank_of(col)] = cnt;
            else {
                flag = 1;
                if (usa) {
                    uint32_t n1 = kSibNone, n2 = kSibNone;   // columns whose abundance reads this one, and through which link
                    bool second = false;
                    if (col >= cfg.ao) { n1 = col - cfg.ao; n2 = col - cfg.uo; }          // A: read by the gene's S and U (their first link)
                    else if (col >= cfg.uo) n1 = col + cfg.uo;                           // U: read by A (first link)
                    else { n1 = col + cfg.ao; second = true; }                           // S: read by A (second link)
                    const bool l1 = is_live(n1), l2 = n2 != kSibNone && is_live(n2);
                    if (l1 || l2) {
                        const uint32_t pid = atomicAdd(&s_P, 1u);
                        sc.pas_val[pid] = cnt;
                        if (l1) (second ? sc.ent_s2 : sc.ent_s1)[rank_of(n1)] = kSibPassive | pid;
                        if (l2) sc.ent_s1[rank_of(n2)] = kSibPassive | pid;
                    }
                }
            }
        }
        uint32_t tot;
        const uint32_t ex = block_excl_scan<NT>(flag, s_ws, tot);
        if (flag) { const uint32_t j = nPU + ex; sc.pu_col[j] = col; sc.pu_cnt[j] = cnt; sc.pu_lb[j] = rank_of(col); }
        nPU += tot;
    }
    __syncthreads();
    // 4b. where a live entry lands in the row: behind ent_ub[e] pass-through columns (a step function of e, written by the steps)
    for (uint32_t j = tid; j < nPU; j += NT) {
        const uint32_t lo = sc.pu_lb[j], hi = j + 1 < nPU ? sc.pu_lb[j + 1] : L;
        for (uint32_t e = lo; e < hi; ++e) sc.ent_ub[e] = j + 1;
    }
    // 5. tier (every thread: the values are uniform)
    const uint32_t P = s_P, K = M;
    const bool narrow = L + P + 2 <= 65536u;   // every state id (entries, passive siblings, the two constants) fits 16 bits
    uint32_t tier;
    {
        const bool ids16 = narrow && K < 65535u && Wc <= 65535u;
        const uint32_t all = em2_lds_all_words(L, P, K, Wc, usa), core = em2_lds_core_words(L, P, usa);
        if (ids16 && all <= kT0Words && L <= 256u * 8u) tier = 0;
        else if (ids16 && all <= kT1Words && L <= 512u * 8u) tier = 1;
        else if (ids16 && all <= kT2Words && L <= 1024u * 16u) tier = 2;
        else if (core <= kT2Words) tier = 3;   // (at most ~13 000 entries: always narrow)
        else tier = 4;
        if (tier < cfg.min_tier) tier = cfg.min_tier;   // (tests: every instance on every size)
    }
    // 6. label words as live ids.  Tiers 3 and 4 stream the class lists from memory every round and are bound by those bytes
    //    (round 5: 4.3-4.6 TB/s in every phase of a round), so they get the words as 16-bit ids next to the 32-bit ones.
    for (uint32_t w = tid; w < Wc; w += NT) {
        const uint32_t id = rank_of(sc.cw[w]);
        sc.cw[w] = id;
        if (tier == 3) sc.cw16[w] = (uint16_t)id;
    }
    uint32_t H = L;
    const bool narrow4 = narrow && !cfg.force_wide;   // (tests: the 32-bit route of the largest instance on small cells)
    if (tier == 4) {
        // 7. A cell whose state does not fit LDS keeps the entries that take most of the traffic there - the ones in most
        // classes (gene popularity is Zipf: a few entries sit in a tenth of all labels) - and the rest in global memory:
        // entries are renumbered hot first (state id = nid[entry]; entries stay in column order for the output row).
        uint32_t* deg = sc.g_pre;
        em2_gsync();   // (label words as live ids, written above by other threads)
        for (uint32_t e = tid; e < L; e += NT) deg[e] = 0;
        if (tid < 64) s_hist[tid] = 0;
        em2_gsync();
        for (uint32_t w = tid; w < Wc; w += NT) atomicAdd(&deg[sc.cw[w]], 1u);
        em2_gsync();
        for (uint32_t e = tid; e < L; e += NT) { const uint32_t d = __hip_atomic_load(&deg[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); atomicAdd(&s_hist[d < 63u ? d : 63u], 1u); }
        em2_gsync();
        if (tid == 0) {   // every degree above t is hot; of degree t, the first `extra` entries
            const uint32_t cap = cfg.min_tier == 4 && (L + 1) / 2 < em2_hot_cap(usa) ? (L + 1) / 2 : em2_hot_cap(usa);   // (tests force small cells here: half of them cold)
            uint32_t t = 63, n = 0;
            while (t > 0 && n + s_hist[t] <= cap) { n += s_hist[t]; --t; }
            const uint32_t extra = t > 0 ? (cap - n < s_hist[t] ? cap - n : s_hist[t]) : 0u;
            s_pick[0] = t; s_pick[1] = extra; s_pick[2] = n + extra;
        }
        em2_gsync();
        const uint32_t t = s_pick[0], extra = s_pick[1];
        H = s_pick[2];
        uint32_t eq_before = 0, hot_before = 0;
        for (uint32_t base = 0; base < L; base += NT) {
            const uint32_t e = base + tid;
            uint32_t d = 0;
            if (e < L) { d = __hip_atomic_load(&deg[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); d = d < 63u ? d : 63u; }
            const uint32_t eq = e < L && d == t && t > 0;
            uint32_t tot_eq, tot_hot;
            const uint32_t eqr = eq_before + block_excl_scan<NT>(eq, s_ws, tot_eq);
            const uint32_t hot = e < L && (d > t || (eq && eqr < extra));
            const uint32_t hr = hot_before + block_excl_scan<NT>(hot, s_ws, tot_hot);
            if (e < L) sc.nid[e] = hot ? hr : H + (e - hr);
            eq_before += tot_eq; hot_before += tot_hot;
        }
        em2_gsync();
        for (uint32_t w = tid; w < Wc; w += NT) {
            const uint32_t id = sc.nid[sc.cw[w]];
            sc.cw[w] = id;
            if (narrow4) sc.cw16[w] = (uint16_t)id;
        }
        // What the rounds read per entry, in STATE order (they walk state ids, not entries: no nid[] per round): the single-label
        // counts over the degree table, which is dead now, and - 16-bit ids - both sibling links in one word over ent_s2.
        uint32_t* st_cnt = sc.g_pre;
        for (uint32_t e = tid; e < L; e += NT) st_cnt[sc.nid[e]] = sc.ent_ucnt[e];
        if (usa) {
            for (uint32_t e = tid; e < L; e += NT) {
                uint32_t a = sc.ent_s1[e], b = sc.ent_s2[e];
                if (!(a & kSibPassive)) a = sc.nid[a];   // (the three special values have the top bit set, too)
                if (!(b & kSibPassive)) b = sc.nid[b];
                if (narrow4) sc.ent_s1[e] = em2_map_sib(a, L, P) | (em2_map_sib(b, L, P) << 16);
                else { sc.ent_s1[e] = a; sc.ent_s2[e] = b; }
            }
            if (narrow4) {
                em2_gsync();   // (every ent_s2[e] has been read)
                for (uint32_t e = tid; e < L; e += NT) sc.ent_s2[sc.nid[e]] = sc.ent_s1[e];
            }
        }
    } else if (usa) {   // tiers 0-3 (state id = entry): both sibling links, as the ids the rounds index with, in one word
        em2_gsync();   // (step 4 wrote links of other threads' entries)
        for (uint32_t e = tid; e < L; e += NT) sc.ent_s1[e] = em2_map_sib(sc.ent_s1[e], L, P) | (em2_map_sib(sc.ent_s2[e], L, P) << 16);
    }
    if (tid == 0) {
        sc.hdr[H_L] = L; sc.hdr[H_P] = P; sc.hdr[H_K] = K; sc.hdr[H_WC] = Wc; sc.hdr[H_NPU] = nPU; sc.hdr[H_FBITS] = em2_fbits(m.nrec);
        sc.hdr[H_TIER] = tier; sc.hdr[H_NHOT] = H; sc.hdr[H_NARROW] = narrow4 ? 1u : 0u;
        const uint32_t at = atomicAdd(&tiers[tier], 1u);
        tiers[8 + (size_t)tier * n_cells + at] = cell;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The rounds.  MODE 0: everything in LDS, per-entry constants in registers (EPT entries per thread); 1: accumulators and
// abundances in LDS, class lists and entry constants streamed.  (Cells beyond that: k_em2_rounds_hybrid below.)
template <typename T> struct IdLoad;
template <> struct IdLoad<uint16_t> { static __device__ __forceinline__ uint32_t at(const uint16_t* p, uint32_t i) { return p[i]; } };
template <> struct IdLoad<uint32_t> { static __device__ __forceinline__ uint32_t at(const uint32_t* p, uint32_t i) { return p[i]; } };


__device__ __forceinline__ void em2_add(unsigned long long* p, unsigned long long v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// One pass over the classes: D in label order, r = 1 / D, every label word's share into its entry's accumulator.
//
// Every load of a trip is UNCONDITIONAL, at an index clamped into the arrays, and its value is masked afterwards.  A load
// under `k < n ? load : 0` sits in a divergent block of its own, and the wait-count pass cannot count across those: it put
// `s_waitcnt vmcnt(0)` in front of every gather whose index came from an earlier (conditional) load, so the sixteen gathers
// of a trip went out one L2 round trip after the other (round 5: the ISA of k_em2_rounds_hybrid).  In straight-line code a
// level's loads issue back to back and the trip is three round trips deep: offsets, label words, abundances.
// `load(e)`: what the memory holds for state id e (nothing but loads), `pick(e, raw)`: its abundance out of that; a scheduling
// barrier between a level's loads and their first use keeps the scheduler from sinking each load to its use (it does, to save
// registers, which serialises them again).  `add(e, q)`: q into e's accumulator.  CPT classes per thread and trip.
// `locate(c, ok, o0, n)`: first word and length of class c (ok: c < K; otherwise any valid word and length 0) - from the offset
// table (LocateByOffsets) or from the cell's runs of equal-length classes (LocateByRuns: no bytes from memory).
template <typename IdT>
struct LocateByOffsets {   // coff[c] .. coff[c + 1]
    const IdT* coff; uint32_t K;
    __device__ __forceinline__ void operator()(uint32_t c, bool ok, uint32_t& o0, uint32_t& n) const {
        const uint32_t cc = ok ? c : K - 1;
        o0 = IdLoad<IdT>::at(coff, cc);
        const uint32_t o1 = IdLoad<IdT>::at(coff, cc + 1);
        n = ok ? o1 - o0 : 0u;
    }
};
// The classes of a cell are laid out longest label first, equal lengths in one run (k_em2_setup): class c of a run starts at
// first word + (c - first class) x length.  A thread's classes ascend, so it walks the run list (LDS, up to 64 entries) with a
// cursor.  The run of labels of 63 words and more has its offsets in coff.
struct LocateByRuns {
    const uint4* runs; const uint32_t* coff; uint32_t ri; uint4 run;
    __device__ __forceinline__ LocateByRuns(const uint4* r, const uint32_t* c) : runs(r), coff(c), ri(0), run(r[0]) {}
    __device__ __forceinline__ void operator()(uint32_t c, bool ok, uint32_t& o0, uint32_t& n) {
        o0 = 0; n = 0;
        if (!ok) return;
        while (c >= run.w && ri < 63u) run = runs[++ri];   // (bounded: a list that does not cover c must not hang the kernel)
        o0 = run.y + (c - run.x) * run.z; n = run.z;
        if (run.z == 0) { o0 = coff[c]; n = coff[c + 1] - o0; }
    }
};
template <int NT, int CPT, int EW, typename IdT, typename Locate, typename Load, typename Pick, typename Add>
__device__ __forceinline__ void em2_class_pass(const IdT* __restrict__ cw, uint32_t c_begin, uint32_t c_end, float scale, Locate& locate, Load load, Pick pick,
                                               Add add, unsigned long long* tq = nullptr) {
#ifdef AFQ_EM_TIMING   // (thread 0's clock per level of a trip; the waits it forces are not in the product build)
    unsigned long long tq_t = wall_clock64();
#define EM2Q(i) do { if (tq && threadIdx.x == 0) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long n_ = wall_clock64(); tq[i] += n_ - tq_t; tq_t = n_; } } while (0)
#else
#define EM2Q(i) do {} while (0)
#endif
    // classes [c_begin, c_end), CPT per thread and trip, the first EW words of each label in registers
    for (uint32_t c0 = c_begin + threadIdx.x; c0 < c_end; c0 += CPT * NT) {
        uint32_t o0[CPT], n[CPT], e[CPT][EW];
        float a[CPT][EW];
#pragma unroll
        for (int j = 0; j < CPT; ++j) { const uint32_t c = c0 + j * NT; locate(c, c < c_end, o0[j], n[j]); }
        EM2Q(0);
#pragma unroll
        for (int j = 0; j < CPT; ++j)
#pragma unroll
            for (int k = 0; k < EW; ++k) e[j][k] = IdLoad<IdT>::at(cw, o0[j] + ((uint32_t)k < n[j] ? (uint32_t)k : 0u));
        EM2Q(1);
        decltype(load(0u)) raw[CPT][EW];
#pragma unroll
        for (int j = 0; j < CPT; ++j)
#pragma unroll
            for (int k = 0; k < EW; ++k) raw[j][k] = load(e[j][k]);
        EM2Q(2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < CPT; ++j)
#pragma unroll
            for (int k = 0; k < EW; ++k) a[j][k] = (uint32_t)k < n[j] ? pick(e[j][k], raw[j][k]) : 0.0f;
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            if (n[j] == 0) continue;
            float d = 0.0f;
#pragma unroll
            for (int k = 0; k < EW; ++k) if ((uint32_t)k < n[j]) d += a[j][k];
            for (uint32_t k = EW; k < n[j]; k += 4) {   // (longer labels: four words a step, the additions in label order)
                uint32_t e4[4]; float a4[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) e4[t] = IdLoad<IdT>::at(cw, o0[j] + (k + t < n[j] ? k + t : 0u));
                decltype(load(0u)) r4[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) r4[t] = load(e4[t]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 4; ++t) a4[t] = pick(e4[t], r4[t]);
#pragma unroll
                for (int t = 0; t < 4; ++t) if (k + t < n[j]) d += a4[t];
            }
            if (!(d > 0.0f)) continue;
            const float r = 1.0f / d;
#pragma unroll
            for (int k = 0; k < EW; ++k)
                if ((uint32_t)k < n[j]) add(e[j][k], (unsigned long long)((a[j][k] * r) * scale));
            for (uint32_t k = EW; k < n[j]; k += 4) {
                uint32_t e4[4]; float a4[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) e4[t] = IdLoad<IdT>::at(cw, o0[j] + (k + t < n[j] ? k + t : 0u));
                decltype(load(0u)) r4[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) r4[t] = load(e4[t]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 4; ++t) a4[t] = pick(e4[t], r4[t]);
#pragma unroll
                for (int t = 0; t < 4; ++t) if (k + t < n[j]) add(e4[t], (unsigned long long)((a4[t] * r) * scale));
            }
        }
        EM2Q(3);
    }
}
// The streamed instances: the classes are laid out longest label first, so the labels beyond four words are the first c4
// classes.  They go through with eight words in registers, half as many classes per thread; one class after the other in the
// four-word form each of them was a chain of two round trips per four further words, twice (sum, shares), and a tailed cell's
// first trip took as long as the other four together (round 5, thread 0's clock per level: 35 of a round's 50 us).
// ... and the labels of one or two words - most classes: two in three on the tailed model, five in six on the plain one - the last
// K - c2: with two words in registers, eight classes per thread (in the four-word form half of their loads and gathers were
// issued for nothing; the pass is bound by instruction issue).
template <int NT, typename IdT, typename Load, typename Pick, typename Add>
__device__ __forceinline__ void em2_class_pass_runs(const IdT* __restrict__ cw, const uint4* runs, const uint32_t* coff, uint32_t c4, uint32_t c2, uint32_t K,
                                                    float scale, Load load, Pick pick, Add add, unsigned long long* tq = nullptr) {
    LocateByRuns loc(runs, coff);
    em2_class_pass<NT, kCptS / 2, 8, IdT>(cw, 0u, c4, scale, loc, load, pick, add, tq);
    em2_class_pass<NT, kCptS, 4, IdT>(cw, c4, c2, scale, loc, load, pick, add, tq);
#if AFQ_EM2_PAIRS_PASS
    em2_class_pass<NT, 2 * kCptS, 2, IdT>(cw, c2, K, scale, loc, load, pick, add, tq);
#else
    em2_class_pass<NT, kCptS, 4, IdT>(cw, c2, K, scale, loc, load, pick, add, tq);
#endif
}

#ifdef AFQ_EM_TIMING
__device__ unsigned long long g_em2_dbg[32768][4];
__device__ uint32_t g_em2_dbg_n;
__global__ void k_em2_dbg_dump() {
    const uint32_t n = g_em2_dbg_n < 32768 ? g_em2_dbg_n : 32768;
    for (uint32_t i = 0; i < n; ++i) printf("em2 blk %llx %llu %llu %llu\n", g_em2_dbg[i][0], g_em2_dbg[i][1], g_em2_dbg[i][2], g_em2_dbg[i][3]);
    printf("em2 blk end of launch\n");
    g_em2_dbg_n = 0;
}
#endif
template <int NT, int MODE, int EPT, uint32_t LDSW>
__global__ __launch_bounds__(NT) void k_em2_rounds(const CellMeta* __restrict__ meta, const uint32_t* __restrict__ nnz_unique,
                                                   const uint32_t* __restrict__ lab_cnt, const uint64_t* __restrict__ em_off,
                                                   uint32_t* __restrict__ scratch, uint32_t* __restrict__ out_nnz,
                                                   const uint32_t* __restrict__ tiers, uint32_t n_cells, uint32_t tier, Em2Cfg cfg) {
    __shared__ uint32_t s_ws[NT / 64];
    __shared__ uint32_t s_flag[2];
    __shared__ __attribute__((aligned(16))) uint32_t s_mem[LDSW];
    __shared__ uint4 s_run[MODE == 1 ? 64 : 1];
#ifdef AFQ_EM_TIMING
    const unsigned long long t_entry = wall_clock64();
#endif
    if (blockIdx.x >= tiers[tier]) return;
    const uint32_t cell = tiers[8 + (size_t)tier * n_cells + blockIdx.x];
    const uint32_t tid = threadIdx.x;
    const uint32_t nU = nnz_unique[cell], W = lab_cnt[2 * cell], M = lab_cnt[2 * cell + 1];
    const bool usa = cfg.usa != 0;
    const Em2Scratch sc = em2_carve(scratch, em_off[cell], nU, W, M, usa);
    const uint32_t L = sc.hdr[H_L], P = sc.hdr[H_P], K = sc.hdr[H_K], Wc = sc.hdr[H_WC], nPU = sc.hdr[H_NPU], F = sc.hdr[H_FBITS];
    if constexpr (MODE == 1) {
        if (tid < sc.hdr[H_NRUN]) { const uint32_t* q = sc.hdr + 16 + 4 * tid; s_run[tid] = make_uint4(q[0], q[1], q[2], q[3]); }
    }
    const float scale = __uint_as_float((127u + F) << 23), inv_scale = __uint_as_float((127u - F) << 23);
    const uint32_t Z0 = L + P, Z1 = L + P + 1;
    // placement
    unsigned long long* acc;
    float *ab, *v;
    {
        acc = reinterpret_cast<unsigned long long*>(s_mem);
        float* f = reinterpret_cast<float*>(s_mem + 2 * L);
        ab = f; if (usa) f += L;
        v = f;
        if (!usa) ab = v;
    }
    const uint16_t* coff16 = nullptr; const uint16_t* cw16 = nullptr;
    if constexpr (MODE == 0) {
        uint16_t* c16 = reinterpret_cast<uint16_t*>(s_mem + em2_lds_core_words(L, P, usa));
        uint16_t* w16 = c16 + ((K + 2) & ~1u);
        for (uint32_t c = tid; c <= K; c += NT) c16[c] = (uint16_t)sc.coff[c];
        for (uint32_t w = tid; w < Wc; w += NT) w16[w] = (uint16_t)sc.cw[w];
        coff16 = c16; cw16 = w16;
    }
    const float uni = 1.0f / (float)cfg.num_alphas;
    auto init_of = [&](uint32_t cnt) -> float { return cfg.init_uniform ? uni : ((float)cnt + 0.5f) * 1e-3f; };
    // per-entry constants: registers (MODE 0) or streamed
    [[maybe_unused]] uint32_t r_cnt[EPT], r_sib[EPT];
    if constexpr (MODE == 0) {
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            const uint32_t e = tid + j * NT;
            r_cnt[j] = 0; r_sib[j] = 0;
            if (e < L) {
                r_cnt[j] = sc.ent_ucnt[e];
                if (usa) r_sib[j] = sc.ent_s1[e];   // (both links, as indices into v)
                v[e] = init_of(r_cnt[j]);
                acc[e] = (unsigned long long)r_cnt[j] << F;
            }
        }
    } else {
        for (uint32_t e = tid; e < L; e += NT) { const uint32_t c = sc.ent_ucnt[e]; v[e] = init_of(c); acc[e] = (unsigned long long)c << F; }
    }
    for (uint32_t p = tid; p < P; p += NT) v[L + p] = init_of(sc.pas_val[p]);
    if (tid == 0) { v[Z0] = init_of(0u); v[Z1] = 0.0f; s_flag[0] = 0; s_flag[1] = 0; }
    __syncthreads();
#ifdef AFQ_EM_TIMING
    unsigned long long tph[4] = {0, 0, 0, 0}, tph_t = wall_clock64();
    const unsigned long long t_begin = tph_t;
#define EM2T(i) do { if (tid == 0) { const unsigned long long n_ = wall_clock64(); tph[i] += n_ - tph_t; tph_t = n_; } } while (0)
#else
#define EM2T(i) do {} while (0)
#endif
    uint32_t it = 0;
    bool conv = true, last_round = false;
    while (it < kMinIter2 || (it < kMaxIter2 && !conv) || last_round) {
        EM2T(3);
        if (usa) {   // (C) what a label word contributes with: own + sibling statuses
            if constexpr (MODE == 0) {
#pragma unroll
                for (int j = 0; j < EPT; ++j) {
                    const uint32_t e = tid + j * NT;
                    if (e < L) ab[e] = (v[r_sib[j] & 0xFFFFu] + v[r_sib[j] >> 16]) + v[e];
                }
            } else {
                for (uint32_t e0 = tid; e0 < L; e0 += 4 * NT) {   // (four entries per thread and trip, their loads unconditional: em2_class_pass)
                    uint32_t sb[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const uint32_t e = e0 + j * NT; sb[j] = sc.ent_s1[e < L ? e : 0u]; }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const uint32_t e = e0 + j * NT; if (e < L) ab[e] = (v[sb[j] & 0xFFFFu] + v[sb[j] >> 16]) + v[e]; }
                }
            }
            __syncthreads();
        }
        // (A+B) classes
        {
            auto load = [&](uint32_t e) -> float { return ab[e]; };
            auto pick = [](uint32_t, float x) -> float { return x; };
            auto add = [&](uint32_t e, unsigned long long q) { em2_add(&acc[e], q); };
            if constexpr (MODE == 0) { LocateByOffsets<uint16_t> loc{coff16, K}; em2_class_pass<NT, 2, 4, uint16_t>(cw16, 0u, K, scale, loc, load, pick, add); }
            else em2_class_pass_runs<NT, uint16_t>(sc.cw16, s_run, sc.coff, sc.hdr[H_C4], sc.hdr[H_C2], K, scale, load, pick, add);
        }
        EM2T(0);
        __syncthreads();
        EM2T(1);
        // (E) entries: new abundance, convergence vote, accumulator back to the single-label count
        bool bad = false;
        if (tid == 0) s_flag[(it + 1) & 1u] = 0;   // the other round's flag: everyone read it before the barrier above
        auto entry = [&](uint32_t e, uint32_t cnt) {
            const unsigned long long a = acc[e];
            acc[e] = (unsigned long long)cnt << F;
            const float x = (float)a * inv_scale;
            const float old = v[e];
            if (x > kAlphaCheckCutoff2 && fabsf(old - x) > kRelDiffTol2) bad = true;
            v[e] = x;
        };
        if constexpr (MODE == 0) {
#pragma unroll
            for (int j = 0; j < EPT; ++j) { const uint32_t e = tid + j * NT; if (e < L) entry(e, r_cnt[j]); }
        } else {
            for (uint32_t e0 = tid; e0 < L; e0 += 4 * NT) {
                uint32_t uc[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { const uint32_t e = e0 + j * NT; uc[j] = sc.ent_ucnt[e < L ? e : 0u]; }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j) { const uint32_t e = e0 + j * NT; if (e < L) entry(e, uc[j]); }
            }
        }
        if (it == 0) {   // an entry outside every label holds its single-label count from the first round on; a status nothing maps to, 0
            for (uint32_t p = tid; p < P; p += NT) v[L + p] = (float)sc.pas_val[p];
            if (tid == 0) v[Z0] = 0.0f;
        }
        if (bad) s_flag[it & 1u] = 1;
        __syncthreads();
        EM2T(2);
        conv = s_flag[it & 1u] == 0;
        ++it;
        if (usa) {   // em_optimize_subset_impl: after the first converged round zero what is below the output floor, one more round (em.rs:391-451)
            if (last_round) break;
            if (it >= kMinIter2 && conv) {
                for (uint32_t e = tid; e < L; e += NT) if (v[e] < kMinOutputAlpha2) v[e] = 0.0f;
                last_round = true;
                __syncthreads();
            }
        }
    }
    // output row: pass-through columns and the live entries at or above the floor, merged by column
    uint32_t* pre = s_mem;   // (over the accumulators, which are dead)
    __syncthreads();
    uint32_t nout = 0;
    for (uint32_t base = 0; base < L; base += NT) {
        const uint32_t e = base + tid;
        const float x = e < L ? v[e] : 0.0f;
        const uint32_t h = x >= kMinOutputAlpha2;
        uint32_t tot;
        const uint32_t ex = block_excl_scan<NT>(h, s_ws, tot);
        if (e < L) {
            pre[e] = nout + ex;   // (LDS instances: over the dead accumulators - the abundances read above sit behind them)
            if (h) sc.out[sc.ent_ub[e] + nout + ex] = make_uint2(sc.ent_col[e], __float_as_uint(x));
        }
        nout += tot;
    }
    if (tid == 0) pre[L] = nout;
    __syncthreads();
    for (uint32_t j = tid; j < nPU; j += NT) sc.out[j + pre[sc.pu_lb[j]]] = make_uint2(sc.pu_col[j], __float_as_uint((float)sc.pu_cnt[j]));
    if (tid == 0) out_nnz[cell] = nPU + nout;
#ifdef AFQ_EM_TIMING
    if (tid == 0 && (blockIdx.x % 100) == 3)
        printf("em2 rounds tier=%u NT=%d L=%u P=%u K=%u Wc=%u nPU=%u it=%u: C+classes(thread 0)=%.1f wait=%.1f entries=%.1f other=%.1f total=%.1f us\n", tier, NT, L, P, K, Wc, nPU, it,
               (double)tph[0] / 100.0, (double)tph[1] / 100.0, (double)tph[2] / 100.0, (double)tph[3] / 100.0, (double)(wall_clock64() - t_begin) / 100.0);
    if (tid == 0) {
        uint32_t hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        uint32_t xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        const uint32_t at = atomicAdd(&g_em2_dbg_n, 1u);
        if (at < 32768) { g_em2_dbg[at][0] = ((unsigned long long)tier << 56) | ((unsigned long long)(xcc & 0xf) << 48) | ((unsigned long long)(hw & 0xffff) << 32) | blockIdx.x;
                          g_em2_dbg[at][1] = t_entry; g_em2_dbg[at][2] = t_begin; g_em2_dbg[at][3] = wall_clock64(); }
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------
// The rounds of a cell whose state does not fit LDS (k_em2_setup step 7): state ids below H - the entries in most classes -
// live in LDS (accumulator, contribution, abundance), the others in the cell's global scratch, as do the class lists and
// the per-entry constants.  A hot entry takes its atomics in LDS; a cold one is in one or two classes: its atomics go to L2
// without meeting another.  Same arithmetic, same bits.
template <int NT>
__global__ __launch_bounds__(NT) void k_em2_rounds_hybrid(const uint32_t* __restrict__ nnz_unique, const uint32_t* __restrict__ lab_cnt,
                                                          const uint64_t* __restrict__ em_off, uint32_t* __restrict__ scratch,
                                                          uint32_t* __restrict__ out_nnz, const uint32_t* __restrict__ tiers, uint32_t n_cells,
                                                          uint32_t tier, Em2Cfg cfg) {
    __shared__ uint32_t s_ws[NT / 64];
    __shared__ uint32_t s_flag[2];
    __shared__ __attribute__((aligned(16))) uint32_t s_mem[kT2Words];
    __shared__ uint4 s_run[64];
#ifdef AFQ_EM_TIMING
    const unsigned long long t_entry = wall_clock64();
#endif
    if (blockIdx.x >= tiers[tier]) return;
    const uint32_t cell = tiers[8 + (size_t)tier * n_cells + blockIdx.x];
    const uint32_t tid = threadIdx.x;
    const uint32_t nU = nnz_unique[cell], W = lab_cnt[2 * cell], M = lab_cnt[2 * cell + 1];
    const bool usa = cfg.usa != 0;
    const Em2Scratch sc = em2_carve(scratch, em_off[cell], nU, W, M, usa);
    const uint32_t L = sc.hdr[H_L], P = sc.hdr[H_P], K = sc.hdr[H_K], nPU = sc.hdr[H_NPU], F = sc.hdr[H_FBITS], H = sc.hdr[H_NHOT];
    const uint32_t c4 = sc.hdr[H_C4], c2 = sc.hdr[H_C2];
    const bool narrow = sc.hdr[H_NARROW] != 0;   // 16-bit state ids: label words in cw16, both sibling links of state s in ent_s2[s]
    const uint32_t* st_cnt = sc.g_pre;           // single-label counts in state order (until the output row takes the array back)
    if (tid < sc.hdr[H_NRUN]) { const uint32_t* q = sc.hdr + 16 + 4 * tid; s_run[tid] = make_uint4(q[0], q[1], q[2], q[3]); }
    const float scale = __uint_as_float((127u + F) << 23), inv_scale = __uint_as_float((127u - F) << 23);
    const uint32_t Z0 = L + P, Z1 = L + P + 1;
    unsigned long long* acc_h = reinterpret_cast<unsigned long long*>(s_mem);
    float* v_h = reinterpret_cast<float*>(s_mem + 2 * H);
    float* ab_h = usa ? v_h + H : v_h;
    unsigned long long* acc_g = sc.g_acc;
    float* v_g = sc.g_v;
    float* ab_g = usa ? sc.g_ab : sc.g_v;
    // A state id is read from BOTH homes, each at an index that is valid there, and the answer picked afterwards: `hot ? lds[i] :
    // global[i]` compiles to a FLAT load of a selected pointer (slower than either, and it counts in both wait counters), and a
    // load under a branch stops the loads behind it (em2_class_pass).  Writes and atomics take two `if`s, not an if / else, so
    // that the two sides are not merged into one flat instruction.
    auto V = [&](uint32_t s2) -> float { const bool hot = s2 < H; const float xh = v_h[hot ? s2 : 0u], xg = v_g[hot ? 0u : s2]; return hot ? xh : xg; };
    auto setV = [&](uint32_t s2, float x) { if (s2 < H) v_h[s2] = x; if (s2 >= H) v_g[s2] = x; };
    auto ab_load = [&](uint32_t s2) -> float2 { const bool hot = s2 < H; return make_float2(ab_h[hot ? s2 : 0u], ab_g[hot ? 0u : s2]); };
    auto ab_pick = [&](uint32_t s2, float2 x) -> float { return s2 < H ? x.x : x.y; };
    auto add = [&](uint32_t s2, unsigned long long q) { if (s2 < H) em2_add(&acc_h[s2], q); if (s2 >= H) em2_add(&acc_g[s2], q); };   // (LDS or L2: never a flat atomic)
    const float uni = 1.0f / (float)cfg.num_alphas;
    auto init_of = [&](uint32_t cnt) -> float { return cfg.init_uniform ? uni : ((float)cnt + 0.5f) * 1e-3f; };
    for (uint32_t e = tid; e < L; e += NT) {
        const uint32_t s2 = sc.nid[e], c = sc.ent_ucnt[e];
        setV(s2, init_of(c));
        if (s2 < H) acc_h[s2] = (unsigned long long)c << F;
        if (s2 >= H) acc_g[s2] = (unsigned long long)c << F;
    }
    for (uint32_t p = tid; p < P; p += NT) v_g[L + p] = init_of(sc.pas_val[p]);
    if (tid == 0) { v_g[Z0] = init_of(0u); v_g[Z1] = 0.0f; s_flag[0] = 0; s_flag[1] = 0; }
    em2_gsync();
#ifdef AFQ_EM_TIMING
    unsigned long long hph[5] = {0, 0, 0, 0, 0}, hq[4] = {0, 0, 0, 0}, hph_t = wall_clock64();
    const unsigned long long h_begin = hph_t;
#define EM2H(i) do { if (tid == 0) { const unsigned long long n_ = wall_clock64(); hph[i] += n_ - hph_t; hph_t = n_; } } while (0)
#else
#define EM2H(i) do {} while (0)
#endif
    uint32_t it = 0;
    bool conv = true, last_round = false;
    while (it < kMinIter2 || (it < kMaxIter2 && !conv) || last_round) {
        EM2H(4);
        if (usa && narrow) {   // (C) by state id: one word holds both links
            for (uint32_t s0 = tid; s0 < L; s0 += 4 * NT) {   // (four states per thread and trip: their link loads, then their gathers, in flight together)
                uint32_t sb[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { const uint32_t st = s0 + j * NT; sb[j] = sc.ent_s2[st < L ? st : 0u]; }
                __builtin_amdgcn_sched_barrier(0);
                float x[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { const uint32_t st = s0 + j * NT; x[j] = (V(sb[j] & 0xFFFFu) + V(sb[j] >> 16)) + V(st < L ? st : 0u); }
#pragma unroll
                for (int j = 0; j < 4; ++j) { const uint32_t st = s0 + j * NT; if (st < L) { if (st < H) ab_h[st] = x[j]; if (st >= H) ab_g[st] = x[j]; } }
            }
            em2_gsync();
        } else if (usa) {   // (ids beyond 16 bits: by entry, through nid)
            for (uint32_t e0 = tid; e0 < L; e0 += 4 * NT) {
                uint32_t s2[4], q1[4], q2[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t e = e0 + j * NT;
                    const bool ok = e < L;
                    const uint32_t ee = ok ? e : 0u;   // (loads unconditional, see em2_class_pass)
                    const uint32_t t0 = sc.nid[ee], t1 = sc.ent_s1[ee], t2 = sc.ent_s2[ee];
                    s2[j] = ok ? t0 : 0u; q1[j] = ok ? t1 : kSibNone; q2[j] = ok ? t2 : kSibNone;
                }
                float x[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) x[j] = (V(em2_map_sib(q1[j], L, P)) + V(em2_map_sib(q2[j], L, P))) + V(s2[j]);
#pragma unroll
                for (int j = 0; j < 4; ++j) if (e0 + j * NT < L) { if (s2[j] < H) ab_h[s2[j]] = x[j]; if (s2[j] >= H) ab_g[s2[j]] = x[j]; }
            }
            em2_gsync();
        }
        EM2H(0);
#ifdef AFQ_EM_TIMING
        unsigned long long* const tq = hq;
#else
        unsigned long long* const tq = nullptr;
#endif
        if (narrow) em2_class_pass_runs<NT, uint16_t>(sc.cw16, s_run, sc.coff, c4, c2, K, scale, ab_load, ab_pick, add, tq);
        else em2_class_pass_runs<NT, uint32_t>(sc.cw, s_run, sc.coff, c4, c2, K, scale, ab_load, ab_pick, add, tq);
        EM2H(1);
        em2_gsync();
        EM2H(2);
        bool bad = false;
        if (tid == 0) s_flag[(it + 1) & 1u] = 0;
        for (uint32_t s0 = tid; s0 < L; s0 += 4 * NT) {   // (E) by state id
            uint32_t uc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { const uint32_t st = s0 + j * NT; uc[j] = st_cnt[st < L ? st : 0u]; }
            unsigned long long a[4];
            float old[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t st = s0 + j * NT;
                const bool ok = st < L, hot = st < H;
                const unsigned long long fresh = (unsigned long long)uc[j] << F;
                const unsigned long long ah = acc_h[hot ? st : 0u];
                old[j] = V(ok ? st : 0u);
                if (ok && hot) acc_h[st] = fresh;
                unsigned long long ag = 0;
                if (ok && !hot) ag = __hip_atomic_exchange(&acc_g[st], fresh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                a[j] = hot ? ah : ag;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t st = s0 + j * NT;
                if (st >= L) continue;
                const float x = (float)a[j] * inv_scale;
                if (x > kAlphaCheckCutoff2 && fabsf(old[j] - x) > kRelDiffTol2) bad = true;
                setV(st, x);
            }
        }
        if (it == 0) {
            for (uint32_t p = tid; p < P; p += NT) v_g[L + p] = (float)sc.pas_val[p];
            if (tid == 0) v_g[Z0] = 0.0f;
        }
        if (bad) s_flag[it & 1u] = 1;
        em2_gsync();
        EM2H(3);
        conv = s_flag[it & 1u] == 0;
        ++it;
        if (usa) {
            if (last_round) break;
            if (it >= kMinIter2 && conv) {
                for (uint32_t st = tid; st < L; st += NT) if (V(st) < kMinOutputAlpha2) setV(st, 0.0f);
                last_round = true;
                em2_gsync();
            }
        }
    }
    uint32_t* pre = sc.g_pre;
    uint32_t nout = 0;
    for (uint32_t base = 0; base < L; base += NT) {
        const uint32_t e = base + tid;
        const float x = e < L ? V(sc.nid[e]) : 0.0f;
        const uint32_t h = x >= kMinOutputAlpha2;
        uint32_t tot;
        const uint32_t ex = block_excl_scan<NT>(h, s_ws, tot);
        if (e < L) {
            pre[e] = nout + ex;
            if (h) sc.out[sc.ent_ub[e] + nout + ex] = make_uint2(sc.ent_col[e], __float_as_uint(x));
        }
        nout += tot;
    }
    if (tid == 0) pre[L] = nout;
    em2_gsync();
    for (uint32_t j = tid; j < nPU; j += NT) sc.out[j + pre[sc.pu_lb[j]]] = make_uint2(sc.pu_col[j], __float_as_uint((float)sc.pu_cnt[j]));
    if (tid == 0) out_nnz[cell] = nPU + nout;
#ifdef AFQ_EM_TIMING
    if (tid == 0) {
        const uint32_t at = atomicAdd(&g_em2_dbg_n, 1u);
        if (at < 32768) { g_em2_dbg[at][0] = ((unsigned long long)tier << 56) | blockIdx.x; g_em2_dbg[at][1] = t_entry; g_em2_dbg[at][2] = t_entry; g_em2_dbg[at][3] = wall_clock64(); }
        if ((blockIdx.x % 100) == 3) printf("em2 hybrid class pass, thread 0: locate=%.1f words=%.1f gathers=%.1f sums+shares=%.1f us\n", (double)hq[0] / 100.0, (double)hq[1] / 100.0, (double)hq[2] / 100.0, (double)hq[3] / 100.0);
        if ((blockIdx.x % 100) == 3) printf("em2 hybrid L=%u H=%u P=%u K=%u Wc=%u it=%u: C=%.1f classes(thread 0)=%.1f wait=%.1f entries=%.1f other=%.1f rounds=%.1f total=%.1f us\n", L, H, P, K, sc.hdr[H_WC], it,
                                            (double)hph[0] / 100.0, (double)hph[1] / 100.0, (double)hph[2] / 100.0, (double)hph[3] / 100.0, (double)hph[4] / 100.0,
                                            (double)(wall_clock64() - h_begin) / 100.0, (double)(wall_clock64() - t_entry) / 100.0);
    }
#endif
}

// Where each cell's scratch slice starts, on the device (so that the EM follows the range's other kernels without a trip to
// the host): exact sizes from the counts the resolution kernels left, packed; tiers[7] is set when the packed total exceeds
// what the host allocated - every kernel below then returns at once and the host sizes the EM itself (afq_api.cpp).
__global__ __launch_bounds__(1024) void k_em2_plan(const uint32_t* __restrict__ nnz_unique, const uint32_t* __restrict__ lab_cnt, uint32_t n_cells,
                                                   uint32_t usa, unsigned long long cap_words, uint64_t* __restrict__ em_off, uint32_t* __restrict__ tiers,
                                                   const DevStatus* __restrict__ st) {
    __shared__ unsigned long long s_sum[1024];
    if (st->err_code) { if (threadIdx.x == 0) tiers[7] = 1; return; }   // (the range failed: its counts are not to be trusted; the host reports the error)
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (n_cells + 1023u) / 1024u, i0 = tid * per, i1 = i0 + per < n_cells ? i0 + per : n_cells;
    unsigned long long sum = 0;
    for (uint32_t i = i0; i < i1; ++i) sum += em2_words(nnz_unique[i], lab_cnt[2 * i], lab_cnt[2 * i + 1], usa != 0);
    s_sum[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        unsigned long long run = 0;
        for (uint32_t t = 0; t < 1024; ++t) { const unsigned long long x = s_sum[t]; s_sum[t] = run; run += x; }
        em_off[n_cells] = run;
        if (run > cap_words) tiers[7] = 1;
    }
    __syncthreads();
    unsigned long long off = s_sum[tid];
    for (uint32_t i = i0; i < i1; ++i) { em_off[i] = off; off += em2_words(nnz_unique[i], lab_cnt[2 * i], lab_cnt[2 * i + 1], usa != 0); }
}

void launch_em2(hipStream_t s, const ResolveArgs& a, uint32_t n_cells, uint64_t* em_off, uint32_t* scratch, uint32_t* out_nnz,
                const uint32_t* em_order, uint32_t* tiers, uint32_t num_alphas, uint32_t init_uniform, uint64_t plan_cap_words) {
    if (!n_cells) return;
    uint32_t min_tier = 0;
    if (const char* e = test_hook("EM2_MIN_TIER")) min_tier = (uint32_t)std::min(4, std::max(0, std::atoi(e)));   // (tests; read per range)
    Em2Cfg cfg{a.usa, num_alphas, a.num_rows / 3, 2 * (a.num_rows / 3), init_uniform, (num_alphas + 31) / 32, min_tier, test_hook_is("EM2_WIDE_IDS", "1") ? 1u : 0u};
    if (!plan_cap_words) (void)hipMemsetAsync(tiers, 0, 32, s);   // (with a device-side plan the range's init kernel has cleared the counters)
    if (plan_cap_words) hipLaunchKernelGGL(k_em2_plan, dim3(1), dim3(1024), 0, s, a.nnz, a.lab_cnt, n_cells, a.usa, (unsigned long long)plan_cap_words, em_off, tiers, a.st);
    hipLaunchKernelGGL(k_em2_setup, dim3(n_cells), dim3(kSetupNT), 8 * cfg.nwb, s, a.meta, a.nnz, a.keys0, a.keys1, a.lab, a.lab_cnt, em_off,
                       scratch, out_nnz, em_order, tiers, n_cells, cfg);
#define EM2_ROUNDS(NT, MODE, EPT, LDSW, TIER) \
    hipLaunchKernelGGL((k_em2_rounds<NT, MODE, EPT, LDSW>), dim3(n_cells), dim3(NT), 0, s, a.meta, a.nnz, a.lab_cnt, em_off, scratch, out_nnz, tiers, n_cells, TIER, cfg)
    hipLaunchKernelGGL(k_em2_rounds_hybrid<1024>, dim3(n_cells), dim3(1024), 0, s, a.nnz, a.lab_cnt, em_off, scratch, out_nnz, tiers, n_cells, 4u, cfg);   // largest first
    EM2_ROUNDS(1024, 1, 1, kT2Words, 3u);
    EM2_ROUNDS(1024, 0, 16, kT2Words, 2u);
    EM2_ROUNDS(512, 0, 8, kT1Words, 1u);
    EM2_ROUNDS(256, 0, 8, kT0Words, 0u);
#undef EM2_ROUNDS
#ifdef AFQ_EM_TIMING
    hipLaunchKernelGGL(k_em2_dbg_dump, dim3(1), dim3(1), 0, s);
#endif
}

// columns the setup kernel's bitmap + rank table can hold in 64 KiB of LDS; beyond that the EM takes the canonical kernels
bool em2_supported(uint32_t num_alphas) { return (num_alphas + 31) / 32 <= 8192; }

}  // namespace afq
