// afq_pug_common.h — pieces shared by the two parsimony paths (afq_pug.hip: one workgroup per cell, every phase in
// one kernel; afq_pug2.hip: phase kernels over UMI partitions): labels as they sit in the chunk, a resolved molecule's
// way into the cell's column list / gene-level classes, and the monochromatic-arborescence covers of small components
// (collapse_vertices / get_num_molecules, src/pugutils.rs:308-391, 1048-1261).  Device code only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "afq_common.h"
#include "afq_kernels.h"
#include "afq_prims.h"

namespace afq {

constexpr uint32_t kMaxGenesPerLabel = 64;        // distinct genes of one molecule the device path carries in registers

struct PugCtx {
    // cell
    const uint32_t* W;       // chunk dwords
    uint32_t HW;             // header dwords of a record
    const uint32_t* t2g;
    uint32_t ref_count, num_genes;
    // config
    uint32_t usa, num_rows, uo, ao, em, exact_umi, large_thresh, umi_pairs, gene_level;
    // outputs
    uint32_t* cols;          // the cell's column list (u32), positions from s_ncols
    uint32_t* labw;          // EM label words
    uint32_t* labd;          // EM label descriptors (off,len)
    uint32_t cols_cap, lab_cap;
    uint32_t* s_cnt;         // LDS: [0] ncols, [1] label words, [2] label count, [3] error flag
    DevStatus* st;
    uint32_t cell;
    uint32_t adj_umi;        // the cover records carry (UMI, reads) in place of an adjacency mask: the covers work the edges out themselves (umi_edge)
};
// has_edge(x -> y) from the two vertices' UMIs and read counts (pugutils.rs:76-99, utils.rs:389-393; the rule k_p2_search's check()
// writes into a pair's direction bits): the same UMI - or one base apart and reads(y) < 2 reads(x); under --umi-edit-dist 0 the same
// UMI only.  (The labels must share a ref as well: the caller's test.)  Inside a component every such pair IS an edge the search
// found - it finds every pair of a cell's vertices that passes this rule and the label test - so a cover that holds a component's
// vertices can work its edges out itself instead of having them delivered (a pass over the pair list with two atomics per pair).
__device__ __forceinline__ bool umi_edge(uint32_t ux, uint32_t cx, uint32_t uy, uint32_t cy, uint32_t exact_umi) {
    const uint32_t x = ux ^ uy;
    if (x == 0) return true;
    if (exact_umi) return false;
    const uint32_t nz = (x | (x >> 1)) & 0x55555555u;   // one bit per base that differs
    return (nz & (nz - 1)) == 0 && cy < 2 * cx;
}

struct Lab {
    const uint32_t* p;
    uint32_t n;
};
__device__ __forceinline__ Lab rec_label(const PugCtx& c, uint32_t rec_dw) {
    Lab l;
    l.n = c.W[rec_dw];
    l.p = c.W + rec_dw + c.HW;
    return l;
}
__device__ __forceinline__ bool lab_contains(const Lab& l, uint32_t t) {  // refs ascending (pugutils.rs:375)
    uint32_t lo = 0, hi = l.n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        const uint32_t v = l.p[mid] & 0x7FFFFFFFu;
        if (v < t) lo = mid + 1; else hi = mid;
    }
    return lo < l.n && (l.p[lo] & 0x7FFFFFFFu) == t;
}
__device__ __forceinline__ bool lab_overlap(const Lab& a, const Lab& b) {  // share >= 1 ref (pugutils.rs:187-204)
    uint32_t i = 0, j = 0;
    while (i < a.n && j < b.n) {
        const uint32_t x = a.p[i] & 0x7FFFFFFFu, y = b.p[j] & 0x7FFFFFFFu;
        if (x == y) return true;
        if (x < y) ++i; else ++j;
    }
    return false;
}
__device__ __forceinline__ bool lab_equal(const Lab& a, const Lab& b) {
    if (a.n != b.n) return false;
    for (uint32_t i = 0; i < a.n; ++i) if ((a.p[i] ^ b.p[i]) & 0x7FFFFFFFu) return false;
    return true;
}

// sorted distinct gene ids of a list of refs (pugutils.rs:1213-1225, 1296-1298); returns the count
// or 0xFFFFFFFF when more than kMaxGenesPerLabel distinct genes turn up.
template <typename GetRef>
__device__ __forceinline__ uint32_t genes_of(const PugCtx& c, uint32_t n, GetRef&& ref, uint32_t* g) {
    uint32_t k = 0;
    for (uint32_t j = 0; j < n; ++j) {
        const uint32_t gid = c.gene_level ? ref(j) : c.t2g[ref(j)];  // gene-level labels already hold gene ids
        uint32_t p = 0;
        while (p < k && g[p] < gid) ++p;
        if (p < k && g[p] == gid) continue;
        if (k == kMaxGenesPerLabel) return 0xFFFFFFFFu;
        for (uint32_t q = k; q > p; --q) g[q] = g[q - 1];
        g[p] = gid;
        ++k;
    }
    return k;
}

// One resolved molecule with gene label g[0..ng): a column, a gene-level class for the EM, or nothing.
// (quant.rs:974-1024 -> extract_counts utils.rs:688-753 / em_optimize(only_unique) em.rs:499-514 / EM)
// molecule_column_n returns the column the molecule counts for (the caller appends it) or 0xFFFFFFFF when there is none -
// dropped, or kept as a class for the EM, which is written here.
__device__ __forceinline__ uint32_t molecule_column_n(const PugCtx& c, const uint32_t* g, uint32_t ng) {
    if (ng == 0xFFFFFFFFu) {   // more than kMaxGenesPerLabel genes: no column under any rule (one gene; USA: at most ten) - only the
        if (c.em) c.s_cnt[3] = kErrPugLimit;   // EM would have to carry the class
        return 0xFFFFFFFFu;
    }
    if (ng == 0) return 0xFFFFFFFFu;
    uint32_t col = 0xFFFFFFFFu;
    if (c.em) {
        if (ng == 1) col = !c.usa ? g[0] : ((g[0] & 1u) == 0 ? (g[0] >> 1) : c.uo + (g[0] >> 1));
        else if (c.usa && ng == 2 && ((g[0] ^ g[1]) & ~1u) == 0) col = c.ao + (g[0] >> 1);
        else {
            const uint32_t off = atomicAdd(&c.s_cnt[1], ng), di = atomicAdd(&c.s_cnt[2], 1u);
            if (off + ng > c.lab_cap || 2 * (di + 1) > c.lab_cap) { c.s_cnt[3] = kErrPugLimit; return 0xFFFFFFFFu; }
            for (uint32_t i = 0; i < ng; ++i) c.labw[off + i] = g[i];
            c.labd[2 * di] = off; c.labd[2 * di + 1] = ng;
            return 0xFFFFFFFFu;
        }
    } else if (!c.usa) {
        if (ng == 1) col = g[0];
    } else if (ng == 1) {
        col = (g[0] & 1u) == 0 ? (g[0] >> 1) : c.uo + (g[0] >> 1);
    } else if (ng == 2) {
        const bool s1 = (g[0] & 1u) == 0, s2 = (g[1] & 1u) == 0;
        if (((g[0] ^ g[1]) & ~1u) == 0) col = c.ao + (g[0] >> 1);
        else if (s1 && !s2) col = g[0] >> 1;
        else if (!s1 && s2) col = g[1] >> 1;
    } else if (ng <= 10) {
        uint32_t nsp = 0, sidx = 0;
        for (uint32_t i = 0; i < ng; ++i) if ((g[i] & 1u) == 0) { if (nsp == 0) sidx = i; ++nsp; }
        if (nsp == 1) {
            const uint32_t sg = g[sidx];
            col = (sidx + 1 < ng && ((sg ^ g[sidx + 1]) & ~1u) == 0) ? c.ao + (sg >> 1) : (sg >> 1);
        }
    }
    if (col == 0xFFFFFFFFu) return col;
    if (col >= c.num_rows) { c.s_cnt[3] = kErrSlotRange; return 0xFFFFFFFFu; }
    return col;
}
// The gene ids in row[0 .. n) (0xFFFFFFFF: none) -> sorted, distinct, in front; returns how many.  In place, by insertion: the
// sorted prefix is never longer than the part already read.  `row` is a lane's row of the wave's LDS stage: the same walk over
// an array of the lane's own (`g[]` above) runs in scratch memory, a round trip to the cache per step.
__device__ __forceinline__ uint32_t sort_unique_in_row(uint32_t* row, uint32_t n) {
    uint32_t ns = 0;
    for (uint32_t j = 0; j < n; ++j) {
        const uint32_t gid = row[j];
        if (gid == 0xFFFFFFFFu) continue;
        uint32_t q = 0;
        while (q < ns && row[q] < gid) ++q;
        if (q < ns && row[q] == gid) continue;
        for (uint32_t r = ns; r > q; --r) row[r] = row[r - 1];
        row[q] = gid;
        ++ns;
    }
    return ns;
}
__device__ __forceinline__ void emit_molecule(const PugCtx& c, const uint32_t* g, uint32_t ng) {
    const uint32_t col = molecule_column_n(c, g, ng);
    if (col == 0xFFFFFFFFu) return;
    const uint32_t p = atomicAdd(&c.s_cnt[0], 1u);
    if (p >= c.cols_cap) { c.s_cnt[3] = kErrPugLimit; return; }
    c.cols[p] = col;
}

// emit_molecule for a gene label of one or two ids (g0 < g1) held in registers - nine molecules in ten.  Returns the column
// the molecule counts for (the CALLER appends it: one reservation per wave, see append_cols) or 0xFFFFFFFF when there is
// none (dropped, or kept as a two-gene class for the EM - written here).
// cls is set when the molecule is a two-gene class (g0, g1) for the EM: the CALLER stores it, one reservation per wave
// (append_class2) - two same-address LDS atomics per molecule were queueing the whole CU behind one word.
__device__ __forceinline__ uint32_t molecule2_column(const PugCtx& c, uint32_t g0, uint32_t g1, uint32_t ng, bool& cls) {
    uint32_t col = 0xFFFFFFFFu;
    auto sua = [&](uint32_t g) { return (g & 1u) == 0 ? (g >> 1) : c.uo + (g >> 1); };
    if (ng == 1) col = c.usa ? sua(g0) : g0;
    else if (c.usa && ((g0 ^ g1) & ~1u) == 0) col = c.ao + (g0 >> 1);
    else if (c.em) {
        cls = true;
        return 0xFFFFFFFFu;
    } else if (c.usa) {
        const bool s1 = (g0 & 1u) == 0, s2 = (g1 & 1u) == 0;
        if (s1 && !s2) col = g0 >> 1;
        else if (!s1 && s2) col = g1 >> 1;
    }
    if (col != 0xFFFFFFFFu && col >= c.num_rows) { c.s_cnt[3] = kErrSlotRange; return 0xFFFFFFFFu; }
    return col;
}
// Up to four refs -> their distinct gene ids, ascending, all in registers (r[i] = 0xFFFFFFFF past the label's end).
// Returns the number of genes; gene i in g[i].  (genes_of with its 64-entry array lives in scratch memory: a load or store
// there is a trip to global memory, dozens per molecule.)
__device__ __forceinline__ uint32_t genes_of4(const PugCtx& c, uint32_t (&g)[4], uint32_t n) {
#pragma unroll
    for (int i = 0; i < 4; ++i) g[i] = (uint32_t)i < n ? (c.gene_level ? g[i] : c.t2g[g[i]]) : 0xFFFFFFFFu;
    auto cs = [&](int a, int b) { const uint32_t lo = g[a] < g[b] ? g[a] : g[b], hi = g[a] < g[b] ? g[b] : g[a]; g[a] = lo; g[b] = hi; };
    cs(0, 1); cs(2, 3); cs(0, 2); cs(1, 3); cs(1, 2);   // sorting network of four
    // drop repeats (0xFFFFFFFF sorts last and is never counted)
    uint32_t o[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    uint32_t k = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bool keep = g[i] != 0xFFFFFFFFu && (i == 0 || g[i] != g[i - 1]);
        if (keep) {
#pragma unroll
            for (int q = 0; q < 4; ++q) if ((uint32_t)q == k) o[q] = g[i];
            ++k;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) g[i] = o[i];
    return k;
}
// emit_molecule for a gene label of up to four ids in registers; like molecule2_column it returns the column (for
// append_cols) or 0xFFFFFFFF, and writes a multi-gene class for the EM itself.
__device__ __forceinline__ uint32_t molecule4_column(const PugCtx& c, const uint32_t (&g)[4], uint32_t ng, bool& cls) {
    if (ng == 0) return 0xFFFFFFFFu;
    if (ng <= 2) return molecule2_column(c, g[0], g[1], ng, cls);   // (cls: the class is (g[0], g[1]))
    uint32_t col = 0xFFFFFFFFu;
    if (c.em) {
        const uint32_t off = atomicAdd(&c.s_cnt[1], ng), di = atomicAdd(&c.s_cnt[2], 1u);
        if (off + ng > c.lab_cap || 2 * (di + 1) > c.lab_cap) { c.s_cnt[3] = kErrPugLimit; return 0xFFFFFFFFu; }
#pragma unroll
        for (int i = 0; i < 4; ++i) if ((uint32_t)i < ng) c.labw[off + i] = g[i];
        c.labd[2 * di] = off; c.labd[2 * di + 1] = ng;
        return 0xFFFFFFFFu;
    }
    if (c.usa) {   // 3..10 genes: exactly one spliced gene -> A if its unspliced partner follows it, else S (utils.rs:719-747)
        uint32_t nsp = 0, sg = 0, nxt = 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if ((uint32_t)i < ng && (g[i] & 1u) == 0) { if (nsp == 0) { sg = g[i]; nxt = i + 1 < 4 && (uint32_t)(i + 1) < ng ? g[i + 1] : 0xFFFFFFFFu; } ++nsp; }
        if (nsp == 1) col = (nxt != 0xFFFFFFFFu && ((sg ^ nxt) & ~1u) == 0) ? c.ao + (sg >> 1) : (sg >> 1);
    }
    if (col != 0xFFFFFFFFu && col >= c.num_rows) { c.s_cnt[3] = kErrSlotRange; return 0xFFFFFFFFu; }
    return col;
}
// Append one column per lane that has one.  Called by all lanes of the wave together: the lanes share ONE reservation on the
// cell's column counter - a same-address LDS atomic per molecule is serviced lane by lane, with sixteen waves queueing.
__device__ __forceinline__ void append_cols(const PugCtx& c, uint32_t col) {
    const bool has = col != 0xFFFFFFFFu;
    const uint64_t m = __ballot(has);
    if (!m) return;
    const uint32_t lane = lane_id(), leader = (uint32_t)__builtin_ctzll(m);
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(&c.s_cnt[0], (uint32_t)__popcll(m));
    base = __builtin_amdgcn_readlane(base, (int)leader);
    if (has) {
        const uint32_t p = base + (uint32_t)__popcll(m & ((1ull << lane) - 1));
        if (p >= c.cols_cap) c.s_cnt[3] = kErrPugLimit; else c.cols[p] = col;
    }
}

// Store one two-gene class per lane that has one (want).  Wave-wide call, like append_cols: one reservation of label words
// and one of descriptors per wave.
__device__ __forceinline__ void append_class2(const PugCtx& c, bool want, uint32_t g0, uint32_t g1) {
    const uint64_t m = __ballot(want);
    if (!m) return;
    const uint32_t lane = lane_id(), leader = (uint32_t)__builtin_ctzll(m), n = (uint32_t)__popcll(m);
    uint32_t off = 0, di = 0;
    if (lane == leader) { off = atomicAdd(&c.s_cnt[1], 2u * n); di = atomicAdd(&c.s_cnt[2], n); }
    off = __builtin_amdgcn_readlane(off, (int)leader);
    di = __builtin_amdgcn_readlane(di, (int)leader);
    if (want) {
        const uint32_t r = (uint32_t)__popcll(m & ((1ull << lane) - 1));
        const uint32_t o = off + 2 * r, d = di + r;
        if (o + 2 > c.lab_cap || 2 * (d + 1) > c.lab_cap) { c.s_cnt[3] = kErrPugLimit; return; }
        c.labw[o] = g0; c.labw[o + 1] = g1;
        c.labd[2 * d] = o; c.labd[2 * d + 1] = 2;
    }
}

// A class of more than kMaxGenesPerLabel genes for the EM (a read that hits a large gene family: rare, but real data has
// them), written straight into the cell's label area by ONE lane: cand(j) is the j-th ref of the arborescence's first
// label - a gene id at gene level - or 0xFFFFFFFF when it is not shared by every vertex; n, the first label's length,
// bounds the class and is what gets reserved (the label area holds one word per alignment of the cell, and every
// molecule's first vertex is a different one).  Distinct genes are kept ascending by insertion, in global memory.
template <typename Cand>
__device__ __forceinline__ void emit_wide_class(const PugCtx& c, uint32_t n, Cand&& cand) {
    const uint32_t off = atomicAdd(&c.s_cnt[1], n), di = atomicAdd(&c.s_cnt[2], 1u);
    if (off + n > c.lab_cap || 2 * (di + 1) > c.lab_cap) { c.s_cnt[3] = kErrPugLimit; return; }
    uint32_t* w = c.labw + off;
    uint32_t k = 0;
    for (uint32_t j = 0; j < n; ++j) {
        const uint32_t t = cand(j);
        if (t == 0xFFFFFFFFu) continue;
        const uint32_t gid = c.gene_level ? t : c.t2g[t];
        uint32_t lo = 0, hi = k;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (w[mid] < gid) lo = mid + 1; else hi = mid; }
        if (lo < k && w[lo] == gid) continue;
        for (uint32_t r = k; r > lo; --r) w[r] = w[r - 1];
        w[lo] = gid;
        ++k;
    }
    c.labd[2 * di] = off; c.labd[2 * di + 1] = k;
}
// the label of the vertex in slot `slot` of the gathered component records (6b): short labels travel in the record
struct RecLab { Lab l; uint32_t r[4]; };
__device__ __forceinline__ void rec_lab(const uint4* mrec, size_t slot, RecLab& o) {
    const uint4 qa = mrec[2 * slot], qb = mrec[2 * slot + 1];
    o.l.n = qa.y;
    if (qa.y <= 4) { o.r[0] = qa.z; o.r[1] = qa.w; o.r[2] = qb.x; o.r[3] = qb.y; o.l.p = o.r; }
    else o.l.p = reinterpret_cast<const uint32_t*>((uintptr_t)(((uint64_t)qa.w << 32) | qa.z));
}
__device__ __forceinline__ bool rec_contains(const uint4& qa, const uint4& qb, uint32_t t) {
    if (qa.y <= 4) return t == qa.z || t == qa.w || t == qb.x || t == qb.y;   // (unused places hold 0xFFFFFFFF: no ref)
    const Lab l{reinterpret_cast<const uint32_t*>((uintptr_t)(((uint64_t)qa.w << 32) | qa.z)), qa.y};
    return lab_contains(l, t);
}
// cand() of emit_wide_class for a component held in those records: ref j of the vertex in slot b0 + fv if every vertex
// of `mask` (bit i = slot b0 + i) has it.  (Everything out of the records' own words: a RecLab points into a private array, which
// puts it - and 72 bytes of every lane of the calling kernel - into scratch memory.)
__device__ __forceinline__ void emit_wide_from_records(const PugCtx& c, const uint4* mrec, size_t b0, uint32_t fv, uint64_t mask) {
    const uint4 fa = mrec[2 * (b0 + fv)], fb = mrec[2 * (b0 + fv) + 1];
    const uint32_t* fp = fa.y > 4 ? reinterpret_cast<const uint32_t*>((uintptr_t)(((uint64_t)fa.w << 32) | fa.z)) : nullptr;
    emit_wide_class(c, fa.y, [&](uint32_t j) -> uint32_t {
        const uint32_t t = fp ? fp[j] & 0x7FFFFFFFu : (j == 0 ? fa.z : j == 1 ? fa.w : j == 2 ? fb.x : fb.y);
        for (uint64_t m = mask; m; m &= m - 1) {
            const uint32_t i = (uint32_t)__builtin_ctzll(m);
            if (i == fv) continue;
            if (!rec_contains(mrec[2 * (b0 + i)], mrec[2 * (b0 + i) + 1], t)) return 0xFFFFFFFFu;
        }
        return t;
    });
}

__device__ __forceinline__ uint64_t wave_or64(uint64_t v) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { lo |= __shfl_xor(lo, d); hi |= __shfl_xor(hi, d); }
    return ((uint64_t)hi << 32) | lo;
}

    // ---- 6b'. components of 3..8 vertices: EIGHT to a wave, a group of eight lanes each ----
    // The cover of 6b on segments of the wave: a ballot is cut to the group's byte, the OR over the frontier's adjacency
    // rows runs over the group's eight lanes, "the label of vertex v" is a shuffle from lane (group base + v), and every
    // loop runs until the last group of the wave is done with it (idle groups are predicated off).  Such components are
    // four in five of all that reach the cover; a wave to each left 59 of its 64 lanes without a vertex.
// The covers run in one of three ways (MODE):
//   kCoverOrdered  the records of a component lie in the reference's vertex order (class by first appearance, then UMI): ties
//                  between equal-size arborescences go to the first one met, as in the reference (afq_pug.hip; cover_big).
//   kCoverDefer    the records lie in NO particular order (afq_pug2.hip's k_p2_cover).  A round whose largest arborescence is
//                  unique does not depend on the order - same winner, same molecule, same vertices left; at the first round that
//                  meets two different vertex sets of the largest size the component is SET ASIDE: its list index and the mask
//                  of its uncovered vertices go to `tied` (four words per entry behind a counter), nothing is emitted for it
//                  from that round on.
//   kCoverResume   the set-aside components, their records meanwhile put into the reference's order by k_p2_tied: `tied` is
//                  the list (n_list entries), a component starts from the uncovered vertices its entry holds.
constexpr int kCoverOrdered = 0, kCoverDefer = 1, kCoverResume = 2;
// Labels of 5..kStageRefs refs are STAGED in LDS for the time their component is covered (`stage`: 64 x kStageRefs words per wave, or
// nullptr): the covers ask "does the label of vertex v hold ref t" and "what is ref j of the label of vertex v" once per candidate
// start, ref and round - through global memory each question was a dependent load (a binary search: four), one after the other, and
// on reads with long labels (the label-tail model) that chain was the cover kernel: 52 of a 287 ms step.  Short labels travel in the
// record as before; longer ones than kStageRefs stay in the chunk.
constexpr uint32_t kStageRefs = 16;
constexpr uint32_t kStageWordsGL = 64 * kStageRefs + 8 * kMaxGenesPerLabel;   // a wave's stage with the gene rows behind it (the GL instances of the covers)
__device__ __forceinline__ bool stage_contains(const uint32_t* row, uint32_t n, uint32_t t) {   // row: n <= kStageRefs refs, ascending, in LDS
    uint32_t lo = 0, hi = n;
#pragma unroll
    for (int step = 0; step < 5; ++step) {   // (2^5 > kStageRefs)
        if (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (row[mid] < t) lo = mid + 1; else hi = mid; }
    }
    return lo < n && row[lo] == t;
}
static_assert(kStageRefs <= 32, "stage_contains makes five halving steps");
// tied: kCoverDefer: [0] the list's counter, entries from tied + 4 (this list's region); kCoverResume: the first entry.
// GL: the genes of a molecule whose first label has more refs than the stage holds are gathered and sorted in LDS - eight rows of
// kMaxGenesPerLabel words behind the stage's 64 x kStageRefs, a row per group - not in a private array of the group's first lane: that
// array is indexed at run time and therefore lives in scratch memory (336 bytes per lane of every kernel that calls this, set up per dispatch).
template <int NWAVES, int MODE = kCoverOrdered, bool GL = false>
__device__ __forceinline__ void cover_tiny8(const PugCtx& C, const uint4* mrec, const uint32_t* mid_off, uint32_t n_tiny, uint32_t wv, uint32_t lane,
                                            uint32_t* tied_cnt = nullptr, uint32_t* tied = nullptr, uint32_t* stage = nullptr, uint32_t ci_base = 0,
                                            const uint16_t* idx = nullptr)
    // (ci_base, kCoverDefer: what a set-aside component's index is counted from - the caller covers a slice of a cell's list, mid_off points
    //  at the slice; idx: the n_tiny components to take, as indices into the slice - the ones cover_lane4 below left over)
    {
        const uint32_t gl = lane & 7u, gbase = lane & ~7u, grp = lane >> 3;
        auto seg_or8 = [](uint32_t x) -> uint32_t { x |= (uint32_t)__shfl_xor((int)x, 1); x |= (uint32_t)__shfl_xor((int)x, 2); x |= (uint32_t)__shfl_xor((int)x, 4); return x; };
        for (uint32_t c0 = wv * 8; c0 < n_tiny; c0 += (NWAVES) * 8) {   // (kCoverResume: n_tiny = entries of the list)
            const bool gvalid = c0 + grp < n_tiny;
            uint32_t ci = c0 + grp;
            if (idx) ci = gvalid ? idx[c0 + grp] : 0u;
            uint32_t uc0 = 0xFFu;
            if constexpr (MODE == kCoverResume) { const uint32_t e = ci; ci = gvalid ? tied[4 * e] : 0u; uc0 = gvalid ? tied[4 * e + 1] & 0xFFu : 0u; }   // (idx: the list ENTRIES to take)
            const uint32_t b0 = gvalid ? mid_off[ci] : 0u, n = gvalid ? mid_off[ci + 1] - b0 : 0u;
            const bool act = gl < n;
            uint4 qa = make_uint4(0, 0, 0, 0), qb = qa;
            if (act) { qa = mrec[2 * (size_t)(b0 + gl)]; qb = mrec[2 * (size_t)(b0 + gl) + 1]; }
            Lab myl{nullptr, act ? qa.y : 0u};
            uint32_t lr0 = 0xFFFFFFFFu, lr1 = 0xFFFFFFFFu, lr2 = 0xFFFFFFFFu, lr3 = 0xFFFFFFFFu;
            if (act && myl.n <= 4) { lr0 = qa.z; lr1 = qa.w; lr2 = qb.x; lr3 = qb.y; }
            else if (act) myl.p = reinterpret_cast<const uint32_t*>((uintptr_t)(((uint64_t)qa.w << 32) | qa.z));
            const bool staged = stage && act && myl.n > 4 && myl.n <= kStageRefs;   // (decided by the label's length alone: any lane can tell for any other)
            if (stage && __any(staged)) {
                __builtin_amdgcn_wave_barrier();   // (the rows of the batch before are no longer read)
                uint32_t tmp[kStageRefs];
#pragma unroll
                for (uint32_t q = 0; q < kStageRefs; ++q) tmp[q] = staged && q < myl.n ? myl.p[q] & 0x7FFFFFFFu : 0xFFFFFFFFu;   // (the loads go out together)
                if (staged) {
#pragma unroll
                    for (uint32_t q = 0; q < kStageRefs; ++q) stage[lane * kStageRefs + q] = tmp[q];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            auto my_contains = [&](uint32_t t) -> bool {
                if (myl.n <= 4) return t == lr0 || t == lr1 || t == lr2 || t == lr3;
                if (staged) return stage_contains(stage + lane * kStageRefs, myl.n, t);
                return lab_contains(myl, t);
            };
            // ref j of the label held by lane `src` (every lane of the wave makes the same three shuffles; n_v, j and src are the caller's)
            auto ref_of = [&](uint32_t src, uint32_t n_v, uint32_t j, bool on) -> uint32_t {
                const uint32_t r = (uint32_t)__shfl((int)(j == 0 ? lr0 : j == 1 ? lr1 : j == 2 ? lr2 : lr3), (int)src);
                const uint64_t pa = (uint64_t)(uintptr_t)myl.p;
                const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)pa, (int)src), hi = (uint32_t)__shfl((int)(uint32_t)(pa >> 32), (int)src);
                if (n_v <= 4 || !on) return r;
                if (stage && n_v <= kStageRefs) return stage[src * kStageRefs + j];   // (src's label is staged: its length says so)
                return reinterpret_cast<const uint32_t*>((uintptr_t)(((uint64_t)hi << 32) | lo))[j] & 0x7FFFFFFFu;
            };
            uint32_t adj = act ? (qb.z & 0xFFu) : 0u;   // local indices < 8
            if (C.adj_umi) {   // (wave-uniform) the record holds (UMI, reads): my out-neighbours are the group's vertices whose label shares a ref with mine and that umi_edge lets me reach
                adj = 0;
                for (uint32_t k = 0; k < 8; ++k) {
                    const uint32_t uk = (uint32_t)__shfl((int)qb.z, (int)(gbase + k)), ck = (uint32_t)__shfl((int)qb.w, (int)(gbase + k));
                    const uint32_t lkn_all = (uint32_t)__shfl((int)myl.n, (int)(gbase + k));
                    const uint32_t lkn = k < n ? lkn_all : 0u;
                    bool ov = false;
                    for (uint32_t j = 0; __any(j < lkn); ++j) {
                        const bool on = j < lkn;
                        const uint32_t t = ref_of(gbase + k, lkn, j, on);
                        ov = ov || (on && act && my_contains(t));
                    }
                    if (act && k < n && k != gl && ov && umi_edge(qb.z, qb.w, uk, ck, C.exact_umi)) adj |= 1u << k;
                }
            }
            uint32_t UC = gvalid ? ((1u << n) - 1u) & uc0 : 0u;
            while (__any(UC != 0)) {
                const uint32_t remaining = (uint32_t)__popc(UC);
                uint32_t best = 0, best_sz = 0, it = UC;
                bool tie = false;   // (kCoverDefer) two different vertex sets of the largest size so far
                while (__any(it != 0)) {   // candidate start vertices, ascending
                    const bool g_on = it != 0;
                    const uint32_t v = g_on ? (uint32_t)__builtin_ctz(it) : 0u;
                    it &= it - 1;   // (0 stays 0)
                    const uint32_t lvn_all = (uint32_t)__shfl((int)myl.n, (int)(gbase + v));
                    const uint32_t lvn = g_on ? lvn_all : 0u;
                    uint32_t mv = 0, mv_sz = 0;
                    bool tie_v = false;
                    for (uint32_t j = 0; __any(j < lvn); ++j) {
                        const bool on = j < lvn;
                        const uint32_t t = ref_of(gbase + v, lvn, j, on);
                        const bool has = on && act && ((UC >> gl) & 1u) && my_contains(t);
                        const uint32_t At = (uint32_t)(__ballot(has) >> gbase) & 0xFFu;
                        uint32_t Rm = 1u << v, F = on ? Rm : 0u;
                        while (__any(F != 0)) {
                            const uint32_t N = seg_or8(((F >> gl) & 1u) ? adj : 0u);
                            F = N & At & ~Rm;
                            Rm |= F;
                        }
                        const uint32_t sz = (uint32_t)__popc(Rm);
                        if (on && sz > mv_sz) { mv_sz = sz; mv = Rm; tie_v = false; }
                        else if (MODE == kCoverDefer && on && sz == mv_sz && Rm != mv) tie_v = true;
                    }
                    if (g_on && mv_sz > best_sz) { best_sz = mv_sz; best = mv; tie = tie_v; }
                    else if (MODE == kCoverDefer && g_on && mv_sz == best_sz && (mv != best || tie_v)) tie = true;
                    if (g_on && mv_sz == remaining) it = 0;
                }
                if constexpr (MODE == kCoverDefer) {
                    if (tie && best != 0) {   // set aside (every lane of the group holds the same tie / UC)
                        if (gl == 0) { const uint32_t e = atomicAdd(tied_cnt, 1u); tied[4 * e] = ci + ci_base; tied[4 * e + 1] = UC; tied[4 * e + 2] = 0u; }
                        UC = 0; best = 0;
                    }
                }
                const bool g_emit = UC != 0;
                if (g_emit && best == 0) { if (gl == 0) C.s_cnt[3] = kErrPugLimit; UC = 0; }  // vertex with an empty label
                // transcripts common to every vertex of the arborescence (pugutils.rs:1161-1188) -> genes
                const uint32_t fv = best ? (uint32_t)__builtin_ctz(best) : 0u;
                const uint32_t lfn_all = (uint32_t)__shfl((int)myl.n, (int)(gbase + fv));
                const uint32_t lfn = best ? lfn_all : 0u;
                uint32_t gpriv[GL ? 1 : kMaxGenesPerLabel];   // (only for unstaged labels over four refs; !GL: the array lives in scratch memory)
                uint32_t* const g = GL ? stage + 64 * kStageRefs + grp * kMaxGenesPerLabel : gpriv;
                uint32_t ng = 0, k4 = 0;
                uint32_t c4[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
                bool wide = false, in_row = false;
                const bool small = lfn <= 4;
                // (a staged first label: the refs every vertex holds are only MARKED in the loop - cm - and their genes looked up by the
                //  group's lanes together afterwards; one lane looking them up inside the loop made one dependent load per ref)
                const bool fstaged = stage && lfn > 4 && lfn <= kStageRefs && !C.gene_level;
                uint32_t cm = 0;
                for (uint32_t j = 0; __any(j < lfn); ++j) {
                    const bool on = j < lfn;
                    const uint32_t t = ref_of(gbase + fv, lfn, j, on);
                    const uint32_t hasm = (uint32_t)(__ballot(on && act && ((best >> gl) & 1u) && my_contains(t)) >> gbase) & 0xFFu;
                    if (on && hasm == best && fstaged) cm |= 1u << j;
                    else if (on && hasm == best && gl == 0) {
                        if (small) {
#pragma unroll
                            for (int w = 0; w < 4; ++w) if ((uint32_t)w == k4) c4[w] = t;
                            ++k4;
                        } else {
                            const uint32_t gid = C.gene_level ? t : C.t2g[t];
                            uint32_t q = 0;
                            while (q < ng && g[q] < gid) ++q;
                            if (!(q < ng && g[q] == gid)) {
                                if (ng == kMaxGenesPerLabel) wide = true;
                                else { for (uint32_t r = ng; r > q; --r) g[r] = g[r - 1]; g[q] = gid; ++ng; }
                            }
                        }
                    }
                }
                if (stage && __any(fstaged)) {   // the marked refs' genes: lane gl of the group takes refs gl and gl + 8 of the first label's row, in place
                    uint32_t* row = stage + (gbase + fv) * kStageRefs;
                    uint32_t ga = 0xFFFFFFFFu, gb2 = 0xFFFFFFFFu;
                    if (fstaged && ((cm >> gl) & 1u)) ga = C.t2g[row[gl]];
                    if (fstaged && ((cm >> (gl + 8)) & 1u)) gb2 = C.t2g[row[gl + 8]];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if (fstaged) { row[gl] = ga; row[gl + 8] = gb2; }   // (the first vertex is covered from this round on: its row is free)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if (fstaged && gl == 0) { ng = sort_unique_in_row(row, lfn); in_row = true; }
                }
                uint32_t col = 0xFFFFFFFFu;
                bool cls = false;
                if (best && gl == 0) {
                    if (small) { const uint32_t n4 = genes_of4(C, c4, k4); col = molecule4_column(C, c4, n4, cls); }
                    else if (in_row) col = molecule_column_n(C, stage + (gbase + fv) * kStageRefs, ng);
                    else if (wide && C.em) emit_wide_from_records(C, mrec, b0, fv, best);
                    else emit_molecule(C, g, wide ? 0xFFFFFFFFu : ng);
                }
                append_cols(C, col);
                append_class2(C, cls, c4[0], c4[1]);
                UC &= ~best;
            }
        }
    }

// ---- components of 3..4 vertices whose labels hold at most four refs each: ONE LANE per component ----
// Nine in ten of the components that reach the covers (a molecule read under two or three UMIs one base apart; labels of one or two
// transcripts, which travel in the records).  cover_tiny8 gives such a component eight lanes, five of them idle, and pays a ballot
// or a shuffle for every "who has ref t" / "whose label is it"; here the whole component - four labels of four refs, four
// adjacency nibbles - sits in one lane's registers, every loop is unrolled over constant indices, and a wave covers 64 components
// at a time.  Same rule, step for step (pugutils.rs:1094-1188): candidates ascending, refs in label order, the first largest
// arborescence wins; kCoverDefer sets a component aside at the first round with two different largest vertex sets.
// Wave-wide call: lane's component = records b0 .. b0 + n - 1 (n = 0: none), uc0 = the vertices still uncovered, ci = its index for
// the set-aside list.  (kCoverResume: uc0 comes from the list, no tie is tracked - the records are in the reference's order.)
struct Lane4 {   // a component of up to four vertices in one lane's registers
    uint32_t rf[4][4];   // the labels' refs (0xFFFFFFFF past a label's end)
    uint32_t ln[4];      // the labels' lengths (<= 4)
    uint32_t um[4], rc[4];   // records with (UMI, reads): those; else the adjacency mask's words
    uint32_t adjm;       // out-neighbours of vertex v: bits 4 v .. 4 v + 3
};
__device__ __forceinline__ void lane4_load(const PugCtx& C, const uint4* mrec, uint32_t b0, uint32_t n, Lane4& L) {
    constexpr uint32_t kNo = 0xFFFFFFFFu;
    L.adjm = 0;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        uint4 qa = make_uint4(0, 0, kNo, kNo), qb = make_uint4(kNo, kNo, 0, 0);
        if ((uint32_t)v < n) { qa = mrec[2 * (size_t)(b0 + v)]; qb = mrec[2 * (size_t)(b0 + v) + 1]; }
        L.ln[v] = qa.y; L.rf[v][0] = qa.z; L.rf[v][1] = qa.w; L.rf[v][2] = qb.x; L.rf[v][3] = qb.y;
        L.um[v] = qb.z; L.rc[v] = qb.w;
        L.adjm |= (qb.z & 0xFu) << (4 * v);
    }
    if (C.adj_umi) {   // (wave-uniform) the records hold (UMI, reads) in place of the mask: the edges from umi_edge and the labels
        L.adjm = 0;
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int u = v + 1; u < 4; ++u) {
                bool ov = false;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t t = L.rf[v][j];
                    ov = ov || ((uint32_t)j < L.ln[v] && (t == L.rf[u][0] || t == L.rf[u][1] || t == L.rf[u][2] || t == L.rf[u][3]));
                }
                ov = ov && (uint32_t)u < n;
                if (ov && umi_edge(L.um[v], L.rc[v], L.um[u], L.rc[u], C.exact_umi)) L.adjm |= 1u << (4 * v + u);
                if (ov && umi_edge(L.um[u], L.rc[u], L.um[v], L.rc[v], C.exact_umi)) L.adjm |= 1u << (4 * u + v);
            }
    }
}
// The vertices of L renumbered: vertex v becomes vertex rank[v] (a permutation of 0..3; vertices past the component's end keep
// their places) - labels, adjacency rows and columns, and the mask uc.  Every index a constant: selects, no memory.
__device__ __forceinline__ void lane4_permute(Lane4& L, const uint32_t (&rank)[4], uint32_t& uc) {
    Lane4 P;
    P.adjm = 0;
    uint32_t nuc = 0;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        P.ln[p] = 0; P.um[p] = 0; P.rc[p] = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) P.rf[p][j] = 0xFFFFFFFFu;
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        uint32_t row = 0;   // v's out-neighbours under the new numbers
#pragma unroll
        for (int u = 0; u < 4; ++u) row |= ((L.adjm >> (4 * v + u)) & 1u) << rank[u];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (rank[v] != (uint32_t)p) continue;
            P.ln[p] = L.ln[v]; P.um[p] = L.um[v]; P.rc[p] = L.rc[v];
#pragma unroll
            for (int j = 0; j < 4; ++j) P.rf[p][j] = L.rf[v][j];
            P.adjm |= row << (4 * p);
        }
        nuc |= ((uc >> v) & 1u) << rank[v];
    }
    L = P; uc = nuc;
}
// The cover rounds of the component in L from the uncovered vertices UC on.  Wave-wide call (a lane without a component: UC = 0).
template <int MODE>
__device__ __forceinline__ void lane4_rounds(const PugCtx& C, const Lane4& L, uint32_t UC, uint32_t ci, uint32_t* tied_cnt, uint32_t* tied) {
    constexpr uint32_t kNo = 0xFFFFFFFFu;
    auto has = [&](int u, uint32_t t) -> bool { return t == L.rf[u][0] || t == L.rf[u][1] || t == L.rf[u][2] || t == L.rf[u][3]; };
    while (__any(UC != 0)) {
        const uint32_t remaining = (uint32_t)__popc(UC);
        uint32_t best = 0, best_sz = 0;
        bool tie = false, done = false;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const bool von = !done && ((UC >> v) & 1u);
            uint32_t mv = 0, mv_sz = 0;
            bool tie_v = false;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool on = von && (uint32_t)j < L.ln[v];
                const uint32_t t = L.rf[v][j];
                uint32_t At = 0;
#pragma unroll
                for (int u = 0; u < 4; ++u) At |= (((UC >> u) & 1u) && has(u, t)) ? 1u << u : 0u;
                uint32_t Rm = 1u << v;
#pragma unroll
                for (int step = 0; step < 3; ++step) {   // (a path through four vertices has three edges)
                    uint32_t N = 0;
#pragma unroll
                    for (int x = 0; x < 4; ++x) N |= ((Rm >> x) & 1u) ? (L.adjm >> (4 * x)) & 0xFu : 0u;
                    Rm |= N & At;
                }
                const uint32_t sz = (uint32_t)__popc(Rm);
                if (on && sz > mv_sz) { mv_sz = sz; mv = Rm; tie_v = false; }
                else if (MODE == kCoverDefer && on && sz == mv_sz && Rm != mv) tie_v = true;
            }
            if (von && mv_sz > best_sz) { best_sz = mv_sz; best = mv; tie = tie_v; }
            else if (MODE == kCoverDefer && von && mv_sz == best_sz && (mv != best || tie_v)) tie = true;
            if (von && mv_sz == remaining) done = true;
        }
        if constexpr (MODE == kCoverDefer) {   // set aside: one reservation per wave
            const bool aside = UC != 0 && tie && best != 0;
            const uint64_t m = __ballot(aside);
            if (m) {
                const uint32_t lane = lane_id(), leader = (uint32_t)__builtin_ctzll(m);
                uint32_t e0 = 0;
                if (lane == leader) e0 = atomicAdd(tied_cnt, (uint32_t)__popcll(m));
                e0 = (uint32_t)__builtin_amdgcn_readlane((int)e0, (int)leader);
                if (aside) {
                    const uint32_t e = e0 + (uint32_t)__popcll(m & ((1ull << lane) - 1));
                    tied[4 * e] = ci; tied[4 * e + 1] = UC; tied[4 * e + 2] = 0u;
                    UC = 0; best = 0;
                }
            }
        }
        if (UC != 0 && best == 0) { C.s_cnt[3] = kErrPugLimit; UC = 0; }   // a vertex with an empty label
        // the refs every vertex of the arborescence holds (pugutils.rs:1161-1188) -> genes -> the molecule's column or class
        const uint32_t fv = best ? (uint32_t)__builtin_ctz(best) : 0u;
        const uint32_t lfn = best ? (fv == 0 ? L.ln[0] : fv == 1 ? L.ln[1] : fv == 2 ? L.ln[2] : L.ln[3]) : 0u;
        uint32_t c4[4] = {kNo, kNo, kNo, kNo}, k4 = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t t = fv == 0 ? L.rf[0][j] : fv == 1 ? L.rf[1][j] : fv == 2 ? L.rf[2][j] : L.rf[3][j];
            bool all = (uint32_t)j < lfn;
#pragma unroll
            for (int u = 0; u < 4; ++u) all = all && (!((best >> u) & 1u) || has(u, t));
            if (all) {
#pragma unroll
                for (int w = 0; w < 4; ++w) if ((uint32_t)w == k4) c4[w] = t;
                ++k4;
            }
        }
        uint32_t col = kNo;
        bool cls = false;
        if (best) { const uint32_t n4 = genes_of4(C, c4, k4); col = molecule4_column(C, c4, n4, cls); }
        append_cols(C, col);
        append_class2(C, cls, c4[0], c4[1]);
        UC &= ~best;
    }
}
template <int MODE>
__device__ __forceinline__ void cover_lane4(const PugCtx& C, const uint4* mrec, uint32_t b0, uint32_t n, uint32_t uc0, uint32_t ci,
                                            uint32_t* tied_cnt = nullptr, uint32_t* tied = nullptr) {
    Lane4 L;
    lane4_load(C, mrec, b0, n, L);
    lane4_rounds<MODE>(C, L, n ? uc0 & ((1u << n) - 1u) : 0u, ci, tied_cnt, tied);
}

// ---- components of 9..64 vertices (entries n_tiny.. of the list): one wave each, adjacency = one 64-bit mask per lane ----
// (kCoverResume: the components are entries n_tiny.. n_mid - 1 of the LIST `tied`, i.e. call it with n_tiny = 0, n_mid = entries)
template <int NWAVES, int MODE = kCoverOrdered, bool GL = false>
__device__ __forceinline__ void cover_wave64(const PugCtx& C, const uint4* mrec, const uint32_t* mid_off, uint32_t n_tiny, uint32_t n_mid, uint32_t wv, uint32_t lane,
                                             uint32_t* tied_cnt = nullptr, uint32_t* tied = nullptr, uint32_t* stage = nullptr, uint32_t ci_base = 0)
    {
      // which component entry e of the walk is, and where its slots begin and end
      auto comp_of = [&](uint32_t e) -> uint32_t { if constexpr (MODE == kCoverResume) return tied[4 * e]; else return e; };
      // offsets two components ahead, records one ahead
      uint32_t ob0 = 0, ob1 = 0, nb0 = 0, nb1 = 0;
      if (n_tiny + wv < n_mid) { const uint32_t cc = comp_of(n_tiny + wv); ob0 = mid_off[cc]; ob1 = mid_off[cc + 1]; }
      if (n_tiny + wv + NWAVES < n_mid) { const uint32_t cc = comp_of(n_tiny + wv + NWAVES); nb0 = mid_off[cc]; nb1 = mid_off[cc + 1]; }
      uint4 ra = make_uint4(0, 0, 0, 0), rb = make_uint4(0, 0, 0, 0);
      if (lane < ob1 - ob0) { ra = mrec[2 * (size_t)(ob0 + lane)]; rb = mrec[2 * (size_t)(ob0 + lane) + 1]; }
    for (uint32_t ci = n_tiny + wv; ci < n_mid; ci += NWAVES) {   // components of 9..64 vertices: a wave each
        const uint32_t n = ob1 - ob0;
        const uint32_t cb0 = ob0;   // this component's first slot
        const bool act = lane < n;
        const uint4 qa = ra, qb = rb;
        {   // next component's records, the one after's offsets
            ob0 = nb0; ob1 = nb1;
            ra = make_uint4(0, 0, 0, 0); rb = ra;
            if (ci + NWAVES < n_mid && lane < ob1 - ob0) { ra = mrec[2 * (size_t)(ob0 + lane)]; rb = mrec[2 * (size_t)(ob0 + lane) + 1]; }
            const uint32_t c2 = ci + 2 * (NWAVES);
            const uint32_t cc = c2 < n_mid ? comp_of(c2) : 0u;
            nb0 = c2 < n_mid ? mid_off[cc] : 0u; nb1 = c2 < n_mid ? mid_off[cc + 1] : 0u;
        }
        Lab myl{nullptr, act ? qa.y : 0u};
        uint32_t lr0 = 0xFFFFFFFFu, lr1 = 0xFFFFFFFFu, lr2 = 0xFFFFFFFFu, lr3 = 0xFFFFFFFFu;
        if (act && myl.n <= 4) { lr0 = qa.z; lr1 = qa.w; lr2 = qb.x; lr3 = qb.y; }
        else if (act) myl.p = reinterpret_cast<const uint32_t*>((uintptr_t)(((uint64_t)qa.w << 32) | qa.z));
        const bool staged = stage && act && myl.n > 4 && myl.n <= kStageRefs;
        if (stage && __any(staged)) {
            __builtin_amdgcn_wave_barrier();   // (the rows of the component before are no longer read)
            uint32_t tmp[kStageRefs];
#pragma unroll
            for (uint32_t q = 0; q < kStageRefs; ++q) tmp[q] = staged && q < myl.n ? myl.p[q] & 0x7FFFFFFFu : 0xFFFFFFFFu;
            if (staged) {
#pragma unroll
                for (uint32_t q = 0; q < kStageRefs; ++q) stage[lane * kStageRefs + q] = tmp[q];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        auto my_contains = [&](uint32_t t) -> bool {
            if (myl.n <= 4) return t == lr0 || t == lr1 || t == lr2 || t == lr3;
            if (staged) return stage_contains(stage + lane * kStageRefs, myl.n, t);
            return lab_contains(myl, t);
        };
        // the label of the vertex held by lane v, out of that lane's registers (a global read only for labels over four refs):
        // the cover below walks candidate labels thousands of times per cell
        auto lane_lab_n = [&](uint32_t v) -> uint32_t { return (uint32_t)__shfl((int)myl.n, (int)v); };
        auto lane_lab_ref = [&](uint32_t v, uint32_t n_v, uint32_t j) -> uint32_t {
            if (n_v <= 4) return (uint32_t)__shfl((int)(j == 0 ? lr0 : j == 1 ? lr1 : j == 2 ? lr2 : lr3), (int)v);
            if (stage && n_v <= kStageRefs) return stage[v * kStageRefs + j];
            const uint64_t pa = (uint64_t)(uintptr_t)myl.p;
            const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)pa, (int)v), hi = (uint32_t)__shfl((int)(uint32_t)(pa >> 32), (int)v);
            return reinterpret_cast<const uint32_t*>((uintptr_t)(((uint64_t)hi << 32) | lo))[j] & 0x7FFFFFFFu;
        };
        uint64_t adj = act ? (((uint64_t)qb.w << 32) | qb.z) : 0ull;
        if (C.adj_umi) {   // (wave-uniform) as in cover_tiny8: the record holds (UMI, reads), the edges are worked out here
            adj = 0;
            for (uint32_t k = 0; k < n; ++k) {
                const uint32_t uk = (uint32_t)__shfl((int)qb.z, (int)k), ck = (uint32_t)__shfl((int)qb.w, (int)k);
                const uint32_t lkn = lane_lab_n(k);
                bool ov = false;
                for (uint32_t j = 0; j < lkn; ++j) { const uint32_t t = lane_lab_ref(k, lkn, j); ov = ov || (act && my_contains(t)); }
                if (act && k != lane && ov && umi_edge(qb.z, qb.w, uk, ck, C.exact_umi)) adj |= 1ull << k;
            }
        }
        uint64_t UC = n == 64 ? ~0ull : ((1ull << n) - 1);
        if constexpr (MODE == kCoverResume) UC &= ((uint64_t)tied[4 * ci + 2] << 32) | tied[4 * ci + 1];
        while (UC) {
            const uint32_t remaining = (uint32_t)__popcll(UC);
            uint64_t best = 0;
            uint32_t best_sz = 0;
            bool tie = false;   // (kCoverDefer) two different vertex sets of the largest size so far
            for (uint64_t it = UC; it; it &= it - 1) {   // ascending vertex id
                const uint32_t v = (uint32_t)__builtin_ctzll(it);
                const uint32_t lvn = lane_lab_n(v);
                uint64_t mv = 0;
                uint32_t mv_sz = 0;
                bool tie_v = false;
                for (uint32_t j = 0; j < lvn; ++j) {
                    const uint32_t t = lane_lab_ref(v, lvn, j);
                    const uint64_t At = __ballot(act && ((UC >> lane) & 1ull) && my_contains(t));
                    uint64_t Rm = 1ull << v, F = Rm;
                    while (F) {
                        const uint64_t N = wave_or64(((F >> lane) & 1ull) ? adj : 0ull);
                        F = N & At & ~Rm;
                        Rm |= F;
                    }
                    const uint32_t sz = (uint32_t)__popcll(Rm);
                    if (sz > mv_sz) { mv_sz = sz; mv = Rm; tie_v = false; }
                    else if (MODE == kCoverDefer && sz == mv_sz && Rm != mv) tie_v = true;
                }
                if (mv_sz > best_sz) { best_sz = mv_sz; best = mv; tie = tie_v; }
                else if (MODE == kCoverDefer && mv_sz == best_sz && (mv != best || tie_v)) tie = true;
                if (mv_sz == remaining) break;
            }
            if (best == 0) { if (lane == 0) C.s_cnt[3] = kErrPugLimit; break; }  // vertex with an empty label
            if constexpr (MODE == kCoverDefer) {
                if (tie) {   // set aside (wave-uniform)
                    if (lane == 0) { const uint32_t e = atomicAdd(tied_cnt, 1u); tied[4 * e] = ci + ci_base; tied[4 * e + 1] = (uint32_t)UC; tied[4 * e + 2] = (uint32_t)(UC >> 32); }
                    break;
                }
            }
            // transcripts common to every vertex of the arborescence (pugutils.rs:1161-1188) -> genes
            const uint32_t fv = (uint32_t)__builtin_ctzll(best);
            const uint32_t lfn = lane_lab_n(fv);
            uint32_t gpriv[GL ? 1 : kMaxGenesPerLabel];   // (only for unstaged labels over four refs; !GL: the array lives in scratch memory)
            uint32_t* const g = GL ? stage + 64 * kStageRefs : gpriv;   // (GL: the first gene row behind the stage, lane 0's)
            uint32_t ng = 0, k4 = 0;
            uint32_t c4[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
            bool wide = false;
            const bool small = lfn <= 4;
            const bool fstaged = stage && lfn > 4 && lfn <= kStageRefs && !C.gene_level;   // (as in cover_tiny8: mark, then look the genes up together)
            uint32_t cm = 0;
            for (uint32_t j = 0; j < lfn; ++j) {
                const uint32_t t = lane_lab_ref(fv, lfn, j);
                const uint64_t has = __ballot(act && ((best >> lane) & 1ull) && my_contains(t));
                if (has != best) continue;
                if (fstaged) { cm |= 1u << j; continue; }
                if (lane == 0) {
                    if (small) {
#pragma unroll
                        for (int w = 0; w < 4; ++w) if ((uint32_t)w == k4) c4[w] = t;
                        ++k4;
                    } else {
                        const uint32_t gid = C.gene_level ? t : C.t2g[t];
                        uint32_t q = 0;
                        while (q < ng && g[q] < gid) ++q;
                        if (!(q < ng && g[q] == gid)) {
                            if (ng == kMaxGenesPerLabel) wide = true;
                            else { for (uint32_t r = ng; r > q; --r) g[r] = g[r - 1]; g[q] = gid; ++ng; }
                        }
                    }
                }
            }
            if (fstaged) {   // (wave-uniform)
                uint32_t* row = stage + fv * kStageRefs;
                uint32_t ga = 0xFFFFFFFFu;
                if (lane < lfn && ((cm >> lane) & 1u)) ga = C.t2g[row[lane]];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (lane < kStageRefs) row[lane] = ga;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (lane == 0) ng = sort_unique_in_row(row, lfn);
            }
            {
                uint32_t col = 0xFFFFFFFFu;
                bool cls = false;
                if (lane == 0) {
                    if (small) { const uint32_t n4 = genes_of4(C, c4, k4); col = molecule4_column(C, c4, n4, cls); }
                    else if (fstaged) col = molecule_column_n(C, stage + fv * kStageRefs, ng);
                    else if (wide && C.em) emit_wide_from_records(C, mrec, cb0, fv, best);
                    else emit_molecule(C, g, wide ? 0xFFFFFFFFu : ng);
                }
                append_cols(C, col);
                append_class2(C, cls, c4[0], c4[1]);
            }
            UC &= ~best;
        }
    }
    }

    // ---- 6c'. components of 65..4096 vertices, one at a time by the whole workgroup ----
    // The same greedy cover over multi-word masks: lane l of a wave holds mask word l, the component's adjacency is n rows of
    // nw = ceil(n / 64) words (`rows`), the uncovered set and the round's best arborescence sit in LDS (s_mask[0], s_mask[1]).
    // The candidates of a round - the uncovered vertices, ascending - are dealt to the waves; the winner is the largest
    // arborescence, the smallest vertex among equals (the reference takes the first it meets in ascending order: pugutils.rs:1090-1160).
template <int NWAVES>
__device__ __forceinline__ void cover_big(const PugCtx& C, const uint4* mrec, const uint32_t* mid_off, uint32_t first, uint32_t count,
                                          const uint32_t* rowoff, const uint64_t* rows_base, uint64_t (*s_mask)[64], uint32_t* s_bestv, uint32_t* s_bestsz,
                                          uint32_t wv, uint32_t lane)
{
    const uint32_t tid = wv * 64 + lane;
    for (uint32_t b = 0; b < count; ++b) {
        const uint32_t c0 = mid_off[first + b], n = mid_off[first + b + 1] - c0, nw = (n + 63) / 64;
        const uint64_t* rows = rows_base + rowoff[b];
        __syncthreads();
        if (tid < nw) s_mask[0][tid] = (tid + 1 < nw || (n & 63) == 0) ? ~0ull : ((1ull << (n & 63)) - 1);
        __syncthreads();
        for (;;) {
            uint32_t rem = 0;
            for (uint32_t w = 0; w < nw; ++w) rem += (uint32_t)__popcll(s_mask[0][w]);
            if (rem == 0) break;
            uint32_t my_best_sz = 0, my_best_v = 0xFFFFFFFFu;
            uint64_t my_best_word = 0;   // lane l: word l of this wave's best arborescence
            const uint64_t ucw = lane < nw ? s_mask[0][lane] : 0ull;
            uint32_t seen = 0;
            for (uint32_t w = 0; w < nw && my_best_sz != rem; ++w) {
                uint64_t bits = s_mask[0][w];
                for (; bits; bits &= bits - 1, ++seen) {
                    if (seen % (NWAVES) != wv) continue;
                    const uint32_t v = w * 64 + (uint32_t)__builtin_ctzll(bits);
                    RecLab lv;
                    rec_lab(mrec, (size_t)c0 + v, lv);
                    uint64_t mvw = 0;
                    uint32_t mv_sz = 0;
                    for (uint32_t j = 0; j < lv.l.n; ++j) {
                        const uint32_t t = lv.l.p[j] & 0x7FFFFFFFu;
                        uint64_t Aw = 0;   // uncovered vertices whose label has t: 64 vertices per step, a lane each
                        for (uint32_t cw = 0; cw < nw; ++cw) {
                            const uint32_t i = cw * 64 + lane;
                            const uint64_t ucword = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(ucw >> 32), (int)cw) << 32) | (uint32_t)__shfl((int)(uint32_t)ucw, (int)cw);
                            bool in = i < n && ((ucword >> lane) & 1ull);
                            if (in) { const uint4 qa = mrec[2 * ((size_t)c0 + i)], qb = mrec[2 * ((size_t)c0 + i) + 1]; in = rec_contains(qa, qb, t); }
                            const uint64_t word = __ballot(in);
                            if (lane == cw) Aw = word;
                        }
                        uint64_t Rw = (lane == (v >> 6)) ? (1ull << (v & 63)) : 0ull, Fw = Rw;
                        for (;;) {
                            uint64_t Nw = 0;   // OR of the rows of the frontier's vertices
                            for (uint32_t fw = 0; fw < nw; ++fw) {
                                uint64_t fb = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(Fw >> 32), (int)fw) << 32) | (uint32_t)__shfl((int)(uint32_t)Fw, (int)fw);
                                for (; fb; fb &= fb - 1) {
                                    const uint32_t x = fw * 64 + (uint32_t)__builtin_ctzll(fb);
                                    if (lane < nw) Nw |= rows[(size_t)x * nw + lane];
                                }
                            }
                            Fw = Nw & Aw & ~Rw;
                            Rw |= Fw;
                            if (!__any(Fw != 0)) break;
                        }
                        uint32_t sz = (uint32_t)__popcll(Rw);
#pragma unroll
                        for (int dd = 32; dd > 0; dd >>= 1) sz += __shfl_xor(sz, dd);
                        if (sz > mv_sz) { mv_sz = sz; mvw = Rw; }
                    }
                    if (mv_sz > my_best_sz) { my_best_sz = mv_sz; my_best_v = v; my_best_word = mvw; }
                    if (my_best_sz == rem) break;   // (everything that is left: no later candidate is larger, none of this wave's is earlier)
                }
            }
            if (lane == 0) { s_bestv[wv] = my_best_v; s_bestsz[wv] = my_best_sz; }
            __syncthreads();
            uint32_t win = 0;
            for (uint32_t w = 1; w < (uint32_t)(NWAVES); ++w)
                if (s_bestsz[w] > s_bestsz[win] || (s_bestsz[w] == s_bestsz[win] && s_bestv[w] < s_bestv[win])) win = w;
            if (s_bestsz[win] == 0) { if (tid == 0) C.s_cnt[3] = kErrPugLimit; break; }   // a vertex with an empty label
            if (wv == win && lane < nw) s_mask[1][lane] = my_best_word;
            __syncthreads();
            if (wv == 0) {   // the refs every vertex of the arborescence has (pugutils.rs:1161-1188) -> genes
                uint32_t fv = 0xFFFFFFFFu;
                for (uint32_t w = 0; w < nw && fv == 0xFFFFFFFFu; ++w) if (s_mask[1][w]) fv = w * 64 + (uint32_t)__builtin_ctzll(s_mask[1][w]);
                RecLab lf;
                rec_lab(mrec, (size_t)c0 + fv, lf);
                auto all_have = [&](uint32_t t) -> bool {   // wave-wide
                    for (uint32_t cw = 0; cw < nw; ++cw) {
                        const uint32_t i = cw * 64 + lane;
                        bool miss = i < n && ((s_mask[1][cw] >> lane) & 1ull);
                        if (miss) { const uint4 qa = mrec[2 * ((size_t)c0 + i)], qb = mrec[2 * ((size_t)c0 + i) + 1]; miss = !rec_contains(qa, qb, t); }
                        if (__any(miss)) return false;
                    }
                    return true;
                };
                uint32_t g[kMaxGenesPerLabel];
                uint32_t ng = 0;
                bool wide = false;
                for (uint32_t j = 0; j < lf.l.n; ++j) {
                    const uint32_t t = lf.l.p[j] & 0x7FFFFFFFu;
                    if (!all_have(t)) continue;
                    if (lane == 0) {
                        const uint32_t gid = C.gene_level ? t : C.t2g[t];
                        uint32_t q = 0;
                        while (q < ng && g[q] < gid) ++q;
                        if (!(q < ng && g[q] == gid)) {
                            if (ng == kMaxGenesPerLabel) wide = true;
                            else { for (uint32_t r = ng; r > q; --r) g[r] = g[r - 1]; g[q] = gid; ++ng; }
                        }
                    }
                }
                if (lane == 0) {
                    if (wide && C.em)
                        emit_wide_class(C, lf.l.n, [&](uint32_t j) -> uint32_t {
                            const uint32_t t = lf.l.p[j] & 0x7FFFFFFFu;
                            for (uint32_t cw = 0; cw < nw; ++cw)
                                for (uint64_t m = s_mask[1][cw]; m; m &= m - 1) {
                                    const size_t sl = (size_t)c0 + cw * 64 + (uint32_t)__builtin_ctzll(m);
                                    if (!rec_contains(mrec[2 * sl], mrec[2 * sl + 1], t)) return 0xFFFFFFFFu;
                                }
                            return t;
                        });
                    else emit_molecule(C, g, wide ? 0xFFFFFFFFu : ng);
                }
            }
            __syncthreads();
            if (tid < nw) s_mask[0][tid] &= ~s_mask[1][tid];
            __syncthreads();
        }
        __syncthreads();
    }
}

}  // namespace afq
