// afq_p2_shared.h — what the two files of the phase-kernel parsimony path share (afq_pug2.hip: partitions, the per-cell graph /
// cover / tie kernels; afq_pugflat.hip: the range-wide graph build): the vertex word and pair layouts, labels by key, the
// per-cell context, the coherence helpers.  Device code only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "afq_common.h"
#include "afq_kernels.h"
#include "afq_prims.h"
#include "afq_pug_common.h"

namespace afq {

typedef unsigned __int128 u128;

constexpr uint32_t kP2Bins = 2048;        // partitions per cell the tile kernels rank in LDS
constexpr uint32_t kP2TabSlots = 512;     // hash table of one partition's vertices (<= 256)
#ifndef AFQ_P2_FILT_LG
#define AFQ_P2_FILT_LG 12
#endif
constexpr uint32_t kP2FiltLg = AFQ_P2_FILT_LG, kP2FiltBits = 1u << kP2FiltLg;    // presence filter in front of it
constexpr uint32_t kVCntMask = 0x3FFu;    // vertex word: reads (10 bits) | label signature (19 bits) << 10 | key tag << 29
constexpr uint64_t kPairF = 1ull << 63, kPairB = 1ull << 62;   // pair (x, y): x -> y / y -> x is an edge

__device__ __forceinline__ uint32_t sig_of(uint32_t t) { return 1u << (t % 19u); }
__device__ __forceinline__ uint32_t fold9(uint32_t u) { u ^= u >> 18; return (u ^ (u >> 9)) & (kP2TabSlots - 1); }       // linear: fold(a ^ b) = fold(a) ^ fold(b)
__device__ __forceinline__ uint32_t fold11(uint32_t u) { return (u ^ (u >> kP2FiltLg) ^ (u >> (2 * kP2FiltLg))) & (kP2FiltBits - 1); }   // (kP2FiltLg bits of a UMI of <= 32 bits; linear too)

__device__ __forceinline__ uint32_t wg_add(uint32_t* p, uint32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ uint32_t wg_min(uint32_t* p, uint32_t v) { return __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// The graph kernel's scratch belongs to one workgroup, so its global-memory atomics are WORKGROUP scope: they execute in the
// XCD's L2 and cost no fabric traffic (as agent-scope operations the same words were 32 GB of HBM-side traffic per launch).
// Two rules keep them coherent with the plain accesses around them: a word other waves change with atomics is READ with an
// atomic too (fetch_or 0: it is answered by the L2, where a plain load may be served by a line this CU's L1 cached before
// the atomic), and plain stores to such a word are followed by gsync() before the next atomic on it.
template <typename T>
__device__ __forceinline__ T ld_l2(const T* p) { return __hip_atomic_fetch_or(const_cast<T*>(p), (T)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
template <typename T>
__device__ __forceinline__ void st_l2(T* p, T v) { *p = v; }
// The barrier between phases that hand each other data through global memory: a wave first waits for its own stores to be
// acknowledged (s_waitcnt vmcnt(0): stores count in vmcnt on gfx9), then goes to the barrier.
__device__ __forceinline__ void gsync() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }

// ---- labels by key: one or two refs sit in the key itself (afq_common.h label_key), longer ones in the chunk -------------
struct KLab {
    uint32_t n;          // refs
    uint32_t r0, r1;     // tags 1, 2
    const uint32_t* p;   // tag 3: the refs in the chunk (orientation bit still on)
};
__device__ __forceinline__ KLab klab(const uint32_t* W, uint32_t HW, uint64_t h, uint32_t off) {
    KLab l{0, 0xFFFFFFFFu, 0xFFFFFFFFu, nullptr};
    const uint32_t tag = (uint32_t)(h >> 62);
    if (tag == 1) { l.n = 1; l.r0 = (uint32_t)h & 0x7FFFFFFFu; }
    else if (tag == 2) { l.n = 2; l.r0 = (uint32_t)(h >> 31) & 0x7FFFFFFFu; l.r1 = (uint32_t)h & 0x7FFFFFFFu; }
    else if (tag == 3) { l.n = W[off]; l.p = W + off + HW; }
    return l;
}
__device__ __forceinline__ uint32_t klab_ref(const KLab& l, uint32_t j) { return l.p ? (l.p[j] & 0x7FFFFFFFu) : (j == 0 ? l.r0 : l.r1); }
__device__ __forceinline__ bool klab_contains(const KLab& l, uint32_t t) {
    if (!l.p) return t == l.r0 || t == l.r1;   // (t is a ref id < 2^31, never the 0xFFFFFFFF filler)
    return lab_contains(Lab{l.p, l.n}, t);
}
__device__ __forceinline__ bool klab_overlap(const KLab& a, const KLab& b) {   // share >= 1 ref (pugutils.rs:187-204)
    if (a.n == 0 || b.n == 0) return false;
    if (!a.p) return klab_contains(b, a.r0) || (a.n > 1 && klab_contains(b, a.r1));
    if (!b.p) return klab_contains(a, b.r0) || (b.n > 1 && klab_contains(a, b.r1));
    return lab_overlap(Lab{a.p, a.n}, Lab{b.p, b.n});
}

__device__ __forceinline__ PugCtx make_ctx(const P2Args& A, const P2Cell& c, uint32_t* cnt) {
    PugCtx C;
    C.W = reinterpret_cast<const uint32_t*>(A.bytes + c.chunk_off);
    C.HW = A.hw; C.t2g = A.t2g; C.ref_count = A.ref_count; C.num_genes = A.num_genes;
    C.usa = A.usa; C.num_rows = A.num_rows; C.uo = A.num_rows / 3; C.ao = 2 * (A.num_rows / 3); C.em = A.em;
    C.exact_umi = A.exact_umi; C.large_thresh = A.large_thresh; C.umi_pairs = A.umi_pairs; C.gene_level = 0;
    C.cols = reinterpret_cast<uint32_t*>(A.keys0 + c.key_off);
    C.cols_cap = 2 * c.n_ref + 2;
    C.labw = A.lab ? A.lab + 2 * c.key_off : nullptr;
    C.labd = A.lab ? C.labw + c.n_ref + 1 : nullptr;
    C.lab_cap = c.n_ref + 1;
    C.s_cnt = cnt; C.st = A.st; C.cell = c.cell; C.adj_umi = 0;
    return C;
}

// (LDS instructions of one wave execute in order; WAVE_SYNC only keeps the compiler from moving code across.)
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

// The two-vertex components of one cell: one molecule each, the refs both labels share (pugutils.rs:1161-1188).  The NT threads of
// the workgroup take the n_pr components of pr_v (two vertices each; tl: touched-vertex number -> slot, or nullptr when the list
// holds slots); stage_wave: this wave's 64 x kStageRefs words of LDS.  Workgroup-wide call.
// (GL: as in cover_tiny8 - a first label of more refs than the stage holds has its shared refs' genes sorted in the first gene row
//  behind the wave's stage, its lanes one after the other, not in a private array.)
template <int NT, bool GL = false>
__device__ __forceinline__ void p2_cover_pairs(const PugCtx& C, const uint64_t* ch, const uint32_t* coff, const uint32_t* tl, const uint32_t* pr_v, uint32_t n_pr, uint32_t* stage_wave) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    for (uint32_t k = tid; k - lane < n_pr; k += NT) {   // (wave-uniform trip count: append_cols is a wave-wide call)
        uint32_t col = 0xFFFFFFFFu, k0 = 0, k1 = 0;
        bool cls = false;
        KLab l{0, 0xFFFFFFFFu, 0xFFFFFFFFu, nullptr}, l2 = l;
        if (k < n_pr) {
            const uint32_t ga = tl ? tl[pr_v[2 * k]] : pr_v[2 * k], gb = tl ? tl[pr_v[2 * k + 1]] : pr_v[2 * k + 1];
            l = klab(C.W, C.HW, ch[ga], coff[ga]); l2 = klab(C.W, C.HW, ch[gb], coff[gb]);
        }
        // (labels of up to kStageRefs refs out of the chunk: the second one goes to this lane's row of the wave's LDS stage, the first
        //  one's refs into registers - both as loads issued together - and "is ref t of the first label in the second" is a search in
        //  LDS; ref by ref through global memory a pair of long labels was a chain of some forty dependent loads)
        const bool lds2 = l2.p && l2.n <= kStageRefs;
        uint32_t* const row = stage_wave + lane * kStageRefs;
        if (__any(lds2)) {
            uint32_t t2[kStageRefs];
#pragma unroll
            for (uint32_t q = 0; q < kStageRefs; ++q) t2[q] = lds2 && q < l2.n ? l2.p[q] & 0x7FFFFFFFu : 0xFFFFFFFFu;
            if (lds2) {
#pragma unroll
                for (uint32_t q = 0; q < kStageRefs; ++q) row[q] = t2[q];
            }
            WAVE_SYNC();
        }
        auto in_l2 = [&](uint32_t t) -> bool { return lds2 ? stage_contains(row, l2.n, t) : klab_contains(l2, t); };
        if (k < n_pr) {
            if (l.n <= 4) {
                uint32_t g4[4];
                uint32_t kk = 0;
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    g4[qq] = 0xFFFFFFFFu;
                    if ((uint32_t)qq < l.n) {
                        const uint32_t t = klab_ref(l, qq);
                        if (in_l2(t)) {
#pragma unroll
                            for (int w = 0; w < 4; ++w) if ((uint32_t)w == kk) g4[w] = t;
                            ++kk;
                        }
                    }
                }
                const uint32_t ng = genes_of4(C, g4, kk);
                col = molecule4_column(C, g4, ng, cls);
                k0 = g4[0]; k1 = g4[1];
            } else if (l.n <= kStageRefs) {
                uint32_t t1[kStageRefs];
#pragma unroll
                for (uint32_t q = 0; q < kStageRefs; ++q) t1[q] = q < l.n ? l.p[q] & 0x7FFFFFFFu : 0xFFFFFFFFu;          // the first label's refs, together
#pragma unroll
                for (uint32_t q = 0; q < kStageRefs; ++q) t1[q] = t1[q] != 0xFFFFFFFFu && in_l2(t1[q]) ? C.t2g[t1[q]] : 0xFFFFFFFFu;   // the shared ones' genes, together
                // (sorted and distinct in the lane's row of the stage - the second label's refs there have been looked at - not in an
                //  array of the lane's own, which would live in scratch memory)
#pragma unroll
                for (uint32_t q = 0; q < kStageRefs; ++q) row[q] = t1[q];
                const uint32_t ng = sort_unique_in_row(row, kStageRefs);
                emit_molecule(C, row, ng);
            } else if (!GL) {
                uint32_t g[kMaxGenesPerLabel];
                uint32_t ng = 0;
                for (uint32_t jj = 0; jj < l.n && ng != 0xFFFFFFFFu; ++jj) {
                    const uint32_t t = l.p[jj] & 0x7FFFFFFFu;
                    if (!in_l2(t)) continue;
                    const uint32_t gid = C.t2g[t];
                    uint32_t qq = 0;
                    while (qq < ng && g[qq] < gid) ++qq;
                    if (qq < ng && g[qq] == gid) continue;
                    if (ng == kMaxGenesPerLabel) { ng = 0xFFFFFFFFu; break; }
                    for (uint32_t r = ng; r > qq; --r) g[r] = g[r - 1];
                    g[qq] = gid;
                    ++ng;
                }
                if (ng == 0xFFFFFFFFu && C.em)
                    emit_wide_class(C, l.n, [&](uint32_t jj) -> uint32_t { const uint32_t t = l.p[jj] & 0x7FFFFFFFu; return in_l2(t) ? t : 0xFFFFFFFFu; });
                else emit_molecule(C, g, ng);
            }
        }
        if constexpr (GL) {   // first labels of more than kStageRefs refs (rare): the lanes that hold one, one after the other, through one LDS gene row
            for (uint64_t lm = __ballot(k < n_pr && l.n > kStageRefs); lm; lm &= lm - 1) {
                if (lane != (uint32_t)__builtin_ctzll(lm)) continue;
                uint32_t* const g = stage_wave + 64 * kStageRefs;
                uint32_t ng = 0;
                for (uint32_t jj = 0; jj < l.n && ng != 0xFFFFFFFFu; ++jj) {
                    const uint32_t t = l.p[jj] & 0x7FFFFFFFu;
                    if (!in_l2(t)) continue;
                    const uint32_t gid = C.t2g[t];
                    uint32_t qq = 0;
                    while (qq < ng && g[qq] < gid) ++qq;
                    if (qq < ng && g[qq] == gid) continue;
                    if (ng == kMaxGenesPerLabel) { ng = 0xFFFFFFFFu; break; }
                    for (uint32_t r = ng; r > qq; --r) g[r] = g[r - 1];
                    g[qq] = gid;
                    ++ng;
                }
                if (ng == 0xFFFFFFFFu && C.em)
                    emit_wide_class(C, l.n, [&](uint32_t jj) -> uint32_t { const uint32_t t = l.p[jj] & 0x7FFFFFFFu; return in_l2(t) ? t : 0xFFFFFFFFu; });
                else emit_molecule(C, g, ng);
            }
        }
        append_cols(C, col);
        append_class2(C, cls, k0, k1);
    }
}

}  // namespace afq
