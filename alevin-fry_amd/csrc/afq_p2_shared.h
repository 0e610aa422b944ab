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
    C.s_cnt = cnt; C.st = A.st; C.cell = c.cell;
    return C;
}

}  // namespace afq
