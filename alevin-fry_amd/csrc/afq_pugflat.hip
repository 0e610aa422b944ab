// afq_pugflat.hip — the graph phase of the parsimony path as RANGE-WIDE FLAT kernels (gfx950, wave64).
//
// Same semantics as the per-cell graph kernel of afq_pug2.hip (reference paths relative to /root/reference):
//   weakly_connected_components          src/pugutils.rs:278-301
//   collapse_vertices / get_num_molecules src/pugutils.rs:308-391, 1094-1146   (the covers themselves: afq_pug_common.h)
// but cut by what the work is made of, not by the cell: rounds 3-5 gave every cell one workgroup that walked some forty
// dependent steps behind workgroup barriers (k_p2_graph: 128 VGPRs, 40 spilled, 98 KB of LDS, 39.8 GB per configs[2] step
// for a union-find).  Here every step is one kernel over the whole range of cells:
//
//   k_pf_count / k_pf_number      the vertices that have an edge (the search flagged them) get dense numbers t in [0, T), in slot
//                                 order: a tile of 4096 read slots counts its flags, the tile counts are scanned (k_pf_scan1/2),
//                                 the tiles number their vertices.  Everything behind this works on dense arrays of T entries.
//   k_pf_union                    one thread per PAIR: compare-and-swap hooking union-find on par[T] (the larger root under the
//                                 smaller, path halving by atomic min).
//   k_pf_root                     one thread per vertex: its root, its position inside its component (an atomic counter per root).
//   k_pf_cats, k_pf_scan1/2       per tile: how many of its roots head a component of two / 3..8 / 9..64 vertices, how many record
//                                 slots those take; a cell with a larger component is routed to the per-cell kernel (rare).  The
//                                 scan of the tile counts IS the layout: a cell's tiles are consecutive, so every cell's
//                                 components become a contiguous run of the range-wide lists (pairs | 3..8 | 9..64) and every
//                                 tile knows where its roots' entries go - no counter, no atomic, the same lists every run.
//   k_pf_cells                    a workgroup per cell: the descriptor the cover kernels read, the lone vertices' staged classes
//                                 into the cell's label area.
//   k_pf_alloc / k_pf_place       roots take their list entry and record slots, vertices write their 32-byte cover records: label,
//                                 UMI, reads.  The pairs are not looked at again: inside a component EVERY two vertices that pass
//                                 has_edge's rule are an edge the search found, so the covers work a component's edges out of its
//                                 vertices' UMIs, reads and labels (umi_edge, afq_pug_common.h) - a pass over the pair list with two
//                                 atomics per pair into 32-byte records (0.64 ms per launch) is what that replaces.
//
// The covers (k_p2_cover, k_p2_tied) then read per-cell descriptors exactly as the per-cell graph kernel wrote them: records in
// slot order, ties set aside (kCoverDefer).
//
// Coherence.  The eight XCDs' L2s are not coherent with each other, and a range-wide kernel's workgroups run on all of them:
// every word that several workgroups change inside ONE kernel is changed with agent-scope atomics only (they execute at the
// memory side), and never shares a kernel with plain stores to the same array.  Reads of par[] inside k_pf_union may be
// stale - a stale parent is an older ancestor, a stale "root" fails its compare-and-swap and is told the truth - everything
// else is read in a later kernel than it was written in.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>

#include "afq_common.h"
#include "afq_kernels.h"
#include "afq_prims.h"
#include "afq_pug_common.h"
#include "afq_p2_shared.h"

namespace afq {

namespace {

constexpr uint32_t kCatShift = 29, kRankMask = (1u << kCatShift) - 1u;
constexpr uint32_t kFCatPair = 1, kFCatTiny = 2, kFCatMid = 3;
// per-tile quantities (P2Args.tq, one array of nta entries each): vertices with an edge; then, of the tile's roots, components of
// two / 3..8 / 9..64 vertices and the record slots of the latter two
constexpr uint32_t kQTouched = 0, kQPr = 1, kQTiny = 2, kQMid = 3, kQSTiny = 4, kQSMid = 5;

__device__ __forceinline__ uint32_t ag_ld(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// size class of a component of n vertices; 0: not for the flat lists - more than 64 vertices, or above --large-graph-thresh (resolved
// winner-take-all, pugutils.rs:916-982): the per-cell kernel takes the cell
__device__ __forceinline__ uint32_t cat_of(uint32_t n, uint32_t large_thresh) { return n > large_thresh ? 0u : n == 2 ? kFCatPair : n <= 8 ? kFCatTiny : n <= 64 ? kFCatMid : 0u; }

// the scanned tile quantity q in front of tile i (i = n_tiles: the total): the scan inside the tile's block of 1024 + the blocks before it
__device__ __forceinline__ uint32_t pf_g(const P2Args& A, uint32_t q, uint32_t i) { return A.tq[(size_t)q * A.nta + i] + A.bq[(size_t)q * A.nba + (i >> 10)]; }

// the dense arrays of the vertices that have an edge, out of the pool (k_pf_scan2 put them there)
struct PfV {
    uint32_t T;
    uint32_t* par;     // union-find parent; after k_pf_root: the root
    uint32_t* cnt;     // root: vertices of its component (k_pf_root); then the component's first record slot / pair entry (k_pf_alloc)
    uint32_t* rk;      // position inside the component (k_pf_root) | size class << 29 (roots, k_pf_alloc)
    uint32_t* umi;     // its UMI
    uint32_t* rc;      // its reads
    uint32_t* tl;      // the vertex's slot inside its cell
    uint32_t* loff;    // its label's record offset (hashed label keys only: labels of one or two refs sit in the key)
    uint32_t* tcell;   // its cell (index into P2Args.cells)
    uint64_t* lh;      // its label key
};
__device__ __forceinline__ PfV pf_v(const P2Args& A) {
    const PfDev& D = *A.pfd;
    PfV v;
    v.T = D.T; v.par = A.pool + D.par; v.cnt = A.pool + D.cnt; v.rk = A.pool + D.rk; v.umi = A.pool + D.umi; v.rc = A.pool + D.rc;
    v.tl = A.pool + D.tl; v.loff = A.pool + D.loff; v.tcell = A.pool + D.tcell; v.lh = reinterpret_cast<uint64_t*>(A.pool + D.lh);
    return v;
}

__device__ __forceinline__ uint32_t nz4(uint32_t w) {   // bit k: bit 0 of byte k of w ("has an edge"; the bits above it: the vertex's reads, k_p2_part)
    w &= 0x01010101u;
    return (w & 1u) | ((w >> 7) & 2u) | ((w >> 14) & 4u) | ((w >> 21) & 8u);
}
// The flags of slots [t0, t1) of a cell (vf: the cell's first flag), sixteen to a thread as one aligned 16-byte load: f(slot, number
// inside the tile, flag byte) for every flagged slot, in slot order.  Workgroup-wide call, 256 threads; returns how many there are.
template <typename F>
__device__ __forceinline__ uint32_t pf_tile_flags(const uint8_t* vf, uint32_t t0, uint32_t t1, uint32_t* s_ws, F&& f) {
    const uintptr_t a = (uintptr_t)(vf + t0), b = (uintptr_t)(vf + t1);
    const uintptr_t c0 = a >> 4, c1 = (b - 1) >> 4;
    uint32_t carry = 0;
    for (uintptr_t cb = c0; cb <= c1; cb += 256) {   // (uniform; a tile of 4096 slots: 256 or 257 chunks)
        const uintptr_t c = cb + threadIdx.x;
        uint32_t fm = 0;
        uint4 w = make_uint4(0, 0, 0, 0);
        if (c <= c1) {
            w = *reinterpret_cast<const uint4*>(c << 4);
            fm = nz4(w.x) | (nz4(w.y) << 4) | (nz4(w.z) << 8) | (nz4(w.w) << 12);
            const uintptr_t lo = c << 4;
            if (lo < a) fm &= 0xFFFFu << (uint32_t)(a - lo);
            if (lo + 16 > b) fm &= 0xFFFFu >> (uint32_t)(lo + 16 - b);
        }
        uint32_t tot;
        uint32_t k = carry + block_excl_scan<256>((uint32_t)__popc(fm), s_ws, tot);
        const uint32_t gbase = (uint32_t)((c << 4) - (uintptr_t)vf);   // (before the cell's first flag in a tile's first chunk: the masked bits make up for it)
        for (; fm; fm &= fm - 1, ++k) {
            const uint32_t b = (uint32_t)__builtin_ctz(fm);
            const uint32_t word = b < 4 ? w.x : b < 8 ? w.y : b < 12 ? w.z : w.w;
            f(gbase + b, k, (word >> (8 * (b & 3u))) & 0xFFu);
        }
        carry += tot;
    }
    return carry;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------
// 1. dense numbers for the vertices that have an edge
__global__ __launch_bounds__(256) void k_pf_count(P2Args A) {
    if (A.st->err_code) return;
    __shared__ uint32_t s_ws[4];
    const uint2 td = A.tiles[blockIdx.x];
    const uint32_t j = td.x;
    uint32_t n = 0;
    if (!A.fb[j]) {   // (a cell k_p2_scan handed back has no vertices, and its flags were never cleared)
        const P2Cell c = A.cells[j];
        const uint32_t t0 = td.y * A.tile, t1 = min(c.R, t0 + A.tile);
        n = pf_tile_flags(A.v_flag + c.rd_base, t0, t1, s_ws, [](uint32_t, uint32_t, uint32_t) {});
    }
    if (threadIdx.x == 0) A.tq[(size_t)kQTouched * A.nta + blockIdx.x] = n;
}

// The scan of Q per-tile quantities, two levels: a workgroup per 1024 tiles scans its tiles in place and leaves its totals, one
// workgroup scans the totals (and takes what the totals size out of the pool).  Entry n_tiles is the end: its scan is the sum.
// q0 > 0 (the component counts): the tiles of a cell that is routed to the per-cell kernel count nothing.
template <int Q>
__global__ __launch_bounds__(1024) void k_pf_scan1(P2Args A, uint32_t q0) {
    if (A.st->err_code) return;
    __shared__ uint32_t s_ws[16];
    const uint32_t i = blockIdx.x * 1024 + threadIdx.x;
    const bool live = i < A.n_tiles && !(q0 > 0 && A.route[A.tiles[i].x]);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        uint32_t* a = A.tq + (size_t)(q0 + q) * A.nta;
        const uint32_t v = live ? a[i] : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan<1024>(v, s_ws, tot);
        if (i < A.nta) a[i] = ex;
        if (threadIdx.x == 0) A.bq[(size_t)(q0 + q) * A.nba + blockIdx.x] = tot;
    }
}
template <int Q>
__global__ __launch_bounds__(1024) void k_pf_scan2(P2Args A, uint32_t q0) {
    if (A.st->err_code) return;
    __shared__ uint32_t s_ws[16];
    __shared__ uint32_t s_tot[Q];
    const uint32_t nb = (A.n_tiles + 1 + 1023) / 1024;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        uint32_t* a = A.bq + (size_t)(q0 + q) * A.nba;
        uint32_t carry = 0;
        for (uint32_t b0 = 0; b0 < nb; b0 += 1024) {
            const uint32_t b = b0 + threadIdx.x;
            const uint32_t v = b < nb ? a[b] : 0u;
            uint32_t tot;
            const uint32_t ex = block_excl_scan<1024>(v, s_ws, tot);
            if (b < nb) a[b] = carry + ex;
            carry += tot;
        }
        if (threadIdx.x == 0) s_tot[q] = carry;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    PfDev& D = *A.pfd;
    if (q0 == 0) {   // T vertices have an edge: their dense arrays
        const unsigned long long T = s_tot[0], words = 10 * T + 16;
        const unsigned long long base = atomicAdd(A.pool_cur, words);
        if (base + words > A.pool_cap || T >= (1u << 31)) { set_err(A.st, T >= (1u << 31) ? kErrPugLimit : kErrPugPool, 0); D.T = 0; return; }
        unsigned long long o = (base + 3) & ~3ull;
        D.lh = o; o += 2 * T;   // (8-byte entries first: o is a multiple of four words)
        D.par = o; o += T; D.cnt = o; o += T; D.rk = o; o += T; D.umi = o; o += T; D.rc = o; o += T; D.tl = o; o += T; D.loff = o; o += T; D.tcell = o;
        D.T = (uint32_t)T;
    } else if (Q >= 5) {   // the lists: pairs, first record slot per listed component (+ the end), records, the covers' set-aside lists
        const unsigned long long NP = s_tot[0], NC = (unsigned long long)s_tot[1] + s_tot[Q >= 5 ? 2 : 0], S = (unsigned long long)s_tot[Q >= 5 ? 3 : 0] + s_tot[Q >= 5 ? 4 : 0];
        const unsigned long long words = 2 * NP + 4 + (NC + 4) + 8 * S + 8 + 4 * (NC + A.n_cells) + 8 + NC / 2 + 4 + S + 4;
        const unsigned long long base = atomicAdd(A.pool_cur, words);
        if (base + words > A.pool_cap || S >= (1ull << 32)) { set_err(A.st, kErrPugPool, 0); return; }
        unsigned long long o = (base + 3) & ~3ull;
        D.mrec = o; o += 8 * S + 4;            // (16-byte aligned: uint4 records)
        D.prv = o; o += 2 * NP + 2;
        D.midoff = o; o += NC + 2;
        o = (o + 3) & ~3ull;
        D.tied = o; o += 4 * (NC + A.n_cells) + 4;
        D.slow = o; o += NC / 2 + 2;   // (16-bit entries, one per listed component)
        D.cmv = o;                    // per record slot: its class's smallest record offset (k_p2_tied, for the vertices of set-aside components)
        D.NP = (uint32_t)NP; D.NC = (uint32_t)NC; D.S = (uint32_t)S;
        A.pool[D.midoff + NC] = (uint32_t)S;   // the list's last offset
    }
}

__global__ __launch_bounds__(256) void k_pf_number(P2Args A) {
    if (A.st->err_code) return;
    __shared__ uint32_t s_ws[4];
    __shared__ uint16_t s_g[4096 + 16];
    __shared__ uint8_t s_rc[4096 + 16];
    const uint2 td = A.tiles[blockIdx.x];
    const uint32_t j = td.x;
    if (A.fb[j]) return;
    const PfV V = pf_v(A);
    const P2Cell c = A.cells[j];
    const uint32_t t0 = td.y * A.tile, t1 = min(c.R, t0 + A.tile);
    uint32_t* lidx = A.lidx + c.rd_base;
    const uint64_t* ch = A.s_h + c.rd_base;
    const uint64_t* cu = A.s_u + c.rd_base;
    const uint32_t* coff = A.v_off + c.rd_base;
    const uint32_t tb = pf_g(A, kQTouched, blockIdx.x);
    const uint32_t nt = pf_tile_flags(A.v_flag + c.rd_base, t0, t1, s_ws, [&](uint32_t g, uint32_t k, uint32_t fb) { s_g[k] = (uint16_t)(g - t0); s_rc[k] = (uint8_t)(fb >> 1); });
    __syncthreads();
    // The vertex's label key (and, for a hashed key, where its label lies) into the dense arrays as well: the covers and the record
    // builder ask for the labels of these vertices - one in four slots - and every such gather out of the per-slot arrays fetched
    // a 64-byte line for 8 or 4 bytes; here the lines are read once, by neighbouring lanes, and everybody behind reads dense arrays.
    // (Four vertices to a thread and trip: the gathers of a trip go out together.)
    // (the vertex's UMI lies in lidx - k_p2_part left it there for the search - on the very line its dense number is about to be
    //  written to; its reads came with its flag: 127 stands for "127 or more", which s_u knows)
    for (uint32_t k0 = threadIdx.x; k0 < nt; k0 += 1024) {
        uint32_t g[4], off[4], um[4], rc[4];
        uint64_t h[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { const uint32_t k = k0 + 256u * (uint32_t)r; g[r] = k < nt ? t0 + s_g[k] : t0; rc[r] = k < nt ? s_rc[k] : 0u; h[r] = k < nt ? ch[g[r]] : 0ull; um[r] = k < nt ? lidx[g[r]] : 0u; }
#pragma unroll
        for (int r = 0; r < 4; ++r) { off[r] = (uint32_t)(h[r] >> 62) == 3 ? coff[g[r]] : 0u; if (rc[r] == 127) rc[r] = (uint32_t)cu[g[r]] & kVCntMask; }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t k = k0 + 256u * (uint32_t)r, t = tb + k;
            if (k >= nt) continue;
            lidx[g[r]] = t;
            V.par[t] = t; V.cnt[t] = 0; V.tl[t] = g[r]; V.tcell[t] = j; V.lh[t] = h[r]; V.loff[t] = off[r]; V.umi[t] = um[r]; V.rc[t] = rc[r];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The pairs the search left, a workgroup over 256 consecutive partitions at a time: every thread brings one partition's count and
// place, a scan lays the partitions' pairs end to end, and the threads take the pairs of the 256 partitions evenly - a partition
// holds anything from none to hundreds - FOUR to a thread and trip, so that a trip's chains of dependent loads run side by side.
// f(where the four pairs lie (nullptr: none), first read slot of each one's cell).
template <typename F>
__device__ __forceinline__ void pf_for_each_pair4(const P2Args& A, uint32_t* s_start, unsigned long long* s_src, uint32_t* s_rdb, uint32_t* s_ws, F&& f) {
    for (uint32_t p0 = blockIdx.x * 256; p0 < A.n_parts; p0 += gridDim.x * 256) {
        const uint32_t gp = p0 + threadIdx.x;
        uint32_t np = 0, rdb = 0;
        unsigned long long src = 0;
        if (gp < A.n_parts) {
            np = A.pnp[gp];
            if (np) {
                const P2Cell& c = A.cells[A.pcell[gp]];
                const unsigned long long so = c.rd_base + A.poff[gp];
                rdb = (uint32_t)c.rd_base;
                // (more pairs than the partition has slots of its own: k_p2_search_over put the list into the pool and left its place in the first own slot)
                src = np > A.pcnt[gp] ? (unsigned long long)(uintptr_t)(A.pool + A.pairs[so]) : (unsigned long long)(uintptr_t)(A.pairs + so);
            }
        }
        uint32_t tot;
        const uint32_t ex = block_excl_scan<256>(np, s_ws, tot);
        s_start[threadIdx.x] = ex; s_src[threadIdx.x] = src; s_rdb[threadIdx.x] = rdb;
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < tot; t += 1024) {
            uint64_t* sp[4];
            uint32_t rb[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t tt = t + 256u * (uint32_t)r;
                sp[r] = nullptr; rb[r] = 0;
                if (tt < tot) {
                    uint32_t lo = 0, hi = 256;   // the last partition that starts at or before tt: the one tt falls into (empty ones share their start with the next)
                    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (s_start[mid] <= tt) lo = mid; else hi = mid; }
                    sp[r] = reinterpret_cast<uint64_t*>((uintptr_t)s_src[lo]) + (tt - s_start[lo]);
                    rb[r] = s_rdb[lo];
                }
            }
            f(sp, rb);
        }
        __syncthreads();
    }
}

// 0. (behind the search, before anything reads a flag) one thread per CANDIDATE pair - k_p2_search writes down every two vertices
//    whose UMIs are as has_edge wants them and whose label signatures share a bit; whether the labels share a REF (pugutils.rs:187-204)
//    takes the label keys, for hashed keys the record offsets and the ref lists in the chunk: dependent gathers, which a wave that
//    owns a partition waits for one after the other and a thread per candidate over the whole range does not notice.  A candidate
//    that fails is cleared (0: no direction bit - every loop over pairs skips it); the end points of the others get bit 0 of their
//    flag bytes.
#ifdef AFQ_CHECK_COUNT
__device__ unsigned long long g_check_n[2];
__global__ void k_check_count_dump() { printf("k_p2_check: %llu candidates, %llu cleared\n", g_check_n[0], g_check_n[1]); g_check_n[0] = g_check_n[1] = 0; }
#endif
#ifndef AFQ_CHECK_PER
#define AFQ_CHECK_PER 2
#endif
__global__ __launch_bounds__(256) void k_p2_check(P2Args A) {
    if (A.st->err_code) return;
    __shared__ uint32_t s_start[256];
    __shared__ unsigned long long s_src[256];
    __shared__ uint32_t s_cellj[256];
    __shared__ uint32_t s_ws[4];
    for (uint32_t p0 = blockIdx.x * 256; p0 < A.n_parts; p0 += gridDim.x * 256) {
        const uint32_t gp = p0 + threadIdx.x;
        uint32_t np = 0, cj = 0;
        unsigned long long src = 0;
        if (gp < A.n_parts) {
            np = A.pnp[gp];
            if (np) {
                cj = A.pcell[gp];
                const unsigned long long so = A.cells[cj].rd_base + A.poff[gp];
                src = np > A.pcnt[gp] ? (unsigned long long)(uintptr_t)(A.pool + A.pairs[so]) : (unsigned long long)(uintptr_t)(A.pairs + so);
            }
        }
        uint32_t tot;
        const uint32_t ex = block_excl_scan<256>(np, s_ws, tot);
        s_start[threadIdx.x] = ex; s_src[threadIdx.x] = src; s_cellj[threadIdx.x] = cj;
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < tot; t += 256 * AFQ_CHECK_PER) {   // (AFQ_CHECK_PER candidates per thread and trip: their gathers side by side)
            uint64_t* sp[AFQ_CHECK_PER];
            uint64_t e[AFQ_CHECK_PER], hx[AFQ_CHECK_PER], hy[AFQ_CHECK_PER];
            uint32_t ox[AFQ_CHECK_PER], oy[AFQ_CHECK_PER], gx[AFQ_CHECK_PER], gy[AFQ_CHECK_PER], cj2[AFQ_CHECK_PER];
            unsigned long long rb[AFQ_CHECK_PER];
#pragma unroll
            for (int r = 0; r < AFQ_CHECK_PER; ++r) {
                const uint32_t tt = t + 256u * (uint32_t)r;
                sp[r] = nullptr; cj2[r] = 0;
                if (tt < tot) {
                    uint32_t lo = 0, hi = 256;   // the last partition that starts at or before tt
                    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (s_start[mid] <= tt) lo = mid; else hi = mid; }
                    sp[r] = reinterpret_cast<uint64_t*>((uintptr_t)s_src[lo]) + (tt - s_start[lo]);
                    cj2[r] = s_cellj[lo];
                }
            }
#pragma unroll
            for (int r = 0; r < AFQ_CHECK_PER; ++r) { e[r] = sp[r] ? *sp[r] : 0ull; rb[r] = A.cells[cj2[r]].rd_base; }
#pragma unroll
            for (int r = 0; r < AFQ_CHECK_PER; ++r) {
                gx[r] = (uint32_t)(e[r] >> 31) & 0x7FFFFFFFu; gy[r] = (uint32_t)e[r] & 0x7FFFFFFFu;
                const bool on = sp[r] != nullptr;
                hx[r] = on ? A.s_h[rb[r] + gx[r]] : 0ull; hy[r] = on ? A.s_h[rb[r] + gy[r]] : 0ull;
            }
#pragma unroll
            for (int r = 0; r < AFQ_CHECK_PER; ++r) {   // (the record offsets of hashed labels only: one or two refs sit in the key)
                ox[r] = sp[r] && (uint32_t)(hx[r] >> 62) == 3 ? A.v_off[rb[r] + gx[r]] : 0u;
                oy[r] = sp[r] && (uint32_t)(hy[r] >> 62) == 3 ? A.v_off[rb[r] + gy[r]] : 0u;
            }
#pragma unroll
            for (int r = 0; r < AFQ_CHECK_PER; ++r) {
                if (!sp[r]) continue;
                bool ok = true;
                if (hx[r] != hy[r] || (uint32_t)(hx[r] >> 62) == 3) {   // (equal hashed keys are equal labels only once somebody has compared them: here)
                    const uint32_t* W = reinterpret_cast<const uint32_t*>(A.bytes + A.cells[cj2[r]].chunk_off);
                    ok = klab_overlap(klab(W, A.hw, hx[r], ox[r]), klab(W, A.hw, hy[r], oy[r]));
                }
#ifdef AFQ_CHECK_COUNT
                atomicAdd(&g_check_n[0], 1ull); if (!ok) atomicAdd(&g_check_n[1], 1ull);
#endif
                if (!ok) { *sp[r] = 0ull; continue; }
                const unsigned long long ax = rb[r] + gx[r], ay = rb[r] + gy[r];
                // (bit 0 of both end points' flag bytes.  A byte read and, if the bit is not there yet, a byte written: threads that race
                //  on a byte write the same value - the bits above bit 0 are k_p2_part's and do not change - and a byte store leaves
                //  its neighbours alone.  As atomic ORs on the bytes' words: 1.07 ms per launch against 0.86; no flags at all: 0.87.)
                const uint8_t fx = A.v_flag[ax], fy = A.v_flag[ay];
                if (!(fx & 1u)) A.v_flag[ax] = fx | 1u;
                if (!(fy & 1u)) A.v_flag[ay] = fy | 1u;
            }
        }
        __syncthreads();
    }
}
void launch_p2_check(hipStream_t s, const P2Args& a) {
    if (a.n_parts) AFQ_LAUNCH(k_p2_check, std::min((a.n_parts + 255) / 256, 4096u), 256, s, a);
#ifdef AFQ_CHECK_COUNT
    hipLaunchKernelGGL(k_check_count_dump, dim3(1), dim3(1), 0, s);
#endif
}

// 2. components: one thread per pair.  A root is only ever hooked under a SMALLER vertex (no cycle); find() halves the path it
//    walks with an atomic min (a racing shortcut still points at an ancestor).
__global__ __launch_bounds__(256) void k_pf_union(P2Args A) {
    if (A.st->err_code) return;
    __shared__ uint32_t s_start[256];
    __shared__ unsigned long long s_src[256];
    __shared__ uint32_t s_rdb[256];
    __shared__ uint32_t s_ws[4];
    const PfV V = pf_v(A);
    uint32_t* par = V.par;
    auto find_from = [&](uint32_t i, uint32_t p) -> uint32_t {   // p = par[i], read already
        while (p != i) {
            const uint32_t gp2 = ag_ld(&par[p]);
            if (gp2 != p) __hip_atomic_fetch_min(&par[i], gp2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            i = p; p = gp2;
        }
        return i;
    };
    pf_for_each_pair4(A, s_start, s_src, s_rdb, s_ws, [&](uint64_t* (&sp)[4], uint32_t (&rb)[4]) {
        uint64_t e[4];
        uint32_t tx[4], ty[4], px[4], py[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { e[r] = sp[r] ? *sp[r] : 0ull; if (!e[r]) sp[r] = nullptr; }   // (0: a candidate k_p2_check cleared)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            tx[r] = sp[r] ? A.lidx[(size_t)rb[r] + ((uint32_t)(e[r] >> 31) & 0x7FFFFFFFu)] : 0u;
            ty[r] = sp[r] ? A.lidx[(size_t)rb[r] + ((uint32_t)e[r] & 0x7FFFFFFFu)] : 0u;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (sp[r] && (tx[r] >= V.T || ty[r] >= V.T)) { set_err(A.st, kErrInternal, 0); sp[r] = nullptr; }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { px[r] = sp[r] ? ag_ld(&par[tx[r]]) : 0u; py[r] = sp[r] ? ag_ld(&par[ty[r]]) : 0u; }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (!sp[r]) continue;
            uint32_t a = find_from(tx[r], px[r]), b = find_from(ty[r], py[r]);
            while (a != b) {
                if (a < b) { const uint32_t t = a; a = b; b = t; }
                uint32_t expected = a;
                if (__hip_atomic_compare_exchange_strong(&par[a], &expected, b, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                a = find_from(a, expected); b = find_from(b, ag_ld(&par[b]));
            }
        }
    });
}

// 3. one thread per vertex (four to a thread and trip): its root (written over its parent: later kernels read it with one load),
//    its position inside its component
__global__ __launch_bounds__(256) void k_pf_root(P2Args A) {
    if (A.st->err_code) return;
    const PfV V = pf_v(A);
    for (uint32_t t0 = blockIdx.x * 1024 + threadIdx.x; t0 < V.T; t0 += gridDim.x * 1024) {
        uint32_t r[4], p[4];
        bool on[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { const uint32_t t = t0 + 256u * (uint32_t)i; on[i] = t < V.T; r[i] = t; p[i] = on[i] ? V.par[t] : t; }
        for (bool go = true; go;) {
            go = false;
#pragma unroll
            for (int i = 0; i < 4; ++i) if (p[i] != r[i]) { r[i] = p[i]; p[i] = V.par[r[i]]; go = true; }
        }
        // (a root is vertex 0 of its component and counts nothing: its counter ends at the component's size less one - a component in
        //  eight has three vertices or more, so more than two atomics in five are roots' and never made)
        uint32_t k[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) k[i] = on[i] && r[i] != t0 + 256u * (uint32_t)i ? atomicAdd(&V.cnt[r[i]], 1u) + 1u : 0u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t t = t0 + 256u * (uint32_t)i;
            if (!on[i]) continue;
            V.rk[t] = k[i];
            if (r[i] != t) V.par[t] = r[i];   // (a walk through t meanwhile meets the old parent or the root: both are ancestors)
        }
    }
}

// 4. a workgroup per tile: the size classes of the components its roots head -> the tile's five counts.  A cell that holds a
//    component for the per-cell kernel is flagged (its tiles then count nothing in the scan).
__global__ __launch_bounds__(256) void k_pf_cats(P2Args A) {
    if (A.st->err_code) return;
    __shared__ uint32_t s_c[5];
    const uint32_t i = blockIdx.x, j = A.tiles[i].x, lane = lane_id();
    if (threadIdx.x < 5) s_c[threadIdx.x] = 0;
    __syncthreads();
    if (!A.fb[j]) {
        const PfV V = pf_v(A);
        const uint32_t tb = pf_g(A, kQTouched, i), te = pf_g(A, kQTouched, i + 1);
        bool big = false;
        for (uint32_t t0 = tb; t0 < te; t0 += 256) {   // (uniform)
            const uint32_t t = t0 + threadIdx.x;
            const bool root = t < te && V.par[t] == t;
            const uint32_t n = root ? V.cnt[t] + 1u : 0u;   // (the root itself is not counted: k_pf_root)
            const uint32_t cat = root ? cat_of(n, A.large_thresh) : 0u;
            big = big || (root && cat == 0);
            const uint32_t npr = (uint32_t)__popcll(__ballot(cat == kFCatPair)), nti = (uint32_t)__popcll(__ballot(cat == kFCatTiny)), nmi = (uint32_t)__popcll(__ballot(cat == kFCatMid));
            uint32_t sti = cat == kFCatTiny ? n : 0u, smi = cat == kFCatMid ? n : 0u;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) { sti += __shfl_xor(sti, d); smi += __shfl_xor(smi, d); }
            if (lane == 0) {
                if (npr) atomicAdd(&s_c[0], npr);
                if (nti) { atomicAdd(&s_c[1], nti); atomicAdd(&s_c[3], sti); }
                if (nmi) { atomicAdd(&s_c[2], nmi); atomicAdd(&s_c[4], smi); }
            }
        }
        if (__any(big) && lane == 0) A.route[j] = 1;   // (the same word from every tile that finds one: any of them will do)
    }
    __syncthreads();
    if (threadIdx.x < 5) A.tq[(size_t)(kQPr + threadIdx.x) * A.nta + i] = s_c[threadIdx.x];
}

// 5. The lone vertices' two-gene classes (em only): k_p2_lone staged them at their partitions' slots, a count per partition.  The scan
//    of the counts over the range's partitions (two levels, as for the tiles; a cell's partitions are consecutive) says where a
//    partition's classes go in its cell's label area; a thread per partition moves them.
__device__ __forceinline__ uint32_t pf_pc(const P2Args& A, uint32_t gp) { return A.pcpre[gp] + A.pbq[gp >> 10]; }   // staged classes in front of partition gp (gp = n_parts: all)
__global__ __launch_bounds__(1024) void k_pf_pscan1(P2Args A) {
    if (A.st->err_code) return;
    __shared__ uint32_t s_ws[16];
    const uint32_t gp = blockIdx.x * 1024 + threadIdx.x;
    const uint32_t v = gp < A.n_parts ? A.pncls[gp] : 0u;
    uint32_t tot;
    const uint32_t ex = block_excl_scan<1024>(v, s_ws, tot);
    A.pcpre[gp] = ex;
    if (threadIdx.x == 0) A.pbq[blockIdx.x] = tot;
}
__global__ __launch_bounds__(1024) void k_pf_pscan2(P2Args A) {
    if (A.st->err_code) return;
    __shared__ uint32_t s_ws[16];
    const uint32_t nb = (A.n_parts + 1 + 1023) / 1024;
    uint32_t carry = 0;
    for (uint32_t b0 = 0; b0 < nb; b0 += 1024) {
        const uint32_t b = b0 + threadIdx.x;
        const uint32_t v = b < nb ? A.pbq[b] : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan<1024>(v, s_ws, tot);
        if (b < nb) A.pbq[b] = carry + ex;
        carry += tot;
    }
}
__global__ __launch_bounds__(256) void k_pf_move(P2Args A) {
    if (A.st->err_code) return;
    // (256 partitions to a workgroup, their classes laid end to end by a scan and taken by the threads evenly - a partition stages
    //  anything from none to dozens; a thread per partition walking its own classes took 0.34 ms per launch)
    __shared__ uint32_t s_start[256], s_nref[256], s_d[256], s_w[256], s_ws[4];
    __shared__ unsigned long long s_src[256], s_lab[256];
    for (uint32_t p0 = blockIdx.x * 256; p0 < A.n_parts; p0 += gridDim.x * 256) {
        const uint32_t gp = p0 + threadIdx.x;
        uint32_t nk = 0, nref = 0, d0 = 0, w0 = 0;
        unsigned long long src = 0, lab = 0;
        if (gp < A.n_parts) {
            nk = A.pncls[gp];
            const uint32_t j = nk ? A.pcell[gp] : 0u;
            if (nk && (A.fb[j] || A.route[j])) nk = 0;   // (the per-cell kernel moves the classes of the cells routed to it itself)
            if (nk) {
                const P2Cell& c = A.cells[j];
                const uint32_t* gc = A.gcnt + 4 * (size_t)j;
                const uint32_t at = pf_pc(A, gp) - pf_pc(A, c.part_base);
                d0 = gc[2] + at; w0 = gc[1] + 2 * at; nref = c.n_ref;
                src = (unsigned long long)(uintptr_t)(A.cstage + c.rd_base + A.poff[gp]);
                lab = (unsigned long long)(uintptr_t)(A.lab + 2 * c.key_off);
            }
        }
        uint32_t tot;
        const uint32_t ex = block_excl_scan<256>(nk, s_ws, tot);
        s_start[threadIdx.x] = ex; s_src[threadIdx.x] = src; s_lab[threadIdx.x] = lab; s_nref[threadIdx.x] = nref; s_d[threadIdx.x] = d0; s_w[threadIdx.x] = w0;
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < tot; t += 256) {
            uint32_t lo = 0, hi = 256;   // the last partition that starts at or before t
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (s_start[mid] <= t) lo = mid; else hi = mid; }
            const uint32_t k = t - s_start[lo];
            const uint64_t v = reinterpret_cast<const uint64_t*>((uintptr_t)s_src[lo])[k];
            uint32_t* labw = reinterpret_cast<uint32_t*>((uintptr_t)s_lab[lo]);
            const uint32_t lab_cap = s_nref[lo] + 1, w = s_w[lo] + 2 * k, d = s_d[lo] + k;
            if (w + 2 > lab_cap || 2 * (d + 1) > lab_cap) continue;   // (k_pf_cells reports it)
            uint32_t* labd = labw + lab_cap;
            labw[w] = (uint32_t)v; labw[w + 1] = (uint32_t)(v >> 32);
            labd[2 * d] = w; labd[2 * d + 1] = 2;
        }
        __syncthreads();
    }
}

// ... and a thread per cell: the descriptor the cover kernels read (the same words the per-cell graph kernel leaves) and the cell's
//    counters as the covers start from them.  A cell k_p2_scan handed back goes onto the one-workgroup kernel's list, a cell with a
//    component of more than 64 vertices onto the per-cell graph kernel's.
__global__ __launch_bounds__(256) void k_pf_cells(P2Args A) {
    if (A.st->err_code) return;
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= A.n_cells) return;
    const P2Cell c = A.cells[j];
    PfDev& D = *A.pfd;
    if (A.fb[j]) { A.fb_list[atomicAdd(A.fb_count, 1u)] = c.cell; return; }   // (a partition over the capacity)
    if (A.route[j]) { A.old_list[atomicAdd(&D.n_old, 1u)] = j; return; }       // (it moves its classes and writes its descriptor itself)
    uint32_t* gc = A.gcnt + 4 * (size_t)j;
    uint32_t w0 = gc[1], d0 = gc[2];
    if (A.em && A.lab) {
        const uint32_t tot = pf_pc(A, c.part_base + (1u << c.lgP)) - pf_pc(A, c.part_base), lab_cap = c.n_ref + 1;
        if (w0 + 2 * tot > lab_cap || 2 * (d0 + tot) > lab_cap) { set_err(A.st, kErrPugLimit, c.cell); return; }
        w0 += 2 * tot; d0 += tot;
    }
    gc[0] = c.R; gc[1] = w0; gc[2] = d0;   // (entries 0..R of the column list belong to the lone-vertex kernel)
    const uint32_t i0 = c.tile0, i1 = c.tile0 + (c.R + A.tile - 1) / A.tile;
    const uint32_t pr0 = pf_g(A, kQPr, i0), ti0 = pf_g(A, kQTiny, i0), mi0 = pf_g(A, kQMid, i0);
    const uint32_t n_pr = pf_g(A, kQPr, i1) - pr0, n_tiny = pf_g(A, kQTiny, i1) - ti0, n_mid = pf_g(A, kQMid, i1) - mi0;
    const uint32_t comp_base = ti0 + mi0;
    uint32_t* d = A.gdesc + kGDescWords * (size_t)j;
    auto put = [&](int at, unsigned long long o) { d[at] = (uint32_t)o; d[at + 1] = (uint32_t)(o >> 32); };
    const unsigned long long tied = D.tied + 4ull * (j + comp_base);
    d[1] = n_pr; d[2] = n_tiny; d[3] = n_tiny + n_mid; d[15] = 0;
    d[4] = 0xFFFFFFFFu; d[5] = 0xFFFFFFFFu;   // (no table from touched-vertex numbers to slots)
    put(6, D.prv + 2ull * pr0); put(8, D.midoff + comp_base); put(10, D.mrec); put(16, tied);
    d[12] = c.R; d[13] = w0; d[14] = d0;
    uint32_t* tp = A.pool + tied;
    tp[0] = 0; tp[1] = 0; tp[2] = 0; tp[3] = 0;
    d[0] = 7u;   // the lists are there (1), in slot order (2: kCoverDefer), records with (UMI, reads) in place of adjacency masks (4)
}

// ... and a thread per tile: where the tile's slices of the lists lie, gathered once (the workgroups that take a tile - k_pf_alloc, the
// covers - used to start with a dozen dependent loads each to work this out: with 33 000 workgroups per range and four to eight to
// a CU that start-up was most of their time)
__global__ __launch_bounds__(256) void k_pf_tiles(P2Args A) {
    if (A.st->err_code) return;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= A.n_tiles) return;
    const uint32_t j = A.tiles[i].x;
    const P2Cell& c = A.cells[j];
    PfTile t{};
    t.j = j; t.live = !(A.fb[j] || A.route[j]);
    t.tb = pf_g(A, kQTouched, i); t.te = pf_g(A, kQTouched, i + 1);
    const uint32_t i0 = c.tile0, i1 = c.tile0 + (c.R + A.tile - 1) / A.tile;
    const uint32_t ti0 = pf_g(A, kQTiny, i0), mi0 = pf_g(A, kQMid, i0), sti0 = pf_g(A, kQSTiny, i0), smi0 = pf_g(A, kQSMid, i0);
    const uint32_t S_tiny = pf_g(A, kQSTiny, i1) - sti0, slot_base = sti0 + smi0;
    t.n_tiny = pf_g(A, kQTiny, i1) - ti0; t.comp_base = ti0 + mi0;
    t.p0 = pf_g(A, kQPr, i); t.n_pr = pf_g(A, kQPr, i + 1) - t.p0;
    t.a0 = pf_g(A, kQTiny, i) - ti0; t.na = pf_g(A, kQTiny, i + 1) - ti0 - t.a0;
    t.b0 = t.n_tiny + (pf_g(A, kQMid, i) - mi0); t.nb = t.n_tiny + (pf_g(A, kQMid, i + 1) - mi0) - t.b0;
    t.s_ti = slot_base + (pf_g(A, kQSTiny, i) - sti0); t.s_mi = slot_base + S_tiny + (pf_g(A, kQSMid, i) - smi0);
    A.ptile[i] = t;
}

// 6. a workgroup per tile: its roots take their entries of the cell's run of the lists and their record slots - where the scans of
//    the tile counts say, in slot order inside the tile (two packed scans per 256 roots: counts of the three classes, record slots)
__global__ __launch_bounds__(256) void k_pf_alloc(P2Args A) {
    if (A.st->err_code) return;
    __shared__ uint32_t s_ws[4];
    const PfTile T = A.ptile[blockIdx.x];
    if (!T.live) return;   // (the per-cell kernel's, or handed back: no entry, no record - k_pf_place passes over roots without a size class)
    const PfV V = pf_v(A);
    const uint32_t tb = T.tb, te = T.te;
    uint32_t c_pr = T.p0;                                                  // next pair entry (range-wide)
    uint32_t c_ti = T.comp_base + T.a0, c_mi = T.comp_base + T.b0;         // next list entries
    uint32_t s_ti = T.s_ti, s_mi = T.s_mi;                                 // next record slots
    uint32_t* mid_off = A.pool + A.pfd->midoff;
    for (uint32_t t0 = tb; t0 < te; t0 += 256) {   // (uniform)
        const uint32_t t = t0 + threadIdx.x;
        const bool root = t < te && V.par[t] == t;
        const uint32_t n = root ? V.cnt[t] + 1u : 0u;
        const uint32_t cat = root ? cat_of(n, A.large_thresh) : 0u;
        // (up to 256 roots per trip: three 10-bit counts in one word; record slots: <= 2048 and <= 16384)
        const uint32_t vc = (cat == kFCatPair ? 1u : 0u) | (cat == kFCatTiny ? 1u << 10 : 0u) | (cat == kFCatMid ? 1u << 20 : 0u);
        const uint32_t vs = (cat == kFCatTiny ? n : 0u) | (cat == kFCatMid ? n << 12 : 0u);
        uint32_t totc, tots;
        const uint32_t exc = block_excl_scan<256>(vc, s_ws, totc);
        const uint32_t exs = block_excl_scan<256>(vs, s_ws, tots);
        if (root) {
            uint32_t where = 0;
            if (cat == kFCatPair) where = 2 * (c_pr + (exc & 0x3FFu));
            else if (cat == kFCatTiny) { where = s_ti + (exs & 0xFFFu); mid_off[c_ti + ((exc >> 10) & 0x3FFu)] = where; }
            else if (cat == kFCatMid) { where = s_mi + (exs >> 12); mid_off[c_mi + (exc >> 20)] = where; }
            V.cnt[t] = where;
            V.rk[t] = (V.rk[t] & kRankMask) | (cat << kCatShift);
        }
        c_pr += totc & 0x3FFu; c_ti += (totc >> 10) & 0x3FFu; c_mi += totc >> 20;
        s_ti += tots & 0xFFFu; s_mi += tots >> 12;
    }
}

// 7. one thread per vertex (two to a thread and trip): into the pair list, or its 32-byte cover record (vertex slot, label length, up to
//    four refs - a longer label: where it lies in the chunk -, UMI, reads).  Everything it reads is dense.
__global__ __launch_bounds__(256) void k_pf_place(P2Args A) {
    if (A.st->err_code) return;
    const PfV V = pf_v(A);
    const PfDev& D = *A.pfd;
    uint32_t* pr_v = A.pool + D.prv;
    uint4* mrec = reinterpret_cast<uint4*>(A.pool + D.mrec);
    for (uint32_t t0 = blockIdx.x * 512 + threadIdx.x; t0 < V.T; t0 += gridDim.x * 512) {
        uint32_t r[2], where[2], cat[2], kk[2], off[2], g[2];
        uint64_t h[2];
        bool on[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) { const uint32_t t = t0 + 256u * (uint32_t)q; on[q] = t < V.T; r[q] = on[q] ? V.par[t] : 0u; kk[q] = on[q] ? V.rk[t] & kRankMask : 0u; h[q] = on[q] ? V.lh[t] : 0ull; off[q] = on[q] ? V.loff[t] : 0u; g[q] = on[q] ? V.tl[t] : 0u; }
#pragma unroll
        for (int q = 0; q < 2; ++q) { where[q] = on[q] ? V.cnt[r[q]] : 0u; cat[q] = on[q] ? V.rk[r[q]] >> kCatShift : 0u; }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (!on[q]) continue;
            const uint32_t t = t0 + 256u * (uint32_t)q;
            if (cat[q] != kFCatTiny && cat[q] != kFCatMid) {
                if (cat[q] == kFCatPair) pr_v[where[q] + kk[q]] = t;   // (the dense number: the two-vertex rule reads the label out of the dense arrays)
                continue;
            }
            uint32_t n = (uint32_t)(h[q] >> 62), r0 = 0xFFFFFFFFu, r1 = 0xFFFFFFFFu, r2 = 0xFFFFFFFFu, r3 = 0xFFFFFFFFu;
            if (n == 1) r0 = (uint32_t)h[q] & 0x7FFFFFFFu;
            else if (n == 2) { r0 = (uint32_t)(h[q] >> 31) & 0x7FFFFFFFu; r1 = (uint32_t)h[q] & 0x7FFFFFFFu; }
            else if (n == 3) {   // a hashed key: the label lies in its record
                const P2Cell& c = A.cells[V.tcell[t]];
                const uint32_t* W = reinterpret_cast<const uint32_t*>(A.bytes + c.chunk_off);
                n = W[off[q]];
                const uint32_t* lp = W + off[q] + A.hw;
                if (n <= 4) { r0 = lp[0] & 0x7FFFFFFFu; r1 = lp[1] & 0x7FFFFFFFu; r2 = lp[2] & 0x7FFFFFFFu; if (n > 3) r3 = lp[3] & 0x7FFFFFFFu; }
                else { const uint64_t pa = (uint64_t)(uintptr_t)lp; r0 = (uint32_t)pa; r1 = (uint32_t)(pa >> 32); }
            }
            const size_t at = (size_t)where[q] + kk[q];
            mrec[2 * at] = make_uint4(g[q], n, r0, r1);
            mrec[2 * at + 1] = make_uint4(r2, r3, V.umi[t], V.rc[t]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// 9. the covers, a workgroup per TILE: the two-vertex rule and the arborescence covers (afq_pug_common.h, unchanged in rule) over the
//    components the tile's roots head - its slice of the cell's pair list, of the 3..8 list (eight components to a wave) and of the
//    9..64 list (a wave each).  Rounds 4-5 gave a cell's components to ONE workgroup (k_p2_cover: a 36 000-read cell's 3 800
//    components one batch after the other, 1024 threads and a CU to itself); a cell's ~9 tiles now run side by side, eight workgroups
//    to a CU.  Columns and classes go to the cell's lists through its counters in global memory (one reservation per wave and
//    batch); a component whose round meets a tie is set aside on the cell's list for k_p2_tied, as before.
// (three kernels, not one: together they were 104 VGPRs and four waves per SIMD - the two-vertex rule alone runs at eight)
__global__ __launch_bounds__(256) void k_pc_pairs(P2Args A) {
    if (A.st->err_code) return;
    __shared__ uint32_t s_stage[4][64 * kStageRefs + kMaxGenesPerLabel];   // (per wave: 64 label rows and ONE gene row behind them: the lanes that need it take turns)
    const PfTile T = A.ptile[blockIdx.x];
    if (!T.live || !T.n_pr) return;
    const uint32_t j = T.j, p0 = T.p0, n_pr = T.n_pr;
    const P2Cell c = A.cells[j];
    const PfDev& D = *A.pfd;
    PugCtx C = make_ctx(A, c, A.gcnt + 4 * (size_t)j);
    C.adj_umi = 1;
    // (the list holds dense numbers, the labels' keys lie in the dense arrays)
    p2_cover_pairs<256, true>(C, reinterpret_cast<const uint64_t*>(A.pool + D.lh), A.pool + D.loff, nullptr, A.pool + D.prv + 2ull * p0, n_pr, s_stage[threadIdx.x >> 6]);
}
// components of 3..4 vertices under short labels: a LANE each (cover_lane4), a WAVE per tile (a tile holds ~115 components of 3..8
// vertices: as a 256-thread workgroup per tile two of its four waves had nothing to do, and at four workgroups to a CU the kernel
// was the latency of one tile's chain of loads, over and over).  What is left - five vertices or more, a label of more than four
// refs - goes onto the tile's slice of a list for k_pc_tiny8.
__global__ __launch_bounds__(256) void k_pc_lane4(P2Args A) {
    if (A.st->err_code) return;
    const uint32_t lane = threadIdx.x & 63u, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= A.n_tiles) return;
    const PfTile T = A.ptile[i];
    if (!T.live || !T.na) return;
    const uint32_t j = T.j;
    const P2Cell c = A.cells[j];
    const PfDev& D = *A.pfd;
    PugCtx C = make_ctx(A, c, A.gcnt + 4 * (size_t)j);
    C.adj_umi = 1;
    const uint4* mrec = reinterpret_cast<const uint4*>(A.pool + D.mrec);
    const uint32_t* mid_off = A.pool + D.midoff + T.comp_base;
    uint32_t* const tied = A.pool + D.tied + 4ull * (j + T.comp_base);
    uint16_t* slow = reinterpret_cast<uint16_t*>(A.pool + D.slow) + T.comp_base + T.a0;   // (one entry per component of the list: the tile's slice)
    uint32_t nslow = 0;
    for (uint32_t c0 = 0; c0 < T.na; c0 += 64) {   // (uniform)
        const uint32_t ci = c0 + lane;
        uint32_t b0c = 0, n = 0;
        if (ci < T.na) { b0c = mid_off[T.a0 + ci]; n = mid_off[T.a0 + ci + 1] - b0c; }
        bool fast = n != 0 && n <= 4;
        if (fast) {
#pragma unroll
            for (int v = 0; v < 4; ++v) if ((uint32_t)v < n) fast = fast && mrec[2 * (size_t)(b0c + v)].y <= 4;
        }
        const uint64_t sm = __ballot(n != 0 && !fast);
        if (n != 0 && !fast) slow[nslow + (uint32_t)__popcll(sm & ((1ull << lane) - 1))] = (uint16_t)ci;
        nslow += (uint32_t)__popcll(sm);
        cover_lane4<kCoverDefer>(C, mrec, b0c, fast ? n : 0u, 0xFu, T.a0 + ci, tied, tied + 4);
    }
    if (lane == 0) A.ptile[i].pad[0] = nslow;
}
// ... and what cover_lane4 left over, eight components to a wave (cover_tiny8), a wave per tile again
__global__ __launch_bounds__(256) void k_pc_tiny8(P2Args A) {
    if (A.st->err_code) return;
    __shared__ uint32_t s_stage[4][kStageWordsGL];   // (per wave: 64 label rows, and eight gene rows behind them - the GL instances of the covers)
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6, i = blockIdx.x * 4 + wv;
    if (i >= A.n_tiles) return;
    const PfTile T = A.ptile[i];
    if (!T.live || !T.na || !T.pad[0]) return;
    const uint32_t j = T.j;
    const P2Cell c = A.cells[j];
    const PfDev& D = *A.pfd;
    PugCtx C = make_ctx(A, c, A.gcnt + 4 * (size_t)j);
    C.adj_umi = 1;
    const uint4* mrec = reinterpret_cast<const uint4*>(A.pool + D.mrec);
    const uint32_t* mid_off = A.pool + D.midoff + T.comp_base;
    uint32_t* const tied = A.pool + D.tied + 4ull * (j + T.comp_base);
    const uint16_t* slow = reinterpret_cast<const uint16_t*>(A.pool + D.slow) + T.comp_base + T.a0;
    cover_tiny8<1, kCoverDefer, true>(C, mrec, mid_off + T.a0, T.pad[0], 0u, lane, tied, tied + 4, s_stage[wv], T.a0, slow);
}
__global__ __launch_bounds__(256) void k_pc_mid(P2Args A) {
    if (A.st->err_code) return;
    __shared__ uint32_t s_stage[4][kStageWordsGL];   // (per wave: 64 label rows, and eight gene rows behind them - the GL instances of the covers)
    const PfTile T = A.ptile[blockIdx.x];
    if (!T.live || !T.nb) return;
    const uint32_t j = T.j, lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const P2Cell c = A.cells[j];
    const PfDev& D = *A.pfd;
    PugCtx C = make_ctx(A, c, A.gcnt + 4 * (size_t)j);
    C.adj_umi = 1;
    const uint4* mrec = reinterpret_cast<const uint4*>(A.pool + D.mrec);
    const uint32_t* mid_off = A.pool + D.midoff + T.comp_base;
    uint32_t* const tied = A.pool + D.tied + 4ull * (j + T.comp_base);
    cover_wave64<4, kCoverDefer, true>(C, mrec, mid_off + T.b0, 0u, T.nb, wv, lane, tied + 1, tied + 4 + 4 * (size_t)T.n_tiny, s_stage[wv], T.b0);
}
// 10. the components the covers set aside at a tie, once k_p2_tied has left their classes' smallest record offsets beside their
//     records (PfDev.cmv): into the reference's order - class by first appearance, then UMI (pugutils.rs:1090-1160 takes the first
//     largest arborescence it meets) - and on with their covers.  A cell's two lists are dealt to the waves of its tiles, 64
//     entries at a time: a component of 3..4 vertices under short labels is renumbered and finished in ONE LANE's registers
//     (lane4_permute, lane4_rounds); the others have their records rewritten in order - a group of eight lanes, or the wave, per
//     component - and cover_tiny8 / cover_wave64 resume them.  (The UMI of an order key lies in the record: nothing here reads a
//     per-slot array.)
__global__ __launch_bounds__(256) void k_pc_resume(P2Args A) {
    if (A.st->err_code) return;
    __shared__ uint32_t s_stage[4][kStageWordsGL];   // (per wave: 64 label rows, and eight gene rows behind them - the GL instances of the covers)
    __shared__ uint16_t s_slow[4][64];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6, i = blockIdx.x * 4 + wv;
    if (i >= A.n_tiles) return;
    const PfTile T = A.ptile[i];
    if (!T.live) return;
    const uint32_t j = T.j;
    const PfDev& D = *A.pfd;
    uint32_t* tied = A.pool + D.tied + 4ull * (j + T.comp_base);
    const uint32_t nA = tied[0], nB = tied[1];
    if (!(nA | nB)) return;
    const P2Cell c = A.cells[j];
    const uint32_t wg = i - c.tile0, n_wg = (c.R + A.tile - 1) / A.tile;
    uint32_t* gc = A.gcnt + 4 * (size_t)j;
    PugCtx C = make_ctx(A, c, gc);
    C.adj_umi = 1;
    uint32_t* listA = tied + 4;
    uint32_t* listB = tied + 4 + 4 * (size_t)T.n_tiny;
    const uint32_t* mid_off = A.pool + D.midoff + T.comp_base;
    uint4* mrec = reinterpret_cast<uint4*>(A.pool + D.mrec);
    const uint32_t* cmv = A.pool + D.cmv;
    // the order key of the vertex in record slot `at`: still uncovered: (its class's smallest record offset, its UMI); covered: behind them, as it lies
    auto order_key = [&](bool act, uint32_t pos, uint64_t uc, uint32_t at, uint32_t umi) -> uint64_t {
        if (!act) return ~0ull;
        if (!((uc >> pos) & 1ull)) return (1ull << 63) | pos;
        const uint32_t mn = cmv[at];
        if (mn >> 31) gc[3] = kErrInternal;
        return ((uint64_t)(mn & 0x7FFFFFFFu) << 32) | umi;
    };
    // ---- the 3..8 list ----
    for (uint32_t e0 = wg * 64; e0 < nA; e0 += n_wg * 64) {   // (uniform per wave)
        const uint32_t e = e0 + lane;
        uint32_t b0c = 0, n = 0, uc = 0, ci = 0;
        uint32_t* en = listA + 4 * (size_t)(e < nA ? e : 0u);
        if (e < nA) { ci = en[0]; uc = en[1] & 0xFFu; b0c = mid_off[ci]; n = mid_off[ci + 1] - b0c; }
        bool fast = n != 0 && n <= 4;
        Lane4 L;
        lane4_load(C, mrec, b0c, fast ? n : 0u, L);
#pragma unroll
        for (int v = 0; v < 4; ++v) fast = fast && L.ln[v] <= 4;
        const uint64_t sm = __ballot(n != 0 && !fast);
        if (n != 0 && !fast) s_slow[wv][__popcll(sm & ((1ull << lane) - 1))] = (uint16_t)lane;   // (the entry's place in this trip's 64)
        const uint32_t nslow = (uint32_t)__popcll(sm);
        if (!fast) uc = 0;
        uint64_t key[4];
        uint32_t rank[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) key[v] = order_key(fast && (uint32_t)v < n, (uint32_t)v, uc, b0c + (uint32_t)v, L.um[v]);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            rank[v] = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                rank[v] += (u != v && key[u] < key[v]) ? 1u : 0u;
                if (u < v && key[u] == key[v] && key[v] != ~0ull) gc[3] = kErrInternal;   // (two vertices of one component with the same class and UMI: cannot be)
            }
            if (!(fast && (uint32_t)v < n)) rank[v] = (uint32_t)v;   // (past the component's end: they stay where they are)
        }
        lane4_permute(L, rank, uc);
        lane4_rounds<kCoverResume>(C, L, uc, ci, nullptr, nullptr);
        if (!nslow) continue;   // (uniform)
        // the others: their records into the reference's order where they lie (a group of eight lanes per component), then cover_tiny8 resumes them
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        {
            const uint32_t gl = lane & 7u, gbase = lane & ~7u, grp = lane >> 3;
            for (uint32_t q0 = 0; q0 < nslow; q0 += 8) {   // (uniform)
                const uint32_t q = q0 + grp;
                const bool mine = q < nslow;
                uint32_t* en2 = listA + 4 * (size_t)(e0 + (mine ? s_slow[wv][q] : 0u));
                const uint32_t ci2 = mine ? en2[0] : 0u;
                const uint32_t uc2 = mine ? en2[1] & 0xFFu : 0u;
                const uint32_t b0 = mine ? mid_off[ci2] : 0u, n2 = mine ? mid_off[ci2 + 1] - b0 : 0u;
                const bool act = gl < n2;
                uint4 qa = make_uint4(0, 0, 0, 0), qb = qa;
                if (act) { qa = mrec[2 * (size_t)(b0 + gl)]; qb = mrec[2 * (size_t)(b0 + gl) + 1]; }
                const uint64_t key2 = order_key(act, gl, uc2, b0 + gl, qb.z);
                uint32_t rk = 0;
#pragma unroll
                for (uint32_t k = 0; k < 8; ++k) {
                    const uint64_t kk = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(key2 >> 32), (int)(gbase + k)) << 32) | (uint32_t)__shfl((int)(uint32_t)key2, (int)(gbase + k));
                    rk += k < n2 && kk < key2 ? 1u : 0u;
                    if (act && k < n2 && k != gl && kk == key2) gc[3] = kErrInternal;
                }
                uint32_t nuc = 0;
#pragma unroll
                for (uint32_t k = 0; k < 8; ++k) {
                    const uint32_t rkk = (uint32_t)__shfl((int)rk, (int)(gbase + k));
                    if (k < n2 && ((uc2 >> k) & 1u)) nuc |= 1u << rkk;
                }
                if (act) {   // (every lane holds its record in registers: the slots can be overwritten; the records keep their (UMI, reads))
                    mrec[2 * (size_t)(b0 + rk)] = qa;
                    mrec[2 * (size_t)(b0 + rk) + 1] = qb;
                }
                if (mine && gl == 0) { en2[1] = nuc; en2[2] = 0u; }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        cover_tiny8<1, kCoverResume, true>(C, mrec, mid_off, nslow, 0u, lane, nullptr, listA + 4 * (size_t)e0, s_stage[wv], 0u, s_slow[wv]);
    }
    // ---- the 9..64 list: this wave's run of it, a component at a time ----
    const uint32_t b_per = (nB + n_wg - 1) / n_wg, b_lo = min(nB, wg * b_per), b_hi = min(nB, b_lo + b_per);
    for (uint32_t e = b_lo; e < b_hi; ++e) {   // (uniform)
        uint32_t* en = listB + 4 * (size_t)e;
        const uint32_t ci = en[0];
        const uint64_t uc = ((uint64_t)en[2] << 32) | en[1];
        const uint32_t b0 = mid_off[ci], n = mid_off[ci + 1] - b0;
        const bool act = lane < n;
        uint4 qa = make_uint4(0, 0, 0, 0), qb = qa;
        if (act) { qa = mrec[2 * (size_t)(b0 + lane)]; qb = mrec[2 * (size_t)(b0 + lane) + 1]; }
        const uint64_t key = order_key(act, lane, uc, b0 + lane, qb.z);
        uint32_t rk = 0;
        for (uint32_t k = 0; k < n; ++k) {
            const uint64_t kk = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(key >> 32), (int)k) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, (int)k);
            rk += kk < key ? 1u : 0u;
            if (kk == key && k != lane && act) gc[3] = kErrInternal;
        }
        uint64_t nuc = 0;
        for (uint32_t k = 0; k < n; ++k) {
            const uint32_t rkk = (uint32_t)__builtin_amdgcn_readlane((int)rk, (int)k);
            if ((uc >> k) & 1ull) nuc |= 1ull << rkk;
        }
        if (act) { mrec[2 * (size_t)(b0 + rk)] = qa; mrec[2 * (size_t)(b0 + rk) + 1] = qb; }
        if (lane == 0) { en[1] = (uint32_t)nuc; en[2] = (uint32_t)(nuc >> 32); }
    }
    if (b_hi > b_lo) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        cover_wave64<1, kCoverResume, true>(C, mrec, mid_off, 0u, b_hi - b_lo, 0u, lane, nullptr, listB + 4 * (size_t)b_lo, s_stage[wv]);
    }
}
// ... and a thread per cell: what the covers left in the cell's counters (an error of theirs, the lengths of its column list and
// label area) to where the kernels behind them read it
__global__ __launch_bounds__(256) void k_pc_finish(P2Args A) {
    if (A.st->err_code) return;
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= A.n_cells || A.fb[j] || A.route[j]) return;
    const uint32_t* gc = A.gcnt + 4 * (size_t)j;
    const uint32_t cell = A.cells[j].cell;
    if (gc[3]) { set_err(A.st, gc[3], cell); return; }
    A.cell_ncols[cell] = gc[0];
    if (A.lab_cnt) { A.lab_cnt[2 * cell] = gc[1]; A.lab_cnt[2 * cell + 1] = gc[2]; }
}

// ---------------------------------------------------------------------------------------------------------------------------
static uint32_t pf_grid(uint64_t items_upper, uint32_t per_wg) { const uint64_t full = (items_upper + per_wg - 1) / per_wg; return (uint32_t)std::min<uint64_t>(std::max<uint64_t>(full, 1), 8192); }

void launch_pf_build(hipStream_t s, const P2Args& a, uint64_t n_reads) {
    if (!a.n_cells) return;
    const uint32_t nb = (a.n_tiles + 1 + 1023) / 1024;
    AFQ_LAUNCH(k_pf_count, a.n_tiles, 256, s, a);
    AFQ_LAUNCH(k_pf_scan1<1>, nb, 1024, s, a, 0u);
    AFQ_LAUNCH(k_pf_scan2<1>, 1, 1024, s, a, 0u);
    AFQ_LAUNCH(k_pf_number, a.n_tiles, 256, s, a);
    AFQ_LAUNCH(k_pf_union, pf_grid(a.n_parts, 256), 256, s, a);
    // (T is known on the device only: the grid from the reads, an upper bound of it; the kernel walks to T)
    AFQ_LAUNCH(k_pf_root, pf_grid(n_reads / 4 + 1, 1024), 256, s, a);
    AFQ_LAUNCH(k_pf_cats, a.n_tiles, 256, s, a);
    AFQ_LAUNCH(k_pf_scan1<5>, nb, 1024, s, a, 1u);
    AFQ_LAUNCH(k_pf_scan2<5>, 1, 1024, s, a, 1u);
    if (a.em && a.lab) {
        AFQ_LAUNCH(k_pf_pscan1, a.npa / 1024, 1024, s, a);
        AFQ_LAUNCH(k_pf_pscan2, 1, 1024, s, a);
        AFQ_LAUNCH(k_pf_move, pf_grid(a.n_parts, 256), 256, s, a);   // (gcnt[1], [2] still hold what k_p2_lone left: k_pf_cells raises them afterwards)
    }
    AFQ_LAUNCH(k_pf_cells, (a.n_cells + 255) / 256, 256, s, a);
    AFQ_LAUNCH(k_pf_tiles, (a.n_tiles + 255) / 256, 256, s, a);
    AFQ_LAUNCH(k_pf_alloc, a.n_tiles, 256, s, a);
    AFQ_LAUNCH(k_pf_place, pf_grid(n_reads / 4 + 1, 512), 256, s, a);
}
void launch_pf_cover(hipStream_t s, const P2Args& a) {
    if (!a.n_cells) return;
    AFQ_LAUNCH(k_pc_pairs, a.n_tiles, 256, s, a);
    AFQ_LAUNCH(k_pc_lane4, (a.n_tiles + 3) / 4, 256, s, a);
    AFQ_LAUNCH(k_pc_tiny8, (a.n_tiles + 3) / 4, 256, s, a);
    AFQ_LAUNCH(k_pc_mid, a.n_tiles, 256, s, a);
}
void launch_pf_resume(hipStream_t s, const P2Args& a) {
    if (!a.n_cells) return;
    AFQ_LAUNCH(k_pc_resume, (a.n_tiles + 3) / 4, 256, s, a);
    AFQ_LAUNCH(k_pc_finish, (a.n_cells + 255) / 256, 256, s, a);
}

}  // namespace afq
