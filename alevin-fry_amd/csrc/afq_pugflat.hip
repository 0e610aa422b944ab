// afq_pugflat.hip — the graph phase of the parsimony path as RANGE-WIDE FLAT kernels (gfx950, wave64).
//
// Same semantics as the per-cell graph kernel of afq_pug2.hip (reference paths relative to /root/reference):
//   weakly_connected_components          src/pugutils.rs:278-301
//   collapse_vertices / get_num_molecules src/pugutils.rs:308-391, 1094-1146   (the covers themselves: afq_pug_common.h)
// but cut by what the work is made of, not by the cell: rounds 3-5 gave every cell one workgroup that walked some forty
// dependent steps behind workgroup barriers (k_p2_graph: 128 VGPRs, 40 spilled, 98 KB of LDS, 39.8 GB per configs[2] step
// for a union-find).  Here every step is one kernel over the whole range of cells:
//
//   k_pf_count / k_pf_tscan / k_pf_number   the vertices that have an edge (the search flagged them) get dense numbers t in
//                                 [0, T): a tile of 4096 read slots counts its flags, one workgroup scans the tile counts,
//                                 the tiles number their vertices.  Everything behind this works on dense arrays of T entries.
//   k_pf_union                    one thread per PAIR: compare-and-swap hooking union-find on par[T] (the larger root under the
//                                 smaller, path halving by atomic min); the pair is rewritten over dense numbers where it lies.
//   k_pf_root                     one thread per vertex: its root, its position inside its component (an atomic counter per root).
//   k_pf_cats / k_pf_classes / k_pf_cscan   component sizes -> per cell: how many components of two / 3..8 / 9..64 vertices and
//                                 how many record slots; a cell with a larger component is routed to the per-cell kernel (rare);
//                                 one workgroup scans the cells: every cell's components become a contiguous run of the
//                                 range-wide lists (pairs | 3..8 | 9..64), the lists come out of the pool in one step.
//   k_pf_alloc / k_pf_place / k_pf_adj      roots take their list entry and record slots, vertices write their 32-byte cover
//                                 records, pairs OR their directions into the records' adjacency masks.
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

__device__ __forceinline__ uint32_t ag_ld(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// size class of a component of n vertices; 0: not for the flat lists - more than 64 vertices, or above --large-graph-thresh (resolved
// winner-take-all, pugutils.rs:916-982): the per-cell kernel takes the cell
__device__ __forceinline__ uint32_t cat_of(uint32_t n, uint32_t large_thresh) { return n > large_thresh ? 0u : n == 2 ? kFCatPair : n <= 8 ? kFCatTiny : n <= 64 ? kFCatMid : 0u; }

// the dense arrays of the touched vertices, out of the pool (k_pf_tscan put them there)
struct PfV {
    uint32_t T;
    uint32_t* tl;      // t -> read slot (range-wide: rd_base + slot inside the cell)
    uint32_t* tcell;   // t -> cell (index into P2Args.cells)
    uint32_t* par;     // union-find parent; after k_pf_root: the root
    uint32_t* cnt;     // root: vertices of its component (k_pf_root); then where its component lies (k_pf_alloc)
    uint32_t* rk;      // position inside the component | category << 29 (roots, k_pf_alloc)
};
__device__ __forceinline__ PfV pf_v(const P2Args& A) {
    const PfDev& D = *A.pfd;
    PfV v;
    v.T = D.T; v.tl = A.pool + D.tl; v.tcell = A.pool + D.tcell; v.par = A.pool + D.par; v.cnt = A.pool + D.cnt; v.rk = A.pool + D.rk;
    return v;
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
        const uint8_t* f = A.v_flag + c.rd_base;
        for (uint32_t g0 = t0 + 16 * threadIdx.x; g0 < t1; g0 += 16 * 256) {
#pragma unroll
            for (int r = 0; r < 16; ++r) n += g0 + r < t1 && f[g0 + r] != 0;
        }
    }
    uint32_t tot;
    (void)block_excl_scan<256>(n, s_ws, tot);
    if (threadIdx.x == 0) A.tcount[blockIdx.x] = tot;
}

// one workgroup: tile counts -> tile offsets, T, and the dense arrays out of the pool
__global__ __launch_bounds__(1024) void k_pf_tscan(P2Args A) {
    if (A.st->err_code) return;
    __shared__ uint32_t s_ws[16];
    uint32_t carry = 0;
    for (uint32_t b = 0; b < A.n_tiles; b += 1024) {
        const uint32_t i = b + threadIdx.x;
        const uint32_t v = i < A.n_tiles ? A.tcount[i] : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan<1024>(v, s_ws, tot);
        if (i < A.n_tiles) A.tbase[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) {
        PfDev& D = *A.pfd;
        const unsigned long long T = carry, words = 5 * T + 16;
        const unsigned long long base = atomicAdd(A.pool_cur, words);
        if (base + words > A.pool_cap || T >= (1u << 31)) { set_err(A.st, T >= (1u << 31) ? kErrPugLimit : kErrPugPool, 0); D.T = 0; return; }
        unsigned long long o = (base + 3) & ~3ull;
        D.tl = o; o += T; D.tcell = o; o += T; D.par = o; o += T; D.cnt = o; o += T; D.rk = o;
        D.T = (uint32_t)T;
        A.tbase[A.n_tiles] = (uint32_t)T;
    }
}

__global__ __launch_bounds__(256) void k_pf_number(P2Args A) {
    if (A.st->err_code) return;
    __shared__ uint32_t s_ws[4];
    const uint2 td = A.tiles[blockIdx.x];
    const uint32_t j = td.x;
    if (A.fb[j]) return;
    const PfV V = pf_v(A);
    const P2Cell c = A.cells[j];
    const uint32_t t0 = td.y * A.tile, t1 = min(c.R, t0 + A.tile);
    const uint8_t* f = A.v_flag + c.rd_base;
    uint32_t* lidx = A.lidx + c.rd_base;
    uint32_t carry = A.tbase[blockIdx.x];
    for (uint32_t base = t0; base < t1; base += 16 * 256) {   // (one trip: a tile is 4096 slots)
        const uint32_t g0 = base + 16 * threadIdx.x;
        uint32_t fm = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) fm |= (uint32_t)(g0 + r < t1 && f[g0 + r] != 0) << r;
        uint32_t tot;
        uint32_t t = carry + block_excl_scan<256>((uint32_t)__popc(fm), s_ws, tot);
        for (; fm; fm &= fm - 1, ++t) {
            const uint32_t g = g0 + (uint32_t)__builtin_ctz(fm);
            lidx[g] = t;
            V.tl[t] = (uint32_t)c.rd_base + g; V.tcell[t] = j; V.par[t] = t; V.cnt[t] = 0; V.rk[t] = 0;
        }
        carry += tot;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The pairs the search left, a workgroup over 256 consecutive partitions at a time: every thread brings one partition's count and
// place, a scan lays the partitions' pairs end to end, and the threads take the pairs of the 256 partitions evenly - a partition
// holds anything from none to hundreds.  f(where the pair lies, first read slot of its cell).
template <typename F>
__device__ __forceinline__ void pf_for_each_pair(const P2Args& A, uint32_t* s_start, unsigned long long* s_src, uint32_t* s_rdb, uint32_t* s_ws, F&& f) {
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
        for (uint32_t t = threadIdx.x; t < tot; t += 256) {
            uint32_t lo = 0, hi = 256;   // the last partition that starts at or before t: the one t falls into (empty ones share their start with the next)
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (s_start[mid] <= t) lo = mid; else hi = mid; }
            f(reinterpret_cast<uint64_t*>((uintptr_t)s_src[lo]) + (t - s_start[lo]), s_rdb[lo]);
        }
        __syncthreads();
    }
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
    auto find = [&](uint32_t i) -> uint32_t {
        uint32_t p = ag_ld(&par[i]);
        while (p != i) {
            const uint32_t gp2 = ag_ld(&par[p]);
            if (gp2 != p) __hip_atomic_fetch_min(&par[i], gp2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            i = p; p = gp2;
        }
        return i;
    };
    pf_for_each_pair(A, s_start, s_src, s_rdb, s_ws, [&](uint64_t* sp, uint32_t rdb) {
        const uint64_t e = *sp;
        const uint32_t tx = A.lidx[(size_t)rdb + ((uint32_t)(e >> 31) & 0x7FFFFFFFu)], ty = A.lidx[(size_t)rdb + ((uint32_t)e & 0x7FFFFFFFu)];
        *sp = (e & (kPairF | kPairB)) | ((uint64_t)tx << 31) | ty;   // the pair over dense numbers, where it lay (k_pf_adj, and the per-cell kernel for the cells routed to it)
        if (tx >= V.T || ty >= V.T) { set_err(A.st, kErrInternal, 0); return; }
        uint32_t a = find(tx), b = find(ty);
        while (a != b) {
            if (a < b) { const uint32_t t = a; a = b; b = t; }
            uint32_t expected = a;
            if (__hip_atomic_compare_exchange_strong(&par[a], &expected, b, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
            a = find(expected); b = find(b);
        }
    });
}

// 3. one thread per vertex: its root (written over its parent: later kernels read it with one load), its position inside its component
__global__ __launch_bounds__(256) void k_pf_root(P2Args A) {
    if (A.st->err_code) return;
    const PfV V = pf_v(A);
    for (uint32_t t = blockIdx.x * 256 + threadIdx.x; t < V.T; t += gridDim.x * 256) {
        uint32_t r = t;
        for (uint32_t p = V.par[r]; p != r; p = V.par[r]) r = p;
        V.rk[t] = atomicAdd(&V.cnt[r], 1u);
        if (r != t) V.par[t] = r;   // (a walk through t meanwhile meets the old parent or the root: both are ancestors)
    }
}

// 4. one thread per root: its component's size class into its cell's counts.  The lanes of a wave nearly always share a cell
//    (dense numbers run cell by cell): one atomic per wave, count and cell; a wave across a cell boundary takes its cells in turn.
__global__ __launch_bounds__(256) void k_pf_cats(P2Args A) {
    if (A.st->err_code) return;
    const PfV V = pf_v(A);
    const uint32_t lane = lane_id();
    for (uint32_t t0 = (blockIdx.x * 256 + threadIdx.x) & ~63u; t0 < V.T; t0 += gridDim.x * 256) {   // (wave-uniform trip count)
        const uint32_t t = t0 + lane;
        const bool root = t < V.T && V.par[t] == t;
        const uint32_t n = root ? V.cnt[t] : 0u, j = root ? V.tcell[t] : 0u;
        const uint32_t cat = root ? cat_of(n, A.large_thresh) : 0u;
        for (uint64_t left = __ballot(root); left;) {
            const uint32_t jj = (uint32_t)__builtin_amdgcn_readlane((int)j, (int)__builtin_ctzll(left));
            const bool mine = root && j == jj;
            left &= ~__ballot(mine);
            PfCell* pc = A.pfc + jj;
            const uint32_t npr = (uint32_t)__popcll(__ballot(mine && cat == kFCatPair)), nti = (uint32_t)__popcll(__ballot(mine && cat == kFCatTiny)),
                           nmi = (uint32_t)__popcll(__ballot(mine && cat == kFCatMid));
            uint32_t sti = mine && cat == kFCatTiny ? n : 0u, smi = mine && cat == kFCatMid ? n : 0u;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) { sti += __shfl_xor(sti, d); smi += __shfl_xor(smi, d); }
            const bool big = __any(mine && cat == 0);
            if (lane == 0) {
                if (npr) atomicAdd(&pc->n_pr, npr);
                if (nti) { atomicAdd(&pc->n_tiny, nti); atomicAdd(&pc->S_tiny, sti); }
                if (nmi) { atomicAdd(&pc->n_mid, nmi); atomicAdd(&pc->S_mid, smi); }
                if (big) atomicOr(&pc->route, 1u);   // a component of more than 64 vertices: the cell goes to the per-cell kernel
            }
        }
    }
}

// 5. a wave per cell: the lone vertices' two-gene classes (k_p2_lone staged them at their partitions' slots) into the cell's label
//    area, and the cell's counters as the covers start from them (em only for the classes; always for the counters).
__global__ __launch_bounds__(256) void k_pf_classes(P2Args A) {
    if (A.st->err_code) return;
    const uint32_t j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = lane_id();
    if (j >= A.n_cells || A.fb[j] || A.pfc[j].route) return;   // (the per-cell kernel moves the classes of the cells routed to it itself)
    const P2Cell c = A.cells[j];
    uint32_t* gc = A.gcnt + 4 * (size_t)j;
    uint32_t w0 = gc[1], d0 = gc[2];
    if (A.em && A.lab) {
        uint32_t* labw = A.lab + 2 * c.key_off;
        uint32_t* labd = labw + c.n_ref + 1;
        const uint32_t lab_cap = c.n_ref + 1;
        const uint64_t* stage = A.cstage + c.rd_base;
        const uint32_t P = 1u << c.lgP;
        bool over = false;
        for (uint32_t base = 0; base < P; base += 64) {
            const uint32_t pp = base + lane;
            const uint32_t nk = pp < P ? A.pncls[c.part_base + pp] : 0u;
            uint32_t tot;
            const uint32_t at = wave_excl_scan(nk, tot);
            if (tot == 0) continue;
            if (w0 + 2 * tot > lab_cap || 2 * (d0 + tot) > lab_cap) { over = true; break; }
            const uint32_t so = nk ? A.poff[c.part_base + pp] : 0u;
            for (uint32_t k = 0; k < nk; ++k) {
                const uint64_t v = stage[so + k];
                const uint32_t w = w0 + 2 * (at + k), d = d0 + at + k;
                labw[w] = (uint32_t)v; labw[w + 1] = (uint32_t)(v >> 32);
                labd[2 * d] = w; labd[2 * d + 1] = 2;
            }
            w0 += 2 * tot; d0 += tot;
        }
        if (over) { if (lane == 0) set_err(A.st, kErrPugLimit, c.cell); return; }
    }
    if (lane == 0) { gc[0] = c.R; gc[1] = w0; gc[2] = d0; }   // (entries 0..R of the column list belong to the lone-vertex kernel)
}

// 6. one workgroup: the cells' counts -> where every cell's components and record slots lie in the range-wide lists, the lists out
//    of the pool, the descriptors the cover kernels read (the same sixteen-plus words per cell the per-cell graph kernel leaves).
__global__ __launch_bounds__(1024) void k_pf_cscan(P2Args A) {
    if (A.st->err_code) return;
    __shared__ uint32_t s_ws[16];
    __shared__ uint32_t s_ok;
    __shared__ unsigned long long s_off[4];   // mrec, prv, midoff, tied
    PfDev& D = *A.pfd;
    uint32_t c_pr = 0, c_comp = 0, c_slot = 0;
    for (uint32_t b = 0; b < A.n_cells; b += 1024) {
        const uint32_t j = b + threadIdx.x;
        uint32_t npr = 0, ncomp = 0, nslot = 0;
        if (j < A.n_cells) {
            PfCell& pc = A.pfc[j];
            const bool fbk = A.fb[j] != 0, old = !fbk && pc.route != 0;
            if (fbk || old) { pc.n_pr = 0; pc.n_tiny = 0; pc.n_mid = 0; pc.S_tiny = 0; pc.S_mid = 0; }
            if (fbk) A.fb_list[atomicAdd(A.fb_count, 1u)] = A.cells[j].cell;   // (k_p2_scan flagged it: a partition over the capacity)
            if (old) A.old_list[atomicAdd(&D.n_old, 1u)] = j;
            npr = pc.n_pr; ncomp = pc.n_tiny + pc.n_mid; nslot = pc.S_tiny + pc.S_mid;
        }
        uint32_t t0, t1, t2;
        const uint32_t e0 = block_excl_scan<1024>(npr, s_ws, t0);
        const uint32_t e1 = block_excl_scan<1024>(ncomp, s_ws, t1);
        const uint32_t e2 = block_excl_scan<1024>(nslot, s_ws, t2);
        if (j < A.n_cells) { PfCell& pc = A.pfc[j]; pc.pr_base = c_pr + e0; pc.comp_base = c_comp + e1; pc.slot_base = c_slot + e2; }
        c_pr += t0; c_comp += t1; c_slot += t2;
    }
    if (threadIdx.x == 0) {
        const unsigned long long NP = c_pr, NC = c_comp, S = c_slot;
        const unsigned long long words = 2 * NP + 4 + (NC + 4) + 8 * S + 8 + 4 * (NC + A.n_cells) + 8;
        const unsigned long long base = atomicAdd(A.pool_cur, words);
        s_ok = base + words <= A.pool_cap;
        if (!s_ok) set_err(A.st, kErrPugPool, 0);
        else {
            unsigned long long o = (base + 3) & ~3ull;
            D.mrec = o; o += 8 * S + 4;            // (16-byte aligned: uint4 records)
            D.prv = o; o += 2 * NP + 2;
            D.midoff = o; o += NC + 2;
            o = (o + 3) & ~3ull;
            D.tied = o;
            D.NP = (uint32_t)NP; D.NC = (uint32_t)NC; D.S = (uint32_t)S;
            A.pool[D.midoff + NC] = (uint32_t)S;   // the list's last offset
            s_off[0] = D.mrec; s_off[1] = D.prv; s_off[2] = D.midoff; s_off[3] = D.tied;
        }
    }
    __syncthreads();
    if (!s_ok) return;
    for (uint32_t j = threadIdx.x; j < A.n_cells; j += 1024) {
        const PfCell& pc = A.pfc[j];
        if (A.fb[j] || pc.route) continue;   // handed back (descriptor stays 0), or the per-cell graph kernel writes it
        uint32_t* d = A.gdesc + kGDescWords * (size_t)j;
        auto put = [&](int at, unsigned long long o) { d[at] = (uint32_t)o; d[at + 1] = (uint32_t)(o >> 32); };
        const unsigned long long tied = s_off[3] + 4ull * (j + pc.comp_base);
        d[1] = pc.n_pr; d[2] = pc.n_tiny; d[3] = pc.n_tiny + pc.n_mid; d[15] = 0;
        d[4] = 0xFFFFFFFFu; d[5] = 0xFFFFFFFFu;   // (no table from touched-vertex numbers to slots: the lists hold slots)
        put(6, s_off[1] + 2ull * pc.pr_base); put(8, s_off[2] + pc.comp_base); put(10, s_off[0]); put(16, tied);
        const uint32_t* gc = A.gcnt + 4 * (size_t)j;
        d[12] = gc[0]; d[13] = gc[1]; d[14] = gc[2];
        uint32_t* tp = A.pool + tied;
        tp[0] = 0; tp[1] = 0; tp[2] = 0; tp[3] = 0;
        d[0] = 3u;   // the lists are there, in slot order (kCoverDefer)
    }
}

// 7. one thread per root: its component's entry of its cell's run of the lists, its record slots
__global__ __launch_bounds__(256) void k_pf_alloc(P2Args A) {
    if (A.st->err_code) return;
    const PfV V = pf_v(A);
    const PfDev& D = *A.pfd;
    uint32_t* mid_off = A.pool + D.midoff;
    const uint32_t lane = lane_id();
    for (uint32_t t0 = (blockIdx.x * 256 + threadIdx.x) & ~63u; t0 < V.T; t0 += gridDim.x * 256) {
        const uint32_t t = t0 + lane;
        const bool root = t < V.T && V.par[t] == t;
        const uint32_t n = root ? V.cnt[t] : 0u, j = root ? V.tcell[t] : 0u;
        uint32_t cat = root ? cat_of(n, A.large_thresh) : 0u;
        uint32_t where = 0;
        for (uint64_t left = __ballot(root); left;) {
            const uint32_t jj = (uint32_t)__builtin_amdgcn_readlane((int)j, (int)__builtin_ctzll(left));
            const bool mine = root && j == jj;
            left &= ~__ballot(mine);
            PfCell* pc = A.pfc + jj;
            if (pc->route) { if (mine) cat = 0; continue; }   // (the cell is the per-cell kernel's)
            // one reservation per wave, cell and size class: components | record slots << 32
            auto take = [&](bool on, uint32_t sz, unsigned long long* ctr, uint32_t& idx, uint32_t& sl) {
                const uint64_t m = __ballot(on);
                if (!m) return;
                uint32_t tot;
                const uint32_t ex = wave_excl_scan(on ? sz : 0u, tot);
                unsigned long long b0 = 0;
                const uint32_t leader = (uint32_t)__builtin_ctzll(m);
                if (lane == leader) b0 = atomicAdd(ctr, (unsigned long long)__popcll(m) | ((unsigned long long)tot << 32));
                const uint32_t blo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b0, (int)leader), bhi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b0 >> 32), (int)leader);
                idx = blo + (uint32_t)__popcll(m & ((1ull << lane) - 1)); sl = bhi + ex;
            };
            uint32_t idx = 0, sl = 0;
            take(mine && cat == kFCatPair, 2u, &pc->fill_pr, idx, sl);
            if (mine && cat == kFCatPair) where = 2 * (pc->pr_base + idx);
            take(mine && cat == kFCatTiny, n, &pc->fill_tiny, idx, sl);
            if (mine && cat == kFCatTiny) { where = pc->slot_base + sl; mid_off[pc->comp_base + idx] = where; }
            take(mine && cat == kFCatMid, n, &pc->fill_mid, idx, sl);
            if (mine && cat == kFCatMid) { where = pc->slot_base + pc->S_tiny + sl; mid_off[pc->comp_base + pc->n_tiny + idx] = where; }
        }
        if (root) { V.cnt[t] = where; V.rk[t] = (V.rk[t] & kRankMask) | (cat << kCatShift); }
    }
}

// 8. one thread per vertex: into the pair list, or its 32-byte cover record (vertex slot, label length, up to four refs - a longer
//    label: where it lies in the chunk -, adjacency mask, filled by k_pf_adj)
__global__ __launch_bounds__(256) void k_pf_place(P2Args A) {
    if (A.st->err_code) return;
    const PfV V = pf_v(A);
    const PfDev& D = *A.pfd;
    uint32_t* pr_v = A.pool + D.prv;
    uint4* mrec = reinterpret_cast<uint4*>(A.pool + D.mrec);
    for (uint32_t t = blockIdx.x * 256 + threadIdx.x; t < V.T; t += gridDim.x * 256) {
        const uint32_t r = V.par[t];
        const uint32_t rkr = V.rk[r], cat = rkr >> kCatShift;
        if (!cat) continue;
        const uint32_t where = V.cnt[r], k = V.rk[t] & kRankMask;
        const P2Cell& c = A.cells[V.tcell[t]];
        const uint32_t s = V.tl[t], g = s - (uint32_t)c.rd_base;
        if (cat == kFCatPair) { pr_v[where + k] = g; continue; }
        const uint32_t* W = reinterpret_cast<const uint32_t*>(A.bytes + c.chunk_off);
        const KLab l = klab(W, A.hw, A.s_h[s], A.v_off[s]);
        uint32_t r0 = 0xFFFFFFFFu, r1 = 0xFFFFFFFFu, r2 = 0xFFFFFFFFu, r3 = 0xFFFFFFFFu;
        if (l.n <= 4) {
            if (l.n > 0) r0 = klab_ref(l, 0);
            if (l.n > 1) r1 = klab_ref(l, 1);
            if (l.n > 2) r2 = l.p[2] & 0x7FFFFFFFu;
            if (l.n > 3) r3 = l.p[3] & 0x7FFFFFFFu;
        } else { const uint64_t pa = (uint64_t)(uintptr_t)l.p; r0 = (uint32_t)pa; r1 = (uint32_t)(pa >> 32); }
        const size_t at = (size_t)where + k;
        mrec[2 * at] = make_uint4(g, l.n, r0, r1);
        mrec[2 * at + 1] = make_uint4(r2, r3, 0u, 0u);
    }
}

// 9. one thread per pair once more: its directions into the adjacency masks of its end points' records (pairs of a two-vertex
//    component need none: the two-vertex rule does not look at directions)
__global__ __launch_bounds__(256) void k_pf_adj(P2Args A) {
    if (A.st->err_code) return;
    __shared__ uint32_t s_start[256];
    __shared__ unsigned long long s_src[256];
    __shared__ uint32_t s_rdb[256];
    __shared__ uint32_t s_ws[4];
    const PfV V = pf_v(A);
    uint4* mrec = reinterpret_cast<uint4*>(A.pool + A.pfd->mrec);
    pf_for_each_pair(A, s_start, s_src, s_rdb, s_ws, [&](uint64_t* sp, uint32_t) {
        const uint64_t e = *sp;
        const uint32_t tx = (uint32_t)(e >> 31) & 0x7FFFFFFFu, ty = (uint32_t)e & 0x7FFFFFFFu;
        const uint32_t r = V.par[tx];
        const uint32_t cat = V.rk[r] >> kCatShift;
        if (cat != kFCatTiny && cat != kFCatMid) return;
        const size_t b0 = V.cnt[r];
        const uint32_t kx = V.rk[tx] & kRankMask, ky = V.rk[ty] & kRankMask;
        if (e & kPairF) __hip_atomic_fetch_or(reinterpret_cast<unsigned long long*>(&mrec[2 * (b0 + kx) + 1].z), 1ull << ky, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // x -> y
        if (e & kPairB) __hip_atomic_fetch_or(reinterpret_cast<unsigned long long*>(&mrec[2 * (b0 + ky) + 1].z), 1ull << kx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // y -> x
    });
}

// ---------------------------------------------------------------------------------------------------------------------------
static uint32_t pf_grid(uint64_t items_upper, uint32_t per_wg) { const uint64_t full = (items_upper + per_wg - 1) / per_wg; return (uint32_t)std::min<uint64_t>(std::max<uint64_t>(full, 1), 8192); }

void launch_pf_build(hipStream_t s, const P2Args& a, uint64_t n_reads) {
    if (!a.n_cells) return;
    AFQ_LAUNCH(k_pf_count, a.n_tiles, 256, s, a);
    AFQ_LAUNCH(k_pf_tscan, 1, 1024, s, a);
    AFQ_LAUNCH(k_pf_number, a.n_tiles, 256, s, a);
    AFQ_LAUNCH(k_pf_union, pf_grid(a.n_parts, 256), 256, s, a);
    // (T is known on the device only: grids from the reads, an upper bound of it; the kernels walk to T)
    const uint32_t gv = pf_grid(n_reads / 4 + 1, 256);
    AFQ_LAUNCH(k_pf_root, gv, 256, s, a);
    AFQ_LAUNCH(k_pf_cats, gv, 256, s, a);
    AFQ_LAUNCH(k_pf_classes, (a.n_cells + 3) / 4, 256, s, a);
    AFQ_LAUNCH(k_pf_cscan, 1, 1024, s, a);
    AFQ_LAUNCH(k_pf_alloc, gv, 256, s, a);
    AFQ_LAUNCH(k_pf_place, gv, 256, s, a);
    AFQ_LAUNCH(k_pf_adj, pf_grid(a.n_parts, 256), 256, s, a);
}

}  // namespace afq
