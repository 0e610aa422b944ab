// afq_pug2.hip — parsimony resolution as PHASE KERNELS over UMI partitions (gfx950, wave64).
//
// Same semantics as afq_pug.hip (reference paths relative to /root/reference):
//   EqMap::init_from_chunk               src/eq_class.rs:823-1036
//   extract_graph / has_edge             src/pugutils.rs:65-267, src/utils.rs:389-393
//   weakly_connected_components          src/pugutils.rs:278-301
//   collapse_vertices / get_num_molecules src/pugutils.rs:308-391, 989-1331
// but organised around what the work is made of instead of around the cell:
//
//   * Everything that touches every read or every vertex - grouping reads into vertices (label, UMI), the one-base
//     neighbour search, the molecules of vertices that have no edge (85 % of them) - runs wave-per-PARTITION: a cell's
//     reads are cut by the low m bits of their UMI into P = 2^m partitions of <= 256 reads (planned mean 80..160), a
//     partition is sorted in one wave's registers, and its hash table for the search is a few KiB of LDS.  A one-base
//     neighbour of a UMI lies in the UMI's own partition unless the change touches the low m bits, and then in exactly
//     one other partition - so the search of a partition is its own vertices plus one pass over the vertices of the
//     partitions a low-bit change away.  Thousands of independent waves, 64-thread workgroups, no workgroup barriers,
//     and no per-cell scratch slice streamed through HBM phase after phase.
//   * Only the vertices that HAVE an edge (~15 %) reach the per-cell graph kernel: components, the two-vertex rule and
//     the arborescence covers (shared with afq_pug.hip through afq_pug_common.h) over compact arrays of touched
//     vertices, 256 threads and 32 KiB of LDS per cell.
//   * The reference's vertex order (class-major, classes by first appearance, UMIs ascending inside a class) only
//     decides ties between equal-size arborescences, i.e. only inside components of three or more vertices.  It is
//     the order of (smallest record offset of the vertex's class, UMI); the class minima are found for the classes that
//     are asked for - and for every class whose label key is a hash, which is also where equal keys are shown to be
//     equal labels - by one streaming pass of the cell's vertices through an LDS table.
//
// A cell the phase kernels cannot take - a partition over 256 reads (skewed UMIs), a component over 64 vertices or
// over --large-graph-thresh, more pairs than planned - is flagged and resolved by afq_pug.hip's kernel afterwards
// (nothing is approximated); gene-level labels and UMI fields over 4 bytes go there directly.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "afq_common.h"
#include "afq_kernels.h"
#include "afq_prims.h"
#include "afq_pug_common.h"

namespace afq {

typedef unsigned __int128 u128;

constexpr uint32_t kP2Tile = 2048;        // reads per histogram / scatter tile
constexpr uint32_t kP2Bins = 2048;        // partitions per cell the tile kernels rank in LDS
constexpr uint32_t kP2TabSlots = 512;     // hash table of one partition's vertices (<= 256)
constexpr uint32_t kP2FiltBits = 2048;    // presence filter in front of it
constexpr uint32_t kP2PairBuf = 128;      // pairs a wave collects in LDS before it reserves room in the cell's list
constexpr uint32_t kVCntMask = 0x3FFu;    // vertex word: reads (10 bits) | label signature (19 bits) << 10 | key tag << 29
constexpr uint64_t kPairF = 1ull << 63, kPairB = 1ull << 62;   // pair (x, y): x -> y / y -> x is an edge

__device__ __forceinline__ uint32_t sig_of(uint32_t t) { return 1u << (t % 19u); }
__device__ __forceinline__ uint32_t fold9(uint32_t u) { u ^= u >> 18; return (u ^ (u >> 9)) & (kP2TabSlots - 1); }       // linear: fold(a ^ b) = fold(a) ^ fold(b)
__device__ __forceinline__ uint32_t fold11(uint32_t u) { return (u ^ (u >> 11) ^ (u >> 22)) & (kP2FiltBits - 1); }

__device__ __forceinline__ uint32_t wg_add(uint32_t* p, uint32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ uint32_t wg_min(uint32_t* p, uint32_t v) { return __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

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

__device__ __forceinline__ PugCtx make_ctx(const P2Args& A, const P2Cell& c, const CellMeta& m, uint32_t* cnt) {
    PugCtx C;
    C.W = reinterpret_cast<const uint32_t*>(A.bytes + m.chunk_off);
    C.HW = A.hw; C.t2g = A.t2g; C.ref_count = A.ref_count; C.num_genes = A.num_genes;
    C.usa = A.usa; C.num_rows = A.num_rows; C.uo = A.num_rows / 3; C.ao = 2 * (A.num_rows / 3); C.em = A.em;
    C.exact_umi = A.exact_umi; C.large_thresh = A.large_thresh; C.umi_pairs = A.umi_pairs; C.gene_level = 0;
    C.cols = reinterpret_cast<uint32_t*>(A.keys0 + m.key_off);
    C.cols_cap = 2 * m.n_ref + 2;
    C.labw = A.lab ? A.lab + 2 * m.key_off : nullptr;
    C.labd = A.lab ? C.labw + m.n_ref + 1 : nullptr;
    C.lab_cap = m.n_ref + 1;
    C.s_cnt = cnt; C.st = A.st; C.cell = c.cell;
    return C;
}

// ---------------------------------------------------------------------------------------------------------------------------
// 1. reads -> partitions (count, offsets, placement): the exact layout, so that a partition is a dense run of the cell's
//    read slots and its size picks the sort network.
__global__ __launch_bounds__(256) void k_p2_hist(P2Args A) {
    __shared__ uint32_t s_hist[kP2Bins];
    const uint2 td = A.tiles[blockIdx.x];
    const P2Cell c = A.cells[td.x];
    const uint32_t t0 = td.y * kP2Tile, t1 = min(c.R, t0 + kP2Tile);
    const uint64_t* src = A.rd_u + c.rd_base;
    uint32_t* gcnt = A.pcnt + c.part_base;
    const uint32_t P = 1u << c.lgP, pm = P - 1;
    if (P > kP2Bins) {
        for (uint32_t i = t0 + threadIdx.x; i < t1; i += 256) atomicAdd(&gcnt[(uint32_t)(src[i] >> 32) & pm], 1u);
        return;
    }
    for (uint32_t b = threadIdx.x; b < P; b += 256) s_hist[b] = 0;
    __syncthreads();
    for (uint32_t i = t0 + threadIdx.x; i < t1; i += 256) atomicAdd(&s_hist[(uint32_t)(src[i] >> 32) & pm], 1u);
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < P; b += 256) { const uint32_t v = s_hist[b]; if (v) atomicAdd(&gcnt[b], v); }
}

// wave per cell: partition offsets inside the cell, the scatter's cursors, partition -> cell, oversize check
__global__ __launch_bounds__(256) void k_p2_scan(P2Args A) {
    const uint32_t j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = lane_id();
    if (j >= A.n_cells) return;
    const P2Cell c = A.cells[j];
    const uint32_t P = 1u << c.lgP;
    uint32_t carry = 0;
    bool over = false;
    for (uint32_t base = 0; base < P; base += 64) {
        const uint32_t i = base + lane;
        const uint32_t v = i < P ? A.pcnt[c.part_base + i] : 0u;
        uint32_t tot;
        const uint32_t ex = wave_excl_scan(v, tot);
        if (i < P) { A.poff[c.part_base + i] = carry + ex; A.pcur[c.part_base + i] = carry + ex; A.pcell[c.part_base + i] = j; }
        over = over || v > A.part_cap;
        carry += tot;
    }
    const bool bad = A.cell_nkeys[c.cell] != c.R;   // the decode emitted another number of reads than the header announced
    if (__any(over) && lane == 0) A.fb[j] = 1;
    if (lane == 0 && (bad || carry != c.R)) set_err(A.st, kErrRecordWalk, c.cell);
}

__global__ __launch_bounds__(256) void k_p2_scatter(P2Args A) {
    constexpr uint32_t E = kP2Tile / 256;
    __shared__ uint64_t s_a[kP2Tile];
    __shared__ uint64_t s_b[kP2Tile];
    __shared__ uint32_t s_cnt[kP2Bins];
    __shared__ uint32_t s_base[kP2Bins];
    __shared__ uint32_t s_ws[4];
    const uint2 td = A.tiles[blockIdx.x];
    const uint32_t j = td.x;
    if (A.fb[j]) return;
    const P2Cell c = A.cells[j];
    const uint32_t t0 = td.y * kP2Tile, t1 = min(c.R, t0 + kP2Tile);
    const uint64_t* su = A.rd_u + c.rd_base;
    const uint64_t* sh = A.rd_h + c.rd_base;
    uint64_t* du = A.s_u + c.rd_base;
    uint64_t* dh = A.s_h + c.rd_base;
    uint32_t* gcur = A.pcur + c.part_base;
    const uint32_t P = 1u << c.lgP, pm = P - 1;
    if (P > kP2Bins) {   // giant cell: one atomic per read
        for (uint32_t i = t0 + threadIdx.x; i < t1; i += 256) {
            const uint64_t u = su[i];
            const uint32_t pos = atomicAdd(&gcur[(uint32_t)(u >> 32) & pm], 1u);
            du[pos] = u; dh[pos] = sh[i];
        }
        return;
    }
    for (uint32_t b = threadIdx.x; b < P; b += 256) s_cnt[b] = 0;
    __syncthreads();
    uint64_t ku[E], kh[E];
    uint32_t rank[E];
#pragma unroll
    for (uint32_t e = 0; e < E; ++e) {
        const uint32_t i = t0 + e * 256 + threadIdx.x;
        ku[e] = 0; kh[e] = 0; rank[e] = 0;
        if (i < t1) {
            ku[e] = su[i]; kh[e] = sh[i];
            const uint32_t b = (uint32_t)(ku[e] >> 32) & pm;
            rank[e] = (b << 16) | atomicAdd(&s_cnt[b], 1u);
        }
    }
    __syncthreads();
    uint32_t carry = 0;
    for (uint32_t base = 0; base < P; base += 256) {
        const uint32_t b = base + threadIdx.x;
        const uint32_t v = b < P ? s_cnt[b] : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan<256>(v, s_ws, tot);
        if (b < P) { s_cnt[b] = carry + ex; s_base[b] = v ? atomicAdd(&gcur[b], v) : 0u; }
        carry += tot;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t e = 0; e < E; ++e) {
        const uint32_t i = t0 + e * 256 + threadIdx.x;
        if (i < t1) { const uint32_t o = s_cnt[rank[e] >> 16] + (rank[e] & 0xFFFFu); s_a[o] = ku[e]; s_b[o] = kh[e]; }
    }
    __syncthreads();
    const uint32_t nt = t1 - t0;
    for (uint32_t i = threadIdx.x; i < nt; i += 256) {
        const uint64_t u = s_a[i];
        const uint32_t b = (uint32_t)(u >> 32) & pm;
        const uint32_t pos = s_base[b] + (i - s_cnt[b]);
        du[pos] = u; dh[pos] = s_b[i];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// 2. one wave per partition: reads sorted by (label key, UMI, record offset) in registers -> vertices.
//    Vertex i of the partition takes read slot i of the partition: s_h = label key, s_u = umi << 32 | word
//    (word = reads | signature << 10 | key tag << 29), v_off = its smallest record offset; slots past the last vertex get
//    word 0.  Reads under a hashed key (tag 3) are compared with the read before them in the run: equal keys inside a
//    partition are equal labels, the graph kernel closes the chain across partitions.
template <int E>
__device__ __forceinline__ void part_body(const P2Args& A, const uint32_t* W, uint32_t gp, uint32_t n, uint64_t o, uint32_t cell) {
    const uint32_t lane = lane_id();
    uint64_t* sh = A.s_h + o;
    uint64_t* su = A.s_u + o;
    u128 a[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const uint32_t i = (uint32_t)e * 64 + lane;
        a[e] = i < n ? (((u128)sh[i] << 64) | su[i]) : ~(u128)0;
    }
    wave_bitonic_sort<E, u128>(a);
    uint64_t vm[E];      // vertex heads of row e
    bool vh[E];
    bool bad = false;
    uint32_t sig[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const uint32_t i = (uint32_t)e * 64 + lane;
        const uint64_t h = (uint64_t)(a[e] >> 64), uo = (uint64_t)a[e];
        uint64_t hp = __shfl_up(h, 1), up = __shfl_up(uo, 1);
        if (e > 0) {
            const uint64_t h63 = __shfl((uint64_t)(a[e > 0 ? e - 1 : 0] >> 64), 63), u63 = __shfl((uint64_t)a[e > 0 ? e - 1 : 0], 63);
            if (lane == 0) { hp = h63; up = u63; }
        }
        const bool valid = i < n;
        const bool chead = valid && (i == 0 || h != hp);
        vh[e] = valid && (chead || (uint32_t)(uo >> 32) != (uint32_t)(up >> 32));
        vm[e] = __ballot(vh[e]);
        const uint32_t tag = (uint32_t)(h >> 62);
        uint32_t sg = 0;
        if (tag == 1) sg = sig_of((uint32_t)h & 0x7FFFFFFFu);
        else if (tag == 2) sg = sig_of((uint32_t)(h >> 31) & 0x7FFFFFFFu) | sig_of((uint32_t)h & 0x7FFFFFFFu);
        else if (tag == 3 && valid) {
            const uint32_t off = (uint32_t)uo;
            const uint32_t ln = W[off];
            const uint32_t* lp = W + off + A.hw;
            if (vh[e]) for (uint32_t q = 0; q < ln; ++q) sg |= sig_of(lp[q] & 0x7FFFFFFFu);
            if (!chead) {   // same key as the read before: the labels must be the same list
                const uint32_t po = (uint32_t)up;
                const uint32_t* pp = W + po + A.hw;
                bool same = W[po] == ln;
                for (uint32_t q = 0; same && q < ln; ++q) same = ((lp[q] ^ pp[q]) & 0x7FFFFFFFu) == 0;
                bad = bad || !same;
            }
        }
        sig[e] = sg;
    }
    if (__any(bad)) { if (lane == 0) set_err(A.st, kErrLabelHash, cell); return; }
    uint32_t before = 0;   // vertices of the rows below e
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const uint32_t i = (uint32_t)e * 64 + lane;
        if (vh[e]) {
            uint32_t next = n;   // first read of the next vertex
#pragma unroll
            for (int f = E - 1; f >= 0; --f) {
                if (f < e) continue;
                uint64_t mk = vm[f];
                if (f == e) mk = lane == 63 ? 0ull : mk & (~0ull << (lane + 1));
                if (mk) next = (uint32_t)f * 64 + (uint32_t)__builtin_ctzll(mk);
            }
            const uint32_t vi = before + (uint32_t)__popcll(vm[e] & ((1ull << lane) - 1));
            const uint64_t h = (uint64_t)(a[e] >> 64), uo = (uint64_t)a[e];
            sh[vi] = h;
            su[vi] = (uo & 0xFFFFFFFF00000000ull) | (next - i) | (sig[e] << 10) | ((uint32_t)(h >> 62) << 29);
            A.v_off[o + vi] = (uint32_t)uo;
        }
        before += (uint32_t)__popcll(vm[e]);
    }
    for (uint32_t i = before + lane; i < n; i += 64) su[i] = 0;   // (every load of the partition's reads is long done: they went into the sort)
    for (uint32_t i = lane; i < n; i += 64) A.v_flag[o + i] = 0;
    if (lane == 0) A.pnv[gp] = before;
}

__global__ __launch_bounds__(64) void k_p2_part(P2Args A) {
    const uint32_t gp = blockIdx.x;
    const uint32_t j = A.pcell[gp];
    if (A.fb[j]) return;
    const P2Cell c = A.cells[j];
    const uint32_t n = A.pcnt[gp];
    if (n == 0) { if (threadIdx.x == 0) A.pnv[gp] = 0; return; }
    const uint64_t o = c.rd_base + A.poff[gp];
    const uint32_t* W = reinterpret_cast<const uint32_t*>(A.bytes + A.meta[c.cell].chunk_off);
    if (n <= 128) part_body<2>(A, W, gp, n, o, c.cell);
    else part_body<4>(A, W, gp, n, o, c.cell);
}

// ---------------------------------------------------------------------------------------------------------------------------
// 3. one wave per partition: its vertices into an LDS table keyed by UMI, then every UMI that can have a neighbour in
//    the table is looked up - the partition's own vertices (same UMI under another label; every one-base change that
//    leaves the low m bits alone) and the vertices of the partitions one low-bit change away (that change).  A pair of
//    vertices is met once - same UMI from the smaller slot, neighbours from the smaller UMI - and both directions of
//    has_edge (pugutils.rs:76-99) are decided there: x -> y unless reads(y) >= 2 reads(x), always at distance 0, and only
//    if the labels share a ref.
__global__ __launch_bounds__(64) void k_p2_search(P2Args A) {
    __shared__ uint32_t t_umi[kP2TabSlots];
    __shared__ uint32_t t_word[kP2TabSlots];
    __shared__ uint16_t t_idx[kP2TabSlots];
    __shared__ uint32_t s_filt[kP2FiltBits / 32];
    __shared__ uint64_t s_pair[kP2PairBuf];
    __shared__ uint32_t s_np, s_at;
    const uint32_t gp = blockIdx.x, lane = threadIdx.x;
    const uint32_t j = A.pcell[gp];
    if (A.fb[j]) return;
    const P2Cell c = A.cells[j];
    const uint32_t nv = A.pnv[gp];
    if (nv == 0) return;
    const uint32_t m = c.lgP, P = 1u << m, p = gp - c.part_base;
    const uint32_t lo_p = A.poff[gp];                 // the partition's first slot inside the cell
    const uint64_t* ch = A.s_h + c.rd_base;           // the cell's vertex arrays
    const uint64_t* cu = A.s_u + c.rd_base;
    const uint32_t* coff = A.v_off + c.rd_base;
    uint8_t* cflag = A.v_flag + c.rd_base;
    const uint32_t* W = reinterpret_cast<const uint32_t*>(A.bytes + A.meta[c.cell].chunk_off);
    for (uint32_t i = lane; i < kP2TabSlots; i += 64) t_word[i] = 0;
    if (lane < kP2FiltBits / 32) s_filt[lane] = 0;
    if (lane == 0) s_np = 0;
    __syncthreads();
    for (uint32_t i = lane; i < nv; i += 64) {
        const uint64_t uw = cu[lo_p + i];
        const uint32_t umi = (uint32_t)(uw >> 32), word = (uint32_t)uw;
        uint32_t slot = fold9(umi);
        while (atomicCAS(&t_word[slot], 0u, word) != 0u) slot = (slot + 1) & (kP2TabSlots - 1);   // (equal UMIs under different labels: consecutive slots of one run)
        t_umi[slot] = umi;
        t_idx[slot] = (uint16_t)i;
        const uint32_t fb = fold11(umi);
        atomicOr(&s_filt[fb >> 5], 1u << (fb & 31u));
    }
    __syncthreads();
    auto filt = [&](uint32_t u) -> bool { const uint32_t f = fold11(u); return (s_filt[f >> 5] >> (f & 31u)) & 1u; };
    // every vertex of the table with UMI pu against vertex x (cell slot gx, word xw)
    auto probe = [&](uint32_t pu, uint32_t gx, uint32_t xw, bool same) {
        const uint32_t xsig = (xw >> 10) & 0x7FFFFu, cx = xw & kVCntMask;
        for (uint32_t slot = fold9(pu);; slot = (slot + 1) & (kP2TabSlots - 1)) {
            const uint32_t w = t_word[slot];
            if (!w) break;
            if (t_umi[slot] != pu) continue;
            if ((((w >> 10) & 0x7FFFFu) & xsig) == 0) continue;   // no ref in common whatever the UMIs
            const uint32_t gy = lo_p + t_idx[slot];
            uint64_t dir = kPairF | kPairB;
            if (same) { if (gy <= gx) continue; }
            else {
                const uint32_t cy = w & kVCntMask;
                dir = (cy < 2 * cx ? kPairF : 0ull) | (cx < 2 * cy ? kPairB : 0ull);
            }
            const uint64_t hx = ch[gx], hy = ch[gy];
            if (hx != hy && !klab_overlap(klab(W, A.hw, hx, coff[gx]), klab(W, A.hw, hy, coff[gy]))) continue;
            cflag[gx] = 1; cflag[gy] = 1;
            const uint64_t pr = dir | ((uint64_t)gx << 31) | gy;
            const uint32_t at = atomicAdd(&s_np, 1u);
            if (at < kP2PairBuf) s_pair[at] = pr;
            else {   // (a partition with more pairs than the buffer: straight to the cell's list)
                const uint32_t k = atomicAdd(&A.pair_n[j], 1u);
                if (k < c.pair_cap) A.pairs[c.pair_base + k] = pr;
            }
        }
    };
    const uint32_t L = A.umi_pairs;
    // own vertices
    for (uint32_t i = lane; i < nv; i += 64) {
        const uint64_t uw = cu[lo_p + i];
        const uint32_t umi = (uint32_t)(uw >> 32), xw = (uint32_t)uw;
        probe(umi, lo_p + i, xw, true);
        if (A.exact_umi) continue;
        uint64_t pmask = 0;   // bit 3 b + d - 1: that neighbour is larger, stays in the partition and passes the filter
        for (uint32_t b = 0; b < L; ++b) {
#pragma unroll
            for (uint32_t d = 1; d < 4; ++d) {
                const uint32_t mk = d << (2 * b);
                if (mk & (P - 1)) continue;
                const uint32_t pu = umi ^ mk;
                if (pu > umi && filt(pu)) pmask |= 1ull << (3 * b + d - 1);
            }
        }
        while (pmask) {
            const uint32_t ix = (uint32_t)__builtin_ctzll(pmask);
            pmask &= pmask - 1;
            probe(umi ^ ((ix % 3 + 1) << (2 * (ix / 3))), lo_p + i, xw, false);
        }
    }
    // vertices of the partitions one low-bit change away
    if (!A.exact_umi)
        for (uint32_t b = 0; 2 * b < m; ++b) {
            for (uint32_t d = 1; d < 4; ++d) {
                const uint32_t mk = d << (2 * b), low = mk & (P - 1);
                if (!low) continue;
                const uint32_t q = c.part_base + (p ^ low);
                const uint32_t nq = A.pnv[q], oq = A.poff[q];
                for (uint32_t i = lane; i < nq; i += 64) {
                    const uint64_t uw = cu[oq + i];
                    const uint32_t umi = (uint32_t)(uw >> 32), pu = umi ^ mk;
                    if (pu > umi && filt(pu)) probe(pu, oq + i, (uint32_t)uw, false);
                }
            }
        }
    __syncthreads();
    const uint32_t np = s_np;
    if (np) {
        const uint32_t nb = np < kP2PairBuf ? np : kP2PairBuf;
        if (lane == 0) s_at = atomicAdd(&A.pair_n[j], nb);
        __syncthreads();
        const uint32_t at = s_at;
        for (uint32_t i = lane; i < nb; i += 64) if (at + i < c.pair_cap) A.pairs[c.pair_base + at + i] = s_pair[i];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// 4. one wave per partition: a vertex without an edge is a component of its own - one molecule, its label's genes
//    (pugutils.rs:1262-1322).  Columns and two-gene classes are collected per wave and placed with one reservation each.
__global__ __launch_bounds__(64) void k_p2_lone(P2Args A) {
    __shared__ uint32_t s_col[256];
    __shared__ uint32_t s_cls[512];
    __shared__ uint32_t s_at[3];
    const uint32_t gp = blockIdx.x, lane = threadIdx.x;
    const uint32_t j = A.pcell[gp];
    if (A.fb[j]) return;
    const uint32_t nv = A.pnv[gp];
    if (nv == 0) return;
    const P2Cell c = A.cells[j];
    const CellMeta m = A.meta[c.cell];
    uint32_t* gc = A.gcnt + 4 * (size_t)j;
    const PugCtx C = make_ctx(A, c, m, gc);   // (the counters are the cell's global ones here: rare paths add to them directly)
    const uint64_t o = c.rd_base + A.poff[gp];
    uint32_t ncol = 0, ncls = 0;   // wave-uniform
    for (uint32_t i = lane; i - lane < nv; i += 64) {
        uint32_t col = 0xFFFFFFFFu, k0 = 0, k1 = 0;
        bool cls = false;
        if (i < nv && !A.v_flag[o + i]) {
            const uint64_t h = A.s_h[o + i];
            const uint32_t tag = (uint32_t)(h >> 62);
            if (tag == 1 || tag == 2) {
                const uint32_t ga = C.t2g[tag == 1 ? (uint32_t)h & 0x7FFFFFFFu : (uint32_t)(h >> 31) & 0x7FFFFFFFu];
                const uint32_t gb = tag == 2 ? C.t2g[(uint32_t)h & 0x7FFFFFFFu] : ga;
                const uint32_t lo = ga < gb ? ga : gb, hi = ga < gb ? gb : ga;
                col = molecule2_column(C, lo, hi, lo == hi ? 1u : 2u, cls);
                k0 = lo; k1 = hi;
            } else if (tag == 3) {
                const Lab l = rec_label(C, A.v_off[o + i]);
                if (l.n <= 4) {
                    uint32_t g4[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) g4[q] = (uint32_t)q < l.n ? l.p[q] & 0x7FFFFFFFu : 0xFFFFFFFFu;
                    const uint32_t ng = genes_of4(C, g4, l.n);
                    col = molecule4_column(C, g4, ng, cls);
                    k0 = g4[0]; k1 = g4[1];
                } else {
                    uint32_t g[kMaxGenesPerLabel];
                    const uint32_t ng = genes_of(C, l.n, [&](uint32_t j2) { return l.p[j2] & 0x7FFFFFFFu; }, g);
                    if (ng == 0xFFFFFFFFu && C.em) emit_wide_class(C, l.n, [&](uint32_t j2) { return l.p[j2] & 0x7FFFFFFFu; });
                    else col = molecule_column_n(C, g, ng);
                }
            }
        }
        const uint64_t mc = __ballot(col != 0xFFFFFFFFu), mk = __ballot(cls);
        if (col != 0xFFFFFFFFu) s_col[ncol + (uint32_t)__popcll(mc & ((1ull << lane) - 1))] = col;
        if (cls) { const uint32_t r = ncls + (uint32_t)__popcll(mk & ((1ull << lane) - 1)); s_cls[2 * r] = k0; s_cls[2 * r + 1] = k1; }
        ncol += (uint32_t)__popcll(mc); ncls += (uint32_t)__popcll(mk);
    }
    __syncthreads();
    if (lane == 0) {
        s_at[0] = ncol ? atomicAdd(&gc[0], ncol) : 0u;
        s_at[1] = ncls ? atomicAdd(&gc[1], 2 * ncls) : 0u;
        s_at[2] = ncls ? atomicAdd(&gc[2], ncls) : 0u;
    }
    __syncthreads();
    for (uint32_t i = lane; i < ncol; i += 64) { const uint32_t q = s_at[0] + i; if (q >= C.cols_cap) gc[3] = kErrPugLimit; else C.cols[q] = s_col[i]; }
    for (uint32_t i = lane; i < ncls; i += 64) {
        const uint32_t w = s_at[1] + 2 * i, d = s_at[2] + i;
        if (w + 2 > C.lab_cap || 2 * (d + 1) > C.lab_cap) { gc[3] = kErrPugLimit; continue; }
        C.labw[w] = s_cls[2 * i]; C.labw[w + 1] = s_cls[2 * i + 1];
        C.labd[2 * d] = w; C.labd[2 * d + 1] = 2;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// 5. one 256-thread workgroup per cell: the vertices that have an edge.
constexpr int kGNT = 256;
constexpr uint32_t kGLds = 8192;          // words of the phase-shared LDS block (32 KiB)
constexpr uint32_t kGTab = 2048;          // class table slots (keys: 16 KiB, minima: 8 KiB of the block)
constexpr uint32_t kGTabLoad = 1300;      // classes a slice may bring

__global__ __launch_bounds__(kGNT) void k_p2_graph(P2Args A) {
    __shared__ __attribute__((aligned(16))) uint32_t s_big[kGLds];
    __shared__ uint32_t s_ws[kGNT / 64];
    __shared__ uint32_t s_cnt[4];
    __shared__ uint32_t s_flag[4];
    __shared__ unsigned long long s_ebase;
    __shared__ uint32_t s_next;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
  for (;;) {
    __syncthreads();
    if (tid == 0) s_next = atomicAdd(A.work_counter, 1u);
    __syncthreads();
    const uint32_t work = s_next;
    if (work >= A.n_cells) return;
    const uint32_t j = A.order[work];
    const P2Cell c = A.cells[j];
    const CellMeta m = A.meta[c.cell];
    const uint32_t R = c.R;
    auto give_up = [&]() {   // the cell goes to the one-workgroup kernel
        if (tid == 0) { A.fb[j] = 1; A.fb_list[atomicAdd(A.fb_count, 1u)] = c.cell; }
    };
    if (A.fb[j]) { if (tid == 0) A.fb_list[atomicAdd(A.fb_count, 1u)] = c.cell; continue; }
    const uint32_t n_pairs = A.pair_n[j];
    if (n_pairs > c.pair_cap) { give_up(); continue; }
    uint32_t* gc = A.gcnt + 4 * (size_t)j;
    if (tid < 4) { s_cnt[tid] = gc[tid]; s_flag[tid] = 0; }
    __syncthreads();
    PugCtx C = make_ctx(A, c, m, s_cnt);
    const uint64_t* ch = A.s_h + c.rd_base;
    const uint64_t* cu = A.s_u + c.rd_base;
    const uint32_t* coff = A.v_off + c.rd_base;
    const uint8_t* cflag = A.v_flag + c.rd_base;
    uint32_t* lidx = A.lidx + c.rd_base;
    const uint64_t* pairs = A.pairs + c.pair_base;
    // ---- scratch out of the pool: everything is sized by the vertices that have an edge ----
    const uint32_t nt_max = min(R, 2 * n_pairs);
    const unsigned long long need = 26ull * nt_max + 2ull * n_pairs + 64;
    if (tid == 0) s_ebase = atomicAdd(A.pool_cur, need);
    __syncthreads();
    if (s_ebase + need > A.pool_cap) { if (tid == 0) set_err(A.st, kErrPugPool, c.cell); return; }
    uint32_t* q = A.pool + ((s_ebase + 3) & ~3ull);
    uint4* mrec = reinterpret_cast<uint4*>(q); q += 8 * (size_t)nt_max;                 // (16-byte aligned)
    uint64_t* comp_sorted = reinterpret_cast<uint64_t*>(q); q += 2 * (size_t)nt_max;
    uint64_t* okey = reinterpret_cast<uint64_t*>(q); q += 2 * (size_t)nt_max;
    uint32_t* tl = q; q += nt_max;            // touched vertices (cell slots), ascending
    uint32_t* eoff = q; q += nt_max + 2;      // out-degree, then edge offsets
    uint32_t* ecur = q; q += nt_max;
    uint32_t* wlg = q; q += nt_max;           // component labels when they do not fit LDS
    uint32_t* comp_start = q; q += nt_max + 2;
    uint32_t* cidx = q; q += nt_max;          // position of a vertex inside its component (reference order)
    uint32_t* mid_list = q; q += nt_max;      // components of 3..8 vertices first, then 9..64
    uint32_t* mid_off = q; q += nt_max + 2;
    uint32_t* slot_comp = q; q += nt_max;
    uint32_t* cmin = q; q += nt_max;          // smallest record offset of the vertex's class (vertices of listed components)
    uint32_t* pr_list = q; q += nt_max;       // two-vertex components
    uint32_t* edges = q;                      // [2 * n_pairs]
    // ---- 1. the touched vertices, and how many vertices carry a hashed label key ----
    uint32_t NT = 0, n3 = 0;
    for (uint32_t base = 0; base < R; base += kGNT) {
        const uint32_t g = base + tid;
        const bool valid = g < R && (uint32_t)cu[g] != 0;
        const bool t = valid && cflag[g];
        const bool h3 = valid && (ch[g] >> 62) == 3;
        uint32_t tot;
        const uint32_t ex = block_excl_scan<kGNT>((uint32_t)t | ((uint32_t)h3 << 16), s_ws, tot);
        if (t) { const uint32_t li = NT + (ex & 0xFFFFu); if (li < nt_max) { tl[li] = g; lidx[g] = li; } }
        NT += tot & 0xFFFFu; n3 += tot >> 16;
    }
    if (NT > nt_max) { if (tid == 0) set_err(A.st, kErrPugLimit, c.cell); return; }   // (cannot happen: a flag is set with a pair)
    // ---- 2. edges as adjacency lists over local ids ----
    for (uint32_t i = tid; i <= NT; i += kGNT) { eoff[i] = 0; if (i < NT) ecur[i] = 0; }
    __syncthreads();
    for (uint32_t k = tid; k < n_pairs; k += kGNT) {
        const uint64_t pr = pairs[k];
        const uint32_t lx = lidx[(uint32_t)(pr >> 31) & 0x7FFFFFFFu], ly = lidx[(uint32_t)pr & 0x7FFFFFFFu];
        if (pr & kPairF) wg_add(&eoff[lx], 1u);
        if (pr & kPairB) wg_add(&eoff[ly], 1u);
    }
    __syncthreads();
    uint32_t Etot = 0;
    for (uint32_t base = 0; base < NT; base += kGNT) {
        const uint32_t i = base + tid;
        const uint32_t d = i < NT ? eoff[i] : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan<kGNT>(d, s_ws, tot);
        __syncthreads();
        if (i < NT) eoff[i] = Etot + ex;
        Etot += tot;
    }
    if (tid == 0) eoff[NT] = Etot;
    __syncthreads();
    for (uint32_t k = tid; k < n_pairs; k += kGNT) {
        const uint64_t pr = pairs[k];
        const uint32_t lx = lidx[(uint32_t)(pr >> 31) & 0x7FFFFFFFu], ly = lidx[(uint32_t)pr & 0x7FFFFFFFu];
        if (pr & kPairF) edges[eoff[lx] + wg_add(&ecur[lx], 1u)] = ly;
        if (pr & kPairB) edges[eoff[ly] + wg_add(&ecur[ly], 1u)] = lx;
    }
    __syncthreads();
    // ---- 3. weakly connected components: min-label propagation + pointer jumping (labels in LDS when they fit) ----
    uint32_t* wl = NT <= kGLds ? s_big : wlg;
    for (uint32_t i = tid; i < NT; i += kGNT) wl[i] = i;
    __syncthreads();
    for (;;) {
        if (tid == 0) s_flag[0] = 0;
        __syncthreads();
        bool chg = false;
        for (uint32_t i = tid; i < NT; i += kGNT) {
            for (uint32_t e = eoff[i]; e < eoff[i + 1]; ++e) {
                const uint32_t y = edges[e];
                const uint32_t a = wl[i], b = wl[y];
                if (a < b) { wg_min(&wl[y], a); chg = true; }
                else if (b < a) { wg_min(&wl[i], b); chg = true; }
            }
        }
        if (chg) s_flag[0] = 1;
        __syncthreads();
        for (int it = 0; it < 4; ++it) {
            for (uint32_t i = tid; i < NT; i += kGNT) { const uint32_t l = wl[i]; const uint32_t ll = wl[l]; if (ll < l) wl[i] = ll; }
            __syncthreads();
        }
        if (!s_flag[0]) break;
    }
    for (uint32_t i = tid; i < NT; i += kGNT) { uint32_t l = wl[i]; while (wl[l] != l) l = wl[l]; comp_sorted[i] = ((uint64_t)l << 20) | i; }
    __syncthreads();
    tiled_bitonic_sort_by<kGNT, kGLds / 2>(comp_sorted, NT, [](uint64_t a, uint64_t b) { return a > b; }, reinterpret_cast<uint64_t*>(s_big));
    uint32_t NC = 0;
    for (uint32_t base = 0; base < NT; base += kGNT) {
        const uint32_t i = base + tid;
        const bool hd = i < NT && (i == 0 || (comp_sorted[i] >> 20) != (comp_sorted[i - 1] >> 20));
        uint32_t tot;
        const uint32_t ex = block_excl_scan<kGNT>(hd, s_ws, tot);
        if (hd) comp_start[NC + ex] = i;
        NC += tot;
    }
    if (tid == 0) comp_start[NC] = NT;
    __syncthreads();
    // ---- 4. the components by size: pairs, 3..8, 9..64; anything else is not for this kernel ----
    {
        uint32_t tiny = 0;
        bool big = false;
        for (uint32_t k = tid; k < NC; k += kGNT) {
            const uint32_t n = comp_start[k + 1] - comp_start[k];
            big = big || n > 64 || n > C.large_thresh;
            tiny += n >= 3 && n <= 8;
        }
        if (big) s_flag[1] = 1;
        if (tiny) atomicAdd(&s_flag[2], tiny);
    }
    __syncthreads();
    if (s_flag[1]) { give_up(); continue; }
    const uint32_t n_tiny = s_flag[2];
    __syncthreads();
    if (tid < 4) s_flag[tid] = 0;
    __syncthreads();
    for (uint32_t k = tid; k < NC; k += kGNT) {
        const uint32_t n = comp_start[k + 1] - comp_start[k];
        if (n == 2) pr_list[atomicAdd(&s_flag[2], 1u)] = k;
        else if (n <= 8) mid_list[atomicAdd(&s_flag[0], 1u)] = k;
        else mid_list[n_tiny + atomicAdd(&s_flag[1], 1u)] = k;
    }
    __syncthreads();
    const uint32_t n_mid = n_tiny + s_flag[1], n_pr = s_flag[2];
    uint32_t S_mid = 0;
    for (uint32_t base = 0; base < n_mid; base += kGNT) {
        const uint32_t ci = base + tid;
        const uint32_t n = ci < n_mid ? comp_start[mid_list[ci] + 1] - comp_start[mid_list[ci]] : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan<kGNT>(n, s_ws, tot);
        if (ci < n_mid) mid_off[ci] = S_mid + ex;
        S_mid += tot;
    }
    if (tid == 0) mid_off[n_mid] = S_mid;
    __syncthreads();
    for (uint32_t ci = tid; ci < n_mid; ci += kGNT) {
        const uint32_t b0 = mid_off[ci], n = mid_off[ci + 1] - b0;
        for (uint32_t i = 0; i < n; ++i) slot_comp[b0 + i] = ci;
    }
    __syncthreads();
    auto vid_at = [&](uint32_t i) -> uint32_t { return (uint32_t)comp_sorted[i] & 0xFFFFFu; };   // local id
    // ---- 5. class minima: for the classes of the listed components' vertices (their order decides ties) and for every
    //         class under a hashed key (equal keys must be equal labels).  The cell's vertices stream through an LDS table
    //         keyed by label key, a slice of the key space at a time when the classes outnumber it. ----
    if (n3 || S_mid) {
        unsigned long long* t_key = reinterpret_cast<unsigned long long*>(s_big);
        uint32_t* t_min = s_big + 2 * kGTab;
        const uint32_t n_slices = (n3 + S_mid + kGTabLoad - 1) / kGTabLoad;
        auto mix = [](uint64_t h) -> uint32_t { return ((uint32_t)h ^ (uint32_t)(h >> 32)) * 0x9E3779B1u; };
        auto slice_of = [&](uint32_t mx) -> uint32_t { return n_slices == 1 ? 0u : (mx >> 11) % n_slices; };
        auto find = [&](uint64_t h, uint32_t mx, bool insert) -> uint32_t {   // slot of h, or kGTab
            uint32_t slot = mx & (kGTab - 1);
            for (uint32_t step = 0; step < kGTab; ++step, slot = (slot + 1) & (kGTab - 1)) {
                unsigned long long k = t_key[slot];
                if (k == h) return slot;
                if (k == ~0ull) {
                    if (!insert) return kGTab;
                    k = atomicCAS(&t_key[slot], ~0ull, (unsigned long long)h);
                    if (k == ~0ull || k == h) return slot;
                }
            }
            return kGTab;
        };
        for (uint32_t sl = 0; sl < n_slices; ++sl) {
            __syncthreads();
            for (uint32_t i = tid; i < kGTab; i += kGNT) { t_key[i] = ~0ull; t_min[i] = 0xFFFFFFFFu; }
            __syncthreads();
            for (uint32_t s = tid; s < S_mid; s += kGNT) {   // the classes that are asked for
                const uint32_t ci = slot_comp[s];
                const uint64_t h = ch[tl[vid_at(comp_start[mid_list[ci]] + (s - mid_off[ci]))]];
                const uint32_t mx = mix(h);
                if (slice_of(mx) == sl && find(h, mx, true) == kGTab) s_flag[3] = 1;
            }
            __syncthreads();
            for (uint32_t g = tid; g < R; g += kGNT) {
                if ((uint32_t)cu[g] == 0) continue;
                const uint64_t h = ch[g];
                const uint32_t mx = mix(h);
                if (slice_of(mx) != sl) continue;
                const uint32_t slot = find(h, mx, (h >> 62) == 3);
                if (slot != kGTab) atomicMin(&t_min[slot], coff[g]);
                else if ((h >> 62) == 3) s_flag[3] = 1;
            }
            __syncthreads();
            if (s_flag[3]) break;
            if (n3)
                for (uint32_t g = tid; g < R; g += kGNT) {   // every vertex under a hashed key against its class's first record
                    if ((uint32_t)cu[g] == 0) continue;
                    const uint64_t h = ch[g];
                    if ((h >> 62) != 3) continue;
                    const uint32_t mx = mix(h);
                    if (slice_of(mx) != sl) continue;
                    const uint32_t rep = t_min[find(h, mx, false)], off = coff[g];
                    if (rep != off && !lab_equal(rec_label(C, off), rec_label(C, rep))) s_cnt[3] = kErrLabelHash;
                }
            for (uint32_t s = tid; s < S_mid; s += kGNT) {
                const uint32_t ci = slot_comp[s];
                const uint32_t li = vid_at(comp_start[mid_list[ci]] + (s - mid_off[ci]));
                const uint64_t h = ch[tl[li]];
                const uint32_t mx = mix(h);
                if (slice_of(mx) == sl) cmin[li] = t_min[find(h, mx, false)];
            }
        }
        __syncthreads();
        if (s_flag[3]) { give_up(); continue; }   // (a slice with more classes than the table: cannot happen with the slice count above)
        if (s_cnt[3]) { if (tid == 0) set_err(A.st, s_cnt[3], c.cell); return; }
    }
    __syncthreads();
    // ---- 6. two-vertex components: one molecule, the refs both labels share (pugutils.rs:1161-1188) ----
    for (uint32_t k = tid; k - lane < n_pr; k += kGNT) {   // (wave-uniform trip count: append_cols is a wave-wide call)
        uint32_t col = 0xFFFFFFFFu, k0 = 0, k1 = 0;
        bool cls = false;
        if (k < n_pr) {
            const uint32_t i0 = comp_start[pr_list[k]];
            const uint32_t ga = tl[vid_at(i0)], gb = tl[vid_at(i0 + 1)];
            const KLab l = klab(C.W, C.HW, ch[ga], coff[ga]), l2 = klab(C.W, C.HW, ch[gb], coff[gb]);
            if (l.n <= 4) {
                uint32_t g4[4];
                uint32_t kk = 0;
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    g4[qq] = 0xFFFFFFFFu;
                    if ((uint32_t)qq < l.n) {
                        const uint32_t t = klab_ref(l, qq);
                        if (klab_contains(l2, t)) {
#pragma unroll
                            for (int w = 0; w < 4; ++w) if ((uint32_t)w == kk) g4[w] = t;
                            ++kk;
                        }
                    }
                }
                const uint32_t ng = genes_of4(C, g4, kk);
                col = molecule4_column(C, g4, ng, cls);
                k0 = g4[0]; k1 = g4[1];
            } else {
                uint32_t g[kMaxGenesPerLabel];
                uint32_t ng = 0;
                for (uint32_t jj = 0; jj < l.n && ng != 0xFFFFFFFFu; ++jj) {
                    const uint32_t t = l.p[jj] & 0x7FFFFFFFu;
                    if (!klab_contains(l2, t)) continue;
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
                    emit_wide_class(C, l.n, [&](uint32_t jj) -> uint32_t { const uint32_t t = l.p[jj] & 0x7FFFFFFFu; return klab_contains(l2, t) ? t : 0xFFFFFFFFu; });
                else emit_molecule(C, g, ng);
            }
        }
        append_cols(C, col);
        append_class2(C, cls, k0, k1);
    }
    // ---- 7. components of 3..64 vertices: their vertices in the reference's order, gathered into the covers' records ----
    for (uint32_t s = tid; s < S_mid; s += kGNT) {   // order key of every listed vertex; rank inside its component
        const uint32_t ci = slot_comp[s];
        const uint32_t li = vid_at(comp_start[mid_list[ci]] + (s - mid_off[ci]));
        okey[s] = ((uint64_t)cmin[li] << 32) | (uint32_t)(cu[tl[li]] >> 32);
    }
    __syncthreads();
    for (uint32_t s = tid; s < S_mid; s += kGNT) {
        const uint32_t ci = slot_comp[s];
        const uint32_t b0 = mid_off[ci], n = mid_off[ci + 1] - b0;
        const uint64_t mine = okey[s];
        uint32_t rank = 0;
        for (uint32_t i = 0; i < n; ++i) rank += okey[b0 + i] < mine;
        cidx[vid_at(comp_start[mid_list[ci]] + (s - b0))] = rank;
    }
    __syncthreads();
    for (uint32_t s = tid; s < S_mid; s += kGNT) {
        const uint32_t ci = slot_comp[s];
        const uint32_t b0 = mid_off[ci];
        const uint32_t li = vid_at(comp_start[mid_list[ci]] + (s - b0));
        const uint32_t g = tl[li];
        const KLab l = klab(C.W, C.HW, ch[g], coff[g]);
        uint32_t r0 = 0xFFFFFFFFu, r1 = 0xFFFFFFFFu, r2 = 0xFFFFFFFFu, r3 = 0xFFFFFFFFu;
        if (l.n <= 4) {
            if (l.n > 0) r0 = klab_ref(l, 0);
            if (l.n > 1) r1 = klab_ref(l, 1);
            if (l.n > 2) r2 = l.p[2] & 0x7FFFFFFFu;
            if (l.n > 3) r3 = l.p[3] & 0x7FFFFFFFu;
        } else { const uint64_t pa = (uint64_t)(uintptr_t)l.p; r0 = (uint32_t)pa; r1 = (uint32_t)(pa >> 32); }
        uint64_t adj = 0;
        for (uint32_t e = eoff[li]; e < eoff[li + 1]; ++e) adj |= 1ull << cidx[edges[e]];
        const size_t at = (size_t)b0 + cidx[li];
        mrec[2 * at] = make_uint4(g, l.n, r0, r1);
        mrec[2 * at + 1] = make_uint4(r2, r3, (uint32_t)adj, (uint32_t)(adj >> 32));
    }
    __syncthreads();
    cover_tiny8<kGNT / 64>(C, mrec, mid_off, n_tiny, wv, lane);
    cover_wave64<kGNT / 64>(C, mrec, mid_off, n_tiny, n_mid, wv, lane);
    __syncthreads();
    if (s_cnt[3]) { if (tid == 0) set_err(A.st, s_cnt[3], c.cell); return; }
    if (tid == 0) {
        A.cell_ncols[c.cell] = s_cnt[0];
        if (A.lab_cnt) { A.lab_cnt[2 * c.cell] = s_cnt[1]; A.lab_cnt[2 * c.cell + 1] = s_cnt[2]; }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
void launch_p2_split(hipStream_t s, const P2Args& a) {
    if (!a.n_cells) return;
    AFQ_LAUNCH(k_p2_hist, a.n_tiles, 256, s, a);
    AFQ_LAUNCH(k_p2_scan, (a.n_cells + 3) / 4, 256, s, a);
    AFQ_LAUNCH(k_p2_scatter, a.n_tiles, 256, s, a);
}
void launch_p2_part(hipStream_t s, const P2Args& a) { if (a.n_parts) AFQ_LAUNCH(k_p2_part, a.n_parts, 64, s, a); }
void launch_p2_search(hipStream_t s, const P2Args& a) { if (a.n_parts) AFQ_LAUNCH(k_p2_search, a.n_parts, 64, s, a); }
void launch_p2_lone(hipStream_t s, const P2Args& a) { if (a.n_parts) AFQ_LAUNCH(k_p2_lone, a.n_parts, 64, s, a); }
void launch_p2_graph(hipStream_t s, const P2Args& a) {
    if (!a.n_cells) return;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    const uint32_t nb = a.n_cells < 4u * (uint32_t)cus ? a.n_cells : 4u * (uint32_t)cus;
    AFQ_LAUNCH(k_p2_graph, nb, kGNT, s, a);
}

}  // namespace afq
