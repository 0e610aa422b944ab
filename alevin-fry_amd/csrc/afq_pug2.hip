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
//   * Only the vertices that HAVE an edge (a quarter) reach the graph phase: components, the two-vertex rule and the
//     arborescence covers (shared with afq_pug.hip through afq_pug_common.h).  Since round 6 that phase is the range-wide
//     kernels of afq_pugflat.hip; the per-cell kernels below (k_p2_graph, k_p2_cover, k_p2_tied<., false>) take the cells it
//     routes to them - a component of more than 64 vertices - and k_p2_tied<., true> finds the class minima of the
//     components the flat covers set aside (AFQ_TEST_P2_GRAPH=cell: the per-cell kernels for every cell, as in rounds 3-5).
//   * The reference's vertex order (class-major, classes by first appearance, UMIs ascending inside a class) only
//     decides ties between equal-size arborescences, i.e. only inside components of three or more vertices.  It is
//     the order of (smallest record offset of the vertex's class, UMI); the class minima are found for the classes that
//     are asked for - the labels of the vertices in such components - by one streaming pass of the cell's vertices
//     through a hash table (LDS, or the pool for big cells), which is also where equal HASHED keys of an asked-for class
//     are shown to be equal labels.  Nothing else takes two vertices for one class: the partition kernel compares the
//     reads it merges into a vertex, the search compares labels under hashed keys by content.
//
// A cell the phase kernels cannot take - a partition over 256 reads (skewed UMIs), a component over 64 vertices or
// over --large-graph-thresh, more pairs than planned - is flagged and resolved by afq_pug.hip's kernel afterwards
// (nothing is approximated); gene-level labels and UMI fields over 4 bytes go there directly.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#include "afq_common.h"
#include "afq_kernels.h"
#include "afq_prims.h"
#include "afq_pug_common.h"
#include "afq_p2_shared.h"

namespace afq {


// The partition kernels' walk (a wave per partition, persistent workgroups): the partitions go to the XCDs - workgroups x, x + 8,
// ... share one, they are dealt round-robin - in runs of 1024, a few cells each, 256 workgroups side by side on a run.  A cell's
// partitions then meet in ONE L2: the search reads a partition's dozen neighbours, the partition and lone-vertex kernels read the
// labels of the cell's chunk.  (With a cell's partitions spread over all eight XCDs the search's FETCH_SIZE was 36 GB per
// configs[2] step, eight times the vertices; in runs 19.9 GB.)  The grid is a multiple of 2048.
template <typename F>
__device__ __forceinline__ void for_each_partition_in_runs(uint32_t n_parts, uint32_t wv, F&& f) {
    const uint32_t r = blockIdx.x / 8, x = blockIdx.x % 8, n_runs = (n_parts + 1023) / 1024;
    for (uint32_t run = (r / 256) * 8 + x; run < n_runs; run += gridDim.x / 256) {
        const uint32_t gp = run * 1024 + (r % 256) * 4 + wv;
        if (gp < n_parts) f(gp);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// 1. reads -> partitions (count, offsets, placement): the exact layout, so that a partition is a dense run of the cell's
//    read slots and its size picks the sort network.
__global__ __launch_bounds__(256) void k_p2_hist(P2Args A) {
    __shared__ uint32_t s_hist[kP2Bins];
    const uint2 td = A.tiles[blockIdx.x];
    const P2Cell c = A.cells[td.x];
    const uint32_t t0 = td.y * A.tile, t1 = min(c.R, t0 + A.tile);
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

// wave per cell: partition offsets inside the cell, the scatter's cursors, partition -> cell, oversize check.  A cell with a
// partition over the capacity is flagged and its partition counts are zeroed: the partition kernels then pass over it.
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
    if (__any(over)) {
        if (lane == 0) A.fb[j] = 1;
        for (uint32_t i = lane; i < P; i += 64) A.pcnt[c.part_base + i] = 0;
    }
    if (lane == 0 && (bad || carry != c.R)) set_err(A.st, kErrRecordWalk, c.cell);
}

// The tile goes through LDS partition-major, so that a partition's run of the tile leaves in consecutive lanes of one store
// (written straight from each read's own registers, its run's other reads by other waves at other times, was "direct" in rounds
// 3-4).  Measured on configs[2] (395 M reads per step, 6.3 GB of payload each way): direct, 2048-read tiles: 7.9 ms and
// WRITE_SIZE 20.9 GB per step - the runs were 64 bytes, written eight bytes at a time through an L2 that holds a few MB of the
// lines all its workgroups have open; direct with 8192-read tiles: 6.4 ms, still 19.5 GB; staged with 4096-read tiles (80 KiB of
// LDS, two workgroups to a CU): 5.1 ms and 7.0 GB.  NT threads take a tile of 8 NT reads.
template <int NT>
__global__ __launch_bounds__(NT) void k_p2_scatter(P2Args A) {
    constexpr uint32_t E = 8, kP2Tile = E * NT;
    static_assert(kP2Tile == kP2TileHost, "the host cuts the cells into tiles of kP2TileHost reads");
    __shared__ uint64_t s_a[kP2Tile];
    __shared__ uint64_t s_b[kP2Tile];
    __shared__ uint32_t s_cnt[kP2Bins];
    __shared__ uint32_t s_base[kP2Bins];
    __shared__ uint32_t s_ws[NT / 64];
    // Workgroups go to the eight XCDs round-robin, each XCD with an L2 of its own.  A tile leaves a run of about eight reads per
    // partition - 64 bytes of each array, seldom on a line boundary - and the next run of that partition comes from the cell's next
    // tile: with tile = blockIdx the two halves of a line were written through two L2s and reached memory as two partial writes
    // (WRITE_SIZE 20.9 GB per step for 6.3 GB of payload).  So the tiles go to the XCDs in runs of sixteen (a 30 000-read cell):
    // a cell's tiles pass through one L2 one behind the other and their runs meet there.
    const uint32_t r = blockIdx.x / 8, tile = ((r / 16) * 8 + blockIdx.x % 8) * 16 + r % 16;   // (the grid is a multiple of 128)
    if (tile >= A.n_tiles) return;
    const uint2 td = A.tiles[tile];
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
        for (uint32_t i = t0 + threadIdx.x; i < t1; i += NT) {
            const uint64_t u = su[i];
            const uint32_t pos = atomicAdd(&gcur[(uint32_t)(u >> 32) & pm], 1u);
            du[pos] = u; dh[pos] = sh[i];
        }
        return;
    }
    for (uint32_t b = threadIdx.x; b < P; b += NT) s_cnt[b] = 0;
    __syncthreads();
    uint64_t ku[E], kh[E];
    uint32_t rank[E];
#pragma unroll
    for (uint32_t e = 0; e < E; ++e) {
        const uint32_t i = t0 + e * NT + threadIdx.x;
        ku[e] = 0; kh[e] = 0; rank[e] = 0;
        if (i < t1) {
            ku[e] = su[i]; kh[e] = sh[i];
            const uint32_t b = (uint32_t)(ku[e] >> 32) & pm;
            rank[e] = (b << 16) | atomicAdd(&s_cnt[b], 1u);
        }
    }
    __syncthreads();
    uint32_t carry = 0;
    for (uint32_t base = 0; base < P; base += NT) {
        const uint32_t b = base + threadIdx.x;
        const uint32_t v = b < P ? s_cnt[b] : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan<NT>(v, s_ws, tot);
        if (b < P) { s_cnt[b] = carry + ex; s_base[b] = v ? atomicAdd(&gcur[b], v) : 0u; }
        carry += tot;
    }
    __syncthreads();
    {
#pragma unroll
        for (uint32_t e = 0; e < E; ++e) {
            const uint32_t i = t0 + e * NT + threadIdx.x;
            if (i < t1) { const uint32_t o = s_cnt[rank[e] >> 16] + (rank[e] & 0xFFFFu); s_a[o] = ku[e]; s_b[o] = kh[e]; }
        }
        __syncthreads();
        const uint32_t nt = t1 - t0;
        for (uint32_t i = threadIdx.x; i < nt; i += NT) {
            const uint64_t u = s_a[i];
            const uint32_t b = (uint32_t)(u >> 32) & pm;
            const uint32_t pos = s_base[b] + (i - s_cnt[b]);
            du[pos] = u; dh[pos] = s_b[i];
        }
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
    if (__any(bad)) {   // two labels under one key: the range is decoded again under another hash (afq_api.cpp); the kernels behind this one
        if (lane == 0) { set_err(A.st, kErrLabelHash, cell); A.pnv[gp] = 0; A.pn3[gp] = 0; }   // find an empty partition
        return;
    }
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
            A.lidx[o + vi] = (uint32_t)(uo >> 32);   // the vertex's UMI on its own: what the search of the neighbouring partitions fetches (the graph kernel reuses the array afterwards)
            // the flag byte: bit 0 "has an edge" (the search sets it), above it the vertex's reads, 127 = "127 or more: see s_u" - the
            // range-wide graph build reads the flags anyway and gets the read counts of the vertices it numbers with them
            A.v_flag[o + vi] = (uint8_t)(min(next - i, 127u) << 1);
        }
        before += (uint32_t)__popcll(vm[e]);
    }
    for (uint32_t i = before + lane; i < n; i += 64) { su[i] = 0; sh[i] = 0; A.v_flag[o + i] = 0; }   // (every load of the partition's reads is long done: they went into the sort)
    uint32_t n3 = 0;   // vertices under a hashed key (they sort last: the key's tag is its top bits)
#pragma unroll
    for (int e = 0; e < E; ++e) n3 += (uint32_t)__popcll(__ballot(vh[e] && (uint32_t)((uint64_t)(a[e] >> 64) >> 62) == 3));
    if (lane == 0) { A.pnv[gp] = before; A.pn3[gp] = n3; }   // (pn3: no kernel reads it any more - the class step used to walk every hashed vertex)
}

__global__ __launch_bounds__(256) void k_p2_part(P2Args A) {
    // (the wave's number through readfirstlane: the compiler then knows the partition index is uniform, and everything read per
    //  partition - its counts, its place, its cell's record - comes through scalar loads into scalar registers)
    for_each_partition_in_runs(A.n_parts, __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), [&](uint32_t gp) {
        const uint32_t n = A.pcnt[gp];
        if (n == 0) { if (lane_id() == 0) A.pnv[gp] = 0; return; }
        const P2Cell c = A.cells[A.pcell[gp]];
        const uint64_t o = c.rd_base + A.poff[gp];
        const uint32_t* W = reinterpret_cast<const uint32_t*>(A.bytes + c.chunk_off);
        if (n <= 128) part_body<2>(A, W, gp, n, o, c.cell);
        else part_body<4>(A, W, gp, n, o, c.cell);
    });
}

// ---------------------------------------------------------------------------------------------------------------------------
// 3. one wave per partition: its vertices into an LDS table keyed by UMI, then every UMI that can have a neighbour in
//    the table is looked up - the partition's own vertices (same UMI under another label; every one-base change that
//    leaves the low m bits alone) and the vertices of the partitions one low-bit change away (that change).  A pair of
//    vertices is met once - same UMI from the smaller slot, neighbours from the smaller UMI - and both directions of
//    has_edge (pugutils.rs:76-99) are decided there: x -> y unless reads(y) >= 2 reads(x), always at distance 0, and only
//    if the labels share a ref.
// (Four partitions to a 256-thread workgroup, a wave each with its own slice of LDS and nothing shared: no workgroup barrier;
// LDS instructions of one wave execute in order, WAVE_SYNC only keeps the compiler from moving code across.)

#ifndef AFQ_SEARCH_FB
#define AFQ_SEARCH_FB 4   // foreign partitions fetched together
#endif
#ifndef AFQ_P2_SEARCH_WGS
#define AFQ_P2_SEARCH_WGS 7   // workgroups per CU the search is compiled for: 72 VGPRs (five spilled), 22.5 KiB of LDS with the 2^12-bit filter.  Measured per configs[2] step: 4 -> 33.5 ms, 5 -> 28.0, 6 -> 25.5 (24.9 with a 2^13-bit filter, which no longer fits seven times), 7 -> 23.7
#endif
struct SearchLds {
    uint32_t umi[kP2TabSlots];
    uint32_t word[kP2TabSlots];
    uint8_t idx[kP2TabSlots];           // (a partition holds at most 256 vertices)
    uint32_t np;
};
static_assert(kP2PartCap <= 256, "SearchLds::idx is a byte");
// OVER: the second pass over the partitions that found more pairs than they have slots of their own - k vertices of one UMI whose
// labels overlap are k (k - 1) / 2 pairs (reads of one molecule that hit different members of a gene family), so a partition of n
// reads can hold up to n^2 / 2.  The first pass left the count in pnp: the list gets that many slots out of the pool, the
// partition's first own slot says where (the graph kernel tells by pnp > pcnt), and the search runs again.  (Until round 4 such a
// cell went to the one-workgroup kernel: on the label-tail workload that kernel was 40 ms of the 370 ms step.  A kernel of its
// own, not a second trip through a loop here: the loop took the search from 79 to 127 VGPRs.)
#ifdef AFQ_SEARCH_TIMING
__device__ unsigned long long g_search_t[10];
#define S_MARK(i) do { const unsigned long long n_ = clock64(); if (lane == 0) atomicAdd(&g_search_t[i], n_ - st_); st_ = n_; } while (0)
__global__ void k_search_timing_dump() {
    unsigned long long tot = 0;
    for (int i = 0; i < 8; ++i) tot += g_search_t[i];
    printf("search wave cycles: setup+build %.1f%% same-umi %.1f%% own filter %.1f%% own drain %.1f%% foreign fetch+filter %.1f%% foreign drain %.1f%% (unused) %.1f%% tail %.1f%% (partitions %llu, %.0f cycles each)\n",
           100.0 * g_search_t[0] / tot, 100.0 * g_search_t[1] / tot, 100.0 * g_search_t[2] / tot, 100.0 * g_search_t[3] / tot, 100.0 * g_search_t[4] / tot,
           100.0 * g_search_t[5] / tot, 100.0 * g_search_t[6] / tot, 100.0 * g_search_t[7] / tot, g_search_t[8], (double)tot / (double)g_search_t[8]);
    for (int i = 0; i < 10; ++i) g_search_t[i] = 0;
}
#else
#define S_MARK(i) do {} while (0)
#endif
template <bool OVER>
__device__ __forceinline__ void search_body(const P2Args& A, uint32_t gp, SearchLds& S, uint32_t* filt_all, uint32_t wv, uint32_t lane) {
    // filt_all: the four waves' presence filters, one behind the other in an array aligned to its own size - a filter word's byte
    // offset is then (wave's offset | word's offset), and the word of fold(umi) ^ fold(change) is at offset(umi) ^ offset(change):
    // ONE exclusive-or per probe, the array's address in the instruction's offset field
    uint32_t* t_umi = S.umi; uint32_t* t_word = S.word; uint8_t* t_idx = S.idx;
    uint32_t* s_filt = filt_all + wv * (kP2FiltBits / 32); uint32_t* s_np = &S.np;
#ifdef AFQ_SEARCH_TIMING
    unsigned long long st_ = clock64();
#endif
    const uint32_t nv = A.pnv[gp];
    if (nv == 0) return;
    const uint32_t j = A.pcell[gp];
    const P2Cell c = A.cells[j];
    const uint32_t m = c.lgP, P = 1u << m, p = gp - c.part_base;
    const uint32_t lo_p = A.poff[gp];                 // the partition's first slot inside the cell
    uint32_t pcap = A.pcnt[gp];                       // ... and how many it has: the partition's pairs go into its own slots of the pair array
    if (OVER && A.pnp[gp] <= pcap) return;
    const uint64_t* cu = A.s_u + c.rd_base;           // the cell's vertices: UMI << 32 | word
    uint64_t* ppair = A.pairs + c.rd_base + lo_p;
    if (OVER) {
        const uint32_t np = A.pnp[gp];
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(A.pool_cur, 2ull * np + 2);
        base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
        if (base + 2ull * np + 2 > A.pool_cap) { if (lane == 0) { set_err(A.st, kErrPugPool, c.cell); A.pnp[gp] = 0; } return; }   // (the host runs the cell again with a larger pool)
        base = (base + 1) & ~1ull;
        if (lane == 0) ppair[0] = base;
        ppair = reinterpret_cast<uint64_t*>(A.pool + base);
        pcap = np;
    }
    // the foreign partitions' places go out first: their latency hides under the table build
    const uint32_t nfor = A.exact_umi ? 0u : 3 * ((m + 1) / 2);
    uint32_t f_n = 0, f_o = 0;
    if (lane < nfor) {
        const uint32_t low = ((lane % 3 + 1) << (2 * (lane / 3))) & (P - 1);
        if (low) { const uint32_t q = c.part_base + (p ^ low); f_n = A.pnv[q]; f_o = A.poff[q]; }
    }
    for (uint32_t i = lane; i < kP2TabSlots; i += 64) t_word[i] = 0;
    for (uint32_t i = lane; i < kP2FiltBits / 32; i += 64) s_filt[i] = 0;
    if (lane == 0) *s_np = 0;
    WAVE_SYNC();
    uint64_t own[4];   // the partition's own vertices stay in registers (<= 256 of them)
#pragma unroll
    for (int r = 0; r < 4; ++r) own[r] = (uint32_t)r * 64 + lane < nv ? cu[lo_p + r * 64 + lane] : 0ull;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t i = (uint32_t)r * 64 + lane;
        if (i >= nv) continue;
        const uint32_t umi = (uint32_t)(own[r] >> 32), word = (uint32_t)own[r];
        uint32_t slot = fold9(umi);
        while (atomicCAS(&t_word[slot], 0u, word) != 0u) slot = (slot + 1) & (kP2TabSlots - 1);   // (equal UMIs under different labels: consecutive slots of one run)
        t_umi[slot] = umi;
        t_idx[slot] = (uint8_t)i;
        const uint32_t fb = fold11(umi);
        atomicOr(&s_filt[fb >> 5], 1u << (fb & 31u));
    }
    WAVE_SYNC();
    S_MARK(0);
    auto filt = [&](uint32_t u) -> bool { const uint32_t f = fold11(u); return (s_filt[f >> 5] >> (f & 31u)) & 1u; };
    // Every vertex of the table with UMI pu against vertex x (cell slot gx, word xw).  A MATCH - x against the table's vertex y in
    // `slot`: UMIs as asked, signatures with a ref in common - is written down as a CANDIDATE pair with both directions of has_edge
    // decided (the read counts are at hand).  Whether the two labels really share a ref is a chain of dependent global loads (label
    // keys, record offsets, for hashed keys the ref lists): until late round 6 that chain ran here - first inside the table walk
    // that found the match, one or two lanes of the wave still in it (two thirds of the kernel's wave cycles), then, rounds 5-6,
    // parked and checked a lane each at the partition's end (still 38 % of them: per-phase clocks, AFQ_SEARCH_TIMING) - and is now
    // k_p2_check's (afq_pugflat.hip), a thread per candidate over the whole range, which has no chain to wait for: it clears the
    // candidates that fail and flags the end points of the others.  (The search flagging the end points of its candidates itself -
    // byte stores of known values, no atomics in k_p2_check - was measured: a vertex that loses all its candidates then goes through
    // the graph phase as a component of one vertex, correctly, but there are enough of them to cost that phase 2.2 ms per configs[2]
    // step for 0.8 saved here.)
    auto probe = [&](uint32_t pu, uint32_t gx, uint32_t xw, bool same) {
        const uint32_t xsig = (xw >> 10) & 0x7FFFFu, cx = xw & kVCntMask;
        for (uint32_t slot = fold9(pu);; slot = (slot + 1) & (kP2TabSlots - 1)) {
            const uint32_t w = t_word[slot];
            if (!w) break;
            if (t_umi[slot] != pu) continue;
            if ((((w >> 10) & 0x7FFFFu) & xsig) == 0) continue;   // no ref in common whatever the UMIs
            const uint32_t gy = lo_p + t_idx[slot], cy = w & kVCntMask;
            if (same && gy <= gx) continue;                       // (a same-UMI pair is met once, from the smaller slot)
            const uint64_t dir = same ? (kPairF | kPairB) : ((cy < 2 * cx ? kPairF : 0ull) | (cx < 2 * cy ? kPairB : 0ull));
            const uint32_t at = atomicAdd(s_np, 1u);   // (LDS, this wave's own counter)
            if (at < pcap) ppair[at] = dir | ((uint64_t)gx << 31) | gy;
        }
    };
    const uint32_t L = A.umi_pairs;
    // own vertices: same UMI under another label, and every one-base change that stays in the partition.  The filter checks are
    // branch-free - a probe that passes (one in thirty, nearly always a false positive) sets its bit in a per-lane mask, and the
    // masks are drained together afterwards: a handful of table walks per wave instead of a divergent one at nearly every step,
    // and no exec-mask bookkeeping per probe (the kernel was as busy on its scalar unit as on its vector units).  The fold of the
    // filter is linear, so a probe's filter bit is fold(umi) ^ fold(change): one XOR per probe, the change's fold a scalar.
    // "met once, from the smaller UMI": umi ^ (d << 2b) > umi iff the top bit of d is clear in the base - a bit test.
    // The passed probes of row r: bit 2 b of ha[r] - base b's change 1; bit 2 b + 1 of ha[r] - its change 2; bit 2 b + 1 of hb[r] - its
    // change 3: each at the bit of the UMI that decides whether the change makes the UMI larger, so "from the smaller UMI" is one AND
    // with ~umi per row behind the loop instead of a test per probe.  The changes outside, the rows inside: a change's fold, its
    // offset and its bit are scalar work.  Per change and row: two exclusive-ors, the filter word, a bit-field extract, a
    // shift-or (until late round 6: eleven vector instructions and, with the rows outside, as many scalar ones).
    uint32_t ha[4] = {0, 0, 0, 0}, hb[4] = {0, 0, 0, 0};
    const uint32_t b_lo = m / 2;
    uint32_t fu[4], fo[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { fu[r] = fold11((uint32_t)(own[r] >> 32)); fo[r] = (wv * (kP2FiltBits / 8)) | ((fu[r] >> 5) << 2); }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t i = (uint32_t)r * 64 + lane;
        if ((uint32_t)r * 64 >= nv) break;   // (uniform)
        if (i < nv) probe((uint32_t)(own[r] >> 32), lo_p + i, (uint32_t)own[r], true);
    }
    S_MARK(1);
    if (!A.exact_umi) {
        const char* fbase = reinterpret_cast<const char*>(filt_all);
        for (uint32_t b = b_lo; b < L; ++b) {   // (bases below m / 2 lie inside the low m bits: every change there leaves the partition)
#pragma unroll
            for (uint32_t d = 1; d < 4; ++d) {
                const uint32_t mk = d << (2 * b);
                if (mk & (P - 1)) continue;   // (scalar: only the base that straddles bit m)
                const uint32_t fm = fold11(mk), fmo = (fm >> 5) << 2, tb = 2 * b + (d == 1 ? 0u : 1u);
#pragma unroll
                for (int r = 0; r < 4; ++r) {   // (measured and not kept: the rows' count as a compile-time constant, four instances of this loop - 19.7 against 17.2 ms, 14 VGPRs spilled instead of 8; all four rows whether the partition has them or not, their filter words asked for together - 16.96 against 16.73)
                    if ((uint32_t)r * 64 >= nv) break;   // (uniform)
                    const uint32_t w = *reinterpret_cast<const uint32_t*>(fbase + (fo[r] ^ fmo));
                    const uint32_t bit = (w >> ((fu[r] ^ fm) & 31u)) & 1u;
                    if (d == 3) hb[r] |= bit << tb; else ha[r] |= bit << tb;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t keep = (uint32_t)r * 64 + lane < nv ? ~(uint32_t)(own[r] >> 32) : 0u;
            ha[r] &= keep; hb[r] &= keep;
        }
    }
    S_MARK(2);
    for (;;) {   // drain: every lane takes its next passed probe, whichever row it is in (static register indices: no scratch)
        const bool mine = (ha[0] | ha[1] | ha[2] | ha[3] | hb[0] | hb[1] | hb[2] | hb[3]) != 0;
        if (!__any(mine)) break;
        if (mine) {
            uint32_t r = 0, hw = 0, three = 0;
            uint64_t ow = 0;
            if (ha[0]) { r = 0; hw = ha[0]; ow = own[0]; ha[0] &= ha[0] - 1; }
            else if (ha[1]) { r = 1; hw = ha[1]; ow = own[1]; ha[1] &= ha[1] - 1; }
            else if (ha[2]) { r = 2; hw = ha[2]; ow = own[2]; ha[2] &= ha[2] - 1; }
            else if (ha[3]) { r = 3; hw = ha[3]; ow = own[3]; ha[3] &= ha[3] - 1; }
            else if (hb[0]) { r = 0; hw = hb[0]; ow = own[0]; three = 1; hb[0] &= hb[0] - 1; }
            else if (hb[1]) { r = 1; hw = hb[1]; ow = own[1]; three = 1; hb[1] &= hb[1] - 1; }
            else if (hb[2]) { r = 2; hw = hb[2]; ow = own[2]; three = 1; hb[2] &= hb[2] - 1; }
            else { r = 3; hw = hb[3]; ow = own[3]; three = 1; hb[3] &= hb[3] - 1; }
            const uint32_t pos = (uint32_t)__builtin_ctz(hw);
            const uint32_t d = three ? 3u : 1u + (pos & 1u);
            probe((uint32_t)(ow >> 32) ^ (d << (pos & ~1u)), lo_p + r * 64 + lane, (uint32_t)ow, false);
        }
    }
    S_MARK(3);
    // The vertices of the partitions one low-bit change away, FOUR partitions at a time (their places are in lanes k .. k + 3's
    // registers): the eight loads of a batch go out together and are waited for once.  A partition at a time with the next one's
    // load behind it, every load's latency was in the open - a partition takes some thirty instructions to go through the filter -
    // twelve round trips one after the other in a kernel whose wave spends 49 us on a partition, nine tenths of it waiting.
    const uint32_t* cl = A.lidx + c.rd_base;   // UMIs only (k_p2_part): four bytes per foreign vertex instead of eight - this fetch is 12 of the 13 partitions a search reads
    auto fetch = [&](uint32_t k, uint32_t (&v)[2], uint32_t& nq, uint32_t& oq) {
        nq = __builtin_amdgcn_readlane(f_n, k); oq = __builtin_amdgcn_readlane(f_o, k);
        v[0] = lane < nq ? cl[oq + lane] : 0u;
        v[1] = lane + 64 < nq ? cl[oq + 64 + lane] : 0u;
    };
    // (their filter checks are branch-free too: a passed probe is a bit (partition k, row r) in a per-lane mask; the drain fetches
    //  that vertex again - it is in the cache - instead of carrying a queue of (probe, slot, word) triples in registers)
    uint64_t fhits = 0;   // bit 2 k + r (k < 24: three changes of at most eight low bases)
    for (uint32_t k0 = 0; k0 < nfor; k0 += AFQ_SEARCH_FB) {
        uint32_t fv[AFQ_SEARCH_FB][2], fnq[AFQ_SEARCH_FB], foq[AFQ_SEARCH_FB];
#pragma unroll
        for (int j = 0; j < AFQ_SEARCH_FB; ++j) {
            fnq[j] = 0; foq[j] = 0; fv[j][0] = 0; fv[j][1] = 0;
            if (k0 + (uint32_t)j < nfor) fetch(k0 + (uint32_t)j, fv[j], fnq[j], foq[j]);
        }
#pragma unroll
        for (int j = 0; j < AFQ_SEARCH_FB; ++j) {
            const uint32_t k = k0 + (uint32_t)j;
            if (k >= nfor) break;   // (uniform)
            const uint32_t nq = fnq[j], oq = foq[j];
            const uint32_t mk = (k % 3 + 1) << (2 * (k / 3));
            const uint32_t fm = fold11(mk), tb = 2 * (k / 3) + (k % 3 == 0 ? 0u : 1u);   // (scalar) the change's fold; the bit of the base that decides "umi ^ mk > umi"
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const uint32_t i = (uint32_t)r * 64 + lane;
                const uint32_t umi = fv[j][r];
                const uint32_t f = fold11(umi) ^ fm;
                const uint32_t ok = (i < nq ? 1u : 0u) & (((umi >> tb) & 1u) ^ 1u) & ((s_filt[f >> 5] >> (f & 31u)) & 1u);
                fhits |= (uint64_t)ok << (2 * k + (uint32_t)r);
            }
            for (uint32_t i = 128 + lane; i < nq; i += 64) {   // (a partition of more than 128 vertices: the rest, plainly)
                const uint32_t umi = cl[oq + i], pu = umi ^ mk;
                if (pu > umi && filt(pu)) probe(pu, oq + i, (uint32_t)cu[oq + i], false);
            }
        }
    }
    S_MARK(4);
    for (; __any(fhits != 0);) {
        const uint32_t ix = fhits ? (uint32_t)__builtin_ctzll(fhits) : 0u;
        const uint32_t k = ix >> 1, r = ix & 1u;
        const uint32_t base = (uint32_t)__shfl((int)f_o, (int)k);   // (every lane takes part in the shuffle: a lane that sat out would hand its neighbour a zero)
        if (fhits) {
            fhits &= fhits - 1;
            const uint32_t gx = base + r * 64 + lane;
            const uint64_t uw = cu[gx];
            probe((uint32_t)(uw >> 32) ^ ((k % 3 + 1) << (2 * (k / 3))), gx, (uint32_t)uw, false);
        }
    }
    WAVE_SYNC();
    S_MARK(5);
    S_MARK(6);
    const uint32_t np = *s_np;
    if (lane == 0 && !OVER) A.pnp[gp] = np;   // (more than pcap: the second pass takes the partition)
    if (lane == 0 && OVER && np != pcap) set_err(A.st, kErrInternal, c.cell);   // (the same search twice: the same pairs)
    WAVE_SYNC();
    S_MARK(7);
#ifdef AFQ_SEARCH_TIMING
    if (lane == 0) atomicAdd(&g_search_t[8], 1ull);
#endif
}
__global__ __launch_bounds__(256, AFQ_P2_SEARCH_WGS) void k_p2_search(P2Args A) {
    if (A.st->err_code) return;   // an earlier kernel of the range failed (e.g. kErrLabelHash in k_p2_part, which then leaves its partition's vertices unwritten): nothing behind it may read that state - the host runs the range again or reports the error
    __shared__ SearchLds s_lds[4];
    __shared__ __attribute__((aligned(kP2FiltBits / 8))) uint32_t s_filt4[4 * (kP2FiltBits / 32)];
    const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    for_each_partition_in_runs(A.n_parts, wv, [&](uint32_t gp) { search_body<false>(A, gp, s_lds[wv], s_filt4, wv, lane); });
}
__global__ __launch_bounds__(256) void k_p2_search_over(P2Args A) {
    if (A.st->err_code) return;
    __shared__ SearchLds s_lds[4];
    __shared__ __attribute__((aligned(kP2FiltBits / 8))) uint32_t s_filt4[4 * (kP2FiltBits / 32)];
    const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    for (uint32_t g0 = (blockIdx.x * 4 + wv) * 64; g0 < A.n_parts; g0 += gridDim.x * 256) {   // 64 partitions' counts at a time, a lane each
        const uint32_t gp = g0 + lane;
        uint64_t over = __ballot(gp < A.n_parts && A.pnp[gp] > A.pcnt[gp]);
        for (; over; over &= over - 1) search_body<true>(A, g0 + (uint32_t)__builtin_ctzll(over), s_lds[wv], s_filt4, wv, lane);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// 4. one wave per partition: a vertex without an edge is a component of its own - one molecule, its label's genes
//    (pugutils.rs:1262-1322).  A partition writes its columns into the cell's column list at its own read slots (a lone
//    vertex of slot i puts its column - or the "no column" filler, which the per-cell histogram passes over - at entry i)
//    and its two-gene classes for the EM into its slots of a staging array: no reservation, no atomic (every wave of a
//    cell adding to the cell's counters was the same-address queue this kernel spent its time in).
// The column of ONE lone vertex whose label has 5..64 refs, by its whole wave: lane j looks up the gene of ref j, the wave sorts
// the genes in its registers, the distinct ones are the heads of their runs - genes_of and molecule_column_n (afq_pug_common.h:
// quant.rs:974-1024 -> utils.rs:688-753 / em.rs:499-514) without the 64-entry array a single lane keeps in scratch memory
// (every load and store there a trip to global memory, an insertion sort of them per molecule: on labels of E[na] = 3 with a
// geometric tail that loop was four fifths of this kernel, 28.7 against 5.9 ms per step).  Returns the column (wave-uniform) or
// 0xFFFFFFFF when the molecule has none - dropped, or kept as a gene-level class for the EM, which is written here.
__device__ __forceinline__ uint32_t wave_label_column(const PugCtx& c, uint32_t rec_dw, uint32_t n, uint32_t lane) {
    const uint32_t* lp = c.W + rec_dw + c.HW;
    uint32_t g[1];
    g[0] = lane < n ? c.t2g[lp[lane] & 0x7FFFFFFFu] : 0xFFFFFFFFu;
    wave_bitonic_sort<1, uint32_t>(g);   // ascending, the padding behind the genes
    const uint32_t gs = g[0];
    const uint32_t prev = __shfl_up(gs, 1);
    const bool head = gs != 0xFFFFFFFFu && (lane == 0 || gs != prev);
    const uint64_t hm = __ballot(head);
    const uint32_t ng = (uint32_t)__popcll(hm);
    if (ng == 0) return 0xFFFFFFFFu;
    const uint32_t b0 = (uint32_t)__builtin_ctzll(hm);
    const uint64_t hm1 = hm & (hm - 1);
    const uint32_t g0 = (uint32_t)__shfl((int)gs, (int)b0);
    const uint32_t g1 = hm1 ? (uint32_t)__shfl((int)gs, (int)__builtin_ctzll(hm1)) : 0xFFFFFFFFu;
    auto sua = [&](uint32_t x) { return (x & 1u) == 0 ? (x >> 1) : c.uo + (x >> 1); };
    uint32_t col = 0xFFFFFFFFu;
    if (c.em) {
        if (ng == 1) col = !c.usa ? g0 : sua(g0);
        else if (c.usa && ng == 2 && ((g0 ^ g1) & ~1u) == 0) col = c.ao + (g0 >> 1);
        else {   // a gene-level class: its genes ascending, one word per distinct gene
            uint32_t off = 0, di = 0;
            if (lane == 0) { off = atomicAdd(&c.s_cnt[1], ng); di = atomicAdd(&c.s_cnt[2], 1u); }
            off = (uint32_t)__builtin_amdgcn_readfirstlane((int)off); di = (uint32_t)__builtin_amdgcn_readfirstlane((int)di);
            if (off + ng > c.lab_cap || 2 * (di + 1) > c.lab_cap) { if (lane == 0) c.s_cnt[3] = kErrPugLimit; return 0xFFFFFFFFu; }
            if (head) c.labw[off + (uint32_t)__popcll(hm & ((1ull << lane) - 1))] = gs;
            if (lane == 0) { c.labd[2 * di] = off; c.labd[2 * di + 1] = ng; }
            return 0xFFFFFFFFu;
        }
    } else if (!c.usa) {
        if (ng == 1) col = g0;
    } else if (ng == 1) {
        col = sua(g0);
    } else if (ng == 2) {
        const bool s1 = (g0 & 1u) == 0, s2 = (g1 & 1u) == 0;
        if (((g0 ^ g1) & ~1u) == 0) col = c.ao + (g0 >> 1);
        else if (s1 && !s2) col = g0 >> 1;
        else if (!s1 && s2) col = g1 >> 1;
    } else if (ng <= 10) {   // exactly one spliced gene among them: it, or - with its unspliced sibling right behind it - ambiguous
        const uint64_t sm = __ballot(head && (gs & 1u) == 0);
        if (__popcll(sm) == 1) {
            const uint32_t ls = (uint32_t)__builtin_ctzll(sm);
            const uint32_t sg = (uint32_t)__shfl((int)gs, (int)ls);
            const uint64_t nm = ls == 63 ? 0ull : hm & ~((2ull << ls) - 1);
            const uint32_t gn = nm ? (uint32_t)__shfl((int)gs, (int)__builtin_ctzll(nm)) : 0xFFFFFFFFu;
            col = (nm && ((sg ^ gn) & ~1u) == 0) ? c.ao + (sg >> 1) : (sg >> 1);
        }
    }
    if (col == 0xFFFFFFFFu) return col;
    if (col >= c.num_rows) { if (lane == 0) c.s_cnt[3] = kErrSlotRange; return 0xFFFFFFFFu; }
    return col;
}

// The column of a lone vertex whose label has 5..8 refs, by its own lane with everything in registers: three in four of the labels
// over four refs on the tail model, and a wave per vertex (wave_label_column) is two dependent round trips to memory per vertex, one
// vertex after the other - the lanes of a row make theirs together.  g: the genes of the label's refs, 0xFFFFFFFF past its end, in any
// order.  A 19-exchange network sorts them, repeats become padding, a second pass moves the padding behind the distinct genes;
// then the rules of molecule_column_n (afq_pug_common.h) with every index a constant.
__device__ __forceinline__ void sort8(uint32_t (&a)[8]) {
#define AFQ_CE(i, j) { const uint32_t lo_ = min(a[i], a[j]), hi_ = max(a[i], a[j]); a[i] = lo_; a[j] = hi_; }
    AFQ_CE(0, 1) AFQ_CE(2, 3) AFQ_CE(4, 5) AFQ_CE(6, 7)
    AFQ_CE(0, 2) AFQ_CE(1, 3) AFQ_CE(4, 6) AFQ_CE(5, 7)
    AFQ_CE(1, 2) AFQ_CE(5, 6) AFQ_CE(0, 4) AFQ_CE(3, 7)
    AFQ_CE(1, 5) AFQ_CE(2, 6)
    AFQ_CE(1, 4) AFQ_CE(3, 6)
    AFQ_CE(2, 4) AFQ_CE(3, 5)
    AFQ_CE(3, 4)
#undef AFQ_CE
}
__device__ __forceinline__ uint32_t molecule8_column(const PugCtx& c, uint32_t (&g)[8]) {
    constexpr uint32_t kNo = 0xFFFFFFFFu;
    sort8(g);
    uint32_t d[8];
    d[0] = g[0];
#pragma unroll
    for (int q = 1; q < 8; ++q) d[q] = g[q] == g[q - 1] ? kNo : g[q];
    sort8(d);
    uint32_t ng = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) ng += d[q] != kNo ? 1u : 0u;
    if (ng == 0) return kNo;
    const uint32_t g0 = d[0], g1 = d[1];
    auto sua = [&](uint32_t x) { return (x & 1u) == 0 ? (x >> 1) : c.uo + (x >> 1); };
    uint32_t col = kNo;
    if (c.em) {
        if (ng == 1) col = !c.usa ? g0 : sua(g0);
        else if (c.usa && ng == 2 && ((g0 ^ g1) & ~1u) == 0) col = c.ao + (g0 >> 1);
        else {   // a gene-level class for the EM: its genes ascending
            const uint32_t off = atomicAdd(&c.s_cnt[1], ng), di = atomicAdd(&c.s_cnt[2], 1u);
            if (off + ng > c.lab_cap || 2 * (di + 1) > c.lab_cap) { c.s_cnt[3] = kErrPugLimit; return kNo; }
#pragma unroll
            for (int q = 0; q < 8; ++q) if ((uint32_t)q < ng) c.labw[off + q] = d[q];
            c.labd[2 * di] = off; c.labd[2 * di + 1] = ng;
            return kNo;
        }
    } else if (!c.usa) {
        if (ng == 1) col = g0;
    } else if (ng == 1) {
        col = sua(g0);
    } else if (ng == 2) {
        const bool s1 = (g0 & 1u) == 0, s2 = (g1 & 1u) == 0;
        if (((g0 ^ g1) & ~1u) == 0) col = c.ao + (g0 >> 1);
        else if (s1 && !s2) col = g0 >> 1;
        else if (!s1 && s2) col = g1 >> 1;
    } else {   // 3..8 genes (the rule holds up to ten): exactly one spliced gene among them - it, or with its unspliced sibling right behind it: ambiguous
        uint32_t nsp = 0, sg = 0, nx = kNo;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if ((uint32_t)q < ng && (d[q] & 1u) == 0) { if (nsp == 0) { sg = d[q]; nx = q + 1 < 8 ? d[q + 1 < 8 ? q + 1 : 7] : kNo; } ++nsp; }
        if (nsp == 1) col = (nx != kNo && ((sg ^ nx) & ~1u) == 0) ? c.ao + (sg >> 1) : (sg >> 1);
    }
    if (col == kNo) return col;
    if (col >= c.num_rows) { c.s_cnt[3] = kErrSlotRange; return kNo; }
    return col;
}

#ifdef AFQ_LONE_TIMING
__device__ unsigned long long g_lone_t[10];
#define L_MARK(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long n_ = clock64(); if (lane == 0) atomicAdd(&g_lone_t[i], n_ - lt_); lt_ = n_; } while (0)
__global__ void k_lone_timing_dump() {
    unsigned long long tot = 0;
    for (int i = 0; i < 7; ++i) tot += g_lone_t[i];
    printf("lone wave cycles: head (scalar chain) %.1f%% keys+flags+offsets %.1f%% label words %.1f%% genes %.1f%% rules+stores %.1f%% coop labels %.1f%% classes %.1f%% (partitions %llu)\n",
           100.0 * g_lone_t[0] / tot, 100.0 * g_lone_t[1] / tot, 100.0 * g_lone_t[2] / tot, 100.0 * g_lone_t[3] / tot, 100.0 * g_lone_t[4] / tot, 100.0 * g_lone_t[5] / tot, 100.0 * g_lone_t[6] / tot, g_lone_t[8]);
    for (int i = 0; i < 10; ++i) g_lone_t[i] = 0;
}
#else
#define L_MARK(i) do {} while (0)
#endif
// L8: labels of 5..8 refs by their own lane (molecule8_column; AFQ_TEST_P2_LONE_COOP=2) - an instance of its own: its eight-entry arrays
// are registers of every lane whether or not a label needs them.
#ifndef AFQ_LONE_ROWS
#define AFQ_LONE_ROWS 2
#endif
// FLAT (late round 6): the rows are any run of a cell's slots, not a partition's - a slot past its partition's vertices holds a zero
// key (k_p2_part), which is "no label", so a slot needs no partition to be resolved; only a staged two-gene class (em) does, and
// the lane that has one works its partition out of the vertex's UMI and takes the next place in that partition's stage.
template <bool L8, bool FLAT>
__device__ __forceinline__ void lone_rows(const P2Args& A, const P2Cell& c, uint32_t j, uint32_t lo_p, uint32_t n, uint32_t nv, uint32_t gp, uint32_t* s_cls, uint32_t* s_g, uint32_t lane) {
#ifdef AFQ_LONE_TIMING
    unsigned long long lt_ = clock64();
#endif
    uint32_t* gc = A.gcnt + 4 * (size_t)j;
    const PugCtx C = make_ctx(A, c, gc);   // (the counters are the cell's global ones here: the rare class writes add to them directly)
    const uint64_t o = c.rd_base + lo_p;
    uint32_t ncls = 0;   // wave-uniform
    // Two rows of 64 slots at a time, every level of the gather chain issued for both rows before anything waits: key / flag /
    // record offset, then (hashed keys only) the label's length and first four refs out of the chunk, then the genes of all
    // refs - five dependent trips per partition where a hashed label used to add four of its own per row it occurred in.
    const uint32_t cw = 2 + c.R * C.HW + c.n_ref;   // dwords of the chunk
    PugCtx Cg = C;
    Cg.gene_level = 1;   // (genes_of4 is handed gene ids below: the gathers are done here, for all slots together)
    L_MARK(0);
    constexpr int RW = FLAT ? AFQ_LONE_ROWS : 2;   // rows of 64 slots in flight per trip
    for (uint32_t r0 = 0; r0 < n; r0 += 64 * RW) {   // (uniform)
        constexpr int NR = L8 ? 8 : 4;   // refs of a label a lane looks at itself
        uint64_t h2[RW];
        uint32_t fl[RW], of[RW], ln[RW], t4[RW][NR], g4[RW][NR];
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const uint32_t i = r0 + (uint32_t)r * 64 + lane;
            h2[r] = i < nv ? A.s_h[o + i] : 0ull;
            fl[r] = i < nv ? (A.v_flag[o + i] & 1u) : 1u;
            of[r] = i < nv ? A.v_off[o + i] : 0u;
        }
        L_MARK(1);
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const uint32_t tag = fl[r] ? 0u : (uint32_t)(h2[r] >> 62);
            ln[r] = tag == 3 ? C.W[of[r]] : tag;   // (tags 1 and 2 are the label's length)
#pragma unroll
            for (int q = 0; q < NR; ++q) t4[r][q] = 0xFFFFFFFFu;
            if (tag == 1) t4[r][0] = (uint32_t)h2[r] & 0x7FFFFFFFu;
            else if (tag == 2) { t4[r][0] = (uint32_t)(h2[r] >> 31) & 0x7FFFFFFFu; t4[r][1] = (uint32_t)h2[r] & 0x7FFFFFFFu; }
            else if (tag == 3) {
                const uint32_t* lp = C.W + of[r] + C.HW;   // (a hashed label has three refs or more; the fourth dword read may be the next record's first - never one past the chunk)
#pragma unroll
                for (int q = 0; q < 4; ++q) t4[r][q] = (q < 3 || of[r] + C.HW + 3 < cw) ? lp[q] & 0x7FFFFFFFu : 0xFFFFFFFFu;
                if constexpr (L8) {   // (the label's length is not here yet: the next four dwords, if the chunk has them, whatever they are)
#pragma unroll
                    for (int q = 4; q < 8; ++q) t4[r][q] = of[r] + C.HW + (uint32_t)q < cw ? lp[q] & 0x7FFFFFFFu : 0xFFFFFFFFu;
                }
            }
        }
        L_MARK(2);
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int q = 0; q < NR; ++q) g4[r][q] = (uint32_t)q < ln[r] && ln[r] <= (uint32_t)NR ? C.t2g[t4[r][q]] : 0xFFFFFFFFu;
        L_MARK(3);
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const uint32_t i = r0 + (uint32_t)r * 64 + lane;
            if (r0 + (uint32_t)r * 64 >= n) break;   // (uniform)
            uint32_t col = 0xFFFFFFFFu, k0 = 0, k1 = 0;
            bool cls = false;
            if (ln[r] != 0 && ln[r] <= 4) {
                uint32_t q4[4] = {g4[r][0], g4[r][1], g4[r][2], g4[r][3]};
                const uint32_t ng = genes_of4(Cg, q4, ln[r]);
                col = molecule4_column(C, q4, ng, cls);
                k0 = q4[0]; k1 = q4[1];
            } else if (L8 && ln[r] <= 8 && ln[r] != 0) {
                if constexpr (L8) col = molecule8_column(C, g4[r]);
            }
            if (A.lone_coop) {   // labels of 5..64 refs: the wave takes them one after the other (wave_label_column)
                for (uint64_t lm = __ballot(ln[r] > (uint32_t)NR && ln[r] <= 64); lm; lm &= lm - 1) {
                    const uint32_t b = (uint32_t)__builtin_ctzll(lm);
                    const uint32_t cb = wave_label_column(C, (uint32_t)__shfl((int)of[r], (int)b), (uint32_t)__shfl((int)ln[r], (int)b), lane);
                    if (lane == b) col = cb;
                }
            }
            if (i < n) C.cols[lo_p + i] = col;
            if constexpr (FLAT) {
                if (cls) {   // (one vertex in a dozen: its partition from its UMI's low bits, the next place of that partition's stage)
                    const uint32_t gq = c.part_base + ((uint32_t)(A.s_u[o + i] >> 32) & ((1u << c.lgP) - 1u));
                    const uint32_t k = atomicAdd(&A.pncls[gq], 1u);
                    A.cstage[c.rd_base + A.poff[gq] + k] = ((uint64_t)k1 << 32) | k0;
                }
            } else {
                const uint64_t mk = __ballot(cls);
                if (cls) { const uint32_t q = ncls + (uint32_t)__popcll(mk & ((1ull << lane) - 1)); s_cls[2 * q] = k0; s_cls[2 * q + 1] = k1; }
                ncls += (uint32_t)__popcll(mk);
            }
        }
        L_MARK(4);
        // labels of more than 64 refs (without the cooperative path: of more than four): one lane after the other, its genes in the
        // wave's LDS row - 64 words of a lane's own were 272 bytes of scratch per lane of every wave, for a label in ten thousand.
        // (Behind the rows' loop, where the gathered refs and genes are dead: inside it the kernel lost its seventh wave per SIMD.)
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            if (r0 + (uint32_t)r * 64 >= n) break;   // (uniform)
            for (uint64_t wm = __ballot(!(L8 && ln[r] <= 8) && ln[r] > (A.lone_coop ? 64u : 4u)); wm; wm &= wm - 1) {
                if (lane == (uint32_t)__builtin_ctzll(wm)) {
                    const Lab l = rec_label(C, of[r]);
                    const uint32_t ng = genes_of(C, l.n, [&](uint32_t j2) { return l.p[j2] & 0x7FFFFFFFu; }, s_g);
                    if (ng == 0xFFFFFFFFu && C.em) emit_wide_class(C, l.n, [&](uint32_t j2) { return l.p[j2] & 0x7FFFFFFFu; });
                    else C.cols[lo_p + r0 + (uint32_t)r * 64 + lane] = molecule_column_n(C, s_g, ng);
                }
                WAVE_SYNC();
            }
        }
    }
    L_MARK(5);
#ifdef AFQ_LONE_TIMING
    if (lane == 0) atomicAdd(&g_lone_t[8], 1ull);
#endif
    if (FLAT || !ncls) return;
    WAVE_SYNC();
    uint64_t* stage = A.cstage + o;   // (at most one class per vertex: the partition's own slots hold them; the graph kernel moves them into the cell's label area)
    for (uint32_t i = lane; i < ncls; i += 64) stage[i] = ((uint64_t)s_cls[2 * i + 1] << 32) | s_cls[2 * i];
    if (lane == 0) A.pncls[gp] = ncls;
    WAVE_SYNC();
    L_MARK(6);
}
template <bool L8>
__device__ __forceinline__ void lone_body(const P2Args& A, uint32_t gp, uint32_t* s_cls, uint32_t* s_g, uint32_t lane) {
    const uint32_t n = A.pcnt[gp], nv = A.pnv[gp], j = A.pcell[gp], lo_p = A.poff[gp];
    if (n == 0) return;
    const P2Cell c = A.cells[j];
    lone_rows<L8, false>(A, c, j, lo_p, n, nv, gp, s_cls, s_g, lane);
}
#ifndef AFQ_LONE_WPE
#define AFQ_LONE_WPE 7   // waves per SIMD k_p2_lone<false> is compiled for
#endif
template <bool L8>
__global__ __launch_bounds__(256, L8 ? 5 : AFQ_LONE_WPE) void k_p2_lone(P2Args A) {
    if (A.st->err_code) return;   // an earlier kernel of the range failed (e.g. kErrLabelHash in k_p2_part, which then leaves its partition's vertices unwritten): nothing behind it may read that state - the host runs the range again or reports the error
    __shared__ uint32_t s_cls4[4][512];
    __shared__ uint32_t s_g4[4][kMaxGenesPerLabel];   // (a wave's row for the genes of a label of more than 64 refs)
    const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    for_each_partition_in_runs(A.n_parts, wv, [&](uint32_t gp) { lone_body<L8>(A, gp, s_cls4[wv], s_g4[wv], lane); });
}
// ... and over the range's tiles of 4096 slots, a wave per 256 slots of one: no partition's head (three dependent scalar loads in front
// of everything), no half-empty rows at a partition's end, no staging of classes through LDS.
#ifndef AFQ_LONE_FLAT
#define AFQ_LONE_FLAT 1
#endif
template <bool L8>
__global__ __launch_bounds__(256, L8 ? 5 : AFQ_LONE_WPE) void k_pl_lone(P2Args A) {
    if (A.st->err_code) return;
    __shared__ uint32_t s_g4[4][kMaxGenesPerLabel];
    const uint2 td = A.tiles[blockIdx.x];
    const uint32_t j = td.x;
    if (A.fb[j]) return;   // (handed back by k_p2_scan: the one-workgroup kernel's)
    const P2Cell c = A.cells[j];
    const uint32_t t0 = td.y * A.tile, t1 = min(c.R, t0 + A.tile);
    const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    for (uint32_t lo = t0 + wv * 256; lo < t1; lo += 1024) { const uint32_t n = min(256u, t1 - lo); lone_rows<L8, true>(A, c, j, lo, n, n, 0u, nullptr, s_g4[wv], lane); }
}

// ---------------------------------------------------------------------------------------------------------------------------
// 5. one workgroup per cell: the vertices that have an edge (about a quarter of them) - components, the two-vertex rule,
//    the covers.  Everything is sized by the pairs the search left; nothing here walks the cell's reads, and the one pass over
//    the cell's vertices (class minima, below) reads eight bytes per vertex.
#ifdef AFQ_PUG_TIMING
#define G_MARK(i) do { __syncthreads(); if (threadIdx.x == 0) tmark[i] = wall_clock64(); } while (0)
#else
#define G_MARK(i) do {} while (0)
#endif
constexpr int kGNT = 256;
#ifndef AFQ_GTAB
#define AFQ_GTAB 4096
#endif
#ifndef AFQ_GRAPH_WPE
#define AFQ_GRAPH_WPE 3
#endif
constexpr uint32_t kGTab = AFQ_GTAB;          // class table slots when it lives in LDS (keys: 32 KiB, minima: 16 KiB of the block)
// (it takes 3/4 of its slots less a margin in classes; beyond that the table is carved out of the pool)
constexpr uint32_t kCatPair = 1, kCatTiny = 2, kCatMid = 3, kCatBig = 4, kCatLarge = 5;   // (Large: above --large-graph-thresh, resolved winner-take-all)

template <int GNT>
__global__ __launch_bounds__(GNT) void k_p2_graph(P2Args A, const uint32_t* list, const uint32_t* n_list, uint32_t work_lo, uint32_t work_hi, uint32_t* counter) {
    // list: the cells to take, entries work_lo .. work_hi - 1 (n_list: the list's length lives on the device - the cells the range-wide
    // build of afq_pugflat.hip routed here because one of their components has more than 64 vertices)
    if (n_list) work_hi = *n_list;
    if (A.st->err_code) return;   // an earlier kernel of the range failed (e.g. kErrLabelHash in k_p2_part, which then leaves its partition's vertices unwritten): nothing behind it may read that state - the host runs the range again or reports the error
    // (the 1024-thread instance is alone on its CU - 16 waves at 128 VGPRs - and takes a class table twice the size: cells of 40-80 k
    //  reads ask for 3 000-6 000 classes, and out of the pool the class step cost them twice as much: 244 against 124 us at 40 k reads)
    constexpr uint32_t GTab = GNT >= 1024 ? 2 * kGTab : kGTab, GTabLoad = GTab * 3 / 4 - 72, GBloomWords = GNT >= 1024 ? 512 : 128, GLds = 3 * GTab + GBloomWords;
    __shared__ __attribute__((aligned(16))) uint32_t s_big[GLds];
    __shared__ uint32_t s_ws[GNT / 64];
    __shared__ uint32_t s_cnt[4];
    __shared__ uint32_t s_flag[4];
    __shared__ unsigned long long s_ebase;
    __shared__ uint32_t s_next;
#ifdef AFQ_PUG_TIMING
    __shared__ unsigned long long tmark[16];
#endif
    const uint32_t tid = threadIdx.x;
  for (;;) {
    __syncthreads();
    if (tid == 0) s_next = work_lo + atomicAdd(counter, 1u);
    __syncthreads();
    const uint32_t work = s_next;
    if (work >= work_hi) return;
    const uint32_t j = list[work];
    const P2Cell c = A.cells[j];
    const uint32_t R = c.R;
    auto give_up = [&]() {   // the cell goes to the one-workgroup kernel
        if (tid == 0) { A.fb[j] = 1; A.fb_list[atomicAdd(A.fb_count, 1u)] = c.cell; }
    };
    if (A.fb[j]) { if (tid == 0) A.fb_list[atomicAdd(A.fb_count, 1u)] = c.cell; continue; }
    uint32_t* gc = A.gcnt + 4 * (size_t)j;
    if (tid < 4) { s_cnt[tid] = tid == 0 ? R : gc[tid]; s_flag[tid] = 0; }   // (entries 0..R of the column list belong to the lone-vertex kernel)
    gsync();
    PugCtx C = make_ctx(A, c, s_cnt);
    const uint64_t* ch = A.s_h + c.rd_base;
    const uint64_t* cu = A.s_u + c.rd_base;
    const uint32_t* coff = A.v_off + c.rd_base;
    uint32_t* lidx = A.lidx + c.rd_base;
    G_MARK(0);
    // ---- 0. what the partition kernels left per partition: pairs (search), two-gene classes (lone vertices), vertices under
    //         a hashed label key ----
    const uint32_t P = 1u << c.lgP;
    const uint32_t* pnp = A.pnp + c.part_base;
    const uint32_t* pcn = A.pcnt + c.part_base;
    const uint32_t* pncls = A.pncls + c.part_base;
    const uint32_t* ppoff = A.poff + c.part_base;
    uint32_t n_pairs = 0, n_cls2 = 0;
    uint32_t* ppre = s_big;   // per partition: first pair | first class (cells of more than 6000 partitions - a million reads - keep it in the pool)
    if (2 * P > GLds) {
        if (tid == 0) s_ebase = atomicAdd(A.pool_cur, 2ull * P + 4);
        gsync();
        if (s_ebase + 2ull * P + 4 > A.pool_cap) { if (tid == 0) set_err(A.st, kErrPugPool, c.cell); return; }
        ppre = A.pool + s_ebase;
        gsync();
    }
    for (uint32_t base = 0; base < P; base += GNT) {
        const uint32_t pp = base + tid;
        const uint32_t a = pp < P ? pnp[pp] : 0u, b = pp < P ? pncls[pp] : 0u;
        uint32_t ta, tb;
        const uint32_t ea = block_excl_scan<GNT>(a, s_ws, ta);
        const uint32_t eb = block_excl_scan<GNT>(b, s_ws, tb);
        if (pp < P) { ppre[2 * pp] = n_pairs + ea; ppre[2 * pp + 1] = n_cls2 + eb; }
        n_pairs += ta; n_cls2 += tb;
    }
    gsync();
    if (n_cls2) {   // the lone vertices' classes into the cell's label area (em only)
        const uint32_t w0 = s_cnt[1], d0 = s_cnt[2];
        if (w0 + 2 * n_cls2 > C.lab_cap || 2 * (d0 + n_cls2) > C.lab_cap) { if (tid == 0) set_err(A.st, kErrPugLimit, c.cell); return; }
        const uint64_t* stage = A.cstage + c.rd_base;
        for (uint32_t pp = tid; pp < P; pp += GNT) {
            const uint32_t nk = pncls[pp], at = ppre[2 * pp + 1], so = ppoff[pp];
            for (uint32_t k = 0; k < nk; ++k) {
                const uint64_t v = stage[so + k];
                const uint32_t w = w0 + 2 * (at + k), d = d0 + at + k;
                C.labw[w] = (uint32_t)v; C.labw[w + 1] = (uint32_t)(v >> 32);
                C.labd[2 * d] = w; C.labd[2 * d + 1] = 2;
            }
        }
        gsync();
        if (tid == 0) { s_cnt[1] = w0 + 2 * n_cls2; s_cnt[2] = d0 + n_cls2; }
        gsync();
    }
    // ---- scratch out of the pool, in three steps as the sizes become known: the pair list (n_pairs), the per-vertex arrays
    //      (NT vertices have an edge), the per-slot arrays of the listed components (S_mid) ----
    auto pool_take = [&](unsigned long long words) -> uint32_t* {   // workgroup-wide call; nullptr = the pool is exhausted (error set)
        gsync();
        if (tid == 0) s_ebase = atomicAdd(A.pool_cur, words + 4);
        gsync();
        if (s_ebase + words + 4 > A.pool_cap) { if (tid == 0) set_err(A.st, kErrPugPool, c.cell); return nullptr; }
        return A.pool + ((s_ebase + 3) & ~3ull);
    };
    uint32_t* q = pool_take(2ull * n_pairs);
    if (!q) return;
    uint64_t* lp = reinterpret_cast<uint64_t*>(q);   // pairs over local ids: x | y << 24 | directions << 48
    G_MARK(1);
    // ---- 1. the touched vertices: the search flagged them; a scan over the flags counts them, a second numbers them (any
    //         numbering will do: the reference's order enters through the order keys of step 6 only) ----
    const uint64_t* psrc = A.pairs + c.rd_base;
    const uint8_t* cflag = A.v_flag + c.rd_base;
    uint32_t NT = 0;
    for (uint32_t g0 = 16 * tid; g0 < R; g0 += 16 * GNT) {   // (sixteen flags per thread and trip in flight: this kernel's phases are chains of round trips)
#pragma unroll
        for (int r = 0; r < 16; ++r) NT += g0 + r < R && (cflag[g0 + r] & 1u) != 0;
    }
    {
        uint32_t tot;
        (void)block_excl_scan<GNT>(NT, s_ws, tot);
        NT = tot;
    }
    if (NT > 2 * n_pairs || NT >= (1u << 24)) { if (tid == 0) set_err(A.st, kErrPugLimit, c.cell); return; }   // (cannot happen: two end points per pair, R < 2^22)
    const uint32_t nt_max = NT;
    q = pool_take(9ull * NT + 18);
    if (!q) return;
    uint32_t* tl = q; q += nt_max;            // touched vertex -> its slot in the cell
    uint32_t* wlg = q; q += nt_max;           // component labels when they do not fit LDS
    uint32_t* root_of = q; q += nt_max;
    uint32_t* rcnt = q; q += nt_max;          // per root: vertices of its component, then its place in the lists (category << 28 | index)
    uint32_t* fill = q; q += nt_max;          // per root: next free slot of its component; afterwards, per vertex:
    uint32_t* cidx = fill;                    // ... its position inside its component (reference order)
    uint32_t* lsize = q; q += nt_max + 2;     // per listed component: size
    uint32_t* mid_off = q; q += nt_max + 12;  // ... first slot (and behind the last one ten words for the cover kernels: see the end)
    uint32_t* pr_v = q; q += nt_max + 2;      // two-vertex components: their vertices, two by two
    {
        uint32_t carry = 0;
        for (uint32_t base = 0; base < R; base += 16 * GNT) {
            const uint32_t g0 = base + 16 * tid;
            uint32_t fm = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) fm |= (uint32_t)(g0 + r < R && (cflag[g0 + r] & 1u) != 0) << r;
            uint32_t tot;
            uint32_t li = carry + block_excl_scan<GNT>((uint32_t)__popc(fm), s_ws, tot);
            for (; fm; fm &= fm - 1) { const uint32_t g = g0 + (uint32_t)__builtin_ctz(fm); tl[li] = g; lidx[g] = li; ++li; }
            carry += tot;
        }
    }
    gsync();
    for (uint32_t pp = tid; pp < P; pp += GNT) {   // the partition's pairs over touched-vertex numbers, four at a time
        const uint32_t nk = pnp[pp], so = ppoff[pp], at = ppre[2 * pp];
        const uint64_t* src = nk > pcn[pp] ? reinterpret_cast<const uint64_t*>(A.pool + psrc[so]) : psrc + so;   // (more pairs than own slots: the search put the list into the pool)
        for (uint32_t k0 = 0; k0 < nk; k0 += 4) {
            uint64_t pr[4];
            uint32_t lx[4], ly[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) pr[r] = k0 + r < nk ? src[k0 + r] : 0ull;
#pragma unroll
            for (int r = 0; r < 4; ++r) { lx[r] = k0 + r < nk ? lidx[(uint32_t)(pr[r] >> 31) & 0x7FFFFFFFu] : 0u; ly[r] = k0 + r < nk ? lidx[(uint32_t)pr[r] & 0x7FFFFFFFu] : 0u; }
#pragma unroll
            for (int r = 0; r < 4; ++r) if (k0 + r < nk) lp[at + k0 + r] = (pr[r] >> 62) ? (uint64_t)lx[r] | ((uint64_t)ly[r] << 24) | ((pr[r] >> 62) << 48) : 0ull;   // (0: a candidate k_p2_check cleared - the loops over lp skip it)
        }
    }
    gsync();
    G_MARK(2);
    // ---- 2. weakly connected components over the pair list: union-find with compare-and-swap hooking (parents in LDS when they
    //         fit, else in the pool through workgroup-scope L2 atomics).  A root is only ever hooked under a SMALLER vertex, so
    //         the parent pointers cannot close a cycle; find() shortens the path it walks (a racing shortcut still points at an
    //         ancestor).  ONE pass over the pairs and one barrier - rounds 3-4 swept the pairs until no label moved, four
    //         pointer-jumping sweeps and six barriers per sweep (a fifth of this kernel on the largest cells) ----
    const bool wl_lds = NT <= GLds;
    uint32_t* wl = wl_lds ? s_big : wlg;
    auto ldw = [&](uint32_t i) -> uint32_t { return wl_lds ? __hip_atomic_load(&wl[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : ld_l2(&wl[i]); };
    for (uint32_t i = tid; i < NT; i += GNT) { if (wl_lds) wl[i] = i; else st_l2(&wl[i], i); st_l2(&rcnt[i], 0u); st_l2(&fill[i], 0u); }
    gsync();
    {
        auto find = [&](uint32_t i) -> uint32_t {
            uint32_t p = ldw(i);
            while (p != i) {
                const uint32_t gp2 = ldw(p);
                if (gp2 != p) wg_min(&wl[i], gp2);   // path halving (a min: never lengthens a path another thread just shortened)
                i = p; p = gp2;
            }
            return i;
        };
        for (uint32_t k = tid; k < n_pairs; k += GNT) {
            const uint64_t e = lp[k];
            if (!e) continue;
            uint32_t a = find((uint32_t)e & 0xFFFFFFu), b = find((uint32_t)(e >> 24) & 0xFFFFFFu);
            while (a != b) {
                if (a < b) { const uint32_t t = a; a = b; b = t; }   // the larger root goes under the smaller
                uint32_t expected = a;
                if (__hip_atomic_compare_exchange_strong(&wl[a], &expected, b, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
                a = find(expected); b = find(b);   // (a had been hooked by somebody else meanwhile: go on from where it hangs now)
            }
        }
    }
    gsync();
    G_MARK(3);
    // ---- 3. the components by counting: sizes per root, then by size pairs / 3..8 / 9..64 / 65..4096 (anything else - a
    //         component of more than 4096 vertices that is still under --large-graph-thresh - is not for this kernel), every
    //         listed component's slots, every vertex into its component's next slot.  A component above --large-graph-thresh
    //         needs no order and no edges: its vertices are listed apart (lg_v), the cover kernel resolves it winner-take-all
    //         (pugutils.rs:916-982) ----
    for (uint32_t i = tid; i < NT; i += GNT) { uint32_t l = ldw(i); for (uint32_t nx = ldw(l); nx != l; nx = ldw(l)) l = nx; root_of[i] = l; wg_add(&rcnt[l], 1u); }
    gsync();
    uint32_t n_pr = 0, n_tiny = 0, n_mid9 = 0, n_bigc = 0;
    bool big = false;
    for (uint32_t base = 0; base < NT; base += GNT) {
        const uint32_t i = base + tid;
        uint32_t cat = 0;
        if (i < NT && root_of[i] == i) {
            const uint32_t n = ld_l2(&rcnt[i]);
            if (n > C.large_thresh) cat = kCatLarge;
            else if (n > A.max_comp) big = true;
            else cat = n == 2 ? kCatPair : n <= 8 ? kCatTiny : n <= 64 ? kCatMid : kCatBig;
        }
        uint32_t tot, tot_mid;   // (two scans: three counts of up to GNT = 1024 do not fit one word)
        const uint32_t ex = block_excl_scan<GNT>((cat == kCatPair) | ((uint32_t)(cat == kCatTiny) << 16), s_ws, tot);
        const uint32_t ex_mid = block_excl_scan<GNT>((uint32_t)(cat == kCatMid) | ((uint32_t)(cat == kCatBig) << 16), s_ws, tot_mid);
        if (cat) {
            const uint32_t n = ld_l2(&rcnt[i]);
            const uint32_t idx = cat == kCatLarge ? atomicAdd(&s_flag[2], 1u)   // (few, and their order is free: an LDS counter instead of a third scan)
                                 : cat == kCatPair ? n_pr + (ex & 0xFFFFu) : cat == kCatTiny ? n_tiny + (ex >> 16) : cat == kCatMid ? n_mid9 + (ex_mid & 0xFFFFu) : n_bigc + (ex_mid >> 16);
            if (cat == kCatLarge) atomicAdd(&s_flag[3], n);
            st_l2(&rcnt[i], (cat << 28) | idx);            // (a root's word is read and written by its own thread only in this pass)
            if (cat == kCatTiny) lsize[idx] = n;          // (the 9..64 ones are placed behind the small ones once those are counted)
            else if (cat != kCatPair) st_l2(&fill[i], n);   // (parked in the root's fill word until then)
        }
        n_pr += tot & 0xFFFFu; n_tiny += tot >> 16; n_mid9 += tot_mid & 0xFFFFu; n_bigc += tot_mid >> 16;
    }
    if (big) s_flag[1] = 1;
    gsync();
    if (s_flag[1]) { give_up(); continue; }
    const uint32_t n_large = s_flag[2], S_large = s_flag[3];
    uint32_t* lg_off = nullptr;   // per component above the threshold: its first entry of lg_v (n_large + 1 offsets)
    uint32_t* lg_v = nullptr;     // ... its vertices (touched-vertex numbers)
    if (n_large) {
        q = pool_take(n_large + 2ull + S_large);
        if (!q) return;
        lg_off = q; lg_v = q + n_large + 2;
    }
    const uint32_t n_mid = n_tiny + n_mid9, n_all = n_mid + n_bigc;   // the list: 3..8, then 9..64, then 65..4096
    auto listed_at = [&](uint32_t cat, uint32_t idx) -> uint32_t { return cat == kCatTiny ? idx : cat == kCatMid ? n_tiny + idx : n_mid + idx; };
    for (uint32_t i = tid; i < NT; i += GNT)
        if (root_of[i] == i) {
            const uint32_t rc = ld_l2(&rcnt[i]), cat = rc >> 28;
            if (cat == kCatMid || cat == kCatBig) { lsize[listed_at(cat, rc & 0xFFFFFFFu)] = ld_l2(&fill[i]); st_l2(&fill[i], 0u); }
            else if (cat == kCatLarge) { lg_off[rc & 0xFFFFFFFu] = ld_l2(&fill[i]); st_l2(&fill[i], 0u); }
        }
    gsync();
    if (tid == 0 && n_large) {   // sizes -> offsets (a handful)
        uint32_t acc = 0;
        for (uint32_t b = 0; b < n_large; ++b) { const uint32_t t = lg_off[b]; lg_off[b] = acc; acc += t; }
        lg_off[n_large] = acc;
    }
    uint32_t S_mid = 0;
    for (uint32_t base = 0; base < n_all; base += GNT) {
        const uint32_t ci = base + tid;
        const uint32_t n = ci < n_all ? lsize[ci] : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan<GNT>(n, s_ws, tot);
        if (ci < n_all) mid_off[ci] = S_mid + ex;
        if (ci == n_mid && ci < n_all) s_flag[0] = S_mid + ex;   // the first slot of the components of more than 64 vertices
        S_mid += tot;
    }
    if (tid == 0) mid_off[n_all] = S_mid;
    gsync();
    // The reference's vertex order inside a component only decides TIES between equal-size arborescences.  Components of up to
    // 64 vertices are therefore covered in whatever order their vertices took their slots (k_p2_cover), a component is set
    // aside at the first round that does meet a tie, and k_p2_tied finds the class minima for those alone - one in a dozen -
    // and finishes them in the reference's order.  The components of more than 64 vertices (rare; cover_big has no notion of
    // being set aside) still get their order here: S_lo.. are their slots.
    // A cell whose classes fit the LDS table keeps round 4's way (every listed component ordered here, the covers in that order):
    // setting aside costs a kernel of its own with a chain of dependent steps per cell (k_p2_tied) - a third of the components of
    // three or more vertices meet a tie - and only pays where the class step is expensive: the big cells of a sample and every
    // cell whose reads carry long labels, whose classes go to a table in the pool (measured, profiles/round5_05_tied.txt).
    const bool defer = S_mid > (A.defer_min == 0xFFFFFFFFu ? GTabLoad : A.defer_min);
    const uint32_t S_lo = !defer ? 0u : n_bigc ? s_flag[0] : S_mid;
    // per slot of a listed component: its vertex, its component, its class minimum; and one region that first holds the class
    // table (when it does not fit LDS) and then the order keys, adjacency masks and cover records (12 words per slot)
    const uint32_t want = S_mid - S_lo;   // (vertices: an upper bound of the classes the table will hold)
    constexpr uint32_t kPoolTab = 1u << 16;
    uint32_t cap = GTab, n_slices = 1;
    if (want > GTabLoad) {   // at most 2^16 pool slots at a time: a cell with more classes takes the key space in slices, a pass per slice
        n_slices = (uint32_t)((5ull * want / 2 + kPoolTab - 1) / kPoolTab);
        while (cap < kPoolTab && (uint64_t)cap * n_slices < 5ull * want / 2) cap <<= 1;
    }
    const unsigned long long u_words = std::max<unsigned long long>(want > GTabLoad ? 3ull * cap + 4 : 0ull, 12ull * S_mid + 8);
    q = pool_take(3ull * S_mid + 8 + u_words + 4ull * n_mid + 16);
    if (!q) return;
    uint32_t* slot_v = q; q += S_mid + 2;       // slot -> vertex
    uint32_t* slot_comp = q; q += S_mid + 2;    // slot -> listed component
    uint32_t* cmin = q; q += S_mid + 2;         // per slot: smallest record offset of the vertex's class
    q += (q - A.pool) & 1;                      // (8-byte aligned from here; the first take was 16-byte aligned and sizes above are even or padded)
    q = A.pool + (((unsigned long long)(q - A.pool) + 3) & ~3ull);
    uint32_t* const u_base = q;
    uint4* mrec = reinterpret_cast<uint4*>(u_base);                                   // [2 * S_mid]
    uint64_t* okey = reinterpret_cast<uint64_t*>(u_base + 8 * (size_t)S_mid);         // [S_mid] (class minimum, UMI)
    unsigned long long* adjp = reinterpret_cast<unsigned long long*>(u_base + 10 * (size_t)S_mid);   // [S_mid] out-neighbours of the vertex at this position, as positions inside its component
    // the components k_p2_cover sets aside at a tie: two counters (3..8 vertices, 9..64), then four words per component
    // (list index, the vertices still uncovered as a 64-bit mask over its slots, a running key count for k_p2_tied's batches)
    uint32_t* const tied = u_base + ((u_words + 3) & ~3ull);
    // components of more than 64 vertices: their adjacency as rows of ceil(n / 64) mask words, n rows each, in a region of its own;
    // where a component's rows start (in 64-bit words) takes over its entry of lsize[], which is read for the last time above
    uint32_t* const rowoff = lsize + n_mid;
    unsigned long long* rows = nullptr;
    if (n_bigc) {
        uint32_t rows_tot = 0;
        for (uint32_t base = 0; base < n_bigc; base += GNT) {
            const uint32_t b = base + tid;
            const uint32_t n = b < n_bigc ? mid_off[n_mid + b + 1] - mid_off[n_mid + b] : 0u;
            uint32_t tot;
            const uint32_t ex = block_excl_scan<GNT>(n * ((n + 63) / 64), s_ws, tot);
            if (b < n_bigc) rowoff[b] = rows_tot + ex;
            rows_tot += tot;
        }
        uint32_t* rq = pool_take(2ull * rows_tot + 4);
        if (!rq) return;
        rows = reinterpret_cast<unsigned long long*>(rq);
        for (uint32_t i = tid; i < rows_tot; i += GNT) st_l2(&rows[i], 0ull);
    }
    for (uint32_t i = tid; i < NT; i += GNT) {
        const uint32_t r = root_of[i], rc = ld_l2(&rcnt[r]), cat = rc >> 28, idx = rc & 0xFFFFFFFu;
        const uint32_t at = wg_add(&fill[r], 1u);
        if (cat == kCatPair) pr_v[2 * idx + at] = i;
        else if (cat == kCatLarge) lg_v[lg_off[idx] + at] = i;
        else {
            const uint32_t ci = listed_at(cat, idx);
            const uint32_t sl = mid_off[ci] + at;
            slot_v[sl] = i; slot_comp[sl] = ci;
        }
    }
    gsync();
    G_MARK(4);
    // ---- 4. class minima for the classes of the listed components' vertices: their order decides ties.  The cell's vertex
    //         slots stream ONCE through a hash table keyed by label key that holds the asked-for classes (slots past a
    //         partition's last vertex hold key 0: no class) - in LDS when they are few, out of the pool otherwise.  A vertex
    //         under a HASHED key that finds its key in the table is compared with the vertex that held the minimum before it -
    //         whichever that was, so every vertex of an asked-for class is chained to the first one that arrived and equal
    //         keys are shown to be equal labels.  (Classes nobody asks for are not looked at: nothing takes two such vertices
    //         for one class - the partition kernel compares the reads it merges into a vertex, the search compares labels
    //         under hashed keys by content.  The pass over every hashed vertex of the cell that used to sit here was a
    //         third of this phase.) ----
    if (want) {
        // the table: in LDS when the classes are few; else in the pool region taken above
        unsigned long long* t_key = reinterpret_cast<unsigned long long*>(s_big);
        uint32_t* t_min = s_big + 2 * GTab;
        uint32_t* s_bloom = s_big + 3 * GTab;   // the last words of the block: 4096 (16 384) bits for the asked-for keys
        uint32_t bloom_shift = GBloomWords == 512 ? 18 : 20, bloom_words = GBloomWords;
        if (want > GTabLoad) {
            t_key = reinterpret_cast<unsigned long long*>(u_base);
            t_min = reinterpret_cast<uint32_t*>(t_key + cap);
            s_bloom = s_big;   // (the LDS block is free then: 2^18 bits of it, or 2^17 under a smaller block)
            bloom_words = GLds >= 8192 ? 8192 : 4096; bloom_shift = GLds >= 8192 ? 14 : 15;
        }
        const uint32_t cmask = cap - 1;
        auto mix = [](uint64_t h) -> uint32_t { uint32_t x = ((uint32_t)h ^ (uint32_t)(h >> 32)) * 0x9E3779B1u; return x ^ (x >> 15); };
        auto slice_of = [&](uint64_t h) -> uint32_t { return n_slices == 1 ? 0u : (uint32_t)(((h ^ (h >> 29)) * 0xD6E8FEB86659FD93ull) >> 40) % n_slices; };
        auto find = [&](uint64_t h, uint32_t mx, bool insert) -> uint32_t {   // slot of h, or 0xFFFFFFFF
            uint32_t slot = mx & cmask;
            for (uint32_t step = 0; step < cap; ++step, slot = (slot + 1) & cmask) {
                unsigned long long k = ld_l2(&t_key[slot]);
                if (k == h) return slot;
                if (k == ~0ull) {
                    if (!insert) return 0xFFFFFFFFu;
                    unsigned long long expected = ~0ull;
                    if (__hip_atomic_compare_exchange_strong(&t_key[slot], &expected, (unsigned long long)h, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return slot;
                    if (expected == h) return slot;
                }
            }
            return 0xFFFFFFFFu;
        };
        for (uint32_t sl = 0; sl < n_slices; ++sl) {
            gsync();
            for (uint32_t i = tid; i < cap; i += GNT) { st_l2(&t_key[i], ~0ull); st_l2(&t_min[i], 0xFFFFFFFFu); }
            for (uint32_t i = tid; i < bloom_words; i += GNT) s_bloom[i] = 0;
            gsync();
            // (Every loop below takes four to six items per thread and level: the loads and L2 atomics of a level go out together and
            //  are waited for once.  One item at a time, a thread went through five dependent round trips per vertex.)
            for (uint32_t s0 = S_lo + tid; s0 - tid < S_mid; s0 += 4 * GNT) {   // the classes that are asked for
                uint32_t v4[4];
                uint64_t h4[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v4[r] = s0 + (uint32_t)r * GNT < S_mid ? slot_v[s0 + (uint32_t)r * GNT] : 0u;
#pragma unroll
                for (int r = 0; r < 4; ++r) v4[r] = s0 + (uint32_t)r * GNT < S_mid ? tl[v4[r]] : 0u;
#pragma unroll
                for (int r = 0; r < 4; ++r) h4[r] = s0 + (uint32_t)r * GNT < S_mid ? ch[v4[r]] : 0ull;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (s0 + (uint32_t)r * GNT >= S_mid || slice_of(h4[r]) != sl) continue;
                    const uint32_t mx = mix(h4[r]);
                    atomicOr(&s_bloom[(mx >> bloom_shift) >> 5], 1u << ((mx >> bloom_shift) & 31u));
                    if (find(h4[r], mx, true) == 0xFFFFFFFFu) s_cnt[3] = kErrInternal;
                }
            }
            gsync();
            // every vertex slot once
            auto find_on = [&](uint64_t h, uint32_t slot, unsigned long long k) -> uint32_t {   // find(), the first probe already made
                for (uint32_t step = 0; step < cap; ++step) {
                    if (k == h) return slot;
                    if (k == ~0ull) return 0xFFFFFFFFu;
                    slot = (slot + 1) & cmask;
                    k = ld_l2(&t_key[slot]);
                }
                return 0xFFFFFFFFu;
            };
            for (uint32_t g0 = tid; g0 - tid < R; g0 += 6 * GNT) {
                uint64_t h4[6];
                unsigned long long k4[6];
                uint32_t sl4[6], off4[6], old4[6];
#pragma unroll
                for (int r = 0; r < 6; ++r) h4[r] = g0 + (uint32_t)r * GNT < R ? ch[g0 + (uint32_t)r * GNT] : 0ull;
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    const uint32_t mx = mix(h4[r]);
                    const bool pass = h4[r] != 0 && ((s_bloom[(mx >> bloom_shift) >> 5] >> ((mx >> bloom_shift) & 31u)) & 1u) && slice_of(h4[r]) == sl;
                    sl4[r] = pass ? mx & cmask : 0xFFFFFFFFu;
                }
#pragma unroll
                for (int r = 0; r < 6; ++r) k4[r] = sl4[r] != 0xFFFFFFFFu ? ld_l2(&t_key[sl4[r]]) : ~0ull;
#pragma unroll
                for (int r = 0; r < 6; ++r) if (sl4[r] != 0xFFFFFFFFu) sl4[r] = find_on(h4[r], sl4[r], k4[r]);
#pragma unroll
                for (int r = 0; r < 6; ++r) off4[r] = sl4[r] != 0xFFFFFFFFu ? coff[g0 + (uint32_t)r * GNT] : 0u;
#pragma unroll
                for (int r = 0; r < 6; ++r) old4[r] = sl4[r] != 0xFFFFFFFFu ? wg_min(&t_min[sl4[r]], off4[r]) : 0xFFFFFFFFu;
#pragma unroll
                for (int r = 0; r < 6; ++r)
                    if (sl4[r] != 0xFFFFFFFFu && (uint32_t)(h4[r] >> 62) == 3 && old4[r] != 0xFFFFFFFFu && old4[r] != off4[r] &&
                        !lab_equal(rec_label(C, off4[r]), rec_label(C, old4[r]))) s_cnt[3] = kErrLabelHash;
            }
            gsync();
            for (uint32_t s0 = S_lo + tid; s0 - tid < S_mid; s0 += 4 * GNT) {
                uint32_t v4[4], sl4[4];
                uint64_t h4[4];
                unsigned long long k4[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v4[r] = s0 + (uint32_t)r * GNT < S_mid ? slot_v[s0 + (uint32_t)r * GNT] : 0u;
#pragma unroll
                for (int r = 0; r < 4; ++r) v4[r] = s0 + (uint32_t)r * GNT < S_mid ? tl[v4[r]] : 0u;
#pragma unroll
                for (int r = 0; r < 4; ++r) h4[r] = s0 + (uint32_t)r * GNT < S_mid ? ch[v4[r]] : 0ull;
#pragma unroll
                for (int r = 0; r < 4; ++r) sl4[r] = s0 + (uint32_t)r * GNT < S_mid && slice_of(h4[r]) == sl ? mix(h4[r]) & cmask : 0xFFFFFFFFu;
#pragma unroll
                for (int r = 0; r < 4; ++r) k4[r] = sl4[r] != 0xFFFFFFFFu ? ld_l2(&t_key[sl4[r]]) : ~0ull;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (sl4[r] == 0xFFFFFFFFu) continue;   // (past the end, or another slice's)
                    const uint32_t slot = find_on(h4[r], sl4[r], k4[r]);
                    const uint32_t mn = slot == 0xFFFFFFFFu ? 0xFFFFFFFFu : ld_l2(&t_min[slot]);
                    if (mn == 0xFFFFFFFFu) s_cnt[3] = kErrInternal;   // (every asked-for class has at least the vertex that asked)
                    cmin[s0 + (uint32_t)r * GNT] = mn;
                }
            }
        }
        gsync();
        if (s_cnt[3]) { if (tid == 0) set_err(A.st, s_cnt[3], c.cell); return; }
    }
    gsync();
    G_MARK(5);
    G_MARK(6);
    // ---- 6. components of 3..64 vertices: their vertices in the reference's order (class by first appearance = smallest
    //         record offset, then UMI), the edges between them as masks over those positions, gathered into the covers' records ----
    for (uint32_t s2 = S_lo + tid; s2 < S_mid; s2 += GNT) okey[s2] = ((uint64_t)cmin[s2] << 32) | (uint32_t)(cu[tl[slot_v[s2]]] >> 32);
    gsync();
    for (uint32_t s2 = tid; s2 < S_mid; s2 += GNT) {
        const uint32_t ci = slot_comp[s2];
        const uint32_t b0 = mid_off[ci], n = mid_off[ci + 1] - b0;
        if (s2 < S_lo) { cidx[slot_v[s2]] = s2 - b0; continue; }   // (up to 64 vertices: the slot order; k_p2_tied reorders the few that need it)
        const uint64_t mine = okey[s2];
        uint32_t rank = 0;
        uint32_t same = 0;
        for (uint32_t i = 0; i < n; ++i) { rank += okey[b0 + i] < mine; same += okey[b0 + i] == mine; }
        if (same != 1) s_cnt[3] = kErrInternal;   // (two vertices of one component with the same class and UMI: cannot be)
        cidx[slot_v[s2]] = rank;
    }
    gsync();
    if (s_cnt[3]) { if (tid == 0) set_err(A.st, s_cnt[3], c.cell); return; }
    for (uint32_t s2 = tid; s2 < S_mid; s2 += GNT) st_l2(&adjp[s2], 0ull);
    gsync();
    for (uint32_t k = tid; k < n_pairs; k += GNT) {
        const uint64_t e = lp[k];
        if (!e) continue;
        const uint32_t x = (uint32_t)e & 0xFFFFFFu, y = (uint32_t)(e >> 24) & 0xFFFFFFu;
        const uint32_t rc = ld_l2(&rcnt[root_of[x]]), cat = rc >> 28;
        if (cat == kCatPair || cat == kCatLarge) continue;
        if (cat == kCatBig) {
            const uint32_t b = rc & 0xFFFFFFFu, n = mid_off[n_mid + b + 1] - mid_off[n_mid + b], nw = (n + 63) / 64;
            unsigned long long* rw = rows + rowoff[b];
            const uint32_t ix = cidx[x], iy = cidx[y];
            if (e & (2ull << 48)) __hip_atomic_fetch_or(&rw[(size_t)ix * nw + (iy >> 6)], 1ull << (iy & 63u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // x -> y
            if (e & (1ull << 48)) __hip_atomic_fetch_or(&rw[(size_t)iy * nw + (ix >> 6)], 1ull << (ix & 63u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // y -> x
            continue;
        }
        const uint32_t b0 = mid_off[listed_at(cat, rc & 0xFFFFFFFu)];   // (x and y share their component)
        if (e & (2ull << 48)) __hip_atomic_fetch_or(&adjp[b0 + cidx[x]], 1ull << cidx[y], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // x -> y
        if (e & (1ull << 48)) __hip_atomic_fetch_or(&adjp[b0 + cidx[y]], 1ull << cidx[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // y -> x
    }
    gsync();
    for (uint32_t s2 = tid; s2 < S_mid; s2 += GNT) {
        const uint32_t li = slot_v[s2];
        const uint32_t g = tl[li];
        const KLab l = klab(C.W, C.HW, ch[g], coff[g]);
        uint32_t r0 = 0xFFFFFFFFu, r1 = 0xFFFFFFFFu, r2 = 0xFFFFFFFFu, r3 = 0xFFFFFFFFu;
        if (l.n <= 4) {
            if (l.n > 0) r0 = klab_ref(l, 0);
            if (l.n > 1) r1 = klab_ref(l, 1);
            if (l.n > 2) r2 = l.p[2] & 0x7FFFFFFFu;
            if (l.n > 3) r3 = l.p[3] & 0x7FFFFFFFu;
        } else { const uint64_t pa = (uint64_t)(uintptr_t)l.p; r0 = (uint32_t)pa; r1 = (uint32_t)(pa >> 32); }
        const size_t at = (size_t)mid_off[slot_comp[s2]] + cidx[li];
        const unsigned long long am = ld_l2(&adjp[at]);
        mrec[2 * at] = make_uint4(g, l.n, r0, r1);
        mrec[2 * at + 1] = make_uint4(r2, r3, (uint32_t)am, (uint32_t)(am >> 32));
    }
    gsync();
    G_MARK(7);
#ifdef AFQ_PUG_TIMING
    if (tid == 0 && (work % 512) < 2) {
        auto us = [&](int a, int b) { return (double)(tmark[b] - tmark[a]) / 100.0; };
        printf("p2 graph cell R=%u pairs=%u NT=%u n_tiny=%u n_mid=%u n_pr=%u S_mid=%u: gather=%.0f touched=%.0f wcc=%.0f comps=%.0f classes=%.0f records=%.0f total=%.0f us\n",
               R, n_pairs, NT, n_tiny, n_mid, n_pr, S_mid, us(0, 1), us(1, 2), us(2, 3), us(3, 4), us(4, 5), us(6, 7), us(0, 7));
    }
#endif
    if (s_cnt[3]) { if (tid == 0) set_err(A.st, s_cnt[3], c.cell); return; }
    // what the cover kernel (k_p2_cover) takes over: where the cell's lists lie in the pool, how many there are, the cell's counters
    if (tid == 0) {
        uint32_t* d = A.gdesc + kGDescWords * (size_t)j;
        auto put = [&](int at, const void* ptr) { const unsigned long long o = (unsigned long long)(reinterpret_cast<const uint32_t*>(ptr) - A.pool); d[at] = (uint32_t)o; d[at + 1] = (uint32_t)(o >> 32); };
        d[1] = n_pr; d[2] = n_tiny; d[3] = n_mid; d[15] = n_bigc | (n_large ? 0x80000000u : 0u);
        put(4, tl); put(6, pr_v); put(8, mid_off); put(10, mrec);
        if (n_bigc || n_large) {   // (where the rows, their offsets and the lists of the components above the threshold lie: behind the list's last offset)
            const unsigned long long ro = n_bigc ? (unsigned long long)(reinterpret_cast<const uint32_t*>(rows) - A.pool) : 0ull, oo = (unsigned long long)(rowoff - A.pool);
            const unsigned long long lo = n_large ? (unsigned long long)(lg_off - A.pool) : 0ull;
            mid_off[n_all + 1] = (uint32_t)ro; mid_off[n_all + 2] = (uint32_t)(ro >> 32);
            mid_off[n_all + 3] = (uint32_t)oo; mid_off[n_all + 4] = (uint32_t)(oo >> 32);
            mid_off[n_all + 5] = n_large; mid_off[n_all + 6] = (uint32_t)lo; mid_off[n_all + 7] = (uint32_t)(lo >> 32);
        }
        put(16, tied);
        tied[0] = 0; tied[1] = 0;
        d[12] = s_cnt[0]; d[13] = s_cnt[1]; d[14] = s_cnt[2];
        d[0] = defer ? 3u : 1u;   // bit 0: the lists are there; bit 1: components of up to 64 vertices lie in slot order (kCoverDefer)
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// 6. one workgroup per cell again, but a kernel of its own: the molecules of the components the graph kernel listed - the
//    two-vertex rule and the arborescence covers (afq_pug_common.h).  They need none of the graph kernel's LDS and a third of its
//    registers; as one kernel the pair was 168 VGPRs with 98 SGPRs spilled and three workgroups to a CU.
// get_num_molecules_large_component (pugutils.rs:916-982) for the components above --large-graph-thresh of one cell: the
// (UMI, gene, reads) triplets of the component's vertices, sorted by the workgroup; per UMI the gene(s) with the most reads
// (resolve_num_molecules_crlike_from_vec, pugutils.rs:644-749).  Rare; thread 0 walks the sorted triplets.  false = the pool
// is exhausted (the error is set).
template <int CNT>
__device__ __forceinline__ bool cover_large(const P2Args& A, const PugCtx& C, const P2Cell& c, const uint32_t* tl, const uint32_t* lg_off, const uint32_t* lg_v,
                                            uint32_t n_large, uint32_t* s_ws, uint32_t* s_nt, unsigned long long* s_base) {
    const uint32_t tid = threadIdx.x;
    const uint64_t* ch = A.s_h + c.rd_base;
    const uint64_t* cu = A.s_u + c.rd_base;
    const uint32_t* coff = A.v_off + c.rd_base;
    // distinct genes of a label that has more of them than kMaxGenesPerLabel: first occurrences, found by looking back
    auto wide_genes = [&](const KLab& l, auto&& f) {
        for (uint32_t j = 0; j < l.n; ++j) {
            const uint32_t gj = C.t2g[klab_ref(l, j)];
            bool first = true;
            for (uint32_t q = 0; q < j && first; ++q) first = C.t2g[klab_ref(l, q)] != gj;
            if (first) f(gj);
        }
    };
    for (uint32_t b = 0; b < n_large; ++b) {
        const uint32_t v0 = lg_off[b], n = lg_off[b + 1] - v0;
        uint32_t cnt = 0;
        for (uint32_t i = tid; i < n; i += CNT) {
            const uint32_t g = tl[lg_v[v0 + i]];
            const KLab l = klab(C.W, C.HW, ch[g], coff[g]);
            uint32_t gg[kMaxGenesPerLabel];
            const uint32_t ng = genes_of(C, l.n, [&](uint32_t j) { return klab_ref(l, j); }, gg);
            if (ng == 0xFFFFFFFFu) wide_genes(l, [&](uint32_t) { ++cnt; }); else cnt += ng;
        }
        uint32_t tot;
        (void)block_excl_scan<CNT>(cnt, s_ws, tot);
        __syncthreads();
        if (tid == 0) { *s_base = atomicAdd(A.pool_cur, 4ull * tot + 4); *s_nt = 0; }
        __syncthreads();
        if (*s_base + 4ull * tot + 4 > A.pool_cap) { if (tid == 0) set_err(A.st, kErrPugPool, c.cell); return false; }
        uint4* trip = reinterpret_cast<uint4*>(A.pool + ((*s_base + 3) & ~3ull));   // (umi, 0, gene, reads)
        for (uint32_t i = tid; i < n; i += CNT) {
            const uint32_t g = tl[lg_v[v0 + i]];
            const uint64_t uw = cu[g];
            const KLab l = klab(C.W, C.HW, ch[g], coff[g]);
            uint32_t gg[kMaxGenesPerLabel];
            const uint32_t ng = genes_of(C, l.n, [&](uint32_t j) { return klab_ref(l, j); }, gg);
            if (ng == 0xFFFFFFFFu) {
                uint32_t k = 0;
                wide_genes(l, [&](uint32_t) { ++k; });
                uint32_t o = atomicAdd(s_nt, k);
                wide_genes(l, [&](uint32_t gid) { trip[o++] = make_uint4((uint32_t)(uw >> 32), 0u, gid, (uint32_t)uw & kVCntMask); });
                continue;
            }
            const uint32_t o = atomicAdd(s_nt, ng);
            for (uint32_t q = 0; q < ng; ++q) trip[o + q] = make_uint4((uint32_t)(uw >> 32), 0u, gg[q], (uint32_t)uw & kVCntMask);
        }
        __syncthreads();
        const uint32_t nt = *s_nt;
        bitonic_sort_by<CNT>(trip, nt, [](const uint4& x, const uint4& y) {
            if (x.x != y.x) return x.x > y.x;
            if (x.z != y.z) return x.z > y.z;
            return x.w > y.w;
        });
        if (tid == 0 && nt) {
            uint32_t best[kMaxGenesPerLabel];
            uint32_t nbest = 0, maxc = 0, aggr = 0;
            uint32_t cur_umi = trip[0].x, cg = trip[0].z;
            bool wide = false;
            uint32_t run0 = 0;   // first triplet of the current UMI
            // a UMI whose tie set has more genes than best[] holds is a class of its own for the EM: the genes whose summed count
            // is the maximum, ascending as the triplets are, written straight into the label area
            auto emit_ties = [&](uint32_t i0, uint32_t i1, uint32_t maxc_) {
                auto each_tied = [&](auto&& f) {
                    for (uint32_t i = i0; i < i1;) {
                        uint32_t j = i, sum = 0;
                        for (; j < i1 && trip[j].z == trip[i].z; ++j) sum += trip[j].w;
                        if (sum == maxc_) f(trip[i].z);
                        i = j;
                    }
                };
                uint32_t k = 0;
                each_tied([&](uint32_t) { ++k; });
                const uint32_t off = atomicAdd(&C.s_cnt[1], k), di = atomicAdd(&C.s_cnt[2], 1u);
                if (off + k > C.lab_cap || 2 * (di + 1) > C.lab_cap) { C.s_cnt[3] = kErrPugLimit; return; }
                uint32_t w = off;
                each_tied([&](uint32_t gid) { C.labw[w++] = gid; });
                C.labd[2 * di] = off; C.labd[2 * di + 1] = k;
            };
            for (uint32_t i = 0; i < nt; ++i) {
                const uint4 t = trip[i];
                if (i == 0 || t.x != cur_umi) {
                    if (i) { if (wide && C.em) emit_ties(run0, i, maxc); else emit_molecule(C, best, wide ? 0xFFFFFFFFu : nbest); }
                    run0 = i;
                    cur_umi = t.x; cg = t.z;
                    nbest = 1; best[0] = t.z; aggr = t.w; maxc = t.w; wide = false;
                } else {
                    if (t.z == cg) aggr += t.w; else { aggr = t.w; cg = t.z; }
                    if (aggr > maxc) {
                        maxc = aggr;
                        if (!(nbest == 1 && best[0] == t.z)) { nbest = 1; best[0] = t.z; wide = false; }
                    } else if (aggr == maxc) {
                        if (nbest == kMaxGenesPerLabel) wide = true; else best[nbest++] = t.z;
                    }
                }
            }
            if (wide && C.em) emit_ties(run0, nt, maxc); else emit_molecule(C, best, wide ? 0xFFFFFFFFu : nbest);
        }
        if (tid == 0) A.alt[c.cell] = 1;   // used_alternative_strategy, pugutils.rs:1070
        __syncthreads();
    }
    return true;
}

template <int CNT>
__global__ __launch_bounds__(CNT, CNT == 256 ? 4 : 1) void k_p2_cover(P2Args A, const uint32_t* list, const uint32_t* n_list, uint32_t work_lo, uint32_t work_hi, uint32_t* counter) {
    if (A.st->err_code) return;
    if (n_list) work_hi = *n_list;   // (the cells the range-wide build routed to the per-cell kernels: a list on the device)
    __shared__ uint32_t s_cnt[4];
    __shared__ uint32_t s_next;
    __shared__ uint64_t s_mask[2][64];                     // the workgroup cover: uncovered vertices, the round's arborescence
    __shared__ uint32_t s_stage[CNT / 64][64 * kStageRefs];   // labels of 5..16 refs of the components a wave is covering (afq_pug_common.h): 4 KiB per wave
    __shared__ uint32_t s_bestv[CNT / 64], s_bestsz[CNT / 64];
    __shared__ uint32_t s_ws[CNT / 64];
    __shared__ uint32_t s_nt;
    __shared__ unsigned long long s_ebase;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
  for (;;) {
    __syncthreads();
    if (tid == 0) s_next = work_lo + atomicAdd(counter, 1u);
    __syncthreads();
    const uint32_t work = s_next;
    if (work >= work_hi) return;
    const uint32_t j = list[work];
    const uint32_t* d = A.gdesc + kGDescWords * (size_t)j;
    if (!(d[0] & 1u)) continue;   // handed to the one-workgroup kernel, or failed (the error is set)
    const bool defer = (d[0] & 2u) != 0;
    const P2Cell c = A.cells[j];
    if (tid < 3) s_cnt[tid] = d[12 + tid];
    if (tid == 3) s_cnt[3] = 0;
    __syncthreads();
    const PugCtx C = make_ctx(A, c, s_cnt);
    const uint64_t* ch = A.s_h + c.rd_base;
    const uint32_t* coff = A.v_off + c.rd_base;
    auto at = [&](int k) -> const uint32_t* { return A.pool + (((unsigned long long)d[k + 1] << 32) | d[k]); };
    const uint32_t n_pr = d[1], n_tiny = d[2], n_mid = d[3];
    const uint32_t* tl = (d[4] & d[5]) == 0xFFFFFFFFu ? nullptr : at(4);   // (the range-wide build lists slots of the cell, not touched-vertex numbers)
    const uint32_t* pr_v = at(6);
    const uint32_t* mid_off = at(8);
    const uint4* mrec = reinterpret_cast<const uint4*>(at(10));
    p2_cover_pairs<CNT>(C, ch, coff, tl, pr_v, n_pr, s_stage[wv]);
    const uint32_t n_bigc = d[15] & 0x7FFFFFFFu;
    const uint32_t* x = mid_off + n_mid + n_bigc + 1;   // (eight words behind the list's last offset: read only where the per-cell graph kernel listed larger components)
    // The records of these components lie in slot order, not in the reference's: a round with ONE largest arborescence does not
    // depend on the order; at the first round that meets a tie the component is set aside for k_p2_tied (afq_pug_common.h).
    uint32_t* const tied = A.pool + (((unsigned long long)d[17] << 32) | d[16]);
    if (defer) {
        cover_tiny8<CNT / 64, kCoverDefer>(C, mrec, mid_off, n_tiny, wv, lane, tied, tied + 4, s_stage[wv]);
        cover_wave64<CNT / 64, kCoverDefer>(C, mrec, mid_off, n_tiny, n_mid, wv, lane, tied + 1, tied + 4 + 4 * (size_t)n_tiny, s_stage[wv]);
    } else {   // (the graph kernel ordered this cell's components itself)
        cover_tiny8<CNT / 64>(C, mrec, mid_off, n_tiny, wv, lane, nullptr, nullptr, s_stage[wv]);
        cover_wave64<CNT / 64>(C, mrec, mid_off, n_tiny, n_mid, wv, lane, nullptr, nullptr, s_stage[wv]);
    }
    if (n_bigc) {   // 65..4096 vertices: the graph kernel left their adjacency as rows of mask words
        const uint64_t* rows = reinterpret_cast<const uint64_t*>(A.pool + (((unsigned long long)x[1] << 32) | x[0]));
        const uint32_t* rowoff = A.pool + (((unsigned long long)x[3] << 32) | x[2]);
        cover_big<CNT / 64>(C, mrec, mid_off, n_mid, n_bigc, rowoff, rows, s_mask, s_bestv, s_bestsz, wv, lane);
    }
    if (d[15] >> 31) {   // components above --large-graph-thresh
        const uint32_t n_large = x[4];
        const uint32_t* lg_off = A.pool + (((unsigned long long)x[6] << 32) | x[5]);
        __syncthreads();
        if (!cover_large<CNT>(A, C, c, tl, lg_off, lg_off + n_large + 2, n_large, s_ws, &s_nt, &s_ebase)) return;
    }
    gsync();
    if (s_cnt[3]) { if (tid == 0) set_err(A.st, s_cnt[3], c.cell); return; }
    if (tid == 0) {
        A.cell_ncols[c.cell] = s_cnt[0];
        if (A.lab_cnt) { A.lab_cnt[2 * c.cell] = s_cnt[1]; A.lab_cnt[2 * c.cell + 1] = s_cnt[2]; }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// 7. one workgroup per cell once more: the components k_p2_cover set aside because a round met a TIE between equal-size
//    arborescences - there, and only there, the reference's vertex order decides (class by first appearance = smallest record
//    offset of the label in the cell, then UMI; pugutils.rs:1090-1160 takes the first largest arborescence it meets).  Round 4
//    found the class minima for EVERY vertex of every component of three or more vertices inside k_p2_graph - a third to a half of
//    that kernel; a dozenth of the components need them.  Per batch of set-aside components (as many as the LDS table holds
//    classes for): their uncovered vertices' label keys into the table, the cell's vertex slots streamed ONCE through it (the
//    smallest record offset per asked-for class; a vertex under a HASHED key that hits an asked-for class is compared with the
//    one that held the minimum before it - equal keys are thereby shown to be equal labels), then a wave per component puts its
//    records into the reference's order (rank by (class minimum, UMI), adjacency masks and the uncovered mask renumbered).  The
//    covers then RESUME them (kCoverResume) from where they stopped.
// CMIN_ONLY (round 6, the cells of the range-wide graph build): the kernel stops behind the class minima - it leaves, per uncovered
// vertex of a set-aside component, its class's smallest record offset beside the vertex's record (PfDev.cmv) - and the components
// are put in order and resumed by k_pc_resume (afq_pugflat.hip), a lane / eight lanes / a wave each across the whole range.  In
// this kernel that resume was half of a cell's time: three batches of eight components per wave, one behind the other, while
// the other cells of the range waited for the CU (profiles/round6_14).  The other instance (the cells the build routed to the
// per-cell kernels: list, n_list) goes on as before.
template <int CNT, bool CMIN_ONLY>
__global__ __launch_bounds__(CNT) void k_p2_tied(P2Args A, const uint32_t* list, const uint32_t* n_list, uint32_t work_lo, uint32_t work_hi, uint32_t* counter) {
    if (A.st->err_code) return;
    if (n_list) work_hi = *n_list;
    constexpr uint32_t TSlots = CNT >= 1024 ? 8192u : 4096u, TKeys = TSlots * 3 / 8, BloomWords = TSlots / 16;   // (a batch holds < TKeys + 64 classes: load <= 0.4)
    __shared__ unsigned long long t_key[TSlots];
    __shared__ uint32_t t_min[TSlots];
    __shared__ uint32_t s_bloom[BloomWords];
    __shared__ uint32_t s_cnt[4];
    __shared__ uint32_t s_next;
    __shared__ uint32_t s_ws[CNT / 64];
#ifdef AFQ_PUG_TIMING
    __shared__ unsigned long long tmark[8];
    unsigned long long t_keys = 0, t_stream = 0, t_order = 0;
#define T_ADD(acc, code) do { __syncthreads(); const unsigned long long t0__ = wall_clock64(); code; __syncthreads(); acc += wall_clock64() - t0__; } while (0)
#endif
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
  for (;;) {
    __syncthreads();
    if (tid == 0) s_next = work_lo + atomicAdd(counter, 1u);
    __syncthreads();
    const uint32_t work = s_next;
    if (work >= work_hi) return;
    const uint32_t j = list[work];
    const uint32_t* d = A.gdesc + kGDescWords * (size_t)j;
    if (CMIN_ONLY != ((d[0] & 4u) != 0)) continue;   // (the range-wide build's cells / the per-cell graph kernel's: an instance each)
    if ((d[0] & 3u) != 3u) continue;   // nothing was set aside (the graph kernel ordered the cell's components itself), or the cell was handed to the one-workgroup kernel, or failed
    const P2Cell c = A.cells[j];
    auto at = [&](int k) -> uint32_t* { return A.pool + (((unsigned long long)d[k + 1] << 32) | d[k]); };
    const uint32_t n_tiny = d[2];
    const uint32_t* mid_off = at(8);
    uint4* mrec = reinterpret_cast<uint4*>(at(10));
    uint32_t* const tied = A.pool + (((unsigned long long)d[17] << 32) | d[16]);
    const uint32_t nA = tied[0], nB = tied[1], nE = nA + nB;
    if (nE == 0) continue;
    uint32_t* const listA = tied + 4;
    uint32_t* const listB = tied + 4 + 4 * (size_t)n_tiny;
    auto entry = [&](uint32_t e) -> uint32_t* { return e < nA ? listA + 4 * (size_t)e : listB + 4 * (size_t)(e - nA); };
    if (tid == 0) { s_cnt[0] = A.cell_ncols[c.cell]; s_cnt[1] = A.lab_cnt ? A.lab_cnt[2 * c.cell] : 0u; s_cnt[2] = A.lab_cnt ? A.lab_cnt[2 * c.cell + 1] : 0u; s_cnt[3] = 0; }
    __syncthreads();
    PugCtx C = make_ctx(A, c, s_cnt);
    const bool umi_recs = (d[0] & 4u) != 0;   // the range-wide build's records: (UMI, reads) where the per-cell graph kernel's hold the adjacency mask - the covers work the edges out (umi_edge), and a reordered record keeps its two words
    C.adj_umi = umi_recs ? 1u : 0u;
    const uint64_t* ch = A.s_h + c.rd_base;
    const uint64_t* cu = A.s_u + c.rd_base;
    const uint32_t* coff = A.v_off + c.rd_base;
    const uint32_t R = c.R;
    // the entries' running key counts (a key per uncovered vertex) cut the list into batches
    uint32_t n_keys = 0;
    for (uint32_t base = 0; base < nE; base += CNT) {
        const uint32_t e = base + tid;
        uint32_t k = 0;
        if (e < nE) { const uint32_t* en = entry(e); k = (uint32_t)__popc(en[1]) + (uint32_t)__popc(en[2]); }
        uint32_t tot;
        const uint32_t ex = block_excl_scan<CNT>(k, s_ws, tot);
        if (e < nE) entry(e)[3] = n_keys + ex;
        n_keys += tot;
    }
    gsync();
    const uint32_t n_batches = (n_keys + TKeys - 1) / TKeys;
    G_MARK(0);
    auto mix = [](uint64_t h) -> uint32_t { uint32_t v = ((uint32_t)h ^ (uint32_t)(h >> 32)) * 0x9E3779B1u; return v ^ (v >> 15); };
    auto bloom_at = [&](uint32_t mx) -> uint32_t { return (mx >> 16) & (BloomWords * 32 - 1); };
    auto find = [&](uint64_t h, uint32_t mx, bool insert) -> uint32_t {   // slot of h, or 0xFFFFFFFF
        uint32_t slot = mx & (TSlots - 1);
        for (uint32_t step = 0; step < TSlots; ++step, slot = (slot + 1) & (TSlots - 1)) {
            const unsigned long long k = __hip_atomic_load(&t_key[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (k == h) return slot;
            if (k == ~0ull) {
                if (!insert) return 0xFFFFFFFFu;
                const unsigned long long old = atomicCAS(&t_key[slot], ~0ull, (unsigned long long)h);
                if (old == ~0ull || old == h) return slot;
            }
        }
        return 0xFFFFFFFFu;
    };
    for (uint32_t b = 0; b < n_batches; ++b) {
        __syncthreads();
        for (uint32_t i = tid; i < TSlots; i += CNT) { t_key[i] = ~0ull; t_min[i] = 0xFFFFFFFFu; }
        for (uint32_t i = tid; i < BloomWords; i += CNT) s_bloom[i] = 0;
        __syncthreads();
        G_MARK(1);
        // the classes that are asked for: the labels of the batch's uncovered vertices
        for (uint32_t e = tid; e < nE; e += CNT) {
            const uint32_t* en = entry(e);
            if (en[3] / TKeys != b) continue;
            const uint32_t b0 = mid_off[en[0]];
            for (uint64_t m = ((uint64_t)en[2] << 32) | en[1]; m; m &= m - 1) {
                const uint32_t g = mrec[2 * (size_t)(b0 + (uint32_t)__builtin_ctzll(m))].x;
                const uint64_t h = ch[g];
                const uint32_t mx = mix(h);
                atomicOr(&s_bloom[bloom_at(mx) >> 5], 1u << (bloom_at(mx) & 31u));
                if (find(h, mx, true) == 0xFFFFFFFFu) s_cnt[3] = kErrInternal;
            }
        }
        __syncthreads();
        G_MARK(2);
        // every vertex slot of the cell once (slots past a partition's last vertex hold key 0: no class), six per thread and trip
        for (uint32_t g0 = tid; g0 - tid < R; g0 += 6 * CNT) {
            uint64_t h6[6];
            uint32_t sl6[6], off6[6], old6[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) h6[r] = g0 + (uint32_t)r * CNT < R ? ch[g0 + (uint32_t)r * CNT] : 0ull;
            // (the record offsets with the keys, for every slot: asked for only where a key hits the table they were a second round trip
            //  per trip behind the first, in a kernel whose cell is six trips long; the asked-for classes are the common ones anyway)
#pragma unroll
            for (int r = 0; r < 6; ++r) off6[r] = g0 + (uint32_t)r * CNT < R ? coff[g0 + (uint32_t)r * CNT] : 0u;
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const uint32_t mx = mix(h6[r]);
                const bool pass = h6[r] != 0 && ((s_bloom[bloom_at(mx) >> 5] >> (bloom_at(mx) & 31u)) & 1u);
                sl6[r] = pass ? find(h6[r], mx, false) : 0xFFFFFFFFu;
            }
#pragma unroll
            for (int r = 0; r < 6; ++r) old6[r] = sl6[r] != 0xFFFFFFFFu ? atomicMin(&t_min[sl6[r]], off6[r]) : 0xFFFFFFFFu;
#pragma unroll
            for (int r = 0; r < 6; ++r)
                if (sl6[r] != 0xFFFFFFFFu && (uint32_t)(h6[r] >> 62) == 3 && old6[r] != 0xFFFFFFFFu && old6[r] != off6[r] &&
                    !lab_equal(rec_label(C, off6[r]), rec_label(C, old6[r]))) s_cnt[3] = kErrLabelHash;
        }
        __syncthreads();
        G_MARK(3);
        if constexpr (CMIN_ONLY) {   // the minima to where k_pc_resume reads them: beside the records
            uint32_t* cmv = A.pool + A.pfd->cmv;
            for (uint32_t e = tid; e < nE; e += CNT) {
                const uint32_t* en = entry(e);
                if (en[3] / TKeys != b) continue;
                const uint32_t b0 = mid_off[en[0]];
                for (uint64_t m = ((uint64_t)en[2] << 32) | en[1]; m; m &= m - 1) {
                    const uint32_t at = b0 + (uint32_t)__builtin_ctzll(m);
                    const uint64_t h = ch[mrec[2 * (size_t)at].x];
                    const uint32_t slot = find(h, mix(h), false);
                    const uint32_t mn = slot == 0xFFFFFFFFu ? 0xFFFFFFFFu : t_min[slot];
                    if (mn >> 31) s_cnt[3] = kErrInternal;   // (every asked-for class has at least the vertex that asked; record offsets are dword offsets inside a chunk: below 2^30)
                    cmv[at] = mn;
                }
            }
            continue;
        }
        // The batch's components, their records into the reference's order.  An order key: the vertices still uncovered by
        // (class minimum, UMI), the covered ones behind them as they lie.  Components of up to eight vertices (the list of
        // cover_tiny8) EIGHT to a wave, a group of eight lanes each - a wave to each was a chain of five dependent round trips
        // per component, two dozen components one after the other per wave: most of this kernel's time.
        auto order_key = [&](bool act, uint32_t pos, uint64_t uc, uint32_t g) -> uint64_t {
            if (!act) return ~0ull;
            if (!((uc >> pos) & 1ull)) return (1ull << 63) | pos;
            const uint64_t h = ch[g];
            const uint32_t slot = find(h, mix(h), false);
            const uint32_t mn = slot == 0xFFFFFFFFu ? 0xFFFFFFFFu : t_min[slot];
            if (mn >> 31) s_cnt[3] = kErrInternal;   // (every asked-for class has at least the vertex that asked; record offsets are dword offsets inside a chunk: below 2^30)
            return ((uint64_t)(mn & 0x7FFFFFFFu) << 32) | (uint32_t)(cu[g] >> 32);
        };
        {
            const uint32_t gl = lane & 7u, gbase = lane & ~7u, grp = lane >> 3;
            for (uint32_t e0 = wv * 8; e0 < nA; e0 += (CNT / 64) * 8) {   // (uniform per wave)
                const uint32_t e = e0 + grp;
                uint32_t* en = listA + 4 * (size_t)(e < nA ? e : 0u);
                const bool mine = e < nA && en[3] / TKeys == b;
                const uint32_t ci = mine ? en[0] : 0u;
                const uint32_t uc = mine ? en[1] & 0xFFu : 0u;
                const uint32_t b0 = mine ? mid_off[ci] : 0u, n = mine ? mid_off[ci + 1] - b0 : 0u;
                const bool act = gl < n;
                uint4 qa = make_uint4(0, 0, 0, 0), qb = qa;
                if (act) { qa = mrec[2 * (size_t)(b0 + gl)]; qb = mrec[2 * (size_t)(b0 + gl) + 1]; }
                const uint64_t key = order_key(act, gl, uc, qa.x);
                uint32_t rank = 0;
#pragma unroll
                for (uint32_t k = 0; k < 8; ++k) {
                    const uint64_t kk = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(key >> 32), (int)(gbase + k)) << 32) | (uint32_t)__shfl((int)(uint32_t)key, (int)(gbase + k));
                    rank += k < n && kk < key ? 1u : 0u;
                    if (act && k < n && k != gl && kk == key) s_cnt[3] = kErrInternal;   // (two vertices of one component with the same class and UMI: cannot be)
                }
                const uint32_t adj = qb.z & 0xFFu;
                uint32_t nadj = 0, nuc = 0;
#pragma unroll
                for (uint32_t k = 0; k < 8; ++k) {
                    const uint32_t rk = (uint32_t)__shfl((int)rank, (int)(gbase + k));
                    if (k < n && ((adj >> k) & 1u)) nadj |= 1u << rk;
                    if (k < n && ((uc >> k) & 1u)) nuc |= 1u << rk;
                }
                if (act) {   // (every lane holds its record in registers: the slots can be overwritten)
                    mrec[2 * (size_t)(b0 + rank)] = qa;
                    mrec[2 * (size_t)(b0 + rank) + 1] = umi_recs ? qb : make_uint4(qb.x, qb.y, nadj, 0u);
                }
                if (mine && gl == 0) { en[1] = nuc; en[2] = 0u; }
            }
        }
        for (uint32_t e = wv; e < nB; e += CNT / 64) {   // 9..64 vertices: a wave each (uniform per wave)
            uint32_t* en = listB + 4 * (size_t)e;
            if (en[3] / TKeys != b) continue;
            const uint32_t ci = en[0];
            const uint64_t uc = ((uint64_t)en[2] << 32) | en[1];
            const uint32_t b0 = mid_off[ci], n = mid_off[ci + 1] - b0;
            const bool act = lane < n;
            uint4 qa = make_uint4(0, 0, 0, 0), qb = qa;
            if (act) { qa = mrec[2 * (size_t)(b0 + lane)]; qb = mrec[2 * (size_t)(b0 + lane) + 1]; }
            const uint64_t key = order_key(act, lane, uc, qa.x);
            uint32_t rank = 0;
            for (uint32_t k = 0; k < n; ++k) {
                const uint64_t kk = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(key >> 32), (int)k) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, (int)k);
                rank += kk < key ? 1u : 0u;
                if (kk == key && k != lane && act) s_cnt[3] = kErrInternal;
            }
            const uint64_t adj = ((uint64_t)qb.w << 32) | qb.z;
            uint64_t nadj = 0, nuc = 0;
            for (uint32_t k = 0; k < n; ++k) {
                const uint32_t rk = (uint32_t)__builtin_amdgcn_readlane((int)rank, (int)k);
                if ((adj >> k) & 1ull) nadj |= 1ull << rk;
                if ((uc >> k) & 1ull) nuc |= 1ull << rk;
            }
            if (act) {
                mrec[2 * (size_t)(b0 + rank)] = qa;
                mrec[2 * (size_t)(b0 + rank) + 1] = umi_recs ? qb : make_uint4(qb.x, qb.y, (uint32_t)nadj, (uint32_t)(nadj >> 32));
            }
            if (lane == 0) { en[1] = (uint32_t)nuc; en[2] = (uint32_t)(nuc >> 32); }
        }
    }
    gsync();
    if (s_cnt[3]) { if (tid == 0) set_err(A.st, s_cnt[3], c.cell); return; }
    if constexpr (CMIN_ONLY) continue;
    G_MARK(4);
    uint32_t* const stage = reinterpret_cast<uint32_t*>(t_key) + (size_t)wv * 64 * kStageRefs;   // (the class table is dead: 4 KiB of it per wave stage the labels)
    cover_tiny8<CNT / 64, kCoverResume>(C, mrec, mid_off, nA, wv, lane, nullptr, listA, stage);
    cover_wave64<CNT / 64, kCoverResume>(C, mrec, mid_off, 0u, nB, wv, lane, nullptr, listB, stage);
    gsync();
    G_MARK(5);
#ifdef AFQ_PUG_TIMING
    if (tid == 0 && (work % 512) < 2) {
        auto us = [&](int a, int b) { return (double)(tmark[b] - tmark[a]) / 100.0; };
        printf("p2 tied cell R=%u entries=%u+%u keys=%u batches=%u: init=%.0f keys=%.0f stream=%.0f order=%.0f resume=%.0f total=%.0f us (last batch's marks)\n",
               R, nA, nB, n_keys, n_batches, us(0, 1), us(1, 2), us(2, 3), us(3, 4), us(4, 5), us(0, 5));
    }
#endif
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
    AFQ_LAUNCH((k_p2_scatter<512>), (a.n_tiles + 127) / 128 * 128, 512, s, a);   // (tiles of kP2TileHost = 8 x 512 reads)
}
// 8192 persistent workgroups of four waves walk the partitions (a wave per partition as its own workgroup cost the lone-vertex
// kernel half its time in dispatch; 4096 and 16384 are within 1 ms); the grid is a multiple of 2048 (for_each_partition_in_runs)
static uint32_t p2_grid(uint32_t n_parts) { const uint32_t full = (n_parts + 3) / 4; return ((full < 8192u ? full : 8192u) + 2047) / 2048 * 2048; }
void launch_p2_part(hipStream_t s, const P2Args& a) { if (a.n_parts) AFQ_LAUNCH(k_p2_part, p2_grid(a.n_parts), 256, s, a); }
void launch_p2_search(hipStream_t s, const P2Args& a) {
    if (!a.n_parts) return;
    AFQ_LAUNCH(k_p2_search, p2_grid(a.n_parts), 256, s, a);
#ifdef AFQ_SEARCH_TIMING
    hipLaunchKernelGGL(k_search_timing_dump, dim3(1), dim3(1), 0, s);
#endif
    AFQ_LAUNCH(k_p2_search_over, std::min((a.n_parts + 255) / 256, 2048u), 256, s, a);   // (the partitions with more pairs than slots, normally none: two counts per partition are read)
    launch_p2_check(s, a);   // (the candidates that fail the label test cleared, the end points of the others flagged: afq_pugflat.hip)
}
void launch_p2_lone(hipStream_t s, const P2Args& a) {
    if (!a.n_parts) return;
    if (AFQ_LONE_FLAT) {
        if (a.lone_coop >= 2) AFQ_LAUNCH(k_pl_lone<true>, a.n_tiles, 256, s, a);
        else AFQ_LAUNCH(k_pl_lone<false>, a.n_tiles, 256, s, a);
    } else if (a.lone_coop >= 2) AFQ_LAUNCH(k_p2_lone<true>, p2_grid(a.n_parts), 256, s, a);
    else AFQ_LAUNCH(k_p2_lone<false>, p2_grid(a.n_parts), 256, s, a);
#ifdef AFQ_LONE_TIMING
    hipLaunchKernelGGL(k_lone_timing_dump, dim3(1), dim3(1), 0, s);
#endif
}
void launch_p2_graph(hipStream_t s, const P2Args& a, uint64_t n_reads) {
    if (!a.n_cells) return;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    // The cells come largest first.  In the per-cell kernels a cell is one workgroup's from its first phase to its last - a chain of
    // dependent steps, whatever its size - and a range's kernel is not over before its largest cell is (a 290 k-read cell: 15 ms at 256
    // threads, twice what the rest of its range takes on the whole chip): the first n_big cells - 15 000 reads or more - get 1024
    // threads each, a CU to themselves, in launches of their own; the others 256 threads, four (graph: 147 VGPRs, 48.5 KiB of LDS),
    // four (cover) and three (ties: 49 KiB) workgroups to a CU.  (Round 5 measured 512-thread workgroups at 128 VGPRs, two cells to a
    // CU, for the cells of 15 000...60 000 / ...100 000 reads: graph + cover + ties 21.8 / 24.1 ms per configs[2] step against 20.4.)
    // Work counters: graph 0 / 2, cover 1 / 3, ties 4 / 5 (256 / 1024 threads).
    const uint32_t n_big = a.n_big < a.n_cells ? a.n_big : a.n_cells, rest = a.n_cells - n_big, ucus = (uint32_t)cus;
    uint32_t* const wc = a.work_counter;
    if (a.graph_flat) {
        // Round 6: the graph phase is range-wide flat kernels (afq_pugflat.hip); the per-cell graph kernel only takes the cells that
        // build routes to it - a component of more than 64 vertices, whose order and adjacency rows it makes - off a list on the device.
        launch_pf_build(s, a, n_reads);
        launch_pf_cover(s, a);   // (the covers of every cell the build did not route away: a workgroup per tile)
        AFQ_LAUNCH(k_p2_graph<1024>, std::min(a.n_cells, ucus), 1024, s, a, a.old_list, &a.pfd->n_old, 0u, 0u, wc + 2);
        AFQ_LAUNCH(k_p2_cover<1024>, std::min(a.n_cells, ucus), 1024, s, a, a.old_list, &a.pfd->n_old, 0u, 0u, wc + 3);
        // the ties: the class minima per cell (the first half of k_p2_tied), order and resume range-wide; the routed cells as before
        if (n_big) AFQ_LAUNCH((k_p2_tied<1024, true>), std::min(n_big, ucus), 1024, s, a, a.order, (const uint32_t*)nullptr, 0u, n_big, wc + 5);
        if (rest) AFQ_LAUNCH((k_p2_tied<kGNT, true>), std::min(rest, 3 * ucus), kGNT, s, a, a.order, (const uint32_t*)nullptr, n_big, a.n_cells, wc + 4);
        launch_pf_resume(s, a);
        AFQ_LAUNCH((k_p2_tied<1024, false>), std::min(a.n_cells, ucus), 1024, s, a, a.old_list, &a.pfd->n_old, 0u, 0u, wc + 6);
        return;
    } else {
        if (n_big) AFQ_LAUNCH(k_p2_graph<1024>, std::min(n_big, ucus), 1024, s, a, a.order, (const uint32_t*)nullptr, 0u, n_big, wc + 2);
        if (rest) AFQ_LAUNCH(k_p2_graph<kGNT>, std::min(rest, 4 * ucus), kGNT, s, a, a.order, (const uint32_t*)nullptr, n_big, a.n_cells, wc);
        if (n_big) AFQ_LAUNCH(k_p2_cover<1024>, std::min(n_big, ucus), 1024, s, a, a.order, (const uint32_t*)nullptr, 0u, n_big, wc + 3);
        if (rest) AFQ_LAUNCH(k_p2_cover<kGNT>, std::min(rest, 4 * ucus), kGNT, s, a, a.order, (const uint32_t*)nullptr, n_big, a.n_cells, wc + 1);
    }
    // ... and the components the covers set aside at a tie, in the reference's order
    if (n_big) AFQ_LAUNCH((k_p2_tied<1024, false>), std::min(n_big, ucus), 1024, s, a, a.order, (const uint32_t*)nullptr, 0u, n_big, wc + 5);
    if (rest) AFQ_LAUNCH((k_p2_tied<kGNT, false>), std::min(rest, 3 * ucus), kGNT, s, a, a.order, (const uint32_t*)nullptr, n_big, a.n_cells, wc + 4);
}

}  // namespace afq
