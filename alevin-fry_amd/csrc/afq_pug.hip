// afq_pug.hip — parsimony resolution on gfx950: per-cell equivalence-class map, parsimonious UMI
// graph (PUG), weakly connected components and the greedy monochromatic-arborescence cover.
//
// Replaces, for one cell (reference paths relative to /root/reference):
//   EqMap::init_from_chunk               src/eq_class.rs:823-1036   (classes by exact ref list, (umi,count) per class)
//   extract_graph / has_edge             src/pugutils.rs:65-267, src/utils.rs:389-393
//   weakly_connected_components          src/pugutils.rs:278-301
//   collapse_vertices                    src/pugutils.rs:308-391
//   get_num_molecules (+large component) src/pugutils.rs:989-1331, 916-982
// Semantics: SURVEY.md appendix B.3-B.5.  Tie-break between equal-size arborescences = ascending
// vertex id in the reference's vertex numbering (class-major, class id = first appearance, UMI
// rank inside the class) - the oracle's canonical order; the reference itself walks a hash set.
//
// One 1024-thread workgroup owns one cell and runs every phase out of a per-cell slice of global
// scratch (L2 resident), so nothing here needs a device-scope atomic except the edge-pool bump.
// Most components are singletons (lane-parallel), small ones (<= 64 vertices) are covered by one
// wave with the adjacency as one 64-bit mask per lane, larger ones (<= 4096) by the workgroup with
// multi-word masks (one mask word per lane, candidate vertices spread over the waves).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "afq_common.h"
#include "afq_kernels.h"
#include "afq_prims.h"
#include "afq_pug_common.h"

namespace afq {

#ifdef AFQ_PUG_TIMING
#define PUG_MARK(i) do { __syncthreads(); if (threadIdx.x == 0 && blockIdx.x < 4) tmark[i] = wall_clock64(); } while (0)
#define PUG_ACC(i) do { __syncthreads(); if (threadIdx.x == 0 && blockIdx.x < 4) { const unsigned long long t_ = wall_clock64(); tacc[i] += t_ - tlast; tlast = t_; } } while (0)
#else
#define PUG_MARK(i) do {} while (0)
#define PUG_ACC(i) do {} while (0)
#endif

constexpr int kPugNT = 1024;
constexpr uint32_t kVidBits = 20;                 // vertices per cell < 2^20

constexpr uint32_t kMaxBigComp = 4096;            // vertices of a component the multi-word cover handles
// Neighbour search out of LDS: the cell's vertices are cut into P = 4^k partitions by the low 2k bits of their UMI and the
// partitions are taken one at a time through a hash table that lives in the 128 KiB LDS block (key = UMI, value = vertex id |
// reads | class).  A 1-mismatch probe of a vertex stays in the vertex's own partition unless it changes one of the low k
// bases, and then it lands in exactly one other partition - so every pass knows which vertices can have a neighbour in it.
constexpr uint32_t kTabSlots = 8192;              // 64 KiB of keys + 64 KiB of values
constexpr uint32_t kPartTarget = 4500;            // planned vertices per partition (a pass has a fixed cost in latency: few, well-filled ones)
constexpr uint32_t kPartMax = 5400;               // a partition above this (skewed UMIs) sends the cell down the global-memory route
constexpr uint32_t kMaxParts = 1024;

// A read as the grouping sort sees it: label key, then UMI and the record it came from in one word - umi << 32 | record
// dword offset when the UMI field is 4 bytes wide (UMIs up to 16 nt: every 10x chemistry), umi << 20 | read index
// otherwise (the offset is then looked up).  Sorted by (h, uo): classes contiguous, UMIs ascending inside a class.
// In memory the record is one little-endian 128-bit integer h << 64 | uo: the sort compares and moves it as that scalar.
struct __attribute__((aligned(16))) SortRec {
    uint64_t uo, h;
};
typedef unsigned __int128 u128;
__device__ __forceinline__ u128 rec_key(const SortRec& r) { return ((u128)r.h << 64) | r.uo; }

// Sample sort of one cell's reads by one workgroup.  A bitonic network over the whole cell costs (log2 n)^2 / 2 passes;
// here the reads are split on sampled keys into buckets of ~2048 (one counting pass, one scatter pass) and every bucket
// is sorted once, in registers (reg_bitonic_sort: cross-lane stages on the VALU, a handful of trips through LDS).
// Keys are unique (uo ends in the read's offset / index), so the splitters always separate.  A bucket over 4096
// records (skewed sample) falls back to the network.  tile: LDS for 4096 records; aux: LDS, 6 * 1024 + 2 words.
constexpr uint32_t kSortBucket = 2048, kSortWaveBucket = 96, kSortWaveMax = 384, kSortTile = 4096, kSortMaxBuckets = 1024;
// (Not inlined: inside the cell kernel its register-resident tiles would share one allocation with everything that is
// live across the sort there, and spill in the inner loops.)
#ifdef AFQ_PUG_TIMING
#define SORT_MARK(i) do { __syncthreads(); if (threadIdx.x == 0 && blockIdx.x < 4 && tm) tm[i] = wall_clock64(); } while (0)
#else
#define SORT_MARK(i) do {} while (0)
#endif
template <int NT>
__device__ __noinline__ void sample_sort_reads(SortRec* sr, uint32_t* bid, uint32_t R, const uint64_t* rd_h, const uint64_t* rd_u,
                                               const uint32_t* rd_o, bool wide_umi, SortRec* tile, uint32_t* aux, uint32_t* s_ws,
                                               [[maybe_unused]] unsigned long long* tm = nullptr) {
    const uint32_t tid = threadIdx.x;
    auto load = [&](uint32_t i) {
        SortRec r;
        r.h = rd_h[i];
        r.uo = wide_umi ? (rd_u[i] << kVidBits) | i : rd_u[i];   // (4-byte UMIs: the decode already wrote umi << 32 | record offset)
        return r;
    };
    const u128 sentinel = ~(u128)0;
    u128* tile128 = reinterpret_cast<u128*>(tile);
    auto sort_tile = [&](SortRec* a, uint32_t n) {   // n <= kSortTile, in place
        u128* a128 = reinterpret_cast<u128*>(a);
        if (n <= NT) block_sort_to_lds<NT, 1, u128>(a128, n, tile128, sentinel);
        else if (n <= 2 * NT) block_sort_to_lds<NT, 2, u128>(a128, n, tile128, sentinel);
        else block_sort_to_lds<NT, 4, u128>(a128, n, tile128, sentinel);
        for (uint32_t i = tid; i < n; i += NT) a128[i] = tile128[i];
        __syncthreads();
    };
    if (R <= kSortTile) {
        for (uint32_t i = tid; i < R; i += NT) sr[i] = load(i);
        __syncthreads();
        sort_tile(sr, R);
        return;
    }
    // Buckets sorted by ONE wave each in its registers (no LDS, no barriers), in the smallest network that holds the bucket:
    // 128, 256 or 512 slots.  A bitonic network costs log2(n)^2/2 stages per slot whether the slot holds a record or a
    // sentinel, so the buckets are planned small (~96 reads: 28 stages in the 128-slot network, three quarters full - the
    // 512-slot one, half full, was 2.4 x the work per read); cells too big for 1024 such buckets get proportionally larger
    // ones, and beyond ~400 k reads ~2048 per bucket, sorted by the workgroup.
    const bool wave_buckets = R <= kSortWaveMax * kSortMaxBuckets;
    uint32_t target = wave_buckets ? kSortWaveBucket : kSortBucket;
    if (wave_buckets && (uint64_t)target * kSortMaxBuckets < R) target = (R + kSortMaxBuckets - 1) / kSortMaxBuckets;
    uint32_t nb = (R + target - 1) / target;
    nb = nb > kSortMaxBuckets ? kSortMaxBuckets : nb;
    const uint32_t ns = 64 * nb < kSortTile ? 64 * nb : kSortTile;
    SortRec* spl = reinterpret_cast<SortRec*>(aux);            // [nb - 1] splitters (4 words each)
    uint32_t* cnt = aux + 4 * kSortMaxBuckets;                  // [nb] bucket sizes, then fill cursors
    uint32_t* off = cnt + kSortMaxBuckets;                      // [nb + 1]
    uint32_t* flag = off + kSortMaxBuckets + 1;
    for (uint32_t i = tid; i < ns; i += NT) sr[i] = load((uint32_t)(((uint64_t)i * R) / ns));
    for (uint32_t i = tid; i < nb; i += NT) cnt[i] = 0;
    if (tid == 0) *flag = 0;
    __syncthreads();
    SORT_MARK(0);
    sort_tile(sr, ns);
    for (uint32_t j = tid; j + 1 < nb; j += NT) spl[j] = sr[(uint32_t)(((uint64_t)(j + 1) * ns) / nb)];
    __syncthreads();
    SORT_MARK(1);
    for (uint32_t i0 = tid; i0 < R; i0 += 4 * NT) {   // four reads per thread and trip: their loads go out together
        SortRec r4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) r4[j] = i0 + j * NT < R ? load(i0 + j * NT) : SortRec{0, 0};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t i = i0 + j * NT;
            if (i >= R) continue;
            uint32_t lo = 0, hi = nb - 1;   // bucket = number of splitters below r
            const u128 rk = rec_key(r4[j]);
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (rk > rec_key(spl[mid])) lo = mid + 1; else hi = mid; }
            bid[i] = lo;
            atomicAdd(&cnt[lo], 1u);
        }
    }
    __syncthreads();
    SORT_MARK(2);
    {
        const uint32_t c = tid < nb ? cnt[tid] : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan<NT>(c, s_ws, tot);
        if (tid < nb) { off[tid] = ex; cnt[tid] = 0; if (c > kSortTile) *flag = 1; }
        if (tid == 0) off[nb] = tot;
        __syncthreads();
    }
    const bool fallback = *flag != 0;
    if (fallback) {   // a bucket the tile cannot hold: the plain network over the whole cell
        for (uint32_t i = tid; i < R; i += NT) sr[i] = load(i);
        __syncthreads();
        tiled_bitonic_sort_by<NT, kSortTile>(reinterpret_cast<u128*>(sr), R, [](u128 a, u128 b) { return a > b; }, tile128);
        return;
    }
    for (uint32_t i0 = tid; i0 < R; i0 += 4 * NT) {
        SortRec r4[4];
        uint32_t b4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const uint32_t i = i0 + j * NT; r4[j] = i < R ? load(i) : SortRec{0, 0}; b4[j] = i < R ? bid[i] : 0u; }
#pragma unroll
        for (int j = 0; j < 4; ++j) if (i0 + j * NT < R) sr[off[b4[j]] + atomicAdd(&cnt[b4[j]], 1u)] = r4[j];
    }
    __syncthreads();
    SORT_MARK(3);
    if (wave_buckets) {
        u128* a128 = reinterpret_cast<u128*>(sr);
        const uint32_t lane = lane_id();
        auto sort_bucket = [&](auto Etag, uint32_t o, uint32_t n) {
            constexpr int E = decltype(Etag)::value;
            u128 a[E];
#pragma unroll
            for (int h = 0; h < E; ++h) a[h] = (uint32_t)(h * 64) + lane < n ? a128[o + h * 64 + lane] : sentinel;
            wave_bitonic_sort<E, u128>(a);
#pragma unroll
            for (int h = 0; h < E; ++h) if ((uint32_t)(h * 64) + lane < n) a128[o + h * 64 + lane] = a[h];
        };
        for (uint32_t b = tid >> 6; b < nb; b += NT / 64) {
            const uint32_t o = off[b], n = off[b + 1] - o;
            if (n < 2 || n > 512) continue;   // oversize buckets (a skewed sample) are left to the workgroup below
            if (n <= 128) sort_bucket(std::integral_constant<int, 2>{}, o, n);
            else if (n <= 256) sort_bucket(std::integral_constant<int, 4>{}, o, n);
            else sort_bucket(std::integral_constant<int, 8>{}, o, n);
        }
        __syncthreads();
        SORT_MARK(4);
        for (uint32_t b = 0; b < nb; ++b) if (off[b + 1] - off[b] > 512) sort_tile(sr + off[b], off[b + 1] - off[b]);
        return;
    }
    for (uint32_t b = 0; b < nb; ++b) sort_tile(sr + off[b], off[b + 1] - off[b]);
}



__global__ __launch_bounds__(kPugNT) void k_pug_cell(PugCellArgs A) {
    __shared__ uint32_t s_ws[kPugNT / 64];
    __shared__ uint32_t s_cnt[4];
    __shared__ uint32_t s_flag[2];
    __shared__ unsigned long long s_ebase;
    __shared__ uint64_t s_mask[4][64];   // multi-word cover: UC, best, scratch
    // 2^20-bit presence filter over the cell's UMIs (128 KiB of the CU's 160 KiB LDS): almost every one of the
    // 1 + 3L neighbour probes of a vertex is a UMI that does not occur in the cell and is rejected here
    // without touching the sorted vertex array.
    __shared__ __attribute__((aligned(16))) uint32_t s_big[1u << 15];   // 128 KiB: sort tiles first, the filter later
    uint32_t* s_bloom = s_big;
    __shared__ uint32_t s_bestv[kPugNT / 64], s_bestsz[kPugNT / 64];
#ifdef AFQ_PUG_TIMING
    __shared__ unsigned long long tmark[24];
    __shared__ unsigned long long tacc[8];
    __shared__ unsigned long long tsort[8];
    __shared__ unsigned long long tlast;
#endif
    __shared__ uint32_t s_next;
    __shared__ uint32_t s_poff[kMaxParts + 1];
    __shared__ uint32_t s_filt[2048];   // 2^16-bit presence filter over the UMIs of the partition in the table
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
  // Persistent workgroup: takes the next cell of the (largest-first) list until the list is empty.  Its scratch
  // slice is reused cell after cell, so the working set of a CU stays the size of ONE cell instead of wandering
  // over a slice per cell (the random probes of the edge phase were paying a TLB miss each).
  for (;;) {
    __syncthreads();
    if (tid == 0) s_next = atomicAdd(A.work_counter, 1u);
    __syncthreads();
    const uint32_t work = s_next;
    if (work >= (A.n_pug_dev ? *A.n_pug_dev : A.n_pug)) return;
    const uint32_t cell = A.pug_cells[work];
    const CellMeta m = A.meta[cell];
    const uint32_t R = m.nrec;
    if (tid < 4) s_cnt[tid] = 0;
    if (tid < 2) s_flag[tid] = 0;
    __syncthreads();
    PugCtx C;
    C.W = reinterpret_cast<const uint32_t*>(A.bytes + m.chunk_off);
    C.HW = A.hw; C.t2g = A.t2g; C.ref_count = A.ref_count; C.num_genes = A.num_genes;
    C.usa = A.usa; C.num_rows = A.num_rows; C.uo = A.num_rows / 3; C.ao = 2 * (A.num_rows / 3); C.em = A.em;
    C.exact_umi = A.exact_umi; C.large_thresh = A.large_thresh; C.umi_pairs = A.umi_pairs; C.gene_level = A.gene_level;
    C.cols = reinterpret_cast<uint32_t*>(A.keys0 + m.key_off);
    C.cols_cap = 2 * m.n_ref + 2;
    C.labw = A.lab ? A.lab + 2 * m.key_off : nullptr;
    C.labd = A.lab ? C.labw + m.n_ref + 1 : nullptr;
    C.lab_cap = m.n_ref + 1;
    C.s_cnt = s_cnt; C.st = A.st; C.cell = cell; C.adj_umi = 0;
    if (A.cell_nkeys[cell] != R) { if (tid == 0) set_err(A.st, kErrRecordWalk, cell); return; }
    if (R >= (1u << kVidBits)) { if (tid == 0) set_err(A.st, kErrPugLimit, cell); return; }

    // ---- scratch carve (u32 words; see pug_scratch_words) ----
    uint32_t* p = A.scratch + A.scr_stride * blockIdx.x;
    SortRec* sr = reinterpret_cast<SortRec*>(p); p += 6 * (size_t)R;           // slab A
    uint64_t* v_umi = reinterpret_cast<uint64_t*>(p); p += 2 * (size_t)R;      // slab B (hash order)
    uint32_t* v_cnt = p; p += R;
    uint32_t* v_cls = p; p += R;
    uint64_t* vv_umi = reinterpret_cast<uint64_t*>(p); p += 2 * (size_t)R;     // slab C (by vertex id)
    uint4* vv = reinterpret_cast<uint4*>(p); p += 4 * (size_t)R;   // per vertex, one 16-byte record = one random access: .x reads, .y class, .z a record (dword offset) carrying its label
    uint32_t* c_vstart = p; p += R + 1;    // slab D (classes, hash order)
    uint32_t* c_minoff = p; p += R;
    uint32_t* c_rep = p; p += R;
    uint32_t* c_base = p; p += R;
    uint32_t* c_order = p; p += R;
    uint32_t* deg = p; p += R + 2;         // out-degree, then edge offsets
    uint32_t* comp_start = p; p += R + 2;
    uint32_t ht_cap = 64;
    while (ht_cap < 2 * R) ht_cap <<= 1;
    p += (p - A.scratch) & 1;              // 8-byte align
    unsigned long long* htab = reinterpret_cast<unsigned long long*>(p); p += 2 * (size_t)ht_cap;   // vertices by UMI (open addressing)
    uint32_t* c_goff = p; p += A.gene_level ? R + 2 : 0;           // gene-level: class -> offset of its gene list
    uint32_t* c_glab = p; p += A.gene_level ? m.n_ref + 2 : 0;      // gene-level: the sorted distinct gene lists
    // aliases into slab A once the sorted reads are consumed
    uint64_t* us = reinterpret_cast<uint64_t*>(sr);                    // 2R : (umi << 20 | vid), sorted
    uint64_t* comp_sorted = us + R;                                    // 2R : (root << 20 | vid), sorted
    uint32_t* wlab = reinterpret_cast<uint32_t*>(comp_sorted + R);     // R  : WCC label
    uint32_t* local_idx = wlab + R;                                    // R  : vid -> index inside its component

    PUG_MARK(0);
    // ---- 1. reads sorted by (label key, umi, offset) ----
    const bool wide_umi = A.umi32 == 0;
    const uint64_t rd_base = A.rd.rd_off[cell];
    auto rec_umi = [&](const SortRec& r) -> uint64_t { return wide_umi ? r.uo >> kVidBits : r.uo >> 32; };
    auto rec_off = [&](const SortRec& r) -> uint32_t { return wide_umi ? A.rd.o[rd_base + (r.uo & ((1u << kVidBits) - 1))] : (uint32_t)r.uo; };
    {
        if (wide_umi) {   // the UMI has to leave room for the 20-bit read index
            for (uint32_t i = tid; i < R; i += kPugNT) if (A.rd.u[rd_base + i] >> (64 - kVidBits)) s_cnt[3] = kErrPugLimit;
            __syncthreads();
            if (s_cnt[3]) { if (tid == 0) set_err(A.st, s_cnt[3], cell); return; }
        }
        uint32_t* bid = reinterpret_cast<uint32_t*>(sr) + 4 * (size_t)R;   // slab A has 6R words, the records take 4R
        sample_sort_reads<kPugNT>(sr, bid, R, A.rd.h + rd_base, A.rd.u + rd_base, A.rd.o + rd_base, wide_umi,
                                  reinterpret_cast<SortRec*>(s_big), s_big + 4 * kSortTile, s_ws
#ifdef AFQ_PUG_TIMING
                                  , tsort
#endif
                                  );
    }
    PUG_MARK(1);
    // ---- 2. vertices = distinct (label, umi); classes = distinct labels ----
    // One pass over the sorted reads: vertex / class heads (two scans), the class's smallest record offset (its first
    // appearance in the file), and the check that equal non-exact keys are equal labels - each such read against the read
    // before it, equality chaining through the class.  A vertex's multiplicity is the distance to the next vertex head.
    auto gene_list = [&](uint32_t rec_dw, uint32_t* g) -> uint32_t {
        const Lab l = rec_label(C, rec_dw);
        uint32_t k = 0;
        for (uint32_t j = 0; j < l.n; ++j) {
            const uint32_t gid = C.t2g[l.p[j] & 0x7FFFFFFFu];
            uint32_t q = 0;
            while (q < k && g[q] < gid) ++q;
            if (q < k && g[q] == gid) continue;
            if (k == kMaxGenesPerLabel) return 0xFFFFFFFFu;
            for (uint32_t r = k; r > q; --r) g[r] = g[r - 1];
            g[q] = gid;
            ++k;
        }
        return k;
    };
    // the same for a read with more distinct genes than kMaxGenesPerLabel (a large gene family; rare): counted, compared and
    // written by looking back over the refs instead of through a register array
    auto gene_first_occurrences = [&](uint32_t rec_dw, auto&& f) {
        const Lab l = rec_label(C, rec_dw);
        for (uint32_t j = 0; j < l.n; ++j) {
            const uint32_t gj = C.t2g[l.p[j] & 0x7FFFFFFFu];
            bool first = true;
            for (uint32_t q = 0; q < j && first; ++q) first = C.t2g[l.p[q] & 0x7FFFFFFFu] != gj;
            if (first) f(gj);
        }
    };
    auto gene_subset_slow = [&](uint32_t a_dw, uint32_t b_dw) -> bool {   // every gene of read a is a gene of read b
        const Lab la = rec_label(C, a_dw), lb = rec_label(C, b_dw);
        for (uint32_t j = 0; j < la.n; ++j) {
            const uint32_t gj = C.t2g[la.p[j] & 0x7FFFFFFFu];
            bool found = false;
            for (uint32_t q = 0; q < lb.n && !found; ++q) found = C.t2g[lb.p[q] & 0x7FFFFFFFu] == gj;
            if (!found) return false;
        }
        return true;
    };
    // Per class a 19-bit signature of its label - bit (id mod 19) for every ref (gene at gene level) in it.  Two labels whose
    // signatures do not meet share no ref, so the neighbour search can drop such a pair on the spot (same-UMI vertices under
    // the transcripts of one gene are the bulk of its matches, and ids that close never collide mod 19); the others still
    // get the exact comparison.  Labels of one or two ids are carried in the sort key itself: no read needed for them.
    uint32_t* c_sig = comp_start;   // (free until the components are listed)
    // Labels of one or two refs - nine in ten - go into the vertex record itself (phase 3): whoever needs a vertex's label
    // then has it with the one 16-byte access it makes anyway, instead of chasing the record offset into the chunk.
    uint32_t* c_r0 = reinterpret_cast<uint32_t*>(htab);   // (the table's space is free until the global route, if taken at all)
    uint32_t* c_r1 = c_r0 + R;
    auto sig_of = [](uint32_t t) -> uint32_t { return 1u << (t % 19u); };
    for (uint32_t i = tid; i < R; i += kPugNT) c_minoff[i] = 0xFFFFFFFFu;   // K <= R
    __syncthreads();
    uint32_t V = 0, K = 0;
    constexpr uint32_t kPer = 4;   // consecutive reads per thread: one scan (vertex heads | class heads << 16) per 4096 reads
    for (uint32_t base = 0; base < R; base += kPer * kPugNT) {
        const uint32_t i0 = base + kPer * tid;
        SortRec cur[kPer];
        SortRec prev{0, 0};
        if (i0 > 0 && i0 < R) prev = sr[i0 - 1];
        uint32_t vh = 0, ch = 0, nv = 0, nc = 0;   // bit j: read i0 + j is a vertex / class head
#pragma unroll
        for (uint32_t j = 0; j < kPer; ++j) {
            const uint32_t i = i0 + j;
            cur[j] = i < R ? sr[i] : SortRec{0, 0};
            if (i < R) {
                const SortRec& pv = j ? cur[j - 1] : prev;
                const bool c = i == 0 || cur[j].h != pv.h;
                const bool v = c || rec_umi(cur[j]) != rec_umi(pv);
                ch |= (uint32_t)c << j; vh |= (uint32_t)v << j;
                nc += c; nv += v;
            }
        }
        uint32_t tot;
        const uint32_t ex = block_excl_scan<kPugNT>(nv | (nc << 16), s_ws, tot);
        uint32_t ev = ex & 0xFFFFu, ec = ex >> 16;
        uint32_t pend_k = 0xFFFFFFFFu, pend_min = 0xFFFFFFFFu;   // the thread's reads of one class share one atomicMin
#pragma unroll
        for (uint32_t j = 0; j < kPer; ++j) {
            const uint32_t i = i0 + j;
            if (i < R) {
                const bool c = (ch >> j) & 1u, v = (vh >> j) & 1u;
                const uint32_t vi = V + ev;                  // index of the vertex whose first read this is (when v)
                const uint32_t k = K + ec - (c ? 0 : 1);     // the read's class
                const uint32_t ro = rec_off(cur[j]);
                if (v) { v_umi[vi] = rec_umi(cur[j]); v_cls[vi] = k; v_cnt[vi] = i; }   // v_cnt: head position for now
                if (c) {
                    c_vstart[k] = vi; c_rep[k] = ro;
                    const uint64_t hk = cur[j].h;
                    const uint32_t tag = (uint32_t)(hk >> 62);
                    uint32_t sg = 0;
                    if (tag == 1) sg = sig_of((uint32_t)hk & 0x7FFFFFFFu);
                    else if (tag == 2) sg = sig_of((uint32_t)(hk >> 31) & 0x7FFFFFFFu) | sig_of((uint32_t)hk & 0x7FFFFFFFu);
                    if (tag == 1) { c_r0[k] = (uint32_t)hk & 0x7FFFFFFFu; c_r1[k] = 0; }
                    else if (tag == 2) { c_r0[k] = (uint32_t)(hk >> 31) & 0x7FFFFFFFu; c_r1[k] = (uint32_t)hk & 0x7FFFFFFFu; }
                    if (tag == 3) {
                        if (!C.gene_level) { const Lab l = rec_label(C, ro); for (uint32_t q = 0; q < l.n; ++q) sg |= sig_of(l.p[q] & 0x7FFFFFFFu); }
                        else { uint32_t g[kMaxGenesPerLabel]; const uint32_t len = gene_list(ro, g); if (len != 0xFFFFFFFFu) for (uint32_t q = 0; q < len; ++q) sg |= sig_of(g[q]); else sg = 0x7FFFFu; }
                    }
                    c_sig[k] = sg | (tag << 20);   // bits 20-21: how many refs the label has (3 = more than two: read it from its record)
                }
                if (k != pend_k) { if (pend_k != 0xFFFFFFFFu) atomicMin(&c_minoff[pend_k], pend_min); pend_k = k; pend_min = ro; }
                else pend_min = ro < pend_min ? ro : pend_min;
                if (!c && !label_key_is_exact(cur[j].h)) {
                    const uint32_t po = rec_off(j ? cur[j - 1] : prev);
                    if (!C.gene_level) {
                        if (!lab_equal(rec_label(C, ro), rec_label(C, po))) s_cnt[3] = kErrLabelHash;
                    } else {
                        uint32_t g[kMaxGenesPerLabel], gp[kMaxGenesPerLabel];
                        const uint32_t len = gene_list(ro, g), lenp = gene_list(po, gp);
                        bool same = len == lenp;
                        if (same && len == 0xFFFFFFFFu) same = gene_subset_slow(ro, po) && gene_subset_slow(po, ro);
                        else for (uint32_t q = 0; same && q < len; ++q) same = g[q] == gp[q];
                        if (!same) s_cnt[3] = kErrLabelHash;
                    }
                }
                ev += v; ec += c;
            }
        }
        if (pend_k != 0xFFFFFFFFu) atomicMin(&c_minoff[pend_k], pend_min);
        V += tot & 0xFFFFu; K += tot >> 16;
    }
    if (tid == 0) c_vstart[K] = V;
    __syncthreads();
    if (s_cnt[3]) { if (tid == 0) set_err(A.st, s_cnt[3], cell); return; }
    // gene-level EqMap (init_from_chunk_gene_level, eq_class.rs:723-821): a class label is the sorted distinct
    // gene ids of its ref list; materialise one list per class (the decode grouped reads by a hash of that set)
    if (C.gene_level) {
        uint32_t carry = 0;
        for (uint32_t base = 0; base < K; base += kPugNT) {
            const uint32_t k = base + tid;
            uint32_t g[kMaxGenesPerLabel];
            uint32_t len = k < K ? gene_list(c_rep[k], g) : 0u;
            const bool widek = len == 0xFFFFFFFFu;
            if (widek) { len = 0; gene_first_occurrences(c_rep[k], [&](uint32_t) { ++len; }); }
            uint32_t tot;
            const uint32_t ex = block_excl_scan<kPugNT>(len, s_ws, tot);
            if (k < K) {
                c_goff[k] = carry + ex;
                uint32_t* dstg = c_glab + carry + ex;
                if (!widek) for (uint32_t i = 0; i < len; ++i) dstg[i] = g[i];
                else {   // ascending by insertion, in place
                    uint32_t kk = 0;
                    gene_first_occurrences(c_rep[k], [&](uint32_t gid) {
                        uint32_t q = kk;
                        for (; q > 0 && dstg[q - 1] > gid; --q) dstg[q] = dstg[q - 1];
                        dstg[q] = gid;
                        ++kk;
                    });
                }
            }
            carry += tot;
        }
        if (tid == 0) c_goff[K] = carry;
        __syncthreads();
    }
    auto vlab = [&](uint32_t v) -> Lab {  // label of a vertex: its class's ref list, or gene list at gene level
        if (!C.gene_level) {
            const uint4 q = vv[v];
            const uint32_t code = q.x >> 20;   // 1 / 2: the refs sit in .z / .w of the vertex record; 0: empty; 3: in the chunk
            if (code == 3) return rec_label(C, q.z);
            return Lab{reinterpret_cast<const uint32_t*>(vv + v) + 2, code};
        }
        const uint32_t k = vv[v].y;
        return Lab{c_glab + c_goff[k], c_goff[k + 1] - c_goff[k]};
    };
    __syncthreads();
    if (s_cnt[3]) { if (tid == 0) set_err(A.st, s_cnt[3], cell); return; }
    PUG_MARK(2);
    // ---- 3. class ids by first appearance; reference vertex ids ----
    // c_order[r] = the class that appears r-th in the file.  The classes' first offsets are distinct dwords of the chunk:
    // one bit each in an LDS map, and a class's rank is the number of bits below its own (no sort); chunks over 2^19
    // dwords sort the offsets instead.
    const uint32_t chunk_dw = m.nbytes / 4;
    if (chunk_dw <= 16384u * 32u) {
        const uint32_t nw = (chunk_dw + 31) / 32;
        uint32_t* bm = s_big;
        uint32_t* bm_rank = s_big + 16384;
        for (uint32_t w = tid; w < nw; w += kPugNT) bm[w] = 0;
        __syncthreads();
        for (uint32_t k = tid; k < K; k += kPugNT) { const uint32_t o = c_minoff[k]; atomicOr(&bm[o >> 5], 1u << (o & 31)); }
        __syncthreads();
        uint32_t carry = 0;
        for (uint32_t base = 0; base < nw; base += kPugNT) {
            const uint32_t w = base + tid;
            const uint32_t c = w < nw ? (uint32_t)__popc(bm[w]) : 0u;
            uint32_t tot;
            const uint32_t ex = block_excl_scan<kPugNT>(c, s_ws, tot);
            if (w < nw) bm_rank[w] = carry + ex;
            carry += tot;
        }
        __syncthreads();
        for (uint32_t k = tid; k < K; k += kPugNT) {
            const uint32_t o = c_minoff[k];
            c_order[bm_rank[o >> 5] + (uint32_t)__popc(bm[o >> 5] & ((1u << (o & 31)) - 1u))] = k;
        }
    } else {
        for (uint32_t k = tid; k < K; k += kPugNT) c_order[k] = k;
        __syncthreads();
        bitonic_sort_by<kPugNT>(c_order, K, [&](uint32_t a, uint32_t b) { return c_minoff[a] > c_minoff[b]; });
    }
    __syncthreads();
    {
        uint32_t carry = 0;
        for (uint32_t base = 0; base < K; base += 8 * kPugNT) {   // eight classes per thread and scan, their lookups issued together
            const uint32_t r0 = base + 8 * tid;
            uint32_t kk[8], nv[8], sum = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) kk[j] = r0 + j < K ? c_order[r0 + j] : 0u;
#pragma unroll
            for (int j = 0; j < 8; ++j) nv[j] = r0 + j < K ? c_vstart[kk[j] + 1] - c_vstart[kk[j]] : 0u;
#pragma unroll
            for (int j = 0; j < 8; ++j) { const uint32_t t = nv[j]; nv[j] = sum; sum += t; }
            uint32_t tot;
            const uint32_t ex = block_excl_scan<kPugNT>(sum, s_ws, tot);
#pragma unroll
            for (int j = 0; j < 8; ++j) if (r0 + j < K) c_base[kk[j]] = carry + ex + nv[j];
            carry += tot;
        }
    }
    __syncthreads();
    for (uint32_t j0 = tid; j0 < V; j0 += 4 * kPugNT) {   // four vertices per thread and trip: the class lookups of all four go out together
        uint32_t k[4], h0[4], h1[4], cb[4], cs[4], sg[4], ra[4], rb[4];
        uint64_t um[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t j = j0 + q * kPugNT;
            const bool on = j < V;
            k[q] = on ? v_cls[j] : 0u; um[q] = on ? v_umi[j] : 0ull;
            h0[q] = on ? v_cnt[j] : 0u; h1[q] = on && j + 1 < V ? v_cnt[j + 1] : R;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            cb[q] = c_base[k[q]]; cs[q] = c_vstart[k[q]];
            sg[q] = C.gene_level ? (3u << 20) : c_sig[k[q]];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t code = sg[q] >> 20;
            const bool inl = code == 1 || code == 2;
            ra[q] = inl ? c_r0[k[q]] : c_rep[k[q]];
            rb[q] = inl ? c_r1[k[q]] : 0u;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t j = j0 + q * kPugNT;
            if (j >= V) continue;
            const uint32_t vid = cb[q] + (j - cs[q]);
            vv_umi[vid] = um[q];
            vv[vid] = make_uint4((h1[q] - h0[q]) | ((sg[q] >> 20) << 20), k[q], ra[q], rb[q]);
        }
    }
    __syncthreads();
    PUG_MARK(3);
    // ---- 4. neighbour search ----
    uint32_t NCAND = 0;
    uint64_t* cand = nullptr;
    uint64_t* pairs = nullptr;
    uint32_t pair_cap = 0;
    constexpr uint64_t kPairRuled = 1ull << 62, kPairCheck = 1ull << 61;   // LDS route: count rule already applied / labels still to compare
    constexpr uint64_t kPairBwd = 1ull << 60, kPairFwd = 1ull << 59;        // which of y -> x, x -> y the pair stands for (the LDS route meets a pair once)
    bool fast = false;
#ifdef AFQ_PUG_TIMING
    if (tid == 0) { for (int q_ = 0; q_ < 8; ++q_) tacc[q_] = 0; tlast = wall_clock64(); }
#endif
    {
        uint32_t lgP = 0;
        while (lgP < 20 && ((uint64_t)kPartTarget << lgP) < V) lgP += 2;
        const uint32_t L = C.umi_pairs;
        if (3 * (L - lgP / 2) > 64 && lgP + 2 <= 2 * L) lgP += 2;   // the per-lane probe mask has 64 bits (UMIs of 22 nt in a tiny cell)
        const uint32_t P = 1u << lgP;
        if (P <= kMaxParts && lgP <= 2 * L && 3 * (L - lgP / 2) <= 64 && !A.force_global_route) {
            uint64_t* pv_umi = v_umi;                                  // slab B is free until the components are listed:
            uint64_t* pv_info = reinterpret_cast<uint64_t*>(v_cnt);    // the vertices again, grouped by partition
            for (uint32_t i = tid; i <= P; i += kPugNT) s_poff[i] = 0;
            if (tid == 0) { s_flag[0] = 0; s_flag[1] = 0; }
            __syncthreads();
            for (uint32_t v = tid; v < V; v += kPugNT) atomicAdd(&s_poff[(uint32_t)vv_umi[v] & (P - 1)], 1u);
            __syncthreads();
            {   // exclusive offsets (P <= 1024 = one value per thread), oversize check
                const uint32_t c = tid < P ? s_poff[tid] : 0u;
                if (c > kPartMax) s_flag[1] = 1;
                uint32_t tot;
                const uint32_t ex = block_excl_scan<kPugNT>(c, s_ws, tot);
                __syncthreads();
                if (tid < P) s_poff[tid] = ex;
                if (tid == 0) s_poff[P] = tot;
                __syncthreads();
            }
            if (!s_flag[1]) {
                uint32_t* fill = s_big;   // per-partition fill cursors (the table is not in use yet)
                for (uint32_t i = tid; i < P; i += kPugNT) fill[i] = 0;
                __syncthreads();
                for (uint32_t v = tid; v < V; v += kPugNT) {
                    const uint64_t umi = vv_umi[v];
                    const uint4 q = vv[v];
                    const uint32_t pp = (uint32_t)umi & (P - 1);
                    const uint32_t o = s_poff[pp] + atomicAdd(&fill[pp], 1u);
                    if (umi >> 44) s_flag[1] = 1;   // (the global route reports it)
                    pv_umi[o] = umi | ((uint64_t)(c_sig[q.y] & 0x7FFFFu) << 44);   // key word of the table: UMI (<= 22 nt) + the label signature
                    pv_info[o] = (uint64_t)v | ((uint64_t)(q.x & 0xFFFFFu) << 20) | ((uint64_t)q.y << 40);
                }
                pair_cap = 2 * V + 4096;
                if (tid == 0) s_ebase = atomicAdd(A.epool_cursor, 2ull * pair_cap + 2);
                __syncthreads();
                if (s_ebase + 2ull * pair_cap + 2 > A.epool_cap) { if (tid == 0) set_err(A.st, kErrPugPool, cell); return; }
                pairs = reinterpret_cast<uint64_t*>(A.epool + ((s_ebase + 1) & ~1ull));
              if (!s_flag[1]) {
                uint64_t* t_key = reinterpret_cast<uint64_t*>(s_big);
                uint64_t* t_val = t_key + kTabSlots;
                constexpr uint64_t kEmptyKey = ~0ull;
                const uint32_t vm = (1u << kVidBits) - 1;
                // Slot of a UMI = its 13-bit XOR fold.  UMIs are random sequences, so the fold spreads them as well as a
                // multiplicative hash would, and it is linear: the slot of a one-base neighbour is the vertex's own slot
                // XOR a constant of the changed base - one instruction per probe where a 64-bit multiply is four
                // quarter-rate ones (the probes, 1 + 3L per vertex, are the whole cost of this phase).
                auto fold13 = [](uint64_t u) -> uint32_t { u ^= u >> 26; const uint32_t t = (uint32_t)u & 0x3FFFFFFu; return (t ^ (t >> 13)) & (kTabSlots - 1); };   // 13-bit pieces of up to 52 bits
                // In front of the table, one bit per UMI in a 2^16-bit map (same kind of fold, 16 bits): a probe that finds its
                // bit clear has no match - the fate of ~95 % of the 3L one-base probes - and costs a handful of
                // instructions with no loop.  A lane first collects the probes that pass in a bit mask, then walks the table
                // for those only: a wave spends table walks on the few probes that can match, not on every probe of every lane.
                auto fold16 = [](uint64_t u) -> uint32_t { u ^= u >> 32; const uint32_t t = (uint32_t)u; return (t ^ (t >> 16)) & 0xFFFFu; };
                auto filt = [&](uint32_t h) -> uint32_t { return (s_filt[h >> 5] >> (h & 31u)) & 1u; };
                // every match of probe UMI `pu` for vertex x: the edge x -> y exists at distance 0, or at distance 1 unless
                // reads(y) >= 2 * reads(x) (has_edge, pugutils.rs:76-99) - provided the labels overlap, which only classes
                // that differ still have to show (stage 2)
                constexpr uint64_t kUmi44 = (1ull << 44) - 1;
#ifdef AFQ_PUG_TIMING
                unsigned long long dbg_steps = 0, dbg_probes = 0;
#define PUG_DBG_STEP ++dbg_steps
#define PUG_DBG_PROBE ++dbg_probes
#else
#define PUG_DBG_STEP
#define PUG_DBG_PROBE
#endif
                auto probe = [&](uint64_t pu, uint32_t slot, uint64_t xinfo, uint32_t xsig, bool same) {
                    PUG_DBG_PROBE;
                    for (;; slot = (slot + 1) & (kTabSlots - 1)) {
                        PUG_DBG_STEP;
                        const uint64_t k = t_key[slot];
                        if (k == kEmptyKey) break;
                        if ((k & kUmi44) != pu) continue;
                        if (((uint32_t)(k >> 44) & xsig) == 0) continue;   // labels without a common ref: no edge whatever the UMIs
                        const uint64_t info = t_val[slot];
                        const uint32_t y = (uint32_t)info & vm, x = (uint32_t)xinfo & vm;
                        // A pair of vertices is met ONCE - equal UMIs from the smaller vertex id, one-base neighbours from the
                        // smaller UMI (the callers only probe upwards) - and both directions are decided here:
                        // x -> y unless reads(y) >= 2 reads(x), y -> x unless reads(x) >= 2 reads(y); always both at distance 0.
                        uint64_t dir = kPairFwd | kPairBwd;
                        if (same) { if (y <= x) continue; }
                        else {
                            const uint32_t cy = (uint32_t)(info >> 20) & vm, cx = (uint32_t)(xinfo >> 20) & vm;
                            dir = (cy < 2 * cx ? kPairFwd : 0ull) | (cx < 2 * cy ? kPairBwd : 0ull);
                            if (!dir) continue;
                        }
                        const uint64_t am = __ballot(true);   // one LDS atomic per wave, not per match
                        const uint32_t leader = (uint32_t)__builtin_ctzll(am);
                        uint32_t base = 0;
                        if (lane == leader) base = atomicAdd(&s_flag[0], (uint32_t)__popcll(am));
                        base = __builtin_amdgcn_readlane(base, (int)leader);
                        const uint32_t kk = base + (uint32_t)__popcll(am & ((1ull << lane) - 1));
                        if (kk < pair_cap)
                            pairs[kk] = kPairRuled | dir | ((uint32_t)(info >> 40) != (uint32_t)(xinfo >> 40) ? kPairCheck : 0ull) | ((uint64_t)same << 63) | ((uint64_t)x << 32) | y;
                    }
                };
                const uint32_t lowb = lgP / 2;   // bases whose change moves a UMI to another partition
                PUG_ACC(0);
                for (uint32_t pp = 0; pp < P; ++pp) {
                    const uint32_t o0 = s_poff[pp], n = s_poff[pp + 1] - o0;
                    if (n == 0) continue;
                    __syncthreads();   // the previous pass is done with the table
                    for (uint32_t i = tid; i < kTabSlots; i += kPugNT) t_key[i] = kEmptyKey;
                    for (uint32_t i = tid; i < 2048; i += kPugNT) s_filt[i] = 0;
                    __syncthreads();
                    PUG_ACC(1);
                    for (uint32_t i = tid; i < n; i += kPugNT) {
                        const uint64_t umi = pv_umi[o0 + i];
                        uint32_t slot = fold13(umi & kUmi44);
                        while (atomicCAS(reinterpret_cast<unsigned long long*>(&t_key[slot]), (unsigned long long)kEmptyKey, (unsigned long long)umi) != kEmptyKey)
                            slot = (slot + 1) & (kTabSlots - 1);   // (equal UMIs under different labels each take a slot of the same run)
                        t_val[slot] = pv_info[o0 + i];
                        const uint32_t hb = fold16(umi & kUmi44);
                        atomicOr(&s_filt[hb >> 5], 1u << (hb & 31u));
                    }
                    __syncthreads();
                    PUG_ACC(2);
                    // the partition's own vertices: distance 0, and every change of a high base.  Three vertices per thread sit
                    // in registers while the (base, substitution) loop - uniform, so its constants are scalar - runs over them.
                    for (uint32_t i0 = tid; i0 < n; i0 += 3 * kPugNT) {
                        uint64_t mu[3], mi[3];
                        uint32_t mh[3], ms[3];
#pragma unroll
                        for (int m2 = 0; m2 < 3; ++m2) {
                            const uint32_t i = i0 + m2 * kPugNT;
                            const uint64_t w = i < n ? pv_umi[o0 + i] : 0ull;
                            mu[m2] = w & kUmi44; ms[m2] = (uint32_t)(w >> 44);
                            mi[m2] = i < n ? pv_info[o0 + i] : 0ull;
                            mh[m2] = fold13(mu[m2]);
                        }
#pragma unroll
                        for (int m2 = 0; m2 < 3; ++m2) if (i0 + m2 * kPugNT < n) probe(mu[m2], mh[m2], mi[m2], ms[m2], true);
                        if (!C.exact_umi) {
                            uint64_t pm[3] = {0, 0, 0};   // bit (3 * (b - lowb) + d - 1): that neighbour's filter bit is set
                            uint32_t mf[3];
#pragma unroll
                            for (int m2 = 0; m2 < 3; ++m2) mf[m2] = fold16(mu[m2]);
                            // vertices this trip really holds (uniform): the unused register slots of a short trip are skipped
                            const uint32_t mcnt = (n - __builtin_amdgcn_readfirstlane(i0 - tid) + kPugNT - 1) / kPugNT;
                            // Per base: the three substitutions' filter words of all (up to) three vertices are read together
                            // - nine LDS reads in flight, not one at a time - and the "upwards only" test is a bit test on the
                            // base itself: d = 1 raises the UMI iff bit 0 of the base is clear, d = 2, 3 iff bit 1 is.
                            for (uint32_t b = lowb; b < L; ++b) {
                                uint32_t fw[3][3], fh[3][3];
#pragma unroll
                                for (uint32_t d = 1; d < 4; ++d) {
                                    const uint32_t df = fold16((uint64_t)d << (2 * b));
#pragma unroll
                                    for (int m2 = 0; m2 < 3; ++m2) {
                                        fh[d - 1][m2] = mf[m2] ^ df;
                                        fw[d - 1][m2] = (uint32_t)m2 < mcnt ? s_filt[fh[d - 1][m2] >> 5] : 0u;
                                    }
                                }
#pragma unroll
                                for (int m2 = 0; m2 < 3; ++m2) {
                                    if ((uint32_t)m2 >= mcnt) continue;
                                    const uint32_t nbase = ~(uint32_t)(mu[m2] >> (2 * b));   // bit 0 / bit 1 set: that bit of the base is clear
                                    const uint32_t up1 = nbase & 1u, up23 = (nbase >> 1) & 1u;
                                    const uint32_t g = ((fw[0][m2] >> (fh[0][m2] & 31u)) & up1) | (((fw[1][m2] >> (fh[1][m2] & 31u)) & up23) << 1) |
                                                       (((fw[2][m2] >> (fh[2][m2] & 31u)) & up23) << 2);
                                    pm[m2] |= (uint64_t)g << (3 * (b - lowb));
                                }
                            }
#pragma unroll
                            for (int m2 = 0; m2 < 3; ++m2) {
                                uint64_t todo = i0 + m2 * kPugNT < n ? pm[m2] : 0ull;
                                while (todo) {
                                    const uint32_t ix = (uint32_t)__builtin_ctzll(todo);
                                    todo &= todo - 1;
                                    const uint32_t b = lowb + ix / 3, d = ix % 3 + 1;
                                    const uint64_t mask = (uint64_t)d << (2 * b);
                                    probe(mu[m2] ^ mask, mh[m2] ^ fold13(mask), mi[m2], ms[m2], false);
                                }
                            }
                        }
                    }
                    PUG_ACC(3);
                    if (!C.exact_umi)
                        for (uint32_t b = 0; b < lowb; ++b) {   // vertices of the partitions one low-base change away: that one change.
                            // The three source partitions of a base are read together (three loads in flight per thread; the
                            // loop is all latency otherwise), the vertex's other word only when the filter lets the probe through.
                            uint32_t oq[3], nq[3], nmax = 0;
#pragma unroll
                            for (uint32_t d = 1; d < 4; ++d) {
                                const uint32_t q = pp ^ (d << (2 * b));
                                oq[d - 1] = s_poff[q]; nq[d - 1] = s_poff[q + 1] - oq[d - 1];
                                nmax = nq[d - 1] > nmax ? nq[d - 1] : nmax;
                            }
                            for (uint32_t i = tid; i < nmax; i += kPugNT) {
                                uint64_t w[3];
#pragma unroll
                                for (int d = 0; d < 3; ++d) w[d] = i < nq[d] ? pv_umi[oq[d] + i] : ~0ull;
#pragma unroll
                                for (int d = 0; d < 3; ++d) {
                                    if (i >= nq[d]) continue;
                                    const uint64_t pu = (w[d] & kUmi44) ^ ((uint64_t)(d + 1) << (2 * b));
                                    if (pu > (w[d] & kUmi44) && filt(fold16(pu))) probe(pu, fold13(pu), pv_info[oq[d] + i], (uint32_t)(w[d] >> 44), false);
                                }
                            }
                        }
                    PUG_ACC(4);
                }
                __syncthreads();
                fast = s_flag[0] <= pair_cap;
#ifdef AFQ_PUG_TIMING
                if (blockIdx.x < 4) { atomicAdd(&tacc[6], dbg_steps); atomicAdd(&tacc[7], dbg_probes); }
                __syncthreads();
#endif
              }   // (more matches than the list holds: a cell full of near-identical UMIs - the global route sizes its lists exactly)
            }
            __syncthreads();
        }
    }
    // ---- 4'. the global-memory route (hash table of the vertices by UMI in the cell's scratch, an LDS presence filter in
    // front of it): cells whose partitions do not fit the LDS table ----
    // (every vertex with a given UMI sits on the probe run that starts at the UMI's home slot)
    constexpr unsigned long long kHtEmpty = ~0ull;
    const uint32_t ht_mask = ht_cap - 1;
    const uint32_t ht_shift = 64 - (uint32_t)__builtin_ctz(ht_cap);
    auto ht_home = [&](uint64_t umi) -> uint32_t { return (uint32_t)((umi * 0xD6E8FEB86659FD93ull) >> ht_shift); };
    uint32_t* vflag = c_minoff;  // dead after phase 3: 1 = another vertex carries the same UMI
    if (!fast) {
    for (uint32_t i = tid; i < ht_cap; i += kPugNT) htab[i] = kHtEmpty;
    for (uint32_t i = tid; i < (1u << 15); i += kPugNT) s_bloom[i] = 0;
    for (uint32_t v = tid; v < V; v += kPugNT) if (vv_umi[v] >> (64 - kVidBits)) s_cnt[3] = kErrPugLimit;
    __syncthreads();
    if (s_cnt[3]) { if (tid == 0) set_err(A.st, s_cnt[3], cell); return; }
    for (uint32_t v = tid; v < V; v += kPugNT) vflag[v] = 0;
    __syncthreads();
    for (uint32_t v = tid; v < V; v += kPugNT) {
        const uint64_t umi = vv_umi[v];
        const unsigned long long e = ((unsigned long long)umi << kVidBits) | v;
        for (uint32_t slot = ht_home(umi);; slot = (slot + 1) & ht_mask) {
            const unsigned long long old = atomicCAS(&htab[slot], kHtEmpty, e);
            if (old == kHtEmpty) break;
            // same UMI under another label: whichever of the two is inserted second walks over the first
            if ((old >> kVidBits) == umi) { vflag[v] = 1; vflag[(uint32_t)old & ((1u << kVidBits) - 1)] = 1; }
        }
    }
    }
    PUG_MARK(11);
    auto bloom_bit = [](uint64_t umi) -> uint32_t { return (uint32_t)((umi * kHashMul) >> 44); };
    auto bloom_bit2 = [](uint64_t umi) -> uint32_t { return (uint32_t)(((umi ^ (umi >> 23)) * 0xD6E8FEB86659FD93ull) >> 44); };
    if (!fast) {
    for (uint32_t v = tid; v < V; v += kPugNT) {
        const uint32_t b = bloom_bit(vv_umi[v]), b2 = bloom_bit2(vv_umi[v]);
        atomicOr(&s_bloom[b >> 5], 1u << (b & 31));
        atomicOr(&s_bloom[b2 >> 5], 1u << (b2 & 31));
    }
    }
    __syncthreads();
    PUG_MARK(4);
    // Edge generation in two steps so that the expensive part runs with full waves: (A) every vertex tests
    // its 1 + 3L probe UMIs against the LDS filter and the survivors (probe UMI, vertex) go to a candidate
    // list; (B) one thread per candidate binary-searches the UMI-sorted vertices and applies the edge rule.
    const uint32_t nprobe = C.exact_umi ? 1u : 1u + 3u * C.umi_pairs;
    auto probe_umi = [&](uint64_t ux, uint32_t pr) -> uint64_t {
        if (!pr) return ux;
        const uint32_t b = (pr - 1) / 3, d = (pr - 1) % 3 + 1;
        return ux ^ ((uint64_t)d << (2 * b));
    };
    auto passes = [&](uint64_t pu) -> bool {
        if (pu >> (64 - kVidBits)) return false;
        const uint32_t b = bloom_bit(pu), b2 = bloom_bit2(pu);
        return ((s_bloom[b >> 5] >> (b & 31)) & (s_bloom[b2 >> 5] >> (b2 & 31))) & 1u;
    };
    // (A) candidates (pu << 20 | x) appended in one pass to a list with room for 4V + 4096 of them (one LDS atomic per
    // wave and probe round); a cell whose filter lets more through than that takes the exact two-pass route
    // (count, reserve, write).
    if (!fast) {
        const uint32_t cand_cap = 4 * V + 4096;
        if (tid == 0) { s_ebase = atomicAdd(A.epool_cursor, 2ull * cand_cap + 2); s_flag[1] = 0; }
        __syncthreads();
        if (s_ebase + 2ull * cand_cap + 2 > A.epool_cap) { if (tid == 0) set_err(A.st, kErrPugPool, cell); return; }
        cand = reinterpret_cast<uint64_t*>(A.epool + ((s_ebase + 1) & ~1ull));
        auto append = [&](uint64_t c) {   // called under divergence: the active lanes share one reservation
            const uint64_t am = __ballot(true);
            const uint32_t leader = (uint32_t)__builtin_ctzll(am);
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(&s_flag[1], (uint32_t)__popcll(am));
            base = __builtin_amdgcn_readlane(base, (int)leader);
            const uint32_t k = base + (uint32_t)__popcll(am & ((1ull << lane) - 1));
            if (k < cand_cap) cand[k] = c;
        };
        for (uint32_t x = tid; x < V; x += kPugNT) {
            const uint64_t ux = vv_umi[x];
            if (vflag[x]) append((ux << kVidBits) | x);  // the distance-0 probe only matters when the UMI occurs under another label too
            for (uint32_t pr = 1; pr < nprobe; ++pr) {
                const uint64_t pu = probe_umi(ux, pr);
                if (passes(pu)) append((pu << kVidBits) | x);
            }
        }
        __syncthreads();
        NCAND = s_flag[1];
        PUG_MARK(12);
        if (NCAND > cand_cap) {
            uint32_t ncand_mine = 0;
            for (uint32_t x = tid; x < V; x += kPugNT) {
                const uint64_t ux = vv_umi[x];
                ncand_mine += vflag[x];
                for (uint32_t pr = 1; pr < nprobe; ++pr) ncand_mine += passes(probe_umi(ux, pr));
            }
            const uint32_t cand_off = block_excl_scan<kPugNT>(ncand_mine, s_ws, NCAND);
            if (tid == 0) s_ebase = atomicAdd(A.epool_cursor, 2ull * NCAND + 2);
            __syncthreads();
            if (s_ebase + 2ull * NCAND + 2 > A.epool_cap) { if (tid == 0) set_err(A.st, kErrPugPool, cell); return; }
            cand = reinterpret_cast<uint64_t*>(A.epool + ((s_ebase + 1) & ~1ull));
            uint32_t o = cand_off;
            for (uint32_t x = tid; x < V; x += kPugNT) {
                const uint64_t ux = vv_umi[x];
                if (vflag[x]) cand[o++] = (ux << kVidBits) | x;
                for (uint32_t pr = 1; pr < nprobe; ++pr) {
                    const uint64_t pu = probe_umi(ux, pr);
                    if (passes(pu)) cand[o++] = (pu << kVidBits) | x;
                }
            }
        }
    }
    PUG_MARK(13);
    uint32_t* tch = c_base;  // dead after phase 3: 1 = the vertex is the target of some edge
    for (uint32_t x = tid; x <= V; x += kPugNT) { deg[x] = 0; if (x < V) tch[x] = 0; }
    __syncthreads();
    // out-neighbours of x: vertices y != x with overlapping labels and UMI distance 0, or distance 1 and
    // count(y) < 2*count(x)  (has_edge, pugutils.rs:76-99: X->Y unless cy >= 2cx)
    auto for_each_edge_of = [&](uint64_t cd, auto&& f) {
        const uint32_t x = (uint32_t)cd & ((1u << kVidBits) - 1);
        const uint64_t pu = cd >> kVidBits;
        const bool same = pu == vv_umi[x];
        const uint4 vx = vv[x];
        const uint32_t cx = vx.x & 0xFFFFFu, kx = vx.y;
        for (uint32_t slot = ht_home(pu);; slot = (slot + 1) & ht_mask) {
            const unsigned long long e = htab[slot];
            if (e == kHtEmpty) break;
            if ((e >> kVidBits) != pu) continue;
            const uint32_t y = (uint32_t)e & ((1u << kVidBits) - 1);
            if (y == x) continue;
            const uint4 vy = vv[y];
            if (!same && !((vy.x & 0xFFFFFu) < 2 * cx)) continue;
            if (vy.y != kx && !lab_overlap(vlab(x), vlab(y))) continue;
            f(x, y);
        }
    };
    // The UMI matches of the candidates go to a list with room for NCAND + V entries; only a cell with piles of
    // same-UMI vertices can overflow it, and then the candidates are walked the plain way (twice: count, fill).
    if (!fast) {
    pair_cap = NCAND + V;
    if (tid == 0) { s_ebase = atomicAdd(A.epool_cursor, 2ull * pair_cap + 2); s_flag[0] = 0; }
    __syncthreads();
    if (s_ebase + 2ull * pair_cap + 2 > A.epool_cap) { if (tid == 0) set_err(A.st, kErrPugPool, cell); return; }
    pairs = reinterpret_cast<uint64_t*>(A.epool + ((s_ebase + 1) & ~1ull));
    }
    // Two stages, each with four independent load chains per thread (a probe of the vertex table or of a
    // label is a far random access - the time goes into waiting, not computing):
    //  stage 1 walks the probe run of every candidate and lists the vertices that carry the probed UMI;
    //  stage 2 applies the edge rule (read counts, then labels) to the listed matches only - about a third
    //  of the candidates - and counts the out-degree of the survivors.
    constexpr uint64_t kNoPair = ~0ull;
    const uint32_t vmask = (1u << kVidBits) - 1;
    for (uint32_t i0 = tid; i0 < NCAND; i0 += 4 * kPugNT) {   // (NCAND = 0 on the LDS route: its matches are listed already)
        uint64_t cd[4], ux[4];
        unsigned long long e[4];
        uint32_t slot[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t i = i0 + j * kPugNT;
            cd[j] = i < NCAND ? cand[i] : 0ull;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t i = i0 + j * kPugNT;
            slot[j] = ht_home(cd[j] >> kVidBits);
            e[j] = i < NCAND ? htab[slot[j]] : kHtEmpty;
            ux[j] = vv_umi[(uint32_t)cd[j] & vmask];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t x = (uint32_t)cd[j] & vmask;
            const uint64_t pu = cd[j] >> kVidBits;
            while (e[j] != kHtEmpty) {
                const uint32_t y = (uint32_t)e[j] & vmask;
                if ((e[j] >> kVidBits) == pu && y != x) {
                    // one LDS atomic per wave, not per match (same-address atomics serialise across the whole CU)
                    const uint64_t am = __ballot(true);
                    const uint32_t leader = (uint32_t)__builtin_ctzll(am);
                    uint32_t base = 0;
                    if (lane == leader) base = atomicAdd(&s_flag[0], (uint32_t)__popcll(am));
                    base = __builtin_amdgcn_readlane(base, (int)leader);
                    const uint32_t k = base + (uint32_t)__popcll(am & ((1ull << lane) - 1));
                    if (k < pair_cap) pairs[k] = ((uint64_t)(pu == ux[j]) << 63) | ((uint64_t)x << 32) | y;
                }
                slot[j] = (slot[j] + 1) & ht_mask;
                e[j] = htab[slot[j]];
            }
        }
    }
    __syncthreads();
    const uint32_t n_pairs = s_flag[0];
    // Out-degrees and fill cursors as 16-bit counters in LDS (two per word, indexed by vertex id) when the cell has at most
    // 65 536 vertices: the counting and the edge fill are LDS atomics instead of global ones (the search's table is done with
    // the 128 KiB block by now, the components' labels move in afterwards).
    const bool lds_deg = fast && n_pairs <= pair_cap && V <= 65536u;
    uint32_t* cw = s_big;
    if (lds_deg) { for (uint32_t w = tid; w < (V + 1) / 2; w += kPugNT) cw[w] = 0; __syncthreads(); }
    auto cw_add = [&](uint32_t v) -> uint32_t { const uint32_t sh = (v & 1u) * 16u; return (atomicAdd(&cw[v >> 1], 1u << sh) >> sh) & 0xFFFFu; };
    if (n_pairs <= pair_cap) {
        for (uint32_t k0 = tid; k0 < n_pairs; k0 += 4 * kPugNT) {
            uint64_t pr[4];
            uint32_t cx[4], cy[4], kx[4], ky[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t k = k0 + j * kPugNT;
                pr[j] = k < n_pairs ? pairs[k] : kNoPair;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (fast) { cx[j] = 1; cy[j] = 0; kx[j] = 0; ky[j] = (pr[j] & kPairCheck) ? 1u : 0u; continue; }   // the count rule is applied, classes compared
                const uint32_t x = pr[j] == kNoPair ? 0u : (uint32_t)(pr[j] >> 32) & vmask, y = pr[j] == kNoPair ? 0u : (uint32_t)pr[j] & vmask;
                const uint4 va = vv[x], vb = vv[y];
                cx[j] = va.x & 0xFFFFFu; cy[j] = vb.x & 0xFFFFFu; kx[j] = va.y; ky[j] = vb.y;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t k = k0 + j * kPugNT;
                if (k >= n_pairs) continue;
                const uint32_t x = (uint32_t)(pr[j] >> 32) & vmask, y = (uint32_t)pr[j] & vmask;
                const bool same = (pr[j] >> 63) != 0;
                bool keep = same || cy[j] < 2 * cx[j];
                if (keep && ky[j] != kx[j]) keep = lab_overlap(vlab(x), vlab(y));
                const uint64_t dir = fast ? pr[j] & (kPairFwd | kPairBwd) : kPairFwd;
                if (keep) {
                    if (lds_deg) { if (dir & kPairFwd) (void)cw_add(x); if (dir & kPairBwd) (void)cw_add(y); }
                    else { if (dir & kPairFwd) atomicAdd(&deg[x], 1u); if (dir & kPairBwd) atomicAdd(&deg[y], 1u); }
                }
                pairs[k] = keep ? (dir | ((uint64_t)x << 32) | y) : kNoPair;
            }
        }
    } else {
        for (uint32_t i = tid; i < NCAND; i += kPugNT) for_each_edge_of(cand[i], [&](uint32_t x, uint32_t) { atomicAdd(&deg[x], 1u); });
    }
    __syncthreads();
    PUG_MARK(14);
    uint32_t E = 0;
    for (uint32_t base = 0; base < V; base += 8 * kPugNT) {   // eight consecutive vertices per thread and scan
        const uint32_t x0 = base + 8 * tid;
        uint32_t d[8], sum = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) d[j] = x0 + j < V ? (lds_deg ? (cw[(x0 + j) >> 1] >> (((x0 + j) & 1u) * 16u)) & 0xFFFFu : deg[x0 + j]) : 0u;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const uint32_t t = d[j]; d[j] = sum; sum += t; }
        uint32_t tot;
        const uint32_t ex = block_excl_scan<kPugNT>(sum, s_ws, tot);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) if (x0 + j < V) { deg[x0 + j] = E + ex + d[j]; c_order[x0 + j] = E + ex + d[j]; }  // c_order (dead after phase 3) = per-vertex fill cursor
        E += tot;
    }
    if (tid == 0) {
        deg[V] = E;
        s_ebase = atomicAdd(A.epool_cursor, (unsigned long long)E);
    }
    __syncthreads();
    if (s_ebase + E > A.epool_cap) { if (tid == 0) set_err(A.st, kErrPugPool, cell); return; }
    uint32_t* edges = A.epool + s_ebase;
    if (lds_deg) {
        for (uint32_t w = tid; w < (V + 1) / 2; w += kPugNT) cw[w] = 0;   // the counters again, as fill cursors
        __syncthreads();
        for (uint32_t k0 = tid; k0 < n_pairs; k0 += 2 * kPugNT) {
            uint64_t pr[2];
            uint32_t dx[2], dy[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) { const uint32_t k = k0 + j * kPugNT; pr[j] = k < n_pairs ? pairs[k] : ~0ull; }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint32_t x = (uint32_t)(pr[j] >> 32) & vmask, y = (uint32_t)pr[j] & vmask;
                dx[j] = pr[j] != ~0ull && (pr[j] & kPairFwd) ? deg[x] : 0u;
                dy[j] = pr[j] != ~0ull && (pr[j] & kPairBwd) ? deg[y] : 0u;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (pr[j] == ~0ull) continue;
                const uint32_t x = (uint32_t)(pr[j] >> 32) & vmask, y = (uint32_t)pr[j] & vmask;
                if (pr[j] & kPairFwd) { edges[dx[j] + cw_add(x)] = y; tch[y] = 1; }
                if (pr[j] & kPairBwd) { edges[dy[j] + cw_add(y)] = x; tch[x] = 1; }
            }
        }
    } else if (n_pairs <= pair_cap) {
        for (uint32_t k = tid; k < n_pairs; k += kPugNT) {
            const uint64_t pr = pairs[k];
            if (pr != ~0ull) {
                const uint32_t x = (uint32_t)(pr >> 32) & vmask, y = (uint32_t)pr & vmask;
                if (pr & kPairFwd) { edges[atomicAdd(&c_order[x], 1u)] = y; tch[y] = 1; }
                if (pr & kPairBwd) { edges[atomicAdd(&c_order[y], 1u)] = x; tch[x] = 1; }
            }
        }
    } else {
        for (uint32_t i = tid; i < NCAND; i += kPugNT)
            for_each_edge_of(cand[i], [&](uint32_t x, uint32_t y) { edges[atomicAdd(&c_order[x], 1u)] = y; tch[y] = 1; });
    }
    __syncthreads();
    PUG_MARK(5);
    // ---- 5. weakly connected components: min-label propagation + pointer jumping ----
    // Only vertices with an edge take part (~15 % of them: most molecules have no UMI neighbour); the others are
    // their own components and are resolved straight from their labels in 6a, without being listed or sorted.
    uint32_t* tl = v_cls;   // slab B is dead: the vertices that have an edge, ascending
    uint32_t NT = 0;
    for (uint32_t base = 0; base < V; base += 8 * kPugNT) {
        const uint32_t v0 = base + 8 * tid;
        uint32_t dg[9], tc[8], hm = 0, sum = 0;
#pragma unroll
        for (int j = 0; j <= 8; ++j) dg[j] = v0 + j <= V ? deg[v0 + j] : 0u;
#pragma unroll
        for (int j = 0; j < 8; ++j) tc[j] = v0 + j < V ? tch[v0 + j] : 0u;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const bool h = v0 + j < V && (dg[j + 1] > dg[j] || tc[j]); hm |= (uint32_t)h << j; sum += h; }
        uint32_t tot;
        const uint32_t ex = block_excl_scan<kPugNT>(sum, s_ws, tot);
        uint32_t o = NT + ex;
#pragma unroll
        for (int j = 0; j < 8; ++j) if ((hm >> j) & 1u) tl[o++] = v0 + j;
        NT += tot;
    }
    // The labels live in LDS, indexed by a vertex's position in tl (ascending with the vertex id, so the smallest label of
    // a component is still its smallest vertex): the sweeps are LDS atomics and LDS reads, the edges are read once per
    // sweep as positions (translated once).  A cell with more than 32 768 touched vertices keeps the labels in global memory.
    const bool wcc_lds = NT <= (1u << 15) && !A.force_global_route;   // (AFQ_TEST_PUG_GLOBAL_ROUTE: tests keep the global-memory variants covered)
    if (wcc_lds) {
        uint32_t* wl = s_big;
        __syncthreads();
        if (tid == 0) s_ebase = atomicAdd(A.epool_cursor, (unsigned long long)E);
        for (uint32_t i = tid; i < NT; i += kPugNT) { wl[i] = i; local_idx[tl[i]] = i; }
        __syncthreads();
        if (s_ebase + E > A.epool_cap) { if (tid == 0) set_err(A.st, kErrPugPool, cell); return; }
        uint32_t* pedge = A.epool + s_ebase;   // edge targets as positions in tl
        for (uint32_t i = tid; i < NT; i += kPugNT) {
            const uint32_t x = tl[i];
            for (uint32_t e = deg[x]; e < deg[x + 1]; ++e) pedge[e] = local_idx[edges[e]];
        }
        __syncthreads();
        for (;;) {
            __syncthreads();   // (all threads have read the previous sweep's flag before it is cleared)
            if (tid == 0) s_flag[0] = 0;
            __syncthreads();
            bool ch = false;
            for (uint32_t i = tid; i < NT; i += kPugNT) {
                const uint32_t x = tl[i];
                for (uint32_t e = deg[x]; e < deg[x + 1]; ++e) {
                    const uint32_t j = pedge[e];
                    const uint32_t a = wl[i], b = wl[j];
                    if (a < b) { atomicMin(&wl[j], a); ch = true; }
                    else if (b < a) { atomicMin(&wl[i], b); ch = true; }
                }
            }
            if (ch) s_flag[0] = 1;
            __syncthreads();
            for (int it = 0; it < 4; ++it) {
                for (uint32_t i = tid; i < NT; i += kPugNT) { const uint32_t l = wl[i]; const uint32_t ll = wl[l]; if (ll < l) wl[i] = ll; }
                __syncthreads();
            }
            if (!s_flag[0]) break;
        }
        PUG_MARK(6);
        for (uint32_t i = tid; i < NT; i += kPugNT) { uint32_t l = wl[i]; while (wl[l] != l) l = wl[l]; comp_sorted[i] = ((uint64_t)tl[l] << kVidBits) | tl[i]; }
        __syncthreads();
    } else {
    for (uint32_t v = tid; v < V; v += kPugNT) wlab[v] = v;
    __syncthreads();
    for (;;) {
        __syncthreads();   // (all threads have read the previous sweep's flag before it is cleared)
        if (tid == 0) s_flag[0] = 0;
        __syncthreads();
        bool ch = false;
        for (uint32_t i = tid; i < NT; i += kPugNT) {
            const uint32_t x = tl[i];
            for (uint32_t e = deg[x]; e < deg[x + 1]; ++e) {
                const uint32_t y = edges[e];
                const uint32_t a = wlab[x], b = wlab[y];
                if (a < b) { atomicMin(&wlab[y], a); ch = true; }
                else if (b < a) { atomicMin(&wlab[x], b); ch = true; }
            }
        }
        if (ch) s_flag[0] = 1;
        __syncthreads();
        for (int it = 0; it < 4; ++it) {
            for (uint32_t i = tid; i < NT; i += kPugNT) { const uint32_t v = tl[i]; const uint32_t l = wlab[v]; const uint32_t ll = wlab[l]; if (ll < l) wlab[v] = ll; }
            __syncthreads();
        }
        if (!s_flag[0]) break;
    }
    PUG_MARK(6);
    // a label may still point at a non-root after the last sweep; chase it
    for (uint32_t i = tid; i < NT; i += kPugNT) { const uint32_t v = tl[i]; uint32_t l = wlab[v]; while (wlab[l] != l) l = wlab[l]; comp_sorted[i] = ((uint64_t)l << kVidBits) | v; }
    __syncthreads();
    }
    tiled_bitonic_sort_by<kPugNT, 8192>(comp_sorted, NT, [](uint64_t a, uint64_t b) { return a > b; }, reinterpret_cast<uint64_t*>(s_big));
    uint32_t NC = 0;   // components of two or more vertices
    for (uint32_t base = 0; base < NT; base += kPugNT) {
        const uint32_t i = base + tid;
        const bool h = i < NT && (i == 0 || (comp_sorted[i] >> kVidBits) != (comp_sorted[i - 1] >> kVidBits));
        uint32_t tot;
        const uint32_t ex = block_excl_scan<kPugNT>(h, s_ws, tot);
        if (h) comp_start[NC + ex] = i;
        NC += tot;
    }
    if (tid == 0) comp_start[NC] = NT;
    __syncthreads();
    for (uint32_t c = tid; c < NC; c += kPugNT)  // vid -> position inside its component (components are short on average)
        if (comp_start[c + 1] - comp_start[c] <= 64)
            for (uint32_t i = comp_start[c]; i < comp_start[c + 1]; ++i) local_idx[(uint32_t)comp_sorted[i] & ((1u << kVidBits) - 1)] = i - comp_start[c];
    __syncthreads();
    auto vid_at = [&](uint32_t i) { return (uint32_t)comp_sorted[i] & ((1u << kVidBits) - 1); };
    // work lists: almost every component is a single vertex; list the others once instead of rescanning
    uint32_t* mid_list = reinterpret_cast<uint32_t*>(v_umi);  // slab B is dead: 2..64-vertex components
    uint32_t* big_list = v_cnt;                               // > 64 vertices or over the large-graph threshold
    if (tid < 2) s_flag[tid] = 0;
    __syncthreads();
    // (components of 3..8 vertices first in the list: they are covered eight to a wave, 6b')
    if (tid == 0) s_next = 0;   // (the work counter's LDS word is free until the next cell)
    __syncthreads();
    {
        uint32_t mine = 0;
        for (uint32_t c = tid; c < NC; c += kPugNT) { const uint32_t n = comp_start[c + 1] - comp_start[c]; mine += n >= 3 && n <= 8 && n <= C.large_thresh; }
        if (mine) atomicAdd(&s_next, mine);
    }
    __syncthreads();
    const uint32_t n_tiny = s_next;
    __syncthreads();
    if (tid == 0) s_next = 0;
    __syncthreads();
    for (uint32_t c = tid; c < NC; c += kPugNT) {
        const uint32_t n = comp_start[c + 1] - comp_start[c];
        if (n < 2 || (n == 2 && n <= C.large_thresh)) continue;  // singletons and pairs are resolved lane-parallel below
        if (n <= 8 && n <= C.large_thresh) mid_list[atomicAdd(&s_next, 1u)] = c;
        else if (n <= 64 && n <= C.large_thresh) mid_list[n_tiny + atomicAdd(&s_flag[0], 1u)] = c;
        else big_list[atomicAdd(&s_flag[1], 1u)] = c;
    }
    __syncthreads();
    const uint32_t n_mid = n_tiny + s_flag[0], n_big = s_flag[1];
    __syncthreads();

    PUG_MARK(7);
    // ---- 6a. single-vertex components: the label's genes (pugutils.rs:1262-1322) ----
    // A two-vertex component is always one molecule: it is weakly connected, so one of the two can reach the
    // other through a shared transcript and the greedy cover takes that 2-vertex arborescence first; its label
    // is the transcripts the two labels share (pugutils.rs:1161-1188) - never empty, an edge needs an overlap.
    // vertices without an edge: one molecule each, the label's genes.  Four vertices per thread and trip, their loads issued
    // together; labels of one or two refs (in the vertex record) never touch the chunk or a gene array.
    // Lone vertices whose label has more than two refs (one in ten) need the chunk and a gene lookup per ref - three dependent
    // reads deep; they are listed here and taken afterwards, a lane each, instead of holding up the wave that met them.
    uint32_t* long_list = c_minoff;   // (dead since phase 3)
    if (tid == 0) s_flag[0] = 0;
    __syncthreads();
    for (uint32_t v0 = tid; v0 - lane < V; v0 += 4 * kPugNT) {   // wave-uniform trip count (append_cols is a wave-wide call)
        uint4 q4[4];
        bool lone[4], shrt[4];
        uint32_t ga[4], gb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t v = v0 + j * kPugNT;
            lone[j] = v < V && !(deg[v + 1] > deg[v] || tch[v]);
            q4[j] = v < V ? vv[v] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t code = q4[j].x >> 20;
            shrt[j] = lone[j] && !C.gene_level && (code == 1 || code == 2);
            ga[j] = shrt[j] ? C.t2g[q4[j].z] : 0u;
            gb[j] = shrt[j] && code == 2 ? C.t2g[q4[j].w] : ga[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t col = 0xFFFFFFFFu, k0 = 0, k1 = 0;
            bool cls = false;
            if (lone[j] && shrt[j]) {
                const uint32_t lo = ga[j] < gb[j] ? ga[j] : gb[j], hi = ga[j] < gb[j] ? gb[j] : ga[j];
                col = molecule2_column(C, lo, hi, lo == hi ? 1u : 2u, cls);
                k0 = lo; k1 = hi;
            }
            const bool later = lone[j] && !shrt[j];
            const uint64_t lm = __ballot(later);
            if (lm) {
                const uint32_t leader = (uint32_t)__builtin_ctzll(lm);
                uint32_t at = 0;
                if (lane == leader) at = atomicAdd(&s_flag[0], (uint32_t)__popcll(lm));
                at = __builtin_amdgcn_readlane(at, (int)leader);
                if (later) long_list[at + (uint32_t)__popcll(lm & ((1ull << lane) - 1))] = v0 + j * kPugNT;
            }
            append_cols(C, col);   // (v0 - lane is wave-uniform: every lane of the wave gets here)
            append_class2(C, cls, k0, k1);
        }
    }
    __syncthreads();
    const uint32_t n_long = s_flag[0];
    for (uint32_t i = tid; i - lane < n_long; i += kPugNT) {
        uint32_t col = 0xFFFFFFFFu, k0 = 0, k1 = 0;
        bool cls = false;
        if (i < n_long) {
            const Lab l = vlab(long_list[i]);
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
                else emit_molecule(C, g, ng);
            }
        }
        append_cols(C, col);
        append_class2(C, cls, k0, k1);
    }
    for (uint32_t c = tid; c - lane < NC; c += kPugNT) {   // (wave-uniform trip count: append_cols is a wave-wide call)
        uint32_t col = 0xFFFFFFFFu, k0 = 0, k1 = 0;
        bool cls = false;
        const uint32_t n = c < NC ? comp_start[c + 1] - comp_start[c] : 0u;
        if (n == 2 && n <= C.large_thresh) {
            const Lab l = vlab(vid_at(comp_start[c])), l2 = vlab(vid_at(comp_start[c] + 1));
            if (l.n <= 4) {   // the shared transcripts of two short labels, in registers
                uint32_t g4[4];
                uint32_t k = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    g4[q] = 0xFFFFFFFFu;
                    if ((uint32_t)q < l.n) {
                        const uint32_t t = l.p[q] & 0x7FFFFFFFu;
                        if (lab_contains(l2, t)) {
#pragma unroll
                            for (int w = 0; w < 4; ++w) if ((uint32_t)w == k) g4[w] = t;
                            ++k;
                        }
                    }
                }
                const uint32_t ng = genes_of4(C, g4, k);
                col = molecule4_column(C, g4, ng, cls);
                k0 = g4[0]; k1 = g4[1];
            } else {
                uint32_t g[kMaxGenesPerLabel];
                uint32_t ng = 0;
                for (uint32_t j = 0; j < l.n && ng != 0xFFFFFFFFu; ++j) {
                    const uint32_t t = l.p[j] & 0x7FFFFFFFu;
                    if (!lab_contains(l2, t)) continue;
                    const uint32_t gid = C.gene_level ? t : C.t2g[t];
                    uint32_t q = 0;
                    while (q < ng && g[q] < gid) ++q;
                    if (q < ng && g[q] == gid) continue;
                    if (ng == kMaxGenesPerLabel) { ng = 0xFFFFFFFFu; break; }
                    for (uint32_t r = ng; r > q; --r) g[r] = g[r - 1];
                    g[q] = gid;
                    ++ng;
                }
                if (ng == 0xFFFFFFFFu && C.em)
                    emit_wide_class(C, l.n, [&](uint32_t j) -> uint32_t { const uint32_t t = l.p[j] & 0x7FFFFFFFu; return lab_contains(l2, t) ? t : 0xFFFFFFFFu; });
                else emit_molecule(C, g, ng);
            }
        }
        append_cols(C, col);
        append_class2(C, cls, k0, k1);
    }
    PUG_MARK(8);
    // ---- 6b. components of 3..64 vertices: one wave each, adjacency = one 64-bit mask per lane ----
    // What a wave needs of its component - per vertex the label and the adjacency mask - sits five dependent global reads deep
    // (list -> component bounds -> vertex id -> vertex record / edge range -> edge targets -> their local index), and a wave
    // working alone on one component pays that chain in full, a thousand times per cell.  So the gathering is done first,
    // by all threads over all such components at once (thread per vertex: the chains of a thousand vertices overlap), into
    // 32-byte records laid out component by component; the cover then reads its records with one access, the next
    // component's already on their way.
    uint32_t* mid_off = mid_list + n_mid;   // [n_mid + 1] first record of each listed component (slab B has the room)
    uint32_t S_mid = 0;
    for (uint32_t base = 0; base < n_mid; base += kPugNT) {
        const uint32_t ci = base + tid;
        const uint32_t n = ci < n_mid ? comp_start[mid_list[ci] + 1] - comp_start[mid_list[ci]] : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan<kPugNT>(n, s_ws, tot);
        if (ci < n_mid) mid_off[ci] = S_mid + ex;
        S_mid += tot;
    }
    if (tid == 0) { mid_off[n_mid] = S_mid; s_ebase = atomicAdd(A.epool_cursor, 9ull * S_mid + 4); }
    __syncthreads();
    if (s_ebase + 9ull * S_mid + 4 > A.epool_cap) { if (tid == 0) set_err(A.st, kErrPugPool, cell); return; }
    uint4* mrec = reinterpret_cast<uint4*>(A.epool + ((s_ebase + 3) & ~3ull));   // two per vertex: {vid, label length, ref0 | ptr lo, ref1 | ptr hi}, {ref2, ref3, adjacency}
    uint32_t* slot_comp = reinterpret_cast<uint32_t*>(mrec + 2 * (size_t)S_mid);
    for (uint32_t ci = tid; ci < n_mid; ci += kPugNT) {
        const uint32_t b0 = mid_off[ci], n = mid_off[ci + 1] - b0;
        for (uint32_t i = 0; i < n; ++i) slot_comp[b0 + i] = ci;
    }
    __syncthreads();
    for (uint32_t sl = tid; sl < S_mid; sl += kPugNT) {
        const uint32_t ci = slot_comp[sl];
        const uint32_t v = vid_at(comp_start[mid_list[ci]] + (sl - mid_off[ci]));
        const Lab l = vlab(v);
        uint32_t r0 = 0xFFFFFFFFu, r1 = 0xFFFFFFFFu, r2 = 0xFFFFFFFFu, r3 = 0xFFFFFFFFu;
        if (l.n <= 4) {   // labels are short: up to four refs travel in the record
            if (l.n > 0) r0 = l.p[0] & 0x7FFFFFFFu;
            if (l.n > 1) r1 = l.p[1] & 0x7FFFFFFFu;
            if (l.n > 2) r2 = l.p[2] & 0x7FFFFFFFu;
            if (l.n > 3) r3 = l.p[3] & 0x7FFFFFFFu;
        } else { const uint64_t pa = (uint64_t)(uintptr_t)l.p; r0 = (uint32_t)pa; r1 = (uint32_t)(pa >> 32); }
        uint64_t adj = 0;
        for (uint32_t e = deg[v]; e < deg[v + 1]; ++e) adj |= 1ull << local_idx[edges[e]];
        mrec[2 * (size_t)sl] = make_uint4(v, l.n, r0, r1);
        mrec[2 * (size_t)sl + 1] = make_uint4(r2, r3, (uint32_t)adj, (uint32_t)(adj >> 32));
    }
    __syncthreads();
    cover_tiny8<kPugNT / 64>(C, mrec, mid_off, n_tiny, wv, lane);   // 6b': components of 3..8 vertices, eight to a wave (afq_pug_common.h)
    cover_wave64<kPugNT / 64>(C, mrec, mid_off, n_tiny, n_mid, wv, lane);   // 9..64 vertices: a wave each
    __syncthreads();
    PUG_MARK(9);
    // ---- 6c. larger components, one at a time by the whole workgroup ----
    for (uint32_t ci = 0; ci < n_big; ++ci) {
        const uint32_t c = big_list[ci];
        const uint32_t c0 = comp_start[c], n = comp_start[c + 1] - c0;
        if (n > C.large_thresh) {
            // get_num_molecules_large_component (pugutils.rs:916-982): winner-take-all over the component's
            // (umi, gene, count) triplets.  Rare; thread 0 walks the triplets sorted by the workgroup.
            // triplets live in the edge pool: 4 words each (umi lo, umi hi, gene, count)
            if (tid == 0) { s_flag[1] = 0; }
            __syncthreads();
            // distinct genes of a label that has more of them than kMaxGenesPerLabel: first occurrences, found by looking back
            auto wide_genes = [&](const Lab& l, auto&& f) {
                for (uint32_t j = 0; j < l.n; ++j) {
                    const uint32_t tj = l.p[j] & 0x7FFFFFFFu;
                    const uint32_t gj = C.gene_level ? tj : C.t2g[tj];
                    bool first = true;
                    for (uint32_t q = 0; q < j && first; ++q) { const uint32_t tq = l.p[q] & 0x7FFFFFFFu; first = (C.gene_level ? tq : C.t2g[tq]) != gj; }
                    if (first) f(gj);
                }
            };
            // count triplets
            uint32_t cnt = 0;
            for (uint32_t i = tid; i < n; i += kPugNT) {
                const Lab l = vlab(vid_at(c0 + i));
                uint32_t g[kMaxGenesPerLabel];
                const uint32_t ng = genes_of(C, l.n, [&](uint32_t j) { return l.p[j] & 0x7FFFFFFFu; }, g);
                if (ng == 0xFFFFFFFFu) wide_genes(l, [&](uint32_t) { ++cnt; }); else cnt += ng;
            }
            uint32_t tot;
            (void)block_excl_scan<kPugNT>(cnt, s_ws, tot);
            if (tid == 0) s_ebase = atomicAdd(A.epool_cursor, 4ull * tot + 4);
            __syncthreads();
            if (s_ebase + 4ull * tot + 4 > A.epool_cap) { if (tid == 0) set_err(A.st, kErrPugPool, cell); return; }
            uint4* trip = reinterpret_cast<uint4*>(A.epool + ((s_ebase + 3) & ~3ull));
            if (tid == 0) s_flag[1] = 0;
            __syncthreads();
            for (uint32_t i = tid; i < n; i += kPugNT) {
                const uint32_t v = vid_at(c0 + i);
                const Lab l = vlab(v);
                uint32_t g[kMaxGenesPerLabel];
                const uint32_t ng = genes_of(C, l.n, [&](uint32_t j) { return l.p[j] & 0x7FFFFFFFu; }, g);
                if (ng == 0xFFFFFFFFu) {   // more genes than g[] holds: the distinct ones, found by looking back (rare)
                    uint32_t k = 0;
                    wide_genes(l, [&](uint32_t) { ++k; });
                    uint32_t o = atomicAdd(&s_flag[1], k);
                    wide_genes(l, [&](uint32_t gid) { trip[o++] = make_uint4((uint32_t)vv_umi[v], (uint32_t)(vv_umi[v] >> 32), gid, vv[v].x & 0xFFFFFu); });
                    continue;
                }
                const uint32_t o = atomicAdd(&s_flag[1], ng);
                for (uint32_t q = 0; q < ng; ++q) trip[o + q] = make_uint4((uint32_t)vv_umi[v], (uint32_t)(vv_umi[v] >> 32), g[q], vv[v].x & 0xFFFFFu);
            }
            __syncthreads();
            const uint32_t nt = s_flag[1];
            bitonic_sort_by<kPugNT>(trip, nt, [](const uint4& a, const uint4& b) {
                if (a.y != b.y) return a.y > b.y;
                if (a.x != b.x) return a.x > b.x;
                if (a.z != b.z) return a.z > b.z;
                return a.w > b.w;
            });
            if (tid == 0 && nt) {  // resolve_num_molecules_crlike_from_vec, pugutils.rs:644-749
                uint32_t best[kMaxGenesPerLabel];
                uint32_t nbest = 0, maxc = 0, aggr = 0;
                uint32_t cu_lo = trip[0].x, cu_hi = trip[0].y, cg = trip[0].z;
                bool wide = false;
                uint32_t run0 = 0;   // first triplet of the current UMI
                // a UMI whose tie set has more genes than best[] holds is a class of its own for the EM: the genes whose
                // summed count is the maximum, ascending as the triplets are, written straight into the label area
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
                    if (t.x != cu_lo || t.y != cu_hi) {
                        if (wide && C.em) emit_ties(run0, i, maxc); else emit_molecule(C, best, wide ? 0xFFFFFFFFu : nbest);
                        run0 = i;
                        cu_lo = t.x; cu_hi = t.y; cg = t.z;
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
            if (tid == 0) A.alt[cell] = 1;  // used_alternative_strategy, pugutils.rs:1070
            __syncthreads();
            continue;
        }
        if (n > kMaxBigComp) { if (tid == 0) set_err(A.st, kErrPugLimit, cell); return; }
        // multi-word cover: nw mask words, lane l of a wave holds word l; rows of the adjacency in the pool
        const uint32_t nw = (n + 63) / 64;
        if (tid == 0) s_ebase = atomicAdd(A.epool_cursor, 2ull * n * nw + 2);
        __syncthreads();
        if (s_ebase + 2ull * n * nw + 2 > A.epool_cap) { if (tid == 0) set_err(A.st, kErrPugPool, cell); return; }
        uint64_t* rows = reinterpret_cast<uint64_t*>(A.epool + ((s_ebase + 1) & ~1ull));
        for (uint32_t i = tid; i < n * nw; i += kPugNT) rows[i] = 0;
        for (uint32_t i = tid; i < n; i += kPugNT) local_idx[vid_at(c0 + i)] = i;
        __syncthreads();
        for (uint32_t i = tid; i < n; i += kPugNT) {
            const uint32_t v = vid_at(c0 + i);
            for (uint32_t e = deg[v]; e < deg[v + 1]; ++e) { const uint32_t y = local_idx[edges[e]]; rows[(size_t)i * nw + (y >> 6)] |= 1ull << (y & 63); }
        }
        if (tid < nw) s_mask[0][tid] = (tid + 1 < nw || (n & 63) == 0) ? ~0ull : ((1ull << (n & 63)) - 1);
        __syncthreads();
        for (;;) {
            // uncovered count
            uint32_t rem = 0;
            for (uint32_t w = 0; w < nw; ++w) rem += (uint32_t)__popcll(s_mask[0][w]);
            if (rem == 0) break;
            // every wave evaluates candidates v = k-th uncovered vertex for k = wv, wv+16, ...
            uint32_t my_best_sz = 0, my_best_v = 0xFFFFFFFFu;
            uint64_t my_best_word = 0;  // lane l: word l of this wave's best arborescence
            const uint64_t ucw = lane < nw ? s_mask[0][lane] : 0ull;
            uint32_t seen = 0;
            for (uint32_t w = 0; w < nw; ++w) {
                uint64_t bits = s_mask[0][w];
                for (; bits; bits &= bits - 1, ++seen) {
                    if (seen % (kPugNT / 64) != wv) continue;
                    const uint32_t v = w * 64 + (uint32_t)__builtin_ctzll(bits);
                    const Lab lv = vlab(vid_at(c0 + v));
                    uint64_t mvw = 0; uint32_t mv_sz = 0;
                    for (uint32_t j = 0; j < lv.n; ++j) {
                        const uint32_t t = lv.p[j] & 0x7FFFFFFFu;
                        // A_t: uncovered vertices whose label contains t (chunks of 64 vertices, lane = vertex)
                        uint64_t Aw = 0;
                        for (uint32_t cw = 0; cw < nw; ++cw) {
                            const uint32_t i = cw * 64 + lane;
                            const uint64_t ucb = __shfl((uint32_t)(ucw >> 32), (int)cw);
                            const uint64_t uca = __shfl((uint32_t)ucw, (int)cw);
                            const uint64_t ucword = (ucb << 32) | uca;
                            const bool in = i < n && ((ucword >> lane) & 1ull) && lab_contains(vlab(vid_at(c0 + i)), t);
                            const uint64_t word = __ballot(in);
                            if (lane == cw) Aw = word;
                        }
                        uint64_t Rw = (lane == (v >> 6)) ? (1ull << (v & 63)) : 0ull, Fw = Rw;
                        for (;;) {
                            // N = OR of the rows of the frontier vertices
                            uint64_t Nw = 0;
                            for (uint32_t fw = 0; fw < nw; ++fw) {
                                const uint32_t flo = __shfl((uint32_t)Fw, (int)fw), fhi = __shfl((uint32_t)(Fw >> 32), (int)fw);
                                uint64_t fb = ((uint64_t)fhi << 32) | flo;
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
                        for (int d = 32; d > 0; d >>= 1) sz += __shfl_xor(sz, d);
                        if (sz > mv_sz) { mv_sz = sz; mvw = Rw; }
                    }
                    if (mv_sz > my_best_sz) { my_best_sz = mv_sz; my_best_v = v; my_best_word = mvw; }
                }
            }
            if (lane == 0) { s_bestv[wv] = my_best_v; s_bestsz[wv] = my_best_sz; }
            __syncthreads();
            // winner: largest size, then smallest vertex (= first in ascending scan order)
            uint32_t win = 0;
            for (uint32_t w = 1; w < kPugNT / 64; ++w)
                if (s_bestsz[w] > s_bestsz[win] || (s_bestsz[w] == s_bestsz[win] && s_bestv[w] < s_bestv[win])) win = w;
            if (s_bestsz[win] == 0) { if (tid == 0) s_cnt[3] = kErrPugLimit; break; }
            if (wv == win && lane < nw) s_mask[1][lane] = my_best_word;
            __syncthreads();
            if (wv == 0) {
                // common transcripts of the arborescence -> genes
                uint32_t fv = 0xFFFFFFFFu;
                for (uint32_t w = 0; w < nw && fv == 0xFFFFFFFFu; ++w) if (s_mask[1][w]) fv = w * 64 + (uint32_t)__builtin_ctzll(s_mask[1][w]);
                const Lab lf = vlab(vid_at(c0 + fv));
                uint32_t g[kMaxGenesPerLabel];
                uint32_t ng = 0;
                bool wide = false;
                for (uint32_t j = 0; j < lf.n; ++j) {
                    const uint32_t t = lf.p[j] & 0x7FFFFFFFu;
                    bool all = true;
                    for (uint32_t cw = 0; cw < nw; ++cw) {
                        const uint32_t i = cw * 64 + lane;
                        const bool inb = i < n && ((s_mask[1][cw] >> lane) & 1ull);
                        const bool miss = inb && !lab_contains(vlab(vid_at(c0 + i)), t);
                        if (__any(miss)) { all = false; break; }
                    }
                    if (!all) continue;
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
                        emit_wide_class(C, lf.n, [&](uint32_t j) -> uint32_t {
                            const uint32_t t = lf.p[j] & 0x7FFFFFFFu;
                            for (uint32_t cw = 0; cw < nw; ++cw)
                                for (uint64_t m = s_mask[1][cw]; m; m &= m - 1)
                                    if (!lab_contains(vlab(vid_at(c0 + cw * 64 + (uint32_t)__builtin_ctzll(m))), t)) return 0xFFFFFFFFu;
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
    __syncthreads();
    PUG_MARK(10);
#ifdef AFQ_PUG_TIMING
    if (tid == 0 && blockIdx.x < 4 && (work % 1024) < 4) {
        auto ms = [&](int a, int b) { return (double)(tmark[b] - tmark[a]) / 1e5; };
        printf("pug sort: sample-load=%.2f sample-sort=%.2f classify=%.2f scan+scatter=%.2f wave-sorts=%.2f ms\n", (double)(tsort[0] - tmark[0]) / 1e5, (double)(tsort[1] - tsort[0]) / 1e5, (double)(tsort[2] - tsort[1]) / 1e5, (double)(tsort[3] - tsort[2]) / 1e5, (double)(tsort[4] - tsort[3]) / 1e5);
        printf("pug lds route: part+scatter=%.2f clear=%.2f insert=%.2f own=%.2f foreign=%.2f ms; probes=%llu run steps=%llu pairs=%u\n", tacc[0] / 1e5, tacc[1] / 1e5, tacc[2] / 1e5, tacc[3] / 1e5, tacc[4] / 1e5, tacc[7], tacc[6], n_pairs);
        printf("pug cell R=%u V=%u K=%u NC=%u nmid=%u nbig=%u: sort=%.2f classes=%.2f umis=%.2f verts=%.2f | htab=%.2f bloom=%.2f cand=%.2f (2pass=%.2f) match+rule=%.2f fill=%.2f | wcc=%.2f comps=%.2f 6a=%.2f mid=%.2f big=%.2f out=%.2f | total=%.2f NCAND=%u E=%u\n",
               R, V, K, NC, n_mid, n_big, ms(0, 1), ms(1, 2), ms(2, 3), 0.0, ms(3, 11), ms(11, 4), ms(4, 12), ms(12, 13), ms(13, 14), ms(14, 5),
               ms(5, 6), ms(6, 7), ms(7, 8), ms(8, 9), 0.0, ms(9, 10), ms(0, 10), NCAND, E);
    }
#endif
    if (s_cnt[3]) { if (tid == 0) set_err(A.st, s_cnt[3], cell); return; }
    if (tid == 0) {
        A.cell_ncols[cell] = s_cnt[0];
        if (A.lab_cnt) { A.lab_cnt[2 * cell] = s_cnt[1]; A.lab_cnt[2 * cell + 1] = s_cnt[2]; }
    }
  }
}

void launch_pug(hipStream_t s, const PugCellArgs& a, uint32_t n_blocks) {
    if (!n_blocks) return;
    hipLaunchKernelGGL(k_pug_cell, dim3(n_blocks), dim3(kPugNT), 0, s, a);
}

// one workgroup per CU fits (LDS filter + 1024 threads)
uint32_t pug_max_blocks() {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    return (uint32_t)(cus > 0 ? cus : 256);
}

uint64_t pug_scratch_words(uint32_t nrec, uint32_t n_ref, bool gene_level) {
    const uint64_t R = nrec;
    uint64_t ht_cap = 64;
    while (ht_cap < 2 * R) ht_cap <<= 1;
    return (6 * R + 4 * R + 6 * R + (R + 1) + 4 * R + (R + 2) + (R + 2) + 1 + 2 * ht_cap + (gene_level ? R + 2 + (uint64_t)n_ref + 2 : 0) + 16 + 3) & ~3ull;   // multiple of 4 words: slab A holds 16-byte records
}

}  // namespace afq
