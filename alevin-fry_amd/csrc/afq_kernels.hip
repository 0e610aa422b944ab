// afq_kernels.hip - the bucket pipeline of winner-take-all ("cr-like") resolution on gfx950 (CDNA4, wave64)
// (replaces src/quant.rs:469-657 / src/pugutils.rs:644-850 / src/utils.rs:673-756 of the reference;
// semantics: SURVEY.md appendix B.2):
//   k_hist, k_bucket_scan, k_scatter   keys -> per-(cell, UMI-hash bucket) ranges (LDS histogram / multisplit)
//   k_bucket_desc, k_resolve           one wave per bucket: LDS hash table keyed by UMI (or register bitonic
//                                      sort), per-UMI arg-max with ties, USA slot rules, cr-like-em class staging
//   k_resolve_mid, k_resolve_big       the sort path for buckets beyond one wave / beyond LDS
//   k_cell_hist                        per-cell LDS histogram of the resolved columns -> (column, count) pairs
//   k_compact                          per-cell pairs -> final CSR
//   k_atac_dedup                       ATAC per-cell fragment de-duplication
// The decode lives in afq_decode.hip, the EM in afq_em.hip, parsimony in afq_pug.hip.
// Integer/byte work bound by HBM and LDS; no MFMA anywhere by design.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "afq_common.h"
#include "afq_kernels.h"
#include "afq_prims.h"

namespace afq {
// ---------------------------------------------------------------------------
// Bucket histogram.  Device-scope atomics leave the XCD (every one is a fabric
// transaction on this 8-XCD part: rocprof WRITE_SIZE showed 3-5x the payload when
// they were issued per record), so counts are first combined in LDS over a tile
// of kTileKeys keys and flushed with one atomic per non-empty bucket per tile.
constexpr uint32_t kTileKeys = kScatterTileHost;  // 2048
#ifndef AFQ_SCATTER_RUN
#define AFQ_SCATTER_RUN 16
#endif
constexpr uint32_t kScatterRun = AFQ_SCATTER_RUN;   // consecutive tiles one XCD takes (k_scatter)
constexpr uint32_t kLdsBins = 2048;               // buckets per cell the LDS paths can hold

// tile -> (cell, tile index inside the cell): a table the planner uploads with the batch.  (It used to be a binary
// search over the cells' tile prefix by thread 0 - fourteen dependent L2 round trips and a barrier in front of every
// 2048-key tile, more time than the tile's own work.)
__global__ __launch_bounds__(256) void k_hist(const uint2* __restrict__ tile_desc,
                                             const CellMeta* __restrict__ meta,
                                             const uint32_t* __restrict__ cell_nkeys,
                                             const uint64_t* __restrict__ keys0, uint32_t* __restrict__ bucket_cnt) {
    __shared__ uint32_t s_hist[kLdsBins];
    const uint2 td = tile_desc[blockIdx.x];
    const uint32_t cell = td.x, lt = td.y;
    const CellMeta m = meta[cell];
    const uint32_t nk = mode_is_pug(m.mode) ? 0u : cell_nkeys[cell];  // PUG cells emit reads, not keys
    const uint32_t t0 = lt * kTileKeys;
    if (t0 >= nk) return;
    const uint32_t t1 = min(nk, t0 + kTileKeys);
    const uint64_t* src = keys0 + m.key_off;
    uint32_t* gcnt = bucket_cnt + m.bucket_base;
    const uint32_t nb = 1u << m.lg_nb;
    if (nb > kLdsBins) {  // giant cell: straight to global
        for (uint32_t i = t0 + threadIdx.x; i < t1; i += 256) atomicAdd(&gcnt[bucket_of(src[i] >> kGeneBits, m.lg_nb)], 1u);
        return;
    }
    for (uint32_t b = threadIdx.x; b < nb; b += 256) s_hist[b] = 0;
    __syncthreads();
    for (uint32_t i = t0 + threadIdx.x; i < t1; i += 256) atomicAdd(&s_hist[bucket_of(src[i] >> kGeneBits, m.lg_nb)], 1u);
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nb; b += 256) {
        const uint32_t c = s_hist[b];
        if (c) atomicAdd(&gcnt[b], c);
    }
}

// per-cell exclusive scan of bucket counts (in place): wave per multi-bucket cell
__global__ __launch_bounds__(256) void k_bucket_scan(const uint32_t* __restrict__ multi_cells, uint32_t n_multi,
                                                    const CellMeta* __restrict__ meta,
                                                    uint32_t* __restrict__ bucket_cnt) {
    const uint32_t ci = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ci >= n_multi) return;
    const CellMeta m = meta[multi_cells[ci]];
    const uint32_t nb = 1u << m.lg_nb;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nb; base += 64) {
        const uint32_t i = base + lane_id();
        uint32_t c = i < nb ? bucket_cnt[m.bucket_base + i] : 0u, tot;
        uint32_t ex = wave_excl_scan(c, tot);
        if (i < nb) bucket_cnt[m.bucket_base + i] = carry + ex;
        carry += tot;
    }
}

// ---------------------------------------------------------------------------
// keys0 -> keys1 grouped by bucket: LDS multisplit of a kTileKeys tile.  Ranks
// inside a bucket come from LDS atomics, one global atomic per non-empty bucket
// reserves the tile's range, and the tile is written out bucket-major so the
// stores of a bucket's run are contiguous.  After the kernel cursor[b] = end
// offset of bucket b inside its cell's region (start = previous bucket's end).
// BINS: buckets per cell the instance ranks in LDS (a cell with more goes key by key).  Nearly every batch is served
// by the 512-bin instance, whose 20 KiB of LDS let eight workgroups share a CU instead of five.
template <uint32_t BINS>
__global__ __launch_bounds__(256) void k_scatter(const uint2* __restrict__ tile_desc,
                                                const CellMeta* __restrict__ meta,
                                                const uint32_t* __restrict__ cell_nkeys,
                                                const uint64_t* __restrict__ keys0, uint64_t* __restrict__ keys1,
                                                uint32_t* __restrict__ cursor, uint32_t* __restrict__ slab_ovf, uint32_t n_tiles) {
    constexpr uint32_t E = kTileKeys / 256;
    __shared__ uint64_t s_keys[kTileKeys];
    __shared__ uint32_t s_cnt[BINS];   // per-bucket count, then tile-local exclusive offset
    __shared__ uint32_t s_base[BINS];  // global position of the tile's first key of the bucket
    __shared__ uint32_t s_ws[4];
    // Workgroups are dealt to the eight XCDs round-robin, each XCD with an L2 of its own; the tiles go to them in runs of sixteen
    // (a median cell), so that the runs a cell's consecutive tiles add to one bucket meet in one L2 instead of reaching memory
    // as partial lines from two.  (The grid is a multiple of 128.)
    constexpr uint32_t kRun = kScatterRun;   // (the grid is a multiple of 8 * kRun: launch_scatter)
    const uint32_t xr = blockIdx.x / 8, tile = ((xr / kRun) * 8 + blockIdx.x % 8) * kRun + xr % kRun;
    if (tile >= n_tiles) return;
    const uint2 td = tile_desc[tile];
    const uint32_t cell = td.x, lt = td.y;
    const CellMeta m = meta[cell];
    const uint32_t nk = mode_is_pug(m.mode) ? 0u : cell_nkeys[cell];  // PUG cells emit reads, not keys
    const uint32_t t0 = lt * kTileKeys;
    if (t0 >= nk) return;
    const uint32_t t1 = min(nk, t0 + kTileKeys);
    const uint64_t* src = keys0 + m.key_off;
    // Fixed slabs (m.slab_cap != 0): bucket b owns slots [b * cap, (b + 1) * cap) of the cell's keys1 region and the cursors
    // start at zero - no counting pass (k_hist) and no scan in front of this kernel.  A bucket that outgrows its slab
    // (one UMI with hundreds of reads) flags the cell; k_fix_slabs then places that cell exactly, from the counts the
    // cursors hold by then.  cap = 0: the exact layout behind k_hist + k_bucket_scan (cursors = exclusive offsets).
    const uint32_t cap = m.slab_cap;
    uint64_t* dst = keys1 + (cap ? m.k1_off : m.key_off);
    uint32_t* gcur = cursor + m.bucket_base;
    const uint32_t nb = 1u << m.lg_nb;
    if (nb > BINS) {  // giant cell: per-key global atomics
        bool over = false;
        for (uint32_t i = t0 + threadIdx.x; i < t1; i += 256) {
            const uint64_t key = src[i];
            const uint32_t b = bucket_of(key >> kGeneBits, m.lg_nb);
            const uint32_t pos = atomicAdd(&gcur[b], 1u);
            if (!cap) dst[pos] = key;
            else if (pos < cap) dst[(uint64_t)b * cap + pos] = key;
            else over = true;
        }
        if (over) atomicOr(&slab_ovf[cell], 1u);
        return;
    }
    for (uint32_t b = threadIdx.x; b < nb; b += 256) s_cnt[b] = 0;
    __syncthreads();
    uint64_t key[E];
    uint32_t rank[E];   // bucket << 16 | rank of the key among the tile's keys of that bucket (both < 2^11)
#pragma unroll
    for (uint32_t e = 0; e < E; ++e) {
        const uint32_t i = t0 + e * 256 + threadIdx.x;
        key[e] = kKeySentinel;
        rank[e] = 0;
        if (i < t1) {
            key[e] = AFQ_LD_SCATTER(&src[i]);
            const uint32_t b = bucket_of(key[e] >> kGeneBits, m.lg_nb);
            rank[e] = (b << 16) | atomicAdd(&s_cnt[b], 1u);
        }
    }
    __syncthreads();
    // exclusive scan of the bucket counts (tile-local offsets) + global reservation
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nb; base += 256) {
        const uint32_t b = base + threadIdx.x;
        const uint32_t c = b < nb ? s_cnt[b] : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan<256>(c, s_ws, tot);
        if (b < nb) {
            s_cnt[b] = carry + ex;
            const uint32_t at = c ? atomicAdd(&gcur[b], c) : 0u;
            s_base[b] = at;
            if (cap && at + c > cap) atomicOr(&slab_ovf[cell], 1u);
        }
        carry += tot;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t e = 0; e < E; ++e) {
        const uint32_t i = t0 + e * 256 + threadIdx.x;
        if (i < t1) s_keys[s_cnt[rank[e] >> 16] + (rank[e] & 0xFFFFu)] = key[e];
    }
    __syncthreads();
    const uint32_t nt = t1 - t0;
    for (uint32_t i = threadIdx.x; i < nt; i += 256) {
        const uint64_t kx = s_keys[i];
        const uint32_t b = bucket_of(kx >> kGeneBits, m.lg_nb);
        const uint32_t pos = s_base[b] + (i - s_cnt[b]);
        if (!cap) dst[pos] = kx;
        else if (pos < cap) dst[(uint64_t)b * cap + pos] = kx;
    }
}

// Cells whose fixed slabs overflowed (normally none): the cursors hold every bucket's true count, so the exact layout is
// one scan away - buckets back to back from the start of the cell's keys1 region, keys placed one atomic each out of
// keys0 (still intact).  Afterwards cursor[b] = end offset of bucket b, as after the exact path.
__global__ __launch_bounds__(256) void k_fix_slabs(const uint32_t* __restrict__ multi_cells, uint32_t n_multi,
                                                  const CellMeta* __restrict__ meta, const uint32_t* __restrict__ cell_nkeys,
                                                  const uint64_t* __restrict__ keys0, uint64_t* __restrict__ keys1,
                                                  uint32_t* __restrict__ cursor, const uint32_t* __restrict__ slab_ovf) {
    __shared__ uint32_t s_ws[4];
    for (uint32_t ci = blockIdx.x; ci < n_multi; ci += gridDim.x) {
        const uint32_t cell = multi_cells[ci];
        if (!slab_ovf[cell]) continue;
        const CellMeta m = meta[cell];
        if (!m.slab_cap || mode_is_pug(m.mode)) continue;
        const uint32_t nb = 1u << m.lg_nb, nk = cell_nkeys[cell];
        uint32_t* gcur = cursor + m.bucket_base;
        uint32_t carry = 0;
        for (uint32_t base = 0; base < nb; base += 256) {
            const uint32_t b = base + threadIdx.x;
            const uint32_t c = b < nb ? gcur[b] : 0u;
            uint32_t tot;
            const uint32_t ex = block_excl_scan<256>(c, s_ws, tot);
            if (b < nb) gcur[b] = carry + ex;
            carry += tot;
        }
        __threadfence();
        __syncthreads();
        const uint64_t* src = keys0 + m.key_off;
        uint64_t* dst = keys1 + m.k1_off;
        for (uint32_t i = threadIdx.x; i < nk; i += 256) {
            const uint64_t key = src[i];
            dst[atomicAdd(&gcur[bucket_of(key >> kGeneBits, m.lg_nb)], 1u)] = key;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// ---------------------------------------------------------------------------
// resolve core, shared by the LDS and the global-scratch variants.
struct ResolveCfg {
    uint32_t usa, num_rows, uo, ao;
    uint32_t mode;  // filled per bucket from its descriptor
    uint32_t pa;    // --sa-model prefer-ambig (USA only): reads of a UMI are tallied per gene, S and U together
    uint32_t sort_only;   // the batch's reads carry many genes each: the UMI table's three counters per slot would overflow for
                          // most UMIs (and the bucket then be sorted anyway) - every bucket takes the sort path straight away
};
__device__ __forceinline__ bool mode_is_em(uint32_t mode) { return mode == kModeCrLikeEm; }


// keys[0..n) sorted ascending.  Builds run starts, then for every UMI picks the
// winner / tie set and maps it to an output column (non-USA: unique winner only,
// src/quant.rs:563-565; USA: src/utils.rs:688-753 == src/quant.rs:557-605).
// emit(col) is called once per resolved UMI.
// In the EM modes (cr-like-em) a UMI whose tie set is not a single output column is kept as a
// gene-level equivalence class (quant.rs:882-924): lab_alloc(nb) returns room for its nb tie genes.
template <int NT, typename RunT, typename Emit, typename LabAlloc>
__device__ __forceinline__ void resolve_sorted(const uint64_t* keys, uint32_t n, RunT* run_start,
                                               uint32_t* ws, const ResolveCfg& rc, Emit&& emit, LabAlloc&& lab_alloc) {
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n; base += NT) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t f = (i < n) && (i == 0 || keys[i] != keys[i - 1]);
        uint32_t tot;
        const uint32_t ex = block_excl_scan<NT>(f, ws, tot);
        if (f) run_start[carry + ex] = (RunT)i;
        carry += tot;
    }
    const uint32_t nruns = carry;
    __syncthreads();
    auto run_end = [&](uint32_t q) -> uint32_t { return q + 1 < nruns ? (uint32_t)run_start[q + 1] : n; };  // no sentinel slot needed
    // reads behind run q's claim on its UMI.  winner-take-all: the run's own (pugutils.rs:644-749); prefer-ambig: the
    // run's plus those of the same gene's other splicing state, ids 2k / 2k+1 being neighbours in key order - a gene
    // seen both ways then wins or ties with both ids in the label (pugutils.rs:505-641)
    auto cnt = [&](uint32_t q) -> uint32_t {
        const uint32_t s = run_start[q];
        uint32_t c = run_end(q) - s;
        if (rc.pa) {
            const uint64_t k = keys[s];
            if (!(k & 1)) {
                if (q + 1 < nruns) { const uint32_t s2 = run_start[q + 1]; if (keys[s2] == k + 1) c += run_end(q + 1) - s2; }
            } else if (q > 0) {
                const uint32_t s0 = run_start[q - 1];
                if (keys[s0] == k - 1) c += s - s0;
            }
        }
        return c;
    };
    if (rc.mode == kModeTrivial) {
        // `trivial`: every distinct (umi, gene) of the single-gene reads is one molecule of that gene
        // (pugutils.rs:899-907); the column is the raw gene id (counts has num_genes entries, pugutils.rs:858).
        for (uint32_t r = threadIdx.x; r < nruns; r += NT) emit((uint32_t)keys[run_start[r]] & kGeneMask);
        return;
    }
    for (uint32_t r = threadIdx.x; r < nruns; r += NT) {
        const uint64_t umi = keys[run_start[r]] >> kGeneBits;
        if (r > 0 && (keys[run_start[r - 1]] >> kGeneBits) == umi) continue;  // not the UMI's first run
        uint32_t maxc = 0;
        for (uint32_t q = r; q < nruns; ++q) {
            const uint32_t s = run_start[q];
            if ((keys[s] >> kGeneBits) != umi) break;
            const uint32_t c = cnt(q);
            maxc = c > maxc ? c : maxc;
        }
        uint32_t nb = 0, g1 = 0, g2 = 0, nsp = 0, first_sp = 0;
        bool prev_first_sp = false, sp_followed = false;
        for (uint32_t q = r; q < nruns; ++q) {
            const uint32_t s = run_start[q];
            const uint64_t kq = keys[s];
            if ((kq >> kGeneBits) != umi) break;
            if (cnt(q) != maxc) continue;
            const uint32_t g = (uint32_t)kq & kGeneMask;
            ++nb;
            if (nb == 1) g1 = g;
            if (nb == 2) g2 = g;
            if (prev_first_sp) { sp_followed = same_gene(first_sp, g); prev_first_sp = false; }
            if (is_spliced(g)) {
                ++nsp;
                if (nsp == 1) { first_sp = g; prev_first_sp = true; }
            }
        }
        uint32_t col = 0xFFFFFFFFu;
        if (mode_is_em(rc.mode)) {
            // single-label classes are counted as columns; for USA that is a label whose S/U/A rewrite
            // (utils.rs:865-925) has one entry: one gene id, or S and U of the same gene
            if (nb == 1) col = !rc.usa ? g1 : (is_spliced(g1) ? (g1 >> 1) : rc.uo + (g1 >> 1));
            else if (rc.usa && nb == 2 && same_gene(g1, g2)) col = rc.ao + (g1 >> 1);
            else {
                uint32_t* dst = lab_alloc(nb);
                if (dst) {
                    uint32_t w = 0;
                    for (uint32_t q = r; q < nruns; ++q) {
                        const uint32_t s = run_start[q];
                        const uint64_t kq = keys[s];
                        if ((kq >> kGeneBits) != umi) break;
                        if (cnt(q) == maxc) dst[w++] = (uint32_t)kq & kGeneMask;
                    }
                }
            }
        } else if (!rc.usa) {
            if (nb == 1) col = g1;
        } else if (nb == 1) {
            col = is_spliced(g1) ? (g1 >> 1) : rc.uo + (g1 >> 1);
        } else if (nb == 2) {
            if (same_gene(g1, g2)) col = rc.ao + (g1 >> 1);
            else if (is_spliced(g1) && !is_spliced(g2)) col = g1 >> 1;
            else if (!is_spliced(g1) && is_spliced(g2)) col = g2 >> 1;
        } else if (nb <= 10) {
            if (nsp == 1) col = sp_followed ? rc.ao + (first_sp >> 1) : (first_sp >> 1);
        }
        if (col != 0xFFFFFFFFu) emit(col);
    }
}

// ---------------------------------------------------------------------------
// Per-bucket descriptors: everything a resolve workgroup needs in one 32-byte load
// (instead of the dependent chain bucket -> cell -> meta -> cursor -> keys).
struct BucketDesc {
    uint64_t src_off;  // first key of the bucket: slot in keys1 (multi-bucket cell) or keys0 (single)
    uint64_t out_off;  // the cell's key_off (column list / pair staging live in its keys0 slots)
    uint32_t n;        // keys in the bucket
    uint32_t cell;
    uint32_t mode_single;  // kMode* of the cell | single << 8 (1: the bucket is the whole cell)
    uint32_t n_ref;        // the cell's key capacity (locates its label area)
};

__global__ void k_bucket_desc(const CellMeta* __restrict__ meta, const uint32_t* __restrict__ bucket_cell,
                              const uint32_t* __restrict__ cell_nkeys, const uint32_t* __restrict__ cursor,
                              const uint32_t* __restrict__ slab_ovf, uint32_t n_buckets, BucketDesc* __restrict__ desc) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_buckets) return;
    const uint32_t cell = bucket_cell[b];
    const CellMeta m = meta[cell];
    BucketDesc d;
    d.cell = cell; d.out_off = m.key_off; d.n_ref = m.n_ref;
    if (m.lg_nb == 0) { d.mode_single = m.mode | 0x100u; d.src_off = m.key_off; d.n = mode_is_pug(m.mode) ? 0u : cell_nkeys[cell]; }
    else if (m.slab_cap && !slab_ovf[cell]) {   // fixed slabs: the cursor is the bucket's count
        d.mode_single = m.mode; d.src_off = m.k1_off + (uint64_t)(b - m.bucket_base) * m.slab_cap; d.n = mode_is_pug(m.mode) ? 0u : cursor[b];
    } else {
        const uint32_t beg = (b == m.bucket_base) ? 0u : cursor[b - 1];
        d.mode_single = m.mode; d.src_off = (m.slab_cap ? m.k1_off : m.key_off) + beg; d.n = mode_is_pug(m.mode) ? 0u : cursor[b] - beg;
    }
    desc[b] = d;
}

// Where a cell's gene-level classes (EM modes) are collected: lab[2*key_off ...] holds the label
// words of the cell's ambiguous molecules, followed (from word n_ref+1) by (offset,len) descriptors.
struct LabArea {
    uint32_t* lab;      // [2 * total key slots] or null outside the EM modes
    uint32_t* lab_cnt;  // per cell: [2*cell] words used, [2*cell+1] molecules
};

__device__ __forceinline__ uint32_t* lab_alloc_global(const LabArea& la, uint32_t cell, uint64_t key_off, uint32_t n_ref,
                                                      uint32_t nb) {
    const uint32_t off = atomicAdd(&la.lab_cnt[2 * cell], nb), di = atomicAdd(&la.lab_cnt[2 * cell + 1], 1u);
    uint32_t* gw = la.lab + 2 * key_off;
    uint32_t* gd = gw + n_ref + 1;
    gd[2 * di] = off; gd[2 * di + 1] = nb;
    return gw + off;
}


#ifdef AFQ_RESOLVE_TIMING
__device__ unsigned long long g_dbg[8];
#define RT_MARK(i) do { if (threadIdx.x == 0 && (blockIdx.x & 1023) == 0) { unsigned long long t_ = clock64(); atomicAdd(&g_dbg[i], t_ - tprev_); atomicAdd(&g_dbg[4 + (i & 3)], 1ull); tprev_ = t_; } } while (0)
#else
#define RT_MARK(i) do {} while (0)
#endif

// What follows the resolution of one bucket: s_cols[0..nc) are its molecules' columns.  A bucket of a
// multi-bucket cell appends them to the cell's column list (one reservation per bucket); a single-bucket
// cell is finished here (columns sorted, run-length counted, written as (column,count) pairs).
template <int NT>
__device__ __forceinline__ void bucket_tail(const BucketDesc& d, uint64_t* __restrict__ keys0,
                                            uint32_t* __restrict__ cell_ncols, uint32_t* __restrict__ nnz,
                                            const uint32_t* s_cols, uint32_t nc, uint32_t* s_sorted, uint16_t* s_run,
                                            uint32_t* s_ws, uint32_t* s_misc) {
    const bool single = (d.mode_single >> 8) != 0;
    if (!single) {
        if (nc == 0) return;
        if (threadIdx.x == 0) s_misc[1] = atomicAdd(&cell_ncols[d.cell], nc);
        __syncthreads();
        uint32_t* out = reinterpret_cast<uint32_t*>(keys0 + d.out_off) + s_misc[1];  // keys0 slots are dead after k_scatter
        for (uint32_t i = threadIdx.x; i < nc; i += NT) out[i] = s_cols[i];
        return;
    }
    // single-bucket cell: sort the columns, run-length count, write (column,count) pairs
    block_sort_any<NT, uint32_t>(s_cols, nc, s_sorted, 0xFFFFFFFFu);
    uint2* out = reinterpret_cast<uint2*>(keys0 + d.out_off);
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nc; base += NT) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t f = (i < nc) && (i == 0 || s_sorted[i] != s_sorted[i - 1]);
        uint32_t tot;
        const uint32_t ex = block_excl_scan<NT>(f, s_ws, tot);
        if (f) s_run[carry + ex] = (uint16_t)i;
        carry += tot;
    }
    if (threadIdx.x == 0) nnz[d.cell] = carry;
    __syncthreads();
    for (uint32_t h = threadIdx.x; h < carry; h += NT) {
        const uint32_t a = s_run[h], e = h + 1 < carry ? (uint32_t)s_run[h + 1] : nc;
        out[h] = make_uint2(s_sorted[a], e - a);
    }
}

// hand the bucket's staged ambiguous molecules (s_misc[2] label words, s_misc[3] molecules) to the cell's label area
template <int NT>
__device__ __forceinline__ void flush_bucket_labels(const BucketDesc& d, const LabArea& la, const uint32_t* s_lab,
                                                    const uint32_t* s_ldesc, uint32_t* s_misc) {
    const uint32_t lw = s_misc[2], ln = s_misc[3];
    if (threadIdx.x == 0) {
        s_misc[4] = atomicAdd(&la.lab_cnt[2 * d.cell], lw);
        s_misc[5] = atomicAdd(&la.lab_cnt[2 * d.cell + 1], ln);
    }
    __syncthreads();
    uint32_t* gw = la.lab + 2 * d.out_off;
    uint32_t* gd = gw + d.n_ref + 1;
    for (uint32_t i = threadIdx.x; i < lw; i += NT) gw[s_misc[4] + i] = s_lab[i];
    for (uint32_t i = threadIdx.x; i < ln; i += NT) {
        gd[2 * (s_misc[5] + i)] = s_ldesc[2 * i] + s_misc[4];
        gd[2 * (s_misc[5] + i) + 1] = s_ldesc[2 * i + 1];
    }
}

// Sort + resolve one bucket held in LDS.  NT threads, up to NT*8 keys.
template <int NT>
__device__ __forceinline__ void resolve_bucket_lds(const BucketDesc& d, uint64_t* __restrict__ keys0,
                                                   const uint64_t* __restrict__ keys1,
                                                   uint32_t* __restrict__ cell_ncols, uint32_t* __restrict__ nnz,
                                                   DevStatus* st, const ResolveCfg& rc, const LabArea& la,
                                                   uint64_t* s_keys, uint16_t* s_run, uint32_t* s_cols,
                                                   uint32_t* s_lab, uint32_t* s_ldesc, uint32_t* s_ws,
                                                   uint32_t* s_misc /* [6] */) {
    const uint32_t n = d.n;
    const bool single = (d.mode_single >> 8) != 0;
    const uint64_t* src = (single ? keys0 : keys1) + d.src_off;
    ResolveCfg rcb = rc;
    rcb.mode = d.mode_single & 0xFFu;
    if (threadIdx.x < 6) s_misc[threadIdx.x] = 0;
#ifdef AFQ_RESOLVE_TIMING
    unsigned long long tprev_ = clock64();
#endif
    block_sort_any<NT, uint64_t>(src, n, s_keys, kKeySentinel);
    resolve_sorted<NT>(s_keys, n, s_run, s_ws, rcb, [&](uint32_t col) {
        if (col >= rc.num_rows) { set_err(st, kErrSlotRange, d.cell); return; }
        s_cols[atomicAdd(&s_misc[0], 1u)] = col;
    }, [&](uint32_t nb) -> uint32_t* {
        if (!la.lab) return nullptr;
        if (!s_lab) return lab_alloc_global(la, d.cell, d.out_off, d.n_ref, nb);
        const uint32_t off = atomicAdd(&s_misc[2], nb), di = atomicAdd(&s_misc[3], 1u);
        s_ldesc[2 * di] = off; s_ldesc[2 * di + 1] = nb;
        return s_lab + off;
    });
    __syncthreads();
    const uint32_t nc = s_misc[0];
    if (s_lab && s_misc[3]) flush_bucket_labels<NT>(d, la, s_lab, s_ldesc, s_misc);
    bucket_tail<NT>(d, keys0, cell_ncols, nnz, s_cols, nc, reinterpret_cast<uint32_t*>(s_keys), s_run, s_ws, s_misc);
}


// ---- hash-table resolution of a cr-like bucket (one wave) ----
// A bucket holds every (umi, gene) key of the UMIs that hash to it, so the winner-take-all rule needs no
// order, only grouping: the wave inserts its keys into an LDS open-addressing table keyed by UMI whose slots
// carry up to kHtPairs (gene:20 | reads:12) counters, then walks the slots once and maps each UMI's
// most-supported gene(s) to a column.  O(1) LDS atomics per key instead of the O(log^2 n) compare-exchanges
// of a sort.  Anything the slots cannot express (a UMI seen with more than kHtPairs genes, a UMI that does
// not fit 32 bits) sends the whole bucket down the sort path - same result, just slower.
constexpr uint32_t kHtKeys = 256;            // buckets up to this many keys take the table (nearly all: the planner aims at kBucketTarget)
constexpr uint32_t kHtCap = kHtKeys + kHtKeys / 2;   // slots: load factor <= 2/3 (of distinct UMIs, usually far fewer than keys)
constexpr uint32_t kHtPairs = 3;
constexpr uint32_t kNoCol = 0xFFFFFFFFu;
static_assert(kHtKeys < (1u << 12), "per-bucket read counts fit the 12-bit counter");

__device__ __forceinline__ uint32_t ht_slot(uint32_t umi, uint32_t cap) {
    uint32_t x = umi ^ (umi >> 15);
    x *= 0x85EBCA6Bu;
    return ((x >> 16) * cap) >> 16;   // cap need not be a power of two
}

// column of one UMI from its (gene, reads) counters; same rule table as resolve_sorted
__device__ __forceinline__ uint32_t col_from_pairs(uint32_t p0, uint32_t p1, uint32_t p2, const ResolveCfg& rc) {
    const uint32_t c0 = p0 & 0xFFFu, c1 = p1 & 0xFFFu, c2 = p2 & 0xFFFu;  // unused counter: 0 reads
    const uint32_t maxc = max(c0, max(c1, c2));
    uint32_t a = c0 == maxc ? p0 >> 12 : kNoCol, b = c1 == maxc ? p1 >> 12 : kNoCol, c = c2 == maxc ? p2 >> 12 : kNoCol;
    uint32_t t;
    if (a > b) { t = a; a = b; b = t; }
    if (b > c) { t = b; b = c; c = t; }
    if (a > b) { t = a; a = b; b = t; }
    const uint32_t nb = (a != kNoCol) + (b != kNoCol) + (c != kNoCol);   // winners a <= b <= c, ascending gene id
    if (!rc.usa) return nb == 1 ? a : kNoCol;
    if (nb == 1) return is_spliced(a) ? (a >> 1) : rc.uo + (a >> 1);
    if (nb == 2) {
        if (same_gene(a, b)) return rc.ao + (a >> 1);
        if (is_spliced(a) && !is_spliced(b)) return a >> 1;
        if (!is_spliced(a) && is_spliced(b)) return b >> 1;
        return kNoCol;
    }
    const uint32_t nsp = is_spliced(a) + is_spliced(b) + is_spliced(c);
    if (nsp != 1) return kNoCol;
    if (is_spliced(a)) return same_gene(a, b) ? rc.ao + (a >> 1) : (a >> 1);
    if (is_spliced(b)) return same_gene(b, c) ? rc.ao + (b >> 1) : (b >> 1);
    return c >> 1;
}

// A UMI seen with more than kHtPairs genes parks the extra keys in a short list and is resolved by one lane
// walking that list (rare: a fraction of a percent of the UMIs).  Same rule table again, stated on order-free
// aggregates of the winner set W: |W|, its two smallest genes, its spliced members, and whether the sibling
// (g+1) of its smallest spliced gene is in W - which is what "the next winner in gene order is the same
// gene" means for ascending ids 2g, 2g+1.
constexpr uint32_t kHtOvf = 64;
constexpr uint32_t kHtMerge = 8;   // genes of one UMI the in-register merge holds (more: the bucket takes the sort path)
// (gene, reads) candidates in registers, a count of 0 = no candidate.  Plain unrolled loops over the two arrays: as a callback
// that walked them (a lambda handed a lambda) the aggregates below were captured by reference twice over, stayed in memory -
// scratch - and every candidate was a store and two loads away from the next (round 5: 108 of the kernel's scratch instructions).
template <int N>
__device__ __forceinline__ uint32_t col_from_candidates(const uint32_t (&cg)[N], const uint32_t (&cc)[N], const ResolveCfg& rc) {
    uint32_t maxc = 0;
#pragma unroll
    for (int q = 0; q < N; ++q) maxc = cc[q] > maxc ? cc[q] : maxc;
    uint32_t nb = 0, g1 = kNoCol, g2 = kNoCol, nsp = 0, first_sp = kNoCol;
#pragma unroll
    for (int q = 0; q < N; ++q) {
        const uint32_t g = cg[q];
        const bool w = cc[q] != 0 && cc[q] == maxc;
        nb += w;
        const bool lt1 = w && g < g1, lt2 = w && !lt1 && g < g2;
        g2 = lt1 ? g1 : (lt2 ? g : g2);
        g1 = lt1 ? g : g1;
        const bool sp = w && is_spliced(g);
        nsp += sp;
        first_sp = sp && g < first_sp ? g : first_sp;
    }
    if (!rc.usa) return nb == 1 ? g1 : kNoCol;
    if (nb == 1) return is_spliced(g1) ? (g1 >> 1) : rc.uo + (g1 >> 1);
    if (nb == 2) {
        if (same_gene(g1, g2)) return rc.ao + (g1 >> 1);
        if (is_spliced(g1) && !is_spliced(g2)) return g1 >> 1;
        if (!is_spliced(g1) && is_spliced(g2)) return g2 >> 1;
        return kNoCol;
    }
    if (nb > 10 || nsp != 1) return kNoCol;
    bool followed = false;
#pragma unroll
    for (int q = 0; q < N; ++q) followed = followed || (cc[q] != 0 && cc[q] == maxc && cg[q] == first_sp + 1);
    return followed ? rc.ao + (first_sp >> 1) : (first_sp >> 1);
}

// cr-like-em: a UMI whose winners are one output column is counted as that column; any other winner set becomes a
// gene-level equivalence class (quant.rs:882-924) = the winners in ascending gene order, staged in LDS.
struct EmStage {
    uint32_t* lab;     // label words
    uint32_t* ldesc;   // (offset, length) per staged molecule
    uint32_t* cnt;     // [0] words used, [1] molecules
};
__device__ __forceinline__ uint32_t em_single_column(uint32_t nb, uint32_t g1, uint32_t g2, const ResolveCfg& rc) {
    if (nb == 1) return !rc.usa ? g1 : (is_spliced(g1) ? (g1 >> 1) : rc.uo + (g1 >> 1));
    if (rc.usa && nb == 2 && same_gene(g1, g2)) return rc.ao + (g1 >> 1);
    return kNoCol;
}
__device__ __forceinline__ uint32_t* em_stage_label(const EmStage& es, uint32_t nb) {
    const uint32_t off = atomicAdd(&es.cnt[0], nb), di = atomicAdd(&es.cnt[1], 1u);
    es.ldesc[2 * di] = off; es.ldesc[2 * di + 1] = nb;
    return es.lab + off;
}
// winners of the three in-slot counters -> column, or a staged label (returns kNoCol then)
__device__ __forceinline__ uint32_t em_from_pairs(uint32_t p0, uint32_t p1, uint32_t p2, const ResolveCfg& rc, const EmStage& es) {
    const uint32_t c0 = p0 & 0xFFFu, c1 = p1 & 0xFFFu, c2 = p2 & 0xFFFu;
    const uint32_t maxc = max(c0, max(c1, c2));
    uint32_t a = c0 == maxc ? p0 >> 12 : kNoCol, b = c1 == maxc ? p1 >> 12 : kNoCol, c = c2 == maxc ? p2 >> 12 : kNoCol;
    uint32_t t;
    if (a > b) { t = a; a = b; b = t; }
    if (b > c) { t = b; b = c; c = t; }
    if (a > b) { t = a; a = b; b = t; }
    const uint32_t nb = (a != kNoCol) + (b != kNoCol) + (c != kNoCol);
    const uint32_t col = em_single_column(nb, a, b, rc);
    if (col != kNoCol) return col;
    uint32_t* dst = em_stage_label(es, nb);
    dst[0] = a; dst[1] = b;
    if (nb > 2) dst[2] = c;
    return kNoCol;
}

// One wave, n <= kHtKeys.  Slot = one 64-bit word (umi:32 | gene:20 | reads:12) holding the UMI and its first
// gene's counter - a UMI seen with one gene, the common case, costs one CAS plus one add per further read - and
// kHtPairs-1 more (gene | reads) counters.  The lane whose CAS claims a slot owns that UMI and resolves it.
// On success the bucket's columns are in s_cols[0..nc) and true is returned.
__device__ __forceinline__ bool resolve_bucket_hash(const uint64_t* __restrict__ src, uint32_t n, const ResolveCfg& rc,
                                                    unsigned long long* s_slot, uint32_t* s_pair, uint64_t* s_ovf,
                                                    uint32_t* s_flag, uint32_t* s_novf, uint32_t* s_cols, DevStatus* st,
                                                    uint32_t cell, uint32_t& nc_out, bool em, const EmStage& es) {
    constexpr uint32_t E = kHtKeys / 64;
    constexpr unsigned long long kEmpty64 = ~0ull;
    const uint32_t lane = threadIdx.x;
#ifdef AFQ_RESOLVE_TIMING
    unsigned long long tprev_ = clock64();
#endif
    uint64_t key[E];
#pragma unroll
    for (uint32_t h = 0; h < E; ++h) key[h] = h * 64 + lane < n ? AFQ_LD_RESOLVE(&src[h * 64 + lane]) : 0ull;
    uint32_t cap = (n + (n >> 1) + 63) & ~63u;   // multiples of 64 slots: 1.5 n rounded up
    cap = cap < 128 ? 128 : cap;
    {
        uint4* u4 = reinterpret_cast<uint4*>(s_slot);
        uint4* p4 = reinterpret_cast<uint4*>(s_pair);
        for (uint32_t i = lane; i < cap / 2; i += 64) u4[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
        for (uint32_t i = lane; i < cap * (kHtPairs - 1) / 4; i += 64) p4[i] = make_uint4(0, 0, 0, 0);
        if (lane < kHtCap / 32) s_flag[lane] = 0;
        if (lane == 0) *s_novf = 0;
    }
    __syncthreads();
#ifdef AFQ_RESOLVE_TIMING
    if (key[0] == 1234567ull) return false;  // wait for the loads so that their latency lands in phase 0
#endif
    RT_MARK(0);
    bool bad = false;
    uint32_t own_slot[E];
#pragma unroll
    for (uint32_t h = 0; h < E; ++h) {
        own_slot[h] = kNoCol;
        if (h * 64 >= n) break;
        if (h * 64 + lane < n) {
            const uint64_t u64 = key[h] >> kGeneBits;
            const uint32_t gene = (uint32_t)key[h] & kGeneMask;
            if (u64 >= 0xFFFFFFFFull) bad = true;  // does not fit the slot word (or would read as "empty")
            else {
                const uint32_t umi = (uint32_t)u64;
                const unsigned long long mine = ((unsigned long long)umi << 32) | (gene << 12) | 1u;
                uint32_t slot = ht_slot(umi, cap);
                // the probe loop only finds the UMI's slot (one exit: the slot was empty - now claimed - or holds this UMI;
                // an occupied slot's UMI word is never 0xFFFFFFFF); what the hit means is sorted out after it
                unsigned long long old;
                for (;;) {
                    old = atomicCAS(&s_slot[slot], kEmpty64, mine);
                    const uint32_t ou = (uint32_t)(old >> 32);
                    if (ou == umi || ou == 0xFFFFFFFFu) break;
                    slot = slot + 1 == cap ? 0u : slot + 1;
                }
                bool done = true;
                if ((uint32_t)(old >> 32) == 0xFFFFFFFFu) own_slot[h] = slot;
                else if ((((uint32_t)old) >> 12) == gene) atomicAdd(&s_slot[slot], 1ull);
                else done = false;
                if (!done) {
                    uint32_t* pr = s_pair + slot * (kHtPairs - 1);
#pragma unroll
                    for (uint32_t q = 0; q < kHtPairs - 1; ++q) {
                        if (!done) {
                            const uint32_t old = atomicCAS(&pr[q], 0u, (gene << 12) | 1u);
                            if (old == 0u) done = true;
                            else if ((old >> 12) == gene) { atomicAdd(&pr[q], 1u); done = true; }
                        }
                    }
                }
                if (!done) {  // the UMI's counters are taken by other genes (and this gene can never get one)
                    const uint32_t k = atomicAdd(s_novf, 1u);
                    if (k < kHtOvf) { s_ovf[k] = key[h]; atomicOr(&s_flag[slot >> 5], 1u << (slot & 31)); }
                    else bad = true;
                }
            }
        }
    }
    if (__any(bad)) return false;
    __syncthreads();
    RT_MARK(1);
    const uint32_t novf = *s_novf;
    uint32_t nc = 0;
#pragma unroll
    for (uint32_t h = 0; h < E; ++h) {
        if (h * 64 >= n) break;
        uint32_t col = kNoCol;
        const uint32_t slot = own_slot[h];
        if (slot != kNoCol) {
            const uint32_t p0 = (uint32_t)s_slot[slot], umi = (uint32_t)(key[h] >> kGeneBits);
            const uint32_t p1 = s_pair[slot * (kHtPairs - 1)], p2 = s_pair[slot * (kHtPairs - 1) + 1];
            if (!novf || !((s_flag[slot >> 5] >> (slot & 31)) & 1u)) col = em ? em_from_pairs(p0, p1, p2, rc, es) : col_from_pairs(p0, p1, p2, rc);
            else {
                // the UMI's three counters plus its parked keys, merged into at most kHtMerge (gene, reads) entries held
                // in registers (predicated writes, no dynamic indexing), then one pass for the rule's aggregates
                uint32_t cg[kHtMerge], cc[kHtMerge];
#pragma unroll
                for (uint32_t q = 0; q < kHtMerge; ++q) { cg[q] = kNoCol; cc[q] = 0; }
                cg[0] = p0 >> 12; cc[0] = p0 & 0xFFFu; cg[1] = p1 >> 12; cc[1] = p1 & 0xFFFu; cg[2] = p2 >> 12; cc[2] = p2 & 0xFFFu;
                uint32_t k = 3;
                for (uint32_t i = 0; i < novf; ++i) {
                    const uint64_t ki = s_ovf[i];
                    if ((uint32_t)(ki >> kGeneBits) != umi) continue;
                    const uint32_t g = (uint32_t)ki & kGeneMask;
                    bool found = false;
#pragma unroll
                    for (uint32_t q = 3; q < kHtMerge; ++q) if (q < k && cg[q] == g) { ++cc[q]; found = true; }
                    if (!found) {
                        if (k == kHtMerge) { bad = true; break; }
#pragma unroll
                        for (uint32_t q = 3; q < kHtMerge; ++q) if (q == k) { cg[q] = g; cc[q] = 1; }
                        ++k;
                    }
                }
                if (!em) {
                    col = col_from_candidates(cg, cc, rc);
                } else if (!bad) {
                    uint32_t maxc = 0, nb = 0, g1 = kNoCol, g2 = kNoCol;
#pragma unroll
                    for (uint32_t q = 0; q < kHtMerge; ++q) maxc = cc[q] > maxc ? cc[q] : maxc;
#pragma unroll
                    for (uint32_t q = 0; q < kHtMerge; ++q)
                        if (cc[q] == maxc) { ++nb; if (cg[q] < g1) { g2 = g1; g1 = cg[q]; } else if (cg[q] < g2) g2 = cg[q]; }
                    col = em_single_column(nb, g1, g2, rc);
                    if (col == kNoCol) {  // the winners in ascending gene order: repeated minimum over <= kHtMerge entries
                        uint32_t* dst = em_stage_label(es, nb);
                        uint32_t last = 0;
                        for (uint32_t w = 0; w < nb; ++w) {
                            uint32_t best = kNoCol;
#pragma unroll
                            for (uint32_t q = 0; q < kHtMerge; ++q)
                                if (cc[q] == maxc && cg[q] < best && (w == 0 || cg[q] > last)) best = cg[q];
                            dst[w] = best;
                            last = best;
                        }
                    }
                }
            }
            if (col != kNoCol && col >= rc.num_rows) { set_err(st, kErrSlotRange, cell); col = kNoCol; }
        }
        const uint64_t m = __ballot(col != kNoCol);
        if (col != kNoCol) s_cols[nc + (uint32_t)__popcll(m & ((1ull << lane) - 1))] = col;
        nc += (uint32_t)__popcll(m);
    }
    if (__any(bad)) return false;  // a UMI with more genes than the merge holds: nothing global was written yet
    __syncthreads();
    RT_MARK(2);
    nc_out = nc;
    return true;
}

// ---- hash-table resolution of a cr-like bucket whose UMIs carry ANY number of genes (one wave) ----
// Reads off gene families map to five, ten, thirty transcripts of as many genes; a molecule's reads then put that many keys
// under its UMI and the three counters of resolve_bucket_hash's slot overflow for most UMIs (the bucket is then sorted after all:
// a batch that averages two or more alignments per record used to take the sort path outright, 34 G keys/s).  Here nothing
// overflows.  Two tables: T1 keyed by (UMI, gene) counts the reads of each pair; T2 keyed by UMI collects, with atomics, the
// order-free aggregates the rule table needs (col_from_candidates): the largest count, how many genes reach it, how many
// of those are spliced, the smallest winner, the smallest spliced winner, and whether an unspliced winner's spliced sibling
// is a winner too.  The lane that claimed a UMI's T2 slot emits its column.  cr-like only (cr-like-em wants the winners as a list).
constexpr uint32_t kH2Cap = kHtKeys + kHtKeys / 2;
static_assert(kHtKeys < (1u << 10), "resolve_bucket_hash2 packs a UMI's winner count and its spliced-winner count into 10-bit fields of agg");
static_assert(kHtKeys < (1u << 12), "... and the reads of a (UMI, gene) pair into 12 bits of the T1 / T2 words");
constexpr uint32_t kH2Words = 2 * kH2Cap /* T1 */ + 2 * kH2Cap /* T2 key | max */ + 3 * kH2Cap /* agg, gmin, smin */ + kHtKeys /* columns */;
__device__ __forceinline__ bool resolve_bucket_hash2(const uint64_t* __restrict__ src, uint32_t n, const ResolveCfg& rc, uint32_t* s_raw,
                                                     DevStatus* st, uint32_t cell, uint32_t& nc_out, uint32_t*& s_cols_out) {
    constexpr uint32_t E = kHtKeys / 64;
    constexpr unsigned long long kEmpty64 = ~0ull;
    const uint32_t lane = threadIdx.x;
    unsigned long long* t1 = reinterpret_cast<unsigned long long*>(s_raw);
    unsigned long long* t2 = t1 + kH2Cap;
    uint32_t* agg = s_raw + 4 * kH2Cap;    // winners: count | spliced ones << 10 | "a winner's spliced sibling wins too" << 31
    uint32_t* gmin = agg + kH2Cap;
    uint32_t* smin = gmin + kH2Cap;
    uint32_t* s_cols = smin + kH2Cap;
    uint64_t key[E];
#pragma unroll
    for (uint32_t h = 0; h < E; ++h) key[h] = h * 64 + lane < n ? src[h * 64 + lane] : 0ull;
    uint32_t cap = (n + (n >> 1) + 63) & ~63u;
    cap = cap < 128 ? 128 : cap;
    for (uint32_t i = lane; i < cap; i += 64) { t1[i] = kEmpty64; t2[i] = kEmpty64; agg[i] = 0; gmin[i] = 0xFFFFFFFFu; smin[i] = 0xFFFFFFFFu; }
    __syncthreads();
    bool bad = false;
    uint32_t s1[E], s2[E];   // my pair's T1 slot when I claimed it; its UMI's T2 slot; bit 31 of s2: I claimed that one too
#pragma unroll
    for (uint32_t h = 0; h < E; ++h) { s1[h] = kNoCol; s2[h] = kNoCol; }
#pragma unroll
    for (uint32_t h = 0; h < E; ++h) {
        if (h * 64 >= n) break;
        if (h * 64 + lane < n) {
            const uint64_t u64 = key[h] >> kGeneBits;
            const uint32_t gene = (uint32_t)key[h] & kGeneMask;
            if (u64 >= 0xFFFFFFFFull) bad = true;
            else {
                const uint32_t umi = (uint32_t)u64;
                const unsigned long long mine = ((unsigned long long)umi << 32) | (gene << 12) | 1u;
                uint32_t slot = ht_slot(umi ^ (gene * 0x9E3779B1u), cap);
                unsigned long long old;
                for (;;) {   // one exit: the slot was empty (now mine) or holds this (UMI, gene)
                    old = atomicCAS(&t1[slot], kEmpty64, mine);
                    if (old == kEmpty64 || (old >> 12) == (mine >> 12)) break;
                    slot = slot + 1 == cap ? 0u : slot + 1;
                }
                if (old != kEmpty64) atomicAdd(&t1[slot], 1ull);
                else {
                    s1[h] = slot;
                    const unsigned long long um = (unsigned long long)umi << 32;
                    uint32_t q = ht_slot(umi, cap);
                    for (;;) {
                        old = atomicCAS(&t2[q], kEmpty64, um);
                        if (old == kEmpty64 || (uint32_t)(old >> 32) == umi) break;
                        q = q + 1 == cap ? 0u : q + 1;
                    }
                    s2[h] = q | (old == kEmpty64 ? 0x80000000u : 0u);
                }
            }
        }
    }
    if (__any(bad)) return false;   // a UMI that does not fit the slot word: the sort path takes the bucket (nothing global was written)
    __syncthreads();
#pragma unroll
    for (uint32_t h = 0; h < E; ++h)   // the largest read count of each UMI
        if (s1[h] != kNoCol) atomicMax(&t2[s2[h] & 0x7FFFFFFFu], (t2[s2[h] & 0x7FFFFFFFu] & 0xFFFFFFFF00000000ull) | ((uint32_t)t1[s1[h]] & 0xFFFu));
    __syncthreads();
#pragma unroll
    for (uint32_t h = 0; h < E; ++h) {   // the winners' aggregates
        if (s1[h] == kNoCol) continue;
        const uint32_t q = s2[h] & 0x7FFFFFFFu;
        const uint32_t cnt = (uint32_t)t1[s1[h]] & 0xFFFu, mx = (uint32_t)t2[q] & 0xFFFu;
        if (cnt != mx) continue;
        const uint32_t gene = (uint32_t)key[h] & kGeneMask, umi = (uint32_t)(key[h] >> kGeneBits);
        const bool sp = rc.usa && is_spliced(gene);
        atomicAdd(&agg[q], 1u | (sp ? 1u << 10 : 0u));
        atomicMin(&gmin[q], gene);
        if (sp) atomicMin(&smin[q], gene);
        if (rc.usa && !sp) {   // does this unspliced winner's spliced sibling win too?
            const unsigned long long want = ((unsigned long long)umi << 32) | ((gene - 1) << 12);
            for (uint32_t slot = ht_slot(umi ^ ((gene - 1) * 0x9E3779B1u), cap);; slot = slot + 1 == cap ? 0u : slot + 1) {
                const unsigned long long e = t1[slot];
                if (e == kEmpty64) break;
                if ((e >> 12) == (want >> 12)) { if (((uint32_t)e & 0xFFFu) == mx) atomicOr(&agg[q], 1u << 31); break; }
            }
        }
    }
    __syncthreads();
    uint32_t nc = 0;
#pragma unroll
    for (uint32_t h = 0; h < E; ++h) {
        if (h * 64 >= n) break;
        uint32_t col = kNoCol;
        if (s2[h] != kNoCol && (s2[h] >> 31)) {   // (same rule table as col_from_candidates, on the aggregates)
            const uint32_t q = s2[h] & 0x7FFFFFFFu, a = agg[q];
            const uint32_t nb = a & 0x3FFu, nsp = (a >> 10) & 0x3FFu, g1 = gmin[q], sg = smin[q];
            const bool followed = a >> 31;
            if (!rc.usa) col = nb == 1 ? g1 : kNoCol;
            else if (nb == 1) col = is_spliced(g1) ? (g1 >> 1) : rc.uo + (g1 >> 1);
            else if (nb == 2) col = followed ? rc.ao + (sg >> 1) : (nsp == 1 ? sg >> 1 : kNoCol);
            else if (nb <= 10 && nsp == 1) col = followed ? rc.ao + (sg >> 1) : (sg >> 1);
            if (col != kNoCol && col >= rc.num_rows) { set_err(st, kErrSlotRange, cell); col = kNoCol; }
        }
        const uint64_t m = __ballot(col != kNoCol);
        if (col != kNoCol) s_cols[nc + (uint32_t)__popcll(m & ((1ull << lane) - 1))] = col;
        nc += (uint32_t)__popcll(m);
    }
    __syncthreads();
    nc_out = nc;
    s_cols_out = s_cols;
    return true;
}

// One wave per bucket (up to kBucketCap keys).  Small workgroups keep many buckets
// in flight per CU, which is what hides the load -> group -> reserve -> store latency
// chain; blocks that run together are spread over different cells (column-major walk)
// so their reservations do not pile onto one counter.
// Single-bucket cells are finished here; buckets of multi-bucket cells append their
// resolved columns to the cell's column list, counted later by k_cell_hist.
constexpr int kResolveNT = 64;
constexpr uint32_t kResolveCols = 8192;
static_assert(kBucketCap <= kResolveNT * 8, "bucket cap <= 8 keys per thread");
template <bool EM, bool MULTI>
__global__ __launch_bounds__(kResolveNT) void k_resolve(const BucketDesc* __restrict__ desc, uint32_t n_buckets,
                                                       uint64_t* __restrict__ keys0,
                                                       const uint64_t* __restrict__ keys1,
                                                       uint32_t* __restrict__ cell_ncols, uint32_t* __restrict__ nnz,
                                                       OverflowEnt* __restrict__ ovf_list, DevStatus* st,
                                                       ResolveCfg rc, LabArea la) {
    // one LDS block carved two ways: the hash table (slot UMIs | counters), or the sort path's arrays
    constexpr uint32_t kSortWords = 2 * kBucketCap + kBucketCap / 2 + kBucketCap + (EM ? 2 * kBucketCap : 0);
    constexpr uint32_t kHashWords = kHtCap * (1 + kHtPairs) + 2 * kHtOvf + kHtCap / 32 + 2 + kHtKeys + (EM ? 2 * kHtKeys : 0);
    constexpr uint32_t kWords01 = kSortWords > kHashWords ? kSortWords : kHashWords;
    constexpr uint32_t kWords = MULTI && kH2Words > kWords01 ? kH2Words : kWords01;
    __shared__ __attribute__((aligned(16))) uint32_t s_raw[kWords];
    __shared__ uint32_t s_ws[kResolveNT / 64];
    __shared__ uint32_t s_misc[6];
    uint64_t* s_keys = reinterpret_cast<uint64_t*>(s_raw);
    uint16_t* s_run = reinterpret_cast<uint16_t*>(s_raw + 2 * kBucketCap);
    uint32_t* s_cols = s_raw + 2 * kBucketCap + kBucketCap / 2;
    uint32_t* s_lab = EM ? s_cols + kBucketCap : nullptr;
    uint32_t* s_ldesc = EM ? s_lab + kBucketCap : nullptr;
    const uint32_t n_cols = min(n_buckets, kResolveCols);
    const uint32_t n_rows = (n_buckets + n_cols - 1) / n_cols;
    const uint32_t b = (blockIdx.x % n_cols) * n_rows + blockIdx.x / n_cols;
    if (b >= n_buckets) return;
    const BucketDesc d = desc[b];
    if (d.n == 0) {
        if ((d.mode_single >> 8) && threadIdx.x == 0) nnz[d.cell] = 0;
        return;
    }
    if (d.n > kBucketCap) {  // only multi-bucket cells can get here (planner keeps single buckets <= target)
        if (threadIdx.x == 0) {
            const uint32_t k = atomicAdd(&st->n_overflow, 1u);
            ovf_list[k].bucket = b;
            ovf_list[k].n = d.n;
        }
        return;
    }
    const uint32_t bmode = d.mode_single & 0xFFu;
    if constexpr (MULTI) {   // a batch of many-gene reads: cr-like buckets through the two-table path; everything else is sorted
        if (bmode == kModeCrLike && d.n <= kHtKeys && !rc.pa) {
            if (threadIdx.x < 6) s_misc[threadIdx.x] = 0;
            uint32_t nc = 0;
            uint32_t* h2cols = nullptr;
            if (resolve_bucket_hash2(((d.mode_single >> 8) ? keys0 : keys1) + d.src_off, d.n, rc, s_raw, st, d.cell, nc, h2cols)) {
                // the tables are dead: their space is the tail's scratch (sorted columns, run starts); the columns sit behind them
                bucket_tail<kResolveNT>(d, keys0, cell_ncols, nnz, h2cols, nc, s_raw, reinterpret_cast<uint16_t*>(s_raw + kHtKeys), s_ws, s_misc);
                return;
            }
            __syncthreads();
        }
    }
    if ((bmode == kModeCrLike || (EM && bmode == kModeCrLikeEm && la.lab)) && d.n <= kHtKeys && !rc.pa && !rc.sort_only) {
        const bool single = (d.mode_single >> 8) != 0;
        unsigned long long* s_slot = reinterpret_cast<unsigned long long*>(s_raw);      // 2 words per slot
        uint32_t* s_pair = s_raw + 2 * kHtCap;                                            // kHtPairs-1 words per slot
        uint64_t* s_ovf = reinterpret_cast<uint64_t*>(s_raw + kHtCap * (1 + kHtPairs));
        uint32_t* s_flag = s_raw + kHtCap * (1 + kHtPairs) + 2 * kHtOvf;
        uint32_t* s_hcols = s_flag + kHtCap / 32 + 2;
        uint32_t* s_hlab = s_hcols + kHtKeys;          // EM only: staged label words / descriptors of this bucket
        if (threadIdx.x < 6) s_misc[threadIdx.x] = 0;
        const EmStage es{s_hlab, s_hlab + kHtKeys, &s_misc[2]};
        uint32_t nc = 0;
        if (resolve_bucket_hash((single ? keys0 : keys1) + d.src_off, d.n, rc, s_slot, s_pair, s_ovf, s_flag, s_flag + kHtCap / 32,
                                s_hcols, st, d.cell, nc, EM && bmode == kModeCrLikeEm, es)) {
            if (EM && s_misc[3]) flush_bucket_labels<kResolveNT>(d, la, es.lab, es.ldesc, s_misc);
            // the table is dead: its space is the tail's scratch (sorted columns, run starts)
            bucket_tail<kResolveNT>(d, keys0, cell_ncols, nnz, s_hcols, nc, s_raw, reinterpret_cast<uint16_t*>(s_raw + kHtKeys),
                                    s_ws, s_misc);
            return;
        }
        __syncthreads();
    }
    resolve_bucket_lds<kResolveNT>(d, keys0, keys1, cell_ncols, nnz, st, rc, la, s_keys, s_run, s_cols, s_lab, s_ldesc, s_ws,
                                   s_misc);
}

// Buckets beyond LDS reach (one UMI carried by thousands of reads, adversarial
// input): same algorithm with the bucket sorted in place in keys1 (normalised
// bitonic network, any n) and the run table in the upper half of the cell's keys0
// slots.  Workgroup-wide call (k_resolve_mid's loop over the overflow list).
constexpr int kBigNT = 1024;   // (= kMidNT: one kernel takes both kinds of overflow bucket)
__device__ __forceinline__ void resolve_bucket_global(const BucketDesc& d, const CellMeta* __restrict__ meta, uint64_t* __restrict__ keys0, uint64_t* __restrict__ keys1,
                                                      uint32_t* __restrict__ cell_ncols, DevStatus* st, const ResolveCfg& rc, const LabArea& la, uint32_t* s_ws) {
    const uint32_t n = d.n;
    const uint32_t n_ref = meta[d.cell].n_ref;
    const uint32_t beg = (uint32_t)(d.src_off - d.out_off);
    uint64_t* keys = keys1 + d.src_off;
    // keys0 region of the cell = 2*n_ref words: words [0, n_ref) hold the column list (<= nkeys <= n_ref
    // entries); the run table of this bucket (<= n entries) lives at words [n_ref + beg, n_ref + beg + n),
    // disjoint between buckets because their [beg, beg+n) key ranges are.
    uint32_t* run = reinterpret_cast<uint32_t*>(keys0 + d.out_off) + n_ref + beg;
    __syncthreads();
    bitonic_sort<kBigNT>(keys, n);
    uint32_t* cols = reinterpret_cast<uint32_t*>(keys0 + d.out_off);
    ResolveCfg rcb = rc;
    rcb.mode = d.mode_single & 0xFFu;
    resolve_sorted<kBigNT>(keys, n, run, s_ws, rcb, [&](uint32_t col) {
        if (col >= rc.num_rows) { set_err(st, kErrSlotRange, d.cell); return; }
        cols[atomicAdd(&cell_ncols[d.cell], 1u)] = col;
    }, [&](uint32_t nb) -> uint32_t* {
        return la.lab ? lab_alloc_global(la, d.cell, d.out_off, d.n_ref, nb) : nullptr;
    });
    __syncthreads();
}

// Buckets over the 2-wave cap but within LDS reach (<= kMidCap keys): persistent
// 1024-thread workgroups loop over the overflow list.
constexpr int kMidNT = 1024;
static_assert(kBigNT == kMidNT, "one kernel takes both kinds of overflow bucket");
constexpr uint32_t kMidCap = kMidNT * 8;
// (Round 6: ONE launch for both kinds of overflow bucket - up to kMidCap keys: sorted in LDS; beyond: in place in keys1, below.  The
//  list is nearly always empty, and a second 5 us launch with its boundary was paid per range for it.)
__global__ __launch_bounds__(kMidNT) void k_resolve_mid(const BucketDesc* __restrict__ desc, const CellMeta* __restrict__ meta,
                                                       uint64_t* __restrict__ keys0,
                                                       uint64_t* __restrict__ keys1,
                                                       uint32_t* __restrict__ cell_ncols, uint32_t* __restrict__ nnz,
                                                       const OverflowEnt* __restrict__ ovf_list, DevStatus* st,
                                                       ResolveCfg rc, LabArea la) {
    __shared__ uint64_t s_keys[kMidCap];
    __shared__ uint16_t s_run[kMidCap];
    __shared__ uint32_t s_cols[kMidCap];
    __shared__ uint32_t s_ws[kMidNT / 64];
    __shared__ uint32_t s_misc[6];
    // EM modes: this rare path writes labels straight to the cell's label area (global atomics per molecule)
    uint32_t* s_lab = nullptr;
    uint32_t* s_ldesc = nullptr;
    const uint32_t novf = st->n_overflow;
    for (uint32_t e = blockIdx.x; e < novf; e += gridDim.x) {
        const BucketDesc d = desc[ovf_list[e].bucket];
        if (ovf_list[e].n > kMidCap) { resolve_bucket_global(d, meta, keys0, keys1, cell_ncols, st, rc, la, s_ws); continue; }   // (uniform)
        __syncthreads();
        resolve_bucket_lds<kMidNT>(d, keys0, keys1, cell_ncols, nnz, st, rc, la, s_keys, s_run, s_cols, s_lab, s_ldesc, s_ws, s_misc);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// Per-cell count of the resolved columns of a multi-bucket cell: one workgroup
// per cell histograms the column list into LDS (32768 bins per pass = 128 KiB of
// the CU's 160 KiB; ceil(num_rows/32768) passes over the L2-resident list), then
// compacts the non-zero bins in column order into (column,count) pairs written
// over the cell's dead keys1 slots.  No global atomics, no dense scratch rows.
constexpr int kHistNT = 1024;
constexpr uint32_t kHistBins = 32768;
__global__ __launch_bounds__(kHistNT) void k_cell_hist(const uint32_t* __restrict__ multi_cells,
                                                      const CellMeta* __restrict__ meta,
                                                      const uint64_t* __restrict__ keys0, uint64_t* __restrict__ keys1,
                                                      const uint32_t* __restrict__ cell_ncols,
                                                      uint32_t* __restrict__ nnz, ResolveCfg rc, uint32_t hist_words) {
    // hist_words of LDS (dynamic): as many as one pass over a gene-level matrix needs, at most kHistBins - a 36 601-column
    // matrix takes 73 KiB, and two workgroups share a CU
    extern __shared__ uint32_t s_hist[];
    __shared__ uint32_t s_ws[kHistNT / 64];
    const uint32_t cell = multi_cells[blockIdx.x];
    const CellMeta m = meta[cell];
    const uint32_t* cols = reinterpret_cast<const uint32_t*>(keys0 + m.key_off);
    uint2* out = reinterpret_cast<uint2*>(keys1 + m.key_off);
    const uint32_t nc = cell_ncols[cell];
    uint32_t carry = 0;
    if (nc < 65536u) {
        // no column can be counted 65536 times: two 16-bit bins per LDS word, 65536 bins per pass - one pass
        // for a gene-level matrix of up to 65536 columns (half the clearing and scanning of the 32-bit version)
        const uint32_t kBins16 = 2 * hist_words;
        for (uint32_t lo = 0; lo < rc.num_rows; lo += kBins16) {
            const uint32_t nbins = min(kBins16, rc.num_rows - lo), nwords = (nbins + 1) / 2;
            for (uint32_t i = threadIdx.x; i < nwords; i += kHistNT) s_hist[i] = 0;
            __syncthreads();
            for (uint32_t i0 = threadIdx.x; i0 < nc; i0 += 4 * kHistNT) {   // four columns per thread and trip, their loads in flight together
                uint32_t cv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) cv[e] = i0 + e * kHistNT < nc ? cols[i0 + e * kHistNT] - lo : 0xFFFFFFFFu;
#pragma unroll
                for (int e = 0; e < 4; ++e) if (cv[e] < nbins) atomicAdd(&s_hist[cv[e] >> 1], 1u << (16 * (cv[e] & 1u)));
            }
            __syncthreads();
            // compaction in column order with ONE workgroup scan per pass: thread t owns the 32 words (64 bins) from word 32 t,
            // notes which of its bins are non-zero in a 64-bit mask (its words read in an order rotated by the lane, so that
            // the lanes of a wave are not all on one LDS bank), and after the scan of the counts walks the set bits.  (A scan per
            // 2048 words was 16 scans and 48 barriers per pass - most of this kernel's time for a USA matrix of 110 k columns.)
            const uint32_t w0 = 32 * threadIdx.x;
            unsigned long long nzm = 0;
            if (w0 < nwords) {
#pragma unroll 8
                for (uint32_t i = 0; i < 32; ++i) {
                    const uint32_t w = (i + threadIdx.x) & 31u;
                    const uint32_t v = w0 + w < nwords ? s_hist[w0 + w] : 0u;
                    nzm |= (unsigned long long)(((v & 0xFFFFu) != 0) | (((v >> 16) != 0) << 1)) << (2 * w);
                }
            }
            uint32_t tot;
            uint32_t o = carry + block_excl_scan<kHistNT>((uint32_t)__popcll(nzm), s_ws, tot);
            while (nzm) {
                const uint32_t k = (uint32_t)__builtin_ctzll(nzm);
                nzm &= nzm - 1;
                const uint32_t v = s_hist[w0 + (k >> 1)];
                out[o++] = make_uint2(lo + 2 * w0 + k, (k & 1u) ? v >> 16 : v & 0xFFFFu);
            }
            carry += tot;
            __syncthreads();
        }
        if (threadIdx.x == 0) nnz[cell] = carry;
        return;
    }
    for (uint32_t lo = 0; lo < rc.num_rows; lo += hist_words) {
        const uint32_t nbins = min(hist_words, rc.num_rows - lo);
        for (uint32_t i = threadIdx.x; i < nbins; i += kHistNT) s_hist[i] = 0;
        __syncthreads();
        for (uint32_t i0 = threadIdx.x; i0 < nc; i0 += 4 * kHistNT) {
            uint32_t cv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) cv[e] = i0 + e * kHistNT < nc ? cols[i0 + e * kHistNT] - lo : 0xFFFFFFFFu;
#pragma unroll
            for (int e = 0; e < 4; ++e) if (cv[e] < nbins) atomicAdd(&s_hist[cv[e]], 1u);
        }
        __syncthreads();
        for (uint32_t base = 0; base < nbins; base += kHistNT * 4) {
            const uint32_t q = base + threadIdx.x * 4;
            uint32_t v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (q + e < nbins) ? s_hist[q + e] : 0u;
            const uint32_t c = (v[0] != 0) + (v[1] != 0) + (v[2] != 0) + (v[3] != 0);
            uint32_t tot;
            uint32_t o = carry + block_excl_scan<kHistNT>(c, s_ws, tot);
#pragma unroll
            for (int e = 0; e < 4; ++e) if (v[e]) out[o++] = make_uint2(lo + q + e, v[e]);
            carry += tot;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) nnz[cell] = carry;
}

// ---------------------------------------------------------------------------
// staging pairs -> final CSR (wave per cell)
// (cap: how many entries gene / val hold.  The compaction that is enqueued right behind a range's kernels - row offsets from
//  k_row_ptr, no trip to the host in between - runs against the buffers as they were when the range was enqueued; rows that
//  do not fit are left where they are and the host, which sees the same total, compacts them after growing the buffers.)
__global__ __launch_bounds__(256) void k_compact(const CellMeta* __restrict__ meta, uint32_t n_cells,
                                                const uint64_t* __restrict__ keys0, const uint64_t* __restrict__ keys1,
                                                const uint32_t* __restrict__ nnz,
                                                const uint64_t* __restrict__ cell_ptr, uint32_t* __restrict__ gene,
                                                float* __restrict__ val, uint64_t cap) {
    const uint32_t cell = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (cell >= n_cells) return;
    if (cell_ptr[n_cells] > cap) return;
    const CellMeta m = meta[cell];
    const uint2* src = reinterpret_cast<const uint2*>(((m.lg_nb || mode_is_pug(m.mode)) ? keys1 : keys0) + m.key_off);
    const uint32_t n = nnz[cell];
    const uint64_t o = cell_ptr[cell];
    for (uint32_t i = lane_id(); i < n; i += 64) {
        const uint2 p = src[i];
        gene[o + i] = p.x;
        val[o + i] = (float)p.y;
    }
}

// ---------------------------------------------------------------------------
// ATAC per-cell fragment de-duplication (src/atac/deduplicate.rs:199-237): sort a
// cell's fragments by (chr, start, frag_len) — HitInfo's Ord, src/atac/sort.rs:47-58;
// the barcode is constant within a cell — and run-length count them.  One
// 1024-thread workgroup per cell sorts in place in a global scratch copy (the
// normalised bitonic network takes any n); cells are independent.
struct Frag {
    uint64_t hi;  // chr << 32 | start
    uint64_t lo;  // frag_len
};
__device__ __forceinline__ bool operator>(const Frag& a, const Frag& b) { return a.hi > b.hi || (a.hi == b.hi && a.lo > b.lo); }
__device__ __forceinline__ bool operator!=(const Frag& a, const Frag& b) { return a.hi != b.hi || a.lo != b.lo; }

constexpr int kAtacNT = 1024;
__global__ __launch_bounds__(kAtacNT) void k_atac_dedup(const uint32_t* __restrict__ ref, const uint32_t* __restrict__ start,
                                                       const uint16_t* __restrict__ flen,
                                                       const uint64_t* __restrict__ cell_ptr, Frag* __restrict__ scratch,
                                                       uint32_t* __restrict__ o_ref, uint32_t* __restrict__ o_start,
                                                       uint16_t* __restrict__ o_flen, uint16_t* __restrict__ o_cnt,
                                                       uint32_t* __restrict__ o_n, const uint32_t* __restrict__ cell_cnt) {
    __shared__ uint32_t s_ws[kAtacNT / 64];
    const uint32_t cell = blockIdx.x;
    const uint64_t b0 = cell_ptr[cell];
    const uint32_t n = cell_cnt ? cell_cnt[cell] : (uint32_t)(cell_ptr[cell + 1] - b0);   // cell_cnt: fragments kept by the RAD decode (cell_ptr = capacity)
    Frag* f = scratch + b0;
    for (uint32_t i = threadIdx.x; i < n; i += kAtacNT) {
        Frag x;
        x.hi = ((uint64_t)ref[b0 + i] << 32) | start[b0 + i];
        x.lo = flen[b0 + i];
        f[i] = x;
    }
    __syncthreads();
    bitonic_sort<kAtacNT>(f, n);
    // run heads -> output slot; run length = distance to the next head (found by scanning forward)
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n; base += kAtacNT) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t h = (i < n) && (i == 0 || f[i] != f[i - 1]);
        uint32_t tot;
        const uint32_t ex = block_excl_scan<kAtacNT>(h, s_ws, tot);
        if (h) {
            uint32_t e = i + 1;
            while (e < n && !(f[e] != f[i])) ++e;
            const uint64_t o = b0 + carry + ex;
            o_ref[o] = (uint32_t)(f[i].hi >> 32);
            o_start[o] = (uint32_t)f[i].hi;
            o_flen[o] = (uint16_t)f[i].lo;
            o_cnt[o] = (uint16_t)(e - i);  // `count as u16`, deduplicate.rs:220
        }
        carry += tot;
    }
    if (threadIdx.x == 0) o_n[cell] = carry;
}

// Fast variant for reference ids below 65536 (every chromosome-level genome): one 64-bit key
// ref:16 | start:32 | frag_len:16 per fragment - its integer order is the (ref, start, frag_len) order - sorted
// with the LDS-tiled bitonic network (16384 keys = 128 KiB per tile) instead of 16-byte records out of global
// memory.  A cell that holds a larger reference id raises *flag and the host reruns the batch with k_atac_dedup.
__global__ __launch_bounds__(kAtacNT) void k_atac_dedup64(const uint32_t* __restrict__ ref, const uint32_t* __restrict__ start,
                                                         const uint16_t* __restrict__ flen,
                                                         const uint64_t* __restrict__ cell_ptr, uint64_t* __restrict__ scratch,
                                                         uint32_t* __restrict__ o_ref, uint32_t* __restrict__ o_start,
                                                         uint16_t* __restrict__ o_flen, uint16_t* __restrict__ o_cnt,
                                                         uint32_t* __restrict__ o_n, uint32_t* __restrict__ flag, const uint32_t* __restrict__ cell_cnt) {
    __shared__ uint32_t s_ws[kAtacNT / 64];
    __shared__ __attribute__((aligned(16))) uint64_t s_tile[16384];
    const uint32_t cell = blockIdx.x;
    const uint64_t b0 = cell_ptr[cell];
    const uint32_t n = cell_cnt ? cell_cnt[cell] : (uint32_t)(cell_ptr[cell + 1] - b0);
    uint64_t* f = scratch + b0;
    bool wide = false;
    for (uint32_t i = threadIdx.x; i < n; i += kAtacNT) {
        const uint32_t r = ref[b0 + i];
        wide = wide || r > 0xFFFFu;
        f[i] = ((uint64_t)r << 48) | ((uint64_t)start[b0 + i] << 16) | flen[b0 + i];
    }
    if (wide) *flag = 1;
    __syncthreads();
    tiled_bitonic_sort_by<kAtacNT, 16384>(f, n, [](uint64_t a, uint64_t b) { return a > b; }, s_tile);
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n; base += kAtacNT) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t h = (i < n) && (i == 0 || f[i] != f[i - 1]);
        uint32_t tot;
        const uint32_t ex = block_excl_scan<kAtacNT>(h, s_ws, tot);
        if (h) {
            const uint64_t k = f[i];
            uint32_t e = i + 1;
            while (e < n && f[e] == k) ++e;
            const uint64_t o = b0 + carry + ex;
            o_ref[o] = (uint32_t)(k >> 48);
            o_start[o] = (uint32_t)(k >> 16);
            o_flen[o] = (uint16_t)k;
            o_cnt[o] = (uint16_t)(e - i);  // `count as u16`, deduplicate.rs:220
        }
        carry += tot;
    }
    if (threadIdx.x == 0) o_n[cell] = carry;
}

// per-cell results (at the cell's input offset) -> one dense run per array, cells in order
__global__ __launch_bounds__(256) void k_atac_compact(const uint64_t* __restrict__ cell_ptr, const uint64_t* __restrict__ out_ptr,
                                                     const uint32_t* __restrict__ i_ref, const uint32_t* __restrict__ i_start,
                                                     const uint16_t* __restrict__ i_flen, const uint16_t* __restrict__ i_cnt,
                                                     uint32_t* __restrict__ o_ref, uint32_t* __restrict__ o_start,
                                                     uint16_t* __restrict__ o_flen, uint16_t* __restrict__ o_cnt,
                                                     unsigned long long* __restrict__ tally, uint4* __restrict__ runs,
                                                     uint32_t* __restrict__ run_ctr, uint32_t run_cap) {
    const uint32_t cell = blockIdx.x;
    const uint64_t src = cell_ptr[cell], dst = out_ptr[cell];
    const uint32_t n = (uint32_t)(out_ptr[cell + 1] - dst);
    uint32_t dup = 0, lng = 0;   // fragments seen more than once / of 2000 bases and more (deduplicate.rs:222-224, 47-63)
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
        const uint16_t fl = i_flen[src + i], ct = i_cnt[src + i];
        const uint32_t r = i_ref[src + i];
        o_ref[dst + i] = r; o_start[dst + i] = i_start[src + i];
        o_flen[dst + i] = fl; o_cnt[dst + i] = ct;
        dup += ct > 1; lng += fl >= 2000;
        // A cell's rows are in (ref, start, frag_len) order, so its ref column is a few runs: (first row, length, ref) of each
        // goes to the host's list - `runs` is host memory as the device sees it - and the column itself stays here (the host
        // writes it from the list while the other three columns cross PCIe; afq_api.cpp).  The row that starts a run finds its end
        // by bisection.
        if (runs && (i == 0 || i_ref[src + i - 1] != r)) {
            uint32_t lo = i + 1, hi = n;
            while (lo < hi) {
                const uint32_t mid = lo + ((hi - lo) >> 1);
                if (i_ref[src + mid] == r) lo = mid + 1; else hi = mid;
            }
            const uint32_t at = atomicAdd(run_ctr, 1u);
            if (at < run_cap) runs[at] = make_uint4((uint32_t)(dst + i), (uint32_t)((dst + i) >> 32), lo - i, r);
        }
    }
    if (tally) {
        for (int d = 32; d; d >>= 1) { dup += __shfl_xor(dup, d); lng += __shfl_xor(lng, d); }
        if ((threadIdx.x & 63) == 0) { if (dup) atomicAdd(&tally[0], (unsigned long long)dup); if (lng) atomicAdd(&tally[1], (unsigned long long)lng); }
    }
}

// ---------------------------------------------------------------------------
// launchers







// first touch of a kernel loads the library's code object on the current device (tens of ms): afq_device_warmup does it early
void warm_code_object() {
    hipFuncAttributes at{};
    (void)hipFuncGetAttributes(&at, reinterpret_cast<const void*>(k_hist));
}

void launch_hist(hipStream_t s, const ResolveArgs& a) {
    if (!a.n_tiles) return;
    AFQ_LAUNCH(k_hist, a.n_tiles, 256, s, a.tile_desc, a.meta, a.cell_nkeys, a.keys0, a.cursor);
}

void launch_bucket_scan(hipStream_t s, const ResolveArgs& a) {
    if (!a.n_multi) return;
    AFQ_LAUNCH(k_bucket_scan, (a.n_multi + 3) / 4, 256, s, a.multi_cells, a.n_multi, a.meta, a.cursor);
}

void launch_scatter(hipStream_t s, const ResolveArgs& a) {
    if (!a.n_tiles) return;
    if ((1u << a.max_lg_nb) <= 512u)
        AFQ_LAUNCH(k_scatter<512>, (a.n_tiles + 8 * kScatterRun - 1) / (8 * kScatterRun) * (8 * kScatterRun), 256, s, a.tile_desc, a.meta, a.cell_nkeys, a.keys0, a.keys1, a.cursor, a.slab_ovf, a.n_tiles);
    else
        AFQ_LAUNCH(k_scatter<kLdsBins>, (a.n_tiles + 8 * kScatterRun - 1) / (8 * kScatterRun) * (8 * kScatterRun), 256, s, a.tile_desc, a.meta, a.cell_nkeys, a.keys0, a.keys1, a.cursor, a.slab_ovf, a.n_tiles);
}

void launch_fix_slabs(hipStream_t s, const ResolveArgs& a) {
    if (!a.n_multi || !a.slabs) return;
    const uint32_t grid = a.n_multi < 512u ? a.n_multi : 512u;
    AFQ_LAUNCH(k_fix_slabs, grid, 256, s, a.multi_cells, a.n_multi, a.meta, a.cell_nkeys, a.keys0, a.keys1, a.cursor, a.slab_ovf);
}

static ResolveCfg make_rc(const ResolveArgs& a) {
    ResolveCfg rc;
    rc.usa = a.usa; rc.num_rows = a.num_rows; rc.uo = a.num_rows / 3; rc.ao = 2 * (a.num_rows / 3); rc.mode = 0;
    rc.pa = a.prefer_ambig;
    rc.sort_only = a.sort_only;
    return rc;
}

void launch_resolve(hipStream_t s, const ResolveArgs& a) {
    if (!a.n_buckets) return;
    ResolveCfg rc = make_rc(a);
    LabArea la{a.lab, a.lab_cnt};
    BucketDesc* desc = reinterpret_cast<BucketDesc*>(a.bucket_desc);
    AFQ_LAUNCH(k_bucket_desc, (a.n_buckets + 255) / 256, 256, s, a.meta, a.bucket_cell, a.cell_nkeys, a.cursor, a.slab_ovf, a.n_buckets, desc);
    const uint32_t n_cols = a.n_buckets < kResolveCols ? a.n_buckets : kResolveCols;
    const uint32_t grid = n_cols * ((a.n_buckets + n_cols - 1) / n_cols);
    if (a.lab)
        AFQ_LAUNCH((k_resolve<true, false>), grid, kResolveNT, s, desc, a.n_buckets, a.keys0, a.keys1, a.cell_ncols, a.nnz, a.ovf_list, a.st, rc, la);
    else if (a.sort_only)
        AFQ_LAUNCH((k_resolve<false, true>), grid, kResolveNT, s, desc, a.n_buckets, a.keys0, a.keys1, a.cell_ncols, a.nnz, a.ovf_list, a.st, rc, la);
    else
        AFQ_LAUNCH((k_resolve<false, false>), grid, kResolveNT, s, desc, a.n_buckets, a.keys0, a.keys1, a.cell_ncols, a.nnz, a.ovf_list, a.st, rc, la);
}

void launch_resolve_big(hipStream_t s, const ResolveArgs& a) {
    if (!a.n_multi) return;
    ResolveCfg rc = make_rc(a);
    LabArea la{a.lab, a.lab_cnt};
    BucketDesc* desc = reinterpret_cast<BucketDesc*>(a.bucket_desc);
    AFQ_LAUNCH(k_resolve_mid, 256, kMidNT, s, desc, a.meta, a.keys0, a.keys1, a.cell_ncols, a.nnz, a.ovf_list, a.st, rc, la);   // (the buckets beyond LDS reach too)
}




size_t bucket_desc_bytes() { return sizeof(BucketDesc); }
#ifdef AFQ_RESOLVE_TIMING
extern "C" void afq_debug_dump() {
    unsigned long long h[8];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(g_dbg), sizeof(h));
    fprintf(stderr, "[resolve cycles/bucket] load+clear=%llu insert=%llu emit=%llu tail=%llu (n=%llu)\n", h[0] / (h[4] + 1), h[1] / (h[5] + 1), h[2] / (h[6] + 1), h[3] / (h[7] + 1), h[4]);
}
#endif

void launch_atac_dedup(hipStream_t s, uint32_t n_cells, const uint32_t* ref, const uint32_t* start, const uint16_t* flen,
                       const uint64_t* cell_ptr, void* scratch, uint32_t* o_ref, uint32_t* o_start, uint16_t* o_flen,
                       uint16_t* o_cnt, uint32_t* o_n, const uint32_t* cell_cnt) {
    if (!n_cells) return;
    AFQ_LAUNCH(k_atac_dedup, n_cells, kAtacNT, s, ref, start, flen, cell_ptr, reinterpret_cast<Frag*>(scratch), o_ref, o_start,
               o_flen, o_cnt, o_n, cell_cnt);
}

void launch_atac_dedup64(hipStream_t s, uint32_t n_cells, const uint32_t* ref, const uint32_t* start, const uint16_t* flen,
                         const uint64_t* cell_ptr, void* scratch, uint32_t* o_ref, uint32_t* o_start, uint16_t* o_flen,
                         uint16_t* o_cnt, uint32_t* o_n, uint32_t* flag, const uint32_t* cell_cnt) {
    if (!n_cells) return;
    AFQ_LAUNCH(k_atac_dedup64, n_cells, kAtacNT, s, ref, start, flen, cell_ptr, reinterpret_cast<uint64_t*>(scratch), o_ref, o_start,
               o_flen, o_cnt, o_n, flag, cell_cnt);
}

void launch_atac_compact(hipStream_t s, uint32_t n_cells, const uint64_t* cell_ptr, const uint64_t* out_ptr, const uint32_t* i_ref,
                         const uint32_t* i_start, const uint16_t* i_flen, const uint16_t* i_cnt, uint32_t* o_ref, uint32_t* o_start,
                         uint16_t* o_flen, uint16_t* o_cnt, unsigned long long* tally, uint4* runs, uint32_t* run_ctr, uint32_t run_cap) {
    if (!n_cells) return;
    AFQ_LAUNCH(k_atac_compact, n_cells, 256, s, cell_ptr, out_ptr, i_ref, i_start, i_flen, i_cnt, o_ref, o_start, o_flen, o_cnt, tally, runs, run_ctr,
               run_cap);
}

void launch_cell_hist(hipStream_t s, const ResolveArgs& a) {
    if (!a.n_hist) return;
    ResolveCfg rc = make_rc(a);
    uint32_t words = (a.num_rows + 1) / 2;
    words = words < 4096u ? 4096u : (words > kHistBins ? kHistBins : words);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_cell_hist), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (int)kHistBins);   // (above 64 KiB needs asking)
    hipLaunchKernelGGL(k_cell_hist, dim3(a.n_hist), dim3(kHistNT), 4 * words, s, a.hist_cells, a.meta, a.keys0, a.keys1, a.cell_ncols, a.nnz, rc, words);
}

void launch_compact(hipStream_t s, const CellMeta* meta, uint32_t n_cells, const uint64_t* keys0, const uint64_t* keys1,
                    const uint32_t* nnz, const uint64_t* cell_ptr, uint32_t* gene, float* val, uint64_t cap) {
    if (!n_cells) return;
    AFQ_LAUNCH(k_compact, (n_cells + 3) / 4, 256, s, meta, n_cells, keys0, keys1, nnz, cell_ptr, gene, val, cap);
}

// Row offsets of a range from its row lengths, on the device (one workgroup: n is a range's cells): cell_ptr[i] = nnz[0] + ... +
// nnz[i - 1], cell_ptr[n] = the range's entries.  The host makes the same sums from the packed block it reads anyway; this
// copy lets the compaction follow the range's kernels without waiting for the host.
// (Round 6: the same launch packs what the host reads when the range is done - status block, flags, row lengths, barcodes, into
//  pinned host memory: k_pack_small's work, which was a 5 us launch of its own in front of this one; pk.out == nullptr: not asked.)
__global__ __launch_bounds__(1024) void k_row_ptr(const uint32_t* __restrict__ nnz, uint32_t n, uint64_t* __restrict__ cell_ptr, PackSmallArgs pk) {
    __shared__ uint32_t s_ws[16];
    if (pk.out) {
        for (uint32_t i = threadIdx.x; i < (n > 16u ? n : 16u); i += 1024) {
            if (i < sizeof(DevStatus) / 4) pk.out[i] = reinterpret_cast<const uint32_t*>(pk.st)[i];
            if (i == 8) pk.out[8] = pk.em_flag ? *pk.em_flag : 0u;
            if (i == 9) pk.out[9] = pk.n_mono ? *pk.n_mono : 0u;
            if (i < n) {
                pk.out[16 + i] = pk.alt[i];
                pk.out[16 + n + i] = nnz[i];
                pk.out[16 + 2 * (size_t)n + i] = pk.em_nnz ? pk.em_nnz[i] : 0u;
                const uint64_t b = pk.bc[i];
                pk.out[16 + 3 * (size_t)n + 2 * (size_t)i] = (uint32_t)b;
                pk.out[16 + 3 * (size_t)n + 2 * (size_t)i + 1] = (uint32_t)(b >> 32);
            }
        }
    }
    uint64_t carry = 0;
    for (uint32_t base = 0; base < n; base += 1024) {   // (a row has at most 2^20 entries: 1024 of them fit 32 bits)
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n ? nnz[i] : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan<1024>(v, s_ws, tot);
        if (i < n) cell_ptr[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) cell_ptr[n] = carry;
}
// The two tables of a range that only restate the cells' plans - bucket -> cell (k_bucket_desc's lookup) and scatter tile ->
// (cell, tile of the cell) - written from the plans, a wave per cell.  They are 4 bytes per bucket and 8 per tile (2.8 + 0.5 MB
// for the first range of a PBMC-10k batch) and used to be filled on the host and uploaded in front of the range's first kernel.
__global__ __launch_bounds__(256) void k_fill_tables(const CellMeta* __restrict__ meta, uint32_t n_cells, uint32_t* __restrict__ bucket_cell,
                                                    uint2* __restrict__ tile_desc) {
    const uint32_t cell = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (cell >= n_cells) return;
    const CellMeta m = meta[cell];
    const uint32_t nb = 1u << m.lg_nb;
    for (uint32_t b = lane_id(); b < nb; b += 64) bucket_cell[m.bucket_base + b] = cell;
    if (m.lg_nb == 0) return;   // (single-bucket and parsimony cells have no tiles)
    const uint32_t nt = (m.n_ref + kTileKeys - 1) / kTileKeys;
    for (uint32_t t = lane_id(); t < nt; t += 64) tile_desc[m.tile_base + t] = make_uint2(cell, t);
}
void launch_fill_tables(hipStream_t s, const CellMeta* meta, uint32_t n_cells, uint32_t* bucket_cell, uint2* tile_desc) {
    if (!n_cells) return;
    AFQ_LAUNCH(k_fill_tables, (n_cells + 3) / 4, 256, s, meta, n_cells, bucket_cell, tile_desc);
}
void launch_row_ptr(hipStream_t s, const uint32_t* nnz, uint32_t n, uint64_t* cell_ptr, const PackSmallArgs& pk) {
    AFQ_LAUNCH(k_row_ptr, 1, 1024, s, nnz, n, cell_ptr, pk);
}

}  // namespace afq
