// afq_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the quant hot path.
//
// Pipeline for winner-take-all ("cr-like") resolution of a batch of cells
// (replaces src/quant.rs:469-657 / src/pugutils.rs:644-850 / src/utils.rs:673-756
// of the reference; semantics: SURVEY.md appendix B.2):
//   k_decode        raw collated-RAD chunk bytes -> (umi<<20|gene) keys, one per
//                   (read, distinct gene); one wave walks one chunk window by window
//   k_bucket_scan   per-cell exclusive scan of the UMI-hash bucket histogram
//   k_scatter       keys -> per-(cell,bucket) ranges
//   k_resolve       one workgroup per bucket: LDS bitonic sort, run-length count of
//                   (umi,gene), per-UMI arg-max with ties, USA slot rules, then
//                   either finishes the cell in LDS (single-bucket cells) or adds
//                   into the cell's dense count row (multi-bucket cells)
//   k_resolve_big   same algorithm out of global scratch for buckets over the LDS cap
//   k_extract_dense dense row -> sorted (column,count) pairs
//   k_compact       per-cell pairs -> final CSR
// Integer/byte work bound by HBM and LDS; no MFMA anywhere by design.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "afq_common.h"
#include "afq_kernels.h"
#include "afq_prims.h"

namespace afq {

// ---------------------------------------------------------------------------
// chunk headers of device-resident input -> (nbytes, nrec) per cell
__global__ void k_gather_headers(const uint8_t* __restrict__ bytes, size_t n_bytes,
                                 const uint64_t* __restrict__ chunk_off, uint32_t n_cells,
                                 uint32_t* __restrict__ hdr) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_cells) return;
    uint64_t off = chunk_off[i];
    uint32_t a = 0, b = 0;
    if (off + 8 <= n_bytes) {
        a = (uint32_t)ld_le<4>(bytes + off);
        b = (uint32_t)ld_le<4>(bytes + off + 4);
    }
    hdr[2 * i] = a;
    hdr[2 * i + 1] = b;
}

// ---------------------------------------------------------------------------
// k_decode: one wave per chunk.  The record stream has no self-synchronisation
// (a record's length is its own na field), so the wave walks it: each 256-byte
// window is loaded coalesced (one dword per lane), a scalar loop follows
// na -> next-record with v_readlane, marking the lanes whose dword starts a
// record; those lanes then decode their record in parallel (gene projection =
// the per-read sort+dedup of src/pugutils.rs:774-781, done as first-occurrence
// dedup since the key order is re-established by the bucket sort).
template <int BW, int UW>
__global__ __launch_bounds__(256) void k_decode(const uint8_t* __restrict__ bytes, size_t n_bytes,
                                               const CellMeta* __restrict__ meta, uint32_t n_cells,
                                               const uint32_t* __restrict__ t2g, uint32_t ref_count,
                                               uint32_t num_genes, uint64_t* __restrict__ keys0,
                                               uint32_t* __restrict__ cell_nkeys,
                                               uint64_t* __restrict__ bc_out, DevStatus* st,
                                               const uint32_t* __restrict__ fix_list, PugOut pug) {
    constexpr uint32_t HDR = 4 + BW + UW;
    constexpr bool AL = (BW % 4 == 0) && (UW % 4 == 0);
    const uint32_t lane = lane_id();
    // plain mode: wave w takes cell w.  Fix-up mode (after a walk-free decoder): the waves loop over the
    // cells k_verify_cells listed as failing the proof (normally none) and re-decode them here.
    const uint32_t n_work = fix_list ? st->n_fallback : n_cells;
  for (uint32_t work = blockIdx.x * 4 + (threadIdx.x >> 6); work < n_work; work += gridDim.x * 4) {
    const uint32_t cell = fix_list ? fix_list[work] : work;
    const CellMeta m = meta[cell];
    const uint64_t abase = m.chunk_off & ~3ull;         // dword-aligned base of the walk
    const uint32_t mis = (uint32_t)(m.chunk_off - abase);
    uint64_t pos = (uint64_t)mis + 8;                    // next record start, bytes from abase
    const uint64_t end = (uint64_t)mis + m.nbytes;       // chunk end, bytes from abase
    const bool al_chunk = AL && mis == 0;
    uint32_t nk_total = 0, rec_seen = 0;
    bool bad = false;

    while (pos < end) {
        const uint64_t w = pos >> 8;  // window index
        const uint64_t wbyte = abase + (w << 8) + lane * 4;
        uint32_t v_cur = 0, v_next = 0;
        if (wbyte + 4 <= n_bytes) v_cur = *(const uint32_t*)(bytes + wbyte);
        else if (wbyte < n_bytes) { for (uint64_t q = wbyte; q < n_bytes; ++q) v_cur |= (uint32_t)bytes[q] << (8 * (q - wbyte)); }
        if (!al_chunk) {
            const uint64_t nb = wbyte + 256;
            if (nb + 4 <= n_bytes) v_next = *(const uint32_t*)(bytes + nb);
            else if (nb < n_bytes) { for (uint64_t q = nb; q < n_bytes; ++q) v_next |= (uint32_t)bytes[q] << (8 * (q - nb)); }
        }
        const uint64_t wend = ((w + 1) << 8) < end ? ((w + 1) << 8) : end;
        uint64_t mask = 0, sub0 = 0, sub1 = 0;
        // scalar walk over the records that start in this window
        while (pos < wend) {
            const uint32_t idx = __builtin_amdgcn_readfirstlane((uint32_t)(pos >> 2) & 63u);
            uint32_t na = __builtin_amdgcn_readlane(v_cur, idx);
            if (!al_chunk) {
                const uint32_t sh = ((uint32_t)pos & 3u) * 8u;
                if (sh) {
                    uint32_t hi = idx < 63 ? __builtin_amdgcn_readlane(v_cur, idx + 1)
                                           : __builtin_amdgcn_readlane(v_next, 0);
                    na = (na >> sh) | (hi << (32 - sh));
                }
                if (pos & 1) sub0 |= 1ull << idx;
                if (pos & 2) sub1 |= 1ull << idx;
            }
            mask |= 1ull << idx;
            const uint64_t rec_bytes = (uint64_t)HDR + 4ull * na;
            if (pos + rec_bytes > end) { bad = true; pos = end; break; }
            pos += rec_bytes;
        }
        rec_seen += (uint32_t)__popcll(mask);

        // lanes whose dword starts a record decode it
        const bool is_start = (mask >> lane) & 1ull;
        uint32_t g[8];
        uint32_t k = 0, na = 0, kcnt = 0, rec_dw = 0;
        bool ovf = false, pug_rec = false;
        uint64_t umi = 0, lhash = 0;
        const uint8_t* rp = nullptr;
        if (is_start && !bad) {
            const uint32_t sub = al_chunk ? 0u : (uint32_t)((sub0 >> lane) & 1ull) | ((uint32_t)((sub1 >> lane) & 1ull) << 1);
            const uint64_t roff = abase + (w << 8) + lane * 4 + sub;
            const uint8_t* rec = bytes + roff;
            na = al_chunk ? v_cur : ld_u32(rec, false);
            umi = ld_le<UW>(rec + 4 + BW);
            if (roff == m.chunk_off + 8) bc_out[cell] = ld_le<BW>(rec + 4);
            if (UW == 8 && (umi >> kUmiBits)) { set_err(st, kErrUmiWide, cell); na = 0; }
            rp = rec + HDR;
            const bool ral = ((((uintptr_t)rp) & 3) == 0);
            if (mode_is_pug(m.mode)) { rec_dw = (uint32_t)((roff - m.chunk_off) >> 2); pug_rec = true; }
            if (mode_is_pug(m.mode) && !mode_pug_gene(m.mode)) {  // txp-level PUG: hash of the ref list
                lhash = label_hash_init(na);
                uint32_t t01[2] = {0, 0};
                for (uint32_t j = 0; j < na; ++j) {
                    const uint32_t t = ld_u32(rp + 4 * j, ral) & 0x7FFFFFFFu;
                    if (t >= ref_count) set_err(st, kErrRefRange, cell);
                    lhash = label_hash_step(lhash, t);
                    if (j < 2) t01[j] = t;
                }
                lhash = label_key(lhash, na, t01[0], t01[1]);
                na = 0;
            }
            for (uint32_t j = 0; j < na; ++j) {
                uint32_t t = ld_u32(rp + 4 * j, ral) & 0x7FFFFFFFu;
                if (t >= ref_count) { set_err(st, kErrRefRange, cell); continue; }
                uint32_t gid = t2g[t];
                if (gid >= num_genes) { set_err(st, kErrGeneRange, cell); continue; }
                bool dup = false;
#pragma unroll
                for (int i = 0; i < 8; ++i) dup |= ((uint32_t)i < k) && (g[i] == gid);
                if (!dup) {
                    if (k < 8) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) if ((uint32_t)i == k) g[i] = gid;
                        ++k;
                    } else { ovf = true; break; }
                }
            }
            kcnt = k;
            if (ovf) {  // > 8 distinct genes: count by first occurrence, O(na^2), rare
                kcnt = 0;
                for (uint32_t j = 0; j < na; ++j) {
                    uint32_t tj = ld_u32(rp + 4 * j, ral) & 0x7FFFFFFFu;
                    if (tj >= ref_count) continue;
                    uint32_t gj = t2g[tj];
                    if (gj >= num_genes) continue;
                    bool first = true;
                    for (uint32_t i = 0; i < j && first; ++i) {
                        uint32_t ti = ld_u32(rp + 4 * i, ral) & 0x7FFFFFFFu;
                        if (ti < ref_count && t2g[ti] == gj) first = false;
                    }
                    kcnt += first;
                    if (first && mode_pug_gene(m.mode)) lhash += gene_set_hash_term(gj);
                }
            }
            if (mode_pug_gene(m.mode)) {  // gene-level PUG: order-independent hash of the read's gene set
                if (!ovf) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) if ((uint32_t)i < k) lhash += gene_set_hash_term(g[i]);
                }
                lhash ^= (uint64_t)kcnt * kHashMul;
                lhash = label_key(lhash, kcnt, g[0], g[1]);  // (kcnt <= 2 implies !ovf: g[0], g[1] are the read's genes)
            }
        }
        if (m.mode == kModeTrivial && kcnt != 1) kcnt = 0;  // multi-gene reads are discarded (pugutils.rs:870-891)
        if (pug_rec) kcnt = 1;
        uint32_t tot;
        const uint32_t ex = wave_excl_scan(kcnt, tot);
        if (pug_rec) {
            const uint32_t o0 = nk_total + ex;
            if (o0 < m.nrec) {
                const uint64_t slot = pug.rd_off[cell] + o0;
                pug.h[slot] = lhash; pug.u[slot] = umi; pug.o[slot] = rec_dw;
            } else bad = true;
        } else if (kcnt) {
            const uint32_t o0 = nk_total + ex;
            uint64_t* dst = keys0 + m.key_off;
            if (o0 + kcnt <= m.n_ref) {
                if (!ovf) {
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if ((uint32_t)i < k) dst[o0 + i] = (umi << kGeneBits) | g[i];
                } else {
                    const bool ral = ((((uintptr_t)rp) & 3) == 0);
                    uint32_t o = o0;
                    for (uint32_t j = 0; j < na; ++j) {
                        uint32_t tj = ld_u32(rp + 4 * j, ral) & 0x7FFFFFFFu;
                        if (tj >= ref_count) continue;
                        uint32_t gj = t2g[tj];
                        if (gj >= num_genes) continue;
                        bool first = true;
                        for (uint32_t i = 0; i < j && first; ++i) {
                            uint32_t ti = ld_u32(rp + 4 * i, ral) & 0x7FFFFFFFu;
                            if (ti < ref_count && t2g[ti] == gj) first = false;
                        }
                        if (first) dst[o++] = (umi << kGeneBits) | gj;
                    }
                }
            } else bad = true;
        }
        nk_total += tot;
        bad = __any(bad);
        if (bad) break;
    }
    if (bad || pos != end || rec_seen != m.nrec) {
        if (lane == 0) set_err(st, kErrRecordWalk, cell);
        nk_total = 0;
    }
    if (lane == 0) {
        cell_nkeys[cell] = nk_total;
        atomicAdd(&st->n_keys, (unsigned long long)nk_total);
    }
  }
}

// Walk-free proof, final step: per cell compare the accumulated candidate count and sizes
// with the chunk header; cells that pass add their key count to the batch total (one atomic per workgroup),
// cells that fail are listed for the sequential re-decode.
__global__ __launch_bounds__(256) void k_verify_cells(const CellMeta* __restrict__ meta, uint32_t n_cells,
                                                     const CellChk* __restrict__ chk,
                                                     const uint32_t* __restrict__ cell_nkeys, DevStatus* st,
                                                     uint32_t* __restrict__ fix_list) {
    __shared__ unsigned long long s_sum;
    if (threadIdx.x == 0) s_sum = 0;
    __syncthreads();
    const uint32_t cell = blockIdx.x * 256 + threadIdx.x;
    unsigned long long keys = 0;
    if (cell < n_cells) {
        const CellMeta m = meta[cell];
        const CellChk c = chk[cell];
        if (c.fail == 0 && c.count == m.nrec && c.words == m.nbytes / 4 - 2) keys = cell_nkeys[cell];
        else fix_list[atomicAdd(&st->n_fallback, 1u)] = cell;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) keys += __shfl_xor(keys, d);
    if (lane_id() == 0 && keys) atomicAdd(&s_sum, keys);
    __syncthreads();
    if (threadIdx.x == 0 && s_sum) atomicAdd(&st->n_keys, s_sum);
}

// ---------------------------------------------------------------------------
// k_slab_setup: per cell, record which cell every 1 KiB slab belongs to and the
// cell's barcode words, so the decode waves start with one dependent load, not five.
template <int BW, int UW>
__global__ __launch_bounds__(256) void k_slab_setup(const uint8_t* __restrict__ bytes,
                                                   const CellMeta* __restrict__ meta, uint32_t n_cells,
                                                   const uint32_t* __restrict__ slab_prefix,
                                                   uint32_t* __restrict__ slab_cell, uint64_t* __restrict__ cell_bc) {
    constexpr uint32_t BWW = BW / 4, UWW = UW / 4, HW = 1 + BWW + UWW;
    const uint32_t cell = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (cell >= n_cells) return;
    const uint32_t a = slab_prefix[cell], b = slab_prefix[cell + 1];
    for (uint32_t s = a + lane_id(); s < b; s += 64) slab_cell[s] = cell;
    if (lane_id() == 0) {
        const CellMeta m = meta[cell];
        const uint32_t* W = reinterpret_cast<const uint32_t*>(bytes + m.chunk_off);
        uint64_t bc = 0;
        if ((m.nbytes >> 2) >= 2 + HW) bc = BWW == 2 ? ((uint64_t)W[4] << 32 | W[3]) : (uint64_t)W[3];
        cell_bc[cell] = bc;
    }
}

// ---------------------------------------------------------------------------
// k_decode_par: walk-free decode for dword-aligned layouts (bc/umi of 4 or 8 bytes).
// In a collated chunk every record carries the cell's barcode, so a record start
// is a dword i whose barcode field equals the barcode of the chunk's first record.
// One wave takes kSlabsPerWave consecutive 1 KiB slabs.  Per slab: stage the slab
// (+ a 64-dword halo) in LDS with coalesced loads, ballot the candidate starts of
// its four 64-dword windows into an LDS list, then one lane per candidate decodes
// the record out of LDS (na, umi, refs), gathers tid_to_gid and emits keys.  The
// raw dwords of the next slab are requested before the current slab's gathers are
// consumed, so the HBM latency of the stream overlaps the L2 latency of the gathers.
// Per candidate it also checks that the position right after the record is again a
// candidate (or the chunk end); with the per-cell sums of candidate count and
// candidate sizes this proves the candidate set IS the sequential parse
// (DESIGN.md "walk-free decode").  Cells that fail the proof are re-decoded by
// the sequential k_decode, so a barcode-valued UMI/ref word costs time, never
// correctness.  Keys of a cell land in arbitrary order (wave-level atomic
// reservation); order is re-established by the bucket sort.
constexpr uint32_t kSlabsPerWave = 4;
constexpr uint32_t kHalo = 64;
constexpr uint32_t kDecodeCols = 8192;
constexpr uint32_t kStage = kSlabWords + kHalo;  // 320 dwords = 5 per lane

template <int BW, int UW, bool PUG>
__global__ __launch_bounds__(256, 6) void k_decode_par(const uint8_t* __restrict__ bytes,
                                                   const CellMeta* __restrict__ meta, uint32_t n_cells,
                                                   const uint32_t* __restrict__ slab_prefix,
                                                   const uint32_t* __restrict__ slab_cell,
                                                   const uint64_t* __restrict__ cell_bc, uint32_t n_slabs,
                                                   const uint32_t* __restrict__ t2g, uint32_t ref_count,
                                                   uint32_t num_genes, uint64_t* __restrict__ keys0,
                                                   uint32_t* __restrict__ cell_nkeys,
                                                   uint64_t* __restrict__ bc_out, CellChk* __restrict__ chk,
                                                   PugOut pug) {
    static_assert(BW % 4 == 0 && UW % 4 == 0, "aligned layouts only");
    constexpr uint32_t BWW = BW / 4, UWW = UW / 4, HW = 1 + BWW + UWW;
    __shared__ uint32_t s_stage[4][kStage];
    __shared__ uint32_t s_list[4][kSlabWords];
    const uint32_t lane = lane_id();
    const uint32_t wv = threadIdx.x >> 6;
    uint32_t* stage = s_stage[wv];
    uint32_t* list = s_list[wv];
    // Waves that run at the same time are spread over the whole input (column-major walk of the
    // slab groups): neighbouring groups belong to one cell and would serialise on that cell's
    // key-reservation counter (same-address device atomics).
    const uint32_t n_groups = (n_slabs + kSlabsPerWave - 1) / kSlabsPerWave;
    const uint32_t n_cols = min(n_groups, kDecodeCols);
    const uint32_t n_rows = (n_groups + n_cols - 1) / n_cols;
    const uint32_t wid = blockIdx.x * 4 + wv;
    const uint32_t grp = (wid % n_cols) * n_rows + wid / n_cols;
    if (wid >= n_cols * n_rows || grp >= n_groups) return;
    const uint32_t slab_a = grp * kSlabsPerWave;
    const uint32_t slab_b = min(n_slabs, slab_a + kSlabsPerWave);
    // the (up to 4) cells of this wave's slabs
    uint32_t my_cell = 0;
    if (lane < slab_b - slab_a) my_cell = slab_cell[slab_a + lane];

    uint32_t cur_cell = 0xFFFFFFFFu;
    CellMeta m{};
    const uint32_t* __restrict__ W = nullptr;
    uint32_t nwords = 0, sp0 = 0, bc_lo = 0, bc_hi = 0;
    uint32_t acc_count = 0, acc_words = 0;
    bool fail = false;
    uint32_t R[5];

    auto load_cell = [&](uint32_t cell) {
        cur_cell = cell;
        m = meta[cell];
        const uint64_t bc = cell_bc[cell];
        bc_lo = (uint32_t)bc; bc_hi = (uint32_t)(bc >> 32);
        W = reinterpret_cast<const uint32_t*>(bytes + m.chunk_off);
        nwords = m.nbytes >> 2;
        sp0 = slab_prefix[cell];
    };
    auto flush_chk = [&]() {
        if (cur_cell == 0xFFFFFFFFu) return;
        uint32_t ws = acc_words;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) ws += __shfl_xor(ws, d);
        const bool any_fail = __any(fail);
        if (lane == 0) {
            if (acc_count) atomicAdd(&chk[cur_cell].count, acc_count);
            if (ws) atomicAdd(&chk[cur_cell].words, ws);
            if (any_fail) atomicOr(&chk[cur_cell].fail, 1u);
        }
        acc_count = 0; acc_words = 0; fail = false;
    };
    auto issue_slab_loads = [&](uint32_t s0) {
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            const uint32_t i = s0 + r * 64 + lane;
            R[r] = i < nwords ? W[i] : 0u;
        }
    };

    load_cell(__builtin_amdgcn_readlane(my_cell, 0));
    issue_slab_loads((slab_a - sp0) * kSlabWords);

    for (uint32_t slab = slab_a; slab < slab_b; ++slab) {
        const uint32_t s0 = (slab - sp0) * kSlabWords;
        // stage this slab (its dwords were requested one iteration ago)
#pragma unroll
        for (int r = 0; r < 5; ++r) stage[r * 64 + lane] = R[r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint32_t ncand = 0;
        if (nwords >= 2 + HW) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t il = r * 64 + lane, i = s0 + il;
                bool cand = false;
                if (i >= 2 && i + HW <= nwords) {
                    cand = stage[il + 1] == bc_lo;
                    if (BWW == 2) cand = cand && (stage[il + 2] == bc_hi);
                }
                const uint64_t mk = __ballot(cand);
                if (cand) list[ncand + __popcll(mk & ((1ull << lane) - 1))] = il;
                ncand += (uint32_t)__popcll(mk);
            }
        } else if (s0 == 0) fail = true;  // cannot hold a record; nrec >= 1 is guaranteed by the planner
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (s0 == 0 && nwords >= 2 + HW) {  // (1) the first record starts right after the chunk header
            if (!(ncand > 0 && list[0] == 2)) fail = true;
        }
        acc_count += ncand;
        // decide what the next iteration needs before the long-latency part
        const bool has_next = slab + 1 < slab_b;
        const uint32_t next_cell = has_next ? __builtin_amdgcn_readlane(my_cell, (int)(slab + 1 - slab_a)) : cur_cell;
        const bool same_next = has_next && next_cell == cur_cell;
        bool prefetched = false;

        for (uint32_t base = 0; base < ncand; base += 64) {
            const uint32_t c = base + lane;
            const bool act = c < ncand;
            uint32_t il = 0, i = 0, na = 0, kcnt = 0, k = 0;
            bool ovf = false;
            uint64_t umi = 0;
            uint32_t g[8];
            auto refw = [&](uint32_t j) -> uint32_t {  // j-th alignment word of this lane's record
                const uint32_t p = il + HW + j;
                return (p < kStage ? stage[p] : W[i + HW + j]) & 0x7FFFFFFFu;
            };
            uint32_t gid0 = 0;
            bool ok0 = false;
            if (act) {
                il = list[c];
                i = s0 + il;
                na = stage[il];
                if (na > nwords || i + HW + na > nwords) { fail = true; na = 0; }
                else {
                    const uint32_t succ = i + HW + na, sl = il + HW + na;  // (2) the next record starts where this one ends
                    if (succ != nwords) {
                        bool ok = succ + HW <= nwords;
                        if (ok) {
                            const uint32_t w1 = sl + 1 < kStage ? stage[sl + 1] : W[succ + 1];
                            ok = w1 == bc_lo;
                            if (BWW == 2 && ok) ok = (sl + 2 < kStage ? stage[sl + 2] : W[succ + 2]) == bc_hi;
                        }
                        if (!ok) fail = true;
                    }
                    acc_words += HW + na;
                    umi = stage[il + 1 + BWW];
                    if (UWW == 2) umi |= (uint64_t)stage[il + 2 + BWW] << 32;
                    if (UWW == 2 && (umi >> kUmiBits)) fail = true;
                    if (i == 2) bc_out[cur_cell] = BWW == 2 ? ((uint64_t)bc_hi << 32 | bc_lo) : (uint64_t)bc_lo;
                    if (na) {
                        const uint32_t t = refw(0);
                        if (t < ref_count) { gid0 = t2g[t]; ok0 = true; } else fail = true;
                    }
                }
            }
            // request the next slab's dwords while the gathers above are in flight
            if (!prefetched && same_next) { issue_slab_loads(s0 + kSlabWords); prefetched = true; }
            const bool pug_rec = PUG && act && mode_is_pug(m.mode);
            uint64_t lhash = 0;
            const bool pug_gene = pug_rec && mode_pug_gene(m.mode);
            if (pug_rec && !pug_gene) {  // txp-level PUG: hash of the ref list
                lhash = label_hash_init(na);
                uint32_t t0 = 0, t1 = 0;
                for (uint32_t j = 0; j < na; ++j) {
                    const uint32_t t = refw(j);
                    if (t >= ref_count) fail = true;
                    lhash = label_hash_step(lhash, t);
                    if (j == 0) t0 = t;
                    if (j == 1) t1 = t;
                }
                lhash = label_key(lhash, na, t0, t1);
            }
            if (act && na && (!pug_rec || pug_gene)) {
                if (ok0) {
                    if (gid0 < num_genes) { g[0] = gid0; k = 1; } else fail = true;
                }
                for (uint32_t j = 1; j < na; ++j) {
                    const uint32_t t = refw(j);
                    if (t >= ref_count) { fail = true; continue; }
                    const uint32_t gid = t2g[t];
                    if (gid >= num_genes) { fail = true; continue; }
                    bool dup = false;
#pragma unroll
                    for (int q = 0; q < 8; ++q) dup |= ((uint32_t)q < k) && (g[q] == gid);
                    if (!dup) {
                        if (k < 8) {
#pragma unroll
                            for (int q = 0; q < 8; ++q) if ((uint32_t)q == k) g[q] = gid;
                            ++k;
                        } else { ovf = true; break; }
                    }
                }
                kcnt = k;
                if (ovf) {  // > 8 distinct genes: first-occurrence count, O(na^2), rare
                    kcnt = 0;
                    for (uint32_t j = 0; j < na; ++j) {
                        const uint32_t tj = refw(j);
                        if (tj >= ref_count) continue;
                        const uint32_t gj = t2g[tj];
                        if (gj >= num_genes) continue;
                        bool first = true;
                        for (uint32_t q = 0; q < j && first; ++q) {
                            const uint32_t tq = refw(q);
                            if (tq < ref_count && t2g[tq] == gj) first = false;
                        }
                        kcnt += first;
                        if (first && pug_gene) lhash += gene_set_hash_term(gj);
                    }
                }
                if (pug_gene) {  // gene-level PUG: order-independent hash of the read's gene set
                    if (!ovf) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) if ((uint32_t)q < k) lhash += gene_set_hash_term(g[q]);
                    }
                    lhash ^= (uint64_t)kcnt * kHashMul;
                    lhash = label_key(lhash, kcnt, g[0], g[1]);  // (kcnt <= 2 implies !ovf: g[0], g[1] are the read's genes)
                }
            }
            if (m.mode == kModeTrivial && kcnt != 1) kcnt = 0;  // multi-gene reads are discarded (pugutils.rs:870-891)
            if (pug_rec) kcnt = 1;
            uint32_t tot;
            const uint32_t ex = wave_excl_scan(kcnt, tot);
            uint32_t wbase = 0;
            if (tot) {
                if (lane == 0) wbase = atomicAdd(&cell_nkeys[cur_cell], tot);
                wbase = __builtin_amdgcn_readfirstlane(wbase);
                if (wbase + tot > ((PUG && mode_is_pug(m.mode)) ? m.nrec : m.n_ref)) { fail = true; kcnt = 0; }
            }
            if (pug_rec) {
                if (kcnt) {
                    const uint64_t slot = pug.rd_off[cur_cell] + wbase + ex;
                    pug.h[slot] = lhash; pug.u[slot] = umi; pug.o[slot] = i;
                }
            } else if (kcnt) {
                uint64_t* dst = keys0 + m.key_off + wbase + ex;
                if (!ovf) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) if ((uint32_t)q < k) dst[q] = (umi << kGeneBits) | g[q];
                } else {
                    uint32_t o = 0;
                    for (uint32_t j = 0; j < na; ++j) {
                        const uint32_t tj = refw(j);
                        if (tj >= ref_count) continue;
                        const uint32_t gj = t2g[tj];
                        if (gj >= num_genes) continue;
                        bool first = true;
                        for (uint32_t q = 0; q < j && first; ++q) {
                            const uint32_t tq = refw(q);
                            if (tq < ref_count && t2g[tq] == gj) first = false;
                        }
                        if (first) dst[o++] = (umi << kGeneBits) | gj;
                    }
                }
            }
        }
        // all lanes are done reading this slab's stage/list before the next iteration overwrites them
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (has_next) {
            if (!same_next) {
                flush_chk();
                load_cell(next_cell);
                issue_slab_loads((slab + 1 - sp0) * kSlabWords);
            } else if (!prefetched) issue_slab_loads(s0 + kSlabWords);
        }
    }
    flush_chk();
}

#ifdef AFQ_DECODE_TIMING
__device__ unsigned long long g_dtm[16];
#define DT_MARK(i) do { if (lane == 0 && (blockIdx.x & 255) == 0 && wv == 0) { unsigned long long t_ = clock64(); atomicAdd(&g_dtm[i], t_ - tprev_); atomicAdd(&g_dtm[8 + i], 1ull); tprev_ = t_; } } while (0)
extern "C" void afq_debug_dump_decode() {
    unsigned long long h[16];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(g_dtm), sizeof(h));
    fprintf(stderr, "[decode cycles/slab]");
    for (int i = 0; i < 8; ++i) if (h[8 + i]) fprintf(stderr, " p%d=%llu", i, h[i] / h[8 + i]);
    fprintf(stderr, " (n=%llu)\n", h[8]);
}
#else
#define DT_MARK(i) do {} while (0)
#endif

// ---------------------------------------------------------------------------
// k_decode_keys: the walk-free decode for batches without parsimony cells, with one lane per DWORD instead
// of one lane per record.  Every dword of a slab asks "which record am I in" - the last candidate start at
// or before it, found with a ballot mask and a count-leading-zeros, or the record carried in from before
// the slab - and, if it is one of that record's alignment words, gathers its gene and emits the key
// (umi << 20 | gene) unless an earlier alignment word of the same record already named that gene.  All the
// tid_to_gid gathers of a slab are independent and issued together (the per-record version chased them one
// alignment at a time), and the work per slab is a fixed, short instruction sequence: the duplicate test
// looks at the three preceding dwords' genes; records with more alignments than that, or that started in an
// earlier slab, take a compact slow loop that exists once in the code.  Candidate lanes also accumulate the
// same proof terms as k_decode_par (count, sizes, successor check).
// A record that starts before the wave's first slab is found by a cooperative backward scan (64 dwords per
// step); inside the wave it is carried from slab to slab.
template <int BW, int UW, bool TRIVIAL>
__global__ __launch_bounds__(256, 6) void k_decode_keys(const uint8_t* __restrict__ bytes,
                                                    const CellMeta* __restrict__ meta, uint32_t n_cells,
                                                    const uint32_t* __restrict__ slab_prefix,
                                                    const uint32_t* __restrict__ slab_cell,
                                                    const uint64_t* __restrict__ cell_bc, uint32_t n_slabs,
                                                    const uint32_t* __restrict__ t2g, uint32_t ref_count,
                                                    uint32_t num_genes, uint64_t* __restrict__ keys0,
                                                    uint32_t* __restrict__ cell_nkeys,
                                                    uint64_t* __restrict__ bc_out, CellChk* __restrict__ chk) {
    static_assert(BW % 4 == 0 && UW % 4 == 0, "aligned layouts only");
    constexpr uint32_t BWW = BW / 4, UWW = UW / 4, HW = 1 + BWW + UWW;
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    __shared__ uint32_t s_stage[4][kStage];
    __shared__ uint32_t s_gene[4][4 + kSlabWords];   // [4 pad] + gene of every alignment word of the slab (kNone elsewhere)
    __shared__ uint32_t s_first[4][kSlabWords];  // dword index of the first alignment word of the dword's record
    const uint32_t lane = lane_id();
    const uint32_t wv = threadIdx.x >> 6;
    uint32_t* stage = s_stage[wv];
    uint32_t* gene_l = s_gene[wv];
    uint32_t* first_l = s_first[wv];
    const uint32_t n_groups = (n_slabs + kSlabsPerWave - 1) / kSlabsPerWave;
    const uint32_t n_cols = min(n_groups, kDecodeCols);
    const uint32_t n_rows = (n_groups + n_cols - 1) / n_cols;
    const uint32_t wid = blockIdx.x * 4 + wv;
    const uint32_t grp = (wid % n_cols) * n_rows + wid / n_cols;
    if (wid >= n_cols * n_rows || grp >= n_groups) return;
    const uint32_t slab_a = grp * kSlabsPerWave;
    const uint32_t slab_b = min(n_slabs, slab_a + kSlabsPerWave);
    uint32_t my_cell = 0;
    if (lane < slab_b - slab_a) my_cell = slab_cell[slab_a + lane];

    uint32_t cur_cell = kNone;
    CellMeta m{};
    const uint32_t* __restrict__ W = nullptr;
    uint32_t nwords = 0, sp0 = 0, bc_lo = 0, bc_hi = 0;
    uint32_t acc_count = 0, acc_words = 0;
    bool fail = false;
    uint32_t R[5];
    uint32_t cin_s = kNone, cin_na = 0, cin_ulo = 0, cin_uhi = 0;  // the record covering the slab's first dword

    auto load_cell = [&](uint32_t cell) {
        cur_cell = cell;
        m = meta[cell];
        const uint64_t bc = cell_bc[cell];
        bc_lo = (uint32_t)bc; bc_hi = (uint32_t)(bc >> 32);
        W = reinterpret_cast<const uint32_t*>(bytes + m.chunk_off);
        nwords = m.nbytes >> 2;
        sp0 = slab_prefix[cell];
        cin_s = kNone;
    };
    auto flush_chk = [&]() {
        if (cur_cell == kNone) return;
        uint32_t ws = acc_words;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) ws += __shfl_xor(ws, d);
        const bool any_fail = __any(fail);
        if (lane == 0) {
            if (acc_count) atomicAdd(&chk[cur_cell].count, acc_count);
            if (ws) atomicAdd(&chk[cur_cell].words, ws);
            if (any_fail) atomicOr(&chk[cur_cell].fail, 1u);
        }
        acc_count = 0; acc_words = 0; fail = false;
    };
    auto issue_slab_loads = [&](uint32_t s0) {
        if (s0 + kStage <= nwords) {
#pragma unroll
            for (int r = 0; r < 5; ++r) R[r] = W[s0 + r * 64 + lane];
        } else {
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                const uint32_t i = s0 + r * 64 + lane;
                R[r] = i < nwords ? W[i] : 0u;
            }
        }
    };
    auto find_carry = [&](uint32_t s0) {  // last candidate start before dword s0
        cin_s = kNone;
        uint32_t back = min(s0, nwords), steps = 0;
        while (back > 2 && cin_s == kNone && steps < (1u << 16)) {
            const uint32_t lo = back >= 64 ? back - 64 : 0u;
            const uint32_t q = lo + lane;
            bool c = q < back && q >= 2 && q + HW <= nwords;
            if (c) { c = W[q + 1] == bc_lo; if (BWW == 2 && c) c = W[q + 2] == bc_hi; }
            const uint64_t mk = __ballot(c);
            if (mk) cin_s = lo + 63 - (uint32_t)__builtin_clzll(mk);
            back = lo;
            ++steps;
        }
        if (cin_s != kNone) {
            cin_na = W[cin_s];
            cin_ulo = W[cin_s + 1 + BWW];
            cin_uhi = UWW == 2 ? W[cin_s + 2 + BWW] : 0u;
        }
    };
    auto gene_at = [&](uint32_t q, uint32_t s0) -> uint32_t {  // gene of alignment word q (< nwords) of the current cell
        if (q >= s0 && q < s0 + kSlabWords) return gene_l[4 + q - s0];
        const uint32_t t = W[q] & 0x7FFFFFFFu;
        return t < ref_count ? t2g[t] : kNone;
    };

    load_cell(__builtin_amdgcn_readlane(my_cell, 0));
    {
        const uint32_t s0 = (slab_a - sp0) * kSlabWords;
        issue_slab_loads(s0);
        if (s0) find_carry(s0);
    }
    const uint64_t le_mask = lane == 63 ? ~0ull : ((2ull << lane) - 1);
#ifdef AFQ_DECODE_TIMING
    unsigned long long tprev_ = clock64();
#endif

    // The loop body is written as unconditional LDS reads + selects: the compiler turns `c ? lds[i] : x` into
    // exec-mask branches (and once even into flat loads), which tripled the instruction count of this kernel.
    for (uint32_t slab = slab_a; slab < slab_b; ++slab) {
        const uint32_t s0 = (slab - sp0) * kSlabWords;
        uint32_t own[4];
#ifdef AFQ_DECODE_TIMING
        if (R[0] == 0x12345677u && R[4] == 0x7654321u) fail = true;  // wait for the slab's loads
#endif
        DT_MARK(0);
#pragma unroll
        for (int r = 0; r < 5; ++r) stage[r * 64 + lane] = R[r];
#pragma unroll
        for (int r = 0; r < 4; ++r) own[r] = R[r];
        if (lane < 4) gene_l[lane] = kNone;  // pad in front of the slab's genes (the duplicate test looks back 3)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // the next slab's dwords are requested now; nothing below depends on them
        const bool has_next = slab + 1 < slab_b;
        const uint32_t next_cell = has_next ? __builtin_amdgcn_readlane(my_cell, (int)(slab + 1 - slab_a)) : cur_cell;
        const bool same_next = has_next && next_cell == cur_cell;
        if (same_next) issue_slab_loads(s0 + kSlabWords);

        uint64_t mk[4];
        const bool triv = TRIVIAL && m.mode == kModeTrivial;  // tiny cells of a trivial run are cr-like (quant.rs:794-938)
        const bool room = nwords >= 2 + HW;
        // dword i can start a record iff 2 <= i and i + HW <= nwords: one unsigned compare of i - 2
        const uint32_t cand_lim = room ? nwords - HW - 1 : 0u;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t il = r * 64 + lane, i = s0 + il;
            const uint32_t w1 = stage[il + 1];
            bool cand = w1 == bc_lo && (i - 2u) < cand_lim;
            if (BWW == 2) { const uint32_t w2 = stage[il + 2]; cand = cand && w2 == bc_hi; }
            mk[r] = __ballot(cand);
        }
        if (s0 == 0 && (!room || !(mk[0] & 4ull))) fail = true;  // (1) the first record starts right after the chunk header
        acc_count += (uint32_t)(__popcll(mk[0]) + __popcll(mk[1]) + __popcll(mk[2]) + __popcll(mk[3]));
        DT_MARK(1);

        // which record is each dword in; alignment words gather their gene
        uint32_t gid[4], ulo[4], uhi[4];
        uint32_t pos[4];               // index of the dword among its record's alignment words, kNone if it is not one
        uint32_t last_before = kNone;  // il of the last candidate in the windows before r (wave-uniform)
        const uint32_t cin_na_eff = cin_s != kNone ? cin_na : 0u;
        bool slow = false;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t il = r * 64 + lane, i = s0 + il;
            const uint64_t within = mk[r] & le_mask;
            const uint32_t sil_w = (uint32_t)(r * 64 + 63) - (uint32_t)__builtin_clzll(within | 1ull);
            const uint32_t sil = within ? sil_w : last_before;
            const bool in_stage = sil != kNone;
            const uint32_t sc = in_stage ? sil : 0u;
            const uint32_t l_na = stage[sc], l_u0 = stage[sc + 1 + BWW], l_u1 = UWW == 2 ? stage[sc + 2 + BWW] : 0u;
            const uint32_t na = in_stage ? l_na : cin_na_eff;
            ulo[r] = in_stage ? l_u0 : cin_ulo;
            uhi[r] = in_stage ? l_u1 : cin_uhi;
            const uint32_t fr = (in_stage ? s0 + sc : cin_s) + HW;
            const uint32_t p = i - fr;  // wraps for the header dwords of the record
            const uint32_t t = own[r] & 0x7FFFFFFFu;
            bool isref = i >= fr && p < na && i < nwords;
            if (isref && t >= ref_count) { fail = true; isref = false; }
            // straight-line gather (lanes that are not alignment words read entry 0): the four windows' loads stay in flight together
            gid[r] = t2g[isref ? t : 0u];
            pos[r] = isref ? p : kNone;
            first_l[il] = fr;
            // more alignments back than the fast duplicate test covers, or some of them in an earlier slab
            slow = slow || (isref && p > 0 && (p > 3 || fr < s0)) || (triv && isref && p == 0 && na > 1);
            if (mk[r]) last_before = (uint32_t)(r * 64 + 63) - (uint32_t)__builtin_clzll(mk[r]);
        }
        // (2) proof terms of the records that start here
        auto mk_at = [&](uint32_t r) -> uint64_t { return r == 0 ? mk[0] : r == 1 ? mk[1] : r == 2 ? mk[2] : mk[3]; };
        bool far_succ = false;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t il = r * 64 + lane, i = s0 + il;
            const bool st = (mk[r] >> lane) & 1ull;
            const uint32_t na_s = own[r];
            const bool fits = na_s <= nwords && i + HW + na_s <= nwords;
            const uint32_t succ = i + HW + na_s, sl = il + HW + na_s;
            const bool in_lds = fits && sl + BWW < kStage;
            const uint32_t slc = in_lds ? sl : 0u;
            const uint32_t w1 = stage[slc + 1], w2 = BWW == 2 ? stage[slc + 2] : 0u;
            const bool at_end = succ == nwords;
            const bool succ_ok = succ + HW <= nwords && w1 == bc_lo && (BWW == 1 || w2 == bc_hi);
            if (st && (!fits || (!at_end && in_lds && !succ_ok))) fail = true;
            far_succ = far_succ || (st && fits && !at_end && !in_lds);
            acc_words += (st && fits) ? HW + na_s : 0u;
            if (UWW == 2) { const uint32_t uh = stage[il + 2 + BWW]; if (st && (uh >> (kUmiBits - 32))) fail = true; }
        }
        if (__any(far_succ)) {  // a record reaching past the staged halo: its successor is checked in global memory
#pragma unroll 1
            for (uint32_t r = 0; r < 4; ++r) {
                const uint32_t il = r * 64 + lane, i = s0 + il;
                if (!((mk_at(r) >> lane) & 1ull)) continue;
                const uint32_t na_s = stage[il];
                if (na_s > nwords || i + HW + na_s > nwords) continue;
                const uint32_t succ = i + HW + na_s, sl = il + HW + na_s;
                if (succ == nwords || sl + BWW < kStage) continue;
                bool ok = succ + HW <= nwords;
                if (ok) { ok = W[succ + 1] == bc_lo; if (BWW == 2 && ok) ok = W[succ + 2] == bc_hi; }
                if (!ok) fail = true;
            }
        }
        if (s0 == 0 && lane == 2 && ((mk[0] >> 2) & 1ull)) bc_out[cur_cell] = BWW == 2 ? ((uint64_t)bc_hi << 32 | bc_lo) : (uint64_t)bc_lo;
        DT_MARK(2);
        // the record the next slab starts in: this slab's last candidate, else the one carried in
        uint32_t ncin_s = cin_s, ncin_na = cin_na, ncin_ulo = cin_ulo, ncin_uhi = cin_uhi;
        if (last_before != kNone) {
            ncin_s = s0 + last_before; ncin_na = stage[last_before]; ncin_ulo = stage[last_before + 1 + BWW];
            ncin_uhi = UWW == 2 ? stage[last_before + 2 + BWW] : 0u;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool isref = pos[r] != kNone;
            if (isref && gid[r] >= num_genes) fail = true;
            gid[r] = (isref && gid[r] < num_genes) ? gid[r] : kNone;
            gene_l[4 + r * 64 + lane] = gid[r];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        DT_MARK(3);
        if (__any(slow)) {
            // one copy of the general rule; a dword that loses clears its gene (a duplicate's own first
            // occurrence stays, so clearing never hides a gene from a later dword of the record)
#pragma unroll 1
            for (uint32_t r = 0; r < 4; ++r) {
                const uint32_t il = r * 64 + lane, i = s0 + il;
                const uint32_t g = gene_l[4 + il], fr = first_l[il];
                if (g == kNone || i < fr) continue;
                const uint32_t p = i - fr;
                bool lose = false;
                if (triv) {  // only reads whose alignments name one gene count (pugutils.rs:870-891)
                    if (p > 0) continue;  // handled by the fast rule below (never emits)
                    const uint32_t S = fr - HW;
                    const uint32_t na = S >= s0 ? stage[S - s0] : W[S];
                    for (uint32_t q = fr + 1; q < fr + na && q < nwords && !lose; ++q) lose = gene_at(q, s0) != g;
                    if (lose) gene_l[4 + il] = kNone - 1;  // "not a single-gene read", still a gene for nobody else
                } else {
                    if (!(p > 3 || (p > 0 && fr < s0))) continue;
                    for (uint32_t q = fr; q < i && !lose; ++q) lose = gene_at(q, s0) == g;
                    if (lose) gene_l[4 + il] = kNone;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        // first occurrence of the gene inside its record
        uint64_t bal[4];
        uint32_t tot = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t il = r * 64 + lane;
            const uint32_t g0 = gene_l[4 + il], g1 = gene_l[3 + il], g2 = gene_l[2 + il], g3 = gene_l[1 + il];
            const uint32_t p = pos[r];
            bool e = gid[r] != kNone;
            if (triv) e = e && p == 0 && g0 == gid[r];
            else {
                const bool deep = p > 3 || p > il;  // the slow loop decided (p > il: the record started before the slab)
                const bool dup = (p >= 1 && g1 == gid[r]) || (p >= 2 && g2 == gid[r]) || (p >= 3 && g3 == gid[r]);
                e = e && !(deep ? g0 == kNone : dup);
            }
            bal[r] = __ballot(e);
            gid[r] = e ? gid[r] : kNone;
            tot += (uint32_t)__popcll(bal[r]);
        }
        DT_MARK(4);
        if (tot) {
            uint32_t wbase = 0;
            if (lane == 0) wbase = atomicAdd(&cell_nkeys[cur_cell], tot);
            wbase = __builtin_amdgcn_readfirstlane(wbase);
            DT_MARK(5);
            if (wbase + tot > m.n_ref) fail = true;
            else {
                uint64_t* dst = keys0 + m.key_off + wbase;
                uint32_t o = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (gid[r] != kNone) {
                        const uint64_t umi = UWW == 2 ? ((uint64_t)uhi[r] << 32 | ulo[r]) : (uint64_t)ulo[r];
                        const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal[r] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal[r], 0u));
                        dst[o + before] = (umi << kGeneBits) | gid[r];
                    }
                    o += (uint32_t)__popcll(bal[r]);
                }
            }
        }
        DT_MARK(6);
        cin_s = ncin_s; cin_na = ncin_na; cin_ulo = ncin_ulo; cin_uhi = ncin_uhi;
        // all lanes are done reading this slab's stage before the next iteration overwrites it
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (has_next && !same_next) {
            flush_chk();
            load_cell(next_cell);
            issue_slab_loads((slab + 1 - sp0) * kSlabWords);
        }
    }
    flush_chk();
}

// ---------------------------------------------------------------------------
// k_decode_recs: walk-free decode with one lane per RECORD, for inputs whose records carry few alignments
// (the planner picks it when the batch averages < 2 alignment words per record; k_decode_keys - one lane per
// dword - is the one that stays flat as records get longer).  The candidates of a slab are compacted into a
// list, then one lane per candidate reads na, the UMI and up to kInl alignment words out of LDS, issues all
// its tid_to_gid gathers together, drops repeated genes with a handful of compares and stores its keys.
// Output positions come from ballots (all first keys, then all second keys, ...), so consecutive lanes write
// consecutive slots.  Records with more alignments, or reaching past the staged halo, go through a serial
// per-lane loop that exists once in the code.  Same proof terms as the other two decoders.
constexpr uint32_t kInl = 3;
template <int BW, int UW, bool TRIVIAL>
__global__ __launch_bounds__(256, 8) void k_decode_recs(const uint8_t* __restrict__ bytes,
                                                    const CellMeta* __restrict__ meta, uint32_t n_cells,
                                                    const uint32_t* __restrict__ slab_prefix,
                                                    const uint32_t* __restrict__ slab_cell,
                                                    const uint64_t* __restrict__ cell_bc, uint32_t n_slabs,
                                                    const uint32_t* __restrict__ t2g, uint32_t ref_count,
                                                    uint32_t num_genes, uint64_t* __restrict__ keys0,
                                                    uint32_t* __restrict__ cell_nkeys,
                                                    uint64_t* __restrict__ bc_out, CellChk* __restrict__ chk) {
    static_assert(BW % 4 == 0 && UW % 4 == 0, "aligned layouts only");
    constexpr uint32_t BWW = BW / 4, UWW = UW / 4, HW = 1 + BWW + UWW;
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    __shared__ uint32_t s_stage[4][kStage];
    __shared__ uint32_t s_list[4][kSlabWords];
    const uint32_t lane = lane_id();
    const uint32_t wv = threadIdx.x >> 6;
    uint32_t* stage = s_stage[wv];
    uint32_t* list = s_list[wv];
    const uint32_t n_groups = (n_slabs + kSlabsPerWave - 1) / kSlabsPerWave;
    const uint32_t n_cols = min(n_groups, kDecodeCols);
    const uint32_t n_rows = (n_groups + n_cols - 1) / n_cols;
    const uint32_t wid = blockIdx.x * 4 + wv;
    const uint32_t grp = (wid % n_cols) * n_rows + wid / n_cols;
    if (wid >= n_cols * n_rows || grp >= n_groups) return;
    const uint32_t slab_a = grp * kSlabsPerWave;
    const uint32_t slab_b = min(n_slabs, slab_a + kSlabsPerWave);
    uint32_t my_cell = 0;
    if (lane < slab_b - slab_a) my_cell = slab_cell[slab_a + lane];

    uint32_t cur_cell = kNone;
    CellMeta m{};
    const uint32_t* __restrict__ W = nullptr;
    uint32_t nwords = 0, sp0 = 0, bc_lo = 0, bc_hi = 0;
    uint32_t acc_count = 0, acc_words = 0;
    bool fail = false;
    uint32_t R[5];

    auto load_cell = [&](uint32_t cell) {
        cur_cell = cell;
        m = meta[cell];
        const uint64_t bc = cell_bc[cell];
        bc_lo = (uint32_t)bc; bc_hi = (uint32_t)(bc >> 32);
        W = reinterpret_cast<const uint32_t*>(bytes + m.chunk_off);
        nwords = m.nbytes >> 2;
        sp0 = slab_prefix[cell];
    };
    auto flush_chk = [&]() {
        if (cur_cell == kNone) return;
        uint32_t ws = acc_words;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) ws += __shfl_xor(ws, d);
        const bool any_fail = __any(fail);
        if (lane == 0) {
            if (acc_count) atomicAdd(&chk[cur_cell].count, acc_count);
            if (ws) atomicAdd(&chk[cur_cell].words, ws);
            if (any_fail) atomicOr(&chk[cur_cell].fail, 1u);
        }
        acc_count = 0; acc_words = 0; fail = false;
    };
    auto issue_slab_loads = [&](uint32_t s0) {
        if (s0 + kStage <= nwords) {
#pragma unroll
            for (int r = 0; r < 5; ++r) R[r] = W[s0 + r * 64 + lane];
        } else {
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                const uint32_t i = s0 + r * 64 + lane;
                R[r] = i < nwords ? W[i] : 0u;
            }
        }
    };

    load_cell(__builtin_amdgcn_readlane(my_cell, 0));
    issue_slab_loads((slab_a - sp0) * kSlabWords);
#ifdef AFQ_DECODE_TIMING
    unsigned long long tprev_ = clock64();
#endif

    for (uint32_t slab = slab_a; slab < slab_b; ++slab) {
        const uint32_t s0 = (slab - sp0) * kSlabWords;
#ifdef AFQ_DECODE_TIMING
        if (R[0] == 0x12345677u && R[4] == 0x7654321u) fail = true;  // wait for the slab's loads
#endif
        DT_MARK(0);
#pragma unroll
        for (int r = 0; r < 5; ++r) stage[r * 64 + lane] = R[r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const bool has_next = slab + 1 < slab_b;
        const uint32_t next_cell = has_next ? __builtin_amdgcn_readlane(my_cell, (int)(slab + 1 - slab_a)) : cur_cell;
        const bool same_next = has_next && next_cell == cur_cell;
        if (same_next) issue_slab_loads(s0 + kSlabWords);

        const bool triv = TRIVIAL && m.mode == kModeTrivial;  // tiny cells of a trivial run are cr-like (quant.rs:794-938)
        const bool room = nwords >= 2 + HW;
        const uint32_t cand_lim = room ? nwords - HW - 1 : 0u;  // dword i can start a record iff (i - 2) < cand_lim
        uint32_t ncand = 0;
        uint64_t mk0 = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t il = r * 64 + lane, i = s0 + il;
            const uint32_t w1 = stage[il + 1];
            bool cand = w1 == bc_lo && (i - 2u) < cand_lim;
            if (BWW == 2) { const uint32_t w2 = stage[il + 2]; cand = cand && w2 == bc_hi; }
            const uint64_t mk = __ballot(cand);
            if (r == 0) mk0 = mk;
            const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
            if (cand) list[ncand + before] = il;
            ncand += (uint32_t)__popcll(mk);
        }
        if (s0 == 0 && (!room || !(mk0 & 4ull))) fail = true;  // (1) the first record starts right after the chunk header
        if (s0 == 0 && lane == 2 && ((mk0 >> 2) & 1ull)) bc_out[cur_cell] = BWW == 2 ? ((uint64_t)bc_hi << 32 | bc_lo) : (uint64_t)bc_lo;
        acc_count += ncand;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        DT_MARK(1);

        for (uint32_t base = 0; base < ncand; base += 64) {
            const uint32_t c = base + lane;
            const bool act = c < ncand;
            const uint32_t il = list[act ? c : 0u], i = s0 + il;
            const uint32_t na = stage[il], u0 = stage[il + 1 + BWW], u1 = UWW == 2 ? stage[il + 2 + BWW] : 0u;
            const bool fits = act && na <= nwords && i + HW + na <= nwords;
            // (2) the next record starts where this one ends
            const uint32_t succ = i + HW + na, sl = il + HW + na;
            const bool in_lds = fits && sl + BWW < kStage;
            const uint32_t slc = in_lds ? sl : 0u;
            const uint32_t w1 = stage[slc + 1], w2 = BWW == 2 ? stage[slc + 2] : 0u;
            const bool at_end = succ == nwords;
            const bool succ_ok = succ + HW <= nwords && w1 == bc_lo && (BWW == 1 || w2 == bc_hi);
            if (act && (!fits || (!at_end && in_lds && !succ_ok))) fail = true;
            if (fits && !at_end && !in_lds) {  // the record reaches past the staged halo (rare)
                bool ok = succ + HW <= nwords;
                if (ok) { ok = W[succ + 1] == bc_lo; if (BWW == 2 && ok) ok = W[succ + 2] == bc_hi; }
                if (!ok) fail = true;
            }
            acc_words += fits ? HW + na : 0u;
            if (UWW == 2 && fits && (u1 >> (kUmiBits - 32))) fail = true;
            const uint64_t umi = UWW == 2 ? ((uint64_t)u1 << 32 | u0) : (uint64_t)u0;
            // alignments: up to kInl inline, all gathers in flight together
            const uint32_t na_eff = fits ? na : 0u;
            const bool slowrec = na_eff > kInl || il + HW + kInl > kStage;
            uint32_t t[kInl], g[kInl];
            bool v[kInl];
#pragma unroll
            for (uint32_t j = 0; j < kInl; ++j) {
                const uint32_t pj = il + HW + j;
                t[j] = stage[pj < kStage ? pj : kStage - 1] & 0x7FFFFFFFu;
                v[j] = !slowrec && j < na_eff;
                if (v[j] && t[j] >= ref_count) { fail = true; v[j] = false; }
                g[j] = t2g[v[j] ? t[j] : 0u];
            }
#pragma unroll
            for (uint32_t j = 0; j < kInl; ++j) {
                if (v[j] && g[j] >= num_genes) { fail = true; v[j] = false; }
#pragma unroll
                for (uint32_t q = 0; q < j; ++q) v[j] = v[j] && !(v[q] && g[q] == g[j]);
            }
            if (triv) {  // only reads whose alignments name one gene count (pugutils.rs:870-891)
                bool multi = false;
#pragma unroll
                for (uint32_t j = 1; j < kInl; ++j) { multi = multi || v[j]; v[j] = false; }
                v[0] = v[0] && !multi;
            }
            uint64_t bal[kInl];
            uint32_t tot = 0;
#pragma unroll
            for (uint32_t j = 0; j < kInl; ++j) { bal[j] = __ballot(v[j]); tot += (uint32_t)__popcll(bal[j]); }
            // long records: serial count now, serial emission after the reservation
            uint32_t scnt = 0, sex = 0;
            auto ref_at = [&](uint32_t j) -> uint32_t {
                const uint32_t pj = il + HW + j;
                return (pj < kStage ? stage[pj] : W[i + HW + j]) & 0x7FFFFFFFu;
            };
            auto for_each_first_gene = [&](auto&& f) {  // distinct genes of the record in first-occurrence order
                for (uint32_t j = 0; j < na_eff; ++j) {
                    const uint32_t tj = ref_at(j);
                    if (tj >= ref_count) { fail = true; continue; }
                    const uint32_t gj = t2g[tj];
                    if (gj >= num_genes) { fail = true; continue; }
                    bool first = true;
                    for (uint32_t q = 0; q < j && first; ++q) {
                        const uint32_t tq = ref_at(q);
                        if (tq < ref_count && t2g[tq] == gj) first = false;
                    }
                    if (first) f(gj);
                }
            };
            const bool any_slow = __any(slowrec && na_eff > 0);
            if (any_slow) {
                if (slowrec) for_each_first_gene([&](uint32_t) { ++scnt; });
                if (triv) scnt = scnt == 1 ? 1u : 0u;
                uint32_t stot;
                sex = tot + wave_excl_scan(scnt, stot);
                tot += stot;
            }
            DT_MARK(2);
            if (tot) {
                uint32_t wbase = 0;
                if (lane == 0) wbase = atomicAdd(&cell_nkeys[cur_cell], tot);
                wbase = __builtin_amdgcn_readfirstlane(wbase);
                DT_MARK(3);
                if (wbase + tot > m.n_ref) fail = true;
                else {
                    uint64_t* dst = keys0 + m.key_off + wbase;
                    uint32_t o = 0;
#pragma unroll
                    for (uint32_t j = 0; j < kInl; ++j) {
                        const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal[j] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal[j], 0u));
                        if (v[j]) dst[o + before] = (umi << kGeneBits) | g[j];
                        o += (uint32_t)__popcll(bal[j]);
                    }
                    if (any_slow && slowrec && scnt) {
                        uint32_t w = sex;
                        for_each_first_gene([&](uint32_t gj) { dst[w++] = (umi << kGeneBits) | gj; });
                    }
                }
            }
            DT_MARK(4);
        }
        // all lanes are done reading this slab's stage/list before the next iteration overwrites them
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (has_next && !same_next) {
            flush_chk();
            load_cell(next_cell);
            issue_slab_loads((slab + 1 - sp0) * kSlabWords);
        }
    }
    flush_chk();
}

// ---------------------------------------------------------------------------
// Bucket histogram.  Device-scope atomics leave the XCD (every one is a fabric
// transaction on this 8-XCD part: rocprof WRITE_SIZE showed 3-5x the payload when
// they were issued per record), so counts are first combined in LDS over a tile
// of kTileKeys keys and flushed with one atomic per non-empty bucket per tile.
constexpr uint32_t kTileKeys = kScatterTileHost;  // 2048
constexpr uint32_t kLdsBins = 2048;               // buckets per cell the LDS paths can hold

__device__ __forceinline__ void tile_to_cell(const uint32_t* __restrict__ tile_prefix, uint32_t n_multi,
                                             uint32_t tile, uint32_t* s_bcast, uint32_t& ci, uint32_t& local_tile) {
    if (threadIdx.x == 0) {
        uint32_t lo = 0, hi = n_multi;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (tile_prefix[mid] <= tile) lo = mid; else hi = mid;
        }
        s_bcast[0] = lo;
    }
    __syncthreads();
    ci = s_bcast[0];
    local_tile = tile - tile_prefix[ci];
}

__global__ __launch_bounds__(256) void k_hist(const uint32_t* __restrict__ multi_cells,
                                             const uint32_t* __restrict__ tile_prefix, uint32_t n_multi,
                                             const CellMeta* __restrict__ meta,
                                             const uint32_t* __restrict__ cell_nkeys,
                                             const uint64_t* __restrict__ keys0, uint32_t* __restrict__ bucket_cnt) {
    __shared__ uint32_t s_hist[kLdsBins];
    __shared__ uint32_t s_b[1];
    uint32_t ci, lt;
    tile_to_cell(tile_prefix, n_multi, blockIdx.x, s_b, ci, lt);
    const uint32_t cell = multi_cells[ci];
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
__global__ __launch_bounds__(256) void k_scatter(const uint32_t* __restrict__ multi_cells,
                                                const uint32_t* __restrict__ tile_prefix, uint32_t n_multi,
                                                const CellMeta* __restrict__ meta,
                                                const uint32_t* __restrict__ cell_nkeys,
                                                const uint64_t* __restrict__ keys0, uint64_t* __restrict__ keys1,
                                                uint32_t* __restrict__ cursor) {
    constexpr uint32_t E = kTileKeys / 256;
    __shared__ uint64_t s_keys[kTileKeys];
    __shared__ uint32_t s_cnt[kLdsBins];   // per-bucket count, then tile-local exclusive offset
    __shared__ uint32_t s_base[kLdsBins];  // global position of the tile's first key of the bucket
    __shared__ uint32_t s_ws[4];
    __shared__ uint32_t s_b[1];
    uint32_t ci, lt;
    tile_to_cell(tile_prefix, n_multi, blockIdx.x, s_b, ci, lt);
    const uint32_t cell = multi_cells[ci];
    const CellMeta m = meta[cell];
    const uint32_t nk = mode_is_pug(m.mode) ? 0u : cell_nkeys[cell];  // PUG cells emit reads, not keys
    const uint32_t t0 = lt * kTileKeys;
    if (t0 >= nk) return;
    const uint32_t t1 = min(nk, t0 + kTileKeys);
    const uint64_t* src = keys0 + m.key_off;
    uint64_t* dst = keys1 + m.key_off;
    uint32_t* gcur = cursor + m.bucket_base;
    const uint32_t nb = 1u << m.lg_nb;
    if (nb > kLdsBins) {  // giant cell: per-key global atomics
        for (uint32_t i = t0 + threadIdx.x; i < t1; i += 256) {
            const uint64_t key = src[i];
            dst[atomicAdd(&gcur[bucket_of(key >> kGeneBits, m.lg_nb)], 1u)] = key;
        }
        return;
    }
    for (uint32_t b = threadIdx.x; b < nb; b += 256) s_cnt[b] = 0;
    __syncthreads();
    uint64_t key[E];
    uint32_t rank[E];
#pragma unroll
    for (uint32_t e = 0; e < E; ++e) {
        const uint32_t i = t0 + e * 256 + threadIdx.x;
        key[e] = kKeySentinel;
        rank[e] = 0;
        if (i < t1) {
            key[e] = src[i];
            rank[e] = atomicAdd(&s_cnt[bucket_of(key[e] >> kGeneBits, m.lg_nb)], 1u);
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
            s_base[b] = c ? atomicAdd(&gcur[b], c) : 0u;
        }
        carry += tot;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t e = 0; e < E; ++e) {
        const uint32_t i = t0 + e * 256 + threadIdx.x;
        if (i < t1) s_keys[s_cnt[bucket_of(key[e] >> kGeneBits, m.lg_nb)] + rank[e]] = key[e];
    }
    __syncthreads();
    const uint32_t nt = t1 - t0;
    for (uint32_t i = threadIdx.x; i < nt; i += 256) {
        const uint64_t kx = s_keys[i];
        const uint32_t b = bucket_of(kx >> kGeneBits, m.lg_nb);
        dst[s_base[b] + (i - s_cnt[b])] = kx;
    }
}

// ---------------------------------------------------------------------------
// Workgroup sort with the data in registers.  Thread t of wave w holds E
// elements; element h of lane l is position w*64*E + h*64 + l of the block of
// N = NT*E elements.  A bitonic compare-exchange at distance j is a cross-lane
// shuffle (j < 64), a register swap inside the thread (64 <= j < 64*E) or, only
// for j >= 64*E, a trip through LDS with barriers - 3 such stages for N <= 2048
// instead of one barrier per stage (45-66) with the data in LDS.
// Cross-lane exchange lane <-> lane^J on the VALU instead of ds_bpermute (which occupies the LDS pipe, the
// bottleneck of the bitonic sort): DPP quad permutes for J = 1, 2; two bank-masked DPP row shifts for J = 4, 8;
// gfx950's v_permlane16_swap / v_permlane32_swap for J = 16, 32 (semantics checked on hardware, scratch/dpp_probe.hip).
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
template <int J>
__device__ __forceinline__ uint32_t xor_lane32(uint32_t x) {
    if constexpr (J == 1) return __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false);
    else if constexpr (J == 2) return __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false);
    else if constexpr (J == 4) {
        uint32_t t = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xF, 0x5, false);   // row_shl:4 into banks 0,2
        return __builtin_amdgcn_update_dpp(t, x, 0x114, 0xF, 0xA, false);          // row_shr:4 into banks 1,3
    } else if constexpr (J == 8) {
        uint32_t t = __builtin_amdgcn_update_dpp(x, x, 0x108, 0xF, 0x3, false);   // row_shl:8 into banks 0,1
        return __builtin_amdgcn_update_dpp(t, x, 0x118, 0xF, 0xC, false);          // row_shr:8 into banks 2,3
    } else if constexpr (J == 16) {
        const v2u_t p = __builtin_amdgcn_permlane16_swap(x, x, false, false);      // .x = rows {0,0,2,2}, .y = rows {1,1,3,3}
        return (lane_id() & 16u) ? p.x : p.y;
    } else {
        const v2u_t p = __builtin_amdgcn_permlane32_swap(x, x, false, false);      // .x = lower half twice, .y = upper half twice
        return (lane_id() & 32u) ? p.x : p.y;
    }
}
template <int J, typename T>
__device__ __forceinline__ T xor_lane(T v);
template <int J, typename T>
__device__ __forceinline__ T xor_lane(T v) {
    if constexpr (sizeof(T) == 8) {
        const uint32_t lo = xor_lane32<J>((uint32_t)v), hi = xor_lane32<J>((uint32_t)((uint64_t)v >> 32));
        return (T)(((uint64_t)hi << 32) | lo);
    } else return (T)xor_lane32<J>((uint32_t)v);
}

template <int E, int J, typename T>
__device__ __forceinline__ void lane_stage(T (&a)[E], uint32_t idx0, uint32_t k) {
    const bool lower = (idx0 & J) == 0;  // J < 64: a property of the lane only
#pragma unroll
    for (int h = 0; h < E; ++h) {
        const bool want_min = lower == (((idx0 + h * 64) & k) == 0);
        const T o = xor_lane<J, T>(a[h]);
        // keep own value iff it is on the wanted side of the partner's: one compare, one select
        a[h] = ((a[h] < o) == want_min) ? a[h] : o;
    }
}

template <int E, int JH, typename T>
__device__ __forceinline__ void reg_stage(T (&a)[E], uint32_t idx0, uint32_t k) {
#pragma unroll
    for (int h = 0; h < E; ++h) {
        if ((h & JH) == 0 && (h | JH) < E) {
            const bool asc = ((idx0 + h * 64) & k) == 0;
            const T x = a[h], y = a[h | JH];
            const bool sw = asc ? (x > y) : (x < y);
            a[h] = sw ? y : x;
            a[h | JH] = sw ? x : y;
        }
    }
}

template <int NT, int E, typename T>
__device__ __forceinline__ void reg_bitonic_sort(T (&a)[E], T* s_x) {
    constexpr uint32_t N = NT * E;
    const uint32_t lane = lane_id();
    const uint32_t wbase = (threadIdx.x >> 6) * 64 * E;
    for (uint32_t k = 2; k <= N; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            if (j < 64) {
                switch (j) {
                    case 1: lane_stage<E, 1, T>(a, wbase + lane, k); break;
                    case 2: lane_stage<E, 2, T>(a, wbase + lane, k); break;
                    case 4: lane_stage<E, 4, T>(a, wbase + lane, k); break;
                    case 8: lane_stage<E, 8, T>(a, wbase + lane, k); break;
                    case 16: lane_stage<E, 16, T>(a, wbase + lane, k); break;
                    default: lane_stage<E, 32, T>(a, wbase + lane, k); break;
                }
            } else if (j < 64 * E) {
                // partner is another register of the same thread; keep the indices compile-time
                if (E >= 2 && j == 64) reg_stage<E, 1, T>(a, wbase + lane, k);
                else if (E >= 4 && j == 128) reg_stage<E, 2, T>(a, wbase + lane, k);
                else if (E >= 8 && j == 256) reg_stage<E, 4, T>(a, wbase + lane, k);
            } else {
                __syncthreads();
#pragma unroll
                for (int h = 0; h < E; ++h) s_x[wbase + h * 64 + lane] = a[h];
                __syncthreads();
#pragma unroll
                for (int h = 0; h < E; ++h) {
                    const uint32_t idx = wbase + h * 64 + lane;
                    const T o = s_x[idx ^ j];
                    const bool want_min = (((idx & j) == 0) == ((idx & k) == 0));
                    const T mn = a[h] < o ? a[h] : o, mx = a[h] < o ? o : a[h];
                    a[h] = want_min ? mn : mx;
                }
            }
        }
    }
}

// load n (<= NT*E) elements, sort ascending, leave them sorted in s_x[0..n)
template <int NT, int E, typename T>
__device__ __forceinline__ void block_sort_to_lds(const T* src, uint32_t n, T* s_x, T sentinel) {
    T a[E];
    const uint32_t wbase = (threadIdx.x >> 6) * 64 * E;
#pragma unroll
    for (int h = 0; h < E; ++h) {
        const uint32_t idx = wbase + h * 64 + lane_id();
        a[h] = idx < n ? src[idx] : sentinel;
    }
    reg_bitonic_sort<NT, E, T>(a, s_x);
    __syncthreads();
#pragma unroll
    for (int h = 0; h < E; ++h) s_x[wbase + h * 64 + lane_id()] = a[h];
    __syncthreads();
}

template <int NT, typename T>
__device__ __forceinline__ void block_sort_any(const T* src, uint32_t n, T* s_x, T sentinel) {
    if (n <= NT) block_sort_to_lds<NT, 1, T>(src, n, s_x, sentinel);
    else if (n <= 2 * NT) block_sort_to_lds<NT, 2, T>(src, n, s_x, sentinel);
    else if (n <= 4 * NT) block_sort_to_lds<NT, 4, T>(src, n, s_x, sentinel);
    else block_sort_to_lds<NT, 8, T>(src, n, s_x, sentinel);
}

// ---------------------------------------------------------------------------
// resolve core, shared by the LDS and the global-scratch variants.
struct ResolveCfg {
    uint32_t usa, num_rows, uo, ao;
    uint32_t mode;  // filled per bucket from its descriptor
};
__device__ __forceinline__ bool mode_is_em(uint32_t mode) { return mode == kModeCrLikeEm; }

__device__ __forceinline__ bool is_spliced(uint32_t g) { return (g & 1u) == 0; }
__device__ __forceinline__ bool same_gene(uint32_t a, uint32_t b) { return (a & ~1u) == (b & ~1u); }

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
            const uint32_t c = run_end(q) - s;
            maxc = c > maxc ? c : maxc;
        }
        uint32_t nb = 0, g1 = 0, g2 = 0, nsp = 0, first_sp = 0;
        bool prev_first_sp = false, sp_followed = false;
        for (uint32_t q = r; q < nruns; ++q) {
            const uint32_t s = run_start[q];
            const uint64_t kq = keys[s];
            if ((kq >> kGeneBits) != umi) break;
            if (run_end(q) - s != maxc) continue;
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
                        if (run_end(q) - s == maxc) dst[w++] = (uint32_t)kq & kGeneMask;
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
                              uint32_t n_buckets, BucketDesc* __restrict__ desc) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_buckets) return;
    const uint32_t cell = bucket_cell[b];
    const CellMeta m = meta[cell];
    BucketDesc d;
    d.cell = cell; d.out_off = m.key_off; d.n_ref = m.n_ref;
    if (m.lg_nb == 0) { d.mode_single = m.mode | 0x100u; d.src_off = m.key_off; d.n = mode_is_pug(m.mode) ? 0u : cell_nkeys[cell]; }
    else {
        const uint32_t beg = (b == m.bucket_base) ? 0u : cursor[b - 1];
        d.mode_single = m.mode; d.src_off = m.key_off + beg; d.n = mode_is_pug(m.mode) ? 0u : cursor[b] - beg;
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
constexpr uint32_t kHtCap = 2 * kHtKeys;      // slots: load factor <= 0.5
constexpr uint32_t kHtPairs = 3;
constexpr uint32_t kNoCol = 0xFFFFFFFFu;
static_assert(kHtKeys < (1u << 12), "per-bucket read counts fit the 12-bit counter");

__device__ __forceinline__ uint32_t ht_slot(uint32_t umi, uint32_t mask) {
    uint32_t x = umi ^ (umi >> 15);
    x *= 0x85EBCA6Bu;
    return (x >> 16) & mask;
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
template <typename ForEach>
__device__ __forceinline__ uint32_t col_from_candidates(ForEach&& for_each, const ResolveCfg& rc) {
    uint32_t maxc = 0;
    for_each([&](uint32_t, uint32_t c) { maxc = c > maxc ? c : maxc; });
    uint32_t nb = 0, g1 = kNoCol, g2 = kNoCol, nsp = 0, first_sp = kNoCol;
    for_each([&](uint32_t g, uint32_t c) {
        if (c != maxc) return;
        ++nb;
        if (g < g1) { g2 = g1; g1 = g; } else if (g < g2) g2 = g;
        if (is_spliced(g)) { ++nsp; first_sp = g < first_sp ? g : first_sp; }
    });
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
    for_each([&](uint32_t g, uint32_t c) { if (c == maxc && g == first_sp + 1) followed = true; });
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
    for (uint32_t h = 0; h < E; ++h) key[h] = h * 64 + lane < n ? src[h * 64 + lane] : 0ull;
    uint32_t cap = 128;
    while (cap < 2 * n) cap <<= 1;
    const uint32_t mask = cap - 1;
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
                uint32_t slot = ht_slot(umi, mask);
                bool done = false;
                for (;;) {
                    const unsigned long long old = atomicCAS(&s_slot[slot], kEmpty64, mine);
                    if (old == kEmpty64) { own_slot[h] = slot; done = true; break; }
                    if ((uint32_t)(old >> 32) == umi) {
                        if ((((uint32_t)old) >> 12) == gene) { atomicAdd(&s_slot[slot], 1ull); done = true; }
                        break;
                    }
                    slot = (slot + 1) & mask;
                }
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
                    col = col_from_candidates([&](auto&& f) {
#pragma unroll
                        for (uint32_t q = 0; q < kHtMerge; ++q) if (cc[q]) f(cg[q], cc[q]);
                    }, rc);
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

// One wave per bucket (up to kBucketCap keys).  Small workgroups keep many buckets
// in flight per CU, which is what hides the load -> group -> reserve -> store latency
// chain; blocks that run together are spread over different cells (column-major walk)
// so their reservations do not pile onto one counter.
// Single-bucket cells are finished here; buckets of multi-bucket cells append their
// resolved columns to the cell's column list, counted later by k_cell_hist.
constexpr int kResolveNT = 64;
constexpr uint32_t kResolveCols = 8192;
static_assert(kBucketCap <= kResolveNT * 8, "bucket cap <= 8 keys per thread");
template <bool EM>
__global__ __launch_bounds__(kResolveNT) void k_resolve(const BucketDesc* __restrict__ desc, uint32_t n_buckets,
                                                       uint64_t* __restrict__ keys0,
                                                       const uint64_t* __restrict__ keys1,
                                                       uint32_t* __restrict__ cell_ncols, uint32_t* __restrict__ nnz,
                                                       OverflowEnt* __restrict__ ovf_list, DevStatus* st,
                                                       ResolveCfg rc, LabArea la) {
    // one LDS block carved two ways: the hash table (slot UMIs | counters), or the sort path's arrays
    constexpr uint32_t kSortWords = 2 * kBucketCap + kBucketCap / 2 + kBucketCap + (EM ? 2 * kBucketCap : 0);
    constexpr uint32_t kHashWords = kHtCap * (1 + kHtPairs) + 2 * kHtOvf + kHtCap / 32 + 2 + kHtKeys + (EM ? 2 * kHtKeys : 0);
    constexpr uint32_t kWords = kSortWords > kHashWords ? kSortWords : kHashWords;
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
    if ((bmode == kModeCrLike || (EM && bmode == kModeCrLikeEm && la.lab)) && d.n <= kHtKeys) {
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

// Buckets over the 2-wave cap but within LDS reach (<= kMidCap keys): persistent
// 1024-thread workgroups loop over the overflow list.
constexpr int kMidNT = 1024;
constexpr uint32_t kMidCap = kMidNT * 8;
__global__ __launch_bounds__(kMidNT) void k_resolve_mid(const BucketDesc* __restrict__ desc,
                                                       uint64_t* __restrict__ keys0,
                                                       const uint64_t* __restrict__ keys1,
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
        if (ovf_list[e].n > kMidCap) continue;
        const BucketDesc d = desc[ovf_list[e].bucket];
        __syncthreads();
        resolve_bucket_lds<kMidNT>(d, keys0, keys1, cell_ncols, nnz, st, rc, la, s_keys, s_run, s_cols, s_lab, s_ldesc, s_ws, s_misc);
        __syncthreads();
    }
}

// Buckets beyond LDS reach (one UMI carried by thousands of reads, adversarial
// input): same algorithm with the bucket sorted in place in keys1 (normalised
// bitonic network, any n) and the run table in the upper half of the cell's keys0
// slots.  Persistent blocks loop over the overflow list.
constexpr int kBigNT = 1024;
__global__ __launch_bounds__(kBigNT) void k_resolve_big(const BucketDesc* __restrict__ desc,
                                                       const CellMeta* __restrict__ meta,
                                                       uint64_t* __restrict__ keys0, uint64_t* __restrict__ keys1,
                                                       uint32_t* __restrict__ cell_ncols,
                                                       const OverflowEnt* __restrict__ ovf_list, DevStatus* st,
                                                       ResolveCfg rc, LabArea la) {
    __shared__ uint32_t s_ws[kBigNT / 64];
    const uint32_t novf = st->n_overflow;
    for (uint32_t e = blockIdx.x; e < novf; e += gridDim.x) {
        if (ovf_list[e].n <= kMidCap) continue;
        const BucketDesc d = desc[ovf_list[e].bucket];
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
                                                      uint32_t* __restrict__ nnz, ResolveCfg rc) {
    __shared__ uint32_t s_hist[kHistBins];
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
        constexpr uint32_t kBins16 = 2 * kHistBins;
        for (uint32_t lo = 0; lo < rc.num_rows; lo += kBins16) {
            const uint32_t nbins = min(kBins16, rc.num_rows - lo), nwords = (nbins + 1) / 2;
            for (uint32_t i = threadIdx.x; i < nwords; i += kHistNT) s_hist[i] = 0;
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < nc; i += kHistNT) {
                const uint32_t c = cols[i] - lo;
                if (c < nbins) atomicAdd(&s_hist[c >> 1], 1u << (16 * (c & 1u)));
            }
            __syncthreads();
            for (uint32_t base = 0; base < nwords; base += kHistNT * 2) {
                const uint32_t q = base + threadIdx.x * 2;   // two words = four bins per thread
                uint32_t v[4];
                const uint32_t w0 = q < nwords ? s_hist[q] : 0u, w1 = q + 1 < nwords ? s_hist[q + 1] : 0u;
                v[0] = w0 & 0xFFFFu; v[1] = w0 >> 16; v[2] = w1 & 0xFFFFu; v[3] = w1 >> 16;
                const uint32_t c = (v[0] != 0) + (v[1] != 0) + (v[2] != 0) + (v[3] != 0);
                uint32_t tot;
                uint32_t o = carry + block_excl_scan<kHistNT>(c, s_ws, tot);
#pragma unroll
                for (int e = 0; e < 4; ++e) if (v[e]) out[o++] = make_uint2(lo + 2 * q + e, v[e]);
                carry += tot;
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) nnz[cell] = carry;
        return;
    }
    for (uint32_t lo = 0; lo < rc.num_rows; lo += kHistBins) {
        const uint32_t nbins = min(kHistBins, rc.num_rows - lo);
        for (uint32_t i = threadIdx.x; i < nbins; i += kHistNT) s_hist[i] = 0;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < nc; i += kHistNT) {
            const uint32_t c = cols[i] - lo;
            if (c < nbins) atomicAdd(&s_hist[c], 1u);
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
// Per-cell EM over the gene-level equivalence classes (src/em.rs).  One workgroup per cell.
//   inputs : the cell's single-label counts = sorted (column,count) pairs (k_cell_hist / k_resolve),
//            and the labels of its ambiguous molecules (label area, written by resolve)
//   steps  : group identical labels into classes (lexicographic order), rewrite USA labels to S/U/A
//            slots (utils.rs:865-925), build the support (labels, and in USA their sibling statuses,
//            em.rs:87-113), then iterate.  One round = (A) thread per class: denominator in label order;
//            (B) thread per support entry: the single-label count first, then the contributions of the
//            classes containing it in class order - the same f32 operation sequence as the sequential
//            loop of em_update (em.rs:189-248, 458-485) under the oracle's canonical class order, so
//            results are bit-identical to the oracle; (C) convergence vote.
//   schedule: non-USA = em_optimize (em.rs:536-572); USA = em_optimize_subset_impl with the extra
//            round after zeroing < 0.01 (em.rs:391-451).
// All arrays live in a per-cell global scratch slice (L2 resident); sizes are tiny next to the decode.
struct EmCfg {
    uint32_t usa, num_alphas, uo, ao, init_uniform;
};
constexpr int kEmNT = 256;
constexpr float kMinOutputAlpha = 0.01f, kAlphaCheckCutoff = 1e-2f, kRelDiffTol = 1e-2f;
constexpr uint32_t kMinIter = 2, kMaxIter = 100;


// Per-cell EM scratch (u32 words; mirrored by em_scratch_words).  Filled by k_em (setup), consumed by k_em_rounds.
struct EmScratch {
    uint2* out; uint64_t* inv_pairs; uint32_t *order, *cls_first, *cls_cnt, *cls_woff, *cls_w, *cls_sidx; float* inv;
    uint32_t *support, *sib1, *sib2, *ucnt; float *a_in, *a_out; uint32_t *slot_off, *aid; uint4 *ent, *lw3;
    uint32_t *act_col, *memb;
};
__device__ __forceinline__ EmScratch em_carve(uint32_t* scratch, uint64_t off, uint32_t nU, uint32_t W, uint32_t M, uint32_t capS) {
    EmScratch e;
    uint32_t* p = scratch + off;
    e.out = reinterpret_cast<uint2*>(p); p += 2 * (capS + 1);           // (column, f32 bits); 8-byte aligned by construction
    e.inv_pairs = reinterpret_cast<uint64_t*>(p); p += 2 * (W + 1);     // (support idx << 32 | class)
    e.order = p; p += M + 1;       // molecule indices sorted by label
    e.cls_first = p; p += M + 1;   // class -> position in `order` of its first molecule
    e.cls_cnt = p; p += M + 1;
    e.cls_woff = p; p += M + 2;    // class -> offset of its EM label in cls_w
    e.cls_w = p; p += W + 1;       // EM labels (slots)
    e.cls_sidx = p; p += W + 1;    // ... as support indices
    e.inv = reinterpret_cast<float*>(p); p += M + 1;
    e.support = p; p += capS + 1;
    e.sib1 = p; p += capS + 1;
    e.sib2 = p; p += capS + 1;
    e.ucnt = p; p += capS + 1;
    e.a_in = reinterpret_cast<float*>(p); p += capS + 2;
    e.a_out = reinterpret_cast<float*>(p); p += capS + 2;
    e.slot_off = p; p += capS + 2;
    e.aid = p; p += capS + 1;          // support idx -> active idx
    p += (4 - ((p - scratch) & 3)) & 3;        // 16-byte records below (slices start 16-byte aligned)
    e.ent = reinterpret_cast<uint4*>(p); p += 4 * (nU + W + 2);   // per active entry: count, sibling ids, first membership
    e.lw3 = reinterpret_cast<uint4*>(p); p += 4 * (W + 1);        // per label word: its entry and the entry's siblings
    e.act_col = p; p += nU + W + 2;
    e.memb = p; p += W + 1;            // class ids of the memberships, entry-major
    return e;
}

__global__ __launch_bounds__(kEmNT) void k_em(const CellMeta* __restrict__ meta, const uint32_t* __restrict__ nnz_unique,
                                             const uint64_t* __restrict__ keys0, const uint64_t* __restrict__ keys1,
                                             const uint32_t* __restrict__ lab, const uint32_t* __restrict__ lab_cnt,
                                             const uint64_t* __restrict__ em_off, uint32_t* __restrict__ scratch,
                                             uint32_t* __restrict__ out_nnz, uint4* __restrict__ em_hdr, const uint32_t* __restrict__ em_order, EmCfg cfg) {
    __shared__ uint32_t s_ws[kEmNT / 64];
    __shared__ __attribute__((aligned(16))) uint32_t s_tile[8192];  // 32 KiB sort tile
    const uint32_t cell = em_order[blockIdx.x];  // largest cells first (the host sorts: input order is arbitrary in real data)
#ifdef AFQ_EM_TIMING
    __shared__ unsigned long long tmark[12];
#define EM_MARK(i) do { __syncthreads(); if (threadIdx.x == 0 && (blockIdx.x % 1000) == 7) tmark[i] = wall_clock64(); } while (0)
#else
#define EM_MARK(i) do {} while (0)
#endif
    const CellMeta m = meta[cell];
    const uint32_t nU = nnz_unique[cell];
    const uint2* U = reinterpret_cast<const uint2*>(((m.lg_nb || mode_is_pug(m.mode)) ? keys1 : keys0) + m.key_off);
    const uint32_t W = lab_cnt[2 * cell], M = lab_cnt[2 * cell + 1];
    const uint32_t* lw = lab + 2 * m.key_off;
    const uint32_t* ld = lw + m.n_ref + 1;
    const uint32_t mult = cfg.usa ? 3u : 1u;
    const uint32_t capS = (nU + W) * mult;
    const EmScratch sc = em_carve(scratch, em_off[cell], nU, W, M, capS);
    uint2* out = sc.out; uint64_t* inv_pairs = sc.inv_pairs; uint32_t* order = sc.order; uint32_t* cls_first = sc.cls_first;
    uint32_t* cls_cnt = sc.cls_cnt; uint32_t* cls_woff = sc.cls_woff; uint32_t* cls_w = sc.cls_w; uint32_t* cls_sidx = sc.cls_sidx;
    uint32_t* support = sc.support; uint32_t* sib1 = sc.sib1; uint32_t* sib2 = sc.sib2; uint32_t* ucnt = sc.ucnt;
    uint32_t* slot_off = sc.slot_off; uint32_t* aid = sc.aid; uint4* ent = sc.ent; uint4* lw3 = sc.lw3;
    uint32_t* act_col = sc.act_col; uint32_t* memb = sc.memb;
    if (M == 0) {  // no multi-label class: the counts are the single-label counts (em.rs:339-341, 499-514)
        for (uint32_t i = threadIdx.x; i < nU; i += kEmNT) out[i] = make_uint2(U[i].x, __float_as_uint((float)U[i].y));
        if (threadIdx.x == 0) { out_nnz[cell] = nU; em_hdr[cell] = make_uint4(0u, 0u, 0u, 1u); }
        return;
    }
    auto lab_gt = [&](uint32_t a, uint32_t b) {  // lexicographic a > b on the gene-level labels
        const uint32_t oa = ld[2 * a], na = ld[2 * a + 1], ob = ld[2 * b], nb = ld[2 * b + 1];
        const uint32_t nm = na < nb ? na : nb;
        for (uint32_t i = 0; i < nm; ++i) {
            const uint32_t x = lw[oa + i], y = lw[ob + i];
            if (x != y) return x > y;
        }
        return na > nb;
    };
    auto lab_ne = [&](uint32_t a, uint32_t b) { return lab_gt(a, b) || lab_gt(b, a); };
    EM_MARK(0);
    // 1. classes = runs of equal labels in lexicographic order
    // The sort runs on 16-byte records out of LDS: a 63-bit key holding the label's first three genes (+1, a missing
    // gene is 0, so key order IS the lexicographic order with shorter labels first) and the molecule index; only labels
    // that tie on three genes and are longer than that fall back to the pointer-chasing comparison.
    struct LabKey { uint64_t key; uint32_t idx, len; };
    LabKey* lk = reinterpret_cast<LabKey*>(inv_pairs);  // 4 words per molecule; inv_pairs holds 2(W+1) >= 4M+2 words (every label has >= 2 genes)
    for (uint32_t i = threadIdx.x; i < M; i += kEmNT) {
        const uint32_t o = ld[2 * i], n = ld[2 * i + 1];
        uint64_t key = (uint64_t)(lw[o] + 1u) << 42;
        if (n > 1) key |= (uint64_t)(lw[o + 1] + 1u) << 21;
        if (n > 2) key |= (uint64_t)(lw[o + 2] + 1u);
        lk[i] = LabKey{key, i, n};
    }
    __syncthreads();
    auto lk_gt = [&](const LabKey& a, const LabKey& b) {
        if (a.key != b.key) return a.key > b.key;
        if (a.len <= 3 && b.len <= 3) return false;  // same three-or-fewer genes: the same label
        return lab_gt(a.idx, b.idx);
    };
    tiled_bitonic_sort_by<kEmNT, 2048>(lk, M, lk_gt, reinterpret_cast<LabKey*>(s_tile));
    for (uint32_t i = threadIdx.x; i < M; i += kEmNT) order[i] = lk[i].idx;
    __syncthreads();
    uint32_t K = 0;
    for (uint32_t base = 0; base < M; base += kEmNT) {
        const uint32_t i = base + threadIdx.x;
        bool head = i < M;
        if (head && i > 0) {
            const LabKey a = lk[i], b = lk[i - 1];
            head = a.key != b.key || ((a.len > 3 || b.len > 3) && lab_ne(a.idx, b.idx));
        }
        const uint32_t h = head;
        uint32_t tot;
        const uint32_t ex = block_excl_scan<kEmNT>(h, s_ws, tot);
        if (h) cls_first[K + ex] = i;
        K += tot;
    }
    __syncthreads();
    EM_MARK(1);
    // 2. EM label of each class: length, then contents
    auto em_label = [&](uint32_t c, uint32_t* dst) -> uint32_t {  // returns the length; writes when dst != null
        const uint32_t mol = order[cls_first[c]];
        const uint32_t o = ld[2 * mol], n = ld[2 * mol + 1];
        uint32_t w = 0;
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t gn = lw[o + i];
            uint32_t idx = gn;
            if (cfg.usa) {
                idx = gn >> 1;
                if (is_spliced(gn)) {
                    if (i + 1 < n && same_gene(gn, lw[o + i + 1])) { idx += cfg.ao; ++i; }
                } else idx += cfg.uo;
            }
            if (dst) dst[w] = idx;
            ++w;
        }
        return w;
    };
    uint32_t Wc = 0;
    for (uint32_t base = 0; base < K; base += kEmNT) {
        const uint32_t c = base + threadIdx.x;
        const uint32_t len = c < K ? em_label(c, nullptr) : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan<kEmNT>(len, s_ws, tot);
        if (c < K) {
            cls_woff[c] = Wc + ex;
            cls_cnt[c] = (c + 1 < K ? cls_first[c + 1] : M) - cls_first[c];
        }
        Wc += tot;
    }
    if (threadIdx.x == 0) cls_woff[K] = Wc;
    __syncthreads();
    for (uint32_t c = threadIdx.x; c < K; c += kEmNT) em_label(c, cls_w + cls_woff[c]);
    __syncthreads();
    EM_MARK(2);
    // 3. support = single-label columns + label slots (+ USA sibling statuses), sorted, distinct.
    // When one bit per output column fits the LDS tile next to its rank table (num_alphas <= 131072: every gene-level
    // matrix in practice), the support is a bitmap: mark, prefix-popcount, and "index of column x in the support" is
    // two LDS reads instead of a sort of 3(nU + W) values and a binary search per lookup.
    const uint32_t nwb = (cfg.num_alphas + 31) >> 5;
    const bool bm = 2 * nwb <= 8192;
    uint32_t* bm_bits = s_tile;
    uint32_t* bm_rank = s_tile + nwb;
    uint32_t S = 0;
    if (bm) {
        for (uint32_t i = threadIdx.x; i < nwb; i += kEmNT) bm_bits[i] = 0;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < nU + Wc; i += kEmNT) {
            const uint32_t x = i < nU ? U[i].x : cls_w[i - nU];
            atomicOr(&bm_bits[x >> 5], 1u << (x & 31));
            if (cfg.usa) {
                uint32_t s1, s2;
                if (x >= cfg.ao) { s1 = x - cfg.uo; s2 = x - cfg.ao; }
                else if (x >= cfg.uo) { s1 = x + cfg.uo; s2 = x - cfg.uo; }
                else { s1 = x + cfg.ao; s2 = x + cfg.uo; }
                atomicOr(&bm_bits[s1 >> 5], 1u << (s1 & 31));
                atomicOr(&bm_bits[s2 >> 5], 1u << (s2 & 31));
            }
        }
        __syncthreads();
        for (uint32_t base = 0; base < nwb; base += kEmNT) {
            const uint32_t w = base + threadIdx.x;
            const uint32_t c = w < nwb ? (uint32_t)__popc(bm_bits[w]) : 0u;
            uint32_t tot;
            const uint32_t ex = block_excl_scan<kEmNT>(c, s_ws, tot);
            if (w < nwb) bm_rank[w] = S + ex;
            S += tot;
        }
        __syncthreads();
        for (uint32_t w = threadIdx.x; w < nwb; w += kEmNT) {
            uint32_t b = bm_bits[w], o = bm_rank[w];
            for (; b; b &= b - 1) support[o++] = (w << 5) + (uint32_t)__builtin_ctz(b);
        }
        __syncthreads();
    } else {
        uint32_t nC = 0;
        {
            const uint32_t nsrc = nU + Wc;
            for (uint32_t i = threadIdx.x; i < nsrc; i += kEmNT) {
                const uint32_t x = i < nU ? U[i].x : cls_w[i - nU];
                support[i * mult] = x;
                if (cfg.usa) {
                    uint32_t s1, s2;
                    if (x >= cfg.ao) { s1 = x - cfg.uo; s2 = x - cfg.ao; }
                    else if (x >= cfg.uo) { s1 = x + cfg.uo; s2 = x - cfg.uo; }
                    else { s1 = x + cfg.ao; s2 = x + cfg.uo; }
                    support[i * mult + 1] = s1;
                    support[i * mult + 2] = s2;
                }
            }
            nC = nsrc * mult;
        }
        __syncthreads();
        tiled_bitonic_sort_by<kEmNT, 8192>(support, nC, [](uint32_t a, uint32_t b) { return a > b; }, s_tile);
        for (uint32_t base = 0; base < nC; base += kEmNT) {  // in-place unique: position S+ex <= i, so reads stay ahead of writes
            const uint32_t i = base + threadIdx.x;
            const uint32_t v = i < nC ? support[i] : 0u;
            const uint32_t h = (i < nC) && (i == 0 || v != support[i - 1]);
            uint32_t tot;
            const uint32_t ex = block_excl_scan<kEmNT>(h, s_ws, tot);
            __syncthreads();
            if (h) support[S + ex] = v;
            S += tot;
            __syncthreads();
        }
    }
    auto sup_index = [&](uint32_t x) -> uint32_t {  // position of column x in the support (x is in it)
        if (bm) return bm_rank[x >> 5] + (uint32_t)__popc(bm_bits[x >> 5] & ((1u << (x & 31)) - 1u));
        return lower_bound_u32(support, S, x);
    };
    // NOTE on the USA support: the reference marks, for a label x, x and its siblings so that reads of
    // get_abundance_for are reset every round (em.rs:351-356).  Marking both siblings for every status is a
    // superset of em.rs:101-109 (which marks exactly the statuses get_abundance_for reads); the extra entries
    // hold 0 throughout and never change a sum.
    for (uint32_t s = threadIdx.x; s < S; s += kEmNT) {
        ucnt[s] = 0;
        sib1[s] = 0xFFFFFFFFu; sib2[s] = 0xFFFFFFFFu;
        if (cfg.usa) {
            const uint32_t x = support[s];
            if (x >= cfg.ao) { sib1[s] = sup_index(x - cfg.uo); sib2[s] = sup_index(x - cfg.ao); }
            else if (x >= cfg.uo) sib1[s] = sup_index(x + cfg.uo);
            else sib1[s] = sup_index(x + cfg.ao);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nU; i += kEmNT) ucnt[sup_index(U[i].x)] = U[i].y;
    for (uint32_t c = threadIdx.x; c < K; c += kEmNT)
        for (uint32_t w = cls_woff[c]; w < cls_woff[c + 1]; ++w) {
            const uint32_t s = sup_index(cls_w[w]);
            cls_sidx[w] = s;
            inv_pairs[w] = ((uint64_t)s << 32) | c;
        }
    __syncthreads();
    EM_MARK(3);
    // 4. inverted index: for every support entry the classes containing it, ascending class
    tiled_bitonic_sort_by<kEmNT, 4096>(inv_pairs, Wc, [](uint64_t a, uint64_t b) { return a > b; }, reinterpret_cast<uint64_t*>(s_tile));
    // slot_off[s] = first pair with support idx >= s: count the memberships per entry, exclusive scan
    for (uint32_t s = threadIdx.x; s <= S; s += kEmNT) slot_off[s] = 0;
    __syncthreads();
    for (uint32_t q = threadIdx.x; q < Wc; q += kEmNT) atomicAdd(&slot_off[(uint32_t)(inv_pairs[q] >> 32)], 1u);
    __syncthreads();
    {
        uint32_t carry = 0;
        for (uint32_t base = 0; base <= S; base += kEmNT) {
            const uint32_t s = base + threadIdx.x;
            const uint32_t c = s <= S ? slot_off[s] : 0u;
            uint32_t tot;
            const uint32_t ex = block_excl_scan<kEmNT>(c, s_ws, tot);
            if (s <= S) slot_off[s] = carry + ex;
            carry += tot;
        }
    }
    __syncthreads();
    EM_MARK(4);
    // 4b. The rounds only ever change entries that have a single-label count or sit in some class label
    // ("active"); every other support entry (the USA sibling statuses marked for em.rs:351-356) is produced
    // as 0 by each round.  Compact the active entries and express everything the rounds touch in active ids:
    // per entry one 16-byte record, per label word one, the memberships as plain class ids.  Two extra slots
    // stand for "an inactive sibling" (the initial value in round 1, 0 afterwards) and "no sibling" (0; adding
    // +0.0f to a non-negative float is exact, so one three-term formula serves every status).
    uint32_t A = 0;
    for (uint32_t base = 0; base < S; base += kEmNT) {
        const uint32_t s2 = base + threadIdx.x;
        const uint32_t h = s2 < S && (ucnt[s2] != 0 || slot_off[s2 + 1] > slot_off[s2]);
        uint32_t tot;
        const uint32_t ex = block_excl_scan<kEmNT>(h, s_ws, tot);
        if (s2 < S) aid[s2] = h ? A + ex : 0xFFFFFFFFu;
        A += tot;
    }
    __syncthreads();
    const uint32_t Z0 = A, Z1 = A + 1;
    auto amap = [&](uint32_t x) -> uint32_t {
        if (x == 0xFFFFFFFFu) return Z1;
        const uint32_t a = aid[x];
        return a == 0xFFFFFFFFu ? Z0 : a;
    };
    for (uint32_t s2 = threadIdx.x; s2 < S; s2 += kEmNT) {
        const uint32_t a = aid[s2];
        if (a == 0xFFFFFFFFu) continue;
        ent[a] = make_uint4(ucnt[s2], amap(sib1[s2]), amap(sib2[s2]), slot_off[s2]);
        act_col[a] = support[s2];
    }
    if (threadIdx.x == 0) ent[A] = make_uint4(0u, Z1, Z1, Wc);
    for (uint32_t w = threadIdx.x; w < Wc; w += kEmNT) {
        const uint32_t s2 = cls_sidx[w];
        lw3[w] = make_uint4(aid[s2], amap(sib1[s2]), amap(sib2[s2]), 0u);
        memb[w] = (uint32_t)inv_pairs[w];
    }
    EM_MARK(4);
    if (threadIdx.x == 0) em_hdr[cell] = make_uint4(A, K, Wc, 0u);  // the rounds run in k_em_rounds
#ifdef AFQ_EM_TIMING
    EM_MARK(5);
    if (threadIdx.x == 0 && (blockIdx.x % 1000) == 7) { printf("em setup nrec=%u nU=%u M=%u K=%u S=%u Wc=%u A=%u:", m.nrec, nU, M, K, S, Wc, A); for (int i = 1; i <= 5; ++i) printf(" p%d=%.3fms", i, (double)(tmark[i] - tmark[i - 1]) / 1e5); printf("\n"); }
#endif
}

// acc + sum over q in [q0, q1), in that order, of (iv(q) >= 0 ? ab * iv(q) : 0) - by the whole wave: the loads
// of 64 memberships go out together, the additions stay one after the other (float addition is not associative
// and the order is the parity contract).  An entry that sits in hundreds of classes (a highly expressed gene)
// otherwise makes its one thread walk hundreds of dependent loads per round while the wave waits.
// All 64 lanes must call it with the same arguments; every lane returns the result.
constexpr uint32_t kEmHeavy = 8;   // memberships above which an entry is summed by the wave
template <typename InvAt>
__device__ __forceinline__ float wave_ordered_sum(float acc, float ab, uint32_t q0, uint32_t q1, InvAt&& inv_at) {
    const uint32_t lane = lane_id();
    for (uint32_t base = q0; base < q1; base += 64) {
        const uint32_t q = base + lane;
        const float iv = q < q1 ? inv_at(q) : -1.0f;
        // every lane forms its own term; a skipped term is +0.0f, which leaves a non-negative sum bit for bit
        // unchanged, so the chain below needs no branches: 64 dependent adds fed by constant-lane reads
        const float term = iv >= 0.0f ? ab * iv : 0.0f;
        const uint32_t tb = __float_as_uint(term);
#pragma unroll
        for (int i = 0; i < 64; ++i) acc += __uint_as_float(__builtin_amdgcn_readlane(tb, i));
    }
    return acc;
}
__device__ __forceinline__ uint32_t bcast_u32(uint32_t v, uint32_t src_lane) { return __builtin_amdgcn_readlane(v, (int)src_lane); }
__device__ __forceinline__ float bcast_f32(float v, uint32_t src_lane) { return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), (int)src_lane)); }

// ---------------------------------------------------------------------------
// The EM rounds.  k_em left, per cell, the compact structures of section 4b in global scratch; the rounds
// themselves used to stream them from L2/HBM every round (~1.5 MB of cache lines per round and cell - with a
// thousand cells in flight that is the memory system's full throughput, for 20-100 rounds).  Here one
// 1024-thread workgroup takes a cell and keeps everything the rounds touch ON CHIP: abundances, 1/denominators,
// class offsets/counts, label words and memberships (16-bit ids) in LDS, the per-entry records in registers
// (8 entries per thread).  A round is then LDS traffic and three barriers.  Cells too big for that (more than
// 8192 active entries or classes, or > 144 KiB of LDS) run the same arithmetic out of global memory.
// Arithmetic and its order are unchanged (bit-identical to the oracle).
constexpr int kEmRNT = 1024;
constexpr uint32_t kEmPer = 8;
constexpr uint32_t kEmLdsWords = 36 * 1024;
__global__ __launch_bounds__(kEmRNT) void k_em_rounds(const CellMeta* __restrict__ meta, const uint32_t* __restrict__ nnz_unique,
                                                     const uint32_t* __restrict__ lab_cnt, const uint64_t* __restrict__ em_off,
                                                     uint32_t* __restrict__ scratch, uint32_t* __restrict__ out_nnz,
                                                     const uint4* __restrict__ em_hdr, const uint32_t* __restrict__ em_order, EmCfg cfg) {
    __shared__ uint32_t s_ws[kEmRNT / 64];
    __shared__ uint32_t s_flag[2];
    __shared__ __attribute__((aligned(16))) uint32_t s_mem[kEmLdsWords];
    const uint32_t cell = em_order[blockIdx.x];
#ifdef AFQ_EM_TIMING
    __shared__ unsigned long long tm2[6];
#define EM2_MARK(i) do { __syncthreads(); if (threadIdx.x == 0 && (blockIdx.x % 1000) == 7) tm2[i] = wall_clock64(); } while (0)
#else
#define EM2_MARK(i) do {} while (0)
#endif
    const uint4 hdr = em_hdr[cell];
    if (hdr.w) return;  // no multi-label class: k_em already wrote the row
    const uint32_t A = hdr.x, K = hdr.y, Wc = hdr.z;
    const uint32_t nU = nnz_unique[cell], W = lab_cnt[2 * cell], M = lab_cnt[2 * cell + 1];
    const uint32_t capS = (nU + W) * (cfg.usa ? 3u : 1u);
    const EmScratch sc = em_carve(scratch, em_off[cell], nU, W, M, capS);
    uint2* out = sc.out;
    const uint4* ent = sc.ent;
    const uint4* lw3 = sc.lw3;
    const uint32_t* memb = sc.memb;
    const uint32_t* cls_woff = sc.cls_woff;
    const uint32_t* cls_cnt = sc.cls_cnt;
    const uint32_t* act_col = sc.act_col;
    const uint32_t Z0 = A, Z1 = A + 1;
    const uint32_t tid = threadIdx.x;
    const uint32_t lw_words = (3 * Wc + 1) / 2, mb_words = (Wc + 1) / 2;
    const uint32_t need = (A + 2) + K + (K + 1) + K + lw_words + mb_words;
    const bool fits = need <= kEmLdsWords && A <= kEmPer * kEmRNT && K <= kEmPer * kEmRNT;
    uint32_t nout = 0;
    [[maybe_unused]] uint32_t it_dbg = 0;
    EM2_MARK(0);
    if (fits) {
        float* vin = reinterpret_cast<float*>(s_mem);
        float* inv = vin + (A + 2);
        uint32_t* woff = reinterpret_cast<uint32_t*>(inv + K);
        uint32_t* cnt = woff + (K + 1);
        uint16_t* lw16 = reinterpret_cast<uint16_t*>(cnt + K);
        uint16_t* mb16 = reinterpret_cast<uint16_t*>(cnt + K + lw_words);
        for (uint32_t c = tid; c <= K; c += kEmRNT) woff[c] = cls_woff[c];
        for (uint32_t c = tid; c < K; c += kEmRNT) cnt[c] = cls_cnt[c];
        for (uint32_t w = tid; w < Wc; w += kEmRNT) {
            const uint4 l = lw3[w];
            lw16[3 * w] = (uint16_t)l.x; lw16[3 * w + 1] = (uint16_t)l.y; lw16[3 * w + 2] = (uint16_t)l.z;
            mb16[w] = (uint16_t)memb[w];
        }
        uint32_t e_cnt[kEmPer], e_sib[kEmPer], e_q0[kEmPer], e_q1[kEmPer];
        float acc[kEmPer];
        const float uni = 1.0f / (float)cfg.num_alphas;
#pragma unroll
        for (uint32_t j = 0; j < kEmPer; ++j) {
            const uint32_t a = tid + j * kEmRNT;
            e_cnt[j] = 0; e_sib[j] = 0; e_q0[j] = 0; e_q1[j] = 0; acc[j] = 0.0f;
            if (a < A) {
                const uint4 e = ent[a];
                e_cnt[j] = e.x; e_sib[j] = e.y | (e.z << 16); e_q0[j] = e.w; e_q1[j] = ent[a + 1].w;
                vin[a] = cfg.init_uniform ? uni : ((float)e.x + 0.5f) * 1e-3f;
            }
        }
        if (tid == 0) { vin[Z0] = cfg.init_uniform ? uni : ((float)0u + 0.5f) * 1e-3f; vin[Z1] = 0.0f; }
        __syncthreads();
        EM2_MARK(1);
        uint32_t it = 0;
        bool conv = true, last_round = false;
        while (it < kMinIter || (it < kMaxIter && !conv) || last_round) {
            // (A) per class: denominator in label order (get_abundance_for, em.rs:167-187)
#pragma unroll
            for (uint32_t j = 0; j < kEmPer; ++j) {
                const uint32_t c = tid + j * kEmRNT;
                if (c < K) {
                    float denom = 0.0f;
                    const uint32_t we = woff[c + 1];
                    for (uint32_t w = woff[c]; w < we; ++w)
                        denom += (vin[lw16[3 * w + 1]] + vin[lw16[3 * w + 2]]) + vin[lw16[3 * w]];
                    inv[c] = denom > 0.0f ? (float)cnt[c] / denom : -1.0f;
                }
            }
            if (tid == 0) s_flag[0] = 0;
            __syncthreads();
            // (B) per active entry: single-label count, then class contributions in class order
            bool bad = false;
#pragma unroll
            for (uint32_t j = 0; j < kEmPer; ++j) {
                const uint32_t a = tid + j * kEmRNT;
                const bool valid = a < A;
                const bool heavy = valid && e_q1[j] - e_q0[j] > kEmHeavy;
                float x = 0.0f, old = 0.0f, ab = 0.0f;
                if (valid) {
                    if (e_cnt[j]) x += (float)e_cnt[j];
                    old = vin[a];
                    ab = (vin[e_sib[j] & 0xFFFFu] + vin[e_sib[j] >> 16]) + old;
                    if (!heavy)
                        for (uint32_t q = e_q0[j]; q < e_q1[j]; ++q) {
                            const float iv = inv[mb16[q]];
                            if (iv >= 0.0f) x += ab * iv;
                        }
                }
                for (uint64_t hm = __ballot(heavy); hm; hm &= hm - 1) {
                    const uint32_t L = (uint32_t)__builtin_ctzll(hm);
                    const float r = wave_ordered_sum(bcast_f32(x, L), bcast_f32(ab, L), bcast_u32(e_q0[j], L), bcast_u32(e_q1[j], L),
                                                     [&](uint32_t q) { return inv[mb16[q]]; });
                    if (lane_id() == L) x = r;
                }
                if (valid) {
                    acc[j] = x;
                    if (x > kAlphaCheckCutoff && fabsf(old - x) > kRelDiffTol) bad = true;
                }
            }
            if (bad) s_flag[0] = 1;
            __syncthreads();  // every read of the old abundances is done
            conv = s_flag[0] == 0;
#pragma unroll
            for (uint32_t j = 0; j < kEmPer; ++j) {
                const uint32_t a = tid + j * kEmRNT;
                if (a < A) vin[a] = acc[j];
            }
            if (tid == 0) vin[Z0] = 0.0f;  // inactive entries come out of every round as 0
            ++it;
            __syncthreads();
            if (cfg.usa) {
                if (last_round) break;
                if (it >= kMinIter && conv) {
#pragma unroll
                    for (uint32_t j = 0; j < kEmPer; ++j) {
                        const uint32_t a = tid + j * kEmRNT;
                        if (a < A && vin[a] < kMinOutputAlpha) vin[a] = 0.0f;
                    }
                    last_round = true;
                    __syncthreads();
                }
            }
        }
        it_dbg = it;
        EM2_MARK(2);
        // floor and emit the non-zero alphas in column order (active ids ascend with the column)
        for (uint32_t base = 0; base < A; base += kEmRNT) {
            const uint32_t a = base + tid;
            float v = a < A ? vin[a] : 0.0f;
            if (v < kMinOutputAlpha) v = 0.0f;
            const uint32_t h = v > 0.0f;
            uint32_t tot;
            const uint32_t ex = block_excl_scan<kEmRNT>(h, s_ws, tot);
            if (h) out[nout + ex] = make_uint2(act_col[a], __float_as_uint(v));
            nout += tot;
        }
    } else {
        // bigger cells: the two randomly accessed arrays (abundances, 1/denominators) still live in LDS when they
        // fit; the entry / label-word / membership records are streamed, coalesced, from global memory
        const bool mid = (A + 2) + K <= kEmLdsWords;
        float* vin = mid ? reinterpret_cast<float*>(s_mem) : sc.a_in;
        float* vout = sc.a_out;
        float* inv = mid ? reinterpret_cast<float*>(s_mem) + (A + 2) : sc.inv;
        const float uni = 1.0f / (float)cfg.num_alphas;
        for (uint32_t a = threadIdx.x; a < A; a += kEmRNT) vin[a] = cfg.init_uniform ? uni : ((float)ent[a].x + 0.5f) * 1e-3f;
        if (threadIdx.x == 0) { vin[Z0] = cfg.init_uniform ? uni : ((float)0u + 0.5f) * 1e-3f; vin[Z1] = 0.0f; }
        __syncthreads();
        uint32_t it = 0;
        bool conv = true, last_round = false;
        while (it < kMinIter || (it < kMaxIter && !conv) || last_round) {
            // (A) per class: denominator in label order (get_abundance_for, em.rs:167-187)
            // four classes per thread per trip, their loads issued together: the rounds are chains of dependent
            // L2 round trips, and a thread walking its classes one at a time has only one chain in flight
            for (uint32_t c0 = threadIdx.x; c0 < K; c0 += 4 * kEmRNT) {
                uint32_t wb[4], we[4], cn[4];
                uint4 l0[4], l1[4], l2[4];
    #pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t c = c0 + j * kEmRNT;
                    const bool ok = c < K;
                    wb[j] = ok ? cls_woff[c] : 0u;
                    we[j] = ok ? cls_woff[c + 1] : 0u;
                    cn[j] = ok ? cls_cnt[c] : 0u;
                }
    #pragma unroll
                for (int j = 0; j < 4; ++j) {
                    l0[j] = lw3[wb[j] < we[j] ? wb[j] : 0u];
                    l1[j] = lw3[wb[j] + 1 < we[j] ? wb[j] + 1 : 0u];
                    l2[j] = lw3[wb[j] + 2 < we[j] ? wb[j] + 2 : 0u];
                }
    #pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t c = c0 + j * kEmRNT;
                    if (c >= K) continue;
                    float denom = 0.0f;
                    if (wb[j] < we[j]) denom += (vin[l0[j].y] + vin[l0[j].z]) + vin[l0[j].x];
                    if (wb[j] + 1 < we[j]) denom += (vin[l1[j].y] + vin[l1[j].z]) + vin[l1[j].x];
                    if (wb[j] + 2 < we[j]) denom += (vin[l2[j].y] + vin[l2[j].z]) + vin[l2[j].x];
                    for (uint32_t w = wb[j] + 3; w < we[j]; ++w) {
                        const uint4 l = lw3[w];
                        denom += (vin[l.y] + vin[l.z]) + vin[l.x];
                    }
                    inv[c] = denom > 0.0f ? (float)cn[j] / denom : -1.0f;
                }
            }
            if (threadIdx.x == 0) s_flag[0] = 0;
            __syncthreads();
            // (B) per active entry: single-label count, then class contributions in class order
            bool bad = false;
            for (uint32_t a0 = threadIdx.x; a0 - lane_id() < A; a0 += 4 * kEmRNT) {  // wave-uniform trip count: the heavy-entry sums need every lane
                uint4 e[4];
                uint32_t qe[4], m0[4], m1[4];
                float i0[4], i1[4];
    #pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t a = a0 + j * kEmRNT;
                    e[j] = ent[a < A ? a : A];         // ent[A] is the sentinel record
                    qe[j] = ent[a < A ? a + 1 : A].w;
                }
    #pragma unroll
                for (int j = 0; j < 4; ++j) {
                    m0[j] = memb[e[j].w < qe[j] ? e[j].w : 0u];
                    m1[j] = memb[e[j].w + 1 < qe[j] ? e[j].w + 1 : 0u];
                }
    #pragma unroll
                for (int j = 0; j < 4; ++j) { i0[j] = inv[m0[j]]; i1[j] = inv[m1[j]]; }
    #pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t a = a0 + j * kEmRNT;
                    const bool valid = a < A;
                    const bool heavy = valid && qe[j] - e[j].w > 2 + kEmHeavy;
                    float acc = 0.0f, old = 0.0f, ab = 0.0f;
                    if (valid) {
                        if (e[j].x) acc += (float)e[j].x;
                        old = vin[a];
                        ab = (vin[e[j].y] + vin[e[j].z]) + old;
                        if (e[j].w < qe[j] && i0[j] >= 0.0f) acc += ab * i0[j];
                        if (e[j].w + 1 < qe[j] && i1[j] >= 0.0f) acc += ab * i1[j];
                        if (!heavy)
                            for (uint32_t q = e[j].w + 2; q < qe[j]; ++q) {
                                const float iv = inv[memb[q]];
                                if (iv >= 0.0f) acc += ab * iv;
                            }
                    }
                    for (uint64_t hm = __ballot(heavy); hm; hm &= hm - 1) {
                        const uint32_t L = (uint32_t)__builtin_ctzll(hm);
                        const float r = wave_ordered_sum(bcast_f32(acc, L), bcast_f32(ab, L), bcast_u32(e[j].w, L) + 2, bcast_u32(qe[j], L),
                                                         [&](uint32_t q) { return inv[memb[q]]; });
                        if (lane_id() == L) acc = r;
                    }
                    if (valid) {
                        vout[a] = acc;
                        if (acc > kAlphaCheckCutoff && fabsf(old - acc) > kRelDiffTol) bad = true;
                    }
                }
            }
            if (bad) s_flag[0] = 1;
            __syncthreads();
            conv = s_flag[0] == 0;
            for (uint32_t a = threadIdx.x; a < A; a += kEmRNT) vin[a] = vout[a];
            if (threadIdx.x == 0) vin[Z0] = 0.0f;  // inactive entries come out of every round as 0
            ++it;
            __syncthreads();
            if (cfg.usa) {
                if (last_round) break;
                if (it >= kMinIter && conv) {
                    for (uint32_t a = threadIdx.x; a < A; a += kEmRNT) if (vin[a] < kMinOutputAlpha) vin[a] = 0.0f;
                    last_round = true;
                    __syncthreads();
                }
            }
        }
        // 6. floor and emit the non-zero alphas in column order (active ids ascend with the column)
        for (uint32_t base = 0; base < A; base += kEmRNT) {
            const uint32_t a = base + threadIdx.x;
            float v = a < A ? vin[a] : 0.0f;
            if (v < kMinOutputAlpha) v = 0.0f;
            const uint32_t h = v > 0.0f;
            uint32_t tot;
            const uint32_t ex = block_excl_scan<kEmRNT>(h, s_ws, tot);
            if (h) out[nout + ex] = make_uint2(act_col[a], __float_as_uint(v));
            nout += tot;
        }
    }
    if (threadIdx.x == 0) out_nnz[cell] = nout;
    EM2_MARK(3);
#ifdef AFQ_EM_TIMING
    if (threadIdx.x == 0 && (blockIdx.x % 1000) == 7) printf("em rounds cell nrec=%u A=%u K=%u Wc=%u need=%u fits=%d it=%u: load=%.3f rounds=%.3f out=%.3f total=%.3f ms\n", meta[cell].nrec, A, K, Wc, need, (int)fits, it_dbg, (double)(tm2[1]-tm2[0])/1e5, (double)(tm2[2]-tm2[1])/1e5, (double)(tm2[3]-tm2[2])/1e5, (double)(tm2[3]-tm2[0])/1e5);
#endif
}

// EM output pairs -> final CSR
__global__ __launch_bounds__(256) void k_compact_em(uint32_t n_cells, const uint64_t* __restrict__ em_off,
                                                   const uint32_t* __restrict__ scratch, const uint32_t* __restrict__ nnz,
                                                   const uint64_t* __restrict__ cell_ptr, uint32_t* __restrict__ gene,
                                                   float* __restrict__ val) {
    const uint32_t cell = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (cell >= n_cells) return;
    const uint2* src = reinterpret_cast<const uint2*>(scratch + em_off[cell]);
    const uint32_t n = nnz[cell];
    const uint64_t o = cell_ptr[cell];
    for (uint32_t i = lane_id(); i < n; i += 64) {
        const uint2 p = src[i];
        gene[o + i] = p.x;
        val[o + i] = __uint_as_float(p.y);
    }
}

// ---------------------------------------------------------------------------
// staging pairs -> final CSR (wave per cell)
__global__ __launch_bounds__(256) void k_compact(const CellMeta* __restrict__ meta, uint32_t n_cells,
                                                const uint64_t* __restrict__ keys0, const uint64_t* __restrict__ keys1,
                                                const uint32_t* __restrict__ nnz,
                                                const uint64_t* __restrict__ cell_ptr, uint32_t* __restrict__ gene,
                                                float* __restrict__ val) {
    const uint32_t cell = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (cell >= n_cells) return;
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
                                                       uint32_t* __restrict__ o_n) {
    __shared__ uint32_t s_ws[kAtacNT / 64];
    const uint32_t cell = blockIdx.x;
    const uint64_t b0 = cell_ptr[cell];
    const uint32_t n = (uint32_t)(cell_ptr[cell + 1] - b0);
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

// ---------------------------------------------------------------------------
// launchers
#define AFQ_LAUNCH(kern, grid, block, stream, ...) hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, stream, __VA_ARGS__)

void launch_gather_headers(hipStream_t s, const uint8_t* bytes, size_t n_bytes, const uint64_t* chunk_off,
                           uint32_t n_cells, uint32_t* hdr) {
    if (!n_cells) return;
    AFQ_LAUNCH(k_gather_headers, (n_cells + 255) / 256, 256, s, bytes, n_bytes, chunk_off, n_cells, hdr);
}

template <int BW, int UW>
static void launch_decode_t(hipStream_t s, const DecodeArgs& a) {
    uint32_t grid = (a.n_cells + 3) / 4;
    if (a.chk) {  // fix-up mode: verify every cell's proof, then a modest persistent grid walks the (normally empty) list
        AFQ_LAUNCH(k_verify_cells, (a.n_cells + 255) / 256, 256, s, a.meta, a.n_cells, a.chk, a.cell_nkeys, a.st, a.fix_list);
        grid = grid < 1024 ? grid : 1024;
    }
    AFQ_LAUNCH((k_decode<BW, UW>), grid, 256, s, a.bytes, a.n_bytes, a.meta, a.n_cells, a.t2g,
               a.ref_count, a.num_genes, a.keys0, a.cell_nkeys, a.bc_out, a.st, a.chk ? a.fix_list : nullptr, a.pug);
}

int launch_decode(hipStream_t s, const DecodeArgs& a, uint32_t bw, uint32_t uw) {
    if (!a.n_cells) return 0;
#define AFQ_CASE(B, U) if (bw == B && uw == U) { launch_decode_t<B, U>(s, a); return 0; }
    AFQ_CASE(4, 4) AFQ_CASE(4, 8) AFQ_CASE(8, 4) AFQ_CASE(8, 8)
    AFQ_CASE(1, 1) AFQ_CASE(1, 2) AFQ_CASE(1, 4) AFQ_CASE(1, 8)
    AFQ_CASE(2, 1) AFQ_CASE(2, 2) AFQ_CASE(2, 4) AFQ_CASE(2, 8)
    AFQ_CASE(4, 1) AFQ_CASE(4, 2) AFQ_CASE(8, 1) AFQ_CASE(8, 2)
#undef AFQ_CASE
    return -1;
}

template <int BW, int UW>
static void launch_decode_par_t(hipStream_t s, const DecodeArgs& a) {
    AFQ_LAUNCH((k_slab_setup<BW, UW>), (a.n_cells + 3) / 4, 256, s, a.bytes, a.meta, a.n_cells, a.slab_prefix,
               a.slab_cell, a.cell_bc);
    const uint32_t n_groups = (a.n_slabs + kSlabsPerWave - 1) / kSlabsPerWave;
    const uint32_t n_cols = n_groups < kDecodeCols ? n_groups : kDecodeCols;
    const uint32_t n_waves = n_cols * ((n_groups + n_cols - 1) / n_cols);
    if (a.pug.h)  // the batch has PUG cells: instance that also emits (label hash, umi, offset) per read
        AFQ_LAUNCH((k_decode_par<BW, UW, true>), (n_waves + 3) / 4, 256, s, a.bytes, a.meta, a.n_cells, a.slab_prefix, a.slab_cell,
                   a.cell_bc, a.n_slabs, a.t2g, a.ref_count, a.num_genes, a.keys0, a.cell_nkeys, a.bc_out,
                   const_cast<CellChk*>(a.chk), a.pug);
    else if (a.short_records && a.trivial)
        AFQ_LAUNCH((k_decode_recs<BW, UW, true>), (n_waves + 3) / 4, 256, s, a.bytes, a.meta, a.n_cells, a.slab_prefix, a.slab_cell,
                   a.cell_bc, a.n_slabs, a.t2g, a.ref_count, a.num_genes, a.keys0, a.cell_nkeys, a.bc_out,
                   const_cast<CellChk*>(a.chk));
    else if (a.short_records)
        AFQ_LAUNCH((k_decode_recs<BW, UW, false>), (n_waves + 3) / 4, 256, s, a.bytes, a.meta, a.n_cells, a.slab_prefix, a.slab_cell,
                   a.cell_bc, a.n_slabs, a.t2g, a.ref_count, a.num_genes, a.keys0, a.cell_nkeys, a.bc_out,
                   const_cast<CellChk*>(a.chk));
    else if (a.trivial)
        AFQ_LAUNCH((k_decode_keys<BW, UW, true>), (n_waves + 3) / 4, 256, s, a.bytes, a.meta, a.n_cells, a.slab_prefix, a.slab_cell,
                   a.cell_bc, a.n_slabs, a.t2g, a.ref_count, a.num_genes, a.keys0, a.cell_nkeys, a.bc_out,
                   const_cast<CellChk*>(a.chk));
    else
        AFQ_LAUNCH((k_decode_keys<BW, UW, false>), (n_waves + 3) / 4, 256, s, a.bytes, a.meta, a.n_cells, a.slab_prefix, a.slab_cell,
                   a.cell_bc, a.n_slabs, a.t2g, a.ref_count, a.num_genes, a.keys0, a.cell_nkeys, a.bc_out,
                   const_cast<CellChk*>(a.chk));
}

bool decode_par_supported(uint32_t bw, uint32_t uw) { return (bw == 4 || bw == 8) && (uw == 4 || uw == 8); }

int launch_decode_par(hipStream_t s, const DecodeArgs& a, uint32_t bw, uint32_t uw) {
    if (!a.n_slabs) return 0;
    if (bw == 4 && uw == 4) { launch_decode_par_t<4, 4>(s, a); return 0; }
    if (bw == 4 && uw == 8) { launch_decode_par_t<4, 8>(s, a); return 0; }
    if (bw == 8 && uw == 4) { launch_decode_par_t<8, 4>(s, a); return 0; }
    if (bw == 8 && uw == 8) { launch_decode_par_t<8, 8>(s, a); return 0; }
    return -1;
}

void launch_hist(hipStream_t s, const ResolveArgs& a) {
    if (!a.n_tiles) return;
    AFQ_LAUNCH(k_hist, a.n_tiles, 256, s, a.multi_cells, a.tile_prefix, a.n_multi, a.meta, a.cell_nkeys, a.keys0, a.cursor);
}

void launch_bucket_scan(hipStream_t s, const ResolveArgs& a) {
    if (!a.n_multi) return;
    AFQ_LAUNCH(k_bucket_scan, (a.n_multi + 3) / 4, 256, s, a.multi_cells, a.n_multi, a.meta, a.cursor);
}

void launch_scatter(hipStream_t s, const ResolveArgs& a) {
    if (!a.n_tiles) return;
    AFQ_LAUNCH(k_scatter, a.n_tiles, 256, s, a.multi_cells, a.tile_prefix, a.n_multi, a.meta, a.cell_nkeys, a.keys0,
               a.keys1, a.cursor);
}

static ResolveCfg make_rc(const ResolveArgs& a) {
    ResolveCfg rc;
    rc.usa = a.usa; rc.num_rows = a.num_rows; rc.uo = a.num_rows / 3; rc.ao = 2 * (a.num_rows / 3); rc.mode = 0;
    return rc;
}

void launch_resolve(hipStream_t s, const ResolveArgs& a) {
    if (!a.n_buckets) return;
    ResolveCfg rc = make_rc(a);
    LabArea la{a.lab, a.lab_cnt};
    BucketDesc* desc = reinterpret_cast<BucketDesc*>(a.bucket_desc);
    AFQ_LAUNCH(k_bucket_desc, (a.n_buckets + 255) / 256, 256, s, a.meta, a.bucket_cell, a.cell_nkeys, a.cursor, a.n_buckets, desc);
    const uint32_t n_cols = a.n_buckets < kResolveCols ? a.n_buckets : kResolveCols;
    const uint32_t grid = n_cols * ((a.n_buckets + n_cols - 1) / n_cols);
    if (a.lab)
        AFQ_LAUNCH(k_resolve<true>, grid, kResolveNT, s, desc, a.n_buckets, a.keys0, a.keys1, a.cell_ncols, a.nnz, a.ovf_list, a.st, rc, la);
    else
        AFQ_LAUNCH(k_resolve<false>, grid, kResolveNT, s, desc, a.n_buckets, a.keys0, a.keys1, a.cell_ncols, a.nnz, a.ovf_list, a.st, rc, la);
}

void launch_resolve_big(hipStream_t s, const ResolveArgs& a) {
    if (!a.n_multi) return;
    ResolveCfg rc = make_rc(a);
    LabArea la{a.lab, a.lab_cnt};
    BucketDesc* desc = reinterpret_cast<BucketDesc*>(a.bucket_desc);
    AFQ_LAUNCH(k_resolve_mid, 256, kMidNT, s, desc, a.keys0, a.keys1, a.cell_ncols, a.nnz, a.ovf_list, a.st, rc, la);
    AFQ_LAUNCH(k_resolve_big, 256, kBigNT, s, desc, a.meta, a.keys0, a.keys1, a.cell_ncols, a.ovf_list, a.st, rc, la);
}

// words of per-cell EM scratch for nU single-label columns, W label words, M ambiguous molecules
uint64_t em_scratch_words(uint32_t nU, uint32_t W, uint32_t M, bool usa) {
    const uint64_t capS = ((uint64_t)nU + W) * (usa ? 3u : 1u);
    // mirrors the carve at the top of k_em
    uint64_t w = 2 * (capS + 1)            // out
                 + 2 * ((uint64_t)W + 1)   // inv_pairs
                 + 3 * ((uint64_t)M + 1)   // order, cls_first, cls_cnt
                 + ((uint64_t)M + 2)       // cls_woff
                 + 2 * ((uint64_t)W + 1)   // cls_w, cls_sidx
                 + ((uint64_t)M + 1)       // inv
                 + 4 * (capS + 1)          // support, sib1, sib2, ucnt
                 + 2 * (capS + 2)          // a_in, a_out
                 + (capS + 2)              // slot_off
                 + (capS + 1)              // aid
                 + 4 * ((uint64_t)nU + W + 2)  // ent
                 + 4 * ((uint64_t)W + 1)   // lw3
                 + ((uint64_t)nU + W + 2)  // act_col
                 + ((uint64_t)W + 1)       // memb
                 + 4;                      // alignment slack for the 16-byte records
    return (w + 3) & ~3ull;  // keep slices 16-byte aligned
}

void launch_em(hipStream_t s, const ResolveArgs& a, uint32_t n_cells, const uint64_t* em_off, uint32_t* scratch,
               uint32_t* out_nnz, void* em_hdr_v, const uint32_t* em_order, uint32_t num_alphas, uint32_t init_uniform) {
    uint4* em_hdr = reinterpret_cast<uint4*>(em_hdr_v);
    if (!n_cells) return;
    EmCfg cfg{a.usa, num_alphas, a.num_rows / 3, 2 * (a.num_rows / 3), init_uniform};
    AFQ_LAUNCH(k_em, n_cells, kEmNT, s, a.meta, a.nnz, a.keys0, a.keys1, a.lab, a.lab_cnt, em_off, scratch, out_nnz, em_hdr, em_order, cfg);
    AFQ_LAUNCH(k_em_rounds, n_cells, kEmRNT, s, a.meta, a.nnz, a.lab_cnt, em_off, scratch, out_nnz, em_hdr, em_order, cfg);
}

void launch_compact_em(hipStream_t s, uint32_t n_cells, const uint64_t* em_off, const uint32_t* scratch, const uint32_t* nnz,
                       const uint64_t* cell_ptr, uint32_t* gene, float* val) {
    if (!n_cells) return;
    AFQ_LAUNCH(k_compact_em, (n_cells + 3) / 4, 256, s, n_cells, em_off, scratch, nnz, cell_ptr, gene, val);
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
                       uint16_t* o_cnt, uint32_t* o_n) {
    if (!n_cells) return;
    AFQ_LAUNCH(k_atac_dedup, n_cells, kAtacNT, s, ref, start, flen, cell_ptr, reinterpret_cast<Frag*>(scratch), o_ref, o_start,
               o_flen, o_cnt, o_n);
}

void launch_cell_hist(hipStream_t s, const ResolveArgs& a) {
    if (!a.n_hist) return;
    ResolveCfg rc = make_rc(a);
    AFQ_LAUNCH(k_cell_hist, a.n_hist, kHistNT, s, a.hist_cells, a.meta, a.keys0, a.keys1, a.cell_ncols, a.nnz, rc);
}

void launch_compact(hipStream_t s, const CellMeta* meta, uint32_t n_cells, const uint64_t* keys0, const uint64_t* keys1,
                    const uint32_t* nnz, const uint64_t* cell_ptr, uint32_t* gene, float* val) {
    if (!n_cells) return;
    AFQ_LAUNCH(k_compact, (n_cells + 3) / 4, 256, s, meta, n_cells, keys0, keys1, nnz, cell_ptr, gene, val);
}

}  // namespace afq
