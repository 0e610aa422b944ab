// afq_decode.hip - collated-RAD chunk bytes -> (umi << 20 | gene) keys (and, for parsimony cells, per-read
// (label key, umi, offset) records) on gfx950.  Replaces the chunk / record decode of the reference
// (libradicl's Chunk<R> reader as used at src/quant.rs:733-757, the per-read gene projection of
// src/pugutils.rs:774-781 / src/quant.rs:469-530):
//   k_gather_headers  chunk headers of a device-resident input
//   k_decode          sequential walk, any field width / alignment; also the re-decode of cells whose proof failed
//   k_slab_setup      slab -> cell table, barcode of every cell
//   k_decode_par      walk-free decode, one lane per record (general; emits the parsimony per-read records)
//   k_decode_keys     walk-free decode, one lane per dword (flat in record length)
//   k_decode_recs     walk-free decode, one lane per record with inline alignments (short records)
//   (k_verify_cells, the last step of the walk-free proof - DESIGN.md section 4 - is the first phase of k_decode's fix-up mode since round 6)
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include <stdint.h>
#include <stdio.h>

#include <algorithm>

#include "afq_common.h"
#include "afq_hooks.h"
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
                                               const CellChk* __restrict__ chk, PugOut pug) {
    constexpr uint32_t HDR = 4 + BW + UW;
    constexpr bool AL = (BW % 4 == 0) && (UW % 4 == 0);
    const uint32_t lane = lane_id();
    // plain mode: wave w takes cell w.  Fix-up mode (chk: after a walk-free decoder): the workgroup first takes the last step of
    // the walk-free proof for its 256 cells, a thread each (the accumulated candidate count and sizes against
    // the chunk header; cells that pass add their key count to the batch total, one atomic per workgroup), and its waves then
    // re-decode the cells that failed it (normally none) right here.  (Until round 6 the proof's last step was a kernel of its own,
    // k_verify_cells, with a list in global memory between the two: one 5 us launch and one boundary more per range.)
    __shared__ unsigned long long s_sum;
    __shared__ uint32_t s_nfail;
    __shared__ uint32_t s_fail[256];
    if (chk) {
        if (threadIdx.x == 0) { s_sum = 0; s_nfail = 0; }
        __syncthreads();
        const uint32_t vc = blockIdx.x * 256 + threadIdx.x;
        unsigned long long keys = 0;
        if (vc < n_cells) {
            const CellMeta vm = meta[vc];
            const CellChk c = chk[vc];
            if (c.fail == 0 && c.count == vm.nrec && c.words == vm.nbytes / 4 - 2) keys = cell_nkeys[vc];
            else s_fail[atomicAdd(&s_nfail, 1u)] = vc;
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) keys += __shfl_xor(keys, d);
        if (lane == 0 && keys) atomicAdd(&s_sum, keys);
        __syncthreads();
        if (threadIdx.x == 0) { if (s_sum) atomicAdd(&st->n_keys, s_sum); if (s_nfail) atomicAdd(&st->n_fallback, s_nfail); }
    }
    const uint32_t n_work = chk ? s_nfail : n_cells;
  for (uint32_t work = chk ? (threadIdx.x >> 6) : blockIdx.x * 4 + (threadIdx.x >> 6); work < n_work; work += chk ? 4u : gridDim.x * 4) {
    const uint32_t cell = chk ? s_fail[work] : work;
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
                lhash = label_hash_init(na) ^ pug.salt;
                uint32_t t01[2] = {0, 0};
                for (uint32_t j = 0; j < na; ++j) {
                    const uint32_t t = ld_u32(rp + 4 * j, ral) & 0x7FFFFFFFu;
                    if (t >= ref_count) set_err(st, kErrRefRange, cell);
                    lhash = label_hash_step(lhash, t);
                    if (j < 2) t01[j] = t;
                }
                lhash = label_key(lhash & pug.mask, na, t01[0], t01[1]);
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
                    if (first && mode_pug_gene(m.mode)) lhash += gene_set_hash_term(gj ^ (uint32_t)pug.salt);
                }
            }
            if (mode_pug_gene(m.mode)) {  // gene-level PUG: order-independent hash of the read's gene set
                if (!ovf) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) if ((uint32_t)i < k) lhash += gene_set_hash_term(g[i] ^ (uint32_t)pug.salt);
                }
                lhash ^= (uint64_t)kcnt * kHashMul;
                lhash = label_key(lhash & pug.mask, kcnt, g[0], g[1]);  // (kcnt <= 2 implies !ovf: g[0], g[1] are the read's genes)
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
                pug.h[slot] = lhash;
                if (UW == 4) pug.u[slot] = (umi << 32) | rec_dw;   // a 4-byte UMI travels with the record offset in one word (what the sort keys on)
                else { pug.u[slot] = umi; pug.o[slot] = rec_dw; }
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

// ---------------------------------------------------------------------------
// k_widen: chunks whose barcode / UMI fields are 1 or 2 bytes wide (UMIs of up to 8 nt: Drop-seq, CEL-Seq2, inDrop ...;
// convert.rs:323-344 picks the narrowest integer) have records of 4 + bw + uw + 4 na bytes - nothing in them is dword
// aligned, and both the walk-free decoders and the parsimony kernel (which reads labels straight out of the chunk)
// want dwords.  Rather than refusing such input, the batch is rewritten once on the device with both fields widened
// to 4 bytes (zero-extended; 8 stays 8): record r of a chunk moves from byte s to s + r * delta, delta = the bytes
// added per record, so every lane that owns a record start knows its destination.  One wave per cell walks the record
// chain exactly like k_decode does (a record's length is its own na).  Everything downstream sees an ordinary
// aligned 4/8-byte layout.
__global__ __launch_bounds__(256) void k_widen(const uint8_t* __restrict__ src, size_t n_src, const uint64_t* __restrict__ src_off,
                                              const CellMeta* __restrict__ meta, uint32_t n_cells, uint32_t bw, uint32_t uw,
                                              uint32_t ebw, uint32_t euw, uint8_t* __restrict__ dst, DevStatus* st, uint32_t bsplit) {
    // bsplit != 0: the barcode field is two integers of bsplit and bw - bsplit bytes (multi-barcode records of unequal widths);
    // each becomes a dword (ebw = 8)
    const uint32_t lane = lane_id();
    const uint32_t HDR = 4 + bw + uw, EHDR = 4 + ebw + euw, delta = EHDR - HDR;
    auto ld_n = [](const uint8_t* p, uint32_t n) -> uint64_t {
        uint64_t v = 0;
        for (uint32_t i = 0; i < n; ++i) v |= (uint64_t)p[i] << (8 * i);
        return v;
    };
    auto st_n = [](uint8_t* p, uint64_t v, uint32_t n) {   // p is dword aligned, n is 4 or 8
        *reinterpret_cast<uint32_t*>(p) = (uint32_t)v;
        if (n == 8) *reinterpret_cast<uint32_t*>(p + 4) = (uint32_t)(v >> 32);
    };
    for (uint32_t cell = blockIdx.x * 4 + (threadIdx.x >> 6); cell < n_cells; cell += gridDim.x * 4) {
        const CellMeta m = meta[cell];               // the WIDENED chunk: offset in dst, nbytes, nrec
        const uint64_t so = src_off[cell];
        const uint64_t src_nbytes = (uint64_t)m.nbytes - (uint64_t)m.nrec * delta;
        const uint64_t abase = so & ~3ull;
        const uint32_t mis = (uint32_t)(so - abase);
        uint64_t pos = (uint64_t)mis + 8;
        const uint64_t end = (uint64_t)mis + src_nbytes;
        uint8_t* const dchunk = dst + m.chunk_off;
        if (lane == 0) { reinterpret_cast<uint32_t*>(dchunk)[0] = m.nbytes; reinterpret_cast<uint32_t*>(dchunk)[1] = m.nrec; }
        uint32_t rec_seen = 0;
        bool bad = false;
        while (pos < end && !bad) {
            const uint64_t w = pos >> 8;
            const uint64_t wbyte = abase + (w << 8) + lane * 4;
            uint32_t v_cur = 0, v_next = 0;
            if (wbyte + 4 <= n_src) v_cur = *(const uint32_t*)(src + wbyte);
            else if (wbyte < n_src) { for (uint64_t q = wbyte; q < n_src; ++q) v_cur |= (uint32_t)src[q] << (8 * (q - wbyte)); }
            const uint64_t nb = wbyte + 256;
            if (nb + 4 <= n_src) v_next = *(const uint32_t*)(src + nb);
            else if (nb < n_src) { for (uint64_t q = nb; q < n_src; ++q) v_next |= (uint32_t)src[q] << (8 * (q - nb)); }
            const uint64_t wend = ((w + 1) << 8) < end ? ((w + 1) << 8) : end;
            uint64_t mask = 0, sub0 = 0, sub1 = 0;
            while (pos < wend) {   // scalar walk over the records that start in this window
                const uint32_t idx = __builtin_amdgcn_readfirstlane((uint32_t)(pos >> 2) & 63u);
                uint32_t na = __builtin_amdgcn_readlane(v_cur, idx);
                const uint32_t sh = ((uint32_t)pos & 3u) * 8u;
                if (sh) {
                    const uint32_t hi = idx < 63 ? __builtin_amdgcn_readlane(v_cur, idx + 1) : __builtin_amdgcn_readlane(v_next, 0);
                    na = (na >> sh) | (hi << (32 - sh));
                }
                if ((mask >> idx) & 1ull) { bad = true; pos = end; break; }   // two record starts inside one dword: records are >= 6 bytes, a dword holds one
                if (pos & 1) sub0 |= 1ull << idx;
                if (pos & 2) sub1 |= 1ull << idx;
                mask |= 1ull << idx;
                const uint64_t rec_bytes = (uint64_t)HDR + 4ull * na;
                if (pos + rec_bytes > end) { bad = true; pos = end; break; }
                pos += rec_bytes;
            }
            const bool is_start = (mask >> lane) & 1ull;
            if (is_start && !bad) {
                const uint32_t rec_idx = rec_seen + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
                const uint32_t sub = (uint32_t)((sub0 >> lane) & 1ull) | ((uint32_t)((sub1 >> lane) & 1ull) << 1);
                const uint64_t roff = abase + (w << 8) + lane * 4 + sub;
                const uint8_t* rec = src + roff;
                const uint32_t na = ld_u32(rec, false);
                const uint64_t drel = (roff - so) + (uint64_t)rec_idx * delta;
                if (rec_idx < m.nrec && drel + EHDR + 4ull * na <= m.nbytes) {
                    uint8_t* d = dchunk + drel;
                    *reinterpret_cast<uint32_t*>(d) = na;
                    if (bsplit) { st_n(d + 4, ld_n(rec + 4, bsplit), 4); st_n(d + 8, ld_n(rec + 4 + bsplit, bw - bsplit), 4); }
                    else st_n(d + 4, ld_n(rec + 4, bw), ebw);
                    st_n(d + 4 + ebw, ld_n(rec + 4 + bw, uw), euw);
                    const uint8_t* rp = rec + HDR;
                    uint32_t* dr = reinterpret_cast<uint32_t*>(d + EHDR);
                    for (uint32_t j = 0; j < na; ++j) dr[j] = ld_u32(rp + 4 * j, false);
                }
            }
            rec_seen += (uint32_t)__popcll(mask);
        }
        if (bad || rec_seen != m.nrec) { if (lane == 0) set_err(st, kErrRecordWalk, cell); }
    }
}

void launch_widen(hipStream_t s, const uint8_t* src, size_t n_src, const uint64_t* src_off, const CellMeta* meta, uint32_t n_cells,
                  uint32_t bw, uint32_t uw, uint32_t ebw, uint32_t euw, uint8_t* dst, DevStatus* st, uint32_t bsplit) {
    if (!n_cells) return;
    const uint32_t grid = std::min<uint32_t>((n_cells + 3) / 4, 8192u);
    AFQ_LAUNCH(k_widen, grid, 256, s, src, n_src, src_off, meta, n_cells, bw, uw, ebw, euw, dst, st, bsplit);
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
#ifndef AFQ_SLABS_PER_WAVE
#define AFQ_SLABS_PER_WAVE 8
#endif
constexpr uint32_t kSlabsPerWave = AFQ_SLABS_PER_WAVE;   // (measured on the headline, k_decode_recs per step: 2: 4.64-4.70 ms, 4: 4.46-4.47, 8: 4.41-4.42, 16: 4.30-4.31 on another box where 8 gave 4.28-4.32; profiles/history/run_r04ac.sh, run_r04ad.sh)
constexpr uint32_t kHalo = 64;
#ifndef AFQ_DECODE_COLS
#define AFQ_DECODE_COLS 8192
#endif
constexpr uint32_t kDecodeCols = AFQ_DECODE_COLS;
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
                lhash = label_hash_init(na) ^ pug.salt;
                uint32_t t0 = 0, t1 = 0;
                for (uint32_t j = 0; j < na; ++j) {
                    const uint32_t t = refw(j);
                    if (t >= ref_count) fail = true;
                    lhash = label_hash_step(lhash, t);
                    if (j == 0) t0 = t;
                    if (j == 1) t1 = t;
                }
                lhash = label_key(lhash & pug.mask, na, t0, t1);
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
                        if (first && pug_gene) lhash += gene_set_hash_term(gj ^ (uint32_t)pug.salt);
                    }
                }
                if (pug_gene) {  // gene-level PUG: order-independent hash of the read's gene set
                    if (!ovf) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) if ((uint32_t)q < k) lhash += gene_set_hash_term(g[q] ^ (uint32_t)pug.salt);
                    }
                    lhash ^= (uint64_t)kcnt * kHashMul;
                    lhash = label_key(lhash & pug.mask, kcnt, g[0], g[1]);  // (kcnt <= 2 implies !ovf: g[0], g[1] are the read's genes)
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
                    pug.h[slot] = lhash;
                    if (UWW == 1) pug.u[slot] = (umi << 32) | i; else { pug.u[slot] = umi; pug.o[slot] = i; }
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
// HD (round 4, late): which alignment word is the first of its record to name its gene is decided through an LDS hash table
// keyed by (record of the slab, gene) - every alignment word inserts itself with its position, the smallest position under a key
// stays - instead of by comparing with the three dwords before it and, for records of more than four alignments, by a serial
// scan back to the record's first alignment (a loop per lane, as long as the slab's longest record: on reads of E[na] = 3 with a
// geometric tail the kernel spent as many instructions on its scalar unit as on its vector units, both at 72 % of their issue
// slots).  The table's cost does not depend on the records' lengths.  The words a record that started before the slab has in
// front of it are looked up in the table (not inserted: any number of them), and a hit means "named before this slab".
constexpr uint32_t kHdLg = 9, kHdSlots = 1u << kHdLg;   // 256 alignment words per slab at most: the table is at most half full
template <int BW, int UW, bool TRIVIAL, bool HD = false>
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
    static_assert(!(TRIVIAL && HD), "the trivial rule keeps the scan");
    __shared__ uint32_t s_gene[4][HD ? 1 : 4 + kSlabWords];   // [4 pad] + gene of every alignment word of the slab (kNone elsewhere)
    __shared__ uint32_t s_first[4][HD ? 1 : kSlabWords];  // dword index of the first alignment word of the dword's record
    __shared__ uint32_t s_tkey[4][HD ? kHdSlots : 1];     // HD: record-of-the-slab << 20 | gene (kNone: free) ...
    __shared__ uint32_t s_tpos[4][HD ? kHdSlots : 1];     // ... and the smallest position (dword of the slab + 1; 0: an earlier slab) that named it
    const uint32_t lane = lane_id();
    const uint32_t wv = threadIdx.x >> 6;
    uint32_t* stage = s_stage[wv];
    uint32_t* gene_l = s_gene[wv];
    uint32_t* first_l = s_first[wv];
    [[maybe_unused]] uint32_t* tkey = s_tkey[wv];
    [[maybe_unused]] uint32_t* tpos = s_tpos[wv];
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
            for (int r = 0; r < 5; ++r) R[r] = AFQ_LD_DECODE(&W[s0 + r * 64 + lane]);
        } else {
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                const uint32_t i = s0 + r * 64 + lane;
                R[r] = i < nwords ? AFQ_LD_DECODE(&W[i]) : 0u;
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
        if constexpr (HD) {
#pragma unroll
            for (uint32_t k = 0; k < kHdSlots / 64; ++k) { tkey[k * 64 + lane] = kNone; tpos[k * 64 + lane] = kNone; }
        } else if (lane < 4) gene_l[lane] = kNone;  // pad in front of the slab's genes (the duplicate test looks back 3)
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
        [[maybe_unused]] uint32_t rid[4];   // HD: the dword's record - its header's place in the slab + 1, 0 for the record the slab starts in
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
            if constexpr (HD) rid[r] = in_stage ? sc + 1u : 0u;
            else {
                first_l[il] = fr;
                // more alignments back than the fast duplicate test covers, or some of them in an earlier slab
                slow = slow || (isref && p > 0 && (p > 3 || fr < s0)) || (triv && isref && p == 0 && na > 1);
            }
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
        uint64_t bal[4];
        uint32_t tot = 0;
        if constexpr (HD) {
            uint32_t hslot[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool isref = pos[r] != kNone;
                if (isref && gid[r] >= num_genes) fail = true;
                gid[r] = (isref && gid[r] < num_genes) ? gid[r] : kNone;
                hslot[r] = 0;
                if (gid[r] != kNone) {   // (record, gene) -> the smallest position that names it
                    const uint32_t key = (rid[r] << 20) | gid[r];
                    uint32_t slot = (key * 0x9E3779B1u) >> (32 - kHdLg);
                    for (;;) {
                        const uint32_t old = atomicCAS(&tkey[slot], kNone, key);
                        if (old == kNone || old == key) break;
                        slot = (slot + 1) & (kHdSlots - 1);
                    }
                    atomicMin(&tpos[slot], (uint32_t)(r * 64) + lane + 1u);
                    hslot[r] = slot;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            DT_MARK(3);
            if (cin_s != kNone && cin_s + HW < s0) {   // the record the slab starts in: what its alignments in front of the slab named counts as named
                const uint32_t cfr = cin_s + HW;
                const uint32_t cend = min(s0, cfr + min(cin_na, nwords));
                for (uint32_t q0 = cfr; q0 < cend; q0 += 64) {
                    const uint32_t q = q0 + lane;
                    const uint32_t t = q < cend ? W[q] & 0x7FFFFFFFu : kNone;
                    const uint32_t g = t < ref_count ? t2g[t] : kNone;
                    if (g < num_genes) {
                        for (uint32_t slot = (g * 0x9E3779B1u) >> (32 - kHdLg);; slot = (slot + 1) & (kHdSlots - 1)) {   // (key = record 0 << 20 | g)
                            const uint32_t k = tkey[slot];
                            if (k == g) tpos[slot] = 0u;
                            if (k == g || k == kNone) break;
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool e = gid[r] != kNone && tpos[hslot[r]] == (uint32_t)(r * 64) + lane + 1u;
                bal[r] = __ballot(e);
                gid[r] = e ? gid[r] : kNone;
                tot += (uint32_t)__popcll(bal[r]);
            }
        } else {
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
// PUG: the batch has parsimony cells; for those a record is not turned into keys but into one read
// (label key, UMI, record offset) for k_pug_cell - same rule as k_decode_par<.., true>.
constexpr uint32_t kInl = 3;
template <int BW, int UW, bool TRIVIAL, bool PUG>
__global__ __launch_bounds__(256, 8) void k_decode_recs(const uint8_t* __restrict__ bytes,
                                                    const CellMeta* __restrict__ meta, uint32_t n_cells,
                                                    const uint32_t* __restrict__ slab_prefix,
                                                    const uint32_t* __restrict__ slab_cell,
                                                    const uint64_t* __restrict__ cell_bc, uint32_t n_slabs,
                                                    const uint32_t* __restrict__ t2g, uint32_t ref_count,
                                                    uint32_t num_genes, uint64_t* __restrict__ keys0,
                                                    uint32_t* __restrict__ cell_nkeys,
                                                    uint64_t* __restrict__ bc_out, CellChk* __restrict__ chk,
                                                    [[maybe_unused]] PugOut pug) {
    static_assert(BW % 4 == 0 && UW % 4 == 0, "aligned layouts only");
    static_assert(!(TRIVIAL && PUG), "trivial and parsimony are different resolutions");
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
            for (int r = 0; r < 5; ++r) R[r] = AFQ_LD_DECODE(&W[s0 + r * 64 + lane]);
        } else {
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                const uint32_t i = s0 + r * 64 + lane;
                R[r] = i < nwords ? AFQ_LD_DECODE(&W[i]) : 0u;
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
            // a parsimony cell at transcript level hashes the refs themselves: no gene lookups for it (wave-uniform)
            const bool pugc = PUG && mode_is_pug(m.mode);
            const bool pug_txp = pugc && !mode_pug_gene(m.mode);
            uint32_t t[kInl], g[kInl];
            bool v[kInl];
#pragma unroll
            for (uint32_t j = 0; j < kInl; ++j) {
                const uint32_t pj = il + HW + j;
                t[j] = stage[pj < kStage ? pj : kStage - 1] & 0x7FFFFFFFu;
                v[j] = !slowrec && j < na_eff;
                if (v[j] && t[j] >= ref_count) { fail = true; v[j] = false; }
                g[j] = pug_txp ? 0u : t2g[v[j] ? t[j] : 0u];
            }
            if (!pug_txp) {
#pragma unroll
                for (uint32_t j = 0; j < kInl; ++j) {
                    if (v[j] && g[j] >= num_genes) { fail = true; v[j] = false; }
#pragma unroll
                    for (uint32_t q = 0; q < j; ++q) v[j] = v[j] && !(v[q] && g[q] == g[j]);
                }
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
            if (PUG && pugc) {   // one read per record: (label key, UMI, record offset); see k_decode_par
                uint64_t lkey;
                if (pug_txp) {
                    uint64_t hs = label_hash_init(na_eff) ^ pug.salt;
                    uint32_t t0 = 0, t1 = 0;
                    if (!slowrec) {
#pragma unroll
                        for (uint32_t j = 0; j < kInl; ++j) if (j < na_eff) hs = label_hash_step(hs, t[j]);
                        t0 = t[0]; t1 = t[1];
                    } else {
                        for (uint32_t j = 0; j < na_eff; ++j) {
                            const uint32_t tj = ref_at(j);
                            if (tj >= ref_count) fail = true;
                            hs = label_hash_step(hs, tj);
                            if (j == 0) t0 = tj;
                            if (j == 1) t1 = tj;
                        }
                    }
                    lkey = label_key(hs & pug.mask, na_eff, t0, t1);
                } else {   // gene level: order-independent hash of the read's distinct genes
                    uint64_t hs = 0;
                    uint32_t kc = 0, ga = 0, gb = 0;
                    auto add = [&](uint32_t gj) { if (kc == 0) ga = gj; else if (kc == 1) gb = gj; ++kc; hs += gene_set_hash_term(gj ^ (uint32_t)pug.salt); };
                    if (!slowrec) {
#pragma unroll
                        for (uint32_t j = 0; j < kInl; ++j) if (v[j]) add(g[j]);
                    } else for_each_first_gene(add);
                    hs ^= (uint64_t)kc * kHashMul;
                    lkey = label_key(hs & pug.mask, kc, ga, gb);
                }
                const uint64_t bm = __ballot(act);
                const uint32_t totp = (uint32_t)__popcll(bm);
                if (totp) {
                    uint32_t wbase = 0;
                    if (lane == 0) wbase = atomicAdd(&cell_nkeys[cur_cell], totp);
                    wbase = __builtin_amdgcn_readfirstlane(wbase);
                    if (wbase + totp > m.nrec) fail = true;
                    else if (act) {
                        const uint64_t slot = pug.rd_off[cur_cell] + wbase + __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
                        pug.h[slot] = lkey;
                        if (UWW == 1) pug.u[slot] = (umi << 32) | i; else { pug.u[slot] = umi; pug.o[slot] = i; }
                    }
                }
                continue;
            }
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
// What a range needs before its first kernel, in ONE launch: every buffer that starts at zero is cleared and every small host
// table is moved from the range's upload arena (one H2D copy) to its buffer.  It used to be some twenty hipMemsetAsync /
// hipMemcpyAsync calls per range - each its own blit kernel on the compute queue.
__global__ __launch_bounds__(256) void k_range_init(RangeInitOps ops, const uint8_t* __restrict__ arena) {
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x, gsz = (size_t)gridDim.x * 256;
    for (uint32_t i = 0; i < ops.n; ++i) {
        const RangeInitOp op = ops.op[i];
        uint8_t* dst = reinterpret_cast<uint8_t*>(op.dst);
        const uint8_t* src = op.src_off == ~0ull ? nullptr : arena + op.src_off;
        const uintptr_t al = reinterpret_cast<uintptr_t>(dst) | (src ? reinterpret_cast<uintptr_t>(src) : 0);
        if ((al & 15) == 0) {   // (hipMalloc'ed buffers and the arena's 16-byte slots: nearly always)
            const size_t n16 = op.bytes / 16;
            uint4* d4 = reinterpret_cast<uint4*>(dst);
            const uint4* s4 = reinterpret_cast<const uint4*>(src);
            for (size_t k = gid; k < n16; k += gsz) d4[k] = src ? s4[k] : make_uint4(0u, 0u, 0u, 0u);
            for (size_t k = n16 * 16 + gid; k < op.bytes; k += gsz) dst[k] = src ? src[k] : (uint8_t)0;
        } else if ((al & 3) == 0) {
            const size_t n4 = op.bytes / 4;
            uint32_t* d1 = reinterpret_cast<uint32_t*>(dst);
            const uint32_t* s1 = reinterpret_cast<const uint32_t*>(src);
            for (size_t k = gid; k < n4; k += gsz) d1[k] = src ? s1[k] : 0u;
            for (size_t k = n4 * 4 + gid; k < op.bytes; k += gsz) dst[k] = src ? src[k] : (uint8_t)0;
        } else {
            for (size_t k = gid; k < op.bytes; k += gsz) dst[k] = src ? src[k] : (uint8_t)0;
        }
    }
}
void launch_range_init(hipStream_t s, const RangeInitOps& ops, const uint8_t* arena) {
    if (!ops.n) return;
    size_t most = 0;
    for (uint32_t i = 0; i < ops.n; ++i) most = most > ops.op[i].bytes ? most : (size_t)ops.op[i].bytes;
    size_t grid = (most / 16 + 255) / 256;
    grid = grid < 1 ? 1 : (grid > 2048 ? 2048 : grid);
    AFQ_LAUNCH(k_range_init, (uint32_t)grid, 256, s, ops, arena);
}

// The handful of per-cell words the host reads when a range is done (status block, flags, row lengths, barcodes), packed into one
// buffer at the end of the range's kernels and copied out by ONE async D2H into pinned memory (it was four synchronous copies).
// Layout (u32 words): [0..7] DevStatus, [8] the EM's "scratch too small" flag, [16, 16+n) alt, then n nnz, n EM nnz, 2n barcode.
__global__ __launch_bounds__(256) void k_pack_small(const DevStatus* __restrict__ st, const uint32_t* __restrict__ em_flag, const uint32_t* __restrict__ alt,
                                                    const uint32_t* __restrict__ nnz, const uint32_t* __restrict__ em_nnz,
                                                    const uint64_t* __restrict__ bc, const uint32_t* __restrict__ n_mono, uint32_t n, uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < sizeof(DevStatus) / 4) out[i] = reinterpret_cast<const uint32_t*>(st)[i];
    if (i == 8) out[8] = em_flag ? *em_flag : 0u;
    if (i == 9) out[9] = n_mono ? *n_mono : 0u;
    if (i < n) {
        out[16 + i] = alt[i];
        out[16 + n + i] = nnz[i];
        out[16 + 2 * (size_t)n + i] = em_nnz ? em_nnz[i] : 0u;
        const uint64_t b = bc[i];
        out[16 + 3 * (size_t)n + 2 * (size_t)i] = (uint32_t)b;
        out[16 + 3 * (size_t)n + 2 * (size_t)i + 1] = (uint32_t)(b >> 32);
    }
}
void launch_pack_small(hipStream_t s, const DevStatus* st, const uint32_t* em_flag, const uint32_t* alt, const uint32_t* nnz, const uint32_t* em_nnz,
                       const uint64_t* bc, const uint32_t* n_mono, uint32_t n, uint32_t* out) {
    AFQ_LAUNCH(k_pack_small, (std::max(n, 16u) + 255) / 256, 256, s, st, em_flag, alt, nnz, em_nnz, bc, n_mono, n, out);
}

// Three small device arrays into (host-mapped, pinned) memory by a kernel: what the host reads between two ranges of a pipelined
// batch.  As async D2H copies they queue on a DMA engine - behind the previous range's gigabyte of results when the runtime has
// given both streams the same engine, which it does or does not depending on what the process did before (seen in the ATAC leg:
// 38 ms per step on its own, 66 ms behind the other legs).
__global__ __launch_bounds__(256) void k_copy_words3(const uint32_t* __restrict__ a, uint32_t na, uint32_t* __restrict__ da,
                                                     const uint32_t* __restrict__ b, uint32_t nb, uint32_t* __restrict__ db,
                                                     const uint32_t* __restrict__ c, uint32_t nc, uint32_t* __restrict__ dc) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < na; i += gridDim.x * 256) da[i] = a[i];
    if (blockIdx.x == 0) {
        for (uint32_t i = threadIdx.x; i < nb; i += 256) db[i] = b[i];
        for (uint32_t i = threadIdx.x; i < nc; i += 256) dc[i] = c[i];
    }
}
void launch_copy_words3(hipStream_t s, const uint32_t* a, uint32_t na, uint32_t* da, const uint32_t* b, uint32_t nb, uint32_t* db,
                        const uint32_t* c, uint32_t nc, uint32_t* dc) {
    AFQ_LAUNCH(k_copy_words3, std::max(1u, std::min(64u, (na + 255) / 256)), 256, s, a, na, da, b, nb, db, c, nc, dc);
}

void launch_gather_headers(hipStream_t s, const uint8_t* bytes, size_t n_bytes, const uint64_t* chunk_off,
                           uint32_t n_cells, uint32_t* hdr) {
    if (!n_cells) return;
    AFQ_LAUNCH(k_gather_headers, (n_cells + 255) / 256, 256, s, bytes, n_bytes, chunk_off, n_cells, hdr);
}

template <int BW, int UW>
static void launch_decode_t(hipStream_t s, const DecodeArgs& a) {
    const uint32_t grid = a.chk ? (a.n_cells + 255) / 256 : (a.n_cells + 3) / 4;   // (fix-up mode: a workgroup verifies 256 cells' proofs and re-decodes the ones that failed)
    AFQ_LAUNCH((k_decode<BW, UW>), grid, 256, s, a.bytes, a.n_bytes, a.meta, a.n_cells, a.t2g,
               a.ref_count, a.num_genes, a.keys0, a.cell_nkeys, a.bc_out, a.st, a.chk, a.pug);
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

// k_decode_keys finds a record's first mention of a gene through an LDS hash table (HD); AFQ_TEST_DECODE_DEDUP=scan: by the look-back
// compares and the serial scan of rounds 2-4 (read per launch: tests run both).  Label-tail workload (configs1_tail, 9.66 GB):
// 14.54 -> 9.01 ms per step, the step 35.0 -> 29.4 ms (profiles/history/run_r04ah.sh).
static bool decode_keys_hash_dedup() { return !test_hook_is("DECODE_DEDUP", "scan"); }
template <int BW, int UW>
static void launch_decode_par_t(hipStream_t s, const DecodeArgs& a) {
    AFQ_LAUNCH((k_slab_setup<BW, UW>), (a.n_cells + 3) / 4, 256, s, a.bytes, a.meta, a.n_cells, a.slab_prefix,
               a.slab_cell, a.cell_bc);
    const uint32_t n_groups = (a.n_slabs + kSlabsPerWave - 1) / kSlabsPerWave;
    const uint32_t n_cols = n_groups < kDecodeCols ? n_groups : kDecodeCols;
    const uint32_t n_waves = n_cols * ((n_groups + n_cols - 1) / n_cols);
    if (a.pug.h && a.short_records && !a.trivial)  // the batch has PUG cells: instances that also emit (label key, umi, offset) per read
        AFQ_LAUNCH((k_decode_recs<BW, UW, false, true>), (n_waves + 3) / 4, 256, s, a.bytes, a.meta, a.n_cells, a.slab_prefix, a.slab_cell,
                   a.cell_bc, a.n_slabs, a.t2g, a.ref_count, a.num_genes, a.keys0, a.cell_nkeys, a.bc_out,
                   const_cast<CellChk*>(a.chk), a.pug);
    else if (a.pug.h)
        AFQ_LAUNCH((k_decode_par<BW, UW, true>), (n_waves + 3) / 4, 256, s, a.bytes, a.meta, a.n_cells, a.slab_prefix, a.slab_cell,
                   a.cell_bc, a.n_slabs, a.t2g, a.ref_count, a.num_genes, a.keys0, a.cell_nkeys, a.bc_out,
                   const_cast<CellChk*>(a.chk), a.pug);
    else if (a.short_records && a.trivial)
        AFQ_LAUNCH((k_decode_recs<BW, UW, true, false>), (n_waves + 3) / 4, 256, s, a.bytes, a.meta, a.n_cells, a.slab_prefix, a.slab_cell,
                   a.cell_bc, a.n_slabs, a.t2g, a.ref_count, a.num_genes, a.keys0, a.cell_nkeys, a.bc_out,
                   const_cast<CellChk*>(a.chk), a.pug);
    else if (a.short_records)
        AFQ_LAUNCH((k_decode_recs<BW, UW, false, false>), (n_waves + 3) / 4, 256, s, a.bytes, a.meta, a.n_cells, a.slab_prefix, a.slab_cell,
                   a.cell_bc, a.n_slabs, a.t2g, a.ref_count, a.num_genes, a.keys0, a.cell_nkeys, a.bc_out,
                   const_cast<CellChk*>(a.chk), a.pug);
    else if (a.trivial)
        AFQ_LAUNCH((k_decode_keys<BW, UW, true>), (n_waves + 3) / 4, 256, s, a.bytes, a.meta, a.n_cells, a.slab_prefix, a.slab_cell,
                   a.cell_bc, a.n_slabs, a.t2g, a.ref_count, a.num_genes, a.keys0, a.cell_nkeys, a.bc_out,
                   const_cast<CellChk*>(a.chk));
    else if (decode_keys_hash_dedup())
        AFQ_LAUNCH((k_decode_keys<BW, UW, false, true>), (n_waves + 3) / 4, 256, s, a.bytes, a.meta, a.n_cells, a.slab_prefix, a.slab_cell,
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

}  // namespace afq
