// afq_atac.hip — scATAC records straight from collated-RAD chunks (BASELINE configs[4]).
//
// What the worker loop of `alevin-fry atac deduplicate` does with a chunk before it sorts (src/atac/deduplicate.rs:199-218 of
// the reference): walk the AtacSeqReadRecords, keep those with exactly one alignment of map_type 4 as (chr, start, frag_len),
// count the ones with more than one alignment and the ones that are not one properly mapped pair.  A record on the wire is
//   na:u32, bc:(u8|u16|u32|u64), na x { ref:u32, type:u8, start_pos:u32, frag_len:u16 }
// (read tag b, alignment tags ref/type/start_pos/frag_len in that order: tests/atac_integration.rs:110-121), i.e. 4 + bc_bytes
// + 11*na bytes: nothing is dword-aligned and a record's length is its own na, so the stream has to be walked - or proven:
//
// Walk-free parse, at byte granularity (the scheme of the RNA decoders): in a collated chunk every record
// carries the cell's barcode, so byte p is a CANDIDATE record start iff the barcode field at p+4 equals the barcode of the
// chunk's first record and its na fits the rest of the chunk.  Let succ(p) = p + 4 + bc_bytes + 11*na(p).  If (1) byte 8 is a
// candidate, (2) every candidate's succ is a candidate or the chunk's end, (3) the candidates number nrec and (4) their sizes
// sum to nbytes-8, then the candidates ARE the sequential parse (the chain from 8 tiles the chunk by (1)+(2); a candidate
// off the chain would add size, contradicting (4)).  One workgroup per cell: pass A ballots the candidate bits into a bitmap,
// pass B checks (2), accumulates (3)/(4) and emits the kept fragments.  A cell whose proof fails (a start position or
// fragment length that happens to spell the barcode, or a corrupt chunk) is re-read by k_atac_walk, one thread walking it
// record by record - the same result, just slowly, and the place where malformed input gets its error code.
#include <hip/hip_runtime.h>

#include "afq_common.h"
#include "afq_kernels.h"
#include "afq_prims.h"

namespace afq {

namespace {

constexpr int kParseNT = 256;

__device__ __forceinline__ uint32_t ld32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint32_t ld16(const uint8_t* p) { uint16_t v; __builtin_memcpy(&v, p, 2); return v; }
__device__ __forceinline__ uint64_t ldbc(const uint8_t* p, uint32_t bcb) {
    if (bcb == 4) return ld32(p);
    if (bcb == 8) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
    if (bcb == 2) return ld16(p);
    return *p;
}

__global__ __launch_bounds__(kParseNT) void k_atac_parse(AtacParseArgs a) {
    __shared__ uint32_t s_red[3][kParseNT / 64];
    __shared__ uint32_t s_cnt, s_multi, s_non;
    const uint32_t cell = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const AtacCell c = a.cells[cell];
    const uint8_t* ch = a.bytes + c.chunk_off;
    const uint32_t nb = c.nbytes, H = 4 + a.bc_bytes;
    uint64_t* bm = a.bitmap + c.bm_off;
    if (tid == 0) { s_cnt = 0; s_multi = 0; s_non = 0; }
    const bool has_first = nb >= 8 + H;
    const uint64_t bc0 = has_first ? ldbc(ch + 12, a.bc_bytes) : 0;
    // ---- pass A: candidate bits (four tiles of 256 positions per trip: the loads of all four are in flight together -
    // one position per thread and trip left the loop waiting out one memory round trip per 256 bytes)
    for (uint32_t base = 0; base < nb; base += 4 * kParseNT) {
        uint64_t bcv[4];
        uint32_t nav[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t p = base + j * kParseNT + tid;
            const bool in = p >= 8 && p + H <= nb;
            bcv[j] = in ? ldbc(ch + p + 4, a.bc_bytes) : ~bc0;
            nav[j] = in ? ld32(ch + p) : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t p = base + j * kParseNT + tid;
            const bool cand = p >= 8 && p + H <= nb && bcv[j] == bc0 && nav[j] <= (nb - p - H) / 11u;
            const uint64_t m = __ballot(cand);
            if (lane == 0 && base + j * kParseNT < nb) bm[p >> 6] = m;
        }
    }
    __threadfence_block();
    __syncthreads();
    // ---- pass B: successor check, counts, kept fragments
    uint32_t n_cand = 0, sum = 0;
    bool fail = false;
    uint32_t* o_ref = a.o_ref + c.out_off;
    uint32_t* o_start = a.o_start + c.out_off;
    uint16_t* o_flen = a.o_flen + c.out_off;
    for (uint32_t base = 0; base < nb; base += 4 * kParseNT) {   // four tiles per trip again: bitmap words, then the record heads, then the successors
        bool cand[4], keep[4];
        uint32_t na[4], ty[4];
        uint64_t sw[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t tb = base + j * kParseNT;
            const uint64_t w = tb < nb ? bm[(tb >> 6) + wave] : 0ull;   // one word per wave
            cand[j] = (w >> lane) & 1ull;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t p = base + j * kParseNT + tid;
            na[j] = cand[j] ? ld32(ch + p) : 0u;
            ty[j] = cand[j] && p + H + 4 < nb ? ch[p + H + 4] : 0u;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t p = base + j * kParseNT + tid;
            const uint32_t s = p + H + 11u * na[j];
            sw[j] = cand[j] && s + H <= nb ? bm[s >> 6] : 0ull;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t p = base + j * kParseNT + tid;
            keep[j] = false;
            if (cand[j]) {
                const uint32_t sz = H + 11u * na[j], s = p + sz;
                const bool ok = s == nb || (s + H <= nb && ((sw[j] >> (s & 63u)) & 1ull));
                fail = fail || !ok;
                ++n_cand; sum += sz;
                keep[j] = na[j] == 1 && ty[j] == 4;
            }
            const uint64_t km = __ballot(keep[j]);
            const uint64_t mm = __ballot(cand[j] && na[j] > 1), nm = __ballot(cand[j] && !keep[j] && na[j] <= 1);
            uint32_t slot0 = 0;
            if (lane == 0) {
                if (km) slot0 = atomicAdd(&s_cnt, (uint32_t)__popcll(km));
                if (mm) atomicAdd(&s_multi, (uint32_t)__popcll(mm));
                if (nm) atomicAdd(&s_non, (uint32_t)__popcll(nm));
            }
            slot0 = __shfl(slot0, 0);
            if (keep[j]) {
                const uint32_t slot = slot0 + (uint32_t)__popcll(km & ((1ull << lane) - 1ull));
                if (slot < c.nrec) {
                    o_ref[slot] = ld32(ch + p + H);
                    o_start[slot] = ld32(ch + p + H + 5);
                    o_flen[slot] = (uint16_t)ld16(ch + p + H + 9);
                } else fail = true;   // more candidates than records: the proof cannot hold
            }
        }
    }
    // block reduction of (count, size sum, fail)
    uint32_t f = fail ? 1u : 0u;
    for (int d = 32; d; d >>= 1) { n_cand += __shfl_xor(n_cand, d); sum += __shfl_xor(sum, d); f |= __shfl_xor(f, d); }
    if (lane == 0) { s_red[0][wave] = n_cand; s_red[1][wave] = sum; s_red[2][wave] = f; }
    __syncthreads();
    if (tid == 0) {
        uint32_t cnt = 0, sm = 0, fl = 0;
        for (int w2 = 0; w2 < kParseNT / 64; ++w2) { cnt += s_red[0][w2]; sm += s_red[1][w2]; fl |= s_red[2][w2]; }
        const bool ok = has_first && !fl && cnt == c.nrec && sm == nb - 8 && ((bm[0] >> 8) & 1ull);
        a.cell_bc[cell] = bc0;
        if (ok) {
            a.cell_cnt[cell] = s_cnt; a.cell_stat[2 * cell] = s_multi; a.cell_stat[2 * cell + 1] = s_non;
        } else {
            a.cell_cnt[cell] = 0; a.cell_stat[2 * cell] = 0; a.cell_stat[2 * cell + 1] = 0;
            a.walk_list[atomicAdd(a.n_walk, 1u)] = cell;
        }
    }
}

// The sequential parse of the cells the proof did not cover: the record loop of deduplicate.rs:199-218 as it stands.
__global__ __launch_bounds__(64) void k_atac_walk(AtacParseArgs a) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= *a.n_walk) return;
    const uint32_t cell = a.walk_list[i];
    const AtacCell c = a.cells[cell];
    const uint8_t* ch = a.bytes + c.chunk_off;
    const uint32_t nb = c.nbytes, H = 4 + a.bc_bytes;
    uint32_t p = 8, kept = 0, multi = 0, non = 0;
    bool bad = false;
    for (uint32_t r = 0; r < c.nrec; ++r) {
        if (p + H > nb) { bad = true; break; }
        const uint32_t na = ld32(ch + p);
        if (na > (nb - p - H) / 11u) { bad = true; break; }
        if (na == 1 && ch[p + H + 4] == 4) {
            a.o_ref[c.out_off + kept] = ld32(ch + p + H);
            a.o_start[c.out_off + kept] = ld32(ch + p + H + 5);
            a.o_flen[c.out_off + kept] = (uint16_t)ld16(ch + p + H + 9);
            ++kept;
        } else if (na > 1) ++multi;
        else ++non;
        p += H + 11u * na;
    }
    if (bad || p != nb) {
        if (atomicCAS(&a.st->err_code, 0u, kErrRecordWalk) == 0u) a.st->err_cell = cell;
        kept = 0;
    }
    atomicAdd(&a.st->n_fallback, 1u);
    a.cell_cnt[cell] = kept; a.cell_stat[2 * cell] = multi; a.cell_stat[2 * cell + 1] = non;
}

}  // namespace

void launch_atac_parse(hipStream_t s, const AtacParseArgs& a) {
    if (!a.n_cells) return;
    AFQ_LAUNCH(k_atac_parse, a.n_cells, kParseNT, s, a);
    AFQ_LAUNCH(k_atac_walk, (a.n_cells + 63) / 64, 64, s, a);
}

}  // namespace afq
