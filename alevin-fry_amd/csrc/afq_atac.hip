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

// Pass A reads the chunk as ALIGNED dwords: a thread takes four consecutive byte positions q0 .. q0+3 (q = byte offset from
// the dword boundary at or below the chunk's first byte) and builds the na and barcode words of each from four loaded
// dwords with funnel shifts - a quarter of the load instructions of one unaligned pair per position, and every byte of the
// chunk fetched once per wave instead of eight times.  The candidate bits of a group of 256 positions are four ballots;
// they are stored AS the ballots (word 4g + k holds positions 256g + 4*lane + k), and pass B addresses them that way.
// Pass B gives every lane one bitmap word: the lane walks its ~3 set bits (na, successor bit, fragment fields), so a wave
// has 64 record chains in flight where a lane per position had three or four.
__device__ __forceinline__ uint32_t bm_word_of(uint32_t q) { return ((q >> 8) << 2) + (q & 3u); }
__device__ __forceinline__ uint32_t bm_bit_of(uint32_t q) { return (q >> 2) & 63u; }

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
    const uint32_t bc0_lo = (uint32_t)bc0, bc0_hi = (uint32_t)(bc0 >> 32);
    const uint32_t lo_mask = a.bc_bytes >= 4 ? 0xFFFFFFFFu : (1u << (8 * a.bc_bytes)) - 1u;
    const uint32_t al = (uint32_t)((uintptr_t)ch & 3u);
    const uint32_t* W = reinterpret_cast<const uint32_t*>(ch - al);
    const uint64_t w_lim = (a.n_bytes - (c.chunk_off - al)) >> 2;   // whole dwords of the input buffer from W on
    const uint32_t w_tail = (uint32_t)((a.n_bytes - (c.chunk_off - al)) & 3u);   // ... and the bytes of a last, partial one
    // dword i of the chunk's aligned view; the buffer's last dword may be partial (a record of 4 + bc + 11 na bytes ends
    // anywhere): its bytes are loaded one by one - read as 0 they made a last record with na = 0 fail the proof, and its
    // cell took the sequential walk for nothing
    auto ldw = [&](uint64_t i) -> uint32_t {
        if (i < w_lim) return W[i];
        if (i > w_lim || !w_tail) return 0u;
        const uint8_t* b = reinterpret_cast<const uint8_t*>(W + i);
        uint32_t v = b[0];
        if (w_tail > 1) v |= (uint32_t)b[1] << 8;
        if (w_tail > 2) v |= (uint32_t)b[2] << 16;
        return v;
    };
    const uint32_t nq = nb + al;                                     // positions q in [al, nq)
    const uint32_t n_groups = (nq + 255) >> 8;
    // ---- pass A: candidate bits, a wave per group of 256 positions, two groups per trip
    for (uint32_t g0 = wave * 2; g0 < n_groups; g0 += (kParseNT / 64) * 2) {
        uint32_t d[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const uint64_t D = (uint64_t)(g0 + u) * 64 + lane;
#pragma unroll
            for (int x = 0; x < 4; ++x) d[u][x] = (g0 + u < n_groups && (x < 3 || a.bc_bytes == 8)) ? ldw(D + x) : 0u;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (g0 + u >= n_groups) break;
            const uint32_t q0 = (g0 + u) * 256 + 4 * lane;
            uint64_t m[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t q = q0 + k;
                const uint32_t na = k ? __builtin_amdgcn_alignbyte(d[u][1], d[u][0], k) : d[u][0];
                const uint32_t bl = k ? __builtin_amdgcn_alignbyte(d[u][2], d[u][1], k) : d[u][1];
                const uint32_t bh = k ? __builtin_amdgcn_alignbyte(d[u][3], d[u][2], k) : d[u][2];
                const uint32_t p = q - al;
                bool cand = q >= al + 8 && p + H <= nb && (bl & lo_mask) == bc0_lo;
                if (a.bc_bytes == 8) cand = cand && bh == bc0_hi;
                cand = cand && na <= (nb - p - H) / 11u;
                m[k] = __ballot(cand);
            }
            if (lane < 4) bm[(uint64_t)(g0 + u) * 4 + lane] = lane == 0 ? m[0] : lane == 1 ? m[1] : lane == 2 ? m[2] : m[3];
        }
    }
    __threadfence_block();
    __syncthreads();
    // ---- pass B: a bitmap word per lane: successor check, counts, kept fragments
    uint32_t n_cand = 0, sum = 0;
    bool fail = false;
    uint32_t* o_ref = a.o_ref + c.out_off;
    uint32_t* o_start = a.o_start + c.out_off;
    uint16_t* o_flen = a.o_flen + c.out_off;
    const uint32_t n_words = n_groups * 4;
    uint32_t multi = 0, non = 0;
    for (uint32_t wi = tid; wi < n_words; wi += kParseNT) {
        uint64_t w = bm[wi];
        const uint32_t qbase = (wi >> 2) * 256 + (wi & 3u);
        const uint32_t nc = (uint32_t)__popcll(w);
        n_cand += nc;
        // first the record heads of up to four candidates (independent loads), then their successors
        uint32_t kept = 0;
        uint32_t kp[4];   // positions of this word's kept records (more than four: a second round)
        while (w) {
            uint32_t pp[4], nn[4], tt[4];
            int n = 0;
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                pp[x] = 0; nn[x] = 0; tt[x] = 0;
                if (w) { const uint32_t b = (uint32_t)__builtin_ctzll(w); w &= w - 1; pp[x] = qbase + 4 * b - al; n = x + 1; }
            }
#pragma unroll
            for (int x = 0; x < 4; ++x) if (x < n) { nn[x] = ld32(ch + pp[x]); tt[x] = pp[x] + H + 4 < nb ? ch[pp[x] + H + 4] : 0u; }
            uint64_t sw[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const uint32_t s = pp[x] + H + 11u * nn[x];
                sw[x] = (x < n && s + H <= nb) ? bm[bm_word_of(s + al)] : 0ull;
            }
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                if (x >= n) continue;
                const uint32_t sz = H + 11u * nn[x], s = pp[x] + sz;
                const bool ok = s == nb || (s + H <= nb && ((sw[x] >> bm_bit_of(s + al)) & 1ull));
                fail = fail || !ok;
                sum += sz;
                const bool keep = nn[x] == 1 && tt[x] == 4;
                if (keep) {
                    if (kept == 4) {   // flush the four held back
                        const uint32_t slot0 = atomicAdd(&s_cnt, 4u);
#pragma unroll
                        for (int y = 0; y < 4; ++y) {
                            if (slot0 + y < c.nrec) { o_ref[slot0 + y] = ld32(ch + kp[y] + H); o_start[slot0 + y] = ld32(ch + kp[y] + H + 5); o_flen[slot0 + y] = (uint16_t)ld16(ch + kp[y] + H + 9); }
                            else fail = true;
                        }
                        kept = 0;
                    }
#pragma unroll
                    for (int y = 0; y < 4; ++y) if ((uint32_t)y == kept) kp[y] = pp[x];
                    ++kept;
                } else if (nn[x] > 1) ++multi;
                else ++non;
            }
        }
        if (kept) {
            const uint32_t slot0 = atomicAdd(&s_cnt, kept);
#pragma unroll
            for (int y = 0; y < 4; ++y) {
                if ((uint32_t)y >= kept) break;
                if (slot0 + y < c.nrec) { o_ref[slot0 + y] = ld32(ch + kp[y] + H); o_start[slot0 + y] = ld32(ch + kp[y] + H + 5); o_flen[slot0 + y] = (uint16_t)ld16(ch + kp[y] + H + 9); }
                else fail = true;   // more candidates than records: the proof cannot hold
            }
        }
    }
    // block reduction of (count, size sum, fail) and of the two tallies
    uint32_t f = fail ? 1u : 0u;
    for (int dd = 32; dd; dd >>= 1) { n_cand += __shfl_xor(n_cand, dd); sum += __shfl_xor(sum, dd); f |= __shfl_xor(f, dd); multi += __shfl_xor(multi, dd); non += __shfl_xor(non, dd); }
    if (lane == 0) { s_red[0][wave] = n_cand; s_red[1][wave] = sum; s_red[2][wave] = f; if (multi) atomicAdd(&s_multi, multi); if (non) atomicAdd(&s_non, non); }
    __syncthreads();
    if (tid == 0) {
        uint32_t cnt = 0, sm = 0, fl = 0;
        for (int w2 = 0; w2 < kParseNT / 64; ++w2) { cnt += s_red[0][w2]; sm += s_red[1][w2]; fl |= s_red[2][w2]; }
        const bool first_ok = has_first && ((bm[bm_word_of(8 + al)] >> bm_bit_of(8 + al)) & 1ull);
        const bool ok = first_ok && !fl && cnt == c.nrec && sm == nb - 8;
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
