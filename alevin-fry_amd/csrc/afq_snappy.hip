// afq_snappy.hip - snappy frame chunks undone on the device (gfx950, wave64).
//
// The reference opens map.collated.rad.sz through snap::read::FrameDecoder (src/quant.rs:373-395; collate.rs:550-554 writes
// it).  The data chunks of the frame format are independent - at most 65 536 bytes of output each - so a stream is a list of
// small, equal jobs: ONE WAVE PER CHUNK, the chunk's output block in LDS (64 KiB: two chunks to a CU), the element tags read
// by all lanes together (the same byte for every lane: the control flow is the wave's, not a lane's), literals and copies
// executed by the lanes side by side - a copy whose source overlaps its own output (offset < length, snappy's run-length
// form) is byte i <- byte (i mod offset) of the pattern, which lanes can do independently - and the finished block written
// out in dwords.  Same acceptance rules as the host decoder (afq_host.cpp: snappy_raw_decompress_into): anything malformed is
// an error, nothing is guessed.  The chunks' CRC-32C words are NOT checked here (the host decoder checks them).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "afq_common.h"
#include "afq_kernels.h"
#include "afq_prims.h"

namespace afq {

constexpr uint32_t kSzBlock = 65536;   // output bytes of a chunk, at most (the frame format's limit)
constexpr uint32_t kSzWin = 8192;     // input bytes held in LDS for the tag parse

#define SZ_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

__global__ __launch_bounds__(64) void k_snappy_frames(const uint8_t* __restrict__ comp, const SzFrame* __restrict__ frames, uint32_t n_frames,
                                                     uint8_t* __restrict__ out, DevStatus* __restrict__ st) {
    extern __shared__ uint8_t s_out[];   // kSzBlock bytes of output, then a kSzWin-byte window of the input
    const uint32_t f = blockIdx.x, lane = threadIdx.x;
    if (f >= n_frames) return;
    const SzFrame fr = frames[f];
    const uint8_t* inb = comp + fr.in_off;
    uint8_t* dst = out + fr.out_off;
    const uint32_t n = (uint32_t)fr.in_len, ulen = fr.ulen;
    // The element tags are read out of a window of the input in LDS: one element after the other, a tag out of global memory
    // is a round trip of its own (measured with the tags read from global memory: 9 GB/s of output on a collated RAD).
    uint8_t* s_in = s_out + kSzBlock;
    uint32_t wbase = 0, wend = 0;   // the window holds input bytes [wbase, wend)
    auto window = [&](uint32_t from) {   // wave-wide: refill from `from`
        SZ_WAVE_SYNC();
        wbase = from;
        wend = min(n, from + kSzWin);
        for (uint32_t i = lane; wbase + i < wend; i += 64) s_in[i] = inb[wbase + i];
        SZ_WAVE_SYNC();
    };
    // control byte i (every lane reads the same one; through readfirstlane the compiler knows it, and the parse is scalar
    // control flow); the callers have made sure [i, i + 5) or the rest of the input is in the window
    auto in = [&](uint32_t i) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)s_in[i - wbase]); };
    if (!fr.compressed) {   // an uncompressed chunk: its bytes are the data
        for (uint32_t i = lane; i < ulen; i += 64) dst[i] = inb[i];
        return;
    }
    bool bad = ulen > kSzBlock;
    // the block starts with its uncompressed length (uvarint): it must be the one the plan was made with
    uint32_t p = 0, shift = 0;
    uint64_t declared = 0;
    if (!bad) window(0);   // (the length prefix is at most five bytes)
    while (!bad) {
        if (p >= n || shift > 35) { bad = true; break; }
        const uint32_t b = in(p++);
        declared |= (uint64_t)(b & 0x7Fu) << shift;
        if (!(b & 0x80u)) break;
        shift += 7;
    }
    bad = bad || declared != ulen;
    uint32_t w = 0;   // bytes of output so far
    while (!bad && p < n) {
        if (p < wbase || p + 5 > wend) { if (wend < n || p < wbase) window(p); }   // a tag and the (up to four) bytes behind it
        const uint32_t tag = in(p++), type = tag & 3u;
        if (type == 0) {   // literal
            uint64_t len = (tag >> 2) + 1;
            if (len > 60) {
                const uint32_t nb = (uint32_t)len - 60;
                if (p + nb > n) { bad = true; break; }
                len = 0;
                for (uint32_t i = 0; i < nb; ++i) len |= (uint64_t)in(p + i) << (8 * i);
                len += 1;
                p += nb;
            }
            if (len > kSzBlock || p + len > n || w + len > ulen) { bad = true; break; }
            if (p >= wbase && p + (uint32_t)len <= wend) {   // a literal inside the window (nearly all are short): LDS to LDS, no trip to memory
                for (uint32_t i = lane; i < (uint32_t)len; i += 64) s_out[w + i] = s_in[p - wbase + i];
            } else {
                for (uint32_t i = lane; i < (uint32_t)len; i += 64) s_out[w + i] = inb[p + i];
            }
            p += (uint32_t)len; w += (uint32_t)len;
        } else {           // copy of earlier output
            uint32_t len, off;
            if (type == 1) { if (p + 1 > n) { bad = true; break; } len = ((tag >> 2) & 7u) + 4; off = ((tag >> 5) << 8) | in(p); p += 1; }
            else if (type == 2) { if (p + 2 > n) { bad = true; break; } len = (tag >> 2) + 1; off = in(p) | (in(p + 1) << 8); p += 2; }
            else { if (p + 4 > n) { bad = true; break; } len = (tag >> 2) + 1; off = in(p) | (in(p + 1) << 8) | (in(p + 2) << 16) | (in(p + 3) << 24); p += 4; }
            if (off == 0 || off > w || w + len > ulen) { bad = true; break; }
            SZ_WAVE_SYNC();   // (the bytes the copy reads were written by the elements before it)
            if (lane < len) s_out[w + lane] = s_out[w - off + (off >= len ? lane : lane % off)];   // (len <= 64: one byte per lane)
            w += len;
        }
        SZ_WAVE_SYNC();
    }
    if (bad || w != ulen) { if (lane == 0) set_err(st, kErrSnappy, f); return; }
    SZ_WAVE_SYNC();
    // the block out: bytes up to the first dword boundary of the destination, dwords, the last bytes
    const uint32_t head = min(ulen, (uint32_t)((4u - (uint32_t)((uintptr_t)dst & 3u)) & 3u));
    if (lane < head) dst[lane] = s_out[lane];
    const uint32_t nd = (ulen - head) >> 2;
    uint32_t* d4 = reinterpret_cast<uint32_t*>(dst + head);
    for (uint32_t k = lane; k < nd; k += 64) {
        const uint32_t q = head + 4 * k;
        d4[k] = (uint32_t)s_out[q] | ((uint32_t)s_out[q + 1] << 8) | ((uint32_t)s_out[q + 2] << 16) | ((uint32_t)s_out[q + 3] << 24);
    }
    const uint32_t tail0 = head + 4 * nd;
    if (tail0 + lane < ulen) dst[tail0 + lane] = s_out[tail0 + lane];   // (at most three bytes)
}

void launch_snappy_frames(hipStream_t s, const uint8_t* comp, const SzFrame* frames, uint32_t n_frames, uint8_t* out, DevStatus* st) {
    if (!n_frames) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_snappy_frames), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kSzBlock + kSzWin));   // (64 KiB and above needs asking)
    hipLaunchKernelGGL(k_snappy_frames, dim3(n_frames), dim3(64), kSzBlock + kSzWin, s, comp, frames, n_frames, out, st);
}

}  // namespace afq
