// afq_prims.h — wave/block primitives shared by the gfx950 kernel files (device code only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "afq_common.h"

namespace afq {

// ---------------------------------------------------------------------------
// wave / block primitives (wave = 64 lanes)
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t& total) {
    uint32_t x = v;
    const uint32_t lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t y = __shfl_up(x, d);
        if (lane >= (uint32_t)d) x += y;
    }
    total = __shfl(x, 63);
    return x - v;
}

// exclusive scan over the NT threads of a block; ws needs NT/64 words of LDS.
template <int NT>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* ws, uint32_t& total) {
    constexpr int NW = NT / 64;
    uint32_t wtot;
    uint32_t ex = wave_excl_scan(v, wtot);
    const uint32_t w = threadIdx.x >> 6;
    __syncthreads();  // ws may still be read from a previous call
    if (lane_id() == 63) ws[w] = wtot;
    __syncthreads();
    uint32_t pre = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        uint32_t t = ws[i];
        if ((uint32_t)i < w) pre += t;
        tot += t;
    }
    total = tot;
    return pre + ex;
}

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
    // (mov_dpp = update_dpp without an old value: every lane is written, so no copy of x is made in front of the DPP move)
    if constexpr (J == 1) return __builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, false);
    else if constexpr (J == 2) return __builtin_amdgcn_mov_dpp(x, 0x4E, 0xF, 0xF, false);
    else if constexpr (J == 4) {
        uint32_t t = __builtin_amdgcn_mov_dpp(x, 0x104, 0xF, 0x5, false);        // row_shl:4 into banks 0,2
        return __builtin_amdgcn_update_dpp(t, x, 0x114, 0xF, 0xA, false);          // row_shr:4 into banks 1,3
    } else if constexpr (J == 8) {
        uint32_t t = __builtin_amdgcn_mov_dpp(x, 0x108, 0xF, 0x3, false);        // row_shl:8 into banks 0,1
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
__device__ __forceinline__ T xor_lane(T v) {   // unsigned integers of 4, 8 or 16 bytes (scalars stay in registers; a struct would not)
    if constexpr (sizeof(T) == 16) {
        const uint64_t lo = (uint64_t)v, hi = (uint64_t)(v >> 64);
        const uint32_t w0 = xor_lane32<J>((uint32_t)lo), w1 = xor_lane32<J>((uint32_t)(lo >> 32));
        const uint32_t w2 = xor_lane32<J>((uint32_t)hi), w3 = xor_lane32<J>((uint32_t)(hi >> 32));
        return ((T)(((uint64_t)w3 << 32) | w2) << 64) | (T)(((uint64_t)w1 << 32) | w0);
    } else if constexpr (sizeof(T) == 8) {
        const uint32_t lo = xor_lane32<J>((uint32_t)v), hi = xor_lane32<J>((uint32_t)((uint64_t)v >> 32));
        return (T)(((uint64_t)hi << 32) | lo);
    } else return (T)xor_lane32<J>((uint32_t)v);
}

// a < b.  For 16-byte keys spelled out on the halves with bitwise connectives, so the three compares meet as lane masks
// on the scalar unit (the compiler's own lowering of the 128-bit compare builds 0/1 values in vector registers: 8
// vector instructions where this is 3).
template <typename T>
__device__ __forceinline__ bool key_lt(T a, T b) {
    if constexpr (sizeof(T) == 16) {
        const uint64_t al = (uint64_t)a, ah = (uint64_t)(a >> 64), bl = (uint64_t)b, bh = (uint64_t)(b >> 64);
        return (ah < bh) | ((ah == bh) & (al < bl));
    } else return a < b;
}

// v_permlane16_swap / v_permlane32_swap between TWO registers: the odd rows (J = 16) / the upper half (J = 32) of x
// change places with the even rows / the lower half of y.  Applied to elements h and h + 1 of a lane this puts every
// position and its partner at distance J into the same lane - the exchange at that distance becomes a register swap
// with no redundant compare (each lane decides a different pair), and the same instruction undoes the move.
template <int J>
__device__ __forceinline__ void swap_rows32(uint32_t& x, uint32_t& y) {
    static_assert(J == 16 || J == 32, "row / half swaps only");
    v2u_t p;
    if constexpr (J == 16) p = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    else p = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    x = p.x; y = p.y;
}
template <int J, typename T>
__device__ __forceinline__ void swap_rows(T& x, T& y) {
    if constexpr (sizeof(T) == 16) {
        uint32_t xw[4] = {(uint32_t)x, (uint32_t)(x >> 32), (uint32_t)(x >> 64), (uint32_t)(x >> 96)};
        uint32_t yw[4] = {(uint32_t)y, (uint32_t)(y >> 32), (uint32_t)(y >> 64), (uint32_t)(y >> 96)};
#pragma unroll
        for (int w = 0; w < 4; ++w) swap_rows32<J>(xw[w], yw[w]);
        x = ((T)(((uint64_t)xw[3] << 32) | xw[2]) << 64) | (T)(((uint64_t)xw[1] << 32) | xw[0]);
        y = ((T)(((uint64_t)yw[3] << 32) | yw[2]) << 64) | (T)(((uint64_t)yw[1] << 32) | yw[0]);
    } else if constexpr (sizeof(T) == 8) {
        uint32_t x0 = (uint32_t)x, x1 = (uint32_t)((uint64_t)x >> 32), y0 = (uint32_t)y, y1 = (uint32_t)((uint64_t)y >> 32);
        swap_rows32<J>(x0, y0); swap_rows32<J>(x1, y1);
        x = (T)(((uint64_t)x1 << 32) | x0); y = (T)(((uint64_t)y1 << 32) | y0);
    } else {
        uint32_t x0 = (uint32_t)x, y0 = (uint32_t)y;
        swap_rows32<J>(x0, y0);
        x = (T)x0; y = (T)y0;
    }
}

template <int E, int J, typename T>
__device__ __forceinline__ void lane_stage(T (&a)[E], uint32_t idx0, uint32_t k) {
    if constexpr (E >= 2 && (J == 16 || J == 32)) {
        // elements h, h + 1 of a lane trade rows: afterwards the lane holds positions p (bit J clear) and p + J
        const uint32_t lane = idx0 & 63u, second = (lane & J) ? 1u : 0u;
        const uint32_t p0 = (idx0 & ~63u) + (lane & ~(uint32_t)J);
#pragma unroll
        for (int h = 0; h < E; h += 2) {
            T x = a[h], y = a[h + 1];
            swap_rows<J, T>(x, y);
            const bool asc = ((p0 + ((uint32_t)h + second) * 64u) & k) == 0;
            const bool sw = key_lt(y, x) == asc;   // (equal keys - sentinels - may swap: no effect)
            a[h] = sw ? y : x;
            a[h + 1] = sw ? x : y;
            swap_rows<J, T>(a[h], a[h + 1]);
        }
    } else {
        const bool lower = (idx0 & J) == 0;  // J < 64: a property of the lane only
#pragma unroll
        for (int h = 0; h < E; ++h) {
            const bool want_min = lower == (((idx0 + h * 64) & k) == 0);
            const T o = xor_lane<J, T>(a[h]);
            // keep own value iff it is on the wanted side of the partner's: one compare, one select
            a[h] = (key_lt(a[h], o) == want_min) ? a[h] : o;
        }
    }
}

template <int E, int JH, typename T>
__device__ __forceinline__ void reg_stage(T (&a)[E], uint32_t idx0, uint32_t k) {
#pragma unroll
    for (int h = 0; h < E; ++h) {
        if ((h & JH) == 0 && (h | JH) < E) {
            const bool asc = ((idx0 + h * 64) & k) == 0;
            const T x = a[h], y = a[h | JH];
            const bool sw = key_lt(y, x) == asc;   // one compare for both directions (equal keys may swap: no effect)
            a[h] = sw ? y : x;
            a[h | JH] = sw ? x : y;
        }
    }
}

// One wave sorts its own 64*E elements (element h of lane l = position h*64 + l) without LDS or barriers: every stage
// is a cross-lane exchange or a register swap.
template <int E, typename T>
__device__ __forceinline__ void wave_bitonic_sort(T (&a)[E]) {
    constexpr uint32_t N = 64 * E;
    const uint32_t lane = lane_id();
    for (uint32_t k = 2; k <= N; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            switch (j) {
                case 1: lane_stage<E, 1, T>(a, lane, k); break;
                case 2: lane_stage<E, 2, T>(a, lane, k); break;
                case 4: lane_stage<E, 4, T>(a, lane, k); break;
                case 8: lane_stage<E, 8, T>(a, lane, k); break;
                case 16: lane_stage<E, 16, T>(a, lane, k); break;
                case 32: lane_stage<E, 32, T>(a, lane, k); break;
                case 64: if constexpr (E >= 2) reg_stage<E, 1, T>(a, lane, k); break;
                case 128: if constexpr (E >= 4) reg_stage<E, 2, T>(a, lane, k); break;
                default: if constexpr (E >= 8) reg_stage<E, 4, T>(a, lane, k); break;
            }
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
                    a[h] = (key_lt(a[h], o) == want_min) ? a[h] : o;
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

// Normalised bitonic network: every comparator is ascending, so positions >= n
// behave as +inf without being stored and any n (not only powers of two) sorts
// in place.  Barrier after every stage.
template <int NT, typename T>
__device__ __forceinline__ void bitonic_sort(T* a, uint32_t n) {
    if (n < 2) return;
    uint32_t np2 = 1;
    while (np2 < n) np2 <<= 1;
    const uint32_t half = np2 >> 1;
    for (uint32_t k = 2; k <= np2; k <<= 1) {
        const uint32_t hk = k >> 1;
        // mirror stage
        for (uint32_t i = threadIdx.x; i < half; i += NT) {
            uint32_t blk = i / hk, o = i - blk * hk;
            uint32_t l = blk * k + o, r = blk * k + (k - 1 - o);
            if (r < n) {
                T x = a[l], y = a[r];
                if (x > y) { a[l] = y; a[r] = x; }
            }
        }
        __syncthreads();
        for (uint32_t j = hk >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < half; i += NT) {
                uint32_t l = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                uint32_t r = l + j;
                if (r < n) {
                    T x = a[l], y = a[r];
                    if (x > y) { a[l] = y; a[r] = x; }
                }
            }
            __syncthreads();
        }
    }
}

template <int NT, typename T, typename Gt>
__device__ __forceinline__ void bitonic_sort_by(T* a, uint32_t n, Gt gt) {
    if (n < 2) return;
    uint32_t np2 = 1;
    while (np2 < n) np2 <<= 1;
    const uint32_t half = np2 >> 1;
    for (uint32_t k = 2; k <= np2; k <<= 1) {
        const uint32_t hk = k >> 1;
        for (uint32_t i = threadIdx.x; i < half; i += NT) {
            uint32_t blk = i / hk, o = i - blk * hk;
            uint32_t l = blk * k + o, r = blk * k + (k - 1 - o);
            if (r < n) {
                T x = a[l], y = a[r];
                if (gt(x, y)) { a[l] = y; a[r] = x; }
            }
        }
        __syncthreads();
        for (uint32_t j = hk >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < half; i += NT) {
                uint32_t l = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                uint32_t r = l + j;
                if (r < n) {
                    T x = a[l], y = a[r];
                    if (gt(x, y)) { a[l] = y; a[r] = x; }
                }
            }
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------
// little-endian field loads at arbitrary byte alignment
template <int W>
__device__ __forceinline__ uint64_t ld_le(const uint8_t* p) {
    if constexpr (W == 4) {
        if ((((uintptr_t)p) & 3) == 0) return *(const uint32_t*)p;
    }
    if constexpr (W == 8) {
        if ((((uintptr_t)p) & 7) == 0) return *(const uint64_t*)p;
    }
    uint64_t v = 0;
#pragma unroll
    for (int i = 0; i < W; ++i) v |= (uint64_t)p[i] << (8 * i);
    return v;
}
__device__ __forceinline__ uint32_t ld_u32(const uint8_t* p, bool aligned) {
    if (aligned) return *(const uint32_t*)p;
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

__device__ __forceinline__ void set_err(DevStatus* st, uint32_t code, uint32_t cell) {
    if (atomicCAS(&st->err_code, 0u, code) == 0u) st->err_cell = cell;
}



// Same normalised network, but every run of stages whose comparators stay inside an aligned TILE of
// elements is executed on an LDS copy of that tile: for n elements only the stages with a partner
// distance >= TILE touch global memory (1+2+...+log2(np2/TILE) of them instead of log2(np2)^2/2).
template <int NT, uint32_t TILE, typename T, typename Gt>
__device__ __forceinline__ void tiled_bitonic_sort_by(T* a, uint32_t n, Gt gt, T* tile) {
    if (n < 2) return;
    uint32_t np2 = 1;
    while (np2 < n) np2 <<= 1;
    auto ce_tile = [&](uint32_t l, uint32_t r, uint32_t lim) {
        if (r < lim) {
            T x = tile[l], y = tile[r];
            if (gt(x, y)) { tile[l] = y; tile[r] = x; }
        }
    };
    // A tile that holds only lim < TILE elements (a cell's last one) needs the comparators inside the first P = 2^ceil(log2 lim) of its
    // slots only: a comparator (l, r) of a stage with partner distance j < P has both ends below P exactly when its index is below
    // P / 2, one with j >= P - or a mirror stage of a level above P - has r >= P >= lim and does nothing.  (Until late round 6 every
    // stage walked all TILE / 2 comparator indices of every tile: a scATAC cell of 18 000 fragments paid for its second tile of
    // 1 616 as for its first of 16 384.)
    auto pow2_at_least = [](uint32_t x) { uint32_t p = 2; while (p < x) p <<= 1; return p; };
    // The half-cleaner stages of a level two at a time: the comparators of the stages with partner distances j and j / 2 that touch the
    // four elements base, base + j/2, base + j, base + 3j/2 touch nothing else, so a thread that holds the four does both stages in
    // registers - half the LDS reads and writes and half the barriers of a level's tail (each stage by itself: two 8- or 16-byte reads
    // and up to two writes per comparator, a barrier per stage; the scATAC sort of 16 384 keys was 105 such visits, now 63).
    // Elements at lim and beyond do not exist (the network treats them as +infinity): the four indices ascend, so do their validities.
    auto stages_from = [&](uint32_t j0, uint32_t hp, uint32_t lim) {
        uint32_t j = j0;
        for (; j >= 2; j >>= 2) {
            const uint32_t h = j >> 1;
            for (uint32_t i = threadIdx.x; i < (hp >> 1); i += NT) {
                const uint32_t b = ((i & ~(h - 1)) << 2) | (i & (h - 1));
                if (b + h >= lim) continue;   // (fewer than two of the four exist: no comparator)
                const bool v2 = b + j < lim, v3 = b + j + h < lim;
                T x0 = tile[b], x1 = tile[b + h], x2 = v2 ? tile[b + j] : x1, x3 = v3 ? tile[b + j + h] : x1;
                if (v2 && gt(x0, x2)) { const T t = x0; x0 = x2; x2 = t; }
                if (v3 && gt(x1, x3)) { const T t = x1; x1 = x3; x3 = t; }
                if (gt(x0, x1)) { const T t = x0; x0 = x1; x1 = t; }
                if (v3 && gt(x2, x3)) { const T t = x2; x2 = x3; x3 = t; }
                tile[b] = x0; tile[b + h] = x1;
                if (v2) tile[b + j] = x2;
                if (v3) tile[b + j + h] = x3;
            }
            __syncthreads();
        }
        if (j == 1) {   // (an odd number of stages: the last one by itself)
            for (uint32_t i = threadIdx.x; i < hp; i += NT) ce_tile(i << 1, (i << 1) + 1, lim);
            __syncthreads();
        }
    };
    // stages with distance < TILE for level k (kk = min(k, TILE) gives the first in-tile stage), tile by tile
    auto tile_pass = [&](uint32_t k, bool with_mirror) {
        for (uint32_t t0 = 0; t0 < n; t0 += TILE) {
            const uint32_t lim = (n - t0 < TILE) ? n - t0 : TILE;
            const uint32_t P = pow2_at_least(lim), hp = P >> 1;   // (uniform)
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < lim; i += NT) tile[i] = a[t0 + i];
            __syncthreads();
            const uint32_t kk = k < TILE ? k : TILE;
            if (with_mirror && kk <= P) {  // only when k <= TILE: the whole level lives in the tile
                const uint32_t hk = kk >> 1;
                for (uint32_t i = threadIdx.x; i < hp; i += NT) {
                    const uint32_t blk = i / hk, o = i - blk * hk;
                    ce_tile(blk * kk + o, blk * kk + (kk - 1 - o), lim);
                }
                __syncthreads();
            }
            uint32_t j0 = with_mirror ? (kk >> 2) : (TILE >> 1);
            if (j0 > hp) j0 = hp;
            stages_from(j0, hp, lim);
            for (uint32_t i = threadIdx.x; i < lim; i += NT) a[t0 + i] = tile[i];
        }
        __syncthreads();
    };
    // levels that fit a tile: sort every tile completely in one visit
    {
        for (uint32_t t0 = 0; t0 < n; t0 += TILE) {
            const uint32_t lim = (n - t0 < TILE) ? n - t0 : TILE;
            const uint32_t P = pow2_at_least(lim), hp = P >> 1;   // (uniform; the tile's elements are sorted after level P)
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < lim; i += NT) tile[i] = a[t0 + i];
            __syncthreads();
            for (uint32_t k = 2; k <= P && k <= np2; k <<= 1) {
                const uint32_t hk = k >> 1;
                for (uint32_t i = threadIdx.x; i < hp; i += NT) {
                    const uint32_t blk = i / hk, o = i - blk * hk;
                    ce_tile(blk * k + o, blk * k + (k - 1 - o), lim);
                }
                __syncthreads();
                stages_from(hk >> 1, hp, lim);
            }
            for (uint32_t i = threadIdx.x; i < lim; i += NT) a[t0 + i] = tile[i];
        }
        __syncthreads();
    }
    const uint32_t half = np2 >> 1;
    for (uint32_t k = 2 * TILE; k <= np2; k <<= 1) {
        const uint32_t hk = k >> 1;
        for (uint32_t i = threadIdx.x; i < half; i += NT) {  // mirror stage, global
            const uint32_t blk = i / hk, o = i - blk * hk;
            const uint32_t l = blk * k + o, r = blk * k + (k - 1 - o);
            if (r < n) {
                T x = a[l], y = a[r];
                if (gt(x, y)) { a[l] = y; a[r] = x; }
            }
        }
        __syncthreads();
        for (uint32_t j = hk >> 1; j >= TILE; j >>= 1) {  // global stages
            for (uint32_t i = threadIdx.x; i < half; i += NT) {
                const uint32_t l = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                const uint32_t r = l + j;
                if (r < n) {
                    T x = a[l], y = a[r];
                    if (gt(x, y)) { a[l] = y; a[r] = x; }
                }
            }
            __syncthreads();
        }
        tile_pass(k, false);  // j = TILE/2 ... 1 inside the tiles
    }
}

__device__ __forceinline__ uint32_t lower_bound_u32(const uint32_t* a, uint32_t n, uint32_t x) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (a[mid] < x) lo = mid + 1; else hi = mid; }
    return lo;
}

// USA gene-id convention (3-column tg-map): spliced ids even, the unspliced sibling = id | 1
__device__ __forceinline__ bool is_spliced(uint32_t g) { return (g & 1u) == 0; }
__device__ __forceinline__ bool same_gene(uint32_t a, uint32_t b) { return (a & ~1u) == (b & ~1u); }

// Loads of data a kernel reads once.  Measured on the headline (profiles/history/run_r04ac.sh, three rounds, per step): the bucket's keys
// in k_resolve non-temporal: 4.42 -> 4.35 ms, and k_cell_hist behind it 0.554 -> 0.529 (what the resolve leaves in L2 is the
// column lists the histograms read) - kept; the input bytes in k_decode_recs: 4.47 -> 4.60 (the next slab's halo is this slab's
// tail); keys0 in k_scatter: 2.61 -> 2.57 but k_resolve 4.42 -> 4.46 - neither kept (make variant DEFS=-DAFQ_NT_DECODE / _SCATTER;
// -DAFQ_PLAIN_RESOLVE_LOADS for the resolve's old loads).
template <typename T>
__device__ __forceinline__ T ld_nt(const T* p) { return __builtin_nontemporal_load(p); }
#ifdef AFQ_NT_DECODE
#define AFQ_LD_DECODE(p) ld_nt(p)
#else
#define AFQ_LD_DECODE(p) (*(p))
#endif
#ifdef AFQ_NT_SCATTER
#define AFQ_LD_SCATTER(p) ld_nt(p)
#else
#define AFQ_LD_SCATTER(p) (*(p))
#endif
#ifdef AFQ_PLAIN_RESOLVE_LOADS
#define AFQ_LD_RESOLVE(p) (*(p))
#else
#define AFQ_LD_RESOLVE(p) ld_nt(p)
#endif

#define AFQ_LAUNCH(kern, grid, block, stream, ...) hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, stream, __VA_ARGS__)

}  // namespace afq
