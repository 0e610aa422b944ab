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
    // stages with distance < TILE for level k (kk = min(k, TILE) gives the first in-tile stage), tile by tile
    auto tile_pass = [&](uint32_t k, bool with_mirror) {
        for (uint32_t t0 = 0; t0 < n; t0 += TILE) {
            const uint32_t lim = (n - t0 < TILE) ? n - t0 : TILE;
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < lim; i += NT) tile[i] = a[t0 + i];
            __syncthreads();
            const uint32_t kk = k < TILE ? k : TILE;
            if (with_mirror) {  // only when k <= TILE: the whole level lives in the tile
                const uint32_t hk = kk >> 1;
                for (uint32_t i = threadIdx.x; i < TILE / 2; i += NT) {
                    const uint32_t blk = i / hk, o = i - blk * hk;
                    ce_tile(blk * kk + o, blk * kk + (kk - 1 - o), lim);
                }
                __syncthreads();
            }
            for (uint32_t j = with_mirror ? (kk >> 2) : (TILE >> 1); j > 0; j >>= 1) {
                for (uint32_t i = threadIdx.x; i < TILE / 2; i += NT) {
                    const uint32_t l = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                    ce_tile(l, l + j, lim);
                }
                __syncthreads();
            }
            for (uint32_t i = threadIdx.x; i < lim; i += NT) a[t0 + i] = tile[i];
        }
        __syncthreads();
    };
    // levels that fit a tile: sort every tile completely in one visit
    {
        for (uint32_t t0 = 0; t0 < n; t0 += TILE) {
            const uint32_t lim = (n - t0 < TILE) ? n - t0 : TILE;
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < lim; i += NT) tile[i] = a[t0 + i];
            __syncthreads();
            for (uint32_t k = 2; k <= TILE && k <= np2; k <<= 1) {
                const uint32_t hk = k >> 1;
                for (uint32_t i = threadIdx.x; i < TILE / 2; i += NT) {
                    const uint32_t blk = i / hk, o = i - blk * hk;
                    ce_tile(blk * k + o, blk * k + (k - 1 - o), lim);
                }
                __syncthreads();
                for (uint32_t j = hk >> 1; j > 0; j >>= 1) {
                    for (uint32_t i = threadIdx.x; i < TILE / 2; i += NT) {
                        const uint32_t l = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                        ce_tile(l, l + j, lim);
                    }
                    __syncthreads();
                }
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

#define AFQ_LAUNCH(kern, grid, block, stream, ...) hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, stream, __VA_ARGS__)

}  // namespace afq
