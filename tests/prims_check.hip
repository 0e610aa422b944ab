// prims_check: the sort / scan primitives of alevin-fry_amd/csrc/afq_prims.h against std::sort / a serial scan on the
// host, for every element type, elements-per-lane count and fill level the kernels instantiate.  Test infrastructure
// (built and run by tests/test_gpu_prims.py); prints "ok <n checks>" or the first mismatch and exits non-zero.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>

#include "../alevin-fry_amd/csrc/afq_prims.h"

using namespace afq;
typedef unsigned __int128 u128;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("hip error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

template <int E, typename T>
__global__ void k_wave_sort(T* a, uint32_t n, T sentinel) {   // one wave per block: its own 64 * E slots
    T* base = a + (size_t)blockIdx.x * 64 * E;
    T r[E];
#pragma unroll
    for (int h = 0; h < E; ++h) r[h] = (uint32_t)(h * 64) + lane_id() < n ? base[h * 64 + lane_id()] : sentinel;
    wave_bitonic_sort<E, T>(r);
#pragma unroll
    for (int h = 0; h < E; ++h) base[h * 64 + lane_id()] = r[h];
}

template <int NT, int E, typename T>
__global__ __launch_bounds__(NT) void k_block_sort(T* a, uint32_t n, T sentinel) {
    __shared__ T s_x[NT * E];
    T* base = a + (size_t)blockIdx.x * NT * E;
    block_sort_to_lds<NT, E, T>(base, n, s_x, sentinel);
    for (uint32_t i = threadIdx.x; i < (uint32_t)(NT * E); i += NT) base[i] = i < n ? s_x[i] : sentinel;
}

template <int NT>
__global__ __launch_bounds__(NT) void k_scan(const uint32_t* v, uint32_t* ex, uint32_t* tot) {
    __shared__ uint32_t ws[NT / 64];
    uint32_t t;
    const uint32_t e = block_excl_scan<NT>(v[blockIdx.x * NT + threadIdx.x], ws, t);
    ex[blockIdx.x * NT + threadIdx.x] = e;
    if (threadIdx.x == NT - 1) tot[blockIdx.x] = t;
}

template <typename T>
static T rnd(std::mt19937_64& g, int narrow) {
    if constexpr (sizeof(T) == 16) {
        // narrow: few distinct high halves, so the low half decides most compares
        const uint64_t hi = narrow ? g() % 5 : g(), lo = narrow == 2 ? g() % 3 : g();
        return ((T)hi << 64) | lo;
    } else return (T)(narrow ? g() % 97 : g());
}

static int n_checks = 0;

template <int E, typename T>
static int check_wave(std::mt19937_64& g) {
    constexpr uint32_t N = 64 * E, B = 48;
    const T sentinel = ~(T)0;
    for (int narrow = 0; narrow < 3; ++narrow)
        for (uint32_t n : {N, N - 1, N / 2 + 1, N / 2, 3u, 1u}) {
            std::vector<T> h(N * B);
            for (auto& x : h) x = rnd<T>(g, narrow);
            T* d;
            CK(hipMalloc(&d, sizeof(T) * h.size()));
            CK(hipMemcpy(d, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice));
            hipLaunchKernelGGL((k_wave_sort<E, T>), dim3(B), dim3(64), 0, 0, d, n, sentinel);
            std::vector<T> got(h.size());
            CK(hipMemcpy(got.data(), d, sizeof(T) * h.size(), hipMemcpyDeviceToHost));
            CK(hipFree(d));
            for (uint32_t b = 0; b < B; ++b) {
                std::vector<T> want(h.begin() + b * N, h.begin() + b * N + n);
                std::sort(want.begin(), want.end());
                want.resize(N, sentinel);
                if (!std::equal(want.begin(), want.end(), got.begin() + b * N)) {
                    std::printf("wave sort mismatch: bytes=%zu E=%d n=%u narrow=%d block=%u\n", sizeof(T), E, n, narrow, b);
                    return 1;
                }
            }
            ++n_checks;
        }
    return 0;
}

template <int NT, int E, typename T>
static int check_block(std::mt19937_64& g) {
    constexpr uint32_t N = NT * E, B = 6;
    const T sentinel = ~(T)0;
    for (int narrow = 0; narrow < 3; ++narrow)
        for (uint32_t n : {N, N - 1, N / 2 + 7, N < 65u ? 33u : 65u, 2u}) {
            std::vector<T> h(N * B);
            for (auto& x : h) x = rnd<T>(g, narrow);
            T* d;
            CK(hipMalloc(&d, sizeof(T) * h.size()));
            CK(hipMemcpy(d, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice));
            hipLaunchKernelGGL((k_block_sort<NT, E, T>), dim3(B), dim3(NT), 0, 0, d, n, sentinel);
            std::vector<T> got(h.size());
            CK(hipMemcpy(got.data(), d, sizeof(T) * h.size(), hipMemcpyDeviceToHost));
            CK(hipFree(d));
            for (uint32_t b = 0; b < B; ++b) {
                std::vector<T> want(h.begin() + b * N, h.begin() + b * N + n);
                std::sort(want.begin(), want.end());
                want.resize(N, sentinel);
                if (!std::equal(want.begin(), want.end(), got.begin() + b * N)) {
                    std::printf("block sort mismatch: bytes=%zu NT=%d E=%d n=%u narrow=%d block=%u\n", sizeof(T), NT, E, n, narrow, b);
                    return 1;
                }
            }
            ++n_checks;
        }
    return 0;
}

template <int NT>
static int check_scan(std::mt19937_64& g) {
    constexpr uint32_t B = 9;
    std::vector<uint32_t> v(NT * B);
    for (auto& x : v) x = (uint32_t)(g() % 1000);
    for (uint32_t i = 0; i < (uint32_t)NT; ++i) v[i] = 0xFFFFFFFFu / NT;   // (sums up to the top of 32 bits)
    uint32_t *dv, *de, *dt;
    CK(hipMalloc(&dv, 4 * v.size())); CK(hipMalloc(&de, 4 * v.size())); CK(hipMalloc(&dt, 4 * B));
    CK(hipMemcpy(dv, v.data(), 4 * v.size(), hipMemcpyHostToDevice));
    hipLaunchKernelGGL((k_scan<NT>), dim3(B), dim3(NT), 0, 0, dv, de, dt);
    std::vector<uint32_t> ex(v.size()), tot(B);
    CK(hipMemcpy(ex.data(), de, 4 * v.size(), hipMemcpyDeviceToHost));
    CK(hipMemcpy(tot.data(), dt, 4 * B, hipMemcpyDeviceToHost));
    CK(hipFree(dv)); CK(hipFree(de)); CK(hipFree(dt));
    for (uint32_t b = 0; b < B; ++b) {
        uint32_t run = 0;
        for (uint32_t i = 0; i < (uint32_t)NT; ++i) {
            if (ex[b * NT + i] != run) { std::printf("scan mismatch: NT=%d block=%u i=%u got=%u want=%u\n", NT, b, i, ex[b * NT + i], run); return 1; }
            run += v[b * NT + i];
        }
        if (tot[b] != run) { std::printf("scan total mismatch: NT=%d block=%u\n", NT, b); return 1; }
    }
    ++n_checks;
    return 0;
}

// tiled_bitonic_sort_by: one workgroup sorts a[0, n) in place through an LDS tile of TILE elements - every fill of the last
// tile (none, one element, just under / at / just over a power of two, a full tile), one to five tiles
template <int NT, uint32_t TILE, typename T>
__global__ __launch_bounds__(NT) void k_tiled_sort(T* a, uint32_t n, uint32_t stride) {
    __shared__ T s_tile[TILE];
    tiled_bitonic_sort_by<NT, TILE>(a + (size_t)blockIdx.x * stride, n, [](T x, T y) { return x > y; }, s_tile);
}
template <int NT, uint32_t TILE, typename T>
static int check_tiled(std::mt19937_64& g, std::vector<uint32_t> sizes) {
    constexpr uint32_t B = 3;
    for (int narrow = 0; narrow < 2; ++narrow)
        for (uint32_t n : sizes) {
            std::vector<T> h((size_t)n * B + 1);
            for (auto& x : h) x = rnd<T>(g, narrow);
            T* d;
            CK(hipMalloc(&d, sizeof(T) * h.size()));
            CK(hipMemcpy(d, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice));
            hipLaunchKernelGGL((k_tiled_sort<NT, TILE, T>), dim3(B), dim3(NT), 0, 0, d, n, n);
            std::vector<T> got(h.size());
            CK(hipMemcpy(got.data(), d, sizeof(T) * h.size(), hipMemcpyDeviceToHost));
            CK(hipFree(d));
            for (uint32_t b = 0; b < B; ++b) {
                std::vector<T> want(h.begin() + (size_t)b * n, h.begin() + (size_t)(b + 1) * n);
                std::sort(want.begin(), want.end());
                if (!std::equal(want.begin(), want.end(), got.begin() + (size_t)b * n)) {
                    std::printf("tiled sort mismatch: bytes=%zu NT=%d TILE=%u n=%u narrow=%d block=%u\n", sizeof(T), NT, TILE, n, narrow, b);
                    return 1;
                }
            }
            if (got.back() != h.back()) { std::printf("tiled sort wrote past its end: TILE=%u n=%u\n", TILE, n); return 1; }
            ++n_checks;
        }
    return 0;
}

int main() {
    std::mt19937_64 g(20260927);
    int rc = 0;
#define RUN(x) do { if (!rc) rc = (x); } while (0)
    RUN((check_wave<1, uint32_t>(g))); RUN((check_wave<2, uint32_t>(g))); RUN((check_wave<4, uint32_t>(g))); RUN((check_wave<8, uint32_t>(g)));
    RUN((check_wave<1, uint64_t>(g))); RUN((check_wave<2, uint64_t>(g))); RUN((check_wave<4, uint64_t>(g))); RUN((check_wave<8, uint64_t>(g)));
    RUN((check_wave<1, u128>(g))); RUN((check_wave<2, u128>(g))); RUN((check_wave<4, u128>(g))); RUN((check_wave<8, u128>(g)));
    RUN((check_block<64, 1, uint64_t>(g))); RUN((check_block<64, 8, uint64_t>(g)));
    RUN((check_block<256, 1, uint32_t>(g))); RUN((check_block<256, 2, uint64_t>(g))); RUN((check_block<256, 4, uint64_t>(g))); RUN((check_block<256, 8, uint64_t>(g)));
    RUN((check_block<1024, 1, u128>(g))); RUN((check_block<1024, 2, u128>(g))); RUN((check_block<1024, 4, u128>(g)));
    RUN((check_block<1024, 1, uint64_t>(g))); RUN((check_block<1024, 2, uint64_t>(g))); RUN((check_block<1024, 4, uint64_t>(g))); RUN((check_block<1024, 8, uint64_t>(g)));
    RUN((check_block<1024, 4, uint32_t>(g)));
    RUN((check_tiled<256, 1024, uint64_t>(g, {1, 2, 3, 5, 63, 64, 65, 511, 512, 513, 1023, 1024, 1025, 1026, 1029, 1100, 1279, 1280, 1281, 1536, 1537, 2047, 2048, 2049, 2052, 2600, 3071, 3073, 4096, 4097, 4500, 5121})));
    RUN((check_tiled<256, 1024, uint32_t>(g, {1, 2, 7, 1000, 1024, 1025, 1153, 2048, 2049, 3000, 4099})));
    RUN((check_tiled<1024, 4096, u128>(g, {4095, 4096, 4097, 4100, 4609, 6000, 8192, 8193, 9000})));
    RUN((check_tiled<1024, 16384, uint64_t>(g, {16383, 16384, 16385, 18000, 20000, 24577, 32768, 33000})));   // (k_atac_dedup64's instance: a cell of 18 000 fragments)
    RUN((check_scan<64>(g))); RUN((check_scan<256>(g))); RUN((check_scan<1024>(g)));
    CK(hipDeviceSynchronize());
    if (rc) return rc;
    std::printf("ok %d checks\n", n_checks);
    return 0;
}
