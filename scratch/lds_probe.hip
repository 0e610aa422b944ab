// LDS open-addressing table (u64 slots carved from a u32 __shared__ array): insert + random probes, timing.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(1024) void k(unsigned n_ins, unsigned n_probe, unsigned long long* out) {
    __shared__ __attribute__((aligned(16))) unsigned s_big[1u << 15];
    unsigned long long* tab = reinterpret_cast<unsigned long long*>(s_big);
    const unsigned tid = threadIdx.x;
    for (unsigned i = tid; i < (1u << 14); i += 1024) tab[i] = ~0ull;
    __syncthreads();
    for (unsigned v = tid; v < n_ins; v += 1024) {
        unsigned long long key = (v * 0x9E3779B97F4A7C15ull) >> 20;
        unsigned slot = (unsigned)(key * 2654435761u) & 16383u;
        const unsigned long long e = (key << 20) | v;
        while (atomicCAS(&tab[slot], ~0ull, e) != ~0ull) slot = (slot + 1) & 16383u;
    }
    __syncthreads();
    unsigned long long acc = 0;
    unsigned long long x = tid * 0x9E3779B97F4A7C15ull + blockIdx.x;
    for (unsigned s = 0; s < n_probe; ++s) {
        x = x * 6364136223846793005ull + 1442695040888963407ull;
        const unsigned long long key = x >> 20;
        for (unsigned slot = (unsigned)(key * 2654435761u) & 16383u;; slot = (slot + 1) & 16383u) {
            const unsigned long long e = tab[slot];
            if (e == ~0ull) break;
            if ((e >> 20) == key) acc += e;
        }
    }
    if (acc == 0x1234567) out[0] = acc;
}
int main() {
    unsigned long long* o; (void)hipMalloc(&o, 8);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k<<<256, 1024>>>(6000, 16, o); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    k<<<256, 1024>>>(6000, 2000, o);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("2000 probes/thread x 1024 threads x 256 blocks: %.3f ms -> %.1f ns per probe per thread, %.1f G probes/s\n", ms, ms * 1e6 / 2000, 256.0 * 1024 * 2000 / ms / 1e6);
    return 0;
}
