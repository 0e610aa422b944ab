// Dependent random reads: every block owns a region of `words` u64; each thread chases `steps` pseudo-random slots.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ __launch_bounds__(1024) void chase(const unsigned long long* base, size_t stride_words, unsigned words, unsigned steps, unsigned long long* out) {
    const unsigned long long* p = base + stride_words * blockIdx.x;
    unsigned long long x = threadIdx.x * 0x9E3779B97F4A7C15ull + blockIdx.x;
    unsigned long long acc = 0;
    for (unsigned s = 0; s < steps; ++s) {
        x = x * 6364136223846793005ull + 1442695040888963407ull + acc;
        const unsigned slot = (unsigned)((x >> 33) % words);
        acc += p[slot];
    }
    if (acc == 0x1234567) out[0] = acc;
}
int main(int argc, char** argv) {
    const unsigned blocks = 256, steps = 256;
    for (size_t mb : {1, 4, 16, 64}) {
        for (int threads : {256, 1024}) {
            const unsigned words = (unsigned)(mb * 1024 * 1024 / 8);
            unsigned long long *d, *o;
            hipMalloc(&d, (size_t)blocks * words * 8);
            hipMalloc(&o, 8);
            hipMemset(d, 0, (size_t)blocks * words * 8);
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            chase<<<blocks, threads>>>(d, words, words, 8, o);
            hipDeviceSynchronize();
            hipEventRecord(a);
            chase<<<blocks, threads>>>(d, words, words, steps, o);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("region %3zu MB/block (total %5zu MB) threads %4d: %.3f ms -> %.2f us per dependent access, %.1f G accesses/s\n", mb, mb * blocks, threads, ms,
                   ms * 1e3 / steps, (double)blocks * threads * steps / ms / 1e6);
            hipFree(d); hipFree(o);
        }
    }
    return 0;
}
