#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* out) {
    unsigned x = threadIdx.x;
    out[0 * 64 + x] = __builtin_amdgcn_update_dpp(999u, x, 0xB1, 0xF, 0xF, false);
    out[1 * 64 + x] = __builtin_amdgcn_update_dpp(999u, x, 0x4E, 0xF, 0xF, false);
    out[2 * 64 + x] = __builtin_amdgcn_update_dpp(999u, x, 0x104, 0xF, 0x5, false);   // row_shl:4, banks 0,2
    out[3 * 64 + x] = __builtin_amdgcn_update_dpp(999u, x, 0x114, 0xF, 0xA, false);   // row_shr:4, banks 1,3
    out[4 * 64 + x] = __builtin_amdgcn_update_dpp(999u, x, 0x108, 0xF, 0x3, false);   // row_shl:8, banks 0,1
    out[5 * 64 + x] = __builtin_amdgcn_update_dpp(999u, x, 0x118, 0xF, 0xC, false);   // row_shr:8, banks 2,3
    v2u a = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    out[6 * 64 + x] = a.x; out[7 * 64 + x] = a.y;
    v2u b = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    out[8 * 64 + x] = b.x; out[9 * 64 + x] = b.y;
}
int main() {
    unsigned* d; hipMalloc(&d, 10 * 64 * 4);
    hipLaunchKernelGGL(k, 1, 64, 0, 0, d);
    unsigned h[640]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[10] = {"quad B1", "quad 4E", "shl4 b0101", "shr4 b1010", "shl8 b0011", "shr8 b1100", "pl16.x", "pl16.y", "pl32.x", "pl32.y"};
    for (int r = 0; r < 10; ++r) { printf("%-11s", names[r]); for (int i = 0; i < 64; ++i) printf(" %u", h[r * 64 + i]); printf("\n"); }
    return 0;
}
