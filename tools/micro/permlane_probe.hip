// Micro-probe: semantics of v_permlane16_swap / v_permlane32_swap as exposed by the HIP builtins (development aid).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned *out) {
    const unsigned u = threadIdx.x;
    u32x2 a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    u32x2 b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    out[threadIdx.x * 4 + 0] = a[0]; out[threadIdx.x * 4 + 1] = a[1];
    out[threadIdx.x * 4 + 2] = b[0]; out[threadIdx.x * 4 + 3] = b[1];
}
int main() {
    unsigned *d, h[256];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 1;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    for (int j = 0; j < 4; ++j) {
        printf("%s: ", j == 0 ? "p16[0]" : j == 1 ? "p16[1]" : j == 2 ? "p32[0]" : "p32[1]");
        for (int l = 0; l < 64; l += 4) printf("%2u ", h[l * 4 + j]);
        printf("  (lanes 0,4,8,...)\n");
    }
    return 0;
}
