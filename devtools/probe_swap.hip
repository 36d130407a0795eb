// Prints what v_permlane32_swap_b32 leaves in its two operands:  hipcc --offload-arch=gfx950 -o probe_swap probe_swap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* o) {
    unsigned a = threadIdx.x, b = 1000 + threadIdx.x;
    u2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    o[threadIdx.x] = r.x; o[64 + threadIdx.x] = r.y;
}
int main() {
    unsigned* d; unsigned h[128];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 1;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    for (int l : {0, 1, 31, 32, 33, 63}) printf("lane %2d: first %4u second %4u\n", l, h[l], h[64 + l]);
    return 0;
}
