// Does a hipMemsetAsync captured into a HIP graph stay ordered between the kernels around it?
//   stream order captured:  dirty(buf) -> [memset(buf, 0, nbytes)] -> check(buf)   (check counts non-zero words)
// replayed many times, for a few sizes, beside the same sequence issued eagerly.
//   hipcc --offload-arch=gfx950 -O2 -o devtools/graph_memset_probe devtools/graph_memset_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

__global__ void dirty(unsigned* buf, int n, unsigned v) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) buf[i] = v;
}
__global__ void check(const unsigned* buf, int n, unsigned long long* bad) {
    int c = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) c += buf[i] != 0u;
    if (c) atomicAdd(bad, (unsigned long long)c);
}
// a longer kernel before the sequence, so that the memset has something to overtake
__global__ void busy(float* x, int n, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = x[i];
    for (int k = 0; k < iters; ++k) v = v * 1.0000001f + 1e-7f;
    x[i] = v;
}

int main() {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    unsigned* buf;
    unsigned long long* bad;
    float* work;
    const int WN = 1 << 20;
    CK(hipMalloc(&buf, 1 << 20));
    CK(hipMalloc(&bad, 8));
    CK(hipMalloc(&work, WN * 4));
    CK(hipMemset(work, 0, WN * 4));
    const int sizes[] = {4, 256, 1024, 65536};
    for (int graph = 0; graph < 2; ++graph)
        for (int s : sizes) {
            const int n = s / 4;
            CK(hipMemset(bad, 0, 8));
            CK(hipDeviceSynchronize());
            hipGraph_t g = nullptr;
            hipGraphExec_t ge = nullptr;
            auto seq = [&]() {
                hipLaunchKernelGGL(busy, dim3(WN / 256), dim3(256), 0, st, work, WN, 200);
                hipLaunchKernelGGL(dirty, dim3(4), dim3(256), 0, st, buf, n, 0x7fc00000u);
                CK(hipMemsetAsync(buf, 0, s, st));
                hipLaunchKernelGGL(check, dim3(4), dim3(256), 0, st, (const unsigned*)buf, n, bad);
            };
            const int reps = 3000;
            if (graph) {
                CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
                seq();
                CK(hipStreamEndCapture(st, &g));
                CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
            } else {
                for (int r = 0; r < reps; ++r) seq();
            }
            CK(hipStreamSynchronize(st));
            unsigned long long h = 0;
            CK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
            printf("%s memset %6d bytes: %llu non-zero words seen after the memset in %d runs\n", graph ? "graph" : "eager", s, h, reps);
            if (ge) CK(hipGraphExecDestroy(ge));
            if (g) CK(hipGraphDestroy(g));
        }
    return 0;
}
