// What a plain streaming kernel reaches on this GPU, for comparison with the engine's HBM-bound kernels:
// copy (1 read : 1 write), read-only, and the pointwise LSTM backward's mix (9 reads : 5 writes of 16 bytes
// per lane over separate arrays).   hipcc -O3 --offload-arch=gfx950 -o bw_probe bw_probe.hip && ./bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int NR, int NW, int UPT>
__global__ void stream_kernel(const uint4* const* in, uint4* const* out, long long n, uint4* sink) {
    // each thread handles UPT units spaced a block apart; reads NR arrays, writes NW
    long long i0 = ((long long)blockIdx.x * UPT) * blockDim.x + threadIdx.x;
    uint4 acc[UPT];
#pragma unroll
    for (int u = 0; u < UPT; ++u) acc[u] = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int u = 0; u < UPT; ++u) {
            const long long i = i0 + (long long)u * blockDim.x;
            if (i < n) { const uint4 v = in[r][i]; acc[u].x ^= v.x; acc[u].y += v.y; acc[u].z ^= v.z; acc[u].w += v.w; }
        }
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
        for (int u = 0; u < UPT; ++u) {
            const long long i = i0 + (long long)u * blockDim.x;
            if (i < n) out[w][i] = make_uint4(acc[u].x + w, acc[u].y, acc[u].z, acc[u].w);
        }
    if (NW == 0) {
#pragma unroll
        for (int u = 0; u < UPT; ++u) if (acc[u].x == 0x12345678u && acc[u].y == 0x9abcdef0u) sink[0] = acc[u];
    }
}

template <int NR, int NW, int UPT>
int run(const char* name, long long n, int threads, uint4** d_in_tab, uint4** d_out_tab, uint4* sink) {
    const long long per_block = (long long)threads * UPT;
    const int grid = (int)((n + per_block - 1) / per_block);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((stream_kernel<NR, NW, UPT>), dim3(grid), dim3(threads), 0, 0, d_in_tab, d_out_tab, n, sink);
    CK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((stream_kernel<NR, NW, UPT>), dim3(grid), dim3(threads), 0, 0, d_in_tab, d_out_tab, n, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)n * 16 * (NR + NW);
    printf("%-34s threads %4d units/thread %d: %7.1f us  %6.2f TB/s\n", name, threads, UPT, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12);
    return 0;
}

int main(int argc, char** argv) {
    // units of 16 bytes per array; argv[1] = multiple of one cell-2 pointwise launch at 256 x 256, B = 4 (1048576 units:
    // 235 MB over the 14 arrays, which the 256 MB Infinity Cache holds from one repetition to the next; 8 = 1.9 GB does not fit)
    const long long n = 1048576ll * (argc > 1 ? atoi(argv[1]) : 8);
    const int NA = 14;
    std::vector<uint4*> bufs(NA);
    for (int i = 0; i < NA; ++i) { CK(hipMalloc(&bufs[i], n * 16)); CK(hipMemset(bufs[i], i, n * 16)); }
    uint4 **d_in, **d_out, *sink;
    CK(hipMalloc(&d_in, 9 * sizeof(uint4*))); CK(hipMalloc(&d_out, 5 * sizeof(uint4*))); CK(hipMalloc(&sink, 16));
    CK(hipMemcpy(d_in, bufs.data(), 9 * sizeof(uint4*), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_out, bufs.data() + 9, 5 * sizeof(uint4*), hipMemcpyHostToDevice));
    for (int threads : {256, 512, 1024}) {
        if (run<1, 1, 1>("copy 1r:1w (16 MB each)", n, threads, d_in, d_out, sink)) return 1;
        if (run<9, 0, 1>("read 9 arrays", n, threads, d_in, d_out, sink)) return 1;
        if (run<9, 5, 1>("9r:5w (pointwise-backward mix)", n, threads, d_in, d_out, sink)) return 1;
        if (run<9, 5, 2>("9r:5w (pointwise-backward mix)", n, threads, d_in, d_out, sink)) return 1;
        if (run<9, 5, 4>("9r:5w (pointwise-backward mix)", n, threads, d_in, d_out, sink)) return 1;
    }
    return 0;
}
