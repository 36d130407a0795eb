// Stand-alone timing of the fused output-conv kernel (plc_post_kernel.h) on synthetic data, with the per-phase
// cycle sums of its instrumented build:
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -fno-slp-vectorize -DPLC_POST_TIMELINE -Iinclude \
//         -Ipl_convlstm_gan_amd/csrc -o devtools/post_probe devtools/post_probe.hip && devtools/post_probe [B T H W]
// Phases per round: 0 product of the next tile (and its gate fetch when it begins a chain), 1 first conv (with the request
// of the tile after next behind its MFMAs), 2 wait at the first barrier, 3 second conv, 4 wait at the second barrier.  Values are garbage; only times
// mean anything.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "plc_post_kernel.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <bool SAVE>
static int run(PostArgs a, int grid, const char* name) {
    auto kern = plc_post_fused_kernel<SAVE>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, POST_LDS_BYTES));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), POST_LDS_BYTES, 0, a);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int iters = 20;
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), POST_LDS_BYTES, 0, a);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%s: %.1f us per launch, %d workgroups, %.2f tiles each\n", name, ms * 1e3 / iters, grid, (double)a.n_tiles / grid);
    std::vector<long long> tl((size_t)grid * 8 * 5);
    CK(hipMemcpy(tl.data(), a.timeline, tl.size() * 8, hipMemcpyDeviceToHost));
    for (int wg : {0, grid / 2}) {
        for (int w : {0, 3, 4, 7}) {
            printf("  wg %3d wave %d cycles:", wg, w);
            long long tot = 0;
            for (int i = 0; i < 5; ++i) { printf(" %9lld", tl[((size_t)wg * 8 + w) * 5 + i]); tot += tl[((size_t)wg * 8 + w) * 5 + i]; }
            printf("  total %lld\n", tot);
        }
    }
    return 0;
}

int main(int argc, char** argv) {
    const int B = argc > 4 ? atoi(argv[1]) : 4, T = argc > 4 ? atoi(argv[2]) : 10;
    const int H = argc > 4 ? atoi(argv[3]) : 256, W = argc > 4 ? atoi(argv[4]) : 256;
    const int NI = B * T;
    const size_t plane = (size_t)H * W * 16;
    char *feat, *gate, *p0, *out, *wpk0, *wpk2, *b0, *b2;
    long long* tl;
    CK(hipMalloc(&feat, NI * 4 * plane)); CK(hipMalloc(&gate, B * 4 * plane)); CK(hipMalloc(&p0, NI * 4 * plane));
    CK(hipMalloc(&out, (size_t)NI * H * W * 4));
    CK(hipMalloc(&wpk0, 18 * 64 * 16)); CK(hipMalloc(&wpk2, 18 * 64 * 16)); CK(hipMalloc(&b0, 128)); CK(hipMalloc(&b2, 128));
    CK(hipMemset(feat, 0x3c, NI * 4 * plane)); CK(hipMemset(gate, 0x3c, B * 4 * plane));       // bf16 0x3c3c ~ 0.0115
    CK(hipMemset(wpk0, 0x3c, 18 * 64 * 16)); CK(hipMemset(wpk2, 0x3c, 18 * 64 * 16));
    CK(hipMemset(b0, 0, 128)); CK(hipMemset(b2, 0, 128));
    PostArgs a;
    memset(&a, 0, sizeof(a));
    a.H = H; a.W = W; a.n_in = B; a.T = T;
    a.tiles_x = (W + POST_TW - 1) / POST_TW; a.tiles_y = (H + POST_TH - 1) / POST_TH;
    a.n_tiles = NI * a.tiles_x * a.tiles_y;
    a.feat = PlcTensor{feat, (long long)(4 * plane), (long long)(B * 4 * plane)};
    a.gate = PlcTensor{gate, (long long)(4 * plane), 0};
    a.p0 = PlcTensor{p0, (long long)(4 * plane), (long long)(B * 4 * plane)};
    a.out = PlcTensor{out, (long long)T * H * W * 4, (long long)H * W * 4};
    a.wpk0 = (const uint4*)wpk0; a.bpk0 = (const float*)b0; a.wpk2 = (const uint4*)wpk2; a.bpk2 = (const float*)b2;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int grid = a.n_tiles < cus ? a.n_tiles : cus;
    CK(hipMalloc(&tl, (size_t)grid * 8 * 5 * 8));
    a.timeline = tl;
    if (run<false>(a, grid, "inference")) return 1;
    if (run<true>(a, grid, "training (saves the intermediate)")) return 1;
    return 0;
}
