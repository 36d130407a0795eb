#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <vector>
#ifndef ABL
#define ABL 0
#endif
#include "conv_abl.h"
int main() {
    const int B = 4, H = 256, W = 256; const long long hw = (long long)H * W;
    char *x, *h, *c, *ho, *co, *wpk, *bpk;
    hipMalloc(&x, B * 2 * hw * 16); hipMalloc(&h, B * 4 * hw * 16); hipMalloc(&c, B * 4 * hw * 32);
    hipMalloc(&ho, B * 4 * hw * 16); hipMalloc(&co, B * 4 * hw * 32);
    hipMalloc(&wpk, 4 * 27 * 64 * 16); hipMalloc(&bpk, 4 * 32 * 4);
    std::vector<unsigned short> r(B * 4 * hw * 8);
    for (size_t i = 0; i < r.size(); ++i) r[i] = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
    hipMemcpy(x, r.data(), B * 2 * hw * 16, hipMemcpyHostToDevice);
    hipMemcpy(h, r.data(), B * 4 * hw * 16, hipMemcpyHostToDevice);
    hipMemset(c, 0, B * 4 * hw * 32);
    for (size_t i = 0; i < 4 * 27 * 64 * 8; ++i) r[i] = (unsigned short)(0x3800 + (rand() & 0x3ff) + ((rand() & 1) << 15));
    hipMemcpy(wpk, r.data(), 4 * 27 * 64 * 16, hipMemcpyHostToDevice);
    hipMemset(bpk, 0, 4 * 32 * 4);
    ConvArgs a; memset(&a, 0, sizeof(a));
    a.H = H; a.W = W; a.n_in = B; a.tiles_x = 8; a.tiles_y = 16;
    a.src0 = {x, 2 * hw * 16, 0}; a.src1 = {h, 4 * hw * 16, 0}; a.c_in = {c, 4 * hw * 32, 0};
    a.out0 = {ho, 4 * hw * 16, 0}; a.out1 = {co, 4 * hw * 32, 0};
    unsigned long long* dbg; hipMalloc(&dbg, 4096); hipMemset(dbg, 0, 4096); a.row_tab = (const float*)dbg;
    unsigned* dbg2; hipMalloc(&dbg2, 8192); hipMemset(dbg2, 0, 8192); a.col_tab = (const float*)dbg2;
    a.wpk = (const uint4*)wpk; a.bpk = (const float*)bpk;
    constexpr int lds = plc_conv_lds_bytes<6, 3, 16, 1>();
    auto kern = plc_conv_mfma_kernel<2, 4, 4, 3, STAGE_CB8, EPI_LSTM, 16, 1>;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0);
        for (int k = 0; k < 50; ++k) hipLaunchKernelGGL(kern, dim3(B * 8 * 16, 1), dim3(256), lds, 0, a);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("ABL=%d: %.2f us per launch (%s)\n", ABL, ms * 1000 / 50, hipGetErrorString(hipGetLastError()));
    }
    { unsigned long long hd[28]; hipMemcpy(hd, dbg, sizeof(hd), hipMemcpyDeviceToHost); for (int i = 0; i < 14; i += 3) printf("  blk %d: pass-loop cycles %llu, realtime ticks %llu -> clock %.0f MHz\n", i*37, hd[2*i], hd[2*i+1], hd[2*i+1] ? 100.0 * hd[2*i] / hd[2*i+1] : 0.0); }
#if defined(STAG) && STAG == 6
    { unsigned hd[512]; hipMemcpy(hd, dbg2, sizeof(hd), hipMemcpyDeviceToHost); for (int i : {0,1,2,8,9,255,256,257,264,511}) printf("blk %d hwid %08x wave %u simd %u cu %u sh %u se %u\n", i, hd[i], hd[i]&15, (hd[i]>>4)&3, (hd[i]>>8)&15, (hd[i]>>12)&1, (hd[i]>>13)&7); }
#endif
    return 0;
}
