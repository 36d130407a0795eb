// The generator's two output convs in ONE kernel (generator.py:198-202 of the reference):
//
//   out = conv3x3_{32->1}( relu( conv3x3_{C2->32}( feat * gate ) ) )
//
// Run separately, the first conv writes its 32-channel result (64 B per pixel and frame) for the second to read
// back with a halo, which is more than half of the bytes the pair moves.  Here a workgroup owns a 14 x 30 pixel
// tile of one frame: it stages the gated input on the tile's 18 x 34 neighbourhood into LDS, runs the first conv
// on the 16 x 32 pixels the second conv needs, leaves that result (bf16, zero outside the image: the second
// conv's zero padding) in LDS, and runs the second conv from there.  Training also needs the intermediate
// result (ReLU mask and weight gradient of the backward pass): SAVE_P0 stores the tile's own pixels of it.
//
// The first conv runs exactly the instructions of the separate kernel (plc_conv_mfma_kernel<4,0,1,3,GATED,ACT_CB8>):
// the same packed weights, the bias as the accumulator's initial value, v_mfma_f32_32x32x16_bf16 over the k-steps in
// the order (tap row, channel-block pair, tap column), ReLU and round-to-bf16 -- its result, which the backward pass
// reads, is bit-identical to that kernel's.  The second conv has ONE output channel; on the 32-row instruction it
// would take as many matrix cycles and weight registers as the first.  It runs on v_mfma_f32_16x16x32_bf16 instead
// (one tap per k-step: K = 4 channel blocks x 8 channels; two 16-pixel column tiles per row), half the cycles and
// half the registers.  Its fp32 sums are formed in another order than the separate kernel's, so the output agrees
// with the two-launch path to fp32 rounding (1e-6 relative), not bit for bit; training and inference both run this
// kernel and agree bit for bit.
//
// Widths: C2 (as whole blocks) = 32, i.e. 4 input channel blocks; other widths keep the separate kernels.
#pragma once
#include "plc_conv_kernel.h"

struct PostArgs {
    int H, W;                 // output grid
    int n_in;                 // B (images are n = t * B + b)
    int tiles_x, tiles_y, n_tiles;
    PlcTensor feat;           // CB8 bf16, 4 channel blocks, per frame
    PlcTensor gate;           // CB8 bf16, 4 channel blocks, per batch element (so = 0)
    PlcTensor p0;             // SAVE_P0: CB8 bf16, 4 channel blocks, per frame
    PlcTensor out;            // fp32 plane per frame
    const uint4* wpk0;        // packed A fragments of the first conv (1 M tile, 18 k-steps)
    const float* bpk0;
    const uint4* wpk2;        // ... of the second conv (1 M tile of which row 0 is real, 18 k-steps), as the separate
    const float* bpk2;        // kernel takes them; the 16-row fragments used here are gathered from it
};

constexpr int POST_TH = 14, POST_TW = 30;                 // output pixels of a tile
constexpr int POST_PH = POST_TH + 2, POST_PW = 34;        // first conv's result: 16 rows x 32 columns (+2 zero columns)
constexpr int POST_IH = POST_TH + 4, POST_IW = 34;        // gated input: 18 rows x 34 columns
constexpr int POST_IN_UNITS = 4 * POST_IH * POST_IW;      // 2448
constexpr int POST_P0_UNITS = 4 * POST_PH * POST_PW;      // 2176
constexpr int POST_LDS_BYTES = 2 * (POST_IN_UNITS + POST_P0_UNITS) * 16;      // two groups' tile buffers

// One pass of a 3x3 conv over 4 channel blocks for output rows r0 and r0 + 1 of an LDS tile laid out
// [channel block][rows][LW columns] (16-byte units): the accumulation order of plc_conv_body's fragment loop.
template <int LROWS, int LW>
__device__ __forceinline__ void plc_post_pass(const char* lds, int r0, int px, int half, const bf16x8_t (&wf)[18],
                                              f32x16_t (&acc)[2]) {
    constexpr int PLANE = LROWS * LW;
    // Fragment f = (LDS row rho, channel-block pair cbp, tap column dx) feeds output row r at tap row rho - r.  Only
    // one group of waves computes at a time (one wave per SIMD), so nothing else hides an LDS round trip: the reads
    // run DEPTH fragments ahead of the MFMAs that consume them and sched_barrier pins that order.
    constexpr int NF = 4 * 2 * 3, DEPTH = 5;
    const char* base = lds + ((r0 * LW + px) + half * PLANE) * 16;
    bf16x8_t fr[NF];
#pragma unroll
    for (int f = 0; f < DEPTH; ++f) {
        const int rho = f / 6, cbp = (f % 6) / 3, dx = f % 3;
        fr[f] = *(const bf16x8_t*)(base + ((rho * LW + dx) + cbp * 2 * PLANE) * 16);
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        if (f + DEPTH < NF) {
            const int g = f + DEPTH;
            const int rho = g / 6, cbp = (g % 6) / 3, dx = g % 3;
            fr[g] = *(const bf16x8_t*)(base + ((rho * LW + dx) + cbp * 2 * PLANE) * 16);
        }
        const int rho = f / 6, cbp = (f % 6) / 3, dx = f % 3;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int dy = rho - r;
            if (dy >= 0 && dy < 3)
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[(dy * 3 + dx) * 2 + cbp], fr[f], acc[r], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

typedef float plc_f32x4_t __attribute__((ext_vector_type(4)));

// The one-output-channel conv for output rows r0, r0 + 1 (32 pixel columns as two 16-column tiles) from an LDS tile
// [channel block][rows][LW columns]: lane (q = lane / 16, n = lane % 16) supplies channel block q of pixel column n.
// acc[r][nt][0] of the lanes with q == 0 is the output of row r, column 16 nt + n.
template <int LROWS, int LW>
__device__ __forceinline__ void plc_post_pass_1ch(const char* lds, int r0, int lane, const bf16x8_t (&wf)[9],
                                                  plc_f32x4_t (&acc)[2][2]) {
    constexpr int PLANE = LROWS * LW;
    constexpr int NF = 4 * 3 * 2, DEPTH = 5;            // fragment f = (LDS row rho, tap column dx, column tile nt)
    const int q = lane >> 4, n = lane & 15;
    const char* base = lds + ((r0 * LW + n) + q * PLANE) * 16;
    bf16x8_t fr[NF];
#pragma unroll
    for (int f = 0; f < DEPTH; ++f) {
        const int rho = f / 6, dx = (f % 6) / 2, nt = f % 2;
        fr[f] = *(const bf16x8_t*)(base + (rho * LW + dx + 16 * nt) * 16);
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        if (f + DEPTH < NF) {
            const int g = f + DEPTH;
            const int rho = g / 6, dx = (g % 6) / 2, nt = g % 2;
            fr[g] = *(const bf16x8_t*)(base + (rho * LW + dx + 16 * nt) * 16);
        }
        const int rho = f / 6, dx = (f % 6) / 2, nt = f % 2;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int dy = rho - r;
            if (dy >= 0 && dy < 3)
                acc[r][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[dy * 3 + dx], fr[f], acc[r][nt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Two groups of four waves per workgroup take turns: while one group runs the two convs on its tile, the other
// stages the gated input of ITS next tile, and the MFMA pipes never wait for a tile to arrive (two independent
// workgroups per CU drift into staging and computing at the same time; the alternation here is forced).  Every
// wave holds both convs' weights, so a group's roles simply alternate from round to round.
template <bool SAVE_P0>
__global__ __launch_bounds__(512, 1) void plc_post_fused_kernel(const PostArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint4 plc_post_lds[];
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6, half = lane >> 5, px = lane & 31;
    const int group = threadIdx.x >> 8;                   // 0 / 1: which of the two wave groups
    uint4* const tin = plc_post_lds + group * (POST_IN_UNITS + POST_P0_UNITS);   // this group's gated input tile
    uint4* const tp0 = tin + POST_IN_UNITS;                                      // ... and first conv's result
    const int H = a.H, W = a.W;
    const long long plane_b = (long long)H * W * 16;

    // stationary weights of both convs and this lane's bias rows (row 8g + 4 half + j = accumulator 4g + j)
    bf16x8_t wf0[18], wf2[9];
#pragma unroll
    for (int s = 0; s < 18; ++s) wf0[s] = __builtin_bit_cast(bf16x8_t, a.wpk0[s * 64 + lane]);
    // second conv, tap t: lane (q, m) holds the 8 channels of block q of output row m, which the 32-row packing
    // keeps as unit u = 4 t + q (k-step u / 2, lane half u % 2) of row m
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int u = 4 * t + (lane >> 4);
        wf2[t] = __builtin_bit_cast(bf16x8_t, a.wpk2[(u >> 1) * 64 + (u & 1) * 32 + (lane & 15)]);
    }
    float4 b0[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) b0[g] = ((const float4*)a.bpk0)[2 * g + half];
    const float bias2 = a.bpk2[0];                        // (row 0 of the packed bias: the one output channel's)
    // the two columns past the first conv's 32 result columns only feed output columns that are discarded
    for (int u = tid; u < 4 * POST_PH * 2; u += 256) {
        const int cb = u / (POST_PH * 2), r = (u / 2) % POST_PH, c = 32 + (u & 1);
        tp0[(cb * POST_PH + r) * POST_PW + c] = make_uint4(0, 0, 0, 0);
    }
    const char* const z16 = (const char*)&plc_zero16;
    // The workgroup's tiles are blockIdx.x, + gridDim.x, ...; the k-th of them belongs to group k & 1.  Round k:
    // group k & 1 stages tile k, the other group computes tile k - 1 (first conv, a barrier, second conv).  Every wave
    // passes the same two barriers per round.
    const int my_tiles = a.n_tiles > (int)blockIdx.x ? (a.n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    auto coords = [&](int k, int& n, int& x0, int& y0) {
        const int tile = blockIdx.x + k * gridDim.x;
        const int tx = tile % a.tiles_x, ty = (tile / a.tiles_x) % a.tiles_y;
        n = tile / (a.tiles_x * a.tiles_y);
        x0 = tx * POST_TW;
        y0 = ty * POST_TH;
    };
    // gated input on rows y0-2 .. y0+15, columns x0-2 .. x0+31, product rounded to bf16 once.  All of a thread's
    // loads (10 units x 2 tensors) are in flight together: one memory round trip per tile, which the other
    // group's two convs cover.
    constexpr int NIT = (POST_IN_UNITS + 255) / 256;      // 10
    auto stage_tile = [&](int k) {
        const int i0 = 0;
        int n, x0, y0;
        coords(k, n, x0, y0);
        const char* fsrc = plc_img(a.feat, n, a.n_in);
        const char* gsrc = plc_img(a.gate, n, a.n_in);
        uint4 f[NIT], g[NIT];
#pragma unroll
        for (int q = 0; q < NIT; ++q) {
            const int u = tid + (i0 + q) * 256;
            const int cb = u / (POST_IH * POST_IW), r = u - cb * (POST_IH * POST_IW);
            const int ly = r / POST_IW, lx = r - ly * POST_IW;
            const int gy = y0 - 2 + ly, gx = x0 - 2 + lx;
            const bool ok = (u < POST_IN_UNITS) & (gy >= 0) & (gy < H) & (gx >= 0) & (gx < W);
            const long long off = cb * plane_b + ((long long)gy * W + gx) * 16;
            f[q] = plc_gload16(ok ? fsrc + off : z16);
            g[q] = plc_gload16(ok ? gsrc + off : z16);
        }
#pragma unroll
        for (int q = 0; q < NIT; ++q) {
            const int u = tid + (i0 + q) * 256;
            const float4 f0 = unpack_bf4(make_uint2(f[q].x, f[q].y)), f1 = unpack_bf4(make_uint2(f[q].z, f[q].w));
            const float4 g0 = unpack_bf4(make_uint2(g[q].x, g[q].y)), g1 = unpack_bf4(make_uint2(g[q].z, g[q].w));
            const uint2 lo = pack_bf4(f0.x * g0.x, f0.y * g0.y, f0.z * g0.z, f0.w * g0.w);
            const uint2 hi = pack_bf4(f1.x * g1.x, f1.y * g1.y, f1.z * g1.z, f1.w * g1.w);
            if (u < POST_IN_UNITS) tin[u] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
    };
    // first conv + ReLU on rows y0-1 .. y0+14, columns x0-1 .. x0+30 -> tp0 (zero outside the image)
    auto conv_first = [&](int k) {
        int n, x0, y0;
        coords(k, n, x0, y0);
        char* p0g = SAVE_P0 ? plc_img(a.p0, n, a.n_in) : nullptr;
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            const int r0 = wave * 4 + pass * 2;            // rows of the result tile
            f32x16_t acc[2];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    acc[r][4 * g + 0] = b0[g].x; acc[r][4 * g + 1] = b0[g].y;
                    acc[r][4 * g + 2] = b0[g].z; acc[r][4 * g + 3] = b0[g].w;
                }
            plc_post_pass<POST_IH, POST_IW>((const char*)tin, r0, px, half, wf0, acc);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int pr = r0 + r, gy = y0 - 1 + pr, gx = x0 - 1 + px;
                const bool inside = (gy >= 0) & (gy < H) & (gx >= 0) & (gx < W);
                // the tile's own pixels of the result: rows 1 .. 14, columns 1 .. 30
                const bool own = inside & (pr >= 1) & (pr <= POST_TH) & (px >= 1) & (px <= POST_TW);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 v = pack_bf4(fmaxf(acc[r][4 * g + 0], 0.0f), fmaxf(acc[r][4 * g + 1], 0.0f),
                                       fmaxf(acc[r][4 * g + 2], 0.0f), fmaxf(acc[r][4 * g + 3], 0.0f));
                    if (SAVE_P0 && own) *(uint2*)(p0g + g * plane_b + ((long long)gy * W + gx) * 16 + half * 8) = v;
                    if (!inside) v = make_uint2(0u, 0u);
                    *(uint2*)((char*)tp0 + ((g * POST_PH + pr) * POST_PW + px) * 16 + half * 8) = v;
                }
            }
        }
    };
    // second conv on the tile's 14 x 30 pixels, from tp0
    auto conv_second = [&](int k) {
        int n, x0, y0;
        coords(k, n, x0, y0);
        float* og = (float*)plc_img(a.out, n, a.n_in);
        const int q = lane >> 4, col = lane & 15;
#pragma unroll 1
        for (int pair = wave; pair < POST_TH / 2; pair += 4) {
            const int r0 = 2 * pair;                       // output rows y0 + r0, y0 + r0 + 1
            plc_f32x4_t acc[2][2];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    acc[r][nt][0] = q == 0 ? bias2 : 0.0f;     // accumulator j of lane (q, .) is output row 4 q + j
                    acc[r][nt][1] = 0.0f; acc[r][nt][2] = 0.0f; acc[r][nt][3] = 0.0f;
                }
            plc_post_pass_1ch<POST_PH, POST_PW>((const char*)tp0, r0, lane, wf2, acc);
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int gy = y0 + r0 + r, c = 16 * nt + col, gx = x0 + c;
                    if (q == 0 && c < POST_TW && gy < H && gx < W) og[(long long)gy * W + gx] = acc[r][nt][0];
                }
        }
    };

#pragma unroll 1
    for (int k = 0; k <= my_tiles; ++k) {
        const bool staging = (k & 1) == group;            // (wave-uniform)
        if (staging) {
            if (k < my_tiles) stage_tile(k);
        } else if (k >= 1) {
            conv_first(k - 1);
        }
        __syncthreads();
        if (!staging && k >= 1) conv_second(k - 1);
        __syncthreads();
    }
}
