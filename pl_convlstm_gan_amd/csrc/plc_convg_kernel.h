// Implicit-GEMM KSxKS convolution with RUN-TIME channel counts: the path taken by every layer width
// the weight-stationary instantiations of plc_conv_kernel.h are not compiled for (e.g. the
// reference's class default hidden_dims = [32, 64]).  Same GEMM view, same LDS tile layout, same
// MFMA (v_mfma_f32_32x32x16_bf16), same epilogues (plc_epilogue) and therefore the same rounding
// points as the specialised kernels; what differs is where the operands come from:
//   * a workgroup (4 waves) owns an 8 x 32 pixel tile of one image and ONE group of up to four
//     32-row output tiles (blockIdx.y); with fewer than four output tiles in the group the waves
//     split the tile's rows instead, so all four always work;
//   * the input channel blocks are contracted in chunks of PLC_G_CHUNK blocks: each chunk's halo
//     tile is staged into LDS (through registers: concat of two sources, optional gate product,
//     zero padding), so LDS use does not grow with the channel count;
//   * the weights are NOT kept in registers: every k-step's A fragment (16 bytes per lane, packed in
//     chunk-major order by plc_pack_weight_kernel) is fetched from global memory / L2 three k-steps
//     ahead of its MFMAs and used for all of the wave's rows.
// It is slower per FLOP than the specialised kernels (A fragments cross the L2 once per 8 pixel rows
// instead of once per workgroup, the tile is re-staged per output group) and any width that matters
// for throughput should get an instantiation there; this one makes every width work.
#pragma once
#include "plc_conv_kernel.h"

constexpr int PLC_G_CHUNK = 8;     // input channel blocks (of 8 channels) contracted per LDS tile
constexpr int PLC_G_TH = 8;        // tile rows

// k-steps of one M tile in chunk-major order (host and device).
__host__ __device__ inline int plc_g_nks(int ncb, int ks) {
    const int taps = ks * ks;
    const int full = ncb / PLC_G_CHUNK, rest = ncb % PLC_G_CHUNK;
    return full * ((taps * PLC_G_CHUNK + 1) / 2) + (rest ? (taps * rest + 1) / 2 : 0);
}

template <int KS>
constexpr int plc_convg_lds_bytes() {
    return PLC_G_CHUNK * (PLC_G_TH + 2 * (KS / 2)) * (32 + 2 * (KS / 2)) * 16;
}

template <int KS, int STAGE, int EPI>
__global__ __launch_bounds__(256, 2) void plc_convg_kernel(const ConvArgs a) {
    constexpr int TH = PLC_G_TH, PAD = KS / 2, LH = TH + 2 * PAD, LW = 32 + 2 * PAD, PLANE = LH * LW;
    constexpr int TAPS = KS * KS;
    constexpr bool IS_LSTM = (EPI == EPI_LSTM || EPI == EPI_LSTM_SAVE);
    static_assert(STAGE == STAGE_CB8 || STAGE == STAGE_CB8_GATED, "the generic conv stages CB8 tensors");
    extern __shared__ __attribute__((aligned(16))) uint4 plc_convg_tile[];
    uint4* const tile = plc_convg_tile;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, px = lane & 31;
    const int ncb0 = a.g_ncb0, ncb = a.g_ncb0 + a.g_ncb1, n_mt = a.g_nmt;
    const int H = a.H, W = a.W;

    // output tiles of this workgroup's group, and how the four waves share them
    const int mt0 = blockIdx.y * 4;
    const int left = n_mt - mt0;
    const int nm = left >= 3 ? 4 : left;           // waves side by side over output tiles: 1, 2 or 4
    const int m = wave % nm, rg = wave / nm;
    const int rows = TH / (4 / nm);                // pixel rows of this wave: 2, 4 or 8
    const int mt = mt0 + m;
    const bool active = mt < n_mt;                 // (a group of three tiles leaves wave 3 without one)

    int n, x0, y0;
    plc_decode_tile(a, TH, blockIdx.x, n, x0, y0);
    const char* s0 = plc_img(a.src0, n, a.n_in);
    const char* s1 = a.g_ncb1 > 0 ? plc_img(a.src1, n, a.n_in) : nullptr;
    const char* gt = STAGE == STAGE_CB8_GATED ? plc_img(a.gate, n, a.n_in) : nullptr;
    const long long plane_b = (long long)H * W * 16;

    f32x16_t acc[TH];
    {
        const float4* bias4 = (const float4*)a.bpk + (active ? mt : 0) * 8;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 b = bias4[2 * g + half];
#pragma unroll
            for (int r = 0; r < TH; ++r) {
                acc[r][4 * g + 0] = b.x; acc[r][4 * g + 1] = b.y; acc[r][4 * g + 2] = b.z; acc[r][4 * g + 3] = b.w;
            }
        }
    }

    const uint4* const wbase = a.wpk + (long long)(active ? mt : 0) * plc_g_nks(ncb, KS) * 64 + lane;
    int sbase = 0;
    for (int cb0 = 0; cb0 < ncb; cb0 += PLC_G_CHUNK) {
        const int nb = ncb - cb0 < PLC_G_CHUNK ? ncb - cb0 : PLC_G_CHUNK;
        __syncthreads();                                       // every wave has finished reading the previous chunk
        for (int u = tid; u < nb * PLANE; u += 256) {
            const int cbl = u / PLANE, r = u - cbl * PLANE;
            const int ly = r / LW, lx = r - ly * LW;
            const int gy = y0 - PAD + ly, gx = x0 - PAD + lx, cb = cb0 + cbl;
            const bool ok = (gy >= 0) & (gy < H) & (gx >= 0) & (gx < W) & !(a.zero_state && cb >= ncb0);
            const long long pix = ((long long)gy * W + gx) * 16;
            const char* src = cb < ncb0 ? s0 + cb * plane_b : s1 + (cb - ncb0) * plane_b;
            uint4 v = plc_gload16(ok ? src + pix : (const char*)&plc_zero16);
            if constexpr (STAGE == STAGE_CB8_GATED) {
                // the product is rounded to bf16 once, as the specialised gated staging does
                const uint4 g = plc_gload16(ok ? gt + cb * plane_b + pix : (const char*)&plc_zero16);
                const float4 f0 = unpack_bf4(make_uint2(v.x, v.y)), f1 = unpack_bf4(make_uint2(v.z, v.w));
                const float4 g0 = unpack_bf4(make_uint2(g.x, g.y)), g1 = unpack_bf4(make_uint2(g.z, g.w));
                const uint2 lo = pack_bf4(f0.x * g0.x, f0.y * g0.y, f0.z * g0.z, f0.w * g0.w);
                const uint2 hi = pack_bf4(f1.x * g1.x, f1.y * g1.y, f1.z * g1.z, f1.w * g1.w);
                v = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
            tile[u] = v;
        }
        __syncthreads();
        const int nks = (TAPS * nb + 1) / 2;
        if (active) {
            // this lane's K unit of k-step s is u = 2s + half = tap * nb + (block inside the chunk)
            int tap = half / nb, cbl = half % nb;
            const char* const row0 = (const char*)tile + ((rg * rows) * LW + px) * 16;
            const uint4* const wsrc = wbase + (long long)sbase * 64;
            bf16x8_t af[4];
#pragma unroll
            for (int k = 0; k < 3; ++k) af[k] = __builtin_bit_cast(bf16x8_t, wsrc[(k < nks ? k : 0) * 64]);
            for (int sb = 0; sb < nks; sb += 4) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int s = sb + k;
                    if (s < nks) {
                        const int sn = s + 3 < nks ? s + 3 : s;          // (the tail re-reads a fragment it does not use)
                        af[(k + 3) & 3] = __builtin_bit_cast(bf16x8_t, wsrc[(long long)sn * 64]);
                        // past the chunk's last unit (odd unit count) the packed weights are zero: any tile address does
                        const int off = tap < TAPS ? ((cbl * LH + tap / KS) * LW + tap % KS) * 16 : 0;
                        const char* const bp = row0 + off;
#pragma unroll
                        for (int r = 0; r < TH; ++r)
                            if (r < rows)
                                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[k], *(const bf16x8_t*)(bp + r * LW * 16),
                                                                                 acc[r], 0, 0, 0);
                        cbl += 2;
                        while (cbl >= nb) { cbl -= nb; ++tap; }
                    }
                }
            }
        }
        sbase += nks;
    }

    if (!active) return;
    // (the epilogues test image bounds per lane; the accumulators are pinned here so that no MFMA, which
    // takes its A rows from all 64 lanes, can be moved under that divergent test)
#pragma unroll
    for (int r = 0; r < TH; ++r) asm volatile("" : "+v"(acc[r]));
#pragma unroll
    for (int r = 0; r < TH; ++r) {
        if (r < rows) {
            const int gy = y0 + rg * rows + r, gx = x0 + px;
            float4 cp = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (IS_LSTM) {
                if (gy < H && gx < W && !a.zero_state)
                    cp = *(const float4*)(plc_img(a.c_in, n, a.n_in) + ((long long)mt * H * W + (long long)gy * W + gx) * 32 + half * 16);
            }
            plc_epilogue<EPI>(a, acc[r], n, mt, n_mt, gy, gx, half, cp);
        }
    }
}
