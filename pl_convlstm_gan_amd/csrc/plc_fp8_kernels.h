// ConvLSTM gate convolution with fp8 (OCP e4m3) operands on the block-scaled matrix instruction of
// gfx950, v_mfma_scale_f32_32x32x64_f8f6f4 (twice the bf16 rate per clock; the unscaled fp8 MFMA runs
// at the bf16 rate).  fp32 accumulation, fp32 cell state, the same fused epilogue as the bf16 cell
// kernel (sigma/tanh gates, c' = f*c + i*g, h' = o*tanh(c')).
//
// Layouts.  Activations: "CB16" fp8, [image][C/16][H][W][16]: a 16-byte unit holds 16 consecutive
// channels of one pixel.  The instruction contracts K = 64 per issue and every lane supplies 32
// consecutive K bytes of each operand (lanes 0-31: K 0..31, lanes 32-63: K 32..63; A row / B column =
// lane & 31), i.e. TWO units: K unit u = tap * NU + (unit inside the pixel's channels), k-step s takes
// units 4s .. 4s+3, lane half h the units 4s + 2h and 4s + 2h + 1 -- one ds_read_b128 each from the
// [unit][row][col] halo tile.  Weights: packed per lane in exactly that order (plc_pack_lstm_fp8_kernel),
// stationary in registers (cell 2: 7 k-steps x 8 VGPRs instead of 27 x 4 for bf16).
// Scaling.  The weights are multiplied by a power of two chosen from the tensor's largest magnitude
// (largest scaled magnitude in [128, 256), e4m3's top binade) before rounding, so the small values of a
// conv weight keep e4m3's three mantissa bits instead of falling into its subnormals; the instruction's
// per-block scale operand (E8M0) undoes the factor in hardware.  Activations are used as they are
// (|h| < 1, the stem output is a ReLU of O(1..10) values), clamped to e4m3's +-448.
#pragma once
#include "plc_conv_kernel.h"
#include "plc_misc_kernels.h"

typedef int plc_v8i __attribute__((ext_vector_type(8)));

// four floats -> four e4m3 bytes (round to nearest even, clamped to the format's +-448)
__device__ __forceinline__ unsigned plc_cvt_fp8x4(float a, float b, float c, float d) {
    a = fminf(fmaxf(a, -448.0f), 448.0f);
    b = fminf(fmaxf(b, -448.0f), 448.0f);
    c = fminf(fmaxf(c, -448.0f), 448.0f);
    d = fminf(fmaxf(d, -448.0f), 448.0f);
    const int lo = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    return (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(c, d, lo, true);
}

// bf16 CB8 [N][ncb][H][W][8]  ->  fp8 CB16 [N][ceil(ncb/2)][H][W][16]
__global__ void plc_cb8_to_cb16_fp8_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, long long N, int ncb,
                                           long long hw) {
    const int nu = (ncb + 1) / 2;
    const long long total = N * nu * hw;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long pix = idx % hw;
        const int u = (idx / hw) % nu;
        const long long n = idx / (hw * nu);
        unsigned o[4];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int cb = 2 * u + k;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (cb < ncb) v = src[(n * ncb + cb) * hw + pix];
            const float4 a = unpack_bf4(make_uint2(v.x, v.y)), b = unpack_bf4(make_uint2(v.z, v.w));
            o[2 * k] = plc_cvt_fp8x4(a.x, a.y, a.z, a.w);
            o[2 * k + 1] = plc_cvt_fp8x4(b.x, b.y, b.z, b.w);
        }
        dst[idx] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// Gate-conv weights fp32 [4Ch][Cin+Ch][3][3] -> fp8 A fragments [M tile][k-step][lane][32 bytes] and the
// scale word.  PLC_FP8_PACK_BLOCKS workgroups of 1024 threads per conv: every workgroup finds max|w| of
// the whole tensor by itself (a few hundred KB from L2; cheaper than a second launch or a grid-wide
// reduction) and then scales, rounds and scatters its share of the fragments.  scale_word[0] = the E8M0
// byte of 2^-e replicated four times (the instruction's scale operand), scale_word[1] = e.
constexpr int PLC_FP8_PACK_BLOCKS = 16;
struct PackFp8Job {
    const float* w;
    unsigned* wpk;
    int* scale_word;
    int Cin, Ch;
};
__global__ __launch_bounds__(1024) void plc_pack_lstm_fp8_kernel(const PackFp8Job j0, const PackFp8Job j1) {
    const PackFp8Job jb = blockIdx.x < PLC_FP8_PACK_BLOCKS ? j0 : j1;
    const int part = blockIdx.x % PLC_FP8_PACK_BLOCKS;
    __shared__ float red[16];
    __shared__ int e_sh;
    const int cout = 4 * jb.Ch, cin = jb.Cin + jb.Ch, nu = cin / 16;
    const int units = 9 * nu, nks = (units + 3) / 4, n_mt = jb.Ch / 8;
    const long long nw = (long long)cout * cin * 9;
    float m = 0.0f;
    for (long long i = threadIdx.x; i < nw; i += 1024) m = fmaxf(m, fabsf(jb.w[i]));
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float mm = 0.0f;
        for (int k = 0; k < 16; ++k) mm = fmaxf(mm, red[k]);
        // max|w| = 1.m * 2^k  ->  e = 7 - k puts the largest scaled magnitude into [128, 256)
        const int E = (int)((__float_as_uint(mm) >> 23) & 0xff);
        int e = (mm > 0.0f && isfinite(mm)) ? 134 - E : 0;
        e = e > 100 ? 100 : (e < -100 ? -100 : e);
        e_sh = e;
        if (part == 0) {
            jb.scale_word[0] = (int)(0x01010101u * (unsigned)(127 - e));   // the E8M0 byte in all four lanes' positions (unsigned: no overflow)
            jb.scale_word[1] = e;
        }
    }
    __syncthreads();
    const float mul = __builtin_ldexpf(1.0f, e_sh);
    const long long total = (long long)n_mt * nks * 64 * 8;           // dwords: 4 fp8 each
    for (long long idx = part * 1024 + threadIdx.x; idx < total; idx += 1024 * PLC_FP8_PACK_BLOCKS) {
        const int d = idx & 7;                                         // dword inside the lane's 32 bytes
        const int lane = (idx >> 3) & 63;
        const int s = (idx >> 9) % nks;
        const int mt = (idx >> 9) / nks;
        const int r = lane & 31, half = lane >> 5;
        const int u = 4 * s + 2 * half + (d >> 2);
        const int o = plc_row_to_cout(PACK_LSTM, mt, r, jb.Ch);
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (u < units) {
            const int tap = u / nu, cu = u % nu;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ci = 16 * cu + 4 * (d & 3) + k;
                v[k] = jb.w[((long long)o * cin + ci) * 9 + tap] * mul;
            }
        }
        jb.wpk[idx] = plc_cvt_fp8x4(v[0], v[1], v[2], v[3]);
    }
}

struct Fp8CellArgs {
    ConvArgs c;            // H, W, n_in, tiles, src0 = x (fp8 CB16), src1 = h (fp8 CB16), c_in, out0 = h' (bf16 CB8),
                           // out1 = c' (fp32 CB8), out2 = gates (bf16 CB8) when saving, wpk = fp8 fragments, bpk, zero_state
    PlcTensor h8_out;      // h' as fp8 CB16: the next step's operand
    const int* scale_word; // from plc_pack_lstm_fp8_kernel
};

template <int NU0, int NU1, int NMT, int TH>
constexpr int plc_fp8_lds_bytes() {
    return (((NU0 + NU1) * (TH + 2) * 34 + 63) / 64 * 64) * 16;
}

template <int NU0, int NU1, int NMT, int TH, bool SAVE>
__global__ __launch_bounds__(256, 2) void plc_lstm_fp8_kernel(const Fp8CellArgs fa) {
    constexpr int NU = NU0 + NU1, UNITS = 9 * NU, NKS = (UNITS + 3) / 4;
    constexpr int LH = TH + 2, LW = 34, PLANE = LH * LW;
    constexpr int RG = 4 / NMT, RPW = TH / RG, NREP = RPW < 4 ? RPW : 4, PASSES = RPW / NREP;
    constexpr int TILE_UNITS = (NU * PLANE + 63) / 64 * 64, NCHUNK = TILE_UNITS / 64;
    static_assert(NMT == 1 || NMT == 2 || NMT == 4, "NMT must be 1, 2 or 4");
    static_assert(RPW % NREP == 0, "rows must split evenly");
    extern __shared__ __attribute__((aligned(16))) uint4 plc_fp8_tile[];
    const ConvArgs& a = fa.c;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, px = lane & 31;
    const int m = wave % NMT, rg = wave / NMT, mt = m, n_mt = NMT;
    const int H = a.H, W = a.W;
    int n, x0, y0;
    plc_decode_tile(a, TH, blockIdx.x, n, x0, y0);

    // stationary A fragments
    plc_v8i wf[NKS];
    {
        const uint4* wsrc = a.wpk + ((long long)mt * NKS * 64 + lane) * 2;
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
            const uint4 lo = wsrc[s * 128], hi = wsrc[s * 128 + 1];
            wf[s] = (plc_v8i){(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
        }
    }
    const int sa = fa.scale_word[0];
    float4 bv[4];
    {
        const float4* bias4 = (const float4*)a.bpk + mt * 8;
#pragma unroll
        for (int g = 0; g < 4; ++g) bv[g] = bias4[2 * g + half];
    }
    // halo tile by LDS-DMA: 64 consecutive units per wave instruction, zeros outside the image
    {
        const char* s0 = plc_img(a.src0, n, a.n_in);
        const char* s1 = plc_img(a.src1, n, a.n_in);
        const long long plane_b = (long long)H * W * 16;
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        for (int ch = wave_u; ch < NCHUNK; ch += 4) {
            const int u = ch * 64 + lane;
            const int cu = u / PLANE, r = u - cu * PLANE;
            const int ly = r / LW, lx = r - ly * LW;
            const int gy = y0 - 1 + ly, gx = x0 - 1 + lx;
            const bool ok = (cu < NU) & (gy >= 0) & (gy < H) & (gx >= 0) & (gx < W) & !(a.zero_state && cu >= NU0);
            const char* base = cu < NU0 ? s0 + cu * plane_b : s1 + (cu - NU0) * plane_b;
            const char* src = ok ? base + ((long long)gy * W + gx) * 16 : (const char*)&plc_zero16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(plc_fp8_tile + ch * 64), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    const char* lds = (const char*)plc_fp8_tile;
    const long long hw = (long long)H * W;
#pragma unroll 1
    for (int pass = 0; pass < PASSES; ++pass) {
        const int row0 = rg * RPW + pass * NREP;
        float4 cpre[NREP];
#pragma unroll
        for (int r = 0; r < NREP; ++r) {
            const int gy = y0 + row0 + r, gx = x0 + px;
            const bool ok = (gy < H) & (gx < W) & !a.zero_state;
            cpre[r] = plc_gload_f4(ok ? plc_img(a.c_in, n, a.n_in) + (mt * hw + (long long)gy * W + gx) * 32 + half * 16
                                      : (const char*)&plc_zero16);
        }
        f32x16_t acc[NREP];
#pragma unroll
        for (int r = 0; r < NREP; ++r)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                acc[r][4 * g + 0] = bv[g].x; acc[r][4 * g + 1] = bv[g].y; acc[r][4 * g + 2] = bv[g].z; acc[r][4 * g + 3] = bv[g].w;
            }
        const char* rowp = lds + (row0 * LW + px) * 16;
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
            // byte offsets of this k-step's units for lane half 0 / 1 (compile-time each; past the last unit
            // the weights are zero and any tile address does)
            auto unit_off = [](int u) constexpr { return u < UNITS ? ((u % NU) * PLANE + (u / NU / 3) * LW + (u / NU) % 3) * 16 : 0; };
            const int o0 = half ? unit_off(4 * s + 2) : unit_off(4 * s);
            const int o1 = half ? unit_off(4 * s + 3) : unit_off(4 * s + 1);
#pragma unroll
            for (int r = 0; r < NREP; ++r) {
                const uint4 lo = *(const uint4*)(rowp + r * LW * 16 + o0), hi = *(const uint4*)(rowp + r * LW * 16 + o1);
                const plc_v8i bf = (plc_v8i){(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
                acc[r] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf[s], bf, acc[r], 0, 0, 0, sa, 0, 0x7F7F7F7F);
            }
        }
        // The epilogue below runs under a per-lane bounds test.  Left alone, the compiler sinks the
        // MFMAs (whose results only the guarded code uses) into that divergent region together with the
        // moves that set up their per-lane scale operand; the matrix instruction takes its A rows and
        // their scales from ALL 64 lanes, so lanes masked off at that point feed it stale scales and
        // every column of a ragged tile comes out wrong.  Pinning the accumulators here keeps the
        // MFMAs in uniform control flow.
#pragma unroll
        for (int r = 0; r < NREP; ++r) asm volatile("" : "+v"(acc[r]));
        // epilogue: accumulator register 4g + j of lane (half, px) = gate g, hidden channel 8 mt + 4 half + j
#pragma unroll
        for (int r = 0; r < NREP; ++r) {
            const int gy = y0 + row0 + r, gx = x0 + px;
            if (gy < H && gx < W) {
                const long long pix = (long long)gy * W + gx;
                const float cpv[4] = {cpre[r].x, cpre[r].y, cpre[r].z, cpre[r].w};
                float ig[4], fg[4], og[4], gg[4], cn[4], hn[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    ig[j] = fast_sigmoid(acc[r][j]);
                    fg[j] = fast_sigmoid(acc[r][4 + j]);
                    og[j] = fast_sigmoid(acc[r][8 + j]);
                    gg[j] = fast_tanh(acc[r][12 + j]);
                    cn[j] = __builtin_fmaf(fg[j], cpv[j], ig[j] * gg[j]);
                    hn[j] = og[j] * fast_tanh(cn[j]);
                }
                *(float4*)(plc_img(a.out1, n, a.n_in) + (mt * hw + pix) * 32 + half * 16) = make_float4(cn[0], cn[1], cn[2], cn[3]);
                const uint2 hb = pack_bf4(hn[0], hn[1], hn[2], hn[3]);
                *(uint2*)(plc_img(a.out0, n, a.n_in) + (mt * hw + pix) * 16 + half * 8) = hb;
                // the fp8 copy is taken from the bf16-rounded value (what every other consumer of h' sees)
                const float4 hr = unpack_bf4(hb);
                *(unsigned*)(plc_img(fa.h8_out, n, a.n_in) + ((mt >> 1) * hw + pix) * 16 + (mt & 1) * 8 + half * 4) =
                    plc_cvt_fp8x4(hr.x, hr.y, hr.z, hr.w);
                if constexpr (SAVE) {
                    char* gp = plc_img(a.out2, n, a.n_in) + pix * 16 + half * 8;
                    const long long plane_b = hw * 16;
                    *(uint2*)(gp + (long long)(0 * n_mt + mt) * plane_b) = pack_bf4(ig[0], ig[1], ig[2], ig[3]);
                    *(uint2*)(gp + (long long)(1 * n_mt + mt) * plane_b) = pack_bf4(fg[0], fg[1], fg[2], fg[3]);
                    *(uint2*)(gp + (long long)(2 * n_mt + mt) * plane_b) = pack_bf4(og[0], og[1], og[2], og[3]);
                    *(uint2*)(gp + (long long)(3 * n_mt + mt) * plane_b) = pack_bf4(gg[0], gg[1], gg[2], gg[3]);
                }
            }
        }
    }
}
