// Pointwise pieces of the backward pass (HBM-bound, one pass each) on CB8 tensors.
// One thread handles one 16-byte unit = 8 channels of one pixel.
#pragma once
#include "plc_common.h"
#include "plc_misc_kernels.h"
#include "plc_lstm_bwd_unit.h"

// (block, n_blocks: this workgroup's position in the grid-stride loop over the launch's units)
__device__ __forceinline__ void plc_lstm_pointwise_bwd_body(const LstmBwdArgs& a, int block, int n_blocks) {
    const long long hw = (long long)a.H * a.W;
    const long long total = (long long)a.B * a.ncb * hw;
    for (long long idx = block * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)n_blocks * blockDim.x) {
        const long long pix = idx % hw;
        const int cb = (idx / hw) % a.ncb;
        const long long b = idx / (hw * a.ncb);
        const long long u = cb * hw + pix;        // unit inside an ncb-block image
        const long long su = (b * a.ncb) * hw + u;              // unit index in [B][ncb] tensors
        const long long gu = (b * 4 * a.ncb) * hw + u;          // unit index of gate 0 in [B][4 ncb]
        F8 zi, zf, zo, zg, dcp;
        plc_lstm_bwd_unit(a, b, u, zi, zf, zo, zg, dcp);
        st_bf8(a.dz + (gu + 0 * a.ncb * hw) * 16, zi);
        st_bf8(a.dz + (gu + 1 * a.ncb * hw) * 16, zf);
        st_bf8(a.dz + (gu + 2 * a.ncb * hw) * 16, zo);
        st_bf8(a.dz + (gu + 3 * a.ncb * hw) * 16, zg);
        if (a.dc_bf16) st_bf8(a.dc_out + su * 16, dcp);
        else st_f8(a.dc_out + su * 32, dcp);
    }
}

__global__ void plc_lstm_pointwise_bwd_kernel(const LstmBwdArgs a) {
    plc_lstm_pointwise_bwd_body(a, blockIdx.x, gridDim.x);
}

// Two independent pointwise backward passes in one launch (cell 2 of step t - 1 and cell 1 of step t both
// wait for the same data-gradient conv and for nothing else): workgroups [0, n_first) take `first`.
__global__ void plc_lstm_pointwise_bwd_pair_kernel(const LstmBwdArgs first, const LstmBwdArgs second, int n_first) {
    if ((int)blockIdx.x < n_first) plc_lstm_pointwise_bwd_body(first, blockIdx.x, n_first);
    else plc_lstm_pointwise_bwd_body(second, blockIdx.x - n_first, gridDim.x - n_first);
}

// Gating backward, once per call, over (b, channel block, pixel); loops over the T frames:
//   d_pre[t,b] = d_feat[t,b] * gate[b]                      (gradient of the un-gated feature map)
//   dG[b]      = sum_t d_feat[t,b] * feat_pre[t,b]
//   d_lu_pre[b]  = dG * gd * gl * (1 - gl)                  (through gate = gd * gl and the sigmoids)
//   d_dem_pre[b] = dG * gl * gd * (1 - gd)
// Images are enumerated t-major (n = t*B + b); every tensor is CB8 bf16 with ncb blocks.
struct GateBwdArgs {
    int B, T, ncb, H, W;
    const char* d_feat;     // [T*B]
    const char* feat_pre;   // [T*B]  feature map before gating
    const char* gate;       // [B]    gd * gl
    const char* gate_dem;   // [B]    gd
    const char* gate_lu;    // [B]    gl
    char* d_pre;            // [T*B] or null (not stored)
    char* d_lu_pre;         // [B]
    char* d_dem_pre;        // [B]
};

__global__ void plc_gate_bwd_kernel(const GateBwdArgs a) {
    const long long hw = (long long)a.H * a.W;
    const long long img = (long long)a.ncb * hw;       // units per image
    const long long total = (long long)a.B * img;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long b = idx / img, u = idx % img;
        const F8 g = ld_bf8(a.gate + idx * 16);
        F8 acc;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc.v[j] = 0.0f;
        for (int t = 0; t < a.T; ++t) {
            const long long o = (((long long)t * a.B + b) * img + u) * 16;
            const F8 df = ld_bf8(a.d_feat + o), fp = ld_bf8(a.feat_pre + o);
            F8 dp;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                dp.v[j] = df.v[j] * g.v[j];
                acc.v[j] += df.v[j] * fp.v[j];
            }
            if (a.d_pre) st_bf8(a.d_pre + o, dp);      // null: the consumer forms d_feat * gate itself
        }
        const F8 gd = ld_bf8(a.gate_dem + idx * 16), gl = ld_bf8(a.gate_lu + idx * 16);
        F8 dl, dd;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            dl.v[j] = acc.v[j] * gd.v[j] * gl.v[j] * (1.0f - gl.v[j]);
            dd.v[j] = acc.v[j] * gl.v[j] * gd.v[j] * (1.0f - gd.v[j]);
        }
        st_bf8(a.d_lu_pre + idx * 16, dl);
        st_bf8(a.d_dem_pre + idx * 16, dd);
    }
}

// Backward of ReLU(PixelShuffle(2)(conv)):  d_conv[4c + 2i + j][y][x] = y_up[c][2y+i][2x+j] > 0 ?
// d_up[c][2y+i][2x+j] : 0.   d_up / y_up: CB8 bf16 [N][C/8][2H][2W]; d_conv: CB8 bf16 [N][4C/8][H][W].
__global__ void plc_pixel_unshuffle_mask_kernel(const char* __restrict__ d_up, const char* __restrict__ y_up,
                                                char* __restrict__ d_conv, int N, int C, int H, int W) {
    // One thread per (image, up-sampled channel block, low-res pixel): it reads the 2x2 quad of
    // 16-byte units of d_up and y_up (every byte used) and writes the four conv-channel units
    // 4*cb + k, k = 0..3, which hold channels c = 2k, 2k+1 of the block at the four quad positions.
    const long long hw = (long long)H * W;
    const int ucb = C / 8;
    const long long total = (long long)N * ucb * hw;
    const long long W2 = 2 * W, hw2 = 4 * hw;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int x = idx % W;
        const int y = (idx / W) % H;
        const int cb = (idx / hw) % ucb;
        const long long n = idx / (hw * ucb);
        const long long up0 = ((n * ucb + cb) * hw2 + (long long)(2 * y) * W2 + 2 * x) * 16;
        F8 g[4];   // g[sub].v[c] = masked gradient of channel c at quad position sub = 2i + j
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            const long long o = up0 + ((long long)(sub >> 1) * W2 + (sub & 1)) * 16;
            const F8 dv = ld_bf8(d_up + o), yv = ld_bf8(y_up + o);
#pragma unroll
            for (int c = 0; c < 8; ++c) g[sub].v[c] = yv.v[c] > 0.0f ? dv.v[c] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            F8 r;   // conv channels 8k..8k+7 of this block's 32: channel 4c + sub, c = 2k, 2k+1
#pragma unroll
            for (int j = 0; j < 8; ++j) r.v[j] = g[j & 3].v[2 * k + (j >> 2)];
            st_bf8(d_conv + (((n * (4 * ucb) + 4 * cb + k) * hw) + (long long)y * W + x) * 16, r);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Gradients with respect to the rollout's inputs (rain, DEM, land use).  Training does not ask for them (the
// data loader's tensors need none), so these are plain one-thread-per-output kernels: small, any width, fp32
// accumulation, and off the step's path unless an input has requires_grad.
// ---------------------------------------------------------------------------------------------
// Stem (generator.py:152-160: ReLU(conv3x3([rain, row, col]))): d rain[b][t][y][x] = sum over stem channels c and
// taps of w[c][0][ky][kx] * dz[c][y - ky + 1][x - kx + 1], dz = the x-part of d[x_t | h1_{t-1}] where the stem's
// output is positive.  dcat: CB8 bf16, image n = t*B + b at n*cat_img bytes, the stem channels first;
// xs: the stem's output, CB8 bf16 [T*B][ncb]; w: fp32 [R1][3][3][3]; d_rain: fp32 [B][T][H][W].
__global__ void plc_stem_input_grad_kernel(const char* __restrict__ dcat, long long cat_img, const char* __restrict__ xs,
                                           long long xs_img, const float* __restrict__ w, float* __restrict__ d_rain,
                                           int B, int T, int R1, int H, int W) {
    const long long hw = (long long)H * W, total = (long long)B * T * hw;
    const int ncb = (R1 + 7) / 8;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int x = idx % W;
        const int y = (idx / W) % H;
        const long long n = idx / hw;
        const int t = n / B, b = n % B;
        float acc = 0.0f;
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = y - ky + 1;
            if (yy < 0 || yy >= H) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int xx = x - kx + 1;
                if (xx < 0 || xx >= W) continue;
                for (int cb = 0; cb < ncb; ++cb) {
                    const long long o = ((long long)cb * hw + (long long)yy * W + xx) * 16;
                    const F8 g = ld_bf8(dcat + n * cat_img + o), a = ld_bf8(xs + n * xs_img + o);
                    for (int j = 0; j < 8 && 8 * cb + j < R1; ++j)
                        if (a.v[j] > 0.0f) acc = __builtin_fmaf(w[(8 * cb + j) * 27 + ky * 3 + kx], g.v[j], acc);
                }
            }
        }
        d_rain[((long long)b * T + t) * hw + (long long)y * W + x] = acc;
    }
}

// First conv of an attention branch (attention.py:8-12: conv3x3(aux -> C/2), ReLU, conv1x1, sigmoid), data
// gradient at the output grid: d_hr[b][c][y][x] = sum over mid channels m and taps of
// w0[m][c][ky][kx] * d_mid[b][m][y - ky + 1][x - kx + 1]  (d_mid already carries the ReLU's mask).
// d_mid: CB8 bf16 [B][ceil(M/8)]; w0: fp32 [M][cin][3][3]; d_hr: fp32 [B][cin][H][W].
__global__ void plc_aux_input_grad_kernel(const char* __restrict__ d_mid, const float* __restrict__ w0,
                                          float* __restrict__ d_hr, int B, int M, int cin, int H, int W) {
    const long long hw = (long long)H * W, total = (long long)B * cin * hw;
    const int mcb = (M + 7) / 8;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int x = idx % W;
        const int y = (idx / W) % H;
        const int c = (idx / hw) % cin;
        const long long b = idx / (hw * cin);
        float acc = 0.0f;
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = y - ky + 1;
            if (yy < 0 || yy >= H) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int xx = x - kx + 1;
                if (xx < 0 || xx >= W) continue;
                for (int cb = 0; cb < mcb; ++cb) {
                    const F8 g = ld_bf8(d_mid + (((b * mcb + cb) * hw) + (long long)yy * W + xx) * 16);
                    for (int j = 0; j < 8 && 8 * cb + j < M; ++j)
                        acc = __builtin_fmaf(w0[((long long)(8 * cb + j) * cin + c) * 9 + ky * 3 + kx], g.v[j], acc);
                }
            }
        }
        d_hr[idx] = acc;
    }
}

// Backward of plc_resize_cb8_kernel (bilinear, align_corners False), gather form: one thread per
// SOURCE unit sums w(dst, src) * d_dst over the destination pixels whose interpolation footprint
// contains it.  The candidate destination range comes from inverting the coordinate map, padded by
// one on each side, and every candidate is decided by re-evaluating the forward's own coordinate
// function, so forward and backward use identical weights and no atomics are needed.
// d_dst: CB8 bf16 [planes][Ho][Wo]; d_src: CB8 bf16 [planes][Hi][Wi]; scales as in the forward (in / out).
__device__ __forceinline__ void plc_resize_dst_range(int i, float scale, int in_size, int out_size, int& lo, int& hi) {
    lo = i == 0 ? 0 : (int)floorf(((float)i - 0.5f) / scale - 0.5f) - 1;
    hi = i == in_size - 1 ? out_size - 1 : (int)ceilf(((float)i + 1.5f) / scale - 0.5f) + 1;
    lo = lo < 0 ? 0 : lo;
    hi = hi > out_size - 1 ? out_size - 1 : hi;
}

__global__ void plc_resize_cb8_bwd_kernel(const uint4* __restrict__ d_dst, uint4* __restrict__ d_src, long long planes,
                                          int Hi, int Wi, int Ho, int Wo, float sh, float sw) {
    const long long hwi = (long long)Hi * Wi, hwo = (long long)Ho * Wo;
    const long long total = planes * hwi;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int ix = idx % Wi;
        const int iy = (idx / Wi) % Hi;
        const long long plane = idx / hwi;
        int oy_lo, oy_hi, ox_lo, ox_hi;
        plc_resize_dst_range(iy, sh, Hi, Ho, oy_lo, oy_hi);
        plc_resize_dst_range(ix, sw, Wi, Wo, ox_lo, ox_hi);
        const char* p = (const char*)(d_dst + plane * hwo);
        F8 acc;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc.v[j] = 0.0f;
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            int y0, y1;
            float wy;
            plc_bilinear_coord(oy, sh, Hi, y0, y1, wy);
            const float ky = (y0 == iy ? 1.0f - wy : 0.0f) + (y1 == iy ? wy : 0.0f);
            if (ky == 0.0f) continue;
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                int x0, x1;
                float wx;
                plc_bilinear_coord(ox, sw, Wi, x0, x1, wx);
                const float kx = (x0 == ix ? 1.0f - wx : 0.0f) + (x1 == ix ? wx : 0.0f);
                if (kx == 0.0f) continue;
                const F8 g = ld_bf8(p + ((long long)oy * Wo + ox) * 16);
                const float k = ky * kx;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc.v[j] = __builtin_fmaf(k, g.v[j], acc.v[j]);
            }
        }
        st_bf8((char*)(d_src + idx), acc);
    }
}

// The same for an fp32 NCHW map, both resampling modes of plc_resize_nchw_to_cb8_kernel (mode 0: bilinear;
// mode 1: legacy 'nearest', whose backward sums the destination pixels that read the source pixel).
// d_dst: fp32 [planes][Ho][Wo]; d_src: fp32 [planes][Hi][Wi]; scales = in / out as in the forward.
__global__ void plc_resize_nchw_bwd_kernel(const float* __restrict__ d_dst, float* __restrict__ d_src, long long planes,
                                           int Hi, int Wi, int Ho, int Wo, int mode, float sh, float sw) {
    const long long hwi = (long long)Hi * Wi, hwo = (long long)Ho * Wo;
    const long long total = planes * hwi;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int ix = idx % Wi;
        const int iy = (idx / Wi) % Hi;
        const float* p = d_dst + (idx / hwi) * hwo;
        int oy_lo, oy_hi, ox_lo, ox_hi;
        float acc = 0.0f;
        if (mode == 0) {
            plc_resize_dst_range(iy, sh, Hi, Ho, oy_lo, oy_hi);
            plc_resize_dst_range(ix, sw, Wi, Wo, ox_lo, ox_hi);
            for (int oy = oy_lo; oy <= oy_hi; ++oy) {
                int y0, y1;
                float wy;
                plc_bilinear_coord(oy, sh, Hi, y0, y1, wy);
                const float ky = (y0 == iy ? 1.0f - wy : 0.0f) + (y1 == iy ? wy : 0.0f);
                if (ky == 0.0f) continue;
                for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                    int x0, x1;
                    float wx;
                    plc_bilinear_coord(ox, sw, Wi, x0, x1, wx);
                    const float kx = (x0 == ix ? 1.0f - wx : 0.0f) + (x1 == ix ? wx : 0.0f);
                    if (kx != 0.0f) acc = __builtin_fmaf(ky * kx, p[(long long)oy * Wo + ox], acc);
                }
            }
        } else {
            // destination d reads source min(floor(d * scale), in - 1): candidates around [i / scale, (i + 1) / scale)
            auto range = [](int i, float scale, int in_size, int out_size, int& lo, int& hi) {
                lo = (int)floorf((float)i / scale) - 1;
                hi = i == in_size - 1 ? out_size - 1 : (int)ceilf((float)(i + 1) / scale) + 1;
                lo = lo < 0 ? 0 : lo;
                hi = hi > out_size - 1 ? out_size - 1 : hi;
            };
            range(iy, sh, Hi, Ho, oy_lo, oy_hi);
            range(ix, sw, Wi, Wo, ox_lo, ox_hi);
            for (int oy = oy_lo; oy <= oy_hi; ++oy) {
                int sy = (int)floorf((float)oy * sh);
                sy = sy < Hi - 1 ? sy : Hi - 1;
                if (sy != iy) continue;
                for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                    int sx = (int)floorf((float)ox * sw);
                    sx = sx < Wi - 1 ? sx : Wi - 1;
                    if (sx == ix) acc += p[(long long)oy * Wo + ox];
                }
            }
        }
        d_src[idx] = acc;
    }
}
