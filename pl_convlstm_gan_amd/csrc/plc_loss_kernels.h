// CombinedLoss and its gradient with respect to the prediction, as three launches
// (src/losses/combined_loss.py:64-191 of the reference):
//
//   plc_loss_pool_kernel      area-average of pred onto the LR grid (adaptive bins), |coarse - lr|
//                             partial sums and sign(coarse - lr) / bin_area per LR cell; skipped when
//                             the two grids are equal (every bin is one pixel)
//   plc_loss_main_kernel      one pass over pred: |d/dx|, |d/dy|, |d/dt| partial sums and the whole
//                             gradient of the conserve / smooth / temporal terms per pixel
//   plc_loss_finalize_kernel  fixed-order sum of the partials, the rain-gauge term (a few thousand
//                             elements), the five scalars, and the gauge term's gradient added
//                             into grad_pred
//
// All four terms are means of absolute values, so d(total)/d(pred) is a sum of signs times
// lambda / count: it does not depend on any upstream value and is produced in the same pass.
// Sums are reduced wave -> block -> fixed-order final sum, without atomics (bit-repeatable).
#pragma once
#include "plc_common.h"

struct LossArgs {
    const float* pred;       // [B][T][Hh][Wh]
    const float* lr;         // [B][T][H][W]
    float* dpred;            // [B][T][Hh][Wh] or null
    float* coarse_sg;        // [B*T][H][W]: sign(coarse - lr) / bin area (unused when identity)
    float* partials;         // [n_main_blocks][4] then [n_pool_blocks]
    int B, T, Hh, Wh, H, W;
    int identity;            // Hh == H && Wh == W
    int sy, sx;              // Hh / H and Wh / W when both divide (every adaptive bin is an sy x sx block), else 0
    int n_main_blocks, n_pool_blocks;
    float kx, ky, kt, kc;    // lambda / element count of each mean (0 when the count is 0)
    // rain-gauge term
    int N, values_per_batch, weight_mode;
    float scale_factor, lambda_point;
    const long long* coords; // [N][2] (row, col) on the LR grid
    const float* values;     // [B][T][N] or [T][N]
    float lambda_conserve, lambda_smooth, lambda_temporal;
    float* out;              // total, point, conserve, smooth, temporal
};

enum { LOSS_W_NONE = 0, LOSS_W_LOG = 1, LOSS_W_SQRT = 2, LOSS_W_STRATIFIED = 3 };

__device__ __forceinline__ float plc_sgn(float v) { return (float)(v > 0.0f) - (float)(v < 0.0f); }

__device__ __forceinline__ float plc_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// Sum of `v` over the NW waves of a block, the same value in every thread.  `red` holds NW floats.
template <int NW = 4>
__device__ __forceinline__ float plc_block_sum(float v, float* red) {
    v = plc_wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.0f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[w];
    __syncthreads();
    return s;
}

// adaptive_avg_pool2d bin of output index i: [floor(i*in/out), ceil((i+1)*in/out))
__device__ __forceinline__ int plc_bin_lo(int i, int in, int out) { return (int)(((long long)i * in) / out); }
__device__ __forceinline__ int plc_bin_hi(int i, int in, int out) { return (int)(((long long)(i + 1) * in + out - 1) / out); }

__global__ __launch_bounds__(256) void plc_loss_pool_kernel(const LossArgs a) {
    __shared__ float red[4];
    const long long total = (long long)a.B * a.T * a.H * a.W;
    float acc = 0.0f;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int j = (int)(idx % a.W);
        const long long r = idx / a.W;
        const int i = (int)(r % a.H);
        const long long bt = r / a.H;
        const float* p = a.pred + bt * a.Hh * a.Wh;
        float s = 0.0f, area;
        if (a.sx > 0 && (a.sx & 3) == 0) {
            // integer scale (the bins are sy x sx blocks, 16-byte aligned rows): the same sum in the same
            // order, read as float4
            const float* q = p + (long long)i * a.sy * a.Wh + j * a.sx;
            for (int y = 0; y < a.sy; ++y, q += a.Wh) {
#pragma unroll 2
                for (int x = 0; x < a.sx; x += 4) {
                    const float4 v = *(const float4*)(q + x);
                    s += v.x; s += v.y; s += v.z; s += v.w;
                }
            }
            area = (float)(a.sy * a.sx);
        } else {
            const int y0 = plc_bin_lo(i, a.Hh, a.H), y1 = plc_bin_hi(i, a.Hh, a.H);
            const int x0 = plc_bin_lo(j, a.Wh, a.W), x1 = plc_bin_hi(j, a.Wh, a.W);
            for (int y = y0; y < y1; ++y)
                for (int x = x0; x < x1; ++x) s += p[(long long)y * a.Wh + x];
            area = (float)((y1 - y0) * (x1 - x0));
        }
        const float d = s / area - a.lr[idx];
        acc += fabsf(d);
        a.coarse_sg[idx] = plc_sgn(d) / area;
    }
    const float s = plc_block_sum(acc, red);
    if (threadIdx.x == 0) a.partials[(long long)a.n_main_blocks * 4 + blockIdx.x] = s;
}

// One thread per V consecutive pixels of a row (V = 4 needs Wh % 4 == 0 and 16-byte aligned bases).
template <int V>
__global__ __launch_bounds__(256) void plc_loss_main_kernel(const LossArgs a) {
    __shared__ float red[4];
    const int Hh = a.Hh, Wh = a.Wh, T = a.T;
    const int wv = Wh / V;
    const long long plane = (long long)Hh * Wh;
    const long long total = (long long)a.B * T * Hh * wv;
    float sx = 0.0f, sy = 0.0f, st = 0.0f, sc = 0.0f;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int xv = (int)(idx % wv);
        long long r = idx / wv;
        const int y = (int)(r % Hh);
        const long long bt = r / Hh;
        const int t = (int)(bt % T);
        const int x0 = xv * V;
        const long long off = bt * plane + (long long)y * Wh + x0;
        const float* p = a.pred + off;
        float cur[V], up[V], dn[V], pv[V], nx[V];
        const bool has_up = y > 0, has_dn = y + 1 < Hh, has_pv = t > 0, has_nx = t + 1 < T;
        if constexpr (V == 4) {
            // A neighbour that does not exist is replaced by the pixel's own address (its value is
            // never used: every use below is guarded by the same has_* flag), so all loads are
            // unconditional and in flight together instead of one round trip after another.
            const float4 c4 = *(const float4*)p;
            const float4 u4 = *(const float4*)(p - (has_up ? Wh : 0));
            const float4 d4 = *(const float4*)(p + (has_dn ? Wh : 0));
            const float4 p4 = *(const float4*)(p - (has_pv ? plane : 0));
            const float4 n4 = *(const float4*)(p + (has_nx ? plane : 0));
            cur[0] = c4.x; cur[1] = c4.y; cur[2] = c4.z; cur[3] = c4.w;
            up[0] = u4.x; up[1] = u4.y; up[2] = u4.z; up[3] = u4.w;
            dn[0] = d4.x; dn[1] = d4.y; dn[2] = d4.z; dn[3] = d4.w;
            pv[0] = p4.x; pv[1] = p4.y; pv[2] = p4.z; pv[3] = p4.w;
            nx[0] = n4.x; nx[1] = n4.y; nx[2] = n4.z; nx[3] = n4.w;
        } else {
#pragma unroll
            for (int k = 0; k < V; ++k) {
                cur[k] = p[k];
                up[k] = p[k - (has_up ? Wh : 0)];
                dn[k] = p[k + (has_dn ? Wh : 0)];
                pv[k] = p[k - (has_pv ? plane : 0)];
                nx[k] = p[k + (has_nx ? plane : 0)];
            }
        }
        const bool has_l = x0 > 0, has_r = x0 + V < Wh;
        const float left = p[has_l ? -1 : 0];
        const float right = p[has_r ? V : 0];
        float g[V];
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float c = cur[k];
            float gx = 0.0f, gy = 0.0f, gt = 0.0f;
            // forward differences owned by this pixel: (x+1) - x, (y+1) - y, (t+1) - t
            const bool r_ok = (k + 1 < V) || has_r;
            const float rv = (k + 1 < V) ? cur[(k + 1) % V] : right;
            if (r_ok) { const float d = rv - c; sx += fabsf(d); gx -= plc_sgn(d); }
            const bool l_ok = (k > 0) || has_l;
            const float lv = (k > 0) ? cur[(k + V - 1) % V] : left;
            if (l_ok) gx += plc_sgn(c - lv);
            if (has_dn) { const float d = dn[k] - c; sy += fabsf(d); gy -= plc_sgn(d); }
            if (has_up) gy += plc_sgn(c - up[k]);
            if (has_nx) { const float d = nx[k] - c; st += fabsf(d); gt -= plc_sgn(d); }
            if (has_pv) gt += plc_sgn(c - pv[k]);
            g[k] = a.kx * gx + a.ky * gy + a.kt * gt;
        }
        if (a.identity) {
            const float* l = a.lr + off;
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const float d = cur[k] - l[k];
                sc += fabsf(d);
                g[k] += a.kc * plc_sgn(d);
            }
        } else if (a.dpred && a.sx > 0) {
            // integer scale: pixel (y, x) lies in exactly one bin, (y / sy, x / sx)
            const float* sg = a.coarse_sg + bt * a.H * a.W + (long long)((unsigned)y / (unsigned)a.sy) * a.W;
#pragma unroll
            for (int k = 0; k < V; ++k) g[k] += a.kc * sg[(unsigned)(x0 + k) / (unsigned)a.sx];
        } else if (a.dpred) {
            // every adaptive bin containing (y, x): the bins of i0-1, i0, i0+1 are the only candidates
            const float* sg = a.coarse_sg + bt * a.H * a.W;
            const int i0 = (int)(((long long)y * a.H) / Hh);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const int x = x0 + k;
                const int j0 = (int)(((long long)x * a.W) / Wh);
                float s = 0.0f;
                for (int i = i0 - 1; i <= i0 + 1; ++i) {
                    if (i < 0 || i >= a.H || y < plc_bin_lo(i, Hh, a.H) || y >= plc_bin_hi(i, Hh, a.H)) continue;
                    for (int j = j0 - 1; j <= j0 + 1; ++j) {
                        if (j < 0 || j >= a.W || x < plc_bin_lo(j, Wh, a.W) || x >= plc_bin_hi(j, Wh, a.W)) continue;
                        s += sg[(long long)i * a.W + j];
                    }
                }
                g[k] += a.kc * s;
            }
        }
        if (a.dpred) {
            float* gp = a.dpred + off;
            if constexpr (V == 4) {
                *(float4*)gp = make_float4(g[0], g[1], g[2], g[3]);
            } else {
#pragma unroll
                for (int k = 0; k < V; ++k) gp[k] = g[k];
            }
        }
    }
    const float bx = plc_block_sum(sx, red);
    const float by = plc_block_sum(sy, red);
    const float bt_ = plc_block_sum(st, red);
    const float bc = plc_block_sum(sc, red);
    if (threadIdx.x == 0) {
        float* o = a.partials + (long long)blockIdx.x * 4;
        o[0] = bx; o[1] = by; o[2] = bt_; o[3] = bc;
    }
}

__device__ __forceinline__ float plc_gauge_weight(float obs, int mode) {
    if (mode == LOSS_W_LOG) return log1pf(obs) + 1.0f;
    if (mode == LOSS_W_SQRT) return sqrtf(obs) + 1.0f;
    if (mode == LOSS_W_STRATIFIED) return obs >= 50.0f ? 5.0f : (obs >= 25.0f ? 3.0f : (obs >= 10.0f ? 2.0f : 1.0f));
    return 1.0f;
}

// Station (row, col) on the prediction grid: trunc((c + 0.5) * s - 0.5) in fp32; false if outside.
__device__ __forceinline__ bool plc_gauge_pixel(const LossArgs& a, int n, int& r, int& c) {
    const float fr = ((float)a.coords[2 * n] + 0.5f) * a.scale_factor - 0.5f;
    const float fc = ((float)a.coords[2 * n + 1] + 0.5f) * a.scale_factor - 0.5f;
    const long long lr_ = (long long)fr, lc = (long long)fc;
    r = (int)lr_; c = (int)lc;
    return lr_ >= 0 && lr_ < a.Hh && lc >= 0 && lc < a.Wh;
}

// Pixel offset of station n inside a frame, or -1 when it falls outside the prediction grid.
constexpr int LOSS_PIX_CACHE = 2048;
constexpr int LOSS_FIN_THREADS = 1024;   // one block; its work is a chain of dependent loads per element
__device__ __forceinline__ int plc_gauge_offset(const LossArgs& a, const int* cache, int n) {
    if (n < LOSS_PIX_CACHE) return cache[n];
    int r, c;
    return plc_gauge_pixel(a, n, r, c) ? r * a.Wh + c : -1;
}

__global__ __launch_bounds__(LOSS_FIN_THREADS) void plc_loss_finalize_kernel(const LossArgs a) {
    __shared__ float red6[6][LOSS_FIN_THREADS / 64];
    __shared__ float sh_k;
    __shared__ int pix[LOSS_PIX_CACHE];
    __shared__ int nxt[LOSS_PIX_CACHE];      // next station (in order) on the same pixel, or -1; bit 30: not the first on it
    const int tid = threadIdx.x;
    for (int n = tid; n < a.N && n < LOSS_PIX_CACHE; n += LOSS_FIN_THREADS) {
        int r, c;
        pix[n] = plc_gauge_pixel(a, n, r, c) ? r * a.Wh + c : -1;
    }
    __syncthreads();
    // Stations sharing a pixel, found ONCE per call (N^2 / 1024 probes per thread) instead of once per
    // frame: the gradient pass below then walks each pixel's chain of stations.
    const bool chains = a.dpred != nullptr && a.N <= LOSS_PIX_CACHE;
    if (chains) {
        for (int n = tid; n < a.N; n += LOSS_FIN_THREADS) {
            const int o = pix[n];
            int next = -1, later = 0;
            for (int m = n + 1; m < a.N && next < 0; ++m) next = pix[m] == o ? m : -1;
            for (int m = 0; m < n && !later; ++m) later = pix[m] == o;
            nxt[n] = (next < 0 ? 0x3fffffff : next) | (later ? 0x40000000 : 0);
        }
    }
    // fixed-order sums of the block partials
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int b = tid; b < a.n_main_blocks; b += LOSS_FIN_THREADS) {
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] += a.partials[(long long)b * 4 + k];
    }
    for (int b = tid; b < a.n_pool_blocks; b += LOSS_FIN_THREADS) s[3] += a.partials[(long long)a.n_main_blocks * 4 + b];
    __syncthreads();   // pix[] is complete

    // rain-gauge term: one (frame, station) element per thread and iteration
    const long long plane = (long long)a.Hh * a.Wh;
    const long long items = (long long)a.B * a.T * a.N;
    float psum = 0.0f, pcnt = 0.0f;
    for (long long it = tid; it < items; it += LOSS_FIN_THREADS) {
        const int n = (int)(it % a.N);
        const int f = (int)(it / a.N);
        const int o = plc_gauge_offset(a, pix, n);
        if (o < 0) continue;
        const int b = f / a.T, t = f % a.T;
        const float obs = a.values[((long long)(a.values_per_batch ? b : 0) * a.T + t) * a.N + n];
        if (obs != obs) continue;
        psum += fabsf(a.pred[(long long)f * plane + o] - obs) * plc_gauge_weight(obs, a.weight_mode);
        pcnt += 1.0f;
    }
    // the six sums in one pass over the block: wave sums, then the waves in a fixed order (one
    // barrier instead of twelve; the loads of the partials and of the gauge pixels overlap)
    float v6[6] = {s[0], s[1], s[2], s[3], psum, pcnt};
#pragma unroll
    for (int k = 0; k < 6; ++k) v6[k] = plc_wave_sum(v6[k]);
    if ((tid & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) red6[k][tid >> 6] = v6[k];
    }
    __syncthreads();
    float tot[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        tot[k] = 0.0f;
#pragma unroll
        for (int w = 0; w < LOSS_FIN_THREADS / 64; ++w) tot[k] += red6[k][w];
    }
    const float Sx = tot[0], Sy = tot[1], St = tot[2], Sc = tot[3], Ps = tot[4], Pc = tot[5];
    if (tid == 0) {
        const float fB = (float)a.B, fT = (float)a.T, fH = (float)a.Hh, fW = (float)a.Wh;
        const float point = Pc > 0.0f ? Ps / Pc : 0.0f;
        const float conserve = Sc / (fB * fT * (float)a.H * (float)a.W);
        const float smooth = Sx / (fB * fT * fH * (fW - 1.0f)) + Sy / (fB * fT * (fH - 1.0f) * fW);
        const float temporal = St / (fB * (fT - 1.0f) * fH * fW);
        a.out[0] = a.lambda_point * point + a.lambda_conserve * conserve + a.lambda_smooth * smooth
                   + a.lambda_temporal * temporal;
        a.out[1] = point; a.out[2] = conserve; a.out[3] = smooth; a.out[4] = temporal;
        sh_k = Pc > 0.0f ? a.lambda_point / Pc : 0.0f;
    }
    __syncthreads();
    if (!a.dpred || sh_k == 0.0f) return;
    // Gradient of the gauge term.  Stations sharing a pixel are added by the first of them, in
    // station order, so the result does not depend on thread scheduling.
    const float k = sh_k;
    for (long long it = tid; it < items; it += LOSS_FIN_THREADS) {
        const int n = (int)(it % a.N);
        const int f = (int)(it / a.N);
        const int o = plc_gauge_offset(a, pix, n);
        if (o < 0) continue;
        bool first = true;
        if (chains) first = (nxt[n] & 0x40000000) == 0;
        else for (int m = 0; m < n && first; ++m) first = plc_gauge_offset(a, pix, m) != o;
        if (!first) continue;
        const int b = f / a.T, t = f % a.T;
        const float* vals = a.values + ((long long)(a.values_per_batch ? b : 0) * a.T + t) * a.N;
        const float pv = a.pred[(long long)f * plane + o];
        float g = a.dpred[(long long)f * plane + o];
        if (chains) {
            for (int m = n; m >= 0;) {           // this pixel's stations, in station order (same sums as the scan)
                const float obs = vals[m];
                if (obs == obs) g += k * plc_gauge_weight(obs, a.weight_mode) * plc_sgn(pv - obs);
                const int link = nxt[m] & 0x3fffffff;
                m = link == 0x3fffffff ? -1 : link;
            }
        } else {
            for (int m = n; m < a.N; ++m) {
                if (plc_gauge_offset(a, pix, m) != o) continue;
                const float obs = vals[m];
                if (obs != obs) continue;
                g += k * plc_gauge_weight(obs, a.weight_mode) * plc_sgn(pv - obs);
            }
        }
        a.dpred[(long long)f * plane + o] = g;
    }
}

