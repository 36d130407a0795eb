// Gradient clipping by global L2 norm + Adam update over a table of tensors, two launches:
//
//   plc_grad_sqnorm_kernel   sum of squares of every gradient element -> one partial per block
//   plc_clip_adam_kernel     every block re-sums the partials in the same fixed order (so all agree
//                            bitwise), forms clip = min(1, max_norm / (norm + 1e-6)), scales the
//                            gradients in place and applies Adam to the tensors marked `update`
//
// It replaces torch.nn.utils.clip_grad_norm_ + torch.optim.Adam.step of the reference's training
// step (src/training/trainer.py:311-313), which on the GPU are ~25 small launches over ~20 tensors.
// The flat element index space is the concatenation of the tensors in table order.
#pragma once
#include "plc_common.h"

constexpr int OPT_MAX_TENSORS = 48;
constexpr int OPT_ELEMS_PER_BLOCK = 1024;   // 256 threads x 4

struct OptTable {
    float* param[OPT_MAX_TENSORS];
    float* grad[OPT_MAX_TENSORS];
    float* exp_avg[OPT_MAX_TENSORS];
    float* exp_avg_sq[OPT_MAX_TENSORS];
    const float* restore[OPT_MAX_TENSORS];  // guarded step: value the gradient returns to on a skip (or null)
    long long start[OPT_MAX_TENSORS + 1];   // first flat index of each tensor; start[n] = total
    int update[OPT_MAX_TENSORS];
    int n;
};

struct AdamScalars {
    float max_norm;        // <= 0: no clipping
    float step_size;       // lr / (1 - beta1^step)
    float beta1, beta2;
    float omb1, omb2;      // 1 - beta1, 1 - beta2, formed in double (1 - 0.999f is off by 1e-5 relative)
    float bc2_sqrt;        // sqrt(1 - beta2^step)
    float eps;
    float weight_decay;    // L2 (added to the gradient), as torch.optim.Adam
};

// Guarded step: whether this call applies its update is decided ON THE DEVICE from a guard scalar
// (the loss: NaN means "skip this batch", src/training/trainer.py:306-308 of the reference), so the
// host need not wait for the loss before launching the backward pass.  The update count lives in
// device memory and only advances when an update is applied; the bias corrections are formed from
// it here, in double like the host path.
struct AdamDyn {
    float step_size;   // lr / (1 - beta1^step)
    float bc2_sqrt;    // sqrt(1 - beta2^step)
    int apply;         // 0: leave parameters, optimizer state and gradients untouched
    int pad;
};

__global__ void plc_adam_prepare_kernel(const float* guard, long long* step_counter, int* skipped, double lr,
                                        double beta1, double beta2, AdamDyn* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float gv = guard ? *guard : 0.0f;
    const bool skip = gv != gv;
    long long s = *step_counter;
    if (!skip) {
        s += 1;
        *step_counter = s;
    }
    if (skipped) *skipped = skip ? 1 : 0;
    const double sd = (double)(s < 1 ? 1 : s);
    out->step_size = (float)(lr / (1.0 - pow(beta1, sd)));
    out->bc2_sqrt = (float)sqrt(1.0 - pow(beta2, sd));
    out->apply = skip ? 0 : 1;
    out->pad = 0;
}

__device__ __forceinline__ int plc_opt_find(const OptTable& tb, long long e) {
    int k = 0;
    while (k + 1 < tb.n && e >= tb.start[k + 1]) ++k;
    return k;
}

__device__ __forceinline__ float plc_opt_block_sum(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float s = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return s;
}

__global__ __launch_bounds__(256) void plc_grad_sqnorm_kernel(const OptTable tb, float* partials) {
    __shared__ float red[4];
    const long long total = tb.start[tb.n];
    const long long base = (long long)blockIdx.x * OPT_ELEMS_PER_BLOCK;
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long long e = base + threadIdx.x + j * 256;
        if (e < total) {
            const int k = plc_opt_find(tb, e);
            const float g = tb.grad[k][e - tb.start[k]];
            s += g * g;
        }
    }
    const float b = plc_opt_block_sum(s, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = b;
}

__global__ __launch_bounds__(256) void plc_clip_adam_kernel(const OptTable tb, const float* partials, int n_partials,
                                                            AdamScalars sc, float* grad_norm, const AdamDyn* dyn) {
    __shared__ float red[4];
    float s = 0.0f;
    for (int i = threadIdx.x; i < n_partials; i += 256) s += partials[i];
    const float norm = sqrtf(plc_opt_block_sum(s, red));
    // torch.clamp(max_norm / (norm + 1e-6), max=1.0): a NaN norm propagates (fminf would answer 1)
    const float ratio = sc.max_norm / (norm + 1e-6f);
    const float clip = sc.max_norm > 0.0f ? (ratio > 1.0f ? 1.0f : ratio) : 1.0f;
    if (grad_norm && blockIdx.x == 0 && threadIdx.x == 0) *grad_norm = norm;
    const long long total = tb.start[tb.n];
    const long long base = (long long)blockIdx.x * OPT_ELEMS_PER_BLOCK;
    if (dyn) {                          // guarded step: scalars and the apply/skip decision come from the device
        if (!dyn->apply) {              // (uniform over the launch) skipped: only accumulated gradients go back
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long long e = base + threadIdx.x + j * 256;
                if (e >= total) continue;
                const int k = plc_opt_find(tb, e);
                if (tb.restore[k]) tb.grad[k][e - tb.start[k]] = tb.restore[k][e - tb.start[k]];
            }
            return;
        }
        sc.step_size = dyn->step_size;
        sc.bc2_sqrt = dyn->bc2_sqrt;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long long e = base + threadIdx.x + j * 256;
        if (e >= total) continue;
        const int k = plc_opt_find(tb, e);
        const long long i = e - tb.start[k];
        float g = tb.grad[k][i] * clip;
        tb.grad[k][i] = g;
        if (!tb.update[k]) continue;
        float p = tb.param[k][i];
        if (sc.weight_decay != 0.0f) g = __builtin_fmaf(sc.weight_decay, p, g);
        float m = tb.exp_avg[k][i], v = tb.exp_avg_sq[k][i];
        m = m + sc.omb1 * (g - m);                                         // exp_avg.lerp_(grad, 1 - beta1)
        v = v * sc.beta2 + sc.omb2 * g * g;                                // mul_(beta2).addcmul_(g, g, 1 - beta2)
        const float denom = sqrtf(v) / sc.bc2_sqrt + sc.eps;  
        p -= sc.step_size * (m / denom);                                   // addcdiv_(exp_avg, denom, -step_size)
        tb.exp_avg[k][i] = m;
        tb.exp_avg_sq[k][i] = v;
        tb.param[k][i] = p;
    }
}

// ---------------------------------------------------------------------------------------------
// Data-parallel gradient exchange: gather the table's gradients into one flat buffer (scaled by this
// rank's sample count, with the NaN flag and the count as two trailing elements) and, after the
// all-reduce, scatter the per-sample mean back.  Two launches replace the dozen small ones of
// cat / mul / fill / div / where.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void plc_dp_pack_kernel(const OptTable tb, float* flat, float n_local, const float* guard,
                                                          float* tail) {
    const long long total = tb.start[tb.n];
    const long long base = (long long)blockIdx.x * OPT_ELEMS_PER_BLOCK;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long long e = base + threadIdx.x + j * 256;
        if (e < total) {
            const int k = plc_opt_find(tb, e);
            flat[e] = tb.grad[k][e - tb.start[k]] * n_local;
        }
    }
    if (tail && blockIdx.x == 0 && threadIdx.x == 0) {
        const float gv = guard ? *guard : 0.0f;
        tail[0] = gv != gv ? 1.0f : 0.0f;
        tail[1] = n_local;
    }
}

__global__ __launch_bounds__(256) void plc_dp_unpack_kernel(const OptTable tb, const float* flat, const float* tail,
                                                            const float* guard_in, float* guard_out) {
    const float count = tail[1];
    const long long total = tb.start[tb.n];
    const long long base = (long long)blockIdx.x * OPT_ELEMS_PER_BLOCK;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long long e = base + threadIdx.x + j * 256;
        if (e < total) {
            const int k = plc_opt_find(tb, e);
            tb.grad[k][e - tb.start[k]] = flat[e] / count;
        }
    }
    if (guard_out && blockIdx.x == 0 && threadIdx.x == 0) {
        const float f = tail[0];
        *guard_out = (f > 0.0f || f != f) ? __builtin_nanf("") : (guard_in ? *guard_in : 0.0f);
    }
}
