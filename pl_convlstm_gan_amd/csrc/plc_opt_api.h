// C ABI of the fused gradient clip + Adam step (included at the end of plc_api.hip).
#pragma once
#include <math.h>

#include "plc_opt_kernels.h"

static long long opt_chunk_elems(const plc_opt_tensor* t, int first, int n) {
    long long s = 0;
    for (int k = first; k < first + n; ++k) s += t[k].numel;
    return s;
}

static int opt_blocks(long long elems) { return (int)((elems + OPT_ELEMS_PER_BLOCK - 1) / OPT_ELEMS_PER_BLOCK); }

// workspace = [block partials of the squared norm | 256 bytes: the guarded step's device scalars]
static size_t opt_partials_bytes(const plc_opt_tensor* t, int n) {
    long long blocks = 0;
    for (int first = 0; first < n; first += OPT_MAX_TENSORS) {
        const int cnt = n - first < OPT_MAX_TENSORS ? n - first : OPT_MAX_TENSORS;
        blocks += opt_blocks(opt_chunk_elems(t, first, cnt));
    }
    return align256((size_t)blocks * sizeof(float));
}

extern "C" size_t plc_clip_adam_workspace_bytes(const plc_opt_tensor* t, int n) {
    if (!t || n <= 0) return 0;
    return opt_partials_bytes(t, n) + 256;
}

// guarded == false: `step` is the number of this update (host counter).
// guarded == true:  the update is applied unless *guard is NaN; *step_counter (device) counts the
//                   applied updates and yields the bias corrections; *skipped (device, optional)
//                   receives 1 when the call was skipped, else 0.
static int clip_adam_impl(const char* who, const plc_opt_tensor* t, int n, double max_norm, double lr, double beta1,
                          double beta2, double eps, double weight_decay, bool guarded, long long step,
                          const float* guard, long long* step_counter, int* skipped, float* grad_norm,
                          void* workspace, size_t workspace_bytes, void* stream) {
    PLC_REQUIRE(t && n > 0, "%s: empty tensor table", who);
    PLC_REQUIRE(lr >= 0.0 && beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0,
                "%s: bad hyper-parameter", who);
    for (int k = 0; k < n; ++k) {
        PLC_REQUIRE(t[k].numel > 0 && t[k].grad, "%s: tensor %d has no gradient or no elements", who, k);
        PLC_REQUIRE(!t[k].update || (t[k].param && t[k].exp_avg && t[k].exp_avg_sq),
                    "%s: tensor %d is to be updated but lacks param / exp_avg / exp_avg_sq", who, k);
    }
    const size_t need = plc_clip_adam_workspace_bytes(t, n);
    PLC_REQUIRE(workspace, "%s: null workspace", who);
    PLC_REQUIRE(((uintptr_t)workspace & 15) == 0, "%s: workspace must be 16-byte aligned", who);
    if (workspace_bytes < need) return plc_fail(PLC_EWORKSPACE, "%s: workspace %zu < %zu bytes", who, workspace_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    float* partials = (float*)workspace;
    AdamDyn* dyn = guarded ? (AdamDyn*)((char*)workspace + opt_partials_bytes(t, n)) : nullptr;

    AdamScalars sc;
    // every derived scalar is formed in double, as the Python optimizer does, and rounded once
    sc.max_norm = (float)max_norm;
    sc.beta1 = (float)beta1;
    sc.beta2 = (float)beta2;
    sc.omb1 = (float)(1.0 - beta1);
    sc.omb2 = (float)(1.0 - beta2);
    sc.eps = (float)eps;
    sc.weight_decay = (float)weight_decay;
    if (guarded) {
        sc.step_size = 0.0f;      // both come from the device (plc_adam_prepare_kernel)
        sc.bc2_sqrt = 1.0f;
        hipLaunchKernelGGL(plc_adam_prepare_kernel, dim3(1), dim3(64), 0, st, guard, step_counter, skipped, lr, beta1,
                           beta2, dyn);
        PLC_HIP(hipGetLastError());
    } else {
        sc.step_size = (float)(lr / (1.0 - pow(beta1, (double)step)));
        sc.bc2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)step));
    }

    int n_partials = 0;
    for (int pass = 0; pass < 2; ++pass) {
        int part_off = 0;
        for (int first = 0; first < n; first += OPT_MAX_TENSORS) {
            const int cnt = n - first < OPT_MAX_TENSORS ? n - first : OPT_MAX_TENSORS;
            OptTable tb;
            memset(&tb, 0, sizeof(tb));
            tb.n = cnt;
            long long off = 0;
            for (int k = 0; k < cnt; ++k) {
                const plc_opt_tensor& s = t[first + k];
                tb.param[k] = s.param; tb.grad[k] = s.grad; tb.exp_avg[k] = s.exp_avg; tb.exp_avg_sq[k] = s.exp_avg_sq;
                tb.update[k] = s.update;
                tb.restore[k] = guarded ? s.restore : nullptr;
                tb.start[k] = off;
                off += s.numel;
            }
            tb.start[cnt] = off;
            const int blocks = opt_blocks(off);
            if (pass == 0)
                hipLaunchKernelGGL(plc_grad_sqnorm_kernel, dim3(blocks), dim3(256), 0, st, tb, partials + part_off);
            else
                hipLaunchKernelGGL(plc_clip_adam_kernel, dim3(blocks), dim3(256), 0, st, tb, (const float*)partials,
                                   n_partials, sc, first == 0 ? grad_norm : (float*)nullptr, (const AdamDyn*)dyn);
            PLC_HIP(hipGetLastError());
            part_off += blocks;
        }
        n_partials = part_off;
    }
    return PLC_OK;
}

extern "C" int plc_clip_adam_step(const plc_opt_tensor* t, int n, double max_norm, double lr, double beta1,
                                  double beta2, double eps, double weight_decay, long long step, float* grad_norm,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    PLC_REQUIRE(step >= 1, "plc_clip_adam_step: step must be >= 1 (the number of this update)");
    return clip_adam_impl("plc_clip_adam_step", t, n, max_norm, lr, beta1, beta2, eps, weight_decay, false, step,
                          nullptr, nullptr, nullptr, grad_norm, workspace, workspace_bytes, stream);
}

extern "C" int plc_clip_adam_step_guarded(const plc_opt_tensor* t, int n, double max_norm, double lr, double beta1,
                                          double beta2, double eps, double weight_decay, const float* guard,
                                          long long* step_counter, int* skipped, float* grad_norm, void* workspace,
                                          size_t workspace_bytes, void* stream) {
    PLC_REQUIRE(step_counter, "plc_clip_adam_step_guarded: null step counter");
    return clip_adam_impl("plc_clip_adam_step_guarded", t, n, max_norm, lr, beta1, beta2, eps, weight_decay, true, 0,
                          guard, step_counter, skipped, grad_norm, workspace, workspace_bytes, stream);
}

// ---------------------------------------------------------------------------------------------
// data-parallel gradient exchange (see include/plconvlstm.h)
// ---------------------------------------------------------------------------------------------
static int dp_grads_impl(const char* who, bool pack, const plc_opt_tensor* t, int n, float* flat, long long flat_elems,
                         float n_local, const float* guard_in, float* guard_out, void* stream) {
    PLC_REQUIRE(t && n > 0 && flat, "%s: empty tensor table or null buffer", who);
    long long total = 0;
    for (int k = 0; k < n; ++k) {
        PLC_REQUIRE(t[k].numel > 0 && t[k].grad, "%s: tensor %d has no gradient or no elements", who, k);
        total += t[k].numel;
    }
    PLC_REQUIRE(flat_elems == total + 2, "%s: the flat buffer must hold the %lld gradient elements + 2, not %lld", who, total,
                flat_elems);
    PLC_REQUIRE(!pack || n_local > 0.0f, "%s: n_local must be positive", who);
    hipStream_t st = (hipStream_t)stream;
    float* tail = flat + total;
    long long done = 0;
    for (int first = 0; first < n; first += OPT_MAX_TENSORS) {
        const int cnt = n - first < OPT_MAX_TENSORS ? n - first : OPT_MAX_TENSORS;
        OptTable tb;
        memset(&tb, 0, sizeof(tb));
        tb.n = cnt;
        long long off = 0;
        for (int k = 0; k < cnt; ++k) {
            tb.grad[k] = t[first + k].grad;
            tb.start[k] = off;
            off += t[first + k].numel;
        }
        tb.start[cnt] = off;
        const bool head = first == 0;       // the first launch also handles the two trailing elements
        if (pack)
            hipLaunchKernelGGL(plc_dp_pack_kernel, dim3(opt_blocks(off)), dim3(256), 0, st, tb, flat + done, n_local, guard_in,
                               head ? tail : (float*)nullptr);
        else
            hipLaunchKernelGGL(plc_dp_unpack_kernel, dim3(opt_blocks(off)), dim3(256), 0, st, tb, (const float*)(flat + done),
                               (const float*)tail, guard_in, head ? guard_out : (float*)nullptr);
        PLC_HIP(hipGetLastError());
        done += off;
    }
    return PLC_OK;
}

extern "C" int plc_dp_pack_grads(const plc_opt_tensor* t, int n, float* flat, long long flat_elems, float n_local,
                                 const float* guard, void* stream) {
    return dp_grads_impl("plc_dp_pack_grads", true, t, n, flat, flat_elems, n_local, guard, nullptr, stream);
}

extern "C" int plc_dp_unpack_grads(const plc_opt_tensor* t, int n, const float* flat, long long flat_elems,
                                   const float* guard_in, float* guard_out, void* stream) {
    return dp_grads_impl("plc_dp_unpack_grads", false, t, n, (float*)flat, flat_elems, 1.0f, guard_in, guard_out, stream);
}
