// C ABI of the fused CombinedLoss (included at the end of plc_api.hip, whose error helpers it uses).
#pragma once
#include "plc_loss_kernels.h"

struct LossPlan {
    int n_main, n_pool, vec;
    size_t off_partials, off_sg, total;
};

static int loss_plan(const plc_loss_desc* d, LossPlan* p) {
    PLC_REQUIRE(d, "plc_combined_loss: null descriptor");
    PLC_REQUIRE(d->B > 0 && d->T > 0 && d->out_H > 0 && d->out_W > 0 && d->H > 0 && d->W > 0 && d->N >= 0,
                "plc_combined_loss: bad sizes B=%d T=%d out=%dx%d lr=%dx%d N=%d", d->B, d->T, d->out_H, d->out_W,
                d->H, d->W, d->N);
    PLC_REQUIRE(d->weight_mode >= PLC_LOSS_W_NONE && d->weight_mode <= PLC_LOSS_W_STRATIFIED,
                "plc_combined_loss: unknown weight_mode %d", d->weight_mode);
    PLC_REQUIRE(d->H <= d->out_H && d->W <= d->out_W, "plc_combined_loss: the LR grid %dx%d is finer than the "
                "prediction %dx%d", d->H, d->W, d->out_H, d->out_W);
    const long long frame_px = (long long)d->out_H * d->out_W;
    PLC_REQUIRE(frame_px < (1ll << 31), "plc_combined_loss: frame too large");
    p->vec = (d->out_W % 4 == 0) ? 4 : 1;
    const long long work = (long long)d->B * d->T * d->out_H * (d->out_W / p->vec);
    long long nb = (work + 255) / 256;
    // one item per thread up to 8192 blocks (a 2048 cap made a quarter of the threads of the c3 shape
    // run a second iteration, which then set the kernel's length)
    p->n_main = (int)(nb < 1 ? 1 : (nb > 8192 ? 8192 : nb));
    const bool identity = d->H == d->out_H && d->W == d->out_W;
    const long long cells = (long long)d->B * d->T * d->H * d->W;
    nb = (cells + 255) / 256;
    p->n_pool = identity ? 0 : (int)(nb < 1 ? 1 : (nb > 2048 ? 2048 : nb));
    p->off_partials = 0;
    p->off_sg = align256(((size_t)p->n_main * 4 + p->n_pool) * sizeof(float));
    p->total = p->off_sg + (identity ? 0 : align256((size_t)cells * sizeof(float)));
    return PLC_OK;
}

extern "C" size_t plc_combined_loss_workspace_bytes(const plc_loss_desc* d) {
    LossPlan p;
    if (loss_plan(d, &p) != PLC_OK) return 0;
    return p.total;
}

extern "C" int plc_combined_loss(const plc_loss_desc* d, void* stream) {
    LossPlan p;
    int rc = loss_plan(d, &p);
    if (rc != PLC_OK) return rc;
    PLC_REQUIRE(d->pred && d->lr_input && d->losses && d->workspace, "plc_combined_loss: null pointer");
    PLC_REQUIRE(d->N == 0 || (d->s_coords && d->s_values), "plc_combined_loss: N > 0 needs s_coords and s_values");
    PLC_REQUIRE(aligned16(d->pred) && aligned16(d->workspace) && (!d->grad_pred || aligned16(d->grad_pred)),
                "plc_combined_loss: pred, grad_pred and workspace must be 16-byte aligned");
    if (d->workspace_bytes < p.total)
        return plc_fail(PLC_EWORKSPACE, "plc_combined_loss: workspace %zu < %zu bytes", d->workspace_bytes, p.total);
    hipStream_t st = (hipStream_t)stream;

    LossArgs a;
    memset(&a, 0, sizeof(a));
    a.pred = d->pred; a.lr = d->lr_input; a.dpred = d->grad_pred;
    a.partials = (float*)((char*)d->workspace + p.off_partials);
    a.coarse_sg = (float*)((char*)d->workspace + p.off_sg);
    a.B = d->B; a.T = d->T; a.Hh = d->out_H; a.Wh = d->out_W; a.H = d->H; a.W = d->W;
    a.identity = p.n_pool == 0;
    if (!a.identity && d->out_H % d->H == 0 && d->out_W % d->W == 0) {
        a.sy = d->out_H / d->H;
        a.sx = d->out_W / d->W;
    }
    a.n_main_blocks = p.n_main; a.n_pool_blocks = p.n_pool;
    // lambda / element count of each mean, in double then rounded once
    const double bt = (double)d->B * d->T;
    const double cx = bt * d->out_H * (d->out_W - 1), cy = bt * (d->out_H - 1) * d->out_W;
    const double ct = (double)d->B * (d->T - 1) * d->out_H * d->out_W, cc = bt * d->H * d->W;
    a.kx = cx > 0 ? (float)(d->lambda_smooth / cx) : 0.0f;
    a.ky = cy > 0 ? (float)(d->lambda_smooth / cy) : 0.0f;
    a.kt = ct > 0 ? (float)(d->lambda_temporal / ct) : 0.0f;
    a.kc = (float)(d->lambda_conserve / cc);
    a.N = d->N; a.values_per_batch = d->values_per_batch; a.weight_mode = d->weight_mode;
    a.scale_factor = d->scale_factor; a.lambda_point = d->lambda_point;
    a.coords = d->s_coords; a.values = d->s_values;
    a.lambda_conserve = d->lambda_conserve; a.lambda_smooth = d->lambda_smooth; a.lambda_temporal = d->lambda_temporal;
    a.out = d->losses;

    if (!a.identity) {
        hipLaunchKernelGGL(plc_loss_pool_kernel, dim3(p.n_pool), dim3(256), 0, st, a);
        PLC_HIP(hipGetLastError());
    }
    if (p.vec == 4) hipLaunchKernelGGL(plc_loss_main_kernel<4>, dim3(p.n_main), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(plc_loss_main_kernel<1>, dim3(p.n_main), dim3(256), 0, st, a);
    PLC_HIP(hipGetLastError());
    hipLaunchKernelGGL(plc_loss_finalize_kernel, dim3(1), dim3(LOSS_FIN_THREADS), 0, st, a);
    PLC_HIP(hipGetLastError());
    return PLC_OK;
}
