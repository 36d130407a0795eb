// The pointwise backward of one ConvLSTM cell step on one 8-channel unit: types, loads and arithmetic
// shared by the standalone pass (plc_bwd_kernels.h) and the data-gradient conv that forms dz while staging
// its input rows (plc_convr_kernel.h).
#pragma once
#include "plc_common.h"

struct F8 {
    float v[8];
};
__device__ __forceinline__ F8 ld_bf8(const char* p) {
    const uint4 u = *(const uint4*)p;
    const float4 a = unpack_bf4(make_uint2(u.x, u.y)), b = unpack_bf4(make_uint2(u.z, u.w));
    F8 r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
}
__device__ __forceinline__ void st_bf8(char* p, const F8& f) {
    *(uint4*)p = make_uint4(pack_bf2(f.v[0], f.v[1]), pack_bf2(f.v[2], f.v[3]), pack_bf2(f.v[4], f.v[5]),
                            pack_bf2(f.v[6], f.v[7]));
}
__device__ __forceinline__ F8 ld_f8(const char* p) {
    const float4 a = *(const float4*)p, b = *(const float4*)(p + 16);
    F8 r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
}
__device__ __forceinline__ void st_f8(char* p, const F8& f) {
    *(float4*)p = make_float4(f.v[0], f.v[1], f.v[2], f.v[3]);
    *(float4*)(p + 16) = make_float4(f.v[4], f.v[5], f.v[6], f.v[7]);
}

// A CB8 bf16 view: `p` already points at the first channel block of interest of image 0;
// img = bytes between images.  p == nullptr means "absent" (treated as zeros).
struct Cb8View {
    const char* p;
    long long img;
};

// ConvLSTM cell, pointwise backward of one timestep (all images of the step):
//   dh = sum of up to three incoming hidden-state gradients, dc_in = gradient of c_t from step t+1
//   tc = tanh(c_t);  do = dh*tc;  dct = dc_in + dh*o*(1-tc^2)
//   dz = [dct*g*i(1-i), dct*c_prev*f(1-f), do*o(1-o), dct*i*(1-g^2)]   (gate-major, like `gates`)
//   dc_prev = dct*f
struct LstmBwdArgs {
    int B, ncb, H, W;              // ncb = Ch / 8
    Cb8View dh0, dh1, dh2;         // bf16 CB8 views of ncb blocks
    Cb8View dh0_mul;               // optional (p null = none): dh0 is taken as bf16(dh0 * dh0_mul), e.g. the
                                   // gradient of the gated feature map times the gate, formed here instead of
                                   // being written out and read back
    int dc_bf16;                   // dc_in / dc_out are bf16 CB8 (16-byte units) instead of fp32 CB8 (32-byte units)
    const char* dc_in;             // fp32 (or bf16) CB8 [B][ncb] or null
    const char* gates;             // bf16 CB8 [B][4*ncb]
    const char* c_prev;            // fp32 CB8 [B][ncb] or null (= zeros: the first timestep)
    const char* c_cur;             // fp32 CB8 [B][ncb] (not read: c_t is rebuilt from gates and c_prev)
    char* dz;                      // bf16 CB8 [B][4*ncb]
    char* dc_out;                  // fp32 (or bf16) CB8 [B][ncb]
};

// One 8-channel unit of the pass: image b, unit u = channel block * H * W + pixel inside an ncb-block image.
// Everything between the addresses and the stores, so that every kernel that runs the pass (the standalone
// launches below, the data-gradient conv that forms dz while staging it) rounds identically.
__device__ __forceinline__ void plc_lstm_bwd_unit(const LstmBwdArgs& a, long long b, long long u, F8& zi, F8& zf, F8& zo,
                                                  F8& zg, F8& dcp) {
    const long long hw = (long long)a.H * a.W;
    F8 dh;
#pragma unroll
    for (int j = 0; j < 8; ++j) dh.v[j] = 0.0f;
    const Cb8View views[3] = {a.dh0, a.dh1, a.dh2};
#pragma unroll
    for (int k = 0; k < 3; ++k)
        if (views[k].p) {
            F8 t = ld_bf8(views[k].p + b * views[k].img + u * 16);
            if (k == 0 && a.dh0_mul.p) {
                const F8 m = ld_bf8(a.dh0_mul.p + b * a.dh0_mul.img + u * 16);
#pragma unroll
                for (int j = 0; j < 8; j += 2) {      // rounded to bf16 like the stored product it replaces
                    const unsigned pk = pack_bf2(t.v[j] * m.v[j], t.v[j + 1] * m.v[j + 1]);
                    t.v[j] = __uint_as_float(pk << 16);
                    t.v[j + 1] = __uint_as_float(pk & 0xffff0000u);
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) dh.v[j] += t.v[j];
        }
    const long long su = (b * a.ncb) * hw + u;              // unit index in [B][ncb] tensors
    const long long gu = (b * 4 * a.ncb) * hw + u;          // unit index of gate 0 in [B][4 ncb]
    F8 dcin;
    if (a.dc_in) dcin = a.dc_bf16 ? ld_bf8(a.dc_in + su * 16) : ld_f8(a.dc_in + su * 32);
    else {
#pragma unroll
        for (int j = 0; j < 8; ++j) dcin.v[j] = 0.0f;
    }
    F8 cp;                                   // absent c_prev = the zero initial state
    if (a.c_prev) cp = ld_f8(a.c_prev + su * 32);
    else {
#pragma unroll
        for (int j = 0; j < 8; ++j) cp.v[j] = 0.0f;
    }
    const F8 gi = ld_bf8(a.gates + (gu + 0 * a.ncb * hw) * 16), gf = ld_bf8(a.gates + (gu + 1 * a.ncb * hw) * 16);
    const F8 go = ld_bf8(a.gates + (gu + 2 * a.ncb * hw) * 16), gg = ld_bf8(a.gates + (gu + 3 * a.ncb * hw) * 16);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        // c_t is rebuilt from the saved gates and c_{t-1} (one fma) instead of being read back:
        // 32 of the 288 bytes a unit moves; the saved gates are bf16, so c_t differs from the
        // forward's by their rounding, well inside the gradient tolerance
        const float tc = fast_tanh(__builtin_fmaf(gf.v[j], cp.v[j], gi.v[j] * gg.v[j]));
        const float d_o = dh.v[j] * tc;
        const float dct = dcin.v[j] + dh.v[j] * go.v[j] * (1.0f - tc * tc);
        zi.v[j] = dct * gg.v[j] * gi.v[j] * (1.0f - gi.v[j]);
        zf.v[j] = dct * cp.v[j] * gf.v[j] * (1.0f - gf.v[j]);
        zo.v[j] = d_o * go.v[j] * (1.0f - go.v[j]);
        zg.v[j] = dct * gi.v[j] * (1.0f - gg.v[j] * gg.v[j]);
        dcp.v[j] = dct * gf.v[j];
    }
}
