// Shared device/host helpers for the gfx950 ConvLSTM kernels.
//
// Internal activation layout ("CB8"): [image][channel block][H][W][8 channels], i.e. one
// 16-byte unit = 8 consecutive channels of one pixel.  bf16 units feed the MFMA operands
// directly (one ds_read_b128 = one 8-deep K fragment); the fp32 cell state uses the same
// indexing with 32-byte units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16_t;

// A tensor addressed per image n of a launch: byte offset = (n % n_in) * si + (n / n_in) * so.
// n_in is shared by every tensor of one launch (it is the batch size B when images are
// enumerated t-major, n = t * B + b).
struct PlcTensor {
    char* p;
    long long si;
    long long so;
};

__device__ __forceinline__ char* plc_img(const PlcTensor& t, int n, int n_in) {
    return t.p + (long long)(n % n_in) * t.si + (long long)(n / n_in) * t.so;
}

__device__ __forceinline__ unsigned short f2bf(float f) {
    __bf16 h = (__bf16)f;  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN
    return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ float bf2f(unsigned short u) {
    return __builtin_bit_cast(float, (unsigned)u << 16);
}
typedef float plc_f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 plc_bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
    // one v_cvt_pk_bf16_f32 (two scalar conversions plus shift/or otherwise)
    const plc_f32x2_t f = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, plc_bf16x2_t));
}
__device__ __forceinline__ uint2 pack_bf4(float a, float b, float c, float d) {
    return make_uint2(pack_bf2(a, b), pack_bf2(c, d));
}
__device__ __forceinline__ float4 unpack_bf4(uint2 v) {
    return make_float4(bf2f((unsigned short)(v.x & 0xffff)), bf2f((unsigned short)(v.x >> 16)),
                       bf2f((unsigned short)(v.y & 0xffff)), bf2f((unsigned short)(v.y >> 16)));
}

// sigmoid / tanh on the hardware exp2 + rcp units (abs error ~1e-7, far below bf16 storage).
__device__ __forceinline__ float fast_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
__device__ __forceinline__ float fast_tanh(float x) {
    // 1 - 2/(e^{2x}+1); saturates cleanly for |x| large (exp2 -> inf gives 1, -> 0 gives -1)
    return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177792681f * x)), 1.0f);
}
