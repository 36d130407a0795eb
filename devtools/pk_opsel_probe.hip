// Does v_pk_fma_f32 with an operand-select broadcast give the same result as the plain form on this GPU while
// another kernel (MFMA + LDS + global traffic) runs beside it?  The victim reads a pair of floats from LDS with
// ds_read2_b32 (as the one-channel weight gradient does), accumulates x * d with four encodings of the packed
// FMA -- plain with a materialised pair, op_sel_hi:[1,0,1] (even register broadcast), op_sel:[0,1,0] (odd
// register broadcast), and the swap op_sel:[0,1,0] op_sel_hi:[1,0,1] -- and with scalar FMAs, and counts the
// lanes whose packed sums differ from the scalar ones.  Run alone and beside the aggressor.
//   hipcc -O3 --offload-arch=gfx950 -o devtools/pk_opsel_probe devtools/pk_opsel_probe.hip && ./devtools/pk_opsel_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16_t;

__global__ __launch_bounds__(256) void victim(const float* src, unsigned* mismatches, int iters) {
    __shared__ float tab[10][66];
    for (int e = threadIdx.x; e < 660; e += 256) tab[e / 66][e % 66] = src[(blockIdx.x * 660 + e) % 65536];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f2 acc_plain = {0.f, 0.f}, acc_lo = {0.f, 0.f}, acc_hi = {0.f, 0.f}, acc_sw = {0.f, 0.f};
    unsigned stale_y = 0, stale_x = 0, truth_y = 0, bad_mul = 0, bad_add = 0, bad_fma0 = 0, bad_fma2 = 0, bad_add0 = 0;
    float s_lo0 = 0.f, s_lo1 = 0.f, s_hi0 = 0.f, s_hi1 = 0.f, s_sw0 = 0.f, s_sw1 = 0.f;
    f2 x = {src[threadIdx.x] + 0.5f, src[threadIdx.x + 256] - 0.25f};
    for (int it = 0; it < iters; ++it) {
        const int row = it % 10;
        f2 d;   // d.x = tab[row][lane], d.y = tab[row][lane + 1]  (one ds_read2_b32)
        const unsigned addr = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)&tab[row][lane];
        asm volatile("ds_read2_b32 %0, %1 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=v"(d) : "v"(addr) : "memory");
        float early_y, early_x;
        asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(early_y), "=&v"(early_x) : "v"(d.y), "v"(d.x));   // first use after the wait
        f2 dd = {d.y, d.y};
        asm volatile("" : "+v"(dd));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc_plain) : "v"(x), "v"(dd));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc_lo) : "v"(x), "v"(d));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc_hi) : "v"(x), "v"(d));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "+v"(acc_sw) : "v"(x), "v"(d));
        // forms the compiler emits by itself elsewhere in the library: a swapped source of a packed mul / add / fma
        {
            f2 r;
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(d), "v"(x));      // {d.y * x.x, d.x * x.y}
            float e0, e1;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e0) : "v"(d.y), "v"(x.x));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e1) : "v"(d.x), "v"(x.y));
            bad_mul += (__float_as_uint(r.x) != __float_as_uint(e0)) | (__float_as_uint(r.y) != __float_as_uint(e1));
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(x), "v"(d));      // {x.x + d.y, x.y + d.x}
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(e0) : "v"(x.x), "v"(d.y));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(e1) : "v"(x.y), "v"(d.x));
            bad_add += (__float_as_uint(r.x) != __float_as_uint(e0)) | (__float_as_uint(r.y) != __float_as_uint(e1));
            f2 c = {1.0f, 2.0f};
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(d), "v"(x), "v"(c));   // {d.y*x.x+1, d.x*x.y+2}
            asm volatile("v_fma_f32 %0, %1, %2, 1.0" : "=v"(e0) : "v"(d.y), "v"(x.x));
            asm volatile("v_fma_f32 %0, %1, %2, 2.0" : "=v"(e1) : "v"(d.x), "v"(x.y));
            bad_fma0 += (__float_as_uint(r.x) != __float_as_uint(e0)) | (__float_as_uint(r.y) != __float_as_uint(e1));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=v"(r) : "v"(x), "v"(c), "v"(d));   // {x.x*1+d.y, x.y*2+d.x}
            asm volatile("v_fma_f32 %0, %1, 1.0, %2" : "=v"(e0) : "v"(x.x), "v"(d.y));
            asm volatile("v_fma_f32 %0, %1, 2.0, %2" : "=v"(e1) : "v"(x.y), "v"(d.x));
            bad_fma2 += (__float_as_uint(r.x) != __float_as_uint(e0)) | (__float_as_uint(r.y) != __float_as_uint(e1));
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(d), "v"(x));      // {d.y + x.x, d.x + x.y}
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(e0) : "v"(d.y), "v"(x.x));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(e1) : "v"(d.x), "v"(x.y));
            bad_add0 += (__float_as_uint(r.x) != __float_as_uint(e0)) | (__float_as_uint(r.y) != __float_as_uint(e1));
        }
        // references with the scalar instruction (the compiler would pack these pairs itself)
#define SFMA(acc_, a_, b_) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc_) : "v"(a_), "v"(b_))
        SFMA(s_lo0, x.x, d.x); SFMA(s_lo1, x.y, d.x);
        SFMA(s_hi0, x.x, d.y); SFMA(s_hi1, x.y, d.y);
        SFMA(s_sw0, x.x, d.y); SFMA(s_sw1, x.y, d.x);
        x.x = x.x * 0.999f + 0.001f; x.y = x.y * 1.001f - 0.001f;
        float late_y, late_x;
        asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(late_y), "=&v"(late_x) : "v"(d.y), "v"(d.x));       // last use of the iteration
        stale_y += __float_as_uint(early_y) != __float_as_uint(late_y);
        stale_x += __float_as_uint(early_x) != __float_as_uint(late_x);
        truth_y += __float_as_uint(late_y) != __float_as_uint(tab[row][lane + 1]);
    }
    if (stale_y) atomicAdd(&mismatches[4], stale_y);
    if (stale_x) atomicAdd(&mismatches[5], stale_x);
    if (truth_y) atomicAdd(&mismatches[6], truth_y);
    if (bad_mul) atomicAdd(&mismatches[7], bad_mul);
    if (bad_add) atomicAdd(&mismatches[8], bad_add);
    if (bad_fma0) atomicAdd(&mismatches[9], bad_fma0);
    if (bad_fma2) atomicAdd(&mismatches[10], bad_fma2);
    if (bad_add0) atomicAdd(&mismatches[11], bad_add0);
    if (acc_plain.x != s_hi0 || acc_plain.y != s_hi1) atomicAdd(&mismatches[0], 1u);
    if (acc_lo.x != s_lo0 || acc_lo.y != s_lo1) atomicAdd(&mismatches[1], 1u);
    if (acc_hi.x != s_hi0 || acc_hi.y != s_hi1) atomicAdd(&mismatches[2], 1u);
    if (acc_sw.x != s_sw0 || acc_sw.y != s_sw1) atomicAdd(&mismatches[3], 1u);
}

__global__ __launch_bounds__(256) void aggressor(const float* src, float* dst, int iters) {
    __shared__ float buf[4096];
    for (int e = threadIdx.x; e < 4096; e += 256) buf[e] = src[(blockIdx.x * 4096 + e) % 65536];
    __syncthreads();
    f32x16_t acc = {0};
    bf16x8_t a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)buf[(threadIdx.x * 8 + j) % 4096]; b[j] = (__bf16)buf[(threadIdx.x * 8 + j + 17) % 4096]; }
    for (int it = 0; it < iters; ++it) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        if ((it & 15) == 0) {
            buf[(threadIdx.x + it) % 4096] = acc[0];
            dst[((size_t)blockIdx.x * 256 + threadIdx.x + (size_t)it * 1024) % (1u << 24)] = acc[1];
            a[0] = (__bf16)buf[(threadIdx.x * 3 + it) % 4096];
        }
    }
    dst[(size_t)blockIdx.x * 256 + threadIdx.x] = acc[2];
}

int main() {
    float *src, *dst; unsigned* mm;
    CK(hipMalloc(&src, 65536 * 4 + 2048)); CK(hipMalloc(&dst, (size_t)(1u << 24) * 4)); CK(hipMalloc(&mm, 64));
    float* h = (float*)malloc(65536 * 4 + 2048);
    for (int i = 0; i < 65536 + 512; ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.0f - 1.0f;
    CK(hipMemcpy(src, h, 65536 * 4 + 2048, hipMemcpyHostToDevice));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    for (int mode = 0; mode < 2; ++mode) {
        unsigned long long tot[16] = {0};
        for (int rep = 0; rep < 20; ++rep) {
            CK(hipMemsetAsync(mm, 0, 64, s1));
            CK(hipStreamSynchronize(s1));
            if (mode) hipLaunchKernelGGL(aggressor, dim3(4096), dim3(256), 0, s2, src, dst, 4000);
            hipLaunchKernelGGL(victim, dim3(768), dim3(256), 0, s1, src, mm, 20000);
            CK(hipDeviceSynchronize());
            unsigned m[16];
            CK(hipMemcpy(m, mm, 64, hipMemcpyDeviceToHost));
            for (int k = 0; k < 16; ++k) tot[k] += m[k];
        }
        printf("%s: lanes with a wrong packed sum over 20 launches (of %d): plain %llu, op_sel_hi:[1,0,1] %llu, op_sel:[0,1,0] %llu, swap %llu\n",
               mode ? "beside the MFMA kernel" : "alone", 20 * 768 * 256, tot[0], tot[1], tot[2], tot[3]);
        printf("    register copies: second dword early != late %llu times, first dword %llu times; late second dword != LDS content %llu times\n", tot[4], tot[5], tot[6]);
        printf("    single instructions with a wrong result (of %.3g each): v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[0,1] %llu, v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0] %llu, v_pk_fma_f32 op_sel:[1,0,0] op_sel_hi:[0,1,1] %llu, v_pk_fma_f32 op_sel:[0,0,1] op_sel_hi:[1,1,0] %llu, v_pk_add_f32 op_sel:[1,0] op_sel_hi:[0,1] %llu\n", 20.0 * 768 * 256 * 20000, tot[7], tot[8], tot[9], tot[10], tot[11]);
    }
    return 0;
}
