// Implicit-GEMM KSxKS convolution (stride 1, "same" zero padding) on gfx950 matrix cores,
// with the consumer's pointwise work fused into the epilogue.
//
// GEMM view (per image):  D[row = output channel][col = pixel] = sum_k Wp[row][k] * In[k][col],
// k = (tap, channel block, channel-in-block).  v_mfma_f32_32x32x16_bf16, weights as the A
// operand, pixels as the B operand:
//   * one workgroup (4 waves) owns a TH x 32 pixel tile of one image; the (TH+2) x 34 input
//     halo tile of every input channel block is staged once into LDS as 16-byte units
//     [channel block][row][col] (8 channels of one pixel per unit), so the B fragment of a
//     32-pixel row at any tap is one conflict-free ds_read_b128 per lane at an immediate offset;
//   * the weights are stationary in registers: a wave holds the A fragments of ONE 32-row
//     output tile for its share of K (16 bytes per lane per k-step), pre-packed by
//     plc_pack_weight_kernel in exactly the lane order the MFMA wants, and sweeps the tile's
//     pixel rows NREP at a time (NREP independent accumulators);
//   * KSPLIT = 2 (used when K is too deep for one wave's registers, e.g. the data-gradient
//     convs whose K is 4*Ch*9): two waves share an output tile, each contracts half of the
//     input channel blocks, and the partial sums are exchanged through LDS once per pass so
//     that each of the two waves finishes one of the pass's two pixel rows;
//   * the row order inside a 32-row output tile is chosen per epilogue so that everything the
//     epilogue combines sits in ONE lane: accumulator register 4g+j of lane (half, x) is row
//     8g + 4*half + j, pixel x.  For the ConvLSTM cell g is the gate (i,f,o,g) and 4*half+j the
//     hidden channel inside an 8-channel block, so sigma/tanh/state update need no cross-lane
//     traffic and h'/c' leave as 8/16-byte stores that are contiguous across the wave.
#pragma once
#include <type_traits>

#include "plc_common.h"

#ifdef PLC_TIMELINE
// Instrumented build only (never the shipped library): lane 0 of every wave records the 100 MHz
// wall clock at phase boundaries into plc_timeline[(block * 4 + wave) * 8 + k].
__device__ long long* plc_timeline = nullptr;
__device__ int plc_timeline_sel = -1;   // only the instantiation with this id records (-1: all)
#define PLC_TL_ID (NCB0 * 10000 + NCB1 * 1000 + STAGE * 100 + EPI * 10 + KSPLIT)
#define PLC_STAMP(k)                                                                                   \
    do {                                                                                               \
        long long* tl_ = (plc_timeline_sel < 0 || plc_timeline_sel == PLC_TL_ID) ? plc_timeline : nullptr; \
        if (tl_ && (threadIdx.x & 63) == 0)                                                            \
            tl_[((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 10 + (k)] = (long long)wall_clock64(); \
    } while (0)
#define PLC_STAMP_HWID()                                                                               \
    do {                                                                                               \
        long long* tl_ = (plc_timeline_sel < 0 || plc_timeline_sel == PLC_TL_ID) ? plc_timeline : nullptr; \
        if (tl_ && (threadIdx.x & 63) == 0) {                                                          \
            tl_[((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 10 + 8] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));  \
            tl_[((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 10 + 9] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)); \
        }                                                                                              \
    } while (0)
#else
#define PLC_STAMP(k) do {} while (0)
#define PLC_STAMP_HWID() do {} while (0)
#endif

enum { STAGE_CB8 = 0, STAGE_CB8_GATED = 1, STAGE_COORD = 2, STAGE_F32_1CH = 3 };
// EPI_LSTM_SAVE = EPI_LSTM that also stores the gate activations i,f,o,g for the backward pass
enum { EPI_LSTM = 0, EPI_ACT_CB8 = 1, EPI_PIXSHUF_RELU = 2, EPI_F32_NCHW = 3, EPI_LSTM_SAVE = 4 };
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SIGMOID = 2 };
enum { MUL_PRODUCT = 0, MUL_RELU_MASK = 1 };

struct ConvArgs {
    int H, W;            // input == output spatial size (PixelShuffle epilogue writes 2H x 2W)
    int tiles_x, tiles_y;
    int n_tiles;         // images * tiles_y * tiles_x
    int n_blocks;        // workgroups per M group (gridDim.x); workgroup b takes tiles b, b + n_blocks, ...
    int n_in;            // inner image count for PlcTensor addressing
    int ncb_out;         // EPI_ACT_CB8: number of 8-channel blocks actually stored
    int cout;            // EPI_F32_NCHW: number of fp32 channels stored
    int act;             // EPI_ACT_CB8 / EPI_F32_NCHW activation
    int save_gates;      // EPI_LSTM: also store i,f,o,g (bf16) for the backward pass
    int zero_state;      // EPI_LSTM: the recurrent inputs (src1 = h, c_in) are all zero and are not read
    int mul_mode;        // EPI_ACT_CB8 with `mul`: MUL_PRODUCT (v *= m) or MUL_RELU_MASK (v = m > 0 ? v : 0)
    int g_ncb0, g_ncb1;  // run-time-width kernel (plc_convg_kernel.h) only: channel blocks of src0 / src1
    int g_nmt;           // ... and the conv's 32-row output tiles
    int seg_rows;        // rolling-window kernel (plc_convr_kernel.h) only: rows per segment
    int t_inner, t_segs; // gated staging only, t_inner > 0: tiles are walked time-innermost (see plc_decode_tile)
    unsigned* sync;      // persistent recurrence (PERSIST != 0): the device-wide barrier's counter, zero at launch
    PlcTensor src0;      // CB8 bf16, first NCB0 channel blocks   (STAGE_COORD / STAGE_F32_1CH: fp32 plane)
    PlcTensor src1;      // CB8 bf16, next NCB1 channel blocks
    PlcTensor gate;      // STAGE_CB8_GATED: CB8 bf16 multiplier of src0
    const float* row_tab;  // STAGE_COORD: linspace(0,1,H)
    const float* col_tab;  // STAGE_COORD: linspace(0,1,W)
    const uint4* wpk;    // packed A fragments [M tile][k split][k step][lane] x 16 B
    const float* bpk;    // packed bias [M tile][32 rows]
    PlcTensor c_in;      // EPI_LSTM: previous cell state, CB8 fp32
    PlcTensor out0;      // EPI_LSTM: h' (CB8 bf16); EPI_ACT_CB8/PIXSHUF: out (CB8 bf16); EPI_F32_NCHW: fp32
    PlcTensor out1;      // EPI_LSTM: c' (CB8 fp32); EPI_ACT_CB8: optional copy of the value before `mul`
    PlcTensor out2;      // EPI_LSTM: gates i,f,o,g (CB8 bf16, 4*Ch channels) when save_gates
    PlcTensor mul;       // EPI_ACT_CB8: optional CB8 bf16 tensor combined into the output (p may be null)
};

// source of out-of-image units for the LDS-DMA staging
__device__ const uint4 plc_zero16 = {0u, 0u, 0u, 0u};

// Loads through a selected pointer, pinned to the global address space (a select between a tensor
// address and a __device__ constant otherwise degrades to a flat load, which also ties up lgkmcnt).
#define PLC_AS1 __attribute__((address_space(1)))
typedef unsigned int plc_u32x4 __attribute__((ext_vector_type(4)));     // uint4 is a class: its copy goes through a generic reference
typedef float plc_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 plc_gload16(const char* p) {
    const plc_u32x4 v = *(const PLC_AS1 plc_u32x4*)p;
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 plc_gload_f4(const char* p) {
    const plc_f32x4 v = *(const PLC_AS1 plc_f32x4*)p;
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float plc_gload4(const char* p) { return *(const PLC_AS1 float*)p; }

// One 16-byte LDS unit (8 channels of pixel (gy, gx) of channel block cb) of the conv input.
// The fp32-plane stagings select the ADDRESS (the tensor, or a zero constant outside the image) and
// load unconditionally: a load inside `if (ok)` into a zero-initialised register makes the compiler
// wait for every earlier load of the staging loop before the next one is issued.
template <int STAGE, int NCB0>
__device__ __forceinline__ uint4 plc_stage_unit(const ConvArgs& a, const char* s0, const char* s1, const char* gt,
                                                int cb, int gy, int gx, int H, int W) {
    uint4 v = make_uint4(0, 0, 0, 0);
    const bool ok = (gy >= 0) & (gy < H) & (gx >= 0) & (gx < W);
    if constexpr (STAGE == STAGE_COORD || STAGE == STAGE_F32_1CH) {
        const char* const z = (const char*)&plc_zero16;
        const float rv = plc_gload4(ok ? s0 + ((long long)gy * W + gx) * 4 : z);
        if constexpr (STAGE == STAGE_COORD) {
            const float row = plc_gload4(ok ? (const char*)(a.row_tab + gy) : z);
            const float col = plc_gload4(ok ? (const char*)(a.col_tab + gx) : z);
            v.x = pack_bf2(rv, row);        // all three are 0 outside the image: the unit is 0
            v.y = pack_bf2(col, 0.0f);
        } else {
            v.x = pack_bf2(rv, 0.0f);
        }
        return v;
    }
    if (ok) {
        const long long plane_b = (long long)H * W * 16;
        const long long pix = ((long long)gy * W + gx) * 16;
        if (STAGE == STAGE_COORD) {
            const float rv = ((const float*)s0)[(long long)gy * W + gx];
            v.x = pack_bf2(rv, a.row_tab[gy]);
            v.y = pack_bf2(a.col_tab[gx], 0.0f);
        } else if (STAGE == STAGE_F32_1CH) {
            v.x = pack_bf2(((const float*)s0)[(long long)gy * W + gx], 0.0f);
        } else if (STAGE == STAGE_CB8_GATED) {
            const uint4 f = *(const uint4*)(s0 + cb * plane_b + pix);
            const uint4 g = *(const uint4*)(gt + cb * plane_b + pix);
            const float4 f0 = unpack_bf4(make_uint2(f.x, f.y)), f1 = unpack_bf4(make_uint2(f.z, f.w));
            const float4 g0 = unpack_bf4(make_uint2(g.x, g.y)), g1 = unpack_bf4(make_uint2(g.z, g.w));
            const uint2 lo = pack_bf4(f0.x * g0.x, f0.y * g0.y, f0.z * g0.z, f0.w * g0.w);
            const uint2 hi = pack_bf4(f1.x * g1.x, f1.y * g1.y, f1.z * g1.z, f1.w * g1.w);
            v = make_uint4(lo.x, lo.y, hi.x, hi.y);
        } else {
            const char* base = (cb < NCB0) ? (s0 + cb * plane_b) : (s1 + (cb - NCB0) * plane_b);
            v = *(const uint4*)(base + pix);
        }
    }
    return v;
}

// Epilogue of one pixel row of one 32-row output tile: acc register 4g+j = row 8g + 4*half + j.
// FULL = the whole tile lies inside the image (no per-lane bounds check, hence no branch).
// PLAIN (EPI_ACT_CB8 only) = the caller guarantees no second output, no multiplier and no sigmoid: the
// epilogue then contains no load at all (a load on any path makes the compiler wait for outstanding
// vector-memory operations wherever its destination registers are reused).
template <int EPI, bool FULL = false, bool PLAIN = false>
__device__ __forceinline__ void plc_epilogue(const ConvArgs& a, const f32x16_t& acc, int n, int mt, int n_mt,
                                             int gy, int gx, int half, const float4& cp) {
    const int H = a.H, W = a.W;
    if constexpr (!FULL) {
        if (gy >= H || gx >= W) return;
    }
    const long long pix = (long long)gy * W + gx;
    const long long plane_b = (long long)H * W * 16;
    if (EPI == EPI_LSTM || EPI == EPI_LSTM_SAVE) {
        const long long coff = ((long long)mt * H * W + pix) * 32 + half * 16;
        const long long hoff = ((long long)mt * H * W + pix) * 16 + half * 8;
        // cp = previous cell state of this lane's 4 channels, fetched before the MFMA loop
        float ig[4], fg[4], og[4], gg[4], cn[4], hn[4];
        const float cpv[4] = {cp.x, cp.y, cp.z, cp.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ig[j] = fast_sigmoid(acc[j]);
            fg[j] = fast_sigmoid(acc[4 + j]);
            og[j] = fast_sigmoid(acc[8 + j]);
            gg[j] = fast_tanh(acc[12 + j]);
            cn[j] = __builtin_fmaf(fg[j], cpv[j], ig[j] * gg[j]);
            hn[j] = og[j] * fast_tanh(cn[j]);
        }
        *(float4*)(plc_img(a.out1, n, a.n_in) + coff) = make_float4(cn[0], cn[1], cn[2], cn[3]);
        *(uint2*)(plc_img(a.out0, n, a.n_in) + hoff) = pack_bf4(hn[0], hn[1], hn[2], hn[3]);
        if (EPI == EPI_LSTM_SAVE) {
            // gates tensor: CB8 bf16 with 4*Ch channels in conv-output order (gate-major); n_mt = Ch / 8
            char* gp = plc_img(a.out2, n, a.n_in) + pix * 16 + half * 8;
            *(uint2*)(gp + (long long)(0 * n_mt + mt) * plane_b) = pack_bf4(ig[0], ig[1], ig[2], ig[3]);
            *(uint2*)(gp + (long long)(1 * n_mt + mt) * plane_b) = pack_bf4(fg[0], fg[1], fg[2], fg[3]);
            *(uint2*)(gp + (long long)(2 * n_mt + mt) * plane_b) = pack_bf4(og[0], og[1], og[2], og[3]);
            *(uint2*)(gp + (long long)(3 * n_mt + mt) * plane_b) = pack_bf4(gg[0], gg[1], gg[2], gg[3]);
        }
    } else if (EPI == EPI_ACT_CB8) {
        char* op = plc_img(a.out0, n, a.n_in);
        char* op1 = a.out1.p ? plc_img(a.out1, n, a.n_in) : nullptr;
        const char* mp = a.mul.p ? plc_img(a.mul, n, a.n_in) : nullptr;
        // The options are wave-uniform, so each test below becomes a scalar branch; taken per element
        // group that was ~50 branches per stored row (1.4 us per pass of the cell-2 data-gradient
        // conv).  The common case -- no second output, no multiplier, identity or ReLU -- is decided
        // once per row and then runs straight-line.
        if (PLAIN || (!op1 && !mp && a.act != ACT_SIGMOID)) {
            char* const o = op + pix * 16 + half * 8;
            const int n_cb = a.ncb_out - 4 * mt;          // channel blocks of this M tile that are stored
            auto row = [&](auto relu_c) {
                constexpr bool RELU = decltype(relu_c)::value;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = RELU ? fmaxf(acc[4 * g + j], 0.0f) : acc[4 * g + j];
                    if (g < n_cb) *(uint2*)(o + (long long)(4 * mt + g) * plane_b) = pack_bf4(v[0], v[1], v[2], v[3]);
                }
            };
            if (a.act == ACT_RELU) row(std::true_type{});
            else row(std::false_type{});
            return;
        }
        // Likewise the data-gradient convs that apply a ReLU mask (or a product) to a plain output:
        // the row's multiplier units are requested together, then combined and stored.
        if (!op1 && mp && a.act == ACT_NONE) {
            const long long o = pix * 16 + half * 8;
            const int n_cb = a.ncb_out - 4 * mt;
            if (n_cb <= 0) return;
            uint2 mraw[4];
#pragma unroll
            for (int g = 0; g < 4; ++g)       // blocks past n_cb re-read block 0 (a valid address), unused
                mraw[g] = *(const uint2*)(mp + o + (long long)(4 * mt + (g < n_cb ? g : 0)) * plane_b);
            auto row = [&](auto mask_c) {
                constexpr bool MASK = decltype(mask_c)::value;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 mv = unpack_bf4(mraw[g]);
                    float v[4] = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                    if (MASK) {
                        v[0] = mv.x > 0.0f ? v[0] : 0.0f; v[1] = mv.y > 0.0f ? v[1] : 0.0f;
                        v[2] = mv.z > 0.0f ? v[2] : 0.0f; v[3] = mv.w > 0.0f ? v[3] : 0.0f;
                    } else {
                        v[0] *= mv.x; v[1] *= mv.y; v[2] *= mv.z; v[3] *= mv.w;
                    }
                    if (g < n_cb) *(uint2*)(op + o + (long long)(4 * mt + g) * plane_b) = pack_bf4(v[0], v[1], v[2], v[3]);
                }
            };
            if (a.mul_mode == MUL_RELU_MASK) row(std::true_type{});
            else row(std::false_type{});
            return;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int cb = 4 * mt + g;
            if (cb >= a.ncb_out) continue;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t = acc[4 * g + j];
                if (a.act == ACT_RELU) t = fmaxf(t, 0.0f);
                else if (a.act == ACT_SIGMOID) t = fast_sigmoid(t);
                v[j] = t;
            }
            const long long off = ((long long)cb * H * W + pix) * 16 + half * 8;
            if (op1) *(uint2*)(op1 + off) = pack_bf4(v[0], v[1], v[2], v[3]);
            if (mp) {
                const float4 mv = unpack_bf4(*(const uint2*)(mp + off));
                if (a.mul_mode == MUL_RELU_MASK) {
                    v[0] = mv.x > 0.0f ? v[0] : 0.0f; v[1] = mv.y > 0.0f ? v[1] : 0.0f;
                    v[2] = mv.z > 0.0f ? v[2] : 0.0f; v[3] = mv.w > 0.0f ? v[3] : 0.0f;
                } else {
                    v[0] *= mv.x; v[1] *= mv.y; v[2] *= mv.z; v[3] *= mv.w;
                }
            }
            *(uint2*)(op + off) = pack_bf4(v[0], v[1], v[2], v[3]);
        }
    } else if (EPI == EPI_PIXSHUF_RELU) {
        // row 8g + 4*half + j  <->  conv channel 4*(8*mt + 4*half + j) + g ; g = 2*i + jj
        char* op = plc_img(a.out0, n, a.n_in);
        const int H2 = 2 * H, W2 = 2 * W;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int oy = 2 * gy + (g >> 1), ox = 2 * gx + (g & 1);
            const long long off = (((long long)mt * H2 + oy) * W2 + ox) * 16 + half * 8;
            *(uint2*)(op + off) = pack_bf4(fmaxf(acc[4 * g + 0], 0.0f), fmaxf(acc[4 * g + 1], 0.0f),
                                           fmaxf(acc[4 * g + 2], 0.0f), fmaxf(acc[4 * g + 3], 0.0f));
        }
    } else {  // EPI_F32_NCHW: channel = 32*mt + 8g + 4*half + j
        float* op = (float*)plc_img(a.out0, n, a.n_in);
        if (a.cout == 1) {
            // single-channel output (the generator's last conv): channel 0 is register 0 of the
            // half-0 lanes of M tile 0; one store instead of 16 channel tests per lane
            if (mt == 0 && half == 0) {
                float t = acc[0];
                if (a.act == ACT_RELU) t = fmaxf(t, 0.0f);
                else if (a.act == ACT_SIGMOID) t = fast_sigmoid(t);
                op[pix] = t;
            }
            return;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ch = 32 * mt + 8 * g + 4 * half + j;
                if (ch >= a.cout) continue;
                float t = acc[4 * g + j];
                if (a.act == ACT_RELU) t = fmaxf(t, 0.0f);
                else if (a.act == ACT_SIGMOID) t = fast_sigmoid(t);
                op[(long long)ch * H * W + pix] = t;
            }
    }
}

// Compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>).
template <int... I, class F>
__device__ __forceinline__ void plc_static_for_seq(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void plc_static_for(F&& f) {
    plc_static_for_seq(std::make_integer_sequence<int, N>{}, f);
}

// Ordinal of the MFMA that combines fragment f with output row r in the fragment-centric K loop
// (fragment f sits at LDS row rho = f / (KS * NCBP); it feeds row r iff 0 <= rho - r < KS).
constexpr int plc_mfma_index(int f, int r, int KS, int NCBP, int NREP) {
    int m = 0;
    for (int ff = 0; ff <= f; ++ff) {
        const int rho = ff / (KS * NCBP);
        for (int rr = 0; rr < NREP; ++rr) {
            const int dy = rho - rr;
            if (dy >= 0 && dy < KS) {
                if (ff == f && rr == r) return m;
                ++m;
            }
        }
    }
    return m;
}

// Dynamic LDS bytes of one instantiation.
template <int NCB, int KS, int TH, int KSPLIT, int STAGE = STAGE_CB8>
constexpr int plc_conv_lds_bytes() {
    // tile rounded up to whole 64-unit LDS-DMA pieces (twice for the gated staging: the multiplier's
    // tile lands next to it), the K-split exchange buffer, 32 units of bias
    return ((NCB * (TH + 2 * (KS / 2)) * (32 + 2 * (KS / 2)) + 63) / 64 * 64 * (STAGE == STAGE_CB8_GATED ? 2 : 1) +
            (KSPLIT == 2 ? 4 * 4 * 64 : 0) + 32) * 16;
}

// Workgroup barrier that orders LDS accesses only.  __syncthreads() carries a workgroup-scope
// fence, for which the compiler also waits vmcnt(0): after an epilogue that means waiting for the
// global stores to retire before any wave may go on.  Where the barrier protects nothing but LDS
// contents (partial-sum exchange, reuse of the tile buffer) this one is enough: the wave's own LDS
// operations have completed (lgkmcnt) and the clobbers keep the compiler from moving memory
// accesses across it.
__device__ __forceinline__ void plc_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Device-wide barrier of a grid that is resident all at once: arrive with a release at agent scope (this
// workgroup's stores leave the XCD's L2), poll the counter with relaxed device-coherent loads, then one
// acquire fence (stale lines are dropped once, not on every poll).  `target` = arrivals that complete
// this barrier; the counter only grows, so barrier k of a launch uses (k + 1) * gridDim.x.
__device__ __forceinline__ void plc_grid_barrier(unsigned* counter, unsigned target) {
    __syncthreads();                               // this workgroup's stores have been issued and counted
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target)
            __builtin_amdgcn_s_sleep(1);
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction takes an immediate): waits until
// at most n of this wave's vector-memory operations are outstanding.  Values above the table wait
// for a smaller count, which is only stricter.
__device__ __forceinline__ void plc_wait_vmcnt_le(int n) {
#define PLC_VMCNT_CASE(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n) {
        PLC_VMCNT_CASE(1) PLC_VMCNT_CASE(2) PLC_VMCNT_CASE(3) PLC_VMCNT_CASE(4) PLC_VMCNT_CASE(5) PLC_VMCNT_CASE(6)
        PLC_VMCNT_CASE(7) PLC_VMCNT_CASE(8) PLC_VMCNT_CASE(9) PLC_VMCNT_CASE(10) PLC_VMCNT_CASE(11) PLC_VMCNT_CASE(12)
        PLC_VMCNT_CASE(13) PLC_VMCNT_CASE(14) PLC_VMCNT_CASE(15) PLC_VMCNT_CASE(16)
        default:
            if (n > 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#undef PLC_VMCNT_CASE
}

// Waves per SIMD to compile for: three (three workgroups per CU) for the light convs, whose time is
// load/store latency rather than MFMA issue; two when the stationary weights need the registers.
template <int NCB, int KS, int KSPLIT, int NMT, int EPI>
constexpr int plc_conv_min_waves() {
    constexpr int weight_vgprs = ((KS * KS * (NCB / KSPLIT) + 1) / 2) * 4;
    // the fp32 NCHW epilogue needs its 64 accumulators live together with the store addresses
    constexpr int limit = EPI == EPI_F32_NCHW ? 40 : 80;
    return (NMT == 1 && EPI != EPI_LSTM && EPI != EPI_LSTM_SAVE && weight_vgprs <= limit) ? 3 : 2;
}

// A wave's stationary A fragments: NKS k-steps, 16 bytes per lane each (`src` is already per lane).
template <int NKS>
__device__ __forceinline__ void plc_load_weights(bf16x8_t (&wf)[NKS], const uint4* src) {
#pragma unroll
    for (int s = 0; s < NKS; ++s) wf[s] = __builtin_bit_cast(bf16x8_t, src[s * 64]);
}

// The same loads issued as inline asm: the compiler keeps no count of them, so the caller must wait
// (s_waitcnt vmcnt) itself before the first use of wf.
template <int NKS>
__device__ __forceinline__ void plc_load_weights_asm(bf16x8_t (&wf)[NKS], const uint4* src) {
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
        const uint4* p = src + s * 64;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(wf[s]) : "v"(p) : "memory");
    }
}

// Whether a workgroup of this instantiation walks several tiles.  The LSTM convs (register file and
// scalar registers full) take exactly one tile per workgroup, and so do the convs that stage
// through registers: they were measured to lose more from fewer workgroups in flight than they
// gain from fetching their (few) weights once.
template <int STAGE, int EPI>
constexpr bool plc_conv_multi_tile() {
    return EPI != EPI_LSTM && EPI != EPI_LSTM_SAVE && (STAGE == STAGE_CB8 || STAGE == STAGE_CB8_GATED);
}

// LDS-DMA requests of one input halo tile (DMA stagings): each wave instruction moves 64 consecutive
// units (1 KiB) straight from global memory into the tile, no VGPRs and no ds_write; out-of-image
// units read a 16-byte block of zeros instead.  All of a wave's pieces are in flight together (the
// caller waits).  Gated staging: the multiplier's tile is fetched the same way into a second region
// GATE_OFF units after the tile.  SPLIT: the pieces holding rows that pass 0 of each of the RG row
// groups reads are requested first; the return value is the number of later pieces this wave issued
// (loads retire in order, so "all but the n_late youngest" is exactly the early set).
// with_gate (wave-uniform) = false: the multiplier's tile is already in LDS (same pixels, an earlier time step).
template <int NCB0, int NCB1, int KS, int TH, int STAGE, bool SPLIT, int RG, int RPW, int NREP, int TILE_UNITS, int GATE_OFF>
__device__ __forceinline__ int plc_dma_issue(const ConvArgs& a, uint4* tile, int lane, int wave_u, int n, int x0, int y0,
                                             int zero_state, bool with_gate = true) {
    constexpr int NCB = NCB0 + NCB1, PAD = KS / 2, LH = TH + 2 * PAD, LW = 32 + 2 * PAD, PLANE = LH * LW;
    constexpr int NCHUNK = TILE_UNITS / 64;
    const int H = a.H, W = a.W;
    const char* s0 = plc_img(a.src0, n, a.n_in);
    const char* s1 = NCB1 > 0 ? plc_img(a.src1, n, a.n_in) : nullptr;
    const char* gt = STAGE == STAGE_CB8_GATED ? plc_img(a.gate, n, a.n_in) : nullptr;
    const long long plane_b = (long long)H * W * 16;
    int n_late = 0;
    for (int phase = 0; phase < (SPLIT ? 2 : 1); ++phase)
        for (int ch = wave_u; ch < NCHUNK; ch += 4) {
            if constexpr (SPLIT) {
                // wave-uniform: does piece ch hold a unit that pass 0 reads?
                const int r0 = (ch * 64) % PLANE;
                bool early = r0 + 63 >= PLANE;                      // runs into the next plane's first rows
                const int lya = r0 / LW, lyb = (r0 + 63) / LW;
#pragma unroll
                for (int g = 0; g < RG; ++g) early |= (lya <= g * RPW + NREP + KS - 2) & (lyb >= g * RPW);
                if (early != (phase == 0)) continue;
                n_late += phase;
            }
            const int u = ch * 64 + lane;
            const int cb = u / PLANE;
            const int r = u - cb * PLANE;
            const int ly = r / LW;
            const int lx = r - ly * LW;
            const int gy = y0 - PAD + ly, gx = x0 - PAD + lx;
            const bool ok = (cb < NCB) & (gy >= 0) & (gy < H) & (gx >= 0) & (gx < W) & !(zero_state && cb >= NCB0);
            const char* base = (cb < NCB0) ? (s0 + cb * plane_b) : (s1 + (cb - NCB0) * plane_b);
            const long long poff = cb * plane_b + ((long long)gy * W + gx) * 16;
            const char* src = ok ? base + ((long long)gy * W + gx) * 16 : (const char*)&plc_zero16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(tile + ch * 64), 16, 0, 0);
            if constexpr (STAGE == STAGE_CB8_GATED) {
                if (with_gate) {
                    const char* gsrc = ok ? gt + poff : (const char*)&plc_zero16;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                                     (__attribute__((address_space(3))) void*)(tile + GATE_OFF + ch * 64), 16, 0, 0);
                }
            }
        }
    return n_late;
}

// Tile index (image-major, then tile row, then tile column) -> image and pixel origin.
// Time-innermost order (a.t_inner > 0; the images are n = t * n_in + b): the index runs over the t_inner
// time steps of a chain first, then over the t_segs chains of a (b, tile row, tile column) position, then
// over the positions, so that a workgroup meets the same pixels of consecutive time steps one after
// another and whatever depends on b and the position alone -- the multiplier tile of the gated
// staging -- stays in LDS along a chain.
__device__ __forceinline__ void plc_decode_tile(const ConvArgs& a, int th, int id, int& n, int& x0, int& y0) {
    if (a.t_inner > 0) {
        const int tt = id % a.t_inner;
        id /= a.t_inner;
        const int seg = id % a.t_segs;
        id /= a.t_segs;
        const int tx = id % a.tiles_x;
        id /= a.tiles_x;
        const int ty = id % a.tiles_y;
        n = (seg * a.t_inner + tt) * a.n_in + id / a.tiles_y;
        x0 = tx * 32;
        y0 = ty * th;
        return;
    }
    const int tx = id % a.tiles_x;
    id /= a.tiles_x;
    const int ty = id % a.tiles_y;
    n = id / a.tiles_y;
    x0 = tx * 32;
    y0 = ty * th;
}

// A workgroup's first tile and the one after `id`: tiles block_x, block_x + n_blocks, ...; in the
// time-innermost order CHAINS block_x, block_x + n_blocks, ..., each walked through its t_inner steps.
__device__ __forceinline__ int plc_first_tile(const ConvArgs& a, int block_x) {
    return a.t_inner > 0 ? block_x * a.t_inner : block_x;
}
__device__ __forceinline__ int plc_next_tile(const ConvArgs& a, int id) {
    if (a.t_inner > 0) return (id + 1) % a.t_inner ? id + 1 : id + 1 + (a.n_blocks - 1) * a.t_inner;
    return id + a.n_blocks;
}

// The work of one workgroup: tile `block_x` (image-major, then tile row, then tile column) of M
// group `block_y` of `n_mgroups`; `tile` is the workgroup's dynamic LDS.
// PERSIST (ConvLSTM cells only): the workgroup keeps its weights and walks the SAME spatial tile through
// all time steps (tile numbers run over the steps: block_x, block_x + n_blocks, ...; the tensors' outer
// stride is one step), meeting every other workgroup of the grid at a device-wide barrier between
// steps: PERSIST == 1 (cell 1) arrives AFTER each step, PERSIST == 2 (cell 2) BEFORE each step, so cell 2
// of step t runs beside cell 1 of step t + 1.  a.zero_state then applies to the first step only.
template <int NCB0, int NCB1, int NMT, int KS, int STAGE, int EPI, int TH, int KSPLIT = 1, int PERSIST = 0>
__device__ __forceinline__ void plc_conv_body(const ConvArgs& a, const int block_x, const int block_y,
                                              const int n_mgroups, uint4* const tile) {
    constexpr int NCB = NCB0 + NCB1;
    constexpr int NCBW = NCB / KSPLIT;       // channel blocks contracted by one wave
    constexpr int TAPS = KS * KS;
    constexpr int UNITS = TAPS * NCBW;
    constexpr int NKS = (UNITS + 1) / 2;     // k-steps per wave
    constexpr int PAD = KS / 2;
    constexpr int LH = TH + 2 * PAD;
    constexpr int LW = 32 + 2 * PAD;
    constexpr int PLANE = LH * LW;           // units per channel block
    constexpr int RG = 4 / (NMT * KSPLIT);   // row groups (wave sets sharing one M tile)
    constexpr int RPW = TH / RG;             // rows per wave (per wave pair when KSPLIT == 2)
    // rows (N tiles) per pass: 4 accumulators, or 2 when the stationary weights already take
    // more than 64 VGPRs (a single 32x32x16 accumulation chain issues back to back anyway)
    constexpr int NREP_MAX = (KSPLIT == 2 || NKS * 4 > 64) ? 2 : 4;
    constexpr int NREP = RPW < NREP_MAX ? RPW : NREP_MAX;
    constexpr int PASSES = RPW / NREP;
    static_assert(KSPLIT == 1 || KSPLIT == 2, "KSPLIT must be 1 or 2");
    static_assert(NCB % KSPLIT == 0, "channel blocks must split evenly");
    static_assert(NMT * KSPLIT == 1 || NMT * KSPLIT == 2 || NMT * KSPLIT == 4, "NMT*KSPLIT must be 1, 2 or 4");
    static_assert(RPW >= 1 && RPW % NREP == 0, "tile rows must split evenly over waves");
    static_assert(KSPLIT == 1 || NREP == 2, "K split needs two rows per pass");
    constexpr int RED_UNITS = KSPLIT == 2 ? 4 * 4 * 64 : 0;   // partial-sum exchange, 16 B units
    constexpr int TILE_UNITS = (NCB * PLANE + 63) / 64 * 64;   // whole 64-unit DMA pieces
    constexpr int BIAS_UNITS = 32;                             // 4 M tiles x 32 rows of fp32 bias
    constexpr int GATE_UNITS = STAGE == STAGE_CB8_GATED ? TILE_UNITS : 0;   // multiplier tile, after the bias
    static_assert((TILE_UNITS + RED_UNITS + BIAS_UNITS + GATE_UNITS) * 16 <= 160 * 1024, "LDS tile too large");
    static_assert((TILE_UNITS + RED_UNITS + BIAS_UNITS + GATE_UNITS) * 16 == plc_conv_lds_bytes<NCB, KS, TH, KSPLIT, STAGE>(),
                  "LDS size mismatch");
    static_assert(STAGE != STAGE_CB8_GATED || NCB1 == 0, "the gated staging takes one source");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int half = lane >> 5;
    const int px = lane & 31;
    const int m = wave % NMT;
    const int ks = (wave / NMT) % KSPLIT;
    const int rg = wave / (NMT * KSPLIT);
    const int mt = block_y * NMT + m;        // global M tile index
    const int n_mt = n_mgroups * NMT;

    const int H = a.H, W = a.W;

    PLC_STAMP(0);
    PLC_STAMP_HWID();
    constexpr bool IS_LSTM = (EPI == EPI_LSTM || EPI == EPI_LSTM_SAVE);
    constexpr bool PIPE = IS_LSTM && KSPLIT == 1 && (NCBW % 2 == 0);
    constexpr bool SPLIT_STAGE = PIPE && STAGE == STAGE_CB8 && PASSES > 1 && NMT % 2 == 0;   // two-phase wait on the halo tile

    // ---- weights: stationary A fragments, fetched once per workgroup ------------------------
    bf16x8_t wf[NKS];
    const uint4* const wsrc = a.wpk + ((long long)(mt * KSPLIT + ks) * NKS) * 64 + lane;
    // (requesting the halo tile before the weights was measured 1.5 us slower per cell launch)
    if constexpr (SPLIT_STAGE) {
        // Two-phase staged tile: weights, bias and tile pieces are requested back to back and the
        // compiler must not count any of them, or it drains ALL of them (it cannot bound the
        // number of tile pieces) before the first MFMA reads a weight register.  The weights are
        // fetched by asm loads and the bias goes straight to LDS by DMA; both are older than every
        // tile piece, so the wait for the early pieces (loads retire in order) covers them.
        plc_load_weights_asm<NKS>(wf, wsrc);
        const int wave_b = __builtin_amdgcn_readfirstlane(wave);
        if (wave_b < NMT / 2)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.bpk + block_y * NMT * 32 + tid),
                                             (__attribute__((address_space(3))) void*)((float*)(tile + TILE_UNITS + RED_UNITS) + wave_b * 64),
                                             4, 0, 0);
    } else {
        plc_load_weights<NKS>(wf, wsrc);
        // this workgroup's bias rows (row order of the packed M tiles) for the pipelined LSTM path
        if (tid < NMT * 32) ((float*)(tile + TILE_UNITS + RED_UNITS))[tid] = a.bpk[block_y * NMT * 32 + tid];
    }
    float4 bv[4];   // bias of this lane's 16 rows (accumulator initial value of the generic pass loop)
    auto load_bias = [&]() {
        const float4* bias4 = (const float4*)a.bpk + mt * 8;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            // the K-split instantiations are the bias-free data-gradient convs
            bv[g] = KSPLIT == 2 ? make_float4(0.f, 0.f, 0.f, 0.f) : bias4[2 * g + half];
        }
        // retire these loads here: a load still pending at the loop head would make the compiler
        // wait vmcnt(0) before the first MFMA of every pass, i.e. also for the prefetch issued there
#pragma unroll
        for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(bv[g].x), "+v"(bv[g].y), "+v"(bv[g].z), "+v"(bv[g].w));
    };
    if constexpr (!PIPE) load_bias();   // the pipelined LSTM path takes its bias from LDS instead

    // ---- the workgroup's tiles: block_x, block_x + n_blocks, ... ----------------------------
    // A launch holds at most as many workgroups as the chip keeps resident; with more tiles than
    // that each workgroup walks several, so the weights (whose re-fetch from L2 would otherwise
    // exceed the activation traffic of the light convs) cross the fabric once per workgroup.
    constexpr int NCHUNK = TILE_UNITS / 64;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // (free functions and a macro rather than capturing lambdas: this compiler's simplifycfg crashes
    // on lambdas of this shape in the multi-tile instantiations)
#define PLC_DMA_ISSUE(n_, x0_, y0_, gate_)                                                              \
    plc_dma_issue<NCB0, NCB1, KS, TH, STAGE, SPLIT_STAGE, RG, RPW, NREP, TILE_UNITS,                    \
                  TILE_UNITS + RED_UNITS + BIAS_UNITS>(a, tile, lane, wave_u, n_, x0_, y0_, zs, gate_)
    // Multi-tile workgroups request tile k+1 as soon as every wave has finished READING tile k (after
    // the last pass's MFMA loop), so the transfer runs under that pass's stores: the epilogue does
    // not touch the LDS tile.  (Timeline of the gated post conv before: 3.4 us tile wait, 0.9 us
    // MFMAs, 1.4 us stores per tile.)  The first tile is requested here.
    constexpr bool PREFETCH = plc_conv_multi_tile<STAGE, EPI>();
    static_assert(PERSIST == 0 || (IS_LSTM && !PREFETCH), "PERSIST is for the ConvLSTM cells");
    int zs = a.zero_state;      // (PERSIST: cleared after the first step)
    if constexpr (PREFETCH) {
        int n_, x0_, y0_;
        plc_decode_tile(a, TH, plc_first_tile(a, block_x), n_, x0_, y0_);
        PLC_DMA_ISSUE(n_, x0_, y0_, true);
    }
    // every wave runs this once per tile (it contains a workgroup barrier)
#define PLC_PREFETCH_NEXT()                                                                             \
    do {                                                                                                \
        if constexpr (PREFETCH) {                                                                       \
            const int next_ = plc_next_tile(a, tile_id);                                                \
            if (next_ < a.n_tiles) {                                                                    \
                plc_lds_barrier(); /* every wave is done reading the current tile */                    \
                int pn_, px0_, py0_;                                                                    \
                plc_decode_tile(a, TH, next_, pn_, px0_, py0_);                                         \
                /* (time-innermost order: the multiplier tile is fetched at the head of a chain only) */ \
                PLC_DMA_ISSUE(pn_, px0_, py0_, a.t_inner == 0 || next_ % a.t_inner == 0);               \
            }                                                                                           \
        }                                                                                               \
    } while (0)
#pragma unroll 1
    for (int tile_id = plc_first_tile(a, block_x); tile_id < a.n_tiles; tile_id = plc_next_tile(a, tile_id)) {
    if constexpr (PERSIST != 0) {
        const int step = (tile_id - block_x) / a.n_blocks;
        zs = step == 0 ? a.zero_state : 0;
        if constexpr (PERSIST == 2) plc_grid_barrier(a.sync, (unsigned)(step + 1) * gridDim.x);
    }
    int n, x0, y0;
    plc_decode_tile(a, TH, tile_id, n, x0, y0);

    PLC_STAMP(1);
    // ---- stage the input halo tile into LDS ----------------------------------------------
    {
        const char* s0 = plc_img(a.src0, n, a.n_in);
        const char* s1 = NCB1 > 0 ? plc_img(a.src1, n, a.n_in) : nullptr;
        const char* gt = STAGE == STAGE_CB8_GATED ? plc_img(a.gate, n, a.n_in) : nullptr;
        if constexpr (STAGE == STAGE_CB8 || STAGE == STAGE_CB8_GATED) {
            // (multi-tile kernels: this tile's requests were issued before the loop / during the
            // previous tile's last pass).  Gated staging: the product (rounded to bf16 once)
            // replaces the tile in a pass over LDS.
            int n_late = 0;
            if constexpr (!PREFETCH) n_late = PLC_DMA_ISSUE(n, x0, y0, true);
            if constexpr (SPLIT_STAGE) plc_wait_vmcnt_le(n_late);
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (STAGE == STAGE_CB8_GATED) {
                // a wave multiplies the pieces it fetched itself, so no barrier is needed in between
                const uint4* gtile = tile + TILE_UNITS + RED_UNITS + BIAS_UNITS;
                for (int ch = wave_u; ch < NCHUNK; ch += 4) {
                    const int u = ch * 64 + lane;
                    const uint4 f = tile[u], g = gtile[u];
                    const float4 f0 = unpack_bf4(make_uint2(f.x, f.y)), f1 = unpack_bf4(make_uint2(f.z, f.w));
                    const float4 g0 = unpack_bf4(make_uint2(g.x, g.y)), g1 = unpack_bf4(make_uint2(g.z, g.w));
                    const uint2 lo = pack_bf4(f0.x * g0.x, f0.y * g0.y, f0.z * g0.z, f0.w * g0.w);
                    const uint2 hi = pack_bf4(f1.x * g1.x, f1.y * g1.y, f1.z * g1.z, f1.w * g1.w);
                    tile[u] = make_uint4(lo.x, lo.y, hi.x, hi.y);
                }
            }
        } else {
            // register staging (the unit needs arithmetic): four independent units in flight
            constexpr int NIT = (NCB * PLANE + 255) / 256;
            for (int i0 = 0; i0 < NIT; i0 += 4) {
                uint4 v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int u = tid + (i0 + k) * 256;
                    const int cb = u / PLANE;
                    const int r = u - cb * PLANE;
                    const int ly = r / LW;
                    const int lx = r - ly * LW;
                    // (units past the tile get row -1, i.e. the zero unit, so the loads stay unconditional)
                    v[k] = plc_stage_unit<STAGE, NCB0>(a, s0, s1, gt, cb, u < NCB * PLANE ? y0 - PAD + ly : -1,
                                                       x0 - PAD + lx, H, W);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int u = tid + (i0 + k) * 256;
                    if (u < NCB * PLANE) tile[u] = v[k];
                }
            }
        }
    }
    PLC_STAMP(2);
    if constexpr (SPLIT_STAGE) {
        // Bare barrier: __syncthreads() carries a workgroup fence, for which the compiler drains
        // every outstanding vector-memory operation, i.e. also the late pieces still in flight.
        // What must be ordered here is LDS only: this wave's early pieces have landed (vmcnt wait
        // above) and its bias write has retired (lgkmcnt); the clobbers keep the compiler from
        // moving LDS accesses across.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    } else {
        __syncthreads();
    }
    PLC_STAMP(3);

    const char* lds = (const char*)tile;
    if constexpr (PIPE) {
        if (y0 + TH <= H && x0 + 32 <= W) {
            // Whole tile inside the image: branch-free, and the epilogue of pass p-1 (sigma/tanh/
            // state update: ~5 VALU instructions per MFMA, two of them transcendental per ~3 MFMAs)
            // is issued in the gaps of pass p's MFMAs of the SAME wave.  An MFMA occupies the
            // SIMD's issue port for 8 of its 32 cycles, so this VALU work is close to free, whereas
            // two co-resident waves were measured not to overlap MFMA and VALU phases by themselves.
            constexpr int NCBP = NCBW / 2, ROWS = NREP + KS - 1, NF = ROWS * KS * NCBP;
            // The epilogue is cut into micro-ops (one activation of one (row, channel) element, a
            // row's stores, the re-initialisation of the consumed accumulators); micro-ops are dealt
            // out evenly over the pass's MFMAs and sched_barrier pins each one behind its MFMA.
            // Two accumulator sets alternate between "being accumulated" and "being finished".
            constexpr int DEPTH = NF < 4 ? NF : 4;
            constexpr int NMF = NREP * KS * KS * NCBP;          // MFMAs per pass
            constexpr bool SAVE = (EPI == EPI_LSTM_SAVE);
            constexpr int OPS_ROW = 21;                         // 4 channels x 5 activations + the row's stores
            constexpr int NOPS = NREP * OPS_ROW + 1;            // + accumulator re-initialisation
            // wave-uniform bases (scalar registers) + one 32-bit per-lane offset
            const long long hwl = (long long)H * W;
            const int mt_u = __builtin_amdgcn_readfirstlane(mt);
            const int rg_u = __builtin_amdgcn_readfirstlane(rg);
            const char* cin_base = plc_img(a.c_in, n, a.n_in) + (mt_u * hwl) * 32;
            char* cout_base = plc_img(a.out1, n, a.n_in) + (mt_u * hwl) * 32;
            char* hout_base = plc_img(a.out0, n, a.n_in) + (mt_u * hwl) * 16;
            char* gate_base = SAVE ? plc_img(a.out2, n, a.n_in) + (mt_u * hwl) * 16 : nullptr;
            const long long gate_stride = (long long)n_mt * hwl * 16;
            const unsigned lane_c = (unsigned)(x0 + px) * 32u + (unsigned)half * 16u;   // fp32 state: 32 B / pixel
            const unsigned lane_h = lane_c >> 1;                                          // bf16: 16 B / pixel
            const float4* bias_l = (const float4*)(tile + TILE_UNITS + RED_UNITS) + m * 8 + half;
            auto init_acc = [&](f32x16_t (&acc)[NREP]) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 b = bias_l[2 * g];
#pragma unroll
                    for (int r = 0; r < NREP; ++r) {
                        acc[r][4 * g + 0] = b.x;
                        acc[r][4 * g + 1] = b.y;
                        acc[r][4 * g + 2] = b.z;
                        acc[r][4 * g + 3] = b.w;
                    }
                }
            };
            // The same ahead of pass 0 of a two-phase staged tile: asm reads, for the reason given
            // at that pass's fragment reads.
            auto init_acc_early = [&](f32x16_t (&acc)[NREP]) {
                typedef float f32x4_t __attribute__((ext_vector_type(4)));
                f32x4_t b[4];
                const unsigned b32 = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)bias_l;
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:32\n\t"
                             "ds_read_b128 %2, %4 offset:64\n\tds_read_b128 %3, %4 offset:96\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3]) : "v"(b32) : "memory");
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int r = 0; r < NREP; ++r) {
                        acc[r][4 * g + 0] = b[g].x;
                        acc[r][4 * g + 1] = b[g].y;
                        acc[r][4 * g + 2] = b[g].z;
                        acc[r][4 * g + 3] = b[g].w;
                    }
            };
            auto run = [&](auto do_mma, auto do_epi, f32x16_t (&acc)[NREP], float4 (&cnew)[NREP],
                           f32x16_t (&prv)[NREP], float4 (&cold)[NREP], int pass) {
                constexpr bool DO_MMA = decltype(do_mma)::value, DO_EPI = decltype(do_epi)::value;
                const int row0 = rg_u * RPW + pass * NREP;     // rows accumulated in this call
                const int prow0 = row0 - NREP;                  // rows finished in this call
                float ti = 0.f, tig = 0.f, to = 0.f, he = 0.f, cn[4] = {0.f, 0.f, 0.f, 0.f}, ge[4] = {0.f, 0.f, 0.f, 0.f};
                unsigned hp[2] = {0u, 0u}, gpk[4] = {0u, 0u, 0u, 0u};
                // gate values leave as one 8-byte store per gate and lane (4 channels of bf16): a wave's
                // store covers 1 KiB of consecutive bytes (4-byte stores per channel pair were twice
                // the instructions, each touching every other dword of the same lines)
                auto rec = [&](auto rc, auto gc, auto jc, float v) {
                    if constexpr (SAVE) {
                        constexpr int r = decltype(rc)::value, g = decltype(gc)::value, j = decltype(jc)::value;
                        if constexpr (j == 0 || j == 2) ge[g] = v;
                        else if constexpr (j == 1) gpk[g] = pack_bf2(ge[g], v);
                        else
                            *(uint2*)(gate_base + g * gate_stride + (long long)(y0 + prow0 + r) * W * 16 + lane_h) =
                                make_uint2(gpk[g], pack_bf2(ge[g], v));
                    }
                };
                // Every activation is sigma(z) = 1/(1 + 2^(-z log2 e)) or tanh(z) = 1 - 2/(1 + 2^(2 z log2 e)):
                // stage A issues the exp2, stage B (one slot later) the rcp and the op's arithmetic, so
                // the two dependent transcendentals of one op never sit back to back in the stream.
                float epend = 0.f;
                auto stage_a = [&](auto uc) {
                    constexpr int u = decltype(uc)::value;
                    if constexpr (u < NOPS - 1 && u % OPS_ROW < 20) {
                        constexpr int r = u / OPS_ROW, j = (u % OPS_ROW) / 5, act = (u % OPS_ROW) % 5;
                        float z;
                        if constexpr (act == 0) z = prv[r][j];
                        else if constexpr (act == 1) z = prv[r][12 + j];
                        else if constexpr (act == 2) z = prv[r][4 + j];
                        else if constexpr (act == 3) z = prv[r][8 + j];
                        else z = cn[j];
                        constexpr float scale = (act == 1 || act == 4) ? 2.88539008177792681f : -1.44269504088896341f;
                        epend = __builtin_amdgcn_exp2f(scale * z);
                    }
                };
                auto stage_b = [&](auto uc) {
                    constexpr int u = decltype(uc)::value;
                    if constexpr (u == NOPS - 1) {
                        if constexpr (DO_MMA) init_acc(prv);    // prv is the next call's accumulator
                    } else if constexpr (u % OPS_ROW < 20) {
                        constexpr int r = u / OPS_ROW, j = (u % OPS_ROW) / 5, act = (u % OPS_ROW) % 5;
                        using J = std::integral_constant<int, j>;
                        using R = std::integral_constant<int, r>;
                        const float rr = __builtin_amdgcn_rcpf(1.0f + epend);
                        if constexpr (act == 0) {
                            ti = rr;
                            rec(R{}, std::integral_constant<int, 0>{}, J{}, ti);
                        } else if constexpr (act == 1) {
                            // explicit fma: every instantiation must round identically (a batch of 4
                            // and four batches of 1 take different tile heights and must agree bitwise)
                            const float g = __builtin_fmaf(-2.0f, rr, 1.0f);
                            tig = ti * g;
                            rec(R{}, std::integral_constant<int, 3>{}, J{}, g);
                        } else if constexpr (act == 2) {
                            const float cj = j == 0 ? cold[r].x : j == 1 ? cold[r].y : j == 2 ? cold[r].z : cold[r].w;
                            cn[j] = __builtin_fmaf(rr, cj, tig);
                            rec(R{}, std::integral_constant<int, 1>{}, J{}, rr);
                        } else if constexpr (act == 3) {
                            to = rr;
                            rec(R{}, std::integral_constant<int, 2>{}, J{}, to);
                        } else {
                            const float h = to * __builtin_fmaf(-2.0f, rr, 1.0f);
                            if constexpr (j & 1) hp[j >> 1] = pack_bf2(he, h);
                            else he = h;
                        }
                    } else {
                        constexpr int r = u / OPS_ROW;
                        const long long rowoff = (long long)(y0 + prow0 + r) * W;     // uniform
                        *(float4*)(cout_base + rowoff * 32 + lane_c) = make_float4(cn[0], cn[1], cn[2], cn[3]);
                        *(uint2*)(hout_base + rowoff * 16 + lane_h) = make_uint2(hp[0], hp[1]);
                    }
                };
                constexpr int SLOTS = NOPS + 1;
                auto micro = [&](auto uc) {   // slot u: finish op u-1, start op u
                    constexpr int u = decltype(uc)::value;
                    if constexpr (u > 0) stage_b(std::integral_constant<int, u - 1>{});
                    if constexpr (u < NOPS) stage_a(std::integral_constant<int, u>{});
                };
                if constexpr (DO_MMA) {
#pragma unroll
                    for (int r = 0; r < NREP; ++r)   // the POINTER is selected, so the load is unconditional and
                                                     // nothing waits for it before its first use one pass later
                        cnew[r] = plc_gload_f4(zs ? (const char*)&plc_zero16
                                                            : cin_base + (long long)(y0 + row0 + r) * W * 32 + lane_c);
                    bf16x8_t fr[NF];
                    const char* fbase = lds + row0 * LW * 16 + (px * 16 + half * PLANE * 16);
                    // Pass 0 of a two-phase staged tile runs while the late pieces of the tile are
                    // still landing.  The compiler waits vmcnt(0) before any LDS read it can see
                    // while an LDS-DMA write is in flight (it cannot tell the rows apart), so this
                    // pass issues its fragment reads as inline asm and places the lgkmcnt waits
                    // itself: LDS reads return in order, so before the MFMAs on fragment f at most
                    // the younger reads (f+1 .. f+DEPTH) may be outstanding.  The wait names fr[f]
                    // as an in/out operand, which keeps the MFMAs that consume it behind the wait.
                    constexpr bool ASM_READS = SPLIT_STAGE && !DO_EPI;
                    const unsigned fbase32 = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)fbase;
                    auto frag = [&](auto fc) {
                        constexpr int f = decltype(fc)::value;
                        constexpr int rho = f / (KS * NCBP), cbp = (f % (KS * NCBP)) / KS, dx = f % KS;
                        constexpr int off = ((rho * LW + dx) + cbp * 2 * PLANE) * 16;
                        bf16x8_t& dst = fr[f];              // (asm operands alone do not capture in a generic lambda)
                        const unsigned base32 = fbase32;
                        if constexpr (ASM_READS) {
                            static_assert(off < 65536, "DS instruction offsets are 16 bits");
                            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(base32), "n"(off) : "memory");
                        } else {
                            dst = *(const bf16x8_t*)(fbase + off);
                        }
                    };
                    plc_static_for<DEPTH>(frag);
                    plc_static_for<NF>([&](auto fc) {
                        constexpr int f = decltype(fc)::value;
                        if constexpr (f + DEPTH < NF) frag(std::integral_constant<int, f + DEPTH>{});
                        if constexpr (ASM_READS) {
                            constexpr int younger = (NF - 1 - f) < DEPTH ? (NF - 1 - f) : DEPTH;
                            bf16x8_t& cur = fr[f];
                            asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(cur) : "n"(younger) : "memory");
                        }
                        constexpr int rho = f / (KS * NCBP), cbp = (f % (KS * NCBP)) / KS, dx = f % KS;
                        plc_static_for<NREP>([&](auto rc) {
                            constexpr int r = decltype(rc)::value, dy = rho - r;
                            if constexpr (dy >= 0 && dy < KS) {
                                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[(dy * KS + dx) * NCBP + cbp], fr[f],
                                                                                 acc[r], 0, 0, 0);
                                if constexpr (DO_EPI) {
                                    constexpr int mi = plc_mfma_index(f, r, KS, NCBP, NREP);
                                    constexpr int u0 = mi * SLOTS / NMF, u1 = (mi + 1) * SLOTS / NMF;
                                    plc_static_for<u1 - u0>([&](auto kc) {
                                        micro(std::integral_constant<int, u0 + decltype(kc)::value>{});
                                    });
                                }
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        });
                    });
                } else {
                    plc_static_for<SLOTS>(micro);
                }
            };
            f32x16_t accA[NREP], accB[NREP];
            float4 cA[NREP], cB[NREP];
            if constexpr (SPLIT_STAGE) {
                init_acc_early(accA);
                init_acc_early(accB);
            } else {
                init_acc(accA);
                init_acc(accB);
            }
            run(std::true_type{}, std::false_type{}, accA, cA, accB, cB, 0);
            if constexpr (SPLIT_STAGE) {       // the rest of the tile, requested before pass 0, has landed
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            PLC_STAMP(4);
#pragma unroll 1
            for (int pass = 1; pass < PASSES; pass += 2) {
                run(std::true_type{}, std::true_type{}, accB, cB, accA, cA, pass);
                if (pass + 1 < PASSES) run(std::true_type{}, std::true_type{}, accA, cA, accB, cB, pass + 1);
            }
            PLC_STAMP(5);
            if constexpr (PASSES % 2 == 0) run(std::false_type{}, std::true_type{}, accA, cA, accB, cB, PASSES);
            else run(std::false_type{}, std::true_type{}, accB, cB, accA, cA, PASSES);
            PLC_STAMP(6);
#ifdef PLC_TIMELINE
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            PLC_STAMP(7);
#endif
            if constexpr (PERSIST != 0) goto tile_done;
            else return;   // (the LSTM convs take one tile per workgroup)
        }
    }

    if constexpr (PIPE) load_bias();   // (only partly filled tiles of the LSTM convs get here)
    if constexpr (SPLIT_STAGE) {       // the generic loop reads the whole tile
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    PLC_STAMP(4);
#pragma unroll 1
    for (int pass = 0; pass < PASSES; ++pass) {
        const int row0 = rg * RPW + pass * NREP;
        // the epilogue's global operand (previous cell state) is requested now so that its
        // latency is covered by the MFMA loop
        float4 cpre[NREP];
#pragma unroll
        for (int r = 0; r < NREP; ++r) {
            cpre[r] = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (IS_LSTM) {
                const int gy = y0 + row0 + r, gx = x0 + px;
                if (gy < H && gx < W && !zs)
                    cpre[r] = *(const float4*)(plc_img(a.c_in, n, a.n_in) +
                                               ((long long)mt * H * W + (long long)gy * W + gx) * 32 + half * 16);
            }
        }
        f32x16_t acc[NREP];
        {
#pragma unroll
            for (int r = 0; r < NREP; ++r)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    acc[r][4 * g + 0] = bv[g].x;
                    acc[r][4 * g + 1] = bv[g].y;
                    acc[r][4 * g + 2] = bv[g].z;
                    acc[r][4 * g + 3] = bv[g].w;
                }
        }
        // per-lane LDS byte address of this pass's first row at tap (0,0), first channel block of this wave
        int abase[NREP];
#pragma unroll
        for (int r = 0; r < NREP; ++r) {
            abase[r] = ((row0 + r) * LW + px) * 16 + ks * NCBW * PLANE * 16;
            if (NCBW % 2 == 0) abase[r] += half * PLANE * 16;
        }
        if constexpr (NCBW % 2 == 0) {
            // Fragment-centric, software-pipelined K loop.  The B fragment at LDS row rho, column
            // shift dx, channel-block pair cbp feeds output row r with tap row dy = rho - r, so
            // each fragment is read once and used by up to KS of the NREP accumulators.  Reads
            // run DEPTH fragments ahead of the MFMAs that consume them (the LDS latency is several
            // MFMA issue slots); sched_barrier pins that order, the compiler inserts the counted
            // lgkmcnt waits.
            constexpr int NCBP = NCBW / 2, ROWS = NREP + KS - 1, NF = ROWS * KS * NCBP;
            constexpr int DEPTH = NF < 6 ? NF : 6;
            bf16x8_t fr[NF];
            const char* fbase = lds + abase[0];
#pragma unroll
            for (int f = 0; f < DEPTH; ++f) {
                const int rho = f / (KS * NCBP), cbp = (f % (KS * NCBP)) / KS, dx = f % KS;
                fr[f] = *(const bf16x8_t*)(fbase + ((rho * LW + dx) + cbp * 2 * PLANE) * 16);
            }
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                if (f + DEPTH < NF) {
                    const int g = f + DEPTH;
                    const int rho = g / (KS * NCBP), cbp = (g % (KS * NCBP)) / KS, dx = g % KS;
                    fr[g] = *(const bf16x8_t*)(fbase + ((rho * LW + dx) + cbp * 2 * PLANE) * 16);
                }
                const int rho = f / (KS * NCBP), cbp = (f % (KS * NCBP)) / KS, dx = f % KS;
#pragma unroll
                for (int r = 0; r < NREP; ++r) {
                    const int dy = rho - r;
                    if (dy >= 0 && dy < KS)
                        acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[(dy * KS + dx) * NCBP + cbp], fr[f], acc[r], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
            // unit u = tap * NCBW + cb; lanes 0-31 take unit 2s, lanes 32-63 unit 2s+1
            const int u0 = 2 * s, u1 = 2 * s + 1;
            const int t0 = u0 / NCBW, c0 = u0 % NCBW;
            const int t1 = (u1 < UNITS) ? u1 / NCBW : 0, c1 = (u1 < UNITS) ? u1 % NCBW : 0;
            const int off0 = (c0 * PLANE + (t0 / KS) * LW + (t0 % KS)) * 16;
            const int off1 = (c1 * PLANE + (t1 / KS) * LW + (t1 % KS)) * 16;
#pragma unroll
            for (int r = 0; r < NREP; ++r) {
                int addr;
                if (NCBW % 2 == 0) addr = abase[r] + off0;         // off1 == off0 + PLANE*16 folded in abase
                else addr = abase[r] + (half ? off1 : off0);
                const bf16x8_t bfrag = *(const bf16x8_t*)(lds + addr);
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s], bfrag, acc[r], 0, 0, 0);
            }
        }

        PLC_STAMP(5);   // (last pass of the tile: MFMA loop done)
        const bool last_pass = pass == PASSES - 1;
        if constexpr (KSPLIT == 1) {
            if (last_pass) PLC_PREFETCH_NEXT();         // next tile's DMA runs under this pass's stores
        }
        if constexpr (KSPLIT == 2) {
            // exchange partial sums: wave ks finishes row ks of the pass, its partner the other row
            float4* red = (float4*)(tile + TILE_UNITS);
            float4* mine = red + wave * 256 + lane;
            const float4* theirs = red + (wave ^ NMT) * 256 + lane;
            if (ks == 0) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    mine[g * 64] = make_float4(acc[1][4 * g], acc[1][4 * g + 1], acc[1][4 * g + 2], acc[1][4 * g + 3]);
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    mine[g * 64] = make_float4(acc[0][4 * g], acc[0][4 * g + 1], acc[0][4 * g + 2], acc[0][4 * g + 3]);
            }
            plc_lds_barrier();
            if (ks == 0) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 t = theirs[g * 64];
                    acc[0][4 * g] += t.x; acc[0][4 * g + 1] += t.y; acc[0][4 * g + 2] += t.z; acc[0][4 * g + 3] += t.w;
                }
                if (last_pass) PLC_PREFETCH_NEXT();      // (after the exchange's LDS reads, before the stores)
                plc_epilogue<EPI>(a, acc[0], n, mt, n_mt, y0 + row0, x0 + px, half, cpre[0]);
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 t = theirs[g * 64];
                    acc[NREP - 1][4 * g] += t.x; acc[NREP - 1][4 * g + 1] += t.y;
                    acc[NREP - 1][4 * g + 2] += t.z; acc[NREP - 1][4 * g + 3] += t.w;
                }
                if (last_pass) PLC_PREFETCH_NEXT();
                plc_epilogue<EPI>(a, acc[NREP - 1], n, mt, n_mt, y0 + row0 + 1, x0 + px, half, cpre[NREP - 1]);
            }
            plc_lds_barrier();   // the exchange buffer is reused by the next pass (the stores above need not have retired)
        } else {
#pragma unroll
            for (int r = 0; r < NREP; ++r) {
                plc_epilogue<EPI>(a, acc[r], n, mt, n_mt, y0 + row0 + r, x0 + px, half, cpre[r]);
                __builtin_amdgcn_sched_barrier(0);  // keep the NREP epilogues from interleaving (register pressure)
            }
        }
    }
    PLC_STAMP(6);
#ifdef PLC_TIMELINE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PLC_STAMP(7);
#endif
tile_done:
    if constexpr (PERSIST == 1) plc_grid_barrier(a.sync, (unsigned)((tile_id - block_x) / a.n_blocks + 1) * gridDim.x);
    if constexpr (!plc_conv_multi_tile<STAGE, EPI>() && PERSIST == 0) break;
    }   // tile loop
#undef PLC_PREFETCH_NEXT
#undef PLC_DMA_ISSUE
}

template <int NCB0, int NCB1, int NMT, int KS, int STAGE, int EPI, int TH, int KSPLIT = 1>
__global__ __launch_bounds__(256, (plc_conv_min_waves<NCB0 + NCB1, KS, KSPLIT, NMT, EPI>()))
void plc_conv_mfma_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint4 plc_conv_tile[];
    plc_conv_body<NCB0, NCB1, NMT, KS, STAGE, EPI, TH, KSPLIT>(a, blockIdx.x, blockIdx.y, gridDim.y, plc_conv_tile);
}

