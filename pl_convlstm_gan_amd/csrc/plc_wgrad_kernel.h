// Weight gradient of a KSxKS conv on gfx950 matrix cores.
//
//   dW[o][ci][tap] = sum over images and pixels of  dz[o][p] * in[ci][p + tap],   db[o] = sum dz[o][p]
//
// GEMM view: D[row = output channel o][col = (tap, input channel)] with the contraction over
// PIXELS.  The activations live in the CB8 layout (8 channels of one pixel per 16-byte unit),
// i.e. the contraction index is the strided one, so both MFMA operands are fetched from LDS with
// ds_read_b64_tr_b16 (the transposing LDS read): a 16-lane group reads a 4-pixel x 16-channel
// block and every lane receives 4 consecutive pixels of its own channel.
//
//   * one persistent workgroup of 8 waves per CU walks 4 x 32 pixel tiles; each tile's dz block
//     and (4+2) x 34 input halo block are double-buffered in LDS (next tile's global loads are in
//     flight while the current tile is contracted);
//   * the whole dW of the conv stays in registers for the kernel's lifetime: wave (m, g) owns the
//     32-row output tile m and NT_W 32-column tiles (f32x16 accumulators);
//   * a column group is 16 input channels at one tap; one extra group whose B operand is the
//     constant 1.0 yields the bias gradient from the same MFMAs;
//   * every workgroup writes its partial dW to its own slab; plc_wgrad_finalize_kernel sums the
//     slabs in a fixed order (bitwise reproducible, no atomics) and scatters into the PyTorch
//     weight layout.
#pragma once
#include "plc_conv_kernel.h"
#include "plc_misc_kernels.h"

typedef __attribute__((__vector_size__(4 * sizeof(short)))) short s16x4_t;
#define PLC_AS3 __attribute__((address_space(3)))

struct WgradArgs {
    ConvArgs in;       // H, W, tiles_x, tiles_y, n_in and the staging fields (src0, src1, gate, tabs);
                       // in.g_ncb0 > 0: only that many of the input channel blocks exist (the rest read as
                       // zero: the last chunk of a conv contracted in chunks of NCB blocks);
                       // in.zero_state: 1 = src1 of the first n_in images is the unstored zero state,
                       // 2 = src0 of the first n_in images is
    PlcTensor dz;      // CB8 bf16, ceil(cout/8) channel blocks
    PlcTensor dz_mask; // optional (p may be null): CB8 bf16 like dz; dz counts only where mask > 0 (ReLU backward)
    int mcb_valid;     // channel blocks present in dz
    int rows_valid;    // output channels of the conv: slab rows that are written (and later reduced)
    int n_tiles;       // n_img * tiles_y * tiles_x
    float* slab;       // [gridDim.x][MT*32][NT*32]
};

constexpr int WG_TH = 4;            // rows of a tile (WgradGeom::TH doubles it for the narrow convs)
constexpr int WG_PF_DEPTH = 4;     // B fragments requested ahead of the MFMA that consumes them (6 measured no better)

// eight bf16 1.0: the "keep everything" ReLU mask of launches without one
__device__ const uint4 plc_ones16 = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};



template <int NCB, int MT, int KS>
struct WgradGeom {
    // Rows of a tile.  The narrow convs (one 32-row block of outputs, at most two input channel
    // blocks) do few MFMAs per pixel, so their time is the per-tile cost of loads and barriers: 8 rows
    // halve it and two workgroups still fit a CU (stem: 88 -> 66 us; 16 rows: 74 us; the 4-block
    // gated conv with 8 rows fits one workgroup only and was slower, 132 -> 140 us).
    static constexpr int TH = (MT == 1 && NCB <= 2) ? 2 * WG_TH : WG_TH;
    static constexpr int TAPS = KS * KS;
    static constexpr int PAD = KS / 2;
    static constexpr int LH = TH + 2 * PAD;
    static constexpr int LW = 32 + 2 * PAD;
    static constexpr int PLANE = LH * LW;
    static constexpr int MCB = 4 * MT;
    static constexpr int NCBP = (NCB + 1) / 2;
    static constexpr int NG = TAPS * NCBP + 1;          // + bias group
    static constexpr int NT = (NG + 1) / 2;
    static constexpr int WN = 8 / MT;                   // N groups of waves
    static constexpr int NT_W = (NT + WN - 1) / WN;
    static constexpr int IN_UNITS = NCB * PLANE;
    static constexpr int DZ_UNITS = MCB * TH * 32;   // dz units of a tile (what the loader fetches)
    // In LDS each dz channel block is padded by 4 units: the four blocks a 32-lane group reads for
    // an A fragment then start 64 B apart mod 256 and tile the bank row; at the natural stride
    // (2048 B = 0 mod 256) they collide 4-way, which made up most of the kernel's bank conflicts.
    static constexpr int DZ_CB = TH * 32 + 4;
    static constexpr int DZ_LDS_UNITS = MCB * DZ_CB;
    static constexpr int CONST_UNITS = ((TH - 1) * LW + 21 + 15) / 16 * 16;   // covers every k-step offset of a tile
    static constexpr int CONST_BASE = IN_UNITS + DZ_LDS_UNITS;
    static constexpr int BUF_UNITS = CONST_BASE + 2 * CONST_UNITS;   // [in | dz | zeros | ones]
    static constexpr int LDS_BYTES = 2 * BUF_UNITS * 16;
    static constexpr int NPF = (IN_UNITS + DZ_UNITS + 511) / 512;
};

// Waves per SIMD to compile for: the instantiations whose LDS double buffer lets two workgroups share a
// CU are launched that way (2 x 8 waves = 4 per SIMD), so they must stay within 128 VGPRs; one register
// more and only one workgroup is resident while the host still launches two per CU (cell 1: 137 -> 172 us).
template <int NCB, int MT, int KS>
constexpr int plc_wgrad_min_waves() {
    return WgradGeom<NCB, MT, KS>::LDS_BYTES <= 80 * 1024 ? 4 : 2;
}

template <int NCB0, int NCB1, int MT, int KS, int STAGE>
__global__ __launch_bounds__(512, (plc_wgrad_min_waves<NCB0 + NCB1, MT, KS>())) void plc_wgrad_kernel(const WgradArgs a) {
    using G = WgradGeom<NCB0 + NCB1, MT, KS>;
    constexpr int NCB = NCB0 + NCB1;
    static_assert(MT == 1 || MT == 2 || MT == 4, "MT must be 1, 2 or 4");
    static_assert(((G::TH - 1) * G::LW + 16) * 16 + 80 <= G::CONST_UNITS * 16, "constant region too small");
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int m = wave % MT;
    const int ng = wave / MT;
    const int g16 = lane >> 4;          // 16-lane group: channels 16*(g16&1).., pixel half g16>>1
    const int khalf = g16 >> 1;
    const int li = lane & 15;
    const int q = li >> 2, p = li & 3;  // lane supplies pixel q, channels 4p..4p+3 of the group's 16
    const int H = a.in.H, W = a.in.W;

    // constant regions of both buffers
    for (int i = tid; i < 2 * 2 * G::CONST_UNITS; i += 512) {
        const int buf = i / (2 * G::CONST_UNITS), r = i % (2 * G::CONST_UNITS);
        const unsigned v = r < G::CONST_UNITS ? 0u : 0x3F803F80u;
        smem[buf * G::BUF_UNITS + G::CONST_BASE + r] = make_uint4(v, v, v, v);
    }

    // per-lane byte addresses (buffer 0, k-step 0) of the two transposed reads of each fragment
    const int pix_off = (8 * khalf + q) * 16 + (p & 1) * 8;
    int a_addr;   // A: dz block, channel block 4m + 2*(g16&1) + (p>>1)
    {
        const int cb = 4 * m + 2 * (g16 & 1) + (p >> 1);
        a_addr = (G::IN_UNITS + cb * G::DZ_CB) * 16 + pix_off;
    }
    int b_addr[G::NT_W];
#pragma unroll
    for (int i = 0; i < G::NT_W; ++i) {
        const int nt = ng * G::NT_W + i;
        const int grp = 2 * nt + (g16 & 1);
        int addr;
        if (grp < G::TAPS * G::NCBP) {
            const int tap = grp / G::NCBP, cbp = grp % G::NCBP;
            const int cb = 2 * cbp + (p >> 1);
            if (cb < NCB) addr = ((cb * G::LH + tap / KS) * G::LW + tap % KS) * 16 + pix_off;
            else addr = G::CONST_BASE * 16 + (p & 1) * 8;                                       // zeros
        } else if (grp == G::TAPS * G::NCBP) {
            addr = (G::CONST_BASE + G::CONST_UNITS) * 16 + (p & 1) * 8;                         // ones (bias)
        } else {
            addr = G::CONST_BASE * 16 + (p & 1) * 8;                                            // zeros
        }
        b_addr[i] = addr;
    }

    f32x16_t acc[G::NT_W];
#pragma unroll
    for (int i = 0; i < G::NT_W; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    const int tiles_per_img = a.in.tiles_x * a.in.tiles_y;

    // ---- tile loader: global -> registers, registers -> LDS ---------------------------------
    // Two phases.  load_tile only SELECTS per-lane addresses (a real one, or a 16-byte constant for
    // lanes outside the image) and then loads unconditionally, so no load sits in a divergent
    // branch and all NPF (+ secondary) loads of a tile are in flight together while the current
    // tile is contracted.  store_tile forms the LDS units from the landed registers: the gated
    // product, the coord-conv unit, the ReLU-masked dz.
    static_assert(STAGE == STAGE_CB8 || STAGE == STAGE_CB8_GATED || STAGE == STAGE_COORD, "unsupported staging");
    constexpr bool HAS2 = STAGE != STAGE_CB8;       // secondary 16-byte load: gate (input part), ReLU mask (dz part)
    constexpr int NPF_IN = (G::IN_UNITS + 511) / 512;             // prefetch slots that hold input-part lanes
    constexpr int NPF2 = HAS2 ? G::NPF : 1;
    constexpr int NPFC = STAGE == STAGE_COORD ? NPF_IN : 1;
    uint4 pf[G::NPF], pf2[NPF2];
    float pf_rain[NPFC], pf_row[NPFC], pf_col[NPFC];
    auto load_tile = [&](int t) {
        const int n = t / tiles_per_img;
        const int rem = t - n * tiles_per_img;
        const int ty = rem / a.in.tiles_x, tx = rem - ty * a.in.tiles_x;
        const int x0 = tx * 32, y0 = ty * G::TH;
        const char* s0 = plc_img(a.in.src0, n, a.in.n_in);
        const char* s1 = NCB1 > 0 ? plc_img(a.in.src1, n, a.in.n_in) : nullptr;
        const char* gt = STAGE == STAGE_CB8_GATED ? plc_img(a.in.gate, n, a.in.n_in) : nullptr;
        const char* dzp = plc_img(a.dz, n, a.in.n_in);
        const bool has_mask = HAS2 && a.dz_mask.p != nullptr;
        const char* mkp = has_mask ? plc_img(a.dz_mask, n, a.in.n_in) : nullptr;
        const char* const z16 = (const char*)&plc_zero16;
        const char* const o16 = (const char*)&plc_ones16;
        const long long plane_b = (long long)H * W * 16;
#pragma unroll
        for (int k = 0; k < G::NPF; ++k) {
            const int u = tid + k * 512;
            const char* p1 = z16;
            const char* p2 = o16;
            const char* pr = z16;
            const char* prow = z16;
            const char* pcol = z16;
            if (u < G::IN_UNITS) {
                const int cb = u / G::PLANE;
                const int r = u - cb * G::PLANE;
                const int ly = r / G::LW, lx = r - ly * G::LW;
                const int gy = y0 - G::PAD + ly, gx = x0 - G::PAD + lx;
                const bool ok = (gy >= 0) & (gy < H) & (gx >= 0) & (gx < W) & (a.in.g_ncb0 == 0 || cb < a.in.g_ncb0);
                const long long pix = (long long)gy * W + gx;
                if (STAGE == STAGE_COORD) {
                    pr = ok ? s0 + pix * 4 : z16;
                    prow = ok ? (const char*)(a.in.row_tab + gy) : z16;
                    pcol = ok ? (const char*)(a.in.col_tab + gx) : z16;
                } else {
                    const char* base = (NCB1 == 0 || cb < NCB0) ? s0 + cb * plane_b : s1 + (cb - NCB0) * plane_b;
                    // zero_state: src1 of the first n_in images is the (unstored) zero initial state
                    const bool skip = ((NCB1 > 0 && cb >= NCB0 && a.in.zero_state != 0) || a.in.zero_state == 2) && n < a.in.n_in;
                    p1 = (ok & !skip) ? base + pix * 16 : z16;
                    if (STAGE == STAGE_CB8_GATED) p2 = ok ? gt + cb * plane_b + pix * 16 : z16;
                }
            } else if (u < G::IN_UNITS + G::DZ_UNITS) {
                const int d = u - G::IN_UNITS;
                const int cb = d / (G::TH * 32);
                const int r = d - cb * (G::TH * 32);
                const int gy = y0 + (r >> 5), gx = x0 + (r & 31);
                const bool ok = (cb < a.mcb_valid) & (gy < H) & (gx < W);
                const long long off = (((long long)cb * H + gy) * W + gx) * 16;
                p1 = ok ? dzp + off : z16;
                if (HAS2) p2 = (ok & has_mask) ? mkp + off : o16;
            }
            pf[k] = plc_gload16(p1);
            if constexpr (HAS2) pf2[k] = plc_gload16(p2);
            if constexpr (STAGE == STAGE_COORD) {
                if (k < NPF_IN) {
                    pf_rain[k] = plc_gload4(pr);
                    pf_row[k] = plc_gload4(prow);
                    pf_col[k] = plc_gload4(pcol);
                }
            }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int k = 0; k < G::NPF; ++k) {
            const int u = tid + k * 512;
            uint4 v = pf[k];
            if constexpr (STAGE == STAGE_CB8_GATED) {
                if (u < G::IN_UNITS) {
                    // src0 * gate, rounded to bf16 (the conv's own gated staging forms the same product)
                    const uint4 f = pf[k], g = pf2[k];
                    const float4 f0 = unpack_bf4(make_uint2(f.x, f.y)), f1 = unpack_bf4(make_uint2(f.z, f.w));
                    const float4 g0 = unpack_bf4(make_uint2(g.x, g.y)), g1 = unpack_bf4(make_uint2(g.z, g.w));
                    const uint2 lo = pack_bf4(f0.x * g0.x, f0.y * g0.y, f0.z * g0.z, f0.w * g0.w);
                    const uint2 hi = pack_bf4(f1.x * g1.x, f1.y * g1.y, f1.z * g1.z, f1.w * g1.w);
                    v = make_uint4(lo.x, lo.y, hi.x, hi.y);
                }
            }
            if constexpr (STAGE == STAGE_COORD) {
                if (k < NPF_IN && u < G::IN_UNITS)      // channels (rain, row coordinate, column coordinate, 0 ...)
                    v = make_uint4(pack_bf2(pf_rain[k], pf_row[k]), pack_bf2(pf_col[k], 0.0f), 0u, 0u);
            }
            if constexpr (HAS2) {
                if (u >= G::IN_UNITS) {
                    // dz counts where mask > 0;  bf16 > 0  <=>  sign bit clear and the other 15 bits not all zero
                    auto keep = [](unsigned m) {
                        const unsigned lo = (m & 0x8000u) == 0 && (m & 0x7fffu) != 0 ? 0x0000ffffu : 0u;
                        const unsigned hi = (m & 0x80000000u) == 0 && (m & 0x7fff0000u) != 0 ? 0xffff0000u : 0u;
                        return lo | hi;
                    };
                    const uint4 mk = pf2[k];
                    v.x &= keep(mk.x); v.y &= keep(mk.y); v.z &= keep(mk.z); v.w &= keep(mk.w);
                }
            }
            // input part at its unit index; dz part with each channel block padded to DZ_CB units
            const int dst = u < G::IN_UNITS ? u : u + (u - G::IN_UNITS) / (G::TH * 32) * (G::DZ_CB - G::TH * 32);
            if (u < G::IN_UNITS + G::DZ_UNITS) smem[buf * G::BUF_UNITS + dst] = v;
        }
    };

    int t = blockIdx.x;
    int buf = 0;
    if (t < a.n_tiles) {
        load_tile(t);
        store_tile(0);
    }
    __syncthreads();
    PLC_AS3 char* lds = (PLC_AS3 char*)smem;

    for (; t < a.n_tiles; t += gridDim.x) {
        const int tn = t + gridDim.x;
        if (tn < a.n_tiles) load_tile(tn);            // next tile's loads fly during the MFMAs below
        const int boff = buf * G::BUF_UNITS * 16;
        // The tile's MFMAs as one flat software pipeline: MFMA j = (k-step j / NT_W, column tile
        // j % NT_W).  Its B fragment is requested DEPTH MFMAs earlier (ring of DEPTH + 1 fragments), its
        // A fragment together with the first B of its k-step.  Left to itself the scheduler sinks
        // each transposing read next to its use and every MFMA then waits out an LDS round trip
        // (matrix pipe 44 % busy with two waves per SIMD); sched_barrier pins the order written
        // here, the compiler still derives the exact lgkmcnt waits.
        constexpr int TOTAL = G::TH * 2 * G::NT_W;
        constexpr int DEPTH = TOTAL < WG_PF_DEPTH ? TOTAL : WG_PF_DEPTH;
        constexpr int BR = DEPTH + 1, AR = DEPTH / G::NT_W + 2;
        bf16x8_t bq[BR], aq[AR];
        auto frag = [&](int addr) {
            const s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((PLC_AS3 s16x4_t*)(lds + boff + addr));
            const s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((PLC_AS3 s16x4_t*)(lds + boff + addr + 64));
            return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
        };
        auto issue = [&](int jj) {      // the fragment reads MFMA jj needs (jj is a constant after unrolling)
            const int ks = jj / G::NT_W, i = jj % G::NT_W;
            const int row = ks >> 1, xk = (ks & 1) * 16;
            if (i == 0) aq[ks % AR] = frag(a_addr + (row * 32 + xk) * 16);
            bq[jj % BR] = frag(b_addr[i] + (row * G::LW + xk) * 16);
        };
        // Only where ONE workgroup fits a CU (two waves per SIMD): with two workgroups the four
        // waves per SIMD already cover the LDS latency and the pinned order was measured slower
        // (cell 1: 144 -> 172 us), so those instantiations keep the compiler's own schedule.
        if constexpr (G::LDS_BYTES > 80 * 1024) {
#pragma unroll
            for (int jj = 0; jj < DEPTH; ++jj) issue(jj);
#pragma unroll
            for (int j = 0; j < TOTAL; ++j) {
                if (j + DEPTH < TOTAL) issue(j + DEPTH);
                __builtin_amdgcn_sched_barrier(0);
                acc[j % G::NT_W] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[(j / G::NT_W) % AR], bq[j % BR],
                                                                           acc[j % G::NT_W], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < G::TH * 2; ++ks) {
                const int row = ks >> 1, xk = (ks & 1) * 16;
                const bf16x8_t af = frag(a_addr + (row * 32 + xk) * 16);
#pragma unroll
                for (int i = 0; i < G::NT_W; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, frag(b_addr[i] + (row * G::LW + xk) * 16),
                                                                     acc[i], 0, 0, 0);
            }
        }
        if (tn < a.n_tiles) store_tile(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    // ---- partial dW of this workgroup -> its slab ---------------------------------------------
    float* slab = a.slab + (long long)blockIdx.x * (MT * 32) * (G::NT * 32);
    const int half = lane >> 5, col = lane & 31;
#pragma unroll
    for (int i = 0; i < G::NT_W; ++i) {
        const int nt = ng * G::NT_W + i;
        if (nt >= G::NT) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
            // (rows past the conv's output channels are never read by the slab reduction)
            if (32 * m + row < a.rows_valid) slab[(long long)(32 * m + row) * (G::NT * 32) + nt * 32 + col] = acc[i][r];
        }
    }
}

// Sum the per-workgroup slabs (fixed order) and scatter into dW [cout][c0+c1][ks][ks], db [cout].
struct FinalizeJob {
    const float* slab;
    float* dw;
    float* db;
    int n_slabs, rows_pad, cols_pad, cout, c0, c1, ks, ncb;
    int ci_off, c_total;            // the job's input channels are channels ci_off.. of a conv with c_total of them
    int first_block, n_blocks;      // position inside a multi-job launch
};

__device__ __forceinline__ void plc_wgrad_finalize_body(const FinalizeJob& jb, int block, int n_blocks,
                                                        float (*part)[33]) {
    // 256 threads = 32 consecutive output elements x 8 slab lanes; lane q sums slabs q, q+8, ...
    // and the 8 partial sums are combined in a fixed order through LDS.
    const float* __restrict__ slab = jb.slab;
    float* __restrict__ dw = jb.dw;
    float* __restrict__ db = jb.db;
    const int n_slabs = jb.n_slabs, rows_pad = jb.rows_pad, cols_pad = jb.cols_pad, cout = jb.cout;
    const int c0 = jb.c0, c1 = jb.c1, ks = jb.ks, ncb = jb.ncb;
    const int taps = ks * ks, ncbp = (ncb + 1) / 2;
    const long long total = (long long)cout * cols_pad;
    const int el = threadIdx.x & 31, q = threadIdx.x >> 5;
    for (long long base = (long long)block * 32; base < total; base += (long long)n_blocks * 32) {
        const long long idx = base + el;
        float s = 0.0f;
        if (idx < total) {
            const float* p = slab + idx;   // slab row stride == cols_pad, so (o, ncol) sits at idx
            const long long sstride = (long long)rows_pad * cols_pad;
            for (int b0 = q; b0 < n_slabs; b0 += 64) {      // 8 independent loads in flight, fixed order
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int b = b0 + 8 * k;
                    v[k] = b < n_slabs ? __builtin_nontemporal_load(p + (long long)b * sstride) : 0.0f;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) s += v[k];
            }
        }
        part[q][el] = s;
        __syncthreads();
        if (q == 0 && idx < total) {
            float t = part[0][el];
#pragma unroll
            for (int k = 1; k < 8; ++k) t += part[k][el];
            const int o = idx / cols_pad, ncol = idx % cols_pad;
            const int grp = ncol >> 4, c16 = ncol & 15;
            if (grp < taps * ncbp) {
                const int tap = grp / ncbp;
                const int cb = 2 * (grp % ncbp) + (c16 >> 3);
                const int ci = cb < ncb ? plc_padded_to_concat(cb, c16 & 7, c0, c1) : -1;
                if (ci >= 0) dw[((long long)o * jb.c_total + jb.ci_off + ci) * taps + tap] = t;
            } else if (grp == taps * ncbp && c16 == 0) {
                if (db) db[o] = t;
            }
        }
        __syncthreads();
    }
}

__global__ void plc_wgrad_finalize_kernel(const FinalizeJob jb) {
    __shared__ float part[8][33];
    plc_wgrad_finalize_body(jb, blockIdx.x, gridDim.x, part);
}

// The slabs of every conv of a backward pass, reduced in ONE launch at its end.
constexpr int FINALIZE_MAX_JOBS = 24;
struct FinalizeTable {
    FinalizeJob job[FINALIZE_MAX_JOBS];
    int n;
};
__global__ void plc_wgrad_finalize_multi_kernel(const FinalizeTable tb) {
    __shared__ float part[8][33];
    int j = 0;
    while (j + 1 < tb.n && (int)blockIdx.x >= tb.job[j + 1].first_block) ++j;
    plc_wgrad_finalize_body(tb.job[j], blockIdx.x - tb.job[j].first_block, tb.job[j].n_blocks, part);
}
