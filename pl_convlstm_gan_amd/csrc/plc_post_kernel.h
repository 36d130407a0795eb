// The generator's two output convs in ONE kernel (generator.py:198-202 of the reference):
//
//   out = conv3x3_{32->1}( relu( conv3x3_{C2->32}( feat * gate ) ) )
//
// Run separately, the first conv writes its 32-channel result (64 B per pixel and frame) for the second to read
// back with a halo, which is more than half of the bytes the pair moves.  Here a workgroup owns a 14 x 30 pixel
// tile of one frame: it stages the gated input on the tile's 18 x 34 neighbourhood into LDS, runs the first conv
// on the 16 x 32 pixels the second conv needs, leaves that result (bf16, zero outside the image: the second
// conv's zero padding) in LDS, and runs the second conv from there.  Training also needs the intermediate
// result (ReLU mask and weight gradient of the backward pass): SAVE_P0 stores the tile's own pixels of it.
//
// The first conv runs exactly the instructions of the separate kernel (plc_conv_mfma_kernel<4,0,1,3,GATED,ACT_CB8>):
// the same packed weights, the bias as the accumulator's initial value, v_mfma_f32_32x32x16_bf16 over the k-steps in
// the order (tap row, channel-block pair, tap column), ReLU and round-to-bf16 -- its result, which the backward pass
// reads, is bit-identical to that kernel's.  The second conv has ONE output channel; on the 32-row instruction it
// would take as many matrix cycles and weight registers as the first.  It runs on v_mfma_f32_16x16x32_bf16 instead
// (one tap per k-step: K = 4 channel blocks x 8 channels; two 16-pixel column tiles per row), half the cycles and
// half the registers.  Its fp32 sums are formed in another order than the separate kernel's, so the output agrees
// with the two-launch path to fp32 rounding (1e-6 relative), not bit for bit; training and inference both run this
// kernel and agree bit for bit.
//
// Widths: C2 (as whole blocks) = 32, i.e. 4 input channel blocks; other widths keep the separate kernels.
#pragma once
#include <type_traits>
#include "plc_conv_kernel.h"

struct PostArgs {
    int H, W;                 // output grid
    int n_in;                 // B (images are n = t * B + b)
    int T;                    // frames per batch element
    int tiles_x, tiles_y, n_tiles;
    PlcTensor feat;           // CB8 bf16, 4 channel blocks, per frame
    PlcTensor gate;           // CB8 bf16, 4 channel blocks, per batch element (so = 0)
    PlcTensor p0;             // SAVE_P0: CB8 bf16, 4 channel blocks, per frame
    PlcTensor out;            // fp32 plane per frame
    const uint4* wpk0;        // packed A fragments of the first conv (1 M tile, 18 k-steps)
    const float* bpk0;
    const uint4* wpk2;        // ... of the second conv (1 M tile of which row 0 is real, 18 k-steps), as the separate
    const float* bpk2;        // kernel takes them; the 16-row fragments used here are gathered from it
#ifdef PLC_POST_TIMELINE
    long long* timeline;      // [workgroup][wave][5] cycle sums (instrumented build only)
#endif
};

constexpr int POST_TH = 14, POST_TW = 30;                 // output pixels of a tile
constexpr int POST_PH = POST_TH + 2, POST_PW = 34;        // first conv's result: 16 rows x 32 columns (+2 zero columns)
constexpr int POST_IH = POST_TH + 4, POST_IW = 34;        // gated input: 18 rows x 34 columns
constexpr int POST_IN_UNITS = 4 * POST_IH * POST_IW;      // 2448
constexpr int POST_P0_UNITS = 4 * POST_PH * POST_PW;      // 2176
// LDS: the two groups' gated input tiles, the raw gate tile of the current (batch element, tile position) and ONE
// first-conv result tile (only one group computes at a time)
// + the second conv's weight fragments and the first conv's bias rows, which would otherwise hold 52 registers
// through the first conv's MFMA loop, where the kernel's register pressure peaks
constexpr int POST_W2_UNITS = 9 * 64, POST_B0_UNITS = 8;
constexpr int POST_LDS_BYTES = (3 * POST_IN_UNITS + POST_P0_UNITS + POST_W2_UNITS + POST_B0_UNITS) * 16;

// One pass of a 3x3 conv over 4 channel blocks for output rows r0 and r0 + 1 of an LDS tile laid out
// [channel block][rows][LW columns] (16-byte units): the accumulation order of plc_conv_body's fragment loop.
// side(f) is called once per fragment, behind its MFMAs: a slice of work that is independent of the conv (the
// staging of later tiles) and so issues while the matrix pipe is busy with them.
template <int LROWS, int LW, class Side>
__device__ __forceinline__ void plc_post_pass(const char* lds, int r0, int px, int half, const bf16x8_t (&wf)[18],
                                              f32x16_t (&acc)[2], Side&& side) {
    constexpr int PLANE = LROWS * LW;
    // Fragment f = (LDS row rho, channel-block pair cbp, tap column dx) feeds output row r at tap row rho - r.  Only
    // one group of waves computes at a time (one wave per SIMD), so nothing else hides an LDS round trip: the reads
    // run DEPTH fragments ahead of the MFMAs that consume them and sched_barrier pins that order.
    constexpr int NF = 4 * 2 * 3, DEPTH = 5;
    const char* base = lds + ((r0 * LW + px) + half * PLANE) * 16;
    bf16x8_t fr[NF];
#pragma unroll
    for (int f = 0; f < DEPTH; ++f) {
        const int rho = f / 6, cbp = (f % 6) / 3, dx = f % 3;
        fr[f] = *(const bf16x8_t*)(base + ((rho * LW + dx) + cbp * 2 * PLANE) * 16);
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        if (f + DEPTH < NF) {
            const int g = f + DEPTH;
            const int rho = g / 6, cbp = (g % 6) / 3, dx = g % 3;
            fr[g] = *(const bf16x8_t*)(base + ((rho * LW + dx) + cbp * 2 * PLANE) * 16);
        }
        const int rho = f / 6, cbp = (f % 6) / 3, dx = f % 3;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int dy = rho - r;
            if (dy >= 0 && dy < 3)
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[(dy * 3 + dx) * 2 + cbp], fr[f], acc[r], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        side(f);
        __builtin_amdgcn_sched_barrier(0);
    }
}

typedef float plc_f32x4_t __attribute__((ext_vector_type(4)));

// Two bf16 products, each rounded to bf16 once: the fp32 product of two bf16 values is exact, so this is what a
// scalar multiply-and-round of each pair gives.  One packed multiply for the pair (the staging is bound by the
// number of vector instructions, four cycles each for a 64-lane wave).
__device__ __forceinline__ unsigned plc_mul_bf2(unsigned x, unsigned y) {
    const plc_f32x2_t xf = {__builtin_bit_cast(float, x << 16), __builtin_bit_cast(float, x & 0xffff0000u)};
    const plc_f32x2_t yf = {__builtin_bit_cast(float, y << 16), __builtin_bit_cast(float, y & 0xffff0000u)};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(xf * yf, plc_bf16x2_t));
}

// The one-output-channel conv for output rows r0, r0 + 1 (32 pixel columns as two 16-column tiles) from an LDS tile
// [channel block][rows][LW columns]: lane (q = lane / 16, n = lane % 16) supplies channel block q of pixel column n.
// acc[r][nt][0] of the lanes with q == 0 is the output of row r, column 16 nt + n.
template <int LROWS, int LW, class Side>
__device__ __forceinline__ void plc_post_pass_1ch(const char* lds, int r0, int lane, const bf16x8_t (&wf)[9],
                                                  plc_f32x4_t (&acc)[2][2], Side&& side) {
    constexpr int PLANE = LROWS * LW;
    constexpr int NF = 4 * 3 * 2, DEPTH = 5;            // fragment f = (LDS row rho, tap column dx, column tile nt)
    const int q = lane >> 4, n = lane & 15;
    const char* base = lds + ((r0 * LW + n) + q * PLANE) * 16;
    bf16x8_t fr[NF];
#pragma unroll
    for (int f = 0; f < DEPTH; ++f) {
        const int rho = f / 6, dx = (f % 6) / 2, nt = f % 2;
        fr[f] = *(const bf16x8_t*)(base + (rho * LW + dx + 16 * nt) * 16);
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        if (f + DEPTH < NF) {
            const int g = f + DEPTH;
            const int rho = g / 6, dx = (g % 6) / 2, nt = g % 2;
            fr[g] = *(const bf16x8_t*)(base + (rho * LW + dx + 16 * nt) * 16);
        }
        const int rho = f / 6, dx = (f % 6) / 2, nt = f % 2;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int dy = rho - r;
            if (dy >= 0 && dy < 3)
                acc[r][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[dy * 3 + dx], fr[f], acc[r][nt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        side(f);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// One workgroup of eight waves (two per SIMD, so that one wave's LDS round trips and epilogue run under the other's
// MFMAs) per CU walks a contiguous run of the time-innermost tile list: item i = frame i % T of chain i / T, a chain
// being one tile position of one batch element.  Nothing but a chain's first tile waits for memory:
//   * a thread requests its five units of the tile after next from inside the first conv's MFMA loop, into registers,
//     and forms the gated product into the OTHER input buffer at the start of the next round (double-buffered input
//     tile);
//   * the gate tile of a chain is the same for all T frames: a thread keeps ITS units of it in an LDS tile that only
//     it reads (no barrier), and fetches them anew, synchronously, when a chain begins (once per T tiles);
//   * in training the intermediate's own pixels leave for global memory as whole 16-byte units from inside the second
//     conv's MFMA loop.
// Round k:  product of tile k + 1  |  first conv of tile k (wave w: result rows 2w, 2w + 1) with the request of tile
// k + 2 behind its MFMAs  | barrier |  second conv of tile k (wave w < 7: output rows 2w, 2w + 1)  | barrier.
// Measured per round with the instrumented build (devtools/post_probe.hip, 26.7 tiles per workgroup at 40 frames of
// 256 x 256): product 2.2-3.6 K cycles (the two waves of a SIMD share its vector ALU), first conv 3.1-3.5 K (2.1 K
// without the requests), second conv 1.5-2.1 K, barriers 2 K; 132 us per launch, 158 us in training.
template <bool SAVE_P0>
__global__ __launch_bounds__(512, 1) void plc_post_fused_kernel(const PostArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint4 plc_post_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, px = lane & 31;
    uint4* const tgate = plc_post_lds + 2 * POST_IN_UNITS;         // raw gate units of the current chain (thread-private slots)
    uint4* const tp0 = plc_post_lds + 3 * POST_IN_UNITS;           // first conv's result
    uint4* const tw2 = tp0 + POST_P0_UNITS;                        // second conv's fragments [tap][lane]
    float4* const tb0 = (float4*)(tw2 + POST_W2_UNITS);            // first conv's bias rows
    const int H = a.H, W = a.W;
    const long long plane_b = (long long)H * W * 16;

    // stationary weights of the first conv; the second conv's fragments and the bias rows wait in LDS (they would
    // otherwise hold 52 registers through the first conv's MFMA loop, where the register pressure peaks)
    bf16x8_t wf0[18];
#pragma unroll
    for (int s = 0; s < 18; ++s) wf0[s] = __builtin_bit_cast(bf16x8_t, a.wpk0[s * 64 + lane]);
    // second conv, tap t: lane (q, m) holds the 8 channels of block q of output row m, which the 32-row packing
    // keeps as unit u = 4 t + q (k-step u / 2, lane half u % 2) of row m
    if (tid < 64) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int u = 4 * t + (lane >> 4);
            tw2[t * 64 + lane] = a.wpk2[(u >> 1) * 64 + (u & 1) * 32 + (lane & 15)];
        }
    } else if (tid < 64 + POST_B0_UNITS) {
        tb0[tid - 64] = ((const float4*)a.bpk0)[tid - 64];
    }
    const float bias2 = a.bpk2[0];                        // (row 0 of the packed bias: the one output channel's)
    // the two columns past the first conv's 32 result columns only feed output columns that are discarded
    for (int u = tid; u < 4 * POST_PH * 2; u += 512) {
        const int cb = u / (POST_PH * 2), r = (u / 2) % POST_PH, c = 32 + (u & 1);
        tp0[(cb * POST_PH + r) * POST_PW + c] = make_uint4(0, 0, 0, 0);
    }
    // The workgroup's items are i0 .. i0 + my_tiles - 1 of the time-innermost list.
    const long long i0 = (long long)a.n_tiles * blockIdx.x / gridDim.x;
    const int my_tiles = (int)((long long)a.n_tiles * (blockIdx.x + 1) / gridDim.x - i0);
    // Positions of the items k, k + 1 and k + 2 of the current round (wave-uniform), advanced by one item per round:
    // no division inside the loop.
    struct Item { int t, b, tx, ty; };
    auto advance = [&](Item it) {
        if (++it.t == a.T) {
            it.t = 0;
            if (++it.tx == a.tiles_x) {
                it.tx = 0;
                if (++it.ty == a.tiles_y) { it.ty = 0; ++it.b; }
            }
        }
        return it;
    };
    Item it0, it1, it2;
    {
        const int per_img = a.tiles_x * a.tiles_y;
        const int chain = (int)(i0 / a.T), tile = chain % per_img;
        it0.t = (int)(i0 - (long long)chain * a.T);
        it0.b = chain / per_img;
        it0.ty = tile / a.tiles_x;
        it0.tx = tile - it0.ty * a.tiles_x;
        it1 = advance(it0);
        it2 = advance(it1);
    }
    auto coords = [&](const Item& it, int& n, int& x0, int& y0) {
        n = it.t * a.n_in + it.b;
        x0 = it.tx * POST_TW;
        y0 = it.ty * POST_TH;
    };
    // The 18 x 34 input neighbourhood of a tile (rows y0-2 .. y0+15, columns x0-2 .. x0+31): thread t takes units
    // t + 512 q, q < 5.  A load is a scalar base plus a 32-bit lane offset (the launcher checks that four channel
    // planes fit 32 bits); the offset of a unit from the neighbourhood's first unit is the one thing kept per unit
    // (five registers).  A unit outside the image, or past the tile's last unit, loads the tile's own first pixel
    // instead, which always exists, and is zeroed when it is consumed.  Only tiles that touch the image border test
    // bounds per lane, from coordinates recomputed on the spot behind an operand the compiler cannot see through
    // (`opaque_zero`): hoisted, they would be a dozen more registers living across the first conv's MFMA loop, where
    // the allocator is at its limit, and a spill reload between two loads makes the second wait for the first
    // (scratch accesses count on vmcnt).
    constexpr int NIT = (POST_IN_UNITS + 511) / 512;      // 5
    auto tile_inner = [&](int x0, int y0) { return y0 >= 2 && y0 + POST_TH + 2 <= H && x0 >= 2 && x0 + POST_TW + 2 <= W; };
    // does unit u of the neighbourhood of the tile at (x0, y0) exist?
    auto unit_ok = [&](int u, int x0, int y0) {
        const int cb = u / (POST_IH * POST_IW), r = u - cb * (POST_IH * POST_IW);
        const int ly = r / POST_IW, lx = r - ly * POST_IW;
        const int gy = y0 - 2 + ly, gx = x0 - 2 + lx;
        return (u < POST_IN_UNITS) & (gy >= 0) & (gy < H) & (gx >= 0) & (gx < W);
    };
    const unsigned safe = (unsigned)(2 * W + 2) * 16u;        // the tile's own first pixel
    unsigned off[NIT];
#pragma unroll
    for (int q = 0; q < NIT; ++q) {
        const int u = tid + q * 512;
        const int cb = u / (POST_IH * POST_IW), r = u - cb * (POST_IH * POST_IW);
        const int ly = r / POST_IW, lx = r - ly * POST_IW;
        off[q] = u < POST_IN_UNITS ? (unsigned)cb * (unsigned)plane_b + (unsigned)(ly * W + lx) * 16u : safe;
    }
    // What a tile's staging needs that is wave-uniform: where its neighbourhood starts and whether it touches the border.
    struct Stage { const char* org; int x0, y0; bool inner; };
    auto stage_of = [&](const PlcTensor& src, const Item& it) {
        int n;
        Stage st;
        coords(it, n, st.x0, st.y0);
        st.org = plc_img(src, n, a.n_in) + ((long long)(st.y0 - 2) * W + (st.x0 - 2)) * 16;
        st.inner = tile_inner(st.x0, st.y0);
        return st;
    };
    auto opaque_zero = [] {
        int z;
        asm volatile("s_mov_b32 %0, 0" : "=s"(z));
        return z;
    };
    // request unit q of a neighbourhood
    // (o: the unit's offset off[q], passed by value: selecting between off[q] and safe through the closure made the
    // compiler keep off[] in scratch memory and load it, on the vmcnt counter, before every request)
    auto request_unit = [&](const Stage& st, int q, unsigned o, uint4& v) {
        if (!st.inner) {
            const unsigned keep = unit_ok(tid + q * 512 + opaque_zero(), st.x0, st.y0) ? 0xffffffffu : 0u;
            o = (o & keep) | (safe & ~keep);
        }
        v = plc_gload16(st.org + o);
    };
    // zero unit q if it does not exist (border tiles; the tail past the tile's last unit is never stored)
    auto mask_unit = [&](const Stage& st, int q, uint4& v) {
        if (!st.inner && !unit_ok(tid + q * 512 + opaque_zero(), st.x0, st.y0)) v = make_uint4(0u, 0u, 0u, 0u);
    };
    // A tile that begins a chain fetches this thread's gate units (the only wait for memory in the loop, once per T
    // tiles).  tgate[u] is written and read by the same thread only.
    auto fetch_gate = [&](const Item& it) {
        const Stage st = stage_of(a.gate, it);
        uint4 g[NIT];
#pragma unroll
        for (int q = 0; q < NIT; ++q) request_unit(st, q, off[q], g[q]);
#pragma unroll
        for (int q = 0; q < NIT; ++q) {
            mask_unit(st, q, g[q]);
            const int u = tid + q * 512;
            if ((q + 1) * 512 <= POST_IN_UNITS || u < POST_IN_UNITS) tgate[u] = g[q];
        }
    };
    // unit q of a tile's gated input from its feature unit f and the thread's gate unit g: the product, rounded to
    // bf16 once
    auto product_unit = [&](const Stage& st, int q, uint4 f, uint4 g, uint4* dst) {
        mask_unit(st, q, f);
        const int u = tid + q * 512;
        if ((q + 1) * 512 <= POST_IN_UNITS || u < POST_IN_UNITS)
            dst[u] = make_uint4(plc_mul_bf2(f.x, g.x), plc_mul_bf2(f.y, g.y), plc_mul_bf2(f.z, g.z), plc_mul_bf2(f.w, g.w));
    };
    auto gate_units = [&](uint4 (&g)[NIT]) {
#pragma unroll
        for (int q = 0; q < NIT; ++q) g[q] = tgate[(tid + q * 512) % POST_IN_UNITS];
    };
    uint4 pf[NIT];      // this thread's feature units of the tile after the one being convolved
    // First conv + ReLU on rows y0-1 .. y0+14, columns x0-1 .. x0+30 -> tp0 (zero outside the image).  Behind its
    // MFMAs: the request of tile it_r's feature units into pf, one unit every other fragment.  (The product does
    // not go there: its temporaries get the fragment ring's registers from the allocator and every slice then waits
    // for the MFMA that still reads them; an LDS read in the loop would wait for every fragment read in flight.)
    auto conv_first = [&](const Item& it, const uint4* tin, const Item& it_r) {
        int n, x0, y0;
        coords(it, n, x0, y0);
        const Stage sr = stage_of(a.feat, it_r);
        const int r0 = wave * 2;                          // rows of the result tile
        f32x16_t acc[2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 b0 = tb0[2 * g + half];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                acc[r][4 * g + 0] = b0.x; acc[r][4 * g + 1] = b0.y;
                acc[r][4 * g + 2] = b0.z; acc[r][4 * g + 3] = b0.w;
            }
        }
        plc_post_pass<POST_IH, POST_IW>((const char*)tin, r0, px, half, wf0, acc, [&](int f) {
            if (f >= 4 && f < 4 + 2 * NIT && (f & 1) == 0) request_unit(sr, (f - 4) / 2, off[(f - 4) / 2], pf[(f - 4) / 2]);
        });
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int pr = r0 + r, gy = y0 - 1 + pr, gx = x0 - 1 + px;
            const bool inside = (gy >= 0) & (gy < H) & (gx >= 0) & (gx < W);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 v = pack_bf4(fmaxf(acc[r][4 * g + 0], 0.0f), fmaxf(acc[r][4 * g + 1], 0.0f),
                                   fmaxf(acc[r][4 * g + 2], 0.0f), fmaxf(acc[r][4 * g + 3], 0.0f));
                if (!inside) v = make_uint2(0u, 0u);
                *(uint2*)((char*)tp0 + ((g * POST_PH + pr) * POST_PW + px) * 16 + half * 8) = v;
            }
        }
    };
    // The tile's own pixels of the first conv's result (rows 1 .. 14, columns 1 .. 30 of tp0) -> global memory, one
    // 16-byte unit per call: thread t takes units t + 512 q, q < 4.
    constexpr int OWN = 4 * POST_TH * POST_TW, NOWN = (OWN + 511) / 512;      // 1680 units, 4 per thread
    auto save_unit = [&](char* p0g, int x0, int y0, int q) {
        const int u = tid + q * 512 + opaque_zero();
        const int g = u / (POST_TH * POST_TW), r = u - g * (POST_TH * POST_TW);
        const int py = r / POST_TW, pxx = r - py * POST_TW;
        const int gy = y0 + py, gx = x0 + pxx;
        if (u < OWN && gy < H && gx < W)
            *(uint4*)(p0g + g * plane_b + ((long long)gy * W + gx) * 16) = tp0[(g * POST_PH + py + 1) * POST_PW + pxx + 1];
    };
    // Second conv on the tile's 14 x 30 pixels, from tp0: wave w < 7 takes output rows 2w, 2w + 1.  Behind its MFMAs
    // (wave 7: on its own) the intermediate's own pixels leave for global memory (training).
    auto conv_second = [&](const Item& it) {
        int n, x0, y0;
        coords(it, n, x0, y0);
        char* p0g = SAVE_P0 ? plc_img(a.p0, n, a.n_in) : nullptr;
        if (wave >= POST_TH / 2) {
            if (SAVE_P0)
#pragma unroll
                for (int q = 0; q < NOWN; ++q) save_unit(p0g, x0, y0, q);
            return;
        }
        float* og = (float*)plc_img(a.out, n, a.n_in);
        const int q = lane >> 4, col = lane & 15;
        bf16x8_t wf2[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) wf2[t] = __builtin_bit_cast(bf16x8_t, tw2[t * 64 + lane]);
        const int r0 = 2 * wave;                           // output rows y0 + r0, y0 + r0 + 1
        plc_f32x4_t acc[2][2];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                acc[r][nt][0] = q == 0 ? bias2 : 0.0f;     // accumulator j of lane (q, .) is output row 4 q + j
                acc[r][nt][1] = 0.0f; acc[r][nt][2] = 0.0f; acc[r][nt][3] = 0.0f;
            }
        plc_post_pass_1ch<POST_PH, POST_PW>((const char*)tp0, r0, lane, wf2, acc, [&](int f) {
            if (SAVE_P0 && (f & 3) == 0 && f / 4 < NOWN) save_unit(p0g, x0, y0, f / 4);
        });
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int gy = y0 + r0 + r, c = 16 * nt + col, gx = x0 + c;
                if (q == 0 && c < POST_TW && gy < H && gx < W) og[(long long)gy * W + gx] = acc[r][nt][0];
            }
    };

    if (my_tiles > 0) {
        const Stage st = stage_of(a.feat, it0);
#pragma unroll
        for (int q = 0; q < NIT; ++q) request_unit(st, q, off[q], pf[q]);
        fetch_gate(it0);
        uint4 gq[NIT];
        gate_units(gq);
#pragma unroll
        for (int q = 0; q < NIT; ++q) product_unit(st, q, pf[q], gq[q], plc_post_lds);
        if (my_tiles > 1) {
            const Stage s1 = stage_of(a.feat, it1);
#pragma unroll
            for (int q = 0; q < NIT; ++q) request_unit(s1, q, off[q], pf[q]);
        }
    }
    // Every load so far has landed before the loop starts.  Without this the compiler cannot tell on every path into
    // the loop that the stationary weights have arrived, waits vmcnt(0) at their first use inside the first conv's
    // MFMA loop in every round, and with that waits out the feature prefetch the round has just issued.
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0), the other counters left alone
    __syncthreads();
#ifdef PLC_POST_TIMELINE
    // per-wave cycle sums of a round's phases, for devtools/post_probe.hip
    long long pt_sum[5] = {0, 0, 0, 0, 0}, pt_last = __builtin_readcyclecounter();
#define PLC_PT(i) do { const long long now_ = __builtin_readcyclecounter(); pt_sum[i] += now_ - pt_last; pt_last = now_; } while (0)
#else
#define PLC_PT(i) do { } while (0)
#endif
#pragma unroll 1
    for (int k = 0; k < my_tiles; ++k) {
        // Gated input of the next tile, into the other input buffer (free since the last round's first conv).  The
        // product and the request inside the first conv run in EVERY round, also when no tile is left (then on
        // whatever the registers hold, into a buffer nobody reads, and from the current tile's addresses): with a
        // path around them the compiler no longer knows at the loop head that last round's loads were consumed,
        // and makes every request wait for the one before it.
        if (k + 1 < my_tiles && it1.t == 0) fetch_gate(it1);          // the next tile begins a chain
        {
            const Stage sp = stage_of(a.feat, it1);
            uint4 gq[NIT];
            gate_units(gq);
#pragma unroll
            for (int q = 0; q < NIT; ++q) product_unit(sp, q, pf[q], gq[q], plc_post_lds + ((k + 1) & 1) * POST_IN_UNITS);
        }
        PLC_PT(0);
        conv_first(it0, plc_post_lds + (k & 1) * POST_IN_UNITS, k + 2 < my_tiles ? it2 : it0);
        PLC_PT(1);
        __syncthreads();
        PLC_PT(2);
        conv_second(it0);
        PLC_PT(3);
        __syncthreads();
        PLC_PT(4);
        it0 = it1;
        it1 = it2;
        it2 = advance(it2);
    }
#ifdef PLC_POST_TIMELINE
    if (a.timeline && lane == 0)
        for (int i = 0; i < 5; ++i) a.timeline[((long long)blockIdx.x * 8 + wave) * 5 + i] = pt_sum[i];
#endif
#undef PLC_PT
}
