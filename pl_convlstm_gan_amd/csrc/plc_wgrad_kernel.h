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
//   * one persistent workgroup of 8 waves per CU walks 4 x 32 pixel tiles in vertical SEGMENTS of
//     seg_tiles tiles (segments b, b + gridDim.x, ...; neighbouring workgroups take horizontally
//     neighbouring segments, so the tiles in flight at any moment cover whole image rows: taking a
//     contiguous run of tiles per workgroup instead scatters them over the image and was measured 15 %
//     slower).  Each tile's dz block and (4+2) x 34 input halo block are double-buffered in LDS (next
//     tile's global loads are in flight while the current tile is contracted), and the two input rows
//     that a tile shares with the one above it in its segment are copied from the other LDS buffer
//     instead of being fetched again (a third of the input bytes);
//   * the whole dW of the conv stays in registers for the kernel's lifetime: wave (m, g) owns the
//     32-row output tile m and NT_W 32-column tiles (f32x16 accumulators);
//   * a column group is 16 input channels at one tap; one extra group whose B operand is the
//     constant 1.0 yields the bias gradient from the same MFMAs;
//   * every workgroup writes its partial dW to its own slab; plc_wgrad_finalize_kernel sums the
//     slabs in a fixed order (bitwise reproducible, no atomics) and scatters into the PyTorch
//     weight layout.
#pragma once
#include "plc_convr_kernel.h"
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
    int n_tiles;       // number of segments: n_img * ceil(tiles_y / seg_tiles) * tiles_x
    int seg_tiles;     // tiles of a vertical segment
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

    // ---- tile loader: global -> registers, registers -> LDS ---------------------------------
    // Two phases.  load_tile only SELECTS per-lane addresses (a real one, or a 16-byte constant for
    // lanes outside the image) and then loads unconditionally, so no load sits in a divergent
    // branch and all loads of a tile are in flight together while the current tile is contracted.
    // store_tile forms the LDS units from the landed registers: the gated product, the coord-conv
    // unit, the ReLU-masked dz.
    // All eight waves run the loader at the same time (right after the tile's barrier), so whatever
    // it costs in instructions is time the matrix cores idle.  Everything about a lane's units that does
    // not depend on the tile is therefore fixed once, before the tile loop: the units are dealt out in
    // prefetch slots that each hold units of ONE tensor (src0, src1, dz; slot k of a group = units
    // 512 k + tid), a slot's byte offset from the tile's origin is one VGPR, and which lanes load /
    // copy is a 64-bit lane mask in scalar registers.  Per tile and slot there remain a 64-bit add and the
    // select against the zero constant; only tiles that touch the image border test bounds per lane.
    static_assert(STAGE == STAGE_CB8 || STAGE == STAGE_CB8_GATED || STAGE == STAGE_COORD, "unsupported staging");
    constexpr bool HAS2 = STAGE != STAGE_CB8;       // secondary 16-byte load: gate (input part), ReLU mask (dz part)
    constexpr int IN0 = NCB0 * G::PLANE, IN1 = NCB1 * G::PLANE;
    constexpr int S0 = (IN0 + 511) / 512, S1 = (IN1 + 511) / 512, SI = S0 + S1, SD = (G::DZ_UNITS + 511) / 512;
    constexpr int NPF = SI + SD;
    constexpr int NPF2 = HAS2 ? NPF : 1;
    constexpr int NPFC = STAGE == STAGE_COORD ? SI : 1;
    constexpr int ROWS_LOG = G::TH == 8 ? 8 : 7;     // log2(TH * 32)
    static_assert(G::TH * 32 == (1 << ROWS_LOG), "tile rows must be 4 or 8");
    static_assert(STAGE != STAGE_COORD || (SI == 1 && NCB1 == 0), "the coord staging has one input channel block");
    uint4 pf[NPF], pf2[NPF2];
    float pf_rain[NPFC], pf_row[NPFC], pf_col[NPFC];
    const char* const z16 = (const char*)&plc_zero16;
    const char* const o16 = (const char*)&plc_ones16;
    unsigned poff[NPF];                 // byte offset of the lane's unit from the slot's tile origin
    unsigned pyx[SI];                   // input slots: row | column << 8 inside the halo tile
    unsigned long long m_load[NPF];     // lanes whose unit exists in its tensor
    unsigned long long m_copy[SI];      // input slots: lanes whose unit lies in the rows shared with the tile above
    {
        const unsigned plane_b = (unsigned)H * (unsigned)W * 16u;    // (the launcher checks that every offset fits 32 bits)
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            if (k < SI) {
                const bool second = k >= S0;
                const int v = tid + 512 * (second ? k - S0 : k);
                const int cbl = v / G::PLANE, r = v - cbl * G::PLANE;
                const int ly = r / G::LW, lx = r - ly * G::LW;
                const int cb = second ? NCB0 + cbl : cbl;                       // channel block of the conv input
                const bool exists = v < (second ? IN1 : IN0);
                const bool present = exists && (a.in.g_ncb0 == 0 || cb < a.in.g_ncb0);
                poff[k] = STAGE == STAGE_COORD ? (unsigned)(ly * W + lx) * 4u
                                               : (unsigned)cbl * plane_b + (unsigned)(ly * W + lx) * 16u;
                pyx[k] = (unsigned)ly | (unsigned)lx << 8;
                m_load[k] = __builtin_amdgcn_ballot_w64(present);
                m_copy[k] = __builtin_amdgcn_ballot_w64(exists && ly < 2 * G::PAD);
            } else {
                const int v = tid + 512 * (k - SI);
                const int cb = v >> ROWS_LOG, r = v & (G::TH * 32 - 1);
                poff[k] = (unsigned)cb * plane_b + (unsigned)((r >> 5) * W + (r & 31)) * 16u;
                m_load[k] = __builtin_amdgcn_ballot_w64(v < G::DZ_UNITS && cb < a.mcb_valid);
            }
        }
    }
    // Segment s = (image n, segment row sy, strip tx), tx fastest; its tiles j = 0 .. len-1 run downwards.
    // A tile other than a segment's first shares its input rows 0 .. 2*PAD-1 with rows TH .. of the tile
    // before it, i.e. of the buffer being contracted while it is loaded.
    const int seg_tiles = a.seg_tiles;
    const int segs_y = (a.in.tiles_y + seg_tiles - 1) / seg_tiles;
    const int n_segs = a.n_tiles;                       // (number of segments)
    const bool has_mask = HAS2 && a.dz_mask.p != nullptr;
    // tile_setup fixes the wave-uniform part of a tile's addresses, load_slot(k) requests slot k.  (Measured:
    // the eight waves queue their loads at the memory pipeline's input for ~1 us per tile after each
    // barrier, whatever the loader costs in instructions; requesting the slots one by one from inside the
    // MFMA loop instead stalls the waves there and was 35 % slower.)
    const char *tb0 = nullptr, *tb1 = nullptr, *tbg = nullptr, *tbd = nullptr, *tbm = nullptr;
    bool t_skip0 = false, t_skip1 = false, t_inner_in = false, t_inner_dz = false, t_share = false;
    int t_x0 = 0, t_y0 = 0;
    // (n, x0, y0: the tile's image and origin; share: its first 2*PAD input rows come from the tile above)
    auto tile_setup = [&](int n, int x0, int y0, bool share) {
        const long long org_in = ((long long)(y0 - G::PAD) * W + (x0 - G::PAD)) * (STAGE == STAGE_COORD ? 4 : 16);
        const long long org_dz = ((long long)y0 * W + x0) * 16;
        tb0 = plc_img(a.in.src0, n, a.in.n_in) + org_in;
        tb1 = NCB1 > 0 ? plc_img(a.in.src1, n, a.in.n_in) + org_in : nullptr;
        tbg = STAGE == STAGE_CB8_GATED ? plc_img(a.in.gate, n, a.in.n_in) + org_in : nullptr;
        tbd = plc_img(a.dz, n, a.in.n_in) + org_dz;
        tbm = has_mask ? plc_img(a.dz_mask, n, a.in.n_in) + org_dz : nullptr;
        // zero_state: src1 (1) or both sources (2) of the first n_in images are the unstored zero initial state
        t_skip0 = a.in.zero_state == 2 && n < a.in.n_in;
        t_skip1 = a.in.zero_state != 0 && n < a.in.n_in;
        t_inner_in = y0 >= G::PAD && y0 + G::TH + G::PAD <= H && x0 >= G::PAD && x0 + 32 + G::PAD <= W;
        t_inner_dz = y0 + G::TH <= H && x0 + 32 <= W;
        t_share = share;
        t_x0 = x0;
        t_y0 = y0;
    };
    auto load_slot = [&](int k) {
        if (k < SI) {
            const bool second = k >= S0;
            unsigned long long m = t_share ? m_load[k] & ~m_copy[k] : m_load[k];
            bool ok = __builtin_amdgcn_inverse_ballot_w64(m);
            const int gy = t_y0 - G::PAD + (int)(pyx[k] & 255u), gx = t_x0 - G::PAD + (int)(pyx[k] >> 8);
            if (!t_inner_in || STAGE == STAGE_COORD) ok = ok & ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W);
            if constexpr (STAGE == STAGE_COORD) {
                pf_rain[k] = plc_gload4(ok ? tb0 + poff[k] : z16);
                pf_row[k] = plc_gload4(ok ? (const char*)(a.in.row_tab + gy) : z16);
                pf_col[k] = plc_gload4(ok ? (const char*)(a.in.col_tab + gx) : z16);
            } else {
                const bool skip = second ? t_skip1 : t_skip0;
                const char* const base = second ? tb1 : tb0;
                pf[k] = plc_gload16((ok & !skip) ? base + poff[k] : z16);
                if constexpr (STAGE == STAGE_CB8_GATED) pf2[k] = plc_gload16(ok ? tbg + poff[k] : z16);
            }
        } else {
            const int v = tid + 512 * (k - SI);
            bool ok = __builtin_amdgcn_inverse_ballot_w64(m_load[k]);
            if (!t_inner_dz) {
                const int r = v & (G::TH * 32 - 1);
                ok = ok & (t_y0 + (r >> 5) < H) & (t_x0 + (r & 31) < W);
            }
            pf[k] = plc_gload16(ok ? tbd + poff[k] : z16);
            if constexpr (HAS2) pf2[k] = plc_gload16((ok & has_mask) ? tbm + poff[k] : o16);
        }
    };
    auto load_tile = [&](int n, int x0, int y0, bool share) {       // (the first tile: everything at once)
        tile_setup(n, x0, y0, share);
#pragma unroll
        for (int k = 0; k < NPF; ++k) load_slot(k);
    };
    auto store_tile = [&](int buf, bool share) {
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            uint4 v = pf[k];
            if (k < SI) {
                const bool second = k >= S0;
                const int u = tid + 512 * (second ? k - S0 : k);
                const int unit = second ? IN0 + u : u;                          // LDS unit inside the buffer
                if constexpr (STAGE == STAGE_CB8_GATED) {
                    // src0 * gate, rounded to bf16 (the conv's own gated staging forms the same product)
                    const uint4 f = pf[k], g = pf2[k];
                    const float4 f0 = unpack_bf4(make_uint2(f.x, f.y)), f1 = unpack_bf4(make_uint2(f.z, f.w));
                    const float4 g0 = unpack_bf4(make_uint2(g.x, g.y)), g1 = unpack_bf4(make_uint2(g.z, g.w));
                    const uint2 lo = pack_bf4(f0.x * g0.x, f0.y * g0.y, f0.z * g0.z, f0.w * g0.w);
                    const uint2 hi = pack_bf4(f1.x * g1.x, f1.y * g1.y, f1.z * g1.z, f1.w * g1.w);
                    v = make_uint4(lo.x, lo.y, hi.x, hi.y);
                }
                if constexpr (STAGE == STAGE_COORD)       // channels (rain, row coordinate, column coordinate, 0 ...)
                    v = make_uint4(pack_bf2(pf_rain[k], pf_row[k]), pack_bf2(pf_col[k], 0.0f), 0u, 0u);
                // the staged unit TH rows further down in the buffer being contracted
                if (share && __builtin_amdgcn_inverse_ballot_w64(m_copy[k])) v = smem[(buf ^ 1) * G::BUF_UNITS + unit + G::TH * G::LW];
                if (512 * ((second ? k - S0 : k) + 1) <= (second ? IN1 : IN0) || u < (second ? IN1 : IN0))
                    smem[buf * G::BUF_UNITS + unit] = v;
            } else {
                const int u = tid + 512 * (k - SI);
                if constexpr (HAS2) {
                    // dz counts where mask > 0;  bf16 > 0  <=>  sign bit clear and the other 15 bits not all zero
                    auto keep = [](unsigned m) {
                        const unsigned lo = (m & 0x8000u) == 0 && (m & 0x7fffu) != 0 ? 0x0000ffffu : 0u;
                        const unsigned hi = (m & 0x80000000u) == 0 && (m & 0x7fff0000u) != 0 ? 0xffff0000u : 0u;
                        return lo | hi;
                    };
                    const uint4 mk = pf2[k];
                    v.x &= keep(mk.x); v.y &= keep(mk.y); v.z &= keep(mk.z); v.w &= keep(mk.w);
                }
                // each dz channel block is padded to DZ_CB units
                const int unit = G::IN_UNITS + (u >> ROWS_LOG) * G::DZ_CB + (u & (G::TH * 32 - 1));
                if (512 * (k - SI + 1) <= G::DZ_UNITS || u < G::DZ_UNITS) smem[buf * G::BUF_UNITS + unit] = v;
            }
        }
    };

    // the workgroup's position: segment, tile inside it, and what they decode to
    auto seg_len = [&](int sy) {
        const int left = a.in.tiles_y - sy * seg_tiles;
        return left < seg_tiles ? left : seg_tiles;
    };
    PLC_RT_DECL();
    int seg = blockIdx.x, j = 0;
    int tn_ = 0, tx_ = 0, sy_ = 0, len = 0;             // image, strip, segment row and length of `seg`
    auto decode = [&](int s, int& n, int& tx, int& sy) {
        tx = s % a.in.tiles_x;
        const int rem = s / a.in.tiles_x;
        sy = rem % segs_y;
        n = rem / segs_y;
    };
    int buf = 0;
    if (seg < n_segs) {
        decode(seg, tn_, tx_, sy_);
        len = seg_len(sy_);
        load_tile(tn_, tx_ * 32, sy_ * seg_tiles * G::TH, false);
        store_tile(0, false);
    }
    __syncthreads();
    PLC_AS3 char* lds = (PLC_AS3 char*)smem;
    PLC_RT(1);                                        // first tile
    while (seg < n_segs) {
        // the tile after this one: the next of the segment, or the first of this workgroup's next segment
        int nseg = seg, nj = j + 1;
        if (nj == len) {
            nseg = seg + gridDim.x;
            nj = 0;
            if (nseg < n_segs) {
                decode(nseg, tn_, tx_, sy_);
                len = seg_len(sy_);
            }
        }
        const bool has_next = nseg < n_segs;
        if (has_next) load_tile(tn_, tx_ * 32, (sy_ * seg_tiles + nj) * G::TH, G::PAD > 0 && nj > 0);   // flies during the MFMAs below
        PLC_RT(2);                                    // next tile's loads requested
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
        PLC_RT(3);                                    // MFMA loop
#ifdef PLC_TIMELINE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PLC_RT(4);                                    // wait for the next tile's loads
#endif
        if (has_next) store_tile(buf ^ 1, G::PAD > 0 && nj > 0);
        PLC_RT(5);                                    // registers -> LDS
        __syncthreads();
        PLC_RT(6);                                    // barrier
        buf ^= 1;
        seg = nseg;
        j = nj;
    }

    PLC_RT_FLUSH(778);
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
    int row_off, cout_total;        // slab row r is row row_off + r of the conv's dz channels; the conv has cout_total outputs
    int lstm_ch;                    // > 0: the dz channels are an LSTM gate conv's outputs as the engine stores them
                                    // (gate-major, each gate padded to whole 8-channel blocks), lstm_ch hidden channels real
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
            const int ncol = idx % cols_pad;
            int o = jb.row_off + (int)(idx / cols_pad);              // position among the dz channels
            if (jb.lstm_ch > 0) {
                const int chp = (jb.lstm_ch + 7) & ~7, hc = o % chp;
                o = hc < jb.lstm_ch ? (o / chp) * jb.lstm_ch + hc : jb.cout_total;
            }
            const int grp = ncol >> 4, c16 = ncol & 15;
            if (o >= jb.cout_total) {
                // a padding row: no such output channel
            } else if (grp < taps * ncbp) {
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
