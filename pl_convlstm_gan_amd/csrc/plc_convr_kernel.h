// 3x3 implicit-GEMM convolution over a ROLLING window of input rows: the form of plc_conv_kernel.h's
// weight-stationary kernel for the convs whose time is the fetch of their input tiles (the cells' data
// gradients, the post convs).  Same GEMM view, same packed weights, same MFMA order per accumulator and
// the same epilogues, hence bit-identical results; what changes is how the input reaches LDS:
//   * a workgroup owns a SEGMENT: 32 pixel columns x seg_rows consecutive rows of one image, and walks
//     it top to bottom A = RG * NREP rows per pass (RG row groups of waves, NREP rows per wave);
//   * LDS holds a ring of 2A + 2 input rows, each row [channel block][34 columns] of 16-byte units.
//     Pass p reads the A + 2 rows of its window while the A rows that pass p + 1 adds are in flight
//     (LDS-DMA, no VGPRs): they were requested one whole pass earlier, right after the barrier that
//     ended pass p - 1's reads of the slots they overwrite;
//   * every input row is therefore fetched once per segment instead of once per (TH + 2)-row tile: the
//     halo factor falls from (TH + 2) / TH to (seg_rows + 2) / seg_rows (1.59 -> 1.20 for the cell-2 data
//     gradient at 256 x 256), and no wave waits for a whole tile between its MFMA phases.
// The compiler waits for every outstanding LDS-DMA before any LDS access it can see, so the B-fragment
// reads of the MFMA loop are inline asm with hand-placed lgkmcnt waits (LDS reads return in order);
// the remaining LDS accesses of a pass (partial-sum exchange, gate product) sit behind the pass's own
// wait for the rows in flight.
//
// PW form (the ConvLSTM cells' data gradients inside a rollout's backward): the input rows are not fetched
// but FORMED -- the conv's input dz is the output of the cell's pointwise backward (plc_lstm_bwd_unit.h),
// so the workgroup computes each row group of its window from the saved gates, c_{t-1}, dL/dc and the
// incoming dL/dh, writes it into the ring, and stores the pixels it owns (dz for the batched weight
// gradient, dL/dc for the next step) to memory.  The standalone pointwise pass disappears: dz is written
// once and never read back by this step, at the price of recomputing the halo pixels (x 1.2).  It produces
// the same bits and moves 16 % fewer bytes, and it is SLOWER than the two kernels it replaces (cell 2 at
// 256 x 256, B = 4: 100 us against 46 + 37): the pass's loads and its ~2000 vector-ALU cycles per unit
// run between the MFMA phases of a kernel with eight waves per CU instead of under twenty waves'
// worth of memory latency, and the 16 halo-column units of a row group cost a second full iteration.
// The rollout therefore uses it only when PLC_FUSED_PW=1 is set.
#pragma once
#include "plc_conv_kernel.h"
#include "plc_lstm_bwd_unit.h"

#ifdef PLC_TIMELINE
// Instrumented build only: per wave, the time spent in each phase of the pass loop, summed over the
// passes (100 MHz ticks), is left in plc_timeline[(block * waves + wave) * 10 + k] when plc_timeline_sel == id
// (777: this kernel, 778: the weight-gradient kernel).
#define PLC_RT_DECL() long long rt_[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; long long rt_last_ = (long long)wall_clock64(); const long long rt_t0_ = rt_last_
#define PLC_RT(k) do { const long long now_ = (long long)wall_clock64(); rt_[k] += now_ - rt_last_; rt_last_ = now_; } while (0)
#define PLC_RT_FLUSH(id_)                                                                               \
    do {                                                                                                \
        if (plc_timeline && plc_timeline_sel == (id_) && (threadIdx.x & 63) == 0) {                     \
            long long* o_ = plc_timeline + ((long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 10; \
            o_[0] = rt_t0_; o_[9] = (long long)wall_clock64();                                          \
            for (int k_ = 1; k_ < 9; ++k_) o_[k_] = rt_[k_];                                            \
        }                                                                                               \
    } while (0)
#else
#define PLC_RT_DECL() do {} while (0)
#define PLC_RT(k) do {} while (0)
#define PLC_RT_FLUSH(id_) do {} while (0)
#endif

template <int NCB, int NMT, int KSPLIT>
struct PlcRoll {
    static constexpr int NCBW = NCB / KSPLIT;          // channel blocks contracted by one wave
    static constexpr int NCBP = NCBW / 2;
    static constexpr int NKS = (9 * NCBW + 1) / 2;     // k-steps per wave
    static constexpr int RG = 4 / (NMT * KSPLIT);      // row groups of waves
    static constexpr int NREP = 2;                     // rows per wave and pass
    static constexpr int A = RG * NREP;                // rows per pass
    static constexpr int RING = 2 * A + 2;             // row slots
    static constexpr int LW = 34;
    static constexpr int ROW_UNITS = NCB * LW;         // 16-byte units of one input row
    static constexpr int PPR = (ROW_UNITS + 63) / 64;  // LDS-DMA pieces per row (the last one may be partial)
    static constexpr int RING_UNITS = RING * ROW_UNITS;
    static constexpr int RED_UNITS = KSPLIT == 2 ? 4 * 4 * 64 : 0;
    static constexpr int GATE_UNITS = (A + 2) * PPR * 64;   // gated staging only: the multiplier's rows of one request
};

template <int NCB, int NMT, int KSPLIT, int STAGE>
constexpr int plc_convr_lds_bytes() {
    using R = PlcRoll<NCB, NMT, KSPLIT>;
    return (R::RING_UNITS + R::RED_UNITS + (STAGE == STAGE_CB8_GATED ? R::GATE_UNITS : 0)) * 16;
}

// What a lane needs to request its units of an input row, fixed for a whole segment: a wave takes the
// row's pieces k = wave, wave + 4, ... (64 units each, the last one possibly partial), so unit
// u = k * 64 + lane = (channel block, column) is the same for every row and only the row's base moves.
template <int NCB, int NMT, int KSPLIT>
struct PlcRollLane {
    static constexpr int KPW = (PlcRoll<NCB, NMT, KSPLIT>::PPR + 3) / 4;   // pieces of a row per wave
    unsigned offc[KPW];   // byte offset from the row's base (column x0 - 1): channel block * plane + column * 16
    unsigned flags;       // bit j: the unit exists; bit 8 + j: and its column lies inside the image
};

template <int NCB, int NMT, int KSPLIT>
__device__ __forceinline__ void plc_roll_lane_setup(PlcRollLane<NCB, NMT, KSPLIT>& L, int lane, int wave_u, int x0, int H, int W) {
    using R = PlcRoll<NCB, NMT, KSPLIT>;
    const unsigned plane_b = (unsigned)H * (unsigned)W * 16u;     // (the launcher checks NCB * plane_b < 2^32)
    L.flags = 0;
#pragma unroll
    for (int j = 0; j < PlcRollLane<NCB, NMT, KSPLIT>::KPW; ++j) {
        const int k = wave_u + 4 * j;
        const int u = k * 64 + lane;
        const int cb = u / R::LW, lx = u - cb * R::LW;
        const int gx = x0 - 1 + lx;
        const bool exists = (k < R::PPR) & (u < R::ROW_UNITS);
        L.offc[j] = (unsigned)cb * plane_b + (unsigned)lx * 16u;
        L.flags |= (exists ? 1u : 0u) << j;
        L.flags |= ((exists & (gx >= 0) & (gx < W)) ? 1u : 0u) << (8 + j);
    }
}

// Requests `count` input rows starting at image row y_first into ring slots slot0, slot0 + 1, ... (mod
// RING).  Out-of-image units read 16 bytes of zeros.  Returns the number of pieces this wave requested
// (callers derive vmcnt bounds from it).
template <int NCB, int NMT, int KSPLIT, int STAGE>
__device__ __forceinline__ int plc_roll_issue(const ConvArgs& a, const PlcRollLane<NCB, NMT, KSPLIT>& L, uint4* ring, uint4* gstage,
                                              int wave_u, const char* s0, const char* gt, int x0, int y_first, int count, int slot0) {
    using R = PlcRoll<NCB, NMT, KSPLIT>;
    constexpr int KPW = PlcRollLane<NCB, NMT, KSPLIT>::KPW;
    const int H = a.H, W = a.W;
    int issued = 0;
#pragma unroll 1
    for (int i = 0; i < count; ++i) {
        int slot = slot0 + i;
        if (slot >= R::RING) slot -= R::RING;
        const int gy = y_first + i;
        const bool rowok = (gy >= 0) & (gy < H);
        const long long rowoff = ((long long)gy * W + x0 - 1) * 16;       // wave-uniform
#pragma unroll
        for (int j = 0; j < KPW; ++j) {
            const int k = wave_u + 4 * j;
            if (k < R::PPR) {
                if (L.flags >> j & 1u) {
                    const bool ok = rowok & ((L.flags >> (8 + j) & 1u) != 0);
                    const char* src = ok ? s0 + rowoff + L.offc[j] : (const char*)&plc_zero16;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(ring + slot * R::ROW_UNITS + k * 64), 16, 0, 0);
                    if constexpr (STAGE == STAGE_CB8_GATED) {
                        const char* gsrc = ok ? gt + rowoff + L.offc[j] : (const char*)&plc_zero16;
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                                         (__attribute__((address_space(3))) void*)(gstage + (i * R::PPR + k) * 64), 16, 0, 0);
                    }
                }
                ++issued;
            }
        }
    }
    return issued;
}

// Gated staging: the rows just fetched become (source x multiplier), rounded to bf16 once.  A wave
// multiplies the pieces it requested itself.
template <int NCB, int NMT, int KSPLIT>
__device__ __forceinline__ void plc_roll_gate(const PlcRollLane<NCB, NMT, KSPLIT>& L, uint4* ring, const uint4* gstage, int lane,
                                              int wave_u, int count, int slot0) {
    using R = PlcRoll<NCB, NMT, KSPLIT>;
    constexpr int KPW = PlcRollLane<NCB, NMT, KSPLIT>::KPW;
#pragma unroll 1
    for (int i = 0; i < count; ++i) {
        int slot = slot0 + i;
        if (slot >= R::RING) slot -= R::RING;
#pragma unroll
        for (int j = 0; j < KPW; ++j) {
            const int k = wave_u + 4 * j;
            if (k < R::PPR && (L.flags >> j & 1u)) {
                uint4* const p = ring + slot * R::ROW_UNITS + k * 64 + lane;
                const uint4 f = *p, g = gstage[(i * R::PPR + k) * 64 + lane];
                const float4 f0 = unpack_bf4(make_uint2(f.x, f.y)), f1 = unpack_bf4(make_uint2(f.z, f.w));
                const float4 g0 = unpack_bf4(make_uint2(g.x, g.y)), g1 = unpack_bf4(make_uint2(g.z, g.w));
                const uint2 lo = pack_bf4(f0.x * g0.x, f0.y * g0.y, f0.z * g0.z, f0.w * g0.w);
                const uint2 hi = pack_bf4(f1.x * g1.x, f1.y * g1.y, f1.z * g1.z, f1.w * g1.w);
                *p = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
        }
    }
}

// The plain EPI_ACT_CB8 epilogue of one output row (no second output, no multiplier, identity or ReLU):
// a wave-uniform row base, one 32-bit lane offset, no per-row address arithmetic on the vector ALU.
__device__ __forceinline__ void plc_roll_store_row(const ConvArgs& a, const f32x16_t& acc, char* out_img, int mt, int gy,
                                                   unsigned lane_o, bool col_in) {
    if (gy >= a.H) return;                                          // wave-uniform
    const long long plane_b = (long long)a.H * a.W * 16;
    char* const ob = out_img + (long long)gy * a.W * 16 + (long long)(4 * mt) * plane_b;
    const int n_cb = a.ncb_out - 4 * mt;                            // channel blocks of this M tile that are stored
    if (!col_in) return;
    if (a.act == ACT_RELU) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
            if (g < n_cb)
                *(uint2*)(ob + g * plane_b + lane_o) = pack_bf4(fmaxf(acc[4 * g], 0.0f), fmaxf(acc[4 * g + 1], 0.0f),
                                                                fmaxf(acc[4 * g + 2], 0.0f), fmaxf(acc[4 * g + 3], 0.0f));
    } else {
#pragma unroll
        for (int g = 0; g < 4; ++g)
            if (g < n_cb) *(uint2*)(ob + g * plane_b + lane_o) = pack_bf4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
    }
}

// PW form: rows y_first .. y_first + count - 1 of the window, computed.  Thread w of the workgroup takes unit
// (row i, hidden channel block hb, column lx) = (w / (NHB * 34), (w / 34) % NHB, w % 34): 34 consecutive
// threads read 544 consecutive bytes of every input tensor.  Pixels outside the image are the conv's zero
// padding.  A pixel is OWNED (its dz / dL/dc go to memory) by the segment whose output rows [ya, yb) and
// columns x0 .. x0 + 31 contain it; halo pixels are computed again by the neighbouring segments.
template <int NCB, int NMT, int KSPLIT>
__device__ __forceinline__ void plc_roll_stage_pw(const LstmBwdArgs& pw, uint4* ring, int tid, int n, int x0, int y_first,
                                                  int count, int slot0, int ya, int yb) {
    using R = PlcRoll<NCB, NMT, KSPLIT>;
    constexpr int NHB = NCB / 4;
    static_assert(NCB % 4 == 0, "dz has four gate blocks per hidden channel block");
    const int H = pw.H, W = pw.W;
    const long long hw = (long long)H * W;
    const int total = count * NHB * R::LW;
#pragma unroll 1
    for (int w = tid; w < total; w += 256) {
        const int lx = w % R::LW, q = w / R::LW;
        const int hb = q % NHB, i = q / NHB;
        int slot = slot0 + i;
        if (slot >= R::RING) slot -= R::RING;
        const int gy = y_first + i, gx = x0 - 1 + lx;
        const bool in_img = (gy >= 0) & (gy < H) & (gx >= 0) & (gx < W);
        F8 zi, zf, zo, zg, dcp;
#pragma unroll
        for (int j = 0; j < 8; ++j) zi.v[j] = zf.v[j] = zo.v[j] = zg.v[j] = dcp.v[j] = 0.0f;
        const long long u = hb * hw + (long long)gy * W + gx;
        if (in_img) plc_lstm_bwd_unit(pw, n, u, zi, zf, zo, zg, dcp);
        const uint4 pi = make_uint4(pack_bf2(zi.v[0], zi.v[1]), pack_bf2(zi.v[2], zi.v[3]), pack_bf2(zi.v[4], zi.v[5]), pack_bf2(zi.v[6], zi.v[7]));
        const uint4 pf = make_uint4(pack_bf2(zf.v[0], zf.v[1]), pack_bf2(zf.v[2], zf.v[3]), pack_bf2(zf.v[4], zf.v[5]), pack_bf2(zf.v[6], zf.v[7]));
        const uint4 po = make_uint4(pack_bf2(zo.v[0], zo.v[1]), pack_bf2(zo.v[2], zo.v[3]), pack_bf2(zo.v[4], zo.v[5]), pack_bf2(zo.v[6], zo.v[7]));
        const uint4 pg = make_uint4(pack_bf2(zg.v[0], zg.v[1]), pack_bf2(zg.v[2], zg.v[3]), pack_bf2(zg.v[4], zg.v[5]), pack_bf2(zg.v[6], zg.v[7]));
        uint4* const r = ring + (slot * NCB + hb) * R::LW + lx;       // gate-major channel blocks: block = gate * NHB + hb
        r[0 * NHB * R::LW] = pi;
        r[1 * NHB * R::LW] = pf;
        r[2 * NHB * R::LW] = po;
        r[3 * NHB * R::LW] = pg;
        if (in_img & (lx >= 1) & (lx <= 32) & (gy >= ya) & (gy < yb)) {
            const long long su = ((long long)n * pw.ncb) * hw + u;          // unit index in [B][ncb] tensors
            const long long gu = ((long long)n * 4 * pw.ncb) * hw + u;      // unit index of gate 0 in [B][4 ncb]
            *(uint4*)(pw.dz + (gu + 0 * pw.ncb * hw) * 16) = pi;
            *(uint4*)(pw.dz + (gu + 1 * pw.ncb * hw) * 16) = pf;
            *(uint4*)(pw.dz + (gu + 2 * pw.ncb * hw) * 16) = po;
            *(uint4*)(pw.dz + (gu + 3 * pw.ncb * hw) * 16) = pg;
            if (pw.dc_bf16) st_bf8(pw.dc_out + su * 16, dcp);
            else st_f8(pw.dc_out + su * 32, dcp);
        }
    }
}

// ConvArgs: n_tiles = number of segments, n_blocks = gridDim.x (workgroup b takes segments b, b + n_blocks,
// ...), tiles_y = segments per 32-column strip, tiles_x = strips per image, seg_rows = rows per segment
// (a multiple of A).  PLAIN: see plc_epilogue.
// block_x / block_y / n_mgroups: the workgroup's position (as blockIdx.x, blockIdx.y, gridDim.y of a launch
// of its own); lds: its dynamic LDS.
// PW: the input rows are formed from `pw` (see the head of this file) instead of being fetched from a.src0.
template <int NCB, int NMT, int STAGE, int EPI, int KSPLIT, bool PLAIN, bool PW = false>
__device__ __forceinline__ void plc_convr_body(const ConvArgs& a, const int block_x, const int block_y, const int n_mgroups,
                                               uint4* const plc_convr_lds, const LstmBwdArgs* const pw = nullptr) {
    using R = PlcRoll<NCB, NMT, KSPLIT>;
    constexpr int NCBW = R::NCBW, NCBP = R::NCBP, NKS = R::NKS, NREP = R::NREP, A = R::A, RING = R::RING, LW = R::LW;
    constexpr int ROWB = R::ROW_UNITS * 16;
    constexpr bool GATED = STAGE == STAGE_CB8_GATED;
    static_assert(STAGE == STAGE_CB8 || STAGE == STAGE_CB8_GATED, "the rolling conv stages CB8 tensors by LDS-DMA");
    static_assert(!PW || (STAGE == STAGE_CB8 && PLAIN), "the PW form is the cells' plain data-gradient conv");
    static_assert(EPI != EPI_LSTM && EPI != EPI_LSTM_SAVE, "the ConvLSTM cells use plc_conv_kernel.h");
    static_assert(NCB % KSPLIT == 0 && NCBW % 2 == 0, "each wave contracts an even number of channel blocks");
    static_assert(NMT * KSPLIT == 1 || NMT * KSPLIT == 2 || NMT * KSPLIT == 4, "NMT*KSPLIT must be 1, 2 or 4");
    uint4* const ring = plc_convr_lds;
    float4* const red = (float4*)(ring + R::RING_UNITS);
    uint4* const gstage = ring + R::RING_UNITS + R::RED_UNITS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, px = lane & 31;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int m = wave_u % NMT, ks = (wave_u / NMT) % KSPLIT, rg = wave_u / (NMT * KSPLIT);
    const int mt = block_y * NMT + m, n_mt = n_mgroups * NMT;
    const int H = a.H, W = a.W;

    // (asm loads: the compiler keeps no count of them, so it does not drain the row requests that follow
    // before the first MFMA; they are older than every row request, whose waits therefore cover them)
    bf16x8_t wf[NKS];
    plc_load_weights_asm<NKS>(wf, a.wpk + ((long long)(mt * KSPLIT + ks) * NKS) * 64 + lane);
    float4 bv[4];
    {
        const float4* bias4 = (const float4*)a.bpk + mt * 8;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            // (the K-split instantiations are the bias-free data-gradient convs)
            bv[g] = KSPLIT == 2 ? make_float4(0.f, 0.f, 0.f, 0.f) : bias4[2 * g + half];
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(bv[g].x), "+v"(bv[g].y), "+v"(bv[g].z), "+v"(bv[g].w));
    }
    // this lane's byte address inside a ring row: channel block (ks * NCBW + half), column px
    const unsigned lane_b = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)ring +
                            (unsigned)(((ks * NCBW + half) * LW + px) * 16);

    PLC_RT_DECL();
    const int bx = block_x;
#pragma unroll 1
    for (int seg = bx; seg < a.n_tiles; seg += a.n_blocks) {
        const int sy = seg % a.tiles_y;
        const int st = seg / a.tiles_y;
        const int tx = st % a.tiles_x, n = st / a.tiles_x;
        const int x0 = tx * 32, ya = sy * a.seg_rows;
        const int yb = ya + a.seg_rows < H ? ya + a.seg_rows : H;
        const int npass = (yb - ya + A - 1) / A;
        const char* const s0 = PW ? nullptr : plc_img(a.src0, n, a.n_in);
        const char* const gt = GATED ? plc_img(a.gate, n, a.n_in) : nullptr;
        PlcRollLane<NCB, NMT, KSPLIT> ln;
        plc_roll_lane_setup<NCB, NMT, KSPLIT>(ln, lane, wave_u, x0, H, W);
        char* const out_img = plc_img(a.out0, n, a.n_in);
        const unsigned lane_o = (unsigned)((x0 + px) * 16 + half * 8);
        const bool col_in = x0 + px < W;
#define PLC_ROLL_ISSUE(y_first_, count_, slot_) \
    plc_roll_issue<NCB, NMT, KSPLIT, STAGE>(a, ln, ring, gstage, wave_u, s0, gt, x0, y_first_, count_, slot_)
#define PLC_ROLL_EPI(acc_, gy_)                                                                         \
    do {                                                                                                \
        if constexpr (PLAIN) plc_roll_store_row(a, acc_, out_img, mt, gy_, lane_o, col_in);              \
        else plc_epilogue<EPI>(a, acc_, n, mt, n_mt, gy_, x0 + px, half, make_float4(0.f, 0.f, 0.f, 0.f)); \
    } while (0)

        // every wave has finished with the previous segment's rows (and its partial sums)
        if (seg != bx) plc_lds_barrier();
        // window 0 = rows ya - 1 .. ya + A in slots 0 .. A + 1; the rows pass 1 adds go to slots A + 2 .. 2A + 1
        if constexpr (PW) {
            plc_roll_stage_pw<NCB, NMT, KSPLIT>(*pw, ring, tid, n, x0, ya - 1, A + 2, 0, ya, yb);
            if (npass > 1) plc_roll_stage_pw<NCB, NMT, KSPLIT>(*pw, ring, tid, n, x0, ya + A + 1, A, A + 2, ya, yb);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the weights' uncounted loads)
            plc_lds_barrier();
        } else if constexpr (GATED) {
            PLC_ROLL_ISSUE(ya - 1, A + 2, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            plc_roll_gate<NCB, NMT, KSPLIT>(ln, ring, gstage, lane, wave_u, A + 2, 0);
            plc_lds_barrier();
            if (npass > 1) PLC_ROLL_ISSUE(ya + A + 1, A, A + 2);
        } else {
            PLC_ROLL_ISSUE(ya - 1, A + 2, 0);
            int n1 = 0;
            if (npass > 1) n1 = PLC_ROLL_ISSUE(ya + A + 1, A, A + 2);
            plc_wait_vmcnt_le(n1);      // loads retire in order: all but the n1 youngest are window 0
            plc_lds_barrier();
        }

        PLC_RT(1);                       // prologue: weights, window 0
        int w = 0;                       // ring slot of the window's first row
#pragma unroll 1
        for (int p = 0; p < npass; ++p) {
            f32x16_t acc[NREP];
#pragma unroll
            for (int r = 0; r < NREP; ++r)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    acc[r][4 * g + 0] = bv[g].x;
                    acc[r][4 * g + 1] = bv[g].y;
                    acc[r][4 * g + 2] = bv[g].z;
                    acc[r][4 * g + 3] = bv[g].w;
                }
            // Fragment-centric K loop (see plc_conv_body): the B fragment at window row rho, column shift
            // dx, channel-block pair cbp feeds output row r with tap row dy = rho - r.
            constexpr int ROWS = NREP + 2, NF = ROWS * 3 * NCBP, DEPTH = NF < 6 ? NF : 6;
            unsigned rowb[ROWS];
#pragma unroll
            for (int rho = 0; rho < ROWS; ++rho) {
                int slot = w + rg * NREP + rho;
                if (slot >= RING) slot -= RING;
                rowb[rho] = lane_b + (unsigned)(slot * ROWB);
            }
            bf16x8_t fr[NF];
            auto frag = [&](auto fc) {
                constexpr int f = decltype(fc)::value;
                constexpr int rho = f / (3 * NCBP), cbp = (f % (3 * NCBP)) / 3, dx = f % 3;
                constexpr int off = (2 * cbp * LW + dx) * 16;
                static_assert(off < 65536, "DS instruction offsets are 16 bits");
                bf16x8_t& dst = fr[f];
                const unsigned base32 = rowb[rho];
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(base32), "n"(off) : "memory");
            };
            plc_static_for<DEPTH>(frag);
            plc_static_for<NF>([&](auto fc) {
                constexpr int f = decltype(fc)::value;
                if constexpr (f + DEPTH < NF) frag(std::integral_constant<int, f + DEPTH>{});
                constexpr int younger = (NF - 1 - f) < DEPTH ? (NF - 1 - f) : DEPTH;
                bf16x8_t& cur = fr[f];
                asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(cur) : "n"(younger) : "memory");
                constexpr int rho = f / (3 * NCBP), cbp = (f % (3 * NCBP)) / 3, dx = f % 3;
                plc_static_for<NREP>([&](auto rc) {
                    constexpr int r = decltype(rc)::value, dy = rho - r;
                    if constexpr (dy >= 0 && dy < 3)
                        acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[(dy * 3 + dx) * NCBP + cbp], fr[f], acc[r], 0, 0, 0);
                });
                __builtin_amdgcn_sched_barrier(0);
            });
            // (the epilogues test image bounds per lane; no MFMA may move under that divergent test)
#pragma unroll
            for (int r = 0; r < NREP; ++r) asm volatile("" : "+v"(acc[r]));
            PLC_RT(2);                   // MFMA loop

            const int row0 = ya + p * A + rg * NREP;
            float4* const mine = red + wave * 256 + lane;
            const float4* const theirs = red + (wave ^ NMT) * 256 + lane;
            if constexpr (KSPLIT == 2) {
                // partial sums: wave ks finishes row ks of the pass, its partner the other row
                if (p > 0) plc_lds_barrier();     // every wave has read the previous pass's partial sums
                if (ks == 0) {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        mine[g * 64] = make_float4(acc[1][4 * g], acc[1][4 * g + 1], acc[1][4 * g + 2], acc[1][4 * g + 3]);
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        mine[g * 64] = make_float4(acc[0][4 * g], acc[0][4 * g + 1], acc[0][4 * g + 2], acc[0][4 * g + 3]);
                }
            }
            PLC_RT(3);                   // exchange barrier + partial-sum writes (incl. the compiler's own wait for the rows in flight)
            // the rows pass p + 1 adds (requested one pass ago) have landed ...
            if constexpr (!PW) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            PLC_RT(4);                   // wait for the rows in flight
            int wn = w + A;
            if (wn >= RING) wn -= RING;
            if constexpr (GATED) {
                int gs = wn + 2;
                if (gs >= RING) gs -= RING;
                if (p + 1 < npass) plc_roll_gate<NCB, NMT, KSPLIT>(ln, ring, gstage, lane, wave_u, A, gs);
            }
            // ... for every wave, and every wave is done reading this pass's window
            plc_lds_barrier();
            PLC_RT(5);                   // barrier
            if constexpr (KSPLIT == 2) {
                float4 t[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) t[g] = theirs[g * 64];
                // (LDS reads done before the next request: the compiler waits for LDS-DMA in flight ahead of any LDS access)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                PLC_RT(7);               // partial-sum reads
                // the rows pass p + 2 adds replace the first A rows of this pass's window
                if constexpr (!PW) {
                    if (p + 2 < npass) PLC_ROLL_ISSUE(ya + (p + 2) * A + 1, A, w);
                }
                PLC_RT(8);               // next request
                if (ks == 0) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        acc[0][4 * g] += t[g].x; acc[0][4 * g + 1] += t[g].y; acc[0][4 * g + 2] += t[g].z; acc[0][4 * g + 3] += t[g].w;
                    }
                    PLC_ROLL_EPI(acc[0], row0);
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        acc[1][4 * g] += t[g].x; acc[1][4 * g + 1] += t[g].y; acc[1][4 * g + 2] += t[g].z; acc[1][4 * g + 3] += t[g].w;
                    }
                    PLC_ROLL_EPI(acc[1], row0 + 1);
                }
            } else {
                if constexpr (!PW) {
                    if (p + 2 < npass) PLC_ROLL_ISSUE(ya + (p + 2) * A + 1, A, w);
                }
#pragma unroll
                for (int r = 0; r < NREP; ++r) {
                    PLC_ROLL_EPI(acc[r], row0 + r);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // PW: with this pass's results on their way out (and their registers free), form the rows pass p + 2 adds
            if constexpr (PW) {
                if (p + 2 < npass) plc_roll_stage_pw<NCB, NMT, KSPLIT>(*pw, ring, tid, n, x0, ya + (p + 2) * A + 1, A, w, ya, yb);
            }
            w = wn;
            PLC_RT(6);                   // partial-sum reads, next request, epilogue
        }
#undef PLC_ROLL_ISSUE
#undef PLC_ROLL_EPI
    }
    PLC_RT_FLUSH(777);
}

template <int NCB, int NMT, int STAGE, int EPI, int KSPLIT, bool PLAIN>
__global__ __launch_bounds__(256, 2) void plc_convr_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint4 plc_convr_smem[];
    plc_convr_body<NCB, NMT, STAGE, EPI, KSPLIT, PLAIN>(a, blockIdx.x, blockIdx.y, gridDim.y, plc_convr_smem);
}

// PW form of the two cells' data gradients (widths 16 -> 32 and 16 -> 16), alone and as the skewed pair.
struct ConvrPwArgs {
    ConvArgs c;
    LstmBwdArgs pw;
};
template <int NCB, int NMT, int KSPLIT>
__global__ __launch_bounds__(256, 2) void plc_convr_pw_kernel(const ConvrPwArgs p) {
    extern __shared__ __attribute__((aligned(16))) uint4 plc_convr_smem[];
    plc_convr_body<NCB, NMT, STAGE_CB8, EPI_ACT_CB8, KSPLIT, true, true>(p.c, blockIdx.x, 0, 1, plc_convr_smem, &p.pw);
}
struct ConvrPwPairArgs {
    ConvrPwArgs c2, c1;
    int n2;              // workgroups [0, n2) run c2
};
__global__ __launch_bounds__(256, 2) void plc_convr_pw_pair_kernel(const ConvrPwPairArgs p) {
    extern __shared__ __attribute__((aligned(16))) uint4 plc_convr_smem[];
    if ((int)blockIdx.x < p.n2)
        plc_convr_body<16, 2, STAGE_CB8, EPI_ACT_CB8, 2, true, true>(p.c2.c, blockIdx.x, 0, 1, plc_convr_smem, &p.c2.pw);
    else
        plc_convr_body<8, 1, STAGE_CB8, EPI_ACT_CB8, 1, true, true>(p.c1.c, blockIdx.x - p.n2, 0, 1, plc_convr_smem, &p.c1.pw);
}

// The two data-gradient convs of one backward step of the two-cell recurrence that do not depend on each
// other (cell 2 of step t - 1 and cell 1 of step t) in ONE launch, cell 2's workgroups first: cell 1's
// segments start on the CUs that cell 2's workgroups leave, instead of after the whole grid has drained.
struct ConvrPairArgs {
    ConvArgs c2, c1;
    int n2;              // workgroups [0, n2) run c2
};
__global__ __launch_bounds__(256, 2) void plc_convr_dgrad_pair_kernel(const ConvrPairArgs p) {
    extern __shared__ __attribute__((aligned(16))) uint4 plc_convr_smem[];
    if ((int)blockIdx.x < p.n2) plc_convr_body<16, 2, STAGE_CB8, EPI_ACT_CB8, 2, true>(p.c2, blockIdx.x, 0, 1, plc_convr_smem);
    else plc_convr_body<8, 1, STAGE_CB8, EPI_ACT_CB8, 1, true>(p.c1, blockIdx.x - p.n2, 0, 1, plc_convr_smem);
}
