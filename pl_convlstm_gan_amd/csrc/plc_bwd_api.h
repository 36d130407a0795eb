// Backward half of the C ABI (included at the end of plc_api.hip): weight-gradient launcher,
// ConvLSTM cell backward, and backpropagation through time over the whole rollout.
#pragma once
#include "plc_bwd_kernels.h"
#include "plc_wgrad_kernel.h"
#include "plc_wgrad1_kernel.h"

// ---------------------------------------------------------------------------------------------
// weight gradient launcher
// ---------------------------------------------------------------------------------------------
// Persistent workgroups per launch: one per CU when the LDS double buffer fills the CU, two when
// two fit.  Measured on the cell-2 launch (1.07 GB of dz + halo tiles per pass, ~325 us = 3.3 TB/s):
// staging the tiles by LDS-DMA two tiles ahead, issuing all of a tile's prefetch loads together
// instead of one after another, and sharing each B fragment between the KS vertical taps (40 %
// fewer transposing LDS reads, 7 % more MFMAs) each left its time unchanged, so neither global
// latency nor the LDS read count is what bounds it; the parallel prefetch pays only where a tile
// takes many loads per thread (the gated staging: 174 -> 136 us).
template <int NCB0, int NCB1, int MT, int KS>
static int wgrad_max_blocks() {
    return (WgradGeom<NCB0 + NCB1, MT, KS>::LDS_BYTES <= 80 * 1024 ? 2 : 1) * cu_count();
}

template <int NCB0, int NCB1, int MT, int KS>
static size_t wgrad_slab_bytes() {
    using G = WgradGeom<NCB0 + NCB1, MT, KS>;
    return align256((size_t)wgrad_max_blocks<NCB0, NCB1, MT, KS>() * (MT * 32) * (G::NT * 32) * sizeof(float));
}

// Slab memory of a backward pass and the pending reductions over it.  Weight-gradient launches
// carve their slabs from the arena and queue a reduction job; arena_flush reduces everything
// queued in ONE launch (normally once, at the end of the pass) and recycles the memory.  When the
// arena or the job table is full the pending jobs are flushed early, so any arena that holds the
// largest single slab works.
struct SlabArena {
    char* base;
    size_t size, used;
    FinalizeTable tb;
    int blocks;
    bool no_early_flush;     // launches of this pass run on more than one stream: reducing before the join would race
};
static void arena_begin(SlabArena* ar, void* base, size_t size) {
    ar->base = (char*)base;
    ar->size = size;
    ar->used = 0;
    ar->tb.n = 0;
    ar->blocks = 0;
    ar->no_early_flush = false;
}
static int arena_flush(SlabArena* ar, hipStream_t st) {
    if (ar->tb.n > 0) {
        hipLaunchKernelGGL(plc_wgrad_finalize_multi_kernel, dim3(ar->blocks), dim3(256), 0, st, ar->tb);
        PLC_HIP(hipGetLastError());
    }
    ar->used = 0;
    ar->tb.n = 0;
    ar->blocks = 0;
    return PLC_OK;
}

// `in` carries H, W, n_in and the staging tensors of the conv input; dz has mcb_valid channel blocks.
template <int NCB0, int NCB1, int MT, int KS, int STAGE>
// cout: dz channels (= slab rows) this launch covers; c0, c1: REAL channel counts of the two conv inputs.
// row_off / cout_total / lstm_ch: where those rows sit among the conv's outputs (FinalizeJob); cout_total = 0 means
// the launch covers the whole conv (cout_total = cout).
static int launch_wgrad(ConvArgs in, PlcTensor dz, int mcb_valid, int n_img, SlabArena* ar,
                        float* dw, float* db, int cout, int c0, int c1, hipStream_t st,
                        const PlcTensor* dz_mask = nullptr, int ci_off = 0, int c_total = 0, int row_off = 0,
                        int cout_total = 0, int lstm_ch = 0) {
    using G = WgradGeom<NCB0 + NCB1, MT, KS>;
    if (dz_mask && STAGE == STAGE_CB8)
        return plc_fail(PLC_EUNSUPPORTED, "wgrad: the plain CB8 staging does not load a dz mask");
    const size_t need = wgrad_slab_bytes<NCB0, NCB1, MT, KS>();
    if (ar->used + need > ar->size || ar->tb.n == FINALIZE_MAX_JOBS) {
        if (ar->no_early_flush) return plc_fail(PLC_EWORKSPACE, "wgrad: slab arena full before the streams were joined");
        int rc = arena_flush(ar, st);
        if (rc) return rc;
    }
    if (need > ar->size) return plc_fail(PLC_EWORKSPACE, "wgrad: slab arena %zu < %zu bytes", ar->size, need);
    float* slab = (float*)(ar->base + ar->used);
    ar->used += need;
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in;
    a.in.tiles_x = cdiv(in.W, 32);
    a.in.tiles_y = cdiv(in.H, G::TH);
    a.dz = dz;
    if (dz_mask) a.dz_mask = *dz_mask;
    a.mcb_valid = mcb_valid;
    a.rows_valid = cout;
    // (the kernel addresses a lane's unit with a 32-bit offset from the tile's origin in its tensor)
    {
        const int blocks = NCB0 > NCB1 ? (NCB0 > 4 * MT ? NCB0 : 4 * MT) : (NCB1 > 4 * MT ? NCB1 : 4 * MT);
        if ((long long)(blocks + 1) * in.H * in.W * 16 >= (1ll << 32))
            return plc_fail(PLC_EINVAL, "wgrad: image %d x %d too large (a tensor of %d channel blocks must stay below 4 GiB)", in.H, in.W, blocks);
    }
    // vertical segments of up to 16 tiles; shorter ones when that leaves workgroups without work
    const int max_blocks = wgrad_max_blocks<NCB0, NCB1, MT, KS>();
    // (the coord-conv staging reads 4 bytes per pixel: nothing to share, and its tiles are best walked row by row)
    int seg_tiles = STAGE == STAGE_COORD ? 1 : (a.in.tiles_y < 16 ? a.in.tiles_y : 16);
    while (seg_tiles > 1 && (long long)n_img * a.in.tiles_x * cdiv(a.in.tiles_y, seg_tiles) < max_blocks) seg_tiles = (seg_tiles + 1) / 2;
    a.seg_tiles = seg_tiles;
    a.n_tiles = n_img * a.in.tiles_x * cdiv(a.in.tiles_y, seg_tiles);
    a.slab = slab;
    int grid = a.n_tiles < max_blocks ? a.n_tiles : max_blocks;
    if (MT == 4) {
        // A workgroup of the 128-row kernels writes a slab of 160-230 KB, as much as four or five of its tiles
        // read: on small grids one workgroup per tile or two made the slabs (written here, read again by the
        // reduction) the larger part of the pass's traffic.  At least four tiles per workgroup there (x8
        // downscaling from 32 x 32: train step -2.0 %, slab reduction 54 -> 40 us; no change from 128 x 128 up,
        // where every workgroup has more anyway).  PLC_WGRAD_MIN_TILES=1 restores one workgroup per segment.
        static const int min_tiles = [] { const char* e = getenv("PLC_WGRAD_MIN_TILES"); return e && *e ? atoi(e) : 4; }();
        const long long tiles = (long long)n_img * a.in.tiles_x * a.in.tiles_y;
        if (min_tiles > 1 && tiles / grid < min_tiles) {
            const long long gmin = tiles / min_tiles;
            grid = (int)(gmin < 1 ? 1 : (gmin < grid ? gmin : grid));
        }
    }
    auto kern = plc_wgrad_kernel<NCB0, NCB1, MT, KS, STAGE>;
    {
        static bool lds_raised[64] = {false};
        int rc = ensure_lds_limit((const void*)kern, G::LDS_BYTES, lds_raised);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), G::LDS_BYTES, st, a);
    PLC_HIP(hipGetLastError());
    FinalizeJob jb;
    memset(&jb, 0, sizeof(jb));
    jb.slab = slab; jb.dw = dw; jb.db = db;
    jb.n_slabs = grid; jb.rows_pad = MT * 32; jb.cols_pad = G::NT * 32;
    jb.cout = cout; jb.c0 = c0; jb.c1 = c1; jb.ks = KS; jb.ncb = NCB0 + NCB1;
    jb.ci_off = ci_off; jb.c_total = c_total > 0 ? c_total : c0 + c1;
    jb.row_off = row_off; jb.cout_total = cout_total > 0 ? cout_total : cout; jb.lstm_ch = lstm_ch;
    const long long groups = ((long long)cout * G::NT * 32 + 31) / 32;      // 32 output elements per block pass
    jb.n_blocks = (int)(groups < 1 ? 1 : (groups > 1024 ? 1024 : groups));
    jb.first_block = ar->blocks;
    ar->blocks += jb.n_blocks;
    ar->tb.job[ar->tb.n++] = jb;
    return PLC_OK;
}

// Weight gradient of a 32 -> 1 channel 3x3 conv on the vector ALUs (plc_wgrad1_kernel.h); x is the
// conv input (bf16 CB8, 4 channel blocks), dy the fp32 gradient planes of its output.  Queues the
// same slab reduction as the matrix kernel.
static size_t wgrad_cout1_slab_bytes() { return align256((size_t)W1_MAX_BLOCKS * W1_COLS * sizeof(float)); }
static int launch_wgrad_cout1(PlcTensor x, PlcTensor dy, int n_in, int H, int W, int n_img, SlabArena* ar, float* dw,
                              float* db, hipStream_t st) {
    const size_t need = wgrad_cout1_slab_bytes();
    if (ar->used + need > ar->size || ar->tb.n == FINALIZE_MAX_JOBS) {
        if (ar->no_early_flush) return plc_fail(PLC_EWORKSPACE, "wgrad: slab arena full before the streams were joined");
        int rc = arena_flush(ar, st);
        if (rc) return rc;
    }
    if (need > ar->size) return plc_fail(PLC_EWORKSPACE, "wgrad: slab arena %zu < %zu bytes", ar->size, need);
    float* slab = (float*)(ar->base + ar->used);
    ar->used += need;
    Wgrad1Args a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.dy = dy; a.n_in = n_in; a.H = H; a.W = W;
    a.strips_x = cdiv(W, 64);
    a.strips_y = cdiv(H, W1_ROWS);
    const long long strips = (long long)n_img * a.strips_x * a.strips_y;
    if (strips <= 0 || strips >= (1ll << 31)) return plc_fail(PLC_EINVAL, "wgrad: bad strip count %lld", strips);
    a.n_strips = (int)strips;
    a.slab = slab;
    // as many workgroups as give every one the same number of strips (+-1), at most W1_MAX_BLOCKS
    const int per = cdiv(a.n_strips, W1_MAX_BLOCKS);
    const int grid = cdiv(a.n_strips, per);
    hipLaunchKernelGGL(plc_wgrad_cout1_kernel, dim3(grid), dim3(256), 0, st, a);
    PLC_HIP(hipGetLastError());
    FinalizeJob jb;
    memset(&jb, 0, sizeof(jb));
    jb.slab = slab; jb.dw = dw; jb.db = db;
    jb.n_slabs = grid; jb.rows_pad = 1; jb.cols_pad = W1_COLS;
    jb.cout = 1; jb.c0 = 32; jb.c1 = 0; jb.ks = 3; jb.ncb = 4;
    jb.ci_off = 0; jb.c_total = 32;
    jb.row_off = 0; jb.cout_total = 1; jb.lstm_ch = 0;
    jb.n_blocks = W1_COLS / 32;
    jb.first_block = ar->blocks;
    ar->blocks += jb.n_blocks;
    ar->tb.job[ar->tb.n++] = jb;
    return PLC_OK;
}

// Weight gradient of a conv of ANY width on an instantiated kernel <NCBK, 0, MT, KS, STAGE>: the dz rows
// are taken 32*MT at a time and the input channel blocks of src0 and then src1 NCBK at a time; every
// (row group, input chunk) pair is one launch on offset pointers whose slab the common reduction
// scatters into its rectangle of dW (row offset by pointer, input-channel offset by ci_off).  dz is
// re-read once per input chunk: the price of running widths that have no instantiation of their own.
// in.zero_state != 0: src1 of the first n_in images is the unstored zero initial state.
// c0, c1, cout: REAL channel counts.  lstm_ch > 0: dz is an LSTM gate conv's (gate-major, gates padded to whole
// blocks; cout = 4 * lstm_ch), so its channel blocks number 4 * ceil(lstm_ch / 8).
template <int NCBK, int MT, int KS, int STAGE>
static int wgrad_chunked(const ConvArgs& in, int c0, int c1, PlcTensor dz, int cout, int n_img, SlabArena* ar,
                         float* dw, float* db, hipStream_t st, const PlcTensor* dz_mask = nullptr, int lstm_ch = 0) {
    const long long plane = (long long)in.H * in.W * 16;
    const int ctot = c0 + c1, mcb = lstm_ch > 0 ? 4 * cdiv(lstm_ch, 8) : cdiv(cout, 8);
    const int rows_all = lstm_ch > 0 ? 8 * mcb : cout;        // dz channel positions that the launches cover
    for (int mg = 0; mg * 4 * MT < mcb; ++mg) {
        PlcTensor dzg = dz, maskg;
        dzg.p += (long long)mg * 4 * MT * plane;
        if (dz_mask) {
            maskg = *dz_mask;
            maskg.p += (long long)mg * 4 * MT * plane;
        }
        const int mcb_valid = mcb - mg * 4 * MT < 4 * MT ? mcb - mg * 4 * MT : 4 * MT;
        const int rows = rows_all - mg * 32 * MT < 32 * MT ? rows_all - mg * 32 * MT : 32 * MT;
        bool first = true;
        for (int src = 0; src < 2; ++src) {
            const int c_src = src == 0 ? c0 : c1, nb_src = cdiv(c_src, 8);
            for (int cb = 0; cb < nb_src; cb += NCBK) {
                ConvArgs ci = in;
                ci.src0 = src == 0 ? in.src0 : in.src1;
                if (STAGE != STAGE_COORD) ci.src0.p += (long long)cb * plane;
                memset(&ci.src1, 0, sizeof(ci.src1));
                if (STAGE == STAGE_CB8_GATED) ci.gate.p += (long long)cb * plane;
                ci.zero_state = (src == 1 && in.zero_state) ? 2 : 0;
                ci.g_ncb0 = nb_src - cb < NCBK ? nb_src - cb : NCBK;
                const int ch_here = c_src - 8 * cb < 8 * NCBK ? c_src - 8 * cb : 8 * NCBK;
                int rc = launch_wgrad<NCBK, 0, MT, KS, STAGE>(ci, dzg, mcb_valid, n_img, ar, dw, (first && db) ? db : nullptr,
                                                              rows, ch_here, 0, st, dz_mask ? &maskg : nullptr,
                                                              (src == 0 ? 0 : c0) + 8 * cb, ctot, mg * 32 * MT, cout, lstm_ch);
                if (rc) return rc;
                first = false;
            }
        }
    }
    return PLC_OK;
}

static Cb8View view(const void* p, long long img) {
    Cb8View v;
    v.p = (const char*)p;
    v.img = img;
    return v;
}

static LstmBwdArgs lstm_pointwise_bwd_args(Cb8View dh0, Cb8View dh1, Cb8View dh2, const void* dc_in, const void* gates,
                                           const void* c_prev, const void* c_cur, void* dz, void* dc_out, int B, int Ch,
                                           int H, int W, Cb8View dh0_mul = Cb8View{nullptr, 0}, int dc_bf16 = 0) {
    LstmBwdArgs a;
    a.B = B; a.ncb = Ch / 8; a.H = H; a.W = W;
    // (PLC_DC_FP32 in the environment keeps the rollout's dc in fp32: A/B timing and accuracy checks)
    static const bool dc_fp32 = [] { const char* e = getenv("PLC_DC_FP32"); return e && *e && *e != '0'; }();
    a.dc_bf16 = dc_bf16 && !dc_fp32;
    a.dh0 = dh0; a.dh1 = dh1; a.dh2 = dh2;
    a.dh0_mul = dh0_mul;
    a.dc_in = (const char*)dc_in; a.gates = (const char*)gates;
    a.c_prev = (const char*)c_prev; a.c_cur = (const char*)c_cur;
    a.dz = (char*)dz; a.dc_out = (char*)dc_out;
    return a;
}
static int lstm_pointwise_bwd_launch(const LstmBwdArgs& a, hipStream_t st) {
    const long long total = (long long)a.B * a.ncb * a.H * a.W;
    hipLaunchKernelGGL(plc_lstm_pointwise_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, st, a);
    PLC_HIP(hipGetLastError());
    return PLC_OK;
}
static int lstm_pointwise_bwd_launch_pair(const LstmBwdArgs& first, const LstmBwdArgs& second, hipStream_t st) {
    const int n1 = ew_grid((long long)first.B * first.ncb * first.H * first.W);
    const int n2 = ew_grid((long long)second.B * second.ncb * second.H * second.W);
    hipLaunchKernelGGL(plc_lstm_pointwise_bwd_pair_kernel, dim3(n1 + n2), dim3(256), 0, st, first, second, n1);
    PLC_HIP(hipGetLastError());
    return PLC_OK;
}
static int lstm_pointwise_bwd(Cb8View dh0, Cb8View dh1, Cb8View dh2, const void* dc_in, const void* gates,
                              const void* c_prev, const void* c_cur, void* dz, void* dc_out, int B, int Ch, int H,
                              int W, hipStream_t st, Cb8View dh0_mul = Cb8View{nullptr, 0}, int dc_bf16 = 0) {
    return lstm_pointwise_bwd_launch(lstm_pointwise_bwd_args(dh0, dh1, dh2, dc_in, gates, c_prev, c_cur, dz, dc_out, B, Ch, H, W,
                                                             dh0_mul, dc_bf16), st);
}

// data gradient of a cell's gate conv: dz [B][4Ch/8] -> dcat [B][(Cin+Ch)/8] (bf16 CB8, [x | h] order)
static ConvArgs cell_dgrad_args(const void* dz, const void* wpkT, const float* bpk0, void* dcat, int B, int Cin, int Ch,
                                int H, int W) {
    const long long hw = (long long)H * W;
    ConvArgs a = conv_args(H, W, B);
    a.src0 = tens(dz, (4 * Ch / 8) * hw * 16, 0);
    a.out0 = tens(dcat, ((Cin + Ch) / 8) * hw * 16, 0);
    a.wpk = (const uint4*)wpkT;
    a.bpk = bpk0;
    a.act = ACT_NONE;
    a.ncb_out = (Cin + Ch) / 8;
    return a;
}
static int cell_dgrad(const void* dz, const void* wpkT, const float* bpk0, void* dcat, int B, int Cin, int Ch,
                      int H, int W, hipStream_t st) {
    const ConvArgs a = cell_dgrad_args(dz, wpkT, bpk0, dcat, B, Cin, Ch, H, W);
    if (Cin == 16 && Ch == 16)
        return plc_no_rolling() ? launch_conv_th<8, 0, 1, 3, STAGE_CB8, EPI_ACT_CB8, 8, 1>(a, B, 1, st)
                                : launch_convr<8, 1, STAGE_CB8, EPI_ACT_CB8, 1, true>(a, B, 1, st);
    if (Cin == 16 && Ch == 32)
        return plc_no_rolling() ? launch_conv_th<16, 0, 2, 3, STAGE_CB8, EPI_ACT_CB8, 4, 2>(a, B, 1, st)
                                : launch_convr<16, 2, STAGE_CB8, EPI_ACT_CB8, 2, true>(a, B, 1, st);
    if (Cin % 8 == 0 && Ch % 8 == 0)
        return launch_convg<3, STAGE_CB8, EPI_ACT_CB8>(a, 4 * Ch / 8, 0, cdiv(Cin + Ch, 32), B, st);
    return plc_fail(PLC_EUNSUPPORTED, "ConvLSTM cell backward (Cin=%d, Ch=%d): padded channel counts expected", Cin, Ch);
}
static int cell_dgrad_ksplit(int Cin, int Ch) { return (Cin == 16 && Ch == 32) ? 2 : 1; }

// The data gradients of cell 2 (16 -> 32 hidden channels, step t - 1) and cell 1 (16 -> 16, step t) in one launch.
static int cell_dgrad_pair(const ConvArgs& c2, const ConvArgs& c1, int B, hipStream_t st, int slots_div = 1) {
    constexpr int lds2 = plc_convr_lds_bytes<16, 2, 2, STAGE_CB8>(), lds1 = plc_convr_lds_bytes<8, 1, 1, STAGE_CB8>();
    constexpr int lds = lds2 > lds1 ? lds2 : lds1;
    static bool lds_raised[64] = {false};
    int rc = ensure_lds_limit((const void*)plc_convr_dgrad_pair_kernel, lds, lds_raised);
    if (rc) return rc;
    ConvrPairArgs p;
    p.c2 = c2;
    p.c1 = c1;
    const int n2 = convr_geometry<16, 2, STAGE_CB8, EPI_ACT_CB8, 2, true>(p.c2, B, 1, slots_div);
    if (n2 < 0) return n2;
    const int n1 = convr_geometry<8, 1, STAGE_CB8, EPI_ACT_CB8, 1, true>(p.c1, B, 1, slots_div);
    if (n1 < 0) return n1;
    p.n2 = n2;
    hipLaunchKernelGGL(plc_convr_dgrad_pair_kernel, dim3(n2 + n1), dim3(256), lds, st, p);
    PLC_HIP(hipGetLastError());
    return PLC_OK;
}

// The same convs in their PW form (plc_convr_kernel.h): the cell's pointwise backward runs inside the conv, which
// forms its input dz row by row instead of reading it.  `pw` as for lstm_pointwise_bwd_launch, `c` as for cell_dgrad.
// (Off unless PLC_FUSED_PW=1: measured slower than the separate pass, see plc_convr_kernel.h.)
static bool plc_fused_pw() {
    static const bool on = [] { const char* e = getenv("PLC_FUSED_PW"); return e && *e && *e != '0'; }();
    return on;
}
template <int NCB, int NMT, int KSPLIT>
static int cell_dgrad_pw(const LstmBwdArgs& pw, const ConvArgs& c, int B, hipStream_t st) {
    constexpr int lds = plc_convr_lds_bytes<NCB, NMT, KSPLIT, STAGE_CB8>();
    auto kern = plc_convr_pw_kernel<NCB, NMT, KSPLIT>;
    static bool lds_raised[64] = {false};
    int rc = ensure_lds_limit((const void*)kern, lds, lds_raised);
    if (rc) return rc;
    ConvrPwArgs p;
    p.c = c;
    p.pw = pw;
    const int grid = convr_geometry<NCB, NMT, STAGE_CB8, EPI_ACT_CB8, KSPLIT, true>(p.c, B, 1);
    if (grid < 0) return grid;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, p);
    PLC_HIP(hipGetLastError());
    return PLC_OK;
}
static int cell_dgrad_pw_pair(const LstmBwdArgs& pw2, const ConvArgs& c2, const LstmBwdArgs& pw1, const ConvArgs& c1, int B,
                              hipStream_t st) {
    constexpr int lds2 = plc_convr_lds_bytes<16, 2, 2, STAGE_CB8>(), lds1 = plc_convr_lds_bytes<8, 1, 1, STAGE_CB8>();
    constexpr int lds = lds2 > lds1 ? lds2 : lds1;
    static bool lds_raised[64] = {false};
    int rc = ensure_lds_limit((const void*)plc_convr_pw_pair_kernel, lds, lds_raised);
    if (rc) return rc;
    ConvrPwPairArgs p;
    p.c2.c = c2;
    p.c2.pw = pw2;
    p.c1.c = c1;
    p.c1.pw = pw1;
    const int n2 = convr_geometry<16, 2, STAGE_CB8, EPI_ACT_CB8, 2, true>(p.c2.c, B, 1);
    if (n2 < 0) return n2;
    const int n1 = convr_geometry<8, 1, STAGE_CB8, EPI_ACT_CB8, 1, true>(p.c1.c, B, 1);
    if (n1 < 0) return n1;
    p.n2 = n2;
    hipLaunchKernelGGL(plc_convr_pw_pair_kernel, dim3(n2 + n1), dim3(256), lds, st, p);
    PLC_HIP(hipGetLastError());
    return PLC_OK;
}

// h0_zero: the first n_in images of `h` are the zero initial state and are not read (a rollout
// does not store them).
// Cin, Ch: the widths as whole blocks (what the tensors hold); Cin_r, Ch_r: the conv's own (0 = the same).
static int cell_wgrad(const void* dz, PlcTensor x, PlcTensor h, int n_in, int n_img, SlabArena* slab,
                      float* dw, float* db, int Cin, int Ch, int H, int W, hipStream_t st, bool h0_zero = false,
                      int Cin_r = 0, int Ch_r = 0) {
    if (Cin_r <= 0) Cin_r = Cin;
    if (Ch_r <= 0) Ch_r = Ch;
    const long long hw = (long long)H * W;
    ConvArgs in = conv_args(H, W, n_in);
    in.src0 = x;
    in.src1 = h;
    in.zero_state = h0_zero ? 1 : 0;
    PlcTensor dzt = tens(dz, (4 * Ch / 8) * hw * 16, (long long)n_in * (4 * Ch / 8) * hw * 16);
    if (Cin == 16 && Ch == 16)
        return launch_wgrad<2, 2, 2, 3, STAGE_CB8>(in, dzt, 8, n_img, slab, dw, db, 64, Cin_r, Ch_r, st, nullptr, 0, 0, 0,
                                                   4 * Ch_r, Ch_r);
    if (Cin == 16 && Ch == 32)
        return launch_wgrad<2, 4, 4, 3, STAGE_CB8>(in, dzt, 16, n_img, slab, dw, db, 128, Cin_r, Ch_r, st, nullptr, 0, 0, 0,
                                                   4 * Ch_r, Ch_r);
    if (Cin % 8 == 0 && Ch % 8 == 0)
        return wgrad_chunked<4, 4, 3, STAGE_CB8>(in, Cin_r, Ch_r, dzt, 4 * Ch_r, n_img, slab, dw, db, st, nullptr, Ch_r);
    return plc_fail(PLC_EUNSUPPORTED, "ConvLSTM cell backward (Cin=%d, Ch=%d): padded channel counts expected", Cin, Ch);
}
static size_t cell_wgrad_slab_bytes(int Cin, int Ch) {
    if (Cin == 16 && Ch == 16) return wgrad_slab_bytes<2, 2, 2, 3>();
    if (Cin == 16 && Ch == 32) return wgrad_slab_bytes<2, 4, 4, 3>();
    return wgrad_slab_bytes<4, 0, 4, 3>();
}

// ---------------------------------------------------------------------------------------------
// ConvLSTM cell backward on boundary-layout tensors
// ---------------------------------------------------------------------------------------------
struct CellBwdWs {
    CellWs f;                 // forward carve (x, h, c, wpk, bpk, ho, co)
    size_t gates, wpkT, bpk0, dh8, dz, dcat, dc, slab, total;
};
static CellBwdWs cell_bwd_ws(int B, int Cin, int Ch, int H, int W) {
    CellBwdWs s;
    Cin = pad8(Cin);            // (the engine's widths are whole 8-channel blocks)
    Ch = pad8(Ch);
    s.f = cell_ws(B, Cin, Ch, H, W);
    const size_t hw = (size_t)H * W;
    size_t o = s.f.total;
    auto take = [&](size_t bytes) { size_t at = o; o += align256(bytes); return at; };
    s.gates = take((size_t)B * (4 * Ch / 8) * hw * 16);
    s.wpkT = take(packedT_weight_bytes(4 * Ch, Cin, Ch, 3, cell_dgrad_ksplit(Cin, Ch)));
    s.bpk0 = take((size_t)packT_nmt(Cin, Ch) * 32 * sizeof(float));
    s.dh8 = take((size_t)B * (Ch / 8) * hw * 16);
    s.dz = take((size_t)B * (4 * Ch / 8) * hw * 16);
    s.dcat = take((size_t)B * ((Cin + Ch) / 8) * hw * 16);
    s.dc = take((size_t)B * (Ch / 8) * hw * 32 * 2);
    s.slab = take(cell_wgrad_slab_bytes(Cin, Ch));
    s.total = o;
    return s;
}

extern "C" int plc_lstm_pointwise_bwd(const void* dh0, long long dh0_image_bytes, const void* dh1,
                                      long long dh1_image_bytes, const void* dh2, long long dh2_image_bytes,
                                      const float* dc_next, const void* gates, const float* c_prev, void* dz,
                                      float* dc, int B, int Ch, int H, int W, void* stream) {
    PLC_REQUIRE(gates && dz && dc, "plc_lstm_pointwise_bwd: null pointer");
    PLC_REQUIRE(B > 0 && H > 0 && W > 0, "plc_lstm_pointwise_bwd: non-positive size");
    PLC_REQUIRE(Ch > 0 && Ch % 8 == 0, "plc_lstm_pointwise_bwd: Ch must be a positive multiple of 8");
    return lstm_pointwise_bwd(view(dh0, dh0_image_bytes), view(dh1, dh1_image_bytes), view(dh2, dh2_image_bytes),
                              dc_next, gates, c_prev, nullptr, dz, dc, B, Ch, H, W, (hipStream_t)stream);
}

extern "C" int plc_lstm_pointwise_bwd_bf16dc(const void* dh0, long long dh0_image_bytes, const void* dh1,
                                             long long dh1_image_bytes, const void* dh2, long long dh2_image_bytes,
                                             const void* dc_next, const void* gates, const float* c_prev, void* dz,
                                             void* dc, int B, int Ch, int H, int W, void* stream) {
    PLC_REQUIRE(gates && dz && dc, "plc_lstm_pointwise_bwd_bf16dc: null pointer");
    PLC_REQUIRE(B > 0 && H > 0 && W > 0, "plc_lstm_pointwise_bwd_bf16dc: non-positive size");
    PLC_REQUIRE(Ch > 0 && Ch % 8 == 0, "plc_lstm_pointwise_bwd_bf16dc: Ch must be a positive multiple of 8");
    return lstm_pointwise_bwd(view(dh0, dh0_image_bytes), view(dh1, dh1_image_bytes), view(dh2, dh2_image_bytes),
                              dc_next, gates, c_prev, nullptr, dz, dc, B, Ch, H, W, (hipStream_t)stream,
                              Cb8View{nullptr, 0}, 1);
}

extern "C" int plc_lstm_pointwise_bwd_pair_bf16dc(const plc_pointwise_bwd_args* first, const plc_pointwise_bwd_args* second,
                                                  int B, int H, int W, void* stream) {
    PLC_REQUIRE(first && second, "plc_lstm_pointwise_bwd_pair_bf16dc: null argument block");
    PLC_REQUIRE(B > 0 && H > 0 && W > 0, "plc_lstm_pointwise_bwd_pair_bf16dc: non-positive size");
    LstmBwdArgs a[2];
    const plc_pointwise_bwd_args* in[2] = {first, second};
    for (int k = 0; k < 2; ++k) {
        const plc_pointwise_bwd_args& q = *in[k];
        PLC_REQUIRE(q.gates && q.dz && q.dc, "plc_lstm_pointwise_bwd_pair_bf16dc: null pointer");
        PLC_REQUIRE(q.Ch > 0 && q.Ch % 8 == 0, "plc_lstm_pointwise_bwd_pair_bf16dc: Ch must be a positive multiple of 8");
        a[k] = lstm_pointwise_bwd_args(view(q.dh0, q.dh0_image_bytes), view(q.dh1, q.dh1_image_bytes), view(nullptr, 0),
                                       q.dc_next, q.gates, q.c_prev, nullptr, q.dz, q.dc, B, q.Ch, H, W,
                                       Cb8View{nullptr, 0}, 1);
    }
    return lstm_pointwise_bwd_launch_pair(a[0], a[1], (hipStream_t)stream);
}

extern "C" size_t plc_convlstm_cell_bwd_workspace_bytes(int B, int Cin, int Ch, int H, int W) {
    if (B <= 0 || Cin <= 0 || Ch <= 0 || H <= 0 || W <= 0) return 0;
    return cell_bwd_ws(B, Cin, Ch, H, W).total;
}

extern "C" int plc_convlstm_cell_bwd(const float* x, const float* h, const float* c, const float* weight,
                                     const float* bias, const float* dh_next, const float* dc_next, float* dx,
                                     float* dh, float* dc, float* dweight, float* dbias, void* workspace,
                                     size_t workspace_bytes, int B, int Cin, int Ch, int H, int W, void* stream) {
    PLC_REQUIRE(x && h && c && weight && dh_next && dc_next && dx && dh && dc && dweight && workspace,
                "plc_convlstm_cell_bwd: null pointer");
    PLC_REQUIRE(B > 0 && H > 0 && W > 0, "plc_convlstm_cell_bwd: non-positive size");
    PLC_REQUIRE(((uintptr_t)workspace & 255) == 0, "plc_convlstm_cell_bwd: workspace must be 256-byte aligned");
    PLC_REQUIRE(Cin > 0 && Ch > 0, "plc_convlstm_cell_bwd: non-positive channel count");
    const CellBwdWs s = cell_bwd_ws(B, Cin, Ch, H, W);
    const int Cin_r = Cin, Ch_r = Ch;       // the cell's own widths; below Cin, Ch are the whole-block widths
    Cin = pad8(Cin);
    Ch = pad8(Ch);
    if (workspace_bytes < s.total)
        return plc_fail(PLC_EWORKSPACE, "plc_convlstm_cell_bwd: workspace %zu < %zu bytes", workspace_bytes, s.total);
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    const long long hw = (long long)H * W;
    int rc;
    // recompute the forward to obtain the gate activations
    if ((rc = plc_nchw_f32_to_cb8_bf16(x, ws + s.f.x, B, Cin_r, H, W, stream))) return rc;
    if ((rc = plc_nchw_f32_to_cb8_bf16(h, ws + s.f.h, B, Ch_r, H, W, stream))) return rc;
    if ((rc = plc_nchw_f32_to_cb8_f32(c, (float*)(ws + s.f.c), B, Ch_r, H, W, stream))) return rc;
    if ((rc = plc_pack_conv_weight(weight, bias, ws + s.f.wpk, (float*)(ws + s.f.bpk), 4 * Ch_r, Cin_r, Ch_r, 3,
                                   PLC_PACK_LSTM, stream)))
        return rc;
    if ((rc = cell_fwd_cb8(ws + s.f.x, ws + s.f.h, (float*)(ws + s.f.c), ws + s.f.wpk, (float*)(ws + s.f.bpk),
                           ws + s.f.ho, (float*)(ws + s.f.co), ws + s.gates, B, Cin, Ch, H, W, st)))
        return rc;
    // pointwise backward
    if ((rc = plc_nchw_f32_to_cb8_bf16(dh_next, ws + s.dh8, B, Ch_r, H, W, stream))) return rc;
    float* dc_in = (float*)(ws + s.dc);
    float* dc_out = (float*)(ws + s.dc + (size_t)B * (Ch / 8) * hw * 32);
    if ((rc = plc_nchw_f32_to_cb8_f32(dc_next, dc_in, B, Ch_r, H, W, stream))) return rc;
    if ((rc = lstm_pointwise_bwd(view(ws + s.dh8, (Ch / 8) * hw * 16), view(nullptr, 0), view(nullptr, 0), dc_in,
                                 ws + s.gates, ws + s.f.c, ws + s.f.co, ws + s.dz, dc_out, B, Ch, H, W, st)))
        return rc;
    // data gradient
    if ((rc = pack_conv_weight_T(weight, ws + s.wpkT, (float*)(ws + s.bpk0), 4 * Ch_r, Cin_r, Ch_r, 3,
                                 cell_dgrad_ksplit(Cin, Ch), st, !cell_fast(Cin, Ch), Ch_r)))
        return rc;
    if ((rc = cell_dgrad(ws + s.dz, ws + s.wpkT, (const float*)(ws + s.bpk0), ws + s.dcat, B, Cin, Ch, H, W, st)))
        return rc;
    // weight gradient
    SlabArena arena;
    arena_begin(&arena, ws + s.slab, cell_wgrad_slab_bytes(Cin, Ch));
    if ((rc = cell_wgrad(ws + s.dz, tens(ws + s.f.x, (Cin / 8) * hw * 16, 0), tens(ws + s.f.h, (Ch / 8) * hw * 16, 0),
                         B, B, &arena, dweight, dbias, Cin, Ch, H, W, st, false, Cin_r, Ch_r)))
        return rc;
    if ((rc = arena_flush(&arena, st))) return rc;
    // back to the boundary layout: dcat = [dx (Cin) | dh (Ch)] per image
    {
        const long long cat_img = ((Cin + Ch) / 8) * hw * 16;
        // dx / dh are strided views of dcat: convert image by image through the generic kernel
        for (int b = 0; b < B; ++b) {
            if ((rc = plc_cb8_bf16_to_nchw_f32(ws + s.dcat + b * cat_img, dx + (long long)b * Cin_r * hw, 1, Cin_r, H, W, stream)))
                return rc;
            if ((rc = plc_cb8_bf16_to_nchw_f32(ws + s.dcat + b * cat_img + (Cin / 8) * hw * 16, dh + (long long)b * Ch_r * hw,
                                               1, Ch_r, H, W, stream)))
                return rc;
        }
    }
    if ((rc = plc_cb8_f32_to_nchw_f32(dc_out, dc, B, Ch_r, H, W, stream))) return rc;
    return PLC_OK;
}

// ---------------------------------------------------------------------------------------------
// Generator rollout backward (BPTT)
// ---------------------------------------------------------------------------------------------
struct BwdPlan {
    size_t post2T, post0T, cell1T, cell2T, lu2T, dem2T, upT[8], bpk0;
    size_t d_pre0, d_feat, d_pre, d_lu_pre, d_dem_pre, d_lu_mid, d_dem_mid, d_aux_hr;
    size_t d_conv[8], d_in[8];
    size_t d_mid, d_up;     // (size_t)-1 when the corresponding bilinear resize is absent
    size_t dz1, dz2, dcat1, dcat2, dc1, dc2, slab, slab_bytes, total;
};

static size_t max_sz(size_t a, size_t b) { return a > b ? a : b; }
// zero bias rows for the data-gradient convs: 32 per output tile of the widest one, at least 8 tiles
static size_t bwd_zero_bias_floats(int C1, int C2) {
    const int widest = C1 + C2 > 2 * C1 ? C1 + C2 : 2 * C1;
    const int tiles = cdiv(widest, 32) > 8 ? cdiv(widest, 32) : 8;
    return (size_t)tiles * 32;
}

static int bwd_plan(const plc_rollout_desc* d, const RolloutPlan& fp, BwdPlan* p) {
    const int C1 = pad8(d->C1), C2 = pad8(d->C2), R2 = d->C2, B = d->B, T = d->T, NI = d->B * d->T;   // (whole-block widths)
    const size_t hw = (size_t)d->H * d->W, ohw = (size_t)d->out_H * d->out_W;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += align256(bytes); return at; };
    // gradients of the feature map before the resize to target_size / before the residual bilinear resize
    p->d_mid = (d->out_H != d->mid_H || d->out_W != d->mid_W)
                   ? take((size_t)NI * (C2 / 8) * d->mid_H * d->mid_W * 16) : (size_t)-1;
    p->d_up = (d->mid_H != fp.up_H || d->mid_W != fp.up_W)
                  ? take((size_t)NI * (C2 / 8) * fp.up_H * fp.up_W * 16) : (size_t)-1;
    p->post2T = take(packedT_weight_bytes(1, 32, 0, 3, 1));
    p->post0T = take(packedT_weight_bytes(32, C2, 0, 3, 1));
    p->cell1T = take(packedT_weight_bytes(4 * C1, C1, C1, 3, 1));
    p->cell2T = take(packedT_weight_bytes(4 * C2, C1, C2, 3, 2));
    p->lu2T = take(packedT_weight_bytes(C2, R2 / 2, 0, 1, 1));
    p->dem2T = take(packedT_weight_bytes(C2, R2 / 2, 0, 1, 1));
    for (int k = 0; k < d->n_blocks; ++k) p->upT[k] = take(packedT_weight_bytes(4 * C2, C2, 0, 3, 2));
    p->bpk0 = take(bwd_zero_bias_floats(C1, C2) * sizeof(float));
    p->d_pre0 = take((size_t)NI * 4 * ohw * 16);
    p->d_feat = take((size_t)NI * (C2 / 8) * ohw * 16);
    p->d_pre = take((size_t)NI * (C2 / 8) * ohw * 16);
    p->d_lu_pre = take((size_t)B * (C2 / 8) * ohw * 16);
    p->d_dem_pre = take((size_t)B * (C2 / 8) * ohw * 16);
    p->d_lu_mid = take((size_t)B * cdiv(R2 / 2, 8) * ohw * 16);
    p->d_dem_mid = take((size_t)B * cdiv(R2 / 2, 8) * ohw * 16);
    p->d_aux_hr = take((size_t)B * (d->L > 1 ? d->L : 1) * ohw * 4);   // fp32 input gradient of an attention branch, output grid
    for (int k = 0; k < d->n_blocks; ++k) {
        const size_t s = (size_t)(d->H << k) * (d->W << k);
        p->d_conv[k] = take((size_t)NI * (4 * C2 / 8) * s * 16);
        p->d_in[k] = take((size_t)NI * (C2 / 8) * s * 16);
    }
    p->dz1 = take((size_t)NI * (4 * C1 / 8) * hw * 16);
    p->dz2 = take((size_t)NI * (4 * C2 / 8) * hw * 16);
    p->dcat1 = take((size_t)NI * (2 * C1 / 8) * hw * 16);
    p->dcat2 = take((size_t)2 * B * ((C1 + C2) / 8) * hw * 16);
    p->dc1 = take((size_t)2 * B * (C1 / 8) * hw * 32);
    p->dc2 = take((size_t)2 * B * (C2 / 8) * hw * 32);
    // slab arena: the slabs of every weight-gradient launch of the pass, so one reduction at the end serves all
    size_t sb = wgrad_slab_bytes<2, 4, 4, 3>() + wgrad_slab_bytes<2, 2, 2, 3>();            // cell 2, cell 1
    sb += wgrad_slab_bytes<4, 0, 1, 3>() + wgrad_cout1_slab_bytes();                         // post0, post2
    sb += wgrad_slab_bytes<1, 0, 1, 3>();                                                    // stem
    sb += 2 * wgrad_slab_bytes<2, 0, 1, 1>();                                                // the two 1x1 gate convs
    sb += wgrad_slab_bytes<1, 0, 1, 3>() + max_sz(wgrad_slab_bytes<1, 0, 1, 3>(), wgrad_slab_bytes<2, 0, 1, 3>());
    sb += (size_t)d->n_blocks * wgrad_slab_bytes<4, 0, 4, 3>();                              // upsample blocks
    // widths without their own instantiations run the chunked launches: room for a few of their slabs
    // (the arena reduces early whenever it runs full)
    if (!widths_fast(C1, C2)) sb += 4 * max_sz(wgrad_slab_bytes<4, 0, 4, 3>(), wgrad_slab_bytes<4, 0, 4, 1>());
    p->slab_bytes = sb;
    p->slab = take(sb);
    p->total = o;
    (void)T;
    return PLC_OK;
}

extern "C" size_t plc_generator_bwd_workspace_bytes(const plc_rollout_desc* d) {
    RolloutPlan fp;
    BwdPlan bp;
    if (rollout_plan(d, &fp) != PLC_OK) return 0;
    if (bwd_plan(d, fp, &bp) != PLC_OK) return 0;
    return bp.total;
}

extern "C" int plc_generator_rollout_bwd(const plc_rollout_desc* d, const plc_rollout_grads* g, void* stream) {
    std::lock_guard<std::mutex> aux_lock(plc_aux_mutex);
    RolloutPlan fp;
    BwdPlan bp;
    int rc = rollout_plan(d, &fp);
    if (rc) return rc;
    PLC_REQUIRE(g, "rollout backward: null gradient descriptor");
    PLC_REQUIRE(d->save_for_backward, "rollout backward: the forward pass was run without save_for_backward");
    if ((rc = bwd_plan(d, fp, &bp))) return rc;
    PLC_REQUIRE(g->grad_out && g->workspace && d->workspace && d->rain && d->row_tab && d->col_tab,
                "rollout backward: null tensor pointer");
    PLC_REQUIRE(g->init_w && g->init_b && g->cell1_w && g->cell1_b && g->cell2_w && g->cell2_b && g->dem0_w &&
                    g->dem0_b && g->dem2_w && g->dem2_b && g->lu0_w && g->lu0_b && g->lu2_w && g->lu2_b &&
                    g->post0_w && g->post0_b && g->post2_w && g->post2_b,
                "rollout backward: null gradient pointer");
    for (int k = 0; k < d->n_blocks; ++k) PLC_REQUIRE(g->up_w[k] && g->up_b[k], "rollout backward: null upsample gradient %d", k);
    PLC_REQUIRE(((uintptr_t)g->workspace & 255) == 0, "rollout backward: workspace must be 256-byte aligned");
    if (g->workspace_bytes < bp.total)
        return plc_fail(PLC_EWORKSPACE, "rollout backward: workspace %zu < %zu bytes", g->workspace_bytes, bp.total);
    if (d->workspace_bytes < fp.total)
        return plc_fail(PLC_EWORKSPACE, "rollout backward: forward workspace %zu < %zu bytes", d->workspace_bytes, fp.total);

    hipStream_t st = (hipStream_t)stream;
    ForkGuard forks(st);              // joins the auxiliary streams on an early (error) return too
    char* wf = (char*)d->workspace;   // forward workspace: saved activations
    char* wb = (char*)g->workspace;   // backward scratch
    // C1, C2: the widths as whole 8-channel blocks (every buffer and kernel); R1, R2: the model's own (packers and
    // the slab reduction, which leave the padding rows and columns out)
    const int B = d->B, T = d->T, H = d->H, W = d->W, C1 = pad8(d->C1), C2 = pad8(d->C2), R1 = d->C1, R2 = d->C2, L = d->L, NI = B * T;
    const int oH = d->out_H, oW = d->out_W;
    const long long hw = (long long)H * W, ohw = (long long)oH * oW;
    SlabArena arena;
    arena_begin(&arena, wb + bp.slab, bp.slab_bytes);
    SlabArena* const slab = &arena;
    // Weight gradients are leaves: nothing but the final slab reduction reads them.  With the instantiated widths
    // (whose arena holds every slab of the pass, so no reduction happens early) those of the two post convs and
    // the attention branches' whole backward go to the side stream and run beside the data-gradient chain and the
    // BPTT loop.  side_aux != nullptr: that is the case.
    PlcAux* side_aux = nullptr;
    if (widths_fast(C1, C2) && plc_use_side_stream((long long)d->B * d->out_H * d->out_W)) {
        if ((rc = plc_aux(&side_aux))) return rc;
        arena.no_early_flush = true;
    }
    hipStream_t const ss = side_aux ? side_aux->side : (hipStream_t)stream;       // the leaves' stream
    const float* bpk0 = (const float*)(wb + bp.bpk0);
    const int mcb = cdiv(R2 / 2, 8);
    const long long fimg = (C2 / 8) * ohw * 16;   // one C2-channel image at output resolution

    hipLaunchKernelGGL(plc_zero_words_kernel, dim3(1), dim3(256), 0, st, (unsigned*)(wb + bp.bpk0),
                       (int)bwd_zero_bias_floats(C1, C2));
    PLC_HIP(hipGetLastError());
    const bool fast = widths_fast(C1, C2);
    const int gen = fast ? 0 : 1;
    // ---- transposed (data-gradient) weights -------------------------------------------------
    {
        PackBatch batch;
        pack_batch_begin(&batch);
        auto packT = [&](const float* w, size_t at, int cout, int c0, int c1, int ks, int ksplit, int g, int lstm_k = 0,
                         int k_blocks = 0) {
            return pack_batch_add(&batch, pack_job_T(w, wb + at, nullptr, cout, c0, c1, ks, ksplit, g, lstm_k, k_blocks), st);
        };
        if ((rc = packT(d->post2_w, bp.post2T, 1, 32, 0, 3, 1, 0))) return rc;
        if ((rc = packT(d->post0_w, bp.post0T, 32, R2, 0, 3, 1, 0))) return rc;
        if ((rc = packT(d->cell1_w, bp.cell1T, 4 * R1, R1, R1, 3, cell_dgrad_ksplit(C1, C1), !cell_fast(C1, C1), R1))) return rc;
        if ((rc = packT(d->cell2_w, bp.cell2T, 4 * R2, R1, R2, 3, cell_dgrad_ksplit(C1, C2), !cell_fast(C1, C2), R2))) return rc;
        if ((rc = packT(d->lu2_w, bp.lu2T, R2, R2 / 2, 0, 1, 1, gen))) return rc;
        if ((rc = packT(d->dem2_w, bp.dem2T, R2, R2 / 2, 0, 1, 1, gen))) return rc;
        // (the gradient before the pixel shuffle holds 4 * C2 channels, those past 4 * R2 zero)
        for (int k = 0; k < d->n_blocks; ++k)
            if ((rc = packT(d->up_w[k], bp.upT[k], 4 * R2, R2, 0, 3, 2, gen, 0, 4 * C2 / 8))) return rc;
        if ((rc = pack_batch_flush(&batch, st))) return rc;
    }
    if (side_aux) {
        PLC_HIP(hipEventRecord(side_aux->sev[4], st));
        PLC_HIP(hipStreamWaitEvent(ss, side_aux->sev[4], 0));
        forks.opened(ss, side_aux->sev[5]);
    }

    // ---- output convs -------------------------------------------------------------------------
    const PlcTensor gout = tens(g->grad_out, (long long)T * ohw * 4, ohw * 4);   // n = t*B + b -> grad_out[b][t]
    {
        // the one-output-channel conv: its weight gradient streams the conv input once on the vector
        // ALUs and reads the loss gradient in fp32 as it is
        // (a leaf, but on the side stream, beside the conv that follows, it only slows that conv by what it saves)
        if ((rc = launch_wgrad_cout1(tens(wf + fp.p0, 4 * ohw * 16, (long long)B * 4 * ohw * 16), gout, B, oH, oW, NI,
                                     slab, g->post2_w, g->post2_b, st)))
            return rc;
        ConvArgs in;
        ConvArgs a = conv_args(oH, oW, B);
        a.src0 = gout;
        a.out0 = tens(wb + bp.d_pre0, 4 * ohw * 16, (long long)B * 4 * ohw * 16);
        a.mul = tens(wf + fp.p0, 4 * ohw * 16, (long long)B * 4 * ohw * 16);
        a.mul_mode = MUL_RELU_MASK;
        a.wpk = (const uint4*)(wb + bp.post2T);
        a.bpk = bpk0;
        a.ncb_out = 4;
        if ((rc = launch_conv<1, 0, 1, 3, STAGE_F32_1CH, EPI_ACT_CB8>(a, NI, 1, st))) return rc;
        if (side_aux) {                  // d_pre0 is ready for the gated post conv's weight gradient
            PLC_HIP(hipEventRecord(side_aux->sev[2], st));
            PLC_HIP(hipStreamWaitEvent(ss, side_aux->sev[2], 0));
        }

        in = conv_args(oH, oW, B);
        in.src0 = tens(wf + fp.fin, fimg, (long long)B * fimg);
        in.gate = tens(wf + fp.gate, fimg, 0);
        if ((rc = fast ? launch_wgrad<4, 0, 1, 3, STAGE_CB8_GATED>(in, tens(wb + bp.d_pre0, 4 * ohw * 16, (long long)B * 4 * ohw * 16),
                                                                   4, NI, slab, g->post0_w, g->post0_b, 32, R2, 0, ss)
                       : wgrad_chunked<4, 1, 3, STAGE_CB8_GATED>(in, R2, 0, tens(wb + bp.d_pre0, 4 * ohw * 16, (long long)B * 4 * ohw * 16),
                                                                 32, NI, slab, g->post0_w, g->post0_b, ss)))
            return rc;
        a = conv_args(oH, oW, B);
        a.src0 = tens(wb + bp.d_pre0, 4 * ohw * 16, (long long)B * 4 * ohw * 16);
        a.out0 = tens(wb + bp.d_feat, fimg, (long long)B * fimg);
        a.wpk = (const uint4*)(wb + bp.post0T);
        a.bpk = bpk0;
        a.ncb_out = C2 / 8;
        if ((rc = launch_conv<4, 0, 1, 3, STAGE_CB8, EPI_ACT_CB8>(a, NI, cdiv(C2, 32), st))) return rc;
    }
    // With no upsample block and no resize between the gates and the recurrence (scale 1), the gradient
    // of the gated feature map IS cell 2's incoming dh: its pointwise backward then forms
    // d_feat * gate itself (the gate is one image per batch element, cache-resident) and the
    // T*B-image product is neither written by the gate backward nor read back.
    const bool direct = d->n_blocks == 0 && bp.d_mid == (size_t)-1 && bp.d_up == (size_t)-1;
    // ---- gating and the two attention branches ------------------------------------------------
    {
        GateBwdArgs ga;
        ga.B = B; ga.T = T; ga.ncb = C2 / 8; ga.H = oH; ga.W = oW;
        ga.d_feat = wb + bp.d_feat;
        ga.feat_pre = wf + fp.fin;
        ga.gate = wf + fp.gate;
        ga.gate_dem = wf + fp.gate_dem;
        ga.gate_lu = wf + fp.gate_lu;
        ga.d_pre = direct ? nullptr : wb + bp.d_pre;
        ga.d_lu_pre = wb + bp.d_lu_pre;
        ga.d_dem_pre = wb + bp.d_dem_pre;
        hipLaunchKernelGGL(plc_gate_bwd_kernel, dim3(ew_grid((long long)B * (C2 / 8) * ohw)), dim3(256), 0, st, ga);
        PLC_HIP(hipGetLastError());

        struct Branch {
            size_t d_pre, d_mid, mid, hr, w2T;
            const float* w0; float *g0w, *g0b, *g2w, *g2b;
            int cin;
        };
        // The branches end in parameter gradients only (the aux maps need none): on the leaves' stream, after the
        // gate backward's outputs.
        hipStream_t st = ss;
        if (side_aux) {
            PLC_HIP(hipEventRecord(side_aux->sev[3], (hipStream_t)stream));
            PLC_HIP(hipStreamWaitEvent(ss, side_aux->sev[3], 0));
        }
        const Branch br[2] = {
            {bp.d_lu_pre, bp.d_lu_mid, fp.lu_mid, fp.lu_hr, bp.lu2T, d->lu0_w, g->lu0_w, g->lu0_b, g->lu2_w, g->lu2_b, L},
            {bp.d_dem_pre, bp.d_dem_mid, fp.dem_mid, fp.dem_hr, bp.dem2T, d->dem0_w, g->dem0_w, g->dem0_b, g->dem2_w, g->dem2_b, 1},
        };
        for (int i = 0; i < 2; ++i) {
            const Branch& q = br[i];
            // 1x1 conv (C2/2 -> C2): weight gradient, then data gradient masked by the ReLU of the 3x3 conv
            ConvArgs in = conv_args(oH, oW, B);
            in.src0 = tens(wf + q.mid, mcb * ohw * 16, 0);
            if ((rc = fast ? launch_wgrad<2, 0, 1, 1, STAGE_CB8>(in, tens(wb + q.d_pre, fimg, 0), C2 / 8, B, slab,
                                                                 q.g2w, q.g2b, R2, R2 / 2, 0, st)
                           : wgrad_chunked<4, 4, 1, STAGE_CB8>(in, R2 / 2, 0, tens(wb + q.d_pre, fimg, 0), R2, B, slab,
                                                               q.g2w, q.g2b, st)))
                return rc;
            ConvArgs a = conv_args(oH, oW, B);
            a.src0 = tens(wb + q.d_pre, fimg, 0);
            a.out0 = tens(wb + q.d_mid, mcb * ohw * 16, 0);
            a.mul = tens(wf + q.mid, mcb * ohw * 16, 0);
            a.mul_mode = MUL_RELU_MASK;
            a.wpk = (const uint4*)(wb + q.w2T);
            a.bpk = bpk0;
            a.ncb_out = mcb;
            if ((rc = fast ? launch_conv<4, 0, 1, 1, STAGE_CB8, EPI_ACT_CB8>(a, B, 1, st)
                           : launch_convg<1, STAGE_CB8, EPI_ACT_CB8>(a, C2 / 8, 0, cdiv(R2 / 2, 32), B, st)))
                return rc;
            // 3x3 conv (aux -> C2/2): weight gradient only (the aux maps need no gradient)
            in = conv_args(oH, oW, B);
            in.src0 = tens(wf + q.hr, cdiv(q.cin, 8) * ohw * 16, 0);
            // (more than 32 mid channels: one launch per 32-row group of dz)
            if (q.cin <= 8)
                rc = wgrad_chunked<1, 1, 3, STAGE_CB8>(in, q.cin, 0, tens(wb + q.d_mid, mcb * ohw * 16, 0), R2 / 2, B, slab,
                                                       q.g0w, q.g0b, st);
            else
                rc = wgrad_chunked<2, 1, 3, STAGE_CB8>(in, q.cin, 0, tens(wb + q.d_mid, mcb * ohw * 16, 0), R2 / 2, B, slab,
                                                       q.g0w, q.g0b, st);
            if (rc) return rc;
            // the aux map's own gradient, when asked for: through the 3x3 conv, then through the resize of
            // generator.py:143-153 (bilinear for the DEM, nearest for land use) back to the map's own grid
            float* const d_aux = i == 0 ? g->d_lu : g->d_dem;
            if (d_aux) {
                const int aH = i == 0 ? d->lu_H : d->dem_H, aW = i == 0 ? d->lu_W : d->dem_W;
                float* const hr = (float*)(wb + bp.d_aux_hr);
                hipLaunchKernelGGL(plc_aux_input_grad_kernel, dim3(ew_grid((long long)B * q.cin * ohw)), dim3(256), 0, st,
                                   (const char*)(wb + q.d_mid), q.w0, hr, B, R2 / 2, q.cin, oH, oW);
                hipLaunchKernelGGL(plc_resize_nchw_bwd_kernel, dim3(ew_grid((long long)B * q.cin * aH * aW)), dim3(256), 0,
                                   st, (const float*)hr, d_aux, (long long)B * q.cin, aH, aW, oH, oW, i == 0 ? 1 : 0,
                                   (float)aH / (float)oH, (float)aW / (float)oW);
                PLC_HIP(hipGetLastError());
            }
        }
    }
    // ---- PixelShuffle upsample blocks, last to first ------------------------------------------
    const char* cur = direct ? wb + bp.d_feat : wb + bp.d_pre;     // gradient of the feature map entering the gates
    // ---- bilinear feature resizes (generator.py:179-195), last to first -------------------------
    if (bp.d_mid != (size_t)-1) {
        const long long planes = (long long)NI * (C2 / 8);
        hipLaunchKernelGGL(plc_resize_cb8_bwd_kernel, dim3(ew_grid(planes * d->mid_H * d->mid_W)), dim3(256), 0, st,
                           (const uint4*)cur, (uint4*)(wb + bp.d_mid), planes, d->mid_H, d->mid_W, oH, oW,
                           d->out_scale_h, d->out_scale_w);
        PLC_HIP(hipGetLastError());
        cur = wb + bp.d_mid;
    }
    if (bp.d_up != (size_t)-1) {
        const long long planes = (long long)NI * (C2 / 8);
        hipLaunchKernelGGL(plc_resize_cb8_bwd_kernel, dim3(ew_grid(planes * fp.up_H * fp.up_W)), dim3(256), 0, st,
                           (const uint4*)cur, (uint4*)(wb + bp.d_up), planes, fp.up_H, fp.up_W, d->mid_H, d->mid_W,
                           d->mid_scale_h, d->mid_scale_w);
        PLC_HIP(hipGetLastError());
        cur = wb + bp.d_up;
    }
    for (int k = d->n_blocks - 1; k >= 0; --k) {
        const int kh = H << k, kw = W << k;
        const long long khw = (long long)kh * kw;
        const char* in_k = (k == 0) ? wf + fp.h2s + (size_t)B * fp.img2 : wf + fp.up_out[k - 1];
        hipLaunchKernelGGL(plc_pixel_unshuffle_mask_kernel, dim3(ew_grid((long long)NI * (C2 / 8) * khw)), dim3(256),
                           0, st, cur, (const char*)(wf + fp.up_out[k]), wb + bp.d_conv[k], NI, C2, kh, kw);
        PLC_HIP(hipGetLastError());
        if (side_aux) {                  // d_conv[k] is ready for the block's weight gradient (a leaf)
            PLC_HIP(hipEventRecord(side_aux->sev[3], st));
            PLC_HIP(hipStreamWaitEvent(ss, side_aux->sev[3], 0));
        }
        ConvArgs in = conv_args(kh, kw, NI);
        in.src0 = tens(in_k, (C2 / 8) * khw * 16, 0);
        if ((rc = fast ? launch_wgrad<4, 0, 4, 3, STAGE_CB8>(in, tens(wb + bp.d_conv[k], (4 * C2 / 8) * khw * 16, 0), 4 * C2 / 8,
                                                             NI, slab, g->up_w[k], g->up_b[k], 4 * R2, R2, 0, ss)
                       : wgrad_chunked<4, 4, 3, STAGE_CB8>(in, R2, 0, tens(wb + bp.d_conv[k], (4 * C2 / 8) * khw * 16, 0), 4 * R2,
                                                           NI, slab, g->up_w[k], g->up_b[k], ss)))
            return rc;
        ConvArgs a = conv_args(kh, kw, NI);
        a.src0 = tens(wb + bp.d_conv[k], (4 * C2 / 8) * khw * 16, 0);
        a.out0 = tens(wb + bp.d_in[k], (C2 / 8) * khw * 16, 0);
        a.wpk = (const uint4*)(wb + bp.upT[k]);
        a.bpk = bpk0;
        a.ncb_out = C2 / 8;
        if ((rc = fast ? launch_conv_th<16, 0, 1, 3, STAGE_CB8, EPI_ACT_CB8, 4, 2>(a, NI, 1, st)
                       : launch_convg<3, STAGE_CB8, EPI_ACT_CB8>(a, 4 * C2 / 8, 0, cdiv(C2, 32), NI, st)))
            return rc;
        cur = wb + bp.d_in[k];
    }
    // ---- backpropagation through time ---------------------------------------------------------
    const size_t s1 = (size_t)B * fp.img1, s2 = (size_t)B * fp.img2;      // bytes of one step (bf16 CB8)
    const size_t cat1 = (size_t)(2 * C1 / 8) * hw * 16, cat2 = (size_t)((C1 + C2) / 8) * hw * 16;
    const size_t plane = (size_t)hw * 16;
    // Per step t (downwards): pointwise backward of cell 2, its data-gradient conv (-> d[h1_t | h2_{t-1}]), then
    // the same for cell 1 (-> d[x_t | h1_{t-1}]).  Cell 2 of step t - 1 needs only cell 2's conv of step t, as
    // cell 1 of step t does, so with the widths the rolling-window convs are built for the loop is skewed:
    // one launch runs both pointwise passes, the next both convs (2T + 2 launches instead of 4T, and cell 1's
    // workgroups fill the CUs that cell 2's leave).  Same kernels' bodies, same results.
    auto pw2_args = [&](int t) {
        const bool last = (t == T - 1);
        const char* dcat2_n = wb + bp.dcat2 + (size_t)((t + 1) & 1) * B * cat2;
        return lstm_pointwise_bwd_args(view(cur + (size_t)t * s2, fp.img2),
                                       last ? view(nullptr, 0) : view(dcat2_n + (C1 / 8) * plane, cat2), view(nullptr, 0),
                                       last ? nullptr : wb + bp.dc2 + (size_t)((t + 1) & 1) * s2 * 2,
                                       wf + fp.gates2 + (size_t)t * s2 * 4,
                                       t == 0 ? nullptr : wf + fp.c2s + (size_t)t * s2 * 2,     // c_{-1} = 0 is not stored
                                       wf + fp.c2s + (size_t)(t + 1) * s2 * 2, wb + bp.dz2 + (size_t)t * s2 * 4,
                                       wb + bp.dc2 + (size_t)(t & 1) * s2 * 2, B, C2, H, W,
                                       direct ? view(wf + fp.gate, fp.img2) : view(nullptr, 0), 1);
    };
    auto pw1_args = [&](int t) {
        const bool last = (t == T - 1);
        const char* dcat1_n = wb + bp.dcat1 + (size_t)(t + 1) * B * cat1;
        return lstm_pointwise_bwd_args(view(wb + bp.dcat2 + (size_t)(t & 1) * B * cat2, cat2),
                                       last ? view(nullptr, 0) : view(dcat1_n + (C1 / 8) * plane, cat1), view(nullptr, 0),
                                       last ? nullptr : wb + bp.dc1 + (size_t)((t + 1) & 1) * s1 * 2,
                                       wf + fp.gates1 + (size_t)t * s1 * 4,
                                       t == 0 ? nullptr : wf + fp.c1s + (size_t)t * s1 * 2,
                                       wf + fp.c1s + (size_t)(t + 1) * s1 * 2, wb + bp.dz1 + (size_t)t * s1 * 4,
                                       wb + bp.dc1 + (size_t)(t & 1) * s1 * 2, B, C1, H, W, view(nullptr, 0), 1);
    };
    auto dg2_args = [&](int t) {
        return cell_dgrad_args(wb + bp.dz2 + (size_t)t * s2 * 4, wb + bp.cell2T, bpk0, wb + bp.dcat2 + (size_t)(t & 1) * B * cat2,
                               B, C1, C2, H, W);
    };
    auto dg1_args = [&](int t) {
        return cell_dgrad_args(wb + bp.dz1 + (size_t)t * s1 * 4, wb + bp.cell1T, bpk0, wb + bp.dcat1 + (size_t)t * B * cat1,
                               B, C1, C1, H, W);
    };
    const bool skew = fast && T > 1 && !plc_no_rolling() && !plc_no_pairing();
    static const bool no_split = [] { const char* e = getenv("PLC_NO_BWD_SPLIT"); return e && *e && *e != '0'; }();
    // (only where a launch is long enough to pay for the second queue: on the small recurrence grids the step is
    // launch-bound and the two chains cost 3 % (32 x 32, B = 4) to 15 % (15 x 12, B = 8) of the train step)
    if (skew && !no_split && !plc_fused_pw() && B >= 2 && (long long)B * hw >= (1ll << 17)) {
        // The batch elements are independent, and the recurrence alternates an HBM-bound launch (pointwise
        // passes) with an MFMA-bound one (data-gradient convs).  The two halves of the batch therefore run as
        // two chains on two streams, the second one launch behind the first: one half's pointwise passes
        // stream beside the other half's convs, which aim at one workgroup per CU (236 VGPRs: 276 of a SIMD's
        // 512 stay free, room for three pointwise waves).  Same kernels, same per-image arithmetic, identical
        // results; c3 / c2 / c4 train step -2.0 / -3.0 / -3.2 % on the same box.
        PlcAux* aux = nullptr;
        if ((rc = plc_aux(&aux))) return rc;
        hipEvent_t* const ev = aux->ev;
        hipStream_t sa = st, sb = aux->stream;
        auto shift_pw = [&](LstmBwdArgs a, int b0, int nb) {
            const long long uimg = (long long)a.ncb * a.H * a.W;      // units per image of an [B][ncb] tensor
            Cb8View* vs[4] = {&a.dh0, &a.dh1, &a.dh2, &a.dh0_mul};
            for (Cb8View* v : vs) if (v->p) v->p += (long long)b0 * v->img;
            const int dcb = a.dc_bf16 ? 16 : 32;
            if (a.dc_in) a.dc_in += b0 * uimg * dcb;
            a.dc_out += b0 * uimg * dcb;
            a.gates += b0 * 4 * uimg * 16;
            if (a.c_prev) a.c_prev += b0 * uimg * 32;
            a.dz += b0 * 4 * uimg * 16;
            a.B = nb;
            return a;
        };
        auto shift_cv = [&](ConvArgs a, int b0, int nb) {
            a.src0.p = (char*)a.src0.p + (long long)b0 * a.src0.si;
            a.out0.p = (char*)a.out0.p + (long long)b0 * a.out0.si;
            a.n_in = nb;
            return a;
        };
        const int nbh[2] = {B / 2, B - B / 2}, b0h[2] = {0, B / 2};
        hipStream_t sh[2] = {sa, sb};
        PLC_HIP(hipEventRecord(ev[0], sa));
        PLC_HIP(hipStreamWaitEvent(sb, ev[0], 0));
        forks.opened(sb, ev[2]);
        // stage k of a chain: 0 = pw2(T-1), 1 = dg2(T-1), then for t = T-1..1: pair pw, pair dg; then pw1(0), dg1(0)
        const int n_stage = 2 + 2 * (T - 1) + 2;
        auto run_stage = [&](int h, int k) -> int {
            const int b0 = b0h[h], nb = nbh[h];
            hipStream_t s = sh[h];
            if (k == 0) return lstm_pointwise_bwd_launch(shift_pw(pw2_args(T - 1), b0, nb), s);
            if (k == 1) {
                const ConvArgs c = shift_cv(dg2_args(T - 1), b0, nb);
                return launch_convr<16, 2, STAGE_CB8, EPI_ACT_CB8, 2, true>(c, nb, 1, s);
            }
            if (k >= n_stage - 2) {
                if (k == n_stage - 2) return lstm_pointwise_bwd_launch(shift_pw(pw1_args(0), b0, nb), s);
                const ConvArgs c = shift_cv(dg1_args(0), b0, nb);
                return launch_convr<8, 1, STAGE_CB8, EPI_ACT_CB8, 1, true>(c, nb, 1, s);
            }
            const int t = T - 1 - (k - 2) / 2;
            if ((k & 1) == 0) return lstm_pointwise_bwd_launch_pair(shift_pw(pw2_args(t - 1), b0, nb), shift_pw(pw1_args(t), b0, nb), s);
            return cell_dgrad_pair(shift_cv(dg2_args(t - 1), b0, nb), shift_cv(dg1_args(t), b0, nb), nb, s, 2);
        };
        // chain B starts one stage behind chain A
        if ((rc = run_stage(0, 0))) return rc;
        if (plc_debug_fail_after_fork()) return plc_fail(PLC_EHIP, "rollout backward: injected failure after the stream fork");
        PLC_HIP(hipEventRecord(ev[1], sa));
        PLC_HIP(hipStreamWaitEvent(sb, ev[1], 0));
        for (int k = 1; k <= n_stage; ++k) {
            if (k < n_stage && (rc = run_stage(0, k))) return rc;
            if ((rc = run_stage(1, k - 1))) return rc;
        }
        PLC_HIP(hipEventRecord(ev[2], sb));
        PLC_HIP(hipStreamWaitEvent(sa, ev[2], 0));
        forks.joined(sb);
    } else if (skew && plc_fused_pw()) {
        // T + 1 launches: each conv forms its own input (the pointwise pass runs inside it)
        if ((rc = cell_dgrad_pw<16, 2, 2>(pw2_args(T - 1), dg2_args(T - 1), B, st))) return rc;
        for (int t = T - 1; t >= 1; --t)
            if ((rc = cell_dgrad_pw_pair(pw2_args(t - 1), dg2_args(t - 1), pw1_args(t), dg1_args(t), B, st))) return rc;
        if ((rc = cell_dgrad_pw<8, 1, 1>(pw1_args(0), dg1_args(0), B, st))) return rc;
    } else if (skew) {
        if ((rc = lstm_pointwise_bwd_launch(pw2_args(T - 1), st))) return rc;
        if ((rc = cell_dgrad(wb + bp.dz2 + (size_t)(T - 1) * s2 * 4, wb + bp.cell2T, bpk0,
                             wb + bp.dcat2 + (size_t)((T - 1) & 1) * B * cat2, B, C1, C2, H, W, st)))
            return rc;
        for (int t = T - 1; t >= 1; --t) {
            if ((rc = lstm_pointwise_bwd_launch_pair(pw2_args(t - 1), pw1_args(t), st))) return rc;
            if ((rc = cell_dgrad_pair(dg2_args(t - 1), dg1_args(t), B, st))) return rc;
        }
        if ((rc = lstm_pointwise_bwd_launch(pw1_args(0), st))) return rc;
        if ((rc = cell_dgrad(wb + bp.dz1, wb + bp.cell1T, bpk0, wb + bp.dcat1, B, C1, C1, H, W, st))) return rc;
    } else {
        for (int t = T - 1; t >= 0; --t) {
            if ((rc = lstm_pointwise_bwd_launch(pw2_args(t), st))) return rc;
            if ((rc = cell_dgrad(wb + bp.dz2 + (size_t)t * s2 * 4, wb + bp.cell2T, bpk0, wb + bp.dcat2 + (size_t)(t & 1) * B * cat2,
                                 B, C1, C2, H, W, st)))
                return rc;
            if ((rc = lstm_pointwise_bwd_launch(pw1_args(t), st))) return rc;
            if ((rc = cell_dgrad(wb + bp.dz1 + (size_t)t * s1 * 4, wb + bp.cell1T, bpk0, wb + bp.dcat1 + (size_t)t * B * cat1,
                                 B, C1, C1, H, W, st)))
                return rc;
        }
    }
    // The three batched weight gradients that close the pass are independent of each other: the stem's and cell
    // 1's go to the side stream and fill the CUs that cell 2's persistent workgroups leave as they finish
    // (c3 train step -0.4 %).
    hipStream_t sw = st;
    if (side_aux) {
        PLC_HIP(hipEventRecord(side_aux->sev[3], st));           // every dz / dcat of the loop is written
        PLC_HIP(hipStreamWaitEvent(ss, side_aux->sev[3], 0));
        sw = ss;
    }
    // ---- stem: weight gradient; the gradient through its ReLU is a mask applied while loading dz ----
    {
        ConvArgs in = conv_args(H, W, B);
        in.src0 = tens(d->rain, (long long)T * hw * 4, hw * 4);
        in.row_tab = d->row_tab;
        in.col_tab = d->col_tab;
        const PlcTensor relu_of = tens(wf + fp.xs, fp.img1, (long long)B * fp.img1);      // stem output
        if ((rc = wgrad_chunked<1, 1, 3, STAGE_COORD>(in, 3, 0, tens(wb + bp.dcat1, cat1, (long long)B * cat1), R1, NI,
                                                      slab, g->init_w, g->init_b, sw, &relu_of)))
            return rc;
        if (g->d_rain) {
            hipLaunchKernelGGL(plc_stem_input_grad_kernel, dim3(ew_grid((long long)NI * hw)), dim3(256), 0, sw,
                               (const char*)(wb + bp.dcat1), (long long)cat1, (const char*)(wf + fp.xs), (long long)fp.img1,
                               d->init_w, g->d_rain, B, T, R1, H, W);
            PLC_HIP(hipGetLastError());
        }
    }
    // ---- gate-conv weight gradients, batched over all T*B frames -------------------------------
    if ((rc = cell_wgrad(wb + bp.dz2, tens(wf + fp.h1s + s1, fp.img1, (long long)B * fp.img1),
                         tens(wf + fp.h2s, fp.img2, (long long)B * fp.img2), B, NI, slab, g->cell2_w,
                         g->cell2_b, C1, C2, H, W, st, true, R1, R2)))
        return rc;
    if ((rc = cell_wgrad(wb + bp.dz1, tens(wf + fp.xs, fp.img1, (long long)B * fp.img1),
                         tens(wf + fp.h1s, fp.img1, (long long)B * fp.img1), B, NI, slab, g->cell1_w,
                         g->cell1_b, C1, C1, H, W, sw, true, R1, R1)))
        return rc;
    // ---- every conv's slabs -> parameter gradients, one launch ----------------------------------
    if (side_aux) {                      // join: every leaf has written its slabs
        PLC_HIP(hipEventRecord(side_aux->sev[5], ss));
        PLC_HIP(hipStreamWaitEvent(st, side_aux->sev[5], 0));
        forks.joined(ss);
    }
    return arena_flush(&arena, st);
}
