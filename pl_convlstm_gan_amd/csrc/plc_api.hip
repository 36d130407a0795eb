// C ABI of libplconvlstm (see include/plconvlstm.h): argument checking, workspace carving and
// kernel launches.  No allocation, no synchronisation; the only device state is the fixed staging
// array of plc_convlstm_cell_fwd_noalloc.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "../../include/plconvlstm.h"
#include "plc_conv_kernel.h"
#include "plc_convg_kernel.h"
#include "plc_convr_kernel.h"
#include "plc_post_kernel.h"
#include "plc_misc_kernels.h"
#include "plc_fp8_kernels.h"

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int plc_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define PLC_HIP(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return plc_fail(PLC_EHIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

#define PLC_REQUIRE(cond, ...)                              \
    do {                                                    \
        if (!(cond)) return plc_fail(PLC_EINVAL, __VA_ARGS__); \
    } while (0)

static inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }
static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
// Channel counts inside the engine are whole 8-channel blocks: a width that is not a multiple of 8 is carried
// with zero channels up to the next multiple (zero weights and biases keep them zero through every layer: a
// padding channel of a ConvLSTM cell has i = f = o = 1/2, g = 0, hence c = h = 0 for ever).
static inline int pad8(int c) { return (c + 7) & ~7; }

extern "C" int plc_version(void) { return PLC_VERSION; }
extern "C" const char* plc_last_error(void) { return g_err; }

static inline int ew_grid(long long total) {
    long long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

// ---------------------------------------------------------------------------------------------
// layout conversion
// ---------------------------------------------------------------------------------------------
extern "C" int plc_nchw_f32_to_cb8_bf16(const float* src, void* dst, int N, int C, int H, int W, void* stream) {
    PLC_REQUIRE(src && dst && N > 0 && C > 0 && H > 0 && W > 0, "plc_nchw_f32_to_cb8_bf16: bad argument");
    PLC_REQUIRE(aligned16(dst), "plc_nchw_f32_to_cb8_bf16: dst must be 16-byte aligned");
    const long long total = (long long)N * cdiv(C, 8) * H * W;
    hipLaunchKernelGGL(plc_nchw_to_cb8_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream,
                       src, (uint4*)dst, N, C, H, W);
    PLC_HIP(hipGetLastError());
    return PLC_OK;
}

extern "C" int plc_cb8_bf16_to_nchw_f32(const void* src, float* dst, int N, int C, int H, int W, void* stream) {
    PLC_REQUIRE(src && dst && N > 0 && C > 0 && H > 0 && W > 0, "plc_cb8_bf16_to_nchw_f32: bad argument");
    PLC_REQUIRE(aligned16(src), "plc_cb8_bf16_to_nchw_f32: src must be 16-byte aligned");
    const long long total = (long long)N * cdiv(C, 8) * H * W;
    hipLaunchKernelGGL(plc_cb8_to_nchw_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream,
                       (const uint4*)src, dst, N, C, H, W);
    PLC_HIP(hipGetLastError());
    return PLC_OK;
}

static int state_layout(const float* src, float* dst, int N, int C, int H, int W, int dir, void* stream,
                        const char* who) {
    PLC_REQUIRE(src && dst && N > 0 && C > 0 && H > 0 && W > 0, "%s: bad argument", who);
    const long long total = (long long)N * pad8(C) * H * W;
    hipLaunchKernelGGL(plc_state_layout_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream,
                       src, dst, N, C, H, W, dir);
    PLC_HIP(hipGetLastError());
    return PLC_OK;
}
extern "C" int plc_nchw_f32_to_cb8_f32(const float* src, float* dst, int N, int C, int H, int W, void* stream) {
    return state_layout(src, dst, N, C, H, W, 0, stream, "plc_nchw_f32_to_cb8_f32");
}
extern "C" int plc_cb8_f32_to_nchw_f32(const float* src, float* dst, int N, int C, int H, int W, void* stream) {
    return state_layout(src, dst, N, C, H, W, 1, stream, "plc_cb8_f32_to_nchw_f32");
}

// ---------------------------------------------------------------------------------------------
// weight packing
// ---------------------------------------------------------------------------------------------
static inline int pack_ncb(int c0, int c1) { return cdiv(c0, 8) + cdiv(c1, 8); }
// Widths with weight-stationary kernel instantiations (plc_conv_kernel.h); every other width that is a
// multiple of 8 takes the run-time-width kernel (plc_convg_kernel.h).
static inline bool cell_fast(int Cin, int Ch) { return Cin == 16 && (Ch == 16 || Ch == 32); }
static inline bool widths_fast(int C1, int C2) { return C1 == 16 && C2 == 32; }
static inline int pack_nks(int c0, int c1, int ks) { return (ks * ks * pack_ncb(c0, c1) + 1) / 2; }
// number of 32-row M tiles; LSTM and PixelShuffle convs regroup rows but keep the count
static inline int pack_nmt(int cout) { return cdiv(cout, 32); }

// (the chunk-major order of the run-time-width kernel pads every chunk to whole k-steps, so it never
// has fewer k-steps than the plain order: buffers are sized for it and serve both)
extern "C" size_t plc_packed_weight_bytes(int cout, int c0, int c1, int ks) {
    if (cout <= 0 || c0 <= 0 || c1 < 0 || ks <= 0) return 0;
    const int nks = plc_g_nks(pack_ncb(c0, c1), ks);
    return align256((size_t)pack_nmt(cout) * (nks > pack_nks(c0, c1, ks) ? nks : pack_nks(c0, c1, ks)) * 64 * 16);
}
extern "C" size_t plc_packed_bias_bytes(int cout) {
    if (cout <= 0) return 0;
    return align256((size_t)pack_nmt(cout) * 32 * sizeof(float));
}

static int pack_blocks(long long elems) {
    const long long b = (elems + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 64 ? 64 : b));
}

// Forward packing job (rows = output channels regrouped per `mode`).
// generic != 0: chunk-major K order for the run-time-width kernel.
static PackJob pack_job(const float* w, const float* bias, void* wpk, float* bpk, int cout, int c0, int c1, int ks,
                        int mode, int generic = 0) {
    PackJob jb;
    memset(&jb, 0, sizeof(jb));
    jb.w = w; jb.bias = bias; jb.wpk = (unsigned short*)wpk; jb.bpk = bpk;
    jb.cout = cout; jb.c0 = c0; jb.c1 = c1; jb.ks = ks; jb.mode = mode;
    jb.n_mt = pack_nmt(cout);
    jb.nks_w = generic ? plc_g_nks(pack_ncb(c0, c1), ks) : pack_nks(c0, c1, ks);
    jb.chunk = generic ? PLC_G_CHUNK : 0;
    jb.ch_per_gate = cout / 4; jb.transposed = 0; jb.ksplit = 1;
    jb.n_blocks = pack_blocks((long long)jb.n_mt * jb.nks_w * 512);
    return jb;
}

// Jobs collected for one launch of plc_pack_multi_kernel.
struct PackBatch {
    PackTable tb;
    int blocks;
};
static void pack_batch_begin(PackBatch* b) {
    b->tb.n = 0;
    b->blocks = 0;
}
static int pack_batch_flush(PackBatch* b, hipStream_t st) {
    if (b->tb.n == 0) return PLC_OK;
    hipLaunchKernelGGL(plc_pack_multi_kernel, dim3(b->blocks), dim3(256), 0, st, b->tb);
    PLC_HIP(hipGetLastError());
    pack_batch_begin(b);
    return PLC_OK;
}
static int pack_batch_add(PackBatch* b, PackJob jb, hipStream_t st) {
    if (b->tb.n == PACK_MAX_JOBS) {
        int rc = pack_batch_flush(b, st);
        if (rc) return rc;
    }
    jb.first_block = b->blocks;
    b->blocks += jb.n_blocks;
    b->tb.job[b->tb.n++] = jb;
    return PLC_OK;
}

extern "C" int plc_pack_conv_weight(const float* w, const float* bias, void* wpk, float* bpk,
                                    int cout, int c0, int c1, int ks, int mode, void* stream) {
    PLC_REQUIRE(w && wpk && bpk, "plc_pack_conv_weight: null pointer");
    PLC_REQUIRE(cout > 0 && c0 > 0 && c1 >= 0 && (ks == 1 || ks == 3), "plc_pack_conv_weight: bad shape");
    PLC_REQUIRE(mode >= 0 && mode <= 2, "plc_pack_conv_weight: bad mode %d", mode);
    if (mode != PLC_PACK_PLAIN) PLC_REQUIRE(cout % 4 == 0, "plc_pack_conv_weight: cout %d must be a multiple of 4 for mode %d", cout, mode);
    PLC_REQUIRE(aligned16(wpk) && aligned16(bpk), "plc_pack_conv_weight: packed buffers must be 16-byte aligned");
    // a ConvLSTM gate conv of a width without a weight-stationary instantiation runs on the
    // run-time-width kernel, which wants its own K order (plc_convlstm_cell_fwd_cb8 makes the same choice)
    const PackJob jb = pack_job(w, bias, wpk, bpk, cout, c0, c1, ks, mode, mode == PLC_PACK_LSTM && !cell_fast(pad8(c0), pad8(c1)));
    hipLaunchKernelGGL(plc_pack_weight_kernel, dim3(jb.n_blocks), dim3(256), 0, (hipStream_t)stream, jb);
    PLC_HIP(hipGetLastError());
    return PLC_OK;
}

// Data-gradient packing: rows = channels of the block-padded concat [c0 | c1], K = the cout output
// channels split over `ksplit` waves.
static inline int packT_nmt(int c0, int c1) { return cdiv(8 * pack_ncb(c0, c1), 32); }
static inline int packT_nks(int cout, int ks, int ksplit) { return (ks * ks * (cdiv(cout, 8) / ksplit) + 1) / 2; }
// K blocks of a data-gradient conv: those of its dz tensor.  An LSTM gate conv's dz is gate-major with every gate
// padded to whole blocks (lstm_k = its real hidden channels); k_blocks > 0 overrides (a dz wider than the conv's outputs).
static inline int packT_kblocks(int cout, int lstm_k, int k_blocks) {
    return k_blocks > 0 ? k_blocks : (lstm_k > 0 ? 4 * cdiv(lstm_k, 8) : cdiv(cout, 8));
}
static size_t packedT_weight_bytes(int cout, int c0, int c1, int ks, int ksplit) {
    const size_t plain = (size_t)ksplit * packT_nks(cout, ks, ksplit), chunked = (size_t)plc_g_nks(cdiv(cout, 8), ks);
    return align256((size_t)packT_nmt(c0, c1) * (plain > chunked ? plain : chunked) * 64 * 16);
}
// (cout, c0, c1: the conv's REAL channel counts)
static PackJob pack_job_T(const float* w, void* wpk, float* bpk, int cout, int c0, int c1, int ks, int ksplit,
                          int generic = 0, int lstm_k = 0, int k_blocks = 0) {
    PackJob jb;
    memset(&jb, 0, sizeof(jb));
    jb.w = w; jb.wpk = (unsigned short*)wpk; jb.bpk = bpk;
    jb.cout = cout; jb.c0 = c0; jb.c1 = c1; jb.ks = ks;
    jb.lstm_k = lstm_k;
    jb.k_blocks = packT_kblocks(cout, lstm_k, k_blocks);
    if (generic) ksplit = 1;
    jb.n_mt = packT_nmt(c0, c1);
    jb.nks_w = generic ? plc_g_nks(jb.k_blocks, ks) : packT_nks(8 * jb.k_blocks, ks, ksplit);
    jb.chunk = generic ? PLC_G_CHUNK : 0;
    jb.transposed = 1; jb.ksplit = ksplit;
    jb.n_blocks = pack_blocks((long long)jb.n_mt * ksplit * jb.nks_w * 512);
    return jb;
}
static int pack_conv_weight_T(const float* w, void* wpk, float* bpk, int cout, int c0, int c1, int ks, int ksplit,
                              hipStream_t st, int generic = 0, int lstm_k = 0) {
    const PackJob jb = pack_job_T(w, wpk, bpk, cout, c0, c1, ks, ksplit, generic, lstm_k);
    hipLaunchKernelGGL(plc_pack_weight_kernel, dim3(jb.n_blocks), dim3(256), 0, st, jb);
    PLC_HIP(hipGetLastError());
    return PLC_OK;
}

// ---------------------------------------------------------------------------------------------
// conv launch
// ---------------------------------------------------------------------------------------------
static PlcTensor tens(const void* p, long long si, long long so) {
    PlcTensor t;
    t.p = (char*)p;
    t.si = si;
    t.so = so;
    return t;
}

// Host-side caches of immutable per-device facts (the only process-wide state the library keeps;
// entries are written once with the same value by whichever thread gets there first).
static int current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
    return dev;
}

// Number of CUs of the current device.
static int cu_count() {
    static int cached[64] = {0};
    const int dev = current_device();
    if (dev < 0) return 256;
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev] = n;
    }
    return cached[dev];
}

// Raises a kernel's dynamic-LDS limit once per (kernel, device) instead of on every launch; `done` is
// the per-instantiation flag array of the caller.
static int ensure_lds_limit(const void* kern, int lds, bool (&done)[64]) {
    const int dev = current_device();
    if (dev >= 0 && done[dev]) return PLC_OK;
    PLC_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    if (dev >= 0) done[dev] = true;
    return PLC_OK;
}

// A second stream per device and the three events of a fork / join around it, created on first use and
// kept for the life of the process: the rollouts run the two halves of a batch as two chains of launches,
// one on the caller's stream and one here (batch elements are independent), so that one half's
// memory-bound phases run beside the other half's arithmetic.  Fork and join are event record / wait
// pairs, which a stream capture follows.
// `side` is a third stream for work nothing in the recurrences waits for: the forward of the two attention
// branches (t-invariant) runs beside the stem and the first cell launches; in the backward pass the weight
// gradients of the two post convs and the branches' whole backward run beside the data-gradient chain and
// the BPTT loop, and are joined before the slab reduction.
// The streams and events are shared by every rollout call on the device, and a wait takes the event's latest
// record AT THE TIME OF THE CALL: the rollout entry points therefore hold plc_aux_mutex while they enqueue, so
// that calls from several host threads cannot pair one call's wait with another call's record.
static std::mutex plc_aux_mutex;
struct PlcAux {
    hipStream_t stream, side;
    hipEvent_t ev[3], sev[6];
};
static int plc_aux(PlcAux** out) {
    static PlcAux aux[64];
    static bool made[64] = {false};
    const int dev = current_device();
    if (dev < 0) return plc_fail(PLC_EHIP, "no current device");
    if (!made[dev]) {
        PLC_HIP(hipStreamCreateWithFlags(&aux[dev].stream, hipStreamNonBlocking));
        PLC_HIP(hipStreamCreateWithFlags(&aux[dev].side, hipStreamNonBlocking));
        for (int k = 0; k < 3; ++k) PLC_HIP(hipEventCreateWithFlags(&aux[dev].ev[k], hipEventDisableTiming));
        for (int k = 0; k < 6; ++k) PLC_HIP(hipEventCreateWithFlags(&aux[dev].sev[k], hipEventDisableTiming));
        made[dev] = true;
    }
    *out = &aux[dev];
    return PLC_OK;
}

// Joins, on every way out of a rollout call, the auxiliary streams the call has forked onto.  A fork is an event
// recorded on the caller's stream and awaited by the auxiliary one; if the call then fails half way (a launch
// error, a full slab arena) and simply returned, work on the auxiliary stream would still be reading and
// writing the caller's workspace after the caller has been told the call is over, and under stream capture
// the capture would end with un-joined work.  opened() notes a fork and the event of its join, joined() that
// the regular join has been issued; the destructor issues the join of whatever is still open.
struct ForkGuard {
    hipStream_t main;
    hipStream_t forked[2];
    hipEvent_t join_ev[2];
    int n;
    explicit ForkGuard(hipStream_t m) : main(m), n(0) {}
    void opened(hipStream_t s, hipEvent_t join) {
        for (int k = 0; k < n; ++k) if (forked[k] == s) return;
        if (n < 2) { forked[n] = s; join_ev[n] = join; ++n; }
    }
    void joined(hipStream_t s) {
        for (int k = 0; k < n; ++k)
            if (forked[k] == s) { forked[k] = forked[n - 1]; join_ev[k] = join_ev[n - 1]; --n; return; }
    }
    ~ForkGuard() {
        for (int k = 0; k < n; ++k)
            if (hipEventRecord(join_ev[k], forked[k]) == hipSuccess) (void)hipStreamWaitEvent(main, join_ev[k], 0);
    }
    ForkGuard(const ForkGuard&) = delete;
    ForkGuard& operator=(const ForkGuard&) = delete;
};

// Test hook: PLC_DEBUG_FAIL_AFTER_FORK=1 in the environment makes the rollout calls fail right after they have
// forked onto an auxiliary stream, which is the only way to reach ForkGuard's destructor path on purpose.
static bool plc_debug_fail_after_fork() {
    static const bool on = [] { const char* e = getenv("PLC_DEBUG_FAIL_AFTER_FORK"); return e && *e && *e != '0'; }();
    return on;
}

// One instantiation at a fixed tile height TH (and K split).
template <int NCB0, int NCB1, int NMT, int KS, int STAGE, int EPI, int TH, int KSPLIT = 1>
static int launch_conv_th(ConvArgs a, int n_img, int n_mgroups, hipStream_t st) {
    constexpr int lds = plc_conv_lds_bytes<NCB0 + NCB1, KS, TH, KSPLIT, STAGE>();
    auto kern = plc_conv_mfma_kernel<NCB0, NCB1, NMT, KS, STAGE, EPI, TH, KSPLIT>;
    if (lds > 65536) {
        static bool lds_raised[64] = {false};
        int rc = ensure_lds_limit((const void*)kern, lds, lds_raised);
        if (rc) return rc;
    }
    a.tiles_x = cdiv(a.W, 32);
    a.tiles_y = cdiv(a.H, TH);
    a.n_tiles = n_img * a.tiles_x * a.tiles_y;
    // as many workgroups as stay resident (register- and LDS-limited), each walking an equal share of tiles
    constexpr int by_waves = plc_conv_min_waves<NCB0 + NCB1, KS, KSPLIT, NMT, EPI>();
    constexpr int by_lds = (160 * 1024) / lds;
    const int resident = cu_count() * (by_waves < by_lds ? by_waves : by_lds) / n_mgroups;
    const int per_wg = plc_conv_multi_tile<STAGE, EPI>() ? cdiv(a.n_tiles, resident > 0 ? resident : 1) : 1;
    a.n_blocks = cdiv(a.n_tiles, per_wg);
    hipLaunchKernelGGL(kern, dim3(a.n_blocks, n_mgroups), dim3(256), lds, st, a);
    PLC_HIP(hipGetLastError());
    return PLC_OK;
}

// The gated conv over T time steps of B images each (images n = t * B + b, multiplier tensor indexed by b
// alone): tiles are walked time-innermost, so the multiplier tile of a (b, position) is fetched once per
// chain of time steps instead of once per tile.  A position's T steps are cut into the fewest chains that
// still give every resident workgroup slot a chain (PLC_GATED_TSEGS overrides the count: measurements).
// Results do not depend on the tile order.
template <int NCB0, int NMT, int KS, int EPI, int TH>
static int launch_conv_gated_tinner(ConvArgs a, int B, int T, hipStream_t st) {
    constexpr int lds = plc_conv_lds_bytes<NCB0, KS, TH, 1, STAGE_CB8_GATED>();
    auto kern = plc_conv_mfma_kernel<NCB0, 0, NMT, KS, STAGE_CB8_GATED, EPI, TH, 1>;
    if (lds > 65536) {
        static bool lds_raised[64] = {false};
        int rc = ensure_lds_limit((const void*)kern, lds, lds_raised);
        if (rc) return rc;
    }
    a.n_in = B;
    a.tiles_x = cdiv(a.W, 32);
    a.tiles_y = cdiv(a.H, TH);
    const long long positions = (long long)B * a.tiles_x * a.tiles_y;
    if (positions * T >= (1ll << 31)) return plc_fail(PLC_EINVAL, "conv: bad tile count");
    a.n_tiles = (int)(positions * T);
    constexpr int by_waves = plc_conv_min_waves<NCB0, KS, 1, NMT, EPI>();
    constexpr int by_lds = (160 * 1024) / lds;
    const int resident = cu_count() * (by_waves < by_lds ? by_waves : by_lds);
    static const int forced = [] { const char* e = getenv("PLC_GATED_TSEGS"); return e ? atoi(e) : 0; }();
    int segs = T;
    for (int k = 1; k <= T; ++k)
        if (T % k == 0 && (positions * k >= 2ll * resident || k == T)) { segs = k; break; }
    if (forced > 0 && T % forced == 0) segs = forced;
    a.t_segs = segs;
    a.t_inner = T / segs;
    const long long chains = positions * segs;
    const long long slots = resident > 0 ? resident : 1;
    const long long per_wg = (chains + slots - 1) / slots;
    a.n_blocks = (int)((chains + per_wg - 1) / per_wg);
    hipLaunchKernelGGL(kern, dim3(a.n_blocks, 1), dim3(256), lds, st, a);
    PLC_HIP(hipGetLastError());
    return PLC_OK;
}

// 16-row tiles when that still yields at least two workgroups per CU, else 8-row tiles.
template <int NCB0, int NCB1, int NMT, int KS, int STAGE, int EPI>
static int launch_conv(ConvArgs a, int n_img, int n_mgroups, hipStream_t st) {
    const long long t16 = (long long)n_img * cdiv(a.H, 16) * cdiv(a.W, 32) * n_mgroups;
    // (single-M-tile convs whose staging is not plain DMA were measured faster with 8-row tiles)
    if (t16 >= 512 && !(NMT == 1 && STAGE != STAGE_CB8)) return launch_conv_th<NCB0, NCB1, NMT, KS, STAGE, EPI, 16>(a, n_img, n_mgroups, st);
    return launch_conv_th<NCB0, NCB1, NMT, KS, STAGE, EPI, 8>(a, n_img, n_mgroups, st);
}

// Whether the attention branches of a rollout with `out_px` = B * out_H * out_W output pixels per time step go
// to the side stream.  Below ~1e5 pixels a step is bound by launches, not by the GPU, and the fork / join only
// adds to that (15 x 12 -> 120 x 96, B = 8: +3 %; 64 x 64, B = 2: +15 % on the train step); above, the branches'
// small launches disappear behind the recurrence (256 x 256, B = 4: forward -2.5 %, train step -1.5 %; x8
// upsampling to 256 x 256: forward -9 %).  PLC_NO_SIDE_STREAM=1 keeps them on the caller's stream (measurements).
static bool plc_use_side_stream(long long out_px) {
    static const bool off = [] { const char* e = getenv("PLC_NO_SIDE_STREAM"); return e && *e && *e != '0'; }();
    return !off && out_px >= (1ll << 17);
}

// PLC_NO_POST_FUSION=1 keeps the generator's two output convs in launches of their own (results are bit-identical).
static bool plc_no_post_fusion() {
    static const bool off = [] { const char* e = getenv("PLC_NO_POST_FUSION"); return e && *e && *e != '0'; }();
    return off;
}

// Debugging / measurement switch: PLC_NO_ROLLING=1 keeps the tile-at-a-time kernels where a rolling-window
// instantiation exists (results are bit-identical either way).
static bool plc_no_rolling() {
    static const bool off = [] { const char* e = getenv("PLC_NO_ROLLING"); return e && *e && *e != '0'; }();
    return off;
}

// The rolling-window kernel (plc_convr_kernel.h): the image is cut into 32-column strips and each strip
// into segments of seg_rows rows, one workgroup per segment, as many segments as workgroups stay
// resident (a single round) when the image is large enough for that.
// Segment geometry of a rolling-window launch into `a`; returns its grid size, or an error code (negative).
template <int NCB, int NMT, int STAGE, int EPI, int KSPLIT, bool PLAIN>
static int convr_geometry(ConvArgs& a, int n_img, int n_mgroups, int slots_div = 1) {
    using R = PlcRoll<NCB, NMT, KSPLIT>;
    constexpr int lds = plc_convr_lds_bytes<NCB, NMT, KSPLIT, STAGE>();
    if (PLAIN && (EPI != EPI_ACT_CB8 || a.out1.p || a.mul.p || a.act == ACT_SIGMOID))
        return plc_fail(PLC_EINVAL, "conv: plain epilogue requested with a second output, multiplier or sigmoid");
    // (the kernel addresses a lane's unit of an input row with a 32-bit offset from the row's base)
    if ((long long)NCB * a.H * a.W * 16 >= (1ll << 32)) return plc_fail(PLC_EINVAL, "conv: image too large for the rolling-window kernel");
    constexpr int by_lds = (160 * 1024) / lds;
    // (slots_div: the launch is meant to share the chip with another one and aims at that fraction of the slots)
    const int resident = cu_count() * (by_lds < 2 ? by_lds : 2) / n_mgroups / slots_div;
    a.tiles_x = cdiv(a.W, 32);
    const long long strips = (long long)n_img * a.tiles_x;
    const long long want = strips < resident ? resident / strips : 1;     // segments per strip
    int seg_rows = cdiv(cdiv(a.H, (int)want), R::A) * R::A;
    if (seg_rows < 2 * R::A) seg_rows = 2 * R::A;
    a.seg_rows = seg_rows;
    a.tiles_y = cdiv(a.H, seg_rows);
    const long long segs = strips * a.tiles_y;
    if (segs <= 0 || segs >= (1ll << 31)) return plc_fail(PLC_EINVAL, "conv: bad segment count %lld", segs);
    a.n_tiles = (int)segs;
    a.n_blocks = segs < resident ? (int)segs : resident;
    return a.n_blocks;
}

template <int NCB, int NMT, int STAGE, int EPI, int KSPLIT, bool PLAIN = false>
static int launch_convr(ConvArgs a, int n_img, int n_mgroups, hipStream_t st) {
    constexpr int lds = plc_convr_lds_bytes<NCB, NMT, KSPLIT, STAGE>();
    auto kern = plc_convr_kernel<NCB, NMT, STAGE, EPI, KSPLIT, PLAIN>;
    if (lds > 65536) {
        static bool lds_raised[64] = {false};
        int rc = ensure_lds_limit((const void*)kern, lds, lds_raised);
        if (rc) return rc;
    }
    const int grid = convr_geometry<NCB, NMT, STAGE, EPI, KSPLIT, PLAIN>(a, n_img, n_mgroups);
    if (grid < 0) return grid;
    hipLaunchKernelGGL(kern, dim3(grid, n_mgroups), dim3(256), lds, st, a);
    PLC_HIP(hipGetLastError());
    return PLC_OK;
}

// The run-time-width kernel: ncb0 / ncb1 input channel blocks from src0 / src1, n_mt 32-row output tiles.
template <int KS, int STAGE, int EPI>
static int launch_convg(ConvArgs a, int ncb0, int ncb1, int n_mt, int n_img, hipStream_t st) {
    if (ncb0 <= 0 || ncb1 < 0 || n_mt <= 0) return plc_fail(PLC_EINVAL, "conv: bad channel counts");
    a.g_ncb0 = ncb0;
    a.g_ncb1 = ncb1;
    a.g_nmt = n_mt;
    a.tiles_x = cdiv(a.W, 32);
    a.tiles_y = cdiv(a.H, PLC_G_TH);
    const long long tiles = (long long)n_img * a.tiles_x * a.tiles_y;
    if (tiles <= 0 || tiles >= (1ll << 31)) return plc_fail(PLC_EINVAL, "conv: bad tile count %lld", tiles);
    a.n_tiles = (int)tiles;
    a.n_blocks = a.n_tiles;
    hipLaunchKernelGGL((plc_convg_kernel<KS, STAGE, EPI>), dim3(a.n_tiles, cdiv(n_mt, 4)), dim3(256),
                       plc_convg_lds_bytes<KS>(), st, a);
    PLC_HIP(hipGetLastError());
    return PLC_OK;
}

static ConvArgs conv_args(int H, int W, int n_in) {
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.H = H;
    a.W = W;
    a.n_in = n_in;
    return a;
}

// ---------------------------------------------------------------------------------------------
// ConvLSTM cell
// ---------------------------------------------------------------------------------------------
static ConvArgs cell_conv_args(const void* x, const void* h, const float* c_in, const void* wpk, const float* bpk,
                               void* h_out, float* c_out, void* gates, int B, int Cin, int Ch, int H, int W) {
    const long long hw = (long long)H * W;
    ConvArgs a = conv_args(H, W, B);
    a.src0 = tens(x, (Cin / 8) * hw * 16, 0);
    a.src1 = tens(h, (Ch / 8) * hw * 16, 0);
    a.c_in = tens(c_in, (Ch / 8) * hw * 32, 0);
    a.out0 = tens(h_out, (Ch / 8) * hw * 16, 0);
    a.out1 = tens(c_out, (Ch / 8) * hw * 32, 0);
    a.out2 = tens(gates, (4 * Ch / 8) * hw * 16, 0);
    a.save_gates = gates != nullptr;
    a.wpk = (const uint4*)wpk;
    a.bpk = bpk;
    return a;
}

// PLC_NO_PERSISTENT=1 in the environment keeps the recurrence on per-step launches (A/B timing, tests).
static bool plc_no_persistent() {
    const char* e = getenv("PLC_NO_PERSISTENT");
    return e && e[0] && e[0] != '0';
}

// PLC_NO_PAIRING=1 keeps cell 1 and cell 2 in launches of their own (A/B timing, tests).
static bool plc_no_pairing() {
    const char* e = getenv("PLC_NO_PAIRING");
    return e && e[0] && e[0] != '0';
}

// One ConvLSTM cell launch; a.save_gates selects the variant that stores i,f,o,g for the backward pass.
static int launch_cell(const ConvArgs& a, int B, int Cin, int Ch, hipStream_t st) {
    const bool gates = a.save_gates != 0;
    if (Cin == 16 && Ch == 16)
        return gates ? launch_conv<2, 2, 2, 3, STAGE_CB8, EPI_LSTM_SAVE>(a, B, 1, st)
                     : launch_conv<2, 2, 2, 3, STAGE_CB8, EPI_LSTM>(a, B, 1, st);
    if (Cin == 16 && Ch == 32)
        return gates ? launch_conv<2, 4, 4, 3, STAGE_CB8, EPI_LSTM_SAVE>(a, B, 1, st)
                     : launch_conv<2, 4, 4, 3, STAGE_CB8, EPI_LSTM>(a, B, 1, st);
    if (Cin % 8 == 0 && Ch % 8 == 0)
        return gates ? launch_convg<3, STAGE_CB8, EPI_LSTM_SAVE>(a, Cin / 8, Ch / 8, Ch / 8, B, st)
                     : launch_convg<3, STAGE_CB8, EPI_LSTM>(a, Cin / 8, Ch / 8, Ch / 8, B, st);
    return plc_fail(PLC_EUNSUPPORTED, "ConvLSTM cell (Cin=%d, Ch=%d): channel counts must be multiples of 8", Cin, Ch);
}

// ---------------------------------------------------------------------------------------------
// The recurrence of a whole rollout in ONE launch (small grids, where a cell launch is mostly fixed
// cost: dispatch, argument and weight fetch, pipeline fill and drain).
// ---------------------------------------------------------------------------------------------
// Every workgroup owns one tile of one of the two cells for the whole rollout.  Phase p runs cell 1 of
// step p next to cell 2 of step p - 1 (cell 2 of a step needs h1 of the same step, cell 1 of the next
// step only h1 / c1 of this one), so T steps take T + 1 phases instead of 2 T dependent launches.
// Between phases all workgroups meet at a counter in global memory: the arriving thread releases its
// workgroup's stores at agent scope (they leave this XCD's L2), the leaving thread acquires (stale
// lines are dropped), so the halo of the next phase's tile sees its neighbours' h.  Every workgroup
// passes every barrier exactly once whether or not it has work in the phase, and the host launches the
// kernel only when the whole grid is resident at once, so the wait always ends.
// The cells' tensors are described once, with the byte stride of one time step as the outer stride of
// every PlcTensor: image index t * B + b then addresses step t, and since tiles are numbered
// image-major a workgroup reaches step t of its own tile by adding t * (tiles of one step) to its
// tile number -- which is exactly the tile loop of plc_conv_body in its PERSIST mode.
struct PersistArgs {
    ConvArgs c1, c2;            // cell 1 / cell 2 with zero_state = 1 (it applies to the first step only)
    int n1;                     // tiles (= workgroups) of one step of cell 1; the rest of the grid is cell 2
};

template <int TH, bool SAVE>
__global__ __launch_bounds__(256, 1) void plc_lstm_persistent_kernel(const PersistArgs p) {
    extern __shared__ __attribute__((aligned(16))) uint4 plc_persist_tile[];
    constexpr int EPI = SAVE ? EPI_LSTM_SAVE : EPI_LSTM;
    // each branch keeps its cell's weights in registers for the whole rollout
    if ((int)blockIdx.x < p.n1)
        plc_conv_body<2, 2, 2, 3, STAGE_CB8, EPI, TH, 1, 1>(p.c1, blockIdx.x, 0, 1, plc_persist_tile);
    else
        plc_conv_body<2, 4, 4, 3, STAGE_CB8, EPI, TH, 1, 2>(p.c2, blockIdx.x - p.n1, 0, 1, plc_persist_tile);
}

// Launches the recurrence as one persistent kernel when both cells' tiles fit the chip at once (and
// the widths are the instantiated [16,32]); *done tells the caller whether it did.
template <int TH, bool SAVE>
static int launch_persistent_th(PersistArgs p, int n_img, int T, hipStream_t st, bool* done) {
    constexpr int lds1 = plc_conv_lds_bytes<4, 3, TH, 1, STAGE_CB8>(), lds2 = plc_conv_lds_bytes<6, 3, TH, 1, STAGE_CB8>();
    constexpr int lds = lds1 > lds2 ? lds1 : lds2;
    auto kern = plc_lstm_persistent_kernel<TH, SAVE>;
    if (lds > 65536) {
        static bool lds_raised[64] = {false};
        int rc = ensure_lds_limit((const void*)kern, lds, lds_raised);
        if (rc) return rc;
    }
    for (ConvArgs* a : {&p.c1, &p.c2}) {
        a->tiles_x = cdiv(a->W, 32);
        a->tiles_y = cdiv(a->H, TH);
        const long long per_step = (long long)n_img * a->tiles_x * a->tiles_y;
        if (per_step * T >= (1ll << 31)) return PLC_OK;
        a->n_blocks = (int)per_step;                 // the tile loop's stride: the same tile, one step on
        a->n_tiles = (int)(per_step * T);            // tile numbers run over all steps
    }
    p.n1 = p.c1.n_blocks;
    const int grid = p.c1.n_blocks + p.c2.n_blocks;
    // (resident workgroups per CU of this instantiation: an immutable fact, asked once per device)
    static int per_cu_cached[64] = {0};
    const int dev = current_device();
    int per_cu = dev >= 0 ? per_cu_cached[dev] : 0;
    if (per_cu == 0) {
        PLC_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kern, 256, lds));
        if (dev >= 0 && per_cu > 0) per_cu_cached[dev] = per_cu;
    }
    // Only grids that take at most a quarter of the chip's workgroup slots: the whole grid must be
    // resident for the barrier to complete, and a second process (or stream) doing the same on this GPU
    // must still find room for ITS whole grid.  Larger grids have launches long enough not to need this.
    if (per_cu <= 0 || grid > per_cu * cu_count() / 4) return PLC_OK;
    hipLaunchKernelGGL(plc_zero_words_kernel, dim3(1), dim3(64), 0, st, (unsigned*)p.c1.sync, 1);
    PLC_HIP(hipGetLastError());
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, p);
    PLC_HIP(hipGetLastError());
    *done = true;
    return PLC_OK;
}

// 8-row tiles: the grids this path accepts are small (see the residency rule above), so the finer tiling
// is what spreads them over the chip.
static int launch_persistent(const PersistArgs& p, int n_img, int T, bool save, hipStream_t st, bool* done) {
    *done = false;
    return save ? launch_persistent_th<8, true>(p, n_img, T, st, done) : launch_persistent_th<8, false>(p, n_img, T, st, done);
}

// ---------------------------------------------------------------------------------------------
// Cell 2 of step t and cell 1 of step t + 1 in one launch (they do not depend on each other: both read
// h1 of step t).  The workgroups of cell 2 come FIRST in the grid and fill the chip; as they retire the
// workgroups of cell 1 take their places, so cell 1's weight and halo-tile fetches run under the tail of
// cell 2's arithmetic instead of on an idle chip, and the next launch's cell 2 does the same under the
// tail of cell 1.  (Interleaving the two kinds so that every CU starts with one of each made them run
// in lockstep and gained nothing; the order is what staggers them.)
// ---------------------------------------------------------------------------------------------
struct PairArgs {
    ConvArgs c2, c1;
    int n2;                     // workgroups (= tiles) of cell 2; the rest of the grid is cell 1
};

template <int TH, bool SAVE>
__global__ __launch_bounds__(256, 2) void plc_lstm_pair_kernel(const PairArgs p) {
    extern __shared__ __attribute__((aligned(16))) uint4 plc_pair_tile[];
    constexpr int EPI = SAVE ? EPI_LSTM_SAVE : EPI_LSTM;
    if ((int)blockIdx.x < p.n2)
        plc_conv_body<2, 4, 4, 3, STAGE_CB8, EPI, TH>(p.c2, blockIdx.x, 0, 1, plc_pair_tile);
    else
        plc_conv_body<2, 2, 2, 3, STAGE_CB8, EPI, TH>(p.c1, blockIdx.x - p.n2, 0, 1, plc_pair_tile);
}

template <int TH, bool SAVE>
static int launch_pair_th(PairArgs p, int n_img, hipStream_t st) {
    constexpr int lds1 = plc_conv_lds_bytes<4, 3, TH, 1, STAGE_CB8>(), lds2 = plc_conv_lds_bytes<6, 3, TH, 1, STAGE_CB8>();
    constexpr int lds = lds1 > lds2 ? lds1 : lds2;
    auto kern = plc_lstm_pair_kernel<TH, SAVE>;
    if (lds > 65536) {
        static bool lds_raised[64] = {false};
        int rc = ensure_lds_limit((const void*)kern, lds, lds_raised);
        if (rc) return rc;
    }
    for (ConvArgs* a : {&p.c2, &p.c1}) {
        a->tiles_x = cdiv(a->W, 32);
        a->tiles_y = cdiv(a->H, TH);
        a->n_tiles = n_img * a->tiles_x * a->tiles_y;
        a->n_blocks = a->n_tiles;
    }
    p.n2 = p.c2.n_tiles;
    hipLaunchKernelGGL(kern, dim3(p.c2.n_tiles + p.c1.n_tiles), dim3(256), lds, st, p);
    PLC_HIP(hipGetLastError());
    return PLC_OK;
}

// widths [16,32] only; the tile height follows launch_conv's rule so that results equal the single launches'
// (n_rule: the image count the tile-height rule is applied to -- a half-batch chain keeps the whole batch's tiles)
static int launch_pair(const ConvArgs& c2, const ConvArgs& c1, int n_img, hipStream_t st, int n_rule = 0) {
    PairArgs p;
    memset(&p, 0, sizeof(p));
    p.c2 = c2;
    p.c1 = c1;
    const bool save = c2.save_gates != 0;
    const long long t16 = (long long)(n_rule > 0 ? n_rule : n_img) * cdiv(c2.H, 16) * cdiv(c2.W, 32);
    if (t16 >= 512) return save ? launch_pair_th<16, true>(p, n_img, st) : launch_pair_th<16, false>(p, n_img, st);
    return save ? launch_pair_th<8, true>(p, n_img, st) : launch_pair_th<8, false>(p, n_img, st);
}

static int cell_fwd_cb8(const void* x, const void* h, const float* c_in, const void* wpk, const float* bpk,
                        void* h_out, float* c_out, void* gates, int B, int Cin, int Ch, int H, int W,
                        hipStream_t st) {
    return launch_cell(cell_conv_args(x, h, c_in, wpk, bpk, h_out, c_out, gates, B, Cin, Ch, H, W), B, Cin, Ch, st);
}

extern "C" int plc_convlstm_cell_fwd_cb8(const void* x, const void* h, const float* c_in, const void* wpk,
                                         const float* bpk, void* h_out, float* c_out, void* gates, int B, int Cin,
                                         int Ch, int H, int W, void* stream) {
    PLC_REQUIRE(x && h && c_in && wpk && bpk && h_out && c_out, "plc_convlstm_cell_fwd_cb8: null pointer");
    PLC_REQUIRE(B > 0 && H > 0 && W > 0, "plc_convlstm_cell_fwd_cb8: non-positive size");
    PLC_REQUIRE(aligned16(x) && aligned16(h) && aligned16(c_in) && aligned16(wpk) && aligned16(bpk) &&
                    aligned16(h_out) && aligned16(c_out) && aligned16(gates),
                "plc_convlstm_cell_fwd_cb8: pointers must be 16-byte aligned");
    return cell_fwd_cb8(x, h, c_in, wpk, bpk, h_out, c_out, gates, B, Cin, Ch, H, W, (hipStream_t)stream);
}

struct CellWs {
    size_t x, h, c, wpk, bpk, ho, co, total;
};
static CellWs cell_ws(int B, int Cin, int Ch, int H, int W) {
    CellWs s;
    const size_t hw = (size_t)H * W;
    size_t o = 0;
    s.x = o; o += align256((size_t)B * cdiv(Cin, 8) * hw * 16);
    s.h = o; o += align256((size_t)B * cdiv(Ch, 8) * hw * 16);
    s.c = o; o += align256((size_t)B * cdiv(Ch, 8) * hw * 32);
    s.wpk = o; o += plc_packed_weight_bytes(4 * Ch, Cin, Ch, 3);
    s.bpk = o; o += plc_packed_bias_bytes(4 * Ch);
    s.ho = o; o += align256((size_t)B * cdiv(Ch, 8) * hw * 16);
    s.co = o; o += align256((size_t)B * cdiv(Ch, 8) * hw * 32);
    s.total = o;
    return s;
}

extern "C" size_t plc_convlstm_cell_workspace_bytes(int B, int Cin, int Ch, int H, int W) {
    if (B <= 0 || Cin <= 0 || Ch <= 0 || H <= 0 || W <= 0) return 0;
    return cell_ws(B, Cin, Ch, H, W).total;
}

extern "C" int plc_convlstm_cell_fwd(const float* x, const float* h, const float* c, const float* weight,
                                     const float* bias, float* h_next, float* c_next, void* workspace,
                                     size_t workspace_bytes, int B, int Cin, int Ch, int H, int W, void* stream) {
    PLC_REQUIRE(x && h && c && weight && h_next && c_next && workspace, "plc_convlstm_cell_fwd: null pointer");
    PLC_REQUIRE(B > 0 && Cin > 0 && Ch > 0 && H > 0 && W > 0, "plc_convlstm_cell_fwd: non-positive size");
    PLC_REQUIRE(((uintptr_t)workspace & 255) == 0, "plc_convlstm_cell_fwd: workspace must be 256-byte aligned");
    const CellWs s = cell_ws(B, Cin, Ch, H, W);
    if (workspace_bytes < s.total)
        return plc_fail(PLC_EWORKSPACE, "plc_convlstm_cell_fwd: workspace %zu < %zu bytes", workspace_bytes, s.total);
    char* ws = (char*)workspace;
    int rc;
    if ((rc = plc_nchw_f32_to_cb8_bf16(x, ws + s.x, B, Cin, H, W, stream))) return rc;
    if ((rc = plc_nchw_f32_to_cb8_bf16(h, ws + s.h, B, Ch, H, W, stream))) return rc;
    if ((rc = plc_nchw_f32_to_cb8_f32(c, (float*)(ws + s.c), B, Ch, H, W, stream))) return rc;
    if ((rc = plc_pack_conv_weight(weight, bias, ws + s.wpk, (float*)(ws + s.bpk), 4 * Ch, Cin, Ch, 3,
                                   PLC_PACK_LSTM, stream)))
        return rc;
    // (widths that are not multiples of 8 run as the next multiple: the conversions above padded with zeros, the
    // packed weights and biases of the padding channels are zero)
    if ((rc = cell_fwd_cb8(ws + s.x, ws + s.h, (float*)(ws + s.c), ws + s.wpk, (float*)(ws + s.bpk), ws + s.ho,
                           (float*)(ws + s.co), nullptr, B, pad8(Cin), pad8(Ch), H, W, (hipStream_t)stream)))
        return rc;
    if ((rc = plc_cb8_bf16_to_nchw_f32(ws + s.ho, h_next, B, Ch, H, W, stream))) return rc;
    if ((rc = plc_cb8_f32_to_nchw_f32((float*)(ws + s.co), c_next, B, Ch, H, W, stream))) return rc;
    return PLC_OK;
}

// ---------------------------------------------------------------------------------------------
// fp8 (e4m3) gate convolution
// ---------------------------------------------------------------------------------------------
static inline int fp8_nks(int Cin, int Ch) { return (9 * ((Cin + Ch) / 16) + 3) / 4; }
static inline bool fp8_supported(int Cin, int Ch) { return Cin == 16 && (Ch == 16 || Ch == 32); }

extern "C" size_t plc_fp8_packed_weight_bytes(int Cin, int Ch) {
    if (!fp8_supported(Cin, Ch)) return 0;
    return align256((size_t)(Ch / 8) * fp8_nks(Cin, Ch) * 64 * 32);
}

static int pack_lstm_fp8_pair(const float* w1, void* wpk1, int* sw1, int Cin1, int Ch1, const float* w2, void* wpk2,
                              int* sw2, int Cin2, int Ch2, hipStream_t st) {
    PackFp8Job j0 = {w1, (unsigned*)wpk1, sw1, Cin1, Ch1};
    PackFp8Job j1 = {w2, (unsigned*)wpk2, sw2, Cin2, Ch2};
    hipLaunchKernelGGL(plc_pack_lstm_fp8_kernel, dim3((w2 ? 2 : 1) * PLC_FP8_PACK_BLOCKS), dim3(1024), 0, st, j0, j1);
    PLC_HIP(hipGetLastError());
    return PLC_OK;
}

extern "C" int plc_pack_lstm_weight_fp8(const float* w, void* wpk8, int* scale_word, int Cin, int Ch, void* stream) {
    PLC_REQUIRE(w && wpk8 && scale_word, "plc_pack_lstm_weight_fp8: null pointer");
    PLC_REQUIRE(aligned16(wpk8), "plc_pack_lstm_weight_fp8: packed buffer must be 16-byte aligned");
    if (!fp8_supported(Cin, Ch))
        return plc_fail(PLC_EUNSUPPORTED, "fp8 gate conv (Cin=%d, Ch=%d) is compiled for {(16,16), (16,32)} only", Cin, Ch);
    return pack_lstm_fp8_pair(w, wpk8, scale_word, Cin, Ch, nullptr, nullptr, nullptr, 0, 0, (hipStream_t)stream);
}

extern "C" int plc_cb8_bf16_to_cb16_fp8(const void* src_cb8, void* dst_cb16, int N, int C, int H, int W, void* stream) {
    PLC_REQUIRE(src_cb8 && dst_cb16 && N > 0 && C > 0 && H > 0 && W > 0, "plc_cb8_bf16_to_cb16_fp8: bad argument");
    PLC_REQUIRE(aligned16(src_cb8) && aligned16(dst_cb16), "plc_cb8_bf16_to_cb16_fp8: pointers must be 16-byte aligned");
    const int ncb = cdiv(C, 8);
    const long long total = (long long)N * cdiv(ncb, 2) * H * W;
    hipLaunchKernelGGL(plc_cb8_to_cb16_fp8_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream,
                       (const uint4*)src_cb8, (uint4*)dst_cb16, (long long)N, ncb, (long long)H * W);
    PLC_HIP(hipGetLastError());
    return PLC_OK;
}

template <int NU0, int NU1, int NMT, int TH>
static int launch_cell_fp8_th(Fp8CellArgs fa, int B, hipStream_t st) {
    fa.c.tiles_x = cdiv(fa.c.W, 32);
    fa.c.tiles_y = cdiv(fa.c.H, TH);
    fa.c.n_tiles = B * fa.c.tiles_x * fa.c.tiles_y;
    fa.c.n_blocks = fa.c.n_tiles;
    constexpr int lds = plc_fp8_lds_bytes<NU0, NU1, NMT, TH>();
    static_assert(lds <= 65536, "fp8 cell tile exceeds the default dynamic LDS limit");
    if (fa.c.save_gates)
        hipLaunchKernelGGL((plc_lstm_fp8_kernel<NU0, NU1, NMT, TH, true>), dim3(fa.c.n_tiles), dim3(256), lds, st, fa);
    else
        hipLaunchKernelGGL((plc_lstm_fp8_kernel<NU0, NU1, NMT, TH, false>), dim3(fa.c.n_tiles), dim3(256), lds, st, fa);
    PLC_HIP(hipGetLastError());
    return PLC_OK;
}

// a = the bf16 cell's arguments with src0 / src1 pointing at the fp8 CB16 copies of x / h and wpk at the fp8 fragments
static int launch_cell_fp8(const ConvArgs& a, void* h8_out, const int* scale_word, int B, int Cin, int Ch, hipStream_t st) {
    Fp8CellArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.c = a;
    const long long hw = (long long)a.H * a.W;
    fa.c.src0 = tens(a.src0.p, (Cin / 16) * hw * 16, 0);
    fa.c.src1 = tens(a.src1.p, (Ch / 16) * hw * 16, 0);
    fa.h8_out = tens(h8_out, (Ch / 16) * hw * 16, 0);
    fa.scale_word = scale_word;
    const long long t16 = (long long)B * cdiv(a.H, 16) * cdiv(a.W, 32);
    if (Cin == 16 && Ch == 16)
        return t16 >= 512 ? launch_cell_fp8_th<1, 1, 2, 16>(fa, B, st) : launch_cell_fp8_th<1, 1, 2, 8>(fa, B, st);
    if (Cin == 16 && Ch == 32)
        return t16 >= 512 ? launch_cell_fp8_th<1, 2, 4, 16>(fa, B, st) : launch_cell_fp8_th<1, 2, 4, 8>(fa, B, st);
    return plc_fail(PLC_EUNSUPPORTED, "fp8 gate conv (Cin=%d, Ch=%d) is compiled for {(16,16), (16,32)} only", Cin, Ch);
}

extern "C" int plc_convlstm_cell_fwd_fp8(const void* x8, const void* h8, const float* c_in, const void* wpk8,
                                         const int* scale_word, const float* bpk, void* h_out, void* h8_out,
                                         float* c_out, void* gates, int B, int Cin, int Ch, int H, int W, void* stream) {
    PLC_REQUIRE(x8 && h8 && c_in && wpk8 && scale_word && bpk && h_out && h8_out && c_out,
                "plc_convlstm_cell_fwd_fp8: null pointer");
    PLC_REQUIRE(B > 0 && H > 0 && W > 0, "plc_convlstm_cell_fwd_fp8: non-positive size");
    PLC_REQUIRE(aligned16(x8) && aligned16(h8) && aligned16(c_in) && aligned16(wpk8) && aligned16(bpk) &&
                    aligned16(h_out) && aligned16(h8_out) && aligned16(c_out) && aligned16(gates),
                "plc_convlstm_cell_fwd_fp8: pointers must be 16-byte aligned");
    const ConvArgs a = cell_conv_args(x8, h8, c_in, wpk8, bpk, h_out, c_out, gates, B, Cin, Ch, H, W);
    return launch_cell_fp8(a, h8_out, scale_word, B, Cin, Ch, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// ConvLSTM cell forward for callers that cannot allocate: the staging area is a fixed array in the
// library's own device image instead of a caller workspace.
// ---------------------------------------------------------------------------------------------
// One copy of this array exists on every GPU the library runs on, for the life of the process.  In
// host code the NAME of a __device__ array is only a handle for the runtime, not a device address:
// its address on the current device comes from hipGetSymbolAddress, is looked up once per device and
// is what every launch below receives.
static __device__ __align__(256) unsigned char plc_static_scratch[PLC_STATIC_SCRATCH_BYTES];
static void* g_static_scratch_ptr[64] = {nullptr};   // per device, written once

static int static_scratch_ptr(void** out) {
    const int dev = current_device();
    if (dev < 0) return plc_fail(PLC_EHIP, "plc_convlstm_cell_fwd_noalloc: no current device (or device index >= 64)");
    if (!g_static_scratch_ptr[dev]) {
        void* p = nullptr;
        PLC_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(plc_static_scratch)));
        if (!p) return plc_fail(PLC_EHIP, "plc_convlstm_cell_fwd_noalloc: the scratch array has no address on device %d", dev);
        g_static_scratch_ptr[dev] = p;
    }
    *out = g_static_scratch_ptr[dev];
    return PLC_OK;
}

extern "C" size_t plc_convlstm_cell_noalloc_max_bytes(void) { return (size_t)PLC_STATIC_SCRATCH_BYTES; }

extern "C" int plc_convlstm_cell_fwd_noalloc(const float* x, const float* h, const float* c, const float* weight,
                                             const float* bias, float* h_next, float* c_next, int B, int Cin,
                                             int Ch, int H, int W, void* stream) {
    PLC_REQUIRE(x && h && c && weight && h_next && c_next, "plc_convlstm_cell_fwd_noalloc: null pointer");
    PLC_REQUIRE(B > 0 && Cin > 0 && Ch > 0 && H > 0 && W > 0, "plc_convlstm_cell_fwd_noalloc: non-positive size");
    const size_t need = cell_ws(B, Cin, Ch, H, W).total;
    if (need > (size_t)PLC_STATIC_SCRATCH_BYTES)
        return plc_fail(PLC_EWORKSPACE,
                        "plc_convlstm_cell_fwd_noalloc: B=%d, Cin=%d, Ch=%d, %dx%d needs %zu bytes of staging, the "
                        "built-in area holds %zu; use plc_convlstm_cell_fwd with a caller workspace",
                        B, Cin, Ch, H, W, need, (size_t)PLC_STATIC_SCRATCH_BYTES);
    void* scratch = nullptr;
    int rc = static_scratch_ptr(&scratch);
    if (rc) return rc;
    // Every byte of the staging area that a kernel reads is written earlier in this same call
    // (layout conversions, weight packing, the cell's outputs), so nothing has to start at zero and
    // the previous call's contents are simply overwritten.
    return plc_convlstm_cell_fwd(x, h, c, weight, bias, h_next, c_next, scratch, (size_t)PLC_STATIC_SCRATCH_BYTES,
                                 B, Cin, Ch, H, W, stream);
}

// ---------------------------------------------------------------------------------------------
// Generator rollout
// ---------------------------------------------------------------------------------------------
struct Packed {
    size_t w, b;
};
struct RolloutPlan {
    // packed parameters
    Packed init, cell1, cell2, dem0, dem2, lu0, lu2, up[8], post0, post2;
    // activations
    size_t xs, h1s, c1s, h2s, c2s, gates1, gates2;
    size_t up_out[8];   // output of upsample block k
    size_t mid, fin;    // after residual resize / after target resize (== previous stage if unused)
    size_t dem_hr, lu_hr, dem_mid, lu_mid, gate_dem, gate_lu, gate, p0;
    size_t sync;        // the persistent recurrence kernel's barrier counter
    // fp8 gate convs (desc.gate_fp8): fp8 copies of the cells' operands, fp8 weight fragments, scale words
    size_t xs8, h1s8, h2s8, cell1_w8, cell2_w8, scale8;
    size_t total;
    // derived sizes
    int up_H, up_W;     // size after the PixelShuffle blocks
    size_t img1, img2;  // bytes of one CB8 bf16 image of C1 / C2 channels at LR size
};

static int rollout_plan(const plc_rollout_desc* d, RolloutPlan* p) {
    PLC_REQUIRE(d, "rollout: null descriptor");
    PLC_REQUIRE(d->B > 0 && d->T > 0 && d->H > 0 && d->W > 0, "rollout: non-positive B/T/H/W");
    PLC_REQUIRE(d->L > 0 && d->L <= 16, "rollout: lu_channels=%d unsupported (1..16)", d->L);
    PLC_REQUIRE(d->n_blocks >= 0 && d->n_blocks <= 8, "rollout: n_blocks=%d out of range", d->n_blocks);
    if (d->C1 <= 0 || d->C2 < 2 || d->C1 > 512 || d->C2 > 512)
        return plc_fail(PLC_EUNSUPPORTED, "rollout: hidden_dims=[%d,%d]: the first must be 1..512, the second 2..512",
                        d->C1, d->C2);
    PLC_REQUIRE(d->dem_H > 0 && d->dem_W > 0 && d->lu_H > 0 && d->lu_W > 0, "rollout: bad aux size");
    // C1, C2: the widths as the engine carries them (whole 8-channel blocks); R1, R2: the model's own
    const int C1 = pad8(d->C1), C2 = pad8(d->C2), R1 = d->C1, R2 = d->C2, B = d->B, T = d->T;
    const size_t hw = (size_t)d->H * d->W;
    p->up_H = d->H << d->n_blocks;
    p->up_W = d->W << d->n_blocks;
    PLC_REQUIRE(d->mid_H > 0 && d->mid_W > 0 && d->out_H > 0 && d->out_W > 0, "rollout: bad mid/out size");
    const size_t ohw = (size_t)d->out_H * d->out_W;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += align256(bytes); return at; };
    auto pk = [&](int cout, int c0, int c1, int ks) {
        Packed q;
        q.w = take(plc_packed_weight_bytes(cout, c0, c1, ks));
        q.b = take(plc_packed_bias_bytes(cout));
        return q;
    };
    p->init = pk(R1, 3, 0, 3);
    p->cell1 = pk(4 * R1, R1, R1, 3);
    p->cell2 = pk(4 * R2, R1, R2, 3);
    p->dem0 = pk(R2 / 2, 1, 0, 3);
    p->dem2 = pk(R2, R2 / 2, 0, 1);
    p->lu0 = pk(R2 / 2, d->L, 0, 3);
    p->lu2 = pk(R2, R2 / 2, 0, 1);
    for (int k = 0; k < d->n_blocks; ++k) p->up[k] = pk(4 * R2, R2, 0, 3);
    p->post0 = pk(32, R2, 0, 3);
    p->post2 = pk(1, 32, 0, 3);
    p->img1 = (size_t)(C1 / 8) * hw * 16;
    p->img2 = (size_t)(C2 / 8) * hw * 16;
    p->xs = take((size_t)T * B * p->img1);
    p->h1s = take((size_t)(T + 1) * B * p->img1);
    p->c1s = take((size_t)(T + 1) * B * p->img1 * 2);
    p->h2s = take((size_t)(T + 1) * B * p->img2);
    p->c2s = take((size_t)(T + 1) * B * p->img2 * 2);
    if (d->save_for_backward) {
        p->gates1 = take((size_t)T * B * p->img1 * 4);
        p->gates2 = take((size_t)T * B * p->img2 * 4);
    } else {
        p->gates1 = p->gates2 = (size_t)-1;
    }
    size_t cur = p->h2s + (size_t)B * p->img2;  // h2_0 .. h2_{T-1}
    for (int k = 0; k < d->n_blocks; ++k) {
        const size_t s = (size_t)(d->H << (k + 1)) * (d->W << (k + 1));
        p->up_out[k] = take((size_t)T * B * (C2 / 8) * s * 16);
        cur = p->up_out[k];
    }
    if (d->mid_H != p->up_H || d->mid_W != p->up_W) {
        p->mid = take((size_t)T * B * (C2 / 8) * d->mid_H * d->mid_W * 16);
        cur = p->mid;
    } else p->mid = cur;
    if (d->out_H != d->mid_H || d->out_W != d->mid_W) {
        p->fin = take((size_t)T * B * (C2 / 8) * ohw * 16);
        cur = p->fin;
    } else p->fin = cur;
    p->dem_hr = take((size_t)B * 1 * ohw * 16);
    p->lu_hr = take((size_t)B * cdiv(d->L, 8) * ohw * 16);
    p->dem_mid = take((size_t)B * cdiv(R2 / 2, 8) * ohw * 16);
    p->lu_mid = take((size_t)B * cdiv(R2 / 2, 8) * ohw * 16);
    p->gate_dem = take((size_t)B * (C2 / 8) * ohw * 16);
    p->gate_lu = d->save_for_backward ? take((size_t)B * (C2 / 8) * ohw * 16) : (size_t)-1;
    p->gate = take((size_t)B * (C2 / 8) * ohw * 16);
    p->p0 = take((size_t)T * B * 4 * ohw * 16);
    p->sync = take(256);
    if (d->gate_fp8) {
        if (!widths_fast(R1, R2))
            return plc_fail(PLC_EUNSUPPORTED, "rollout: the fp8 gate conv is compiled for hidden_dims=[16,32] only (got [%d,%d])", R1, R2);
        p->xs8 = take((size_t)T * B * (C1 / 16) * hw * 16);
        p->h1s8 = take((size_t)(T + 1) * B * (C1 / 16) * hw * 16);
        p->h2s8 = take((size_t)(T + 1) * B * (C2 / 16) * hw * 16);
        p->cell1_w8 = take(plc_fp8_packed_weight_bytes(C1, C1));
        p->cell2_w8 = take(plc_fp8_packed_weight_bytes(C1, C2));
        p->scale8 = take(256);
    } else {
        p->xs8 = p->h1s8 = p->h2s8 = p->cell1_w8 = p->cell2_w8 = p->scale8 = (size_t)-1;
    }
    p->total = o;
    return PLC_OK;
}

extern "C" size_t plc_generator_workspace_bytes(const plc_rollout_desc* d) {
    RolloutPlan p;
    if (rollout_plan(d, &p) != PLC_OK) return 0;
    return p.total;
}

extern "C" size_t plc_generator_workspace_offset(const plc_rollout_desc* d, const char* name) {
    RolloutPlan p;
    if (!name || rollout_plan(d, &p) != PLC_OK) return (size_t)-1;
    if (!strcmp(name, "xs")) return p.xs;
    if (!strcmp(name, "h1s")) return p.h1s;
    if (!strcmp(name, "c1s")) return p.c1s;
    if (!strcmp(name, "h2s")) return p.h2s;
    if (!strcmp(name, "c2s")) return p.c2s;
    if (!strcmp(name, "gates1")) return p.gates1;
    if (!strcmp(name, "gates2")) return p.gates2;
    if (!strcmp(name, "gate")) return p.gate;
    if (!strcmp(name, "p0")) return p.p0;
    if (!strcmp(name, "feat")) return p.fin;
    return (size_t)-1;
}

extern "C" int plc_generator_rollout_fwd(const plc_rollout_desc* d, void* stream) {
    std::lock_guard<std::mutex> aux_lock(plc_aux_mutex);
    RolloutPlan p;
    int rc = rollout_plan(d, &p);
    if (rc) return rc;
    PLC_REQUIRE(d->rain && d->dem && d->lu && d->row_tab && d->col_tab && d->out && d->workspace,
                "rollout: null tensor pointer");
    PLC_REQUIRE(d->init_w && d->init_b && d->cell1_w && d->cell1_b && d->cell2_w && d->cell2_b && d->dem0_w &&
                    d->dem0_b && d->dem2_w && d->dem2_b && d->lu0_w && d->lu0_b && d->lu2_w && d->lu2_b &&
                    d->post0_w && d->post0_b && d->post2_w && d->post2_b,
                "rollout: null parameter pointer");
    for (int k = 0; k < d->n_blocks; ++k) PLC_REQUIRE(d->up_w[k] && d->up_b[k], "rollout: null upsample parameter %d", k);
    PLC_REQUIRE(((uintptr_t)d->workspace & 255) == 0, "rollout: workspace must be 256-byte aligned");
    if (d->workspace_bytes < p.total)
        return plc_fail(PLC_EWORKSPACE, "rollout: workspace %zu < %zu bytes", d->workspace_bytes, p.total);

    hipStream_t st = (hipStream_t)stream;
    ForkGuard forks(st);         // joins the auxiliary streams on an early (error) return too
    char* ws = (char*)d->workspace;
    // C1, C2: the widths as whole 8-channel blocks (what every buffer and kernel sees); R1, R2: the model's own
    // widths (what the weight packers see: the padding rows and columns they emit are zero)
    const int B = d->B, T = d->T, H = d->H, W = d->W, C1 = pad8(d->C1), C2 = pad8(d->C2), R1 = d->C1, R2 = d->C2, L = d->L;
    const int oH = d->out_H, oW = d->out_W;
    const long long hw = (long long)H * W, ohw = (long long)oH * oW;

    // ---- parameters -> MFMA fragment order ----------------------------------------------
    PackBatch batch;
    pack_batch_begin(&batch);
    // [16,32] has weight-stationary instantiations for every conv; other widths send the convs whose
    // shape depends on the widths through the run-time-width kernel (gen = its K order).  The stem, the
    // aux 3x3 convs and the last conv have fixed input widths and use the instantiated kernels always.
    const bool fast = widths_fast(C1, C2);
    const int gen = fast ? 0 : 1;
    auto pack = [&](const float* w, const float* b, const Packed& q, int cout, int c0, int c1, int ks, int mode, int g) {
        return pack_batch_add(&batch, pack_job(w, b, ws + q.w, (float*)(ws + q.b), cout, c0, c1, ks, mode, g), st);
    };
    if ((rc = pack(d->init_w, d->init_b, p.init, R1, 3, 0, 3, PLC_PACK_PLAIN, 0))) return rc;
    if ((rc = pack(d->cell1_w, d->cell1_b, p.cell1, 4 * R1, R1, R1, 3, PLC_PACK_LSTM, !cell_fast(C1, C1)))) return rc;
    if ((rc = pack(d->cell2_w, d->cell2_b, p.cell2, 4 * R2, R1, R2, 3, PLC_PACK_LSTM, !cell_fast(C1, C2)))) return rc;
    if ((rc = pack(d->dem0_w, d->dem0_b, p.dem0, R2 / 2, 1, 0, 3, PLC_PACK_PLAIN, 0))) return rc;
    if ((rc = pack(d->dem2_w, d->dem2_b, p.dem2, R2, R2 / 2, 0, 1, PLC_PACK_PLAIN, gen))) return rc;
    if ((rc = pack(d->lu0_w, d->lu0_b, p.lu0, R2 / 2, L, 0, 3, PLC_PACK_PLAIN, 0))) return rc;
    if ((rc = pack(d->lu2_w, d->lu2_b, p.lu2, R2, R2 / 2, 0, 1, PLC_PACK_PLAIN, gen))) return rc;
    for (int k = 0; k < d->n_blocks; ++k)
        if ((rc = pack(d->up_w[k], d->up_b[k], p.up[k], 4 * R2, R2, 0, 3, PLC_PACK_PIXSHUF, gen))) return rc;
    if ((rc = pack(d->post0_w, d->post0_b, p.post0, 32, R2, 0, 3, PLC_PACK_PLAIN, gen))) return rc;
    if ((rc = pack(d->post2_w, d->post2_b, p.post2, 1, 32, 0, 3, PLC_PACK_PLAIN, 0))) return rc;
    if ((rc = pack_batch_flush(&batch, st))) return rc;

    // ---- t-invariant gates: sigmoid(conv1x1(relu(conv3x3(aux)))) for DEM and land use, multiplied
    // Nothing before the gated post conv reads them, so they run on the side stream, beside the stem and the
    // first cell launches (six small launches over B images that would otherwise hold the chip alone).
    PlcAux* side_aux = nullptr;
    if (fast && plc_use_side_stream((long long)B * ohw)) {
        if ((rc = plc_aux(&side_aux))) return rc;
        PLC_HIP(hipEventRecord(side_aux->sev[0], (hipStream_t)stream));          // the packed weights are ready
        PLC_HIP(hipStreamWaitEvent(side_aux->side, side_aux->sev[0], 0));
        forks.opened(side_aux->side, side_aux->sev[1]);
    }
    {
        hipStream_t st = side_aux ? side_aux->side : (hipStream_t)stream;        // (this block's stream)
        const long long tot_d = (long long)B * 1 * ohw, tot_l = (long long)B * cdiv(L, 8) * ohw;
        hipLaunchKernelGGL(plc_resize_nchw_to_cb8_kernel, dim3(ew_grid(tot_d)), dim3(256), 0, st, d->dem,
                           (uint4*)(ws + p.dem_hr), B, 1, d->dem_H, d->dem_W, oH, oW, 0,
                           (float)d->dem_H / (float)oH, (float)d->dem_W / (float)oW);
        hipLaunchKernelGGL(plc_resize_nchw_to_cb8_kernel, dim3(ew_grid(tot_l)), dim3(256), 0, st, d->lu,
                           (uint4*)(ws + p.lu_hr), B, L, d->lu_H, d->lu_W, oH, oW, 1,
                           (float)d->lu_H / (float)oH, (float)d->lu_W / (float)oW);
        PLC_HIP(hipGetLastError());
        const int mcb = cdiv(R2 / 2, 8);
        ConvArgs a = conv_args(oH, oW, B);
        a.src0 = tens(ws + p.dem_hr, ohw * 16, 0);
        a.out0 = tens(ws + p.dem_mid, mcb * ohw * 16, 0);
        a.wpk = (const uint4*)(ws + p.dem0.w);
        a.bpk = (const float*)(ws + p.dem0.b);
        a.act = ACT_RELU;
        a.ncb_out = mcb;
        const int mid_groups = cdiv(mcb, 4);          // 32-row output groups of the aux 3x3 convs
        if ((rc = launch_conv<1, 0, 1, 3, STAGE_CB8, EPI_ACT_CB8>(a, B, mid_groups, st))) return rc;

        a = conv_args(oH, oW, B);
        a.src0 = tens(ws + p.dem_mid, mcb * ohw * 16, 0);
        a.out0 = tens(ws + p.gate_dem, (C2 / 8) * ohw * 16, 0);
        a.wpk = (const uint4*)(ws + p.dem2.w);
        a.bpk = (const float*)(ws + p.dem2.b);
        a.act = ACT_SIGMOID;
        a.ncb_out = C2 / 8;
        if ((rc = fast ? launch_conv<2, 0, 1, 1, STAGE_CB8, EPI_ACT_CB8>(a, B, 1, st)
                       : launch_convg<1, STAGE_CB8, EPI_ACT_CB8>(a, mcb, 0, cdiv(C2, 32), B, st)))
            return rc;

        a = conv_args(oH, oW, B);
        a.src0 = tens(ws + p.lu_hr, cdiv(L, 8) * ohw * 16, 0);
        a.out0 = tens(ws + p.lu_mid, mcb * ohw * 16, 0);
        a.wpk = (const uint4*)(ws + p.lu0.w);
        a.bpk = (const float*)(ws + p.lu0.b);
        a.act = ACT_RELU;
        a.ncb_out = mcb;
        if (L <= 8) rc = launch_conv<1, 0, 1, 3, STAGE_CB8, EPI_ACT_CB8>(a, B, mid_groups, st);
        else rc = launch_conv<2, 0, 1, 3, STAGE_CB8, EPI_ACT_CB8>(a, B, mid_groups, st);
        if (rc) return rc;

        a = conv_args(oH, oW, B);
        a.src0 = tens(ws + p.lu_mid, mcb * ohw * 16, 0);
        a.out0 = tens(ws + p.gate, (C2 / 8) * ohw * 16, 0);
        a.mul = tens(ws + p.gate_dem, (C2 / 8) * ohw * 16, 0);
        if (d->save_for_backward) a.out1 = tens(ws + p.gate_lu, (C2 / 8) * ohw * 16, 0);
        a.wpk = (const uint4*)(ws + p.lu2.w);
        a.bpk = (const float*)(ws + p.lu2.b);
        a.act = ACT_SIGMOID;
        a.ncb_out = C2 / 8;
        if ((rc = fast ? launch_conv<2, 0, 1, 1, STAGE_CB8, EPI_ACT_CB8>(a, B, 1, st)
                       : launch_convg<1, STAGE_CB8, EPI_ACT_CB8>(a, mcb, 0, cdiv(C2, 32), B, st)))
            return rc;
        if (side_aux) PLC_HIP(hipEventRecord(side_aux->sev[1], st));
    }

    // ---- stem over all T*B frames: x_t = relu(conv3x3([rain, row, col])) ------------------
    {
        ConvArgs a = conv_args(H, W, B);
        a.src0 = tens(d->rain, (long long)T * hw * 4, hw * 4);   // n = t*B + b  ->  rain[b][t]
        a.row_tab = d->row_tab;
        a.col_tab = d->col_tab;
        a.out0 = tens(ws + p.xs, p.img1, (long long)B * p.img1);
        a.wpk = (const uint4*)(ws + p.init.w);
        a.bpk = (const float*)(ws + p.init.b);
        a.act = ACT_RELU;
        a.ncb_out = C1 / 8;
        if ((rc = launch_conv<1, 0, 1, 3, STAGE_COORD, EPI_ACT_CB8>(a, T * B, cdiv(C1, 32), st))) return rc;
    }

    // ---- recurrence -----------------------------------------------------------------------
    {
        const size_t s1 = (size_t)B * p.img1, s2 = (size_t)B * p.img2;
        const bool save = d->save_for_backward != 0;
        // The initial states are zero (generator.py:156-160).  Nothing reads slot 0 of the state
        // arrays: the step-0 cells take zero_state, the backward's pointwise kernel takes a null
        // c_prev at t = 0 and its batched weight-gradient launches skip h of the first B images.
        // So the slot is never written either (no memsets, 75 MB per training forward at c3).
        auto cell1 = [&](int t) {
            ConvArgs a = cell_conv_args(ws + p.xs + t * s1, ws + p.h1s + t * s1, (const float*)(ws + p.c1s + t * s1 * 2),
                                        ws + p.cell1.w, (const float*)(ws + p.cell1.b), ws + p.h1s + (t + 1) * s1,
                                        (float*)(ws + p.c1s + (t + 1) * s1 * 2),
                                        save ? ws + p.gates1 + (size_t)t * s1 * 4 : nullptr, B, C1, C1, H, W);
            a.zero_state = t == 0;
            return a;
        };
        auto cell2 = [&](int t) {
            ConvArgs a = cell_conv_args(ws + p.h1s + (t + 1) * s1, ws + p.h2s + t * s2, (const float*)(ws + p.c2s + t * s2 * 2),
                                        ws + p.cell2.w, (const float*)(ws + p.cell2.b), ws + p.h2s + (t + 1) * s2,
                                        (float*)(ws + p.c2s + (t + 1) * s2 * 2),
                                        save ? ws + p.gates2 + (size_t)t * s2 * 4 : nullptr, B, C1, C2, H, W);
            a.zero_state = t == 0;
            return a;
        };
        // (Running cell 1 of step t+1 in the same launch as cell 2 of step t was measured: the two
        // kinds of workgroup share each SIMD's matrix pipe and finish together, so nothing is gained.)
        bool persistent = false;
        if (d->gate_fp8) {
            // fp8 operands for the two gate convs: the stem output is converted once for all frames, every
            // cell launch writes h' as bf16 (post conv, backward pass) and as fp8 (the next gate convs)
            const size_t s1f = (size_t)B * (C1 / 16) * hw * 16, s2f = (size_t)B * (C2 / 16) * hw * 16;
            if ((rc = pack_lstm_fp8_pair(d->cell1_w, ws + p.cell1_w8, (int*)(ws + p.scale8), C1, C1, d->cell2_w,
                                         ws + p.cell2_w8, (int*)(ws + p.scale8) + 8, C1, C2, st)))
                return rc;
            if ((rc = plc_cb8_bf16_to_cb16_fp8(ws + p.xs, ws + p.xs8, T * B, C1, H, W, stream))) return rc;
            for (int t = 0; t < T; ++t) {
                ConvArgs a1 = cell1(t);
                a1.src0.p = ws + p.xs8 + t * s1f;
                a1.src1.p = ws + p.h1s8 + t * s1f;
                a1.wpk = (const uint4*)(ws + p.cell1_w8);
                if ((rc = launch_cell_fp8(a1, ws + p.h1s8 + (t + 1) * s1f, (const int*)(ws + p.scale8), B, C1, C1, st))) return rc;
                ConvArgs a2 = cell2(t);
                a2.src0.p = ws + p.h1s8 + (t + 1) * s1f;
                a2.src1.p = ws + p.h2s8 + t * s2f;
                a2.wpk = (const uint4*)(ws + p.cell2_w8);
                if ((rc = launch_cell_fp8(a2, ws + p.h2s8 + (t + 1) * s2f, (const int*)(ws + p.scale8) + 8, B, C1, C2, st))) return rc;
            }
            persistent = true;      // (the recurrence has run)
        } else if (fast && T > 1 && !plc_no_persistent()) {
            PersistArgs pa;
            memset(&pa, 0, sizeof(pa));
            auto stepped = [&](ConvArgs a, long long s_in0, long long s_h) {      // outer stride = one time step
                a.src0.so = s_in0;
                a.src1.so = s_h;
                a.c_in.so = 2 * s_h;
                a.out0.so = s_h;
                a.out1.so = 2 * s_h;
                a.out2.so = 4 * s_h;
                return a;
            };
            pa.c1 = stepped(cell1(0), (long long)s1, (long long)s1);
            pa.c2 = stepped(cell2(0), (long long)s1, (long long)s2);
            pa.c1.sync = pa.c2.sync = (unsigned*)(ws + p.sync);
            if ((rc = launch_persistent(pa, B, T, save, st, &persistent))) return rc;
        }
        static const bool no_fwd_split = [] { const char* e = getenv("PLC_NO_FWD_SPLIT"); return e && *e && *e != '0'; }();
        const long long t16_all = (long long)B * cdiv(H, 16) * cdiv(W, 32);
        if (!persistent && fast && T > 1 && !plc_no_pairing() && !no_fwd_split && B >= 2 && (long long)B * hw >= (1ll << 17)) {
            // Two half-batch chains on two streams, the second one launch behind the first (as the backward
            // recurrence does): a cell launch starts with a burst of weight and halo-tile fetches and ends with
            // a drain, during which the other chain's workgroups have the matrix pipes.  Same kernels, same
            // tiles per image, identical results.
            PlcAux* aux = nullptr;
            if ((rc = plc_aux(&aux))) return rc;
            hipStream_t sh[2] = {st, aux->stream};
            const int nbh[2] = {B / 2, B - B / 2}, b0h[2] = {0, B / 2};
            auto half = [&](ConvArgs a, int h) {
                PlcTensor* ts[6] = {&a.src0, &a.src1, &a.c_in, &a.out0, &a.out1, &a.out2};
                for (PlcTensor* t : ts) if (t->p) t->p += (long long)b0h[h] * t->si;
                a.n_in = nbh[h];
                return a;
            };
            // stage k of a chain: 0 = cell 1 (0); k = 1 .. T-1: cell 2 (k-1) + cell 1 (k); T = cell 2 (T-1)
            auto stage = [&](int h, int k) -> int {
                const int nb = nbh[h];
                // (the tile height the whole batch would get: launch_conv's rule)
                if (k == 0) {
                    ConvArgs a = half(cell1(0), h);
                    if (t16_all >= 512)
                        return save ? launch_conv_th<2, 2, 2, 3, STAGE_CB8, EPI_LSTM_SAVE, 16>(a, nb, 1, sh[h])
                                    : launch_conv_th<2, 2, 2, 3, STAGE_CB8, EPI_LSTM, 16>(a, nb, 1, sh[h]);
                    return save ? launch_conv_th<2, 2, 2, 3, STAGE_CB8, EPI_LSTM_SAVE, 8>(a, nb, 1, sh[h])
                                : launch_conv_th<2, 2, 2, 3, STAGE_CB8, EPI_LSTM, 8>(a, nb, 1, sh[h]);
                }
                if (k == T) {
                    ConvArgs a = half(cell2(T - 1), h);
                    if (t16_all >= 512)
                        return save ? launch_conv_th<2, 4, 4, 3, STAGE_CB8, EPI_LSTM_SAVE, 16>(a, nb, 1, sh[h])
                                    : launch_conv_th<2, 4, 4, 3, STAGE_CB8, EPI_LSTM, 16>(a, nb, 1, sh[h]);
                    return save ? launch_conv_th<2, 4, 4, 3, STAGE_CB8, EPI_LSTM_SAVE, 8>(a, nb, 1, sh[h])
                                : launch_conv_th<2, 4, 4, 3, STAGE_CB8, EPI_LSTM, 8>(a, nb, 1, sh[h]);
                }
                return launch_pair(half(cell2(k - 1), h), half(cell1(k), h), nb, sh[h], B);
            };
            PLC_HIP(hipEventRecord(aux->ev[0], st));
            PLC_HIP(hipStreamWaitEvent(aux->stream, aux->ev[0], 0));
            forks.opened(aux->stream, aux->ev[2]);
            if ((rc = stage(0, 0))) return rc;
            if (plc_debug_fail_after_fork()) return plc_fail(PLC_EHIP, "rollout: injected failure after the stream fork");
            PLC_HIP(hipEventRecord(aux->ev[1], st));
            PLC_HIP(hipStreamWaitEvent(aux->stream, aux->ev[1], 0));
            for (int k = 1; k <= T + 1; ++k) {
                if (k <= T && (rc = stage(0, k))) return rc;
                if ((rc = stage(1, k - 1))) return rc;
            }
            PLC_HIP(hipEventRecord(aux->ev[2], aux->stream));
            PLC_HIP(hipStreamWaitEvent(st, aux->ev[2], 0));
            forks.joined(aux->stream);
        } else if (!persistent && fast && T > 1 && !plc_no_pairing()) {
            // cell 1 (0) | cell 2 (t-1) + cell 1 (t), t = 1 .. T-1 | cell 2 (T-1)
            if ((rc = launch_cell(cell1(0), B, C1, C1, st))) return rc;
            for (int t = 1; t < T; ++t)
                if ((rc = launch_pair(cell2(t - 1), cell1(t), B, st))) return rc;
            if ((rc = launch_cell(cell2(T - 1), B, C1, C2, st))) return rc;
        } else if (!persistent) {
            for (int t = 0; t < T; ++t) {
                if ((rc = launch_cell(cell1(t), B, C1, C1, st))) return rc;
                if ((rc = launch_cell(cell2(t), B, C1, C2, st))) return rc;
            }
        }
    }

    // ---- everything after the recurrence is batched over the T*B frames ----------------------
    const int NI = T * B;
    const char* feat = ws + p.h2s + (size_t)B * p.img2;
    int fh = H, fw = W;
    for (int k = 0; k < d->n_blocks; ++k) {
        ConvArgs a = conv_args(fh, fw, NI);
        a.src0 = tens(feat, (long long)(C2 / 8) * fh * fw * 16, 0);
        a.out0 = tens(ws + p.up_out[k], (long long)(C2 / 8) * fh * fw * 4 * 16, 0);
        a.wpk = (const uint4*)(ws + p.up[k].w);
        a.bpk = (const float*)(ws + p.up[k].b);
        if ((rc = fast ? launch_conv<4, 0, 4, 3, STAGE_CB8, EPI_PIXSHUF_RELU>(a, NI, 1, st)
                       : launch_convg<3, STAGE_CB8, EPI_PIXSHUF_RELU>(a, C2 / 8, 0, C2 / 8, NI, st)))
            return rc;
        feat = ws + p.up_out[k];
        fh *= 2;
        fw *= 2;
    }
    if (d->mid_H != fh || d->mid_W != fw) {
        const long long tot = (long long)NI * (C2 / 8) * d->mid_H * d->mid_W;
        hipLaunchKernelGGL(plc_resize_cb8_kernel, dim3(ew_grid(tot)), dim3(256), 0, st, (const uint4*)feat,
                           (uint4*)(ws + p.mid), NI, C2 / 8, fh, fw, d->mid_H, d->mid_W, d->mid_scale_h,
                           d->mid_scale_w);
        PLC_HIP(hipGetLastError());
        feat = ws + p.mid;
        fh = d->mid_H;
        fw = d->mid_W;
    }
    if (oH != fh || oW != fw) {
        const long long tot = (long long)NI * (C2 / 8) * ohw;
        hipLaunchKernelGGL(plc_resize_cb8_kernel, dim3(ew_grid(tot)), dim3(256), 0, st, (const uint4*)feat,
                           (uint4*)(ws + p.fin), NI, C2 / 8, fh, fw, oH, oW, d->out_scale_h, d->out_scale_w);
        PLC_HIP(hipGetLastError());
        feat = ws + p.fin;
    }
    if (side_aux) {
        PLC_HIP(hipStreamWaitEvent(st, side_aux->sev[1], 0));           // the gate maps are ready
        forks.joined(side_aux->side);
    }
    // (the fused kernel addresses a frame's four channel planes with 32-bit lane offsets)
    if (C2 == 32 && !plc_no_post_fusion() && ohw * 64 < (1ll << 32)) {
        // both output convs in one kernel (plc_post_kernel.h): the 32-channel intermediate stays in LDS; a training
        // forward also stores it for the backward pass.  That intermediate is bit-identical to the first of the two
        // launches below; the output agrees with them to fp32 rounding (the second conv sums in another order).
        PostArgs a;
        memset(&a, 0, sizeof(a));
        a.H = oH; a.W = oW; a.n_in = B; a.T = T;
        a.tiles_x = cdiv(oW, POST_TW);
        a.tiles_y = cdiv(oH, POST_TH);
        const long long tiles = (long long)NI * a.tiles_x * a.tiles_y;
        if (tiles >= (1ll << 31)) return plc_fail(PLC_EINVAL, "rollout: bad tile count");
        a.n_tiles = (int)tiles;
        a.feat = tens(feat, (C2 / 8) * ohw * 16, (long long)B * (C2 / 8) * ohw * 16);
        a.gate = tens(ws + p.gate, (C2 / 8) * ohw * 16, 0);
        a.p0 = tens(ws + p.p0, 4 * ohw * 16, (long long)B * 4 * ohw * 16);
        a.out = tens(d->out, (long long)T * ohw * 4, ohw * 4);      // n = t*B + b  ->  out[b][t]
        a.wpk0 = (const uint4*)(ws + p.post0.w);
        a.bpk0 = (const float*)(ws + p.post0.b);
        a.wpk2 = (const uint4*)(ws + p.post2.w);
        a.bpk2 = (const float*)(ws + p.post2.b);
        const bool save = d->save_for_backward != 0;
        auto kern = save ? plc_post_fused_kernel<true> : plc_post_fused_kernel<false>;
        static bool lds_raised[2][64] = {{false}};
        if ((rc = ensure_lds_limit((const void*)kern, POST_LDS_BYTES, lds_raised[save ? 1 : 0]))) return rc;
        // one workgroup of eight waves per CU, each a contiguous run of the time-innermost tile list
        const int resident = cu_count();
        const int grid = a.n_tiles < resident ? (a.n_tiles < 1 ? 1 : a.n_tiles) : resident;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), POST_LDS_BYTES, st, a);
        PLC_HIP(hipGetLastError());
        return PLC_OK;
    }
    {
        ConvArgs a = conv_args(oH, oW, B);
        a.src0 = tens(feat, (C2 / 8) * ohw * 16, (long long)B * (C2 / 8) * ohw * 16);
        a.gate = tens(ws + p.gate, (C2 / 8) * ohw * 16, 0);
        a.out0 = tens(ws + p.p0, 4 * ohw * 16, (long long)B * 4 * ohw * 16);
        a.wpk = (const uint4*)(ws + p.post0.w);
        a.bpk = (const float*)(ws + p.post0.b);
        a.act = ACT_RELU;
        a.ncb_out = 4;
        static const bool no_tinner = [] { const char* e = getenv("PLC_NO_GATED_TINNER"); return e && *e && *e != '0'; }();
        if ((rc = !fast ? launch_convg<3, STAGE_CB8_GATED, EPI_ACT_CB8>(a, C2 / 8, 0, 1, NI, st)
                  : (T > 1 && !no_tinner) ? launch_conv_gated_tinner<4, 1, 3, EPI_ACT_CB8, 8>(a, B, T, st)
                                          : launch_conv<4, 0, 1, 3, STAGE_CB8_GATED, EPI_ACT_CB8>(a, NI, 1, st)))
            return rc;

        a = conv_args(oH, oW, B);
        a.src0 = tens(ws + p.p0, 4 * ohw * 16, (long long)B * 4 * ohw * 16);
        a.out0 = tens(d->out, (long long)T * ohw * 4, ohw * 4);  // n = t*B + b  ->  out[b][t]
        a.wpk = (const uint4*)(ws + p.post2.w);
        a.bpk = (const float*)(ws + p.post2.b);
        a.act = ACT_NONE;
        a.cout = 1;
        if ((rc = launch_conv<4, 0, 1, 3, STAGE_CB8, EPI_F32_NCHW>(a, NI, 1, st))) return rc;
    }
    return PLC_OK;
}

#include "plc_bwd_api.h"
#include "plc_loss_api.h"
#include "plc_opt_api.h"

#ifdef PLC_TIMELINE
// Instrumented build only: point the kernels' time-stamp buffer at `buf` (device memory, or null).
extern "C" int plc_debug_set_timeline(void* buf) {
    PLC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(plc_timeline), &buf, sizeof(buf)));
    return PLC_OK;
}
extern "C" int plc_debug_select_timeline(int id) {
    PLC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(plc_timeline_sel), &id, sizeof(id)));
    return PLC_OK;
}
#endif
