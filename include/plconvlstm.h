/* libplconvlstm -- C ABI of the MI355X (gfx950) ConvLSTM rollout engine.
 *
 * The reference (Tomzhuiowewie/Pl-ConvLSTM-GAN) is pure Python/PyTorch and has no FFI; the
 * entry points below replace, one for one, the PyTorch call sites on its hot path.  Each
 * declaration names the reference lines whose arithmetic it implements.
 *
 * Conventions
 *   - every function returns 0 on success, a negative PLC_E* code on failure; the message of
 *     the last failure on the calling thread is available from plc_last_error();
 *   - the caller owns every buffer (workspaces included); the library allocates no device memory.
 *     The only device state it keeps is the fixed staging array of plc_convlstm_cell_fwd_noalloc (see
 *     there); the host state is a per-device cache of immutable facts (CU count, "LDS limit raised"
 *     flags, that array's address) and, created on the first rollout call on a device and kept, two
 *     auxiliary streams with the events of their forks and joins;
 *   - every call is asynchronous on the hipStream_t passed as `stream` (a `void*` here so the
 *     header needs no HIP include); nothing synchronises with the host.  The two rollout calls may put
 *     part of their work on the auxiliary streams (independent halves of the batch, work that nothing
 *     in the recurrence waits for): it is forked from `stream` by an event and joined back into it
 *     before the call's last launch, so for the caller everything is ordered on `stream` as if it
 *     had run there, and a stream capture of the call is valid; a call that fails after such a fork still
 *     joins the auxiliary streams before it returns its error code.  Rollout calls on one device share the
 *     auxiliary streams: calls issued from several host threads at once are serialised there (their
 *     results are unaffected).  A stream capture that contains a rollout call pulls the auxiliary streams
 *     into the capture until it ends: while such a capture is open, no other thread may issue rollout
 *     calls on that device (they would be enqueued onto a capturing stream);
 *   - tensors crossing the boundary are fp32, NCHW-contiguous, 16-byte aligned, exactly as the
 *     reference's modules take and return them.  Internally activations are kept as bf16 in a
 *     channel-blocked layout ("CB8": [image][C/8][H][W][8]) with fp32 accumulation and an fp32
 *     cell state.
 */
#ifndef PLCONVLSTM_H
#define PLCONVLSTM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PLC_VERSION 102

enum {
    PLC_OK = 0,
    PLC_EINVAL = -1,      /* bad argument (null pointer, non-positive size, misaligned pointer) */
    PLC_EUNSUPPORTED = -2,/* shape outside the compiled kernel set (see message) */
    PLC_EWORKSPACE = -3,  /* workspace too small */
    PLC_EHIP = -4         /* a HIP runtime call failed */
};

int plc_version(void);
const char* plc_last_error(void);

/* ---------------------------------------------------------------------------------------
 * Layout conversion between the boundary layout (NCHW fp32) and the internal CB8 layouts.
 * `cb8` buffers hold N * ceil(C/8) * H * W * 16 bytes (bf16) or * 32 bytes (fp32 state).
 * ------------------------------------------------------------------------------------- */
int plc_nchw_f32_to_cb8_bf16(const float* src, void* dst_cb8, int N, int C, int H, int W, void* stream);
int plc_cb8_bf16_to_nchw_f32(const void* src_cb8, float* dst, int N, int C, int H, int W, void* stream);
int plc_nchw_f32_to_cb8_f32(const float* src, float* dst_cb8, int N, int C, int H, int W, void* stream);
int plc_cb8_f32_to_nchw_f32(const float* src_cb8, float* dst, int N, int C, int H, int W, void* stream);

/* ---------------------------------------------------------------------------------------
 * Weight packing: PyTorch Conv2d weight [cout][c0+c1][ks][ks] (+ bias [cout] or NULL) ->
 * MFMA A-fragment order.  c0/c1 are the channel counts of the two concatenated inputs
 * (c1 = 0 for a plain conv).  mode: 0 plain, 1 ConvLSTM gate conv (rows regrouped so the four
 * gates of a hidden channel meet in one lane), 2 conv feeding PixelShuffle(2).
 * Sizes: plc_packed_weight_bytes / plc_packed_bias_bytes.
 * ------------------------------------------------------------------------------------- */
enum { PLC_PACK_PLAIN = 0, PLC_PACK_LSTM = 1, PLC_PACK_PIXSHUF = 2 };
size_t plc_packed_weight_bytes(int cout, int c0, int c1, int ks);
size_t plc_packed_bias_bytes(int cout);
int plc_pack_conv_weight(const float* w, const float* bias, void* wpk, float* bpk,
                         int cout, int c0, int c1, int ks, int mode, void* stream);

/* ---------------------------------------------------------------------------------------
 * ConvLSTM cell, forward.  Replaces ConvLSTMCell.forward (src/models/convlstm.py:16-28):
 *   z = conv3x3(cat[x, h]) + b ; i,f,o = sigmoid(z[0:3Ch]) ; g = tanh(z[3Ch:4Ch]) ;
 *   c' = f*c + i*g ; h' = o*tanh(c')          (gate order i,f,o,g; input order [x | h])
 * The gate conv runs as an implicit GEMM on v_mfma_f32_32x32x16_bf16 with the activations and
 * the state update fused into the epilogue.
 *
 * plc_convlstm_cell_fwd_cb8: operands already in the internal layouts (the rollout's primitive).
 *   x: CB8 bf16 [B][Cin/8][H][W][8], h: CB8 bf16 [B][Ch/8]..., c_in/c_out: CB8 fp32,
 *   h_out: CB8 bf16, gates: NULL or CB8 bf16 [B][4Ch/8][H][W][8] (i,f,o,g saved for backward),
 *   wpk/bpk from plc_pack_conv_weight(mode = PLC_PACK_LSTM).
 * plc_convlstm_cell_fwd: boundary layout (NCHW fp32) in and out; `workspace` must hold
 *   plc_convlstm_cell_workspace_bytes(...) bytes.
 * The CB8 form takes whole 8-channel blocks (Cin, Ch multiples of 8; (16,16) and (16,32) run on weight-stationary
 * instantiations, others on the run-time-width kernel).  The boundary form takes ANY Cin, Ch > 0, as the
 * reference's ConvLSTMCell does: widths that are not multiples of 8 are carried as the next multiple with zero
 * channels (plc_pack_conv_weight emits zero rows, columns and biases for them, which keeps them zero).
 * ------------------------------------------------------------------------------------- */
int plc_convlstm_cell_fwd_cb8(const void* x, const void* h, const float* c_in,
                              const void* wpk, const float* bpk,
                              void* h_out, float* c_out, void* gates,
                              int B, int Cin, int Ch, int H, int W, void* stream);
size_t plc_convlstm_cell_workspace_bytes(int B, int Cin, int Ch, int H, int W);
int plc_convlstm_cell_fwd(const float* x, const float* h, const float* c,
                          const float* weight, const float* bias,
                          float* h_next, float* c_next,
                          void* workspace, size_t workspace_bytes,
                          int B, int Cin, int Ch, int H, int W, void* stream);

/* ---------------------------------------------------------------------------------------
 * ConvLSTM cell forward with fp8 gate-conv operands.  Same arithmetic as plc_convlstm_cell_fwd_cb8
 * (convlstm.py:16-28) with x, h and the conv weights rounded to OCP e4m3 (fp32 accumulation, fp32 cell
 * state, fp32 bias), on v_mfma_scale_f32_32x32x64_f8f6f4.  The reference computes in fp32; the stated
 * tolerance of this path is in the tests (tests/test_gpu_fp8.py).
 *   x8, h8:    fp8 "CB16" [B][C/16][H][W][16] (plc_cb8_bf16_to_cb16_fp8 converts a bf16 CB8 tensor)
 *   wpk8, scale_word: plc_pack_lstm_weight_fp8 (weights scaled by a power of two taken from max|w|, the
 *              instruction's block-scale operand undoes it; scale_word is 2 ints)
 *   bpk:       packed bias of plc_pack_conv_weight(mode = PLC_PACK_LSTM)
 *   h_out (bf16 CB8), h8_out (fp8 CB16, the next step's operand), c_out (fp32 CB8), gates (NULL or bf16 CB8)
 * Supported: (Cin, Ch) in {(16,16), (16,32)}.
 * ------------------------------------------------------------------------------------- */
size_t plc_fp8_packed_weight_bytes(int Cin, int Ch);
int plc_pack_lstm_weight_fp8(const float* w, void* wpk8, int* scale_word, int Cin, int Ch, void* stream);
int plc_cb8_bf16_to_cb16_fp8(const void* src_cb8, void* dst_cb16, int N, int C, int H, int W, void* stream);
int plc_convlstm_cell_fwd_fp8(const void* x8, const void* h8, const float* c_in, const void* wpk8,
                              const int* scale_word, const float* bpk, void* h_out, void* h8_out,
                              float* c_out, void* gates, int B, int Cin, int Ch, int H, int W, void* stream);

/* ---------------------------------------------------------------------------------------
 * ConvLSTM cell forward without any allocation.  This entry point has no counterpart in the
 * reference: it exists for callers that cannot allocate device memory -- code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc -- and therefore cannot bring the
 * workspace plc_convlstm_cell_fwd asks for.  Same arithmetic and same kernels as
 * plc_convlstm_cell_fwd (identical results, bit for bit); the staging buffers (layout conversions,
 * packed weights, CB8 outputs) live in a fixed array inside the library's device image:
 *   - size limit: the call needs plc_convlstm_cell_workspace_bytes(B, Cin, Ch, H, W) bytes of it and
 *     fails with PLC_EWORKSPACE beyond PLC_STATIC_SCRATCH_BYTES (32 MiB: e.g. (Cin, Ch) = (16, 32)
 *     up to 80 000 pixels per call, B = 1 at 256x256); nothing runs in that case;
 *   - every GPU the library is used on holds its own copy of the array until the process ends;
 *   - calls on one GPU share that GPU's array: calls that overlap in time on one device (two streams,
 *     two threads) race; issue them on one stream or serialise them;
 *   - the call makes no hipMalloc / hipFree and no allocating library call; its first use on a device
 *     looks the array's address up once (hipGetSymbolAddress), so make one call per device before
 *     starting a graph capture.
 * ------------------------------------------------------------------------------------- */
#define PLC_STATIC_SCRATCH_BYTES (32u << 20)
size_t plc_convlstm_cell_noalloc_max_bytes(void);
int plc_convlstm_cell_fwd_noalloc(const float* x, const float* h, const float* c,
                                  const float* weight, const float* bias,
                                  float* h_next, float* c_next,
                                  int B, int Cin, int Ch, int H, int W, void* stream);

/* ---------------------------------------------------------------------------------------
 * Generator rollout, forward.  Replaces Generator.forward (src/models/generator.py:94-205):
 * coord-conv stem (coordconv.py:3-10 + generator.py:166-168), two stacked ConvLSTM cells
 * unrolled over T (generator.py:170-171), PixelShuffle upsample blocks (generator.py:24-28,
 * 174-176), residual / target bilinear resize (generator.py:179-195), DEM and land-use gates
 * (attention.py:12-13, 25-26; evaluated once per call, they do not depend on t) and the two
 * output convs (generator.py:67-71, 202).
 * ------------------------------------------------------------------------------------- */
typedef struct plc_rollout_desc {
    /* sizes */
    int B, T, H, W;          /* rain_lr is [B][T][1][H][W] */
    int C1, C2;              /* hidden_dims */
    int L;                   /* land-use channels */
    int dem_H, dem_W;        /* dem is [B][1][dem_H][dem_W] */
    int lu_H, lu_W;          /* lu  is [B][L][lu_H][lu_W] */
    int n_blocks;            /* PixelShuffle(2) blocks */
    int mid_H, mid_W;        /* size after the residual bilinear resize (== H*2^n_blocks if none) */
    int out_H, out_W;        /* final size (after the optional resize to target_size) */
    float mid_scale_h, mid_scale_w;   /* source/destination coordinate scales of the residual resize */
    float out_scale_h, out_scale_w;   /* ... of the resize to target_size */
    int save_for_backward;   /* keep gate activations in the workspace */
    int gate_fp8;            /* 1: the two gate convs take fp8 (e4m3) operands (hidden_dims [16,32] only); 0: bf16 */
    /* inputs (fp32, device) */
    const float* rain;
    const float* dem;
    const float* lu;
    const float* row_tab;    /* linspace(0,1,H)  [H] */
    const float* col_tab;    /* linspace(0,1,W)  [W] */
    /* parameters (fp32, device, PyTorch layouts; state_dict names in comments) */
    const float* init_w; const float* init_b;       /* init_conv.{weight,bias}      [C1][3][3][3]   */
    const float* cell1_w; const float* cell1_b;     /* cell1.conv.*                 [4C1][2C1][3][3]*/
    const float* cell2_w; const float* cell2_b;     /* cell2.conv.*                 [4C2][C1+C2][3][3] */
    const float* dem0_w; const float* dem0_b;       /* dem_attn.conv.0.*            [C2/2][1][3][3] */
    const float* dem2_w; const float* dem2_b;       /* dem_attn.conv.2.*            [C2][C2/2][1][1]*/
    const float* lu0_w; const float* lu0_b;         /* lu_attn.conv.0.*             [C2/2][L][3][3] */
    const float* lu2_w; const float* lu2_b;         /* lu_attn.conv.2.*             [C2][C2/2][1][1]*/
    const float* up_w[8]; const float* up_b[8];     /* upsample_blocks.k.conv.*     [4C2][C2][3][3] */
    const float* post0_w; const float* post0_b;     /* post_process.0.*             [32][C2][3][3]  */
    const float* post2_w; const float* post2_b;     /* post_process.2.*             [1][32][3][3]   */
    /* output (fp32, device): [B][T][1][out_H][out_W] */
    float* out;
    /* caller-owned scratch of plc_generator_workspace_bytes(desc) bytes, 256-byte aligned */
    void* workspace;
    size_t workspace_bytes;
} plc_rollout_desc;

size_t plc_generator_workspace_bytes(const plc_rollout_desc* d);
int plc_generator_rollout_fwd(const plc_rollout_desc* d, void* stream);

/* Introspection used by tests and the benchmark: byte offset inside the workspace of a named
 * internal buffer ("h1s", "c1s", "h2s", "c2s", "xs", "gates1", "gates2", "gate", "p0"), or
 * (size_t)-1 if unknown. */
size_t plc_generator_workspace_offset(const plc_rollout_desc* d, const char* name);

/* ---------------------------------------------------------------------------------------
 * Backward passes.  The reference obtains them from autograd (loss.backward(),
 * src/training/trainer.py:310); here they are explicit kernels: the pointwise LSTM backward,
 * data-gradient convs (the forward kernel with transposed, tap-flipped weights) and
 * weight-gradient GEMMs contracting over pixels.
 *
 * plc_convlstm_cell_bwd: gradients of ConvLSTMCell.forward (convlstm.py:16-28) given
 *   dL/dh' (dh_next) and dL/dc' (dc_next); all tensors NCHW fp32; dweight [4Ch][Cin+Ch][3][3],
 *   dbias [4Ch] (may be NULL) are overwritten.
 * plc_generator_rollout_bwd: backpropagation through time over the rollout that
 *   plc_generator_rollout_fwd ran with save_for_backward = 1 on the SAME descriptor (its
 *   workspace holds the saved activations).  Writes the gradient of every parameter, for every
 *   geometry the forward accepts (PixelShuffle blocks, the residual bilinear resize and the resize
 *   to target_size, generator.py:174-195, included).
 * ------------------------------------------------------------------------------------- */
size_t plc_convlstm_cell_bwd_workspace_bytes(int B, int Cin, int Ch, int H, int W);
int plc_convlstm_cell_bwd(const float* x, const float* h, const float* c,
                          const float* weight, const float* bias,
                          const float* dh_next, const float* dc_next,
                          float* dx, float* dh, float* dc, float* dweight, float* dbias,
                          void* workspace, size_t workspace_bytes,
                          int B, int Cin, int Ch, int H, int W, void* stream);

/* The pointwise part of that backward on the engine's own layouts, one timestep of B images:
 * autograd of convlstm.py:22-28 (sigmoid/tanh gates, c' = f*c + i*g, h' = o*tanh(c')).
 *   dh0, dh1, dh2: up to three incoming dL/dh' contributions, CB8 bf16 [B][Ch/8], each with its own
 *                  byte stride between images (a view into a wider tensor is allowed); NULL = absent.
 *   dc_next: dL/dc' (CB8 fp32 [B][Ch/8]) or NULL;  gates: saved i,f,o,g (CB8 bf16 [B][4Ch/8]);
 *   c_prev: c (CB8 fp32) or NULL (= zeros, the first timestep of a rollout).
 *   Writes dz (CB8 bf16 [B][4Ch/8], gate-major) and dc (CB8 fp32).
 * 256 bytes move per 8-channel unit with two dh inputs and dc_next present. */
int plc_lstm_pointwise_bwd(const void* dh0, long long dh0_image_bytes, const void* dh1, long long dh1_image_bytes,
                           const void* dh2, long long dh2_image_bytes, const float* dc_next, const void* gates,
                           const float* c_prev, void* dz, float* dc, int B, int Ch, int H, int W, void* stream);
/* The form the rollout's backward pass uses: dL/dc travels between time steps as bf16 CB8 (16-byte units)
 * like every other gradient tensor of the pass (dh, dz), which takes 32 of the 256 bytes per unit off
 * the kernel (the parameter gradients do not move: the rounding is unbiased and they are sums over
 * millions of pixels).  dc_next / dc: CB8 bf16 [B][Ch/8]; everything else as above.  224 bytes per unit. */
int plc_lstm_pointwise_bwd_bf16dc(const void* dh0, long long dh0_image_bytes, const void* dh1, long long dh1_image_bytes,
                                  const void* dh2, long long dh2_image_bytes, const void* dc_next, const void* gates,
                                  const float* c_prev, void* dz, void* dc, int B, int Ch, int H, int W, void* stream);

/* Two such passes in ONE launch: what the rollout's backward pass issues per step of its skewed loop (cell 2 of
 * step t - 1 and cell 1 of step t wait for the same data-gradient conv and for nothing else, so they share a
 * launch; plc_generator_rollout_bwd runs it on each half of the batch).  Every field as the argument of the same
 * name of plc_lstm_pointwise_bwd_bf16dc; B, H, W are common to both.  Exported so that the benchmark can time the
 * launch the training step really makes. */
typedef struct plc_pointwise_bwd_args {
    const void* dh0; long long dh0_image_bytes;
    const void* dh1; long long dh1_image_bytes;
    const void* dc_next;     /* CB8 bf16 or NULL */
    const void* gates;
    const float* c_prev;     /* CB8 fp32 or NULL */
    void* dz;
    void* dc;                /* CB8 bf16 */
    int Ch;
} plc_pointwise_bwd_args;
int plc_lstm_pointwise_bwd_pair_bf16dc(const plc_pointwise_bwd_args* first, const plc_pointwise_bwd_args* second,
                                       int B, int H, int W, void* stream);

typedef struct plc_rollout_grads {
    const float* grad_out;   /* dL/d out, fp32 [B][T][1][out_H][out_W] */
    /* parameter gradients (fp32, device, same shapes as the parameters; overwritten) */
    float* init_w; float* init_b;
    float* cell1_w; float* cell1_b;
    float* cell2_w; float* cell2_b;
    float* dem0_w; float* dem0_b;
    float* dem2_w; float* dem2_b;
    float* lu0_w; float* lu0_b;
    float* lu2_w; float* lu2_b;
    float* up_w[8]; float* up_b[8];
    float* post0_w; float* post0_b;
    float* post2_w; float* post2_b;
    /* caller-owned scratch of plc_generator_bwd_workspace_bytes(desc) bytes, 256-byte aligned */
    void* workspace;
    size_t workspace_bytes;
    /* Optional gradients with respect to the inputs, what autograd returns for generator.py:101-205 when an input
     * has requires_grad (fp32, device, the inputs' own shapes; overwritten).  NULL = not wanted, the case of the
     * training step: nothing is computed for it. */
    float* d_rain;           /* [B][T][1][H][W] */
    float* d_dem;            /* [B][1][dem_H][dem_W] */
    float* d_lu;             /* [B][L][lu_H][lu_W] */
} plc_rollout_grads;

size_t plc_generator_bwd_workspace_bytes(const plc_rollout_desc* d);
int plc_generator_rollout_bwd(const plc_rollout_desc* d, const plc_rollout_grads* g, void* stream);

/* ---------------------------------------------------------------------------------------
 * CombinedLoss and its gradient.  Replaces CombinedLoss.forward
 * (src/losses/combined_loss.py:173-191) and the autograd backward of its four terms:
 *   point     mean over observed gauges of |pred - obs| * weight(obs)   (combined_loss.py:79-141;
 *             weights combined_loss.py:23-59; gauge pixel = trunc((coord + 0.5) * scale - 0.5))
 *   conserve  mean |area_interpolate(pred -> LR grid) - lr_input|       (combined_loss.py:64-74)
 *   smooth    mean|d/dx| + mean|d/dy|                                   (combined_loss.py:146-155)
 *   temporal  mean|pred[t+1] - pred[t]|  (NaN when T == 1, as the reference) (combined_loss.py:160-168)
 * One pass over pred produces the five scalars (device memory, no host sync) and, when
 * grad_pred is given, d total / d pred.  Sums use a fixed reduction order (bit-repeatable).
 * ------------------------------------------------------------------------------------- */
enum { PLC_LOSS_W_NONE = 0, PLC_LOSS_W_LOG = 1, PLC_LOSS_W_SQRT = 2, PLC_LOSS_W_STRATIFIED = 3 };
typedef struct plc_loss_desc {
    int B, T;
    int out_H, out_W;        /* pred is [B][T][1][out_H][out_W] */
    int H, W;                /* lr_input is [B][T][1][H][W] */
    int N;                   /* rain gauges; 0 = no gauge term (point = 0) */
    int values_per_batch;    /* 1: s_values is [B][T][N]; 0: [T][N], shared by the batch */
    int weight_mode;         /* PLC_LOSS_W_* (use_weighted_loss = False -> PLC_LOSS_W_NONE) */
    float scale_factor;
    float lambda_point, lambda_conserve, lambda_smooth, lambda_temporal;
    const float* pred;
    const float* lr_input;
    const long long* s_coords;  /* int64 [N][2] (row, col) on the LR grid (sample 0's when batched) */
    const float* s_values;      /* NaN = not observed */
    float* losses;           /* [5]: total, point, conserve, smooth, temporal */
    float* grad_pred;        /* NULL or [B][T][1][out_H][out_W], overwritten */
    void* workspace;         /* plc_combined_loss_workspace_bytes(desc) bytes, 16-byte aligned */
    size_t workspace_bytes;
} plc_loss_desc;

size_t plc_combined_loss_workspace_bytes(const plc_loss_desc* d);
int plc_combined_loss(const plc_loss_desc* d, void* stream);

/* ---------------------------------------------------------------------------------------
 * Gradient clipping + Adam.  Replaces torch.nn.utils.clip_grad_norm_(model.parameters(),
 * max_norm) followed by torch.optim.Adam.step() in the reference's training step
 * (src/training/trainer.py:311-313; Adam with the reference's settings: lr from the config, default
 * betas / eps, weight_decay = 0, no amsgrad):
 *   norm = sqrt(sum over all listed gradients of g^2);  clip = min(1, max_norm / (norm + 1e-6))
 *   g *= clip (in place, every listed tensor; a NaN norm gives a NaN clip factor, as torch.clamp does)
 *   for tensors with update != 0:  g += weight_decay * p;  m += (1 - beta1)(g - m);
 *       v = beta2 v + (1 - beta2) g^2;  p -= lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps)
 * Tensors with update == 0 count in the norm and are clipped but not stepped (the reference's
 * optimizer does not hold the generator's lazily created upsample blocks, yet its clip covers
 * them).  `step` is the 1-based number of this update.  The table `t` is read on the host during
 * the call; `grad_norm` (device, 1 float, may be NULL) receives the norm before clipping.
 * max_norm <= 0 disables clipping.  Two launches, no host synchronisation.
 * ------------------------------------------------------------------------------------- */
typedef struct plc_opt_tensor {
    float* param;        /* may be NULL when update == 0 */
    float* grad;
    float* exp_avg;      /* Adam first moment  (may be NULL when update == 0) */
    float* exp_avg_sq;   /* Adam second moment (may be NULL when update == 0) */
    long long numel;
    int update;
    const float* restore; /* guarded step only, may be NULL: what `grad` is set back to when the step is skipped */
} plc_opt_tensor;

size_t plc_clip_adam_workspace_bytes(const plc_opt_tensor* t, int n);
int plc_clip_adam_step(const plc_opt_tensor* t, int n, double max_norm, double lr, double beta1, double beta2,
                       double eps, double weight_decay, long long step, float* grad_norm,
                       void* workspace, size_t workspace_bytes, void* stream);

/* The same step with the reference's NaN-batch skip (src/training/trainer.py:306-308,
 * `if torch.isnan(loss): continue`) decided on the device, so that the host can launch the backward
 * pass and this step without waiting for the loss:
 *   guard        device float (the loss; may be NULL = never skip).  If it is NaN the call changes
 *                no parameter, moment or counter, and it leaves the gradients as they are except those of
 *                table entries with a `restore` pointer: such a gradient is overwritten with the numel
 *                floats at `restore`.  This is for tensors whose gradient ACCUMULATES over steps because no
 *                optimizer zeroes it (the reference's optimizer is built before the generator's lazy
 *                upsample blocks exist, trainer.py:153-158 + generator.py:129-132): the reference skips a
 *                NaN batch BEFORE loss.backward() (trainer.py:306-310), so a caller that has already run the
 *                backward pass hands in a copy taken before it and gets the accumulated value back.
 *   step_counter device int64, the number of updates applied so far (0 for a fresh optimizer); it is
 *                incremented by an applied update and its new value is the `step` of the formula
 *                above (bias corrections formed in double on the device).
 *   skipped      device int (may be NULL): 1 if this call was skipped, else 0.
 * `grad_norm` is written in both cases.  Three launches, no host synchronisation.  The workspace
 * (plc_clip_adam_workspace_bytes) must be 16-byte aligned. */
int plc_clip_adam_step_guarded(const plc_opt_tensor* t, int n, double max_norm, double lr, double beta1, double beta2,
                               double eps, double weight_decay, const float* guard, long long* step_counter,
                               int* skipped, float* grad_norm, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * Data-parallel gradient exchange: the two ends of the ONE all-reduce of a training step.
 * No counterpart in the reference (single process, single device: src/training/trainer.py:20); they stand
 * where DistributedDataParallel's bucket copies would, between loss.backward() (trainer.py:310) and
 * clip_grad_norm_ (trainer.py:311).  The caller owns `flat` and performs the collective itself
 * (all-reduce SUM over the ranks) between the two calls.
 *
 *   plc_dp_pack_grads    flat[0 .. G) = n_local * (the gradients of the table, in table order),
 *                        flat[G] = 1 if *guard is NaN else 0 (guard NULL: 0), flat[G + 1] = n_local;
 *                        G = sum of numel, flat_elems must be G + 2.  One launch.
 *   plc_dp_unpack_grads  after the SUM: every gradient of the table = flat[i] / flat[G + 1] (the per-sample
 *                        mean over the global batch when n_local was each rank's sample count), and, when
 *                        guard_out is not NULL, *guard_out = NaN if flat[G] > 0 or is NaN (some rank's loss
 *                        was NaN: every rank then skips the step) else *guard_in.  One launch.
 * Only `grad` and `numel` of the table entries are read.
 * ------------------------------------------------------------------------------------- */
int plc_dp_pack_grads(const plc_opt_tensor* t, int n, float* flat, long long flat_elems, float n_local,
                      const float* guard, void* stream);
int plc_dp_unpack_grads(const plc_opt_tensor* t, int n, const float* flat, long long flat_elems,
                        const float* guard_in, float* guard_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PLCONVLSTM_H */
