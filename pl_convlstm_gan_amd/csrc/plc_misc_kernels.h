// Weight packing, layout conversion and resampling kernels (all HBM-bound, one pass each).
#pragma once
#include "plc_common.h"

enum { PACK_PLAIN = 0, PACK_LSTM = 1, PACK_PIXSHUF = 2 };

// Output-channel index of row r (0..31) of M tile mt.  Row r = 8g + j8 (see plc_conv_kernel.h).
__device__ __forceinline__ int plc_row_to_cout(int mode, int mt, int r, int ch_per_gate) {
    const int g = r >> 3, j8 = r & 7;
    // gate-major conv channels; hidden channels past ch_per_gate (a width that is not a multiple of 8) are padding
    if (mode == PACK_LSTM) return 8 * mt + j8 < ch_per_gate ? g * ch_per_gate + 8 * mt + j8 : (1 << 30);
    if (mode == PACK_PIXSHUF) return 4 * (8 * mt + j8) + g;             // PixelShuffle(2): 4c + 2i + j
    return 32 * mt + r;
}

// Map a channel index of the block-padded concat [src0 blocks | src1 blocks] to the index in the
// unpadded concat [c0 | c1], or -1 for a padding channel.
__device__ __forceinline__ int plc_padded_to_concat(int cb, int j, int c0, int c1) {
    const int ncb0 = (c0 + 7) / 8;
    if (cb < ncb0) return (8 * cb + j < c0) ? 8 * cb + j : -1;
    return (8 * (cb - ncb0) + j < c1) ? c0 + 8 * (cb - ncb0) + j : -1;
}

// w: fp32 [cout][c0 + c1][ks][ks] (cross-correlation weights, PyTorch layout)
// wpk: bf16 [n_mt][ksplit][nks_w][64 lanes][8].  Lane l = (half, row) of k-split part q holds
//      k index 16 s + 8 half + j = unit u = 2s + half = tap * ncbw + cb' of channel block
//      cb = q * ncbw + cb', channel 8 cb + j.
// transposed == 0 (forward conv):  rows = output channels (regrouped per `mode`), K channels = the
//      block-padded concat [src0 | src1];
// transposed != 0 (data gradient): rows = channels of the block-padded concat [src0 | src1]
//      (plain order), K channels = the cout output channels, taps flipped:
//      W'[ci][o][dy][dx] = w[o][ci][ks-1-dy][ks-1-dx].
// chunk > 0 (the run-time-width conv kernel, ksplit 1): the K units are ordered chunk-major -- the
//      channel blocks are taken `chunk` at a time and inside a chunk of nb blocks unit u = tap * nb +
//      (block inside the chunk); a chunk with an odd unit count ends in a zero unit.
// One conv's packing job; `block` of `n_blocks` (256 threads each) share it grid-stride fashion.
struct PackJob {
    const float* w;
    const float* bias;
    unsigned short* wpk;
    float* bpk;
    int cout, c0, c1, ks, mode, n_mt, nks_w, ch_per_gate, transposed, ksplit;
    int chunk;
    int lstm_k;                     // transposed jobs: > 0 = the K channels are an LSTM gate conv's outputs as the engine
                                    // stores them (gate-major, each gate padded to whole 8-channel blocks) and this many
                                    // hidden channels are real; 0 = plain channel order
    int k_blocks;                   // transposed jobs: 8-channel blocks of the K axis (>= ceil(cout / 8); the rest are zero)
    int first_block, n_blocks;      // position inside a multi-job launch
};

__device__ __forceinline__ void plc_pack_weight_body(const PackJob& jb, int block, int n_blocks) {
    const float* __restrict__ w = jb.w;
    const float* __restrict__ bias = jb.bias;
    unsigned short* __restrict__ wpk = jb.wpk;
    float* __restrict__ bpk = jb.bpk;
    const int cout = jb.cout, c0 = jb.c0, c1 = jb.c1, ks = jb.ks, mode = jb.mode, n_mt = jb.n_mt, nks_w = jb.nks_w;
    const int ch_per_gate = jb.ch_per_gate, transposed = jb.transposed, ksplit = jb.ksplit;
    const int lstm_nmt = (jb.lstm_k + 7) / 8;
    const int ncb_k = transposed ? jb.k_blocks : (c0 + 7) / 8 + (c1 + 7) / 8;
    const int ncbw = ncb_k / ksplit;
    const int taps = ks * ks;
    const long long total = (long long)n_mt * ksplit * nks_w * 64 * 8;
    for (long long idx = block * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)n_blocks * blockDim.x) {
        const int j = idx & 7;
        const int lane = (idx >> 3) & 63;
        long long rest = idx >> 9;
        const int s = rest % nks_w;
        rest /= nks_w;
        const int q = rest % ksplit;
        const int mt = rest / ksplit;
        const int r = lane & 31, half = lane >> 5;
        int tap, cb;
        if (jb.chunk > 0) {
            // every chunk but the last is full, so the k-step's chunk is a plain quotient
            const int nks_full = (taps * jb.chunk + 1) / 2;
            const int c = s / nks_full;
            const int nb = ncb_k - c * jb.chunk < jb.chunk ? ncb_k - c * jb.chunk : jb.chunk;
            const int u = 2 * (s - c * nks_full) + half;
            tap = u / nb;
            cb = c * jb.chunk + u % nb;
        } else {
            const int u = 2 * s + half;
            tap = u / ncbw;
            cb = q * ncbw + u % ncbw;
        }
        float v = 0.0f;
        if (tap < taps) {
            const int dy = tap / ks, dx = tap % ks;
            if (!transposed) {
                const int o = plc_row_to_cout(mode, mt, r, ch_per_gate);
                const int ci = plc_padded_to_concat(cb, j, c0, c1);
                if (o < cout && ci >= 0) v = w[(((long long)o * (c0 + c1) + ci) * ks + dy) * ks + dx];
            } else {
                const int rowg = 32 * mt + r;
                const int ci = plc_padded_to_concat(rowg >> 3, rowg & 7, c0, c1);
                int o = 8 * cb + j;
                if (jb.lstm_k > 0) {            // block cb = gate g, hidden channels 8 * (cb % lstm_nmt) ..
                    const int hc = 8 * (cb % lstm_nmt) + j;
                    o = hc < jb.lstm_k ? (cb / lstm_nmt) * jb.lstm_k + hc : (1 << 30);
                }
                if (o < cout && ci >= 0 && (rowg >> 3) < (c0 + 7) / 8 + (c1 + 7) / 8)
                    v = w[(((long long)o * (c0 + c1) + ci) * ks + (ks - 1 - dy)) * ks + (ks - 1 - dx)];
            }
        }
        wpk[idx] = f2bf(v);
    }
    if (bpk) {
        for (int i = block * blockDim.x + threadIdx.x; i < n_mt * 32; i += n_blocks * blockDim.x) {
            float b = 0.0f;
            if (!transposed && bias) {
                const int o = plc_row_to_cout(mode, i >> 5, i & 31, ch_per_gate);
                if (o < cout) b = bias[o];
            }
            bpk[i] = b;
        }
    }
}

__global__ void plc_pack_weight_kernel(const PackJob jb) { plc_pack_weight_body(jb, blockIdx.x, gridDim.x); }

// Every conv of a rollout in ONE launch (the weights change every optimizer step, so they are
// re-packed per call; one launch instead of ten to twenty tiny ones).
constexpr int PACK_MAX_JOBS = 24;
struct PackTable {
    PackJob job[PACK_MAX_JOBS];
    int n;
};
__global__ void plc_pack_multi_kernel(const PackTable tb) {
    int j = 0;
    while (j + 1 < tb.n && (int)blockIdx.x >= tb.job[j + 1].first_block) ++j;
    plc_pack_weight_body(tb.job[j], blockIdx.x - tb.job[j].first_block, tb.job[j].n_blocks);
}

// Zero n 32-bit words.  Used in place of hipMemsetAsync: inside a captured HIP graph a memset node of this
// runtime is not reliably ordered between the kernel nodes around it (devtools/graph_memset_probe.hip), a
// kernel node is.
__global__ void plc_zero_words_kernel(unsigned* __restrict__ p, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0u;
}

// NCHW fp32 -> CB8 bf16 (channels padded with zeros to a multiple of 8).
__global__ void plc_nchw_to_cb8_kernel(const float* __restrict__ src, uint4* __restrict__ dst,
                                       int N, int C, int H, int W) {
    const int ncb = (C + 7) / 8;
    const long long hw = (long long)H * W;
    const long long total = (long long)N * ncb * hw;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long pix = idx % hw;
        const int cb = (idx / hw) % ncb;
        const long long n = idx / (hw * ncb);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = 8 * cb + j;
            v[j] = c < C ? src[(n * C + c) * hw + pix] : 0.0f;
        }
        dst[idx] = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
    }
}

// CB8 bf16 -> NCHW fp32
__global__ void plc_cb8_to_nchw_kernel(const uint4* __restrict__ src, float* __restrict__ dst,
                                       int N, int C, int H, int W) {
    const int ncb = (C + 7) / 8;
    const long long hw = (long long)H * W;
    const long long total = (long long)N * ncb * hw;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long pix = idx % hw;
        const int cb = (idx / hw) % ncb;
        const long long n = idx / (hw * ncb);
        const uint4 u = src[idx];
        const float4 a = unpack_bf4(make_uint2(u.x, u.y)), b = unpack_bf4(make_uint2(u.z, u.w));
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = 8 * cb + j;
            if (c < C) dst[(n * C + c) * hw + pix] = v[j];
        }
    }
}

// NCHW fp32 <-> CB8 fp32 (cell state); dir 0: NCHW -> CB8 (channels padded with zeros to a multiple of 8),
// dir 1: CB8 -> NCHW.
__global__ void plc_state_layout_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                        int N, int C, int H, int W, int dir) {
    const long long hw = (long long)H * W;
    const int ncb = (C + 7) / 8;
    const long long total = (long long)N * ncb * 8 * hw;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        // idx enumerates the CB8 side: [n][cb][pix][j]
        const int j = idx & 7;
        const long long pix = (idx >> 3) % hw;
        const int cb = ((idx >> 3) / hw) % ncb;
        const long long n = (idx >> 3) / (hw * ncb);
        const bool real = 8 * cb + j < C;
        const long long nchw = (n * C + 8 * cb + j) * hw + pix;
        if (dir == 0) dst[idx] = real ? src[nchw] : 0.0f;
        else if (real) dst[nchw] = src[idx];
    }
}

// PyTorch 'bilinear', align_corners=False source coordinate for destination index d.
__device__ __forceinline__ void plc_bilinear_coord(int d, float scale, int in_size, int& i0, int& i1, float& w1) {
    float s = ((float)d + 0.5f) * scale - 0.5f;
    s = s < 0.0f ? 0.0f : s;
    i0 = (int)s;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    w1 = s - (float)i0;
}

// Resample an NCHW fp32 map to (Ho, Wo) and emit CB8 bf16.  mode 0: bilinear (align_corners
// False, scale = in/out), mode 1: legacy 'nearest' (src = floor(dst * in/out)).
__global__ void plc_resize_nchw_to_cb8_kernel(const float* __restrict__ src, uint4* __restrict__ dst,
                                              int N, int C, int Hi, int Wi, int Ho, int Wo, int mode,
                                              float sh, float sw) {
    const int ncb = (C + 7) / 8;
    const long long hwo = (long long)Ho * Wo, hwi = (long long)Hi * Wi;
    const long long total = (long long)N * ncb * hwo;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int ox = idx % Wo;
        const int oy = (idx / Wo) % Ho;
        const int cb = (idx / hwo) % ncb;
        const long long n = idx / (hwo * ncb);
        float v[8];
        if (mode == 0) {
            int y0, y1, x0, x1;
            float wy, wx;
            plc_bilinear_coord(oy, sh, Hi, y0, y1, wy);
            plc_bilinear_coord(ox, sw, Wi, x0, x1, wx);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = 8 * cb + j;
                if (c < C) {
                    const float* p = src + (n * C + c) * hwi;
                    const float a = p[(long long)y0 * Wi + x0], b = p[(long long)y0 * Wi + x1];
                    const float cc = p[(long long)y1 * Wi + x0], d = p[(long long)y1 * Wi + x1];
                    v[j] = (1.0f - wy) * ((1.0f - wx) * a + wx * b) + wy * ((1.0f - wx) * cc + wx * d);
                } else v[j] = 0.0f;
            }
        } else {
            int iy = (int)floorf((float)oy * sh), ix = (int)floorf((float)ox * sw);
            iy = iy < Hi - 1 ? iy : Hi - 1;
            ix = ix < Wi - 1 ? ix : Wi - 1;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = 8 * cb + j;
                v[j] = c < C ? src[(n * C + c) * hwi + (long long)iy * Wi + ix] : 0.0f;
            }
        }
        dst[idx] = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
    }
}

// Bilinear (align_corners False) resample of a CB8 bf16 tensor [N][ncb][Hi][Wi] -> [N][ncb][Ho][Wo].
__global__ void plc_resize_cb8_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int N, int ncb,
                                      int Hi, int Wi, int Ho, int Wo, float sh, float sw) {
    const long long hwo = (long long)Ho * Wo, hwi = (long long)Hi * Wi;
    const long long total = (long long)N * ncb * hwo;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int ox = idx % Wo;
        const int oy = (idx / Wo) % Ho;
        const long long plane = idx / hwo;
        int y0, y1, x0, x1;
        float wy, wx;
        plc_bilinear_coord(oy, sh, Hi, y0, y1, wy);
        plc_bilinear_coord(ox, sw, Wi, x0, x1, wx);
        const uint4* p = src + plane * hwi;
        const uint4 ua = p[(long long)y0 * Wi + x0], ub = p[(long long)y0 * Wi + x1];
        const uint4 uc = p[(long long)y1 * Wi + x0], ud = p[(long long)y1 * Wi + x1];
        const unsigned wa[4] = {ua.x, ua.y, ua.z, ua.w}, wb[4] = {ub.x, ub.y, ub.z, ub.w};
        const unsigned wc[4] = {uc.x, uc.y, uc.z, uc.w}, wd[4] = {ud.x, ud.y, ud.z, ud.w};
        unsigned o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float r[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int sft = 16 * e;
                const float a = bf2f((unsigned short)(wa[k] >> sft)), b = bf2f((unsigned short)(wb[k] >> sft));
                const float c = bf2f((unsigned short)(wc[k] >> sft)), d = bf2f((unsigned short)(wd[k] >> sft));
                r[e] = (1.0f - wy) * ((1.0f - wx) * a + wx * b) + wy * ((1.0f - wx) * c + wx * d);
            }
            o[k] = pack_bf2(r[0], r[1]);
        }
        dst[idx] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}
