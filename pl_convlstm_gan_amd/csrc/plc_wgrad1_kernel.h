// Weight gradient of a 3x3 conv with 32 input channels and ONE output channel (the generator's
// last conv, reference generator.py:67-71):
//     dW[c][ky][kx] = sum over images and pixels  dy[y][x] * in[c][y+ky-1][x+kx-1],   db = sum dy.
// On the matrix kernel this conv fills one of 32 MFMA rows and needs dy copied into 16-byte
// units first; here it is a streaming reduction on the vector ALUs: every lane owns one 8-channel
// unit of the conv input, multiplies it with the 3x3 neighbourhood of dy (fp32, read from the
// loss gradient as it is) into 72 running sums, and the sums of a wave's 64 lanes are folded once
// per workgroup.  The conv input is read exactly once (HBM-bound: 64 B per pixel), dy comes through
// an LDS tile with its halo.  The result goes to a one-row slab per workgroup in the column order of
// the matrix kernel's slabs, so the common slab reduction turns it into dW and db.
#pragma once
#include "plc_conv_kernel.h"

constexpr int W1_ROWS = 8;          // rows of one strip (a strip is W1_ROWS x 64 pixels of one image)
constexpr int W1_COLS = 320;        // slab row: 9 taps x 32 channels, the bias sum at 288, padded to 32
constexpr int W1_MAX_BLOCKS = 768;  // three 4-wave workgroups per CU: what 150 registers per lane keep resident

struct Wgrad1Args {
    PlcTensor x;    // conv input, bf16 CB8 with 4 channel blocks per image
    PlcTensor dy;   // gradient of the conv output, one fp32 plane per image
    int n_in, H, W, strips_x, strips_y, n_strips;
    float* slab;    // [gridDim.x][W1_COLS]
};

typedef float plc_f32x2 __attribute__((ext_vector_type(2)));

// One level of the transposing reduction: lane pairs that differ in `bit` split the N running
// values between them (the lane with the bit clear keeps the lower half) and add the partner's
// copy of the half they keep; after all six levels every value's 64-lane sum sits in one lane.
template <int N>
__device__ __forceinline__ void plc_fold_level(float* v, int bit, int lane, int& base, int& n_valid) {
    constexpr int HALF = (N + 1) / 2;
    const bool up = (lane & bit) != 0;
#pragma unroll
    for (int i = 0; i < HALF; ++i) {
        const float lo = v[i];
        const float hi = i + HALF < N ? v[i + HALF] : 0.0f;
        const float send = up ? lo : hi;
        const float keep = up ? hi : lo;
        v[i] = keep + __shfl_xor(send, bit);
    }
    base += up ? HALF : 0;
    n_valid = up ? (n_valid > HALF ? n_valid - HALF : 0) : (n_valid < HALF ? n_valid : HALF);
}

__global__ __launch_bounds__(256) void plc_wgrad_cout1_kernel(const Wgrad1Args a) {
    __shared__ float dyt[W1_ROWS + 2][66];   // dy rows y0-1 .. y0+W1_ROWS, columns x0-1 .. x0+64, zero outside the image
    const int lane = threadIdx.x & 63, cb = threadIdx.x >> 6;
    const int H = a.H, W = a.W;

    plc_f32x2 acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t][j] = plc_f32x2{0.0f, 0.0f};
    float bsum = 0.0f;

    for (int s = blockIdx.x; s < a.n_strips; s += gridDim.x) {
        const int xb = s % a.strips_x, yb = (s / a.strips_x) % a.strips_y, n = s / (a.strips_x * a.strips_y);
        const int x0 = xb * 64, y0 = yb * W1_ROWS;
        // conv input of this lane's column, all rows of the strip in flight at once
        const char* xi = plc_img(a.x, n, a.n_in);
        const int xl = x0 + lane;
        uint4 raw[W1_ROWS];
        bool ok[W1_ROWS];
#pragma unroll
        for (int r = 0; r < W1_ROWS; ++r) {
            ok[r] = xl < W && y0 + r < H;
            const char* p = xi + (((long long)cb * H + (y0 + r)) * W + xl) * 16;
            raw[r] = plc_gload16(ok[r] ? p : xi);
        }
        __syncthreads();   // the previous strip's window reads are done
        {
            const float* dp = (const float*)plc_img(a.dy, n, a.n_in);
            for (int e = threadIdx.x; e < (W1_ROWS + 2) * 66; e += 256) {
                const int ty = e / 66, tx = e % 66;
                const int gy = y0 - 1 + ty, gx = x0 - 1 + tx;
                const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
                const float v = plc_gload4((const char*)(in ? dp + (long long)gy * W + gx : dp));
                dyt[ty][tx] = in ? v : 0.0f;
            }
        }
        __syncthreads();
        // window d[j][i] = dy(y - 1 + j, x - 1 + i); tap (ky, kx) pairs in(y, x) with dy(y - ky + 1, x - kx + 1)
        float d[3][3];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 3; ++i) d[j + 1][i] = dyt[j][lane + i];
#pragma unroll
        for (int r = 0; r < W1_ROWS; ++r) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                d[0][i] = d[1][i];
                d[1][i] = d[2][i];
                d[2][i] = dyt[r + 2][lane + i];
            }
            bsum += d[1][1];
            plc_f32x2 xv[4];
            const unsigned rw[4] = {raw[r].x, raw[r].y, raw[r].z, raw[r].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned w = ok[r] ? rw[j] : 0u;
                xv[j] = plc_f32x2{__builtin_bit_cast(float, w << 16), __builtin_bit_cast(float, w & 0xffff0000u)};
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float dv = d[2 - ky][2 - kx];
                    // The broadcast pair is MATERIALISED (two moves) instead of being left to the instruction's
                    // operand select.  The compiler otherwise encodes the taps whose dy value sits in the odd
                    // register of a ds_read2 pair (kx == 1) as v_pk_fma_f32 ... op_sel:[0,1,0], and on the MI355X
                    // exactly those 96 of the 288 packed FMAs gave sums that varied from run to run (a few
                    // entries per slab row, 1e-2 relative) whenever a conv kernel ran on the chip at the same
                    // time -- never alone, never beside a memset, never for the op_sel_hi:[1,0,1] (even
                    // register) or plain forms.  With the pair in its own registers the kernel is bit-stable
                    // beside any other kernel and as fast as before (it is bound by its reads).
                    plc_f32x2 d2 = plc_f32x2{dv, dv};
                    asm volatile("" : "+v"(d2));
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[ky * 3 + kx][j] = __builtin_elementwise_fma(xv[j], d2, acc[ky * 3 + kx][j]);
                }
        }
    }

    // ---- 64 lanes x 72 sums -> 72 sums, each in one lane -----------------------------------------
    float v[72];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[t * 8 + 2 * j] = acc[t][j][0];
            v[t * 8 + 2 * j + 1] = acc[t][j][1];
        }
    int base = 0, n_valid = 72;
    plc_fold_level<72>(v, 1, lane, base, n_valid);
    plc_fold_level<36>(v, 2, lane, base, n_valid);
    plc_fold_level<18>(v, 4, lane, base, n_valid);
    plc_fold_level<9>(v, 8, lane, base, n_valid);
    plc_fold_level<5>(v, 16, lane, base, n_valid);
    plc_fold_level<3>(v, 32, lane, base, n_valid);
    float* slab = a.slab + (long long)blockIdx.x * W1_COLS;
#pragma unroll
    for (int i = 0; i < 2; ++i)
        if (i < n_valid) {
            const int k = base + i;                       // k = tap * 8 + channel within the block
            slab[(k >> 3) * 32 + cb * 8 + (k & 7)] = v[i];
        }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) bsum += __shfl_xor(bsum, m);
    if (threadIdx.x == 0) slab[288] = bsum;
}
