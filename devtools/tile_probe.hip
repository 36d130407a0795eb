// What the memory system gives a conv-shaped access pattern, without any arithmetic: persistent workgroups
// walk (TH x TW)-pixel tiles of a [N][NCB][H][W] tensor of 16-byte units, read each tile with a one-pixel
// halo from all NCB channel planes and write the tile's NCB_OUT planes (8 or 16 bytes per lane, as the conv
// epilogues do).  For comparison with the engine's post / data-gradient convs (TH = 8, TW = 32, NCB = 4).
//   hipcc -O3 --offload-arch=gfx950 -o tile_probe tile_probe.hip && ./tile_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct Args {
    const uint4* in; uint4* out; uint4* sink;
    int N, H, W, ncb, ncb_out, th, tw, tiles_x, tiles_y, n_tiles, halo, st8, order;
};

__global__ __launch_bounds__(256) void tile_kernel(const Args a) {
    const long long plane = (long long)a.H * a.W;
    const int lh = a.th + 2 * a.halo, lw = a.tw + 2 * a.halo;
    const int units = a.ncb * lh * lw;
    for (int id = blockIdx.x; id < a.n_tiles; id += gridDim.x) {
        int t = id;
        int n, tx, ty;
        if (a.order == 0) { tx = t % a.tiles_x; t /= a.tiles_x; ty = t % a.tiles_y; n = t / a.tiles_y; }
        else { ty = t % a.tiles_y; t /= a.tiles_y; tx = t % a.tiles_x; n = t / a.tiles_x; }      // column-major within an image
        const int x0 = tx * a.tw - a.halo, y0 = ty * a.th - a.halo;
        uint4 acc = make_uint4(0, 0, 0, 0);
        const uint4* img = a.in + (long long)n * a.ncb * plane;
        for (int u0 = 0; u0 < units; u0 += 256 * 4) {
            uint4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int u = u0 + k * 256 + threadIdx.x;
                const int cb = u / (lh * lw), r = u - cb * lh * lw, ly = r / lw, lx = r - ly * lw;
                const int gy = y0 + ly, gx = x0 + lx;
                const bool ok = (u < units) & (gy >= 0) & (gy < a.H) & (gx >= 0) & (gx < a.W);
                v[k] = ok ? img[cb * plane + (long long)gy * a.W + gx] : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) { acc.x ^= v[k].x; acc.y += v[k].y; acc.z ^= v[k].z; acc.w += v[k].w; }
        }
        // stores: tile pixels x ncb_out planes
        uint4* oimg = a.out + (long long)n * a.ncb_out * plane;
        const int opix = a.th * a.tw;
        if (a.st8) {
            // 8 bytes per lane: lanes 0-31 the low half of 32 consecutive units, lanes 32-63 the high half
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, px = lane & 31;
            for (int w = wave; w < a.ncb_out * opix / 32; w += 4) {
                const int cb = w / (opix / 32), r = w - cb * (opix / 32);
                const int ly = r / (a.tw / 32), lx = (r - ly * (a.tw / 32)) * 32 + px;
                const int gy = ty * a.th + ly, gx = tx * a.tw + lx;
                if (gy < a.H && gx < a.W) {
                    uint2* p = (uint2*)(oimg + cb * plane + (long long)gy * a.W + gx) + half;
                    *p = make_uint2(acc.x + w, acc.y);
                }
            }
        } else {
            for (int u = threadIdx.x; u < a.ncb_out * opix; u += 256) {
                const int cb = u / opix, r = u - cb * opix, ly = r / a.tw, lx = r - ly * a.tw;
                const int gy = ty * a.th + ly, gx = tx * a.tw + lx;
                if (gy < a.H && gx < a.W) oimg[cb * plane + (long long)gy * a.W + gx] = make_uint4(acc.x + u, acc.y, acc.z, acc.w);
            }
        }
        if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) a.sink[0] = acc;
    }
}

int main(int argc, char** argv) {
    const int N = 40 * 4, H = 256, W = 256, NCB = 4;       // 4 tensors' worth, used in rotation: beyond the Infinity Cache
    const long long units = (long long)N * NCB * H * W;
    uint4 *in, *out, *sink;
    CK(hipMalloc(&in, units * 16)); CK(hipMalloc(&out, units * 16)); CK(hipMalloc(&sink, 16));
    CK(hipMemset(in, 1, units * 16)); CK(hipMemset(out, 0, units * 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%4s %4s %5s %4s %5s %5s %6s | %8s %8s %8s\n", "th", "tw", "halo", "st8", "order", "wg/cu", "ncbout", "us", "MB moved", "TB/s");
    struct Cfg { int th, tw, halo, st8, order, wgcu, ncb_out; };
    const Cfg cfgs[] = {
        {8, 32, 1, 1, 0, 3, 4}, {8, 32, 1, 0, 0, 3, 4}, {8, 32, 0, 1, 0, 3, 4}, {8, 32, 1, 1, 0, 2, 4}, {8, 32, 1, 1, 0, 4, 4}, {8, 32, 1, 1, 0, 8, 4},
        {16, 32, 1, 1, 0, 3, 4}, {8, 64, 1, 1, 0, 3, 4}, {8, 128, 1, 1, 0, 3, 4}, {8, 256, 1, 1, 0, 3, 4}, {4, 256, 1, 1, 0, 3, 4},
        {16, 64, 1, 1, 0, 3, 4}, {32, 32, 1, 1, 0, 3, 4}, {8, 32, 1, 1, 1, 3, 4},
        {8, 32, 1, 1, 0, 3, 0}, {8, 256, 1, 1, 0, 3, 0}, {8, 32, 0, 1, 0, 3, 0},
    };
    for (const Cfg& c : cfgs) {
        Args a;
        a.sink = sink; a.N = 40; a.H = H; a.W = W; a.ncb = NCB; a.ncb_out = c.ncb_out; a.th = c.th; a.tw = c.tw;
        a.halo = c.halo; a.st8 = c.st8; a.order = c.order;
        a.tiles_x = (W + c.tw - 1) / c.tw; a.tiles_y = (H + c.th - 1) / c.th; a.n_tiles = a.N * a.tiles_x * a.tiles_y;
        const int grid = 256 * c.wgcu < a.n_tiles ? 256 * c.wgcu : a.n_tiles;
        const long long per = 40ll * NCB * H * W;
        auto launch = [&](int rep) {
            a.in = in + (rep % 4) * per; a.out = out + (rep % 4) * per;
            hipLaunchKernelGGL(tile_kernel, dim3(grid), dim3(256), 0, 0, a);
        };
        for (int i = 0; i < 4; ++i) launch(i);
        CK(hipEventRecord(e0));
        const int reps = 20;
        for (int i = 0; i < reps; ++i) launch(i);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        const double rd = (double)a.n_tiles * NCB * (c.th + 2 * c.halo) * (c.tw + 2 * c.halo) * 16;
        const double wr = (double)40 * c.ncb_out * H * W * 16;
        printf("%4d %4d %5d %4d %5d %5d %6d | %8.1f %8.1f %8.2f\n", c.th, c.tw, c.halo, c.st8, c.order, c.wgcu, c.ncb_out, ms / reps * 1e3,
               (rd + wr) / 1e6, (rd + wr) / (ms / reps * 1e-3) / 1e12);
    }
    return 0;
}
