// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands on gfx950: which (lane, byte) of the A / B
// fragments is which (row / column, k), and what the scale operands do.  Exact small-integer data, so any
// wrong hypothesis shows up as a mismatch.  Build: hipcc --offload-arch=gfx950 -O2 -o probe devtools/probe_mfma_fp8.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void probe(const unsigned char* a, const unsigned char* b, float* c, int scale_a, int scale_b) {
    const int lane = threadIdx.x;
    v8i av, bv;
    const int* ap = (const int*)(a + lane * 32);
    const int* bp = (const int*)(b + lane * 32);
    for (int i = 0; i < 8; ++i) { av[i] = ap[i]; bv[i] = bp[i]; }
    v16f acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 0, 0, 0, scale_a, 0, scale_b);
    for (int i = 0; i < 16; ++i) c[lane * 16 + i] = acc[i];
}

static unsigned char e4m3(int v) {   // small integers -8..8, exactly representable
    if (v == 0) return 0;
    unsigned char s = v < 0 ? 0x80 : 0;
    int m = abs(v), e = 0;
    while ((1 << (e + 1)) <= m) ++e;             // m in [2^e, 2^(e+1))
    int frac = ((m << 3) >> e) & 7;              // 3 mantissa bits (exact for m <= 15 with e<=3)
    return s | ((e + 7) << 3) | frac;
}

int main() {
    int A[32][64], B[64][32];
    for (int r = 0; r < 32; ++r) for (int k = 0; k < 64; ++k) A[r][k] = (r + 2 * k) % 7 - 3;
    for (int k = 0; k < 64; ++k) for (int cc = 0; cc < 32; ++cc) B[k][cc] = (3 * cc + k) % 5 - 2;
    double C[32][32];
    for (int r = 0; r < 32; ++r) for (int cc = 0; cc < 32; ++cc) { double s = 0; for (int k = 0; k < 64; ++k) s += A[r][k] * B[k][cc]; C[r][cc] = s; }
    unsigned char ha[64 * 32], hb[64 * 32];
    unsigned char *da, *db; float* dc; float hc[64 * 16];
    hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dc, sizeof(hc));
    // hypotheses for k(lane half h, byte j): 0: 32h + j;  1: 16h + j (j<16), 32 + 16h + (j-16);  2: 2j + h;  3: 8*(4*(j/8)... ) blocked by 8: 16*(j/8) + 8h + j%8
    for (int hyp = 0; hyp < 4; ++hyp) {
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) {
            const int h = l >> 5, r = l & 31;
            int k;
            if (hyp == 0) k = 32 * h + j;
            else if (hyp == 1) k = j < 16 ? 16 * h + j : 32 + 16 * h + (j - 16);
            else if (hyp == 2) k = 2 * j + h;
            else k = 16 * (j / 8) + 8 * h + j % 8;
            ha[l * 32 + j] = e4m3(A[r][k]);
            hb[l * 32 + j] = e4m3(B[k][r]);
        }
        hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice);
        hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
        for (int sc = 0; sc < 3; ++sc) {
            const int sa = sc == 0 ? 0x7F7F7F7F : (sc == 1 ? 0x7979797 * 0 + 0x79797979 : 0x7F7F7F7F);
            const int sb = sc == 2 ? 0x81818181 : 0x7F7F7F7F;
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dc, sa, sb);
            hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost);
            // C/D map: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
            int bad = 0; double ratio = 0; int nr = 0;
            for (int l = 0; l < 64; ++l) for (int i = 0; i < 16; ++i) {
                const int col = l & 31, row = (i & 3) + 8 * (i >> 2) + 4 * (l >> 5);
                const double want = C[row][col];
                if (want != 0) { ratio += hc[l * 16 + i] / want; ++nr; }
                if (fabs(hc[l * 16 + i] - want) > 1e-3) ++bad;
            }
            printf("hypothesis %d scale case %d (a=%08x b=%08x): mismatches %d of 1024, mean ratio got/want %.6f\n", hyp, sc, sa, sb, bad, ratio / nr);
        }
    }
    return 0;
}
