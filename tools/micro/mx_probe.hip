// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 (MX fp8) on gfx950: operand layout, E8M0 scale semantics.  Result (MI355X): with lane
// (i = l & 31, kg = l >> 5) holding bytes p = 0..31 <-> K = 32 kg + p of row i (A) / column i (B), format code 0 = e4m3 and the
// scale byte 127 the product is exact; scale byte 128 doubles it, 125 quarters it (byte 0 of the scale VGPR with opsel 0).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const unsigned char* A, const unsigned char* B, float* D, int sa, int sb) {
    const int l = threadIdx.x, i = l & 31, kg = l >> 5;
    v8i a, b;
    for (int q = 0; q < 8; ++q) {
        unsigned wa = 0, wb = 0;
        for (int t = 0; t < 4; ++t) {
            const int k = 32 * kg + 4 * q + t;
            wa |= (unsigned)A[i * 64 + k] << (8 * t);          // A[i][k]
            wb |= (unsigned)B[k * 32 + i] << (8 * t);          // B[k][j = i]
        }
        a[q] = (int)wa; b[q] = (int)wb;
    }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * kg) * 32 + i] = c[r];   // row, col = lane & 31
}
static float dec(unsigned char v) { int s = v >> 7, e = (v >> 3) & 15, m = v & 7; float x = e ? ldexpf(1.f + m / 8.f, e - 7) : ldexpf(m / 8.f, -6); return s ? -x : x; }
int main() {
    unsigned char hA[32 * 64], hB[64 * 32];
    const unsigned char vals[8] = {0x38, 0x40, 0x30, 0x3C, 0x44, 0xB8, 0x28, 0x00};   // 1, 2, .5, 1.5, 3, -1, .25, 0
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 64; ++k) hA[i * 64 + k] = vals[(i * 7 + k * 3 + (k >> 4)) % 8];
    for (int k = 0; k < 64; ++k) for (int j = 0; j < 32; ++j) hB[k * 32 + j] = vals[(k * 5 + j * 11 + (j >> 3)) % 8];
    unsigned char *dA, *dB; float* dD; float hD[1024];
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    const int scales[4][2] = {{127, 127}, {128, 127}, {127, 125}, {0x7f7f7f80, 127}};
    for (int t = 0; t < 4; ++t) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, scales[t][0], scales[t][1]);
        hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        double worst = 0, ratio = 0; int cnt = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            double ref = 0; for (int k = 0; k < 64; ++k) ref += (double)dec(hA[i * 64 + k]) * dec(hB[k * 32 + j]);
            if (fabs(ref) > 1e-9) { ratio += hD[i * 32 + j] / ref; ++cnt; }
            worst = fmax(worst, fabs(hD[i * 32 + j] - ref));
        }
        printf("scale_a %08x scale_b %08x: max|D - ref| = %g, mean D/ref = %g, D[3][5] = %g\n", scales[t][0], scales[t][1], worst, ratio / cnt, hD[3 * 32 + 5]);
    }
    return 0;
}
