// Rounding behaviour of the MFMA accumulators on gfx950: is the fp32 accumulation of the 16-bit-input instructions
// (v_mfma_f32_32x32x16_bf16 / _f16) unbiased?  512 independent waves each accumulate S k-steps of pseudo-random operands that are EXACT in the
// 16-bit format (so the only rounding is the instruction's own), against an fp64 evaluation of the same sum on the device.
// Reported per instruction: rms error, and the MEAN SIGNED error, both relative to the mean |result| -- a rounding that is
// unbiased has |mean signed| << rms / sqrt(#outputs); a floor (toward -inf) shows up as a negative mean whatever the
// operand signs.  v_mfma_f32_16x16x4_f32 (the exact-f32 kernels) as control.
//   hipcc --offload-arch=gfx950 -O2 -o mfma_bias tools/micro/mfma_bias.hip && ./mfma_bias
// Operand layout (guide: MI355X_MICROARCH.md, and the kernels of this repo): 32x32x16: lane l holds row/col l & 31 and the 8
// consecutive k = 8 (l >> 5) + p; D register r of lane l is row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
// value in (-1, 1) with `bits` significant bits, exact in bf16 (bits <= 8) / fp16 (bits <= 11); spread > 0: times 2^-(0..spread)
__host__ __device__ inline float val(unsigned seed, int bits, int spread) {
    const unsigned h = mix(seed);
    const int m = (int)(h & ((1u << bits) - 1)) - (1 << (bits - 1));
    float v = (float)m / (float)(1 << (bits - 1));
    if (spread) v = ldexpf(v, -(int)((h >> 20) % (unsigned)(spread + 1)));
    return v;
}
__device__ unsigned g_salt() { return blockIdx.x * 0x9e3779b9u; }
__device__ inline float a_val(int step, int i, int k, int bits, int spread) { return val((0x10000000u + step * 65536 + i * 256 + k) ^ g_salt(), bits, spread); }
__device__ inline float b_val(int step, int k, int j, int bits, int spread) { return val((0x50000000u + step * 65536 + j * 256 + k) ^ g_salt(), bits, spread); }

template <int KIND>   // 0 bf16, 1 f16
__global__ void k16(int steps, int spread, int flip, float* out, double* ref) {
    const int l = threadIdx.x, c = l & 31, kg = l >> 5;
    const int bits = KIND ? 11 : 8;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    double d[16];
    for (int r = 0; r < 16; ++r) d[r] = 0.0;
    const float sg = (flip && (c & 1)) ? -1.f : 1.f;          // flip: odd columns carry -B (the kernels' sign alternation)
    for (int s = 0; s < steps; ++s) {
        if (KIND == 0) {
            bf16x8 a, b;
            for (int p = 0; p < 8; ++p) { a[p] = (__bf16)a_val(s, c, 8 * kg + p, bits, spread); b[p] = (__bf16)(sg * b_val(s, 8 * kg + p, c, bits, spread)); }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        } else {
            f16x8 a, b;
            for (int p = 0; p < 8; ++p) { a[p] = (_Float16)a_val(s, c, 8 * kg + p, bits, spread); b[p] = (_Float16)(sg * b_val(s, 8 * kg + p, c, bits, spread)); }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        }
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * kg;
            for (int k = 0; k < 16; ++k) d[r] += (double)a_val(s, row, k, bits, spread) * (double)b_val(s, k, c, bits, spread);
        }
    }
    for (int r = 0; r < 16; ++r) { out[blockIdx.x * 1024 + l * 16 + r] = sg * acc[r]; ref[blockIdx.x * 1024 + l * 16 + r] = d[r]; }
}

// control: v_mfma_f32_16x16x4_f32: lane l: A row l & 15, k = l >> 4; B col l & 15, k = l >> 4; D reg r: row 4 (l >> 4) + r, col l & 15
__global__ void k32(int steps, int spread, float* out, double* ref) {
    const int l = threadIdx.x, c = l & 15, kg = l >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    double d[4] = {0, 0, 0, 0};
    for (int s = 0; s < steps; ++s) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a_val(s, c, kg, 11, spread), b_val(s, kg, c, 11, spread), acc, 0, 0, 0);
        for (int r = 0; r < 4; ++r)
            for (int k = 0; k < 4; ++k) d[r] += (double)a_val(s, 4 * kg + r, k, 11, spread) * (double)b_val(s, k, c, 11, spread);
    }
    for (int r = 0; r < 4; ++r) { out[blockIdx.x * 256 + l * 4 + r] = acc[r]; ref[blockIdx.x * 256 + l * 4 + r] = d[r]; }
}

static void report(const char* name, const std::vector<float>& o, const std::vector<double>& r) {
    double se = 0, s2 = 0, sa = 0;
    for (size_t i = 0; i < o.size(); ++i) { const double e = (double)o[i] - r[i]; se += e; s2 += e * e; sa += fabs(r[i]); }
    const double n = (double)o.size(), ma = sa / n;
    printf("%-34s rms err / mean|D| %9.2e   mean signed err / mean|D| %+9.2e   (noise floor of the mean %8.1e)\n", name,
           sqrt(s2 / n) / ma, se / n / ma, sqrt(s2 / n) / ma / sqrt(n));
}

int main() {
    float* out; double* ref;
    const int NB = 512;                                        // independent waves (different operands): 524288 outputs per line
    hipMalloc(&out, NB * 1024 * 4); hipMalloc(&ref, NB * 1024 * 8);
    std::vector<float> o(NB * 1024); std::vector<double> r(NB * 1024);
    for (int spread : {0, 8})
        for (int steps : {8, 24, 256}) {
            char name[96];
            for (int kind = 0; kind < 2; ++kind)
                for (int flip = 0; flip < 2; ++flip) {
                    if (kind == 0) hipLaunchKernelGGL(k16<0>, dim3(NB), dim3(64), 0, 0, steps, spread, flip, out, ref);
                    else hipLaunchKernelGGL(k16<1>, dim3(NB), dim3(64), 0, 0, steps, spread, flip, out, ref);
                    hipMemcpy(o.data(), out, NB * 1024 * 4, hipMemcpyDeviceToHost); hipMemcpy(r.data(), ref, NB * 1024 * 8, hipMemcpyDeviceToHost);
                    snprintf(name, sizeof name, "%s steps %3d spread 2^-%d%s", kind ? "32x32x16_f16 " : "32x32x16_bf16", steps, spread, flip ? " alt" : "");
                    report(name, o, r);
                }
            hipLaunchKernelGGL(k32, dim3(NB), dim3(64), 0, 0, steps, spread, out, ref);
            o.resize(NB * 256); r.resize(NB * 256);
            hipMemcpy(o.data(), out, NB * 256 * 4, hipMemcpyDeviceToHost); hipMemcpy(r.data(), ref, NB * 256 * 8, hipMemcpyDeviceToHost);
            snprintf(name, sizeof name, "16x16x4_f32   steps %3d spread 2^-%d", steps, spread);
            report(name, o, r);
            o.resize(NB * 1024); r.resize(NB * 1024);
        }
    return 0;
}
