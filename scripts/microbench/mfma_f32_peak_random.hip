// As mfma_f32_peak.hip, but the MFMA operands are 16 pseudo-random values per lane cycled through the loop (realistic
// operand toggling).  Calibrates the MFMA rate the chip sustains on random fp32 data under its power limit.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ inline float rnd(unsigned s) {
    s ^= s << 13; s ^= s >> 17; s ^= s << 5; s *= 2654435761u; s ^= s >> 15;
    return (float)(s & 0xFFFFFF) / 8388608.0f - 1.0f;
}
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* clk, int zero) {
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float r[16];
    for (int i = 0; i < 16; ++i) r[i] = zero ? 0.f : rnd((blockIdx.x * 256 + threadIdx.x) * 16 + i + 1);
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; u += 4) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(r[u], r[(u + 5) & 15], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(r[u + 1], r[(u + 6) & 15], a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(r[u + 2], r[(u + 7) & 15], a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(r[u + 3], r[(u + 8) & 15], a3, 0, 0, 0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int e = 0; e < 16; ++e) s += a0[e] + a1[e] + a2[e] + a3[e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}
int main() {
    const int blocks = 256 * 2, iters = 12500;
    float* out; unsigned long long* clk;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int zero = 1; zero >= 0; --zero)
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, clk, zero);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(blocks * 2);
            hipMemcpy(h.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
            double flop = (double)blocks * 4 * iters * 16.0 * 32 * 32 * 2 * 2;
            printf("%s operands, rep %d: %.2f ms  %.1f TFLOP/s  in-kernel clock %.3f GHz\n", zero ? "zero  " : "random", rep, ms, flop / ms / 1e9,
                   (double)h[0] / h[1] * 0.1);
        }
    return 0;
}
