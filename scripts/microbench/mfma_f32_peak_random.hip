// As mfma_f32_peak.hip, but the MFMA operands are 16 pseudo-random values per lane cycled through the loop (realistic
// operand toggling).  Calibrates the MFMA rate the chip sustains on random fp32 data under its power limit.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
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
    // MI355X_MICROARCH.md, DVFS item 6: the clock is read after >= 2 s of back-to-back launches, median over workgroups
    const int blocks = 256 * 2, iters = 12500;
    float* out; unsigned long long* clk;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int zero = 1; zero >= 0; --zero) {
        double elapsed = 0; float ms = 0; int launches = 0;
        while (elapsed < 2500.0) {
            hipEventRecord(e0);
            for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, clk, zero);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            elapsed += ms; launches += 20;
        }
        std::vector<unsigned long long> h(blocks * 2);
        hipMemcpy(h.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
        std::vector<double> c;
        for (int b = 0; b < blocks; ++b) if (h[2 * b + 1]) c.push_back((double)h[2 * b] / h[2 * b + 1] * 0.1);
        std::sort(c.begin(), c.end());
        double flop = (double)blocks * 4 * iters * 16.0 * 32 * 32 * 2 * 2;
        printf("bare v_mfma_f32_32x32x2_f32 loop, %s operands: after %.1f s of back-to-back launches (%d): %.3f ms/launch  %.1f TFLOP/s  in-kernel clock "
               "median %.3f GHz (min %.3f, max %.3f over %zu workgroups)\n", zero ? "zero  " : "random", elapsed * 1e-3, launches, ms / 20, flop / (ms / 20) / 1e9,
               c[c.size() / 2], c.front(), c.back(), c.size());
    }
    return 0;
}
