// What do row_ror:8 + v_permlane16_swap + v_permlane32_swap do to a wave?  (conv_igemm.hip::sum_lane_bits_345)
//   hipcc --offload-arch=gfx950 -O3 scripts/microbench/lane_sum.hip -o /tmp/lane_sum && /tmp/lane_sum
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* p, float* q16, float* q32, float* qd) {
    float x = p[threadIdx.x];
    qd[threadIdx.x] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xF, 0xF, false));
    const unsigned a = __builtin_bit_cast(unsigned, x);
    // (the builtin's second result comes back equal to the first with this hipcc, whatever the operands: inline assembly, both registers in / out)
    unsigned a1 = a, b1 = a;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a1), "+v"(b1));
    q16[threadIdx.x] = __builtin_bit_cast(float, a1); q16[64 + threadIdx.x] = __builtin_bit_cast(float, b1);
    unsigned a2 = a, b2 = a;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a2), "+v"(b2));
    q32[threadIdx.x] = __builtin_bit_cast(float, a2); q32[64 + threadIdx.x] = __builtin_bit_cast(float, b2);
}
int main() {
    float h[64], o[5 * 64]; float* d; 
    for (int i = 0; i < 64; ++i) h[i] = (float)i;
    hipMalloc(&d, 6 * 64 * 4); hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, d + 64, d + 192, d + 320);
    hipMemcpy(o, d + 64, 5 * 64 * 4, hipMemcpyDeviceToHost);
    const char* names[5] = {"permlane16_swap [0]", "permlane16_swap [1]", "permlane32_swap [0]", "permlane32_swap [1]", "dpp row_ror:8"};
    for (int r = 0; r < 5; ++r) { printf("%-20s", names[r]); for (int i = 0; i < 64; ++i) printf(" %2.0f", o[r * 64 + i]); printf("\n"); }
    return 0;
}
