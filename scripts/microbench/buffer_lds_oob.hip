// Checks on hardware that `buffer_load_dwordx4 ... lds` writes ZEROS to LDS for lanes whose offset is out of the buffer's range
// (the conv kernels rely on it for padding taps / tails).  Build: hipcc --offload-arch=gfx950 -O3 buffer_lds_oob.hip -o /tmp/oob
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* x, float* y, int n) {
    __shared__ float s[256];
    for (int i = threadIdx.x; i < 256; i += 64) s[i] = -7.f;          // sentinel: must be overwritten by data or by zeros
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, n * 4, 0x00020000);
    int off = threadIdx.x * 16;
    if (threadIdx.x & 1) off = -1;                                     // out of range
    if ((threadIdx.x & 7) == 6) off = n * 4 - 8;                       // straddles the end: out of range as a whole
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)s, 16, off, 0, 0, 0);
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) y[i] = s[i];
}
int main() {
    const int n = 256;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = 1.f + i;
    float *x, *y;
    hipMalloc(&x, n * 4); hipMalloc(&y, n * 4);
    hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, x, y, n);
    std::vector<float> o(n);
    hipMemcpy(o.data(), y, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const bool oob = (l & 1) || (l & 7) == 6;
        for (int e = 0; e < 4; ++e) {
            const float want = oob ? 0.f : 1.f + l * 4 + e;
            if (o[l * 4 + e] != want) { if (bad < 8) printf("lane %d elem %d: got %g want %g\n", l, e, o[l * 4 + e], want); ++bad; }
        }
    }
    printf(bad ? "FAIL %d\n" : "OK: out-of-range lanes wrote zeros\n", bad);
    return bad != 0;
}
