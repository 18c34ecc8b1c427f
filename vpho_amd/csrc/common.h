// Shared host-side helpers for the vpho_hip C ABI (error slot, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdarg>

namespace vpho {
char* err_slot();
int fail(const char* fmt, ...);
inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("%s: %s", what, hipGetErrorString(e));
    return 0;
}
}  // namespace vpho

#define VPHO_REQUIRE(cond, ...) do { if (!(cond)) return vpho::fail(__VA_ARGS__); } while (0)
#define VPHO_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return vpho::fail("%s: %s", #call, hipGetErrorString(e_)); } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
