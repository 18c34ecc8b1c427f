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

// Opt-in per-kernel-class timing with HIP events on the launch stream (used by bench.py's roofline leg).
enum ProfClass { PROF_CONV128 = 0, PROF_CONV64 = 1, PROF_SCORE_HEAD = 2, PROF_CONV128x64 = 3,
                 // HBM-bound kernels: `bytes` = algorithmic bytes of the launch (operands read once + results written once)
                 PROF_MANO_FK = 4, PROF_OBJ_PHYSICS = 5, PROF_HAND_FUSE = 6, PROF_ROI_ALIGN = 7, PROF_RESIZE = 8, PROF_WINOGRAD = 9,
                 // widened rows: weight-gradient TN GEMM by tile class (train.py), the persistent pseudo-force optimiser (force_optim.py)
                 PROF_WGRAD64 = 10, PROF_WGRAD128 = 11, PROF_FORCE_OPTIM = 12, PROF_POSE_ENC = 13, PROF_NCLASS = 14 };
bool prof_on(int cls);
void prof_record(int cls, hipEvent_t start, hipEvent_t stop, double flops, double bytes);
struct ProfScope {
    int cls; hipStream_t s; double flops, bytes; bool on; hipEvent_t e0, e1;
    ProfScope(int c, hipStream_t st, double f, double b = 0) : cls(c), s(st), flops(f), bytes(b), on(c >= 0 && prof_on(c)) {        // c < 0: never timed
        if (on) { (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventRecord(e0, s); }
    }
    ~ProfScope() { if (on) { (void)hipEventRecord(e1, s); prof_record(cls, e0, e1, flops, bytes); } }
};
}  // namespace vpho

#define VPHO_REQUIRE(cond, ...) do { if (!(cond)) return vpho::fail(__VA_ARGS__); } while (0)
#define VPHO_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return vpho::fail("%s: %s", #call, hipGetErrorString(e_)); } while (0)

// Barrier that publishes `buffer_load ... lds` (LDS-DMA) tiles to the other waves of the workgroup: every wave first waits for ITS
// OWN outstanding fills.  __syncthreads() alone is not enough: the compiler places the vmcnt wait of an LDS-DMA by what the issuing
// wave itself reads afterwards and may leave a barrier with the fill still in flight -- other waves then read the tile too early.
#define VPHO_SYNC_LDS_DMA() do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); } while (0)

// F.interpolate(mode='bilinear', align_corners=False): src = max((dst+0.5)*scale-0.5, 0), scale = in/out
__device__ inline void lin_src(int dst, float scale, int in_size, int& i0, int& i1, float& l1) {
    float s = ((float)dst + 0.5f) * scale - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = s - (float)i0;
}

// In-kernel clock stamps (MI355X_MICROARCH.md, DVFS item 6): DIAGNOSTIC builds only (scripts/build_stamps.sh compiles the chip-filling
// kernels with -DVPHO_CLOCK_STAMPS into scripts/_ab/libvpho_hip_stamps.so; the product library never defines the macro and contains
// no stamp -- tests/test_abi.py).  A workgroup stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) once in front of and once
// behind its main loop; the differences go to a buffer of their own that nothing else reads.  clock = d memtime / d memrealtime x 100 MHz.
#ifdef VPHO_CLOCK_STAMPS
#define VPHO_STAMP_SLOTS 65536
#define VPHO_STAMP_DECL(name)                                                                                                      \
    __device__ unsigned long long name##_stamps[2 * VPHO_STAMP_SLOTS];                                                             \
    extern "C" __attribute__((visibility("default"))) int vpho_diag_stamps_##name(unsigned long long* host, int slots, int clear) { \
        if (host && hipMemcpyFromSymbol(host, HIP_SYMBOL(name##_stamps), (size_t)slots * 16) != hipSuccess) return 1;               \
        if (clear) { void* p = nullptr; if (hipGetSymbolAddress(&p, HIP_SYMBOL(name##_stamps)) != hipSuccess || hipMemset(p, 0, sizeof(name##_stamps)) != hipSuccess) return 2; } \
        return 0;                                                                                                                  \
    }
#define VPHO_STAMP_BEGIN()                                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                                             \
    const unsigned long long st_t0_ = __builtin_amdgcn_s_memtime(), st_r0_ = __builtin_amdgcn_s_memrealtime();                     \
    __builtin_amdgcn_sched_barrier(0)
#define VPHO_STAMP_END(name, slot)                                                                                                 \
    do {                                                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
        const unsigned long long st_t1_ = __builtin_amdgcn_s_memtime(), st_r1_ = __builtin_amdgcn_s_memrealtime();                 \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
        if (threadIdx.x == 0) { const unsigned sl_ = (unsigned)(slot) & (VPHO_STAMP_SLOTS - 1);                                    \
                                name##_stamps[2 * sl_] = st_t1_ - st_t0_; name##_stamps[2 * sl_ + 1] = st_r1_ - st_r0_; }         \
    } while (0)
#else
#define VPHO_STAMP_DECL(name)
#define VPHO_STAMP_BEGIN() do {} while (0)
#define VPHO_STAMP_END(name, slot) do {} while (0)
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
