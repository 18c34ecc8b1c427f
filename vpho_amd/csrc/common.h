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
// Raises a kernel's dynamic-LDS limit (hipFuncAttributeMaxDynamicSharedMemorySize) once per (kernel, device): the attribute belongs to
// the device's code object, so a process-wide flag would leave a second device's launches without it.
int dynamic_lds(const void* kernel, int bytes);
}  // namespace vpho

#define VPHO_REQUIRE(cond, ...) do { if (!(cond)) return vpho::fail(__VA_ARGS__); } while (0)
#define VPHO_DYN_LDS(kernel, bytes) do { if (vpho::dynamic_lds(reinterpret_cast<const void*>(kernel), (int)(bytes))) return 1; } while (0)
#define VPHO_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return vpho::fail("%s: %s", #call, hipGetErrorString(e_)); } while (0)

// Barrier that publishes `buffer_load ... lds` (LDS-DMA) tiles to the other waves of the workgroup: every wave first waits for ITS
// OWN outstanding fills.  __syncthreads() alone is not enough: the compiler places the vmcnt wait of an LDS-DMA by what the issuing
// wave itself reads afterwards and may leave a barrier with the fill still in flight -- other waves then read the tile too early.
#define VPHO_SYNC_LDS_DMA() do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); } while (0)
// Workgroup barrier WITHOUT __syncthreads()' memory fence, for the persistent kernels: with global stores of the previous tile (or the next
// tile's LDS-DMA fills) still in flight the fence makes the compiler put `s_waitcnt vmcnt(0)` in front of the barrier -- the wait the
// counted vmcnt is there to avoid.  What the barrier has to publish here is LDS data only: this wave's LDS operations are complete
// (lgkmcnt(0)), LDS-DMA tiles are covered by the caller's own counted vmcnt wait.  Global memory is NOT ordered by it.
#define VPHO_BARRIER_LDS_ONLY() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// Wave priority around the main loop of the chip-filling kernels (two workgroups share a CU: one's epilogue -- vector ALU, LDS, stores --
// runs beside the other's matrix loop; the SIMD's arbiter otherwise serves the OLDER wave first, i.e. the one in its epilogue).
#ifndef VPHO_MAINLOOP_PRIO
#define VPHO_MAINLOOP_PRIO 0
#endif
#if VPHO_MAINLOOP_PRIO
#define VPHO_PRIO_MAIN() __builtin_amdgcn_s_setprio(VPHO_MAINLOOP_PRIO)
#define VPHO_PRIO_REST() __builtin_amdgcn_s_setprio(0)
#else
#define VPHO_PRIO_MAIN() ((void)0)
#define VPHO_PRIO_REST() ((void)0)
#endif

// F.interpolate(mode='bilinear', align_corners=False): src = max((dst+0.5)*scale-0.5, 0), scale = in/out
__device__ inline void lin_src(int dst, float scale, int in_size, int& i0, int& i1, float& l1) {
    float s = ((float)dst + 0.5f) * scale - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = s - (float)i0;
}

// In-kernel stamps (MI355X_MICROARCH.md, DVFS item 6; cdna_hip_programming.md 7, "In-kernel stamps"): DIAGNOSTIC builds only
// (scripts/build_stamps.sh compiles the chip-filling kernels with -DVPHO_CLOCK_STAMPS into scripts/_ab/libvpho_hip_stamps.so; the
// product library never defines the macro and contains no stamp -- tests/test_abi.py).  A workgroup reads s_memtime (shader cycles) at
// up to six points of its life -- [0] entry, [1] set-up done / first fills requested, [2] main loop begins (first stage landed),
// [3] main loop ends, [4] epilogue done (stores issued), [5] free -- and s_memrealtime (100 MHz, one counter for the whole chip) at
// entry and exit, plus the CU it ran on; the record goes to a buffer of its own that nothing else reads.
//   clock = ([3] - [2]) / (realtime over the same span) is reported from [0]..[4] / realtime entry..exit;  the realtime stamps of all
//   workgroups of a launch give the launch's timeline per CU (scripts/inkernel_clock.py).
#ifdef VPHO_CLOCK_STAMPS
#define VPHO_STAMP_SLOTS 65536
#define VPHO_STAMP_WORDS 10
#define VPHO_STAMP_DECL(name)                                                                                                      \
    __device__ unsigned long long name##_stamps[VPHO_STAMP_WORDS * VPHO_STAMP_SLOTS];                                              \
    extern "C" __attribute__((visibility("default"))) int vpho_diag_stamps_##name(unsigned long long* host, int slots, int clear) { \
        if (host && hipMemcpyFromSymbol(host, HIP_SYMBOL(name##_stamps), (size_t)slots * VPHO_STAMP_WORDS * 8) != hipSuccess) return 1; \
        if (clear) { void* p = nullptr; if (hipGetSymbolAddress(&p, HIP_SYMBOL(name##_stamps)) != hipSuccess || hipMemset(p, 0, sizeof(name##_stamps)) != hipSuccess) return 2; } \
        return 0;                                                                                                                  \
    }
#define VPHO_STAMP_INIT()                                                                                                          \
    unsigned long long st_t_[6] = {0, 0, 0, 0, 0, 0};                                                                              \
    const unsigned long long st_r0_ = __builtin_amdgcn_s_memrealtime();                                                            \
    unsigned long long st_rb_ = 0, st_re_ = 0;                                                                                     \
    st_t_[0] = __builtin_amdgcn_s_memtime();                                                                                       \
    __builtin_amdgcn_sched_barrier(0)
#define VPHO_STAMP_AT(k)                                                                                                           \
    do { __builtin_amdgcn_sched_barrier(0); st_t_[k] = __builtin_amdgcn_s_memtime();                                               \
         if ((k) == 2) st_rb_ = __builtin_amdgcn_s_memrealtime();                                                                  \
         if ((k) == 3) st_re_ = __builtin_amdgcn_s_memrealtime();                                                                  \
         __builtin_amdgcn_sched_barrier(0); } while (0)
#define VPHO_STAMP_WRITE(name, slot)                                                                                               \
    do {                                                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
        const unsigned long long st_r1_ = __builtin_amdgcn_s_memrealtime();                                                        \
        if (threadIdx.x == 0) {                                                                                                    \
            unsigned long long* q_ = name##_stamps + (size_t)((unsigned)(slot) & (VPHO_STAMP_SLOTS - 1)) * VPHO_STAMP_WORDS;       \
            for (int i_ = 0; i_ < 6; ++i_) q_[i_] = st_t_[i_];                                                                     \
            q_[6] = st_r0_; q_[7] = st_r1_;                                                                                        \
            q_[8] = ((unsigned long long)(st_re_ - st_rb_) << 32) | ((unsigned long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xF) << 16) | (unsigned long long)(__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xFFFF); \
            q_[9] = st_rb_;                                                                                                        \
        }                                                                                                                          \
    } while (0)
#else
#define VPHO_STAMP_DECL(name)
#define VPHO_STAMP_INIT() do {} while (0)
#define VPHO_STAMP_AT(k) do {} while (0)
#define VPHO_STAMP_WRITE(name, slot) do {} while (0)
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
