#include "common.h"
#include <map>
#include <mutex>
#include <utility>
#include <vector>
#include "../../include/vpho_hip.h"

namespace vpho {
static thread_local char g_err[512] = "";
char* err_slot() { return g_err; }
int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}

struct ProfState { bool on = false; std::vector<std::pair<hipEvent_t, hipEvent_t>> ev; double flops = 0, bytes = 0; };
static ProfState g_prof[PROF_NCLASS];
static std::mutex g_prof_mu;
bool prof_on(int cls) { return g_prof[cls].on; }
void prof_record(int cls, hipEvent_t a, hipEvent_t b, double flops, double bytes) {
    std::lock_guard<std::mutex> l(g_prof_mu);
    g_prof[cls].ev.emplace_back(a, b);
    g_prof[cls].flops += flops;
    g_prof[cls].bytes += bytes;
}

static std::map<std::pair<const void*, int>, int> g_dyn_lds;
static std::mutex g_dyn_lds_mu;
int dynamic_lds(const void* kernel, int bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail("hipGetDevice: %s", hipGetErrorString(e));
    std::lock_guard<std::mutex> l(g_dyn_lds_mu);
    int& have = g_dyn_lds[{kernel, dev}];
    if (have >= bytes) return 0;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return fail("hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d): %s", bytes, hipGetErrorString(e));
    have = bytes;
    return 0;
}
}  // namespace vpho

extern "C" int vpho_prof_enable(int cls, int on) {
    VPHO_REQUIRE(cls >= 0 && cls < vpho::PROF_NCLASS, "vpho_prof_enable: class %d", cls);
    vpho::g_prof[cls].on = on != 0;
    return 0;
}

extern "C" int vpho_prof_collect(int cls, double* total_ms, long long* launches, double* total_flops, double* total_bytes) {
    VPHO_REQUIRE(cls >= 0 && cls < vpho::PROF_NCLASS && total_ms && launches && total_flops && total_bytes, "vpho_prof_collect: bad argument");
    std::lock_guard<std::mutex> l(vpho::g_prof_mu);
    auto& st = vpho::g_prof[cls];
    double ms = 0;
    for (auto& e : st.ev) {
        VPHO_HIP(hipEventSynchronize(e.second));
        float t = 0;
        VPHO_HIP(hipEventElapsedTime(&t, e.first, e.second));
        ms += t;
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    *total_ms = ms; *launches = (long long)st.ev.size(); *total_flops = st.flops; *total_bytes = st.bytes;
    st.ev.clear(); st.flops = 0; st.bytes = 0;
    return 0;
}

extern "C" const char* vpho_last_error(void) { return vpho::err_slot(); }
extern "C" int vpho_abi_version(void) { return 12; }
