#include "common.h"
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
}  // namespace vpho

extern "C" const char* vpho_last_error(void) { return vpho::err_slot(); }
extern "C" int vpho_abi_version(void) { return 1; }
