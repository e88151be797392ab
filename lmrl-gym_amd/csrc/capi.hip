// capi.hip — library-wide C ABI pieces (error text, version, device probe).
#include <stdarg.h>

#include "../../include/lmrl_amd.h"
#include "common.h"

namespace lmrl {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace lmrl

extern "C" {

const char *lmrl_last_error(void) { return lmrl::g_err; }

int lmrl_version(void) { return 100; }

const char *lmrl_device_arch(void) {
    static char arch[256] = "";
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return "";
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) return "";
    snprintf(arch, sizeof(arch), "%s", p.gcnArchName);
    return arch;
}
}
