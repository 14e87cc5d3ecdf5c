// Error text, version string and launch check for libpcf_hip.so.
#include "pcf_common.h"

namespace pcf {

char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(PCF_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return ok();
}

}  // namespace pcf

extern "C" {
const char* pcf_hip_version(void) { return "pcf_hip 0.1 gfx950"; }
const char* pcf_hip_last_error(void) { return pcf::err_buf(); }
}
