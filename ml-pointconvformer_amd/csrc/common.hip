// Error text, version string and launch check for libpcf_hip.so.
#include "pcf_common.h"

#include <atomic>
#include <mutex>
#include <string>
#include <string.h>

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

// Launch log (test / diagnostic hook): while enabled, every launch site that reports through check_launch() appends its
// label -- the kernel's name where a dispatcher chooses between variants -- so a caller can see which kernels an
// operator call selected.  Off by default; bounded.
static std::atomic<int> g_log_on{0};
static std::mutex g_log_mu;
static std::string g_log;

void note_launch(const char* what) {
    if (!g_log_on.load(std::memory_order_relaxed)) return;
    std::lock_guard<std::mutex> lock(g_log_mu);
    if (g_log.size() < (1u << 20)) { g_log += what; g_log += '\n'; }
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(PCF_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    note_launch(what);
    return ok();
}

}  // namespace pcf

extern "C" {
const char* pcf_hip_version(void) { return "pcf_hip 0.1 gfx950"; }
const char* pcf_hip_last_error(void) { return pcf::err_buf(); }

void pcf_hip_launch_log_enable(int on) {
    std::lock_guard<std::mutex> lock(pcf::g_log_mu);
    pcf::g_log.clear();
    pcf::g_log_on.store(on ? 1 : 0, std::memory_order_relaxed);
}

size_t pcf_hip_launch_log_read(char* buf, size_t capacity) {
    std::lock_guard<std::mutex> lock(pcf::g_log_mu);
    const size_t need = pcf::g_log.size() + 1;
    if (buf && capacity >= need) {
        memcpy(buf, pcf::g_log.c_str(), need);
        pcf::g_log.clear();
    }
    return need;
}
}
