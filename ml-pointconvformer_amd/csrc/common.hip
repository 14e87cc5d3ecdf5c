// Error text, version string and launch check for libpcf_hip.so.
#include "pcf_common.h"

#include <atomic>
#include <mutex>
#include <string>
#include <stdlib.h>
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

__global__ __launch_bounds__(BLOCK) void zero_kernel(unsigned char* p, size_t n) {
    zero_item(p, n, (size_t)blockIdx.x * BLOCK + threadIdx.x, (size_t)gridDim.x * BLOCK);
}

// 16-byte pieces where both pointers allow it, bytes otherwise (the few copies of the library are a handful of floats)
__global__ __launch_bounds__(BLOCK) void copy_kernel(unsigned char* dst, const unsigned char* src, size_t n) {
    const size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x, T = (size_t)gridDim.x * BLOCK;
    if (((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) == 0) {
        const size_t quads = n / 16;
        for (size_t i = t; i < quads; i += T) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
        for (size_t i = quads * 16 + t; i < n; i += T) dst[i] = src[i];
    } else {
        for (size_t i = t; i < n; i += T) dst[i] = src[i];
    }
}

static int fill_grid(size_t bytes) {
    const size_t blocks = (bytes / 16 + BLOCK - 1) / BLOCK;
    return (int)(blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks));
}

// Diagnostic switch for the A/B of DESIGN.md "graph replay fault": PCF_ZERO_WITH_MEMSET=1 (read once) clears with
// hipMemsetAsync again, i.e. puts the memset nodes back into captured graphs.  Never set by the package.
static bool zero_with_memset() {
    static const bool v = [] { const char* e = getenv("PCF_ZERO_WITH_MEMSET"); return e && e[0] == '1'; }();
    return v;
}

hipError_t zero_async(void* p, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return hipSuccess;
    if (!p) return hipErrorInvalidValue;
    if (zero_with_memset()) return hipMemsetAsync(p, 0, bytes, stream);
    hipLaunchKernelGGL(zero_kernel, dim3(fill_grid(bytes)), dim3(BLOCK), 0, stream, static_cast<unsigned char*>(p), bytes);
    return hipGetLastError();
}

hipError_t copy_async(void* dst, const void* src, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return hipSuccess;
    if (!dst || !src) return hipErrorInvalidValue;
    hipLaunchKernelGGL(copy_kernel, dim3(fill_grid(bytes)), dim3(BLOCK), 0, stream, static_cast<unsigned char*>(dst),
                       static_cast<const unsigned char*>(src), bytes);
    return hipGetLastError();
}

}  // namespace pcf

extern "C" {
// Host emulation of zero_kernel's indexing over `blocks` x BLOCK threads (test hook: no GPU needed).
void pcf_hip_zero_host(void* p, size_t bytes, int blocks) {
    const size_t T = (size_t)(blocks < 1 ? 1 : blocks) * pcf::BLOCK;
    for (size_t t = 0; t < T; ++t) pcf::zero_item(static_cast<unsigned char*>(p), bytes, t, T);
}

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
