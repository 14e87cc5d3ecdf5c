// Shared host/device helpers for libpcf_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/pcf_hip.h"

namespace pcf {

constexpr int WAVE = 64;
constexpr int BLOCK = 256;          // 4 waves per workgroup, one per SIMD
constexpr int NWAVE = BLOCK / WAVE;
constexpr int LDS_BUDGET = 64 * 1024;   // target per-workgroup LDS so >= 2 workgroups fit a CU
constexpr int LDS_MAX = 160 * 1024;

// ---- per-thread error text -------------------------------------------------------------------
char* err_buf();
int fail(int code, const char* fmt, ...);
inline int ok() { err_buf()[0] = 0; return PCF_OK; }
int check_launch(const char* what);
void note_launch(const char* what);

#define PCF_REQUIRE(cond, ...)                                        \
    do {                                                              \
        if (!(cond)) return ::pcf::fail(PCF_E_BADARG, __VA_ARGS__);   \
    } while (0)

// Zero-fill / device-to-device copy as ORDINARY KERNELS on `stream` (common.hip).  The library never calls hipMemsetAsync /
// hipMemcpyAsync: under stream capture those become memset / memcpy graph nodes, and a replayed graph whose cell
// histogram was cleared by a memset node ran the dependent kernel on uncleared memory ("graph replay fault").
// A kernel node is ordered and made visible exactly like every other kernel of the chain.
hipError_t zero_async(void* p, size_t bytes, hipStream_t stream);
hipError_t copy_async(void* dst, const void* src, size_t bytes, hipStream_t stream);

// One thread's share of zeroing bytes [0, n) at p (any alignment): the unaligned head and tail bytes go to the first 15
// threads, the 16-byte-aligned body to everyone in 16-byte stores.  Shared by the kernel and its host emulation
// (pcf_hip_zero_host, tests/test_abi_cpu.py).
__host__ __device__ inline void zero_item(unsigned char* p, size_t n, size_t t, size_t T) {
    size_t head = (16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15;
    if (head > n) head = n;
    const size_t quads = (n - head) / 16;
    const size_t tail = n - head - quads * 16;
    if (t < head) p[t] = 0;
    uint4* body = reinterpret_cast<uint4*>(p + head);
    for (size_t i = t; i < quads; i += T) body[i] = make_uint4(0u, 0u, 0u, 0u);
    if (t < tail) p[head + quads * 16 + t] = 0;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- device helpers --------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

__device__ __forceinline__ float4 fma4(float a, float4 b, float4 c) {
    c.x = fmaf(a, b.x, c.x);
    c.y = fmaf(a, b.y, c.y);
    c.z = fmaf(a, b.z, c.z);
    c.w = fmaf(a, b.w, c.w);
    return c;
}
__device__ __forceinline__ float dot4(float4 a, float4 b, float acc) {
    acc = fmaf(a.x, b.x, acc);
    acc = fmaf(a.y, b.y, acc);
    acc = fmaf(a.z, b.z, acc);
    acc = fmaf(a.w, b.w, acc);
    return acc;
}

// c % H without an integer division when H is a power of two (it always is in the reference's configs)
struct HeadMod {
    int H, mask;
    __host__ __device__ explicit HeadMod(int h) : H(h), mask(((h & (h - 1)) == 0) ? h - 1 : -1) {}
    __device__ __forceinline__ int operator()(int c) const { return mask >= 0 ? (c & mask) : (c % H); }
};

}  // namespace pcf
