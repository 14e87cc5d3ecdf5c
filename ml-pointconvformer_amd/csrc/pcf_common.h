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
