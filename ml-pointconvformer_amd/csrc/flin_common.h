// Device helpers shared by the point-level Linear + BatchNorm kernels (fused_linear.hip, point_chain.hip).
#pragma once
#include "pcf_common.h"

namespace pcf {

__device__ __forceinline__ int ceil_div_dev(int a, int b) { return (a + b - 1) / b; }
__device__ __forceinline__ float fl_act(int act, float u) {
    if (act == 1) return fmaxf(u, 0.f);
    if (act == 2) return u > 0.f ? u : 0.1f * u;
    if (act == 3) return 1.f / (1.f + __expf(-u));
    return u;
}
__device__ __forceinline__ float fl_dact(int act, float u) {
    if (act == 1) return u > 0.f ? 1.f : 0.f;
    if (act == 2) return u > 0.f ? 1.f : 0.1f;
    if (act == 3) { const float s = 1.f / (1.f + __expf(-u)); return s * (1.f - s); }
    return 1.f;
}

// Hand-over between workgroups of ONE launch (per-workgroup partial sums -> the workgroup that finishes last).  The L2s
// of the eight XCDs are not coherent with each other for ordinary accesses, so an agent-scope release (__threadfence)
// writes back every dirty L2 line of the XCD -- with the megabytes of output a kernel has just stored that costs more
// than the kernel (measured: 21 -> 75 us for an [80k, 256] x [256, 32] product).  Instead the few hundred values that
// cross workgroups are written and read with agent-scope read-modify-write atomics (exchange / add 0), which are performed
// at the point all XCDs share, and the writer WAITS for the old value to come back before it takes its ticket:
//   * a workgroup-scope release fence compiles to no vector-memory wait at all on this target (the waves of a workgroup share
//     one L1), so "atomic store; fence(workgroup); ticket" let the ticket overtake the store now and then -- with 2048 short
//     workgroups per launch (pe_chain_kernel) one launch in ten combined a stale partial list (batch statistics off by
//     1e-5, sometimes garbage; found as run-to-run differences of a training iteration);
//   * an atomic store is acknowledged by the writer's L2 and an atomic load may be served by the reader's.
__device__ __forceinline__ void st_agent(float* p, float v) {
    const float old = __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" ::"v"(old));             // the returned value is waited for: the exchange has been performed
}
__device__ __forceinline__ float ld_agent(const float* p) {
    return __hip_atomic_fetch_add(const_cast<float*>(p), 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(double* p, double v) {
    const double old = __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" ::"v"(old));
}
__device__ __forceinline__ double ld_agent(const double* p) {
    return __hip_atomic_fetch_add(const_cast<double*>(p), 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// ticket resets by the last workgroup: also through the shared point (a plain store would sit in this XCD's L2)
__device__ __forceinline__ void st_agent(int* p, int v) {
    const int old = __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" ::"v"(old));
}
__device__ __forceinline__ void publish() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void observe() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }

// BatchNorm-backward constants of one channel from S1 = sum g, S2 = sum g * z (z the raw pre-BatchNorm value)
__device__ __forceinline__ void bn_bwd_constants(float* cst, int C, int col, double S1, double S2, double R, float* dgamma, float* dbeta,
                                                 float* dbias) {
    if (dbias) dbias[col] = 0.f;          // bias in front of a batch-statistics BatchNorm: identically zero gradient
    const double sc = (double)cst[0 * C + col], mean = (double)cst[2 * C + col], rs = (double)cst[3 * C + col];
    const double x0 = -mean * rs;
    const double dg = rs * S2 + x0 * S1;             // sum g * xhat
    if (dbeta) dbeta[col] = (float)S1;
    if (dgamma) dgamma[col] = (float)dg;
    const double m1 = S1 / R, m2 = dg / R;
    cst[4 * C + col] = (float)(-sc * m2 * rs);
    cst[5 * C + col] = (float)(-sc * (m1 + m2 * x0));
}

// Forward constants of one channel from S1 = sum z, S2 = sum z^2 over R rows: record rows 0..3 and the running statistics
__device__ __forceinline__ void bn_fwd_constants(float* cst, int C, int col, double S1, double S2, double R, const float* gamma,
                                                 const float* beta, float* running_mean, float* running_var, float eps, float momentum) {
    const double mean = S1 / R;
    double var = S2 / R - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rs = 1.0 / sqrt(var + (double)eps);
    const double sc = rs * (double)gamma[col];
    cst[0 * C + col] = (float)sc;
    cst[1 * C + col] = (float)((double)beta[col] - mean * sc);
    cst[2 * C + col] = (float)mean;
    cst[3 * C + col] = (float)rs;
    if (running_mean) {
        const double unbiased = R > 1.0 ? var * R / (R - 1.0) : var;
        running_mean[col] = (float)((1.0 - momentum) * running_mean[col] + momentum * mean);
        running_var[col] = (float)((1.0 - momentum) * running_var[col] + momentum * unbiased);
    }
}

// The per-workgroup column sums of a launch -> the layer's record, as a launch of its own: part [nparts][2][N] summed in
// index order (double).  MODE 0: forward statistics -> cst rows 0..3, running statistics.  MODE 1: BatchNorm-backward sums
// -> dgamma, dbeta, zero bias gradient, cst rows 4, 5.  On replayed graphs this costs ~3 us against ~10 us for the in-kernel
// hand-over (tickets, agent-scope exchanges that have to be waited for): the default (pcf_hip_set_flin_finish).
struct FinishArgs {
    const float* part; int nparts, N; long long R;
    float* cst;
    const float* gamma; const float* beta; float* running_mean; float* running_var; float eps, momentum;
    float* dgamma; float* dbeta; float* dbias;
};
template <int MODE>
__global__ __launch_bounds__(BLOCK) void flin_finish_kernel(const FinishArgs a) {
    __shared__ double red[2][BLOCK];
    const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5, c = blockIdx.x * 32 + cl;
    double s1 = 0.0, s2 = 0.0;
    if (c < a.N) {
        double t1 = 0.0, t2 = 0.0;
        int q = sl;
        for (; q + 8 < a.nparts; q += 16) {
            s1 += (double)a.part[((size_t)q * 2 + 0) * a.N + c]; s2 += (double)a.part[((size_t)q * 2 + 1) * a.N + c];
            t1 += (double)a.part[((size_t)(q + 8) * 2 + 0) * a.N + c]; t2 += (double)a.part[((size_t)(q + 8) * 2 + 1) * a.N + c];
        }
        if (q < a.nparts) { s1 += (double)a.part[((size_t)q * 2 + 0) * a.N + c]; s2 += (double)a.part[((size_t)q * 2 + 1) * a.N + c]; }
        s1 += t1; s2 += t2;
    }
    red[0][threadIdx.x] = s1; red[1][threadIdx.x] = s2;
    __syncthreads();
    if (threadIdx.x < 32 && c < a.N) {
        double a1 = 0.0, a2 = 0.0;
#pragma unroll
        for (int i = 0; i < BLOCK / 32; ++i) { a1 += red[0][i * 32 + cl]; a2 += red[1][i * 32 + cl]; }
        if (MODE == 0) bn_fwd_constants(a.cst, a.N, c, a1, a2, (double)a.R, a.gamma, a.beta, a.running_mean, a.running_var, a.eps, a.momentum);
        else bn_bwd_constants(a.cst, a.N, c, a1, a2, (double)a.R, a.dgamma, a.dbeta, a.dbias);
    }
}
template <int MODE>
static inline int launch_finish(const FinishArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(flin_finish_kernel<MODE>, dim3((a.N + 31) / 32), dim3(BLOCK), 0, s, a);
    return check_launch("flin_finish_kernel");
}

}  // namespace pcf
