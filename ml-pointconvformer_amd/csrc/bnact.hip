// BatchNorm (+ activation) over the rows of a wide [R, C] matrix, forward and backward, and the three
// products of a dense linear's backward.  Used for the point-level Linear_BN layers whose channel counts
// exceed the 64 of the per-edge engine (edge_mlp.hip): PCFLayer.linear (C_mid*C_in/4 -> C/2), the
// decoder's linears (layers.py:973-981), UnaryBlocks of the deeper levels (layer_utils.py:281-315).
// z = x.W^T + b comes from the fp32 MFMA contraction in gemm.hip; here:
//   stats    : per-column sum / sum of squares -> mean, rstd (+ running statistics)       2 kernels
//   forward  : y = act((z - mean) * rstd * gamma + beta), 16-byte accesses, in place allowed
//   backward : column sums of g and g*xhat (g = dy * act'(u)), then dz = rstd*gamma*(g - m1 - xhat*m2)
// All HBM-bound single passes; partial sums per workgroup are combined in a fixed order (deterministic).
#include <algorithm>

#include "pcf_common.h"

namespace pcf {

int gemm_f32(const float*, bool, int, const float*, bool, int, const float*, float*, int, int, int, int, int, hipStream_t);
int choose_splits(int, int, int);
int slab_sum(const float*, float*, long long, int, hipStream_t);
int colsum_blocks(int);
int colsum(const float*, float*, float*, int, int, hipStream_t);

__device__ __forceinline__ float bn_act_fwd(int act, float u) {
    if (act == 1) return fmaxf(u, 0.f);
    if (act == 2) return u > 0.f ? u : 0.1f * u;
    if (act == 3) return 1.f / (1.f + __expf(-u));
    return u;
}
__device__ __forceinline__ float bn_act_bwd(int act, float u) {
    if (act == 1) return u > 0.f ? 1.f : 0.f;
    if (act == 2) return u > 0.f ? 1.f : 0.1f;
    if (act == 3) { const float s = 1.f / (1.f + __expf(-u)); return s * (1.f - s); }
    return 1.f;
}

// Column partial sums of one or two per-element quantities.  Threads: TX column lanes x TY row lanes; blockIdx.x
// walks row ranges and blockIdx.y the TX-wide column chunks, so a short, wide matrix (1.4k x 512 at the coarse
// levels) still spreads over the chip: one workgroup looping over its chunks is a serial chain of load latencies.
//   MODE 0: (z, z*z)      MODE 1: (g, g*xhat) with g = dy * act'(u)
template <int MODE>
__global__ __launch_bounds__(BLOCK) void col_partials_kernel(const float* __restrict__ z, const float* __restrict__ dy,
                                                             long long R, int C, int TX, long long rows_per_block,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             int act, float* __restrict__ part,
                                                             const float* __restrict__ res = nullptr) {
    __shared__ float s1[BLOCK], s2[BLOCK];
    const int TY = BLOCK / TX;
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = min(R, r0 + rows_per_block);
    {
        const int c = blockIdx.y * TX + tx;
        float a = 0.f, b = 0.f;
        if (c < C) {
            float mu = 0.f, rs = 1.f, ga = 1.f, be = 0.f;
            if (MODE == 1 && mean) { mu = mean[c]; rs = rstd[c]; ga = gamma[c]; be = beta[c]; }
            auto term = [&](long long r, float& sa, float& sb) {
                const float v = z[(size_t)r * C + c];
                if (MODE == 0) { sa += v; sb += v * v; }
                else {
                    const float xh = (v - mu) * rs;
                    const float g = dy[(size_t)r * C + c] * bn_act_bwd(act, (mean ? xh * ga + be : v) + (res ? res[(size_t)r * C + c] : 0.f));
                    sa += g; sb += g * xh;
                }
            };
            float a1 = 0.f, b1 = 0.f, a2 = 0.f, b2 = 0.f, a3 = 0.f, b3 = 0.f;
            long long r = r0 + ty;
            for (; r + 3 * TY < r1; r += 4 * TY) {      // four loads in flight per lane
                term(r, a, b); term(r + TY, a1, b1); term(r + 2 * TY, a2, b2); term(r + 3 * TY, a3, b3);
            }
            for (; r < r1; r += TY) term(r, a, b);
            a = (a + a1) + (a2 + a3); b = (b + b1) + (b2 + b3);
        }
        s1[threadIdx.x] = a; s2[threadIdx.x] = b;
        __syncthreads();
        if (ty == 0 && c < C) {
            for (int y = 1; y < TY; ++y) { a += s1[y * TX + tx]; b += s2[y * TX + tx]; }
            part[((size_t)blockIdx.x * 2 + 0) * C + c] = a;
            part[((size_t)blockIdx.x * 2 + 1) * C + c] = b;
        }
    }
}

// combine partials: one 1024-thread workgroup per 16 columns x 64 slices of the partial list, two
// independent fp64 chains per slice and sum (a serial walk of ~300 partials is ~20 us of load latency)
constexpr int FIN_THREADS = 1024;
constexpr int FIN_COLS = 16;
constexpr int FIN_SLICES = FIN_THREADS / FIN_COLS;
template <int MODE>
__global__ __launch_bounds__(FIN_THREADS) void col_finalize_kernel(const float* __restrict__ part, int nblocks, long long R, int C,
                                                                   float eps, float momentum, float* __restrict__ running_mean,
                                                                   float* __restrict__ running_var, float* __restrict__ o1,
                                                                   float* __restrict__ o2, float* __restrict__ m1,
                                                                   float* __restrict__ m2, float* __restrict__ zero_out = nullptr) {
    __shared__ double sa[FIN_THREADS], sb[FIN_THREADS];
    const int c = blockIdx.x * FIN_COLS + (threadIdx.x & (FIN_COLS - 1));
    const int slice = threadIdx.x / FIN_COLS;
    double a = 0.0, b = 0.0;
    if (c < C) {
        double a1 = 0.0, b1 = 0.0;
        int p = slice;
        for (; p + FIN_SLICES < nblocks; p += 2 * FIN_SLICES) {
            a += (double)part[((size_t)p * 2 + 0) * C + c];
            b += (double)part[((size_t)p * 2 + 1) * C + c];
            a1 += (double)part[((size_t)(p + FIN_SLICES) * 2 + 0) * C + c];
            b1 += (double)part[((size_t)(p + FIN_SLICES) * 2 + 1) * C + c];
        }
        if (p < nblocks) {
            a += (double)part[((size_t)p * 2 + 0) * C + c];
            b += (double)part[((size_t)p * 2 + 1) * C + c];
        }
        a += a1; b += b1;
    }
    sa[threadIdx.x] = a; sb[threadIdx.x] = b;
    __syncthreads();
    if (threadIdx.x < FIN_COLS && c < C) {
        a = 0.0; b = 0.0;
#pragma unroll 8
        for (int sl = 0; sl < FIN_SLICES; ++sl) { a += sa[sl * FIN_COLS + threadIdx.x]; b += sb[sl * FIN_COLS + threadIdx.x]; }
        const double n = (double)R;
        if (MODE == 0) {
            const double mean = a / n;
            double var = b / n - mean * mean;
            if (var < 0.0) var = 0.0;
            o1[c] = (float)mean;
            o2[c] = (float)(1.0 / sqrt(var + (double)eps));
            if (running_mean) {
                const double unbiased = R > 1 ? var * n / (n - 1.0) : var;
                running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
                running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
            }
        } else {
            if (zero_out) zero_out[c] = 0.f;     // gradient of the bias in front of this batch-statistics BatchNorm
            o1[c] = (float)a;          // dbeta
            o2[c] = (float)b;          // dgamma
            m1[c] = (float)(a / n);
            m2[c] = (float)(b / n);
        }
    }
}

// Per-column constants of the four consecutive columns a lane works on.  When the grid stride in elements
// is a multiple of C (the host arranges that whenever C divides 4*BLOCK) they are loaded once per lane.
struct Col4 { float mu[4], rs[4], ga[4], be[4]; };
__device__ __forceinline__ void load_col4(Col4& k, int c, const float* mean, const float* rstd, const float* gamma,
                                          const float* beta) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { k.mu[j] = mean[c + j]; k.rs[j] = rstd[c + j]; k.ga[j] = gamma[c + j]; k.be[j] = beta[c + j]; }
}

__global__ __launch_bounds__(BLOCK) void bnact_fwd_kernel(const float* __restrict__ z, float* __restrict__ y, long long total,
                                                          int C, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          int act, int vec, int fixed_cols, const float* __restrict__ res) {
    if (vec) {
        const long long units = total >> 2;
        const long long u0 = (long long)blockIdx.x * BLOCK + threadIdx.x;
        Col4 k;
        if (mean && fixed_cols) load_col4(k, (int)((u0 * 4) % C), mean, rstd, gamma, beta);
        for (long long u = u0; u < units; u += (long long)gridDim.x * BLOCK) {
            if (mean && !fixed_cols) load_col4(k, (int)((u * 4) % C), mean, rstd, gamma, beta);
            float4 v = ld4(z + u * 4);
            float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (res) rv = ld4(res + u * 4);
            float* e = &v.x;
            const float* re = &rv.x;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t = e[j];
                if (mean) t = (t - k.mu[j]) * k.rs[j] * k.ga[j] + k.be[j];
                e[j] = bn_act_fwd(act, t + re[j]);
            }
            st4(y + u * 4, v);
        }
    } else {
        for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < total; i += (long long)gridDim.x * BLOCK) {
            const int c = (int)(i % C);
            float t = z[i];
            if (mean) t = (t - mean[c]) * rstd[c] * gamma[c] + beta[c];
            y[i] = bn_act_fwd(act, t + (res ? res[i] : 0.f));
        }
    }
}

__global__ __launch_bounds__(BLOCK) void bnact_bwd_kernel(const float* __restrict__ z, const float* __restrict__ dy,
                                                          float* __restrict__ dz, long long total, int C,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ m1, const float* __restrict__ m2, int act,
                                                          const float* __restrict__ res, float* __restrict__ dres) {
    // one element per lane and round: a float4 form of this kernel measured ~2x slower on gfx950 (31 vs 13 us at [80k, 64])
    for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < total; i += (long long)gridDim.x * BLOCK) {
        const int c = (int)(i % C);
        const float v = z[i];
        const float rsd = res ? res[i] : 0.f;
        if (mean) {
            const float xh = (v - mean[c]) * rstd[c];
            const float g = dy[i] * bn_act_bwd(act, xh * gamma[c] + beta[c] + rsd);
            if (dres) dres[i] = g;
            dz[i] = rstd[c] * gamma[c] * (m1 ? (g - m1[c] - xh * m2[c]) : g);
        } else {
            const float g = dy[i] * bn_act_bwd(act, v + rsd);
            if (dres) dres[i] = g;
            dz[i] = g;
        }
    }
}

static int tx_for(int C) { int t = 1; while (t < C && t < 64) t <<= 1; return t; }
// row ranges of the column reductions: >= 16 rows each, ~2048 workgroups over rows x column chunks, <= 1024 ranges
static int stats_blocks(long long R, int C) {
    const long long chunks = ceil_div(C, tx_for(C));
    const long long want = std::max<long long>(256, 2048 / chunks);
    return (int)std::max<long long>(1, std::min<long long>({(R + 15) / 16, want, 1024ll}));
}
// element-wise grids: every workgroup runs the same number of grid-stride rounds (a capped grid with a ragged
// last round leaves most of the chip idle for up to half of a short kernel)
static int ew_grid(long long n) {
    const long long need = std::max<long long>(1, (n + BLOCK - 1) / BLOCK), cap = 256 * 32;
    const long long rounds = (need + cap - 1) / cap;
    return (int)((need + rounds - 1) / rounds);
}

}  // namespace pcf

extern "C" {

size_t pcf_hip_bnact_workspace_bytes(long long R, int C) {
    if (R < 0 || C < 0) return 0;
    return ((size_t)pcf::stats_blocks(R, C) * 2 * C + 2 * (size_t)C) * 4 + 256;
}

int pcf_hip_bnact_stats(const float* z, long long R, int C, float eps, float momentum, float* running_mean,
                        float* running_var, float* mean_out, float* rstd_out, void* workspace, size_t workspace_bytes,
                        void* stream) {
    using namespace pcf;
    PCF_REQUIRE(R > 0 && C >= 1, "bnact_stats: need R > 0 and C >= 1 (R=%lld C=%d)", R, C);
    PCF_REQUIRE(z && mean_out && rstd_out && workspace && aligned16(workspace) &&
                    workspace_bytes >= pcf_hip_bnact_workspace_bytes(R, C), "bnact_stats: null pointer or small workspace");
    PCF_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bnact_stats: need both running stats or none");
    hipStream_t s = (hipStream_t)stream;
    float* part = static_cast<float*>(workspace);
    const int nb = stats_blocks(R, C);
    const long long rpb = (R + nb - 1) / nb;
    hipLaunchKernelGGL(col_partials_kernel<0>, dim3(nb, ceil_div(C, tx_for(C))), dim3(BLOCK), 0, s, z, nullptr, R, C, tx_for(C), rpb, nullptr, nullptr,
                       nullptr, nullptr, 0, part);
    hipLaunchKernelGGL(col_finalize_kernel<0>, dim3(ceil_div(C, FIN_COLS)), dim3(FIN_THREADS), 0, s, part, nb, R, C, eps, momentum,
                       running_mean, running_var, mean_out, rstd_out, nullptr, nullptr);
    return check_launch("bnact statistics");
}

int pcf_hip_bnact_forward_res(const float* z, const float* residual, long long R, int C, const float* mean,
                              const float* rstd, const float* gamma, const float* beta, int act, float* y, void* stream) {
    using namespace pcf;
    PCF_REQUIRE(R >= 0 && C >= 1 && act >= 0 && act <= 3, "bnact_forward: bad arguments");
    if (R == 0) return ok();
    PCF_REQUIRE(z && y && (!mean || (rstd && gamma && beta)), "bnact_forward: null pointer");
    const long long total = R * C;
    const int vec = (C % 4 == 0) && aligned16(z) && aligned16(y) && aligned16(residual);
    hipLaunchKernelGGL(bnact_fwd_kernel, dim3(ew_grid(vec ? total / 4 : total)), dim3(BLOCK), 0, (hipStream_t)stream, z, y,
                       total, C, mean, rstd, gamma, beta, act, vec, (4 * BLOCK) % C == 0, residual);
    return check_launch("bnact forward");
}

int pcf_hip_bnact_forward(const float* z, long long R, int C, const float* mean, const float* rstd, const float* gamma,
                          const float* beta, int act, float* y, void* stream) {
    return pcf_hip_bnact_forward_res(z, nullptr, R, C, mean, rstd, gamma, beta, act, y, stream);
}

int pcf_hip_bnact_backward(const float* z, const float* dy, long long R, int C, const float* mean, const float* rstd,
                           const float* gamma, const float* beta, int batch_stats, int act, float* dz, float* dgamma,
                           float* dbeta, void* workspace, size_t workspace_bytes, void* stream) {
    return pcf_hip_bnact_backward_res(z, nullptr, dy, R, C, mean, rstd, gamma, beta, batch_stats, act, dz, nullptr, dgamma,
                                      dbeta, nullptr, workspace, workspace_bytes, stream);
}

int pcf_hip_bnact_backward_res(const float* z, const float* residual, const float* dy, long long R, int C, const float* mean,
                               const float* rstd, const float* gamma, const float* beta, int batch_stats, int act,
                               float* dz, float* dresidual, float* dgamma, float* dbeta, float* dbias_zero,
                               void* workspace, size_t workspace_bytes, void* stream) {
    using namespace pcf;
    PCF_REQUIRE(R >= 0 && C >= 1 && act >= 0 && act <= 3, "bnact_backward: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const bool bn = mean != nullptr;
    if (R == 0) {
        if (bn) { (void)zero_async(dgamma, (size_t)C * 4, s); (void)zero_async(dbeta, (size_t)C * 4, s); }
        return ok();
    }
    PCF_REQUIRE(z && dy && dz && (!bn || (rstd && gamma && beta && dgamma && dbeta)), "bnact_backward: null pointer");
    PCF_REQUIRE(!bn || (workspace && aligned16(workspace) && workspace_bytes >= pcf_hip_bnact_workspace_bytes(R, C)),
                "bnact_backward: workspace too small or misaligned");
    float* m1 = nullptr;
    float* m2 = nullptr;
    if (bn) {
        float* wsf = static_cast<float*>(workspace);
        m1 = wsf; m2 = wsf + C;
        float* part = wsf + 2 * (size_t)C;
        const int nb = stats_blocks(R, C);
        const long long rpb = (R + nb - 1) / nb;
        hipLaunchKernelGGL(col_partials_kernel<1>, dim3(nb, ceil_div(C, tx_for(C))), dim3(BLOCK), 0, s, z, dy, R, C, tx_for(C), rpb, mean, rstd, gamma,
                           beta, act, part, residual);
        hipLaunchKernelGGL(col_finalize_kernel<1>, dim3(ceil_div(C, FIN_COLS)), dim3(FIN_THREADS), 0, s, part, nb, R, C, 0.f, 0.f, nullptr,
                           nullptr, dbeta, dgamma, m1, m2, batch_stats ? dbias_zero : nullptr);
        if (int e = check_launch("bnact backward reductions")) return e;
        if (!batch_stats) { m1 = nullptr; m2 = nullptr; }
    }
    hipLaunchKernelGGL(bnact_bwd_kernel, dim3(std::min(ew_grid(R * C), 4096)), dim3(BLOCK), 0, s, z, dy, dz, R * C, C, mean,
                       rstd, gamma, beta, m1, m2, act, residual, dresidual);
    return check_launch("bnact backward");
}

// Backward of z = x.W^T + b for dense [R, Cin] x: dx = dz.W, dW = dz^T.x (split over the rows), db = colsum(dz).
size_t pcf_hip_linear_backward_workspace_bytes(long long R, int Cin, int Cout) {
    if (R < 0 || Cin < 0 || Cout < 0) return 0;
    const int splits = pcf::choose_splits(Cout, Cin, (int)std::min<long long>(R, 0x7fffffff));
    return ((size_t)splits * Cout * Cin + (size_t)pcf::colsum_blocks((int)std::min<long long>(R, 0x7fffffff)) * Cout) * 4 + 512;
}

int pcf_hip_linear_backward(const float* dz, const float* x, const float* W, long long R, int Cin, int Cout, float* dx,
                            float* dW, float* db, void* workspace, size_t workspace_bytes, void* stream) {
    using namespace pcf;
    PCF_REQUIRE(R >= 0 && R < (1ll << 31) && Cin >= 1 && Cout >= 1, "linear_backward: bad sizes");
    PCF_REQUIRE(W && dW && workspace && aligned16(workspace) &&
                    workspace_bytes >= pcf_hip_linear_backward_workspace_bytes(R, Cin, Cout),
                "linear_backward: null pointer or small workspace");
    hipStream_t s = (hipStream_t)stream;
    if (R == 0) {
        (void)zero_async(dW, (size_t)Cout * Cin * 4, s);
        if (db) (void)zero_async(db, (size_t)Cout * 4, s);
        return ok();
    }
    PCF_REQUIRE(dz && x, "linear_backward: null pointer");
    const int Ri = (int)R;
    if (dx)
        if (int e = gemm_f32(dz, true, Cout, W, false, Cin, nullptr, dx, Cin, Ri, Cin, Cout, 1, s)) return e;
    const int splits = choose_splits(Cout, Cin, Ri);
    float* slabs = static_cast<float*>(workspace);
    if (splits > 1) {
        if (int e = gemm_f32(dz, false, Cout, x, false, Cin, nullptr, slabs, Cin, Cout, Cin, Ri, splits, s)) return e;
        if (int e = slab_sum(slabs, dW, (long long)Cout * Cin, splits, s)) return e;
    } else {
        if (int e = gemm_f32(dz, false, Cout, x, false, Cin, nullptr, dW, Cin, Cout, Cin, Ri, 1, s)) return e;
    }
    if (db) return colsum(dz, slabs + align_up((size_t)splits * Cout * Cin, 64), db, Ri, Cout, s);
    return ok();
}

}  // extern "C"
