// Per-neighbourhood attention arithmetic of the reference's ablation layers, forward and backward:
//   softmax_aggregate   PointTransformerLayer (layers.py:519-527): softmax over the K neighbours of the attention logits
//                       [R, K, J] and the weighted neighbour sum out[r, c] = sum_k v[r, k, c] * softmax_k(logit)[r, k, c % J]
//                       (share_planes groups of J channels share a weight);
//   qk_score            MultiHeadGuidanceQK (layers.py:100-114): sigmoid(scale * <q[r, k, h, :], key[r, h, :]>) per head;
//   layer_norm          nn.LayerNorm over the channel axis of the guidance query / key (layers.py:33-36, 52-53).
// All three are HBM-bound element-wise / short-reduction kernels: one lane per output element or per (row, group), the
// K (<= 64) neighbours walked in registers.  No BASELINE config uses these layers (transformer_type 'PCF',
// attention_type 'subtraction', layer_norm_guidance False); they are here so that every layer type of layers.py runs
// on HIP end to end.
#include <algorithm>

#include "pcf_common.h"

namespace pcf {

int slab_sum(const float* slabs, float* C, long long count, int splits, hipStream_t s);      // gemm.hip

static inline int grid_for(long long n, int cap = 256 * 32) {
    return (int)std::max<long long>(1, std::min<long long>((n + BLOCK - 1) / BLOCK, cap));
}

// ---- softmax over K + weighted sum ------------------------------------------------------------------------------
// thread (r, j): softmax of logit[r, :, j]; out[r, si * J + j] = sum_k v[r, k, si * J + j] * sm[r, k, j] for every share si
__global__ __launch_bounds__(BLOCK) void softmax_agg_fwd_kernel(const float* __restrict__ v, const float* __restrict__ logit,
                                                                float* __restrict__ out, float* __restrict__ sm, long long R,
                                                                int K, int C, int J) {
    const int S = C / J;
    for (long long t = (long long)blockIdx.x * BLOCK + threadIdx.x; t < R * J; t += (long long)gridDim.x * BLOCK) {
        const long long r = t / J;
        const int j = (int)(t - r * J);
        const float* lg = logit + (size_t)r * K * J + j;
        float mx = -INFINITY;
        for (int k = 0; k < K; ++k) mx = fmaxf(mx, lg[(size_t)k * J]);
        float den = 0.f;
        for (int k = 0; k < K; ++k) den += __expf(lg[(size_t)k * J] - mx);
        const float inv = 1.f / den;
        float* smr = sm + (size_t)r * K * J + j;
        for (int k = 0; k < K; ++k) smr[(size_t)k * J] = __expf(lg[(size_t)k * J] - mx) * inv;
        for (int si = 0; si < S; ++si) {
            const int c = si * J + j;
            const float* vr = v + (size_t)r * K * C + c;
            float acc = 0.f;
            for (int k = 0; k < K; ++k) acc += vr[(size_t)k * C] * smr[(size_t)k * J];
            out[(size_t)r * C + c] = acc;
        }
    }
}
// dv[r,k,c] = dout[r,c] * sm[r,k,c%J];  dsm[r,k,j] = sum_si dout[r,c] v[r,k,c];  dlogit = sm * (dsm - sum_k sm dsm)
__global__ __launch_bounds__(BLOCK) void softmax_agg_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ v,
                                                                const float* __restrict__ sm, float* __restrict__ dv,
                                                                float* __restrict__ dlogit, long long R, int K, int C, int J) {
    const int S = C / J;
    for (long long t = (long long)blockIdx.x * BLOCK + threadIdx.x; t < R * J; t += (long long)gridDim.x * BLOCK) {
        const long long r = t / J;
        const int j = (int)(t - r * J);
        const float* smr = sm + (size_t)r * K * J + j;
        float dot = 0.f;
        for (int k = 0; k < K; ++k) {
            float dsm = 0.f;
            for (int si = 0; si < S; ++si) {
                const int c = si * J + j;
                dsm += dout[(size_t)r * C + c] * v[((size_t)r * K + k) * C + c];
            }
            dot += smr[(size_t)k * J] * dsm;
        }
        for (int k = 0; k < K; ++k) {
            const float w = smr[(size_t)k * J];
            float dsm = 0.f;
            for (int si = 0; si < S; ++si) {
                const int c = si * J + j;
                const float g = dout[(size_t)r * C + c];
                dsm += g * v[((size_t)r * K + k) * C + c];
                dv[((size_t)r * K + k) * C + c] = g * w;
            }
            dlogit[((size_t)r * K + k) * J + j] = w * (dsm - dot);
        }
    }
}

// ---- per-head dot product + sigmoid -------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void qk_score_fwd_kernel(const float* __restrict__ q, const float* __restrict__ key,
                                                             float* __restrict__ score, long long R, int K, int H, int D, float scale) {
    for (long long t = (long long)blockIdx.x * BLOCK + threadIdx.x; t < R * K * H; t += (long long)gridDim.x * BLOCK) {
        const int h = (int)(t % H);
        const long long rk = t / H, r = rk / K;
        const float* qp = q + (size_t)t * D;
        const float* kp = key + ((size_t)r * H + h) * D;
        float acc = 0.f;
        for (int d = 0; d < D; ++d) acc += qp[d] * kp[d];
        score[t] = 1.f / (1.f + __expf(-acc * scale));
    }
}
// thread (r, h): g = dscore * s (1 - s) * scale;  dq[r,k,h,:] = g key[r,h,:];  dkey[r,h,:] = sum_k g q[r,k,h,:]   (D <= 64)
__global__ __launch_bounds__(BLOCK) void qk_score_bwd_kernel(const float* __restrict__ dscore, const float* __restrict__ score,
                                                             const float* __restrict__ q, const float* __restrict__ key,
                                                             float* __restrict__ dq, float* __restrict__ dkey, long long R, int K,
                                                             int H, int D, float scale) {
    for (long long t = (long long)blockIdx.x * BLOCK + threadIdx.x; t < R * H * D; t += (long long)gridDim.x * BLOCK) {
        const int d = (int)(t % D);
        const long long rh = t / D;
        const int h = (int)(rh % H);
        const long long r = rh / H;
        const float kv = key[t];
        float acc = 0.f;
        for (int k = 0; k < K; ++k) {
            const size_t e = ((size_t)r * K + k) * H + h;
            const float s = score[e];
            const float g = dscore[e] * s * (1.f - s) * scale;
            acc += g * q[e * D + d];
            dq[e * D + d] = g * kv;
        }
        dkey[t] = acc;
    }
}

// ---- LayerNorm over the last axis: one wave per row, lanes stride the channels --------------------------------------
__global__ __launch_bounds__(BLOCK) void layer_norm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* __restrict__ y,
                                                               float* __restrict__ mean, float* __restrict__ rstd, long long R, int C,
                                                               float eps) {
    const int lane = lane_id();
    for (long long r = (long long)blockIdx.x * NWAVE + wave_id(); r < R; r += (long long)gridDim.x * NWAVE) {
        const float* xr = x + (size_t)r * C;
        float s = 0.f;
        for (int c = lane; c < C; c += WAVE) s += xr[c];
        for (int off = WAVE / 2; off > 0; off >>= 1) s += __shfl_xor(s, off, WAVE);
        const float mu = s / (float)C;
        float q = 0.f;
        for (int c = lane; c < C; c += WAVE) { const float d = xr[c] - mu; q += d * d; }
        for (int off = WAVE / 2; off > 0; off >>= 1) q += __shfl_xor(q, off, WAVE);
        const float rs = rsqrtf(q / (float)C + eps);
        for (int c = lane; c < C; c += WAVE) y[(size_t)r * C + c] = (xr[c] - mu) * rs * gamma[c] + beta[c];
        if (lane == 0) { mean[r] = mu; rstd[r] = rs; }
    }
}
// dx = rstd * (g - mean_c(g) - xhat * mean_c(g xhat)), g = dy * gamma; per-workgroup partials of dgamma = sum dy xhat, dbeta = sum dy
__global__ __launch_bounds__(BLOCK) void layer_norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                               const float* __restrict__ gamma, const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, float* __restrict__ dx,
                                                               float* __restrict__ part, long long R, int C) {
    extern __shared__ float sh[];                 // [NWAVE][2][C]
    const int lane = lane_id(), wave = wave_id();
    float* mine = sh + (size_t)wave * 2 * C;
    for (int c = lane; c < 2 * C; c += WAVE) mine[c] = 0.f;
    for (long long r = (long long)blockIdx.x * NWAVE + wave; r < R; r += (long long)gridDim.x * NWAVE) {
        const float mu = mean[r], rs = rstd[r];
        float s1 = 0.f, s2 = 0.f;
        for (int c = lane; c < C; c += WAVE) {
            const float xh = (x[(size_t)r * C + c] - mu) * rs, d = dy[(size_t)r * C + c], g = d * gamma[c];
            s1 += g; s2 += g * xh;
            mine[c] += d * xh;
            mine[C + c] += d;
        }
        for (int off = WAVE / 2; off > 0; off >>= 1) { s1 += __shfl_xor(s1, off, WAVE); s2 += __shfl_xor(s2, off, WAVE); }
        s1 /= (float)C; s2 /= (float)C;
        for (int c = lane; c < C; c += WAVE) {
            const float xh = (x[(size_t)r * C + c] - mu) * rs, g = dy[(size_t)r * C + c] * gamma[c];
            dx[(size_t)r * C + c] = rs * (g - s1 - xh * s2);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * C; c += BLOCK) {
        float t = 0.f;
        for (int w = 0; w < NWAVE; ++w) t += sh[(size_t)w * 2 * C + c];
        part[(size_t)blockIdx.x * 2 * C + c] = t;
    }
}

}  // namespace pcf

extern "C" {

int pcf_hip_softmax_aggregate_forward(const float* v, const float* logit, float* out, float* sm, long long R, int K, int C,
                                      int J, void* stream) {
    using namespace pcf;
    PCF_REQUIRE(R >= 0 && K >= 1 && C >= 1 && J >= 1 && C % J == 0, "softmax_aggregate: bad sizes (K=%d C=%d J=%d)", K, C, J);
    if (R == 0) return ok();
    PCF_REQUIRE(v && logit && out && sm, "softmax_aggregate: null pointer");
    hipLaunchKernelGGL(softmax_agg_fwd_kernel, dim3(grid_for(R * J)), dim3(BLOCK), 0, (hipStream_t)stream, v, logit, out, sm, R, K, C, J);
    return check_launch("softmax_agg_fwd_kernel");
}

int pcf_hip_softmax_aggregate_backward(const float* dout, const float* v, const float* sm, float* dv, float* dlogit,
                                       long long R, int K, int C, int J, void* stream) {
    using namespace pcf;
    PCF_REQUIRE(R >= 0 && K >= 1 && C >= 1 && J >= 1 && C % J == 0, "softmax_aggregate_backward: bad sizes");
    if (R == 0) return ok();
    PCF_REQUIRE(dout && v && sm && dv && dlogit, "softmax_aggregate_backward: null pointer");
    hipLaunchKernelGGL(softmax_agg_bwd_kernel, dim3(grid_for(R * J)), dim3(BLOCK), 0, (hipStream_t)stream, dout, v, sm, dv, dlogit, R, K, C, J);
    return check_launch("softmax_agg_bwd_kernel");
}

int pcf_hip_qk_score_forward(const float* q, const float* key, float* score, long long R, int K, int H, int D, float scale,
                             void* stream) {
    using namespace pcf;
    PCF_REQUIRE(R >= 0 && K >= 1 && H >= 1 && D >= 1, "qk_score: bad sizes");
    if (R == 0) return ok();
    PCF_REQUIRE(q && key && score, "qk_score: null pointer");
    hipLaunchKernelGGL(qk_score_fwd_kernel, dim3(grid_for(R * K * H)), dim3(BLOCK), 0, (hipStream_t)stream, q, key, score, R, K, H, D, scale);
    return check_launch("qk_score_fwd_kernel");
}

int pcf_hip_qk_score_backward(const float* dscore, const float* score, const float* q, const float* key, float* dq, float* dkey,
                              long long R, int K, int H, int D, float scale, void* stream) {
    using namespace pcf;
    PCF_REQUIRE(R >= 0 && K >= 1 && H >= 1 && D >= 1, "qk_score_backward: bad sizes");
    if (R == 0) return ok();
    PCF_REQUIRE(dscore && score && q && key && dq && dkey, "qk_score_backward: null pointer");
    hipLaunchKernelGGL(qk_score_bwd_kernel, dim3(grid_for(R * H * D)), dim3(BLOCK), 0, (hipStream_t)stream, dscore, score, q, key, dq,
                       dkey, R, K, H, D, scale);
    return check_launch("qk_score_bwd_kernel");
}

int pcf_hip_layer_norm_forward(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                               long long R, int C, float eps, void* stream) {
    using namespace pcf;
    PCF_REQUIRE(R >= 0 && C >= 1, "layer_norm: bad sizes");
    if (R == 0) return ok();
    PCF_REQUIRE(x && gamma && beta && y && mean && rstd, "layer_norm: null pointer");
    hipLaunchKernelGGL(layer_norm_fwd_kernel, dim3(grid_for(R * WAVE)), dim3(BLOCK), 0, (hipStream_t)stream, x, gamma, beta, y, mean,
                       rstd, R, C, eps);
    return check_launch("layer_norm_fwd_kernel");
}

size_t pcf_hip_layer_norm_backward_workspace_bytes(long long R, int C) {
    if (R < 0 || C < 1) return 0;
    return (size_t)1025 * 2 * C * 4 + 256;
}

int pcf_hip_layer_norm_backward(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                float* dx, float* dgamma, float* dbeta, long long R, int C, void* workspace,
                                size_t workspace_bytes, void* stream) {
    using namespace pcf;
    PCF_REQUIRE(R >= 0 && C >= 1 && C <= 2048, "layer_norm_backward: bad sizes");
    PCF_REQUIRE(dgamma && dbeta, "layer_norm_backward: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (R == 0) {
        (void)zero_async(dgamma, (size_t)C * 4, s);
        (void)zero_async(dbeta, (size_t)C * 4, s);
        return ok();
    }
    PCF_REQUIRE(dy && x && gamma && mean && rstd && dx && workspace && aligned16(workspace) &&
                workspace_bytes >= pcf_hip_layer_norm_backward_workspace_bytes(R, C), "layer_norm_backward: null pointer or small workspace");
    const int nb = (int)std::max<long long>(1, std::min<long long>((R + NWAVE - 1) / NWAVE, 1024));
    float* part = static_cast<float*>(workspace);
    hipLaunchKernelGGL(layer_norm_bwd_kernel, dim3(nb), dim3(BLOCK), (size_t)NWAVE * 2 * C * 4, s, dy, x, gamma, mean, rstd, dx, part, R, C);
    if (int e = check_launch("layer_norm_bwd_kernel")) return e;
    // part is [nb][2C]: dgamma = columns 0..C-1, dbeta = columns C..2C-1 of the slab sum (into a scratch row, then split)
    float* tot = part + (size_t)nb * 2 * C;
    if (int e = slab_sum(part, tot, 2 * C, nb, s)) return e;
    if (copy_async(dgamma, tot, (size_t)C * 4, s) != hipSuccess || copy_async(dbeta, tot + C, (size_t)C * 4, s) != hipSuccess)
        return fail(PCF_E_LAUNCH, "layer_norm_backward: copy");
    return ok();
}

}  // extern "C"
