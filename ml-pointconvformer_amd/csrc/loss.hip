// Cross entropy of the segmentation head (train_ScanNet_DDP_WarmUP.py:243, criterion(pred, target) at :404):
// nn.CrossEntropyLoss(ignore_index, label_smoothing), reduction "mean" over the rows whose target is not ignored.
//   loss_i = (1 - eps) * (-log p_i[t_i]) + eps * (-(1 / C) sum_c log p_i[c]),      L = sum_i loss_i / #valid
//   dL/dlogit_i[c] = (p_i[c] - (1 - eps) [c == t_i] - eps / C) / #valid            (zero for ignored rows)
// torch runs log_softmax + a one-workgroup nll reduction (166 us forward, 109 us backward on 144k rows x 20 classes); here
// a lane owns a row (C <= 64 classes in registers), the forward also leaves the unnormalised gradient, a one-workgroup
// kernel sums the per-workgroup partials in index order (deterministic) and the backward is one scaling pass.
#include "pcf_common.h"

namespace pcf {

constexpr int CE_MAXC = 64;

__global__ __launch_bounds__(BLOCK) void ce_forward_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target, long long R,
                                                         int C, long long ignore_index, float smoothing, float* __restrict__ dlogits,
                                                         float* __restrict__ part) {
    __shared__ float red[2][NWAVE];
    float loss = 0.f, cnt = 0.f;
    for (long long r = (long long)blockIdx.x * BLOCK + threadIdx.x; r < R; r += (long long)gridDim.x * BLOCK) {
        const float* x = logits + (size_t)r * C;
        float* d = dlogits + (size_t)r * C;
        const long long t = target[r];
        if (t == ignore_index || t < 0 || t >= C) {
            for (int c = 0; c < C; ++c) d[c] = 0.f;
            continue;
        }
        float m = -INFINITY;
        for (int c = 0; c < C; ++c) m = fmaxf(m, x[c]);
        float se = 0.f, sx = 0.f;
        for (int c = 0; c < C; ++c) { se += __expf(x[c] - m); sx += x[c]; }
        const float lse = m + __logf(se);
        // -log p[t] = lse - x[t];   -(1/C) sum_c log p[c] = lse - sx / C
        loss += (1.f - smoothing) * (lse - x[t]) + smoothing * (lse - sx / (float)C);
        cnt += 1.f;
        const float inv = 1.f / se, base = smoothing / (float)C;
        for (int c = 0; c < C; ++c) d[c] = __expf(x[c] - m) * inv - base - (c == t ? 1.f - smoothing : 0.f);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { loss += __shfl_xor(loss, o, WAVE); cnt += __shfl_xor(cnt, o, WAVE); }
    if (lane_id() == 0) { red[0][wave_id()] = loss; red[1][wave_id()] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        part[2 * blockIdx.x + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

// out[0] = mean loss, out[1] = number of valid rows
__global__ __launch_bounds__(BLOCK) void ce_finish_kernel(const float* __restrict__ part, int n, float* __restrict__ out) {
    __shared__ double sl[BLOCK], sc[BLOCK];
    double l = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < n; i += BLOCK) { l += (double)part[2 * i]; c += (double)part[2 * i + 1]; }
    sl[threadIdx.x] = l; sc[threadIdx.x] = c;
    __syncthreads();
    for (int o = BLOCK / 2; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) { sl[threadIdx.x] += sl[threadIdx.x + o]; sc[threadIdx.x] += sc[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = sc[0] > 0.0 ? (float)(sl[0] / sc[0]) : NAN; out[1] = (float)sc[0]; }
}

__global__ __launch_bounds__(BLOCK) void ce_backward_kernel(const float* __restrict__ dlogits, const float* __restrict__ grad_out,
                                                          const float* __restrict__ stat, long long n, float* __restrict__ dx) {
    const float s = grad_out[0] / stat[1];
    for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * BLOCK) dx[i] = dlogits[i] * s;
}

static inline int ce_grid(long long R) { return (int)std::max<long long>(1, std::min<long long>((R + BLOCK - 1) / BLOCK, 1024)); }

}  // namespace pcf

extern "C" {

size_t pcf_hip_cross_entropy_workspace_bytes(long long R) { return (size_t)pcf::ce_grid(R > 0 ? R : 1) * 2 * 4 + 64; }

// logits [R, C] (C <= 64), target [R] int64 -> stat[0] = mean loss over the rows with target != ignore_index (NaN when there
// is none, as torch), stat[1] = their number; dlogits [R, C] = the gradient of the loss SUM w.r.t. the logits
int pcf_hip_cross_entropy_forward(const float* logits, const int64_t* target, long long R, int C, long long ignore_index,
                                  float label_smoothing, float* stat, float* dlogits, void* workspace, size_t workspace_bytes,
                                  void* stream) {
    using namespace pcf;
    PCF_REQUIRE(R >= 0 && C >= 1 && C <= CE_MAXC, "cross_entropy: 1 <= classes <= %d (got %d)", CE_MAXC, C);
    PCF_REQUIRE(label_smoothing >= 0.f && label_smoothing <= 1.f, "cross_entropy: label_smoothing outside [0, 1]");
    PCF_REQUIRE(stat && workspace && workspace_bytes >= pcf_hip_cross_entropy_workspace_bytes(R) && (R == 0 || (logits && target && dlogits)),
                "cross_entropy: null pointer or small workspace");
    hipStream_t s = (hipStream_t)stream;
    float* part = static_cast<float*>(workspace);
    const int grid = R > 0 ? ce_grid(R) : 0;
    if (grid) hipLaunchKernelGGL(ce_forward_kernel, dim3(grid), dim3(BLOCK), 0, s, logits, target, R, C, ignore_index, label_smoothing, dlogits, part);
    hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(BLOCK), 0, s, part, grid, stat);
    return check_launch("cross_entropy forward");
}

// dx = dlogits * grad_out / stat[1]
int pcf_hip_cross_entropy_backward(const float* dlogits, const float* grad_out, const float* stat, long long R, int C, float* dx,
                                   void* stream) {
    using namespace pcf;
    PCF_REQUIRE(R >= 0 && C >= 1, "cross_entropy_backward: bad sizes");
    if (R == 0) return ok();
    PCF_REQUIRE(dlogits && grad_out && stat && dx, "cross_entropy_backward: null pointer");
    const long long n = R * C;
    const int grid = (int)std::max<long long>(1, std::min<long long>((n + BLOCK - 1) / BLOCK, 4096));
    hipLaunchKernelGGL(ce_backward_kernel, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, dlogits, grad_out, stat, n, dx);
    return check_launch("cross_entropy backward");
}

}  // extern "C"
