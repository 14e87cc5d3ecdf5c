// Gradient clipping + AdamW over all parameter tensors of a model in a handful of launches.
//
// The training loop does clip_grad_norm_(parameters, 10) and AdamW.step() on ~200 small tensors per iteration
// (train_ScanNet_DDP_WarmUP.py:237-241, :421); through torch's multi-tensor kernels that is ~45 launches and 0.7 ms of GPU
// time for 1.9 M parameters (30 MB of traffic: ~8 us at HBM speed).  Here the tensor lists travel as KERNEL ARGUMENTS
// (MT_MAX tensors per launch: pointers + counts fit the 4 KB argument segment), so a captured HIP graph holds them by value
// and no device-side pointer table has to be kept in step with the allocator:
//   mt_sqnorm_kernel   per 4096-element chunk: sum of squares of the gradient -> one partial per workgroup
//   mt_finish_kernel   one workgroup: partials (index order, double) -> total 2-norm, clip coefficient
//                      min(max_norm / (norm + 1e-6), 1), step += 1, the two bias corrections
//   mt_adamw_kernel    per chunk: g *= coef (written back: the gradients ARE clipped, as clip_grad_norm_ leaves them),
//                      decoupled weight decay, moment updates, parameter update -- torch's fused AdamW arithmetic
// Learning rate, step counter and coefficients live in a small device record, so a replayed graph sees their current values.
#include "pcf_common.h"

namespace pcf {

constexpr int MT_MAX = 72;               // tensors per launch
constexpr int MT_CHUNK = 4096;           // elements per workgroup
struct MtArgs {
    float* p[MT_MAX]; float* g[MT_MAX]; float* m[MT_MAX]; float* v[MT_MAX];
    int count[MT_MAX];
    int block0[MT_MAX + 1];              // first workgroup of each tensor
    int n;
    float* rec;                          // [8]: lr, step, norm, coef, 1 - beta1^step, sqrt(1 - beta2^step), -, -
    float* partials; int partial0;       // sqnorm: this launch's workgroups write partials[partial0 + blockIdx.x]
    float beta1, beta2, omb1, omb2, eps, weight_decay;       // omb = 1 - beta, rounded from double as torch's scalars are
};

__device__ __forceinline__ int mt_tensor(const MtArgs& a, int block) {
    int lo = 0, hi = a.n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (a.block0[mid] <= block) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__global__ __launch_bounds__(BLOCK) void mt_sqnorm_kernel(const MtArgs a) {
    __shared__ float red[NWAVE];
    const int t = mt_tensor(a, blockIdx.x);
    const int base = (blockIdx.x - a.block0[t]) * MT_CHUNK, n = a.count[t];
    const float* g = a.g[t];
    float s = 0.f;
    for (int i = base + threadIdx.x; i < min(base + MT_CHUNK, n); i += BLOCK) { const float x = g[i]; s = fmaf(x, x, s); }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, WAVE);
    if (lane_id() == 0) red[wave_id()] = s;
    __syncthreads();
    if (threadIdx.x == 0) a.partials[a.partial0 + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(BLOCK) void mt_finish_kernel(const float* partials, int n, float* rec, float max_norm, double beta1, double beta2) {
    __shared__ double red[BLOCK];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += BLOCK) s += (double)partials[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = BLOCK / 2; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(red[0]);
        const float step = rec[1] + 1.f;
        rec[1] = step;
        rec[2] = norm;
        rec[3] = max_norm > 0.f ? fminf(max_norm / (norm + 1e-6f), 1.f) : 1.f;
        rec[4] = (float)(1.0 - pow(beta1, (double)step));
        rec[5] = (float)sqrt(1.0 - pow(beta2, (double)step));
    }
}

__global__ __launch_bounds__(BLOCK) void mt_adamw_kernel(const MtArgs a) {
    const int t = mt_tensor(a, blockIdx.x);
    const int base = (blockIdx.x - a.block0[t]) * MT_CHUNK, n = a.count[t];
    float* p = a.p[t]; float* g = a.g[t]; float* m = a.m[t]; float* v = a.v[t];
    const float lr = a.rec[0], coef = a.rec[3], bc1 = a.rec[4], bc2s = a.rec[5];
    const float step_size = lr / bc1;
    for (int i = base + threadIdx.x; i < min(base + MT_CHUNK, n); i += BLOCK) {
        const float gi = g[i] * coef;
        float pi = p[i];
        pi -= lr * a.weight_decay * pi;
        float mi = m[i], vi = v[i];
        mi = mi + (gi - mi) * a.omb1;
        vi = a.beta2 * vi + a.omb2 * gi * gi;
        const float denom = sqrtf(vi) / bc2s + a.eps;
        pi -= step_size * mi / denom;
        p[i] = pi; m[i] = mi; v[i] = vi;
        if (coef != 1.f) g[i] = gi;
    }
}

}  // namespace pcf

extern "C" {

int pcf_hip_adamw_max_tensors(void) { return pcf::MT_MAX; }
int pcf_hip_adamw_chunk(void) { return pcf::MT_CHUNK; }

// One list of at most pcf_hip_adamw_max_tensors() tensors.  phase 0: partial sums of squares of the gradients into
// partials[partial0 ...] (one per 4096-element chunk, tensors in list order); phase 1: the AdamW update with the record's
// coefficients.  rec: device float[8] (see optimizer.hip); counts in elements.
int pcf_hip_adamw_list(int phase, int n, float* const* params, float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                       const long long* counts, float* rec, float* partials, int partial0, double beta1, double beta2, double eps,
                       double weight_decay, void* stream) {
    using namespace pcf;
    PCF_REQUIRE(n >= 0 && n <= MT_MAX, "adamw_list: at most %d tensors per call", MT_MAX);
    PCF_REQUIRE(phase == 0 || phase == 1, "adamw_list: phase is 0 or 1");
    if (n == 0) return ok();
    PCF_REQUIRE(params && grads && exp_avg && exp_avg_sq && counts && rec && (phase == 1 || partials), "adamw_list: null pointer");
    MtArgs a{};
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        PCF_REQUIRE(counts[i] >= 1 && counts[i] < (1ll << 31), "adamw_list: bad element count");
        PCF_REQUIRE(grads[i] && (phase == 0 || (params[i] && exp_avg[i] && exp_avg_sq[i])), "adamw_list: null tensor");
        a.p[i] = params[i]; a.g[i] = grads[i]; a.m[i] = exp_avg[i]; a.v[i] = exp_avg_sq[i]; a.count[i] = (int)counts[i];
        a.block0[i] = blocks;
        blocks += ceil_div(counts[i], MT_CHUNK);
    }
    a.block0[n] = blocks;
    a.n = n; a.rec = rec; a.partials = partials; a.partial0 = partial0;
    a.beta1 = (float)beta1; a.beta2 = (float)beta2; a.omb1 = (float)(1.0 - beta1); a.omb2 = (float)(1.0 - beta2);
    a.eps = (float)eps; a.weight_decay = (float)weight_decay;
    hipStream_t s = (hipStream_t)stream;
    if (phase == 0) hipLaunchKernelGGL(mt_sqnorm_kernel, dim3(blocks), dim3(BLOCK), 0, s, a);
    else hipLaunchKernelGGL(mt_adamw_kernel, dim3(blocks), dim3(BLOCK), 0, s, a);
    return check_launch(phase == 0 ? "mt_sqnorm_kernel" : "mt_adamw_kernel");
}

// total gradient norm from the partials, clip coefficient (max_norm <= 0: no clipping), step += 1, bias corrections
int pcf_hip_adamw_finish(const float* partials, int n_partials, float* rec, float max_norm, double beta1, double beta2, void* stream) {
    using namespace pcf;
    PCF_REQUIRE(partials && rec && n_partials >= 0, "adamw_finish: bad arguments");
    hipLaunchKernelGGL(mt_finish_kernel, dim3(1), dim3(BLOCK), 0, (hipStream_t)stream, partials, n_partials, rec, max_norm, beta1, beta2);
    return check_launch("mt_finish_kernel");
}

}  // extern "C"
