// Aggregate + linear entry points (forward, backward with atomics, backward over the inverse CSR).
// Composition of aggregate.hip and gemm.hip; see include/pcf_hip.h for the contracts.
#include <algorithm>

#include "pcf_common.h"

namespace pcf {
int aggregate_forward(const float*, const int64_t*, const float*, const float*, const float*, float*, int, int, int, int,
                      int, int, int, int, hipStream_t);
int aggregate_backward(const float*, const float*, const int64_t*, const float*, const float*, const float*, float*,
                       float*, float*, float*, float*, int, int, int, int, int, int, int, int, hipStream_t);
int csr_reduce(const float*, const int32_t*, const uint8_t*, const int32_t*, float*, int, int, int, int, int, int, int,
               hipStream_t);
int csr_gather1(const float*, const float*, const int32_t*, const uint8_t*, const int32_t*, float*, int, int, int, int,
                int, int, int, int, hipStream_t);
bool agg1_covers(bool, int, int, int, int);
int gemm_f32(const float*, bool, int, const float*, bool, int, const float*, float*, int, int, int, int, int, hipStream_t);
int choose_splits(int, int, int);
int slab_sum(const float*, float*, long long, int, hipStream_t);
int colsum_blocks(int);
int colsum(const float*, float*, float*, int, int, hipStream_t);

struct BwdWorkspace {
    size_t off_dpcv, off_slabs, off_part, off_contrib, bytes;
    int splits;
};

static BwdWorkspace plan_ws(int B, int N, int Nout, int K, int Ci, int Ca, int Cm, int Co, bool with_contrib) {
    BwdWorkspace w{};
    const size_t total = (size_t)B * Nout;
    const size_t J = (size_t)(Ci + Ca) * Cm;
    w.splits = choose_splits(Co, (int)J, (int)std::min<size_t>(total, 0x7fffffff));
    size_t off = 0;
    w.off_dpcv = off; off = align_up(off + total * J * 4, 256);
    w.off_slabs = off; off = align_up(off + (w.splits > 1 ? (size_t)w.splits * Co * J * 4 : 0), 256);
    w.off_part = off; off = align_up(off + (size_t)colsum_blocks((int)total) * Co * 4, 256);
    if (agg1_covers(false, K, Ci, Ca, Cm)) with_contrib = false;      // grad_x is gathered from d(pconv_out)
    w.off_contrib = off; off = align_up(off + (with_contrib ? total * K * Ci * 4 : 0), 256);
    w.bytes = off;
    return w;
}

static int linear_backward_common(const float* gout, const float* lin_w, const float* pconv_out, float* grad_lin_w,
                                  float* grad_lin_b, char* ws, const BwdWorkspace& w, int total, int J, int Co,
                                  hipStream_t s) {
    float* dpcv = reinterpret_cast<float*>(ws + w.off_dpcv);
    // d(pconv_out) = grad_out . lin_w           [total, J]
    if (int e = gemm_f32(gout, true, Co, lin_w, false, J, nullptr, dpcv, J, total, J, Co, 1, s)) return e;
    // d(lin_w) = grad_out^T . pconv_out          [Co, J], reduction over the points split into slabs
    if (w.splits > 1) {
        float* slabs = reinterpret_cast<float*>(ws + w.off_slabs);
        if (int e = gemm_f32(gout, false, Co, pconv_out, false, J, nullptr, slabs, J, Co, J, total, w.splits, s)) return e;
        if (int e = slab_sum(slabs, grad_lin_w, (long long)Co * J, w.splits, s)) return e;
    } else {
        if (int e = gemm_f32(gout, false, Co, pconv_out, false, J, nullptr, grad_lin_w, J, Co, J, total, 1, s)) return e;
    }
    // d(lin_b) = column sums of grad_out
    return colsum(gout, reinterpret_cast<float*>(ws + w.off_part), grad_lin_b, total, Co, s);
}

}  // namespace pcf

extern "C" {

int pcf_hip_pconv_linear_forward(const float* x, const int64_t* idx, const float* w, const float* add,
                                 const float* lin_w, const float* lin_b, float* out, float* pconv_out, int B, int N,
                                 int Nout, int K, int Ci, int Ca, int Cm, int Co, void* stream) {
    using namespace pcf;
    PCF_REQUIRE(Co >= 1, "pconv_linear_forward: C_out must be >= 1 (got %d)", Co);
    hipStream_t s = (hipStream_t)stream;
    if (int e = aggregate_forward(x, idx, nullptr, w, add, pconv_out, B, N, Nout, K, Ci, Ca, Cm, 1, s)) return e;
    const int total = B * Nout, J = (Ci + Ca) * Cm;
    if (total == 0) return ok();
    PCF_REQUIRE(lin_w && out, "pconv_linear_forward: null pointer (lin_w=%p out=%p)", (const void*)lin_w, (void*)out);
    return gemm_f32(pconv_out, true, J, lin_w, true, J, lin_b, out, Co, total, Co, J, 1, s);
}

size_t pcf_hip_pconv_linear_backward_workspace_bytes(int B, int N, int Nout, int K, int Ci, int Ca, int Cm, int Co) {
    (void)N;
    return pcf::plan_ws(B, N, Nout, K, Ci, Ca, Cm, Co, false).bytes;
}

int pcf_hip_pconv_linear_backward(const float* grad_out, const float* x, const int64_t* idx, const float* w,
                                  const float* add, const float* lin_w, const float* pconv_out, float* grad_x,
                                  float* grad_w, float* grad_add, float* grad_lin_w, float* grad_lin_b, void* workspace,
                                  size_t workspace_bytes, int B, int N, int Nout, int K, int Ci, int Ca, int Cm, int Co,
                                  void* stream) {
    using namespace pcf;
    PCF_REQUIRE(Co >= 1 && Ci + Ca >= 1 && Cm >= 1 && B >= 0 && Nout >= 0, "pconv_linear_backward: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    const int total = B * Nout, J = (Ci + Ca) * Cm;
    const BwdWorkspace ws = plan_ws(B, N, Nout, K, Ci, Ca, Cm, Co, false);
    PCF_REQUIRE(workspace_bytes >= ws.bytes && (workspace || ws.bytes == 0) && aligned16(workspace),
                "pconv_linear_backward: workspace too small or misaligned (%zu < %zu)", workspace_bytes, ws.bytes);
    PCF_REQUIRE(grad_lin_w && grad_lin_b, "pconv_linear_backward: null grad_lin_w / grad_lin_b");
    if (total == 0) {
        (void)zero_async(grad_lin_w, (size_t)Co * J * 4, s);
        (void)zero_async(grad_lin_b, (size_t)Co * 4, s);
        return aggregate_backward(nullptr, x, idx, nullptr, w, add, grad_x, nullptr, nullptr, grad_w, grad_add, B, N,
                                  Nout, K, Ci, Ca, Cm, 1, s);
    }
    PCF_REQUIRE(grad_out && lin_w && pconv_out, "pconv_linear_backward: null pointer");
    char* wsp = static_cast<char*>(workspace);
    if (int e = linear_backward_common(grad_out, lin_w, pconv_out, grad_lin_w, grad_lin_b, wsp, ws, total, J, Co, s))
        return e;
    return aggregate_backward(reinterpret_cast<const float*>(wsp + ws.off_dpcv), x, idx, nullptr, w, add, grad_x, nullptr,
                              nullptr, grad_w, grad_add, B, N, Nout, K, Ci, Ca, Cm, 1, s);
}

size_t pcf_hip_pconv_linear_opt_backward_workspace_bytes(int B, int N, int Nout, int K, int Ci, int Ca, int Cm,
                                                         int Co) {
    return pcf::plan_ws(B, N, Nout, K, Ci, Ca, Cm, Co, true).bytes;
}

int pcf_hip_pconv_linear_opt_backward(const float* grad_out, const float* x, const int32_t* inv_neighbors,
                                      const uint8_t* inv_k, const int32_t* inv_idx, const int64_t* idx, const float* w,
                                      const float* add, const float* lin_w, const float* pconv_out, float* grad_x,
                                      float* grad_w, float* grad_add, float* grad_lin_w, float* grad_lin_b,
                                      void* workspace, size_t workspace_bytes, int B, int N, int Nout, int K, int Ci,
                                      int Ca, int Cm, int Co, int inv_len, int inv_idx_len, void* stream) {
    using namespace pcf;
    PCF_REQUIRE(Co >= 1 && Ci + Ca >= 1 && Cm >= 1 && B >= 0 && Nout >= 0, "pconv_linear_opt_backward: bad sizes");
    // same check the reference makes (pconv_ops.cu:885), plus the +1 the CSR end pointer needs
    PCF_REQUIRE(inv_idx_len >= N + 1, "inverse_neighbor_idx size must be >= N + 1 (got %d, N=%d)", inv_idx_len, N);
    PCF_REQUIRE(inv_len >= 0 && (inv_len == 0 || (inv_neighbors && inv_k)) && inv_idx,
                "pconv_linear_opt_backward: null inverse index");
    hipStream_t s = (hipStream_t)stream;
    const int total = B * Nout, J = (Ci + Ca) * Cm;
    const BwdWorkspace ws = plan_ws(B, N, Nout, K, Ci, Ca, Cm, Co, true);
    PCF_REQUIRE(workspace_bytes >= ws.bytes && (workspace || ws.bytes == 0) && aligned16(workspace),
                "pconv_linear_opt_backward: workspace too small or misaligned (%zu < %zu)", workspace_bytes, ws.bytes);
    PCF_REQUIRE(grad_lin_w && grad_lin_b && (grad_x || (long long)B * N * Ci == 0),
                "pconv_linear_opt_backward: null gradient pointer");
    char* wsp = static_cast<char*>(workspace);
    if (total == 0) {
        (void)zero_async(grad_lin_w, (size_t)Co * J * 4, s);
        (void)zero_async(grad_lin_b, (size_t)Co * 4, s);
        if ((size_t)B * N * Ci) (void)zero_async(grad_x, (size_t)B * N * Ci * 4, s);
        return ok();
    }
    PCF_REQUIRE(grad_out && lin_w && pconv_out, "pconv_linear_opt_backward: null pointer");
    if (int e = linear_backward_common(grad_out, lin_w, pconv_out, grad_lin_w, grad_lin_b, wsp, ws, total, J, Co, s))
        return e;
    if (Ci > 0 && agg1_covers(false, K, Ci, Ca, Cm)) {
        // C_mid = 1 (the decoder): grad_x gathers rows of d(pconv_out) through the CSR, no per-edge rows in between
        const float* dpcv = reinterpret_cast<const float*>(wsp + ws.off_dpcv);
        if (int e = aggregate_backward(dpcv, x, idx, nullptr, w, add, nullptr, nullptr, nullptr, grad_w, grad_add, B, N,
                                       Nout, K, Ci, Ca, Cm, 1, s))
            return e;
        return csr_gather1(dpcv, w, inv_neighbors, inv_k, inv_idx, grad_x, B, N, Nout, K, Ci, J, inv_len, inv_idx_len, s);
    }
    float* contrib = reinterpret_cast<float*>(wsp + ws.off_contrib);
    if (Ci == 0) contrib = nullptr;
    if (int e = aggregate_backward(reinterpret_cast<const float*>(wsp + ws.off_dpcv), x, idx, nullptr, w, add, nullptr,
                                   contrib, nullptr, grad_w, grad_add, B, N, Nout, K, Ci, Ca, Cm, 1, s))
        return e;
    return csr_reduce(contrib, inv_neighbors, inv_k, inv_idx, grad_x, B, N, Nout, K, Ci, inv_len, inv_idx_len, s);
}

}  // extern "C"
