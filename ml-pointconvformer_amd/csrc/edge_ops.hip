// Per-edge helpers around the aggregate (gfx950): row gather / scatter-add, gather+max over the
// neighbourhood, and the fused "edge geometry" kernel that turns coordinates + normals + the
// neighbour table straight into the offsets and the 12-channel viewpoint-invariant descriptor.
//
// Replaces index_points (layer_utils.py:13-30, an advanced-indexing gather that materialises
// [B,M,K,C] and back-propagates through index_put_) and VI_coordinate_transform
// (layer_utils.py:176-231, ~15 elementwise / matmul temporaries of size M*K*3) and the
// gather + max shortcut of layers.py:403-408.  All are HBM-bound: one pass, 16-byte accesses where
// the layout allows, no temporaries.
#include <algorithm>

#include "pcf_common.h"

namespace pcf {

// ---- gather rows: out[b, s, :] = table[b, idx[b, s], :]  (zeros for out-of-range indices) --------
template <bool VEC>
__global__ __launch_bounds__(BLOCK) void gather_rows_kernel(const float* __restrict__ table,
                                                            const int64_t* __restrict__ idx, float* __restrict__ out,
                                                            int N, long long S, int C, long long units) {
    const int per_row = VEC ? (C >> 2) : C;
    for (long long u = (long long)blockIdx.x * BLOCK + threadIdx.x; u < units; u += (long long)gridDim.x * BLOCK) {
        const long long row = u / per_row;             // over B*S
        const int c = (int)(u - row * per_row);
        const long long b = row / S;
        const int64_t j = idx[row];
        const bool okj = j >= 0 && j < N;
        if (VEC) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (okj) v = ld4(table + ((size_t)(b * N + j)) * C + c * 4);
            st4(out + (size_t)u * 4, v);
        } else {
            out[u] = okj ? table[((size_t)(b * N + j)) * C + c] : 0.f;
        }
    }
}

// grad_table[b, idx[b,s], :] += grad_rows[b, s, :]   (float atomics; grad_table zeroed by the host wrapper)
__global__ __launch_bounds__(BLOCK) void scatter_add_rows_kernel(const float* __restrict__ grad_rows,
                                                                 const int64_t* __restrict__ idx,
                                                                 float* __restrict__ grad_table, int N, long long S,
                                                                 int C, long long units) {
    for (long long u = (long long)blockIdx.x * BLOCK + threadIdx.x; u < units; u += (long long)gridDim.x * BLOCK) {
        const long long row = u / C;
        const int c = (int)(u - row * C);
        const long long b = row / S;
        const int64_t j = idx[row];
        if (j >= 0 && j < N) atomicAdd(grad_table + ((size_t)(b * N + j)) * C + c, grad_rows[u]);
    }
}

// ---- gather + max over K: out[b,m,c] = max_k table[b, idx[b,m,k], c];  argk = first maximiser -----
__global__ __launch_bounds__(BLOCK) void gather_max_kernel(const float* __restrict__ table,
                                                           const int64_t* __restrict__ idx, float* __restrict__ out,
                                                           uint8_t* __restrict__ argk, int N, int M, int K, int C,
                                                           long long units) {
    for (long long u = (long long)blockIdx.x * BLOCK + threadIdx.x; u < units; u += (long long)gridDim.x * BLOCK) {
        const long long pt = u / C;                    // over B*M
        const int c = (int)(u - pt * C);
        const long long b = pt / M;
        float best = -__builtin_inff();
        int bk = 0;
        for (int k = 0; k < K; ++k) {
            const int64_t j = idx[pt * K + k];
            const float v = (j >= 0 && j < N) ? table[((size_t)(b * N + j)) * C + c] : 0.f;
            if (v > best) { best = v; bk = k; }
        }
        out[u] = best;
        argk[u] = (uint8_t)bk;
    }
}

__global__ __launch_bounds__(BLOCK) void gather_max_backward_kernel(const float* __restrict__ gout,
                                                                    const int64_t* __restrict__ idx,
                                                                    const uint8_t* __restrict__ argk,
                                                                    float* __restrict__ grad_table, int N, int M, int K,
                                                                    int C, long long units) {
    for (long long u = (long long)blockIdx.x * BLOCK + threadIdx.x; u < units; u += (long long)gridDim.x * BLOCK) {
        const long long pt = u / C;
        const int c = (int)(u - pt * C);
        const long long b = pt / M;
        const int64_t j = idx[pt * K + argk[u]];
        if (j >= 0 && j < N) atomicAdd(grad_table + ((size_t)(b * N + j)) * C + c, gout[u]);
    }
}

// ---- edge geometry -------------------------------------------------------------------------------
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 ld3(const float* p) { return {p[0], p[1], p[2]}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ V3 unit(V3 a) {      // F.normalize: a / max(|a|, 1e-12)
    const float inv = 1.f / fmaxf(sqrtf(dot(a, a)), 1e-12f);      // one division, three products (<= 1 ulp from a / n)
    return {a.x * inv, a.y * inv, a.z * inv};
}

// The 12 channels of layer_utils.py:190-231: n_j.n_i, r^.n_i, r^.n_j, r.n_i, r^.n_j, n_j.v, n_j.w,
// r.(n_j x n_i), |r|, r   with r^ = unit(r), v = unit(n_i - (r^.n_i) r^), w = unit(r^ x v).
__device__ __forceinline__ void vi_channels(V3 r, V3 nj, V3 ni, float* o) {
    const V3 rh = unit(r);
    const float proj = dot(rh, ni);
    const V3 v = unit({ni.x - proj * rh.x, ni.y - proj * rh.y, ni.z - proj * rh.z});
    const V3 w = unit(cross(rh, v));
    const float t3 = dot(rh, nj);
    o[0] = dot(nj, ni);
    o[1] = proj;
    o[2] = t3;
    o[3] = dot(r, ni);
    o[4] = t3;
    o[5] = dot(nj, v);
    o[6] = dot(nj, w);
    o[7] = dot(r, cross(nj, ni));
    o[8] = sqrtf(dot(r, r));
    o[9] = r.x;
    o[10] = r.y;
    o[11] = r.z;
}

__device__ __forceinline__ void store12(float* dst, const float* o, bool vec) {
    if (vec) {
        st4(dst, make_float4(o[0], o[1], o[2], o[3]));
        st4(dst + 4, make_float4(o[4], o[5], o[6], o[7]));
        st4(dst + 8, make_float4(o[8], o[9], o[10], o[11]));
    } else {
#pragma unroll
        for (int i = 0; i < 12; ++i) dst[i] = o[i];
    }
}

// The 64 x 12 descriptors of a wave (lane = edge e0 + lane) go through LDS so that each of the three stores writes
// one contiguous KiB: with every lane storing its own 48-byte row as three float4 (16-byte pieces at a 48-byte
// stride) the kernel took 57 us at 1.28 M edges, with this 32 us.  `st`: WAVE * 13 floats of this wave.
__device__ __forceinline__ void store12_wave(float* vi, long long e0, long long edges, const float* o, bool vec, float* st,
                                             int lane) {
    if (vec && e0 + WAVE <= edges) {
#pragma unroll
        for (int i = 0; i < 12; ++i) st[lane * 13 + i] = o[i];
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int q = it * WAVE + lane;              // 16-byte chunk of the wave's 3 KiB
            const int ed = q / 3, off = (q - ed * 3) * 4;
            const float* src = st + ed * 13 + off;
            st4(vi + (size_t)e0 * 12 + (size_t)q * 4, make_float4(src[0], src[1], src[2], src[3]));
        }
    } else if (e0 + lane < edges) {
        store12(vi + (size_t)(e0 + lane) * 12, o, vec);
    }
}

// one lane per edge; rel and/or vi may be null
__global__ __launch_bounds__(BLOCK) void edge_geometry_kernel(const float* __restrict__ ref_xyz,
                                                              const float* __restrict__ ref_norm,
                                                              const int64_t* __restrict__ idx,
                                                              const float* __restrict__ ctr_xyz,
                                                              const float* __restrict__ ctr_norm, float* __restrict__ rel,
                                                              float* __restrict__ vi, int N, int M, int K, long long edges,
                                                              bool vec) {
    __shared__ float stage[NWAVE][WAVE * 13];
    const int lane = lane_id();
    for (long long e0 = ((long long)blockIdx.x * BLOCK + threadIdx.x) - lane; e0 < edges; e0 += (long long)gridDim.x * BLOCK) {
        const long long e = e0 + lane;
        float o[12];
        if (e < edges) {
            const long long pt = e / K;
            const long long b = pt / M;
            const int64_t j = idx[e];
            const bool okj = j >= 0 && j < N;
            const size_t src = (size_t)(b * N + (okj ? j : 0)) * 3;
            const V3 c = ld3(ctr_xyz + (size_t)pt * 3);
            V3 p = okj ? ld3(ref_xyz + src) : c;
            const V3 r = {p.x - c.x, p.y - c.y, p.z - c.z};
            if (rel) { rel[e * 3] = r.x; rel[e * 3 + 1] = r.y; rel[e * 3 + 2] = r.z; }
            if (vi) {
                const V3 nj = okj ? ld3(ref_norm + src) : V3{0.f, 0.f, 0.f};
                vi_channels(r, nj, ld3(ctr_norm + (size_t)pt * 3), o);
            }
        }
        if (vi) store12_wave(vi, e0, edges, o, vec, stage[wave_id()], lane);
    }
}

__global__ __launch_bounds__(BLOCK) void vi_from_gathered_kernel(const float* __restrict__ rel,
                                                                 const float* __restrict__ nbr_norm,
                                                                 const float* __restrict__ ctr_norm,
                                                                 float* __restrict__ vi, int K, long long edges, bool vec) {
    __shared__ float stage[NWAVE][WAVE * 13];
    const int lane = lane_id();
    for (long long e0 = ((long long)blockIdx.x * BLOCK + threadIdx.x) - lane; e0 < edges; e0 += (long long)gridDim.x * BLOCK) {
        const long long e = e0 + lane;
        float o[12];
        if (e < edges) vi_channels(ld3(rel + e * 3), ld3(nbr_norm + e * 3), ld3(ctr_norm + (e / K) * 3), o);
        store12_wave(vi, e0, edges, o, vec, stage[wave_id()], lane);
    }
}


// ---- guidance difference: s[pt,k,:] = q[pt,k,:] - key[pt,:],  q = [ gx[idx[pt,k]] | pe[pt,k] ] ------
// key = q[pt,0,:] (self neighbourhoods: neighbour 0 is the point itself) or max_k q[pt,k,:] (strided).
// Replaces the cat / slice-or-max / repeat / subtract of layers.py:372-381 and layers.py:52-53, which
// materialise three [B,M,K,2G] tensors.  One lane per (point, channel).
__global__ __launch_bounds__(BLOCK) void guidance_diff_fwd_kernel(const float* __restrict__ gx,
                                                                  const int64_t* __restrict__ idx,
                                                                  const float* __restrict__ pe, float* __restrict__ s,
                                                                  uint8_t* __restrict__ argk, int N, int M, int K, int G,
                                                                  int P, int use_max, long long units) {
    const int C = G + P;
    for (long long u = (long long)blockIdx.x * BLOCK + threadIdx.x; u < units; u += (long long)gridDim.x * BLOCK) {
        const long long pt = u / C;
        const int c = (int)(u - pt * C);
        const long long b = pt / M;
        auto q = [&](int k) -> float {
            if (c < G) {
                const int64_t j = idx[pt * K + k];
                return (j >= 0 && j < N) ? gx[((size_t)(b * N + j)) * G + c] : 0.f;
            }
            return pe[((size_t)pt * K + k) * P + (c - G)];
        };
        float key = q(0);
        int bk = 0;
        if (use_max)
            for (int k = 1; k < K; ++k) { const float v = q(k); if (v > key) { key = v; bk = k; } }
        if (argk) argk[u] = (uint8_t)bk;
        for (int k = 0; k < K; ++k) s[((size_t)pt * K + k) * C + c] = q(k) - key;
    }
}

// dq[pt,k,c] = ds[pt,k,c] - [k == key index] * sum_k' ds[pt,k',c];  gathered part scattered into dgx.
__global__ __launch_bounds__(BLOCK) void guidance_diff_bwd_kernel(const float* __restrict__ ds,
                                                                  const int64_t* __restrict__ idx,
                                                                  const uint8_t* __restrict__ argk, float* __restrict__ dgx,
                                                                  float* __restrict__ dpe, int N, int M, int K, int G, int P,
                                                                  long long units) {
    const int C = G + P;
    for (long long u = (long long)blockIdx.x * BLOCK + threadIdx.x; u < units; u += (long long)gridDim.x * BLOCK) {
        const long long pt = u / C;
        const int c = (int)(u - pt * C);
        const long long b = pt / M;
        float tot = 0.f;
        for (int k = 0; k < K; ++k) tot += ds[((size_t)pt * K + k) * C + c];
        const int kk = argk ? (int)argk[u] : 0;
        for (int k = 0; k < K; ++k) {
            float d = ds[((size_t)pt * K + k) * C + c];
            if (k == kk) d -= tot;
            if (c < G) {
                const int64_t j = idx[pt * K + k];
                if (j >= 0 && j < N) atomicAdd(dgx + ((size_t)(b * N + j)) * G + c, d);
            } else {
                dpe[((size_t)pt * K + k) * P + (c - G)] = d;
            }
        }
    }
}

// grid-stride kernels: every workgroup runs the same number of rounds (a capped grid with a ragged last round
// idles most of the chip for up to half of a short kernel)
static int grid_for(long long units) {
    const long long need = std::max<long long>(1, (units + BLOCK - 1) / BLOCK), cap = 256 * 32;
    const long long rounds = (need + cap - 1) / cap;
    return (int)((need + rounds - 1) / rounds);
}

}  // namespace pcf

extern "C" {

int pcf_hip_gather_rows(const float* table, const int64_t* idx, float* out, int B, int N, long long S, int C,
                        void* stream) {
    using namespace pcf;
    PCF_REQUIRE(B >= 0 && N >= 0 && S >= 0 && C >= 0, "gather_rows: negative size");
    const long long total = (long long)B * S * C;
    if (total == 0) return ok();
    PCF_REQUIRE(idx && out && (table || N == 0), "gather_rows: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (C % 4 == 0 && aligned16(table) && aligned16(out)) {
        const long long units = total / 4;
        hipLaunchKernelGGL(gather_rows_kernel<true>, dim3(grid_for(units)), dim3(BLOCK), 0, s, table, idx, out, N, S, C, units);
    } else {
        hipLaunchKernelGGL(gather_rows_kernel<false>, dim3(grid_for(total)), dim3(BLOCK), 0, s, table, idx, out, N, S, C, total);
    }
    return check_launch("gather_rows");
}

int pcf_hip_scatter_add_rows(const float* grad_rows, const int64_t* idx, float* grad_table, int B, int N, long long S,
                             int C, void* stream) {
    using namespace pcf;
    PCF_REQUIRE(B >= 0 && N >= 0 && S >= 0 && C >= 0, "scatter_add_rows: negative size");
    hipStream_t s = (hipStream_t)stream;
    if ((size_t)B * N * C) {
        PCF_REQUIRE(grad_table, "scatter_add_rows: grad_table is null");
        hipError_t e = zero_async(grad_table, (size_t)B * N * C * 4, s);
        if (e != hipSuccess) return fail(PCF_E_LAUNCH, "scatter_add_rows: memset: %s", hipGetErrorString(e));
    }
    const long long total = (long long)B * S * C;
    if (total == 0 || N == 0) return ok();
    PCF_REQUIRE(grad_rows && idx, "scatter_add_rows: null pointer");
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(grid_for(total)), dim3(BLOCK), 0, s, grad_rows, idx, grad_table, N, S, C, total);
    return check_launch("scatter_add_rows");
}

int pcf_hip_gather_max(const float* table, const int64_t* idx, float* out, uint8_t* argk, int B, int N, int M, int K,
                       int C, void* stream) {
    using namespace pcf;
    PCF_REQUIRE(B >= 0 && N >= 0 && M >= 0 && C >= 0 && K >= 1 && K <= 255, "gather_max: bad size (K must be 1..255)");
    const long long total = (long long)B * M * C;
    if (total == 0) return ok();
    PCF_REQUIRE(idx && out && argk && (table || N == 0), "gather_max: null pointer");
    hipLaunchKernelGGL(gather_max_kernel, dim3(grid_for(total)), dim3(BLOCK), 0, (hipStream_t)stream, table, idx, out, argk, N, M, K, C, total);
    return check_launch("gather_max");
}

int pcf_hip_gather_max_backward(const float* grad_out, const int64_t* idx, const uint8_t* argk, float* grad_table, int B,
                                int N, int M, int K, int C, void* stream) {
    using namespace pcf;
    PCF_REQUIRE(B >= 0 && N >= 0 && M >= 0 && C >= 0 && K >= 1 && K <= 255, "gather_max_backward: bad size");
    hipStream_t s = (hipStream_t)stream;
    if ((size_t)B * N * C) {
        PCF_REQUIRE(grad_table, "gather_max_backward: grad_table is null");
        hipError_t e = zero_async(grad_table, (size_t)B * N * C * 4, s);
        if (e != hipSuccess) return fail(PCF_E_LAUNCH, "gather_max_backward: memset: %s", hipGetErrorString(e));
    }
    const long long total = (long long)B * M * C;
    if (total == 0 || N == 0) return ok();
    PCF_REQUIRE(grad_out && idx && argk, "gather_max_backward: null pointer");
    hipLaunchKernelGGL(gather_max_backward_kernel, dim3(grid_for(total)), dim3(BLOCK), 0, s, grad_out, idx, argk, grad_table, N, M, K, C, total);
    return check_launch("gather_max_backward");
}

int pcf_hip_edge_geometry(const float* ref_xyz, const float* ref_norm, const int64_t* idx, const float* ctr_xyz,
                          const float* ctr_norm, float* rel, float* vi, int B, int N, int M, int K, void* stream) {
    using namespace pcf;
    PCF_REQUIRE(B >= 0 && N >= 0 && M >= 0 && K >= 1, "edge_geometry: bad size");
    const long long edges = (long long)B * M * K;
    if (edges == 0 || (!rel && !vi)) return ok();
    PCF_REQUIRE(ref_xyz && idx && ctr_xyz, "edge_geometry: null pointer");
    PCF_REQUIRE(!vi || (ref_norm && ctr_norm), "edge_geometry: VI output requested but normals are null");
    hipLaunchKernelGGL(edge_geometry_kernel, dim3(grid_for(edges)), dim3(BLOCK), 0, (hipStream_t)stream, ref_xyz, ref_norm,
                       idx, ctr_xyz, ctr_norm, rel, vi, N, M, K, edges, vi && aligned16(vi));
    return check_launch("edge_geometry");
}

int pcf_hip_vi_from_gathered(const float* rel, const float* nbr_norm, const float* ctr_norm, float* vi, int B, int M,
                             int K, void* stream) {
    using namespace pcf;
    PCF_REQUIRE(B >= 0 && M >= 0 && K >= 1, "vi_from_gathered: bad size");
    const long long edges = (long long)B * M * K;
    if (edges == 0) return ok();
    PCF_REQUIRE(rel && nbr_norm && ctr_norm && vi, "vi_from_gathered: null pointer");
    hipLaunchKernelGGL(vi_from_gathered_kernel, dim3(grid_for(edges)), dim3(BLOCK), 0, (hipStream_t)stream, rel, nbr_norm,
                       ctr_norm, vi, K, edges, aligned16(vi));
    return check_launch("vi_from_gathered");
}

int pcf_hip_guidance_diff_forward(const float* gx, const int64_t* idx, const float* pe, float* s, uint8_t* argk, int B,
                                  int N, int M, int K, int G, int P, int use_max, void* stream) {
    using namespace pcf;
    PCF_REQUIRE(B >= 0 && N >= 0 && M >= 0 && K >= 1 && K <= 255 && G >= 0 && P >= 0 && G + P >= 1, "guidance_diff: bad size");
    const long long units = (long long)B * M * (G + P);
    if (units == 0) return ok();
    PCF_REQUIRE(idx && s && (G == 0 || gx) && (P == 0 || pe), "guidance_diff: null pointer");
    PCF_REQUIRE(!use_max || argk, "guidance_diff: max key needs the argmax output");
    hipLaunchKernelGGL(guidance_diff_fwd_kernel, dim3(grid_for(units)), dim3(BLOCK), 0, (hipStream_t)stream, gx, idx, pe, s,
                       argk, N, M, K, G, P, use_max, units);
    return check_launch("guidance_diff forward");
}

int pcf_hip_guidance_diff_backward(const float* ds, const int64_t* idx, const uint8_t* argk, float* dgx, float* dpe, int B,
                                   int N, int M, int K, int G, int P, void* stream) {
    using namespace pcf;
    PCF_REQUIRE(B >= 0 && N >= 0 && M >= 0 && K >= 1 && K <= 255 && G >= 0 && P >= 0 && G + P >= 1, "guidance_diff_backward: bad size");
    hipStream_t st = (hipStream_t)stream;
    if ((size_t)B * N * G) {
        PCF_REQUIRE(dgx, "guidance_diff_backward: dgx is null");
        hipError_t e = zero_async(dgx, (size_t)B * N * G * 4, st);
        if (e != hipSuccess) return fail(PCF_E_LAUNCH, "guidance_diff_backward: memset: %s", hipGetErrorString(e));
    }
    const long long units = (long long)B * M * (G + P);
    if (units == 0) return ok();
    PCF_REQUIRE(ds && idx && (P == 0 || dpe), "guidance_diff_backward: null pointer");
    hipLaunchKernelGGL(guidance_diff_bwd_kernel, dim3(grid_for(units)), dim3(BLOCK), 0, st, ds, idx, argk, dgx, dpe, N, M, K,
                       G, P, units);
    return check_launch("guidance_diff backward");
}

}  // extern "C"
