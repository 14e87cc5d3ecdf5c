// Neighbourhood aggregate of PointConv / PointConvFormer, forward and backward, for gfx950.
//
//   out[n, c*Cm+m] = sum_k T[n,k,c] * w[n,k,m],   T[n,k,:] = [ x[idx[n,k], :] (*) guid[n,k, c%H] | add[n,k,:] ]
//
// One workgroup (4 waves) owns P consecutive output points.  The irregular part -- the K gathered
// rows per point -- is read with 16-byte loads that are contiguous inside a row, staged once in
// LDS next to the point's w / guid / grad_out tiles, and every contraction over K, C or Cm then
// runs out of LDS with one wave per point.  HBM traffic is therefore the algorithmic minimum:
// each index, weight, guidance value and output element is touched exactly once, gathered rows
// once per edge (served mostly from L2 / Infinity Cache: neighbourhoods overlap).
//
// Replaces pcf_cuda_forward/backward_kernel (pcf_ops.cu:27-141) and pconv_cuda_forward/
// backward_kernel (pconv_ops.cu:40-103,240-290).  Backward is the adjoint of the forward
// layout.  grad_x is either float-atomic scatter-add (no CSR available) or a plain
// store of every edge's contribution row for the CSR gather-reduce in csr_reduce_kernel.
#include <algorithm>
#include <atomic>
#include <cstdlib>

#include "pcf_common.h"

namespace pcf {

struct AggArgs {
    // forward inputs
    const float* x;        // [B*N, Ci]
    const int64_t* idx;    // [total, K]
    const float* guid;     // [total, K, H] or null
    const float* w;        // [total, K, Cm]
    const float* add;      // [total, K, Ca] or null
    float* out;            // fwd: [total, CT*Cm]
    // backward
    const float* gout;     // [total, CT*Cm]
    float* gx;             // [B*N, Ci]      atomic target (SCATTER_ATOMIC)
    float* contrib;        // [total*K, Ci]  per-edge rows (SCATTER_STORE)
    float* gguid;          // [total, K, H]
    float* gw;             // [total, K, Cm]
    float* gadd;           // [total, K, Ca]
    int total, N, Nout, K, Ci, Ca, Cm, H;
    int P;                 // points per workgroup
    int TS;                // LDS row stride of the gathered tile (floats)
    int GS;                // LDS row stride of the grad_out tile (floats)
    int offG, offT, offD, offW, offO;   // LDS offsets in floats (index table sits at 0)
};

__device__ __forceinline__ int pow2_ceil_dev(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

// ---- phases shared by forward and backward -----------------------------------------------------
// phase 0: neighbour table -> LDS as int32 row numbers into the flattened [B*N] input (-1 = skip),
//          guidance tile -> LDS.
// FX: the BASELINE layer shape (K = 16, Ci = 16, Ca = 0, H = 8; Cm = 16 through CM) as compile-time constants --
// the generic kernels spend more instructions on loop control and index arithmetic than on the products.
template <bool FX>
__device__ __forceinline__ void stage_index_and_guidance(const AggArgs& a, int n0, int np, int* sIdx, float* sG) {
    const int tid = threadIdx.x;
    const int K = FX ? 16 : a.K;
    const int H = FX ? 8 : a.H;
    for (int u = tid; u < np * K; u += BLOCK) {
        const int n = n0 + u / K;
        const int b = n / a.Nout;
        const int64_t j = a.idx[(size_t)n0 * K + u];
        sIdx[u] = (j >= 0 && j < a.N) ? (int)((int64_t)b * a.N + j) : -1;
    }
    if (a.guid) {
        const int cnt = np * K * H;
        const float* g = a.guid + (size_t)n0 * K * H;
        for (int u = tid; u < cnt; u += BLOCK) sG[u] = g[u];
    }
}

// phase 1: gathered rows (optionally head-modulated), appended features and weights -> LDS.
template <bool VROW, bool MODULATE, bool FX, int FXTS>
__device__ __forceinline__ void stage_tiles(const AggArgs& a, int n0, int np, const int* sIdx, const float* sG,
                                            float* sT, float* sW, int Cm) {
    const int tid = threadIdx.x;
    const int K = FX ? 16 : a.K, Ci = FX ? 16 : a.Ci, Ca = FX ? 0 : a.Ca, TS = FX ? FXTS : a.TS, H = FX ? 8 : a.H;
    const HeadMod hm(H > 0 ? H : 1);
    if (VROW) {
        const int ci4 = Ci >> 2;
        const int cnt = np * K * ci4;
#pragma unroll 4
        for (int u = tid; u < cnt; u += BLOCK) {
            const int pk = u / ci4;
            const int c = (u - pk * ci4) << 2;
            const int row = sIdx[pk];
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row >= 0) v = ld4(a.x + (size_t)row * Ci + c);
            if (MODULATE && a.guid) {
                const float* g = sG + pk * H;
                v.x *= g[hm(c)];
                v.y *= g[hm(c + 1)];
                v.z *= g[hm(c + 2)];
                v.w *= g[hm(c + 3)];
            }
            st4(sT + pk * TS + c, v);
        }
        if (Ca > 0) {
            const int ca4 = Ca >> 2;
            const int cnta = np * K * ca4;
            const float* src = a.add + (size_t)n0 * K * Ca;
#pragma unroll 2
            for (int u = tid; u < cnta; u += BLOCK) {
                const int pk = u / ca4;
                const int c = (u - pk * ca4) << 2;
                st4(sT + pk * TS + Ci + c, ld4(src + (size_t)u * 4));
            }
        }
    } else {
        const int cnt = np * K * Ci;
        for (int u = tid; u < cnt; u += BLOCK) {
            const int pk = u / Ci;
            const int c = u - pk * Ci;
            const int row = sIdx[pk];
            float v = row >= 0 ? a.x[(size_t)row * Ci + c] : 0.f;
            if (MODULATE && a.guid) v *= sG[pk * H + hm(c)];
            sT[pk * TS + c] = v;
        }
        if (Ca > 0) {
            const int cnta = np * K * Ca;
            const float* src = a.add + (size_t)n0 * K * Ca;
            for (int u = tid; u < cnta; u += BLOCK) {
                const int pk = u / Ca;
                sT[pk * TS + Ci + (u - pk * Ca)] = src[u];
            }
        }
    }
    // weights: a contiguous [np*K*Cm] run
    {
        const int cnt = np * K * Cm;
        const float* src = a.w + (size_t)n0 * K * Cm;
        if ((Cm & 3) == 0 && VROW) {
#pragma unroll 2
            for (int u = tid; u < (cnt >> 2); u += BLOCK) st4(sW + u * 4, ld4(src + (size_t)u * 4));
        } else {
            for (int u = tid; u < cnt; u += BLOCK) sW[u] = src[u];
        }
    }
}

// ---- forward ------------------------------------------------------------------------------------
// CM > 0: compile-time Cm (multiple of 4 -> four m per lane, else one); CM == 0: run-time Cm.
template <int CM, bool VROW, bool FX = false>
__global__ __launch_bounds__(BLOCK) void agg_fwd_kernel(const AggArgs a) {
    extern __shared__ __align__(16) float smem[];
    int* sIdx = reinterpret_cast<int*>(smem);
    float* sG = smem + a.offG;
    float* sT = smem + a.offT;
    float* sW = smem + a.offW;
    const int n0 = blockIdx.x * a.P;
    const int np = min(a.P, a.total - n0);
    const int K = FX ? 16 : a.K, CT = FX ? 16 : a.Ci + a.Ca, TS = FX ? 16 : a.TS;
    const int Cm = CM ? CM : a.Cm;

    stage_index_and_guidance<FX>(a, n0, np, sIdx, sG);
    __syncthreads();
    stage_tiles<VROW, true, FX, 16>(a, n0, np, sIdx, sG, sT, sW, Cm);
    __syncthreads();

    const int lane = lane_id();
    for (int p = wave_id(); p < np; p += NWAVE) {
        const float* tp = sT + p * K * TS;
        const float* wp = sW + p * K * Cm;
        float* op = a.out + (size_t)(n0 + p) * CT * Cm;
        if (CM > 0 && (CM & 3) == 0) {
            constexpr int MQ = CM > 0 ? (CM >> 2) : 1;
            for (int it = lane; it < CT * MQ; it += WAVE) {
                const int c = it / MQ;
                const int mv = (it - c * MQ) << 2;
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
                for (int k = 0; k < K; ++k) acc = fma4(tp[k * TS + c], ld4(wp + k * Cm + mv), acc);
                st4(op + (size_t)it * 4, acc);
            }
        } else {
            for (int it = lane; it < CT * Cm; it += WAVE) {
                const int c = it / Cm;
                const int m = it - c * Cm;
                float acc = 0.f;
#pragma unroll 4
                for (int k = 0; k < K; ++k) acc = fmaf(tp[k * TS + c], wp[k * Cm + m], acc);
                op[it] = acc;
            }
        }
    }
}

// ---- backward -----------------------------------------------------------------------------------
template <int CM, bool VROW, bool SCATTER_ATOMIC, bool FX = false>
__global__ __launch_bounds__(BLOCK) void agg_bwd_kernel(const AggArgs a) {
    extern __shared__ __align__(16) float smem[];
    int* sIdx = reinterpret_cast<int*>(smem);
    float* sG = smem + a.offG;
    float* sT = smem + a.offT;   // raw gathered rows | add ; gathered part becomes x*guid in phase 2
    float* sD = smem + a.offD;   // (dL/dT) * x_raw, reduced per head in phase 3 (guided only)
    float* sW = smem + a.offW;
    float* sO = smem + a.offO;   // grad_out tile, row stride GS
    const int tid = threadIdx.x;
    const int n0 = blockIdx.x * a.P;
    const int np = min(a.P, a.total - n0);
    const int K = FX ? 16 : a.K, Ci = FX ? 16 : a.Ci, Ca = FX ? 0 : a.Ca, CT = Ci + Ca, TS = FX ? 20 : a.TS,
              GS = FX ? 20 : a.GS, H = FX ? 8 : a.H;
    const int Cm = CM ? CM : a.Cm;
    const bool guided = a.guid != nullptr;
    const HeadMod hm(H > 0 ? H : 1);

    stage_index_and_guidance<FX>(a, n0, np, sIdx, sG);
    __syncthreads();
    stage_tiles<VROW, false, FX, 20>(a, n0, np, sIdx, sG, sT, sW, Cm);
    {   // grad_out: contiguous [np*CT*Cm] run -> rows of stride GS
        const float* src = a.gout + (size_t)n0 * CT * Cm;
        if (CM > 0 && (CM & 3) == 0 && VROW) {
            constexpr int MQ = CM > 0 ? (CM >> 2) : 1;
            const int cnt = np * CT * MQ;
#pragma unroll 2
            for (int u = tid; u < cnt; u += BLOCK) {
                const int r = u / MQ;
                st4(sO + r * GS + ((u - r * MQ) << 2), ld4(src + (size_t)u * 4));
            }
        } else {
            const int cnt = np * CT * Cm;
            for (int u = tid; u < cnt; u += BLOCK) {
                const int r = u / Cm;
                sO[r * GS + (u - r * Cm)] = src[u];
            }
        }
    }
    __syncthreads();

    const int lane = lane_id();
    // phase 2: dT[k,c] = sum_m gout[c,m] * w[k,m];  scatter / store its gathered part, write its appended part.
    {
        const int stride_c = CT >= WAVE ? WAVE : pow2_ceil_dev(CT);
        const int groups = WAVE / stride_c;
        const int cl = lane & (stride_c - 1);
        const int kg = lane / stride_c;
        for (int p = wave_id(); p < np; p += NWAVE) {
            const size_t n = (size_t)(n0 + p);
            for (int c = cl; c < CT; c += stride_c) {
                float go[CM > 0 ? CM : 1];
                if (CM > 0) {
                    if ((CM & 3) == 0) {
#pragma unroll
                        for (int q = 0; q < CM / 4; ++q) {
                            const float4 v = ld4(sO + (p * CT + c) * GS + q * 4);
                            go[q * 4 + 0] = v.x; go[q * 4 + 1] = v.y; go[q * 4 + 2] = v.z; go[q * 4 + 3] = v.w;
                        }
                    } else {
#pragma unroll
                        for (int m = 0; m < CM; ++m) go[m] = sO[(p * CT + c) * GS + m];
                    }
                }
                for (int k = kg; k < K; k += groups) {
                    const int pk = p * K + k;
                    const float* wr = sW + pk * Cm;
                    float dT = 0.f;
                    if (CM > 0) {
                        if ((CM & 3) == 0) {
#pragma unroll
                            for (int q = 0; q < CM / 4; ++q)
                                dT = dot4(make_float4(go[q * 4], go[q * 4 + 1], go[q * 4 + 2], go[q * 4 + 3]),
                                          ld4(wr + q * 4), dT);
                        } else {
#pragma unroll
                            for (int m = 0; m < CM; ++m) dT = fmaf(go[m], wr[m], dT);
                        }
                    } else {
                        const float* orow = sO + (p * CT + c) * GS;
                        for (int m = 0; m < Cm; ++m) dT = fmaf(orow[m], wr[m], dT);
                    }
                    if (c < Ci) {
                        float gh = 1.f;
                        if (guided) {
                            gh = sG[pk * H + hm(c)];
                            const float xr = sT[pk * TS + c];
                            sD[pk * TS + c] = dT * xr;
                            sT[pk * TS + c] = xr * gh;
                        }
                        if (SCATTER_ATOMIC) {
                            const int row = sIdx[pk];
                            if (row >= 0) atomicAdd(a.gx + (size_t)row * Ci + c, dT * gh);
                        } else {
                            a.contrib[(n * K + k) * Ci + c] = dT * gh;
                        }
                    } else {
                        a.gadd[(n * K + k) * Ca + (c - Ci)] = dT;
                    }
                }
            }
        }
    }
    __syncthreads();
    // phase 3: grad_w[k,m] = sum_c gout[c,m] * T[k,c];  grad_guid[k,h] = sum_{c%H==h} dT[k,c]*x[k,c]
    for (int p = wave_id(); p < np; p += NWAVE) {
        const size_t n = (size_t)(n0 + p);
        const float* orow = sO + p * CT * GS;
        if (CM > 0 && (CM & 3) == 0) {
            constexpr int MQ = CM > 0 ? (CM >> 2) : 1;
            for (int it = lane; it < K * MQ; it += WAVE) {
                const int k = it / MQ;
                const int mv = (it - k * MQ) << 2;
                const float* tr = sT + (p * K + k) * TS;
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
                for (int c = 0; c < CT; ++c) acc = fma4(tr[c], ld4(orow + c * GS + mv), acc);
                st4(a.gw + (n * K) * Cm + (size_t)it * 4, acc);
            }
        } else {
            for (int it = lane; it < K * Cm; it += WAVE) {
                const int k = it / Cm;
                const int m = it - k * Cm;
                const float* tr = sT + (p * K + k) * TS;
                float acc = 0.f;
#pragma unroll 4
                for (int c = 0; c < CT; ++c) acc = fmaf(tr[c], orow[c * GS + m], acc);
                a.gw[(n * K) * Cm + it] = acc;
            }
        }
        if (guided) {
            for (int it = lane; it < K * H; it += WAVE) {
                const int k = it / H;
                const int h = it - k * H;
                const float* dr = sD + (p * K + k) * TS;
                float acc = 0.f;
                for (int c = h; c < Ci; c += H) acc += dr[c];
                a.gguid[(n * K) * H + it] = acc;
            }
        }
    }
}

// ---- PCFLayer shapes (K = 16, H = 8, guided; written up at the BASELINE shape Ci = Cm = 16) on the matrix cores ------
// The LDS kernel above is bound by its own instruction streams at this shape (SQ counters: 46 us of VALU + 51 us of
// LDS for 80k points, not HBM): per point it runs two 16x16x16 contractions as 128 wave-wide FMAs fed from LDS.  They
// are exactly two v_mfma_f32_16x16x4_f32 chains (4 instructions each, exact fp32 products):
//   dT[k][c] = sum_m w[k][m] * gout[c][m]        A = w row k, B = gout row c: both operands are 16-byte loads of
//                                                row-major rows with the contraction step s covering m = 4q + s
//   gw[k][m] = sum_c T[k][c] * gout[c][m]        A = x row k (gathered, 16 B per lane) * guidance, B = gout column
// with q = lane >> 4; the accumulator leaves lane (lo = lane & 15, q) holding rows k = 4q..4q+3 of column lo, which
// is the layout the row-contiguous scatter (64-byte segments per k), grad_guid (pairs c, c+8 one DPP apart) and the
// grad_w stores want.  No LDS, no barriers: one wave per point, ~20 independent loads in flight.
// 4x4 transpose between the four registers of a lane and the four lanes of its quad (two DPP exchange stages):
// afterwards lane b holds in v[r] what lane r of the quad held in v[b].
template <int CTRL>
__device__ __forceinline__ float dpp_quad(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ void quad_transpose(float (&v)[4], int lane) {
    const bool b0 = lane & 1, b1 = lane & 2;
#pragma unroll
    for (int r = 0; r < 4; r += 2) {                  // partner lane ^ 1: quad_perm [1,0,3,2]
        const float recv = dpp_quad<0xb1>(b0 ? v[r] : v[r + 1]);
        if (b0) v[r] = recv; else v[r + 1] = recv;
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {                     // partner lane ^ 2: quad_perm [2,3,0,1]
        const float recv = dpp_quad<0x4e>(b1 ? v[r] : v[r + 2]);
        if (b1) v[r] = recv; else v[r + 2] = recv;
    }
}

typedef float v4f __attribute__((ext_vector_type(4)));
#define PCF_MFMA16(A, B, C) __builtin_amdgcn_mfma_f32_16x16x4f32((A), (B), (C), 0, 0, 0)

// Shapes the matrix-core kernels take: K = 16, H = 8, guided, no appended features, Ci a multiple of 16 (walked in
// 16-channel tiles) and C_mid = 16 (PCFLayers of the BASELINE configs: Ci 16..96) or C_mid = 4 (the 10cm-lite model).
// For C_mid = 4 a w / gout / out row is one 16-byte quad: product 1 is ONE matrix instruction (contraction length 4),
// and the 4-column operands / results live in lanes lo < 4 of every 16-lane row.
// CI: compile-time channel count (the BASELINE shape: one tile, no loop, constant strides) or 0 = a.Ci at run time.
template <int CM, int CI>
__global__ __launch_bounds__(BLOCK) void agg_bwd_mfma_kernel(const AggArgs a) {
    const int lane = lane_id();
    const int lo = lane & 15, q = lane >> 4;
    const int Ci = CI ? CI : a.Ci, ntile = Ci >> 4;
    for (int n = blockIdx.x * NWAVE + wave_id(); n < a.total; n += gridDim.x * NWAVE) {
        const int b = n / a.Nout;
        const size_t e0 = (size_t)n * 16;                        // first edge of the point
        const int64_t j = a.idx[e0 + lo];
        const int rowl = (j >= 0 && j < a.N) ? (int)((int64_t)b * a.N + j) : -1;       // neighbour row of k = lo
        const int kb = 4 * q + (lane & 3);
        const int rowb = __shfl(rowl, kb, WAVE);                                         // ... of k = 4q + b
        int rowk[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) rowk[r] = __shfl(rowl, 4 * q + r, WAVE);             // ... of k = 4q + r
        const float* go = a.gout + (size_t)n * Ci * CM;
        // per-point operands: w as the A operand of product 1, guidance in both layouts
        float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (CM == 16) w4 = ld4(a.w + (e0 + lo) * 16 + 4 * q);                           // w[k = lo][m = 4q + s]
        else w4.x = a.w[(e0 + lo) * 4 + q];                                              // w[k = lo][m = q]
        const float4 g4 = ld4(a.guid + (e0 + lo) * 8 + 4 * (q & 1));                     // guid[k = lo][(4q + s) % 8]
        const float4 gsT = ld4(a.guid + (e0 + kb) * 8 + (lo & 4));                       // guid[k = 4q + b][4(a & 1)..]
        v4f gw = {0.f, 0.f, 0.f, 0.f};                                                   // grad_w[k = 4q + r][m = lo]
        float pr[4] = {0.f, 0.f, 0.f, 0.f};
        // one 16-channel tile of operands: gout rows for product 1 (A side: row pieces m = 4q + s), and x / gout in the
        // accumulator's layout as 16-byte row pieces for the quad transpose
        struct Tile { float4 go4, xT, x4, goT; };
        auto load_tile = [&](int c0, Tile& t) {
            t.go4 = t.xT = t.x4 = t.goT = make_float4(0.f, 0.f, 0.f, 0.f);
            if (CM == 16) t.go4 = ld4(go + (size_t)(c0 + lo) * 16 + 4 * q);             // gout[c = c0 + lo][m = 4q + s]
            else t.go4.x = go[(size_t)(c0 + lo) * 4 + q];                                // gout[c = c0 + lo][m = q]
            if (rowb >= 0) t.xT = ld4(a.x + (size_t)rowb * Ci + c0 + (lo & ~3));         // x[k = 4q + b][c0 + 4a..]
            if (rowl >= 0) t.x4 = ld4(a.x + (size_t)rowl * Ci + c0 + 4 * q);             // x[k = lo][c0 + 4q + s]
            if (CM == 16) t.goT = ld4(go + (size_t)(c0 + kb) * 16 + (lo & ~3));          // gout[c0 + 4q + b][m = 4a..]
            else if (lo < 4) t.goT = ld4(go + (size_t)(c0 + kb) * 4);                    // gout[c0 + 4q + b][m = 0..3]
        };
        Tile cur;
        load_tile(0, cur);
        float gs[4] = {gsT.x, gsT.y, gsT.z, gsT.w};
        quad_transpose(gs, lane);                                                        // -> guid[k = 4q + r][lo % 8]
        for (int ct = 0; ct < ntile; ++ct) {
            const int c0 = 16 * ct;
            Tile nxt = cur;
            if (ct + 1 < ntile) load_tile(c0 + 16, nxt);          // the next tile's loads go out before this tile's atomics
            // product 1: dT[k = 4q + r][c = c0 + lo] = sum_m w[k][m] * gout[c][m]; both operands are row loads
            v4f dT = {0.f, 0.f, 0.f, 0.f};
            dT = PCF_MFMA16(w4.x, cur.go4.x, dT);
            if (CM == 16) {
                dT = PCF_MFMA16(w4.y, cur.go4.y, dT); dT = PCF_MFMA16(w4.z, cur.go4.z, dT); dT = PCF_MFMA16(w4.w, cur.go4.w, dT);
            }
            float xs[4] = {cur.xT.x, cur.xT.y, cur.xT.z, cur.xT.w};
            float gB[4] = {cur.goT.x, cur.goT.y, cur.goT.z, cur.goT.w};
            quad_transpose(xs, lane);                                                    // -> x[k = 4q + r][c = c0 + lo]
            quad_transpose(gB, lane);                                                    // -> gout[c0 + 4q + r][m = lo]
            // product 2: grad_w[k][m] += sum_c T[k][c] * gout[c][m], T = x * guid
            gw = PCF_MFMA16(cur.x4.x * g4.x, gB[0], gw); gw = PCF_MFMA16(cur.x4.y * g4.y, gB[1], gw);
            gw = PCF_MFMA16(cur.x4.z * g4.z, gB[2], gw); gw = PCF_MFMA16(cur.x4.w * g4.w, gB[3], gw);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (rowk[r] >= 0) atomicAdd(a.gx + (size_t)rowk[r] * Ci + c0 + lo, dT[r] * gs[r]);
                pr[r] = fmaf(dT[r], xs[r], pr[r]);                                       // grad_guid[k][h]: sum over c % 8 == h
            }
            cur = nxt;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) pr[r] += __shfl_xor(pr[r], 8, WAVE);
        // stores: the 4x4 transpose inside every quad of lanes turns "4 rows x one column" into "one row x 4 columns",
        // so grad_w leaves as ONE 16-byte store per lane covering the point's contiguous rows (four stores of 64-byte
        // pieces ran at 3.7 TB/s) and grad_guid as one per lane of the lower half-rows (512 contiguous bytes)
        float gwr[4] = {gw[0], gw[1], gw[2], gw[3]};
        quad_transpose(gwr, lane);
        quad_transpose(pr, lane);
        const float4 gwv = make_float4(gwr[0], gwr[1], gwr[2], gwr[3]);
        if (CM == 16) st4(a.gw + (e0 + kb) * 16 + (lo & ~3), gwv);
        else if (lo < 4) st4(a.gw + (e0 + kb) * 4, gwv);
        if (lo < 8) st4(a.gguid + (e0 + kb) * 8 + (lo & ~3), make_float4(pr[0], pr[1], pr[2], pr[3]));
    }
}

// Forward at the same shapes: out[c][m] = sum_k T[k][c] * w[k][m], T = x[idx[k]][c] * guid[k][c % 8].  A (lane (c = lo, q),
// step s <-> k = 4q + s) and B (w[k = 4q + s][m = lo]) are both "rows 4q..4q+3 of column lo": 16-byte loads of row
// pieces + the quad transpose; the accumulator (rows c = 4q + r of column m) goes back through the transpose and
// leaves as one 16-byte store per lane = the tile's contiguous rows.
template <int CM, int CI>
__global__ __launch_bounds__(BLOCK) void agg_fwd_mfma_kernel(const AggArgs a) {
    const int lane = lane_id();
    const int lo = lane & 15, q = lane >> 4;
    const int Ci = CI ? CI : a.Ci, ntile = Ci >> 4;
    for (int n = blockIdx.x * NWAVE + wave_id(); n < a.total; n += gridDim.x * NWAVE) {
        const int b = n / a.Nout;
        const size_t e0 = (size_t)n * 16;
        const int kb = 4 * q + (lane & 3);
        const int64_t j = a.idx[e0 + kb];
        const int rowb = (j >= 0 && j < a.N) ? (int)((int64_t)b * a.N + j) : -1;
        float4 wT = make_float4(0.f, 0.f, 0.f, 0.f);
        if (CM == 16) wT = ld4(a.w + (e0 + kb) * 16 + (lo & ~3));                        // w[k = 4q + b][m = 4a..]
        else if (lo < 4) wT = ld4(a.w + (e0 + kb) * 4);                                  // w[k = 4q + b][m = 0..3]
        const float4 gsT = ld4(a.guid + (e0 + kb) * 8 + (lo & 4));                       // guid[k = 4q + b][4(a & 1)..]
        float wB[4] = {wT.x, wT.y, wT.z, wT.w};
        quad_transpose(wB, lane);                                                        // -> w[k = 4q + r][m = lo]
        float* out = a.out + (size_t)n * Ci * CM;
        for (int ct = 0; ct < ntile; ++ct) {
            const int c0 = 16 * ct;
            float4 xT = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rowb >= 0) xT = ld4(a.x + (size_t)rowb * Ci + c0 + (lo & ~3));           // x[k = 4q + b][c0 + 4a..]
            float t[4] = {xT.x * gsT.x, xT.y * gsT.y, xT.z * gsT.z, xT.w * gsT.w};       // T[k = 4q + b][c0 + 4a..]
            quad_transpose(t, lane);                                                     // -> T[k = 4q + r][c = c0 + lo]
            v4f acc = {0.f, 0.f, 0.f, 0.f};                                              // out[c = c0 + 4q + r][m = lo]
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) acc = PCF_MFMA16(t[s2], wB[s2], acc);
            float o[4] = {acc[0], acc[1], acc[2], acc[3]};
            quad_transpose(o, lane);                                                     // -> out[c = c0 + 4q + b][m = 4a..]
            const float4 ov = make_float4(o[0], o[1], o[2], o[3]);
            if (CM == 16) st4(out + (size_t)(c0 + kb) * 16 + (lo & ~3), ov);
            else if (lo < 4) st4(out + (size_t)(c0 + kb) * 4, ov);
        }
    }
}

// Unguided form (PointConvStridePE of the BASELINE configs: Ci = 16 gathered + Ca = 16 appended channels, K = 16): the
// same two kernels with T[k][c] = x[idx[k]][c] for c < Ci and add[n, k, c - Ci] behind it, walked in 16-channel tiles
// (both widths multiples of 16).  Backward: gathered tiles of dT leave as float atomics (ATOMIC) or as the contribution
// rows of the CSR reduce, appended tiles as grad_add -- both through the quad transpose as 16-byte stores.
template <int CM>
__global__ __launch_bounds__(BLOCK) void pconv_fwd_mfma_kernel(const AggArgs a) {
    const int lane = lane_id();
    const int lo = lane & 15, q = lane >> 4;
    const int Ci = a.Ci, Ca = a.Ca, CT = Ci + Ca, ntile = CT >> 4;
    for (int n = blockIdx.x * NWAVE + wave_id(); n < a.total; n += gridDim.x * NWAVE) {
        const int b = n / a.Nout;
        const size_t e0 = (size_t)n * 16;
        const int kb = 4 * q + (lane & 3);
        const int64_t j = a.idx[e0 + kb];
        const int rowb = (j >= 0 && j < a.N) ? (int)((int64_t)b * a.N + j) : -1;
        float4 wT = make_float4(0.f, 0.f, 0.f, 0.f);
        if (CM == 16) wT = ld4(a.w + (e0 + kb) * 16 + (lo & ~3));                        // w[k = 4q + b][m = 4a..]
        else if (lo < 4) wT = ld4(a.w + (e0 + kb) * 4);                                  // w[k = 4q + b][m = 0..3]
        float wB[4] = {wT.x, wT.y, wT.z, wT.w};
        quad_transpose(wB, lane);                                                        // -> w[k = 4q + r][m = lo]
        float* out = a.out + (size_t)n * CT * CM;
        for (int ct = 0; ct < ntile; ++ct) {
            const int c0 = 16 * ct;
            float4 xT = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c0 < Ci) { if (rowb >= 0) xT = ld4(a.x + (size_t)rowb * Ci + c0 + (lo & ~3)); }     // x[k = 4q + b][c0 + 4a..]
            else xT = ld4(a.add + (e0 + kb) * Ca + (c0 - Ci) + (lo & ~3));                            // add[k = 4q + b][..]
            float t[4] = {xT.x, xT.y, xT.z, xT.w};
            quad_transpose(t, lane);                                                     // -> T[k = 4q + r][c = c0 + lo]
            v4f acc = {0.f, 0.f, 0.f, 0.f};                                              // out[c = c0 + 4q + r][m = lo]
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) acc = PCF_MFMA16(t[s2], wB[s2], acc);
            float o[4] = {acc[0], acc[1], acc[2], acc[3]};
            quad_transpose(o, lane);                                                     // -> out[c = c0 + 4q + b][m = 4a..]
            const float4 ov = make_float4(o[0], o[1], o[2], o[3]);
            if (CM == 16) st4(out + (size_t)(c0 + kb) * 16 + (lo & ~3), ov);
            else if (lo < 4) st4(out + (size_t)(c0 + kb) * 4, ov);
        }
    }
}

template <int CM, bool ATOMIC>
__global__ __launch_bounds__(BLOCK) void pconv_bwd_mfma_kernel(const AggArgs a) {
    const int lane = lane_id();
    const int lo = lane & 15, q = lane >> 4;
    const int Ci = a.Ci, Ca = a.Ca, CT = Ci + Ca, ntile = CT >> 4;
    for (int n = blockIdx.x * NWAVE + wave_id(); n < a.total; n += gridDim.x * NWAVE) {
        const int b = n / a.Nout;
        const size_t e0 = (size_t)n * 16;
        const int64_t j = a.idx[e0 + lo];
        const int rowl = (j >= 0 && j < a.N) ? (int)((int64_t)b * a.N + j) : -1;       // neighbour row of k = lo
        const int kb = 4 * q + (lane & 3);
        int rowk[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) rowk[r] = __shfl(rowl, 4 * q + r, WAVE);             // ... of k = 4q + r
        const float* go = a.gout + (size_t)n * CT * CM;
        float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (CM == 16) w4 = ld4(a.w + (e0 + lo) * 16 + 4 * q);                           // w[k = lo][m = 4q + s]
        else w4.x = a.w[(e0 + lo) * 4 + q];                                              // w[k = lo][m = q]
        v4f gw = {0.f, 0.f, 0.f, 0.f};                                                   // grad_w[k = 4q + r][m = lo]
        for (int ct = 0; ct < ntile; ++ct) {
            const int c0 = 16 * ct;
            const bool gathered = c0 < Ci;
            float4 go4 = make_float4(0.f, 0.f, 0.f, 0.f), x4 = go4, goT = go4;
            if (CM == 16) go4 = ld4(go + (size_t)(c0 + lo) * 16 + 4 * q);                // gout[c = c0 + lo][m = 4q + s]
            else go4.x = go[(size_t)(c0 + lo) * 4 + q];                                  // gout[c = c0 + lo][m = q]
            if (gathered) { if (rowl >= 0) x4 = ld4(a.x + (size_t)rowl * Ci + c0 + 4 * q); }         // T[k = lo][c0 + 4q + s]
            else x4 = ld4(a.add + (e0 + lo) * Ca + (c0 - Ci) + 4 * q);
            if (CM == 16) goT = ld4(go + (size_t)(c0 + kb) * 16 + (lo & ~3));            // gout[c0 + 4q + b][m = 4a..]
            else if (lo < 4) goT = ld4(go + (size_t)(c0 + kb) * 4);                      // gout[c0 + 4q + b][m = 0..3]
            v4f dT = {0.f, 0.f, 0.f, 0.f};                                               // dT[k = 4q + r][c = c0 + lo]
            dT = PCF_MFMA16(w4.x, go4.x, dT);
            if (CM == 16) { dT = PCF_MFMA16(w4.y, go4.y, dT); dT = PCF_MFMA16(w4.z, go4.z, dT); dT = PCF_MFMA16(w4.w, go4.w, dT); }
            float gB[4] = {goT.x, goT.y, goT.z, goT.w};
            quad_transpose(gB, lane);                                                    // -> gout[c0 + 4q + r][m = lo]
            gw = PCF_MFMA16(x4.x, gB[0], gw); gw = PCF_MFMA16(x4.y, gB[1], gw);
            gw = PCF_MFMA16(x4.z, gB[2], gw); gw = PCF_MFMA16(x4.w, gB[3], gw);
            if (gathered && ATOMIC) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (rowk[r] >= 0) atomicAdd(a.gx + (size_t)rowk[r] * Ci + c0 + lo, dT[r]);
            } else {
                float d[4] = {dT[0], dT[1], dT[2], dT[3]};
                quad_transpose(d, lane);                                                 // -> dT[k = 4q + b][c0 + 4a..]
                const float4 dv = make_float4(d[0], d[1], d[2], d[3]);
                if (gathered) st4(a.contrib + (e0 + kb) * Ci + c0 + (lo & ~3), dv);
                else st4(a.gadd + (e0 + kb) * Ca + (c0 - Ci) + (lo & ~3), dv);
            }
        }
        float gwr[4] = {gw[0], gw[1], gw[2], gw[3]};
        quad_transpose(gwr, lane);
        const float4 gwv = make_float4(gwr[0], gwr[1], gwr[2], gwr[3]);
        if (CM == 16) st4(a.gw + (e0 + kb) * 16 + (lo & ~3), gwv);
        else if (lo < 4) st4(a.gw + (e0 + kb) * 4, gwv);
    }
}

// The BASELINE shape (Ci = Cm = 16) written out without the tile loop: every load of a point is issued before the first
// use (measured 89 us against 96 us for the tiled form instantiated at Ci = 16, 80k points).
__global__ __launch_bounds__(BLOCK) void agg_bwd_fx_mfma_kernel(const AggArgs a) {
    const int lane = lane_id();
    const int lo = lane & 15, q = lane >> 4;
    for (int n = blockIdx.x * NWAVE + wave_id(); n < a.total; n += gridDim.x * NWAVE) {
        const int b = n / a.Nout;
        const size_t e0 = (size_t)n * 16;                        // first edge of the point
        const int64_t j = a.idx[e0 + lo];
        const int rowl = (j >= 0 && j < a.N) ? (int)((int64_t)b * a.N + j) : -1;       // neighbour row of k = lo
        const float* go = a.gout + (size_t)n * 256;
        // operands of both products; everything is issued before the first use
        const float4 w4 = ld4(a.w + (e0 + lo) * 16 + 4 * q);                          // w[k = lo][m = 4q + s]
        const float4 go4 = ld4(go + lo * 16 + 4 * q);                                 // gout[c = lo][m = 4q + s]
        const float4 g4 = ld4(a.guid + (e0 + lo) * 8 + 4 * (q & 1));                  // guid[k = lo][(4q + s) % 8]
        float4 x4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rowl >= 0) x4 = ld4(a.x + (size_t)rowl * 16 + 4 * q);                     // x[k = lo][c = 4q + s]
        // the same three tiles in the accumulator's layout (rows k = 4q + r of column lo): 16-byte loads of row pieces
        // (lane (4a + b, q): row 4q + b, columns 4a..4a+3 -- a contiguous KiB per instruction for gout) turned by the
        // quad transpose; four scalar loads of 64-byte pieces each were slower
        const int kb = 4 * q + (lane & 3);
        const int rowb = __shfl(rowl, kb, WAVE);
        const float4 goT = ld4(go + kb * 16 + (lo & ~3));                                // gout[c = 4q + b][m = 4a..]
        const float4 gsT = ld4(a.guid + (e0 + kb) * 8 + (lo & 4));                       // guid[k = 4q + b][4(a & 1)..]
        float4 xT = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rowb >= 0) xT = ld4(a.x + (size_t)rowb * 16 + (lo & ~3));                   // x[k = 4q + b][c = 4a..]
        int rowk[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) rowk[r] = __shfl(rowl, 4 * q + r, WAVE);             // neighbour row of k = 4q + r
        float gB[4] = {goT.x, goT.y, goT.z, goT.w};                                      // -> gout[c = 4q + r][m = lo]
        float gs[4] = {gsT.x, gsT.y, gsT.z, gsT.w};                                      // -> guid[k = 4q + r][lo % 8]
        float xs[4] = {xT.x, xT.y, xT.z, xT.w};                                          // -> x[k = 4q + r][c = lo]
        quad_transpose(gB, lane);
        quad_transpose(gs, lane);
        quad_transpose(xs, lane);

        v4f dT = {0.f, 0.f, 0.f, 0.f};                                                // dT[k = 4q + r][c = lo]
        dT = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.x, go4.x, dT, 0, 0, 0);
        dT = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.y, go4.y, dT, 0, 0, 0);
        dT = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.z, go4.z, dT, 0, 0, 0);
        dT = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.w, go4.w, dT, 0, 0, 0);
        v4f gw = {0.f, 0.f, 0.f, 0.f};                                                // grad_w[k = 4q + r][m = lo]
        gw = __builtin_amdgcn_mfma_f32_16x16x4f32(x4.x * g4.x, gB[0], gw, 0, 0, 0);
        gw = __builtin_amdgcn_mfma_f32_16x16x4f32(x4.y * g4.y, gB[1], gw, 0, 0, 0);
        gw = __builtin_amdgcn_mfma_f32_16x16x4f32(x4.z * g4.z, gB[2], gw, 0, 0, 0);
        gw = __builtin_amdgcn_mfma_f32_16x16x4f32(x4.w * g4.w, gB[3], gw, 0, 0, 0);
        float pr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (rowk[r] >= 0) atomicAdd(a.gx + (size_t)rowk[r] * 16 + lo, dT[r] * gs[r]);
            pr[r] = dT[r] * xs[r];                                                    // grad_guid[k][h] = sum over c % 8 == h
            pr[r] += __shfl_xor(pr[r], 8, WAVE);
        }
        // stores: a 4x4 transpose inside every quad of lanes turns "4 rows x one column" into "one row x 4 columns",
        // so grad_w leaves as ONE 16-byte store per lane covering the point's contiguous KiB (four stores of 64-byte
        // pieces ran at 3.7 TB/s) and grad_guid as one per lane of the lower half-rows (512 contiguous bytes)
        float gwr[4] = {gw[0], gw[1], gw[2], gw[3]};
        quad_transpose(gwr, lane);
        quad_transpose(pr, lane);
        const size_t e = e0 + 4 * q + (lane & 3);
        st4(a.gw + e * 16 + (lo & ~3), make_float4(gwr[0], gwr[1], gwr[2], gwr[3]));
        if (lo < 8) st4(a.gguid + e * 8 + (lo & ~3), make_float4(pr[0], pr[1], pr[2], pr[3]));
    }
}

// Forward at the same shape: out[c][m] = sum_k T[k][c] * w[k][m], T = x[idx[k]][c] * guid[k][c % 8].  A (lane (c = lo, q),
// step s <-> k = 4q + s) and B (w[k = 4q + s][m = lo]) are both "rows 4q..4q+3 of column lo": 16-byte loads of row
// pieces + the quad transpose; the accumulator (rows c = 4q + r of column m) goes back through the transpose and
// leaves as one 16-byte store per lane = the point's contiguous KiB.
__global__ __launch_bounds__(BLOCK) void agg_fwd_fx_mfma_kernel(const AggArgs a) {
    const int lane = lane_id();
    const int lo = lane & 15, q = lane >> 4;
    for (int n = blockIdx.x * NWAVE + wave_id(); n < a.total; n += gridDim.x * NWAVE) {
        const int b = n / a.Nout;
        const size_t e0 = (size_t)n * 16;
        const int kb = 4 * q + (lane & 3);
        const int64_t j = a.idx[e0 + kb];
        const int rowb = (j >= 0 && j < a.N) ? (int)((int64_t)b * a.N + j) : -1;
        const float4 wT = ld4(a.w + (e0 + kb) * 16 + (lo & ~3));                         // w[k = 4q + b][m = 4a..]
        const float4 gsT = ld4(a.guid + (e0 + kb) * 8 + (lo & 4));                       // guid[k = 4q + b][4(a & 1)..]
        float4 xT = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rowb >= 0) xT = ld4(a.x + (size_t)rowb * 16 + (lo & ~3));                   // x[k = 4q + b][c = 4a..]
        float t[4] = {xT.x * gsT.x, xT.y * gsT.y, xT.z * gsT.z, xT.w * gsT.w};           // T[k = 4q + b][c = 4a..] (c % 8 = 4(a & 1)..)
        float wB[4] = {wT.x, wT.y, wT.z, wT.w};
        quad_transpose(t, lane);                                                         // -> T[k = 4q + r][c = lo]
        quad_transpose(wB, lane);                                                        // -> w[k = 4q + r][m = lo]
        v4f acc = {0.f, 0.f, 0.f, 0.f};                                                  // out[c = 4q + r][m = lo]
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(t[s2], wB[s2], acc, 0, 0, 0);
        float o[4] = {acc[0], acc[1], acc[2], acc[3]};
        quad_transpose(o, lane);                                                         // -> out[c = 4q + b][m = 4a..]
        st4(a.out + (size_t)n * 256 + kb * 16 + (lo & ~3), make_float4(o[0], o[1], o[2], o[3]));
    }
}

// ---- CSR gather-reduce of the per-edge contribution rows -----------------------------------------
// grad_x[r, :] = sum over the inverse list of input row r of contrib[(n*K + k), :], in list order
// (deterministic).  One wave per input row; lane groups take alternate list entries.
// Stands in for input_only_backward_kernel (pconv_ops.cu:539-619) but covers every row.
__global__ __launch_bounds__(BLOCK) void csr_reduce_kernel(const float* __restrict__ contrib,
                                                           const int32_t* __restrict__ inv_n,
                                                           const uint8_t* __restrict__ inv_k,
                                                           const int32_t* __restrict__ inv_idx, float* __restrict__ gx,
                                                           int B, int N, int Nout, int K, int Ci, int inv_len,
                                                           int inv_idx_len) {
    const int lane = lane_id();
    const int stride_c = Ci >= WAVE ? WAVE : pow2_ceil_dev(Ci);
    const int groups = WAVE / stride_c;
    const int cl = lane & (stride_c - 1);
    const int eg = lane / stride_c;
    const long long rows = (long long)B * N;
    for (long long r = (long long)blockIdx.x * NWAVE + wave_id(); r < rows; r += (long long)gridDim.x * NWAVE) {
        const int b = (int)(r / N);
        const int p = (int)(r - (long long)b * N);
        int beg = inv_idx[(size_t)b * inv_idx_len + p];
        int end = inv_idx[(size_t)b * inv_idx_len + p + 1];
        beg = max(0, min(beg, inv_len));
        end = max(beg, min(end, inv_len));
        const int32_t* ln = inv_n + (size_t)b * inv_len;
        const uint8_t* lk = inv_k + (size_t)b * inv_len;
        for (int c0 = 0; c0 < Ci; c0 += stride_c) {
            const int c = c0 + cl;
            float acc = 0.f;
            if (c < Ci) {
                for (int j = beg + eg; j < end; j += groups) {
                    const int n = ln[j];
                    const int k = lk[j];
                    if (n >= 0 && n < Nout && k < K) acc += contrib[(((size_t)b * Nout + n) * K + k) * Ci + c];
                }
            }
            for (int off = stride_c; off < WAVE; off <<= 1) acc += __shfl_xor(acc, off, WAVE);
            if (eg == 0 && c < Ci) gx[(size_t)r * Ci + c] = acc;
        }
    }
}


// ---- C_mid = 1, unguided: the decoder's PointConvTransposePE (mid_dim_back = 1, Ci 128..384) ---------------
// out[n, c] = sum_k w[n,k] * T[n,k,c] is a weighted sum of K rows and its backward two more of the same shape, so
// nothing is staged in LDS: one wave per output point, one lane per 16-byte quad of the row, the K row loads of a
// point all in flight at once (the generic kernels above run one lane per (k, m) pair = 16 of 64 lanes here).
template <bool AL>
__device__ __forceinline__ float4 ldq(const float* p) {
    if (AL) return ld4(p);
    return make_float4(p[0], p[1], p[2], p[3]);
}
template <bool AL>
__device__ __forceinline__ void stq(float* p, float4 v) {
    if (AL) { st4(p, v); return; }
    p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
}
__device__ __forceinline__ float lane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
    return v;
}

// neighbour rows (-1 = skip) and weights of point n, one per lane k < K <= 64
__device__ __forceinline__ void agg1_point(const AggArgs& a, int n, int lane, int& row, float& wk) {
    row = -1; wk = 0.f;
    if (lane < a.K) {
        const int b = n / a.Nout;
        const int64_t j = a.idx[(size_t)n * a.K + lane];
        if (j >= 0 && j < a.N) row = (int)((int64_t)b * a.N + j);
        wk = a.w[(size_t)n * a.K + lane];
    }
}

template <bool AL>
__global__ __launch_bounds__(BLOCK) void agg1_fwd_kernel(const AggArgs a) {
    const int lane = lane_id();
    const int K = a.K, Ci = a.Ci, Ca = a.Ca, CT = Ci + Ca, Q = CT >> 2, QI = Ci >> 2;
    for (int n = blockIdx.x * NWAVE + wave_id(); n < a.total; n += gridDim.x * NWAVE) {
        int row; float wk;
        agg1_point(a, n, lane, row, wk);
        for (int q0 = 0; q0 < Q; q0 += WAVE) {
            const int q = q0 + lane;
            const bool gathered = q < QI, appended = !gathered && q < Q;
            const float* ap = a.add + (size_t)n * K * Ca + (appended ? (q - QI) * 4 : 0);
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int k0 = 0; k0 < K; k0 += 8) {          // eight rows in flight (lanes >= K hold row -1, weight 0)
                float4 v[8];
                float wv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = min(k0 + u, WAVE - 1);
                    const int r = __builtin_amdgcn_readlane(row, k);
                    wv[u] = lane_f(wk, k);
                    v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (gathered) { if (r >= 0) v[u] = ldq<AL>(a.x + (size_t)r * Ci + q * 4); }
                    else if (appended && k0 + u < K) v[u] = ldq<AL>(ap + (size_t)k * Ca);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = fma4(wv[u], v[u], acc);
            }
            if (q < Q) stq<AL>(a.out + (size_t)n * CT + q * 4, acc);
        }
    }
}

// grad_w[n,k] = <gout[n,:], T[n,k,:]>, grad_add[n,k,:] = w[n,k] * gout[n, Ci:], and for ATOMIC the scatter
// grad_x[idx[n,k], :] += w[n,k] * gout[n, :Ci]; otherwise grad_x comes from csr_gather1_kernel below.
// NP = 64-lane passes over the quads of a row (CT <= 256 * NP).
template <bool AL, bool ATOMIC, int NP>
__global__ __launch_bounds__(BLOCK) void agg1_bwd_kernel(const AggArgs a) {
    const int lane = lane_id();
    const int K = a.K, Ci = a.Ci, Ca = a.Ca, CT = Ci + Ca, Q = CT >> 2, QI = Ci >> 2;
    for (int n = blockIdx.x * NWAVE + wave_id(); n < a.total; n += gridDim.x * NWAVE) {
        int row; float wk;
        agg1_point(a, n, lane, row, wk);
        float4 g[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int q = p * WAVE + lane;
            g[p] = q < Q ? ldq<AL>(a.gout + (size_t)n * CT + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float mine = 0.f;
        for (int k0 = 0; k0 < K; k0 += 4) {              // four rows in flight (lanes >= K hold row -1, weight 0)
            float part[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = min(k0 + u, WAVE - 1);
                const int r = __builtin_amdgcn_readlane(row, k);
                const float wv = lane_f(wk, k);
                part[u] = 0.f;
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const int q = p * WAVE + lane;
                    if (q < QI) {
                        if (r >= 0) {
                            part[u] = dot4(g[p], ldq<AL>(a.x + (size_t)r * Ci + q * 4), part[u]);
                            if (ATOMIC) {
                                float* dst = a.gx + (size_t)r * Ci + q * 4;
                                atomicAdd(dst + 0, wv * g[p].x); atomicAdd(dst + 1, wv * g[p].y);
                                atomicAdd(dst + 2, wv * g[p].z); atomicAdd(dst + 3, wv * g[p].w);
                            }
                        }
                    } else if (q < Q && k0 + u < K) {
                        const size_t o = ((size_t)n * K + k) * Ca + (q - QI) * 4;
                        part[u] = dot4(g[p], ldq<AL>(a.add + o), part[u]);
                        stq<AL>(a.gadd + o, make_float4(wv * g[p].x, wv * g[p].y, wv * g[p].z, wv * g[p].w));
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float t = wave_sum(part[u]);
                if (lane == k0 + u) mine = t;
            }
        }
        if (lane < K) a.gw[(size_t)n * K + lane] = mine;
    }
}

// grad_x[r, :] = sum over the inverse list of input row r of w[n,k] * dout[n, :Ci]  (C_mid = 1): the rows of the
// aggregate's output gradient [total, J] are gathered directly -- no per-edge contribution rows are written and read
// back (2 x E x Ci x 4 B at the finest level = 2.4 GB per layer).  One wave per input row; 64 list entries' (n, k, w)
// are fetched by the lanes at once, then lane groups of LPE lanes walk alternate entries in a fixed order
// (deterministic) and are summed by xor-shuffles.  Stands in for input_only_backward_kernel (pconv_ops.cu:539-619).
template <bool AL, int NP>
__global__ __launch_bounds__(BLOCK) void csr_gather1_kernel(const float* __restrict__ dout, const float* __restrict__ w,
                                                            const int32_t* __restrict__ inv_n,
                                                            const uint8_t* __restrict__ inv_k,
                                                            const int32_t* __restrict__ inv_idx, float* __restrict__ gx,
                                                            int B, int N, int Nout, int K, int Ci, int J, int inv_len,
                                                            int inv_idx_len) {
    const int lane = lane_id();
    const int QI = Ci >> 2;
    const int LPE = QI >= WAVE ? WAVE : pow2_ceil_dev(QI);     // lanes per list entry
    const int G = WAVE / LPE;
    const int cl = lane & (LPE - 1), eg = lane / LPE;
    const long long rows = (long long)B * N;
    for (long long r = (long long)blockIdx.x * NWAVE + wave_id(); r < rows; r += (long long)gridDim.x * NWAVE) {
        const int b = (int)(r / N);
        const int p = (int)(r - (long long)b * N);
        int beg = inv_idx[(size_t)b * inv_idx_len + p];
        int end = inv_idx[(size_t)b * inv_idx_len + p + 1];
        beg = max(0, min(beg, inv_len));
        end = max(beg, min(end, inv_len));
        const int32_t* ln = inv_n + (size_t)b * inv_len;
        const uint8_t* lk = inv_k + (size_t)b * inv_len;
        float4 acc[NP];
#pragma unroll
        for (int u = 0; u < NP; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int base = beg; base < end; base += WAVE) {
            const int j = base + lane;
            int erow = -1;
            float ew = 0.f;
            if (j < end) {
                const int n = ln[j];
                const int k = lk[j];
                if (n >= 0 && n < Nout && k < K) {
                    erow = b * Nout + n;
                    ew = w[(size_t)erow * K + k];
                }
            }
            const int cnt = min(WAVE, end - base);
            for (int e0 = 0; e0 < cnt; e0 += 4 * G) {    // uniform trip count: every lane reaches the shuffles
                int rr[4];
                float ww[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {                // four rows in flight per lane group
                    const int e = e0 + t * G + eg;
                    rr[t] = __shfl(erow, e & (WAVE - 1), WAVE);
                    ww[t] = __shfl(ew, e & (WAVE - 1), WAVE);
                    if (e >= cnt) rr[t] = -1;
                }
#pragma unroll
                for (int u = 0; u < NP; ++u) {
                    const int q = u * WAVE + cl;
                    float4 v[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        v[t] = (q < QI && rr[t] >= 0) ? ldq<AL>(dout + (size_t)rr[t] * J + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[u] = fma4(ww[t], v[t], acc[u]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            for (int off = LPE; off < WAVE; off <<= 1) {
                acc[u].x += __shfl_xor(acc[u].x, off, WAVE); acc[u].y += __shfl_xor(acc[u].y, off, WAVE);
                acc[u].z += __shfl_xor(acc[u].z, off, WAVE); acc[u].w += __shfl_xor(acc[u].w, off, WAVE);
            }
            const int q = u * WAVE + cl;
            if (eg == 0 && q < QI) stq<AL>(gx + (size_t)r * Ci + q * 4, acc[u]);
        }
    }
}

// ---- unguided backward with few channels: one THREAD per edge (engine 3, opt-in) --------------------------------
// The level-0 PointConv of the 10cm / 5cm models (Ci 6, Ca 12, C_mid 16, K 16, 144k points) runs the generic LDS kernel at
// 14 % of the HBM roof (716 us): three block-wide phases over a 22 KB tile for 18 channels of work.
// Per edge the work is two short loops over the CT = Ci + Ca channels against the point's gout tile [CT][CM]:
//     dT[c]   = sum_m gout[c][m] * w[e][m]      -> scatter / contribution row (c < Ci), grad_add (c >= Ci)
//     gw[m]  += gout[c][m] * T[e][c]            T = gathered x row | appended features
// so a lane can own an edge outright: its w row and gw accumulator stay in registers, the gout rows are 16-byte loads that
// the 16 lanes of a point share (one cache line, broadcast), rows of w / add / grad_w / grad_add are lane-contiguous.  No
// LDS, no barriers, no cross-lane traffic -- which also makes the body plain C++: agg_bwd_edge_item is __host__ __device__
// and pcf_hip_pconv_backward_edge_host runs it on host memory, so the arithmetic is checked against the oracle on a CPU
// (tests/test_abi_cpu.py).  Same fmaf order as agg_bwd_kernel: results are bit-identical to it (float atomics aside).
// Not measured yet (written in a round without GPU access): selected only by pcf_hip_set_aggregate_engine(3).
__host__ __device__ __forceinline__ void edge_atomic_add(float* p, float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicAdd(p, v);
#else
    *p += v;
#endif
}

template <int CM, bool AL>
__host__ __device__ __forceinline__ void edge_load_row(const float* p, float (&g)[CM]) {
    if (AL) {
#pragma unroll
        for (int q = 0; q < CM / 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(p + 4 * q);
            g[4 * q] = v.x; g[4 * q + 1] = v.y; g[4 * q + 2] = v.z; g[4 * q + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int m = 0; m < CM; ++m) g[m] = p[m];
    }
}

template <int CM, bool AL>
__host__ __device__ __forceinline__ void agg_bwd_edge_item(const AggArgs& a, long long e, bool atomic_scatter) {
    const int K = a.K, Ci = a.Ci, Ca = a.Ca, CT = Ci + Ca;
    const long long n = e / K;
    const int b = (int)(n / a.Nout);
    const int64_t j = a.idx[e];
    const long long row = (j >= 0 && j < a.N) ? (long long)b * a.N + j : -1;
    float wv[CM], gw[CM], g[CM];
    edge_load_row<CM, AL>(a.w + (size_t)e * CM, wv);
#pragma unroll
    for (int m = 0; m < CM; ++m) gw[m] = 0.f;
    const float* go = a.gout + (size_t)n * CT * CM;
    for (int c = 0; c < Ci; ++c) {                                   // gathered channels
        const float t = row >= 0 ? a.x[(size_t)row * Ci + c] : 0.f;
        edge_load_row<CM, AL>(go + (size_t)c * CM, g);
        float dT = 0.f;
#pragma unroll
        for (int m = 0; m < CM; ++m) { dT = fmaf(g[m], wv[m], dT); gw[m] = fmaf(t, g[m], gw[m]); }
        if (atomic_scatter) { if (row >= 0) edge_atomic_add(a.gx + (size_t)row * Ci + c, dT); }
        else a.contrib[(size_t)e * Ci + c] = dT;
    }
    const float* ar = a.add + (size_t)e * Ca;                        // appended channels (never dereferenced when Ca = 0)
    float* gar = a.gadd + (size_t)e * Ca;
    int ca = 0;
    if (AL && (Ca & 3) == 0) {
        for (; ca < Ca; ca += 4) {
            const float4 t4 = *reinterpret_cast<const float4*>(ar + ca);
            const float tt[4] = {t4.x, t4.y, t4.z, t4.w};
            float d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                edge_load_row<CM, AL>(go + (size_t)(Ci + ca + u) * CM, g);
                float dT = 0.f;
#pragma unroll
                for (int m = 0; m < CM; ++m) { dT = fmaf(g[m], wv[m], dT); gw[m] = fmaf(tt[u], g[m], gw[m]); }
                d[u] = dT;
            }
            *reinterpret_cast<float4*>(gar + ca) = make_float4(d[0], d[1], d[2], d[3]);
        }
    }
    for (; ca < Ca; ++ca) {
        const float t = ar[ca];
        edge_load_row<CM, AL>(go + (size_t)(Ci + ca) * CM, g);
        float dT = 0.f;
#pragma unroll
        for (int m = 0; m < CM; ++m) { dT = fmaf(g[m], wv[m], dT); gw[m] = fmaf(t, g[m], gw[m]); }
        gar[ca] = dT;
    }
    float* gwr = a.gw + (size_t)e * CM;
    if (AL) {
#pragma unroll
        for (int q = 0; q < CM / 4; ++q)
            *reinterpret_cast<float4*>(gwr + 4 * q) = make_float4(gw[4 * q], gw[4 * q + 1], gw[4 * q + 2], gw[4 * q + 3]);
    } else {
#pragma unroll
        for (int m = 0; m < CM; ++m) gwr[m] = gw[m];
    }
}

template <int CM, bool AL, bool ATOMIC>
__global__ __launch_bounds__(BLOCK) void agg_bwd_edge_kernel(const AggArgs a) {
    const long long E = (long long)a.total * a.K;
    for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < E; e += (long long)gridDim.x * BLOCK)
        agg_bwd_edge_item<CM, AL>(a, e, ATOMIC);
}

// ---- host side ----------------------------------------------------------------------------------
static int pow2_ceil(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

struct Plan {
    int P, TS, GS, offG, offT, offD, offW, offO;
    size_t lds_bytes;
    bool vrow;
    bool fixed_shape;      // K = 16, Ci = 16, Ca = 0, H = 8, Cm = 16, guided, 16-byte rows
    bool mfma_shape;       // K = 16, H = 8, guided, Ca = 0, Ci % 16 == 0, Cm in {4, 16}, 16-byte rows: the matrix-core kernels
    bool pconv_mfma_shape; // K = 16, unguided, Ci % 16 == 0, Ca % 16 == 0, Ci + Ca >= 16, Cm in {4, 16}, 16-byte rows
    int cm_t;   // template Cm (0 = run-time)
    bool al;    // every pointer of the call is 16-byte aligned
};

static int make_plan(Plan& pl, bool backward, bool guided, int total, int K, int Ci, int Ca, int Cm, int H,
                     bool ptrs_aligned) {
    const int CT = Ci + Ca;
    pl.vrow = ptrs_aligned && (Ci % 4 == 0) && (Ca % 4 == 0);
    pl.al = ptrs_aligned;
    const bool cm_vec = (Cm % 4 == 0);
    pl.cm_t = 0;
    if (Cm == 1) pl.cm_t = 1;
    else if (pl.vrow && cm_vec && (Cm == 4 || Cm == 8 || Cm == 16 || Cm == 32)) pl.cm_t = Cm;
    // row stride of the gathered tile: 16-byte rows with an odd number of quads (bank spread for the
    // transposed read in grad_w), or an odd float count on the scalar path.
    if (pl.vrow) {
        int q = CT / 4;
        if ((q & 1) == 0) q += 1;
        pl.TS = backward ? q * 4 : CT;
    } else {
        pl.TS = backward ? (CT | 1) : CT;
    }
    pl.GS = (pl.cm_t >= 4) ? Cm + 4 : Cm;
    pl.fixed_shape = guided && pl.vrow && pl.cm_t == 16 && K == 16 && Ci == 16 && Ca == 0 && H == 8 &&
                     pl.TS == (backward ? 20 : 16) && pl.GS == 20;
    pl.mfma_shape = guided && pl.vrow && K == 16 && H == 8 && Ca == 0 && Ci >= 16 && Ci % 16 == 0 && (Cm == 16 || Cm == 4);
    pl.pconv_mfma_shape = !guided && pl.vrow && K == 16 && Ci % 16 == 0 && Ca % 16 == 0 && Ci + Ca >= 16 && (Cm == 16 || Cm == 4);
    auto r4 = [](size_t v) { return (v + 3) / 4 * 4; };
    const size_t per_pt_idx = K;
    const size_t per_pt_g = guided ? (size_t)K * H : 0;
    const size_t per_pt_t = (size_t)K * pl.TS;
    const size_t per_pt_d = (backward && guided) ? (size_t)K * pl.TS : 0;
    const size_t per_pt_w = (size_t)K * Cm;
    const size_t per_pt_o = backward ? (size_t)CT * pl.GS : 0;
    const size_t per_pt = per_pt_idx + per_pt_g + per_pt_t + per_pt_d + per_pt_w + per_pt_o;
    const size_t slack = 6 * 4;   // rounding of each region to 4 floats
    if ((per_pt + slack) * 4 > (size_t)LDS_MAX)
        return fail(PCF_E_UNSUPPORTED, "aggregate: one point needs %zu B of LDS (K=%d Ci=%d Ca=%d Cm=%d), limit %d",
                    (per_pt + slack) * 4, K, Ci, Ca, Cm, LDS_MAX);
    // Points per workgroup.  The backward kernel has three block-wide phases with the HBM latency of its staging
    // exposed in each, so it wants resident workgroups more than amortisation: 4 points (one per wave, ~22 KB of
    // LDS, 7 workgroups per CU) measured 122 us against 154 us with 8 points at N = 80k; the forward (one staging
    // phase) is fastest with 8.
    const int pmax = (backward && guided) ? NWAVE : 8;       // the unguided (PConv, C_mid = 1) backward measured faster with 8
    int P = (int)((LDS_BUDGET / 4 - slack) / per_pt);
    if (P < 1) P = 1;
    if (P > pmax) P = pmax;
    while (P > NWAVE && total / P < 1024) --P;
    pl.P = P;
    size_t off = r4((size_t)P * per_pt_idx);
    pl.offG = (int)off; off = r4(off + (size_t)P * per_pt_g);
    pl.offT = (int)off; off = r4(off + (size_t)P * per_pt_t);
    pl.offD = (int)off; off = r4(off + (size_t)P * per_pt_d);
    pl.offW = (int)off; off = r4(off + (size_t)P * per_pt_w);
    pl.offO = (int)off; off = r4(off + (size_t)P * per_pt_o);
    pl.lds_bytes = off * 4;
    return PCF_OK;
}

template <typename KernelT>
static int launch(KernelT kernel, const AggArgs& a, const Plan& pl, hipStream_t stream, const char* what) {
    if (pl.lds_bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds_bytes);
        if (e != hipSuccess) return fail(PCF_E_LAUNCH, "%s: hipFuncSetAttribute(%zu B LDS): %s", what, pl.lds_bytes,
                                         hipGetErrorString(e));
    }
    const int grid = ceil_div(a.total, pl.P);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(BLOCK), pl.lds_bytes, stream, a);
    return check_launch(what);
}

// Which kernel family serves the shapes the matrix-core kernels cover: 0 = default dispatch, 1 = the LDS kernels (the
// cross-check of the matrix-core ones), 2 = the tiled matrix-core kernels also where a loop-free instance exists
// and the unguided C_mid = 4 forward.  Process-wide, set through pcf_hip_set_aggregate_engine(); the environment
// (PCF_AGG_LDS=1 / PCF_AGG_TILED=1) only provides the initial value.
static std::atomic<int> g_agg_engine{-1};
static int agg_engine() {
    int v = g_agg_engine.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* l = getenv("PCF_AGG_LDS");
        const char* t = getenv("PCF_AGG_TILED");
        v = (l && l[0] == '1') ? 1 : (t && t[0] == '1') ? 2 : 0;
        g_agg_engine.store(v, std::memory_order_relaxed);
    }
    return v;
}
static bool agg_lds_only() { return agg_engine() == 1; }
static bool agg_tiled_only() { return agg_engine() == 2; }
static bool agg_edge_engine() { return agg_engine() == 3; }

static int launch_fwd(const AggArgs& a, const Plan& pl, hipStream_t s) {
#define PCF_FWD(CMV)                                                                   \
    return pl.vrow ? launch(agg_fwd_kernel<CMV, true>, a, pl, s, "agg_fwd_kernel<" #CMV ",true>")  \
                   : launch(agg_fwd_kernel<CMV, false>, a, pl, s, "agg_fwd_kernel<" #CMV ",false>")
    if (pl.mfma_shape && !agg_lds_only()) {
        const int grid = (int)std::min<long long>(ceil_div(a.total, NWAVE), 256 * 64);
        if (a.Cm == 16 && a.Ci == 16 && !agg_tiled_only()) {
            hipLaunchKernelGGL(agg_fwd_fx_mfma_kernel, dim3(grid), dim3(BLOCK), 0, s, a);
            return check_launch("agg_fwd_fx_mfma_kernel");
        }
        if (a.Cm == 16) hipLaunchKernelGGL((agg_fwd_mfma_kernel<16, 0>), dim3(grid), dim3(BLOCK), 0, s, a);
        else hipLaunchKernelGGL((agg_fwd_mfma_kernel<4, 0>), dim3(grid), dim3(BLOCK), 0, s, a);
        return check_launch(a.Cm == 16 ? "agg_fwd_mfma_kernel<16,0>" : "agg_fwd_mfma_kernel<4,0>");
    }
    // unguided forward: C_mid = 16 only (at C_mid = 4 the 64-byte output rows make the LDS kernel's coalesced stores
    // the better deal: 69 vs 82 us at 144k points, Ci = Ca = 16); PCF_AGG_TILED=1 forces it (tests)
    if (pl.pconv_mfma_shape && !agg_lds_only() && (a.Cm == 16 || agg_tiled_only())) {
        const int grid = (int)std::min<long long>(ceil_div(a.total, NWAVE), 256 * 64);
        if (a.Cm == 16) hipLaunchKernelGGL(pconv_fwd_mfma_kernel<16>, dim3(grid), dim3(BLOCK), 0, s, a);
        else hipLaunchKernelGGL(pconv_fwd_mfma_kernel<4>, dim3(grid), dim3(BLOCK), 0, s, a);
        return check_launch(a.Cm == 16 ? "pconv_fwd_mfma_kernel<16>" : "pconv_fwd_mfma_kernel<4>");
    }
    if (pl.fixed_shape) return launch(agg_fwd_kernel<16, true, true>, a, pl, s, "agg_fwd_kernel<16,true,true>");
    switch (pl.cm_t) {
        case 1: PCF_FWD(1);
        case 4: return launch(agg_fwd_kernel<4, true>, a, pl, s, "agg_fwd_kernel<4,true>");
        case 8: return launch(agg_fwd_kernel<8, true>, a, pl, s, "agg_fwd_kernel<8,true>");
        case 16: return launch(agg_fwd_kernel<16, true>, a, pl, s, "agg_fwd_kernel<16,true>");
        case 32: return launch(agg_fwd_kernel<32, true>, a, pl, s, "agg_fwd_kernel<32,true>");
        default: PCF_FWD(0);
    }
#undef PCF_FWD
}

// thread-per-edge backward (engine 3): unguided, C_mid 4 or 16, at most 64 channels per edge
static bool edge_covers(const AggArgs& a) { return !a.guid && (a.Cm == 16 || a.Cm == 4) && a.Ci + a.Ca <= 64 && a.Ci + a.Ca >= 1; }

template <bool ATOMIC>
static int launch_bwd_edge(const AggArgs& a, const Plan& pl, hipStream_t s) {
    const long long E = (long long)a.total * a.K;
    const int grid = (int)std::max<long long>(1, std::min<long long>((E + BLOCK - 1) / BLOCK, 256 * 64));
    const dim3 gd(grid), bd(BLOCK);
    if (a.Cm == 16) {
        if (pl.al) hipLaunchKernelGGL((agg_bwd_edge_kernel<16, true, ATOMIC>), gd, bd, 0, s, a);
        else hipLaunchKernelGGL((agg_bwd_edge_kernel<16, false, ATOMIC>), gd, bd, 0, s, a);
    } else {
        if (pl.al) hipLaunchKernelGGL((agg_bwd_edge_kernel<4, true, ATOMIC>), gd, bd, 0, s, a);
        else hipLaunchKernelGGL((agg_bwd_edge_kernel<4, false, ATOMIC>), gd, bd, 0, s, a);
    }
    return check_launch(a.Cm == 16 ? "agg_bwd_edge_kernel<16>" : "agg_bwd_edge_kernel<4>");
}

template <bool ATOMIC>
static int launch_bwd_mode(const AggArgs& a, const Plan& pl, hipStream_t s) {
    if (agg_edge_engine() && edge_covers(a)) return launch_bwd_edge<ATOMIC>(a, pl, s);
#define PCF_BWD(CMV)                                                                            \
    return pl.vrow ? launch(agg_bwd_kernel<CMV, true, ATOMIC>, a, pl, s, "agg_bwd_kernel<" #CMV ",true>")  \
                   : launch(agg_bwd_kernel<CMV, false, ATOMIC>, a, pl, s, "agg_bwd_kernel<" #CMV ",false>")
    if (pl.mfma_shape && ATOMIC && !agg_lds_only()) {
        const int grid = (int)std::min<long long>(ceil_div(a.total, NWAVE), 256 * 64);
        if (a.Cm == 16 && a.Ci == 16 && !agg_tiled_only()) {
            hipLaunchKernelGGL(agg_bwd_fx_mfma_kernel, dim3(grid), dim3(BLOCK), 0, s, a);
            return check_launch("agg_bwd_fx_mfma_kernel");
        }
        if (a.Cm == 16) hipLaunchKernelGGL((agg_bwd_mfma_kernel<16, 0>), dim3(grid), dim3(BLOCK), 0, s, a);
        else hipLaunchKernelGGL((agg_bwd_mfma_kernel<4, 0>), dim3(grid), dim3(BLOCK), 0, s, a);
        return check_launch(a.Cm == 16 ? "agg_bwd_mfma_kernel<16,0>" : "agg_bwd_mfma_kernel<4,0>");
    }
    if (pl.pconv_mfma_shape && !agg_lds_only() && (ATOMIC || a.contrib || a.Ci == 0)) {
        const int grid = (int)std::min<long long>(ceil_div(a.total, NWAVE), 256 * 64);
        if (a.Cm == 16) hipLaunchKernelGGL((pconv_bwd_mfma_kernel<16, ATOMIC>), dim3(grid), dim3(BLOCK), 0, s, a);
        else hipLaunchKernelGGL((pconv_bwd_mfma_kernel<4, ATOMIC>), dim3(grid), dim3(BLOCK), 0, s, a);
        return check_launch(a.Cm == 16 ? "pconv_bwd_mfma_kernel<16>" : "pconv_bwd_mfma_kernel<4>");
    }
    if (pl.fixed_shape) return launch(agg_bwd_kernel<16, true, ATOMIC, true>, a, pl, s, "agg_bwd_kernel<16,true,fx>");
    switch (pl.cm_t) {
        case 1: PCF_BWD(1);
        case 4: return launch(agg_bwd_kernel<4, true, ATOMIC>, a, pl, s, "agg_bwd_kernel<4,true>");
        case 8: return launch(agg_bwd_kernel<8, true, ATOMIC>, a, pl, s, "agg_bwd_kernel<8,true>");
        case 16: return launch(agg_bwd_kernel<16, true, ATOMIC>, a, pl, s, "agg_bwd_kernel<16,true>");
        case 32: return launch(agg_bwd_kernel<32, true, ATOMIC>, a, pl, s, "agg_bwd_kernel<32,true>");
        default: PCF_BWD(0);
    }
#undef PCF_BWD
}

// C_mid = 1 without guidance, rows of whole 16-byte quads, K one-per-lane: the LDS-free kernels.
bool agg1_covers(bool guided, int K, int Ci, int Ca, int Cm) {
    return !guided && Cm == 1 && K <= WAVE && Ci % 4 == 0 && Ca % 4 == 0 && Ci + Ca <= 2 * 4 * WAVE;
}
static int agg1_grid(int total) { return (int)std::min<long long>(ceil_div(total, NWAVE), 256 * 64); }

static int check_dims(const char* who, int B, int N, int Nout, int K, int Ci, int Ca, int Cm, int H, bool guided) {
    PCF_REQUIRE(B >= 0 && N >= 0 && Nout >= 0, "%s: negative batch/point count (B=%d N=%d Nout=%d)", who, B, N, Nout);
    PCF_REQUIRE(K >= 1 && Ci >= 0 && Ca >= 0 && Ci + Ca >= 1 && Cm >= 1, "%s: bad K/Ci/Ca/Cm (%d/%d/%d/%d)", who, K, Ci,
                Ca, Cm);
    PCF_REQUIRE(!guided || H >= 1, "%s: num_heads must be >= 1 (got %d)", who, H);
    PCF_REQUIRE((long long)B * N < (1ll << 31) && (long long)B * Nout * K < (1ll << 31),
                "%s: B*N or B*Nout*K does not fit 31 bits", who);
    return PCF_OK;
}

// Forward of both operators.  guid == null -> PConv; add == null / Ca == 0 -> no appended features.
int aggregate_forward(const float* x, const int64_t* idx, const float* guid, const float* w, const float* add,
                      float* out, int B, int N, int Nout, int K, int Ci, int Ca, int Cm, int H, hipStream_t stream) {
    if (int e = check_dims("aggregate forward", B, N, Nout, K, Ci, Ca, Cm, H, guid != nullptr)) return e;
    const int total = B * Nout;
    if (total == 0) return ok();
    PCF_REQUIRE(x || N == 0 || Ci == 0, "aggregate forward: x is null");
    PCF_REQUIRE(idx && w && out, "aggregate forward: null pointer (idx=%p w=%p out=%p)", (const void*)idx,
                (const void*)w, (void*)out);
    PCF_REQUIRE(Ca == 0 || add, "aggregate forward: Ca=%d but additional features pointer is null", Ca);
    const bool al = aligned16(x) && aligned16(w) && aligned16(out) && (Ca == 0 || aligned16(add)) && (!guid || aligned16(guid));
    AggArgs a{};
    a.x = x; a.idx = idx; a.guid = guid; a.w = w; a.add = add; a.out = out;
    a.total = total; a.N = N; a.Nout = Nout; a.K = K; a.Ci = Ci; a.Ca = Ca; a.Cm = Cm; a.H = guid ? H : 1;
    if (agg1_covers(guid != nullptr, K, Ci, Ca, Cm)) {
        if (al) hipLaunchKernelGGL(agg1_fwd_kernel<true>, dim3(agg1_grid(total)), dim3(BLOCK), 0, stream, a);
        else hipLaunchKernelGGL(agg1_fwd_kernel<false>, dim3(agg1_grid(total)), dim3(BLOCK), 0, stream, a);
        return check_launch("agg1_fwd_kernel");
    }
    Plan pl;
    if (int e = make_plan(pl, false, guid != nullptr, total, K, Ci, Ca, Cm, H, al)) return e;
    a.P = pl.P; a.TS = pl.TS; a.GS = pl.GS;
    a.offG = pl.offG; a.offT = pl.offT; a.offD = pl.offD; a.offW = pl.offW; a.offO = pl.offO;
    return launch_fwd(a, pl, stream);
}

template <bool AL, bool ATOMIC>
static int launch_agg1_bwd(const AggArgs& a, hipStream_t s) {
    const dim3 grid(agg1_grid(a.total));
    if (a.Ci + a.Ca <= 4 * WAVE) hipLaunchKernelGGL((agg1_bwd_kernel<AL, ATOMIC, 1>), grid, dim3(BLOCK), 0, s, a);
    else hipLaunchKernelGGL((agg1_bwd_kernel<AL, ATOMIC, 2>), grid, dim3(BLOCK), 0, s, a);
    return check_launch("agg1_bwd_kernel");
}

// Backward of both operators.  Exactly one of gx (atomic scatter; zeroed here) / contrib (per-edge
// rows for csr_reduce) is non-null -- or neither where agg1_covers() holds and the caller takes grad_x from
// csr_gather1().
int aggregate_backward(const float* gout, const float* x, const int64_t* idx, const float* guid, const float* w,
                       const float* add, float* gx, float* contrib, float* gguid, float* gw, float* gadd, int B,
                       int N, int Nout, int K, int Ci, int Ca, int Cm, int H, hipStream_t stream) {
    if (int e = check_dims("aggregate backward", B, N, Nout, K, Ci, Ca, Cm, H, guid != nullptr)) return e;
    const int total = B * Nout;
    if (gx && (size_t)B * N * Ci > 0) {
        hipError_t e = zero_async(gx, (size_t)B * N * Ci * sizeof(float), stream);
        if (e != hipSuccess) return fail(PCF_E_LAUNCH, "aggregate backward: memset grad_x: %s", hipGetErrorString(e));
    }
    if (total == 0) return ok();
    PCF_REQUIRE(gout && idx && w && gw, "aggregate backward: null pointer");
    const bool cm1 = agg1_covers(guid != nullptr, K, Ci, Ca, Cm) && !contrib;
    PCF_REQUIRE((gx != nullptr) != (contrib != nullptr) || Ci == 0 || (cm1 && !gx),
                "aggregate backward: need exactly one of grad_x / contrib");
    PCF_REQUIRE(!guid || gguid, "aggregate backward: grad_guid is null");
    PCF_REQUIRE(Ca == 0 || (add && gadd), "aggregate backward: Ca=%d but add/grad_add pointer is null", Ca);
    const bool al = aligned16(x) && aligned16(w) && aligned16(gout) && aligned16(gw) && (Ca == 0 || (aligned16(add) && aligned16(gadd))) &&
                    (!guid || (aligned16(guid) && aligned16(gguid))) && (!contrib || aligned16(contrib));
    AggArgs a{};
    a.x = x; a.idx = idx; a.guid = guid; a.w = w; a.add = add; a.gout = gout;
    a.gx = gx; a.contrib = contrib; a.gguid = gguid; a.gw = gw; a.gadd = gadd;
    a.total = total; a.N = N; a.Nout = Nout; a.K = K; a.Ci = Ci; a.Ca = Ca; a.Cm = Cm; a.H = guid ? H : 1;
    if (cm1) {
        const bool al1 = al && (Ca == 0 || aligned16(gadd));
        if (gx) return al1 ? launch_agg1_bwd<true, true>(a, stream) : launch_agg1_bwd<false, true>(a, stream);
        return al1 ? launch_agg1_bwd<true, false>(a, stream) : launch_agg1_bwd<false, false>(a, stream);
    }
    Plan pl;
    if (int e = make_plan(pl, true, guid != nullptr, total, K, Ci, Ca, Cm, H, al)) return e;
    a.P = pl.P; a.TS = pl.TS; a.GS = pl.GS;
    a.offG = pl.offG; a.offT = pl.offT; a.offD = pl.offD; a.offW = pl.offW; a.offO = pl.offO;
    return gx ? launch_bwd_mode<true>(a, pl, stream) : launch_bwd_mode<false>(a, pl, stream);
}

int csr_reduce(const float* contrib, const int32_t* inv_n, const uint8_t* inv_k, const int32_t* inv_idx, float* gx,
               int B, int N, int Nout, int K, int Ci, int inv_len, int inv_idx_len, hipStream_t stream) {
    const long long rows = (long long)B * N;
    if (rows == 0 || Ci == 0) return ok();
    const int grid = (int)std::min<long long>((rows + NWAVE - 1) / NWAVE, 256 * 32);
    hipLaunchKernelGGL(csr_reduce_kernel, dim3(grid), dim3(BLOCK), 0, stream, contrib, inv_n, inv_k, inv_idx, gx, B, N,
                       Nout, K, Ci, inv_len, inv_idx_len);
    return check_launch("csr_reduce_kernel");
}

// grad_x of the C_mid = 1 aggregate straight from its output gradient dout [B*Nout, J] (J = Ci + Ca) and the CSR.
int csr_gather1(const float* dout, const float* w, const int32_t* inv_n, const uint8_t* inv_k, const int32_t* inv_idx,
                float* gx, int B, int N, int Nout, int K, int Ci, int J, int inv_len, int inv_idx_len,
                hipStream_t stream) {
    const long long rows = (long long)B * N;
    if (rows == 0 || Ci == 0) return ok();
    const dim3 grid((unsigned)std::min<long long>((rows + NWAVE - 1) / NWAVE, 256 * 64));
    const bool al = aligned16(dout) && aligned16(gx) && J % 4 == 0;
#define PCF_G1(ALV, NPV)                                                                                            \
    hipLaunchKernelGGL((csr_gather1_kernel<ALV, NPV>), grid, dim3(BLOCK), 0, stream, dout, w, inv_n, inv_k, inv_idx, gx, \
                       B, N, Nout, K, Ci, J, inv_len, inv_idx_len)
    if (Ci <= 4 * WAVE) { if (al) PCF_G1(true, 1); else PCF_G1(false, 1); }
    else { if (al) PCF_G1(true, 2); else PCF_G1(false, 2); }
#undef PCF_G1
    return check_launch("csr_gather1_kernel");
}

}  // namespace pcf

// ---- C ABI ----------------------------------------------------------------------------------------
extern "C" {

int pcf_hip_pcf_forward(const float* x, const int64_t* idx, const float* guid, const float* w, float* out, int B,
                        int N, int Nout, int K, int Ci, int Cm, int H, void* stream) {
    if (!guid && B * Nout > 0) return pcf::fail(PCF_E_BADARG, "pcf_forward: guidance is null");
    return pcf::aggregate_forward(x, idx, guid, w, nullptr, out, B, N, Nout, K, Ci, 0, Cm, H, (hipStream_t)stream);
}

int pcf_hip_pcf_backward(const float* grad_out, const float* x, const int64_t* idx, const float* guid, const float* w,
                         float* grad_x, float* grad_guid, float* grad_w, int B, int N, int Nout, int K, int Ci, int Cm,
                         int H, void* stream) {
    if (!guid && B * Nout > 0) return pcf::fail(PCF_E_BADARG, "pcf_backward: guidance is null");
    if (!grad_x && (long long)B * N * Ci > 0) return pcf::fail(PCF_E_BADARG, "pcf_backward: grad_x is null");
    return pcf::aggregate_backward(grad_out, x, idx, guid, w, nullptr, grad_x, nullptr, grad_guid, grad_w, nullptr, B, N,
                                   Nout, K, Ci, 0, Cm, H, (hipStream_t)stream);
}

int pcf_hip_pconv_forward(const float* x, const int64_t* idx, const float* w, const float* add, float* out, int B,
                          int N, int Nout, int K, int Ci, int Ca, int Cm, void* stream) {
    return pcf::aggregate_forward(x, idx, nullptr, w, add, out, B, N, Nout, K, Ci, Ca, Cm, 1, (hipStream_t)stream);
}

int pcf_hip_pconv_backward(const float* grad_out, const float* x, const int64_t* idx, const float* w, const float* add,
                           float* grad_x, float* grad_w, float* grad_add, int B, int N, int Nout, int K, int Ci, int Ca,
                           int Cm, void* stream) {
    if (!grad_x && (long long)B * N * Ci > 0) return pcf::fail(PCF_E_BADARG, "pconv_backward: grad_x is null");
    return pcf::aggregate_backward(grad_out, x, idx, nullptr, w, add, grad_x, nullptr, nullptr, grad_w, grad_add, B, N,
                                   Nout, K, Ci, Ca, Cm, 1, (hipStream_t)stream);
}

size_t pcf_hip_pcf_backward_csr_workspace_bytes(int B, int Nout, int K, int Ci) {
    if (B < 0 || Nout < 0 || K < 0 || Ci < 0) return 0;
    return (size_t)B * Nout * K * Ci * sizeof(float) + 256;
}

int pcf_hip_pcf_backward_csr(const float* grad_out, const float* x, const int32_t* inv_neighbors, const uint8_t* inv_k,
                             const int32_t* inv_idx, const int64_t* idx, const float* guid, const float* w,
                             float* grad_x, float* grad_guid, float* grad_w, void* workspace, size_t workspace_bytes,
                             int B, int N, int Nout, int K, int Ci, int Cm, int H, int inv_len, int inv_idx_len,
                             void* stream) {
    using namespace pcf;
    if (!guid && B * Nout > 0) return fail(PCF_E_BADARG, "pcf_backward_csr: guidance is null");
    PCF_REQUIRE(inv_idx_len >= N + 1, "pcf_backward_csr: inverse_neighbor_idx size must be >= N + 1 (got %d, N=%d)",
                inv_idx_len, N);
    PCF_REQUIRE(inv_idx && (inv_len == 0 || (inv_neighbors && inv_k)), "pcf_backward_csr: null inverse index");
    PCF_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= pcf_hip_pcf_backward_csr_workspace_bytes(B, Nout, K, Ci),
                "pcf_backward_csr: workspace too small or misaligned");
    PCF_REQUIRE(grad_x || (long long)B * N * Ci == 0, "pcf_backward_csr: grad_x is null");
    hipStream_t s = (hipStream_t)stream;
    float* contrib = static_cast<float*>(workspace);
    if (B * Nout == 0) {
        if ((size_t)B * N * Ci) (void)zero_async(grad_x, (size_t)B * N * Ci * 4, s);
        return ok();
    }
    if (int e = aggregate_backward(grad_out, x, idx, guid, w, nullptr, nullptr, contrib, grad_guid, grad_w, nullptr, B, N,
                                   Nout, K, Ci, 0, Cm, H, s))
        return e;
    return csr_reduce(contrib, inv_neighbors, inv_k, inv_idx, grad_x, B, N, Nout, K, Ci, inv_len, inv_idx_len, s);
}

int pcf_hip_set_aggregate_engine(int engine) {
    if (engine < 0 || engine > 3)
        return pcf::fail(PCF_E_BADARG, "set_aggregate_engine: 0 (default), 1 (LDS), 2 (tiled) or 3 (thread-per-edge unguided backward), got %d", engine);
    pcf::g_agg_engine.store(engine, std::memory_order_relaxed);
    return pcf::ok();
}

int pcf_hip_get_aggregate_engine(void) { return pcf::agg_engine(); }

// Host emulation of agg_bwd_edge_kernel: the same __host__ __device__ item function over every edge, on HOST memory
// (test hook, no GPU).  grad_x accumulates (the caller zeroes it) when `atomic` != 0, else per-edge rows go to `contrib`.
int pcf_hip_pconv_backward_edge_host(const float* grad_out, const float* x, const int64_t* idx, const float* w, const float* add,
                                     float* grad_x, float* contrib, float* grad_w, float* grad_add, int B, int N, int Nout, int K,
                                     int Ci, int Ca, int Cm, int atomic) {
    using namespace pcf;
    PCF_REQUIRE(Cm == 16 || Cm == 4, "pconv_backward_edge_host: C_mid must be 4 or 16 (got %d)", Cm);
    PCF_REQUIRE(B >= 0 && N >= 0 && Nout >= 0 && K >= 1 && Ci >= 0 && Ca >= 0 && Ci + Ca >= 1 && Ci + Ca <= 64, "pconv_backward_edge_host: bad sizes");
    PCF_REQUIRE(grad_out && idx && w && grad_w && (Ci == 0 || (x && (atomic ? grad_x != nullptr : contrib != nullptr))) && (Ca == 0 || (add && grad_add)),
                "pconv_backward_edge_host: null pointer");
    AggArgs a{};
    a.x = x; a.idx = idx; a.w = w; a.add = add; a.gout = grad_out; a.gx = grad_x; a.contrib = contrib; a.gw = grad_w; a.gadd = grad_add;
    a.total = B * Nout; a.N = N; a.Nout = Nout; a.K = K; a.Ci = Ci; a.Ca = Ca; a.Cm = Cm; a.H = 1;
    const bool al = aligned16(x) && aligned16(w) && aligned16(grad_out) && aligned16(grad_w) && (Ca == 0 || (aligned16(add) && aligned16(grad_add)));
    const long long E = (long long)a.total * K;
    for (long long e = 0; e < E; ++e) {
        if (Cm == 16) { if (al) agg_bwd_edge_item<16, true>(a, e, atomic != 0); else agg_bwd_edge_item<16, false>(a, e, atomic != 0); }
        else { if (al) agg_bwd_edge_item<4, true>(a, e, atomic != 0); else agg_bwd_edge_item<4, false>(a, e, atomic != 0); }
    }
    return ok();
}

}  // extern "C"
