// Per-edge MLP layers of PointConv / PointConvFormer on gfx950:  y = act( BN( x . W^T + b ) )
// over R = B*M*K rows with tiny channel counts (Cin, Cout <= 64), forward and backward, with
// batch-statistics BatchNorm (training) or running statistics (inference) or no BN at all.
//
// Replaces, for the [B,M,K,C] tensors, Linear_BN.forward (layer_utils.py:272-277: nn.Linear, a
// permute to channel-second, CpBatchNorm2d, a permute back) plus the ReLU / sigmoid that follows
// it in WeightNet (layers.py:163-171), MultiHeadGuidance (layers.py:47-68) and PCFLayer.mlp_conv
// (layers.py:361-362), and the autograd of all of it.  On the reference each such layer is a
// GEMM with an 8..64-wide inner dimension over 1.28 M rows followed by 3-4 BatchNorm kernels; the
// weight gradient is a [Cout x R] . [R x Cin] reduction that library GEMMs handle very badly
// (0.77-1.5 ms each on MI355X for R = 1.28 M).  Here a layer is
//   forward : one pass for the BN statistics (skipped in inference) + one pass that writes y;
//   backward: one pass for the two BN reductions + one pass that writes dx and accumulates dW,
// every pass reading x (and dy) exactly once and recomputing z = x.W^T + b in registers instead of
// storing it.  One lane owns one row: x sits in VGPRs, W is read from LDS as wave-wide broadcasts.
// The only real contraction, dW = dz^T . x over all rows, runs on the matrix cores
// (v_mfma_f32_16x16x4_f32, exact fp32) from LDS-staged 64-row tiles.
#include <algorithm>

#include "pcf_common.h"
#include "edge_mlp.h"

namespace pcf {

constexpr int OC = 16;              // output channels handled per LDS round
constexpr int MAXCH = 4;            // Cout <= 64
constexpr int ZS = OC + 1;          // row stride of the per-wave [64 x 16] scratch tile (odd: conflict-free)

__device__ __forceinline__ float act_fwd(int ACT, float u) {
    if (ACT == ACT_RELU) return fmaxf(u, 0.f);
    if (ACT == ACT_LEAKY) return u > 0.f ? u : 0.1f * u;
    if (ACT == ACT_SIGMOID) return 1.f / (1.f + __expf(-u));
    return u;
}
__device__ __forceinline__ float act_bwd(int ACT, float u) {       // d act / d u
    if (ACT == ACT_RELU) return u > 0.f ? 1.f : 0.f;
    if (ACT == ACT_LEAKY) return u > 0.f ? 1.f : 0.1f;
    if (ACT == ACT_SIGMOID) { const float s = 1.f / (1.f + __expf(-u)); return s * (1.f - s); }
    return 1.f;
}

// Stage W (zero-padded to CIN columns) and the per-channel vectors in LDS.
template <int CIN>
__device__ __forceinline__ void stage_weights(const RowLin& a, float* sW, float* sV) {
    for (int u = threadIdx.x; u < a.Cout * CIN; u += BLOCK) {
        const int o = u / CIN, i = u - o * CIN;
        sW[u] = i < a.Cin ? a.W[o * a.Cin + i] : 0.f;
    }
    // sV: [0]=b [1]=mean [2]=rstd [3]=gamma [4]=beta [5]=m1 [6]=m2, each 64 wide
    for (int u = threadIdx.x; u < 7 * 64; u += BLOCK) {
        const int v = u >> 6, o = u & 63;
        const float* src = v == 0 ? a.b : v == 1 ? a.mean : v == 2 ? a.rstd : v == 3 ? a.gamma : v == 4 ? a.beta
                         : v == 5 ? a.m1 : a.m2;
        float d = (v == 2 || v == 3) ? 1.f : 0.f;
        sV[u] = (src && o < a.Cout) ? src[o] : d;
    }
}

template <int CIN>
__device__ __forceinline__ void load_row(const RowLin& a, long long row, bool valid, float (&x)[CIN]) {
#pragma unroll
    for (int i = 0; i < CIN; ++i) x[i] = 0.f;
    if (!valid) return;
    const float* p = a.x + (size_t)row * a.Cin;
    if (a.vec_x) {
#pragma unroll
        for (int q = 0; q < CIN / 4; ++q)
            if (q * 4 < a.Cin) {
                const float4 v = ld4(p + q * 4);
                x[q * 4] = v.x; x[q * 4 + 1] = v.y; x[q * 4 + 2] = v.z; x[q * 4 + 3] = v.w;
            }
    } else {
#pragma unroll
        for (int i = 0; i < CIN; ++i)
            if (i < a.Cin) x[i] = p[i];
    }
}

template <int CIN>
__device__ __forceinline__ float dot_row(const float (&x)[CIN], const float* wrow, float acc) {
#pragma unroll
    for (int q = 0; q < CIN / 4; ++q)
        acc = dot4(make_float4(x[q * 4], x[q * 4 + 1], x[q * 4 + 2], x[q * 4 + 3]), ld4(wrow + q * 4), acc);
    return acc;
}


// Gathered additive row (guidance first layer): g[o] = gadd[b*gN + gidx[row]][o], zeros when absent.
__device__ __forceinline__ long long gather_row_index(const RowLin& a, long long row, bool valid) {
    if (!a.gadd || !valid) return -1;
    const int64_t j = a.gidx[row];
    if (j < 0 || j >= a.gN) return -1;
    return (row / a.rows_per_batch) * a.gN + j;
}
__device__ __forceinline__ void load_gadd(const RowLin& a, long long grow, float (&g)[16]) {
#pragma unroll
    for (int o = 0; o < 16; ++o) g[o] = 0.f;
    if (grow < 0) return;
    const float* p = a.gadd + (size_t)grow * a.Cout;
#pragma unroll
    for (int o = 0; o < 16; ++o)
        if (o < a.Cout) g[o] = p[o];
}

// Pre-activation of this lane's row for output o:  x.W[o] (+ gathered term) (- the same quantity of the
// first row of the lane's group) + b[o].  Contains a wave shuffle: every lane of the wave must call it.
template <int CIN>
__device__ __forceinline__ float pre_act(const RowLin& a, const float (&x)[CIN], const float* sW, const float* sV, int o,
                                         const float (&g)[16], int lane) {
    float t = dot_row<CIN>(x, sW + o * CIN, 0.f);
    if (a.gadd) t += g[o & 15];
    if (a.group > 1) t -= __shfl(t, lane & ~(a.group - 1), WAVE);
    return t + sV[o];
}

// Sum, over the 64 rows of a wave tile, of one or two per-row values for 16 channels: lane (o, q)
// adds rows q*16 .. q*16+15 of channel o from the scratch tile(s).
__device__ __forceinline__ void tile_colsum(const float* t1, const float* t2, int lane, float& s1, float& s2) {
    const int o = lane & 15, q = lane >> 4;
    const float* p1 = t1 + (q * 16) * ZS + o;
    const float* p2 = t2 + (q * 16) * ZS + o;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s1 += p1[r * ZS]; s2 += p2[r * ZS]; }
}

// Workgroup-level combine of the per-lane (channel, quarter) accumulators -> part[blockIdx][2][64]
__device__ __forceinline__ void write_block_partials(float (&s1)[MAXCH], float (&s2)[MAXCH], float* red, float* part) {
    const int lane = lane_id(), wave = wave_id();
#pragma unroll
    for (int ch = 0; ch < MAXCH; ++ch) {
        float a = s1[ch], b = s2[ch];
        a += __shfl_xor(a, 16, WAVE); a += __shfl_xor(a, 32, WAVE);
        b += __shfl_xor(b, 16, WAVE); b += __shfl_xor(b, 32, WAVE);
        if (lane < 16) { red[wave * 128 + ch * 16 + lane] = a; red[wave * 128 + 64 + ch * 16 + lane] = b; }
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NWAVE; ++w) t += red[w * 128 + threadIdx.x];
        part[(size_t)blockIdx.x * 128 + threadIdx.x] = t;
    }
}

// ---- K1: sum z, sum z^2 per output channel ------------------------------------------------------
template <int CIN>
__global__ __launch_bounds__(BLOCK) void rowlin_stats_kernel(const RowLin a) {
    extern __shared__ __align__(16) float smem[];
    float* sW = smem;                        // [Cout][CIN]
    float* sV = sW + 64 * CIN;               // 7 x 64
    float* sZ = sV + 7 * 64;                 // per wave: 2 tiles [64][ZS]
    float* red = sZ + NWAVE * 2 * 64 * ZS;   // [4][128]
    stage_weights<CIN>(a, sW, sV);
    __syncthreads();
    const int lane = lane_id(), wave = wave_id();
    float* t1 = sZ + wave * 2 * 64 * ZS;
    float* t2 = t1 + 64 * ZS;
    float s1[MAXCH] = {0.f, 0.f, 0.f, 0.f}, s2[MAXCH] = {0.f, 0.f, 0.f, 0.f};
    const long long ntiles = (a.R + WAVE - 1) / WAVE;
    for (long long t = (long long)blockIdx.x * NWAVE + wave; t < ntiles; t += (long long)gridDim.x * NWAVE) {
        const long long row = t * WAVE + lane;
        const bool valid = row < a.R;
        float x[CIN];
        load_row<CIN>(a, row, valid, x);
        float gv[16];
        load_gadd(a, gather_row_index(a, row, valid), gv);
#pragma unroll
        for (int ch = 0; ch < MAXCH; ++ch) {
            if (ch * OC < a.Cout) {
#pragma unroll
                for (int j = 0; j < OC; ++j) {
                    const int o = ch * OC + j;
                    float z = 0.f;
                    if (o < a.Cout) z = pre_act<CIN>(a, x, sW, sV, o, gv, lane);
                    if (!valid) z = 0.f;
                    t1[lane * ZS + j] = z;
                    t2[lane * ZS + j] = z * z;
                }
                tile_colsum(t1, t2, lane, s1[ch], s2[ch]);
            }
        }
    }
    write_block_partials(s1, s2, red, a.part);
}

// Sum the workgroup partials part[p][which*64 + c] over p for 8 channels x {sum, second sum} per
// workgroup: 16 values x 64 slices of the partial list, two loads in flight per thread, fp64 combine.
// Grid = 8 workgroups of 1024 threads covers the 64 channels.
__device__ __forceinline__ void reduce_partials_8ch(const float* __restrict__ part, int nblocks, double (*sh)[16],
                                                    double& sum0, double& sum1) {
    const int v = threadIdx.x & 15, slice = threadIdx.x >> 4;          // v = which*8 + channel-in-block
    const int col = (v >> 3) * 64 + blockIdx.x * 8 + (v & 7);
    double a0 = 0.0, a1 = 0.0;
    int p = slice;
    for (; p + 64 < nblocks; p += 128) {
        a0 += (double)part[(size_t)p * 128 + col];
        a1 += (double)part[(size_t)(p + 64) * 128 + col];
    }
    for (; p < nblocks; p += 64) a0 += (double)part[(size_t)p * 128 + col];
    sh[slice][v] = a0 + a1;
    __syncthreads();
    sum0 = 0.0; sum1 = 0.0;
    if (threadIdx.x < 8) {
        for (int sl = 0; sl < 64; ++sl) { sum0 += sh[sl][threadIdx.x]; sum1 += sh[sl][8 + threadIdx.x]; }
    }
}

// ---- K2: statistics -> mean, rstd, running-stat update --------------------------------------------
__global__ __launch_bounds__(1024) void bn_stats_finalize_kernel(const float* __restrict__ part, int nblocks, long long R,
                                                                  int Cout, float eps, float momentum,
                                                                  float* __restrict__ running_mean,
                                                                  float* __restrict__ running_var,
                                                                  float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    __shared__ double sh[64][16];
    double s0, s1;
    reduce_partials_8ch(part, nblocks, sh, s0, s1);
    const int t = blockIdx.x * 8 + threadIdx.x;
    if (threadIdx.x < 8 && t < Cout) {
        const double n = (double)R;
        const double mean = s0 / n;
        double var = s1 / n - mean * mean;
        if (var < 0.0) var = 0.0;
        mean_out[t] = (float)mean;
        rstd_out[t] = (float)(1.0 / sqrt(var + (double)eps));
        if (running_mean) {
            const double unbiased = R > 1 ? var * n / (n - 1.0) : var;
            running_mean[t] = (float)((1.0 - momentum) * running_mean[t] + momentum * mean);
            running_var[t] = (float)((1.0 - momentum) * running_var[t] + momentum * unbiased);
        }
    }
}

// ---- K3: y = act(BN(x W^T + b)) -------------------------------------------------------------------
template <int CIN>
__global__ __launch_bounds__(BLOCK) void rowlin_fwd_kernel(const RowLin a) {
    const int ACT = a.act;
    extern __shared__ __align__(16) float smem[];
    float* sW = smem;
    float* sV = sW + 64 * CIN;
    stage_weights<CIN>(a, sW, sV);
    __syncthreads();
    const int lane = lane_id(), wave = wave_id();
    const bool bn = a.mean != nullptr;
    const long long ntiles = (a.R + WAVE - 1) / WAVE;
    for (long long t = (long long)blockIdx.x * NWAVE + wave; t < ntiles; t += (long long)gridDim.x * NWAVE) {
        const long long row = t * WAVE + lane;
        const bool valid = row < a.R;
        float x[CIN];
        load_row<CIN>(a, row, valid, x);
        float gv[16];
        load_gadd(a, gather_row_index(a, row, valid), gv);
        float* yr = a.y + (size_t)row * a.Cout;
        for (int o0 = 0; o0 < a.Cout; o0 += 4) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int o = o0 + j;
                float u = 0.f;
                if (o < a.Cout) {
                    u = pre_act<CIN>(a, x, sW, sV, o, gv, lane);
                    if (bn) u = (u - sV[64 + o]) * sV[128 + o] * sV[192 + o] + sV[256 + o];
                    u = act_fwd(ACT, u);
                }
                v[j] = u;
            }
            if (!valid) continue;
            if (a.vec_y) st4(yr + o0, make_float4(v[0], v[1], v[2], v[3]));
            else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (o0 + j < a.Cout) yr[o0 + j] = v[j];
            }
        }
    }
}

// Per-row recomputation shared by the backward kernels: g = dy * act'(u), xhat.
template <int CIN>
__device__ __forceinline__ void row_grad(const RowLin& a, const float (&gv)[16], int lane, int ACT, const float (&x)[CIN],
                                         const float* sW, const float* sV, int o, bool bn, float dyv, float& g, float& xhat) {
    const float z = pre_act<CIN>(a, x, sW, sV, o, gv, lane);      // wave shuffle inside: all lanes call this
    float u = z;
    xhat = 0.f;
    if (bn) {
        xhat = (z - sV[64 + o]) * sV[128 + o];
        u = xhat * sV[192 + o] + sV[256 + o];
    }
    g = dyv * act_bwd(ACT, u);
}

template <int CIN>
__device__ __forceinline__ void load_dy16(const RowLin& a, long long row, bool valid, int o0, float (&d)[OC]) {
#pragma unroll
    for (int j = 0; j < OC; ++j) d[j] = 0.f;
    if (!valid) return;
    const float* p = a.dy + (size_t)row * a.Cout + o0;
    if (a.vec_y) {
#pragma unroll
        for (int q = 0; q < OC / 4; ++q)
            if (o0 + q * 4 < a.Cout) {
                const float4 v = ld4(p + q * 4);
                d[q * 4] = v.x; d[q * 4 + 1] = v.y; d[q * 4 + 2] = v.z; d[q * 4 + 3] = v.w;
            }
    } else {
#pragma unroll
        for (int j = 0; j < OC; ++j)
            if (o0 + j < a.Cout) d[j] = p[j];
    }
}

// ---- K4: sum g, sum g*xhat per channel (BatchNorm backward reductions) ----------------------------
template <int CIN>
__global__ __launch_bounds__(BLOCK) void rowlin_bwd_reduce_kernel(const RowLin a) {
    const int ACT = a.act;
    extern __shared__ __align__(16) float smem[];
    float* sW = smem;
    float* sV = sW + 64 * CIN;
    float* sZ = sV + 7 * 64;
    float* red = sZ + NWAVE * 2 * 64 * ZS;
    stage_weights<CIN>(a, sW, sV);
    __syncthreads();
    const int lane = lane_id(), wave = wave_id();
    float* t1 = sZ + wave * 2 * 64 * ZS;
    float* t2 = t1 + 64 * ZS;
    float s1[MAXCH] = {0.f, 0.f, 0.f, 0.f}, s2[MAXCH] = {0.f, 0.f, 0.f, 0.f};
    const long long ntiles = (a.R + WAVE - 1) / WAVE;
    for (long long t = (long long)blockIdx.x * NWAVE + wave; t < ntiles; t += (long long)gridDim.x * NWAVE) {
        const long long row = t * WAVE + lane;
        const bool valid = row < a.R;
        float x[CIN];
        load_row<CIN>(a, row, valid, x);
        float gv[16];
        load_gadd(a, gather_row_index(a, row, valid), gv);
#pragma unroll
        for (int ch = 0; ch < MAXCH; ++ch) {
            if (ch * OC < a.Cout) {
                float d[OC];
                load_dy16<CIN>(a, row, valid, ch * OC, d);
#pragma unroll
                for (int j = 0; j < OC; ++j) {
                    const int o = ch * OC + j;
                    float g = 0.f, xh = 0.f;
                    if (o < a.Cout) row_grad<CIN>(a, gv, lane, ACT, x, sW, sV, o, true, d[j], g, xh);
                    if (!valid) { g = 0.f; xh = 0.f; }
                    t1[lane * ZS + j] = g;
                    t2[lane * ZS + j] = g * xh;
                }
                tile_colsum(t1, t2, lane, s1[ch], s2[ch]);
            }
        }
    }
    write_block_partials(s1, s2, red, a.part);
}

// ---- K5: reductions -> dgamma, dbeta, m1, m2 ------------------------------------------------------
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* __restrict__ part, int nblocks, long long R,
                                                                int Cout, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta, float* __restrict__ m1,
                                                                float* __restrict__ m2) {
    __shared__ double sh[64][16];
    double s0, s1;
    reduce_partials_8ch(part, nblocks, sh, s0, s1);
    const int o = blockIdx.x * 8 + threadIdx.x;
    if (threadIdx.x < 8 && o < Cout) {
        dbeta[o] = (float)s0; m1[o] = (float)(s0 / (double)R);
        dgamma[o] = (float)s1; m2[o] = (float)(s1 / (double)R);
    }
}

// ---- K6: dx rows + per-workgroup partial dW / db ---------------------------------------------------
// LDS row strides of the MFMA operand tiles: a multiple of 4 floats with an odd number of quads, so
// that a lane writing its row with 16-byte stores and the fragment reads both stay conflict-light.
__host__ __device__ constexpr int tile_stride(int c) {
    int s = (c < 16 ? 16 : c) + 4;
    return ((s / 4) % 2 == 0) ? s + 4 : s;
}

template <int CIN>
__global__ __launch_bounds__(BLOCK) void rowlin_bwd_apply_kernel(const RowLin a) {
    const int ACT = a.act;
    constexpr int NT = (CIN + 15) / 16;            // 16-wide column tiles of x
    constexpr int XS = tile_stride(CIN);           // x tile row stride
    constexpr int DS = tile_stride(OC);            // dz tile row stride (= 20)
    extern __shared__ __align__(16) float smem[];
    float* sW = smem;
    float* sV = sW + 64 * CIN;
    float* sX = sV + 7 * 64;                        // per wave [64][XS]
    float* sD = sX + NWAVE * 64 * XS;               // per wave [64][DS]
    float* red = sD + NWAVE * 64 * DS;              // [64][NT*16] combine buffer + [64] db
    int* sGi = reinterpret_cast<int*>(red + 64 * NT * 16 + 64);   // per wave [64] gather targets
    stage_weights<CIN>(a, sW, sV);
    const int lane = lane_id(), wave = wave_id();
    int* gi = sGi + wave * 64;
    float* xt = sX + wave * 64 * XS;
    float* dt = sD + wave * 64 * DS;
    for (int u = lane; u < 64 * XS; u += WAVE) xt[u] = 0.f;     // pad columns stay zero for ever
    __syncthreads();
    const bool bn = a.mean != nullptr;
    const bool bstat = bn && a.batch_stats;
    f32x4 acc[MAXCH][NT];
#pragma unroll
    for (int ch = 0; ch < MAXCH; ++ch)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[ch][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dbs[MAXCH] = {0.f, 0.f, 0.f, 0.f};
    const int fo = lane & 15, fq = lane >> 4;
    const long long ntiles = (a.R + WAVE - 1) / WAVE;
    for (long long t = (long long)blockIdx.x * NWAVE + wave; t < ntiles; t += (long long)gridDim.x * NWAVE) {
        const long long row = t * WAVE + lane;
        const bool valid = row < a.R;
        float x[CIN];
        load_row<CIN>(a, row, valid, x);
#pragma unroll
        for (int q = 0; q < CIN / 4; ++q)
            st4(xt + lane * XS + q * 4, make_float4(x[q * 4], x[q * 4 + 1], x[q * 4 + 2], x[q * 4 + 3]));
        float gv[16];
        const long long grow = gather_row_index(a, row, valid);
        load_gadd(a, grow, gv);
        float dxr[CIN];
#pragma unroll
        for (int i = 0; i < CIN; ++i) dxr[i] = 0.f;
#pragma unroll
        for (int ch = 0; ch < MAXCH; ++ch) {
            if (ch * OC < a.Cout) {
                float d[OC];
                load_dy16<CIN>(a, row, valid, ch * OC, d);
                // dz = dL/d(pre-activation) for the 16 channels of this round
#pragma unroll
                for (int j = 0; j < OC; ++j) {
                    const int o = ch * OC + j;
                    float dz = 0.f;
                    if (o < a.Cout) {
                        float g, xh;
                        row_grad<CIN>(a, gv, lane, ACT, x, sW, sV, o, bn, d[j], g, xh);
                        dz = g;
                        if (bn) dz = sV[128 + o] * sV[192 + o] * (bstat ? (g - sV[320 + o] - xh * sV[384 + o]) : g);
                    }
                    d[j] = valid ? dz : 0.f;
                }
#pragma unroll
                for (int q = 0; q < OC / 4; ++q)
                    st4(dt + lane * DS + q * 4, make_float4(d[q * 4], d[q * 4 + 1], d[q * 4 + 2], d[q * 4 + 3]));
                {   // db: column sums of the dz tile, lane (o, quarter)
                    const float* p = dt + (fq * 16) * DS + fo;
                    float sdb = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) sdb += p[r * DS];
                    dbs[ch] += sdb;
                }
                if (a.group > 1) {
                    // z[k] = t[k] - t[first] + b  =>  dt[k] = dz[k] - [k is first] * sum over the group of dz
                    const bool first = (lane & (a.group - 1)) == 0;
#pragma unroll
                    for (int j = 0; j < OC; ++j) {
                        float tot = d[j];
                        for (int off = 1; off < a.group; off <<= 1) tot += __shfl_xor(tot, off, WAVE);
                        if (first) d[j] -= tot;
                    }
#pragma unroll
                    for (int q = 0; q < OC / 4; ++q)
                        st4(dt + lane * DS + q * 4, make_float4(d[q * 4], d[q * 4 + 1], d[q * 4 + 2], d[q * 4 + 3]));
                }
                if (a.dx) {
#pragma unroll
                    for (int j = 0; j < OC; ++j) {
                        const int o = ch * OC + j;
                        if (o < a.Cout) {
                            const float dz = d[j];
                            const float* wr = sW + o * CIN;
#pragma unroll
                            for (int q = 0; q < CIN / 4; ++q) {
                                const float4 wv = ld4(wr + q * 4);
                                dxr[q * 4] = fmaf(dz, wv.x, dxr[q * 4]);
                                dxr[q * 4 + 1] = fmaf(dz, wv.y, dxr[q * 4 + 1]);
                                dxr[q * 4 + 2] = fmaf(dz, wv.z, dxr[q * 4 + 2]);
                                dxr[q * 4 + 3] = fmaf(dz, wv.w, dxr[q * 4 + 3]);
                            }
                        }
                    }
                }
                // dW[o][i] += sum_rows dt[row][o] * x[row][i]   (A = dt^T, B = x), 16 k-steps of 4 rows
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const float av = dt[(s * 4 + fq) * DS + fo];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const float bv = xt[(s * 4 + fq) * XS + nt * 16 + fo];
                        acc[ch][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[ch][nt], 0, 0, 0);
                    }
                }
                if (a.dgadd && ch == 0) {
                    // scatter dt into the per-point table: lanes walk the tile as (row, channel) pairs so that
                    // one wave instruction covers whole rows of the table.
                    gi[lane] = (int)grow;                        // -1 = no target; tables are < 2^31 rows
                    const int C = a.Cout;
                    for (int e = lane; e < WAVE * C; e += WAVE) {
                        const int r = e / C, o = e - r * C;
                        const int tgt = gi[r];
                        if (tgt >= 0) atomicAdd(a.dgadd + (size_t)tgt * C + o, dt[r * DS + o]);
                    }
                }
            }
        }
        if (a.dx && valid) {
            float* p = a.dx + (size_t)row * a.Cin;
            if (a.vec_x) {
#pragma unroll
                for (int q = 0; q < CIN / 4; ++q)
                    if (q * 4 < a.Cin) st4(p + q * 4, make_float4(dxr[q * 4], dxr[q * 4 + 1], dxr[q * 4 + 2], dxr[q * 4 + 3]));
            } else {
#pragma unroll
                for (int i = 0; i < CIN; ++i)
                    if (i < a.Cin) p[i] = dxr[i];
            }
        }
    }
    // combine the four waves: partial layout per workgroup = [64 (o)][NT*16 (i)] then [64] db
    constexpr int PW = NT * 16;
    float* outp = a.part + (size_t)blockIdx.x * (64 * PW + 64);
    __syncthreads();
    for (int w = 0; w < NWAVE; ++w) {
        if (wave == w) {
#pragma unroll
            for (int ch = 0; ch < MAXCH; ++ch) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int o = ch * 16 + fq * 4 + r, i = nt * 16 + fo;     // D: row = 4*(lane>>4)+r, col = lane&15
                        float* q = red + o * PW + i;
                        *q = (w == 0 ? 0.f : *q) + acc[ch][nt][r];
                    }
                float sdb = dbs[ch];
                sdb += __shfl_xor(sdb, 16, WAVE);
                sdb += __shfl_xor(sdb, 32, WAVE);
                if (lane < 16) {
                    float* q = red + 64 * PW + ch * 16 + lane;
                    *q = (w == 0 ? 0.f : *q) + sdb;
                }
            }
        }
        __syncthreads();
    }
    for (int u = threadIdx.x; u < 64 * PW + 64; u += BLOCK) outp[u] = red[u];
}

// ---- K7: sum the per-workgroup partials -> dW [Cout, Cin], db [Cout] ------------------------------
// 64 outputs per workgroup x 16 slices of the partial list; consecutive lanes read consecutive addresses
// of one partial, four loads in flight per lane; fixed combine order (deterministic).
__global__ __launch_bounds__(1024) void rowlin_param_reduce_kernel(const float* __restrict__ part, int nblocks, int PW,
                                                                    int Cin, int Cout, float* __restrict__ dW,
                                                                    float* __restrict__ db) {
    __shared__ float sh[16][64];
    const int total = Cout * Cin + Cout;
    const int stride = 64 * PW + 64;
    const int u = blockIdx.x * 64 + (threadIdx.x & 63);
    const int slice = threadIdx.x >> 6;
    float acc = 0.f;
    if (u < total) {
        int src;
        if (u < Cout * Cin) { const int o = u / Cin, i = u - o * Cin; src = o * PW + i; }
        else src = 64 * PW + (u - Cout * Cin);
        const float* q = part + src;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int p = slice;
        for (; p + 48 < nblocks; p += 64) {
            a0 += q[(size_t)p * stride];
            a1 += q[(size_t)(p + 16) * stride];
            a2 += q[(size_t)(p + 32) * stride];
            a3 += q[(size_t)(p + 48) * stride];
        }
        for (; p < nblocks; p += 16) a0 += q[(size_t)p * stride];
        acc = (a0 + a1) + (a2 + a3);
    }
    sh[slice][threadIdx.x & 63] = acc;
    __syncthreads();
    if (threadIdx.x < 64 && u < total) {
        float v = 0.f;
#pragma unroll
        for (int sl = 0; sl < 16; ++sl) v += sh[sl][threadIdx.x];
        if (u < Cout * Cin) dW[u] = v; else db[u - Cout * Cin] = v;
    }
}

// ---- host ---------------------------------------------------------------------------------------
static int cin_template(int Cin) {
    if (Cin <= 4) return 4;
    if (Cin <= 8) return 8;
    if (Cin <= 12) return 12;
    if (Cin <= 16) return 16;
    if (Cin <= 32) return 32;
    if (Cin <= 48) return 48;
    if (Cin <= 64) return 64;
    return 0;
}
static int rowlin_grid(long long R) {
    const long long tiles = (R + WAVE - 1) / WAVE;
    return (int)std::max<long long>(1, std::min<long long>((tiles + NWAVE - 1) / NWAVE, 512));
}
static size_t lds_stats(int CT) { return (size_t)(64 * CT + 7 * 64 + NWAVE * 2 * 64 * ZS + NWAVE * 128) * 4; }
static size_t lds_fwd(int CT) { return (size_t)(64 * CT + 7 * 64) * 4; }
static size_t lds_apply(int CT) {
    const int NT = (CT + 15) / 16;
    return (size_t)(64 * CT + 7 * 64 + NWAVE * 64 * tile_stride(CT) + NWAVE * 64 * tile_stride(OC) + 64 * NT * 16 + 64 + NWAVE * 64) * 4;
}
static size_t part_floats_stats(int grid) { return (size_t)grid * 128; }
static size_t part_floats_apply(int grid, int CT) { return (size_t)grid * (64 * ((CT + 15) / 16) * 16 + 64); }

template <typename KernelT>
static int launch_rowlin(KernelT k, const RowLin& a, int grid, size_t lds, hipStream_t s, const char* what) {
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(PCF_E_LAUNCH, "%s: hipFuncSetAttribute: %s", what, hipGetErrorString(e));
    }
    hipLaunchKernelGGL(k, dim3(grid), dim3(BLOCK), lds, s, a);
    return check_launch(what);
}

#define PCF_CIN_SWITCH(CT, EXPR)            \
    switch (CT) {                           \
        case 4: { constexpr int C = 4; EXPR; } break;   \
        case 8: { constexpr int C = 8; EXPR; } break;   \
        case 12: { constexpr int C = 12; EXPR; } break; \
        case 16: { constexpr int C = 16; EXPR; } break; \
        case 32: { constexpr int C = 32; EXPR; } break; \
        case 48: { constexpr int C = 48; EXPR; } break; \
        default: { constexpr int C = 64; EXPR; } break; \
    }
static int check_rowlin(const char* who, long long R, int Cin, int Cout, int act) {
    PCF_REQUIRE(R >= 0 && Cin >= 1 && Cout >= 1, "%s: bad sizes (R=%lld Cin=%d Cout=%d)", who, R, Cin, Cout);
    if (Cin > 64 || Cout > 64)
        return fail(PCF_E_UNSUPPORTED, "%s: Cin=%d / Cout=%d exceed the 64-channel limit of the per-edge MLP kernels", who, Cin, Cout);
    PCF_REQUIRE(act >= 0 && act <= 3, "%s: unknown activation %d", who, act);
    return PCF_OK;
}

}  // namespace pcf

extern "C" {

size_t pcf_hip_rowlin_workspace_bytes(int Cin, int Cout) {
    (void)Cout;
    const int CT = pcf::cin_template(Cin > 0 ? Cin : 1);
    // lane-per-row kernels: up to 512 workgroup partials; matrix-core kernels: up to 2048
    const size_t a = pcf::part_floats_stats(2048), b = pcf::part_floats_apply(2048, CT ? CT : 64);
    return (std::max(a, b) + 4 * 64) * 4 + 256;
}

static int check_extras(const char* who, long long R, int Cout, const float* gadd, const int64_t* gidx,
                        long long rows_per_batch, int gN, int group) {
    using namespace pcf;
    if (!gadd && group <= 1) return PCF_OK;
    if (Cout > 16) return fail(PCF_E_UNSUPPORTED, "%s: gathered term / key subtraction need Cout <= 16 (got %d)", who, Cout);
    if (gadd) PCF_REQUIRE(gidx && gN >= 0 && rows_per_batch > 0 && R % rows_per_batch == 0,
                          "%s: gathered term needs gidx, gN >= 0 and rows_per_batch dividing R", who);
    if (group > 1) PCF_REQUIRE(group <= 64 && (group & (group - 1)) == 0 && R % group == 0,
                               "%s: group must be a power of two <= 64 dividing R (got %d)", who, group);
    return PCF_OK;
}

int pcf_hip_rowlin_bn_stats_ex(const float* x, long long R, int Cin, const float* W, const float* b, int Cout, float eps,
                               float momentum, float* running_mean, float* running_var, float* mean_out, float* rstd_out,
                               const float* gadd, const int64_t* gidx, long long rows_per_batch, int gN, int group,
                               void* workspace, size_t workspace_bytes, void* stream) {
    using namespace pcf;
    if (int e = check_rowlin("rowlin_bn_stats", R, Cin, Cout, 0)) return e;
    if (int e = check_extras("rowlin_bn_stats", R, Cout, gadd, gidx, rows_per_batch, gN, group)) return e;
    PCF_REQUIRE(R > 0, "rowlin_bn_stats: batch statistics of zero rows");
    PCF_REQUIRE(x && W && b && mean_out && rstd_out && workspace, "rowlin_bn_stats: null pointer");
    PCF_REQUIRE(workspace_bytes >= pcf_hip_rowlin_workspace_bytes(Cin, Cout) && aligned16(workspace),
                "rowlin_bn_stats: workspace too small or misaligned");
    PCF_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "rowlin_bn_stats: need both running stats or none");
    hipStream_t s = (hipStream_t)stream;
    const int CT = cin_template(Cin);
    const int grid = rowlin_grid(R);
    RowLin a{};
    a.x = x; a.W = W; a.b = b; a.R = R; a.Cin = Cin; a.Cout = Cout;
    a.gadd = gadd; a.gidx = gidx; a.rows_per_batch = rows_per_batch; a.gN = gN; a.group = group;
    a.part = static_cast<float*>(workspace);
    a.vec_x = (Cin % 4 == 0) && aligned16(x);
    int rc = PCF_OK;
    int nparts = grid;
    if (rowlin_mfma_supported(a)) {
        nparts = rowlin_mfma_grid(R, Cin);
        rc = rowlin_mfma_stats(a, nparts, s);
    } else {
        PCF_CIN_SWITCH(CT, rc = launch_rowlin(rowlin_stats_kernel<C>, a, grid, lds_stats(C), s, "per-edge linear: BN statistics"));
    }
    if (rc) return rc;
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(8), dim3(1024), 0, s, a.part, nparts, R, Cout, eps, momentum,
                       running_mean, running_var, mean_out, rstd_out);
    return check_launch("per-edge linear: BN finalize");
}

int pcf_hip_rowlin_bn_stats(const float* x, long long R, int Cin, const float* W, const float* b, int Cout, float eps,
                            float momentum, float* running_mean, float* running_var, float* mean_out, float* rstd_out,
                            void* workspace, size_t workspace_bytes, void* stream) {
    return pcf_hip_rowlin_bn_stats_ex(x, R, Cin, W, b, Cout, eps, momentum, running_mean, running_var, mean_out, rstd_out,
                                      nullptr, nullptr, 0, 0, 0, workspace, workspace_bytes, stream);
}

int pcf_hip_rowlin_forward_ex(const float* x, long long R, int Cin, const float* W, const float* b, int Cout,
                              const float* mean, const float* rstd, const float* gamma, const float* beta, int act,
                              const float* gadd, const int64_t* gidx, long long rows_per_batch, int gN, int group,
                              float* y, void* stream) {
    using namespace pcf;
    if (int e = check_rowlin("rowlin_forward", R, Cin, Cout, act)) return e;
    if (int e = check_extras("rowlin_forward", R, Cout, gadd, gidx, rows_per_batch, gN, group)) return e;
    if (R == 0) return ok();
    PCF_REQUIRE(x && W && b && y, "rowlin_forward: null pointer");
    PCF_REQUIRE(!mean || (rstd && gamma && beta), "rowlin_forward: BN needs mean, rstd, gamma and beta");
    const int CT = cin_template(Cin);
    RowLin a{};
    a.x = x; a.y = y; a.W = W; a.b = b; a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.beta = beta;
    a.R = R; a.Cin = Cin; a.Cout = Cout; a.act = act;
    a.gadd = gadd; a.gidx = gidx; a.rows_per_batch = rows_per_batch; a.gN = gN; a.group = group;
    a.vec_x = (Cin % 4 == 0) && aligned16(x);
    a.vec_y = (Cout % 4 == 0) && aligned16(y);
    int rc = PCF_OK;
    if (rowlin_mfma_supported(a)) return rowlin_mfma_forward(a, (hipStream_t)stream);
    PCF_CIN_SWITCH(CT, rc = launch_rowlin(rowlin_fwd_kernel<C>, a, rowlin_grid(R), lds_fwd(C), (hipStream_t)stream,
                                          "per-edge linear forward"));
    return rc;
}

int pcf_hip_rowlin_forward(const float* x, long long R, int Cin, const float* W, const float* b, int Cout,
                           const float* mean, const float* rstd, const float* gamma, const float* beta, int act,
                           float* y, void* stream) {
    return pcf_hip_rowlin_forward_ex(x, R, Cin, W, b, Cout, mean, rstd, gamma, beta, act, nullptr, nullptr, 0, 0, 0, y, stream);
}

int pcf_hip_rowlin_backward_ex(const float* x, const float* dy, long long R, int Cin, const float* W, const float* b,
                               int Cout, const float* mean, const float* rstd, const float* gamma, const float* beta,
                               int batch_stats, int act, const float* gadd, const int64_t* gidx, long long rows_per_batch,
                               int gN, int group, float* dx, float* dW, float* db, float* dgamma, float* dbeta,
                               float* dgadd, void* workspace, size_t workspace_bytes, void* stream) {
    using namespace pcf;
    if (int e = check_rowlin("rowlin_backward", R, Cin, Cout, act)) return e;
    if (int e = check_extras("rowlin_backward", R, Cout, gadd, gidx, rows_per_batch, gN, group)) return e;
    PCF_REQUIRE(W && b && dW && db && workspace, "rowlin_backward: null pointer");
    PCF_REQUIRE(workspace_bytes >= pcf_hip_rowlin_workspace_bytes(Cin, Cout) && aligned16(workspace),
                "rowlin_backward: workspace too small or misaligned");
    PCF_REQUIRE(!gadd || dgadd, "rowlin_backward: gathered term given but its gradient buffer is null");
    const bool bn = mean != nullptr;
    PCF_REQUIRE(!bn || (rstd && gamma && beta && dgamma && dbeta), "rowlin_backward: BN needs rstd/gamma/beta and their grads");
    hipStream_t s = (hipStream_t)stream;
    if (gadd && rows_per_batch > 0) {
        const size_t n = (size_t)(R / rows_per_batch) * gN * Cout;
        if (n) {
            hipError_t e = zero_async(dgadd, n * 4, s);
            if (e != hipSuccess) return fail(PCF_E_LAUNCH, "rowlin_backward: memset: %s", hipGetErrorString(e));
        }
    }
    if (R == 0) {
        (void)zero_async(dW, (size_t)Cout * Cin * 4, s);
        (void)zero_async(db, (size_t)Cout * 4, s);
        if (bn) { (void)zero_async(dgamma, (size_t)Cout * 4, s); (void)zero_async(dbeta, (size_t)Cout * 4, s); }
        return ok();
    }
    PCF_REQUIRE(x && dy, "rowlin_backward: null pointer");
    const int CT = cin_template(Cin);
    const int grid = rowlin_grid(R);
    float* wsf = static_cast<float*>(workspace);
    float* m1 = wsf;                 // [64]
    float* m2 = wsf + 64;            // [64]
    float* part = wsf + 256;         // partial sums
    RowLin a{};
    a.x = x; a.dy = dy; a.dx = dx; a.W = W; a.b = b; a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.beta = beta;
    a.R = R; a.Cin = Cin; a.Cout = Cout; a.batch_stats = batch_stats; a.part = part; a.act = act;
    a.gadd = gadd; a.gidx = gidx; a.dgadd = gadd ? dgadd : nullptr; a.rows_per_batch = rows_per_batch; a.gN = gN; a.group = group;
    a.vec_x = (Cin % 4 == 0) && aligned16(x) && (!dx || aligned16(dx));
    a.vec_y = (Cout % 4 == 0) && aligned16(dy);
    int rc = PCF_OK;
    const bool mfma = rowlin_mfma_supported(a);
    const int nparts = mfma ? rowlin_mfma_grid(R, Cin) : grid;
    if (bn) {
        // dgamma / dbeta (and, with batch statistics, the two means the dz formula needs)
        if (mfma) rc = rowlin_mfma_bwd_reduce(a, nparts, s);
        else PCF_CIN_SWITCH(CT, rc = launch_rowlin(rowlin_bwd_reduce_kernel<C>, a, grid, lds_stats(C), s,
                                                   "per-edge linear: BN backward reductions"));
        if (rc) return rc;
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(8), dim3(1024), 0, s, part, nparts, R, Cout, dgamma, dbeta, m1, m2);
        if (int e = check_launch("per-edge linear: BN backward finalize")) return e;
        a.m1 = m1; a.m2 = m2;
    }
    if (mfma) rc = rowlin_mfma_bwd_apply(a, nparts, s);
    else PCF_CIN_SWITCH(CT, rc = launch_rowlin(rowlin_bwd_apply_kernel<C>, a, grid, lds_apply(C), s, "per-edge linear backward"));
    if (rc) return rc;
    const int PW = mfma ? ((Cin + 15) / 16) * 16 : ((CT + 15) / 16) * 16;
    hipLaunchKernelGGL(rowlin_param_reduce_kernel, dim3(ceil_div(Cout * Cin + Cout, 64)), dim3(1024), 0, s, part, nparts,
                       PW, Cin, Cout, dW, db);
    return check_launch("per-edge linear: parameter-gradient reduction");
}

int pcf_hip_rowlin_backward(const float* x, const float* dy, long long R, int Cin, const float* W, const float* b,
                            int Cout, const float* mean, const float* rstd, const float* gamma, const float* beta,
                            int batch_stats, int act, float* dx, float* dW, float* db, float* dgamma, float* dbeta,
                            void* workspace, size_t workspace_bytes, void* stream) {
    return pcf_hip_rowlin_backward_ex(x, dy, R, Cin, W, b, Cout, mean, rstd, gamma, beta, batch_stats, act, nullptr, nullptr,
                                      0, 0, 0, dx, dW, db, dgamma, dbeta, nullptr, workspace, workspace_bytes, stream);
}

}  // extern "C"
