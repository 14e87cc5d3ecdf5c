// Matrix-core versions of the four per-edge Linear+BatchNorm+activation kernels (see edge_mlp.hip for the
// operator and the lane-per-row versions, which remain the fallback for 64x64-class layers and K > 16).
//
// Transposed formulation on v_mfma_f32_16x16x4_f32 (exact fp32), tile = 16 rows:
//     Z[o][p] = sum_c W[o][c] * X[c][p]          A = weight fragment (resident in VGPRs for the whole kernel)
//                                                 B = the row tile, loaded straight from global memory:
//                                                     lane (p = l & 15, g = l >> 4) reads X[row p][4g .. 4g+3]
//                                                     (16 bytes) and register s is the B operand of the
//                                                     contraction step that covers channels {4k + s}
// The accumulator comes out as "lane (p, g), register r = channel 4g + r of row p", which is again a 16-byte
// piece of the output row: y, dx and the BatchNorm arithmetic need no data movement at all, per-channel
// sums are per-lane running sums reduced over the 16 row lanes once at the end, and the backward data
// path dX = W^T dZ is the same trick with transposed weight fragments.  Only the weight gradient
// dW = dZ . X^T contracts over rows and needs the two operands transposed through a 16x17 LDS tile.
// Compared with the lane-per-row kernels (weights broadcast from LDS, one FMA per MAC): no LDS traffic for
// weights, 16x fewer issued instructions, and the kernels become plain HBM streams.
#include <algorithm>

#include "edge_mlp.h"

namespace pcf {

#define PCF_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_16x16x4f32((A), (B), (C), 0, 0, 0)

__device__ __forceinline__ float act_f(int act, float u) {
    if (act == ACT_RELU) return fmaxf(u, 0.f);
    if (act == ACT_LEAKY) return u > 0.f ? u : 0.1f * u;
    if (act == ACT_SIGMOID) return 1.f / (1.f + __expf(-u));
    return u;
}
__device__ __forceinline__ float act_d(int act, float u) {
    if (act == ACT_RELU) return u > 0.f ? 1.f : 0.f;
    if (act == ACT_LEAKY) return u > 0.f ? 1.f : 0.1f;
    if (act == ACT_SIGMOID) { const float s = 1.f / (1.f + __expf(-u)); return s * (1.f - s); }
    return 1.f;
}

// per-channel vectors in LDS: [0]=bias [1]=mean [2]=rstd [3]=gamma [4]=beta [5]=m1 [6]=m2, 64 wide
__device__ __forceinline__ void stage_vectors(const RowLin& a, float (*sv)[64]) {
    for (int u = threadIdx.x; u < 7 * 64; u += BLOCK) {
        const int v = u >> 6, o = u & 63;
        const float* src = v == 0 ? a.b : v == 1 ? a.mean : v == 2 ? a.rstd : v == 3 ? a.gamma : v == 4 ? a.beta
                         : v == 5 ? a.m1 : a.m2;
        sv[v][o] = (src && o < a.Cout) ? src[o] : ((v == 2 || v == 3) ? 1.f : 0.f);
    }
}

template <int TI, int TO>
struct Tile {
    f32x4 x[TI];
    f32x4 z[TO];
};

template <int TI, int TO>
__device__ __forceinline__ void load_weights(const RowLin& a, int p, int g, f32x4 (&w)[TO][TI]) {
#pragma unroll
    for (int to = 0; to < TO; ++to)
#pragma unroll
        for (int ti = 0; ti < TI; ++ti)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int o = 16 * to + p, c = 16 * ti + 4 * g + s;
                w[to][ti][s] = (o < a.Cout && c < a.Cin) ? a.W[o * a.Cin + c] : 0.f;
            }
}

template <int TI>
__device__ __forceinline__ void load_rows(const RowLin& a, long long row, bool valid, int g, f32x4 (&x)[TI]) {
#pragma unroll
    for (int ti = 0; ti < TI; ++ti) {
        x[ti] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int c = 16 * ti + 4 * g;
        if (valid && c < a.Cin) {
            const float* q = a.x + (size_t)row * a.Cin + c;
            if (a.vec_x) { const float4 v = ld4(q); x[ti][0] = v.x; x[ti][1] = v.y; x[ti][2] = v.z; x[ti][3] = v.w; }
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) x[ti][r] = (c + r < a.Cin) ? q[r] : 0.f;
            }
        }
    }
}

// z = x.W^T (+ gathered per-point term) (- the same for the first row of the group) + b, D-layout
template <int TI, int TO>
__device__ __forceinline__ void pre_activation(const RowLin& a, const f32x4 (&w)[TO][TI], const f32x4 (&x)[TI], long long row,
                                               bool valid, int lane, int p, int g, const float (*sv)[64], long long& grow,
                                               f32x4 (&z)[TO]) {
#pragma unroll
    for (int to = 0; to < TO; ++to) {
        z[to] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ti = 0; ti < TI; ++ti)
#pragma unroll
            for (int s = 0; s < 4; ++s) z[to] = PCF_MFMA(w[to][ti][s], x[ti][s], z[to]);
    }
    grow = -1;
    if (a.gadd) {
        if (valid) {
            const int64_t j = a.gidx[row];
            if (j >= 0 && j < a.gN) grow = (row / a.rows_per_batch) * a.gN + j;
        }
        if (grow >= 0 && 4 * g < a.Cout) {
            const float* q = a.gadd + (size_t)grow * a.Cout + 4 * g;
#pragma unroll
            for (int r = 0; r < 4; ++r) if (4 * g + r < a.Cout) z[0][r] += q[r];
        }
    }
    if (a.group > 1) {
        const int lead = (lane & ~15) | (p & ~(a.group - 1));
#pragma unroll
        for (int r = 0; r < 4; ++r) z[0][r] -= __shfl(z[0][r], lead, WAVE);
    }
#pragma unroll
    for (int to = 0; to < TO; ++to) {
        const float4 bv = ld4(&sv[0][16 * to + 4 * g]);
        z[to][0] += bv.x; z[to][1] += bv.y; z[to][2] += bv.z; z[to][3] += bv.w;
    }
}

// reduce per-lane channel sums over the 16 row lanes and write the workgroup partial [2][64]
template <int TO>
__device__ __forceinline__ void write_sums(f32x4 (&s1)[TO], f32x4 (&s2)[TO], float (*red)[2][64], float* part, int lane, int wave,
                                           int p, int g) {
    for (int u = threadIdx.x; u < NWAVE * 128; u += BLOCK) (&red[0][0][0])[u] = 0.f;
    __syncthreads();
#pragma unroll
    for (int to = 0; to < TO; ++to)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v1 = s1[to][r], v2 = s2[to][r];
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) { v1 += __shfl_xor(v1, off, WAVE); v2 += __shfl_xor(v2, off, WAVE); }
            if (p == 0) { red[wave][0][16 * to + 4 * g + r] = v1; red[wave][1][16 * to + 4 * g + r] = v2; }
        }
    __syncthreads();
    if (threadIdx.x < 128) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NWAVE; ++w) t += (&red[w][0][0])[threadIdx.x];
        part[(size_t)blockIdx.x * 128 + threadIdx.x] = t;
    }
}

// ---- statistics: sum z, sum z^2 ------------------------------------------------------------------------
template <int TI, int TO>
__global__ __launch_bounds__(BLOCK) void rowlin_mfma_stats_kernel(const RowLin a) {
    __shared__ __align__(16) float sv[7][64];
    __shared__ float red[NWAVE][2][64];
    stage_vectors(a, sv);
    __syncthreads();
    const int lane = lane_id(), wave = wave_id(), p = lane & 15, g = lane >> 4;
    f32x4 w[TO][TI];
    load_weights<TI, TO>(a, p, g, w);
    f32x4 s1[TO], s2[TO];
#pragma unroll
    for (int to = 0; to < TO; ++to) { s1[to] = f32x4{0.f, 0.f, 0.f, 0.f}; s2[to] = s1[to]; }
    const long long ntiles = (a.R + 15) / 16;
    // the rows of the next tile are requested before this one is processed (short inputs are latency-bound)
    const long long tstep = (long long)gridDim.x * NWAVE;
    long long t = (long long)blockIdx.x * NWAVE + wave;
    f32x4 xn[TI];
    load_rows<TI>(a, t * 16 + p, t * 16 + p < a.R, g, xn);
    for (; t < ntiles; t += tstep) {
        const long long row = t * 16 + p;
        const bool valid = row < a.R;
        f32x4 x[TI], z[TO];
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) x[ti] = xn[ti];
        if (t + tstep < ntiles) load_rows<TI>(a, (t + tstep) * 16 + p, (t + tstep) * 16 + p < a.R, g, xn);
        long long grow;
        pre_activation<TI, TO>(a, w, x, row, valid, lane, p, g, sv, grow, z);
        const float m = valid ? 1.f : 0.f;
#pragma unroll
        for (int to = 0; to < TO; ++to)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = (16 * to + 4 * g + r < a.Cout) ? z[to][r] * m : 0.f;
                s1[to][r] += v; s2[to][r] += v * v;
            }
    }
    write_sums<TO>(s1, s2, red, a.part, lane, wave, p, g);
}

// ---- forward -------------------------------------------------------------------------------------------
template <int TI, int TO>
__global__ __launch_bounds__(BLOCK) void rowlin_mfma_fwd_kernel(const RowLin a) {
    __shared__ __align__(16) float sv[7][64];
    stage_vectors(a, sv);
    __syncthreads();
    const int lane = lane_id(), wave = wave_id(), p = lane & 15, g = lane >> 4;
    f32x4 w[TO][TI];
    load_weights<TI, TO>(a, p, g, w);
    const bool bn = a.mean != nullptr;
    const long long ntiles = (a.R + 15) / 16;
    // the rows of the next tile are requested before this one is processed (short inputs are latency-bound)
    const long long tstep = (long long)gridDim.x * NWAVE;
    long long t = (long long)blockIdx.x * NWAVE + wave;
    f32x4 xn[TI];
    load_rows<TI>(a, t * 16 + p, t * 16 + p < a.R, g, xn);
    for (; t < ntiles; t += tstep) {
        const long long row = t * 16 + p;
        const bool valid = row < a.R;
        f32x4 x[TI], z[TO];
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) x[ti] = xn[ti];
        if (t + tstep < ntiles) load_rows<TI>(a, (t + tstep) * 16 + p, (t + tstep) * 16 + p < a.R, g, xn);
        long long grow;
        pre_activation<TI, TO>(a, w, x, row, valid, lane, p, g, sv, grow, z);
        if (!valid) continue;
#pragma unroll
        for (int to = 0; to < TO; ++to) {
            const int c0 = 16 * to + 4 * g;
            if (c0 >= a.Cout) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float u = z[to][r];
                if (bn) u = (u - sv[1][c0 + r]) * sv[2][c0 + r] * sv[3][c0 + r] + sv[4][c0 + r];
                v[r] = act_f(a.act, u);
            }
            float* q = a.y + (size_t)row * a.Cout + c0;
            if (a.vec_y) st4(q, make_float4(v[0], v[1], v[2], v[3]));
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (c0 + r < a.Cout) q[r] = v[r];
            }
        }
    }
}

// upstream gradient of the four channels of every output tile held by this lane (zeros outside the matrix)
template <int TO>
__device__ __forceinline__ void load_dy(const RowLin& a, long long row, bool valid, int g, f32x4 (&dv)[TO]) {
#pragma unroll
    for (int to = 0; to < TO; ++to) {
        const int c0 = 16 * to + 4 * g;
        dv[to] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!valid || c0 >= a.Cout) continue;
        const float* q = a.dy + (size_t)row * a.Cout + c0;
        if (a.vec_y) { const float4 v = ld4(q); dv[to][0] = v.x; dv[to][1] = v.y; dv[to][2] = v.z; dv[to][3] = v.w; }
        else {
#pragma unroll
            for (int r = 0; r < 4; ++r) dv[to][r] = (c0 + r < a.Cout) ? q[r] : 0.f;
        }
    }
}

// g = dy * act'(u) and xhat for the four channels of tile `to` held by this lane
template <int TO>
__device__ __forceinline__ void grad_terms(const RowLin& a, const f32x4 (&z)[TO], const f32x4 (&dv)[TO], bool valid, int g,
                                           bool bn, const float (*sv)[64], f32x4 (&gr)[TO], f32x4 (&xh)[TO]) {
#pragma unroll
    for (int to = 0; to < TO; ++to) {
        const int c0 = 16 * to + 4 * g;
        gr[to] = f32x4{0.f, 0.f, 0.f, 0.f};
        xh[to] = gr[to];
        if (!valid || c0 >= a.Cout) continue;
        const f32x4 d = dv[to];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (c0 + r >= a.Cout) continue;
            float u = z[to][r];
            if (bn) {
                xh[to][r] = (u - sv[1][c0 + r]) * sv[2][c0 + r];
                u = xh[to][r] * sv[3][c0 + r] + sv[4][c0 + r];
            }
            gr[to][r] = d[r] * act_d(a.act, u);
        }
    }
}

// ---- backward reductions: sum g, sum g*xhat -------------------------------------------------------------
template <int TI, int TO>
__global__ __launch_bounds__(BLOCK) void rowlin_mfma_bwd_reduce_kernel(const RowLin a) {
    __shared__ __align__(16) float sv[7][64];
    __shared__ float red[NWAVE][2][64];
    stage_vectors(a, sv);
    __syncthreads();
    const int lane = lane_id(), wave = wave_id(), p = lane & 15, g = lane >> 4;
    f32x4 w[TO][TI];
    load_weights<TI, TO>(a, p, g, w);
    f32x4 s1[TO], s2[TO];
#pragma unroll
    for (int to = 0; to < TO; ++to) { s1[to] = f32x4{0.f, 0.f, 0.f, 0.f}; s2[to] = s1[to]; }
    const long long ntiles = (a.R + 15) / 16;
    const long long tstep = (long long)gridDim.x * NWAVE;
    long long t = (long long)blockIdx.x * NWAVE + wave;
    f32x4 xn[TI], dn[TO];
    load_rows<TI>(a, t * 16 + p, t * 16 + p < a.R, g, xn);
    load_dy<TO>(a, t * 16 + p, t * 16 + p < a.R, g, dn);
    for (; t < ntiles; t += tstep) {
        const long long row = t * 16 + p;
        const bool valid = row < a.R;
        f32x4 x[TI], dv[TO], z[TO], gr[TO], xh[TO];
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) x[ti] = xn[ti];
#pragma unroll
        for (int to = 0; to < TO; ++to) dv[to] = dn[to];
        if (t + tstep < ntiles) {
            load_rows<TI>(a, (t + tstep) * 16 + p, (t + tstep) * 16 + p < a.R, g, xn);
            load_dy<TO>(a, (t + tstep) * 16 + p, (t + tstep) * 16 + p < a.R, g, dn);
        }
        long long grow;
        pre_activation<TI, TO>(a, w, x, row, valid, lane, p, g, sv, grow, z);
        grad_terms<TO>(a, z, dv, valid, g, true, sv, gr, xh);
#pragma unroll
        for (int to = 0; to < TO; ++to)
#pragma unroll
            for (int r = 0; r < 4; ++r) { s1[to][r] += gr[to][r]; s2[to][r] += gr[to][r] * xh[to][r]; }
    }
    write_sums<TO>(s1, s2, red, a.part, lane, wave, p, g);
}

// ---- backward apply: dx rows, per-workgroup partial dW / db, scatter of the gathered term's gradient ----
constexpr int TT = 17;      // row stride of the 16x16 transposition tiles

template <int TI, int TO>
__global__ __launch_bounds__(BLOCK) void rowlin_mfma_bwd_apply_kernel(const RowLin a) {
    constexpr int PW = TI * 16;
    __shared__ __align__(16) float sv[7][64];
    __shared__ float tz[NWAVE][TO][16 * TT];      // dz tiles, [channel][row]
    __shared__ float tx[NWAVE][TI][16 * TT];      // x tiles,  [channel][row]
    __shared__ float red[64 * PW + 64];
    __shared__ int gi[NWAVE][16];
    stage_vectors(a, sv);
    for (int u = threadIdx.x; u < 64 * PW + 64; u += BLOCK) red[u] = 0.f;
    __syncthreads();
    const int lane = lane_id(), wave = wave_id(), p = lane & 15, g = lane >> 4;
    const bool bn = a.mean != nullptr;
    const bool bstat = bn && a.batch_stats;
    f32x4 w[TO][TI], wt[TI][TO];
    load_weights<TI, TO>(a, p, g, w);
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int to = 0; to < TO; ++to)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int o = 16 * to + 4 * g + s, c = 16 * ti + p;       // A[i = c][k <-> o = 4k + s]
                wt[ti][to][s] = (o < a.Cout && c < a.Cin) ? a.W[o * a.Cin + c] : 0.f;
            }
    f32x4 accw[TO][TI], dbs[TO];
#pragma unroll
    for (int to = 0; to < TO; ++to) {
        dbs[to] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) accw[to][ti] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const long long ntiles = (a.R + 15) / 16;
    const long long tstep = (long long)gridDim.x * NWAVE;
    long long t = (long long)blockIdx.x * NWAVE + wave;
    f32x4 xn[TI], dn[TO];
    load_rows<TI>(a, t * 16 + p, t * 16 + p < a.R, g, xn);
    load_dy<TO>(a, t * 16 + p, t * 16 + p < a.R, g, dn);
    for (; t < ntiles; t += tstep) {
        const long long row = t * 16 + p;
        const bool valid = row < a.R;
        f32x4 x[TI], dv[TO], z[TO], dz[TO], xh[TO];
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) x[ti] = xn[ti];
#pragma unroll
        for (int to = 0; to < TO; ++to) dv[to] = dn[to];
        if (t + tstep < ntiles) {
            load_rows<TI>(a, (t + tstep) * 16 + p, (t + tstep) * 16 + p < a.R, g, xn);
            load_dy<TO>(a, (t + tstep) * 16 + p, (t + tstep) * 16 + p < a.R, g, dn);
        }
        long long grow;
        pre_activation<TI, TO>(a, w, x, row, valid, lane, p, g, sv, grow, z);
        grad_terms<TO>(a, z, dv, valid, g, bn, sv, dz, xh);
#pragma unroll
        for (int to = 0; to < TO; ++to)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 16 * to + 4 * g + r;
                float d = dz[to][r];
                if (bn) d = sv[2][c & 63] * sv[3][c & 63] * (bstat ? (d - sv[5][c & 63] - xh[to][r] * sv[6][c & 63]) : d);
                if (!valid || c >= a.Cout) d = 0.f;
                dz[to][r] = d;
                dbs[to][r] += d;
            }
        if (a.group > 1) {       // z[k] = t[k] - t[first] + b  =>  dt[k] = dz[k] - [k first] * sum over the group
            const bool first = (p & (a.group - 1)) == 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float tot = dz[0][r];
                for (int off = 1; off < a.group; off <<= 1) tot += __shfl_xor(tot, off, WAVE);
                if (first) dz[0][r] -= tot;
            }
        }
        if (a.dx) {
#pragma unroll
            for (int ti = 0; ti < TI; ++ti) {
                f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int to = 0; to < TO; ++to)
#pragma unroll
                    for (int s = 0; s < 4; ++s) d = PCF_MFMA(wt[ti][to][s], dz[to][s], d);
                const int c0 = 16 * ti + 4 * g;
                if (valid && c0 < a.Cin) {
                    float* q = a.dx + (size_t)row * a.Cin + c0;
                    if (a.vec_x) st4(q, make_float4(d[0], d[1], d[2], d[3]));
                    else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (c0 + r < a.Cin) q[r] = d[r];
                    }
                }
            }
        }
        // dW[o][c] += sum over the 16 rows of dz[o][row] * x[c][row]: both operands transposed through LDS
#pragma unroll
        for (int to = 0; to < TO; ++to)
#pragma unroll
            for (int r = 0; r < 4; ++r) tz[wave][to][(4 * g + r) * TT + p] = dz[to][r];
#pragma unroll
        for (int ti = 0; ti < TI; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) tx[wave][ti][(4 * g + r) * TT + p] = x[ti][r];
        if (a.dgadd) {
            // Gradient of the gathered per-point term: scatter dt rows with float atomics.  Straight from the
            // accumulator layout every lane would add four isolated 4-byte words (measured: the atomics alone
            // took ~180 us); re-read from the transposition tile instead so that consecutive lanes cover
            // consecutive channels of one table row (Cout lanes = one contiguous row segment).
            if (g == 0) gi[wave][p] = (int)grow;
            const int C = a.Cout;
            for (int e = lane; e < 16 * C; e += WAVE) {
                const int r = e / C, o = e - r * C;
                const int tgt = gi[wave][r];
                if (tgt >= 0) atomicAdd(a.dgadd + (size_t)tgt * C + o, tz[wave][0][o * TT + r]);
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float af[TO], bf[TI];
#pragma unroll
            for (int to = 0; to < TO; ++to) af[to] = tz[wave][to][p * TT + 4 * s + g];     // A[i = o = p][k = g <-> row 4s + g]
#pragma unroll
            for (int ti = 0; ti < TI; ++ti) bf[ti] = tx[wave][ti][p * TT + 4 * s + g];     // B[k = g][j = c = p]
#pragma unroll
            for (int to = 0; to < TO; ++to)
#pragma unroll
                for (int ti = 0; ti < TI; ++ti) accw[to][ti] = PCF_MFMA(af[to], bf[ti], accw[to][ti]);
        }
    }
    // combine the four waves in wave order (deterministic): red[o][c], then db
    for (int wv = 0; wv < NWAVE; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int to = 0; to < TO; ++to) {
#pragma unroll
                for (int ti = 0; ti < TI; ++ti)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[(16 * to + 4 * g + r) * PW + 16 * ti + p] += accw[to][ti][r];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = dbs[to][r];
#pragma unroll
                    for (int off = 1; off < 16; off <<= 1) v += __shfl_xor(v, off, WAVE);
                    if (p == 0) red[64 * PW + 16 * to + 4 * g + r] += v;
                }
            }
        }
        __syncthreads();
    }
    float* outp = a.part + (size_t)blockIdx.x * (64 * PW + 64);
    for (int u = threadIdx.x; u < 64 * PW + 64; u += BLOCK) outp[u] = red[u];
}

// ---- host dispatch --------------------------------------------------------------------------------------
static inline int tiles_of(int c) { return (c + 15) / 16; }

bool rowlin_mfma_supported(const RowLin& a) {
    const int ti = tiles_of(a.Cin), to = tiles_of(a.Cout);
    if (a.Cin < 1 || a.Cout < 1 || ti > 4 || to > 4 || ti * to > 4) return false;
    if (a.group > 16) return false;
    return true;
}

// A wave keeps one 16-row tile of loads in flight, so the latency hiding comes from resident waves: up to 2048
// workgroups.  Against that, every workgroup pays a fixed prologue and (backward) writes a Cout x 16*TI partial of
// dW -- 16 KB at Cin = 64 -- so wide-input layers get TI tiles per wave: at R = 80k, Cin = 64 that is 313
// workgroups (24 us) instead of 1250 (32 us); the 8..16-wide layers take two tiles per wave.
int rowlin_mfma_grid(long long R, int Cin) {
    const long long tiles = (R + 15) / 16;
    const long long per_wg = (long long)NWAVE * std::max(tiles_of(Cin), 2);      // measured: 2 beats 1 and 4 for 16-wide inputs
    return (int)std::max<long long>(1, std::min<long long>((tiles + per_wg - 1) / per_wg, 2048));
}

#define PCF_TILE_SWITCH(KERNEL, GRID)                                                                              \
    do {                                                                                                           \
        const int ti = tiles_of(a.Cin), to = tiles_of(a.Cout);                                                     \
        const dim3 gd(GRID), bd(BLOCK);                                                                            \
        if (ti == 1 && to == 1) hipLaunchKernelGGL((KERNEL<1, 1>), gd, bd, 0, s, a);                               \
        else if (ti == 1 && to == 2) hipLaunchKernelGGL((KERNEL<1, 2>), gd, bd, 0, s, a);                          \
        else if (ti == 2 && to == 1) hipLaunchKernelGGL((KERNEL<2, 1>), gd, bd, 0, s, a);                          \
        else if (ti == 1 && to == 3) hipLaunchKernelGGL((KERNEL<1, 3>), gd, bd, 0, s, a);                          \
        else if (ti == 3 && to == 1) hipLaunchKernelGGL((KERNEL<3, 1>), gd, bd, 0, s, a);                          \
        else if (ti == 1 && to == 4) hipLaunchKernelGGL((KERNEL<1, 4>), gd, bd, 0, s, a);                          \
        else if (ti == 4 && to == 1) hipLaunchKernelGGL((KERNEL<4, 1>), gd, bd, 0, s, a);                          \
        else hipLaunchKernelGGL((KERNEL<2, 2>), gd, bd, 0, s, a);                                                  \
    } while (0)

int rowlin_mfma_stats(const RowLin& a, int grid, hipStream_t s) {
    PCF_TILE_SWITCH(rowlin_mfma_stats_kernel, grid);
    return check_launch("per-edge linear (MFMA): BN statistics");
}
int rowlin_mfma_forward(const RowLin& a, hipStream_t s) {
    PCF_TILE_SWITCH(rowlin_mfma_fwd_kernel, rowlin_mfma_grid(a.R, a.Cin));
    return check_launch("per-edge linear (MFMA) forward");
}
int rowlin_mfma_bwd_reduce(const RowLin& a, int grid, hipStream_t s) {
    PCF_TILE_SWITCH(rowlin_mfma_bwd_reduce_kernel, grid);
    return check_launch("per-edge linear (MFMA): BN backward reductions");
}
int rowlin_mfma_bwd_apply(const RowLin& a, int grid, hipStream_t s) {
    PCF_TILE_SWITCH(rowlin_mfma_bwd_apply_kernel, grid);
    return check_launch("per-edge linear (MFMA) backward");
}

}  // namespace pcf
