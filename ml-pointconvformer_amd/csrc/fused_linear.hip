// Point-level Linear + BatchNorm (+ activation) chains with the BatchNorm folded into the neighbouring contractions.
//
// The reference runs every point-level layer as Linear -> BatchNorm1d -> activation (layer_utils.py:241-315; in a
// PCFLayer: unary1, guidance_unary, linear, unary2, layers.py:335,369,393-400).  Layer-at-a-time execution costs ten
// launches per layer and step (contraction, column partials, finalize, normalise; backward: partials, finalize, dz,
// two contractions, slab sum) that each stream an [R, C] matrix through HBM once more.  Here a layer keeps ONE tensor,
// its raw pre-BatchNorm output z = x W^T + b, and one small record of per-channel constants
//
//     cst[0] = sc = rstd * gamma      cst[1] = sh = beta - mean * sc       (y = act(z * sc + sh))
//     cst[2] = mean                   cst[3] = rstd
//     cst[4] = D1, cst[5] = D0        (backward: dz = g * sc + z * D1 + D0,  g = dy * act'(z * sc + sh))
//
// and every consumer applies the normalisation on the fly:
//   flin_fwd_kernel     z' = act(z * sc + sh) W'^T + b'   -- the previous layer's BatchNorm + activation ride in the A-tile
//                       loader (the activated tensor is written once, as a side output, only where something else
//                       gathers it); the epilogue takes the column sums of z' and the LAST workgroup to finish turns the
//                       per-workgroup partials into mean / rstd / running statistics / (sc, sh): no separate statistics,
//                       finalize or normalise launches.
//   flin_bwd_in_kernel  dx = dz W with dz formed in the A-tile loader from (dy, z, cst); the epilogue adds a second
//                       gradient stream, stores dx, and -- when the input was itself act(BN(z_prev)) -- takes the two
//                       BatchNorm-backward column sums of the PRODUCING layer (sum g_prev, sum g_prev z_prev); the last
//                       workgroup writes that layer's dgamma, dbeta and (D1, D0).
//   flin_bwd_w_kernel   dW = dz^T act(BN(z_prev)) over row ranges (both operands formed in the loaders), slabs summed by
//                       the last workgroup of each output tile in a fixed order.
//   bn_bwd_stats_kernel the top of a chain: g = dy * act'(z * sc + sh + residual), its column sums, (D1, D0).
// Everything is deterministic: partials are combined in index order by whichever workgroup happens to be last.
// fp32 MFMA (v_mfma_f32_32x32x2_f32, exact products) as in gemm.hip; these products are HBM-bound.
#include <algorithm>

#include "pcf_common.h"
#include "flin_common.h"

namespace pcf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int FK = 16;                  // contraction step staged through LDS
constexpr int FL_MAXY = 1024;           // row-range workgroups per column tile (= partials the last one combines)

// [ROWS x FK] operand tile: global -> registers (float4 units) -> LDS image [k][ROWS + 4].  KCONT: element (r, k) at
// p[r * ld + k] (units run along k); else at p[k * ld + r] (units run along r).
template <int ROWS, bool KCONT, int KS = FK>
struct Tile {
    static constexpr int UNITS = ROWS * KS / 4;
    static constexpr int NV = (UNITS + BLOCK - 1) / BLOCK;
    // LDS row stride.  KCONT tiles are written with scalar stores, 16 lanes of a wave along k at KS = 64 (4 at KS = 16):
    // stride = 1 (mod 32) spreads those over all banks, where ROWS + 4 would put the 16 on two.
    static constexpr int STRIDE = (KCONT && KS >= 64) ? ROWS + 1 : ROWS + 4;
    float4 v[NV];

    __device__ __forceinline__ static bool unit(int i, int& r, int& k) {
        const int u = threadIdx.x + i * BLOCK;
        if (UNITS % BLOCK != 0 && u >= UNITS) return false;
        if (KCONT) { r = u / (KS / 4); k = (u % (KS / 4)) * 4; }
        else       { k = u / (ROWS / 4); r = (u % (ROWS / 4)) * 4; }
        return true;
    }
    // four consecutive elements along the unit's direction starting at (gr, gk); zero outside [rmax) x [kmax)
    template <bool VEC>
    __device__ __forceinline__ static float4 fetch(const float* p, long long ld, long long gr, long long gk, long long rmax, long long kmax) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (KCONT) {
            if (gr < rmax) {
                const float* q = p + gr * ld + gk;
                if (VEC && gk + 3 < kmax) t = ld4(q);
                else {
                    if (gk < kmax) t.x = q[0];
                    if (gk + 1 < kmax) t.y = q[1];
                    if (gk + 2 < kmax) t.z = q[2];
                    if (gk + 3 < kmax) t.w = q[3];
                }
            }
        } else {
            if (gk < kmax) {
                const float* q = p + gk * ld + gr;
                if (VEC && gr + 3 < rmax) t = ld4(q);
                else {
                    if (gr < rmax) t.x = q[0];
                    if (gr + 1 < rmax) t.y = q[1];
                    if (gr + 2 < rmax) t.z = q[2];
                    if (gr + 3 < rmax) t.w = q[3];
                }
            }
        }
        return t;
    }
    __device__ __forceinline__ void store(float* s) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int r, k;
            if (!unit(i, r, k)) break;
            if (KCONT) {
                s[(k + 0) * STRIDE + r] = v[i].x;
                s[(k + 1) * STRIDE + r] = v[i].y;
                s[(k + 2) * STRIDE + r] = v[i].z;
                s[(k + 3) * STRIDE + r] = v[i].w;
            } else {
                st4(s + k * STRIDE + r, v[i]);
            }
        }
    }
};

// four per-channel constants starting at channel c (zero beyond C)
__device__ __forceinline__ float4 cst4(const float* row, int c, int C) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c + 3 < C && (c & 3) == 0) return ld4(row + c);
    if (c < C) t.x = row[c];
    if (c + 1 < C) t.y = row[c + 1];
    if (c + 2 < C) t.z = row[c + 2];
    if (c + 3 < C) t.w = row[c + 3];
    return t;
}
__device__ __forceinline__ float4 affine_act4(float4 v, float4 sc, float4 sh, int act) {
    return make_float4(fl_act(act, v.x * sc.x + sh.x), fl_act(act, v.y * sc.y + sh.y), fl_act(act, v.z * sc.z + sh.z),
                       fl_act(act, v.w * sc.w + sh.w));
}
// dz of four elements: g = dy * act'(z * sc + sh), dz = g * sc + z * D1 + D0
__device__ __forceinline__ float4 dz4(float4 dy, float4 z, float4 sc, float4 sh, float4 d1, float4 d0, int act) {
    float4 o;
    o.x = dy.x * fl_dact(act, z.x * sc.x + sh.x) * sc.x + (z.x * d1.x + d0.x);
    o.y = dy.y * fl_dact(act, z.y * sc.y + sh.y) * sc.y + (z.y * d1.y + d0.y);
    o.z = dy.z * fl_dact(act, z.z * sc.z + sh.z) * sc.z + (z.z * d1.z + d0.z);
    o.w = dy.w * fl_dact(act, z.w * sc.w + sh.w) * sc.w + (z.w * d1.w + d0.w);
    return o;
}
__device__ __forceinline__ float4 mask4(float4 v, bool a, bool b, bool c, bool d) {
    return make_float4(a ? v.x : 0.f, b ? v.y : 0.f, c ? v.z : 0.f, d ? v.w : 0.f);
}

template <int WM, int WN, int STRIDE_A, int STRIDE_B, int KS = FK>
__device__ __forceinline__ f32x16 mfma_steps(const float* sA, const float* sB, int wm, int wn, int lane, f32x16 acc) {
    const float* pa = sA + wm * 32 + (lane & 31);
    const float* pb = sB + wn * 32 + (lane & 31);
    const int fk = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < KS; kk += 2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[(kk + fk) * STRIDE_A], pb[(kk + fk) * STRIDE_B], acc, 0, 0, 0);
    return acc;
}

// Per-workgroup column sums -> partial list -> totals in sh_tot[2][BN] (doubles) in the workgroup that finishes last.
// A two-level tree keeps every serial walk short: groups of FL_GROUP consecutive row-range workgroups; the last
// workgroup of a group sums that group's partials (index order) into a group partial, the last group to finish sums the
// group partials (index order).  Deterministic whichever workgroups happen to be last.  After the acquire half of
// __threadfence() plain loads see what the other workgroups released before taking their tickets.
//   s1 / s2: this lane's sums of its column (lanes l and l + 32 share a column);
//   part: [gridDim.y][2][N] followed by [groups][2][N];  ticket: gridDim.x * (FL_MAXG + 1) ints, zero on entry and exit.
constexpr int FL_GROUP = 32;
constexpr int FL_MAXG = (FL_MAXY + FL_GROUP - 1) / FL_GROUP;
template <int WM, int WN>
__device__ __forceinline__ bool column_totals(float s1, float s2, int n0, int N, float* part, int* ticket, double (*sh_tot)[WN * 32]) {
    constexpr int BN = WN * 32;
    __shared__ float csum[WM * WN][2][32];
    __shared__ int s_last;
    const int lane = lane_id(), wave = wave_id();
    s1 += __shfl_xor(s1, 32, WAVE);
    s2 += __shfl_xor(s2, 32, WAVE);
    if (lane < 32) { csum[wave][0][lane] = s1; csum[wave][1][lane] = s2; }
    __syncthreads();
    const int which = threadIdx.x / BN, cl = threadIdx.x % BN, col = n0 + cl;     // threads [0, 2 BN): (sum | second sum, column)
    const bool mine = threadIdx.x < 2 * BN && col < N;
    if (threadIdx.x < 2 * BN) {
        const int wn = cl >> 5, c = cl & 31;
        float a = 0.f;
#pragma unroll
        for (int wm = 0; wm < WM; ++wm) a += csum[wm * WN + wn][which][c];
        if (!ticket) {                               // a separate launch combines the lists (flin_finish_kernel)
            if (col < N) part[((size_t)blockIdx.y * 2 + which) * N + col] = a;
        } else if (col < N) st_agent(part + ((size_t)blockIdx.y * 2 + which) * N + col, a);
    }
    if (!ticket) return false;
    const int groups = ((int)gridDim.y + FL_GROUP - 1) / FL_GROUP, grp = blockIdx.y / FL_GROUP;
    const int gsize = min(FL_GROUP, (int)gridDim.y - grp * FL_GROUP);
    int* t1 = ticket + blockIdx.x * (FL_MAXG + 1);
    float* gpart = part + (size_t)gridDim.y * 2 * N;
    publish();
    __syncthreads();
    if (threadIdx.x == 0) s_last = (atomicAdd(&t1[grp], 1) == gsize - 1) ? 1 : 0;
    __syncthreads();
    if (!s_last) return false;
    observe();
    float gsum = 0.f;
    if (mine) {
        const float* p = part + ((size_t)(grp * FL_GROUP) * 2 + which) * N + col;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int i = 0;
        for (; i + 3 < gsize; i += 4) {
            a0 += ld_agent(p + (size_t)(i + 0) * 2 * N); a1 += ld_agent(p + (size_t)(i + 1) * 2 * N);
            a2 += ld_agent(p + (size_t)(i + 2) * 2 * N); a3 += ld_agent(p + (size_t)(i + 3) * 2 * N);
        }
        for (; i < gsize; ++i) a0 += ld_agent(p + (size_t)i * 2 * N);
        gsum = (a0 + a1) + (a2 + a3);
    }
    if (threadIdx.x == 0) st_agent(&t1[grp], 0);
    if (groups == 1) {                              // small problems: one level, one ticket
        if (threadIdx.x < 2 * BN) sh_tot[which][cl] = (double)gsum;
        __syncthreads();
        return true;
    }
    if (mine) st_agent(gpart + ((size_t)grp * 2 + which) * N + col, gsum);
    publish();
    __syncthreads();
    if (threadIdx.x == 0) s_last = (atomicAdd(&t1[FL_MAXG], 1) == groups - 1) ? 1 : 0;
    __syncthreads();
    if (!s_last) return false;
    observe();
    if (threadIdx.x < 2 * BN) {
        double a0 = 0.0, a1 = 0.0;
        if (col < N) {
            const float* p = gpart + (size_t)which * N + col;
            int i = 0;
            for (; i + 1 < groups; i += 2) { a0 += (double)ld_agent(p + (size_t)i * 2 * N); a1 += (double)ld_agent(p + (size_t)(i + 1) * 2 * N); }
            if (i < groups) a0 += (double)ld_agent(p + (size_t)i * 2 * N);
        }
        sh_tot[which][cl] = a0 + a1;
    }
    if (threadIdx.x == 0) st_agent(&t1[FL_MAXG], 0);
    __syncthreads();
    return true;
}

// ------------------------------------------------------------------------------------------------------------------
struct FwdArgs {
    const float* A;              // [M, K] raw input (or the producer's z when `pre` is given)
    const float* pre; int pre_act;   // producer's cst (rows 0, 1) and activation, or null: A is used as it is
    float* side;                 // [M, K] receives the transformed A, or null
    const float* W; const float* bias;   // [N, K], [N]
    float* Z;                    // [M, N]
    int M, N, K;
    float* part; int* ticket;    // statistics scratch (see column_totals); cst == null: no statistics
    float* cst;                  // [6][N]: rows 0..3 written
    const float* gamma; const float* beta; float* running_mean; float* running_var; float eps, momentum;
};

template <int WM, int WN, bool VEC>
__global__ __launch_bounds__(BLOCK) void flin_fwd_kernel(const FwdArgs g) {
    constexpr int BM = WM * 32, BN = WN * 32;
    using TA = Tile<BM, true>;
    using TB = Tile<BN, true>;
    __shared__ __align__(16) float sA[FK * TA::STRIDE];
    __shared__ __align__(16) float sB[FK * TB::STRIDE];
    __shared__ double sh_tot[2][BN];
    const int n0 = blockIdx.x * BN;
    const int wave = wave_id(), lane = lane_id(), wm = wave / WN, wn = wave % WN;
    const int mtiles = ceil_div_dev(g.M, BM);
    const int col = n0 + wn * 32 + (lane & 31);
    const float bv = (g.bias && col < g.N) ? g.bias[col] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    TA ta;
    TB tb;
    auto load_a = [&](int m0, int k0) {
#pragma unroll
        for (int i = 0; i < TA::NV; ++i) {
            int r, k;
            if (!TA::unit(i, r, k)) break;
            float4 t = TA::template fetch<VEC>(g.A, g.K, m0 + r, k0 + k, g.M, g.K);
            if (g.pre) {
                t = affine_act4(t, cst4(g.pre, k0 + k, g.K), cst4(g.pre + g.K, k0 + k, g.K), g.pre_act);
                if (g.side && blockIdx.x == 0 && m0 + r < g.M) {
                    float* q = g.side + (size_t)(m0 + r) * g.K + k0 + k;
                    if (VEC && k0 + k + 3 < g.K) st4(q, t);
                    else {
                        if (k0 + k < g.K) q[0] = t.x;
                        if (k0 + k + 1 < g.K) q[1] = t.y;
                        if (k0 + k + 2 < g.K) q[2] = t.z;
                        if (k0 + k + 3 < g.K) q[3] = t.w;
                    }
                }
            }
            ta.v[i] = t;
        }
    };
    auto load_b = [&](int k0) {
#pragma unroll
        for (int i = 0; i < TB::NV; ++i) {
            int r, k;
            if (!TB::unit(i, r, k)) break;
            tb.v[i] = TB::template fetch<VEC>(g.W, g.K, n0 + r, k0 + k, g.N, g.K);
        }
    };
    for (int mt = blockIdx.y; mt < mtiles; mt += gridDim.y) {
        const int m0 = mt * BM;
        f32x16 acc = {0};
        load_a(m0, 0);
        load_b(0);
        for (int k0 = 0; k0 < g.K; k0 += FK) {
            __syncthreads();
            ta.store(sA);
            tb.store(sB);
            __syncthreads();
            if (k0 + FK < g.K) { load_a(m0, k0 + FK); load_b(k0 + FK); }
            acc = mfma_steps<WM, WN, TA::STRIDE, TB::STRIDE>(sA, sB, wm, wn, lane, acc);
        }
        if (col < g.N) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < g.M) {
                    const float v = acc[r] + bv;
                    g.Z[(size_t)row * g.N + col] = v;
                    s1 += v;
                    s2 += v * v;
                }
            }
        }
    }
    if (!g.cst) return;
    if (!column_totals<WM, WN>(s1, s2, n0, g.N, g.part, g.ticket, sh_tot)) return;
    if (threadIdx.x < BN && n0 + threadIdx.x < g.N) {
        const int c = n0 + threadIdx.x;
        const double n = (double)g.M, mean = sh_tot[0][threadIdx.x] / n;
        double var = sh_tot[1][threadIdx.x] / n - mean * mean;
        if (var < 0.0) var = 0.0;
        const double rs = 1.0 / sqrt(var + (double)g.eps);
        const double sc = rs * (double)g.gamma[c];
        g.cst[0 * g.N + c] = (float)sc;
        g.cst[1 * g.N + c] = (float)((double)g.beta[c] - mean * sc);
        g.cst[2 * g.N + c] = (float)mean;
        g.cst[3 * g.N + c] = (float)rs;
        if (g.running_mean) {
            const double unbiased = g.M > 1 ? var * n / (n - 1.0) : var;
            g.running_mean[c] = (float)((1.0 - g.momentum) * g.running_mean[c] + g.momentum * mean);
            g.running_var[c] = (float)((1.0 - g.momentum) * g.running_var[c] + g.momentum * unbiased);
        }
    }
}

// Few rows, long contraction (the coarse levels of the models: 40-700 tiles of 64 x 64, K up to 1536): a launch is bound by
// the serial matrix work of ONE workgroup (K / 2 products of 64 cycles per wave), not by memory.  Split-K form: a 32 x 32
// tile per workgroup (4 x the workgroups), 64 contraction elements staged per step of which each of the four waves takes 16,
// the four accumulators summed through LDS in wave order (deterministic), then the usual epilogue with one thread per four
// outputs.  Same arguments, statistics and ticket layout as flin_fwd_kernel (column tiles are 32 wide).
constexpr int SK = 64;
template <int STRIDE_A, int STRIDE_B>
__device__ __forceinline__ f32x16 mfma_quarter(const float* sA, const float* sB, int wave, int lane, f32x16 acc) {
    const float* pa = sA + (lane & 31);
    const float* pb = sB + (lane & 31);
    const int k0 = wave * (SK / NWAVE) + (lane >> 5);
#pragma unroll
    for (int kk = 0; kk < SK / NWAVE; kk += 2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[(k0 + kk) * STRIDE_A], pb[(k0 + kk) * STRIDE_B], acc, 0, 0, 0);
    return acc;
}
// the four waves' 32 x 32 accumulators -> sums of four outputs per thread: rows (threadIdx.x >> 5) + 8 i, column threadIdx.x & 31
__device__ __forceinline__ void reduce_quarters(f32x16 acc, float (*sred)[32 * 33], float (&out)[4]) {
    const int lane = lane_id(), wave = wave_id();
#pragma unroll
    for (int r = 0; r < 16; ++r) sred[wave][((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 33 + (lane & 31)] = acc[r];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int o = ((threadIdx.x >> 5) + 8 * i) * 33 + (threadIdx.x & 31);
        out[i] = (sred[0][o] + sred[1][o]) + (sred[2][o] + sred[3][o]);
    }
    __syncthreads();
}

__global__ __launch_bounds__(BLOCK) void flin_fwd_sk_kernel(const FwdArgs g) {
    using TA = Tile<32, true, SK>;
    using TB = Tile<32, true, SK>;
    __shared__ __align__(16) float sA[SK * TA::STRIDE];
    __shared__ __align__(16) float sB[SK * TB::STRIDE];
    __shared__ float sred[NWAVE][32 * 33];
    __shared__ double sh_tot[2][32];
    const int n0 = blockIdx.x * 32;
    const int wave = wave_id(), lane = lane_id();
    const int mtiles = ceil_div_dev(g.M, 32);
    const int col = n0 + (threadIdx.x & 31);
    const float bv = (g.bias && col < g.N) ? g.bias[col] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    TA ta;
    TB tb;
    auto load_a = [&](int m0, int k0) {
#pragma unroll
        for (int i = 0; i < TA::NV; ++i) {
            int r, k;
            if (!TA::unit(i, r, k)) break;
            float4 t = TA::template fetch<true>(g.A, g.K, m0 + r, k0 + k, g.M, g.K);
            if (g.pre) {
                t = affine_act4(t, cst4(g.pre, k0 + k, g.K), cst4(g.pre + g.K, k0 + k, g.K), g.pre_act);
                if (g.side && blockIdx.x == 0 && m0 + r < g.M && k0 + k + 3 < g.K) st4(g.side + (size_t)(m0 + r) * g.K + k0 + k, t);
            }
            ta.v[i] = t;
        }
    };
    auto load_b = [&](int k0) {
#pragma unroll
        for (int i = 0; i < TB::NV; ++i) {
            int r, k;
            if (!TB::unit(i, r, k)) break;
            tb.v[i] = TB::template fetch<true>(g.W, g.K, n0 + r, k0 + k, g.N, g.K);
        }
    };
    for (int mt = blockIdx.y; mt < mtiles; mt += gridDim.y) {
        const int m0 = mt * 32;
        f32x16 acc = {0};
        load_a(m0, 0);
        load_b(0);
        for (int k0 = 0; k0 < g.K; k0 += SK) {
            __syncthreads();
            ta.store(sA);
            tb.store(sB);
            __syncthreads();
            if (k0 + SK < g.K) { load_a(m0, k0 + SK); load_b(k0 + SK); }
            acc = mfma_quarter<TA::STRIDE, TB::STRIDE>(sA, sB, wave, lane, acc);
        }
        float o[4];
        reduce_quarters(acc, sred, o);
        if (col < g.N) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = m0 + (threadIdx.x >> 5) + 8 * i;
                if (row < g.M) {
                    const float v = o[i] + bv;
                    g.Z[(size_t)row * g.N + col] = v;
                    s1 += v;
                    s2 += v * v;
                }
            }
        }
    }
    if (!g.cst) return;
    if (!column_totals<4, 1>(s1, s2, n0, g.N, g.part, g.ticket, sh_tot)) return;
    if (threadIdx.x < 32 && n0 + threadIdx.x < g.N)
        bn_fwd_constants(g.cst, g.N, n0 + threadIdx.x, sh_tot[0][threadIdx.x], sh_tot[1][threadIdx.x], (double)g.M, g.gamma, g.beta,
                         g.running_mean, g.running_var, g.eps, g.momentum);
}

// ------------------------------------------------------------------------------------------------------------------
struct BwdInArgs {
    const float* dy; const float* z;     // [M, K]: gradient w.r.t. this layer's activated output, its raw output
    const float* cst; int act;           // [6][K] of this layer
    const float* W;                      // [K, N]
    const float* add;                    // [M, N] second gradient stream into the input, or null
    float* dx;                           // [M, N]
    const float* zp; float* cstp; int actp;   // producing layer: raw output [M, N], record [6][N] (rows 4, 5 written), activation
    float* dgamma_p; float* dbeta_p; float* dbias_p;
    float* part; int* ticket;
    int M, N, K;
};

template <int WM, int WN, bool VEC>
__global__ __launch_bounds__(BLOCK) void flin_bwd_in_kernel(const BwdInArgs g) {
    constexpr int BM = WM * 32, BN = WN * 32;
    using TA = Tile<BM, true>;
    using TB = Tile<BN, false>;
    __shared__ __align__(16) float sA[FK * TA::STRIDE];
    __shared__ __align__(16) float sB[FK * TB::STRIDE];
    __shared__ double sh_tot[2][BN];
    const int n0 = blockIdx.x * BN;
    const int wave = wave_id(), lane = lane_id(), wm = wave / WN, wn = wave % WN;
    const int mtiles = ceil_div_dev(g.M, BM);
    const int col = n0 + wn * 32 + (lane & 31);
    float scp = 0.f, shp = 0.f;
    if (g.cstp && col < g.N) { scp = g.cstp[col]; shp = g.cstp[g.N + col]; }
    float s1 = 0.f, s2 = 0.f;
    TA ta;
    TB tb;
    auto load_a = [&](int m0, int k0) {
#pragma unroll
        for (int i = 0; i < TA::NV; ++i) {
            int r, k;
            if (!TA::unit(i, r, k)) break;
            const int gk = k0 + k;
            const float4 dy = TA::template fetch<VEC>(g.dy, g.K, m0 + r, gk, g.M, g.K);
            const float4 z = TA::template fetch<VEC>(g.z, g.K, m0 + r, gk, g.M, g.K);
            float4 t = dz4(dy, z, cst4(g.cst, gk, g.K), cst4(g.cst + g.K, gk, g.K), cst4(g.cst + 4 * g.K, gk, g.K),
                           cst4(g.cst + 5 * g.K, gk, g.K), g.act);
            const bool rv = m0 + r < g.M;
            ta.v[i] = mask4(t, rv && gk < g.K, rv && gk + 1 < g.K, rv && gk + 2 < g.K, rv && gk + 3 < g.K);
        }
    };
    auto load_b = [&](int k0) {
#pragma unroll
        for (int i = 0; i < TB::NV; ++i) {
            int r, k;
            if (!TB::unit(i, r, k)) break;
            tb.v[i] = TB::template fetch<VEC>(g.W, g.N, n0 + r, k0 + k, g.N, g.K);
        }
    };
    for (int mt = blockIdx.y; mt < mtiles; mt += gridDim.y) {
        const int m0 = mt * BM;
        f32x16 acc = {0};
        load_a(m0, 0);
        load_b(0);
        for (int k0 = 0; k0 < g.K; k0 += FK) {
            __syncthreads();
            ta.store(sA);
            tb.store(sB);
            __syncthreads();
            if (k0 + FK < g.K) { load_a(m0, k0 + FK); load_b(k0 + FK); }
            acc = mfma_steps<WM, WN, TA::STRIDE, TB::STRIDE>(sA, sB, wm, wn, lane, acc);
        }
        if (col < g.N) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < g.M) {
                    const size_t o = (size_t)row * g.N + col;
                    float v = acc[r];
                    if (g.add) v += g.add[o];
                    g.dx[o] = v;
                    if (g.cstp) {
                        const float zp = g.zp[o];
                        const float gp = v * fl_dact(g.actp, zp * scp + shp);
                        s1 += gp;
                        s2 += gp * zp;
                    }
                }
            }
        }
    }
    if (!g.cstp) return;
    if (!column_totals<WM, WN>(s1, s2, n0, g.N, g.part, g.ticket, sh_tot)) return;
    if (threadIdx.x < BN && n0 + threadIdx.x < g.N)
        bn_bwd_constants(g.cstp, g.N, n0 + threadIdx.x, sh_tot[0][threadIdx.x], sh_tot[1][threadIdx.x], (double)g.M, g.dgamma_p, g.dbeta_p, g.dbias_p);
}

// split-K form of flin_bwd_in_kernel (see flin_fwd_sk_kernel)
__global__ __launch_bounds__(BLOCK) void flin_bwd_in_sk_kernel(const BwdInArgs g) {
    using TA = Tile<32, true, SK>;
    using TB = Tile<32, false, SK>;
    __shared__ __align__(16) float sA[SK * TA::STRIDE];
    __shared__ __align__(16) float sB[SK * TB::STRIDE];
    __shared__ float sred[NWAVE][32 * 33];
    __shared__ double sh_tot[2][32];
    const int n0 = blockIdx.x * 32;
    const int wave = wave_id(), lane = lane_id();
    const int mtiles = ceil_div_dev(g.M, 32);
    const int col = n0 + (threadIdx.x & 31);
    float scp = 0.f, shp = 0.f;
    if (g.cstp && col < g.N) { scp = g.cstp[col]; shp = g.cstp[g.N + col]; }
    float s1 = 0.f, s2 = 0.f;
    TA ta;
    TB tb;
    auto load_a = [&](int m0, int k0) {
#pragma unroll
        for (int i = 0; i < TA::NV; ++i) {
            int r, k;
            if (!TA::unit(i, r, k)) break;
            const int gk = k0 + k;
            const float4 dy = TA::template fetch<true>(g.dy, g.K, m0 + r, gk, g.M, g.K);
            const float4 z = TA::template fetch<true>(g.z, g.K, m0 + r, gk, g.M, g.K);
            float4 t = dz4(dy, z, cst4(g.cst, gk, g.K), cst4(g.cst + g.K, gk, g.K), cst4(g.cst + 4 * g.K, gk, g.K),
                           cst4(g.cst + 5 * g.K, gk, g.K), g.act);
            const bool rv = m0 + r < g.M;
            ta.v[i] = mask4(t, rv && gk < g.K, rv && gk + 1 < g.K, rv && gk + 2 < g.K, rv && gk + 3 < g.K);
        }
    };
    auto load_b = [&](int k0) {
#pragma unroll
        for (int i = 0; i < TB::NV; ++i) {
            int r, k;
            if (!TB::unit(i, r, k)) break;
            tb.v[i] = TB::template fetch<true>(g.W, g.N, n0 + r, k0 + k, g.N, g.K);
        }
    };
    for (int mt = blockIdx.y; mt < mtiles; mt += gridDim.y) {
        const int m0 = mt * 32;
        f32x16 acc = {0};
        load_a(m0, 0);
        load_b(0);
        for (int k0 = 0; k0 < g.K; k0 += SK) {
            __syncthreads();
            ta.store(sA);
            tb.store(sB);
            __syncthreads();
            if (k0 + SK < g.K) { load_a(m0, k0 + SK); load_b(k0 + SK); }
            acc = mfma_quarter<TA::STRIDE, TB::STRIDE>(sA, sB, wave, lane, acc);
        }
        float o[4];
        reduce_quarters(acc, sred, o);
        if (col < g.N) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = m0 + (threadIdx.x >> 5) + 8 * i;
                if (row < g.M) {
                    const size_t off = (size_t)row * g.N + col;
                    float v = o[i];
                    if (g.add) v += g.add[off];
                    g.dx[off] = v;
                    if (g.cstp) {
                        const float zp = g.zp[off];
                        const float gp = v * fl_dact(g.actp, zp * scp + shp);
                        s1 += gp;
                        s2 += gp * zp;
                    }
                }
            }
        }
    }
    if (!g.cstp) return;
    if (!column_totals<4, 1>(s1, s2, n0, g.N, g.part, g.ticket, sh_tot)) return;
    if (threadIdx.x < 32 && n0 + threadIdx.x < g.N)
        bn_bwd_constants(g.cstp, g.N, n0 + threadIdx.x, sh_tot[0][threadIdx.x], sh_tot[1][threadIdx.x], (double)g.M, g.dgamma_p, g.dbeta_p, g.dbias_p);
}

// ------------------------------------------------------------------------------------------------------------------
struct BwdWArgs {
    const float* dy; const float* z; const float* cst; int act;      // this layer: [R, M] twice, [6][M]
    const float* xin; const float* pre; int pre_act;                 // its input: [R, N] (raw), producer's record or null
    float* slabs;                // [gridDim.z][M][N]: one partial product per row range (summed by slab_sum_multi_kernel)
    int M, N;
    long long R;
    int r_per_split;             // multiple of BW_KS
};
constexpr int BW_KS = 32;                // rows per contraction step: six 16-byte loads in flight per lane
constexpr int BW_SPLITS_MAX = 512;
// The slabs are summed by a separate launch (one for all weight gradients of a chain): letting the last workgroup of a
// tile sum ~500 slabs of 16 KB behind tickets cost 40-65 us per product against 21-31 us for the product itself.

// Fixed-order sums of up to SSM_MAX slab lists in one launch: out_i[e] = sum_s slabs_i[s][e].  64 elements x 16 slices of
// the slab list per workgroup, four loads in flight per lane.
constexpr int SSM_MAX = 8;
struct SlabSumArgs {
    const float* slabs[SSM_MAX]; float* out[SSM_MAX]; long long count[SSM_MAX]; int splits[SSM_MAX]; int block0[SSM_MAX + 1]; int n;
};
__global__ __launch_bounds__(1024) void slab_sum_multi_kernel(const SlabSumArgs a) {
    __shared__ float sh[1024];
    int i = 0;
    while (i + 1 < a.n && (int)blockIdx.x >= a.block0[i + 1]) ++i;
    const long long e = (long long)(blockIdx.x - a.block0[i]) * 64 + (threadIdx.x & 63);
    const int slice = threadIdx.x >> 6, S = a.splits[i];
    const long long count = a.count[i];
    const float* src = a.slabs[i];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (e < count) {
        int s = slice;
        for (; s + 48 < S; s += 64) {
            a0 += src[(size_t)s * count + e];
            a1 += src[(size_t)(s + 16) * count + e];
            a2 += src[(size_t)(s + 32) * count + e];
            a3 += src[(size_t)(s + 48) * count + e];
        }
        for (; s < S; s += 16) a0 += src[(size_t)s * count + e];
    }
    sh[threadIdx.x] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (threadIdx.x < 64 && e < count) {
        float t = 0.f;
#pragma unroll
        for (int sl = 0; sl < 16; ++sl) t += sh[sl * 64 + threadIdx.x];
        a.out[i][e] = t;
    }
}

// tile 64 x 64 (WM = WN = 2) or 32 x 128 (WM = 1, WN = 4: layers with at most 32 output channels)
template <int WM, int WN, bool VEC>
__global__ __launch_bounds__(BLOCK) void flin_bwd_w_kernel(const BwdWArgs g) {
    constexpr int BM = WM * 32, BN = WN * 32;
    using TA = Tile<BM, false, BW_KS>;
    using TB = Tile<BN, false, BW_KS>;
    __shared__ __align__(16) float sA[BW_KS * TA::STRIDE];
    __shared__ __align__(16) float sB[BW_KS * TB::STRIDE];
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int wave = wave_id(), lane = lane_id(), wm = wave / WN, wn = wave % WN;
    const long long rbeg = (long long)blockIdx.z * g.r_per_split;
    const long long rend = rbeg + g.r_per_split < g.R ? rbeg + g.r_per_split : g.R;
    TA ta;
    TB tb;
    // per-unit channel constants do not change along the row walk: load them once
    float4 a_sc[TA::NV], a_sh[TA::NV], a_d1[TA::NV], a_d0[TA::NV], b_sc[TB::NV], b_sh[TB::NV];
#pragma unroll
    for (int i = 0; i < TA::NV; ++i) {
        int r, k;
        if (!TA::unit(i, r, k)) break;
        a_sc[i] = cst4(g.cst, m0 + r, g.M); a_sh[i] = cst4(g.cst + g.M, m0 + r, g.M);
        a_d1[i] = cst4(g.cst + 4 * g.M, m0 + r, g.M); a_d0[i] = cst4(g.cst + 5 * g.M, m0 + r, g.M);
    }
#pragma unroll
    for (int i = 0; i < TB::NV; ++i) {
        int r, k;
        if (!TB::unit(i, r, k)) break;
        if (g.pre) { b_sc[i] = cst4(g.pre, n0 + r, g.N); b_sh[i] = cst4(g.pre + g.N, n0 + r, g.N); }
    }
    auto load = [&](long long k0) {
#pragma unroll
        for (int i = 0; i < TA::NV; ++i) {
            int r, k;
            if (!TA::unit(i, r, k)) break;
            const float4 dy = TA::template fetch<VEC>(g.dy, g.M, m0 + r, k0 + k, g.M, rend);
            const float4 z = TA::template fetch<VEC>(g.z, g.M, m0 + r, k0 + k, g.M, rend);
            const float4 t = dz4(dy, z, a_sc[i], a_sh[i], a_d1[i], a_d0[i], g.act);
            const bool kv = k0 + k < rend;
            ta.v[i] = mask4(t, kv && m0 + r < g.M, kv && m0 + r + 1 < g.M, kv && m0 + r + 2 < g.M, kv && m0 + r + 3 < g.M);
        }
#pragma unroll
        for (int i = 0; i < TB::NV; ++i) {
            int r, k;
            if (!TB::unit(i, r, k)) break;
            float4 t = TB::template fetch<VEC>(g.xin, g.N, n0 + r, k0 + k, g.N, rend);
            if (g.pre) t = affine_act4(t, b_sc[i], b_sh[i], g.pre_act);
            tb.v[i] = t;                 // rows beyond rend meet a zero dz
        }
    };
    f32x16 acc = {0};
    if (rbeg < rend) load(rbeg);
    for (long long k0 = rbeg; k0 < rend; k0 += BW_KS) {
        __syncthreads();
        ta.store(sA);
        tb.store(sB);
        __syncthreads();
        if (k0 + BW_KS < rend) load(k0 + BW_KS);
        acc = mfma_steps<WM, WN, TA::STRIDE, TB::STRIDE, BW_KS>(sA, sB, wm, wn, lane, acc);
    }
    const int col = n0 + wn * 32 + (lane & 31);
    const size_t mn = (size_t)g.M * g.N;
    float* slab = g.slabs + (size_t)blockIdx.z * mn;
    if (col < g.N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < g.M) slab[(size_t)row * g.N + col] = acc[r];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Top of a chain: y = act(z * sc + sh + residual), given dy.  g = dy * act'(.) is written (it is also the gradient of the
// residual) and its column sums give this layer's dgamma, dbeta and (D1, D0).  Threads: TX column lanes x TY row lanes.
struct TopArgs {
    const float* dy; const float* z; const float* res;   // [R, C]
    float* cst; int act;
    float* g;                                            // [R, C]
    float* dgamma; float* dbeta; float* dbias;
    float* part; int* ticket;                            // part: [gridDim.x][2][C]; ticket: [gridDim.y]
    long long R; int C, TX; long long rows_per_block;
};

__global__ __launch_bounds__(BLOCK) void bn_bwd_stats_kernel(const TopArgs a) {
    __shared__ float s1[BLOCK], s2[BLOCK];
    __shared__ double red[BLOCK][2];
    __shared__ int s_last;
    const int TX = a.TX, TY = BLOCK / TX;
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const long long r0 = (long long)blockIdx.x * a.rows_per_block;
    const long long r1 = r0 + a.rows_per_block < a.R ? r0 + a.rows_per_block : a.R;
    const int c = blockIdx.y * TX + tx;
    float sa = 0.f, sb = 0.f;
    if (c < a.C) {
        const float sc = a.cst[c], sh = a.cst[a.C + c];
        float sa1 = 0.f, sb1 = 0.f, sa2 = 0.f, sb2 = 0.f, sa3 = 0.f, sb3 = 0.f;
        auto term = [&](long long r, float& pa, float& pb) {
            const size_t o = (size_t)r * a.C + c;
            const float v = a.z[o];
            const float gg = a.dy[o] * fl_dact(a.act, v * sc + sh + (a.res ? a.res[o] : 0.f));
            a.g[o] = gg;
            pa += gg; pb += gg * v;
        };
        long long r = r0 + ty;
        for (; r + 3 * TY < r1; r += 4 * TY) {        // four rows in flight per lane
            term(r, sa, sb); term(r + TY, sa1, sb1); term(r + 2 * TY, sa2, sb2); term(r + 3 * TY, sa3, sb3);
        }
        for (; r < r1; r += TY) term(r, sa, sb);
        sa = (sa + sa1) + (sa2 + sa3); sb = (sb + sb1) + (sb2 + sb3);
    }
    s1[threadIdx.x] = sa; s2[threadIdx.x] = sb;
    __syncthreads();
    if (ty == 0 && c < a.C) {
        for (int y = 1; y < TY; ++y) { sa += s1[y * TX + tx]; sb += s2[y * TX + tx]; }
        if (!a.ticket) {                             // a separate launch combines the lists (flin_finish_kernel)
            a.part[((size_t)blockIdx.x * 2 + 0) * a.C + c] = sa;
            a.part[((size_t)blockIdx.x * 2 + 1) * a.C + c] = sb;
        } else {
            st_agent(a.part + ((size_t)blockIdx.x * 2 + 0) * a.C + c, sa);
            st_agent(a.part + ((size_t)blockIdx.x * 2 + 1) * a.C + c, sb);
        }
    }
    if (!a.ticket) return;
    // two-level, fixed-order combination by whichever workgroups finish last (see column_totals)
    const int nb = gridDim.x, groups = (nb + FL_GROUP - 1) / FL_GROUP, grp = blockIdx.x / FL_GROUP;
    const int gsize = min(FL_GROUP, nb - grp * FL_GROUP);
    int* t1 = a.ticket + blockIdx.y * (FL_MAXG + 1);
    float* gpart = a.part + (size_t)nb * 2 * a.C;
    publish();
    __syncthreads();
    if (threadIdx.x == 0) s_last = (atomicAdd(&t1[grp], 1) == gsize - 1) ? 1 : 0;
    __syncthreads();
    if (!s_last) return;
    observe();
    if (ty < 2 && c < a.C) {                      // ty = which sum
        const float* p = a.part + ((size_t)(grp * FL_GROUP) * 2 + ty) * a.C + c;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int i = 0;
        for (; i + 3 < gsize; i += 4) {
            a0 += ld_agent(p + (size_t)(i + 0) * 2 * a.C); a1 += ld_agent(p + (size_t)(i + 1) * 2 * a.C);
            a2 += ld_agent(p + (size_t)(i + 2) * 2 * a.C); a3 += ld_agent(p + (size_t)(i + 3) * 2 * a.C);
        }
        for (; i < gsize; ++i) a0 += ld_agent(p + (size_t)i * 2 * a.C);
        st_agent(gpart + ((size_t)grp * 2 + ty) * a.C + c, (a0 + a1) + (a2 + a3));
    }
    if (threadIdx.x == 0) st_agent(&t1[grp], 0);
    publish();
    __syncthreads();
    if (threadIdx.x == 0) s_last = (atomicAdd(&t1[FL_MAXG], 1) == groups - 1) ? 1 : 0;
    __syncthreads();
    if (!s_last) return;
    observe();
    if (ty < 2 && c < a.C) {
        double a0 = 0.0, a1 = 0.0;
        const float* p = gpart + (size_t)ty * a.C + c;
        int i = 0;
        for (; i + 1 < groups; i += 2) { a0 += (double)ld_agent(p + (size_t)i * 2 * a.C); a1 += (double)ld_agent(p + (size_t)(i + 1) * 2 * a.C); }
        if (i < groups) a0 += (double)ld_agent(p + (size_t)i * 2 * a.C);
        red[threadIdx.x][0] = a0 + a1;
    }
    __syncthreads();
    if (ty == 0 && c < a.C)
        bn_bwd_constants(a.cst, a.C, c, red[tx][0], red[TX + tx][0], (double)a.R, a.dgamma, a.dbeta, a.dbias);
    if (threadIdx.x == 0) st_agent(&t1[FL_MAXG], 0);
}

static int g_finish_launch = 1;
// ---- host ------------------------------------------------------------------------------------------------------------
static inline int rows_grid(int mtiles, int xtiles) {
    const int want = std::max(1, 2048 / std::max(1, xtiles));
    return std::max(1, std::min({mtiles, want, FL_MAXY}));
}
static inline bool vec_ok(const void* p, long long ld) { return aligned16(p) && ld % 4 == 0; }
// the split-K kernels: fewer than two 64 x 64 (128 x 32 for narrow outputs) tiles per CU and a contraction long enough for
// the serial matrix work to dominate (pcf_hip_set_flin_split_k: 0 / 1 force them off / on where they apply, -1 = this rule)
static int g_split_k = -1;
static inline bool split_k_pays(long long M, int N, int K) {
    const int force = g_split_k;
    if (force == 0) return false;
    const long long tiles = N <= 32 ? (long long)ceil_div(M, 128) : (long long)ceil_div(M, 64) * ceil_div(N, 64);
    if (force == 1) return K >= 64;
    return K >= 128 && tiles <= 512;
}
// row ranges of the weight gradient: ~1024 workgroups overall, at least 4 steps each, at most BW_SPLITS_MAX slabs and 32 MB
// of them.  Layers with at most 32 output channels use 32 x 128 tiles.
static inline bool bw_narrow(int M) { return M <= 32; }
static inline int bw_splits(long long R, int M, int N, long long* rows_per_split) {
    const long long tiles = bw_narrow(M) ? (long long)ceil_div(N, 128) : (long long)ceil_div(N, 64) * ceil_div(M, 64);
    const long long want = (1024 + tiles - 1) / tiles;
    const long long by_rows = std::max<long long>(1, R / (BW_KS * 2));
    const long long by_bytes = std::max<long long>(1, (32ll << 20) / ((long long)M * N * 4));
    long long splits = std::max<long long>(1, std::min<long long>({want, by_rows, by_bytes, (long long)BW_SPLITS_MAX}));
    long long rps = (std::max<long long>(R, 1) + splits - 1) / splits;
    rps = (rps + BW_KS - 1) / BW_KS * BW_KS;
    if (rows_per_split) *rows_per_split = rps;
    return (int)((std::max<long long>(R, 1) + rps - 1) / rps);
}

}  // namespace pcf

extern "C" {

// scratch: column partials ([FL_MAXY][2][N]) or weight slabs ([splits][M][N]); tickets are a separate, persistent,
// zero-initialised int buffer the kernels leave zeroed (pcf_hip_flin_ticket_ints() entries per call)
size_t pcf_hip_flin_workspace_bytes(long long rows, int c_out, int c_in) {
    if (rows < 0 || c_out < 0 || c_in < 0) return 0;
    const size_t partials = (size_t)(pcf::FL_MAXY + pcf::FL_MAXG) * 2 * (size_t)std::max(c_out, c_in) * 4;
    const size_t slabs = (size_t)pcf::bw_splits(rows, std::max(c_out, 1), std::max(c_in, 1), nullptr) * c_out * c_in * 4;
    return std::max(partials, slabs) + 256;
}
int pcf_hip_flin_ticket_ints(void) { return 8192; }
int pcf_hip_set_flin_finish(int separate_launch) {
    pcf::g_finish_launch = separate_launch ? 1 : 0;
    return pcf::ok();
}
int pcf_hip_set_flin_split_k(int mode) {
    if (mode < -1 || mode > 1) return pcf::fail(PCF_E_BADARG, "set_flin_split_k: mode is -1 (automatic), 0 (off) or 1 (on)");
    pcf::g_split_k = mode;
    return pcf::ok();
}

int pcf_hip_flin_forward(const float* A, long long M, int K, const float* pre, int pre_act, float* side, const float* W,
                         const float* bias, int N, float* Z, float* cst, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, float eps, float momentum, void* workspace,
                         size_t workspace_bytes, int* tickets, void* stream) {
    using namespace pcf;
    PCF_REQUIRE(M >= 0 && K >= 1 && N >= 1 && M < (1ll << 31), "flin_forward: bad sizes (M=%lld K=%d N=%d)", M, K, N);
    if (M == 0) return ok();
    PCF_REQUIRE(A && W && Z, "flin_forward: null pointer");
    PCF_REQUIRE(!cst || (gamma && beta && workspace && tickets && aligned16(workspace) &&
                         workspace_bytes >= pcf_hip_flin_workspace_bytes(M, N, K)),
                "flin_forward: statistics need gamma, beta, a workspace and tickets");
    FwdArgs g{};
    g.A = A; g.pre = pre; g.pre_act = pre_act; g.side = side; g.W = W; g.bias = bias; g.Z = Z;
    g.M = (int)M; g.N = N; g.K = K; g.part = static_cast<float*>(workspace); g.ticket = tickets; g.cst = cst;
    g.gamma = gamma; g.beta = beta; g.running_mean = running_mean; g.running_var = running_var; g.eps = eps; g.momentum = momentum;
    const bool vec = vec_ok(A, K) && vec_ok(W, K) && (!side || vec_ok(side, K)) && (!pre || vec_ok(pre, K));
    hipStream_t s = (hipStream_t)stream;
    const bool finish = cst && g_finish_launch;      // statistics combined by a launch of their own
    if (finish) g.ticket = nullptr;
    int nparts = 0;
    if (vec && split_k_pays(M, N, K)) {
        const int xt = ceil_div(N, 32);
        PCF_REQUIRE(xt * (FL_MAXG + 1) <= pcf_hip_flin_ticket_ints(), "flin_forward: too many output channels");
        nparts = rows_grid(ceil_div(M, 32), xt);
        hipLaunchKernelGGL(flin_fwd_sk_kernel, dim3(xt, nparts), dim3(BLOCK), 0, s, g);
    } else if (N <= 32) {
        dim3 grid(1, rows_grid(ceil_div(M, 128), 1));
        nparts = grid.y;
        if (vec) hipLaunchKernelGGL((flin_fwd_kernel<4, 1, true>), grid, dim3(BLOCK), 0, s, g);
        else hipLaunchKernelGGL((flin_fwd_kernel<4, 1, false>), grid, dim3(BLOCK), 0, s, g);
    } else {
        const int xt = ceil_div(N, 64);
        PCF_REQUIRE(xt * (FL_MAXG + 1) <= pcf_hip_flin_ticket_ints(), "flin_forward: too many output channels");
        dim3 grid(xt, rows_grid(ceil_div(M, 64), xt));
        nparts = grid.y;
        if (vec) hipLaunchKernelGGL((flin_fwd_kernel<2, 2, true>), grid, dim3(BLOCK), 0, s, g);
        else hipLaunchKernelGGL((flin_fwd_kernel<2, 2, false>), grid, dim3(BLOCK), 0, s, g);
    }
    if (int e = check_launch("flin_fwd_kernel")) return e;
    if (!finish) return ok();
    FinishArgs f{};
    f.part = g.part; f.nparts = nparts; f.N = N; f.R = M; f.cst = cst; f.gamma = gamma; f.beta = beta;
    f.running_mean = running_mean; f.running_var = running_var; f.eps = eps; f.momentum = momentum;
    return launch_finish<0>(f, s);
}

int pcf_hip_flin_backward_input(const float* dy, const float* z, const float* cst, int act, long long M, int K, const float* W,
                                int N, const float* add, float* dx, const float* zp, float* cstp, int actp, float* dgamma_p,
                                float* dbeta_p, float* dbias_p, void* workspace, size_t workspace_bytes, int* tickets, void* stream) {
    using namespace pcf;
    PCF_REQUIRE(M >= 0 && K >= 1 && N >= 1 && M < (1ll << 31), "flin_backward_input: bad sizes");
    if (M == 0) return ok();
    PCF_REQUIRE(dy && z && cst && W && dx, "flin_backward_input: null pointer");
    PCF_REQUIRE(!cstp || (zp && workspace && tickets && aligned16(workspace) && workspace_bytes >= pcf_hip_flin_workspace_bytes(M, K, N)),
                "flin_backward_input: producer statistics need its raw output, a workspace and tickets");
    BwdInArgs g{};
    g.dy = dy; g.z = z; g.cst = cst; g.act = act; g.W = W; g.add = add; g.dx = dx; g.zp = zp; g.cstp = cstp; g.actp = actp;
    g.dgamma_p = dgamma_p; g.dbeta_p = dbeta_p; g.dbias_p = dbias_p; g.part = static_cast<float*>(workspace); g.ticket = tickets;
    g.M = (int)M; g.N = N; g.K = K;
    const bool vec = vec_ok(dy, K) && vec_ok(z, K) && vec_ok(W, N) && vec_ok(cst, K);
    hipStream_t s = (hipStream_t)stream;
    const bool finish = cstp && g_finish_launch;
    if (finish) g.ticket = nullptr;
    int nparts = 0;
    if (vec && split_k_pays(M, N, K)) {
        const int xt = ceil_div(N, 32);
        PCF_REQUIRE(xt * (FL_MAXG + 1) <= pcf_hip_flin_ticket_ints(), "flin_backward_input: too many input channels");
        nparts = rows_grid(ceil_div(M, 32), xt);
        hipLaunchKernelGGL(flin_bwd_in_sk_kernel, dim3(xt, nparts), dim3(BLOCK), 0, s, g);
    } else if (N <= 32) {
        dim3 grid(1, rows_grid(ceil_div(M, 128), 1));
        nparts = grid.y;
        if (vec) hipLaunchKernelGGL((flin_bwd_in_kernel<4, 1, true>), grid, dim3(BLOCK), 0, s, g);
        else hipLaunchKernelGGL((flin_bwd_in_kernel<4, 1, false>), grid, dim3(BLOCK), 0, s, g);
    } else {
        const int xt = ceil_div(N, 64);
        PCF_REQUIRE(xt * (FL_MAXG + 1) <= pcf_hip_flin_ticket_ints(), "flin_backward_input: too many input channels");
        dim3 grid(xt, rows_grid(ceil_div(M, 64), xt));
        nparts = grid.y;
        if (vec) hipLaunchKernelGGL((flin_bwd_in_kernel<2, 2, true>), grid, dim3(BLOCK), 0, s, g);
        else hipLaunchKernelGGL((flin_bwd_in_kernel<2, 2, false>), grid, dim3(BLOCK), 0, s, g);
    }
    if (int e = check_launch("flin_bwd_in_kernel")) return e;
    if (!finish) return ok();
    FinishArgs f{};
    f.part = g.part; f.nparts = nparts; f.N = N; f.R = M; f.cst = cstp; f.dgamma = dgamma_p; f.dbeta = dbeta_p; f.dbias = dbias_p;
    return launch_finish<1>(f, s);
}

int pcf_hip_flin_backward_weight_splits(long long R, int M, int N) {
    if (R < 0 || M < 1 || N < 1) return 0;
    return pcf::bw_splits(R, M, N, nullptr);
}

int pcf_hip_flin_backward_weight_slabs(const float* dy, const float* z, const float* cst, int act, const float* xin, const float* pre,
                                       int pre_act, long long R, int M, int N, float* slabs, size_t slabs_bytes, void* stream) {
    using namespace pcf;
    PCF_REQUIRE(R >= 1 && M >= 1 && N >= 1, "flin_backward_weight_slabs: bad sizes");
    long long rps = 0;
    const int splits = bw_splits(R, M, N, &rps);
    PCF_REQUIRE(dy && z && cst && xin && slabs && aligned16(slabs) && slabs_bytes >= (size_t)splits * M * N * 4,
                "flin_backward_weight_slabs: null pointer or small slab buffer");
    const bool narrow = bw_narrow(M);
    const int xt = narrow ? ceil_div(N, 128) : ceil_div(N, 64), yt = narrow ? 1 : ceil_div(M, 64);
    BwdWArgs g{};
    g.dy = dy; g.z = z; g.cst = cst; g.act = act; g.xin = xin; g.pre = pre; g.pre_act = pre_act;
    g.slabs = slabs; g.M = M; g.N = N; g.R = R; g.r_per_split = (int)rps;
    const bool vec = vec_ok(dy, M) && vec_ok(z, M) && vec_ok(xin, N) && vec_ok(cst, M) && (!pre || vec_ok(pre, N));
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(xt, yt, splits);
    if (narrow) {
        if (vec) hipLaunchKernelGGL((flin_bwd_w_kernel<1, 4, true>), grid, dim3(BLOCK), 0, s, g);
        else hipLaunchKernelGGL((flin_bwd_w_kernel<1, 4, false>), grid, dim3(BLOCK), 0, s, g);
    } else {
        if (vec) hipLaunchKernelGGL((flin_bwd_w_kernel<2, 2, true>), grid, dim3(BLOCK), 0, s, g);
        else hipLaunchKernelGGL((flin_bwd_w_kernel<2, 2, false>), grid, dim3(BLOCK), 0, s, g);
    }
    return check_launch("flin_bwd_w_kernel");
}

int pcf_hip_slab_sum_multi(int n, const float* const* slabs, float* const* out, const long long* counts, const int* splits,
                           void* stream) {
    using namespace pcf;
    PCF_REQUIRE(n >= 0 && n <= SSM_MAX, "slab_sum_multi: at most %d lists (got %d)", SSM_MAX, n);
    if (n == 0) return ok();
    PCF_REQUIRE(slabs && out && counts && splits, "slab_sum_multi: null pointer");
    SlabSumArgs a{};
    a.n = n;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        PCF_REQUIRE(slabs[i] && out[i] && counts[i] >= 0 && splits[i] >= 1, "slab_sum_multi: bad list %d", i);
        a.slabs[i] = slabs[i]; a.out[i] = out[i]; a.count[i] = counts[i]; a.splits[i] = splits[i];
        a.block0[i] = blocks;
        blocks += (int)((counts[i] + 63) / 64);
    }
    a.block0[n] = blocks;
    if (blocks == 0) return ok();
    hipLaunchKernelGGL(slab_sum_multi_kernel, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, a);
    return check_launch("slab_sum_multi_kernel");
}

int pcf_hip_bn_backward_stats(const float* dy, const float* z, const float* res, float* cst, int act, long long R, int C,
                              float* g, float* dgamma, float* dbeta, float* dbias, void* workspace, size_t workspace_bytes, int* tickets,
                              void* stream) {
    using namespace pcf;
    PCF_REQUIRE(R >= 0 && C >= 1, "bn_backward_stats: bad sizes");
    if (R == 0) return ok();
    PCF_REQUIRE(dy && z && cst && g && workspace && tickets && aligned16(workspace) &&
                workspace_bytes >= pcf_hip_flin_workspace_bytes(R, C, C), "bn_backward_stats: null pointer or small workspace");
    int tx = 1;
    while (tx < C && tx < 64) tx <<= 1;
    const int chunks = ceil_div(C, tx);
    PCF_REQUIRE(chunks * (FL_MAXG + 1) <= pcf_hip_flin_ticket_ints(), "bn_backward_stats: too many channels");
    const long long want = std::max<long long>(64, 2048 / chunks);
    const int nb = (int)std::max<long long>(1, std::min<long long>({(R + 15) / 16, want, (long long)FL_MAXY}));
    TopArgs a{};
    a.dy = dy; a.z = z; a.res = res; a.cst = cst; a.act = act; a.g = g; a.dgamma = dgamma; a.dbeta = dbeta; a.dbias = dbias;
    a.part = static_cast<float*>(workspace); a.ticket = g_finish_launch ? nullptr : tickets; a.R = R; a.C = C; a.TX = tx;
    a.rows_per_block = (R + nb - 1) / nb;
    hipLaunchKernelGGL(bn_bwd_stats_kernel, dim3(nb, chunks), dim3(BLOCK), 0, (hipStream_t)stream, a);
    if (int e = check_launch("bn_bwd_stats_kernel")) return e;
    if (!g_finish_launch) return ok();
    FinishArgs f{};
    f.part = a.part; f.nparts = nb; f.N = C; f.R = R; f.cst = cst; f.dgamma = dgamma; f.dbeta = dbeta; f.dbias = dbias;
    return launch_finish<1>(f, (hipStream_t)stream);
}

}  // extern "C"
