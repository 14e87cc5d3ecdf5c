// The point-level layers of a PCFLayer as row chains on the matrix cores, for widths whose weights fit the workgroup.
//
//   head  x [R, 16 KI] --unary1: Linear+BN+LeakyReLU--> fx [R, 16 NM] --guidance_unary: Linear+BN--> gx [R, 16 NG]
//                                                                     --Wa (gathered half of the first guidance layer)--> u [R, 8]
//   tail  agg [R, 16 KA] --linear: Linear+BN+ReLU--> y3 [R, 16 NH] --unary2: Linear+BN--> (+ shortcut, LeakyReLU) out [R, 16 NO]
//                                                                                          (layers.py:335, 369, 393-414)
//
// At the BASELINE shape (80k rows) every one of these layers is a 5-80 MB pass that a stand-alone contraction kernel spends
// 15-40 us on, most of it launch / ramp / tail latency (csrc/fused_linear.hip: 19 launches, 435 us per step).  Here the same
// transposed MFMA formulation as the edge graph (edge_chain.hip: v_mfma_f32_16x16x4_f32, 16 rows per tile, lane (p, g)
// holds channels 4g..4g+3 of row p, a layer's accumulator IS the next layer's B operand) runs whole layer chains per pass,
// a pass ending where a BatchNorm's batch statistics are not known yet:
//   head forward   H1: z1 = W1 x, statistics          H2: fx = LeakyReLU(BN1(z1)) stored, statistics of z2 = W2 fx
//                  H3: z2 recomputed from fx, gx = BN2(z2), u = Wa gx stored                 (z2 / gx never touch HBM)
//   head backward  B1: dgx = Wa^T du, BN2-backward sums, dWa        B2: dz2, dW2, dfx = W2^T dz2 + dfx(aggregate), masked by
//                  LeakyReLU'(BN1(z1)) -> g1 stored with its sums   B3: dz1, dW1, dx = W1^T dz1
// Weight fragments (forward and transposed) and BatchNorm constants live in LDS, weight-gradient tiles are accumulated per
// wave as outer products over the tile's 16 rows (operands turned through 16 x 20 LDS tiles as in edge_chain_bwd.hip) and
// reduced by one final launch; statistics are finished by the last workgroup of the pass (agent-scope atomics, no L2
// write-back: flin_common.h).  fp32 throughout, exact products; deterministic.  The same machinery runs the two-layer
// positional-encoding MLPs of the strided / transposed PointConvs over the EDGES (pe_chain_kernel, below).
#include <algorithm>

#include "edge_chain.h"
#include "flin_common.h"

#pragma clang fp contract(fast)

namespace pcf {

constexpr int PT = 20;                   // row stride of the 16 x 16 transposition tiles
constexpr int PC_MAXB = 256;             // workgroups per pass: a wave per SIMD, tile loads hidden by the hand pipelining below
constexpr int PC_GROUP = 16;             // partial lists per first-level group of pass_totals
constexpr int PC_CH = 8;                 // width of u (hidden width of the guidance MLP)
constexpr int PE_MAXB = 2048;            // the edge-row chains (millions of rows, 34-140 registers): up to eight waves per SIMD
// floats of the statistics area of a pass of at most nb workgroups whose widest layer has maxc channels: the workgroups'
// lists + the groups' doubles
static inline size_t pc_part_floats(int maxc, int nb = PC_MAXB) { return (size_t)nb * 2 * maxc + (size_t)(nb / PC_GROUP + 1) * 2 * maxc * 2; }

__device__ __forceinline__ f32x4 v4(float4 v) { return f32x4{v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ float4 f4(f32x4 v) { return make_float4(v[0], v[1], v[2], v[3]); }
template <int CTRL> __device__ __forceinline__ float pc_dpp(float v) {
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xf, 0xf, true));
}
// sum over the 16 lanes of a row (every lane gets it): DPP only
__device__ __forceinline__ float row_sum16(float v) {
    v += pc_dpp<0xB1>(v);
    v += pc_dpp<0x4E>(v);
    v += pc_dpp<0x141>(v);
    v += pc_dpp<0x140>(v);
    return v;
}
// acc += fragment . operand (four contraction steps); fragments are float4 per lane in LDS
__device__ __forceinline__ f32x4 pmm(const float4* wl, int frag, int lane, f32x4 operand, f32x4 acc) {
    const f32x4 w = v4(wl[frag * WAVE + lane]);
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = PCF_MFMA(w[s], operand[s], acc);
    return acc;
}
__device__ __forceinline__ void put_tile(float* buf, f32x4 v, int p, int g) { st4(buf + p * PT + 4 * g, f4(v)); }
// dW tile += sum over the 16 rows of dz[o][row] * in[c][row]
__device__ __forceinline__ f32x4 pouter(const float* dzbuf, const float* inbuf, int p, int g, f32x4 acc) {
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = PCF_MFMA(dzbuf[(4 * s + g) * PT + p], inbuf[(4 * s + g) * PT + p], acc);
    return acc;
}
// channels c0..c0+3 of `row` of an [R, C] matrix (C a multiple of 4); zero for rows beyond R
__device__ __forceinline__ f32x4 ld_row4(const float* base, long long row, long long R, int C, int c0) {
    if (row >= R) return f32x4{0.f, 0.f, 0.f, 0.f};
    return v4(ld4(base + (size_t)row * C + c0));
}
__device__ __forceinline__ void st_row4(float* base, long long row, long long R, int C, int c0, f32x4 v) {
    if (row < R) st4(base + (size_t)row * C + c0, f4(v));
}
__device__ __forceinline__ f32x4 leaky4(f32x4 v) {
    return f32x4{v[0] > 0.f ? v[0] : 0.1f * v[0], v[1] > 0.f ? v[1] : 0.1f * v[1], v[2] > 0.f ? v[2] : 0.1f * v[2], v[3] > 0.f ? v[3] : 0.1f * v[3]};
}
__device__ __forceinline__ f32x4 dleaky4(f32x4 d, f32x4 pre) {
    return f32x4{pre[0] > 0.f ? d[0] : 0.1f * d[0], pre[1] > 0.f ? d[1] : 0.1f * d[1], pre[2] > 0.f ? d[2] : 0.1f * d[2], pre[3] > 0.f ? d[3] : 0.1f * d[3]};
}
__device__ __forceinline__ f32x4 relu4p(f32x4 v) { return f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)}; }
__device__ __forceinline__ f32x4 drelu4(f32x4 d, f32x4 pre) {
    return f32x4{pre[0] > 0.f ? d[0] : 0.f, pre[1] > 0.f ? d[1] : 0.f, pre[2] > 0.f ? d[2] : 0.f, pre[3] > 0.f ? d[3] : 0.f};
}
// four per-channel constants of row `which` of a [6][C] record, channels c0..c0+3
__device__ __forceinline__ f32x4 cst_row4(const float* cst, int C, int which, int c0) { return v4(ld4(cst + which * C + c0)); }

// Column sums of a pass: every lane holds sums of its four channels per 16-channel tile; fold the 16 row lanes, the four
// waves (fixed order) and publish the workgroup's [2][NC] list.  The lists are combined by the workgroup that arrives last
// (tickets; hand-over rules in flin_common.h): up to PC_MAXB lists in ONE level with 16 loads in flight per lane, more (the
// edge-row chains: up to 2048) in two levels -- the PC_GROUP lists of a group, then the group sums.  Fixed summation order,
// double from the first combination on.  True in the one workgroup that ends up with the totals in tot[2 NC].  ticket == null:
// the lists are left for flin_finish_kernel (a launch of its own; pcf_hip_set_row_chain_finish).
template <int NT>
__device__ __forceinline__ bool pass_totals(const f32x4 (&s1)[NT], const f32x4 (&s2)[NT], float* part, int* ticket, double* tot) {
    constexpr int NC = 16 * NT, NV = 2 * NC;
    constexpr int SL = BLOCK / NV >= 1 ? BLOCK / NV : 1;
    static_assert(NV <= BLOCK, "one thread per column sum");
    __shared__ float wsum[NWAVE][NV];
    __shared__ double red[BLOCK];
    __shared__ int s_last;
    double* gpart = reinterpret_cast<double*>(part + (size_t)gridDim.x * NV);
    const int lane = lane_id(), wave = wave_id(), p = lane & 15, g = lane >> 4;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float a = row_sum16(s1[t][r]), b = row_sum16(s2[t][r]);
            if (p == 0) { wsum[wave][16 * t + 4 * g + r] = a; wsum[wave][NC + 16 * t + 4 * g + r] = b; }
        }
    __syncthreads();
    for (int v = threadIdx.x; v < NV; v += BLOCK) {
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < NWAVE; ++w) a += wsum[w][v];
        if (!ticket) part[(size_t)blockIdx.x * NV + v] = a;      // combined by flin_finish_kernel (its [nparts][2][NC] layout)
        else st_agent(part + (size_t)blockIdx.x * NV + v, a);
    }
    if (!ticket) return false;
    const int v = threadIdx.x % NV, sl = threadIdx.x / NV;
    if (gridDim.x <= PC_MAXB) {
        // up to 256 lists: ONE level, the last workgroup reads them all with 16 loads in flight per lane (a second ticket round
        // costs more than the longer walk)
        publish();
        __syncthreads();
        if (threadIdx.x == 0) s_last = (atomicAdd(ticket, 1) == (int)gridDim.x - 1) ? 1 : 0;
        __syncthreads();
        if (!s_last) return false;
        observe();
        double a = 0.0;
        if (sl < SL) {
            int q = sl;
            for (; q + 15 * SL < (int)gridDim.x; q += 16 * SL) {
                float t[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) t[i] = ld_agent(part + (size_t)(q + i * SL) * NV + v);
#pragma unroll
                for (int i = 0; i < 16; ++i) a += (double)t[i];
            }
            for (; q < (int)gridDim.x; q += SL) a += (double)ld_agent(part + (size_t)q * NV + v);
        }
        red[threadIdx.x] = a;
        __syncthreads();
        if (threadIdx.x < NV) {
            double b = 0.0;
#pragma unroll
            for (int s = 0; s < SL; ++s) b += red[s * NV + threadIdx.x];
            tot[threadIdx.x] = b;
        }
        if (threadIdx.x == 0) st_agent(ticket, 0);
        __syncthreads();
        return true;
    }
    const int grp = blockIdx.x / PC_GROUP, ngroups = ((int)gridDim.x + PC_GROUP - 1) / PC_GROUP;
    const int g0 = grp * PC_GROUP, gsize = min(PC_GROUP, (int)gridDim.x - g0);
    publish();
    __syncthreads();
    if (threadIdx.x == 0) s_last = (atomicAdd(ticket + 1 + grp, 1) == gsize - 1) ? 1 : 0;
    __syncthreads();
    if (!s_last) return false;
    observe();
    {
        float acc[PC_GROUP / SL + 1];
        int n = 0;
        if (sl < SL)
#pragma unroll
            for (int q = sl; q < PC_GROUP; q += SL) acc[n++] = q < gsize ? ld_agent(part + (size_t)(g0 + q) * NV + v) : 0.f;
        double a = 0.0;
        for (int i = 0; i < n; ++i) a += (double)acc[i];
        red[threadIdx.x] = a;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double a = 0.0;
#pragma unroll
        for (int s = 0; s < SL; ++s) a += red[s * NV + threadIdx.x];
        st_agent(gpart + (size_t)grp * NV + threadIdx.x, a);
    }
    if (threadIdx.x == 0) st_agent(ticket + 1 + grp, 0);
    publish();
    __syncthreads();
    if (threadIdx.x == 0) s_last = (atomicAdd(ticket, 1) == ngroups - 1) ? 1 : 0;
    __syncthreads();
    if (!s_last) return false;
    observe();
    {
        double a = 0.0;
        if (sl < SL)
            for (int q = sl; q < ngroups; q += SL)
                a += ld_agent(gpart + (size_t)q * NV + v);
        red[threadIdx.x] = a;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double a = 0.0;
#pragma unroll
        for (int s = 0; s < SL; ++s) a += red[s * NV + threadIdx.x];
        tot[threadIdx.x] = a;
    }
    if (threadIdx.x == 0) st_agent(ticket, 0);
    __syncthreads();
    return true;
}

// per-wave weight-gradient tiles -> workgroup sum (fixed wave order) -> this workgroup's slot of the partial list
template <int NTILE>
__device__ __forceinline__ void flush_dw(const f32x4 (&acc)[NTILE], float* red, float* part) {
    const int lane = lane_id(), wave = wave_id(), p = lane & 15, g = lane >> 4;
    for (int t = threadIdx.x; t < NTILE * 256; t += BLOCK) red[t] = 0.f;
    __syncthreads();
    for (int wv = 0; wv < NWAVE; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int i = 0; i < NTILE; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[i * 256 + (4 * g + r) * 16 + p] += acc[i][r];
        }
        __syncthreads();
    }
    float* outp = part + (size_t)blockIdx.x * NTILE * 256;
    for (int t = threadIdx.x; t < NTILE * 256; t += BLOCK) outp[t] = red[t];
}

// statistics of a pass combined by the last workgroup (0, default) or by a launch of their own (1).  Measured on replayed
// graphs: tickets 0.99-1.00 ms per layer step against 1.01 ms, 20.9 against 21.25 ms per 10cm-lite iteration -- the opposite of
// the contraction kernels of fused_linear.hip, whose two-level hand-over over up to 1024 lists is dearer.
static int g_pc_finish = 0;
static inline int pc_finish_fwd(const float* part, int nparts, int C, long long R, float* cst, const float* gamma, const float* beta,
                                float* rm, float* rv, float eps, float mom, hipStream_t s) {
    FinishArgs f{};
    f.part = part; f.nparts = nparts; f.N = C; f.R = R; f.cst = cst; f.gamma = gamma; f.beta = beta; f.running_mean = rm; f.running_var = rv;
    f.eps = eps; f.momentum = mom;
    return launch_finish<0>(f, s);
}
static inline int pc_finish_bwd(const float* part, int nparts, int C, long long R, float* cst, float* dgamma, float* dbeta, float* dbias,
                                hipStream_t s) {
    FinishArgs f{};
    f.part = part; f.nparts = nparts; f.N = C; f.R = R; f.cst = cst; f.dgamma = dgamma; f.dbeta = dbeta; f.dbias = dbias;
    return launch_finish<1>(f, s);
}
// ---- head ---------------------------------------------------------------------------------------------------------
struct HeadArgs {
    const float* x; float* z1; float* fx; float* u;              // [R,16KI] in; [R,16NM] raw unary1 output; fx; [R,8]
    const float* W1; const float* b1; const float* W2; const float* b2; const float* Wa;
    float* cst1; float* cst2;                                    // records [6][16NM], [6][16NG]
    const float* gamma1; const float* beta1; float* rmean1; float* rvar1;
    const float* gamma2; const float* beta2; float* rmean2; float* rvar2;
    float eps, mom1, mom2;
    // backward
    const float* dfx; const float* du; float* g1; float* dx;     // [R,16NM], [R,8], [R,16NM] scratch, [R,16KI]
    float* dgamma1; float* dbeta1; float* db1; float* dgamma2; float* dbeta2; float* db2;
    float* part; int* ticket;                                    // statistics partials, one ticket
    float* pdwa; float* pdw2; float* pdw1;                       // weight-gradient partial tiles [blocks][tiles][256]
    long long R;
};

template <int KI, int NM, int NG>
struct HeadFrags {
    static constexpr int W1 = 0;                         // (nm, kt): A[o = 16nm + p][c = 16kt + 4g + s]
    static constexpr int W2 = W1 + NM * KI;              // (ng, nm)
    static constexpr int WA = W2 + NG * NM;              // (ng):     A[o = p < 8][c = 16ng + 4g + s]
    static constexpr int WAT = WA + NG;                  // (ng):     A[c = 16ng + p][o = 4g + s < 8]
    static constexpr int W2T = WAT + NG;                 // (nm, ng): A[c = 16nm + p][o = 16ng + 4g + s]
    static constexpr int W1T = W2T + NM * NG;            // (kt, nm): A[c = 16kt + p][o = 16nm + 4g + s]
    static constexpr int COUNT = W1T + KI * NM;
};

template <int KI, int NM, int NG>
__device__ __forceinline__ void stage_head(const HeadArgs& a, float4* wl, bool transposed) {
    using F = HeadFrags<KI, NM, NG>;
    const int n = transposed ? F::COUNT : F::WAT;
    for (int t = threadIdx.x; t < n * WAVE; t += BLOCK) {
        const int fr = t / WAVE, l = t % WAVE, p = l & 15, g = l >> 4;
        float v[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = 4 * g + s;
            if (fr < F::W2) { const int nm = fr / KI, kt = fr % KI; v[s] = wfrag(a.W1, 16 * NM, 16 * KI, 16 * nm + p, 16 * kt + k); }
            else if (fr < F::WA) { const int q = fr - F::W2, ng = q / NM, nm = q % NM; v[s] = wfrag(a.W2, 16 * NG, 16 * NM, 16 * ng + p, 16 * nm + k); }
            else if (fr < F::WAT) { const int ng = fr - F::WA; v[s] = wfrag(a.Wa, PC_CH, 16 * NG, p, 16 * ng + k); }
            else if (fr < F::W2T) { const int ng = fr - F::WAT; v[s] = wfrag(a.Wa, PC_CH, 16 * NG, k, 16 * ng + p); }
            else if (fr < F::W1T) { const int q = fr - F::W2T, nm = q / NG, ng = q % NG; v[s] = wfrag(a.W2, 16 * NG, 16 * NM, 16 * ng + k, 16 * nm + p); }
            else { const int q = fr - F::W1T, kt = q / NM, nm = q % NM; v[s] = wfrag(a.W1, 16 * NM, 16 * KI, 16 * nm + k, 16 * kt + p); }
        }
        wl[t] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// PASS 1: z1 + statistics | 2: fx + statistics of z2 | 3: u
template <int KI, int NM, int NG, int PASS>
__global__ __launch_bounds__(BLOCK) void head_fwd_kernel(const HeadArgs a) {
    using F = HeadFrags<KI, NM, NG>;
    __shared__ float4 wl[F::WAT * WAVE];
    __shared__ double tot[2 * 16 * (NM > NG ? NM : NG)];
    const int lane = lane_id(), p = lane & 15, g = lane >> 4;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    constexpr int CM = 16 * NM, CG = 16 * NG;
    constexpr int NIN = PASS == 1 ? KI : NM;                 // 16-channel tiles of the row a pass reads: x | z1 | fx
    const float* in = PASS == 1 ? a.x : PASS == 2 ? a.z1 : a.fx;
    const long long ntiles = (a.R + 15) / 16, stride = (long long)gridDim.x * NWAVE;
    long long t = (long long)blockIdx.x * NWAVE + wave_id();
    f32x4 nxt[NIN];                                          // software pipeline: see the tail kernels
    if (t < ntiles) {
#pragma unroll
        for (int i = 0; i < NIN; ++i) nxt[i] = ld_row4(in, t * 16 + p, a.R, 16 * NIN, 16 * i + 4 * g);
    }
    stage_head<KI, NM, NG>(a, wl, false);
    constexpr int NS = PASS == 1 ? NM : NG;
    f32x4 s1[NS], s2[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) { s1[i] = zero4; s2[i] = zero4; }
    f32x4 bias1[NM], bias2[NG], sc1[NM], sh1[NM], sc2[NG], sh2[NG];
#pragma unroll
    for (int nm = 0; nm < NM; ++nm) {
        bias1[nm] = v4(ld4(a.b1 + 16 * nm + 4 * g));
        if (PASS >= 2) { sc1[nm] = cst_row4(a.cst1, CM, 0, 16 * nm + 4 * g); sh1[nm] = cst_row4(a.cst1, CM, 1, 16 * nm + 4 * g); }
    }
#pragma unroll
    for (int ng = 0; ng < NG; ++ng) {
        bias2[ng] = v4(ld4(a.b2 + 16 * ng + 4 * g));
        if (PASS == 3) { sc2[ng] = cst_row4(a.cst2, CG, 0, 16 * ng + 4 * g); sh2[ng] = cst_row4(a.cst2, CG, 1, 16 * ng + 4 * g); }
    }
    __syncthreads();
    for (; t < ntiles; t += stride) {
        const long long row = t * 16 + p;
        const float valid = row < a.R ? 1.f : 0.f;
        f32x4 x[NIN];
#pragma unroll
        for (int i = 0; i < NIN; ++i) x[i] = nxt[i];
        if (t + stride < ntiles) {
#pragma unroll
            for (int i = 0; i < NIN; ++i) nxt[i] = ld_row4(in, (t + stride) * 16 + p, a.R, 16 * NIN, 16 * i + 4 * g);
        }
        f32x4 y1[NM];
        if (PASS == 1) {
#pragma unroll
            for (int nm = 0; nm < NM; ++nm) {
                f32x4 z = zero4;
#pragma unroll
                for (int kt = 0; kt < KI; ++kt) z = pmm(wl, F::W1 + nm * KI + kt, lane, x[kt], z);
                z += bias1[nm];
                st_row4(a.z1, row, a.R, CM, 16 * nm + 4 * g, z);
                s1[nm] += z * valid; s2[nm] += z * z * valid;
            }
            continue;
        }
        if (PASS == 2) {
#pragma unroll
            for (int nm = 0; nm < NM; ++nm) {
                y1[nm] = leaky4(x[nm] * sc1[nm] + sh1[nm]);
                st_row4(a.fx, row, a.R, CM, 16 * nm + 4 * g, y1[nm]);
            }
        } else {
#pragma unroll
            for (int nm = 0; nm < NM; ++nm) y1[nm] = x[nm];
        }
        f32x4 uacc = zero4;
#pragma unroll
        for (int ng = 0; ng < NG; ++ng) {
            f32x4 z = zero4;
#pragma unroll
            for (int nm = 0; nm < NM; ++nm) z = pmm(wl, F::W2 + ng * NM + nm, lane, y1[nm], z);
            z += bias2[ng];
            if (PASS == 2) { s1[ng] += z * valid; s2[ng] += z * z * valid; }
            else uacc = pmm(wl, F::WA + ng, lane, z * sc2[ng] + sh2[ng], uacc);
        }
        if (PASS == 3 && g < 2) st_row4(a.u, row, a.R, PC_CH, 4 * g, uacc);
    }
    if (PASS == 3) return;
    if (!pass_totals<NS>(s1, s2, a.part, a.ticket, tot)) return;
    constexpr int NC = 16 * NS;
    if (threadIdx.x < NC) {
        if (PASS == 1) bn_fwd_constants(a.cst1, NC, threadIdx.x, tot[threadIdx.x], tot[NC + threadIdx.x], (double)a.R, a.gamma1, a.beta1, a.rmean1, a.rvar1, a.eps, a.mom1);
        else bn_fwd_constants(a.cst2, NC, threadIdx.x, tot[threadIdx.x], tot[NC + threadIdx.x], (double)a.R, a.gamma2, a.beta2, a.rmean2, a.rvar2, a.eps, a.mom2);
    }
}

// PASS 1: dgx = Wa^T du; sums of layer 2 (no activation: g = dgx); dWa tiles.
// PASS 2: dz2; dW2 tiles; dfx = W2^T dz2 + dfx(aggregate); g1 = dfx * LeakyReLU'(BN1(z1)) stored; sums of layer 1.
// PASS 3: dz1 from (g1, z1); dW1 tiles; dx = W1^T dz1.
template <int KI, int NM, int NG, int PASS>
__global__ __launch_bounds__(BLOCK) void head_bwd_kernel(const HeadArgs a) {
    using F = HeadFrags<KI, NM, NG>;
    constexpr int CI = 16 * KI, CM = 16 * NM, CG = 16 * NG;
    constexpr int NDW = PASS == 1 ? NG : PASS == 2 ? NG * NM : NM * KI;
    constexpr int NTB = PASS == 1 ? 1 + NG : PASS == 2 ? NG + NM : NM + KI;          // transposition tiles per wave
    __shared__ float4 wl[F::COUNT * WAVE];
    __shared__ __align__(16) float tbuf[NWAVE * NTB * 16 * PT];
    __shared__ float red[NDW * 256];
    __shared__ double tot[2 * 16 * (NM > NG ? NM : NG)];
    const int lane = lane_id(), wave = wave_id(), p = lane & 15, g = lane >> 4;
    float* tb = tbuf + wave * NTB * 16 * PT;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    constexpr int NIN = PASS == 1 ? 1 + NM : PASS == 2 ? 1 + 3 * NM : 2 * NM + KI;        // f32x4 a lane loads per tile
    const long long ntiles = (a.R + 15) / 16, stride = (long long)gridDim.x * NWAVE;
    long long t = (long long)blockIdx.x * NWAVE + wave_id();
    // PASS 1: du, fx | 2: du, fx, dfx, z1 | 3: g1, z1, x
    auto load_tile = [&](long long tt, f32x4 (&v)[NIN]) {
        const long long row = tt * 16 + p;
        if (PASS == 3) {
#pragma unroll
            for (int nm = 0; nm < NM; ++nm) {
                v[nm] = ld_row4(a.g1, row, a.R, CM, 16 * nm + 4 * g);
                v[NM + nm] = ld_row4(a.z1, row, a.R, CM, 16 * nm + 4 * g);
            }
#pragma unroll
            for (int kt = 0; kt < KI; ++kt) v[2 * NM + kt] = ld_row4(a.x, row, a.R, CI, 16 * kt + 4 * g);
        } else {
            v[0] = g < 2 ? ld_row4(a.du, row, a.R, PC_CH, 4 * g) : zero4;
#pragma unroll
            for (int nm = 0; nm < NM; ++nm) {
                v[1 + nm] = ld_row4(a.fx, row, a.R, CM, 16 * nm + 4 * g);
                if (PASS == 2) {
                    v[1 + NM + nm] = a.dfx ? ld_row4(a.dfx, row, a.R, CM, 16 * nm + 4 * g) : zero4;
                    v[1 + 2 * NM + nm] = ld_row4(a.z1, row, a.R, CM, 16 * nm + 4 * g);
                }
            }
        }
    };
    f32x4 nxt[NIN];                                          // software pipeline: see the tail kernels
    if (t < ntiles) load_tile(t, nxt);
    stage_head<KI, NM, NG>(a, wl, true);
    constexpr int NS = PASS == 1 ? NG : NM;
    f32x4 s1[NS], s2[NS], dw[NDW];
#pragma unroll
    for (int i = 0; i < NS; ++i) { s1[i] = zero4; s2[i] = zero4; }
#pragma unroll
    for (int i = 0; i < NDW; ++i) dw[i] = zero4;
    f32x4 bias2[NG], sc2[NG], sh2[NG], d12[NG], d02[NG], sc1[NM], sh1[NM], d11[NM], d01[NM];
#pragma unroll
    for (int ng = 0; ng < NG; ++ng) {
        bias2[ng] = v4(ld4(a.b2 + 16 * ng + 4 * g));
        sc2[ng] = cst_row4(a.cst2, CG, 0, 16 * ng + 4 * g); sh2[ng] = cst_row4(a.cst2, CG, 1, 16 * ng + 4 * g);
        if (PASS == 2) { d12[ng] = cst_row4(a.cst2, CG, 4, 16 * ng + 4 * g); d02[ng] = cst_row4(a.cst2, CG, 5, 16 * ng + 4 * g); }
    }
#pragma unroll
    for (int nm = 0; nm < NM; ++nm) {
        sc1[nm] = cst_row4(a.cst1, CM, 0, 16 * nm + 4 * g); sh1[nm] = cst_row4(a.cst1, CM, 1, 16 * nm + 4 * g);
        if (PASS == 3) { d11[nm] = cst_row4(a.cst1, CM, 4, 16 * nm + 4 * g); d01[nm] = cst_row4(a.cst1, CM, 5, 16 * nm + 4 * g); }
    }
    __syncthreads();
    for (; t < ntiles; t += stride) {
        const long long row = t * 16 + p;
        const float valid = row < a.R ? 1.f : 0.f;
        f32x4 v[NIN];
#pragma unroll
        for (int i = 0; i < NIN; ++i) v[i] = nxt[i];
        if (t + stride < ntiles) load_tile(t + stride, nxt);
        if (PASS == 3) {
            f32x4 dz1[NM], x[KI];
#pragma unroll
            for (int nm = 0; nm < NM; ++nm) {
                const f32x4 gg = v[nm], z = v[NM + nm];
                dz1[nm] = (gg * sc1[nm] + (z * d11[nm] + d01[nm])) * valid;
                put_tile(tb + nm * 16 * PT, dz1[nm], p, g);
            }
#pragma unroll
            for (int kt = 0; kt < KI; ++kt) {
                x[kt] = v[2 * NM + kt];
                put_tile(tb + (NM + kt) * 16 * PT, x[kt], p, g);
            }
#pragma unroll
            for (int nm = 0; nm < NM; ++nm)
#pragma unroll
                for (int kt = 0; kt < KI; ++kt) dw[nm * KI + kt] = pouter(tb + nm * 16 * PT, tb + (NM + kt) * 16 * PT, p, g, dw[nm * KI + kt]);
#pragma unroll
            for (int kt = 0; kt < KI; ++kt) {
                f32x4 d = zero4;
#pragma unroll
                for (int nm = 0; nm < NM; ++nm) d = pmm(wl, F::W1T + kt * NM + nm, lane, dz1[nm], d);
                st_row4(a.dx, row, a.R, CI, 16 * kt + 4 * g, d);
            }
            continue;
        }
        // passes 1 and 2: du -> dgx; z2 recomputed from fx
        const f32x4 du = v[0];
        f32x4 fx[NM], dgx[NG], z2[NG];
#pragma unroll
        for (int nm = 0; nm < NM; ++nm) fx[nm] = v[1 + nm];
#pragma unroll
        for (int ng = 0; ng < NG; ++ng) {
            dgx[ng] = pmm(wl, F::WAT + ng, lane, du, zero4);
            f32x4 z = zero4;
#pragma unroll
            for (int nm = 0; nm < NM; ++nm) z = pmm(wl, F::W2 + ng * NM + nm, lane, fx[nm], z);
            z2[ng] = z + bias2[ng];
        }
        if (PASS == 1) {
            put_tile(tb, du, p, g);
#pragma unroll
            for (int ng = 0; ng < NG; ++ng) {
                s1[ng] += dgx[ng] * valid; s2[ng] += dgx[ng] * z2[ng] * valid;
                put_tile(tb + (1 + ng) * 16 * PT, (z2[ng] * sc2[ng] + sh2[ng]) * valid, p, g);       // gx
                dw[ng] = pouter(tb, tb + (1 + ng) * 16 * PT, p, g, dw[ng]);
            }
            continue;
        }
        f32x4 dz2[NG];
#pragma unroll
        for (int ng = 0; ng < NG; ++ng) {
            dz2[ng] = (dgx[ng] * sc2[ng] + (z2[ng] * d12[ng] + d02[ng])) * valid;
            put_tile(tb + ng * 16 * PT, dz2[ng], p, g);
        }
#pragma unroll
        for (int nm = 0; nm < NM; ++nm) put_tile(tb + (NG + nm) * 16 * PT, fx[nm], p, g);
#pragma unroll
        for (int ng = 0; ng < NG; ++ng)
#pragma unroll
            for (int nm = 0; nm < NM; ++nm) dw[ng * NM + nm] = pouter(tb + ng * 16 * PT, tb + (NG + nm) * 16 * PT, p, g, dw[ng * NM + nm]);
#pragma unroll
        for (int nm = 0; nm < NM; ++nm) {
            f32x4 d = PASS == 2 ? v[1 + NM + nm] : zero4;
#pragma unroll
            for (int ng = 0; ng < NG; ++ng) d = pmm(wl, F::W2T + nm * NG + ng, lane, dz2[ng], d);
            const f32x4 z = PASS == 2 ? v[1 + 2 * NM + nm] : zero4;
            const f32x4 gg = dleaky4(d, z * sc1[nm] + sh1[nm]) * valid;
            st_row4(a.g1, row, a.R, CM, 16 * nm + 4 * g, gg);
            s1[nm] += gg; s2[nm] += gg * z;
        }
    }
    flush_dw<NDW>(dw, red, PASS == 1 ? a.pdwa : PASS == 2 ? a.pdw2 : a.pdw1);
    if (PASS == 3) return;
    if (!pass_totals<NS>(s1, s2, a.part, a.ticket, tot)) return;
    constexpr int NC = 16 * NS;
    if (threadIdx.x < NC) {
        if (PASS == 1) bn_bwd_constants(a.cst2, NC, threadIdx.x, tot[threadIdx.x], tot[NC + threadIdx.x], (double)a.R, a.dgamma2, a.dbeta2, a.db2);
        else bn_bwd_constants(a.cst1, NC, threadIdx.x, tot[threadIdx.x], tot[NC + threadIdx.x], (double)a.R, a.dgamma1, a.dbeta1, a.db1);
    }
}

// ---- tail ---------------------------------------------------------------------------------------------------------
//   T1: z3 = W3 agg + b3 stored, statistics          T2: z4 = W4 ReLU(BN3(z3)) + b4 stored, statistics
//   (out = LeakyReLU(BN4(z4) + shortcut) is the column-BatchNorm kernel of bnact.hip)
//   P1: g4 = dout * LeakyReLU'(BN4(z4) + shortcut) stored (also the shortcut's gradient), sums of layer 4
//   P2: dz4, dW4 tiles, g3 = (W4^T dz4) * ReLU'(BN3(z3)) stored, sums of layer 3
//   P3: dz3, dagg = W3^T dz3                              (dW3, 2 x 16 tiles, is left to flin_bwd_w_kernel)
struct TailArgs {
    const float* agg; float* z3; float* z4;                        // [R,16KA] in; raw outputs [R,16NH], [R,16NO]
    const float* W3; const float* b3; const float* W4; const float* b4;
    float* cst3; float* cst4;
    const float* gamma3; const float* beta3; float* rmean3; float* rvar3;
    const float* gamma4; const float* beta4; float* rmean4; float* rvar4;
    float eps, mom3, mom4;
    const float* dout; const float* res; float* g4; float* g3; float* dagg;      // backward
    float* dgamma3; float* dbeta3; float* db3; float* dgamma4; float* dbeta4; float* db4;
    float* part; int* ticket; float* pdw4;
    long long R;
};

// Weight fragments of a [rows, cols] matrix (both multiples of 16) into LDS, read with coalesced 16-byte loads.
//   plain:      fragment (n = o / 16, k = c / 16) at index n * (cols / 16) + k, lane (p = o % 16, g = c % 16 / 4), s = c % 4
//   transposed: fragment (k = c / 16, n = o / 16) at index k * (rows / 16) + n, lane (p = c % 16, g = o % 16 / 4), s = o % 4
// Two halves, so that the loads of all units are in flight together and a tile prefetch can be issued between them: the
// loads return in order, and waiting for the staging loads does not wait for the prefetch behind them.
template <int ROWS, int COLS>
struct StagedMatrix {
    static constexpr int UNITS = ROWS * COLS / 4, NV = (UNITS + BLOCK - 1) / BLOCK;
    float4 v[NV];
    __device__ __forceinline__ void load(const float* W) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int u = threadIdx.x + i * BLOCK;
            v[i] = (UNITS % BLOCK == 0 || u < UNITS) ? ld4(W + (size_t)u * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __device__ __forceinline__ void store(float4* wl, bool transposed) const {
        constexpr int C4 = COLS / 4;
        float* wf = reinterpret_cast<float*>(wl);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int u = threadIdx.x + i * BLOCK;
            if (UNITS % BLOCK != 0 && u >= UNITS) break;
            const int o = u / C4, c = 4 * (u % C4);
            if (!transposed) {
                wl[((o >> 4) * (COLS >> 4) + (c >> 4)) * WAVE + ((c & 15) >> 2) * 16 + (o & 15)] = v[i];
            } else {
                float* dst = wf + ((size_t)(((c >> 4) * (ROWS >> 4) + (o >> 4)) * WAVE + ((o & 15) >> 2) * 16 + (c & 15))) * 4 + (o & 3);
                dst[0] = v[i].x; dst[4] = v[i].y; dst[8] = v[i].z; dst[12] = v[i].w;
            }
        }
    }
};

// The tile loops below are software-pipelined by hand: the loads of a wave's first tile are issued in front of the weight
// staging, those of its next tile in front of the current tile's matrix products (a wave has two or three tiles, and one or
// two waves share a SIMD: nobody else would hide the 2-3 us a tile load takes).
template <int KA, int NH, int NO, int PASS>
__global__ __launch_bounds__(BLOCK) void tail_fwd_kernel(const TailArgs a) {
    constexpr int NFR = PASS == 1 ? NH * KA : NO * NH;
    constexpr int CA = 16 * KA, CH3 = 16 * NH, CO = 16 * NO;
    constexpr int NS = PASS == 1 ? NH : NO;
    constexpr int NIN = PASS == 1 ? KA : NH;                 // 16-channel tiles of the input row
    __shared__ float4 wl[NFR * WAVE];
    __shared__ double tot[2 * 16 * (NH > NO ? NH : NO)];
    const int lane = lane_id(), p = lane & 15, g = lane >> 4;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const long long ntiles = (a.R + 15) / 16, stride = (long long)gridDim.x * NWAVE;
    long long t = (long long)blockIdx.x * NWAVE + wave_id();
    const float* in = PASS == 1 ? a.agg : a.z3;
    StagedMatrix<PASS == 1 ? CH3 : CO, PASS == 1 ? CA : CH3> wst;
    wst.load(PASS == 1 ? a.W3 : a.W4);
    f32x4 nxt[NIN];
    if (t < ntiles) {
#pragma unroll
        for (int i = 0; i < NIN; ++i) nxt[i] = ld_row4(in, t * 16 + p, a.R, 16 * NIN, 16 * i + 4 * g);
    }
    wst.store(wl, false);
    f32x4 s1[NS], s2[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) { s1[i] = zero4; s2[i] = zero4; }
    f32x4 bias[NS], sc3[NH], sh3[NH];
#pragma unroll
    for (int i = 0; i < NS; ++i) bias[i] = v4(ld4((PASS == 1 ? a.b3 : a.b4) + 16 * i + 4 * g));
    if (PASS == 2) {
#pragma unroll
        for (int nh = 0; nh < NH; ++nh) { sc3[nh] = cst_row4(a.cst3, CH3, 0, 16 * nh + 4 * g); sh3[nh] = cst_row4(a.cst3, CH3, 1, 16 * nh + 4 * g); }
    }
    __syncthreads();
    for (; t < ntiles; t += stride) {
        const long long row = t * 16 + p;
        const float valid = row < a.R ? 1.f : 0.f;
        f32x4 x[NIN];
#pragma unroll
        for (int i = 0; i < NIN; ++i) x[i] = nxt[i];
        if (t + stride < ntiles) {
#pragma unroll
            for (int i = 0; i < NIN; ++i) nxt[i] = ld_row4(in, (t + stride) * 16 + p, a.R, 16 * NIN, 16 * i + 4 * g);
        }
        if (PASS == 1) {
            f32x4 z[NH];
#pragma unroll
            for (int nh = 0; nh < NH; ++nh) z[nh] = zero4;
#pragma unroll
            for (int ka = 0; ka < KA; ++ka)
#pragma unroll
                for (int nh = 0; nh < NH; ++nh) z[nh] = pmm(wl, nh * KA + ka, lane, x[ka], z[nh]);
#pragma unroll
            for (int nh = 0; nh < NH; ++nh) {
                z[nh] += bias[nh];
                st_row4(a.z3, row, a.R, CH3, 16 * nh + 4 * g, z[nh]);
                s1[nh] += z[nh] * valid; s2[nh] += z[nh] * z[nh] * valid;
            }
        } else {
            f32x4 y3[NH];
#pragma unroll
            for (int nh = 0; nh < NH; ++nh) y3[nh] = relu4p(x[nh] * sc3[nh] + sh3[nh]);
#pragma unroll
            for (int no = 0; no < NO; ++no) {
                f32x4 z = zero4;
#pragma unroll
                for (int nh = 0; nh < NH; ++nh) z = pmm(wl, no * NH + nh, lane, y3[nh], z);
                z += bias[no];
                st_row4(a.z4, row, a.R, CO, 16 * no + 4 * g, z);
                s1[no] += z * valid; s2[no] += z * z * valid;
            }
        }
    }
    if (!pass_totals<NS>(s1, s2, a.part, a.ticket, tot)) return;
    constexpr int NC = 16 * NS;
    if (threadIdx.x < NC) {
        if (PASS == 1) bn_fwd_constants(a.cst3, NC, threadIdx.x, tot[threadIdx.x], tot[NC + threadIdx.x], (double)a.R, a.gamma3, a.beta3, a.rmean3, a.rvar3, a.eps, a.mom3);
        else bn_fwd_constants(a.cst4, NC, threadIdx.x, tot[threadIdx.x], tot[NC + threadIdx.x], (double)a.R, a.gamma4, a.beta4, a.rmean4, a.rvar4, a.eps, a.mom4);
    }
}

template <int KA, int NH, int NO, int PASS>
__global__ __launch_bounds__(BLOCK) void tail_bwd_kernel(const TailArgs a) {
    constexpr int CA = 16 * KA, CH3 = 16 * NH, CO = 16 * NO;
    constexpr int NFR = PASS == 1 ? 1 : PASS == 2 ? NH * NO : KA * NH;
    constexpr int NDW = PASS == 2 ? NO * NH : 1;
    constexpr int NTB = PASS == 2 ? NO + NH : 1;
    constexpr int NIN = PASS == 1 ? 3 * NO : PASS == 2 ? 2 * NO + NH : 2 * NH;      // f32x4 a lane loads per tile
    __shared__ float4 wl[NFR * WAVE];
    __shared__ __align__(16) float tbuf[NWAVE * NTB * 16 * PT];
    __shared__ float red[NDW * 256];
    __shared__ double tot[2 * 16 * (NH > NO ? NH : NO)];
    const int lane = lane_id(), wave = wave_id(), p = lane & 15, g = lane >> 4;
    const long long ntiles = (a.R + 15) / 16, stride = (long long)gridDim.x * NWAVE;
    long long t = (long long)blockIdx.x * NWAVE + wave_id();
    // PASS 1: z4, shortcut, dout | 2: g4, z4, z3 | 3: g3, z3
    auto load_tile = [&](long long tt, f32x4 (&v)[NIN]) {
        const long long row = tt * 16 + p;
        if (PASS == 1) {
#pragma unroll
            for (int no = 0; no < NO; ++no) {
                v[no] = ld_row4(a.z4, row, a.R, CO, 16 * no + 4 * g);
                v[NO + no] = ld_row4(a.res, row, a.R, CO, 16 * no + 4 * g);
                v[2 * NO + no] = ld_row4(a.dout, row, a.R, CO, 16 * no + 4 * g);
            }
        } else if (PASS == 2) {
#pragma unroll
            for (int no = 0; no < NO; ++no) {
                v[no] = ld_row4(a.g4, row, a.R, CO, 16 * no + 4 * g);
                v[NO + no] = ld_row4(a.z4, row, a.R, CO, 16 * no + 4 * g);
            }
#pragma unroll
            for (int nh = 0; nh < NH; ++nh) v[2 * NO + nh] = ld_row4(a.z3, row, a.R, CH3, 16 * nh + 4 * g);
        } else {
#pragma unroll
            for (int nh = 0; nh < NH; ++nh) {
                v[nh] = ld_row4(a.g3, row, a.R, CH3, 16 * nh + 4 * g);
                v[NH + nh] = ld_row4(a.z3, row, a.R, CH3, 16 * nh + 4 * g);
            }
        }
    };
    StagedMatrix<PASS == 3 ? CH3 : CO, PASS == 3 ? CA : CH3> wst;
    if (PASS >= 2) wst.load(PASS == 3 ? a.W3 : a.W4);
    f32x4 nxt[NIN];
    if (t < ntiles) load_tile(t, nxt);
    if (PASS >= 2) wst.store(wl, true);
    float* tb = tbuf + wave * NTB * 16 * PT;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    constexpr int NS = PASS == 1 ? NO : NH;
    f32x4 s1[NS], s2[NS], dw[NDW];
#pragma unroll
    for (int i = 0; i < NS; ++i) { s1[i] = zero4; s2[i] = zero4; }
#pragma unroll
    for (int i = 0; i < NDW; ++i) dw[i] = zero4;
    f32x4 sc4[NO], sh4[NO], da4[NO], db4[NO], sc3[NH], sh3[NH], da3[NH], db3[NH];
    if (PASS <= 2) {
#pragma unroll
        for (int no = 0; no < NO; ++no) {
            const int c0 = 16 * no + 4 * g;
            sc4[no] = cst_row4(a.cst4, CO, 0, c0);
            if (PASS == 1) sh4[no] = cst_row4(a.cst4, CO, 1, c0);
            else { da4[no] = cst_row4(a.cst4, CO, 4, c0); db4[no] = cst_row4(a.cst4, CO, 5, c0); }
        }
    }
    if (PASS >= 2) {
#pragma unroll
        for (int nh = 0; nh < NH; ++nh) {
            const int c0 = 16 * nh + 4 * g;
            sc3[nh] = cst_row4(a.cst3, CH3, 0, c0);
            if (PASS == 2) sh3[nh] = cst_row4(a.cst3, CH3, 1, c0);
            else { da3[nh] = cst_row4(a.cst3, CH3, 4, c0); db3[nh] = cst_row4(a.cst3, CH3, 5, c0); }
        }
    }
    __syncthreads();
    for (; t < ntiles; t += stride) {
        const long long row = t * 16 + p;
        const float valid = row < a.R ? 1.f : 0.f;
        f32x4 v[NIN];
#pragma unroll
        for (int i = 0; i < NIN; ++i) v[i] = nxt[i];
        if (t + stride < ntiles) load_tile(t + stride, nxt);
        if (PASS == 1) {
#pragma unroll
            for (int no = 0; no < NO; ++no) {
                const f32x4 z = v[no];
                const f32x4 pre = z * sc4[no] + sh4[no] + v[NO + no];
                const f32x4 gg = dleaky4(v[2 * NO + no], pre) * valid;
                st_row4(a.g4, row, a.R, CO, 16 * no + 4 * g, gg);
                s1[no] += gg; s2[no] += gg * z;
            }
        } else if (PASS == 2) {
            f32x4 dz4[NO];
#pragma unroll
            for (int no = 0; no < NO; ++no) {
                dz4[no] = (v[no] * sc4[no] + (v[NO + no] * da4[no] + db4[no])) * valid;
                put_tile(tb + no * 16 * PT, dz4[no], p, g);
            }
            f32x4 pre3[NH];
#pragma unroll
            for (int nh = 0; nh < NH; ++nh) {
                pre3[nh] = v[2 * NO + nh] * sc3[nh] + sh3[nh];
                put_tile(tb + (NO + nh) * 16 * PT, relu4p(pre3[nh]) * valid, p, g);
            }
#pragma unroll
            for (int no = 0; no < NO; ++no)
#pragma unroll
                for (int nh = 0; nh < NH; ++nh) dw[no * NH + nh] = pouter(tb + no * 16 * PT, tb + (NO + nh) * 16 * PT, p, g, dw[no * NH + nh]);
#pragma unroll
            for (int nh = 0; nh < NH; ++nh) {
                f32x4 d = zero4;
#pragma unroll
                for (int no = 0; no < NO; ++no) d = pmm(wl, nh * NO + no, lane, dz4[no], d);
                const f32x4 gg = drelu4(d, pre3[nh]) * valid;
                st_row4(a.g3, row, a.R, CH3, 16 * nh + 4 * g, gg);
                s1[nh] += gg; s2[nh] += gg * v[2 * NO + nh];
            }
        } else {
            f32x4 dz3[NH];
#pragma unroll
            for (int nh = 0; nh < NH; ++nh) dz3[nh] = (v[nh] * sc3[nh] + (v[NH + nh] * da3[nh] + db3[nh])) * valid;
#pragma unroll
            for (int ka = 0; ka < KA; ++ka) {
                f32x4 d = zero4;
#pragma unroll
                for (int nh = 0; nh < NH; ++nh) d = pmm(wl, ka * NH + nh, lane, dz3[nh], d);
                st_row4(a.dagg, row, a.R, CA, 16 * ka + 4 * g, d);
            }
        }
    }
    if (PASS == 3) return;
    if (PASS == 2) flush_dw<NDW>(dw, red, a.pdw4);
    if (!pass_totals<NS>(s1, s2, a.part, a.ticket, tot)) return;
    constexpr int NC = 16 * NS;
    if (threadIdx.x < NC) {
        if (PASS == 1) bn_bwd_constants(a.cst4, NC, threadIdx.x, tot[threadIdx.x], tot[NC + threadIdx.x], (double)a.R, a.dgamma4, a.dbeta4, a.db4);
        else bn_bwd_constants(a.cst3, NC, threadIdx.x, tot[threadIdx.x], tot[NC + threadIdx.x], (double)a.R, a.dgamma3, a.dbeta3, a.db3);
    }
}

// ---- positional-encoding MLP of the strided / transposed PointConvs ------------------------------------------------------
// pe_convs = WeightNet(3, L, hidden_unit=[H]) (layers.py:599-604, 941-946): two Linear+BN+ReLU layers 3 -> H -> L on every
// edge, rows = edges.  The three offsets are ONE contraction step of the 16x16x4 product (k = 0..2, the fourth zero), so a
// lane loads one float per tile and layer 1 is one MFMA per 16 hidden channels; nothing but the input offsets and the
// output is kept: every pass recomputes the chain in front of it.
//   forward   F1: statistics of z1           F2: statistics of z2 = W2 ReLU(BN1(z1))         F3: out = ReLU(BN2(z2)) stored
//   backward  B1: g2 = dout * ReLU'(.), sums of layer 2     B2: dz2, dW2 tiles, g1 = (W2^T dz2) * ReLU'(.), sums of layer 1
//             B3: dz1, dW1 tiles                                                  (the offsets carry no gradient)
struct PeArgs {
    const float* rel; float* out;                                // [E,3] in; [E,16NL]
    const float* W1; const float* b1; const float* W2; const float* b2;     // [H,3], [H], [L,H], [L]
    float* cst1; float* cst2;
    const float* gamma1; const float* beta1; float* rmean1; float* rvar1;
    const float* gamma2; const float* beta2; float* rmean2; float* rvar2;
    float eps, mom1, mom2;
    const float* dout;                                           // backward: [E,16NL]
    float* dgamma1; float* dbeta1; float* db1; float* dgamma2; float* dbeta2; float* db2;
    float* part; int* ticket; float* pdw1; float* pdw2;
    long long R;
};

// FWD: PASS 1..3 as above; !FWD: PASS 1..3 = B1..B3
template <int NH, int NL, int PASS, bool FWD>
__global__ __launch_bounds__(BLOCK) void pe_chain_kernel(const PeArgs a) {
    constexpr int CH = 16 * NH, CL = 16 * NL;
    constexpr bool L2 = !(FWD && PASS == 1);                   // layer 2 is evaluated
    constexpr bool TR = !FWD && PASS >= 2;                     // W2^T is needed
    constexpr int NDW = FWD ? 1 : PASS == 2 ? NL * NH : PASS == 3 ? NH : 1;
    constexpr int NTB = FWD ? 1 : PASS == 2 ? NL + NH : PASS == 3 ? NH : 1;
    constexpr int NS = FWD ? (PASS == 1 ? NH : NL) : (PASS == 1 ? NL : NH);
    __shared__ float4 wl[(L2 ? NL * NH : 1) * WAVE];
    __shared__ float4 wt[(TR ? NH * NL : 1) * WAVE];
    __shared__ __align__(16) float tbuf[NWAVE * NTB * 16 * PT];
    __shared__ float red[NDW * 256];
    __shared__ double tot[2 * 16 * (NH > NL ? NH : NL)];
    const int lane = lane_id(), wave = wave_id(), p = lane & 15, g = lane >> 4;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const long long ntiles = (a.R + 15) / 16, stride = (long long)gridDim.x * NWAVE;
    long long t = (long long)blockIdx.x * NWAVE + wave_id();
    constexpr int ND = FWD ? 1 : NL;
    auto load_x = [&](long long tt) { const long long row = tt * 16 + p; return (g < 3 && row < a.R) ? a.rel[row * 3 + g] : 0.f; };
    auto load_d = [&](long long tt, f32x4 (&d)[ND]) {
        if (!FWD) {
#pragma unroll
            for (int nl = 0; nl < NL; ++nl) d[nl] = ld_row4(a.dout, tt * 16 + p, a.R, CL, 16 * nl + 4 * g);
        }
    };
    StagedMatrix<CL, CH> wst;
    if (L2) wst.load(a.W2);
    float xn = 0.f;
    f32x4 dn[ND];
    if (t < ntiles) { xn = load_x(t); load_d(t, dn); }
    if (L2) wst.store(wl, false);
    if (TR) wst.store(wt, true);
    float* tb = tbuf + wave * NTB * 16 * PT;
    float w1[NH];
    f32x4 bias1[NH], sc1[NH], sh1[NH], d11[NH], d01[NH], bias2[NL], sc2[NL], sh2[NL], d12[NL], d02[NL];
#pragma unroll
    for (int nh = 0; nh < NH; ++nh) {
        const int c0 = 16 * nh + 4 * g;
        w1[nh] = g < 3 ? a.W1[(16 * nh + p) * 3 + g] : 0.f;
        bias1[nh] = v4(ld4(a.b1 + c0));
        if (L2) { sc1[nh] = cst_row4(a.cst1, CH, 0, c0); sh1[nh] = cst_row4(a.cst1, CH, 1, c0); }
        if (!FWD && PASS == 3) { d11[nh] = cst_row4(a.cst1, CH, 4, c0); d01[nh] = cst_row4(a.cst1, CH, 5, c0); }
    }
    if (L2) {
#pragma unroll
        for (int nl = 0; nl < NL; ++nl) {
            const int c0 = 16 * nl + 4 * g;
            bias2[nl] = v4(ld4(a.b2 + c0));
            if (!(FWD && PASS == 2)) { sc2[nl] = cst_row4(a.cst2, CL, 0, c0); sh2[nl] = cst_row4(a.cst2, CL, 1, c0); }
            if (TR) { d12[nl] = cst_row4(a.cst2, CL, 4, c0); d02[nl] = cst_row4(a.cst2, CL, 5, c0); }
        }
    }
    f32x4 s1[NS], s2[NS], dw[NDW];
#pragma unroll
    for (int i = 0; i < NS; ++i) { s1[i] = zero4; s2[i] = zero4; }
#pragma unroll
    for (int i = 0; i < NDW; ++i) dw[i] = zero4;
    __syncthreads();
    for (; t < ntiles; t += stride) {
        const long long row = t * 16 + p;
        const float valid = row < a.R ? 1.f : 0.f;
        const float x = xn;
        f32x4 dout[ND];
#pragma unroll
        for (int i = 0; i < ND; ++i) dout[i] = dn[i];
        if (t + stride < ntiles) { xn = load_x(t + stride); load_d(t + stride, dn); }
        f32x4 z1[NH], y1[NH];
#pragma unroll
        for (int nh = 0; nh < NH; ++nh) {
            z1[nh] = PCF_MFMA(w1[nh], x, bias1[nh]);
            if (FWD && PASS == 1) { s1[nh] += z1[nh] * valid; s2[nh] += z1[nh] * z1[nh] * valid; }
            else y1[nh] = relu4p(z1[nh] * sc1[nh] + sh1[nh]);
        }
        if (FWD && PASS == 1) continue;
        f32x4 z2[NL];
#pragma unroll
        for (int nl = 0; nl < NL; ++nl) {
            f32x4 z = bias2[nl];
#pragma unroll
            for (int nh = 0; nh < NH; ++nh) z = pmm(wl, nl * NH + nh, lane, y1[nh], z);
            z2[nl] = z;
            if (FWD && PASS == 2) { s1[nl] += z * valid; s2[nl] += z * z * valid; }
            if (FWD && PASS == 3) st_row4(a.out, row, a.R, CL, 16 * nl + 4 * g, relu4p(z * sc2[nl] + sh2[nl]));
        }
        if (FWD) continue;
        f32x4 g2[NL];
#pragma unroll
        for (int nl = 0; nl < NL; ++nl) {
            g2[nl] = drelu4(dout[nl], z2[nl] * sc2[nl] + sh2[nl]) * valid;
            if (PASS == 1) { s1[nl] += g2[nl]; s2[nl] += g2[nl] * z2[nl]; }
        }
        if (PASS == 1) continue;
        f32x4 dz2[NL];
#pragma unroll
        for (int nl = 0; nl < NL; ++nl) {
            dz2[nl] = (g2[nl] * sc2[nl] + (z2[nl] * d12[nl] + d02[nl])) * valid;
            if (PASS == 2) put_tile(tb + nl * 16 * PT, dz2[nl], p, g);
        }
        if (PASS == 2) {
#pragma unroll
            for (int nh = 0; nh < NH; ++nh) put_tile(tb + (NL + nh) * 16 * PT, y1[nh] * valid, p, g);
#pragma unroll
            for (int nl = 0; nl < NL; ++nl)
#pragma unroll
                for (int nh = 0; nh < NH; ++nh) dw[nl * NH + nh] = pouter(tb + nl * 16 * PT, tb + (NL + nh) * 16 * PT, p, g, dw[nl * NH + nh]);
        }
        float relT[4];
        if (PASS == 3) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {                   // offsets of row 4s + g, coordinate p: the tile's 192 bytes again
                const long long r = t * 16 + 4 * s + g;
                relT[s] = (p < 3 && r < a.R) ? a.rel[r * 3 + p] : 0.f;
            }
        }
#pragma unroll
        for (int nh = 0; nh < NH; ++nh) {
            f32x4 d = zero4;
#pragma unroll
            for (int nl = 0; nl < NL; ++nl) d = pmm(wt, nh * NL + nl, lane, dz2[nl], d);
            const f32x4 g1 = drelu4(d, z1[nh] * sc1[nh] + sh1[nh]) * valid;
            if (PASS == 2) { s1[nh] += g1; s2[nh] += g1 * z1[nh]; }
            else {
                const f32x4 dz1 = (g1 * sc1[nh] + (z1[nh] * d11[nh] + d01[nh])) * valid;
                put_tile(tb + nh * 16 * PT, dz1, p, g);
                const float* dzb = tb + nh * 16 * PT;
#pragma unroll
                for (int s = 0; s < 4; ++s) dw[nh] = PCF_MFMA(dzb[(4 * s + g) * PT + p], relT[s], dw[nh]);
            }
        }
    }
    if (!FWD && PASS >= 2) flush_dw<NDW>(dw, red, PASS == 2 ? a.pdw2 : a.pdw1);
    if (PASS == 3) return;
    if (!pass_totals<NS>(s1, s2, a.part, a.ticket, tot)) return;
    constexpr int NC = 16 * NS;
    if (threadIdx.x < NC) {
        const double S1 = tot[threadIdx.x], S2 = tot[NC + threadIdx.x];
        if (FWD && PASS == 1) bn_fwd_constants(a.cst1, NC, threadIdx.x, S1, S2, (double)a.R, a.gamma1, a.beta1, a.rmean1, a.rvar1, a.eps, a.mom1);
        if (FWD && PASS == 2) bn_fwd_constants(a.cst2, NC, threadIdx.x, S1, S2, (double)a.R, a.gamma2, a.beta2, a.rmean2, a.rvar2, a.eps, a.mom2);
        if (!FWD && PASS == 1) bn_bwd_constants(a.cst2, NC, threadIdx.x, S1, S2, (double)a.R, a.dgamma2, a.dbeta2, a.db2);
        if (!FWD && PASS == 2) bn_bwd_constants(a.cst1, NC, threadIdx.x, S1, S2, (double)a.R, a.dgamma1, a.dbeta1, a.db1);
    }
}

template <int NH, int NL>
static void pe_launch_forward(const PeArgs& a, int grid, hipStream_t s, bool fin) {
    hipLaunchKernelGGL((pe_chain_kernel<NH, NL, 1, true>), dim3(grid), dim3(BLOCK), 0, s, a);
    if (fin) (void)pc_finish_fwd(a.part, grid, 16 * NH, a.R, a.cst1, a.gamma1, a.beta1, a.rmean1, a.rvar1, a.eps, a.mom1, s);
    hipLaunchKernelGGL((pe_chain_kernel<NH, NL, 2, true>), dim3(grid), dim3(BLOCK), 0, s, a);
    if (fin) (void)pc_finish_fwd(a.part, grid, 16 * NL, a.R, a.cst2, a.gamma2, a.beta2, a.rmean2, a.rvar2, a.eps, a.mom2, s);
    hipLaunchKernelGGL((pe_chain_kernel<NH, NL, 3, true>), dim3(grid), dim3(BLOCK), 0, s, a);
}
template <int NH, int NL>
static void pe_launch_backward(const PeArgs& a, int grid, hipStream_t s, bool fin) {
    hipLaunchKernelGGL((pe_chain_kernel<NH, NL, 1, false>), dim3(grid), dim3(BLOCK), 0, s, a);
    if (fin) (void)pc_finish_bwd(a.part, grid, 16 * NL, a.R, a.cst2, a.dgamma2, a.dbeta2, a.db2, s);
    hipLaunchKernelGGL((pe_chain_kernel<NH, NL, 2, false>), dim3(grid), dim3(BLOCK), 0, s, a);
    if (fin) (void)pc_finish_bwd(a.part, grid, 16 * NH, a.R, a.cst1, a.dgamma1, a.dbeta1, a.db1, s);
    hipLaunchKernelGGL((pe_chain_kernel<NH, NL, 3, false>), dim3(grid), dim3(BLOCK), 0, s, a);
}

// Weight-gradient partial tiles -> matrices.  Up to DWR_MAX groups; group i: nblocks lists of `tiles` 16x16 tiles laid out
// [tile = to * tiles_c + tc][o_local][c_local]; out[o][c] for o < rows, c < cols (row-major, leading dimension cols).
constexpr int DWR_MAX = 4;
struct DwReduceArgs { const float* part[DWR_MAX]; float* out[DWR_MAX]; int tiles_c[DWR_MAX], tiles[DWR_MAX], rows[DWR_MAX], cols[DWR_MAX], block0[DWR_MAX + 1]; int n, nblocks; };

__global__ __launch_bounds__(1024) void dw_reduce_kernel(const DwReduceArgs f) {
    __shared__ float sh[16][64];
    int i = 0;
    while (i + 1 < f.n && (int)blockIdx.x >= f.block0[i + 1]) ++i;
    const int e = (blockIdx.x - f.block0[i]) * 64 + (threadIdx.x & 63);          // element of the group's tile list
    const int slice = threadIdx.x >> 6, total = f.tiles[i] * 256;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (e < total) {
        const float* src = f.part[i] + e;
        const size_t stride = (size_t)total;
        int q = slice;
        for (; q + 48 < f.nblocks; q += 64) {
            a0 += src[(size_t)q * stride]; a1 += src[(size_t)(q + 16) * stride];
            a2 += src[(size_t)(q + 32) * stride]; a3 += src[(size_t)(q + 48) * stride];
        }
        for (; q < f.nblocks; q += 16) a0 += src[(size_t)q * stride];
    }
    sh[slice][threadIdx.x & 63] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (threadIdx.x >= 64 || e >= total) return;
    float t = 0.f;
#pragma unroll
    for (int sl = 0; sl < 16; ++sl) t += sh[sl][threadIdx.x];
    const int tile = e >> 8, ol = (e >> 4) & 15, cl = e & 15;
    const int o = (tile / f.tiles_c[i]) * 16 + ol, c = (tile % f.tiles_c[i]) * 16 + cl;
    if (o < f.rows[i] && c < f.cols[i]) f.out[i][(size_t)o * f.cols[i] + c] = t;
}

// edge rows: at least four tiles per wave, up to PE_MAXB workgroups
static inline int pe_grid(long long E) {
    const long long tiles = (E + 15) / 16;
    return (int)std::max<long long>(1, std::min<long long>((tiles + 4 * NWAVE - 1) / (4 * NWAVE), PE_MAXB));
}
static inline int pc_grid(long long R) {
    const long long tiles = (R + 15) / 16;
    return (int)std::max<long long>(1, std::min<long long>((tiles + NWAVE - 1) / NWAVE, PC_MAXB));
}

}  // namespace pcf

extern "C" {

// widths the row chains are instantiated for: (C_in, mid, guidance width) of the head
int pcf_hip_point_head_supported(int c_in, int mid, int g) {
    return (c_in == 64 && mid == 16 && g == 32) ? 1 : 0;
}

size_t pcf_hip_point_head_workspace_bytes(long long R, int c_in, int mid, int g) {
    if (R < 0 || !pcf_hip_point_head_supported(c_in, mid, g)) return 0;
    const size_t nb = pcf::PC_MAXB;
    const size_t stats = pcf::pc_part_floats(std::max(mid, g)) * 4;
    const size_t dwa = nb * (g / 16) * 256 * 4, dw2 = nb * (g / 16) * (mid / 16) * 256 * 4, dw1 = nb * (mid / 16) * (c_in / 16) * 256 * 4;
    return stats + dwa + dw2 + dw1 + (size_t)R * mid * 4 + 1024;
}

// forward: z1, fx, u, records cst1 / cst2 (rows 0..3) and the running statistics of both BatchNorms
int pcf_hip_point_head_forward(const float* x, long long R, int c_in, int mid, int g, const float* W1, const float* b1,
                               const float* gamma1, const float* beta1, float* rmean1, float* rvar1, float mom1, const float* W2,
                               const float* b2, const float* gamma2, const float* beta2, float* rmean2, float* rvar2, float mom2,
                               const float* Wa, float eps, float* z1, float* fx, float* u, float* cst1, float* cst2, void* workspace,
                               size_t workspace_bytes, int* tickets, void* stream) {
    using namespace pcf;
    if (!pcf_hip_point_head_supported(c_in, mid, g))
        return fail(PCF_E_UNSUPPORTED, "point_head: widths (%d, %d, %d) are not instantiated", c_in, mid, g);
    PCF_REQUIRE(R >= 0 && R < (1ll << 31), "point_head: bad row count");
    if (R == 0) return ok();
    PCF_REQUIRE(x && W1 && b1 && gamma1 && beta1 && W2 && b2 && gamma2 && beta2 && Wa && z1 && fx && u && cst1 && cst2 && tickets,
                "point_head_forward: null pointer");
    PCF_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= pcf_hip_point_head_workspace_bytes(R, c_in, mid, g),
                "point_head_forward: workspace too small or misaligned");
    PCF_REQUIRE(aligned16(x) && aligned16(z1) && aligned16(fx) && aligned16(u) && aligned16(cst1) && aligned16(cst2) && aligned16(b1) && aligned16(b2),
                "point_head_forward: buffers must be 16-byte aligned");
    HeadArgs a{};
    a.x = x; a.z1 = z1; a.fx = fx; a.u = u; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.Wa = Wa; a.cst1 = cst1; a.cst2 = cst2;
    a.gamma1 = gamma1; a.beta1 = beta1; a.rmean1 = rmean1; a.rvar1 = rvar1; a.gamma2 = gamma2; a.beta2 = beta2; a.rmean2 = rmean2; a.rvar2 = rvar2;
    a.eps = eps; a.mom1 = mom1; a.mom2 = mom2; a.part = static_cast<float*>(workspace); a.ticket = tickets; a.R = R;
    hipStream_t s = (hipStream_t)stream;
    const int grid = pc_grid(R);
    const bool fin = g_pc_finish != 0;
    if (fin) a.ticket = nullptr;
    hipLaunchKernelGGL((head_fwd_kernel<4, 1, 2, 1>), dim3(grid), dim3(BLOCK), 0, s, a);
    if (fin) (void)pc_finish_fwd(a.part, grid, mid, R, cst1, gamma1, beta1, rmean1, rvar1, eps, mom1, s);
    hipLaunchKernelGGL((head_fwd_kernel<4, 1, 2, 2>), dim3(grid), dim3(BLOCK), 0, s, a);
    if (fin) (void)pc_finish_fwd(a.part, grid, g, R, cst2, gamma2, beta2, rmean2, rvar2, eps, mom2, s);
    hipLaunchKernelGGL((head_fwd_kernel<4, 1, 2, 3>), dim3(grid), dim3(BLOCK), 0, s, a);
    return check_launch("head_fwd_kernel<1,2,3>");
}

// backward: dx, parameter gradients of both layers and of Wa; cst1 / cst2 rows 4, 5 are written on the way
int pcf_hip_point_head_backward(const float* dfx, const float* du, const float* x, const float* z1, const float* fx, long long R,
                                int c_in, int mid, int g, const float* W1, const float* W2, const float* b2, const float* Wa,
                                float* cst1, float* cst2, float* dx, float* dW1, float* db1, float* dgamma1, float* dbeta1,
                                float* dW2, float* db2, float* dgamma2, float* dbeta2, float* dWa, void* workspace,
                                size_t workspace_bytes, int* tickets, void* stream) {
    using namespace pcf;
    if (!pcf_hip_point_head_supported(c_in, mid, g))
        return fail(PCF_E_UNSUPPORTED, "point_head: widths (%d, %d, %d) are not instantiated", c_in, mid, g);
    PCF_REQUIRE(R >= 1 && R < (1ll << 31), "point_head_backward: bad row count");
    PCF_REQUIRE(du && x && z1 && fx && W1 && W2 && b2 && Wa && cst1 && cst2 && dx && dW1 && db1 && dgamma1 && dbeta1 && dW2 && db2 &&
                dgamma2 && dbeta2 && dWa && tickets, "point_head_backward: null pointer");
    PCF_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= pcf_hip_point_head_workspace_bytes(R, c_in, mid, g),
                "point_head_backward: workspace too small or misaligned");
    PCF_REQUIRE(aligned16(du) && (!dfx || aligned16(dfx)) && aligned16(dx) && aligned16(x) && aligned16(z1) && aligned16(fx),
                "point_head_backward: buffers must be 16-byte aligned");
    HeadArgs a{};
    a.x = x; a.z1 = const_cast<float*>(z1); a.fx = const_cast<float*>(fx); a.W1 = W1; a.W2 = W2; a.b2 = b2; a.Wa = Wa;
    a.cst1 = cst1; a.cst2 = cst2; a.dfx = dfx; a.du = du; a.dx = dx;
    a.dgamma1 = dgamma1; a.dbeta1 = dbeta1; a.db1 = db1; a.dgamma2 = dgamma2; a.dbeta2 = dbeta2; a.db2 = db2;
    a.ticket = tickets; a.R = R;
    const size_t nb = PC_MAXB;
    float* w = static_cast<float*>(workspace);
    a.part = w; w += pc_part_floats(std::max(mid, g));
    a.pdwa = w; w += nb * (g / 16) * 256;
    a.pdw2 = w; w += nb * (g / 16) * (mid / 16) * 256;
    a.pdw1 = w; w += nb * (mid / 16) * (c_in / 16) * 256;
    a.g1 = w;
    hipStream_t s = (hipStream_t)stream;
    const int grid = pc_grid(R);
    const bool fin = g_pc_finish != 0;
    if (fin) a.ticket = nullptr;
    hipLaunchKernelGGL((head_bwd_kernel<4, 1, 2, 1>), dim3(grid), dim3(BLOCK), 0, s, a);
    if (fin) (void)pc_finish_bwd(a.part, grid, g, R, cst2, dgamma2, dbeta2, db2, s);
    hipLaunchKernelGGL((head_bwd_kernel<4, 1, 2, 2>), dim3(grid), dim3(BLOCK), 0, s, a);
    if (fin) (void)pc_finish_bwd(a.part, grid, mid, R, cst1, dgamma1, dbeta1, db1, s);
    hipLaunchKernelGGL((head_bwd_kernel<4, 1, 2, 3>), dim3(grid), dim3(BLOCK), 0, s, a);
    if (int e = check_launch("head_bwd_kernel<1,2,3>")) return e;
    DwReduceArgs r{};
    r.n = 3; r.nblocks = grid;
    const float* parts[3] = {a.pdwa, a.pdw2, a.pdw1};
    float* outs[3] = {dWa, dW2, dW1};
    const int tc[3] = {g / 16, mid / 16, c_in / 16}, tl[3] = {g / 16, (g / 16) * (mid / 16), (mid / 16) * (c_in / 16)};
    const int rows[3] = {PC_CH, g, mid}, cols[3] = {g, mid, c_in};
    int blocks = 0;
    for (int i = 0; i < 3; ++i) {
        r.part[i] = parts[i]; r.out[i] = outs[i]; r.tiles_c[i] = tc[i]; r.tiles[i] = tl[i]; r.rows[i] = rows[i]; r.cols[i] = cols[i];
        r.block0[i] = blocks; blocks += tl[i] * 4;
    }
    r.block0[3] = blocks;
    hipLaunchKernelGGL(dw_reduce_kernel, dim3(blocks), dim3(1024), 0, s, r);
    return check_launch("dw_reduce_kernel");
}

// widths of the tail: (mid * C_mid, C_out / 2, C_out)
int pcf_hip_point_tail_supported(int c_agg, int c_half, int c_out) {
    return (c_agg == 256 && c_half == 32 && c_out == 64) ? 1 : 0;
}

size_t pcf_hip_point_tail_workspace_bytes(long long R, int c_agg, int c_half, int c_out) {
    if (R < 0 || !pcf_hip_point_tail_supported(c_agg, c_half, c_out)) return 0;
    const size_t nb = pcf::PC_MAXB;
    return pcf::pc_part_floats(std::max(c_half, c_out)) * 4 + nb * (c_out / 16) * (c_half / 16) * 256 * 4 + 1024;
}

// z3, z4 (raw layer outputs), cst3 / cst4 rows 0..3, running statistics; out = act(BN4(z4) + shortcut) is the caller's
// pcf_hip_bnact_forward_res with mean = cst4 row 2, rstd = cst4 row 3
int pcf_hip_point_tail_forward(const float* agg, long long R, int c_agg, int c_half, int c_out, const float* W3, const float* b3,
                               const float* gamma3, const float* beta3, float* rmean3, float* rvar3, float mom3, const float* W4,
                               const float* b4, const float* gamma4, const float* beta4, float* rmean4, float* rvar4, float mom4,
                               float eps, float* z3, float* z4, float* cst3, float* cst4, void* workspace, size_t workspace_bytes,
                               int* tickets, void* stream) {
    using namespace pcf;
    if (!pcf_hip_point_tail_supported(c_agg, c_half, c_out))
        return fail(PCF_E_UNSUPPORTED, "point_tail: widths (%d, %d, %d) are not instantiated", c_agg, c_half, c_out);
    PCF_REQUIRE(R >= 0 && R < (1ll << 31), "point_tail: bad row count");
    if (R == 0) return ok();
    PCF_REQUIRE(agg && W3 && b3 && gamma3 && beta3 && W4 && b4 && gamma4 && beta4 && z3 && z4 && cst3 && cst4 && tickets,
                "point_tail_forward: null pointer");
    PCF_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= pcf_hip_point_tail_workspace_bytes(R, c_agg, c_half, c_out),
                "point_tail_forward: workspace too small or misaligned");
    PCF_REQUIRE(aligned16(agg) && aligned16(z3) && aligned16(z4) && aligned16(cst3) && aligned16(cst4) && aligned16(b3) && aligned16(b4),
                "point_tail_forward: buffers must be 16-byte aligned");
    TailArgs a{};
    a.agg = agg; a.z3 = z3; a.z4 = z4; a.W3 = W3; a.b3 = b3; a.W4 = W4; a.b4 = b4; a.cst3 = cst3; a.cst4 = cst4;
    a.gamma3 = gamma3; a.beta3 = beta3; a.rmean3 = rmean3; a.rvar3 = rvar3; a.gamma4 = gamma4; a.beta4 = beta4; a.rmean4 = rmean4; a.rvar4 = rvar4;
    a.eps = eps; a.mom3 = mom3; a.mom4 = mom4; a.part = static_cast<float*>(workspace); a.ticket = tickets; a.R = R;
    hipStream_t s = (hipStream_t)stream;
    const int grid = pc_grid(R);
    const bool fin = g_pc_finish != 0;
    if (fin) a.ticket = nullptr;
    hipLaunchKernelGGL((tail_fwd_kernel<16, 2, 4, 1>), dim3(grid), dim3(BLOCK), 0, s, a);
    if (fin) (void)pc_finish_fwd(a.part, grid, c_half, R, cst3, gamma3, beta3, rmean3, rvar3, eps, mom3, s);
    hipLaunchKernelGGL((tail_fwd_kernel<16, 2, 4, 2>), dim3(grid), dim3(BLOCK), 0, s, a);
    if (fin) (void)pc_finish_fwd(a.part, grid, c_out, R, cst4, gamma4, beta4, rmean4, rvar4, eps, mom4, s);
    return check_launch("tail_fwd_kernel<1,2>");
}

// g4 (= gradient of the shortcut), g3 (scratch in the workspace), dagg, dW4 and the BatchNorm / bias gradients of both
// layers; cst3 / cst4 rows 4, 5 are written.  dW3 = dz3^T agg is the caller's pcf_hip_flin_backward_weight_slabs on
// (g3_out, z3, cst3, ReLU, agg): g3_out [R, c_half] receives g3.
int pcf_hip_point_tail_backward(const float* dout, const float* res, const float* z3, const float* z4, long long R, int c_agg,
                                int c_half, int c_out, const float* W3, const float* W4, float* cst3, float* cst4, float* g4,
                                float* g3_out, float* dagg, float* dW4, float* db3, float* dgamma3, float* dbeta3, float* db4,
                                float* dgamma4, float* dbeta4, void* workspace, size_t workspace_bytes, int* tickets, void* stream) {
    using namespace pcf;
    if (!pcf_hip_point_tail_supported(c_agg, c_half, c_out))
        return fail(PCF_E_UNSUPPORTED, "point_tail: widths (%d, %d, %d) are not instantiated", c_agg, c_half, c_out);
    PCF_REQUIRE(R >= 1 && R < (1ll << 31), "point_tail_backward: bad row count");
    PCF_REQUIRE(dout && res && z3 && z4 && W3 && W4 && cst3 && cst4 && g4 && g3_out && dagg && dW4 && db3 && dgamma3 && dbeta3 && db4 &&
                dgamma4 && dbeta4 && tickets, "point_tail_backward: null pointer");
    PCF_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= pcf_hip_point_tail_workspace_bytes(R, c_agg, c_half, c_out),
                "point_tail_backward: workspace too small or misaligned");
    PCF_REQUIRE(aligned16(dout) && aligned16(res) && aligned16(g4) && aligned16(g3_out) && aligned16(dagg),
                "point_tail_backward: buffers must be 16-byte aligned");
    TailArgs a{};
    a.z3 = const_cast<float*>(z3); a.z4 = const_cast<float*>(z4); a.W3 = W3; a.W4 = W4; a.cst3 = cst3; a.cst4 = cst4;
    a.dout = dout; a.res = res; a.g4 = g4; a.g3 = g3_out; a.dagg = dagg;
    a.dgamma3 = dgamma3; a.dbeta3 = dbeta3; a.db3 = db3; a.dgamma4 = dgamma4; a.dbeta4 = dbeta4; a.db4 = db4;
    a.ticket = tickets; a.R = R;
    float* w = static_cast<float*>(workspace);
    a.part = w; w += pc_part_floats(std::max(c_half, c_out));
    a.pdw4 = w;
    hipStream_t s = (hipStream_t)stream;
    const int grid = pc_grid(R);
    const bool fin = g_pc_finish != 0;
    if (fin) a.ticket = nullptr;
    hipLaunchKernelGGL((tail_bwd_kernel<16, 2, 4, 1>), dim3(grid), dim3(BLOCK), 0, s, a);
    if (fin) (void)pc_finish_bwd(a.part, grid, c_out, R, cst4, dgamma4, dbeta4, db4, s);
    hipLaunchKernelGGL((tail_bwd_kernel<16, 2, 4, 2>), dim3(grid), dim3(BLOCK), 0, s, a);
    if (fin) (void)pc_finish_bwd(a.part, grid, c_half, R, cst3, dgamma3, dbeta3, db3, s);
    hipLaunchKernelGGL((tail_bwd_kernel<16, 2, 4, 3>), dim3(grid), dim3(BLOCK), 0, s, a);
    if (int e = check_launch("tail_bwd_kernel<1,2,3>")) return e;
    DwReduceArgs r{};
    r.n = 1; r.nblocks = grid;
    r.part[0] = a.pdw4; r.out[0] = dW4; r.tiles_c[0] = c_half / 16; r.tiles[0] = (c_out / 16) * (c_half / 16); r.rows[0] = c_out; r.cols[0] = c_half;
    r.block0[0] = 0; r.block0[1] = r.tiles[0] * 4;
    hipLaunchKernelGGL(dw_reduce_kernel, dim3(r.block0[1]), dim3(1024), 0, s, r);
    return check_launch("dw_reduce_kernel");
}

// (hidden width, output width) of the positional-encoding MLPs the chain is instantiated for: out_channel / 4 and
// min(out_channel / 4, 32) for out_channel = 64, 128, 256 (the 512-wide levels have a few thousand edges: not worth the
// 128-channel instantiation, whose weight fragments do not fit the register file)
int pcf_hip_pe_chain_supported(int hidden, int c_out) {
    return ((hidden == 16 && c_out == 16) || ((hidden == 32 || hidden == 64) && c_out == 32)) ? 1 : 0;
}

size_t pcf_hip_pe_chain_workspace_bytes(long long E, int hidden, int c_out) {
    if (E < 0 || !pcf_hip_pe_chain_supported(hidden, c_out)) return 0;
    const size_t nb = pcf::pe_grid(E);
    return pcf::pc_part_floats(std::max(hidden, c_out), pcf::PE_MAXB) * 4 + nb * ((size_t)(hidden / 16) * (c_out / 16) + hidden / 16) * 256 * 4 + 1024;
}

// out [E, c_out] = ReLU(BN2(W2 ReLU(BN1(W1 rel + b1)) + b2)) with batch statistics; cst1 [6][hidden], cst2 [6][c_out] rows
// 0..3 and the running statistics are written
int pcf_hip_pe_chain_forward(const float* rel, long long E, int hidden, int c_out, const float* W1, const float* b1,
                             const float* gamma1, const float* beta1, float* rmean1, float* rvar1, float mom1, const float* W2,
                             const float* b2, const float* gamma2, const float* beta2, float* rmean2, float* rvar2, float mom2,
                             float eps, float* out, float* cst1, float* cst2, void* workspace, size_t workspace_bytes, int* tickets,
                             void* stream) {
    using namespace pcf;
    if (!pcf_hip_pe_chain_supported(hidden, c_out))
        return fail(PCF_E_UNSUPPORTED, "pe_chain: widths (%d, %d) are not instantiated", hidden, c_out);
    PCF_REQUIRE(E >= 0 && E < (1ll << 31), "pe_chain: bad row count");
    if (E == 0) return ok();
    PCF_REQUIRE(rel && W1 && b1 && gamma1 && beta1 && W2 && b2 && gamma2 && beta2 && out && cst1 && cst2 && tickets, "pe_chain_forward: null pointer");
    PCF_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= pcf_hip_pe_chain_workspace_bytes(E, hidden, c_out),
                "pe_chain_forward: workspace too small or misaligned");
    PCF_REQUIRE(aligned16(out) && aligned16(cst1) && aligned16(cst2) && aligned16(b1) && aligned16(b2) && aligned16(W2),
                "pe_chain_forward: buffers must be 16-byte aligned");
    PeArgs a{};
    a.rel = rel; a.out = out; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.cst1 = cst1; a.cst2 = cst2;
    a.gamma1 = gamma1; a.beta1 = beta1; a.rmean1 = rmean1; a.rvar1 = rvar1; a.gamma2 = gamma2; a.beta2 = beta2; a.rmean2 = rmean2; a.rvar2 = rvar2;
    a.eps = eps; a.mom1 = mom1; a.mom2 = mom2; a.part = static_cast<float*>(workspace); a.ticket = tickets; a.R = E;
    hipStream_t s = (hipStream_t)stream;
    const int grid = pe_grid(E);
    const bool fin = g_pc_finish != 0;
    if (fin) a.ticket = nullptr;
    if (hidden == 16) pe_launch_forward<1, 1>(a, grid, s, fin);
    else if (hidden == 32) pe_launch_forward<2, 2>(a, grid, s, fin);
    else pe_launch_forward<4, 2>(a, grid, s, fin);
    return check_launch("pe_chain_kernel<forward>");
}

// parameter gradients of both layers from dout [E, c_out]; cst1 / cst2 rows 4, 5 are written
int pcf_hip_pe_chain_backward(const float* dout, const float* rel, long long E, int hidden, int c_out, const float* W1, const float* b1,
                              const float* W2, const float* b2, float* cst1, float* cst2, float* dW1, float* db1, float* dgamma1,
                              float* dbeta1, float* dW2, float* db2, float* dgamma2, float* dbeta2, void* workspace,
                              size_t workspace_bytes, int* tickets, void* stream) {
    using namespace pcf;
    if (!pcf_hip_pe_chain_supported(hidden, c_out))
        return fail(PCF_E_UNSUPPORTED, "pe_chain: widths (%d, %d) are not instantiated", hidden, c_out);
    PCF_REQUIRE(E >= 1 && E < (1ll << 31), "pe_chain_backward: bad row count");
    PCF_REQUIRE(dout && rel && W1 && b1 && W2 && b2 && cst1 && cst2 && dW1 && db1 && dgamma1 && dbeta1 && dW2 && db2 && dgamma2 && dbeta2 &&
                tickets, "pe_chain_backward: null pointer");
    PCF_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= pcf_hip_pe_chain_workspace_bytes(E, hidden, c_out),
                "pe_chain_backward: workspace too small or misaligned");
    PCF_REQUIRE(aligned16(dout) && aligned16(cst1) && aligned16(cst2) && aligned16(b1) && aligned16(b2) && aligned16(W2),
                "pe_chain_backward: buffers must be 16-byte aligned");
    PeArgs a{};
    a.rel = rel; a.dout = dout; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.cst1 = cst1; a.cst2 = cst2;
    a.dgamma1 = dgamma1; a.dbeta1 = dbeta1; a.db1 = db1; a.dgamma2 = dgamma2; a.dbeta2 = dbeta2; a.db2 = db2;
    a.ticket = tickets; a.R = E;
    float* w = static_cast<float*>(workspace);
    a.part = w; w += pc_part_floats(std::max(hidden, c_out), PE_MAXB);
    const int nh = hidden / 16, nl = c_out / 16;
    const int grid = pe_grid(E);
    a.pdw2 = w; w += (size_t)grid * nl * nh * 256;
    a.pdw1 = w;
    hipStream_t s = (hipStream_t)stream;
    const bool fin = g_pc_finish != 0;
    if (fin) a.ticket = nullptr;
    if (hidden == 16) pe_launch_backward<1, 1>(a, grid, s, fin);
    else if (hidden == 32) pe_launch_backward<2, 2>(a, grid, s, fin);
    else pe_launch_backward<4, 2>(a, grid, s, fin);
    if (int e = check_launch("pe_chain_kernel<backward>")) return e;
    DwReduceArgs r{};
    r.n = 2; r.nblocks = grid;
    r.part[0] = a.pdw2; r.out[0] = dW2; r.tiles_c[0] = nh; r.tiles[0] = nl * nh; r.rows[0] = c_out; r.cols[0] = hidden;
    r.part[1] = a.pdw1; r.out[1] = dW1; r.tiles_c[1] = 1; r.tiles[1] = nh; r.rows[1] = hidden; r.cols[1] = 3;
    r.block0[0] = 0; r.block0[1] = r.tiles[0] * 4; r.block0[2] = r.block0[1] + r.tiles[1] * 4;
    hipLaunchKernelGGL(dw_reduce_kernel, dim3(r.block0[2]), dim3(1024), 0, s, r);
    return check_launch("dw_reduce_kernel");
}

// how the row chains' per-workgroup column sums become a layer's record: 1 a small launch of its own, 0 the last workgroup
int pcf_hip_set_row_chain_finish(int separate_launch) {
    pcf::g_pc_finish = separate_launch ? 1 : 0;
    return pcf::ok();
}

}  // extern "C"
