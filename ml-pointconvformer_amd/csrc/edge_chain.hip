// Fused forward of the per-edge graph of a PCFLayer on gfx950 (self neighbourhoods: key = neighbour 0; strided layers:
// key = maximum of the query over the neighbourhood, ChainArgs::ukey).
//
//        VI[e] (<=12) --mlp_conv--> pe (<=32) --+                           guidance branch
//                                               +-- Wb.pe + u[idx[e]] - (same for the key edge) + b
//                                                   --BN,ReLU--> h1 (8) --g2,BN,sigmoid--> score (H <= 8)
//        VI[e] --w1--> a1 (8) --w2--> a2 (8) --w3--> w (Cm <= 16)         WeightNet branch (BN+ReLU each)
//
// (layers.py:361-384 with MultiHeadGuidance :47-68 and WeightNet :163-171; u = Wa.guidance_x is the
// per-point half of the first guidance layer, see pcf_hip_rowlin_*_ex.)  In training every BatchNorm
// needs the statistics of its pre-activation over ALL edges before the next layer can run, which is
// three dependent global reductions.  Layer-at-a-time execution (edge_mlp_mfma.hip) costs 12 kernels that
// stream [E, 8..32] activations back and forth (~2 kB per edge); here:
//     pass 1: statistics of mlp_conv and w1                      (reads the 48-byte VI row)
//     pass 2: recomputes them, runs g1 and w2, takes their statistics and stores their raw 8-channel
//             accumulators (2 x 32 B per edge)                   (reads VI, idx, u)
//     pass 3: statistics of g2 and w3 from the stored accumulators      (pcf_chain_tail_kernel<true>)
//     pass 4: score and w from the stored accumulators                  (pcf_chain_tail_kernel<false>)
// The two stored accumulators are also where the fused backward (edge_chain_bwd.hip) restarts.  Callers that
// want the layer-at-a-time backward instead get pe / a1 / h1 / a2 written by the full-recompute passes
// (pcf_chain_kernel<3>, <4>); in inference (running statistics) only pcf_chain_kernel<4> runs and nothing but
// score and w is written.  The chain kernels are fp32-ALU-bound (the fp32 MFMA shares the VALU's lanes), the
// tail kernels HBM-bound.
#include <algorithm>

#include "edge_chain.h"

// fused multiply-adds in the per-element BatchNorm algebra, as in edge_chain_bwd.hip (the library default is
// contraction off): the backward recomputes the first layers with the same rounding
#pragma clang fp contract(fast)

namespace pcf {

// BatchNorm constants of output-tile slot `f` (0 pe lo, 1 pe hi, 2 w1, 3 g1, 4 w2, 5 g2, 6 w3), channel c of the
// tile: BN(acc + bias) = acc * scale + shift.  Kept in LDS ([slot][scale|shift][16]); a lane reads its four
// channels 4g..4g+3 with one ds_read_b128 each when a tile needs them, instead of pinning 56 VGPRs.
__device__ __forceinline__ void stage_frags(const ChainArgs& a, float (*cf)[2][16], int level) {
    const int layer_of[7] = {L_PE, L_PE, L_W1, L_G1, L_W2, L_G2, L_W3};
    const int tile_of[7] = {0, 1, 0, 0, 0, 0, 0};
    const int cout_of[7] = {a.g, a.g, CH, CH, CH, a.heads, a.cm};
    const int need_level[7] = {2, 2, 2, 3, 3, 4, 4};        // statistics of slot f exist from this level on
    for (int t = threadIdx.x; t < 7 * 16; t += BLOCK) {
        const int f = t >> 4, c = t & 15;
        const int layer = layer_of[f], o = 16 * tile_of[f] + c;
        float sc = 1.f, sh = 0.f;
        if (o < cout_of[f] && level >= need_level[f] && a.gamma[layer]) {        // null: layer absent (WeightNet-only use)
            sc = a.rstd[layer][o] * a.gamma[layer][o];
            sh = (a.b[layer][o] - a.mean[layer][o]) * sc + a.beta[layer][o];
        }
        cf[f][0][c] = sc;
        cf[f][1][c] = sh;
    }
}

__device__ __forceinline__ f32x4 bn_relu(f32x4 v, const float (*cf)[16], int g) {
    const float4 sc = ld4(&cf[0][4 * g]), sh = ld4(&cf[1][4 * g]);
    v[0] = fmaxf(v[0] * sc.x + sh.x, 0.f); v[1] = fmaxf(v[1] * sc.y + sh.y, 0.f);
    v[2] = fmaxf(v[2] * sc.z + sh.z, 0.f); v[3] = fmaxf(v[3] * sc.w + sh.w, 0.f);
    return v;
}
__device__ __forceinline__ f32x4 bn_only(f32x4 v, const float (*cf)[16], int g) {
    const float4 sc = ld4(&cf[0][4 * g]), sh = ld4(&cf[1][4 * g]);
    v[0] = v[0] * sc.x + sh.x; v[1] = v[1] * sc.y + sh.y; v[2] = v[2] * sc.z + sh.z; v[3] = v[3] * sc.w + sh.w;
    return v;
}

template <int LEVEL>
__global__ __launch_bounds__(BLOCK) void pcf_chain_kernel(const ChainArgs a) {
    __shared__ float red[NWAVE][3][2][16];
    __shared__ __align__(16) float cf[7][2][16];
    stage_frags(a, cf, LEVEL);
    __syncthreads();
    const int lane = lane_id(), wave = wave_id();
    const int p = lane & 15, g = lane >> 4;
    // weight fragments: wf[s] = W[o = 16*tile + p][c = 16*utile + 4*g + s]
    f32x4 w_pe[2], w_w1, w_g1[2], w_w2, w_g2, w_w3;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        w_pe[0][s] = wfrag(a.W[L_PE], a.g, a.cv, p, 4 * g + s);
        w_pe[1][s] = wfrag(a.W[L_PE], a.g, a.cv, 16 + p, 4 * g + s);
        w_w1[s] = wfrag(a.W[L_W1], CH, a.cv, p, 4 * g + s);
        w_g1[0][s] = LEVEL >= 2 ? wfrag(a.W[L_G1], CH, a.g, p, 4 * g + s) : 0.f;
        w_g1[1][s] = LEVEL >= 2 ? wfrag(a.W[L_G1], CH, a.g, p, 16 + 4 * g + s) : 0.f;
        w_w2[s] = LEVEL >= 2 ? wfrag(a.W[L_W2], CH, CH, p, 4 * g + s) : 0.f;
        w_g2[s] = LEVEL >= 3 ? wfrag(a.W[L_G2], a.heads, CH, p, 4 * g + s) : 0.f;
        w_w3[s] = LEVEL >= 3 ? wfrag(a.W[L_W3], a.cm, CH, p, 4 * g + s) : 0.f;
    }
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 s1[3] = {zero4, zero4, zero4}, s2[3] = {zero4, zero4, zero4};     // statistics of raw accumulators (bias added later)
    const int lead = (lane & ~15) | (p & ~(a.K - 1));
    const long long ntiles = a.E / 16;
    const long long tstride = (long long)gridDim.x * NWAVE;
    // Loads of tile t+1 (VI, neighbour index) and the gathered u row of tile t+1 are issued before the
    // MFMAs of tile t: one wave keeps two tiles of global loads in flight behind ~1000 cycles of matrix work.
    auto load_x = [&](long long tt) -> f32x4 {
        f32x4 xv = zero4;
        if (tt < ntiles && 4 * g < a.cv) {
            const float* q = a.vi + (size_t)(tt * 16 + p) * a.cv + 4 * g;
            if (a.vec_vi) { const float4 v = ld4(q); xv[0] = v.x; xv[1] = v.y; xv[2] = v.z; xv[3] = v.w; }
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) xv[r] = (4 * g + r < a.cv) ? q[r] : 0.f;
            }
        }
        return xv;
    };
    BatchWalk walk;
    walk.init(a.rows_per_batch);
    auto load_j = [&](long long tt) -> long long {       // called with increasing tt only
        if (LEVEL < 2 || tt >= ntiles) return -1;
        const int batch = walk.batch_of(tt * 16, p);
        if (g >= 2) return -1;
        const int64_t j = a.idx[tt * 16 + p];
        return (j >= 0 && j < a.N) ? (long long)batch * a.N + j : -1;
    };
    auto load_u = [&](long long row) -> f32x4 {
        f32x4 uv = zero4;
        if (LEVEL >= 2 && row >= 0) { const float4 v = ld4(a.u + (size_t)row * CH + 4 * g); uv[0] = v.x; uv[1] = v.y; uv[2] = v.z; uv[3] = v.w; }
        return uv;
    };
    long long t = (long long)blockIdx.x * NWAVE + wave;
    f32x4 x = load_x(t), x_next = load_x(t + tstride);
    f32x4 ucur = load_u(load_j(t));
    long long j_next = load_j(t + tstride);
    for (; t < ntiles; t += tstride) {
        const long long e = t * 16 + p;
        asm volatile("" ::: "memory");     // keeps the (loop-invariant) BN constants in LDS instead of 56 pinned VGPRs
        const f32x4 u_next = load_u(j_next);                       // gather for tile t+1
        const long long j_nn = load_j(t + 2 * tstride);            // index for tile t+2
        const f32x4 x_nn = load_x(t + 2 * tstride);                // input of tile t+2
        f32x4 pe0 = zero4, pe1 = zero4, a1 = zero4;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            pe0 = PCF_MFMA(w_pe[0][s], x[s], pe0);
            pe1 = PCF_MFMA(w_pe[1][s], x[s], pe1);
            a1 = PCF_MFMA(w_w1[s], x[s], a1);
        }
        if (LEVEL == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s1[0][r] += pe0[r]; s2[0][r] += pe0[r] * pe0[r];
                s1[1][r] += pe1[r]; s2[1][r] += pe1[r] * pe1[r];
                s1[2][r] += a1[r];  s2[2][r] += a1[r] * a1[r];
            }
            x = x_next; x_next = x_nn; ucur = u_next; j_next = j_nn;
            continue;
        }
        pe0 = bn_relu(pe0, cf[0], g);
        pe1 = bn_relu(pe1, cf[1], g);
        a1 = bn_relu(a1, cf[2], g);
        if (LEVEL == 2) {
            if (a.pe) {
                float* q = a.pe + (size_t)e * a.g;
                if ((a.g & 3) == 0) {
                    if (4 * g < a.g) st4(q + 4 * g, make_float4(pe0[0], pe0[1], pe0[2], pe0[3]));
                    if (16 + 4 * g < a.g) st4(q + 16 + 4 * g, make_float4(pe1[0], pe1[1], pe1[2], pe1[3]));
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (4 * g + r < a.g) q[4 * g + r] = pe0[r];
                        if (16 + 4 * g + r < a.g) q[16 + 4 * g + r] = pe1[r];
                    }
                }
            }
            if (a.a1 && g < 2) st4(a.a1 + (size_t)e * CH + 4 * g, make_float4(a1[0], a1[1], a1[2], a1[3]));
        }
        // g1: Wb.pe + u[idx[e]] - (same for the key edge);  w2
        f32x4 h1 = zero4, h1b = zero4, a2 = zero4;      // two accumulators halve the dependent-MFMA chain of g1
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            h1 = PCF_MFMA(w_g1[0][s], pe0[s], h1);
            h1b = PCF_MFMA(w_g1[1][s], pe1[s], h1b);
            a2 = PCF_MFMA(w_w2[s], a1[s], a2);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) h1[r] += h1b[r];
#pragma unroll
        for (int r = 0; r < 4; ++r) h1[r] += ucur[r];
        if (a.ukey) {
            // strided: key = maximum of the query over the neighbourhood; Wb . max pe here, Wa . max gx in ukey
            const f32x4 pm0 = nbr_max(pe0, a.K), pm1 = nbr_max(pe1, a.K);
            f32x4 hk = zero4, hkb = zero4;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                hk = PCF_MFMA(w_g1[0][s], pm0[s], hk);
                hkb = PCF_MFMA(w_g1[1][s], pm1[s], hkb);
            }
            f32x4 uk = zero4;
            if (g < 2) { const float4 v = ld4(a.ukey + (size_t)(e / a.K) * CH + 4 * g); uk[0] = v.x; uk[1] = v.y; uk[2] = v.z; uk[3] = v.w; }
#pragma unroll
            for (int r = 0; r < 4; ++r) h1[r] -= (hk[r] + hkb[r]) + uk[r];
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) h1[r] -= __shfl(h1[r], lead, WAVE);
        }
        if (LEVEL == 2) {
            // the raw accumulators of g1 / w2: passes 3, 4 and the fused backward restart from them
            if (g < 2) {
                if (a.h1_acc) st4(a.h1_acc + (size_t)e * CH + 4 * g, make_float4(h1[0], h1[1], h1[2], h1[3]));
                if (a.a2_acc) st4(a.a2_acc + (size_t)e * CH + 4 * g, make_float4(a2[0], a2[1], a2[2], a2[3]));
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s1[0][r] += h1[r]; s2[0][r] += h1[r] * h1[r];
                s1[1][r] += a2[r]; s2[1][r] += a2[r] * a2[r];
            }
            x = x_next; x_next = x_nn; ucur = u_next; j_next = j_nn;
            continue;
        }
        h1 = bn_relu(h1, cf[3], g);
        a2 = bn_relu(a2, cf[4], g);
        if (LEVEL == 3 && g < 2) {
            if (a.h1) st4(a.h1 + (size_t)e * CH + 4 * g, make_float4(h1[0], h1[1], h1[2], h1[3]));
            if (a.a2) st4(a.a2 + (size_t)e * CH + 4 * g, make_float4(a2[0], a2[1], a2[2], a2[3]));
        }
        f32x4 sc = zero4, wv = zero4;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            sc = PCF_MFMA(w_g2[s], h1[s], sc);
            wv = PCF_MFMA(w_w3[s], a2[s], wv);
        }
        if (LEVEL == 3) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s1[0][r] += sc[r]; s2[0][r] += sc[r] * sc[r];
                s1[1][r] += wv[r]; s2[1][r] += wv[r] * wv[r];
            }
            x = x_next; x_next = x_nn; ucur = u_next; j_next = j_nn;
            continue;
        }
        sc = bn_only(sc, cf[5], g);
#pragma unroll
        for (int r = 0; r < 4; ++r) sc[r] = 1.f / (1.f + __expf(-sc[r]));
        wv = bn_relu(wv, cf[6], g);
        {
            float* qs = a.score + (size_t)e * a.heads;
            float* qw = a.w + (size_t)e * a.cm;
            if ((a.heads & 3) == 0) { if (4 * g < a.heads) st4(qs + 4 * g, make_float4(sc[0], sc[1], sc[2], sc[3])); }
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (4 * g + r < a.heads) qs[4 * g + r] = sc[r];
            }
            if ((a.cm & 3) == 0) { if (4 * g < a.cm) st4(qw + 4 * g, make_float4(wv[0], wv[1], wv[2], wv[3])); }
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (4 * g + r < a.cm) qw[4 * g + r] = wv[r];
            }
        }
        x = x_next; x_next = x_nn; ucur = u_next; j_next = j_nn;
    }
    if (LEVEL == 4) return;
    // reduce the per-lane sums over the 16 edge lanes; lane p == 0 of group g then owns channels 4g..4g+3
#pragma unroll
    for (int gi = 0; gi < 3; ++gi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v1 = s1[gi][r], v2 = s2[gi][r];
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) { v1 += __shfl_xor(v1, off, WAVE); v2 += __shfl_xor(v2, off, WAVE); }
            if (p == 0) { red[wave][gi][0][4 * g + r] = v1; red[wave][gi][1][4 * g + r] = v2; }
        }
    __syncthreads();
    if (threadIdx.x < 96) {
        const int gi = threadIdx.x / 32, which = (threadIdx.x >> 4) & 1, c = threadIdx.x & 15;
        float tsum = 0.f;
#pragma unroll
        for (int w = 0; w < NWAVE; ++w) tsum += red[w][gi][which][c];
        a.part[(size_t)blockIdx.x * 96 + threadIdx.x] = tsum;
    }
}

// Passes 3 and 4 of the training forward when pass 2 kept the raw accumulators of g1 and w2: the top layers
// (g2, w3) are one 8-wide product away from them, so VI, the gathered term and the first two layers of each branch
// are not recomputed (8 matrix instructions per tile instead of 32).  STATS: batch statistics of the raw g2 / w3
// accumulators (64 B read per edge); else score and w (64 B read + 96 B written).
template <bool STATS>
__global__ __launch_bounds__(BLOCK) void pcf_chain_tail_kernel(const ChainArgs a) {
    __shared__ __align__(16) float cf[7][2][16];
    __shared__ float red[NWAVE][3][2][16];
    stage_frags(a, cf, STATS ? 3 : 4);
    if (STATS)
        for (int t = threadIdx.x; t < NWAVE * 96; t += BLOCK) (&red[0][0][0][0])[t] = 0.f;
    __syncthreads();
    const int lane = lane_id(), wave = wave_id();
    const int p = lane & 15, g = lane >> 4;
    f32x4 w_g2, w_w3;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        w_g2[s] = wfrag(a.W[L_G2], a.heads, CH, p, 4 * g + s);
        w_w3[s] = wfrag(a.W[L_W3], a.cm, CH, p, 4 * g + s);
    }
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 s1[2] = {zero4, zero4}, s2[2] = {zero4, zero4};
    const long long ntiles = a.E / 16;
    for (long long t = (long long)blockIdx.x * NWAVE + wave; t < ntiles; t += (long long)gridDim.x * NWAVE) {
        const long long e = t * 16 + p;
        f32x4 h1 = zero4, a2 = zero4;
        if (g < 2) {
            const float4 v1 = ld4(a.h1_acc + (size_t)e * CH + 4 * g), v2 = ld4(a.a2_acc + (size_t)e * CH + 4 * g);
            h1 = f32x4{v1.x, v1.y, v1.z, v1.w};
            a2 = f32x4{v2.x, v2.y, v2.z, v2.w};
        }
        h1 = bn_relu(h1, cf[3], g);
        a2 = bn_relu(a2, cf[4], g);
        if (g >= 2) { h1 = zero4; a2 = zero4; }          // channels 8..15 do not exist
        f32x4 sc = zero4, wv = zero4;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            sc = PCF_MFMA(w_g2[s], h1[s], sc);
            wv = PCF_MFMA(w_w3[s], a2[s], wv);
        }
        if (STATS) {
            s1[0] += sc; s2[0] += sc * sc;
            s1[1] += wv; s2[1] += wv * wv;
            continue;
        }
        sc = bn_only(sc, cf[5], g);
#pragma unroll
        for (int r = 0; r < 4; ++r) sc[r] = 1.f / (1.f + __expf(-sc[r]));
        wv = bn_relu(wv, cf[6], g);
        float* qs = a.score + (size_t)e * a.heads;
        float* qw = a.w + (size_t)e * a.cm;
        if ((a.heads & 3) == 0) { if (4 * g < a.heads) st4(qs + 4 * g, make_float4(sc[0], sc[1], sc[2], sc[3])); }
        else {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (4 * g + r < a.heads) qs[4 * g + r] = sc[r];
        }
        if ((a.cm & 3) == 0) { if (4 * g < a.cm) st4(qw + 4 * g, make_float4(wv[0], wv[1], wv[2], wv[3])); }
        else {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (4 * g + r < a.cm) qw[4 * g + r] = wv[r];
        }
    }
    if (!STATS) return;
#pragma unroll
    for (int gi = 0; gi < 2; ++gi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v1 = s1[gi][r], v2 = s2[gi][r];
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) { v1 += __shfl_xor(v1, off, WAVE); v2 += __shfl_xor(v2, off, WAVE); }
            if (p == 0) { red[wave][gi][0][4 * g + r] = v1; red[wave][gi][1][4 * g + r] = v2; }
        }
    __syncthreads();
    if (threadIdx.x < 96) {
        const int gi = threadIdx.x / 32, which = (threadIdx.x >> 4) & 1, c = threadIdx.x & 15;
        float tsum = 0.f;
#pragma unroll
        for (int w = 0; w < NWAVE; ++w) tsum += red[w][gi][which][c];
        a.part[(size_t)blockIdx.x * 96 + threadIdx.x] = tsum;
    }
}

// ---- WeightNet alone (PointConv family: layers.py:127-191 feeding pconv / pconv_linear) -------------------------
// The same three-layer branch (cin -> 8 -> 8 -> C_mid, Linear + BatchNorm + ReLU each) without a guidance branch:
// any neighbourhood structure (strided, transposed, any K), one lane group per edge row.
//   LEVEL 1: statistics of w1                (reads the input row)
//   LEVEL 2: statistics of w2, stores its raw accumulator a2_acc [E, 8]
//   LEVEL 3: statistics of w3 from a2_acc    LEVEL 4: w = relu(BN(w3)) from a2_acc (FROM_ACC) or from the input row
//            (inference: running statistics, one pass)
template <int LEVEL, bool FROM_ACC>
__global__ __launch_bounds__(BLOCK) void wn_chain_kernel(const ChainArgs a) {
    __shared__ float red[NWAVE][3][2][16];
    __shared__ __align__(16) float cf[7][2][16];
    stage_frags(a, cf, LEVEL);
    for (int t = threadIdx.x; t < NWAVE * 96; t += BLOCK) (&red[0][0][0][0])[t] = 0.f;
    __syncthreads();
    const int lane = lane_id(), wave = wave_id();
    const int p = lane & 15, g = lane >> 4;
    f32x4 w_w1, w_w2, w_w3;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        w_w1[s] = FROM_ACC ? 0.f : wfrag(a.W[L_W1], CH, a.cv, p, 4 * g + s);
        w_w2[s] = (FROM_ACC || LEVEL < 2) ? 0.f : wfrag(a.W[L_W2], CH, CH, p, 4 * g + s);
        w_w3[s] = LEVEL >= 3 ? wfrag(a.W[L_W3], a.cm, CH, p, 4 * g + s) : 0.f;
    }
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 s1 = zero4, s2 = zero4;
    const long long ntiles = a.E / 16;
    for (long long t = (long long)blockIdx.x * NWAVE + wave; t < ntiles; t += (long long)gridDim.x * NWAVE) {
        const long long e = t * 16 + p;
        f32x4 a2 = zero4;
        if (FROM_ACC) {
            if (g < 2) { const float4 v = ld4(a.a2_acc + (size_t)e * CH + 4 * g); a2 = f32x4{v.x, v.y, v.z, v.w}; }
        } else {
            f32x4 x = zero4;
            if (4 * g < a.cv) {
                const float* q = a.vi + (size_t)e * a.cv + 4 * g;
                if (a.vec_vi) { const float4 v = ld4(q); x = f32x4{v.x, v.y, v.z, v.w}; }
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[r] = (4 * g + r < a.cv) ? q[r] : 0.f;
                }
            }
            f32x4 a1 = zero4;
#pragma unroll
            for (int s = 0; s < 4; ++s) a1 = PCF_MFMA(w_w1[s], x[s], a1);
            if (LEVEL == 1) { s1 += a1; s2 += a1 * a1; continue; }
            a1 = bn_relu(a1, cf[2], g);
#pragma unroll
            for (int s = 0; s < 4; ++s) a2 = PCF_MFMA(w_w2[s], a1[s], a2);
            if (LEVEL == 2) {
                if (a.a2_acc && g < 2) st4(a.a2_acc + (size_t)e * CH + 4 * g, make_float4(a2[0], a2[1], a2[2], a2[3]));
                s1 += a2; s2 += a2 * a2;
                continue;
            }
        }
        a2 = bn_relu(a2, cf[4], g);
        if (g >= 2) a2 = zero4;                          // channels 8..15 do not exist
        f32x4 wv = zero4;
#pragma unroll
        for (int s = 0; s < 4; ++s) wv = PCF_MFMA(w_w3[s], a2[s], wv);
        if (LEVEL == 3) { s1 += wv; s2 += wv * wv; continue; }
        wv = bn_relu(wv, cf[6], g);
        float* qw = a.w + (size_t)e * a.cm;
        if ((a.cm & 3) == 0) { if (4 * g < a.cm) st4(qw + 4 * g, make_float4(wv[0], wv[1], wv[2], wv[3])); }
        else {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (4 * g + r < a.cm) qw[4 * g + r] = wv[r];
        }
    }
    if (LEVEL == 4) return;
    const int gi = LEVEL == 1 ? 2 : 1;                      // group of the partial list: w1 -> 2, w2 / w3 -> 1
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float v1 = s1[r], v2 = s2[r];
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) { v1 += __shfl_xor(v1, off, WAVE); v2 += __shfl_xor(v2, off, WAVE); }
        if (p == 0) { red[wave][gi][0][4 * g + r] = v1; red[wave][gi][1][4 * g + r] = v2; }
    }
    __syncthreads();
    if (threadIdx.x < 96) {
        const int q = threadIdx.x / 32, which = (threadIdx.x >> 4) & 1, c = threadIdx.x & 15;
        float tsum = 0.f;
#pragma unroll
        for (int w = 0; w < NWAVE; ++w) tsum += red[w][q][which][c];
        a.part[(size_t)blockIdx.x * 96 + threadIdx.x] = tsum;
    }
}

// Statistics of one pass: up to three groups of 16 channels; each group belongs to a layer at a channel
// offset.  One thread per (group, channel): fp64 sum over the workgroup partials.
struct FinGroup { float* mean; float* rstd; float* running_mean; float* running_var; const float* bias; int chan0; int count; };
struct FinArgs { FinGroup g[3]; const float* part; int nblocks; long long R; float eps; float momentum; };

__global__ __launch_bounds__(1024) void chain_finalize_kernel(const FinArgs f) {
    // 12 workgroups: (group, quarter of its 16 channels); 8 values (sum | second sum of 4 channels) x 128 slices of
    // the partial list each, so a slice walks 8-16 partials (3 workgroups x 32 slices took 8-10 us)
    __shared__ double sh[128][8];
    const int gq = blockIdx.x >> 2, quarter = blockIdx.x & 3;
    const int v = threadIdx.x & 7, slice = threadIdx.x >> 3;
    const int col = (v >> 2) * 16 + quarter * 4 + (v & 3);          // which * 16 + channel
    {
        double a0 = 0.0, a1 = 0.0;
        int p = slice;
        for (; p + 128 < f.nblocks; p += 256) {
            a0 += (double)f.part[(size_t)p * 96 + gq * 32 + col];
            a1 += (double)f.part[(size_t)(p + 128) * 96 + gq * 32 + col];
        }
        if (p < f.nblocks) a0 += (double)f.part[(size_t)p * 96 + gq * 32 + col];
        sh[slice][v] = a0 + a1;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int gi = gq, c = quarter * 4 + threadIdx.x;
        double sums[2] = {0.0, 0.0};
#pragma unroll
        for (int which = 0; which < 2; ++which)
            for (int sl = 0; sl < 128; ++sl) sums[which] += sh[sl][which * 4 + threadIdx.x];
        const FinGroup& g = f.g[gi];
        if (g.mean && c < g.count) {
            const int o = g.chan0 + c;
            const double n = (double)f.R;
            const double m0 = sums[0] / n;                  // statistics of z - bias
            double var = sums[1] / n - m0 * m0;
            if (var < 0.0) var = 0.0;
            const double mean = m0 + (double)g.bias[o];
            g.mean[o] = (float)mean;
            g.rstd[o] = (float)(1.0 / sqrt(var + (double)f.eps));
            if (g.running_mean) {
                const double unb = f.R > 1 ? var * n / (n - 1.0) : var;
                g.running_mean[o] = (float)((1.0 - f.momentum) * g.running_mean[o] + f.momentum * mean);
                g.running_var[o] = (float)((1.0 - f.momentum) * g.running_var[o] + f.momentum * unb);
            }
        }
    }
}

}  // namespace pcf

extern "C" {

size_t pcf_hip_pcf_chain_workspace_bytes(void) { return (size_t)2048 * 96 * 4 + 1024; }

// stats [12][64] floats (device): mean of layer l at stats + l*64, rstd at stats + (6 + l)*64, l in the
// order mlp_conv, g1, g2, w1, w2, w3.  Training (batch_stats != 0): computed here and the running
// statistics (nullable) updated; inference: the caller fills them from the running statistics.
static int chain_forward_impl(const float* ukey, const float* vi, const int64_t* idx, const float* u, long long E, long long rows_per_batch,
                              int N, int K, int cv, int g, int heads, int cm, const float* const* W, const float* const* b,
                              const float* const* gamma, const float* const* beta, float* const* running_mean,
                              float* const* running_var, float eps, float momentum, int batch_stats, float* stats,
                              float* pe, float* a1, float* h1, float* a2, float* h1_acc, float* a2_acc, float* score,
                              float* w, void* workspace, size_t workspace_bytes, void* stream) {
    using namespace pcf;
    if (ukey && (K < 2 || !aligned16(ukey)))
        return fail(PCF_E_UNSUPPORTED, "pcf_chain (maximum key): K >= 2 and a 16-byte aligned ukey (K=%d)", K);
    PCF_REQUIRE(E >= 0 && rows_per_batch > 0 && N >= 0, "pcf_chain: bad sizes");
    if (cv < 1 || cv > CV || g < 1 || g > CG || heads < 1 || heads > CHD || cm < 1 || cm > CMX)
        return fail(PCF_E_UNSUPPORTED, "pcf_chain: widths outside the fused kernel (cv=%d<=12, g=%d<=32, heads=%d<=8, cm=%d<=16)", cv, g, heads, cm);
    if (K < 1 || K > 16 || (K & (K - 1)) != 0 || E % 16 != 0 || E % rows_per_batch != 0 || rows_per_batch < 16)
        return fail(PCF_E_UNSUPPORTED, "pcf_chain: K must be a power of two <= 16, the edge count a multiple of 16 and >= 16 edges per batch (K=%d)", K);
    if (E == 0) return ok();
    PCF_REQUIRE(vi && idx && u && W && b && gamma && beta && stats && score && w, "pcf_chain: null pointer");
    PCF_REQUIRE(aligned16(score) && aligned16(w) && aligned16(pe) && aligned16(a1) && aligned16(h1) && aligned16(a2) &&
                    aligned16(u) && aligned16(h1_acc) && aligned16(a2_acc), "pcf_chain: buffers must be 16-byte aligned");       // null activation pointers: not written
    PCF_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= pcf_hip_pcf_chain_workspace_bytes(),
                "pcf_chain: workspace too small or misaligned");
    hipStream_t s = (hipStream_t)stream;
    ChainArgs a{};
    a.vi = vi; a.idx = idx; a.u = u; a.E = E; a.rows_per_batch = rows_per_batch; a.N = N; a.K = K;
    a.cv = cv; a.g = g; a.heads = heads; a.cm = cm;
    for (int l = 0; l < 6; ++l) {
        PCF_REQUIRE(W[l] && b[l] && gamma[l] && beta[l], "pcf_chain: null parameter of layer %d", l);
        a.W[l] = W[l]; a.b[l] = b[l]; a.gamma[l] = gamma[l]; a.beta[l] = beta[l];
        a.mean[l] = stats + l * 64; a.rstd[l] = stats + (6 + l) * 64;
    }
    a.pe = pe; a.a1 = a1; a.h1 = h1; a.a2 = a2; a.score = score; a.w = w;
    a.h1_acc = h1_acc; a.a2_acc = a2_acc; a.ukey = ukey;
    a.part = static_cast<float*>(workspace);
    a.vec_vi = (cv % 4 == 0) && aligned16(vi);
    // passes 3 and 4 restart from the accumulators pass 2 stores, unless the caller wants h1 / a2 themselves
    const bool from_acc = h1_acc && a2_acc && !h1 && !a2;
    if (batch_stats) {
        for (int pass = 0; pass < 3; ++pass) {
            int grid;
            if (pass == 0) { grid = chain_grid(E); hipLaunchKernelGGL(pcf_chain_kernel<1>, dim3(grid), dim3(BLOCK), 0, s, a); }
            else if (pass == 1) { grid = std::min<long long>(2048, std::max<long long>(1, (E / 16 + NWAVE - 1) / NWAVE)); hipLaunchKernelGGL(pcf_chain_kernel<2>, dim3(grid), dim3(BLOCK), 0, s, a); }
            else if (from_acc) { grid = chain_grid(E); hipLaunchKernelGGL(pcf_chain_tail_kernel<true>, dim3(grid), dim3(BLOCK), 0, s, a); }
            else { grid = chain_grid(E); hipLaunchKernelGGL(pcf_chain_kernel<3>, dim3(grid), dim3(BLOCK), 0, s, a); }
            if (int e = check_launch("pcf_chain pass")) return e;
            FinArgs f{};
            f.part = a.part; f.nblocks = grid; f.R = E; f.eps = eps; f.momentum = momentum;
            auto set = [&](int gi, int layer, int chan0, int count) {
                f.g[gi].mean = stats + layer * 64; f.g[gi].rstd = stats + (6 + layer) * 64;
                f.g[gi].running_mean = running_mean ? running_mean[layer] : nullptr;
                f.g[gi].running_var = running_var ? running_var[layer] : nullptr;
                f.g[gi].bias = b[layer]; f.g[gi].chan0 = chan0; f.g[gi].count = count;
            };
            if (pass == 0) { set(0, L_PE, 0, std::min(g, 16)); set(1, L_PE, 16, std::max(g - 16, 0)); set(2, L_W1, 0, CH); }
            else if (pass == 1) { set(0, L_G1, 0, CH); set(1, L_W2, 0, CH); }
            else { set(0, L_G2, 0, heads); set(1, L_W3, 0, cm); }
            hipLaunchKernelGGL(chain_finalize_kernel, dim3(12), dim3(1024), 0, s, f);
            if (int e = check_launch("pcf_chain finalize")) return e;
        }
    }
    if (batch_stats && from_acc) {
        const long long tiles = E / 16;
        const int grid = (int)std::max<long long>(1, std::min<long long>((tiles + NWAVE - 1) / NWAVE, 2048));
        hipLaunchKernelGGL(pcf_chain_tail_kernel<false>, dim3(grid), dim3(BLOCK), 0, s, a);
    } else {
        hipLaunchKernelGGL(pcf_chain_kernel<4>, dim3(chain_grid(E)), dim3(BLOCK), 0, s, a);
    }
    return check_launch("pcf_chain final pass");
}

int pcf_hip_pcf_chain_forward(const float* vi, const int64_t* idx, const float* u, long long E, long long rows_per_batch,
                              int N, int K, int cv, int g, int heads, int cm, const float* const* W, const float* const* b,
                              const float* const* gamma, const float* const* beta, float* const* running_mean,
                              float* const* running_var, float eps, float momentum, int batch_stats, float* stats,
                              float* pe, float* a1, float* h1, float* a2, float* h1_acc, float* a2_acc, float* score,
                              float* w, void* workspace, size_t workspace_bytes, void* stream) {
    return chain_forward_impl(nullptr, vi, idx, u, E, rows_per_batch, N, K, cv, g, heads, cm, W, b, gamma, beta, running_mean,
                              running_var, eps, momentum, batch_stats, stats, pe, a1, h1, a2, h1_acc, a2_acc, score, w, workspace,
                              workspace_bytes, stream);
}

// Strided layers: the same graph with key = maximum of the query over the neighbourhood (layers.py:372-375).  ukey
// [E / K, 8] = Wa . max_k guidance_x[idx[n, k]] per centre (K >= 2); everything else as pcf_hip_pcf_chain_forward.
int pcf_hip_pcf_chain_forward_maxkey(const float* ukey, const float* vi, const int64_t* idx, const float* u, long long E,
                                     long long rows_per_batch, int N, int K, int cv, int g, int heads, int cm,
                                     const float* const* W, const float* const* b, const float* const* gamma,
                                     const float* const* beta, float* const* running_mean, float* const* running_var,
                                     float eps, float momentum, int batch_stats, float* stats, float* pe, float* a1, float* h1,
                                     float* a2, float* h1_acc, float* a2_acc, float* score, float* w, void* workspace,
                                     size_t workspace_bytes, void* stream) {
    if (!ukey) return pcf::fail(PCF_E_BADARG, "pcf_chain_forward_maxkey: ukey is null");
    return chain_forward_impl(ukey, vi, idx, u, E, rows_per_batch, N, K, cv, g, heads, cm, W, b, gamma, beta, running_mean,
                              running_var, eps, momentum, batch_stats, stats, pe, a1, h1, a2, h1_acc, a2_acc, score, w, workspace,
                              workspace_bytes, stream);
}

// WeightNet alone.  Layer order w1, w2, w3; stats [12][64] laid out as for the full chain (mean of layer l at
// stats + 64*l with l = 3, 4, 5 for w1, w2, w3; rstd at stats + 64*(6+l)).  cin <= 12, hidden widths 8, cm <= 16,
// E % 16 == 0.  a2_acc (nullable in inference) receives the raw accumulator of w2 for the fused backward.
int pcf_hip_weightnet_chain_forward(const float* x, long long E, int cin, int cm, const float* const* W,
                                    const float* const* b, const float* const* gamma, const float* const* beta,
                                    float* const* running_mean, float* const* running_var, float eps, float momentum,
                                    int batch_stats, float* stats, float* a2_acc, float* w, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    using namespace pcf;
    PCF_REQUIRE(E >= 0, "weightnet_chain: bad sizes");
    if (cin < 1 || cin > CV || cm < 1 || cm > CMX || E % 16 != 0)
        return fail(PCF_E_UNSUPPORTED, "weightnet_chain: needs cin <= 12, C_mid <= 16 and a multiple of 16 edges (cin=%d cm=%d)", cin, cm);
    if (E == 0) return ok();
    PCF_REQUIRE(x && W && b && gamma && beta && stats && w, "weightnet_chain: null pointer");
    PCF_REQUIRE(!batch_stats || a2_acc, "weightnet_chain: training needs the accumulator buffer");
    PCF_REQUIRE(aligned16(w) && aligned16(a2_acc), "weightnet_chain: buffers must be 16-byte aligned");
    PCF_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= pcf_hip_pcf_chain_workspace_bytes(),
                "weightnet_chain: workspace too small or misaligned");
    hipStream_t s = (hipStream_t)stream;
    ChainArgs a{};
    a.vi = x; a.E = E; a.rows_per_batch = E; a.cv = cin; a.g = 0; a.heads = 0; a.cm = cm; a.K = 1;
    const int layers[3] = {L_W1, L_W2, L_W3};
    for (int i = 0; i < 3; ++i) {
        const int l = layers[i];
        PCF_REQUIRE(W[i] && b[i] && gamma[i] && beta[i], "weightnet_chain: null parameter of layer %d", i);
        a.W[l] = W[i]; a.b[l] = b[i]; a.gamma[l] = gamma[i]; a.beta[l] = beta[i];
    }
    for (int l = 0; l < 6; ++l) { a.mean[l] = stats + l * 64; a.rstd[l] = stats + (6 + l) * 64; }
    a.a2_acc = a2_acc; a.w = w;
    a.part = static_cast<float*>(workspace);
    a.vec_vi = (cin % 4 == 0) && aligned16(x);
    const int grid = chain_grid(E);
    if (!batch_stats) {
        hipLaunchKernelGGL((wn_chain_kernel<4, false>), dim3(grid), dim3(BLOCK), 0, s, a);
        return check_launch("weightnet_chain inference pass");
    }
    for (int pass = 0; pass < 3; ++pass) {
        if (pass == 0) hipLaunchKernelGGL((wn_chain_kernel<1, false>), dim3(grid), dim3(BLOCK), 0, s, a);
        else if (pass == 1) hipLaunchKernelGGL((wn_chain_kernel<2, false>), dim3(grid), dim3(BLOCK), 0, s, a);
        else hipLaunchKernelGGL((wn_chain_kernel<3, true>), dim3(grid), dim3(BLOCK), 0, s, a);
        if (int e = check_launch("weightnet_chain pass")) return e;
        FinArgs f{};
        f.part = a.part; f.nblocks = grid; f.R = E; f.eps = eps; f.momentum = momentum;
        const int i = pass, l = layers[pass];
        const int gi = pass == 0 ? 2 : 1;
        f.g[gi].mean = stats + l * 64; f.g[gi].rstd = stats + (6 + l) * 64;
        f.g[gi].running_mean = running_mean ? running_mean[i] : nullptr;
        f.g[gi].running_var = running_var ? running_var[i] : nullptr;
        f.g[gi].bias = b[i]; f.g[gi].chan0 = 0; f.g[gi].count = pass == 2 ? cm : CH;
        hipLaunchKernelGGL(chain_finalize_kernel, dim3(12), dim3(1024), 0, s, f);
        if (int e = check_launch("weightnet_chain finalize")) return e;
    }
    hipLaunchKernelGGL((wn_chain_kernel<4, true>), dim3(std::min<long long>(2048, std::max<long long>(1, (E / 16 + NWAVE - 1) / NWAVE))),
                       dim3(BLOCK), 0, s, a);
    return check_launch("weightnet_chain final pass");
}

}  // extern "C"
