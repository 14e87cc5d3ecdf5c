// Fused backward of the per-edge graph of a PCFLayer (self neighbourhoods, training-mode BatchNorm) on gfx950.
//
// Adjoint of edge_chain.hip (layers.py:361-384, MultiHeadGuidance :47-68, WeightNet :163-171).  A BatchNorm in
// training mode couples every edge: with g = dy * act'(.), xhat the normalised pre-activation,
//     dz = gamma * rstd * (g - mean(g) - xhat * mean(g * xhat)),    dbeta = sum g,   dgamma = sum g * xhat
// so layer l's dz needs two global sums that depend on the dz of the layer above it -- three dependent
// reductions down each branch (the last one is removed by linearity, below).  Layer-at-a-time execution (edge_mlp_mfma.hip) reads and writes every
// [E, 8..32] activation and gradient twice per layer (~2 kB per edge).  Here each of three passes recomputes what
// it needs and walks the gradient down to the first layer whose sums are still unknown.  The forward keeps ONE
// thing per branch: the raw 8-channel accumulator of the middle layer (g1 / w2: 2 x 32 B per edge, written by
// pass 2 of edge_chain.hip).  From it the top layer (g2 / w3) is one 8-wide product away, so passes 1 and 2 never
// touch VI, the gathered term or the key subtraction; the first layers (mlp_conv / w1) are recomputed from the
// 48-byte VI row only where their masks and inputs are needed (pass 3):
//     pass 1: sums of g2 and w3                                  (reads the two accumulators, dscore, dw)
//     pass 2: dz of g2, w3 -> dW of g2, w3;  g = dh1, da2 masked by the ReLU of g1, w2 -> stored (2 x 32 B per
//             edge) with their sums: the top layers are finished here and never touched again
//     pass 3: dz of g1, w2 (from the stored g), their dW and the gradient of the gathered term u (row-contiguous
//             float atomics); g = dpe, da1 masked by the ReLU of mlp_conv, w1 with their sums.  The dz of these
//             first layers, dz = g.SC + acc.D1 + D0, is LINEAR in what the sums decide (D0, D1), and so is their
//             dW = sum dz (x) VI = SC.[sum g (x) VI] + D1.[W.(sum VI (x) VI)] + D0.[sum VI]: the pass accumulates
//             the three bracketed moments (outer products on the matrix cores, operands transposed through LDS),
//             a one-workgroup kernel combines them once the sums are known.  There is no fourth pass.
// 84 matrix instructions per 16 edges over the three passes instead of 208 with four full recomputes; 160-190 B
// read per edge and pass, 64 B written once.  Bias gradients of a Linear that feeds a training-mode
// BatchNorm are identically zero (the mean subtraction cancels them) and are written as zeros.
//
// Matrix-core formulation as in edge_chain.hip (transposed, 16 edges per tile, lane (p = l & 15, g = l >> 4)
// holds channels 4g..4g+3 of edge p): backward products dIn[c][p] = sum_o W[o][c] dz[o][p] use the weight
// fragment A[i = c][k <-> o = 4k + s], so a dz accumulator is again directly the B operand.  The 13 weight
// fragments (8 forward, 5 transposed) and the BatchNorm constants live in LDS, not in registers.
#include <algorithm>
#include <atomic>
#include <cstdlib>

#include "edge_chain.h"

// fused multiply-adds in the per-element BatchNorm algebra (the library default is contraction off)
#pragma clang fp contract(fast)

namespace pcf {

constexpr int TT = 20;              // row stride of the 16x16 transposition tiles ([edge][channel], 16-byte aligned rows)
constexpr int NFRAG = 13;
constexpr int NSLOT = 7;            // output-tile slots: pe lo, pe hi, w1, g1, w2, g2, w3
constexpr int NDW = 8;              // 16x16 tiles a workgroup of pass 3 accumulates (see chain_bwd_reduce_kernel)
// per-channel constants of a slot: pre-activation = acc * SC + SH (acc = raw accumulator, bias folded in),
// dz = g * SC + acc * D1 + D0  (the BatchNorm backward written in terms of the raw accumulator, see stage_consts)
enum { K_SC = 0, K_SH = 1, K_D0 = 2, K_D1 = 3, NCONST = 4 };
enum { S_PE0 = 0, S_PE1 = 1, S_W1 = 2, S_G1 = 3, S_W2 = 4, S_G2 = 5, S_W3 = 6 };

struct ChainBwdArgs {
    ChainArgs f;                // forward description (activation pointers unused)
    const float* dscore;        // [E, heads]
    const float* dw;            // [E, cm]
    const float* h1_acc;        // [E, 8] raw accumulator of g1 (after the key subtraction, bias not added)
    const float* a2_acc;        // [E, 8] raw accumulator of w2
    const float* gmean[6];      // mean over edges of g          (device [64] per layer, filled pass by pass)
    const float* gxmean[6];     // mean over edges of g * xhat
    float* gh1;                 // [E, 8] dh1 * [h1 > 0], written by pass 2, read by pass 3 (workspace)
    float* ga2;                 // [E, 8] da2 * [a2 > 0]
    float* part_top;            // [blocks][2][256] dW tiles of g2 / w3 from pass 2
    float* du;                  // [B*N, 8], zeroed by the host; float atomics
    float* dukey;               // strided (f.ukey != null): [centres, 8] gradient of ukey = - sum over the neighbourhood of dz(g1)
    int wn_only;                // WeightNet alone: every workgroup runs that branch (and accumulates the VI' moments)
    float* part;                // [blocks][NDW][256] dW / moment tiles of pass 3
    float* part_sums;           // [blocks][96] per-channel sums of the running pass
};

// constants of slot s, channel c of the tile.  With rs = rstd, x0 = (b - mean) * rs, xhat = acc * rs + x0:
//   pre = xhat * gamma + beta = acc * (rs * gamma) + (x0 * gamma + beta)
//   dz  = gamma * rs * (g - mean(g) - xhat * mean(g xhat)) = g * SC + acc * D1 + D0
//         with D1 = -SC * mean(g xhat) * rs,  D0 = -SC * (mean(g) + mean(g xhat) * x0)
// so neither xhat nor the BatchNorm statistics are touched per element.
__device__ __forceinline__ void stage_consts(const ChainBwdArgs& a, float (*cf)[NCONST][16], int level) {
    const int layer_of[NSLOT] = {L_PE, L_PE, L_W1, L_G1, L_W2, L_G2, L_W3};
    const int tile_of[NSLOT] = {0, 1, 0, 0, 0, 0, 0};
    const int cout_of[NSLOT] = {a.f.g, a.f.g, CH, CH, CH, a.f.heads, a.f.cm};
    const int need_level[NSLOT] = {4, 4, 4, 3, 3, 2, 2};          // the sums of slot s exist from this pass on
    for (int t = threadIdx.x; t < NSLOT * 16; t += BLOCK) {
        const int s = t >> 4, c = t & 15;
        const int layer = layer_of[s], o = 16 * tile_of[s] + c;
        float v[NCONST] = {0.f, 0.f, 0.f, 0.f};
        if (o < cout_of[s] && a.f.gamma[layer]) {          // null: layer absent (WeightNet-only use)
            const float rs = a.f.rstd[layer][o], x0 = (a.f.b[layer][o] - a.f.mean[layer][o]) * rs;
            v[K_SC] = rs * a.f.gamma[layer][o];
            v[K_SH] = x0 * a.f.gamma[layer][o] + a.f.beta[layer][o];
            if (level >= need_level[s]) {
                const float gm = a.gmean[layer][o], gxm = a.gxmean[layer][o];
                v[K_D1] = -v[K_SC] * gxm * rs;
                v[K_D0] = -v[K_SC] * (gm + gxm * x0);
            }
        }
#pragma unroll
        for (int k = 0; k < NCONST; ++k) cf[s][k][c] = v[k];
    }
}

// weight fragments by lane: wl[f][lane] = the four A-operand values (contraction steps s = 0..3) of fragment f
__device__ __forceinline__ void stage_weights(const ChainBwdArgs& a, float4* wl) {
    const ChainArgs& f = a.f;
    for (int t = threadIdx.x; t < NFRAG * WAVE; t += BLOCK) {
        const int fr = t / WAVE, l = t % WAVE, p = l & 15, g = l >> 4;
        float v[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = 4 * g + s;
            switch (fr) {
                // forward: A[o = p][c = k]
                case 0: v[s] = wfrag(f.W[L_PE], f.g, f.cv, p, k); break;
                case 1: v[s] = wfrag(f.W[L_PE], f.g, f.cv, 16 + p, k); break;
                case 2: v[s] = wfrag(f.W[L_W1], CH, f.cv, p, k); break;
                case 3: v[s] = wfrag(f.W[L_G1], CH, f.g, p, k); break;
                case 4: v[s] = wfrag(f.W[L_G1], CH, f.g, p, 16 + k); break;
                case 5: v[s] = wfrag(f.W[L_W2], CH, CH, p, k); break;
                case 6: v[s] = wfrag(f.W[L_G2], f.heads, CH, p, k); break;
                case 7: v[s] = wfrag(f.W[L_W3], f.cm, CH, p, k); break;
                // transposed: A[c = p][o = k]
                case 8: v[s] = wfrag(f.W[L_G2], f.heads, CH, k, p); break;
                case 9: v[s] = wfrag(f.W[L_W3], f.cm, CH, k, p); break;
                case 10: v[s] = wfrag(f.W[L_G1], CH, f.g, k, p); break;
                case 11: v[s] = wfrag(f.W[L_G1], CH, f.g, k, 16 + p); break;
                default: v[s] = wfrag(f.W[L_W2], CH, CH, k, p); break;
            }
        }
        wl[t] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

__device__ __forceinline__ f32x4 to_v4(float4 v) { return f32x4{v.x, v.y, v.z, v.w}; }

// acc += W_fragment . operand  (four contraction steps)
__device__ __forceinline__ f32x4 mm(const float4* wl, int frag, int lane, f32x4 operand, f32x4 acc) {
    const f32x4 w = to_v4(wl[frag * WAVE + lane]);
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = PCF_MFMA(w[s], operand[s], acc);
    return acc;
}

// pre-activation (BatchNorm applied) of a raw accumulator tile
__device__ __forceinline__ f32x4 pre_of(f32x4 acc, const float (*k)[16], int g) {
    const f32x4 sc = to_v4(ld4(&k[K_SC][4 * g])), sh = to_v4(ld4(&k[K_SH][4 * g]));
    return acc * sc + sh;
}
__device__ __forceinline__ f32x4 relu4(f32x4 v) {
    return f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
}
__device__ __forceinline__ f32x4 mask_pos(f32x4 d, f32x4 pre) {
    return f32x4{pre[0] > 0.f ? d[0] : 0.f, pre[1] > 0.f ? d[1] : 0.f, pre[2] > 0.f ? d[2] : 0.f, pre[3] > 0.f ? d[3] : 0.f};
}
// dz of a BatchNorm layer from g = dy * act' and the raw accumulator; zero outside the layer's channels
// (all constants are 0 there)
__device__ __forceinline__ f32x4 bn_dz(f32x4 gr, f32x4 acc, const float (*k)[16], int g) {
    const f32x4 sc = to_v4(ld4(&k[K_SC][4 * g])), d0 = to_v4(ld4(&k[K_D0][4 * g])), d1 = to_v4(ld4(&k[K_D1][4 * g]));
    return gr * sc + (acc * d1 + d0);
}

// one 16x16 tile (lane (p, g): channels 4g..4g+3 of edge p) -> LDS [edge][channel], one 16-byte store per lane
__device__ __forceinline__ void put_tile(float* buf, f32x4 v, int p, int g) {
    st4(buf + p * TT + 4 * g, make_float4(v[0], v[1], v[2], v[3]));
}
// dW tile += sum over the 16 edges of dz[o][e] * in[c][e]: operand lane (i = l & 15, k = l >> 4) of step s holds
// element [channel i][edge 4s + k]
__device__ __forceinline__ f32x4 outer(const float* dzbuf, const float* inbuf, int p, int g, f32x4 acc) {
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = PCF_MFMA(dzbuf[(4 * s + g) * TT + p], inbuf[(4 * s + g) * TT + p], acc);
    return acc;
}

// Tile loaders shared by the two branches.
struct TileIO {
    const ChainBwdArgs& a;
    int p, g;
    long long ntiles;
    __device__ __forceinline__ f32x4 load_x(long long tt) const {
        const ChainArgs& f = a.f;
        f32x4 xv = {0.f, 0.f, 0.f, 0.f};
        if (tt < ntiles && 4 * g < f.cv) {
            const float* q = f.vi + (size_t)(tt * 16 + p) * f.cv + 4 * g;
            if (f.vec_vi) xv = to_v4(ld4(q));
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) xv[r] = (4 * g + r < f.cv) ? q[r] : 0.f;
            }
        }
        return xv;
    }
    __device__ __forceinline__ f32x4 load_grad(const float* base, long long tt, int C) const {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (tt < ntiles && 4 * g < C) {
            const float* q = base + (size_t)(tt * 16 + p) * C + 4 * g;
            if ((C & 3) == 0) v = to_v4(ld4(q));
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (4 * g + r < C) ? q[r] : 0.f;
            }
        }
        return v;
    }
};

// The two branches of the edge graph share nothing but the VI row, so every pass runs them in DIFFERENT
// workgroups of one launch (5 of every 8 workgroups take the guidance branch, which has 5/8 of the matrix work):
// each wave then holds the live state of one branch only -- about half the registers, twice the resident waves.
//
// Guidance branch: VI -> pe (<= 32) -> g1 (8, + gathered u, - key) -> g2 (heads, sigmoid).
//   sums: groups {L1: g2 | L2: g1 | L3: pe lo, pe hi};   L2: dW tile of g2;   L3: tiles 0, 1, 3, 4, 6 and du.
template <int LEVEL>
__device__ __forceinline__ void guidance_branch(const ChainBwdArgs& a, const float (*cf)[NCONST][16], const float4* wl,
                                                float* red, float* red_s, float* red_top, float* tb, int* gi,
                                                long long t0, long long tstride) {
    const ChainArgs& f = a.f;
    const int lane = lane_id(), wave = wave_id(), p = lane & 15, g = lane >> 4;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const long long ntiles = f.E / 16;
    const TileIO io{a, p, g, ntiles};
    f32x4 s1[2] = {zero4, zero4}, s2[2] = {zero4, zero4};
    f32x4 accw[5] = {zero4, zero4, zero4, zero4, zero4};          // pe lo, pe hi, g1 (in lo), g1 (in hi), g2
    const bool first = (p & (f.K - 1)) == 0;
    BatchWalk walk;
    walk.init(f.rows_per_batch);
    auto load_j = [&](long long tt) -> long long {               // called with increasing tt only; pass 3 (du rows)
        const int batch = walk.batch_of(tt * 16, p);
        const int64_t j = f.idx[tt * 16 + p];
        return (j >= 0 && j < f.N) ? (long long)batch * f.N + j : -1;
    };
    for (long long t = t0; t < ntiles; t += tstride) {
        asm volatile("" ::: "memory");          // LDS-resident weights / constants are re-read per tile, not hoisted into VGPRs
        const f32x4 ac_h1 = io.load_grad(a.h1_acc, t, CH);
        if (LEVEL <= 2) {
            // ---- the top layer: g2 from the stored accumulator of g1 ----
            const f32x4 dsc = io.load_grad(a.dscore, t, f.heads);
            const f32x4 y_h1 = relu4(pre_of(ac_h1, cf[S_G1], g));
            const f32x4 ac_sc = mm(wl, 6, lane, y_h1, zero4);
            const f32x4 pre_sc = pre_of(ac_sc, cf[S_G2], g);
            f32x4 g_sc;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sg = 1.f / (1.f + __expf(-pre_sc[r]));
                g_sc[r] = dsc[r] * sg * (1.f - sg);
            }
            if (LEVEL == 1) {
                s1[0] += g_sc; s2[0] += g_sc * ac_sc;
            } else {
                const f32x4 dz_sc = bn_dz(g_sc, ac_sc, cf[S_G2], g);
                put_tile(tb + 0 * 16 * TT, dz_sc, p, g);
                put_tile(tb + 1 * 16 * TT, y_h1, p, g);
                accw[4] = outer(tb + 0 * 16 * TT, tb + 1 * 16 * TT, p, g, accw[4]);
                const f32x4 g_h1 = mask_pos(mm(wl, 8, lane, dz_sc, zero4), y_h1);
                if (g < 2) st4(a.gh1 + (size_t)(t * 16 + p) * CH + 4 * g, make_float4(g_h1[0], g_h1[1], g_h1[2], g_h1[3]));
                s1[0] += g_h1; s2[0] += g_h1 * ac_h1;
            }
            continue;
        }
        // ---- pass 3: from the stored g of g1 down through the positional encoding ----
        const f32x4 g_h1 = io.load_grad(a.gh1, t, CH);
        const f32x4 x = io.load_x(t);
        const long long j_cur = load_j(t);
        const f32x4 ac_pe0 = mm(wl, 0, lane, x, zero4);
        const f32x4 ac_pe1 = mm(wl, 1, lane, x, zero4);
        const f32x4 y_pe0 = relu4(pre_of(ac_pe0, cf[S_PE0], g));
        const f32x4 y_pe1 = relu4(pre_of(ac_pe1, cf[S_PE1], g));
        f32x4 dq = bn_dz(g_h1, ac_h1, cf[S_G1], g);
        // z[k] = q[k] - q[key] + b  =>  dq[k] = dz[k] - [k is the key] * (sum over the neighbourhood)
        f32x4 tot4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float tot = dq[r];
            for (int off = 1; off < f.K; off <<= 1) tot += __shfl_xor(tot, off, WAVE);
            tot4[r] = tot;
            if (first && !f.ukey) dq[r] -= tot;
        }
        // strided: key = max over the neighbourhood.  z[k] = Wb (pe[k] - pemax) + u[idx k] - ukey + b: the gathered half's key
        // gradient leaves as dukey, the positional half's goes to the FIRST edge attaining the maximum of each channel
        f32x4 pm0 = zero4, pm1 = zero4, kf0 = zero4, kf1 = zero4;
        if (f.ukey) {
            if (first && g < 2) st4(a.dukey + (size_t)((t * 16 + p) / f.K) * CH + 4 * g, make_float4(-tot4[0], -tot4[1], -tot4[2], -tot4[3]));
            pm0 = nbr_max(y_pe0, f.K); pm1 = nbr_max(y_pe1, f.K);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float c0 = y_pe0[r] == pm0[r] ? (float)p : 16.f, c1 = y_pe1[r] == pm1[r] ? (float)p : 16.f;
                kf0[r] = nbr_min(c0, f.K) == (float)p ? 1.f : 0.f;
                kf1[r] = nbr_min(c1, f.K) == (float)p ? 1.f : 0.f;
            }
        }
        {
            put_tile(tb + 0 * 16 * TT, dq, p, g);
            put_tile(tb + 1 * 16 * TT, y_pe0 - pm0, p, g);
            put_tile(tb + 2 * 16 * TT, y_pe1 - pm1, p, g);
            // gradient of the gathered term: consecutive lanes cover the 8 consecutive channels of one row of du
            if (g == 0) gi[p] = (int)j_cur;
            for (int e = lane; e < 16 * CH; e += WAVE) {
                const int r = e / CH, o = e - r * CH;
                const int tgt = gi[r];
                if (tgt >= 0) atomicAdd(a.du + (size_t)tgt * CH + o, tb[r * TT + o]);
            }
            accw[2] = outer(tb + 0 * 16 * TT, tb + 1 * 16 * TT, p, g, accw[2]);
            accw[3] = outer(tb + 0 * 16 * TT, tb + 2 * 16 * TT, p, g, accw[3]);
        }
        f32x4 d_pe0 = mm(wl, 10, lane, dq, zero4), d_pe1 = mm(wl, 11, lane, dq, zero4);
        if (f.ukey) {
            d_pe0 -= kf0 * mm(wl, 10, lane, tot4, zero4);
            d_pe1 -= kf1 * mm(wl, 11, lane, tot4, zero4);
        }
        const f32x4 g_pe0 = mask_pos(d_pe0, y_pe0);
        const f32x4 g_pe1 = mask_pos(d_pe1, y_pe1);
        s1[0] += g_pe0; s2[0] += g_pe0 * ac_pe0;
        s1[1] += g_pe1; s2[1] += g_pe1 * ac_pe1;
        {
            // moments for the dW of mlp_conv: sum g (x) VI (two tiles) and sum VI' (x) VI' with VI' = (VI, 1)
            f32x4 x1 = x;
            if (g == 3) x1[0] = 1.f;                // channel 12 is always padding (cv <= 12)
            put_tile(tb + 0 * 16 * TT, g_pe0, p, g);
            put_tile(tb + 1 * 16 * TT, g_pe1, p, g);
            put_tile(tb + 2 * 16 * TT, x1, p, g);
            accw[0] = outer(tb + 0 * 16 * TT, tb + 2 * 16 * TT, p, g, accw[0]);
            accw[1] = outer(tb + 1 * 16 * TT, tb + 2 * 16 * TT, p, g, accw[1]);
            accw[4] = outer(tb + 2 * 16 * TT, tb + 2 * 16 * TT, p, g, accw[4]);
        }
    }
    if (LEVEL == 3) {
        // tiles: 0, 1 sum g_pe (x) VI' | 3, 4 dW of g1 | 6 sum VI' (x) VI'
        const int tiles[5] = {0, 1, 3, 4, 6};
        for (int wv = 0; wv < NWAVE; ++wv) {        // wave order: deterministic
            if (wave == wv) {
#pragma unroll
                for (int i = 0; i < 5; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[tiles[i] * 256 + (4 * g + r) * 16 + p] += accw[i][r];
            }
            __syncthreads();
        }
    }
    {
        if (LEVEL == 2) {                           // dW tile of g2 -> red_top[0]
            for (int wv = 0; wv < NWAVE; ++wv) {
                if (wave == wv) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) red_top[(4 * g + r) * 16 + p] += accw[4][r];
                }
                __syncthreads();
            }
        }
        // L1: g2 -> group 0.  L2: g1 -> group 0.  L3: pe lo, pe hi -> groups 0, 1.
        float (*rw)[3][2][16] = reinterpret_cast<float (*)[3][2][16]>(red_s);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v1 = s1[q][r], v2 = s2[q][r];
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) { v1 += __shfl_xor(v1, off, WAVE); v2 += __shfl_xor(v2, off, WAVE); }
                if (p == 0) { rw[wave][q][0][4 * g + r] = v1; rw[wave][q][1][4 * g + r] = v2; }
            }
    }
}

// WeightNet branch: VI -> w1 (8) -> w2 (8) -> w3 (C_mid), ReLU each.
//   sums: groups {L1: w3 -> 1 | L2: w2 -> 1 | L3: w1 -> 2};   L2: dW tile of w3;   L3: tiles 2, 5.
template <int LEVEL>
__device__ __forceinline__ void weightnet_branch(const ChainBwdArgs& a, const float (*cf)[NCONST][16], const float4* wl,
                                                 float* red, float* red_s, float* red_top, float* tb, long long t0,
                                                 long long tstride) {
    const ChainArgs& f = a.f;
    const int lane = lane_id(), wave = wave_id(), p = lane & 15, g = lane >> 4;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const long long ntiles = f.E / 16;
    const TileIO io{a, p, g, ntiles};
    f32x4 s1 = zero4, s2 = zero4;
    f32x4 accw[3] = {zero4, zero4, zero4};                        // w1, w2, w3
    for (long long t = t0; t < ntiles; t += tstride) {
        asm volatile("" ::: "memory");
        const f32x4 ac_a2 = io.load_grad(a.a2_acc, t, CH);
        if (LEVEL <= 2) {
            // the top layer: w3 from the stored accumulator of w2
            const f32x4 dwv = io.load_grad(a.dw, t, f.cm);
            const f32x4 y_a2 = relu4(pre_of(ac_a2, cf[S_W2], g));
            const f32x4 ac_w = mm(wl, 7, lane, y_a2, zero4);
            const f32x4 g_w = mask_pos(dwv, pre_of(ac_w, cf[S_W3], g));
            if (LEVEL == 1) {
                s1 += g_w; s2 += g_w * ac_w;
            } else {
                const f32x4 dz_w = bn_dz(g_w, ac_w, cf[S_W3], g);
                put_tile(tb + 0 * 16 * TT, dz_w, p, g);
                put_tile(tb + 1 * 16 * TT, y_a2, p, g);
                accw[2] = outer(tb + 0 * 16 * TT, tb + 1 * 16 * TT, p, g, accw[2]);
                const f32x4 g_a2 = mask_pos(mm(wl, 9, lane, dz_w, zero4), y_a2);
                if (g < 2) st4(a.ga2 + (size_t)(t * 16 + p) * CH + 4 * g, make_float4(g_a2[0], g_a2[1], g_a2[2], g_a2[3]));
                s1 += g_a2; s2 += g_a2 * ac_a2;
            }
            continue;
        }
        // pass 3: from the stored g of w2 down through w1
        const f32x4 g_a2 = io.load_grad(a.ga2, t, CH);
        const f32x4 x = io.load_x(t);
        const f32x4 ac_a1 = mm(wl, 2, lane, x, zero4);
        const f32x4 y_a1 = relu4(pre_of(ac_a1, cf[S_W1], g));
        const f32x4 dz_a2 = bn_dz(g_a2, ac_a2, cf[S_W2], g);
        {
            put_tile(tb + 0 * 16 * TT, dz_a2, p, g);
            put_tile(tb + 1 * 16 * TT, y_a1, p, g);
            accw[1] = outer(tb + 0 * 16 * TT, tb + 1 * 16 * TT, p, g, accw[1]);
        }
        const f32x4 g_a1 = mask_pos(mm(wl, 12, lane, dz_a2, zero4), y_a1);
        s1 += g_a1; s2 += g_a1 * ac_a1;
        {   // moment for the dW of w1: sum g (x) VI' (the guidance workgroups accumulate sum VI' (x) VI')
            f32x4 x1 = x;
            if (g == 3) x1[0] = 1.f;
            put_tile(tb + 0 * 16 * TT, g_a1, p, g);
            put_tile(tb + 1 * 16 * TT, x1, p, g);
            accw[0] = outer(tb + 0 * 16 * TT, tb + 1 * 16 * TT, p, g, accw[0]);
            if (a.wn_only) accw[2] = outer(tb + 1 * 16 * TT, tb + 1 * 16 * TT, p, g, accw[2]);     // no guidance workgroups
        }
    }
    if (LEVEL == 3) {
        const int tiles[3] = {2, 5, 6};         // 2: sum g_a1 (x) VI' | 5: dW of w2 | 6: sum VI' (x) VI' (wn_only)
        for (int wv = 0; wv < NWAVE; ++wv) {
            if (wave == wv) {
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (i < 2 || a.wn_only) red[tiles[i] * 256 + (4 * g + r) * 16 + p] += accw[i][r];
            }
            __syncthreads();
        }
    }
    {
        if (LEVEL == 2) {                           // dW tile of w3 -> red_top[1]
            for (int wv = 0; wv < NWAVE; ++wv) {
                if (wave == wv) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) red_top[256 + (4 * g + r) * 16 + p] += accw[2][r];
                }
                __syncthreads();
            }
        }
        float (*rw)[3][2][16] = reinterpret_cast<float (*)[3][2][16]>(red_s);
        const int q = LEVEL == 3 ? 2 : 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v1 = s1[r], v2 = s2[r];
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) { v1 += __shfl_xor(v1, off, WAVE); v2 += __shfl_xor(v2, off, WAVE); }
            if (p == 0) { rw[wave][q][0][4 * g + r] = v1; rw[wave][q][1][4 * g + r] = v2; }
        }
    }
}

// ---- pass 3 without LDS transposes ---------------------------------------------------------------------------
// The outer products dW += sum_e dz[o][e] * in[c][e] contract over EDGES, so both operands are wanted with the channel on
// the low lane bits and the edge on (lane >> 4, register) -- "layout T" -- while the layer products keep the edge on
// the low lane bits and the channel on (lane >> 4, register) -- "layout E".  Transposing six 16x16 tiles per 16 edges
// through LDS (one b128 store + eight b32 reads per operand) made this pass LDS-bound (87 us of LDS against 54 us of
// matrix work at 1.28 M edges).  The matrix core transposes for free: with the activation as the A operand and
// the weight fragment as the B operand, D[i = edge][j = channel] = sum_c Y[edge][c] W[channel][c] lands in layout T
// (the SAME fragment registers serve both roles, since lane (p, g) step s holds W[p][4g + s] either way).  So
//   * everything that only feeds outer products or element-wise masks is PRODUCED in layout T (mlp_conv / w1 recompute,
//     the masked gradients g_pe / g_a1),
//   * what is read from memory is read in both layouts (the second read hits the cache lines of the first),
//   * the few tensors needed both ways (dq, dz of w2) are computed twice from those loads (8 fused multiply-adds),
//   * per-channel BatchNorm constants of layout T are one scalar per lane, kept in registers.
// Neighbourhood sums of the key subtraction in layout T run over registers and over lane bits 4-5: v_permlane16_swap /
// v_permlane32_swap (gfx950) of a register with itself give the two butterfly partners without touching LDS.
__device__ __forceinline__ float xor16_sum(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// lane i <- lane i + 8 of its row of 16 (zero beyond the row)
__device__ __forceinline__ float row_shl8(float v) {
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x108, 0xf, 0xf, true));
}
// sum over the aligned group of K (power of two <= 16) consecutive lanes of a row, in every lane of the group: DPP only
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float group_sum_dpp(float v, int K) {
    if (K >= 2) v += dpp_mov<0xB1>(v);          // quad_perm [1,0,3,2]
    if (K >= 4) v += dpp_mov<0x4E>(v);          // quad_perm [2,3,0,1]
    if (K >= 8) v += dpp_mov<0x141>(v);         // row_half_mirror: the other quad of the half row
    if (K >= 16) v += dpp_mov<0x140>(v);        // row_mirror: the other half row
    return v;
}
// activation (layout E) as the A operand: result in layout T
__device__ __forceinline__ f32x4 mm_t(const float4* wl, int frag, int lane, f32x4 operand, f32x4 acc) {
    const f32x4 w = to_v4(wl[frag * WAVE + lane]);
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = PCF_MFMA(operand[s], w[s], acc);
    return acc;
}
__device__ __forceinline__ f32x4 outer_t(f32x4 dz, f32x4 in, f32x4 acc) {
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = PCF_MFMA(dz[s], in[s], acc);
    return acc;
}
// two [E, 8] tensors of one tile in layout T with four loads: lanes p < 8 read channel p of `lo`, lanes p >= 8 channel
// p - 8 of `hi`; a row shift brings the second to the lanes of the first.  Valid in lanes p < 8, zero elsewhere.
__device__ __forceinline__ void load_pair_t(const float* lo, const float* hi, long long t, int p, int g, f32x4& vlo, f32x4& vhi) {
    const float* src = (p < 8 ? lo : hi) + (size_t)(t * 16 + 4 * g) * CH + (p & 7);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float v = src[r * CH];
        vhi[r] = row_shl8(v);
        vlo[r] = p < 8 ? v : 0.f;
    }
}
// VI' = (VI, 1) of one tile in layout T (channel 12 is always padding, cv <= 12)
__device__ __forceinline__ f32x4 load_x1_t(const ChainArgs& f, long long t, int p, int g) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (p < f.cv) {
        const float* q = f.vi + (size_t)(t * 16 + 4 * g) * f.cv + p;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = q[r * f.cv];
    } else if (p == 12) {
        v = f32x4{1.f, 1.f, 1.f, 1.f};
    }
    return v;
}
__device__ __forceinline__ f32x4 relu_bn_t(f32x4 acc, float sc, float sh) {
    return f32x4{fmaxf(acc[0] * sc + sh, 0.f), fmaxf(acc[1] * sc + sh, 0.f), fmaxf(acc[2] * sc + sh, 0.f), fmaxf(acc[3] * sc + sh, 0.f)};
}
__device__ __forceinline__ float sum4(f32x4 v) { return (v[0] + v[1]) + (v[2] + v[3]); }
// per-channel sums of layout T live one per lane (channel p): fold lane bits 4-5, lane g == 0 writes
__device__ __forceinline__ void put_sums_t(float (*rw)[3][2][16], int wave, int q, int p, int g, float v1, float v2) {
    v1 = xor32_sum(xor16_sum(v1));
    v2 = xor32_sum(xor16_sum(v2));
    if (g == 0) { rw[wave][q][0][p] = v1; rw[wave][q][1][p] = v2; }
}

__device__ __forceinline__ void guidance_branch3_t(const ChainBwdArgs& a, const float (*cf)[NCONST][16], const float4* wl,
                                                   float* red, float* red_s, long long t0, long long tstride) {
    const ChainArgs& f = a.f;
    const int lane = lane_id(), wave = wave_id(), p = lane & 15, g = lane >> 4;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const long long ntiles = f.E / 16;
    const TileIO io{a, p, g, ntiles};
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
    f32x4 accw[5] = {zero4, zero4, zero4, zero4, zero4};          // pe lo, pe hi, g1 (in lo), g1 (in hi), VI' (x) VI'
    const bool first = (p & (f.K - 1)) == 0;
    // layout T constants: channel p of mlp_conv lo / hi, channel p of g1
    const float sc_pe0 = cf[S_PE0][K_SC][p], sh_pe0 = cf[S_PE0][K_SH][p], sc_pe1 = cf[S_PE1][K_SC][p], sh_pe1 = cf[S_PE1][K_SH][p];
    const float sc_g1 = cf[S_G1][K_SC][p], d0_g1 = cf[S_G1][K_D0][p], d1_g1 = cf[S_G1][K_D1][p];
    BatchWalk walk;
    walk.init(f.rows_per_batch);
    for (long long t = t0; t < ntiles; t += tstride) {
        asm volatile("" ::: "memory");          // LDS-resident weights / constants are re-read per tile, not hoisted into VGPRs
        // loads in the order of their use: the VI row first (the mlp_conv recompute runs while the others land)
        const f32x4 x = io.load_x(t);
        const f32x4 x1t = load_x1_t(f, t, p, g);
        const f32x4 ac_h1 = io.load_grad(a.h1_acc, t, CH);
        const f32x4 g_h1 = io.load_grad(a.gh1, t, CH);
        f32x4 g_h1t, ac_h1t;
        load_pair_t(a.gh1, a.h1_acc, t, p, g, g_h1t, ac_h1t);
        // du rows of the four edges 4g..4g+3 this lane holds in layout T
        int tgt[4];
        {
            (void)walk.batch_of(t * 16, 0);              // advances the walk to this tile (wave-uniform)
            const int e_rel = (int)(walk.next_start - t * 16);                          // first edge of the next batch, tile-relative
            const longlong2* ib = reinterpret_cast<const longlong2*>(f.idx + t * 16 + 4 * g);
            const longlong2 j01 = ib[0], j23 = ib[1];
            const long long jj[4] = {j01.x, j01.y, j23.x, j23.y};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int batch = walk.b + (4 * g + r >= e_rel ? 1 : 0);
                tgt[r] = (jj[r] >= 0 && jj[r] < f.N) ? batch * f.N + (int)jj[r] : -1;
            }
        }
        // mlp_conv recomputed straight into layout T
        const f32x4 ac_pe0 = mm_t(wl, 0, lane, x, zero4);
        const f32x4 ac_pe1 = mm_t(wl, 1, lane, x, zero4);
        const f32x4 y_pe0 = relu_bn_t(ac_pe0, sc_pe0, sh_pe0);
        const f32x4 y_pe1 = relu_bn_t(ac_pe1, sc_pe1, sh_pe1);
        // dz of g1 in both layouts; z[k] = q[k] - q[key] + b  =>  dq[k] = dz[k] - [k is the key] * (sum over the neighbourhood)
        f32x4 dq = bn_dz(g_h1, ac_h1, cf[S_G1], g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float tot = group_sum_dpp(dq[r], f.K);
            if (first) dq[r] -= tot;
        }
        f32x4 dqt = g_h1t * sc_g1 + (ac_h1t * d1_g1 + d0_g1);
        if (p >= CH) dqt = zero4;
        if (f.K >= 4) {
            float tot = sum4(dqt);
            if (f.K >= 8) tot = xor16_sum(tot);
            if (f.K >= 16) tot = xor32_sum(tot);
            if (((4 * g) & (f.K - 1)) == 0) dqt[0] -= tot;
        } else if (f.K == 2) {
            const float t01 = dqt[0] + dqt[1], t23 = dqt[2] + dqt[3];
            dqt[0] -= t01; dqt[2] -= t23;
        } else {
            dqt = zero4;                                 // K = 1: every edge is its own key
        }
        // gradient of the gathered term: lanes p = 0..7 cover the 8 consecutive channels of one row of du
        if (p < CH) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
            {
#ifndef PCF_EXPERIMENT_NO_DU
                if (tgt[r] >= 0) atomicAdd(a.du + (size_t)(unsigned)tgt[r] * CH + p, dqt[r]);
#endif
            }
        }
        accw[2] = outer_t(dqt, y_pe0, accw[2]);
        accw[3] = outer_t(dqt, y_pe1, accw[3]);
        const f32x4 g_pe0 = mask_pos(mm_t(wl, 10, lane, dq, zero4), y_pe0);
        const f32x4 g_pe1 = mask_pos(mm_t(wl, 11, lane, dq, zero4), y_pe1);
        s1[0] += sum4(g_pe0); s2[0] += sum4(g_pe0 * ac_pe0);
        s1[1] += sum4(g_pe1); s2[1] += sum4(g_pe1 * ac_pe1);
        // moments for the dW of mlp_conv: sum g (x) VI' (two tiles) and sum VI' (x) VI'
        accw[0] = outer_t(g_pe0, x1t, accw[0]);
        accw[1] = outer_t(g_pe1, x1t, accw[1]);
        accw[4] = outer_t(x1t, x1t, accw[4]);
    }
    // tiles: 0, 1 sum g_pe (x) VI' | 3, 4 dW of g1 | 6 sum VI' (x) VI'
    const int tiles[5] = {0, 1, 3, 4, 6};
    for (int wv = 0; wv < NWAVE; ++wv) {        // wave order: deterministic
        if (wave == wv) {
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[tiles[i] * 256 + (4 * g + r) * 16 + p] += accw[i][r];
        }
        __syncthreads();
    }
    float (*rw)[3][2][16] = reinterpret_cast<float (*)[3][2][16]>(red_s);
    put_sums_t(rw, wave, 0, p, g, s1[0], s2[0]);
    put_sums_t(rw, wave, 1, p, g, s1[1], s2[1]);
}

__device__ __forceinline__ void weightnet_branch3_t(const ChainBwdArgs& a, const float (*cf)[NCONST][16], const float4* wl,
                                                    float* red, float* red_s, long long t0, long long tstride) {
    const ChainArgs& f = a.f;
    const int lane = lane_id(), wave = wave_id(), p = lane & 15, g = lane >> 4;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const long long ntiles = f.E / 16;
    const TileIO io{a, p, g, ntiles};
    float s1 = 0.f, s2 = 0.f;
    f32x4 accw[3] = {zero4, zero4, zero4};                        // w1 moment, dW of w2, VI' (x) VI' (wn_only)
    const float sc_w1 = cf[S_W1][K_SC][p], sh_w1 = cf[S_W1][K_SH][p];
    const float sc_w2 = cf[S_W2][K_SC][p], d0_w2 = cf[S_W2][K_D0][p], d1_w2 = cf[S_W2][K_D1][p];
    for (long long t = t0; t < ntiles; t += tstride) {
        asm volatile("" ::: "memory");
        const f32x4 ac_a2 = io.load_grad(a.a2_acc, t, CH);
        const f32x4 g_a2 = io.load_grad(a.ga2, t, CH);
        const f32x4 x = io.load_x(t);
        f32x4 g_a2t, ac_a2t;
        load_pair_t(a.ga2, a.a2_acc, t, p, g, g_a2t, ac_a2t);
        const f32x4 x1t = load_x1_t(f, t, p, g);
        const f32x4 ac_a1 = mm_t(wl, 2, lane, x, zero4);
        const f32x4 y_a1 = relu_bn_t(ac_a1, sc_w1, sh_w1);
        const f32x4 dz_a2 = bn_dz(g_a2, ac_a2, cf[S_W2], g);
        f32x4 dz_a2t = g_a2t * sc_w2 + (ac_a2t * d1_w2 + d0_w2);
        if (p >= CH) dz_a2t = zero4;
        accw[1] = outer_t(dz_a2t, y_a1, accw[1]);
        const f32x4 g_a1 = mask_pos(mm_t(wl, 12, lane, dz_a2, zero4), y_a1);
        s1 += sum4(g_a1); s2 += sum4(g_a1 * ac_a1);
        accw[0] = outer_t(g_a1, x1t, accw[0]);
        if (a.wn_only) accw[2] = outer_t(x1t, x1t, accw[2]);     // no guidance workgroups
    }
    const int tiles[3] = {2, 5, 6};         // 2: sum g_a1 (x) VI' | 5: dW of w2 | 6: sum VI' (x) VI' (wn_only)
    for (int wv = 0; wv < NWAVE; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (i < 2 || a.wn_only) red[tiles[i] * 256 + (4 * g + r) * 16 + p] += accw[i][r];
        }
        __syncthreads();
    }
    float (*rw)[3][2][16] = reinterpret_cast<float (*)[3][2][16]>(red_s);
    put_sums_t(rw, wave, 2, p, g, s1, s2);
}

// workgroup b of the launch: branch and rank among the workgroups of its branch; NG of every 8 workgroups take
// the guidance branch (SplitOf, chosen by measurement)
#ifndef PCF_NG3T
#define PCF_NG3T 5      // guidance share (of 8 workgroups) of the register-layout pass 3, by measurement
#endif
template <int NG> __device__ __host__ inline bool is_guidance_block(int b) { return (b & 7) < NG; }
template <int NG> __device__ __host__ inline int branch_rank(int b) {
    return is_guidance_block<NG>(b) ? (b >> 3) * NG + (b & 7) : (b >> 3) * (8 - NG) + (b & 7) - NG;
}
template <int NG> __host__ inline int branch_blocks(int grid, bool guidance) {       // grid is a multiple of 8
    return guidance ? (grid >> 3) * NG : (grid >> 3) * (8 - NG);
}
// matrix instructions per tile, guidance : WeightNet = 4 : 4 (pass 1), 12 : 12 (pass 2), 36 : 16 (pass 3).
// Measured: passes 1-2 do not care between 4 and 6 of 8 (they are closer to the memory side); pass 3 took 196 us
// with 6 of 8, 229 us with 5, 358 us with 7 (before the 128-VGPR bound, which brings 6 of 8 to 147 us).
template <int LEVEL, bool REGT = false> struct SplitOf { static constexpr int NG = LEVEL == 3 ? (REGT ? PCF_NG3T : 6) : 5; };

// at least 4 waves per SIMD (<= 128 VGPRs): the last pass lands on 132 without the bound and runs 196 us instead of 147
// REGT (pass 3 only): operands of the outer products produced in registers in the transposed layout (above) instead of
// being turned through LDS tiles; the LDS form stays as the cross-check (pcf_hip_set_chain_backward_engine).
template <int LEVEL, bool REGT = false>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(4, 8))) void pcf_chain_bwd_kernel(const ChainBwdArgs a, int grid8) {
    __shared__ __align__(16) float cf[NSLOT][NCONST][16];
    __shared__ float4 wl[NFRAG * WAVE];
    __shared__ float red[LEVEL == 3 ? NDW * 256 : 1];          // dW / moment tiles of the last pass
    __shared__ float red_s[NWAVE * 96];                         // per-channel sums of the pass
    __shared__ float red_top[LEVEL == 2 ? 2 * 256 : 1];
    __shared__ __align__(16) float tbuf[REGT ? 4 : LEVEL == 3 ? NWAVE * 3 * 16 * TT : (LEVEL == 2 ? NWAVE * 2 * 16 * TT : 4)];
    __shared__ int gi[NWAVE][16];
    stage_consts(a, cf, LEVEL);
    stage_weights(a, wl);
    for (int t = threadIdx.x; t < NWAVE * 96; t += BLOCK) red_s[t] = 0.f;
    if (LEVEL == 3)
        for (int t = threadIdx.x; t < NDW * 256; t += BLOCK) red[t] = 0.f;
    if (LEVEL == 2)
        for (int t = threadIdx.x; t < 2 * 256; t += BLOCK) red_top[t] = 0.f;
    __syncthreads();
    const int wave = wave_id();
    float* tb = LEVEL == 3 ? tbuf + wave * 3 * 16 * TT : (LEVEL == 2 ? tbuf + wave * 2 * 16 * TT : tbuf);
    constexpr int NG = SplitOf<LEVEL, REGT>::NG;
    const int rank = branch_rank<NG>(blockIdx.x);
    if (REGT && LEVEL == 3) {
        if (a.wn_only) weightnet_branch3_t(a, cf, wl, red, red_s, (long long)blockIdx.x * NWAVE + wave, (long long)gridDim.x * NWAVE);
        else if (is_guidance_block<NG>(blockIdx.x))
            guidance_branch3_t(a, cf, wl, red, red_s, (long long)rank * NWAVE + wave, (long long)grid8 * NG * NWAVE);
        else weightnet_branch3_t(a, cf, wl, red, red_s, (long long)rank * NWAVE + wave, (long long)grid8 * (8 - NG) * NWAVE);
    } else if (a.wn_only)
        weightnet_branch<LEVEL>(a, cf, wl, red, red_s, red_top, tb, (long long)blockIdx.x * NWAVE + wave,
                                (long long)gridDim.x * NWAVE);
    else if (is_guidance_block<NG>(blockIdx.x))
        guidance_branch<LEVEL>(a, cf, wl, red, red_s, red_top, tb, gi[wave], (long long)rank * NWAVE + wave,
                               (long long)grid8 * NG * NWAVE);
    else
        weightnet_branch<LEVEL>(a, cf, wl, red, red_s, red_top, tb, (long long)rank * NWAVE + wave,
                                (long long)grid8 * (8 - NG) * NWAVE);
    __syncthreads();
    if (LEVEL == 3) {
        float* outp = a.part + (size_t)blockIdx.x * (NDW * 256);
        for (int u = threadIdx.x; u < NDW * 256; u += BLOCK) outp[u] = red[u];
    }
    if (LEVEL == 2) {
        float* outp = a.part_top + (size_t)blockIdx.x * 512;
        for (int u = threadIdx.x; u < 512; u += BLOCK) outp[u] = red_top[u];
    }
    if (threadIdx.x < 96) {
        const float (*rw)[3][2][16] = reinterpret_cast<const float (*)[3][2][16]>(red_s);
        const int q = threadIdx.x / 32, which = (threadIdx.x >> 4) & 1, c = threadIdx.x & 15;
        float tsum = 0.f;
#pragma unroll
        for (int w = 0; w < NWAVE; ++w) tsum += rw[w][q][which][c];
        a.part_sums[(size_t)blockIdx.x * 96 + threadIdx.x] = tsum;
    }
}

// Sums of one pass: up to three groups of 16 channels, each a slice [chan0, chan0 + count) of one layer.
struct BwdFinGroup { float* dbeta; float* dgamma; float* gmean; float* gxmean; const float* mean; const float* rstd; const float* bias; int chan0; int count; };
struct BwdFinArgs { BwdFinGroup g[3]; const float* part; int nblocks; long long R; };

__global__ __launch_bounds__(1024) void chain_bwd_finalize_kernel(const BwdFinArgs f) {
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
        const int q = gq, c = quarter * 4 + threadIdx.x;
        double sums[2] = {0.0, 0.0};
#pragma unroll
        for (int which = 0; which < 2; ++which)
            for (int sl = 0; sl < 128; ++sl) sums[which] += sh[sl][which * 4 + threadIdx.x];
        const BwdFinGroup& g = f.g[q];
        if (g.dbeta && c < g.count) {
            // sums[1] = sum g * acc with acc the raw accumulator; xhat = acc * rstd + (b - mean) * rstd
            const int o = g.chan0 + c;
            const double rs = (double)g.rstd[o], x0 = ((double)g.bias[o] - (double)g.mean[o]) * rs;
            const double dgamma = rs * sums[1] + x0 * sums[0];
            g.dbeta[o] = (float)sums[0];
            g.dgamma[o] = (float)dgamma;
            g.gmean[o] = (float)(sums[0] / (double)f.R);
            g.gxmean[o] = (float)(dgamma / (double)f.R);
        }
    }
}

// After pass 3: (1) fixed-order reduction of the per-workgroup tiles and sums into a small buffer,
//   red[0 .. 7*256)    tiles of pass 3: 0, 1 sum g_pe (x) VI' | 2 sum g_a1 (x) VI' | 3, 4 dW g1 | 5 dW w2 | 6 sum VI' (x) VI'
//   red[7*256 .. 9*256) tiles of pass 2: dW g2, dW w3
//   red[9*256 .. +96)   sums of pass 3: (sum g, sum g.acc) of mlp_conv lo / hi and w1
// (2) one workgroup turns them into the six dW, the dgamma / dbeta of mlp_conv and w1 and the zero db.
constexpr int RED_TILES = 9 * 256;
constexpr int RED_TOTAL = RED_TILES + 96;

struct BwdReduceArgs { const float* part; const float* part_top; const float* part_sums; int nblocks; float* red; };

__global__ __launch_bounds__(1024) void chain_bwd_reduce_kernel(const BwdReduceArgs f) {
    __shared__ float sh[16][64];
    const int e = blockIdx.x * 64 + (threadIdx.x & 63);
    const int slice = threadIdx.x >> 6;
    const float* src = nullptr;
    size_t stride = 0;
    if (e < 7 * 256) { src = f.part + e; stride = NDW * 256; }
    else if (e < RED_TILES) { src = f.part_top + (e - 7 * 256); stride = 512; }
    else if (e < RED_TOTAL) { src = f.part_sums + (e - RED_TILES); stride = 96; }
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (src) {
        int p = slice;
        for (; p + 48 < f.nblocks; p += 64) {
            a0 += src[(size_t)p * stride];
            a1 += src[(size_t)(p + 16) * stride];
            a2 += src[(size_t)(p + 32) * stride];
            a3 += src[(size_t)(p + 48) * stride];
        }
        for (; p < f.nblocks; p += 16) a0 += src[(size_t)p * stride];
    }
    sh[slice][threadIdx.x & 63] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (threadIdx.x >= 64 || !src) return;
    float tot = 0.f;
#pragma unroll
    for (int sl = 0; sl < 16; ++sl) tot += sh[sl][threadIdx.x];
    f.red[e] = tot;
}

struct BwdCombineArgs {
    const float* red;
    const float* W_pe; const float* W_w1;            // [g, cv], [8, cv]
    const float* stats;                              // forward statistics: mean at l*64, rstd at (6+l)*64
    const float* b_pe; const float* b_w1; const float* gamma_pe; const float* gamma_w1;
    float* dW[6]; float* db[6];
    float* dgamma_pe; float* dbeta_pe; float* dgamma_w1; float* dbeta_w1;
    long long R;
    int cv, g, heads, cm;
};

__global__ __launch_bounds__(BLOCK) void chain_bwd_combine_kernel(const BwdCombineArgs f) {
    __shared__ float sc[48], d0[48], d1[48];         // mlp_conv channels 0..31, w1 channels 32..39
    const float* red = f.red;
    const int tid = threadIdx.x;
    if (tid < 40) {
        const bool pe = tid < 32;
        const int o = pe ? tid : tid - 32;
        const int layer = pe ? L_PE : L_W1;
        const bool valid = pe ? o < f.g : true;
        float vsc = 0.f, vd0 = 0.f, vd1 = 0.f;
        if (valid) {
            // sums live in groups: mlp_conv lo -> group 0, hi -> group 1, w1 -> group 2; [group][sum | sum g.acc][16]
            const int grp = pe ? (o >> 4) : 2, c = pe ? (o & 15) : o;
            const double S1 = (double)red[RED_TILES + grp * 32 + c], S2 = (double)red[RED_TILES + grp * 32 + 16 + c];
            const double rs = (double)f.stats[(6 + layer) * 64 + o];
            const double bias = (double)(pe ? f.b_pe[o] : f.b_w1[o]);
            const double x0 = (bias - (double)f.stats[layer * 64 + o]) * rs;
            const double gamma = (double)(pe ? f.gamma_pe[o] : f.gamma_w1[o]);
            const double dgamma = rs * S2 + x0 * S1;                       // sum g * xhat
            (pe ? f.dbeta_pe : f.dbeta_w1)[o] = (float)S1;
            (pe ? f.dgamma_pe : f.dgamma_w1)[o] = (float)dgamma;
            const double gm = S1 / (double)f.R, gxm = dgamma / (double)f.R;
            const double s = rs * gamma;
            vsc = (float)s;
            vd1 = (float)(-s * gxm * rs);
            vd0 = (float)(-s * (gm + gxm * x0));
        }
        sc[tid] = vsc; d0[tid] = vd0; d1[tid] = vd1;
    }
    // zero bias gradients (bias in front of a training-mode BatchNorm)
    if (tid >= 64 && tid < 64 + 6 * 32) {
        const int l = (tid - 64) >> 5, o = (tid - 64) & 31;
        const int couts[6] = {f.g, CH, f.heads, CH, CH, f.cm};
        if (o < couts[l] && f.db[l]) f.db[l][o] = 0.f;
    }
    __syncthreads();
    const float* XX = red + 6 * 256;                 // [c'][c] = sum VI'[c'] VI'[c], VI'[12] = 1
    // dW of mlp_conv and w1: SC * M1 + D1 * (W . XX) + D0 * sum VI
    for (int e = tid; e < 40 * 16; e += BLOCK) {
        const int row = e >> 4, c = e & 15;           // row: 0..31 mlp_conv, 32..39 w1
        const bool pe = row < 32;
        const int o = pe ? row : row - 32;
        if (c >= f.cv || (pe && o >= f.g)) continue;
        const float* W = pe ? f.W_pe : f.W_w1;
        const float M1 = pe ? red[(o >> 4) * 256 + (o & 15) * 16 + c] : red[2 * 256 + o * 16 + c];
        float M2 = 0.f;
        for (int k = 0; k < f.cv; ++k) M2 += W[o * f.cv + k] * XX[k * 16 + c];
        const float M3 = XX[12 * 16 + c];
        const float v = sc[row] * M1 + d1[row] * M2 + d0[row] * M3;
        (pe ? f.dW[L_PE] : f.dW[L_W1])[o * f.cv + c] = v;
    }
    // the four upper layers: tiles are the gradients themselves
    for (int e = tid; e < 5 * 256; e += BLOCK) {
        const int t = e >> 8, o = (e >> 4) & 15, c = e & 15;
        const float v = red[(t < 3 ? 3 + t : 4 + t) * 256 + o * 16 + c];      // t: 0 g1 lo, 1 g1 hi, 2 w2, 3 g2, 4 w3
        switch (t) {
            case 0: if (o < CH && c < f.g) f.dW[L_G1][o * f.g + c] = v; break;
            case 1: if (o < CH && 16 + c < f.g) f.dW[L_G1][o * f.g + 16 + c] = v; break;
            case 2: if (o < CH && c < CH) f.dW[L_W2][o * CH + c] = v; break;
            case 3: if (o < f.heads && c < CH) f.dW[L_G2][o * CH + c] = v; break;
            default: if (o < f.cm && c < CH) f.dW[L_W3][o * CH + c] = v; break;
        }
    }
}

}  // namespace pcf

// Pass 3 of the fused backward: 1 = outer-product operands turned through LDS tiles (default), 0 = produced in registers
// in the transposed lane layout.  Measured at 1.28 M edges (profiles/r02_pass3_engines.txt): the register form takes the
// LDS pipe from 87 to 9 us per launch -- below the 54 us of matrix work -- but its 19 vector-memory instructions per tile
// (the second, strided read of every operand) against 8 leave the four resident waves waiting on memory: 152 us against
// 145 us per launch.  It stays selectable (PCF_CHAIN_BWD_LDS=0 or pcf_hip_set_chain_backward_engine(0)) and is held to the
// LDS form by the tests.
static std::atomic<int> g_chain_bwd_lds{-1};
static bool chain_bwd_lds_transposes() {
    int v = g_chain_bwd_lds.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("PCF_CHAIN_BWD_LDS");
        v = (e && e[0] == '0') ? 0 : 1;
        g_chain_bwd_lds.store(v, std::memory_order_relaxed);
    }
    return v == 1;
}

static int chain_backward_impl(const float* ukey, float* dukey, bool wn_only, const float* vi, const int64_t* idx, const float* h1_acc, const float* a2_acc,
                               const float* dscore, const float* dw, long long E, long long rows_per_batch, int N, int K, int cv, int g, int heads, int cm,
                               const float* const* W, const float* const* b, const float* const* gamma,
                               const float* const* beta, const float* stats, float* du, float* const* dW, float* const* db,
                               float* const* dgamma, float* const* dbeta, void* workspace, size_t workspace_bytes,
                               void* stream) {
    using namespace pcf;
    PCF_REQUIRE(E >= 0 && rows_per_batch > 0 && N >= 0, "pcf_chain_backward: bad sizes");
    if (ukey && (K < 2 || !dukey || !aligned16(dukey)))
        return fail(PCF_E_UNSUPPORTED, "pcf_chain_backward (maximum key): K >= 2 and dukey a 16-byte aligned [E / K, 8] buffer");
    if (cv < 1 || cv > CV || cm < 1 || cm > CMX || (!wn_only && (g < 1 || g > CG || heads < 1 || heads > CHD)))
        return fail(PCF_E_UNSUPPORTED, "pcf_chain_backward: widths outside the fused kernel (cv=%d<=12, g=%d<=32, heads=%d<=8, cm=%d<=16)", cv, g, heads, cm);
    if (E % 16 != 0 || (!wn_only && (K < 1 || K > 16 || (K & (K - 1)) != 0 || E % rows_per_batch != 0 || rows_per_batch < 16)))
        return fail(PCF_E_UNSUPPORTED, "pcf_chain_backward: K must be a power of two <= 16, the edge count a multiple of 16 and >= 16 edges per batch (K=%d)", K);
    PCF_REQUIRE(W && b && gamma && beta && stats && (du || wn_only) && dW && db && dgamma && dbeta, "pcf_chain_backward: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int couts[6] = {g, CH, heads, CH, CH, cm};
    const int cins[6] = {cv, g, CH, cv, CH, CH};
    for (int l = 0; l < 6; ++l) {
        if (wn_only && l < 3) continue;                   // no guidance layers
        PCF_REQUIRE(W[l] && b[l] && gamma[l] && beta[l] && dW[l] && db[l] && dgamma[l] && dbeta[l],
                    "pcf_chain_backward: null parameter of layer %d", l);
        if (E == 0) {
            (void)zero_async(db[l], (size_t)couts[l] * 4, s);
            (void)zero_async(dW[l], (size_t)couts[l] * cins[l] * 4, s);
            (void)zero_async(dgamma[l], (size_t)couts[l] * 4, s);
            (void)zero_async(dbeta[l], (size_t)couts[l] * 4, s);
        }
    }
    const long long batches = E / rows_per_batch;
    if (!wn_only && batches * N > 0 && zero_async(du, (size_t)batches * N * CH * 4, s) != hipSuccess)
        return fail(PCF_E_LAUNCH, "pcf_chain_backward: memset");
    if (E == 0) return ok();
    PCF_REQUIRE(vi && a2_acc && dw && (wn_only || (idx && h1_acc && dscore)), "pcf_chain_backward: null pointer");
    PCF_REQUIRE(aligned16(h1_acc) && aligned16(a2_acc) && aligned16(dscore) && aligned16(dw) && aligned16(du),
                "pcf_chain_backward: buffers must be 16-byte aligned");
    PCF_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= pcf_hip_pcf_chain_backward_workspace_bytes(E),
                "pcf_chain_backward: workspace too small or misaligned");
    PCF_REQUIRE(batches * N < (1ll << 31), "pcf_chain_backward: too many points");
    ChainBwdArgs a{};
    a.f.vi = vi; a.f.idx = idx; a.f.u = nullptr; a.f.E = E; a.f.rows_per_batch = rows_per_batch; a.f.N = N; a.f.K = K;
    a.f.cv = cv; a.f.g = g; a.f.heads = heads; a.f.cm = cm;
    a.f.vec_vi = (cv % 4 == 0) && aligned16(vi);
    float* means = static_cast<float*>(workspace);           // [12][64]: mean g, then mean g*xhat, per layer
    a.part = means + 12 * 64;
    a.part_top = a.part + (size_t)1024 * NDW * 256;
    a.part_sums = a.part_top + (size_t)1024 * 512;
    float* redbuf = a.part_sums + (size_t)2048 * 96;   // RED_TOTAL <= 4096 floats
    a.gh1 = redbuf + 4096;                             // 16-byte aligned: every region is a multiple of 64 floats
    a.ga2 = a.gh1 + (size_t)E * CH;
    for (int l = 0; l < 6; ++l) {
        a.f.W[l] = W[l]; a.f.b[l] = b[l]; a.f.gamma[l] = gamma[l]; a.f.beta[l] = beta[l];
        a.f.mean[l] = stats + l * 64; a.f.rstd[l] = stats + (6 + l) * 64;
        a.gmean[l] = means + l * 64; a.gxmean[l] = means + (6 + l) * 64;
    }
    a.dscore = dscore; a.dw = dw; a.du = du; a.h1_acc = h1_acc; a.a2_acc = a2_acc; a.wn_only = wn_only ? 1 : 0;
    a.f.ukey = ukey; a.dukey = dukey;
    // both branches need at least one workgroup; a multiple of 8 keeps the 5 : 3 split exact
    const int grid = std::max(8, (chain_grid(E) + 7) / 8 * 8);
    // pass 1 mostly streams (2048 workgroups keep more bytes in flight: 57 -> 49 us); passes 2-3 stay at 1024
    const long long wg_needed = (E / 16 + NWAVE - 1) / NWAVE;
    const int grid1 = std::max(8, (int)((std::min<long long>(wg_needed, 2048) + 7) / 8 * 8));
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 0) hipLaunchKernelGGL(pcf_chain_bwd_kernel<1>, dim3(grid1), dim3(BLOCK), 0, s, a, grid1 / 8);
        else hipLaunchKernelGGL(pcf_chain_bwd_kernel<2>, dim3(grid), dim3(BLOCK), 0, s, a, grid / 8);
        if (int e = check_launch("pcf_chain_backward pass")) return e;
        BwdFinArgs fa{};
        fa.part = a.part_sums; fa.nblocks = pass == 0 ? grid1 : grid; fa.R = E;
        auto set = [&](int q, int layer, int chan0, int count) {
            fa.g[q].dbeta = dbeta[layer]; fa.g[q].dgamma = dgamma[layer];
            fa.g[q].gmean = means + layer * 64; fa.g[q].gxmean = means + (6 + layer) * 64;
            fa.g[q].mean = stats + layer * 64; fa.g[q].rstd = stats + (6 + layer) * 64; fa.g[q].bias = b[layer];
            fa.g[q].chan0 = chan0; fa.g[q].count = count;
        };
        // (guidance workgroups fill group 0, WeightNet workgroups group 1; the rest are zeros)
        if (pass == 0) { if (!wn_only) set(0, L_G2, 0, heads); set(1, L_W3, 0, cm); }
        else { if (!wn_only) set(0, L_G1, 0, CH); set(1, L_W2, 0, CH); }
        hipLaunchKernelGGL(chain_bwd_finalize_kernel, dim3(12), dim3(1024), 0, s, fa);
        if (int e = check_launch("pcf_chain_backward finalize")) return e;
    }
    if (chain_bwd_lds_transposes() || ukey) {          // the maximum-key form exists with the LDS transposes only
        hipLaunchKernelGGL((pcf_chain_bwd_kernel<3, false>), dim3(grid), dim3(BLOCK), 0, s, a, grid / 8);
        if (int e = check_launch("pcf_chain_bwd_kernel<3> (LDS transposes)")) return e;
    } else {
        hipLaunchKernelGGL((pcf_chain_bwd_kernel<3, true>), dim3(grid), dim3(BLOCK), 0, s, a, grid / 8);
        if (int e = check_launch("pcf_chain_bwd_kernel<3> (register layouts)")) return e;
    }
    BwdReduceArgs ra{a.part, a.part_top, a.part_sums, grid, redbuf};
    hipLaunchKernelGGL(chain_bwd_reduce_kernel, dim3((RED_TOTAL + 63) / 64), dim3(1024), 0, s, ra);
    BwdCombineArgs ca{};
    ca.red = redbuf; ca.W_pe = W[L_PE]; ca.W_w1 = W[L_W1]; ca.stats = stats;
    ca.b_pe = b[L_PE]; ca.b_w1 = b[L_W1]; ca.gamma_pe = gamma[L_PE]; ca.gamma_w1 = gamma[L_W1];
    for (int l = 0; l < 6; ++l) { ca.dW[l] = dW[l]; ca.db[l] = db[l]; }
    ca.dgamma_pe = dgamma[L_PE]; ca.dbeta_pe = dbeta[L_PE]; ca.dgamma_w1 = dgamma[L_W1]; ca.dbeta_w1 = dbeta[L_W1];
    ca.R = E; ca.cv = cv; ca.g = g; ca.heads = heads; ca.cm = cm;
    hipLaunchKernelGGL(chain_bwd_combine_kernel, dim3(1), dim3(BLOCK), 0, s, ca);
    return check_launch("pcf_chain_backward parameter gradients");
}


extern "C" {

int pcf_hip_set_chain_backward_engine(int lds_transposes) {
    if (lds_transposes != 0 && lds_transposes != 1)
        return pcf::fail(PCF_E_BADARG, "set_chain_backward_engine: 0 (register layouts) or 1 (LDS transposes), got %d", lds_transposes);
    g_chain_bwd_lds.store(lds_transposes, std::memory_order_relaxed);
    return pcf::ok();
}

size_t pcf_hip_pcf_chain_backward_workspace_bytes(long long E) {
    // per-workgroup partials of passes 1-4, the 12 x 64 per-channel means, and the two [E, 8] gradients pass 2 hands
    // to pass 3
    return ((size_t)1024 * pcf::NDW * 256 + (size_t)1024 * 512 + (size_t)2048 * 96 + 4096 + 12 * 64) * 4 +
           (size_t)(E > 0 ? E : 0) * 2 * pcf::CH * 4 + 1024;
}

int pcf_hip_pcf_chain_backward(const float* vi, const int64_t* idx, const float* h1_acc, const float* a2_acc,
                               const float* dscore, const float* dw, long long E, long long rows_per_batch, int N, int K,
                               int cv, int g, int heads, int cm, const float* const* W, const float* const* b,
                               const float* const* gamma, const float* const* beta, const float* stats, float* du,
                               float* const* dW, float* const* db, float* const* dgamma, float* const* dbeta,
                               void* workspace, size_t workspace_bytes, void* stream) {
    return chain_backward_impl(nullptr, nullptr, false, vi, idx, h1_acc, a2_acc, dscore, dw, E, rows_per_batch, N, K, cv, g, heads, cm, W, b,
                               gamma, beta, stats, du, dW, db, dgamma, dbeta, workspace, workspace_bytes, stream);
}

// adjoint of pcf_hip_pcf_chain_forward_maxkey: additionally dukey [E / K, 8] (ukey itself is not read: a non-null pointer
// selects the form)
int pcf_hip_pcf_chain_backward_maxkey(const float* ukey, float* dukey, const float* vi, const int64_t* idx, const float* h1_acc,
                                      const float* a2_acc, const float* dscore, const float* dw, long long E,
                                      long long rows_per_batch, int N, int K, int cv, int g, int heads, int cm,
                                      const float* const* W, const float* const* b, const float* const* gamma,
                                      const float* const* beta, const float* stats, float* du, float* const* dW,
                                      float* const* db, float* const* dgamma, float* const* dbeta, void* workspace,
                                      size_t workspace_bytes, void* stream) {
    if (!ukey || !dukey) return pcf::fail(PCF_E_BADARG, "pcf_chain_backward_maxkey: ukey / dukey is null");
    return chain_backward_impl(ukey, dukey, false, vi, idx, h1_acc, a2_acc, dscore, dw, E, rows_per_batch, N, K, cv, g, heads, cm, W,
                               b, gamma, beta, stats, du, dW, db, dgamma, dbeta, workspace, workspace_bytes, stream);
}

// WeightNet alone (adjoint of pcf_hip_weightnet_chain_forward): layer order w1, w2, w3 in every array.
int pcf_hip_weightnet_chain_backward(const float* x, const float* a2_acc, const float* dw, long long E, int cin, int cm,
                                     const float* const* W, const float* const* b, const float* const* gamma,
                                     const float* const* beta, const float* stats, float* const* dW, float* const* db,
                                     float* const* dgamma, float* const* dbeta, void* workspace, size_t workspace_bytes,
                                     void* stream) {
    using namespace pcf;
    PCF_REQUIRE(W && b && gamma && beta && dW && db && dgamma && dbeta, "weightnet_chain_backward: null pointer");
    const float* W6[6] = {nullptr, nullptr, nullptr, W[0], W[1], W[2]};
    const float* b6[6] = {nullptr, nullptr, nullptr, b[0], b[1], b[2]};
    const float* g6[6] = {nullptr, nullptr, nullptr, gamma[0], gamma[1], gamma[2]};
    const float* be6[6] = {nullptr, nullptr, nullptr, beta[0], beta[1], beta[2]};
    float* dW6[6] = {nullptr, nullptr, nullptr, dW[0], dW[1], dW[2]};
    float* db6[6] = {nullptr, nullptr, nullptr, db[0], db[1], db[2]};
    float* dg6[6] = {nullptr, nullptr, nullptr, dgamma[0], dgamma[1], dgamma[2]};
    float* dbe6[6] = {nullptr, nullptr, nullptr, dbeta[0], dbeta[1], dbeta[2]};
    return chain_backward_impl(nullptr, nullptr, true, x, nullptr, nullptr, a2_acc, nullptr, dw, E, E > 0 ? E : 1, 0, 1, cin, 0, 0, cm, W6, b6, g6,
                               be6, stats, nullptr, dW6, db6, dg6, dbe6, workspace, workspace_bytes, stream);
}

}  // extern "C"
