// Shared declarations of the fused PCFLayer edge-graph kernels (edge_chain.hip forward, edge_chain_bwd.hip backward).
#pragma once
#include "pcf_common.h"

namespace pcf {

constexpr int CV = 12;     // max width of the WeightNet input (VI: 12, plain offsets: 3)
constexpr int CG = 32;     // max width of the positional encoding (guidance_feat_len)
constexpr int CH = 8;      // hidden width of the guidance MLP and of WeightNet
constexpr int CHD = 8;     // max heads
constexpr int CMX = 16;    // max C_mid

enum { L_PE = 0, L_G1 = 1, L_G2 = 2, L_W1 = 3, L_W2 = 4, L_W3 = 5 };

struct ChainArgs {
    const float* vi;            // [E, cv]
    const int64_t* idx;         // [E] batch-local neighbour index of every edge (for u)
    const float* u;             // [B*N, 8]
    long long E, rows_per_batch;
    int N, K, cv, g, heads, cm;
    const float* W[6];
    const float* b[6];
    const float* gamma[6];
    const float* beta[6];
    const float* mean[6];       // device [64] each; filled pass by pass
    const float* rstd[6];
    float* pe; float* a1; float* h1; float* a2; float* score; float* w;
    float* h1_acc; float* a2_acc;   // [E, 8] raw accumulators (pre-BatchNorm, bias not added) of g1 and w2, for the fused backward
    float* part;                // [blocks][2][64] partial sums of the pass
    int vec_vi;
    // Strided layers (layers.py:372-375): the key is the MAXIMUM of the query over the neighbourhood instead of neighbour 0.
    // Non-null ukey selects that form: ukey [centres, 8] = Wa . max_k guidance_x[idx[n, k]] (the gathered half, formed per
    // centre by the caller); the positional half Wb . max_k pe[n, k] is taken over the K lanes of the neighbourhood (2 <= K).
    const float* ukey;
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- the chain on the matrix cores --------------------------------------------------------------------
// Transposed formulation: for a tile of 16 edges, Z[o][p] = sum_c W[o][c] * Y[c][p] with
// v_mfma_f32_16x16x4_f32 (exact fp32).  Accumulator layout (C/D map): lane l, register r holds
// channel 4*(l>>4) + r of edge p = l & 15.  The B operand wants, for contraction step k = l>>4, one value
// per lane of edge l & 15 -- so register s of the previous layer's accumulator IS the B operand of
// contraction step s if that step is defined to cover channels {4k + s}: the contraction order is a free
// choice as long as the weight fragment follows it (A[o][k] = W[o][4k + s]).  Hence
//   * the six weight matrices live in 32 VGPRs for the whole kernel (loaded once),
//   * a layer's output feeds the next layer with NO data movement (no LDS, no shuffles),
//   * BatchNorm parameters are per-lane constants (channel 4*(l>>4)+r), statistics are per-lane running
//     sums over tiles, reduced across the 16 edge lanes once at the end.
// K <= 16 so a neighbourhood never straddles a tile (the key edge is lane l & ~(K-1) of the same group).
__device__ __forceinline__ float wfrag(const float* W, int Cout, int Cin, int o, int c) {
    return (W && o < Cout && c < Cin) ? W[o * Cin + c] : 0.f;
}

// Batch of edge e0 + p (e0 = 16 * tile, wave-uniform) for a non-decreasing sequence of tiles, without the
// software 64-bit division `e / rows_per_batch` costs per call (~100 VALU instructions).  rows_per_batch >= 16,
// so at most one batch boundary falls inside a tile.
struct BatchWalk {
    long long rpb, next_start;
    int b;
    __device__ __forceinline__ void init(long long rows_per_batch) { rpb = rows_per_batch; next_start = rows_per_batch; b = 0; }
    __device__ __forceinline__ int batch_of(long long e0, int p) {
        if (e0 >= next_start) {
            if (e0 - next_start < 4 * rpb) {
                do { ++b; next_start += rpb; } while (e0 >= next_start);
            } else {
                b = (int)(e0 / rpb);
                next_start = (long long)(b + 1) * rpb;
            }
        }
        return b + (e0 + p >= next_start ? 1 : 0);
    }
};

#define PCF_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_16x16x4f32((A), (B), (C), 0, 0, 0)

// reductions over lanes of a row (the 16 edges of a tile, one register each): DPP only, every lane gets the result
template <int CTRL> __device__ __forceinline__ float row_dpp(float v) {
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xf, 0xf, true));
}
// over the K consecutive lanes of a neighbourhood (K a power of two, 2..16; groups are K-aligned)
__device__ __forceinline__ float nbr_max(float v, int K) {
    v = fmaxf(v, row_dpp<0xB1>(v));                      // quad_perm [1,0,3,2]
    if (K >= 4) v = fmaxf(v, row_dpp<0x4E>(v));          // quad_perm [2,3,0,1]
    if (K >= 8) v = fmaxf(v, row_dpp<0x141>(v));         // row_half_mirror
    if (K >= 16) v = fmaxf(v, row_dpp<0x140>(v));        // row_mirror
    return v;
}
__device__ __forceinline__ float nbr_min(float v, int K) {
    v = fminf(v, row_dpp<0xB1>(v));
    if (K >= 4) v = fminf(v, row_dpp<0x4E>(v));
    if (K >= 8) v = fminf(v, row_dpp<0x141>(v));
    if (K >= 16) v = fminf(v, row_dpp<0x140>(v));
    return v;
}
__device__ __forceinline__ f32x4 nbr_max(f32x4 v, int K) { return f32x4{nbr_max(v[0], K), nbr_max(v[1], K), nbr_max(v[2], K), nbr_max(v[3], K)}; }

// up to 4 workgroups per CU; every wave walks tiles with a grid stride
inline int chain_grid(long long E) {
    const long long tiles = E / 16;
    const long long g = (tiles + NWAVE - 1) / NWAVE;
    return (int)(g < 1 ? 1 : (g > 1024 ? 1024 : g));
}

}  // namespace pcf
