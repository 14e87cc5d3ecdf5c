// fp32 MFMA contractions for the linear stage that follows the aggregate (and its two backward
// products).  gfx950 has an exact-fp32 matrix instruction, v_mfma_f32_32x32x2_f32 (one rounding per
// product, k-ordered, bit-identical to an fmaf chain), so the results match a plain fp32 reference
// to rounding and the 1e-3 parity bar needs no reduced-precision argument.
//
//   C[M,N] = sum_k A(m,k) * B(k,n)  (+ bias[n])
//
// A and B are each addressed either "k-contiguous" (row-major [rows, Kd]) or "row-contiguous"
// ([Kd, rows]); that covers the three products of the path without transposed copies:
//   forward      out   = pconv_out . lin_w^T       A k-contig, B k-contig        (pconv_ops.cu:1092-1131)
//   backward #1  d_pcv = grad_out . lin_w          A k-contig, B row-contig      (pconv_ops.cu:328-334)
//   backward #2  d_W   = grad_out^T . pconv_out    A row-contig, B row-contig, split over Kd
//                                                                              (pconv_ops.cu:376-387,517-533)
// Workgroup = 4 waves, each owning a 32x32 accumulator tile (16 VGPRs); tile 64x64 (2x2 waves) or
// 128x32 (4x1, for narrow outputs); K-step 16 staged through LDS with register prefetch of the next
// step.  The products here are memory-bound (arithmetic intensity <= Co/2 flop per byte of A), so
// the tile shape is chosen to read A exactly once whenever N <= 64.
#include <algorithm>

#include "pcf_common.h"

namespace pcf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GBK = 16;

struct GemmArgs {
    const float* A;
    const float* Bm;
    const float* bias;
    float* C;
    int M, N, Kd;
    int lda, ldb, ldc;
    int k_per_split;   // Kd range per blockIdx.z (multiple of GBK)
    long long c_split_stride;
};

// Global -> registers for one [ROWS x GBK] operand tile.  KCONT: element (r,k) at p[r*ld + k];
// otherwise at p[k*ld + r].  Out-of-range elements read as zero.
template <int ROWS, bool KCONT, bool VEC>
struct TileLoader {
    static constexpr int ELEMS = ROWS * GBK;
    static constexpr int UNITS = ELEMS / 4;                      // float4 units in the tile
    static constexpr int NV = (UNITS + BLOCK - 1) / BLOCK;       // units per thread (last may be idle)
    float4 v[NV];

    __device__ __forceinline__ void load(const float* p, int ld, int r0, int k0, int rmax, int kmax) {
        const int tid = threadIdx.x;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int u = tid + i * BLOCK;     // float4 unit
            if (UNITS % BLOCK != 0 && u >= UNITS) break;
            int r, k;
            if (KCONT) { r = u / (GBK / 4); k = (u % (GBK / 4)) * 4; }
            else       { k = u / (ROWS / 4); r = (u % (ROWS / 4)) * 4; }
            const int gr = r0 + r, gk = k0 + k;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (KCONT) {
                if (gr < rmax) {
                    const float* q = p + (size_t)gr * ld + gk;
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
                    const float* q = p + (size_t)gk * ld + gr;
                    if (VEC && gr + 3 < rmax) t = ld4(q);
                    else {
                        if (gr < rmax) t.x = q[0];
                        if (gr + 1 < rmax) t.y = q[1];
                        if (gr + 2 < rmax) t.z = q[2];
                        if (gr + 3 < rmax) t.w = q[3];
                    }
                }
            }
            v[i] = t;
        }
    }

    // LDS image is always [k][ROWS + pad] so that an MFMA fragment read (32 consecutive rows at one k)
    // is conflict-free: KCONT tiles are transposed on the way in.
    static constexpr int LDS_STRIDE = ROWS + 4;
    __device__ __forceinline__ void store(float* s) const {
        const int tid = threadIdx.x;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int u = tid + i * BLOCK;
            if (UNITS % BLOCK != 0 && u >= UNITS) break;
            if (KCONT) {
                const int r = u / (GBK / 4), k = (u % (GBK / 4)) * 4;
                s[(k + 0) * LDS_STRIDE + r] = v[i].x;
                s[(k + 1) * LDS_STRIDE + r] = v[i].y;
                s[(k + 2) * LDS_STRIDE + r] = v[i].z;
                s[(k + 3) * LDS_STRIDE + r] = v[i].w;
            } else {
                const int k = u / (ROWS / 4), r = (u % (ROWS / 4)) * 4;
                st4(s + k * LDS_STRIDE + r, v[i]);
            }
        }
    }
};

template <int WM, int WN, bool A_KCONT, bool B_KCONT, bool VEC>
__global__ __launch_bounds__(BLOCK) void gemm_f32_kernel(const GemmArgs g) {
    constexpr int BM = WM * 32, BN = WN * 32;
    using LA = TileLoader<BM, A_KCONT, VEC>;
    using LB = TileLoader<BN, B_KCONT, VEC>;
    __shared__ __align__(16) float sA[GBK * LA::LDS_STRIDE];
    __shared__ __align__(16) float sB[GBK * LB::LDS_STRIDE];

    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kbeg = blockIdx.z * g.k_per_split;
    const int kend = min(g.Kd, kbeg + g.k_per_split);
    const int wave = wave_id(), lane = lane_id();
    const int wm = wave / WN, wn = wave % WN;
    const int frow = lane & 31, fk = lane >> 5;

    f32x16 acc = {0};
    LA la;
    LB lb;
    if (kbeg < kend) {
        la.load(g.A, g.lda, m0, kbeg, g.M, kend);
        lb.load(g.Bm, g.ldb, n0, kbeg, g.N, kend);
    }
    for (int k0 = kbeg; k0 < kend; k0 += GBK) {
        __syncthreads();               // previous step's fragment reads are done
        la.store(sA);
        lb.store(sB);
        __syncthreads();
        if (k0 + GBK < kend) {         // prefetch next step while the MFMAs run
            la.load(g.A, g.lda, m0, k0 + GBK, g.M, kend);
            lb.load(g.Bm, g.ldb, n0, k0 + GBK, g.N, kend);
        }
        const float* pa = sA + wm * 32 + frow;
        const float* pb = sB + wn * 32 + frow;
#pragma unroll
        for (int kk = 0; kk < GBK; kk += 2) {
            const float a = pa[(kk + fk) * LA::LDS_STRIDE];
            const float b = pb[(kk + fk) * LB::LDS_STRIDE];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
    // C/D map of the 32x32 tile: col = lane & 31, row = (r & 3) + 8*(r >> 2) + 4*(lane >> 5)
    float* C = g.C + (size_t)blockIdx.z * g.c_split_stride;
    const int col = n0 + wn * 32 + (lane & 31);
    if (col < g.N) {
        const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < g.M) C[(size_t)row * g.ldc + col] = acc[r] + bv;
        }
    }
}

// Sum split-K slabs: C[i] = sum_s slab[s][i] in a fixed order (deterministic).  SL slices of the slab
// list per element, four loads in flight per lane.
template <int SL>
__global__ __launch_bounds__(64 * SL) void slab_sum_kernel(const float* __restrict__ slabs, float* __restrict__ C,
                                                           long long count, int splits) {
    __shared__ float sh[64 * SL];
    const long long i = (long long)blockIdx.x * 64 + (threadIdx.x & 63);
    const int slice = threadIdx.x >> 6;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (i < count) {
        int s = slice;
        for (; s + 3 * SL < splits; s += 4 * SL) {
            a0 += slabs[(size_t)s * count + i];
            a1 += slabs[(size_t)(s + SL) * count + i];
            a2 += slabs[(size_t)(s + 2 * SL) * count + i];
            a3 += slabs[(size_t)(s + 3 * SL) * count + i];
        }
        for (; s < splits; s += SL) a0 += slabs[(size_t)s * count + i];
    }
    sh[threadIdx.x] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (threadIdx.x < 64 && i < count) {
        float t = 0.f;
#pragma unroll
        for (int sl = 0; sl < SL; ++sl) t += sh[sl * 64 + threadIdx.x];
        C[i] = t;
    }
}

// Column sums of a [rows, cols] matrix in two deterministic stages (grad of the linear bias).  A workgroup
// owns a band of rows; its 256 threads are laid out TX columns x 256/TX row lanes so that narrow matrices
// still use every lane, and the row lanes are combined through LDS in a fixed order.
__global__ __launch_bounds__(BLOCK) void colsum_partial_kernel(const float* __restrict__ src, float* __restrict__ part,
                                                               int rows, int cols, int rows_per_block, int tx) {
    __shared__ float sh[BLOCK];
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(rows, r0 + rows_per_block);
    const int ty = BLOCK / tx, cx = threadIdx.x % tx, ry = threadIdx.x / tx;
    for (int cb = 0; cb < cols; cb += tx) {
        const int c = cb + cx;
        float acc0 = 0.f, acc1 = 0.f;
        if (c < cols) {
            int r = r0 + ry;
            for (; r + ty < r1; r += 2 * ty) {
                acc0 += src[(size_t)r * cols + c];
                acc1 += src[(size_t)(r + ty) * cols + c];
            }
            if (r < r1) acc0 += src[(size_t)r * cols + c];
        }
        sh[threadIdx.x] = acc0 + acc1;
        __syncthreads();
        if (ry == 0 && c < cols) {
            float t = 0.f;
            for (int j = 0; j < ty; ++j) t += sh[j * tx + cx];
            part[(size_t)blockIdx.x * cols + c] = t;
        }
        __syncthreads();
    }
}

template <int WM, int WN, bool AK, bool BK_>
static int launch_gemm(const GemmArgs& g, int splits, bool vec, hipStream_t s) {
    dim3 grid(ceil_div(g.N, WN * 32), ceil_div(g.M, WM * 32), splits);
    if (vec) hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, AK, BK_, true>), grid, dim3(BLOCK), 0, s, g);
    else hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, AK, BK_, false>), grid, dim3(BLOCK), 0, s, g);
    return check_launch("fp32 MFMA contraction");
}

template <bool AK, bool BK_>
static int dispatch_tile(const GemmArgs& g, int splits, bool vec, hipStream_t s) {
    if (g.N <= 32) return launch_gemm<4, 1, AK, BK_>(g, splits, vec, s);
    return launch_gemm<2, 2, AK, BK_>(g, splits, vec, s);
}

// C = A . B with the addressing modes above.  splits > 1 writes `splits` partial slabs to C
// (C must then hold splits*M*N floats).
int gemm_f32(const float* A, bool a_kcont, int lda, const float* Bm, bool b_kcont, int ldb, const float* bias, float* C,
             int ldc, int M, int N, int Kd, int splits, hipStream_t s) {
    if (M == 0 || N == 0) return ok();
    GemmArgs g{A, Bm, bias, C, M, N, Kd, lda, ldb, ldc, 0, 0};
    if (splits < 1) splits = 1;
    int kps = ceil_div(std::max(Kd, 1), splits);
    kps = (kps + GBK - 1) / GBK * GBK;
    g.k_per_split = kps;
    g.c_split_stride = (long long)M * ldc;
    const bool vec = aligned16(A) && aligned16(Bm) && (lda % 4 == 0) && (ldb % 4 == 0);
    if (a_kcont && b_kcont) return dispatch_tile<true, true>(g, splits, vec, s);
    if (a_kcont && !b_kcont) return dispatch_tile<true, false>(g, splits, vec, s);
    if (!a_kcont && !b_kcont) return dispatch_tile<false, false>(g, splits, vec, s);
    return dispatch_tile<false, true>(g, splits, vec, s);
}

int choose_splits(int M, int N, int Kd) {
    const long long tiles = (long long)ceil_div(M, N <= 32 ? 128 : 64) * ceil_div(N, N <= 32 ? 32 : 64);
    long long want = (1024 + tiles - 1) / tiles;             // ~4 workgroups per CU overall
    const long long max_by_k = std::max(1, Kd / (GBK * 8));  // at least 8 K-steps per split
    return (int)std::max<long long>(1, std::min<long long>(std::min<long long>(want, max_by_k), 512));
}

int slab_sum(const float* slabs, float* C, long long count, int splits, hipStream_t s) {
    if (count == 0) return ok();
    const int grid = (int)((count + 63) / 64);
    // few elements, many slabs (bias gradients): spread the slab list over 16 waves of one workgroup
    if (splits >= 64) hipLaunchKernelGGL(slab_sum_kernel<16>, dim3(grid), dim3(1024), 0, s, slabs, C, count, splits);
    else hipLaunchKernelGGL(slab_sum_kernel<4>, dim3(grid), dim3(256), 0, s, slabs, C, count, splits);
    return check_launch("split-K slab sum");
}

// out[cols] = column sums of src[rows, cols]; `part` holds colsum_blocks(rows)*cols floats.
int colsum_blocks(int rows) { return std::max(1, std::min(1024, ceil_div(rows, 64))); }
int colsum(const float* src, float* part, float* out, int rows, int cols, hipStream_t s) {
    if (cols == 0) return ok();
    const int nb = colsum_blocks(rows);
    const int rpb = ceil_div(std::max(rows, 1), nb);
    int tx = 1;
    while (tx < cols && tx < BLOCK) tx <<= 1;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nb), dim3(BLOCK), 0, s, src, part, rows, cols, rpb, tx);
    if (int e = check_launch("bias-gradient partial sums")) return e;
    return slab_sum(part, out, cols, nb, s);
}

}  // namespace pcf

extern "C" int pcf_hip_gemm_nt(const float* A, const float* Bm, const float* bias, float* C, int M, int N, int Kd,
                               void* stream) {
    if (M < 0 || N < 0 || Kd < 0) return pcf::fail(PCF_E_BADARG, "gemm_nt: negative size");
    if (M > 0 && N > 0 && (!A || !Bm || !C) && Kd > 0) return pcf::fail(PCF_E_BADARG, "gemm_nt: null pointer");
    return pcf::gemm_f32(A, true, Kd, Bm, true, Kd, bias, C, N, M, N, Kd, 1, (hipStream_t)stream);
}
