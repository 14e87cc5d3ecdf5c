// k nearest neighbours over a packed batch, and the CSR transpose of a neighbour table (gfx950).
//
// kNN: exact brute force.  One lane = one query, which keeps its K best (distance, index) pairs
// sorted in registers; a workgroup streams its sample's reference points through LDS in tiles and
// every lane reads the same reference at the same time (LDS broadcast, one ds_read_b128 per
// reference per wave).  Distances are ((rx-qx)^2 + (ry-qy)^2) + (rz-qz)^2 with one fp32 rounding per
// operation (no fma contraction), references are visited in index order and an equal distance never
// displaces an earlier one, so the result is the unique (distance, index)-ascending list and is
// bit-identical to oracle/knn_ref.c.  Replaces the per-(sample, level, relation) KeOps argKmin calls
// of knn_post_dataloader_utils.py:22-87,171-223 with one launch per (level, relation).
//
// CSR transpose: histogram (int atomics) -> 3-kernel exclusive scan -> atomic fill of edge ids ->
// per-bucket sort by edge id, so every bucket lists its (query, k) pairs in ascending order whatever
// order the atomics landed in.  Replaces count_neighbors / compute_inv_idx (a single-thread scan) /
// fill_inverse of knn.cu:24-168, whose bucket order is run-to-run random.
#include <algorithm>
#include <vector>

#include "pcf_common.h"

namespace pcf {

// =================================================================================================
// kNN
// =================================================================================================
constexpr int KNN_TILE = 2048;   // reference points per LDS tile (32 KiB as float4)

template <int KMAX>
__global__ __launch_bounds__(BLOCK) void knn_kernel(const float* __restrict__ ref, const float* __restrict__ query,
                                                    const int32_t* __restrict__ ref_off,
                                                    const int32_t* __restrict__ query_off, int K,
                                                    int64_t* __restrict__ out) {
    __shared__ float4 tile[KNN_TILE];
    const int seg = blockIdx.y;
    const int q0 = query_off[seg], q1 = query_off[seg + 1];
    const int r0 = ref_off[seg], r1 = ref_off[seg + 1];
    if ((int)(blockIdx.x * BLOCK) >= q1 - q0) return;     // whole workgroup leaves together
    const int q = q0 + blockIdx.x * BLOCK + threadIdx.x;
    const bool active = q < q1;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (active) { qx = query[3 * (size_t)q]; qy = query[3 * (size_t)q + 1]; qz = query[3 * (size_t)q + 2]; }

    float bd[KMAX];
    int bi[KMAX];
#pragma unroll
    for (int s = 0; s < KMAX; ++s) { bd[s] = __builtin_inff(); bi[s] = -1; }

    for (int t0 = r0; t0 < r1; t0 += KNN_TILE) {
        const int cnt = min(KNN_TILE, r1 - t0);
        __syncthreads();
        for (int j = threadIdx.x; j < cnt; j += BLOCK) {
            const float* p = ref + 3 * (size_t)(t0 + j);
            tile[j] = make_float4(p[0], p[1], p[2], 0.f);
        }
        __syncthreads();
        if (active) {
            for (int j = 0; j < cnt; ++j) {
                const float4 r = tile[j];
                const float dx = __fsub_rn(r.x, qx), dy = __fsub_rn(r.y, qy), dz = __fsub_rn(r.z, qz);
                const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
                if (d < bd[KMAX - 1]) {
                    const int id = t0 + j;
#pragma unroll
                    for (int s = KMAX - 1; s > 0; --s) {
                        const bool shift = d < bd[s - 1];       // the old element s-1 moves to s
                        const bool here = d < bd[s];            // ... otherwise d lands at s if it beats the old s
                        bi[s] = shift ? bi[s - 1] : (here ? id : bi[s]);
                        bd[s] = shift ? bd[s - 1] : (here ? d : bd[s]);
                    }
                    if (d < bd[0]) { bd[0] = d; bi[0] = id; }
                }
            }
        }
    }
    if (active) {
#pragma unroll
        for (int s = 0; s < KMAX; ++s)
            if (s < K) out[(size_t)q * K + s] = (int64_t)bi[s];
    }
}


// ---- one wave per query: the coarse levels -------------------------------------------------------------------
// With a few thousand queries the lane-per-query kernels above and in knn_grid.hip are pure latency (a level of
// 1.3k points takes 100-220 us: six workgroups walking their candidates one by one).  Here the 64 lanes of a wave
// share one query: its sample's references are taken 64 x WPER at a time, every lane holds WPER candidate keys
// (distance bits << 32 | packed index: unsigned order == the oracle's (distance, index) order) plus one slot for a
// carried best, and K rounds of {lane minimum, xor-butterfly minimum over the wave, winner drops its key} extract the
// K smallest; the K best so far ride along in lanes 0..K-1 from chunk to chunk.  Same distance expression and
// roundings as knn_kernel, so the lists are bit-identical.  Cost ~ queries x ceil(refs per sample / 1024).
constexpr int WPER = 16;
constexpr unsigned long long KEY_NONE = ~0ull;

// minimum over the 64 lanes, returned to all of them: DPP moves (quad swaps, row shifts, row broadcasts) and one
// readlane of lane 63 -- an xor butterfly of 64-bit values is twelve LDS-crossbar permutes per call, and the selection
// makes K calls per chunk.  The kernel is VALU-bound on its 64-bit compare/select chains (~190 instructions per round),
// which is why the host only picks it for the small levels (pcf_cuda.KNN_WAVE_MAX_WORK).
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_min_u64(unsigned long long v) {
    const unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    const unsigned olo = (unsigned)__builtin_amdgcn_update_dpp((int)lo, (int)lo, CTRL, 0xf, 0xf, false);
    const unsigned ohi = (unsigned)__builtin_amdgcn_update_dpp((int)hi, (int)hi, CTRL, 0xf, 0xf, false);
    const unsigned long long o = ((unsigned long long)ohi << 32) | olo;     // lanes without a source keep their own value
    return o < v ? o : v;
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
    v = dpp_min_u64<0xb1>(v);     // quad_perm [1,0,3,2]
    v = dpp_min_u64<0x4e>(v);     // quad_perm [2,3,0,1]
    v = dpp_min_u64<0x114>(v);    // row_shr:4
    v = dpp_min_u64<0x118>(v);    // row_shr:8   -> lane 15 of every row holds the row's minimum
    v = dpp_min_u64<0x142>(v);    // row_bcast:15 -> lanes of rows 1..3 see the previous row's lane 15
    v = dpp_min_u64<0x143>(v);    // row_bcast:31 -> lane 63 holds the minimum of the wave
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, WAVE - 1);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), WAVE - 1);
    return ((unsigned long long)hi << 32) | lo;
}

__global__ __launch_bounds__(BLOCK) void knn_wave_kernel(const float* __restrict__ ref, const float* __restrict__ query,
                                                         const int32_t* __restrict__ ref_off,
                                                         const int32_t* __restrict__ query_off, int n_seg, int n_query,
                                                         int K, int64_t* __restrict__ out) {
    const int lane = lane_id();
    for (int q = blockIdx.x * NWAVE + wave_id(); q < n_query; q += gridDim.x * NWAVE) {
        int lo = 0, hi = n_seg;                              // query_off[lo] <= q < query_off[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (query_off[mid] <= q) lo = mid; else hi = mid;
        }
        const int r0 = ref_off[lo], r1 = ref_off[lo + 1];
        const float qx = query[3 * (size_t)q], qy = query[3 * (size_t)q + 1], qz = query[3 * (size_t)q + 2];
        unsigned long long carry = KEY_NONE;                 // lane j < K: the j-th best so far
        for (int c0 = r0; c0 < r1; c0 += WAVE * WPER) {
            unsigned long long pool[WPER + 1];
            float rx[WPER], ry[WPER], rz[WPER];
#pragma unroll
            for (int s = 0; s < WPER; ++s) {                 // unconditional (clamped) loads: all of them in flight at once
                const float* p = ref + 3 * (size_t)min(c0 + s * WAVE + lane, r1 - 1);
                rx[s] = p[0]; ry[s] = p[1]; rz[s] = p[2];
            }
            asm volatile("" ::: "memory");                   // keeps hipcc from sinking each load next to its use (2 in flight)
#pragma unroll
            for (int s = 0; s < WPER; ++s) {
                const int j = c0 + s * WAVE + lane;
                const float dx = __fsub_rn(rx[s], qx), dy = __fsub_rn(ry[s], qy), dz = __fsub_rn(rz[s], qz);
                const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
                pool[s] = j < r1 ? (((unsigned long long)__float_as_uint(d) << 32) | (unsigned)j) : KEY_NONE;
            }
            pool[WPER] = carry;
            unsigned long long next = KEY_NONE;
            for (int j = 0; j < K; ++j) {
                unsigned long long m = pool[0];
#pragma unroll
                for (int s = 1; s <= WPER; ++s) m = pool[s] < m ? pool[s] : m;
                m = wave_min_u64(m);
                if (m == KEY_NONE) break;                    // fewer than K candidates so far (wave-uniform)
#pragma unroll
                for (int s = 0; s <= WPER; ++s) pool[s] = pool[s] == m ? KEY_NONE : pool[s];
                if (lane == j) next = m;
            }
            carry = next;
        }
        if (lane < K) out[(size_t)q * K + lane] = carry == KEY_NONE ? (int64_t)-1 : (int64_t)(unsigned)(carry & 0xffffffffu);
    }
}

// =================================================================================================
// CSR transpose
// =================================================================================================
__device__ __forceinline__ void csr_count_body(const int64_t* __restrict__ idx, int32_t* __restrict__ counts,
                                                          long long edges, int total_points, int bx, int gx) {
    for (long long e = (long long)bx * BLOCK + threadIdx.x; e < edges; e += (long long)gx * BLOCK) {
        const int64_t t = idx[e];
        if (t >= 0 && t < total_points) atomicAdd(&counts[t], 1);
    }
}
__global__ __launch_bounds__(BLOCK) void csr_count_kernel(const int64_t* __restrict__ idx, int32_t* __restrict__ counts,
                                                          long long edges, int total_points) {
    csr_count_body(idx, counts, edges, total_points, blockIdx.x, gridDim.x);
}

constexpr int SCAN_CHUNK = 1024;   // elements per workgroup (4 per thread)

__device__ __forceinline__ int block_exclusive_scan(int v, int* total) {
    __shared__ int wave_sums[NWAVE];
    const int lane = lane_id(), wave = wave_id();
    int inc = v;
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
        const int t = __shfl_up(inc, off, WAVE);
        if (lane >= off) inc += t;
    }
    if (lane == WAVE - 1) wave_sums[wave] = inc;
    __syncthreads();
    int base = 0, all = 0;
#pragma unroll
    for (int w = 0; w < NWAVE; ++w) {
        const int s = wave_sums[w];
        if (w < wave) base += s;
        all += s;
    }
    __syncthreads();
    *total = all;
    return base + inc - v;
}

// phase a: per-chunk totals
__device__ __forceinline__ void scan_chunk_sums_body(const int32_t* __restrict__ counts,
                                                                int32_t* __restrict__ chunk_sums, int n, int bx, int gx) {
    const int base = bx * SCAN_CHUNK + threadIdx.x * 4;
    int v = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (base + i < n) v += counts[base + i];
    int total;
    block_exclusive_scan(v, &total);
    if (threadIdx.x == 0) chunk_sums[bx] = total;
}
__global__ __launch_bounds__(BLOCK) void scan_chunk_sums_kernel(const int32_t* __restrict__ counts,
                                                                int32_t* __restrict__ chunk_sums, int n) {
    scan_chunk_sums_body(counts, chunk_sums, n, blockIdx.x, gridDim.x);
}

// phase b: exclusive scan of the chunk totals by one workgroup
__device__ __forceinline__ void scan_chunk_offsets_body(int32_t* __restrict__ chunk_sums, int nchunks, int bx, int gx) {
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int c0 = 0; c0 < nchunks; c0 += BLOCK) {
        const int i = c0 + threadIdx.x;
        const int v = i < nchunks ? chunk_sums[i] : 0;
        int total;
        const int ex = block_exclusive_scan(v, &total);
        const int cbase = carry;
        if (i < nchunks) chunk_sums[i] = cbase + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry = cbase + total;
        __syncthreads();
    }
}
__global__ __launch_bounds__(BLOCK) void scan_chunk_offsets_kernel(int32_t* __restrict__ chunk_sums, int nchunks) {
    scan_chunk_offsets_body(chunk_sums, nchunks, blockIdx.x, gridDim.x);
}

// phase c: inv_idx[i] = exclusive prefix of counts; inv_idx[n] = grand total.  `clear` leaves counts zeroed for its
// second life as the per-bucket cursor of the fill pass (saves a memset launch per call).
__device__ __forceinline__ void scan_write_body(int32_t* __restrict__ counts,
                                                           const int32_t* __restrict__ chunk_offsets,
                                                           int32_t* __restrict__ inv_idx, int n, bool clear, int bx, int gx) {
    const int base = bx * SCAN_CHUNK + threadIdx.x * 4;
    int c[4];
    int v = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        c[i] = (base + i < n) ? counts[base + i] : 0;
        v += c[i];
        if (clear && base + i < n) counts[base + i] = 0;
    }
    int total;
    int run = chunk_offsets[bx] + block_exclusive_scan(v, &total);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (base + i < n) inv_idx[base + i] = run;
        run += c[i];
        if (base + i == n - 1) inv_idx[n] = run;
    }
}
__global__ __launch_bounds__(BLOCK) void scan_write_kernel(int32_t* __restrict__ counts,
                                                           const int32_t* __restrict__ chunk_offsets,
                                                           int32_t* __restrict__ inv_idx, int n, bool clear) {
    scan_write_body(counts, chunk_offsets, inv_idx, n, clear, blockIdx.x, gridDim.x);
}

// out[i] = sum of counts[0..i) for i in [0, n]; chunk_tmp holds ceil(n / 1024) ints.  (also used by knn_grid.hip)
int exclusive_scan_i32(int32_t* counts, int32_t* chunk_tmp, int32_t* out, int n, bool clear_counts, hipStream_t s) {
    if (n <= 0) {
        hipError_t e = zero_async(out, 4, s);
        return e == hipSuccess ? ok() : fail(PCF_E_LAUNCH, "scan: %s", hipGetErrorString(e));
    }
    const int nchunks = ceil_div(n, SCAN_CHUNK);
    hipLaunchKernelGGL(scan_chunk_sums_kernel, dim3(nchunks), dim3(BLOCK), 0, s, counts, chunk_tmp, n);
    hipLaunchKernelGGL(scan_chunk_offsets_kernel, dim3(1), dim3(BLOCK), 0, s, chunk_tmp, nchunks);
    hipLaunchKernelGGL(scan_write_kernel, dim3(nchunks), dim3(BLOCK), 0, s, counts, chunk_tmp, out, n, clear_counts);
    return check_launch("exclusive scan");
}

__device__ __forceinline__ void csr_fill_body(const int64_t* __restrict__ idx,
                                                         const int32_t* __restrict__ inv_idx,
                                                         int32_t* __restrict__ cursor, uint32_t* __restrict__ keys,
                                                         long long edges, int total_points, int bx, int gx) {
    for (long long e = (long long)bx * BLOCK + threadIdx.x; e < edges; e += (long long)gx * BLOCK) {
        const int64_t t = idx[e];
        if (t >= 0 && t < total_points) {
            const long long slot = (long long)inv_idx[t] + atomicAdd(&cursor[t], 1);
            // inside the table whenever the histogram was clear on entry; never a wild store if it was not
            if (slot >= 0 && slot < edges) keys[slot] = (uint32_t)e;
        }
    }
}
__global__ __launch_bounds__(BLOCK) void csr_fill_kernel(const int64_t* __restrict__ idx,
                                                         const int32_t* __restrict__ inv_idx,
                                                         int32_t* __restrict__ cursor, uint32_t* __restrict__ keys,
                                                         long long edges, int total_points) {
    csr_fill_body(idx, inv_idx, cursor, keys, edges, total_points, blockIdx.x, gridDim.x);
}

__device__ __forceinline__ void emit(uint32_t key, int pos, int K, int32_t* inv_n, uint8_t* inv_k) {
    const uint32_t n = key / (uint32_t)K;
    inv_n[pos] = (int32_t)n;
    inv_k[pos] = (uint8_t)(key - n * (uint32_t)K);
}

// Buckets of <= 64 entries: one wave each, bitonic network over the lanes.  Larger buckets are
// appended to a work list for csr_sort_large_kernel.
__device__ __forceinline__ void csr_sort_small_body(const uint32_t* __restrict__ keys,
                                                               const int32_t* __restrict__ inv_idx, int total_points,
                                                               int K, int32_t* __restrict__ inv_n,
                                                               uint8_t* __restrict__ inv_k, int32_t* __restrict__ big_list,
                                                               int32_t* __restrict__ big_count, long long edges, int bx, int gx) {
    const int lane = lane_id();
    // slots past the last valid edge (out-of-range neighbour indices leave some) read as zero
    for (long long e = (long long)max(inv_idx[total_points], 0) + (long long)bx * BLOCK + threadIdx.x; e < edges;
         e += (long long)gx * BLOCK) {
        inv_n[e] = 0;
        inv_k[e] = 0;
    }
    for (int t = bx * NWAVE + wave_id(); t < total_points; t += gx * NWAVE) {
        const int beg = inv_idx[t], end = inv_idx[t + 1];
        const int d = end - beg;
        if (d <= 0 || beg < 0 || end > edges) continue;          // the second and third only with a corrupted offset table
        if (d > WAVE) {
            if (lane == 0) big_list[atomicAdd(big_count, 1)] = t;
            continue;
        }
        uint32_t v = lane < d ? keys[beg + lane] : 0xffffffffu;
#pragma unroll
        for (int k = 2; k <= WAVE; k <<= 1) {
#pragma unroll
            for (int j = k >> 1; j > 0; j >>= 1) {
                const uint32_t o = __shfl_xor(v, j, WAVE);
                const bool up = (lane & k) == 0;
                const bool low = (lane & j) == 0;
                v = (low == up) ? min(v, o) : max(v, o);
            }
        }
        if (lane < d) emit(v, beg + lane, K, inv_n, inv_k);
    }
}
__global__ __launch_bounds__(BLOCK) void csr_sort_small_kernel(const uint32_t* __restrict__ keys,
                                                               const int32_t* __restrict__ inv_idx, int total_points,
                                                               int K, int32_t* __restrict__ inv_n,
                                                               uint8_t* __restrict__ inv_k, int32_t* __restrict__ big_list,
                                                               int32_t* __restrict__ big_count, long long edges) {
    csr_sort_small_body(keys, inv_idx, total_points, K, inv_n, inv_k, big_list, big_count, edges, blockIdx.x, gridDim.x);
}

constexpr int SORT_LDS = 4096;   // keys a workgroup sorts in LDS

// Buckets of > 64 entries: one workgroup each.  Up to SORT_LDS keys: bitonic sort in LDS.  Beyond
// that (adversarial tables only): rank by counting, O(d^2 / 256) -- slow but exact.
__device__ __forceinline__ void csr_sort_large_body(const uint32_t* __restrict__ keys,
                                                               const int32_t* __restrict__ inv_idx, int K,
                                                               int32_t* __restrict__ inv_n, uint8_t* __restrict__ inv_k,
                                                               const int32_t* __restrict__ big_list,
                                                               const int32_t* __restrict__ big_count, long long edges, int bx,
                                                               int gx) {
    __shared__ uint32_t s[SORT_LDS];
    const int nbig = *big_count;
    for (int w = bx; w < nbig; w += gx) {
        const int t = big_list[w];
        const int beg = inv_idx[t], d = inv_idx[t + 1] - beg;
        if (beg < 0 || d <= 0 || (long long)beg + d > edges) continue;      // uniform per workgroup; corrupted table only
        if (d <= SORT_LDS) {
            int n2 = 1;
            while (n2 < d) n2 <<= 1;
            __syncthreads();
            for (int i = threadIdx.x; i < n2; i += BLOCK) s[i] = i < d ? keys[beg + i] : 0xffffffffu;
            __syncthreads();
            for (int k = 2; k <= n2; k <<= 1) {
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int i = threadIdx.x; i < n2; i += BLOCK) {
                        const int p = i ^ j;
                        if (p > i) {
                            const uint32_t a = s[i], b = s[p];
                            const bool up = (i & k) == 0;
                            if ((a > b) == up) { s[i] = b; s[p] = a; }
                        }
                    }
                    __syncthreads();
                }
            }
            for (int i = threadIdx.x; i < d; i += BLOCK) emit(s[i], beg + i, K, inv_n, inv_k);
        } else {
            for (int i = threadIdx.x; i < d; i += BLOCK) {
                const uint32_t mine = keys[beg + i];
                int rank = 0;
                for (int j = 0; j < d; ++j) rank += keys[beg + j] < mine;
                emit(mine, beg + rank, K, inv_n, inv_k);
            }
        }
    }
}
__global__ __launch_bounds__(BLOCK) void csr_sort_large_kernel(const uint32_t* __restrict__ keys,
                                                               const int32_t* __restrict__ inv_idx, int K,
                                                               int32_t* __restrict__ inv_n, uint8_t* __restrict__ inv_k,
                                                               const int32_t* __restrict__ big_list,
                                                               const int32_t* __restrict__ big_count, long long edges) {
    csr_sort_large_body(keys, inv_idx, K, inv_n, inv_k, big_list, big_count, edges, blockIdx.x, gridDim.x);
}

struct CsrWs {
    size_t off_counts, off_chunks, off_keys, off_big, off_bigcount, bytes;
    int nchunks;
};
static CsrWs csr_plan(int Nq, int K, int total_points) {
    CsrWs w{};
    w.nchunks = std::max(1, ceil_div(total_points, SCAN_CHUNK));
    size_t off = 0;
    w.off_counts = off;   off = align_up(off + (size_t)(std::max(total_points, 1) + 1) * 4, 256);   // + the big-bucket count
    w.off_chunks = off;   off = align_up(off + (size_t)w.nchunks * 4, 256);
    w.off_keys = off;     off = align_up(off + (size_t)Nq * K * 4 + 4, 256);
    w.off_big = off;      off = align_up(off + (size_t)std::max(total_points, 1) * 4, 256);
    w.off_bigcount = w.off_counts + (size_t)std::max(total_points, 1) * 4;      // cleared with the counts in one memset
    w.bytes = off;
    return w;
}


// ---- all edge sets of an iteration in one pass of launches ----------------------------------------------------
// A training iteration transposes 3 x levels neighbour tables (util/common_util.py:281-309: self, forward and
// propagate edges of every level): 13 calls x 8 launches for the 10cm-lite model, most of them on tables too small to
// fill the chip.  Here blockIdx.y picks the table and every phase runs once for all of them.
constexpr int CSR_BATCH_MAX = 16;
struct CsrProblem {
    const int64_t* idx;
    int32_t* nb;
    uint8_t* kb;
    int32_t* xb;
    int32_t *counts, *chunks, *big, *bigc;
    uint32_t* keys;
    long long edges;
    int total_points, K, nchunks;
};
struct CsrBatch { CsrProblem p[CSR_BATCH_MAX]; };

__global__ __launch_bounds__(BLOCK) void csr_count_batch_kernel(const CsrBatch b) {
    const CsrProblem& q = b.p[blockIdx.y];
    csr_count_body(q.idx, q.counts, q.edges, q.total_points, blockIdx.x, gridDim.x);
}
__global__ __launch_bounds__(BLOCK) void scan_chunk_sums_batch_kernel(const CsrBatch b) {
    const CsrProblem& q = b.p[blockIdx.y];
    if ((int)blockIdx.x < q.nchunks) scan_chunk_sums_body(q.counts, q.chunks, q.total_points, blockIdx.x, gridDim.x);
}
__global__ __launch_bounds__(BLOCK) void scan_chunk_offsets_batch_kernel(const CsrBatch b) {
    const CsrProblem& q = b.p[blockIdx.y];
    scan_chunk_offsets_body(q.chunks, q.nchunks, 0, 1);
}
__global__ __launch_bounds__(BLOCK) void scan_write_batch_kernel(const CsrBatch b) {
    const CsrProblem& q = b.p[blockIdx.y];
    if ((int)blockIdx.x < q.nchunks) scan_write_body(q.counts, q.chunks, q.xb, q.total_points, true, blockIdx.x, gridDim.x);
}
__global__ __launch_bounds__(BLOCK) void csr_fill_batch_kernel(const CsrBatch b) {
    const CsrProblem& q = b.p[blockIdx.y];
    csr_fill_body(q.idx, q.xb, q.counts, q.keys, q.edges, q.total_points, blockIdx.x, gridDim.x);
}
__global__ __launch_bounds__(BLOCK) void csr_sort_small_batch_kernel(const CsrBatch b) {
    const CsrProblem& q = b.p[blockIdx.y];
    csr_sort_small_body(q.keys, q.xb, q.total_points, q.K, q.nb, q.kb, q.big, q.bigc, q.edges, blockIdx.x, gridDim.x);
}
__global__ __launch_bounds__(BLOCK) void csr_sort_large_batch_kernel(const CsrBatch b) {
    const CsrProblem& q = b.p[blockIdx.y];
    csr_sort_large_body(q.keys, q.xb, q.K, q.nb, q.kb, q.big, q.bigc, q.edges, blockIdx.x, gridDim.x);
}

// workspace of a batch: [counts + big-bucket count of every table (zeroed by one memset)] [chunk sums, keys, big lists]
struct CsrBatchWs { size_t zero_bytes, bytes; };
static CsrBatchWs csr_batch_plan(int n, const int* Nq, const int* K, const int* total_points, size_t* off_counts,
                                 size_t* off_chunks, size_t* off_keys, size_t* off_big) {
    size_t off = 0;
    for (int i = 0; i < n; ++i) {
        if (off_counts) off_counts[i] = off;
        off = align_up(off + ((size_t)std::max(total_points[i], 1) + 1) * 4, 256);
    }
    CsrBatchWs w{off, 0};
    for (int i = 0; i < n; ++i) {
        const int nchunks = std::max(1, ceil_div(total_points[i], SCAN_CHUNK));
        if (off_chunks) off_chunks[i] = off;
        off = align_up(off + (size_t)nchunks * 4, 256);
        if (off_keys) off_keys[i] = off;
        off = align_up(off + (size_t)Nq[i] * K[i] * 4 + 4, 256);
        if (off_big) off_big[i] = off;
        off = align_up(off + (size_t)std::max(total_points[i], 1) * 4, 256);
    }
    w.bytes = off;
    return w;
}

}  // namespace pcf

extern "C" {

int pcf_hip_knn(const float* ref, const float* query, const int32_t* ref_off, const int32_t* query_off, int n_seg,
                int max_queries_per_seg, int K, int64_t* out, void* stream) {
    using namespace pcf;
    PCF_REQUIRE(K >= 1 && K <= 64, "knn: K must be in [1,64] (got %d)", K);
    PCF_REQUIRE(n_seg >= 0 && max_queries_per_seg >= 0, "knn: negative size");
    if (n_seg == 0 || max_queries_per_seg == 0) return ok();
    PCF_REQUIRE(ref && query && ref_off && query_off && out, "knn: null pointer");
    PCF_REQUIRE(n_seg <= 65535, "knn: more than 65535 samples in one packed batch");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(ceil_div(max_queries_per_seg, BLOCK), n_seg);
    if (K <= 8) hipLaunchKernelGGL(knn_kernel<8>, grid, dim3(BLOCK), 0, s, ref, query, ref_off, query_off, K, out);
    else if (K <= 16) hipLaunchKernelGGL(knn_kernel<16>, grid, dim3(BLOCK), 0, s, ref, query, ref_off, query_off, K, out);
    else if (K <= 32) hipLaunchKernelGGL(knn_kernel<32>, grid, dim3(BLOCK), 0, s, ref, query, ref_off, query_off, K, out);
    else hipLaunchKernelGGL(knn_kernel<64>, grid, dim3(BLOCK), 0, s, ref, query, ref_off, query_off, K, out);
    return check_launch("kNN");
}

int pcf_hip_knn_wave(const float* ref, const float* query, const int32_t* ref_off, const int32_t* query_off, int n_seg,
                     int n_query, int K, int64_t* out, void* stream) {
    using namespace pcf;
    PCF_REQUIRE(K >= 1 && K <= 64, "knn_wave: K must be in [1,64] (got %d)", K);
    PCF_REQUIRE(n_seg >= 0 && n_query >= 0, "knn_wave: negative size");
    if (n_seg == 0 || n_query == 0) return ok();
    PCF_REQUIRE(ref_off && query_off && query && out, "knn_wave: null pointer");
    const int grid = std::min(ceil_div(n_query, NWAVE), 256 * 64);
    hipLaunchKernelGGL(knn_wave_kernel, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, ref, query, ref_off, query_off, n_seg,
                       n_query, K, out);
    return check_launch("kNN (wave per query)");
}

size_t pcf_hip_knn_inverse_workspace_bytes(int B, int Nq, int K, int total_points) {
    (void)B;
    if (Nq < 0 || K < 0 || total_points < 0) return 0;
    return pcf::csr_plan(Nq, K, total_points).bytes;
}

int pcf_hip_knn_inverse(const int64_t* idx, int32_t* inv_neighbors, uint8_t* inv_k, int32_t* inv_idx, void* workspace,
                        size_t workspace_bytes, int B, int Nq, int K, int total_points, void* stream) {
    using namespace pcf;
    PCF_REQUIRE(B >= 0 && Nq >= 0 && total_points >= 0, "knn_inverse: negative size");
    PCF_REQUIRE(K >= 1 && K <= 255, "knn_inverse: K must be in [1,255] (inv_k is uint8), got %d", K);
    PCF_REQUIRE((long long)Nq * K < (1ll << 31), "knn_inverse: Nq*K does not fit 31 bits");
    const CsrWs w = csr_plan(Nq, K, total_points);
    PCF_REQUIRE(workspace_bytes >= w.bytes && workspace && aligned16(workspace),
                "knn_inverse: workspace too small or misaligned (%zu < %zu)", workspace_bytes, w.bytes);
    PCF_REQUIRE(inv_idx && (B * (long long)Nq == 0 || (idx && inv_neighbors && inv_k)), "knn_inverse: null pointer");
    hipStream_t s = (hipStream_t)stream;
    char* ws = static_cast<char*>(workspace);
    int32_t* counts = reinterpret_cast<int32_t*>(ws + w.off_counts);
    int32_t* chunks = reinterpret_cast<int32_t*>(ws + w.off_chunks);
    uint32_t* keys = reinterpret_cast<uint32_t*>(ws + w.off_keys);
    int32_t* big = reinterpret_cast<int32_t*>(ws + w.off_big);
    int32_t* bigc = reinterpret_cast<int32_t*>(ws + w.off_bigcount);
    const long long edges = (long long)Nq * K;
    const int egrid = (int)std::max<long long>(1, std::min<long long>((edges + BLOCK - 1) / BLOCK, 4096));
#define PCF_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) return fail(PCF_E_LAUNCH, "knn_inverse: %s", hipGetErrorString(e_)); \
    } while (0)
    for (int b = 0; b < B; ++b) {
        const int64_t* ib = idx + (size_t)b * edges;
        int32_t* nb = inv_neighbors + (size_t)b * edges;
        uint8_t* kb = inv_k + (size_t)b * edges;
        int32_t* xb = inv_idx + (size_t)b * (total_points + 1);
        if (total_points == 0) {
            if (edges) {
                PCF_HIP(zero_async(nb, (size_t)edges * 4, s));
                PCF_HIP(zero_async(kb, (size_t)edges, s));
            }
            PCF_HIP(zero_async(xb, 4, s));
            continue;
        }
        PCF_HIP(zero_async(counts, (size_t)(total_points + 1) * 4, s));       // histogram + big-bucket count
        if (edges) {
            hipLaunchKernelGGL(csr_count_kernel, dim3(egrid), dim3(BLOCK), 0, s, ib, counts, edges, total_points);
            if (int e = check_launch("knn_inverse histogram")) return e;
        }
        hipLaunchKernelGGL(scan_chunk_sums_kernel, dim3(w.nchunks), dim3(BLOCK), 0, s, counts, chunks, total_points);
        hipLaunchKernelGGL(scan_chunk_offsets_kernel, dim3(1), dim3(BLOCK), 0, s, chunks, w.nchunks);
        hipLaunchKernelGGL(scan_write_kernel, dim3(w.nchunks), dim3(BLOCK), 0, s, counts, chunks, xb, total_points, true);
        if (int e = check_launch("knn_inverse scan")) return e;
        if (edges) {
            hipLaunchKernelGGL(csr_fill_kernel, dim3(egrid), dim3(BLOCK), 0, s, ib, xb, counts, keys, edges, total_points);
            const int sgrid = std::max(1, std::min(ceil_div(total_points, NWAVE), 8192));
            hipLaunchKernelGGL(csr_sort_small_kernel, dim3(sgrid), dim3(BLOCK), 0, s, keys, xb, total_points, K, nb, kb,
                               big, bigc, edges);
            hipLaunchKernelGGL(csr_sort_large_kernel, dim3(1024), dim3(BLOCK), 0, s, keys, xb, K, nb, kb, big, bigc, edges);
            if (int e = check_launch("knn_inverse fill/sort")) return e;
        }
    }
#undef PCF_HIP
    return ok();
}


size_t pcf_hip_knn_inverse_batched_workspace_bytes(int n_tables, const int* Nq, const int* K, const int* total_points) {
    if (n_tables < 0 || (n_tables > 0 && (!Nq || !K || !total_points))) return 0;
    for (int i = 0; i < n_tables; ++i)
        if (Nq[i] < 0 || K[i] < 1 || total_points[i] < 0) return 0;
    return pcf::csr_batch_plan(n_tables, Nq, K, total_points, nullptr, nullptr, nullptr, nullptr).bytes;
}

int pcf_hip_knn_inverse_batched(int n_tables, const int64_t* const* idx, int32_t* const* inv_neighbors,
                                uint8_t* const* inv_k, int32_t* const* inv_idx, const int* Nq, const int* K,
                                const int* total_points, void* workspace, size_t workspace_bytes, void* stream) {
    using namespace pcf;
    PCF_REQUIRE(n_tables >= 0 && n_tables <= 4096, "knn_inverse_batched: bad table count %d", n_tables);
    if (n_tables == 0) return ok();
    PCF_REQUIRE(idx && inv_neighbors && inv_k && inv_idx && Nq && K && total_points, "knn_inverse_batched: null table list");
    for (int i = 0; i < n_tables; ++i) {
        PCF_REQUIRE(Nq[i] >= 1 && total_points[i] >= 1, "knn_inverse_batched: table %d is empty (Nq=%d, total_points=%d): "
                    "use pcf_hip_knn_inverse for it", i, Nq[i], total_points[i]);
        PCF_REQUIRE(K[i] >= 1 && K[i] <= 255, "knn_inverse_batched: K must be in [1,255] (inv_k is uint8), got %d", K[i]);
        PCF_REQUIRE((long long)Nq[i] * K[i] < (1ll << 31), "knn_inverse_batched: Nq*K does not fit 31 bits");
        PCF_REQUIRE(idx[i] && inv_neighbors[i] && inv_k[i] && inv_idx[i], "knn_inverse_batched: null pointer in table %d", i);
    }
    std::vector<size_t> oc(n_tables), och(n_tables), ok_(n_tables), ob(n_tables);
    const CsrBatchWs w = csr_batch_plan(n_tables, Nq, K, total_points, oc.data(), och.data(), ok_.data(), ob.data());
    PCF_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= w.bytes,
                "knn_inverse_batched: workspace too small or misaligned (%zu < %zu)", workspace_bytes, w.bytes);
    hipStream_t s = (hipStream_t)stream;
    char* ws = static_cast<char*>(workspace);
    if (zero_async(ws, w.zero_bytes, s) != hipSuccess) return fail(PCF_E_LAUNCH, "knn_inverse_batched: memset failed");
    for (int base = 0; base < n_tables; base += CSR_BATCH_MAX) {
        const int nb = std::min(CSR_BATCH_MAX, n_tables - base);
        CsrBatch b{};
        long long max_edges = 0;
        int max_chunks = 1, max_points = 1;
        for (int j = 0; j < nb; ++j) {
            const int i = base + j;
            CsrProblem& q = b.p[j];
            q.idx = idx[i]; q.nb = inv_neighbors[i]; q.kb = inv_k[i]; q.xb = inv_idx[i];
            q.counts = reinterpret_cast<int32_t*>(ws + oc[i]);
            q.bigc = q.counts + total_points[i];
            q.chunks = reinterpret_cast<int32_t*>(ws + och[i]);
            q.keys = reinterpret_cast<uint32_t*>(ws + ok_[i]);
            q.big = reinterpret_cast<int32_t*>(ws + ob[i]);
            q.edges = (long long)Nq[i] * K[i];
            q.total_points = total_points[i]; q.K = K[i];
            q.nchunks = std::max(1, ceil_div(total_points[i], SCAN_CHUNK));
            max_edges = std::max(max_edges, q.edges);
            max_chunks = std::max(max_chunks, q.nchunks);
            max_points = std::max(max_points, total_points[i]);
        }
        const int egrid = (int)std::max<long long>(1, std::min<long long>((max_edges + BLOCK - 1) / BLOCK, 4096));
        const int sgrid = std::max(1, std::min(ceil_div(max_points, NWAVE), 8192));
        hipLaunchKernelGGL(csr_count_batch_kernel, dim3(egrid, nb), dim3(BLOCK), 0, s, b);
        hipLaunchKernelGGL(scan_chunk_sums_batch_kernel, dim3(max_chunks, nb), dim3(BLOCK), 0, s, b);
        hipLaunchKernelGGL(scan_chunk_offsets_batch_kernel, dim3(1, nb), dim3(BLOCK), 0, s, b);
        hipLaunchKernelGGL(scan_write_batch_kernel, dim3(max_chunks, nb), dim3(BLOCK), 0, s, b);
        hipLaunchKernelGGL(csr_fill_batch_kernel, dim3(egrid, nb), dim3(BLOCK), 0, s, b);
        hipLaunchKernelGGL(csr_sort_small_batch_kernel, dim3(sgrid, nb), dim3(BLOCK), 0, s, b);
        hipLaunchKernelGGL(csr_sort_large_batch_kernel, dim3(256, nb), dim3(BLOCK), 0, s, b);
        if (int e = check_launch("knn_inverse_batched")) return e;
    }
    return ok();
}

}  // extern "C"
