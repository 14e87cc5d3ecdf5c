// Barycentre grid subsampling of a packed batch of point clouds on the GPU: the multi-resolution levels the
// reference builds in its dataloader workers with cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp
// (called per sample and level from datasetCommon.py:384-421 / :462), moved next to the kNN so the whole
// post-dataloader path is device-side.
//
//   per sample s:  origin = floor(min_corner * (1/dl)) * dl,  (i,j,k) = floor((p - origin) / dl)       (:29-31, :58-60)
//   per occupied voxel: barycentre of its points and the mean of their features (the normals)           (:64-72, :91-98)
//
// The reference walks the points once and accumulates into a hash map; the sums are float and sequential in point
// order.  Here: 64-bit keys (sample | k | j | i) -> stable radix sort of (key, point index) -> run heads -> one lane
// per (voxel, channel) adds the run's values IN POINT ORDER, so barycentres and features are bit-identical to the
// reference's; voxels come out sorted by (sample, k, j, i) -- the reference's own linear index order -- instead of
// std::unordered_map order.  Label voting (grid_subsampling.h:45-57) is not on the training path and stays on the CPU.
// HBM-bound integer / byte work: ~25 B/point read, 12 B/point of sort pairs x 8 radix passes, one gather of the
// point and feature rows.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "pcf_common.h"

namespace pcf {

int exclusive_scan_i32(int32_t* counts, int32_t* chunk_tmp, int32_t* out, int n, bool clear_counts, hipStream_t s);   // knn.hip

constexpr int AXIS_BITS = 18;                 // voxels per axis and sample < 262144 (5 km at 2 cm)
constexpr int SEG_BITS = 64 - 3 * AXIS_BITS;  // samples per packed batch < 1024

struct SegBox {
    float ox, oy, oz;      // origin corner (:31)
    int ok;
};

__device__ __forceinline__ int seg_of_point(const int32_t* off, int n_seg, int i) {
    int lo = 0, hi = n_seg;           // off[lo] <= i < off[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (off[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

// ---- 1. per-sample bounding box -> origin corner; one workgroup per sample -----------------------------------
__global__ __launch_bounds__(BLOCK) void seg_box_kernel(const float* __restrict__ pts, const int32_t* __restrict__ seg_off,
                                                        float dl, SegBox* __restrict__ boxes, int32_t* __restrict__ status) {
    __shared__ float red[6][NWAVE];
    const int s = blockIdx.x;
    const int beg = seg_off[s], end = seg_off[s + 1];
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = beg + threadIdx.x; i < end; i += BLOCK) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = pts[3 * (size_t)i + a];
            mn[a] = fminf(mn[a], v);
            mx[a] = fmaxf(mx[a], v);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        for (int off = WAVE / 2; off > 0; off >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], off, WAVE));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off, WAVE));
        }
        if (lane_id() == 0) { red[a][wave_id()] = mn[a]; red[3 + a][wave_id()] = mx[a]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        SegBox b{0.f, 0.f, 0.f, 1};
        if (end > beg) {
            float o[3];
            const float inv = 1.f / dl;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                float lo = red[a][0], hi = red[3 + a][0];
                for (int w = 1; w < NWAVE; ++w) { lo = fminf(lo, red[a][w]); hi = fmaxf(hi, red[3 + a][w]); }
                o[a] = floorf(lo * inv) * dl;
                const float cells = floorf((hi - o[a]) / dl);
                if (!(cells >= 0.f && cells < (float)(1 << AXIS_BITS))) b.ok = 0;      // also catches NaN / inf
            }
            b.ox = o[0]; b.oy = o[1]; b.oz = o[2];
            if (!b.ok) atomicOr(status, 1);
        }
        boxes[s] = b;
    }
}

// ---- 2. voxel key of every point ------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void voxel_key_kernel(const float* __restrict__ pts, const int32_t* __restrict__ seg_off,
                                                          int n_seg, int n, float dl, const SegBox* __restrict__ boxes,
                                                          unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals) {
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const int s = seg_of_point(seg_off, n_seg, i);
        const SegBox b = boxes[s];
        unsigned long long key = (unsigned long long)s << (3 * AXIS_BITS);
        if (b.ok) {
            const unsigned long long ix = (unsigned long long)floorf((pts[3 * (size_t)i] - b.ox) / dl);
            const unsigned long long iy = (unsigned long long)floorf((pts[3 * (size_t)i + 1] - b.oy) / dl);
            const unsigned long long iz = (unsigned long long)floorf((pts[3 * (size_t)i + 2] - b.oz) / dl);
            constexpr unsigned long long M = (1ull << AXIS_BITS) - 1;      // the box test in seg_box_kernel keeps them below
            key |= ((iz & M) << (2 * AXIS_BITS)) | ((iy & M) << AXIS_BITS) | (ix & M);
        }
        keys[i] = key;
        vals[i] = (uint32_t)i;
    }
}

// ---- 3. run heads of the sorted keys ----------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void run_head_kernel(const unsigned long long* __restrict__ keys, int n,
                                                         int32_t* __restrict__ flags) {
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK)
        flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}

// vid[i] = heads before sorted position i (exclusive scan of the flags); vid[n] = number of voxels
__global__ __launch_bounds__(BLOCK) void run_start_kernel(const unsigned long long* __restrict__ keys,
                                                          const int32_t* __restrict__ vid, int n,
                                                          int32_t* __restrict__ start, int32_t* __restrict__ seg_counts,
                                                          int32_t* __restrict__ total) {
    const int lane = lane_id();
    for (int i0 = blockIdx.x * BLOCK; i0 < n; i0 += gridDim.x * BLOCK) {       // whole waves stay in the loop (ballots)
        const int i = i0 + threadIdx.x;
        bool head = false;
        int seg = -1;
        if (i < n) {
            head = (i == 0) || keys[i] != keys[i - 1];
            seg = (int)(keys[i] >> (3 * AXIS_BITS));
            if (head) start[vid[i]] = i;
            if (i == n - 1) {
                start[vid[n]] = n;
                *total = vid[n];
            }
        }
        // voxels per sample: one atomic per (wave, sample) -- 50k heads on four counters serialise otherwise (205 us)
        unsigned long long todo = __ballot(head);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int s0 = __shfl(seg, leader, WAVE);
            const unsigned long long m = __ballot(head && seg == s0);
            if (lane == leader) atomicAdd(&seg_counts[s0], (int)__popcll(m));
            todo &= ~m;
        }
    }
}

// ---- 4. one lane per (voxel, channel): sequential float sums in point order -------------------------------------
__global__ __launch_bounds__(BLOCK) void voxel_mean_kernel(const float* __restrict__ pts, const float* __restrict__ feats,
                                                           int F, const uint32_t* __restrict__ order,
                                                           const int32_t* __restrict__ start,
                                                           const int32_t* __restrict__ total, long long slots,
                                                           float* __restrict__ out_pts, float* __restrict__ out_feats) {
    const int C = 3 + F;
    const long long work = (long long)(*total) * C;
    for (long long t = (long long)blockIdx.x * BLOCK + threadIdx.x; t < work && t < slots; t += (long long)gridDim.x * BLOCK) {
        const int v = (int)(t / C);
        const int c = (int)(t - (long long)v * C);
        const int beg = start[v], end = start[v + 1];
        const float* src = c < 3 ? pts + c : feats + (c - 3);
        const int ld = c < 3 ? 3 : F;
        float acc = 0.f;
        for (int j = beg; j < end; ++j) acc += src[(size_t)order[j] * ld];
        const int count = end - beg;
        if (c < 3) out_pts[(size_t)v * 3 + c] = acc * (float)(1.0 / (double)count);       // :91
        else out_feats[(size_t)v * F + (c - 3)] = acc / (float)count;                       // :94-98
    }
}

// ---- level-0 voxelisation: one point per occupied voxel (util/voxelize.py:44-82) ----------------------------------
// key = FNV64-1A over floor(coord / voxel_size) taken as uint64 per axis (:10-22, :58-62).  The division runs in double:
// numpy promotes the float32 coordinates against the 0-d float64 array np.array(voxel_size) (:58) -- under NumPy >= 2
// (NEP 50), which is what the fixtures were generated with (2.2.6).  Under NumPy 1.x value-based casting keeps a float32
// array divided by a 0-d float64 array in float32, so points within one float32 rounding of a voxel face can floor
// differently there: parity for float32 input is pinned for NumPy >= 2 semantics only (the reference's data loader hands
// float64 coordinates, which divide in double under both).  'deterministic' mode returns the LOWEST point index of a voxel
// where the reference's unstable argsort returns an arbitrary member.
// T = float (promoted to double before the division, above) or double (coordinates handed over in float64: the quotient
// numpy computes under any version).
template <typename T>
__global__ __launch_bounds__(BLOCK) void fnv_key_kernel(const T* __restrict__ pts, int n, double voxel,
                                                        unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals) {
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        unsigned long long h = 14695981039346656037ull;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double d = floor((double)pts[3 * (size_t)i + a] / voxel);
            h *= 1099511628211ull;
            h ^= (unsigned long long)(long long)d;          // astype(uint64) of a (possibly negative) whole number
        }
        keys[i] = h;
        vals[i] = (uint32_t)i;
    }
}
// run starts of the sorted keys: start[v] = first sorted position of voxel v, start[total] = n; total, longest run
__global__ __launch_bounds__(BLOCK) void vox_start_kernel(const unsigned long long* __restrict__ keys, const int32_t* __restrict__ vid,
                                                          int n, int32_t* __restrict__ start, int32_t* __restrict__ total) {
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        if (i == 0 || keys[i] != keys[i - 1]) start[vid[i]] = i;
        if (i == n - 1) { start[vid[n]] = n; total[0] = vid[n]; }
    }
}
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
// mode 0: first point of the voxel (lowest index: the sort is stable); 1: a pseudo-random one (seed); 2: point `rank % count`
__global__ __launch_bounds__(BLOCK) void vox_pick_kernel(const uint32_t* __restrict__ order, const int32_t* __restrict__ start,
                                                         int32_t* __restrict__ total, int n, int mode, unsigned long long seed,
                                                         int rank, int64_t* __restrict__ out) {
    const int nv = total[0];
    int longest = 0;
    for (int v = blockIdx.x * BLOCK + threadIdx.x; v < nv && v < n; v += gridDim.x * BLOCK) {
        const int beg = start[v], cnt = start[v + 1] - beg;
        int off = 0;
        if (mode == 1) off = (int)(splitmix64(seed ^ (unsigned long long)v * 0x9E3779B97F4A7C15ull) % (unsigned long long)cnt);
        else if (mode == 2) off = rank % cnt;
        out[v] = (int64_t)order[beg + off];
        longest = max(longest, cnt);
    }
    for (int off = WAVE / 2; off > 0; off >>= 1) longest = max(longest, __shfl_xor(longest, off, WAVE));
    if (lane_id() == 0 && longest > 0) atomicMax(&total[1], longest);
}

struct SubWs {
    size_t off_boxes, off_keys_a, off_keys_b, off_vals_a, off_vals_b, off_flags, off_vid, off_chunks, off_start, off_sort, sort_bytes,
        bytes;
};

static SubWs sub_plan(int n, int n_seg) {
    SubWs w{};
    const size_t np = (size_t)std::max(n, 1);
    size_t off = 0;
    w.off_boxes = off;  off = align_up(off + (size_t)std::max(n_seg, 1) * sizeof(SegBox), 256);
    w.off_keys_a = off; off = align_up(off + np * 8, 256);
    w.off_keys_b = off; off = align_up(off + np * 8, 256);
    w.off_vals_a = off; off = align_up(off + np * 4, 256);
    w.off_vals_b = off; off = align_up(off + np * 4, 256);
    w.off_flags = off;  off = align_up(off + (np + 1) * 4, 256);
    w.off_vid = off;    off = align_up(off + (np + 2) * 4, 256);
    w.off_chunks = off; off = align_up(off + (np / 1024 + 2) * 4, 256);
    w.off_start = off;  off = align_up(off + (np + 2) * 4, 256);
    size_t sb = 0;
    (void)rocprim::radix_sort_pairs(nullptr, sb, (const unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                    (const uint32_t*)nullptr, (uint32_t*)nullptr, np, 0, 64, (hipStream_t) nullptr);
    w.sort_bytes = sb;
    w.off_sort = off;   off = align_up(off + sb, 256);
    w.bytes = off;
    return w;
}

}  // namespace pcf

extern "C" {

size_t pcf_hip_grid_subsample_workspace_bytes(int n_points, int n_seg) {
    if (n_points < 0 || n_seg < 0) return 0;
    return pcf::sub_plan(n_points, n_seg).bytes;
}

int pcf_hip_grid_subsample(const float* points, const float* features, const int32_t* seg_off, int n_seg, int n_points,
                           int F, float sampleDl, float* out_points, float* out_features, int32_t* out_seg_counts,
                           int32_t* out_total, void* workspace, size_t workspace_bytes, void* stream) {
    using namespace pcf;
    PCF_REQUIRE(n_seg >= 0 && n_points >= 0 && F >= 0, "grid_subsample: negative size");
    PCF_REQUIRE(n_seg < (1 << SEG_BITS), "grid_subsample: more than %d samples in one packed batch", (1 << SEG_BITS) - 1);
    PCF_REQUIRE(sampleDl > 0.f, "grid_subsample: sampleDl must be positive (got %g)", (double)sampleDl);
    PCF_REQUIRE(out_total && (n_seg == 0 || (out_seg_counts && seg_off)), "grid_subsample: null count output / offsets");
    hipStream_t s = (hipStream_t)stream;
#define PCF_HIP(call)                                                                                \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) return fail(PCF_E_LAUNCH, "grid_subsample: %s", hipGetErrorString(e_)); \
    } while (0)
    PCF_HIP(zero_async(out_total, 2 * sizeof(int32_t), s));          // [0] voxels, [1] status bits
    if (n_seg) PCF_HIP(zero_async(out_seg_counts, (size_t)n_seg * 4, s));
    if (n_points == 0 || n_seg == 0) return ok();
    PCF_REQUIRE(points && out_points && (F == 0 || (features && out_features)), "grid_subsample: null pointer");
    const SubWs w = sub_plan(n_points, n_seg);
    PCF_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= w.bytes, "grid_subsample: workspace too small or misaligned");
    char* ws = static_cast<char*>(workspace);
    SegBox* boxes = reinterpret_cast<SegBox*>(ws + w.off_boxes);
    auto* keys_a = reinterpret_cast<unsigned long long*>(ws + w.off_keys_a);
    auto* keys_b = reinterpret_cast<unsigned long long*>(ws + w.off_keys_b);
    auto* vals_a = reinterpret_cast<uint32_t*>(ws + w.off_vals_a);
    auto* vals_b = reinterpret_cast<uint32_t*>(ws + w.off_vals_b);
    int32_t* flags = reinterpret_cast<int32_t*>(ws + w.off_flags);
    int32_t* vid = reinterpret_cast<int32_t*>(ws + w.off_vid);
    int32_t* chunks = reinterpret_cast<int32_t*>(ws + w.off_chunks);
    int32_t* start = reinterpret_cast<int32_t*>(ws + w.off_start);
    const int n = n_points;
    const int pgrid = std::max(1, std::min(ceil_div(n, BLOCK), 4096));
    hipLaunchKernelGGL(seg_box_kernel, dim3(n_seg), dim3(BLOCK), 0, s, points, seg_off, sampleDl, boxes, out_total + 1);
    hipLaunchKernelGGL(voxel_key_kernel, dim3(pgrid), dim3(BLOCK), 0, s, points, seg_off, n_seg, n, sampleDl, boxes, keys_a, vals_a);
    if (int e = check_launch("grid_subsample: keys")) return e;
    size_t sb = w.sort_bytes;
    PCF_HIP(rocprim::radix_sort_pairs(ws + w.off_sort, sb, (const unsigned long long*)keys_a, keys_b, (const uint32_t*)vals_a, vals_b,
                                      (size_t)n, 0, 64, s));
    hipLaunchKernelGGL(run_head_kernel, dim3(pgrid), dim3(BLOCK), 0, s, keys_b, n, flags);
    if (int e = exclusive_scan_i32(flags, chunks, vid, n, false, s)) return e;
    hipLaunchKernelGGL(run_start_kernel, dim3(pgrid), dim3(BLOCK), 0, s, keys_b, vid, n, start, out_seg_counts, out_total);
    const long long slots = (long long)n * (3 + F);
    const int mgrid = (int)std::max<long long>(1, std::min<long long>((slots + BLOCK - 1) / BLOCK, 8192));
    hipLaunchKernelGGL(voxel_mean_kernel, dim3(mgrid), dim3(BLOCK), 0, s, points, features, F, vals_b, start, out_total, slots,
                       out_points, out_features);
#undef PCF_HIP
    return check_launch("grid_subsample: means");
}

size_t pcf_hip_voxelize_workspace_bytes(int n_points) {
    if (n_points < 0) return 0;
    return pcf::sub_plan(n_points, 1).bytes;
}

static int voxelize_impl(const void* points, bool f64, int n_points, double voxel_size, int mode, unsigned long long seed, int rank,
                         int64_t* out_index, int32_t* out_total, void* workspace, size_t workspace_bytes, void* stream) {
    using namespace pcf;
    PCF_REQUIRE(n_points >= 0 && voxel_size > 0.0 && mode >= 0 && mode <= 2 && rank >= 0, "voxelize: bad arguments (n=%d voxel=%g mode=%d)",
                n_points, voxel_size, mode);
    PCF_REQUIRE(out_total, "voxelize: null count output");
    hipStream_t s = (hipStream_t)stream;
    if (zero_async(out_total, 2 * sizeof(int32_t), s) != hipSuccess) return fail(PCF_E_LAUNCH, "voxelize: memset");
    if (n_points == 0) return ok();
    const SubWs w = sub_plan(n_points, 1);
    PCF_REQUIRE(points && out_index && workspace && aligned16(workspace) && workspace_bytes >= w.bytes,
                "voxelize: null pointer or small / misaligned workspace");
    char* ws = static_cast<char*>(workspace);
    auto* keys_a = reinterpret_cast<unsigned long long*>(ws + w.off_keys_a);
    auto* keys_b = reinterpret_cast<unsigned long long*>(ws + w.off_keys_b);
    auto* vals_a = reinterpret_cast<uint32_t*>(ws + w.off_vals_a);
    auto* vals_b = reinterpret_cast<uint32_t*>(ws + w.off_vals_b);
    int32_t* flags = reinterpret_cast<int32_t*>(ws + w.off_flags);
    int32_t* vid = reinterpret_cast<int32_t*>(ws + w.off_vid);
    int32_t* chunks = reinterpret_cast<int32_t*>(ws + w.off_chunks);
    int32_t* start = reinterpret_cast<int32_t*>(ws + w.off_start);
    const int n = n_points;
    const int pgrid = std::max(1, std::min(ceil_div(n, BLOCK), 4096));
    if (f64) hipLaunchKernelGGL(fnv_key_kernel<double>, dim3(pgrid), dim3(BLOCK), 0, s, static_cast<const double*>(points), n, voxel_size, keys_a, vals_a);
    else hipLaunchKernelGGL(fnv_key_kernel<float>, dim3(pgrid), dim3(BLOCK), 0, s, static_cast<const float*>(points), n, voxel_size, keys_a, vals_a);
    if (int e = check_launch("voxelize: keys")) return e;
    size_t sb = w.sort_bytes;
    if (rocprim::radix_sort_pairs(ws + w.off_sort, sb, (const unsigned long long*)keys_a, keys_b, (const uint32_t*)vals_a, vals_b,
                                  (size_t)n, 0, 64, s) != hipSuccess)
        return fail(PCF_E_LAUNCH, "voxelize: radix sort");
    hipLaunchKernelGGL(run_head_kernel, dim3(pgrid), dim3(BLOCK), 0, s, keys_b, n, flags);
    if (int e = exclusive_scan_i32(flags, chunks, vid, n, false, s)) return e;
    hipLaunchKernelGGL(vox_start_kernel, dim3(pgrid), dim3(BLOCK), 0, s, keys_b, vid, n, start, out_total);
    hipLaunchKernelGGL(vox_pick_kernel, dim3(pgrid), dim3(BLOCK), 0, s, vals_b, start, out_total, n, mode, seed, rank, out_index);
    return check_launch("voxelize: pick");
}

int pcf_hip_voxelize(const float* points, int n_points, double voxel_size, int mode, unsigned long long seed, int rank,
                     int64_t* out_index, int32_t* out_total, void* workspace, size_t workspace_bytes, void* stream) {
    return voxelize_impl(points, false, n_points, voxel_size, mode, seed, rank, out_index, out_total, workspace, workspace_bytes, stream);
}

int pcf_hip_voxelize_f64(const double* points, int n_points, double voxel_size, int mode, unsigned long long seed, int rank,
                         int64_t* out_index, int32_t* out_total, void* workspace, size_t workspace_bytes, void* stream) {
    return voxelize_impl(points, true, n_points, voxel_size, mode, seed, rank, out_index, out_total, workspace, workspace_bytes, stream);
}

}  // extern "C"
