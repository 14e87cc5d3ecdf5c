/* CPU oracle for the packed kNN and its CSR transpose -- TEST INFRASTRUCTURE ONLY.
 *
 * Same definition as oracle/pcf_oracle.py:knn_bruteforce, in C so that it finishes in seconds at
 * tens of thousands of points: distance = ((rx-qx)^2 + (ry-qy)^2) + (rz-qz)^2 in fp32 with one
 * rounding per operation (build with -ffp-contract=off), K smallest by (distance, index).
 * Follows what knn_post_dataloader_utils.py:22-41 asks of KeOps (argKmin of that expression) and the
 * per-sample loop of :171-223.  Parity: pinned by definition + KDTree agreement on untied queries
 * (tests/test_oracle_golden.py); the reference holds no kNN fixture.
 * Never linked into, or called from, the product.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

void oracle_knn_packed(const float* ref, const float* query, const int32_t* ref_off, const int32_t* query_off,
                       int n_seg, int K, int64_t* out) {
    float* bd = (float*)malloc(sizeof(float) * (size_t)K);
    int64_t* bi = (int64_t*)malloc(sizeof(int64_t) * (size_t)K);
    for (int s = 0; s < n_seg; ++s) {
        for (int q = query_off[s]; q < query_off[s + 1]; ++q) {
            const float qx = query[3 * (size_t)q], qy = query[3 * (size_t)q + 1], qz = query[3 * (size_t)q + 2];
            for (int k = 0; k < K; ++k) { bd[k] = INFINITY; bi[k] = -1; }
            for (int r = ref_off[s]; r < ref_off[s + 1]; ++r) {
                const float dx = ref[3 * (size_t)r] - qx, dy = ref[3 * (size_t)r + 1] - qy,
                            dz = ref[3 * (size_t)r + 2] - qz;
                const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
                const float xy = xx + yy;
                const float d = xy + zz;
                if (!(d < bd[K - 1])) continue;
                int pos = K - 1;                         /* insert after every element <= d */
                while (pos > 0 && d < bd[pos - 1]) { bd[pos] = bd[pos - 1]; bi[pos] = bi[pos - 1]; --pos; }
                bd[pos] = d;
                bi[pos] = r;
            }
            for (int k = 0; k < K; ++k) out[(size_t)q * K + k] = bi[k];
        }
    }
    free(bd);
    free(bi);
}

/* CSR transpose, buckets in (query, k) order.  (knn.cu:104-168; test_kernels.py:177-213) */
void oracle_knn_inverse(const int64_t* idx, int Nq, int K, int total_points, int32_t* inv_neighbors, uint8_t* inv_k,
                        int32_t* inv_idx) {
    const size_t edges = (size_t)Nq * K;
    for (int i = 0; i <= total_points; ++i) inv_idx[i] = 0;
    for (size_t e = 0; e < edges; ++e) {
        inv_neighbors[e] = 0;
        inv_k[e] = 0;
        if (idx[e] >= 0 && idx[e] < total_points) inv_idx[idx[e] + 1]++;
    }
    for (int i = 0; i < total_points; ++i) inv_idx[i + 1] += inv_idx[i];
    int32_t* cur = (int32_t*)malloc(sizeof(int32_t) * (size_t)(total_points + 1));
    for (int i = 0; i < total_points; ++i) cur[i] = inv_idx[i];
    for (size_t e = 0; e < edges; ++e) {
        const int64_t t = idx[e];
        if (t >= 0 && t < total_points) {
            inv_neighbors[cur[t]] = (int32_t)(e / K);
            inv_k[cur[t]] = (uint8_t)(e % K);
            cur[t]++;
        }
    }
    free(cur);
}
