// Shared between edge_mlp.hip (lane-per-row kernels) and edge_mlp_mfma.hip (matrix-core kernels).
#pragma once
#include "pcf_common.h"

namespace pcf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_LEAKY = 2, ACT_SIGMOID = 3 };

struct RowLin {
    const float* x;      // [R, Cin]
    const float* dy;     // [R, Cout]            (backward)
    float* y;            // [R, Cout]            (forward)
    float* dx;           // [R, Cin] or null     (backward)
    const float* W;      // [Cout, Cin]
    const float* b;      // [Cout]
    const float* mean;   // [Cout] or null (no BN)
    const float* rstd;   // [Cout]
    const float* gamma;  // [Cout]
    const float* beta;   // [Cout]
    const float* m1;     // [Cout] sum(g)/R          (backward apply, batch statistics)
    const float* m2;     // [Cout] sum(g*xhat)/R
    float* part;         // per-workgroup partial sums
    long long R;
    int Cin, Cout;
    int batch_stats;     // backward: BN used batch statistics
    int act;             // Act
    // optional extras of the guidance first layer (Cout <= 16):
    const float* gadd;       // [B*gN, Cout] per-point term gathered through gidx and added to x.W^T
    const int64_t* gidx;     // [R] batch-local row of gadd for every row (out of range = no term)
    float* dgadd;            // backward: [B*gN, Cout] float-atomic target (zeroed by the host wrapper)
    long long rows_per_batch;
    int gN;
    int group;               // K (power of two <= 64): subtract the value of the group's first row; 0 = off
    int vec_x, vec_y;    // 16-byte row access allowed
};

// matrix-core implementations (edge_mlp_mfma.hip); return PCF_E_UNSUPPORTED-free bool: true if launched
bool rowlin_mfma_supported(const RowLin& a);
int rowlin_mfma_stats(const RowLin& a, int grid, hipStream_t s);
int rowlin_mfma_forward(const RowLin& a, hipStream_t s);
int rowlin_mfma_bwd_reduce(const RowLin& a, int grid, hipStream_t s);
int rowlin_mfma_bwd_apply(const RowLin& a, int grid, hipStream_t s);
int rowlin_mfma_grid(long long R, int Cin);

}  // namespace pcf
