/*
 * pcf_hip.h -- C ABI of libpcf_hip.so, the MI355X (gfx950) implementation of the
 * PointConvFormer hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch types.  Every entry point
 * replaces one function of the reference's `pcf_cuda` torch extension (or one of its Python kNN
 * helpers); the reference interface each one stands in for is cited as file:line relative to the
 * reference root.  `ml-pointconvformer_amd/pcf_cuda/__init__.py` binds these through ctypes under
 * the reference's own module and function names (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - All pointers are DEVICE pointers (HIP), fp32 / int64 / int32 / uint8 as stated, dense
 *     row-major ("contiguous" in the reference's CHECK_CONTIGUOUS sense, pcf.h:14-24).
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Calls only enqueue work;
 *     they never synchronise the device or the host (the reference does, pconv_ops.cu:887-889).
 *   - Return value: 0 on success, a negative PCF_E_* code otherwise; pcf_hip_last_error() gives
 *     the message for the calling thread.  Nothing is launched when an error is returned.
 *   - Flattened aggregate channel index is c*C_mid + m, c over the C_in gathered channels first and
 *     then the C_add appended ones; head of gathered channel c is c % H (pcf_ops.cu:60-68,
 *     pconv_ops.cu:88-101, layers.py:387-390).  Backward kernels are the exact adjoint of that
 *     forward layout (the reference's CUDA backward is not).
 *   - Neighbour indices outside [0, N) contribute nothing (forward) and receive nothing (backward);
 *     the reference reads out of bounds there.
 *   - Outputs are fully written by the call; no pre-zeroing by the caller is needed.
 *   - Workspaces: ask the matching *_workspace_bytes(), pass a device buffer at least that large
 *     (16-byte aligned).  Contents are scratch.
 */
#ifndef PCF_HIP_H
#define PCF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCF_OK 0
#define PCF_E_BADARG (-1)     /* bad shape / null pointer / misaligned workspace */
#define PCF_E_UNSUPPORTED (-2)/* shape outside what the kernels cover (message says which) */
#define PCF_E_LAUNCH (-3)     /* HIP reported an error at launch */

/* Library identity: "pcf_hip <version> gfx950". */
const char* pcf_hip_version(void);
/* Message of the last error returned on this thread ("" if none). */
const char* pcf_hip_last_error(void);

/* ---- diagnostics (no reference counterpart: the reference has one kernel per operator) --------------------------
 * Launch log: while enabled, every kernel launch of the library appends one line (the kernel's name where a
 * dispatcher chooses between variants, else the launch site's label).  _enable clears the log; _read copies the
 * NUL-terminated text and clears it when `capacity` suffices, and always returns the bytes needed.  Tests use it to
 * assert WHICH kernels a model configuration dispatches to. */
void pcf_hip_launch_log_enable(int on);
size_t pcf_hip_launch_log_read(char* buf, size_t capacity);
/* Buffers are cleared (and the handful of device-to-device copies made) by ordinary kernels of this library, never by
 * hipMemsetAsync / hipMemcpyAsync: captured into a HIP graph those calls become memset / memcpy nodes, and a replayed
 * training iteration whose kNN cell histogram was cleared by a memset node ran the dependent counting sort on uncleared
 * memory (DESIGN.md "graph replay fault").  pcf_hip_zero_host runs the zero kernel's per-thread indexing (head bytes,
 * 16-byte body, tail bytes) on HOST memory over `blocks` x 256 emulated threads -- a test hook that needs no GPU. */
void pcf_hip_zero_host(void* p, size_t bytes, int blocks);
/* Kernel family for the aggregate shapes the matrix-core kernels cover: 0 default, 1 LDS-tiled kernels (cross-check),
 * 2 tiled matrix-core kernels everywhere, 3 thread-per-edge backward of small unguided layers (see below).  Process-wide; the environment (PCF_AGG_LDS=1 / PCF_AGG_TILED=1) only sets
 * the initial value. */
int pcf_hip_set_aggregate_engine(int engine);
int pcf_hip_get_aggregate_engine(void);
/* Engine 3 (opt-in, unmeasured): pconv_backward / pconv_linear*_backward of unguided layers with C_mid 4 or 16 and at most
 * 64 channels per edge through a thread-per-edge kernel without LDS (the level-0 PointConv, layers.py:813-906).  Its
 * per-edge body is __host__ __device__; pcf_hip_pconv_backward_edge_host runs it over every edge on HOST memory (test
 * hook, no GPU; arguments as pcf_hip_pconv_backward plus `contrib` [B*Nout*K, C_in] and `atomic`: accumulate into a
 * zeroed grad_x, or write per-edge contribution rows). */
int pcf_hip_pconv_backward_edge_host(const float* grad_out, const float* x, const int64_t* idx, const float* w, const float* add,
                                     float* grad_x, float* contrib, float* grad_w, float* grad_add, int B, int N, int Nout, int K,
                                     int Ci, int Ca, int Cm, int atomic);
/* Last pass of the fused edge-graph backward (pcf_hip_pcf_chain_backward / pcf_hip_weightnet_chain_backward): 1 = the
 * operands of its outer products are turned through LDS tiles (default: faster by 5 % at 1.28 M edges), 0 = they are
 * produced in registers in the transposed lane layout (no LDS traffic, more vector-memory instructions).  Process-wide;
 * PCF_CHAIN_BWD_LDS=0 sets the initial value. */
int pcf_hip_set_chain_backward_engine(int lds_transposes);

/* ---- point-level Linear + BatchNorm (+ activation) chains --------------------------------------------------------
 * replaces layer_utils.Linear_BN / UnaryBlock (layer_utils.py:241-315) where a PCFLayer strings them together
 * (layers.py:335 unary1, :369 guidance_unary, :393-394 linear + ReLU, :397-400 unary2, :414 residual + LeakyReLU), training
 * mode (batch statistics).  A layer keeps its RAW output z = x W^T + b [R, C] and a record cst [6][C] of per-channel
 * constants: sc = rstd * gamma, sh = beta - mean * sc (so y = act(z * sc + sh)), mean, rstd, and after its backward
 * statistics D1, D0 (dz = g * sc + z * D1 + D0 with g = dy * act'(z * sc + sh)).  Consumers apply the normalisation while
 * loading their operand tiles; statistics are finished by the last workgroup of the launch that produced the partial
 * sums (no separate statistics / finalize / normalise launches).  `tickets`: a persistent, zero-initialised int buffer
 * of pcf_hip_flin_ticket_ints() entries that the kernels leave zeroed; `workspace`: pcf_hip_flin_workspace_bytes().
 * Activation codes: 0 none, 1 ReLU, 2 LeakyReLU(0.1), 3 sigmoid. */
size_t pcf_hip_flin_workspace_bytes(long long rows, int c_out, int c_in);
int pcf_hip_flin_ticket_ints(void);
/* pcf_hip_flin_forward / _backward_input have a split-K form (32 x 32 tiles, the four waves of a workgroup share the
 * contraction) for few rows and long contractions; mode -1: chosen by size (default), 0: never, 1: wherever K >= 64. */
int pcf_hip_set_flin_split_k(int mode);
/* how the per-workgroup column sums of the flin kernels become a layer's record: 1 (default) a small launch of its own,
 * 0 inside the producing kernel by the workgroup that finishes last (tickets). */
int pcf_hip_set_flin_finish(int separate_launch);
int pcf_hip_set_row_chain_finish(int separate_launch);   /* the same choice for the row chains of point_chain.hip */
/* Z[M,N] = f(A)[M,K] W[N,K]^T + bias, f = act_pre(A * pre[0] + pre[1]) when `pre` (the producer's record) is given, else
 * identity; `side` (nullable) receives f(A).  cst != null: batch statistics of Z -> cst rows 0..3 and the running
 * statistics (momentum update with the unbiased variance, as nn.BatchNorm1d). */
int pcf_hip_flin_forward(const float* A, long long M, int K, const float* pre, int pre_act, float* side, const float* W,
                         const float* bias, int N, float* Z, float* cst, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, float eps, float momentum, void* workspace,
                         size_t workspace_bytes, int* tickets, void* stream);
/* dx[M,N] = dz[M,K] W[K,N] (+ add), dz from (dy, z, cst, act) of this layer.  With cstp != null the input was
 * act_p(z_p * sc_p + sh_p): the BatchNorm-backward sums of that producing layer are taken from (dx, zp) and its dgamma,
 * dbeta, zero bias gradient and record rows 4, 5 are written. */
int pcf_hip_flin_backward_input(const float* dy, const float* z, const float* cst, int act, long long M, int K, const float* W,
                                int N, const float* add, float* dx, const float* zp, float* cstp, int actp, float* dgamma_p,
                                float* dbeta_p, float* dbias_p, void* workspace, size_t workspace_bytes, int* tickets, void* stream);
/* dW[M,N] = sum_r dz[r,M]^T f(xin)[r,N], dz as above, f the producer's normalisation + activation (pre == null: identity):
 * `_slabs` writes pcf_hip_flin_backward_weight_splits(R, M, N) partial products [splits][M][N] (one per row range);
 * pcf_hip_slab_sum_multi adds the slab lists of up to 8 weight gradients in ONE launch, in a fixed order
 * (out_i[e] = sum_s slabs_i[s][e], counts_i = M_i * N_i). */
int pcf_hip_flin_backward_weight_splits(long long R, int M, int N);
int pcf_hip_flin_backward_weight_slabs(const float* dy, const float* z, const float* cst, int act, const float* xin, const float* pre,
                                       int pre_act, long long R, int M, int N, float* slabs, size_t slabs_bytes, void* stream);
int pcf_hip_slab_sum_multi(int n, const float* const* slabs, float* const* out, const long long* counts, const int* splits,
                           void* stream);
/* Top of a chain, y = act(z * sc + sh + res): g = dy * act'(.) [R,C] (also the gradient of `res`), dgamma, dbeta, zero
 * bias gradient and record rows 4, 5 of this layer. */
int pcf_hip_bn_backward_stats(const float* dy, const float* z, const float* res, float* cst, int act, long long R, int C,
                              float* g, float* dgamma, float* dbeta, float* dbias, void* workspace, size_t workspace_bytes, int* tickets,
                              void* stream);

/* ---- PCFLayer point-level layers as row chains (narrow widths) ------------------------------------------------------
 * The head of a PCFLayer -- unary1 (Linear+BN+LeakyReLU, layers.py:335), guidance_unary (Linear+BN, :369) and the gathered
 * half of the first guidance layer (u = guidance_x Wa^T) -- in three passes forward and three backward that keep every
 * intermediate of a 16-row tile in registers (weights and BatchNorm constants in LDS); see csrc/point_chain.hip.
 * x [R,c_in] -> z1 [R,mid] (raw unary1 output, kept for the backward), fx [R,mid], u [R,8]; records cst1 [6][mid], cst2 [6][g]
 * as in the flin_* entry points.  `tickets`: one zero int the kernels leave zeroed.  _supported(): widths instantiated. */
int pcf_hip_point_head_supported(int c_in, int mid, int g);
size_t pcf_hip_point_head_workspace_bytes(long long R, int c_in, int mid, int g);
int pcf_hip_point_head_forward(const float* x, long long R, int c_in, int mid, int g, const float* W1, const float* b1,
                               const float* gamma1, const float* beta1, float* rmean1, float* rvar1, float mom1, const float* W2,
                               const float* b2, const float* gamma2, const float* beta2, float* rmean2, float* rvar2, float mom2,
                               const float* Wa, float eps, float* z1, float* fx, float* u, float* cst1, float* cst2, void* workspace,
                               size_t workspace_bytes, int* tickets, void* stream);
int pcf_hip_point_head_backward(const float* dfx, const float* du, const float* x, const float* z1, const float* fx, long long R,
                                int c_in, int mid, int g, const float* W1, const float* W2, const float* b2, const float* Wa,
                                float* cst1, float* cst2, float* dx, float* dW1, float* db1, float* dgamma1, float* dbeta1,
                                float* dW2, float* db2, float* dgamma2, float* dbeta2, float* dWa, void* workspace,
                                size_t workspace_bytes, int* tickets, void* stream);

/* The tail of a PCFLayer -- linear (Linear+BN+ReLU on the aggregate, layers.py:393-394) and unary2 (Linear+BN, :397-400) in
 * front of the residual sum (:414) -- as row chains: z3 [R,c_half], z4 [R,c_out] raw, records cst3 / cst4.  The final
 * out = LeakyReLU(BN4(z4) + shortcut) is pcf_hip_bnact_forward_res with mean / rstd = rows 2 / 3 of cst4.  Backward: g4
 * (the gradient of the shortcut), dagg, dW4, BatchNorm / bias gradients; g3_out receives dy3 masked by the ReLU, from which
 * the caller takes dW3 = dz3^T agg with pcf_hip_flin_backward_weight_slabs(g3_out, z3, cst3, act 1, agg). */
int pcf_hip_point_tail_supported(int c_agg, int c_half, int c_out);
size_t pcf_hip_point_tail_workspace_bytes(long long R, int c_agg, int c_half, int c_out);
int pcf_hip_point_tail_forward(const float* agg, long long R, int c_agg, int c_half, int c_out, const float* W3, const float* b3,
                               const float* gamma3, const float* beta3, float* rmean3, float* rvar3, float mom3, const float* W4,
                               const float* b4, const float* gamma4, const float* beta4, float* rmean4, float* rvar4, float mom4,
                               float eps, float* z3, float* z4, float* cst3, float* cst4, void* workspace, size_t workspace_bytes,
                               int* tickets, void* stream);
int pcf_hip_point_tail_backward(const float* dout, const float* res, const float* z3, const float* z4, long long R, int c_agg,
                                int c_half, int c_out, const float* W3, const float* W4, float* cst3, float* cst4, float* g4,
                                float* g3_out, float* dagg, float* dW4, float* db3, float* dgamma3, float* dbeta3, float* db4,
                                float* dgamma4, float* dbeta4, void* workspace, size_t workspace_bytes, int* tickets, void* stream);

/* pe_convs of PointConvStridePE / PointConvTransposePE (layers.py:599-604, 941-946): WeightNet(3, c_out, [hidden]) = two
 * Linear+BN+ReLU layers on the edge offsets rel [E,3], training mode (batch statistics), as three-pass row chains that keep
 * nothing but the input and the output.  Backward: the eight parameter gradients (the offsets carry none). */
int pcf_hip_pe_chain_supported(int hidden, int c_out);
size_t pcf_hip_pe_chain_workspace_bytes(long long E, int hidden, int c_out);
int pcf_hip_pe_chain_forward(const float* rel, long long E, int hidden, int c_out, const float* W1, const float* b1,
                             const float* gamma1, const float* beta1, float* rmean1, float* rvar1, float mom1, const float* W2,
                             const float* b2, const float* gamma2, const float* beta2, float* rmean2, float* rvar2, float mom2,
                             float eps, float* out, float* cst1, float* cst2, void* workspace, size_t workspace_bytes, int* tickets,
                             void* stream);
int pcf_hip_pe_chain_backward(const float* dout, const float* rel, long long E, int hidden, int c_out, const float* W1, const float* b1,
                              const float* W2, const float* b2, float* cst1, float* cst2, float* dW1, float* db1, float* dgamma1,
                              float* dbeta1, float* dW2, float* db2, float* dgamma2, float* dbeta2, void* workspace,
                              size_t workspace_bytes, int* tickets, void* stream);

/* clip_grad_norm_ + AdamW.step() of the training loop (train_ScanNet_DDP_WarmUP.py:237-241, :421) over lists of parameter
 * tensors carried as kernel arguments (at most pcf_hip_adamw_max_tensors() per call; one partial per pcf_hip_adamw_chunk()
 * elements).  rec: device float[8] = lr, step, norm, coef, 1 - beta1^step, sqrt(1 - beta2^step).  Sequence per step: phase 0
 * per list, pcf_hip_adamw_finish, phase 1 per list.  Arithmetic of torch.optim.AdamW(fused=True). */
int pcf_hip_adamw_max_tensors(void);
int pcf_hip_adamw_chunk(void);
int pcf_hip_adamw_list(int phase, int n, float* const* params, float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                       const long long* counts, float* rec, float* partials, int partial0, double beta1, double beta2, double eps,
                       double weight_decay, void* stream);
int pcf_hip_adamw_finish(const float* partials, int n_partials, float* rec, float max_norm, double beta1, double beta2, void* stream);

/* criterion(pred, target) of the training loop (train_ScanNet_DDP_WarmUP.py:243, :404): nn.CrossEntropyLoss(ignore_index,
 * label_smoothing), mean over the rows that are not ignored.  Forward: stat[0] = loss, stat[1] = number of valid rows, dlogits =
 * gradient of the loss sum; backward: dx = dlogits * grad_out / stat[1].  At most 64 classes; deterministic. */
size_t pcf_hip_cross_entropy_workspace_bytes(long long R);
int pcf_hip_cross_entropy_forward(const float* logits, const int64_t* target, long long R, int C, long long ignore_index,
                                  float label_smoothing, float* stat, float* dlogits, void* workspace, size_t workspace_bytes,
                                  void* stream);
int pcf_hip_cross_entropy_backward(const float* dlogits, const float* grad_out, const float* stat, long long R, int C, float* dx,
                                   void* stream);

/* ---- attention arithmetic of the ablation layers ------------------------------------------------
 * softmax_aggregate: PointTransformerLayer.forward, layers.py:519-527.  v [R,K,C], logit [R,K,J] (J divides C: the
 *   share_planes groups) -> sm = softmax over K of logit (saved for the backward), out[r,c] = sum_k v[r,k,c] * sm[r,k,c % J].
 * qk_score: MultiHeadGuidanceQK.forward, layers.py:100-114.  q [R,K,H,D], key [R,H,D] -> sigmoid(scale * <q, key>) [R,K,H].
 * layer_norm: nn.LayerNorm over the last axis (layers.py:33-36, 52-53), x [R,C]; mean / rstd [R] are saved for the backward. */
int pcf_hip_softmax_aggregate_forward(const float* v, const float* logit, float* out, float* sm, long long R, int K, int C,
                                      int J, void* stream);
int pcf_hip_softmax_aggregate_backward(const float* dout, const float* v, const float* sm, float* dv, float* dlogit,
                                       long long R, int K, int C, int J, void* stream);
int pcf_hip_qk_score_forward(const float* q, const float* key, float* score, long long R, int K, int H, int D, float scale,
                             void* stream);
int pcf_hip_qk_score_backward(const float* dscore, const float* score, const float* q, const float* key, float* dq, float* dkey,
                              long long R, int K, int H, int D, float scale, void* stream);
int pcf_hip_layer_norm_forward(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                               long long R, int C, float eps, void* stream);
size_t pcf_hip_layer_norm_backward_workspace_bytes(long long R, int C);
int pcf_hip_layer_norm_backward(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                float* dx, float* dgamma, float* dbeta, long long R, int C, void* workspace,
                                size_t workspace_bytes, void* stream);

/* ---- guided aggregate (PCF) ------------------------------------------------------------------
 * replaces pcf_cuda.pcf_forward / pcf_backward        (pcf_cuda.cpp:10-11, pcf.h:38-66,
 *                                                       pcf_ops.cu:27-71,87-141,143-202)
 * x [B,N,Ci] f32, idx [B,Nout,K] i64, guid [B,Nout,K,H] f32, w [B,Nout,K,Cm] f32
 * out [B,Nout,Ci*Cm]:  out[b,n,c*Cm+m] = sum_k x[b,idx[b,n,k],c] * guid[b,n,k,c%H] * w[b,n,k,m]
 */
int pcf_hip_pcf_forward(const float* x, const int64_t* idx, const float* guid, const float* w,
                        float* out, int B, int N, int Nout, int K, int Ci, int Cm, int H, void* stream);
/* grad_x [B,N,Ci] (scatter-add, float atomics), grad_guid [B,Nout,K,H], grad_w [B,Nout,K,Cm]. */
int pcf_hip_pcf_backward(const float* grad_out, const float* x, const int64_t* idx, const float* guid,
                         const float* w, float* grad_x, float* grad_guid, float* grad_w, int B, int N,
                         int Nout, int K, int Ci, int Cm, int H, void* stream);

/* pcf_backward with grad_x as a deterministic gather-reduce over the inverse CSR (no atomics): the
 * PCFLayer receives inv_neighbors / inv_k / inv_idx from the model (layers.py:315-317) but the
 * reference never uses them there.  Workspace holds one contribution row per edge. */
size_t pcf_hip_pcf_backward_csr_workspace_bytes(int B, int Nout, int K, int Ci);
int pcf_hip_pcf_backward_csr(const float* grad_out, const float* x, const int32_t* inv_neighbors, const uint8_t* inv_k,
                             const int32_t* inv_idx, const int64_t* idx, const float* guid, const float* w,
                             float* grad_x, float* grad_guid, float* grad_w, void* workspace, size_t workspace_bytes,
                             int B, int N, int Nout, int K, int Ci, int Cm, int H, int inv_len, int inv_idx_len,
                             void* stream);

/* ---- unguided aggregate with appended per-edge features (PConv) ------------------------------
 * replaces pcf_cuda.pconv_forward / pconv_backward     (pcf_cuda.cpp:12,14, pcf.h:81-112,
 *                                                       pconv_ops.cu:40-103,240-290,648-776)
 * add [B,Nout,K,Ca] f32 (Ca may be 0; then `add`/`grad_add` may be NULL)
 * out [B,Nout,(Ci+Ca)*Cm]: out[b,n,c*Cm+m] = sum_k cat(x[b,idx[b,n,k]], add[b,n,k])[c] * w[b,n,k,m]
 */
int pcf_hip_pconv_forward(const float* x, const int64_t* idx, const float* w, const float* add,
                          float* out, int B, int N, int Nout, int K, int Ci, int Ca, int Cm, void* stream);
int pcf_hip_pconv_backward(const float* grad_out, const float* x, const int64_t* idx, const float* w,
                           const float* add, float* grad_x, float* grad_w, float* grad_add, int B, int N,
                           int Nout, int K, int Ci, int Ca, int Cm, void* stream);

/* ---- aggregate + linear ----------------------------------------------------------------------
 * replaces pcf_cuda.pconv_linear_forward AND pcf_cuda.pconv_linear_cutlass_forward
 *                                                      (pcf_cuda.cpp:13,18, pcf.h:131-138,243-250,
 *                                                       pconv_ops.cu:129-225,691-736,969-1269)
 * lin_w [Co,(Ci+Ca)*Cm], lin_b [Co];  out [B,Nout,Co] = pconv_out . lin_w^T + lin_b,
 * pconv_out [B,Nout,(Ci+Ca)*Cm] is returned as well (the reference saves it for backward).
 */
int pcf_hip_pconv_linear_forward(const float* x, const int64_t* idx, const float* w, const float* add,
                                 const float* lin_w, const float* lin_b, float* out, float* pconv_out,
                                 int B, int N, int Nout, int K, int Ci, int Ca, int Cm, int Co, void* stream);

/* replaces pcf_cuda.pconv_linear_backward              (pcf_cuda.cpp:15, pcf.h:162-170,
 *                                                       pconv_ops.cu:293-388,797-845)
 * grad_x via float atomics.  grad_lin_w [Co,(Ci+Ca)*Cm], grad_lin_b [Co]. */
size_t pcf_hip_pconv_linear_backward_workspace_bytes(int B, int N, int Nout, int K, int Ci, int Ca, int Cm,
                                                     int Co);
int pcf_hip_pconv_linear_backward(const float* grad_out, const float* x, const int64_t* idx, const float* w,
                                  const float* add, const float* lin_w, const float* pconv_out, float* grad_x,
                                  float* grad_w, float* grad_add, float* grad_lin_w, float* grad_lin_b,
                                  void* workspace, size_t workspace_bytes, int B, int N, int Nout, int K,
                                  int Ci, int Ca, int Cm, int Co, void* stream);

/* replaces pcf_cuda.pconv_linear_opt_backward          (pcf_cuda.cpp:16, pcf.h:213-224,
 *                                                       pconv_ops.cu:391-619,863-948)
 * inv_neighbors i32 [B,inv_len], inv_k u8 [B,inv_len], inv_idx i32 [B,inv_idx_len] with
 * inv_idx_len >= N+1: the CSR transpose of idx as produced by pcf_hip_knn_inverse.  grad_x is
 * a deterministic gather-reduce over that CSR for EVERY input row (no atomics).  The reference
 * uses the CSR only for rows >= Nout and does not check it matches idx; neither does this. */
size_t pcf_hip_pconv_linear_opt_backward_workspace_bytes(int B, int N, int Nout, int K, int Ci, int Ca,
                                                         int Cm, int Co);
int pcf_hip_pconv_linear_opt_backward(const float* grad_out, const float* x, const int32_t* inv_neighbors,
                                      const uint8_t* inv_k, const int32_t* inv_idx, const int64_t* idx,
                                      const float* w, const float* add, const float* lin_w,
                                      const float* pconv_out, float* grad_x, float* grad_w, float* grad_add,
                                      float* grad_lin_w, float* grad_lin_b, void* workspace,
                                      size_t workspace_bytes, int B, int N, int Nout, int K, int Ci, int Ca,
                                      int Cm, int Co, int inv_len, int inv_idx_len, void* stream);

/* ---- CSR transpose of the neighbour table ----------------------------------------------------
 * replaces pcf_cuda.compute_knn_inverse                (pcf_cuda.cpp:17, pcf.h:183-186,
 *                                                       knn.cu:24-168)
 * idx [B,Nq,K] i64 -> inv_neighbors i32 [B,Nq*K], inv_k u8 [B,Nq*K], inv_idx i32 [B,total_points+1].
 * Entries of idx outside [0,total_points) are skipped (knn.cu:38,76); unused tail entries are 0.
 * Buckets are ordered by (query, k) -- deterministic; the reference's order is atomics-order.
 * K <= 255 (inv_k is a byte). */
size_t pcf_hip_knn_inverse_workspace_bytes(int B, int Nq, int K, int total_points);
int pcf_hip_knn_inverse(const int64_t* idx, int32_t* inv_neighbors, uint8_t* inv_k, int32_t* inv_idx,
                        void* workspace, size_t workspace_bytes, int B, int Nq, int K, int total_points,
                        void* stream);

/* ---- k nearest neighbours over a packed batch ------------------------------------------------
 * replaces knn_post_dataloader_utils.compute_knn / knn_keops, one call per (level, relation)
 * instead of one per (sample, level, relation)         (knn_post_dataloader_utils.py:22-87,171-223)
 * ref [Nr,3] f32, query [Nq,3] f32 packed over n_seg samples; ref_off / query_off i32 [n_seg+1]
 * (device) are the per-sample prefix offsets.  out i64 [Nq,K]: for each query the K nearest refs
 * OF ITS OWN SAMPLE as global (packed) ref indices, ordered by (distance, index) ascending, where
 * distance = ((rx-qx)^2 + (ry-qy)^2) + (rz-qz)^2 in fp32, one rounding per operation.
 * Samples with fewer than K refs get -1 in the unfilled slots.  K <= 64. */
int pcf_hip_knn(const float* ref, const float* query, const int32_t* ref_off, const int32_t* query_off,
                int n_seg, int max_queries_per_seg, int K, int64_t* out, void* stream);

/* Same contract and bit-identical output as pcf_hip_knn, one wave per query (64 lanes share the distance evaluations
 * and select the K smallest by wave-wide minimum rounds): the engine for the coarse levels, where a few thousand
 * queries leave lane-per-query kernels latency-bound.  Cost ~ n_query x ceil(refs per sample / 1024). */
int pcf_hip_knn_wave(const float* ref, const float* query, const int32_t* ref_off, const int32_t* query_off,
                     int n_seg, int n_query, int K, int64_t* out, void* stream);

/* Same contract and bit-identical output as pcf_hip_knn, through a uniform-grid index over the reference
 * points (cell sort + ring search); the engine for large clouds.  n_ref / n_query are the packed totals. */
size_t pcf_hip_knn_grid_workspace_bytes(int n_ref, int n_seg);
int pcf_hip_knn_grid(const float* ref, const float* query, const int32_t* ref_off, const int32_t* query_off,
                     int n_seg, int n_ref, int n_query, int K, int64_t* out, void* workspace,
                     size_t workspace_bytes, void* stream);

/* All neighbour tables of a training iteration transposed in one pass of launches (8 instead of 8 per table):
 * replaces the call pattern of util/common_util.compute_knn_inverse (:281-309: one pcf_cuda.compute_knn_inverse per
 * level and relation = 3 x levels calls per iteration).  Table i: idx[i] i64 [Nq[i], K[i]] -> inv_neighbors[i] i32
 * [Nq[i]*K[i]], inv_k[i] u8 [Nq[i]*K[i]], inv_idx[i] i32 [total_points[i]+1]; same contents as pcf_hip_knn_inverse with
 * B = 1 (buckets sorted by (query, k)).  The pointer and size lists are host arrays; every table must be non-empty
 * (Nq >= 1, total_points >= 1). */
size_t pcf_hip_knn_inverse_batched_workspace_bytes(int n_tables, const int* Nq, const int* K, const int* total_points);
int pcf_hip_knn_inverse_batched(int n_tables, const int64_t* const* idx, int32_t* const* inv_neighbors,
                                uint8_t* const* inv_k, int32_t* const* inv_idx, const int* Nq, const int* K,
                                const int* total_points, void* workspace, size_t workspace_bytes, void* stream);

/* ---- multi-resolution levels: barycentre grid subsampling of a packed batch --------------------------
 * replaces cpp_subsampling.compute(points, features=..., sampleDl=..., method="barycenters")
 *   (cpp_wrappers/cpp_subsampling/wrapper.cpp:200-290 -> grid_subsampling/grid_subsampling.cpp:9-110),
 * as called per sample and level from datasetCommon.grid_subsampling / subsample (datasetCommon.py:17-67, :384-421).
 * points [N,3] f32 and features [N,F] f32 (F may be 0: features NULL) packed over n_seg samples; seg_off i32
 * [n_seg+1] (device) are the per-sample prefix offsets.  Per sample: origin = floor(min_corner * (1/dl)) * dl and
 * voxel (i,j,k) = floor((p - origin) / dl) in fp32 as the reference computes them; every occupied voxel yields the
 * barycentre of its points (float sum in point order x float(1.0/count)) and the mean of their features (float sum
 * / float(count)) -- bit-identical to the reference.  Output order: by sample, then by the reference's linear voxel
 * index i + nX*j + nX*nY*k ascending (the reference itself emits std::unordered_map order).
 * out_points [N,3] / out_features [N,F] have room for N rows; out_seg_counts i32 [n_seg] receives the voxels per
 * sample and out_total i32 [2] = {voxel total, status} (status != 0: a sample spans >= 2^18 voxels along an axis or
 * holds non-finite coordinates).  All on `stream`; nothing is read back by the library. */
size_t pcf_hip_grid_subsample_workspace_bytes(int n_points, int n_seg);
int pcf_hip_grid_subsample(const float* points, const float* features, const int32_t* seg_off, int n_seg, int n_points,
                           int F, float sampleDl, float* out_points, float* out_features, int32_t* out_seg_counts,
                           int32_t* out_total, void* workspace, size_t workspace_bytes, void* stream);

/* ---- level-0 voxelisation: at most one point per occupied voxel --------------------------------------------------
 * replaces util/voxelize.py:44-82 `voxelize(coord, voxel_size, hash_type='fnv', mode=...)`, which thins the raw scan to the
 * level-0 resolution in the dataloader (scannet_data_loader_color_DDP.py:210-215).  points [N,3] f32 (one cloud);
 * key = FNV64-1A of floor(coord / voxel_size) per axis as uint64 (the quotient in double, as numpy computes it);
 * out_index i64 [N] (room for N) receives the chosen point of every voxel in ascending key order -- the reference's
 * idx_sort order --; mode 0: the voxel's first point in index order ('deterministic'), 1: a pseudo-random point
 * (splitmix64 of seed and voxel rank: 'random'), 2: point `rank` mod the voxel's count (one set of 'multiple').
 * out_total i32 [2] = {voxels, points in the fullest voxel}. */
size_t pcf_hip_voxelize_workspace_bytes(int n_points);
int pcf_hip_voxelize(const float* points, int n_points, double voxel_size, int mode, unsigned long long seed, int rank,
                     int64_t* out_index, int32_t* out_total, void* workspace, size_t workspace_bytes, void* stream);
/* The same for float64 coordinates (a loader that keeps its coordinates in double: the quotient is then taken on the double
 * values themselves, as numpy does for float64 input under any version -- not on their float32 roundings). */
int pcf_hip_voxelize_f64(const double* points, int n_points, double voxel_size, int mode, unsigned long long seed, int rank,
                     int64_t* out_index, int32_t* out_total, void* workspace, size_t workspace_bytes, void* stream);

/* ---- per-edge helpers around the aggregate ---------------------------------------------------
 * replace index_points (layer_utils.py:13-30) and its index_put_ backward: */
/* out[b,s,:] = table[b, idx[b,s], :]   table [B,N,C], idx [B,S] i64 (S = M*K for a neighbour table) */
int pcf_hip_gather_rows(const float* table, const int64_t* idx, float* out, int B, int N, long long S, int C,
                        void* stream);
/* grad_table[b, idx[b,s], :] += grad_rows[b,s,:]  (grad_table is zeroed first; float atomics) */
int pcf_hip_scatter_add_rows(const float* grad_rows, const int64_t* idx, float* grad_table, int B, int N,
                             long long S, int C, void* stream);
/* replaces torch.max(index_points(dense_feats, nei_inds), dim=2)[0] (layers.py:403-408,728-733):
 * out[b,m,c] = max_k table[b, idx[b,m,k], c]; argk u8 [B,M,C] = first maximising k (for backward). */
int pcf_hip_gather_max(const float* table, const int64_t* idx, float* out, uint8_t* argk, int B, int N, int M,
                       int K, int C, void* stream);
int pcf_hip_gather_max_backward(const float* grad_out, const int64_t* idx, const uint8_t* argk, float* grad_table,
                                int B, int N, int M, int K, int C, void* stream);
/* replaces index_points(xyz) - centre, index_points(normals) and VI_coordinate_transform
 * (layers.py:337-358, layer_utils.py:176-231) in one pass:
 * rel [B,M,K,3] = ref_xyz[idx] - ctr_xyz (nullable), vi [B,M,K,12] (nullable; needs both normals). */
int pcf_hip_edge_geometry(const float* ref_xyz, const float* ref_norm, const int64_t* idx, const float* ctr_xyz,
                          const float* ctr_norm, float* rel, float* vi, int B, int N, int M, int K, void* stream);
/* VI_coordinate_transform on already-gathered tensors (layer_utils.py:176-231). */
int pcf_hip_vi_from_gathered(const float* rel, const float* nbr_norm, const float* ctr_norm, float* vi, int B, int M,
                             int K, void* stream);

/* ---- per-edge MLP layer: y = act(BN(x . W^T + b)) over R rows, Cin, Cout <= 64 ----------------
 * replaces Linear_BN.forward on [B,M,K,C] tensors (layer_utils.py:241-277 with
 * util/cp_batchnorm.py:9-30) plus the activation after it in WeightNet (layers.py:163-171),
 * MultiHeadGuidance (layers.py:47-68) and PCFLayer.mlp_conv (layers.py:361-362), and its autograd.
 * act: 0 none, 1 ReLU, 2 LeakyReLU(0.1), 3 sigmoid.  mean == NULL means "no BatchNorm".
 * Workspace: pcf_hip_rowlin_workspace_bytes(Cin, Cout), 16-byte aligned, scratch. */
size_t pcf_hip_rowlin_workspace_bytes(int Cin, int Cout);
/* Batch statistics of z = x.W^T + b (biased variance): mean_out, rstd_out = 1/sqrt(var+eps) [Cout];
 * running_mean / running_var (nullable) are updated with `momentum` (unbiased variance), as
 * F.batch_norm does in training mode. */
int pcf_hip_rowlin_bn_stats(const float* x, long long R, int Cin, const float* W, const float* b, int Cout, float eps,
                            float momentum, float* running_mean, float* running_var, float* mean_out,
                            float* rstd_out, void* workspace, size_t workspace_bytes, void* stream);
int pcf_hip_rowlin_forward(const float* x, long long R, int Cin, const float* W, const float* b, int Cout,
                           const float* mean, const float* rstd, const float* gamma, const float* beta, int act,
                           float* y, void* stream);
/* dy [R,Cout] -> dx [R,Cin] (nullable: skipped), dW [Cout,Cin], db [Cout], dgamma / dbeta [Cout].
 * batch_stats != 0: BN used the statistics of this batch (training); 0: fixed statistics. */
int pcf_hip_rowlin_backward(const float* x, const float* dy, long long R, int Cin, const float* W, const float* b,
                            int Cout, const float* mean, const float* rstd, const float* gamma, const float* beta,
                            int batch_stats, int act, float* dx, float* dW, float* db, float* dgamma, float* dbeta,
                            void* workspace, size_t workspace_bytes, void* stream);

/* Extended forms used by the first layer of the guidance MLP (MultiHeadGuidance, layers.py:47-68,
 * fed by layers.py:372-381).  With W = [Wa | Wb] split at the gathered / positional halves of the
 * query, W.(q - key) = (u[idx[e]] + Wb.pe[e]) - (the same for the key edge), u = Wa.guidance_x being a
 * PER-POINT product: the 64-wide q - key tensor never exists.
 *   gadd [B*gN, Cout] (u), gidx i64 [R] batch-local rows of gadd (the neighbour table), rows_per_batch = M*K;
 *   group = K (power of two <= 64): subtract the pre-bias value of the group's first row (key = neighbour 0).
 * Cout <= 16.  dgadd [B*gN, Cout] receives the gradient of gadd (zeroed here, float atomics). */
int pcf_hip_rowlin_bn_stats_ex(const float* x, long long R, int Cin, const float* W, const float* b, int Cout, float eps,
                               float momentum, float* running_mean, float* running_var, float* mean_out,
                               float* rstd_out, const float* gadd, const int64_t* gidx, long long rows_per_batch,
                               int gN, int group, void* workspace, size_t workspace_bytes, void* stream);
int pcf_hip_rowlin_forward_ex(const float* x, long long R, int Cin, const float* W, const float* b, int Cout,
                              const float* mean, const float* rstd, const float* gamma, const float* beta, int act,
                              const float* gadd, const int64_t* gidx, long long rows_per_batch, int gN, int group,
                              float* y, void* stream);
int pcf_hip_rowlin_backward_ex(const float* x, const float* dy, long long R, int Cin, const float* W, const float* b,
                               int Cout, const float* mean, const float* rstd, const float* gamma, const float* beta,
                               int batch_stats, int act, const float* gadd, const int64_t* gidx,
                               long long rows_per_batch, int gN, int group, float* dx, float* dW, float* db,
                               float* dgamma, float* dbeta, float* dgadd, void* workspace, size_t workspace_bytes,
                               void* stream);

/* ---- wide point-level Linear_BN: BatchNorm (+activation) over the rows of [R, C], any C ------------
 * replaces Linear_BN / UnaryBlock on [B,N,C] tensors wider than 64 channels (layer_utils.py:241-319;
 * PCFLayer.linear layers.py:273-277,393; decoder linears layers.py:973-981): z = x.W^T + b comes from
 * pcf_hip_gemm_nt, these kernels do the statistics, y = act(BN(z)) and its backward, and
 * pcf_hip_linear_backward the three products dx = dz.W, dW = dz^T.x, db = colsum(dz) on the MFMA path. */
size_t pcf_hip_bnact_workspace_bytes(long long R, int C);
int pcf_hip_bnact_stats(const float* z, long long R, int C, float eps, float momentum, float* running_mean,
                        float* running_var, float* mean_out, float* rstd_out, void* workspace,
                        size_t workspace_bytes, void* stream);
int pcf_hip_bnact_forward(const float* z, long long R, int C, const float* mean, const float* rstd,
                          const float* gamma, const float* beta, int act, float* y, void* stream);
int pcf_hip_bnact_backward(const float* z, const float* dy, long long R, int C, const float* mean, const float* rstd,
                           const float* gamma, const float* beta, int batch_stats, int act, float* dz,
                           float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, void* stream);
/* The same with a residual branch (leaky_relu(unary2(.) + shortcut), layers.py:408-416 / :737-741): y = act(BN(z) +
 * residual); backward also returns dresidual = dy * act'(.) (nullable) and, with batch statistics, writes C zeros
 * to dbias_zero (nullable): the gradient of the Linear bias that feeds this BatchNorm. */
int pcf_hip_bnact_forward_res(const float* z, const float* residual, long long R, int C, const float* mean,
                              const float* rstd, const float* gamma, const float* beta, int act, float* y, void* stream);
int pcf_hip_bnact_backward_res(const float* z, const float* residual, const float* dy, long long R, int C, const float* mean,
                               const float* rstd, const float* gamma, const float* beta, int batch_stats, int act,
                               float* dz, float* dresidual, float* dgamma, float* dbeta, float* dbias_zero,
                               void* workspace, size_t workspace_bytes, void* stream);
size_t pcf_hip_linear_backward_workspace_bytes(long long R, int Cin, int Cout);
int pcf_hip_linear_backward(const float* dz, const float* x, const float* W, long long R, int Cin, int Cout,
                            float* dx, float* dW, float* db, void* workspace, size_t workspace_bytes, void* stream);

/* ---- guidance difference (query - key) --------------------------------------------------------
 * replaces layers.py:372-381 + layers.py:52-53: q = cat(index_points(guidance_x, nei), feat_pe),
 * key = q[:, :, :1] (self) or q.max(dim=2) (strided, use_max != 0), s = q - key.
 * gx [B,N,G], idx [B,M,K], pe [B,M,K,P] -> s [B,M,K,G+P]; argk u8 [B,M,G+P] = key index (max only). */
int pcf_hip_guidance_diff_forward(const float* gx, const int64_t* idx, const float* pe, float* s, uint8_t* argk, int B,
                                  int N, int M, int K, int G, int P, int use_max, void* stream);
/* ds [B,M,K,G+P] -> dgx [B,N,G] (zeroed, float atomics), dpe [B,M,K,P]; argk NULL = key was k = 0. */
int pcf_hip_guidance_diff_backward(const float* ds, const int64_t* idx, const uint8_t* argk, float* dgx, float* dpe,
                                   int B, int N, int M, int K, int G, int P, void* stream);

/* ---- fused per-edge graph of a PCFLayer over self neighbourhoods ---------------------------------
 * replaces, in ONE entry point per direction, the edge-level part of PCFLayer.forward (layers.py:361-384):
 *   feat_pe = mlp_conv(VI)                                     Linear_BN(12 -> g) + ReLU        layers.py:361
 *   score   = MultiHeadGuidance(cat(gathered guidance, feat_pe) - key)      two Linear_BN       layers.py:47-68,372-381
 *   weights = WeightNet(VI)                                     three Linear_BN + ReLU           layers.py:163-171,384
 * vi [E, cv] (E = B*M*K edges, cv <= 12), idx i64 [E] neighbour table, u [B*N, 8] = per-point half of the first
 * guidance layer (see pcf_hip_rowlin_*_ex), rows_per_batch = M*K.  Layer order everywhere: mlp_conv, g1
 * (positional half of its weight, [8, g]), g2, w1, w2, w3; hidden widths 8; g <= 32, heads <= 8, cm <= 16;
 * K a power of two <= 16 and E % 16 == 0 (PCF_E_UNSUPPORTED otherwise: callers fall back to the
 * layer-at-a-time entry points above).
 * stats [12][64] floats on the device: mean of layer l at stats + 64*l, 1/sqrt(var+eps) at stats + 64*(6+l).
 * forward, batch_stats != 0: statistics are computed (four passes: the first two from vi, the last two from the stored accumulators), written to stats and folded into
 * the running statistics (nullable); batch_stats == 0: the caller provides them and only score / w are produced.
 * pe, a1, h1, a2 (nullable) receive the intermediate activations [E, g], [E, 8], [E, 8], [E, 8] (for the
 * layer-at-a-time backward); h1_acc, a2_acc (nullable) the raw 8-channel accumulators of g1 and w2 (pre-BatchNorm,
 * bias not added; for pcf_hip_pcf_chain_backward). */
size_t pcf_hip_pcf_chain_workspace_bytes(void);
int pcf_hip_pcf_chain_forward(const float* vi, const int64_t* idx, const float* u, long long E, long long rows_per_batch,
                              int N, int K, int cv, int g, int heads, int cm, const float* const* W,
                              const float* const* b, const float* const* gamma, const float* const* beta,
                              float* const* running_mean, float* const* running_var, float eps, float momentum,
                              int batch_stats, float* stats, float* pe, float* a1, float* h1, float* a2, float* h1_acc,
                              float* a2_acc, float* score, float* w, void* workspace, size_t workspace_bytes,
                              void* stream);
/* Adjoint of the training-mode forward given dscore [E, heads], dw [E, cm] (from pcf_hip_pcf_backward), the
 * stats and the h1_acc / a2_acc of the forward: du [B*N, 8] (zeroed here, float atomics) and, per layer, dW, db,
 * dgamma, dbeta.  Three passes: the top layers are recomputed from the two stored accumulators, the first layers
 * from vi, their weight gradients from moments that are linear in the BatchNorm sums; 64 B per edge of masked
 * gradients go through the workspace.  db is written as zeros (a bias in front of a training-mode BatchNorm has
 * an identically zero gradient). */
size_t pcf_hip_pcf_chain_backward_workspace_bytes(long long E);
int pcf_hip_pcf_chain_backward(const float* vi, const int64_t* idx, const float* h1_acc, const float* a2_acc,
                               const float* dscore, const float* dw, long long E, long long rows_per_batch, int N, int K, int cv, int g, int heads, int cm,
                               const float* const* W, const float* const* b, const float* const* gamma,
                               const float* const* beta, const float* stats, float* du, float* const* dW,
                               float* const* db, float* const* dgamma, float* const* dbeta, void* workspace,
                               size_t workspace_bytes, void* stream);

/* Strided PCFLayers (layers.py:372-375): key = maximum of the query over the neighbourhood instead of neighbour 0.  ukey
 * [E / K, 8] = Wa . max_k guidance_x[idx[n, k]] is the gathered half of that key, formed per centre by the caller (K >= 2);
 * the positional half is taken inside the kernels.  The backward adds dukey [E / K, 8]; both forms share every kernel. */
int pcf_hip_pcf_chain_forward_maxkey(const float* ukey, const float* vi, const int64_t* idx, const float* u, long long E,
                                     long long rows_per_batch, int N, int K, int cv, int g, int heads, int cm,
                                     const float* const* W, const float* const* b, const float* const* gamma,
                                     const float* const* beta, float* const* running_mean, float* const* running_var,
                                     float eps, float momentum, int batch_stats, float* stats, float* pe, float* a1, float* h1,
                                     float* a2, float* h1_acc, float* a2_acc, float* score, float* w, void* workspace,
                                     size_t workspace_bytes, void* stream);
int pcf_hip_pcf_chain_backward_maxkey(const float* ukey, float* dukey, const float* vi, const int64_t* idx, const float* h1_acc,
                                      const float* a2_acc, const float* dscore, const float* dw, long long E,
                                      long long rows_per_batch, int N, int K, int cv, int g, int heads, int cm,
                                      const float* const* W, const float* const* b, const float* const* gamma,
                                      const float* const* beta, const float* stats, float* du, float* const* dW,
                                      float* const* db, float* const* dgamma, float* const* dbeta, void* workspace,
                                      size_t workspace_bytes, void* stream);

/* WeightNet alone (layers.py:127-191: Linear_BN + ReLU x 3, cin -> 8 -> 8 -> C_mid), the branch every PointConv-family
 * layer feeds its aggregate with: same kernels as the WeightNet branch of the fused PCFLayer edge graph, no
 * neighbourhood structure needed (strided / transposed / any K).  x [E, cin] (cin <= 12), E % 16 == 0, C_mid <= 16.
 * Arrays of 3: w1, w2, w3.  stats [12][64] as above (rows 3..5 and 9..11 are used).  a2_acc [E, 8] (training):
 * raw accumulator of w2, restart point of the last two forward passes and of the backward. */
int pcf_hip_weightnet_chain_forward(const float* x, long long E, int cin, int cm, const float* const* W,
                                    const float* const* b, const float* const* gamma, const float* const* beta,
                                    float* const* running_mean, float* const* running_var, float eps, float momentum,
                                    int batch_stats, float* stats, float* a2_acc, float* w, void* workspace,
                                    size_t workspace_bytes, void* stream);
/* workspace: pcf_hip_pcf_chain_backward_workspace_bytes(E) */
int pcf_hip_weightnet_chain_backward(const float* x, const float* a2_acc, const float* dw, long long E, int cin, int cm,
                                     const float* const* W, const float* const* b, const float* const* gamma,
                                     const float* const* beta, const float* stats, float* const* dW, float* const* db,
                                     float* const* dgamma, float* const* dbeta, void* workspace, size_t workspace_bytes,
                                     void* stream);

/* ---- dense fp32 contraction used by the linear stage (exposed for tests / roofline) ----------
 * C[M,N] = A[M,Kd] . B^T  (+ bias[N] if bias != NULL), B given as [N,Kd] row-major.  MFMA f32. */
int pcf_hip_gemm_nt(const float* A, const float* Bm, const float* bias, float* C, int M, int N, int Kd,
                    void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PCF_HIP_H */
