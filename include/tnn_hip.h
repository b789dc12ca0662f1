/*
 * tnn_hip.h — C-ABI of the MI355X (gfx950) dense-tensor backend for tinynn-autograd.
 *
 * The reference (borgwang/tinynn-autograd) has no FFI of its own: its seam is the Python module
 * pair core/tensor.py + core/ops.py whose numpy expressions are listed below next to the entry
 * point that replaces each of them.  This header is that seam restated as a plain C interface:
 * `extern "C"`, raw device pointers + sizes, no torch / numpy types.  The ctypes binding lives in
 * tinynn-autograd_amd/_lib.py; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; tnn_last_error() gives the text
 *     (thread-local).  Nothing is retained past a call except the stream, the pool and comm handles.
 *   - all device work is enqueued on ONE library-owned HIP stream (tnn_stream_sync() waits for it);
 *     calls are asynchronous unless stated otherwise.
 *   - arrays are dense row-major ("C order"); leading dimensions are in ELEMENTS.
 *   - dtype codes: TNN_F32 (compute type of the hot path), TNN_F64 (exact-test mode, the reference's
 *     de-facto type), TNN_I64 (indices), TNN_U8 (numpy bool masks).
 */
#ifndef TNN_HIP_H_
#define TNN_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TNN_API __attribute__((visibility("default")))

enum { TNN_F32 = 0, TNN_F64 = 1, TNN_I64 = 2, TNN_U8 = 3, TNN_BF16 = 4 /* storage type of the bf16 path only */ };

/* binary elementwise ops — reference core/ops.py:33 (add), :66 (mul), :94 (div), :122 (pow),
 * :167 (maximum), :192 (minimum) and the arithmetic inside their vjp bodies */
enum { TNN_ADD = 0, TNN_SUB = 1, TNN_MUL = 2, TNN_DIV = 3, TNN_POW = 4, TNN_MAX = 5, TNN_MIN = 6 };
/* comparisons — core/tensor.py:48-58 (raw bool arrays), core/ops.py:170,173,195,198,229,238,338,340 */
enum { TNN_GT = 0, TNN_GE = 1, TNN_LT = 2, TNN_LE = 3, TNN_EQ = 4, TNN_NE = 5 };
/* unary ops — core/ops.py:217 (exp), :244 (log), :294 (neg); sqrt/square/abs/recip are the pieces of
 * core/optimizer.py:70-77; sigmoid/tanh are fast paths for core/layers.py:79-80,88-89 */
enum { TNN_NEG = 0, TNN_EXP = 1, TNN_LOG = 2, TNN_SQRT = 3, TNN_SQUARE = 4, TNN_ABS = 5,
       TNN_RECIP = 6, TNN_SIGMOID = 7, TNN_TANH = 8, TNN_COPY = 9 };
/* reductions — core/ops.py:42,46,51,54 (un-broadcast sums), :226 (max), :235 (min), :253 (sum) */
enum { TNN_RSUM = 0, TNN_RMAX = 1, TNN_RMIN = 2 };
/* the other optimizers of core/optimizer.py:82-164 (tnn_optim_step) */
enum { TNN_OPT_MOMENTUM = 0, TNN_OPT_RMSPROP = 1, TNN_OPT_ADAGRAD = 2, TNN_OPT_ADADELTA = 3 };
/* activation codes for fused epilogues — core/layers.py:97-98 (ReLU = clip(x, 0.0)) */
enum { TNN_ACT_NONE = 0, TNN_ACT_RELU = 1 };

/* ------------------------------------------------------------------ runtime ------------------ */
TNN_API int tnn_init(int device);                 /* idempotent; creates the stream + pool         */
TNN_API int tnn_shutdown(void);
TNN_API const char* tnn_last_error(void);
TNN_API int tnn_backend_kind(void);               /* 1 = HIP gfx950 library, 2 = CPU test twin      */
TNN_API int tnn_device_props(int* cu_count, int* clock_khz, int64_t* hbm_bytes,
                             char* name, int name_len);

/* caching pool allocator (replaces numpy's malloc behind every op output, core/tensor.py:20) */
TNN_API int tnn_malloc(size_t bytes, void** out);
TNN_API int tnn_free(void* p);
TNN_API int tnn_pool_stats(int64_t* live_bytes, int64_t* cached_bytes, int64_t* device_allocs);
TNN_API int tnn_pool_trim(void);

TNN_API int tnn_memcpy_h2d(void* dst, const void* src, size_t bytes);  /* returns when src is reusable */
TNN_API int tnn_memcpy_d2h(void* dst, const void* src, size_t bytes);  /* synchronises the stream      */
TNN_API int tnn_memcpy_d2d(void* dst, const void* src, size_t bytes);
TNN_API int tnn_memset(void* dst, int byte, size_t bytes);
TNN_API int tnn_fill(void* dst, double value, int64_t n, int dtype);   /* np.zeros / np.ones_like */
TNN_API int tnn_stream_sync(void);

/* HIP events on the library stream (bench.py times kernels with these) */
TNN_API int tnn_event_create(void** ev);
TNN_API int tnn_event_record(void* ev);
TNN_API int tnn_event_elapsed_ms(void* start, void* stop, float* ms);  /* synchronises on stop */
TNN_API int tnn_event_destroy(void* ev);

/* hipGraph capture of everything enqueued between begin/end on the library stream.
 * Buffers allocated while capturing stay owned by the graph until tnn_graph_destroy. */
TNN_API int tnn_graph_capture_begin(void);
TNN_API int tnn_graph_capture_end(void** graph_exec);
TNN_API int tnn_graph_launch(void* graph_exec);
TNN_API int tnn_graph_destroy(void* graph_exec);

/* ------------------------------------------------------------------ GEMM (K1) ----------------- */
/* C[M,N] = alpha * op(A) * op(B) + beta * C.  op(A) is [M,K]: transA=0 -> A stored [M,K] (lda>=K),
 * transA=1 -> A stored [K,M] (lda>=M).  op(B) is [K,N]: transB=0 -> B stored [K,N], transB=1 ->
 * B stored [N,K].  Replaces core/ops.py:151 (A@B, NN), :157 (G@B.T, NT), :160 (A.T@G, TN) without
 * materialising a transpose.  f32: MFMA v_mfma_f32_32x32x2_f32, LDS-staged.  f64: VALU tiles. */
TNN_API int tnn_gemm(int transA, int transB, int64_t M, int64_t N, int64_t K, double alpha,
                     const void* A, int64_t lda, const void* B, int64_t ldb, double beta,
                     void* C, int64_t ldc, int dtype);

/* Fused epilogues (K8).  C = act(op(A)*op(B) + bias[N]).  relu_sign=1 additionally stores
 * negative pre-activations as -0.0f so the ReLU mask (x >= 0, core/ops.py:338) survives in the
 * sign bit of the output; used only by the whole-step trainer.  Replaces core/layers.py:49 + :98. */
TNN_API int tnn_gemm_bias_act(int transA, int transB, int64_t M, int64_t N, int64_t K,
                              const void* A, int64_t lda, const void* B, int64_t ldb,
                              const void* bias, int act, int relu_sign,
                              void* C, int64_t ldc, int dtype);
/* C = (op(A)*op(B)) * mask(Y) with mask = !signbit(Y[M,N]) (Y produced with relu_sign=1):
 * dX·[x>=0] of core/ops.py:157 + :342-343 in one pass. */
TNN_API int tnn_gemm_mask(int transA, int transB, int64_t M, int64_t N, int64_t K,
                          const void* A, int64_t lda, const void* B, int64_t ldb,
                          const void* Y, int64_t ldy, void* C, int64_t ldc, int dtype);

/* dW[M,N] = A^T G with A stored [K,M], G stored [K,N] (core/ops.py:159-160) and, in the same launch for
 * MNIST-size layers, db[N] = column-sum of G (the un-broadcast of the bias add, core/ops.py:52-54).
 * db may be NULL.  Large shapes run the GEMM and the column reduction as two launches. */
TNN_API int tnn_gemm_tn_colsum(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda,
                               const void* G, int64_t ldg, void* dW, int64_t ldc, void* db, int dtype);
/* dW[M,N] = A^T G as above, CONSUMED by Adam in the GEMM's epilogue (core/optimizer.py:67-79, the maths of tnn_adam; pows_f64
 * = {b1^t, b2^t} already advanced — tnn_adam_tick): p / m / v [M, N] (dense, ld = N) are updated in place; g_out [M, N]
 * receives the gradient itself when not NULL.  The fp32 product is MFMA-bound, so the optimizer's traffic rides under it
 * and the separate optimizer launch over the weights disappears from the single-GPU step.  Shapes the tiled kernel does
 * not take (and f64) run the two launches this replaces. */
TNN_API int tnn_gemm_tn_adam(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* G, int64_t ldg,
                             void* g_out, void* p, void* m, void* v, double lr, double b1, double b2, double eps,
                             const void* pows_f64, int dtype);
/* The same with the layer's BIAS in the launch (core/ops.py:52-54 + core/optimizer.py:67-79): db [N] = column sums of G
 * (may be NULL = tnn_gemm_tn_adam), produced by the workgroups of tile row 0 from the operand fragments they stream anyway,
 * and — when pb / mb / vb [N] are given — Adam applied to the bias right there: no column-reduction launches and no
 * optimizer launch for the bias.  Shapes the tiled kernel does not take run the launches this replaces. */
TNN_API int tnn_gemm_tn_adam_bias(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* G, int64_t ldg,
                                  void* g_out, void* p, void* m, void* v, void* db, void* pb, void* mb, void* vb, double lr,
                                  double b1, double b2, double eps, const void* pows_f64, int dtype);

/* Backward of one Dense layer y = x w + b given dz = dL/dy (all dense row-major):
 *   dw[n_in,n_out] = x^T dz (core/ops.py:159-160),  db[n_out] = column-sum dz (:52-54),
 *   dx[rows,n_in]  = (dz w^T) * !signbit(mask_src[rows,n_in])  (:156-157 + ReLU vjp :342-343); dx may be NULL.
 * One launch for MNIST-size layers, the three separate kernels otherwise. */
TNN_API int tnn_dense_bwd(int64_t rows, int64_t n_in, int64_t n_out, const void* x, const void* dz,
                          const void* w, void* dw, void* db, void* dx, const void* mask_src, int dtype);
/* Backward of the FIRST Dense layer (its input needs no gradient) with the whole Adam step folded in, for the
 * single-GPU training step: dw = x^T dz, db = column-sum dz as in tnn_dense_bwd, then Adam (tnn_adam maths, pows NOT
 * advanced here) on this layer's weights (p_w, m_w, v_w: [n_in, n_out]) and bias (p_b, m_b, v_b: [n_out]) from the
 * gradients just produced, and on one extra flat range of flat_n elements (every other layer's parameters, whose
 * gradients flat_g are already final).  MNIST-size layers: ONE launch — the optimizer costs no launch of its own;
 * other shapes / f64 run the launches this replaces.  dw may be NULL: the weight gradient is then consumed by Adam without
 * being stored (tnn_mlp_keep_grads(h, 0)). */
TNN_API int tnn_dense_bwd_first_adam(int64_t rows, int64_t n_in, int64_t n_out, const void* x, const void* dz, void* dw,
                                     void* db, void* p_w, void* m_w, void* v_w, void* p_b, void* m_b, void* v_b,
                                     void* flat_p, const void* flat_g, void* flat_m, void* flat_v, int64_t flat_n,
                                     double lr, double b1, double b2, double eps, const void* pows_f64, int dtype);
/* Backward of the FIRST Dense layer + all-reduce of the whole gradient arena + Adam, for the data-parallel step
 * (examples/mnist/run.py:82-83: the gradients are summed over the ranks before the optimizer sees them;
 * core/optimizer.py:67-79): dw / db as in tnn_dense_bwd land at grads[w_off ..] / grads[b_off ..], then
 * tnn_allreduce_adam(grads, n_reduce, ...) (pows NOT advanced; grads[scalar_index] -> *scalar_dst).  On the xGMI
 * peer-to-peer transport and MNIST-size layers (<= 256 rows per rank) ONE launch: the product's tiles are pushed straight into
 * the owning ranks' receive slots instead of being stored and read back.  Anything else: the two calls it replaces. */
TNN_API int tnn_dense_bwd_first_allreduce_adam(int64_t rows, int64_t n_in, int64_t n_out, const void* x, const void* dz,
                                               void* grads, int64_t n_reduce, int64_t w_off, int64_t b_off, void* p, void* m,
                                               void* v, int64_t n_params, double lr, double b1, double b2, double eps,
                                               const void* pows_f64, int64_t scalar_index, void* scalar_dst, int dtype);

/* ------------------------------------------------------------------ elementwise (K2,K3) ------- */
/* out[shape] = a (op) b with numpy broadcasting expressed as element strides (0 = broadcast dim).
 * ndim <= 6; out is dense row-major of `shape`. */
TNN_API int tnn_ewise_binary(int op, const void* a, const int64_t* stride_a,
                             const void* b, const int64_t* stride_b,
                             void* out, int ndim, const int64_t* shape, int dtype);
/* out = a (op) s  (scalar_lhs=0)  or  s (op) a  (scalar_lhs=1); s is a host scalar kernel argument
 * (Python numbers wrapped by as_tensor, core/tensor.py:7-10, never become device buffers). */
TNN_API int tnn_ewise_scalar(int op, const void* a, double s, int scalar_lhs,
                             void* out, int64_t n, int dtype);
/* out_u8 = a (cmp) b, same broadcasting contract */
TNN_API int tnn_ewise_compare(int cmp, const void* a, const int64_t* stride_a,
                              const void* b, const int64_t* stride_b,
                              void* out_u8, int ndim, const int64_t* shape, int dtype);
TNN_API int tnn_compare_scalar(int cmp, const void* a, double s, void* out_u8, int64_t n, int dtype);
TNN_API int tnn_ewise_unary(int op, const void* in, void* out, int64_t n, int dtype);
/* numpy ndarray.clip(min,max) with either bound optional — core/ops.py:334 */
TNN_API int tnn_clip(const void* in, int has_min, double vmin, int has_max, double vmax,
                     void* out, int64_t n, int dtype);
/* out = g * [(!has_min || x>=min) && (!has_max || x<=max)] — core/ops.py:336-343 (mask recomputed
 * from the saved input instead of being stored as a bool array) */
TNN_API int tnn_clip_bwd(const void* g, const void* x, int has_min, double vmin, int has_max,
                         double vmax, void* out, int64_t n, int dtype);
/* out = g where the SIGN BIT of y is clear, else 0: the ReLU vjp grad * [z >= 0] (core/ops.py:342-343) when y is a
 * fused Dense+ReLU output that keeps the mask in the sign bit of zero (z < 0 -> -0.0, z >= 0 -> |z|) */
TNN_API int tnn_mul_signmask(const void* g, const void* y, void* out, int64_t n, int dtype);
/* out = g * mask_u8 (grad * (a >= b) style vjps, core/ops.py:170,173,195,198,229,238) */
TNN_API int tnn_mul_mask(const void* g, const void* mask_u8, void* out, int64_t n, int dtype);
/* y += alpha * x — core/tensor.py:163 (self.grad += grad) and :66-68 (param += step) */
TNN_API int tnn_axpy(void* y, double alpha, const void* x, int64_t n, int dtype);
TNN_API int tnn_cast(const void* in, int in_dtype, void* out, int out_dtype, int64_t n);

/* ------------------------------------------------------------------ reductions (K4) ----------- */
/* in viewed as [outer, red, inner] -> out [outer, inner]; covers axis=None (1,n,1), axis=0 of a
 * matrix (1,R,C) = bias gradient, axis=1 (R,C,1).  Deterministic (no atomics). */
TNN_API int tnn_reduce(int rop, const void* in, void* out, int64_t outer, int64_t red,
                       int64_t inner, int dtype);
/* first-max index per row (np.argmax(x, axis=1), examples/mnist/run.py:89) -> int64 */
TNN_API int tnn_argmax_rows(const void* in, void* out_i64, int64_t rows, int64_t cols, int dtype);

/* ------------------------------------------------------------------ data movement (K5,K6) ----- */
/* out (dense, `shape`) = in gathered with element strides: N-d transpose (core/ops.py:269),
 * broadcast-back of the sum vjp (:257-263, stride 0), basic slices (:283) */
TNN_API int tnn_strided_copy(const void* in, const int64_t* in_stride, void* out, int ndim,
                             const int64_t* shape, int dtype);
/* out (strided view, `shape`) = in (dense): pad forward (core/ops.py:313) and getitem vjp (:286-288) */
TNN_API int tnn_strided_scatter(const void* in, void* out, const int64_t* out_stride, int ndim,
                                const int64_t* shape, int dtype);
/* out[i,:] = src[idx[i],:]  — utils/data_iterator.py:27-28 (inputs[idx]) */
TNN_API int tnn_gather_rows(const void* src, const void* idx_i64, void* out, int64_t n_idx,
                            int64_t row_elems, int64_t src_rows, int dtype);
/* dst[idx[i],:] = src[i,:]  — core/ops.py:287 (recover_grad[key] = grad) */
TNN_API int tnn_scatter_rows(const void* src, const void* idx_i64, void* dst, int64_t n_idx,
                             int64_t row_elems, int64_t dst_rows, int dtype);
/* onehot[i, labels[i]] = 1 — examples/mnist/run.py:27-28 (np.eye(n)[targets]) */
/* out[i] = *(T*)ptrs_u64[i], i < n: n scalars living in n separate device buffers gathered into one vector by ONE launch (the
 * per-step 0-d losses of a training loop, examples/mnist/run.py:84, read back once per epoch).  ptrs_u64: device array of n
 * device addresses.  dtype float32 / float64.  No reference counterpart (new). */
TNN_API int tnn_gather_scalars(const void* ptrs_u64, void* out, int64_t n, int dtype);
TNN_API int tnn_one_hot(const void* labels_i64, void* out, int64_t n, int64_t classes, int dtype);

/* ------------------------------------------------------------------ fused hot-path ops -------- */
/* y = act(x + bias[N]) for x [M,N] — core/layers.py:49 (+ b) and :98 */
TNN_API int tnn_bias_act(const void* x, const void* bias, int act, void* y, int64_t M, int64_t N,
                         int dtype);

/* Whole-batch softmax NLL, core/losses.py:24-32 (max and sum-exp are GLOBAL over [m,c]).
 * stats = device [2] {M = max z, S = sum exp(z - M)} of this shard.  A data-parallel caller merges
 * the shards' stats (tnn_lse_merge after an all-gather) before calling the backward. */
TNN_API int tnn_softmax_nll_stats(const void* z, int64_t m, int64_t c, void* stats, int dtype);
/* stats_all = [n_shards,2] -> stats = global {M, S} (log-sum-exp merge S = sum S_r e^{M_r-M}) */
TNN_API int tnn_lse_merge(const void* stats_all, int n_shards, void* stats, int dtype);
/* loss_out[0] = sum_i -log(sum_k p_ik y_ik) / m_global over THIS shard's rows,
 * dz = p - (e*y/q)/m_global with p = exp(z-M)/S, q_i = sum_k e_ik y_ik (= p - y/m for one-hot y).
 * dz may be NULL (loss only). */
TNN_API int tnn_softmax_nll_fwd_bwd(const void* z, const void* y, int64_t m, int64_t c,
                                    int64_t m_global, const void* stats, void* loss_out,
                                    void* dz, int dtype);

/* stats + loss + dz of one UNSHARDED batch in a single launch when m*c is small (falls back to the
 * three-kernel sequence otherwise); stats_out (device [2], may be NULL) receives {M, S}. */
TNN_API int tnn_softmax_nll_fused(const void* z, const void* y, int64_t m, int64_t c, void* stats_out,
                                  void* loss_out, void* dz, int dtype);
/* Data-parallel form of the same single launch (f32, shard fits one workgroup, peer-to-peer transport enabled):
 * m = this rank's rows, the softmax spans all m_global rows of all ranks; the kernel exchanges the shards'
 * {max, sum-exp} over xGMI itself (C2).  stats_out = GLOBAL {M, S}; loss_out = this rank's share of the loss
 * (the shares sum to the whole-batch loss, core/losses.py:30-32); dz uses 1/m_global. */
TNN_API int tnn_softmax_nll_fused_sharded(const void* z, const void* y, int64_t m, int64_t c, int64_t m_global,
                                          void* stats_out, void* loss_out, void* dz, int dtype);
/* The single-launch loss with everything a whole-step trainer hangs on it: sharded != 0 = the data-parallel form
 * above (f32), else m_global must equal m; adam_pows_f64 != NULL: thread 0 also advances Adam's {b1^t, b2^t}
 * (pows[0] *= b1, pows[1] *= b2) so the optimizer needs no prologue launch.  Requires m*c <= 4096 (f32) /
 * 2048 (f64) and m <= 1024 — it is an error otherwise (callers pick the multi-launch sequence themselves). */
TNN_API int tnn_softmax_nll_fused_tick(const void* z, const void* y, int64_t m, int64_t c, int64_t m_global,
                                       int sharded, void* stats_out, void* loss_out, void* dz, int dtype,
                                       void* adam_pows_f64, double b1, double b2);

/* Classifier head of an unsharded step in one launch (MNIST-size heads: n_classes <= 16, n_hidden % 16 == 0,
 * the activations fit in LDS; anything else runs as gemm_bias_act + softmax_nll_fused + dense_bwd):
 *   logits = a w + b (core/layers.py:49), whole-batch softmax NLL -> loss, dz (core/losses.py:24-32),
 *   dw = a^T dz, db = column-sum dz, da = (dz w^T) * !signbit(a)   (core/ops.py:156-160, :52-54, :342-343).
 * a is the previous layer's sign-encoded ReLU output; da may be NULL (single-layer net). */
TNN_API int tnn_mlp_head(int64_t rows, int64_t n_hidden, int64_t n_classes, const void* a, const void* w,
                         const void* b, const void* y, void* logits, void* dz, void* stats, void* loss,
                         void* dw, void* db, void* da, int dtype);

/* The same head as ONE MULTI-WORKGROUP launch for the single-GPU MNIST-size step (csrc/tnn_head.hip) — replaces the three
 * launches core/layers.py:49 (last Dense forward) | core/losses.py:24-32 (loss) | core/ops.py:156-160 (last Dense
 * backward): every workgroup obtains the logits and the loss statistics for itself, then produces its share of dw / da.
 * logit_partials != NULL: [n_hidden / 16][rows][n_classes] partial logits written by tnn_dense_fwd_head_partials (the
 * previous layer's tiles); they are only added up.  NULL: the workgroups compute a w themselves.
 * adam_pows_f64 != NULL: {b1^t, b2^t} are advanced here like tnn_softmax_nll_fused_tick does.  logits / dz / stats /
 * loss / da may be NULL.  Only shapes tnn_mlp_head_fits() accepts (f32, 10 classes, 128 hidden units, <= 128 rows). */
TNN_API int tnn_mlp_head_fits(int64_t rows, int64_t n_hidden, int64_t n_classes, int dtype, int* fits);
TNN_API int tnn_mlp_head_tick(int64_t rows, int64_t n_hidden, int64_t n_classes, const void* a, const void* w,
                              const void* b, const void* y, const void* logit_partials, void* logits, void* dz,
                              void* stats, void* loss, void* dw, void* db, void* da, int dtype, void* adam_pows_f64,
                              double b1, double b2);

/* The head above AND the backward of the hidden layer in front of it in ONE launch (4-launch step: csrc/tnn_head.hip,
 * mlp_head_bwd_kernel) — core/ops.py:156-160 for the last two Dense layers + core/ops.py:342-343 (ReLU mask) +
 * core/losses.py:24-32: the tiles of the hidden layer's backward derive their slice of its dz = (dz_head w^T) * [a >= 0]
 * themselves from the partial logits, so that dz is never written to memory.
 *   x [rows, n_in]: the hidden layer's input (sign-encoded ReLU output, it is also dx's mask source), w1 [n_in, n_hidden],
 *   a / w / b / y / logit_partials / logits / dz / stats / loss / dw / db as in tnn_mlp_head_tick (logit_partials required),
 *   dw1 [n_in, n_hidden], db1 [n_hidden], dx [rows, n_in] = (dz1 w1^T) * [x >= 0].
 * Shapes: tnn_mlp_head_bwd_fits(). */
/* tnn_mlp_head_bwd_fits: can tnn_mlp_head_bwd_tick take this head?  The tuned kernels take 128 hidden units x 10 classes; a
 * generic kernel takes any head with n_hidden %% 16 == 0, 16 <= n_hidden <= 256, n_classes <= 16 — both need f32,
 * rows <= 128 and n_in %% 16 == 0 (the reference's own 70 -> 30 -> 10 tail once the trainer has padded the hidden widths). */
TNN_API int tnn_mlp_head_bwd_fits(int64_t rows, int64_t n_in, int64_t n_hidden, int64_t n_classes, int dtype, int* fits);
/* Allocate, outside any hipGraph capture, the hand-off memory tnn_mlp_head_bwd_tick_ext uses to work on the 128-row blocks of
 * a batch in parallel (generic heads only; the trainer calls it when it is created, so that the first step may already be
 * inside a capture).  No reference counterpart (new). */
TNN_API int tnn_mlp_head_bwd_reserve(int64_t max_rows, int64_t n_in, int64_t n_hidden, int64_t n_classes);
TNN_API int tnn_mlp_head_bwd_tick(int64_t rows, int64_t n_in, int64_t n_hidden, int64_t n_classes, const void* x,
                                  const void* w1, const void* a, const void* w, const void* b, const void* y,
                                  const void* logit_partials, void* logits, void* dz, void* stats, void* loss, void* dw,
                                  void* db, void* dw1, void* db1, void* dx, int dtype, void* adam_pows_f64, double b1,
                                  double b2);
/* The data-parallel form (core/losses.py:26-27 — the softmax spans the GLOBAL batch of m_global rows, this rank holds
 * `rows` of them).  tnn_mlp_head_bwd_tick whose workgroups take the batch statistics from stats_pairs ([n_pairs][2] float32
 * {M_q, S_q}: every rank's pair, or one already merged pair — written by tnn_dense_fwd_head_partials_stats below [+ the
 * all-gather of the pairs on RCCL]) and do no cross-row reduction of their own; dz / dw / db / dw1 / db1 / dx are this
 * rank's contributions to the global gradients and *loss its share of the global loss — the all-reduce of the gradient
 * arena (tnn_allreduce_adam) sums both.  The data-parallel step is then forward x 2 | [all-gather] | head + hidden
 * backward | first-layer backward | all-reduce + Adam: 5 launches + the collectives, the same form on every transport.
 * rows <= 1024 here (nothing couples the rows inside the launch once the statistics come from memory: they are walked in
 * blocks of 128); with m_global == rows it is also the single-GPU step for batches of more than 128 rows.  n_pairs < 0:
 * logit_partials holds the WHOLE logits [rows, n_classes] without the bias and stats_pairs -n_pairs pairs
 * (tnn_dense_fwd_rows_head_stats). */
TNN_API int tnn_mlp_head_bwd_tick_ext(int64_t rows, int64_t m_global, int64_t n_in, int64_t n_hidden, int64_t n_classes,
                                      const void* x, const void* w1, const void* a, const void* w, const void* b,
                                      const void* y, const void* logit_partials, const void* stats_pairs, int n_pairs,
                                      void* logits, void* dz, void* stats, void* loss, void* dw, void* db, void* dw1,
                                      void* db1, void* dx, int dtype, void* adam_pows_f64, double b1, double b2);
/* The data-parallel form on the xGMI peer-to-peer transport with the statistics exchange DEFERRED into this launch
 * (core/losses.py:26-27; needs tnn_p2p_connect): the forward launch in front is issued with exchange = 2
 * (tnn_dense_fwd_head_partials_stats / tnn_dense_fwd_rows_head_stats_merged below) and has NO statistics tail; here every
 * workgroup reduces this SHARD's {max, sum-exp} itself — n_pairs = 0: from the partial logits, rows <= 128, exactly as
 * tnn_mlp_head_bwd_tick does on one GPU; n_pairs < 0: by merging the -n_pairs panel pairs in shard_pairs, logit_partials =
 * whole logits (row-panel forward, 128 hidden units x 10 classes, rows <= 1024) — ONE workgroup pushes the pair to every peer
 * (tagged 16-byte stores), and every workgroup merges the ranks' pairs, in rank order, from its own tagged slots.  Outputs as
 * tnn_mlp_head_bwd_tick_ext (contributions to the global gradients, this rank's share of the global loss).  A peer that never
 * sends: bounded wait, sticky failure word (tnn_p2p_status), the update behind the launch is discarded. */
/* tnn_mlp_head_bwd_xchg_fits: the shapes above, the transport enabled, and — because EVERY workgroup of that launch waits for
 * the peers — all ranks whose launches run on this GPU fit the device together (always true with one process per GPU). */
TNN_API int tnn_mlp_head_bwd_xchg_fits(int64_t rows, int64_t n_in, int64_t n_hidden, int64_t n_classes, int dtype, int* fits);
TNN_API int tnn_mlp_head_bwd_tick_xchg(int64_t rows, int64_t m_global, int64_t n_in, int64_t n_hidden, int64_t n_classes,
                                       const void* x, const void* w1, const void* a, const void* w, const void* b,
                                       const void* y, const void* logit_partials, const void* shard_pairs, int n_pairs,
                                       void* logits, void* dz, void* stats, void* loss, void* dw, void* db, void* dw1,
                                       void* db1, void* dx, int dtype, void* adam_pows_f64, double b1, double b2);
/* Forward of the hidden Dense layer in front of the classifier, C = act(A B + bias) like tnn_gemm_bias_act (NN form,
 * core/layers.py:49,98), which ALSO emits the next layer's logits as per-tile partial sums:
 *   head_z[tn][row][c] = sum_{col in [16 tn, 16 tn + 16)} C[row][col] * head_w[col][c]      (head_z: [ceil(N/16)][M][head_c])
 * — the first half of the classifier's forward (core/layers.py:49 of the NEXT layer) done where the activations are
 * still in registers.  One launch for MNIST-size layers; other shapes run the GEMM and a small second kernel. f32 only. */
TNN_API int tnn_dense_fwd_head_partials(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B,
                                        int64_t ldb, const void* bias, int act, int relu_sign, void* C, int64_t ldc,
                                        const void* head_w, int64_t head_c, void* head_z, int dtype);

/* The same launch in a data-parallel step: the workgroup that finishes LAST (agent-scope arrival counters in ticket_u32 —
 * 128 bytes of device memory, zero before the first call; every launch leaves the counters at zero — one counter up to
 * 128 rows; beyond, one per block of 128 rows plus one for the blocks' pairs, which are staged behind them) also reduces this shard's
 * whole-batch softmax statistics {max, sum-exp} (core/losses.py:25-27) from the partial logits + head_b (classifier bias
 * [head_c]) and writes them to out_pair_f32[2]; exchange != 0 (needs tnn_p2p_connect): it exchanges the pair with the
 * peers over xGMI and writes the MERGED pair instead.  y [M, head_c] (labels) rides along for symmetry with the head
 * kernels' staging.  No statistics launch and nobody waits for a peer inside the head launch that follows.
 * exchange == 2 (M <= 128; needs tnn_p2p_connect): the DEFERRED form — no statistics tail at all, ticket_u32 / out_pair_f32
 * untouched; the launch is tnn_dense_fwd_head_partials plus one thread that advances the sequence number tagging the pairs
 * tnn_mlp_head_bwd_tick_xchg (which must follow) exchanges.
 * f32, M <= 1024, N == 128, head_c == 10, 16-B aligned operands. */
TNN_API int tnn_dense_fwd_head_partials_stats(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B,
                                              int64_t ldb, const void* bias, int act, int relu_sign, void* C, int64_t ldc,
                                              const void* head_w, int64_t head_c, void* head_z, const void* head_b,
                                              const void* y, void* ticket_u32, void* out_pair_f32, int exchange, int dtype);

/* The hidden layer's forward in front of a classifier head for batches of more than 128 rows on ONE GPU, row-panel form
 * (core/layers.py:49,97-98 + the first half of core/losses.py:24-27): C = relu(A B + bias) as above, and because a
 * workgroup owns 16 whole rows it also finishes their logits and their softmax statistics:
 *   head_z_full [M, head_c] = C head_w            (WITHOUT head_b: tnn_mlp_head_bwd_tick_ext adds it, as it does to partials)
 *   pairs_f32 [ceil(M / 16)][2] = {max, sum-exp relative to it} of (C head_w + head_b) over each 16-row panel
 * to be handed to tnn_mlp_head_bwd_tick_ext as logit_partials / stats_pairs with n_pairs = -ceil(M / 16) (negative: whole
 * logits instead of H / 16 partial sums).  No arrival counter and no re-read of partial logits at the tail of the launch.
 * f32, M <= 1024, N == 128, head_c == 10, act == TNN_ACT_RELU, A 16-B aligned with lda and K multiples of 4. */
TNN_API int tnn_dense_fwd_rows_head_stats(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B,
                                          int64_t ldb, const void* bias, int act, int relu_sign, void* C, int64_t ldc,
                                          const void* head_w, int64_t head_c, void* head_z_full, const void* head_b,
                                          void* pairs_f32, int dtype);
/* ... whose workgroups also MERGE the panels' pairs inside the launch (the data-parallel step at more than 128 rows per rank):
 * the last workgroup to finish (arrival counter ticket_u32[0]: zero on entry, left zero) merges the ceil(M / 16) pairs in panel
 * order and leaves ONE pair in out_pair_f32 [2] — exchanged and merged with the other ranks' first when `exchange` is set
 * (xGMI peer-to-peer transport; tnn_dense_fwd_head_partials_stats's exchange).  head_z_full as above: hand it to
 * tnn_mlp_head_bwd_tick_ext with n_pairs = -1 (own merged pair) or -world (the all-gathered pairs).  core/losses.py:26-27.
 * exchange == 2 (needs tnn_p2p_connect): the DEFERRED form — the panels' pairs only (tnn_dense_fwd_rows_head_stats: no ticket,
 * no merge, out_pair_f32 untouched) plus the sequence advance for tnn_mlp_head_bwd_tick_xchg with n_pairs = -ceil(M / 16). */
TNN_API int tnn_dense_fwd_rows_head_stats_merged(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B,
                                                 int64_t ldb, const void* bias, int act, int relu_sign, void* C, int64_t ldc,
                                                 const void* head_w, int64_t head_c, void* head_z_full, const void* head_b,
                                                 void* pairs_f32, void* ticket_u32, void* out_pair_f32, int exchange, int dtype);

/* Sum-of-squares loss used by config C and test/test_autograd.py:119-121:
 * loss_out[0] = sum((pred - y)**2) / m_global over this shard, dpred = 2 (pred - y) / m_global
 * (the ops chain sub_ -> pow_(2) -> sum_ -> div_ of core/ops.py:61,121,252,93 in one pass). */
TNN_API int tnn_mse_fwd_bwd(const void* pred, const void* y, int64_t n, int64_t m_global,
                            void* loss_out, void* dpred, int dtype);
/* The same as the loss launch of a whole training step: loss_out2 (may be NULL) receives the loss too (the step's slot of a
 * loss history, examples/mnist/run.py:84 — no copy launch), and adam_pows_f64 != NULL: {b1^t, b2^t} *= {b1, b2} by one
 * thread of the launch (what tnn_adam_tick does as a launch of its own). */
TNN_API int tnn_mse_fwd_bwd_tick(const void* pred, const void* y, int64_t n, int64_t m_global, void* loss_out,
                                 void* loss_out2, void* dpred, int dtype, void* adam_pows_f64, double b1, double b2);

/* SGD: p += -lr * g — core/optimizer.py:46-47 + core/model.py:59-61 */
TNN_API int tnn_sgd(void* p, const void* g, int64_t n, double lr, int dtype);
/* Momentum / RMSProp / Adagrad / Adadelta `_compute_step` (core/optimizer.py:102-107, :121-123, :138-141,
 * :157-163) as one pass over the flat arena.  s1, s2 = the optimizer's state vectors, zero-initialised by the caller
 * (Momentum: acc, -; RMSProp: mean square, momentum; Adagrad: G, -; Adadelta: E[g^2], E[delta^2]); a, b =
 * (momentum, -), (decay, momentum), (-, -), (decay, -).  step_out (may be NULL) receives the step; p (may be NULL)
 * is updated in place, p += step (core/model.py:59-61). */
TNN_API int tnn_optim_step(int kind, void* p, const void* g, void* s1, void* s2, void* step_out, int64_t n,
                           double lr, double a, double b, double eps, int dtype);
/* Fused Adam on the flat arena, core/optimizer.py:67-79 + core/model.py:59-61:
 *   m += (1-b1)(g-m); v += (1-b2)(g*g-v); p += -lr*(m/(1-b1^t))/(sqrt(v/(1-b2^t))+eps)
 * pows = device double[4] {b1^(t-1), b2^(t-1), reserved, reserved}; initialise to {1, 1, 0, 0}.  The call
 * itself advances it on the device (a one-thread kernel multiplies in b1, b2 before the update reads it), so
 * that a captured hipGraph replays the right bias correction every step without a host-side step counter.
 * If step_out != NULL
 * the step is written there and p is left untouched (the reference's _compute_step contract). */
TNN_API int tnn_adam(void* p, const void* g, void* m, void* v, int64_t n, double lr, double b1,
                     double b2, double eps, void* pows_f64, void* step_out, int dtype);
/* tnn_adam with its one-thread prologue under the caller's control: advance = 0 when pows was already advanced for
 * this step (tnn_softmax_nll_fused_tick does it inside the loss kernel); scalar_src/scalar_dst (both or neither) copy
 * one scalar of `dtype` in that prologue — e.g. the loss of this step into a loss history (run.py:84).  With
 * advance = 0 and no scalar the prologue launch disappears. */
TNN_API int tnn_adam_ex(void* p, const void* g, void* m, void* v, int64_t n, double lr, double b1,
                        double b2, double eps, void* pows_f64, void* step_out, int dtype, int advance,
                        const void* scalar_src, void* scalar_dst);

/* ------------------------------------------------------------------ bf16 path (configs[4]) ---- */
/* bf16 storage, fp32 accumulation (v_mfma_f32_32x32x16_bf16), fp32 master weights + Adam state.  One GEMM
 * form: C[M,N] = A[M,K] * B[N,K]^T with BOTH operands K-contiguous — the bf16 trainer keeps a transposed weight
 * copy (forward), uses W itself for dX, and transposed activation copies for dW, so ops.dot_'s three products
 * (core/ops.py:151,157,160) all take this form.  K % 64 == 0, lda/ldb % 8 == 0, 16-B aligned bases.
 * Epilogue (exclusive): bias_f32[N] (+ReLU, optional sign-bit mask) or mask_y (bf16 [M,N], zero where its sign
 * bit is set).  c_dtype = TNN_BF16 or TNN_F32. */
TNN_API int tnn_gemm_bf16_nt(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B,
                             int64_t ldb, void* C, int64_t ldc, int c_dtype, const void* bias_f32, int act,
                             int relu_sign, const void* mask_y, int64_t ldy);
/* ... with a SECOND bf16 output: C_t [N, ldct] = the transpose of C (element (m, n) at C_t[n * ldct + m]) — the K-contiguous
 * operand the dW product of the backward pass needs for a^T and dz^T (core/ops.py:159-160), written by the epilogue that
 * already holds the finished tile in LDS instead of by a tnn_transpose_bf16 launch.  Output dtype is bf16.  (Shapes the
 * split-K kernel does not take fall back to GEMM + transpose: two launches, same bytes.) */
TNN_API int tnn_gemm_bf16_nt_t(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb,
                               void* C, int64_t ldc, const void* bias_f32, int act, int relu_sign, const void* mask_y,
                               int64_t ldy, void* C_t, int64_t ldct);
/* Allocate, outside any hipGraph capture, whatever hand-off memory tnn_gemm_bf16_nt / _nt_t needs for this shape (the
 * split-K kernel's slabs), so that the first call may be inside a capture and eager and captured steps run the same
 * kernel.  No reference counterpart (new). */
TNN_API int tnn_gemm_bf16_reserve(int64_t M, int64_t N, int64_t K);
TNN_API int tnn_transpose_bf16(const void* in, void* out, int64_t rows, int64_t cols);     /* [R,C] -> [C,R] */
/* two independent transposes in ONE launch: the K-contiguous operands a^T and dz^T of a dW product written just in front of it */
TNN_API int tnn_transpose2_bf16(const void* in1, void* out1, int64_t rows1, int64_t cols1, const void* in2, void* out2,
                                int64_t rows2, int64_t cols2);
TNN_API int tnn_cast_bf16(const void* in, void* out, int64_t n, int to_bf16);              /* f32 <-> bf16 (RNE) */
TNN_API int tnn_colsum_bf16(const void* in, void* out_f32, int64_t rows, int64_t cols);    /* bias gradient */
TNN_API int tnn_mse_bf16(const void* pred, const void* y, int64_t n, int64_t m_global, void* loss_out_f32,
                         void* dpred);
/* ... as the loss launch of a whole step: loss_out2 (may be NULL) receives the loss too (no copy launch) and, with
 * adam_pows_f64 != NULL, one thread advances {b1^t, b2^t} (tnn_adam_tick without its launch). */
TNN_API int tnn_mse_bf16_tick(const void* pred, const void* y, int64_t n, int64_t m_global, void* loss_out_f32,
                              void* loss_out2_f32, void* dpred, void* adam_pows_f64, double b1, double b2);
/* The loss launch of a whole bf16 step, with the backward pass's operand preparation folded in (ONE launch instead of
 * tnn_mse_bf16_tick's two + two tnn_transpose_bf16): loss = sum((pred - y)^2) / m_global to loss_out_f32 (and loss_out2_f32
 * when not NULL), dpred = 2 (pred - y) / m_global [rows, cols] AND its transpose dpred_t [cols, rows] (NULL = skip), x_t
 * [x_cols, rows] = the transpose of the batch x [rows, x_cols] (x / x_t NULL = skip), Adam's {b1^t, b2^t} advanced when
 * adam_pows_f64 != NULL.  rows, cols, x_cols multiples of 64.  partials_f64: workspace of rows / 64 * cols / 64 doubles;
 * ticket_u32: 64 words that are ZERO on entry and are left zero (two-level arrival counters of the loss reduction).
 * core/losses.py (sum-of-squares form), core/ops.py:159-160 for the transposed operands. */
TNN_API int tnn_mse_bf16_prep(const void* pred, const void* y, int64_t rows, int64_t cols, int64_t m_global,
                              void* loss_out_f32, void* loss_out2_f32, void* dpred, void* dpred_t, const void* x,
                              int64_t x_cols, void* x_t, void* partials_f64, void* ticket_u32, void* adam_pows_f64,
                              double b1, double b2);
/* The bias of one bf16 Dense layer in ONE launch: db_f32 [cols] = column sums of dz (bf16 [rows, cols], core/ops.py:52-54)
 * and, when p_master / m / v [cols] are given, Adam on the fp32 master bias (core/optimizer.py:67-79, pows already advanced)
 * + its bf16 copy w_bf16 (may be NULL).  Replaces tnn_colsum_bf16 (two launches at this size) + tnn_adam_master_bf16_2d. */
TNN_API int tnn_bias_bf16_adam(const void* dz, int64_t rows, int64_t cols, void* db_f32, void* p_master, void* m, void* v,
                               void* w_bf16, double lr, double b1, double b2, double eps, const void* pows_f64);
/* Adam on the fp32 master parameters (same maths as tnn_adam) that also refreshes the bf16 working copy */
TNN_API int tnn_adam_master_bf16(void* p_master, const void* g, void* m, void* v, void* w_bf16, int64_t n,
                                 double lr, double b1, double b2, double eps, void* pows_f64);
/* The same update on one [rows, cols] weight matrix that ALSO writes the transposed bf16 copy wT_bf16
 * [cols, rows] the forward GEMM reads (NULL = skip it), so the refreshed weights are not re-read by a
 * separate transpose.  advance != 0 advances the beta powers first (do it once per optimizer step). */
TNN_API int tnn_adam_master_bf16_2d(void* p_master, const void* g, void* m, void* v, void* w_bf16, void* wT_bf16,
                                    int64_t rows, int64_t cols, double lr, double b1, double b2, double eps,
                                    void* pows_f64, int advance);
/* dW = A B^T (A [M, K], B [N, K], bf16, K-contiguous, fp32 accumulation — core/ops.py:160 with the transposed copies of
 * tnn_transpose_bf16 as operands) CONSUMED by Adam in the GEMM epilogue (core/optimizer.py:67-79, the maths of
 * tnn_adam_master_bf16_2d): p / m / v [M, N] fp32 updated in place, w_bf16 [M, N] and wT_bf16 [N, M] refreshed (either
 * may be NULL = skip: the first layer's [in, out] copy has no reader — no dX is formed for the input).  g_out_f32 [M, N] receives the gradient itself when not NULL; NULL saves its 4 B write and the 4 B re-read
 * of the separate optimizer launch per parameter (of 36).  pows_f64 = {b1^t, b2^t} ALREADY advanced (tnn_adam_tick).
 * Single-GPU step only: a data-parallel step reduces the gradients between the two. */
TNN_API int tnn_gemm_bf16_nt_adam(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb,
                                  void* g_out_f32, void* p_master, void* m, void* v, void* w_bf16, void* wT_bf16,
                                  double lr, double b1, double b2, double eps, const void* pows_f64);
/* tnn_bias_bf16_adam for SEVERAL layers in one launch (the last launch of the single-GPU bf16 step): arrays of n_layers
 * pointers / column counts, dz[l] bf16 [rows, cols[l]]; p_master / m / v / w_bf16 arrays may be NULL as a whole (gradients only)
 * and w_bf16[l] individually.  Layer by layer the same summation order as tnn_bias_bf16_adam.  n_layers <= 16. */
TNN_API int tnn_bias_bf16_adam_multi(int n_layers, const void* const* dz, int64_t rows, const int64_t* cols,
                                     void* const* db_f32, void* const* p_master, void* const* m, void* const* v,
                                     void* const* w_bf16, double lr, double b1, double b2, double eps, const void* pows_f64);
/* Adam on a flat slice of the fp32 master parameters whose gradient arrives as bf16 (the reduce-scattered slice of the
 * sharded-optimizer step): p / m / v fp32 updated in place, the slice's bf16 working copy refreshed.  pows_f64 already
 * advanced for this step. */
TNN_API int tnn_adam_master_g16(void* p_master, const void* g_bf16, void* m, void* v, void* w_bf16, int64_t n, double lr,
                                double b1, double b2, double eps, const void* pows_f64);
/* {b1^t, b2^t} *= {b1, b2}: the once-per-step advance of Adam's bias-correction state as a launch of its own */
TNN_API int tnn_adam_tick(void* pows_f64, double b1, double b2);

/* ------------------------------------------------------------------ whole-step MLP trainer ---- */
/* One object = Dense/ReLU stack + whole-batch softmax NLL (loss_kind 0) or sum-of-squares/m
 * (loss_kind 1, the (err**2).sum()/m of test/test_autograd.py:119-121) + SGD (opt 0) / Adam (opt 1) /
 * Momentum, RMSProp, Adagrad, Adadelta (opt 2 + TNN_OPT_*; their hyper-parameters a, b travel in b1, b2),
 * i.e. the loop body of examples/mnist/run.py:79-83 as 4 launches (3-layer net, class head that fits tnn_mlp_head_fits; 7 otherwise) on device-resident state:
 * params | grads | m | v live in one flat arena each, ordered layer by layer, "w" then "b"
 * (core/layers.py:35, core/optimizer.py:14-15).
 * dtype TNN_BF16 (configs[4]): x, y, activations are bf16; the arenas stay fp32 (master weights, gradients,
 * Adam state); loss_kind 1 + Adam only; call tnn_mlp_sync_params after writing the parameter arena. */
TNN_API int tnn_mlp_create(int n_layers, const int64_t* widths, int64_t max_rows, int loss_kind,
                           int opt_kind, double lr, double b1, double b2, double eps, int dtype,
                           void** handle);
TNN_API int tnn_mlp_destroy(void* handle);
TNN_API int tnn_mlp_arena(void* handle, void** params, void** grads, void** m, void** v,
                          int64_t* n_params);
TNN_API int tnn_mlp_param_offset(void* handle, int layer, int which, int64_t* offset, int64_t* count);
/* the rest of the optimizer state for checkpoints: device double[4] {b1^t, b2^t, -, -} of Adam (see tnn_adam);
 * together with the params / m / v arenas it is everything a resumed run needs */
TNN_API int tnn_mlp_optimizer_state(void* handle, void** pows_f64);
/* logits[rows, widths[n]] = net(x[rows, widths[0]]) */
TNN_API int tnn_mlp_forward(void* handle, const void* x, int64_t rows, void* logits);
/* forward + loss stats of this shard (phase 1); stats = device [2] */
TNN_API int tnn_mlp_forward_stats(void* handle, const void* x, int64_t rows, void* stats);
/* loss + full backward into the grad arena (phase 2); y = targets [rows, widths[n]] */
TNN_API int tnn_mlp_backward(void* handle, const void* x, const void* y, int64_t rows,
                             int64_t m_global, const void* stats, void* loss_out);
/* optimizer update from the grad arena (phase 3) */
TNN_API int tnn_mlp_update(void* handle);
/* phases 1-3 back to back for the single-GPU case; loss_out = device scalar (may be NULL) */
TNN_API int tnn_mlp_step(void* handle, const void* x, const void* y, int64_t rows, void* loss_out);
/* one data-parallel step through the communicator of tnn_comm_init: forward + shard stats, all-gather + merge,
 * loss/backward with the global batch size, all-reduce of the gradient arena (+ loss slot), update */
TNN_API int tnn_mlp_step_sharded(void* handle, const void* x, const void* y, int64_t rows, void* loss_out);
/* Measurement hook (bench.py's per-launch timings): restrict tnn_mlp_step to its primitive calls number
 * [first, first + count) in issue order (count < 0: the whole step again); *calls_in_last_step = primitive calls (= kernel
 * launches for the MNIST-size step) the last tnn_mlp_step went through.  No reference counterpart (new). */
TNN_API int tnn_mlp_launch_window(void* handle, int first, int count, int* calls_in_last_step);
/* keep != 0 (default): after tnn_mlp_step every gradient is in the gradient arena (tnn_mlp_arenas).  keep == 0: a step
 * may consume weight gradients where they are produced (bf16 trainer: tnn_gemm_bf16_nt_adam) without storing them —
 * bias gradients and the loss are still written. */
TNN_API int tnn_mlp_keep_grads(void* handle, int keep);
/* after the parameter arena was written from outside (initial weights): refresh derived copies — the bf16
 * working copies W, W^T of a TNN_BF16 trainer; no-op for f32 / f64 */
TNN_API int tnn_mlp_sync_params(void* handle);
/* bf16 trainer: the bf16 working copy of the whole parameter arena (arena order, element offsets of
 * tnn_mlp_param_offset).  In the data-parallel sharded-optimizer step it is the complete, rank-identical copy of the
 * weights (each rank's fp32 master arena is authoritative for its own rows only). */
TNN_API int tnn_mlp_bf16_weights(void* handle, void** w_bf16);
/* After sharded-optimizer steps at world > 1 (tnn_mlp_step_sharded on a TNN_BF16 trainer) each rank's fp32 master / m / v
 * arenas are current for its own row slice of every weight matrix only.  tnn_mlp_masters_sharded: *world = the world size
 * they are sharded over, 0 when whole.  tnn_mlp_gather_masters (COLLECTIVE: every rank calls it) all-gathers the owned
 * slices so the arenas are whole on every rank — what reads the parameters the way core/model.py:24-33 exposes them, or a
 * checkpoint, needs first.  No-ops for other trainers.  tnn_mlp_sync_params also marks the arenas whole (the caller has
 * just written them). */
TNN_API int tnn_mlp_masters_sharded(void* handle, int* world);
TNN_API int tnn_mlp_gather_masters(void* handle);
/* intermediate activations for parity tests: layer l output [rows, widths[l+1]] */
TNN_API int tnn_mlp_activation(void* handle, int layer, void** ptr);

/* ------------------------------------------------------------------ RCCL over xGMI (C1, C2) --- */
/* New relative to the reference (it has no communication).  One process per GPU. */
TNN_API int tnn_comm_unique_id(void* id128);                /* rank 0: ncclGetUniqueId (128 bytes) */
TNN_API int tnn_comm_init(int rank, int world, const void* id128);
TNN_API int tnn_comm_destroy(void);
TNN_API int tnn_comm_world(int* rank, int* world);
/* in-place all-reduce on the library stream; rop = TNN_RSUM / TNN_RMAX */
TNN_API int tnn_allreduce(void* buf, int64_t n, int dtype, int rop);
TNN_API int tnn_allgather(const void* send, void* recv, int64_t n_per_rank, int dtype);
/* Bucketed, overlapped C1 for large arenas: the SUM all-reduce of one gradient bucket runs on a communication
 * stream, ordered after everything enqueued so far on the library stream, while the library stream continues (e.g.
 * with the next layer's backward).  tnn_comm_join makes the library stream wait for every outstanding bucket —
 * call it before anything reads the buckets (the optimizer).  Small messages the peer-to-peer path carries, or a
 * missing RCCL communicator, make it the ordinary tnn_allreduce. */
TNN_API int tnn_allreduce_async(void* buf, int64_t n, int dtype, int rop);
TNN_API int tnn_comm_join(void);
/* wait only for the OLDEST outstanding bucket (buckets complete in issue order): lets the optimizer start on the
 * last layer's parameters while the earlier layers' buckets are still on the links */
TNN_API int tnn_comm_wait_oldest(void);
/* Sharded-optimizer exchange for arenas that are bandwidth problems (configs[4]: 268 M parameters).  New relative to
 * the reference (no communication there); replaces "all-reduce the gradient, every rank runs the same Adam"
 * (run.py:82-83) by: reduce-scatter the gradient (rank r receives the SUM of slice r), Adam on the OWNED slice only
 * (core/optimizer.py:67-79), all-gather of the refreshed bf16 weights.  Same bytes on the links as one all-reduce of
 * the wire dtype, optimizer traffic divided by the world size.
 * tnn_reduce_scatter: recv[0:n] <- SUM over ranks of send_r[rank*n : (rank+1)*n]; recv may be send + rank*n (in place).
 * dtype TNN_F32 / TNN_F64 / TNN_BF16.  World 1 without a communicator: a copy. */
TNN_API int tnn_reduce_scatter(const void* send, void* recv, int64_t n_per_rank, int dtype);
/* tnn_comm_chain_begin .. tnn_comm_chain_end: every library call in between is enqueued on the COMMUNICATION stream
 * instead of the library stream, ordered behind everything enqueued on the library stream so far; the library stream
 * itself continues (the next layer's backward).  chain_end files one "done" event that tnn_comm_wait_oldest /
 * tnn_comm_join wait for, like a bucket of tnn_allreduce_async.  Without an RCCL communicator (world 1) the chain runs
 * inline on the library stream. */
TNN_API int tnn_comm_chain_begin(void);
TNN_API int tnn_comm_chain_end(void);
/* C1 and the optimizer in one call (run.py:82-83 with the exchange in between): grads[0:n_reduce] <- SUM over ranks,
 * then tnn_adam_ex(p, grads, m, v, n_params <= n_reduce, ..., advance, scalar_src = grads + scalar_index, scalar_dst).
 * On the peer-to-peer transport (f32, advance == 0) the update is applied by the all-reduce kernel's last stage, so
 * the reduced gradient is consumed from registers and no optimizer launch follows. */
TNN_API int tnn_allreduce_adam(void* grads, int64_t n_reduce, void* p, void* m, void* v, int64_t n_params, double lr,
                               double b1, double b2, double eps, void* pows_f64, int advance, int dtype,
                               int64_t scalar_index, void* scalar_dst);

/* ------------------------------------------------ xGMI peer-to-peer transport under C1 / C2 --- */
/* Latency path for small messages (the 0.94 MB MNIST gradient arena, the {max, sum-exp} pairs): every rank
 * creates one uncached region and exports it (64-byte hipIpcMemHandle), the host side exchanges the handles
 * (torch.distributed/gloo in dist.py), every rank maps all peers.  From then on tnn_allreduce (f32 SUM, up to
 * max_bytes) and tnn_allgather (<= 256 B per rank) are single kernels of posted peer stores and flag barriers on
 * the library stream, hipGraph-capturable, bit-identical on every rank; anything else still goes to RCCL.
 * rank/world here also serve tnn_comm_world when no RCCL communicator exists. */
/* Before tnn_p2p_create: reserve `slot_bytes` (a multiple of 4096; 0 = none, the default) of staging per (parity, source rank)
 * in the region the NEXT group creates — 2 x world x slot_bytes in all.  With it, and without an RCCL communicator
 * (tnn_comm_init), tnn_reduce_scatter (bf16 / f32 sums: fp32 accumulation in rank order, one rounding) and tnn_allgather of
 * any size go over the mapped regions (direct exchange, a flag barrier per workgroup, messages beyond a slot in chunks) — the
 * sharded-optimizer step of configs[4] (tnn_mlp_step_sharded on a bf16 trainer) then runs on a peer-to-peer-only group, e.g.
 * several ranks on ONE GPU, which RCCL refuses.  No reference counterpart (new). */
TNN_API int tnn_p2p_set_bulk_bytes(int64_t slot_bytes);
TNN_API int tnn_p2p_create(int rank, int world, int64_t max_bytes, void* handle64_out);
TNN_API int tnn_p2p_connect(const void* handles /* world x 64 bytes, rank order */);
/* workgroups of the all-reduce kernel (0 = pick from the message size; TNN_P2P_BLOCKS sets the initial value).
 * Collective: every rank must set the same value before the next all-reduce. */
TNN_API int tnn_p2p_tune(int allreduce_blocks);
/* Start-up check of the deferred statistics exchange (tnn_mlp_head_bwd_tick_xchg): a 64-workgroup launch in which one workgroup
 * pushes {m_mine, s_mine} to every peer and EVERY workgroup merges the ranks' pairs from its own tagged slots, as the head launch
 * of a data-parallel step does; out_pairs_f32 (device, [64][2] float32) receives what each workgroup ended up with.  Collective. */
TNN_API int tnn_p2p_xchg_selftest(double m_mine, double s_mine, void* out_pairs_f32);
TNN_API int tnn_p2p_enable(int on);                         /* route eligible collectives here (default after connect) */
/* dead != 0: a barrier timed out (TNN_P2P_TIMEOUT_MS, default 20000) - results since then are invalid; synchronises */
TNN_API int tnn_p2p_status(int* connected, int* enabled, int* dead);
/* *failed = 1 once a peer barrier has timed out (host-pinned mirror of the sticky device word: NO stream
 * synchronisation, cheap enough to call before every step / graph replay).  From that moment every collective on the
 * transport — the one that timed out, later launches, captured graph replays — is discarded on the device (buffers,
 * parameters and optimizer state keep their previous contents) and every host call into it returns an error. */
TNN_API int tnn_p2p_poll_failed(int* failed);
/* on != 0: optimizer-update kernels launched from now on become no-ops once the transport's dead word is set (see
 * above); on == 0: back to unconditional updates.  tnn_mlp_step_sharded brackets itself with the pair. */
/* Diagnostics of a timed-out wait, read from the host-pinned mirror without a stream sync: words16[0] = 0, or which
 * wait gave up first (1 collective flag barrier, 2 statistics exchange), [1] the value
 * it expected, [2] the last value it saw, [3] peer / workgroup, [4] flag row / slot parity. */
TNN_API int tnn_p2p_debug(int* words16);
TNN_API int tnn_p2p_guard_updates(int on);
TNN_API int tnn_p2p_destroy(void);

#ifdef __cplusplus
}
#endif
#endif /* TNN_HIP_H_ */
