/*
 * mi355x_recsys.h — C-ABI of libmi355x_recsys.so
 *
 * Hand-written gfx950 (MI355X / CDNA4) HIP kernels for the embedding-lookup +
 * feature-interaction hot path of chenxing1999/recsys-benchmark.
 *
 * Boundary conventions:
 *   - every pointer is a DEVICE pointer unless the parameter says "host";
 *   - the caller owns every buffer; the library allocates no persistent memory;
 *   - every entry point enqueues on `stream` (a hipStream_t passed as void*)
 *     and returns without synchronising; no hidden syncs, graph-capture safe;
 *   - an entry point enqueues its kernel(s) and NOTHING ELSE, with five documented exceptions that also enqueue one
 *     hipMemsetAsync on the same stream (a memset node when captured): mi_slot_fm_bwd zero-fills gbuf (unused slots must
 *     read as zero gradients), mi_bpr_fwd / mi_rowsq_fwd / mi_lse_diag_fwd zero the 4-byte arrival ticket at the end
 *     of their workspace, mi_tt_plan_level zeroes its p digit counters;
 *   - return value: MI_OK (0) or a negative MI_ERR_* code; no exceptions cross
 *     the ABI; calls are re-entrant;
 *   - `err` (nullable) is a device int32 word: kernels OR a bit into it when an
 *     index is out of range (the lookup then yields zeros and touches no memory
 *     out of bounds).  The reference raises IndexError from nn.Embedding on CPU
 *     (src/models/embeddings/base.py:74-75); the Python host side turns a
 *     non-zero word into IndexError when asked to check.
 *   - indices are int64 exactly as the reference passes them
 *     (src/models/deepfm.py:88); rows are fp32 row-major.
 *
 * Each declaration cites the reference interface (file:line under
 * /root/reference) whose arithmetic it replaces.
 */
#ifndef MI355X_RECSYS_H
#define MI355X_RECSYS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_API __attribute__((visibility("default")))

/* 2: round 3 — mi_sort_field_rows / mi_bpr_* / mi_rowsq_* had gained arguments under version 1 (callers built against
 * that header must rebuild); mi_gather_fm_fwd_ld, mi_sparse_adam_sorted's row stride and the later additions are new. */
/* 3: round 4 — mi_tail_bn_fwd gained `shift` and `nrep`, mi_tail_bn_bwd `dbias` and `affine` (a caller built against version 2 passes shorter structs). */
#define MI_ABI_VERSION 3

#define MI_OK 0
#define MI_ERR_INVALID_ARG (-1)  /* null pointer, negative size, bad enum */
#define MI_ERR_UNSUPPORTED (-2)  /* shape outside what the kernels cover */
#define MI_ERR_LAUNCH (-3)       /* hipLaunch / hipGetLastError failed */
#define MI_ERR_STATE (-4)        /* profiling ring misuse */

/* bits OR-ed into *err by the kernels */
#define MI_IDX_OUT_OF_RANGE 1
#define MI_IDX_OUT_OF_FIELD 2    /* mi_sort_field_rows: id inside [0,N) but outside its own field's range */

/* ---- library info -------------------------------------------------------- */
MI_API int mi_abi_version(void);
MI_API const char *mi_strerror(int code);

/* ---- a1-a4: DeepFM gather + FM 2nd order + first-order term ---------------
 * Replaces, in ONE kernel (src/models/deepfm.py:88-98):
 *   x = x + offsets                      (:88)
 *   emb = nn.Embedding(x)                (:89 -> embeddings/base.py:74-75)
 *   0.5*sum_d[(sum_f e)^2 - sum_f e^2]   (:91-92,98)
 *   EmbeddingBag(N,1,"sum")(x) + bias    (:95)
 * idx      int64[B,F] raw per-field ids (no offsets added)
 * offsets  int64[F]
 * W        fp32[N,D], w1 fp32[N] (fc.weight viewed flat), bias fp32[1] (nullable)
 * emb_out  fp32[B,F,D]   rows_out int64[B,F] (= idx+offsets; nullable)
 * yfm_out  fp32[B]
 */
MI_API int mi_gather_fm_fwd(const int64_t *idx, const int64_t *offsets,
                            const float *W, const float *w1, const float *bias,
                            float *emb_out, float *yfm_out, int64_t *rows_out,
                            int64_t B, int32_t F, int32_t D, int64_t N,
                            int32_t *err, void *stream);

/* The same lookup over tables with a ROW STRIDE (src/models/deepfm.py:47-51 declares the two tables; this is their
 * storage, not their arithmetic): row n of W starts at W + n*ldw floats (ldw >= D, a multiple of 4), the first-order
 * weight of row n is w1[n*ldw1].  (ldw, ldw1) = (D, 1) is mi_gather_fm_fwd.  (32, 32) with w1 = W + 16 reads ONE packed
 * table fp32[N,32] = {16 embedding floats, first-order weight, padding}: a lookup then touches one 128-byte line
 * instead of two unrelated 64-byte sectors (DeepFM.pack_tables() on the Python side; both parameters become views of
 * the packed buffer and state_dict() still emits the reference's two tensors). */
MI_API int mi_gather_fm_fwd_ld(const int64_t *idx, const int64_t *offsets,
                               const float *W, int64_t ldw, const float *w1, int64_t ldw1, const float *bias,
                               float *emb_out, float *yfm_out, int64_t *rows_out,
                               int64_t B, int32_t F, int32_t D, int64_t N,
                               int32_t *err, void *stream);

/* The same forward, additionally storing sum_out[b, :] = sum_f emb[b, f, :] (fp32[B, D], nullable) — the one quantity of the
 * FM backward (src/models/deepfm.py:91-92 through autograd: dE_bf = g_b (S_b - e_bf)) that a kernel seeing only SOME fields
 * of a sample cannot re-derive; mi_tail_dgrad_gemm_fm (below) consumes it. */
MI_API int mi_gather_fm_fwd_sum(const int64_t *idx, const int64_t *offsets,
                                const float *W, int64_t ldw, const float *w1, int64_t ldw1, const float *bias,
                                float *emb_out, float *yfm_out, int64_t *rows_out, float *sum_out,
                                int64_t B, int32_t F, int32_t D, int64_t N,
                                int32_t *err, void *stream);

/* Touch the table rows the NEXT batch's lookup will read (idx int64[B,F] raw ids of that batch, other arguments as
 * mi_gather_fm_fwd_ld): pulls their 128-byte lines into the 256 MiB Infinity Cache.  Meant for a side stream under an
 * MFMA-bound kernel of the current step; writes nothing.  (No reference counterpart: src/trainer/deepfm.py:40-60 gets the
 * next batch from its DataLoader while the step runs; this is what the device does with that lead.) */
MI_API int mi_prefetch_rows(const int64_t *idx, const int64_t *offsets, const float *W, int64_t ldw, const float *w1,
                            int64_t ldw1, int64_t B, int32_t F, int64_t N, void *stream);
/* The same job as a host struct, for a launch that carries it in extra workgroups (mi_gemm_f32_multi_ride). */
typedef struct mi_prefetch_rows_job {
  const int64_t *idx;       /* device int64[B, F]: the NEXT batch's raw ids */
  const int64_t *offsets;   /* device int64[F], nullable */
  const float *W;           /* the embedding table (row pitch ldw floats) */
  const float *w1;          /* the first-order table (row pitch ldw1), nullable */
  int64_t ldw, ldw1, B, N;
  int32_t F;
} mi_prefetch_rows_job;

/* Backward of the above, row-sparse form (the MI355X-native default):
 *   gvals[b,f,:] = g_emb[b,f,:] + g_y[b] * (S_b - emb[b,f,:]),  S_b = sum_f emb
 *   g1vals[b,f]  = g_y[b]
 * i.e. one gradient row per lookup, written with plain coalesced stores; the
 * caller pairs them with rows_out from the forward as an (uncoalesced) COO
 * gradient of W / w1.  Autograd equivalent: embedding_dense_backward +
 * _embedding_bag_dense_backward of src/models/deepfm.py:89,95 before the
 * duplicate-row sum.  g_emb is nullable (no deep branch).
  * gbias (nullable, both backward forms and mi_slot_fm_bwd): receives sum_b g_y[b], the gradient of the
 * scalar bias (src/models/deepfm.py:51,95), added up in a fixed order by workgroup 0 of the same launch.
 */
MI_API int mi_gather_fm_bwd_rows(const float *emb, const float *g_y,
                                 const float *g_emb, float *gvals, float *g1vals,
                                 float *gbias, int64_t B, int32_t F, int32_t D, void *stream);

/* Backward, dense form (the reference's weight.grad semantics): scatter-adds
 * the same rows into caller-zeroed gW fp32[N,D] / gw1 fp32[N] with float
 * atomics (sum order differs from the CPU index_add: fp32 tolerance).
 * rows int64[B,F] = idx+offsets as produced by the forward.
 */
MI_API int mi_gather_fm_bwd_dense(const int64_t *rows, const float *emb,
                                  const float *g_y, const float *g_emb,
                                  float *gW, float *gw1, float *gbias, int64_t B, int32_t F,
                                  int32_t D, int64_t N, void *stream);

/* ---- a2: plain row gather (IEmbedding.forward on a vanilla table) ----------
 * src/models/embeddings/base.py:74-75 (nn.Embedding on int64[n] ids, already
 * offset).  out fp32[n,D].
 */
MI_API int mi_gather_rows_fwd(const int64_t *idx, const float *W, float *out,
                              int64_t n, int32_t D, int64_t N, int32_t *err,
                              void *stream);

/* dense backward of the row gather: gW[idx[i],:] += g[i,:] (float atomics). */
MI_API int mi_scatter_add_rows(const int64_t *idx, const float *g, float *gW,
                               int64_t n, int32_t D, int64_t N, void *stream);

/* ---- a3/a4 over an already gathered emb (DeepFM on a compressed table) -------
 * y_fm[b] = 0.5*sum_d[(sum_f e)^2 - sum_f e^2] + sum_f w1[rows[b,f]] + bias
 * (src/models/deepfm.py:91-98).  Its backward is mi_gather_fm_bwd_rows (gradient
 * w.r.t. emb) plus mi_scatter_add_rows with D=1 (dense w1 gradient).
 */
MI_API int mi_fm_fwd(const float *emb, const int64_t *rows, const float *w1,
                     const float *bias, float *yfm, int64_t B, int32_t F, int32_t D,
                     int64_t N, int32_t *err, void *stream);

/* ---- a6-a8: two-table compositional lookups, index math fused ---------------
 *   i1 = idx % mod1  -> row of T1 fp32[n1,De]     i2 = idx / div2 -> row of T2 fp32[n2,De]
 *   out = T1'[i1] (op) T2'[i2],  op: 0 mult, 1 add, 2 cat
 *   xform 0: T' = T
 *         1: T' = sign(T)*relu(|T| - sigmoid(S))   (S1,S2 fp32 threshold logits)
 *         2: T' = T * M                            (M1,M2 bool masks, 1 byte each)
 * QR hashing   (src/models/embeddings/qr_embedding.py:95-109): T1=emb1 (divider rows),
 *              T2=emb2, mod1 = div2 = divider, xform 0.  For op=cat the output is
 *              [n/F, 2F, De] (torch.cat(dim=1) on a [B,F] input), F=1 for 1-D input.
 * CERP         (src/models/embeddings/cerp_embedding.py:142-175): T1=P, T2=Q,
 *              mod1 = bucket_size, div2 = q_entity_per_row, op add, xform 1.
 * CERP retrain (cerp_embedding.py:329-367): xform 2.
 * idx int64[n]; negative ids and rows beyond n1/n2 are flagged in *err and yield zeros.
 */
MI_API int mi_dual_gather_fwd(const int64_t *idx, const float *T1, const float *T2,
                              const float *S1, const float *S2, const uint8_t *M1,
                              const uint8_t *M2, float *out, int64_t n, int32_t F,
                              int32_t De, int64_t n1, int64_t n2, int64_t mod1,
                              int64_t div2, int32_t op, int32_t xform, int32_t *err,
                              void *stream);
/* ... with the model's `x + offsets` (src/models/dcn.py:204 / deepfm.py:86: per-field ids [B, F] + cumulative field
 * offsets [F]) done inside the lookup: idx holds the raw ids, id i gets offsets[i % F] added, and the sums — what the
 * reference hands to its embedding module, what the backward and the sparse optimizer's field sort read — are written to
 * rows_out int64[n].  offsets == NULL: mi_dual_gather_fwd (rows_out untouched).  De % 4 == 0 with 16-byte aligned
 * tables only (MI_ERR_UNSUPPORTED otherwise: add the offsets first). */
MI_API int mi_dual_gather_fwd_off(const int64_t *idx, const int64_t *offsets, int64_t *rows_out, const float *T1,
                                  const float *T2, const float *S1, const float *S2, const uint8_t *M1,
                                  const uint8_t *M2, float *out, int64_t n, int32_t F, int32_t De, int64_t n1,
                                  int64_t n2, int64_t mod1, int64_t div2, int32_t op, int32_t xform, int32_t *err,
                                  void *stream);

/* Dense gradients of the above into caller-zeroed gT1/gT2 (and gS1/gS2 for xform 1),
 * float atomics; tables of <= 4096 elements are pre-summed per workgroup in LDS. */
MI_API int mi_dual_gather_bwd(const int64_t *idx, const float *g_out, const float *T1,
                              const float *T2, const float *S1, const float *S2,
                              const uint8_t *M1, const uint8_t *M2, float *gT1, float *gT2,
                              float *gS1, float *gS2, int64_t n, int32_t F, int32_t De,
                              int64_t n1, int64_t n2, int64_t mod1, int64_t div2,
                              int32_t op, int32_t xform, void *stream);
/* Row form of table 2's gradient (round 4; plain tables only — QRHashingEmbedding, src/models/embeddings/qr_embedding.py:95-109
 * through autograd): g2vals fp32[n, De] receives lookup i's contribution to row rows2[i] = idx[i] / div2 of table 2 (rows2
 * int64[n]; an id out of range gets row 0 and a ZERO value row) — an uncoalesced COO gradient like
 * nn.Embedding(sparse=True)'s, written with plain coalesced stores: no float atomics into scattered rows, no [n2, De]
 * zero-fill.  gT1 (dense) is added to (caller-zeroed) as above, EXCEPT in the joined form below.
 * workspace (nullable): mi_dual_gather_bwd_rows_workspace_elems(De, n1) floats (0: this shape has no use for one) whose first
 * word is ZERO on entry (the kernel leaves it zero): table 1's per-workgroup sums are then joined by the last workgroup to
 * arrive, in a fixed order, instead of by same-address float atomics — and gT1 is WRITTEN, not added to (no zero fill
 * needed), whenever a workspace is passed and mi_dual_gather_bwd_rows_overwrites(De, n1) != 0 (16-byte aligned operands). */
MI_API int32_t mi_dual_gather_bwd_rows_overwrites(int32_t De, int64_t n1);
MI_API int64_t mi_dual_gather_bwd_rows_workspace_elems(int32_t De, int64_t n1);
MI_API int mi_dual_gather_bwd_rows(const int64_t *idx, const float *g_out, const float *T1, const float *T2, float *gT1,
                                   float *g2vals, int64_t *rows2, int64_t n, int32_t F, int32_t De, int64_t n1,
                                   int64_t n2, int64_t mod1, int64_t div2, int32_t op, float *workspace, void *stream);
/* The same, told which FIELDS of idx[B, F] have a handful of values (n = B * F; xform 0 only, else the hint is ignored):
 * small_fields int32[n_small] = their column indices, is_small uint8[F] the same as flags, field_row0 int64[F] = the first
 * row of table 2 a field's ids reach (all device arrays).  A small field's B lookups land on <= 16 rows of table 2 from
 * field_row0[f]; their gradient is summed per field in registers by extra workgroups and added once per (row, column) and
 * workgroup instead of B same-address atomics (an id outside that span is still added, atomically).  n_small = 0: plain. */
MI_API int mi_dual_gather_bwd_fields(const int64_t *idx, const float *g_out, const float *T1, const float *T2,
                                     const float *S1, const float *S2, const uint8_t *M1, const uint8_t *M2, float *gT1,
                                     float *gT2, float *gS1, float *gS2, int64_t n, int32_t F, int32_t De, int64_t n1,
                                     int64_t n2, int64_t mod1, int64_t div2, int32_t op, int32_t xform,
                                     const int32_t *small_fields, int32_t n_small, const int64_t *field_row0,
                                     const uint8_t *is_small, void *stream);


/* ---- §8f rank 4 (first flavour): single-table gather with a per-element transform -------------
 * PEP (src/models/embeddings/pep_embedding.py:82-92): out = sign(w) relu(|w| - sigmoid(s)), w = W[idx],
 *   s = S[row*srs + d*scs] — strides (0,0) global, (0,1) dimension, (1,0) feature, (D,1) feature_dim.
 * RetrainPep (:211-221): out = W[idx] * M (bool mask, 1 byte).  xform as in mi_dual_gather_fwd.
 * Backward: dense gW (and gS, s_numel elements) caller-zeroed, float atomics; small S pre-summed in LDS.
 */
MI_API int mi_xform_gather_fwd(const int64_t *idx, const float *W, const float *S,
                               const uint8_t *M, int64_t srs, int64_t scs, float *out,
                               int64_t n, int32_t D, int64_t N, int32_t xform, int32_t *err,
                               void *stream);
MI_API int mi_xform_gather_bwd(const int64_t *idx, const float *g_out, const float *W,
                               const float *S, const uint8_t *M, int64_t srs, int64_t scs,
                               float *gW, float *gS, int64_t s_numel, int64_t n, int32_t D,
                               int64_t N, int32_t xform, void *stream);

/* Quantised tables, inference only (src/models/embeddings/ptq_emb.py:24-25,85-91):
 * qtype 1: W is fp16[N,D], out = fp32(W[idx]); 2 / 3: W is int8 / int16 codes, out = (code - bias[0]) * scale[0]
 * with bias a device int8 / int16 word and scale a device fp32 word.
 */
MI_API int mi_gather_rows_quant(const int64_t *idx, const void *W, int32_t qtype,
                                const float *scale, const void *bias, float *out, int64_t n,
                                int32_t D, int64_t N, int32_t *err, void *stream);

/* ---- §8f rank 4 (third flavour): quantisation-aware-training lookup -----------------------------
 * QAT_EmbInt.forward = row gather + StotasticRounding (src/models/embeddings/qat_emb.py:16-45,117-119):
 *   q = clamp(W[idx]/scale, q_min, q_max); out = (floor(q) + [u > floor(q) + 1 - q]) * scale,
 *   q_min = -2^(n_bits-1), q_max = 2^(n_bits-1) - 1, scale a device scalar.
 * u = prob[element] when prob != NULL (the reference's torch.rand_like draw, for parity tests), else the
 * library's counter generator on (seed[0] + salt, element index) — the backward re-derives the same rounding.
 * idx == NULL: rows 0..n-1 of W (rounding an already reduced bag).  Backward (:49-84): dW[idx] += g
 * (float atomics into a caller-zeroed dense gradient; plain store when idx == NULL),
 * dscale[0] += sum g * m, m = q_max | q_min at the clamps, else (rounded - W/scale).  dW / dscale nullable. */
MI_API int mi_qat_gather_fwd(const int64_t *idx, const float *W, const float *scale, int32_t n_bits,
                             const float *prob, const int64_t *seed, int64_t salt, float *out,
                             int64_t n, int32_t D, int64_t N, int32_t *err, void *stream);
MI_API int mi_qat_gather_bwd(const int64_t *idx, const float *W, const float *scale, int32_t n_bits,
                             const float *prob, const int64_t *seed, int64_t salt, const float *g,
                             float *dW, float *dscale, int64_t n, int32_t D, int64_t N, void *stream);

/* ---- §8f rank 4 (fourth flavour): OptEmbed supernet lookup ---------------------------------------
 * OptEmbed.forward / get_weight (src/models/embeddings/deepfm_opt_embed.py:150-243) with _MaskEmbeddingModule
 * and BinaryStep (optembed_utils.py:10-112):
 *   y[i,:] = W[idx[i],:] * [ ||W[idx[i]]||_norm - t[tix] > 0 ] * [dim <= dmax[i]]
 * tix = tix[i] if given, else i % F (one threshold per field of a [B, F] lookup); t NULL = no row mask,
 * dmax NULL = all dimensions.  Backward: dW (dense, caller-zeroed, float atomics) gets the masked g plus the
 * straight-through term (sum_d g_d m_d w_d) * a(u) * d||w||/dw with BinaryStep's surrogate a(u); dt[tix] -= the
 * same scalar (caller-zeroed).  dW / dt nullable.                                                        */
MI_API int mi_optembed_fwd(const int64_t *idx, const float *W, const float *t, const int64_t *tix,
                           int32_t F, const int64_t *dmax, int32_t norm, float *out, int64_t n,
                           int32_t D, int64_t N, int32_t *err, void *stream);
MI_API int mi_optembed_bwd(const int64_t *idx, const float *W, const float *t, const int64_t *tix,
                           int32_t F, const int64_t *dmax, int32_t norm, const float *g, float *dW,
                           float *dt, int64_t n, int32_t D, int64_t N, void *stream);

/* ---- a11: CSR-pruned table rows (numba kernels K1/K2) ------------------------
 * src/models/embeddings/pruned_embedding.py:136-204: out[i,:] = dense row ids[i] of the
 * CSR matrix (values fp32, crow/col int64).  out fp32[n,D] need not be pre-zeroed.
 */
MI_API int mi_csr_rows_fwd(const float *values, const int64_t *crow, const int64_t *col,
                           const int64_t *ids, float *out, int64_t n, int32_t D,
                           int64_t N, int32_t *err, void *stream);

/* ---- a9: DHE universal hash features ------------------------------------------
 * src/models/embeddings/dh_embedding.py:213-236:
 *   out[i,k] = 2*(((slopes[k]*(ids[i]+prefix+1)+bias[k]) mod primes[k]) mod m)/(m-1) - 1
 * int64 floor-mod (bit-exact with torch %), then fp32 in the reference's op order.
 */
MI_API int mi_dhe_hash(const int64_t *ids, const int64_t *slopes, const int64_t *bias,
                       const int64_t *primes, float *out, int64_t n, int32_t K,
                       int64_t prefix, int64_t m, void *stream);

/* ---- a14: LightGCN propagation step, CSR SpMM with fused layer-sum ----------
 * src/models/lightgcn.py:82-88 (`step = matrix @ step; res = res + step; res / (L+1)`):
 *   y        = A[row,:] . X                      A in CSR: crow int32[n_rows+1], col int32, val fp32
 *   Y[row]   = y                                 (Y nullable: the last layer needs no `step`)
 *   acc_out[row] = (acc_in[row] + y) * scale     (acc_out nullable; acc_in nullable = 0; in place ok)
 * The reference hands over int64 crow/col (torch CSR); the host converts them to int32 once per
 * matrix.  X fp32[n_cols,D] may be split in two row segments (Xa rows [0,x_split), Xb the rest;
 * Xb null = one segment) so cat(user_table, item_table) (lightgcn.py:71-77) is never built;
 * likewise acc_in.  short_rows / long_rows (nullable pair) list the rows handled one-per-wave
 * and one-per-workgroup (hubs); null = every row one-per-wave.
 * The backward of `matrix @ step` is the same call on the transposed CSR.
 */
MI_API int mi_spmm_csr(const int32_t *crow, const int32_t *col, const float *val,
                       const float *Xa, const float *Xb, int32_t x_split, float *Y,
                       const float *acc_in_a, const float *acc_in_b, int32_t acc_split,
                       float *acc_out, float scale, int32_t n_rows, int32_t D,
                       const int32_t *short_rows, int32_t n_short,
                       const int32_t *long_rows, int32_t n_long, void *stream);
/* The same product with a row mask of X: bit (r & 31) of xmask[r >> 5] == 0 promises that row r of X = [Xa; Xb] is all
 * zeros, and the planned kernel does not fetch it (NULL: mi_spmm_csr).  mi_row_mask builds such a mask from X itself
 * ((n_rows + 31) / 32 words; D / 4 a power of two <= 64).  Serves the FIRST layer of a backward propagation
 * (src/models/lightgcn.py:82-88 through autograd): the incoming gradient is non-zero only on the batch's rows. */
MI_API int mi_row_mask(const float *Xa, const float *Xb, int32_t x_split, int32_t n_rows, int32_t D, uint32_t *mask,
                       void *stream);
MI_API int mi_spmm_csr_masked(const int32_t *crow, const int32_t *col, const float *val, const float *Xa,
                              const float *Xb, int32_t x_split, float *Y, const float *acc_in_a,
                              const float *acc_in_b, int32_t acc_split, float *acc_out, float scale, int32_t n_rows,
                              int32_t D, const int32_t *short_rows, int32_t n_short, const int32_t *long_rows,
                              int32_t n_long, const uint32_t *xmask, void *stream);
/* The same with a selection of OUTPUT rows for the LAST layer of a training step's propagation, which is read at the
 * batch's rows only: short_rows = the batch's non-hub rows (repeats allowed — Y / acc_out must then not alias the
 * inputs), long_rows = the graph's hub rows, of which only those whose byte in hub_need is set are computed (the byte is
 * cleared again by the kernel; NULL: all of them).  mi_batch_row_list builds both from a batch: out[3B] = users | U + pos |
 * U + neg with hubs (is_hub[row] != 0) and out-of-range ids replaced by `filler` (any non-hub row) and hub_need[row] = 1
 * for the hubs.  Rows that are not selected keep whatever Y / acc_out held. */
MI_API int mi_batch_row_list(const int64_t *users, const int64_t *pos, const int64_t *neg, int64_t B, int64_t U,
                             int64_t n_rows, const uint8_t *is_hub, int32_t filler, int32_t *out, uint8_t *hub_need,
                             void *stream);
MI_API int mi_spmm_csr_sel(const int32_t *crow, const int32_t *col, const float *val, const float *Xa,
                           const float *Xb, int32_t x_split, float *Y, const float *acc_in_a,
                           const float *acc_in_b, int32_t acc_split, float *acc_out, float scale, int32_t n_rows,
                           int32_t D, const int32_t *short_rows, int32_t n_short, const int32_t *long_rows,
                           int32_t n_long, const uint32_t *xmask, uint8_t *hub_need, void *stream);

/* The same product (and fused epilogue) in the TILED form (round 3; csrc/spmm.hip): a workgroup owns a tile of consecutive
 * output rows whose sums live in LDS for the whole launch, the tile's edges — one per 16-lane group, gathered row of X
 * times value added with LDS float atomics — are walked column block by column block so that the block of X being
 * gathered from stays in the XCD's L2.  Built once per sparsity pattern by the caller (CsrPlan.tiles on the Python side):
 *   tile_edge0, tile_row0  int32[ntiles + 1]  first edge / first output row of every tile; the tiles' row ranges cover
 *                          [0, n_rows) exactly once, at most max_tile_rows rows each (max_tile_rows * D * 4 <= 64 KiB);
 *   ecr   int32[nnz] = column | (row - tile_row0[tile]) << 23   (columns < 2^23), eval fp32[nnz] the values, both in
 *                          tile order with each tile's edges grouped by column block.
 * X / acc_in as two row segments, Y / acc_out / scale as in mi_spmm_csr.  The order of the LDS adds depends on timing:
 * results agree with mi_spmm_csr to float rounding, not bitwise. */
MI_API int mi_spmm_tiled(const int32_t *tile_edge0, const int32_t *tile_row0, int32_t ntiles, int32_t max_tile_rows,
                         const int32_t *ecr, const float *eval, const float *Xa, const float *Xb, int32_t x_split,
                         float *Y, const float *acc_in_a, const float *acc_in_b, int32_t acc_split, float *acc_out,
                         float scale, int32_t D, void *stream);

/* Round 4: task-balanced, slice-phased SpMM (src/models/lightgcn.py:79-87, the same product).  The columns of A (rows of
 * X) are cut into slices of <= ~2 MiB of X; the rows of A are packed once per sparsity pattern into tasks of about equal work
 * whose edges are stored slice by slice: every wave of the chip then gathers from the same slice of X at about the same time
 * and each XCD's 4 MiB L2 serves the re-uses (the row-per-wave kernel re-fetches X 8.4x).  NPW = 256 / D lane groups per wave.
 *   tptr  int32[n_tasks, NPW + 1]  narrow task: lane group k's edges = [tptr[k], tptr[k+1]); wide task: [tptr[0], tptr[1])
 *   trows int32[n_tasks, 4 NPW]    narrow: group k owns rows trows[j NPW + k], j = 0..3; wide: rows trows[j NPW]; -1 = none
 *   twide uint8[n_tasks]           1 = wide (all groups stride one edge range: rows of 48..256 nonzeros)
 *   ecol  int32[n_edges]           column | (which of its owner's rows) << 28, slice by slice inside a range; eval the values
 *   long_rows: hub rows, computed from the CSR (crow, col, val) by an 8-wave workgroup each; tasks + hubs cover every row once.
 * Other arguments as mi_spmm_csr_masked.  Fixed summation order (no atomics). */
MI_API int mi_spmm_sliced(const int32_t *crow, const int32_t *col, const float *val, const int32_t *tptr,
                          const int32_t *trows, const uint8_t *twide, const int32_t *ecol, const float *eval,
                          int32_t n_tasks, const float *Xa, const float *Xb, int32_t x_split, float *Y,
                          const float *acc_in_a, const float *acc_in_b, int32_t acc_split, float *acc_out,
                          float scale, int32_t D, const int32_t *long_rows, int32_t n_long, const uint32_t *xmask,
                          void *stream);

/* ---- a10: TT-Rec lookup (TTRecTorch semantics) --------------------------------
 * src/models/embeddings/tensortrain_embeddings.py:100-150: mixed-radix split of idx over
 * p_shapes, one slice per core (core c: fp32[1, p_c, r_c*q_c*r_{c+1}] viewed (p_c,r_c,q_c,r_{c+1})),
 * chain contraction -> out fp32[n, D], D = prod(q_shapes).  `cores`, `gcores`, `p_shapes`,
 * `q_shapes`, `ranks` (ncores+1 entries, first and last 1) are HOST arrays; cores[c]/gcores[c] are
 * device pointers.  2 <= ncores <= 4.  Ids outside [0, N) are flagged and yield zeros.
 * mi_tt_bwd adds the dense gradients of every core into caller-zeroed gcores[c] (float atomics).
 */
MI_API int mi_tt_fwd(const int64_t *idx, const float *const *cores, int32_t ncores,
                     const int32_t *p_shapes, const int32_t *q_shapes, const int32_t *ranks,
                     float *out, int64_t n, int32_t D, int64_t N, int32_t *err, void *stream);
MI_API int mi_tt_bwd(const int64_t *idx, const float *g_out, const float *const *cores,
                     float *const *gcores, int32_t ncores, const int32_t *p_shapes,
                     const int32_t *q_shapes, const int32_t *ranks, int64_t n, int32_t D,
                     int64_t N, void *stream);

/* ---- a10, grouped form: the TT-Rec lookup as one GEMM per core slice ---------------------------
 * Same arithmetic as mi_tt_fwd/bwd (tensortrain_embeddings.py:100-150).  The lookups of level c are
 * ordered by their digit i_c, so that all lookups sharing the slice core_c[i_c] are consecutive rows:
 *   res_c[(l,h), :] = res_{c-1}[(l,h), :] . core_c[i_c(l)]   viewed [r_c, q_c*r_{c+1}]
 * runs on the MFMA units through mi_gemm_f32_row_groups (forward and input gradient) and
 * mi_gemm_f32_k_groups (slice gradients); the host side (recsys-benchmark_amd/_kernels.py) builds the
 * orderings with device-side sorts, no host sync.
 *   mi_tt_digits: digits[c*n + i] = c-th mixed-radix digit of idx[i] over p_shapes (int32); ids outside
 *     [0, N) give digits 0, valid[i] = 0 and MI_IDX_OUT_OF_RANGE in *err.
 *   mi_move_chunks: dst[dst_row[i]*dst_stride + e] (= or +=, float atomics) src[src_row[i]*src_stride + e]
 *     for e < width (multiple of 4), zeroed / skipped where mask[i] == 0; NULL row arrays = identity.
 *   mi_gemm_f32_row_groups: C[M,N] = A[M,K] . opB(B + mtile_b[m/64]*sB); 64-row tiles with mtile_b < 0
 *     are skipped (the rows of a group are padded to whole tiles).
 *   mi_gemm_f32_k_groups: for each segment s: C[kseg[s][2]*sC ...][M,N] += A[k0:k0+K, :M]^T . B[k0:k0+K, :N]
 *     with (k0, K) = kseg[s][0:2], float atomics, C zeroed by the caller.
 *   mi_tt_plan_level: counting sort of the lookups by digit[n] in [0, p), p <= 4096: pos[i] = first row of lookup i in the
 *     level's layout (groups in digit order, each padded to whole 64-row tiles, H rows per lookup; order inside a
 *     group unspecified), mtile_b[ntiles] = group of every row tile (-1 unused), kseg[nseg][3] = the groups' rows
 *     cut into segments of <= seg rows, pbeg[p] = first row of every group.  ntiles >= ceil(n*H/64) + p,
 *     nseg >= ceil(n*H/seg) + p; workspace: 2*p int32.
 *   mi_segment_sum: out[kseg[s][2]*ldo + j] += sum_{r<K} X[(k0+r)*ldx + j], j < width (core 0's gradient:
 *     the lookups sharing a first digit are consecutive rows).                                          */
MI_API int mi_tt_digits(const int64_t *idx, int64_t n, int64_t N, const int32_t *p_shapes,
                        int32_t ncores, int32_t *digits, uint8_t *valid, int32_t *err, void *stream);
MI_API int mi_move_chunks(const float *src, const int64_t *src_row, int64_t src_stride, float *dst,
                          const int64_t *dst_row, int64_t dst_stride, int32_t width, int64_t n,
                          const uint8_t *mask, int32_t accumulate, void *stream);
MI_API int mi_gemm_f32_row_groups(const float *A, const float *B, float *C, int32_t M, int32_t N,
                                  int32_t K, int32_t lda, int32_t ldb, int32_t ldc, int32_t transB,
                                  int64_t sB, const int32_t *mtile_b, void *stream);
/*   mi_tt_last_fwd / mi_tt_last_bwd: the LAST level (rank 1 on the right, q = q_last output columns) without a GEMM:
 *     out[l, h*q + j] = sum_k chunk_l[h, k] * core[digit[l]][k, j], chunk_l = src + src_row[l]*src_stride (H x K floats);
 *     bwd: dst + dst_row[l]*dst_stride <- dchunk_l[h, k] = sum_j g[l, h*q+j] * core[digit[l]][k, j]   (dst nullable);
 *          gcore[grp][k, j] += sum over each kseg segment (first, count, grp) of lookups order[first + r]   (gcore nullable,
 *          caller-zeroed, float atomics).  H*q <= 64 and a divisor of 64, H*K <= 512, K*q <= 512.                                   */
MI_API int mi_tt_last_fwd(const float *src, const int64_t *src_row, int64_t src_stride, const float *core,
                          const int32_t *digit, const uint8_t *valid, int32_t H, int32_t K, int32_t q,
                          float *out, int64_t n, void *stream);
MI_API int mi_tt_last_bwd(const float *src, const int64_t *src_row, int64_t src_stride, const float *core,
                          const int32_t *digit, const uint8_t *valid, int32_t H, int32_t K, int32_t q,
                          const float *g, float *dst, const int64_t *dst_row, int64_t dst_stride,
                          const int64_t *order, const int64_t *kseg, int32_t nseg, float *gcore,
                          int64_t n, void *stream);
MI_API int mi_tt_plan_level(const int32_t *digit, int64_t n, int32_t p, int32_t H, int32_t seg,
                            int32_t *workspace, int64_t *pbeg, int64_t *pos, int32_t *mtile_b,
                            int32_t ntiles, int64_t *kseg, int32_t nseg, void *stream);
MI_API int mi_segment_sum(const float *X, int64_t ldx, int32_t width, const int64_t *kseg,
                          int32_t nseg, float *out, int64_t ldo, void *stream);
MI_API int mi_gemm_f32_k_groups(const float *A, const float *B, float *C, int32_t M, int32_t N,
                                int32_t lda, int32_t ldb, int32_t ldc, int64_t sC, const int64_t *kseg,
                                int32_t nseg, void *stream);

/* ---- a12/a13: fp32 MFMA GEMM with fused CrossNet epilogues ------------------
 * C[M,N] = epi( sum_{g<kgroups} opA(A + g*gA)[M,K] . opB(B + g*gB)[K,N] ), batched over `batch`
 * with element strides sA/sB/sC (and sR1/sR2/sC2).  transA=0: A[m*lda+k], 1: A[k*lda+m];
 * transB=0: B[k*ldb+n], 1: B[n*ldb+k] (nn.Linear weight layout).  v_mfma_f32_32x32x2_f32, exact fp32.
 * epi: 0 C=acc | 1 C=acc+bias[n] | 2 C=tanh(acc)
 *      3 lin=acc+bias[n]*rs(m); C=R1+R2*lin; C2(opt)=lin   rs(m)=sum_{e<nrs} rowscale[m*nrs+e] (1 if null)
 *        -> DCNHead `x_l + x_0*(W x_l + b)` (src/models/layer_dcn.py:137-139) and the DCN_MixHead
 *           output `x_l + x_0 * sum_e g_e (U_e h_e + b)` (:102-113) with K = E*rank
 *      4 C=R1+acc(+R2)(+sum_{e<nrs} rowscale[m*nrs+e]*bias[e*N+n] when rowscale and bias are given: the gate's share
 *        dgate.G of dx_l, layer_dcn.py:100-113 differentiated) | 5 C=acc*(1-R1^2) (tanh') | 6 h=tanh(acc); C=h; C2=h*rowscale[m*nrs+z] | 7 C+=acc
 * splitk > 1 (epi 0 or 7 only; for epi 0 the caller zeroes C): the K loop is cut into `splitk` slices
 * on separate workgroups that add into C with float atomics — for the weight-gradient products
 * whose M x N is a handful of tiles while K is the batch (dW = dlin^T x_l, K = 4096).
 */
MI_API int mi_gemm_f32(const float *A, const float *B, float *C, int32_t M, int32_t N,
                       int32_t K, int32_t lda, int32_t ldb, int32_t ldc, int32_t transA,
                       int32_t transB, int32_t batch, int64_t sA, int64_t sB, int64_t sC,
                       int32_t kgroups, int64_t gA, int64_t gB, int32_t epi,
                       const float *bias, const float *R1, int32_t ldr1, int64_t sR1,
                       const float *R2, int32_t ldr2, int64_t sR2, const float *rowscale,
                       int32_t nrs, float *C2, int32_t ldc2, int64_t sC2, int32_t splitk,
                       void *stream);

/* Several independent products of ONE operand layout in a single launch — the weight gradients of a CrossNet backward
 * (src/models/layer_dcn.py:27-115 through autograd: dU, dC, dV, dG of every layer are each a few 64x64 tiles with a
 * B-long reduction; one at a time none fills the chip).  Problem j: C_j[batch] (+)= opA(A_j) opB(B_j), batch strides
 * sA/sB/sC, `splitk` K-slices that meet in float atomics (then C_j must be zeroed by the caller, or accumulate != 0);
 * splitk = 0 leaves the cut to the library (C_j zeroed / accumulate as for any cut).
 * n <= 16; transA/transB as in mi_gemm_f32 and common to all problems.  `probs` is a HOST array read during the call.
 * The form transA = 1, transB = 0 (both operands reduction-major: dW = dz^T a, also the MLP tail's weight gradients)
 * runs — when every problem has K % 32 == 0, M % 4 == N % 4 == 0 and 16-byte aligned operands / strides — on a kernel
 * that moves both operands global -> LDS by DMA in their memory layout (MI_GEMM_TN_DMA=0 in the environment: the
 * general kernel).
 * mi_gemm_f32_multi_plan: the cut that call would make, by host arithmetic only (nothing launched, no GPU needed):
 * *kind = 1 (the LDS-DMA kernel) or 0 (the general one), splitk[j] = K-slices of problem j (n entries; 0 for an empty
 * problem), *workgroups = the launch's size. */
typedef struct mi_gemm_problem {
  const float *A, *B;
  float *C;
  int32_t M, N, K;
  int32_t lda, ldb, ldc;
  int32_t batch;
  int64_t sA, sB, sC;
  int32_t splitk;
  int32_t accumulate;
} mi_gemm_problem;
MI_API int mi_gemm_f32_multi(const mi_gemm_problem *probs, int32_t n, int32_t transA, int32_t transB, void *stream);
/* ... carrying a mi_prefetch_rows job in workgroups past the products' grid: the MLP tail's weight-gradient launch is
 * MFMA-bound (~45 us with the HBM idle), the lookup of the NEXT step is a chain of HBM round trips — touched here, its table
 * lines wait in the Infinity Cache (DeepFM.prefetch_next; src/trainer/deepfm.py:40-60 holds the next batch while the step
 * runs).  ride == NULL: mi_gemm_f32_multi.  A launch that cannot carry it (not the A^T B LDS-DMA form) runs the job as a
 * launch of its own first. */
MI_API int mi_gemm_f32_multi_ride(const mi_gemm_problem *probs, int32_t n, int32_t transA, int32_t transB,
                                  const mi_prefetch_rows_job *ride, void *stream);
MI_API int mi_gemm_f32_multi_plan(const mi_gemm_problem *probs, int32_t n, int32_t transA, int32_t transB, int32_t *kind,
                                  int64_t *workgroups, int32_t *splitk);

/* The same contraction on a second tiling, for the large products of a DCN_MixHead layer (layer_dcn.py:90-115 and their
 * gradients): C[M,N] = epi(A[M,K] . B), A row-major.  A workgroup owns a 64-row panel of A by one of nt equal column ranges
 * (<= 112 columns), nt chosen so that the grid is whole rounds of 256 workgroups (N = 352: 4 x 88) — where 64 x 64 tiles
 * leave half of the second round empty.  B: b_layout 0: B(k,n) = B[(k/gw)*gstride + n*ldb + k%gw] (nn.Linear layout;
 * grouped: the experts' V_e^T side by side along k), b_layout 1: B(k,n) = B[(n/gw)*gstride + k*ldb + n%gw]; gw = the
 * whole extent and gstride = 0 for an ungrouped B.  epi: 0, 2, 3, 4 of mi_gemm_f32 (R1, R2, C2 share C's row pitch ldc;
 * epi 3/4 take bias / rowscale / nrs as there).  Everything 16-B aligned and N, K, gw, the pitches multiples of 4, else
 * MI_ERR_UNSUPPORTED (the caller keeps mi_gemm_f32). */
MI_API int mi_gemm_f32_panel(const float *A, int32_t lda, const float *B, int32_t ldb, int32_t b_layout,
                             int32_t gw, int64_t gstride, float *C, int32_t ldc, int32_t M, int32_t N,
                             int32_t K, int32_t epi, const float *bias, const float *R1, const float *R2,
                             const float *rowscale, int32_t nrs, float *C2, void *stream);

/* The per-expert middle of a DCN_MixHead layer (layer_dcn.py:96-108) with the r x r product in the epilogue of the d-long
 * one; a workgroup owns 64 rows and one expert e.  V fp32[E,d,r], C fp32[E,r,r], U fp32[E,r,d], gate fp32[M,E], the H / dZ
 * matrices fp32[M,E*r].
 *   mi_mix_expert_fwd: H1_e = tanh(x V_e); H2_e = tanh(H1_e C_e); H2g_e = H2_e * gate[m,e]
 *   mi_mix_expert_bwd: dH = dT U_e^T (never stored); dgate[m,e] = sum_k dH*H2_e + dgs[m];
 *                      dZ2_e = dH * gate[m,e] * (1 - H2_e^2); dZ1_e = (dZ2_e C_e^T) * (1 - H1_e^2)
 * r in {16, 32, 64}, d % 4 == 0, 16-B aligned; else MI_ERR_UNSUPPORTED (the caller keeps the separate products). */
MI_API int mi_mix_expert_fwd(const float *x, const float *V, const float *C, const float *gate, float *H1,
                             float *H2, float *H2g, int32_t M, int32_t d, int32_t E, int32_t r, void *stream);
MI_API int mi_mix_expert_bwd(const float *dT, const float *U, const float *C, const float *gate,
                             const float *H1, const float *H2, const float *dgs, float *dgate, float *dZ2,
                             float *dZ1, int32_t M, int32_t d, int32_t E, int32_t r, void *stream);

/* Elementwise / reduction pieces of the CrossNet backward (layer_dcn.py:90-140 differentiated):
 *   mi_cross_bwd_pre: dlin = g*x0; dx0 (+)= g*lin          (n elements)
 *   mi_colsum:        out[n] += sum_m X[m,n]*rs(m)          (out caller-zeroed; bias gradients);
 *                     rs_sum[0] += sum_m rs(m) if given (a 1-output Linear's dW and db in one launch)
 *   mi_rowdot:        out[m]  = sum_n X[m,n]*v[n] (+ bias[0]) (+ addend[m])   (also the forward of a 1-output Linear)
 *   mi_outer:         out[m,n] = g[m]*w[n]                       (its backward w.r.t. the input)
 *   mi_mix_gate_bwd:  dgate[m,e] = sum_k dH2g*H2 + dgsum[m]; dZ2 = dH2g*gate[m,e]*(1-H2^2)   ([M,E*r] operands)
 */
MI_API int mi_cross_bwd_pre(const float *g, const float *x0, const float *lin, float *dlin,
                            float *dx0, int64_t n, int32_t accumulate, void *stream);
MI_API int mi_colsum(const float *X, int32_t ldx, const float *rowscale, int32_t nrs,
                     float *out, float *rs_sum, int32_t M, int32_t N, void *stream);
MI_API int mi_rowdot(const float *X, int32_t ldx, const float *v, const float *bias,
                     const float *addend, float *out, int32_t M, int32_t N, void *stream);
/* The trainer's loss on the logits (src/trainer/deepfm.py:32,51: BCEWithLogitsLoss, reduction "mean"):
 * loss[0] = mean(max(x,0) - x*y + log1p(exp(-|x|)));  dx = g[0]*(sigmoid(x) - y)/n.  One launch each way; dx_unit
 * (nullable, [n]): the forward also writes dx for g = 1, so a backward seeded with 1 has nothing to launch. */
MI_API int mi_bce_logits_fwd(const float *x, const float *y, float *loss, float *dx_unit, int64_t n,
                             void *stream);
MI_API int mi_bce_logits_bwd(const float *x, const float *y, const float *g, float *dx, int64_t n,
                             void *stream);
MI_API int mi_outer(const float *g, const float *w, float *out, int32_t M, int32_t N,
                    void *stream);
MI_API int mi_mix_gate_bwd(const float *dH2g, const float *H2, const float *gate,
                           const float *dgsum, float *dgate, float *dZ2, int32_t M, int32_t E,
                           int32_t r, void *stream);
/*   mi_rowdot_multi:  out[m,e] = sum_n X[m,n]*W[e,n], e < E <= 8 — the DCN_MixHead gate g_e = x_l . G_e
 *                     (layer_dcn.py:100-103) as one pass over the row (N % 4 == 0, 16-B aligned; else MI_ERR_UNSUPPORTED).
 *   mi_cross_bwd_head: mi_cross_bwd_pre + the bias gradient db[n] += sum_m dlin[m,n]*rs(m) (rs = sum_e gate[m,e], 1 when
 *                     gate is NULL; db caller-zeroed) + dgs[m] = sum_n dlin[m,n]*b[n] (nullable) in ONE pass over the rows
 *                     (N % 4 == 0, N <= 1024, 16-B aligned; else MI_ERR_UNSUPPORTED and the caller uses the three calls). */
MI_API int mi_rowdot_multi(const float *X, int32_t ldx, const float *W, float *out, int32_t M,
                           int32_t N, int32_t E, void *stream);
MI_API int mi_cross_bwd_head(const float *g, const float *x0, const float *lin, const float *gate,
                             int32_t E, const float *b, float *dlin, float *dx0, int32_t accumulate,
                             float *db, float *dgs, int32_t M, int32_t N, void *stream);

/* ---- a5 (memory-bound part): BatchNorm1d + ReLU + Dropout around the MLP's Linears ---------
 * src/models/deepfm.py:53-66 / src/models/dcn.py:56-66.  Z fp32[M,N] (row stride ldz) is a Linear's
 * output.  has_bn=0 skips the normalisation (use_batchnorm=False).  training: batch statistics
 * (biased variance for the normalisation, running stats updated with `momentum` and the unbiased
 * variance, like nn.BatchNorm1d) — `stats` fp32[2,N] must be zeroed by the caller; eval: running
 * stats.  Dropout keeps an element iff splitmix64(seed[0]+salt, index) >= p*2^32 and scales by
 * 1/(1-p); the keep mask (1 byte/element) and save_mean/save_rstd fp32[N] feed the backward.
 * bump_seed (training BN only): the statistics launch first does seed[0] += 1 — one new dropout
 * stream per pass without a separate launch; num_batches_tracked (nullable) gets += 1.
 * mean_offset (nullable, training BN): added to the batch mean in the running_mean update only.  A Linear's
 * bias cancels inside a training-mode BatchNorm (it shifts z and its batch mean alike), so the caller may run the
 * contraction WITHOUT the bias, pass the bias here, and get the same y, gradients and running statistics.
 * Backward: dY NULL means a rank-1 upstream gradient dY[m,n] = gvec[m]*wvec[n] — the backward of a following
 * 1-output Linear (the tail's last layer), computed on the fly instead of being written and read back twice.
 * dgamma_dbeta fp32[2,N] (zeroed by the caller) receives dgamma then dbeta;
 * dZ = gamma*rstd*(dyh - dbeta/M - zh*dgamma/M) in training.
 */
MI_API int mi_bn_relu_dropout_fwd(const float *Z, int32_t ldz, int32_t M, int32_t N,
                                  int32_t has_bn, int32_t training, const float *gamma,
                                  const float *beta, float *running_mean, float *running_var,
                                  float momentum, float eps, float p, int64_t *seed,
                                  int64_t salt, int32_t bump_seed, int64_t *num_batches_tracked,
                                  float *stats, const float *mean_offset, float *Y, uint8_t *keep,
                                  float *save_mean, float *save_rstd, void *stream);
MI_API int mi_bn_relu_dropout_bwd(const float *dY, const float *Z, int32_t ldz, int32_t M,
                                  int32_t N, int32_t has_bn, int32_t training,
                                  const uint8_t *keep, float p, const float *gamma,
                                  const float *beta, const float *save_mean,
                                  const float *save_rstd, float *dgamma_dbeta, float *dZ,
                                  const float *gvec, const float *wvec, void *stream);

/* ---- §8f rank 3: the contrastive loss of the LightGCN step ------------------------------------
 * Reference: info_nce (src/losses.py:25-47) as called by the trainer (src/trainer/lightgcn.py:215-229).
 * mi_rownorm_fwd: Y[r,:] = X[r,:] * inv[r], inv[r] = 1/max(|X[r,:]|_2, eps)  (F.normalize(dim=1), eps 1e-12).
 * mi_rownorm_bwd: dX = inv * (dY - Y <Y,dY>)  (dX = inv * dY for a row clamped at eps).
 * mi_lse_diag_fwd: S fp32 [n,n] (row stride ld) raw scores; lse[i] = logsumexp_j(S[i,j]*inv_t);
 *   loss[0] = mean_i (lse[i] - S[i,i]*inv_t).  workspace: mi_lse_diag_workspace_elems(n) floats.
 * mi_lse_diag_bwd: S[i,j] <- g[0]/n * inv_t * (exp(S[i,j]*inv_t - lse[i]) - [i==j])  (in place: the gradient
 *   with respect to the raw scores; the two GEMMs that follow are mi_gemm_f32 calls).
 *   symmetric != 0 (both views are one matrix): writes dS + dS^T instead, so that the gradient is one product (dS+dS^T) v.
 *   valid uint8[n] + count fp32[1] (both or neither): rows/columns with valid == 0 are treated as absent and the mean runs
 *   over count[0] rows — the fixed-shape form of the trainer's torch.unique (src/trainer/lightgcn.py:407-413): the batch's
 *   rows as they come with repeated ids masked out, so the step has no data-dependent shape and can be captured.
 */
MI_API int mi_rownorm_fwd(const float *X, int64_t n, int32_t D, float eps, float *Y, float *inv,
                          void *stream);
MI_API int mi_rownorm_bwd(const float *Y, const float *inv, const float *dY, int64_t n, int32_t D,
                          float eps, float *dX, void *stream);
MI_API int64_t mi_lse_diag_workspace_elems(int32_t n);
MI_API int mi_lse_diag_fwd(const float *S, int64_t ld, int32_t n, float inv_t, const uint8_t *valid,
                           const float *count, float *lse, float *workspace, float *loss,
                           void *stream);
MI_API int mi_lse_diag_bwd(float *S, int64_t ld, int32_t n, float inv_t, const uint8_t *valid,
                           const float *count, const float *lse, const float *g, int32_t symmetric,
                           void *stream);

/* ---- §8f rank 1: fused row-sparse optimizer steps on row-form gradients -------------------
 * Reference: get_optimizers' sparse branch (src/models/deepfm.py:163-184): torch.optim.SparseAdam
 * on the embedding (no weight decay) / SGD with weight_decay=0 on it.
 * mi_sparse_adam_sorted: rows_sorted int64[n] ascending, perm int64[n] (sorted position -> entry of
 *   vals fp32[n,D]); duplicates are summed (== grad.coalesce()), then for each touched row
 *   m += (1-b1)(g-m); v += (1-b2)(g*g-v); W -= step_size * m/(sqrt(v)+eps),
 *   step_size = lr*sqrt(1-b2^t)/(1-b1^t) computed by the host (torch/optim/sparse_adam.py).
 *   step_size_dev (nullable): read the step size from device memory instead — mi_adam_tick(step, step_size, lr, b1, b2)
 *   does step[0] += 1 and writes it there, so a captured training step needs no host-side counter.
 *   acc fp32[n,D] is scratch (contents irrelevant before and after): a row repeated more often than one wave pass
 *   covers leaves a partial sum per pass there, which a second kernel adds in a fixed order (no atomics: the step is
 *   deterministic).
 */
MI_API int mi_sparse_adam_sorted(const int64_t *rows_sorted, const int64_t *perm,
                                 const float *vals, float *W, float *exp_avg,
                                 float *exp_avg_sq, float *acc, int64_t n, int32_t D,
                                 int64_t N, float step_size, const float *step_size_dev,
                                 double beta1, double beta2, float eps, void *stream);
/* the same step on a table with a row stride (row r of W starts at W + r*ldw floats, ldw >= D; exp_avg / exp_avg_sq stay
 * contiguous [N,D]): the packed DeepFM tables of mi_gather_fm_fwd_ld */
MI_API int mi_sparse_adam_sorted_ld(const int64_t *rows_sorted, const int64_t *perm,
                                    const float *vals, float *W, int64_t ldw, float *exp_avg,
                                    float *exp_avg_sq, float *acc, int64_t n, int32_t D,
                                    int64_t N, float step_size, const float *step_size_dev,
                                    double beta1, double beta2, float eps, void *stream);
/* mi_coalesce_rows_sorted: the deterministic dense gradient of a table out of its row-form gradient (what the reference's
 *   CPU index_add accumulates in a fixed order, src/models/embeddings/base.py:74-75 through autograd): G[row,:] = sum of
 *   vals[perm[i],:] over the sorted positions with rows_sorted[i] == row, added in a fixed order (the segmented sums of
 *   mi_sparse_adam_sorted, stored instead of applied; no atomics).  G fp32[N,D] zero-filled by the caller; acc fp32[n,D]
 *   scratch. */
MI_API int mi_coalesce_rows_sorted(const int64_t *rows_sorted, const int64_t *perm, const float *vals, float *G,
                                   float *acc, int64_t n, int32_t D, int64_t N, void *stream);
/* mi_sort_field_rows: rows int64[B,F] with rows[b,f] in [offsets[f], offsets[f+1]) (offsets ascending, offsets[F] := N —
 *   the ids DeepFM forms at src/models/deepfm.py:88).  rows_sorted[B*F] ascending and perm[i] = flat position b*F+f of
 *   the i-th smallest, equal ids in ascending b (a stable sort); an id outside its field's range comes back as N behind
 *   its field's valid ids (and, when it lies inside [0, N) — a row the lookup accepted — MI_IDX_OUT_OF_FIELD is OR-ed
 *   into *err: the row-wise optimizer skips id N, so that gradient is dropped and the caller must be told).  Every column is cut into runs of 1024 ids sorted by one workgroup each, then merged by
 *   rank; workspace: mi_sort_field_rows_workspace_bytes(B, F) bytes (0 when B <= 1024).  B <= 65536, N < 2^32-2, else
 *   MI_ERR_UNSUPPORTED (the caller sorts generically).
 */
MI_API int64_t mi_sort_field_rows_workspace_bytes(int64_t B, int32_t F);
MI_API int mi_sort_field_rows(const int64_t *rows, const int64_t *offsets, int64_t N, int64_t B,
                              int32_t F, int64_t *rows_sorted, int64_t *perm, void *workspace,
                              int32_t *err, void *stream);
MI_API int mi_adam_tick(float *step, float *step_size, double lr, double beta1, double beta2,
                        void *stream);
/* the same for `count` tables that share lr and betas, in one launch per 8 (host arrays of device pointers) */
MI_API int mi_adam_tick_multi(float *const *steps, float *const *step_sizes, int32_t count, double lr,
                              double beta1, double beta2, void *stream);
/* mi_adam_dense_multi: torch.optim.Adam (L2 weight decay, no amsgrad) over `count` dense fp32 tensors in as few launches as
 *   possible — get_optimizers' dense groups (src/models/deepfm.py:178-193).  The five pointer arrays and numels live in
 *   HOST memory; steps[i] points at tensor i's device-side step count (a float, torch's capturable layout), read as
 *   t = step + 1 and advanced by one after the update:
 *   g += wd*p; m += (1-b1)(g-m); v = b2 v + (1-b2) g^2; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps).
 *   The betas are doubles: 1 - beta is formed in double and then rounded (1.f - 0.999f is off by 1.3e-5).
 *   tickets (nullable): ceil(count/24) device words, zero between calls — the last workgroup of each launch advances the
 *   step counts itself; without them a second one-wave launch does.
 */
MI_API int mi_adam_dense_multi(float *const *params, const float *const *grads, float *const *exp_avgs,
                               float *const *exp_avg_sqs, float *const *steps, const int64_t *numels,
                               int32_t count, float lr, double beta1, double beta2, float eps,
                               float weight_decay, uint32_t *tickets, void *stream);
MI_API int mi_scatter_axpy_rows(const int64_t *idx, const float *g, float alpha, float *W,
                                int64_t n, int32_t D, int64_t N, void *stream);

/* ---- §8f rank 3: the LightGCN step either side of the propagation -------------------------------
 * mi_bpr_fwd/bwd: src/losses.py:6-22 on rows picked from the propagated tables
 *   (torch.index_select x3, src/trainer/lightgcn.py:395-399): x_b = u_b.(p_b - n_b) with
 *   u_b = U[ui[b]], p_b = P[pi[b]], n_b = Nn[ni[b]] (NULL index arrays = row b);
 *   loss = mean_b -logsigmoid(x_b) (partials joined in index order: deterministic), sig[b] = sigmoid(-x_b).
 *   bwd: dU[ui[b]] += c_b (p_b - n_b), dP[pi[b]] += c_b u_b, dN[ni[b]] -= c_b u_b, c_b = -g sig[b] / B
 *   (float atomics into caller-zeroed dense gradients when the index array is given, stores otherwise;
 *   any of dU/dP/dN may be NULL).  workspace: mi_bpr_workspace_elems(B) floats.
 *   nU/nP/nN = rows of U/P/Nn: a triple with an index outside its table touches no memory, counts as
 *   u = p = n = 0 forward, adds nothing backward, and ORs MI_IDX_OUT_OF_RANGE into *err (the
 *   reference's index_select raises: src/trainer/lightgcn.py:395-397).  Same for mi_rowsq_*.
 * mi_mask_topk_rows: src/trainer/lightgcn.py:122-138: for row b (user users[b], or b if NULL)
 *   scores[b, col[crow[u] .. crow[u+1])] = -inf (in place; crow may be NULL = no mask), then the
 *   indices (and optionally values) of the k largest entries, score descending, ties by ascending
 *   index.  k <= 256.                                                                               */
/* mi_rowsq_fwd/bwd: LightGCN.get_reg_loss (src/models/lightgcn.py:90-100) on plain tables:
 *   out = (sum_b |U[ui[b]]|^2 + |P[pi[b]]|^2 + |Nn[ni[b]]|^2) / (2B), joined in index order;
 *   bwd: dU[ui[b]] += g U[ui[b]] / B (and likewise dP, dN; float atomics, caller-zeroed, nullable).
 *   workspace: mi_bpr_workspace_elems(B) floats.                                                      */
MI_API int mi_rowsq_fwd(const float *U, const int64_t *ui, const float *P, const int64_t *pi,
                        const float *Nn, const int64_t *ni, int64_t B, int32_t D, int64_t nU, int64_t nP,
                        int64_t nN, int32_t *err, float *workspace, float *out, void *stream);
MI_API int mi_rowsq_fwd_armed(const float *U, const int64_t *ui, const float *P, const int64_t *pi,
                              const float *Nn, const int64_t *ni, int64_t B, int32_t D, int64_t nU, int64_t nP,
                              int64_t nN, int32_t *err, float *workspace, float *out, void *stream);
MI_API int mi_rowsq_bwd(const float *U, const int64_t *ui, const float *P, const int64_t *pi,
                        const float *Nn, const int64_t *ni, int64_t B, int32_t D, int64_t nU, int64_t nP,
                        int64_t nN, const float *g, float *dU, float *dP, float *dN, void *stream);
MI_API int64_t mi_bpr_workspace_elems(int64_t B);
MI_API int mi_bpr_fwd(const float *U, const int64_t *ui, const float *P, const int64_t *pi,
                      const float *Nn, const int64_t *ni, int64_t B, int32_t D, int64_t nU, int64_t nP,
                      int64_t nN, int32_t *err, float *sig, float *workspace, float *loss, void *stream);
/* _armed: the same launch without the memset node that zeroes the arrival ticket — the caller promises the last word
 * of `workspace` is zero on entry; the kernel leaves it zero, so a workspace zeroed once serves every later call. */
MI_API int mi_bpr_fwd_armed(const float *U, const int64_t *ui, const float *P, const int64_t *pi,
                            const float *Nn, const int64_t *ni, int64_t B, int32_t D, int64_t nU, int64_t nP,
                            int64_t nN, int32_t *err, float *sig, float *workspace, float *loss, void *stream);
/* ... with another scalar term of the step's objective joined in: loss = BPR + plus_weight * plus[0] (plus nullable; the
 * reference trainer's `loss = bpr + reg_weight * reg`, src/trainer/lightgcn.py:401-404, without a scale and an add launch).
 * With plus, `loss` has TWO words: loss[0] the sum, loss[1] the bare BPR term (what a trainer logs); without, one.
 * armed != 0: mi_bpr_fwd_armed's promise about the ticket word. */
MI_API int mi_bpr_fwd_plus(const float *U, const int64_t *ui, const float *P, const int64_t *pi, const float *Nn,
                           const int64_t *ni, int64_t B, int32_t D, int64_t nU, int64_t nP, int64_t nN, int32_t *err,
                           float *sig, float *workspace, int32_t armed, const float *plus, float plus_weight,
                           float *loss, void *stream);
MI_API int mi_bpr_bwd(const float *U, const int64_t *ui, const float *P, const int64_t *pi,
                      const float *Nn, const int64_t *ni, int64_t B, int32_t D, int64_t nU, int64_t nP,
                      int64_t nN, const float *sig, const float *g, float *dU, float *dP, float *dN,
                      void *stream);
MI_API int mi_mask_topk_rows(float *scores, int64_t ld, int64_t nrows, int64_t ncol,
                             const int64_t *users, const int64_t *crow, const int64_t *col, int32_t k,
                             int64_t *out_idx, float *out_val, void *stream);

/* ---- §8e: device-side routing of the row-sharded DeepFM lookup -------------------------------
 * No reference counterpart (the reference keeps one table on one device,
 * src/models/embeddings/base.py:52-57); the arithmetic either side of the exchange is the
 * reference's (src/models/deepfm.py:88-98).  Tables are sharded owner = row % world,
 * local = row / world; every shard ends in one extra SINK row that padding slots point at.
 *
 * mi_route_buckets: rows = idx + offsets[i % F] (offsets may be NULL) for n = B*F lookups; STABLE
 *   bucketing by owner into send_rows[world, cap] (owner-local row ids; unused slots = the owner's
 *   sink row ceil((N - owner) / world)), slot[n] = owner*cap + position of every lookup.  Lookups
 *   that overflow a bucket or are outside [0, N) get slot = world*cap (the dump slot) and OR 1 into
 *   *overflow / MI_IDX_OUT_OF_RANGE into *err.  workspace: mi_route_workspace_elems(n, world) int32.
 * mi_gather_pack_rows: out[i, 0:D] = W[rows[i], :], out[i, D] = w1[rows[i]], out[i, D+1:D+4] = 0;
 *   out rows are D + 4 floats (one all-to-all carries both tables).  D = 4 * 2^k only.
 * mi_slot_fm_fwd: mi_gather_fm_fwd over the received packed rows buf[nrows, D+4] addressed by slot
 *   (row nrows-1 = the zeroed dump row): emb[B,F,D], yfm[B].
 * mi_slot_fm_bwd: zero gbuf[nslot, D+4], then the row gradients of mi_gather_fm_bwd_rows written to
 *   gbuf[slot[b,f]] (first-order gradient in column D); slots >= nslot are skipped.
 * mi_unpack_rows: vals[i, 0:D] = packed[i, 0:D], lin[i] = packed[i, D] for m rows of D + 4 floats: the gradient rows
 *   received by the owner as the two contiguous value arrays of its shards' row-form gradients.  */
MI_API int64_t mi_route_workspace_elems(int64_t n, int32_t world);
MI_API int mi_route_buckets(const int64_t *idx, const int64_t *offsets, int64_t n, int32_t F,
                            int32_t world, int64_t N, int64_t cap, int32_t *workspace,
                            int64_t *send_rows, int64_t *slot, int32_t *overflow, int32_t *err,
                            void *stream);
MI_API int mi_gather_pack_rows(const int64_t *rows, const float *W, const float *w1, float *out,
                               int64_t m, int32_t D, int64_t Nl, int32_t *err, void *stream);
MI_API int mi_unpack_rows(const float *packed, float *vals, float *lin, int64_t m, int32_t D, void *stream);
MI_API int mi_slot_fm_fwd(const int64_t *slot, const float *buf, int64_t nrows, const float *bias,
                          float *emb_out, float *yfm_out, int64_t B, int32_t F, int32_t D,
                          int32_t *err, void *stream);
MI_API int mi_slot_fm_bwd(const int64_t *slot, const float *emb, const float *g_y,
                          const float *g_emb, float *gbuf, float *gbias, int64_t nslot, int64_t B,
                          int32_t F, int32_t D, void *stream);

/* ---- §8 a5 / f.2: the MLP tail as fused MFMA kernels (recsys-benchmark_amd/csrc/tail.hip) ---------------------------
 * Replaces, for (Linear, BatchNorm1d(training), ReLU, Dropout) x k + Linear(., 1) — src/models/deepfm.py:53-66,100-102
 * and src/models/dcn.py:56-66 — the library GEMMs and the separate BatchNorm / ReLU / Dropout passes.
 * An ACTIVATION is described by its saved pre-activation Z[M, ld] and per-feature constants:
 *     a(m, c) = max((Z[m,c] - mu[c]) * sc[c] + be[c], 0) * keep(m, c) / (1 - p)
 *   keep(m, c) = bit (c & 7) of keep_bits[(m*ld + c) >> 3] (ld % 8 == 0; used when p > 0), one bit per element written
 *   once per step by mi_tail_dropout_masks: 16 bits of splitmix64(seed[0] + salt', (m*ld + c)/4) >= p * 65536
 *   (nlayers <= 8 layers in one launch; salts / ps / lds / bits are HOST arrays, layers with p <= 0 are skipped).
 *   mu == NULL: the matrix is used as it is (the tail's input).
 * A PRE-ACTIVATION GRADIENT is described by DY (gradient w.r.t. the BatchNorm output, ReLU / dropout already applied),
 *   Z and constants:  dz(m, c) = al[c] * DY[m,c] + bz[c] * (Z[m,c] - mu[c]) + de[c];  al == NULL: DY is used as it is.
 * All matrices fp32 row-major, 16-byte aligned, leading dimensions and N, K multiples of 4 (else MI_ERR_UNSUPPORTED).
 * No atomics anywhere: every reduction is joined in a fixed order (bit-reproducible steps).
 *
 * mi_tail_fwd_gemm:  Z[M,N] = a(X)[M,K] . W[N,K]^T (no bias: it cancels in a training-mode BatchNorm; see mean_offset);
 *   part (nullable, mi_tail_part_elems(M, N) floats): (mean, M2) of every 64-row tile of Z, per column.
 *   a_out (nullable, [M,K] with X's pitch; needs x_mu): the activation a(X) as the operand load computed it — kept for the
 *   weight-gradient product when that runs on mi_gemm_f32_multi.
 * mi_tail_bn_finalize_fwd: joins `part` (Chan's formula, tile order) into mu = batch mean, sc = gamma * rstd, be = beta,
 *   rstd; running_mean / running_var updated as F.batch_norm(training=True) does (unbiased variance; mean_offset =
 *   the Linear's bias, added to the mean only there); *num_batches_tracked += 1; *seed_bump += 1 (nullable each).
 * mi_tail_head_fwd:  out[m] = sum_n a(m, n) w[n] + b[0] + add[m]  (b, add nullable).
 * mi_tail_head_bwd:  g[M] = dL/dout -> DY[M,N] of the last hidden layer, part[nblk, N, 2] = (sum dy, sum dy (z - mu)),
 *   wpart[nblk, N + 4] = (sum_m g a(m, n) ..., sum_m g); nblk = mi_tail_head_blocks(M).
 * mi_tail_bn_finalize_bwd: joins part[nblk, N, 2] -> dgamma, dbeta (nullable) and al, bz, de; wpart (nullable) -> dw[N], db[1].
 * mi_tail_dgrad_gemm: da[M,K] = dz[M,N] . W[N,K]; with the layer below given (p_mu != NULL):
 *   OUT = da * keep_prev/(1-p) * [pre_prev > 0] (= DY of that layer), part[MT, K, 2] (nullable) its column sums;
 *   otherwise OUT = da (the gradient of the tail's input).  dz_out (nullable, [M,N] with DY's pitch; needs al): dz as the
 *   operand load computed it, for the same purpose.
 * mi_tail_wgrad_gemm: dW[N,K] = dz[M,N]^T . a_prev[M,K]; the batch is cut into mi_tail_wgrad_splits(M, N, K) slices,
 *   slab (splits * N * K floats) holds their partial products, added in slice order into dW.                      */
/* The joins WITHOUT launches of their own (round 3): the kernel that CONSUMES an activation (or a pre-activation gradient)
 * joins the statistics behind its constants in its prologue — every workgroup for itself, in a fixed order, into LDS;
 * workgroup 0 writes the results where mi_tail_bn_finalize_fwd / _bwd would have (the backward pass and the optimizer
 * read them there).  Same arithmetic for every workgroup, no atomics: deterministic.  Host structs, read at call time.
 * mi_tail_bn_fwd: what mi_tail_bn_finalize_fwd takes — part[MT, N, 2] of the producing product, gamma / beta /
 *   mean_offset (nullable), running statistics + counters (nullable), momentum, eps — and its outputs mu, sc, be, rstd [N].
 * mi_tail_bn_bwd: what mi_tail_bn_finalize_bwd takes — part[nblk, N, 2], gamma (nullable), rstd, outputs dgamma / dbeta
 *   (nullable), al, bz, de [N]; wpart[nwblk, N + 4] (nullable) -> dw[N], db[1] (joined by a second workgroup).
 * mi_tail_fwd_gemm_m / mi_tail_head_fwd_m (x_stats / stats != NULL: the mu / sc / be arguments are ignored) and
 * mi_tail_dgrad_gemm_m (sums != NULL: al / bz / de are ignored; mu and Zl are still read); NULL = the plain entry point.
 * Features behind the joined constants <= 1024. */
typedef struct mi_tail_bn_fwd {
  const float *part;
  const float *gamma, *beta, *mean_offset;
  float *running_mean, *running_var;
  int64_t *num_batches_tracked, *seed_bump;
  float *mu, *sc, *be, *rstd;
  float momentum, eps;
  const float *shift;     /* nrep > 0: the shift the sums were taken around, [N] (mi_tail_fwd_gemm_s's shift_out) */
  int32_t nrep;           /* 0: `part` = tile statistics [MT, N, 2]; > 0: `part` = shifted sums fp64[nrep, 2, N] (mi_tail_fwd_gemm_s) */
} mi_tail_bn_fwd;
typedef struct mi_tail_bn_bwd {
  const float *part;
  const float *gamma, *rstd;
  float *dgamma, *dbeta, *al, *bz, *de;
  const float *wpart;
  float *dw, *db;
  int32_t nblk, nwblk;
  float *dbias;           /* affine != 0, nullable: the Linear bias's gradient al * sum dy, [N] */
  float *db2;             /* nullable: a second destination of db[0] (DeepFM: the scalar `_bias` is added to every logit like
                             the head's bias, so it has the same gradient — written here instead of by a copy launch) */
  int32_t affine;         /* != 0: fixed statistics (eval-mode BatchNorm) or none: dz = al dy, bz = de = 0 (mi_tail_bn_finalize_bwd_a) */
} mi_tail_bn_bwd;
MI_API int mi_tail_fwd_gemm_m(const float *X, int32_t ldx, const float *x_mu, const float *x_sc, const float *x_be,
                              float x_p, const uint8_t *x_keep, const float *W, int32_t ldw, float *Z, int32_t ldz,
                              float *part, float *a_out, int32_t M, int32_t N, int32_t K,
                              const mi_tail_bn_fwd *x_stats, void *stream);
/* BatchNorm statistics WITHOUT finalize launches (round 4).  The producing product adds, per column c and 64-row tile,
 *   t1 = sum (z - s_c),  t2 = sum (z - s_c)^2,   s_c = shift_running_mean[c] - shift_mean_offset[c]  (each nullable: 0)
 * (formed from the tile's exact count / mean / M2) into part[r, 0, c] / part[r, 1, c] — `part` holds DOUBLES here:
 * fp64[sum_reps, 2, N] = 4 sum_reps N floats, 8-byte aligned, ZEROED by the caller, r = row tile % sum_reps — with f64
 * atomics (the two sums cancel by (mean - s)^2 / var when the variance is derived: fp32 sums lose that many digits)
 * and stores s into shift_out[N]; the consuming kernel (mi_tail_fwd_gemm_* / mi_tail_head_fwd_m with a mi_tail_bn_fwd whose
 * nrep = sum_reps, part and shift point at these buffers) derives mean = s + t1/M, M2 = t2 - t1^2/M and the constants in
 * its prologue from (4 nrep + 3) N floats and its first workgroup writes mu / sc / be / rstd and the running statistics
 * (F.batch_norm(training=True) semantics, src/models/deepfm.py:57-58).  sum_reps = 0 is mi_tail_fwd_gemm_m.
 * mi_tail_dgrad_gemm_s: the backward mirror — part_reps > 0: part is fp32[part_reps, K, 2] (zeroed by the caller), every row
 * tile ADDS its column sums (sum dy, sum dy (z - mu)) into row (tile % part_reps); a mi_tail_bn_bwd with nblk = part_reps
 * over the same buffer then joins them in the next product's prologue (or mi_tail_bn_finalize_bwd does).
 * Float-atomic order: results vary in the last bits between runs; deterministic callers use the tile-statistics forms. */
MI_API int mi_tail_fwd_gemm_s(const float *X, int32_t ldx, const float *x_mu, const float *x_sc, const float *x_be,
                              float x_p, const uint8_t *x_keep, const float *W, int32_t ldw, float *Z, int32_t ldz,
                              float *part, float *a_out, int32_t M, int32_t N, int32_t K,
                              const mi_tail_bn_fwd *x_stats, int32_t sum_reps, float *shift_out,
                              const float *shift_running_mean, const float *shift_mean_offset, void *stream);
/* An INFERENCE forward's last hidden layer with the head in its epilogue (eval-mode BatchNorm or none on that layer, no
 * dropout, nothing kept): z = a(X) W^T is not stored; every 64-row x <= 112-column tile adds
 * sum_n relu((z - mu[n]) sc[n] + be[n]) w[n] into out[m] with a float atomic (out ZEROED by an earlier launch — the lookup's
 * riders do it in DeepFM's forward; column tile 0 also adds b[0] and add[m]): src/models/deepfm.py:100-105 under
 * model.eval() as one launch less. */
typedef struct mi_tail_head_in_epilogue {
  const float *w;            /* [N] the head Linear(N, 1)'s weight */
  const float *b;            /* [1], nullable */
  const float *add;          /* [M], nullable (DeepFM: y_fm) */
  const float *mu, *sc, *be; /* [N] constants of THIS layer's activation (mi_tail_affine_consts) */
  float *out;                /* [M] logits, zero on entry */
} mi_tail_head_in_epilogue;
MI_API int mi_tail_fwd_gemm_head(const float *X, int32_t ldx, const float *x_mu, const float *x_sc, const float *x_be,
                                 float x_p, const uint8_t *x_keep, const float *W, int32_t ldw, int32_t M, int32_t N,
                                 int32_t K, const mi_tail_bn_fwd *x_stats, const mi_tail_head_in_epilogue *head,
                                 void *stream);
MI_API int mi_tail_dgrad_gemm_s(const float *DY, const float *Zl, int32_t ld, const float *mu, const float *al,
                                const float *bz, const float *de, const float *W, int32_t ldw, const float *pZ,
                                int32_t pld, const float *p_mu, const float *p_sc, const float *p_be, float p_p,
                                const uint8_t *p_keep, float *OUT, int32_t ldo, float *part, int32_t part_reps,
                                float *dz_out, int32_t M, int32_t N, int32_t K, const mi_tail_bn_bwd *sums,
                                void *stream);
MI_API int mi_tail_head_fwd_m(const float *Z, int32_t ldz, const float *mu, const float *sc, const float *be, float p,
                              const uint8_t *keep, const float *w, const float *b, const float *add, float *out,
                              int32_t M, int32_t N, const mi_tail_bn_fwd *stats, void *stream);
MI_API int mi_tail_dgrad_gemm_m(const float *DY, const float *Zl, int32_t ld, const float *mu, const float *al,
                                const float *bz, const float *de, const float *W, int32_t ldw, const float *pZ,
                                int32_t pld, const float *p_mu, const float *p_sc, const float *p_be, float p_p,
                                const uint8_t *p_keep, float *OUT, int32_t ldo, float *part, float *dz_out, int32_t M,
                                int32_t N, int32_t K, const mi_tail_bn_bwd *sums, void *stream);
/* DeepFM: the FIRST layer's input-gradient product with the whole of mi_gather_fm_bwd_rows in its epilogue
 * (src/models/deepfm.py:88-102 through autograd).  The tail's input is the embedding block emb[M, K = F*D]; instead of
 * storing da = dz . W and launching the gather backward on it, the epilogue writes the lookup table's row-form gradient
 *   gvals[m, f, :] = da[m, f*D : (f+1)*D] + g_y[m] * (emb_sum[m, :] - emb[m, f, :])       (fp32[M*F, D], == [M, K])
 *   g1vals[m, f]   = g_y[m]                                                               (fp32[M, F], nullable)
 * emb_sum = mi_gather_fm_fwd_sum's sum_out.  The scalar bias's gradient sum_m g_y[m] equals the head bias's (both are
 * added to every logit): callers take it from mi_tail_bn_finalize_bwd's db.  One launch and 2 * 4 * M * K bytes (da written,
 * da read back) less than the two-kernel form.  Other arguments as mi_tail_dgrad_gemm_m with no layer below. */
MI_API int mi_tail_dgrad_gemm_fm(const float *DY, const float *Zl, int32_t ld, const float *mu, const float *al,
                                 const float *bz, const float *de, const float *W, int32_t ldw, float *gvals,
                                 float *dz_out, int32_t M, int32_t N, int32_t K, const mi_tail_bn_bwd *sums,
                                 const float *emb, const float *emb_sum, const float *g_y, float *g1vals, int32_t D,
                                 void *stream);
/* ... for the table-sharded lookup (mi_slot_fm_fwd's operands: rows gathered out of a receive buffer of packed
 * {D embedding floats, first-order weight, 3 pad} rows at row slot[m, f]): gbuf fp32[nrows, D + 4] is that buffer's gradient,
 * row slot[m, f] receives the embedding part and g_y[m] at column D — mi_slot_fm_bwd in the epilogue.  Rows no sample
 * points at keep what the caller put there (zeros: they travel back to owners as padding). */
MI_API int mi_tail_dgrad_gemm_fm_slot(const float *DY, const float *Zl, int32_t ld, const float *mu, const float *al,
                                      const float *bz, const float *de, const float *W, int32_t ldw, float *gbuf,
                                      float *dz_out, int32_t M, int32_t N, int32_t K, const mi_tail_bn_bwd *sums,
                                      const float *emb, const float *emb_sum, const float *g_y, const int64_t *slot,
                                      int32_t D, void *stream);
MI_API int mi_tail_dropout_masks(const int64_t *seed, int32_t nlayers, const int64_t *salts, const float *ps,
                                 const int32_t *lds, uint8_t *const *bits, int32_t M, void *stream);
/* the same launch also zero-fills zero_buf[0 .. zero_floats) (a multiple of 4 floats, 16-byte aligned): the backward pass's
 * accumulation buffer (split-K weight gradients) without a fill launch of its own */
MI_API int mi_tail_dropout_masks_z(const int64_t *seed, int32_t nlayers, const int64_t *salts, const float *ps,
                                   const int32_t *lds, uint8_t *const *bits, int32_t M, float *zero_buf,
                                   int64_t zero_floats, void *stream);
MI_API int mi_tail_fwd_gemm(const float *X, int32_t ldx, const float *x_mu, const float *x_sc, const float *x_be,
                            float x_p, const uint8_t *x_keep, const float *W, int32_t ldw, float *Z, int32_t ldz,
                            float *part, float *a_out, int32_t M, int32_t N, int32_t K, void *stream);
MI_API int64_t mi_tail_part_elems(int32_t M, int32_t N);
MI_API int mi_tail_bn_finalize_fwd(const float *part, int32_t M, int32_t N, const float *gamma, const float *beta,
                                   const float *mean_offset, float *running_mean, float *running_var, float momentum,
                                   float eps, int64_t *num_batches_tracked, int64_t *seed_bump, float *mu, float *sc,
                                   float *be, float *rstd, void *stream);
/* mi_tail_bn_finalize_fwd with the work of mi_tail_dropout_masks_z riding along in extra workgroups of the same launch
 * (ride == NULL: the plain entry point).  The layer whose statistics are joined is the FIRST of the step, so every kernel
 * that reads keep bits comes later.  seed_bump must not point at ride->seed (MI_ERR_INVALID_ARG): a launch cannot both
 * draw from the seed and advance it; let a later finalize launch bump it.  Host struct, read at call time. */
/* The arguments of mi_tail_affine_consts (below) as a host struct: a second job a carrying launch can take. */
typedef struct mi_tail_affine_job {
  int32_t nlayers;                          /* <= 8 */
  const int32_t *widths;                    /* host [nlayers] */
  const float *const *gamma, *const *beta, *const *running_mean, *const *running_var, *const *bias;   /* host arrays of device pointers */
  const float *eps;                         /* host [nlayers] */
  float *const *mu, *const *sc, *const *be, *const *rstd;                                             /* host arrays of device pointers */
} mi_tail_affine_job;
typedef struct mi_tail_mask_ride {
  const int64_t *seed;          /* device */
  int32_t nlayers;              /* <= 8 */
  int32_t M;
  const int64_t *salts;         /* host [nlayers] */
  const float *ps;              /* host [nlayers] */
  const int32_t *lds;           /* host [nlayers] */
  uint8_t *const *bits;         /* host [nlayers] of device pointers (NULL where ps == 0) */
  float *zero_buf;              /* device, nullable */
  int64_t zero_floats;
  const mi_tail_affine_job *affine;   /* nullable: the constants of the tail's fixed-statistics layers computed by further extra
                                         workgroups (read by mi_gather_fm_fwd_ride only: an inference forward is the lookup, the
                                         products and the head — no launch for the constants) */
} mi_tail_mask_ride;
MI_API int mi_tail_bn_finalize_fwd_r(const float *part, int32_t M, int32_t N, const float *gamma, const float *beta,
                                     const float *mean_offset, float *running_mean, float *running_var, float momentum,
                                     float eps, int64_t *num_batches_tracked, int64_t *seed_bump, float *mu, float *sc,
                                     float *be, float *rstd, const mi_tail_mask_ride *ride, void *stream);
/* The gather + FM forward (mi_gather_fm_fwd_sum) carrying the same job in extra workgroups at the end of its grid: in
 * DeepFM's fused step (src/models/deepfm.py:79-105) it is the first kernel of the step — every reader of the keep bits
 * and every adder into zero_buf is launched later — so the tail needs no mask launch and no finalize launch to carry one.
 * ride == NULL: mi_gather_fm_fwd_sum.  offsets may be NULL (ids used as they are: the sharded step's slot lookup, W / w1 the
 * receive buffer and its column D, ldw = ldw1 = D + 4).  A gather form without the extra workgroups (D % 4, F > 64, unaligned) runs the job
 * as a launch of its own first.  The seed must be advanced by a LATER kernel. */
MI_API int mi_gather_fm_fwd_ride(const int64_t *idx, const int64_t *offsets, const float *W, int64_t ldw, const float *w1,
                                 int64_t ldw1, const float *bias, float *emb_out, float *yfm_out, int64_t *rows_out,
                                 float *sum_out, int64_t B, int32_t F, int32_t D, int64_t N, int32_t *err,
                                 const mi_tail_mask_ride *ride, void *stream);
MI_API int mi_tail_head_fwd(const float *Z, int32_t ldz, const float *mu, const float *sc, const float *be, float p,
                            const uint8_t *keep, const float *w, const float *b, const float *add, float *out,
                            int32_t M, int32_t N, void *stream);
MI_API int32_t mi_tail_head_blocks(int32_t M);
MI_API int mi_tail_head_bwd(const float *Z, int32_t ldz, const float *mu, const float *sc, const float *be, float p,
                            const uint8_t *keep, const float *g, const float *w, float *DY, float *part, float *wpart,
                            int32_t M, int32_t N, void *stream);
/* sum_reps > 0: part is fp32[sum_reps, N, 2] and wpart fp32[sum_reps, N + 4], both ZEROED by the caller; workgroup b ADDS
 * its sums into row b % sum_reps (float atomics) — a mi_tail_bn_bwd with nblk = nwblk = sum_reps then joins them in the next
 * product's prologue and the head level needs no finalize launch.  sum_reps = 0 is mi_tail_head_bwd. */
MI_API int mi_tail_head_bwd_s(const float *Z, int32_t ldz, const float *mu, const float *sc, const float *be, float p,
                              const uint8_t *keep, const float *g, const float *w, float *DY, float *part, float *wpart,
                              int32_t sum_reps, int32_t M, int32_t N, void *stream);
/* Head + criterion + head backward in ONE launch, for a step whose labels are known at forward time and whose criterion is
 * BCE-with-logits, reduction "mean" (the reference trainer's: src/trainer/deepfm.py:32,51 around the head of
 * src/models/deepfm.py:100-105).  Per row m: out[m] = the logit (mi_tail_head_fwd_m), its loss term, g[m] =
 * (sigmoid(out[m]) - y[m]) / M (mi_bce_logits_fwd's dx_unit), DY / part / wpart as mi_tail_head_bwd_s(g) writes them
 * (sum_reps > 0 rows, zeroed by the caller).  loss_ws: fp32[mi_tail_head_bce_ws_elems(sum_reps)], ZEROED by the caller;
 * loss_ws[0] is the loss afterwards (the rest are per-replica partial sums and arrival counts).  g and the sums are those
 * of the upstream gradient upstream[0] — a device scalar, the one the caller seeds the backward with (NULL: exactly 1; a
 * table-sharded step seeds 1 / world) — a backward that arrives with another gradient calls mi_tail_head_bwd_s (after
 * zeroing part / wpart again).  N <= 512.  stats as in mi_tail_head_fwd_m (NULL: mu / sc / be are read). */
MI_API int32_t mi_tail_head_bce_ws_elems(int32_t sum_reps);
MI_API int mi_tail_head_bce(const float *Z, int32_t ldz, const float *mu, const float *sc, const float *be, float p,
                            const uint8_t *keep, const float *w, const float *b, const float *add, const float *y,
                            float *out, float *g, float *DY, float *part, float *wpart, int32_t sum_reps, float *loss_ws,
                            int32_t M, int32_t N, const mi_tail_bn_fwd *stats, const float *upstream, void *stream);
MI_API int mi_tail_bn_finalize_bwd(const float *part, int32_t nblk, int32_t M, int32_t N, const float *gamma,
                                   const float *rstd, float *dgamma, float *dbeta, float *al, float *bz, float *de,
                                   const float *wpart, int32_t nwblk, float *dw, float *db, void *stream);
/* Layers that normalise with FIXED statistics or not at all — eval mode (model.eval(): BatchNorm1d on its running
 * statistics, src/models/deepfm.py:57-58) and use_batchnorm=False (DeepFM's default, src/models/deepfm.py:20,56) — run on
 * the same products: a = relu((z - mu) sc + be) with z = x W^T (the Linear bias stays out of the product) and
 *   BatchNorm in eval mode: rstd = 1/sqrt(running_var + eps), mu = running_mean - bias, sc = gamma rstd, be = beta
 *   no BatchNorm:           mu = 0, sc = 1, be = bias, rstd = 1
 * mi_tail_affine_consts writes these for up to 8 layers in ONE launch (host arrays of device pointers; running_mean[l] ==
 * NULL: layer l has no BatchNorm; gamma / beta / bias nullable per layer).
 * mi_tail_bn_finalize_bwd_a: affine != 0 -> al = gamma rstd, bz = de = 0 (dz = al dy), dgamma = rstd sum dy (z - mu),
 *   dbeta = sum dy, dbias (nullable) = al sum dy = the Linear bias's gradient; affine == 0 is mi_tail_bn_finalize_bwd.
 * mi_tail_head_fwd_m with a mi_tail_bn_fwd whose `part` is NULL joins nothing and only advances *seed_bump (nullable): how
 *   a step without any statistics to join moves its dropout seed on. */
MI_API int mi_tail_affine_consts(int32_t nlayers, const int32_t *widths, const float *const *gamma,
                                 const float *const *beta, const float *const *running_mean,
                                 const float *const *running_var, const float *const *bias, const float *eps,
                                 float *const *mu, float *const *sc, float *const *be, float *const *rstd, void *stream);
MI_API int mi_tail_bn_finalize_bwd_a(const float *part, int32_t nblk, int32_t M, int32_t N, const float *gamma,
                                     const float *rstd, float *dgamma, float *dbeta, float *al, float *bz, float *de,
                                     const float *wpart, int32_t nwblk, float *dw, float *db, int32_t affine,
                                     float *dbias, void *stream);
/* ... and with a second destination db2[0] (nullable) for the head bias's gradient (see mi_tail_bn_bwd.db2) */
MI_API int mi_tail_bn_finalize_bwd_b(const float *part, int32_t nblk, int32_t M, int32_t N, const float *gamma,
                                     const float *rstd, float *dgamma, float *dbeta, float *al, float *bz, float *de,
                                     const float *wpart, int32_t nwblk, float *dw, float *db, int32_t affine,
                                     float *dbias, float *db2, void *stream);
MI_API int mi_tail_dgrad_gemm(const float *DY, const float *Zl, int32_t ld, const float *mu, const float *al,
                              const float *bz, const float *de, const float *W, int32_t ldw, const float *pZ,
                              int32_t pld, const float *p_mu, const float *p_sc, const float *p_be, float p_p,
                              const uint8_t *p_keep, float *OUT, int32_t ldo, float *part, float *dz_out, int32_t M,
                              int32_t N, int32_t K, void *stream);
MI_API int32_t mi_tail_wgrad_splits(int32_t M, int32_t N, int32_t K);
MI_API int mi_tail_wgrad_gemm(const float *DY, const float *Zl, int32_t ld, const float *mu, const float *al,
                              const float *bz, const float *de, const float *pZ, int32_t pld, const float *p_mu,
                              const float *p_sc, const float *p_be, float p_p, const uint8_t *p_keep, float *slab,
                              float *dW, int32_t M, int32_t N, int32_t K, void *stream);

/* ---- a9: DHEmbedding's MLP (src/models/embeddings/dh_embedding.py:100-117,345-356) -----------------------------------
 * Per layer Linear -> Mish (use_bn 0) | Linear -> BatchNorm1d -> Mish (use_bn 2, the default) | Linear -> Mish ->
 * BatchNorm1d (use_bn 1, the LightGCN configs).  The contractions are mi_tail_fwd_gemm_s / mi_tail_dgrad_gemm_s on plain
 * operands (+ mi_gemm_f32_multi for dW); these three entry points are the column-wise element work between them, with
 * BatchNorm statistics / gradient column sums as per-64-row-tile partials in the layouts mi_tail_bn_finalize_fwd / _bwd join.
 *   mi_col_act_fwd: out[M,N] = ACT((in - mu[c]) sc[c] + be[c]), act 0 = identity, 1 = mish (x tanh(log1p(exp x)), as
 *     torch's CPU kernel); mu / sc / be nullable (0 / 1 / 0); in has row pitch ld; part (nullable,
 *     mi_tail_part_elems(M, N) floats) = (mean, M2) of `out` per tile and column.
 *   mi_col_act_bwd: dy = g ACT'(pre) (dy nullable: with act 0 it equals g), part[tile, N, 2] = (sum dy, sum dy (in - mu)).
 *   mi_bn_mish_bwd (use_bn 1: a = BatchNorm(m), m = mish(z + bias)): dz = (al g + bz (m - mu) + de) mish'(z + bias),
 *     part[tile, N, 2] = (sum dz, 0); al / bz / de as mi_tail_bn_finalize_bwd[_a] writes them. */
MI_API int mi_col_act_fwd(const float *in, int32_t ld, const float *mu, const float *sc, const float *be, int32_t act,
                          float *out, float *part, int32_t M, int32_t N, void *stream);
MI_API int mi_col_act_bwd(const float *g, const float *in, int32_t ld, const float *mu, const float *sc, const float *be,
                          int32_t act, float *dy, float *part, int32_t M, int32_t N, void *stream);
/* dz[M,N] = al dy + bz (z - mu) + de: the BatchNorm backward materialised (a layer whose input needs no gradient has no
 * input-gradient product that would compute it in its operand load) */
MI_API int mi_bn_dz(const float *dy, const float *z, int32_t ldz, const float *mu, const float *al, const float *bz,
                    const float *de, float *dz, int32_t M, int32_t N, void *stream);
MI_API int mi_bn_mish_bwd(const float *g, const float *m_act, const float *z, int32_t ldz, const float *bias,
                          const float *mu, const float *al, const float *bz, const float *de, float *dz, float *part,
                          int32_t M, int32_t N, void *stream);

/* ---- §8e: the sharded lookup's collectives on the CALLER's stream (recsys-benchmark_amd/csrc/comm.hip) ------------------
 * No reference counterpart.  The library owns its own RCCL communicator (resolved with dlopen at first use: every entry
 * point returns MI_ERR_UNSUPPORTED when no librccl can be loaded) so that the all-to-alls and the all-reduce are enqueued
 * on the stream the kernels run on — torch.distributed's process-group stream costs two event hand-offs per collective.
 * mi_comm_unique_id: HOST buffer of 128 bytes, filled on one rank and distributed by the caller.  mi_comm_init is
 * collective (every rank calls it with the same id; the current HIP device is the rank's GPU).  mi_comm_all_to_all:
 * recv[p*bytes .. ) <- rank p's send[me*bytes .. ) for every peer p (ncclGroupStart / Send / Recv / GroupEnd).
 * mi_comm_all_reduce_sum_f32: in place.  mi_comm_abort: for a communicator whose enqueued work does not complete
 * (ncclCommAbort: stops its kernels, frees it without waiting); mi_comm_destroy is the orderly end. */
MI_API int mi_comm_available(void);   /* MI_OK when librccl could be loaded in this process; agree on it before mi_comm_init */
MI_API int mi_comm_unique_id(char *id128);
MI_API int mi_comm_init(const char *id128, int32_t world, int32_t rank, void **comm_out);
MI_API int mi_comm_destroy(void *comm);
MI_API int mi_comm_abort(void *comm);
MI_API int mi_comm_all_to_all(void *comm, const void *send, void *recv, int64_t bytes_per_peer, void *stream);
MI_API int mi_comm_all_reduce_sum_f32(void *comm, float *buf, int64_t count, void *stream);

/* ---- profiling ring (bench.py's per-kernel HIP-event timing) ---------------
 * When enabled every launcher brackets its kernel with a hipEvent pair on the
 * launch stream.  Not for use under graph capture.
 */
MI_API int mi_prof_enable(int32_t capacity); /* >0: (re)arm with that many records; 0: disable */
MI_API int mi_prof_count(void);
/* An empty kernel of the given geometry through the same launcher / timing ring ("empty"): the duration the
 * per-dispatch clock reads for a launch that does nothing — the floor under every kernel time it reports. */
MI_API int mi_prof_empty_launch(int32_t grid, int32_t block, void *stream);
/* Synchronises on record i's stop event; name_out is a host buffer of >=64 bytes. */
MI_API int mi_prof_read(int32_t i, char *name_out, float *ms_out);

#ifdef __cplusplus
}
#endif
#endif /* MI355X_RECSYS_H */
