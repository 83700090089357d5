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

#define MI_ABI_VERSION 1

#define MI_OK 0
#define MI_ERR_INVALID_ARG (-1)  /* null pointer, negative size, bad enum */
#define MI_ERR_UNSUPPORTED (-2)  /* shape outside what the kernels cover */
#define MI_ERR_LAUNCH (-3)       /* hipLaunch / hipGetLastError failed */
#define MI_ERR_STATE (-4)        /* profiling ring misuse */

/* bits OR-ed into *err by the kernels */
#define MI_IDX_OUT_OF_RANGE 1

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

/* Backward of the above, row-sparse form (the MI355X-native default):
 *   gvals[b,f,:] = g_emb[b,f,:] + g_y[b] * (S_b - emb[b,f,:]),  S_b = sum_f emb
 *   g1vals[b,f]  = g_y[b]
 * i.e. one gradient row per lookup, written with plain coalesced stores; the
 * caller pairs them with rows_out from the forward as an (uncoalesced) COO
 * gradient of W / w1.  Autograd equivalent: embedding_dense_backward +
 * _embedding_bag_dense_backward of src/models/deepfm.py:89,95 before the
 * duplicate-row sum.  g_emb is nullable (no deep branch).
 */
MI_API int mi_gather_fm_bwd_rows(const float *emb, const float *g_y,
                                 const float *g_emb, float *gvals, float *g1vals,
                                 int64_t B, int32_t F, int32_t D, void *stream);

/* Backward, dense form (the reference's weight.grad semantics): scatter-adds
 * the same rows into caller-zeroed gW fp32[N,D] / gw1 fp32[N] with float
 * atomics (sum order differs from the CPU index_add: fp32 tolerance).
 * rows int64[B,F] = idx+offsets as produced by the forward.
 */
MI_API int mi_gather_fm_bwd_dense(const int64_t *rows, const float *emb,
                                  const float *g_y, const float *g_emb,
                                  float *gW, float *gw1, int64_t B, int32_t F,
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

/* ---- profiling ring (bench.py's per-kernel HIP-event timing) ---------------
 * When enabled every launcher brackets its kernel with a hipEvent pair on the
 * launch stream.  Not for use under graph capture.
 */
MI_API int mi_prof_enable(int32_t capacity); /* >0: (re)arm with that many records; 0: disable */
MI_API int mi_prof_count(void);
/* Synchronises on record i's stop event; name_out is a host buffer of >=64 bytes. */
MI_API int mi_prof_read(int32_t i, char *name_out, float *ms_out);

#ifdef __cplusplus
}
#endif
#endif /* MI355X_RECSYS_H */
