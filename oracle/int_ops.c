/* ORACLE — test infrastructure, NOT product code.
 *
 * Plain-C restatement of the integer / byte arithmetic on the hot path, for bit-exact checks
 * (indices and hash values must match the reference exactly; SURVEY.md §7 "Integer semantics").
 * Built by oracle/Makefile into oracle/liboracle_int.so and called from tests/ through ctypes.
 *
 *   oracle_dhe_hash      src/models/embeddings/dh_embedding.py:213-236
 *   oracle_csr_rows      src/models/embeddings/pruned_embedding.py:187-204 (numba CPU kernel K2)
 *   oracle_qr_split      src/models/embeddings/qr_embedding.py:95-97
 *   oracle_cerp_split    src/models/embeddings/cerp_embedding.py:152-153
 *   oracle_tt_split      src/models/embeddings/tensortrain_embeddings.py:137-143
 */
#include <stdint.h>
#include <string.h>

/* Python/torch floor-mod for int64 */
static int64_t floormod(int64_t a, int64_t p) {
  int64_t r = a % p;
  if (r != 0 && ((r < 0) != (p < 0))) r += p;
  return r;
}
static int64_t floordiv(int64_t a, int64_t b) {
  int64_t q = a / b;
  if ((a % b != 0) && ((a < 0) != (b < 0))) q -= 1;
  return q;
}

void oracle_dhe_hash(const int64_t *ids, int64_t n, const int64_t *slopes, const int64_t *bias,
                     const int64_t *primes, int64_t k, int64_t prefix, int64_t m, int64_t *h_out,
                     float *feat_out) {
  for (int64_t i = 0; i < n; ++i)
    for (int64_t j = 0; j < k; ++j) {
      /* two's-complement wrap like torch's int64 kernels */
      uint64_t prod = (uint64_t)slopes[j] * (uint64_t)(ids[i] + prefix + 1) + (uint64_t)bias[j];
      int64_t h = floormod(floormod((int64_t)prod, primes[j]), m);
      if (h_out) h_out[i * k + j] = h;
      if (feat_out) {
        float f = (float)h / (float)(m - 1); /* int64 / int -> fp32 true division */
        f = f * 2.0f;
        feat_out[i * k + j] = f - 1.0f;
      }
    }
}

void oracle_csr_rows(const float *values, const int64_t *crow, const int64_t *col, const int64_t *ids,
                     float *out, int64_t n, int64_t hidden) {
  memset(out, 0, (size_t)(n * hidden) * sizeof(float));
  for (int64_t i = 0; i < n; ++i)
    for (int64_t j = crow[ids[i]]; j < crow[ids[i] + 1]; ++j) out[i * hidden + col[j]] = values[j];
}

void oracle_qr_split(const int64_t *idx, int64_t n, int64_t divider, int64_t *rem, int64_t *quo) {
  for (int64_t i = 0; i < n; ++i) {
    rem[i] = floormod(idx[i], divider);
    quo[i] = floordiv(idx[i], divider);
  }
}

void oracle_cerp_split(const int64_t *idx, int64_t n, int64_t q_entity_per_row, int64_t bucket, int64_t *q,
                       int64_t *p) {
  for (int64_t i = 0; i < n; ++i) {
    q[i] = idx[i] / q_entity_per_row; /* rounding_mode="trunc" */
    p[i] = floormod(idx[i], bucket);
  }
}

/* mixed-radix split of an id over tt_p_shapes: out[c*n + i] = digit c of idx[i] */
void oracle_tt_split(const int64_t *idx, int64_t n, const int64_t *p_shapes, int64_t ncores, int64_t *out) {
  for (int64_t i = 0; i < n; ++i) {
    int64_t big = 1, rest = idx[i];
    for (int64_t c = 0; c < ncores; ++c) big *= p_shapes[c];
    for (int64_t c = 0; c < ncores; ++c) {
      big /= p_shapes[c];
      out[c * n + i] = floordiv(rest, big);
      rest = floormod(rest, big);
    }
  }
}
