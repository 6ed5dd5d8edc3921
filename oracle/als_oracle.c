/*
 * oracle/als_oracle.c -- CPU restatement of the reference's ALS hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity *checker* for the HIP
 * kernels in implicit_amd/csrc/.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load the library built from it; the product
 * package never links, imports or falls back to it.
 *
 * Each function restates one routine of benfred/implicit's Cython CPU path
 * (paths relative to /root/reference):
 *
 *   oracle_gramian              np.dot(Y.T, Y)            implicit/cpu/_als.pyx:70,164,268
 *   oracle_least_squares_cg     _least_squares_cg         implicit/cpu/_als.pyx:152-248
 *   oracle_least_squares_chol   _least_squares            implicit/cpu/_als.pyx:75-142
 *   oracle_calculate_loss       _calculate_loss           implicit/cpu/_als.pyx:257-308
 *   oracle_select               implicit::select<float>   implicit/cpu/select.h:12-40
 *   oracle_topk                 topk/_topk_batch          implicit/cpu/topk.pyx:15-67
 *
 * The reference reaches BLAS/LAPACK (sdot, saxpy, ssymv, sscal, sposv, sgemm)
 * through SciPy (scipy-openblas 0.3.29 in this image); here they are plain fp32
 * loops, so results agree with the reference to fp32 summation-order noise, not
 * bit for bit.  PINNING: tests/test_oracle_pin.py checks every function against
 * (a) golden vectors produced by the *compiled reference itself*
 * (tests/golden/, generator tests/golden/make_golden.py) and (b) the compiled
 * reference in oracle/_ref when present.
 *
 * Plain C99 + optional OpenMP.  All matrices row-major fp32, indices int32
 * (indptr int64-safe variant not needed at oracle sizes).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* fp32 dot product with 16 interleaved partial sums combined by a pairwise tree -- the
 * summation shape of a SIMD BLAS sdot (the reference reaches OpenBLAS's AVX-512 kernels); a
 * single sequential fp32 accumulator is ~10x noisier on the all-positive cold-start factors. */
static inline float dotf(const float *a, const float *b, int n) {
  float acc[16] = {0};
  int i = 0;
  for (; i + 16 <= n; i += 16)
    for (int l = 0; l < 16; ++l) acc[l] += a[i + l] * b[i + l];
  for (int l = 0; i < n; ++i, ++l) acc[l] += a[i] * b[i];
  for (int w = 8; w >= 1; w >>= 1)
    for (int l = 0; l < w; ++l) acc[l] += acc[l + w];
  return acc[0];
}

static inline void axpyf(float alpha, const float *x, float *y, int n) {
  for (int i = 0; i < n; ++i) y[i] += alpha * x[i];
}

/* y = alpha * A x with A symmetric f x f (reference: ssymv 'U', beta = 0) */
static inline void symvf(float alpha, const float *A, const float *x, float *y, int f) {
  for (int i = 0; i < f; ++i) y[i] = alpha * dotf(A + (size_t)i * f, x, f);
}

static int pick_threads(int num_threads) {
#ifdef _OPENMP
  return num_threads > 0 ? num_threads : omp_get_max_threads();
#else
  (void)num_threads;
  return 1;
#endif
}

/* out[f x f] = Y^T Y, Y is rows x f.  fp32 result of an fp64 accumulation. */
void oracle_gramian(const float *Y, int64_t rows, int f, float *out) {
  double *acc = (double *)calloc((size_t)f * f, sizeof(double));
  for (int64_t r = 0; r < rows; ++r) {
    const float *y = Y + r * f;
    for (int i = 0; i < f; ++i) {
      double yi = y[i];
      for (int j = i; j < f; ++j) acc[(size_t)i * f + j] += yi * (double)y[j];
    }
  }
  for (int i = 0; i < f; ++i)
    for (int j = i; j < f; ++j) out[(size_t)i * f + j] = out[(size_t)j * f + i] = (float)acc[(size_t)i * f + j];
  free(acc);
}

/*
 * Conjugate-gradient half sweep, implicit/cpu/_als.pyx:152-248.
 * A0 = YtY + reg*I is passed in already regularised (the reference forms it at
 * _als.pyx:164; the GPU boundary, als.cu:154, receives it regularised too).
 * X is updated in place (warm start).
 */
void oracle_least_squares_cg(const int32_t *indptr, const int32_t *indices, const float *data,
                             int64_t rows, float *X, const float *Y, const float *A0, int f,
                             int cg_steps, int num_threads) {
  int nt = pick_threads(num_threads);
  (void)nt;
#pragma omp parallel num_threads(nt)
  {
    float *Ap = (float *)malloc(sizeof(float) * f);
    float *p = (float *)malloc(sizeof(float) * f);
    float *r = (float *)malloc(sizeof(float) * f);
#pragma omp for schedule(dynamic, 8)
    for (int64_t u = 0; u < rows; ++u) {
      float *x = X + u * f;
      if (indptr[u] == indptr[u + 1]) { /* :182-184 */
        memset(x, 0, sizeof(float) * f);
        continue;
      }
      symvf(-1.0f, A0, x, r, f); /* :187-188 */
      for (int32_t k = indptr[u]; k < indptr[u + 1]; ++k) { /* :190-201 */
        const float *y = Y + (size_t)indices[k] * f;
        float confidence = data[k], temp;
        if (confidence > 0) {
          temp = confidence;
        } else {
          temp = 0;
          confidence = -1 * confidence;
        }
        temp = temp - (confidence - 1) * dotf(y, x, f);
        axpyf(temp, y, r, f);
      }
      memcpy(p, r, sizeof(float) * f);
      float rsold = dotf(r, r, f); /* :204 */
      if (rsold < 1e-20f) continue; /* :206 */
      for (int it = 0; it < cg_steps; ++it) {
        symvf(1.0f, A0, p, Ap, f); /* :211-212 */
        for (int32_t k = indptr[u]; k < indptr[u + 1]; ++k) { /* :214-222 */
          const float *y = Y + (size_t)indices[k] * f;
          float confidence = data[k];
          if (confidence < 0) confidence = -1 * confidence;
          float temp = (confidence - 1) * dotf(y, p, f);
          axpyf(temp, y, Ap, f);
        }
        float alpha = rsold / dotf(p, Ap, f); /* :225 */
        axpyf(alpha, p, x, f);                /* :228 */
        axpyf(-alpha, Ap, r, f);              /* :231-232 */
        float rsnew = dotf(r, r, f);          /* :234 */
        if (rsnew < 1e-20f) break;            /* :235 */
        float beta = rsnew / rsold;           /* :239-242 */
        for (int i = 0; i < f; ++i) p[i] = beta * p[i] + r[i];
        rsold = rsnew;
      }
    }
    free(Ap);
    free(p);
    free(r);
  }
}

/*
 * The same CG half sweep carried out in fp64 from the same fp32 inputs (the reference's
 * `floating = double` instantiation of _als.pyx:152-248 fed with up-cast factors).  Used by the
 * parity tests to measure the fp32 oracle's own rounding noise: a GPU result is accepted when
 * its distance to this fp64 answer is within the tolerance or the oracle's own distance to it.
 * A0 is rebuilt in fp64 from Y (+ reg on the diagonal); X64 (rows x f doubles) is in/out.
 */
void oracle_least_squares_cg_f64(const int32_t *indptr, const int32_t *indices, const float *data,
                                 int64_t rows, double *X64, const float *Y, int64_t ycount, int f,
                                 double regularization, int cg_steps, int num_threads) {
  int nt = pick_threads(num_threads);
  (void)nt;
  double *A0 = (double *)calloc((size_t)f * f, sizeof(double));
  for (int64_t r = 0; r < ycount; ++r) {
    const float *y = Y + r * f;
    for (int i = 0; i < f; ++i)
      for (int j = 0; j < f; ++j) A0[(size_t)i * f + j] += (double)y[i] * (double)y[j];
  }
  for (int i = 0; i < f; ++i) A0[(size_t)i * f + i] += regularization;
#pragma omp parallel num_threads(nt)
  {
    double *Ap = (double *)malloc(sizeof(double) * f);
    double *p = (double *)malloc(sizeof(double) * f);
    double *r = (double *)malloc(sizeof(double) * f);
#pragma omp for schedule(dynamic, 8)
    for (int64_t u = 0; u < rows; ++u) {
      double *x = X64 + u * f;
      if (indptr[u] == indptr[u + 1]) {
        memset(x, 0, sizeof(double) * f);
        continue;
      }
      for (int i = 0; i < f; ++i) {
        double s = 0;
        for (int j = 0; j < f; ++j) s += A0[(size_t)i * f + j] * x[j];
        r[i] = -s;
      }
      for (int32_t k = indptr[u]; k < indptr[u + 1]; ++k) {
        const float *y = Y + (size_t)indices[k] * f;
        double confidence = data[k], temp, d = 0;
        if (confidence > 0) {
          temp = confidence;
        } else {
          temp = 0;
          confidence = -confidence;
        }
        for (int i = 0; i < f; ++i) d += (double)y[i] * x[i];
        temp = temp - (confidence - 1) * d;
        for (int i = 0; i < f; ++i) r[i] += temp * (double)y[i];
      }
      double rsold = 0;
      for (int i = 0; i < f; ++i) {
        p[i] = r[i];
        rsold += r[i] * r[i];
      }
      if (rsold < 1e-20) continue;
      for (int it = 0; it < cg_steps; ++it) {
        for (int i = 0; i < f; ++i) {
          double s = 0;
          for (int j = 0; j < f; ++j) s += A0[(size_t)i * f + j] * p[j];
          Ap[i] = s;
        }
        for (int32_t k = indptr[u]; k < indptr[u + 1]; ++k) {
          const float *y = Y + (size_t)indices[k] * f;
          double confidence = fabs((double)data[k]), d = 0;
          for (int i = 0; i < f; ++i) d += (double)y[i] * p[i];
          double temp = (confidence - 1) * d;
          for (int i = 0; i < f; ++i) Ap[i] += temp * (double)y[i];
        }
        double pAp = 0;
        for (int i = 0; i < f; ++i) pAp += p[i] * Ap[i];
        double alpha = rsold / pAp, rsnew = 0;
        for (int i = 0; i < f; ++i) {
          x[i] += alpha * p[i];
          r[i] -= alpha * Ap[i];
          rsnew += r[i] * r[i];
        }
        if (rsnew < 1e-20) break;
        double beta = rsnew / rsold;
        for (int i = 0; i < f; ++i) p[i] = beta * p[i] + r[i];
        rsold = rsnew;
      }
    }
    free(Ap);
    free(p);
    free(r);
  }
  free(A0);
}

/* In-place upper Cholesky A = U^T U of a row-major symmetric matrix; solves A x = b into b.
 * Returns 0, or i+1 when the leading minor of order i+1 is not positive definite
 * (LAPACK sposv's info convention, _als.pyx:127,131-138). */
static int chol_solve(float *A, float *b, int f) {
  for (int j = 0; j < f; ++j) {
    float d = A[(size_t)j * f + j];
    for (int k = 0; k < j; ++k) d -= A[(size_t)k * f + j] * A[(size_t)k * f + j];
    if (!(d > 0.f)) return j + 1;
    d = sqrtf(d);
    A[(size_t)j * f + j] = d;
    for (int i = j + 1; i < f; ++i) {
      float s = A[(size_t)j * f + i];
      for (int k = 0; k < j; ++k) s -= A[(size_t)k * f + j] * A[(size_t)k * f + i];
      A[(size_t)j * f + i] = s / d;
    }
  }
  /* U^T z = b */
  for (int i = 0; i < f; ++i) {
    float s = b[i];
    for (int k = 0; k < i; ++k) s -= A[(size_t)k * f + i] * b[k];
    b[i] = s / A[(size_t)i * f + i];
  }
  /* U x = z */
  for (int i = f - 1; i >= 0; --i) {
    float s = b[i];
    for (int k = i + 1; k < f; ++k) s -= A[(size_t)i * f + k] * b[k];
    b[i] = s / A[(size_t)i * f + i];
  }
  return 0;
}

/*
 * Cholesky half sweep, implicit/cpu/_als.pyx:75-142.
 * YtY is passed UNregularised; reg is a double added to the diagonal (:85).
 * The previous X[u] is ignored (cold solve).  Returns 0, or 1 + the first row
 * whose posv failed with *err_out = posv's info (:131-138 raises ValueError).
 */
int64_t oracle_least_squares_chol(const float *YtY, const int32_t *indptr, const int32_t *indices,
                                  const float *data, int64_t rows, float *X, const float *Y, int f,
                                  double regularization, int num_threads, int *err_out) {
  int nt = pick_threads(num_threads);
  (void)nt;
  float *initialA = (float *)malloc(sizeof(float) * f * f);
  for (int i = 0; i < f; ++i)
    for (int j = 0; j < f; ++j)
      initialA[(size_t)i * f + j] = (float)((double)YtY[(size_t)i * f + j] + (i == j ? regularization : 0.0));
  int64_t failed = 0;
  int failed_err = 0;
#pragma omp parallel num_threads(nt)
  {
    float *A = (float *)malloc(sizeof(float) * f * f);
    float *b = (float *)malloc(sizeof(float) * f);
#pragma omp for schedule(dynamic, 8)
    for (int64_t u = 0; u < rows; ++u) {
      if (indptr[u] == indptr[u + 1]) { /* :98-100 */
        memset(X + u * f, 0, sizeof(float) * f);
        continue;
      }
      memcpy(A, initialA, sizeof(float) * f * f);
      memset(b, 0, sizeof(float) * f);
      for (int32_t k = indptr[u]; k < indptr[u + 1]; ++k) { /* :108-124 */
        const float *y = Y + (size_t)indices[k] * f;
        float confidence = data[k];
        if (confidence > 0)
          axpyf(confidence, y, b, f);
        else
          confidence = -1 * confidence;
        for (int j = 0; j < f; ++j) {
          float temp = (confidence - 1) * y[j];
          axpyf(temp, y, A + (size_t)j * f, f);
        }
      }
      int err = chol_solve(A, b, f); /* :127 */
      if (!err) {
        memcpy(X + u * f, b, sizeof(float) * f); /* :130 */
      } else {
#pragma omp critical
        if (!failed || u + 1 < failed) {
          failed = u + 1;
          failed_err = err;
        }
      }
    }
    free(A);
    free(b);
  }
  free(initialA);
  if (err_out) *err_out = failed_err;
  return failed;
}

/*
 * Training loss, implicit/cpu/_als.pyx:257-308.  YtY here is UNregularised (:268).
 * Accumulators are double as in the reference (:272).
 */
double oracle_calculate_loss(const int32_t *indptr, const int32_t *indices, const float *data,
                             int64_t users, int64_t items, int64_t nnz, const float *X,
                             const float *Y, const float *YtY, int f, float regularization,
                             int num_threads) {
  int nt = pick_threads(num_threads);
  (void)nt;
  double loss = 0, total_confidence = 0, item_norm = 0, user_norm = 0;
#pragma omp parallel num_threads(nt) reduction(+ : loss, total_confidence, item_norm, user_norm)
  {
    float *r = (float *)malloc(sizeof(float) * f);
#pragma omp for schedule(dynamic, 8)
    for (int64_t u = 0; u < users; ++u) {
      const float *x = X + u * f;
      symvf(1.0f, YtY, x, r, f); /* :280 */
      for (int32_t k = indptr[u]; k < indptr[u + 1]; ++k) {
        const float *y = Y + (size_t)indices[k] * f;
        float confidence = data[k], temp;
        if (confidence > 0) {
          temp = -2 * confidence;
        } else {
          temp = 0;
          confidence = -1 * confidence;
        }
        temp = temp + (confidence - 1) * dotf(y, x, f);
        axpyf(temp, y, r, f);
        total_confidence += confidence;
        loss += confidence;
      }
      loss += dotf(r, x, f);
      user_norm += dotf(x, x, f);
    }
#pragma omp for schedule(dynamic, 8)
    for (int64_t i = 0; i < items; ++i) item_norm += dotf(Y + i * f, Y + i * f, f);
    free(r);
  }
  loss += regularization * (item_norm + user_norm);
  return loss / (total_confidence + (double)users * (double)items - (double)nnz);
}

/* ---- top-k selection: implicit/cpu/select.h:12-40 -------------------------------------------
 * A size-k min-heap of (score, col) pairs under std::greater<pair> (lexicographic).  A candidate
 * enters iff the heap is not full or score > heap_min.score (STRICT, :23); eviction removes the
 * lexicographically smallest pair; output is sorted descending by (score, col) (:33).  The min
 * pair is unique (cols are unique) so the retained set does not depend on the heap's internal
 * layout and this restatement is exact, boundary ties included.
 */
typedef struct {
  float score;
  int col;
} pair_t;

static inline int pair_less(pair_t a, pair_t b) { /* a < b lexicographically */
  return a.score < b.score || (!(b.score < a.score) && a.col < b.col);
}

static void sift_down(pair_t *h, int n, int i) {
  for (;;) {
    int l = 2 * i + 1, r = l + 1, m = i;
    if (l < n && pair_less(h[l], h[m])) m = l;
    if (r < n && pair_less(h[r], h[m])) m = r;
    if (m == i) return;
    pair_t t = h[i];
    h[i] = h[m];
    h[m] = t;
    i = m;
  }
}

static void sift_up(pair_t *h, int i) {
  while (i > 0) {
    int parent = (i - 1) / 2;
    if (!pair_less(h[i], h[parent])) return;
    pair_t t = h[i];
    h[i] = h[parent];
    h[parent] = t;
    i = parent;
  }
}

void oracle_select(const float *batch, int rows, int cols, int k, int32_t *ids, float *distances) {
  pair_t *h = (pair_t *)malloc(sizeof(pair_t) * (size_t)(k > 0 ? k : 1));
  for (int row = 0; row < rows; ++row) {
    int n = 0;
    const float *s = batch + (size_t)row * cols;
    for (int col = 0; col < cols; ++col) {
      float score = s[col];
      if (n < k || score > h[0].score) {
        if (n >= k) { /* pop the min pair */
          h[0] = h[n - 1];
          --n;
          sift_down(h, n, 0);
        }
        h[n].score = score;
        h[n].col = col;
        sift_up(h, n);
        ++n;
      }
    }
    /* sort_heap under greater<> => descending (score, col): repeatedly extract the min to the back */
    for (int m = n; m > 1; --m) {
      pair_t t = h[0];
      h[0] = h[m - 1];
      h[m - 1] = t;
      sift_down(h, m - 1, 0);
    }
    for (int i = 0; i < n; ++i) { /* entries i >= n stay untouched (topk.pyx:20-21 zero-fills them) */
      ids[(size_t)row * k + i] = h[i].col;
      distances[(size_t)row * k + i] = h[i].score;
    }
  }
  free(h);
}

/*
 * topk: implicit/cpu/topk.pyx:15-67.  scores = query . items^T (:47); optional divide by
 * item_norms (:48-49); per-query filter (CSR liked-items: filt_indptr/filt_indices, :52-54) and
 * global item filter (:55-56) set to -FLT_MAX (:51); then select.  ids/distances must be
 * zero-initialised by the caller (:20-21).
 */
void oracle_topk(const float *items, int n_items, const float *query, int n_query, int f, int k,
                 const float *item_norms, const int32_t *filt_indptr, const int32_t *filt_indices,
                 const int32_t *filter_items, int n_filter_items, int32_t *ids, float *distances,
                 int num_threads) {
  int nt = pick_threads(num_threads);
  (void)nt;
#pragma omp parallel num_threads(nt)
  {
    float *scores = (float *)malloc(sizeof(float) * (size_t)n_items);
#pragma omp for schedule(dynamic, 4)
    for (int q = 0; q < n_query; ++q) {
      const float *qv = query + (size_t)q * f;
      for (int i = 0; i < n_items; ++i) {
        float s = dotf(qv, items + (size_t)i * f, f);
        if (item_norms) s = s / item_norms[i];
        scores[i] = s;
      }
      if (filt_indptr)
        for (int32_t j = filt_indptr[q]; j < filt_indptr[q + 1]; ++j) scores[filt_indices[j]] = -FLT_MAX;
      for (int j = 0; j < n_filter_items; ++j) scores[filter_items[j]] = -FLT_MAX;
      oracle_select(scores, 1, n_items, k, ids + (size_t)q * k, distances + (size_t)q * k);
    }
    free(scores);
  }
}

/* row L2 norms with zeros replaced by 1e-10: cpu/matrix_factorization_base.py:233-247 */
void oracle_norms(const float *Y, int64_t rows, int f, float *out) {
  for (int64_t r = 0; r < rows; ++r) {
    float n = sqrtf(dotf(Y + r * f, Y + r * f, f));
    out[r] = n == 0.f ? 1e-10f : n;
  }
}

int oracle_num_threads(void) { return pick_threads(0); }
