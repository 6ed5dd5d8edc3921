// K2: Cholesky half sweep  X[u] = (YtY + reg I + Y_u^T (C_u - I) Y_u)^-1  Y_u^T C_u p_u
//
// NEW on the GPU side (the reference's CUDA path is CG-only, implicit/gpu/als.cu); restates the CPU
// oracle implicit/cpu/_als.pyx:75-142: A = YtY + reg*I, b = sum_{c>0} c*y,
// A += (|c|-1) y y^T, posv, cold solve (previous X ignored), empty rows -> 0, a non-positive pivot
// reports the row (the oracle raises ValueError there, _als.pyx:131-138).
//
// One workgroup per row.  The (f+1) x (f+1) AUGMENTED lower triangle [A b; b^T .] lives in LDS with an
// odd leading dimension (bank-conflict-free column walks); factoring its first f columns leaves
// z = L^-1 b in the last row for free, and the back substitution L^T x = z is done by one wavefront
// with readlane broadcasts (no block barriers).  Gathered factor rows are staged through LDS in
// tiles of TILE rows, each a fully coalesced read.
#include "common.h"

namespace imp {

constexpr int kCholTile = 8;

__device__ __forceinline__ float bcast_lane(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

__global__ __launch_bounds__(256) void als_cholesky_kernel(const int32_t *__restrict__ order, int first, int count,
                                                           const int32_t *__restrict__ indptr,
                                                           const int32_t *__restrict__ indices,
                                                           const float *__restrict__ data, float *__restrict__ X,
                                                           const float *__restrict__ Y, const float *__restrict__ YtY,
                                                           int f, float reg, int lda, unsigned long long *failed_row) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *A = smem;                              // [(f+1)][lda] lower triangle used, row f = b^T -> z^T
  float *yt = A + (size_t)(f + 1) * lda;        // [TILE][f]   gathered rows
  float *ut = yt + (size_t)kCholTile * f;       // [TILE][f+1] (|c|-1) * y, last = c+
  const int tid = threadIdx.x;
  const int ti = tid >> 4, tj = tid & 15;

  for (int ri = blockIdx.x; ri < count; ri += gridDim.x) {
    const int u = order[first + ri];
    const int row_begin = indptr[u], row_end = indptr[u + 1];

    // A = YtY + reg I (lower triangle), b = 0
    for (int i = ti; i <= f; i += 16)
      for (int j = tj; j <= i && j < f; j += 16)
        A[i * lda + j] = i < f ? YtY[(size_t)i * f + j] + (i == j ? reg : 0.f) : 0.f;
    __syncthreads();

    for (int k0 = row_begin; k0 < row_end; k0 += kCholTile) {
      const int cnt = min(kCholTile, row_end - k0);
      // stage the tile: one wave per gathered row -> coalesced
      for (int e = tid; e < kCholTile * f; e += 256) {
        int t = e / f, c = e - t * f;
        float yv = 0.f, uv = 0.f;
        if (t < cnt) {
          float conf = data[k0 + t];
          float a = conf > 0.f ? conf : -conf;
          yv = Y[(size_t)indices[k0 + t] * f + c];
          uv = (a - 1.f) * yv;
        }
        yt[t * f + c] = yv;
        ut[t * (f + 1) + c] = uv;
      }
      if (tid < kCholTile) {
        float conf = tid < cnt ? data[k0 + tid] : 0.f;
        ut[tid * (f + 1) + f] = conf > 0.f ? conf : 0.f;
      }
      __syncthreads();
      for (int i = ti; i <= f; i += 16)
        for (int j = tj; j <= i && j < f; j += 16) {
          float s = A[i * lda + j];
#pragma unroll
          for (int t = 0; t < kCholTile; ++t) s = fmaf(ut[t * (f + 1) + i], yt[t * f + j], s);
          A[i * lda + j] = s;
        }
      __syncthreads();
    }

    // right-looking Cholesky of the first f columns of the augmented triangle
    bool ok = true;
    for (int k = 0; k < f; ++k) {
      float d = A[k * lda + k];
      if (!(d > 0.f)) {  // uniform: every thread reads the same LDS word
        ok = false;
        break;
      }
      float inv = 1.0f / sqrtf(d);
      __syncthreads();  // everyone has read the pivot
      for (int i = k + tid; i <= f; i += 256) A[i * lda + k] = i == k ? sqrtf(d) : A[i * lda + k] * inv;
      __syncthreads();
      for (int i = k + 1 + ti; i <= f; i += 16) {
        float lik = A[i * lda + k];
        for (int j = k + 1 + tj; j <= i && j < f; j += 16) A[i * lda + j] = fmaf(-lik, A[j * lda + k], A[i * lda + j]);
      }
      __syncthreads();
    }
    if (!ok) {
      if (tid == 0) atomicMin(failed_row, (unsigned long long)u);
      __syncthreads();
      continue;
    }
    // back substitution L^T x = z with one wavefront; lane l owns z[l + 64 m]
    if (tid < 64) {
      constexpr int MAXV = 4;  // f <= 256
      float z[MAXV];
#pragma unroll
      for (int m = 0; m < MAXV; ++m) {
        int i = tid + 64 * m;
        z[m] = i < f ? A[f * lda + i] : 0.f;
      }
      for (int k = f - 1; k >= 0; --k) {
        float zk = 0.f;
#pragma unroll
        for (int m = 0; m < MAXV; ++m)
          if ((k >> 6) == m) zk = bcast_lane(z[m], k & 63);
        float xk = zk / A[k * lda + k];
#pragma unroll
        for (int m = 0; m < MAXV; ++m) {
          int i = tid + 64 * m;
          if (i < k) z[m] = fmaf(-A[k * lda + i], xk, z[m]);
          if (i == k) z[m] = xk;
        }
      }
#pragma unroll
      for (int m = 0; m < MAXV; ++m) {
        int i = tid + 64 * m;
        if (i < f) X[(size_t)u * f + i] = z[m];
      }
    }
    __syncthreads();
  }
}


// ---- f <= 64: one WAVEFRONT per row, the whole system in registers ---------------------------------------------
// Lane i owns row i of A (64 VGPRs) and b[i].  A-build: for every nonzero the gathered factor row arrives as one
// coalesced wave load (lane j holds y[j]); y[j] is broadcast with v_readlane and lane i accumulates
// A[i][j] += ((|c|-1) y[i]) * y[j] for all j.  Right-looking Cholesky with the k loop fully unrolled (static register
// indices): pivot and column entries travel by v_readlane, no LDS and no barrier; the forward substitution rides
// along (b is treated as an extra column); the back substitution does one wave reduction per unknown.
// ~100 VALU instructions per nonzero + ~5.5K per row instead of 3 block barriers per column.
template <int FMAX>
__global__ __launch_bounds__(256) void als_cholesky_wave_kernel(const int32_t *__restrict__ order, int first, int count,
                                                                const int32_t *__restrict__ indptr,
                                                                const int32_t *__restrict__ indices,
                                                                const float *__restrict__ data, float *__restrict__ X,
                                                                const float *__restrict__ Y, const float *__restrict__ YtY,
                                                                int f, float reg, unsigned long long *failed_row) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const bool row_ok = lane < f;
  for (int ri = wave; ri < count; ri += nwaves) {
    const int u = __builtin_amdgcn_readfirstlane(order[first + ri]);
    const int row_begin = __builtin_amdgcn_readfirstlane(indptr[u]);
    const int row_end = __builtin_amdgcn_readfirstlane(indptr[u + 1]);
    float A[FMAX];
    float b = 0.f;
    // A = YtY + reg I   (row `lane`; rows/cols >= f are the identity so the unrolled factorisation stays finite)
#pragma unroll
    for (int j = 0; j < FMAX; ++j) {
      float a = (row_ok && j < f) ? YtY[(size_t)lane * f + j] : 0.f;
      A[j] = a + ((j == lane) ? (j < f ? reg : 1.f) : 0.f);
    }
    for (int k0 = row_begin; k0 < row_end; k0 += 64) {
      const int cnt = min(64, row_end - k0);
      const int my_idx = indices[k0 + min(lane, cnt - 1)];
      const float my_c = lane < cnt ? data[k0 + lane] : 0.f;
      for (int t0 = 0; t0 < cnt; t0 += 8) {
        // 8 gathers in flight per trip (entries past cnt repeat the last valid row with confidence 1 -> weight 0)
        float y8[8], c8[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int t = min(t0 + q, cnt - 1);
          const unsigned col = (unsigned)__builtin_amdgcn_readlane(my_idx, t);
          c8[q] = t0 + q < cnt ? bcast_lane(my_c, t) : 1.f;
          y8[q] = row_ok ? Y[(size_t)col * f + lane] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float c = c8[q], y = y8[q];
          const float a = c > 0.f ? c : -c;
          if (t0 + q < cnt && c > 0.f) b = fmaf(c, y, b);  // wave-uniform branch
          const float wy = (a - 1.f) * y;
#pragma unroll
          for (int j = 0; j < FMAX; ++j) A[j] = fmaf(wy, bcast_lane(y, j), A[j]);
        }
      }
    }
    // Cholesky A = L L^T in place (lane i keeps L[i][0..i]); z = L^-1 b in `b`
    bool ok = true;
#pragma unroll
    for (int k = 0; k < FMAX; ++k) {
      const float d = bcast_lane(A[k], k);  // pivot
      if (!(d > 0.f)) ok = false;
      const float sd = sqrtf(d);
      const float lik = lane == k ? sd : A[k] / sd;  // L[i][k] for i >= k (rows above k hold garbage, unused);
      A[k] = lik;                                    // true divisions: reg = 0 systems are badly conditioned
      const float zk = bcast_lane(b, k) / sd;        // z_k = b_k / L_kk
      b = lane == k ? zk : fmaf(-lik, zk, b);
#pragma unroll
      for (int j = k + 1; j < FMAX; ++j) A[j] = fmaf(-lik, bcast_lane(lik, j), A[j]);
    }
    if (!ok) {
      if (lane == 0) atomicMin(failed_row, (unsigned long long)u);
      continue;
    }
    // back substitution L^T x = z:  x_k = (z_k - sum_{i>k} L[i][k] x_i) / L[k][k]; lane i ends with x_i in `b`
#pragma unroll
    for (int k = FMAX - 1; k >= 0; --k) {
      float t = lane > k ? A[k] * b : 0.f;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
      const float xk = (bcast_lane(b, k) - t) / bcast_lane(A[k], k);
      if (lane == k) b = xk;
    }
    if (row_ok) X[(size_t)u * f + lane] = b;
  }
}

// ---- f <= 64, f even: MFMA A-build + register Cholesky, one wavefront per row (ALL row lengths) ---------------------
// A_u = YtY + reg I + sum_k (|c_k|-1) y_k y_k^T is a SYRK: with v_mfma_f32_32x32x2_f32 two nonzeros are one k-step.  Lane
// (r = l & 31, h = l >> 5) loads ONE float2 = factors (2r, 2r+1) of nonzero 2s + h (32 lanes cover the whole 64-factor
// row), so the factor set splits into EVEN and ODD factors and three 32x32 accumulator tiles cover the symmetric matrix:
//   T_ee += (w y_e) y_e^T,  T_oe += (w y_o) y_e^T,  T_oo += (w y_o) y_o^T        (T_eo = T_oe^T)
// i.e. 96 matrix-pipe cycles per nonzero instead of ~250 VALU cycles, off the VALU.  The tiles (+ YtY + reg I) go through
// a wave-private LDS image [64][65] to reach the row-per-lane layout of the register factorisation (als_cholesky_wave_kernel).
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void als_cholesky_mfma_kernel(const int32_t *__restrict__ order, int first, int count,
                                                                const int32_t *__restrict__ indptr,
                                                                const int32_t *__restrict__ indices,
                                                                const float *__restrict__ data, float *__restrict__ X,
                                                                const float *__restrict__ Y, const float *__restrict__ YtY,
                                                                int f, float reg, unsigned long long *failed_row) {
  constexpr int FMAX = 64, LDA = 65, KS = 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const int wslot = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float *As = smem + (size_t)wslot * FMAX * LDA;  // wave-private
  const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const int r = lane & 31, h = lane >> 5;
  const bool pair_ok = 2 * r < f;  // f even: both factors of the pair exist or neither
  const bool row_ok = lane < f;

  for (int ri = wave; ri < count; ri += nwaves) {
    const int u = __builtin_amdgcn_readfirstlane(order[first + ri]);
    const int row_begin = __builtin_amdgcn_readfirstlane(indptr[u]);
    const int row_end = __builtin_amdgcn_readfirstlane(indptr[u + 1]);
    f32x16 Tee, Toe, Too;
#pragma unroll
    for (int e = 0; e < 16; ++e) Tee[e] = Toe[e] = Too[e] = 0.f;
    float be = 0.f, bo = 0.f;  // b partials of this lane's half: factors 2r and 2r+1

    for (int k0 = row_begin; k0 < row_end; k0 += 64) {
      const int cnt = min(64, row_end - k0);
      const int my_idx = indices[k0 + min(lane, cnt - 1)];
      const float my_c = lane < cnt ? data[k0 + lane] : 1.f;  // confidence 1 -> weight 0, c+ masked below
      for (int s0 = 0; s0 < cnt; s0 += 2 * KS) {  // KS k-steps (2 KS nonzeros) per trip: KS gathers in flight
        float2 y[KS];
        float w[KS], cp[KS];
#pragma unroll
        for (int q = 0; q < KS; ++q) {
          const int t = s0 + 2 * q + h;  // this half's nonzero of k-step q
          const int tc = min(t, cnt - 1);
          const unsigned col = (unsigned)__shfl(my_idx, tc, 64);
          const float c = __shfl(my_c, tc, 64);
          const bool ok = t < cnt;
          w[q] = ok ? fabsf(c) - 1.f : 0.f;
          cp[q] = (ok && c > 0.f) ? c : 0.f;
          y[q] = pair_ok ? *reinterpret_cast<const float2 *>(Y + (size_t)col * f + 2 * r) : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int q = 0; q < KS; ++q) {
          const float ae = w[q] * y[q].x, ao = w[q] * y[q].y;
          Tee = __builtin_amdgcn_mfma_f32_32x32x2f32(ae, y[q].x, Tee, 0, 0, 0);
          Toe = __builtin_amdgcn_mfma_f32_32x32x2f32(ao, y[q].x, Toe, 0, 0, 0);
          Too = __builtin_amdgcn_mfma_f32_32x32x2f32(ao, y[q].y, Too, 0, 0, 0);
          be = fmaf(cp[q], y[q].x, be);
          bo = fmaf(cp[q], y[q].y, bo);
        }
      }
    }
    // tiles -> wave-private LDS image of the full symmetric A (C/D layout: col j' = lane & 31, row i' = (e&3)+8(e>>2)+4h)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int ip = (e & 3) + 8 * (e >> 2) + 4 * h, jp = r;
      As[(2 * ip) * LDA + 2 * jp] = Tee[e];          // A[2i'][2j']      (T_ee is itself symmetric: every (i',j') written)
      As[(2 * ip + 1) * LDA + 2 * jp + 1] = Too[e];  // A[2i'+1][2j'+1]
      As[(2 * ip + 1) * LDA + 2 * jp] = Toe[e];      // A[2i'+1][2j']
      As[(2 * jp) * LDA + 2 * ip + 1] = Toe[e];      // and its mirror A[2j'][2i'+1]
    }
    // b: sum the two halves, then lane i needs b[i]: factor 2r (even) / 2r+1 (odd) live in lanes r and r+32
    be += __shfl_xor(be, 32, 64);
    bo += __shfl_xor(bo, 32, 64);
    const float b_even_src = __shfl(be, lane >> 1, 64), b_odd_src = __shfl(bo, lane >> 1, 64);
    float b = (lane & 1) ? b_odd_src : b_even_src;  // b[lane]
    if (!row_ok) b = 0.f;
    // row-per-lane registers: A[lane][j] + YtY + reg I (identity padding beyond f keeps the unrolled factorisation finite)
    float A[FMAX];
#pragma unroll
    for (int j = 0; j < FMAX; ++j) {
      const float g0 = (row_ok && j < f) ? YtY[(size_t)lane * f + j] + As[lane * LDA + j] : 0.f;
      A[j] = g0 + ((j == lane) ? (j < f ? reg : 1.f) : 0.f);
    }
    bool ok = true;
#pragma unroll
    for (int k = 0; k < FMAX; ++k) {
      const float d = bcast_lane(A[k], k);  // pivot
      if (!(d > 0.f)) ok = false;
      const float sd = sqrtf(d);
      const float lik = lane == k ? sd : A[k] / sd;  // L[i][k] for i >= k (rows above k hold garbage, unused)
      A[k] = lik;
      const float zk = bcast_lane(b, k) / sd;  // z_k = b_k / L_kk
      b = lane == k ? zk : fmaf(-lik, zk, b);
#pragma unroll
      for (int j = k + 1; j < FMAX; ++j) A[j] = fmaf(-lik, bcast_lane(lik, j), A[j]);
    }
    if (!ok) {
      if (lane == 0) atomicMin(failed_row, (unsigned long long)u);
      continue;
    }
    // back substitution L^T x = z, column oriented: after x_k is known every lane i < k does z_i -= L[k][i] x_k.  L[k][i] is
    // row k of L (lane k's registers), so L goes through the wave-private LDS image once to get its transpose per lane:
    // one FMA per unknown instead of one wave reduction per unknown.
#pragma unroll
    for (int j = 0; j < FMAX; ++j) As[lane * LDA + j] = A[j];
#pragma unroll
    for (int k = 0; k < FMAX; ++k) A[k] = As[k * LDA + lane];  // A[k] = L[k][lane] (valid for k >= lane)
#pragma unroll
    for (int k = FMAX - 1; k >= 0; --k) {
      const float xk = bcast_lane(b, k) / bcast_lane(A[k], k);  // lane k holds z_k (fully updated) and L[k][k]
      b = lane == k ? xk : (lane < k ? fmaf(-A[k], xk, b) : b);
    }
    if (row_ok) X[(size_t)u * f + lane] = b;
  }
}

void zero_rows(const int32_t *order, int first, int count, float *X, int f);  // als_cg.hip

static unsigned long long *g_failed = nullptr;

// returns -1, or the smallest failing row
int64_t least_squares_cholesky(const imp_csr *C, imp_matrix *X, const imp_matrix *YtY, const imp_matrix *Y, double reg) {
  const int f = (int)X->cols;
  if (f > 160) throw std::invalid_argument("least_squares_cholesky: factors must be <= 160 in this build");
  int lda = (f + 1) | 1;  // odd
  size_t lds = ((size_t)(f + 1) * lda + (size_t)kCholTile * f + (size_t)kCholTile * (f + 1)) * sizeof(float);
  if (!g_failed) IMP_CHECK_HIP(hipMalloc(&g_failed, sizeof(unsigned long long)));
  IMP_CHECK_HIP(hipMemsetAsync(g_failed, 0xFF, sizeof(unsigned long long), stream()));
  int nonempty = C->nonempty();
  static const bool no_wave = getenv("IMP_CHOL_NO_WAVE") != nullptr;
  static const bool no_mfma = getenv("IMP_CHOL_NO_MFMA") != nullptr;
  if (nonempty > 0 && f <= 64 && f % 2 == 0 && !no_mfma && !no_wave) {
    // every non-empty row: MFMA A-build + register Cholesky, one wavefront per row
    size_t lds_m = (size_t)4 * 64 * 65 * sizeof(float);
    IMP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(als_cholesky_mfma_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m));
    int grid = std::min((nonempty + 3) / 4, ctx().num_cus * 2);
    {
      IMP_PROF("als_cholesky_mfma_rows");
      als_cholesky_mfma_kernel<<<grid, 256, lds_m, stream()>>>(C->order.data(), 0, nonempty, C->indptr.data(), C->indices.data(),
                                                               C->data.data(), X->f32(), Y->f32(), YtY->f32(), f, (float)reg,
                                                               g_failed);
      IMP_CHECK_HIP(hipGetLastError());
    }
    zero_rows(C->order.data(), C->first_empty(), C->n_empty(), X->f32(), f);
    unsigned long long failed_m = 0;
    IMP_CHECK_HIP(hipMemcpyAsync(&failed_m, g_failed, sizeof(failed_m), hipMemcpyDeviceToHost, stream()));
    sync();
    return failed_m == ~0ULL ? -1 : (int64_t)failed_m;
  }
  // f <= 64 (odd f): rows up to 256 nnz go to the register-resident wave kernel; longer rows (and any larger f) to the
  // workgroup kernel, whose 256 threads share the A-build of one row
  const int n_block = (f <= 64 && !no_wave) ? C->bin_start[2] : nonempty;
  const int n_wave = nonempty - n_block;
  if (n_wave > 0) {
    int grid = std::min((n_wave + 3) / 4, ctx().num_cus * 8);
    IMP_PROF("als_cholesky_wave_rows");
    if (f <= 32)
      als_cholesky_wave_kernel<32><<<grid, 256, 0, stream()>>>(C->order.data(), n_block, n_wave, C->indptr.data(),
                                                               C->indices.data(), C->data.data(), X->f32(), Y->f32(),
                                                               YtY->f32(), f, (float)reg, g_failed);
    else
      als_cholesky_wave_kernel<64><<<grid, 256, 0, stream()>>>(C->order.data(), n_block, n_wave, C->indptr.data(),
                                                               C->indices.data(), C->data.data(), X->f32(), Y->f32(),
                                                               YtY->f32(), f, (float)reg, g_failed);
    IMP_CHECK_HIP(hipGetLastError());
  }
  if (n_block > 0) {
    IMP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(als_cholesky_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / lds));
    int grid = std::min(n_block, ctx().num_cus * per_cu);
    IMP_PROF("als_cholesky_rows");
    als_cholesky_kernel<<<grid, 256, lds, stream()>>>(C->order.data(), 0, n_block, C->indptr.data(), C->indices.data(),
                                                      C->data.data(), X->f32(), Y->f32(), YtY->f32(), f, (float)reg, lda,
                                                      g_failed);
    IMP_CHECK_HIP(hipGetLastError());
  }
  zero_rows(C->order.data(), C->first_empty(), C->n_empty(), X->f32(), f);
  unsigned long long failed = 0;
  IMP_CHECK_HIP(hipMemcpyAsync(&failed, g_failed, sizeof(failed), hipMemcpyDeviceToHost, stream()));
  sync();
  return failed == ~0ULL ? -1 : (int64_t)failed;
}

}  // namespace imp
