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
#include <type_traits>

#include "common.h"

namespace imp {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kCholTile = 8;

__device__ __forceinline__ float bcast_lane(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// ---- round 4: the same workgroup-per-row scheme with REGISTER BLOCKS and PANELS -------------------------------------------------
// The kernel above meets a workgroup barrier three times per column and pays 16 LDS reads for 8 FMAs in its A-build: 546 ms per
// configs[2]-shaped iteration at f = 128, 0.02 of the fp32 peak -- 400 barrier phases of ~2 K cycles per row.  Here:
//   * A-build: a thread owns up to four 4 x 4 blocks of the lower triangle and keeps them in registers across ALL tiles of the
//     row (two 16-byte LDS reads per 16 FMAs), the triangle is written once;
//   * factorisation in panels of 8 columns: wavefront 0 alone factors the panel (no barrier inside: a wave's LDS operations
//     execute in order; lane l owns rows k0 + l, k0 + l + 64, ...) and leaves a dense, aligned copy L[., k0 .. k0+7] in a side
//     buffer; all four waves then apply the rank-8 update to the trailing triangle block by block from that copy (8 LDS reads
//     of 16 bytes per 128 FMAs) -- two barriers per PANEL;
//   * the augmented row f (b^T -> z^T) rides along as before, the back substitution is unchanged.
// Arithmetic: the same products, panel by panel instead of column by column (association of the trailing sums differs).
// f > 256 (round 6; the CPU reference takes any f, _als.pyx:75-142): GLOBAL = true keeps the augmented triangle in a per-workgroup
// slice of a device workspace instead of the LDS (square layout; L2 / infinity-cache resident: 0.4 MB at f = 320, 4.2 MB at
// f = 1024) -- every phase that passes data between threads through it is already separated by a workgroup barrier, which orders
// global memory inside a workgroup as well.  ROWS = rows of the panel a lane holds (64 ROWS >= f + 1), MAXV = f / 64 rounded up.
// A coverage path: correct, an order of magnitude off the LDS kernels' rate per flop.
constexpr int kCholPanel = 8;
template <bool PACKED, bool GLOBAL = false, int ROWS = 5, int MAXV_ = 4>
__global__ __launch_bounds__(256) void als_cholesky_blocked_kernel(const int32_t *__restrict__ order, int first, int count,
                                                                   const int32_t *__restrict__ indptr,
                                                                   const int32_t *__restrict__ indices,
                                                                   const float *__restrict__ data, float *__restrict__ X,
                                                                   const float *__restrict__ Y, const float *__restrict__ YtY,
                                                                   int f, float reg, int lda, unsigned long long *failed_row,
                                                                   int ko,  // ko: timing-only knock-out mask (IMP_CHOL_KO), 0 in production
                                                                   const unsigned *__restrict__ dev_count = nullptr,  // rows of `order` to take, if fewer than `count`
                                                                   float *__restrict__ global_ws = nullptr) {
  static_assert(!(PACKED && GLOBAL), "the workspace form uses the square layout");
  if (dev_count) count = min(count, (int)*dev_count);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int m = f + 1;                         // rows of the augmented triangle (row f = b^T -> z^T), f columns
  const int nbr = (m + 3) >> 2, nbc = (f + 3) >> 2;  // 4 x 4 blocks
  const int FS = 4 * nbc, US = 4 * nbr;        // padded strides of the staged tile
  const size_t a_words = PACKED ? (size_t)m * (m + 1) / 2 : (size_t)m * lda;
  float *A = GLOBAL ? global_ws + (size_t)blockIdx.x * ((a_words + 3) & ~(size_t)3) : smem;
  auto at = [&](int i, int j) { return PACKED ? i * (i + 1) / 2 + j : i * lda + j; };  // j <= i
  float *yt = GLOBAL ? smem : A + ((a_words + 3) & ~(size_t)3);  // [TILE][FS]  gathered rows, zero beyond f
  float *ut = yt + (size_t)kCholTile * FS;       // [TILE][US]  (|c|-1) y, c+ at index f, zero beyond
  float *panel = ut + (size_t)kCholTile * US;    // [m][8]      L[i][k0 .. k0+7] of the current panel
  int *flag = reinterpret_cast<int *>(panel + (size_t)m * kCholPanel);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // this thread's blocks of the lower triangle: block-row bi has min(bi + 1, nbc) blocks, numbered row after row
  const int n_blocks = nbc * (nbc + 1) / 2 + (nbr > nbc ? nbc : 0);
  auto block_of = [&](int blk, int &bi, int &bj) {
    if (blk >= nbc * (nbc + 1) / 2) {
      bi = nbc, bj = blk - nbc * (nbc + 1) / 2;
    } else {
      bi = (int)((sqrtf(8.f * (float)blk + 1.f) - 1.f) * 0.5f);
      while (bi * (bi + 1) / 2 > blk) --bi;
      while ((bi + 1) * (bi + 2) / 2 <= blk) ++bi;
      bj = blk - bi * (bi + 1) / 2;
    }
  };
  constexpr int MAXB = 4;  // blocks held at a time: n_blocks <= 1024 covers f <= 176 in one round, f = 256 takes three

  for (int ri = blockIdx.x; ri < count; ri += gridDim.x) {
    const int u = order[first + ri];
    const int row_begin = indptr[u], row_end = indptr[u + 1];
    if (tid == 0) *flag = 0;
    // ---- A = YtY + reg I + sum (|c|-1) y y^T, b = sum c+ y: register blocks, rounds of MAXB blocks per thread -----------------
    for (int round0 = 0; round0 < n_blocks; round0 += 256 * MAXB) {
      float acc[MAXB][4][4];
      int bi[MAXB], bj[MAXB];
#pragma unroll
      for (int b = 0; b < MAXB; ++b) {
        const int blk = round0 + tid + 256 * b;
        bi[b] = bj[b] = -1;
        if (blk < n_blocks) block_of(blk, bi[b], bj[b]);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int i = 4 * bi[b] + r, j = 4 * bj[b] + c;
            acc[b][r][c] = (bi[b] >= 0 && i < f && j < f && j <= i) ? YtY[(size_t)i * f + j] + (i == j ? reg : 0.f) : 0.f;
          }
      }
      for (int k0 = row_begin; k0 < row_end; k0 += kCholTile) {
        const int cnt = min(kCholTile, row_end - k0);
        __syncthreads();  // the previous tile has been consumed
        for (int e = tid; e < kCholTile * US; e += 256) {  // US >= FS: one sweep fills both
          const int t = e / US, c = e - t * US;
          float yv = 0.f, uv = 0.f;
          if (t < cnt) {
            const float conf = data[k0 + t];
            if (c < f) {
              yv = Y[(size_t)indices[k0 + t] * f + c];
              uv = (fabsf(conf) - 1.f) * yv;
            } else if (c == f) {
              uv = conf > 0.f ? conf : 0.f;
            }
          }
          if (c < FS) yt[t * FS + c] = yv;
          ut[t * US + c] = uv;
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < MAXB; ++b) {
          if (bi[b] < 0 || (ko & 1)) continue;
#pragma unroll
          for (int t = 0; t < kCholTile; ++t) {
            const float4 u4 = *reinterpret_cast<const float4 *>(ut + t * US + 4 * bi[b]);
            const float4 y4 = *reinterpret_cast<const float4 *>(yt + t * FS + 4 * bj[b]);
            const float uu[4] = {u4.x, u4.y, u4.z, u4.w}, yy[4] = {y4.x, y4.y, y4.z, y4.w};
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
              for (int c = 0; c < 4; ++c) acc[b][r][c] = fmaf(uu[r], yy[c], acc[b][r][c]);
          }
        }
      }
#pragma unroll
      for (int b = 0; b < MAXB; ++b) {
        if (bi[b] < 0) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int i = 4 * bi[b] + r, j = 4 * bj[b] + c;
            if (i < m && j < f && j <= i) A[at(i, j)] = acc[b][r][c];
          }
      }
    }
    __syncthreads();

    // ---- right-looking Cholesky of the first f columns, 8 columns per panel ---------------------------------------------------
    for (int k0 = 0; k0 < f; k0 += kCholPanel) {
      const int nb = min(kCholPanel, f - k0);
      if (wave == 0 && !(ko & 2)) {
        // The panel -- columns k0 .. k0 + nb - 1 over rows k0 .. f -- in REGISTERS of one wavefront: lane l holds rows
        // k0 + l + 64 r.  Row k0 + c is lane c's first row, so the pivot and the sub-diagonal entries a column step needs travel
        // by v_readlane; nothing touches the LDS between the load and the store of the panel (a first version walked the
        // panel through the LDS column by column: ~70 dependent LDS round trips per column, 640 K cycles per row).
        constexpr int R = ROWS;  // 64 R >= f + 1 rows (5: f <= 256)
        float pr[R][kCholPanel];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int i = k0 + lane + 64 * r;
#pragma unroll
          for (int c = 0; c < kCholPanel; ++c) pr[r][c] = (i < m && c < nb && k0 + c <= i) ? A[at(i, k0 + c)] : 0.f;
        }
        bool bad = false;
#pragma unroll
        for (int c = 0; c < kCholPanel; ++c) {
          if (c < nb && !bad) {  // wave-uniform
            const float d = bcast_lane(pr[0][c], c);  // the pivot: row k0 + c lives in lane c
            if (!(d > 0.f)) {
              bad = true;
            } else {
              const float root = sqrtf(d), inv = 1.0f / root;
#pragma unroll
              for (int r = 0; r < R; ++r) pr[r][c] = (r == 0 && lane == c) ? root : pr[r][c] * inv;  // rows above the pivot: unused
#pragma unroll
              for (int c2 = c + 1; c2 < kCholPanel; ++c2) {
                const float l2 = bcast_lane(pr[0][c], c2);  // L[k0 + c2][k0 + c]
#pragma unroll
                for (int r = 0; r < R; ++r) pr[r][c2] = fmaf(-pr[r][c], l2, pr[r][c2]);
              }
            }
          }
        }
        if (bad && lane == 0) *flag = 1;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int i = k0 + lane + 64 * r;
          if (i < m) {
#pragma unroll
            for (int c = 0; c < kCholPanel; ++c) {
              if (c < nb && k0 + c <= i) A[at(i, k0 + c)] = pr[r][c];
              panel[i * kCholPanel + c] = pr[r][c];  // dense copy for the trailing update (rows below the panel only are read)
            }
          }
        }
      }
      __syncthreads();
      if (*flag) break;  // uniform
      // trailing update: A[i][j] -= sum_c L[i][k0+c] L[j][k0+c] for j >= k0 + nb (a multiple of 4 unless this is the last panel)
      const int jb0 = (k0 + nb) >> 2;
      if (nb == kCholPanel && jb0 < nbr && !(ko & 4)) {
        // blocks (bi, bj) with jb0 <= bj <= bi: numbered row after row from block-row jb0
        const int rows_b = nbr - jb0;
        for (int t = tid;; t += 256) {
          // decode t -> (bi, bj) inside the trailing triangle of block-rows jb0 .. nbr-1 with min(bi - jb0 + 1, nbc - jb0) blocks each
          int bi2, bj2;
          {
            const int w = nbc - jb0;  // full width of the trailing block triangle (block-columns jb0 .. nbc-1)
            const int tri = w * (w + 1) / 2;
            if (t >= tri + (rows_b > w ? w : 0) || w <= 0) break;
            if (t >= tri) {
              bi2 = nbc, bj2 = jb0 + (t - tri);
            } else {
              int q = (int)((sqrtf(8.f * (float)t + 1.f) - 1.f) * 0.5f);
              while (q * (q + 1) / 2 > t) --q;
              while ((q + 1) * (q + 2) / 2 <= t) ++q;
              bi2 = jb0 + q, bj2 = jb0 + (t - q * (q + 1) / 2);
            }
          }
          float li[4][kCholPanel], lj[4][kCholPanel];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = min(4 * bi2 + r, m - 1), j = min(4 * bj2 + r, m - 1);
            const float4 a0 = *reinterpret_cast<const float4 *>(panel + i * kCholPanel), a1 = *reinterpret_cast<const float4 *>(panel + i * kCholPanel + 4);
            const float4 b0 = *reinterpret_cast<const float4 *>(panel + j * kCholPanel), b1 = *reinterpret_cast<const float4 *>(panel + j * kCholPanel + 4);
            li[r][0] = a0.x, li[r][1] = a0.y, li[r][2] = a0.z, li[r][3] = a0.w, li[r][4] = a1.x, li[r][5] = a1.y, li[r][6] = a1.z, li[r][7] = a1.w;
            lj[r][0] = b0.x, lj[r][1] = b0.y, lj[r][2] = b0.z, lj[r][3] = b0.w, lj[r][4] = b1.x, lj[r][5] = b1.y, lj[r][6] = b1.z, lj[r][7] = b1.w;
          }
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const int i = 4 * bi2 + r, j = 4 * bj2 + c;
              if (i < m && j < f && j <= i) {
                float s = A[at(i, j)];
#pragma unroll
                for (int k = 0; k < kCholPanel; ++k) s = fmaf(-li[r][k], lj[c][k], s);
                A[at(i, j)] = s;
              }
            }
        }
      }
      __syncthreads();
    }
    if (*flag) {
      if (tid == 0) atomicMin(failed_row, (unsigned long long)u);
      __syncthreads();
      continue;
    }
    // back substitution L^T x = z with one wavefront; lane l owns z[l + 64 m]
    if (tid < 64 && !(ko & 8)) {
      constexpr int MAXV = MAXV_;  // 64 MAXV >= f (4: f <= 256)
      float z[MAXV];
#pragma unroll
      for (int mm = 0; mm < MAXV; ++mm) {
        int i = tid + 64 * mm;
        z[mm] = i < f ? A[at(f, i)] : 0.f;
      }
      for (int k = f - 1; k >= 0; --k) {
        float zk = 0.f;
#pragma unroll
        for (int mm = 0; mm < MAXV; ++mm)
          if ((k >> 6) == mm) zk = bcast_lane(z[mm], k & 63);
        float xk = zk / A[at(k, k)];
#pragma unroll
        for (int mm = 0; mm < MAXV; ++mm) {
          int i = tid + 64 * mm;
          if (i < k) z[mm] = fmaf(-A[at(k, i)], xk, z[mm]);
          if (i == k) z[mm] = xk;
        }
      }
#pragma unroll
      for (int mm = 0; mm < MAXV; ++mm) {
        int i = tid + 64 * mm;
        if (i < f) X[(size_t)u * f + i] = z[mm];
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

// ---- f = 64 (BASELINE configs[1]): MFMA A-build + left-looking Cholesky, one wavefront per row (ALL row lengths) --------
// A_u = YtY + reg I + sum_k (|c_k|-1) y_k y_k^T is a SYRK: with v_mfma_f32_32x32x2_f32 two nonzeros are one k-step.  Lane
// (r = l & 31, h = l >> 5) loads ONE float2 = factors (2r, 2r+1) of nonzero 2s + h (32 lanes cover the whole 64-factor
// row), so the factor set splits into EVEN and ODD factors and three 32x32 accumulator tiles cover the symmetric matrix:
//   T_ee += (w y_e) y_e^T,  T_oe += (w y_o) y_e^T,  T_oo += (w y_o) y_o^T        (T_eo = T_oe^T)
// i.e. 96 matrix-pipe cycles per nonzero instead of ~250 VALU cycles, off the VALU.  The gathers of trip t + 1 are in
// flight during the MFMAs of trip t, the (column, confidence) pairs of the next 64 nonzeros during the current 64.
//
// The tiles go through a wave-private LDS image to the row-per-lane layout (lane i owns row i).  Only the LOWER triangle is
// kept, row i padded to a multiple of 4 floats at offset 4 (a+1)(2a+b), i = 4a+b (8.7 KB per wave instead of 17 KB), and
// G = YtY + reg I is staged once per workgroup: 52 KB of LDS per workgroup = 3 workgroups (12 waves) per CU.
//
// Factorisation: LEFT-looking, column by column.  Lane i keeps row i of L in registers; at step k every lane forms
//   s_i = A[i][k] - sum_{j<k} L[i][j] L[k][j]
// with its own registers and row k of L read from the image as BROADCAST ds_read_b128 (all lanes, same address; issued
// back to back, then consumed), 4 independent accumulators; the pivot travels by one v_readlane, L[i][k] = s_i / sqrt(s_k)
// is stored to the image, and the forward substitution rides along (one FMA per step).
//
// History (configs[1], ms per iteration): round 1 was right-looking with the column broadcast lane by lane (v_readlane +
// FMA per element: 12.7 K straight-line instructions per row, 76 KB of code) and read YtY row-wise: 46-57 ms.  Per-phase
// cycle counters (IMP_CHOL_STATS=1) then showed where a 50-nonzero row's 120 K cycles went: 77 K in the conversion phase
// (the row-invariant YtY loads were hoisted out of the row loop, spilled, and came back from scratch memory one dependent
// round trip at a time; the 128 lane masks likewise, through v_writelane / v_readlane), 24 K factorisation, 14 K SYRK.
typedef float f32x16 __attribute__((ext_vector_type(16)));

// compile-time loop: body(std::integral_constant<int, i>{}) for i = 0 .. N-1.  The factorisation indexes 64 registers by
// the column number: `#pragma unroll` gave up on the nested loops once and put the row into scratch memory (353 K cycles).
template <int N, int I = 0, typename Body> __device__ __forceinline__ void static_for(Body &&body) {
  if constexpr (I < N) {
    body(std::integral_constant<int, I>{});
    static_for<N, I + 1>(body);
  }
}

__host__ __device__ constexpr int chol_rowoff(int i) { return 4 * ((i >> 2) + 1) * (2 * (i >> 2) + (i & 3)); }
constexpr int kCholTri = chol_rowoff(63) + 64;  // 2176 floats

// A-build over the nonzeros [row_begin, row_end) of one row: T_ee / T_oe / T_oo += Y^T diag(|c| - 1) Y (even / odd factor
// split, see the kernel), be / bo += sum c+ y.  Shared by the row kernel and by the segment kernel of the long rows.
__device__ __forceinline__ void chol_syrk_range(const int32_t *__restrict__ indices, const float *__restrict__ data,
                                                const float *__restrict__ Y, int row_begin, int row_end, int lane, f32x16 &Tee,
                                                f32x16 &Toe, f32x16 &Too, float &be, float &bo) {
  constexpr int F = 64, KS = 4;
  const int r = lane & 31, h = lane >> 5;
  // one trip = KS k-steps = 2 KS nonzeros; lane (r, h) handles nonzero 2 q + h of every k-step q
  float2 y[2][KS];
  float w[2][KS], cp[2][KS];
  auto fetch = [&](int buf, int my_idx, float my_c, int s0, int cnt) {
#pragma unroll
    for (int q = 0; q < KS; ++q) {
      const int t = s0 + 2 * q + h;
      const int tc = min(t, cnt - 1);
      const unsigned col = (unsigned)__shfl(my_idx, tc, 64);
      const float c = __shfl(my_c, tc, 64);
      const bool ok = t < cnt;
      w[buf][q] = ok ? fabsf(c) - 1.f : 0.f;
      cp[buf][q] = (ok && c > 0.f) ? c : 0.f;
      y[buf][q] = *reinterpret_cast<const float2 *>(Y + (size_t)col * F + 2 * r);
    }
  };
  auto multiply = [&](int buf) {
#pragma unroll
    for (int q = 0; q < KS; ++q) {
      const float ae = w[buf][q] * y[buf][q].x, ao = w[buf][q] * y[buf][q].y;
      Tee = __builtin_amdgcn_mfma_f32_32x32x2f32(ae, y[buf][q].x, Tee, 0, 0, 0);
      Toe = __builtin_amdgcn_mfma_f32_32x32x2f32(ao, y[buf][q].x, Toe, 0, 0, 0);
      Too = __builtin_amdgcn_mfma_f32_32x32x2f32(ao, y[buf][q].y, Too, 0, 0, 0);
      be = fmaf(cp[buf][q], y[buf][q].x, be);
      bo = fmaf(cp[buf][q], y[buf][q].y, bo);
    }
  };
  if (row_begin < row_end) {
    int idx_next = indices[row_begin + min(lane, row_end - row_begin - 1)];
    float c_next = row_begin + lane < row_end ? data[row_begin + lane] : 1.f;  // confidence 1 -> weight 0, c+ masked
    for (int k0 = row_begin; k0 < row_end; k0 += 64) {
      const int cnt = min(64, row_end - k0);
      const int my_idx = idx_next;
      const float my_c = c_next;
      if (k0 + 64 < row_end) {  // entries of the next 64 nonzeros
        idx_next = indices[k0 + 64 + min(lane, row_end - k0 - 65)];
        c_next = k0 + 64 + lane < row_end ? data[k0 + 64 + lane] : 1.f;
      }
      fetch(0, my_idx, my_c, 0, cnt);
      for (int s0 = 0; s0 < cnt; s0 += 4 * KS) {  // two trips per round: the other buffer's gathers fly during the MFMAs
        if (s0 + 2 * KS < cnt) fetch(1, my_idx, my_c, s0 + 2 * KS, cnt);
        multiply(0);
        if (s0 + 2 * KS < cnt) {
          if (s0 + 4 * KS < cnt) fetch(0, my_idx, my_c, s0 + 4 * KS, cnt);
          multiply(1);
        }
      }
    }
  }
}

// Rows of more than kCholLongRow nonzeros: the A-build of a row is serial in its wavefront (96 matrix-pipe cycles per
// nonzero), and a popular item's row has tens of thousands of them -- the 79 K-nnz row of the configs[1] shape kept one
// wavefront busy for 12.5 ms of a 15.6 ms launch.  Their nonzeros are therefore cut into segments of kCholSegment
// (imp_csr::plan_chol), one wavefront per segment builds a partial (T_ee, T_oe, T_oo, b) here, and the row kernel sums a long
// row's partials in segment order instead of walking its nonzeros.   workspace: [segment][50][64 lanes]
constexpr int kCholPartial = 50 * 64;
__global__ __launch_bounds__(256) void als_cholesky_f64_partial_kernel(const LongPlanDev plan, const int32_t *__restrict__ indices,
                                                                       const float *__restrict__ data, const float *__restrict__ Y,
                                                                       float *__restrict__ ws) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int sg = wave; sg < plan.n_seg; sg += nwaves) {
    f32x16 Tee, Toe, Too;
#pragma unroll
    for (int e = 0; e < 16; ++e) Tee[e] = Toe[e] = Too[e] = 0.f;
    float be = 0.f, bo = 0.f;
    chol_syrk_range(indices, data, Y, plan.seg_begin[sg], plan.seg_end[sg], lane, Tee, Toe, Too, be, bo);
    float *out = ws + (size_t)sg * kCholPartial + lane;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      out[e * 64] = Tee[e];
      out[(16 + e) * 64] = Toe[e];
      out[(32 + e) * 64] = Too[e];
    }
    out[48 * 64] = be;
    out[49 * 64] = bo;
  }
}

template <bool STATS>
__global__ __launch_bounds__(256, 3) void als_cholesky_f64_kernel(const int32_t *__restrict__ order, int first, int count,
                                                                  const int32_t *__restrict__ indptr,
                                                                  const int32_t *__restrict__ indices,
                                                                  const float *__restrict__ data, float *__restrict__ X,
                                                                  const float *__restrict__ Y, const float *__restrict__ YtY,
                                                                  float reg, unsigned long long *failed_row,
                                                                  unsigned long long *stats, const LongPlanDev plan,
                                                                  const float *__restrict__ partials) {
  constexpr int F = 64, GLD = 68;
  // STATS (debug, IMP_CHOL_STATS=1): s_memtime ticks per phase summed over waves -- [1] SYRK loop (entries + gathers + MFMA)
  // [2] tiles -> LDS image -> row registers  [3] factorisation  [4] back substitution + store  [7] rows
  unsigned long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last = 0;
  auto tick = [&](int slot) {
    if constexpr (STATS) {
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      unsigned long long now = __builtin_amdgcn_s_memtime();
      if (slot >= 0) tk[slot] += now - t_last;
      t_last = now;
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *Gs = smem;  // [64][GLD]  YtY + reg I
  const int lane = threadIdx.x & 63;
  const int wslot = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float *As = smem + F * GLD + (size_t)wslot * kCholTri;  // wave-private lower-triangular image
  for (int e = threadIdx.x; e < F * F; e += 256) {
    const int rr = e >> 6, cc = e & 63;
    Gs[rr * GLD + cc] = YtY[e] + (rr == cc ? reg : 0.f);
  }
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const int my_off = chol_rowoff(lane);

  for (int ri = wave; ri < count; ri += nwaves) {
    const int u = __builtin_amdgcn_readfirstlane(order[first + ri]);
    const int row_begin = __builtin_amdgcn_readfirstlane(indptr[u]);
    const int row_end = __builtin_amdgcn_readfirstlane(indptr[u + 1]);
    // lane ids are kept opaque per row: every lane mask below (lane == k, lane >= k, ...) is row-invariant, and hoisted out of
    // the row loop they are more SGPR pairs than exist
    int lane_v = lane;
    asm volatile("" : "+v"(lane_v));
    f32x16 Tee, Toe, Too;
#pragma unroll
    for (int e = 0; e < 16; ++e) Tee[e] = Toe[e] = Too[e] = 0.f;
    float be = 0.f, bo = 0.f;  // b partials of this lane's half: factors 2r and 2r+1
    tick(-1);

    if (ri < plan.n_long) {  // long row (the first plan.n_long entries of the schedule): partials of its segments, in order
      for (int sg = plan.row_seg[ri]; sg < plan.row_seg[ri + 1]; ++sg) {
        const float *in = partials + (size_t)sg * kCholPartial + lane;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          Tee[e] += in[e * 64];
          Toe[e] += in[(16 + e) * 64];
          Too[e] += in[(32 + e) * 64];
        }
        be += in[48 * 64];
        bo += in[49 * 64];
      }
    } else {
      chol_syrk_range(indices, data, Y, row_begin, row_end, lane, Tee, Toe, Too, be, bo);
    }
    tick(1);
    // tiles -> lower triangle of the image (C/D layout: col j' = lane & 31, row i' = (e&3) + 8 (e>>2) + 4 h)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int ip = (e & 3) + 8 * (e >> 2) + 4 * (lane_v >> 5), jp = lane_v & 31;
      const int off_e = chol_rowoff(2 * ip), off_o = chol_rowoff(2 * ip + 1);
      if (ip >= jp) {
        As[off_e + 2 * jp] = Tee[e];      // A[2i'][2j']
        As[off_o + 2 * jp + 1] = Too[e];  // A[2i'+1][2j'+1]
        As[off_o + 2 * jp] = Toe[e];      // A[2i'+1][2j']
      } else {
        As[chol_rowoff(2 * jp) + 2 * ip + 1] = Toe[e];  // its mirror A[2j'][2i'+1] (T_oe covers odd rows x even columns)
      }
    }
    // b: sum the two halves, then lane i needs b[i]: factor 2r (even) / 2r+1 (odd) live in lanes r and r+32
    be += __shfl_xor(be, 32, 64);
    bo += __shfl_xor(bo, 32, 64);
    const float b_even_src = __shfl(be, lane >> 1, 64), b_odd_src = __shfl(bo, lane >> 1, 64);
    float b = (lane_v & 1) ? b_odd_src : b_even_src;  // b[lane]
    // row-per-lane registers: A[lane][j] = image + G for j <= lane (beyond: other rows' words, never used)
    // kept as register PAIRS: the dot products below run on v_pk_fma_f32 (two FMAs per issue slot)
    f32x2 A2[F / 2];
#define A(j) A2[(j) >> 1][(j) & 1]
#pragma unroll
    for (int j = 0; j < F; j += 4) {
      const float4 t = *reinterpret_cast<const float4 *>(As + my_off + j);
      const float4 g = *reinterpret_cast<const float4 *>(Gs + lane * GLD + j);
      A(j) = t.x + g.x, A(j + 1) = t.y + g.y, A(j + 2) = t.z + g.z, A(j + 3) = t.w + g.w;
    }
    tick(2);
    bool ok = true;
    float dinv = 0.f;  // lane k keeps 1 / L[k][k] for the back substitution
    // Row k of L (columns 0..k-1) is read from the image as broadcast ds_read_b128 (same address in every lane).  Software
    // pipeline: the chunks of row k+1 that do not contain column k are already final, so their reads are issued right
    // after step k's dot product and fly during its pivot chain (v_readlane -> v_rsq -> Newton -> column scale -> store);
    // only the chunk holding column k is read after the store.  lrow is free by then: no extra registers.
    // Round 4: four columns per step.  Column by column (round 2's form, removed in round 6) the row's critical path was 64
    // times [dot product -> pivot chain -> LDS store of the column -> LDS read of the next row's last chunk]: an LDS round trip
    // per column that nothing covers at three waves per SIMD (IMP_CHOL_STATS: 34.8 K of a 50-nonzero row's 60 K cycles).  A
    // block of columns k0 .. k0 + 3 needs rows k0 .. k0 + 3 of L only in columns < k0, all final when the block starts: their
    // chunks are streamed as broadcast ds_read_b128, four rows at a time into 16 accumulators (the same products, the same
    // four running sums per column as before); the 4 x 4 diagonal block is then factorised in registers -- the three
    // sub-diagonal entries a pivot step needs travel by v_readlane, not through the image -- and the four new columns go to
    // the image in ONE ds_write_b128 per lane.  16 LDS round trips per row instead of 64; the forward substitution rides
    // along as before.
    // Round 6: the bulk of those sums on the matrix cores.  At the start of every 16-column PANEL P >= 1 the contributions of
    // all earlier panels to its columns, T[i][c] = sum_{j < 16 P} L[i][j] L[16 P + c][j], come from v_mfma_f32_16x16x4_f32 --
    // operands read straight from the image (columns < 16 P of every row are final), one 16 x 16 tile per row block >= P, 40
    // MFMAs per row in all -- go through the image's not-yet-written positions (row, 16 P + c) to the row-per-lane layout and
    // are subtracted from the lane's A values; the four-column steps then only sum over the columns of their own panel.  It
    // replaces ~1150 of a row's ~6 K vector instructions (packed FMAs + broadcast LDS reads): the kernel is vector-issue bound.
#ifndef IMP_CHOL_NO_PANEL_MFMA
    constexpr bool kPanelMfma = true;
#else
    constexpr bool kPanelMfma = false;
#endif
    static_for<F / 4>([&](auto bc) {
      constexpr int k0 = 4 * decltype(bc)::value;
      constexpr int j_first = kPanelMfma ? 16 * (k0 / 16) : 0;  // first column the four-column step still sums over
      if constexpr (kPanelMfma && k0 % 16 == 0 && k0 > 0) {
        constexpr int P = k0 / 16;
        const int l16 = lane_v & 15, lq = lane_v >> 4;
        const int off_b = chol_rowoff(16 * P + l16) + lq;  // B[k = lq][c = l16] = L[16 P + l16][4 ks + lq]
        static_for<4 - P>([&](auto rbc) {
          constexpr int rb = P + decltype(rbc)::value;
          const int off_a = chol_rowoff(16 * rb + l16) + lq;  // A[i = l16][k = lq] = L[16 rb + l16][4 ks + lq]
          f32x4 d = {0.f, 0.f, 0.f, 0.f};
          static_for<4 * P>([&](auto ksc) {
            constexpr int ks = decltype(ksc)::value;
            d = __builtin_amdgcn_mfma_f32_16x16x4f32(As[off_a + 4 * ks], As[off_b + 4 * ks], d, 0, 0, 0);
          });
          // D[i = 4 lq + r][c = l16] -> image position (row 16 rb + 4 lq + r, column k0 + l16) where the row has storage there
          // (rows are padded to whole 4-float chunks: column <= row | 3); the positions above the diagonal are never read back
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * rb + 4 * lq + r;
            if ((row | 3) >= k0 + l16) As[chol_rowoff(row) + k0 + l16] = d[r];
          }
        });
#pragma unroll
        for (int j = 0; j < 16; j += 4) {  // this lane's row (lanes above the panel read other rows' words into values never used)
          const float4 t4 = *reinterpret_cast<const float4 *>(As + my_off + k0 + j);
          A(k0 + j) -= t4.x, A(k0 + j + 1) -= t4.y, A(k0 + j + 2) -= t4.z, A(k0 + j + 3) -= t4.w;
        }
      }
      f32x2 acc[4][2];
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c][0] = acc[c][1] = f32x2{0.f, 0.f};
      static_for<(k0 - j_first) / 4>([&](auto cc) {
        constexpr int j = j_first + 4 * decltype(cc)::value;
        float4 l[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) l[c] = *reinterpret_cast<const float4 *>(As + chol_rowoff(k0 + c) + j);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          acc[c][0] = __builtin_elementwise_fma(A2[j / 2], (f32x2){l[c].x, l[c].y}, acc[c][0]);
          acc[c][1] = __builtin_elementwise_fma(A2[j / 2 + 1], (f32x2){l[c].z, l[c].w}, acc[c][1]);
        }
      });
      float t[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) t[c] = A(k0 + c) - ((acc[c][0].x + acc[c][0].y) + (acc[c][1].x + acc[c][1].y));
      static_for<4>([&](auto cc) {
        constexpr int c = decltype(cc)::value, k = k0 + c;
        const float d = bcast_lane(t[c], k);  // pivot (a non-positive one turns the column, and L[k][k] with it, into NaNs: checked once, below)
        const float r0 = __builtin_amdgcn_rsqf(d);  // 1 / sqrt(d): v_rsq_f32 + one Newton step (see the column-wise form)
        const float inv = fmaf(r0, fmaf(-0.5f * d * r0, r0, 0.5f), r0);
        const float lik = t[c] * inv;  // L[i][k] for i >= k
        A(k) = lik;
#pragma unroll
        for (int c2 = c + 1; c2 < 4; ++c2) t[c2] = fmaf(-lik, bcast_lane(lik, k0 + c2), t[c2]);  // - L[i][k] L[k0+c2][k]
        // the forward substitution rides along: rows below k lose L[i][k] z_k, z_k = b_k / L_kk; lane k keeps its REDUCED
        // right-hand side b_k (z_k = b_k / L_kk is taken for all lanes at once after the loop: one select less per column)
        const float zk = bcast_lane(b, k) * inv;
        b = lane_v > k ? fmaf(-lik, zk, b) : b;
      });
      // rows k0 and below get their four new columns; a row inside the block writes words beyond its diagonal into its own
      // padding (rows are padded to whole 4-float chunks), which nothing reads
      if (lane_v >= k0) *reinterpret_cast<float4 *>(As + my_off + k0) = make_float4(A(k0), A(k0 + 1), A(k0 + 2), A(k0 + 3));
    });
    {
      // 1 / L[k][k] for the back substitution and the positive-definiteness check, once per row instead of a compare and a
      // select per column: lane k reads its diagonal from the image (d x rsq(d), a NaN or an infinity if any pivot up to k
      // was not positive) -- v_rcp_f32 + one Newton step
      const float ldiag = As[my_off + lane_v];
      const float r0 = __builtin_amdgcn_rcpf(ldiag);
      dinv = fmaf(r0, fmaf(-ldiag, r0, 1.f), r0);
      ok = __ballot(ldiag > 0.f && ldiag <= 3.0e38f) == ~0ull;
    }
    tick(3);
    if (!ok) {
      if (lane == 0) atomicMin(failed_row, (unsigned long long)u);
      continue;
    }
    // back substitution L^T x = z, column oriented: after x_k is known every lane i < k does z_i -= L[k][i] x_k; L[k][i] is
    // row k of the image read ACROSS the lanes (contiguous, conflict-free) -- one FMA per unknown.  The 64 reads do not
    // depend on the chain: they are all issued first (into the registers the row of L no longer needs).
#pragma unroll
    for (int k = 0; k < F; ++k) A(k) = As[chol_rowoff(k) + lane];  // L[k][lane] (meaningful for lane < k; beyond: other rows' words)
    __builtin_amdgcn_sched_barrier(0);
    b *= dinv * dinv;  // lane k: reduced b_k -> z_k = b_k / L[k][k] -> z_k / L[k][k]; the pending corrections are scaled the same way as they arrive
    static_for<F>([&](auto kc) {
      constexpr int k = F - 1 - decltype(kc)::value;
      const float xk = bcast_lane(b, k);  // x_k: lane k's value is final once all x_j, j > k, have been applied
      b = lane_v < k ? fmaf(-A(k) * dinv, xk, b) : b;
    });
    X[(size_t)u * F + lane] = b;
#undef A
    tick(4);
    if constexpr (STATS) tk[7] += 1;
  }
  if constexpr (STATS) {
    if (lane == 0)
      for (int i = 0; i < 8; ++i) atomicAdd(&stats[i], tk[i]);
  }
}

void zero_rows(const int32_t *order, int first, int count, float *X, int f);  // als_cg.hip

// returns -1, or the smallest failing row
int64_t least_squares_cholesky(const imp_csr *C, imp_matrix *X, const imp_matrix *YtY, const imp_matrix *Y, double reg) {
  const int f = (int)X->cols;
  if (f > 1024) throw std::invalid_argument("least_squares_cholesky: factors must be <= 1024 (as the reference's GPU kernels, als.cu:177-182)");
  // 64 < f < 128 (the reference's CPU default is 100): zero-padded onto the f = 128 path below -- Y and the gramian padded, the gramian
  // with a unit diagonal block: the padded system is block diagonal and its solution the original one followed by zeros (its 4 x 4
  // blocks factorise in the same order; the padded block rows never touch the others).  125 -> about 62 ms per configs[2]-shaped
  // iteration at f = 100 against the workgroup kernel.  IMP_CHOL_PAD=0 (and the switches that ask for the workgroup kernels) keep it.
  static const bool chol_pad = !(getenv("IMP_CHOL_PAD") && atoi(getenv("IMP_CHOL_PAD")) == 0) &&
                               !(getenv("IMP_CHOL_NM") && atoi(getenv("IMP_CHOL_NM")) == 0);
  // (not for a handful of rows against a large Y -- fold-in calls: the padded copy of Y would cost more than the solve)
  if (f > 64 && f < 128 && chol_pad && C->nonempty() > 0 && (size_t)C->nnz * 4 >= Y->rows && X->itemsize == 4 && Y->itemsize == 4) {
    constexpr int F = 128;
    const size_t rx = (size_t)C->rows;
    cholesky_pad_in(X, Y, YtY, rx, F);
    auto &c = ctx();
    imp_matrix Xp, Yp, Gp;
    Xp.rows = rx, Xp.cols = F, Xp.data = c.pad_x.data();
    Yp.rows = Y->rows, Yp.cols = F, Yp.data = c.pad_y.data();
    Gp.rows = F, Gp.cols = F, Gp.data = c.pad_gram.data();
    const int64_t failed = least_squares_cholesky(C, &Xp, &Gp, &Yp, reg);
    cholesky_pad_out(X, rx, F);
    sync();
    return failed;
  }
  int lda = (f + 1) | 1;  // odd
  // Packed rows of the augmented triangle (i (i + 1) / 2 + j): a must beyond f = 160, where the square image no longer fits the
  // LDS, and a gain well below that -- at f = 128 the packed image lets THREE workgroups share a CU instead of two (measured
  // 248 -> 174 ms per configs[2]-shaped iteration for one multiplication more per address)
  const bool packed = f >= 96;
  size_t lds = ((packed ? (size_t)(f + 1) * (f + 2) / 2 : (size_t)(f + 1) * lda) + (size_t)kCholTile * f + (size_t)kCholTile * (f + 1)) *
               sizeof(float);
  auto &failb = ctx().chol_failed;
  if (failb.size < 1) failb.alloc(1);
  unsigned long long *g_failed = failb.data();
  IMP_CHECK_HIP(hipMemsetAsync(g_failed, 0xFF, sizeof(unsigned long long), stream()));
  int nonempty = C->nonempty();
  if (nonempty > 0 && f == 64) {
    // every non-empty row: MFMA A-build + left-looking Cholesky, one wavefront per row
    const size_t lds_m = ((size_t)64 * 68 + 4 * kCholTri) * sizeof(float);  // 52 KB: 3 workgroups per CU
    // 159 VGPRs, 52 KB LDS: 3 workgroups per CU resident; 8x that many are launched -- smaller fixed shares of the length-sorted
    // schedule, dealt by the hardware dispatcher as slots free up, even out the end of the launch (configs[1]: 14.8 -> 13.8 ms)
    const int grid = std::min((nonempty + 3) / 4, ctx().num_cus * 3 * std::max(8, ctx().oversub));
    // long rows: segment partials of the A-build first (see als_cholesky_f64_partial_kernel)
    LongPlanDev plan = C->plan_chol.dev(C->order.data());
    const float *partials = nullptr;
    if (plan.n_seg > 0) {
      auto &ws = ctx().long_ws;
      const size_t need = (size_t)plan.n_seg * kCholPartial;
      if (ws.size < need) ws.alloc(need);
      partials = ws.data();
      IMP_PROF("als_cholesky_long_partials");
      const int pgrid = std::min((plan.n_seg + 3) / 4, ctx().num_cus * 8);
      als_cholesky_f64_partial_kernel<<<pgrid, 256, 0, stream()>>>(plan, C->indices.data(), C->data.data(), Y->f32(), ws.data());
      IMP_CHECK_HIP(hipGetLastError());
    }
    {
      IMP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(als_cholesky_f64_kernel<false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m));
      IMP_PROF("als_cholesky_mfma_rows");
      als_cholesky_f64_kernel<false><<<grid, 256, lds_m, stream()>>>(C->order.data(), 0, nonempty, C->indptr.data(), C->indices.data(),
                                                                     C->data.data(), X->f32(), Y->f32(), YtY->f32(), (float)reg,
                                                                     g_failed, nullptr, plan, partials);
      IMP_CHECK_HIP(hipGetLastError());
    }
    zero_rows(C->order.data(), C->first_empty(), C->n_empty(), X->f32(), f);
    unsigned long long failed_m = 0;
    IMP_CHECK_HIP(hipMemcpyAsync(&failed_m, g_failed, sizeof(failed_m), hipMemcpyDeviceToHost, stream()));
    sync();
    return failed_m == ~0ULL ? -1 : (int64_t)failed_m;
  }
  // f = 128 (round 5): the rows' normal matrices on the matrix cores, factorised on their LDS images (als_cg_nm.hip); the
  // workgroup kernel below then only takes the rows that path listed (non-positive or non-finite pivots: normally none).
  // IMP_CHOL_NM=0: every row on the workgroup kernel (A/B, parity)
  static const bool chol_nm = !(getenv("IMP_CHOL_NM") && atoi(getenv("IMP_CHOL_NM")) == 0);
  CholNmList nm_list{nullptr, nullptr, 0};
  const bool use_nm = f == 128 && chol_nm && nonempty > 0;
  if (use_nm) nm_list = least_squares_cholesky_nm(C, X->f32(), Y->f32(), Y->rows, YtY->f32(), (float)reg);
  // other f <= 64: rows up to 256 nnz go to the register-resident wave kernel; longer rows (and any larger f) to the
  // workgroup kernel, whose 256 threads share the A-build of one row
  const int n_block = f <= 64 ? C->bin_start[2] : nonempty;
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
    {
      const int m = f + 1, nbr = (m + 3) / 4, nbc = (f + 3) / 4;
      const size_t a_words = packed ? (size_t)m * (m + 1) / 2 : (size_t)m * lda;
      lds = (((a_words + 3) & ~(size_t)3) + (size_t)kCholTile * 4 * (nbc + nbr) + (size_t)m * kCholPanel + 4) * sizeof(float);
    }
    if (f > 256) {
      // the triangle in a device workspace (als_cholesky_blocked_kernel<.., GLOBAL>): two workgroups per CU, a slice each
      const int m = f + 1, nbr = (m + 3) / 4, nbc = (f + 3) / 4;
      const size_t a_words = ((size_t)m * lda + 3) & ~(size_t)3;
      lds = ((size_t)kCholTile * 4 * (nbc + nbr) + (size_t)m * kCholPanel + 4) * sizeof(float);
      const int grid = std::min(n_block, ctx().num_cus * 2);
      auto &ws = ctx().long_ws;
      if (ws.size < a_words * (size_t)grid) ws.alloc(a_words * (size_t)grid);
      IMP_PROF("als_cholesky_rows");
      auto kern = als_cholesky_blocked_kernel<false, true, 17, 16>;
      IMP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      kern<<<grid, 256, lds, stream()>>>(C->order.data(), 0, n_block, C->indptr.data(), C->indices.data(), C->data.data(), X->f32(),
                                         Y->f32(), YtY->f32(), f, (float)reg, lda, g_failed, 0, nullptr, ws.data());
      IMP_CHECK_HIP(hipGetLastError());
    } else {
    int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / lds));
      int grid = std::min(n_block, ctx().num_cus * per_cu * 4);  // smaller fixed shares of the length-sorted schedule
      IMP_PROF("als_cholesky_rows");
      constexpr int ko = 0;  // (timing-only knock-outs of the phases: 1 A-build, 2 panel, 4 trailing update, 8 back substitution)
      auto kern = packed ? als_cholesky_blocked_kernel<true> : als_cholesky_blocked_kernel<false>;
      IMP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      if (use_nm)  // the listed rows only (the list stands in for the schedule; a zero count makes every workgroup return at once)
        kern<<<std::min(nm_list.capacity, ctx().num_cus * per_cu), 256, lds, stream()>>>(
            reinterpret_cast<const int32_t *>(nm_list.rows), 0, nm_list.capacity, C->indptr.data(), C->indices.data(), C->data.data(),
            X->f32(), Y->f32(), YtY->f32(), f, (float)reg, lda, g_failed, ko, nm_list.count, nullptr);
      else
        kern<<<grid, 256, lds, stream()>>>(C->order.data(), 0, n_block, C->indptr.data(), C->indices.data(), C->data.data(), X->f32(),
                                           Y->f32(), YtY->f32(), f, (float)reg, lda, g_failed, ko, nullptr, nullptr);
      IMP_CHECK_HIP(hipGetLastError());
    }
  }
  zero_rows(C->order.data(), C->first_empty(), C->n_empty(), X->f32(), f);
  unsigned long long failed = 0;
  IMP_CHECK_HIP(hipMemcpyAsync(&failed, g_failed, sizeof(failed), hipMemcpyDeviceToHost, stream()));
  sync();
  return failed == ~0ULL ? -1 : (int64_t)failed;
}

}  // namespace imp
