// K1: warm-started conjugate-gradient half sweep  X[u] <- CG_s( (YtY + reg I) + Y_u^T (C_u - I) Y_u , Y_u^T C_u p_u )
//
// Numerics follow the reference's CPU oracle step by step (implicit/cpu/_als.pyx:152-248, SURVEY
// App. A.1): residual with the confidence branch, rsold < 1e-20 early-out (x untouched), <= cg_steps
// iterations, rsnew < 1e-20 break, empty rows zeroed.  It replaces the reference's CUDA launcher
// and kernel (implicit/gpu/als.cu:23-111,154-197) behind LeastSquaresSolver::least_squares.
//
// MI355X mapping (not the reference's one-thread-per-factor block):
//   * a 64-lane wavefront owns a row; lane l holds VPL consecutive factors of x, r, p, Ap in
//     registers, so every gathered factor row Y[i,:] is ONE fully coalesced wave load
//     (dwordx2 at f=128, dwordx4 at f=256) and the dot / axpy of the oracle's inner loop are
//     2*VPL FMAs per lane plus one DPP wave reduction -- no LDS round trip, no block barrier per nnz
//     (the reference does two __syncthreads per nnz, dot.cuh:27-59);
//   * (YtY + reg I) is staged once per workgroup in LDS (64 KiB at f=128) and applied as a
//     broadcast mat-vec: p_j comes from v_readlane, row j of the gramian from one conflict-free
//     ds_read per lane;
//   * rows are scheduled by length class (imp_csr::order): short rows one wave each, long rows one
//     workgroup each with the nnz and the gramian rows split over its waves and the partial
//     vectors combined through LDS in a fixed order (deterministic, identical in every wave).
#include "common.h"
#include "wave_ops.h"

namespace imp {

// acc += A0[j0..j1) contribution of the symmetric mat-vec: acc[e] += sum_j A0[j][e] * vec[j]
// A0s: LDS image (leading dimension LD = 64*VPL, zero padded) or the global f x f matrix (LD = f).
template <int VPL, bool VEC>
__device__ __forceinline__ void gram_matvec(const float *A0s, int LD, int lane, const float (&vec)[VPL],
                                            float (&acc)[VPL], int j_begin, int j_end) {
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    // j = elem(l, v): iterate over the lanes that own slot v
    for (int l = 0; l < 64; ++l) {
      int j = VEC ? l * VPL + v : l + 64 * v;
      if (j < j_begin || j >= j_end) continue;  // wave-uniform
      float pj = lane_bcast(vec[v], l);
      float row[VPL];
      load_row<VPL, VEC>(A0s + (size_t)j * LD, LD, lane, row);
#pragma unroll
      for (int w = 0; w < VPL; ++w) acc[w] = fmaf(pj, row[w], acc[w]);
    }
  }
}

// One pass over (a slice of) the row's nonzeros:  acc += sum_k w_k * y_k with
//   FIRST : w = (c > 0 ? c : 0) - (|c| - 1) * (y_k . vec)      (_als.pyx:190-201)
//   else  : w = (|c| - 1) * (y_k . vec)                         (_als.pyx:214-222)
template <int VPL, bool VEC, bool FIRST>
__device__ __forceinline__ void sparse_pass(const int32_t *__restrict__ indices, const float *__restrict__ data,
                                            const float *__restrict__ Y, int f, int lane, int begin, int end,
                                            int chunk_stride, const float (&vec)[VPL], float (&acc)[VPL]) {
  constexpr int U = 4;
  for (int k0 = begin; k0 < end; k0 += chunk_stride) {
    int cnt = min(64, end - k0);
    int my_idx = 0;
    float my_c = 1.f;  // |c| - 1 = 0 and c > 0 ... neutralised below through the count guard
    if (lane < cnt) {
      my_idx = indices[k0 + lane];
      my_c = data[k0 + lane];
    }
    for (int j = 0; j < cnt; j += U) {
      float y[U][VPL];
      float d[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        int jj = min(j + u, cnt - 1);
        int col = lane_bcast(my_idx, jj);
        load_row<VPL, VEC>(Y + (size_t)col * f, f, lane, y[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) d[u] = wave_allsum(dot_local<VPL>(y[u], vec));
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (j + u < cnt) {  // wave-uniform
          float c = lane_bcast(my_c, j + u);
          float w;
          if (FIRST) {
            float t = c > 0.f ? c : 0.f;
            float a = c > 0.f ? c : -c;
            w = t - (a - 1.f) * d[u];
          } else {
            float a = c < 0.f ? -c : c;
            w = (a - 1.f) * d[u];
          }
#pragma unroll
          for (int v = 0; v < VPL; ++v) acc[v] = fmaf(w, y[u][v], acc[v]);
        }
      }
    }
  }
}

// Fixed-order sum of the WPR per-wave partial vectors through LDS; every wave ends with the same bits.
template <int VPL, int WPR>
__device__ __forceinline__ void combine(float *scratch, int wave, int lane, float (&acc)[VPL]) {
  if constexpr (WPR > 1) {
    constexpr int LD = 64 * VPL;
    __syncthreads();  // previous readers done
#pragma unroll
    for (int v = 0; v < VPL; ++v) scratch[wave * LD + lane * VPL + v] = acc[v];
    __syncthreads();
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < WPR; ++w) s += scratch[w * LD + lane * VPL + v];
      acc[v] = s;
    }
  }
}

// BLOCK threads; WPR waves cooperate on one row (WPR == 1: each wave walks its own rows).
template <int VPL, bool VEC, int WPR, int BLOCK, bool A_LDS>
__global__ __launch_bounds__(BLOCK) void als_cg_kernel(const int32_t *__restrict__ order, int first, int count,
                                                       const int32_t *__restrict__ indptr,
                                                       const int32_t *__restrict__ indices,
                                                       const float *__restrict__ data, float *__restrict__ X,
                                                       const float *__restrict__ Y, const float *__restrict__ A0,
                                                       int f, int cg_steps) {
  constexpr int LD = 64 * VPL;
  constexpr int WAVES = BLOCK / 64;
  constexpr int GROUPS = WAVES / WPR;  // rows in flight per block
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *A0s = smem;                                   // [f][LD] when A_LDS
  float *scratch = smem + (A_LDS ? (size_t)f * LD : 0);  // [GROUPS][WPR][LD]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int group = wave / WPR, sub = wave % WPR;

  const float *Amat;
  if constexpr (A_LDS) {
    for (int i = threadIdx.x; i < f * LD; i += BLOCK) {
      int r = i / LD, c = i - r * LD;
      A0s[i] = c < f ? A0[(size_t)r * f + c] : 0.f;
    }
    __syncthreads();
    Amat = A0s;
  } else {
    Amat = A0;
  }
  const int lda = A_LDS ? LD : f;
  float *my_scratch = scratch + (size_t)group * WPR * LD;

  // gramian rows handled by this wave in the dense mat-vec
  const int j_begin = (int)((long)f * sub / WPR), j_end = (int)((long)f * (sub + 1) / WPR);

  for (int i = blockIdx.x * GROUPS + group; i < count; i += gridDim.x * GROUPS) {
    const int u = __builtin_amdgcn_readfirstlane(order[first + i]);
    const int row_begin = __builtin_amdgcn_readfirstlane(indptr[u]);
    const int row_end = __builtin_amdgcn_readfirstlane(indptr[u + 1]);
    float *xrow = X + (size_t)u * f;
    float x[VPL], r[VPL], p[VPL], Ap[VPL];
    load_row<VPL, VEC>(xrow, f, lane, x);

    // r = -(A0 x) + sum_k (c+ - (|c|-1) y.x) y        (_als.pyx:187-201)
#pragma unroll
    for (int v = 0; v < VPL; ++v) Ap[v] = 0.f;
    gram_matvec<VPL, VEC>(Amat, lda, lane, x, Ap, j_begin, j_end);
#pragma unroll
    for (int v = 0; v < VPL; ++v) r[v] = -Ap[v];
    sparse_pass<VPL, VEC, true>(indices, data, Y, f, lane, row_begin + sub * 64, row_end, 64 * WPR, x, r);
    combine<VPL, WPR>(my_scratch, sub, lane, r);

#pragma unroll
    for (int v = 0; v < VPL; ++v) p[v] = r[v];
    float rsold = wave_allsum(dot_local<VPL>(r, r));
    if (rsold >= 1e-20f) {  // else: leave x untouched (_als.pyx:206)
      for (int it = 0; it < cg_steps; ++it) {
#pragma unroll
        for (int v = 0; v < VPL; ++v) Ap[v] = 0.f;
        gram_matvec<VPL, VEC>(Amat, lda, lane, p, Ap, j_begin, j_end);
        sparse_pass<VPL, VEC, false>(indices, data, Y, f, lane, row_begin + sub * 64, row_end, 64 * WPR, p, Ap);
        combine<VPL, WPR>(my_scratch, sub, lane, Ap);

        float alpha = rsold / wave_allsum(dot_local<VPL>(p, Ap));
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
          x[v] = fmaf(alpha, p[v], x[v]);
          r[v] = fmaf(-alpha, Ap[v], r[v]);
        }
        float rsnew = wave_allsum(dot_local<VPL>(r, r));
        if (rsnew < 1e-20f) break;
        float beta = rsnew / rsold;
#pragma unroll
        for (int v = 0; v < VPL; ++v) p[v] = fmaf(beta, p[v], r[v]);
        rsold = rsnew;
      }
      if (sub == 0) {
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
          int e = elem<VPL, VEC>(lane, v);
          if (e < f) xrow[e] = x[v];
        }
      }
    }
  }
}

__global__ void zero_rows_kernel(const int32_t *__restrict__ order, int first, int count, float *__restrict__ X, int f) {
  size_t total = (size_t)count * f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t r = i / f, c = i - r * f;
    X[(size_t)order[first + r] * f + c] = 0.f;
  }
}

void zero_rows(const int32_t *order, int first, int count, float *X, int f) {
  if (count <= 0) return;
  IMP_PROF("zero_rows");
  size_t total = (size_t)count * f;
  int grid = (int)std::min<size_t>((total + 255) / 256, (size_t)ctx().num_cus * 8);
  zero_rows_kernel<<<grid, 256, 0, stream()>>>(order, first, count, X, f);
  IMP_CHECK_HIP(hipGetLastError());
}

template <int VPL, bool VEC, int WPR, int BLOCK, bool A_LDS>
static void launch_bin(const imp_csr *C, int first, int count, float *X, const float *Y, const float *A0, int f,
                       int cg_steps, const char *name) {
  if (count <= 0) return;
  constexpr int LD = 64 * VPL;
  constexpr int GROUPS = (BLOCK / 64) / WPR;
  size_t lds = (A_LDS ? (size_t)f * LD : 0) * sizeof(float) + (size_t)(BLOCK / 64) * LD * sizeof(float);
  auto kern = als_cg_kernel<VPL, VEC, WPR, BLOCK, A_LDS>;
  IMP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int blocks_per_cu = std::max<size_t>(1, std::min<size_t>(2048 / BLOCK, (160 * 1024) / std::max<size_t>(lds, 1)));
  int grid = std::min((count + GROUPS - 1) / GROUPS, ctx().num_cus * blocks_per_cu);
  IMP_PROF(name);
  kern<<<grid, BLOCK, lds, stream()>>>(C->order.data(), first, count, C->indptr.data(), C->indices.data(),
                                       C->data.data(), X, Y, A0, f, cg_steps);
  IMP_CHECK_HIP(hipGetLastError());
}

template <int VPL, bool VEC, bool A_LDS>
static void launch_all(const imp_csr *C, float *X, const float *Y, const float *A0, int f, int cg_steps) {
  constexpr int BLOCK = 512;
  // bin 0: long rows, one workgroup per row; bin 1: one wave per row; bin 2: empty rows
  launch_bin<VPL, VEC, BLOCK / 64, BLOCK, A_LDS>(C, C->bin_start[0], C->bin_start[1] - C->bin_start[0], X, Y, A0, f,
                                                 cg_steps, "als_cg_block_rows");
  launch_bin<VPL, VEC, 1, BLOCK, A_LDS>(C, C->bin_start[1], C->bin_start[2] - C->bin_start[1], X, Y, A0, f, cg_steps,
                                        "als_cg_wave_rows");
  zero_rows(C->order.data(), C->bin_start[2], C->bin_start[3] - C->bin_start[2], X, f);
}

void least_squares_cg(const imp_csr *C, imp_matrix *X, const imp_matrix *YtY, const imp_matrix *Y, int cg_steps) {
  const int f = (int)X->cols;
  float *x = X->f32();
  const float *y = Y->f32();
  const float *a0 = YtY->f32();
  if (f == 64) launch_all<1, true, true>(C, x, y, a0, f, cg_steps);
  else if (f == 128) launch_all<2, true, true>(C, x, y, a0, f, cg_steps);
  else if (f == 256) launch_all<4, true, false>(C, x, y, a0, f, cg_steps);
  else if (f < 64) launch_all<1, false, true>(C, x, y, a0, f, cg_steps);
  else if (f < 128) launch_all<2, false, true>(C, x, y, a0, f, cg_steps);
  else if (f < 192) launch_all<3, false, true>(C, x, y, a0, f, cg_steps);
  else if (f < 256) launch_all<4, false, false>(C, x, y, a0, f, cg_steps);
  else if (f <= 384) launch_all<6, false, false>(C, x, y, a0, f, cg_steps);
  else if (f <= 512) launch_all<8, false, false>(C, x, y, a0, f, cg_steps);
  else throw std::invalid_argument("least_squares: factors must be <= 512 in this build");
}

}  // namespace imp
