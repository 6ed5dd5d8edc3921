// K1: warm-started conjugate-gradient half sweep  X[u] <- CG_s( (YtY + reg I) + Y_u^T (C_u - I) Y_u , Y_u^T C_u p_u )
//
// Numerics follow the reference's CPU oracle step by step (implicit/cpu/_als.pyx:152-248, SURVEY
// App. A.1): residual with the confidence branch, rsold < 1e-20 early-out (x untouched), <= cg_steps
// iterations, rsnew < 1e-20 break, empty rows zeroed.  It replaces the reference's CUDA launcher
// and kernel (implicit/gpu/als.cu:23-111,154-197) behind LeastSquaresSolver::least_squares.
//
// MI355X mapping (not the reference's one-thread-per-factor block with a block reduction per nnz):
//   * a 64-lane wavefront owns a row; lane l holds VPL consecutive factors of x, r, p, Ap in registers, so
//     every gathered factor row Y[i,:] is ONE fully coalesced wave load (dwordx2 at f=128);
//   * the nonzeros of a row are processed in TILES of T gathered rows held in registers: T partial dot
//     products per lane are reduced with a butterfly REDUCE-SCATTER (v_permlane32_swap, v_permlane16_swap,
//     then DPP row rotations) -- ~2.5 cross-lane instructions per dot instead of 7 -- the T weights are
//     formed in the lanes that own them, broadcast with v_readlane and applied as T axpys;
//   * rows with <= T nonzeros keep their gathered tile in registers across the 1+cg_steps passes, so
//     their factor rows are read from memory exactly once (the roofline's single-pass traffic);
//   * (YtY + reg I) is staged once per workgroup in LDS (64 KiB at f=128) and applied as a broadcast
//     mat-vec: p_j from v_readlane, row j of the gramian from one conflict-free ds_read per lane;
//   * rows are scheduled by length class (imp_csr::order): short (resident tile), mid (streamed
//     tiles), and LONG rows (> kLongRow nnz) which are cut into segments: every CG pass becomes a
//     segment-parallel partial kernel plus a per-row combine/update kernel, so a 150K-nnz row is
//     spread over the whole chip instead of serialising one workgroup (fixed summation order).
//   * any other f <= 512 runs a generic lane-strided variant of the same structure.
#include <type_traits>

#include "common.h"
#include "wave_ops.h"
#include "als_qtile.h"
#include "als_tile.h"

namespace imp {

template <typename T> void least_squares_cg_q(const imp_csr *C, T *X, const T *Y, const float *A0, int f, int cg_steps);  // als_cg_q.hip
void least_squares_cg_w256(const imp_csr *C, float *X, const float *Y, const float *A0, int cg_steps);  // als_cg_w256.hip
template <typename T> void least_squares_cg_nm(const imp_csr *C, T *X, const T *Y, size_t y_rows, const float *A0, int f, int cg_steps);  // als_cg_nm.hip

// ---- generic per-nnz pass (any f): 4 gathers in flight, one DPP all-reduce per dot -------------------
template <int VPL, bool VEC, bool FIRST>
__device__ __forceinline__ void sparse_pass_simple(const int32_t *__restrict__ indices, const float *__restrict__ data,
                                                   const float *__restrict__ Y, int f, int lane, int begin, int end,
                                                   const float (&vec)[VPL], float (&acc)[VPL]) {
  constexpr int U = 4;
  for (int k0 = begin; k0 < end; k0 += 64) {
    int cnt = min(64, end - k0);
    int my_idx = 0;
    float my_c = 0.f;
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
          float w = nnz_weight<FIRST>(lane_bcast(my_c, j + u), d[u]);
#pragma unroll
          for (int v = 0; v < VPL; ++v) acc[v] = fmaf(w, y[u][v], acc[v]);
        }
      }
    }
  }
}

template <int VPL, bool VEC, bool FIRST>
__device__ __forceinline__ void sparse_pass(const int32_t *__restrict__ indices, const float *__restrict__ data,
                                            const float *__restrict__ Y, int f, int lane, int begin, int end,
                                            const float (&vec)[VPL], float (&acc)[VPL]) {
  if constexpr (VEC)
    sparse_pass_tiled<VPL, tile_size<VPL>(), FIRST>(indices, data, Y, f, lane, begin, end, vec, acc);
  else
    sparse_pass_simple<VPL, VEC, FIRST>(indices, data, Y, f, lane, begin, end, vec, acc);
}

template <int VPL, bool VEC, int BLOCK, bool A_LDS>
__device__ __forceinline__ const float *stage_gramian(float *smem, const float *__restrict__ A0, int f) {
  constexpr int LD = 64 * VPL;
  if constexpr (A_LDS) {
    for (int i = threadIdx.x; i < f * LD; i += BLOCK) {
      int r = i / LD, c = i - r * LD;
      smem[i] = c < f ? A0[(size_t)r * f + c] : 0.f;
    }
    __syncthreads();
    return smem;
  } else {
    return A0;
  }
}

// ---- fused kernel: one wavefront per row, rows [first, first+count) of the schedule -------------------------
// RESIDENT: every row has <= T nonzeros and its gathered tile stays in registers across all passes.
template <int VPL, bool VEC, int BLOCK, bool A_LDS, bool RESIDENT>
__global__ __launch_bounds__(BLOCK, (A_LDS && VPL == 2) ? 4 : 2) void als_cg_kernel(const int32_t *__restrict__ order, int first, int count,
                                                       const int32_t *__restrict__ indptr,
                                                       const int32_t *__restrict__ indices,
                                                       const float *__restrict__ data, float *__restrict__ X,
                                                       const float *__restrict__ Y, const float *__restrict__ A0,
                                                       int f, int cg_steps) {
  constexpr int LD = 64 * VPL;
  constexpr int WAVES = BLOCK / 64;
  constexpr int T = tile_size<VPL>();
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const float *Amat = stage_gramian<VPL, VEC, BLOCK, A_LDS>(smem, A0, f);
  const int lda = A_LDS ? LD : f;

  for (int i = blockIdx.x * WAVES + wave; i < count; i += gridDim.x * WAVES) {
    const int u = __builtin_amdgcn_readfirstlane(order[first + i]);
    const int row_begin = __builtin_amdgcn_readfirstlane(indptr[u]);
    const int row_end = __builtin_amdgcn_readfirstlane(indptr[u + 1]);
    float *xrow = X + (size_t)u * f;
    float x[VPL], r[VPL], p[VPL], Ap[VPL];
    load_row<VPL, VEC>(xrow, f, lane, x);

    Tile<VPL, T> tile;
    if constexpr (RESIDENT) load_tile<VPL, T>(tile, indices, data, Y, f, lane, row_begin, row_end);

    // r = -(A0 x) + sum_k (c+ - (|c|-1) y.x) y        (_als.pyx:187-201)
#pragma unroll
    for (int v = 0; v < VPL; ++v) Ap[v] = 0.f;
    gram_matvec<VPL, VEC>(Amat, lda, lane, x, Ap, 0, f);
#pragma unroll
    for (int v = 0; v < VPL; ++v) r[v] = -Ap[v];
    if constexpr (RESIDENT)
      tile_apply<VPL, T, true>(tile, lane, row_begin, row_end, x, r);
    else
      sparse_pass<VPL, VEC, true>(indices, data, Y, f, lane, row_begin, row_end, x, r);

#pragma unroll
    for (int v = 0; v < VPL; ++v) p[v] = r[v];
    float rsold = wave_allsum(dot_local<VPL>(r, r));
    if (rsold >= 1e-20f) {  // else: leave x untouched (_als.pyx:206)
      for (int it = 0; it < cg_steps; ++it) {
#pragma unroll
        for (int v = 0; v < VPL; ++v) Ap[v] = 0.f;
        gram_matvec<VPL, VEC>(Amat, lda, lane, p, Ap, 0, f);
        if constexpr (RESIDENT)
          tile_apply<VPL, T, false>(tile, lane, row_begin, row_end, p, Ap);
        else
          sparse_pass<VPL, VEC, false>(indices, data, Y, f, lane, row_begin, row_end, p, Ap);

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
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        int e = elem<VPL, VEC>(lane, v);
        if (e < f) xrow[e] = x[v];
      }
    }
  }
}

// ---- f = 256: one wavefront per row, the gramian shared by the workgroup --------------------------------------------
// 256 KB of gramian fit no LDS, and read from L2 by every wavefront for every pass they are two thirds of the generic
// kernel's traffic (a 140-nnz row gathers 140 KB per pass and reads 256 KB of gramian).  Here the 8 wavefronts of a workgroup
// run the passes of their 8 rows in lock step and stage the gramian through LDS in slices of 32 rows (32 KB): one L2 read
// per workgroup and pass instead of eight.  Early exits become per-wave predicates (every wave takes every barrier); the
// arithmetic per row is the generic kernel's, operation for operation.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK, BLOCK / 128) void als_cg_f256_kernel(const int32_t *__restrict__ order, int first, int count,
                                                                       const int32_t *__restrict__ indptr,
                                                                       const int32_t *__restrict__ indices,
                                                                       const float *__restrict__ data, float *__restrict__ X,
                                                                       const float *__restrict__ Y, const float *__restrict__ A0,
                                                                       int cg_steps) {
  constexpr int VPL = 4, F = 256, WAVES = BLOCK / 64, SL = 32;
  __shared__ __attribute__((aligned(16))) float slice[SL * F];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // acc += A0 vec (factor j of the operand lives in lane j / 4, slot j % 4), all waves together
  auto dense = [&](const float (&vec)[VPL], float (&acc)[VPL], bool work) {
    for (int s = 0; s < F / SL; ++s) {
      __syncthreads();  // the previous slice has been consumed
      for (int e = threadIdx.x; e < SL * F / 4; e += BLOCK)
        reinterpret_cast<float4 *>(slice)[e] = reinterpret_cast<const float4 *>(A0 + (size_t)SL * s * F)[e];
      __syncthreads();
      if (work) {  // wave-uniform
#pragma unroll 1
        for (int jj0 = 0; jj0 < SL; jj0 += 8) {  // 8 rows of the slice in flight (all 32 at once cost 128 registers)
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int jj = jj0 + q;
            const float pj = lane_bcast(vec[q & 3], (SL / 4) * s + (jj >> 2));
            const float4 row = reinterpret_cast<const float4 *>(slice + jj * F)[lane];
            acc[0] = fmaf(pj, row.x, acc[0]);
            acc[1] = fmaf(pj, row.y, acc[1]);
            acc[2] = fmaf(pj, row.z, acc[2]);
            acc[3] = fmaf(pj, row.w, acc[3]);
          }
        }
      }
    }
  };
  for (int i0 = blockIdx.x * WAVES; i0 < count; i0 += gridDim.x * WAVES) {
    const bool valid = i0 + wave < count;
    const int u = __builtin_amdgcn_readfirstlane(order[first + min(i0 + wave, count - 1)]);
    const int row_begin = __builtin_amdgcn_readfirstlane(indptr[u]);
    const int row_end = __builtin_amdgcn_readfirstlane(indptr[u + 1]);
    float *xrow = X + (size_t)u * F;
    float x[VPL], r[VPL], p[VPL], Ap[VPL];
    load_row<VPL, true>(xrow, F, lane, x);
    // r = -(A0 x) + sum_k (c+ - (|c|-1) y.x) y        (_als.pyx:187-201)
#pragma unroll
    for (int v = 0; v < VPL; ++v) Ap[v] = 0.f;
    dense(x, Ap, valid);
#pragma unroll
    for (int v = 0; v < VPL; ++v) r[v] = -Ap[v];
    if (valid) sparse_pass<VPL, true, true>(indices, data, Y, F, lane, row_begin, row_end, x, r);
#pragma unroll
    for (int v = 0; v < VPL; ++v) p[v] = r[v];
    float rsold = wave_allsum(dot_local<VPL>(r, r));
    bool active = valid && rsold >= 1e-20f;  // else: leave x untouched (_als.pyx:206)
    const bool store = active;
    for (int it = 0; it < cg_steps; ++it) {
#pragma unroll
      for (int v = 0; v < VPL; ++v) Ap[v] = 0.f;
      dense(p, Ap, active);
      if (active) {
        sparse_pass<VPL, true, false>(indices, data, Y, F, lane, row_begin, row_end, p, Ap);
        float alpha = rsold / wave_allsum(dot_local<VPL>(p, Ap));
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
          x[v] = fmaf(alpha, p[v], x[v]);
          r[v] = fmaf(-alpha, Ap[v], r[v]);
        }
        float rsnew = wave_allsum(dot_local<VPL>(r, r));
        if (rsnew < 1e-20f) {
          active = false;  // the oracle breaks here (_als.pyx:235)
        } else {
          float beta = rsnew / rsold;
#pragma unroll
          for (int v = 0; v < VPL; ++v) p[v] = fmaf(beta, p[v], r[v]);
          rsold = rsnew;
        }
      }
    }
    if (store) {
#pragma unroll
      for (int v = 0; v < VPL; ++v) xrow[elem<VPL, true>(lane, v)] = x[v];
    }
  }
}

static void launch_f256(const imp_csr *C, int first, int count, float *X, const float *Y, const float *A0, int cg_steps,
                        const char *name) {
  if (count <= 0) return;
  constexpr int BLOCK = 512;
  int grid = std::min((count + BLOCK / 64 - 1) / (BLOCK / 64), ctx().num_cus * 2 * ctx().oversub);
  IMP_PROF(name);
  als_cg_f256_kernel<BLOCK><<<grid, BLOCK, 0, stream()>>>(C->order.data(), first, count, C->indptr.data(), C->indices.data(),
                                                         C->data.data(), X, Y, A0, cg_steps);
  IMP_CHECK_HIP(hipGetLastError());
}

// ---- long rows: every CG pass = segment-parallel partial kernel + per-row combine/update kernel -------------
// workspace (floats): partial[n_seg][LD] | rvec[n_long][LD] | pvec[n_long][LD] | scal[n_long][2] (rsold, done)
template <int VPL, bool VEC, bool FIRST>
__global__ __launch_bounds__(256) void cg_long_partial_kernel(const LongPlanDev plan, const int32_t *__restrict__ indices,
                                                              const float *__restrict__ data, const float *__restrict__ X,
                                                              const float *__restrict__ Y, int f, float *__restrict__ partial,
                                                              const float *__restrict__ pvec, const float *__restrict__ scal) {
  constexpr int LD = 64 * VPL;
  const int lane = threadIdx.x & 63;
  // the workgroups with blockIdx % 8 == x run on XCD x (observed placement; a speed matter only) and sweep that
  // XCD's part of the execution order (imp_csr_create: column stripes dealt to the XCDs)
  const int xcd = blockIdx.x & 7;
  const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x >> 3) * (blockDim.x >> 6) + (threadIdx.x >> 6)));
  const int nwaves = (gridDim.x >> 3) * (blockDim.x >> 6);
  for (int i = plan.xcd_start[xcd] + wave; i < plan.xcd_start[xcd + 1]; i += nwaves) {
    const int s = __builtin_amdgcn_readfirstlane(plan.seg_exec[i]);
    const int li = __builtin_amdgcn_readfirstlane(plan.seg_row[s]);
    if (!FIRST && scal[2 * li + 1] != 0.f) continue;  // row finished (early exit)
    const int begin = __builtin_amdgcn_readfirstlane(plan.seg_begin[s]);
    const int end = __builtin_amdgcn_readfirstlane(plan.seg_end[s]);
    float vec[VPL], acc[VPL];
    if (FIRST)
      load_row<VPL, VEC>(X + (size_t)plan.rows[li] * f, f, lane, vec);
    else
      load_row<VPL, VEC>(pvec + (size_t)li * LD, VEC ? LD : f, lane, vec);
#pragma unroll
    for (int v = 0; v < VPL; ++v) acc[v] = 0.f;
    sparse_pass<VPL, VEC, FIRST>(indices, data, Y, f, lane, begin, end, vec, acc);
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      int e = elem<VPL, VEC>(lane, v);
      partial[(size_t)s * LD + e] = acc[v];
    }
  }
}

// Quarter-layout variant of the partial kernel (f = 64, 128): same segments, ~half the VALU work per tile pass.
// The operand vector is read from memory directly in expanded form; the partial result is stored by factor index,
// so cg_long_combine_kernel is unchanged.
template <int F, bool FIRST, typename T>
__global__ __launch_bounds__(256) void cg_long_partial_q_kernel(const LongPlanDev plan, const int32_t *__restrict__ indices,
                                                                const float *__restrict__ data, const T *__restrict__ X,
                                                                const T *__restrict__ Y, float *__restrict__ partial,
                                                                const float *__restrict__ pvec, const float *__restrict__ scal) {
  constexpr int FE = F / 16, FC = F / 64, LD = F;
  const int lane = threadIdx.x & 63;
  const int xcd = blockIdx.x & 7;  // see cg_long_partial_kernel
  const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x >> 3) * (blockDim.x >> 6) + (threadIdx.x >> 6)));
  const int nwaves = (gridDim.x >> 3) * (blockDim.x >> 6);
  const int stop = plan.xcd_start[xcd + 1];
  int i = plan.xcd_start[xcd] + wave;
  if (i >= stop) return;
  // Striped plans have many short segments (a few entries of one row inside one column stripe), so the loop is
  // pipelined like the row loop of the resident kernels: segment descriptors are read two segments ahead (scalar
  // loads), the (column, confidence) pairs of the next tile -- of this segment or of the next one -- one tile ahead.
  auto descriptor = [&](int at, int &s, int &li, int &begin, int &end) {
    s = plan.seg_exec[min(at, stop - 1)];
    li = plan.seg_row[s];
    begin = plan.seg_begin[s];
    end = plan.seg_end[s];
  };
  int s1, li1, b1, e1, s2, li2, b2, e2, col_next;
  float c_next;
  descriptor(i, s1, li1, b1, e1);
  descriptor(i + nwaves, s2, li2, b2, e2);
  fetch_entries(indices, data, lane, b1, e1, col_next, c_next);
  for (; i < stop; i += nwaves) {
    const int s = s1, li = li1, begin = b1, end = e1;
    s1 = s2, li1 = li2, b1 = b2, e1 = e2;
    descriptor(i + 2 * nwaves, s2, li2, b2, e2);
    const bool skip = !FIRST && scal[2 * li + 1] != 0.f;  // row finished (early exit): no arithmetic, no partial
    float ve[FE], ae[FE];
#pragma unroll
    for (int e = 0; e < FE; e += 4) {
      float4 v;  // the operand in expanded form: x of the row (factor storage type) or the fp32 search direction
      if constexpr (FIRST) v = load4(X + (size_t)plan.rows[li] * F + 4 * (lane & 15) + 16 * e);
      else v = load4(pvec + (size_t)li * LD + 4 * (lane & 15) + 16 * e);
      ve[e] = v.x, ve[e + 1] = v.y, ve[e + 2] = v.z, ve[e + 3] = v.w;
      ae[e] = ae[e + 1] = ae[e + 2] = ae[e + 3] = 0.f;
    }
    for (int k0 = begin; k0 < end; k0 += 32) {
      QTile<F> tile;
      load_qtile_staged<F>(tile, col_next, c_next, Y, lane, skip ? 0 : min(32, end - k0));
      const bool last = k0 + 32 >= end;  // wave-uniform: the next tile belongs to the next segment
      fetch_entries(indices, data, lane, last ? b1 : k0 + 32, last ? e1 : end, col_next, c_next);
      qtile_apply<F, FIRST>(tile, ve, ae);
    }
    if (!skip) {
      float ac[FC];
      reduce_expanded<F>(ae, ac);
      store_compact<F>(partial + (size_t)s * LD, lane, ac);
    }
  }
}

// PHASE 0: r = sum(partials) - A0 x ; p = r ; rsold = r.r ; done = rsold < 1e-20
// PHASE 1: Ap = sum(partials) + A0 p ; alpha ; x += alpha p ; r -= alpha Ap ; rsnew ; done |= rsnew < 1e-20 ; p = r + beta p
// row of X in the factor storage type <-> the lane-contiguous fp32 registers of the generic kernels (VEC layouts only for fp16)
template <int VPL, bool VEC> __device__ __forceinline__ void load_xrow(const float *row, int f, int lane, float (&x)[VPL]) {
  load_row<VPL, VEC>(row, f, lane, x);
}
template <int VPL, bool VEC> __device__ __forceinline__ void load_xrow(const __half *row, int f, int lane, float (&x)[VPL]) {
  static_assert(VEC && VPL <= 2, "fp16 rows: f = 64 / 128 only");
  if constexpr (VPL == 2) {
    const float2 t = load2(row + 2 * lane);
    x[0] = t.x, x[1] = t.y;
  } else {
    x[0] = load1(row + lane);
  }
}
template <typename T> __device__ __forceinline__ void store_x(T *p, float v) { store1(p, v); }

// ROWBLOCK: one WORKGROUP per row instead of one wavefront -- for plans of few, very long rows (the rows beyond the cluster
// kernels' reach: 98 rows with up to 288 + segment partials each at C3), where a single wavefront walking a row's partials
// is the launch's critical path.  The wavefronts sum strided subsets of the partials, wavefront 0 adds the subset sums
// in a fixed order and carries on alone.
template <int VPL, bool VEC, int BLOCK, bool A_LDS, int PHASE, bool ROWBLOCK, typename T>
__global__ __launch_bounds__(BLOCK) void cg_long_combine_kernel(const LongPlanDev plan, T *__restrict__ X,
                                                                const float *__restrict__ A0, int f,
                                                                const float *__restrict__ partial, float *__restrict__ rvec,
                                                                float *__restrict__ pvec, float *__restrict__ scal,
                                                                float *__restrict__ xvec) {
  // xvec (fp16 factor storage only): the fp32 iterate of every long row between passes -- X itself would round it to
  // fp16 after every CG step, where the resident kernels (and the reference, als.cu:45-109) round once at the end
  constexpr int LD = 64 * VPL;
  constexpr int WAVES = BLOCK / 64;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const float *Amat = stage_gramian<VPL, VEC, BLOCK, A_LDS>(smem, A0, f);
  const int lda = A_LDS ? LD : f;
  const int vld = VEC ? LD : f;  // logical length for guarded loads from the LD-strided workspaces
  float *red = smem + (A_LDS ? (size_t)f * LD : 0);  // [WAVES][LD], ROWBLOCK only
  for (int li = ROWBLOCK ? (int)blockIdx.x : (int)blockIdx.x * WAVES + wave; li < plan.n_long;
       li += ROWBLOCK ? (int)gridDim.x : (int)gridDim.x * WAVES) {
    if (PHASE == 1 && scal[2 * li + 1] != 0.f) continue;  // ROWBLOCK: the same for the whole workgroup
    T *xrow = X + (size_t)plan.rows[li] * f;
    float acc[VPL];
    const int s0 = plan.row_seg[li], s1 = plan.row_seg[li + 1];
    if constexpr (ROWBLOCK) {
      constexpr int NS = 4;
      float a4[NS][VPL];
#pragma unroll
      for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int v = 0; v < VPL; ++v) a4[i][v] = 0.f;
      int s = s0 + wave;
      for (; s + (NS - 1) * WAVES < s1; s += NS * WAVES) {
        float t[NS][VPL];
#pragma unroll
        for (int i = 0; i < NS; ++i) load_row<VPL, VEC>(partial + (size_t)(s + i * WAVES) * LD, vld, lane, t[i]);
#pragma unroll
        for (int i = 0; i < NS; ++i)
#pragma unroll
          for (int v = 0; v < VPL; ++v) a4[i][v] += t[i][v];
      }
      for (; s < s1; s += WAVES) {
        float t[VPL];
        load_row<VPL, VEC>(partial + (size_t)s * LD, vld, lane, t);
#pragma unroll
        for (int v = 0; v < VPL; ++v) a4[0][v] += t[v];
      }
#pragma unroll
      for (int v = 0; v < VPL; ++v) red[wave * LD + elem<VPL, VEC>(lane, v)] = (a4[0][v] + a4[1][v]) + (a4[2][v] + a4[3][v]);
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
          float t = 0.f;
#pragma unroll
          for (int w = 0; w < WAVES; ++w) t += red[w * LD + elem<VPL, VEC>(lane, v)];
          acc[v] = t;
        }
      }
      __syncthreads();  // `red` is free again
      if (wave != 0) continue;
    } else {  // fixed association: NS interleaved running sums, folded pairwise (a 147 K-nnz row has 288 segment partials;
       // one dependent chain of loads + adds would make that row the launch's critical path)
      constexpr int NS = 8;
      float a8[NS][VPL];
#pragma unroll
      for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int v = 0; v < VPL; ++v) a8[i][v] = 0.f;
      int s = s0;
      for (; s + NS <= s1; s += NS) {
        float t[NS][VPL];
#pragma unroll
        for (int i = 0; i < NS; ++i) load_row<VPL, VEC>(partial + (size_t)(s + i) * LD, vld, lane, t[i]);
#pragma unroll
        for (int i = 0; i < NS; ++i)
#pragma unroll
          for (int v = 0; v < VPL; ++v) a8[i][v] += t[i][v];
      }
      for (int i = 0; s < s1; ++s, ++i) {  // at most NS - 1 left: one more (partial) trip
        float t[VPL];
        load_row<VPL, VEC>(partial + (size_t)s * LD, vld, lane, t);
#pragma unroll
        for (int j = 0; j < NS; ++j)
          if (j == i)
#pragma unroll
            for (int v = 0; v < VPL; ++v) a8[j][v] += t[v];
      }
#pragma unroll
      for (int v = 0; v < VPL; ++v)
        acc[v] = ((a8[0][v] + a8[1][v]) + (a8[2][v] + a8[3][v])) + ((a8[4][v] + a8[5][v]) + (a8[6][v] + a8[7][v]));
    }
    float x[VPL], dense[VPL];
    if (PHASE == 1 && xvec) load_row<VPL, VEC>(xvec + (size_t)li * LD, VEC ? LD : f, lane, x);
    else load_xrow<VPL, VEC>(xrow, f, lane, x);
#pragma unroll
    for (int v = 0; v < VPL; ++v) dense[v] = 0.f;
    if (PHASE == 0) {
      gram_matvec<VPL, VEC>(Amat, lda, lane, x, dense, 0, f);
      float r[VPL];
#pragma unroll
      for (int v = 0; v < VPL; ++v) r[v] = acc[v] - dense[v];
      float rsold = wave_allsum(dot_local<VPL>(r, r));
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        int e = elem<VPL, VEC>(lane, v);
        rvec[(size_t)li * LD + e] = r[v];
        pvec[(size_t)li * LD + e] = r[v];
        if (xvec) xvec[(size_t)li * LD + e] = x[v];
      }
      if (lane == 0) {
        scal[2 * li] = rsold;
        scal[2 * li + 1] = rsold < 1e-20f ? 1.f : 0.f;
      }
    } else {
      float p[VPL], r[VPL];
      load_row<VPL, VEC>(pvec + (size_t)li * LD, vld, lane, p);
      load_row<VPL, VEC>(rvec + (size_t)li * LD, vld, lane, r);
      gram_matvec<VPL, VEC>(Amat, lda, lane, p, dense, 0, f);
      float Ap[VPL];
#pragma unroll
      for (int v = 0; v < VPL; ++v) Ap[v] = dense[v] + acc[v];
      float rsold = scal[2 * li];
      float alpha = rsold / wave_allsum(dot_local<VPL>(p, Ap));
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        x[v] = fmaf(alpha, p[v], x[v]);
        r[v] = fmaf(-alpha, Ap[v], r[v]);
      }
      float rsnew = wave_allsum(dot_local<VPL>(r, r));
      float beta = rsnew / rsold;
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        int e = elem<VPL, VEC>(lane, v);
        if (e < f) store_x(xrow + e, x[v]);
        if (xvec) xvec[(size_t)li * LD + e] = x[v];
        rvec[(size_t)li * LD + e] = r[v];
        pvec[(size_t)li * LD + e] = fmaf(beta, p[v], r[v]);
      }
      if (lane == 0) {
        scal[2 * li] = rsnew;
        if (rsnew < 1e-20f) scal[2 * li + 1] = 1.f;
      }
    }
  }
}

template <typename T>
__global__ void zero_rows_kernel(const int32_t *__restrict__ order, int first, int count, T *__restrict__ X, int f) {
  size_t total = (size_t)count * f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t r = i / f, c = i - r * f;
    store1(X + (size_t)order[first + r] * f + c, 0.f);
  }
}

template <typename T> static void zero_rows_t(const int32_t *order, int first, int count, T *X, int f) {
  if (count <= 0) return;
  IMP_PROF("zero_rows");
  size_t total = (size_t)count * f;
  int grid = (int)std::min<size_t>((total + 255) / 256, (size_t)ctx().num_cus * 8);
  zero_rows_kernel<T><<<grid, 256, 0, stream()>>>(order, first, count, X, f);
  IMP_CHECK_HIP(hipGetLastError());
}
void zero_rows(const int32_t *order, int first, int count, float *X, int f) { zero_rows_t<float>(order, first, count, X, f); }

template <int VPL, bool VEC, bool A_LDS, bool RESIDENT>
static void launch_fused(const imp_csr *C, int first, int count, float *X, const float *Y, const float *A0, int f,
                         int cg_steps, const char *name) {
  if (count <= 0) return;
  constexpr int BLOCK = 512;
  constexpr int LD = 64 * VPL;
  size_t lds = (A_LDS ? (size_t)f * LD : 0) * sizeof(float);
  auto kern = als_cg_kernel<VPL, VEC, BLOCK, A_LDS, RESIDENT>;
  IMP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)std::max<size_t>(lds, 16)));
  int blocks_per_cu = (int)std::max<size_t>(1, std::min<size_t>(2048 / BLOCK, (160 * 1024) / std::max<size_t>(lds, 1)));
  int grid = std::min((count + BLOCK / 64 - 1) / (BLOCK / 64), ctx().num_cus * blocks_per_cu * ctx().oversub);
  IMP_PROF(name);
  kern<<<grid, BLOCK, lds, stream()>>>(C->order.data(), first, count, C->indptr.data(), C->indices.data(), C->data.data(), X, Y,
                                       A0, f, cg_steps);
  IMP_CHECK_HIP(hipGetLastError());
}

template <int VPL, bool VEC, bool A_LDS, typename T>
static void launch_long(const imp_csr *C, const LongPlan &lp, T *X, const T *Y, const float *A0, int f, int cg_steps) {
  const int n_long = lp.n_long, n_seg = lp.n_seg;
  if (n_long <= 0) return;
  constexpr int BLOCK = 512;
  constexpr int LD = 64 * VPL;
  constexpr bool kHalf = !std::is_same<T, float>::value;
  size_t need = ((size_t)n_seg + (kHalf ? 3 : 2) * (size_t)n_long) * LD + 2 * (size_t)n_long;
  auto &ws = ctx().long_ws;
  if (ws.size < need) ws.alloc(need);
  float *partial = ws.data();
  float *rvec = partial + (size_t)n_seg * LD;
  float *pvec = rvec + (size_t)n_long * LD;
  float *scal = pvec + (size_t)n_long * LD;
  float *xvec = kHalf ? scal + 2 * (size_t)n_long : nullptr;
  LongPlanDev plan = lp.dev(C->order.data());

  // few rows (the streamed remainder of the f = 64 / 128 path): one workgroup per row in the combine kernel
  const bool rowblock = n_long <= ctx().num_cus * 2;
  size_t lds = ((A_LDS ? (size_t)f * LD : 0) + (rowblock ? (size_t)(BLOCK / 64) * LD : 0)) * sizeof(float);
  auto comb0 = rowblock ? cg_long_combine_kernel<VPL, VEC, BLOCK, A_LDS, 0, true, T> : cg_long_combine_kernel<VPL, VEC, BLOCK, A_LDS, 0, false, T>;
  auto comb1 = rowblock ? cg_long_combine_kernel<VPL, VEC, BLOCK, A_LDS, 1, true, T> : cg_long_combine_kernel<VPL, VEC, BLOCK, A_LDS, 1, false, T>;
  IMP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(comb0), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)std::max<size_t>(lds, 16)));
  IMP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(comb1), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)std::max<size_t>(lds, 16)));
  int grid_part = std::min(((n_seg + 3) / 4 + 7) / 8 * 8, ctx().num_cus * 8);  // a multiple of 8: blockIdx % 8 = XCD
  int grid_comb = rowblock ? n_long : std::min((n_long + BLOCK / 64 - 1) / (BLOCK / 64), ctx().num_cus * 2);
  {
    IMP_PROF("als_cg_long_partial");
    if constexpr (VEC && (VPL == 1 || VPL == 2))
      cg_long_partial_q_kernel<64 * VPL, true, T>
          <<<grid_part, 256, 0, stream()>>>(plan, C->indices.data(), C->data.data(), X, Y, partial, pvec, scal);
    else if constexpr (std::is_same<T, float>::value)
      cg_long_partial_kernel<VPL, VEC, true>
          <<<grid_part, 256, 0, stream()>>>(plan, C->indices.data(), C->data.data(), X, Y, f, partial, pvec, scal);
    IMP_CHECK_HIP(hipGetLastError());
  }
  {
    IMP_PROF("als_cg_long_combine");
    comb0<<<grid_comb, BLOCK, lds, stream()>>>(plan, X, A0, f, partial, rvec, pvec, scal, xvec);
    IMP_CHECK_HIP(hipGetLastError());
  }
  for (int it = 0; it < cg_steps; ++it) {
    {
      IMP_PROF("als_cg_long_partial");
      if constexpr (VEC && (VPL == 1 || VPL == 2))
        cg_long_partial_q_kernel<64 * VPL, false, T>
            <<<grid_part, 256, 0, stream()>>>(plan, C->indices.data(), C->data.data(), X, Y, partial, pvec, scal);
      else if constexpr (std::is_same<T, float>::value)
        cg_long_partial_kernel<VPL, VEC, false>
            <<<grid_part, 256, 0, stream()>>>(plan, C->indices.data(), C->data.data(), X, Y, f, partial, pvec, scal);
      IMP_CHECK_HIP(hipGetLastError());
    }
    {
      IMP_PROF("als_cg_long_combine");
      comb1<<<grid_comb, BLOCK, lds, stream()>>>(plan, X, A0, f, partial, rvec, pvec, scal, xvec);
      IMP_CHECK_HIP(hipGetLastError());
    }
  }
}

template <int VPL, bool VEC, bool A_LDS, typename T>
static void launch_all(const imp_csr *C, T *X, const T *Y, size_t y_rows, const float *A0, int f, int cg_steps) {
  // schedule classes (imp_csr): 0 long (segment-split), 1..4 mid, 5..6 short, 7 empty
  const int32_t *b = C->bin_start;
  if constexpr (VEC && A_LDS && (VPL == 1 || VPL == 2)) {
    // f = 64 / 128: rows of more than 512 nonzeros through their explicit normal matrix on the matrix cores (als_cg_nm.hip).
    // IMP_NM=0: one streamed pass per CG step instead (A/B, parity; the workgroup clusters of rounds 2-3 are gone)
    if (nm_enabled()) least_squares_cg_nm<T>(C, X, Y, y_rows, A0, f, cg_steps);
    else launch_long<VPL, VEC, A_LDS, T>(C, C->plan_all, X, Y, A0, f, cg_steps);
    least_squares_cg_q<T>(C, X, Y, A0, f, cg_steps);  // quarter-layout register tiles, wave teams (als_cg_q.hip)
  } else if constexpr (std::is_same<T, float>::value) {
    launch_long<VPL, VEC, A_LDS, T>(C, C->plan_all, X, Y, A0, f, cg_steps);
    if constexpr (VEC && VPL == 4 && !A_LDS) {  // f = 256: workgroup-shared gramian
      if (w256_enabled()) {  // round 5: rows of <= 256 nonzeros resident, 16 / WPR rows per workgroup in lock step (als_cg_w256.hip)
        launch_f256(C, b[1], b[2] - b[1], X, Y, A0, cg_steps, "als_cg_mid_rows");
        least_squares_cg_w256(C, X, Y, A0, cg_steps);
      } else {  // IMP_F256_OLD=1: every row streamed (A/B, parity)
        launch_f256(C, b[1], b[7] - b[1], X, Y, A0, cg_steps, "als_cg_mid_rows");
      }
      zero_rows_t<T>(C->order.data(), C->first_empty(), C->n_empty(), X, f);
      return;
    }
    bool resident_ok = false;
    if constexpr (VEC) resident_ok = tile_size<VPL>() >= imp_csr::kShortRow;
    if constexpr (VEC) {
      if (resident_ok) {
        launch_fused<VPL, VEC, A_LDS, false>(C, b[1], b[5] - b[1], X, Y, A0, f, cg_steps, "als_cg_mid_rows");
        launch_fused<VPL, VEC, A_LDS, true>(C, b[5], b[7] - b[5], X, Y, A0, f, cg_steps, "als_cg_short_rows");
      }
    }
    if (!resident_ok) launch_fused<VPL, VEC, A_LDS, false>(C, b[1], b[7] - b[1], X, Y, A0, f, cg_steps, "als_cg_mid_rows");
  }
  zero_rows_t<T>(C->order.data(), C->first_empty(), C->n_empty(), X, f);
}

// fp16 factor storage is handled natively (converted in registers) by the f = 64 / 128 kernels
bool cg_native_half(int f) { return f == 64 || f == 128; }

// ---- other factor counts below 128: zero-padded onto the f = 64 / 128 kernels ------------------------------------------------
// The resident-tile kernels exist for f = 64 and 128.  Any other f < 128 (the reference's CPU default is 100) ran the generic
// lane-strided one-wave-per-row kernel: 23.0 ms per configs[2]-shaped iteration at f = 100 against 4.6 at f = 128, 13-16 ms at
// f = 16 / 32 / 50 against 2.7 at f = 64 (gpurun_out/r3s).  Padding is exact for CG: with the extra columns of X and Y zero and
// the gramian extended by a unit diagonal block, residual, search direction and iterate stay zero in the padded components
// (b = 0, x0 = 0 there), and every dot product only gains exact zeros.  Cost: one padded copy of Y and of the solved rows of X
// in, the rows of X out -- (R_y + 2 R_x)(f + F) 4 bytes per half sweep, ~0.2 ms at configs[2] -- and the workspaces.
__global__ void pad_rows_kernel(const float *__restrict__ src, float *__restrict__ dst, size_t rows, int f, int F,
                                const int *__restrict__ skip = nullptr) {
  if (skip && *skip) return;  // the padded copy is still the one this call needs (pad_check_kernel)
  const size_t n = rows * (size_t)F;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / F;
    const int c = (int)(i - r * F);
    dst[i] = c < f ? src[r * f + c] : 0.f;
  }
}
__global__ void unpad_rows_kernel(const float *__restrict__ src, float *__restrict__ dst, size_t rows, int f, int F) {
  const size_t n = rows * (size_t)f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / f;
    dst[i] = src[r * F + (i - r * f)];
  }
}
// *same = 1 iff the f x f gramian of this call equals, bit for bit, the top-left block of the padded gramian of the previous
// one (single workgroup; the flag starts at 1 and any differing element clears it)
__global__ void pad_check_kernel(const float *__restrict__ gram, const float *__restrict__ padded, int f, int F, int *same) {
  if (threadIdx.x == 0) *same = 1;
  __syncthreads();
  bool differ = false;
  for (int i = threadIdx.x; i < f * f; i += blockDim.x) {
    const int r = i / f, c = i - r * f;
    differ |= __float_as_uint(gram[i]) != __float_as_uint(padded[(size_t)r * F + c]);
  }
  if (differ) *same = 0;
}
__global__ void pad_gram_kernel(const float *__restrict__ src, float *__restrict__ dst, int f, int F) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < F * F; i += gridDim.x * blockDim.x) {
    const int r = i / F, c = i - r * F;
    dst[i] = (r < f && c < f) ? src[r * f + c] : (r == c ? 1.f : 0.f);
  }
}

static void least_squares_cg_padded(const imp_csr *C, imp_matrix *X, const imp_matrix *YtY, const imp_matrix *Y, int cg_steps, int F) {
  const int f = (int)X->cols;
  auto &c = ctx();
  const size_t rx = (size_t)C->rows, ry = Y->rows;
  if (c.pad_x.size < rx * F) c.pad_x.alloc(rx * F);
  if (c.pad_y.size < ry * F) {
    c.pad_y_src = nullptr;  // the old copy goes with its buffer (before the alloc: freeing it reports a write to that memory)
    c.pad_y.alloc(ry * F);
  }
  if (c.pad_gram.size < (size_t)F * F) c.pad_gram.alloc((size_t)F * F);
  auto grid = [&](size_t n) { return (int)std::max<size_t>(1, std::min<size_t>((n + 255) / 256, (size_t)c.num_cus * 16)); };
  {
    IMP_PROF("pad_factors");
    // The padded copy of Y is re-used when this call solves against the SAME matrix under the SAME gramian as the previous
    // one -- the K row chunks of a sharded half sweep (4 redundant copies of a 10 M-row replica otherwise).  Same address, shape
    // and factor counts are checked here; "same contents" is decided on the device, with no host wait, through the gramian:
    // whoever changes Y recomputes YtY (the solve is meaningless otherwise), so a gramian equal bit for bit to the one the copy
    // was made under vouches for it.  The flag is read by the pad kernel itself, which then returns at once.
    const int *skip = nullptr;
    if (ry && c.pad_y_src == Y->data && c.pad_y_rows == ry && c.pad_y_f == f && c.pad_y_F == F) {
      if (c.pad_same.size < 1) c.pad_same.alloc(1);
      pad_check_kernel<<<1, 1024, 0, stream()>>>(YtY->f32(), c.pad_gram.data(), f, F, c.pad_same.data());
      skip = c.pad_same.data();
    }
    if (ry) pad_rows_kernel<<<grid(ry * F), 256, 0, stream()>>>(Y->f32(), c.pad_y.data(), ry, f, F, skip);
    c.pad_y_src = Y->data, c.pad_y_rows = ry, c.pad_y_f = f, c.pad_y_F = F;
    if (rx) pad_rows_kernel<<<grid(rx * F), 256, 0, stream()>>>(X->f32(), c.pad_x.data(), rx, f, F);
    pad_gram_kernel<<<grid((size_t)F * F), 256, 0, stream()>>>(YtY->f32(), c.pad_gram.data(), f, F);
    IMP_CHECK_HIP(hipGetLastError());
  }
  if (F == 64) launch_all<1, true, true, float>(C, c.pad_x.data(), c.pad_y.data(), ry, c.pad_gram.data(), F, cg_steps);
  else if (F == 128) launch_all<2, true, true, float>(C, c.pad_x.data(), c.pad_y.data(), ry, c.pad_gram.data(), F, cg_steps);
  else launch_all<4, true, false, float>(C, c.pad_x.data(), c.pad_y.data(), ry, c.pad_gram.data(), F, cg_steps);
  {
    IMP_PROF("unpad_factors");
    if (rx) unpad_rows_kernel<<<grid(rx * f), 256, 0, stream()>>>(c.pad_x.data(), X->f32(), rx, f, F);
    IMP_CHECK_HIP(hipGetLastError());
  }
}

// The Cholesky entry's use of the same workspaces (als_cholesky.hip: 64 < f < 128 rides the f = 128 path): Y, the solved rows of X and the
// gramian zero-padded to F columns, the gramian with a unit diagonal block -- the padded system is block diagonal, its solution the
// original one followed by zeros.  The CG path's "same Y as last time" shortcut does not survive another user of pad_y.
void cholesky_pad_in(const imp_matrix *X, const imp_matrix *Y, const imp_matrix *YtY, size_t rx, int F) {
  const int f = (int)X->cols;
  auto &c = ctx();
  const size_t ry = Y->rows;
  if (c.pad_x.size < rx * F) c.pad_x.alloc(rx * F);
  c.pad_y_src = nullptr;
  if (c.pad_y.size < ry * F) c.pad_y.alloc(ry * F);
  if (c.pad_gram.size < (size_t)F * F) c.pad_gram.alloc((size_t)F * F);
  auto grid = [&](size_t n) { return (int)std::max<size_t>(1, std::min<size_t>((n + 255) / 256, (size_t)c.num_cus * 16)); };
  IMP_PROF("pad_factors");
  if (ry) pad_rows_kernel<<<grid(ry * F), 256, 0, stream()>>>(Y->f32(), c.pad_y.data(), ry, f, F);
  if (rx) pad_rows_kernel<<<grid(rx * F), 256, 0, stream()>>>(X->f32(), c.pad_x.data(), rx, f, F);
  pad_gram_kernel<<<grid((size_t)F * F), 256, 0, stream()>>>(YtY->f32(), c.pad_gram.data(), f, F);
  IMP_CHECK_HIP(hipGetLastError());
}
void cholesky_pad_out(imp_matrix *X, size_t rx, int F) {
  auto &c = ctx();
  auto grid = [&](size_t n) { return (int)std::max<size_t>(1, std::min<size_t>((n + 255) / 256, (size_t)c.num_cus * 16)); };
  IMP_PROF("unpad_factors");
  if (rx) unpad_rows_kernel<<<grid(rx * X->cols), 256, 0, stream()>>>(c.pad_x.data(), X->f32(), rx, (int)X->cols, F);
  IMP_CHECK_HIP(hipGetLastError());
}

void least_squares_cg(const imp_csr *C, imp_matrix *X, const imp_matrix *YtY, const imp_matrix *Y, int cg_steps) {
  note_device_write(X->data, (size_t)C->rows * X->cols * X->itemsize);  // the rows this call solves
  // one event pair around ALL launches of the half sweep (every row class): what bench.py's whole-step `roofline` is timed on
  IMP_PROF("als_cg_half_sweep");
  const int f = (int)X->cols;
  const float *a0 = YtY->f32();
  if (X->itemsize == 2) {
    if (!cg_native_half(f)) throw std::invalid_argument("least_squares: fp16 factors with this factor count are converted by the caller");
    __half *x = reinterpret_cast<__half *>(X->data);
    const __half *y = reinterpret_cast<const __half *>(Y->data);
    if (f == 64) launch_all<1, true, true, __half>(C, x, y, Y->rows, a0, f, cg_steps);
    else launch_all<2, true, true, __half>(C, x, y, Y->rows, a0, f, cg_steps);
    return;
  }
  float *x = X->f32();
  const float *y = Y->f32();
  // IMP_NO_PAD=1: factor counts other than 64 / 128 on the generic kernels (A/B, parity)
  static const bool no_pad = getenv("IMP_NO_PAD") != nullptr;
  if (!no_pad && f < 256 && f != 64 && f != 128 && f >= 1) {  // 129 .. 255 (the reference publishes f = 192) ride the f = 256 kernels
    least_squares_cg_padded(C, X, YtY, Y, cg_steps, f < 64 ? 64 : (f < 128 ? 128 : 256));
    return;
  }
  if (f == 64) launch_all<1, true, true, float>(C, x, y, Y->rows, a0, f, cg_steps);
  else if (f == 128) launch_all<2, true, true, float>(C, x, y, Y->rows, a0, f, cg_steps);
  else if (f == 256) launch_all<4, true, false, float>(C, x, y, Y->rows, a0, f, cg_steps);
  else if (f < 64) launch_all<1, false, true, float>(C, x, y, Y->rows, a0, f, cg_steps);
  else if (f < 128) launch_all<2, false, true, float>(C, x, y, Y->rows, a0, f, cg_steps);
  else if (f < 192) launch_all<3, false, true, float>(C, x, y, Y->rows, a0, f, cg_steps);
  else if (f < 256) launch_all<4, false, false, float>(C, x, y, Y->rows, a0, f, cg_steps);
  else if (f <= 384) launch_all<6, false, false, float>(C, x, y, Y->rows, a0, f, cg_steps);
  else if (f <= 512) launch_all<8, false, false, float>(C, x, y, Y->rows, a0, f, cg_steps);
  else if (f <= 768) launch_all<12, false, false, float>(C, x, y, Y->rows, a0, f, cg_steps);
  else if (f <= 1024) launch_all<16, false, false, float>(C, x, y, Y->rows, a0, f, cg_steps);  // the reference's limit: one thread per factor, als.cu:177-179
  else throw std::invalid_argument("least_squares: factors must be <= 1024 (as the reference, implicit/gpu/als.cu:177-182)");
}

}  // namespace imp
