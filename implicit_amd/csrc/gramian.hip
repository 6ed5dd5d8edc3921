// K3: gramian  YtY = Y^T Y + reg * I   (f x f fp32, Y is N x f row-major fp32)
//
// Replaces LeastSquaresSolver::calculate_yty (implicit/gpu/als.cu:122-152: cublasSgemm + the
// l2_regularize_kernel); oracle: np.dot(Y.T, Y) at implicit/cpu/_als.pyx:70,164.
//
// MFMA-bound (2 N f^2 flop over 4 N f bytes).  Uses the exact-fp32 matrix instruction
// v_mfma_f32_32x32x2_f32: for a pair of rows (k = 2) lane l feeds A[i][k] = Y[r0+k][32 ti + i] and
// B[k][j] = Y[r0+k][32 tj + j] straight from global memory (both operands are 128-byte coalesced
// segments of the same two rows -- no LDS staging needed), accumulating a 32x32 tile of Y^T Y.
// The row range is split over grid.x (split-K); partial tiles go to a workspace and a second
// kernel sums them in a FIXED order and adds reg on the diagonal, so the result is deterministic.
#include <hip/hip_fp16.h>

#include "common.h"

namespace imp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// TJ = number of 32-wide column tiles this block covers (<= 8); tile rows: one per wave.
template <int TJ, typename T>
__global__ __launch_bounds__(256) void gramian_partial_kernel(const T *__restrict__ Y, long n_rows, int f,
                                                              long rows_per_chunk, float *__restrict__ ws) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int ti = blockIdx.y * 4 + wave;       // tile row of this wave
  const int tj0 = blockIdx.z * 8;             // first tile column of this block
  const int n_tiles = (f + 31) / 32;
  const int i_col = 32 * ti + (lane & 31);
  const int khalf = lane >> 5;
  const long r_begin = (long)blockIdx.x * rows_per_chunk;
  const long r_end = min(n_rows, r_begin + rows_per_chunk);

  f32x16 acc[TJ];
#pragma unroll
  for (int t = 0; t < TJ; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  if (ti < n_tiles) {
    // 4 k-steps (8 rows) per trip, two register sets: the 4 (1 + TJ) loads of trip t + 1 are in flight during the MFMAs
    // of trip t.  Everything is branch-free -- out-of-range rows / columns load a clamped address and are zeroed by an
    // AND with an all-ones / all-zeros word, applied only when the operand is consumed: a select is turned back into a
    // guarded load (and branches make the compiler drain vmcnt to 0 at every join), an AND right after the load would
    // wait for the load it is supposed to hide.
    const int ca = min(i_col, f - 1), mask_a = -(int)(i_col < f);
    int cb[TJ], mask_b[TJ];
#pragma unroll
    for (int t = 0; t < TJ; ++t) {
      const int c = 32 * (tj0 + t) + (lane & 31);
      mask_b[t] = -(int)(c < f);
      cb[t] = min(c, f - 1);
    }
    float a[2][4], b[2][4][TJ];
    int mask_r[2][4];
    auto fetch = [&](int buf, long r0) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const long r = r0 + 2 * s + khalf;
        mask_r[buf][s] = -(int)(r < r_end);
        const T *row = Y + min(r, n_rows - 1) * (long)f;
        a[buf][s] = (float)row[ca];  // fp16 storage converts here; the products are fp32
#pragma unroll
        for (int t = 0; t < TJ; ++t) b[buf][s][t] = (float)row[cb[t]];
      }
    };
    auto masked = [](float v, int m) { return __int_as_float(__float_as_int(v) & m); };
    auto multiply = [&](int buf) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float av = masked(a[buf][s], mask_r[buf][s] & mask_a);
#pragma unroll
        for (int t = 0; t < TJ; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, masked(b[buf][s][t], mask_r[buf][s] & mask_b[t]), acc[t], 0, 0, 0);
      }
    };
    fetch(0, r_begin);
    for (long r0 = r_begin; r0 < r_end; r0 += 16) {
      // the scheduling fences keep the machine scheduler from sinking the loads back down to their uses
      fetch(1, r0 + 8);  // rows past r_end read the last row and are masked to zero
      __builtin_amdgcn_sched_barrier(0);
      multiply(0);
      __builtin_amdgcn_sched_barrier(0);
      fetch(0, r0 + 16);
      __builtin_amdgcn_sched_barrier(0);
      multiply(1);
      __builtin_amdgcn_sched_barrier(0);
    }
    // C/D layout of the 32x32 tile: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
    float *out = ws + (size_t)blockIdx.x * f * f;
#pragma unroll
    for (int t = 0; t < TJ; ++t) {
      int c = 32 * (tj0 + t) + (lane & 31);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        int rr = 32 * ti + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (rr < f && c < f) out[(size_t)rr * f + c] = acc[t][e];
      }
    }
  }
}

// out = sum over chunks (fixed order) + reg on the diagonal.  One wavefront per 64 output elements per
// chunk slice would be overkill: each thread owns one element and walks the chunks with 4 independent
// accumulators (fixed association -> deterministic).
__global__ void gramian_reduce_kernel(const float *__restrict__ ws, int chunks, int f, float reg, float *__restrict__ out) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= f * f) return;
  const size_t stride = (size_t)f * f;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int c = 0;
  for (; c + 4 <= chunks; c += 4) {
    s0 += ws[(size_t)c * stride + idx];
    s1 += ws[(size_t)(c + 1) * stride + idx];
    s2 += ws[(size_t)(c + 2) * stride + idx];
    s3 += ws[(size_t)(c + 3) * stride + idx];
  }
  for (; c < chunks; ++c) s0 += ws[(size_t)c * stride + idx];
  float s = (s0 + s1) + (s2 + s3);
  int r = idx / f, col = idx - r * f;
  if (r == col) s += reg;
  out[idx] = s;
}

// out (f x f) = Y^T Y + reg I over rows [0, n_rows) of Y
template <typename T> static void gramian_t(const T *Y, long n_rows, int f, float reg, float *out) {
  const int n_tiles = (f + 31) / 32;
  const int gy = (n_tiles + 3) / 4, gz = (n_tiles + 7) / 8;
  long target_chunks = std::max(1, ctx().num_cus * 2 / (gy * gz));
  long rows_per_chunk = std::max<long>(256, (n_rows + target_chunks - 1) / target_chunks);
  rows_per_chunk = (rows_per_chunk + 7) / 8 * 8;
  // no rows (an empty Matrix, or an empty shard of the multi-GPU driver): the sum over zero chunks, i.e. reg * I -- what
  // the reference's GEMM + l2_regularize pair leaves (als.cu:122-152)
  int chunks = n_rows <= 0 ? 0 : (int)std::max<long>(1, (n_rows + rows_per_chunk - 1) / rows_per_chunk);
  size_t need = (size_t)std::max(chunks, 1) * f * f;
  auto &wsbuf = ctx().gram_ws;
  if (wsbuf.size < need) wsbuf.alloc(need);
  dim3 grid(std::max(chunks, 1), gy, gz);
  if (chunks > 0) {
    IMP_PROF("gramian_partial");
    int tj = std::min(n_tiles, 8);
    float *ws = wsbuf.data();
    switch (tj) {
      case 1: gramian_partial_kernel<1, T><<<grid, 256, 0, stream()>>>(Y, n_rows, f, rows_per_chunk, ws); break;
      case 2: gramian_partial_kernel<2, T><<<grid, 256, 0, stream()>>>(Y, n_rows, f, rows_per_chunk, ws); break;
      case 3: gramian_partial_kernel<3, T><<<grid, 256, 0, stream()>>>(Y, n_rows, f, rows_per_chunk, ws); break;
      case 4: gramian_partial_kernel<4, T><<<grid, 256, 0, stream()>>>(Y, n_rows, f, rows_per_chunk, ws); break;
      case 5: gramian_partial_kernel<5, T><<<grid, 256, 0, stream()>>>(Y, n_rows, f, rows_per_chunk, ws); break;
      case 6: gramian_partial_kernel<6, T><<<grid, 256, 0, stream()>>>(Y, n_rows, f, rows_per_chunk, ws); break;
      case 7: gramian_partial_kernel<7, T><<<grid, 256, 0, stream()>>>(Y, n_rows, f, rows_per_chunk, ws); break;
      default: gramian_partial_kernel<8, T><<<grid, 256, 0, stream()>>>(Y, n_rows, f, rows_per_chunk, ws); break;
    }
    IMP_CHECK_HIP(hipGetLastError());
  }
  {
    IMP_PROF("gramian_reduce");
    gramian_reduce_kernel<<<(f * f + 63) / 64, 64, 0, stream()>>>(wsbuf.data(), chunks, f, reg, out);
    IMP_CHECK_HIP(hipGetLastError());
  }
}

void gramian(const float *Y, long n_rows, int f, float reg, float *out) { gramian_t<float>(Y, n_rows, f, reg, out); }
void gramian_half(const void *Y, long n_rows, int f, float reg, float *out) {
  gramian_t<__half>(reinterpret_cast<const __half *>(Y), n_rows, f, reg, out);
}

}  // namespace imp
