// K3: gramian  YtY = Y^T Y + reg * I   (f x f fp32, Y is N x f row-major fp32 or fp16)
//
// Replaces LeastSquaresSolver::calculate_yty (implicit/gpu/als.cu:122-152: cublasSgemm + the
// l2_regularize_kernel); oracle: np.dot(Y.T, Y) at implicit/cpu/_als.pyx:70,164.
//
// MFMA-bound (2 N f^2 flop over 4 N f bytes: 64 flop per byte at f = 128 against ~20 at the fp32 matrix peak).  Uses the
// exact-fp32 matrix instruction v_mfma_f32_32x32x2_f32: for a pair of rows (k = 2) lane l feeds A[i][k] = Y[r0+k][32 ti + i]
// and B[k][j] = Y[r0+k][32 tj + j] straight from global memory (both operands are 128-byte coalesced segments of the same
// two rows -- no LDS staging needed), accumulating a 32x32 tile of Y^T Y.
//
// The product is symmetric, so only the tile pairs ti <= tj are computed (10 of 16 at f = 128) and the reduce kernel
// mirrors them.  The pairs are dealt round-robin to the 4 wavefronts of a workgroup (3, 3, 2, 2 at f = 128); wavefront w
// sits on SIMD w, whose matrix pipe it shares with the same-numbered wavefronts of the other resident workgroups, so the
// deal is rotated by two places in every other workgroup (by index parity and by the bit that separates the workgroups a
// CU receives when blocks are dealt round-robin over 256 CUs) and each SIMD sees the average load.  The row range is split
// over grid.x (split-K, 4 workgroups per CU: with one trip of look-ahead per wavefront it takes 4 wavefronts per SIMD to
// cover the load latency); partial tiles go to a workspace and a second kernel sums them in a FIXED order and adds reg on
// the diagonal, so the result is deterministic (and exactly symmetric).
#include <hip/hip_fp16.h>

#include <type_traits>

#include "common.h"

namespace imp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {
constexpr int kMaxPairsPerWave = 3;

// NV = tile pairs of this wavefront (compile-time: no branches inside the row loop)
template <int NV, typename T>
__device__ __forceinline__ void gramian_wave(const T *__restrict__ Y, long n_rows, int f, long r_begin, long r_end,
                                             const int (&ti)[kMaxPairsPerWave], const int (&tj)[kMaxPairsPerWave],
                                             const int (&pair)[kMaxPairsPerWave], int n_pairs, float *__restrict__ out, int lane) {
  const int khalf = lane >> 5;
  f32x16 acc[NV];
#pragma unroll
  for (int t = 0; t < NV; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  // 4 k-steps (8 rows) per trip, two register sets: the 8 NV loads of trip t + 1 are in flight during the MFMAs of
  // trip t.  Everything is branch-free -- out-of-range rows / columns load a clamped address and are zeroed by an AND with
  // an all-ones / all-zeros word, applied only when the operand is consumed: a select is turned back into a guarded load
  // (and branches make the compiler drain vmcnt to 0 at every join), an AND right after the load would wait for the load
  // it is supposed to hide.
  int ca[NV], cb[NV], mask_a[NV], mask_b[NV];
#pragma unroll
  for (int t = 0; t < NV; ++t) {
    const int c0 = 32 * ti[t] + (lane & 31), c1 = 32 * tj[t] + (lane & 31);
    mask_a[t] = -(int)(c0 < f), mask_b[t] = -(int)(c1 < f);
    ca[t] = min(c0, f - 1), cb[t] = min(c1, f - 1);
  }
  float a[2][4][NV], b[2][4][NV];
  int mask_r[2][4];
  auto fetch = [&](int buf, long r0) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const long r = r0 + 2 * s + khalf;
      mask_r[buf][s] = -(int)(r < r_end);
      const T *row = Y + min(r, n_rows - 1) * (long)f;
#pragma unroll
      for (int t = 0; t < NV; ++t) {
        a[buf][s][t] = (float)row[ca[t]];  // fp16 storage converts here; the products are fp32
        b[buf][s][t] = (float)row[cb[t]];
      }
    }
  };
  auto masked = [](float v, int m) { return __int_as_float(__float_as_int(v) & m); };
  auto multiply = [&](int buf) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int t = 0; t < NV; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(masked(a[buf][s][t], mask_r[buf][s] & mask_a[t]),
                                                      masked(b[buf][s][t], mask_r[buf][s] & mask_b[t]), acc[t], 0, 0, 0);
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
  // C/D layout of the 32x32 tile: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5); tiles are stored whole
  // (32 x 32, tile-contiguous, padding included) at [pair]
#pragma unroll
  for (int t = 0; t < NV; ++t) {
    float *tile = out + ((size_t)pair[t]) * 1024;
    (void)n_pairs;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int rr = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
      tile[rr * 32 + (lane & 31)] = acc[t][e];
    }
  }
}
}  // namespace

// workspace: [chunk][pair][32 x 32]; pairs = upper-triangular tile pairs in row-major order
template <int TPW, typename T>
__global__ __launch_bounds__(256) void gramian_partial_kernel(const T *__restrict__ Y, long n_rows, int f,
                                                              long rows_per_chunk, float *__restrict__ ws) {
  static_assert(TPW <= kMaxPairsPerWave, "pairs per wave");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int n_tiles = (f + 31) / 32, n_pairs = n_tiles * (n_tiles + 1) / 2;
  const int wrot = (wave + 2 * (int)((blockIdx.x ^ (blockIdx.x >> 8)) & 1)) & 3;  // see the header comment
  int ti[kMaxPairsPerWave] = {0, 0, 0}, tj[kMaxPairsPerWave] = {0, 0, 0}, pair[kMaxPairsPerWave] = {0, 0, 0};
  int nv = 0;
#pragma unroll
  for (int s = 0; s < TPW; ++s) {
    const int p = ((int)blockIdx.y * TPW + s) * 4 + wrot;
    if (p < n_pairs) {  // valid slots are a prefix (p grows with s)
      int row = 0, left = p;
      while (left >= n_tiles - row) left -= n_tiles - row, ++row;  // row-major upper triangle
      ti[s] = row, tj[s] = row + left, pair[s] = p;
      nv = s + 1;
    }
  }
  const long r_begin = (long)blockIdx.x * rows_per_chunk;
  const long r_end = min(n_rows, r_begin + rows_per_chunk);
  float *out = ws + (size_t)blockIdx.x * n_pairs * 1024;
  if (nv == 3) {
    if constexpr (TPW >= 3) gramian_wave<3, T>(Y, n_rows, f, r_begin, r_end, ti, tj, pair, n_pairs, out, lane);
  } else if (nv == 2) {
    if constexpr (TPW >= 2) gramian_wave<2, T>(Y, n_rows, f, r_begin, r_end, ti, tj, pair, n_pairs, out, lane);
  } else if (nv == 1) {
    gramian_wave<1, T>(Y, n_rows, f, r_begin, r_end, ti, tj, pair, n_pairs, out, lane);
  }
}

// ---- f = 64 / 128: one vector load per row feeds every tile ---------------------------------------------------------
// With f = 32 VW (VW = 2, 4) lane i loads the VW consecutive factors VW i .. VW i + VW - 1 of a row in ONE load; taking
// "tile c" to be the columns {VW i + c} (a column permutation of Y, undone when the result is written), that register
// vector holds the lane's operand for all VW tiles, as A and as B alike: 1 load per k-step instead of 2 per tile pair,
// a third of the staging registers -- which buys a third buffer, i.e. two trips (8 k-steps) of look-ahead.
template <int N, int I = 0, typename Fn> __device__ __forceinline__ void static_for(Fn &&fn) {
  if constexpr (I < N) {
    fn(std::integral_constant<int, I>{});
    static_for<N, I + 1>(fn);
  }
}
constexpr int tri_row(int n, int p) { int row = 0; while (p >= n - row) p -= n - row, ++row; return row; }
constexpr int tri_col(int n, int p) { int row = 0; while (p >= n - row) p -= n - row, ++row; return row + p; }

template <typename T, int VW> struct FactorVec;
template <> struct FactorVec<float, 4> { using type = float4; };
template <> struct FactorVec<float, 2> { using type = float2; };
template <> struct FactorVec<__half, 4> { using type = uint2; };
template <> struct FactorVec<__half, 2> { using type = unsigned; };

template <int VW> __device__ __forceinline__ void unpack(const float4 &v, float (&o)[VW]) { o[0] = v.x, o[1] = v.y, o[2] = v.z, o[3] = v.w; }
template <int VW> __device__ __forceinline__ void unpack(const float2 &v, float (&o)[VW]) { o[0] = v.x, o[1] = v.y; }
template <int VW> __device__ __forceinline__ void unpack(const uint2 &v, float (&o)[VW]) {
  const float2 a = __half22float2(*reinterpret_cast<const __half2 *>(&v.x)), b = __half22float2(*reinterpret_cast<const __half2 *>(&v.y));
  o[0] = a.x, o[1] = a.y, o[2] = b.x, o[3] = b.y;
}
template <int VW> __device__ __forceinline__ void unpack(const unsigned &v, float (&o)[VW]) {
  const float2 a = __half22float2(*reinterpret_cast<const __half2 *>(&v));
  o[0] = a.x, o[1] = a.y;
}

// WROT = this wavefront's place in the round-robin deal: pairs WROT, WROT + 4, ... of the VW (VW + 1) / 2
template <int VW, int WROT, typename T>
__device__ __forceinline__ void gramian_wave_vec(const T *__restrict__ Y, long n_rows, long r_begin, long r_end,
                                                 float *__restrict__ out, int lane) {
  constexpr int F = 32 * VW, NP = VW * (VW + 1) / 2, NV = (NP - WROT + 3) / 4;
  using Vec = typename FactorVec<T, VW>::type;
  if constexpr (NV > 0) {
    const int khalf = lane >> 5;
    f32x16 acc[NV];
#pragma unroll
    for (int t = 0; t < NV; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    Vec v[3][4];
    // address = wave-uniform row base (scalar registers) + one loop-invariant 32-bit lane offset: lanes 0..31 read the
    // even row of the pair, lanes 32..63 the odd one.  The pair is clamped into the matrix as a whole; at the very last
    // row of an odd-length matrix both halves read that row (the odd half is masked out below)
    const unsigned off_even = (unsigned)(VW * (lane & 31) * sizeof(T));
    const unsigned off_pair = off_even + (unsigned)(khalf * F * sizeof(T));
    auto fetch = [&](int buf, long r0) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const long rb = min(r0 + 2 * s, n_rows - 1);  // wave-uniform
        const unsigned off = rb + 1 < n_rows ? off_pair : off_even;
        v[buf][s] = *reinterpret_cast<const Vec *>(reinterpret_cast<const char *>(Y + rb * (long)F) + off);
      }
    };
    // rows past r_end (only in the last trips of a chunk) read a clamped address and are zeroed when consumed (TAIL);
    // the tile pair of accumulator t is a compile-time constant (static_for: the operands are picked by register name)
    auto multiply = [&](auto tail_tag, int buf, long r0) {
      constexpr bool TAIL = decltype(tail_tag)::value;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float o[VW];
        unpack<VW>(v[buf][s], o);
        if constexpr (TAIL) {
          const int m = -(int)(r0 + 2 * s + khalf < r_end);
#pragma unroll
          for (int c = 0; c < VW; ++c) o[c] = __int_as_float(__float_as_int(o[c]) & m);
        }
        static_for<NV>([&](auto tc) {
          constexpr int t = decltype(tc)::value, ci = tri_row(VW, WROT + 4 * t), cj = tri_col(VW, WROT + 4 * t);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(o[ci], o[cj], acc[t], 0, 0, 0);
        });
      }
    };
    fetch(0, r_begin);
    fetch(1, r_begin + 8);
    long r0 = r_begin;
    for (; r0 + 24 <= r_end; r0 += 24) {
      // the scheduling fences keep the machine scheduler from sinking the loads back down to their uses
      fetch(2, r0 + 16);
      __builtin_amdgcn_sched_barrier(0);
      multiply(std::false_type{}, 0, r0);
      __builtin_amdgcn_sched_barrier(0);
      fetch(0, r0 + 24);
      __builtin_amdgcn_sched_barrier(0);
      multiply(std::false_type{}, 1, r0 + 8);
      __builtin_amdgcn_sched_barrier(0);
      fetch(1, r0 + 32);
      __builtin_amdgcn_sched_barrier(0);
      multiply(std::false_type{}, 2, r0 + 16);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (r0 < r_end) {  // at most 23 rows left: buffers 0 and 1 hold the trips at r0 and r0 + 8
      fetch(2, r0 + 16);
      multiply(std::true_type{}, 0, r0);
      multiply(std::true_type{}, 1, r0 + 8);
      multiply(std::true_type{}, 2, r0 + 16);
    }
#pragma unroll
    for (int t = 0; t < NV; ++t) {
      float *tile = out + (size_t)(WROT + 4 * t) * 1024;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int rr = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        tile[rr * 32 + (lane & 31)] = acc[t][e];
      }
    }
  }
}

template <int VW, typename T>
__global__ __launch_bounds__(256) void gramian_partial_vec_kernel(const T *__restrict__ Y, long n_rows, long rows_per_chunk,
                                                                  float *__restrict__ ws) {
  constexpr int NP = VW * (VW + 1) / 2;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wrot = (wave + 2 * (int)((blockIdx.x ^ (blockIdx.x >> 8)) & 1)) & 3;  // see the header comment
  const long r_begin = (long)blockIdx.x * rows_per_chunk;
  const long r_end = min(n_rows, r_begin + rows_per_chunk);
  float *out = ws + (size_t)blockIdx.x * NP * 1024;
  if (wrot == 0) gramian_wave_vec<VW, 0, T>(Y, n_rows, r_begin, r_end, out, lane);
  else if (wrot == 1) gramian_wave_vec<VW, 1, T>(Y, n_rows, r_begin, r_end, out, lane);
  else if (wrot == 2) gramian_wave_vec<VW, 2, T>(Y, n_rows, r_begin, r_end, out, lane);
  else gramian_wave_vec<VW, 3, T>(Y, n_rows, r_begin, r_end, out, lane);
}

// (The split-bf16 form of the f = 128 gramian -- three bf16 terms per operand on v_mfma_f32_32x32x16_bf16, round 5: the launch
// 12-20 % shorter, every CG kernel behind it 4-8 % slower on some boxes -- was opt-in and is removed; HISTORY.md section 4.3.)


// out = sum over chunks (fixed order) + reg on the diagonal, mirrored.  Block = 64 elements of one tile pair x 16 chunk
// groups: group g walks chunks g, g + 16, ... with 4 independent accumulators, the 16 group sums are then added in a fixed
// order -- deterministic, and enough loads in flight to stream the workspace.
// vw > 1: the tiles are those of the vector path -- tile c = columns {vw i + c}, one 32 vw-column group.
__global__ __launch_bounds__(1024) void gramian_reduce_kernel(const float *__restrict__ ws, int chunks, int f, float reg,
                                                              float *__restrict__ out, int vw) {
  __shared__ float part[16][64];
  const int n_tiles = vw > 1 ? vw : (f + 31) / 32, n_pairs = n_tiles * (n_tiles + 1) / 2;
  const int e = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int p = blockIdx.x / 16, idx = (blockIdx.x % 16) * 64 + e;  // element idx of tile pair p
  const size_t stride = (size_t)n_pairs * 1024;
  const float *src = ws + (size_t)p * 1024 + idx;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int c = g;
  for (; c + 48 < chunks; c += 64) {
    s0 += src[(size_t)c * stride];
    s1 += src[(size_t)(c + 16) * stride];
    s2 += src[(size_t)(c + 32) * stride];
    s3 += src[(size_t)(c + 48) * stride];
  }
  for (; c < chunks; c += 16) s0 += src[(size_t)c * stride];
  part[g][e] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g == 0) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += part[k][e];
    int row = 0, left = p;
    while (left >= n_tiles - row) left -= n_tiles - row, ++row;
    const int r = vw > 1 ? vw * (idx / 32) + row : 32 * row + idx / 32;
    const int col = vw > 1 ? vw * (idx & 31) + row + left : 32 * (row + left) + (idx & 31);
    // off-diagonal tile pairs are mirrored; the diagonal ones are computed whole, but only their upper triangle is used (the
    // split-bf16 partials are not bitwise symmetric; the fp32 ones are, and lose nothing)
    if (r < f && col < f && (left != 0 || r <= col)) {
      if (r == col) s += reg;
      out[(size_t)r * f + col] = s;
      if (r != col) out[(size_t)col * f + r] = s;
    }
  }
}

// out (f x f) = Y^T Y + reg I over rows [0, n_rows) of Y
template <typename T> static void gramian_t(const T *Y, long n_rows, int f, float reg, float *out) {
  const int vw = f == 128 ? 4 : (f == 64 ? 2 : 1);
  const int n_tiles = vw > 1 ? vw : (f + 31) / 32, n_pairs = n_tiles * (n_tiles + 1) / 2;
  const int tpw = std::min(kMaxPairsPerWave, (n_pairs + 3) / 4);
  const int gy = vw > 1 ? 1 : (n_pairs + 4 * tpw - 1) / (4 * tpw);
  long target_chunks = std::max(1, ctx().num_cus * 4 / gy);
  long rows_per_chunk = std::max<long>(256, (n_rows + target_chunks - 1) / target_chunks);
  rows_per_chunk = vw > 1 ? (rows_per_chunk + 23) / 24 * 24 : (rows_per_chunk + 15) / 16 * 16;  // whole trips
  // no rows (an empty Matrix, or an empty shard of the multi-GPU driver): the sum over zero chunks, i.e. reg * I -- what
  // the reference's GEMM + l2_regularize pair leaves (als.cu:122-152)
  int chunks = n_rows <= 0 ? 0 : (int)std::max<long>(1, (n_rows + rows_per_chunk - 1) / rows_per_chunk);
  size_t need = (size_t)std::max(chunks, 1) * n_pairs * 1024;
  auto &wsbuf = ctx().gram_ws;
  if (wsbuf.size < need) wsbuf.alloc(need);
  if (chunks > 0) {
    IMP_PROF("gramian_partial");
    dim3 grid(chunks, gy, 1);
    float *ws = wsbuf.data();
    if (vw == 4) gramian_partial_vec_kernel<4, T><<<grid, 256, 0, stream()>>>(Y, n_rows, rows_per_chunk, ws);
    else if (vw == 2) gramian_partial_vec_kernel<2, T><<<grid, 256, 0, stream()>>>(Y, n_rows, rows_per_chunk, ws);
    else if (tpw == 1) gramian_partial_kernel<1, T><<<grid, 256, 0, stream()>>>(Y, n_rows, f, rows_per_chunk, ws);
    else if (tpw == 2) gramian_partial_kernel<2, T><<<grid, 256, 0, stream()>>>(Y, n_rows, f, rows_per_chunk, ws);
    else gramian_partial_kernel<3, T><<<grid, 256, 0, stream()>>>(Y, n_rows, f, rows_per_chunk, ws);
    IMP_CHECK_HIP(hipGetLastError());
  }
  {
    IMP_PROF("gramian_reduce");
    gramian_reduce_kernel<<<n_pairs * 16, 1024, 0, stream()>>>(wsbuf.data(), chunks, f, reg, out, vw);
    IMP_CHECK_HIP(hipGetLastError());
  }
}

void gramian(const float *Y, long n_rows, int f, float reg, float *out) { gramian_t<float>(Y, n_rows, f, reg, out); }
void gramian_half(const void *Y, long n_rows, int f, float reg, float *out) {
  gramian_t<__half>(reinterpret_cast<const __half *>(Y), n_rows, f, reg, out);
}

}  // namespace imp
