// Register-tile building blocks of the generic CG kernels (als_cg.hip; the f = 64 / 128 kernels use als_qtile.h).
#ifndef IMPLICIT_AMD_CSRC_ALS_TILE_H_
#define IMPLICIT_AMD_CSRC_ALS_TILE_H_
#include "wave_ops.h"

namespace imp {

// acc[e] += sum_{j in [j_begin, j_end)} A0[j][e] * vec[j]   (A0 symmetric)
// A0s: LDS image (leading dimension LD = 64*VPL, zero padded) or the global f x f matrix (LD = f).
template <int VPL, bool VEC>
__device__ __forceinline__ void gram_matvec(const float *A0s, int LD, int lane, const float (&vec)[VPL],
                                            float (&acc)[VPL], int j_begin, int j_end) {
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    // factor j = elem(l, v); walk only the lanes l whose j lies in [j_begin, j_end)
    const int l_begin = VEC ? (j_begin - v + VPL - 1) / VPL : max(j_begin - 64 * v, 0);
    const int l_end = VEC ? min(64, (j_end - v + VPL - 1) / VPL) : min(64, j_end - 64 * v);
#pragma unroll 8
    for (int l = l_begin; l < l_end; ++l) {
      const int j = VEC ? l * VPL + v : l + 64 * v;
      float pj = lane_bcast(vec[v], l);
      float row[VPL];
      load_row<VPL, VEC>(A0s + (size_t)j * LD, LD, lane, row);
#pragma unroll
      for (int w = 0; w < VPL; ++w) acc[w] = fmaf(pj, row[w], acc[w]);
    }
  }
}

// w_k of one nonzero:  FIRST: (c > 0 ? c : 0) - (|c| - 1) * d   (_als.pyx:190-201)
//                      else : (|c| - 1) * d                      (_als.pyx:214-222)
template <bool FIRST> __device__ __forceinline__ float nnz_weight(float c, float d) {
  float a = c > 0.f ? c : -c;
  float t = (FIRST && c > 0.f) ? c : 0.f;
  return FIRST ? t - (a - 1.f) * d : (a - 1.f) * d;
}

// ---- tiled pass (vector layouts f = 64, 128, 256) ----------------------------------------------------------
// T gathered rows live in registers.  After the reduce-scatter lane (16-lane row r, slot j) owns the dot
// product of tile entry t = J*r + j, J = T/4.
__device__ __forceinline__ float swap32_sum(float a, float b) {
  // lanes 0-31 get a[l] + a[l+32]; lanes 32-63 get b[l-32] + b[l]
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
__device__ __forceinline__ float swap16_sum(float a, float b) {
  // even 16-lane rows get a[row] + a[row+1]; odd rows get b[row-1] + b[row]
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
__device__ __forceinline__ float row_allsum(float v) {
  v += dpp_mov<0x128>(v);  // row_ror:8
  v += dpp_mov<0x124>(v);  // row_ror:4
  v += dpp_mov<0x122>(v);  // row_ror:2
  v += dpp_mov<0x121>(v);  // row_ror:1
  return v;
}

template <int VPL, int T> struct Tile {
  float y[T][VPL];   // gathered factor rows (entries beyond cnt repeat a valid row or are zero)
  float cm1[T / 4];  // |c| - 1 of the entries this lane's 16-lane row owns after the reduce-scatter (0 if invalid)
  float cpos[T / 4]; // max(c, 0) of the same entries (0 if invalid); only the first pass reads it
  int cnt;           // valid entries (wave-uniform)
};

template <int VPL, int T>
__device__ __forceinline__ void load_tile(Tile<VPL, T> &tile, const int32_t *__restrict__ indices,
                                          const float *__restrict__ data, const float *__restrict__ Y, int f, int lane,
                                          int k0, int end) {
  constexpr int J = T / 4;
  const int cnt = max(0, min(T, end - k0));
  tile.cnt = cnt;
  // lanes beyond cnt repeat the last valid entry: their gathers stay in bounds and their weights are masked
  const int my_idx = cnt > 0 ? indices[k0 + min(lane, cnt - 1)] : 0;
  const int base = k0 + J * (lane >> 4);
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const bool ok = base + j < end;
    const float c = ok ? data[base + j] : 0.f;
    tile.cm1[j] = ok ? fabsf(c) - 1.f : 0.f;  // invalid slots get weight 0 in every pass
    tile.cpos[j] = c > 0.f ? c : 0.f;
  }
  // all column ids first (one wait on the index load), then the gathers back to back in groups of 8
  unsigned col[T];
#pragma unroll
  for (int t = 0; t < T; ++t) col[t] = (unsigned)lane_bcast(my_idx, t);
#pragma unroll
  for (int g = 0; g < T / 8; ++g) {
    if (8 * g < cnt) {  // wave-uniform
#pragma unroll
      for (int t = 8 * g; t < 8 * g + 8; ++t) load_row<VPL, true>(Y + (size_t)col[t] * f, f, lane, tile.y[t]);
    } else {
#pragma unroll
      for (int t = 8 * g; t < 8 * g + 8; ++t)
#pragma unroll
        for (int v = 0; v < VPL; ++v) tile.y[t][v] = 0.f;
    }
  }
}

template <int VPL, int T, bool FIRST>
__device__ __forceinline__ void tile_apply(const Tile<VPL, T> &tile, int lane, int k0, int end,
                                           const float (&vec)[VPL], float (&acc)[VPL]) {
  constexpr int J = T / 4;
  float part[T];
#pragma unroll
  for (int t = 0; t < T; ++t) part[t] = dot_local<VPL>(tile.y[t], vec);
  float h[T / 2];
#pragma unroll
  for (int t = 0; t < T / 2; t += 4) {
    float *lo = &part[t], *hi = &part[t + T / 2];
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %4\n\tv_permlane32_swap_b32 %1, %5\n\t"
                 "v_permlane32_swap_b32 %2, %6\n\tv_permlane32_swap_b32 %3, %7"
                 : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]));
#pragma unroll
    for (int u = 0; u < 4; ++u) h[t + u] = lo[u] + hi[u];
  }
  float q[J];
#pragma unroll
  for (int j = 0; j < J; j += 4) {
    float *lo = &h[j], *hi = &h[j + J];
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %4\n\tv_permlane16_swap_b32 %1, %5\n\t"
                 "v_permlane16_swap_b32 %2, %6\n\tv_permlane16_swap_b32 %3, %7"
                 : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]));
#pragma unroll
    for (int u = 0; u < 4; ++u) q[j + u] = lo[u] + hi[u];
  }
  float w[J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    float d = row_allsum(q[j]);
    // FIRST: c+ - (|c|-1) d   else: (|c|-1) d     (_als.pyx:190-201, 214-222); same operations as nnz_weight
    w[j] = FIRST ? tile.cpos[j] - tile.cm1[j] * d : tile.cm1[j] * d;
  }
#pragma unroll
  for (int g = 0; g < T / 8; ++g) {
    if (8 * g < tile.cnt) {  // wave-uniform; entries past cnt inside the group carry weight 0
#pragma unroll
      for (int t = 8 * g; t < 8 * g + 8; ++t) {
        float wt = lane_bcast(w[t % J], 16 * (t / J));
#pragma unroll
        for (int v = 0; v < VPL; ++v) acc[v] = fmaf(wt, tile.y[t][v], acc[v]);
      }
    }
  }
}

// acc += sum over nnz [begin, end) streamed through register tiles
template <int VPL, int T, bool FIRST>
__device__ __forceinline__ void sparse_pass_tiled(const int32_t *__restrict__ indices, const float *__restrict__ data,
                                                  const float *__restrict__ Y, int f, int lane, int begin, int end,
                                                  const float (&vec)[VPL], float (&acc)[VPL]) {
  for (int k0 = begin; k0 < end; k0 += T) {
    Tile<VPL, T> tile;
    load_tile<VPL, T>(tile, indices, data, Y, f, lane, k0, end);
    tile_apply<VPL, T, FIRST>(tile, lane, k0, end, vec, acc);
  }
}

template <int VPL> constexpr int tile_size() { return VPL <= 2 ? 32 : 16; }

}  // namespace imp
#endif  // IMPLICIT_AMD_CSRC_ALS_TILE_H_
