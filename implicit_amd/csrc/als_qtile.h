// "Quarter layout" register tiles for the CG kernels (f = 64, 128).
//
// A wavefront is split into its four 16-lane DPP rows ("groups").  Group g holds tile entries t = 4 q + g
// (q = 0..7, 32 entries per wave); lane (g, m) keeps FE = f/16 factors (float4 pieces 4 m + 64 b) of each of
// its 8 entries (64 VGPRs at f = 128, the same budget as the lane-owns-2-factors tile of als_tile.h).  What changes is
// the cross-lane work per CG pass:
//   * a dot product is FE FMAs per lane + a reduction over the 16 lanes of ONE DPP row (4 row_ror adds) -- no
//     permlane swaps, and the 4 groups reduce 4 different entries at once;
//   * every lane of a group ends with the dot (hence the weight) of its own entries, so the axpy needs no
//     v_readlane broadcast;
//   * the accumulator (FE factors per lane, partial over the group's entries) is brought back to the COMPACT
//     layout -- lane (g, m) owns FC = f/64 of its FE expanded slots, slots FC g + c -- by a two-level reduce-scatter across the
//     groups (FE/2 v_permlane32_swap + FE/4 v_permlane16_swap);  the CG state x, r, p, Ap lives in compact form and
//     only the operand vector of a pass is expanded to FE per lane (3 FC swaps).
// About 150 VALU instructions per full-tile pass instead of ~330, and entries beyond the row's count are skipped
// in whole steps (4 q >= cnt is wave-uniform).
#ifndef IMPLICIT_AMD_CSRC_ALS_QTILE_H_
#define IMPLICIT_AMD_CSRC_ALS_QTILE_H_
#include <hip/hip_fp16.h>

#include "als_tile.h"

namespace imp {

template <int F> struct QL {
  static constexpr int FE = F / 16;  // expanded factors per lane
  static constexpr int FC = F / 64;  // compact factors per lane
  static constexpr int EQ = 8;       // entries per group -> 32 per wave
  static_assert(F == 64 || F == 128, "quarter layout is built for f = 64 and 128");
  // expanded slot e of lane (g, m) is factor 64 (e >> 2) + 4 m + (e & 3): every float4 of a lane is one 16-byte piece of
  // a 256-byte run covered by its 16-lane group (coalesced gathers, conflict-free ds_read_b128 of gramian rows)
  __device__ static __forceinline__ int efactor(int lane, int e) { return 64 * (e >> 2) + 4 * (lane & 15) + (e & 3); }
  // compact slot c of lane (g, m) = expanded slot FC g + c
  __device__ static __forceinline__ int cfactor(int lane, int c) { return efactor(lane, FC * (lane >> 4) + c); }
};

// ---- factor storage: fp32, or fp16 converted in registers at the point of the load / store (the reference's kernels do the
// same per element, implicit/gpu/als.cu:41,55,109 + convert.cuh:7-17); all arithmetic and the CG state stay fp32 ---------
__device__ __forceinline__ float4 load4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float4 load4(const __half *p) {  // 4 consecutive factors = one 8-byte load
  const uint2 raw = *reinterpret_cast<const uint2 *>(p);
  const float2 a = __half22float2(*reinterpret_cast<const __half2 *>(&raw.x));
  const float2 b = __half22float2(*reinterpret_cast<const __half2 *>(&raw.y));
  return make_float4(a.x, a.y, b.x, b.y);
}
__device__ __forceinline__ float2 load2(const float *p) { return *reinterpret_cast<const float2 *>(p); }
__device__ __forceinline__ float2 load2(const __half *p) { return __half22float2(*reinterpret_cast<const __half2 *>(p)); }
__device__ __forceinline__ float load1(const float *p) { return *p; }
__device__ __forceinline__ float load1(const __half *p) { return __half2float(*p); }
__device__ __forceinline__ void store2(float *p, float a, float b) { *reinterpret_cast<float2 *>(p) = make_float2(a, b); }
__device__ __forceinline__ void store2(__half *p, float a, float b) { *reinterpret_cast<__half2 *>(p) = __floats2half2_rn(a, b); }
__device__ __forceinline__ void store1(float *p, float a) { *p = a; }
__device__ __forceinline__ void store1(__half *p, float a) { *p = __float2half_rn(a); }

// compact <-> memory (a row of X, or an LD-strided LDS vector)
template <int F, typename T> __device__ __forceinline__ void load_compact(const T *__restrict__ row, int lane, float (&v)[F / 64]) {
  const T *p = row + QL<F>::cfactor(lane, 0);
  if constexpr (F == 128) {
    const float2 t = load2(p);
    v[0] = t.x, v[1] = t.y;
  } else {
    v[0] = load1(p);
  }
}
template <int F, typename T> __device__ __forceinline__ void store_compact(T *__restrict__ row, int lane, const float (&v)[F / 64]) {
  T *p = row + QL<F>::cfactor(lane, 0);
  if constexpr (F == 128) {
    store2(p, v[0], v[1]);
  } else {
    store1(p, v[0]);
  }
}

// all-gather across the 4 groups: compact (FC per lane) -> expanded (FE per lane)
template <int F> __device__ __forceinline__ void expand_vector(const float (&vc)[F / 64], float (&ve)[F / 16]) {
  constexpr int FC = QL<F>::FC;
  float pair[2 * FC];  // values of the even / odd group of this lane's group pair
#pragma unroll
  for (int c = 0; c < FC; ++c) {
    float a = vc[c], b = vc[c];
    // odd rows of a <-> even rows of b: afterwards a = the even group's value, b = the odd group's, in both rows
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    pair[c] = a;
    pair[FC + c] = b;
  }
#pragma unroll
  for (int i = 0; i < 2 * FC; ++i) {
    float a = pair[i], b = pair[i];
    // lanes 32-63 of a <-> lanes 0-31 of b: a = the value held by groups 0/1, b = the value held by groups 2/3
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    ve[i] = a;
    ve[2 * FC + i] = b;
  }
}

// reduce-scatter across the 4 groups: expanded partial sums (FE per lane) -> compact totals (FC per lane)
template <int F> __device__ __forceinline__ void reduce_expanded(const float (&ae)[F / 16], float (&ac)[F / 64]) {
  constexpr int FE = QL<F>::FE, FC = QL<F>::FC;
  float h[FE / 2];
#pragma unroll
  for (int i = 0; i < FE / 2; ++i) h[i] = swap32_sum(ae[i], ae[i + FE / 2]);  // groups 0/1 keep e < FE/2
#pragma unroll
  for (int c = 0; c < FC; ++c) ac[c] = swap16_sum(h[c], h[c + FC]);  // even group keeps the first FC of its half
}

template <int F> struct QTile {
  float y[QL<F>::EQ][QL<F>::FE];  // this lane's slice of its group's 8 entries
  float c[QL<F>::EQ];             // raw confidence; entries beyond the row carry -1, whose weights |c| - 1 and c+ are both 0
  int cnt;                        // valid entries of the whole 32-entry tile (wave-uniform)
};

// entry t = 4 q + g of the tile covers nnz k0 + t; lanes past the end repeat the last valid entry with weight 0
template <int F, typename T>
__device__ __forceinline__ void load_qtile(QTile<F> &tile, const int32_t *__restrict__ indices,
                                           const float *__restrict__ data, const T *__restrict__ Y, int lane, int k0,
                                           int end) {
  constexpr int FE = QL<F>::FE, EQ = QL<F>::EQ;
  const int cnt = max(0, min(4 * EQ, end - k0));
  tile.cnt = cnt;
  const int g = lane >> 4;
  unsigned col[EQ];
#pragma unroll
  for (int q = 0; q < EQ; ++q) {
    const int t = 4 * q + g;
    const bool ok = t < cnt;
    const int k = k0 + (cnt > 0 ? min(t, cnt - 1) : 0);
    col[q] = cnt > 0 ? (unsigned)indices[k] : 0u;
    tile.c[q] = ok ? data[k] : -1.f;
  }
  // gathers back to back, in two wave-uniform halves (q < 4 covers tile entries 0..15); few branches keep the
  // compiler's vmcnt bookkeeping exact so that all loads of a half are in flight together
  auto gather = [&](int q) {
    const T *src = Y + (size_t)col[q] * F + 4 * (lane & 15);
#pragma unroll
    for (int e = 0; e < FE; e += 4) {
      const float4 v = load4(src + 16 * e);  // expanded slots e..e+3 = factors 64 (e/4) + 4 m ..
      tile.y[q][e] = v.x, tile.y[q][e + 1] = v.y, tile.y[q][e + 2] = v.z, tile.y[q][e + 3] = v.w;
    }
  };
  auto clear = [&](int q) {
#pragma unroll
    for (int e = 0; e < FE; ++e) tile.y[q][e] = 0.f;
  };
  if (cnt > 16) {
#pragma unroll
    for (int q = 0; q < EQ; ++q) gather(q);
  } else if (cnt > 0) {
#pragma unroll
    for (int q = 0; q < EQ / 2; ++q) gather(q);
#pragma unroll
    for (int q = EQ / 2; q < EQ; ++q) clear(q);
  } else {
#pragma unroll
    for (int q = 0; q < EQ; ++q) clear(q);
  }
}

// Staged variant for the resident kernels: the (column, confidence) pairs of a tile are fetched one row AHEAD, one entry
// per lane (lanes l and l + 32 both hold entry min(l, cnt - 1) of the slice) -- two registers that stay in flight
// during the previous row's CG passes -- so that a row starts with its gather addresses already in hand (one HBM
// round trip per row on the critical path instead of two).  `end` > 0 and [end - 1] must be a valid entry.
__device__ __forceinline__ void fetch_entries(const int32_t *__restrict__ indices, const float *__restrict__ data, int lane,
                                              int k0, int end, int &col, float &c) {
  const int k = min(k0 + (lane & 31), end - 1);
  col = indices[k];
  c = data[k];
}

template <int F, typename T>
__device__ __forceinline__ void load_qtile_staged(QTile<F> &tile, int col_reg, float c_reg, const T *__restrict__ Y,
                                                  int lane, int cnt) {
  constexpr int FE = QL<F>::FE, EQ = QL<F>::EQ;
  tile.cnt = cnt;
  const int g = lane >> 4;
  unsigned col[EQ];
  int src = 4 * g;  // byte address of the source lane; kept opaque so that the 8 addresses are re-derived per row
  asm volatile("" : "+v"(src));  // (hoisted out of the row loop they would cost 8 registers the kernel does not have)
#pragma unroll
  for (int q = 0; q < EQ; ++q) {
    const int t = 4 * q + g;  // entry t lives in lane t
    col[q] = (unsigned)__builtin_amdgcn_ds_bpermute(src + 16 * q, col_reg);
    const float cv = __int_as_float(__builtin_amdgcn_ds_bpermute(src + 16 * q, __float_as_int(c_reg)));
    tile.c[q] = t < cnt ? cv : -1.f;
  }
  auto gather = [&](int q) {
    const T *src = Y + (size_t)col[q] * F + 4 * (lane & 15);
#pragma unroll
    for (int e = 0; e < FE; e += 4) {
      const float4 v = load4(src + 16 * e);
      tile.y[q][e] = v.x, tile.y[q][e + 1] = v.y, tile.y[q][e + 2] = v.z, tile.y[q][e + 3] = v.w;
    }
  };
  auto clear = [&](int q) {
#pragma unroll
    for (int e = 0; e < FE; ++e) tile.y[q][e] = 0.f;
  };
  if (cnt > 16) {
#pragma unroll
    for (int q = 0; q < EQ; ++q) gather(q);
  } else if (cnt > 0) {
#pragma unroll
    for (int q = 0; q < EQ / 2; ++q) gather(q);
#pragma unroll
    for (int q = EQ / 2; q < EQ; ++q) clear(q);
  } else {
#pragma unroll
    for (int q = 0; q < EQ; ++q) clear(q);
  }
}

// ae += sum over this group's entries of w y, with the dots against the expanded vector ve
//   FIRST: w = c+ - (|c|-1) d    else: w = (|c|-1) d      (_als.pyx:190-201, 214-222)
template <int F, bool FIRST>
__device__ __forceinline__ void qtile_apply(const QTile<F> &tile, const float (&ve)[F / 16], float (&ae)[F / 16]) {
  constexpr int FE = QL<F>::FE, EQ = QL<F>::EQ;
  // this lane's share of y_q . v with two running sums (even / odd slots): maps onto v_pk_fma_f32
  auto partial = [&](int q) {
    float lo = 0.f, hi = 0.f;
#pragma unroll
    for (int e = 0; e < FE; e += 2) {
      lo = fmaf(tile.y[q][e], ve[e], lo);
      hi = fmaf(tile.y[q][e + 1], ve[e + 1], hi);
    }
    return lo + hi;
  };
  auto axpy = [&](int q, float d) {
    const float c = tile.c[q], cm1 = fabsf(c) - 1.f;  // two VALU ops per step instead of 8 more live registers
    const float w = FIRST ? (c > 0.f ? c : 0.f) - cm1 * d : cm1 * d;
#pragma unroll
    for (int e = 0; e < FE; ++e) ae[e] = fmaf(w, tile.y[q][e], ae[e]);
  };
  if (tile.cnt == 4 * EQ) {
    // full tile (most wave-tiles of the mid and long rows): no per-step branches, so the four independent
    // dot -> DPP-reduce -> axpy chains of a half can be interleaved by the scheduler instead of running back to back
#pragma unroll
    for (int h = 0; h < EQ; h += 4) {
      float d[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) d[q] = partial(h + q);
#pragma unroll
      for (int q = 0; q < 4; ++q) d[q] += dpp_mov<0x128>(d[q]);  // row_ror:8
#pragma unroll
      for (int q = 0; q < 4; ++q) d[q] += dpp_mov<0x124>(d[q]);  // row_ror:4
#pragma unroll
      for (int q = 0; q < 4; ++q) d[q] += dpp_mov<0x122>(d[q]);  // row_ror:2
#pragma unroll
      for (int q = 0; q < 4; ++q) d[q] += dpp_mov<0x121>(d[q]);  // row_ror:1
#pragma unroll
      for (int q = 0; q < 4; ++q) axpy(h + q, d[q]);
    }
    return;
  }
  // partial tile: steps in pairs (two interleaved chains), pairs beyond the row's count skipped (wave-uniform); the
  // second step of the last pair may be all padding, whose weights are 0
#pragma unroll
  for (int q = 0; q < EQ; q += 2) {
    if (4 * q < tile.cnt) {
      float d0 = partial(q), d1 = partial(q + 1);
      d0 += dpp_mov<0x128>(d0), d1 += dpp_mov<0x128>(d1);
      d0 += dpp_mov<0x124>(d0), d1 += dpp_mov<0x124>(d1);
      d0 += dpp_mov<0x122>(d0), d1 += dpp_mov<0x122>(d1);
      d0 += dpp_mov<0x121>(d0), d1 += dpp_mov<0x121>(d1);
      axpy(q, d0);
      axpy(q + 1, d1);
    }
  }
}

// Dense part split over the groups: group g adds A0[j][.] * v[j] for j = j_begin + 4 s + g, s = 0..NJ-1.  `vec_lds` is
// this wave's private LDS copy of the operand vector in natural factor order (wave-synchronous: written by the caller,
// no barrier).  With a tile resident the compiler is at its register limit and would issue the LDS reads one at a
// time, each waiting out the full LDS latency; the loop is therefore staged by hand -- the reads of B steps are issued
// back to back into their own registers, then consumed.
template <int F, int NJ>
__device__ __forceinline__ void gram_matvec_q(const float *A0s, int lda, const float *vec_lds, int lane, int j_begin,
                                              float (&ae)[F / 16]) {
  constexpr int FE = QL<F>::FE, B = NJ % 4 == 0 ? 4 : (NJ % 2 == 0 ? 2 : 1);
  const int g = lane >> 4;
  const float *vp = vec_lds + j_begin + g;
  const float *row = A0s + (size_t)(j_begin + g) * lda + 4 * (lane & 15);
#pragma unroll 1
  for (int s0 = 0; s0 < NJ; s0 += B) {
    float vj[B];
    float4 a[B][FE / 4];
#pragma unroll
    for (int b = 0; b < B; ++b) {
      vj[b] = vp[4 * (s0 + b)];
#pragma unroll
      for (int e = 0; e < FE / 4; ++e) a[b][e] = *reinterpret_cast<const float4 *>(row + (size_t)4 * (s0 + b) * lda + 64 * e);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int b = 0; b < B; ++b)
#pragma unroll
      for (int e = 0; e < FE / 4; ++e) {
        ae[4 * e] = fmaf(vj[b], a[b][e].x, ae[4 * e]);
        ae[4 * e + 1] = fmaf(vj[b], a[b][e].y, ae[4 * e + 1]);
        ae[4 * e + 2] = fmaf(vj[b], a[b][e].z, ae[4 * e + 2]);
        ae[4 * e + 3] = fmaf(vj[b], a[b][e].w, ae[4 * e + 3]);
      }
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int F> __device__ __forceinline__ float dot_compact(const float (&a)[F / 64], const float (&b)[F / 64]) {
  return wave_allsum(dot_local<F / 64>(a, b));
}

}  // namespace imp
#endif  // IMPLICIT_AMD_CSRC_ALS_QTILE_H_
