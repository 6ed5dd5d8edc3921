// wave64 building blocks shared by the ALS kernels (gfx950): DPP reductions, lane broadcast and the
// lane <-> factor layouts.  Replaces the reference's 32-lane shuffle + shared-memory block reduction
// (implicit/gpu/dot.cuh:9-59) -- here a row is owned by ONE wavefront, so a dot product is a
// register-only DPP tree with no LDS traffic and no barrier.
#ifndef IMPLICIT_AMD_CSRC_WAVE_OPS_H_
#define IMPLICIT_AMD_CSRC_WAVE_OPS_H_
#include <hip/hip_runtime.h>

namespace imp {

// ---- wave64 reductions on DPP (no LDS crossbar) --------------------------------------------------
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}

// Sum over the 64 lanes, returned in every lane (via SGPR broadcast of lane 63).
__device__ __forceinline__ float wave_allsum(float v) {
  v += dpp_mov<0x128>(v);        // row_ror:8
  v += dpp_mov<0x124>(v);        // row_ror:4
  v += dpp_mov<0x122>(v);        // row_ror:2
  v += dpp_mov<0x121>(v);        // row_ror:1   -> every lane holds its 16-lane row sum
  v += dpp_mov<0x142, 0xA>(v);   // row_bcast:15 into rows 1,3
  v += dpp_mov<0x143, 0xC>(v);   // row_bcast:31 into rows 2,3 -> lane 63 = total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Maximum over the 64 lanes, in every lane (NaN inputs are dropped by fmaxf).
__device__ __forceinline__ float wave_allmax(float v) {
  v = fmaxf(v, dpp_mov<0x128>(v));
  v = fmaxf(v, dpp_mov<0x124>(v));
  v = fmaxf(v, dpp_mov<0x122>(v));
  v = fmaxf(v, dpp_mov<0x121>(v));  // every lane: the maximum of its 16-lane row
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

__device__ __forceinline__ float lane_bcast(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ int lane_bcast(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

// element index owned by (lane, slot): contiguous chunks when f == 64*VPL (vector loads), lane-strided otherwise
template <int VPL, bool VEC> __device__ __forceinline__ int elem(int lane, int v) {
  return VEC ? lane * VPL + v : lane + 64 * v;
}

template <int VPL, bool VEC>
__device__ __forceinline__ void load_row(const float *__restrict__ base, int f, int lane, float (&y)[VPL]) {
  if constexpr (VEC) {
    if constexpr (VPL == 1) {
      y[0] = base[lane];
    } else if constexpr (VPL == 2) {
      float2 t = *reinterpret_cast<const float2 *>(base + lane * 2);
      y[0] = t.x, y[1] = t.y;
    } else {
      static_assert(VPL == 4, "vector path supports f = 64, 128, 256");
      float4 t = *reinterpret_cast<const float4 *>(base + lane * 4);
      y[0] = t.x, y[1] = t.y, y[2] = t.z, y[3] = t.w;
    }
  } else {
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      int e = lane + 64 * v;
      y[v] = e < f ? base[e] : 0.f;
    }
  }
}

template <int VPL> __device__ __forceinline__ float dot_local(const float (&a)[VPL], const float (&b)[VPL]) {
  float s = 0.f;
#pragma unroll
  for (int v = 0; v < VPL; ++v) s = fmaf(a[v], b[v], s);
  return s;
}


}  // namespace imp
#endif  // IMPLICIT_AMD_CSRC_WAVE_OPS_H_
