// Building blocks shared by the leader-protocol team kernels: the fp32 form (als_cg_qf.hip) and the packed-half form that keeps
// twice the entries per wavefront (als_cg_qh.hip).
#ifndef IMPLICIT_AMD_CSRC_ALS_QF_COMMON_H_
#define IMPLICIT_AMD_CSRC_ALS_QF_COMMON_H_
#include <type_traits>
#include <utility>

#include "als_qtile.h"

namespace imp {

// The compiler hoists everything derived from the lane id out of the row loop (byte offsets, 64-bit gather bases, LDS
// addresses: a dozen registers) and then spills it, because the tile fills the file.  Lane-derived values are therefore
// re-derived where they are used, from a copy of the lane id the optimiser cannot see through.
__device__ __forceinline__ int opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

// explicit packed math: pairs of adjacent expanded slots travel as one 64-bit register pair (v_pk_fma_f32); left to the
// SLP vectoriser the dots came out as scalar v_fmac chains once the operand arrived by ds_read_b128
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int I> using idx_t = std::integral_constant<int, I>;
template <int N, typename Fn, int... Is> __device__ __forceinline__ void static_for_impl(Fn &&fn, std::integer_sequence<int, Is...>) {
  (fn(idx_t<Is>{}), ...);
}
template <int N, typename Fn> __device__ __forceinline__ void static_for(Fn &&fn) {
  static_for_impl<N>(fn, std::make_integer_sequence<int, N>{});
}

// The gramian rows of one wave and pass dealt to 16 ticks, four per pair of tile steps.  The wave's F / WPR rows are cut
// into four runs of NJ consecutive rows, one per 16-lane group: step s of group g is row j_begin + g NJ + s, so a group's
// operand entries p_j are consecutive and travel two at a time (ds_read_b64 costs the LDS the same two cycles as a b32).
// One step = FE/4 ds_read_b128 in flight per tick (8 registers at f = 128) + the operand pair.
template <int F, int NJ, int TICKS = 16> struct DenseTicks {  // TICKS = 4 per pair of tile steps: 16 for 32-entry tiles, 32 for 64
  static constexpr int FE = F / 16, Q4 = FE / 4;
  static constexpr int EVERY = TICKS / NJ;  // ticks K with K % EVERY == 0 carry one step
  static_assert(NJ >= 1 && NJ <= TICKS && TICKS % NJ == 0 && (NJ & (NJ - 1)) == 0, "steps per pass");
  float4 a[Q4];
  f32x2 vj2;
  template <int K> __device__ __forceinline__ void issue(const float *row, const float *vp) {
    if constexpr (K % EVERY == 0) {
      constexpr int s = K / EVERY;
      if constexpr (NJ == 1) vj2 = f32x2{vp[0], 0.f};
      else if constexpr (s % 2 == 0) vj2 = *reinterpret_cast<const f32x2 *>(vp + s);
#pragma unroll
      for (int e = 0; e < Q4; ++e) a[e] = *reinterpret_cast<const float4 *>(row + (size_t)s * F + 64 * e);
    }
  }
  template <int K> __device__ __forceinline__ void consume(f32x2 (&ae)[FE / 2]) {
    if constexpr (K % EVERY == 0) {
      constexpr int s = K / EVERY;
      const float vj = (s % 2 == 0) ? vj2.x : vj2.y;
      const f32x2 v2 = {vj, vj};
#pragma unroll
      for (int e = 0; e < Q4; ++e) {
        ae[2 * e] = __builtin_elementwise_fma(v2, f32x2{a[e].x, a[e].y}, ae[2 * e]);
        ae[2 * e + 1] = __builtin_elementwise_fma(v2, f32x2{a[e].z, a[e].w}, ae[2 * e + 1]);
      }
    }
  }
};

// the dots of two tile steps, reduced over the 16 lanes of each group TOGETHER: after the first level the lower half-row
// carries d0's pair sums and the upper half d1's, the remaining three levels (half-row mirror, quad xor 1, quad xor 2: all
// inside a half-row) then serve both.  Lanes 0-7 of every row end with the total of d0, lanes 8-15 with the total of d1.
__device__ __forceinline__ float reduce_pair(float d0, float d1) {
  float u = d1 + dpp_mov<0x128>(d1);  // row_ror:8
  const float s0 = d0 + dpp_mov<0x128>(d0);
  u = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, u), __builtin_bit_cast(int, s0), 0xE4, 0xF, 0x3,
                                                            false));  // quad_perm:[0,1,2,3] into banks 0, 1 = lanes 0-7
  u += dpp_mov<0x141>(u);  // row_half_mirror
  u += dpp_mov<0xB1>(u);   // quad_perm:[1,0,3,2]
  u += dpp_mov<0x4E>(u);   // quad_perm:[2,3,0,1]
  return u;
}
template <int LANE> __device__ __forceinline__ float row_bcast_from(float v) {  // row_newbcast:LANE (gfx90a+)
  return dpp_mov<0x150 + LANE>(v);
}

// control word the leader publishes with every operand
enum : unsigned { kGo = 1u, kLast = 2u };

// tunables of the team protocol (compile-time: s_sleep / s_setprio take immediates; -D overrides for A/B builds,
// implicit_amd/_build.py build_variant)
#ifndef IMP_TEAM_NAP_FIRST
#define IMP_TEAM_NAP_FIRST 6   // a worker's first nap while the leader updates (64-cycle units)
#endif
#ifndef IMP_TEAM_NAP_NEXT
#define IMP_TEAM_NAP_NEXT 2    // its later naps
#endif
#ifndef IMP_TEAM_NAP_LEADER
#define IMP_TEAM_NAP_LEADER 1  // the leader's naps while it waits for the arrivals
#endif
#ifndef IMP_TEAM_LEADER_PRIO
#define IMP_TEAM_LEADER_PRIO 0 // wave priority of a leader from "arrivals complete" to "operand published" (the team idles meanwhile)
#endif


}  // namespace imp
#endif  // IMPLICIT_AMD_CSRC_ALS_QF_COMMON_H_
