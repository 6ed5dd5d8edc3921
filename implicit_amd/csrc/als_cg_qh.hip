// K1h: the mid-row CG half sweep for float16 factor STORAGE with the tile kept as it is stored -- round 4.
//
// Arithmetic contract: the oracle's CG (implicit/cpu/_als.pyx:152-248) in fp32 on the fp16-rounded factors, as the reference's
// kernels do for dtype = float16 (implicit/gpu/als.cu:41,55,109 + convert.cuh:7-17).
//
// The idea.  The resident-tile kernels are latency-bound pipelines: a team's rows advance one rendezvous per CG pass, a wavefront
// issues for ~13 % of that cycle, and what a SIMD retires is set by how many rows are in flight, i.e. by registers per row
// (DESIGN.md section 4.1).  The fp32 kernels (als_cg_qf.hip) convert an fp16 factor row at the load and hold it as fp32: half the
// HBM bytes, the same registers, no rolling gather through the conversion -- round 3 measured fp16 storage 8 % SLOWER than fp32.
// Here a tile entry stays packed (two halves per register) for all passes and is widened inside the FMA (`v_fma_mix_f32`: f16
// operand, fp32 accumulation; inline asm -- left to itself the compiler widens the tile once per pass and keeps the copy, 150
// spilled registers), and the registers are spent on 64 entries per wavefront instead of 32: every row class runs on HALF the
// wavefronts (rows of 33..64 nonzeros on ONE, no team protocol at all; up to 128 on two, 256 on four, 512 on eight), twice the
// rows are in flight per CU, and the raw tile rolls in during the last pass again.
//
// What it buys, and what caps it.  configs[2], fp16 storage, per iteration (gpurun_out/r5k, same box): row classes (32,64] /
// (64,128] / (128,256] / (256,512] 0.84 / 0.70 / 0.47 / 0.27 ms against 1.00 / 0.83 / 0.58 / 0.36 ms for the fp32-tile kernels
// (-18 %), whole iteration 4.39-4.53 against 4.91-4.96 ms -- fp16 storage is now level with fp32 storage (4.47-4.6 ms) instead
// of behind it.  It cannot get ahead: `v_fma_mix_f32` issues every 4.16 cycles per SIMD and `v_cvt_f32_f16` every 4.06, against
// 2.3 for a plain `v_fma_f32` and 4.4 for a `v_pk_fma_f32` that does TWO fp32 FMAs per lane (profiles/r04_micro_valu_rate.txt):
// widening a half costs as much issue time as the FMA it feeds, whichever instruction does it, so the tile part of a pass takes
// twice the vector time of the fp32 tile (knock-outs, gpurun_out/r5j: without the tile FMAs 4.03 ms, without the gramian part
// 4.13, without both 3.23).  The only full-rate mixed form, `v_dot2c_f32_f16`, needs BOTH operands in f16.  IMP_HALF_TILE64=0
// selects the fp32-tile kernels for fp16 storage (A/B, in the switch test).
//
// Everything else is the leader protocol of als_cg_qf.hip (two LDS counters per team, operand published in natural order, fused
// dense ticks, weight table, pair-wise DPP reduction, x-only last step); the shared pieces live in als_qf_common.h.
#include <hip/hip_fp16.h>

#include <type_traits>

#include "als_qf_common.h"
#include "common.h"

namespace imp {

namespace {
constexpr int kHT = 64;           // tile entries per wavefront
constexpr int kHPairs = kHT / 8;  // pairs of tile steps (a step = 4 entries, one per 16-lane group)

// lanes l hold entry min(l, cnt - 1) of the wave's slice (one entry per lane, all 64 lanes)
__device__ __forceinline__ void fetch_entries64(const int32_t *__restrict__ indices, const float *__restrict__ data, int lane, int k0,
                                                int end, int &col, float &c) {
  const int k = min(k0 + lane, end - 1);
  col = indices[k];
  c = data[k];
}

// d = float(half of yh) * b + c in ONE instruction.  Written as asm: given fmaf(half -> float, ..) twice on the same register (the
// dot and the axpy of an entry) the compiler converts the tile to fp32 once per pass and keeps the copy -- 64 more live registers,
// i.e. the very thing the packed tile exists to avoid (150 spilled registers).
__device__ __forceinline__ float fma_mix_lo(unsigned yh, float b, float c) {
  float d;
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(yh), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ float fma_mix_hi(unsigned yh, float b, float c) {
  float d;
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(yh), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ float mul_mix_lo(unsigned yh, float b) {
  float d;
  asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(yh), "v"(b));
  return d;
}
__device__ __forceinline__ float mul_mix_hi(unsigned yh, float b) {
  float d;
  asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(yh), "v"(b));
  return d;
}

// One register-resident element of the tile = expanded slots (2 h, 2 h + 1) of an entry: two halves in one register.  (The fp32
// form of this tile -- a packed fp32 pair per element, two "fat" wavefronts per SIMD -- measured slower than the 32-entry tiles of
// als_cg_qf.hip in round 4 and was removed in round 6.)  The products associate alike in both (and as in als_cg_qf.hip): even slots in
// one running sum, odd slots in the other.
template <typename ST> struct Tile64;
template <> struct Tile64<__half> {
  typedef unsigned elem;
  template <int H> static __device__ __forceinline__ void gather(elem (&yq)[H], const __half *p) {
#pragma unroll
    for (int h = 0; h < H; h += 2) {  // 4 halves = one 8-byte load
      const uint2 raw = *reinterpret_cast<const uint2 *>(p + 32 * h);
      yq[h] = raw.x, yq[h + 1] = raw.y;
    }
  }
  template <int H> static __device__ __forceinline__ float dot(const elem (&yq)[H], const f32x2 (&ve)[H]) {
    float s0 = mul_mix_lo(yq[0], ve[0].x), s1 = mul_mix_hi(yq[0], ve[0].y);
#pragma unroll
    for (int h = 1; h < H; ++h) {
      s0 = fma_mix_lo(yq[h], ve[h].x, s0);
      s1 = fma_mix_hi(yq[h], ve[h].y, s1);
    }
    return s0 + s1;
  }
  template <int H> static __device__ __forceinline__ void axpy(const elem (&yq)[H], float w, f32x2 (&ae)[H]) {
#pragma unroll
    for (int h = 0; h < H; ++h) {
      ae[h].x = fma_mix_lo(yq[h], w, ae[h].x);
      ae[h].y = fma_mix_hi(yq[h], w, ae[h].y);
    }
  }
};

// Entries of tile steps 2 P and 2 P + 1 (lanes 8 P .. 8 P + 7 of the staged registers): gather addresses by ds_bpermute, weights
// |c| - 1 and c+ to the wave's LDS table (cw[t], cw[64 + t]), the factor rows as they are stored.
template <int F, int P, typename ST>
__device__ __forceinline__ void gather_pair64(typename Tile64<ST>::elem (&y)[kHT / 4][F / 32], float *cw, int col_reg, float c_reg, int cnt,
                                              const ST *__restrict__ Y, int lane) {
  lane = opaque(lane);
  if ((lane >> 3) == P) {
    const bool ok = lane < cnt;
    cw[lane] = ok ? fabsf(c_reg) - 1.f : 0.f;
    cw[kHT + lane] = ok ? fmaxf(c_reg, 0.f) : 0.f;
  }
  const int src = 4 * (lane >> 4);  // byte address of the source lane: entry t = 4 q + g sits in lane t
#pragma unroll
  for (int q = 2 * P; q < 2 * P + 2; ++q) {
    const unsigned col = (unsigned)__builtin_amdgcn_ds_bpermute(src + 16 * q, col_reg);
    Tile64<ST>::template gather<F / 32>(y[q], Y + (size_t)col * F + 4 * (lane & 15));
  }
}

template <int H> struct NoDenseTicks {  // IMP_QH_KO_DENSE
  template <int K> __device__ __forceinline__ void issue(const float *, const float *) {}
  template <int K> __device__ __forceinline__ void consume(f32x2 (&)[H]) {}
};

// One pass over this wave's share of a row (fused_pass of als_cg_qf.hip with a packed tile of 8 pairs and 32 dense ticks).
// The products associate exactly as in the fp32 kernel: even expanded slots in one running sum, odd slots in the other.
template <int F, int NJ, bool FIRST, bool LAST, typename ST>
__device__ __forceinline__ void fused_pass64(typename Tile64<ST>::elem (&y)[kHT / 4][F / 32], float *cw, int cnt, const float *vt,
                                             int j_begin, const float *A0s, float (&acc)[F / 64], int lane, int cnt_nx, int &col_nx,
                                             float &c_nx, const ST *__restrict__ Y, const int32_t *__restrict__ indices,
                                             const float *__restrict__ data, int k0_nx2, int end_nx2) {
  constexpr int FE = F / 16, H = FE / 2;
  if constexpr (LAST) {  // one wait for the staged entries, before any rolling gather
    col_nx = opaque(col_nx);
    c_nx = __int_as_float(opaque(__float_as_int(c_nx)));
  }
  f32x2 ve[H], ae[H];
  const float *row, *vp, *cwg;
  {
    const int ln = opaque(lane);
    const int g = ln >> 4, m = ln & 15;
#pragma unroll
    for (int e = 0; e < FE; e += 4) {  // the operand, expanded: slot e of lane (g, m) is factor 64 (e / 4) + 4 m + (e & 3)
      const float4 t = *reinterpret_cast<const float4 *>(vt + 16 * e + 4 * m);
      ve[e / 2] = f32x2{t.x, t.y}, ve[e / 2 + 1] = f32x2{t.z, t.w};
    }
    vp = vt + j_begin + g * NJ;
    row = A0s + (size_t)(j_begin + g * NJ) * F + 4 * m;
    cwg = cw + g;  // this group's entries: t = 4 q + g
  }
#pragma unroll
  for (int h = 0; h < H; ++h) ae[h] = f32x2{0.f, 0.f};
#ifdef IMP_QH_KO_DENSE  // timing-only knock-out: no gramian part
  NoDenseTicks<H> dt;
#else
  DenseTicks<F, NJ, 4 * kHPairs> dt;
#endif
  auto partial = [&](int q) {
#ifdef IMP_QH_KO_TILE  // timing-only knock-out: no tile arithmetic (results wrong)
    return ve[0].x;
#endif
    return Tile64<ST>::template dot<H>(y[q], ve);
  };
  auto axpy = [&](int q, float w) {
#ifdef IMP_QH_KO_TILE
    ae[0].x += w;
    return;
#endif
    Tile64<ST>::template axpy<H>(y[q], w, ae);
  };
  static_for<kHPairs>([&](auto Pc) {
    constexpr int P = decltype(Pc)::value;
    if (8 * P < cnt) {  // wave-uniform
      dt.template issue<4 * P>(row, vp);
      const float cm1_0 = cwg[8 * P], cm1_1 = cwg[8 * P + 4];
      float cp_0 = 0.f, cp_1 = 0.f;
      if constexpr (FIRST) cp_0 = cwg[kHT + 8 * P], cp_1 = cwg[kHT + 8 * P + 4];
      __builtin_amdgcn_sched_barrier(0);
      const float d0 = partial(2 * P);
      __builtin_amdgcn_sched_barrier(0);
      dt.template consume<4 * P>(ae);
      dt.template issue<4 * P + 1>(row, vp);
      __builtin_amdgcn_sched_barrier(0);
      const float d1 = partial(2 * P + 1);
      const float u = reduce_pair(d0, d1);
      dt.template consume<4 * P + 1>(ae);
      // the whole first pass is accumulated negated: w' = (|c|-1) d - c+
      const float w0 = FIRST ? fmaf(cm1_0, row_bcast_from<0>(u), -cp_0) : cm1_0 * row_bcast_from<0>(u);
      const float w1 = FIRST ? fmaf(cm1_1, row_bcast_from<8>(u), -cp_1) : cm1_1 * row_bcast_from<8>(u);
      __builtin_amdgcn_sched_barrier(0);
      dt.template issue<4 * P + 2>(row, vp);
      __builtin_amdgcn_sched_barrier(0);
      axpy(2 * P, w0);
      __builtin_amdgcn_sched_barrier(0);
      dt.template consume<4 * P + 2>(ae);
      dt.template issue<4 * P + 3>(row, vp);
      __builtin_amdgcn_sched_barrier(0);
      axpy(2 * P + 1, w1);
      __builtin_amdgcn_sched_barrier(0);
      dt.template consume<4 * P + 3>(ae);
    } else {  // no entries left: the remaining gramian rows
      asm volatile("" ::: "memory");
      static_for<4>([&](auto Kc) {
        constexpr int K = 4 * P + decltype(Kc)::value;
        dt.template issue<K>(row, vp);
        __builtin_amdgcn_sched_barrier(0);
        dt.template consume<K>(ae);
      });
    }
    if constexpr (LAST) {
      if (8 * P < cnt_nx) gather_pair64<F, P, ST>(y, cw, col_nx, c_nx, cnt_nx, Y, lane);
    }
    __builtin_amdgcn_sched_barrier(0);
  });
  float aes[FE];
#pragma unroll
  for (int h = 0; h < H; ++h) aes[2 * h] = ae[h].x, aes[2 * h + 1] = ae[h].y;
  reduce_expanded<F>(aes, acc);
  if constexpr (LAST) fetch_entries64(indices, data, opaque(lane), k0_nx2, end_nx2, col_nx, c_nx);
}
}  // namespace

// Rows [first, first + count) of the schedule, a team of WPR wavefronts per row, up to 64 WPR nonzeros per row.
// Registers: the 64-entry tile is F / 2 registers for fp32 storage (128 at f = 128: two "fat" wavefronts per SIMD), F / 4 for fp16.
template <int F, int WPR, int BLOCK, typename ST>
__global__ __launch_bounds__(BLOCK, (F == 64 ? 8 : 4) / (int)(sizeof(ST) / 2)) void als_cg_q64team_kernel(
    const int32_t *__restrict__ order, int first, int count, const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
    const float *__restrict__ data, ST *__restrict__ X, const ST *__restrict__ Y, const float *__restrict__ A0, int cg_steps) {
  constexpr int FC = F / 64, FE = F / 16, T = kHT, WAVES = BLOCK / 64, TEAMS = WAVES / WPR, NJ = F / WPR / 4;
  static_assert(WPR <= WAVES && (F / WPR) % 4 == 0 && NJ <= 4 * kHPairs, "team width");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *A0s = smem;                            // [F][F]
  float *parts = A0s + (size_t)F * F;           // [WAVES][F]  partial vectors of the waves
  float *vts = parts + (size_t)WAVES * F;       // [TEAMS][F]  the operand the leader published (natural factor order)
  float *cws = vts + (size_t)TEAMS * F;         // [WAVES][2 T]  per-entry weights |c| - 1 and c+ of the resident tile
  unsigned *ctl = reinterpret_cast<unsigned *>(cws + (size_t)WAVES * 2 * T);  // [TEAMS][4]  arrivals, generation, control words
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int team = wave / WPR, sub = wave % WPR;
  const bool leader = sub == 0;
  for (int e = threadIdx.x; e < F * F; e += BLOCK) A0s[e] = A0[e];
  if (threadIdx.x < 4 * TEAMS) ctl[threadIdx.x] = 0u;
  __syncthreads();  // the only workgroup-wide barrier
  const int j_begin = F * sub / WPR;
  float *vt = vts + (size_t)team * F;
  float *cw = cws + (size_t)wave * 2 * T;
  unsigned *arrivals = ctl + 4 * team, *generation = arrivals + 1, *words = arrivals + 2;

  // ---- team protocol (als_cg_qf.hip has the commentary) ----------------------------------------------------------------
  unsigned gen = 0, pub = 0, arr_target = 0;
  auto lds_off = [](const void *ptr) { return (unsigned)(size_t)ptr; };
  const unsigned arrivals_off = lds_off(arrivals), generation_off = lds_off(generation), words_off = lds_off(words);
  const unsigned cf4 = 4u * (unsigned)QL<F>::cfactor(lane, 0);
  auto publish = [&](unsigned w) {  // leader
    ++pub;
    if (lane == 0)
      asm volatile("ds_write_b32 %0, %1\n\tds_add_u32 %2, %3" ::"v"(words_off + 4u * (pub & 1u)), "v"(w), "v"(generation_off), "v"(1u)
                   : "memory");
  };
  auto poll = [&](unsigned off) {
    typedef __attribute__((address_space(3))) volatile unsigned lds_word;
    return (unsigned)__builtin_amdgcn_readfirstlane(*(lds_word *)(size_t)off);
  };
  auto await_operand = [&]() -> unsigned {
    ++gen;
    if constexpr (WPR > 1) {
      if (poll(generation_off) < gen) {
        __builtin_amdgcn_s_sleep(IMP_TEAM_NAP_FIRST);
        while (poll(generation_off) < gen) __builtin_amdgcn_s_sleep(IMP_TEAM_NAP_NEXT);
      }
    }
    return poll(words_off + 4u * (gen & 1u));
  };
  auto arrive = [&](const float (&acc)[FC]) {
    float *slot = reinterpret_cast<float *>(reinterpret_cast<char *>(parts + (size_t)wave * F) + cf4);
    if constexpr (FC == 2) *reinterpret_cast<float2 *>(slot) = make_float2(acc[0], acc[1]);
    else slot[0] = acc[0];
    if constexpr (WPR > 1) {
      if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(arrivals_off), "v"(1u) : "memory");
    }
  };
  auto collect = [&](float (&acc)[FC]) {  // leader: wait for the team, sum its partials in wave order
    arr_target += WPR;
    if constexpr (WPR > 1) {
      while (poll(arrivals_off) < arr_target) __builtin_amdgcn_s_sleep(IMP_TEAM_NAP_LEADER);
    }
    const float *slot = reinterpret_cast<const float *>(reinterpret_cast<const char *>(parts + (size_t)(team * WPR) * F) + cf4);
#pragma unroll
    for (int c = 0; c < FC; ++c) acc[c] = 0.f;
#pragma unroll
    for (int w = 0; w < WPR; ++w) {
      if constexpr (FC == 2) {
        const float2 t = *reinterpret_cast<const float2 *>(slot + (size_t)w * F);
        acc[0] += t.x, acc[1] += t.y;
      } else {
        acc[0] += slot[(size_t)w * F];
      }
    }
  };
  auto operand_slot = [&]() { return reinterpret_cast<float *>(reinterpret_cast<char *>(vt) + cf4); };
  auto put_operand = [&](const float (&v)[FC]) {
    float *slot = operand_slot();
    if constexpr (FC == 2) *reinterpret_cast<float2 *>(slot) = make_float2(v[0], v[1]);
    else slot[0] = v[0];
  };
  auto get_operand = [&](float (&v)[FC]) {
    const float *slot = operand_slot();
    if constexpr (FC == 2) {
      const float2 t = *reinterpret_cast<const float2 *>(slot);
      v[0] = t.x, v[1] = t.y;
    } else {
      v[0] = slot[0];
    }
  };

  auto row_id = [&](int i) { return order[first + min(i, count - 1)]; };  // uniform address: scalar load
  const int i_step = gridDim.x * TEAMS, i_first = blockIdx.x * TEAMS + team;
  auto slice = [&](int rb, int re, int &k0, int &cnt) {  // even shares rounded up to whole 4-entry tile steps
    const int chunk = min(T, (((re - rb) + WPR - 1) / WPR + 3) & ~3);
    k0 = min(rb + chunk * sub, re);
    cnt = min(chunk, re - k0);
  };
  int id0 = row_id(i_first), id1 = row_id(i_first + i_step), id2 = row_id(i_first + 2 * i_step), id3 = row_id(i_first + 3 * i_step);
  int b0 = indptr[id0], e0 = indptr[id0 + 1], b1 = indptr[id1], e1 = indptr[id1 + 1], b2 = indptr[id2], e2 = indptr[id2 + 1];
  int ent_col, ent_cnt, k0;
  float ent_c;
  slice(b0, e0, k0, ent_cnt);
  fetch_entries64(indices, data, opaque(lane), k0, max(k0 + ent_cnt, b0 + 1), ent_col, ent_c);
  auto kill = [](float (&v)[FC]) {
#pragma unroll
    for (int cc = 0; cc < FC; ++cc) v[cc] = 0.f;
  };
  bool tile_ready = false;
  int cnt = 0;
  typename Tile64<ST>::elem y[T / 4][FE / 2];  // the resident tile: 64 entries
  float x[FC];
  kill(x);
  for (int i = i_first; i < count; i += i_step) {
    ST *xrow = X + (size_t)id0 * F;
    if (!tile_ready) {  // first row of the wave, or the previous row ended before its last pass: plain row start
      cnt = ent_cnt;
      ent_col = opaque(ent_col);
      ent_c = __int_as_float(opaque(__float_as_int(ent_c)));
      static_for<kHPairs>([&](auto Pc) {
        constexpr int P = decltype(Pc)::value;
        if (8 * P < cnt) gather_pair64<F, P, ST>(y, cw, ent_col, ent_c, cnt, Y, lane);
      });
      slice(b1, e1, k0, ent_cnt);
      fetch_entries64(indices, data, opaque(lane), k0, max(k0 + ent_cnt, b1 + 1), ent_col, ent_c);
      if (leader) load_compact<F>(xrow, opaque(lane), x);
      else kill(x);
    }
    // ent_* now describe row i + i_step
    float xc[FC], r[FC], p[FC], Ap[FC], rsold = 0.f;  // leader state
#pragma unroll
    for (int cc = 0; cc < FC; ++cc) xc[cc] = r[cc] = 0.f;
    bool store = false;
    if (leader) {
      put_operand(x);
#pragma unroll
      for (int cc = 0; cc < FC; ++cc) xc[cc] = x[cc];
      publish(kGo);
    }
    unsigned w = await_operand();
    {
      float acc[FC];
      fused_pass64<F, NJ, true, false, ST>(y, cw, cnt, vt, j_begin, A0s, acc, lane, 0, ent_col, ent_c, Y, nullptr, nullptr, 0, 0);
      arrive(acc);
    }
    if (leader) {
      collect(r);
#pragma unroll
      for (int cc = 0; cc < FC; ++cc) r[cc] = -r[cc], p[cc] = r[cc];
      rsold = dot_compact<F>(r, r);
      store = rsold >= 1e-20f;  // else: x untouched (_als.pyx:206)
      if (store && cg_steps > 0) {
        put_operand(p);
        publish(kGo | (cg_steps == 1 ? kLast : 0u));
      } else {
        publish(0u);
      }
    }
    w = await_operand();
    for (int it = 0; (w & (kGo | kLast)) == kGo; ++it) {  // all steps but the last
      float acc[FC];
      fused_pass64<F, NJ, false, false, ST>(y, cw, cnt, vt, j_begin, A0s, acc, lane, 0, ent_col, ent_c, Y, nullptr, nullptr, 0, 0);
      arrive(acc);
      if (leader) {
        collect(Ap);
        get_operand(p);
        const float alpha = rsold * __builtin_amdgcn_rcpf(dot_compact<F>(p, Ap));
#pragma unroll
        for (int cc = 0; cc < FC; ++cc) {
          xc[cc] = fmaf(alpha, p[cc], xc[cc]);
          r[cc] = fmaf(-alpha, Ap[cc], r[cc]);
        }
        const float rsnew = dot_compact<F>(r, r);
        if (rsnew < 1e-20f) {
          publish(0u);  // the oracle breaks here (_als.pyx:235)
        } else {
          const float beta = rsnew * __builtin_amdgcn_rcpf(rsold);
#pragma unroll
          for (int cc = 0; cc < FC; ++cc) p[cc] = fmaf(beta, p[cc], r[cc]);
          rsold = rsnew;
          put_operand(p);
          publish(kGo | (it + 2 >= cg_steps ? kLast : 0u));
        }
      }
      w = await_operand();
    }
    // the last step: its pass rolls the next row's tile in, only its x update is evaluated (_als.pyx:226-241)
    const bool rolled = (w & kGo) != 0u;
    if (w & kGo) {
      float acc[FC];
      int k2, cnt2;
      slice(b2, e2, k2, cnt2);
      if (i + i_step >= count) ent_cnt = 0;  // no next row: nothing to gather
      fused_pass64<F, NJ, false, true, ST>(y, cw, cnt, vt, j_begin, A0s, acc, lane, ent_cnt, ent_col, ent_c, Y, indices, data, k2,
                                       max(k2 + cnt2, b2 + 1));
      cnt = ent_cnt;
      ent_cnt = cnt2;
      if (leader) load_compact<F>(X + (size_t)id1 * F, opaque(lane), x);  // the next row's iterate
      else kill(x);
      arrive(acc);
      if (leader) {
        collect(Ap);
        get_operand(p);
        const float alpha = rsold * __builtin_amdgcn_rcpf(dot_compact<F>(p, Ap));
#pragma unroll
        for (int cc = 0; cc < FC; ++cc) xc[cc] = fmaf(alpha, p[cc], xc[cc]);
        publish(0u);
      }
      (void)await_operand();  // the stop generation
    } else {
      kill(x);
    }
    if (leader && store) store_compact<F>(xrow, opaque(lane), xc);
    tile_ready = rolled;
    id0 = id1, id1 = id2, id2 = id3, id3 = row_id(i + 4 * i_step);
    b0 = b1, e0 = e1, b1 = b2, e1 = e2, b2 = indptr[id2], e2 = indptr[id2 + 1];
  }
}

template <int F, int WPR, int BLOCK, typename ST>
static void launch_q64team(const imp_csr *C, int first, int count, ST *X, const ST *Y, const float *A0, int cg_steps, const char *name) {
  if (count <= 0) return;
  constexpr int WAVES = BLOCK / 64, TEAMS = WAVES / WPR;
  const size_t lds = ((size_t)F * F + (size_t)WAVES * F + (size_t)TEAMS * F + 2 * kHT * WAVES + 4 * TEAMS) * sizeof(float);
  auto kern = als_cg_q64team_kernel<F, WPR, BLOCK, ST>;
  IMP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int per_cu = 0;  // registers decide at f = 128 / fp32 (one 8-wave workgroup per CU), the LDS elsewhere
  IMP_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, BLOCK, lds));
  per_cu = std::max(1, per_cu);
  constexpr int kBaseOversub = WPR <= 2 ? 4 : 2;  // as launch_qfteam for the same row classes
  const int grid = std::min((count + TEAMS - 1) / TEAMS, ctx().num_cus * per_cu * std::max(kBaseOversub, ctx().oversub));
  IMP_PROF(name);
  kern<<<grid, BLOCK, lds, stream()>>>(C->order.data(), first, count, C->indptr.data(), C->indices.data(), C->data.data(), X, Y, A0,
                                      cg_steps);
  IMP_CHECK_HIP(hipGetLastError());
}

// width: wavefronts per row, 1 / 2 / 4 / 8 (rows of up to 64 / 128 / 256 / 512 nonzeros)
template <typename ST>
void launch_team_tile64(const imp_csr *C, int f, int width, int first, int count, ST *X, const ST *Y, const float *A0, int cg_steps,
                        const char *name) {
  auto run = [&](auto Fc) {
    constexpr int F = decltype(Fc)::value;
    switch (width) {
      case 8: launch_q64team<F, 8, 512, ST>(C, first, count, X, Y, A0, cg_steps, name); break;
      case 4: launch_q64team<F, 4, 512, ST>(C, first, count, X, Y, A0, cg_steps, name); break;
      case 2: launch_q64team<F, 2, 512, ST>(C, first, count, X, Y, A0, cg_steps, name); break;
      case 1: launch_q64team<F, 1, 512, ST>(C, first, count, X, Y, A0, cg_steps, name); break;
      default: throw std::invalid_argument("launch_team_tile64: team width");
    }
  };
  if (f == 128) run(idx_t<128>{});
  else if (f == 64) run(idx_t<64>{});
  else throw std::invalid_argument("launch_team_tile64: f must be 64 or 128");
}
template void launch_team_tile64<__half>(const imp_csr *, int, int, int, int, __half *, const __half *, const float *, int, const char *);

}  // namespace imp
