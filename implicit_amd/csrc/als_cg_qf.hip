// K1f: the mid-row CG half sweep (a team of WPR wavefronts per row, whole row resident; als_cg_q.hip has the design) with the
// passes re-scheduled around their two latencies -- round 3.
//
// Arithmetic contract: the oracle's CG (implicit/cpu/_als.pyx:152-248).  What changes against als_cg_qteam_kernel is the
// ORDER of work inside a wavefront, not the work:
//
//   * fused pass.  A pass adds two independent things into the same expanded accumulators: the dense part (this wave's
//     gramian rows times the operand, LDS reads) and the tile part (dots and axpys over the resident entries, registers
//     only).  Run one after the other, the dense part was a chain of dependent LDS round trips (the register file is full,
//     so only a few reads can be in flight: 20 round trips per pass at two waves per row, 2.2 K of a pass's 7 K cycles) during
//     which the wave issued nothing.  Here the gramian rows of a pass are dealt to 16 "ticks", and every tick's reads are
//     issued BEFORE a tile half-step (the dots or the axpys of one entry per group) and consumed AFTER it: the LDS latency hides under the wave's own vector work.
//   * rolling gather.  The last pass of a row frees the tile registers pair by pair; the gathers of the NEXT row's entries
//     are issued into a pair as soon as its last axpy is done, so the rest of the pass, the team combine, the CG update and
//     the store run under the next row's gather latency (it used to be exposed at every row start: 27 % of a wave's time).
//     Row metadata therefore runs one row deeper (ids 4 rows ahead, nnz ranges 3, entries 2).
//   * the first pass accumulates (A0 x - sum w y) and negates once in compact form.
#include <type_traits>

#include "als_qtile.h"
#include "common.h"

namespace imp {

// The compiler hoists everything derived from the lane id out of the row loop (byte offsets, 64-bit gather bases, LDS
// addresses: a dozen registers) and then spills it, because the tile fills the file.  Lane-derived values are therefore
// re-derived where they are used, from a copy of the lane id the optimiser cannot see through.
__device__ __forceinline__ int opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

template <int I> using idx_t = std::integral_constant<int, I>;
template <int N, typename Fn, int... Is> __device__ __forceinline__ void static_for_impl(Fn &&fn, std::integer_sequence<int, Is...>) {
  (fn(idx_t<Is>{}), ...);
}
template <int N, typename Fn> __device__ __forceinline__ void static_for(Fn &&fn) {
  static_for_impl<N>(fn, std::make_integer_sequence<int, N>{});
}

// The gramian rows of one wave and pass (NJ steps of 4 rows, one per 16-lane group) dealt to 16 ticks, four per pair of tile
// steps; one step = FE/4 ds_read_b128 + one ds_read_b32 in flight per tick (9 registers at f = 128).
template <int F, int NJ> struct DenseTicks {
  static constexpr int FE = F / 16, Q4 = FE / 4;
  static constexpr int EVERY = 16 / NJ;  // ticks K with K % EVERY == 0 carry one step
  static_assert(NJ == 16 || NJ == 8 || NJ == 4 || NJ == 2 || NJ == 1, "steps per pass");
  float4 a[Q4];
  float vj;
  template <int K> __device__ __forceinline__ void issue(const float *row, const float *vp) {
    if constexpr (K % EVERY == 0) {
      constexpr int s = K / EVERY;
      vj = vp[4 * s];
#pragma unroll
      for (int e = 0; e < Q4; ++e) a[e] = *reinterpret_cast<const float4 *>(row + (size_t)4 * s * F + 64 * e);
    }
  }
  template <int K> __device__ __forceinline__ void consume(float (&ae)[FE]) {
    if constexpr (K % EVERY == 0) {
#pragma unroll
      for (int e = 0; e < Q4; ++e) {
        ae[4 * e] = fmaf(vj, a[e].x, ae[4 * e]);
        ae[4 * e + 1] = fmaf(vj, a[e].y, ae[4 * e + 1]);
        ae[4 * e + 2] = fmaf(vj, a[e].z, ae[4 * e + 2]);
        ae[4 * e + 3] = fmaf(vj, a[e].w, ae[4 * e + 3]);
      }
    }
  }
};

// Entries of tile steps 2 P and 2 P + 1.  The staged registers hold entry min(l, cnt - 1) of the wave's slice in lanes l and
// l + 32 (fetch_entries): the gather addresses travel by ds_bpermute (entry t = 4 q + g -> the 16 lanes of group g), the two
// weights every pass derives from a confidence -- |c| - 1 and c+ = max(c, 0), both 0 for the padding entries -- are written
// ONCE to a wave-private LDS table by the lanes that hold the entries (cw[t] = |c| - 1, cw[32 + t] = c+) and read back
// per step as a group-wide broadcast: 8 registers less than carrying them, and no per-pass abs / max.
template <int F, int P, typename ST>
__device__ __forceinline__ void gather_pair(float (&y)[8][F / 16], float *cw, int col_reg, float c_reg, int cnt,
                                            const ST *__restrict__ Y, int lane) {
  constexpr int FE = F / 16;
  lane = opaque(lane);
  if ((lane >> 3) == P) {  // lanes 8 P .. 8 P + 7 hold the entries of this pair
    const bool ok = lane < cnt;
    cw[lane] = ok ? fabsf(c_reg) - 1.f : 0.f;
    cw[32 + lane] = ok ? fmaxf(c_reg, 0.f) : 0.f;
  }
  const int src = 4 * (lane >> 4);  // byte address of the source lane
#pragma unroll
  for (int q = 2 * P; q < 2 * P + 2; ++q) {
    const unsigned col = (unsigned)__builtin_amdgcn_ds_bpermute(src + 16 * q, col_reg);
    const ST *p = Y + (size_t)col * F + 4 * (lane & 15);
#pragma unroll
    for (int e = 0; e < FE; e += 4) {
      const float4 v = load4(p + 16 * e);
      y[q][e] = v.x, y[q][e + 1] = v.y, y[q][e + 2] = v.z, y[q][e + 3] = v.w;
    }
  }
}

// One pass over this wave's share of a row: acc (compact) = [its gramian rows] . v  +  [its tile entries] weights.
//   FIRST: v = x, weights c+ - (|c|-1) y.x, the dense part enters negated (_als.pyx:187-201)
//   else : weights (|c|-1) y.v (_als.pyx:214-222)
//   LAST : the tile registers (and weight-table slots) of pair P are re-filled with the next row's entries once the pair is done
template <int F, int NJ, bool FIRST, bool LAST, typename ST>
__device__ __forceinline__ void fused_pass(float (&y)[8][F / 16], float *cw, int cnt, const float (&v)[F / 64],
                                           float (&acc)[F / 64], const float *row, const float *vp, float *myvec, int lane,
                                           int cnt_nx, int &col_nx, float &c_nx, const ST *__restrict__ Y,
                                           const ST *__restrict__ x_next_row, float (&xn)[F / 64],
                                           const int32_t *__restrict__ indices, const float *__restrict__ data, int k0_nx2,
                                           int end_nx2) {
  constexpr int FE = F / 16, FC = F / 64;
  if constexpr (LAST) {
    // The staged entries were requested a row ago.  Passing them through an opaque copy makes the compiler wait for them
    // HERE, once, while nothing else is in flight; without it every use inside the pass would wait for "all loads so far"
    // (vmcnt(0): its counter bookkeeping does not survive the branches of the pass) -- i.e. for the rolling gathers of
    // the pairs before.
    col_nx = opaque(col_nx);
    c_nx = __int_as_float(opaque(__float_as_int(c_nx)));
  }
  {
    const int ln = opaque(lane);
#pragma unroll
    for (int cc = 0; cc < FC; ++cc) myvec[QL<F>::cfactor(ln, cc)] = v[cc];  // wave-private copy for the p_j reads: no barrier
  }
  float ve[FE], ae[FE];
  expand_vector<F>(v, ve);
#pragma unroll
  for (int e = 0; e < FE; ++e) ae[e] = 0.f;
  DenseTicks<F, NJ> dt;
  auto partial = [&](int q) {
    float lo = 0.f, hi = 0.f;
#pragma unroll
    for (int e = 0; e < FE; e += 2) {
      lo = fmaf(y[q][e], ve[e], lo);
      hi = fmaf(y[q][e + 1], ve[e + 1], hi);
    }
    return lo + hi;
  };
  const float *cwg = cw + (opaque(lane) >> 4);  // this group's entries: t = 4 q + g
  auto axpy = [&](int q, float d, float cm1, float cp) {
    const float w = FIRST ? fmaf(cm1, d, -cp) : cm1 * d;  // the whole first pass is accumulated negated
#pragma unroll
    for (int e = 0; e < FE; ++e) ae[e] = fmaf(w, y[q][e], ae[e]);
  };
  static_for<4>([&](auto Pc) {
    constexpr int P = decltype(Pc)::value;
    if (8 * P < cnt) {  // wave-uniform
      dt.template issue<4 * P>(row, vp);
      const float cm1_0 = cwg[8 * P], cm1_1 = cwg[8 * P + 4];
      float cp_0 = 0.f, cp_1 = 0.f;
      if constexpr (FIRST) cp_0 = cwg[32 + 8 * P], cp_1 = cwg[32 + 8 * P + 4];
      __builtin_amdgcn_sched_barrier(0);
      float d0 = partial(2 * P);
      __builtin_amdgcn_sched_barrier(0);
      dt.template consume<4 * P>(ae);
      dt.template issue<4 * P + 1>(row, vp);
      __builtin_amdgcn_sched_barrier(0);
      float d1 = partial(2 * P + 1);
      d0 += dpp_mov<0x128>(d0), d1 += dpp_mov<0x128>(d1);  // row_ror:8
      d0 += dpp_mov<0x124>(d0), d1 += dpp_mov<0x124>(d1);  // row_ror:4
      d0 += dpp_mov<0x122>(d0), d1 += dpp_mov<0x122>(d1);  // row_ror:2
      d0 += dpp_mov<0x121>(d0), d1 += dpp_mov<0x121>(d1);  // row_ror:1
      __builtin_amdgcn_sched_barrier(0);
      dt.template consume<4 * P + 1>(ae);
      dt.template issue<4 * P + 2>(row, vp);
      __builtin_amdgcn_sched_barrier(0);
      axpy(2 * P, d0, cm1_0, cp_0);
      __builtin_amdgcn_sched_barrier(0);
      dt.template consume<4 * P + 2>(ae);
      dt.template issue<4 * P + 3>(row, vp);
      __builtin_amdgcn_sched_barrier(0);
      axpy(2 * P + 1, d1, cm1_1, cp_1);
      __builtin_amdgcn_sched_barrier(0);
      dt.template consume<4 * P + 3>(ae);
    } else {  // no entries left: the remaining gramian rows
      static_for<4>([&](auto Kc) {
        constexpr int K = 4 * P + decltype(Kc)::value;
        dt.template issue<K>(row, vp);
        __builtin_amdgcn_sched_barrier(0);
        dt.template consume<K>(ae);
      });
    }
    if constexpr (LAST) {
      if (8 * P < cnt_nx) gather_pair<F, P>(y, cw, col_nx, c_nx, cnt_nx, Y, lane);
    }
    __builtin_amdgcn_sched_barrier(0);
  });
  reduce_expanded<F>(ae, acc);
  if constexpr (LAST) {
    // the staged entries are used up: stage those of the row after the next, THEN request the next row's iterate -- loads
    // complete in order, and the iterate is the first thing the next row waits for
    fetch_entries(indices, data, opaque(lane), k0_nx2, end_nx2, col_nx, c_nx);
    load_compact<F>(x_next_row, opaque(lane), xn);
  }
  if constexpr (FIRST) {
#pragma unroll
    for (int cc = 0; cc < FC; ++cc) acc[cc] = -acc[cc];
  }
}

template <int F, int WPR, int BLOCK, typename ST>
__global__ __launch_bounds__(BLOCK, 4) void als_cg_qfteam_kernel(const int32_t *__restrict__ order, int first, int count,
                                                                 const int32_t *__restrict__ indptr,
                                                                 const int32_t *__restrict__ indices,
                                                                 const float *__restrict__ data, ST *__restrict__ X,
                                                                 const ST *__restrict__ Y, const float *__restrict__ A0,
                                                                 int cg_steps) {
  constexpr int FC = F / 64, FE = F / 16, T = 32, WAVES = BLOCK / 64, TEAMS = WAVES / WPR, NJ = F / WPR / 4;
  constexpr bool ROLL = std::is_same<ST, float>::value;  // fp16 storage converts at the load: no rolling gather
  static_assert(WPR <= WAVES && (F / WPR) % 4 == 0, "team width");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *A0s = smem;                              // [F][F]
  float *scratch = A0s + (size_t)F * F;           // [2][WAVES][F]  partial vectors of the combine, double-buffered
  float *vecs = scratch + (size_t)2 * WAVES * F;  // [WAVES][F]  wave-private copy of the operand vector (natural order)
  float *cws = vecs + (size_t)WAVES * F;          // [WAVES][64]  per-entry weights |c| - 1 and c+ of the resident tile (gather_pair)
  unsigned *arrivals = reinterpret_cast<unsigned *>(cws + (size_t)WAVES * 64);  // [TEAMS] monotonic team-barrier counters
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int team = wave / WPR, sub = wave % WPR;
  for (int e = threadIdx.x; e < F * F; e += BLOCK) A0s[e] = A0[e];
  if (threadIdx.x < TEAMS) arrivals[threadIdx.x] = 0u;
  __syncthreads();  // the only workgroup-wide barrier: from here on the teams run their rows independently
  const int j_begin = F * sub / WPR;
  float *myvec = vecs + (size_t)wave * F;
  float *cw = cws + (size_t)wave * 64;

  unsigned arrive_target = 0;
  auto team_sync = [&]() {  // als_cg_q.hip: a team meets on a monotonic LDS counter, teams stay independent
    if constexpr (WPR == WAVES) {
      __syncthreads();
    } else {
      arrive_target += WPR;
      if (lane == 0) __hip_atomic_fetch_add(&arrivals[team], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      while (__builtin_amdgcn_readfirstlane(
                 __hip_atomic_load(&arrivals[team], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) < arrive_target)
        __builtin_amdgcn_s_sleep(1);
    }
  };
  int parity = 0;
  auto combine = [&](float (&acc)[FC]) {  // sum of the team's WPR partial vectors in a fixed order (als_cg_q.hip)
    if constexpr (WPR == 1) return;
    const int ln = opaque(lane);
    float *buf = scratch + (size_t)parity * WAVES * F;
    parity ^= 1;
#pragma unroll
    for (int c = 0; c < FC; ++c) buf[wave * F + QL<F>::cfactor(ln, c)] = acc[c];
    team_sync();
#pragma unroll
    for (int c = 0; c < FC; ++c) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < WPR; ++w) s += buf[(team * WPR + w) * F + QL<F>::cfactor(ln, c)];
      acc[c] = s;
    }
  };
  // LDS addresses of the dense part, re-derived per pass
  auto dense_ptrs = [&](const float *&row, const float *&vp, float *&mv) {
    const int ln = opaque(lane);
    const int g = ln >> 4;
    vp = myvec + j_begin + g;
    row = A0s + (size_t)(j_begin + g) * F + 4 * (ln & 15);
    mv = myvec;
  };

  // this team's rows: i = (blockIdx.x + k gridDim.x) TEAMS + team; rows past the end re-read the last row
  auto row_id = [&](int i) { return order[first + min(i, count - 1)]; };  // uniform address: scalar load
  const int i_step = gridDim.x * TEAMS, i_first = blockIdx.x * TEAMS + team;
  auto slice = [&](int rb, int re, int &k0, int &cnt) {  // even shares rounded up to whole 4-entry tile steps
    const int chunk = min(T, (((re - rb) + WPR - 1) / WPR + 3) & ~3);
    k0 = min(rb + chunk * sub, re);
    cnt = min(chunk, re - k0);
  };
  // dependent loads per row: schedule entry -> row id -> nnz range -> entries -> factor rows; each stage runs one row further
  // ahead than the next: ids 4 rows, ranges 3, entries 2 (1 when the tile was not rolled in), factor rows 1 (rolled) or 0
  int id0 = row_id(i_first), id1 = row_id(i_first + i_step), id2 = row_id(i_first + 2 * i_step), id3 = row_id(i_first + 3 * i_step);
  int b0 = indptr[id0], e0 = indptr[id0 + 1], b1 = indptr[id1], e1 = indptr[id1 + 1], b2 = indptr[id2], e2 = indptr[id2 + 1];
  // ent_*: staged entries (one per lane) of the next row whose tile has to be gathered
  int ent_col, ent_cnt, k0;
  float ent_c;
  slice(b0, e0, k0, ent_cnt);
  fetch_entries(indices, data, opaque(lane), k0, max(k0 + ent_cnt, b0 + 1), ent_col, ent_c);
  bool tile_ready = false;  // the tile and the iterate of the row at the top of the loop body are already on their way
  int cnt = 0;
  float y[8][FE], x[FC];
  for (int i = i_first; i < count; i += i_step) {
    ST *xrow = X + (size_t)id0 * F;
    float r[FC], p[FC], Ap[FC];
    if (!tile_ready) {  // first row of the wave, or the previous row ended before its last pass: plain row start
      cnt = ent_cnt;
      ent_col = opaque(ent_col);  // one wait for the staged entries, before the gathers (see fused_pass)
      ent_c = __int_as_float(opaque(__float_as_int(ent_c)));
      static_for<4>([&](auto Pc) {
        constexpr int P = decltype(Pc)::value;
        if (8 * P < cnt) gather_pair<F, P>(y, cw, ent_col, ent_c, cnt, Y, lane);
      });
      slice(b1, e1, k0, ent_cnt);
      fetch_entries(indices, data, opaque(lane), k0, max(k0 + ent_cnt, b1 + 1), ent_col, ent_c);
      load_compact<F>(xrow, opaque(lane), x);  // last: loads complete in order and the first pass starts with x
    }
    // ent_* now describe row i + i_step
    const float *row, *vp;
    float *mv;
    dense_ptrs(row, vp, mv);
    fused_pass<F, NJ, true, false, ST>(y, cw, cnt, x, r, row, vp, mv, lane, 0, ent_col, ent_c, Y, nullptr, x, nullptr, nullptr, 0, 0);
    // `x` is the loop-carried register pair the last pass loads the NEXT row's iterate into; this row's iterate moves on
    // as xc (copying the loaded value at the end of the row instead would wait for every gather issued before it)
    float xc[FC];
#pragma unroll
    for (int cc = 0; cc < FC; ++cc) xc[cc] = x[cc];
    combine(r);
#pragma unroll
    for (int cc = 0; cc < FC; ++cc) p[cc] = r[cc];
    float rsold = dot_compact<F>(r, r);
    bool active = rsold >= 1e-20f;  // else: x untouched (_als.pyx:206)
    const bool store = active && sub == 0;
    for (int it = 0; it + 1 < cg_steps && active; ++it) {
      dense_ptrs(row, vp, mv);
      fused_pass<F, NJ, false, false, ST>(y, cw, cnt, p, Ap, row, vp, mv, lane, 0, ent_col, ent_c, Y, nullptr, x, nullptr, nullptr, 0, 0);
      combine(Ap);
      // the operand's wave-private LDS copy is still in place: reading it back frees p's registers across the pass
      load_compact<F>(static_cast<const float *>(mv), opaque(lane), p);
      const float alpha = rsold / dot_compact<F>(p, Ap);
#pragma unroll
      for (int cc = 0; cc < FC; ++cc) {
        xc[cc] = fmaf(alpha, p[cc], xc[cc]);
        r[cc] = fmaf(-alpha, Ap[cc], r[cc]);
      }
      const float rsnew = dot_compact<F>(r, r);
      if (rsnew < 1e-20f) {
        active = false;  // the oracle breaks here (_als.pyx:235); the whole team takes the same branch
      } else {
        const float beta = rsnew / rsold;
#pragma unroll
        for (int cc = 0; cc < FC; ++cc) p[cc] = fmaf(beta, p[cc], r[cc]);
        rsold = rsnew;
      }
    }
    // The last step stands outside the loop: its pass rolls the next row's tile in, and only its x update is evaluated --
    // the oracle's r, rsnew and p of the last step (_als.pyx:226-241) are never read again.
    const bool rolled = ROLL && active && cg_steps > 0;
    if (active && cg_steps > 0) {
      dense_ptrs(row, vp, mv);
      if constexpr (ROLL) {  // the tile of row i + i_step rolls in; the entries of row i + 2 i_step get staged
        int k2, cnt2;
        slice(b2, e2, k2, cnt2);
        if (i + i_step >= count) ent_cnt = 0;  // no next row (the schedule index is clamped): nothing to gather
        fused_pass<F, NJ, false, true, ST>(y, cw, cnt, p, Ap, row, vp, mv, lane, ent_cnt, ent_col, ent_c, Y, X + (size_t)id1 * F, x,
                                           indices, data, k2, max(k2 + cnt2, b2 + 1));
        cnt = ent_cnt;
        ent_cnt = cnt2;
      } else
        fused_pass<F, NJ, false, false, ST>(y, cw, cnt, p, Ap, row, vp, mv, lane, 0, ent_col, ent_c, Y, nullptr, x, nullptr, nullptr, 0, 0);
      combine(Ap);
      load_compact<F>(static_cast<const float *>(mv), opaque(lane), p);
      const float alpha = rsold / dot_compact<F>(p, Ap);
#pragma unroll
      for (int cc = 0; cc < FC; ++cc) xc[cc] = fmaf(alpha, p[cc], xc[cc]);
    }
    if (store) store_compact<F>(xrow, opaque(lane), xc);
    tile_ready = rolled;
    id0 = id1, id1 = id2, id2 = id3, id3 = row_id(i + 4 * i_step);
    b0 = b1, e0 = e1, b1 = b2, e1 = e2, b2 = indptr[id2], e2 = indptr[id2 + 1];
  }
}

template <int F, int WPR, int BLOCK, typename T>
static void launch_qfteam(const imp_csr *C, int first, int count, T *X, const T *Y, const float *A0, int cg_steps,
                          const char *name) {
  if (count <= 0) return;
  constexpr int WAVES = BLOCK / 64, TEAMS = WAVES / WPR;
  size_t lds = ((size_t)F * F + 3 * WAVES * F + 64 * WAVES + TEAMS) * sizeof(float);
  auto kern = als_cg_qfteam_kernel<F, WPR, BLOCK, T>;
  IMP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int per_cu = (int)std::max<size_t>(1, std::min<size_t>(2048 / BLOCK, (160 * 1024) / lds));
  constexpr int kBaseOversub = WPR <= 4 ? 4 : (WPR == 8 ? 2 : 1);  // als_cg_q.hip launch_qteam
  int grid = std::min((count + TEAMS - 1) / TEAMS, ctx().num_cus * per_cu * std::max(kBaseOversub, ctx().oversub));
  IMP_PROF(name);
  kern<<<grid, BLOCK, lds, stream()>>>(C->order.data(), first, count, C->indptr.data(), C->indices.data(), C->data.data(), X, Y,
                                      A0, cg_steps);
  IMP_CHECK_HIP(hipGetLastError());
}

// width: 1 (f = 64 short rows), 2, 4, 8, 16
template <typename T>
void launch_team_fused(const imp_csr *C, int f, int width, int first, int count, T *X, const T *Y, const float *A0, int cg_steps,
                       const char *name) {
  auto run = [&](auto Fc) {
    constexpr int F = decltype(Fc)::value;
    switch (width) {
      case 16: launch_qfteam<F, 16, 1024, T>(C, first, count, X, Y, A0, cg_steps, name); break;
      case 8: launch_qfteam<F, 8, 512, T>(C, first, count, X, Y, A0, cg_steps, name); break;
      case 4: launch_qfteam<F, 4, 512, T>(C, first, count, X, Y, A0, cg_steps, name); break;
      case 2: launch_qfteam<F, 2, 512, T>(C, first, count, X, Y, A0, cg_steps, name); break;
      case 1:
        if constexpr (F == 64) launch_qfteam<F, 1, 512, T>(C, first, count, X, Y, A0, cg_steps, name);
        else throw std::invalid_argument("launch_team_fused: one wave per row needs f = 64");
        break;
      default: throw std::invalid_argument("launch_team_fused: team width");
    }
  };
  if (f == 128) run(idx_t<128>{});
  else if (f == 64) run(idx_t<64>{});
  else throw std::invalid_argument("launch_team_fused: f must be 64 or 128");
}
template void launch_team_fused<float>(const imp_csr *, int, int, int, int, float *, const float *, const float *, int, const char *);
template void launch_team_fused<__half>(const imp_csr *, int, int, int, int, __half *, const __half *, const float *, int,
                                        const char *);

}  // namespace imp
