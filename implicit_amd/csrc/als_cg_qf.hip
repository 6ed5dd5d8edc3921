// K1f: the mid-row CG half sweep (a team of WPR wavefronts per row, whole row resident in registers; als_cg_q.hip has the
// tile design) rebuilt around the resource that actually bounds it -- round 3.
//
// Arithmetic contract: the oracle's CG (implicit/cpu/_als.pyx:152-248).
//
// What the round-2 team kernels spent (profiles/micro/valu_rate.hip, profiles/r04_micro_valu_rate.txt; HISTORY.md section 4.1 has
// the full picture): saturated, a SIMD retires a plain vector instruction every 2.3 cycles and a packed FMA, a DPP form or an
// SGPR-operand form every 4.1-4.4; one wave alone issues only every 5.5-6 cycles.  The round-2 kernels executed 1.03 G vector
// instructions per C3 iteration for the mid-row classes, and only ~55 % of them were the FMAs of the dense part and of the tile;
// the rest was per-wavefront bookkeeping, REPLICATED in every wavefront of a team:
//   - the CG scalars (two wave-wide dot reductions, two IEEE divisions, the x / r / p updates) -- every wave of a team
//     did the identical arithmetic on identical bits;
//   - the operand's expansion from the compact to the quarter layout (6 v_permlane swaps + 12 register copies per pass)
//     and the sum of the team's partial vectors in every wave.
// (A first version of the micro-benchmark, run on a box in a low-power state, read 7.5 cycles for everything and led to the
// conclusion "100 % issue bound"; the instruction count was worth cutting anyway: 808 M now.)
// This kernel gives that work to ONE wavefront per team (the leader, sub == 0) and turns the rest into LDS traffic, which
// has issue slots of its own:
//   * the leader alone sums the team's partial vectors, does the CG update and PUBLISHES the next operand in LDS (natural
//     factor order) together with a go / last / stop word; the other waves wait on the team's generation counter (an idle
//     wave costs no issue slots -- that is the point) and read the operand back already expanded: two ds_read_b128 per
//     lane, no swaps.  Two counters per team (arrivals A, generation B), no workgroup barrier after the prologue;
//   * a / b by v_rcp_f32 (1 ulp) instead of the 12-instruction IEEE sequence; the last CG step only updates x;
//   * the dots of a pair of tile steps are reduced together (5 DPP adds for two values instead of 8) and the weight is
//     applied straight from the lane that holds the total (row_newbcast operand of the multiply);
//   * per-entry weights |c| - 1 and c+ live in an LDS table written once per row (gather_pair).
// Kept from the first round-3 version: fused passes (the gramian rows of a pass are dealt to 16 ticks whose LDS reads are
// issued before a tile half-step and consumed after it) and the rolling gather (the last pass of a row re-fills each pair
// of tile registers with the next row's entries as soon as the pair is done; metadata runs ids 4 rows ahead, nnz ranges
// 3, entries 2).
#include <type_traits>

#include "als_qf_common.h"
#include "common.h"

namespace imp {

// Entries of tile steps 2 P and 2 P + 1.  The staged registers hold entry min(l, cnt - 1) of the wave's slice in lanes l and
// l + 32 (fetch_entries): the gather addresses travel by ds_bpermute (entry t = 4 q + g -> the 16 lanes of group g), the two
// weights every pass derives from a confidence -- |c| - 1 and c+ = max(c, 0), both 0 for the padding entries -- are written
// ONCE to a wave-private LDS table by the lanes that hold the entries (cw[t] = |c| - 1, cw[32 + t] = c+) and read back
// per step as a group-wide broadcast: 8 registers less than carrying them, and no per-pass abs / max.
template <int F, int P, typename ST>
__device__ __forceinline__ void gather_pair(f32x2 (&y)[8][F / 32], float *cw, int col_reg, float c_reg, int cnt,
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
      y[q][e / 2] = f32x2{v.x, v.y}, y[q][e / 2 + 1] = f32x2{v.z, v.w};
    }
  }
}

// One pass over this wave's share of a row: acc (compact) = [its gramian rows] . v  +  [its tile entries] weights, v being
// the operand the team's leader published in LDS (`vt`, natural factor order).
//   FIRST: v = x, weights c+ - (|c|-1) y.x, the dense part enters negated (_als.pyx:187-201): the pass accumulates
//          A0 x - sum w y and the caller takes the sum of the team's partials with a minus sign
//   else : weights (|c|-1) y.v (_als.pyx:214-222)
//   LAST : the tile registers (and weight-table slots) of pair P are re-filled with the next row's entries once the pair is done
template <int F, int NJ, bool FIRST, bool LAST, typename ST>
__device__ __forceinline__ void fused_pass(f32x2 (&y)[8][F / 32], float *cw, int cnt, const float *vt, int j_begin,
                                           const float *A0s, float (&acc)[F / 64], int lane, int cnt_nx, int &col_nx,
                                           float &c_nx, const ST *__restrict__ Y, const int32_t *__restrict__ indices,
                                           const float *__restrict__ data, int k0_nx2, int end_nx2) {
  constexpr int FE = F / 16, H = FE / 2;
  if constexpr (LAST) {
    // The staged entries were requested a row ago.  Passing them through an opaque copy makes the compiler wait for them
    // HERE, once, while nothing else is in flight; without it every use inside the pass would wait for "all loads so far"
    // (vmcnt(0): its counter bookkeeping does not survive the branches of the pass) -- i.e. for the rolling gathers of
    // the pairs before.
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
  DenseTicks<F, NJ> dt;
  auto partial = [&](int q) {  // this lane's share of y_q . v: even / odd slots in the two halves of one packed accumulator
    f32x2 s = y[q][0] * ve[0];
#pragma unroll
    for (int h = 1; h < H; ++h) s = __builtin_elementwise_fma(y[q][h], ve[h], s);
    return s.x + s.y;
  };
  auto axpy = [&](int q, float w) {
    const f32x2 w2 = {w, w};
#pragma unroll
    for (int h = 0; h < H; ++h) ae[h] = __builtin_elementwise_fma(w2, y[q][h], ae[h]);
  };
  static_for<4>([&](auto Pc) {
    constexpr int P = decltype(Pc)::value;
    if (8 * P < cnt) {  // wave-uniform
      dt.template issue<4 * P>(row, vp);
      const float cm1_0 = cwg[8 * P], cm1_1 = cwg[8 * P + 4];
      float cp_0 = 0.f, cp_1 = 0.f;
      if constexpr (FIRST) cp_0 = cwg[32 + 8 * P], cp_1 = cwg[32 + 8 * P + 4];
      __builtin_amdgcn_sched_barrier(0);
      const float d0 = partial(2 * P);
      __builtin_amdgcn_sched_barrier(0);
      dt.template consume<4 * P>(ae);
      dt.template issue<4 * P + 1>(row, vp);
      __builtin_amdgcn_sched_barrier(0);
      const float d1 = partial(2 * P + 1);
      // no fence between the reduction and the packed FMAs of the tick: they fill the wait states of its dependent DPP chain
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
      // (the empty statement keeps the two branches from starting alike: the compiler otherwise hoists "read, wait,
      // consume" of the first tick above the branch and the tick's LDS latency is exposed again)
      asm volatile("" ::: "memory");
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
  float aes[FE];
#pragma unroll
  for (int h = 0; h < H; ++h) aes[2 * h] = ae[h].x, aes[2 * h + 1] = ae[h].y;
  reduce_expanded<F>(aes, acc);
  if constexpr (LAST) {
    // the staged entries are used up: stage those of the row after the next (loads complete in order: before the leader's
    // request for the next row's iterate, which is the first thing the next row waits for)
    fetch_entries(indices, data, opaque(lane), k0_nx2, end_nx2, col_nx, c_nx);
  }
}

template <int F, int WPR, int BLOCK, typename ST>
__global__ __launch_bounds__(BLOCK, F == 64 ? 8 : 4) void als_cg_qfteam_kernel(const int32_t *__restrict__ order, int first, int count,
                                                                 const int32_t *__restrict__ indptr,
                                                                 const int32_t *__restrict__ indices,
                                                                 const float *__restrict__ data, ST *__restrict__ X,
                                                                 const ST *__restrict__ Y, const float *__restrict__ A0,
                                                                 int cg_steps) {
  constexpr int FC = F / 64, FE = F / 16, T = 32, WAVES = BLOCK / 64, TEAMS = WAVES / WPR, NJ = F / WPR / 4;
  constexpr bool ROLL = std::is_same<ST, float>::value;  // fp16 storage converts at the load: no rolling gather
  static_assert(WPR <= WAVES && (F / WPR) % 4 == 0, "team width");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *A0s = smem;                            // [F][F]
  float *parts = A0s + (size_t)F * F;           // [WAVES][F]  partial vectors of the waves (compact slots at their natural index)
  float *vts = parts + (size_t)WAVES * F;       // [TEAMS][F]  the operand the leader published (natural factor order)
  float *cws = vts + (size_t)TEAMS * F;         // [WAVES][64]  per-entry weights |c| - 1 and c+ of the resident tile (gather_pair)
  unsigned *ctl = reinterpret_cast<unsigned *>(cws + (size_t)WAVES * 64);  // [TEAMS][4]  arrivals A, generation B, control words
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int team = wave / WPR, sub = wave % WPR;
  const bool leader = sub == 0;
  for (int e = threadIdx.x; e < F * F; e += BLOCK) A0s[e] = A0[e];
  if (threadIdx.x < 4 * TEAMS) ctl[threadIdx.x] = 0u;
  __syncthreads();  // the only workgroup-wide barrier: from here on the teams run their rows independently
  const int j_begin = F * sub / WPR;
  float *vt = vts + (size_t)team * F;
  float *cw = cws + (size_t)wave * 64;
  unsigned *arrivals = ctl + 4 * team, *generation = arrivals + 1, *words = arrivals + 2;

  // ---- team protocol ---------------------------------------------------------------------------------------------------
  // worker (every wave, the leader included): wait for generation g -> read word + operand -> pass -> partial to LDS -> arrive
  // leader: wait for WPR arrivals -> sum the partials in wave order -> CG update -> operand + word to LDS -> generation + 1
  // Both counters are monotonic; a wave's LDS operations execute in order, so a partial is in place before its arrival is
  // counted and an operand before its generation is.  The operand slot and the partial slots are single-buffered: the
  // leader overwrites the operand only after all WPR arrivals of the pass that read it, and a wave overwrites its partial
  // only after the next generation, which the leader publishes after having summed it.  The control word has two slots
  // (generation parity): a "stop" generation expects no arrivals, so the leader may publish the next row's first generation
  // before a slow wave has read the stop word -- but never a second one, which needs that wave's arrival.
  unsigned gen = 0, pub = 0, arr_target = 0;
  // LDS byte offsets (the low half of a flat LDS address is the offset inside the workgroup's allocation)
  auto lds_off = [](const void *ptr) { return (unsigned)(size_t)ptr; };
  const unsigned arrivals_off = lds_off(arrivals), generation_off = lds_off(generation), words_off = lds_off(words);
  // the one lane-derived value that stays in a register for the whole kernel: byte offset of this lane's compact slots
  // inside a natural-order vector (the other lane-derived addresses are rebuilt where they are used)
  const unsigned cf4 = 4u * (unsigned)QL<F>::cfactor(lane, 0);
  // The counters are bumped with a bare ds_add_u32 from lane 0: the LDS executes a wave's operations in order, so the
  // partial vector / operand written just before is in place when the counter moves -- no release fence (s_waitcnt), and
  // none of the lane-counting code the compiler wraps around an atomic add inside a divergent branch.
  auto publish = [&](unsigned w) {  // leader
    ++pub;
    if (lane == 0)
      asm volatile("ds_write_b32 %0, %1\n\tds_add_u32 %2, %3" ::"v"(words_off + 4u * (pub & 1u)), "v"(w), "v"(generation_off), "v"(1u)
                   : "memory");
    if constexpr (IMP_TEAM_LEADER_PRIO > 0) __builtin_amdgcn_s_setprio(0);
  };
  auto poll = [&](unsigned off) {  // one ds_read_b32 of a counter, made wave-uniform
    typedef __attribute__((address_space(3))) volatile unsigned lds_word;
    return (unsigned)__builtin_amdgcn_readfirstlane(*(lds_word *)(size_t)off);
  };
  auto await_operand = [&]() -> unsigned {  // every wave; returns the control word
    ++gen;
    if constexpr (WPR > 1) {
      // every poll costs two vector-issue slots (address + readfirstlane): the first nap covers most of the leader's update
      if (poll(generation_off) < gen) {
        __builtin_amdgcn_s_sleep(IMP_TEAM_NAP_FIRST);
        while (poll(generation_off) < gen) __builtin_amdgcn_s_sleep(IMP_TEAM_NAP_NEXT);
      }
    }
    return poll(words_off + 4u * (gen & 1u));
  };
  auto arrive = [&](const float (&acc)[FC]) {  // every wave: partial vector to LDS, then count the arrival
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
    if constexpr (IMP_TEAM_LEADER_PRIO > 0) __builtin_amdgcn_s_setprio(IMP_TEAM_LEADER_PRIO);
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
  auto put_operand = [&](const float (&v)[FC]) {  // leader: compact -> natural order in the team's operand slot
    float *slot = operand_slot();
    if constexpr (FC == 2) *reinterpret_cast<float2 *>(slot) = make_float2(v[0], v[1]);
    else slot[0] = v[0];
  };
  auto get_operand = [&](float (&v)[FC]) {  // leader: the operand is still in its slot -- no registers across the pass
    const float *slot = operand_slot();
    if constexpr (FC == 2) {
      const float2 t = *reinterpret_cast<const float2 *>(slot);
      v[0] = t.x, v[1] = t.y;
    } else {
      v[0] = slot[0];
    }
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
  // x is only meaningful between a load and the top of the next row; every other path overwrites it, so that the compiler
  // does not carry (and spill) the old value across the passes
  auto kill = [](float (&v)[FC]) {
#pragma unroll
    for (int cc = 0; cc < FC; ++cc) v[cc] = 0.f;
  };
  bool tile_ready = false;  // the tile (and, in the leader, the iterate) of the row at the top of the body are on their way
  int cnt = 0;
  f32x2 y[8][FE / 2];
  float x[FC];  // x: the leader's loop-carried iterate registers (the last step loads the NEXT row's into them)
#pragma unroll
  for (int cc = 0; cc < FC; ++cc) x[cc] = 0.f;
  for (int i = i_first; i < count; i += i_step) {
    ST *xrow = X + (size_t)id0 * F;
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
      if (leader) load_compact<F>(xrow, opaque(lane), x);  // last: loads complete in order and the row starts with x
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
      for (int cc = 0; cc < FC; ++cc) xc[cc] = x[cc];  // this row's iterate moves on as xc; x is re-loaded for the next row
      publish(kGo);
    }
    unsigned w = await_operand();
    {
      float acc[FC];
      fused_pass<F, NJ, true, false, ST>(y, cw, cnt, vt, j_begin, A0s, acc, lane, 0, ent_col, ent_c, Y, nullptr, nullptr, 0, 0);
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
      fused_pass<F, NJ, false, false, ST>(y, cw, cnt, vt, j_begin, A0s, acc, lane, 0, ent_col, ent_c, Y, nullptr, nullptr, 0, 0);
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
    // The last step stands outside the loop (the compiler must see that nothing of the row follows it): its pass rolls
    // the next row's tile in, and only its x update is evaluated -- the oracle's r, rsnew and p of the last step
    // (_als.pyx:226-241) are never read again.
    const bool rolled = ROLL && (w & kGo) != 0u;
    if (w & kGo) {
      float acc[FC];
      if constexpr (ROLL) {  // the tile of row i + i_step rolls in; the entries of row i + 2 i_step get staged
        int k2, cnt2;
        slice(b2, e2, k2, cnt2);
        if (i + i_step >= count) ent_cnt = 0;  // no next row (the schedule index is clamped): nothing to gather
        fused_pass<F, NJ, false, true, ST>(y, cw, cnt, vt, j_begin, A0s, acc, lane, ent_cnt, ent_col, ent_c, Y, indices, data, k2,
                                           max(k2 + cnt2, b2 + 1));
        cnt = ent_cnt;
        ent_cnt = cnt2;
        if (leader) load_compact<F>(X + (size_t)id1 * F, opaque(lane), x);  // the next row's iterate, into the carried registers
        else kill(x);
      } else {
        kill(x);
        fused_pass<F, NJ, false, false, ST>(y, cw, cnt, vt, j_begin, A0s, acc, lane, 0, ent_col, ent_c, Y, nullptr, nullptr, 0, 0);
      }
      arrive(acc);
      if (leader) {
        collect(Ap);
        get_operand(p);
        const float alpha = rsold * __builtin_amdgcn_rcpf(dot_compact<F>(p, Ap));
#pragma unroll
        for (int cc = 0; cc < FC; ++cc) xc[cc] = fmaf(alpha, p[cc], xc[cc]);
        publish(0u);
      }
      (void)await_operand();  // the stop generation: keeps every wave's count in step with the leader's
    } else {
      kill(x);
    }
    if (leader && store) store_compact<F>(xrow, opaque(lane), xc);
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
  size_t lds = ((size_t)F * F + (size_t)WAVES * F + (size_t)TEAMS * F + 64 * WAVES + 4 * TEAMS) * sizeof(float);
  auto kern = als_cg_qfteam_kernel<F, WPR, BLOCK, T>;
  IMP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int per_cu = (int)std::max<size_t>(1, std::min<size_t>(2048 / BLOCK, (160 * 1024) / lds));
  // als_cg_q.hip launch_qteam; the 16-wave team at f = 64 (16 KB of gramian to stage, two workgroups per CU) takes 2 as well:
  // configs[1]-shaped CG 1.47 -> 1.42 ms, and a fixed share on a contended CU is what took seconds once (HISTORY.md section 6)
  constexpr int kBaseOversub = WPR <= 4 ? 4 : (WPR == 8 ? 2 : (F == 64 ? 2 : 1));
  int grid = std::min((count + TEAMS - 1) / TEAMS, ctx().num_cus * per_cu * std::max(kBaseOversub, ctx().oversub));
  IMP_PROF(name);
  kern<<<grid, BLOCK, lds, stream()>>>(C->order.data(), first, count, C->indptr.data(), C->indices.data(), C->data.data(), X, Y,
                                      A0, cg_steps);
  IMP_CHECK_HIP(hipGetLastError());
}

// ---- short rows (<= 32 nnz): one wave per row, 16 rows per workgroup in lock step, the dense part as ONE fp32 MFMA product --
// (als_cg_q.hip als_cg_qgroup_kernel has the product's layout: waves publish their operand in LDS, the 16-factor output
// tiles x K-slices of A0 . P^T are dealt to the 16 waves, results return through LDS.)  Round 3 on top of it:
//   * the operand a wave has just published is read back EXPANDED (two ds_read_b128) instead of 6 permlane swaps;
//   * weight table, pair-wise DPP reduction, v_rcp divisions, x-only last step as in the team kernel above;
//   * rolling gather: a workgroup holds its CU alone (96 KB of LDS), so while its 16 waves waited for the gathers of a
//     new group of rows the CU did nothing; the last pass of a group now re-fills each pair of tile registers with the next
//     group's entries as soon as the pair is done.
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int F> struct QFGroupCfg {
  static constexpr int LD = F + 8;          // A0 / P / Out row stride in LDS (conflict-free b128 fragment reads)
  static constexpr int NT = F / 16;         // 16-factor output tiles
  static constexpr int KH = 16 / NT;        // K-slices so that NT * KH == 16 waves
  static constexpr int KB = (F / 16) / KH;  // 16-factor k-blocks per wave
  static constexpr size_t lds_floats = (size_t)F * LD + 16 * LD + (size_t)KH * 16 * LD + 16 * 64;
  // BF3 form (f = 128): the gramian and the operands as three bf16 terms each (hi + mid + lo = the fp32 value to 2^-24),
  // rows of LDB bf16; the fp32 gramian image is not kept
  static constexpr int LDB = F + 8;
  static constexpr size_t lds_bytes_bf3 = (size_t)3 * F * LDB * 2 + (size_t)3 * 16 * LDB * 2 +
                                          ((size_t)16 * LD + (size_t)KH * 16 * LD + 16 * 64) * sizeof(float);
};
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
// x = hi + mid + lo + O(2^-24 |x|): every term the nearest bf16 of what is left (the subtractions are exact in fp32)
__device__ __forceinline__ void split_bf16(float x, __bf16 &hi, __bf16 &mid, __bf16 &lo) {
  hi = (__bf16)x;
  const float r1 = x - (float)hi;
  mid = (__bf16)r1;
  lo = (__bf16)(r1 - (float)mid);
}

// tile part of a pass: acc (compact) = sum over the resident entries of w y, operand read expanded from `vrow` (natural order)
//   FIRST: w = c+ - (|c|-1) y.x   else: w = (|c|-1) y.v
template <int F, bool FIRST, bool LAST, typename ST>
__device__ __forceinline__ void tile_pass(f32x2 (&y)[8][F / 32], float *cw, int cnt, const float *vrow, float (&acc)[F / 64], int lane,
                                          int cnt_nx, int &col_nx, float &c_nx, const ST *__restrict__ Y) {
  constexpr int FE = F / 16, H = FE / 2;
  if constexpr (LAST) {  // one wait for the staged entries, before any rolling gather (fused_pass)
    col_nx = opaque(col_nx);
    c_nx = __int_as_float(opaque(__float_as_int(c_nx)));
  }
  f32x2 ve[H], ae[H];
  const float *cwg;
  {
    const int ln = opaque(lane);
    const int g = ln >> 4, m = ln & 15;
#pragma unroll
    for (int e = 0; e < FE; e += 4) {
      const float4 t = *reinterpret_cast<const float4 *>(vrow + 16 * e + 4 * m);
      ve[e / 2] = f32x2{t.x, t.y}, ve[e / 2 + 1] = f32x2{t.z, t.w};
    }
    cwg = cw + g;
  }
#pragma unroll
  for (int h = 0; h < H; ++h) ae[h] = f32x2{0.f, 0.f};
  auto partial = [&](int q) {
    f32x2 s = y[q][0] * ve[0];
#pragma unroll
    for (int h = 1; h < H; ++h) s = __builtin_elementwise_fma(y[q][h], ve[h], s);
    return s.x + s.y;
  };
  auto axpy = [&](int q, float w) {
    const f32x2 w2 = {w, w};
#pragma unroll
    for (int h = 0; h < H; ++h) ae[h] = __builtin_elementwise_fma(w2, y[q][h], ae[h]);
  };
  static_for<4>([&](auto Pc) {
    constexpr int P = decltype(Pc)::value;
    if (8 * P < cnt) {  // wave-uniform
      const float cm1_0 = cwg[8 * P], cm1_1 = cwg[8 * P + 4];
      float cp_0 = 0.f, cp_1 = 0.f;
      if constexpr (FIRST) cp_0 = cwg[32 + 8 * P], cp_1 = cwg[32 + 8 * P + 4];
      const float u = reduce_pair(partial(2 * P), partial(2 * P + 1));
      const float w0 = FIRST ? fmaf(-cm1_0, row_bcast_from<0>(u), cp_0) : cm1_0 * row_bcast_from<0>(u);
      const float w1 = FIRST ? fmaf(-cm1_1, row_bcast_from<8>(u), cp_1) : cm1_1 * row_bcast_from<8>(u);
      axpy(2 * P, w0);
      axpy(2 * P + 1, w1);
    }
    if constexpr (LAST) {
      if (8 * P < cnt_nx) gather_pair<F, P>(y, cw, col_nx, c_nx, cnt_nx, Y, lane);
    }
  });
  float aes[FE];
#pragma unroll
  for (int h = 0; h < H; ++h) aes[2 * h] = ae[h].x, aes[2 * h + 1] = ae[h].y;
  reduce_expanded<F>(aes, acc);
}

template <int F, bool STAGGER, bool BF3, typename ST>
__global__ __launch_bounds__(1024) void als_cg_qfgroup_kernel(const int32_t *__restrict__ order, int first, int count,
                                                              const int32_t *__restrict__ indptr,
                                                              const int32_t *__restrict__ indices,
                                                              const float *__restrict__ data, ST *__restrict__ X,
                                                              const ST *__restrict__ Y, const float *__restrict__ A0, int cg_steps) {
  using Cfg = QFGroupCfg<F>;
  constexpr int FC = F / 64, FE = F / 16, LD = Cfg::LD;
  constexpr bool ROLL = std::is_same<ST, float>::value;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int LDB = Cfg::LDB;
  // fp32 form: [A0s F x LD][Ps][Outs][cws]      BF3 form: [A0 hi | mid | lo, bf16 F x LDB each][Pb hi | mid | lo, 16 x LDB][Ps][Outs][cws]
  __bf16 *A0b = reinterpret_cast<__bf16 *>(smem);       // BF3: term t at A0b + t F LDB
  __bf16 *Pb = A0b + (size_t)3 * F * LDB;               // BF3: term t at Pb + t 16 LDB
  float *A0s = smem;                                   // fp32 form: [F][LD]
  float *Ps = BF3 ? reinterpret_cast<float *>(Pb + (size_t)3 * 16 * LDB) : A0s + (size_t)F * LD;  // [16][LD] operands (natural order)
  float *Outs = Ps + 16 * LD;                          // [KH][16][LD]  K-slice partial products
  float *cws = Outs + (size_t)Cfg::KH * 16 * LD;       // [16][64]  per-entry weights (gather_pair)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int e = threadIdx.x; e < F * F; e += 1024) {
    int r = e / F, c = e - r * F;
    if constexpr (BF3) {
      __bf16 h, m, l;
      split_bf16(A0[e], h, m, l);
      A0b[r * LDB + c] = h, A0b[(size_t)F * LDB + r * LDB + c] = m, A0b[(size_t)2 * F * LDB + r * LDB + c] = l;
    } else {
      A0s[r * LD + c] = A0[e];
    }
  }
  __syncthreads();
  float *prow = Ps + (size_t)wave * LD;
  float *cw = cws + (size_t)wave * 64;
  const unsigned cf = (unsigned)QL<F>::cfactor(lane, 0);  // this lane's compact slots inside a natural-order vector

  // out (compact) = A0 . vec for this wave's row; every wave of the workgroup takes both barriers (inactive rows publish 0)
  auto publish = [&](const float (&vec)[FC], bool valid) {
    if constexpr (FC == 2) *reinterpret_cast<float2 *>(prow + cf) = valid ? make_float2(vec[0], vec[1]) : make_float2(0.f, 0.f);
    else prow[cf] = valid ? vec[0] : 0.f;
    if constexpr (BF3) {  // the same operand as three bf16 terms for the matrix cores (FC == 2: one packed pair per term)
      bf16x2 t[3];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        __bf16 h, m, l;
        split_bf16(valid ? vec[c] : 0.f, h, m, l);
        t[0][c] = h, t[1][c] = m, t[2][c] = l;
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) *reinterpret_cast<bf16x2 *>(Pb + (size_t)k * 16 * LDB + (size_t)wave * LDB + cf) = t[k];
    }
    __syncthreads();
  };
  auto product = [&]() {  // this wave's (output tile, K-slice) of A0 . P^T for the 16 rows -> Outs
    const int ti = wave % Cfg::NT, kh = wave / Cfg::NT;
    const int ln = opaque(lane);
    const int i = ln & 15, kq = ln >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if constexpr (BF3) {
      // fp32-equivalent product on the bf16 matrix cores: A0 = Ah + Am + Al, p = ph + pm + pl (each to 2^-24), and the six
      // partial products down to 2^-16 relative weight, smallest first, accumulated in fp32 -- 12 MFMAs of K = 32 per wave
      // and pass instead of 16 fp32 MFMAs of K = 4, at a quarter of the instruction time each, and on hardware the vector
      // pipe does not share (v_mfma_f32_16x16x4_f32 runs at the VECTOR rate and, measured, does not overlap with the tile
      // entries' packed FMAs; these do).  A and B fragments use the same (lane group, element) -> k assignment, which is all
      // the contraction needs.
      static_assert(!BF3 || Cfg::KB * 16 == 64, "BF3 product: 64 k per wave");
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int k0 = kh * 64 + 32 * b + 8 * kq;
        const __bf16 *ar = A0b + (size_t)(16 * ti + i) * LDB + k0, *pr = Pb + (size_t)i * LDB + k0;
        const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(ar), am = *reinterpret_cast<const bf16x8 *>(ar + (size_t)F * LDB),
                     al = *reinterpret_cast<const bf16x8 *>(ar + (size_t)2 * F * LDB);
        const bf16x8 ph = *reinterpret_cast<const bf16x8 *>(pr), pm = *reinterpret_cast<const bf16x8 *>(pr + (size_t)16 * LDB),
                     pl = *reinterpret_cast<const bf16x8 *>(pr + (size_t)2 * 16 * LDB);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, ph, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, pl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, pm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, ph, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, pm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, ph, acc, 0, 0, 0);
      }
    } else {
#pragma unroll
    for (int kb = 0; kb < Cfg::KB; ++kb) {
      const int k0 = (kh * Cfg::KB + kb) * 16 + 4 * kq;
      const float4 a = *reinterpret_cast<const float4 *>(A0s + (16 * ti + i) * LD + k0);
      const float4 b = *reinterpret_cast<const float4 *>(Ps + i * LD + k0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
    }
    }
    *reinterpret_cast<float4 *>(Outs + (kh * 16 + i) * LD + 16 * ti + 4 * kq) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  };
  auto collect = [&](float (&out)[FC]) {  // after the barrier that follows the product: the K-slices of this wave's row
#pragma unroll
    for (int c = 0; c < FC; ++c) out[c] = 0.f;
#pragma unroll
    for (int h = 0; h < Cfg::KH; ++h) {
      const float *o = Outs + (h * 16 + wave) * LD + cf;
      if constexpr (FC == 2) {
        const float2 t = *reinterpret_cast<const float2 *>(o);
        out[0] += t.x, out[1] += t.y;
      } else {
        out[0] += o[0];
      }
    }
  };
  auto dense = [&](const float (&vec)[FC], bool valid, float (&out)[FC]) {
    publish(vec, valid);
    product();
    __syncthreads();
    collect(out);
  };
  // STAGGER: between the two barriers of a pass every wave has a matrix-pipe block (its 4 KB MFMAs of the 16-row product) and a
  // vector-pipe block (its own row's tile entries, which need only the operand it published itself).  With all waves of a
  // SIMD in the same block one pipe idled while the other worked (knock-outs: the product, the tile entries and the barriers
  // each "cost" 40 % of the kernel).  Waves 0-3 and 8-11 now run the product first, waves 4-7 and 12-15 their tile entries
  // first -- two of each kind per SIMD (wave w sits on SIMD w mod 4) -- so the pipes work side by side, with the same
  // barriers and the same arithmetic.
  const bool product_first = ((wave >> 2) & 1) == 0;

  const int groups = (count + 15) / 16, g_step = gridDim.x;
  // row of this wave in group g (groups past the end and rows past the count re-read the last row and stay invalid)
  auto row_id = [&](int g) { return order[first + min(g * 16 + wave, count - 1)]; };  // uniform address: scalar load
  auto row_valid = [&](int g) { return g < groups && g * 16 + wave < count; };
  int id0 = row_id(blockIdx.x), id1 = row_id(blockIdx.x + g_step), id2 = row_id(blockIdx.x + 2 * g_step), id3 = row_id(blockIdx.x + 3 * g_step);
  int b0 = indptr[id0], e0 = indptr[id0 + 1], b1 = indptr[id1], e1 = indptr[id1 + 1], b2 = indptr[id2], e2 = indptr[id2 + 1];
  int ent_col, ent_cnt = row_valid(blockIdx.x) ? e0 - b0 : 0;
  float ent_c;
  fetch_entries(indices, data, opaque(lane), b0, max(e0, b0 + 1), ent_col, ent_c);
  bool tile_ready = false;
  int cnt = 0;
  f32x2 y[8][FE / 2];
  float x[FC];
  auto kill = [](float (&v)[FC]) {
#pragma unroll
    for (int cc = 0; cc < FC; ++cc) v[cc] = 0.f;
  };
  kill(x);
  for (int g = blockIdx.x; g < groups; g += g_step) {
    const bool valid = row_valid(g);
    ST *xrow = X + (size_t)id0 * F;
    if (!tile_ready) {  // first group, or this wave's previous row ended before its last pass
      cnt = ent_cnt;
      ent_col = opaque(ent_col);
      ent_c = __int_as_float(opaque(__float_as_int(ent_c)));
      static_for<4>([&](auto Pc) {
        constexpr int P = decltype(Pc)::value;
        if (8 * P < cnt) gather_pair<F, P>(y, cw, ent_col, ent_c, cnt, Y, lane);
      });
      ent_cnt = row_valid(g + g_step) ? e1 - b1 : 0;
      fetch_entries(indices, data, opaque(lane), b1, max(e1, b1 + 1), ent_col, ent_c);
      load_compact<F>(xrow, opaque(lane), x);
    }
    // ent_* now describe this wave's row of group g + g_step
    float xc[FC], r[FC], p[FC], Ap[FC], sp[FC];
#pragma unroll
    for (int cc = 0; cc < FC; ++cc) xc[cc] = x[cc];
    // r = -(A0 x) + sum_k (c+ - (|c|-1) y.x) y        (_als.pyx:187-201)
    if constexpr (STAGGER) {
      publish(xc, valid);
      if (!product_first) tile_pass<F, true, false, ST>(y, cw, cnt, prow, sp, lane, 0, ent_col, ent_c, Y);
      product();
      if (product_first) tile_pass<F, true, false, ST>(y, cw, cnt, prow, sp, lane, 0, ent_col, ent_c, Y);
      __syncthreads();
      collect(Ap);
    } else {
      dense(xc, valid, Ap);
      tile_pass<F, true, false, ST>(y, cw, cnt, prow, sp, lane, 0, ent_col, ent_c, Y);
    }
#pragma unroll
    for (int cc = 0; cc < FC; ++cc) p[cc] = r[cc] = sp[cc] - Ap[cc];
    float rsold = dot_compact<F>(r, r);
    bool active = valid && rsold >= 1e-20f;  // else: x untouched (_als.pyx:206)
    const bool store = active;
    for (int it = 0; it + 1 < cg_steps; ++it) {  // all steps but the last; every wave takes the barriers of dense()
      if constexpr (STAGGER) {
        publish(p, active);
        if (active && !product_first) tile_pass<F, false, false, ST>(y, cw, cnt, prow, sp, lane, 0, ent_col, ent_c, Y);
        product();
        if (active && product_first) tile_pass<F, false, false, ST>(y, cw, cnt, prow, sp, lane, 0, ent_col, ent_c, Y);
        __syncthreads();
        collect(Ap);
      } else {
        dense(p, active, Ap);
      }
      if (active) {  // wave-uniform
        if constexpr (!STAGGER) tile_pass<F, false, false, ST>(y, cw, cnt, prow, sp, lane, 0, ent_col, ent_c, Y);
#pragma unroll
        for (int cc = 0; cc < FC; ++cc) Ap[cc] += sp[cc];
        const float alpha = rsold * __builtin_amdgcn_rcpf(dot_compact<F>(p, Ap));
#pragma unroll
        for (int cc = 0; cc < FC; ++cc) {
          xc[cc] = fmaf(alpha, p[cc], xc[cc]);
          r[cc] = fmaf(-alpha, Ap[cc], r[cc]);
        }
        const float rsnew = dot_compact<F>(r, r);
        if (rsnew < 1e-20f) {
          active = false;  // the oracle breaks here (_als.pyx:235); the wave keeps taking the barriers
        } else {
          const float beta = rsnew * __builtin_amdgcn_rcpf(rsold);
#pragma unroll
          for (int cc = 0; cc < FC; ++cc) p[cc] = fmaf(beta, p[cc], r[cc]);
          rsold = rsnew;
        }
      }
    }
    // last step: only its x update is evaluated (_als.pyx:226-241 compute r, rsnew, p that nothing reads); its tile pass
    // rolls the next group's entries in
    bool rolled = false;
    if (cg_steps > 0) {
      if constexpr (STAGGER) {
        publish(p, active);
        auto last_tiles = [&]() {
          if constexpr (ROLL) tile_pass<F, false, true, ST>(y, cw, cnt, prow, sp, lane, ent_cnt, ent_col, ent_c, Y);
          else tile_pass<F, false, false, ST>(y, cw, cnt, prow, sp, lane, 0, ent_col, ent_c, Y);
        };
        if (active && !product_first) last_tiles();
        product();
        if (active && product_first) last_tiles();
        __syncthreads();
        collect(Ap);
      } else {
        dense(p, active, Ap);
      }
      if (active) {
        if constexpr (ROLL) {
          if constexpr (!STAGGER) tile_pass<F, false, true, ST>(y, cw, cnt, prow, sp, lane, ent_cnt, ent_col, ent_c, Y);
          cnt = ent_cnt;
          ent_cnt = row_valid(g + 2 * g_step) ? e2 - b2 : 0;
          fetch_entries(indices, data, opaque(lane), b2, max(e2, b2 + 1), ent_col, ent_c);
          load_compact<F>(X + (size_t)id1 * F, opaque(lane), x);
          rolled = true;
        } else {
          if constexpr (!STAGGER) tile_pass<F, false, false, ST>(y, cw, cnt, prow, sp, lane, 0, ent_col, ent_c, Y);
          kill(x);
        }
#pragma unroll
        for (int cc = 0; cc < FC; ++cc) Ap[cc] += sp[cc];
        const float alpha = rsold * __builtin_amdgcn_rcpf(dot_compact<F>(p, Ap));
#pragma unroll
        for (int cc = 0; cc < FC; ++cc) xc[cc] = fmaf(alpha, p[cc], xc[cc]);
      } else {
        kill(x);
      }
    } else {
      kill(x);
    }
    if (store) store_compact<F>(xrow, opaque(lane), xc);
    tile_ready = rolled;
    id0 = id1, id1 = id2, id2 = id3, id3 = row_id(g + 4 * g_step);
    b0 = b1, e0 = e1, b1 = b2, e1 = e2, b2 = indptr[id2], e2 = indptr[id2 + 1];
  }
}

template <typename T>
static void launch_qfgroup(const imp_csr *C, int first, int count, T *X, const T *Y, const float *A0, int cg_steps, const char *name) {
  if (count <= 0) return;
  // the one form kept: staggered pipes, the product on the bf16 matrix cores with three-term operands (f = 128).  The unstaggered
  // order and the exact-fp32 MFMA product (template arguments STAGGER / BF3 = false; IMP_SHORT_STAGGER, IMP_SHORT_BF16X3 until
  // round 5) are no longer instantiated.
  constexpr int F = 128;
  const size_t lds = QFGroupCfg<F>::lds_bytes_bf3;
  auto kern = als_cg_qfgroup_kernel<F, true, true, T>;
  IMP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int grid = std::min((count + 15) / 16, ctx().num_cus * std::max(2, ctx().oversub));  // (1 / 2 / 3 per CU measured within 1 %)
  IMP_PROF(name);
  kern<<<grid, 1024, lds, stream()>>>(C->order.data(), first, count, C->indptr.data(), C->indices.data(), C->data.data(), X, Y, A0,
                                      cg_steps);
  IMP_CHECK_HIP(hipGetLastError());
}

template <typename T>
void launch_group_fused(const imp_csr *C, int f, int first, int count, T *X, const T *Y, const float *A0, int cg_steps, const char *name) {
  if (f == 128) launch_qfgroup<T>(C, first, count, X, Y, A0, cg_steps, name);
  else throw std::invalid_argument("launch_group_fused: f must be 128 (f = 64 short rows run on independent wavefronts)");
}
template void launch_group_fused<float>(const imp_csr *, int, int, int, float *, const float *, const float *, int, const char *);
template void launch_group_fused<__half>(const imp_csr *, int, int, int, __half *, const __half *, const float *, int, const char *);

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
