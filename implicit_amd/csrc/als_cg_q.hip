// K1q: the CG half sweep for f = 64 / 128 on quarter-layout register tiles (als_qtile.h).
//
// Arithmetic contract: the oracle's CG (implicit/cpu/_als.pyx:152-248).  Schedule --
// short rows: one wavefront per row (f = 128: 16 rows per workgroup in lock step, gramian product on fp32 MFMA);
// mid rows: a team of 2/4/8/16 wavefronts per row with the whole row resident.  The per-pass work of a wave is
// organised around the four 16-lane DPP rows instead of the whole wave: dots reduce inside one DPP row, weights
// need no broadcast, and accumulators return to the compact CG-state layout through two permlane-swap levels.
#include <type_traits>

#include "als_qtile.h"
#include "common.h"

namespace imp {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- MFMA gramian product for 16 lock-step rows (compact layout in, compact layout out) -------------------------
template <int F> struct QGroupCfg {
  static constexpr int LD = F + 8;          // A0 / P / Out row stride in LDS (conflict-free b128 fragment reads)
  static constexpr int NT = F / 16;         // 16-factor output tiles
  static constexpr int KH = 16 / NT;        // K-slices so that NT * KH == 16 waves
  static constexpr int KB = (F / 16) / KH;  // 16-factor k-blocks per wave
  static constexpr size_t lds_floats = (size_t)F * LD + 16 * LD + (size_t)KH * 16 * LD;
};

template <int F>
__device__ __forceinline__ void group_gram_matvec_q(const float *A0s, float *Ps, float *Outs, int wave, int lane, bool valid,
                                                    const float (&vec)[F / 64], float (&out)[F / 64]) {
  using Cfg = QGroupCfg<F>;
  constexpr int FC = F / 64, LD = Cfg::LD;
#pragma unroll
  for (int c = 0; c < FC; ++c) Ps[wave * LD + QL<F>::cfactor(lane, c)] = valid ? vec[c] : 0.f;
  __syncthreads();
  const int ti = wave % Cfg::NT, kh = wave / Cfg::NT;
  const int i = lane & 15, kq = lane >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
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
  *reinterpret_cast<float4 *>(Outs + (kh * 16 + i) * LD + 16 * ti + 4 * kq) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  __syncthreads();
#pragma unroll
  for (int c = 0; c < FC; ++c) {
    float s = 0.f;
#pragma unroll
    for (int h = 0; h < Cfg::KH; ++h) s += Outs[(h * 16 + wave) * LD + QL<F>::cfactor(lane, c)];
    out[c] = s;
  }
}

// ---- short rows (<= 32 nnz): one wave per row, 16 rows per workgroup ----------------------------------------------
template <int F, typename T>
__global__ __launch_bounds__(1024) void als_cg_qgroup_kernel(const int32_t *__restrict__ order, int first, int count,
                                                             const int32_t *__restrict__ indptr,
                                                             const int32_t *__restrict__ indices,
                                                             const float *__restrict__ data, T *__restrict__ X,
                                                             const T *__restrict__ Y, const float *__restrict__ A0,
                                                             int cg_steps) {
  using Cfg = QGroupCfg<F>;
  constexpr int FC = F / 64, FE = F / 16, LD = Cfg::LD;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *A0s = smem;
  float *Ps = A0s + (size_t)F * LD;
  float *Outs = Ps + 16 * LD;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int e = threadIdx.x; e < F * F; e += 1024) {
    int r = e / F, c = e - r * F;
    A0s[r * LD + c] = A0[e];
  }
  __syncthreads();

  const int groups = (count + 15) / 16;
  // groups past the end re-read the last row and are masked by `valid`
  // schedule entry -> row id -> nnz range -> entries are four dependent loads; each stage runs one group further ahead
  // than the next, so none of them is waited for when it is issued: row ids 3 groups ahead, nnz ranges 2, entries 1
  auto row_id = [&](int g) { return order[first + min(g * 16 + wave, count - 1)]; };  // uniform address: scalar load
  const int g_step = gridDim.x;
  int u1 = row_id(blockIdx.x), u2 = row_id(blockIdx.x + g_step), u3 = row_id(blockIdx.x + 2 * g_step);
  int rb1 = indptr[u1], re1 = indptr[u1 + 1], rb2 = indptr[u2], re2 = indptr[u2 + 1];
  int col_next;
  float c_next;
  fetch_entries(indices, data, lane, rb1, re1, col_next, c_next);
  for (int g = blockIdx.x; g < groups; g += g_step) {
    const bool valid = g * 16 + wave < count;
    const int u = u1, row_begin = rb1, row_end = re1;
    u1 = u2, rb1 = rb2, re1 = re2;                    // group g + 1: complete
    u2 = u3, rb2 = indptr[u2], re2 = indptr[u2 + 1];  // group g + 2: row id known -> its range
    u3 = row_id(g + 3 * g_step);                      // group g + 3: row id
    T *xrow = X + (size_t)u * F;
    float x[FC], r[FC], p[FC], Ap[FC], sp[FC];
    load_compact<F>(xrow, lane, x);
    QTile<F> tile;
    load_qtile_staged<F>(tile, col_next, c_next, Y, lane, valid ? row_end - row_begin : 0);
    fetch_entries(indices, data, lane, rb1, re1, col_next, c_next);  // entries of group g + 1

    float ve[FE], ae[FE];
    // r = -(A0 x) + sum_k (c+ - (|c|-1) y.x) y        (_als.pyx:187-201)
    group_gram_matvec_q<F>(A0s, Ps, Outs, wave, lane, valid, x, Ap);
    expand_vector<F>(x, ve);
#pragma unroll
    for (int e = 0; e < FE; ++e) ae[e] = 0.f;
    qtile_apply<F, true>(tile, ve, ae);
    reduce_expanded<F>(ae, sp);
#pragma unroll
    for (int c = 0; c < FC; ++c) p[c] = r[c] = sp[c] - Ap[c];
    float rsold = dot_compact<F>(r, r);
    bool active = valid && rsold >= 1e-20f;  // else: x untouched (_als.pyx:206)
    const bool store = active;

    for (int it = 0; it < cg_steps; ++it) {
      group_gram_matvec_q<F>(A0s, Ps, Outs, wave, lane, active, p, Ap);
      if (active) {  // wave-uniform
        expand_vector<F>(p, ve);
#pragma unroll
        for (int e = 0; e < FE; ++e) ae[e] = 0.f;
        qtile_apply<F, false>(tile, ve, ae);
        reduce_expanded<F>(ae, sp);
#pragma unroll
        for (int c = 0; c < FC; ++c) Ap[c] += sp[c];
        float alpha = rsold / dot_compact<F>(p, Ap);
#pragma unroll
        for (int c = 0; c < FC; ++c) {
          x[c] = fmaf(alpha, p[c], x[c]);
          r[c] = fmaf(-alpha, Ap[c], r[c]);
        }
        float rsnew = dot_compact<F>(r, r);
        if (rsnew < 1e-20f) {
          active = false;  // the oracle breaks here (_als.pyx:235); keep taking the barriers
        } else {
          float beta = rsnew / rsold;
#pragma unroll
          for (int c = 0; c < FC; ++c) p[c] = fmaf(beta, p[c], r[c]);
          rsold = rsnew;
        }
      }
    }
    if (store) store_compact<F>(xrow, lane, x);
  }
}

// ---- mid rows: a team of WPR wavefronts per row, the whole row resident -----------------------------------------------
// STATS (debug, IMP_CG_STATS=1): s_memtime ticks (shader-clock cycles on gfx950) summed over waves per phase --
//   [0] row start -> tile resident (gathers drained)  [1] operand vector to LDS + expand  [2] dense part
//   [3] tile entries  [4] reduce-scatter  [5] combine (barriers included)  [6] dots / CG update  [7] wave-rows
template <int F, int WPR, int BLOCK, bool STATS, typename ST>
__global__ __launch_bounds__(BLOCK, 4) void als_cg_qteam_kernel(const int32_t *__restrict__ order, int first, int count,
                                                                const int32_t *__restrict__ indptr,
                                                                const int32_t *__restrict__ indices,
                                                                const float *__restrict__ data, ST *__restrict__ X,
                                                                const ST *__restrict__ Y, const float *__restrict__ A0,
                                                                int cg_steps, unsigned long long *__restrict__ stats = nullptr) {
  unsigned long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last = 0;
  auto tick = [&](int slot) {  // charge the time since the previous tick to `slot`
    if constexpr (STATS) {
      __builtin_amdgcn_sched_barrier(0);
      unsigned long long now = __builtin_amdgcn_s_memtime();
      if (slot >= 0) tk[slot] += now - t_last;
      t_last = now;
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  constexpr int FC = F / 64, FE = F / 16, T = 32, WAVES = BLOCK / 64, TEAMS = WAVES / WPR;
  static_assert(WPR <= WAVES && (F / WPR) % 4 == 0, "team width");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *A0s = smem;                             // [F][F]
  float *scratch = A0s + (size_t)F * F;          // [2][WAVES][F]  partial vectors of the combine, double-buffered
  float *vecs = scratch + (size_t)2 * WAVES * F;  // [WAVES][F]  wave-private copy of the operand vector (natural order)
  unsigned *arrivals = reinterpret_cast<unsigned *>(vecs + (size_t)WAVES * F);  // [TEAMS] monotonic team-barrier counters
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int team = wave / WPR, sub = wave % WPR;
  for (int e = threadIdx.x; e < F * F; e += BLOCK) A0s[e] = A0[e];
  if (threadIdx.x < TEAMS) arrivals[threadIdx.x] = 0u;
  __syncthreads();  // the only workgroup-wide barrier: from here on the teams run their rows independently
  const int j_begin = F * sub / WPR;
  float *myvec = vecs + (size_t)wave * F;

  // Team-local barrier.  s_barrier is workgroup-wide, which would hold the TEAMS rows of a workgroup in lock step --
  // every wave of the CU gathering at once, then every wave hammering the LDS at once.  The waves of a workgroup
  // are co-resident, so a team can meet on a monotonic LDS counter instead (each wave keeps its own target); teams then
  // drift apart and one team's gather latency hides under the others' arithmetic.
  unsigned arrive_target = 0;
  auto team_sync = [&]() {
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
  // sum of the team's WPR partial vectors (fixed order); every wave of the team gets the same bits.  The partials
  // alternate between two buffers, so one meeting per combine is enough: a wave can only overwrite a buffer two
  // combines later, which it cannot reach before its team mates have passed the combine in between.
  int parity = 0;
  auto combine = [&](float (&acc)[FC]) {
    if constexpr (WPR == 1) return;  // independent waves: nothing to combine
    float *buf = scratch + (size_t)parity * WAVES * F;
    parity ^= 1;
#pragma unroll
    for (int c = 0; c < FC; ++c) buf[wave * F + QL<F>::cfactor(lane, c)] = acc[c];
    team_sync();
#pragma unroll
    for (int c = 0; c < FC; ++c) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < WPR; ++w) s += buf[(team * WPR + w) * F + QL<F>::cfactor(lane, c)];
      acc[c] = s;
    }
    tick(5);
  };
  // one pass: acc (compact) = [A0 rows of this wave] . v + [tile entries of this wave] weights
  auto pass = [&](auto first_tag, const QTile<F> &tile, const float (&v)[FC], float (&acc)[FC], bool work) {
    float ve[FE], ae[FE];
#pragma unroll
    for (int e = 0; e < FE; ++e) ae[e] = 0.f;
    tick(6);
    if (work) {  // wave-uniform, identical across the team
#pragma unroll
      for (int c = 0; c < FC; ++c) myvec[QL<F>::cfactor(lane, c)] = v[c];  // wave-private: no barrier needed
      // the dense part runs before the operand is expanded: with the tile resident the register file is full, and
      // every register not live here is one more LDS read the loop below can keep in flight
      gram_matvec_q<F, F / WPR / 4>(A0s, F, myvec, lane, j_begin, ae);
      __builtin_amdgcn_sched_barrier(0);
      tick(2);
      expand_vector<F>(v, ve);
      tick(1);
      qtile_apply<F, decltype(first_tag)::value>(tile, ve, ae);
      tick(3);
    }
    reduce_expanded<F>(ae, acc);
    tick(4);
  };

  // this team's rows: i = (blockIdx.x + k gridDim.x) TEAMS + team; rows past the end re-read the last row
  auto row_id = [&](int i) { return order[first + min(i, count - 1)]; };  // uniform address: scalar load
  const int i_step = gridDim.x * TEAMS, i_first = blockIdx.x * TEAMS + team;
  // four dependent loads per row (schedule entry -> row id -> nnz range -> entries), each stage one row further ahead
  // than the next: row ids 3 rows ahead, nnz ranges 2, entries 1 -- nothing is waited for at the point of issue
  int u1 = row_id(i_first), u2 = row_id(i_first + i_step), u3 = row_id(i_first + 2 * i_step);
  int rb1 = indptr[u1], re1 = indptr[u1 + 1], rb2 = indptr[u2], re2 = indptr[u2 + 1];
  int col_next;
  float c_next;
  // A row's entries are dealt to the team in EVEN shares (rounded up to whole 4-entry tile steps), not 32 at a time:
  // with 32-entry slices the first waves of a team carried full tiles and the last ones little or nothing, and since
  // wave w of a workgroup sits on SIMD (w mod 4) the full-tile waves of every team shared the same SIMDs.
  auto slice = [&](int rb, int re, int &k0, int &cnt) {
    const int chunk = min(T, (((re - rb) + WPR - 1) / WPR + 3) & ~3);
    k0 = min(rb + chunk * sub, re);
    cnt = min(chunk, re - k0);
  };
  int k0_next, cnt_next;
  slice(rb1, re1, k0_next, cnt_next);
  fetch_entries(indices, data, lane, k0_next, max(k0_next + cnt_next, rb1 + 1), col_next, c_next);
  for (int i = i_first; i < count; i += i_step) {
    constexpr bool valid = true;
    const int u = u1;
    // scalar stages first (they share lgkmcnt with the LDS: the first LDS wait of the row also waits for them, and by
    // then the gathers below have covered their latency)
    u1 = u2, rb1 = rb2, re1 = re2;                    // row i + step: complete
    u2 = u3, rb2 = indptr[u2], re2 = indptr[u2 + 1];  // row i + 2 step: row id known -> its range
    u3 = row_id(i + 3 * i_step);                      // row i + 3 step: row id
    ST *xrow = X + (size_t)u * F;
    float x[FC], r[FC], p[FC], Ap[FC];
    tick(-1);
    const int cnt = cnt_next;  // this wave's slice of the row (may be empty)
    QTile<F> tile;
    load_compact<F>(xrow, lane, x);  // first in the queue: the dense part of the first pass only needs x
    load_qtile_staged<F>(tile, col_next, c_next, Y, lane, cnt);
    slice(rb1, re1, k0_next, cnt_next);
    fetch_entries(indices, data, lane, k0_next, max(k0_next + cnt_next, rb1 + 1), col_next, c_next);  // entries of row i + step
    if constexpr (STATS) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      tick(0);
      tk[7] += 1;
    }

    // r = -(A0 x) + sum_k (c+ - (|c|-1) y.x) y        (_als.pyx:187-201): the dense part enters with a minus sign
    {
      float ve[FE], ae[FE];
#pragma unroll
      for (int e = 0; e < FE; ++e) ae[e] = 0.f;
#pragma unroll
      for (int c = 0; c < FC; ++c) myvec[QL<F>::cfactor(lane, c)] = x[c];
      gram_matvec_q<F, F / WPR / 4>(A0s, F, myvec, lane, j_begin, ae);
#pragma unroll
      for (int e = 0; e < FE; ++e) ae[e] = -ae[e];
      __builtin_amdgcn_sched_barrier(0);
      tick(2);
      expand_vector<F>(x, ve);
      tick(1);
      qtile_apply<F, true>(tile, ve, ae);
      tick(3);
      reduce_expanded<F>(ae, r);
      tick(4);
    }
    combine(r);
#pragma unroll
    for (int c = 0; c < FC; ++c) p[c] = r[c];
    float rsold = dot_compact<F>(r, r);
    bool active = valid && rsold >= 1e-20f;  // else: x untouched (_als.pyx:206)
    const bool store = active && sub == 0;

    for (int it = 0; it < cg_steps; ++it) {
      pass(std::false_type{}, tile, p, Ap, active);
      combine(Ap);
      if (active) {
        float alpha = rsold / dot_compact<F>(p, Ap);
#pragma unroll
        for (int c = 0; c < FC; ++c) {
          x[c] = fmaf(alpha, p[c], x[c]);
          r[c] = fmaf(-alpha, Ap[c], r[c]);
        }
        float rsnew = dot_compact<F>(r, r);
        if (rsnew < 1e-20f) {
          active = false;  // the oracle breaks here (_als.pyx:235); the whole team takes the same branch
        } else {
          float beta = rsnew / rsold;
#pragma unroll
          for (int c = 0; c < FC; ++c) p[c] = fmaf(beta, p[c], r[c]);
          rsold = rsnew;
        }
      }
    }
    if (store) store_compact<F>(xrow, lane, x);
    tick(6);
  }
  if constexpr (STATS) {
    if (lane == 0)
      for (int i = 0; i < 8; ++i) atomicAdd(&stats[i], tk[i]);
  }
}

template <int F, typename T>
static void launch_qgroup(const imp_csr *C, int first, int count, T *X, const T *Y, const float *A0, int cg_steps,
                          const char *name) {
  if (count <= 0) return;
  size_t lds = QGroupCfg<F>::lds_floats * sizeof(float);
  auto kern = als_cg_qgroup_kernel<F, T>;
  IMP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  static const int per_cu = getenv("IMP_QGROUP_PER_CU") ? std::max(1, atoi(getenv("IMP_QGROUP_PER_CU"))) : 2;
  int grid = std::min((count + 15) / 16, ctx().num_cus * std::max(per_cu, ctx().oversub));
  IMP_PROF(name);
  kern<<<grid, 1024, lds, stream()>>>(C->order.data(), first, count, C->indptr.data(), C->indices.data(), C->data.data(), X, Y, A0,
                                      cg_steps);
  IMP_CHECK_HIP(hipGetLastError());
}

template <int F, int WPR, int BLOCK, typename T>
static void launch_qteam(const imp_csr *C, int first, int count, T *X, const T *Y, const float *A0, int cg_steps,
                         const char *name) {
  if (count <= 0) return;
  constexpr int WAVES = BLOCK / 64, TEAMS = WAVES / WPR;
  size_t lds = ((size_t)F * F + 3 * WAVES * F + TEAMS) * sizeof(float);
  auto kern = als_cg_qteam_kernel<F, WPR, BLOCK, false, T>;
  IMP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int per_cu = (int)std::max<size_t>(1, std::min<size_t>(2048 / BLOCK, (160 * 1024) / lds));
  // workgroups per resident slot: fixed shares of a length-sorted schedule leave the slots unevenly loaded towards the end
  // of the launch; smaller shares dealt by the hardware dispatcher even that out (C3: (32,64] 1.11 -> 1.04 ms at 4x,
  // (64,128] 0.88 -> 0.84 at 4x, (128,256] 0.585 -> 0.574 at 2x; the one-workgroup-per-CU kernels lose: each new
  // workgroup stages the gramian again).  The multi-GPU driver's factor (Context::oversub) is a floor, not a multiplier
  constexpr int kBaseOversub = WPR <= 4 ? 4 : (WPR == 8 ? 2 : 1);
  int grid = std::min((count + TEAMS - 1) / TEAMS, ctx().num_cus * per_cu * std::max(kBaseOversub, ctx().oversub));
  static const bool want_stats = getenv("IMP_CG_STATS") != nullptr;
  if (want_stats) {  // debug: per-phase tick sums of this launch, printed to stderr
    static unsigned long long *stats = nullptr;
    if (!stats) IMP_CHECK_HIP(hipMalloc(&stats, 8 * sizeof(unsigned long long)));
    IMP_CHECK_HIP(hipMemsetAsync(stats, 0, 8 * sizeof(unsigned long long), stream()));
    auto skern = als_cg_qteam_kernel<F, WPR, BLOCK, true, T>;
    IMP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(skern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    skern<<<grid, BLOCK, lds, stream()>>>(C->order.data(), first, count, C->indptr.data(), C->indices.data(), C->data.data(), X,
                                         Y, A0, cg_steps, stats);
    unsigned long long h[8];
    IMP_CHECK_HIP(hipMemcpyAsync(h, stats, sizeof(h), hipMemcpyDeviceToHost, stream()));
    IMP_CHECK_HIP(hipStreamSynchronize(stream()));
    const double n = h[7] ? (double)h[7] : 1.0;
    fprintf(stderr,
            "[cg-stats] %s rows=%d wave-rows=%.0f  cycles/wave-row: gather %.1f vec+expand %.1f dense %.1f entries %.1f "
            "reduce %.1f combine %.1f update %.1f\n",
            name, count, n, h[0] / n, h[1] / n, h[2] / n, h[3] / n, h[4] / n, h[5] / n, h[6] / n);
    return;
  }
  IMP_PROF(name);
  kern<<<grid, BLOCK, lds, stream()>>>(C->order.data(), first, count, C->indptr.data(), C->indices.data(), C->data.data(), X, Y,
                                      A0, cg_steps, nullptr);
  IMP_CHECK_HIP(hipGetLastError());
}

// als_cg_qf.hip: the team kernels with fused passes and rolling gathers (round 3)
template <typename T>
void launch_team_fused(const imp_csr *C, int f, int width, int first, int count, T *X, const T *Y, const float *A0, int cg_steps,
                       const char *name);

template <typename T>
void launch_group_fused(const imp_csr *C, int f, int first, int count, T *X, const T *Y, const float *A0, int cg_steps, const char *name);

// als_cg_qh.hip: 64-entry tiles, half the wavefronts per row (round 4): packed halves for float16 storage, "fat" wavefronts
// (two per SIMD at f = 128) for fp32
template <typename ST>
void launch_team_tile64(const imp_csr *C, int f, int width, int first, int count, ST *X, const ST *Y, const float *A0, int cg_steps,
                        const char *name);

template <int F, typename T> static void run_classes_q(const imp_csr *C, T *X, const T *Y, const float *A0, int cg_steps) {
  const int32_t *b = C->bin_start;  // classes: 1 (256,512]  2 (128,256]  3 (64,128]  4 (32,64]  5 (16,32]  6 (0,16]
  {
    // 64-entry tiles.  fp16 storage: on by default (IMP_HALF_TILE64=0: the fp32-tile kernels below; 4.9 against 4.4-4.5 ms per
    // configs[2] iteration).  fp32 storage: IMP_TILE64=<mask> selects it per class (1: (256,512]  2: (128,256]  4: (64,128]
    // 8: (32,64]).
    static const bool half64 = !(getenv("IMP_HALF_TILE64") && atoi(getenv("IMP_HALF_TILE64")) == 0);
    static const int float64 = getenv("IMP_TILE64") ? atoi(getenv("IMP_TILE64")) : 0;
    const int mask = std::is_same<T, __half>::value ? (half64 ? 15 : 0) : float64;
    if (mask) {
      auto cls = [&](int bit, int width64, int width32, int lo, int hi, const char *name) {
        class_stream_next();
        if (mask & bit) launch_team_tile64<T>(C, F, width64, lo, hi - lo, X, Y, A0, cg_steps, name);
        else launch_team_fused<T>(C, F, width32, lo, hi - lo, X, Y, A0, cg_steps, name);
      };
      if (!team16_as_cluster()) cls(1, 8, 16, b[1], b[2], "als_cg_team16_rows");  // names: the row class
      cls(2, 4, 8, b[2], b[3], "als_cg_team8_rows");
      cls(4, 2, 4, b[3], b[4], "als_cg_team4_rows");
      cls(8, 1, 2, b[4], b[5], "als_cg_team2_rows");
      class_stream_next();
      if constexpr (F == 64) launch_team_fused<T>(C, F, 1, b[5], b[7] - b[5], X, Y, A0, cg_steps, "als_cg_short_rows");
      else launch_group_fused<T>(C, F, b[5], b[7] - b[5], X, Y, A0, cg_steps, "als_cg_short_rows");
      return;
    }
  }
  // IMP_TEAM_FUSED=0: the round-2 team kernels (dense part, then tile part; gathers at the row start) -- A/B and the
  // IMP_CG_STATS instrumentation; a bit mask selects the round-3 kernel per team width (1: 16 waves, 2: 8, 4: 4, 8: 2, 16: 1, 32: the lock-step short-row kernel)
  static const int fused = getenv("IMP_TEAM_FUSED") ? atoi(getenv("IMP_TEAM_FUSED")) : 63;
  auto team = [&](int bit, int width, int first, int count, const char *name, auto old) {
    class_stream_next();
    if (fused & bit) launch_team_fused<T>(C, F, width, first, count, X, Y, A0, cg_steps, name);  // IMP_CG_STATS: its own instrumented form
    else old(first, count, name);
  };
  if (!team16_as_cluster())
    team(1, 16, b[1], b[2] - b[1], "als_cg_team16_rows",
         [&](int fr, int n, const char *nm) { launch_qteam<F, 16, 1024, T>(C, fr, n, X, Y, A0, cg_steps, nm); });
  team(2, 8, b[2], b[3] - b[2], "als_cg_team8_rows",
       [&](int fr, int n, const char *nm) { launch_qteam<F, 8, 512, T>(C, fr, n, X, Y, A0, cg_steps, nm); });
  team(4, 4, b[3], b[4] - b[3], "als_cg_team4_rows",
       [&](int fr, int n, const char *nm) { launch_qteam<F, 4, 512, T>(C, fr, n, X, Y, A0, cg_steps, nm); });
  team(8, 2, b[4], b[5] - b[4], "als_cg_team2_rows",
       [&](int fr, int n, const char *nm) { launch_qteam<F, 2, 512, T>(C, fr, n, X, Y, A0, cg_steps, nm); });
  // short rows.  f = 128: 16 rows per workgroup in lock step with the gramian product on fp32 MFMA (measured 1.16 ms per
  // C3 iteration against 1.38 ms for independent waves with the VALU product -- at f = 128 the product is 57 % of a short
  // row's arithmetic); f = 64: independent waves win (C2: 0.84 against 0.94 ms), the product is a quarter of the size
  // and the lock step costs more than the matrix pipe saves.  IMP_SHORT_TEAM1=0/1 forces one or the other (A/B).
  static const int short_team1 = getenv("IMP_SHORT_TEAM1") ? atoi(getenv("IMP_SHORT_TEAM1")) : -1;
  const bool team1 = short_team1 >= 0 ? short_team1 != 0 : F == 64;
  if (!team1 || F != 64) class_stream_next();  // (the f = 64 team form of the short rows goes through `team`, which moves on itself)
  if (team1) {
    if constexpr (F == 64)
      team(16, 1, b[5], b[7] - b[5], "als_cg_short_rows",
           [&](int fr, int n, const char *nm) { launch_qteam<F, 1, 512, T>(C, fr, n, X, Y, A0, cg_steps, nm); });
    else launch_qteam<F, 1, 512, T>(C, b[5], b[7] - b[5], X, Y, A0, cg_steps, "als_cg_short_rows");
  }
  else if (fused & 32) launch_group_fused<T>(C, F, b[5], b[7] - b[5], X, Y, A0, cg_steps, "als_cg_short_rows");
  else launch_qgroup<F, T>(C, b[5], b[7] - b[5], X, Y, A0, cg_steps, "als_cg_short_rows");  // tile steps beyond cnt are skipped
}

template <typename T> void least_squares_cg_q(const imp_csr *C, T *X, const T *Y, const float *A0, int f, int cg_steps) {
  if (f == 128) run_classes_q<128, T>(C, X, Y, A0, cg_steps);
  else if (f == 64) run_classes_q<64, T>(C, X, Y, A0, cg_steps);
  else throw std::invalid_argument("least_squares_cg_q: f must be 64 or 128");
}
template void least_squares_cg_q<float>(const imp_csr *, float *, const float *, const float *, int, int);
template void least_squares_cg_q<__half>(const imp_csr *, __half *, const __half *, const float *, int, int);

}  // namespace imp
