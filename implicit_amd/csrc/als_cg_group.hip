// K1g: CG half sweep for f = 64 / 128 with the dense (YtY + reg I) . p product on the matrix cores.
//
// Same arithmetic contract as als_cg.hip (oracle: implicit/cpu/_als.pyx:152-248).  At f = 128 and the
// typical ~50 nonzeros per row, the f x f gramian product is MORE than half of the FMAs of a row
// (f^2 vs 2 n f per pass), and as a per-wave broadcast mat-vec it costs ~5 VALU/LDS instructions per
// gramian row.  Here a 1024-thread workgroup owns SIXTEEN rows (one wavefront each, consecutive in the
// length-sorted schedule so they have almost the same nnz) and runs them in lockstep:
//
//   every pass:  each wave writes its vector (x or p) into LDS  P[16][f]            -> barrier
//                D[f][16] = A0 . P^T  with v_mfma_f32_16x16x4_f32 (exact fp32):      the f/16 output tiles x
//                the 16/(f/16) K-slices are dealt to the 16 waves, A0 and P fragments are ds_read_b128
//                (4 consecutive k per lane: the k index is permuted identically for both operands),
//                partial tiles go to LDS  Out[KH][16][f]                              -> barrier
//                each wave reads its row of Out (sum of the KH K-slices) back in its own
//                lane-owns-2-factors layout and continues with the sparse part (als_tile.h).
//
// Early exits of the oracle (rsold < 1e-20, rsnew < 1e-20) become per-wave `active` predicates so that
// all waves execute the same barriers.  LDS: A0 (f x (f+8), conflict-free b128 fragment reads) + P + Out
// = 95 KiB at f = 128 -> one 16-wave workgroup per CU.
#include "als_tile.h"
#include "common.h"

namespace imp {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int F> struct GroupCfg {
  static constexpr int VPL = F / 64;
  static constexpr int LD = F + 8;          // leading dimension of A0 / P / Out rows in LDS (floats)
  static constexpr int NT = F / 16;         // 16-factor output tiles
  static constexpr int KH = 16 / NT;        // K-slices so that NT * KH == 16 waves
  static constexpr int KB = (F / 16) / KH;  // 16-factor k-blocks per wave
  static constexpr size_t lds_floats = (size_t)F * LD + 16 * LD + (size_t)KH * 16 * LD;
};

// acc <- (A0 . vec) for this wave's row; all 16 waves of the block must call it together.
template <int F>
__device__ __forceinline__ void group_gram_matvec(const float *A0s, float *Ps, float *Outs, int wave, int lane, bool valid,
                                                  const float (&vec)[F / 64], float (&out)[F / 64]) {
  using Cfg = GroupCfg<F>;
  constexpr int VPL = Cfg::VPL, LD = Cfg::LD;
  // publish this row's vector (zeros for idle waves so their columns stay finite)
#pragma unroll
  for (int v = 0; v < VPL; ++v) Ps[wave * LD + lane * VPL + v] = valid ? vec[v] : 0.f;
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
  // D tile: column = lane & 15 (row of the group), rows 4*(lane>>4) + reg (factor within the tile)
  *reinterpret_cast<float4 *>(Outs + (kh * 16 + i) * LD + 16 * ti + 4 * kq) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  __syncthreads();
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    float s = 0.f;
#pragma unroll
    for (int h = 0; h < Cfg::KH; ++h) s += Outs[(h * 16 + wave) * LD + lane * VPL + v];
    out[v] = s;
  }
}

template <int F, bool RESIDENT, int T>
__global__ __launch_bounds__(1024) void als_cg_group_kernel(const int32_t *__restrict__ order, int first, int count,
                                                            const int32_t *__restrict__ indptr,
                                                            const int32_t *__restrict__ indices,
                                                            const float *__restrict__ data, float *__restrict__ X,
                                                            const float *__restrict__ Y, const float *__restrict__ A0,
                                                            int cg_steps) {
  using Cfg = GroupCfg<F>;
  constexpr int VPL = Cfg::VPL, LD = Cfg::LD;
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
  for (int g = blockIdx.x; g < groups; g += gridDim.x) {
    const int i = g * 16 + wave;
    const bool valid = i < count;
    int u = 0, row_begin = 0, row_end = 0;
    if (valid) {
      u = __builtin_amdgcn_readfirstlane(order[first + i]);
      row_begin = __builtin_amdgcn_readfirstlane(indptr[u]);
      row_end = __builtin_amdgcn_readfirstlane(indptr[u + 1]);
    }
    float *xrow = X + (size_t)u * F;
    float x[VPL], r[VPL], p[VPL], Ap[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) x[v] = 0.f;
    if (valid) load_row<VPL, true>(xrow, F, lane, x);

    Tile<VPL, T> tile;
    if constexpr (RESIDENT) load_tile<VPL, T>(tile, indices, data, Y, F, lane, row_begin, row_end);

    // r = -(A0 x) + sum_k (c+ - (|c|-1) y.x) y        (_als.pyx:187-201)
    group_gram_matvec<F>(A0s, Ps, Outs, wave, lane, valid, x, Ap);
#pragma unroll
    for (int v = 0; v < VPL; ++v) r[v] = -Ap[v];
    if constexpr (RESIDENT)
      tile_apply<VPL, T, true>(tile, lane, row_begin, row_end, x, r);
    else
      sparse_pass_tiled<VPL, T, true>(indices, data, Y, F, lane, row_begin, row_end, x, r);
#pragma unroll
    for (int v = 0; v < VPL; ++v) p[v] = r[v];
    float rsold = wave_allsum(dot_local<VPL>(r, r));
    bool active = valid && rsold >= 1e-20f;  // else: x untouched (_als.pyx:206)
    const bool store = active;

    for (int it = 0; it < cg_steps; ++it) {
      group_gram_matvec<F>(A0s, Ps, Outs, wave, lane, active, p, Ap);
      if (active) {  // wave-uniform
        if constexpr (RESIDENT)
          tile_apply<VPL, T, false>(tile, lane, row_begin, row_end, p, Ap);
        else
          sparse_pass_tiled<VPL, T, false>(indices, data, Y, F, lane, row_begin, row_end, p, Ap);
        float alpha = rsold / wave_allsum(dot_local<VPL>(p, Ap));
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
          x[v] = fmaf(alpha, p[v], x[v]);
          r[v] = fmaf(-alpha, Ap[v], r[v]);
        }
        float rsnew = wave_allsum(dot_local<VPL>(r, r));
        if (rsnew < 1e-20f) {
          active = false;  // the oracle breaks here (_als.pyx:235); keep taking the barriers
        } else {
          float beta = rsnew / rsold;
#pragma unroll
          for (int v = 0; v < VPL; ++v) p[v] = fmaf(beta, p[v], r[v]);
          rsold = rsnew;
        }
      }
    }
    if (store) {
#pragma unroll
      for (int v = 0; v < VPL; ++v) xrow[lane * VPL + v] = x[v];
    }
  }
}

// ---- mid rows: a TEAM of WPR wavefronts per row, the whole row resident in registers ---------------------------
// Wave `sub` of the team keeps nnz [32 sub, 32 sub + 32) of the row as a register tile for all 1+cg_steps passes
// (the factor rows are gathered ONCE), applies the gramian rows [f sub / WPR, f (sub+1) / WPR) on the VALU, and the
// WPR partial vectors are summed through LDS in a fixed order; every wave of the team then performs the same
// CG update on identical bits.  A 1024-thread workgroup runs 16 / WPR teams in lockstep.
// Optional phase timers (s_memtime ticks summed over waves): [0] row metadata + x, [1] tile gather (drained),
// [2] compute between barriers, [3] barrier waits, [4] groups.  Enabled by IMP_CG_STATS=1 (debug only).
#define IMP_TICK() (stats ? __builtin_amdgcn_s_memtime() : 0ull)

template <int F, int WPR, int BLOCK>
__global__ __launch_bounds__(BLOCK, BLOCK == 512 ? 4 : 4) void als_cg_team_kernel(const int32_t *__restrict__ order, int first, int count,
                                                           const int32_t *__restrict__ indptr,
                                                           const int32_t *__restrict__ indices,
                                                           const float *__restrict__ data, float *__restrict__ X,
                                                           const float *__restrict__ Y, const float *__restrict__ A0,
                                                           int cg_steps, unsigned long long *__restrict__ stats) {
  unsigned long long t_meta = 0, t_gather = 0, t_comp = 0, t_bar = 0, n_groups = 0, t0 = 0, t1 = 0;
  constexpr int VPL = F / 64, T = 32, WAVES = BLOCK / 64, TEAMS = WAVES / WPR;
  static_assert(WPR <= WAVES, "a team cannot be wider than the workgroup");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *A0s = smem;                    // [F][F]
  float *scratch = A0s + (size_t)F * F;  // [WAVES][F]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int team = wave / WPR, sub = wave % WPR;
  for (int e = threadIdx.x; e < F * F; e += BLOCK) A0s[e] = A0[e];
  __syncthreads();
  const int j_begin = F * sub / WPR, j_end = F * (sub + 1) / WPR;

  auto combine = [&](float (&acc)[VPL]) {
#pragma unroll
    for (int v = 0; v < VPL; ++v) scratch[wave * F + lane * VPL + v] = acc[v];
    t1 = IMP_TICK();
    t_comp += t1 - t0;
    __syncthreads();
    t0 = IMP_TICK();
    t_bar += t0 - t1;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < WPR; ++w) s += scratch[(team * WPR + w) * F + lane * VPL + v];
      acc[v] = s;
    }
    t1 = IMP_TICK();
    t_comp += t1 - t0;
    __syncthreads();
    t0 = IMP_TICK();
    t_bar += t0 - t1;
  };

  const int groups = (count + TEAMS - 1) / TEAMS;
  for (int g = blockIdx.x; g < groups; g += gridDim.x) {
    const int i = g * TEAMS + team;
    const bool valid = i < count;
    int u = 0, row_begin = 0, row_end = 0;
    t0 = IMP_TICK();
    if (valid) {
      u = __builtin_amdgcn_readfirstlane(order[first + i]);
      row_begin = __builtin_amdgcn_readfirstlane(indptr[u]);
      row_end = __builtin_amdgcn_readfirstlane(indptr[u + 1]);
    }
    float *xrow = X + (size_t)u * F;
    float x[VPL], r[VPL], p[VPL], Ap[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) x[v] = 0.f;
    if (valid) load_row<VPL, true>(xrow, F, lane, x);
    if (stats) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      t1 = IMP_TICK();
      t_meta += t1 - t0;
      t0 = t1;
    }
    const int k0 = row_begin + T * sub;  // this wave's slice of the row (may be empty)
    Tile<VPL, T> tile;
    load_tile<VPL, T>(tile, indices, data, Y, F, lane, min(k0, row_end), row_end);
    if (stats) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      t1 = IMP_TICK();
      t_gather += t1 - t0;
      t0 = t1;
      ++n_groups;
    }

    // r = -(A0 x) + sum_k (c+ - (|c|-1) y.x) y        (_als.pyx:187-201)
#pragma unroll
    for (int v = 0; v < VPL; ++v) Ap[v] = 0.f;
    gram_matvec<VPL, true>(A0s, F, lane, x, Ap, j_begin, j_end);
#pragma unroll
    for (int v = 0; v < VPL; ++v) r[v] = -Ap[v];
    tile_apply<VPL, T, true>(tile, lane, min(k0, row_end), row_end, x, r);
    combine(r);
#pragma unroll
    for (int v = 0; v < VPL; ++v) p[v] = r[v];
    float rsold = wave_allsum(dot_local<VPL>(r, r));
    bool active = valid && rsold >= 1e-20f;  // else: x untouched (_als.pyx:206)
    const bool store = active && sub == 0;

    for (int it = 0; it < cg_steps; ++it) {
#pragma unroll
      for (int v = 0; v < VPL; ++v) Ap[v] = 0.f;
      if (active) {  // wave-uniform and identical across the team
        gram_matvec<VPL, true>(A0s, F, lane, p, Ap, j_begin, j_end);
        tile_apply<VPL, T, false>(tile, lane, min(k0, row_end), row_end, p, Ap);
      }
      combine(Ap);
      if (active) {
        float alpha = rsold / wave_allsum(dot_local<VPL>(p, Ap));
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
          x[v] = fmaf(alpha, p[v], x[v]);
          r[v] = fmaf(-alpha, Ap[v], r[v]);
        }
        float rsnew = wave_allsum(dot_local<VPL>(r, r));
        if (rsnew < 1e-20f) {
          active = false;  // the oracle breaks here (_als.pyx:235); keep taking the barriers
        } else {
          float beta = rsnew / rsold;
#pragma unroll
          for (int v = 0; v < VPL; ++v) p[v] = fmaf(beta, p[v], r[v]);
          rsold = rsnew;
        }
      }
    }
    if (store) {
#pragma unroll
      for (int v = 0; v < VPL; ++v) xrow[lane * VPL + v] = x[v];
    }
    if (stats) t_comp += IMP_TICK() - t0;
  }
  if (stats && lane == 0) {
    atomicAdd(&stats[0], t_meta);
    atomicAdd(&stats[1], t_gather);
    atomicAdd(&stats[2], t_comp);
    atomicAdd(&stats[3], t_bar);
    atomicAdd(&stats[4], n_groups);
  }
}

template <int F, bool RESIDENT, int T>
static void launch_group(const imp_csr *C, int first, int count, float *X, const float *Y, const float *A0, int cg_steps,
                         const char *name) {
  if (count <= 0) return;
  size_t lds = GroupCfg<F>::lds_floats * sizeof(float);
  auto kern = als_cg_group_kernel<F, RESIDENT, T>;
  IMP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int per_cu = (int)std::max<size_t>(1, std::min<size_t>(2, (160 * 1024) / lds));
  int grid = std::min((count + 15) / 16, ctx().num_cus * per_cu);
  IMP_PROF(name);
  kern<<<grid, 1024, lds, stream()>>>(C->order.data(), first, count, C->indptr.data(), C->indices.data(), C->data.data(), X, Y, A0,
                                      cg_steps);
  IMP_CHECK_HIP(hipGetLastError());
}

template <int F, int WPR, int BLOCK>
static void launch_team(const imp_csr *C, int first, int count, float *X, const float *Y, const float *A0, int cg_steps,
                        const char *name) {
  if (count <= 0) return;
  constexpr int WAVES = BLOCK / 64, TEAMS = WAVES / WPR;
  size_t lds = ((size_t)F * F + WAVES * F) * sizeof(float);
  auto kern = als_cg_team_kernel<F, WPR, BLOCK>;
  IMP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int per_cu = (int)std::max<size_t>(1, std::min<size_t>(2048 / BLOCK, (160 * 1024) / lds));
  int grid = std::min((count + TEAMS - 1) / TEAMS, ctx().num_cus * per_cu);
  static unsigned long long *stats = nullptr;
  static const bool want_stats = getenv("IMP_CG_STATS") != nullptr;
  if (want_stats && !stats) {
    IMP_CHECK_HIP(hipMalloc(&stats, 8 * sizeof(unsigned long long)));
  }
  if (want_stats) IMP_CHECK_HIP(hipMemsetAsync(stats, 0, 8 * sizeof(unsigned long long), stream()));
  {
    IMP_PROF(name);
    kern<<<grid, BLOCK, lds, stream()>>>(C->order.data(), first, count, C->indptr.data(), C->indices.data(), C->data.data(), X, Y,
                                        A0, cg_steps, want_stats ? stats : nullptr);
    IMP_CHECK_HIP(hipGetLastError());
  }
  if (want_stats) {
    unsigned long long h[8];
    IMP_CHECK_HIP(hipMemcpyAsync(h, stats, sizeof(h), hipMemcpyDeviceToHost, stream()));
    sync();
    double waves = (double)grid * (BLOCK / 64);
    fprintf(stderr, "[cg-stats] %s grid=%d waves=%.0f groups/wave=%.1f  per-group ticks: meta %.0f gather %.0f compute %.0f barrier %.0f\n",
            name, grid, waves, h[4] / waves, (double)h[0] / h[4], (double)h[1] / h[4], (double)h[2] / h[4], (double)h[3] / h[4]);
  }
}

template <int F>
static void run_classes(const imp_csr *C, float *X, const float *Y, const float *A0, int cg_steps) {
  const int32_t *b = C->bin_start;  // classes: 1 (256,512]  2 (128,256]  3 (64,128]  4 (32,64]  5 (16,32]  6 (0,16]
  // 512-thread workgroups: two fit per CU (LDS 68 KiB, 4 waves/SIMD) and run out of phase, so one gathers
  // while the other computes; the 16-wave team needs the whole CU
  static const bool big = getenv("IMP_TEAM_1024") != nullptr;
  launch_team<F, 16, 1024>(C, b[1], b[2] - b[1], X, Y, A0, cg_steps, "als_cg_team16_rows");
  if (big) {
    launch_team<F, 8, 1024>(C, b[2], b[3] - b[2], X, Y, A0, cg_steps, "als_cg_team8_rows");
    launch_team<F, 4, 1024>(C, b[3], b[4] - b[3], X, Y, A0, cg_steps, "als_cg_team4_rows");
    launch_team<F, 2, 1024>(C, b[4], b[5] - b[4], X, Y, A0, cg_steps, "als_cg_team2_rows");
  } else {
    launch_team<F, 8, 512>(C, b[2], b[3] - b[2], X, Y, A0, cg_steps, "als_cg_team8_rows");
    launch_team<F, 4, 512>(C, b[3], b[4] - b[3], X, Y, A0, cg_steps, "als_cg_team4_rows");
    launch_team<F, 2, 512>(C, b[4], b[5] - b[4], X, Y, A0, cg_steps, "als_cg_team2_rows");
  }
  launch_group<F, true, 32>(C, b[5], b[6] - b[5], X, Y, A0, cg_steps, "als_cg_short_rows");
  launch_group<F, true, 16>(C, b[6], b[7] - b[6], X, Y, A0, cg_steps, "als_cg_short16_rows");
}

// mid (wave teams, resident tiles) and short (one wave per row, MFMA gramian product) classes for f = 64 or 128
void least_squares_cg_group(const imp_csr *C, float *X, const float *Y, const float *A0, int f, int cg_steps) {
  if (f == 128) run_classes<128>(C, X, Y, A0, cg_steps);
  else if (f == 64) run_classes<64>(C, X, Y, A0, cg_steps);
  else throw std::invalid_argument("least_squares_cg_group: f must be 64 or 128");
}

}  // namespace imp
