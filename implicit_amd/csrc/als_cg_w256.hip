// CG half sweep at f = 256 (BASELINE configs[4]; the reference's published f = 192 / 256 columns ride it zero-padded): rows of up to
// 256 nonzeros with their gathered factor rows RESIDENT in registers for all 1 + cg_steps passes.
//
// What it replaces.  The reference's kernel (implicit/gpu/als.cu:23-111) walks a row's nonzeros once per CG pass; round 2's
// als_cg_f256_kernel (als_cg.hip) did the same with register tiles re-gathered every pass and the 256 KB gramian staged through
// the LDS slice by slice: 4x the gather traffic of the roofline, measured at the fabric limit (8.5 TB/s of gathers = 0.19-0.27 of
// the HBM roofline).  At f = 256 a gathered row is 1 KB, a wavefront's registers hold 16 of them, and the gramian fits no LDS,
// so the f = 128 team kernels (als_cg_qf.hip) do not carry over.  This kernel does:
//
//   * one 1024-thread workgroup per CU = 16 wavefronts, each keeping a tile of 16 gathered rows in the quarter layout of
//     als_qtile.h (lane (g, m) of the wave's four 16-lane groups holds factors 64 p + 4 m + c of entries 4 q + g: 64 VGPRs);
//   * a row is solved by a TEAM of WPR = 1 / 2 / 4 / 8 / 16 wavefronts (<= 16 / 32 / 64 / 128 / 256 nonzeros, dealt in even
//     shares of whole 4-entry steps), R = 16 / WPR rows per workgroup in LOCK STEP;
//   * the dense part (YtY + reg I) [v_0 .. v_{R-1}] of a pass is ONE matrix-core product for the R rows: the gramian is split
//     once per call into fp16 halves G 2^k = H + L (w256_prep_kernel; k brings max |G| to 2^13..2^14, so the low half of every
//     element above 2^-17 of the largest is a normal fp16 number: H + L is G to 2^-24 relative), stored in the fragment order
//     of v_mfma_f32_16x16x32_f16; every wavefront owns the 16 output factors 16 w .. 16 w + 15 and streams its 16 x 256 slice of H
//     and L from L2 straight into registers, except for the part of H that fits the LDS beside the R rows' vectors -- 4 of the 8
//     k-steps at R = 16, 6 at R = 8, 7 at R = 4, all of them from R = 2 down -- which it keeps there for the whole launch (8 .. 12 KB
//     per wavefront and pass from L2, eight loads in flight); the operand vectors are published by the
//     rows' leader wavefronts as fp16 halves too, scaled per row and pass to 2^14.  All four products H h, H l, L h, L l, fp32
//     accumulation: exact in the 22-bit halves;
//   * the tile part runs on the VALU as in the f = 128 kernels (dot over a DPP row, weights from an LDS table, packed FMAs),
//     the team's partial vectors and the matrix-core result meet in the LDS, and the row's LEADER wavefront does the CG update
//     (als.cu:45-109 / _als.pyx:179-244 step by step) on x, r, p kept in the LDS, lane l owning factors 4 l .. 4 l + 3;
//   * two workgroup barriers per pass; in the passes after the first, half of the wavefronts of every SIMD run their tile
//     entries first and the others the product first, so that the vector and the matrix pipe work side by side.
//
// Rows of 257 .. 512 nonzeros stay on als_cg_f256_kernel, longer ones on the segment-parallel streamed kernels (als_cg.hip).
// IMP_F256_OLD=1 keeps the round-2 kernel for every row (A/B, parity).
#include "als_qf_common.h"
#include "common.h"
#include "wave_ops.h"

namespace imp {

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

struct W256 {
  static constexpr int F = 256;
  static constexpr int LDO = F + 4;   // floats per row of the product's result (conflict-free b128 writes by 16 columns)
  static constexpr size_t kFragHalves = (size_t)16 * 8 * 2 * 512;  // [wave][k-step][H | L][lane][8]
};
// LDS map of a workgroup that solves R rows at a time (bytes).  What the R rows do not need goes to H: a wavefront keeps the
// first KRES of the 8 k-steps of its slice of H in the LDS for the whole launch (all of them from teams of 8 up) and streams
// only the rest -- the product is bound by what it reads from L2 per pass.
template <int R> struct W256Lds {
  static constexpr int F = W256::F, LDO = W256::LDO;
  static constexpr int oV = 0;                          // [R rows][256] f32    operand of the pass, natural factor order
  static constexpr int oR = oV + R * F * 4;             // [R rows][256] f32    residual
  static constexpr int oSp = oR + R * F * 4;            // [16 waves][256] f32  tile part of each wavefront
  // R <= 8: the two halves of an operand are two COLUMNS of one MFMA (2 j: high half of row j, 2 j + 1: low half), the leader adds
  // the two result columns; R = 16: separate operand sets (high | low), four products per k-step
  static constexpr int COLS = R <= 8 ? 2 * R : 16;
  static constexpr int oOut = oSp + 16 * F * 4;         // [COLS][LDO] f32      matrix-core product (scaled units)
  // the operands of the matrix cores in FRAGMENT order: [set][k-step][lane = column + 16 kq][8 halves], all 16 columns of a set
  // whatever R is.  A row-major [column][k] array with padded rows -- the layout of the f = 128 short-row kernel -- is not
  // conflict-free under the real lane groups of ds_read_b128 ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md), and re-reading
  // column n % 2R for the unused columns made it worse: SQ_LDS_BANK_CONFLICT read 0.77 of the LDS's active cycles
  // (profiles/r05_w256_counters.txt).  A lane-linear 1 KB block per k-step is conflict-free by construction.
  static constexpr int SETS = R <= 8 ? 1 : 2;           // paired columns: one set; R = 16: high halves | low halves
  static constexpr int oPb = oOut + COLS * LDO * 4;     // [SETS][8 k-steps][64 lanes][8] f16
  static constexpr int oCw = oPb + SETS * 8 * 1024;     // [16 waves][32] f32   |c| - 1, c+ of the resident entries
  static constexpr int oAct = oCw + 16 * 32 * 4;        // [16 rows] int        row still iterating
  static constexpr int oH = oAct + 16 * 4;              // [16 waves][KRES][64 lanes][8] f16   resident part of H (fragment order)
  static constexpr int kLdsMax = 160 * 1024;
  static constexpr int KRES = (kLdsMax - oH) / (16 * 1024) >= 8 ? 8 : (kLdsMax - oH) / (16 * 1024);
  static constexpr int bytes = oH + 16 * KRES * 1024;
  static_assert(oH % 16 == 0 && oPb % 16 == 0 && KRES >= 1, "LDS map");
};

#ifdef W256_STATS  // timing-only build (build_variant): shader-clock cycles per phase, summed over the waves of a launch
__device__ unsigned long long g_w256_stats[8];  // [0] product [2] tile entries [3] wait at B2 [4] leader update [5] wait at B1 [6] group start [7] lifetime
#endif
__device__ __forceinline__ int wave_of(unsigned tid) { return __builtin_amdgcn_readfirstlane((int)(tid >> 6)); }
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// G 2^k = H + L in fragment order; hdr[0] = 2^-k.  One workgroup: 64 K elements.
__global__ __launch_bounds__(1024) void w256_prep_kernel(const float *__restrict__ A0, _Float16 *__restrict__ gfrag, float *__restrict__ hdr) {
  __shared__ float red[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float m = 0.f;
  for (int e = tid; e < 256 * 256; e += 1024) m = fmaxf(m, fabsf(A0[e]));
  m = wave_allmax(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) m = fmaxf(m, red[w]);
  int k = 0;
  if (m > 0.f && m < 3.0e38f) k = min(max(13 - ((int)((__float_as_uint(m) >> 23) & 255u) - 127), -100), 100);
  const float s = __uint_as_float((unsigned)(127 + k) << 23);
  if (tid == 0) hdr[0] = __uint_as_float((unsigned)(127 - k) << 23);
  for (int idx = tid; idx < 16 * 8 * 64; idx += 1024) {
    const int ln = idx & 63, ks = (idx >> 6) & 7, ti = idx >> 9;
    const float *src = A0 + (size_t)(16 * ti + (ln & 15)) * 256 + 32 * ks + 8 * (ln >> 4);
    h8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = src[e] * s;
      hi[e] = (_Float16)v;
      lo[e] = (_Float16)(v - (float)hi[e]);
    }
    _Float16 *dst = gfrag + ((size_t)(ti * 8 + ks) * 2) * 512 + ln * 8;
    *reinterpret_cast<h8 *>(dst) = hi;
    *reinterpret_cast<h8 *>(dst + 512) = lo;
  }
}

template <int WPR>
__global__ __launch_bounds__(1024) void als_cg_w256_kernel(const int32_t *__restrict__ order, int first, int count,
                                                           const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                           const float *__restrict__ data, float *__restrict__ X,
                                                           const float *__restrict__ Y, const _Float16 *__restrict__ gfrag,
                                                           const float *__restrict__ hdr, int cg_steps,
                                                           int ko) {  // ko: timing-only knock-outs (IMP_W256_KO), 0 in production
  constexpr int F = W256::F, R = 16 / WPR, LDO = W256::LDO;
  using M = W256Lds<R>;
  constexpr int KRES = M::KRES;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float *const Vs = reinterpret_cast<float *>(lds + M::oV);
  float *const Rs = reinterpret_cast<float *>(lds + M::oR);
  float *const Sp = reinterpret_cast<float *>(lds + M::oSp);
  float *const Out = reinterpret_cast<float *>(lds + M::oOut);
  _Float16 *const Pb = reinterpret_cast<_Float16 *>(lds + M::oPb);
  float *const Cw = reinterpret_cast<float *>(lds + M::oCw);
  int *const Act = reinterpret_cast<int *>(lds + M::oAct);
  h8 *const Hres = reinterpret_cast<h8 *>(lds + M::oH) + (size_t)wave_of(threadIdx.x) * KRES * 64;  // this wave's resident blocks
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = wave / WPR, s = wave % WPR;  // row of the group, position in its team
  const bool leader = s == 0;
#ifdef W256_STATS
  unsigned long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last = __builtin_amdgcn_s_memtime();
  const unsigned long long t_begin = t_last;
  auto tick = [&](int slot) {
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long now = __builtin_amdgcn_s_memtime();
    tk[slot] += now - t_last;
    t_last = now;
    __builtin_amdgcn_sched_barrier(0);
  };
#else
  auto tick = [](int) {};
#endif
  const float ginv = hdr[0];
  for (int e = threadIdx.x; e < M::SETS * 8 * 1024 / 4; e += 1024) reinterpret_cast<unsigned *>(Pb)[e] = 0u;  // unused columns
  float *const vrow = Vs + j * F, *const cw = Cw + wave * 32;
  const h8 *const gfw = reinterpret_cast<const h8 *>(gfrag) + (size_t)wave * 16 * 64;  // this wave's 16 blocks of 64 x 16 bytes
#pragma unroll
  for (int ks = 0; ks < KRES; ++ks) Hres[ks * 64 + lane] = gfw[(ks * 2) * 64 + lane];  // wave-private: read back by the same lanes
  __syncthreads();

  // ---- matrix-core part: Out[n][16 wave + ..] = sum_k (H + L)[16 wave + i][k] (ph + pl)[n][k], three products ------------------
  // k-steps 0 .. KRES-1 take H from the LDS; per pass a wavefront streams L (8 blocks) and the rest of H from L2, eight loads in
  // flight.  Four products: H h, H l, L h and L l (the last is 2^-22 of the sum: with it the product is exact in the 22-bit halves)
  // The MFMA's 16 columns are the group's R rows (2 R with paired columns); the columns beyond stay zero and nobody reads
  // their results.
  auto product = [&]() __attribute__((always_inline)) {
    if (ko & 1) return;
    const int ln = opaque(lane);
    const int n = ln & 15, kq = ln >> 4;
    const h8 *gl = gfw + ln;
    const h8 *hres = Hres + ln;
    constexpr bool PAIRED = R <= 8;  // both halves of the operands in one B fragment (columns 2 j, 2 j + 1)
    const h8 *bfrag = reinterpret_cast<const h8 *>(Pb) + ln;  // block ks of set t: bfrag[(t * 8 + ks) * 64]
    // two accumulators: the products with L and with H form independent chains (a dependent MFMA waits out its predecessor)
    v4f acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
    // The streamed blocks in the order they are used -- for k-step ks: H[ks] if it is not resident, then L[ks] -- through a ring of
    // NB registers-quads: NB loads in flight, the slot of a block is re-filled as soon as the block has been multiplied.  (All
    // eight L blocks plus the streamed H at once, the first form, left no registers for anything else: the R = 8 kernel
    // spilled four registers of the TILE and reloaded them in every pass.)
    constexpr int NH = 8 - KRES;       // k-steps of H that are streamed: KRES .. 7
    constexpr int NF = 8 + NH, NB = 6;  // blocks per pass, ring size
    h8 ring[NB];
    // block i of the sequence: k-steps below KRES contribute one block (L), the others two (H, then L)
    auto block_of = [](int i, int &ks, int &is_h) {
      if (i < KRES) {
        ks = i, is_h = 0;
      } else {
        ks = KRES + (i - KRES) / 2, is_h = ((i - KRES) & 1) == 0;
      }
    };
    auto fetch = [&](auto Ic) {
      constexpr int i = decltype(Ic)::value;
      int ks = 0, is_h = 0;
      block_of(i, ks, is_h);
#ifdef W256_KO_NOLOAD  // timing-only builds (implicit_amd/_build.py build_variant)
      ring[i % NB] = hres[0];
#else
      ring[i % NB] = gl[(ks * 2 + (is_h ? 0 : 1)) * 64];
#endif
    };
    static_for<(NB < NF ? NB : NF)>([&](auto Ic) { fetch(Ic); });
    // (without the fence the scheduler sinks every load to its first use to save registers: two loads in flight, an L2 round
    // trip per k-step)
    __builtin_amdgcn_sched_barrier(0);
    auto step = [&](int ks, const h8 &H, const h8 &L) {
#ifdef W256_KO_NOB
      const h8 bh = L, bl = H;
#else
      const h8 bh = bfrag[ks * 64];
      h8 bl = bh;
      if constexpr (!PAIRED) bl = bfrag[(8 + ks) * 64];
#endif
#ifdef W256_KO_NOMFMA
      asm volatile("" ::"v"(L), "v"(H), "v"(bh), "v"(bl));
#else
      if constexpr (PAIRED) {
        acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(L, bh, acc2, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(H, bh, acc, 0, 0, 0);
      } else {  // (one chain: the R = 16 form has no registers for a second accumulator)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(L, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(L, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(H, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(H, bh, acc, 0, 0, 0);
      }
#endif
    };
    static_for<8>([&](auto Kc) {
      constexpr int ks = decltype(Kc)::value;
      // index of this k-step's L block in the sequence (its H block, if streamed, is the one before)
      constexpr int iL = ks < KRES ? ks : KRES + 2 * (ks - KRES) + 1;
      if constexpr (ks < KRES) step(ks, hres[ks * 64], ring[iL % NB]);
      else step(ks, ring[(iL - 1) % NB], ring[iL % NB]);
      // re-fill the slots just used (fenced: the re-fill may not move above the MFMAs that read the slot, nor sink to its use)
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (ks >= KRES && iL - 1 + NB < NF) fetch(idx_t<(iL - 1 + NB < NF ? iL - 1 + NB : 0)>{});
      if constexpr (iL + NB < NF) fetch(idx_t<(iL + NB < NF ? iL + NB : 0)>{});
      __builtin_amdgcn_sched_barrier(0);
    });
    if (n < M::COLS)
      *reinterpret_cast<float4 *>(Out + (size_t)n * LDO + 16 * wave + 4 * kq) =
          make_float4(acc[0] + acc2[0], acc[1] + acc2[1], acc[2] + acc2[2], acc[3] + acc2[3]);
  };

  const int groups = (count + R - 1) / R, g_step = gridDim.x;
  // row of this team in group gg: id, nonzero range, this wave's share [k0, k0 + cnt)
  struct Share {
    int u, k0, cnt;
    bool valid;
  };
  auto share_of = [&](int gg) {
    Share sh;
    const int ri = gg * R + j;
    sh.valid = gg < groups && ri < count;
    sh.u = __builtin_amdgcn_readfirstlane(order[first + min(ri, count - 1)]);
    const int rb = __builtin_amdgcn_readfirstlane(indptr[sh.u]), re = __builtin_amdgcn_readfirstlane(indptr[sh.u + 1]);
    const int n_row = sh.valid ? re - rb : 0;
    const int per = 4 * ((n_row + 4 * WPR - 1) / (4 * WPR));  // <= 16: even shares of whole 4-entry steps
    sh.k0 = rb + s * per;
    sh.cnt = max(0, min(per, rb + n_row - sh.k0));
    return sh;
  };
  // entry min(t, cnt - 1) of a share in lane t (and t + 16, ..): two registers that stay in flight during the previous group
  auto fetch_entries16 = [&](const Share &sh, int &col, float &c) {
    col = 0, c = -1.f;
    if (sh.cnt > 0) {  // wave-uniform
      const int k = sh.k0 + min(opaque(lane) & 15, sh.cnt - 1);
      col = indices[k];
      c = data[k];
    }
  };
  // tile slots 2 P, 2 P + 1 (entries 8 P .. 8 P + 7 of a share) from the staged entries: weights to the table, eight gathers per
  // lane.  Entries past the count inside the pair repeat the share's last row with weight 0.
  f32x2 y[4][8];
  auto gather_pair = [&](auto Pc, int col, float c, int n_ent) __attribute__((always_inline)) {
    constexpr int P = decltype(Pc)::value;
    const int ln = opaque(lane);
    if ((ln >> 3) == P) {  // lanes 8 P .. 8 P + 7 hold the pair's entries
      const bool ok = ln < n_ent;
      cw[ln] = ok ? fabsf(c) - 1.f : 0.f;
      cw[16 + ln] = ok ? fmaxf(c, 0.f) : 0.f;
    }
    const int src = 4 * (ln >> 4);
#pragma unroll
    for (int q = 2 * P; q < 2 * P + 2; ++q) {
      const unsigned cq = (unsigned)__builtin_amdgcn_ds_bpermute(src + 16 * q, col);
      const float *p = Y + (size_t)cq * F + 4 * (ln & 15);
#pragma unroll
      for (int pc = 0; pc < 4; ++pc) {
        const float4 v = *reinterpret_cast<const float4 *>(p + 64 * pc);
        y[q][2 * pc] = f32x2{v.x, v.y}, y[q][2 * pc + 1] = f32x2{v.z, v.w};
      }
    }
  };
  // cur: the group being solved; nxt: the one after it.  ent_*: the staged entries of the group whose tile is not gathered yet --
  // cur's until its tile is, nxt's from then on (two registers in flight while cur iterates).  The last pass of a group ROLLS
  // nxt's tile in, pair by pair, as soon as a pair of registers has been used for the last time: the gathers (a quarter of a
  // group's time when they start at its top, with the CU idle: one workgroup per CU) then overlap the leader's last update.
  if (cg_steps <= 0) return;  // nothing is stored without a CG step (_als.pyx:208-244)
  Share cur = share_of(blockIdx.x), nxt = cur;
  int ent_col;
  float ent_c;
  fetch_entries16(cur, ent_col, ent_c);
  float4 x_next = make_float4(0.f, 0.f, 0.f, 0.f);  // leader: the next group's iterate, requested ahead of the rolling gathers
  // the first group of the workgroup: plain start; every later tile rolls in during its predecessor's last pass
  if (!(ko & 4)) {
    static_for<2>([&](auto Pc) {
      if (8 * decltype(Pc)::value < cur.cnt) gather_pair(Pc, ent_col, ent_c, cur.cnt);  // wave-uniform
    });
  }
  nxt = share_of(blockIdx.x + g_step);
  fetch_entries16(nxt, ent_col, ent_c);
  if (leader && cur.valid) x_next = *reinterpret_cast<const float4 *>(X + (size_t)cur.u * F + 4 * opaque(lane));
  for (int g = blockIdx.x; g < groups; g += g_step) {
    const bool valid = cur.valid;
    const int u = cur.u, cnt = cur.cnt;
    // from here on: the tile is cur's, ent_* are nxt's entries, x_next is cur's iterate

    // ---- tile part of a pass: Sp[wave] = sum over the resident entries of w y ------------------------------------------------
    //   FIRST: w = c+ - (|c|-1) y.x   else: w = (|c|-1) y.p      (_als.pyx:190-201, 214-222)
    //   LAST : the registers (and table slots) of a pair are re-filled with nxt's entries once the pair is done
    auto tile_pass = [&](auto first_c, auto last_c, bool on) __attribute__((always_inline)) {
      constexpr bool FIRST = decltype(first_c)::value, LAST = decltype(last_c)::value;
      const int ln = opaque(lane);
      int roll_col = 0;
      float roll_c = 0.f;
      if constexpr (LAST) {  // one wait for the staged entries, before any rolling gather
        roll_col = opaque(ent_col);
        roll_c = __int_as_float(opaque(__float_as_int(ent_c)));
      }
      const bool work = on && !(ko & 2);
      f32x2 ve[8], ae[8];
      {  // the operand, expanded: slot e of lane (g, m) is factor 64 (e / 4) + 4 m + (e & 3)
        const float *vp = vrow + 4 * (ln & 15);
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) {
          const float4 t = *reinterpret_cast<const float4 *>(vp + 64 * pc);
          ve[2 * pc] = f32x2{t.x, t.y}, ve[2 * pc + 1] = f32x2{t.z, t.w};
        }
      }
#pragma unroll
      for (int h = 0; h < 8; ++h) ae[h] = f32x2{0.f, 0.f};
      const float *cwg = cw + (ln >> 4);
      auto partial = [&](int q) {
        f32x2 sacc = y[q][0] * ve[0];
#pragma unroll
        for (int h = 1; h < 8; ++h) sacc = __builtin_elementwise_fma(y[q][h], ve[h], sacc);
        return sacc.x + sacc.y;
      };
      auto axpy = [&](int q, float w) {
        const f32x2 w2 = {w, w};
#pragma unroll
        for (int h = 0; h < 8; ++h) ae[h] = __builtin_elementwise_fma(w2, y[q][h], ae[h]);
      };
      static_for<2>([&](auto Pc) {
        constexpr int P = decltype(Pc)::value;
        if (work && 8 * P < cnt) {  // wave-uniform
          const float cm1_0 = cwg[8 * P], cm1_1 = cwg[8 * P + 4];
          float cp_0 = 0.f, cp_1 = 0.f;
          if constexpr (FIRST) cp_0 = cwg[16 + 8 * P], cp_1 = cwg[16 + 8 * P + 4];
          const float uu = reduce_pair(partial(2 * P), partial(2 * P + 1));
          const float w0 = FIRST ? fmaf(-cm1_0, row_bcast_from<0>(uu), cp_0) : cm1_0 * row_bcast_from<0>(uu);
          const float w1 = FIRST ? fmaf(-cm1_1, row_bcast_from<8>(uu), cp_1) : cm1_1 * row_bcast_from<8>(uu);
          axpy(2 * P, w0);
          axpy(2 * P + 1, w1);
        }
        if constexpr (LAST) {
          // ONE place per pair where its registers are re-filled, whether or not the pair did any work (two places -- an `else`
          // for the waves whose row had stopped -- made the compiler keep a second copy of the tile: 40 registers spilled);
          // fenced: hoisted above the pair's last FMAs the gathers would need a second set of registers as well
          __builtin_amdgcn_sched_barrier(0);
          if (8 * P < nxt.cnt && !(ko & 4)) gather_pair(Pc, roll_col, roll_c, nxt.cnt);
          __builtin_amdgcn_sched_barrier(0);
        }
      });
      // reduce-scatter across the four groups: expanded slot e of lane (g, m) is factor 64 (e >> 2) + 4 m + (e & 3); group g
      // ends with slots 4 g .. 4 g + 3, i.e. lane l with factors 4 l .. 4 l + 3 (all zeros for a wave without work)
      float hsum[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        hsum[2 * i] = swap32_sum(ae[i].x, ae[i + 4].x);
        hsum[2 * i + 1] = swap32_sum(ae[i].y, ae[i + 4].y);
      }
      const float4 out = make_float4(swap16_sum(hsum[0], hsum[4]), swap16_sum(hsum[1], hsum[5]), swap16_sum(hsum[2], hsum[6]),
                                     swap16_sum(hsum[3], hsum[7]));
      *reinterpret_cast<float4 *>(Sp + (size_t)wave * F + 4 * ln) = out;
    };

    // ---- leader: publish an operand (fp32 for the tile entries, scaled fp16 halves for the matrix cores) ----------------------
    float inv_s = 1.f;  // 1 / (operand scale) of what this leader published last
    auto publish = [&](float4 v, bool on) {
      const int ln = opaque(lane);
      if (!on) v = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4 *>(vrow + 4 * ln) = v;
      const float mx = wave_allmax(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
      int k = 0;
      if (mx > 0.f && mx < 3.0e38f) k = min(max(14 - ((int)((__float_as_uint(mx) >> 23) & 255u) - 127), -100), 100);
      const float sc = __uint_as_float((unsigned)(127 + k) << 23);
      inv_s = __uint_as_float((unsigned)(127 - k) << 23);
      const float a0 = v.x * sc, a1 = v.y * sc, a2 = v.z * sc, a3 = v.w * sc;
      h4 hi, lo;
      hi[0] = (_Float16)a0, hi[1] = (_Float16)a1, hi[2] = (_Float16)a2, hi[3] = (_Float16)a3;
      lo[0] = (_Float16)(a0 - (float)hi[0]), lo[1] = (_Float16)(a1 - (float)hi[1]);
      lo[2] = (_Float16)(a2 - (float)hi[2]), lo[3] = (_Float16)(a3 - (float)hi[3]);
      // lane l holds k = 4 l .. 4 l + 3: k-step l / 8, lane group kq = (l % 8) / 2, elements 4 (l % 2) ..; column c of a set sits
      // in lane c + 16 kq of the block
      constexpr bool PAIRED = R <= 8;
      const int c_hi = PAIRED ? 2 * j : j, c_lo = PAIRED ? 2 * j + 1 : j;
      _Float16 *blk = Pb + (size_t)(ln >> 3) * 512 + (size_t)(16 * ((ln & 7) >> 1)) * 8 + 4 * (ln & 1);
      *reinterpret_cast<h4 *>(blk + c_hi * 8) = hi;
      *reinterpret_cast<h4 *>(blk + (PAIRED ? 0 : 8 * 512) + c_lo * 8) = lo;
    };
    auto dot4 = [](const float4 &a, const float4 &b) { return wave_allsum(fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w)))); };
    // dense product of this row + the team's tile parts, lane l: factors 4 l ..
    auto collect = [&](float4 &dense, float4 &sparse) {
      const int ln = opaque(lane);
      const float sc = ginv * inv_s;
      float4 o;
      if constexpr (R <= 8) {  // the high-half and the low-half column of this row
        const float4 oh = *reinterpret_cast<const float4 *>(Out + (size_t)(2 * j) * LDO + 4 * ln);
        const float4 ol = *reinterpret_cast<const float4 *>(Out + (size_t)(2 * j + 1) * LDO + 4 * ln);
        o = make_float4(oh.x + ol.x, oh.y + ol.y, oh.z + ol.z, oh.w + ol.w);
      } else {
        o = *reinterpret_cast<const float4 *>(Out + (size_t)j * LDO + 4 * ln);
      }
      dense = make_float4(o.x * sc, o.y * sc, o.z * sc, o.w * sc);
      sparse = *reinterpret_cast<const float4 *>(Sp + (size_t)wave * F + 4 * ln);
#pragma unroll 4
      for (int t = 1; t < WPR; ++t) {  // (all 15 reads of a 16-wave team in flight at once spill)
        const float4 q4 = *reinterpret_cast<const float4 *>(Sp + (size_t)(wave + t) * F + 4 * ln);
        sparse.x += q4.x, sparse.y += q4.y, sparse.z += q4.z, sparse.w += q4.w;
      }
    };

    float *const xrow = X + (size_t)u * F;
    float rsold = 0.f;
    bool active = false;
    float4 x4 = x_next;  // the iterate (leader): lane l, factors 4 l ..
    if (leader) {
      publish(x4, valid);
      if (lane == 0) Act[j] = valid ? 1 : 0;
    }
    tick(6);
    lds_barrier();
    tick(5);
    // ---- pass 0: r = -(A0 x) + sum_k (c+ - (|c|-1) y.x) y        (_als.pyx:187-201); the product first: it needs no gather ----
    product();
    tick(0);
    tile_pass(std::true_type{}, std::false_type{}, valid);
    tick(2);
    lds_barrier();
    tick(3);
    if (leader) {
      const int ln = opaque(lane);
      float4 dn, sp4;
      collect(dn, sp4);
      const float4 r4 = make_float4(sp4.x - dn.x, sp4.y - dn.y, sp4.z - dn.z, sp4.w - dn.w);
      rsold = dot4(r4, r4);
      active = valid && rsold >= 1e-20f;  // else: x untouched (_als.pyx:206)
      *reinterpret_cast<float4 *>(Rs + (size_t)j * F + 4 * ln) = r4;
      publish(r4, active);
      if (lane == 0) Act[j] = active ? 1 : 0;
    }
    tick(4);
    lds_barrier();
    tick(5);
    const bool tiles_first = ((wave >> 2) & 1) != 0;
    for (int it = 0; it + 1 < cg_steps; ++it) {  // all steps but the last
      const bool on = __builtin_amdgcn_readfirstlane(Act[j]) != 0;
      if (tiles_first) {
        tile_pass(std::false_type{}, std::false_type{}, on);
        tick(2);
        product();
        tick(0);
      } else {
        product();
        tick(0);
        tile_pass(std::false_type{}, std::false_type{}, on);
        tick(2);
      }
      lds_barrier();
      tick(3);
      if (leader && active && !(ko & 8)) {  // wave-uniform
        const int ln = opaque(lane);
        float4 dn, sp4;
        collect(dn, sp4);
        const float4 Ap = make_float4(dn.x + sp4.x, dn.y + sp4.y, dn.z + sp4.z, dn.w + sp4.w);
        const float4 p4 = *reinterpret_cast<const float4 *>(vrow + 4 * ln);
        const float alpha = rsold * __builtin_amdgcn_rcpf(dot4(p4, Ap));
        x4.x = fmaf(alpha, p4.x, x4.x), x4.y = fmaf(alpha, p4.y, x4.y), x4.z = fmaf(alpha, p4.z, x4.z), x4.w = fmaf(alpha, p4.w, x4.w);
        float4 r4 = *reinterpret_cast<const float4 *>(Rs + (size_t)j * F + 4 * ln);
        r4.x = fmaf(-alpha, Ap.x, r4.x), r4.y = fmaf(-alpha, Ap.y, r4.y), r4.z = fmaf(-alpha, Ap.z, r4.z), r4.w = fmaf(-alpha, Ap.w, r4.w);
        const float rsnew = dot4(r4, r4);
        if (rsnew < 1e-20f) {  // the oracle breaks here (_als.pyx:235)
          *reinterpret_cast<float4 *>(xrow + 4 * ln) = x4;
          active = false;
          publish(make_float4(0.f, 0.f, 0.f, 0.f), false);
        } else {
          const float beta = rsnew * __builtin_amdgcn_rcpf(rsold);
          const float4 pn = make_float4(fmaf(beta, p4.x, r4.x), fmaf(beta, p4.y, r4.y), fmaf(beta, p4.z, r4.z), fmaf(beta, p4.w, r4.w));
          *reinterpret_cast<float4 *>(Rs + (size_t)j * F + 4 * ln) = r4;
          publish(pn, true);
          rsold = rsnew;
        }
        if (lane == 0) Act[j] = active ? 1 : 0;
      }
      tick(4);
      lds_barrier();
      tick(5);
    }
    // The last step stands outside the loop (the compiler must see that nothing of this group follows it): the product first in
    // every wave (its loads must not queue behind the gathers), then the tile entries with nxt's tile rolling in behind them;
    // the next iterate is requested ahead of the gathers (loads return in order).  Only the step's x update is evaluated: its
    // r, rsnew and p are never read (_als.pyx:226-241).
    {
      const bool on = __builtin_amdgcn_readfirstlane(Act[j]) != 0;
      product();
      tick(0);
      float4 xn = make_float4(0.f, 0.f, 0.f, 0.f);
      if (leader && nxt.valid) xn = *reinterpret_cast<const float4 *>(X + (size_t)nxt.u * F + 4 * opaque(lane));
      tile_pass(std::false_type{}, std::true_type{}, on);
      tick(2);
      lds_barrier();
      tick(3);
      if (leader && active && !(ko & 8)) {  // wave-uniform
        const int ln = opaque(lane);
        float4 dn, sp4;
        collect(dn, sp4);
        const float4 Ap = make_float4(dn.x + sp4.x, dn.y + sp4.y, dn.z + sp4.z, dn.w + sp4.w);
        const float4 p4 = *reinterpret_cast<const float4 *>(vrow + 4 * ln);
        const float alpha = rsold * __builtin_amdgcn_rcpf(dot4(p4, Ap));
        x4.x = fmaf(alpha, p4.x, x4.x), x4.y = fmaf(alpha, p4.y, x4.y), x4.z = fmaf(alpha, p4.z, x4.z), x4.w = fmaf(alpha, p4.w, x4.w);
        *reinterpret_cast<float4 *>(xrow + 4 * ln) = x4;
      }
      x_next = xn;
      tick(4);
      lds_barrier();
      tick(5);
    }
    cur = nxt;
    nxt = share_of(g + 2 * g_step);
    fetch_entries16(nxt, ent_col, ent_c);
  }
#ifdef W256_STATS
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 7; ++k) atomicAdd(&g_w256_stats[k], tk[k]);
    atomicAdd(&g_w256_stats[7], (unsigned long long)__builtin_amdgcn_s_memtime() - t_begin);
  }
#endif
}

template <int WPR>
void launch_w256_class(const imp_csr *C, int first, int count, float *X, const float *Y, const _Float16 *gfrag, const float *hdr,
                       int cg_steps, const char *name) {
  if (count <= 0) return;
  constexpr int R = 16 / WPR;
  auto kern = als_cg_w256_kernel<WPR>;
  constexpr int lds_bytes = W256Lds<R>::bytes;
  IMP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  const int groups = (count + R - 1) / R;
  // one workgroup per CU is resident; four times as many with proportionally smaller shares even out the end of the launch
  const int grid = std::min(groups, ctx().num_cus * std::max(4, ctx().oversub));
  constexpr int ko = 0;  // (timing-only knock-outs: 1 product, 2 tile entries, 4 gathers, 8 leader update)
  IMP_PROF(name);
  kern<<<grid, 1024, lds_bytes, stream()>>>(C->order.data(), first, count, C->indptr.data(), C->indices.data(), C->data.data(), X, Y,
                                              gfrag, hdr, cg_steps, ko);
  IMP_CHECK_HIP(hipGetLastError());
#ifdef W256_STATS
  {
    unsigned long long h[8], z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    IMP_CHECK_HIP(hipStreamSynchronize(stream()));
    IMP_CHECK_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_w256_stats), sizeof(h)));
    IMP_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_w256_stats), z, sizeof(z)));
    const double wg = (double)grid * 16.0, gr = (double)groups * 16.0;  // waves launched, wave-groups processed
    fprintf(stderr, "[w256-stats] %s groups=%d  cycles per wave and group: product %.0f  tiles %.0f  wait-B2 %.0f  update %.0f  wait-B1 %.0f  "
                    "start %.0f | lifetime per wave %.0f\n", name, groups, h[0] / gr, h[2] / gr, h[3] / gr, h[4] / gr, h[5] / gr, h[6] / gr, h[7] / wg);
  }
#endif
}

}  // namespace

bool w256_enabled() {
  static const bool on = getenv("IMP_F256_OLD") == nullptr;
  return on;
}

// rows of 1 .. 256 nonzeros (schedule classes 2 .. 6) of an f = 256 half sweep
void least_squares_cg_w256(const imp_csr *C, float *X, const float *Y, const float *A0, int cg_steps) {
  const int32_t *b = C->bin_start;
  if (b[7] - b[2] <= 0) return;
  auto &ws = ctx().w256_ws;
  const size_t need = W256::kFragHalves / 2 + 16;  // floats: fragments (fp16) + header
  if (ws.size < need) ws.alloc(need);
  float *hdr = ws.data();
  _Float16 *gfrag = reinterpret_cast<_Float16 *>(ws.data() + 16);
  {
    IMP_PROF("als_cg_w256_prep");
    w256_prep_kernel<<<1, 1024, 0, stream()>>>(A0, gfrag, hdr);
    IMP_CHECK_HIP(hipGetLastError());
  }
  launch_w256_class<16>(C, b[2], b[3] - b[2], X, Y, gfrag, hdr, cg_steps, "als_cg_w256_team16_rows");
  launch_w256_class<8>(C, b[3], b[4] - b[3], X, Y, gfrag, hdr, cg_steps, "als_cg_w256_team8_rows");
  launch_w256_class<4>(C, b[4], b[5] - b[4], X, Y, gfrag, hdr, cg_steps, "als_cg_w256_team4_rows");
  launch_w256_class<2>(C, b[5], b[6] - b[5], X, Y, gfrag, hdr, cg_steps, "als_cg_w256_team2_rows");
  launch_w256_class<1>(C, b[6], b[7] - b[6], X, Y, gfrag, hdr, cg_steps, "als_cg_w256_short_rows");
}

}  // namespace imp
