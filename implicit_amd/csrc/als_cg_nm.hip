// CG for the LONG rows (> 512 nonzeros) of the f = 64 / 128 path through the row's explicit normal matrix, built on the matrix cores.
//
// What it replaces.  The reference's kernel (implicit/gpu/als.cu:23-111) applies A_u = YtY + sum_k (c_k - 1) y_k y_k^T to a vector
// once per CG pass by walking the row's nonzeros: 1 + cg_steps gathers of every y_k.  The resident-tile kernels of the shorter
// classes gather once and keep the tile in registers; a row of thousands of nonzeros does not fit a workgroup's registers, so
// rounds 2-3 spread it over a cluster of workgroups that meet through memory after every pass (als_cg_cluster.hip: latency-bound,
// 0.28 of the HBM roofline) and streamed the rows beyond 4096 nonzeros once per pass (2.7x the algorithmic traffic).
//
// Here a long row's A_u is formed ONCE, explicitly -- a rank-nnz symmetric update, i.e. matrix-core work -- and the CG passes then
// run on that f x f matrix in the LDS (1 + cg_steps dense products of 16 K multiply-adds: nothing beside nnz * f^2).  Every y_k is
// gathered exactly once, by exactly one wavefront.
//
//   * Operand layout without a transpose: lane (a = l % M, g = l / M) gathers FOUR consecutive factors 4a .. 4a+3 of EIGHT
//     nonzeros (8 x global_load_dwordx4, each a fully coalesced 512-byte row at f = 128).  Component c of those loads, over the 8
//     nonzeros, is precisely what v_mfma_f32_32x32x16_f16 wants from lane l as its A or B operand (row a of a 32 x K block, K
//     positions 8g .. 8g+7) if the factor index is READ AS i = 4 m + I: the f factors split into 4 interleaved blocks I = i % 4 of
//     M = f/4 rows m = i / 4.  A_u then consists of 4 x 4 tiles T(I,J)[m][n] = A_u[4m+I][4n+J]; the 10 tiles with I >= J cover
//     every unordered pair.  f = 64: the same with v_mfma_f32_16x16x32_f16 (M = 16, 4 lane groups, 32 nonzeros per step).
//   * Precision: fp32 operands are split into two fp16 halves x = h + l (h = rn16(x), l = rn16(x - h)) and the tile takes three
//     products  u_h y_h + u_l y_h + u_h y_l  (u and y carry the weight w = |c| - 1 between them, see nm_build), fp32
//     accumulation.  |x - (h + l)| <= max(2^-23 |x|, 2^-25): the low half is an fp16 subnormal below |x| = 1/8, so factors of
//     0.01 .. 0.1 carry 18 .. 21 significant bits per operand and a product is good to ~1e-6 of itself, unbiased
//     (tests/test_nm_split_model.py restates this arithmetic in numpy); a row's sum comes out CLOSER to the float64 answer than
//     the fp32 reference's dot-product chain (PARITY.md).  fp16 factor storage: y IS an fp16 number, two products.  Factors
//     beyond +-13 would overflow the operands of a 10^7 confidence (no ALS factor is that large).
//   * Inside the workgroup the TILES are dealt to the four wavefronts (3, 3, 2, 2) and the converted operands travel through the
//     LDS: every nonzero is converted once, every tile's sum lives in one wavefront's accumulators from the first nonzero to
//     the last, and the image of A_u is written by the tiles' owners with plain stores (nm_build).
//   * Rows of more than `segment` nonzeros (imp_csr::plan_nm: 2048 .. 16384, by the amount of long-row work) are cut into
//     segments; their partial images go through a workspace, a second kernel sums them in segment order (one workgroup per
//     sixteenth of an image) and a third runs the CG.  Every other row is finished by the workgroup that built its matrix.
//     Segments are handed out by a ticket counter, longest first.
//   * CG on the LDS image: the reference's recurrences (als.cu:45-109) with the product A_u p evaluated from the explicit matrix;
//     256 threads, 256/f threads per matrix row.
//
// Measured (configs[2], per iteration): long-row class 0.89 -> 0.54 ms, 0.27 -> 0.46 of the HBM roofline; the rounds are bounded
// by the alternation of their two phases (conversion on the VALU, products on the matrix pipe: two workgroups per CU do not stay
// in antiphase), not by memory: knock-outs and counters in DESIGN.md section 4.2.  IMP_NM=0 restores the cluster + streamed
// kernels (A/B, parity).
#include "als_qf_common.h"
#include "common.h"
#include "wave_ops.h"

namespace imp {

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int F> struct NmShape;
template <> struct NmShape<128> {
  static constexpr int M = 32, KG = 2, NACC = 16;
  using acc_t = f32x16;
  __device__ static __forceinline__ acc_t mfma(half8 a, half8 b, acc_t c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
  // C/D layout of the 32 x 32 tile: column = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
  __device__ static __forceinline__ int row(int lane, int e) { return (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5); }
  __device__ static __forceinline__ int col(int lane) { return lane & 31; }
};
template <> struct NmShape<64> {
  static constexpr int M = 16, KG = 4, NACC = 4;
  using acc_t = f32x4v;
  __device__ static __forceinline__ acc_t mfma(half8 a, half8 b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
  // 16 x 16 tile: column = lane & 15, row = 4 (lane >> 4) + e
  __device__ static __forceinline__ int row(int lane, int e) { return 4 * (lane >> 4) + e; }
  __device__ static __forceinline__ int col(int lane) { return lane & 15; }
};

template <int F> struct NmLayout {
  static constexpr int M = F / 4;              // rows of a tile
  static constexpr int TS = M + 1;             // row stride inside a tile: rows m and columns n both walk all LDS banks
  static constexpr int IMG = 16 * M * TS;      // floats of the image: 16 tiles (I, J), tile (I, J)[m][n] = A[4m+I][4n+J]
  static constexpr int KCH = 8 * NmShape<F>::KG;  // nonzeros per step (one wavefront's gathers)
  static constexpr int ROUND = 4 * KCH;        // nonzeros per round: one step from each of the four wavefronts
  static constexpr int kTiles = 10;            // (I, J), I >= J
  static constexpr int kSlots = 3;             // tiles per wavefront: 3, 3, 2, 2
  // operand exchange: [step][kind: uh ul yh yl][block][lane] quads of 16 bytes
  static constexpr int QUAD = 64 * 4;          // floats of one operand quad (one b128 per lane)
  static constexpr int OPS = 4 * 16 * QUAD;    // floats of the exchange buffer (64 KB); the image re-uses the space afterwards
  static constexpr int NP = 256 / F;           // thread groups sharing a matrix row in the CG phase
  static constexpr int kVec = IMG > OPS ? IMG : OPS;  // offset of the vectors behind the image / exchange buffer
  static constexpr size_t lds_floats = (size_t)kVec + 2 * F + NP * F + 64 + 4 * F;  // image / operands | b | p | partial products | reduction slots | b per wavefront
  __host__ __device__ static constexpr int at(int I, int J, int m, int n) { return ((I * 4 + J) * M + m) * TS + n; }
};
__device__ constexpr int nm_tile_i(int t) { return t < 1 ? 0 : t < 3 ? 1 : t < 6 ? 2 : 3; }
__device__ constexpr int nm_tile_j(int t) { return t - (nm_tile_i(t) * (nm_tile_i(t) + 1)) / 2; }

// two fp32 -> (high halves, low halves), both packed
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned &hi, unsigned &lo) {
  const f32x2 v = {x0, x1};
  const half2v h = __builtin_convertvector(v, half2v);
  const f32x2 r = v - __builtin_convertvector(h, f32x2);
  const half2v l = __builtin_convertvector(r, half2v);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ unsigned pack_pair(float x0, float x1) {
  const f32x2 v = {x0, x1};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, half2v));
}
__device__ __forceinline__ float comp(const float4 &v, int c) { return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w; }

// The regularised gramian in image order (once per launch; the workgroups copy it linearly).  Block 0 also resets the ticket and
// picks the launch's OPERAND SCALE 2^k (ctl[1]): the fp16 split keeps 22 bits of an operand only while its low half is a normal
// fp16 number, i.e. while |operand| >= 1/8 -- cold-start factors of 0.005 would sit ten binades below that.  k brings the
// root-mean-square of the largest factor column (from the gramian's diagonal: sqrt(max_j A0[j][j] / rows of Y), an upper bound
// because the diagonal carries the regularisation) to about 8; it never scales down (k >= 0), and a row whose scaled operands
// leave the fp16 range is caught by nm_cg's finiteness check and re-solved in fp32 (the fix-up list).  ctl[2] counts those rows.
template <int F> __global__ void nm_gram_image_kernel(const float *__restrict__ A0, float *__restrict__ img, int *__restrict__ ctl,
                                                      float y_rows, int forced_k, float reg = 0.f) {  // reg: added to the diagonal (Cholesky path: YtY arrives bare)
  using L = NmLayout<F>;
  if (blockIdx.x == 0 && threadIdx.x < 64) {
    float d = 0.f;
    for (int j = threadIdx.x; j < F; j += 64) d = fmaxf(d, A0[(size_t)j * F + j]);
    d = wave_allmax(d);
    if (threadIdx.x == 0) {
      int k = 0;
      const float rms = sqrtf((d + reg) / fmaxf(y_rows, 1.f));
      if (rms > 0.f && rms < 8.f) k = min((int)floorf(log2f(8.f / rms)), 16);  // NaN / inf / zero diagonal: k = 0
      ctl[0] = 0;
      ctl[1] = forced_k >= 0 ? forced_k : k;
      ctl[2] = 0;
      ctl[3] = 0;
    }
  }
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < L::IMG; e += gridDim.x * blockDim.x) {
    const int n = e % L::TS, m = (e / L::TS) % L::M, t = e / (L::TS * L::M);
    img[e] = n < L::M ? A0[(size_t)(4 * m + t / 4) * F + 4 * n + t % 4] + ((m == n && t / 4 == t % 4) ? reg : 0.f) : 0.f;
  }
}

// Workgroup barrier that orders the LDS only.  __syncthreads() is a fence over ALL memory: it drains vmcnt, i.e. waits for the
// gathers of the coming rounds that are in flight on purpose (measured: 12 K cycles per round instead of ~4 K).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- the rank-(end - begin) update of one segment ---------------------------------------------------------------------------
// Rounds of 4 steps.  Wavefront w PRODUCES step w of a round -- gathers its 8 x 4 factors per lane, scales, splits, and leaves the
// sixteen operand quads (uh ul yh yl x 4 blocks) in the LDS -- and, behind a barrier, CONSUMES all four steps for the tiles it
// OWNS (3, 3, 2, 2 of the ten): a tile's sum over the nonzeros lives in one wavefront's accumulators from the first round to the
// last, so there is no reduction across wavefronts at the end (a first version split the nonzeros instead, every wavefront with
// all ten tiles: 509 registers = one wavefront per SIMD at 6 cycles per instruction, and 156 K cycles of ds_add_f32 per row to
// add the four copies up).  Registers: 48 accumulators + 32 landing + the operands in transit; two workgroups per CU.
//
// Weights without a pre-pass (round 5): w = |c| - 1 enters both operands as sqrt|w| -- z = 2^k sqrt|w| y is split ONCE and is the y
// operand as it stands and, with the sign of w flipped into the packed halves, the u operand (steps whose weights are all >= 0,
// i.e. every step of the usual data, do not even write u).  Round 4 dealt w to two DIFFERENT operands (w 2^-e and 2^e, exact
// scalings) and converted every gathered value twice; fp16 factor storage still does, because 2^e y is then an fp16 number and
// needs no low half.  Nothing bounds |y| here: operands beyond the fp16 range turn into infinities, the image and the CG scalars
// stop being finite, and nm_cg hands the row to the fp32 fix-up kernel instead of storing it.
//
// Returns with the image complete in the LDS (gramian added when `whole`), behind a barrier.
template <int F, typename T, bool TRIM = false>  // TRIM: the last round multiplies only the steps that hold nonzeros (short rows)
__device__ __forceinline__ void nm_build(const int32_t *__restrict__ indices, const float *__restrict__ data, const T *__restrict__ Y,
                                         const float *__restrict__ gram_img, bool whole, int begin, int end, float *smem, float *bvec,
                                         int tid, int ko, int scale_k) {
  using S = NmShape<F>;
  using L = NmLayout<F>;
  constexpr bool kHalf = !std::is_same<T, float>::value;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int a = lane % S::M, g = lane / S::M;
  // this wavefront's tiles: slots 0 .. n_slots-1 = tiles first .. first + n_slots - 1
  const int first = wave < 2 ? 3 * wave : 2 * wave + 2;  // 0, 3, 6, 8
  const int n_slots = wave < 2 ? 3 : 2;
  int tI[L::kSlots], tJ[L::kSlots];
#pragma unroll
  for (int k = 0; k < L::kSlots; ++k) {
    const int t = min(first + k, L::kTiles - 1);
    tI[k] = t < 1 ? 0 : t < 3 ? 1 : t < 6 ? 2 : 3;
    tJ[k] = t - tI[k] * (tI[k] + 1) / 2;
  }
  typename S::acc_t acc[L::kSlots];
#pragma unroll
  for (int k = 0; k < L::kSlots; ++k)
#pragma unroll
    for (int e = 0; e < S::NACC; ++e) acc[k][e] = 0.f;
  float b4[4] = {0.f, 0.f, 0.f, 0.f};
  const float scale2 = __uint_as_float((unsigned)(127 + 2 * scale_k) << 23), unscale2 = __uint_as_float((unsigned)(127 - 2 * scale_k) << 23);

  // Every round issues the same loads whether they are needed or not: s_waitcnt counts loads statically, and one conditional
  // gather makes the compiler assume the worst path -- the counts it then emits (vmcnt 7 .. 0) drain the whole queue at every
  // round, prefetch included.
  const int n_rounds = (ko & 1) ? 0 : (end - begin + L::ROUND - 1) / L::ROUND;
  // entry `lane % KCH` of this wavefront's step: column, the two weight factors, c+
  struct Entry {
    int col;
    float c;
  };
  auto load_entry = [&](int r) {
    Entry en;
    const int k = begin + r * L::ROUND + wave * L::KCH + (lane % L::KCH);
    const int kk = min(k, end - 1);
    en.col = indices[kk];
    const float c = data[kk];          // unconditional (see n_rounds)
    en.c = k < end ? c : -1.f;         // beyond the segment: confidence -1 -> weight 0, no share in b
    return en;
  };
  // One round of gathers in flight per wavefront beside the one being worked on (a second landing buffer is 32 registers this
  // kernel does not have: measured 0.90 against 0.95 ms with it, before the consume phase was pipelined)
  float4 yb[8];
  auto gather = [&](const Entry &en, float4 (&y)[8]) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int col = __shfl(en.col, 8 * g + q, 64);
      y[q] = load4(Y + (size_t)col * F + 4 * a);
    }
  };
  float *ops = smem;
  int *negflag = reinterpret_cast<int *>(bvec + 2 * F + L::NP * F + 32);  // [4] (the reduction slots use the first 32 words)
  // entries: e0 = this round, e1 = next (its gather is issued at the end of this round's produce phase), e2 = the one after
  Entry e0{0, -1.f}, e1 = e0, e2 = e0;
  if (n_rounds > 0) {
    e0 = load_entry(0), e1 = load_entry(1);  // beyond the segment: the last entry with weight 0
    gather(e0, yb);
  }
  // ---- consume: the sequence of one wavefront per step, written out per wavefront so that a block's operands are read ONCE for
  // all the tiles that use them (28 quads per step and workgroup instead of 40: the exchange through the LDS is what bounds the
  // round), and the operands of the next tile / the next step are on their way while the three products of the current tile run
  //   wave 0: (0,0) (1,0) (1,1)   wave 1: (2,0) (2,1) (2,2)   wave 2: (3,0) (3,1)   wave 3: (3,2) (3,3)
  struct Pair {
    half8 h, l;
  };
  const float *lane_ops = ops + 4 * lane;
  int negbits = 0;  // bit st: step st of the round carries a negative weight, its u quads are in the buffer (else u = z)
  auto ld_u = [&](int st, int I) {  // uh, ul of block I
    Pair p;
    const int ku = (negbits >> st) & 1 ? 0 : 2;
    p.h = __builtin_bit_cast(half8, *reinterpret_cast<const u32x4 *>(lane_ops + (st * 16 + ku * 4 + I) * L::QUAD));
    p.l = __builtin_bit_cast(half8, *reinterpret_cast<const u32x4 *>(lane_ops + (st * 16 + (ku + 1) * 4 + I) * L::QUAD));
    return p;
  };
  auto ld_y = [&](int st, int J) {  // yh, yl of block J (fp16 factors: 2^e y is an fp16 number too, no low half)
    Pair p;
    p.h = __builtin_bit_cast(half8, *reinterpret_cast<const u32x4 *>(lane_ops + (st * 16 + 2 * 4 + J) * L::QUAD));
    if constexpr (!kHalf) p.l = __builtin_bit_cast(half8, *reinterpret_cast<const u32x4 *>(lane_ops + (st * 16 + 3 * 4 + J) * L::QUAD));
    else p.l = p.h;
    return p;
  };
  auto products = [&](auto k, const Pair &u, const Pair &y) {
    acc[k.value] = S::mfma(u.h, y.h, acc[k.value]);
    acc[k.value] = S::mfma(u.l, y.h, acc[k.value]);
    if constexpr (!kHalf) acc[k.value] = S::mfma(u.h, y.l, acc[k.value]);
  };
  auto consume = [&](int s0, int s1) {  // steps s0 .. s1 - 1 of the exchange buffer
    negbits = __builtin_amdgcn_readfirstlane(negflag[0] | (negflag[1] << 1) | (negflag[2] << 2) | (negflag[3] << 3));
    if (wave == 0) {
      Pair u0 = ld_u(s0, 0), y0 = ld_y(s0, 0);
#pragma unroll 1
      for (int st = s0; st < s1; ++st) {
        const Pair u1 = ld_u(st, 1);
        products(idx_t<0>{}, u0, y0);
        const Pair y1 = ld_y(st, 1);
        products(idx_t<1>{}, u1, y0);
        u0 = ld_u(min(st + 1, s1 - 1), 0), y0 = ld_y(min(st + 1, s1 - 1), 0);  // after the last step: a re-read nobody uses
        products(idx_t<2>{}, u1, y1);
      }
    } else if (wave == 1) {
      Pair u2 = ld_u(s0, 2), y0 = ld_y(s0, 0);
#pragma unroll 1
      for (int st = s0; st < s1; ++st) {
        const Pair y1 = ld_y(st, 1);
        products(idx_t<0>{}, u2, y0);
        const Pair y2 = ld_y(st, 2);
        products(idx_t<1>{}, u2, y1);
        const Pair un = ld_u(min(st + 1, s1 - 1), 2);
        y0 = ld_y(min(st + 1, s1 - 1), 0);
        products(idx_t<2>{}, u2, y2);
        u2 = un;
      }
    } else {
      const int J0 = wave == 2 ? 0 : 2;  // wave 2: (3,0) (3,1); wave 3: (3,2) (3,3)
      Pair u3 = ld_u(s0, 3), ya = ld_y(s0, J0);
#pragma unroll 1
      for (int st = s0; st < s1; ++st) {
        const Pair yb2 = ld_y(st, J0 + 1);
        products(idx_t<0>{}, u3, ya);
        const Pair un = ld_u(min(st + 1, s1 - 1), 3);
        ya = ld_y(min(st + 1, s1 - 1), J0);
        products(idx_t<1>{}, u3, yb2);
        u3 = un;
      }
    }
  };
  auto produce = [&](int r, float4 (&y)[8]) {
    // The entries rotate HERE, not where e2 is loaded: a register move of a value still in flight makes the compiler wait for
    // its load, and -- the queue being in order -- at the end of the produce phase that wait would also cover the gathers just
    // issued.  Here it covers the loads older than e2: the rows of THIS round, which are due anyway.
    if (r > 0) e0 = e1, e1 = e2;
    e2 = load_entry(r + 2);  // ahead of this round's gathers in the (in-order) load queue
    // ---- produce step `wave` of this round
    {
      // this lane's entry.  fp32 factors: w = |c| - 1 (times the launch's operand scale 4^k, nm_gram_image_kernel: the image comes
      // out scaled by 4^k and is scaled back, exactly, where it is written) enters BOTH operands as sqrt|w|: z = sqrt|w| y is
      // split once and serves as y operand and -- with the sign of w, a bit flip of the packed halves -- as u operand: half the
      // conversions of dealing w to two different operands (round 4), and steps without a negative weight (every step of the
      // usual data, c >= 1) exchange half the quads.  v_sqrt_f32 is good to 1 ulp: z z carries w to 2^-23.
      // fp16 factors: w dealt as wa = w 2^-e and sb = 2^e, e = floor((exponent of |w|) / 2) clamped to [-12, 24] -- exact
      // scalings, 2^e y stays an fp16 number and needs no low half.
      const float w_mine = (fabsf(e0.c) - 1.f) * scale2;
      const float cp_mine = e0.c > 0.f ? e0.c : 0.f;
      float *mine = ops + (size_t)wave * 16 * L::QUAD + 4 * lane;
      if constexpr (!kHalf) {
        const float sq_mine = __builtin_amdgcn_sqrtf(fabsf(w_mine));
        const unsigned ng_mine = w_mine < 0.f ? 0x8000u : 0u;
        const bool any_neg = __builtin_amdgcn_ballot_w64(ng_mine != 0u) != 0ull;  // wave-uniform (lanes >= KCH hold copies)
        if (lane == 0) negflag[wave] = any_neg ? 1 : 0;
        float sq[8];
        unsigned ng[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          sq[q] = __shfl(sq_mine, 8 * g + q, 64);
          ng[q] = (unsigned)__shfl((int)ng_mine, 8 * g + q, 64);
          const float cp = __shfl(cp_mine, 8 * g + q, 64);
          b4[0] = fmaf(cp, y[q].x, b4[0]);
          b4[1] = fmaf(cp, y[q].y, b4[1]);
          b4[2] = fmaf(cp, y[q].z, b4[2]);
          b4[3] = fmaf(cp, y[q].w, b4[3]);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          u32x4 zh, zl;
#pragma unroll
          for (int q = 0; q < 8; q += 2) {
            unsigned h, l;
            split_pair(sq[q] * comp(y[q], c), sq[q + 1] * comp(y[q + 1], c), h, l);
            zh[q / 2] = h, zl[q / 2] = l;
          }
          *reinterpret_cast<u32x4 *>(mine + (2 * 4 + c) * L::QUAD) = zh;
          *reinterpret_cast<u32x4 *>(mine + (3 * 4 + c) * L::QUAD) = zl;
          if (any_neg) {  // u = sign(w) z, only where the step has a negative weight (the consumers read z otherwise)
            u32x4 uh, ul;
#pragma unroll
            for (int q = 0; q < 8; q += 2) {
              const unsigned m = ng[q] | (ng[q + 1] << 16);
              uh[q / 2] = zh[q / 2] ^ m, ul[q / 2] = zl[q / 2] ^ m;
            }
            *reinterpret_cast<u32x4 *>(mine + (0 * 4 + c) * L::QUAD) = uh;
            *reinterpret_cast<u32x4 *>(mine + (1 * 4 + c) * L::QUAD) = ul;
          }
          __builtin_amdgcn_sched_barrier(0);  // one block's quads at a time: interleaved blocks spill, and a spill reload is a
        }                                     // vector-memory load -- waiting for it (vmcnt 0) drains the gathers in flight
      } else {
        unsigned hb = (((__float_as_uint(w_mine) & 0x7f800000u) + (127u << 23)) >> 1) & 0x7f800000u;
        hb = min(max(hb, (127u - 12u) << 23), (127u + 24u) << 23);
        const float sb_mine = __uint_as_float(hb), wa_mine = w_mine * __uint_as_float((254u << 23) - hb);
        if (lane == 0) negflag[wave] = 1;  // the u quads are always written
        float wa[8], sb[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          wa[q] = __shfl(wa_mine, 8 * g + q, 64);
          sb[q] = __shfl(sb_mine, 8 * g + q, 64);
          const float cp = __shfl(cp_mine, 8 * g + q, 64);
          b4[0] = fmaf(cp, y[q].x, b4[0]);
          b4[1] = fmaf(cp, y[q].y, b4[1]);
          b4[2] = fmaf(cp, y[q].z, b4[2]);
          b4[3] = fmaf(cp, y[q].w, b4[3]);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          u32x4 uh, ul, yh;
#pragma unroll
          for (int q = 0; q < 8; q += 2) {
            const float y0 = comp(y[q], c), y1 = comp(y[q + 1], c);
            unsigned h, l;
            yh[q / 2] = pack_pair(sb[q] * y0, sb[q + 1] * y1);
            split_pair(wa[q] * y0, wa[q + 1] * y1, h, l);
            uh[q / 2] = h, ul[q / 2] = l;
          }
          *reinterpret_cast<u32x4 *>(mine + (0 * 4 + c) * L::QUAD) = uh;
          *reinterpret_cast<u32x4 *>(mine + (1 * 4 + c) * L::QUAD) = ul;
          *reinterpret_cast<u32x4 *>(mine + (2 * 4 + c) * L::QUAD) = yh;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    gather(e1, y);  // the landing registers are free again (beyond the last round: the segment's last row, unused)
  };
  // (A staggered schedule -- wavefronts 0, 1 convert in the first half of a trip, 2, 3 in the second, everybody multiplying the
  // other pair's steps meanwhile, so that each workgroup keeps both pipes busy by itself -- measured 0.67 against 0.52 ms: the
  // converting pair's conversion + two steps of products is a longer critical path per 32 nonzeros than the two-phase round's per 64.)
#pragma unroll 1
  for (int r = 0; r < n_rounds; ++r) {
    produce(r, yb);
    lds_barrier();
    if constexpr (TRIM) consume(0, min(4, (end - begin - r * L::ROUND + L::KCH - 1) / L::KCH));  // a row of 20 nonzeros has one step
    else consume(0, 4);  // (raising the wave priority here changes nothing: 0.556 against 0.560 ms)
    lds_barrier();  // the exchange buffer is free again (and, after the last round, free for the image)
  }
  // b: lane (a, g) holds its nonzeros' share of factors 4a .. 4a+3 = positions c M + a
  if (!(ko & 2)) {
    // (summed in a FIXED order: across the lane groups here, across the wavefronts by the caller -- ds_add_f32 would add in
    // whatever order the wavefronts arrive, and the same sweep would differ in the last bit from run to run)
    float *bstage = bvec + 2 * F + L::NP * F + 64;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float v = b4[c];
      if constexpr (S::KG == 4) v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (g == 0) bstage[wave * F + c * L::M + a] = v;
    }
    // the image: every tile (and its mirror) is written by its one owner, gramian added on the way
    float *img = smem;
    const int n = S::col(lane);
#pragma unroll
    for (int k = 0; k < L::kSlots; ++k) {
      if (k < n_slots) {
        const int I = tI[k], J = tJ[k];
        float gv[S::NACC];  // the tile's share of the gramian: all loads in flight before the first store
#pragma unroll
        for (int e = 0; e < S::NACC; ++e) gv[e] = 0.f;
        if (whole) {
#pragma unroll
          for (int e = 0; e < S::NACC; ++e) gv[e] = gram_img[L::at(I, J, S::row(lane, e), n)];
        }
#pragma unroll
        for (int e = 0; e < S::NACC; ++e) {
          const int m = S::row(lane, e);
          const float v = fmaf(acc[k][e], unscale2, gv[e]);
          img[L::at(I, J, m, n)] = v;
          if (I != J) img[L::at(J, I, n, m)] = v;
        }
      }
    }
  }
  __syncthreads();
}

// ---- CG on the LDS image (als.cu:45-109 with A_u p from the explicit matrix) ----------------------------------------------------
// Vectors live in image order: position r = I M + m is factor 4m + I.  Thread (r = tid % F, grp = tid / F) forms the products of
// row r with the column blocks J of its group: lanes walk m, i.e. consecutive tile rows of stride M + 1 -- conflict-free.
template <int F, typename T>
__device__ __forceinline__ bool nm_cg(const float *img, const float *bvec, float *pv, float *parts, float *red, T *xrow, int cg_steps, int tid) {
  using L = NmLayout<F>;
  constexpr int M = L::M, NP = L::NP, JPG = 4 / NP;
  const int r = tid % F, grp = tid / F, lane = tid & 63, wave = tid >> 6;
  const int I = r / M, m = r % M;
  int slot = 0;
  auto matvec = [&]() {
    float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jj = 0; jj < JPG; ++jj) {
      const int J = grp * JPG + jj;
      const float *row = img + L::at(I, J, m, 0);
#pragma unroll
      for (int n = 0; n < M; n += 4) {
        const float4 p4 = *reinterpret_cast<const float4 *>(pv + J * M + n);
        s[0] = fmaf(row[n], p4.x, s[0]), s[1] = fmaf(row[n + 1], p4.y, s[1]);
        s[2] = fmaf(row[n + 2], p4.z, s[2]), s[3] = fmaf(row[n + 3], p4.w, s[3]);
      }
    }
    parts[grp * F + r] = (s[0] + s[1]) + (s[2] + s[3]);
    __syncthreads();
    float q = parts[r];
#pragma unroll
    for (int gq = 1; gq < NP; ++gq) q += parts[gq * F + r];
    return q;
  };
  auto block_sum = [&](float v) {  // over the F positions (the threads of group 0 contribute)
    v = wave_allsum(grp == 0 ? v : 0.f);
    if (lane == 0) red[4 * slot + wave] = v;
    __syncthreads();
    const float s = (red[4 * slot] + red[4 * slot + 1]) + (red[4 * slot + 2] + red[4 * slot + 3]);
    slot = (slot + 1) & 7;
    return s;
  };
  // Operands beyond the fp16 range leave infinities in the image; whatever they touch stops being finite.  The squared norms
  // below see every component of every residual and of every A p (through p.Ap) -- one non-finite value in them and the row is
  // NOT stored: the caller lists it for the fp32 fix-up kernel.  (A row whose inputs are not finite to begin with takes the same
  // way and gets the reference's own non-finite answer there.)
  auto finite = [](float v) { return fabsf(v) <= 3.0e38f; };
  float x = load1(xrow + 4 * m + I);
  if (grp == 0) pv[r] = x;
  __syncthreads();
  float res = bvec[r] - matvec();
  float p = res;
  float rsold = block_sum(res * res);
  if (!finite(rsold)) return true;
  if (rsold < 1e-20f) return false;
  for (int it = 0; it < cg_steps; ++it) {
    if (grp == 0) pv[r] = p;
    __syncthreads();
    const float Ap = matvec();
    const float pAp = block_sum(p * Ap);
    const float alpha = rsold / pAp;
    x = fmaf(alpha, p, x);
    res = fmaf(-alpha, Ap, res);
    const float rsnew = block_sum(res * res);
    if (!finite(pAp) || !finite(rsnew) || !finite(alpha)) return true;
    if (rsnew < 1e-20f) break;
    p = fmaf(rsnew / rsold, p, res);
    rsold = rsnew;
  }
  if (grp == 0) store1(xrow + 4 * m + I, x);
  return false;
}

#ifdef CHOL_NM_STATS  // timing-only build: shader-clock cycles per phase of als_chol_nm_rows_kernel, wave 0 lane 0 of every workgroup
__device__ unsigned long long g_chol_nm_stats[8];  // [0] build [1] block load [2] factorisation [3] L store [4] back substitution [5] rows
#define CHOL_TICK(slot)                                                          \
  do {                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                           \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();                \
    if (threadIdx.x == 0) atomicAdd(&g_chol_nm_stats[slot], now_ - chol_t_last); \
    chol_t_last = now_;                                                          \
    __builtin_amdgcn_sched_barrier(0);                                           \
  } while (0)
__device__ unsigned long long chol_t_last_dummy;
#else
#define CHOL_TICK(slot) ((void)0)
#endif
struct NoGram {  // nm_chol's gramian argument when the image already carries the gramian
  __device__ __forceinline__ float operator()(int, int, int, int) const { return 0.f; }
};
// ---- Cholesky on the LDS image (round 5): x = A_u^-1 b, what the CPU reference's posv does per row (_als.pyx:75-142) ---------------
// f = 128, 256 threads.  The image's 4 x 4 interleaving makes position (m, n) of its sixteen tiles the 4 x 4 block
// A[4m .. 4m+3][4n .. 4n+3]: the 528 blocks of the lower triangle are dealt to the threads (at most three each) and stay in
// REGISTERS for the whole factorisation -- right-looking, one block column per turn:
//   * everybody reads the (updated) diagonal block its owner published and factors it redundantly (4 x 4 Cholesky, v_rsq + one
//     Newton step): no serial phase, no extra barrier;
//   * the owners of the blocks below it solve them against the diagonal factor (they are L now) and publish the panel;
//   * barrier; every block to the right takes its rank-4 update from the panel (two 4 x 4 operands from the LDS, 64 FMAs), the
//     owner of the next diagonal block publishes it; barrier.
// b rides along as a 129th row (thread t holds b[4t .. 4t+3]): z = L^-1 b is complete when the factorisation is.  L then goes to
// the LDS row-major (the image's space: F (F + 4) floats exactly) and x = L^-T z is a blocked back substitution, one barrier per
// block.  Returns true -- and stores nothing -- when a pivot is not positive (or not finite): the caller lists the row for the
// workgroup-per-row fp32 kernel, which then decides whether it is a failure (_als.pyx:136-138).
template <int F, typename GB>
__device__ __forceinline__ bool nm_chol(float *img, float *bvec, float *scr, float *xrow, int tid, const GB &gblocks) {
  static_assert(F == 128, "block ownership and the LDS budget are laid out for f = 128");
#ifdef CHOL_NM_STATS
  unsigned long long chol_t_last = __builtin_amdgcn_s_memtime();
#endif
  using L = NmLayout<F>;
  constexpr int NB = F / 4, NBLK = NB * (NB + 1) / 2, LDL = F + 4, SL = (NBLK + 255) / 256;
  static_assert(F * LDL <= L::kVec, "the row-major factor re-uses the image's space");
  float *pan = scr;                // [2][F][4]  panel: L[i][4 kb .. 4 kb + 3] of the current block column
  float *dblk = scr + 2 * F * 4;   // [2][16]    the diagonal block of the coming turn, as updated so far
  int bm[SL], bn[SL];
  float B[SL][4][4];
#pragma unroll
  for (int s = 0; s < SL; ++s) {
    const int bid = tid + 256 * s;
    bm[s] = bn[s] = -1;
    if (bid < NBLK) {
      // Blocks are numbered by block COLUMN from the right (column NB-1 first), top to bottom inside a column: the blocks still to
      // be updated at turn kb (columns > kb) are then a PREFIX of the numbering, i.e. the first slot of the first threads -- a
      // wavefront runs ceil(active / 256) passes of the update per turn instead of one per slot it holds a live block in (row by
      // row, most turns cost every wavefront two passes).
      int t = (int)((sqrtf(8.f * (float)bid + 1.f) - 1.f) * 0.5f);
      while (t * (t + 1) / 2 > bid) --t;
      while ((t + 1) * (t + 2) / 2 <= bid) ++t;
      bn[s] = NB - 1 - t, bm[s] = bn[s] + (bid - t * (t + 1) / 2);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) B[s][a][c] = img[L::at(a, c, bm[s], bn[s])] + gblocks(a, c, bm[s], bn[s]);
    }
  }
  float rb[4] = {0.f, 0.f, 0.f, 0.f};  // b[4 tid ..] (the image's vectors are stored position I M + m = factor 4 m + I)
  if (tid < NB) {
#pragma unroll
    for (int c = 0; c < 4; ++c) rb[c] = bvec[c * L::M + tid];
  }
  __syncthreads();  // the image and b have been read: their space is free
  CHOL_TICK(1);
#pragma unroll
  for (int s = 0; s < SL; ++s) {
    if (bm[s] == 0 && bn[s] == 0) {  // the first diagonal block's owner
#pragma unroll
      for (int a = 0; a < 4; ++a) *reinterpret_cast<float4 *>(dblk + 4 * a) = make_float4(B[s][a][0], B[s][a][1], B[s][a][2], B[s][a][3]);
    }
  }
  __syncthreads();
  bool fail = false;
#pragma unroll 1
  for (int kb = 0; kb < NB; ++kb) {
    const float *D = dblk + (kb & 1) * 16;
    float *P = pan + (size_t)(kb & 1) * F * 4;
    const float4 d0 = *reinterpret_cast<const float4 *>(D), d1 = *reinterpret_cast<const float4 *>(D + 4),
                 d2 = *reinterpret_cast<const float4 *>(D + 8), d3 = *reinterpret_cast<const float4 *>(D + 12);
    auto rsq = [](float d) {  // 1 / sqrt(d): v_rsq_f32 + one Newton step
      const float r = __builtin_amdgcn_rsqf(d);
      return r * fmaf(-0.5f * d * r, r, 1.5f);
    };
    // 4 x 4 Cholesky of the diagonal block, by every thread alike
    const float p0 = d0.x, r0 = rsq(p0);
    const float l10 = d1.x * r0, l20 = d2.x * r0, l30 = d3.x * r0;
    const float p1 = fmaf(-l10, l10, d1.y), r1 = rsq(p1);
    const float l21 = fmaf(-l20, l10, d2.y) * r1, l31 = fmaf(-l30, l10, d3.y) * r1;
    const float p2 = fmaf(-l21, l21, fmaf(-l20, l20, d2.z)), r2 = rsq(p2);
    const float l32 = fmaf(-l31, l21, fmaf(-l30, l20, d3.z)) * r2;
    const float p3 = fmaf(-l32, l32, fmaf(-l31, l31, fmaf(-l30, l30, d3.w))), r3 = rsq(p3);
    if (!(p0 > 0.f && p1 > 0.f && p2 > 0.f && p3 > 0.f && p3 < 3.0e38f)) {  // uniform: every thread holds the same values
      fail = true;
      break;
    }
    auto solve_row = [&](float (&v)[4]) {  // v <- v L_D^-T
      const float x0 = v[0] * r0;
      const float x1 = fmaf(-x0, l10, v[1]) * r1;
      const float x2 = fmaf(-x1, l21, fmaf(-x0, l20, v[2])) * r2;
      const float x3 = fmaf(-x2, l32, fmaf(-x1, l31, fmaf(-x0, l30, v[3]))) * r3;
      v[0] = x0, v[1] = x1, v[2] = x2, v[3] = x3;
    };
#pragma unroll
    for (int s = 0; s < SL; ++s) {
      if (bn[s] == kb) {
        if (bm[s] > kb) {
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            solve_row(B[s][a]);
            *reinterpret_cast<float4 *>(P + (size_t)(4 * bm[s] + a) * 4) = make_float4(B[s][a][0], B[s][a][1], B[s][a][2], B[s][a][3]);
          }
        } else {  // the diagonal block itself: its factor, with the RECIPROCALS of the pivots' roots on the diagonal (what the
                  // back substitution multiplies by)
          B[s][0][0] = r0, B[s][1][0] = l10, B[s][1][1] = r1, B[s][2][0] = l20, B[s][2][1] = l21, B[s][2][2] = r2;
          B[s][3][0] = l30, B[s][3][1] = l31, B[s][3][2] = l32, B[s][3][3] = r3;
        }
      }
    }
    if (tid == kb) {  // z[4 kb ..]: the right-hand side's block of this column
      solve_row(rb);
      *reinterpret_cast<float4 *>(bvec + 4 * kb) = make_float4(rb[0], rb[1], rb[2], rb[3]);
    }
    __syncthreads();
    float *Dn = dblk + ((kb + 1) & 1) * 16;
#pragma unroll
    for (int s = 0; s < SL; ++s) {
      if (bn[s] > kb) {  // (bm >= bn > kb)
        // (a transposed second copy of the panel and 32 v_pk_fma_f32 per block instead of 64 FMAs was measured slower: 117 K
        // against 99 K cycles per row for the factorisation)
        float4 pm[4], pn[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          pm[a] = *reinterpret_cast<const float4 *>(P + (size_t)(4 * bm[s] + a) * 4);
          pn[a] = *reinterpret_cast<const float4 *>(P + (size_t)(4 * bn[s] + a) * 4);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int c = 0; c < 4; ++c)
            B[s][a][c] = fmaf(-pm[a].w, pn[c].w, fmaf(-pm[a].z, pn[c].z, fmaf(-pm[a].y, pn[c].y, fmaf(-pm[a].x, pn[c].x, B[s][a][c]))));
        if (bn[s] == kb + 1 && bm[s] == kb + 1) {
#pragma unroll
          for (int a = 0; a < 4; ++a) *reinterpret_cast<float4 *>(Dn + 4 * a) = make_float4(B[s][a][0], B[s][a][1], B[s][a][2], B[s][a][3]);
        }
      }
    }
    if (tid < NB && tid > kb) {
      const float4 z4 = *reinterpret_cast<const float4 *>(bvec + 4 * kb);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4 pc = *reinterpret_cast<const float4 *>(P + (size_t)(4 * tid + c) * 4);
        rb[c] = fmaf(-z4.w, pc.w, fmaf(-z4.z, pc.z, fmaf(-z4.y, pc.y, fmaf(-z4.x, pc.x, rb[c]))));
      }
    }
    __syncthreads();
  }
  CHOL_TICK(2);
  if (fail) return true;
  // L, row-major, into the image's space (the lower triangle; nothing above the diagonal is read)
#pragma unroll
  for (int s = 0; s < SL; ++s) {
    if (bm[s] >= 0) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
        *reinterpret_cast<float4 *>(img + (size_t)(4 * bm[s] + a) * LDL + 4 * bn[s]) = make_float4(B[s][a][0], B[s][a][1], B[s][a][2], B[s][a][3]);
    }
  }
  __syncthreads();
  CHOL_TICK(3);
  // x = L^-T z, four unknowns per turn from the bottom: every thread solves the 4 x 4 block alike, thread i < 4 kb takes the
  // solved unknowns out of z[i]
  float *xs = pan;
#pragma unroll 1
  for (int kb = NB - 1; kb >= 0; --kb) {
    const float *row = img + (size_t)(4 * kb) * LDL + 4 * kb;
    const float4 q0 = *reinterpret_cast<const float4 *>(row), q1 = *reinterpret_cast<const float4 *>(row + LDL),
                 q2 = *reinterpret_cast<const float4 *>(row + 2 * LDL), q3 = *reinterpret_cast<const float4 *>(row + 3 * LDL);
    const float4 z = *reinterpret_cast<const float4 *>(bvec + 4 * kb);
    const float x3 = z.w * q3.w;  // (the diagonal holds 1 / L[i][i])
    const float x2 = fmaf(-q3.z, x3, z.z) * q2.z;
    const float x1 = fmaf(-q3.y, x3, fmaf(-q2.y, x2, z.y)) * q1.y;
    const float x0 = fmaf(-q3.x, x3, fmaf(-q2.x, x2, fmaf(-q1.x, x1, z.x))) * q0.x;
    if (tid < 4 * kb) {  // (elements below 4 kb: nobody reads them in this turn)
      const float *col = img + (size_t)(4 * kb) * LDL + tid;
      bvec[tid] = fmaf(-col[3 * LDL], x3, fmaf(-col[2 * LDL], x2, fmaf(-col[LDL], x1, fmaf(-col[0], x0, bvec[tid]))));
    }
    if (tid == 0) *reinterpret_cast<float4 *>(xs + 4 * kb) = make_float4(x0, x1, x2, x3);
    __syncthreads();
  }
  CHOL_TICK(4);
  if (tid < F) xrow[tid] = xs[tid];
  return false;
}

// One workgroup per segment of plan_nm at a time; a row that is ONE segment is solved here, the others leave partial[seg] =
// (image part, b part).  Segments are handed out through a ticket counter in plan order (longest first): with fixed shares the
// average wavefront was alive for 66 % of the launch (segments of up to `nm_segment` nonzeros = up to 100 us each in a 320 us launch).
template <int F, typename T, bool CHOL = false>
__global__ __launch_bounds__(256, 2) void als_cg_nm_kernel(const LongPlanDev plan, const int32_t *__restrict__ indices,
                                                           const float *__restrict__ data, T *__restrict__ X, const T *__restrict__ Y,
                                                           const float *__restrict__ gram_img, int cg_steps, float *__restrict__ partial,
                                                           int *__restrict__ ticket,  // [0] ticket [1] operand scale k [2] rows left to the fix-up
                                                           unsigned *__restrict__ fix_rows,
                                                           int ko) {  // ko: timing-only knock-outs (IMP_NM_KO), 0 in production
  using L = NmLayout<F>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ int next_item;
  float *img = smem, *bvec = img + L::kVec, *pv = bvec + F, *parts = pv + F, *red = parts + L::NP * F;
  const int tid = threadIdx.x;
  const int scale_k = __builtin_amdgcn_readfirstlane(ticket[1]);
  for (int s = blockIdx.x; s < plan.n_seg;) {
    const int li = plan.seg_row[s];
    const int begin = plan.seg_begin[s], end = plan.seg_end[s];
    const bool whole = plan.row_seg[li + 1] - plan.row_seg[li] == 1;
    __syncthreads();  // the previous item's CG has read the image and b
    nm_build<F, T>(indices, data, Y, gram_img, whole, begin, end, smem, bvec, tid, ko, scale_k);
    if (tid < F) {
      const float *bstage = bvec + 2 * F + L::NP * F + 64;
      bvec[tid] = (bstage[tid] + bstage[F + tid]) + (bstage[2 * F + tid] + bstage[3 * F + tid]);
    }
    __syncthreads();
    // the next ticket is drawn here: its round trip hides behind the CG / the store of the partial image
    int drawn = 0;
    if (tid == 0) drawn = (int)gridDim.x + atomicAdd(ticket, 1);
    if (!(ko & 4)) {
      if (whole) {
        bool bad = false;
        if constexpr (CHOL) bad = nm_chol<F>(img, bvec, pv, X + (size_t)plan.rows[li] * F, tid, NoGram{});  // (a separate instantiation: inlined
        else bad = nm_cg<F, T>(img, bvec, pv, parts, red, X + (size_t)plan.rows[li] * F, cg_steps, tid);  // beside the CG it cost the CG kernel 9 spills)
        if (bad && tid == 0) fix_rows[atomicAdd(ticket + 2, 1)] = (unsigned)plan.rows[li];
      } else {
        float *out = partial + (size_t)s * (L::IMG + F);
        for (int e = tid; e < L::IMG / 4; e += 256) reinterpret_cast<float4 *>(out)[e] = reinterpret_cast<const float4 *>(img)[e];
        if (tid < F) out[L::IMG + tid] = bvec[tid];
      }
    }
    if (tid == 0) next_item = drawn;
    __syncthreads();
    s = next_item;
  }
}

// Rows of more than one segment (the first n_multi rows of the plan), step 1: partial[first segment] = gramian + the sum of the
// row's partial images in segment order.  One workgroup per (row, sixteenth of the image); the loads of eight segments are in
// flight per thread (a first version had one workgroup walk a row's partials, four loads in flight: 143 us per launch, a third
// of the long rows' time).
template <int F>
__global__ __launch_bounds__(256) void als_cg_nm_reduce_kernel(const LongPlanDev plan, int n_multi, const float *__restrict__ gram_img,
                                                               float *__restrict__ partial) {
  using L = NmLayout<F>;
  constexpr int CH = 16, PER = L::IMG / CH / 4;  // float4 pieces per chunk
  static_assert(L::IMG % (CH * 4) == 0, "chunks of whole float4 pieces");
  for (int item = blockIdx.x; item < n_multi * CH; item += gridDim.x) {
    const int li = item / CH, chunk = item % CH;
    const int s0 = plan.row_seg[li], s1 = plan.row_seg[li + 1];
    const size_t stride = (size_t)(L::IMG + F) / 4;  // float4 pieces per partial (IMG + F is a multiple of 4)
    for (int e = threadIdx.x; e < PER + (chunk == 0 ? F / 4 : 0); e += 256) {
      // chunk 0 also sums the b parts (they sit behind the image)
      const size_t off = e < PER ? (size_t)chunk * PER + e : (size_t)L::IMG / 4 + (e - PER);
      float4 *base = reinterpret_cast<float4 *>(partial) + off;
      float4 sum = e < PER ? reinterpret_cast<const float4 *>(gram_img)[off] : make_float4(0.f, 0.f, 0.f, 0.f);
      int s = s0;
      for (; s + 8 <= s1; s += 8) {
        float4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = base[(size_t)(s + q) * stride];
#pragma unroll
        for (int q = 0; q < 8; ++q) sum.x += v[q].x, sum.y += v[q].y, sum.z += v[q].z, sum.w += v[q].w;
      }
      for (; s < s1; ++s) {
        const float4 v = base[(size_t)s * stride];
        sum.x += v.x, sum.y += v.y, sum.z += v.z, sum.w += v.w;
      }
      base[(size_t)s0 * stride] = sum;
    }
  }
}

// step 2: the CG of those rows on the summed image
template <int F, typename T, bool CHOL = false>
__global__ __launch_bounds__(256) void als_cg_nm_finish_kernel(const LongPlanDev plan, int n_multi, T *__restrict__ X, int cg_steps,
                                                               const float *__restrict__ partial, int *__restrict__ ctl,
                                                               unsigned *__restrict__ fix_rows) {
  using L = NmLayout<F>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *img = smem, *bvec = img + L::kVec, *pv = bvec + F, *parts = pv + F, *red = parts + L::NP * F;
  const int tid = threadIdx.x;
  for (int li = blockIdx.x; li < n_multi; li += gridDim.x) {
    const float *in = partial + (size_t)plan.row_seg[li] * (L::IMG + F);
    __syncthreads();
    for (int e = tid; e < L::IMG / 4; e += 256) reinterpret_cast<float4 *>(img)[e] = reinterpret_cast<const float4 *>(in)[e];
    if (tid < F) bvec[tid] = in[L::IMG + tid];
    __syncthreads();
    bool bad = false;
    if constexpr (CHOL) bad = nm_chol<F>(img, bvec, pv, X + (size_t)plan.rows[li] * F, tid, NoGram{});
    else bad = nm_cg<F, T>(img, bvec, pv, parts, red, X + (size_t)plan.rows[li] * F, cg_steps, tid);
    if (bad && tid == 0) fix_rows[atomicAdd(ctl + 2, 1)] = (unsigned)plan.rows[li];
  }
}

// Cholesky, rows of the classes below the long one (1 .. 512 nonzeros): one workgroup per row at a time, rows handed out by a
// ticket (ctl[3]) in schedule order -- the row's normal matrix on the matrix cores (nm_build, the gramian with reg on its diagonal
// added on the way), the factorisation on the LDS image.
template <int F>
__global__ __launch_bounds__(256, 2) void als_chol_nm_rows_kernel(const int32_t *__restrict__ order, int first, int count,
                                                                  const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                                  const float *__restrict__ data, float *__restrict__ X,
                                                                  const float *__restrict__ Y, const float *__restrict__ gram_img,
                                                                  int *__restrict__ ctl, unsigned *__restrict__ fix_rows) {
  using L = NmLayout<F>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ int next_item;
  float *img = smem, *bvec = img + L::kVec, *pv = bvec + F;
  const int tid = threadIdx.x;
  const int scale_k = __builtin_amdgcn_readfirstlane(ctl[1]);
  // The image is built WITHOUT the gramian (nm_build's own addition reads it tile slot by tile slot: three dependent L2 round trips
  // per row); nm_chol adds this thread's blocks of it where it takes its blocks out of the image -- 48 loads in flight at once.
  // (Keeping the blocks in registers for the whole launch does not fit beside nm_build's 250 registers: 132 spilled.)
  struct GramBlocks {
    const float *img;
    __device__ __forceinline__ float operator()(int a, int c, int m, int n) const { return img[L::at(a, c, m, n)]; }
  } gb{gram_img};
  for (int i = blockIdx.x; i < count;) {
    const int u = order[first + i];
    const int begin = indptr[u], end = indptr[u + 1];
    __syncthreads();  // the previous row's solve has read the factor and z
#ifdef CHOL_NM_STATS
    unsigned long long chol_t_last = __builtin_amdgcn_s_memtime();
#endif
    nm_build<F, float, true>(indices, data, Y, gram_img, false, begin, end, smem, bvec, tid, 0, scale_k);
    if (tid < F) {
      const float *bstage = bvec + 2 * F + L::NP * F + 64;
      bvec[tid] = (bstage[tid] + bstage[F + tid]) + (bstage[2 * F + tid] + bstage[3 * F + tid]);
    }
    __syncthreads();
    CHOL_TICK(0);
#ifdef CHOL_NM_STATS
    if (tid == 0) atomicAdd(&g_chol_nm_stats[5], 1ull);
#endif
    int drawn = 0;
    if (tid == 0) drawn = (int)gridDim.x + atomicAdd(ctl + 3, 1);
    const bool bad = nm_chol<F>(img, bvec, pv, X + (size_t)u * F, tid, gb);
    if (bad && tid == 0) fix_rows[atomicAdd(ctl + 2, 1)] = (unsigned)u;
    if (tid == 0) next_item = drawn;
    __syncthreads();
    i = next_item;
  }
}

template <int F, typename T> void launch_nm(const imp_csr *C, T *X, const T *Y, size_t y_rows, const float *A0, int cg_steps) {
  const LongPlan &lp = C->plan_nm;
  if (lp.n_seg <= 0) return;
  using L = NmLayout<F>;
  static_assert(L::IMG % 4 == 0, "the image is copied in 16-byte pieces");
  const size_t lds = L::lds_floats * sizeof(float);
  const int n_multi = C->nm_multi_rows, n_multi_seg = C->nm_multi_segs;
  auto &ws = ctx().long_ws;
  const size_t need = (size_t)L::IMG + (size_t)n_multi_seg * (L::IMG + F);  // gramian image | partial images
  if (ws.size < need) ws.alloc(need);
  float *gram_img = ws.data(), *partial = gram_img + L::IMG;
  auto &tk = ctx().nm_ticket;  // [0] ticket, [1] operand scale, [2] number of rows left to the fix-up kernel
  if (tk.size < 4) tk.alloc(4, true);
  auto &fix = ctx().nm_fix_rows;
  if (fix.size < (size_t)lp.n_long) fix.alloc((size_t)lp.n_long);
  LongPlanDev plan = lp.dev(C->order.data());
  auto kern = als_cg_nm_kernel<F, T>;
  auto fin = als_cg_nm_finish_kernel<F, T>;
  IMP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  IMP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fin), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  {
    IMP_PROF("als_cg_nm_rows");
    constexpr int forced_k = -1;  // (>= 0: a fixed operand scale 2^k instead of the one taken from the gramian's diagonal)
    nm_gram_image_kernel<F><<<(L::IMG + 255) / 256, 256, 0, stream()>>>(A0, gram_img, tk.data(), (float)y_rows, std::min(forced_k, 16));
    const int grid = std::min(lp.n_seg, ctx().num_cus * 2);  // two resident workgroups per CU; the ticket counter balances them
    constexpr int ko = 0;  // (timing-only knock-outs of the kernel's phases: 1 rounds, 2 image, 4 CG)
    kern<<<grid, 256, lds, stream()>>>(plan, C->indices.data(), C->data.data(), X, Y, gram_img, cg_steps, partial, tk.data(), fix.data(), ko);
    IMP_CHECK_HIP(hipGetLastError());
  }
  if (n_multi > 0) {
    IMP_PROF("als_cg_nm_finish");
    als_cg_nm_reduce_kernel<F><<<std::min(n_multi * 16, ctx().num_cus * 16), 256, 0, stream()>>>(plan, n_multi, gram_img, partial);
    fin<<<std::min(n_multi, ctx().num_cus * 2), 256, lds, stream()>>>(plan, n_multi, X, cg_steps, partial, tk.data(), fix.data());
    IMP_CHECK_HIP(hipGetLastError());
  }
  // rows whose operands left the fp16 range (normally none: the kernel reads a zero and exits)
  launch_cg_fixup<F, T>(reinterpret_cast<const unsigned *>(tk.data() + 2), fix.data(), lp.n_long, C, X, Y, A0, cg_steps);
}

}  // namespace

bool nm_enabled() {
  static const bool on = !(getenv("IMP_NM") && atoi(getenv("IMP_NM")) == 0);
  return on;
}

// Cholesky half sweep at f = 128 (round 5): every non-empty row's normal matrix on the matrix cores, factorised on its LDS image
// (nm_chol).  Queues: the gramian image (reg on the diagonal), the long rows through the segment plan of the CG path (partial
// images, their sum, the finishing kernel in Cholesky mode), every other row through als_chol_nm_rows_kernel.  Rows whose pivots
// are not positive and finite are NOT stored: they are listed (count / rows on the device) for the caller's fp32 kernel.
CholNmList least_squares_cholesky_nm(const imp_csr *C, float *X, const float *Y, size_t y_rows, const float *YtY, float reg) {
  constexpr int F = 128;
  using L = NmLayout<F>;
  const LongPlan &lp = C->plan_nm;
  const int32_t *b = C->bin_start;
  const size_t lds = (L::lds_floats + 128) * sizeof(float);  // + the factorisation's panel and diagonal-block buffers
  const int n_multi = C->nm_multi_rows, n_multi_seg = C->nm_multi_segs;
  auto &ws = ctx().long_ws;
  const size_t need = (size_t)L::IMG + (size_t)n_multi_seg * (L::IMG + F);
  if (ws.size < need) ws.alloc(need);
  float *gram_img = ws.data(), *partial = gram_img + L::IMG;
  auto &tk = ctx().nm_ticket;  // [0] segment ticket, [1] operand scale, [2] rows left to the caller, [3] row ticket
  if (tk.size < 4) tk.alloc(4, true);
  auto &fix = ctx().nm_fix_rows;
  const int capacity = C->nonempty();
  if (fix.size < (size_t)std::max(capacity, 1)) fix.alloc((size_t)std::max(capacity, 1));
  constexpr int forced_k = -1;
  {
    IMP_PROF("als_cholesky_nm_long");
    nm_gram_image_kernel<F><<<(L::IMG + 255) / 256, 256, 0, stream()>>>(YtY, gram_img, tk.data(), (float)y_rows, std::min(forced_k, 16), reg);
    if (lp.n_seg > 0) {
      LongPlanDev plan = lp.dev(C->order.data());
      auto kern = als_cg_nm_kernel<F, float, true>;
      auto fin = als_cg_nm_finish_kernel<F, float, true>;
      IMP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      IMP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fin), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      kern<<<std::min(lp.n_seg, ctx().num_cus * 2), 256, lds, stream()>>>(plan, C->indices.data(), C->data.data(), X, Y, gram_img, -1, partial,
                                                                          tk.data(), fix.data(), 0);
      if (n_multi > 0) {
        als_cg_nm_reduce_kernel<F><<<std::min(n_multi * 16, ctx().num_cus * 16), 256, 0, stream()>>>(plan, n_multi, gram_img, partial);
        fin<<<std::min(n_multi, ctx().num_cus * 2), 256, lds, stream()>>>(plan, n_multi, X, -1, partial, tk.data(), fix.data());
      }
    }
    IMP_CHECK_HIP(hipGetLastError());
  }
  const int count = b[7] - b[1];
  if (count > 0) {
    IMP_PROF("als_cholesky_nm_rows");
    auto rk = als_chol_nm_rows_kernel<F>;
    IMP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(rk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    rk<<<std::min(count, ctx().num_cus * 2), 256, lds, stream()>>>(C->order.data(), b[1], count, C->indptr.data(), C->indices.data(),
                                                                    C->data.data(), X, Y, gram_img, tk.data(), fix.data());
    IMP_CHECK_HIP(hipGetLastError());
#ifdef CHOL_NM_STATS
    {
      unsigned long long h[8], z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      IMP_CHECK_HIP(hipStreamSynchronize(stream()));
      IMP_CHECK_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_chol_nm_stats), sizeof(h)));
      IMP_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_chol_nm_stats), z, sizeof(z)));
      const double n = (double)std::max<unsigned long long>(1, h[5]);
      fprintf(stderr, "[chol-nm-stats] rows=%llu cycles per row: build %.0f  block load %.0f  factorisation %.0f  L store %.0f  back substitution %.0f\n",
              h[5], h[0] / n, h[1] / n, h[2] / n, h[3] / n, h[4] / n);
    }
#endif
  }
  return CholNmList{reinterpret_cast<const unsigned *>(tk.data() + 2), fix.data(), capacity};
}

template <typename T> void least_squares_cg_nm(const imp_csr *C, T *X, const T *Y, size_t y_rows, const float *A0, int f, int cg_steps) {
  if (f == 128) launch_nm<128, T>(C, X, Y, y_rows, A0, cg_steps);
  else if (f == 64) launch_nm<64, T>(C, X, Y, y_rows, A0, cg_steps);
  else throw std::invalid_argument("least_squares_cg_nm: f must be 64 or 128");
}
template void least_squares_cg_nm<float>(const imp_csr *, float *, const float *, size_t, const float *, int, int);
template void least_squares_cg_nm<__half>(const imp_csr *, __half *, const __half *, size_t, const float *, int, int);

}  // namespace imp
