// K4: top-k dot-product scorer behind recommend() / similar_items() / similar_users().
//
// Replaces KnnQuery::topk (implicit/gpu/knn.cu:77-265: cuBLAS GEMM + three Thrust passes + RAFT
// select_k).  Semantics follow the CPU oracle implicit/cpu/topk.pyx:15-67 + select.h:12-40:
//   scores = query . items^T (exact fp32), optional divide by item_norms, per-query (COO) and global
//   item filters set to -FLT_MAX, then the k best per row, written best-first.
// Output order (score desc, column desc) is the oracle's.  When more entries tie with the k-th score than
// there are places left, the retained set follows the reference heap's arrival-order rule exactly
// (closed form in select_kernel), so ids are bit-identical to select.h even on all-zero / all-filtered rows.
//
// Stage 1  score_gemm_kernel : fp32 MFMA (v_mfma_f32_32x32x2_f32), LDS-staged K-tiles of both operands
//          with odd row stride (conflict-free fragment reads), norm divide fused in the epilogue.
// Stage 2  filters (tiny scatter kernels).
// Stage 3  select_kernel : one workgroup per query row, MSB-first 8-bit radix select on the 64-bit key
//          (ordered(score) << 32 | column) over the score bytes with early exit once the boundary bucket
//          is taken whole, exact tie resolution otherwise, a gather of the k winners and an in-LDS bitonic sort.
#include <cfloat>
#include <type_traits>

#include "als_qtile.h"  // load4: fp32 / fp16 factor storage converted in registers
#include "common.h"

namespace imp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBM = 64;   // queries per block tile
constexpr int kBN = 128;  // items per block tile
constexpr int kBK = 64;   // factors per K step
constexpr int kLd = kBK + 1;

// scores[q][i] = sum_k Q[q][k] * I[i][k]   (optionally / norms[i])
__global__ __launch_bounds__(256) void score_gemm_kernel(const float *__restrict__ Q, int nq, const float *__restrict__ I,
                                                         int ni, int f, const float *__restrict__ norms,
                                                         float *__restrict__ S) {
  __shared__ float Qs[kBM * kLd];
  __shared__ float Is[kBN * kLd];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q0 = blockIdx.y * kBM, i0 = blockIdx.x * kBN;
  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  for (int k0 = 0; k0 < f; k0 += kBK) {
    // stage: consecutive threads walk the factor dimension -> coalesced global reads
    for (int e = tid; e < kBM * kBK; e += 256) {
      int r = e / kBK, c = e - r * kBK;
      int q = q0 + r, k = k0 + c;
      Qs[r * kLd + c] = (q < nq && k < f) ? Q[(size_t)q * f + k] : 0.f;
    }
    for (int e = tid; e < kBN * kBK; e += 256) {
      int r = e / kBK, c = e - r * kBK;
      int i = i0 + r, k = k0 + c;
      Is[r * kLd + c] = (i < ni && k < f) ? I[(size_t)i * f + k] : 0.f;
    }
    __syncthreads();
    // wave w: queries [0,64) x items [32w, 32w+32): two 32x32 tiles
    const int kh = lane >> 5, l31 = lane & 31;
#pragma unroll 8
    for (int kk = 0; kk < kBK; kk += 2) {
      float b = Is[(32 * wave + l31) * kLd + kk + kh];
      float a0 = Qs[l31 * kLd + kk + kh];
      float a1 = Qs[(32 + l31) * kLd + kk + kh];
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc[1], 0, 0, 0);
    }
    __syncthreads();
  }
  const int item = i0 + 32 * wave + (lane & 31);
  if (item < ni) {
    float inv_valid = norms ? norms[item] : 1.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        int q = q0 + 32 * t + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (q < nq) {
          float s = acc[t][e];
          if (norms) s = s / inv_valid;
          S[(size_t)q * ni + item] = s;
        }
      }
  }
}

__global__ void item_filter_kernel(float *__restrict__ S, int rows, int ni, const int32_t *__restrict__ items, int n_items) {
  size_t total = (size_t)rows * n_items;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int col = items[i % n_items];
    size_t row = i / n_items;
    if (col >= 0 && col < ni) S[row * ni + col] = -FLT_MAX;
  }
}

__global__ void coo_filter_kernel(float *__restrict__ S, int start, int end, int ni, const int32_t *__restrict__ row,
                                  const int32_t *__restrict__ col, size_t nnz) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nnz; i += (size_t)gridDim.x * blockDim.x) {
    int r = row[i], c = col[i];
    if (r >= start && r < end && c >= 0 && c < ni) S[(size_t)(r - start) * ni + c] = -FLT_MAX;
  }
}

__device__ __forceinline__ uint32_t ordered(float s) {
  uint32_t u = __float_as_uint(s);
  if (u == 0x80000000u) u = 0u;  // -0.0 compares equal to +0.0 in the reference's heap
  return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float unordered(uint32_t u) {
  return __uint_as_float(u ^ ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}
__device__ __forceinline__ uint64_t make_key(float s, int col) { return ((uint64_t)ordered(s) << 32) | (uint32_t)col; }

// One workgroup per query row.  `cand` = k 64-bit slots (LDS when it fits, else global scratch).
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void select_kernel(const float *__restrict__ S, int ni, int k, int kpad,
                                                       int32_t *__restrict__ out_ids, float *__restrict__ out_dist,
                                                       int out_stride, uint64_t *__restrict__ global_cand, int use_lds,
                                                       const int *__restrict__ only_flagged) {
  if (only_flagged && !only_flagged[blockIdx.x]) return;  // rows already finished by select_pruned_kernel
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ unsigned int hist[256];
  __shared__ unsigned int sh_bucket, sh_remaining, sh_count;
  uint64_t *cand = use_lds ? reinterpret_cast<uint64_t *>(smem_raw) : global_cand + (size_t)blockIdx.x * kpad;
  const float *row = S + (size_t)blockIdx.x * ni;
  const int tid = threadIdx.x;

  uint64_t prefix = 0, mask = 0;
  unsigned int remaining = k;  // k <= ni guaranteed by the host
  bool ambiguous = false;      // more keys tie with the k-th SCORE than there are places left
  for (int digit = 7; digit >= 4; --digit) {  // the four score bytes of the key
    const int shift = digit * 8;
    for (int i = tid; i < 256; i += BLOCK) hist[i] = 0;
    __syncthreads();
    for (int i = tid; i < ni; i += BLOCK) {
      uint64_t key = make_key(row[i], i);
      if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 0xFF], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      unsigned int acc = 0;
      int b = 255;
      for (; b > 0; --b) {
        if (acc + hist[b] >= remaining) break;
        acc += hist[b];
      }
      sh_bucket = b;
      sh_remaining = remaining - acc;
      sh_count = hist[b];
    }
    __syncthreads();
    prefix |= (uint64_t)sh_bucket << shift;
    mask |= (uint64_t)0xFF << shift;
    remaining = sh_remaining;
    bool whole = sh_count == remaining;
    __syncthreads();
    if (whole) break;  // boundary bucket taken entirely: no finer digits needed
    if (digit == 4) ambiguous = true;
  }

  if (tid == 0) sh_count = 0;
  for (int i = tid; i < kpad; i += BLOCK) cand[i] = 0;  // pads sort last
  __syncthreads();
  if (!ambiguous) {
    // (key & mask) >= prefix selects exactly k keys
    for (int i = tid; i < ni; i += BLOCK) {
      uint64_t key = make_key(row[i], i);
      if ((key & mask) >= prefix) {
        unsigned int slot = atomicAdd(&sh_count, 1u);
        if (slot < (unsigned)kpad) cand[slot] = key;
      }
    }
  } else {
    // Exact emulation of the reference's heap (implicit/cpu/select.h:12-40) for ties at the k-th score t:
    // tied entries enter in column order until the heap holds k entries >= t (saturation column s);
    // each later entry > t then evicts the tied entry with the LOWEST column.  So the survivors are the
    // tied columns <= s minus the e lowest ones, e = #{column > s : score > t}.
    const uint32_t t32 = (uint32_t)(prefix >> 32);
    __shared__ int sh_s;
    __shared__ unsigned int sh_e;
    if (tid < 64) {
      unsigned int running = 0, e = 0;
      int s_col = -1;
      for (int c0 = 0; c0 < ni; c0 += 64) {
        const int c = c0 + tid;
        const uint32_t key = c < ni ? ordered(row[c]) : 0u;
        const bool ge = c < ni && key >= t32, gt = c < ni && key > t32;
        const unsigned long long m_ge = __ballot(ge), m_gt = __ballot(gt);
        if (s_col < 0) {
          const unsigned int cnt = __popcll(m_ge);
          if (running + cnt >= (unsigned)k) {
            const unsigned int need = k - running;  // the need-th set bit of m_ge is the saturation column
            const unsigned int incl = __popcll(m_ge & ((2ull << tid) - 1ull));
            const unsigned long long hit = __ballot(ge && incl == need);
            const int L = __ffsll((long long)hit) - 1;
            s_col = c0 + L;
            e += __popcll(m_gt & ~((2ull << L) - 1ull));
          } else {
            running += cnt;
          }
        } else {
          e += __popcll(m_gt);
        }
      }
      if (tid == 0) {
        sh_s = s_col;
        sh_e = e;
      }
    }
    __syncthreads();
    for (int i = tid; i < ni; i += BLOCK) {  // everything strictly above the tie score
      const float sc = row[i];
      if (ordered(sc) > t32) {
        unsigned int slot = atomicAdd(&sh_count, 1u);
        if (slot < (unsigned)kpad) cand[slot] = make_key(sc, i);
      }
    }
    if (tid < 64) {  // the surviving ties, ranked by column
      const int s_col = sh_s;
      const unsigned int e = sh_e;
      unsigned int seen = 0;
      for (int c0 = 0; c0 <= s_col; c0 += 64) {
        const int c = c0 + tid;
        const float sc = c < ni ? row[c] : 0.f;
        const bool tie = c <= s_col && ordered(sc) == t32;
        const unsigned long long m_tie = __ballot(tie);
        const unsigned int rank = seen + __popcll(m_tie & ((1ull << tid) - 1ull)) + 1;
        if (tie && rank > e) {
          unsigned int slot = atomicAdd(&sh_count, 1u);
          if (slot < (unsigned)kpad) cand[slot] = make_key(sc, c);
        }
        seen += __popcll(m_tie);
      }
    }
  }
  __syncthreads();
  // bitonic sort, descending
  for (int size = 2; size <= kpad; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (kpad >> 1); t += BLOCK) {
        int lo = 2 * t - (t & (stride - 1));
        int hi = lo + stride;
        bool desc = ((lo & size) == 0);
        uint64_t a = cand[lo], b = cand[hi];
        if ((a < b) == desc) {
          cand[lo] = b;
          cand[hi] = a;
        }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < k; i += BLOCK) {
    uint64_t key = cand[i];
    out_ids[(size_t)blockIdx.x * out_stride + i] = (int32_t)(uint32_t)key;
    out_dist[(size_t)blockIdx.x * out_stride + i] = unordered((uint32_t)(key >> 32));
  }
}


// ---- fast path (f % 8 == 0): 128 x 128 block tile, 4 waves as 2 x 2, each wave 64 queries x 64 items = 2 x 2 MFMA tiles -----
// Split-bf16 form (f % 16 == 0; materialising path and fallback): v_mfma_f32_32x32x16_bf16 on three-way split operands, both
// operands staged per workgroup and 16-factor step by LDS-DMA (see the kernel).  fp16 form (f % 16 == 0; the emit path's two
// GEMMs, round 5): the same staging, two fp16 terms per value and three products (H2 at split8_f16).  Exact-fp32 form (v_mfma_f32_32x32x2_f32; f % 8 == 0 off the
// 16-grid, IMP_TOPK_FP32_MFMA=1): operands straight from global memory, no LDS, no barriers -- lane (r = l & 31, kh = l >> 5)
// loads ONE float4 = 4 consecutive factors [k0 + 4 kh, +4) of its query / item row per 8-factor block; MFMA step s of the
// block uses factor k0 + 4 kh + s for BOTH operands (the k index inside an MFMA is a free permutation), so one dwordx4 per
// operand tile feeds 4 MFMAs.  Epilogues: optional divide by the item norm, then the coalesced score write with the
// per-(query, 64-item) maximum for the pruned select below (MODE 0), the compact threshold subset (1) or the emit test (2).
constexpr int kTileItems = 64;  // granularity of the tile maxima

// MODE 0: scores written to S[q][item] + per-(query, 64-item) maxima          (materialising path)
// MODE 1: only every `block_stride`-th 128-item block is computed, into the compact S[q][128 b + ...] (threshold pre-pass)
// MODE 2: nothing is materialised: scores >= tau[q] that are not filtered (bitmaps) are appended to the query's candidate list
struct EmitArgs {
  const uint32_t *tau;        // [nq] ordered key of the query's threshold
  const uint32_t *row_bits;   // [nq][words] per-query filter bitmap (may be null)
  const uint32_t *item_bits;  // [words] global item filter bitmap (may be null)
  int words;
  uint64_t *cand;             // [nq][cap]
  unsigned int *count;        // [nq]
  int cap;
  float *row_unscale;         // [nq] resident form: the keys carry raw accumulators; factor that turns a row's into scores (else null)
  float *row_eps;             // [nq] screened emit pass (MODE 3): the error bound of the row's one-product accumulators (else null)
};

// TQ / TI: storage type of the query / item factors (float or __half).  fp16 factors are read as they are stored and
// converted in registers (8-byte loads; the reference hands fp16 operands straight to the GEMM with fp32 accumulation,
// implicit/gpu/knn.cu:117-128) -- bit-identical to scoring an fp32 copy, without writing and re-reading one per call.
//
// BF3 (f % 16 == 0, the default): the product on the bf16 matrix cores with every operand value split into three bf16 terms
// in registers (hi + mid + lo = the fp32 value to 2^-24) and the six partial products down to 2^-16 relative weight accumulated
// in fp32, smallest first ("3xBF16"): measured error against fp64 BELOW that of an fp32 FMA chain (2.6e-8 vs 2e-7 relative at
// f = 128), 24 v_mfma_f32_32x32x16_bf16 per 16 factors of a 64 x 64 wave tile instead of 32 v_mfma_f32_32x32x2_f32 at twice
// the instruction time each -- the fp32 MFMA runs at the VECTOR rate, 1/16 of the bf16 rate.  The splits (11 vector
// instructions per pair of values) run on the vector pipe beside the matrix pipe.  A and B fragments share the
// (lane >> 5, element) -> k assignment, which is all the contraction needs; C/D layout is that of the fp32 form.
typedef __bf16 tk_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split8_bf16(const float4 &v0, const float4 &v1, tk_bf16x8 &h, tk_bf16x8 &m, tk_bf16x8 &l) {
  const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const __bf16 hi = (__bf16)x[e];
    const float r1 = x[e] - (float)hi;
    const __bf16 mid = (__bf16)r1;
    h[e] = hi, m[e] = mid, l[e] = (__bf16)(r1 - (float)mid);
  }
}

// Query rows split ahead of the GEMM (emit path) and stored in FRAGMENT order: for every tile of 32 query rows, step of 16
// factors and term (hi, mid, lo -- the same bits split8_bf16 produces) the 64 lanes' 8-element operands lie side by side, so
// a wavefront's operand load is ONE contiguous KB (8 whole cache lines) instead of 64 pieces of 16 bytes in 32 lines, and the
// 2 x n_item_blocks workgroups that read a query block no longer repeat the split (1000 x 128 values once against 2285 times
// at configs[2]).  Rows are padded to whole 128-row query blocks with zeros.  `TQ = split_bf16` selects it in score_gemm_direct_kernel
// (Q then points at the first tile of the launch).  IMP_TOPK_NO_QSPLIT=1: split in the GEMM as for the items (A/B).
struct split_bf16 {
  __bf16 v;
};
template <typename T>
__global__ void split_query_rows_kernel(const T *__restrict__ Q, __bf16 *__restrict__ out, size_t rows, size_t rows_pad, int f) {
  const size_t n = rows_pad * (size_t)f;
  const int steps16 = f / 16;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t q = i / f;
    const int c = (int)(i - q * f);
    const float x = q < rows ? (float)Q[i] : 0.f;
    const __bf16 hi = (__bf16)x;
    const float r1 = x - (float)hi;
    const __bf16 mid = (__bf16)r1;
    const int lane = (int)(q & 31) + 32 * ((c >> 3) & 1);
    __bf16 *o = out + ((((q >> 5) * steps16 + (c >> 4)) * 3) * 64 + lane) * 8 + (c & 7);
    o[0] = hi, o[64 * 8] = mid, o[2 * 64 * 8] = (__bf16)(r1 - (float)mid);
  }
}

template <typename TQ> struct presplit {
  static constexpr int terms = 0;
};
template <> struct presplit<split_bf16> {
  static constexpr int terms = 3;
};

// four consecutive factors as they are stored, and their fp32 values
template <typename T> struct raw4 {
  using type = float4;
};
template <> struct raw4<__half> {
  using type = uint2;
};
template <typename T> using raw4_t = typename raw4<T>::type;
__device__ __forceinline__ float4 load_raw4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ uint2 load_raw4(const __half *p) { return *reinterpret_cast<const uint2 *>(p); }
__device__ __forceinline__ float4 widen4(const float4 &v) { return v; }
__device__ __forceinline__ float4 widen4(const uint2 &raw) {
  const float2 a = __half22float2(*reinterpret_cast<const __half2 *>(&raw.x));
  const float2 b = __half22float2(*reinterpret_cast<const __half2 *>(&raw.y));
  return make_float4(a.x, a.y, b.x, b.y);
}

#ifndef IMP_TOPK_MIN_WAVES
#define IMP_TOPK_MIN_WAVES 3  // waves per SIMD the register allocation leaves room for (split-bf16 emit GEMM at C3: 2 waves 0.83 ms, 3: 0.73, 4: 0.80)
#endif
template <int MODE, typename TQ = float, typename TI = float, bool BF3 = false>
__global__ __launch_bounds__(256, IMP_TOPK_MIN_WAVES) void score_gemm_direct_kernel(const TQ *__restrict__ Q, int nq, const TI *__restrict__ I,
                                                                int ni, int f, const float *__restrict__ norms,
                                                                float *__restrict__ S, float *__restrict__ tile_max,
                                                                int n_tiles, int block_stride, EmitArgs emit) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, kh = lane >> 5;
  const int q_base = blockIdx.y * 128 + 64 * (wave >> 1);
  const int i_base = (MODE == 1 ? blockIdx.x * block_stride : blockIdx.x) * 128 + 64 * (wave & 1);
  constexpr bool QS = presplit<TQ>::terms > 0;  // query rows already split, in fragment order: 3 bf16 or 2 fp16 terms per value
  constexpr int NT = QS ? presplit<TQ>::terms : 3;
  static_assert(!QS || BF3, "split query rows feed the matrix-core forms only");
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  if constexpr (BF3) {
    // 16 factors per step, both operands through LDS, staged ONCE per workgroup and step by LDS-DMA (no staging registers):
    // a wavefront load of lane (r, kh)'s own 16 bytes of its row touches 32 cache lines for 1 KB, and with every wave loading
    // its own operands each tile crossed L2 -> L1 twice; the texture addresser and the L2 port, not the matrix pipe, set the
    // pace (emit GEMM at configs[2]: 0.72 ms that way; 0.56 with the queries in fragment order; 0.52 with wave-private LDS
    // staging of the items; knock-outs: MFMAs 0.17 ms, loads 0.17, splits 0.07, epilogue 0.06-0.27).
    //  * factor ROWS (items; queries when they are not pre-split): 1 KB per DMA instruction = whole 64- / 32-byte row pieces
    //    (16 rows fp32, 32 rows fp16), the 16-byte chunks of a row XOR-swizzled -- on the SOURCE side, the DMA destination is
    //    lane-linear -- so that the 32 rows of a fragment read spread over all banks;
    //  * pre-split queries: 12 (8) fragment pieces of 1 KB (4 tiles x 3 bf16 / 2 fp16 terms), lane-linear as stored.
    // The 8 (4) + 12 (8) instructions of a step are dealt to the four waves; double buffer, one barrier per step: a wave requests
    // step s + 1 after the barrier of step s, which every wave reaches only with its fragment reads of step s - 1 consumed.
    constexpr int CHI = (int)sizeof(TI), CHQ = QS ? 4 : (int)sizeof(TQ);  // 16-byte chunks per row and step: 4 (fp32), 2 (fp16)
    constexpr int Q_BYTES = QS ? 4 * NT * 1024 : CHQ * 2048, I_BYTES = CHI * 2048;
    __shared__ __attribute__((aligned(1024))) unsigned char stage[2][Q_BYTES + I_BYTES];
    const int i_block = i_base - 64 * (wave & 1), q_block = q_base - 64 * (wave >> 1);
    // DMA sources of this wave (rows are clamped: what the extra rows produce is discarded)
    constexpr int NI_W = 2 * CHI / 4, NQ_W = QS ? NT : 2 * CHQ / 4;  // instructions per wave and step
    const TI *isrc[NI_W];
    const void *qsrc[NQ_W];
#pragma unroll
    for (int i = 0; i < NI_W; ++i) {
      const int j = wave + 4 * i, rho = lane / CHI, pos = lane % CHI, sw = (rho / (8 / CHI)) & (CHI - 1);
      isrc[i] = I + (size_t)min(i_block + j * (64 / CHI) + rho, ni - 1) * f + (pos ^ sw) * (16 / (int)sizeof(TI));
    }
#pragma unroll
    for (int i = 0; i < NQ_W; ++i) {
      if constexpr (QS) {
        const int idx = wave * NT + i, tile = idx / NT, plane = idx % NT;
        qsrc[i] = reinterpret_cast<const uint16_t *>(Q) + (((size_t)(q_block / 32 + tile) * (f / 16)) * NT + plane) * 512 + lane * 8;
      } else {
        const int j = wave + 4 * i, rho = lane / CHQ, pos = lane % CHQ, sw = (rho / (8 / CHQ)) & (CHQ - 1);
        qsrc[i] = Q + (size_t)min(q_block + j * (64 / CHQ) + rho, nq - 1) * f + (pos ^ sw) * (16 / (int)sizeof(TQ));
      }
    }
    auto dma = [&](int buf, int s16) {
#pragma unroll
      for (int i = 0; i < NQ_W; ++i) {
        const unsigned char *src = reinterpret_cast<const unsigned char *>(qsrc[i]) + (QS ? (size_t)s16 * NT * 1024 : (size_t)s16 * 16 * sizeof(TQ));
        const int slot = QS ? wave * NT + i : wave + 4 * i;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)&stage[buf][slot * 1024], 16, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < NI_W; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(isrc[i] + 16 * s16),
                                         (__attribute__((address_space(3))) void *)&stage[buf][Q_BYTES + (wave + 4 * i) * 1024], 16, 0, 0);
    };
    // fragment offsets of lane (r, kh): row R of a 128-row region -> piece R / rows-per-piece, swizzled chunk
    auto row_offset = [&](int R, int CH) {
      const int rpi = 64 / CH, j = R / rpi, rh = R % rpi, sw = (rh / (8 / CH)) & (CH - 1), c0 = CH == 4 ? 2 * kh : kh;
      return j * 1024 + (rh * CH + (c0 ^ sw)) * 16;
    };
    int q_off[2], i_off[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      i_off[t] = Q_BYTES + row_offset(64 * (wave & 1) + 32 * t + r, CHI);
      q_off[t] = QS ? ((2 * (wave >> 1) + t) * NT * 64 + lane) * 16 : row_offset(64 * (wave >> 1) + 32 * t + r, CHQ);
    }
    auto read_rows = [&](const unsigned char *base, int off, int CH, float4 &v0, float4 &v1) {
      if (CH == 4) {
        v0 = *reinterpret_cast<const float4 *>(base + off);
        v1 = *reinterpret_cast<const float4 *>(base + (off ^ 16));
      } else {
        const uint4 raw = *reinterpret_cast<const uint4 *>(base + off);
        v0 = widen4(uint2{raw.x, raw.y}), v1 = widen4(uint2{raw.z, raw.w});
      }
    };
    auto multiply16 = [&](int buf) {
      const unsigned char *base = &stage[buf][0];
      tk_bf16x8 ah[2], am[2], al[2], bh[2], bm[2], bl[2];
      float4 ra[2][2], rb[2][2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if constexpr (QS) {
          ah[t] = *reinterpret_cast<const tk_bf16x8 *>(base + q_off[t]);
          am[t] = *reinterpret_cast<const tk_bf16x8 *>(base + q_off[t] + 1024);
          al[t] = *reinterpret_cast<const tk_bf16x8 *>(base + q_off[t] + 2048);
        } else if constexpr (!QS) {
          read_rows(base, q_off[t], CHQ, ra[t][0], ra[t][1]);
        }
        read_rows(base, i_off[t], CHI, rb[t][0], rb[t][1]);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if constexpr (!QS) split8_bf16(ra[t][0], ra[t][1], ah[t], am[t], al[t]);
        split8_bf16(rb[t][0], rb[t][1], bh[t], bm[t], bl[t]);
      }
      return [=](auto &accr) {
#pragma unroll
        for (int tq = 0; tq < 2; ++tq)
#pragma unroll
          for (int ti = 0; ti < 2; ++ti) {
            f32x16 c = accr[tq][ti];
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[tq], bh[ti], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tq], bl[ti], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[tq], bm[ti], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[tq], bh[ti], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tq], bm[ti], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tq], bh[ti], c, 0, 0, 0);
            accr[tq][ti] = c;
          }
      };
    };
    const int steps16 = f / 16;
    dma(0, 0);
    for (int s16 = 0; s16 < steps16; ++s16) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of step s16 have landed ...
      __syncthreads();                                   // ... and so have everybody else's
      auto products = multiply16(s16 & 1);             // fragments out of LDS, splits
      if (s16 + 1 < steps16) dma((s16 + 1) & 1, s16 + 1);
      products(acc);
    }
  } else {
  // 8 factors per step; the operands of step s + 1 are requested before the 16 MFMAs of step s (two register sets), so
  // the L2 round trip of a step hides under the matrix work of the previous one
  const TQ *qp[2];
  const TI *ip[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    qp[t] = Q + (size_t)min(q_base + 32 * t + r, nq - 1) * f + 4 * kh;  // clamped rows: results are discarded
    ip[t] = I + (size_t)min(i_base + 32 * t + r, ni - 1) * f + 4 * kh;
  }
  raw4_t<TQ> a0[2], a1[2];
  raw4_t<TI> b0[2], b1[2];
  auto fetch = [&](raw4_t<TQ> (&a)[2], raw4_t<TI> (&b)[2], int k0) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      a[t] = load_raw4(qp[t] + k0);
      b[t] = load_raw4(ip[t] + k0);
    }
  };
  auto multiply = [&](const raw4_t<TQ> (&ar)[2], const raw4_t<TI> (&br)[2]) {
    const float4 a[2] = {widen4(ar[0]), widen4(ar[1])}, b[2] = {widen4(br[0]), widen4(br[1])};
#pragma unroll
    for (int tq = 0; tq < 2; ++tq)
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) {
        acc[tq][ti] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tq].x, b[ti].x, acc[tq][ti], 0, 0, 0);
        acc[tq][ti] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tq].y, b[ti].y, acc[tq][ti], 0, 0, 0);
        acc[tq][ti] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tq].z, b[ti].z, acc[tq][ti], 0, 0, 0);
        acc[tq][ti] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tq].w, b[ti].w, acc[tq][ti], 0, 0, 0);
      }
  };
  const int steps = f / 8;
  fetch(a0, b0, 0);
  int s = 0;
  for (; s + 2 <= steps; s += 2) {
    fetch(a1, b1, 8 * (s + 1));
    multiply(a0, b0);
    fetch(a0, b0, 8 * min(s + 2, steps - 1));  // past the end: re-reads the last step, unused
    multiply(a1, b1);
  }
  if (s < steps) multiply(a0, b0);
  }
  // C/D layout: column (item) = lane & 31, row (query) = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
  float nrm[2];
#pragma unroll
  for (int ti = 0; ti < 2; ++ti) {
    int item = i_base + 32 * ti + r;
    nrm[ti] = (norms && item < ni) ? norms[item] : 1.f;
  }
  // cosine scores: ONE wave-uniform branch around the 64 divisions.  Written per element ("if (norms) sc = sc / nrm") the
  // compiler turned the test into a select and every lane ran 64 IEEE division sequences (~700 vector instructions, more
  // than half of the main loop's count at f = 128) in recommend() calls, which have no norms at all.
  if (norms) {
#pragma unroll
    for (int tq = 0; tq < 2; ++tq)
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[tq][ti][e] = acc[tq][ti][e] / nrm[ti];
  }
  if constexpr (MODE == 1) {
    // compact layout: block b of the subset occupies columns [128 b, 128 b + 128); items past the end score -FLT_MAX
    const int sub_cols = gridDim.x * 128;
    const int c_base = blockIdx.x * 128 + 64 * (wave & 1);
#pragma unroll
    for (int tq = 0; tq < 2; ++tq)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int q = q_base + 32 * tq + (e & 3) + 8 * (e >> 2) + 4 * kh;
        if (q < nq) {
#pragma unroll
          for (int ti = 0; ti < 2; ++ti) {
            float sc = acc[tq][ti][e];
            S[(size_t)q * sub_cols + c_base + 32 * ti + r] = (i_base + 32 * ti + r < ni) ? sc : -FLT_MAX;
          }
        }
      }
    return;
  }
  if constexpr (MODE == 2) {
    // emission is rare (a few hundred scores per query row out of all items).  The thresholds of four consecutive query rows
    // come in one 16-byte load (the tau buffer is padded to whole 128-row blocks); a score is first compared with the
    // threshold as a float -- one instruction, true for every score the exact test accepts (and for a NaN) -- and only the
    // few that pass take the exact test on the ordered keys, the filter look-ups and the append.
#pragma unroll
    for (int tq = 0; tq < 2; ++tq)
#pragma unroll
      for (int eg = 0; eg < 4; ++eg) {
        const int q0 = q_base + 32 * tq + 8 * eg + 4 * kh;
        const uint4 t4 = *reinterpret_cast<const uint4 *>(emit.tau + q0);
        const uint32_t tk[4] = {t4.x, t4.y, t4.z, t4.w};
        // eight scores per lane against four thresholds, folded into one wave-wide test: most groups have no survivor
        bool any = false;
#pragma unroll
        for (int el = 0; el < 4; ++el) {
          const float tf = unordered(tk[el]);
          any |= !(acc[tq][0][4 * eg + el] < tf) | !(acc[tq][1][4 * eg + el] < tf);
        }
        if (__builtin_amdgcn_ballot_w64(any) == 0) continue;
#pragma unroll
        for (int el = 0; el < 4; ++el) {
          const int e = 4 * eg + el, q = q0 + el;
          const uint32_t t = tk[el];
          const float tf = unordered(t);
#pragma unroll
          for (int ti = 0; ti < 2; ++ti) {
            float sc = acc[tq][ti][e];
            if (!(sc < tf)) {
              if (!(sc == sc)) sc = INFINITY;  // (a NaN operand: select_candidates sends the row to the materialising path)
              const int item = i_base + 32 * ti + r;
              if (q < nq && item < ni && ordered(sc) >= t) {
                const uint32_t bit = 1u << (item & 31);
                bool filtered = emit.item_bits && (emit.item_bits[item >> 5] & bit);
                if (!filtered && emit.row_bits) filtered = emit.row_bits[(size_t)q * emit.words + (item >> 5)] & bit;
                if (!filtered) {
                  const unsigned int slot = atomicAdd(&emit.count[q], 1u);
                  if (slot < (unsigned)emit.cap) emit.cand[(size_t)q * emit.cap + slot] = make_key(sc, item);
                }
              }
            }
          }
        }
      }
    return;
  }
  const int tile = i_base / kTileItems;
  if (q_base + 64 <= nq && i_base + 64 <= ni) {
    // interior tile (wave-uniform): no per-element guards, so the 64 stores of a lane are queued back to back
#pragma unroll
    for (int tq = 0; tq < 2; ++tq)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int q = q_base + 32 * tq + (e & 3) + 8 * (e >> 2) + 4 * kh;
        float sc0 = acc[tq][0][e], sc1 = acc[tq][1][e];
        float *row = S + (size_t)q * ni + i_base + r;
        row[0] = sc0;
        row[32] = sc1;
        float m = fmaxf(sc0, sc1);
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
        if (r == 0) tile_max[(size_t)q * n_tiles + tile] = m;
      }
    return;
  }
#pragma unroll
  for (int tq = 0; tq < 2; ++tq)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int q = q_base + 32 * tq + (e & 3) + 8 * (e >> 2) + 4 * kh;
      float m = -FLT_MAX;
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) {
        const int item = i_base + 32 * ti + r;
        float sc = acc[tq][ti][e];
        if (item < ni) {
          if (q < nq) S[(size_t)q * ni + item] = sc;
          m = fmaxf(m, sc);
        }
      }
      // maximum over the 32 lanes that hold this query row (same lane >> 5)
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
      if (r == 0 && q < nq && tile < n_tiles) tile_max[(size_t)q * n_tiles + tile] = m;
    }
}

#include "topk_resident.h"  // the fp16 two-term form: queries resident in registers, cached item planes (round 6)

// After the filter scatters: recompute the maximum of every 64-item tile a filter touched (one thread per filter entry;
// several entries of one tile write the same value).  Keeps tau = the k-th largest tile maximum a valid AND tight lower
// bound without any slack for filtered entries.
__device__ __forceinline__ void refresh_tile_max(const float *__restrict__ S, float *__restrict__ tile_max, size_t row, int col,
                                                 int ni, int n_tiles) {
  const int tile = col / kTileItems;
  const int c0 = tile * kTileItems, c1 = min(ni, c0 + kTileItems);
  float m = -FLT_MAX;
  for (int c = c0; c < c1; ++c) m = fmaxf(m, S[row * ni + c]);
  tile_max[row * n_tiles + tile] = m;
}

__global__ void item_filter_refresh_kernel(const float *__restrict__ S, float *__restrict__ tile_max, int rows, int ni, int n_tiles,
                                           const int32_t *__restrict__ items, int n_items) {
  size_t total = (size_t)rows * n_items;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int col = items[i % n_items];
    if (col >= 0 && col < ni) refresh_tile_max(S, tile_max, i / n_items, col, ni, n_tiles);
  }
}

__global__ void coo_filter_refresh_kernel(const float *__restrict__ S, float *__restrict__ tile_max, int start, int end, int ni,
                                          int n_tiles, const int32_t *__restrict__ row, const int32_t *__restrict__ col,
                                          size_t nnz) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nnz; i += (size_t)gridDim.x * blockDim.x) {
    int r = row[i], c = col[i];
    if (r >= start && r < end && c >= 0 && c < ni) refresh_tile_max(S, tile_max, (size_t)(r - start), c, ni, n_tiles);
  }
}

// The best k of a candidate list sorted by (score desc, column desc) that holds EVERY surviving entry >= the k-th score
// (both callers guarantee it: their thresholds are lower bounds of the k-th score), ties at the k-th score included.
template <int BLOCK>
__device__ void write_best_k(const uint64_t *cand, unsigned int n_c, int k, int q, int32_t *__restrict__ out_ids,
                             float *__restrict__ out_dist, int out_stride, int *__restrict__ fallback, float unscale_q) {
  const int tid = threadIdx.x;
  const uint32_t t32 = (uint32_t)(cand[k - 1] >> 32);
  const bool tie = (int)n_c > k && t32 == (uint32_t)(cand[k] >> 32);
  if (!tie) {
    if (tid == 0) fallback[q] = 0;
    for (int i = tid; i < k; i += BLOCK) {
      uint64_t key = cand[i];
      out_ids[(size_t)q * out_stride + i] = (int32_t)(uint32_t)key;
      out_dist[(size_t)q * out_stride + i] = unordered((uint32_t)(key >> 32)) * unscale_q;
    }
    return;
  }
  // Exact tie at the k-th score t.  The reference's heap (implicit/cpu/select.h:12-40) keeps, of the entries tied at t,
  // those that arrived before the heap was full of entries >= t (saturation column s = column of the k-th entry >= t in
  // column order) minus the e lowest columns, e = #{column > s : score > t} (closed form derived in select_kernel).  Every
  // entry >= t is in the candidate list (t >= tau), so the rule can be evaluated right here: the list is sorted by
  // (score desc, column desc), i.e. [0, g) are the entries > t and [g, g + m) the ties, highest column first.
  __shared__ unsigned int sh_g, sh_m, sh_e, sh_above;
  __shared__ int sh_s;
  if (tid == 0) sh_g = sh_m = sh_e = sh_above = 0;
  __syncthreads();
  for (int i = tid; i < (int)n_c; i += BLOCK) {
    const uint32_t key32 = (uint32_t)(cand[i] >> 32);
    if (key32 > t32) atomicAdd(&sh_g, 1u);
    else if (key32 == t32) atomicAdd(&sh_m, 1u);
  }
  __syncthreads();
  const int g = (int)sh_g, m = (int)sh_m, ng = g + m;
  if (ng > 1024) {  // thousands of tied entries: the quadratic rank below is not worth it, the materialising path takes the row
    if (tid == 0) fallback[q] = 1;
    return;
  }
  for (int i = tid; i < ng; i += BLOCK) {  // rank in ascending column order; rank k-1 is the saturation column
    const uint32_t col = (uint32_t)cand[i];
    int rank = 0;
    for (int j = 0; j < ng; ++j) rank += (uint32_t)cand[j] < col;
    if (rank == k - 1) sh_s = (int)col;
  }
  __syncthreads();
  const uint32_t s_col = (uint32_t)sh_s;
  for (int i = tid; i < ng; i += BLOCK) {
    const uint32_t col = (uint32_t)cand[i];
    if (i < g && col > s_col) atomicAdd(&sh_e, 1u);      // later arrivals above t: each evicts the lowest tied column
    if (i >= g && col > s_col) atomicAdd(&sh_above, 1u);  // ties that arrived after saturation never entered
  }
  __syncthreads();
  const int i0 = g + (int)sh_above, i1 = g + m - (int)sh_e;  // surviving ties: [i0, i1) -- columns <= s minus the e lowest
  if (tid == 0) fallback[q] = (g + (i1 - i0) == k) ? 0 : 1;   // always k by construction; anything else -> the exact path
  for (int i = tid; i < ng; i += BLOCK) {
    int slot = -1;
    if (i < g) slot = i;
    else if (i >= i0 && i < i1) slot = g + (i - i0);
    if (slot >= 0 && slot < k) {
      const uint64_t key = cand[i];
      out_ids[(size_t)q * out_stride + slot] = (int32_t)(uint32_t)key;
      out_dist[(size_t)q * out_stride + slot] = unordered((uint32_t)(key >> 32)) * unscale_q;
    }
  }
}

// Pruned select: tau = the k-th largest tile maximum (maxima refreshed after the filters) is a lower bound of the k-th
// best surviving score: k distinct tiles each hold a surviving score >= tau.
// The tiles whose maximum reaches tau (about m of them) are scanned for scores >= tau (tens to a few hundred), which
// go to LDS; a bitonic sort orders them and the best k are written.  Rows that overflow the candidate buffer or have too few tiles raise `fallback[row]` and are redone by
// select_kernel; an exact tie at the k-th score is resolved inside the list (write_best_k: every entry >= the k-th score is in it).
constexpr int kCandCap = 4096;  // 32 KiB of LDS; heavy users (many liked items lower tau) need the headroom

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void select_pruned_kernel(const float *__restrict__ S, const float *__restrict__ tile_max,
                                                              int ni, int n_tiles, int k, int extra,
                                                              const int *__restrict__ filter_counts,
                                                              int32_t *__restrict__ out_ids, float *__restrict__ out_dist,
                                                              int out_stride, int *__restrict__ fallback) {
  __shared__ uint64_t cand[kCandCap];
  __shared__ unsigned int hist[256];
  __shared__ unsigned int sh_bucket, sh_remaining, sh_count;
  const int tid = threadIdx.x;
  const int q = blockIdx.x;
  const float *row = S + (size_t)q * ni;
  const float *tm = tile_max + (size_t)q * n_tiles;
  const int m = k + extra + (filter_counts ? filter_counts[q] : 0);
  if (m > n_tiles || k > kCandCap) {  // uniform
    if (tid == 0) fallback[q] = 1;
    return;
  }
  // m-th largest tile maximum.  Up to kRankTiles maxima (26 744 items: 418): by COUNTING -- the keys go to LDS (the upper half of
  // the candidate buffer, free until the scan below), every thread counts the keys greater than each of its own (index breaks
  // ties), the key with m - 1 greater ones is tau: one barrier instead of four histogram passes whose LDS atomics all land in
  // the two or three bins a row's maxima share (0.072 -> 0.02 ms per 1000 rows at configs[4]'s similar_items shape).  Beyond
  // that: 4-pass byte radix select over the (L2-resident) maxima
  constexpr int kRankTiles = 1024;
  uint32_t prefix = 0, mask = 0;
  unsigned int remaining = m;
  if (n_tiles <= kRankTiles) {
    uint32_t *keys = reinterpret_cast<uint32_t *>(cand + kCandCap / 2);
    for (int i = tid; i < n_tiles; i += BLOCK) keys[i] = ordered(tm[i]);
    __syncthreads();
    for (int i = tid; i < n_tiles; i += BLOCK) {
      const uint32_t key = keys[i];
      int greater = 0;
      for (int j = 0; j < n_tiles; ++j) {
        const uint32_t o = keys[j];
        greater += (o > key) || (o == key && j < i);
      }
      if (greater == m - 1) sh_bucket = key;
    }
    __syncthreads();
    prefix = sh_bucket;
    __syncthreads();  // (everybody has read tau and the keys: the candidate scan may overwrite them)
  } else
  for (int digit = 3; digit >= 0; --digit) {
    const int shift = digit * 8;
    for (int i = tid; i < 256; i += BLOCK) hist[i] = 0;
    __syncthreads();
    for (int i = tid; i < n_tiles; i += BLOCK) {
      uint32_t key = ordered(tm[i]);
      if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 0xFF], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      unsigned int acc = 0;
      int b = 255;
      for (; b > 0; --b) {
        if (acc + hist[b] >= remaining) break;
        acc += hist[b];
      }
      sh_bucket = b;
      sh_remaining = remaining - acc;
    }
    __syncthreads();
    prefix |= (uint32_t)sh_bucket << shift;
    mask |= 0xFFu << shift;
    remaining = sh_remaining;
    __syncthreads();
  }
  const uint32_t tau = prefix;  // exact key of the m-th largest maximum

  if (tid == 0) sh_count = 0;
  __syncthreads();
  // every score >= tau lives in a tile whose maximum is >= tau (the maxima are exact: the GEMM epilogue writes them
  // and the filter kernels refresh the tiles they touch), so only those tiles -- about m of the n_tiles -- are read
  // from the score row.  The hit tiles are listed first (LDS, the upper half of the candidate buffer) and their scores then
  // read by ALL threads with four independent loads per trip: one tile per wavefront and step (the first form) was a chain of
  // a dozen dependent round trips to the freshly written score row per wavefront
  {
    uint32_t *hit = reinterpret_cast<uint32_t *>(cand + kCandCap / 2);  // [<= n_tiles] (n_tiles <= 2 kCandCap: checked below)
    __shared__ unsigned int sh_hits;
    if (tid == 0) sh_hits = 0;
    __syncthreads();
    const bool listed = n_tiles <= kCandCap;  // (4 bytes per tile in a 16 KB half buffer)
    if (listed) {
      for (int t = tid; t < n_tiles; t += BLOCK)
        if (ordered(tm[t]) >= tau) hit[atomicAdd(&sh_hits, 1u)] = (uint32_t)t;
      __syncthreads();
      const unsigned int n_hit = sh_hits;
      // the list and the candidates share the buffer: candidates go to the lower half only while the list is being read
      // (more than kCandCap / 2 of them: the row overflows -> fallback, as a full buffer would)
      const unsigned int total = n_hit * kTileItems;
      for (unsigned int base = tid; base < total; base += 4 * BLOCK) {
        float sc[4];
        int item[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const unsigned int idx = base + u * BLOCK;
          item[u] = idx < total ? (int)(hit[idx / kTileItems] * kTileItems + idx % kTileItems) : ni;
          sc[u] = item[u] < ni ? row[item[u]] : -FLT_MAX;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (item[u] < ni && ordered(sc[u]) >= tau) {
            unsigned int slot = atomicAdd(&sh_count, 1u);
            if (slot < (unsigned)kCandCap / 2) cand[slot] = make_key(sc[u], item[u]);
          }
        }
      }
      __syncthreads();
      if (sh_count > (unsigned)kCandCap / 2) {  // uniform
        if (tid == 0) fallback[q] = 1;
        return;
      }
    } else {
      const int lane = tid & 63, wave = tid >> 6;
      for (int base = wave * 64; base < n_tiles; base += (BLOCK / 64) * 64) {
        const int t = base + lane;
        unsigned long long hits = __ballot(t < n_tiles && ordered(tm[t]) >= tau);
        while (hits) {
          const int item = (base + __builtin_ctzll(hits)) * kTileItems + lane;
          hits &= hits - 1;
          if (item < ni) {
            const float sc = row[item];
            if (ordered(sc) >= tau) {
              unsigned int slot = atomicAdd(&sh_count, 1u);
              if (slot < (unsigned)kCandCap) cand[slot] = make_key(sc, item);
            }
          }
        }
      }
    }
  }
  __syncthreads();
  const unsigned int n_c = sh_count;
  if (n_c > (unsigned)kCandCap || n_c < (unsigned)k) {
    if (tid == 0) fallback[q] = 1;
    return;
  }
  if (n_c <= 256u) {  // short list: ordered by counting (distinct keys: the column is part of them), one barrier
    uint64_t *sorted = cand + kCandCap / 2;
    for (int i = tid; i < (int)n_c; i += BLOCK) {
      const uint64_t key = cand[i];
      int rank = 0;
      for (int j = 0; j < (int)n_c; ++j) rank += cand[j] > key;
      sorted[rank] = key;
    }
    __syncthreads();
    write_best_k<BLOCK>(sorted, n_c, k, q, out_ids, out_dist, out_stride, fallback, 1.0f);
    return;
  }
  int npad = 2;
  while (npad < (int)n_c) npad <<= 1;
  for (int i = n_c + tid; i < npad; i += BLOCK) cand[i] = 0;
  __syncthreads();
  for (int size = 2; size <= npad; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (npad >> 1); t += BLOCK) {
        int lo = 2 * t - (t & (stride - 1));
        int hi = lo + stride;
        bool desc = ((lo & size) == 0);
        uint64_t a = cand[lo], b = cand[hi];
        if ((a < b) == desc) {
          cand[lo] = b;
          cand[hi] = a;
        }
      }
      __syncthreads();
    }
  }
  write_best_k<BLOCK>(cand, n_c, k, q, out_ids, out_dist, out_stride, fallback, 1.0f);
}

// ---- emit path: top-k without the score matrix -----------------------------------------------------------------------
// The materialising path writes and re-reads batch x items scores (1.17 GB per 1000-query batch at 292 K items: the
// GEMM ran at the speed of that write, ~1 TB/s, not of the matrix pipe).  Here:
//   1. pre-pass: the scores of every kSubStride-th 128-item block (3 % of the items) go to a small buffer, filters are
//      applied to it, and tau[q] = its k-th largest entry -- a valid lower bound of the k-th best score over ALL items,
//      because those k entries are themselves unfiltered scores of the row;
//   2. the full GEMM (MODE 2) appends every unfiltered score >= tau[q] to the query's candidate list (about
//      kSubStride x k of them): filters are looked up in bitmaps, only for the scores that pass the threshold;
//   3. one workgroup per query sorts its candidates and writes the best k.  Overflowing lists, lists shorter than k and
//      an exact tie at the k-th score (the reference heap's arrival-order rule needs the whole row) raise `fallback`:
//      those queries are redone by the materialising path, 64 at a time.
// Scores are the same MFMA accumulations in both passes (same kernel body, same k order): bit-identical.
constexpr int kSubStride = 32;   // default / largest stride of the pre-pass subset (emit_stride() lowers it for large k)
constexpr int kEmitCap = 4096;  // candidates per query (32 KiB: the LDS sort buffer of select_candidates_kernel)

__global__ void coo_bitmap_kernel(uint32_t *__restrict__ bits, int words, int start, int end, int ni, const int32_t *__restrict__ row,
                                  const int32_t *__restrict__ col, size_t nnz, float *__restrict__ S_sub, int sub_cols, int sub_stride) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nnz; i += (size_t)gridDim.x * blockDim.x) {
    const int r = row[i], c = col[i];
    if (r < start || r >= end || c < 0 || c >= ni) continue;
    atomicOr(&bits[(size_t)(r - start) * words + (c >> 5)], 1u << (c & 31));
    const int blk = c >> 7;
    if (S_sub && blk % sub_stride == 0) S_sub[(size_t)(r - start) * sub_cols + (blk / sub_stride) * 128 + (c & 127)] = -FLT_MAX;
  }
}

// the words coo_bitmap_kernel touched, back to zero: a batch leaves its per-query bitmap as it found it, so the next batch needs no
// 36 MB memset (1000 queries x 292 385 items) -- this is 44 K stores, queued behind the batch where nobody waits for it
__global__ void coo_bitmap_clear_kernel(uint32_t *__restrict__ bits, int words, int start, int end, int ni, const int32_t *__restrict__ row,
                                        const int32_t *__restrict__ col, size_t nnz) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nnz; i += (size_t)gridDim.x * blockDim.x) {
    const int r = row[i], c = col[i];
    if (r < start || r >= end || c < 0 || c >= ni) continue;
    bits[(size_t)(r - start) * words + (c >> 5)] = 0u;
  }
}

__global__ void item_bitmap_kernel(uint32_t *__restrict__ bits, int ni, const int32_t *__restrict__ items, int n_items,
                                   float *__restrict__ S_sub, int rows, int sub_cols, int sub_stride) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)n_items; i += (size_t)gridDim.x * blockDim.x) {
    const int c = items[i];
    if (c < 0 || c >= ni) continue;
    atomicOr(&bits[c >> 5], 1u << (c & 31));
    const int blk = c >> 7;
    if (blk % sub_stride == 0)
      for (int q = 0; q < rows; ++q) S_sub[(size_t)q * sub_cols + (blk / sub_stride) * 128 + (c & 127)] = -FLT_MAX;
  }
}

// key of the m-th largest of n values (4-pass byte radix select; one workgroup)
template <int BLOCK>
__device__ __forceinline__ uint32_t kth_largest_key(const float *__restrict__ v, int n, unsigned int m, unsigned int *hist,
                                                    unsigned int *sh_bucket, unsigned int *sh_remaining) {
  const int tid = threadIdx.x;
  uint32_t prefix = 0, mask = 0;
  unsigned int remaining = m;
  for (int digit = 3; digit >= 0; --digit) {
    const int shift = digit * 8;
    for (int i = tid; i < 256; i += BLOCK) hist[i] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += BLOCK) {
      uint32_t key = ordered(v[i]);
      if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 0xFF], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      unsigned int acc = 0;
      int b = 255;
      for (; b > 0; --b) {
        if (acc + hist[b] >= remaining) break;
        acc += hist[b];
      }
      *sh_bucket = b;
      *sh_remaining = remaining - acc;
    }
    __syncthreads();
    prefix |= (uint32_t)*sh_bucket << shift;
    mask |= 0xFFu << shift;
    remaining = *sh_remaining;
    __syncthreads();
  }
  return prefix;
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void subset_threshold_kernel(const float *__restrict__ S_sub, int sub_cols, int k,
                                                                 uint32_t *__restrict__ tau, unsigned int *__restrict__ count) {
  __shared__ unsigned int hist[256];
  __shared__ unsigned int sh_bucket, sh_remaining;
  const int q = blockIdx.x;
  const uint32_t key = kth_largest_key<BLOCK>(S_sub + (size_t)q * sub_cols, sub_cols, (unsigned)min(k, sub_cols), hist, &sh_bucket,
                                              &sh_remaining);
  if (threadIdx.x == 0) {
    tau[q] = key;  // ordered(-FLT_MAX) when fewer than k subset entries survive the filters: everything is emitted -> fallback
    count[q] = 0;
  }
}

// Small k (the recommend() case, k = 10).  The threshold only has to be a LOWER BOUND of the row's k-th best score, so the
// subset's exact k-th largest entry is not needed: every thread keeps the maximum of the entries it scanned (BLOCK disjoint
// groups) and tau = the k-th largest of those BLOCK maxima -- k distinct entries of the subset reach it.  With 256 groups and
// k = 10 it is the subset's exact k-th largest in 84 % of the rows and its (k+1)-th or (k+2)-th otherwise (a few more candidates).
// Each wavefront takes its k largest lane maxima out by DPP rounds (no LDS, no barrier), wavefront 0 merges the 4 x k survivors:
// ONE barrier per row instead of one per round with a rescan of 24 registers (round 3's exact form: 23 us per 1000 queries).
// Filtered entries carry ordered(-FLT_MAX); a row with fewer than k live groups gets that as tau: everything is emitted -> fallback.
__device__ __forceinline__ uint32_t wave_allmax_u32(uint32_t v) {
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, false));  // row_ror:8
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xF, 0xF, false));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x122, 0xF, 0xF, false));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xF, 0xF, false));  // every lane: its 16-lane row's maximum
  const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
  const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), r3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
  return max(max(r0, r1), max(r2, r3));
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void subset_threshold_groupmax_kernel(const float *__restrict__ S_sub, int sub_cols, int k,
                                                                          uint32_t *__restrict__ tau, unsigned int *__restrict__ count) {
  constexpr int WAVES = BLOCK / 64;
  static_assert(WAVES * 32 <= 128, "the merge holds two values per lane");
  __shared__ uint32_t part[WAVES * 32];
  const int tid = threadIdx.x, q = blockIdx.x, lane = tid & 63, wave = tid >> 6;
  const float *v = S_sub + (size_t)q * sub_cols;
  const uint32_t floor_key = ordered(-FLT_MAX);
  uint32_t m = floor_key;  // an empty group counts as filtered
  for (int i = tid; i < sub_cols; i += BLOCK) m = max(m, ordered(v[i]));
  const int rounds = min(k, 32);
  uint32_t mine = 0u;  // lane `it` of a wavefront ends with the wavefront's it-th largest group maximum; 0 = taken / none
  for (int it = 0; it < rounds; ++it) {
    const uint32_t top = wave_allmax_u32(m);
    if (lane == it) mine = top;
    const unsigned long long holders = __ballot(m == top);
    if (lane == (int)__builtin_ctzll(holders)) m = 0u;
  }
  if (lane < 32) part[wave * 32 + lane] = lane < rounds ? mine : 0u;
  __syncthreads();
  if (wave == 0) {
    uint32_t a = part[lane], b = part[64 + lane];
    uint32_t kth = 0u;
    for (int it = 0; it < rounds; ++it) {
      const uint32_t lm = max(a, b);
      kth = wave_allmax_u32(lm);
      const unsigned long long holders = __ballot(lm == kth);
      if (lane == (int)__builtin_ctzll(holders)) {
        if (a == kth) a = 0u;
        else b = 0u;
      }
    }
    if (lane == 0) {
      tau[q] = max(kth, floor_key);
      count[q] = 0;
    }
  }
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void select_candidates_kernel(const uint64_t *__restrict__ gcand, const unsigned int *__restrict__ count,
                                                                  int cap, int k, int32_t *__restrict__ out_ids,
                                                                  float *__restrict__ out_dist, int out_stride,
                                                                  int *__restrict__ fallback, const float *__restrict__ row_unscale) {
  __shared__ uint64_t cand[kEmitCap];
  const int tid = threadIdx.x, q = blockIdx.x;
  const float unscale_q = row_unscale ? row_unscale[q] : 1.f;  // (a power of two: exact)
  const unsigned int n_c = count[q];
  if (n_c > (unsigned)cap || n_c < (unsigned)k) {  // uniform
    if (tid == 0) fallback[q] = 1;
    return;
  }
  int npad = 2;
  while (npad < (int)n_c) npad <<= 1;
  for (int i = tid; i < npad; i += BLOCK) cand[i] = i < (int)n_c ? gcand[(size_t)q * cap + i] : 0;  // pads sort last
  __syncthreads();
  for (int size = 2; size <= npad; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (npad >> 1); t += BLOCK) {
        int lo = 2 * t - (t & (stride - 1));
        int hi = lo + stride;
        bool desc = ((lo & size) == 0);
        uint64_t a = cand[lo], b = cand[hi];
        if ((a < b) == desc) {
          cand[lo] = b;
          cand[hi] = a;
        }
      }
      __syncthreads();
    }
  }
  if ((uint32_t)(cand[0] >> 32) >= 0xFF800000u) {  // best candidate +inf / NaN (uniform): an operand left the fp16 form's range
    if (tid == 0) fallback[q] = 1;                 // (or the scores really are infinite) -- the exact path decides
    return;
  }
  write_best_k<BLOCK>(cand, n_c, k, q, out_ids, out_dist, out_stride, fallback, unscale_q);
}

// Candidates of the SCREENED emit pass (topk_resident.h MODE 3): the keys carry one-product accumulators, each within eps_q of the
// row's scaled exact score.  Sort by them; with A_k the k-th largest, every entry whose exact score reaches the k-th EXACT score
// has an accumulator >= A_k - 2 eps_q (k entries have exact scores >= A_k - eps_q, so the k-th exact score is at least that; an
// entry that reaches it lies at most eps_q below in the approximate order).  Those R entries -- k plus a handful -- are re-scored
// from the stored factors in fp32 (one wavefront per entry, the reference's own arithmetic: an fp32 dot product of the row and
// the item, implicit/gpu/knn.cu:131-147 / cpu/topk.pyx:45-47), ordered by (score desc, column desc) and written with the heap's
// tie rule.  More than kScreenCap of them (eps_q is a worst-case bound: tiny queries against a catalogue with huge outliers) or a
// non-finite score: the row goes to the exact path.
constexpr int kScreenCap = 1024;
#ifdef RQ_SCREEN_STATS  // variant builds: rows, candidates, re-scored entries of the screened select
__device__ unsigned long long rq_screen_stats[4];
#endif
template <int BLOCK, typename TQs, typename TIs>
__global__ __launch_bounds__(BLOCK) void select_screened_kernel(const uint64_t *__restrict__ gcand, const unsigned int *__restrict__ count,
                                                                int cap, int k, int32_t *__restrict__ out_ids, float *__restrict__ out_dist,
                                                                int out_stride, int *__restrict__ fallback, const float *__restrict__ row_eps,
                                                                const TQs *__restrict__ Q, const TIs *__restrict__ I, int f,
                                                                int *__restrict__ n_out) {
  __shared__ uint64_t cand[kEmitCap];
  __shared__ unsigned int sh_r;
  const int tid = threadIdx.x, q = blockIdx.x;
  const unsigned int n_c = count[q];
  if (tid == 0) n_out[q] = (int)min(n_c, (unsigned)cap + 1u);
  if (n_c > (unsigned)cap || n_c < (unsigned)k) {  // uniform
    if (tid == 0) fallback[q] = 1;
    return;
  }
  // the query row's share of this lane (EIGHT lanes per re-scored entry, four consecutive factors per lane and trip; f <= 256:
  // the resident form's limit), requested before anything else: its latency runs under the sort
  const int g = tid & 7;
  float4 qv[8];  // (f is a multiple of 8 on this path: whole 4-factor pieces, one vector load each)
  {
    const TQs *qrow = Q + (size_t)q * f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = 4 * g + 32 * j;
      qv[j] = c < f ? load4(qrow + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  int npad = 2;
  while (npad < (int)n_c) npad <<= 1;
  for (int i = tid; i < npad; i += BLOCK) cand[i] = i < (int)n_c ? gcand[(size_t)q * cap + i] : 0;  // pads sort last
  if (tid == 0) sh_r = 0;
  __syncthreads();
  auto sort_desc = [&](uint64_t *v, int n2) {
    for (int size = 2; size <= n2; size <<= 1) {
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int t = tid; t < (n2 >> 1); t += BLOCK) {
          int lo = 2 * t - (t & (stride - 1));
          int hi = lo + stride;
          bool desc = ((lo & size) == 0);
          uint64_t a = v[lo], b = v[hi];
          if ((a < b) == desc) {
            v[lo] = b;
            v[hi] = a;
          }
        }
        __syncthreads();
      }
    }
  };
  // short lists (the usual case: a few dozen entries) are ordered by counting -- entry i goes to the position "entries greater
  // than it" (keys are distinct: the column is part of them) -- one barrier instead of the bitonic network's log^2 n; beyond
  // kRankSort entries the quadratic count loses to the network
  auto rank_sort = [&](const uint64_t *in, uint64_t *out, int n) {
    for (int i = tid; i < n; i += BLOCK) {
      const uint64_t key = in[i];
      int rank = 0;
      for (int j = 0; j < n; ++j) rank += in[j] > key;
      out[rank] = key;
    }
    __syncthreads();
  };
  constexpr int kRankSort = 128;
  uint64_t *lst = cand;
  if (n_c <= (unsigned)kRankSort) {
    lst = cand + 2048;
    rank_sort(cand, lst, (int)n_c);
  } else {
    sort_desc(cand, npad);
  }
  const float eps = row_eps[q];
  const float a_k = unordered((uint32_t)(lst[k - 1] >> 32));
  const float cut = a_k - 2.f * eps;
  if (!(eps >= 0.f) || !(eps <= FLT_MAX) || !(a_k == a_k) || (uint32_t)(lst[0] >> 32) >= 0xFF800000u) {  // (uniform) NaN / inf somewhere
    if (tid == 0) fallback[q] = 1;
    return;
  }
  // R = entries with an accumulator >= cut: a prefix of the sorted list
  for (int i = tid; i < (int)n_c; i += BLOCK)
    if (unordered((uint32_t)(lst[i] >> 32)) >= cut) atomicMax(&sh_r, (unsigned)i + 1u);
  __syncthreads();
  const int n_r = (int)sh_r;
#ifdef RQ_SCREEN_STATS
  if (tid == 0) atomicAdd(&rq_screen_stats[0], 1ull), atomicAdd(&rq_screen_stats[1], (unsigned long long)n_c), atomicAdd(&rq_screen_stats[2], (unsigned long long)n_r), atomicMax(&rq_screen_stats[3], (unsigned long long)n_r);
#endif
  if (n_r > kScreenCap) {  // uniform
    if (tid == 0) fallback[q] = 1;
    return;
  }
  // exact scores of the R entries.  Nearly every row re-scores k plus one or two, but a row whose scores are all small against its
  // bound re-scores hundreds (624 seen on the bench's trained factors) and the launch lasts as long as its slowest row: eight
  // lanes per entry and two entries per group and trip = 128 gathers of a row in flight; fixed-order sum over the group
  bool bad = false;
  constexpr int PER = BLOCK / 8;
  for (int i0 = tid >> 3; i0 < n_r; i0 += 2 * PER) {
    const int i1 = i0 + PER;
    const bool two = i1 < n_r;
    const int item0 = (int)(uint32_t)lst[i0], item1 = two ? (int)(uint32_t)lst[i1] : item0;
#ifdef RQ_SCREEN_KO  // (timing only) 1: no gathers -- every entry reads item row 0;  2: no re-scoring at all
    const TIs *r0 = I + (size_t)((RQ_SCREEN_KO & 1) ? 0 : item0) * f, *r1 = I + (size_t)((RQ_SCREEN_KO & 1) ? 0 : item1) * f;
    if (RQ_SCREEN_KO & 2) break;
#else
    const TIs *r0 = I + (size_t)item0 * f, *r1 = I + (size_t)item1 * f;
#endif
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = 4 * g + 32 * j;
      if (c < f) {
        const float4 y0 = load4(r0 + c), y1 = load4(r1 + c);
        s0 = fmaf(qv[j].x, y0.x, s0), s0 = fmaf(qv[j].y, y0.y, s0), s0 = fmaf(qv[j].z, y0.z, s0), s0 = fmaf(qv[j].w, y0.w, s0);
        s1 = fmaf(qv[j].x, y1.x, s1), s1 = fmaf(qv[j].y, y1.y, s1), s1 = fmaf(qv[j].z, y1.z, s1), s1 = fmaf(qv[j].w, y1.w, s1);
      }
    }
    s0 += __shfl_xor(s0, 4, 64), s0 += __shfl_xor(s0, 2, 64), s0 += __shfl_xor(s0, 1, 64);
    s1 += __shfl_xor(s1, 4, 64), s1 += __shfl_xor(s1, 2, 64), s1 += __shfl_xor(s1, 1, 64);
    if (!(fabsf(s0) <= FLT_MAX) || !(fabsf(s1) <= FLT_MAX)) bad = true;
    if (g == 0) {
      lst[i0] = make_key(s0, item0);
      if (two) lst[i1] = make_key(s1, item1);
    }
  }
  if (__syncthreads_or(bad)) {
    if (tid == 0) fallback[q] = 1;
    return;
  }
  uint64_t *fin = lst;
  if (n_r <= kRankSort) {
    fin = lst == cand ? cand + 2048 : cand;  // (the two regions do not overlap)
    rank_sort(lst, fin, n_r);
  } else {  // (n_r <= kScreenCap = 1024: the pads stay inside the list's 2048-entry region)
    int rpad = 2;
    while (rpad < n_r) rpad <<= 1;
    for (int i = n_r + tid; i < rpad; i += BLOCK) lst[i] = 0;
    __syncthreads();
    sort_desc(lst, rpad);
  }
  write_best_k<BLOCK>(fin, (unsigned)n_r, k, q, out_ids, out_dist, out_stride, fallback, 1.f);
}

// factor counts that are not a multiple of 16 (the reference's CPU default is 100): rows zero-padded to the next multiple, fp16
// converted on the way -- extra zero factors change no dot product, and the scores come from the direct-operand kernels
template <typename T>
__global__ void pad_factor_rows_kernel(const T *__restrict__ src, float *__restrict__ dst, size_t rows, int f, int F) {
  const size_t n = rows * (size_t)F;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / F;
    const int c = (int)(i - r * F);
    dst[i] = c < f ? load1(src + r * f + c) : 0.f;
  }
}

// fallback rows: query rows gathered into a compact matrix, filters replayed from the bitmaps onto the materialised scores
template <typename TQ>
__global__ void gather_query_rows_kernel(const TQ *__restrict__ Q, const int32_t *__restrict__ rows, int n, int f,
                                         float *__restrict__ out) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)n * f; i += (size_t)gridDim.x * blockDim.x)
    out[i] = load1(Q + (size_t)rows[i / f] * f + i % f);
}

__global__ void bitmap_filter_kernel(float *__restrict__ S, float *__restrict__ tile_max, int ni, int n_tiles,
                                     const int32_t *__restrict__ rows, int n, const uint32_t *__restrict__ row_bits,
                                     const uint32_t *__restrict__ item_bits, int words) {
  // one thread per (row, 64-item tile): clears the filtered scores of the tile and recomputes its maximum if any was hit
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)n * n_tiles; i += (size_t)gridDim.x * blockDim.x) {
    const int j = (int)(i / n_tiles), tile = (int)(i % n_tiles);
    uint32_t w0 = 0, w1 = 0;
    const int wi = 2 * tile;
    if (item_bits) {
      w0 |= item_bits[wi];
      if (wi + 1 < words) w1 |= item_bits[wi + 1];
    }
    if (row_bits) {
      const uint32_t *rb = row_bits + (size_t)rows[j] * words;
      w0 |= rb[wi];
      if (wi + 1 < words) w1 |= rb[wi + 1];
    }
    if (!(w0 | w1)) continue;
    float *srow = S + (size_t)j * ni;
    float m = -FLT_MAX;
    const int c0 = tile * kTileItems, c1 = min(ni, c0 + kTileItems);
    for (int c = c0; c < c1; ++c) {
      const uint32_t w = (c - c0) < 32 ? w0 : w1;
      if (w & (1u << (c & 31))) srow[c] = -FLT_MAX;
      m = fmaxf(m, srow[c]);
    }
    tile_max[(size_t)j * n_tiles + tile] = m;
  }
}

__global__ void scatter_topk_rows_kernel(const int32_t *__restrict__ ids, const float *__restrict__ dist, const int32_t *__restrict__ rows,
                                         int n, int k, int32_t *__restrict__ out_ids, float *__restrict__ out_dist) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)n * k; i += (size_t)gridDim.x * blockDim.x) {
    const size_t dst = (size_t)rows[i / k] * k + i % k;
    out_ids[dst] = ids[i];
    out_dist[dst] = dist[i];
  }
}

static bool is_host_pointer(const void *p) {
  hipPointerAttribute_t attr;
  hipError_t err = hipPointerGetAttributes(&attr, p);
  if (err != hipSuccess) {
    (void)hipGetLastError();  // clear
    return true;
  }
  return attr.type == hipMemoryTypeHost || attr.type == hipMemoryTypeUnregistered;
}

}  // namespace imp

using namespace imp;

extern "C" int imp_matrix_astype(const imp_matrix *src, size_t itemsize, imp_matrix **out);

struct imp_knn {
  size_t max_temp_memory = 0;
  // screened emit path: multiplier (1, 2, 4) of the threshold pre-pass's subset stride, steered by the candidate counts of the
  // previous batch of the same (catalogue size, k) -- see the feedback rule in imp_knn_topk
  int stride_boost = 1;
  size_t boost_ni = 0;
  int boost_k = 0;
  // the per-query filter bitmap is all zeros between batches (every batch clears the words it set); true: not known to be (fresh
  // or regrown memory, a call that ended in an error) -- the next batch zeroes its part wholesale first
  bool row_bits_dirty = true;
  // persistent workspaces (grown on demand): no hipMalloc on the query path after the first call
  DeviceArray<float> scores, tile_max;
  DeviceArray<uint64_t> gcand;
  DeviceArray<int32_t> dev_ids, counts, fallback;
  DeviceArray<float> dev_dist;
  // emit path
  DeviceArray<float> sub_scores, fb_query, fb_dist;
  DeviceArray<float> pad_items, pad_query;  // zero-padded fp32 copies for factor counts that are not a multiple of 16
  DeviceArray<split_bf16> query_split;      // [nq][3][f] bf16 (or [nq][2][f] fp16) terms of the query rows (emit path, split forms)
  // fp16 two-term form (topk_resident.h): the item matrix as fragment-ordered planes, made once per catalogue VERSION -- `key`
  // names the memory they were made from and is cleared by any write to it through the library (note_device_write); memory the
  // library cannot vouch for (wrapped foreign pointers, matrices whose address was handed out) is split again on every call
  struct ItemPlanes {
    DerivedCache key;
    DeviceArray<_Float16> planes;
    DeviceArray<int> exp;          // scale exponent of the item matrix (from its exact maximum)
    DeviceArray<unsigned> maxbits;
    DeviceArray<unsigned> ne;      // bits of max || y 2^e ||_2, max || y 2^e - high plane ||_2, max ratio of the two (screened emit pass)
    DeviceArray<unsigned> tile_n;  // bits of max || y 2^e ||_2 per 32-item tile
    size_t rows = 0, cols = 0, itemsize = 0;
    int KS = 0;
    ItemPlanes() { register_derived_cache(&key); }
    ~ItemPlanes() { unregister_derived_cache(&key); }
  } item_planes;
  DeviceArray<_Float16> query_planes;
  DeviceArray<int> query_exp;
  DeviceArray<float> query_err;    // [2][rows]: || q 2^e - high plane ||, || high plane || per query row of the call
  DeviceArray<float> row_eps;
  DeviceArray<float> row_unscale;  // per row of an emit batch: what turns the raw accumulators its candidate keys carry into scores
  DeviceArray<uint32_t> tau, row_bits, item_bits;
  DeviceArray<unsigned int> cand_count;
  DeviceArray<uint64_t> cand;
  DeviceArray<int32_t> fb_rows, fb_ids;
  // page-locked, device-addressable host memory of the emit path: fallback flags and (host outputs) ids / scores are written
  // there by the kernels themselves (see imp_knn_topk)
  struct Pinned {
    void *p = nullptr;
    size_t bytes = 0;
    void *ensure(size_t n) {
      if (bytes < n) {
        if (p) (void)hipHostFree(p);
        p = nullptr, bytes = 0;
        IMP_CHECK_HIP(hipHostMalloc(&p, n, hipHostMallocDefault));
        bytes = n;
      }
      return p;
    }
    ~Pinned() {
      if (p) (void)hipHostFree(p);
    }
  } host_stage;
  template <typename T> static T *ensure(DeviceArray<T> &a, size_t n) {
    if (a.size < n) a.alloc(n);
    return a.data();
  }
};

extern "C" {

int imp_knn_create(size_t max_temp_memory, imp_knn **out) {
  return guarded([&] {
    (void)ctx();
    if (!max_temp_memory) {
      size_t free_b = 0, total_b = 0;
      IMP_CHECK_HIP(hipMemGetInfo(&free_b, &total_b));
      max_temp_memory = std::min<size_t>(free_b / 2, (size_t)4 << 30);
    }
    auto k = new imp_knn();
    k->max_temp_memory = max_temp_memory;
    *out = k;
  });
}

int imp_knn_destroy(imp_knn *k) {
  return guarded([&] { delete k; });
}

int imp_knn_topk(imp_knn *knn, const imp_matrix *items_in, const imp_matrix *query_in, int k, int32_t *indices, float *distances,
                 const imp_matrix *item_norms, const imp_coo *query_filter, const imp_intvector *item_filter) {
  return guarded([&] {
    if (query_in->cols != items_in->cols) throw std::invalid_argument("Must have same number of columns in each matrix for topk");
    if (query_in->itemsize != items_in->itemsize) throw std::invalid_argument("Must have same dtype in each matrix for topk");
    if (k < 0) throw std::invalid_argument("k must be >= 0 for topk");
    if (item_norms && (item_norms->itemsize != 4 || item_norms->rows * item_norms->cols != items_in->rows))
      throw std::invalid_argument("item_norms must be a float32 matrix with one entry per item");
    const size_t nq = query_in->rows, ni = items_in->rows;
    const int f_in = (int)items_in->cols;
    if (nq == 0 || k == 0) return;
    if (ni > (size_t)INT32_MAX) throw std::invalid_argument("too many items for topk");

    const int k_eff = (int)std::min<size_t>((size_t)k, ni);
    // fast path: direct-operand MFMA GEMM (+ emit path, or tile maxima + single-pass pruned select)
    static const bool no_fast = getenv("IMP_TOPK_NO_FAST") != nullptr;
    // factor counts off the 16-grid ride the fast path on zero-padded fp32 copies (272 K -> 1.1 M recs/s at f = 100, configs[2]
    // items; the copies cost ~0.1 ms per call at that size)
    const bool padded = !no_fast && f_in % 16 != 0 && f_in >= 1 && k_eff <= kCandCap;
    const int f = padded ? (f_in + 15) / 16 * 16 : f_in;
    const bool fast = !no_fast && (f % 8 == 0) && k_eff <= kCandCap;
    // fp16 factors (reference: SgemmEx on fp16 operands with fp32 accumulation, knn.cu:117-128): the direct-operand kernels
    // read them as stored and convert in registers; only the general path (any f, LDS-staged GEMM) scores an fp32 copy
    const bool half_direct = items_in->itemsize == 2 && fast && !padded;
    std::unique_ptr<imp_matrix> items_conv, query_conv;
    const imp_matrix *items = items_in, *query = query_in;
    imp_matrix items_pad, query_pad;  // views of the padded workspaces (no ownership)
    if (padded) {
      float *pi = imp_knn::ensure(knn->pad_items, ni * (size_t)f), *pq = imp_knn::ensure(knn->pad_query, nq * (size_t)f);
      IMP_PROF("pad_factors");
      auto grid = [&](size_t n) { return (int)std::max<size_t>(1, std::min<size_t>((n + 255) / 256, (size_t)ctx().num_cus * 16)); };
      if (items_in->itemsize == 4) {
        pad_factor_rows_kernel<float><<<grid(ni * f), 256, 0, stream()>>>(items_in->f32(), pi, ni, f_in, f);
        pad_factor_rows_kernel<float><<<grid(nq * f), 256, 0, stream()>>>(query_in->f32(), pq, nq, f_in, f);
      } else {
        pad_factor_rows_kernel<__half><<<grid(ni * f), 256, 0, stream()>>>(reinterpret_cast<const __half *>(items_in->data), pi, ni, f_in, f);
        pad_factor_rows_kernel<__half><<<grid(nq * f), 256, 0, stream()>>>(reinterpret_cast<const __half *>(query_in->data), pq, nq, f_in, f);
      }
      IMP_CHECK_HIP(hipGetLastError());
      items_pad.rows = ni, items_pad.cols = (size_t)f, items_pad.itemsize = 4, items_pad.data = pi;
      query_pad.rows = nq, query_pad.cols = (size_t)f, query_pad.itemsize = 4, query_pad.data = pq;
      items = &items_pad;
      query = &query_pad;
    } else if (items_in->itemsize == 2 && !half_direct) {
      imp_matrix *t = nullptr;
      if (imp_matrix_astype(items_in, 4, &t) != IMP_OK) throw std::runtime_error(imp_last_error());
      items_conv.reset(t);
      if (imp_matrix_astype(query_in, 4, &t) != IMP_OK) throw std::runtime_error(imp_last_error());
      query_conv.reset(t);
      items = items_conv.get();
      query = query_conv.get();
    }
    int kpad = 1;
    while (kpad < k_eff) kpad <<= 1;
    if (kpad < 2) kpad = 2;

    const bool host_ids = is_host_pointer(indices), host_dist = is_host_pointer(distances);
    int32_t *d_ids = indices;
    float *d_dist = distances;
    // Host outputs: the select kernels write ids and scores (and the emit path its fallback flags) STRAIGHT into page-locked
    // host memory the device can address -- [2048 flags][ids of the call][scores of the call] -- and after the host wait they
    // are simply there.  Any D2H copy instead costs more than the whole candidate sort: into pageable memory (the caller's
    // numpy arrays) the runtime stages it with a host wait of its own (three per emit call: ~0.1 of a 0.59 ms call), and an
    // ASYNCHRONOUS copy queued behind the kernels, page-locked or not, took ~0.4 ms to start on this stack (0.59 -> 1.0 ms
    // per call, gpurun_out/r4o, r4q).  Very large results (> 64 MB per array) keep the device buffers and the copies.
    constexpr size_t kFlagSlots = 2048;  // = the emit path's batch
    const size_t out_words = nq * (size_t)k;
    const bool stage_results = (host_ids || host_dist) && out_words <= ((size_t)16 << 20);
    int *host_flags = static_cast<int *>(knn->host_stage.ensure((2 * kFlagSlots + (stage_results ? 2 * out_words : 0)) * 4));
    int *host_counts = host_flags + kFlagSlots;  // screened select: the length of every row's candidate list (feeds the stride rule)
    int32_t *stage_ids = reinterpret_cast<int32_t *>(host_flags + 2 * kFlagSlots);
    float *stage_dist = reinterpret_cast<float *>(host_flags + 2 * kFlagSlots + out_words);
    if (host_ids) {
      if (stage_results) {
        d_ids = stage_ids;
        if (k_eff < k) std::copy(indices, indices + out_words, stage_ids);  // entries past k_eff keep the caller's initial values (topk.pyx:20-21 zero-fills them)
      } else {
        d_ids = imp_knn::ensure(knn->dev_ids, out_words);
        if (k_eff < k) IMP_CHECK_HIP(hipMemcpyAsync(d_ids, indices, out_words * 4, hipMemcpyHostToDevice, stream()));
      }
    }
    if (host_dist) {
      if (stage_results) {
        d_dist = stage_dist;
        if (k_eff < k) std::copy(distances, distances + out_words, stage_dist);
      } else {
        d_dist = imp_knn::ensure(knn->dev_dist, out_words);
        if (k_eff < k) IMP_CHECK_HIP(hipMemcpyAsync(d_dist, distances, out_words * 4, hipMemcpyHostToDevice, stream()));
      }
    }
    auto deliver = [&] {  // end of a call: wait, then hand the results over
      if (stage_results) {
        sync();
        if (host_ids) std::copy(stage_ids, stage_ids + out_words, indices);
        if (host_dist) std::copy(stage_dist, stage_dist + out_words, distances);
      } else {
        if (host_ids) IMP_CHECK_HIP(hipMemcpyAsync(indices, d_ids, out_words * 4, hipMemcpyDeviceToHost, stream()));
        if (host_dist) IMP_CHECK_HIP(hipMemcpyAsync(distances, d_dist, out_words * 4, hipMemcpyDeviceToHost, stream()));
        sync();
      }
    };

    size_t temp = std::min<size_t>(knn->max_temp_memory, (size_t)4 << 30);
    size_t batch = std::max<size_t>(1, std::min<size_t>(nq, temp / (sizeof(float) * ni)));
    static const bool no_emit_alloc = getenv("IMP_TOPK_NO_EMIT") != nullptr;
    // emit path (no score matrix) when the candidate lists stay SPARSE: every `stride`-th 128-item block is scored first and
    // about stride x k entries per query survive its threshold.  The stride is chosen so that they fill at most half of a
    // query's kEmitCap slots AND are at most one in 64 of the items -- every 64-item tile with a survivor costs an atomic on
    // the query's counter and a scattered store (measured at configs[4]'s similar_items shape, 26 744 items, k = 100,
    // stride 20: 7.5 % of all scores survive and the emit GEMM runs at 10 TFLOP/s against 33 for the materialising path) --
    // and a stride below 8 (a pre-pass of more than an eighth of the GEMM) is not worth it either.
    // The stride is a trade between the pre-pass (scores of 1 / stride of the items materialised, filtered, selected from) and the
    // candidate lists (about stride x k entries on unstructured data).  On TRAINED factors a row's best scores stand far above the
    // bulk and the lists stay short (22 entries at stride 32 on the bench's factors), so the screened path lets the stride grow:
    // after every batch the mean list length decides -- under 48: twice the stride next time (up to 4 x 32), over 400 or more than
    // one row in a hundred sent to the exact path: half.  Results do not depend on the stride; random factors stay at 32
    // (stride 128 there: 1280-entry lists and every row's staging overflowing -- measured, profiles/r06_topk_resident_knockouts.txt).
    if (knn->boost_ni != ni || knn->boost_k != k_eff) knn->stride_boost = 1, knn->boost_ni = ni, knn->boost_k = k_eff;
    const int stride_cap = item_norms ? kSubStride : kSubStride * knn->stride_boost;
    const int stride = (int)std::min<size_t>(std::min(stride_cap, kEmitCap / (2 * std::max(1, k_eff))), ni / ((size_t)64 * std::max(1, k_eff)));
    const bool emit_shape = (f % 8 == 0) && k_eff == k && k_eff <= 256 && stride >= 8;
    const bool will_emit = !no_emit_alloc && getenv("IMP_TOPK_NO_FAST") == nullptr && emit_shape;
    float *scores = will_emit ? nullptr : imp_knn::ensure(knn->scores, batch * ni);  // the emit path materialises fallback rows only
    const bool use_lds = (size_t)kpad * 8 <= 96 * 1024;
    uint64_t *gcand = use_lds ? nullptr : imp_knn::ensure(knn->gcand, batch * (size_t)kpad);

    const int n_tiles = (int)((ni + kTileItems - 1) / kTileItems);
    float *tile_max = (fast && !will_emit) ? imp_knn::ensure(knn->tile_max, batch * (size_t)n_tiles) : nullptr;
    int *fallback = (fast && !will_emit) ? imp_knn::ensure(knn->fallback, batch) : nullptr;
    int *counts = nullptr;  // tile maxima are refreshed after the filters: no slack for filtered entries is needed
    const int extra = 0;

    auto run = [&](const auto *Qb, const auto *Ib, auto Bf3c) {
      using TQ = std::remove_cv_t<std::remove_pointer_t<decltype(Qb)>>;
      using TI = std::remove_cv_t<std::remove_pointer_t<decltype(Ib)>>;
      constexpr bool BF3 = decltype(Bf3c)::value;
    // fp16 two-term planes for the resident-query kernels (topk_resident.h): the item matrix once per catalogue version (cached in
    // the handle), the query rows of this call with one scale per row
    const float *q_err_a = nullptr, *q_err_b = nullptr;
    const unsigned *item_ne = nullptr;
    const float *item_tile_n = nullptr;
    auto prepare_planes = [&](int KS, const _Float16 *&iplanes, const int *&iexp, const _Float16 *&qplanes, const int *&qexp) {
        IMP_PROF("split_query_rows");
        auto &ip = knn->item_planes;
        const size_t ni_pad = (ni + 127) / 128 * 128, F = (size_t)KS * 16;
        const bool same = ip.key.src == items_in->data && ip.rows == ni && ip.cols == (size_t)f_in && ip.itemsize == items_in->itemsize && ip.KS == KS;
        if (!same) {
          ip.key.src = nullptr;
          if (ip.planes.size < ni_pad * F * 2) ip.planes.alloc(ni_pad * F * 2);
          if (ip.exp.size < 1) ip.exp.alloc(1), ip.maxbits.alloc(1), ip.ne.alloc(4);
          if (ip.tile_n.size < ni_pad / 32) ip.tile_n.alloc(ni_pad / 32);
          IMP_CHECK_HIP(hipMemsetAsync(ip.maxbits.data(), 0, sizeof(unsigned), stream()));
          IMP_CHECK_HIP(hipMemsetAsync(ip.ne.data(), 0, 4 * sizeof(unsigned), stream()));
          IMP_CHECK_HIP(hipMemsetAsync(ip.tile_n.data(), 0, (ni_pad / 32) * sizeof(unsigned), stream()));
          const int g1 = (int)std::max<size_t>(1, std::min<size_t>((ni * (size_t)f + 255) / 256, (size_t)ctx().num_cus * 8));
          rq_absmax_kernel<TI><<<g1, 256, 0, stream()>>>(Ib, ni * (size_t)f, ip.maxbits.data());
          rq_item_exp_kernel<<<1, 1, 0, stream()>>>(ip.maxbits.data(), ip.exp.data());
          const int g2 = (int)std::max<size_t>(1, std::min<size_t>((ni_pad * (F / 8) + 255) / 256, (size_t)ctx().num_cus * 16));
          rq_split_items_kernel<TI><<<g2, 256, 0, stream()>>>(Ib, ip.planes.data(), ni, ni_pad, f, KS, ip.exp.data());
          rq_item_err_kernel<TI><<<(int)std::min<size_t>((ni + 3) / 4, (size_t)ctx().num_cus * 16), 256, 0, stream()>>>(Ib, ni, f, ip.exp.data(),
                                                                                                                    ip.ne.data(), ip.tile_n.data());
          ip.rows = ni, ip.cols = (size_t)f_in, ip.itemsize = items_in->itemsize, ip.KS = KS;
          const bool trusted = items_in->storage && items_in->storage->owned && !items_in->storage->exposed;
          if (trusted) ip.key.src = items_in->data, ip.key.bytes = items_in->bytes();
        }
        iplanes = ip.planes.data(), iexp = ip.exp.data();
        const size_t nq_pad = rq_query_pad(nq);
        _Float16 *qp = imp_knn::ensure(knn->query_planes, nq_pad * F * 2);
        int *qe = imp_knn::ensure(knn->query_exp, nq_pad);
        float *qerr = imp_knn::ensure(knn->query_err, 2 * nq_pad);
        rq_split_queries_kernel<TQ><<<(int)std::min<size_t>((nq_pad + 3) / 4, (size_t)ctx().num_cus * 8), 256, 0, stream()>>>(Qb, qp, qe, nq, nq_pad, f, KS,
                                                                                                                             qerr, qerr + nq_pad);
        IMP_CHECK_HIP(hipGetLastError());
        qplanes = qp, qexp = qe;
        q_err_a = qerr, q_err_b = qerr + nq_pad, item_ne = ip.ne.data(), item_tile_n = reinterpret_cast<const float *>(ip.tile_n.data());
    };
    static const bool resident_env = !(getenv("IMP_TOPK_RESIDENT") && atoi(getenv("IMP_TOPK_RESIDENT")) == 0);
    constexpr bool bf16x3 = false;
    // emit path (no score matrix): large item sets, k small against the candidate lists
    static const bool no_emit = getenv("IMP_TOPK_NO_EMIT") != nullptr;
    const bool emit_path = fast && !no_emit && emit_shape;
    if (emit_path) {
      const float *norms = item_norms ? item_norms->f32() : nullptr;
      const int words = (int)((ni + 31) / 32);
      const int n_blocks = (int)((ni + 127) / 128), n_sub = (n_blocks + stride - 1) / stride, sub_cols = n_sub * 128;
      const size_t ebatch = std::min<size_t>(nq, 2048);
      constexpr int FB = 64;  // fallback rows per materialised group
      float *sub = imp_knn::ensure(knn->sub_scores, ebatch * (size_t)sub_cols);
      uint32_t *tau = imp_knn::ensure(knn->tau, (ebatch + 127) / 128 * 128);  // the emit epilogue loads thresholds four rows at a time
      unsigned int *cnt = imp_knn::ensure(knn->cand_count, ebatch);
      uint64_t *cand = imp_knn::ensure(knn->cand, ebatch * (size_t)kEmitCap);
      float *row_unscale = imp_knn::ensure(knn->row_unscale, rq_query_pad(ebatch));
      int *fallback_e = host_flags;  // one flag per row of the batch, read by the host after the batch's wait
      const bool have_coo = query_filter && query_filter->nnz, have_items = item_filter && item_filter->size;
      if (have_coo && knn->row_bits.size < ebatch * (size_t)words) knn->row_bits_dirty = true;  // (regrown: fresh memory)
      uint32_t *row_bits = have_coo ? imp_knn::ensure(knn->row_bits, ebatch * (size_t)words) : nullptr;
      auto clear_row_bits = [&](size_t start, size_t end) {  // behind a batch: the words it set, back to zero
        int grid = (int)std::min<size_t>(((size_t)query_filter->nnz + 255) / 256, (size_t)ctx().num_cus * 8);
        coo_bitmap_clear_kernel<<<grid, 256, 0, stream()>>>(row_bits, words, (int)start, (int)end, (int)ni, query_filter->row.data(),
                                                            query_filter->col.data(), (size_t)query_filter->nnz);
        IMP_CHECK_HIP(hipGetLastError());
        knn->row_bits_dirty = false;
      };
      uint32_t *item_bits = have_items ? imp_knn::ensure(knn->item_bits, (size_t)words) : nullptr;
      if (have_items) IMP_CHECK_HIP(hipMemsetAsync(item_bits, 0, (size_t)words * 4, stream()));
      static_assert(kFlagSlots >= 2048, "one flag per row of an emit batch");
      const int *flags = host_flags;
      std::vector<int32_t> fb_list;
      constexpr bool no_qsplit = false;
      constexpr bool kCanSplit = BF3;
      const bool qsplit = kCanSplit && !no_qsplit;
      // fp16 two-term form with the queries resident in registers and the item planes cached (topk_resident.h): every factor count
      // that pads to 32 / 64 / 128 / 256; fp16-stored factors too (their values are their own high halves: scores stay bit-identical
      // to scoring fp32 copies of them).  IMP_TOPK_RESIDENT=0: the six-product 128 x 128 kernel of rounds 3-4 (A/B, parity)
      const int KS = rq_ks_for(f);
      const bool resident = kCanSplit && resident_env && !bf16x3 && KS > 0;
      split_bf16 *qs = nullptr;
      const _Float16 *iplanes = nullptr, *qplanes = nullptr;
      const int *iexp = nullptr, *qexp = nullptr;
      if (resident) {
        prepare_planes(KS, iplanes, iexp, qplanes, qexp);
      } else if (qsplit) {
        IMP_PROF("split_query_rows");
        const size_t nq_pad = (nq + 127) / 128 * 128;  // whole 128-row query blocks: a workgroup reads all four tiles of its block
        qs = imp_knn::ensure(knn->query_split, nq_pad * 3 * (size_t)f);
        const int grid = (int)std::max<size_t>(1, std::min<size_t>((nq_pad * (size_t)f + 255) / 256, (size_t)ctx().num_cus * 16));
        split_query_rows_kernel<TQ><<<grid, 256, 0, stream()>>>(Qb, reinterpret_cast<__bf16 *>(qs), nq, nq_pad, f);
        IMP_CHECK_HIP(hipGetLastError());
      }
      // the two GEMM launches of a batch: query rows pre-split (default) or in their storage type
      auto gemm = [&](auto mode_c, size_t start, dim3 grid, int rows, float *S_out, int bstride, const EmitArgs &ea) {
        constexpr int M = decltype(mode_c)::value;
        const float *norms_p = item_norms ? item_norms->f32() : nullptr;
        if constexpr (kCanSplit) {
          if (resident) {
            ResidentArgs ra{};
            ra.qsplit = qplanes + (start / 32) * (size_t)KS * 2 * 512;
            ra.isplit = iplanes, ra.qexp = qexp + start, ra.iexp = iexp;
            ra.nq = rows, ra.ni = (int)ni, ra.norms = norms_p;
            ra.n_blocks = (int)grid.x, ra.block_stride = bstride;
            ra.S = S_out, ra.sub_cols = (int)grid.x * 128, ra.emit = ea;
            ra.qa = q_err_a ? q_err_a + start : nullptr, ra.qb = q_err_b ? q_err_b + start : nullptr, ra.ine = item_ne, ra.tile_n = item_tile_n;
            launch_score_resident<M>(KS, ra, rows);
            return;
          }
        }
        if constexpr (M == 3) {
          throw std::logic_error("the screened emit pass exists in the resident form only");
        } else {
          if constexpr (kCanSplit) {
            if (qsplit) {
              score_gemm_direct_kernel<M, split_bf16, TI, true><<<grid, 256, 0, stream()>>>(qs + start * 3 * (size_t)f, rows, Ib, (int)ni, f, norms_p,
                                                                                      S_out, nullptr, 0, bstride, ea);
              return;
            }
          }
          score_gemm_direct_kernel<M, TQ, TI, BF3><<<grid, 256, 0, stream()>>>(Qb + start * f, rows, Ib, (int)ni, f, norms_p, S_out, nullptr, 0,
                                                                              bstride, ea);
        }
      };
      // screened emit pass (topk_resident.h MODE 3): one-product scores against tau - eps, the few candidates that can still be among
      // the best k re-scored in fp32 by the select kernel.  Dot-product scores only (no item norms); IMP_TOPK_SCREEN=0: three products
      static const bool screen_env = !(getenv("IMP_TOPK_SCREEN") && atoi(getenv("IMP_TOPK_SCREEN")) == 0);
      const bool screen = resident && screen_env && !item_norms;
      float *row_eps = screen ? imp_knn::ensure(knn->row_eps, rq_query_pad(ebatch)) : nullptr;
      for (size_t start = 0; start < nq; start += ebatch) {
        const size_t end = std::min(nq, start + ebatch), rows = end - start;
        const auto *qptr = Qb + start * f;
        const unsigned qblocks = (unsigned)((rows + 127) / 128);
        if (have_coo) {
          if (knn->row_bits_dirty) IMP_CHECK_HIP(hipMemsetAsync(row_bits, 0, knn->row_bits.size * sizeof(uint32_t), stream()));
          knn->row_bits_dirty = true;  // (until this batch's clear is queued)
        }
        {
          IMP_PROF("score_gemm_subset");
          gemm(std::integral_constant<int, 1>{}, start, dim3((unsigned)n_sub, qblocks), (int)rows, sub, stride, EmitArgs{});
          IMP_CHECK_HIP(hipGetLastError());
        }
        if (have_items) {
          IMP_PROF("item_filter");
          int grid = (int)std::min<size_t>((item_filter->size + 255) / 256, (size_t)ctx().num_cus * 8);
          item_bitmap_kernel<<<grid, 256, 0, stream()>>>(item_bits, (int)ni, item_filter->v.data(), (int)item_filter->size, sub, (int)rows,
                                                         sub_cols, stride);
          IMP_CHECK_HIP(hipGetLastError());
        }
        if (have_coo) {
          IMP_PROF("coo_filter");
          int grid = (int)std::min<size_t>(((size_t)query_filter->nnz + 255) / 256, (size_t)ctx().num_cus * 8);
          coo_bitmap_kernel<<<grid, 256, 0, stream()>>>(row_bits, words, (int)start, (int)end, (int)ni, query_filter->row.data(),
                                                        query_filter->col.data(), (size_t)query_filter->nnz, sub, sub_cols, stride);
          IMP_CHECK_HIP(hipGetLastError());
        }
        {
          IMP_PROF("topk_threshold");
          if (k_eff <= 32)
            subset_threshold_groupmax_kernel<256><<<(unsigned)rows, 256, 0, stream()>>>(sub, sub_cols, k_eff, tau, cnt);
          else
            subset_threshold_kernel<512><<<(unsigned)rows, 512, 0, stream()>>>(sub, sub_cols, k_eff, tau, cnt);
          IMP_CHECK_HIP(hipGetLastError());
        }
        {
          IMP_PROF("score_gemm");
          EmitArgs ea{tau, row_bits, item_bits, words, cand, cnt, kEmitCap, resident ? row_unscale : nullptr, row_eps};
          if (screen) gemm(std::integral_constant<int, 3>{}, start, dim3((unsigned)n_blocks, qblocks), (int)rows, nullptr, 1, ea);
          else gemm(std::integral_constant<int, 2>{}, start, dim3((unsigned)n_blocks, qblocks), (int)rows, nullptr, 1, ea);
          IMP_CHECK_HIP(hipGetLastError());
        }
        {
          IMP_PROF("topk_select_candidates");
          if (screen)
            select_screened_kernel<512, TQ, TI><<<(unsigned)rows, 512, 0, stream()>>>(cand, cnt, kEmitCap, k_eff, d_ids + start * k, d_dist + start * k, k,
                                                                                     fallback_e, row_eps, Qb + start * f, Ib, f, host_counts);
          else
            select_candidates_kernel<512><<<(unsigned)rows, 512, 0, stream()>>>(cand, cnt, kEmitCap, k_eff, d_ids + start * k,
                                                                               d_dist + start * k, k, fallback_e, resident ? row_unscale : nullptr);
          IMP_CHECK_HIP(hipGetLastError());
#ifdef RQ_SCREEN_STATS
          if (screen) {
            unsigned long long h[4];
            IMP_CHECK_HIP(hipStreamSynchronize(stream()));
            IMP_CHECK_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(rq_screen_stats), sizeof(h)));
            fprintf(stderr, "[screen-stats] rows %llu, candidates per row %.1f, re-scored per row %.1f (largest so far: %llu)\n", h[0],
                    (double)h[1] / std::max(1ull, h[0]), (double)h[2] / std::max(1ull, h[0]), h[3]);
          }
#endif
        }
        sync();
        fb_list.clear();
        for (size_t i = 0; i < rows; ++i)
          if (flags[i]) fb_list.push_back((int32_t)i);
        if (screen && fb_list.size() * 8 > rows) {
          // The screen separated nothing for much of this batch: scores so concentrated that an 11-bit product cannot tell the
          // best k from the bulk (factors a sweep or two from an all-positive start: every score within 1e-3 of the next) -- the
          // lists overflowed.  The WHOLE batch is redone by the three-product emit pass with its exact threshold test (same
          // thresholds, lists reset) before anything goes to the row-by-row exact path: 0.25 ms per 1000 rows instead of 1.3.
          // Decided by this batch's own outcome, not by the handle's history: the same call gives the same bits every time.
          IMP_PROF("topk_exact_emit_retry");
          IMP_CHECK_HIP(hipMemsetAsync(cnt, 0, rows * sizeof(unsigned int), stream()));
          EmitArgs ea{tau, row_bits, item_bits, words, cand, cnt, kEmitCap, row_unscale, nullptr};
          gemm(std::integral_constant<int, 2>{}, start, dim3((unsigned)n_blocks, qblocks), (int)rows, nullptr, 1, ea);
          select_candidates_kernel<512><<<(unsigned)rows, 512, 0, stream()>>>(cand, cnt, kEmitCap, k_eff, d_ids + start * k, d_dist + start * k, k,
                                                                             fallback_e, row_unscale);
          IMP_CHECK_HIP(hipGetLastError());
          sync();
          fb_list.clear();
          for (size_t i = 0; i < rows; ++i)
            if (flags[i]) fb_list.push_back((int32_t)i);
        }
        if (screen && rows >= 64) {  // the stride rule (above): mean list length and exact-path rows of this batch
          size_t total = 0;
          for (size_t i = 0; i < rows; ++i) total += (size_t)host_counts[i];
          const size_t mean = total / rows;
          if (fb_list.size() * 100 > rows || mean > 400) knn->stride_boost = std::max(1, knn->stride_boost / 2);
          else if (mean < 48 && knn->stride_boost < 4) knn->stride_boost *= 2;
        }
        static const bool debug = getenv("IMP_TOPK_DEBUG") != nullptr;
        if (debug && !fb_list.empty()) {
          std::vector<unsigned int> hc(rows);
          std::vector<uint32_t> ht(rows);
          IMP_CHECK_HIP(hipMemcpy(hc.data(), cnt, rows * 4, hipMemcpyDeviceToHost));
          IMP_CHECK_HIP(hipMemcpy(ht.data(), tau, rows * 4, hipMemcpyDeviceToHost));
          fprintf(stderr, "[topk-debug] batch at %zu: %zu fallback rows:", start, fb_list.size());
          for (size_t i = 0; i < std::min<size_t>(fb_list.size(), 8); ++i)
            fprintf(stderr, " (row %d count %u tau-key %08x)", fb_list[i], hc[fb_list[i]], ht[fb_list[i]]);
          fprintf(stderr, "\n");
        }
        if (!fb_list.empty()) {  // overflow / short list / exact tie at the k-th score: the materialising path, FB rows at a time
          IMP_PROF("topk_fallback");
          int32_t *d_rows = imp_knn::ensure(knn->fb_rows, fb_list.size());
          IMP_CHECK_HIP(hipMemcpyAsync(d_rows, fb_list.data(), fb_list.size() * 4, hipMemcpyHostToDevice, stream()));
          float *fbq = imp_knn::ensure(knn->fb_query, (size_t)FB * f);
          float *fscores = imp_knn::ensure(knn->scores, (size_t)FB * ni);
          float *ftile = imp_knn::ensure(knn->tile_max, (size_t)FB * n_tiles);
          int32_t *fids = imp_knn::ensure(knn->fb_ids, (size_t)FB * k);
          float *fdist = imp_knn::ensure(knn->fb_dist, (size_t)FB * k);
          uint64_t *fg = use_lds ? nullptr : imp_knn::ensure(knn->gcand, (size_t)FB * kpad);
          const size_t lds = use_lds ? (size_t)kpad * 8 : 0;
          auto kern = select_kernel<512>;
          IMP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)std::max<size_t>(lds, 1)));
          for (size_t g0 = 0; g0 < fb_list.size(); g0 += FB) {
            const int n = (int)std::min<size_t>(FB, fb_list.size() - g0);
            gather_query_rows_kernel<TQ><<<std::max(1, (n * f + 255) / 256), 256, 0, stream()>>>(qptr, d_rows + g0, n, f, fbq);
            score_gemm_direct_kernel<0, float, TI, BF3><<<dim3((unsigned)n_blocks, 1), 256, 0, stream()>>>(fbq, n, Ib, (int)ni, f, norms, fscores,
                                                                                         ftile, n_tiles, 1, EmitArgs{});
            if (have_coo || have_items) {
              int grid = (int)std::min<size_t>(((size_t)n * n_tiles + 255) / 256, (size_t)ctx().num_cus * 8);
              bitmap_filter_kernel<<<grid, 256, 0, stream()>>>(fscores, ftile, (int)ni, n_tiles, d_rows + g0, n, row_bits, item_bits, words);
            }
            kern<<<(unsigned)n, 512, lds, stream()>>>(fscores, (int)ni, k_eff, kpad, fids, fdist, k, fg, use_lds ? 1 : 0, nullptr);
            scatter_topk_rows_kernel<<<std::max(1, (n * k + 255) / 256), 256, 0, stream()>>>(fids, fdist, d_rows + g0, n, k,
                                                                                         d_ids + start * k, d_dist + start * k);
            IMP_CHECK_HIP(hipGetLastError());
          }
        }
        if (have_coo && end < nq) clear_row_bits(start, end);  // (the last batch's: behind the call's wait, below)
      }
      deliver();
      if (have_coo && nq > 0) clear_row_bits((nq - 1) / ebatch * ebatch, nq);
      return;
    }

    const int KS0 = rq_ks_for(f);
    const bool resident0 = BF3 && fast && resident_env && !bf16x3 && KS0 > 0;
    const _Float16 *iplanes0 = nullptr, *qplanes0 = nullptr;
    const int *iexp0 = nullptr, *qexp0 = nullptr;
    if (resident0) prepare_planes(KS0, iplanes0, iexp0, qplanes0, qexp0);
    for (size_t start = 0; start < nq; start += batch) {
      size_t end = std::min(nq, start + batch), rows = end - start;
      bool filters_applied = false;
      if (resident0) {
        IMP_PROF("score_gemm");
        ResidentArgs ra{};
        ra.qsplit = qplanes0 + (start / 32) * (size_t)KS0 * 2 * 512;
        ra.isplit = iplanes0, ra.qexp = qexp0 + start, ra.iexp = iexp0;
        ra.nq = (int)rows, ra.ni = (int)ni, ra.norms = item_norms ? item_norms->f32() : nullptr;
        ra.n_blocks = (int)((ni + 127) / 128), ra.block_stride = 1;
        ra.S = scores, ra.tile_max = tile_max, ra.n_tiles64 = n_tiles;
        launch_score_resident<0>(KS0, ra, (int)rows);
      } else if (fast) {
        IMP_PROF("score_gemm");
        dim3 grid((unsigned)((ni + 127) / 128), (unsigned)((rows + 127) / 128));
        score_gemm_direct_kernel<0, TQ, TI, BF3><<<grid, 256, 0, stream()>>>(Qb + start * f, (int)rows, Ib, (int)ni, f,
                                                                item_norms ? item_norms->f32() : nullptr, scores, tile_max,
                                                                n_tiles, 1, EmitArgs{});
        IMP_CHECK_HIP(hipGetLastError());
      } else {
        IMP_PROF("score_gemm_lds");
        dim3 grid((unsigned)((ni + kBN - 1) / kBN), (unsigned)((rows + kBM - 1) / kBM));
        if constexpr (std::is_same<TQ, float>::value) {  // the general path always runs on fp32 (copies of fp16 factors)
          score_gemm_kernel<<<grid, 256, 0, stream()>>>(Qb + start * f, (int)rows, Ib, (int)ni, f,
                                                        item_norms ? item_norms->f32() : nullptr, scores);
          IMP_CHECK_HIP(hipGetLastError());
        }
      }
      if (item_filter && item_filter->size) {
        IMP_PROF("item_filter");
        size_t total = rows * item_filter->size;
        int grid = (int)std::min<size_t>((total + 255) / 256, (size_t)ctx().num_cus * 8);
        item_filter_kernel<<<grid, 256, 0, stream()>>>(scores, (int)rows, (int)ni, item_filter->v.data(), (int)item_filter->size);
        IMP_CHECK_HIP(hipGetLastError());
        filters_applied = true;
      }
      if (query_filter && query_filter->nnz) {
        IMP_PROF("coo_filter");
        int grid = (int)std::min<size_t>(((size_t)query_filter->nnz + 255) / 256, (size_t)ctx().num_cus * 8);
        coo_filter_kernel<<<grid, 256, 0, stream()>>>(scores, (int)start, (int)end, (int)ni, query_filter->row.data(),
                                                      query_filter->col.data(), (size_t)query_filter->nnz);
        IMP_CHECK_HIP(hipGetLastError());
        filters_applied = true;
      }
      if (fast && filters_applied) {
        // second phase (all filter writes are done): refresh the maxima of the touched tiles
        IMP_PROF("filter_tile_refresh");
        if (item_filter && item_filter->size) {
          size_t total = rows * item_filter->size;
          int grid = (int)std::min<size_t>((total + 255) / 256, (size_t)ctx().num_cus * 8);
          item_filter_refresh_kernel<<<grid, 256, 0, stream()>>>(scores, tile_max, (int)rows, (int)ni, n_tiles,
                                                                 item_filter->v.data(), (int)item_filter->size);
        }
        if (query_filter && query_filter->nnz) {
          int grid = (int)std::min<size_t>(((size_t)query_filter->nnz + 255) / 256, (size_t)ctx().num_cus * 8);
          coo_filter_refresh_kernel<<<grid, 256, 0, stream()>>>(scores, tile_max, (int)start, (int)end, (int)ni, n_tiles,
                                                                query_filter->row.data(), query_filter->col.data(),
                                                                (size_t)query_filter->nnz);
        }
        IMP_CHECK_HIP(hipGetLastError());
      }
      if (fast) {
        IMP_PROF("topk_select_pruned");
        select_pruned_kernel<512><<<(unsigned)rows, 512, 0, stream()>>>(scores, tile_max, (int)ni, n_tiles, k_eff, extra, counts,
                                                                       d_ids + start * k, d_dist + start * k, k, fallback);
        IMP_CHECK_HIP(hipGetLastError());
      }
      {
        IMP_PROF("topk_select");
        size_t lds = use_lds ? (size_t)kpad * 8 : 0;
        auto kern = select_kernel<512>;
        IMP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)std::max<size_t>(lds, 1)));
        kern<<<(unsigned)rows, 512, lds, stream()>>>(scores, (int)ni, k_eff, kpad, d_ids + start * k, d_dist + start * k, k, gcand,
                                                     use_lds ? 1 : 0, fast ? fallback : nullptr);
        IMP_CHECK_HIP(hipGetLastError());
      }
      static const bool debug_m = getenv("IMP_TOPK_DEBUG") != nullptr;
      if (debug_m && fast) {  // how many rows the pruned select handed to the exact select
        std::vector<int> hf(rows);
        IMP_CHECK_HIP(hipMemcpy(hf.data(), fallback, rows * sizeof(int), hipMemcpyDeviceToHost));
        size_t nfb = 0, first_fb = 0;
        for (size_t i = 0; i < rows; ++i)
          if (hf[i]) {
            if (!nfb) first_fb = i;
            ++nfb;
          }
        fprintf(stderr, "[topk-debug] materialising batch at %zu: %zu of %zu rows re-done by the exact select (first: row %zu)\n", start, nfb,
                rows, first_fb);
      }
    }
    deliver();
    };
    // IMP_TOPK_FP32_MFMA=1: the exact-fp32 MFMA form (v_mfma_f32_32x32x2_f32) instead of the split-bf16 one (A/B, parity)
    static const bool exact_mfma = getenv("IMP_TOPK_FP32_MFMA") != nullptr;
    const bool bf3 = fast && !exact_mfma && f % 16 == 0;
    if (half_direct) {
      const __half *qh = reinterpret_cast<const __half *>(query_in->data), *ih = reinterpret_cast<const __half *>(items_in->data);
      if (bf3) run(qh, ih, std::true_type{});
      else run(qh, ih, std::false_type{});
    } else if (bf3) {
      run(query->f32(), items->f32(), std::true_type{});
    } else {
      run(query->f32(), items->f32(), std::false_type{});
    }
  });
}

}  // extern "C"
