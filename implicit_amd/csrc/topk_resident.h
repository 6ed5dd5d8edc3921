// Scoring GEMM of the top-k path with the QUERY operand resident in registers (round 6).  Included by topk.hip (namespace imp).
//
// Replaces the 128 x 128 block-tile form of score_gemm_direct_kernel for the two-term fp16 product (l h, h l, h h on
// v_mfma_f32_32x32x16_f16; reference semantics: the fp32 GEMM of implicit/gpu/knn.cu:131-147 == implicit/cpu/topk.pyx:45-47).
// That kernel ran one barrier per 16 factors, four workgroups per CU in step, and its phases -- DMA, fragment reads + splits,
// MFMAs, epilogue -- added up instead of overlapping (profiles/r05_topk_h2_knockouts.txt: 0.089 ms of MFMAs inside a 0.30 ms
// launch; 18 280 workgroups of 8 k-steps each).  Here:
//   * K = f is small (<= 256) and the catalogue is long: a wavefront keeps its query rows' A fragments for ALL of K in
//     registers (2 tiles x 8 k-steps x 2 terms x 4 VGPRs = 128 at f = 128) and only the ITEM operand streams;
//   * the item matrix is split ONCE per catalogue version into fragment-ordered fp16 planes (item_split_kernel; cached in the
//     KnnQuery handle, invalidated through note_device_write): a 32-item tile is KS x 2 KB of contiguous memory, moved by LDS-DMA
//     as it lies and read back by lane-linear ds_read_b128 -- no split, no swizzle, no conversion in the loop;
//   * persistent workgroups (2 per CU, 4 wavefronts, 128 TQ queries) walk a contiguous range of item tiles through a ring of
//     LDS slots: ONE raw s_barrier per tile (48 MFMAs per wavefront at f = 128), the DMA of the next NSTAGE - 1 tiles in flight
//     across it (counted s_waitcnt vmcnt(N), never 0);
//   * the four query blocks that read an item range are dispatched to the same XCD (blockIdx % 8) next to each other, so a
//     range crosses the fabric once per XCD instead of once per query block.
// Per-ROW query scales (one batch-wide scale flushed every row far below the batch maximum to fp16 zero) and
// the EXACT item maximum (the cached split pass sees every value: no sampled maximum, no overflow case left).
#ifndef IMPLICIT_AMD_CSRC_TOPK_RESIDENT_H_
#define IMPLICIT_AMD_CSRC_TOPK_RESIDENT_H_

typedef _Float16 rq_f16x8 __attribute__((ext_vector_type(8)));
template <int N, int I = 0, typename Fn> __device__ __forceinline__ void rq_static_for(Fn &&fn) {
  if constexpr (I < N) {
    fn(std::integral_constant<int, I>{});
    rq_static_for<N, I + 1>(fn);
  }
}

// A 16-byte LDS read the compiler does not track, and the counted wait that goes with it.  hipcc waits for its own LDS reads
// with lgkmcnt(0) wherever the three-set fragment prefetch below needs one of them -- i.e. also for the reads it has JUST
// issued for two k-steps ahead: a full LDS round trip exposed in front of the matrix pipe three times per item tile
// (profiles/r06_topk_resident_knockouts.txt).  LDS operations of a wavefront return in order, so "all but the N newest" is exact.
template <int OFFSET> __device__ __forceinline__ void rq_lds_read16(rq_f16x8 &dst, unsigned lds_addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_addr), "n"(OFFSET));
}
template <int N> __device__ __forceinline__ void rq_lds_wait(rq_f16x8 &x, rq_f16x8 &y) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(x), "+v"(y) : "n"(N));
}
#ifndef RQ_AHEAD
#define RQ_AHEAD 2  // k-steps of item fragments in flight ahead of the products (emit pass); 1 .. 4 measure the same (r06 knock-outs file)
#endif
#ifdef RQ_CLOCK  // variant builds: shader cycles and 100 MHz wall ticks one wavefront of the emit pass spends in its item loop
__device__ unsigned long long rq_clock_dbg[4];
#endif
#ifndef RQ_KO
#define RQ_KO 0  // timing-only knock-outs (build variants, wrong results): 1 no MFMAs, 2 no item DMA in the loop, 4 no epilogue, 8 no barrier, 16 tests but no survivor work
#endif

// scale exponent that brings a magnitude (bits of a non-negative float) to [2^11, 2^12); clamped for zero / non-finite rows
__device__ __forceinline__ int rq_scale_exp(unsigned maxbits) { return max(-60, min(60, 11 - ((int)(maxbits >> 23) - 127))); }
__device__ __forceinline__ float rq_pow2(int k) { return __uint_as_float((unsigned)(max(-126, min(127, k)) + 127) << 23); }

// largest magnitude of a matrix (bits of a non-negative float, so unsigned order is value order) -> *out via atomicMax;
// *out is zeroed by the caller
template <typename T> __global__ __launch_bounds__(256) void rq_absmax_kernel(const T *__restrict__ v, size_t n, unsigned *__restrict__ out) {
  float m = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf((float)v[i]));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}
// exp_out[0] = the scale exponent of the item matrix, from its exact maximum
__global__ void rq_item_exp_kernel(const unsigned *__restrict__ maxbits, int *__restrict__ exp_out) { exp_out[0] = rq_scale_exp(maxbits[0]); }

// ne[0] / ne[1] = bits of max_i || y_i 2^e ||_2 and of max_i || y_i 2^e - its high plane ||_2 (non-negative floats: unsigned order);
// one wavefront per item row; zeroed by the caller.  Once per catalogue version, with the planes.
// tile_n[t] = bits of max || y 2^e || over the 32 items of tile t (zeroed by the caller): the emit pass starts a tile's accumulators
// at -(tau - c_q tile_n[t]) instead of using the catalogue-wide bound.
template <typename T>
__global__ __launch_bounds__(256) void rq_item_err_kernel(const T *__restrict__ I, size_t rows, int f, const int *__restrict__ exp_in,
                                                          unsigned *__restrict__ ne, unsigned *__restrict__ tile_n) {
  const float s = rq_pow2(exp_in[0]);
  const int lane = threadIdx.x & 63;
  float mn = 0.f, me = 0.f;
  for (size_t row = blockIdx.x * (size_t)(blockDim.x >> 6) + (threadIdx.x >> 6); row < rows; row += (size_t)gridDim.x * (blockDim.x >> 6)) {
    float n2 = 0.f, e2 = 0.f;
    for (int c = lane; c < f; c += 64) {
      const float x = (float)I[row * (size_t)f + c] * s;
      const float d = x - (float)(_Float16)x;
      n2 = fmaf(x, x, n2), e2 = fmaf(d, d, e2);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) n2 += __shfl_xor(n2, off, 64), e2 += __shfl_xor(e2, off, 64);
    mn = fmaxf(mn, n2), me = fmaxf(me, e2);
    if (lane == 0) {
      atomicMax(&tile_n[row >> 5], __float_as_uint(sqrtf(n2) * 1.0000002f));
      // ne[2]: the largest ratio || y - y_h || / || y || of any item (an all-zero row: 0) -- with it E_i <= ne[2] N_i for every item
      if (n2 > 0.f) atomicMax(&ne[2], __float_as_uint(sqrtf(e2) / sqrtf(n2) * 1.000001f));
    }
  }
  if (lane == 0) {
    atomicMax(&ne[0], __float_as_uint(sqrtf(mn)));
    atomicMax(&ne[1], __float_as_uint(sqrtf(me)));
  }
}

// Fragment order (A and B operand of v_mfma_f32_32x32x16_f16 alike): element (row, c) of a 32-row tile goes to
//   ((tile KS + c / 16) 2 + term) 512 + lane 8 + (c & 7),   lane = (row & 31) + 32 ((c >> 3) & 1)
// -- lane (r, kh) of k-step s holds factors 16 s + 8 kh .. + 7 of row r.  Rows / factors past the matrix are zero.
template <typename T>
__global__ __launch_bounds__(256) void rq_split_items_kernel(const T *__restrict__ I, _Float16 *__restrict__ out, size_t rows, size_t rows_pad,
                                                             int f, int KS, const int *__restrict__ exp_in) {
  const float s = rq_pow2(exp_in[0]);
  const int F = KS * 16;
  // one thread per 8 consecutive factors of a row: a 16-byte store per term, 32-byte (fp32) read
  const size_t n = rows_pad * (size_t)(F / 8);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t row = i / (F / 8);
    const int c0 = (int)(i - row * (F / 8)) * 8;
    rq_f16x8 h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = (row < rows && c0 + e < f) ? (float)I[row * (size_t)f + c0 + e] * s : 0.f;
      const _Float16 hi = (_Float16)x;
      h[e] = hi, l[e] = (_Float16)(x - (float)hi);
    }
    const int lane = (int)(row & 31) + 32 * ((c0 >> 3) & 1);
    _Float16 *o = out + ((((row >> 5) * KS + (c0 >> 4)) * 2) * 64 + lane) * 8;
    *reinterpret_cast<rq_f16x8 *>(o) = h;
    *reinterpret_cast<rq_f16x8 *>(o + 512) = l;
  }
}

// Query rows: ONE wavefront per row -- row maximum, its own scale exponent (qexp[row]), the two planes in fragment order.
// qa / qb (may be null): || q 2^e - high plane ||_2 and || high plane ||_2 of the row -- what the screened emit pass (MODE 3) bounds
// the error of its one-product scores with.
template <typename T>
__global__ __launch_bounds__(256) void rq_split_queries_kernel(const T *__restrict__ Q, _Float16 *__restrict__ out, int *__restrict__ qexp,
                                                               size_t rows, size_t rows_pad, int f, int KS, float *__restrict__ qa,
                                                               float *__restrict__ qb) {
  const int lane = threadIdx.x & 63;
  const int F = KS * 16;
  for (size_t row = blockIdx.x * (size_t)(blockDim.x >> 6) + (threadIdx.x >> 6); row < rows_pad; row += (size_t)gridDim.x * (blockDim.x >> 6)) {
    float m = 0.f;
    if (row < rows)
      for (int c = lane; c < f; c += 64) m = fmaxf(m, fabsf((float)Q[row * (size_t)f + c]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    const int e = rq_scale_exp(__float_as_uint(m));  // (a NaN / inf row: clamped; its scores come out NaN and the row goes to the exact path)
    if (lane == 0) qexp[row] = e;
    const float s = rq_pow2(e);
    float ea = 0.f, eb = 0.f;
    for (int c = lane; c < F; c += 64) {
      const float x = (row < rows && c < f) ? (float)Q[row * (size_t)f + c] * s : 0.f;
      const _Float16 hi = (_Float16)x;
      const int ln = (int)(row & 31) + 32 * ((c >> 3) & 1);
      _Float16 *o = out + ((((row >> 5) * KS + (c >> 4)) * 2) * 64 + ln) * 8 + (c & 7);
      o[0] = hi, o[512] = (_Float16)(x - (float)hi);
      const float d = x - (float)hi;
      ea = fmaf(d, d, ea), eb = fmaf((float)hi, (float)hi, eb);
    }
    if (qa) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) ea += __shfl_xor(ea, off, 64), eb += __shfl_xor(eb, off, 64);
      if (lane == 0) qa[row] = sqrtf(ea), qb[row] = sqrtf(eb);
    }
  }
}

// MODE 3: bound of |q.y - q_h.y_h| in the scaled domain (see the kernel's header), with slack for the fp32 accumulation of the
// products (exact themselves: 11 x 11 bits), for the rounding of the four norms and for the threshold's own arithmetic
__device__ __forceinline__ float rq_screen_eps(float qa, float qb, float N, float E) {
  return (qa * N + qb * E) * 1.00390625f + qb * N * 3.814697265625e-6f;  // (1 + 2^-8), 2^-18
}

// the same bound for ONE item of norm N_i: E_i <= rho N_i (rho = ne[2]) gives eps_i <= rq_screen_coef(qa, qb, rho) N_i
__device__ __forceinline__ float rq_screen_coef(float qa, float qb, float rho) {
  return (qa + qb * rho) * 1.00390625f + qb * 3.814697265625e-6f;
}

// a candidate that passed its threshold: filter bitmaps (the batch's item filter, the query's liked items), then the query's list
__device__ __forceinline__ void rq_append(const EmitArgs &e, int q, int item, unsigned long long key) {
  const uint32_t bit = 1u << (item & 31);
  bool filtered = e.item_bits && (e.item_bits[item >> 5] & bit);
  if (!filtered && e.row_bits) filtered = e.row_bits[(size_t)q * e.words + (item >> 5)] & bit;
  if (!filtered) {
    const unsigned int sl = atomicAdd(&e.count[q], 1u);
    if (sl < (unsigned)e.cap) e.cand[(size_t)q * e.cap + sl] = key;
  }
}

struct ResidentArgs {
  const _Float16 *qsplit;  // planes of the launch's first query row (a multiple of 128 TQ rows follows, zero-padded)
  const _Float16 *isplit;  // planes of item 0 (whole 128-item blocks, zero-padded)
  const int *qexp;         // per query row of the launch
  const int *iexp;         // device scalar
  int nq, ni;
  const float *norms;      // item norms (cosine scores) or null
  int n_blocks;            // 128-item blocks the launch walks (MODE 1: the subset's blocks, every block_stride-th of the catalogue)
  int block_stride;        // MODE 1
  int chunks, n_qb;        // grid = chunks x n_qb workgroups (chunks a multiple of 8)
  float *S;                // MODE 0: scores [nq][ni];  MODE 1: compact subset scores [nq][sub_cols]
  int sub_cols;
  float *tile_max;         // MODE 0: [nq][n_tiles64]
  int n_tiles64;
  EmitArgs emit;           // MODE 2 / 3
  const float *qa, *qb;    // MODE 3: per query row || q 2^e - high plane ||, || high plane ||
  const unsigned *ine;     // MODE 3: bits of max || y 2^e ||, max || y 2^e - high plane || over the catalogue, max ratio of the two per item
  const float *tile_n;     // MODE 3: max || y 2^e || per 32-item tile
};

// LDS slots of the item ring: as many as leave room for two workgroups per CU (the emit pass also stages its candidates in LDS)
template <int KS, int MODE> constexpr int rq_stages() {
  constexpr int NI = 1;
  constexpr int SLOT = NI * KS * 2048 + 4 * 256 * NI;
  constexpr int budget = MODE >= 2 ? 56 * 1024 : 72 * 1024;
  return 4 * SLOT <= budget ? 4 : (3 * SLOT <= budget ? 3 : 2);
}
constexpr int kRqStageCap = 1536;  // candidates a workgroup of the emit pass stages in LDS before its one flush

// MODE 0: scores + per-(query, 64-item) maxima (materialising path)   TQ = 1
// MODE 1: compact subset scores (threshold pre-pass)                  MODE 2: candidates >= tau appended (emit pass)
// MODE 3: the SCREENED emit pass (no item norms).  The emit pass only has to FIND the entries that may reach the threshold, so it
// scores with ONE product -- high plane x high plane, a third of the matrix work and half the LDS / DMA bytes -- and tests against
// tau - eps_q, eps_q = a_q N + b_q E a rigorous bound of |q.y - q_h.y_h| (Cauchy-Schwarz: a_q = ||q - q_h||, b_q = ||q_h||,
// N = max ||y||, E = max ||y - y_h||; + slack for the fp32 accumulation and for tau's own rounding).  Candidates carry the
// APPROXIMATE accumulator; select_screened_kernel re-scores the few that can still be among the best k from the stored factors
// in fp32 (entries within 2 eps_q of the k-th approximate score -- every entry whose exact score reaches the k-th exact score is
// among them) and orders those.  The matrix pipe was 84 % busy at a power-limited 1.67 GHz with three products
// (profiles/r06_topk_resident_knockouts.txt): fewer products is the lever that is left.
template <int KS, int TQ, int MODE>
__global__ __launch_bounds__(256, 2) void score_resident_kernel(ResidentArgs a) {
  constexpr int NI = 1;                                    // 32-item tiles per pipeline step
  static_assert(MODE != 0 || TQ == 1, "the materialising form keeps one query tile per wavefront (its 64-item maxima live across two steps)");
  constexpr int TILE_BYTES = KS * 2048;
  constexpr int STEP_BYTES = NI * TILE_BYTES;
  constexpr int PAD_BYTES = 4 * 256 * NI;                  // per wavefront: the norms of its step's items (256 B per tile)
  constexpr int SLOT = STEP_BYTES + PAD_BYTES;
  constexpr int NSTAGE = rq_stages<KS, MODE>();
  constexpr int PIECES = STEP_BYTES / 1024;                // 1 KB DMA instructions per step
  static_assert(PIECES % 4 == 0 || PIECES == 2, "pieces are dealt to the four wavefronts");
  // (KS = 1 is not instantiated; PIECES = 2 x KS x NI >= 4.)  MODE 3 stages the high-plane pieces only (the even ones); with fewer
  // of them than wavefronts (KS = 2) the spare wavefronts repeat a piece, so every wavefront issues the same number of loads
  constexpr int PER_WAVE_DATA = (MODE == 3) ? (PIECES / 2 >= 4 ? PIECES / 8 : 1) : (PIECES >= 4 ? PIECES / 4 : 1);
  constexpr int PER_WAVE = PER_WAVE_DATA + NI;             // + the norms piece(s)
  constexpr int QROWS = 128 * TQ;
  extern __shared__ __attribute__((aligned(1024))) unsigned char rq_smem[];
  unsigned char *ring = rq_smem;
  // objects of their own, NOT part of the ring: hipcc orders every LDS read that may alias a pending LDS-DMA behind vmcnt(0) --
  // reads of these arrays in the epilogue would otherwise drain the ring on every step
  __shared__ float tau_s[QROWS];    // threshold in the scaled domain (MODE 2, no norms) / as it is
  __shared__ float unscale[QROWS];  // 2^-(qexp + iexp)
  // MODE 2: a score that passes its threshold is STAGED in LDS (an LDS atomic: ~100 cycles) and everything that needs a global
  // round trip -- the filter bitmaps, the atomic on the query's list, the store -- happens once, at the end, for all staged
  // entries in parallel.  About stride x k entries per query survive the threshold, i.e. two per wavefront and step: inline, the
  // three dependent round trips (~4 us) behind every one of them were six times the step's matrix time (0.64 us).
  constexpr bool EMIT = MODE >= 2, ONE = MODE == 3;  // ONE: one product (high planes only)
  __shared__ unsigned long long st_key[EMIT ? kRqStageCap : 1];
  __shared__ unsigned short st_row[EMIT ? kRqStageCap : 1];
  __shared__ unsigned st_n, st_over;  // staged entries; sticky: a step staged more than the buffer had room for (entries were dropped)
  constexpr int kTileNormCap = 512;  // steps whose tile norms a workgroup keeps in LDS (beyond: the catalogue-wide maximum)
  __shared__ float tn_s[ONE ? kTileNormCap : 1];
  if (EMIT && threadIdx.x == 0) st_n = 0u, st_over = 0u;

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c31 = lane & 31, kh = lane >> 5;
  // workgroup -> (query block, item chunk): the n_qb workgroups of a chunk sit on ONE XCD (blockIdx % 8), next to each other
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int qb = j % a.n_qb, chunk = (j / a.n_qb) * 8 + xcd;
  // chunk c walks the 128-item blocks c, c + chunks, c + 2 chunks, ...: catalogues are often ordered by popularity, and the items
  // that pass the thresholds then sit in the first blocks -- a contiguous range would hand one workgroup most of the survivors
  // (measured on the bench's trained factors: its LDS staging overflowed and every row of its query block went to the exact path)
  // The emit pass deals single 32-item TILES (t = c, c + chunks, ...): what it stages before filtering includes the users' liked
  // items, and on such a catalogue a tenth of all interactions sit in the first two 128-item blocks.
  constexpr int GRAIN = EMIT ? 1 : 4;  // tiles per dealt unit
  const int n_units = a.n_blocks * (4 / GRAIN);
  const int my_units = chunk < n_units ? (n_units - chunk + a.chunks - 1) / a.chunks : 0;
  const int steps = my_units * GRAIN;
  if constexpr (EMIT) {
    // candidates travel with their RAW accumulator as the key's score (one scale per query row: the order is the scores'); the
    // select kernel scales the k winners out.  The first workgroup of a query block leaves the factors for it (even when its own
    // item range is empty: more chunks than item blocks)
    if (chunk == 0 && a.emit.row_unscale) {
      const int ie = a.iexp[0];
      for (int r = threadIdx.x; r < 128 * TQ; r += 256) {
        const int q = qb * 128 * TQ + r;
        a.emit.row_unscale[q] = a.norms ? 1.f : rq_pow2(-(a.qexp[q] + ie));
      }
    }
    if constexpr (ONE) {
      if (chunk == 0) {
        const float N = __uint_as_float(a.ine[0]), E = __uint_as_float(a.ine[1]);
        for (int r = threadIdx.x; r < 128 * TQ; r += 256) {
          const int q = qb * 128 * TQ + r;
          a.emit.row_eps[q] = rq_screen_eps(a.qa[q], a.qb[q], N, E);
        }
      }
    }
  }
  if (steps <= 0) return;
  const int q_wave = qb * QROWS + wave * 32 * TQ;          // first query row of this wavefront (launch-relative)
  const int last_tile = (a.ni + 31) / 32 - 1;

  // ---- prologue: scales and thresholds of the workgroup's rows, the resident query fragments -----------------------------
  const int iexp = a.iexp[0];
  if constexpr (ONE) {
    for (int s = threadIdx.x; s < min(steps, kTileNormCap); s += 256) tn_s[s] = a.tile_n[min(chunk + s * a.chunks, last_tile)];
  }
  for (int r = threadIdx.x; r < QROWS; r += 256) {
    const int q = qb * QROWS + r;
    const int e = a.qexp[q] + iexp;                        // (qexp is padded to whole query blocks)
    unscale[r] = rq_pow2(-e);
    if constexpr (EMIT) {
      const float t = q < a.nq ? unordered(a.emit.tau[q]) : INFINITY;
      // without norms the test runs on the raw accumulators: tau 2^e (exact unless it leaves the range, where it saturates the
      // safe way: -inf / the largest finite value let more through, never less)
      // without norms the test runs on the raw accumulators against tau 2^e.  The product is exact while it stays a normal
      // number; outside that range the threshold moves the SAFE way (more candidates, never fewer)
      float ts = t;
      if (!a.norms) {
        ts = t * rq_pow2(e);
        if (t > 0.f) {
          if (!(ts <= FLT_MAX)) ts = FLT_MAX;      // overflow: the largest finite threshold
          if (ts < FLT_MIN) ts = 0.f;              // underflow (rounding in the subnormal range could round UP)
        } else if (t < 0.f) {
          if (!(ts >= -FLT_MAX)) ts = -INFINITY;   // overflow: everything passes
          if (ts > -FLT_MIN) ts = -FLT_MIN;
        }
        if (q >= a.nq) ts = INFINITY;
      }
      // screened pass: everything whose exact score can reach tau (a NaN / inf bound lets everything through: the row overflows
      // and goes to the exact path)
      // screened pass: a tile's accumulators start at -(tau - c_q N_t), N_t = the tile's largest item norm (a NaN / inf coefficient
      // lets everything through: the row overflows and goes to the exact path); c_q rides in `unscale` (not needed without norms)
      if constexpr (ONE) unscale[r] = q < a.nq ? rq_screen_coef(a.qa[q], a.qb[q], __uint_as_float(a.ine[2])) : 0.f;
      tau_s[r] = ts;
    }
  }
  rq_f16x8 ah[TQ][KS], al[ONE ? 1 : TQ][ONE ? 1 : KS];
#pragma unroll
  for (int tq = 0; tq < TQ; ++tq) {
    const _Float16 *src = a.qsplit + ((size_t)(q_wave / 32 + tq) * KS * 2) * 512 + lane * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      ah[tq][ks] = *reinterpret_cast<const rq_f16x8 *>(src + (size_t)(ks * 2) * 512);
      if constexpr (!ONE) al[tq][ks] = *reinterpret_cast<const rq_f16x8 *>(src + (size_t)(ks * 2 + 1) * 512);
    }
  }

  // The fragments must have ARRIVED before the first LDS-DMA is issued: hipcc waits for an ordinary load at its first use with
  // vmcnt(0) -- inside the loop that would drain the DMA ring on every step (cdna_hip_programming.md, "pipelining across
  // barriers").  Passing them through an opaque copy puts that one wait here.
#pragma unroll
  for (int tq = 0; tq < TQ; ++tq)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      asm volatile("" : "+v"(ah[tq][ks]));
      if constexpr (!ONE) asm volatile("" : "+v"(al[tq][ks]));
    }
  __syncthreads();  // tau_s / unscale are in place (nothing LDS-bound is in flight yet: a plain barrier)
  // MODE 3: the accumulators START at -(tau - eps) of their rows (the first MFMA's C operand), so the threshold test is a maximum
  // and a sign -- no subtraction, no LDS read per step.  The three-product pass cannot do this (a survivor's score rebuilt as
  // d + tau is an ulp or two off the threshold pass's value of the same dot product: rows then find k - 1 candidates); here the
  // candidates are re-scored anyway, the keys only have to keep each row's ORDER (an offset per row does), and the rounding of the
  // offset sum (2^-23 of |tau|) is inside the bound's slack term
  [[maybe_unused]] float nts[ONE ? TQ : 1][16], cqr[ONE ? TQ : 1][16];
  [[maybe_unused]] float norm_max = 0.f;
  if constexpr (ONE) {
    norm_max = __uint_as_float(a.ine[0]);
    asm volatile("" : "+v"(norm_max));
#pragma unroll
    for (int tq = 0; tq < TQ; ++tq)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int r = wave * 32 * TQ + 32 * tq + 4 * kh + (e & 3) + 8 * (e >> 2);
        nts[tq][e] = -tau_s[r], cqr[tq][e] = unscale[r];
      }
  }

  // ---- item stream -----------------------------------------------------------------------------------------------------
  // step s of the workgroup = item tiles tile_of(s) .. + NI - 1; past the end the last tile is staged again (results unused)
  auto block_of = [&](int s) { return chunk + (s >> 2) * a.chunks; };  // (MODE 0 / 1)
  auto first_tile = [&](int s) {
    if constexpr (EMIT) return chunk + s * a.chunks;
    const int blk = block_of(s), sub = s & 3;
    return (MODE == 1 ? blk * a.block_stride : blk) * 4 + sub;
  };
  const unsigned char *ibase = reinterpret_cast<const unsigned char *>(a.isplit);
  auto dma = [&](int s) {
    unsigned char *slot = ring + (s % NSTAGE) * SLOT;
    const int t0 = first_tile(min(s, steps - 1));
#pragma unroll
    for (int p = 0; p < PER_WAVE_DATA; ++p) {
      int piece = wave * PER_WAVE_DATA + p;                // 0 .. PIECES - 1 over the NI tiles of the step
      if constexpr (MODE == 3) piece = 2 * (piece % (PIECES / 2));  // high planes: pieces 0, 2, 4, ...
      const int tile = min(t0 + piece / (TILE_BYTES / 1024), last_tile), within = piece % (TILE_BYTES / 1024);
      const unsigned char *src = ibase + (size_t)tile * TILE_BYTES + within * 1024 + lane * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                       (__attribute__((address_space(3))) void *)(slot + piece * 1024), 16, 0, 0);
    }
    // the norms of the step's items, into this wavefront's own pad (lane c of tile n: item 32 (t0 + n) + c; lanes 32-63 repeat)
#pragma unroll
    for (int n = 0; n < NI; ++n) {
      const int item = min(32 * min(t0 + n, last_tile) + c31, a.ni - 1);
      const float *src = a.norms ? a.norms + item : reinterpret_cast<const float *>(ibase) + lane;  // (no norms: any valid address)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                       (__attribute__((address_space(3))) void *)(slot + STEP_BYTES + (wave * NI + n) * 256), 4, 0, 0);
    }
  };
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s) dma(s);

  if constexpr (EMIT) {
    // ---- emit pass.  d = acc - tau_s (the row thresholds in the scaled domain, four rows per LDS read), "does anything in this lane
    // pass" five v_max3 and one compare per query tile, and the per-element work (which element, the exact test on the ordered key, the LDS staging) is left to the
    // lanes that have a survivor.  A compare per accumulator element against thresholds read from LDS cost ~1.4 K vector cycles per
    // step beside 1.5 K matrix cycles, and the two did not overlap (profiles/r06_topk_resident_knockouts.txt).  (Starting the
    // accumulators AT -tau -- the first MFMA's C operand -- would save the adds, but a survivor's score would then be rebuilt as
    // d + tau, an ulp or two off the threshold pass's value of the same dot product: a row whose k best all sit in the sampled
    // subset then finds k - 1 candidates -- measured: 6-13 rows per 1000 sent to the exact path.)  With item norms (cosine
    // scores) the finished score is what has to be compared: that form converts first and tests second.
#ifdef RQ_CLOCK
    const long long rq_c0 = __builtin_readcyclecounter(), rq_w0 = wall_clock64();
#endif
    // candidates staged so far go to the lists EARLY when the staging buffer is nearly full: read right behind a step's barrier
    // the count is the same in every wavefront (all of them are past the previous step's test), so the decision is uniform.
    // Popularity-dominated scores (the same few hundred items best for every query) put 256 survivors per such item into ONE
    // workgroup: without this its buffer overflowed and every row of its query block went to the exact path
    // room a single step may need: three quarters of the buffer (an item that passes for every row of the block stages 256 entries
    // by itself; a step that needs more drops entries -- remembered in st_over, and the rule at the end sends the block's rows to
    // the exact path.  A fuzz run found the first form of this, which forgot the drop at the flush: profiles/scripts/r6F_fuzz.py)
    constexpr unsigned kStageHead = kRqStageCap - kRqStageCap / 4;
    auto flush_staged = [&]() {
      const unsigned n_st = min(st_n, (unsigned)kRqStageCap);
      for (unsigned i = threadIdx.x; i < n_st; i += 256) {
        const unsigned long long key = st_key[i];
        rq_append(a.emit, qb * QROWS + (int)st_row[i], (int)(uint32_t)key, key);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the appends' memory operations share the counter of the counted DMA waits)
      __syncthreads();
      if (threadIdx.x == 0) {
        if (st_n > (unsigned)kRqStageCap) st_over = 1u;  // (a single step outran the headroom: remembered for the rule at the end)
        st_n = 0u;
      }
      __syncthreads();
    };
    [[maybe_unused]] float tile_norm_next = ONE ? tn_s[0] : 0.f;  // (norm_max: loaded in the prologue -- a global load in this loop would be waited for with vmcnt(0), draining the ring)
    for (int s = 0; s < steps; ++s) {
      if (!(RQ_KO & 2)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * PER_WAVE) : "memory");
      // (the raw barrier does not wait for LDS operations: a wavefront's staging stores of the previous step's test must have
      // landed before another wavefront may flush them -- none of its fragment reads is in flight at this point, so the wait is free)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (!(RQ_KO & 8)) __builtin_amdgcn_s_barrier();
      if (st_n > (unsigned)kRqStageCap - kStageHead) flush_staged();  // (uniform; rare)
      if (!(RQ_KO & 2)) dma(s + NSTAGE - 1);
      const unsigned char *slot = ring + (s % NSTAGE) * SLOT;
      // (read one step ahead: the accumulators' start values -- and with them the first product -- would otherwise wait for an LDS
      // round trip at the top of every step)
      [[maybe_unused]] const float tile_norm = tile_norm_next;
      if constexpr (ONE) tile_norm_next = s + 1 < kTileNormCap ? tn_s[s + 1] : norm_max;
      f32x16 acc[TQ];
#pragma unroll
      for (int tq = 0; tq < TQ; ++tq)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[tq][e] = ONE ? fmaf(cqr[tq][e], tile_norm, nts[tq][e]) : 0.f;
      // item fragments TWO k-steps ahead of the MFMAs that use them (three register sets): with the reads of k-step ks + 1 issued
      // only after the MFMAs of ks, an LDS round trip under load (eight wavefronts reading, the DMA writing) outlasted the 192
      // cycles those MFMAs cover, and the matrix pipe idled a third of the loop (SQ_VALU_MFMA_BUSY: 0.50 of the launch)
      constexpr int AHEAD = RQ_AHEAD, SETS = AHEAD + 1;  // fragment sets: k-steps in flight ahead of the products
      rq_f16x8 bh[SETS], bl[SETS];
      const unsigned slot_lds = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char *)(slot + lane * 16);
      auto ldb = [&](auto ks_c, auto set_c) {
        constexpr int ks = decltype(ks_c)::value, set = decltype(set_c)::value;
        if (RQ_KO & 32) {  // (timing only: no fragment reads)
          asm volatile("" : "=v"(bh[set]), "=v"(bl[set]));
          return;
        }
        rq_lds_read16<(ks * 2) * 1024>(bh[set], slot_lds);
        if constexpr (ONE) asm volatile("" : "=v"(bl[set]));  // (never read)
        else rq_lds_read16<(ks * 2 + 1) * 1024>(bl[set], slot_lds);
      };
      using std::integral_constant;
      rq_static_for<(AHEAD < KS ? AHEAD : KS)>([&](auto Pc) {
        constexpr int p = decltype(Pc)::value;
        ldb(integral_constant<int, p>{}, integral_constant<int, p % SETS>{});
      });
      __builtin_amdgcn_sched_barrier(0);
      rq_static_for<KS>([&](auto Kc) {
        constexpr int ks = decltype(Kc)::value;
        if constexpr (ks + AHEAD < KS) ldb(integral_constant<int, ks + AHEAD>{}, integral_constant<int, (ks + AHEAD) % SETS>{});
        // the reads of k-step ks have returned when at most those of the (up to AHEAD) later k-steps are outstanding
        rq_lds_wait<(ONE ? 1 : 2) * (KS - 1 - ks < AHEAD ? KS - 1 - ks : AHEAD)>(bh[ks % SETS], bl[ks % SETS]);
        __builtin_amdgcn_sched_barrier(0);  // (left alone, the scheduler sinks every read to just before its first use)
        if (RQ_KO & 1) {
#pragma unroll
          for (int tq = 0; tq < TQ; ++tq) asm volatile("" ::"v"(bh[ks % SETS]), "v"(bl[ks % SETS]), "v"(ah[tq][ks]));
        } else {  // the query tiles' chains in turn: a dependent MFMA directly behind its predecessor waits for the result
          if constexpr (!ONE) {
#pragma unroll
            for (int tq = 0; tq < TQ; ++tq) acc[tq] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[tq][ks], bh[ks % SETS], acc[tq], 0, 0, 0);
#pragma unroll
            for (int tq = 0; tq < TQ; ++tq) acc[tq] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tq][ks], bl[ks % SETS], acc[tq], 0, 0, 0);
          }
#pragma unroll
          for (int tq = 0; tq < TQ; ++tq) acc[tq] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tq][ks], bh[ks % SETS], acc[tq], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      if (RQ_KO & 4) {
        float sink = 0.f;
#pragma unroll
        for (int tq = 0; tq < TQ; ++tq)
#pragma unroll
          for (int e = 0; e < 16; ++e) sink += acc[tq][e];
        if (sink == 1.2345e-33f) a.S[0] = sink;
        continue;
      }
      const int item = 32 * first_tile(s) + c31;
      const float nrm = a.norms ? *reinterpret_cast<const float *>(slot + STEP_BYTES + wave * 256 + c31 * 4) : 1.f;
#pragma unroll
      for (int tq = 0; tq < TQ; ++tq) {
        f32x16 &c = acc[tq];
        const int r_base = wave * 32 * TQ + 32 * tq + 4 * kh;
        if (a.norms) {  // (uniform) cosine scores: scale out, divide; negts holds -tau as it is
#pragma unroll
          for (int eg = 0; eg < 4; ++eg) {
            const float4 us = *reinterpret_cast<const float4 *>(unscale + r_base + 8 * eg);
            c[4 * eg] = c[4 * eg] * us.x / nrm, c[4 * eg + 1] = c[4 * eg + 1] * us.y / nrm;
            c[4 * eg + 2] = c[4 * eg + 2] * us.z / nrm, c[4 * eg + 3] = c[4 * eg + 3] * us.w / nrm;
          }
        }
        // d = score - threshold: its sign is exact.  A NaN (a non-finite operand) is dropped by the maxima while a number stands
        // beside it: its row then collects fewer than k candidates and goes to the exact path, as it should
        f32x16 d;
        if constexpr (ONE) {
          d = c;  // (the accumulators started at minus their thresholds)
        } else {
#pragma unroll
          for (int eg = 0; eg < 4; ++eg) {
            const float4 ts = *reinterpret_cast<const float4 *>(tau_s + r_base + 8 * eg);
            d[4 * eg] = c[4 * eg] - ts.x, d[4 * eg + 1] = c[4 * eg + 1] - ts.y, d[4 * eg + 2] = c[4 * eg + 2] - ts.z, d[4 * eg + 3] = c[4 * eg + 3] - ts.w;
          }
        }
        const float m = fmaxf(fmaxf(fmaxf(fmaxf(d[0], d[1]), fmaxf(d[2], d[3])), fmaxf(fmaxf(d[4], d[5]), fmaxf(d[6], d[7]))),
                              fmaxf(fmaxf(fmaxf(d[8], d[9]), fmaxf(d[10], d[11])), fmaxf(fmaxf(d[12], d[13]), fmaxf(d[14], d[15]))));
        const bool hit = !(m < 0.f);
        if (__builtin_amdgcn_ballot_w64(hit) == 0) continue;
        // About two scores per wavefront and step pass.  d >= 0 IS the exact test (a power-of-two scale moves no comparison), so a
        // survivor needs no second look: its raw accumulator is staged as it is (a NaN as +inf: its row goes to the exact path)
        if (!(RQ_KO & 19)) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            if (!(d[e] < 0.f)) {
              const int r = r_base + (e & 3) + 8 * (e >> 2), q = qb * QROWS + r;
              float sc = c[e];
              if (!(sc == sc)) sc = INFINITY;
              if (q < a.nq && item < a.ni) {
                const unsigned at = atomicAdd(&st_n, 1u);  // LDS
                if (at < (unsigned)kRqStageCap) st_key[at] = make_key(sc, item), st_row[at] = (unsigned short)r;
              }  // (at >= capacity: dealt with once, at the flush)
            }
          }
        }
      }
    }
#ifdef RQ_CLOCK
    if (blockIdx.x == 17 && threadIdx.x == 0) {
      rq_clock_dbg[0] = (unsigned long long)(__builtin_readcyclecounter() - rq_c0), rq_clock_dbg[1] = (unsigned long long)(wall_clock64() - rq_w0);
      rq_clock_dbg[2] = (unsigned long long)steps;
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stagings past the end
    __syncthreads();
    const unsigned n_all = st_n, n_st = min(n_all, (unsigned)kRqStageCap);
    for (unsigned i = threadIdx.x; i < n_st; i += 256) {
      const unsigned long long key = st_key[i];
      rq_append(a.emit, qb * QROWS + (int)st_row[i], (int)(uint32_t)key, key);
    }
    if (n_all > (unsigned)kRqStageCap || st_over) {
      // the staging buffer overflowed (a workgroup expects ~640 of its 1536 entries): candidates were dropped, WHOSE is not known --
      // every row of the query block is declared overflowed and re-done by the exact path (slow, correct, and not seen so far)
      for (int r = threadIdx.x; r < QROWS; r += 256)
        if (qb * QROWS + r < a.nq) atomicAdd(&a.emit.count[qb * QROWS + r], (unsigned)a.emit.cap + 1u);
    }
    return;
  }

  float tmax[MODE == 0 ? 16 : 1];
  for (int s = 0; s < steps; ++s) {
    // this wavefront's pieces of step s have landed (the NSTAGE - 2 younger steps may still fly) ... and everybody else's; all
    // wavefronts are also done with the slot of step s - 1, which the DMA of step s + NSTAGE - 1 overwrites
    if (!(RQ_KO & 2)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * PER_WAVE) : "memory");
    if (!(RQ_KO & 8)) __builtin_amdgcn_s_barrier();
    if (!(RQ_KO & 2)) dma(s + NSTAGE - 1);
    const unsigned char *slot = ring + (s % NSTAGE) * SLOT;
    f32x16 acc[TQ][NI];
#pragma unroll
    for (int tq = 0; tq < TQ; ++tq)
#pragma unroll
      for (int n = 0; n < NI; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[tq][n][e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int n = 0; n < NI; ++n) {
        const rq_f16x8 bh = *reinterpret_cast<const rq_f16x8 *>(slot + n * TILE_BYTES + (ks * 2) * 1024 + lane * 16);
        const rq_f16x8 bl = *reinterpret_cast<const rq_f16x8 *>(slot + n * TILE_BYTES + (ks * 2 + 1) * 1024 + lane * 16);
#pragma unroll
        for (int tq = 0; tq < TQ; ++tq) {
          f32x16 c = acc[tq][n];
          if (RQ_KO & 1) {
            asm volatile("" ::"v"(bh), "v"(bl), "v"(ah[tq][ks]), "v"(al[tq][ks]));
          } else {
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[tq][ks], bh, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tq][ks], bl, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tq][ks], bh, c, 0, 0, 0);
          }
          acc[tq][n] = c;
        }
      }
    }
    // ---- epilogue of the step: C/D layout -- column (item) = lane & 31, row (query) = (e & 3) + 8 (e >> 2) + 4 (lane >> 5) --
    const int t0 = first_tile(s);
    if (RQ_KO & 4) {
      float sink = 0.f;
#pragma unroll
      for (int tq = 0; tq < TQ; ++tq)
#pragma unroll
        for (int e = 0; e < 16; ++e) sink += acc[tq][0][e];
      if (sink == 1.2345e-33f) a.S[0] = sink;
      continue;
    }
#pragma unroll
    for (int n = 0; n < NI; ++n) {
      const int item = 32 * (t0 + n) + c31;
      const float nrm = a.norms ? *reinterpret_cast<const float *>(slot + STEP_BYTES + (wave * NI + n) * 256 + c31 * 4) : 1.f;
#pragma unroll
      for (int tq = 0; tq < TQ; ++tq) {
        const int r_base = wave * 32 * TQ + 32 * tq + 4 * kh;  // workgroup-relative row of e = 0
        f32x16 &c = acc[tq][n];
        // real scores: scale out (a power of two: exact), divide by the item norm.  The emit pass without norms skips this and
        // tests the raw accumulators against thresholds in the scaled domain (tau_s)
        {
#pragma unroll
          for (int eg = 0; eg < 4; ++eg) {
            const float4 us = *reinterpret_cast<const float4 *>(unscale + r_base + 8 * eg);
            c[4 * eg] *= us.x, c[4 * eg + 1] *= us.y, c[4 * eg + 2] *= us.z, c[4 * eg + 3] *= us.w;
          }
          if (a.norms) {
#pragma unroll
            for (int e = 0; e < 16; ++e) c[e] = c[e] / nrm;
          }
        }
        if constexpr (MODE == 1) {
          // compact layout: block b of the subset occupies columns [128 b, 128 b + 128); items past the end score -FLT_MAX
          const int col = (block_of(s) * 4 + (s & 3) + n) * 32 + c31;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int q = qb * QROWS + r_base + (e & 3) + 8 * (e >> 2);
            float v = c[e];
            if (!(v == v)) v = INFINITY;
            if (q < a.nq) a.S[(size_t)q * a.sub_cols + col] = item < a.ni ? v : -FLT_MAX;
          }
        } else {
          // MODE 0: scores and the maximum of each (query, 64-item tile): an even 32-item tile opens the pair, the odd one closes
          // it (a workgroup walks whole 128-item blocks: pairs never straddle two workgroups)
          const bool opens = (t0 & 1) == 0;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int q = qb * QROWS + r_base + (e & 3) + 8 * (e >> 2);
            const float v = c[e];
            if (q < a.nq && item < a.ni) a.S[(size_t)q * a.ni + item] = v;
            const float m = item < a.ni ? v : -FLT_MAX;
            tmax[e] = opens ? m : fmaxf(tmax[e], m);
          }
          if (!opens) {
            const int tile64 = (32 * t0) / kTileItems;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              float m = tmax[e];
#pragma unroll
              for (int off = 16; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));  // over the 32 lanes of this kh
              const int q = qb * QROWS + r_base + (e & 3) + 8 * (e >> 2);
              if (c31 == 0 && q < a.nq && tile64 < a.n_tiles64) a.tile_max[(size_t)q * a.n_tiles64 + tile64] = m;
            }
          }
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stagings past the end
}

template <int KS, int TQ, int MODE> static size_t rq_lds_bytes() {
  constexpr int NI = 1;
  constexpr int SLOT = NI * KS * 2048 + 4 * 256 * NI;
  return (size_t)rq_stages<KS, MODE>() * SLOT;
}

// launch: KS = padded factors / 16 in {2, 4, 8, 16};  TQ = 2 up to f = 128 (MODE 1 / 2), else 1
template <int MODE> static void launch_score_resident(int KS, ResidentArgs a, int n_query_rows) {
  const int num_cus = ctx().num_cus;
  auto go = [&](auto ks_c, auto tq_c) {
    constexpr int K = decltype(ks_c)::value, T = decltype(tq_c)::value;
    a.n_qb = std::max(1, (n_query_rows + 128 * T - 1) / (128 * T));
    const int m = std::max(1, (2 * num_cus) / (8 * a.n_qb));
    a.chunks = 8 * m;
    auto kern = score_resident_kernel<K, T, MODE>;
    const size_t lds = rq_lds_bytes<K, T, MODE>();
    static bool attr_done[64] = {};  // (per instantiation and device: the attribute call costs a microsecond or two of every launch)
    const int dev = ctx().device;
    if (dev < 0 || dev >= 64 || !attr_done[dev]) {
      IMP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      if (dev >= 0 && dev < 64) attr_done[dev] = true;
    }
    kern<<<a.chunks * a.n_qb, 256, lds, stream()>>>(a);
    IMP_CHECK_HIP(hipGetLastError());
#ifdef RQ_CLOCK
    if (MODE >= 2) {
      unsigned long long h[4] = {0, 0, 0, 0};
      IMP_CHECK_HIP(hipStreamSynchronize(stream()));
      IMP_CHECK_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(rq_clock_dbg), sizeof(h)));
      if (h[1]) fprintf(stderr, "[rq-clock] item loop of one wavefront: %llu shader cycles in %.2f us = %.0f MHz, %llu steps, %.0f cycles per step\n",
                        h[0], h[1] / 100.0, h[0] / (h[1] / 100.0), h[2], (double)h[0] / (double)std::max<unsigned long long>(1, h[2]));
    }
#endif
  };
  using std::integral_constant;
  constexpr int TQ12 = MODE == 0 ? 1 : 2;
  if constexpr (MODE == 3) {  // one plane of query fragments: twice the query rows per wavefront fit
    if (KS == 16) {
      go(integral_constant<int, 16>{}, integral_constant<int, 2>{});
      return;
    }
  }
  if (KS == 2) go(integral_constant<int, 2>{}, integral_constant<int, TQ12>{});
  else if (KS == 4) go(integral_constant<int, 4>{}, integral_constant<int, TQ12>{});
  else if (KS == 8) go(integral_constant<int, 8>{}, integral_constant<int, TQ12>{});
  else if (KS == 16) go(integral_constant<int, 16>{}, integral_constant<int, 1>{});
  else throw std::invalid_argument("score_resident: factors must pad to 32, 64, 128 or 256");
}
// query rows a launch's planes must be padded to (whole query blocks of the widest form)
static inline size_t rq_query_pad(size_t rows) { return (rows + 255) / 256 * 256; }
static inline int rq_ks_for(int f) { return f <= 32 ? 2 : (f <= 64 ? 4 : (f <= 128 ? 8 : (f <= 256 ? 16 : 0))); }

#endif  // IMPLICIT_AMD_CSRC_TOPK_RESIDENT_H_
