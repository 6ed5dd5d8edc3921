// K1c: CG for rows of 513 .. 4096 nonzeros with the whole row RESIDENT in the registers of a cluster of workgroups.
//
// Arithmetic contract: the oracle's CG (implicit/cpu/_als.pyx:152-248), as in als_cg_q.hip.  A row of n nonzeros does
// not fit the register file of one compute unit beyond 512 entries at f = 128 (a CU has 512 KB of vector registers, an
// entry is 512 bytes, and the CG state needs room too), which is why round 1 streamed such rows 1 + cg_steps times.  Here
// a row is dealt to a CLUSTER of CL = 4 / 8 / 16 workgroups of 8 wavefronts (256 entries per workgroup, two workgroups
// per CU), every wavefront keeps its 32-entry tile for all passes as in the team kernels, and per pass the cluster
// exchanges ONE f-vector per workgroup through L2:
//
//   wave partial --LDS--> workgroup partial --global, tagged--> all CL workgroup partials --LDS--> every wave sums them
//
// in a fixed order, so all CL workgroups hold bit-identical CG scalars and take the same early-exit branches without
// any further agreement.  The exchange needs no flag and no fence: a partial travels as 8-byte {value, sequence number}
// granules written and read with relaxed agent-scope atomics (sc1: served by L2 / the fabric, never by a CU's L1); a
// reader polls the granules themselves until all carry the current sequence number.  The slots are double-buffered (a
// workgroup can run at most one exchange ahead of the slowest member) and zeroed by the host before every launch, so a
// sequence number never repeats within a slot's lifetime.
//
// Placement: workgroup b runs on XCD b % 8 (observed; a speed matter only), so the CL members of a cluster are the
// workgroups x + 8 (CL q + m) -- one XCD, one L2; the members verify it (XCC id in the tag bits of the first exchange)
// before they rely on it.  Residency: a cluster's members are neighbours in the dispatch order of their XCD.  With a grid of
// at most the resident capacity (2 workgroups per CU) every cluster is resident from the start; with an oversubscribed grid
// (Context::oversub, multi-GPU driver) or with part of the device held by another stream's kernels, the slots a finished
// cluster frees go to the next cluster as a whole, and a partly resident cluster merely waits for slots that complete
// clusters keep freeing.  Every poll is bounded all the same, and a timed-out exchange costs time, not correctness (round 4):
//   * a workgroup whose poll expired is FAULTED for the rest of the launch: it keeps taking part in the exchanges (so that
//     nobody waits for it) but marks its granules as poisoned, which faults whoever reads them; a reader that finds a slot
//     already carrying a LATER sequence number (the cluster moved on without it) faults at once;
//   * a faulted cluster never stores an iterate: the owner of the store appends the row id to a device-side list instead, so
//     every row is either solved completely or still holds the iterate it had;
//   * `als_cg_fault_fixup_kernel`, queued right behind the cluster launches, re-solves the listed rows (normally none: it
//     reads a zero and exits) by streaming them, one wavefront per row -- on the device, in stream order, so it also works in
//     deferred mode where the host looks at nothing until the end of the iteration.  The host-visible fault word only makes
//     the next synchronisation print a warning.
#include <type_traits>

#include "als_qtile.h"
#include "common.h"

namespace imp {

#ifndef IMP_CLUSTER_POLL_NAP
#define IMP_CLUSTER_POLL_NAP 2  // 64-cycle units between two polls of an exchange (0 / 1 / 2 / 4 measured alike on the configs[1] shape, gpurun_out/r4a)
#endif
namespace {
constexpr int kClusterWaves = 8;       // wavefronts per workgroup
// A poll gives up by the constant-rate wall clock (100 MHz), not by counting polls: a partly resident cluster may have
// to wait for a whole share of the clusters ahead of it (and, beside a resident collective, for the slots that holds) --
// milliseconds; 4 s is far beyond any legitimate wait and still turns a lost member into an error instead of a hang.
constexpr long long kWaitLimitTicks = 400'000'000ll;
// tag layout of a granule's upper word: [31:28] XCC id of the sender, [27] poison (the sender is faulted), [26:0] sequence number
constexpr unsigned kSeqMask = 0x07FFFFFFu, kPoison = 0x08000000u;
}  // namespace

// STATS (debug, IMP_CG_STATS=1): s_memtime ticks summed over waves -- [0] row start -> tile resident  [1] passes
//   [2] exchanges (barriers, publish, poll, sum)  [3] dots / CG update  [7] wave-rows
template <int F, int CL, bool STATS, typename ST>
__global__ __launch_bounds__(64 * kClusterWaves, 4) void als_cg_cluster_kernel(
    const int32_t *__restrict__ order, int first, int count, const int32_t *__restrict__ indptr,
    const int32_t *__restrict__ indices, const float *__restrict__ data, ST *__restrict__ X, const ST *__restrict__ Y,
    const float *__restrict__ A0, int cg_steps, unsigned long long *xchg, unsigned *fault, int allow_plain, unsigned *fault_count,
    unsigned *fault_rows, int fault_capacity, long long wait_limit, int debug_drop, unsigned long long *__restrict__ stats = nullptr) {
  unsigned long long tk[4] = {0, 0, 0, 0}, t_last = 0, t_rows = 0;
  auto tick = [&](int slot) {  // charge the time since the previous tick to `slot`
    if constexpr (STATS) {
      __builtin_amdgcn_sched_barrier(0);
      unsigned long long now = __builtin_amdgcn_s_memtime();
      if (slot >= 0) tk[slot] += now - t_last;
      t_last = now;
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  constexpr int FC = F / 64, FE = F / 16, T = 32, WAVES = kClusterWaves, W = WAVES * CL;
  constexpr int WD = W < F / 4 ? W : F / 4;  // wavefronts of the cluster that share the dense product: 4 NJ gramian rows each
  constexpr int NJ = F / WD / 4;
  constexpr int PER = (CL + WAVES - 1) / WAVES;  // exchange slots polled per wavefront
  static_assert(F % (4 * WD) == 0 && CL >= 2, "cluster shape");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *A0s = smem;                     // [F][F]  (only in the workgroups that own dense rows)
  float *scratch = A0s + (size_t)F * F;  // [WAVES][F]  wave partials of the combine; between combines: wave-private operand copy
  float *xb = scratch + (size_t)WAVES * F;  // [CL][F]  the cluster's workgroup partials after the exchange
  int *xflag = reinterpret_cast<int *>(xb + (size_t)CL * F);  // [CL]  tag of member m's granules in the last exchange
  int *wg_fault = xflag + 16;                                  // [1]   set by any wave of this workgroup that faulted
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
  const int m = jx % CL;                          // member index inside the cluster
  const int ncl = ((int)(gridDim.x >> 3) / CL) * 8;  // clusters in the grid
  const int cid = (jx / CL) * 8 + xcd;
  const int g = m * WAVES + wave;  // wavefront index inside the cluster
  const bool dense = g < WD;       // wave-uniform
  const int j_begin = (F / WD) * g;
  if (m * WAVES < WD)
    for (int e = threadIdx.x; e < F * F; e += 64 * WAVES) A0s[e] = A0[e];
  if (threadIdx.x == 0) *wg_fault = 0;
  __syncthreads();
  float *myvec = scratch + (size_t)wave * F;
  unsigned long long *slots = xchg + (size_t)cid * 2 * CL * 64 * FC;

  // ---- the exchange -------------------------------------------------------------------------------------------------
  unsigned seq = 0;
  int parity = 0;
  bool faulted = false;
  // Placement check: every granule's tag carries the sender's XCC id (HW_REG_XCC_ID[3:0]) in its top four bits.  The
  // first exchange of the kernel is written through; once a workgroup has seen that all CL members report its own id --
  // every member sees the same CL tags, hence takes the same decision -- later exchanges use plain stores.
  const unsigned xcc_tag = (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) << 28;
  bool same_xcd = false;
  auto combine = [&](float (&acc)[FC]) {
    // addresses of the exchange are re-derived per call: hoisted out of the row loop they would cost registers the
    // kernel does not have (the tile fills the file) and come back as scratch reloads
    int ln = lane;
    asm volatile("" : "+v"(ln));
#pragma unroll
    for (int c = 0; c < FC; ++c) scratch[wave * F + QL<F>::cfactor(ln, c)] = acc[c];
    __syncthreads();  // B1: wave partials visible; every wave has finished reading xb of the previous exchange
    ++seq;
    unsigned long long *slot = slots + (size_t)parity * CL * 64 * FC;
    parity ^= 1;
    if (wave == 0) {  // workgroup partial (fixed order) -> this member's slot
#pragma unroll
      for (int c = 0; c < FC; ++c) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) s += scratch[w * F + QL<F>::cfactor(ln, c)];
        const unsigned long long granule =
            ((unsigned long long)(seq | xcc_tag | (faulted ? kPoison : 0u)) << 32) | (unsigned long long)__float_as_uint(s);
        // test hook (IMP_DEBUG_CLUSTER_DROP=n): member 1 of cluster 0 never publishes its n-th exchange
        if (debug_drop > 0 && cid == 0 && m == 1 && seq == (unsigned)debug_drop) continue;
        // members on one XCD share its L2: a plain store lands there and the readers' sc1 loads (which bypass only
        // their L1) hit it; an sc1 store writes through to the fabric and drops the line, which the readers then
        // fetch at the cross-XCD latency -- required when the members sit on different XCDs, whose L2s are not coherent
        if (same_xcd) __hip_atomic_store(slot + ((m * 64 + ln) * FC + c), granule, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_store(slot + ((m * 64 + ln) * FC + c), granule, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (wave < CL) {  // wave w collects members w, w + 8, ...: all of them in flight together
      unsigned long long granule[PER][FC];
      int spins = 0;
      long long wait_since = 0;
      while (true) {
        bool ok = true, ahead = false, poisoned = false;
#pragma unroll
        for (int k = 0; k < PER; ++k)
#pragma unroll
          for (int c = 0; c < FC; ++c) {
            granule[k][c] = __hip_atomic_load(slot + (((wave + k * WAVES) * 64 + ln) * FC + c), __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_AGENT);
            const unsigned tag = (unsigned)(granule[k][c] >> 32);
            ok = ok && (tag & kSeqMask) == seq;
            ahead = ahead || (tag & kSeqMask) > seq;   // the slot already belongs to a later exchange: this one is lost
            poisoned = poisoned || ((tag & kSeqMask) == seq && (tag & kPoison));
          }
        if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) {
          if (__builtin_amdgcn_ballot_w64(poisoned) != 0ull) {  // a faulted member took part: its partial is not to be trusted
            if (!faulted && lane == 0) __hip_atomic_store(fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            faulted = true;
          }
          break;
        }
        bool expired = __builtin_amdgcn_ballot_w64(ahead) != 0ull;
        if ((++spins & 255) == 0) {  // the clock is read once per 256 polls
          const long long now = (long long)wall_clock64();
          if (wait_since == 0) wait_since = now;
          expired = expired || now - wait_since > wait_limit;
        }
        if (faulted || expired) {  // never expected: give up instead of hanging the device
          if (!faulted && lane == 0) __hip_atomic_store(fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          faulted = true;
          break;
        }
        __builtin_amdgcn_s_sleep(IMP_CLUSTER_POLL_NAP);
      }
#pragma unroll
      for (int k = 0; k < PER; ++k)
#pragma unroll
        for (int c = 0; c < FC; ++c) xb[(wave + k * WAVES) * F + QL<F>::cfactor(ln, c)] = __uint_as_float((unsigned)granule[k][c]);
#pragma unroll
      for (int k = 0; k < PER; ++k) xflag[wave + k * WAVES] = (int)(granule[k][0] >> 32);  // the tag: read by the placement check
      if (faulted && lane == 0) *wg_fault = 1;
    }
    __syncthreads();  // B2: all CL partials in LDS; wave 0 has finished reading the wave partials
    faulted = *wg_fault != 0;  // a fault is the whole workgroup's: its next granules carry the poison bit, its rows are not stored
#pragma unroll
    for (int c = 0; c < FC; ++c) {
      float s = 0.f;
#pragma unroll
      for (int mm = 0; mm < CL; ++mm) s += xb[mm * F + QL<F>::cfactor(ln, c)];
      acc[c] = s;
    }
  };

  // one pass: ae (expanded, partial over this wave's work) = sign * [A0 rows of this wave] . v + [tile entries] weights
  auto pass = [&](auto first_tag, const QTile<F> &tile, const float (&v)[FC], float (&acc)[FC], bool work) {
    constexpr bool FIRST = decltype(first_tag)::value;
    float ve[FE], ae[FE];
#pragma unroll
    for (int e = 0; e < FE; ++e) ae[e] = 0.f;
    if (work) {  // wave-uniform, identical across the cluster
      if (dense) {
#pragma unroll
        for (int c = 0; c < FC; ++c) myvec[QL<F>::cfactor(lane, c)] = v[c];  // wave-private: no barrier needed
        gram_matvec_q<F, NJ>(A0s, F, myvec, lane, j_begin, ae);
        if constexpr (FIRST) {
#pragma unroll
          for (int e = 0; e < FE; ++e) ae[e] = -ae[e];  // r = b - A x: the dense part enters with a minus sign
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      expand_vector<F>(v, ve);
      qtile_apply<F, FIRST>(tile, ve, ae);
    }
    reduce_expanded<F>(ae, acc);
  };

  // this cluster's rows: i = cid + k ncl; the loop bounds depend on cid alone, so all members run the same exchanges
  auto row_id = [&](int i) { return order[first + min(i, count - 1)]; };  // uniform address: scalar load
  int u1 = row_id(cid), u2 = row_id(cid + ncl), u3 = row_id(cid + 2 * ncl);
  int rb1 = indptr[u1], re1 = indptr[u1 + 1], rb2 = indptr[u2], re2 = indptr[u2 + 1];
  int col_next;
  float c_next;
  // even shares of a row for the W wavefronts of the cluster, rounded up to whole 4-entry tile steps (als_cg_q.hip)
  auto slice = [&](int rb, int re, int &k0, int &cnt) {
    const int chunk = min(T, (((re - rb) + W - 1) / W + 3) & ~3);
    k0 = min(rb + chunk * g, re);
    cnt = min(chunk, re - k0);
  };
  int k0_next, cnt_next;
  slice(rb1, re1, k0_next, cnt_next);
  fetch_entries(indices, data, lane, k0_next, max(k0_next + cnt_next, rb1 + 1), col_next, c_next);
  for (int i = cid; i < count; i += ncl) {
    const int u = u1;
    u1 = u2, rb1 = rb2, re1 = re2;                    // row i + ncl: complete
    u2 = u3, rb2 = indptr[u2], re2 = indptr[u2 + 1];  // row i + 2 ncl: row id known -> its range
    u3 = row_id(i + 3 * ncl);                         // row i + 3 ncl: row id
    ST *xrow = X + (size_t)u * F;
    float x[FC], r[FC], p[FC], Ap[FC];
    const int cnt = cnt_next;  // this wave's slice of the row (may be empty)
    QTile<F> tile;
    load_compact<F>(xrow, lane, x);
    tick(-1);
    load_qtile_staged<F>(tile, col_next, c_next, Y, lane, cnt);
    slice(rb1, re1, k0_next, cnt_next);
    fetch_entries(indices, data, lane, k0_next, max(k0_next + cnt_next, rb1 + 1), col_next, c_next);  // entries of row i + ncl
    if constexpr (STATS) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      tick(0);
      t_rows += 1;
    }

    // r = -(A0 x) + sum_k (c+ - (|c|-1) y.x) y        (_als.pyx:187-201)
    pass(std::true_type{}, tile, x, r, true);
    tick(1);
    combine(r);
    if (seq == 1u && allow_plain) {  // after the kernel's first exchange: did every member report this workgroup's XCC?
      const bool agree = lane < CL ? ((unsigned)xflag[lane] & 0xF0000000u) == xcc_tag : true;
      same_xcd = __builtin_amdgcn_ballot_w64(!agree) == 0ull;
    }
    tick(2);
#pragma unroll
    for (int c = 0; c < FC; ++c) p[c] = r[c];
    float rsold = dot_compact<F>(r, r);
    bool active = rsold >= 1e-20f;  // else: x untouched (_als.pyx:206)
    const bool store = active && g == 0;

    for (int it = 0; it < cg_steps; ++it) {
      tick(3);
      pass(std::false_type{}, tile, p, Ap, active);
      tick(1);
      combine(Ap);
      tick(2);
      if (active) {
        float alpha = rsold / dot_compact<F>(p, Ap);
#pragma unroll
        for (int c = 0; c < FC; ++c) {
          x[c] = fmaf(alpha, p[c], x[c]);
          r[c] = fmaf(-alpha, Ap[c], r[c]);
        }
        float rsnew = dot_compact<F>(r, r);
        if (rsnew < 1e-20f) {
          active = false;  // the oracle breaks here (_als.pyx:235); the whole cluster takes the same branch
        } else {
          float beta = rsnew / rsold;
#pragma unroll
          for (int c = 0; c < FC; ++c) p[c] = fmaf(beta, p[c], r[c]);
          rsold = rsnew;
        }
      }
    }
    if (g == 0) {  // the owner of the row's store
      if (faulted) {  // an exchange of this row (or an earlier one) was lost: the row keeps its iterate and goes to the fix-up list
        if (lane == 0) {
          const unsigned at = atomicAdd(fault_count, 1u);
          if ((int)at < fault_capacity) fault_rows[at] = (unsigned)u;
        }
      } else if (store) {
        store_compact<F>(xrow, lane, x);
      }
    }
    tick(3);
  }
  if constexpr (STATS) {
    if (lane == 0) {
      for (int i = 0; i < 4; ++i) atomicAdd(&stats[i], tk[i]);
      atomicAdd(&stats[7], t_rows);
    }
  }
}

// ---- fix-up of the rows another kernel left unsolved ---------------------------------------------------------------------
// Two producers: a faulted cluster (above) and the normal-matrix kernels (als_cg_nm.hip: a row whose fp16-split operands left the
// fp16 range).  One wavefront per listed row, everything streamed in fp32: lane l owns the FC = F / 64 consecutive factors FC l ..;
// a nonzero is one coalesced row read, one wave-wide dot product, one axpy (four nonzeros' reads in flight); the gramian comes
// from global memory (L2) row by row with the operand broadcast from a wave-private LDS copy.  The oracle's CG step by step
// (_als.pyx:179-244); only the summation order differs from the producers'.  Slow (a millisecond for a 4096-nonzero row) and
// never expected to have work; `total` (host-mapped, imp_solver_fixup_rows) counts the rows it has re-solved.
template <int F, typename ST>
__global__ __launch_bounds__(256) void als_cg_fault_fixup_kernel(const unsigned *__restrict__ fault_count,
                                                                 const unsigned *__restrict__ fault_rows, int capacity,
                                                                 const int32_t *__restrict__ indptr,
                                                                 const int32_t *__restrict__ indices,
                                                                 const float *__restrict__ data, ST *__restrict__ X,
                                                                 const ST *__restrict__ Y, const float *__restrict__ A0, int cg_steps,
                                                                 unsigned long long *total) {
  constexpr int FC = F / 64, WAVES = 4, U = 4;
  __shared__ float vecs[WAVES][F];
  const int n = min((int)fault_count[0], capacity);
  if (n == 0) return;
  if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_fetch_add(total, (unsigned long long)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float *vec = vecs[wave];
  auto load_vec = [&](const ST *row, float (&v)[FC]) {
#pragma unroll
    for (int c = 0; c < FC; ++c) v[c] = load1(row + FC * lane + c);
  };
  // acc = sign * A0 v + sum_k w_k y_k,  w_k = FIRST ? c+ - (|c|-1) y_k.v : (|c|-1) y_k.v
  auto apply = [&](bool first, int rb, int re, const float (&v)[FC], float (&acc)[FC]) {
#pragma unroll
    for (int c = 0; c < FC; ++c) vec[FC * lane + c] = v[c];  // wave-private: no barrier
#pragma unroll
    for (int c = 0; c < FC; ++c) acc[c] = 0.f;
    for (int j = 0; j < F; ++j) {
      const float vj = vec[j];
#pragma unroll
      for (int c = 0; c < FC; ++c) acc[c] = fmaf(A0[(size_t)j * F + FC * lane + c], vj, acc[c]);
    }
    if (first) {
#pragma unroll
      for (int c = 0; c < FC; ++c) acc[c] = -acc[c];
    }
    for (int k0 = rb; k0 < re; k0 += U) {
      float conf[U], y[U][FC];
#pragma unroll
      for (int q = 0; q < U; ++q) {
        const int k = min(k0 + q, re - 1);
        conf[q] = data[k];
        load_vec(Y + (size_t)indices[k] * F, y[q]);
      }
#pragma unroll
      for (int q = 0; q < U; ++q) {
        if (k0 + q < re) {  // wave-uniform
          const float d = wave_allsum(dot_local<FC>(y[q], v));
          const float cm1 = fabsf(conf[q]) - 1.f;
          const float w = first ? fmaxf(conf[q], 0.f) - cm1 * d : cm1 * d;
#pragma unroll
          for (int c = 0; c < FC; ++c) acc[c] = fmaf(w, y[q][c], acc[c]);
        }
      }
    }
  };
  for (int i = blockIdx.x * WAVES + wave; i < n; i += gridDim.x * WAVES) {
    const int u = (int)fault_rows[i];
    const int rb = indptr[u], re = indptr[u + 1];
    ST *xrow = X + (size_t)u * F;
    float x[FC], r[FC], p[FC], Ap[FC];
    load_vec(xrow, x);
    apply(true, rb, re, x, r);
#pragma unroll
    for (int c = 0; c < FC; ++c) p[c] = r[c];
    float rsold = wave_allsum(dot_local<FC>(r, r));
    if (rsold < 1e-20f) continue;  // x untouched (_als.pyx:206)
    for (int it = 0; it < cg_steps; ++it) {
      apply(false, rb, re, p, Ap);
      const float alpha = rsold / wave_allsum(dot_local<FC>(p, Ap));
#pragma unroll
      for (int c = 0; c < FC; ++c) {
        x[c] = fmaf(alpha, p[c], x[c]);
        r[c] = fmaf(-alpha, Ap[c], r[c]);
      }
      const float rsnew = wave_allsum(dot_local<FC>(r, r));
      if (rsnew < 1e-20f) break;
      const float beta = rsnew / rsold;
#pragma unroll
      for (int c = 0; c < FC; ++c) p[c] = fmaf(beta, p[c], r[c]);
      rsold = rsnew;
    }
#pragma unroll
    for (int c = 0; c < FC; ++c) store1(xrow + FC * lane + c, x[c]);
  }
}

// host-mapped counter of the rows the fix-up kernel has re-solved on this device (imp_solver_fixup_rows)
unsigned long long *fixup_total() {
  auto &c = ctx();
  if (!c.fixup_total) {
    IMP_CHECK_HIP(hipHostMalloc(reinterpret_cast<void **>(&c.fixup_total), sizeof(unsigned long long), hipHostMallocMapped));
    *c.fixup_total = 0ull;
  }
  return c.fixup_total;
}

// queued behind the kernels that fill the list; normally reads a zero and exits
template <int F, typename T>
void launch_cg_fixup(const unsigned *count, const unsigned *rows, int capacity, const imp_csr *C, T *X, const T *Y, const float *A0,
                     int cg_steps) {
  if (capacity <= 0) return;
  IMP_PROF("als_cg_fixup");
  als_cg_fault_fixup_kernel<F, T><<<std::min((capacity + 3) / 4, ctx().num_cus * 2), 256, 0, stream()>>>(
      count, rows, capacity, C->indptr.data(), C->indices.data(), C->data.data(), X, Y, A0, cg_steps, fixup_total());
  IMP_CHECK_HIP(hipGetLastError());
}
template void launch_cg_fixup<64, float>(const unsigned *, const unsigned *, int, const imp_csr *, float *, const float *, const float *, int);
template void launch_cg_fixup<128, float>(const unsigned *, const unsigned *, int, const imp_csr *, float *, const float *, const float *, int);
template void launch_cg_fixup<64, __half>(const unsigned *, const unsigned *, int, const imp_csr *, __half *, const __half *, const float *, int);
template void launch_cg_fixup<128, __half>(const unsigned *, const unsigned *, int, const imp_csr *, __half *, const __half *, const float *, int);

template <int F, int CL, typename T>
static void launch_cluster(const imp_csr *C, int first, int count, T *X, const T *Y, const float *A0, int cg_steps,
                           unsigned long long *xchg, unsigned *fault_count, unsigned *fault_rows, int fault_capacity, const char *name) {
  if (count <= 0) return;
  // IMP_CLUSTER_WAIT_MS: how long a poll waits before it declares the exchange lost (default 4 s; tests lower it);
  // IMP_DEBUG_CLUSTER_DROP=n: member 1 of cluster 0 withholds its n-th exchange (exercises the fault path)
  static const long long wait_limit = getenv("IMP_CLUSTER_WAIT_MS") ? std::max(1, atoi(getenv("IMP_CLUSTER_WAIT_MS"))) * 100'000ll : kWaitLimitTicks;
  static const int debug_drop = getenv("IMP_DEBUG_CLUSTER_DROP") ? atoi(getenv("IMP_DEBUG_CLUSTER_DROP")) : 0;
  constexpr int FC = F / 64, BLOCK = 64 * kClusterWaves;
  const size_t lds = ((size_t)F * F + (size_t)kClusterWaves * F + (size_t)CL * F + 32) * sizeof(float);
  auto kern = als_cg_cluster_kernel<F, CL, false, T>;
  IMP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  // co-resident workgroups: what the occupancy query admits per CU (2 by design), never more than 2; the cluster
  // protocol only needs ONE complete cluster resident, which any grid in dispatch order provides
  int per_cu = 0;
  IMP_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, BLOCK, lds));
  per_cu = std::max(1, std::min(per_cu, 2));
  const int max_clusters_per_xcd = std::max(1, ctx().num_cus * per_cu / 8 / CL) * ctx().oversub;
  const int clusters_per_xcd = std::min(max_clusters_per_xcd, (count + 7) / 8);
  const int grid = 8 * CL * clusters_per_xcd;
  (void)FC;
  static const bool allow_plain = getenv("IMP_CLUSTER_SC1") == nullptr;  // IMP_CLUSTER_SC1=1: write-through stores always (A/B)
  static const bool want_stats = getenv("IMP_CG_STATS") != nullptr;
  if (want_stats) {  // debug: per-phase tick sums of this launch, printed to stderr
    static unsigned long long *stats = nullptr;
    if (!stats) IMP_CHECK_HIP(hipMalloc(&stats, 8 * sizeof(unsigned long long)));
    IMP_CHECK_HIP(hipMemsetAsync(stats, 0, 8 * sizeof(unsigned long long), stream()));
    auto skern = als_cg_cluster_kernel<F, CL, true, T>;
    IMP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(skern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    skern<<<grid, BLOCK, lds, stream()>>>(C->order.data(), first, count, C->indptr.data(), C->indices.data(), C->data.data(), X, Y,
                                         A0, cg_steps, xchg, ctx().cluster_fault, allow_plain ? 1 : 0, fault_count, fault_rows, fault_capacity, wait_limit, debug_drop, stats);
    unsigned long long h[8];
    IMP_CHECK_HIP(hipMemcpyAsync(h, stats, sizeof(h), hipMemcpyDeviceToHost, stream()));
    IMP_CHECK_HIP(hipStreamSynchronize(stream()));
    const double n = h[7] ? (double)h[7] : 1.0;
    fprintf(stderr, "[cg-stats] %s rows=%d grid=%d wave-rows=%.0f  cycles/wave-row: gather %.1f passes %.1f exchanges %.1f update %.1f\n",
            name, count, grid, n, h[0] / n, h[1] / n, h[2] / n, h[3] / n);
    return;
  }
  IMP_PROF(name);
  kern<<<grid, BLOCK, lds, stream()>>>(C->order.data(), first, count, C->indptr.data(), C->indices.data(), C->data.data(), X, Y,
                                      A0, cg_steps, xchg, ctx().cluster_fault, allow_plain ? 1 : 0, fault_count, fault_rows, fault_capacity, wait_limit, debug_drop, nullptr);
  IMP_CHECK_HIP(hipGetLastError());
}

// rows of (256, 512] nonzeros fit one 16-wave workgroup (als_cg_q.hip, team16) -- but that kernel holds a CU with a single
// workgroup, whose gather and combine phases nothing overlaps; as a cluster of TWO 8-wave workgroups the row shares its CUs
// with another row.  IMP_TEAM16_CLUSTER=1 selects it (A/B).
bool team16_as_cluster() {
  static const bool on = getenv("IMP_TEAM16_CLUSTER") != nullptr && getenv("IMP_NO_CLUSTER") == nullptr;
  return on;
}

template <int F, typename T> static void run_clusters(const imp_csr *C, T *X, const T *Y, const float *A0, int cg_steps) {
  const int32_t *cut = C->cluster_cut;  // rows longer than 4096 / 2048 / 1024 / 512: classes (2048,4096] (1024,2048] (512,1024]
  const int32_t *b = C->bin_start;
  const bool with16 = team16_as_cluster() && b[2] - b[1] > 0;
  if (cut[3] - cut[0] <= 0 && !with16) return;
  auto &c = ctx();
  // exchange slots: [class][cluster][2][CL][64 FC] granules; clusters * CL <= workgroups in flight <= 2 per CU
  constexpr size_t FC = F / 64;
  const size_t per_class = (size_t)c.num_cus * 2 * c.oversub * 2 * 64 * FC;
  // + one granule behind the slots: the counter of the fix-up list, so that ONE memset resets both
  if (c.cluster_xchg.size < 4 * per_class + 1) c.cluster_xchg.alloc(4 * per_class + 1);
  if (!c.cluster_fault) {
    IMP_CHECK_HIP(hipHostMalloc(reinterpret_cast<void **>(&c.cluster_fault), sizeof(unsigned), hipHostMallocMapped));
    *c.cluster_fault = 0u;
  }
  unsigned long long *xchg = c.cluster_xchg.data();
  // fix-up list: *fault_count = number of rows a faulted cluster left unsolved, fault_rows[0 ..] = their ids (every cluster row
  // at most once)
  const int capacity = (cut[3] - cut[0]) + (with16 ? b[2] - b[1] : 0);
  if (c.cluster_fault_rows.size < (size_t)capacity) c.cluster_fault_rows.alloc((size_t)capacity);
  unsigned *fault_rows = c.cluster_fault_rows.data();
  unsigned *fault_count = reinterpret_cast<unsigned *>(xchg + 4 * per_class);
  {
    IMP_PROF("als_cg_cluster_reset");
    IMP_CHECK_HIP(hipMemsetAsync(xchg, 0, (4 * per_class + 1) * sizeof(unsigned long long), stream()));
  }
  launch_cluster<F, 16, T>(C, cut[0], cut[1] - cut[0], X, Y, A0, cg_steps, xchg, fault_count, fault_rows, capacity, "als_cg_cluster16_rows");
  launch_cluster<F, 8, T>(C, cut[1], cut[2] - cut[1], X, Y, A0, cg_steps, xchg + per_class, fault_count, fault_rows, capacity, "als_cg_cluster8_rows");
  launch_cluster<F, 4, T>(C, cut[2], cut[3] - cut[2], X, Y, A0, cg_steps, xchg + 2 * per_class, fault_count, fault_rows, capacity, "als_cg_cluster4_rows");
  if (with16)
    launch_cluster<F, 2, T>(C, b[1], b[2] - b[1], X, Y, A0, cg_steps, xchg + 3 * per_class, fault_count, fault_rows, capacity, "als_cg_team16_rows");
  // after a lost exchange: re-solves the rows the faulted clusters left untouched
  launch_cg_fixup<F, T>(fault_count, fault_rows, capacity, C, X, Y, A0, cg_steps);
}

// true if a cluster kernel of an earlier launch on this device gave up on an exchange (looked at after a stream
// synchronisation); clears the word and says so on stderr -- the rows concerned were re-solved by the fix-up kernel, so this is
// a performance event (a 4 s wait), not an error
bool cluster_fault_pending() {
  auto &c = ctx();
  if (!c.cluster_fault || *c.cluster_fault == 0u) return false;
  *c.cluster_fault = 0u;
  fprintf(stderr, "[implicit_amd] warning: a cluster exchange of the CG sweep timed out on device %d; the rows of the clusters "
                  "concerned were re-solved by the streamed fix-up kernel (als_cg_cluster.hip)\n", c.device);
  return true;
}

template <typename T> void least_squares_cg_cluster(const imp_csr *C, T *X, const T *Y, const float *A0, int f, int cg_steps) {
  if (f == 128) run_clusters<128, T>(C, X, Y, A0, cg_steps);
  else if (f == 64) run_clusters<64, T>(C, X, Y, A0, cg_steps);
  else throw std::invalid_argument("least_squares_cg_cluster: f must be 64 or 128");
}
template void least_squares_cg_cluster<float>(const imp_csr *, float *, const float *, const float *, int, int);
template void least_squares_cg_cluster<__half>(const imp_csr *, __half *, const __half *, const float *, int, int);

}  // namespace imp
