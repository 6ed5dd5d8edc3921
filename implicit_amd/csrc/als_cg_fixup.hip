// Fix-up solver of the f = 64 / 128 CG path: rows a fast kernel could not finish are re-solved in fp32, one wavefront per row.
// Arithmetic contract: the oracle's CG (implicit/cpu/_als.pyx:152-248).
#include <type_traits>

#include "als_qtile.h"
#include "common.h"

namespace imp {

// ---- fix-up of the rows another kernel left unsolved ---------------------------------------------------------------------
// Producer: the normal-matrix kernels (als_cg_nm.hip: a row whose fp16-split operands left the fp16 range; until round 5 also a
// cluster of workgroups whose exchange was lost -- those kernels are gone).  One wavefront per listed row, everything streamed in fp32: lane l owns the FC = F / 64 consecutive factors FC l ..;
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


}  // namespace imp
