// Micro-benchmark: sustained rate of v_mfma_f32_32x32x16_f16 from registers -- 1 / 2 / 4 dependent accumulator chains per
// wavefront, 1 or 2 wavefronts per SIMD, operands shared or distinct per MFMA (the scoring kernel's pattern: 48 MFMAs over
// two chains with a different A or B register quad each).  Prints cycles per MFMA per SIMD (ideal: 32 = 8 passes) and TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f16_rate profiles/micro/mfma_f16_rate.hip && /tmp/mfma_f16_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ unsigned long long clk[2];

__device__ __forceinline__ void mfma(f32x16 &c, const f16x8 &a, const f16x8 &b) {
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

template <int NACC, int NOPS>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0) {
  f16x8 a[NOPS], b[NOPS];
#pragma unroll
  for (int i = 0; i < NOPS; ++i) {
    const _Float16 x = (_Float16)(a0 + (float)i), y = (_Float16)(a0 * 0.5f + (float)i);
    a[i] = f16x8{x, x, x, x, x, x, x, x};
    b[i] = f16x8{y, y, y, y, y, y, y, y};
    asm volatile("" : "+v"(a[i]), "+v"(b[i]));
  }
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) {
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    asm volatile("" : "+v"(acc[i]));
  }
  const long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int o = 0; o < NOPS; ++o)
#pragma unroll
      for (int i = 0; i < NACC; ++i) mfma(acc[i], a[o], b[(o + i) % NOPS]);
  }
  if (blockIdx.x == 3 && threadIdx.x == 0 && iters > 100) {
    clk[0] = (unsigned long long)(__builtin_readcyclecounter() - c0), clk[1] = (unsigned long long)(wall_clock64() - w0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  if (s == 1.2345f) out[0] = s;
}

template <typename K> static void run(const char *name, K kern, int blocks, int iters, int per_iter, float *out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  kern<<<blocks, 256>>>(out, 10, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kern<<<blocks, 256>>>(out, iters, 1.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = (double)blocks / 256.0 * iters * per_iter;  // one wavefront of each block per SIMD
  const double cyc = ms * 1e-3 * 2.4e9 / mfma_per_simd;
  const double tf = (double)blocks * 4 * iters * per_iter * 32768.0 / (ms * 1e-3) / 1e12;
  unsigned long long h[2];
  hipMemcpyFromSymbol(h, HIP_SYMBOL(clk), sizeof(h));
  const double mhz = h[0] / (h[1] / 100.0);  // wall_clock64: 100 MHz
  printf("%-44s blocks %4d  %.1f ns per MFMA per SIMD  %.0f TFLOP/s   in-kernel shader clock %.0f MHz -> %.1f shader cycles per MFMA per SIMD\n", name,
         blocks, cyc / 2.4, tf, mhz, (double)h[0] / ((double)iters * per_iter * (blocks / 256.0)));
}

int main() {
  float *out;
  hipMalloc(&out, 4);
  const int iters = 20000;
  for (int blocks : {256, 512}) {
    run("1 chain, 1 operand pair", k<1, 1>, blocks, iters, 1, out);
    run("2 chains, 1 operand pair", k<2, 1>, blocks, iters, 2, out);
    run("4 chains, 1 operand pair", k<4, 1>, blocks, iters, 4, out);
    run("1 chain, 8 operand pairs", k<1, 8>, blocks, iters / 8, 8, out);
    run("2 chains, 8 operand pairs (scoring pattern)", k<2, 8>, blocks, iters / 8, 16, out);
    run("4 chains, 8 operand pairs", k<4, 8>, blocks, iters / 8, 32, out);
  }
  return 0;
}
