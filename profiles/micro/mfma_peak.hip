// Micro-benchmark: sustained rate of v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 with operands in registers (no memory
// traffic), N independent accumulators per wave, W waves per SIMD.  Gives the achievable fp32-MFMA ceiling (clock under
// load included) that the gramian and top-k GEMM kernels are measured against.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak profiles/micro/mfma_peak.hip && /tmp/mfma_peak
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma32(float *out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float a = a0 + threadIdx.x * 1e-6f, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void mfma16(float *out, int iters, float a0, float b0) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
  float a = a0 + threadIdx.x * 1e-6f, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int e = 0; e < 4; ++e) s += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
static double run(K kernel, int blocks, int iters, int nacc, double flop_per_mfma, float *out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  kernel<<<blocks, 256>>>(out, iters / 10, 1.f, 1.f);  // warm-up
  hipEventRecord(e0);
  kernel<<<blocks, 256>>>(out, iters, 1.f, 1.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  double flop = (double)blocks * 4 /*waves*/ * iters * nacc * flop_per_mfma;
  return flop / (ms * 1e-3) / 1e12;
}

int main() {
  float *out;
  hipMalloc(&out, 4096 * 256 * sizeof(float));
  const int iters = 20000;
  for (int blocks : {256, 512, 1024, 2048}) {
    printf("blocks=%4d (%.0f waves/SIMD)  32x32x2: 1 acc %.1f  2 acc %.1f  4 acc %.1f TFLOP/s | 16x16x4: 1 acc %.1f  4 acc %.1f TFLOP/s\n",
           blocks, blocks / 256.0, run(mfma32<1>, blocks, iters, 1, 4096.0, out), run(mfma32<2>, blocks, iters, 2, 4096.0, out),
           run(mfma32<4>, blocks, iters, 4, 4096.0, out), run(mfma16<1>, blocks, iters, 1, 2048.0, out),
           run(mfma16<4>, blocks, iters, 4, 2048.0, out));
  }
  return 0;
}
