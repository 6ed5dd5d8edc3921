// Issue rates of the vector FMA forms the CG kernels are built from (round 3): v_fma_f32, v_pk_fma_f32, v_fmac_f32 with a
// DPP operand, ds_read_b128 latency.  One workgroup of W waves per CU, ITER x 64 independent instructions per wave.
// build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE> __global__ __launch_bounds__(1024) void rate_kernel(float *out, int iters) {
  float a[16], b = threadIdx.x * 1e-9f, c = 1.0f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p[8];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x + i;
#pragma unroll
  for (int i = 0; i < 8; ++i) p[i] = f2{a[2 * i], a[2 * i + 1]};
  f2 bb = {b, b}, cc = {c, c};
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b));
    } else if constexpr (MODE == 1) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(cc), "v"(bb));
    } else if constexpr (MODE == 2) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b), "v"(c));
    } else if constexpr (MODE == 3) {  // dependent chain of pk_fma: latency
#pragma unroll
      for (int r = 0; r < 64; ++r) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[0]) : "v"(cc), "v"(bb));
    } else if constexpr (MODE == 4) {  // dependent chain of v_fma
#pragma unroll
      for (int r = 0; r < 64; ++r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(c), "v"(b));
    } else if constexpr (MODE == 5) {  // alternating pk / plain, independent
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(cc), "v"(bb));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b));
        }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(64) void lds_latency_kernel(unsigned long long *out, int n) {
  __shared__ __attribute__((aligned(16))) float buf[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) buf[i] = (float)((i * 4 + 16) % 4096);
  __syncthreads();
  int idx = threadIdx.x * 4;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
    float4 v = *reinterpret_cast<float4 *>(&buf[idx & 4092]);
    idx = (int)v.x;  // dependent read
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[0] = t1 - t0, out[1] = idx;
}

template <int MODE> int run(const char *name, int waves_per_cu, int instr_per_iter, double flops_per_instr_lane) {
  float *out;
  const int cus = 256, iters = 20000;
  CHECK(hipMalloc(&out, (size_t)cus * 1024 * 4));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  rate_kernel<MODE><<<cus, waves_per_cu * 64>>>(out, 100);
  CHECK(hipEventRecord(e0));
  rate_kernel<MODE><<<cus, waves_per_cu * 64>>>(out, iters);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  double instr = (double)cus * waves_per_cu * iters * instr_per_iter;
  double cycles_per_instr_per_simd = ms * 1e-3 * 2.4e9 / (instr / (cus * 4.0));
  printf("%-28s waves/CU %2d: %.3f ms, %.2f cycles per wave-instruction per SIMD (at 2.4 GHz), %.1f TFLOP/s\n", name, waves_per_cu, ms,
         cycles_per_instr_per_simd, instr * 64 * flops_per_instr_lane / (ms * 1e-3) / 1e12);
  CHECK(hipFree(out));
  return 0;
}

int main() {
  for (int w : {4, 8, 16}) {
    run<0>("v_fma_f32 (independent)", w, 64, 2);
    run<1>("v_pk_fma_f32 (independent)", w, 64, 4);
    run<2>("v_fmac_f32_dpp newbcast", w, 64, 2);
    run<5>("pk + plain alternating", w, 64, 3);
  }
  run<3>("v_pk_fma_f32 dependent", 4, 64, 4);
  run<4>("v_fma_f32 dependent", 4, 64, 2);
  unsigned long long *o;
  CHECK(hipMalloc(&o, 16));
  lds_latency_kernel<<<1, 64>>>(o, 10000);
  unsigned long long h[2];
  CHECK(hipMemcpy(h, o, 16, hipMemcpyDeviceToHost));
  printf("ds_read_b128 dependent chain: %.1f s_memtime ticks per read (100 MHz ticks x 24 = shader cycles if memtime is the 100 MHz counter)\n", h[0] / 10000.0);
  return 0;
}
