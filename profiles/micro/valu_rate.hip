// Issue rates of the vector FMA forms the CG kernels are built from (round 3): v_fma_f32, v_pk_fma_f32, v_fmac_f32 with a
// DPP operand, ds_read_b128 latency.  One workgroup of W waves per CU, ITER x 64 independent instructions per wave.
// build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// MODE 0-5: see main(); 6: v_fmac_f32 (VOP2, two vector sources + accumulator); 7: v_fmac_f32 with a scalar multiplicand;
// 8: v_mul_f32 (two vector sources); 9: v_mov_b32.  `where` (one word per workgroup) receives HW_ID | XCC_ID << 16 so that the
// host can count the CUs the grid really ran on (a rate per SIMD means nothing if two workgroups shared a CU and another idled).
template <int MODE> __global__ __launch_bounds__(1024) void rate_kernel(float *out, int iters, unsigned *where) {
  if (threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    where[blockIdx.x] = (hw & 0xff00u) | ((xcc & 0xfu) << 16);  // cu_id 11:8, sh_id 12, se_id 15:13
  }
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
    } else if constexpr (MODE == 6) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b));
    } else if constexpr (MODE == 7) {
      float sc = __builtin_amdgcn_readfirstlane(c);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(sc), "v"(b));
    } else if constexpr (MODE == 8) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
    } else if constexpr (MODE == 9) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(a[i + 1]));
          asm volatile("v_mov_b32 %0, %1" : "+v"(a[i + 1]) : "v"(b));
        }
    } else if constexpr (MODE == 10) {  // round 4: f16 operand widened inside the FMA (the packed-half tile idea)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(c), "v"(b));
    } else if constexpr (MODE == 13) {  // the HIGH half as the f16 operand
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(c), "v"(b));
    } else if constexpr (MODE == 14) {  // dependent chain of fma_mix: latency
#pragma unroll
      for (int r = 0; r < 64; ++r) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(a[0]) : "v"(c), "v"(b));
    } else if constexpr (MODE == 15) {  // fma_mix and pk_fma alternating, independent
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(c), "v"(b));
          asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(cc), "v"(bb));
        }
    } else if constexpr (MODE == 16) {  // two chains of 4 dependent fma_mix per "dot", as the packed-tile kernel issues them
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        float s0 = a[0], s1 = a[1];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(s0) : "v"(a[2 + h]), "v"(b));
          asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(s1) : "v"(a[2 + h]), "v"(c));
        }
        a[0] = s0, a[1] = s1;
      }
    } else if constexpr (MODE == 11) {  // v_cvt_f32_f16
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_cvt_f32_f16 %0, %1" : "+v"(a[i]) : "v"(c));
    } else if constexpr (MODE == 12) {  // v_dot2c_f32_f16: two f16 x f16 products accumulated in fp32
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b));
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

template <int MODE> int run(const char *name, int waves_per_cu, int instr_per_iter, double flops_per_instr_lane, int grid_mult = 1) {
  float *out;
  unsigned *where;
  const int cus = 256 * grid_mult, iters = 20000 / grid_mult;
  CHECK(hipMalloc(&out, (size_t)cus * 1024 * 4));
  CHECK(hipMalloc(&where, (size_t)cus * 4));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  rate_kernel<MODE><<<cus, waves_per_cu * 64>>>(out, 100, where);
  CHECK(hipEventRecord(e0));
  rate_kernel<MODE><<<cus, waves_per_cu * 64>>>(out, iters, where);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned> h(cus);
  CHECK(hipMemcpy(h.data(), where, (size_t)cus * 4, hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  const int distinct = (int)(std::unique(h.begin(), h.end()) - h.begin());
  double instr = (double)cus * waves_per_cu * iters * instr_per_iter;
  // per SIMD of the CUs that were really used (`distinct`), not of the 256 the grid was sized for
  double cycles_per_instr_per_simd = ms * 1e-3 * 2.4e9 / (instr / (distinct * 4.0));
  printf("%-28s %4d wgs x %2d waves on %3d distinct CUs: %.3f ms, %.2f cycles per wave-instruction per SIMD (at 2.4 GHz), %.1f TFLOP/s\n", name,
         cus, waves_per_cu, distinct, ms, cycles_per_instr_per_simd, instr * 64 * flops_per_instr_lane / (ms * 1e-3) / 1e12);
  CHECK(hipFree(out));
  CHECK(hipFree(where));
  return 0;
}

int main() {
  for (int w : {4, 8, 16}) {
    run<0>("v_fma_f32 (independent)", w, 64, 2);
    run<1>("v_pk_fma_f32 (independent)", w, 64, 4);
    run<2>("v_fmac_f32_dpp newbcast", w, 64, 2);
    run<5>("pk + plain alternating", w, 64, 3);
  }
  // saturating grids: 8 workgroups per CU's worth of 4-wave workgroups (8 waves per SIMD whatever the placement)
  run<0>("v_fma_f32 (independent)", 4, 64, 2, 8);
  run<1>("v_pk_fma_f32 (independent)", 4, 64, 4, 8);
  run<6>("v_fmac_f32 vop2 v,v", 4, 64, 2, 8);
  run<7>("v_fmac_f32 vop2 s,v", 4, 64, 2, 8);
  run<8>("v_mul_f32 v,v", 4, 64, 1, 8);
  run<9>("v_mov_b32", 4, 64, 0, 8);
  run<2>("v_fmac_f32_dpp newbcast", 4, 64, 2, 8);
  run<10>("v_fma_mix_f32 f16,f32,f32", 4, 64, 2, 8);
  run<13>("v_fma_mix_f32 hi half", 4, 64, 2, 8);
  run<15>("fma_mix + pk_fma alternating", 4, 64, 3, 8);
  run<16>("fma_mix 2 chains of 4", 4, 64, 2, 8);
  run<16>("fma_mix 2 chains of 4, 1 wave/SIMD", 4, 64, 2);
  run<14>("v_fma_mix_f32 dependent", 4, 64, 2);
  run<11>("v_cvt_f32_f16", 4, 64, 0, 8);
  run<12>("v_dot2c_f32_f16", 4, 64, 4, 8);
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
