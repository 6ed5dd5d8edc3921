// Micro-probe for v_mfma_f32_4x4x1_16b_f32 (the 4-column-granular fp32 MFMA used by the batched CG dense product):
//   1. operand / result lane maps, measured with one-hot A operands (incl. the cbsz / abid broadcast controls);
//   2. sustained issue rate with N independent accumulators;
//   3. whether VALU FMAs of the SAME wave overlap with its MFMAs (interleaved stream).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma4x4_probe profiles/micro/mfma4x4_probe.hip && /tmp/mfma4x4_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// block la: A is one-hot at lane la, B[l] = l + 1  ->  D[lane][r] = lb + 1 where (la, lb) feeds output (lane, r)
template <int CBSZ, int ABID, int BLGP> __global__ void layout_kernel(float *out) {
  const int lane = threadIdx.x, la = blockIdx.x;
  float a = lane == la ? 1.f : 0.f, b = (float)(lane + 1);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc, CBSZ, ABID, BLGP);
  for (int r = 0; r < 4; ++r) out[(la * 64 + lane) * 4 + r] = acc[r];
}

template <int CBSZ, int ABID, int BLGP> static void layout(const char *tag) {
  float *d;
  hipMalloc(&d, 64 * 64 * 4 * sizeof(float));
  layout_kernel<CBSZ, ABID, BLGP><<<64, 64>>>(d);
  std::vector<float> h(64 * 64 * 4);
  hipMemcpy(h.data(), d, h.size() * sizeof(float), hipMemcpyDeviceToHost);
  printf("== layout cbsz=%d abid=%d blgp=%d (%s): out(lane, reg) <- A lane, B lane\n", CBSZ, ABID, BLGP, tag);
  for (int lane = 0; lane < 64; ++lane) {
    printf("lane %2d:", lane);
    for (int r = 0; r < 4; ++r) {
      int found = 0;
      for (int la = 0; la < 64; ++la) {
        float v = h[(la * 64 + lane) * 4 + r];
        if (v != 0.f) {
          printf("  r%d<-(A%2d,B%2d)", r, la, (int)v - 1);
          ++found;
        }
      }
      if (!found) printf("  r%d<-none", r);
    }
    printf("\n");
  }
  hipFree(d);
}

template <int NACC, int NVALU> __global__ __launch_bounds__(256) void rate_kernel(float *out, int iters, float a0, float b0) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float v[NVALU > 0 ? NVALU : 1];
  for (int i = 0; i < NVALU; ++i) v[i] = a0 * i;
  float a = a0 + threadIdx.x * 1e-6f, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 4, 0, 0);
#pragma unroll
      for (int j = 0; j < NVALU / NACC; ++j) v[i * (NVALU / NACC) + j] = fmaf(v[i * (NVALU / NACC) + j], b, a);
    }
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int e = 0; e < 4; ++e) s += acc[i][e];
  for (int i = 0; i < NVALU; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int NVALU> static void rate(int waves_per_simd) {
  float *d;
  int blocks = 256 * waves_per_simd;  // 256-thread blocks = 4 waves = one per SIMD
  hipMalloc(&d, (size_t)blocks * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000;
  rate_kernel<NACC, NVALU><<<blocks, 256>>>(d, iters / 10, 1.f, 1.f);
  hipEventRecord(e0);
  rate_kernel<NACC, NVALU><<<blocks, 256>>>(d, iters, 1.f, 1.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  // cycles per MFMA per SIMD assuming 2.4 GHz: waves_per_simd waves share one SIMD
  double mfma_per_simd = (double)waves_per_simd * iters * NACC;
  double cyc = ms * 1e-3 * 2.4e9 / mfma_per_simd;
  double tf = (double)blocks * 4 * iters * NACC * 512.0 / (ms * 1e-3) / 1e12;
  printf("rate: NACC=%d NVALU/iter=%d waves/SIMD=%d : %.3f ms, %.1f cyc(2.4GHz)/MFMA/SIMD, %.1f TF (mfma only), valu/mfma=%.1f\n", NACC,
         NVALU, waves_per_simd, ms, cyc, tf, (double)NVALU / NACC);
  hipFree(d);
}

int main() {
  layout<0, 0, 0>("plain: 16 independent 4x4x1 blocks");
  layout<4, 0, 0>("A of block 0 broadcast to all 16 blocks");
  layout<4, 5, 0>("A of block 5 broadcast to all 16 blocks");
  layout<2, 1, 0>("A of block 1 of each 4-block group broadcast inside the group");
  rate<1, 0>(1);
  rate<4, 0>(1);
  rate<8, 0>(1);
  rate<4, 0>(4);
  rate<8, 0>(4);
  rate<8, 8>(1);
  rate<8, 16>(1);
  rate<8, 32>(1);
  rate<8, 16>(4);
  rate<8, 32>(4);
  rate<8, 64>(4);
  return 0;
}
