// Which SIMD does wave w of a workgroup land on?  HW_REG_HW_ID (gfx9 family): wave_id [3:0], simd_id [5:4], cu_id [11:8].
// hipcc --offload-arch=gfx950 -O2 -o /tmp/wave_simd_map profiles/micro/wave_simd_map.hip && /tmp/wave_simd_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned *out) {
  const int wave = threadIdx.x >> 6;
  unsigned hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + wave] = hw;
  // stay resident a little so that several workgroups share a CU
  for (int i = 0; i < 2000; ++i) asm volatile("s_nop 15");
}
int main() {
  for (int block : {512, 768, 1024}) {
    const int waves = block / 64, grid = 1024;
    unsigned *d;
    hipMalloc(&d, sizeof(unsigned) * grid * waves);
    probe<<<grid, block>>>(d);
    std::vector<unsigned> h(grid * waves);
    hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
    // histogram of the simd pattern over the workgroups
    int same_mod4 = 0;
    printf("block %d: first 4 workgroups (wave: simd/wave_slot/cu):\n", block);
    for (int b = 0; b < grid; ++b) {
      bool ok = true;
      for (int w = 0; w < waves; ++w) {
        const unsigned hw = h[b * waves + w];
        const int simd = (hw >> 4) & 3;
        if (simd != ((((h[b * waves] >> 4) & 3) + w) & 3)) ok = false;
        if (b < 4) printf(" %d:%d/%d/%d", w, simd, hw & 15, (hw >> 8) & 15);
      }
      if (b < 4) printf("\n");
      same_mod4 += ok;
    }
    printf("block %d: %d of %d workgroups have simd(w) = (simd(0) + w) mod 4\n", block, same_mod4, grid);
    hipFree(d);
  }
  return 0;
}
