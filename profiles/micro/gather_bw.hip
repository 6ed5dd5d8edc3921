// Micro-benchmark: what the memory system delivers for the CG kernels' access pattern -- random 512-byte factor rows
// (f = 128 fp32) gathered into registers in the quarter layout (each 16-lane group reads one 256-byte run per
// dwordx4 instruction), with almost no arithmetic.  Sets the achievable ceiling that roofline.frac is measured
// against the 8 TB/s spec.   Parameters: table rows, waves per SIMD (occupancy), loads in flight per lane.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_bw profiles/micro/gather_bw.hip && /tmp/gather_bw
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

// each wave: loop over its tiles; a tile = 4 * EQ rows (group g of the wave takes rows 4 q + g), EQ*2 dwordx4 per lane
template <int EQ, int BLOCK, int MINW>
__global__ __launch_bounds__(BLOCK, MINW) void gather_kernel(const float *__restrict__ Y, const int32_t *__restrict__ idx,
                                                             long n_tiles, float *__restrict__ out) {
  const int lane = threadIdx.x & 63, g = lane >> 4;
  const long wave = (blockIdx.x * (long)BLOCK + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * BLOCK) >> 6;
  float4 acc = {0.f, 0.f, 0.f, 0.f};
  int col_next[EQ];
  if (wave < n_tiles)
#pragma unroll
    for (int q = 0; q < EQ; ++q) col_next[q] = idx[wave * 4 * EQ + 4 * q + g];
  for (long t = wave; t < n_tiles; t += nwaves) {
    float4 v[EQ][2];
#pragma unroll
    for (int q = 0; q < EQ; ++q) {
      const float *src = Y + (size_t)col_next[q] * 128 + 4 * (lane & 15);
      v[q][0] = *reinterpret_cast<const float4 *>(src);
      v[q][1] = *reinterpret_cast<const float4 *>(src + 64);
    }
    const long tn = t + nwaves < n_tiles ? t + nwaves : t;
#pragma unroll
    for (int q = 0; q < EQ; ++q) col_next[q] = idx[tn * 4 * EQ + 4 * q + g];
#pragma unroll
    for (int q = 0; q < EQ; ++q) {
      acc.x += v[q][0].x + v[q][1].x;
      acc.y += v[q][0].y + v[q][1].y;
      acc.z += v[q][0].z + v[q][1].z;
      acc.w += v[q][0].w + v[q][1].w;
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

// plain streaming read of the same volume (float4 per lane, contiguous)
__global__ __launch_bounds__(256) void stream_kernel(const float4 *__restrict__ src, size_t n, float *__restrict__ out) {
  float4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = src[i];
    acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

template <int EQ, int BLOCK, int MINW>
static void run(const char *tag, const float *Y, const int32_t *idx, long n_rows_gathered, int blocks_per_cu, float *out) {
  const long n_tiles = n_rows_gathered / (4 * EQ);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int grid = 256 * blocks_per_cu;
  gather_kernel<EQ, BLOCK, MINW><<<grid, BLOCK>>>(Y, idx, n_tiles, out);
  hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) gather_kernel<EQ, BLOCK, MINW><<<grid, BLOCK>>>(Y, idx, n_tiles, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 3;
  printf("%-44s rows/tile=%2d waves/CU=%2d : %7.3f ms  %7.1f GB/s\n", tag, 4 * EQ, blocks_per_cu * BLOCK / 64, ms,
         n_tiles * 4.0 * EQ * 512 / (ms * 1e-3) / 1e9);
}

int main() {
  const long table_rows[2] = {292385, 4000000};  // 150 MB (C3 item factors: fits the 256 MB Infinity Cache) / 2 GB
  const long n_gather = 16L << 20;                // 16 M gathered rows = 8.6 GB
  float *out;
  hipMalloc(&out, 64);
  for (int ti = 0; ti < 2; ++ti) {
    const long R = table_rows[ti];
    float *Y;
    hipMalloc(&Y, (size_t)R * 512);
    hipMemset(Y, 0, (size_t)R * 512);
    for (int dist = 0; dist < 2; ++dist) {
      std::vector<int32_t> h(n_gather);
      std::mt19937_64 rng(1);
      std::uniform_real_distribution<double> u(0.0, 1.0);
      for (long i = 0; i < n_gather; ++i) {
        double r = u(rng);
        h[i] = (int32_t)std::min<long>(R - 1, (long)(R * (dist ? r * r * r : r)));  // uniform / power-law (gamma = 3) popularity
      }
      int32_t *idx;
      hipMalloc(&idx, n_gather * 4);
      hipMemcpy(idx, h.data(), n_gather * 4, hipMemcpyHostToDevice);
      printf("== table %ld rows (%.0f MB), %s column distribution\n", R, R * 512 / 1e6, dist ? "power-law(3)" : "uniform");
      run<8, 512, 4>("32-row tiles, 512-thread WG", Y, idx, n_gather, 2, out);
      run<8, 256, 4>("32-row tiles, 256-thread WG", Y, idx, n_gather, 4, out);
      run<4, 256, 8>("16-row tiles, 8 waves/SIMD", Y, idx, n_gather, 8, out);
      run<8, 256, 4>("32-row tiles, 2 waves/SIMD", Y, idx, n_gather, 2, out);
      run<4, 256, 8>("16-row tiles, 4 waves/SIMD", Y, idx, n_gather, 4, out);
      run<2, 256, 8>("8-row tiles, 8 waves/SIMD", Y, idx, n_gather, 8, out);
      hipFree(idx);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const size_t n4 = (size_t)R * 32;
    stream_kernel<<<2048, 256>>>((const float4 *)Y, n4, out);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) stream_kernel<<<2048, 256>>>((const float4 *)Y, n4, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("stream read of the table: %.3f ms  %.1f GB/s\n", ms / 5, n4 * 16.0 / (ms / 5 * 1e-3) / 1e9);
    hipFree(Y);
  }
  return 0;
}
