// K8: RandomState -- uniform / normal factor initialisation on the device.
//
// Replaces implicit/gpu/random.cu:14-40 (cuRAND default generator + Thrust affine rescale) with a
// counter-based Philox4x32-10 kernel: element i of draw d is a pure function of (seed, d, i), so the
// result does not depend on the launch geometry.  PARITY UNPINNED by design: neither cuRAND's stream
// nor the CPU path's numpy Generator is reproduced (no reference test depends on specific random
// values); parity runs inject identical initial factors from the host instead.
#include "common.h"

namespace imp {

struct u32x4 {
  uint32_t x, y, z, w;
};

__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
    uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0, c1 = n1, c2 = n2, c3 = n3;
    k0 += W0, k1 += W1;
  }
  return {c0, c1, c2, c3};
}

__device__ __forceinline__ float u01(uint32_t x) { return ((x >> 8) + 0.5f) * (1.0f / 16777216.0f); }  // (0,1)

__global__ void uniform_kernel(float *__restrict__ out, size_t n, uint64_t seed, uint32_t draw, float low, float high) {
  size_t quads = (n + 3) / 4;
  for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < quads; q += (size_t)gridDim.x * blockDim.x) {
    u32x4 r = philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), draw, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
    float v[4] = {u01(r.x), u01(r.y), u01(r.z), u01(r.w)};
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (4 * q + j < n) out[4 * q + j] = low + (high - low) * v[j];
  }
}

__global__ void normal_kernel(float *__restrict__ out, size_t n, uint64_t seed, uint32_t draw, float mean, float stddev) {
  size_t quads = (n + 3) / 4;
  for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < quads; q += (size_t)gridDim.x * blockDim.x) {
    u32x4 r = philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), draw, 1u, (uint32_t)seed, (uint32_t)(seed >> 32));
    float u[4] = {u01(r.x), u01(r.y), u01(r.z), u01(r.w)};
    float v[4];
    float m0 = sqrtf(-2.f * logf(u[0])), m1 = sqrtf(-2.f * logf(u[2]));
    v[0] = m0 * cosf(6.2831853071795865f * u[1]);
    v[1] = m0 * sinf(6.2831853071795865f * u[1]);
    v[2] = m1 * cosf(6.2831853071795865f * u[3]);
    v[3] = m1 * sinf(6.2831853071795865f * u[3]);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (4 * q + j < n) out[4 * q + j] = mean + stddev * v[j];
  }
}

}  // namespace imp

using namespace imp;

imp_matrix *imp_internal_new_matrix(size_t rows, size_t cols, size_t itemsize, bool zero);

struct imp_random {
  uint64_t seed = 42;
  uint32_t draws = 0;
};

extern "C" {

int imp_random_create(int64_t seed, imp_random **out) {
  return guarded([&] {
    (void)ctx();
    auto r = new imp_random();
    r->seed = (uint64_t)seed;
    *out = r;
  });
}
int imp_random_destroy(imp_random *r) {
  return guarded([&] { delete r; });
}

int imp_random_uniform(imp_random *r, size_t rows, size_t cols, float low, float high, imp_matrix **out) {
  return guarded([&] {
    std::unique_ptr<imp_matrix> m(imp_internal_new_matrix(rows, cols, 4, false));
    size_t n = rows * cols;
    if (n) {
      IMP_PROF("rng_uniform");
      int grid = (int)std::min<size_t>((n / 4 + 255) / 256 + 1, (size_t)ctx().num_cus * 8);
      uniform_kernel<<<grid, 256, 0, stream()>>>(m->f32(), n, r->seed, r->draws++, low, high);
      IMP_CHECK_HIP(hipGetLastError());
    }
    sync();
    *out = m.release();
  });
}

int imp_random_randn(imp_random *r, size_t rows, size_t cols, float mean, float stddev, imp_matrix **out) {
  return guarded([&] {
    std::unique_ptr<imp_matrix> m(imp_internal_new_matrix(rows, cols, 4, false));
    size_t n = rows * cols;
    if (n) {
      IMP_PROF("rng_normal");
      int grid = (int)std::min<size_t>((n / 4 + 255) / 256 + 1, (size_t)ctx().num_cus * 8);
      normal_kernel<<<grid, 256, 0, stream()>>>(m->f32(), n, r->seed, r->draws++, mean, stddev);
      IMP_CHECK_HIP(hipGetLastError());
    }
    sync();
    *out = m.release();
  });
}

}  // extern "C"
