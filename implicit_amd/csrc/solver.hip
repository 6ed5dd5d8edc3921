// LeastSquaresSolver C-ABI (include/implicit_hip.h) + K7, the training-loss kernel.
//
// Replaces implicit/gpu/als.cu:118-281 (reference).  Argument validation mirrors als.cu:124,158-167
// (std::invalid_argument -> ValueError through the binding).
#include <functional>

#include "common.h"
#include "wave_ops.h"

namespace imp {

void gramian(const float *Y, long n_rows, int f, float reg, float *out);                                    // gramian.hip
void gramian_half(const void *Y, long n_rows, int f, float reg, float *out);                               // fp16 rows, fp32 products
bool cg_native_half(int f);                                                                                 // als_cg.hip
void least_squares_cg(const imp_csr *C, imp_matrix *X, const imp_matrix *YtY, const imp_matrix *Y, int cg_steps);  // als_cg.hip
int64_t least_squares_cholesky(const imp_csr *C, imp_matrix *X, const imp_matrix *YtY, const imp_matrix *Y, double reg);

// K7: loss numerator terms, one wavefront per user row (oracle: implicit/cpu/_als.pyx:257-308):
//   r = YtY x + sum_k ((c>0 ? -2c : 0) + (|c|-1) y.x) y ;  loss += r.x + sum |c| ; user_norm += x.x
// fp64 accumulation as in the oracle (:272); per-wave partials are added with fp64 atomics.
template <int VPL>
__global__ __launch_bounds__(256) void als_loss_kernel(const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                       const float *__restrict__ data, int users,
                                                       const float *__restrict__ X, const float *__restrict__ Y,
                                                       const float *__restrict__ YtY, int f, double *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  double loss = 0, conf_sum = 0, user_norm = 0;
  for (int u = wave; u < users; u += nwaves) {
    float x[VPL], r[VPL];
    load_row<VPL, false>(X + (size_t)u * f, f, lane, x);
#pragma unroll
    for (int v = 0; v < VPL; ++v) r[v] = 0.f;
    // r = YtY x (symmetric: column e of YtY == row e)
#pragma unroll
    for (int v = 0; v < VPL; ++v)
      for (int l = 0; l < 64; ++l) {
        int j = l + 64 * v;
        if (j >= f) break;
        float xj = lane_bcast(x[v], l);
        float row[VPL];
        load_row<VPL, false>(YtY + (size_t)j * f, f, lane, row);
#pragma unroll
        for (int w = 0; w < VPL; ++w) r[w] = fmaf(xj, row[w], r[w]);
      }
    const int b = indptr[u], e = indptr[u + 1];
    for (int k = b; k < e; ++k) {
      float y[VPL];
      load_row<VPL, false>(Y + (size_t)indices[k] * f, f, lane, y);
      float c = data[k];
      float t = c > 0.f ? -2.f * c : 0.f;
      float a = c > 0.f ? c : -c;
      float w = t + (a - 1.f) * wave_allsum(dot_local<VPL>(y, x));
#pragma unroll
      for (int v = 0; v < VPL; ++v) r[v] = fmaf(w, y[v], r[v]);
      conf_sum += a;
    }
    loss += (double)wave_allsum(dot_local<VPL>(r, x));
    user_norm += (double)wave_allsum(dot_local<VPL>(x, x));
  }
  if (lane == 0) {
    atomicAdd(&out[0], loss + conf_sum);
    atomicAdd(&out[1], conf_sum);
    atomicAdd(&out[2], user_norm);
  }
}

__global__ void sumsq_rows_kernel(const float *__restrict__ Y, size_t n, double *__restrict__ out) {
  double s = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float v = Y[i];
    s += (double)v * v;
  }
  // wave butterfly on doubles via two 32-bit halves is overkill here: LDS tree
  __shared__ double red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(out, red[0]);
}

template <int VPL>
static void launch_loss(const imp_csr *C, const float *X, const float *Y, const float *YtY, int f, double *buf) {
  int grid = std::min((C->rows + 3) / 4, ctx().num_cus * 8);
  als_loss_kernel<VPL><<<std::max(grid, 1), 256, 0, stream()>>>(C->indptr.data(), C->indices.data(), C->data.data(), C->rows, X, Y,
                                                               YtY, f, buf);
}

// rows of X <-> rows of one block of a (possibly multi-block) CSR matrix
template <typename Fn> static void for_each_part(const imp_csr *C, const imp_matrix *X, Fn &&fn) {
  if (C->parts.empty()) {
    fn(C, X);
    return;
  }
  for (size_t i = 0; i < C->parts.size(); ++i) {
    imp_matrix view = *X;  // row-range view sharing storage
    view.rows = (size_t)C->parts[i]->rows;
    view.data = reinterpret_cast<char *>(X->data) + (size_t)C->part_row0[i] * X->cols * X->itemsize;
    fn(C->parts[i].get(), &view);
  }
}

float calculate_loss(const imp_csr *C, const imp_matrix *X, const imp_matrix *Y, float reg) {
  const int f = (int)X->cols;
  if (f > 1024) throw std::invalid_argument("calculate_loss: factors must be <= 1024 (as the reference, implicit/gpu/als.cu:262-264)");
  auto &lossb = ctx().loss_buf;
  if (lossb.size < 4) lossb.alloc(4);
  double *g_loss_buf = lossb.data();
  IMP_CHECK_HIP(hipMemsetAsync(g_loss_buf, 0, 4 * sizeof(double), stream()));
  DeviceArray<float> yty;
  yty.alloc((size_t)f * f);
  gramian(Y->f32(), (long)Y->rows, f, 0.f, yty.data());
  {
    IMP_PROF("als_loss_rows");
    int vpl = (f + 63) / 64;
    for_each_part(C, X, [&](const imp_csr *part, const imp_matrix *x) {
      switch (vpl) {
        case 1: launch_loss<1>(part, x->f32(), Y->f32(), yty.data(), f, g_loss_buf); break;
        case 2: launch_loss<2>(part, x->f32(), Y->f32(), yty.data(), f, g_loss_buf); break;
        case 3: launch_loss<3>(part, x->f32(), Y->f32(), yty.data(), f, g_loss_buf); break;
        case 4: launch_loss<4>(part, x->f32(), Y->f32(), yty.data(), f, g_loss_buf); break;
        case 5: case 6: case 7: case 8: launch_loss<8>(part, x->f32(), Y->f32(), yty.data(), f, g_loss_buf); break;
        case 9: case 10: case 11: case 12: launch_loss<12>(part, x->f32(), Y->f32(), yty.data(), f, g_loss_buf); break;
        default: launch_loss<16>(part, x->f32(), Y->f32(), yty.data(), f, g_loss_buf); break;
      }
      IMP_CHECK_HIP(hipGetLastError());
    });
    size_t n = Y->rows * Y->cols;
    if (n) {
      int grid = (int)std::min<size_t>((n + 255) / 256, (size_t)ctx().num_cus * 4);
      sumsq_rows_kernel<<<grid, 256, 0, stream()>>>(Y->f32(), n, g_loss_buf + 3);
      IMP_CHECK_HIP(hipGetLastError());
    }
  }
  double h[4];
  IMP_CHECK_HIP(hipMemcpyAsync(h, g_loss_buf, sizeof(h), hipMemcpyDeviceToHost, stream()));
  sync();
  double loss = h[0] + (double)reg * (h[3] + h[2]);
  double denom = h[1] + (double)C->rows * (double)C->cols - (double)C->nnz;
  return (float)(loss / denom);
}

static void check_solver_args(const imp_csr *C, const imp_matrix *X, const imp_matrix *YtY, const imp_matrix *Y) {
  if (X->cols != Y->cols) throw std::invalid_argument("X and Y should have the same number of columns");
  if (X->cols != YtY->cols) throw std::invalid_argument("Columns of X don't match number of columns of YtY");
  if (YtY->rows != YtY->cols) throw std::invalid_argument("YtY must be square");
  // as als.cu:162-165: Cui may be SMALLER than the factor matrices (the reference's own partial_fit_items passes a row
  // whose column count predates users added since, tests/als_test.py:273-301); only the first Cui.rows rows of X are solved
  if ((size_t)C->rows > X->rows) throw std::invalid_argument("Dimensionality mismatch between rows of Cui and rows of X");
  if ((size_t)C->cols > Y->rows) throw std::invalid_argument("Dimensionality mismatch between cols of Cui and rows of Y");
  if (X->itemsize != Y->itemsize) throw std::invalid_argument("X and Y should have the same dtype");
  if (YtY->itemsize != 4) throw std::invalid_argument("YtY must be float32");
}

}  // namespace imp

using namespace imp;

extern "C" int imp_matrix_astype(const imp_matrix *src, size_t itemsize, imp_matrix **out);

struct imp_solver {
  int dummy = 0;
};

extern "C" {

int imp_solver_create(imp_solver **out) {
  return guarded([&] {
    (void)ctx();
    *out = new imp_solver();
  });
}

int imp_solver_destroy(imp_solver *s) {
  return guarded([&] { delete s; });
}

int imp_solver_calculate_yty(imp_solver *, const imp_matrix *Y, imp_matrix *YtY, float regularization) {
  return guarded([&] {
    if (YtY->cols != Y->cols) throw std::invalid_argument("YtY and Y should have the same number of columns");
    if (YtY->rows != YtY->cols) throw std::invalid_argument("YtY must be square");
    if (YtY->itemsize != 4) throw std::invalid_argument("YtY must be float32");
    if (Y->itemsize == 4) gramian(Y->f32(), (long)Y->rows, (int)Y->cols, regularization, YtY->f32());
    else gramian_half(Y->data, (long)Y->rows, (int)Y->cols, regularization, YtY->f32());  // converted in registers
    sync_call();
  });
}

static void run_with_f32(const imp_csr *cui, imp_matrix *X, const imp_matrix *Y,
                         const std::function<void(imp_matrix *, const imp_matrix *)> &body) {
  if (X->itemsize == 4) {
    body(X, Y);
    return;
  }
  // fp16 storage: up-convert, solve in fp32, down-convert X back in place
  imp_matrix *x32 = nullptr, *y32 = nullptr, *x16 = nullptr;
  if (imp_matrix_astype(X, 4, &x32) != IMP_OK) throw std::runtime_error(imp_last_error());
  std::unique_ptr<imp_matrix> gx(x32);
  if (imp_matrix_astype(Y, 4, &y32) != IMP_OK) throw std::runtime_error(imp_last_error());
  std::unique_ptr<imp_matrix> gy(y32);
  body(x32, y32);
  if (imp_matrix_astype(x32, 2, &x16) != IMP_OK) throw std::runtime_error(imp_last_error());
  std::unique_ptr<imp_matrix> g16(x16);
  IMP_CHECK_HIP(hipMemcpyAsync(X->data, x16->data, X->bytes(), hipMemcpyDeviceToDevice, stream()));
  sync();
}

int imp_solver_least_squares(imp_solver *, const imp_csr *cui, imp_matrix *X, const imp_matrix *YtY, const imp_matrix *Y,
                             int cg_steps) {
  return guarded([&] {
    check_solver_args(cui, X, YtY, Y);
    if (cg_steps < 0) throw std::invalid_argument("cg_steps must be >= 0");
    auto body = [&](imp_matrix *x, const imp_matrix *y) {
      for_each_part(cui, x, [&](const imp_csr *part, const imp_matrix *xp) {
        least_squares_cg(part, const_cast<imp_matrix *>(xp), YtY, y, cg_steps);
      });
      if (ctx().deferred) return;
      sync();
    };
    // fp16 factor storage: the f = 64 / 128 kernels load and store it directly (half the gather bytes, fp32 arithmetic, as
    // als.cu:41,55,109); other factor counts go through an fp32 copy
    if (X->itemsize == 2 && cg_native_half((int)X->cols)) body(X, Y);
    else run_with_f32(cui, X, Y, body);
  });
}

int imp_solver_least_squares_cholesky(imp_solver *, const imp_csr *cui, imp_matrix *X, const imp_matrix *YtY,
                                      const imp_matrix *Y, double regularization, int64_t *failed_row) {
  return guarded([&] {
    check_solver_args(cui, X, YtY, Y);
    note_device_write(X->data, X->bytes());  // the rows this call solves
    int64_t failed = -1;
    run_with_f32(cui, X, Y, [&](imp_matrix *x, const imp_matrix *y) {
      size_t block = 0;
      for_each_part(cui, x, [&](const imp_csr *part, const imp_matrix *xp) {
        const int64_t bad = least_squares_cholesky(part, const_cast<imp_matrix *>(xp), YtY, y, regularization);
        if (bad >= 0 && failed < 0) failed = bad + (cui->parts.empty() ? 0 : cui->part_row0[block]);
        ++block;
      });
    });
    if (failed_row) *failed_row = failed;
    if (failed >= 0)
      throw std::invalid_argument("cholesky factorisation failed on row " + std::to_string(failed) +
                                  ". Try increasing the regularization parameter.");
  });
}

int imp_solver_calculate_loss(imp_solver *, const imp_csr *cui, const imp_matrix *X, const imp_matrix *Y, float regularization,
                              float *loss_out) {
  return guarded([&] {
    if (X->cols != Y->cols) throw std::invalid_argument("X and Y should have the same number of columns");
    if ((size_t)cui->rows != X->rows) throw std::invalid_argument("Dimensionality mismatch between rows of Cui and rows of X");
    if ((size_t)cui->cols != Y->rows) throw std::invalid_argument("Dimensionality mismatch between cols of Cui and rows of Y");
    if (X->itemsize == 4 && Y->itemsize == 4) {
      *loss_out = calculate_loss(cui, X, Y, regularization);
    } else {
      imp_matrix *x32 = nullptr, *y32 = nullptr;
      if (imp_matrix_astype(X, 4, &x32) != IMP_OK) throw std::runtime_error(imp_last_error());
      std::unique_ptr<imp_matrix> gx(x32);
      if (imp_matrix_astype(Y, 4, &y32) != IMP_OK) throw std::runtime_error(imp_last_error());
      std::unique_ptr<imp_matrix> gy(y32);
      *loss_out = calculate_loss(cui, x32, y32, regularization);
    }
  });
}

}  // extern "C"
