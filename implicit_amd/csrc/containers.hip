// Device containers behind the C-ABI: Matrix / Vector<int> / CSRMatrix / COOMatrix, the library
// context (one HIP stream per device), error state and the launch profiler.
//
// Replaces implicit/gpu/matrix.cu (reference) -- plain hipMalloc-backed storage instead of RMM
// buffers and managed memory; the Thrust lambdas (copy_rowids, assign_rows, convert_array) and
// calculate_norms_kernel become the small wave64 kernels below.
#include <hip/hip_fp16.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>

#include "common.h"

namespace imp {

static thread_local std::string g_last_error;
void set_last_error(const std::string &msg) { g_last_error = msg; }

static std::mutex g_ctx_mutex;
// leaked on purpose: the contexts own device buffers, and static destructors run after the HIP runtime has shut down
static auto &g_contexts = *new std::map<int, std::unique_ptr<Context>>;

Context &ctx() {
  int dev = 0;
  IMP_CHECK_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(g_ctx_mutex);
  auto it = g_contexts.find(dev);
  if (it == g_contexts.end()) {
    auto c = std::make_unique<Context>();
    c->device = dev;
    IMP_CHECK_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    hipDeviceProp_t prop;
    IMP_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    c->num_cus = prop.multiProcessorCount;
    if (const char *e = getenv("IMP_OVERSUB")) c->oversub = std::max(1, atoi(e));
    it = g_contexts.emplace(dev, std::move(c)).first;
  }
  return *it->second;
}

static auto &g_derived = *new std::vector<DerivedCache *>;  // (leaked like the contexts; guarded by g_ctx_mutex)
void register_derived_cache(DerivedCache *c) {
  std::lock_guard<std::mutex> lock(g_ctx_mutex);
  g_derived.push_back(c);
}
void unregister_derived_cache(DerivedCache *c) {
  std::lock_guard<std::mutex> lock(g_ctx_mutex);
  g_derived.erase(std::remove(g_derived.begin(), g_derived.end(), c), g_derived.end());
}

void note_device_write(const void *dst, size_t bytes) {
  const char *a = static_cast<const char *>(dst);
  std::lock_guard<std::mutex> lock(g_ctx_mutex);
  for (DerivedCache *d : g_derived) {
    const char *b = static_cast<const char *>(d->src);
    if (b && a < b + d->bytes && b < a + bytes) d->src = nullptr;
  }
  for (auto &kv : g_contexts) {
    Context &c = *kv.second;
    if (!c.pad_y_src) continue;
    const char *b = static_cast<const char *>(c.pad_y_src);
    const size_t b_bytes = c.pad_y_rows * (size_t)c.pad_y_f * sizeof(float);
    if (a < b + b_bytes && b < a + bytes) c.pad_y_src = nullptr;
  }
}

std::unique_lock<std::recursive_mutex> lock_device() {
  try {
    return std::unique_lock<std::recursive_mutex>(ctx().mutex);
  } catch (...) {
    return std::unique_lock<std::recursive_mutex>();  // no usable device: the body reports the error itself
  }
}

void sync() { IMP_CHECK_HIP(hipStreamSynchronize(stream())); }
void sync_call() {
  if (!ctx().deferred) sync();
}

// ---- profiler -----------------------------------------------------------------------------------
struct ProfEntry {
  double total_ms = 0;
  int64_t launches = 0;
};
struct ProfPending {
  const char *name;
  hipEvent_t start, stop;
};
static std::recursive_mutex g_prof_mutex;  // the profiler tables are process-wide; per-device call locks do not cover them
static bool g_prof_on = false;
static std::string g_prof_filter;
static std::map<std::string, ProfEntry> g_prof;
static std::vector<ProfPending> g_prof_pending;
static std::vector<hipEvent_t> g_event_pool;

bool prof_enabled() { return g_prof_on; }

static hipEvent_t get_event() {
  if (!g_event_pool.empty()) {
    hipEvent_t e = g_event_pool.back();
    g_event_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  IMP_CHECK_HIP(hipEventCreate(&e));
  return e;
}

static void prof_account(const ProfPending &p) {
  float ms = 0;
  if (hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess) {
    auto &e = g_prof[p.name];
    e.total_ms += ms;
    e.launches += 1;
  }
  g_event_pool.push_back(p.start);
  g_event_pool.push_back(p.stop);
}

// Recycle the event pairs whose kernels have finished (no waiting): creating fresh events for every launch of a long
// timed loop costs far more stream time than recording them.
static void prof_collect_finished() {
  size_t done = 0;
  while (done < g_prof_pending.size() && hipEventQuery(g_prof_pending[done].stop) == hipSuccess) prof_account(g_prof_pending[done++]);
  if (done) g_prof_pending.erase(g_prof_pending.begin(), g_prof_pending.begin() + (long)done);
}

ProfScope::ProfScope(const char *n) : name(n) {
  if (!g_prof_on) return;
  std::lock_guard<std::recursive_mutex> plock(g_prof_mutex);
  if (!g_prof_filter.empty() && !strstr(n, g_prof_filter.c_str())) return;
  if (g_event_pool.size() < 2 && g_prof_pending.size() >= 32) prof_collect_finished();
  start = get_event();
  stop = get_event();
  IMP_CHECK_HIP(hipEventRecord(start, stream()));
}

ProfScope::~ProfScope() {
  if (!start) return;
  std::lock_guard<std::recursive_mutex> plock(g_prof_mutex);
  (void)hipEventRecord(stop, stream());
  g_prof_pending.push_back({name, start, stop});
}

static void prof_flush() {
  std::lock_guard<std::recursive_mutex> plock(g_prof_mutex);
  if (g_prof_pending.empty()) return;
  sync();
  for (auto &p : g_prof_pending) prof_account(p);
  g_prof_pending.clear();
}

// ---- storage ------------------------------------------------------------------------------------
// Recycled small blocks (<= 4 MB, at most 32 per size class and device: a worst case of ~250 MB) are only a cache: an allocation
// that fails gives them all back to the runtime and tries once more.  Safe to hand out without a fence because every library
// stream is joined to the ONE library stream before an entry point returns (class streams, exchange stream, occupy stream),
// and the next user of a block queues its work behind that stream.
static void flush_small_blocks() {
  // only the CALLING thread's device: its lists are what a failed allocation on this device can get back, and it is the one
  // device this thread may synchronise before the blocks go (kernels still in flight on it may touch a recycled block; another
  // device's lists are left to that device's own out-of-memory path)
  std::vector<void *> blocks;
  Context &c = ctx();
  {
    std::lock_guard<std::mutex> g(c.small_mutex);
    for (auto &list : c.small_free) {
      blocks.insert(blocks.end(), list.begin(), list.end());
      list.clear();
    }
  }
  if (!blocks.empty()) (void)hipDeviceSynchronize();
  for (void *b : blocks) (void)hipFree(b);
}
static void device_malloc(void **out, size_t bytes) {
  hipError_t e = hipMalloc(out, bytes);
  if (e == hipErrorOutOfMemory) {
    (void)hipGetLastError();
    flush_small_blocks();
    e = hipMalloc(out, bytes);
  }
  IMP_CHECK_HIP(e);
}

Storage::Storage(size_t bytes_, bool zero) : bytes(bytes_) {
  if (bytes) {
    if (bytes <= kSmallMax) {
      size_class = 8;
      while (((size_t)1 << size_class) < bytes) ++size_class;
      home = &ctx();
      {
        std::lock_guard<std::mutex> g(home->small_mutex);
        auto &list = home->small_free[size_class];
        if (!list.empty()) {
          ptr = list.back();
          list.pop_back();
        }
      }
      if (!ptr) device_malloc(&ptr, (size_t)1 << size_class);
    } else {
      device_malloc(&ptr, bytes);
    }
    if (zero) IMP_CHECK_HIP(hipMemsetAsync(ptr, 0, bytes, stream()));
  }
}
Storage::~Storage() {
  if (!owned || !ptr) return;
  note_device_write(ptr, bytes);  // the memory is about to be re-used for something else
  if (home) {
    std::lock_guard<std::mutex> g(home->small_mutex);
    auto &list = home->small_free[size_class];
    if (list.size() < 32) {
      list.push_back(ptr);
      return;
    }
  }
  (void)hipFree(ptr);
}

// ---- small kernels ------------------------------------------------------------------------------
template <typename T>
__global__ void gather_rows_kernel(const T *__restrict__ src, const int32_t *__restrict__ rowids,
                                   T *__restrict__ dst, size_t nrows, size_t cols) {
  size_t total = nrows * cols;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    size_t r = i / cols, c = i - r * cols;
    dst[i] = src[(size_t)rowids[r] * cols + c];
  }
}

template <typename T>
__global__ void scatter_rows_kernel(T *__restrict__ dst, const int32_t *__restrict__ rowids,
                                    const T *__restrict__ src, size_t nrows, size_t cols) {
  size_t total = nrows * cols;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    size_t r = i / cols, c = i - r * cols;
    dst[(size_t)rowids[r] * cols + c] = src[i];
  }
}

__global__ void cast_f32_f16_kernel(const float *__restrict__ src, __half *__restrict__ dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = __float2half(src[i]);
}
__global__ void cast_f16_f32_kernel(const __half *__restrict__ src, float *__restrict__ dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = __half2float(src[i]);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// one wavefront per row: coalesced strided read, butterfly sum, zero norms -> 1e-10
// (matrix.cu:173-188; CPU semantics cpu/matrix_factorization_base.py:233-247)
template <typename T>
__global__ void row_norms_kernel(const T *__restrict__ data, float *__restrict__ out, size_t rows, size_t cols) {
  size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
  size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
  int lane = threadIdx.x & 63;
  for (size_t r = wave; r < rows; r += nwaves) {
    float s = 0.f;
    for (size_t c = lane; c < cols; c += 64) {
      float v = (float)data[r * cols + c];
      s += v * v;
    }
    s = wave_sum(s);
    if (lane == 0) {
      float n = sqrtf(s);
      out[r] = n == 0.f ? 1e-10f : n;
    }
  }
}

static int grid_for(size_t work_items, int block = 256) {
  size_t blocks = (work_items + block - 1) / block;
  size_t cap = (size_t)ctx().num_cus * 8;
  return (int)std::max<size_t>(1, std::min(blocks, cap));
}

// stand-in for a resident collective kernel (imp_debug_occupy): 256 threads and 32 KB of LDS per workgroup, spinning on the
// constant-rate wall clock
__global__ __launch_bounds__(256) void occupy_kernel(long long ticks) {
  extern __shared__ float pad[];
  pad[threadIdx.x] = 0.f;
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

// imp_debug_core_clock: one wavefront counts core-clock cycles (s_memtime) over a stretch of the constant-rate wall clock
__global__ void core_clock_kernel(long long ticks, unsigned long long *out) {
  const long long w0 = wall_clock64();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  long long w1 = w0;
  while (w1 - w0 < ticks) {
    __builtin_amdgcn_s_sleep(8);
    w1 = wall_clock64();
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[0] = c1 - c0, out[1] = (unsigned long long)(w1 - w0);
}

static imp_matrix *new_matrix(size_t rows, size_t cols, size_t itemsize, bool zero) {
  if (itemsize != 4 && itemsize != 2) throw std::invalid_argument("invalid itemsize for Matrix (must be 2 or 4)");
  auto m = std::make_unique<imp_matrix>();
  m->rows = rows;
  m->cols = cols;
  m->itemsize = itemsize;
  m->storage = std::make_shared<Storage>(rows * cols * itemsize, zero);
  m->data = m->storage->ptr;
  return m.release();
}

}  // namespace imp

using namespace imp;

// make the helper visible to the other translation units
imp_matrix *imp_internal_new_matrix(size_t rows, size_t cols, size_t itemsize, bool zero) {
  return new_matrix(rows, cols, itemsize, zero);
}

extern "C" {

const char *imp_last_error(void) { return g_last_error.c_str(); }

const char *imp_version(void) { return "implicit_hip 0.1 (gfx950)"; }

int imp_get_device_count(int *count) {
  return guarded([&] {
    int c = 0;
    IMP_CHECK_HIP(hipGetDeviceCount(&c));
    if (c <= 0) throw std::runtime_error("no HIP device available");
    *count = c;
  });
}

int imp_set_device(int device) {
  return guarded([&] { IMP_CHECK_HIP(hipSetDevice(device)); });
}
int imp_set_oversubscribe(int factor) {
  return guarded([&] {
    if (factor < 1 || factor > 64) throw std::invalid_argument("oversubscription factor must be in [1, 64]");
    ctx().oversub = factor;
  });
}
int imp_get_oversubscribe(int *factor) {
  return guarded([&] { *factor = ctx().oversub; });
}
int imp_set_deferred_sync(int on) {
  return guarded([&] {
    if (!on) {
      sync();  // leaving the mode: everything queued so far is complete on return
    }
    ctx().deferred = on != 0;
  });
}
int imp_debug_occupy(int workgroups, int microseconds) {
  return guarded([&] {
    if (workgroups <= 0 || microseconds <= 0) return;
    auto &c = ctx();
    if (!c.occupy_stream) IMP_CHECK_HIP(hipStreamCreateWithFlags(&c.occupy_stream, hipStreamNonBlocking));
    int rate_khz = 100000;  // wall_clock64 ticks per millisecond
    IMP_CHECK_HIP(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, c.device));
    const long long ticks = (long long)microseconds * rate_khz / 1000;
    occupy_kernel<<<workgroups, 256, 32 * 1024, c.occupy_stream>>>(ticks);
    IMP_CHECK_HIP(hipGetLastError());
  });
}
int imp_debug_core_clock(int microseconds, double *mhz) {
  return guarded([&] {
    if (!mhz) throw std::invalid_argument("imp_debug_core_clock: null output");
    auto &c = ctx();
    int rate_khz = 100000;
    IMP_CHECK_HIP(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, c.device));
    DeviceArray<unsigned long long> out;
    out.alloc(2, microseconds >= 0);  // (beside: no fill -- it would be queued BEHIND the work on the library stream; the probe writes both words)
    // microseconds < 0: the probe runs on the side stream BESIDE whatever is queued on the library stream (deferred mode: the
    // clock the kernels of a queued step run at, not the clock the chip recovers to behind them)
    const bool beside = microseconds < 0;
    if (beside && !c.occupy_stream) IMP_CHECK_HIP(hipStreamCreateWithFlags(&c.occupy_stream, hipStreamNonBlocking));
    hipStream_t st = beside ? c.occupy_stream : stream();
    core_clock_kernel<<<1, 64, 0, st>>>((long long)std::max(1, std::abs(microseconds)) * rate_khz / 1000, out.data());
    IMP_CHECK_HIP(hipGetLastError());
    unsigned long long h[2] = {0, 0};
    IMP_CHECK_HIP(hipMemcpyAsync(h, out.data(), sizeof(h), hipMemcpyDeviceToHost, st));
    if (beside) IMP_CHECK_HIP(hipStreamSynchronize(st));
    else sync();
    *mhz = h[1] ? (double)h[0] / ((double)h[1] / rate_khz * 1e3) : 0.0;  // cycles per microsecond
  });
}
int imp_get_device(int *device) {
  return guarded([&] { IMP_CHECK_HIP(hipGetDevice(device)); });
}
int imp_solver_fixup_rows(unsigned long long *count, int reset) {
  return guarded([&] {
    sync();  // the counter is written by kernels on the library stream
    unsigned long long *total = fixup_total();
    if (count) *count = *total;
    if (reset) *total = 0ull;
  });
}
int imp_device_synchronize(void) {
  return guarded([&] {
    sync();
    IMP_CHECK_HIP(hipDeviceSynchronize());
  });
}
int imp_release_workspaces(void) {
  return guarded([&] {
    sync();
    auto &c = ctx();
    // scratch the solver paths grow on demand and otherwise keep for the life of the process (a 10 M-row replica padded from
    // f = 100 to 128 is 5 GB): dropped here, re-allocated by the next call that needs them
    c.gram_ws = {};
    c.long_ws = {};
    c.pad_x = {};
    c.pad_y = {};
    c.pad_y_src = nullptr;
    c.pad_gram = {};
    c.nm_fix_rows = {};
    c.w256_ws = {};
    std::lock_guard<std::mutex> g(c.small_mutex);
    for (auto &list : c.small_free) {
      for (void *p : list) (void)hipFree(p);
      list.clear();
    }
  });
}
int imp_mem_get_info(size_t *free_bytes, size_t *total_bytes) {
  return guarded([&] { IMP_CHECK_HIP(hipMemGetInfo(free_bytes, total_bytes)); });
}

// ---- host-side helper: parallel CSR transpose ------------------------------------------------------
// fit() needs both orientations of the confidence matrix (the reference transposes on the host with scipy, implicit/gpu/als.py:121:
// single-threaded, 0.11 s for configs[2]'s 17 M nonzeros -- a third of this path's whole set-up).  Stable counting
// transpose over T threads: thread t owns a run of rows balanced by nonzeros and counts its entries per column; the
// column offsets of thread t are the column's start plus the counts of the threads before it; every thread then scatters
// its rows in order -- rows stay sorted inside every output row, bit-identical to scipy's result on canonical input.
int imp_host_csr_transpose(int32_t rows, int32_t cols, int64_t nnz, const int32_t *indptr, const int32_t *indices, const float *data,
                           int32_t *t_indptr, int32_t *t_indices, float *t_data, int threads) {
  return guarded_host([&] {
    if (rows < 0 || cols < 0 || nnz < 0 || nnz > INT32_MAX) throw std::invalid_argument("host_csr_transpose: sizes out of range");
    if (indptr[0] != 0 || indptr[rows] != nnz) throw std::invalid_argument("host_csr_transpose: indptr does not span the nonzeros");
    // per-thread histograms cost T x cols counters: keep them under 256 MB
    int T = std::max(1, std::min(threads > 0 ? threads : (int)std::thread::hardware_concurrency() / 2, 32));
    while (T > 1 && (size_t)T * (size_t)cols > ((size_t)64 << 20)) --T;
    if (nnz < (1 << 16)) T = 1;
    std::vector<int32_t> cut(T + 1, rows);
    cut[0] = 0;
    for (int t = 1; t < T; ++t) {
      const int64_t target = nnz * t / T;
      cut[t] = (int32_t)(std::lower_bound(indptr, indptr + rows + 1, (int32_t)target) - indptr);
      cut[t] = std::min(rows, std::max(cut[t], cut[t - 1]));
    }
    // a decreasing indptr would send the scatter below out of bounds: checked before anything is indexed through it
    for (int32_t r = 0; r < rows; ++r)
      if (indptr[r + 1] < indptr[r]) throw std::invalid_argument("host_csr_transpose: indptr must be non-decreasing (row " + std::to_string(r) + ")");
    std::vector<std::vector<int32_t>> hist(T);
    for (auto &h : hist) h.assign((size_t)cols, 0);  // allocated HERE: a bad_alloc inside a worker thread would terminate the process
    std::string error;
    std::mutex error_mutex;
    auto parallel = [&](auto &&fn) {
      // an exception escaping a std::thread body calls std::terminate, and so does destroying a joinable thread: every
      // body is fenced, every started thread joined, and the first failure is re-thrown on the calling thread
      auto fenced = [&](int t) {
        try {
          fn(t);
        } catch (const std::exception &e) {
          std::lock_guard<std::mutex> g(error_mutex);
          if (error.empty()) error = std::string("host_csr_transpose: ") + e.what();
        } catch (...) {
          std::lock_guard<std::mutex> g(error_mutex);
          if (error.empty()) error = "host_csr_transpose: unknown failure in a worker thread";
        }
      };
      std::vector<std::thread> pool;
      pool.reserve((size_t)T);
      int started = 1;
      try {
        for (; started < T; ++started) pool.emplace_back(fenced, started);
      } catch (...) {  // the system refused another thread: the caller's thread does the remaining shares
      }
      fenced(0);
      for (int t = started; t < T; ++t) fenced(t);
      for (auto &th : pool) th.join();
      if (!error.empty()) throw std::invalid_argument(error);
    };
    parallel([&](int t) {
      auto &h = hist[t];
      for (int64_t k = indptr[cut[t]]; k < indptr[cut[t + 1]]; ++k) {
        const int32_t c = indices[k];
        if (c < 0 || c >= cols) throw std::invalid_argument("column index out of range");
        ++h[c];
      }
    });
    // column totals -> t_indptr; then every thread's histogram becomes its first output slot per column
    parallel([&](int t) {  // columns split evenly over the threads
      const int32_t c0 = (int32_t)((int64_t)cols * t / T), c1 = (int32_t)((int64_t)cols * (t + 1) / T);
      for (int32_t c = c0; c < c1; ++c) {
        int32_t sum = 0;
        for (int u = 0; u < T; ++u) sum += hist[u][c];
        t_indptr[c + 1] = sum;
      }
    });
    t_indptr[0] = 0;
    for (int32_t c = 0; c < cols; ++c) t_indptr[c + 1] += t_indptr[c];
    parallel([&](int t) {
      const int32_t c0 = (int32_t)((int64_t)cols * t / T), c1 = (int32_t)((int64_t)cols * (t + 1) / T);
      for (int32_t c = c0; c < c1; ++c) {
        int32_t at = t_indptr[c];
        for (int u = 0; u < T; ++u) {
          const int32_t n = hist[u][c];
          hist[u][c] = at;
          at += n;
        }
      }
    });
    parallel([&](int t) {
      auto &pos = hist[t];
      for (int32_t r = cut[t]; r < cut[t + 1]; ++r)
        for (int32_t k = indptr[r]; k < indptr[r + 1]; ++k) {
          const int32_t at = pos[indices[k]]++;
          t_indices[at] = r;
          t_data[at] = data[k];
        }
    });
  });
}

// ---- Matrix -------------------------------------------------------------------------------------
int imp_matrix_create(size_t rows, size_t cols, const void *host_data, size_t itemsize, imp_matrix **out) {
  return guarded([&] {
    imp_matrix *m = new_matrix(rows, cols, itemsize, host_data == nullptr);
    std::unique_ptr<imp_matrix> guard(m);
    if (host_data && m->bytes()) {
      IMP_CHECK_HIP(hipMemcpyAsync(m->data, host_data, m->bytes(), hipMemcpyHostToDevice, stream()));
    }
    sync();
    *out = guard.release();
  });
}

int imp_matrix_wrap_device(size_t rows, size_t cols, void *device_ptr, size_t itemsize, imp_matrix **out) {
  return guarded([&] {
    if (itemsize != 4 && itemsize != 2) throw std::invalid_argument("invalid itemsize for Matrix (must be 2 or 4)");
    auto m = std::make_unique<imp_matrix>();
    m->rows = rows;
    m->cols = cols;
    m->itemsize = itemsize;
    m->storage = std::make_shared<Storage>(device_ptr);
    m->data = device_ptr;
    *out = m.release();
  });
}

int imp_matrix_row(const imp_matrix *src, size_t rowid, imp_matrix **out) {
  return guarded([&] {
    if (rowid >= src->rows) throw std::invalid_argument("row index out of bounds for matrix");
    auto m = std::make_unique<imp_matrix>(*src);
    m->rows = 1;
    m->data = reinterpret_cast<char *>(src->data) + rowid * src->cols * src->itemsize;
    *out = m.release();
  });
}

int imp_matrix_slice(const imp_matrix *src, size_t start, size_t end, imp_matrix **out) {
  return guarded([&] {
    if (end < start) throw std::invalid_argument("end_rowid < start_rowid for matrix slice");
    if (end > src->rows) throw std::invalid_argument("row index out of bounds for matrix");
    auto m = std::make_unique<imp_matrix>(*src);
    m->rows = end - start;
    m->data = reinterpret_cast<char *>(src->data) + start * src->cols * src->itemsize;
    *out = m.release();
  });
}

int imp_matrix_gather(const imp_matrix *src, const imp_intvector *rowids, imp_matrix **out) {
  return guarded([&] {
    std::unique_ptr<imp_matrix> m(new_matrix(rowids->size, src->cols, src->itemsize, false));
    size_t total = m->rows * m->cols;
    if (total) {
      IMP_PROF("gather_rows");
      if (src->itemsize == 4)
        gather_rows_kernel<float><<<grid_for(total), 256, 0, stream()>>>(
            (const float *)src->data, rowids->v.data(), (float *)m->data, m->rows, m->cols);
      else
        gather_rows_kernel<__half><<<grid_for(total), 256, 0, stream()>>>(
            (const __half *)src->data, rowids->v.data(), (__half *)m->data, m->rows, m->cols);
      IMP_CHECK_HIP(hipGetLastError());
    }
    sync();
    *out = m.release();
  });
}

int imp_matrix_resize(imp_matrix *m, size_t rows, size_t cols) {
  return guarded([&] {
    if (cols != m->cols) throw std::logic_error("changing number of columns in Matrix::resize is not implemented yet");
    if (rows < m->rows) throw std::logic_error("reducing number of rows in Matrix::resize is not implemented yet");
    auto storage = std::make_shared<Storage>(rows * cols * m->itemsize, true);
    if (m->bytes()) IMP_CHECK_HIP(hipMemcpyAsync(storage->ptr, m->data, m->bytes(), hipMemcpyDeviceToDevice, stream()));
    sync();
    m->storage = storage;
    m->data = storage->ptr;
    m->rows = rows;
  });
}

int imp_matrix_assign_rows(imp_matrix *m, const imp_intvector *rowids, const imp_matrix *other) {
  return guarded([&] {
    if (other->cols != m->cols) throw std::invalid_argument("column dimension mismatch for Matrix::assign_rows");
    if (other->rows != rowids->size) throw std::invalid_argument("row dimension mismatch for Matrix::assign_rows");
    // the reference scatters fp32 only (matrix.cu:133-134); fp16 -> fp16 is accepted here as well, so that a half-precision
    // model's partial_fit_* stays on the device
    if (other->itemsize != m->itemsize) throw std::invalid_argument("dtype mismatch for Matrix::assign_rows");
    size_t total = other->rows * other->cols;
    note_device_write(m->data, m->bytes());
    if (total) {
      IMP_PROF("scatter_rows");
      if (m->itemsize == 4)
        scatter_rows_kernel<float><<<grid_for(total), 256, 0, stream()>>>(m->f32(), rowids->v.data(), other->f32(), other->rows,
                                                                          other->cols);
      else
        scatter_rows_kernel<__half><<<grid_for(total), 256, 0, stream()>>>((__half *)m->data, rowids->v.data(),
                                                                           (const __half *)other->data, other->rows, other->cols);
      IMP_CHECK_HIP(hipGetLastError());
    }
    sync();
  });
}

int imp_matrix_astype(const imp_matrix *src, size_t itemsize, imp_matrix **out) {
  return guarded([&] {
    std::unique_ptr<imp_matrix> m(new_matrix(src->rows, src->cols, itemsize, false));
    size_t n = src->rows * src->cols;
    if (n) {
      if (itemsize == src->itemsize) {
        IMP_CHECK_HIP(hipMemcpyAsync(m->data, src->data, src->bytes(), hipMemcpyDeviceToDevice, stream()));
      } else if (itemsize == 2) {
        IMP_PROF("cast_f32_f16");
        cast_f32_f16_kernel<<<grid_for(n), 256, 0, stream()>>>((const float *)src->data, (__half *)m->data, n);
      } else {
        IMP_PROF("cast_f16_f32");
        cast_f16_f32_kernel<<<grid_for(n), 256, 0, stream()>>>((const __half *)src->data, (float *)m->data, n);
      }
      IMP_CHECK_HIP(hipGetLastError());
    }
    sync();
    *out = m.release();
  });
}

int imp_matrix_calculate_norms(const imp_matrix *src, imp_matrix **out) {
  return guarded([&] {
    std::unique_ptr<imp_matrix> m(new_matrix(1, src->rows, 4, false));
    if (src->rows) {
      IMP_PROF("row_norms");
      int grid = grid_for(src->rows * 64);
      if (src->itemsize == 4)
        row_norms_kernel<float><<<grid, 256, 0, stream()>>>((const float *)src->data, (float *)m->data, src->rows, src->cols);
      else
        row_norms_kernel<__half><<<grid, 256, 0, stream()>>>((const __half *)src->data, (float *)m->data, src->rows, src->cols);
      IMP_CHECK_HIP(hipGetLastError());
    }
    sync();
    *out = m.release();
  });
}

int imp_matrix_to_host(const imp_matrix *m, void *host_out) {
  return guarded([&] {
    if (m->bytes()) IMP_CHECK_HIP(hipMemcpyAsync(host_out, m->data, m->bytes(), hipMemcpyDeviceToHost, stream()));
    sync();
  });
}

int imp_matrix_from_host(imp_matrix *m, const void *host_in) {
  return guarded([&] {
    note_device_write(m->data, m->bytes());
    if (m->bytes()) IMP_CHECK_HIP(hipMemcpyAsync(m->data, host_in, m->bytes(), hipMemcpyHostToDevice, stream()));
    sync();
  });
}

int imp_matrix_copy_rows(imp_matrix *dst, size_t dst_row, const imp_matrix *src, size_t src_row, size_t rows) {
  return guarded([&] {
    if (dst->cols != src->cols || dst->itemsize != src->itemsize)
      throw std::invalid_argument("copy_rows: the two matrices must have rows of the same width and itemsize");
    if (dst_row + rows > dst->rows || src_row + rows > src->rows) throw imp::out_of_range_error("copy_rows: row range out of bounds");
    const size_t row_bytes = dst->cols * dst->itemsize;
    note_device_write(static_cast<char *>(dst->data) + dst_row * row_bytes, rows * row_bytes);
    if (rows && row_bytes)
      IMP_CHECK_HIP(hipMemcpyAsync(static_cast<char *>(dst->data) + dst_row * row_bytes,
                                   static_cast<const char *>(src->data) + src_row * row_bytes, rows * row_bytes,
                                   hipMemcpyDeviceToDevice, stream()));
    sync_call();  // queue-only in deferred mode, like the exchange it stands in for
  });
}

int imp_matrix_shape(const imp_matrix *m, size_t *rows, size_t *cols, size_t *itemsize) {
  return guarded([&] {
    if (rows) *rows = m->rows;
    if (cols) *cols = m->cols;
    if (itemsize) *itemsize = m->itemsize;
  });
}

int imp_matrix_device_ptr(const imp_matrix *m, void **ptr) {
  return guarded([&] {
    *ptr = m->data;
    if (m->storage) m->storage->exposed = true;  // whoever holds the address may write through it: nothing derived from this memory is cached
  });
}

int imp_matrix_destroy(imp_matrix *m) {
  return guarded([&] { delete m; });
}

// ---- Vector<int> --------------------------------------------------------------------------------
int imp_intvector_create(const int32_t *host_data, size_t size, imp_intvector **out) {
  return guarded([&] {
    auto v = std::make_unique<imp_intvector>();
    v->size = size;
    v->v.upload(host_data, size);
    sync();
    *out = v.release();
  });
}
int imp_intvector_destroy(imp_intvector *v) {
  return guarded([&] { delete v; });
}

// ---- CSR / COO -----------------------------------------------------------------------------------
int imp_csr_create(int32_t rows, int32_t cols, int64_t nnz, const int32_t *indptr, const int32_t *indices,
                   const float *data, imp_csr **out) {
  return guarded([&] {
    constexpr bool timing = false;  // (debug: where the construction time goes, on stderr)
    auto t_mark = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
      if (!timing) return;
      sync();
      auto now = std::chrono::steady_clock::now();
      fprintf(stderr, "[csr-timing] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_mark).count());
      t_mark = now;
    };
    if (rows < 0 || cols < 0 || nnz < 0) throw std::invalid_argument("negative dimension for CSRMatrix");
    if (nnz > INT32_MAX) throw std::invalid_argument("CSRMatrix with more than 2^31-1 nonzeros is not supported");
    if (rows && indptr[rows] != nnz) throw std::invalid_argument("indptr[rows] != nonzeros for CSRMatrix");
    if (rows && indptr[0] != 0) throw std::invalid_argument("indptr[0] != 0 for CSRMatrix");
    // a malformed matrix would turn into out-of-bounds gathers on the device: indptr must not decrease, column ids
    // must lie in [0, cols) (scipy's check_format(full_check=True) conditions; one pass over the host arrays)
    for (int32_t r = 0; r < rows; ++r)
      if (indptr[r + 1] < indptr[r]) throw std::invalid_argument("indptr must be non-decreasing for CSRMatrix (row " + std::to_string(r) + ")");
    {
      int32_t lo = 0, hi = -1;
      for (int64_t k = 0; k < nnz; ++k) {
        lo = std::min(lo, indices[k]);
        hi = std::max(hi, indices[k]);
      }
      if (nnz && (lo < 0 || hi >= cols))
        throw std::invalid_argument("column index out of range for CSRMatrix (" + std::to_string(lo < 0 ? lo : hi) + " not in [0, " +
                                    std::to_string(cols) + "))");
    }
    lap("validation");
    auto m = std::make_unique<imp_csr>();
    m->rows = rows;
    m->cols = cols;
    m->nnz = nnz;
    m->indptr.upload(indptr, (size_t)rows + 1);
    m->indices.upload(indices, (size_t)nnz);
    m->data.upload(data, (size_t)nnz);
    lap("upload indptr/indices/data");

    // Row schedule: counting sort of row ids by descending length, cut into length classes.
    int32_t segment = imp_csr::kSegment;
    if (const char *e = getenv("IMP_SEGMENT")) segment = std::max(32, atoi(e));
    int32_t max_len = 0;
    for (int32_t r = 0; r < rows; ++r) max_len = std::max(max_len, indptr[r + 1] - indptr[r]);
    m->max_row = max_len;
    std::vector<int32_t> count((size_t)max_len + 2, 0);
    for (int32_t r = 0; r < rows; ++r) count[indptr[r + 1] - indptr[r]]++;
    std::vector<int32_t> start((size_t)max_len + 2, 0);
    int32_t acc = 0;
    int32_t class_count[imp_csr::kBins] = {0};
    for (int32_t len = max_len; len >= 0; --len) {
      start[len] = acc;
      acc += count[len];
      int b = 0;
      while (len <= imp_csr::kClassMax[b + 1]) ++b;  // kClassMax[b+1] < len <= kClassMax[b]
      class_count[b] += count[len];
    }
    std::vector<int32_t> order((size_t)rows);
    for (int32_t r = 0; r < rows; ++r) order[start[indptr[r + 1] - indptr[r]]++] = r;
    m->order.upload(order.data(), order.size());
    lap("row schedule (counting sort)");
    m->bin_start[0] = 0;
    for (int b = 0; b < imp_csr::kBins; ++b) m->bin_start[b + 1] = m->bin_start[b] + class_count[b];
    const int32_t n_long = class_count[0];

    // long rows -> segments.
    //
    // Plain plan: consecutive runs of <= kSegment nonzeros.  Striped plan: the long rows of a popular-item side gather
    // the SAME factor rows over and over (C3 item side: 3.3 M long-row nonzeros over 359 K columns), but a plain
    // segment spans far more of the factor matrix than an L2 holds, so every pass streams them from the
    // Infinity Cache / HBM again.  If the rows are column-sorted and the re-use is >= 4, rows are cut at multiples of
    // `stripe` columns instead; the stripes are dealt to the 8 XCDs (greedy by weight) and each XCD's workgroups
    // (blockIdx % 8) sweep their stripes one after the other, so that the 2 MB of factor rows a stripe covers are
    // fetched into that XCD's L2 and hit by every long row (measured: partial kernel 2.6x faster with fully
    // L2-resident gathers; 1.3x with the real plan, whose segments are short).  Segments stay in row-major order (the combine kernel sums a row's partials in that fixed
    // order); `seg_exec` is the execution order.
    // width: 12288 columns (6 MB of factor rows at f = 128) was the best of {2048 .. 32768} on C3 (359 K columns; narrower
    // = more, shorter segments), 6144 the best of {2048 .. 12288} on the ml-20m shape (138 K columns: the 8 XCDs need
    // enough stripes to balance) -- hence about 24 stripes, between 4096 and 12288 columns
    int32_t stripe = std::min(12288, std::max(4096, (cols / 24 + 1023) / 1024 * 1024));
    if (const char *e = getenv("IMP_STRIPE")) stripe = std::max(0, atoi(e));
    double stripe_reuse = 4.0;  // minimum gathers per column of the gathered matrix for the striped plan
    auto build_plan = [&](int32_t n_plan, LongPlan &lp, double stripe_reuse, int32_t segment) {
      int64_t long_nnz = 0;
      bool sorted = true;
      for (int32_t li = 0; li < n_plan; ++li) {
        const int32_t r = order[li];
        long_nnz += indptr[r + 1] - indptr[r];
        if (stripe > 0 && sorted) sorted = std::is_sorted(indices + indptr[r], indices + indptr[r + 1]);
      }
      // ... and only when a row still leaves >= 32 nonzeros per stripe on average: with many more stripes than that (configs[3]'s
      // item side: 10 M columns = 814 stripes under rows of ~800 nonzeros) the cut would produce one- and two-entry segments,
      // hundreds of millions of them (23 s of plan building before this rule)
      const int64_t n_stripes_all = stripe > 0 ? ((int64_t)cols + stripe - 1) / stripe : 1;
      const bool striped = stripe > 0 && sorted && n_plan > 0 && (double)long_nnz >= stripe_reuse * (double)cols &&
                           (double)long_nnz >= 32.0 * (double)n_stripes_all * (double)n_plan;
      std::vector<int32_t> row_seg((size_t)n_plan + 1, 0), seg_row, seg_begin, seg_end, seg_stripe;
      for (int32_t li = 0; li < n_plan; ++li) {
        const int32_t r = order[li];
        row_seg[li] = (int32_t)seg_row.size();
        int32_t pos = indptr[r];
        const int32_t row_end = indptr[r + 1];
        while (pos < row_end) {
          int32_t hi = row_end, st = 0;
          if (striped) {
            st = indices[pos] / stripe;
            const int64_t bound = ((int64_t)st + 1) * stripe;
            hi = (int32_t)(std::lower_bound(indices + pos, indices + row_end, bound,
                                            [](int32_t c, int64_t b) { return (int64_t)c < b; }) -
                           indices);
          }
          for (int32_t b = pos; b < hi; b += segment) {
            seg_row.push_back(li);
            seg_begin.push_back(b);
            seg_end.push_back(std::min(hi, b + segment));
            seg_stripe.push_back(st);
          }
          pos = hi;
        }
      }
      row_seg[n_plan] = (int32_t)seg_row.size();
      const int32_t n_seg = (int32_t)seg_row.size();
      std::vector<int32_t> seg_exec((size_t)n_seg);
      if (striped) {
        const int32_t n_stripes = (cols + stripe - 1) / stripe;
        std::vector<int64_t> weight((size_t)n_stripes, 0);
        for (int32_t s = 0; s < n_seg; ++s) weight[seg_stripe[s]] += seg_end[s] - seg_begin[s] + 16;  // + per-segment overhead
        std::vector<int32_t> by_weight((size_t)n_stripes);
        for (int32_t i = 0; i < n_stripes; ++i) by_weight[i] = i;
        std::stable_sort(by_weight.begin(), by_weight.end(), [&](int32_t a, int32_t b) { return weight[a] > weight[b]; });
        int64_t load[8] = {0};
        std::vector<int32_t> stripe_xcd((size_t)n_stripes, 0), stripe_rank((size_t)n_stripes, 0);
        int32_t per_xcd[8] = {0};
        for (int32_t st : by_weight) {  // heaviest first onto the least loaded XCD
          int x = (int)(std::min_element(load, load + 8) - load);
          load[x] += weight[st];
          stripe_xcd[st] = x;
          stripe_rank[st] = per_xcd[x]++;
        }
        std::vector<int32_t> ids((size_t)n_seg);
        for (int32_t s = 0; s < n_seg; ++s) ids[s] = s;
        std::stable_sort(ids.begin(), ids.end(), [&](int32_t a, int32_t b) {
          const int32_t sa = seg_stripe[a], sb = seg_stripe[b];
          if (stripe_xcd[sa] != stripe_xcd[sb]) return stripe_xcd[sa] < stripe_xcd[sb];
          return stripe_rank[sa] < stripe_rank[sb];  // equal stripe: ascending segment id = ascending row
        });
        seg_exec = ids;
        int32_t posx = 0;
        for (int x = 0; x < 8; ++x) {
          lp.xcd_start[x] = posx;
          while (posx < n_seg && stripe_xcd[seg_stripe[seg_exec[posx]]] == x) ++posx;
        }
        lp.xcd_start[8] = n_seg;
        lp.stripe = stripe;
      } else {
        // plain plan: runs of 4 consecutive segments dealt round-robin to the XCDs (neighbouring segments of a row,
        // i.e. neighbouring column ranges, stay on one XCD)
        int32_t posx = 0;
        for (int x = 0; x < 8; ++x) {
          lp.xcd_start[x] = posx;
          for (int32_t s = 0; s < n_seg; ++s)
            if ((s / 4) % 8 == x) seg_exec[posx++] = s;
        }
        lp.xcd_start[8] = n_seg;
      }
      lp.n_long = n_plan;
      lp.n_seg = n_seg;
      lp.seg_exec.upload(seg_exec.data(), seg_exec.size());
      lp.row_seg.upload(row_seg.data(), row_seg.size());
      lp.seg_row.upload(seg_row.data(), seg_row.size());
      lp.seg_begin.upload(seg_begin.data(), seg_begin.size());
      lp.seg_end.upload(seg_end.data(), seg_end.size());
      sync();  // the uploads read pageable host vectors that die with this scope
    };
    build_plan(n_long, m->plan_all, stripe_reuse, segment);
    lap("long-row plan (all)");
    // rows of more than kCholLongRow nonzeros: the first n_chol_long entries of `order` (sorted by descending length)
    {
      int32_t longer = 0;
      for (int32_t len = max_len; len > imp_csr::kCholLongRow; --len) longer += count[len];
      m->n_chol_long = longer;
    }
    build_plan(m->n_chol_long, m->plan_chol, 1e30, imp_csr::kCholSegment);  // never striped
    {
      int64_t long_nnz = 0;
      for (int32_t li = 0; li < n_long; ++li) long_nnz += indptr[order[li] + 1] - indptr[order[li]];
      int32_t seg = 2048;
      while (seg < 16384 && (int64_t)seg * ctx().num_cus * 8 < long_nnz) seg *= 2;
      if (const char *e = getenv("IMP_NM_SEGMENT")) seg = std::max(64, atoi(e));
      m->nm_segment = seg;
      for (int32_t li = 0; li < n_long; ++li) {
        const int32_t len = indptr[order[li] + 1] - indptr[order[li]];
        if (len <= seg) break;  // descending lengths
        m->nm_multi_rows++;
        m->nm_multi_segs += (len + seg - 1) / seg;
      }
      build_plan(n_long, m->plan_nm, 1e30, seg);  // never striped: a row's segments are consecutive runs
    }
    sync();
    lap("long-row plans (xl, chol, nm)");
    *out = m.release();
  });
}

// int64 row offsets (scipy switches indptr AND indices to int64 once nnz >= 2^31; the CPU reference accepts both widths,
// _als.pyx:76).  Up to `part_limit` nonzeros the matrix is one block; beyond that it is cut into consecutive row blocks of
// at most that many nonzeros each -- the kernels keep their 32-bit offsets.  IMP_CSR_PART_NNZ lowers the limit (tests).
int imp_csr_create64(int32_t rows, int32_t cols, int64_t nnz, const int64_t *indptr, const int32_t *indices, const float *data,
                     imp_csr **out) {
  return guarded([&] {
    if (rows < 0 || cols < 0 || nnz < 0) throw std::invalid_argument("negative dimension for CSRMatrix");
    if (rows && indptr[rows] != nnz) throw std::invalid_argument("indptr[rows] != nonzeros for CSRMatrix");
    if (rows && indptr[0] != 0) throw std::invalid_argument("indptr[0] != 0 for CSRMatrix");
    for (int32_t r = 0; r < rows; ++r)
      if (indptr[r + 1] < indptr[r]) throw std::invalid_argument("indptr must be non-decreasing for CSRMatrix (row " + std::to_string(r) + ")");
    int64_t part_limit = INT32_MAX;
    if (const char *e = getenv("IMP_CSR_PART_NNZ")) part_limit = std::max<int64_t>(1, std::min<int64_t>(INT32_MAX, atoll(e)));
    auto build = [&](int32_t r0, int32_t r1) {
      std::vector<int32_t> local((size_t)(r1 - r0) + 1);
      const int64_t base = indptr[r0];
      for (int32_t r = r0; r <= r1; ++r) local[r - r0] = (int32_t)(indptr[r] - base);
      imp_csr *part = nullptr;
      if (imp_csr_create(r1 - r0, cols, indptr[r1] - base, local.data(), indices + base, data + base, &part) != IMP_OK)
        throw std::invalid_argument(imp_last_error());
      return std::unique_ptr<imp_csr>(part);
    };
    if (nnz <= part_limit) {
      *out = build(0, rows).release();
      return;
    }
    auto top = std::make_unique<imp_csr>();
    top->rows = rows, top->cols = cols, top->nnz = nnz;
    for (int32_t r0 = 0; r0 < rows;) {
      int32_t r1 = r0;
      while (r1 < rows && indptr[r1 + 1] - indptr[r0] <= part_limit) ++r1;
      if (r1 == r0) throw std::invalid_argument("a single row exceeds the per-block nonzero limit of CSRMatrix");
      top->part_row0.push_back(r0);
      top->parts.push_back(build(r0, r1));
      top->max_row = std::max(top->max_row, top->parts.back()->max_row);
      r0 = r1;
    }
    *out = top.release();
  });
}

int imp_csr_shape(const imp_csr *m, int32_t *rows, int32_t *cols, int64_t *nonzeros) {
  return guarded([&] {
    if (rows) *rows = m->rows;
    if (cols) *cols = m->cols;
    if (nonzeros) *nonzeros = m->nnz;
  });
}

int imp_csr_destroy(imp_csr *m) {
  return guarded([&] { delete m; });
}

int imp_coo_create(int32_t rows, int32_t cols, int64_t nnz, const int32_t *row, const int32_t *col,
                   const float *data, imp_coo **out) {
  return guarded([&] {
    if (rows < 0 || cols < 0 || nnz < 0) throw std::invalid_argument("negative dimension for COOMatrix");
    auto m = std::make_unique<imp_coo>();
    m->rows = rows;
    m->cols = cols;
    m->nnz = nnz;
    m->row.upload(row, (size_t)nnz);
    m->col.upload(col, (size_t)nnz);
    if (data) m->data.upload(data, (size_t)nnz);
    sync();
    *out = m.release();
  });
}

// (row, col) pattern of a host CSR matrix -> device COO.  One wavefront per row writes the row ids, every thread copies its
// share of the column ids; both inputs are read from page-locked host memory in place (coalesced, once)
__global__ void coo_from_csr_kernel(const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices, int rows, int64_t nnz,
                                    int32_t *__restrict__ row_out, int32_t *__restrict__ col_out) {
  const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x, n = (size_t)gridDim.x * blockDim.x;
  for (size_t i = t; i < (size_t)nnz; i += n) col_out[i] = indices[i];
  const int lane = threadIdx.x & 63;
  for (size_t r = t >> 6; r < (size_t)rows; r += n >> 6) {
    const int b = indptr[r], e = indptr[r + 1];
    for (int j = b + lane; j < e; j += 64) row_out[j] = (int32_t)r;
  }
}

int imp_coo_create_from_csr_pattern(int32_t rows, int32_t cols, const void *indptr, int indptr_is_64, const int32_t *indices,
                                    imp_coo **out) {
  return guarded([&] {
    if (rows < 0 || cols < 0) throw std::invalid_argument("negative dimension for COOMatrix");
    if (!indptr) throw std::invalid_argument("COOMatrix.from_csr_pattern: indptr is required");
    const int64_t *p64 = indptr_is_64 ? static_cast<const int64_t *>(indptr) : nullptr;
    const int32_t *p32 = indptr_is_64 ? nullptr : static_cast<const int32_t *>(indptr);
    auto at = [&](int64_t i) -> int64_t { return p64 ? p64[i] : (int64_t)p32[i]; };
    const int64_t base = at(0), nnz = at(rows) - base;
    if (nnz < 0 || nnz > INT32_MAX) throw std::invalid_argument("COOMatrix.from_csr_pattern: nonzero count out of range");
    if (nnz && !indices) throw std::invalid_argument("COOMatrix.from_csr_pattern: indices are required");
    auto m = std::make_unique<imp_coo>();
    m->rows = rows, m->cols = cols, m->nnz = nnz;
    m->row.alloc((size_t)nnz);
    m->col.alloc((size_t)nnz);
    if (nnz) {
      Context &c = ctx();
      const size_t words = (size_t)rows + 1 + (size_t)nnz;
      if (!c.pin_stage_ev) IMP_CHECK_HIP(hipEventCreateWithFlags(&c.pin_stage_ev, hipEventDisableTiming));
      else IMP_CHECK_HIP(hipEventSynchronize(c.pin_stage_ev));  // the previous expand kernel is done with the buffer
      if (c.pin_stage_bytes < words * 4) {
        if (c.pin_stage) (void)hipHostFree(c.pin_stage);
        c.pin_stage = nullptr, c.pin_stage_bytes = 0;
        const size_t want = std::max(words * 4 * 2, (size_t)1 << 20);
        IMP_CHECK_HIP(hipHostMalloc(&c.pin_stage, want, hipHostMallocDefault));
        c.pin_stage_bytes = want;
      }
      int32_t *sp = static_cast<int32_t *>(c.pin_stage), *si = sp + rows + 1;
      int64_t prev = 0;
      for (int64_t r = 0; r <= rows; ++r) {
        const int64_t v = at(r) - base;
        if (v < prev || v > nnz) throw std::invalid_argument("COOMatrix.from_csr_pattern: indptr must be non-decreasing");
        sp[r] = (int32_t)v, prev = v;
      }
      std::memcpy(si, indices + base, (size_t)nnz * 4);
      const int grid = (int)std::min<size_t>(((size_t)std::max<int64_t>(nnz, (int64_t)rows * 64) + 255) / 256, (size_t)c.num_cus * 8);
      coo_from_csr_kernel<<<grid, 256, 0, stream()>>>(sp, si, rows, nnz, m->row.data(), m->col.data());
      IMP_CHECK_HIP(hipGetLastError());
      IMP_CHECK_HIP(hipEventRecord(c.pin_stage_ev, stream()));
      // no host wait: the caller's arrays have been copied, and whatever reads the matrix is queued behind the kernel
    }
    *out = m.release();
  });
}

int imp_coo_destroy(imp_coo *m) {
  return guarded([&] { delete m; });
}

// ---- profiler -------------------------------------------------------------------------------------
int imp_prof_enable(int on) {
  return guarded([&] {
    std::lock_guard<std::recursive_mutex> plock(g_prof_mutex);
    prof_flush();
    g_prof_on = on != 0;
  });
}
int imp_prof_filter(const char *substr) {
  return guarded([&] {
    std::lock_guard<std::recursive_mutex> plock(g_prof_mutex);
    prof_flush();
    g_prof_filter = substr ? substr : "";
  });
}
int imp_prof_reset(void) {
  return guarded([&] {
    std::lock_guard<std::recursive_mutex> plock(g_prof_mutex);
    prof_flush();
    g_prof.clear();
  });
}
int imp_prof_get(const char *kernel, double *total_ms, int64_t *launches) {
  return guarded([&] {
    std::lock_guard<std::recursive_mutex> plock(g_prof_mutex);
    prof_flush();
    auto it = g_prof.find(kernel);
    if (total_ms) *total_ms = it == g_prof.end() ? 0.0 : it->second.total_ms;
    if (launches) *launches = it == g_prof.end() ? 0 : it->second.launches;
  });
}
int imp_prof_names(char *buf, size_t buflen) {
  return guarded([&] {
    std::lock_guard<std::recursive_mutex> plock(g_prof_mutex);
    prof_flush();
    std::string s;
    for (auto &kv : g_prof) {
      if (!s.empty()) s += "\n";
      s += kv.first;
    }
    if (buflen) {
      size_t n = std::min(buflen - 1, s.size());
      memcpy(buf, s.data(), n);
      buf[n] = 0;
    }
  });
}

}  // extern "C"
