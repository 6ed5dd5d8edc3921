// Internal shared declarations for libimplicit_hip.so (not part of the C-ABI).
#ifndef IMPLICIT_AMD_CSRC_COMMON_H_
#define IMPLICIT_AMD_CSRC_COMMON_H_

#include <hip/hip_runtime.h>

#include <climits>
#include <cstdint>
#include <cstdio>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/implicit_hip.h"

namespace imp {

// ---- error plumbing: C++ exceptions inside, status codes at the extern "C" edge -------------
struct out_of_range_error : std::runtime_error {
  using std::runtime_error::runtime_error;
};

void set_last_error(const std::string &msg);

#define IMP_CHECK_HIP(expr)                                                                    \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(_e) + " (" +    \
                               __FILE__ + ":" + std::to_string(__LINE__) + ")");             \
    }                                                                                          \
  } while (0)

// ---- per-device context: one stream, lazily created; owns every scratch buffer the kernels share ------------------
struct Context;
Context &ctx();  // context of the current device
// Serialises the C-ABI calls of one device: the library stream, the profiler's pending events and the workspaces below
// are shared by everything that runs on a device, and ctypes drops the GIL around every call -- two Python threads
// calling into the same device would otherwise interleave kernels on the same scratch buffers (the reference allocates
// its temporaries per call).  Calls on different devices do not block each other.
std::unique_lock<std::recursive_mutex> lock_device();

// wraps the body of every extern "C" function
template <typename F> int guarded(F &&body) {
  try {
    auto lock = lock_device();
    body();
    return IMP_OK;
  } catch (const std::invalid_argument &e) {
    set_last_error(e.what());
    return IMP_INVALID_ARGUMENT;
  } catch (const out_of_range_error &e) {
    set_last_error(e.what());
    return IMP_OUT_OF_RANGE;
  } catch (const std::exception &e) {
    set_last_error(e.what());
    return IMP_RUNTIME_ERROR;
  } catch (...) {
    set_last_error("unknown error");
    return IMP_RUNTIME_ERROR;
  }
}

// host-only entry points (no stream, no workspace): same error mapping without the device's call lock, so that they can run
// beside device calls of another thread (fit() transposes the matrix while the first CSR uploads)
template <typename F> int guarded_host(F &&body) {
  try {
    body();
    return IMP_OK;
  } catch (const std::invalid_argument &e) {
    set_last_error(e.what());
    return IMP_INVALID_ARGUMENT;
  } catch (const out_of_range_error &e) {
    set_last_error(e.what());
    return IMP_OUT_OF_RANGE;
  } catch (const std::exception &e) {
    set_last_error(e.what());
    return IMP_RUNTIME_ERROR;
  } catch (...) {
    set_last_error("unknown error");
    return IMP_RUNTIME_ERROR;
  }
}

inline hipStream_t stream();
void sync();  // hipStreamSynchronize on the library stream
// end of a C-ABI call that only queues device work (solver sweeps, gramian, all-reduce): host wait unless the device is in
// deferred mode (imp_set_deferred_sync), where the caller orders a whole iteration with ONE imp_device_synchronize
void sync_call();
unsigned long long *fixup_total();  // als_cg_fixup.hip: host-mapped count of the rows the fix-up kernel re-solved on this device
bool w256_enabled();         // als_cg_w256.hip: resident lock-step kernels for the rows of <= 256 nonzeros at f = 256 (IMP_F256_OLD=1: round-2 kernel)
bool nm_enabled();           // als_cg_nm.hip: long rows of the f = 64 / 128 path through their explicit normal matrix (IMP_NM=0: streamed)

// ---- launch-time profiler (HIP events on the library stream) ------------------------------------
struct ProfScope {
  explicit ProfScope(const char *name);
  ~ProfScope();
  const char *name;
  hipEvent_t start = nullptr, stop = nullptr;
};
bool prof_enabled();
#define IMP_PROF(name) ::imp::ProfScope _prof_scope_(name)

// ---- device storage ---------------------------------------------------------------------------
struct Context;
struct Storage {
  void *ptr = nullptr;
  size_t bytes = 0;
  bool owned = true;
  // small blocks (<= kSmallMax) come from, and return to, their device's free lists instead of hipMalloc / hipFree: a
  // recommend() batch creates and destroys an IntVector and a COO filter (20-50 us of allocator time per object, hipFree
  // synchronises).  Safe without a synchronisation: every side stream (row-class streams, exchange stream) is joined to the
  // library stream of its device before the entry point that used it returns, so a recycled block's next use -- queued on that
  // stream -- is ordered behind its last.  The lists are a cache: an allocation that fails empties them and retries.
  Context *home = nullptr;
  int size_class = -1;
  bool exposed = false;  // the address left the library (imp_matrix_device_ptr): writes to it can no longer be tracked
  static constexpr size_t kSmallMax = (size_t)4 << 20;
  Storage(size_t bytes_, bool zero);
  Storage(void *foreign) : ptr(foreign), owned(false) {}
  ~Storage();
  Storage(const Storage &) = delete;
  Storage &operator=(const Storage &) = delete;
};

template <typename T> struct DeviceArray {
  std::shared_ptr<Storage> storage;
  size_t size = 0;
  T *data() const { return storage ? reinterpret_cast<T *>(storage->ptr) : nullptr; }
  void alloc(size_t n, bool zero = false) {
    storage = std::make_shared<Storage>(n * sizeof(T), zero);
    size = n;
  }
  void upload(const T *host, size_t n) {
    alloc(n);
    if (n) IMP_CHECK_HIP(hipMemcpyAsync(data(), host, n * sizeof(T), hipMemcpyHostToDevice, stream()));
  }
};

// Scratch buffers are per DEVICE (a process may drive several devices through imp_set_device) and are only touched under
// that device's call lock.
struct Context {
  int device = 0;
  hipStream_t stream = nullptr;
  int num_cus = 256;
  // Persistent row kernels launch `slots()` workgroups, each with a fixed share of the rows.  oversub > 1 (multi-GPU
  // driver, imp_set_oversubscribe) launches that many times more workgroups with proportionally smaller shares: when part
  // of the device is held by another stream's kernels (RCCL send / recv) the workgroups that have to wait for a slot no
  // longer carry a full share, and the hardware dispatcher balances the rest.
  int oversub = 1;
  // imp_set_deferred_sync: the queue-only entry points return without a host wait (the multi-GPU driver queues a whole
  // iteration -- K solve chunks, their exchanges, two all-reduces -- and waits once)
  bool deferred = false;
  hipStream_t occupy_stream = nullptr;  // imp_debug_occupy
  std::recursive_mutex mutex;
  std::mutex small_mutex;                  // the free lists below (a Storage may die on a thread that holds another device's lock)
  std::vector<void *> small_free[24];      // [log2 size]: recycled device blocks of 256 B .. 4 MB (Storage)
  // page-locked, device-addressable staging of imp_coo_create_from_csr_pattern: the expand kernel reads it in place; the event
  // marks that kernel's end (the next call waits for it before overwriting the buffer)
  void *pin_stage = nullptr;
  size_t pin_stage_bytes = 0;
  hipEvent_t pin_stage_ev = nullptr;
  DeviceArray<float> gram_ws;     // split-K partial gramians (gramian.hip)
  DeviceArray<float> long_ws;     // partial vectors / CG state of the long rows (als_cg.hip)
  DeviceArray<float> pad_x, pad_y, pad_gram;  // zero-padded copies for factor counts that ride the f = 64 / 128 / 256 kernels (als_cg.hip)
  // which matrix pad_y currently holds a padded copy of (als_cg.hip least_squares_cg_padded): the chunks of a sharded half sweep
  // solve against the same Y, one padded copy serves them all
  const void *pad_y_src = nullptr;
  size_t pad_y_rows = 0;
  int pad_y_f = 0, pad_y_F = 0;
  DeviceArray<int> pad_same;  // device flag: the gramian of this call equals the one pad_y was made under
  DeviceArray<double> loss_buf;   // 4 accumulators of the loss kernel (solver.hip)
  DeviceArray<unsigned long long> chol_failed;  // smallest failing row of a Cholesky sweep (als_cholesky.hip)
  DeviceArray<float> barrier_word;              // operand of the RCCL barrier (comm.hip)
  unsigned long long *fixup_total = nullptr;     // host-mapped: rows re-solved by the fp32 fix-up kernel since the last imp_solver_fixup_rows(reset)
  DeviceArray<float> w256_ws;                    // fp16-split gramian in fragment order + header (als_cg_w256.hip)
  DeviceArray<unsigned> nm_fix_rows;             // rows the normal-matrix kernels left to the fix-up kernel (operands beyond the fp16 range)
  DeviceArray<int> nm_ticket;                    // work counter of the normal-matrix kernel (als_cg_nm.hip), reset by every launch
};
inline hipStream_t stream() { return ctx().stream; }
// a C-ABI entry point is about to write `bytes` at `dst` through the library (or the memory is being freed): a padded copy of Y
// made from memory it overlaps (least_squares_cg_padded) is no longer to be trusted -- on WHICHEVER device's context the copy
// lives (device addresses are unique across the devices of a process; a Storage may die on a thread whose current device is
// not the one that made the copy).  containers.hip.
void note_device_write(const void *dst, size_t bytes);
// Something derived from device memory the library owns and kept across calls (the fragment-ordered fp16 planes of an item
// matrix in a KnnQuery handle, topk.hip): `src` / `bytes` name what it was made from; note_device_write clears `src` of every
// registered cache the written range overlaps.  Holds for memory only the library writes: a Storage whose address was handed
// out (imp_matrix_device_ptr) or that wraps foreign memory (imp_matrix_wrap_device) is `exposed` and never cached from.
struct DerivedCache {
  const void *src = nullptr;
  size_t bytes = 0;
};
void register_derived_cache(DerivedCache *c);
void unregister_derived_cache(DerivedCache *c);

}  // namespace imp

// ---- handle definitions (opaque in the public header) -------------------------------------------
struct imp_matrix {
  size_t rows = 0, cols = 0, itemsize = 4;
  void *data = nullptr;  // may point inside storage (views)
  std::shared_ptr<imp::Storage> storage;
  size_t bytes() const { return rows * cols * itemsize; }
  float *f32() const {
    if (itemsize != 4) throw std::runtime_error("can't cast Matrix to float*");
    return reinterpret_cast<float *>(data);
  }
};

struct imp_intvector {
  imp::DeviceArray<int32_t> v;
  size_t size = 0;
};

// Device view of the long-row plan (passed to kernels by value).
struct LongPlanDev {
  int n_long, n_seg;
  const int32_t *rows;       // [n_long]   row id of each long row
  const int32_t *row_seg;    // [n_long+1] first segment of each long row
  const int32_t *seg_row;    // [n_seg]    long-row index of each segment
  const int32_t *seg_begin;  // [n_seg]    nnz range of each segment
  const int32_t *seg_end;    // [n_seg]
  const int32_t *seg_exec;   // [n_seg]    execution order of the partial kernel: segment ids grouped by XCD
  int xcd_start[9];          // seg_exec[xcd_start[x] .. xcd_start[x+1]) is swept by the workgroups with blockIdx % 8 == x
};

// Host-side owner of one long-row plan (device arrays + the per-XCD cut of the execution order).
struct LongPlan {
  int32_t n_long = 0, n_seg = 0;
  imp::DeviceArray<int32_t> row_seg, seg_row, seg_begin, seg_end, seg_exec;
  int32_t xcd_start[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  int32_t stripe = 0;  // column-stripe width (0: rows cut into plain kSegment runs)
  LongPlanDev dev(const int32_t *rows) const {
    LongPlanDev d{n_long, n_seg, rows, row_seg.data(), seg_row.data(), seg_begin.data(), seg_end.data(), seg_exec.data(), {0}};
    for (int x = 0; x < 9; ++x) d.xcd_start[x] = xcd_start[x];
    return d;
  }
};

// Rows are scheduled in length classes so that the work per wavefront is even and the longest rows
// start first (SURVEY section 7 "load imbalance"): `order` = row ids sorted by descending nnz;
// class b covers order[bin_start[b] .. bin_start[b+1]) and holds the rows with
// kClassMax[b+1] < nnz <= kClassMax[b]:
//   0 long  (> 512 nnz): cut into segments of <= kSegment nnz (column-striped when the rows re-use the gathered
//        matrix enough, see imp_csr_create), segment-parallel CG passes
//   1..4 mid (256,512], (128,256], (64,128], (32,64]: a TEAM of 16/8/4/2 wavefronts per row, every
//        wavefront keeps one 32-row gathered tile in registers for all passes
//   5 short (16,32] and 6 short (0,16]: one wavefront per row, resident tile of 32 / 16 entries, 16 rows per
//        workgroup in lock step (MFMA gramian product)
//   7 empty
struct imp_csr {
  static constexpr int kBins = 8;
  static constexpr int kShortRow = 32;
  static constexpr int kLongRow = 512;
  static constexpr int kSegment = 512;
  static constexpr int32_t kClassMax[kBins + 1] = {INT32_MAX, 512, 256, 128, 64, 32, 16, 0, -1};
  int32_t rows = 0, cols = 0;
  int64_t nnz = 0;
  imp::DeviceArray<int32_t> indptr, indices;
  imp::DeviceArray<float> data;
  imp::DeviceArray<int32_t> order;
  int32_t bin_start[kBins + 1] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  int32_t max_row = 0;
  // long-row plans: `plan_all` covers every row of class 0 (> kLongRow nonzeros) and is what the generic kernels (and the f = 64 /
  // 128 path under IMP_NM=0) stream.  Rows of more than kCholLongRow nonzeros (the first n_chol_long entries of `order`) cut into
  // plain runs of kCholSegment: the A-build of the f = 64 Cholesky kernel is segment-parallel for them (als_cholesky.hip)
  LongPlan plan_all;
  static constexpr int kCholLongRow = 1024, kCholSegment = 1024;
  LongPlan plan_chol;
  int32_t n_chol_long = 0;
  // every row of class 0 cut into plain runs of `nm_segment` nonzeros (2048 .. 16384: about eight segments per CU and launch, so
  // that neither the tail of the launch nor the partial matrices of the cut rows weigh): the work list of the normal-matrix
  // kernels (als_cg_nm.hip).  The first nm_multi_rows rows (those longer than one segment) own the first nm_multi_segs segments.
  LongPlan plan_nm;
  int32_t nm_segment = 2048, nm_multi_rows = 0, nm_multi_segs = 0;
  // A matrix with more than 2^31 - 1 nonzeros (imp_csr_create64) is held as consecutive row blocks, each a complete
  // imp_csr of its own with int32 offsets; the top-level object then only carries rows / cols / nnz and the solver
  // entry points walk the blocks (every row solve is independent of the others).
  std::vector<std::unique_ptr<imp_csr>> parts;
  std::vector<int32_t> part_row0;  // first row of each block
  int32_t nonempty() const { return bin_start[kBins - 1]; }
  int32_t first_empty() const { return bin_start[kBins - 1]; }
  int32_t n_empty() const { return bin_start[kBins] - bin_start[kBins - 1]; }
};

struct imp_coo {
  int32_t rows = 0, cols = 0;
  int64_t nnz = 0;
  imp::DeviceArray<int32_t> row, col;
  imp::DeviceArray<float> data;
};

namespace imp {
// als_cg_nm.hip: Cholesky half sweep at f = 128 through the rows' normal matrices on the matrix cores; what it could not factorise
// comes back as a device-side list for the workgroup-per-row fp32 kernel (als_cholesky.hip)
// 64 < f < 128 Cholesky on the f = 128 path (als_cg.hip owns the pad kernels and workspaces)
void cholesky_pad_in(const imp_matrix *X, const imp_matrix *Y, const imp_matrix *YtY, size_t rx, int F);
void cholesky_pad_out(imp_matrix *X, size_t rx, int F);
struct CholNmList {
  const unsigned *count, *rows;
  int capacity;
};
CholNmList least_squares_cholesky_nm(const imp_csr *C, float *X, const float *Y, size_t y_rows, const float *YtY, float reg);
// als_cg_fixup.hip: the fp32 one-wavefront-per-row solver for the rows listed in rows[0 .. *count) (F = 64 / 128, float / __half)
template <int F, typename T>
void launch_cg_fixup(const unsigned *count, const unsigned *rows, int capacity, const imp_csr *C, T *X, const T *Y, const float *A0,
                     int cg_steps);
}  // namespace imp

#endif  // IMPLICIT_AMD_CSRC_COMMON_H_
