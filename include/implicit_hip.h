/*
 * implicit_hip.h -- C-ABI of libimplicit_hip.so, the MI355X (gfx950) replacement for the
 * native half of benfred/implicit's `implicit.gpu` plug-in.
 *
 * Every entry point below is what a binding for this path binds instead of the reference's C++
 * classes (which Cython reaches through implicit/gpu/{matrix,als,knn,random,utils}.pxd).  The
 * reference interface each one replaces is cited as file:line relative to the reference tree.
 * Plain pointers, sizes and opaque handles only: no C++ types, no torch types.
 *
 * Conventions
 *   - every function returns an imp_status; on failure imp_last_error() (thread-local) holds the
 *     message.  The mapping to the reference's exceptions (implicit/gpu/utils.h:15-73 and the
 *     Cython `except +` translation) is: IMP_INVALID_ARGUMENT -> std::invalid_argument ->
 *     ValueError; IMP_OUT_OF_RANGE -> IndexError; IMP_RUNTIME_ERROR -> std::runtime_error /
 *     std::logic_error -> RuntimeError.  No C++ exception ever crosses this boundary.
 *   - all calls are synchronous on return, like the reference's (cudaDeviceSynchronize after each
 *     kernel, implicit/gpu/als.cu:147,151,196,276; sync_stream, knn.cu:254).  Work is issued on
 *     one library-owned HIP stream per device.
 *   - matrices are row-major; itemsize 4 = fp32, 2 = fp16 storage (implicit/gpu/matrix.h:23-90).
 *   - handles own device memory; row/slice views share their parent's storage by reference count
 *     (reference: shared_ptr<rmm::device_buffer>, matrix.h:90).
 */
#ifndef IMPLICIT_HIP_H_
#define IMPLICIT_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  IMP_OK = 0,
  IMP_INVALID_ARGUMENT = 1, /* ValueError   */
  IMP_OUT_OF_RANGE = 2,     /* IndexError   */
  IMP_RUNTIME_ERROR = 3     /* RuntimeError */
} imp_status;

typedef struct imp_matrix imp_matrix;       /* implicit::gpu::Matrix       matrix.h:23-90  */
typedef struct imp_intvector imp_intvector; /* implicit::gpu::Vector<int>  matrix.h:12-21  */
typedef struct imp_csr imp_csr;             /* implicit::gpu::CSRMatrix    matrix.h:92-99  */
typedef struct imp_coo imp_coo;             /* implicit::gpu::COOMatrix    matrix.h:101-108 */
typedef struct imp_solver imp_solver;       /* implicit::gpu::LeastSquaresSolver  als.h:11-24 */
typedef struct imp_knn imp_knn;             /* implicit::gpu::KnnQuery     knn.h:15-35     */
typedef struct imp_random imp_random;       /* implicit::gpu::RandomState  random.h:10-21  */
typedef struct imp_comm imp_comm;           /* NEW: RCCL communicator (reference: "TODO: multi-gpu", als.cu:169) */

/* ---- library / device ------------------------------------------------------------------- */
const char *imp_last_error(void);
/* get_device_count(), utils.h:75-79: fails with IMP_RUNTIME_ERROR when no device is usable. */
int imp_get_device_count(int *count);
int imp_set_device(int device);
int imp_get_device(int *device);
/* NEW (multi-GPU robustness).  The row kernels are persistent: by default exactly as many workgroups as the device holds at
 * once, each with a fixed share of the rows.  While another stream's kernels (RCCL send / recv) hold part of the device, the
 * workgroups that must wait for a slot would do their whole share late.  factor > 1 launches factor x as many workgroups
 * with proportionally smaller shares; the hardware dispatcher then balances them over the slots that are free. */
int imp_set_oversubscribe(int factor);
int imp_get_oversubscribe(int *factor);
/* NEW.  Every call is synchronous on return by default (convention above).  on = 1 puts the CURRENT device into deferred
 * mode: imp_solver_calculate_yty, imp_solver_least_squares and imp_comm_allreduce_sum only queue their work on the library
 * stream and return; the caller orders a whole iteration with one imp_device_synchronize (which also reports a timed-out
 * cluster exchange).  The multi-GPU driver uses it so that a chunk's exchange is queued without a host round trip per chunk.
 * on = 0 waits for everything queued and restores the default. */
int imp_set_deferred_sync(int on);
/* Measurement aid: occupies `workgroups` x (256 threads, 32 KB LDS) for about `microseconds` on a stream of its own. */
int imp_debug_occupy(int workgroups, int microseconds);
/* Measurement aid: the shader core clock right now, in MHz -- a one-wavefront kernel queued on the library stream counts core
 * cycles over `microseconds` of the constant-rate wall clock (behind whatever is queued: the clock the chip has recovered to).
 * microseconds < 0: the probe runs on a side stream BESIDE the work queued on the library stream (deferred mode) -- the clock
 * the queued kernels themselves run at (a power-limited chip clocks an MFMA-heavy kernel well below its idle clock). */
int imp_debug_core_clock(int microseconds, double *mhz);
int imp_device_synchronize(void);
/* NEW.  Rows of CG sweeps on the CURRENT device that a fast kernel declined to store and the fp32 one-wavefront-per-row fix-up
 * kernel re-solved instead, since the last reset: rows of a cluster whose exchange through memory was lost, and long rows whose
 * fp16-split matrix-core operands left the fp16 range (factors or confidences far outside anything ALS produces).  Results are
 * correct either way; a non-zero count says the sweep ran slower than it should.  Waits for the library stream. */
int imp_solver_fixup_rows(unsigned long long *count, int reset);
int imp_mem_get_info(size_t *free_bytes, size_t *total_bytes);
/* NEW: frees the per-device scratch buffers the solver paths grow on demand (split-K gramian partials, long-row CG state, the
 * zero-padded copies other factor counts ride the f = 64 / 128 / 256 kernels on, cluster exchange slots); the next call that
 * needs one allocates it again.  fit() calls it when it returns. */
int imp_release_workspaces(void);
const char *imp_version(void);

/* NEW, host-side helper (no device work, usable without a GPU): stable parallel transpose of a CSR matrix with int32 offsets
 * (the reference's fit transposes on the host with scipy, implicit/gpu/als.py:121).  Outputs are caller-allocated:
 * t_indptr[cols + 1], t_indices[nonzeros], t_data[nonzeros]; row ids inside every output row come out ascending.
 * threads <= 0: half the hardware threads, at most 32. */
int imp_host_csr_transpose(int32_t rows, int32_t cols, int64_t nonzeros, const int32_t *indptr, const int32_t *indices,
                           const float *data, int32_t *t_indptr, int32_t *t_indices, float *t_data, int threads);

/* ---- Matrix (matrix.h:23-90, matrix.cu:34-220) -------------------------------------------- */
/* Matrix(rows, cols, data, allocate=true, itemsize): allocates; copies rows*cols*itemsize bytes
 * from host_data when non-NULL, zero-fills otherwise (matrix.cu:80-96). */
int imp_matrix_create(size_t rows, size_t cols, const void *host_data, size_t itemsize, imp_matrix **out);
/* Matrix(rows, cols, device_ptr, allocate=false, itemsize): wraps foreign device memory, never
 * frees it (matrix.cu:93-95; __cuda_array_interface__ path of _cuda.pyx:99-104). */
int imp_matrix_wrap_device(size_t rows, size_t cols, void *device_ptr, size_t itemsize, imp_matrix **out);
/* Matrix(other, rowid): one-row view sharing storage (matrix.cu:34-40). */
int imp_matrix_row(const imp_matrix *m, size_t rowid, imp_matrix **out);
/* Matrix(other, start, end): row-range view sharing storage (matrix.cu:42-53). */
int imp_matrix_slice(const imp_matrix *m, size_t start, size_t end, imp_matrix **out);
/* Matrix(other, rowids): gather-copy of the selected rows (matrix.cu:55-78). */
int imp_matrix_gather(const imp_matrix *m, const imp_intvector *rowids, imp_matrix **out);
/* Matrix::resize: grows rows only, zero-fills the new rows (matrix.cu:98-120). */
int imp_matrix_resize(imp_matrix *m, size_t rows, size_t cols);
/* Matrix::assign_rows(rowids, other): scatter rows of `other` (fp32 only, matrix.cu:122-143). */
int imp_matrix_assign_rows(imp_matrix *m, const imp_intvector *rowids, const imp_matrix *other);
/* Matrix::astype(itemsize): fp32 <-> fp16 copy (matrix.cu:153-171). */
int imp_matrix_astype(const imp_matrix *m, size_t itemsize, imp_matrix **out);
/* Matrix::calculate_norms(): 1 x rows fp32 L2 norms, zeros replaced by 1e-10 (matrix.cu:173-215). */
int imp_matrix_calculate_norms(const imp_matrix *m, imp_matrix **out);
/* Matrix::to_host (matrix.cu:217-220). */
int imp_matrix_to_host(const imp_matrix *m, void *host_out);
/* NEW: overwrite the matrix from host memory (same shape/itemsize); used to re-seed parity runs. */
int imp_matrix_from_host(imp_matrix *m, const void *host_in);
/* NEW: device-to-device copy of `rows` whole rows, src[src_row ..] -> dst[dst_row ..] (same width and itemsize), queued on
 * the library stream (queue-only in deferred mode).  What an exchange between LOGICAL ranks that share one device is made
 * of (implicit_amd/gpu/local_comm.py: the N-rank driver exercised on one GPU). */
int imp_matrix_copy_rows(imp_matrix *dst, size_t dst_row, const imp_matrix *src, size_t src_row, size_t rows);
int imp_matrix_shape(const imp_matrix *m, size_t *rows, size_t *cols, size_t *itemsize);
int imp_matrix_device_ptr(const imp_matrix *m, void **ptr);
int imp_matrix_destroy(imp_matrix *m);

/* ---- Vector<int> / CSRMatrix / COOMatrix (matrix.cu:13-32, 222-280) ------------------------- */
int imp_intvector_create(const int32_t *host_data, size_t size, imp_intvector **out);
int imp_intvector_destroy(imp_intvector *v);
/* CSRMatrix(rows, cols, nonzeros, indptr, indices, data): H2D copy into plain device memory (the
 * reference uses managed memory + ReadMostly advise, matrix.cu:222-251).  Also builds the
 * row-length bins the solver kernels schedule from (device-side metadata, not visible here). */
int imp_csr_create(int32_t rows, int32_t cols, int64_t nonzeros, const int32_t *indptr,
                   const int32_t *indices, const float *data, imp_csr **out);
/* NEW: the same with 64-bit row offsets (the CPU reference accepts both widths, _als.pyx:76; the CUDA class is int32
 * only, matrix.h:92-99).  More than 2^31 - 1 nonzeros are held as consecutive row blocks internally; indices stay
 * int32 (cols < 2^31). */
int imp_csr_create64(int32_t rows, int32_t cols, int64_t nonzeros, const int64_t *indptr,
                     const int32_t *indices, const float *data, imp_csr **out);
int imp_csr_shape(const imp_csr *m, int32_t *rows, int32_t *cols, int64_t *nonzeros);
int imp_csr_destroy(imp_csr *m);
int imp_coo_create(int32_t rows, int32_t cols, int64_t nonzeros, const int32_t *row,
                   const int32_t *col, const float *data, imp_coo **out);
/* NEW: the (row, col) pattern of a HOST CSR matrix as a device COO without values -- all the top-k filters read
 * (knn.cu:197-214 reads row / col only).  `indptr`: rows + 1 offsets, int32 or int64 (`indptr_is_64`); `indices`: int32.
 * Both are copied to page-locked staging and expanded by a kernel: no blocking upload, no host-side row expansion.
 * recommend() builds one per batch (the reference: scipy tocoo() + three Vector uploads, matrix_factorization_base.py:109-112,
 * matrix.cu:253-262). */
int imp_coo_create_from_csr_pattern(int32_t rows, int32_t cols, const void *indptr, int indptr_is_64,
                                    const int32_t *indices, imp_coo **out);
int imp_coo_destroy(imp_coo *m);

/* ---- LeastSquaresSolver (als.h:11-24, als.cu:118-281) ---------------------------------------- */
int imp_solver_create(imp_solver **out);
int imp_solver_destroy(imp_solver *s);
/* calculate_yty(Y, &YtY, regularization): YtY(f x f fp32) = Y^T Y + reg*I  (als.cu:122-152).
 * MFMA (v_mfma_f32_32x32x2_f32) split over row blocks with a fixed-order second stage. */
int imp_solver_calculate_yty(imp_solver *s, const imp_matrix *Y, imp_matrix *YtY, float regularization);
/* least_squares(Cui, &X, YtY, Y, cg_steps): warm-started CG half sweep, X in place; YtY is the
 * regularised gramian (als.cu:154-197; numerics follow the CPU oracle _als.pyx:152-248). */
int imp_solver_least_squares(imp_solver *s, const imp_csr *cui, imp_matrix *X, const imp_matrix *YtY,
                             const imp_matrix *Y, int cg_steps);
/* NEW (the reference GPU path has no Cholesky solver; mirrors the CPU _als._least_squares,
 * _als.pyx:75-142): YtY is UNregularised, `regularization` (double) is added to the diagonal,
 * the previous X is ignored.  On a non-positive-definite row returns IMP_INVALID_ARGUMENT
 * (ValueError, as _als.pyx:136-138) and *failed_row = that row (else -1). */
int imp_solver_least_squares_cholesky(imp_solver *s, const imp_csr *cui, imp_matrix *X,
                                      const imp_matrix *YtY, const imp_matrix *Y,
                                      double regularization, int64_t *failed_row);
/* calculate_loss(Cui, X, Y, regularization) (als.cu:253-281; oracle _als.pyx:257-308). */
int imp_solver_calculate_loss(imp_solver *s, const imp_csr *cui, const imp_matrix *X,
                              const imp_matrix *Y, float regularization, float *loss_out);

/* ---- KnnQuery (knn.h:15-35, knn.cu:56-265) ---------------------------------------------------- */
/* KnnQuery(max_temp_memory): 0 = min(free/2, 4 GiB) (knn.cu:56-75). */
int imp_knn_create(size_t max_temp_memory, imp_knn **out);
int imp_knn_destroy(imp_knn *k);
/* topk(items, query, k, indices, distances, item_norms, query_filter, item_filter) (knn.cu:77-265).
 * indices/distances are caller-allocated [query.rows x k] HOST or DEVICE buffers (auto-detected,
 * knn.cu:40-54,147-164).  Rows are written best-first in the total order (score desc, column
 * desc); when k > items.rows only the first items.rows entries of each row are written
 * (knn.cu:239).  Filtered entries score -FLT_MAX (topk.pyx:51). */
int imp_knn_topk(imp_knn *k, const imp_matrix *items, const imp_matrix *query, int topk,
                 int32_t *indices, float *distances, const imp_matrix *item_norms,
                 const imp_coo *query_filter, const imp_intvector *item_filter);

/* ---- RandomState (random.h:10-21, random.cu:14-40) ------------------------------------------- */
int imp_random_create(int64_t seed, imp_random **out);
int imp_random_destroy(imp_random *r);
int imp_random_uniform(imp_random *r, size_t rows, size_t cols, float low, float high, imp_matrix **out);
int imp_random_randn(imp_random *r, size_t rows, size_t cols, float mean, float stddev, imp_matrix **out);

/* ---- NEW: multi-GPU exchange over RCCL / xGMI (no reference counterpart) ------------------------ */
/* One process per GPU.  Rank 0 calls imp_comm_unique_id, the host side broadcasts the 128 bytes by
 * any means (torch.distributed store, MPI, a file) and every rank calls imp_comm_init_rank. */
#define IMP_COMM_UNIQUE_ID_BYTES 128
int imp_comm_unique_id(void *id_out);
int imp_comm_init_rank(const void *id, int nranks, int rank, imp_comm **out);
int imp_comm_destroy(imp_comm *c);
/* in-place sum all-reduce of an fp32 matrix (the f x f gramian). */
int imp_comm_allreduce_sum(imp_comm *c, imp_matrix *m);
/* all-gather of row shards: rank r contributes rows [row_offsets[r], row_offsets[r+1]) of `full`
 * (already in place in its own copy); afterwards every rank holds all rows. */
int imp_comm_allgather_rows(imp_comm *c, imp_matrix *full, const int64_t *row_offsets);
/* Pipelined form of the all-gather: rows [row_lo[r], row_hi[r]) are owned by rank r (nranks entries each).  _begin
 * queues the exchange on a second stream behind the work already queued on the library stream and returns; _end makes
 * the library stream wait for all exchanges queued since (no host wait).  A half sweep solved in K row chunks calls
 * _begin after each chunk and _end once, so chunk k travels over xGMI while chunk k+1 is being solved. */
int imp_comm_allgather_rows_begin(imp_comm *c, imp_matrix *full, const int64_t *row_lo, const int64_t *row_hi);
int imp_comm_allgather_rows_end(imp_comm *c);
/* Personalised exchange of row ranges (set-up of a sharded fit: pieces of the transposed shard).  Rows
 * [send_lo[p], send_hi[p]) of `send` go to rank p, where they arrive as rows [recv_lo[q], recv_hi[q]) of `recv` (q = the
 * sender); the two matrices must have rows of the same byte size, bytes travel untouched.  Synchronous. */
int imp_comm_alltoall_rows(imp_comm *c, const imp_matrix *send, const int64_t *send_lo, const int64_t *send_hi,
                           imp_matrix *recv, const int64_t *recv_lo, const int64_t *recv_hi);
int imp_comm_barrier(imp_comm *c);
/* out[0], out[1]: the number of ranks a one-per-rank sum all-reduce counts on the communicator of the library stream and on the
 * second communicator the pipelined exchange (allgather_rows_begin) runs on -- RCCL serialises the operations of one
 * communicator, so the gramian all-reduce and the row exchange never share one.  The benchmark driver asserts both == N. */
int imp_comm_ranks_seen(imp_comm *c, int *out);

/* ---- NEW: measurement hooks (bench.py's roofline leg) -------------------------------------------- */
/* When enabled every kernel launch is bracketed by HIP events on the library stream; totals are
 * accumulated per kernel name.  imp_prof_get returns 0 launches for unknown names. */
int imp_prof_enable(int on);
/* Restrict the event pairs to kernels whose name contains `substr` (NULL or "" = every kernel): an event pair costs a few
 * microseconds of stream time, so a timed region that only needs one kernel family should not pay for all of them. */
int imp_prof_filter(const char *substr);
int imp_prof_reset(void);
int imp_prof_get(const char *kernel, double *total_ms, int64_t *launches);
/* '\n'-separated list of kernel names seen so far, copied into buf (truncated to buflen). */
int imp_prof_names(char *buf, size_t buflen);

#ifdef __cplusplus
}
#endif
#endif /* IMPLICIT_HIP_H_ */
