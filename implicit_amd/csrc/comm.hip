// Multi-GPU exchange: one process per GPU, RCCL over xGMI.  No counterpart in the reference
// ("TODO: multi-gpu support", implicit/gpu/als.cu:169).  Only two collectives exist on the ALS data
// path (multi-GPU section): the f x f gramian all-reduce and the all-gather of the freshly
// solved factor-row shards; both run on the library stream, in place on replicated buffers.
#include <rccl/rccl.h>

#include <cstring>

#include "common.h"

#define IMP_CHECK_NCCL(expr)                                                                        \
  do {                                                                                              \
    ncclResult_t _r = (expr);                                                                       \
    if (_r != ncclSuccess) {                                                                        \
      throw std::runtime_error(std::string("RCCL error: ") + ncclGetErrorString(_r) + " (" +       \
                               __FILE__ + ":" + std::to_string(__LINE__) + ")");                  \
    }                                                                                               \
  } while (0)

struct imp_comm {
  ncclComm_t comm = nullptr;
  // A second communicator over the same ranks for the pipelined row exchange.  RCCL serialises the operations of ONE
  // communicator whatever streams they are queued on (and may add cross-stream dependencies of its own), so the gramian
  // all-reduce on the library stream and the grouped send / recv on `xchg_stream` must not share one: at best the overlap the
  // pipeline is built for disappears, at worst two ranks that interleave the two differently wait for each other.  Everything
  // queued on the library stream uses `comm`, everything on `xchg_stream` uses `xchg_comm`.
  ncclComm_t xchg_comm = nullptr;
  int nranks = 1, rank = 0;
  // pipelined exchange (allgather_rows_begin / _end): collectives are queued on their own stream behind an event of
  // the compute stream, so the solve of the next row chunk overlaps the exchange of the previous one
  hipStream_t xchg_stream = nullptr;
  hipEvent_t solved = nullptr, exchanged = nullptr;
  bool pending = false;
};

using namespace imp;

extern "C" {

static_assert(sizeof(ncclUniqueId) <= IMP_COMM_UNIQUE_ID_BYTES, "unique id does not fit");

int imp_comm_unique_id(void *id_out) {
  return guarded([&] {
    ncclUniqueId id;
    IMP_CHECK_NCCL(ncclGetUniqueId(&id));
    std::memset(id_out, 0, IMP_COMM_UNIQUE_ID_BYTES);
    std::memcpy(id_out, &id, sizeof(id));
  });
}

int imp_comm_init_rank(const void *id_bytes, int nranks, int rank, imp_comm **out) {
  return guarded([&] {
    if (nranks < 1 || rank < 0 || rank >= nranks) throw std::invalid_argument("invalid rank / nranks for imp_comm_init_rank");
    (void)ctx();
    auto c = std::make_unique<imp_comm>();
    c->nranks = nranks;
    c->rank = rank;
    ncclUniqueId id;
    std::memcpy(&id, id_bytes, sizeof(id));
    IMP_CHECK_NCCL(ncclCommInitRank(&c->comm, nranks, id, rank));
    // the exchange stream's communicator: a duplicate of the first (same ranks, same order).  ncclCommSplit with one colour is
    // the library's own way to get one; should this RCCL refuse it, rank 0 draws a second id and hands it round through the
    // first communicator (a 128-byte broadcast) for a plain ncclCommInitRank.
    if (ncclCommSplit(c->comm, 0, rank, &c->xchg_comm, nullptr) != ncclSuccess || !c->xchg_comm) {
      c->xchg_comm = nullptr;
      ncclUniqueId id2;
      std::memset(&id2, 0, sizeof(id2));
      if (rank == 0) IMP_CHECK_NCCL(ncclGetUniqueId(&id2));
      DeviceArray<unsigned char> buf;
      buf.alloc(sizeof(id2));
      IMP_CHECK_HIP(hipMemcpyAsync(buf.data(), &id2, sizeof(id2), hipMemcpyHostToDevice, stream()));
      IMP_CHECK_NCCL(ncclBroadcast(buf.data(), buf.data(), sizeof(id2), ncclChar, 0, c->comm, stream()));
      IMP_CHECK_HIP(hipMemcpyAsync(&id2, buf.data(), sizeof(id2), hipMemcpyDeviceToHost, stream()));
      sync();
      IMP_CHECK_NCCL(ncclCommInitRank(&c->xchg_comm, nranks, id2, rank));
    }
    IMP_CHECK_HIP(hipStreamCreateWithFlags(&c->xchg_stream, hipStreamNonBlocking));
    IMP_CHECK_HIP(hipEventCreateWithFlags(&c->solved, hipEventDisableTiming));
    IMP_CHECK_HIP(hipEventCreateWithFlags(&c->exchanged, hipEventDisableTiming));
    *out = c.release();
  });
}

int imp_comm_destroy(imp_comm *c) {
  return guarded([&] {
    if (c && c->xchg_stream) (void)hipStreamSynchronize(c->xchg_stream);
    if (c && c->xchg_comm) (void)ncclCommDestroy(c->xchg_comm);
    if (c && c->comm) (void)ncclCommDestroy(c->comm);
    if (c && c->solved) (void)hipEventDestroy(c->solved);
    if (c && c->exchanged) (void)hipEventDestroy(c->exchanged);
    if (c && c->xchg_stream) (void)hipStreamDestroy(c->xchg_stream);
    delete c;
  });
}

// All-gather of ragged row ranges as ONE group of point-to-point transfers: this rank's rows go to every peer and
// every peer's rows come in.  xGMI is a full mesh of point-to-point links (7 per GPU), so the direct exchange puts each
// shard on each link exactly once and all links work at the same time -- a ring would pipe every shard through the
// neighbours' links instead.
static void exchange_rows(imp_comm *c, ncclComm_t comm, imp_matrix *full, const int64_t *row_lo, const int64_t *row_hi, hipStream_t on) {
  note_device_write(full->data, full->bytes());
  const size_t row_bytes = full->cols * full->itemsize;
  char *base = reinterpret_cast<char *>(full->data);
  const size_t mine = (size_t)(row_hi[c->rank] - row_lo[c->rank]) * row_bytes;
  IMP_CHECK_NCCL(ncclGroupStart());
  for (int peer = 0; peer < c->nranks; ++peer) {
    if (peer == c->rank) continue;
    const size_t theirs = (size_t)(row_hi[peer] - row_lo[peer]) * row_bytes;
    if (mine) IMP_CHECK_NCCL(ncclSend(base + (size_t)row_lo[c->rank] * row_bytes, mine, ncclChar, peer, comm, on));
    if (theirs) IMP_CHECK_NCCL(ncclRecv(base + (size_t)row_lo[peer] * row_bytes, theirs, ncclChar, peer, comm, on));
  }
  IMP_CHECK_NCCL(ncclGroupEnd());
}

int imp_comm_allreduce_sum(imp_comm *c, imp_matrix *m) {
  return guarded([&] {
    if (m->itemsize != 4) throw std::invalid_argument("allreduce_sum needs a float32 matrix");
    note_device_write(m->data, m->bytes());
    IMP_PROF("rccl_allreduce");
    IMP_CHECK_NCCL(ncclAllReduce(m->data, m->data, m->rows * m->cols, ncclFloat, ncclSum, c->comm, stream()));
    sync_call();
  });
}

// Personalised exchange of row ranges (set-up path: every rank hands every other rank the piece of its transposed shard
// that falls into the other's item range).  Rows [send_lo[p], send_hi[p]) of `send` go to rank p and arrive there as rows
// [recv_lo[me], recv_hi[me]) of its `recv`; the piece a rank keeps for itself is a device copy.  Payload bytes travel
// untouched (ncclChar), so int32 columns may ride in an fp32 matrix.
int imp_comm_alltoall_rows(imp_comm *c, const imp_matrix *send, const int64_t *send_lo, const int64_t *send_hi, imp_matrix *recv,
                           const int64_t *recv_lo, const int64_t *recv_hi) {
  return guarded([&] {
    const size_t sb = send->cols * send->itemsize, rb = recv->cols * recv->itemsize;
    if (sb != rb) throw std::invalid_argument("alltoall_rows: send and recv rows differ in size");
    for (int p = 0; p < c->nranks; ++p) {
      if (send_lo[p] < 0 || send_hi[p] < send_lo[p] || (size_t)send_hi[p] > send->rows)
        throw std::invalid_argument("alltoall_rows: send range outside the matrix");
      if (recv_lo[p] < 0 || recv_hi[p] < recv_lo[p] || (size_t)recv_hi[p] > recv->rows)
        throw std::invalid_argument("alltoall_rows: recv range outside the matrix");
    }
    if (send_hi[c->rank] - send_lo[c->rank] != recv_hi[c->rank] - recv_lo[c->rank])
      throw std::invalid_argument("alltoall_rows: the piece a rank keeps must have the same size on both sides");
    note_device_write(recv->data, recv->bytes());
    IMP_PROF("rccl_alltoall_rows");
    const char *sbase = reinterpret_cast<const char *>(send->data);
    char *rbase = reinterpret_cast<char *>(recv->data);
    const size_t own = (size_t)(send_hi[c->rank] - send_lo[c->rank]) * sb;
    if (own)
      IMP_CHECK_HIP(hipMemcpyAsync(rbase + (size_t)recv_lo[c->rank] * rb, sbase + (size_t)send_lo[c->rank] * sb, own,
                                   hipMemcpyDeviceToDevice, stream()));
    IMP_CHECK_NCCL(ncclGroupStart());
    for (int peer = 0; peer < c->nranks; ++peer) {
      if (peer == c->rank) continue;
      const size_t out = (size_t)(send_hi[peer] - send_lo[peer]) * sb, in = (size_t)(recv_hi[peer] - recv_lo[peer]) * rb;
      if (out) IMP_CHECK_NCCL(ncclSend(sbase + (size_t)send_lo[peer] * sb, out, ncclChar, peer, c->comm, stream()));
      if (in) IMP_CHECK_NCCL(ncclRecv(rbase + (size_t)recv_lo[peer] * rb, in, ncclChar, peer, c->comm, stream()));
    }
    IMP_CHECK_NCCL(ncclGroupEnd());
    sync();
  });
}

int imp_comm_allgather_rows(imp_comm *c, imp_matrix *full, const int64_t *row_offsets) {
  return guarded([&] {
    if (row_offsets[0] != 0 || (size_t)row_offsets[c->nranks] != full->rows)
      throw std::invalid_argument("row_offsets must span [0, rows] for allgather_rows");
    IMP_PROF("rccl_allgather_rows");
    exchange_rows(c, c->comm, full, row_offsets, row_offsets + 1, stream());  // shards may be ragged (library stream: first communicator)
    sync();
  });
}

// Rows [row_lo[r], row_hi[r]) of `full` are owned by rank r.  Everything the library stream has queued so far (the
// solve that produced this rank's rows) is ordered before the exchange; the call returns without waiting.
int imp_comm_allgather_rows_begin(imp_comm *c, imp_matrix *full, const int64_t *row_lo, const int64_t *row_hi) {
  return guarded([&] {
    for (int r = 0; r < c->nranks; ++r)
      if (row_lo[r] < 0 || row_hi[r] < row_lo[r] || (size_t)row_hi[r] > full->rows)
        throw std::invalid_argument("row range outside the matrix in allgather_rows_begin");
    IMP_CHECK_HIP(hipEventRecord(c->solved, stream()));
    IMP_CHECK_HIP(hipStreamWaitEvent(c->xchg_stream, c->solved, 0));
    exchange_rows(c, c->xchg_comm, full, row_lo, row_hi, c->xchg_stream);  // its own communicator: never behind the gramian all-reduce
    c->pending = true;
  });
}

// Orders everything queued by _begin before whatever the library stream runs next (no host wait).
int imp_comm_allgather_rows_end(imp_comm *c) {
  return guarded([&] {
    if (!c->pending) return;
    IMP_CHECK_HIP(hipEventRecord(c->exchanged, c->xchg_stream));
    IMP_CHECK_HIP(hipStreamWaitEvent(stream(), c->exchanged, 0));
    c->pending = false;
  });
}

// Number of ranks the communicators actually connect: every rank contributes 1 to a sum all-reduce on each of the two (the
// benchmark driver asserts it equals the world size it was launched with; a mis-wired rendezvous would otherwise "scale" by
// running N independent replicas).  out[0]: the library stream's communicator, out[1]: the exchange stream's.
int imp_comm_ranks_seen(imp_comm *c, int *out) {
  return guarded([&] {
    DeviceArray<float> one;
    one.alloc(2);
    const float ones[2] = {1.f, 1.f};
    IMP_CHECK_HIP(hipMemcpyAsync(one.data(), ones, sizeof(ones), hipMemcpyHostToDevice, stream()));
    IMP_CHECK_NCCL(ncclAllReduce(one.data(), one.data(), 1, ncclFloat, ncclSum, c->comm, stream()));
    sync();  // (the second communicator's collective is queued only when the first one is done: no two in flight on one stream)
    IMP_CHECK_NCCL(ncclAllReduce(one.data() + 1, one.data() + 1, 1, ncclFloat, ncclSum, c->xchg_comm, stream()));
    float back[2];
    IMP_CHECK_HIP(hipMemcpyAsync(back, one.data(), sizeof(back), hipMemcpyDeviceToHost, stream()));
    sync();
    out[0] = (int)(back[0] + 0.5f), out[1] = (int)(back[1] + 0.5f);
  });
}

int imp_comm_barrier(imp_comm *c) {
  return guarded([&] {
    auto &word = ctx().barrier_word;
    if (word.size < 1) word.alloc(1, true);
    float *dummy = word.data();
    IMP_CHECK_NCCL(ncclAllReduce(dummy, dummy, 1, ncclFloat, ncclSum, c->comm, stream()));
    sync();
  });
}

}  // extern "C"
