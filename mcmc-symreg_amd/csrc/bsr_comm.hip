// The one exchange of the multi-GPU path: an all-gather of fixed-size chain records through RCCL on the context's
// stream (include/bsr_hip.h: bsr_comm_*).
#include "bsr_ctx.h"

// ---------------------------------------------------------------------------------------------------------------
// RCCL: one communicator per process/GPU, one all-gather of fixed-size accepted-tree records.
extern "C" int bsr_comm_unique_id(void* id128) {
  if (!id128) return BSR_E_ARG;
  static_assert(sizeof(ncclUniqueId) <= BSR_COMM_ID_BYTES, "ncclUniqueId larger than BSR_COMM_ID_BYTES");
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return BSR_E_COMM;
  memset(id128, 0, BSR_COMM_ID_BYTES);
  memcpy(id128, &id, sizeof id);
  return BSR_OK;
}

extern "C" int bsr_comm_init(bsr_ctx* c, int32_t nranks, int32_t rank, const void* id128) {
  if (!c || !id128 || nranks <= 0 || rank < 0 || rank >= nranks) return BSR_E_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  ncclUniqueId id;
  memcpy(&id, id128, sizeof id);
  ncclResult_t r = ncclCommInitRank(&c->comm, nranks, id, rank);
  if (r != ncclSuccess) {
    c->err = std::string("ncclCommInitRank: ") + ncclGetErrorString(r);
    c->comm = nullptr;
    return BSR_E_COMM;
  }
  return BSR_OK;
}

extern "C" int bsr_comm_allgather(bsr_ctx* c, const void* send, void* recv, int64_t bytes_per_rank) {
  if (!c || !send || !recv || bytes_per_rank <= 0) return BSR_E_ARG;
  if (!c->comm) return fail(c, BSR_E_STATE, "bsr_comm_allgather: communicator not initialised");
  HIPCHK(c, hipSetDevice(c->device));
  int nranks = 0;
  ncclCommCount(c->comm, &nranks);
  const size_t need = (size_t)bytes_per_rank * (nranks + 1);
  if (need > c->comm_cap) {
    if (c->comm_buf) HIPCHK(c, hipFree(c->comm_buf));
    c->comm_buf = nullptr;
    HIPCHK(c, hipMalloc(&c->comm_buf, need));
    c->comm_cap = need;
  }
  char* dsend = (char*)c->comm_buf;
  char* drecv = dsend + bytes_per_rank;
  HIPCHK(c, hipMemcpyAsync(dsend, send, bytes_per_rank, hipMemcpyHostToDevice, c->stream));
  ncclResult_t r = ncclAllGather(dsend, drecv, (size_t)bytes_per_rank, ncclChar, c->comm, c->stream);
  if (r != ncclSuccess) {
    c->err = std::string("ncclAllGather: ") + ncclGetErrorString(r);
    return BSR_E_COMM;
  }
  HIPCHK(c, hipMemcpyAsync(recv, drecv, (size_t)bytes_per_rank * nranks, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return BSR_OK;
}

extern "C" int bsr_comm_destroy(bsr_ctx* c) {
  if (!c) return BSR_E_ARG;
  if (c->comm) {
    ncclCommDestroy(c->comm);
    c->comm = nullptr;
  }
  return BSR_OK;
}
