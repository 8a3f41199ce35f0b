// spfe_comm.hip — multi-GPU: ONE RCCL all-gather of the fixed-stride records per batch, inside the C ABI.
#include "spfe_host.h"
using namespace spfe_host;

extern "C" {

// ---- multi-GPU: RCCL all-gather of the records -----------------------------------
namespace {
void *open_rccl() {
  // an already loaded librccl (e.g. the one torch.distributed brought) is reused by soname; the handle is kept for the
  // life of the process (one dlopen, never closed: communicators may outlive any one extractor handle)
  static void *const lib = []() -> void * {   // (function-local static: initialised once, also under concurrent first calls)
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
      if (void *l = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) return l;
    return nullptr;
  }();
  return lib;
}
}  // namespace

int spfe_comm_unique_id(void *id, size_t cap) {
  if (!id || cap < NCCL_UNIQUE_ID_BYTES) return fail(SPFE_EINVAL, "unique id buffer must hold %d bytes", NCCL_UNIQUE_ID_BYTES);
  void *lib = open_rccl();
  if (!lib) return fail(SPFE_EHIP, "librccl not found: %s", dlerror());
  auto get = reinterpret_cast<pfn_ncclGetUniqueId>(dlsym(lib, "ncclGetUniqueId"));
  auto err = reinterpret_cast<pfn_ncclGetErrorString>(dlsym(lib, "ncclGetErrorString"));
  if (!get || !err) return fail(SPFE_EHIP, "librccl lacks ncclGetUniqueId");
  ncclUniqueId u;
  const ncclResult_t r = get(&u);
  if (r != ncclSuccess) return fail(SPFE_EHIP, "ncclGetUniqueId: %s", err(r));
  memcpy(id, &u, NCCL_UNIQUE_ID_BYTES);
  return SPFE_OK;
}

int spfe_comm_init(spfe_handle h, const void *id, int rank, int world) {
  if (!h || !id) return fail(SPFE_EINVAL, "null argument");
  if (world < 1 || rank < 0 || rank >= world) return fail(SPFE_EINVAL, "rank %d / world %d", rank, world);
  if (h->comm) return fail(SPFE_EINVAL, "communicator already initialised (spfe_comm_destroy first)");
  HIP_TRY(hipSetDevice(h->cfg.device));
  if (!h->rccl_lib) {
    h->rccl_lib = open_rccl();
    if (!h->rccl_lib) return fail(SPFE_EHIP, "librccl not found: %s", dlerror());
    h->p_ncclCommInitRank = reinterpret_cast<pfn_ncclCommInitRank>(dlsym(h->rccl_lib, "ncclCommInitRank"));
    h->p_ncclCommDestroy = reinterpret_cast<pfn_ncclCommDestroy>(dlsym(h->rccl_lib, "ncclCommDestroy"));
    h->p_ncclCommCount = reinterpret_cast<pfn_ncclCommCount>(dlsym(h->rccl_lib, "ncclCommCount"));
    h->p_ncclAllGather = reinterpret_cast<pfn_ncclAllGather>(dlsym(h->rccl_lib, "ncclAllGather"));
    h->p_ncclGetErrorString = reinterpret_cast<pfn_ncclGetErrorString>(dlsym(h->rccl_lib, "ncclGetErrorString"));
    if (!h->p_ncclCommInitRank || !h->p_ncclCommDestroy || !h->p_ncclCommCount || !h->p_ncclAllGather || !h->p_ncclGetErrorString)
      return fail(SPFE_EHIP, "librccl lacks a required entry point");
  }
  ncclUniqueId u;
  memcpy(&u, id, NCCL_UNIQUE_ID_BYTES);
  const ncclResult_t r = h->p_ncclCommInitRank(&h->comm, world, u, rank);
  if (r != ncclSuccess) {
    h->comm = nullptr;
    return fail(SPFE_EHIP, "ncclCommInitRank(rank %d of %d, device %d): %s", rank, world, h->cfg.device,
                h->p_ncclGetErrorString(r));
  }
  // The collective runs on the SIDE stream, behind the covariance kernels of the batch it gathers (call
  // spfe_allgather_records for batch i before enqueueing batch i + 1, as parallel.ShardedExtractor does): no stream sits in
  // a hardware queue waiting for the covariance event.  HIP maps streams onto a few hardware queues; a waiting stream that
  // lands on the compute stream's queue holds the NEXT batch's convolutions back (measured on the host path: half the
  // throughput).  SPFE_COMM_OWN_STREAM=1: a communication stream of its own that waits for the batch's event.
  // (the one switch that is not the handle's but the communicator's: read each time one is made, so that ONE handle can run
  // both forms — bench.py's comm_stream_ab leg)
  h->comm_own_stream = getenv("SPFE_COMM_OWN_STREAM") && atoi(getenv("SPFE_COMM_OWN_STREAM")) != 0;
  if (!h->comm_stream) {
    if (h->comm_own_stream) HIP_TRY(hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
    else h->comm_stream = h->side;
  }
  if (!h->ev_gather) HIP_TRY(hipEventCreateWithFlags(&h->ev_gather, hipEventDisableTiming));
  h->comm_rank = rank;
  h->comm_world = world;
  h->gather_recorded = false;
  return SPFE_OK;
}

int spfe_comm_destroy(spfe_handle h) {
  if (!h) return fail(SPFE_EINVAL, "null handle");
  if (h->comm_stream) (void)hipStreamSynchronize(h->comm_stream);
  if (h->comm && h->p_ncclCommDestroy) (void)h->p_ncclCommDestroy(h->comm);
  h->comm = nullptr;
  if (h->ev_gather) { (void)hipEventDestroy(h->ev_gather); h->ev_gather = nullptr; }
  if (h->comm_stream && h->comm_own_stream) (void)hipStreamDestroy(h->comm_stream);
  h->comm_stream = nullptr;
  h->comm_world = 0;
  h->gather_recorded = false;
  return SPFE_OK;
}

void *spfe_comm_stream(spfe_handle h) { return h ? reinterpret_cast<void *>(h->comm_stream) : nullptr; }

int spfe_comm_count(spfe_handle h, int *count) {
  if (!h || !count) return fail(SPFE_EINVAL, "null argument");
  if (!h->comm) return fail(SPFE_EINVAL, "spfe_comm_init has not been called");
  const ncclResult_t r = h->p_ncclCommCount(h->comm, count);   // what RCCL itself says, not what the caller passed in
  if (r != ncclSuccess) return fail(SPFE_EHIP, "ncclCommCount: %s", h->p_ncclGetErrorString(r));
  return SPFE_OK;
}

int spfe_allgather_records(spfe_handle h, long ticket, const void *d_local, void *d_all, int frames_per_rank) {
  if (!h || !d_local || !d_all) return fail(SPFE_EINVAL, "null argument");
  if (!h->comm) return fail(SPFE_EINVAL, "spfe_comm_init has not been called");
  if (frames_per_rank < 1) return fail(SPFE_EINVAL, "frames_per_rank %d", frames_per_rank);
  if (ticket < 0 || ticket >= api_tickets(h) || ticket + spfe_handle_s::NTICKET <= api_tickets(h))
    return fail(SPFE_EINVAL, "ticket %ld is not one of the last %d calls", ticket, spfe_handle_s::NTICKET);
  HIP_TRY(hipSetDevice(h->cfg.device));
  // on the side stream the gather simply follows the batch's covariance kernels (and everything enqueued there since:
  // gather batch i before enqueueing batch i + 1); a stream of its own waits for exactly this batch's records.  Either
  // way the gather of batch i runs beside the convolutions of batch i + 1
  // (always: in pipelined calls the covariance kernels sit on the side stream in front of the gather and the event has been
  // recorded there — a wait that is satisfied when it is reached; in synchronous calls the chain runs on the launch stream
  // (round 4) and this wait is what orders the gather behind it)
  const spfe_handle_s::TicketRef tr = ticket_ref(h, ticket);   // (a handle with a twin: whichever of the two ran that call)
  HIP_TRY(hipStreamWaitEvent(h->comm_stream, tr.who->ev_cov[tr.local % spfe_handle_s::NTICKET], 0));
  const size_t count = (size_t)frames_per_rank * h->rl.bytes;   // bytes as ncclUint8; RCCL counts are size_t
  const ncclResult_t r = h->p_ncclAllGather(d_local, d_all, count, ncclUint8, h->comm, h->comm_stream);
  if (r != ncclSuccess) return fail(SPFE_EHIP, "ncclAllGather(%zu bytes per rank): %s", count, h->p_ncclGetErrorString(r));
  HIP_TRY(hipEventRecord(h->ev_gather, h->comm_stream));
  h->gather_recorded = true;
  return SPFE_OK;
}

int spfe_comm_wait(spfe_handle h, void *stream) {
  if (!h) return fail(SPFE_EINVAL, "null handle");
  if (!h->gather_recorded) return SPFE_OK;
  HIP_TRY(hipSetDevice(h->cfg.device));
  hipStream_t s = stream ? reinterpret_cast<hipStream_t>(stream) : h->stream;
  HIP_TRY(hipStreamWaitEvent(s, h->ev_gather, 0));
  return SPFE_OK;
}

}  // extern "C"
