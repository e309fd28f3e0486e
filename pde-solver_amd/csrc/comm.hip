// comm.hip — the library's own RCCL communicator: halo exchange and scalar all-reduce of the slab loop issued from C++ on HIP streams,
// without the Python / torch.distributed call path in the iteration.
//
// The reference has no multi-process code at all (its only MPI-adjacent line is fenics_mcp_server.py:141); what is distributed here is the
// linear solve inside its time loop (:702-716): slabs of node planes, one process per GPU.  Measured on one MI355X with a
// one-rank group (tools/dist_overhead.py): through torch.distributed the host spends 20 us per iteration on the one all-reduce call and 90 us
// on the batch of four point-to-point calls of an exchange — more than a rank of config 5 computes in an iteration (24-58 us).  RCCL is
// resolved at run time from whatever copy the process already holds (torch's bundled librccl.so.1 when torch is loaded, the system one
// otherwise: ONE RCCL and one HIP runtime per process, as for libamdhip64), so libpdegpu.so itself does not depend on it and the single-GPU
// tools never load it.
//
// Stream protocol of one exchange: the compute stream records `ready` once the planes to send are final; the communicator's own stream waits for
// it, runs ONE ncclGroup of sends and receives (the two xGMI links to the slab neighbours) and records `done`; the compute stream waits for `done`
// only in front of the launches that read ghost planes — everything enqueued in between overlaps the transfer.  The scalar all-reduce runs in
// place on the compute stream itself (nothing can overlap it: the next kernel needs its result).
//
// ONE ncclComm_t serves both: the grouped sends / receives go on the communicator's stream, the all-reduce on the compute stream.  RCCL serialises
// the operations of one communicator in the order they are ISSUED, whatever their streams, so this is correct exactly because every rank issues the
// same sequence (exchange group, all-reduce, exchange group, ... — pdegpu_session_iterate_slab and SlabCG.iterate are that sequence, and a rank never
// takes a data-dependent branch between two communications).  The first run on more than one GPU checks it with bench.py's single-GPU parity
// figures (`single_gpu_check`); until then it is rehearsed with one rank and the self-loop only.
//
// stdout is the reference server's JSON-RPC channel (fenics_mcp_server.py:41-52, 223-225): RCCL prints a version banner there when a communicator
// is initialised, so fd 1 points at stderr for the duration of ncclCommInitRank.
#include "common.h"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <unistd.h>
#include <cstdio>
#include <cstring>

namespace {
// fd 1 -> fd 2 while alive (process-wide: communicators are created collectively at start-up, not concurrently with a writer of stdout)
struct StdoutToStderr {
    int saved = -1;
    StdoutToStderr() {
        fflush(stdout);
        saved = dup(1);
        if (saved >= 0 && dup2(2, 1) < 0) {
            close(saved);
            saved = -1;
        }
    }
    ~StdoutToStderr() {
        if (saved < 0) return;
        fflush(stdout);                      // what the library printed sits in stdio's buffer: it must leave while fd 1 still is stderr
        (void)dup2(saved, 1);
        close(saved);
    }
};

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

// the RCCL the process already holds (RTLD_NOLOAD), else the one the loader finds
RcclApi* rccl_api() {
    static RcclApi api;
    static bool tried = false;
    if (tried) return &api;
    tried = true;
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
        api.handle = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
        if (api.handle) break;
    }
    if (!api.handle)
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            api.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (api.handle) break;
        }
    if (!api.handle) {
        const char* why = dlerror();                          // (a second call would return NULL: dlerror clears its state)
        api.error = std::string("librccl.so.1 cannot be loaded: ") + (why ? why : "unknown");
        return &api;
    }
    auto sym = [&](const char* n) {
        void* p = dlsym(api.handle, n);
        if (!p && api.error.empty()) api.error = std::string("RCCL symbol missing: ") + n;
        return p;
    };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
    api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
    api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
    api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    return &api;
}
}  // namespace

struct pdegpu_comm {
    pdegpu_ctx* ctx = nullptr;
    RcclApi* api = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    hipStream_t cstream = nullptr;          // the exchange runs here, beside the compute stream
    hipEvent_t ev_ready = nullptr, ev_done = nullptr;
    bool self_loop = false;                 // one-GPU rehearsal: both slab neighbours are this rank itself
};

#define PG_NCCL(c, call)                                                                                         \
    do {                                                                                                         \
        ncclResult_t r__ = (call);                                                                               \
        if (r__ != ncclSuccess)                                                                                  \
            return pdegpu_set_err((c)->ctx, PDEGPU_EHIP, "%s failed: %s (%s:%d)", #call,                         \
                                  (c)->api->GetErrorString ? (c)->api->GetErrorString(r__) : "?", __FILE__, __LINE__); \
    } while (0)

extern "C" int pdegpu_comm_unique_id(void* id128_out) {
    if (!id128_out) return PDEGPU_EINVAL;
    RcclApi* api = rccl_api();
    if (!api->error.empty() || !api->GetUniqueId) return PDEGPU_EUNSUPPORTED;
    ncclUniqueId id;
    if (api->GetUniqueId(&id) != ncclSuccess) return PDEGPU_EHIP;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId");
    memcpy(id128_out, &id, sizeof id);
    return PDEGPU_OK;
}

extern "C" int pdegpu_comm_create(pdegpu_ctx* ctx, int rank, int world, const void* id128, pdegpu_comm** out) {
    if (!ctx) return PDEGPU_EINVAL;
    if (!out || !id128 || world < 1 || rank < 0 || rank >= world) return pdegpu_set_err(ctx, PDEGPU_EINVAL, "pdegpu_comm_create: bad argument");
    *out = nullptr;
    RcclApi* api = rccl_api();
    if (!api->error.empty()) return pdegpu_set_err(ctx, PDEGPU_EUNSUPPORTED, "%s", api->error.c_str());
    PG_HIP(ctx, hipSetDevice(ctx->device));
    pdegpu_comm* c = new pdegpu_comm();
    c->ctx = ctx;
    c->api = api;
    c->rank = rank;
    c->world = world;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclResult_t r;
    {
        StdoutToStderr quiet;
        r = api->CommInitRank(&c->comm, world, id, rank);
    }
    if (r != ncclSuccess) {
        const int rc = pdegpu_set_err(ctx, PDEGPU_EHIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, api->GetErrorString ? api->GetErrorString(r) : "?");
        delete c;
        return rc;
    }
    // (hipEventReleaseToDevice on the two events changes nothing measurable: 0.1983 vs 0.1982 ms per iteration of a rank of 8 on 512^3)
    if (hipStreamCreateWithFlags(&c->cstream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming) != hipSuccess) {
        pdegpu_comm_destroy(c);
        return pdegpu_set_err(ctx, PDEGPU_EHIP, "pdegpu_comm_create: stream / event creation failed");
    }
    *out = c;
    return PDEGPU_OK;
}

extern "C" void pdegpu_comm_destroy(pdegpu_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->ctx->device);
    if (c->cstream) (void)hipStreamSynchronize(c->cstream);
    if (c->comm && c->api->CommDestroy) (void)c->api->CommDestroy(c->comm);
    if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
    if (c->ev_done) (void)hipEventDestroy(c->ev_done);
    if (c->cstream) (void)hipStreamDestroy(c->cstream);
    delete c;
}

// One-GPU rehearsal of a rank's iteration (tools/slab_surrogate.py): with on != 0 a slab's two neighbours are this rank itself — its first / last
// owned plane come back into its own ghost planes as real RCCL point-to-point messages of the real size, and iterate_slab exchanges although the
// communicator has one rank.  The iterates mean nothing then; the launches, the stream protocol and the messages are those of a rank in the middle.
extern "C" int pdegpu_comm_set_self_loop(pdegpu_comm* c, int on) {
    if (!c) return PDEGPU_EINVAL;
    c->self_loop = on != 0;
    return PDEGPU_OK;
}
extern "C" int pdegpu_comm_rank(const pdegpu_comm* c) { return c ? c->rank : -1; }
extern "C" int pdegpu_comm_world(const pdegpu_comm* c) { return c ? c->world : -1; }

// for hosts that group the messages of several sessions themselves (tests: two slabs of one process exchanging through sends to self)
extern "C" int pdegpu_comm_group_begin(pdegpu_comm* c) {
    if (!c) return PDEGPU_EINVAL;
    PG_HIP(c->ctx, hipSetDevice(c->ctx->device));
    PG_TRY(comm_ready(c, c->ctx->stream));
    return comm_group_start(c);
}
extern "C" int pdegpu_comm_group_end(pdegpu_comm* c) {
    if (!c) return PDEGPU_EINVAL;
    PG_TRY(comm_group_end(c));
    return comm_done(c);
}

// ---- pieces the session code composes (heat.hip, elasticity.hip) ------------------------------------------------------------------------
int comm_rank(const pdegpu_comm* c) { return c->rank; }
int comm_world(const pdegpu_comm* c) { return c->world; }
bool comm_self_loop(const pdegpu_comm* c) { return c->self_loop; }

// the planes to send are final on `compute` from here on: the exchange stream may start
int comm_ready(pdegpu_comm* c, hipStream_t compute) {
    PG_HIP(c->ctx, hipEventRecord(c->ev_ready, compute));
    PG_HIP(c->ctx, hipStreamWaitEvent(c->cstream, c->ev_ready, 0));
    return PDEGPU_OK;
}
int comm_group_start(pdegpu_comm* c) {
    PG_NCCL(c, c->api->GroupStart());
    return PDEGPU_OK;
}
int comm_group_end(pdegpu_comm* c) {
    PG_NCCL(c, c->api->GroupEnd());
    return PDEGPU_OK;
}
int comm_send(pdegpu_comm* c, const double* buf, size_t n, int peer) {
    if (peer < 0 || peer >= c->world) return pdegpu_set_err(c->ctx, PDEGPU_EINVAL, "send to rank %d of %d", peer, c->world);
    PG_NCCL(c, c->api->Send(buf, n, ncclDouble, peer, c->comm, c->cstream));
    return PDEGPU_OK;
}
int comm_recv(pdegpu_comm* c, double* buf, size_t n, int peer) {
    if (peer < 0 || peer >= c->world) return pdegpu_set_err(c->ctx, PDEGPU_EINVAL, "receive from rank %d of %d", peer, c->world);
    PG_NCCL(c, c->api->Recv(buf, n, ncclDouble, peer, c->comm, c->cstream));
    return PDEGPU_OK;
}
// the exchange posted so far is complete once `done` fires
int comm_done(pdegpu_comm* c) {
    PG_HIP(c->ctx, hipEventRecord(c->ev_done, c->cstream));
    return PDEGPU_OK;
}
// launches enqueued on `compute` after this call see the received planes
int comm_wait(pdegpu_comm* c, hipStream_t compute) {
    PG_HIP(c->ctx, hipStreamWaitEvent(compute, c->ev_done, 0));
    return PDEGPU_OK;
}
// sum over the ranks, in place, on the compute stream
int comm_allreduce(pdegpu_comm* c, double* buf, size_t n, hipStream_t compute) {
    PG_NCCL(c, c->api->AllReduce(buf, buf, n, ncclDouble, ncclSum, c->comm, compute));
    return PDEGPU_OK;
}
