// transport_rccl.cpp -- libstcsp_rccl.so: the stcsp_transport (include/stcsp_sharded.h) over RCCL, one process per GPU.
// Kept out of libstcsp_hip.so so that the engine library has no RCCL dependency (a host that brings its own transport --
// MPI, the in-process one -- never loads librccl).
//
//   all_gather_i64 / _bytes   ncclAllGather on a small device staging buffer + one stream synchronisation (the host needs
//                             the values: they size the next exchange and decide termination)
//   all_to_all_v              grouped ncclSend / ncclRecv on the ENGINE's stream: ordered behind the kernels that produced
//                             the records and in front of the commit / adopt kernels that consume them, no host wait.
//                             xGMI is point to point (every peer one direct hop), the messages are KB..MB: latency bound.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <string>

#include "stcsp_sharded.h"

namespace {
struct Rccl {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;  // for the gathers (the record exchanges run on the engine's stream)
    char *d_stage = nullptr;
    size_t stage_bytes = 0;
    std::string err;
    stcsp_transport api{};
    int fail(const char *what, const char *why) {
        err = std::string(what) + ": " + why;
        return -1;
    }
};
#define HIPT(call)                                                           \
    do {                                                                     \
        hipError_t e_ = (call);                                              \
        if (e_ != hipSuccess) return t->fail(#call, hipGetErrorString(e_)); \
    } while (0)
#define NCCLT(call)                                                            \
    do {                                                                       \
        ncclResult_t r_ = (call);                                              \
        if (r_ != ncclSuccess) return t->fail(#call, ncclGetErrorString(r_)); \
    } while (0)

int gather_bytes(void *self, const void *mine, int64_t n, void *all) {
    Rccl *t = (Rccl *)self;
    if (n <= 0) return 0;
    HIPT(hipSetDevice(t->device));
    const size_t need = (size_t)n * (size_t)(t->world + 1);
    if (t->stage_bytes < need) {
        if (t->d_stage) HIPT(hipFree(t->d_stage));
        t->d_stage = nullptr;
        HIPT(hipMalloc((void **)&t->d_stage, need * 2));
        t->stage_bytes = need * 2;
    }
    char *d_mine = t->d_stage, *d_all = t->d_stage + n;
    HIPT(hipMemcpyAsync(d_mine, mine, (size_t)n, hipMemcpyHostToDevice, t->stream));
    NCCLT(ncclAllGather(d_mine, d_all, (size_t)n, ncclInt8, t->comm, t->stream));
    HIPT(hipMemcpyAsync(all, d_all, (size_t)n * t->world, hipMemcpyDeviceToHost, t->stream));
    HIPT(hipStreamSynchronize(t->stream));
    return 0;
}
int gather_i64(void *self, const int64_t *mine, int32_t n, int64_t *all) { return gather_bytes(self, mine, (int64_t)n * 8, all); }
int all_to_all_v(void *self, const void *send, const int64_t *send_words, void *recv, const int64_t *recv_words, void *stream) {
    Rccl *t = (Rccl *)self;
    HIPT(hipSetDevice(t->device));
    const uint32_t *s = (const uint32_t *)send;
    uint32_t *r = (uint32_t *)recv;
    NCCLT(ncclGroupStart());
    for (int p = 0; p < t->world; p++) {
        if (send_words[p]) NCCLT(ncclSend(s, (size_t)send_words[p], ncclUint32, p, t->comm, (hipStream_t)stream));
        if (recv_words[p]) NCCLT(ncclRecv(r, (size_t)recv_words[p], ncclUint32, p, t->comm, (hipStream_t)stream));
        s += send_words[p];
        r += recv_words[p];
    }
    NCCLT(ncclGroupEnd());
    return 0;
}
const char *last_error(void *self) { return ((Rccl *)self)->err.c_str(); }
}  // namespace

extern "C" {
int stcsp_rccl_unique_id(void *id_out) {
    static_assert(sizeof(ncclUniqueId) <= STCSP_RCCL_ID_BYTES, "unique id size");
    if (!id_out) return STCSP_E_INVALID;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return STCSP_E_DEVICE;
    memset(id_out, 0, STCSP_RCCL_ID_BYTES);
    memcpy(id_out, &id, sizeof id);
    return STCSP_OK;
}
int stcsp_transport_rccl_create(const void *unique_id, int32_t rank, int32_t world, int32_t device, stcsp_transport **out) {
    if (!unique_id || !out || world < 1 || rank < 0 || rank >= world) return STCSP_E_INVALID;
    Rccl *t = new Rccl();
    t->rank = rank;
    t->world = world;
    t->device = device;
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof id);
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking) != hipSuccess ||
        ncclCommInitRank(&t->comm, world, id, rank) != ncclSuccess) {
        delete t;
        return STCSP_E_DEVICE;
    }
    t->api = stcsp_transport{t, rank, world, gather_i64, gather_bytes, all_to_all_v, last_error};
    *out = &t->api;
    return STCSP_OK;
}
void stcsp_transport_rccl_destroy(stcsp_transport *tr) {
    if (!tr) return;
    Rccl *t = (Rccl *)tr->self;
    (void)hipSetDevice(t->device);
    if (t->stream) (void)hipStreamSynchronize(t->stream);
    if (t->comm) (void)ncclCommDestroy(t->comm);
    if (t->d_stage) (void)hipFree(t->d_stage);
    if (t->stream) (void)hipStreamDestroy(t->stream);
    delete t;
}
}
