// comm.cpp -- the one collective of the path, in the C ABI: a gather of the per-rank results on a
// root rank over RCCL / xGMI (independent frames are sharded over GPUs, "one
// RCCL gather of the final estimates at the end"), plus the barrier / max-reduction a timing harness
// needs.  One process per GPU; the 128-byte RCCL unique id travels between the processes by whatever
// means the host has (rescan_line_sted_amd/sharding.py: a file next to the launcher).
//
// librccl.so (570 MB) is loaded on first use only: single-GPU users of librlsted.so never pay for it.
// The gather is direct: every rank sends to the root, the root posts all receives in one group, so
// all 7 xGMI links of the root carry data at once (a ring would be bound by one link).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <string>
#include <vector>

#include "ctx.hpp"

namespace {

struct Rccl {
    void* so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

static void rccl_load(Rccl& r);
Rccl* rccl() {   // loaded once, whichever thread asks first (function-local static: initialisation is thread safe)
    static Rccl r = [] {
        Rccl x;
        rccl_load(x);
        return x;
    }();
    return &r;
}
static void rccl_load(Rccl& r) {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r.so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (r.so) break;
    }
    if (!r.so) {
        r.error = std::string("cannot load librccl.so: ") + dlerror();
        return;
    }
#define RL_SYM(field, sym)                                                  \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.so, sym));       \
    if (!r.field) r.error = std::string("librccl.so lacks ") + sym;
    RL_SYM(GetUniqueId, "ncclGetUniqueId")
    RL_SYM(CommInitRank, "ncclCommInitRank")
    RL_SYM(CommDestroy, "ncclCommDestroy")
    RL_SYM(GroupStart, "ncclGroupStart")
    RL_SYM(GroupEnd, "ncclGroupEnd")
    RL_SYM(Send, "ncclSend")
    RL_SYM(Recv, "ncclRecv")
    RL_SYM(AllReduce, "ncclAllReduce")
    RL_SYM(Broadcast, "ncclBroadcast")
    RL_SYM(GetErrorString, "ncclGetErrorString")
#undef RL_SYM
}

// ncclGroupStart ... ncclGroupEnd around a block of sends / receives.  The group is ALWAYS closed when the scope is
// left, error returns included: an open group on this thread would swallow every later RCCL call.
struct GroupScope {
    Rccl* r;
    bool open = false;
    explicit GroupScope(Rccl* r_) : r(r_) {}
    ncclResult_t start() {
        const ncclResult_t rc = r->GroupStart();
        open = rc == ncclSuccess;
        return rc;
    }
    ncclResult_t end() {
        open = false;
        return r->GroupEnd();
    }
    ~GroupScope() {
        if (open) (void)r->GroupEnd();
    }
};

#define RCCL_TRY(expr)                                                                                       \
    do {                                                                                                     \
        ncclResult_t r_ = (expr);                                                                            \
        if (r_ != ncclSuccess) return rl::fail(RL_ERR_HIP, std::string(#expr) + ": " + rccl()->GetErrorString(r_)); \
    } while (0)

}  // namespace

struct rl_comm {
    rl_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    void* gathered = nullptr;   // root: the last plan gather's result (device); what rl_gather_device hands out
    size_t gathered_bytes = 0;
    void* staging = nullptr;    // rl_comm_gather_host's own device staging (never aliases `gathered`)
    size_t staging_bytes = 0;
    double* scalar = nullptr;   // device scratch for the reductions
};

extern "C" {

int rl_comm_unique_id(void* id128) {
    if (!id128) return fail(RL_ERR_INVALID, "id128 is NULL");
    Rccl* r = rccl();
    if (!r->error.empty()) return fail(RL_ERR_UNSUPPORTED, r->error);
    ncclUniqueId id;
    RCCL_TRY(r->GetUniqueId(&id));
    static_assert(sizeof(id) == RL_COMM_ID_BYTES, "RCCL unique id size");
    memcpy(id128, &id, sizeof(id));
    return RL_OK;
}

int rl_comm_create(rl_ctx* ctx, int rank, int world, const void* id128, rl_comm** out) {
    if (!ctx || !id128 || !out) return fail(RL_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail(RL_ERR_INVALID, "bad rank / world size");
    Rccl* r = rccl();
    if (!r->error.empty()) return fail(RL_ERR_UNSUPPORTED, r->error);
    HIP_TRY(hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    rl_comm* c = new rl_comm;
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    ncclResult_t rc = r->CommInitRank(&c->comm, world, id, rank);
    if (rc != ncclSuccess) {
        delete c;
        return fail(RL_ERR_HIP, std::string("ncclCommInitRank: ") + r->GetErrorString(rc));
    }
    hipError_t e = hipMalloc((void**)&c->scalar, 2 * sizeof(double));
    if (e != hipSuccess) {
        r->CommDestroy(c->comm);
        delete c;
        return fail(RL_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
    }
    *out = c;
    return RL_OK;
}

int rl_comm_destroy(rl_comm* c) {
    if (!c) return RL_OK;
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    if (c->comm) rccl()->CommDestroy(c->comm);
    if (c->gathered) (void)hipFree(c->gathered);
    if (c->staging) (void)hipFree(c->staging);
    if (c->scalar) (void)hipFree(c->scalar);
    delete c;
    return RL_OK;
}

int rl_comm_info(const rl_comm* c, int* rank, int* world) {
    if (!c) return fail(RL_ERR_INVALID, "comm is NULL");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    return RL_OK;
}

int rl_comm_allreduce_max(rl_comm* c, double* value) {
    if (!c || !value) return fail(RL_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(c->ctx->device));
    hipStream_t s = c->ctx->stream;
    HIP_TRY(hipMemcpyAsync(c->scalar, value, sizeof(double), hipMemcpyHostToDevice, s));
    RCCL_TRY(rccl()->AllReduce(c->scalar, c->scalar + 1, 1, ncclDouble, ncclMax, c->comm, s));
    HIP_TRY(hipMemcpyAsync(value, c->scalar + 1, sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return RL_OK;
}

int rl_comm_barrier(rl_comm* c) {
    if (!c) return fail(RL_ERR_INVALID, "comm is NULL");
    HIP_TRY(hipSetDevice(c->ctx->device));
    HIP_TRY(hipDeviceSynchronize());   // everything this rank has queued, on any stream
    double one = 1.0;
    RL_TRY(rl_comm_allreduce_max(c, &one));
    return RL_OK;
}

// counts[r] frames from every rank r, concatenated rank-major on the root.
static int gather_impl(rl_comm* c, rl_deconv* plan, int which, int root, const int* counts) {
    if (!c || !plan || !counts) return fail(RL_ERR_INVALID, "NULL argument");
    if (root < 0 || root >= c->world) return fail(RL_ERR_INVALID, "no such root rank");
    int B = 0, V = 0, ny = 0, nx = 0;
    RL_TRY(rl_deconv_dims(plan, &B, &V, &ny, &nx));
    void* src = nullptr;
    size_t n_elems = 0;
    int dtype = RL_F32;
    RL_TRY(rl_deconv_device_ptr(plan, which, &src, &n_elems, &dtype));
    const size_t per_frame = n_elems / (size_t)B;   // elements of one frame (views included)
    if (counts[c->rank] < 0 || counts[c->rank] > B) return fail(RL_ERR_INVALID, "counts[rank] exceeds the plan's batch");
    const size_t es = dtype == RL_F32 ? 4 : 8;
    const ncclDataType_t nt = dtype == RL_F32 ? ncclFloat : ncclDouble;
    HIP_TRY(hipSetDevice(c->ctx->device));
    hipStream_t s = c->ctx->stream;
    HIP_TRY(hipDeviceSynchronize());   // the plan's slice streams have finished writing the buffer
    Rccl* r = rccl();
    if (c->rank == root) {
        size_t total = 0;
        for (int k = 0; k < c->world; ++k) {
            if (counts[k] < 0) return fail(RL_ERR_INVALID, "negative count");
            total += (size_t)counts[k];
        }
        const size_t need = total * per_frame * es;
        if (need > c->gathered_bytes) {
            if (c->gathered) HIP_TRY(hipFree(c->gathered));
            c->gathered = nullptr;
            c->gathered_bytes = 0;
            HIP_TRY(hipMalloc(&c->gathered, need ? need : 1));
            c->gathered_bytes = need;
        }
        GroupScope group(r);
        RCCL_TRY(group.start());
        size_t off = 0;
        for (int k = 0; k < c->world; ++k) {
            const size_t n = (size_t)counts[k] * per_frame;
            char* dst = (char*)c->gathered + off * es;
            if (k == root) {
                HIP_TRY(hipMemcpyAsync(dst, src, n * es, hipMemcpyDeviceToDevice, s));
            } else if (n) {
                RCCL_TRY(r->Recv(dst, n, nt, k, c->comm, s));
            }
            off += n;
        }
        RCCL_TRY(group.end());
    } else {
        const size_t n = (size_t)counts[c->rank] * per_frame;
        if (n) {
            GroupScope group(r);
            RCCL_TRY(group.start());
            RCCL_TRY(r->Send(src, n, nt, root, c->comm, s));
            RCCL_TRY(group.end());
        }
    }
    HIP_TRY(hipStreamSynchronize(s));
    return RL_OK;
}

// Host arrays of any size: counts[r] float64 values from rank r, rank-major on the root, staged through
// device buffers (what the sweep harness sends: padded result stacks that no longer live in a plan).
int rl_comm_gather_host(rl_comm* c, const double* local, const size_t* counts, int root, double* out) {
    if (!c || !counts) return fail(RL_ERR_INVALID, "NULL argument");
    if (root < 0 || root >= c->world) return fail(RL_ERR_INVALID, "no such root rank");
    HIP_TRY(hipSetDevice(c->ctx->device));
    hipStream_t s = c->ctx->stream;
    Rccl* r = rccl();
    const size_t mine = counts[c->rank];
    if (mine && !local) return fail(RL_ERR_INVALID, "local is NULL");
    size_t total = 0;
    for (int k = 0; k < c->world; ++k) total += counts[k];
    const size_t need = (c->rank == root ? total : mine) * sizeof(double);
    if (need > c->staging_bytes) {
        if (c->staging) HIP_TRY(hipFree(c->staging));
        c->staging = nullptr;
        c->staging_bytes = 0;
        HIP_TRY(hipMalloc(&c->staging, need ? need : 1));
        c->staging_bytes = need;
    }
    double* buf = (double*)c->staging;
    if (c->rank == root) {
        if (total && !out) return fail(RL_ERR_INVALID, "out is NULL on the root rank");
        size_t off = 0;
        GroupScope group(r);
        RCCL_TRY(group.start());
        for (int k = 0; k < c->world; ++k) {
            if (k == root) {
                if (mine) HIP_TRY(hipMemcpyAsync(buf + off, local, mine * sizeof(double), hipMemcpyHostToDevice, s));
            } else if (counts[k]) {
                RCCL_TRY(r->Recv(buf + off, counts[k], ncclDouble, k, c->comm, s));
            }
            off += counts[k];
        }
        RCCL_TRY(group.end());
        if (total) HIP_TRY(hipMemcpyAsync(out, buf, total * sizeof(double), hipMemcpyDeviceToHost, s));
    } else if (mine) {
        HIP_TRY(hipMemcpyAsync(buf, local, mine * sizeof(double), hipMemcpyHostToDevice, s));
        GroupScope group(r);
        RCCL_TRY(group.start());
        RCCL_TRY(r->Send(buf, mine, ncclDouble, root, c->comm, s));
        RCCL_TRY(group.end());
    }
    HIP_TRY(hipStreamSynchronize(s));
    return RL_OK;
}

// The sweep's gather: counts[r] elements from rank r's device buffer, rank-major into the root's device buffer -- no padding, no
// host staging, the plans' own arithmetic type (SURVEY 8e sized config 4's gather at 75 MB of fp32).
int rl_comm_gather_device(rl_comm* c, const void* dev_local, const size_t* counts, int dtype, int root, void* dev_out) {
    if (!c || !counts) return fail(RL_ERR_INVALID, "NULL argument");
    if (root < 0 || root >= c->world) return fail(RL_ERR_INVALID, "no such root rank");
    if (dtype != RL_F32 && dtype != RL_F64) return fail(RL_ERR_INVALID, "dtype must be RL_F32 or RL_F64");
    const size_t mine = counts[c->rank], es = dtype == RL_F32 ? 4 : 8;
    if (mine && !dev_local) return fail(RL_ERR_INVALID, "dev_local is NULL");
    const ncclDataType_t nt = dtype == RL_F32 ? ncclFloat : ncclDouble;
    HIP_TRY(hipSetDevice(c->ctx->device));
    hipStream_t s = c->ctx->stream;
    HIP_TRY(hipDeviceSynchronize());   // whatever produced dev_local, on any stream, has finished
    Rccl* r = rccl();
    if (c->rank == root) {
        size_t total = 0;
        for (int k = 0; k < c->world; ++k) total += counts[k];
        if (total && !dev_out) return fail(RL_ERR_INVALID, "dev_out is NULL on the root rank");
        GroupScope group(r);
        RCCL_TRY(group.start());
        size_t off = 0;
        for (int k = 0; k < c->world; ++k) {
            char* dst = (char*)dev_out + off * es;
            if (k == root) {
                if (mine) HIP_TRY(hipMemcpyAsync(dst, dev_local, mine * es, hipMemcpyDeviceToDevice, s));
            } else if (counts[k]) {
                RCCL_TRY(r->Recv(dst, counts[k], nt, k, c->comm, s));
            }
            off += counts[k];
        }
        RCCL_TRY(group.end());
    } else if (mine) {
        GroupScope group(r);
        RCCL_TRY(group.start());
        RCCL_TRY(r->Send(dev_local, mine, nt, root, c->comm, s));
        RCCL_TRY(group.end());
    }
    HIP_TRY(hipStreamSynchronize(s));
    return RL_OK;
}

int rl_comm_bcast_host(rl_comm* c, double* buf, size_t n, int root) {
    if (!c || (n && !buf)) return fail(RL_ERR_INVALID, "NULL argument");
    if (root < 0 || root >= c->world) return fail(RL_ERR_INVALID, "no such root rank");
    if (n == 0) return RL_OK;
    HIP_TRY(hipSetDevice(c->ctx->device));
    hipStream_t s = c->ctx->stream;
    const size_t need = n * sizeof(double);
    if (need > c->staging_bytes) {
        HIP_TRY(hipStreamSynchronize(s));
        if (c->staging) HIP_TRY(hipFree(c->staging));
        c->staging = nullptr;
        c->staging_bytes = 0;
        HIP_TRY(hipMalloc(&c->staging, need));
        c->staging_bytes = need;
    }
    if (c->rank == root) HIP_TRY(hipMemcpyAsync(c->staging, buf, need, hipMemcpyHostToDevice, s));
    RCCL_TRY(rccl()->Broadcast(c->staging, c->staging, n, ncclDouble, root, c->comm, s));
    if (c->rank != root) HIP_TRY(hipMemcpyAsync(buf, c->staging, need, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return RL_OK;
}

int rl_gather_device(rl_comm* c, rl_deconv* plan, int which, int root, const int* counts, void** dev_out,
                     size_t* n_elements, int* dtype_out) {
    RL_TRY(gather_impl(c, plan, which, root, counts));
    int B = 0, V = 0, ny = 0, nx = 0, dtype = RL_F32;
    void* src = nullptr;
    size_t n_elems = 0;
    RL_TRY(rl_deconv_dims(plan, &B, &V, &ny, &nx));
    RL_TRY(rl_deconv_device_ptr(plan, which, &src, &n_elems, &dtype));
    size_t total = 0;
    for (int k = 0; k < c->world; ++k) total += (size_t)counts[k];
    if (dev_out) *dev_out = c->rank == root ? c->gathered : nullptr;
    if (n_elements) *n_elements = c->rank == root ? total * (n_elems / (size_t)B) : 0;
    if (dtype_out) *dtype_out = dtype;
    return RL_OK;
}

int rl_gather(rl_comm* c, rl_deconv* plan, int which, int root, const int* counts, double* host_out) {
    void* dev = nullptr;
    size_t n = 0;
    int dtype = RL_F32;
    RL_TRY(rl_gather_device(c, plan, which, root, counts, &dev, &n, &dtype));
    if (c->rank != root) return RL_OK;
    if (!host_out) return fail(RL_ERR_INVALID, "host_out is NULL on the root rank");
    if (dtype == RL_F64) {
        HIP_TRY(hipMemcpy(host_out, dev, n * 8, hipMemcpyDeviceToHost));
    } else {   // the final gather is small (config 4: 75 MB): widen on the host
        std::vector<float> tmp(n);
        HIP_TRY(hipMemcpy(tmp.data(), dev, n * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; ++i) host_out[i] = (double)tmp[i];
    }
    return RL_OK;
}

}  // extern "C"
