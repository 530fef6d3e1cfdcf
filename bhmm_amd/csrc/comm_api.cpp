// comm_api.cpp -- multi-GPU entry points of the C ABI (include/bhmm_amd.h, section 2b): one
// all-reduce of the packed sufficient statistics per EM iteration / Gibbs sweep, the distributed
// form of the sums at bhmm/estimators/maximum_likelihood.py:271-282.  RCCL is loaded on first use
// (dlopen): the library itself does not depend on it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <mutex>
#include <string>

#include "ctx.hpp"

namespace bhmm {
int invalid_arg(const std::string &msg);
}

namespace {

struct rccl_uid { // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128), passed by value
    char internal[BHMM_COMM_ID_BYTES];
};
typedef void *rccl_comm_t;
typedef int (*fn_get_uid)(rccl_uid *);
typedef int (*fn_init_rank)(rccl_comm_t *, int, rccl_uid, int);
typedef int (*fn_allreduce)(const void *, void *, size_t, int, int, rccl_comm_t, hipStream_t);
typedef int (*fn_destroy)(rccl_comm_t);
typedef const char *(*fn_errstr)(int);

struct Rccl {
    void *h = nullptr;
    fn_get_uid get_uid = nullptr;
    fn_init_rank init_rank = nullptr;
    fn_allreduce allreduce = nullptr;
    fn_destroy destroy = nullptr;
    fn_errstr errstr = nullptr;
    std::string why;
};

Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char *nm : names)
            if ((r.h = dlopen(nm, RTLD_NOW | RTLD_LOCAL)))
                break;
        if (!r.h) {
            r.why = std::string("RCCL is not available: ") + dlerror();
            return;
        }
        r.get_uid = reinterpret_cast<fn_get_uid>(dlsym(r.h, "ncclGetUniqueId"));
        r.init_rank = reinterpret_cast<fn_init_rank>(dlsym(r.h, "ncclCommInitRank"));
        r.allreduce = reinterpret_cast<fn_allreduce>(dlsym(r.h, "ncclAllReduce"));
        r.destroy = reinterpret_cast<fn_destroy>(dlsym(r.h, "ncclCommDestroy"));
        r.errstr = reinterpret_cast<fn_errstr>(dlsym(r.h, "ncclGetErrorString"));
        if (!r.get_uid || !r.init_rank || !r.allreduce || !r.destroy)
            r.why = "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy";
    });
    return r;
}

int rccl_fail(int e, const char *what)
{
    Rccl &r = rccl();
    bhmm::set_error(std::string(what) + ": " + (r.errstr ? r.errstr(e) : "RCCL error") + " (" +
                    std::to_string(e) + ")");
    return BHMM_ERR_HIP;
}

} // namespace

struct bhmm_comm {
    rccl_comm_t comm = nullptr;
    int device = 0, nranks = 1, rank = 0;
};

extern "C" {

int bhmm_comm_unique_id(void *id)
{
    if (!id)
        return bhmm::invalid_arg("id == NULL");
    Rccl &r = rccl();
    if (!r.why.empty())
        return bhmm::invalid_arg(r.why);
    rccl_uid u;
    const int e = r.get_uid(&u);
    if (e)
        return rccl_fail(e, "ncclGetUniqueId");
    memcpy(id, u.internal, BHMM_COMM_ID_BYTES);
    return BHMM_OK;
}

int bhmm_comm_init_rank(bhmm_comm **out, int device, int nranks, int rank, const void *id)
{
    if (!out || !id || nranks < 1 || rank < 0 || rank >= nranks)
        return bhmm::invalid_arg("bhmm_comm_init_rank: bad argument");
    Rccl &r = rccl();
    if (!r.why.empty())
        return bhmm::invalid_arg(r.why);
    BHMM_HIP(hipSetDevice(device));
    bhmm_comm *c = new (std::nothrow) bhmm_comm;
    if (!c)
        return BHMM_ERR_NO_MEM;
    c->device = device;
    c->nranks = nranks;
    c->rank = rank;
    rccl_uid u;
    memcpy(u.internal, id, BHMM_COMM_ID_BYTES);
    const int e = r.init_rank(&c->comm, nranks, u, rank);
    if (e) {
        delete c;
        return rccl_fail(e, "ncclCommInitRank");
    }
    *out = c;
    return BHMM_OK;
}

int bhmm_comm_destroy(bhmm_comm *comm)
{
    if (!comm)
        return BHMM_OK;
    if (comm->comm) {
        (void)hipSetDevice(comm->device);
        (void)rccl().destroy(comm->comm);
    }
    delete comm;
    return BHMM_OK;
}

int bhmm_comm_size(const bhmm_comm *comm, int *nranks, int *rank)
{
    if (!comm)
        return bhmm::invalid_arg("comm == NULL");
    if (nranks)
        *nranks = comm->nranks;
    if (rank)
        *rank = comm->rank;
    return BHMM_OK;
}

int bhmm_ctx_allreduce_stats(bhmm_ctx *ctx, bhmm_comm *comm, double *stats_dev, int64_t count)
{
    if (!ctx || !comm || !stats_dev || count < 0)
        return bhmm::invalid_arg("bhmm_ctx_allreduce_stats: bad argument");
    if (comm->device != ctx->device)
        return bhmm::invalid_arg("communicator and context live on different devices");
    if (count == 0)
        return BHMM_OK;
    BHMM_HIP(hipSetDevice(ctx->device));
    // ncclDouble = 8, ncclSum = 0 (rccl.h:448-467); in place, on the context's stream
    const int e = rccl().allreduce(stats_dev, stats_dev, (size_t)count, 8, 0, comm->comm, ctx->stream);
    if (e)
        return rccl_fail(e, "ncclAllReduce");
    return BHMM_OK;
}

} // extern "C"
