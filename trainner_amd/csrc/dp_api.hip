// Data-parallel collectives behind the C ABI (tnr_dp_init / allreduce_bucket / finalize): RCCL over
// xGMI, one communicator per process (= per GPU).  A C / C++ / ctypes consumer can drive the gradient exchange of the SR
// step without PyTorch: rank 0 creates the 128-byte unique id (tnr_dp_unique_id), shares it out of band (file, socket,
// MPI, torch.distributed ...), every rank calls tnr_dp_init, then per optimiser step tnr_dp_allreduce_bucket for each
// gradient bucket on a side HIP stream (ncclAvg: the mean over ranks, in place) and tnr_dp_finalize at exit.
// The reference's counterpart is nn.DataParallel's reduce_add of replica gradients onto GPU 0 (networks.py:252-255).
// librccl is opened lazily (dlopen): the rest of the library has no link-time dependency on it, and a process that already
// loaded an RCCL (PyTorch bundles one) shares that copy.
#include <dlfcn.h>
#include <string.h>

#include "common.h"

namespace {

typedef struct { char internal[128]; } dp_unique_id;          // ncclUniqueId
typedef void *dp_comm_t;                                      // ncclComm_t
typedef int (*fn_get_unique_id)(dp_unique_id *);
typedef int (*fn_comm_init_rank)(dp_comm_t *, int, dp_unique_id, int);
typedef int (*fn_all_reduce)(const void *, void *, size_t, int, int, dp_comm_t, hipStream_t);
typedef int (*fn_broadcast)(const void *, void *, size_t, int, int, dp_comm_t, hipStream_t);
typedef int (*fn_comm_destroy)(dp_comm_t);
typedef int (*fn_comm_count)(dp_comm_t, int *);
typedef const char *(*fn_error_string)(int);

struct Rccl {
    void *h = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_all_reduce all_reduce = nullptr;
    fn_broadcast broadcast = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_comm_count comm_count = nullptr;
    fn_error_string error_string = nullptr;
} g_rccl;

constexpr int NCCL_FLOAT32 = 7, NCCL_SUM = 0, NCCL_AVG = 4;   // ncclDataType_t / ncclRedOp_t values of rccl.h

int load_rccl() {
    if (g_rccl.h) return TNR_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) {
        tnr_set_error("tnr_dp: cannot open librccl (%s)", dlerror());
        return TNR_ELAUNCH;
    }
    g_rccl.get_unique_id = (fn_get_unique_id)dlsym(h, "ncclGetUniqueId");
    g_rccl.comm_init_rank = (fn_comm_init_rank)dlsym(h, "ncclCommInitRank");
    g_rccl.all_reduce = (fn_all_reduce)dlsym(h, "ncclAllReduce");
    g_rccl.broadcast = (fn_broadcast)dlsym(h, "ncclBroadcast");
    g_rccl.comm_destroy = (fn_comm_destroy)dlsym(h, "ncclCommDestroy");
    g_rccl.comm_count = (fn_comm_count)dlsym(h, "ncclCommCount");
    g_rccl.error_string = (fn_error_string)dlsym(h, "ncclGetErrorString");
    if (!g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.all_reduce || !g_rccl.broadcast || !g_rccl.comm_destroy) {
        tnr_set_error("tnr_dp: librccl lacks a required symbol");
        return TNR_ELAUNCH;
    }
    g_rccl.h = h;
    return TNR_OK;
}

int check_rccl(int rc, const char *what) {
    if (rc == 0) return TNR_OK;
    tnr_set_error("%s: RCCL error %d (%s)", what, rc, g_rccl.error_string ? g_rccl.error_string(rc) : "?");
    return TNR_ELAUNCH;
}

}  // namespace

extern "C" int tnr_dp_unique_id(void *id128) {
    TNR_REQUIRE(id128 != nullptr, "dp_unique_id: null pointer");
    const int rc = load_rccl();
    if (rc != TNR_OK) return rc;
    return check_rccl(g_rccl.get_unique_id((dp_unique_id *)id128), "dp_unique_id");
}

extern "C" int tnr_dp_init(const void *id128, int32_t rank, int32_t world, void **comm) {
    TNR_REQUIRE(id128 != nullptr && comm != nullptr && world >= 1 && rank >= 0 && rank < world, "dp_init: bad arguments");
    const int rc = load_rccl();
    if (rc != TNR_OK) return rc;
    dp_unique_id id;
    memcpy(&id, id128, sizeof(id));
    dp_comm_t c = nullptr;
    const int r = check_rccl(g_rccl.comm_init_rank(&c, world, id, rank), "dp_init");
    if (r != TNR_OK) return r;
    *comm = c;
    return TNR_OK;
}

extern "C" int tnr_dp_allreduce_bucket(void *comm, float *buf, int64_t count, int32_t average, void *stream) {
    TNR_REQUIRE(comm != nullptr && buf != nullptr && count >= 0 && g_rccl.h != nullptr, "dp_allreduce_bucket: bad arguments");
    if (count == 0) return TNR_OK;
    return check_rccl(g_rccl.all_reduce(buf, buf, (size_t)count, NCCL_FLOAT32, average ? NCCL_AVG : NCCL_SUM, comm, (hipStream_t)stream),
                      "dp_allreduce_bucket");
}

extern "C" int tnr_dp_broadcast(void *comm, float *buf, int64_t count, int32_t root, void *stream) {
    TNR_REQUIRE(comm != nullptr && buf != nullptr && count >= 0 && g_rccl.h != nullptr, "dp_broadcast: bad arguments");
    if (count == 0) return TNR_OK;
    return check_rccl(g_rccl.broadcast(buf, buf, (size_t)count, NCCL_FLOAT32, root, comm, (hipStream_t)stream), "dp_broadcast");
}

extern "C" int tnr_dp_comm_count(void *comm, int32_t *nranks) {
    TNR_REQUIRE(comm != nullptr && nranks != nullptr && g_rccl.h != nullptr && g_rccl.comm_count != nullptr, "dp_comm_count: bad arguments");
    int n = 0;
    const int rc = check_rccl(g_rccl.comm_count(comm, &n), "dp_comm_count");
    *nranks = n;
    return rc;
}

extern "C" int tnr_dp_finalize(void *comm) {
    TNR_REQUIRE(comm != nullptr && g_rccl.h != nullptr, "dp_finalize: bad arguments");
    return check_rccl(g_rccl.comm_destroy(comm), "dp_finalize");
}
