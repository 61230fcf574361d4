// Multi-GPU exchange steps of the hot path behind the C ABI: one RCCL communicator per process (one process per GPU), the
// all-gather of the sharded pool's embeddings (the only data-path collective; it replaces the
// accelerator.gather sites of the reference, e.g. methods/semi_supervised_learning/textual_prompt.py:146-147, 285-286) and
// the mean all-reduce of the prompt gradients (<= 2.1 MB; DDP's gradient all-reduce behind accelerator.backward,
// textual_prompt.py:131).  xGMI is a point-to-point mesh: both messages are single, contiguous buffers (12.8 MB per rank
// for 50 000 x 512 f32 on 8 GPUs), one collective per pass / per step, enqueued on the caller's stream.
//
// RCCL is resolved at run time (dlopen) instead of at link time: the host process usually already carries one RCCL -- PyTorch
// bundles its own librccl.so -- and loading a second copy next to it would double the communicator bootstrap state.  The
// library path is GRIP_RCCL_LIBRARY when set (the Python host passes torch's), else "librccl.so" from the loader path.
#include <dlfcn.h>
#include <stdlib.h>

#include "common.h"

namespace {
struct UniqueId { char internal[128]; };               // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void* Comm;                                     // ncclComm_t
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(Comm*, int, UniqueId, int);
typedef int (*CommDestroyFn)(Comm);
typedef int (*AllGatherFn)(const void*, void*, size_t, int, Comm, hipStream_t);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef const char* (*GetErrorStringFn)(int);
constexpr int kFloat32 = 7, kSum = 0;                   // ncclFloat32, ncclSum

struct Rccl {
    void* lib = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    AllGatherFn all_gather = nullptr;
    AllReduceFn all_reduce = nullptr;
    GetErrorStringFn error_string = nullptr;
};
Rccl g_rccl;

int load_rccl() {
    if (g_rccl.lib) return GRIP_OK;
    const char* path = getenv("GRIP_RCCL_LIBRARY");
    void* h = dlopen(path && *path ? path : "librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { grip_set_error("comm: cannot load RCCL (%s)", dlerror()); return GRIP_ERR_STATE; }
    Rccl r;
    r.lib = h;
    r.get_unique_id = (GetUniqueIdFn)dlsym(h, "ncclGetUniqueId");
    r.comm_init_rank = (CommInitRankFn)dlsym(h, "ncclCommInitRank");
    r.comm_destroy = (CommDestroyFn)dlsym(h, "ncclCommDestroy");
    r.all_gather = (AllGatherFn)dlsym(h, "ncclAllGather");
    r.all_reduce = (AllReduceFn)dlsym(h, "ncclAllReduce");
    r.error_string = (GetErrorStringFn)dlsym(h, "ncclGetErrorString");
    if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_gather || !r.all_reduce) {
        grip_set_error("comm: the RCCL library lacks a required symbol");
        return GRIP_ERR_STATE;
    }
    g_rccl = r;
    return GRIP_OK;
}

int check_nccl(int rc, const char* what) {
    if (rc == 0) return GRIP_OK;
    grip_set_error("comm: %s failed: %s", what, g_rccl.error_string ? g_rccl.error_string(rc) : "RCCL error");
    return GRIP_ERR_HIP;
}
}  // namespace

struct grip_comm {
    Comm comm = nullptr;
    int n_ranks = 1, rank = 0;
};

extern "C" int grip_comm_unique_id(uint8_t* id128) {
    GRIP_REQUIRE(id128, "comm_unique_id: null pointer");
    int rc = load_rccl();
    if (rc) return rc;
    UniqueId id;
    if ((rc = check_nccl(g_rccl.get_unique_id(&id), "ncclGetUniqueId"))) return rc;
    memcpy(id128, id.internal, sizeof(id.internal));
    return GRIP_OK;
}

extern "C" int grip_comm_init_rank(const uint8_t* id128, int n_ranks, int rank, grip_comm** out) {
    GRIP_REQUIRE(id128 && out && n_ranks >= 1 && rank >= 0 && rank < n_ranks, "comm_init_rank: bad arguments (n_ranks=%d rank=%d)", n_ranks, rank);
    int rc = load_rccl();
    if (rc) return rc;
    UniqueId id;
    memcpy(id.internal, id128, sizeof(id.internal));
    grip_comm* c = new grip_comm();
    c->n_ranks = n_ranks;
    c->rank = rank;
    if ((rc = check_nccl(g_rccl.comm_init_rank(&c->comm, n_ranks, id, rank), "ncclCommInitRank"))) { delete c; return rc; }
    *out = c;
    return GRIP_OK;
}

extern "C" int grip_comm_destroy(grip_comm* c) {
    if (!c) return GRIP_OK;
    int rc = c->comm ? check_nccl(g_rccl.comm_destroy(c->comm), "ncclCommDestroy") : GRIP_OK;
    delete c;
    return rc;
}

extern "C" int grip_allgather_embeddings(grip_comm* c, const float* local, float* global, int64_t rows_per_rank, int e, void* stream) {
    GRIP_REQUIRE(c && local && global && rows_per_rank > 0 && e > 0, "allgather_embeddings: bad arguments");
    return check_nccl(g_rccl.all_gather(local, global, (size_t)rows_per_rank * (size_t)e, kFloat32, c->comm, (hipStream_t)stream), "ncclAllGather");
}

// The sum is scaled afterwards by the caller-visible convention of DDP: every rank ends with the MEAN of the ranks' gradients.
__global__ void scale_kernel(float* g, int64_t n, float s) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) g[i] *= s;
}

extern "C" int grip_allreduce_mean(grip_comm* c, float* grads, int64_t n, void* stream) {
    GRIP_REQUIRE(c && grads && n > 0, "allreduce_mean: bad arguments");
    int rc = check_nccl(g_rccl.all_reduce(grads, grads, (size_t)n, kFloat32, kSum, c->comm, (hipStream_t)stream), "ncclAllReduce");
    if (rc) return rc;
    if (c->n_ranks > 1) {
        const int blocks = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
        hipLaunchKernelGGL(scale_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, grads, n, 1.0f / (float)c->n_ranks);
        GRIP_CHECK_HIP(hipGetLastError());
    }
    return GRIP_OK;
}
