// The exchange step of the row-sharded search: RCCL all-gather over xGMI behind the C-ABI.
// The reference is single-process: there is no call to mirror; the shape (one small all-gather of the partial top-k lists
// per stage, merge on every rank) is the one optimized-rag_amd/sharded.py uses through torch.distributed. librccl is opened
// with dlopen at the first call, so single-GPU users carry no dependency on it; a copy already mapped into the process
// (a host that also imported torch) is preferred over a second one.
#include "common.h"

#include <dlfcn.h>

#include <cstring>

namespace {
typedef struct { char internal[128]; } nccl_uid;
typedef void* nccl_comm;
struct rccl_api {
    int (*get_unique_id)(nccl_uid*) = nullptr;
    int (*comm_init_rank)(nccl_comm*, int, nccl_uid, int) = nullptr;
    int (*all_gather)(const void*, void*, size_t, int, nccl_comm, hipStream_t) = nullptr;
    int (*comm_destroy)(nccl_comm) = nullptr;
    int (*comm_count)(nccl_comm, int*) = nullptr;
    const char* (*error_string)(int) = nullptr;
    bool ok = false;
};

rccl_api& rccl() {
    static rccl_api api = [] {
        rccl_api a;
        void* lib = nullptr;
        for (const char* name : {"librccl.so", "librccl.so.1"})
            if (!lib) lib = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
            if (!lib) lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (!lib) return a;
        a.get_unique_id = reinterpret_cast<decltype(a.get_unique_id)>(dlsym(lib, "ncclGetUniqueId"));
        a.comm_init_rank = reinterpret_cast<decltype(a.comm_init_rank)>(dlsym(lib, "ncclCommInitRank"));
        a.all_gather = reinterpret_cast<decltype(a.all_gather)>(dlsym(lib, "ncclAllGather"));
        a.comm_destroy = reinterpret_cast<decltype(a.comm_destroy)>(dlsym(lib, "ncclCommDestroy"));
        a.comm_count = reinterpret_cast<decltype(a.comm_count)>(dlsym(lib, "ncclCommCount"));
        a.error_string = reinterpret_cast<decltype(a.error_string)>(dlsym(lib, "ncclGetErrorString"));
        a.ok = a.get_unique_id && a.comm_init_rank && a.all_gather && a.comm_destroy;
        return a;
    }();
    return api;
}

int rccl_fail(rag_ctx* h, const char* what, int rc) {
    h->err = std::string(what) + ": " + (rccl().error_string ? rccl().error_string(rc) : "RCCL error");
    return RAG_ERR_HIP;
}
}  // namespace

void comm_free(rag_ctx* h) {
    if (h->comm && rccl().ok) rccl().comm_destroy(h->comm);
    h->comm = nullptr;
    h->comm_rank = 0;
    h->comm_world = 1;
}

extern "C" {

int rag_comm_unique_id(void* id128_out) {
    if (!id128_out || !rccl().ok) return RAG_ERR_ARG;
    nccl_uid id;
    if (rccl().get_unique_id(&id) != 0) return RAG_ERR_HIP;
    std::memcpy(id128_out, id.internal, sizeof(id.internal));
    return RAG_OK;
}

int rag_comm_init(rag_handle_t h, int rank, int world, const void* id128) {
    if (!h) return RAG_ERR_ARG;
    std::lock_guard<std::mutex> lock_(h->mu);
    ARG_CHECK(h, id128 && world >= 1 && rank >= 0 && rank < world, "comm_init: bad rank / world / id");
    ARG_CHECK(h, rccl().ok, "comm_init: librccl.so could not be opened");
    HIP_TRY(h, hipSetDevice(h->device));
    comm_free(h);
    nccl_uid id;
    std::memcpy(id.internal, id128, sizeof(id.internal));
    nccl_comm c = nullptr;
    const int rc = rccl().comm_init_rank(&c, world, id, rank);
    if (rc != 0) return rccl_fail(h, "ncclCommInitRank", rc);
    h->comm = c;
    h->comm_rank = rank;
    h->comm_world = world;
    return RAG_OK;
}

int rag_comm_allgather_dev(rag_handle_t h, const void* send_dev, void* recv_dev, size_t bytes, void* stream) {
    if (!h) return RAG_ERR_ARG;
    std::lock_guard<std::mutex> lock_(h->mu);
    ARG_CHECK(h, h->comm != nullptr, "comm_allgather: rag_comm_init has not run");
    ARG_CHECK(h, send_dev && recv_dev && bytes > 0, "comm_allgather: bad arguments");
    HIP_TRY(h, hipSetDevice(h->device));
    const int rc = rccl().all_gather(send_dev, recv_dev, bytes, 0 /* ncclInt8 */, h->comm, (hipStream_t)stream);
    if (rc != 0) return rccl_fail(h, "ncclAllGather", rc);
    return RAG_OK;
}

int rag_comm_count(rag_handle_t h, int* count_out) {
    if (!h) return RAG_ERR_ARG;
    std::lock_guard<std::mutex> lock_(h->mu);
    ARG_CHECK(h, count_out != nullptr, "comm_count: null");
    ARG_CHECK(h, h->comm != nullptr, "comm_count: rag_comm_init has not run");
    ARG_CHECK(h, rccl().comm_count != nullptr, "comm_count: ncclCommCount not found in librccl");
    const int rc = rccl().comm_count(h->comm, count_out);
    if (rc != 0) return rccl_fail(h, "ncclCommCount", rc);
    return RAG_OK;
}

int rag_comm_destroy(rag_handle_t h) {
    if (!h) return RAG_ERR_ARG;
    std::lock_guard<std::mutex> lock_(h->mu);
    comm_free(h);
    return RAG_OK;
}

}  // extern "C"
