// pgi_comm.hip -- the pose path's single exchange step: all-gather of the per-edge records.
//
// The reference has no distributed layer at all (its parallelism is the OpenMP loop of
// pose_graph_builder.h:391-413); BASELINE.json:north_star shards the image pairs over the GPUs of a node and
// gathers the per-edge rotations over xGMI before rotation averaging.  One process per GPU; two transports:
//   RCCL : librccl is opened at run time (dlopen; a process that already carries torch's copy reuses it), so
//          libpgi.so has no link-time dependency on it and single-GPU users never load it.  Equal blocks use
//          ncclAllGather; uneven blocks (sum(N)-balanced shards, partial scheduler waves) are exchanged without
//          padding as one group of point-to-point ncclSend/ncclRecv (always closed, also on errors) -- xGMI is a full mesh of point-to-point
//          links, so every rank pushes its block to its 7 peers over 7 different links.
//   host : an all-gather-v callback over host memory (ranks that share one device -- RCCL rejects duplicate
//          devices -- and CPU-side tests); the records make one D2H and one H2D hop.
#include "pgi_internal.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>  // types only; every function is resolved with dlsym

#include <cstring>
#include <vector>

namespace pgi {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    std::string error;
};

static RcclApi* rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* override_path = getenv("PGI_RCCL_LIB");
        const char* names[] = {override_path, "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        // a copy that is already mapped (torch's) wins: two RCCL instances in one process would each own a bootstrap
        for (const char* nm : names)
            if (nm && !api.handle) api.handle = dlopen(nm, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL);
        for (const char* nm : names)
            if (nm && !api.handle) api.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
        if (!api.handle) {
            const char* why = dlerror();
            api.error = std::string("librccl not found: ") + (why ? why : "?");
            return;
        }
        auto sym = [&](const char* n) {
            void* p = dlsym(api.handle, n);
            if (!p && api.error.empty()) api.error = std::string("librccl lacks ") + n;
            return p;
        };
        api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
        api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
        api.Send = (decltype(api.Send))sym("ncclSend");
        api.Recv = (decltype(api.Recv))sym("ncclRecv");
        api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
        api.GetVersion = (decltype(api.GetVersion))dlsym(api.handle, "ncclGetVersion");  // optional
    });
    return &api;
}

#define RCCL_TRY(api, x)                                                                                        \
    do {                                                                                                        \
        ncclResult_t _r = (x);                                                                                  \
        if (_r != ncclSuccess) return fail(PGI_ERR_COMM, std::string(#x) + ": " + (api)->GetErrorString(_r));   \
    } while (0)

// rank-ordered byte offsets of the blocks
static std::vector<uint64_t> block_offsets(const uint64_t* bytes, uint32_t world) {
    std::vector<uint64_t> off((size_t)world + 1, 0);
    for (uint32_t r = 0; r < world; ++r) off[r + 1] = off[r] + bytes[r];
    return off;
}

static int allgatherv_locked(pgi_ctx* ctx, const void* d_local, const uint64_t* h_bytes, void* d_all) {
    const uint32_t world = ctx->comm_world, rank = ctx->comm_rank;
    const std::vector<uint64_t> off = block_offsets(h_bytes, world);
    const uint64_t mine = h_bytes[rank], total = off[world];
    if (total == 0) return PGI_SUCCESS;
    if (!d_all || (mine && !d_local)) return fail(PGI_ERR_INVALID, "allgather: null buffer");
    HIP_TRY(hipSetDevice(ctx->device));
    char* all = (char*)d_all;
    const bool in_place = (const char*)d_local == all + off[rank];
    if (ctx->comm_kind == 0) {  // single process
        if (mine && !in_place) HIP_TRY(hipMemcpyAsync(all, d_local, mine, hipMemcpyDeviceToDevice, ctx->stream));
        return PGI_SUCCESS;
    }
    if (ctx->comm_kind == 1) {
        RcclApi* api = rccl_api();
        ncclComm_t comm = (ncclComm_t)ctx->comm_rccl;
        bool equal = true;
        for (uint32_t r = 1; r < world; ++r) equal &= h_bytes[r] == h_bytes[0];
        if (equal) {
            RCCL_TRY(api, api->AllGather(d_local, all, (size_t)mine, ncclChar, comm, ctx->stream));
            return PGI_SUCCESS;
        }
        if (mine && !in_place) HIP_TRY(hipMemcpyAsync(all + off[rank], d_local, mine, hipMemcpyDeviceToDevice, ctx->stream));
        // A group that was opened is ALWAYS closed: an early return between GroupStart and GroupEnd would leave the group
        // open on this thread and every later RCCL call would only be queued.  The first error is kept, the remaining
        // calls of the group are skipped, GroupEnd runs, then the call fails.
        RCCL_TRY(api, api->GroupStart());
        ncclResult_t first = ncclSuccess;
        const char* what = "";
        for (uint32_t k = 1; k < world && first == ncclSuccess; ++k) {  // peer order rotated by rank: every link is busy from the first step
            const uint32_t to = (rank + k) % world, from = (rank + world - k) % world;
            if (mine) {
                first = api->Send(d_local, (size_t)mine, ncclChar, (int)to, comm, ctx->stream);
                what = "ncclSend";
            }
            if (first == ncclSuccess && h_bytes[from]) {
                first = api->Recv(all + off[from], (size_t)h_bytes[from], ncclChar, (int)from, comm, ctx->stream);
                what = "ncclRecv";
            }
        }
        const ncclResult_t ended = api->GroupEnd();
        if (first != ncclSuccess) return fail(PGI_ERR_COMM, std::string(what) + ": " + api->GetErrorString(first));
        if (ended != ncclSuccess) return fail(PGI_ERR_COMM, std::string("ncclGroupEnd: ") + api->GetErrorString(ended));
        return PGI_SUCCESS;
    }
    // host transport
    std::vector<char> send((size_t)mine), recv((size_t)total);
    if (mine) HIP_TRY(hipMemcpyAsync(send.data(), d_local, mine, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    const int rc = ctx->comm_fn(ctx->comm_user, send.data(), mine, recv.data(), h_bytes, world);
    if (rc != 0) return fail(PGI_ERR_COMM, "allgather: the host transport callback failed (" + std::to_string(rc) + ")");
    HIP_TRY(hipMemcpyAsync(all, recv.data(), total, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // recv is a local buffer
    return PGI_SUCCESS;
}

}  // namespace pgi

using namespace pgi;

extern "C" {

int pgi_comm_unique_id(uint8_t id[PGI_COMM_ID_BYTES]) {
    if (!id) return fail(PGI_ERR_INVALID, "null argument");
    static_assert(sizeof(ncclUniqueId) == PGI_COMM_ID_BYTES, "unique id size");
    RcclApi* api = rccl_api();
    if (!api->error.empty()) return fail(PGI_ERR_COMM, api->error);
    ncclUniqueId u;
    RCCL_TRY(api, api->GetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return PGI_SUCCESS;
}

int pgi_comm_rccl_probe(int* version) {
    RcclApi* api = rccl_api();
    if (version) *version = 0;
    if (!api->error.empty()) return fail(PGI_ERR_COMM, api->error);
    if (version && api->GetVersion) (void)api->GetVersion(version);
    return PGI_SUCCESS;
}

int pgi_comm_destroy(pgi_ctx* ctx) {
    if (!ctx) return fail(PGI_ERR_INVALID, "null ctx");
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (ctx->comm_kind == 1 && ctx->comm_rccl) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)rccl_api()->CommDestroy((ncclComm_t)ctx->comm_rccl);
    }
    ctx->comm_rccl = nullptr;
    ctx->comm_fn = nullptr;
    ctx->comm_user = nullptr;
    ctx->comm_kind = 0;
    ctx->comm_world = 1;
    ctx->comm_rank = 0;
    return PGI_SUCCESS;
}

int pgi_comm_init_rccl(pgi_ctx* ctx, uint32_t world, uint32_t rank, const uint8_t id[PGI_COMM_ID_BYTES]) {
    if (!ctx || !id) return fail(PGI_ERR_INVALID, "null argument");
    if (world == 0 || rank >= world) return fail(PGI_ERR_INVALID, "rank out of range");
    RcclApi* api = rccl_api();
    if (!api->error.empty()) return fail(PGI_ERR_COMM, api->error);
    (void)pgi_comm_destroy(ctx);
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclComm_t comm = nullptr;
    RCCL_TRY(api, api->CommInitRank(&comm, (int)world, u, (int)rank));
    ctx->comm_rccl = comm;
    ctx->comm_kind = 1;
    ctx->comm_world = world;
    ctx->comm_rank = rank;
    return PGI_SUCCESS;
}

int pgi_comm_init_host(pgi_ctx* ctx, uint32_t world, uint32_t rank, pgi_allgatherv_fn fn, void* user) {
    if (!ctx || !fn) return fail(PGI_ERR_INVALID, "null argument");
    if (world == 0 || rank >= world) return fail(PGI_ERR_INVALID, "rank out of range");
    (void)pgi_comm_destroy(ctx);
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->comm_fn = fn;
    ctx->comm_user = user;
    ctx->comm_kind = 2;
    ctx->comm_world = world;
    ctx->comm_rank = rank;
    return PGI_SUCCESS;
}

int pgi_comm_info(pgi_ctx* ctx, uint32_t* world, uint32_t* rank, uint32_t* kind) {
    if (!ctx) return fail(PGI_ERR_INVALID, "null ctx");
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (world) *world = ctx->comm_world;
    if (rank) *rank = ctx->comm_rank;
    if (kind) *kind = (uint32_t)ctx->comm_kind;
    return PGI_SUCCESS;
}

int pgi_allgatherv(pgi_ctx* ctx, const void* d_local, const uint64_t* h_bytes, void* d_all) {
    if (!ctx || !h_bytes) return fail(PGI_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(ctx->mu);
    return allgatherv_locked(ctx, d_local, h_bytes, d_all);
}

int pgi_allgather_edges(pgi_ctx* ctx, const pgi_edge* d_local, const uint32_t* h_counts, pgi_edge* d_all) {
    if (!ctx || !h_counts) return fail(PGI_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(ctx->mu);
    std::vector<uint64_t> bytes(ctx->comm_world);
    for (uint32_t r = 0; r < ctx->comm_world; ++r) bytes[r] = (uint64_t)h_counts[r] * sizeof(pgi_edge);
    return allgatherv_locked(ctx, d_local, bytes.data(), d_all);
}

}  // extern "C"
