// pgi_internal.hpp -- shared by the translation units of libpgi.so (not part of the C ABI)
#pragma once
#include <hip/hip_runtime.h>
#include <condition_variable>
#include <mutex>
#include <string>
#include "../../include/pgi.h"

namespace pgi {
std::string& last_error_ref();
inline int fail(int code, const std::string& msg) {
    last_error_ref() = msg;
    return code;
}
}  // namespace pgi
#define HIP_TRY(x)                                                                                   \
    do {                                                                                             \
        hipError_t _e = (x);                                                                         \
        if (_e != hipSuccess)                                                                        \
            return pgi::fail(PGI_ERR_DEVICE, std::string(#x) + ": " + hipGetErrorString(_e));        \
    } while (0)

struct pgi_ctx {
    int device;
    hipStream_t stream;
    pgi_params prm;
    std::mutex mu;
    int max_lds = 0;
    int n_cus = 256;
    unsigned long long* d_prof = nullptr;
    uint32_t* d_bucket = nullptr;  // size-bucket lists of the last ragged batch
    size_t bucket_bytes = 0;
    // pool of private slots for the re-entrant single-pair drop-in (pgi_estimate_pose): the reference's seam is
    // called from kCoreNumber OpenMP threads (pose_graph_builder.h:391-392), so concurrent callers must overlap
    struct PairSlot {
        hipStream_t stream = nullptr;
        void* d = nullptr;       // device scratch
        void* h = nullptr;       // pinned staging (hipHostMalloc): true async DMA in both directions
        size_t bytes = 0;
        bool busy = false;
    } pslot[PGI_PAIR_SLOTS];
    std::mutex slot_mu;
    std::condition_variable slot_cv;
    // multi-GPU exchange (pgi_comm.hip)
    uint32_t comm_world = 1, comm_rank = 0;
    int comm_kind = 0;             // 0 none, 1 RCCL, 2 host callback
    void* comm_rccl = nullptr;     // ncclComm_t
    pgi_allgatherv_fn comm_fn = nullptr;
    void* comm_user = nullptr;
    // double-buffered device slots of pgi_estimate_pose_batch_host (own streams: copies overlap kernels)
    struct HostSlot {
        void* d = nullptr;
        size_t bytes = 0;
        uint32_t* d_bucket = nullptr;
        size_t bucket_bytes = 0;
        void* h_small = nullptr;  // page-locked staging of the chunk's per-pair arrays
        size_t h_small_bytes = 0;
        void* h_io = nullptr;     // page-locked staging of the chunk's rows (in) and results (out) when the caller's buffers are pageable
        size_t h_io_bytes = 0;
        long pending = -1;        // chunk whose results wait in h_io for the copy into the caller's buffers
        hipStream_t stream = nullptr;
        hipEvent_t in_done = nullptr, k_done = nullptr, out_done = nullptr;
        bool used = false;
    } hslot[4];
    hipStream_t copy_in = nullptr, copy_out = nullptr;  // high-priority input stream of the host-pointer path (copy_out: unused)
    // page-locked buffers worked on in place (estimate_host_direct): per-pair arrays, and the mirror of rows outside LDS
    void* d_direct = nullptr;
    size_t direct_bytes = 0;
    void* d_mirror = nullptr;
    size_t mirror_bytes = 0;
    int host_direct = 1;  // env PGI_HOST_DIRECT=0: always copy page-locked buffers through HBM
    int resident_wgs = 768;  // workgroups of K1 the device keeps resident (CUs x 3): the chunk quantum of the host path
    void* d_match_ws = nullptr;  // descriptor-matching workspace (views, partial top-2, column best)
    size_t match_ws_bytes = 0;
    size_t guided_arena_cap = ~(size_t)0;  // bytes the guided scan's list arena may take (lowered when the device runs short; pgi_match.hip)
    uint32_t* d_match_cnt = nullptr;  // per-pair flagged-row counters of the last screened match (forward, then backward)
    uint32_t match_cnt_pairs = 0;
    void* h_match_stage[2] = {nullptr, nullptr};  // page-locked mirrors of the screened matcher's pair tables, used in turn
    size_t match_stage_bytes[2] = {0, 0};
    hipEvent_t match_stage_ev[2] = {nullptr, nullptr};  // their last uploads
    int match_stage_next = 0;
    int match_screen = 1;  // bf16 screening + exact f32 verification when the views carry the data (env PGI_MATCH_SCREEN)
    int match_waves = 4;  // wavefronts per matching workgroup (4 or 8; env PGI_MATCH_WAVES)
    int lds_min_wgs = 2;  // stage rows in LDS only if this many workgroups still fit per CU
    int hybrid_rows = 1;  // pairs just above the 4-workgroup LDS capacity keep their tail rows in HBM/L2 (env PGI_HYBRID_ROWS)
    // ragged batches: the occupancy classes of one batch are independent launches; they run CONCURRENTLY on side streams
    // (fork / join by events around the caller's stream), so one class's workgroups fill the tail of another's
    hipStream_t class_stream[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t class_fork = nullptr, class_join[4] = {nullptr, nullptr, nullptr, nullptr};
    int class_overlap = 1;  // env PGI_CLASS_OVERLAP=0: one class after the other on the caller's stream
    int k1_nw = 0;          // wavefronts per pair of K1: 0 = by batch size (launch_estimate), 1 / 2 / 4 forced (env PGI_K1_NW)
    int k1_persistent = 1;  // size-class launches as persistent grids (resident workgroups pull pairs); env PGI_K1_PERSISTENT=0: grid = pairs
};
