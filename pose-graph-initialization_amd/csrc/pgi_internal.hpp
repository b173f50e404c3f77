// pgi_internal.hpp -- shared by the translation units of libpgi.so (not part of the C ABI)
#pragma once
#include <hip/hip_runtime.h>
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
    // scratch for the single-pair drop-in
    void* d_scratch = nullptr;
    size_t scratch_bytes = 0;
    int max_lds = 0;
    unsigned long long* d_prof = nullptr;
    uint32_t* d_bucket = nullptr;  // size-bucket lists of the last ragged batch
    size_t bucket_bytes = 0;
    bool lds_attr_set = false;
    // double-buffered device slots of pgi_estimate_pose_batch_host (own streams: copies overlap kernels)
    struct HostSlot {
        void* d = nullptr;
        size_t bytes = 0;
        uint32_t* d_bucket = nullptr;
        size_t bucket_bytes = 0;
        hipStream_t stream = nullptr;
    } hslot[2];
    void* d_match_ws = nullptr;  // descriptor-matching workspace (views, partial top-2, column best)
    size_t match_ws_bytes = 0;
    uint32_t* d_match_cnt = nullptr;  // per-pair flagged-row counters of the last screened match (forward, then backward)
    uint32_t match_cnt_pairs = 0;
    int match_screen = 1;  // bf16 screening + exact f32 verification when the views carry the data (env PGI_MATCH_SCREEN)
    int match_waves = 4;  // wavefronts per matching workgroup (4 or 8; env PGI_MATCH_WAVES)
    int lds_min_wgs = 2;  // stage rows in LDS only if this many workgroups still fit per CU
};
