// pgi_tracklets.hip -- multi-view tracklets resident in HBM (reference: point_track.h:541-711).
//
// What the reference does, one inlier match (P_s, P_d) at a time: every track holding P_s that does not hold P_d
// gains P_d, every track that held P_d before the match and does not hold P_s gains P_s, and if nothing was extended a
// new two-member track is started (tracks are never merged, so a point sits in several).  getCorrespondences walks the
// destination view's track list in append order.  Observable state is therefore ONE append-only event log
//     event = (track, view, keypoint)       "track gained the point; the view's list gained the track"
// in the sequential order of the reference: a track's members, a view's track list and a point's track list are the
// events with that track / view / point, in log order.  The store keeps the log and three stable re-orderings of it
// (by track, by view, by point), rebuilt with radix sorts after every batch of add() calls.
//
// add() in parallel: a match reads and extends only the track LISTS OF ITS TWO POINTS ("does track T hold P" is
// "is T in P's list": both are appended together), so two matches are independent unless they share a point.  A batch
// (all pairs of a scheduler wave, in commit order) is flattened to matches k = 0..M-1; for every point the matches
// touching it are ranked in k order (one stable sort); a match runs when both its points have seen all lower-ranked
// matches (per-point cursors, release/acquire), one lane per match, rounds of launches until all are done.  Each run
// appends events tagged (k, j); sorting the batch's events by tag restores the sequential order exactly, and numbering
// the new tracks by k restores the reference's track indices.  Matches of one pair are independent unless keypoints
// repeat, so the number of rounds is the longest chain of pairs handing the same physical point on -- tens, not M.
//
// The reference's quirks are kept (oracle/tracklets_oracle.py): the first point ever registered has id 0 == "unseen",
// so its second visit forgets its first track (here: that point's list skips its first entry once it has been visited
// again, while track 0 still counts as holding it); getCorrespondences returns up to max + 1 entries; a track listing the
// source view twice before the destination yields destination index 0.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include "pgi_internal.hpp"

namespace {

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr int kSubBits = 12;  // events per match < 4096
constexpr int NT = 256;

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    // grow-only; `keep` bytes of the old contents survive
    hipError_t reserve(size_t want, hipStream_t s, size_t keep = 0) {
        if (want <= bytes) return hipSuccess;
        size_t nb = std::max(want, bytes + bytes / 2);
        nb = (nb + 255) & ~(size_t)255;
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, nb);
        if (e != hipSuccess) return e;
        if (keep && p) {
            e = hipMemcpyAsync(q, p, keep, hipMemcpyDeviceToDevice, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) { (void)hipFree(q); return e; }
        } else if (p) {
            e = hipStreamSynchronize(s);  // kernels in flight may still use the old block
            if (e != hipSuccess) { (void)hipFree(q); return e; }
        }
        if (p) (void)hipFree(p);
        p = q;
        bytes = nb;
        return hipSuccess;
    }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

struct DeviceState {       // lives in device memory, mirrored to the host between phases
    uint64_t qkey;         // the first point ever registered
    uint32_t qphase;       // 0: nothing registered yet, 1: that point has been visited once, 2: visited again
    uint32_t n_matches;    // M of the running batch
    uint32_t n_new;        // events written by the running batch
    uint32_t n_done;       // matches finished
    uint32_t overflow;     // bit 0: event buffer full, bit 1: more than 4095 events in one match
    uint32_t n_new_tracks;
};

}  // namespace

struct pgi_tracklets {
    pgi_ctx* ctx = nullptr;
    uint32_t n_views = 0;
    uint32_t n_tracks = 0;
    size_t n_events = 0;
    // canonical log
    DevBuf ev_track, ev_key;
    // re-orderings
    DevBuf mem_key, trk_begin;      // members by track; offsets n_tracks + 1
    DevBuf pt_key, pt_track;        // sorted point keys; tracks by point
    DevBuf view_track, view_begin;  // tracks by view; offsets n_views + 1
    DevBuf state;                   // DeviceState
    // scratch
    DevBuf sort_tmp, a64, b64, a32, b32, c32, d32;
    DevBuf pairs, pair_off;
    DevBuf keyS, keyD, segS, segD, rankS, rankD, done, newflag, newrank;
    DevBuf seg_lo, seg_hi, seg_head, seg_last, seg_cnt, cursor;
    DevBuf nev_track, nev_key, nev_seq, nev_next;
    size_t nev_cap = 0;
    uint32_t last_rounds = 0;
};

namespace {

struct PairDesc {
    uint32_t view_src, view_dst, n_max, pad;
    const uint32_t* d_src;
    const uint32_t* d_dst;
    const uint8_t* d_mask;
    const uint32_t* d_count;
};

__device__ inline uint32_t load_acquire(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void store_release(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }

// block-wide exclusive prefix of a flag, in thread order; returns the block total through `total`
__device__ inline uint32_t block_rank(bool flag, uint32_t* wave_tot /* NT/64 + 1 words of LDS */, uint32_t& total) {
    const uint64_t m = __ballot(flag);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t in_wave = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();  // previous use of wave_tot is over
    if (lane == 0) wave_tot[w] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t before = 0, all = 0;
    for (int i = 0; i < NT / 64; ++i) {
        const uint32_t t = wave_tot[i];
        if (i < w) before += t;
        all += t;
    }
    total = all;
    return before + in_wave;
}

__device__ inline bool pair_row_valid(const PairDesc& d, uint32_t n, uint32_t i) { return i < n && (!d.d_mask || d.d_mask[i] != 0); }

// (1) masked-in matches per pair
__global__ void __launch_bounds__(NT) trk_count_kernel(const PairDesc* pairs, uint32_t* valid) {
    __shared__ uint32_t wt[NT / 64 + 1];
    const PairDesc d = pairs[blockIdx.x];
    const uint32_t n = d.d_count ? min(*d.d_count, d.n_max) : d.n_max;
    uint32_t mine = 0;
    for (uint32_t i = threadIdx.x; i < n; i += NT) mine += pair_row_valid(d, n, i) ? 1u : 0u;
    for (int o = 32; o; o >>= 1) mine += __shfl_down(mine, o);
    if ((threadIdx.x & 63) == 0) wt[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t s = 0;
        for (int i = 0; i < NT / 64; ++i) s += wt[i];
        valid[blockIdx.x] = s;
    }
}

// (2) exclusive scan over the pairs of the batch (one block; batches hold hundreds to thousands of pairs)
__global__ void __launch_bounds__(NT) trk_pair_scan_kernel(const uint32_t* valid, uint32_t* off, uint32_t n_pairs, DeviceState* st) {
    __shared__ uint32_t wt[NT / 64 + 1];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t b = 0; b < n_pairs; b += NT) {
        const uint32_t i = b + threadIdx.x;
        const uint32_t v = i < n_pairs ? valid[i] : 0u;
        // inclusive scan inside the wave, then across waves
        uint32_t x = v;
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(x, o);
            if (lane >= o) x += y;
        }
        if (lane == 63) wt[w] = x;
        __syncthreads();
        uint32_t before = carry;
        for (int j = 0; j < w; ++j) before += wt[j];
        if (i < n_pairs) off[i] = before + x - v;
        __syncthreads();
        if (threadIdx.x == NT - 1) carry = before + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        off[n_pairs] = carry;
        st->n_matches = carry;
        st->n_new = 0;
        st->n_done = 0;
        st->overflow = 0;
        st->n_new_tracks = 0;
    }
}

// (3) flatten: match k = off[pair] + (rank among the pair's masked-in rows); the two point keys go to the sort input
// at 2k (source) and 2k + 1 (destination) so that a stable sort by key keeps k order inside every point
__global__ void __launch_bounds__(NT) trk_flatten_kernel(const PairDesc* pairs, const uint32_t* off, uint64_t* keyS, uint64_t* keyD,
                                                         uint64_t* ent_key) {
    __shared__ uint32_t wt[NT / 64 + 1];
    const PairDesc d = pairs[blockIdx.x];
    const uint32_t n = d.d_count ? min(*d.d_count, d.n_max) : d.n_max;
    uint32_t base = off[blockIdx.x];
    for (uint32_t b = 0; b < n; b += NT) {
        const uint32_t i = b + threadIdx.x;
        const bool ok = pair_row_valid(d, n, i);
        uint32_t total;
        const uint32_t r = block_rank(ok, wt, total);
        if (ok) {
            const uint32_t k = base + r;
            const uint64_t ks = ((uint64_t)d.view_src << 32) | d.d_src[i], kd = ((uint64_t)d.view_dst << 32) | d.d_dst[i];
            keyS[k] = ks;
            keyD[k] = kd;
            ent_key[2 * (size_t)k] = ks;
            ent_key[2 * (size_t)k + 1] = kd;
        }
        base += total;
    }
}

__device__ inline uint32_t lower_bound64(const uint64_t* a, uint32_t n, uint64_t key) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__device__ inline uint32_t lower_bound32(const uint32_t* a, uint32_t n, uint32_t key) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// (4a) segment heads of the sorted entries
__global__ void trk_heads_kernel(const uint64_t* key, uint32_t n, uint32_t* head) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) head[i] = (i == 0 || key[i] != key[i - 1]) ? 1u : 0u;
}
// (4b) per segment: where it starts, and where the point's committed track list lies in pt_track
__global__ void trk_segment_kernel(const uint64_t* key, const uint32_t* seg_incl, uint32_t n, const uint64_t* pt_key, uint32_t n_old,
                                   uint32_t* seg_start, uint32_t* seg_lo, uint32_t* seg_hi) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (i != 0 && key[i] == key[i - 1]) return;
    const uint32_t s = seg_incl[i] - 1;
    seg_start[s] = i;
    const uint32_t lo = lower_bound64(pt_key, n_old, key[i]);
    uint32_t hi = lo;  // lists are short: walk to the end of the run
    while (hi < n_old && pt_key[hi] == key[i]) ++hi;
    seg_lo[s] = lo;
    seg_hi[s] = hi;
}
// (4c) every entry learns its segment and its rank inside it
__global__ void trk_rank_kernel(const uint32_t* val, const uint32_t* seg_incl, const uint32_t* seg_start, uint32_t n, uint32_t* segS,
                                uint32_t* segD, uint32_t* rankS, uint32_t* rankD) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = seg_incl[i] - 1, r = i - seg_start[s], e = val[i], k = e >> 1;
    if (e & 1u) { segD[k] = s; rankD[k] = r; } else { segS[k] = s; rankS[k] = r; }
}

struct RoundArgs {
    DeviceState* st;
    const uint64_t *keyS, *keyD;
    const uint32_t *segS, *segD, *rankS, *rankD;
    uint8_t* done;
    uint32_t* newflag;
    const uint32_t *seg_lo, *seg_hi;
    uint32_t *seg_head, *seg_last, *seg_cnt, *cursor;
    const uint32_t* pt_track;  // committed lists
    uint32_t *nev_track, *nev_next;
    uint64_t *nev_key, *nev_seq;
    uint32_t nev_cap, n_matches, track_base;
    int first_ever;  // the store was empty when the batch started
};

// a point's track list: committed part [lo, hi) of pt_track, then `cnt` cells of this batch chained from `head`;
// `skip` drops the very first entry (the forgotten first track of the first point ever registered)
struct PointList {
    uint32_t lo, hi, head, cnt;
    bool skip;
};

template <class F>
__device__ inline void for_each_track(const RoundArgs& a, const PointList& l, F f) {
    bool skip = l.skip;
    for (uint32_t i = l.lo; i < l.hi; ++i) {
        if (skip) { skip = false; continue; }
        f(a.pt_track[i]);
    }
    uint32_t c = l.head;
    for (uint32_t i = 0; i < l.cnt; ++i, c = a.nev_next[c]) {
        if (skip) { skip = false; continue; }
        f(a.nev_track[c]);
    }
}
__device__ inline bool list_holds(const RoundArgs& a, const PointList& l, uint32_t track) {
    bool found = false;
    for_each_track(a, l, [&](uint32_t t) { found |= (t == track); });
    return found;
}

// (5) one lane per match; runs it if both its points are ready for it
__global__ void __launch_bounds__(NT) trk_round_kernel(RoundArgs a) {
    const uint32_t k = blockIdx.x * NT + threadIdx.x;
    if (k >= a.n_matches || a.done[k]) return;
    const uint32_t sS = a.segS[k], sD = a.segD[k], rS = a.rankS[k], rD = a.rankD[k];
    for (int attempt = 0; attempt < 4; ++attempt) {
        if (load_acquire(&a.cursor[sS]) != rS || load_acquire(&a.cursor[sD]) != rD) continue;
        const uint64_t kS = a.keyS[k], kD = a.keyD[k];
        // the id-0 quirk (point_track.h:651-657)
        uint32_t phase = a.st->qphase;
        uint64_t qkey = a.st->qkey;
        if (a.first_ever && k == 0) {
            a.st->qkey = qkey = kS;
            a.st->qphase = phase = 1;
        } else if (phase == 1 && (kS == qkey || kD == qkey)) {
            a.st->qphase = phase = 2;
        }
        const bool qS = phase == 2 && kS == qkey, qD = phase == 2 && kD == qkey;
        const PointList LS{a.seg_lo[sS], a.seg_hi[sS], a.seg_head[sS], a.seg_cnt[sS], qS};
        const PointList LD{a.seg_lo[sD], a.seg_hi[sD], a.seg_head[sD], a.seg_cnt[sD], qD};
        uint32_t sub = 0;
        auto emit = [&](uint32_t track, uint64_t key, uint32_t seg) {
            const uint32_t e = atomicAdd(&a.st->n_new, 1u);
            if (e >= a.nev_cap) { atomicOr(&a.st->overflow, 1u); return; }
            if (sub >= (1u << kSubBits)) { atomicOr(&a.st->overflow, 2u); return; }
            a.nev_track[e] = track;
            a.nev_key[e] = key;
            a.nev_seq[e] = ((uint64_t)k << kSubBits) | sub++;
            a.nev_next[e] = kNone;
            const uint32_t c = a.seg_cnt[seg];
            if (c == 0) a.seg_head[seg] = e; else a.nev_next[a.seg_last[seg]] = e;
            a.seg_last[seg] = e;
            a.seg_cnt[seg] = c + 1;
        };
        // every track of the source point learns the destination point (point_track.h:664-681) ...
        for_each_track(a, LS, [&](uint32_t t) {
            if (list_holds(a, LD, t) || (qD && t == 0)) return;
            emit(t, kD, sD);
        });
        // ... and every earlier track of the destination point learns the source point (:683-699)
        for_each_track(a, LD, [&](uint32_t t) {
            if (list_holds(a, LS, t) || (qS && t == 0)) return;
            emit(t, kS, sS);
        });
        if (sub == 0) {  // nothing extended: the match starts a track (:701-709)
            const uint32_t t = a.track_base + k;
            emit(t, kS, sS);
            emit(t, kD, sD);
            a.newflag[k] = 1;
        }
        __threadfence();
        store_release(&a.cursor[sS], rS + 1);
        store_release(&a.cursor[sD], rD + 1);
        a.done[k] = 1;
        const uint64_t act = __ballot(1);
        if ((threadIdx.x & 63) == (uint32_t)__ffsll((long long)act) - 1u) atomicAdd(&a.st->n_done, (uint32_t)__popcll(act));
        return;
    }
}

// (6) after the rounds: events in tag order, provisional track numbers replaced, appended to the log
__global__ void trk_append_kernel(const uint32_t* order, const uint32_t* nev_track, const uint64_t* nev_key, const uint32_t* newrank,
                                  uint32_t n_new, uint32_t track_base, uint32_t* ev_track, uint64_t* ev_key) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_new) return;
    const uint32_t e = order[i];
    uint32_t t = nev_track[e];
    if (t >= track_base) t = track_base + newrank[t - track_base];
    ev_track[i] = t;
    ev_key[i] = nev_key[e];
}

__global__ void trk_gather64_kernel(const uint32_t* idx, const uint64_t* src, uint32_t n, uint64_t* dst) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}
__global__ void trk_views_of_kernel(const uint64_t* key, uint32_t n, uint32_t* view) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) view[i] = (uint32_t)(key[i] >> 32);
}
// offsets of the runs of a sorted id array: begin[id] = first position holding a value >= id
__global__ void trk_offsets_kernel(const uint32_t* sorted, uint32_t n, uint32_t n_ids, uint32_t* begin) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id <= n_ids) begin[id] = id == n_ids ? n : lower_bound32(sorted, n, id);
}

// (7) getCorrespondences (point_track.h:568-631): one workgroup per query walks the destination view's track list
__global__ void __launch_bounds__(NT) trk_get_kernel(const uint32_t* q_src, const uint32_t* q_dst, uint32_t n_views, const uint32_t* view_begin,
                                                     const uint32_t* view_track, const uint32_t* trk_begin, const uint64_t* mem_key,
                                                     uint32_t limit, uint32_t out_stride, uint32_t* out_src, uint32_t* out_dst,
                                                     uint32_t* out_cnt) {
    __shared__ uint32_t wt[NT / 64 + 1];
    const uint32_t q = blockIdx.x, vs = q_src[q], vd = q_dst[q];
    uint32_t base = 0;
    if (vs < n_views && vd < n_views) {
        const uint32_t b = view_begin[vd], e = view_begin[vd + 1];
        for (uint32_t c = b; c < e && base < limit; c += NT) {
            const uint32_t i = c + threadIdx.x;
            bool touches = false;
            uint32_t ps = 0, pd = 0;
            if (i < e) {
                const uint32_t t = view_track[i];
                int hits = 0;
                for (uint32_t m = trk_begin[t], me = trk_begin[t + 1]; m < me; ++m) {
                    const uint64_t key = mem_key[m];
                    const uint32_t v = (uint32_t)(key >> 32), p = (uint32_t)key;
                    if (v == vs) touches = true;
                    if (hits < 2) {
                        if (v == vs) { ps = p; ++hits; }
                        else if (v == vd) { pd = p; ++hits; }
                    }
                    if (hits >= 2 && touches) break;
                }
            }
            uint32_t total;
            const uint32_t r = base + block_rank(touches, wt, total);
            if (touches && r < limit) {
                out_src[(size_t)q * out_stride + r] = ps;
                out_dst[(size_t)q * out_stride + r] = pd;
            }
            base += total;
        }
    }
    if (threadIdx.x == 0) out_cnt[q] = min(base, limit);
}

int bits_for(uint64_t n) {  // bits needed to hold values 0..n-1
    int b = 1;
    while (b < 64 && (1ull << b) < n) ++b;
    return b;
}

#define TRK_TRY(x)                                                                                        \
    do {                                                                                                  \
        hipError_t _e = (x);                                                                              \
        if (_e != hipSuccess) return pgi::fail(_e == hipErrorOutOfMemory ? PGI_ERR_NOMEM : PGI_ERR_DEVICE, \
                                               std::string(#x) + ": " + hipGetErrorString(_e));           \
    } while (0)

inline dim3 grid_for(size_t n, int block = NT) { return dim3((unsigned)((n + block - 1) / block)); }

// stable sort of (key, value) pairs by the low `bits` of the key
template <class K>
int sort_pairs(pgi_tracklets* t, const K* kin, K* kout, const uint32_t* vin, uint32_t* vout, size_t n, int bits, hipStream_t s) {
    size_t tmp = 0;
    TRK_TRY(rocprim::radix_sort_pairs(nullptr, tmp, kin, kout, vin, vout, n, 0u, (unsigned)bits, s));
    TRK_TRY(t->sort_tmp.reserve(tmp, s));
    TRK_TRY(rocprim::radix_sort_pairs(t->sort_tmp.p, tmp, kin, kout, vin, vout, n, 0u, (unsigned)bits, s));
    return PGI_SUCCESS;
}
template <class K>
int sort_pairs_iota(pgi_tracklets* t, const K* kin, K* kout, uint32_t* vout, size_t n, int bits, hipStream_t s) {
    rocprim::counting_iterator<uint32_t> iota(0u);
    size_t tmp = 0;
    TRK_TRY(rocprim::radix_sort_pairs(nullptr, tmp, kin, kout, iota, vout, n, 0u, (unsigned)bits, s));
    TRK_TRY(t->sort_tmp.reserve(tmp, s));
    TRK_TRY(rocprim::radix_sort_pairs(t->sort_tmp.p, tmp, kin, kout, iota, vout, n, 0u, (unsigned)bits, s));
    return PGI_SUCCESS;
}
int scan_u32(pgi_tracklets* t, const uint32_t* in, uint32_t* out, size_t n, bool inclusive, hipStream_t s) {
    size_t tmp = 0;
    if (inclusive) {
        TRK_TRY(rocprim::inclusive_scan(nullptr, tmp, in, out, n, rocprim::plus<uint32_t>(), s));
        TRK_TRY(t->sort_tmp.reserve(tmp, s));
        TRK_TRY(rocprim::inclusive_scan(t->sort_tmp.p, tmp, in, out, n, rocprim::plus<uint32_t>(), s));
    } else {
        TRK_TRY(rocprim::exclusive_scan(nullptr, tmp, in, out, 0u, n, rocprim::plus<uint32_t>(), s));
        TRK_TRY(t->sort_tmp.reserve(tmp, s));
        TRK_TRY(rocprim::exclusive_scan(t->sort_tmp.p, tmp, in, out, 0u, n, rocprim::plus<uint32_t>(), s));
    }
    return PGI_SUCCESS;
}

int read_state(pgi_tracklets* t, DeviceState& h, hipStream_t s) {
    TRK_TRY(hipMemcpyAsync(&h, t->state.p, sizeof h, hipMemcpyDeviceToHost, s));
    TRK_TRY(hipStreamSynchronize(s));
    return PGI_SUCCESS;
}

// the three re-orderings of the whole log
int rebuild_views(pgi_tracklets* t, hipStream_t s) {
    const size_t N = t->n_events;
    if (!N) return PGI_SUCCESS;
    TRK_TRY(t->a32.reserve(N * 4, s));
    TRK_TRY(t->b32.reserve(N * 4, s));
    TRK_TRY(t->c32.reserve(N * 4, s));
    TRK_TRY(t->a64.reserve(N * 8, s));
    TRK_TRY(t->mem_key.reserve(N * 8, s));
    TRK_TRY(t->pt_key.reserve(N * 8, s));
    TRK_TRY(t->pt_track.reserve(N * 4, s));
    TRK_TRY(t->view_track.reserve(N * 4, s));
    TRK_TRY(t->trk_begin.reserve(((size_t)t->n_tracks + 2) * 4, s));
    TRK_TRY(t->view_begin.reserve(((size_t)t->n_views + 2) * 4, s));
    int rc;
    // by track: members in log order
    if ((rc = sort_pairs_iota<uint32_t>(t, t->ev_track.as<uint32_t>(), t->a32.as<uint32_t>(), t->b32.as<uint32_t>(), N, bits_for(t->n_tracks), s))) return rc;
    hipLaunchKernelGGL(trk_gather64_kernel, grid_for(N), dim3(NT), 0, s, t->b32.as<uint32_t>(), t->ev_key.as<uint64_t>(), (uint32_t)N, t->mem_key.as<uint64_t>());
    hipLaunchKernelGGL(trk_offsets_kernel, grid_for((size_t)t->n_tracks + 1), dim3(NT), 0, s, t->a32.as<uint32_t>(), (uint32_t)N, t->n_tracks, t->trk_begin.as<uint32_t>());
    // by point: tracks in log order
    if ((rc = sort_pairs<uint64_t>(t, t->ev_key.as<uint64_t>(), t->pt_key.as<uint64_t>(), t->ev_track.as<uint32_t>(), t->pt_track.as<uint32_t>(), N,
                                   32 + bits_for(t->n_views), s))) return rc;
    // by view: tracks in log order
    hipLaunchKernelGGL(trk_views_of_kernel, grid_for(N), dim3(NT), 0, s, t->ev_key.as<uint64_t>(), (uint32_t)N, t->c32.as<uint32_t>());
    if ((rc = sort_pairs<uint32_t>(t, t->c32.as<uint32_t>(), t->a32.as<uint32_t>(), t->ev_track.as<uint32_t>(), t->view_track.as<uint32_t>(), N,
                                   bits_for(t->n_views), s))) return rc;
    hipLaunchKernelGGL(trk_offsets_kernel, grid_for((size_t)t->n_views + 1), dim3(NT), 0, s, t->a32.as<uint32_t>(), (uint32_t)N, t->n_views, t->view_begin.as<uint32_t>());
    TRK_TRY(hipGetLastError());
    return PGI_SUCCESS;
}

}  // namespace

extern "C" {

pgi_tracklets* pgi_tracklets_create(pgi_ctx* ctx, uint32_t n_views) {
    if (!ctx || !n_views) { pgi::fail(PGI_ERR_INVALID, "pgi_tracklets_create: null context or zero views"); return nullptr; }
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (hipSetDevice(ctx->device) != hipSuccess) { pgi::fail(PGI_ERR_DEVICE, "pgi_tracklets_create: hipSetDevice failed"); return nullptr; }
    pgi_tracklets* t = new pgi_tracklets();
    t->ctx = ctx;
    t->n_views = n_views;
    const size_t vb = ((size_t)n_views + 2) * 4;
    if (t->state.reserve(sizeof(DeviceState), ctx->stream) != hipSuccess || t->view_begin.reserve(vb, ctx->stream) != hipSuccess ||
        t->trk_begin.reserve(8, ctx->stream) != hipSuccess ||
        hipMemsetAsync(t->state.p, 0, sizeof(DeviceState), ctx->stream) != hipSuccess ||
        hipMemsetAsync(t->view_begin.p, 0, vb, ctx->stream) != hipSuccess || hipMemsetAsync(t->trk_begin.p, 0, 8, ctx->stream) != hipSuccess) {
        pgi::fail(PGI_ERR_NOMEM, "pgi_tracklets_create: device allocation failed");
        delete t;
        return nullptr;
    }
    return t;
}

void pgi_tracklets_destroy(pgi_tracklets* t) {
    if (!t) return;
    {
        std::lock_guard<std::mutex> lock(t->ctx->mu);
        (void)hipSetDevice(t->ctx->device);
        (void)hipStreamSynchronize(t->ctx->stream);
    }
    delete t;
}

int pgi_tracklets_info(const pgi_tracklets* t, uint64_t* n_tracks, uint64_t* n_events, uint32_t* last_rounds) {
    if (!t) return pgi::fail(PGI_ERR_INVALID, "pgi_tracklets_info: null store");
    if (n_tracks) *n_tracks = t->n_tracks;
    if (n_events) *n_events = t->n_events;
    if (last_rounds) *last_rounds = t->last_rounds;
    return PGI_SUCCESS;
}

int pgi_tracklets_add_batch(pgi_tracklets* t, const pgi_tracklet_pair* h_pairs, uint32_t n_pairs) {
    if (!t || (!h_pairs && n_pairs)) return pgi::fail(PGI_ERR_INVALID, "pgi_tracklets_add_batch: null argument");
    if (!n_pairs) return PGI_SUCCESS;
    pgi_ctx* ctx = t->ctx;
    std::lock_guard<std::mutex> lock(ctx->mu);
    TRK_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    std::vector<PairDesc> desc(n_pairs);
    size_t m_cap = 0;
    for (uint32_t i = 0; i < n_pairs; ++i) {
        const pgi_tracklet_pair& p = h_pairs[i];
        if (p.view_src >= t->n_views || p.view_dst >= t->n_views || p.view_src == p.view_dst)
            return pgi::fail(PGI_ERR_INVALID, "pgi_tracklets_add_batch: view index out of range, or source == destination");
        if (p.n_max && (!p.d_src || !p.d_dst)) return pgi::fail(PGI_ERR_INVALID, "pgi_tracklets_add_batch: null match arrays");
        desc[i] = PairDesc{p.view_src, p.view_dst, p.n_max, 0, p.d_src, p.d_dst, p.d_mask, p.d_count};
        m_cap += p.n_max;
    }
    if (m_cap >= (1ull << 31)) return pgi::fail(PGI_ERR_TOO_LARGE, "pgi_tracklets_add_batch: more than 2^31 matches in one batch");
    if (!m_cap) return PGI_SUCCESS;
    TRK_TRY(t->pairs.reserve(desc.size() * sizeof(PairDesc), s));
    TRK_TRY(t->pair_off.reserve(((size_t)n_pairs * 2 + 2) * 4, s));
    TRK_TRY(hipMemcpyAsync(t->pairs.p, desc.data(), desc.size() * sizeof(PairDesc), hipMemcpyHostToDevice, s));
    uint32_t* valid = t->pair_off.as<uint32_t>() + n_pairs + 1;
    TRK_TRY(t->keyS.reserve(m_cap * 8, s));
    TRK_TRY(t->keyD.reserve(m_cap * 8, s));
    TRK_TRY(t->a64.reserve(m_cap * 16, s));
    TRK_TRY(t->b64.reserve(m_cap * 16, s));
    DeviceState* st = t->state.as<DeviceState>();
    hipLaunchKernelGGL(trk_count_kernel, dim3(n_pairs), dim3(NT), 0, s, t->pairs.as<PairDesc>(), valid);
    hipLaunchKernelGGL(trk_pair_scan_kernel, dim3(1), dim3(NT), 0, s, valid, t->pair_off.as<uint32_t>(), n_pairs, st);
    hipLaunchKernelGGL(trk_flatten_kernel, dim3(n_pairs), dim3(NT), 0, s, t->pairs.as<PairDesc>(), t->pair_off.as<uint32_t>(),
                       t->keyS.as<uint64_t>(), t->keyD.as<uint64_t>(), t->a64.as<uint64_t>());
    TRK_TRY(hipGetLastError());
    DeviceState h;
    int rc;
    if ((rc = read_state(t, h, s))) return rc;
    const uint32_t M = h.n_matches;
    if (!M) return PGI_SUCCESS;
    const size_t E = (size_t)M * 2;
    // per-point segments and ranks
    for (DevBuf* b : {&t->segS, &t->segD, &t->rankS, &t->rankD, &t->newflag, &t->newrank}) TRK_TRY(b->reserve(((size_t)M + 1) * 4, s));
    TRK_TRY(t->done.reserve(M, s));
    for (DevBuf* b : {&t->a32, &t->b32, &t->c32, &t->d32, &t->seg_lo, &t->seg_hi, &t->seg_head, &t->seg_last, &t->seg_cnt, &t->cursor})
        TRK_TRY(b->reserve(E * 4, s));
    if ((rc = sort_pairs_iota<uint64_t>(t, t->a64.as<uint64_t>(), t->b64.as<uint64_t>(), t->a32.as<uint32_t>(), E, 32 + bits_for(t->n_views), s))) return rc;
    hipLaunchKernelGGL(trk_heads_kernel, grid_for(E), dim3(NT), 0, s, t->b64.as<uint64_t>(), (uint32_t)E, t->b32.as<uint32_t>());
    if ((rc = scan_u32(t, t->b32.as<uint32_t>(), t->c32.as<uint32_t>(), E, true, s))) return rc;
    hipLaunchKernelGGL(trk_segment_kernel, grid_for(E), dim3(NT), 0, s, t->b64.as<uint64_t>(), t->c32.as<uint32_t>(), (uint32_t)E,
                       t->pt_key.as<uint64_t>(), (uint32_t)t->n_events, t->d32.as<uint32_t>(), t->seg_lo.as<uint32_t>(), t->seg_hi.as<uint32_t>());
    hipLaunchKernelGGL(trk_rank_kernel, grid_for(E), dim3(NT), 0, s, t->a32.as<uint32_t>(), t->c32.as<uint32_t>(), t->d32.as<uint32_t>(), (uint32_t)E,
                       t->segS.as<uint32_t>(), t->segD.as<uint32_t>(), t->rankS.as<uint32_t>(), t->rankD.as<uint32_t>());
    TRK_TRY(hipGetLastError());

    // rounds; the event buffer starts at 3 events per match and the batch is simply replayed with a larger one if it
    // overflows (everything a round writes is per-batch scratch; the committed log is read-only until the append)
    size_t cap = std::max<size_t>(t->nev_cap, (size_t)M * 3 + 1024);
    const DeviceState committed = h;
    for (;;) {
        TRK_TRY(t->nev_track.reserve(cap * 4, s));
        TRK_TRY(t->nev_next.reserve(cap * 4, s));
        TRK_TRY(t->nev_key.reserve(cap * 8, s));
        TRK_TRY(t->nev_seq.reserve(cap * 8, s));
        t->nev_cap = cap;
        TRK_TRY(hipMemsetAsync(t->done.p, 0, M, s));
        TRK_TRY(hipMemsetAsync(t->newflag.p, 0, ((size_t)M + 1) * 4, s));
        TRK_TRY(hipMemsetAsync(t->seg_cnt.p, 0, E * 4, s));
        TRK_TRY(hipMemsetAsync(t->cursor.p, 0, E * 4, s));
        DeviceState reset = committed;
        reset.n_new = reset.n_done = reset.overflow = reset.n_new_tracks = 0;
        TRK_TRY(hipMemcpyAsync(st, &reset, sizeof reset, hipMemcpyHostToDevice, s));
        RoundArgs a{};
        a.st = st;
        a.keyS = t->keyS.as<uint64_t>(); a.keyD = t->keyD.as<uint64_t>();
        a.segS = t->segS.as<uint32_t>(); a.segD = t->segD.as<uint32_t>(); a.rankS = t->rankS.as<uint32_t>(); a.rankD = t->rankD.as<uint32_t>();
        a.done = t->done.as<uint8_t>(); a.newflag = t->newflag.as<uint32_t>();
        a.seg_lo = t->seg_lo.as<uint32_t>(); a.seg_hi = t->seg_hi.as<uint32_t>();
        a.seg_head = t->seg_head.as<uint32_t>(); a.seg_last = t->seg_last.as<uint32_t>(); a.seg_cnt = t->seg_cnt.as<uint32_t>();
        a.cursor = t->cursor.as<uint32_t>();
        a.pt_track = t->pt_track.as<uint32_t>();
        a.nev_track = t->nev_track.as<uint32_t>(); a.nev_next = t->nev_next.as<uint32_t>();
        a.nev_key = t->nev_key.as<uint64_t>(); a.nev_seq = t->nev_seq.as<uint64_t>();
        a.nev_cap = (uint32_t)std::min<size_t>(cap, 0xFFFFFFF0u);
        a.n_matches = M;
        a.track_base = t->n_tracks;
        a.first_ever = t->n_events == 0 ? 1 : 0;
        uint32_t rounds = 0, last_done = 0;
        int burst = 4;
        for (;;) {
            for (int r = 0; r < burst; ++r) hipLaunchKernelGGL(trk_round_kernel, grid_for(M), dim3(NT), 0, s, a);
            rounds += burst;
            TRK_TRY(hipGetLastError());
            if ((rc = read_state(t, h, s))) return rc;
            if (h.n_done >= M || h.overflow) break;
            if (h.n_done == last_done) return pgi::fail(PGI_ERR_DEVICE, "pgi_tracklets_add_batch: no match became ready (internal error)");
            last_done = h.n_done;
            burst = std::min(burst * 2, 32);
        }
        t->last_rounds = rounds;
        if (h.overflow & 2u) return pgi::fail(PGI_ERR_TOO_LARGE, "pgi_tracklets_add_batch: a match extended more than 4095 tracks");
        if (!(h.overflow & 1u)) break;
        cap = std::max<size_t>(cap * 2, (size_t)h.n_new + 1024);
        if (cap > 0xFFFFFFF0u) return pgi::fail(PGI_ERR_TOO_LARGE, "pgi_tracklets_add_batch: event buffer beyond 2^32 entries");
    }
    const uint32_t n_new = h.n_new;
    // new tracks are numbered in match order, the order the reference creates them in
    if ((rc = scan_u32(t, t->newflag.as<uint32_t>(), t->newrank.as<uint32_t>(), (size_t)M + 1, false, s))) return rc;
    uint32_t created = 0;
    TRK_TRY(hipMemcpyAsync(&created, t->newrank.as<uint32_t>() + M, 4, hipMemcpyDeviceToHost, s));
    // events back in sequential order
    TRK_TRY(t->a64.reserve((size_t)n_new * 8, s));
    TRK_TRY(t->a32.reserve((size_t)n_new * 4, s));
    if ((rc = sort_pairs_iota<uint64_t>(t, t->nev_seq.as<uint64_t>(), t->a64.as<uint64_t>(), t->a32.as<uint32_t>(), n_new, kSubBits + bits_for(M), s))) return rc;
    const size_t N0 = t->n_events, N1 = N0 + n_new;
    if (N1 >= 0xFFFFFFF0u) return pgi::fail(PGI_ERR_TOO_LARGE, "pgi_tracklets_add_batch: more than 2^32 events");
    TRK_TRY(t->ev_track.reserve(N1 * 4, s, N0 * 4));
    TRK_TRY(t->ev_key.reserve(N1 * 8, s, N0 * 8));
    hipLaunchKernelGGL(trk_append_kernel, grid_for(n_new), dim3(NT), 0, s, t->a32.as<uint32_t>(), t->nev_track.as<uint32_t>(), t->nev_key.as<uint64_t>(),
                       t->newrank.as<uint32_t>(), n_new, t->n_tracks, t->ev_track.as<uint32_t>() + N0, t->ev_key.as<uint64_t>() + N0);
    TRK_TRY(hipGetLastError());
    TRK_TRY(hipStreamSynchronize(s));
    t->n_tracks += created;
    t->n_events = N1;
    return rebuild_views(t, s);
}

int pgi_tracklets_get_batch(pgi_tracklets* t, const uint32_t* h_view_src, const uint32_t* h_view_dst, uint32_t n_queries, uint32_t max_n,
                            uint32_t out_stride, uint32_t* d_src_idx, uint32_t* d_dst_idx, uint32_t* d_count) {
    if (!t || (n_queries && (!h_view_src || !h_view_dst || !d_src_idx || !d_dst_idx || !d_count)))
        return pgi::fail(PGI_ERR_INVALID, "pgi_tracklets_get_batch: null argument");
    if (!n_queries) return PGI_SUCCESS;
    const uint64_t limit = (uint64_t)max_n + 1;  // the reference stops only after exceeding the maximum (point_track.h:626-627)
    if (out_stride < limit) return pgi::fail(PGI_ERR_INVALID, "pgi_tracklets_get_batch: out_stride must be at least max_n + 1");
    pgi_ctx* ctx = t->ctx;
    std::lock_guard<std::mutex> lock(ctx->mu);
    TRK_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    TRK_TRY(t->pair_off.reserve((size_t)n_queries * 8, s));
    uint32_t* dq = t->pair_off.as<uint32_t>();
    TRK_TRY(hipMemcpyAsync(dq, h_view_src, (size_t)n_queries * 4, hipMemcpyHostToDevice, s));
    TRK_TRY(hipMemcpyAsync(dq + n_queries, h_view_dst, (size_t)n_queries * 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(trk_get_kernel, dim3(n_queries), dim3(NT), 0, s, dq, dq + n_queries, t->n_views, t->view_begin.as<uint32_t>(),
                       t->view_track.as<uint32_t>(), t->trk_begin.as<uint32_t>(), t->mem_key.as<uint64_t>(), (uint32_t)limit, out_stride, d_src_idx,
                       d_dst_idx, d_count);
    TRK_TRY(hipGetLastError());
    // the query arrays are pageable host memory: the copies above have completed on return, the kernel is still queued
    return PGI_SUCCESS;
}

// diagnostics / tests: the members of one track in insertion order, (view << 32 | keypoint) each
int pgi_tracklets_track(pgi_tracklets* t, uint64_t index, uint64_t* h_members, uint32_t capacity, uint32_t* n_members) {
    if (!t || !n_members) return pgi::fail(PGI_ERR_INVALID, "pgi_tracklets_track: null argument");
    if (index >= t->n_tracks) return pgi::fail(PGI_ERR_INVALID, "pgi_tracklets_track: no such track");
    pgi_ctx* ctx = t->ctx;
    std::lock_guard<std::mutex> lock(ctx->mu);
    TRK_TRY(hipSetDevice(ctx->device));
    uint32_t be[2];
    TRK_TRY(hipMemcpyAsync(be, t->trk_begin.as<uint32_t>() + index, 8, hipMemcpyDeviceToHost, ctx->stream));
    TRK_TRY(hipStreamSynchronize(ctx->stream));
    *n_members = be[1] - be[0];
    const uint32_t n = std::min(*n_members, capacity);
    if (n && h_members) {
        TRK_TRY(hipMemcpyAsync(h_members, t->mem_key.as<uint64_t>() + be[0], (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
        TRK_TRY(hipStreamSynchronize(ctx->stream));
    }
    return PGI_SUCCESS;
}

}  // extern "C"
