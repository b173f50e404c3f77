// pgi_tracklets.hip -- multi-view tracklets resident in HBM (reference: point_track.h:541-711).
//
// What the reference does, one inlier match (P_s, P_d) at a time: every track holding P_s that does not hold P_d
// gains P_d, every track that held P_d before the match and does not hold P_s gains P_s, and if nothing was extended a
// new two-member track is started (tracks are never merged, so a point sits in several).  getCorrespondences walks the
// destination view's track list in append order.  Observable state is therefore ONE append-only event log
//     event = (track, view, keypoint)       "track gained the point; the view's list gained the track"
// in the sequential order of the reference: a track's members, a view's track list and a point's track list are the
// events with that track / view / point, in log order.  The store keeps three stable orderings of the log (by track, by
// view, by point); a batch's events are sorted by each key and merged in (old before new), O(new log new + total).
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
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include "pgi_internal.hpp"

namespace {

constexpr int kSubBits = 16;  // events per match < 65535
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
    uint32_t n_valid;      // valid events among the reserved slots (set after the rounds)
};

}  // namespace

struct pgi_tracklets {
    pgi_ctx* ctx = nullptr;
    uint32_t n_views = 0;
    uint32_t n_tracks = 0;
    size_t n_events = 0;
    // The log in its three stable orderings (each: sorted keys + payload).  A batch's events are sorted by each key and
    // MERGED in (old before new at equal keys), into the other half of a ping-pong pair.
    DevBuf trk_id[2], mem_key[2];   // by track: track ids (sorted), members
    DevBuf pt_key[2], pt_track[2];  // by point: point keys (sorted), tracks
    DevBuf view_id[2], view_track[2];  // by view: view ids (sorted), tracks
    int cur = 0;                    // which half is current
    DevBuf trk_begin, view_begin;   // offsets: n_tracks + 1, n_views + 1
    DevBuf new_track, new_key;      // the running batch's events in sequential order
    DevBuf iota;                    // 0, 1, 2, ... (value input of the index sorts)
    DevBuf wide_in, wide_out;       // 32-bit keys widened for the 64-bit sort
    DevBuf state;                   // DeviceState
    // scratch
    DevBuf sort_tmp, a64, b64, a32, b32, c32, d32;
    DevBuf pairs, pair_off;
    DevBuf keyS, keyD, segS, segD, rankS, rankD, done, newflag, newrank;
    DevBuf seg_lo, seg_hi, seg_cnt, cursor, run_start, run_len;
    DevBuf nev_track, nev_key, nev_seq;
    size_t nev_cap = 0;
    double slots_per_match = 0;  // of the last batch: sizes the next batch's event buffer
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


// block-wide exclusive prefix of a flag, in thread order; returns the block total through `total`
template <int WAVES = NT / 64>
__device__ inline uint32_t block_rank(bool flag, uint32_t* wave_tot /* WAVES + 1 words of LDS */, uint32_t& total) {
    const uint64_t m = __ballot(flag);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t in_wave = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();  // previous use of wave_tot is over
    if (lane == 0) wave_tot[w] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t before = 0, all = 0;
    for (int i = 0; i < WAVES; ++i) {
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
        st->n_valid = 0;
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
    const uint32_t *seg_start, *seg_lo, *seg_hi;
    uint32_t* seg_cnt;
    uint64_t* cursor;               // per point: (launch that wrote it) << 32 | matches of the batch it has seen
    uint32_t *run_start, *run_len;  // per (point, match) incidence of the batch, in the sorted entry order
    const uint32_t* pt_track;       // committed lists
    uint32_t* nev_track;
    uint64_t *nev_key, *nev_seq;
    uint32_t nev_cap, n_matches, track_base;
    uint32_t list_cap;  // entries per staged list (kListCap or kListCapHuge)
    uint32_t launch;    // 1, 2, ... within the batch
    int first_ever;   // the store was empty when the batch started
};

// tracks per point a wavefront stages in LDS: the usual configuration (four wavefronts per workgroup), and the one the
// host falls back to for the rest of a batch once a longer list shows up (one wavefront per workgroup, 128 KB of LDS)
constexpr uint32_t kListCap = 1024, kListCapHuge = 16384;

__device__ inline void wave_sync() {  // DS operations of one wavefront execute in order: only the compiler needs a fence
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Stages a point's track list in LDS: the committed part [lo, hi) of pt_track, then the runs appended by the `rank`
// lower-ranked matches of the batch (their descriptors sit side by side in the run directory, so neither the directory
// nor the runs are chased: every lane takes one run).  Returns the length, or cap + 1 if it does not fit.
__device__ inline uint32_t stage_list(const RoundArgs& a, uint32_t seg, uint32_t rank, uint32_t* lds, int lane, uint32_t cap) {
    const uint32_t lo = a.seg_lo[seg], n_old = a.seg_hi[seg] - lo, d0 = a.seg_start[seg];
    if (n_old > cap) return cap + 1;
    for (uint32_t i = lane; i < n_old; i += 64) lds[i] = a.pt_track[lo + i];
    uint32_t pos = n_old;
    for (uint32_t r0 = 0; r0 < rank; r0 += 64) {
        const uint32_t r = r0 + lane;
        uint32_t start = 0, len = 0;
        if (r < rank) { start = a.run_start[d0 + r]; len = a.run_len[d0 + r]; }
        uint32_t incl = len;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(incl, o);
            if (lane >= o) incl += y;
        }
        const uint32_t total = __shfl(incl, 63);
        if (pos + total > cap) return cap + 1;
        const uint32_t at = pos + incl - len;
        for (uint32_t c = 0; c < len; ++c) lds[at + c] = a.nev_track[start + c];
        pos += total;
    }
    return pos;
}

// (5) One launch = one level of the dependency order.  A workgroup looks at 256 matches, one lane each; those whose two
// points have seen all lower-ranked matches BEFORE THIS LAUNCH are run by the workgroup's wavefronts, one match per
// wavefront at a time: both track lists are staged in LDS, the lanes test membership and append in list order.
// Cursors carry the launch that wrote them, so a match never builds on something written by the running launch:
// the kernel boundary is the only synchronisation between workgroups (the XCDs' L2s are not coherent with one another
// inside a launch, and agent-scope release/acquire per match -- an L2 write-back and invalidate each -- cost 20x more).
// Event slots are reserved per workgroup from an upper bound (the two list lengths); unused slots are tagged invalid
// and sorted away afterwards.
__device__ inline bool cursor_ready(uint64_t c, uint32_t rank, uint32_t launch) { return (uint32_t)c == rank && (uint32_t)(c >> 32) < launch; }

template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64) trk_round_kernel(RoundArgs a) {
    constexpr int BT = WAVES * 64;
    __shared__ uint32_t wt[WAVES + 1];
    __shared__ uint32_t ready_k[BT], ready_first[BT];
    __shared__ uint32_t slot_base;
    extern __shared__ uint32_t lists[];  // WAVES x 2 x list_cap
    const uint32_t cap = a.list_cap;
    const uint32_t k = blockIdx.x * BT + threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    bool live = k < a.n_matches && !a.done[k];
    uint32_t sS = 0, sD = 0, rS = 0, rD = 0;
    if (live) { sS = a.segS[k]; sD = a.segD[k]; rS = a.rankS[k]; rD = a.rankD[k]; }
    uint32_t finished = 0;
    if (a.st->overflow & 1u) return;  // (set by an earlier launch of the burst) nothing more fits until the host has grown the buffer
    {
        const bool ready = live && cursor_ready(a.cursor[sS], rS, a.launch) && cursor_ready(a.cursor[sD], rD, a.launch);
        uint32_t need = 0;
        if (ready) need = max(2u, (a.seg_hi[sS] - a.seg_lo[sS]) + a.seg_cnt[sS] + (a.seg_hi[sD] - a.seg_lo[sD]) + a.seg_cnt[sD]);
        // ready matches in lane order, and their slots
        uint32_t n_ready;
        const uint32_t r = block_rank<WAVES>(ready, wt, n_ready);
        uint32_t incl = need;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(incl, o);
            if (lane >= o) incl += y;
        }
        __syncthreads();
        if (lane == 63) wt[w] = incl;
        __syncthreads();
        uint32_t before = 0, total = 0;
        for (int i = 0; i < WAVES; ++i) {
            if (i < w) before += wt[i];
            total += wt[i];
        }
        if (threadIdx.x == 0 && total) slot_base = atomicAdd(&a.st->n_new, total);
        __syncthreads();
        if (!n_ready) return;
        if ((uint64_t)slot_base + total > a.nev_cap) {
            // event buffer full: these matches stay pending, the slots stay reserved (and unused: the buffer is pre-filled
            // with the invalid tag); the host grows the buffer past n_new and goes on
            if (threadIdx.x == 0) atomicOr(&a.st->overflow, 1u);
            return;
        }
        if (ready) {
            ready_k[r] = k;
            ready_first[r] = slot_base + before + incl - need;
            live = false;
        }
        __syncthreads();
        uint32_t* LS = lists + (size_t)w * 2 * cap;
        uint32_t* LD = LS + cap;
        for (uint32_t j = w; j < n_ready; j += WAVES) {
            const uint32_t m = ready_k[j], first = ready_first[j];
            const uint32_t mS = a.segS[m], mD = a.segD[m], mrS = a.rankS[m], mrD = a.rankD[m];
            const uint64_t kS = a.keyS[m], kD = a.keyD[m];
            // the id-0 quirk (point_track.h:651-657): the same decision in every lane, written by one
            uint32_t phase = a.st->qphase;
            uint64_t qkey = a.st->qkey;
            if (a.first_ever && m == 0) {
                qkey = kS;
                phase = 1;
                if (lane == 0) { a.st->qkey = qkey; a.st->qphase = 1; }
            } else if (phase == 1 && (kS == qkey || kD == qkey)) {
                phase = 2;
                if (lane == 0) a.st->qphase = 2;
            }
            const bool qS = phase == 2 && kS == qkey, qD = phase == 2 && kD == qkey;
            wave_sync();  // the previous match of this wavefront is done with the lists
            uint32_t nS = stage_list(a, mS, mrS, LS, lane, cap), nD = stage_list(a, mD, mrD, LD, lane, cap);
            wave_sync();
            uint32_t nA = 0, nB = 0;
            if (nS > cap || nD > cap) {
                // a list too long for this configuration: the match stays pending (its slots are given up as invalid)
                // and the host switches to the large-list configuration for the rest of the batch
                if (lane == 0) atomicOr(&a.st->overflow, 4u);
                continue;
            } else {
                // the forgotten first track of the first point ever registered: its list starts one entry later
                const uint32_t* S = LS + (qS && nS ? 1 : 0);
                const uint32_t* D = LD + (qD && nD ? 1 : 0);
                nS -= (qS && nS ? 1 : 0);
                nD -= (qD && nD ? 1 : 0);
                // every track of the source point learns the destination point (point_track.h:664-681) ...
                for (uint32_t i0 = 0; i0 < nS; i0 += 64) {
                    const uint32_t i = i0 + lane;
                    bool add = false;
                    uint32_t t = 0;
                    if (i < nS) {
                        t = S[i];
                        add = !(qD && t == 0);
                        for (uint32_t q = 0; q < nD && add; ++q) add = D[q] != t;
                    }
                    const uint64_t mask = __ballot(add);
                    if (add) {
                        const uint32_t sub = nA + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
                        if (sub < (1u << kSubBits) - 1u) {
                            const uint32_t e = first + sub;
                            a.nev_track[e] = t;
                            a.nev_key[e] = kD;
                            a.nev_seq[e] = ((uint64_t)m << kSubBits) | sub;
                        }
                    }
                    nA += (uint32_t)__popcll(mask);
                }
                // ... and every earlier track of the destination point learns the source point (:683-699)
                for (uint32_t i0 = 0; i0 < nD; i0 += 64) {
                    const uint32_t i = i0 + lane;
                    bool add = false;
                    uint32_t t = 0;
                    if (i < nD) {
                        t = D[i];
                        add = !(qS && t == 0);
                        for (uint32_t q = 0; q < nS && add; ++q) add = S[q] != t;
                    }
                    const uint64_t mask = __ballot(add);
                    if (add) {
                        const uint32_t sub = nA + nB + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
                        if (sub < (1u << kSubBits) - 1u) {
                            const uint32_t e = first + sub;
                            a.nev_track[e] = t;
                            a.nev_key[e] = kS;
                            a.nev_seq[e] = ((uint64_t)m << kSubBits) | sub;
                        }
                    }
                    nB += (uint32_t)__popcll(mask);
                }
                if (nA + nB >= (1u << kSubBits) - 1u) {
                    if (lane == 0) atomicOr(&a.st->overflow, 2u);
                    nA = nB = 0;
                }
            }
            const uint32_t used = nA + nB;
            if (lane == 0) {
                if (used == 0) {
                    // nothing extended: the match starts a track (:701-709); source member first
                    const uint32_t t = a.track_base + m;
                    a.nev_track[first] = t;     a.nev_key[first] = kS;     a.nev_seq[first] = ((uint64_t)m << kSubBits) | 0u;
                    a.nev_track[first + 1] = t; a.nev_key[first + 1] = kD; a.nev_seq[first + 1] = ((uint64_t)m << kSubBits) | 1u;
                    a.newflag[m] = 1;
                    a.run_start[a.seg_start[mS] + mrS] = first;     a.run_len[a.seg_start[mS] + mrS] = 1;
                    a.run_start[a.seg_start[mD] + mrD] = first + 1; a.run_len[a.seg_start[mD] + mrD] = 1;
                    a.seg_cnt[mS] += 1;
                    a.seg_cnt[mD] += 1;
                } else {
                    a.run_start[a.seg_start[mD] + mrD] = first;      a.run_len[a.seg_start[mD] + mrD] = nA;
                    a.run_start[a.seg_start[mS] + mrS] = first + nA; a.run_len[a.seg_start[mS] + mrS] = nB;
                    a.seg_cnt[mD] += nA;
                    a.seg_cnt[mS] += nB;
                }
            }
            if (lane == 0) {
                a.cursor[mS] = ((uint64_t)a.launch << 32) | (mrS + 1);
                a.cursor[mD] = ((uint64_t)a.launch << 32) | (mrD + 1);
                a.done[m] = 1;
                ++finished;
            }
        }
    }
    if (finished) atomicAdd(&a.st->n_done, finished);
}

__global__ void trk_fill64_kernel(uint64_t* p, size_t n, uint64_t value) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = value;
}

// number of valid tags in the sorted tag array = position of the first invalid one
__global__ void trk_count_valid_kernel(const uint64_t* sorted_seq, uint32_t n, uint64_t invalid_seq, uint32_t* out) {
    if (threadIdx.x == 0) *out = lower_bound64(sorted_seq, n, invalid_seq);
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

// Merge of two sorted runs by rank: an old element moves up by the number of new keys strictly below it, a new element
// lands after every old key that is not greater (old before new at equal keys) -- two binary searches per element instead
// of a re-sort of the whole log.
template <class K>
__device__ inline uint32_t lower_bound_t(const K* a, uint32_t n, K key) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}
template <class K>
__device__ inline uint32_t upper_bound_t(const K* a, uint32_t n, K key) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (!(key < a[mid])) lo = mid + 1; else hi = mid;
    }
    return lo;
}
template <class K, class V>
__global__ void trk_merge_kernel(const K* old_k, const V* old_v, uint32_t n_old, const K* new_k, const V* new_v, uint32_t n_new, K* out_k,
                                 V* out_v) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_old) {
        const K k = old_k[i];
        const uint32_t pos = i + lower_bound_t<K>(new_k, n_new, k);
        out_k[pos] = k;
        out_v[pos] = old_v[i];
    } else if (i < n_old + n_new) {
        const uint32_t j = i - n_old;
        const K k = new_k[j];
        const uint32_t pos = j + upper_bound_t<K>(old_k, n_old, k);
        out_k[pos] = k;
        out_v[pos] = new_v[j];
    }
}
__global__ void trk_gather32_kernel(const uint32_t* idx, const uint32_t* src, uint32_t n, uint32_t* dst) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
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
// (1024 lanes: a wave of queries is a few dozen workgroups, each a chain of dependent loads per list chunk -- the wider the
// chunk, the shorter the chain: 128 queries of scripts/tracklets_bench.py 0.98 -> 0.62 ms against 256 lanes)
constexpr int kGetThreads = 1024;
__global__ void __launch_bounds__(kGetThreads) trk_get_kernel(const uint32_t* q_src, const uint32_t* q_dst, uint32_t n_views, const uint32_t* view_begin,
                                                     const uint32_t* view_track, const uint32_t* trk_begin, const uint64_t* mem_key,
                                                     uint32_t limit, uint32_t out_stride, uint32_t* out_src, uint32_t* out_dst,
                                                     uint32_t* out_cnt) {
    __shared__ uint32_t wt[kGetThreads / 64 + 1];
    const uint32_t q = blockIdx.x, vs = q_src[q], vd = q_dst[q];
    uint32_t base = 0;
    if (vs < n_views && vd < n_views) {
        const uint32_t b = view_begin[vd], e = view_begin[vd + 1];
        for (uint32_t c = b; c < e && base < limit; c += kGetThreads) {
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
            const uint32_t r = base + block_rank<kGetThreads / 64>(touches, wt, total);
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

__global__ void trk_iota_kernel(uint32_t* out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = i;
}

// stable sort of (key, value) pairs by the low `bits` of the key
template <class K>
int sort_pairs(pgi_tracklets* t, const K* kin, K* kout, const uint32_t* vin, uint32_t* vout, size_t n, int bits, hipStream_t s) {
    size_t tmp = 0;
    TRK_TRY(rocprim::radix_sort_pairs(nullptr, tmp, kin, kout, vin, vout, n, 0u, (unsigned)bits, s));
    TRK_TRY(t->sort_tmp.reserve(tmp, s));
    TRK_TRY(rocprim::radix_sort_pairs(t->sort_tmp.p, tmp, kin, kout, vin, vout, n, 0u, (unsigned)bits, s));
    return PGI_SUCCESS;
}
// 32-bit keys travel through the 64-bit sort (widened, then narrowed again): one instantiation of the library's sort kernels
// instead of two -- they are most of this file's code object, whose load is paid by the first add of a process
__global__ void trk_widen_kernel(const uint32_t* in, uint32_t n, uint64_t* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i];
}
__global__ void trk_narrow_kernel(const uint64_t* in, uint32_t n, uint32_t* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (uint32_t)in[i];
}
int sort_pairs_u32(pgi_tracklets* t, const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout, size_t n, int bits, hipStream_t s) {
    if (!n) return PGI_SUCCESS;
    TRK_TRY(t->wide_in.reserve(n * 8, s));
    TRK_TRY(t->wide_out.reserve(n * 8, s));
    hipLaunchKernelGGL(trk_widen_kernel, grid_for(n), dim3(NT), 0, s, kin, (uint32_t)n, t->wide_in.as<uint64_t>());
    const int rc = sort_pairs<uint64_t>(t, t->wide_in.as<uint64_t>(), t->wide_out.as<uint64_t>(), vin, vout, n, bits, s);
    if (rc) return rc;
    hipLaunchKernelGGL(trk_narrow_kernel, grid_for(n), dim3(NT), 0, s, t->wide_out.as<uint64_t>(), (uint32_t)n, kout);
    return PGI_SUCCESS;
}

// values = 0, 1, 2, ...: an explicit index array, so that this is the SAME library instantiation as sort_pairs (a counting
// iterator as the value input is a second copy of every sort kernel: the code object of this file was 6.3 MB, and loading
// it cost the first add of a process 17 ms)
template <class K>
int sort_pairs_iota(pgi_tracklets* t, const K* kin, K* kout, uint32_t* vout, size_t n, int bits, hipStream_t s) {
    TRK_TRY(t->iota.reserve(n * 4, s));
    if (n) hipLaunchKernelGGL(trk_iota_kernel, grid_for(n), dim3(NT), 0, s, t->iota.as<uint32_t>(), (uint32_t)n);
    if constexpr (sizeof(K) == 4) return sort_pairs_u32(t, kin, kout, t->iota.as<uint32_t>(), vout, n, bits, s);
    else return sort_pairs<K>(t, kin, kout, t->iota.as<uint32_t>(), vout, n, bits, s);
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

// PGI_TRACKLETS_TIMING=1: wall-clock per phase of add_batch on stderr (each mark synchronises the stream)
struct PhaseTimer {
    bool on;
    hipStream_t s;
    std::chrono::steady_clock::time_point t0;
    explicit PhaseTimer(hipStream_t stream) : on(std::getenv("PGI_TRACKLETS_TIMING") != nullptr), s(stream), t0(std::chrono::steady_clock::now()) {}
    void mark(const char* what) {
        if (!on) return;
        (void)hipStreamSynchronize(s);
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[tracklets] %-28s %9.3f ms\n", what, 1e3 * std::chrono::duration<double>(t1 - t0).count());
        t0 = t1;
    }
};

int read_state(pgi_tracklets* t, DeviceState& h, hipStream_t s) {
    TRK_TRY(hipMemcpyAsync(&h, t->state.p, sizeof h, hipMemcpyDeviceToHost, s));
    TRK_TRY(hipStreamSynchronize(s));
    return PGI_SUCCESS;
}

// merges the batch's n_new events (new_track / new_key, sequential order) into the three orderings; t->n_events is the
// number of events already merged
int merge_batch(pgi_tracklets* t, uint32_t n_new, hipStream_t s) {
    const uint32_t N0 = (uint32_t)t->n_events, N1 = N0 + n_new;
    if (!n_new) return PGI_SUCCESS;
    const int c = t->cur, o = c ^ 1;
    for (DevBuf* b : {&t->a32, &t->b32, &t->c32, &t->d32}) TRK_TRY(b->reserve((size_t)n_new * 4, s));
    TRK_TRY(t->a64.reserve((size_t)n_new * 8, s));
    TRK_TRY(t->b64.reserve((size_t)n_new * 8, s));
    TRK_TRY(t->trk_id[o].reserve((size_t)N1 * 4, s));
    TRK_TRY(t->mem_key[o].reserve((size_t)N1 * 8, s));
    TRK_TRY(t->pt_key[o].reserve((size_t)N1 * 8, s));
    TRK_TRY(t->pt_track[o].reserve((size_t)N1 * 4, s));
    TRK_TRY(t->view_id[o].reserve((size_t)N1 * 4, s));
    TRK_TRY(t->view_track[o].reserve((size_t)N1 * 4, s));
    TRK_TRY(t->trk_begin.reserve(((size_t)t->n_tracks + 2) * 4, s));
    TRK_TRY(t->view_begin.reserve(((size_t)t->n_views + 2) * 4, s));
    const uint32_t* nt = t->new_track.as<uint32_t>();
    const uint64_t* nk = t->new_key.as<uint64_t>();
    int rc;
    if (!N0) {  // nothing to merge with: sort straight into place
        if ((rc = sort_pairs_iota<uint32_t>(t, nt, t->trk_id[o].as<uint32_t>(), t->b32.as<uint32_t>(), n_new, bits_for(t->n_tracks), s))) return rc;
        hipLaunchKernelGGL(trk_gather64_kernel, grid_for(n_new), dim3(NT), 0, s, t->b32.as<uint32_t>(), nk, n_new, t->mem_key[o].as<uint64_t>());
        if ((rc = sort_pairs<uint64_t>(t, nk, t->pt_key[o].as<uint64_t>(), nt, t->pt_track[o].as<uint32_t>(), n_new, 32 + bits_for(t->n_views), s))) return rc;
        hipLaunchKernelGGL(trk_views_of_kernel, grid_for(n_new), dim3(NT), 0, s, nk, n_new, t->d32.as<uint32_t>());
        if ((rc = sort_pairs_u32(t, t->d32.as<uint32_t>(), t->view_id[o].as<uint32_t>(), nt, t->view_track[o].as<uint32_t>(), n_new, bits_for(t->n_views), s))) return rc;
        hipLaunchKernelGGL(trk_offsets_kernel, grid_for((size_t)t->n_tracks + 1), dim3(NT), 0, s, t->trk_id[o].as<uint32_t>(), N1, t->n_tracks, t->trk_begin.as<uint32_t>());
        hipLaunchKernelGGL(trk_offsets_kernel, grid_for((size_t)t->n_views + 1), dim3(NT), 0, s, t->view_id[o].as<uint32_t>(), N1, t->n_views, t->view_begin.as<uint32_t>());
        TRK_TRY(hipGetLastError());
        TRK_TRY(hipStreamSynchronize(s));
        t->cur = o;
        t->n_events = N1;
        return PGI_SUCCESS;
    }
    // by track: (track id, member)
    if ((rc = sort_pairs_iota<uint32_t>(t, nt, t->a32.as<uint32_t>(), t->b32.as<uint32_t>(), n_new, bits_for(t->n_tracks), s))) return rc;
    hipLaunchKernelGGL(trk_gather64_kernel, grid_for(n_new), dim3(NT), 0, s, t->b32.as<uint32_t>(), nk, n_new, t->a64.as<uint64_t>());
    hipLaunchKernelGGL((trk_merge_kernel<uint32_t, uint64_t>), grid_for(N1), dim3(NT), 0, s, t->trk_id[c].as<uint32_t>(), t->mem_key[c].as<uint64_t>(), N0,
                       t->a32.as<uint32_t>(), t->a64.as<uint64_t>(), n_new, t->trk_id[o].as<uint32_t>(), t->mem_key[o].as<uint64_t>());
    hipLaunchKernelGGL(trk_offsets_kernel, grid_for((size_t)t->n_tracks + 1), dim3(NT), 0, s, t->trk_id[o].as<uint32_t>(), N1, t->n_tracks, t->trk_begin.as<uint32_t>());
    // by point: (point key, track)
    if ((rc = sort_pairs<uint64_t>(t, nk, t->b64.as<uint64_t>(), nt, t->c32.as<uint32_t>(), n_new, 32 + bits_for(t->n_views), s))) return rc;
    hipLaunchKernelGGL((trk_merge_kernel<uint64_t, uint32_t>), grid_for(N1), dim3(NT), 0, s, t->pt_key[c].as<uint64_t>(), t->pt_track[c].as<uint32_t>(), N0,
                       t->b64.as<uint64_t>(), t->c32.as<uint32_t>(), n_new, t->pt_key[o].as<uint64_t>(), t->pt_track[o].as<uint32_t>());
    // by view: (view id, track)
    hipLaunchKernelGGL(trk_views_of_kernel, grid_for(n_new), dim3(NT), 0, s, nk, n_new, t->d32.as<uint32_t>());
    if ((rc = sort_pairs_u32(t, t->d32.as<uint32_t>(), t->a32.as<uint32_t>(), nt, t->b32.as<uint32_t>(), n_new, bits_for(t->n_views), s))) return rc;
    hipLaunchKernelGGL((trk_merge_kernel<uint32_t, uint32_t>), grid_for(N1), dim3(NT), 0, s, t->view_id[c].as<uint32_t>(), t->view_track[c].as<uint32_t>(), N0,
                       t->a32.as<uint32_t>(), t->b32.as<uint32_t>(), n_new, t->view_id[o].as<uint32_t>(), t->view_track[o].as<uint32_t>());
    hipLaunchKernelGGL(trk_offsets_kernel, grid_for((size_t)t->n_views + 1), dim3(NT), 0, s, t->view_id[o].as<uint32_t>(), N1, t->n_views, t->view_begin.as<uint32_t>());
    TRK_TRY(hipGetLastError());
    TRK_TRY(hipStreamSynchronize(s));
    t->cur = o;
    t->n_events = N1;
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
    PhaseTimer timer(s);
    TRK_TRY(t->pairs.reserve(desc.size() * sizeof(PairDesc), s));
    TRK_TRY(t->pair_off.reserve(((size_t)n_pairs * 2 + 2) * 4, s));
    TRK_TRY(hipMemcpyAsync(t->pairs.p, desc.data(), desc.size() * sizeof(PairDesc), hipMemcpyHostToDevice, s));
    uint32_t* valid = t->pair_off.as<uint32_t>() + n_pairs + 1;
    TRK_TRY(t->keyS.reserve(m_cap * 8, s));
    TRK_TRY(t->keyD.reserve(m_cap * 8, s));
    TRK_TRY(t->a64.reserve(m_cap * 16, s));
    TRK_TRY(t->b64.reserve(m_cap * 16, s));
    DeviceState* st = t->state.as<DeviceState>();
    timer.mark("(entry, buffers)");
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
    timer.mark("flatten");
    // per-point segments and ranks
    for (DevBuf* b : {&t->segS, &t->segD, &t->rankS, &t->rankD, &t->newflag, &t->newrank}) TRK_TRY(b->reserve(((size_t)M + 1) * 4, s));
    TRK_TRY(t->done.reserve(M, s));
    for (DevBuf* b : {&t->a32, &t->b32, &t->c32, &t->d32, &t->seg_lo, &t->seg_hi, &t->seg_cnt, &t->run_start, &t->run_len})
        TRK_TRY(b->reserve(E * 4, s));
    TRK_TRY(t->cursor.reserve(E * 8, s));
    timer.mark("(buffers)");
    if ((rc = sort_pairs_iota<uint64_t>(t, t->a64.as<uint64_t>(), t->b64.as<uint64_t>(), t->a32.as<uint32_t>(), E, 32 + bits_for(t->n_views), s))) return rc;
    timer.mark("sort incidences");
    hipLaunchKernelGGL(trk_heads_kernel, grid_for(E), dim3(NT), 0, s, t->b64.as<uint64_t>(), (uint32_t)E, t->b32.as<uint32_t>());
    if ((rc = scan_u32(t, t->b32.as<uint32_t>(), t->c32.as<uint32_t>(), E, true, s))) return rc;
    hipLaunchKernelGGL(trk_segment_kernel, grid_for(E), dim3(NT), 0, s, t->b64.as<uint64_t>(), t->c32.as<uint32_t>(), (uint32_t)E,
                       t->pt_key[t->cur].as<uint64_t>(), (uint32_t)t->n_events, t->d32.as<uint32_t>(), t->seg_lo.as<uint32_t>(), t->seg_hi.as<uint32_t>());
    hipLaunchKernelGGL(trk_rank_kernel, grid_for(E), dim3(NT), 0, s, t->a32.as<uint32_t>(), t->c32.as<uint32_t>(), t->d32.as<uint32_t>(), (uint32_t)E,
                       t->segS.as<uint32_t>(), t->segD.as<uint32_t>(), t->rankS.as<uint32_t>(), t->rankD.as<uint32_t>());
    TRK_TRY(hipGetLastError());

    timer.mark("point segments and ranks");
    // rounds: one launch per level until every match is done.  The event buffer grows in place when a launch runs out
    // of slots (the workgroups that found it full did nothing and try again).
    size_t cap = std::max<size_t>(t->nev_cap, (size_t)((double)M * std::max(8.0, 1.5 * t->slots_per_match)) + 1024);
    if (const char* e = std::getenv("PGI_TRACKLETS_EVENT_CAP")) cap = std::max(64, std::atoi(e));  // tests: force the growth path
    TRK_TRY(t->nev_track.reserve(cap * 4, s));
    TRK_TRY(t->nev_key.reserve(cap * 8, s));
    TRK_TRY(t->nev_seq.reserve(cap * 8, s));
    t->nev_cap = cap;
    const uint64_t invalid_seq = (uint64_t)M << kSubBits;  // tag of a slot that holds no event: sorts behind every real one
    hipLaunchKernelGGL(trk_fill64_kernel, grid_for(cap), dim3(NT), 0, s, t->nev_seq.as<uint64_t>(), cap, invalid_seq);
    TRK_TRY(hipMemsetAsync(t->done.p, 0, M, s));
    TRK_TRY(hipMemsetAsync(t->newflag.p, 0, ((size_t)M + 1) * 4, s));
    TRK_TRY(hipMemsetAsync(t->seg_cnt.p, 0, E * 4, s));
    TRK_TRY(hipMemsetAsync(t->cursor.p, 0, E * 8, s));
    timer.mark("(buffers)");
    RoundArgs a{};
    a.st = st;
    a.keyS = t->keyS.as<uint64_t>(); a.keyD = t->keyD.as<uint64_t>();
    a.segS = t->segS.as<uint32_t>(); a.segD = t->segD.as<uint32_t>(); a.rankS = t->rankS.as<uint32_t>(); a.rankD = t->rankD.as<uint32_t>();
    a.done = t->done.as<uint8_t>(); a.newflag = t->newflag.as<uint32_t>();
    a.seg_start = t->d32.as<uint32_t>(); a.seg_lo = t->seg_lo.as<uint32_t>(); a.seg_hi = t->seg_hi.as<uint32_t>();
    a.seg_cnt = t->seg_cnt.as<uint32_t>(); a.cursor = t->cursor.as<uint64_t>();
    a.run_start = t->run_start.as<uint32_t>(); a.run_len = t->run_len.as<uint32_t>();
    a.pt_track = t->pt_track[t->cur].as<uint32_t>();
    a.n_matches = M;
    a.track_base = t->n_tracks;
    a.first_ever = t->n_events == 0 ? 1 : 0;
    uint32_t rounds = 0, last_done = 0;
    int burst = 8;
    // PGI_TRACKLETS_LIST_CAP (tests): a smaller usual configuration, so that the large-list path is exercised
    uint32_t list_cap = kListCap;
    if (const char* e = std::getenv("PGI_TRACKLETS_LIST_CAP")) list_cap = std::min<uint32_t>(kListCap, std::max(8, std::atoi(e)));
    bool huge = false;
    a.list_cap = list_cap;
    // A batch that fails leaves the store as it was: the rounds advance the device-side quirk state (phase, key, counters)
    // and the track count moves before the merge, so EVERY non-success exit from here on -- the explicit give-ups, a
    // failed reserve / memset / scan / sort / merge -- restores both; only a completed merge dismisses the guard.
    struct Rollback {
        DeviceState* st;
        DeviceState committed;
        hipStream_t s;
        pgi_tracklets* t;
        uint32_t tracks;
        bool armed;
        ~Rollback() {
            if (!armed) return;
            (void)hipMemcpyAsync(st, &committed, sizeof committed, hipMemcpyHostToDevice, s);
            (void)hipStreamSynchronize(s);
            t->n_tracks = tracks;
        }
    } rollback{st, h, s, t, t->n_tracks, true};
    auto give_up = [&](int code, const char* msg) { return pgi::fail(code, msg); };
    for (;;) {
        a.nev_track = t->nev_track.as<uint32_t>();
        a.nev_key = t->nev_key.as<uint64_t>(); a.nev_seq = t->nev_seq.as<uint64_t>();
        a.nev_cap = (uint32_t)std::min<size_t>(cap, 0xFFFFFFF0u);
        for (int r = 0; r < burst; ++r) {
            a.launch = ++rounds;
            if (!huge) hipLaunchKernelGGL((trk_round_kernel<4>), grid_for(M, 256), dim3(256), (size_t)4 * 2 * list_cap * 4, s, a);
            else hipLaunchKernelGGL((trk_round_kernel<1>), grid_for(M, 64), dim3(64), (size_t)2 * kListCapHuge * 4, s, a);
        }
        TRK_TRY(hipGetLastError());
        if ((rc = read_state(t, h, s))) return rc;
        if (timer.on) {
            char what[96];
            std::snprintf(what, sizeof what, "  %d launches: %u of %u done%s", burst, h.n_done, M, (h.overflow & 1u) ? " (buffer full)" : "");
            timer.mark(what);
        }
        if ((h.overflow & 2u) || ((h.overflow & 4u) && huge))
            return give_up(PGI_ERR_TOO_LARGE, "pgi_tracklets_add_batch: a keypoint sits in more than 16384 tracks");
        if (h.n_done >= M) break;
        if (h.overflow & 4u) {  // lists beyond the usual configuration: large-list launches from here on
            huge = true;
            a.list_cap = kListCapHuge;
            TRK_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&trk_round_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)(2 * kListCapHuge * 4)));
        }
        if (h.overflow & 1u) {
            if (cap >= 0xFFFFFFF0u) return give_up(PGI_ERR_TOO_LARGE, "pgi_tracklets_add_batch: event buffer beyond 2^32 entries");
            const size_t old = cap;
            cap = std::min<size_t>(std::max<size_t>(cap * 2, (size_t)h.n_new + (size_t)h.n_new / 4), 0xFFFFFFF0u);
            if (h.n_new >= cap) return give_up(PGI_ERR_TOO_LARGE, "pgi_tracklets_add_batch: event buffer beyond 2^32 entries");
            TRK_TRY(t->nev_track.reserve(cap * 4, s, old * 4));
            TRK_TRY(t->nev_key.reserve(cap * 8, s, old * 8));
            TRK_TRY(t->nev_seq.reserve(cap * 8, s, old * 8));
            hipLaunchKernelGGL(trk_fill64_kernel, grid_for(cap - old), dim3(NT), 0, s, t->nev_seq.as<uint64_t>() + old, cap - old, invalid_seq);
            t->nev_cap = cap;
        }
        if (h.overflow) {
            TRK_TRY(hipMemsetAsync(&st->overflow, 0, 4, s));
        } else if (h.n_done == last_done) {
            return give_up(PGI_ERR_DEVICE, "pgi_tracklets_add_batch: no match became ready (internal error)");
        }
        last_done = h.n_done;
        burst = std::min(burst * 2, 32);
    }
    t->last_rounds = rounds;
    timer.mark("rounds");
    const uint32_t n_slots = h.n_new;  // reserved slots; the unused ones carry the invalid tag and sort to the end
    t->slots_per_match = (double)n_slots / (double)M;
    // new tracks are numbered in match order, the order the reference creates them in
    if ((rc = scan_u32(t, t->newflag.as<uint32_t>(), t->newrank.as<uint32_t>(), (size_t)M + 1, false, s))) return rc;
    uint32_t created = 0;
    TRK_TRY(hipMemcpyAsync(&created, t->newrank.as<uint32_t>() + M, 4, hipMemcpyDeviceToHost, s));
    // events back in sequential order
    TRK_TRY(t->a64.reserve((size_t)n_slots * 8, s));
    TRK_TRY(t->a32.reserve((size_t)n_slots * 4, s));
    if ((rc = sort_pairs_iota<uint64_t>(t, t->nev_seq.as<uint64_t>(), t->a64.as<uint64_t>(), t->a32.as<uint32_t>(), n_slots,
                                        kSubBits + bits_for((uint64_t)M + 1), s))) return rc;
    hipLaunchKernelGGL(trk_count_valid_kernel, dim3(1), dim3(64), 0, s, t->a64.as<uint64_t>(), n_slots, invalid_seq, &st->n_valid);
    uint32_t n_new = 0;
    TRK_TRY(hipMemcpyAsync(&n_new, &st->n_valid, 4, hipMemcpyDeviceToHost, s));
    TRK_TRY(hipStreamSynchronize(s));
    if ((size_t)t->n_events + n_new >= 0xFFFFFFF0u) return pgi::fail(PGI_ERR_TOO_LARGE, "pgi_tracklets_add_batch: more than 2^32 events");
    TRK_TRY(t->new_track.reserve((size_t)n_new * 4 + 4, s));
    TRK_TRY(t->new_key.reserve((size_t)n_new * 8 + 8, s));
    hipLaunchKernelGGL(trk_append_kernel, grid_for(n_new), dim3(NT), 0, s, t->a32.as<uint32_t>(), t->nev_track.as<uint32_t>(), t->nev_key.as<uint64_t>(),
                       t->newrank.as<uint32_t>(), n_new, t->n_tracks, t->new_track.as<uint32_t>(), t->new_key.as<uint64_t>());
    TRK_TRY(hipGetLastError());
    TRK_TRY(hipStreamSynchronize(s));
    t->n_tracks += created;
    timer.mark("order events");
    rc = merge_batch(t, n_new, s);
    timer.mark("merge into the three orderings");
    if (rc == PGI_SUCCESS) rollback.armed = false;
    return rc;
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
    hipLaunchKernelGGL(trk_get_kernel, dim3(n_queries), dim3(kGetThreads), 0, s, dq, dq + n_queries, t->n_views, t->view_begin.as<uint32_t>(),
                       t->view_track[t->cur].as<uint32_t>(), t->trk_begin.as<uint32_t>(), t->mem_key[t->cur].as<uint64_t>(), (uint32_t)limit, out_stride, d_src_idx,
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
        TRK_TRY(hipMemcpyAsync(h_members, t->mem_key[t->cur].as<uint64_t>() + be[0], (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
        TRK_TRY(hipStreamSynchronize(ctx->stream));
    }
    return PGI_SUCCESS;
}

}  // extern "C"
