// pgi_kernels.hip -- HIP kernels (gfx950) + C ABI of the pairwise relative-pose engine.
//
// K1 estimate_pose_kernel : one 256-thread workgroup per image pair.  Correspondences are
//    staged once from the flattened (pair,corr) SoA in HBM into LDS as float4 rows;
//    each wavefront solves four 5-point hypotheses at a time (one per 16-lane DPP row),
//    scores the resulting models over all rows with __ballot/popcount, the workgroup
//    keeps the round's best, refits it on its inliers (n-point LO) and stops by the
//    confidence rule; the epilogue writes the inlier mask and decomposes E into (R,t)
//    with a cheirality vote -- "essential + decompose" in a single launch.
//    Replaces PoseGraphBuilder::estimatePose (pose_graph_builder.h:940-1078).
// K2 score_pose_*         : one model per pair -> inlier mask + count (HBM-bound).
//    Replaces EssentialMatrixEvaluator::getInliers / InTraversalPoseTester::test
//    (graph_traversal.h:136-168, 194-233).
// K3 decompose_kernel     : E + rows -> (R,t) (pose_utils.h:144-252).
// K5 five_point_kernel    : minimal solver on explicit samples (parity / debugging).
#include "pgi_device.hpp"
#include "pgi_internal.hpp"

#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace pgi {

// Wavefronts per workgroup (= per image pair) are a TEMPLATE parameter of K1 since round 5: one, two or four.  A pair's fit
// is the same instruction stream whatever the number (hypotheses, scores and merges are order-free), so every choice gives
// the same bits; what changes is how the work of MANY pairs packs onto the chip.  With one wavefront per pair nothing waits at
// a round barrier and no wavefront idles beside wave 0's refits -- 14 % more pairs per second on the dense V = 5000 scene's
// 106 000 pairs -- but a pair then takes four times as long, and so does the wind-down of a launch: small batches stay with
// four (launch_estimate picks; PGI_K1_NW overrides).
constexpr int kMaxNW = 4;
constexpr uint32_t kNw1Pairs = 98304, kNw2Pairs = 12288;  // launch_estimate's rule (batch sizes from which 1 / 2 wavefronts per pair pay)
constexpr int QCAP = 40;       // models per wavefront pass (4 hypotheses x 10 roots)
constexpr double QMAGIC = 393216.0;  // 1.5 * 2^18: summands rounded to multiples of 2^-34

// Graph-cut local optimisation (prm.lo_graph_cut): the labelling context of a pair.  It lives in the workgroup's shared state,
// written once by thread 0 and read by the (noinline) labelling code -- the fit itself carries ONE pointer to it and nothing
// else, so the mode costs the default path no registers (passing the pieces through refit_wave0 spilled 21 more VGPRs in the
// round loop: 2.7x HBM traffic on config 2).
struct GcCtx {
    uint32_t* prev;      // [n] predecessor of a row in its cell's chain (kGcNone: none); global
    uint32_t* msg;       // [n] message words of the sweeps; global
    uint8_t* lab;        // [n] the labelling (the pair's part of the mask output: the epilogue overwrites it at the end)
    uint32_t* last;      // LDS, kGcCells words: last row seen per cell while the chains are built; 64 words of scratch afterwards
    const float* e_lds;  // the model the cut is taken under (sh->bestE)
    uint32_t lambda64;
    uint32_t built;      // the pair's chains exist (built by wave 0 at the first refit)
};

template <int NW>
struct WgShared {
    float bestE[9];
    int best_score;  // -1: none
    uint32_t best_ninl;
    float candE[NW][9];
    int cand_score[NW];
    uint32_t cand_ninl[NW];
    uint32_t cand_hyp[NW];
    uint32_t wave_cnt[NW];
    uint32_t votes[4];
    uint32_t q_count[NW];
    uint32_t q_hyp[NW][QCAP];
    uint32_t pass_ctr;
    uint32_t nbar;  // pre-verification bar of the next round (best's inliers after merge, before LO)
    uint32_t mask_cnt;
    int lo_score;       // results of a refit run by wave 0 (guess path) / refit counter
    uint32_t lo_ninl, lo_ni, lo_runs;
    float loE[9];
    double Rt[21];  // R1[9] R2[9] t[3]
    GcCtx gc;  // graph-cut local optimisation: the pair's labelling context (see GcCtx)
};

struct K1Args {
    const float* x1;
    const float* y1;
    const float* x2;
    const float* y2;
    const uint64_t* off;
    const double* thr;
    const double* guess;
    const uint8_t* has_guess;
    pgi_edge* edges;
    uint8_t* masks;
    uint32_t n_pairs;
    uint32_t pts_cap;  // LDS rows reserved (multiple of 64); 0 = rows stay in HBM/L2
    uint64_t pair_id_base;
    uint64_t seed;
    pgi_params prm;
    unsigned long long* prof;  // kProfSlots counters (profiling builds) or nullptr
    const uint32_t* pair_list;   // size bucket: indices of the pairs of this launch (nullptr: all pairs)
    const uint32_t* pair_count;  // number of valid entries in pair_list
    uint32_t* pair_head;         // persistent class grids: next unclaimed entry of pair_list (nullptr: entry = blockIdx.x)
    // Rows consumed in place from page-locked host memory (nullptr: the rows are at x1..y2).  K1 reads every row once
    // while staging; rows that do not fit in LDS are copied to x1..y2 (then a device mirror) on the way.
    const float* src_x1;
    const float* src_y1;
    const float* src_x2;
    const float* src_y2;
    // graph-cut local optimisation (prm.lo_graph_cut): per-row scratch of the batch (same row offsets as the coordinates) and
    // the byte offset, inside the workgroup's dynamic LDS, of the 16 KB cell table the chain builder uses
    uint32_t* gc_prev;
    uint32_t* gc_msg;
    uint32_t gc_lds_off;
};

// rows either in LDS (float4, NaN padded) or gathered from the SoA in global memory
// LDS_PTS: 0 = every row from HBM/L2, 1 = every row staged in LDS, 2 = hybrid (the first lds_n rows in LDS, the tail from HBM/L2)
template <int LDS_PTS>
struct Rows {
    const float4* lds;
    const float* x1;
    const float* y1;
    const float* x2;
    const float* y2;
    uint32_t n;
    uint32_t lds_n;  // rows resident in LDS (multiple of 64); hybrid launches keep the rest in HBM/L2
    PGI_DEV float4 get(uint32_t i) const {
        if constexpr (LDS_PTS == 1) {
            return lds[i];
        } else if constexpr (LDS_PTS == 2) {
            if (i < lds_n) return lds[i];  // wave-uniform in the scoring loops (64-row steps)
            const float nanv = __builtin_nanf("");
            if (i < n) return make_float4(x1[i], y1[i], x2[i], y2[i]);
            return make_float4(nanv, nanv, nanv, nanv);
        } else {
            const float nanv = __builtin_nanf("");
            if (i < n) return make_float4(x1[i], y1[i], x2[i], y2[i]);
            return make_float4(nanv, nanv, nanv, nanv);
        }
    }
};

PGI_DEV void edge_clear(pgi_edge* e) {
    double* d = reinterpret_cast<double*>(e);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(pgi_edge) / 8); ++i) d[i] = 0.0;
}

// lane index recomputed on the spot (volatile: never merged with -- and kept live from -- an earlier computation)
PGI_DEV int fresh_lane_id() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

PGI_DEV float rfl(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

// Score queue models [m_begin, m_end) in blocks of four; keeps the first maximum that beats
// `floor_score` (the best of earlier rounds: only a strictly better model can matter).
// The counters are wave-uniform popcounts of ballots (SGPR arithmetic).  A block is
// abandoned as soon as none of its models can still exceed the bar -- exact, because the
// levels only add: final score <= partial + 4 * rows left.
template <int LDS_PTS>
PGI_DEV void score_queue(const Rows<LDS_PTS>& rows, uint32_t n, uint32_t npad, const float* queue,
                         const uint32_t* qhyp, int m_begin, int m_end, float thr2, int lane,
                         int floor_score, uint32_t n_bar, int& b_score, uint32_t& b_ninl, uint32_t& b_hyp,
                         int& b_idx) {
    // pre-verification: rows within 1.5*thr among the first 64 must reach a quarter of what a model as
    // good as the bar (n_bar inliers) is expected to show
    const uint32_t k_min = n_bar ? (uint32_t)((16ull * n_bar) / n) : 0u;
    for (int m0 = m_begin; m0 < m_end; m0 += 4) {
        f32x2 e[2][9];  // models (m0, m0 + 1) and (m0 + 2, m0 + 3), packed
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) {
            const int m = min(m0 + mm, m_end - 1);
#pragma unroll
            for (int c = 0; c < 9; ++c) e[mm >> 1][c][mm & 1] = queue[9 * m + c];  // VGPRs: no constant-bus moves
        }
        const int bar = __builtin_amdgcn_readfirstlane(max(floor_score, b_score));  // wave-uniform
        const bool two = m_end - m0 > 2;  // wave-uniform: the second packed pair holds real models (else: copies of the last)
        uint32_t sc[4] = {0, 0, 0, 0}, ni[4] = {0, 0, 0, 0};
        uint32_t alive = 0xFu;
        bool dead = false;
        float4 pnext = rows.get(lane);
        for (uint32_t base = 0; base < npad; base += 64) {
            const float4 p = pnext;
            if constexpr (LDS_PTS != 1) {  // rows that may come from HBM/L2: fetch the next 64 while these are scored
                if (base + 64 < npad) pnext = rows.get(base + 64 + lane);
            } else {
                if (base + 64 < npad) pnext = rows.lds[base + 64 + lane];
            }
            uint32_t c3v[4] = {0, 0, 0, 0};
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                if (pr == 1 && !two) continue;
                f32x2 r2, den;
                sampson_terms2(e[pr], p.x, p.y, p.z, p.w, r2, den);
                const f32x2 t = den * thr2, t0 = t * 0.25f, t1 = t * 0.5625f, t3 = t * 2.25f;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int mm = 2 * pr + h;
                    const uint32_t c0 = __popcll(__ballot(r2[h] < t0[h]));
                    const uint32_t c1 = __popcll(__ballot(r2[h] < t1[h]));
                    const uint32_t c2 = __popcll(__ballot(r2[h] < t[h]));
                    const uint32_t c3 = __popcll(__ballot(r2[h] < t3[h]));
                    sc[mm] += (c0 + c1) + (c2 + c3);
                    ni[mm] += c2;
                    c3v[mm] = c3;
                }
            }
            if (base == 0) {
                alive = (c3v[0] >= k_min ? 1u : 0u) | (c3v[1] >= k_min ? 2u : 0u) | (c3v[2] >= k_min ? 4u : 0u) |
                        (c3v[3] >= k_min ? 8u : 0u);
            }
            const uint32_t seen = min(n, base + 64u);
            const uint32_t smax = max(max((alive & 1u) ? sc[0] : 0u, (alive & 2u) ? sc[1] : 0u),
                                      max((alive & 4u) ? sc[2] : 0u, (alive & 8u) ? sc[3] : 0u));
            if (!alive || (int)(smax + 4u * (n - seen)) <= bar) {
                dead = true;
                break;
            }
        }
        if (dead) continue;
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) {
            if (m0 + mm < m_end && ((alive >> mm) & 1u) && (int)sc[mm] > max(floor_score, b_score)) {
                b_score = (int)sc[mm];
                b_ninl = ni[mm];
                b_hyp = qhyp ? (uint32_t)__builtin_amdgcn_readfirstlane((int)qhyp[m0 + mm]) : 0u;
                b_idx = m0 + mm;
            }
        }
    }
}

// Append the valid models of a wavefront (one per lane) to its queue in lane order.
PGI_DEV int enqueue_models(bool valid, const float E32[9], uint32_t hyp, float* queue, uint32_t* qhyp,
                           int lane) {
    const uint64_t bal = __ballot(valid);
    const int slot = __popcll(bal & ((1ull << lane) - 1ull));
    if (valid) {
#pragma unroll
        for (int c = 0; c < 9; ++c) queue[9 * slot + c] = E32[c];
        qhyp[slot] = hyp;
    }
    return __popcll(bal);
}

// ---- graph-cut local optimisation: the labelling step on ONE wavefront (specification: oracle/pgi_oracle.c, pgo_gc_*) --------
// prev[i] = the last row before i in the same 4-D grid cell.  64 rows per step: every lane looks its cell's last row up, all
// write themselves, and a read-back tells who lost a collision -- only cells that occur twice in one step are sorted out one
// by one (ballot over the lanes of the cell: the nearest lower lane is the predecessor, the highest lane stays in the table).
template <int LDS_PTS>
__device__ __noinline__ void gc_build_chains(const Rows<LDS_PTS>* rows_p, uint32_t n, uint32_t npad, const GcCtx* gp, int lane) {
    const Rows<LDS_PTS> rows = *rows_p;
    const GcCtx g = *gp;
    for (uint32_t c = (uint32_t)lane; c < kGcCells; c += 64u) g.last[c] = kGcNone;
    wave_sync();
    for (uint32_t base = 0; base < npad; base += 64u) {
        const uint32_t i = base + (uint32_t)lane;
        const bool valid = i < n;
        const float4 p = rows.get(i);
        const uint32_t c = valid ? gc_cell(p) : 0u;
        uint32_t prev = valid ? g.last[c] : kGcNone;
        wave_sync();
        if (valid) g.last[c] = i;
        wave_sync();
        const bool lost = valid && g.last[c] != i;
        wave_sync();
        uint64_t rem = __ballot(lost);
        while (rem) {  // (wave-uniform) one cell that occurs more than once in this step
            const int leader = __ffsll((unsigned long long)rem) - 1;
            const uint32_t cl = (uint32_t)__builtin_amdgcn_readlane((int)c, leader);
            const bool mine = valid && c == cl;
            const uint64_t same = __ballot(mine);
            if (mine) {
                const uint64_t lower = same & ((1ull << lane) - 1ull);
                if (lower) prev = base + (uint32_t)(63 - __clzll((long long)lower));
                if ((same >> lane) == 1ull) g.last[c] = i;  // the highest lane of the cell
            }
            rem &= ~same;
        }
        wave_sync();
        if (valid) gc_store(g.prev + i, prev);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wave_sync();
}

// The minimum cut under model e (LDS, 9 floats): forward sweep (messages delta along the chains), backward sweep (labels), both
// 64 rows per step in index order; dependencies inside a step are resolved by iterating over the lanes that have become ready
// (a chain rarely has two rows in one step).  Writes lab[i] in {0, 1} and returns the number of inliers.
template <int LDS_PTS>
__device__ __noinline__ uint32_t gc_label_wave(const Rows<LDS_PTS>* rows_p, uint32_t n, uint32_t npad, float thr2, GcCtx* gp, int lane) {
    const Rows<LDS_PTS> rows = *rows_p;
    if (!__builtin_amdgcn_readfirstlane((int)gp->built)) {  // the pair's chains, once
        gc_build_chains<LDS_PTS>(rows_p, n, npad, gp, lane);
        if (lane == 0) gp->built = 1u;
        wave_sync();
    }
    const GcCtx g = *gp;
    const uint32_t lam = g.lambda64;
    float e[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) e[c] = g.e_lds[c];
    for (uint32_t base = 0; base < npad; base += 64u) {
        const uint32_t i = base + (uint32_t)lane;
        const bool valid = i < n;
        const float4 p = rows.get(i);
        const uint32_t k = valid ? gc_kernel_level(e, p, thr2) : 0u;
        const uint32_t pv = valid ? gc_load(g.prev + i) : kGcNone;
        const bool has = pv != kGcNone, inblk = has && pv >= base;
        int d = 0;
        bool done = !valid;
        if (valid && !inblk) {
            int dp = 0;
            uint32_t kp = 0;
            if (has) {
                const uint32_t w = gc_load(g.msg + pv);
                dp = gc_word_delta(w);
                kp = gc_word_k(w);
            }
            d = gc_delta(k, has, dp, kp, lam);
            done = true;
        }
        const int src = inblk ? (int)(pv - base) : 0;
        for (;;) {
            const uint64_t dm = __ballot(done);
            if (dm == ~0ull) break;
            const uint32_t ws = (uint32_t)__shfl((int)gc_pack(d, k), src);
            if (!done && ((dm >> src) & 1ull)) {
                d = gc_delta(k, true, gc_word_delta(ws), gc_word_k(ws), lam);
                done = true;
            }
        }
        if (valid) gc_store(g.msg + i, gc_pack(d, k));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint32_t cnt = 0;
    uint32_t* succ_of = g.last;  // 64 words of LDS scratch (the cell table is dead once the chains exist)
    for (uint32_t base = npad; base >= 64u;) {
        base -= 64u;
        const uint32_t i = base + (uint32_t)lane;
        const bool valid = i < n;
        const uint32_t w = valid ? gc_load(g.msg + i) : 0u;
        const uint32_t pv = valid ? gc_load(g.prev + i) : kGcNone;
        const uint32_t k = gc_word_k(w) & 31u;
        const int d = gc_word_delta(w);
        succ_of[lane] = 64u;
        wave_sync();
        if (valid && pv != kGcNone && pv >= base) succ_of[pv - base] = (uint32_t)lane;
        wave_sync();
        const uint32_t succ = succ_of[lane];
        wave_sync();
        uint32_t lab = (w >> 6) & 1u;                     // decided by a successor in a higher step ...
        if (succ == 64u && !((w >> 5) & 1u)) lab = d < 0 ? 1u : 0u;  // ... or the end of a chain: its own minimum (ties: outlier)
        bool done = !valid || succ == 64u;
        const int sl = succ == 64u ? 0 : (int)succ;
        for (;;) {
            const uint64_t dm = __ballot(done);
            if (dm == ~0ull) break;
            const uint32_t ls = (uint32_t)__shfl((int)lab, sl), ks = (uint32_t)__shfl((int)k, sl);
            if (!done && ((dm >> sl) & 1ull)) {
                lab = gc_prev_label(ls, ks, d, k, lam);
                done = true;
            }
        }
        if (valid && pv != kGcNone && pv < base) {  // the predecessor sits in a lower step: leave it the decision
            const uint32_t wp = gc_load(g.msg + pv);
            gc_store(g.msg + pv, wp | 32u | (gc_prev_label(lab, k, gc_word_delta(wp), gc_word_k(wp) & 31u, lam) << 6));
        }
        if (valid) g.lab[i] = (uint8_t)lab;
        cnt += (uint32_t)__popcll(__ballot(valid && lab));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    wave_sync();
    return cnt;
}

// (the row source by value: its address must not leave the fit, or the whole fit keeps it in scratch memory)
template <int LDS_PTS>
__device__ __noinline__ uint32_t gc_label_rows(const float4* lds, const float* x1, const float* y1, const float* x2, const float* y2, uint32_t n,
                                               uint32_t lds_n, uint32_t npad, float thr2, GcCtx* gp, int lane) {
    Rows<LDS_PTS> rows;
    rows.lds = lds;
    rows.x1 = x1; rows.y1 = y1; rows.x2 = x2; rows.y2 = y2;
    rows.n = n;
    rows.lds_n = lds_n;
    return gc_label_wave<LDS_PTS>(&rows, n, npad, thr2, gp, lane);
}

// Inlier set of model E at bound tau2 -> exact 9x9 normal matrix in loA (LDS) and the inlier
// count, by ONE wavefront (local optimisation runs on wave 0 while the others solve hypotheses).
// Summands are pre-rounded to 2^-34 so every summation order agrees.  Three sweeps of <= 18
// accumulators keep the register footprint small; tri (45 doubles) is scratch for the triangle.
// (lab != nullptr: the row set is a labelling -- graph-cut local optimisation -- instead of the rows inside the bound)
template <int LDS_PTS>
PGI_DEV uint32_t normal_matrix_wave(const Rows<LDS_PTS>& rows, uint32_t npad, const float E[9], float tau2,
                                    double* loA, double* tri, int lane, const uint8_t* lab = nullptr) {
    uint32_t cnt = 0;
#pragma unroll
    for (int blk = 0; blk < 3; ++blk) {
        constexpr int kLo[3] = {0, 17, 35}, kRow0[3] = {0, 2, 5}, kRow1[3] = {2, 5, 9}, kCnt[3] = {17, 18, 10};
        double S[18];
#pragma unroll
        for (int i = 0; i < 18; ++i) S[i] = 0.0;
        for (uint32_t base = 0; base < npad; base += 64) {
            const float4 p = rows.get(base + lane);
            float r2, den;
            sampson_terms(E, p.x, p.y, p.z, p.w, r2, den);
            const bool in = lab ? (base + (uint32_t)lane < rows.n && lab[base + lane] != 0) : (r2 < tau2 * den);
            if (blk == 0) cnt += __popcll(__ballot(in));
            if (in) {
                const double x1 = p.x, y1 = p.y, x2 = p.z, y2 = p.w;
                double a[9];
                a[0] = x2 * x1; a[1] = x2 * y1; a[2] = x2;
                a[3] = y2 * x1; a[4] = y2 * y1; a[5] = y2;
                a[6] = x1;      a[7] = y1;      a[8] = 1.0;
                constexpr int kTri[9] = {0, 9, 17, 24, 30, 35, 39, 42, 44};
#pragma unroll
                for (int ii = kRow0[blk]; ii < kRow1[blk]; ++ii)
#pragma unroll
                    for (int jj = ii; jj < 9; ++jj) {
                        const int k = kTri[ii] + (jj - ii) - kLo[blk];
                        double t = a[ii] * a[jj];
                        t = (t + QMAGIC) - QMAGIC;
                        S[k] = S[k] + t;
                    }
            }
        }
#pragma unroll
        for (int i = 0; i < kCnt[blk]; ++i) {
            const double v = wave_sum_exact(S[i]);
            if (lane == 0) tri[kLo[blk] + i] = v;
        }
    }
    wave_sync();
    if (lane < 45) {  // unrank lane -> (i,j), i <= j
        int i = 0, rem = lane;
        while (rem >= 9 - i) {
            rem -= 9 - i;
            ++i;
        }
        const int j = i + rem;
        const double v = tri[lane];
        loA[9 * i + j] = v;
        loA[9 * j + i] = v;
    }
    wave_sync();
    return cnt;
}

// Workgroup-cooperative variant (all NW wavefronts; two barriers): used for the FIRST refit after a
// merge, where every wavefront is synchronised anyway.  partial: NW*45 doubles of dead scratch.
template <int LDS_PTS, int NW>
PGI_DEV uint32_t normal_matrix_wg(const Rows<LDS_PTS>& rows, uint32_t npad, const float E[9], float tau2,
                                  double* loA, double* partial, WgShared<NW>* sh, int tid) {
    constexpr int NT = NW * 64;
    const int lane = tid & 63, w = tid >> 6;
    uint32_t cnt = 0;
#pragma unroll
    for (int blk = 0; blk < 3; ++blk) {
        constexpr int kLo[3] = {0, 17, 35}, kRow0[3] = {0, 2, 5}, kRow1[3] = {2, 5, 9}, kCnt[3] = {17, 18, 10};
        double S[18];
#pragma unroll
        for (int i = 0; i < 18; ++i) S[i] = 0.0;
        for (uint32_t base = 0; base < npad; base += NT) {
            const uint32_t i = base + tid;
            const float nanv = __builtin_nanf("");
            const float4 p = (i < npad) ? rows.get(i) : make_float4(nanv, nanv, nanv, nanv);
            float r2, den;
            sampson_terms(E, p.x, p.y, p.z, p.w, r2, den);
            const bool in = r2 < tau2 * den;
            if (blk == 0) cnt += __popcll(__ballot(in));
            if (in) {
                const double x1 = p.x, y1 = p.y, x2 = p.z, y2 = p.w;
                double a[9];
                a[0] = x2 * x1; a[1] = x2 * y1; a[2] = x2;
                a[3] = y2 * x1; a[4] = y2 * y1; a[5] = y2;
                a[6] = x1;      a[7] = y1;      a[8] = 1.0;
                constexpr int kTri[9] = {0, 9, 17, 24, 30, 35, 39, 42, 44};
#pragma unroll
                for (int ii = kRow0[blk]; ii < kRow1[blk]; ++ii)
#pragma unroll
                    for (int jj = ii; jj < 9; ++jj) {
                        const int k = kTri[ii] + (jj - ii) - kLo[blk];
                        double t = a[ii] * a[jj];
                        t = (t + QMAGIC) - QMAGIC;
                        S[k] = S[k] + t;
                    }
            }
        }
#pragma unroll
        for (int i = 0; i < kCnt[blk]; ++i) {
            const double v = wave_sum_exact(S[i]);
            if (lane == 0) partial[45 * w + kLo[blk] + i] = v;
        }
    }
    if (lane == 0) sh->wave_cnt[w] = cnt;
    __syncthreads();
    if (tid < 45) {
        double v = partial[tid];
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) v = v + partial[45 * ww + tid];
        int i = 0, rem = tid;
        while (rem >= 9 - i) {
            rem -= 9 - i;
            ++i;
        }
        const int j = i + rem;
        loA[9 * i + j] = v;
        loA[9 * j + i] = v;
    }
    uint32_t total = 0;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) total += sh->wave_cnt[ww];
    __syncthreads();
    return total;
}

// Tournament-ordered cyclic Jacobi on the 9x9 matrix in LDS (one wavefront, 36 lanes:
// pair m = lane / 9, index l = lane % 9), then the 4 smallest eigenvectors -> basis of
// every group of this wavefront (W = smallest, then Z, Y, X).
PGI_DEV void jacobi9_wave(double* A, double* V, double* basis0, int lane) {
    for (int i = lane; i < 81; i += 64) V[i] = (i % 10 == 0) ? 1.0 : 0.0;
    wave_sync();
    const int m = lane / 9, l = lane - 9 * m;
    const bool act = lane < 36;
    // pair m of round r is ((r + m + 1) mod 9, (r + 8 - m) mod 9): both members advance by one per round, so they are
    // carried and wrapped instead of recomputed (the two modulo-9 reductions per round were integer multiply sequences)
    int pr = act ? m + 1 : 1, qr = act ? 8 - m : 8;
    for (int sw = 0; sw < kJacobiSweeps; ++sw)
        for (int r = 0; r < 9; ++r) {
            const int p = min(pr, qr), q = max(pr, qr);
            pr = pr == 8 ? 0 : pr + 1;
            qr = qr == 8 ? 0 : qr + 1;
            double c = 1.0, s = 0.0;
            if (act) {
                const double apq = A[9 * p + q];
                if (apq != 0.0) {  // one division, two square roots per rotation
                    const double al = A[10 * q] - A[10 * p], be = 2.0 * apq;
                    const double h = sqrt(fma(al, al, be * be));
                    const double d = fabs(al) + h;
                    const double r = sqrt(fma(d, d, be * be));
                    const double inv = 1.0 / r;
                    c = d * inv;
                    s = (al >= 0.0 ? be : -be) * inv;
                }
            }
            wave_sync();
            if (act) {  // column phase: A <- A J, V <- V J
                const double ap = A[9 * l + p], aq = A[9 * l + q];
                A[9 * l + p] = fma(c, ap, -(s * aq));
                A[9 * l + q] = fma(s, ap, c * aq);
                const double vp = V[9 * l + p], vq = V[9 * l + q];
                V[9 * l + p] = fma(c, vp, -(s * vq));
                V[9 * l + q] = fma(s, vp, c * vq);
            }
            wave_sync();
            if (act) {  // row phase: A <- J^T A
                const double ap = A[9 * p + l], aq = A[9 * q + l];
                A[9 * p + l] = fma(c, ap, -(s * aq));
                A[9 * q + l] = fma(s, ap, c * aq);
            }
            wave_sync();
        }
    // four smallest eigenvalues, first minimum wins ties
    double d[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) d[i] = A[10 * i];
    uint32_t taken = 0;
#pragma unroll
    for (int rank = 0; rank < 4; ++rank) {
        int bi = -1;
        double bv = 0.0;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const bool free_i = !((taken >> i) & 1u);
            if (free_i && (bi < 0 || d[i] < bv)) {
                bi = i;
                bv = d[i];
            }
        }
        taken |= 1u << bi;
        if (lane < 9) basis0[9 * (3 - rank) + lane] = V[9 * lane + bi];
    }
    wave_sync();
}

// n-point refit of model E's inlier set (bound tau2), executed by ONE wavefront (wave 0) without any
// workgroup barrier: normal matrix -> 9x9 Jacobi -> Nister back-end -> score the <= 10 roots.
// Outputs the inlier count and the best refit model that beats `floor_score` (r_score = -1: none): the model stays in
// wave 0's queue (LDS) at index r_idx -- the caller copies it from there (nine registers carried out of here were
// spilled around the round loop).
// LDS: A, V and the triangle scratch alias wave 0's unused solver groups; the queue is wave 0's.
template <int LDS_PTS>
PGI_DEV uint32_t refit_wave0(const Rows<LDS_PTS>& rows, uint32_t n, uint32_t npad, const float E[9], float tau2,
                             float thr2, double* wscr0, uint32_t* qhyp0, int lane, int floor_score, uint32_t n_bar,
                             int& r_score, uint32_t& r_ninl, int& r_idx, Prof& prof, int ni_pre = -1,
                             uint32_t lin_pct = 0u, GcCtx* gc = nullptr) {
    double* loA = wscr0 + W_REGA + G_REGA_SZ;  // 81
    double* loV = loA + 81;                    // 81 (ends at W_REGA + 228 <= W_DOUBLES)
    double* tri = wscr0 + G_BASIS_SZ;          // 45 doubles over basis of groups 1..2
    float* queue0 = reinterpret_cast<float*>(wscr0 + W_REGA);
    r_score = -1;
    r_ninl = 0;
    r_idx = -1;
    prof.mark<11>();
    uint32_t ni;
    if (ni_pre >= 0) {
        ni = (uint32_t)ni_pre;  // the workgroup already built A (first refit after a merge)
    } else if (gc) {
        // graph-cut local optimisation: the refit's rows are the minimum cut of the spatial-coherence energy under the best model
        (void)gc_label_rows<LDS_PTS>(rows.lds, rows.x1, rows.y1, rows.x2, rows.y2, n, rows.lds_n, npad, thr2, gc, lane);
        ni = normal_matrix_wave<LDS_PTS>(rows, npad, E, tau2, loA, tri, lane, gc->lab);
    } else {
        ni = normal_matrix_wave<LDS_PTS>(rows, npad, E, tau2, loA, tri, lane);
    }
    prof.mark<12>();
    if (ni < 5) return ni;  // wave-uniform
    jacobi9_wave(loA, loV, wscr0 + W_BASIS, lane);
    prof.mark<13>();
    float E32[9];
    int cnt;
    if (lin_pct && (uint64_t)ni * 100u >= (uint64_t)n * lin_pct) {  // wave-uniform
        // Linear refit: the eigenvector of the smallest eigenvalue (jacobi9_wave left it as basis vector W) is the
        // least-squares solution of the epipolar equations; unit Frobenius norm, f32.  Lanes 0..8 hold one entry each.
        const double ev = wscr0[W_BASIS + 27 + (lane < 9 ? lane : 0)];
        double n2 = 0.0;
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            const double ec = __shfl(ev, c);
            n2 = fma(ec, ec, n2);
        }
        const double inv = 1.0 / sqrt(n2);
#pragma unroll
        for (int c = 0; c < 9; ++c) E32[c] = (float)(__shfl(ev, c) * inv);
        wave_sync();
        cnt = enqueue_models(lane == 0 && (n2 > 0.0), E32, 0u, queue0, qhyp0, lane);
    } else {
        const int g = lane >> 4, s = lane & 15;  // only group 0 holds the refit; groups 1..3 idle along
        const bool valid = backend_group<false, 14, false>(group_scratch(wscr0, g), s, g * 16,
                                                            [](int) { return make_float4(0.f, 0.f, 0.f, 0.f); }, E32, nullptr, prof);
        wave_sync();
        cnt = enqueue_models(valid && g == 0, E32, 0u, queue0, qhyp0, lane);
    }
    wave_sync();
    prof.mark<20>();
    uint32_t b_hyp = 0;
    score_queue<LDS_PTS>(rows, n, npad, queue0, nullptr, 0, cnt, thr2, lane, floor_score, n_bar, r_score, r_ninl, b_hyp,
                         r_idx);
    wave_sync();
    prof.mark<21>();
    return ni;
}

template <int LDS_PTS, bool GUESS, int NW>
__device__ __forceinline__ void estimate_pair(const K1Args& a, const uint32_t pair, char* smem) {
    constexpr int NT = NW * 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave index: uniform, lives in an SGPR
    const uint64_t o = a.off[pair];
    const uint32_t n = (uint32_t)(a.off[pair + 1] - o);
    const uint32_t npad = (n + 63u) & ~63u;

    float4* pts = reinterpret_cast<float4*>(smem);
    double* wscr_all = reinterpret_cast<double*>(smem + (size_t)a.pts_cap * 16);  // NW * W_DOUBLES
    WgShared<NW>* sh = reinterpret_cast<WgShared<NW>*>(wscr_all + NW * W_DOUBLES);
    // sample stash of the variants whose rows may live in HBM/L2 (NW * 4 groups * 5 rows), behind the shared state
    float4* smp_stash = reinterpret_cast<float4*>(reinterpret_cast<char*>(sh) + ((sizeof(WgShared<NW>) + 15) & ~(size_t)15));
    (void)smp_stash;
    double* wscr = wscr_all + w * W_DOUBLES;
    float* queue = reinterpret_cast<float*>(wscr + W_REGA);  // overlays region A after each solve

    Rows<LDS_PTS> rows;
    rows.lds = pts;
    rows.x1 = a.x1 + o; rows.y1 = a.y1 + o; rows.x2 = a.x2 + o; rows.y2 = a.y2 + o;
    rows.n = n;
    rows.lds_n = LDS_PTS != 0 ? min(npad, a.pts_cap) : 0u;

    pgi_edge* edge = a.edges + pair;
    uint8_t* mask = a.masks + o;
    Prof prof;
    prof.start();

    if (LDS_PTS == 1 && npad > a.pts_cap) {  // the caller's max_corr was too small: never overrun the LDS staging area
        for (uint32_t i = tid; i < n; i += NT) mask[i] = 0;
        if (tid == 0) {
            edge_clear(edge);
            edge->status = PGI_EDGE_TOO_MANY_ROWS;
        }
        return;
    }
    // ---- stage the pair: coalesced SoA reads from HBM -> float4 rows in LDS ----
    {
        const bool direct = a.src_x1 != nullptr;  // wave-uniform
        const float* sx1 = direct ? a.src_x1 + o : rows.x1;
        const float* sy1 = direct ? a.src_y1 + o : rows.y1;
        const float* sx2 = direct ? a.src_x2 + o : rows.x2;
        const float* sy2 = direct ? a.src_y2 + o : rows.y2;
        if constexpr (LDS_PTS != 0) {
            const float nanv = __builtin_nanf("");
            for (uint32_t i = tid; i < rows.lds_n; i += NT)
                pts[i] = (i < n) ? make_float4(sx1[i], sy1[i], sx2[i], sy2[i]) : make_float4(nanv, nanv, nanv, nanv);
        }
        if constexpr (LDS_PTS != 1) {
            if (direct) {  // the rows beyond the LDS part are read many times: bring them over the bus once
                float* mx1 = const_cast<float*>(rows.x1);
                float* my1 = const_cast<float*>(rows.y1);
                float* mx2 = const_cast<float*>(rows.x2);
                float* my2 = const_cast<float*>(rows.y2);
                for (uint32_t i = rows.lds_n + tid; i < n; i += NT) {
                    mx1[i] = sx1[i];
                    my1[i] = sy1[i];
                    mx2[i] = sx2[i];
                    my2[i] = sy2[i];
                }
            }
        }
    }
    if (tid == 0) {
        sh->best_score = -1;
        sh->best_ninl = 0;
        sh->votes[0] = sh->votes[1] = sh->votes[2] = sh->votes[3] = 0;
        sh->mask_cnt = 0;
        sh->pass_ctr = 0;
        sh->nbar = 0;
        sh->lo_runs = 0;
        if (a.gc_prev) {  // graph-cut local optimisation: the pair's labelling context
            sh->gc.prev = a.gc_prev + o;
            sh->gc.msg = a.gc_msg + o;
            sh->gc.lab = mask;
            sh->gc.last = reinterpret_cast<uint32_t*>(smem + a.gc_lds_off);
            sh->gc.e_lds = sh->bestE;
            sh->gc.lambda64 = min(a.prm.lo_graph_cut, 255u);
            sh->gc.built = 0;
        }
    }
    __syncthreads();
    prof.mark<0>();

    const double thr = a.thr[pair];
    const float thr2 = (float)(thr * thr);
    const pgi_params prm = a.prm;

    if (n < 5) {
        for (uint32_t i = tid; i < n; i += NT) mask[i] = 0;
        if (tid == 0) {
            edge_clear(edge);
            edge->status = PGI_EDGE_FEW_POINTS;
        }
        return;
    }

    uint32_t out_iters = 0, out_lo = 0, out_used_guess = 0, out_score = 0;
    // (the final model is never carried in registers: it sits in LDS -- sh->loE after an accepted guess refit, sh->bestE in
    // every other path -- and the epilogue reads it there)
    float mask_tau2 = thr2;
    bool success = false, have_model = false;
    uint32_t guess_ninl = 0;

    bool guided_done = false;  // the rotation-guided path delivered the model: skip the robust fit
    // Local optimisation of the current best (sh->bestE): up to lo_iters n-point refits while they improve.  Called by
    // ALL threads right after a merge (the first normal matrix is built by the whole workgroup); the refits themselves
    // run on wave 0, which is the only writer of the best until the next workgroup barrier.
    auto local_optimise = [&]() {
        if constexpr (!GUESS) return;  // only the guess variants call it (keeps the plain kernel's code unchanged)
        int ni_first = -1;
        GcCtx* const gc = a.gc_prev ? &sh->gc : nullptr;  // (uniform: a kernel argument)
        if constexpr (NW > 1) {  // (a one-wavefront workgroup builds every normal matrix inside refit_wave0)
            if (prm.lo_iters && !gc) {  // (graph-cut mode: the row set is wave 0's labelling, no shared first matrix)
                float bE0[9];
#pragma unroll
                for (int c = 0; c < 9; ++c) bE0[c] = sh->bestE[c];
                int td = tid;
                asm volatile("" : "+v"(td));
                ni_first = (int)normal_matrix_wg<LDS_PTS, NW>(rows, npad, bE0, thr2, wscr_all + W_REGA + G_REGA_SZ, wscr_all + W_DOUBLES, sh, td);
            }
        }
        if (w == 0) {
            for (uint32_t it = 0; it < prm.lo_iters; ++it) {
                float bE[9];
#pragma unroll
                for (int c = 0; c < 9; ++c) bE[c] = sh->bestE[c];
                const int cur_best = __builtin_amdgcn_readfirstlane(sh->best_score);
                const uint32_t cur_ninl = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh->best_ninl);
                int r_score, r_idx;
                uint32_t r_ninl;
                int ln = lane;  // opaque copy: keeps the refit's lane-derived addresses out of the round loop's live set
                asm volatile("" : "+v"(ln));
                const uint32_t ni = refit_wave0<LDS_PTS>(rows, n, npad, bE, thr2, thr2, wscr_all, sh->q_hyp[0], ln, cur_best, cur_ninl,
                                                                r_score, r_ninl, r_idx, prof, it == 0 ? ni_first : -1, prm.lo_linear_pct, gc);
                if (ni < 5) break;
                if (lane == 0) sh->lo_runs += 1;
                if (!(r_score > cur_best)) break;
                if (lane == 0) {
                    sh->best_score = r_score;
                    sh->best_ninl = r_ninl;
                }
                if (ln < 9) sh->bestE[ln] = reinterpret_cast<const float*>(wscr_all + W_REGA)[9 * r_idx + ln];
                wave_sync();
            }
        }
    };
    // ---- rotation-guided re-estimation of a guessed pose (guess_mode 1; BASELINE config 5) ----
    // The rotation of a chained pose is metrically meaningful, its translation is not: keep R, re-estimate t.  Every
    // row gives t . (p2 x R p1) = 0, so two rows fix the direction.  32 two-point hypotheses -- eight per wavefront, one
    // per lane -- are scored like any other model, the best seeds the usual local optimisation, and the result is
    // accepted at min_inliers; otherwise the robust fit below runs from scratch.
    if (GUESS && a.has_guess[pair] && prm.guess_mode == 1u) {
        const double* G = a.guess + 12 * (size_t)pair;
        const double R[9] = {G[0], G[1], G[2], G[3], G[4], G[5], G[6], G[7], G[8]};
        const uint64_t rng_g = mix64(a.seed ^ mix64(a.pair_id_base + pair));
        int ln = lane;
        asm volatile("" : "+v"(ln));
        constexpr int kPerWave = 32 / NW;  // 32 two-point hypotheses, shared evenly by the workgroup's wavefronts
        const bool mine = ln < kPerWave;
        const uint32_t h = (uint32_t)(w * kPerWave + (mine ? ln : 0));
        const uint32_t i0 = draw_index(rng_g, 0x40000000u + h, 0u, n);
        uint32_t i1 = i0;
        for (uint32_t k = 1; k < 64u && i1 == i0; ++k) i1 = draw_index(rng_g, 0x40000000u + h, k, n);
        double nv[2][3];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float4 p = rows.get(q == 0 ? i0 : i1);
            const double X1[3] = {(double)p.x, (double)p.y, 1.0}, X2[3] = {(double)p.z, (double)p.w, 1.0};
            double av[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) av[i] = fma(R[3 * i], X1[0], fma(R[3 * i + 1], X1[1], R[3 * i + 2]));
            cross3(X2, av, nv[q]);
        }
        double t[3];
        cross3(nv[0], nv[1], t);
        const double t2 = fma(t[0], t[0], fma(t[1], t[1], t[2] * t[2]));
        const double tx[9] = {0, -t[2], t[1], t[2], 0, -t[0], -t[1], t[0], 0};
        double Eg[9], n2 = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                double sacc = 0;
#pragma unroll
                for (int k = 0; k < 3; ++k) sacc += tx[3 * i + k] * R[3 * k + j];
                Eg[3 * i + j] = sacc;
            }
#pragma unroll
        for (int m = 0; m < 9; ++m) n2 = fma(Eg[m], Eg[m], n2);
        const double inv = 1.0 / sqrt(n2);
        float E32[9];
#pragma unroll
        for (int m = 0; m < 9; ++m) E32[m] = (float)(Eg[m] * inv);
        const bool valid = mine && (t2 > 1e-30) && (n2 > 0.0);
        const int cnt = enqueue_models(valid, E32, h, queue, sh->q_hyp[w], ln);
        wave_sync();
        int gb_score = -1, gb_idx = -1;
        uint32_t gb_ninl = 0, gb_hyp = 0;
        score_queue<LDS_PTS>(rows, n, npad, queue, sh->q_hyp[w], 0, cnt, thr2, ln, -1, 0u, gb_score, gb_ninl, gb_hyp, gb_idx);
        if (gb_idx >= 0 && ln < 9) sh->candE[w][ln] = queue[9 * gb_idx + ln];
        if (lane == 0) {
            sh->cand_score[w] = gb_score;
            sh->cand_ninl[w] = gb_ninl;
            sh->cand_hyp[w] = gb_hyp;
        }
        __syncthreads();
        int rb = -1, rbw = -1;
        uint32_t rbh = 0;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) {  // score desc, hypothesis index asc
            const int cs = sh->cand_score[ww];
            const uint32_t ch = sh->cand_hyp[ww];
            if (cs > rb || (cs == rb && cs >= 0 && ch < rbh)) {
                rb = cs;
                rbw = ww;
                rbh = ch;
            }
        }
        __syncthreads();
        if (rb >= 0) {  // workgroup-uniform
            if (tid == 0) {
                sh->best_score = rb;
                sh->best_ninl = sh->cand_ninl[rbw];
#pragma unroll
                for (int c = 0; c < 9; ++c) sh->bestE[c] = sh->candE[rbw][c];
            }
            __syncthreads();
            local_optimise();
            __syncthreads();
            if (sh->best_ninl >= prm.min_inliers) {
                guided_done = true;
                have_model = true;
                out_used_guess = 1;
                out_iters = 32;
                out_lo = sh->lo_runs;
                out_score = (uint32_t)sh->best_score;
            }
        }
        if (!guided_done) {  // nothing usable: the robust fit starts from scratch
            __syncthreads();
            if (tid == 0) {
                sh->best_score = -1;
                sh->best_ninl = 0;
                sh->lo_runs = 0;
            }
            __syncthreads();
        }
    }

    // ---- pose guess (pose_graph_builder.h:974-1029) ----
    if (GUESS && a.has_guess[pair] && prm.guess_mode != 1u) {  // GUESS: the batch carries guesses (chosen at launch)
        const double* G = a.guess + 12 * (size_t)pair;
        const double R[9] = {G[0], G[1], G[2], G[3], G[4], G[5], G[6], G[7], G[8]};
        const double t[3] = {G[9], G[10], G[11]};
        // E = [t]x R (pose_utils.h:74-86), plain mul/add like the reference
        const double tx[9] = {0, -t[2], t[1], t[2], 0, -t[0], -t[1], t[0], 0};
        double Eg[9], n2 = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                double s = 0;
#pragma unroll
                for (int k = 0; k < 3; ++k) s += tx[3 * i + k] * R[3 * k + j];
                Eg[3 * i + j] = s;
            }
#pragma unroll
        for (int m = 0; m < 9; ++m) n2 = fma(Eg[m], Eg[m], n2);
        const double inv = 1.0 / sqrt(n2);
        float Ef[9];
#pragma unroll
        for (int m = 0; m < 9; ++m) Ef[m] = (float)(Eg[m] * inv);
        const double trunc = 1.5 * thr;
        const float tau2 = prm.guess_quirk ? (float)trunc : (float)(trunc * trunc);
        if (w == 0) {  // all-inlier refit of the guess (:1013-1020) on wave 0; the others wait
            __builtin_amdgcn_s_setprio(3);  // (the workgroup's critical path, as in the round loop)
            int rs0, ri0;
            uint32_t rn0;
            const uint32_t ni0 = refit_wave0<LDS_PTS>(rows, n, npad, Ef, tau2, thr2, wscr_all, sh->q_hyp[0], lane, -1, 0u, rs0, rn0, ri0, prof);
            if (lane == 0) {
                sh->lo_ni = ni0;
                sh->lo_score = rs0;
                sh->lo_ninl = rn0;
            }
            if (ri0 >= 0 && lane < 9) sh->loE[lane] = reinterpret_cast<const float*>(wscr_all + W_REGA)[9 * ri0 + lane];
            __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();
        const uint32_t ni = sh->lo_ni;
        const int r_score = sh->lo_score;
        if (ni >= 5 && r_score >= 0 && ni >= prm.min_inliers) {
            success = true;
            have_model = true;
            out_used_guess = 1;
            out_lo = 1;
            out_score = (uint32_t)r_score;
            guess_ninl = ni;
            mask_tau2 = tau2;
            // the mask of the guess path is the guess's inlier set (:1000-1009)
            if (tid == 0) {
#pragma unroll
                for (int c = 0; c < 9; ++c) sh->bestE[c] = Ef[c];
            }
            __syncthreads();
        }
    }

    // ---- robust fit (pose_graph_builder.h:1031-1055) ----
    if (!success && !guided_done) {
        const uint64_t rng_base = mix64(a.seed ^ mix64(a.pair_id_base + pair));
        const uint32_t rs = prm.round_size ? prm.round_size : 32u;
        const uint32_t budget = prm.fixed_budget ? prm.fixed_budget : prm.max_iters;
        uint32_t hyps = 0;
        const uint32_t n_pass = (rs + 3u) / 4u;
        // best of the round this wavefront is working on (first maximum in hypothesis order)
        int wb_score = -1;  // its model is parked in sh->candE[w] (LDS), not in registers
        uint32_t wb_ninl = 0, wb_hyp = 0;
        // one pass = four hypotheses (one per 16-lane group): sample, solve, score
        auto do_pass = [&](uint32_t pass, uint32_t base_hyp, int floor_score, uint32_t n_bar) {
            // Lane-derived LDS addresses are recomputed per pass from an opaque copy of the lane id: otherwise the
            // compiler hoists dozens of them out of the round loop, keeps them live across the whole fit and spills.
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const int g = ln >> 4, s = ln & 15;
            const uint32_t local = pass * 4 + g;
            const bool active = local < rs;
            const uint32_t hyp = base_hyp + (active ? local : 0u);
            const GroupScratch gs = group_scratch(wscr, g);
            uint32_t idx[5];
            // (region A of the group is dead between passes: the five accepted draws travel through its first words)
            sample5_group(rng_base, hyp, prm.sampler ? progressive_rows(hyp, n, prm.max_iters) : n, s, g * 16, ln,
                          reinterpret_cast<uint32_t*>(gs.rega), idx);
            uint32_t my = idx[0];
#pragma unroll
            for (int k = 1; k < 5; ++k)
                if (s == k) my = idx[k];
            const float4 mine = rows.get(my);
            prof.mark<1>();
            nullspace5_group(mine, s, g * 16, gs);
            prof.mark<2>();
            float E32[9];
            // The orientation test needs the five sample rows again.  Rows in LDS: re-read them (cheaper than keeping
            // 20 registers live through the solve).  Rows that may sit in HBM/L2 (hybrid / global variants): the lanes
            // that gathered them park them in a small LDS stash (sub-lane k of the group holds sample k) -- a second
            // round trip to memory per root would cost far more.
            if constexpr (LDS_PTS != 1) {
                if (s < 5) smp_stash[(w * 4 + g) * 5 + s] = mine;
                wave_sync();
            }
            const bool valid = backend_group<false, 3, true>(
                gs, s, g * 16,
                [&](int i) {
                    if constexpr (LDS_PTS == 1) return rows.get(idx[i]);
                    else return smp_stash[(w * 4 + g) * 5 + i];
                },
                E32, nullptr, prof);
            wave_sync();  // every group is done with region A: the queue may overlay it
            const int cnt = enqueue_models(valid && active, E32, hyp, queue, sh->q_hyp[w], ln);
            wave_sync();
            prof.mark<9>();
            int bidx = -1;
            score_queue<LDS_PTS>(rows, n, npad, queue, sh->q_hyp[w], 0, cnt, thr2, ln, floor_score, n_bar, wb_score,
                                 wb_ninl, wb_hyp, bidx);
            if (bidx >= 0 && ln < 9) sh->candE[w][ln] = queue[9 * bidx + ln];
            wave_sync();
            prof.mark<10>();
        };
        // wavefronts pull passes of the current round from a shared counter
        auto pull_pass = [&]() -> uint32_t {
            uint32_t pass = 0;
            if (lane == 0) pass = atomicAdd(&sh->pass_ctr, 1u);
            return (uint32_t)__builtin_amdgcn_readfirstlane((int)pass);
        };
        // Local optimisation runs on wave 0 ALONE while the other wavefronts already solve the next round:
        // hypotheses never depend on the current best, so the result equals the sequential order
        // (merge -> LO -> termination test -> next round); the termination test of an improving round
        // is simply evaluated one barrier later, and the speculative round is dropped if it fires.
        int floor_score = -1;   // best score after the last merge (pre-LO): exact bail-out bar
        uint32_t n_bar = 0;     // pre-verification bar of the round in flight (spec: post-merge, pre-LO)
        bool deferred = false;  // an LO may be running on wave 0; its termination test is pending
        auto terminated = [&](uint32_t at_hyps) {
            if (prm.fixed_budget || sh->best_score < 0 || sh->best_ninl < 5) return false;
            // (opaque copies: (double)n and 1 - confidence are loop invariants the compiler would otherwise hoist out of
            // the round loop and keep -- spilled -- across it)
            uint32_t nn = n;
            double conf = prm.confidence;
            asm volatile("" : "+s"(nn), "+s"(conf));
            const double rho = (double)sh->best_ninl / (double)nn;
            const double r5 = ((rho * rho) * (rho * rho)) * rho;
            const double q = 1.0 - r5;
            return pow_uint(q, at_hyps) <= 1.0 - conf;
        };
        while (hyps < budget) {
            for (;;) {
                const uint32_t pass = pull_pass();
                if (pass >= n_pass) break;
                do_pass(pass, hyps, floor_score, n_bar);
            }
            if (lane == 0) {
                sh->cand_score[w] = wb_score;
                sh->cand_ninl[w] = wb_ninl;
                sh->cand_hyp[w] = wb_hyp;
            }
            wb_score = -1;
            __syncthreads();  // A: the round's passes are done, and so is a concurrent LO on wave 0
            prof.mark<22>();
            if (deferred) {  // termination test of the previous (improving) round with its post-LO best
                deferred = false;
                if (terminated(hyps)) break;  // the round just computed is discarded (hyps not advanced)
            }
            hyps += rs;
            // round best: score desc, hypothesis index asc
            int rb = -1, rbw = -1;
            uint32_t rbh = 0;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) {
                const int cs = sh->cand_score[ww];
                const uint32_t ch = sh->cand_hyp[ww];
                if (cs > rb || (cs == rb && cs >= 0 && ch < rbh)) {
                    rb = cs;
                    rbw = ww;
                    rbh = ch;
                }
            }
            const bool improve = (rb >= 0) && (rb > sh->best_score);
            __syncthreads();  // B: everyone has read the candidates and the best
            if (tid == 0) {
                sh->pass_ctr = 0;
                if (improve) {
                    sh->best_score = rb;
                    sh->best_ninl = sh->cand_ninl[rbw];
                    sh->nbar = sh->cand_ninl[rbw];
#pragma unroll
                    for (int c = 0; c < 9; ++c) sh->bestE[c] = sh->candE[rbw][c];
                } else {
                    sh->nbar = sh->best_score >= 0 ? sh->best_ninl : 0u;
                }
            }
            __syncthreads();  // C
            floor_score = __builtin_amdgcn_readfirstlane(sh->best_score);
            n_bar = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh->nbar);
            if (improve) {
                // (same steps as local_optimise() above, spelled out: inside the round loop the compiler schedules the
                // inline form measurably better than the shared lambda -- 2 % on BASELINE config 2)
                int ni_first = -1;
                GcCtx* const gc = a.gc_prev ? &sh->gc : nullptr;  // (uniform: a kernel argument)
                if (NW > 1 && prm.lo_iters && !gc) {  // every wavefront is here anyway: build the first refit's normal matrix together
                    float bE0[9];
#pragma unroll
                    for (int c = 0; c < 9; ++c) bE0[c] = sh->bestE[c];
                    // thread id rebuilt from the wave index (uniform) and a fresh lane count: no VGPR carried -- and
                    // spilled -- across the round loop for this once-per-improvement step
                    const int td = (w << 6) + fresh_lane_id();
                    ni_first = (int)normal_matrix_wg<LDS_PTS, NW>(rows, npad, bE0, thr2, wscr_all + W_REGA + G_REGA_SZ,
                                                              wscr_all + W_DOUBLES, sh, td);
                }
                if (w == 0) {  // n-point refits while they improve (only wave 0 touches the best from here to A)
                    // The refits are the workgroup's critical path (the other wavefronts finish the next round's passes
                    // and then wait at barrier A): let this wavefront issue ahead of its SIMD's other wavefronts.
                    __builtin_amdgcn_s_setprio(3);
                    for (uint32_t it = 0; it < prm.lo_iters; ++it) {
                        float bE[9];
#pragma unroll
                        for (int c = 0; c < 9; ++c) bE[c] = sh->bestE[c];
                        const int cur_best = __builtin_amdgcn_readfirstlane(sh->best_score);
                        const uint32_t cur_ninl = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh->best_ninl);
                        int r_score, r_idx;
                        uint32_t r_ninl;
                        int ln = lane;  // opaque copy: keeps the refit's lane-derived addresses out of the round loop's live set
                        asm volatile("" : "+v"(ln));
                        const uint32_t ni = refit_wave0<LDS_PTS>(rows, n, npad, bE, thr2, thr2, wscr_all, sh->q_hyp[0], ln, cur_best,
                                                                       cur_ninl, r_score, r_ninl, r_idx, prof,
                                                                       it == 0 ? ni_first : -1, prm.lo_linear_pct, gc);
                        if (ni < 5) break;
                        if (lane == 0) sh->lo_runs += 1;
                        if (!(r_score > cur_best)) break;
                        if (lane == 0) {
                            sh->best_score = r_score;
                            sh->best_ninl = r_ninl;
                        }
                        if (ln < 9) sh->bestE[ln] = reinterpret_cast<const float*>(wscr_all + W_REGA)[9 * r_idx + ln];
                        wave_sync();
                    }
                    __builtin_amdgcn_s_setprio(0);
                }
                if constexpr (NW == 1) {  // nobody solved a speculative round meanwhile: the sequential order itself
                    if (terminated(hyps)) break;
                } else {
                    deferred = true;
                }
            } else if (terminated(hyps)) {
                break;
            }
        }
        __syncthreads();  // a refit may still be running on wave 0 when the budget ends the loop
        out_lo = sh->lo_runs;
        out_iters = hyps;
        have_model = sh->best_score >= 0;
        if (have_model) out_score = (uint32_t)sh->best_score;
    }

    prof.mark<23>();
    // thread / lane ids rebuilt here (wave index is uniform, the lane count is recomputed): the epilogue then keeps no
    // VGPR alive across the whole fit
    const int ew = w, etid = (ew << 6) + fresh_lane_id(), elane = etid & 63;
    // ---- epilogue: mask, count, decomposition (pose_graph_builder.h:1057-1075) ----
    if (!have_model) {
        for (uint32_t i = etid; i < n; i += NT) mask[i] = 0;
        if (etid == 0) {
            edge_clear(edge);
            edge->status = PGI_EDGE_FEW_INLIERS;
            edge->iters = out_iters;
            edge->lo_runs = out_lo;
        }
        return;
    }
    const float* finalE = success ? sh->loE : sh->bestE;  // LDS
    float maskE[9];  // the guess path masks with the guess itself (:1000-1009), parked in sh->bestE
#pragma unroll
    for (int c = 0; c < 9; ++c) maskE[c] = sh->bestE[c];
    // Wave 0 decomposes the model (3x3 SVD spread over three lanes, pgi_device.hpp) while waves 1..3 write the inlier
    // mask; the cheirality vote needs both and follows the barrier.
    uint32_t mc = 0;
    if (ew == 0) decompose_wave(finalE, sh->Rt, elane);
    if (NW == 1 || ew != 0) {  // (a one-wavefront workgroup does both, one after the other)
        constexpr uint32_t kMaskLanes = NW == 1 ? 64u : (uint32_t)(NT - 64);
        for (uint32_t base = 0; base < npad; base += kMaskLanes) {
            const uint32_t i = base + (uint32_t)(NW == 1 ? etid : etid - 64);
            const float nanv = __builtin_nanf("");
            const float4 p = (i < npad) ? rows.get(i) : make_float4(nanv, nanv, nanv, nanv);
            float r2, den;
            sampson_terms(maskE, p.x, p.y, p.z, p.w, r2, den);
            const bool in = r2 < mask_tau2 * den;
            if (i < n) mask[i] = in ? 1 : 0;
            mc += __popcll(__ballot(in));
        }
        if (elane == 0) atomicAdd(&sh->mask_cnt, mc);
    }
    __syncthreads();
    double R1[9], R2[9], tt[3];
#pragma unroll
    for (int c = 0; c < 9; ++c) {
        R1[c] = sh->Rt[c];
        R2[c] = sh->Rt[9 + c];
    }
    tt[0] = sh->Rt[18]; tt[1] = sh->Rt[19]; tt[2] = sh->Rt[20];
    uint32_t v0 = 0, v1 = 0, v2 = 0, v3 = 0;
    for (uint32_t base = 0; base < npad; base += NT) {
        const uint32_t i = base + etid;
        const float nanv = __builtin_nanf("");
        const float4 p = (i < npad) ? rows.get(i) : make_float4(nanv, nanv, nanv, nanv);
        float r2, den;
        sampson_terms(maskE, p.x, p.y, p.z, p.w, r2, den);
        const bool in = r2 < mask_tau2 * den;
        const bool voter = (i < n) && (prm.vote_all_rows || in);
        uint32_t b1 = 0, b2 = 0;
        if (voter) {
            const double X1[3] = {p.x, p.y, 1.0}, X2[3] = {p.z, p.w, 1.0};
            double x2t[3];
            cross3(X2, tt, x2t);
            b1 = cheirality_bits(R1, tt, X1, X2, x2t);
            b2 = cheirality_bits(R2, tt, X1, X2, x2t);
        }
        v0 += __popcll(__ballot(b1 & 1u));
        v1 += __popcll(__ballot(b1 & 2u));
        v2 += __popcll(__ballot(b2 & 1u));
        v3 += __popcll(__ballot(b2 & 2u));
    }
    if (elane == 0) {
        atomicAdd(&sh->votes[0], v0);
        atomicAdd(&sh->votes[1], v1);
        atomicAdd(&sh->votes[2], v2);
        atomicAdd(&sh->votes[3], v3);
    }
    __syncthreads();
    if (etid == 0) {
        edge_clear(edge);
        const uint32_t n_inl = success ? guess_ninl : sh->mask_cnt;
#pragma unroll
        for (int c = 0; c < 9; ++c) edge->E[c] = (double)finalE[c];
        edge->n_inl = n_inl;
        edge->score = out_score;
        edge->iters = out_iters;
        edge->lo_runs = out_lo;
        edge->used_guess = out_used_guess;
        if (n_inl < prm.min_inliers) {
            edge->status = PGI_EDGE_FEW_INLIERS;  // :1053-1054 (no decomposition)
        } else {
            const uint32_t vt0 = sh->votes[0], vt1 = sh->votes[1], vt2 = sh->votes[2], vt3 = sh->votes[3];
            uint32_t best = 0, bv = vt0;
            if (vt1 > bv) { bv = vt1; best = 1; }
            if (vt2 > bv) { bv = vt2; best = 2; }
            if (vt3 > bv) { bv = vt3; best = 3; }
            const bool second = (best >> 1) != 0;
            bool bad = false;
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                const double rv = second ? R2[c] : R1[c];
                edge->R[c] = rv;
                bad |= !(rv == rv);
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                edge->t[c] = (best & 1u) ? -tt[c] : tt[c];
                bad |= !(tt[c] == tt[c]);
            }
            edge->votes = bv;
            edge->cand = best;
            edge->status = bad ? PGI_EDGE_NAN : PGI_EDGE_OK;  // :1069-1070
            if (!bad) {
                // E of the record = the essential matrix OF THE RETURNED POSE, [t]x R (pose_utils.h:74-86) at unit Frobenius
                // norm with the fitted model's sign: rank 2 whatever refit the model came from (the reference's E is
                // cv::findEssentialMat's, pose_graph_builder.h:1057-1066).  Operation order of pgo_estimate_pose.
                const double ts = (best & 1u) ? -1.0 : 1.0;
                const double t0 = ts * tt[0], t1 = ts * tt[1], t2 = ts * tt[2];
                const double tx[9] = {0, -t2, t1, t2, 0, -t0, -t1, t0, 0};
                double Ex[9], dot = 0.0;
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        double sacc = 0;
#pragma unroll
                        for (int k = 0; k < 3; ++k) sacc += tx[3 * i + k] * (second ? R2[3 * k + j] : R1[3 * k + j]);
                        Ex[3 * i + j] = sacc;
                    }
#pragma unroll
                for (int m = 0; m < 9; ++m) dot = fma(Ex[m], (double)finalE[m], dot);
                const double sc = dot < 0.0 ? -0.70710678118654752440 : 0.70710678118654752440;
#pragma unroll
                for (int m = 0; m < 9; ++m) edge->E[m] = Ex[m] * sc;
            }
        }
    }
    prof.mark<24>();
    prof.flush(a.prof, lane);
}

// The kernel: one pair per workgroup (grid = pairs), or -- a size-bucket launch -- workgroups that take entries of the
// bucket's list: entry blockIdx.x when the grid covers the whole list, or (round 4, the default for the variants WITHOUT the
// guess path) as a PERSISTENT grid of as many workgroups as the chip keeps resident, each taking whatever entry the shared
// head counter hands out next until the list is empty; the claimed index travels through the first word of the row area,
// which is dead between pairs.  No 10 000-workgroup grid of an empty class rides along any more, and BASELINE config 2's
// launch is 3.5 % shorter (6.35 -> 6.13 ms, identical results: scripts/k1_variants_ab.sh).  The loop around the fit costs
// registers -- 2 / 26 / 4 spilled VGPRs in the three variants, none of them inside the hypothesis passes (same time as the
// loop-free build at every N when the grid covers the list) -- but 64-164 in the guess variants, which therefore keep the
// one-pair form (-DPGI_K1_LOOP_GUESS builds them with the loop for experiments).
template <int LDS_PTS, bool GUESS, int NW>
__device__ __forceinline__ void estimate_pose_body(const K1Args& a, char* smem) {
#ifdef PGI_K1_LOOP_GUESS
    constexpr bool kLoop = true;
#else
    constexpr bool kLoop = !GUESS;
#endif
    if constexpr (kLoop) {
        const bool persistent = a.pair_list != nullptr && a.pair_head != nullptr;  // uniform
        uint32_t slot = blockIdx.x;
        for (;;) {
            if (persistent) {
                uint32_t* claim = reinterpret_cast<uint32_t*>(smem);
                if (threadIdx.x == 0) *claim = atomicAdd(a.pair_head, 1u);
                __syncthreads();
                slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)*claim);
                __syncthreads();  // everyone has read the claim: the row area is free again
            }
            uint32_t pair = slot;
            if (a.pair_list) {  // this launch serves one size bucket (its own LDS size => its own occupancy)
                if (slot >= *a.pair_count) return;
                pair = a.pair_list[slot];
            }
            estimate_pair<LDS_PTS, GUESS, NW>(a, pair, smem);
            if (!persistent) return;
            __syncthreads();  // the pair's results are written, its LDS state is dead
        }
    } else {
        uint32_t pair = blockIdx.x;
        if (a.pair_list) {
            if (blockIdx.x >= *a.pair_count) return;
            pair = a.pair_list[blockIdx.x];
        }
        estimate_pair<LDS_PTS, GUESS, NW>(a, pair, smem);
    }
}

template <int LDS_PTS, bool GUESS, int NW>
__global__ __launch_bounds__(NW * 64, 4) void estimate_pose_kernel(const K1Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef PGI_PROFILE
    // slot occupancy of a launch (instrumented build only, scripts/profile_phases.py): when its workgroups came and went on the
    // 100 MHz wall clock, per rows variant: [0] sum of residence times, [1] last exit, [2] ~first start, [3] workgroups
    const unsigned long long wg_t0 = wall_clock64();
#endif
    estimate_pose_body<LDS_PTS, GUESS, NW>(a, smem);
#ifdef PGI_PROFILE
    if (a.prof && threadIdx.x == 0) {
        const unsigned long long wg_t1 = wall_clock64();
        unsigned long long* o = a.prof + kProfSlots + 4 * LDS_PTS;
        atomicAdd(o + 0, wg_t1 - wg_t0);
        atomicMax(o + 1, wg_t1);
        atomicMax(o + 2, ~wg_t0);
        atomicAdd(o + 3, 1ull);
    }
#endif
}

// Size buckets: dynamic LDS is per launch, so ragged batches are split by row count and every bucket
// is launched with the LDS (hence the occupancy) its pairs need.  caps[b] = largest row count of
// bucket b (ascending: 4, 3, 2, 1 workgroups per CU); pairs above caps[3] go to bucket 4 (rows stay in HBM/L2).
__global__ __launch_bounds__(256) void bucket_pairs_kernel(const uint64_t* __restrict__ off, uint32_t n_pairs, uint32_t cap0,
                                                           uint32_t cap1, uint32_t cap2, uint32_t cap3, uint32_t* __restrict__ lists,
                                                           uint32_t* __restrict__ counts) {
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int b = -1;
    if (p < n_pairs) {
        const uint32_t n = (uint32_t)(off[p + 1] - off[p]);
        b = n <= cap0 ? 0 : n <= cap1 ? 1 : n <= cap2 ? 2 : n <= cap3 ? 3 : 4;
    }
    // One atomic per wavefront and class (r02: one per pair -- 10 000 same-address atomics were 116 us of every launch):
    // the lanes of a class are counted by ballot, lane 0 reserves the block, every lane takes its rank inside it.
    // The order inside a bucket is irrelevant: results are per pair.
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const uint64_t m = __ballot(b == k);
        if (!m) continue;  // wave-uniform
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&counts[k], (uint32_t)__popcll(m));
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (b == k) lists[(size_t)k * n_pairs + base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = p;
    }
}

// ------------------------------------------------------------------------------------------------
// K2: one model per pair; one wavefront per pair streams the SoA rows with float4 loads.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void score_pose_kernel(const float* __restrict__ x1, const float* __restrict__ y1,
                                                         const float* __restrict__ x2, const float* __restrict__ y2,
                                                         const uint64_t* __restrict__ off, const double* __restrict__ Ein,
                                                         const double* __restrict__ tau2in, uint32_t n_pairs,
                                                         uint32_t* __restrict__ counts, uint8_t* __restrict__ masks) {
    const int lane = threadIdx.x & 63;
    const uint32_t pair = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= n_pairs) return;
    const uint64_t o = off[pair];
    const uint32_t n = (uint32_t)(off[pair + 1] - o);
    float e[9];
    {
        double n2 = 0.0, Ed[9];
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            Ed[c] = Ein[9 * (size_t)pair + c];
            n2 = fma(Ed[c], Ed[c], n2);
        }
        const double inv = 1.0 / sqrt(n2);
#pragma unroll
        for (int c = 0; c < 9; ++c) e[c] = (float)(Ed[c] * inv);
    }
    const float tau2 = (float)tau2in[pair];
    const float* px1 = x1 + o;
    const float* py1 = y1 + o;
    const float* px2 = x2 + o;
    const float* py2 = y2 + o;
    uint8_t* pm = masks ? masks + o : nullptr;
    uint32_t cnt = 0;
    // Head rows until the float4 streams are 16-byte aligned.  The head is derived from the actual addresses (the
    // caller's arrays may be slices); the vector body runs only if all four streams -- and the packed mask stores --
    // share that alignment, otherwise every row takes the scalar path.
    const uint32_t head_x = (uint32_t)((16u - ((uintptr_t)px1 & 15u)) & 15u) >> 2;
    const bool vec_ok = (((uintptr_t)px1 | (uintptr_t)py1 | (uintptr_t)px2 | (uintptr_t)py2) & 3u) == 0 &&
                        (((uintptr_t)px1 ^ (uintptr_t)py1) & 15u) == 0 && (((uintptr_t)px1 ^ (uintptr_t)px2) & 15u) == 0 &&
                        (((uintptr_t)px1 ^ (uintptr_t)py2) & 15u) == 0 && (!pm || (((uintptr_t)pm + head_x) & 3u) == 0);
    const uint32_t head = vec_ok ? min(n, head_x) : n;
    for (uint32_t i = lane; i < head; i += 64) {
        float r2, den;
        sampson_terms(e, px1[i], py1[i], px2[i], py2[i], r2, den);
        const bool in = r2 < tau2 * den;
        if (pm) pm[i] = in;
        cnt += in;
    }
    const uint32_t nv = (n - head) / 4;
    const float4* vx1 = reinterpret_cast<const float4*>(px1 + head);
    const float4* vy1 = reinterpret_cast<const float4*>(py1 + head);
    const float4* vx2 = reinterpret_cast<const float4*>(px2 + head);
    const float4* vy2 = reinterpret_cast<const float4*>(py2 + head);
#pragma clang loop unroll(disable)
    for (uint32_t i = lane; i < nv; i += 64) {
        const float4 a = vx1[i], b = vy1[i], c = vx2[i], d = vy2[i];
        float r2, den;
        uint32_t m = 0;
        sampson_terms(e, a.x, b.x, c.x, d.x, r2, den); m |= (r2 < tau2 * den) ? 1u : 0u;
        sampson_terms(e, a.y, b.y, c.y, d.y, r2, den); m |= (r2 < tau2 * den) ? 0x100u : 0u;
        sampson_terms(e, a.z, b.z, c.z, d.z, r2, den); m |= (r2 < tau2 * den) ? 0x10000u : 0u;
        sampson_terms(e, a.w, b.w, c.w, d.w, r2, den); m |= (r2 < tau2 * den) ? 0x1000000u : 0u;
        if (pm) *reinterpret_cast<uint32_t*>(pm + head + 4 * (size_t)i) = m;
        cnt += __popc(m);
    }
    const uint32_t done = head + 4 * nv;
    if (done + lane < n) {
        const uint32_t i = done + lane;
        float r2, den;
        sampson_terms(e, px1[i], py1[i], px2[i], py2[i], r2, den);
        const bool in = r2 < tau2 * den;
        if (pm) pm[i] = in;
        cnt += in;
    }
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) cnt += __shfl_xor(cnt, sft);
    if (lane == 0) counts[pair] = cnt;
}

// f64 AoS rows, the reference's own operation order (graph_traversal.h:107-115)
__global__ __launch_bounds__(256) void score_pose_f64_kernel(const double* __restrict__ corr, const uint64_t* __restrict__ off,
                                                             const double* __restrict__ Ein, const double* __restrict__ tau2in,
                                                             uint32_t n_pairs, uint32_t* __restrict__ counts,
                                                             uint8_t* __restrict__ masks) {
    const int lane = threadIdx.x & 63;
    const uint32_t pair = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= n_pairs) return;
    const uint64_t o = off[pair];
    const uint32_t n = (uint32_t)(off[pair + 1] - o);
    const double* E = Ein + 9 * (size_t)pair;
    const double e11 = E[0], e12 = E[1], e13 = E[2], e21 = E[3], e22 = E[4], e23 = E[5], e31 = E[6], e32 = E[7],
                 e33 = E[8];
    const double tau2 = tau2in[pair];
    const double4* rowsv = reinterpret_cast<const double4*>(corr + 4 * o);
    uint32_t cnt = 0;
    for (uint32_t i = lane; i < n; i += 64) {
        const double4 s = rowsv[i];
        const double x1 = s.x, y1 = s.y, x2 = s.z, y2 = s.w;
        const double rxc = e11 * x2 + e21 * y2 + e31;
        const double ryc = e12 * x2 + e22 * y2 + e32;
        const double rwc = e13 * x2 + e23 * y2 + e33;
        const double r = (x1 * rxc + y1 * ryc + rwc);
        const double rx = e11 * x1 + e12 * y1 + e13;
        const double ry = e21 * x1 + e22 * y1 + e23;
        const double sq = r * r / (rxc * rxc + ryc * ryc + rx * rx + ry * ry);
        const bool in = sq < tau2;
        if (masks) masks[o + i] = in;
        cnt += in;
    }
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) cnt += __shfl_xor(cnt, sft);
    if (lane == 0) counts[pair] = cnt;
}

// The accept / reject decision of InTraversalPoseTester::test for a whole wave, on the device: a chained pose stays a guess
// only if the screening launch counted at least `min_count` rows for it (pose_graph_builder.h:809: 5).
__global__ __launch_bounds__(256) void screen_guesses_kernel(const uint32_t* __restrict__ counts, uint32_t min_count,
                                                             uint8_t* __restrict__ has, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n && has[i] && counts[i] < min_count) has[i] = 0;
}

// One pair for the re-entrant host seam (pgi_score_pose_f64_host): the same arithmetic, one 256-thread workgroup walking
// the rows in chunks of 256 IN ORDER, so that InTraversalPoseTester::test's early exit (graph_traversal.h:221-225: return
// at the kMinimumInlierNumber-th inlier) stops the scan after the chunk in which that inlier falls.  out[0] = rows that
// pass (all rows when early == 0 or a mask is wanted, else the rows seen until the exit), out[1] = 1 iff early > 0 was
// reached.  prm: E[9], tau2.
__global__ __launch_bounds__(256) void score_pose_f64_one_kernel(const double* __restrict__ corr, uint32_t n,
                                                                 const double* __restrict__ prm, uint32_t early,
                                                                 uint32_t* __restrict__ out, uint8_t* __restrict__ mask) {
    __shared__ uint32_t total;
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid == 0) total = 0;
    __syncthreads();
    const double e11 = prm[0], e12 = prm[1], e13 = prm[2], e21 = prm[3], e22 = prm[4], e23 = prm[5], e31 = prm[6], e32 = prm[7],
                 e33 = prm[8], tau2 = prm[9];
    const double4* rowsv = reinterpret_cast<const double4*>(corr);
    const bool may_stop = early > 0 && mask == nullptr;
    uint32_t seen = 0;
    for (uint32_t base = 0; base < n; base += 256) {
        const uint32_t i = base + (uint32_t)tid;
        bool in = false;
        if (i < n) {
            const double4 s = rowsv[i];
            const double x1 = s.x, y1 = s.y, x2 = s.z, y2 = s.w;
            const double rxc = e11 * x2 + e21 * y2 + e31;
            const double ryc = e12 * x2 + e22 * y2 + e32;
            const double rwc = e13 * x2 + e23 * y2 + e33;
            const double r = (x1 * rxc + y1 * ryc + rwc);
            const double rx = e11 * x1 + e12 * y1 + e13;
            const double ry = e21 * x1 + e22 * y1 + e23;
            const double sq = r * r / (rxc * rxc + ryc * ryc + rx * rx + ry * ry);
            in = sq < tau2;
            if (mask) mask[i] = in;
        }
        const uint32_t c = (uint32_t)__popcll(__ballot(in));
        if (lane == 0 && c) atomicAdd(&total, c);
        __syncthreads();
        seen = total;
        __syncthreads();
        if (may_stop && seen >= early) break;  // workgroup-uniform
    }
    if (tid == 0) {
        out[0] = seen;
        out[1] = (early > 0 && seen >= early) ? 1u : 0u;
    }
}

// ------------------------------------------------------------------------------------------------
// K3: decomposition of a given E per pair (one workgroup per pair).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void decompose_kernel(const float* __restrict__ x1, const float* __restrict__ y1,
                                                        const float* __restrict__ x2, const float* __restrict__ y2,
                                                        const uint64_t* __restrict__ off, const double* __restrict__ Ein,
                                                        const uint8_t* __restrict__ masks, pgi_edge* __restrict__ edges) {
    __shared__ double Rt[21];
    __shared__ uint32_t votes[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t pair = blockIdx.x;
    const uint64_t o = off[pair];
    const uint32_t n = (uint32_t)(off[pair + 1] - o);
    if (tid < 64) decompose_wave(Ein + 9 * (size_t)pair, Rt, tid);
    if (tid == 0) votes[0] = votes[1] = votes[2] = votes[3] = 0;
    __syncthreads();
    double R1[9], R2[9], tt[3];
    for (int c = 0; c < 9; ++c) {
        R1[c] = Rt[c];
        R2[c] = Rt[9 + c];
    }
    tt[0] = Rt[18]; tt[1] = Rt[19]; tt[2] = Rt[20];
    uint32_t v0 = 0, v1 = 0, v2 = 0, v3 = 0;
    for (uint32_t base = 0; base < n; base += 256) {
        const uint32_t i = base + tid;
        const bool voter = (i < n) && (!masks || masks[o + i]);
        uint32_t b1 = 0, b2 = 0;
        if (voter) {
            const double X1[3] = {x1[o + i], y1[o + i], 1.0}, X2[3] = {x2[o + i], y2[o + i], 1.0};
            double x2t[3];
            cross3(X2, tt, x2t);
            b1 = cheirality_bits(R1, tt, X1, X2, x2t);
            b2 = cheirality_bits(R2, tt, X1, X2, x2t);
        }
        v0 += __popcll(__ballot(b1 & 1u));
        v1 += __popcll(__ballot(b1 & 2u));
        v2 += __popcll(__ballot(b2 & 1u));
        v3 += __popcll(__ballot(b2 & 2u));
    }
    if (lane == 0) {
        atomicAdd(&votes[0], v0);
        atomicAdd(&votes[1], v1);
        atomicAdd(&votes[2], v2);
        atomicAdd(&votes[3], v3);
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t best = 0;
        for (uint32_t c = 1; c < 4; ++c)
            if (votes[c] > votes[best]) best = c;
        pgi_edge* e = edges + pair;
        const double* Rb = (best >> 1) ? R2 : R1;
        for (int c = 0; c < 9; ++c) e->R[c] = Rb[c];
        for (int c = 0; c < 3; ++c) e->t[c] = (best & 1u) ? -tt[c] : tt[c];
        e->votes = votes[best];
        e->cand = best;
    }
}

// One pair for the re-entrant host seam (pgi_pose_from_essential_host): rows in the reference's layout (n x 4 f64, the
// cv::Mat of pose_utils.h:172), rounded to f32 like every row the engine estimates on, so the result equals
// decompose_kernel's on the converted batch bit for bit.  prm: E[9]; out: R[9], t[3] (doubles), then votes, cand (u32).
__global__ __launch_bounds__(256) void decompose_one_kernel(const double* __restrict__ corr, uint32_t n,
                                                            const double* __restrict__ prm, const uint8_t* __restrict__ mask,
                                                            double* __restrict__ out) {
    __shared__ double Rt[21];
    __shared__ uint32_t votes[4];
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid < 64) decompose_wave(prm, Rt, tid);
    if (tid == 0) votes[0] = votes[1] = votes[2] = votes[3] = 0;
    __syncthreads();
    double R1[9], R2[9], tt[3];
    for (int c = 0; c < 9; ++c) {
        R1[c] = Rt[c];
        R2[c] = Rt[9 + c];
    }
    tt[0] = Rt[18]; tt[1] = Rt[19]; tt[2] = Rt[20];
    const double4* rowsv = reinterpret_cast<const double4*>(corr);
    uint32_t v0 = 0, v1 = 0, v2 = 0, v3 = 0;
    for (uint32_t base = 0; base < n; base += 256) {
        const uint32_t i = base + tid;
        const bool voter = (i < n) && (!mask || mask[i]);
        uint32_t b1 = 0, b2 = 0;
        if (voter) {
            const double4 s = rowsv[i];
            const double X1[3] = {(double)(float)s.x, (double)(float)s.y, 1.0}, X2[3] = {(double)(float)s.z, (double)(float)s.w, 1.0};
            double x2t[3];
            cross3(X2, tt, x2t);
            b1 = cheirality_bits(R1, tt, X1, X2, x2t);
            b2 = cheirality_bits(R2, tt, X1, X2, x2t);
        }
        v0 += __popcll(__ballot(b1 & 1u));
        v1 += __popcll(__ballot(b1 & 2u));
        v2 += __popcll(__ballot(b2 & 1u));
        v3 += __popcll(__ballot(b2 & 2u));
    }
    if (lane == 0) {
        atomicAdd(&votes[0], v0);
        atomicAdd(&votes[1], v1);
        atomicAdd(&votes[2], v2);
        atomicAdd(&votes[3], v3);
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t best = 0;
        for (uint32_t c = 1; c < 4; ++c)
            if (votes[c] > votes[best]) best = c;  // first maximum wins (pose_utils.h:243-250)
        const double* Rb = (best >> 1) ? R2 : R1;
        for (int c = 0; c < 9; ++c) out[c] = Rb[c];
        for (int c = 0; c < 3; ++c) out[9 + c] = (best & 1u) ? -tt[c] : tt[c];
        uint32_t* tail = reinterpret_cast<uint32_t*>(out + 12);
        tail[0] = votes[best];
        tail[1] = best;
    }
}

// ------------------------------------------------------------------------------------------------
// K5: minimal solver on explicit samples; one 16-lane group per sample, 4 samples per wavefront.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void five_point_kernel(const float* __restrict__ pts, uint32_t n_samples,
                                                        float* __restrict__ models, uint32_t* __restrict__ counts,
                                                        double* __restrict__ dbgout) {
    __shared__ double scr[W_DOUBLES];
    const int lane = threadIdx.x, g = lane >> 4, s = lane & 15;
    const uint32_t smp_i = min(blockIdx.x * 4 + g, n_samples - 1);
    const bool active = blockIdx.x * 4 + g < n_samples;
    float4 smp[5];
    for (int k = 0; k < 5; ++k) {
        const float* p = pts + 20 * (size_t)smp_i + 4 * k;
        smp[k] = make_float4(p[0], p[1], p[2], p[3]);
    }
    float4 mine = smp[0];
    for (int k = 1; k < 5; ++k)
        if (s == k) mine = smp[k];
    const GroupScratch gs = group_scratch(scr, g);
    nullspace5_group(mine, s, g * 16, gs);
    float E32[9];
    bool valid;
    if (dbgout) {
        double* d = dbgout + (size_t)smp_i * PGI_DBG_DOUBLES;
        if (active && s < 9)
            for (int f = 0; f < 4; ++f) d[9 * f + s] = gs.basis[9 * f + s];
        BackendDbg dbg{d + 36, d + 236, d + 336, d + 347, d + 357};
        if (active && s < 10) d[347 + s] = 0.0;
        Prof prof;
        valid = backend_group<true, 3, true>(gs, s, g * 16, [&](int i) { return smp[i]; }, E32, &dbg, prof);
    } else {
        Prof prof;
        valid = backend_group<false, 3, true>(gs, s, g * 16, [&](int i) { return smp[i]; }, E32, nullptr, prof);
    }
    // compact the valid roots of each group in root order
    const uint64_t bal = __ballot(valid);
    const uint32_t gb = (uint32_t)((bal >> (16 * g)) & 0xFFFFu);
    const int slot = __popc(gb & ((1u << s) - 1u));
    if (active && valid)
        for (int c = 0; c < 9; ++c) models[(size_t)smp_i * 90 + 9 * slot + c] = E32[c];
    if (active && s == 0) counts[smp_i] = __popc(gb);
}

}  // namespace pgi

// =================================================================================================
// C ABI
// =================================================================================================
using namespace pgi;

namespace pgi {
std::string& last_error_ref() {
    static thread_local std::string e;
    return e;
}
}  // namespace pgi

// LDS of a K1 workgroup of `nw` wavefronts besides its staged rows; the sample stash (nw x 4 groups x 5 rows; 1280 B at four
// wavefronts) belongs to the variants whose rows may lie outside LDS (0 and 2)
static size_t k1_fixed_lds(int nw, bool stash) {
    const size_t shared = nw == 1 ? sizeof(WgShared<1>) : nw == 2 ? sizeof(WgShared<2>) : sizeof(WgShared<4>);
    return (size_t)nw * W_DOUBLES * 8 + shared + 64 + (stash ? (size_t)nw * 4 * 5 * sizeof(float4) : 0);
}
// the kernel instance for (rows variant, guess path, wavefronts per pair)
template <int NWv>
static void launch_k1(int lds_pts, bool guesses, dim3 grid, size_t lds, hipStream_t s, const K1Args& a) {
    const dim3 block(NWv * 64);
    if (lds_pts == 2) {
        if (guesses) hipLaunchKernelGGL((estimate_pose_kernel<2, true, NWv>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((estimate_pose_kernel<2, false, NWv>), grid, block, lds, s, a);
    } else if (lds_pts == 1) {
        if (guesses) hipLaunchKernelGGL((estimate_pose_kernel<1, true, NWv>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((estimate_pose_kernel<1, false, NWv>), grid, block, lds, s, a);
    } else if constexpr (NWv == kMaxNW) {  // rows from HBM/L2 only: pairs beyond every LDS class, which exist at four wavefronts only
        if (guesses) hipLaunchKernelGGL((estimate_pose_kernel<0, true, NWv>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((estimate_pose_kernel<0, false, NWv>), grid, block, lds, s, a);
    }
}
static void launch_k1(int nw, int lds_pts, bool guesses, dim3 grid, size_t lds, hipStream_t s, const K1Args& a) {
    if (nw == 1) launch_k1<1>(lds_pts, guesses, grid, lds, s, a);
    else if (nw == 2) launch_k1<2>(lds_pts, guesses, grid, lds, s, a);
    else launch_k1<4>(lds_pts, guesses, grid, lds, s, a);
}

extern "C" {

const char* pgi_last_error(void) { return pgi::last_error_ref().c_str(); }

int pgi_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void pgi_default_params(pgi_params* p) {
    p->confidence = 0.99;
    p->max_iters = 1000;
    p->round_size = 32;
    p->lo_iters = 2;
    p->min_inliers = 20;
    p->fixed_budget = 0;
    p->guess_quirk = 1;
    p->vote_all_rows = 0;
    p->guess_mode = 0;
    p->lo_linear_pct = 35;
    p->sampler = 0;
    p->lo_graph_cut = 0;
}

pgi_ctx* pgi_create(int device, const pgi_params* params) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
        pgi::last_error_ref() = "pgi_create: no HIP device (this library has no CPU fallback)";
        return nullptr;
    }
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess) device = 0;
    }
    if (device >= n) {
        pgi::last_error_ref() = "pgi_create: device index out of range";
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) {
        pgi::last_error_ref() = "pgi_create: hipSetDevice failed";
        return nullptr;
    }
    pgi_ctx* c = new (std::nothrow) pgi_ctx();
    if (!c) return nullptr;
    c->device = device;
    c->stream = nullptr;
    if (params) c->prm = *params; else pgi_default_params(&c->prm);
    int lds = 0;
    (void)hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device);
    c->max_lds = lds > 0 ? lds : 65536;
    int cus = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    if (cus > 0) c->resident_wgs = 3 * cus;
    c->n_cus = cus > 0 ? cus : 256;
    // K1 may use the whole LDS of a CU for staged rows
    (void)hipFuncSetAttribute((const void*)estimate_pose_kernel<1, true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, c->max_lds);
    (void)hipFuncSetAttribute((const void*)estimate_pose_kernel<1, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, c->max_lds);
    (void)hipFuncSetAttribute((const void*)estimate_pose_kernel<2, true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, c->max_lds);
    (void)hipFuncSetAttribute((const void*)estimate_pose_kernel<2, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, c->max_lds);
    if (const char* e = getenv("PGI_K1_NW")) c->k1_nw = atoi(e);
    if (const char* e = getenv("PGI_LDS_MIN_WGS")) c->lds_min_wgs = atoi(e);
    if (const char* e = getenv("PGI_HYBRID_ROWS")) c->hybrid_rows = atoi(e);
    if (const char* e = getenv("PGI_CLASS_OVERLAP")) c->class_overlap = atoi(e);
    if (const char* e = getenv("PGI_K1_PERSISTENT")) c->k1_persistent = atoi(e);
    if (const char* e = getenv("PGI_HOST_DIRECT")) c->host_direct = atoi(e);
    if (const char* e = getenv("PGI_MATCH_WAVES")) c->match_waves = atoi(e);
    if (const char* e = getenv("PGI_MATCH_SCREEN")) c->match_screen = atoi(e);
    {   // The first host <-> device copy of a process sets up the runtime's copy path (~20 ms, measured in the C++ driver's
        // first upload): pay it here, where contexts are made, not inside the first batch.
        // Small pageable copies and large page-locked ones take different routes inside the runtime: warm both.
        constexpr size_t kWarm = 1u << 20;
        void *dw = nullptr, *pw = nullptr;
        unsigned long long hw = 0;
        if (hipMalloc(&dw, kWarm) == hipSuccess) {
            (void)hipMemcpy(dw, &hw, sizeof hw, hipMemcpyHostToDevice);
            (void)hipMemcpy(&hw, dw, sizeof hw, hipMemcpyDeviceToHost);
            if (hipHostMalloc(&pw, kWarm, hipHostMallocDefault) == hipSuccess) {
                memset(pw, 0, kWarm);
                (void)hipMemcpy(dw, pw, kWarm, hipMemcpyHostToDevice);
                (void)hipMemcpy(pw, dw, kWarm, hipMemcpyDeviceToHost);
                (void)hipHostFree(pw);
            }
            (void)hipFree(dw);
        }
        (void)hipGetLastError();
    }
    return c;
}

void pgi_destroy(pgi_ctx* ctx) {
    if (!ctx) return;
    (void)pgi_comm_destroy(ctx);
    for (auto& S : ctx->pslot) {
        if (S.d) (void)hipFree(S.d);
        if (S.h) (void)hipHostFree(S.h);
        if (S.stream) (void)hipStreamDestroy(S.stream);
    }
    if (ctx->d_match_ws) (void)hipFree(ctx->d_match_ws);
    if (ctx->d_bucket) (void)hipFree(ctx->d_bucket);
    for (int k = 0; k < 4; ++k) {
        if (ctx->hslot[k].d) (void)hipFree(ctx->hslot[k].d);
        if (ctx->hslot[k].d_bucket) (void)hipFree(ctx->hslot[k].d_bucket);
        if (ctx->hslot[k].h_small) (void)hipHostFree(ctx->hslot[k].h_small);
        if (ctx->hslot[k].h_io) (void)hipHostFree(ctx->hslot[k].h_io);
        if (ctx->hslot[k].stream) (void)hipStreamDestroy(ctx->hslot[k].stream);
        if (ctx->hslot[k].in_done) (void)hipEventDestroy(ctx->hslot[k].in_done);
        if (ctx->hslot[k].k_done) (void)hipEventDestroy(ctx->hslot[k].k_done);
        if (ctx->hslot[k].out_done) (void)hipEventDestroy(ctx->hslot[k].out_done);
    }
    for (int k = 0; k < 2; ++k) {
        if (ctx->h_match_stage[k]) (void)hipHostFree(ctx->h_match_stage[k]);
        if (ctx->match_stage_ev[k]) (void)hipEventDestroy(ctx->match_stage_ev[k]);
    }
    for (int k = 0; k < 4; ++k) {
        if (ctx->class_stream[k]) (void)hipStreamDestroy(ctx->class_stream[k]);
        if (ctx->class_join[k]) (void)hipEventDestroy(ctx->class_join[k]);
    }
    if (ctx->class_fork) (void)hipEventDestroy(ctx->class_fork);
    if (ctx->d_direct) (void)hipFree(ctx->d_direct);
    if (ctx->d_mirror) (void)hipFree(ctx->d_mirror);
    if (ctx->copy_in) (void)hipStreamDestroy(ctx->copy_in);
    if (ctx->copy_out) (void)hipStreamDestroy(ctx->copy_out);
    delete ctx;
}

int pgi_set_stream(pgi_ctx* ctx, void* s) {
    if (!ctx) return fail(PGI_ERR_INVALID, "null ctx");
    std::lock_guard<std::mutex> lk(ctx->mu);
    hipStream_t ns = (hipStream_t)s;
    if (ns == ctx->stream) return PGI_SUCCESS;
    // The asynchronous entry points share context-owned scratch (matching workspace, bucket lists): work enqueued
    // on the new stream must not start before what is already queued on the old one has finished with it.
    HIP_TRY(hipSetDevice(ctx->device));
    hipEvent_t ev;
    HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev, ctx->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(ns, ev, 0);
    (void)hipEventDestroy(ev);  // released once the recorded work completes
    if (e != hipSuccess) return fail(PGI_ERR_DEVICE, std::string("pgi_set_stream: ") + hipGetErrorString(e));
    ctx->stream = ns;
    return PGI_SUCCESS;
}

int pgi_get_stream(pgi_ctx* ctx, void** out) {
    if (!ctx || !out) return fail(PGI_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(ctx->mu);
    *out = (void*)ctx->stream;
    return PGI_SUCCESS;
}

int pgi_get_device(pgi_ctx* ctx, int* out) {
    if (!ctx || !out) return fail(PGI_ERR_INVALID, "null argument");
    *out = ctx->device;
    return PGI_SUCCESS;
}

int pgi_set_params(pgi_ctx* ctx, const pgi_params* p) {
    if (!ctx || !p) return fail(PGI_ERR_INVALID, "null argument");
    if (!(p->confidence > 0.0 && p->confidence < 1.0)) return fail(PGI_ERR_INVALID, "confidence must be in (0,1)");
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->prm = *p;
    return PGI_SUCCESS;
}

int pgi_get_params(pgi_ctx* ctx, pgi_params* p) {
    if (!ctx || !p) return fail(PGI_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(ctx->mu);
    *p = ctx->prm;
    return PGI_SUCCESS;
}

int pgi_synchronize(pgi_ctx* ctx) {
    if (!ctx) return fail(PGI_ERR_INVALID, "null ctx");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PGI_SUCCESS;
}

// profiling builds only (-DPGI_PROFILE): device buffer of kProfSlots 64-bit cycle counters
int pgi_internal_set_profile_buffer(pgi_ctx* ctx, unsigned long long* d_buf) {
    if (!ctx) return PGI_ERR_INVALID;
    ctx->d_prof = d_buf;
    return PGI_SUCCESS;
}

// enqueues K1 for a device-resident batch on `stream`; `bucket` is the caller's scratch for the size-bucket lists
// largest 64-multiple of rows that still lets wgs_per_cu workgroups share a CU
static uint32_t k1_rows_cap(const pgi_ctx* ctx, int wgs_per_cu, size_t fixed_bytes) {
    const size_t budget = (size_t)ctx->max_lds / (size_t)wgs_per_cu;
    return budget > fixed_bytes ? (uint32_t)(((budget - fixed_bytes) / 16) & ~(size_t)63) : 0u;
}

// `src`: four page-locked host arrays (device-visible addresses) the rows are consumed from in place; b->d_x1..d_y2 are
// then the device mirror for rows that do not fit in LDS.
static int launch_estimate(pgi_ctx* ctx, const pgi_params& prm, const pgi_batch* b, pgi_edge* d_edges, uint8_t* d_masks,
                           hipStream_t stream, uint32_t** bucket, size_t* bucket_cap, const float* const* src = nullptr,
                           uint64_t rows_hint = 0, uint32_t* gc_ext = nullptr) {
    if (!ctx || !b || !d_edges || !d_masks) return fail(PGI_ERR_INVALID, "null argument");
    if (b->n_pairs == 0) return PGI_SUCCESS;
    if (!b->d_x1 || !b->d_y1 || !b->d_x2 || !b->d_y2 || !b->d_offsets || !b->d_thr)
        return fail(PGI_ERR_INVALID, "batch pointers missing");
    K1Args a;
    a.x1 = b->d_x1; a.y1 = b->d_y1; a.x2 = b->d_x2; a.y2 = b->d_y2;
    a.off = b->d_offsets; a.thr = b->d_thr; a.guess = b->d_guess_Rt; a.has_guess = b->d_has_guess;
    a.edges = d_edges; a.masks = d_masks; a.n_pairs = b->n_pairs;
    a.pair_id_base = b->pair_id_base; a.seed = b->seed; a.prm = prm;
    a.prof = ctx->d_prof;
    a.pair_list = nullptr;
    a.pair_count = nullptr;
    a.pair_head = nullptr;
    a.src_x1 = src ? src[0] : nullptr; a.src_y1 = src ? src[1] : nullptr;
    a.src_x2 = src ? src[2] : nullptr; a.src_y2 = src ? src[3] : nullptr;
    a.gc_prev = nullptr; a.gc_msg = nullptr; a.gc_lds_off = 0;
    HIP_TRY(hipSetDevice(ctx->device));
    // Graph-cut local optimisation (prm.lo_graph_cut): two scratch words per row of the batch, indexed like the coordinates.
    // The caller brings them (gc_ext: the single-pair entry) or they follow the size-bucket lists in the caller's scratch; a
    // device-resident batch does not tell its row count, so it is read back (one 8-byte copy and a stream synchronisation per
    // call: the price of the optional mode).
    const bool gc_on = prm.lo_graph_cut != 0u;
    uint64_t gc_rows = 0;
    if (gc_on) {
        if (src) return fail(PGI_ERR_INVALID, "lo_graph_cut: not with rows consumed in place from page-locked host memory (PGI_HOST_DIRECT=0)");
        gc_rows = rows_hint;
        if (!gc_rows) {
            uint64_t last = 0;
            HIP_TRY(hipMemcpyAsync(&last, b->d_offsets + b->n_pairs, sizeof last, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            gc_rows = last;
        }
        if (!gc_ext && !bucket) return fail(PGI_ERR_INVALID, "launch_estimate: lo_graph_cut without scratch");
    }
    const size_t gc_words = gc_on && !gc_ext ? (size_t)2 * gc_rows + 32 : 0;
    // Wavefronts per pair for this call (see kMaxNW above).  Measured on the dense V = 5000 scene's rows (scripts/k1_dense_bench.py,
    // K1D_MAXPAIRS): two wavefronts per pair tie with four at 12 000 pairs and win above (-11 % at 24 000, -14 % at 48 000), one wins from about 10^5 pairs on (its
    // steady rate is 32 % above four's, but a launch winds down for 9 ms: the pairs that run to max_iters are one wavefront's work).
    int nw = ctx->k1_nw == 1 || ctx->k1_nw == 2 || ctx->k1_nw == 4 ? ctx->k1_nw : b->n_pairs >= kNw1Pairs ? 1 : b->n_pairs >= kNw2Pairs ? 2 : 4;
    const size_t gc_lds = gc_on ? (size_t)kGcCells * 4 : 0;  // the chain builder's cell table, behind everything else
    const size_t fixed = k1_fixed_lds(nw, false) + gc_lds, fixed_stash = k1_fixed_lds(nw, true) + gc_lds;
    const bool guesses = b->d_guess_Rt != nullptr && b->d_has_guess != nullptr;  // selects the kernel variant with the guess path
    auto rows_cap_of = [&](int wgs_per_cu, size_t fixed_bytes) { return k1_rows_cap(ctx, wgs_per_cu, fixed_bytes); };
    auto rows_cap = [&](int wgs_per_cu) { return rows_cap_of(wgs_per_cu, fixed); };
    // Occupancy levels: 16 / 12 / 8 / 4 wavefronts per CU = 4 / 3 / 2 / 1 workgroups of four wavefronts.  Below four wavefronts
    // per pair there are two classes only -- rows whole in LDS at the top occupancy, and everything larger HYBRID (the first
    // rows in LDS, the tail from HBM/L2) at the same occupancy.
    const int kLevelWgs[4] = {16 / nw, std::max(1, 12 / nw), std::max(1, 8 / nw), std::max(1, 4 / nw)};
    const uint32_t cap4 = rows_cap(kLevelWgs[0]);
    const uint32_t cap3 = nw < 4 ? 0x7FFFFFC0u : rows_cap(kLevelWgs[1]), cap2 = nw < 4 ? 0x7FFFFFC0u : rows_cap(kLevelWgs[2]),
                   cap1 = nw < 4 ? 0x7FFFFFC0u : rows_cap(kLevelWgs[3]);
    bool hybrid = false;
    int class_wgs = kLevelWgs[0];  // workgroups per CU of the class being launched (sizes a persistent grid)
    hipStream_t ls = stream;  // the stream the next launch goes to (a class's side stream when classes overlap)
    // grid of a launch: one workgroup per pair, or -- persistent class launch -- as many as stay resident at once
    auto grid_of = [&]() {
        if (!a.pair_head) return dim3(b->n_pairs);
        return dim3(std::min<uint32_t>(b->n_pairs, (uint32_t)(ctx->n_cus * class_wgs)));
    };
    auto launch_lds = [&](uint32_t cap_rows) {
        a.pts_cap = cap_rows;
        const size_t lds = (size_t)cap_rows * 16 + (hybrid ? fixed_stash : fixed);
        a.gc_lds_off = (uint32_t)(lds - gc_lds);
        launch_k1(nw, hybrid ? 2 : 1, guesses, grid_of(), lds, ls, a);  // hybrid: first cap_rows rows in LDS, the tail from HBM/L2
    };
    auto launch_global = [&]() {  // rows stay in HBM/L2 (pairs beyond the LDS capacity)
        a.pts_cap = 0;
        a.gc_lds_off = (uint32_t)(fixed_stash - gc_lds);
        // (the rows-from-memory variant exists at four wavefronts per pair only; below that the hybrid variant with an empty LDS
        //  part is the same thing -- launch_k1 would otherwise have nothing to launch and the pairs' records would stay unwritten)
        launch_k1(nw, nw < 4 ? 2 : 0, guesses, grid_of(), fixed_stash, ls, a);
    };
    const uint32_t cap = (b->max_corr + 63u) & ~63u;
    // The kernel is compiled for 128 VGPRs (four wavefronts per SIMD), so LDS decides the occupancy: pairs of up to
    // cap4 rows run four workgroups per CU (+10-13 % over three), cap3 three, cap2 two.  Rows are staged in LDS only
    // while at least `lds_min_wgs` workgroups still fit per CU; bigger pairs read their rows from HBM/L2 at full
    // occupancy instead (measured: faster than LDS at one workgroup per CU, scripts/lds_capacity_probe.py).
    // PGI_LDS_MIN_WGS overrides the default for experiments.
    const uint32_t lds_cap = ctx->lds_min_wgs >= 4 ? cap4 : ctx->lds_min_wgs == 3 ? cap3 : ctx->lds_min_wgs == 2 ? cap2 : cap1;
    const bool ragged = !(cap <= cap4 || b->n_pairs < 64);
    const size_t list_words = ragged ? (size_t)5 * b->n_pairs + 16 : 0;  // counts[8] | heads[8] | five lists
    if (list_words + gc_words) {
        if (!bucket) return fail(PGI_ERR_INVALID, "launch_estimate: ragged batch without bucket scratch");
        const size_t need = (list_words + gc_words) * sizeof(uint32_t);
        if (need > *bucket_cap) {
            HIP_TRY(hipStreamSynchronize(stream));
            if (*bucket) (void)hipFree(*bucket);
            *bucket = nullptr;
            *bucket_cap = 0;
            HIP_TRY(hipMalloc((void**)bucket, need + (gc_words ? need / 4 : 0)));
            *bucket_cap = need + (gc_words ? need / 4 : 0);
        }
    }
    if (gc_on) {
        uint32_t* w = gc_ext ? gc_ext : *bucket + ((list_words + 15) & ~(size_t)15);
        a.gc_prev = w;
        a.gc_msg = w + gc_rows;
    }
    if (!ragged) {  // every pair already gets the top occupancy (or the batch is tiny): one launch
        if (nw < 4 && cap > cap4) {  // (a tiny batch of large pairs in the builds that know two classes: the hybrid one)
            hybrid = true;
            launch_lds(rows_cap_of(kLevelWgs[0], fixed_stash));
            hybrid = false;
        } else if (cap <= lds_cap) launch_lds(cap); else launch_global();
    } else {  // bucket by row count on the device, one launch per occupancy class
        uint32_t* counts = *bucket;
        uint32_t* heads = *bucket + 8;
        uint32_t* lists = *bucket + 16;
        HIP_TRY(hipMemsetAsync(counts, 0, 16 * sizeof(uint32_t), stream));
        hipLaunchKernelGGL(bucket_pairs_kernel, dim3((b->n_pairs + 255) / 256), dim3(256), 0, stream, b->d_offsets,
                           b->n_pairs, cap4, cap3, cap2, cap1, lists, counts);
        const uint32_t caps[4] = {cap4, cap3, cap2, cap1};
        int n_classes = 1;
        while (n_classes < 5 && cap > caps[n_classes - 1]) ++n_classes;  // classes a pair of this batch can fall into
        // The classes are independent launches (disjoint pairs, disjoint outputs).  With more than one, all but the last
        // go to side streams that wait for the bucket lists (fork event) and are joined back into the caller's stream:
        // the workgroups of one class fill the CUs another class's tail leaves idle, and an empty class costs nothing.
        bool overlap = ctx->class_overlap && n_classes > 1;
        if (overlap) {
            if (!ctx->class_fork && hipEventCreateWithFlags(&ctx->class_fork, hipEventDisableTiming) != hipSuccess) overlap = false;
            for (int k = 0; k < n_classes - 1 && overlap; ++k) {
                if (!ctx->class_stream[k] && hipStreamCreateWithFlags(&ctx->class_stream[k], hipStreamNonBlocking) != hipSuccess) overlap = false;
                if (overlap && !ctx->class_join[k] && hipEventCreateWithFlags(&ctx->class_join[k], hipEventDisableTiming) != hipSuccess) overlap = false;
            }
            if (!overlap) (void)hipGetLastError();
        }
        if (overlap) HIP_TRY(hipEventRecord(ctx->class_fork, stream));
        // From here on side streams may hold launched kernels: an error must not return before they are joined back into
        // the caller's stream, or the caller could reuse / free d_edges, d_masks and the bucket lists under them.
        hipError_t first_err = hipSuccess;
        auto keep = [&](hipError_t e) {
            if (e != hipSuccess && first_err == hipSuccess) first_err = e;
            return e == hipSuccess;
        };
        int joined = 0;  // side classes whose join event has been recorded
        for (int k = 0; k < n_classes && first_err == hipSuccess; ++k) {
            a.pair_list = lists + (size_t)k * b->n_pairs;
            a.pair_count = counts + k;
#ifdef PGI_K1_LOOP_GUESS
            a.pair_head = ctx->k1_persistent ? heads + k : nullptr;
#else
            a.pair_head = ctx->k1_persistent && !guesses ? heads + k : nullptr;  // (the guess variants have no loop)
#endif
            const bool hybrid_class = k == 1 && (ctx->hybrid_rows || nw < 4);  // (below four wavefronts per pair there are two classes only)
            class_wgs = k == 4 || (k < 4 && caps[k] > lds_cap) ? kLevelWgs[0] : hybrid_class ? kLevelWgs[0] : kLevelWgs[k];
            const bool side = overlap && k < n_classes - 1;  // the last (largest-row) class stays on the caller's stream
            ls = side ? ctx->class_stream[k] : stream;
            if (side && !keep(hipStreamWaitEvent(ls, ctx->class_fork, 0))) break;
            // class 1 (cap4 < rows <= cap3): with `hybrid_rows` the first cap4 rows stay in LDS and the tail is read
            // from HBM/L2, which keeps four workgroups per CU instead of three
            hybrid = hybrid_class;
            if (hybrid) launch_lds(rows_cap_of(kLevelWgs[0], fixed_stash));  // 1280 rows in LDS next to the sample stash (NW = 4)
            else if (k < 4 && caps[k] <= lds_cap) launch_lds(std::min(caps[k], cap)); else launch_global();
            hybrid = false;
            keep(hipGetLastError());
            if (side) {
                if (keep(hipEventRecord(ctx->class_join[k], ls))) joined = k + 1;
                else (void)hipStreamSynchronize(ls);  // no event to wait on: drain the side stream here
            }
        }
        ls = stream;
        for (int k = 0; k < joined; ++k)
            if (!keep(hipStreamWaitEvent(stream, ctx->class_join[k], 0))) (void)hipStreamSynchronize(ctx->class_stream[k]);
        if (first_err != hipSuccess) return fail(PGI_ERR_DEVICE, hipGetErrorString(first_err));
    }
    HIP_TRY(hipGetLastError());
    return PGI_SUCCESS;
}

int pgi_estimate_pose_batch(pgi_ctx* ctx, const pgi_batch* b, pgi_edge* d_edges, uint8_t* d_masks) {
    if (!ctx) return fail(PGI_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(ctx->mu);  // the size-bucket scratch belongs to the context
    return launch_estimate(ctx, ctx->prm, b, d_edges, d_masks, ctx->stream, &ctx->d_bucket, &ctx->bucket_bytes);
}

// Page-locked inputs AND results: K1 works on the caller's buffers IN PLACE over PCIe.  The kernel reads every row exactly
// once (its staging loop) and writes every result once, so a copy to HBM first only adds a pipeline with a head (the first
// chunk's copy) and a tail (the last chunk's kernel and its results) around transfers the kernel can issue itself: a kernel
// reading host memory moves 53 GB/s here against 57 GB/s for hipMemcpyAsync (scripts/probes/zero_copy_probe.hip), overlapped
// with the fits of the other resident workgroups.  Only the per-pair arrays are uploaded; pairs too large for LDS copy
// their tail rows into a device mirror while staging.  One launch sequence, nothing to bring back.
// BASELINE config 2 host to host: 7.7-8.0 ms against 8.1-8.3 ms for the copy pipeline below (scripts/host_direct_bench.py).
static int estimate_host_direct(pgi_ctx* ctx, const float* const* src, const uint64_t* h_offsets, const double* h_thr,
                                const double* h_guess_Rt, const uint8_t* h_has_guess, uint32_t n_pairs, uint64_t pair_id_base,
                                uint64_t seed, pgi_edge* map_edges, uint8_t* map_masks) {
    const bool guesses = h_guess_Rt != nullptr;
    const uint64_t base = h_offsets[0], rows = h_offsets[n_pairs] - base;
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t o_off = 0, o_thr = o_off + up(((size_t)n_pairs + 1) * 8), o_guess = o_thr + up((size_t)n_pairs * 8),
                 o_has = o_guess + up((size_t)n_pairs * 96), total = o_has + up(n_pairs);
    pgi_ctx::HostSlot& S = ctx->hslot[0];
    if (!S.stream) HIP_TRY(hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking));
    if (total > ctx->direct_bytes) {
        HIP_TRY(hipStreamSynchronize(S.stream));
        if (ctx->d_direct) (void)hipFree(ctx->d_direct);
        ctx->d_direct = nullptr; ctx->direct_bytes = 0;
        HIP_TRY(hipMalloc(&ctx->d_direct, total + total / 4));
        ctx->direct_bytes = total + total / 4;
    }
    if (total > S.h_small_bytes) {
        if (S.h_small) (void)hipHostFree(S.h_small);
        S.h_small = nullptr; S.h_small_bytes = 0;
        HIP_TRY(hipHostMalloc(&S.h_small, total + total / 4, hipHostMallocDefault));
        S.h_small_bytes = total + total / 4;
    }
    char* hs = (char*)S.h_small;
    char* d = (char*)ctx->d_direct;
    uint64_t* ol = reinterpret_cast<uint64_t*>(hs + o_off);
    uint32_t max_corr = 0;
    for (uint32_t k = 0; k <= n_pairs; ++k) ol[k] = h_offsets[k] - base;
    for (uint32_t k = 0; k < n_pairs; ++k) max_corr = std::max(max_corr, (uint32_t)(ol[k + 1] - ol[k]));
    memcpy(hs + o_thr, h_thr, (size_t)n_pairs * 8);
    if (guesses) {
        memcpy(hs + o_guess, h_guess_Rt, (size_t)n_pairs * 96);
        memcpy(hs + o_has, h_has_guess, n_pairs);
    }
    HIP_TRY(hipMemcpyAsync(d, hs, guesses ? o_has + n_pairs : o_thr + (size_t)n_pairs * 8, hipMemcpyHostToDevice, S.stream));
    // the rows of pairs beyond the four-workgroup LDS capacity live partly (or wholly) outside LDS: device mirror
    const bool need_mirror = ((max_corr + 63u) & ~63u) > k1_rows_cap(ctx, 4, k1_fixed_lds(4, false));
    const float* mirror[4] = {src[0], src[1], src[2], src[3]};
    if (need_mirror) {
        const size_t one = up((size_t)rows * 4);
        if (4 * one > ctx->mirror_bytes) {
            HIP_TRY(hipStreamSynchronize(S.stream));
            if (ctx->d_mirror) (void)hipFree(ctx->d_mirror);
            ctx->d_mirror = nullptr; ctx->mirror_bytes = 0;
            HIP_TRY(hipMalloc(&ctx->d_mirror, 4 * one));
            ctx->mirror_bytes = 4 * one;
        }
        for (int k = 0; k < 4; ++k) mirror[k] = reinterpret_cast<const float*>((char*)ctx->d_mirror + (size_t)k * one);
    }
    pgi_batch b{};
    b.d_x1 = mirror[0]; b.d_y1 = mirror[1]; b.d_x2 = mirror[2]; b.d_y2 = mirror[3];
    b.d_offsets = reinterpret_cast<const uint64_t*>(d + o_off);
    b.d_thr = reinterpret_cast<const double*>(d + o_thr);
    b.d_guess_Rt = guesses ? reinterpret_cast<const double*>(d + o_guess) : nullptr;
    b.d_has_guess = guesses ? reinterpret_cast<const uint8_t*>(d + o_has) : nullptr;
    b.n_pairs = n_pairs; b.max_corr = max_corr; b.pair_id_base = pair_id_base; b.seed = seed;
    const int rc = launch_estimate(ctx, ctx->prm, &b, map_edges, map_masks, S.stream, &S.d_bucket, &S.bucket_bytes, src);
    if (rc < 0) return rc;
    HIP_TRY(hipStreamSynchronize(S.stream));  // the results are in the caller's buffers
    return PGI_SUCCESS;
}

// Host buffers in, host buffers out: the batch is cut into chunks that travel through two device slots on two
// streams, so the H2D copy of chunk c+1 and the D2H copy of chunk c-1 overlap the kernel of chunk c (PCIe-inclusive
// throughput ~ max(copy, compute) instead of their sum).  Pair ids are global, so the result equals the one-launch one.
int pgi_estimate_pose_batch_host(pgi_ctx* ctx, const float* h_x1, const float* h_y1, const float* h_x2, const float* h_y2,
                                 const uint64_t* h_offsets, const double* h_thr, const double* h_guess_Rt,
                                 const uint8_t* h_has_guess, uint32_t n_pairs, uint64_t pair_id_base, uint64_t seed,
                                 pgi_edge* h_edges, uint8_t* h_masks) {
    if (!ctx) return fail(PGI_ERR_INVALID, "null ctx");
    if (n_pairs == 0) return PGI_SUCCESS;
    if (!h_x1 || !h_y1 || !h_x2 || !h_y2 || !h_offsets || !h_thr || !h_edges || !h_masks)
        return fail(PGI_ERR_INVALID, "null argument");
    if ((h_guess_Rt == nullptr) != (h_has_guess == nullptr)) return fail(PGI_ERR_INVALID, "guesses and flags go together");
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    const bool guesses = h_guess_Rt != nullptr;
    // Are the caller's buffers page-locked (pgi_host_register / hipHostMalloc)?  Then copies are true asynchronous DMA.
    auto page_locked = [](const void* p) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, p) != hipSuccess) {
            (void)hipGetLastError();  // unregistered host memory: not an error for us
            return false;
        }
        return at.type == hipMemoryTypeHost;
    };
    // A range must be page-locked as a whole or not at all: the runtime rejects a copy that leaves a registered range
    // ("invalid argument"), and a kernel working in place would fault past its end.  Checked up front, on both ends.
    {
        const uint64_t rows_all = h_offsets[n_pairs] - h_offsets[0];
        const void* lo[6] = {h_x1, h_y1, h_x2, h_y2, h_edges, h_masks};
        const size_t len[6] = {(size_t)rows_all * 4, (size_t)rows_all * 4, (size_t)rows_all * 4, (size_t)rows_all * 4,
                               (size_t)n_pairs * sizeof(pgi_edge), (size_t)rows_all};
        for (int k = 0; k < 6; ++k)
            if (len[k] && page_locked(lo[k]) != page_locked(static_cast<const char*>(lo[k]) + len[k] - 1))
                return fail(PGI_ERR_INVALID, "pgi_estimate_pose_batch_host: a buffer is page-locked only in part (register the whole "
                                             "range the batch uses, or none of it)");
    }
    const bool pin_in = page_locked(h_x1) && page_locked(h_y1) && page_locked(h_x2) && page_locked(h_y2);
    // (results always return on the chunk's kernel stream, two launches behind the front: measured faster than an
    // immediate copy on a dedicated stream even for page-locked result buffers, 8.6 ms vs 9.0-13 ms)
    const bool pinned = pin_in;
    // Pageable buffers never meet the device: rows and results pass through page-locked blocks the slots own (hipHostMalloc).
    // Handing a pageable pointer to hipMemcpyAsync makes the runtime page-lock the caller's pages on the fly; in a long process
    // (heap pages shared with other allocations, ranges registered and unregistered shortly before) that ended, now and then,
    // in "Memory access fault by GPU" on a heap address (round 5, tests/test_gpu_parity.py after the partial-registration cases).
    const bool pin_out = page_locked(h_edges) && page_locked(h_masks);
    if (pinned && ctx->host_direct && !ctx->prm.lo_graph_cut && page_locked(h_edges) && page_locked(h_masks)) {  // work on the caller's buffers in place
        void* hp[6] = {const_cast<float*>(h_x1), const_cast<float*>(h_y1), const_cast<float*>(h_x2), const_cast<float*>(h_y2), h_edges, h_masks};
        void* dp[6] = {};
        // The kernel will touch the WHOLE of every range: both ends must be mapped, and contiguously (two separate
        // registrations that happen to cover both ends do not make one device range) -- else the copy pipeline runs.
        const uint64_t rows_total = h_offsets[n_pairs] - h_offsets[0];
        const size_t span[6] = {(size_t)rows_total * 4, (size_t)rows_total * 4, (size_t)rows_total * 4, (size_t)rows_total * 4,
                                (size_t)n_pairs * sizeof(pgi_edge), (size_t)rows_total};
        bool mapped = true;
        for (int k = 0; k < 6 && mapped; ++k) {
            void* last = nullptr;
            char* last_byte = static_cast<char*>(hp[k]) + (span[k] ? span[k] - 1 : 0);  // (the arrays start at the batch's first row)
            if (hipHostGetDevicePointer(&dp[k], hp[k], 0) != hipSuccess || !dp[k] || !page_locked(last_byte) ||
                hipHostGetDevicePointer(&last, last_byte, 0) != hipSuccess || !last ||
                static_cast<char*>(last) - static_cast<char*>(dp[k]) != last_byte - static_cast<char*>(hp[k])) {
                (void)hipGetLastError();  // page-locked but not (wholly, contiguously) mapped into the device's address space: copy instead
                mapped = false;
            }
        }
        if (mapped) {
            const float* src[4] = {static_cast<const float*>(dp[0]), static_cast<const float*>(dp[1]), static_cast<const float*>(dp[2]),
                                   static_cast<const float*>(dp[3])};
            return estimate_host_direct(ctx, src, h_offsets, h_thr, h_guess_Rt, h_has_guess, n_pairs, pair_id_base, seed,
                                        static_cast<pgi_edge*>(dp[4]), static_cast<uint8_t*>(dp[5]));
        }
    }
    // Chunks are multiples of the number of resident workgroups (no chunk ends in a mostly empty last wave of
    // workgroups).  Page-locked: PCIe (~57 GB/s) is only ~1.3x faster than K1 consumes rows, so equal, small chunks keep
    // the copy front just ahead of the kernels and expose only the first chunk's copy.  Pageable: the runtime stages
    // every copy on the calling thread, so fewer, larger chunks (a small first one) amortise that better.
    const uint32_t quantum = (uint32_t)std::max(64, ctx->resident_wgs);
    uint32_t first_q = pinned ? 2 : 1, rest_q = pinned ? 2 : 3;
    bool chunks_from_env = false;
    if (const char* e = getenv("PGI_HOST_CHUNKS")) {  // "first,rest" in quanta (experiments)
        unsigned a = 0, b2 = 0;
        if (sscanf(e, "%u,%u", &a, &b2) == 2 && a && b2) { first_q = a; rest_q = b2; chunks_from_env = true; }
    }
    std::vector<uint32_t> ramp;  // PGI_HOST_CHUNK_PAIRS="a,b,c": pairs of the first chunks, the last value repeats (experiments)
    if (const char* e = getenv("PGI_HOST_CHUNK_PAIRS")) {
        for (const char* q = e; *q;) {
            char* end = nullptr;
            const unsigned long v = strtoul(q, &end, 10);
            if (end == q) break;
            if (v) ramp.push_back((uint32_t)v);
            q = (*end == ',') ? end + 1 : end;
        }
    }
    std::vector<uint32_t> cuts(1, 0u);
    for (uint32_t p = 0; p < n_pairs;) {
        uint32_t want = (cuts.size() == 1 ? first_q : rest_q) * quantum;
        // page-locked inputs: short first chunks (the kernels start sooner), then 1.5 quanta -- 8.1-8.2 ms against 8.25-8.6 ms
        // for equal chunks of two quanta on BASELINE config 2 (scripts/host_direct_bench.py ramp ...)
        if (pinned && !chunks_from_env) want = cuts.size() == 1 ? quantum / 2 : cuts.size() == 2 ? quantum : quantum + quantum / 2;
        if (!ramp.empty()) want = ramp[std::min(cuts.size() - 1, ramp.size() - 1)];
        uint32_t q = p;
        const uint64_t r0 = h_offsets[p];
        while (q < n_pairs && q - p < want && (q == p || h_offsets[q + 1] - r0 <= 8000000ull)) ++q;
        if (n_pairs - q < quantum / 2) q = n_pairs;  // no tiny last chunk
        cuts.push_back(q);
        p = q;
    }
    // the kernel of the last chunk is pure tail (nothing left to overlap it with): keep that chunk to one quantum (shorter
    // ramp-down chunks were slower: a chunk's kernel lasts as long as its slowest pair, whatever its size)
    if (pinned && cuts.size() >= 3) {
        const uint32_t a0 = cuts[cuts.size() - 2], b0 = cuts.back();
        if (b0 - a0 >= quantum + quantum / 2) cuts.insert(cuts.end() - 1, b0 - quantum);
    }
    const size_t n_chunks = cuts.size() - 1;
    struct Lay { size_t x1, y1, x2, y2, off, thr, guess, has, edges, masks, total; };
    auto layout = [&](uint64_t rows, uint32_t pairs) {
        auto up = [](size_t v) { return (v + 255) / 256 * 256; };
        Lay L;
        size_t o = 0;
        L.x1 = o; o += up(rows * 4); L.y1 = o; o += up(rows * 4); L.x2 = o; o += up(rows * 4); L.y2 = o; o += up(rows * 4);
        L.off = o; o += up(((size_t)pairs + 1) * 8); L.thr = o; o += up((size_t)pairs * 8);
        L.guess = o; o += up((size_t)pairs * 96); L.has = o; o += up(pairs);
        L.edges = o; o += up((size_t)pairs * sizeof(pgi_edge)); L.masks = o; o += up(rows);
        L.total = o;
        return L;
    };
    // Streams: input copies travel on ONE dedicated HIGH-PRIORITY stream, kernels (and the result copies behind them)
    // alternate between TWO streams -- consecutive chunks overlap, and that is all the overlap there is to have.  A copy
    // that shares a stream -- or a priority -- with K1 is starved by the ten thousand workgroups K1 keeps queued
    // (measured: page-locked buffers were SLOWER than pageable ones that way); events carry the dependencies.  Three
    // streams, not seven: the runtime multiplexes streams onto a handful of hardware queues, and with one kernel stream
    // per device slot two chunks that should overlap could end up on the same queue (20 % slower in a process that had
    // already created other streams, e.g. bench.py after its resident runs; GPU_MAX_HW_QUEUES=12 made it go away).
    constexpr size_t kSlots = 4;
    if (!ctx->copy_in) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);  // hi = numerically lowest = highest priority
        HIP_TRY(hipStreamCreateWithPriority(&ctx->copy_in, hipStreamNonBlocking, hi));
    }
    for (size_t k = 0; k < kSlots; ++k) {
        pgi_ctx::HostSlot& S = ctx->hslot[k];
        if (k < 2 && !S.stream) HIP_TRY(hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking));
        if (!S.in_done) {
            HIP_TRY(hipEventCreateWithFlags(&S.in_done, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&S.k_done, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&S.out_done, hipEventDisableTiming));
        }
        S.used = false;
        S.pending = -1;
    }
    Lay lay[kSlots];
    auto fetch = [&](size_t c) -> int {  // edge records and masks of chunk c back to the caller's buffers
        pgi_ctx::HostSlot& S = ctx->hslot[c % kSlots];
        const Lay& L = lay[c % kSlots];
        const uint32_t p0 = cuts[c], np = cuts[c + 1] - p0;
        const uint64_t r0 = h_offsets[p0] - h_offsets[0], rows = h_offsets[p0 + np] - h_offsets[p0];
        char* d = (char*)S.d;
        hipStream_t out = ctx->hslot[c & 1].stream;  // behind the chunk's own kernel
        HIP_TRY(hipStreamWaitEvent(out, S.k_done, 0));
        if (pin_out) {
            HIP_TRY(hipMemcpyAsync(h_edges + p0, d + L.edges, (size_t)np * sizeof(pgi_edge), hipMemcpyDeviceToHost, out));
            if (rows) HIP_TRY(hipMemcpyAsync(h_masks + r0, d + L.masks, rows, hipMemcpyDeviceToHost, out));
        } else {  // edge records and masks are neighbours on the device: one copy into the slot's block, handed over by drain()
            HIP_TRY(hipMemcpyAsync((char*)S.h_io + (L.edges - L.x1), d + L.edges, (L.masks + rows) - L.edges, hipMemcpyDeviceToHost, out));
            S.pending = (long)c;
        }
        HIP_TRY(hipEventRecord(S.out_done, out));
        return PGI_SUCCESS;
    };
    auto drain = [&](pgi_ctx::HostSlot& S, const Lay& L) {  // (after out_done) results of the slot's chunk into the caller's buffers
        if (S.pending < 0) return;
        const size_t c = (size_t)S.pending;
        const uint32_t p0 = cuts[c], np = cuts[c + 1] - p0;
        const uint64_t r0 = h_offsets[p0] - h_offsets[0], rows = h_offsets[p0 + np] - h_offsets[p0];
        memcpy(h_edges + p0, (char*)S.h_io + (L.edges - L.x1), (size_t)np * sizeof(pgi_edge));
        if (rows) memcpy(h_masks + r0, (char*)S.h_io + (L.masks - L.x1), rows);
        S.pending = -1;
    };
    uint32_t max_corr_of[kSlots] = {};
    // host side of a chunk in two halves, so that the copies of chunk c + 1 are queued BEFORE the launches of chunk c
    // (the half-dozen launches per chunk otherwise left the copy stream idle for ~60 us per chunk)
    auto stage_in = [&](size_t c) -> int {
        pgi_ctx::HostSlot& S = ctx->hslot[c % kSlots];
        const uint32_t p0 = cuts[c], np = cuts[c + 1] - p0;
        const uint64_t rbase = h_offsets[p0], rows = h_offsets[p0 + np] - rbase, r0 = rbase - h_offsets[0];
        const Lay L = layout(rows, np);
        if (S.used) {
            HIP_TRY(hipEventSynchronize(S.out_done));  // the slot's previous chunk has left the device
            drain(S, lay[c % kSlots]);
        }
        if ((!pin_in || !pin_out) && L.total > S.h_io_bytes) {
            if (S.h_io) (void)hipHostFree(S.h_io);
            S.h_io = nullptr; S.h_io_bytes = 0;
            HIP_TRY(hipHostMalloc(&S.h_io, L.total + L.total / 4, hipHostMallocDefault));
            S.h_io_bytes = L.total + L.total / 4;
        }
        if (L.total > S.bytes) {
            if (S.d) (void)hipFree(S.d);
            S.d = nullptr; S.bytes = 0;
            HIP_TRY(hipMalloc(&S.d, L.total + L.total / 4));
            S.bytes = L.total + L.total / 4;
        }
        char* d = (char*)S.d;
        // The per-pair arrays (rebased offsets, thresholds, guesses) travel as ONE copy from a page-locked staging block
        // of the slot that mirrors the device layout [L.off, L.has + np): four small copies from pageable memory cost
        // ~0.14 ms of the copy stream per chunk (13 % of its time on BASELINE config 2; profiles/r02_host_timeline.txt)
        const size_t small_bytes = (L.has + np) - L.off;
        if (small_bytes > S.h_small_bytes) {
            if (S.h_small) (void)hipHostFree(S.h_small);
            S.h_small = nullptr; S.h_small_bytes = 0;
            HIP_TRY(hipHostMalloc(&S.h_small, small_bytes + small_bytes / 4, hipHostMallocDefault));
            S.h_small_bytes = small_bytes + small_bytes / 4;
        }
        char* hs = (char*)S.h_small;
        uint64_t* ol = reinterpret_cast<uint64_t*>(hs);
        uint32_t max_corr = 0;
        for (uint32_t k = 0; k <= np; ++k) ol[k] = h_offsets[p0 + k] - rbase;
        for (uint32_t k = 0; k < np; ++k) max_corr = std::max(max_corr, (uint32_t)(ol[k + 1] - ol[k]));
        memcpy(hs + (L.thr - L.off), h_thr + p0, (size_t)np * 8);
        if (guesses) {
            memcpy(hs + (L.guess - L.off), h_guess_Rt + 12 * (size_t)p0, (size_t)np * 96);
            memcpy(hs + (L.has - L.off), h_has_guess + p0, np);
        }
        // pageable buffers: every copy is staged by the runtime on this thread; keep them on the slot's own stream so
        // that the copies of neighbouring chunks still overlap (one shared copy stream would serialise them)
        hipStream_t in = ctx->copy_in;
        if (rows && pin_in) {
            HIP_TRY(hipMemcpyAsync(d + L.x1, h_x1 + r0, rows * 4, hipMemcpyHostToDevice, in));
            HIP_TRY(hipMemcpyAsync(d + L.y1, h_y1 + r0, rows * 4, hipMemcpyHostToDevice, in));
            HIP_TRY(hipMemcpyAsync(d + L.x2, h_x2 + r0, rows * 4, hipMemcpyHostToDevice, in));
            HIP_TRY(hipMemcpyAsync(d + L.y2, h_y2 + r0, rows * 4, hipMemcpyHostToDevice, in));
        } else if (rows) {  // pageable: the four arrays into the slot's block in the device layout, one copy
            char* io = (char*)S.h_io;
            const float* from[4] = {h_x1 + r0, h_y1 + r0, h_x2 + r0, h_y2 + r0};
            const size_t at[4] = {0, L.y1 - L.x1, L.x2 - L.x1, L.y2 - L.x1};
            if (rows * 4 >= ((size_t)1 << 20)) {  // four arrays, four threads: one thread copies at ~10 GB/s, a chunk holds 16 MB
                std::thread helpers[3];
                for (int k = 1; k < 4; ++k) helpers[k - 1] = std::thread([=] { memcpy(io + at[k], from[k], rows * 4); });
                memcpy(io + at[0], from[0], rows * 4);
                for (std::thread& th : helpers) th.join();
            } else {
                for (int k = 0; k < 4; ++k) memcpy(io + at[k], from[k], rows * 4);
            }
            HIP_TRY(hipMemcpyAsync(d + L.x1, io, (L.y2 + rows * 4) - L.x1, hipMemcpyHostToDevice, in));
        }
        HIP_TRY(hipMemcpyAsync(d + L.off, hs, guesses ? small_bytes : (L.thr + (size_t)np * 8) - L.off, hipMemcpyHostToDevice, in));
        HIP_TRY(hipEventRecord(S.in_done, in));
        lay[c % kSlots] = L;
        max_corr_of[c % kSlots] = max_corr;
        return PGI_SUCCESS;
    };
    auto launch = [&](size_t c) -> int {
        pgi_ctx::HostSlot& S = ctx->hslot[c % kSlots];
        const Lay& L = lay[c % kSlots];
        const uint32_t p0 = cuts[c], np = cuts[c + 1] - p0, max_corr = max_corr_of[c % kSlots];
        char* d = (char*)S.d;
        hipStream_t ks = ctx->hslot[c & 1].stream;
        HIP_TRY(hipStreamWaitEvent(ks, S.in_done, 0));
        pgi_batch b{};
        b.d_x1 = (const float*)(d + L.x1); b.d_y1 = (const float*)(d + L.y1); b.d_x2 = (const float*)(d + L.x2); b.d_y2 = (const float*)(d + L.y2);
        b.d_offsets = (const uint64_t*)(d + L.off); b.d_thr = (const double*)(d + L.thr);
        b.d_guess_Rt = guesses ? (const double*)(d + L.guess) : nullptr;
        b.d_has_guess = guesses ? (const uint8_t*)(d + L.has) : nullptr;
        b.n_pairs = np; b.max_corr = max_corr; b.pair_id_base = pair_id_base + p0; b.seed = seed;
        const int rc = launch_estimate(ctx, ctx->prm, &b, (pgi_edge*)(d + L.edges), (uint8_t*)(d + L.masks), ks, &S.d_bucket, &S.bucket_bytes, nullptr,
                                       h_offsets[cuts[c + 1]] - h_offsets[cuts[c]]);
        if (rc < 0) return rc;
        HIP_TRY(hipEventRecord(S.k_done, ks));
        S.used = true;
        return PGI_SUCCESS;
    };
    {
        const int rc0 = stage_in(0);
        if (rc0 < 0) return rc0;
    }
    for (size_t c = 0; c < n_chunks; ++c) {
        if (c + 1 < n_chunks) {
            const int rc1 = stage_in(c + 1);
            if (rc1 < 0) return rc1;
        }
        // Results of chunk c - 2 go out on the stream both chunks share, queued BEFORE the kernel of chunk c (the stream is
        // first in, first out).  Two launches behind the front: a copy into pageable memory blocks this thread until
        // its kernel is done.
        if (c >= 2) {
            const int rc2 = fetch(c - 2);
            if (rc2 < 0) return rc2;
        }
        const int rcl = launch(c);
        if (rcl < 0) return rcl;
    }
    for (size_t c = n_chunks >= 2 ? n_chunks - 2 : 0; c < n_chunks; ++c) {
        const int rc2 = fetch(c);
        if (rc2 < 0) return rc2;
    }
    for (size_t k = 0; k < kSlots; ++k)
        if (ctx->hslot[k].used) {
            HIP_TRY(hipEventSynchronize(ctx->hslot[k].out_done));
            drain(ctx->hslot[k], lay[k]);
        }
    return PGI_SUCCESS;
}

namespace {
// takes a free slot of the pool (blocks while all PGI_PAIR_SLOTS are busy); released by the destructor
struct SlotLease {
    pgi_ctx* ctx;
    pgi_ctx::PairSlot* S;
    explicit SlotLease(pgi_ctx* c) : ctx(c), S(nullptr) {
        std::unique_lock<std::mutex> lk(ctx->slot_mu);
        for (;;) {
            pgi_ctx::PairSlot* fresh = nullptr;
            for (auto& s : ctx->pslot) {
                if (s.busy) continue;
                if (s.stream) { S = &s; break; }  // prefer a slot whose stream and buffers already exist
                if (!fresh) fresh = &s;
            }
            if (!S) S = fresh;
            if (S) break;
            ctx->slot_cv.wait(lk);
        }
        S->busy = true;
    }
    ~SlotLease() {
        {
            std::lock_guard<std::mutex> lk(ctx->slot_mu);
            S->busy = false;
        }
        ctx->slot_cv.notify_one();
    }
};
}  // namespace

int pgi_estimate_pose(pgi_ctx* ctx, const double* corr, uint32_t n, double thr, const double* guesses, uint32_t g,
                      uint32_t min_inliers, uint64_t seed, uint64_t pair_id, pgi_edge* h_edge, uint8_t* h_mask) {
    if (!ctx || !corr || !h_edge || !h_mask) return fail(PGI_ERR_INVALID, "null argument");
    if (g && !guesses) return fail(PGI_ERR_INVALID, "guess count without guesses");
    pgi_params prm;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        prm = ctx->prm;
    }
    if (min_inliers) prm.min_inliers = min_inliers;  // this call only (the seam's kMinimumInlierNumber_ argument)
    HIP_TRY(hipSetDevice(ctx->device));
    SlotLease lease(ctx);
    pgi_ctx::PairSlot& S = *lease.S;
    if (!S.stream) HIP_TRY(hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking));
    // layout (host staging == device scratch): x1 y1 x2 y2 (nf floats each) | off[2] | thr | guess[12] | has | edge | mask
    const size_t nf = ((size_t)n + 3) & ~(size_t)3;
    const size_t o_off = 4 * nf * 4, o_thr = o_off + 16, o_guess = o_thr + 8, o_has = o_guess + 96, o_edge = o_has + 8,
                 o_mask = o_edge + sizeof(pgi_edge), o_gc = (o_mask + nf + 63) & ~(size_t)63,
                 bytes = o_gc + (prm.lo_graph_cut ? 8 * nf : 0) + 64;  // (graph-cut local optimisation: two scratch words per row)
    if (bytes > S.bytes) {
        const size_t cap = bytes + bytes / 2;
        if (S.d) (void)hipFree(S.d);
        if (S.h) (void)hipHostFree(S.h);
        S.d = S.h = nullptr;
        S.bytes = 0;
        HIP_TRY(hipMalloc(&S.d, cap));
        HIP_TRY(hipHostMalloc(&S.h, cap, hipHostMallocDefault));
        S.bytes = cap;
    }
    char* h = (char*)S.h;
    char* d = (char*)S.d;
    float* hx1 = (float*)h;
    float* hy1 = hx1 + nf;
    float* hx2 = hy1 + nf;
    float* hy2 = hx2 + nf;
    for (uint32_t i = 0; i < n; ++i) {  // cv::Mat N x 4 CV_64F rows -> f32 SoA, straight into the pinned staging
        hx1[i] = (float)corr[4 * (size_t)i + 0];
        hy1[i] = (float)corr[4 * (size_t)i + 1];
        hx2[i] = (float)corr[4 * (size_t)i + 2];
        hy2[i] = (float)corr[4 * (size_t)i + 3];
    }
    uint64_t* hoff = (uint64_t*)(h + o_off);
    hoff[0] = 0;
    hoff[1] = n;
    *(double*)(h + o_thr) = thr;
    if (g) memcpy(h + o_guess, guesses + 12 * (size_t)(g - 1), 96);  // the last guess wins (:974-1029)
    h[o_has] = g ? 1 : 0;
    HIP_TRY(hipMemcpyAsync(d, h, o_edge, hipMemcpyHostToDevice, S.stream));
    pgi_batch b;
    b.d_x1 = (float*)d; b.d_y1 = b.d_x1 + nf; b.d_x2 = b.d_y1 + nf; b.d_y2 = b.d_x2 + nf;
    b.d_offsets = (uint64_t*)(d + o_off);
    b.d_thr = (double*)(d + o_thr);
    b.d_guess_Rt = g ? (double*)(d + o_guess) : nullptr;
    b.d_has_guess = g ? (uint8_t*)(d + o_has) : nullptr;
    b.n_pairs = 1; b.max_corr = n; b.pair_id_base = pair_id; b.seed = seed;
    // a single pair is never bucketed, so the launch needs no context-owned scratch and no context lock
    int rc = launch_estimate(ctx, prm, &b, (pgi_edge*)(d + o_edge), (uint8_t*)(d + o_mask), S.stream, nullptr, nullptr, nullptr, nf,
                             prm.lo_graph_cut ? (uint32_t*)(d + o_gc) : nullptr);
    if (rc != PGI_SUCCESS) return rc;
    HIP_TRY(hipMemcpyAsync(h + o_edge, d + o_edge, sizeof(pgi_edge) + n, hipMemcpyDeviceToHost, S.stream));
    HIP_TRY(hipStreamSynchronize(S.stream));
    memcpy(h_edge, h + o_edge, sizeof(pgi_edge));
    if (n) memcpy(h_mask, h + o_mask, n);
    return h_edge->status == PGI_EDGE_OK ? 1 : 0;
}

int pgi_host_register(void* h_ptr, uint64_t bytes) {
    if (!h_ptr || !bytes) return fail(PGI_ERR_INVALID, "null argument");
    HIP_TRY(hipHostRegister(h_ptr, (size_t)bytes, hipHostRegisterMapped | hipHostRegisterPortable));  // mapped: K1 may read it in place
    return PGI_SUCCESS;
}

int pgi_host_unregister(void* h_ptr) {
    if (!h_ptr) return fail(PGI_ERR_INVALID, "null argument");
    HIP_TRY(hipHostUnregister(h_ptr));
    return PGI_SUCCESS;
}

int pgi_score_pose_batch(pgi_ctx* ctx, const pgi_batch* b, const double* d_E, const double* d_tau2,
                         uint32_t* d_counts, uint8_t* d_masks) {
    if (!ctx || !b || !d_E || !d_tau2 || !d_counts) return fail(PGI_ERR_INVALID, "null argument");
    if (b->n_pairs == 0) return PGI_SUCCESS;
    HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(score_pose_kernel, dim3((b->n_pairs + 3) / 4), dim3(256), 0, ctx->stream, b->d_x1, b->d_y1,
                       b->d_x2, b->d_y2, b->d_offsets, d_E, d_tau2, b->n_pairs, d_counts, d_masks);
    HIP_TRY(hipGetLastError());
    return PGI_SUCCESS;
}

int pgi_score_pose_f64(pgi_ctx* ctx, const double* d_corr, const uint64_t* d_off, uint32_t n_pairs, const double* d_E,
                       const double* d_tau2, uint32_t* d_counts, uint8_t* d_masks) {
    if (!ctx || !d_corr || !d_off || !d_E || !d_tau2 || !d_counts) return fail(PGI_ERR_INVALID, "null argument");
    if (n_pairs == 0) return PGI_SUCCESS;
    HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(score_pose_f64_kernel, dim3((n_pairs + 3) / 4), dim3(256), 0, ctx->stream, d_corr, d_off, d_E,
                       d_tau2, n_pairs, d_counts, d_masks);
    HIP_TRY(hipGetLastError());
    return PGI_SUCCESS;
}

int pgi_screen_guesses(pgi_ctx* ctx, const uint32_t* d_counts, uint32_t min_count, uint8_t* d_has_guess, uint32_t n_pairs) {
    if (!ctx || (n_pairs && (!d_counts || !d_has_guess))) return fail(PGI_ERR_INVALID, "null argument");
    if (n_pairs == 0) return PGI_SUCCESS;
    HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(screen_guesses_kernel, dim3((n_pairs + 255) / 256), dim3(256), 0, ctx->stream, d_counts, min_count, d_has_guess, n_pairs);
    HIP_TRY(hipGetLastError());
    return PGI_SUCCESS;
}

int pgi_score_pose_f64_host(pgi_ctx* ctx, const double* h_corr_aos, uint32_t n, const double E[9], double tau2,
                            uint32_t early_exit_at, uint32_t* count, uint8_t* h_mask) {
    if (!ctx || !E || !count || (n && !h_corr_aos)) return fail(PGI_ERR_INVALID, "null argument");
    if (n == 0) {
        *count = 0;
        return 0;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    SlotLease lease(ctx);  // same pool as pgi_estimate_pose: A* threads and estimator threads share 32 private slots
    pgi_ctx::PairSlot& S = *lease.S;
    if (!S.stream) HIP_TRY(hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking));
    // layout (pinned staging == device scratch): rows (n x 4 f64) | E[9], tau2 | out[2] | mask (n)
    const size_t o_prm = (size_t)n * 32, o_out = o_prm + 80, o_mask = o_out + 16, bytes = o_mask + n + 64;
    if (bytes > S.bytes) {
        const size_t cap = bytes + bytes / 2;
        if (S.d) (void)hipFree(S.d);
        if (S.h) (void)hipHostFree(S.h);
        S.d = S.h = nullptr;
        S.bytes = 0;
        HIP_TRY(hipMalloc(&S.d, cap));
        HIP_TRY(hipHostMalloc(&S.h, cap, hipHostMallocDefault));
        S.bytes = cap;
    }
    char* h = (char*)S.h;
    char* d = (char*)S.d;
    memcpy(h, h_corr_aos, (size_t)n * 32);  // the caller's cv::Mat is pageable: one pass into the pinned block, one DMA
    memcpy(h + o_prm, E, 72);
    memcpy(h + o_prm + 72, &tau2, 8);
    HIP_TRY(hipMemcpyAsync(d, h, o_out, hipMemcpyHostToDevice, S.stream));
    hipLaunchKernelGGL(score_pose_f64_one_kernel, dim3(1), dim3(256), 0, S.stream, (const double*)d, n, (const double*)(d + o_prm),
                       early_exit_at, (uint32_t*)(d + o_out), h_mask ? (uint8_t*)(d + o_mask) : nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h + o_out, d + o_out, h_mask ? 16 + (size_t)n : 16, hipMemcpyDeviceToHost, S.stream));
    HIP_TRY(hipStreamSynchronize(S.stream));
    const uint32_t* out = (const uint32_t*)(h + o_out);
    const bool reached = out[1] != 0;
    *count = reached ? early_exit_at : out[0];  // the reference returns AT that inlier (graph_traversal.h:221-225)
    if (h_mask) memcpy(h_mask, h + o_mask, n);
    return reached ? 1 : 0;
}

int pgi_pose_from_essential_host(pgi_ctx* ctx, const double E[9], const double* h_corr_aos, uint32_t n, const uint8_t* h_mask,
                                 double R[9], double t[3], uint32_t* votes, uint32_t* cand) {
    if (!ctx || !E || !R || !t || (n && !h_corr_aos)) return fail(PGI_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    SlotLease lease(ctx);  // the pool of pgi_estimate_pose / pgi_score_pose_f64_host: re-entrant, no allocation once warm
    pgi_ctx::PairSlot& S = *lease.S;
    if (!S.stream) HIP_TRY(hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking));
    // layout (pinned staging == device scratch): rows (n x 4 f64) | E[9] (+ pad) | mask (n, padded to 8) | out: R[9] t[3] votes cand
    const size_t o_prm = (size_t)n * 32, o_mask = o_prm + 80, o_out = o_mask + (((size_t)n + 7) & ~(size_t)7), bytes = o_out + 104 + 64;
    if (bytes > S.bytes) {
        const size_t cap = bytes + bytes / 2;
        if (S.d) (void)hipFree(S.d);
        if (S.h) (void)hipHostFree(S.h);
        S.d = S.h = nullptr;
        S.bytes = 0;
        HIP_TRY(hipMalloc(&S.d, cap));
        HIP_TRY(hipHostMalloc(&S.h, cap, hipHostMallocDefault));
        S.bytes = cap;
    }
    char* h = (char*)S.h;
    char* d = (char*)S.d;
    if (n) memcpy(h, h_corr_aos, (size_t)n * 32);
    memcpy(h + o_prm, E, 72);
    const bool masked = h_mask != nullptr && !ctx->prm.vote_all_rows;  // vote_all_rows: the reference's population (pose_utils.h:203)
    if (masked && n) memcpy(h + o_mask, h_mask, n);
    HIP_TRY(hipMemcpyAsync(d, h, masked ? o_out : o_mask, hipMemcpyHostToDevice, S.stream));
    hipLaunchKernelGGL(decompose_one_kernel, dim3(1), dim3(256), 0, S.stream, (const double*)d, n, (const double*)(d + o_prm),
                       masked ? (const uint8_t*)(d + o_mask) : nullptr, (double*)(d + o_out));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h + o_out, d + o_out, 104, hipMemcpyDeviceToHost, S.stream));
    HIP_TRY(hipStreamSynchronize(S.stream));
    memcpy(R, h + o_out, 72);
    memcpy(t, h + o_out + 72, 24);
    const uint32_t* tail = (const uint32_t*)(h + o_out + 96);
    if (votes) *votes = tail[0];
    if (cand) *cand = tail[1];
    return PGI_SUCCESS;
}

int pgi_decompose_batch(pgi_ctx* ctx, const pgi_batch* b, const double* d_E, const uint8_t* d_masks,
                        pgi_edge* d_edges) {
    if (!ctx || !b || !d_E || !d_edges) return fail(PGI_ERR_INVALID, "null argument");
    if (b->n_pairs == 0) return PGI_SUCCESS;
    HIP_TRY(hipSetDevice(ctx->device));
    if (ctx->prm.vote_all_rows) d_masks = nullptr;  // the reference's population (pose_utils.h:203), as in K1's epilogue
    hipLaunchKernelGGL(decompose_kernel, dim3(b->n_pairs), dim3(256), 0, ctx->stream, b->d_x1, b->d_y1, b->d_x2,
                       b->d_y2, b->d_offsets, d_E, d_masks, d_edges);
    HIP_TRY(hipGetLastError());
    return PGI_SUCCESS;
}

int pgi_five_point_batch(pgi_ctx* ctx, const float* d_pts, uint32_t n_samples, float* d_models, uint32_t* d_counts,
                         double* d_dbg) {
    if (!ctx || !d_pts || !d_models || !d_counts) return fail(PGI_ERR_INVALID, "null argument");
    if (n_samples == 0) return PGI_SUCCESS;
    HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(five_point_kernel, dim3((n_samples + 3) / 4), dim3(64), 0, ctx->stream, d_pts, n_samples,
                       d_models, d_counts, d_dbg);
    HIP_TRY(hipGetLastError());
    return PGI_SUCCESS;
}

}  // extern "C"
