/*
 * pgi_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See pgi_oracle.h for scope, the "parity unpinned" statement and citations.
 * Build: gcc -std=c11 -O2 -ffp-contract=off -mfma -mavx2 -fopenmp -fPIC -shared
 * Every FP statement is one IEEE op in a fixed order (the HIP kernels mirror it).
 */
#include "pgi_oracle.h"
#define _USE_MATH_DEFINES
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ======================================================================= */
/* reference in-tree arithmetic (f64)                                       */
/* ======================================================================= */

/* graph_traversal.h:86-116 -- same operand order, plain mul/add (no fma)  */
double pgo_ref_sampson_sq(const double s[4], const double E[9]) {
    const double x1 = s[0], y1 = s[1], x2 = s[2], y2 = s[3];
    const double e11 = E[0], e12 = E[1], e13 = E[2], e21 = E[3], e22 = E[4], e23 = E[5],
                 e31 = E[6], e32 = E[7], e33 = E[8];
    double rxc = e11 * x2 + e21 * y2 + e31;
    double ryc = e12 * x2 + e22 * y2 + e32;
    double rwc = e13 * x2 + e23 * y2 + e33;
    double r = (x1 * rxc + y1 * ryc + rwc);
    double rx = e11 * x1 + e12 * y1 + e13;
    double ry = e21 * x1 + e22 * y1 + e23;
    return r * r / (rxc * rxc + ryc * ryc + rx * rx + ry * ry);
}

/* graph_traversal.h:136-168: squared residual compared with the UN-squared
 * threshold (line 164 uses kThreshold_, not kSquaredThreshold of line 149). */
uint32_t pgo_ref_get_inliers(const double* c, uint32_t n, const double E[9], double thr,
                             uint32_t* idx) {
    uint32_t k = 0;
    for (uint32_t i = 0; i < n; ++i)
        if (pgo_ref_sampson_sq(c + 4 * (size_t)i, E) < thr) idx[k++] = i;
    return k;
}

/* pose_utils.h:74-86: E = [t]x R */
void pgo_ref_essential_from_pose(const double R[9], const double t[3], double E[9]) {
    const double tx[9] = {0, -t[2], t[1], t[2], 0, -t[0], -t[1], t[0], 0};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += tx[3 * i + k] * R[3 * k + j];
            E[3 * i + j] = s;
        }
}

/* graph_traversal.h:194-233: squared threshold, early exit at min_inl */
int pgo_ref_pose_test(const double* c, uint32_t n, const double R[9], const double t[3],
                      double thr, uint32_t min_inl, uint32_t* n_inl) {
    double E[9];
    pgo_ref_essential_from_pose(R, t, E);
    const double thr2 = thr * thr;
    *n_inl = 0;
    for (uint32_t i = 0; i < n; ++i)
        if (pgo_ref_sampson_sq(c + 4 * (size_t)i, E) < thr2) {
            ++*n_inl;
            if (*n_inl >= min_inl) return 1;
        }
    return 0;
}

/* graph_traversal.h:340-344: pose = T_edge * pose, or T_edge^-1 * pose */
void pgo_ref_chain_pose(const double Re[9], const double te[3], int inverted, double R[9],
                        double t[3]) {
    double A[9], a[3];
    if (inverted) { /* inverse of (Re,te) = (Re^T, -Re^T te) */
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) A[3 * i + j] = Re[3 * j + i];
        for (int i = 0; i < 3; ++i)
            a[i] = -(A[3 * i] * te[0] + A[3 * i + 1] * te[1] + A[3 * i + 2] * te[2]);
    } else {
        memcpy(A, Re, sizeof A);
        memcpy(a, te, sizeof a);
    }
    double Rn[9], tn[3];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j)
            Rn[3 * i + j] = A[3 * i] * R[j] + A[3 * i + 1] * R[3 + j] + A[3 * i + 2] * R[6 + j];
        tn[i] = A[3 * i] * t[0] + A[3 * i + 1] * t[1] + A[3 * i + 2] * t[2] + a[i];
    }
    memcpy(R, Rn, sizeof Rn);
    memcpy(t, tn, sizeof tn);
}

/* pose_graph_builder.h:864-938 */
void pgo_ref_normalize_corr(const float* ks, const float* kd, const uint32_t* ms,
                            const uint32_t* md, uint32_t m, double f_src, double w_src,
                            double h_src, double f_dst, double w_dst, double h_dst,
                            int src_for_dst, double thr_px, double* c, double* thr_norm) {
    /* K = [f,0,w/2; 0,f,h/2; 0,0,1]  (pose_graph_builder.h:286) */
    const double sfx = f_src, sfy = f_src, spx = w_src / 2.0, spy = h_src / 2.0;
    double dfx = f_dst, dfy = f_dst, dpx = w_dst / 2.0, dpy = h_dst / 2.0;
    if (src_for_dst) { dfx = sfx; dfy = sfy; dpx = spx; dpy = spy; } /* :908-912 */
    for (uint32_t i = 0; i < m; ++i) {
        c[4 * i + 0] = ((double)ks[2 * ms[i] + 0] - spx) / sfx;
        c[4 * i + 1] = ((double)ks[2 * ms[i] + 1] - spy) / sfy;
        c[4 * i + 2] = ((double)kd[2 * md[i] + 0] - dpx) / dfx;
        c[4 * i + 3] = ((double)kd[2 * md[i] + 1] - dpy) / dfy;
    }
    *thr_norm = thr_px / ((sfx + sfy + dfx + dfy) / 4.0); /* :934-937 */
}

/* ======================================================================= */
/* engine spec                                                              */
/* ======================================================================= */

void pgo_default_params(pgo_params* p) {
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
    p->lo_graph_cut = 0;
    p->sampler = 0;
}

uint64_t pgo_mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* murmur3 finaliser: the per-draw hash (32-bit multiplies only) */
static inline uint32_t fmix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}
uint32_t pgo_draw_index(uint64_t base, uint32_t hyp, uint32_t k, uint32_t n) {
    const uint32_t lo = (uint32_t)base, hi = (uint32_t)(base >> 32);
    const uint32_t h = fmix32((lo ^ (hyp * 0x9E3779B1u)) + (hi ^ (k * 0x85EBCA77u)));
    return (uint32_t)(((uint64_t)h * (uint64_t)n) >> 32);
}
/* five distinct indices; counter-based, independent of execution order: draws k = 0,1,2,... are
 * accepted in order unless they repeat an accepted index */
void pgo_sample5(uint64_t seed, uint64_t pair_id, uint32_t hyp, uint32_t n, uint32_t idx[5]) {
    const uint64_t base = pgo_mix64(seed ^ pgo_mix64(pair_id));
    uint32_t k = 0, got = 0;
    while (got < 5) {
        const uint32_t j = pgo_draw_index(base, hyp, k, n);
        ++k;
        int dup = 0;
        for (uint32_t q = 0; q < got; ++q) dup |= (idx[q] == j);
        if (!dup || k >= 64) idx[got++] = j; /* k>=64: give up on distinctness */
    }
}

/* Progressive sampling (pgi_params.sampler = 1): hypothesis `hyp` samples the first n(hyp) rows,
 * n(hyp) = max(5, floor(n * (s / 64)^(1/5))) with s = ceil(64 * (hyp + 1) / T) (T = max_iters): PROSAC's growth
 * (n / N)^5 ~ t / T_N in 64 steps.  Integer arithmetic on a table of floor(2^32 * (i / 64)^(1/5)) (exact values), the
 * same table as the kernels' (progressive_rows, csrc/pgi_device.hpp). */
static const uint32_t kFifthRoot64[65] = {
    0x00000000u, 0x6F6E336Bu, 0x80000000u, 0x8ACFF893u, 0x93088C35u, 0x99BE7209u, 0x9F741C86u, 0xA472339Du, 0xA8E5A29Du, 0xACEC422Fu, 0xB09AFB46u,
    0xB4010FB1u, 0xB729FEAEu, 0xBA1EAD01u, 0xBCE6228Cu, 0xBF860882u, 0xC203001Du, 0xC460DFEFu, 0xC6A2E032u, 0xC8CBBB8Du, 0xCADDC7B6u, 0xCCDB0842u,
    0xCEC53D43u, 0xD09DEEBCu, 0xD26675BFu, 0xD42003BEu, 0xD5CBA87Au, 0xD76A56DCu, 0xD8FCE8FAu, 0xDA842364u, 0xDC00B7F1u, 0xDD734813u, 0xDEDC66D6u,
    0xE03C9A92u, 0xE1945E57u, 0xE2E42331u, 0xE42C513Du, 0xE56D4891u, 0xE6A76213u, 0xE7DAF02Eu, 0xE9083F70u, 0xEA2F9717u, 0xEB513993u, 0xEC6D64ECu,
    0xED84532Eu, 0xEE963AB8u, 0xEFA34E8Fu, 0xF0ABBEA6u, 0xF1AFB819u, 0xF2AF6567u, 0xF3AAEEA8u, 0xF4A279B9u, 0xF5962A67u, 0xF6862294u, 0xF772825Eu,
    0xF85B6838u, 0xF940F10Eu, 0xFA23385Bu, 0xFB025844u, 0xFBDE69ABu, 0xFCB78447u, 0xFD8DBEB6u, 0xFE612E8Du, 0xFF31E869u, 0xFFFFFFFFu};
uint32_t pgo_progressive_rows(uint32_t hyp, uint32_t n, uint32_t T) {
    if (n <= 5 || T == 0) return n;
    const uint64_t step = ((uint64_t)(hyp + 1) * 64u + (T - 1)) / T;
    if (step >= 64) return n;
    const uint32_t m = (uint32_t)(((uint64_t)kFifthRoot64[step] * (uint64_t)n) >> 32);
    return m < 5 ? 5 : m;
}

/* ---- scoring (f32) ---------------------------------------------------- */
static inline void sampson_terms(const float e[9], float x1, float y1, float x2, float y2,
                                 float* r2, float* den) {
    const float rxc = fmaf(e[0], x2, fmaf(e[3], y2, e[6]));
    const float ryc = fmaf(e[1], x2, fmaf(e[4], y2, e[7]));
    const float rwc = fmaf(e[2], x2, fmaf(e[5], y2, e[8]));
    const float r = fmaf(x1, rxc, fmaf(y1, ryc, rwc));
    const float rx = fmaf(e[0], x1, fmaf(e[1], y1, e[2]));
    const float ry = fmaf(e[3], x1, fmaf(e[4], y1, e[5]));
    *den = fmaf(rxc, rxc, fmaf(ryc, ryc, fmaf(rx, rx, ry * ry)));
    *r2 = r * r;
}

void pgo_score_model(const float E[9], const float* x1, const float* y1, const float* x2,
                     const float* y2, uint32_t n, double thr, uint32_t* score,
                     uint32_t* n_inl) {
    const float thr2 = (float)(thr * thr);
    uint32_t s = 0, c = 0;
    for (uint32_t i = 0; i < n; ++i) {
        float r2, den;
        sampson_terms(E, x1[i], y1[i], x2[i], y2[i], &r2, &den);
        const float t = thr2 * den;
        /* levels at sampson distance {0.5, 0.75, 1, 1.5} * thr */
        s += (r2 < 0.25f * t) + (r2 < 0.5625f * t) + (r2 < t) + (r2 < 2.25f * t);
        c += (r2 < t);
    }
    *score = s;
    *n_inl = c;
}

/* pgo_score_model with an exact early exit: the levels only add, so a model whose partial score plus 4 per remaining
 * row cannot exceed `bar` (the best score that still matters) can be dropped without changing any result.  Checked
 * every 64 rows, like the kernels do (score_queue, csrc/pgi_kernels.hip).  Returns 0 if the model was dropped. */
static int score_model_bounded(const float E[9], const float* x1, const float* y1, const float* x2, const float* y2,
                               uint32_t n, double thr, long bar, uint32_t* score, uint32_t* n_inl) {
    const float thr2 = (float)(thr * thr);
    uint32_t s = 0, c = 0;
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t end = base + 64 < n ? base + 64 : n;
        for (uint32_t i = base; i < end; ++i) {
            float r2, den;
            sampson_terms(E, x1[i], y1[i], x2[i], y2[i], &r2, &den);
            const float t = thr2 * den;
            s += (r2 < 0.25f * t) + (r2 < 0.5625f * t) + (r2 < t) + (r2 < 2.25f * t);
            c += (r2 < t);
        }
        if ((long)s + 4l * (long)(n - end) <= bar) return 0;
    }
    *score = s;
    *n_inl = c;
    return 1;
}

uint32_t pgo_mask_model(const float E[9], const float* x1, const float* y1, const float* x2,
                        const float* y2, uint32_t n, float tau2, uint8_t* mask) {
    uint32_t c = 0;
    for (uint32_t i = 0; i < n; ++i) {
        float r2, den;
        sampson_terms(E, x1[i], y1[i], x2[i], y2[i], &r2, &den);
        const uint8_t in = r2 < tau2 * den;
        mask[i] = in;
        c += in;
    }
    return c;
}

/* ---- graph-cut local optimisation: the labelling step of GC-RANSAC (Barath & Matas, CVPR 2018) -------------------------
 * north_star names the estimator "GC-RANSAC"; the reference keeps only a commented-out binding of it
 * (/root/reference/src/pyposegraphbuilder/src/bindings.cpp:5,228), so this follows the published algorithm: when a new best
 * model is found, the rows the refit uses are not "residual below the threshold" but the labelling L (inlier / outlier) that
 * minimises
 *     E(L) = sum_p U_p(L_p) + lambda * sum_{(p,q) neighbours} B_pq(L_p, L_q)
 * with the kernel K_p = max(0, 1 - d_p^2 / (1.5 thr)^2) of the squared Sampson distance d_p^2,
 *     U_p(outlier) = K_p, U_p(inlier) = 1 - K_p,
 *     B_pq = 1 if the labels differ, (K_p + K_q) / 2 if both are outliers, 1 - (K_p + K_q) / 2 if both are inliers
 * (the paper's eq. 2-4) -- a row next to good rows is pulled in, a stray row inside the band among bad neighbours is pushed out.
 * What is specific to this build, so that CPU and GPU agree bit for bit and the cut is cheap:
 *  - K is quantised to PGO_GC_LEVELS = 16 levels by a ladder of f32 comparisons r^2 < (j / 16) (1.5 thr)^2 den (the operands of
 *    the scoring staircase): every energy is an integer, U = 128 k, lambda enters as lambda64 = lambda * 64 (9 ~ 0.14, the
 *    paper's value): B = lambda64 * (k_p + k_q), lambda64 * (32 - k_p - k_q), lambda64 * 32 (all x 2 against the formulas above,
 *    like U);
 *  - neighbours: the paper's grid neighbourhood over the 4-D correspondence space (cells of 1/8 in normalised image
 *    coordinates per axis), with the rows of a cell linked IN INDEX ORDER -- a spanning path of the paper's clique -- so the
 *    neighbourhood graph is a set of chains;
 *  - on a set of chains the minimum s-t cut of this (submodular) energy is found exactly by a forward / backward sweep
 *    (dynamic programming over delta = m(inlier) - m(outlier)); ties go to "outlier".  tests/test_graph_cut.py checks the
 *    sweep's energy against a generic max-flow (scipy) on the same graphs. */
uint32_t pgo_gc_cell(float x1, float y1, float x2, float y2) {
    const int a = (int)floorf(x1 * 8.0f), b = (int)floorf(y1 * 8.0f), c = (int)floorf(x2 * 8.0f), d = (int)floorf(y2 * 8.0f);
    return (uint32_t)(a & 7) | ((uint32_t)(b & 7) << 3) | ((uint32_t)(c & 7) << 6) | ((uint32_t)(d & 7) << 9);
}

/* prev[i] = the last row before i in the same cell (PGO_GC_NONE: i starts a chain) */
void pgo_gc_chains(const float* x1, const float* y1, const float* x2, const float* y2, uint32_t n, uint32_t* prev) {
    uint32_t last[4096];
    for (int c = 0; c < 4096; ++c) last[c] = PGO_GC_NONE;
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t c = pgo_gc_cell(x1[i], y1[i], x2[i], y2[i]);
        prev[i] = last[c];
        last[c] = i;
    }
}

/* k in 0..16: how many of the levels j = 1..16 the row passes, r^2 < (j / 16) * (1.5 thr)^2 * den */
uint32_t pgo_gc_kernel_level(const float E[9], float x1, float y1, float x2, float y2, float thr2) {
    float r2, den;
    sampson_terms(E, x1, y1, x2, y2, &r2, &den);
    const float t = (den * thr2) * 2.25f;
    uint32_t k = 0;
    for (uint32_t j = 1; j <= PGO_GC_LEVELS; ++j) k += r2 < t * ((float)j * 0.0625f);
    return k;
}

static inline void gc_pairwise(uint32_t kp, uint32_t kq, uint32_t lambda64, int32_t* v00, int32_t* v11, int32_t* v01) {
    *v00 = (int32_t)(lambda64 * (kp + kq));
    *v11 = (int32_t)(lambda64 * (2u * PGO_GC_LEVELS - kp - kq));
    *v01 = (int32_t)(lambda64 * 2u * PGO_GC_LEVELS);
}

/* energy of a labelling (test hook: the sweep's result against a generic max-flow) */
int64_t pgo_gc_energy(const uint32_t* k, const uint32_t* prev, const uint8_t* labels, uint32_t n, uint32_t lambda64) {
    int64_t e = 0;
    for (uint32_t i = 0; i < n; ++i) {
        e += labels[i] ? (int64_t)PGO_GC_UNARY * (PGO_GC_LEVELS - k[i]) : (int64_t)PGO_GC_UNARY * k[i];
        if (prev[i] != PGO_GC_NONE) {
            int32_t v00, v11, v01;
            gc_pairwise(k[prev[i]], k[i], lambda64, &v00, &v11, &v01);
            e += labels[i] != labels[prev[i]] ? v01 : (labels[i] ? v11 : v00);
        }
    }
    return e;
}

/* The cut itself: kernel levels k and chains prev in, labels out (1 = inlier side of the minimum cut); returns the inliers. */
uint32_t pgo_gc_cut(const uint32_t* k, const uint32_t* prev, uint32_t n, uint32_t lambda64, uint8_t* labels) {
    int32_t* delta = (int32_t*)malloc(sizeof(int32_t) * (n ? n : 1));
    /* forward: delta_i = m_i(inlier) - m_i(outlier), messages along prev (prev[i] < i) */
    for (uint32_t i = 0; i < n; ++i) {
        int32_t d = (int32_t)PGO_GC_UNARY * ((int32_t)PGO_GC_LEVELS - 2 * (int32_t)k[i]); /* U(in) - U(out) */
        if (prev[i] != PGO_GC_NONE) {
            int32_t v00, v11, v01;
            gc_pairwise(k[prev[i]], k[i], lambda64, &v00, &v11, &v01);
            const int32_t dp = delta[prev[i]];
            const int32_t in1 = dp + v11 < v01 ? dp + v11 : v01; /* best way into "inlier", relative to m_prev(outlier) */
            const int32_t in0 = dp + v01 < v00 ? dp + v01 : v00; /* ... into "outlier" */
            d += in1 - in0;
        }
        delta[i] = d;
        labels[i] = 2; /* undecided */
    }
    /* backward: a row nobody has decided yet ends its chain and takes its own minimum; every row decides its predecessor */
    uint32_t cnt = 0;
    for (uint32_t ii = n; ii-- > 0;) {
        if (labels[ii] == 2) labels[ii] = delta[ii] < 0;
        cnt += labels[ii];
        if (prev[ii] != PGO_GC_NONE) {
            int32_t v00, v11, v01;
            gc_pairwise(k[prev[ii]], k[ii], lambda64, &v00, &v11, &v01);
            const int32_t dp = delta[prev[ii]];
            labels[prev[ii]] = labels[ii] ? (dp + v11 < v01) : (dp + v01 < v00);
        }
    }
    free(delta);
    return cnt;
}

/* labels[i] = 1: row i is an inlier of the minimum cut under model E; returns their number (energy: optional out) */
uint32_t pgo_gc_labels(const float E[9], const float* x1, const float* y1, const float* x2, const float* y2, uint32_t n,
                       double thr, uint32_t lambda64, const uint32_t* prev, uint8_t* labels, int64_t* energy) {
    const float thr2 = (float)(thr * thr);
    uint32_t* k = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
    for (uint32_t i = 0; i < n; ++i) k[i] = pgo_gc_kernel_level(E, x1[i], y1[i], x2[i], y2[i], thr2);
    const uint32_t cnt = pgo_gc_cut(k, prev, n, lambda64, labels);
    if (energy) *energy = pgo_gc_energy(k, prev, labels, n, lambda64);
    free(k);
    return cnt;
}

/* Pre-verification (T(d,d)-style, Matas & Chum): once a best model with n_bar inliers exists, a
 * candidate must show at least floor(16*n_bar/n) rows within 1.5*thr among the FIRST min(64,n)
 * rows -- a quarter of what a model as good as the best is expected to show -- or it is dropped
 * unscored.  Keyed only on state fixed at round start, so it is execution-order independent. */
int pgo_preverify(const float E[9], const float* x1, const float* y1, const float* x2, const float* y2,
                  uint32_t n, double thr, uint32_t n_bar) {
    if (n_bar == 0) return 1;
    const float thr2 = (float)(thr * thr);
    const uint32_t k_min = (uint32_t)((16ull * n_bar) / n);
    const uint32_t m = n < 64 ? n : 64;
    uint32_t c3 = 0;
    for (uint32_t i = 0; i < m; ++i) {
        float r2, den;
        sampson_terms(E, x1[i], y1[i], x2[i], y2[i], &r2, &den);
        c3 += (r2 < 2.25f * (thr2 * den));
    }
    return c3 >= k_min;
}

/* ---- pivot key: high word of |a| with the row packed into the low 4 bits */
static inline int32_t pivot_key(double a, int row) {
    uint64_t b;
    memcpy(&b, &a, 8);
    const uint32_t hi = (uint32_t)(b >> 32) & 0x7FFFFFFFu;
    return (int32_t)((hi & 0xFFFFFFF0u) | (uint32_t)(15 - row));
}

/* Gauss-Jordan with partial pivoting on an R x C matrix over the first P
 * columns; prow[k] = row holding the pivot of column k. */
static void gauss_jordan(double* a, int R, int C, int P, int* prow) {
    int used[16] = {0};
    for (int k = 0; k < P; ++k) {
        int p = 0;
        int32_t best = -1;
        for (int r = 0; r < R; ++r) {
            const int32_t key = used[r] ? -1 : pivot_key(a[r * C + k], r);
            if (key > best) { best = key; p = r; }
        }
        used[p] = 1;
        prow[k] = p;
        const double inv = 1.0 / a[p * C + k];
        for (int j = k + 1; j < C; ++j) a[p * C + j] = a[p * C + j] * inv;
        for (int r = 0; r < R; ++r) { /* the pivot row runs the same update with factor 0 (lane-uniform code) */
            const double f = (r == p) ? 0.0 : a[r * C + k];
            if (r == p) continue;
            for (int j = k + 1; j < C; ++j) a[r * C + j] = fma(-f, a[p * C + j], a[r * C + j]);
        }
        for (int j = k + 1; j < C; ++j) a[p * C + j] = fma(-0.0, a[p * C + j], a[p * C + j]);
    }
}

/* ---- 5x9 null space ---------------------------------------------------- */
static inline void epi_row(const float p[4], double a[9]) {
    const double x1 = p[0], y1 = p[1], x2 = p[2], y2 = p[3];
    a[0] = x2 * x1; a[1] = x2 * y1; a[2] = x2;
    a[3] = y2 * x1; a[4] = y2 * y1; a[5] = y2;
    a[6] = x1;      a[7] = y1;      a[8] = 1.0;
}

/* dot product as a fixed pairwise tree (maps onto a 16-lane butterfly):
 * ((p0+p1)+(p2+p3)) + ((p4+p5)+(p6+p7)), then + p8 */
static inline double dot9_tree(const double* a, const double* b) {
    double p[9];
    for (int i = 0; i < 9; ++i) p[i] = a[i] * b[i];
    const double q0 = (p[0] + p[1]) + (p[2] + p[3]);
    const double q1 = (p[4] + p[5]) + (p[6] + p[7]);
    return (q0 + q1) + p[8];
}

static void orthonormalise4(double v[4][9]) {
    for (int f = 0; f < 4; ++f) {
        for (int g = 0; g < f; ++g) {
            const double d = dot9_tree(v[g], v[f]);
            for (int i = 0; i < 9; ++i) v[f][i] = fma(-d, v[g][i], v[f][i]);
        }
        const double nn = dot9_tree(v[f], v[f]);
        const double inv = 1.0 / sqrt(nn);
        for (int i = 0; i < 9; ++i) v[f][i] = v[f][i] * inv;
    }
}

void pgo_nullspace5(const float pts[5][4], double basis[36]) {
    double a[5 * 9];
    int prow[5];
    for (int r = 0; r < 5; ++r) epi_row(pts[r], a + 9 * r);
    gauss_jordan(a, 5, 9, 5, prow);
    double v[4][9];
    for (int f = 0; f < 4; ++f) {
        for (int k = 0; k < 5; ++k) v[f][k] = -a[prow[k] * 9 + 5 + f];
        for (int g = 0; g < 4; ++g) v[f][5 + g] = (g == f) ? 1.0 : 0.0;
    }
    orthonormalise4(v);
    memcpy(basis, v, sizeof v);
}

/* ---- Nister back-end ---------------------------------------------------- */
/* lin: [x,y,z,1]; quad: [x2,xy,xz,x,y2,yz,y,z2,z,1];
 * cubic (Nister's order): x3 y3 x2y xy2 x2z x2 y2z y2 xyz xy | xz2 xz x yz2 yz y z3 z2 z 1 */
static const int QI[4][4] = {{0, 1, 2, 3}, {1, 4, 5, 6}, {2, 5, 7, 8}, {3, 6, 8, 9}};
static const int CI[10][4] = {{0, 2, 4, 5},     {2, 3, 8, 9},     {4, 8, 10, 11}, {5, 9, 11, 12},
                              {3, 1, 6, 7},     {8, 6, 13, 14},   {9, 7, 14, 15}, {10, 13, 16, 17},
                              {11, 14, 17, 18}, {12, 15, 18, 19}};
/* phase-1 quads: up to 3 terms (entryA, entryB, sign); sign 0 = unused */
static const int QT[9][3][3] = {
    {{0, 0, 1}, {1, 1, 1}, {2, 2, 1}}, /* Q0 = EEt_00 */
    {{0, 3, 1}, {1, 4, 1}, {2, 5, 1}}, /* Q1 = EEt_01 */
    {{0, 6, 1}, {1, 7, 1}, {2, 8, 1}}, /* Q2 = EEt_02 */
    {{3, 3, 1}, {4, 4, 1}, {5, 5, 1}}, /* Q3 = EEt_11 */
    {{3, 6, 1}, {4, 7, 1}, {5, 8, 1}}, /* Q4 = EEt_12 */
    {{6, 6, 1}, {7, 7, 1}, {8, 8, 1}}, /* Q5 = EEt_22 */
    {{4, 8, 1}, {5, 7, -1}, {0, 0, 0}}, /* Q6 = E11E22 - E12E21 */
    {{3, 8, 1}, {5, 6, -1}, {0, 0, 0}}, /* Q7 = E10E22 - E12E20 */
    {{3, 7, 1}, {4, 6, -1}, {0, 0, 0}}, /* Q8 = E10E21 - E11E20 */
};
/* phase-2 cubic rows: 3 terms (quad index, entry index, sign).  Quads 0,3,5
 * are read as Lambda_ii = Q - tr/2 for rows 1..9. */
static const int LAM[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};

static void build_constraints(const double basis[36], double cons[10][20]) {
    double L[9][4];
    for (int e = 0; e < 9; ++e)
        for (int b = 0; b < 4; ++b) L[e][b] = basis[9 * b + e];
    double Q[9][10];
    for (int q = 0; q < 9; ++q) {
        for (int c = 0; c < 10; ++c) Q[q][c] = 0.0;
        for (int t = 0; t < 3; ++t) {
            const int sg = QT[q][t][2];
            if (!sg) continue;
            const double* A = L[QT[q][t][0]];
            const double* B = L[QT[q][t][1]];
            for (int a = 0; a < 4; ++a) {
                const double av = sg > 0 ? A[a] : -A[a];
                for (int b = 0; b < 4; ++b) Q[q][QI[a][b]] = fma(av, B[b], Q[q][QI[a][b]]);
            }
        }
    }
    double tr[10], Lam[6][10];
    for (int c = 0; c < 10; ++c) tr[c] = (Q[0][c] + Q[3][c]) + Q[5][c];
    for (int q = 0; q < 6; ++q)
        for (int c = 0; c < 10; ++c)
            Lam[q][c] = (q == 0 || q == 3 || q == 5) ? Q[q][c] - 0.5 * tr[c] : Q[q][c];
    for (int row = 0; row < 10; ++row) {
        double* c = cons[row];
        for (int m = 0; m < 20; ++m) c[m] = 0.0;
        for (int k = 0; k < 3; ++k) {
            const double* Qk;
            const double* Lk;
            int sg = 1;
            if (row == 0) {
                Qk = Q[6 + k];
                Lk = L[k];
                sg = (k == 1) ? -1 : 1;
            } else {
                const int i = (row - 1) / 3, j = (row - 1) % 3;
                Qk = Lam[LAM[i][k]];
                Lk = L[3 * k + j];
            }
            for (int q = 0; q < 10; ++q)
                for (int l = 0; l < 4; ++l) {
                    const double lv = sg > 0 ? Lk[l] : -Lk[l];
                    c[CI[q][l]] = fma(Qk[q], lv, c[CI[q][l]]);
                }
        }
    }
}

static inline double grid_point(int j) { /* z = u/(1-u^2), u uniform in (-1,1) */
    const double u = (double)(2 * j - PGO_GRID) / (double)(PGO_GRID + 1);
    return u / (1.0 - u * u);
}

static inline double horner10(const double p[11], double x) {
    double v = p[10];
    for (int c = 9; c >= 0; --c) v = fma(v, x, p[c]);
    return v;
}

static double refine_root(const double p[11], double lo, double hi, int slo) {
    /* safeguarded Newton; returns the evaluated iterate with the smallest |p| */
    double x = 0.5 * (lo + hi), xbest = x, vbest = INFINITY;
    for (int it = 0; it < PGO_NEWTON_ITERS; ++it) {
        double v = p[10], d = 0.0;
        for (int c = 9; c >= 0; --c) {
            d = fma(d, x, v);
            v = fma(v, x, p[c]);
        }
        const double av = fabs(v);
        if (av < vbest) { vbest = av; xbest = x; }
        if ((v < 0.0) == slo) lo = x; else hi = x;
        double xn = x - v / d;
        if (!(xn >= lo && xn <= hi)) xn = 0.5 * (lo + hi);
        x = xn;
    }
    return xbest;
}

static inline void cross3(const double a[3], const double b[3], double c[3]) {
    c[0] = fma(a[1], b[2], -(a[2] * b[1]));
    c[1] = fma(a[2], b[0], -(a[0] * b[2]));
    c[2] = fma(a[0], b[1], -(a[1] * b[0]));
}

/* oriented epipolar constraint on the minimal sample: all five
 * (e2 x x2) . (E x1) must share one strict sign */
static int orientation_ok(const double E[9], const float (*s)[4], uint32_t ns) {
    const double c0[3] = {E[0], E[3], E[6]}, c1[3] = {E[1], E[4], E[7]}, c2[3] = {E[2], E[5], E[8]};
    double e01[3], e02[3], e12[3];
    cross3(c0, c1, e01);
    cross3(c0, c2, e02);
    cross3(c1, c2, e12);
    const double n01 = fma(e01[0], e01[0], fma(e01[1], e01[1], e01[2] * e01[2]));
    const double n02 = fma(e02[0], e02[0], fma(e02[1], e02[1], e02[2] * e02[2]));
    const double n12 = fma(e12[0], e12[0], fma(e12[1], e12[1], e12[2] * e12[2]));
    const double* ep = e01;
    double nb = n01;
    if (n02 > nb) { nb = n02; ep = e02; }
    if (n12 > nb) { nb = n12; ep = e12; }
    uint32_t npos = 0, nneg = 0;
    for (uint32_t i = 0; i < ns; ++i) {
        const double x1 = s[i][0], y1 = s[i][1], x2 = s[i][2], y2 = s[i][3];
        const double l0 = fma(E[0], x1, fma(E[1], y1, E[2]));
        const double l1 = fma(E[3], x1, fma(E[4], y1, E[5]));
        const double l2 = fma(E[6], x1, fma(E[7], y1, E[8]));
        /* c = ep x (x2,y2,1) */
        const double cx = fma(ep[1], 1.0, -(ep[2] * y2));
        const double cy = fma(ep[2], x2, -(ep[0] * 1.0));
        const double cz = fma(ep[0], y2, -(ep[1] * x2));
        const double sgn = fma(cx, l0, fma(cy, l1, cz * l2));
        npos += sgn > 0.0;
        nneg += sgn < 0.0;
    }
    return npos == ns || nneg == ns;
}

uint32_t pgo_backend(const double basis[36], const float (*sample)[4], uint32_t n_sample,
                     float models[PGO_MAX_MODELS][9], pgo_backend_dbg* dbg) {
    double cons[10][20];
    build_constraints(basis, cons);
    if (dbg) memcpy(dbg->cons, cons, sizeof cons);
    int prow[10];
    gauss_jordan(&cons[0][0], 10, 20, 10, prow);
    /* right block by pivot column */
    double red[10][10];
    for (int k = 0; k < 10; ++k)
        for (int m = 0; m < 10; ++m) red[k][m] = cons[prow[k]][10 + m];
    if (dbg) memcpy(dbg->red, red, sizeof red);
    /* B(z): rows from (e,f) = (4,5),(6,7),(8,9) */
    double bx[3][4], by[3][4], bc[3][5];
    for (int i = 0; i < 3; ++i) {
        const double* e = red[4 + 2 * i];
        const double* f = red[5 + 2 * i];
        bx[i][0] = e[2]; bx[i][1] = e[1] - f[2]; bx[i][2] = e[0] - f[1]; bx[i][3] = -f[0];
        by[i][0] = e[5]; by[i][1] = e[4] - f[5]; by[i][2] = e[3] - f[4]; by[i][3] = -f[3];
        bc[i][0] = e[9]; bc[i][1] = e[8] - f[9]; bc[i][2] = e[7] - f[8];
        bc[i][3] = e[6] - f[7]; bc[i][4] = -f[6];
    }
    /* det B = bc0*(bx1 by2 - by1 bx2) - bc1*(bx0 by2 - by0 bx2) + bc2*(bx0 by1 - by0 bx1) */
    static const int MN[3][2] = {{1, 2}, {0, 2}, {0, 1}};
    double T[3][11];
    for (int i = 0; i < 3; ++i) {
        const int r = MN[i][0], s = MN[i][1];
        double m[7] = {0, 0, 0, 0, 0, 0, 0};
        for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 4; ++b) m[a + b] = fma(bx[r][a], by[s][b], m[a + b]);
        for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 4; ++b) m[a + b] = fma(-by[r][a], bx[s][b], m[a + b]);
        for (int c = 0; c < 11; ++c) T[i][c] = 0.0;
        for (int a = 0; a < 5; ++a)
            for (int b = 0; b < 7; ++b) T[i][a + b] = fma(bc[i][a], m[b], T[i][a + b]);
    }
    double poly[11];
    for (int c = 0; c < 11; ++c) poly[c] = (T[0][c] - T[1][c]) + T[2][c];
    if (dbg) memcpy(dbg->poly, poly, sizeof poly);
    /* bracket sign changes on the fixed grid, refine */
    double roots[PGO_MAX_MODELS], rpoly[11];
    for (int c = 0; c < 11; ++c) rpoly[c] = poly[10 - c];
    uint32_t nr = 0;
    double gprev = grid_point(0);
    int sprev = horner10(poly, gprev) < 0.0;
    for (int j = 1; j <= PGO_GRID; ++j) {
        const double g = grid_point(j);
        const int s = horner10(poly, g) < 0.0;
        if (s != sprev && nr < PGO_MAX_MODELS) {
            if (gprev >= 1.0 || g <= -1.0) { /* |z|>1: refine w = 1/z on the reversed polynomial */
                const double w = refine_root(rpoly, 1.0 / g, 1.0 / gprev, s);
                roots[nr++] = 1.0 / w;
            } else {
                roots[nr++] = refine_root(poly, gprev, g, sprev);
            }
        }
        gprev = g;
        sprev = s;
    }
    if (dbg) {
        dbg->n_roots = nr;
        for (uint32_t i = 0; i < PGO_MAX_MODELS; ++i) dbg->roots[i] = i < nr ? roots[i] : 0.0;
    }
    /* back-substitute */
    uint32_t nm = 0;
    for (uint32_t ri = 0; ri < nr; ++ri) {
        const double z = roots[ri];
        double rw[3][3];
        for (int i = 0; i < 3; ++i) {
            rw[i][0] = fma(fma(fma(bx[i][3], z, bx[i][2]), z, bx[i][1]), z, bx[i][0]);
            rw[i][1] = fma(fma(fma(by[i][3], z, by[i][2]), z, by[i][1]), z, by[i][0]);
            rw[i][2] = fma(fma(fma(fma(bc[i][4], z, bc[i][3]), z, bc[i][2]), z, bc[i][1]), z, bc[i][0]);
        }
        double c01[3], c02[3], c12[3];
        cross3(rw[0], rw[1], c01);
        cross3(rw[0], rw[2], c02);
        cross3(rw[1], rw[2], c12);
        const double* cb = c01;
        double wb = fabs(c01[2]);
        if (fabs(c02[2]) > wb) { wb = fabs(c02[2]); cb = c02; }
        if (fabs(c12[2]) > wb) { wb = fabs(c12[2]); cb = c12; }
        const double x = cb[0] / cb[2], y = cb[1] / cb[2];
        double E[9], n2 = 0.0;
        for (int m = 0; m < 9; ++m) {
            E[m] = fma(x, basis[m], fma(y, basis[9 + m], fma(z, basis[18 + m], basis[27 + m])));
            n2 = fma(E[m], E[m], n2);
        }
        if (!(n2 > 0.0) || !(n2 < 1.0e300)) continue; /* NaN / inf / zero */
        const double inv = 1.0 / sqrt(n2);
        for (int m = 0; m < 9; ++m) E[m] = E[m] * inv;
        if (n_sample && !orientation_ok(E, sample, n_sample)) continue;
        for (int m = 0; m < 9; ++m) models[nm][m] = (float)E[m];
        ++nm;
    }
    return nm;
}

uint32_t pgo_five_point(const float pts[5][4], float models[PGO_MAX_MODELS][9],
                        pgo_backend_dbg* dbg) {
    double basis[36];
    pgo_nullspace5(pts, basis);
    return pgo_backend(basis, pts, 5, models, dbg);
}

/* ---- n-point refit ------------------------------------------------------ */
/* Summands are rounded to multiples of 2^-34 ((t+M)-M with M = 1.5*2^18), so
 * the f64 sums are exact and independent of summation order as long as
 * |partial sums| < 2^18 (normalised image coordinates, |x|,|y| <~ 2). */
#define PGO_QMAGIC 393216.0
void pgo_normal_matrix(const float* x1, const float* y1, const float* x2, const float* y2,
                       const uint8_t* mask, uint32_t n, double A[81]) {
    double S[9][9];
    memset(S, 0, sizeof S);
    for (uint32_t p = 0; p < n; ++p) {
        if (mask && !mask[p]) continue;
        const float pt[4] = {x1[p], y1[p], x2[p], y2[p]};
        double a[9];
        epi_row(pt, a);
        for (int i = 0; i < 9; ++i)
            for (int j = i; j < 9; ++j) {
                double t = a[i] * a[j];
                t = (t + PGO_QMAGIC) - PGO_QMAGIC;
                S[i][j] = S[i][j] + t;
            }
    }
    for (int i = 0; i < 9; ++i)
        for (int j = 0; j < 9; ++j) A[9 * i + j] = (i <= j) ? S[i][j] : S[j][i];
}

/* cyclic Jacobi, tournament order: round r pairs {(r+m)%9,(r-m)%9}, m=1..4.
 * All four (c,s) come from A before the step; column phase (A<-AJ, V<-VJ),
 * then row phase (A<-J^T A). */
void pgo_jacobi9(double A[81], double V[81]) {
    for (int i = 0; i < 81; ++i) V[i] = 0.0;
    for (int i = 0; i < 9; ++i) V[10 * i] = 1.0;
    for (int sw = 0; sw < PGO_JACOBI9_SWEEPS; ++sw)
        for (int r = 0; r < 9; ++r) {
            int P[4], Q[4];
            double C[4], S[4];
            for (int m = 1; m <= 4; ++m) {
                int p = (r + m) % 9, q = (r + 9 - m) % 9;
                if (p > q) { int tq = p; p = q; q = tq; }
                P[m - 1] = p;
                Q[m - 1] = q;
                const double apq = A[9 * p + q];
                double c = 1.0, s = 0.0;
                if (apq != 0.0) {
                    /* t = sgn(al) be / (|al| + hypot(al,be)), c = d/r, s = sgn(al) be / r:
                     * the classic rotation with one division and two square roots */
                    const double al = A[10 * q] - A[10 * p], be = 2.0 * apq;
                    const double h = sqrt(fma(al, al, be * be));
                    const double d = fabs(al) + h;
                    const double r = sqrt(fma(d, d, be * be));
                    const double inv = 1.0 / r;
                    c = d * inv;
                    s = (al >= 0.0 ? be : -be) * inv;
                }
                C[m - 1] = c;
                S[m - 1] = s;
            }
            for (int m = 0; m < 4; ++m) {
                const int p = P[m], q = Q[m];
                const double c = C[m], s = S[m];
                for (int l = 0; l < 9; ++l) {
                    const double ap = A[9 * l + p], aq = A[9 * l + q];
                    A[9 * l + p] = fma(c, ap, -(s * aq));
                    A[9 * l + q] = fma(s, ap, c * aq);
                    const double vp = V[9 * l + p], vq = V[9 * l + q];
                    V[9 * l + p] = fma(c, vp, -(s * vq));
                    V[9 * l + q] = fma(s, vp, c * vq);
                }
            }
            for (int m = 0; m < 4; ++m) {
                const int p = P[m], q = Q[m];
                const double c = C[m], s = S[m];
                for (int l = 0; l < 9; ++l) {
                    const double ap = A[9 * p + l], aq = A[9 * q + l];
                    A[9 * p + l] = fma(c, ap, -(s * aq));
                    A[9 * q + l] = fma(s, ap, c * aq);
                }
            }
        }
}

/* basis = eigenvectors of the 4 smallest eigenvalues: W = smallest, then Z, Y, X */
void pgo_basis_from_eigen(const double A[81], const double V[81], double basis[36]) {
    int taken[9] = {0};
    for (int rank = 0; rank < 4; ++rank) {
        int bi = -1;
        for (int i = 0; i < 9; ++i) {
            if (taken[i]) continue;
            if (bi < 0 || A[10 * i] < A[10 * bi]) bi = i;
        }
        taken[bi] = 1;
        double* dst = basis + 9 * (3 - rank);
        for (int l = 0; l < 9; ++l) dst[l] = V[9 * l + bi];
    }
}

uint32_t pgo_npoint(const float* x1, const float* y1, const float* x2, const float* y2,
                    const uint8_t* mask, uint32_t n, float models[PGO_MAX_MODELS][9]) {
    double A[81], V[81], basis[36];
    pgo_normal_matrix(x1, y1, x2, y2, mask, n, A);
    pgo_jacobi9(A, V);
    pgo_basis_from_eigen(A, V, basis);
    return pgo_backend(basis, NULL, 0, models, NULL);
}

/* Linear refit of an inlier set: the eigenvector of the smallest eigenvalue of the 9x9 normal matrix (first minimum
 * wins ties) = the least-squares solution of the epipolar equations, scaled to unit Frobenius norm and rounded to
 * f32.  No projection onto the essential manifold here: the model is only scored (the Sampson distance is defined for
 * any 3x3 matrix) and, if it ends up as the final model, decomposed by the SVD of pose_utils.h:144-169, which takes
 * U and V of whatever matrix it is given.  Returns the number of models (0 or 1). */
uint32_t pgo_linear_refit(const float* x1, const float* y1, const float* x2, const float* y2, const uint8_t* mask,
                          uint32_t n, float model[9]) {
    double A[81], V[81];
    pgo_normal_matrix(x1, y1, x2, y2, mask, n, A);
    pgo_jacobi9(A, V);
    int bi = 0;
    for (int i = 1; i < 9; ++i)
        if (A[10 * i] < A[10 * bi]) bi = i;
    double e[9], n2 = 0.0;
    for (int l = 0; l < 9; ++l) e[l] = V[9 * l + bi];
    for (int m = 0; m < 9; ++m) n2 = fma(e[m], e[m], n2);
    if (!(n2 > 0.0)) return 0;
    const double inv = 1.0 / sqrt(n2);
    for (int m = 0; m < 9; ++m) model[m] = (float)(e[m] * inv);
    return 1;
}

/* ---- decomposition ------------------------------------------------------ */
/* one-sided Jacobi on the columns of G = E*V; sigma sorted descending;
 * U2 = U0 x U1 (det U = +1); det V fixed by flipping V2 (pose_utils.h:157-163) */
void pgo_svd3(const double E[9], double U[9], double S[3], double V[9]) {
    double G[9];
    memcpy(G, E, sizeof G);
    for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
    static const int PQ[3][2] = {{0, 1}, {0, 2}, {1, 2}};
    for (int sw = 0; sw < PGO_SVD3_SWEEPS; ++sw)
        for (int k = 0; k < 3; ++k) {
            const int p = PQ[k][0], q = PQ[k][1];
            const double al = fma(G[p], G[p], fma(G[3 + p], G[3 + p], G[6 + p] * G[6 + p]));
            const double be = fma(G[q], G[q], fma(G[3 + q], G[3 + q], G[6 + q] * G[6 + q]));
            const double ga = fma(G[p], G[q], fma(G[3 + p], G[3 + q], G[6 + p] * G[6 + q]));
            if (ga == 0.0) continue;
            /* same one-division rotation as pgo_jacobi9: al' = be - al, be' = 2 ga */
            const double da = be - al, db = 2.0 * ga;
            const double h = sqrt(fma(da, da, db * db));
            const double d = fabs(da) + h;
            const double r = sqrt(fma(d, d, db * db));
            const double inv = 1.0 / r;
            const double c = d * inv, s = (da >= 0.0 ? db : -db) * inv;
            for (int l = 0; l < 3; ++l) {
                const double gp = G[3 * l + p], gq = G[3 * l + q];
                G[3 * l + p] = fma(c, gp, -(s * gq));
                G[3 * l + q] = fma(s, gp, c * gq);
                const double vp = V[3 * l + p], vq = V[3 * l + q];
                V[3 * l + p] = fma(c, vp, -(s * vq));
                V[3 * l + q] = fma(s, vp, c * vq);
            }
        }
    double sg[3];
    for (int j = 0; j < 3; ++j)
        sg[j] = sqrt(fma(G[j], G[j], fma(G[3 + j], G[3 + j], G[6 + j] * G[6 + j])));
    int ord[3] = {0, 1, 2};
    /* stable selection sort, descending */
    for (int a = 0; a < 2; ++a)
        for (int b = a + 1; b < 3; ++b)
            if (sg[ord[b]] > sg[ord[a]]) { int tq = ord[a]; ord[a] = ord[b]; ord[b] = tq; }
    double Gs[9], Vs[9];
    for (int j = 0; j < 3; ++j) {
        S[j] = sg[ord[j]];
        for (int l = 0; l < 3; ++l) {
            Gs[3 * l + j] = G[3 * l + ord[j]];
            Vs[3 * l + j] = V[3 * l + ord[j]];
        }
    }
    for (int j = 0; j < 2; ++j) {
        const double inv = 1.0 / S[j];
        for (int l = 0; l < 3; ++l) U[3 * l + j] = Gs[3 * l + j] * inv;
    }
    const double u0[3] = {U[0], U[3], U[6]}, u1[3] = {U[1], U[4], U[7]};
    double u2[3];
    cross3(u0, u1, u2);
    U[2] = u2[0]; U[5] = u2[1]; U[8] = u2[2];
    const double v0[3] = {Vs[0], Vs[3], Vs[6]}, v1[3] = {Vs[1], Vs[4], Vs[7]};
    double vc[3];
    cross3(v0, v1, vc);
    const double dv = fma(vc[0], Vs[2], fma(vc[1], Vs[5], vc[2] * Vs[8]));
    if (dv < 0.0) { Vs[2] = -Vs[2]; Vs[5] = -Vs[5]; Vs[8] = -Vs[8]; }
    memcpy(V, Vs, sizeof Vs);
}

/* pose_utils.h:144-169 (R1 = U D V^T, R2 = U D^T V^T, t = U[:,2]) and
 * :172-252 (candidates 0:(R1,+t) 1:(R1,-t) 2:(R2,+t) 3:(R2,-t); vote; first
 * max wins).  Cheirality: depth signs from lambda2*x2 = lambda1*R*x1 + t
 * (the reference's raw projected.z<0 test on a sign-ambiguous homogeneous
 * point is NOT reproduced -- SURVEY §8a-9 landmine, DESIGN.md deviations). */
void pgo_decompose(const double E[9], const float* x1, const float* y1, const float* x2,
                   const float* y2, const uint8_t* mask, uint32_t n, int vote_all, double R[9],
                   double t[3], uint32_t votes[4], uint32_t* cand) {
    double U[9], S[3], V[9];
    pgo_svd3(E, U, S, V);
    double Rc[2][9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            const double a = U[3 * i + 1] * V[3 * j + 0]; /* U1 V0^T */
            const double b = U[3 * i + 0] * V[3 * j + 1]; /* U0 V1^T */
            const double c = U[3 * i + 2] * V[3 * j + 2]; /* U2 V2^T */
            Rc[0][3 * i + j] = (b - a) + c;
            Rc[1][3 * i + j] = (a - b) + c;
        }
    double tt[3] = {U[2], U[5], U[8]};
    const double tn = 1.0 / sqrt(fma(tt[0], tt[0], fma(tt[1], tt[1], tt[2] * tt[2])));
    for (int i = 0; i < 3; ++i) tt[i] = tt[i] * tn;
    votes[0] = votes[1] = votes[2] = votes[3] = 0;
    for (uint32_t p = 0; p < n; ++p) {
        if (!vote_all && mask && !mask[p]) continue;
        const double X1[3] = {x1[p], y1[p], 1.0}, X2[3] = {x2[p], y2[p], 1.0};
        double x2t[3];
        cross3(X2, tt, x2t);
        for (int rI = 0; rI < 2; ++rI) {
            const double* Rm = Rc[rI];
            double a[3], nn[3], at[3];
            for (int i = 0; i < 3; ++i)
                a[i] = fma(Rm[3 * i], X1[0], fma(Rm[3 * i + 1], X1[1], Rm[3 * i + 2]));
            cross3(a, X2, nn);
            cross3(a, tt, at);
            const double d1 = fma(x2t[0], nn[0], fma(x2t[1], nn[1], x2t[2] * nn[2]));
            const double d2 = fma(at[0], nn[0], fma(at[1], nn[1], at[2] * nn[2]));
            votes[2 * rI + 0] += (d1 > 0.0) && (d2 > 0.0);
            votes[2 * rI + 1] += (d1 < 0.0) && (d2 < 0.0);
        }
    }
    uint32_t best = 0;
    for (uint32_t c = 1; c < 4; ++c)
        if (votes[c] > votes[best]) best = c;
    *cand = best;
    memcpy(R, Rc[best >> 1], 9 * sizeof(double));
    for (int i = 0; i < 3; ++i) t[i] = (best & 1) ? -tt[i] : tt[i];
}

/* ======================================================================= */
/* LITERAL restatement of the reference's candidate selection: */
/* pose::getPoseFromEssentialMatrix pose_utils.h:172-252 with                    */
/* pose::linearTriangulation pose_utils.h:491-506 and decomposeEssentialMatrix    */
/* pose_utils.h:144-169.  NOT what the product computes (that is pgo_decompose:   */
/* depth signs, inlier rows) -- this exists to MEASURE how the two rules differ.  */
/* ======================================================================= */
/* Right singular vectors by one-sided (Hestenes) Jacobi: A (n x n, row-major) V = U S, columns of V sorted by
 * decreasing singular value like Eigen::JacobiSVD.  V is a product of plane rotations started from the identity plus
 * the final column permutation; its column SIGNS are this routine's own -- Eigen's are not reproducible here. */
static void jacobi_right_vectors(const double* A, int n, double* V, double* sv) {
    double B[16];
    for (int i = 0; i < n * n; ++i) B[i] = A[i];
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) V[n * i + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 40; ++sweep) {
        int rotated = 0;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                double al = 0, be = 0, ga = 0;
                for (int k = 0; k < n; ++k) {
                    al += B[n * k + p] * B[n * k + p];
                    be += B[n * k + q] * B[n * k + q];
                    ga += B[n * k + p] * B[n * k + q];
                }
                if (fabs(ga) <= 1e-17 * sqrt(al * be) || ga == 0.0) continue;
                rotated = 1;
                const double zeta = (be - al) / (2.0 * ga);
                const double tt = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + tt * tt), sn = c * tt;
                for (int k = 0; k < n; ++k) {
                    const double bp = B[n * k + p], bq = B[n * k + q];
                    B[n * k + p] = c * bp - sn * bq;
                    B[n * k + q] = sn * bp + c * bq;
                    const double vp = V[n * k + p], vq = V[n * k + q];
                    V[n * k + p] = c * vp - sn * vq;
                    V[n * k + q] = sn * vp + c * vq;
                }
            }
        if (!rotated) break;
    }
    for (int j = 0; j < n; ++j) {
        double s2 = 0;
        for (int k = 0; k < n; ++k) s2 += B[n * k + j] * B[n * k + j];
        sv[j] = sqrt(s2);
    }
    for (int a = 0; a < n - 1; ++a) { /* selection sort, descending; swap columns of V along */
        int m = a;
        for (int b = a + 1; b < n; ++b)
            if (sv[b] > sv[m]) m = b;
        if (m != a) {
            const double ts = sv[a]; sv[a] = sv[m]; sv[m] = ts;
            for (int k = 0; k < n; ++k) {
                const double tv = V[n * k + a]; V[n * k + a] = V[n * k + m]; V[n * k + m] = tv;
            }
        }
    }
}

/* pose_utils.h:491-506: rows x*P3-P1, y*P3-P2 for both views; null vector = last column of V (sign NOT normalised) */
void pgo_ref_linear_triangulation(const double P1[12], const double P2[12], const double pt[4], double X[4]) {
    double D[16], V[16], sv[4];
    for (int c = 0; c < 4; ++c) {
        D[0 + c] = pt[0] * P1[8 + c] - P1[0 + c];
        D[4 + c] = pt[1] * P1[8 + c] - P1[4 + c];
        D[8 + c] = pt[2] * P2[8 + c] - P2[0 + c];
        D[12 + c] = pt[3] * P2[8 + c] - P2[4 + c];
    }
    jacobi_right_vectors(D, 4, V, sv);
    for (int k = 0; k < 4; ++k) X[k] = V[4 * k + 3];
}

static double det3(const double* M) {
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

/* pose_utils.h:144-169, literally: full U, V of E; last column of U / V negated when det < 0;
 * R1 = U d V^T, R2 = U d^T V^T with d = [0 1 0; -1 0 0; 0 0 1]; t = U.col(2).normalized() */
void pgo_ref_decompose_essential(const double E[9], double R1[9], double R2[9], double t[3]) {
    double U[9], V[9], sv[3], Et[9];
    jacobi_right_vectors(E, 3, V, sv);           /* E = U S V^T: V from E ...            */
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Et[3 * i + j] = E[3 * j + i];
    /* ... and U = E V S^-1 for the two non-zero singular values, third column = U0 x U1 (any full U does: the
     * reference then fixes det U = +1 by flipping exactly that column) */
    for (int j = 0; j < 2; ++j)
        for (int i = 0; i < 3; ++i) {
            double a = 0;
            for (int k = 0; k < 3; ++k) a += E[3 * i + k] * V[3 * k + j];
            U[3 * i + j] = a / sv[j];
        }
    U[2] = U[3] * U[7] - U[6] * U[4];
    U[5] = U[6] * U[1] - U[0] * U[7];
    U[8] = U[0] * U[4] - U[3] * U[1];
    (void)Et;
    if (det3(U) < 0) { U[2] = -U[2]; U[5] = -U[5]; U[8] = -U[8]; }
    if (det3(V) < 0) { V[2] = -V[2]; V[5] = -V[5]; V[8] = -V[8]; }
    static const double d[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1};
    double Ud[9], Udt[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double a = 0, b = 0;
            for (int k = 0; k < 3; ++k) {
                a += U[3 * i + k] * d[3 * k + j];
                b += U[3 * i + k] * d[3 * j + k];
            }
            Ud[3 * i + j] = a;
            Udt[3 * i + j] = b;
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double a = 0, b = 0;
            for (int k = 0; k < 3; ++k) {
                a += Ud[3 * i + k] * V[3 * j + k];
                b += Udt[3 * i + k] * V[3 * j + k];
            }
            R1[3 * i + j] = a;
            R2[3 * i + j] = b;
        }
    const double nn = sqrt(U[2] * U[2] + U[5] * U[5] + U[8] * U[8]);
    t[0] = U[2] / nn; t[1] = U[5] / nn; t[2] = U[8] / nn;
}

/* pose_utils.h:172-252.  For each of the 4 candidates (R1,+t) (R1,-t) (R2,+t) (R2,-t) and EVERY row: DLT point, skip
 * when raw projected z < 0 in view 1 or view 2 (:208-216; no division by w), squared reprojection error in both
 * views (:223-224), strict-< arg-min per row (:226-230), one vote per row (:237-239), first maximum wins (:242-245).
 * The raw-z test depends on the SIGN of the homogeneous null vector, which in the reference is whatever Eigen's
 * JacobiSVD returns.  All three possibilities are evaluated from the same null vectors:
 *   mode 0 "raw"  : the sign this file's Jacobi produces            (one arbitrary convention)
 *   mode 1 "w>=0" : null vector scaled so that w >= 0                (then z<0 IS the cheirality test)
 *   mode 2 "w<=0" : the opposite sign                                 (then points BEHIND both cameras pass)
 * out_R[m], out_t[m], out_votes[m][4], out_cand[m] for m = 0..2.  Returns the vote count of mode 0's winner (:251). */
int pgo_ref_pose_from_essential(const double E[9], const double* corr_aos, uint32_t n, double out_R[3][9],
                                double out_t[3][3], uint32_t out_votes[3][4], uint32_t out_cand[3]) {
    double R1[9], R2[9], t[3];
    pgo_ref_decompose_essential(E, R1, R2, t);
    const double* rot[4] = {R1, R1, R2, R2};
    const double P1[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    double* bestd = (double*)malloc((size_t)n * 3 * sizeof(double));
    int* bestp = (int*)malloc((size_t)n * 3 * sizeof(int));
    for (size_t k = 0; k < (size_t)n * 3; ++k) { bestd[k] = DBL_MAX; bestp[k] = 5; }
    for (int i = 0; i < 4; ++i) {
        double P2[12];
        const double sg = (i % 2) ? -1.0 : 1.0;
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) P2[4 * r + c] = rot[i][3 * r + c];
            P2[4 * r + 3] = sg * t[r];
        }
        for (uint32_t p = 0; p < n; ++p) {
            const double* pt = corr_aos + 4 * (size_t)p;
            double X[4];
            pgo_ref_linear_triangulation(P1, P2, pt, X);
            for (int m = 0; m < 3; ++m) {
                double Y[4];
                const double f = (m == 0) ? 1.0 : (m == 1) ? (X[3] < 0 ? -1.0 : 1.0) : (X[3] > 0 ? -1.0 : 1.0);
                for (int k = 0; k < 4; ++k) Y[k] = f * X[k];
                const double p1[3] = {Y[0], Y[1], Y[2]}; /* proj_1 = [I | 0] */
                if (p1[2] < 0) continue;
                double p2[3];
                for (int r = 0; r < 3; ++r) p2[r] = P2[4 * r] * Y[0] + P2[4 * r + 1] * Y[1] + P2[4 * r + 2] * Y[2] + P2[4 * r + 3] * Y[3];
                if (p2[2] < 0) continue;
                const double e1x = p1[0] / p1[2] - pt[0], e1y = p1[1] / p1[2] - pt[1];
                const double e2x = p2[0] / p2[2] - pt[2], e2y = p2[1] / p2[2] - pt[3];
                const double err = (e1x * e1x + e1y * e1y) + (e2x * e2x + e2y * e2y);
                if (err < bestd[3 * (size_t)p + m]) {
                    bestd[3 * (size_t)p + m] = err;
                    bestp[3 * (size_t)p + m] = i;
                }
            }
        }
    }
    for (int m = 0; m < 3; ++m) {
        uint32_t* v = out_votes[m];
        v[0] = v[1] = v[2] = v[3] = 0;
        for (uint32_t p = 0; p < n; ++p)
            if (bestp[3 * (size_t)p + m] < 5) ++v[bestp[3 * (size_t)p + m]];
        uint32_t best = 0;
        for (uint32_t c = 1; c < 4; ++c)
            if (v[c] > v[best]) best = c;
        out_cand[m] = best;
        memcpy(out_R[m], rot[best], 9 * sizeof(double));
        for (int k = 0; k < 3; ++k) out_t[m][k] = ((best % 2) ? -1.0 : 1.0) * t[k];
    }
    free(bestd);
    free(bestp);
    return (int)out_votes[0][out_cand[0]];
}

/* Agreement study (scripts/candidate_agreement.py): for every pair, the product's rule (pgo_decompose on the inlier
 * rows, what the HIP kernels reproduce bit for bit) next to the literal rule in its three sign conventions.
 * flags[p]: bit m (m=0..2): literal mode m chose the same rotation as the product; bit 4+m: the same translation sign;
 * bit 7: pair evaluated (edge OK); bit 8: the product's vote_all_rows=1 variant gives literal mode 1's (R, t);
 * bits 9/10/11: translation points the ground-truth way (t . t_gt > 0) for product(inliers) / product(all rows) /
 * literal mode 1.  t_gt: n_pairs x 3. */
void pgo_candidate_agreement_batch(const float* x1, const float* y1, const float* x2, const float* y2,
                                   const uint64_t* offsets, uint32_t n_pairs, const pgo_edge* edges, const uint8_t* masks,
                                   const double* t_gt, uint16_t* flags, int threads) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel for schedule(dynamic, 8)
    for (int64_t p = 0; p < (int64_t)n_pairs; ++p) {
        flags[p] = 0;
        if (edges[p].status != PGO_OK) continue;
        const uint64_t o = offsets[p];
        const uint32_t n = (uint32_t)(offsets[p + 1] - o);
        double* corr = (double*)malloc((size_t)n * 4 * sizeof(double));
        for (uint32_t i = 0; i < n; ++i) { /* the reference's N x 4 f64 matrix */
            corr[4 * i] = x1[o + i]; corr[4 * i + 1] = y1[o + i]; corr[4 * i + 2] = x2[o + i]; corr[4 * i + 3] = y2[o + i];
        }
        double R[3][9], t[3][3];
        uint32_t votes[3][4], cand[3];
        pgo_ref_pose_from_essential(edges[p].E, corr, n, R, t, votes, cand);
        uint16_t f = 0x80;
        for (int m = 0; m < 3; ++m) {
            double dR = 0, dt = 0;
            for (int k = 0; k < 9; ++k) dR = fmax(dR, fabs(R[m][k] - edges[p].R[k]));
            for (int k = 0; k < 3; ++k) dt = fmax(dt, fabs(t[m][k] - edges[p].t[k]));
            if (dR < 1e-6) f |= (uint16_t)(1u << m);
            if (dt < 1e-6) f |= (uint16_t)(1u << (4 + m));
        }
        double Ra[9], ta[3];
        uint32_t va[4], ca;
        pgo_decompose(edges[p].E, x1 + o, y1 + o, x2 + o, y2 + o, masks + o, n, 1, Ra, ta, va, &ca);
        double dRa = 0, dta = 0;
        for (int k = 0; k < 9; ++k) dRa = fmax(dRa, fabs(Ra[k] - R[1][k]));
        for (int k = 0; k < 3; ++k) dta = fmax(dta, fabs(ta[k] - t[1][k]));
        if (dRa < 1e-6 && dta < 1e-6) f |= 0x100;
        if (t_gt) {
            const double* g = t_gt + 3 * (size_t)p;
            if (edges[p].t[0] * g[0] + edges[p].t[1] * g[1] + edges[p].t[2] * g[2] > 0) f |= 0x200;
            if (ta[0] * g[0] + ta[1] * g[1] + ta[2] * g[2] > 0) f |= 0x400;
            if (t[1][0] * g[0] + t[1][1] * g[1] + t[1][2] * g[2] > 0) f |= 0x800;
        }
        flags[p] = f;
        free(corr);
    }
}

/* Unit-Frobenius f32 model from an f64 E, as the one-model scoring kernel (K2, pgi_score_pose_batch) prepares it:
 * n2 by an fma chain over the nine entries in order, one reciprocal square root, product rounded to f32. */
void pgo_model_from_essential(const double E[9], float e32[9]) {
    double n2 = 0.0;
    for (int c = 0; c < 9; ++c) n2 = fma(E[c], E[c], n2);
    const double inv = 1.0 / sqrt(n2);
    for (int c = 0; c < 9; ++c) e32[c] = (float)(E[c] * inv);
}

/* ---- robust estimator ---------------------------------------------------- */
static double pow_uint(double q, uint32_t k) { /* binary exponentiation, fixed order */
    double r = 1.0, b = q;
    while (k) {
        if (k & 1u) r = r * b;
        b = b * b;
        k >>= 1;
    }
    return r;
}

typedef struct {
    float E[9];
    uint32_t score, n_inl;
    int valid;
} best_t;

/* n-point refits while they improve (at most lo_iters) */
static void local_optimise(const float* x1, const float* y1, const float* x2, const float* y2,
                           uint32_t n, double thr, const pgo_params* prm, best_t* best,
                           uint8_t* mask, uint32_t* lo_runs) {
    const float thr2 = (float)(thr * thr);
    uint32_t* prev = NULL; /* graph-cut mode: the neighbourhood chains of the pair */
    if (prm->lo_graph_cut) {
        prev = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
        pgo_gc_chains(x1, y1, x2, y2, n, prev);
    }
    for (uint32_t it = 0; it < prm->lo_iters; ++it) {
        const uint32_t ni = prev ? pgo_gc_labels(best->E, x1, y1, x2, y2, n, thr, prm->lo_graph_cut < 255u ? prm->lo_graph_cut : 255u, prev, mask, NULL)
                                 : pgo_mask_model(best->E, x1, y1, x2, y2, n, thr2, mask);
        if (ni < 5) break;
        float models[PGO_MAX_MODELS][9];
        uint32_t nm;
        /* large inlier sets: the linear refit (smallest eigenvector of the normal matrix) is as accurate as the n-point
         * Nister refit and far cheaper; small ones keep the Nister refit, which tolerates the outliers inside a poor
         * model's inlier band better */
        if (prm->lo_linear_pct && (uint64_t)ni * 100u >= (uint64_t)n * prm->lo_linear_pct)
            nm = pgo_linear_refit(x1, y1, x2, y2, mask, n, models[0]);
        else
            nm = pgo_npoint(x1, y1, x2, y2, mask, n, models);
        ++*lo_runs;
        int improved = 0;
        uint32_t bs = best->score, bi = 0, bn = 0;
        for (uint32_t m = 0; m < nm; ++m) {
            uint32_t s, c;
            if (!pgo_preverify(models[m], x1, y1, x2, y2, n, thr, best->n_inl)) continue;
            pgo_score_model(models[m], x1, y1, x2, y2, n, thr, &s, &c);
            if (s > bs) { bs = s; bi = m; bn = c; improved = 1; }
        }
        if (!improved) break;
        memcpy(best->E, models[bi], sizeof best->E);
        best->score = bs;
        best->n_inl = bn;
    }
    free(prev);
}

void pgo_ransac_essential(const float* x1, const float* y1, const float* x2, const float* y2,
                          uint32_t n, double thr, const pgo_params* prm, uint64_t seed,
                          uint64_t pair_id, pgo_edge* out, uint8_t* mask) {
    memset(out, 0, sizeof *out);
    memset(mask, 0, n);
    if (n < 5) { out->status = PGO_FAIL_FEW_POINTS; return; }
    best_t best;
    memset(&best, 0, sizeof best);
    const uint32_t rs = prm->round_size ? prm->round_size : 32;
    const uint32_t budget = prm->fixed_budget ? prm->fixed_budget : prm->max_iters;
    uint32_t hyps = 0, lo_runs = 0;
    /* pre-verification bar of the NEXT round: inliers of the best right after a round's merge and
     * BEFORE its local optimisation (so hypotheses of round r+1 may start while LO of round r runs) */
    uint32_t n_bar = 0;
    while (hyps < budget) {
        /* one round: rs hypotheses, best by (score desc, hyp asc, root asc) */
        best_t rb;
        memset(&rb, 0, sizeof rb);
        for (uint32_t h = hyps; h < hyps + rs; ++h) {
            uint32_t idx[5];
            float pts[5][4], models[PGO_MAX_MODELS][9];
            pgo_sample5(seed, pair_id, h, prm->sampler ? pgo_progressive_rows(h, n, prm->max_iters) : n, idx);
            for (int k = 0; k < 5; ++k) {
                pts[k][0] = x1[idx[k]]; pts[k][1] = y1[idx[k]];
                pts[k][2] = x2[idx[k]]; pts[k][3] = y2[idx[k]];
            }
            const uint32_t nm = pgo_five_point(pts, models, NULL);
            for (uint32_t m = 0; m < nm; ++m) {
                uint32_t s, c;
                if (!pgo_preverify(models[m], x1, y1, x2, y2, n, thr, n_bar)) continue;
                /* only a model that beats the round's best so far AND the best of earlier rounds can change anything */
                long bar = rb.valid ? (long)rb.score : -1;
                if (best.valid && (long)best.score > bar) bar = (long)best.score;
                if (!score_model_bounded(models[m], x1, y1, x2, y2, n, thr, bar, &s, &c)) continue;
                if (!rb.valid || s > rb.score) {
                    memcpy(rb.E, models[m], sizeof rb.E);
                    rb.score = s; rb.n_inl = c; rb.valid = 1;
                }
            }
        }
        hyps += rs;
        if (rb.valid && (!best.valid || rb.score > best.score)) {
            best = rb;
            n_bar = best.n_inl;
            local_optimise(x1, y1, x2, y2, n, thr, prm, &best, mask, &lo_runs);
        } else {
            n_bar = best.valid ? best.n_inl : 0;
        }
        if (!prm->fixed_budget && best.valid && best.n_inl >= 5) {
            const double rho = (double)best.n_inl / (double)n;
            const double r5 = ((rho * rho) * (rho * rho)) * rho;
            const double q = 1.0 - r5;
            if (pow_uint(q, hyps) <= 1.0 - prm->confidence) break;
        }
    }
    out->iters = hyps;
    out->lo_runs = lo_runs;
    if (!best.valid) { out->status = PGO_FAIL_FEW_INLIERS; memset(mask, 0, n); return; }
    const float thr2 = (float)(thr * thr);
    out->n_inl = pgo_mask_model(best.E, x1, y1, x2, y2, n, thr2, mask);
    out->score = best.score;
    for (int m = 0; m < 9; ++m) out->E[m] = (double)best.E[m];
    out->status = out->n_inl >= prm->min_inliers ? PGO_OK : PGO_FAIL_FEW_INLIERS;
}

static int has_nan(const double* v, int n) {
    for (int i = 0; i < n; ++i)
        if (!(v[i] == v[i])) return 1;
    return 0;
}

/* pose_graph_builder.h:940-1078 */
void pgo_estimate_pose(const float* x1, const float* y1, const float* x2, const float* y2,
                       uint32_t n, double thr, const double* guess, const pgo_params* prm,
                       uint64_t seed, uint64_t pair_id, pgo_edge* out, uint8_t* mask) {
    memset(out, 0, sizeof *out);
    int success = 0;
    float Ebest[9];
    if (guess && n >= 5 && prm->guess_mode == 1) {
        /* Rotation-guided re-estimation (BASELINE config 5): the rotation of a chained pose is
         * metrically meaningful, its translation is not (unit per-edge baselines are composed), so keep R and
         * re-estimate the translation direction: every correspondence gives t . (p2 x R p1) = 0, two rows fix t.
         * One round of PGO_GUIDED_HYPS two-point hypotheses, scored like any model; the best one seeds the usual
         * local optimisation; accepted when it reaches min_inliers, otherwise the full robust fit runs from scratch. */
        const double* R = guess;
        const uint64_t base = pgo_mix64(seed ^ pgo_mix64(pair_id));
        best_t best;
        memset(&best, 0, sizeof best);
        for (uint32_t h = 0; h < PGO_GUIDED_HYPS; ++h) {
            const uint32_t i0 = pgo_draw_index(base, 0x40000000u + h, 0, n);
            uint32_t i1 = i0;
            for (uint32_t k = 1; k < 64 && i1 == i0; ++k) i1 = pgo_draw_index(base, 0x40000000u + h, k, n);
            double nv[2][3];
            const uint32_t id[2] = {i0, i1};
            for (int q = 0; q < 2; ++q) {
                const double X1[3] = {x1[id[q]], y1[id[q]], 1.0}, X2[3] = {x2[id[q]], y2[id[q]], 1.0};
                double a[3];
                for (int i = 0; i < 3; ++i) a[i] = fma(R[3 * i], X1[0], fma(R[3 * i + 1], X1[1], R[3 * i + 2]));
                cross3(X2, a, nv[q]);
            }
            double t[3];
            cross3(nv[0], nv[1], t);
            const double t2 = fma(t[0], t[0], fma(t[1], t[1], t[2] * t[2]));
            if (!(t2 > 1e-30)) continue;
            double Eg[9], n2 = 0.0;
            pgo_ref_essential_from_pose(R, t, Eg);
            for (int m = 0; m < 9; ++m) n2 = fma(Eg[m], Eg[m], n2);
            if (!(n2 > 0.0)) continue;
            const double inv = 1.0 / sqrt(n2);
            float Ef[9];
            for (int m = 0; m < 9; ++m) Ef[m] = (float)(Eg[m] * inv);
            uint32_t sc, c;
            pgo_score_model(Ef, x1, y1, x2, y2, n, thr, &sc, &c);
            if (!best.valid || sc > best.score) {
                memcpy(best.E, Ef, sizeof best.E);
                best.score = sc; best.n_inl = c; best.valid = 1;
            }
        }
        uint32_t lo_runs = 0;
        if (best.valid) local_optimise(x1, y1, x2, y2, n, thr, prm, &best, mask, &lo_runs);
        if (best.valid && best.n_inl >= prm->min_inliers) {
            const float thr2 = (float)(thr * thr);
            success = 2;
            out->n_inl = pgo_mask_model(best.E, x1, y1, x2, y2, n, thr2, mask);
            out->score = best.score;
            out->used_guess = 1;
            out->iters = PGO_GUIDED_HYPS;
            out->lo_runs = lo_runs;
            for (int m = 0; m < 9; ++m) out->E[m] = (double)best.E[m];
        }
    } else if (guess && n >= 5) { /* :974-1029 */
        double Eg[9], n2 = 0.0;
        pgo_ref_essential_from_pose(guess, guess + 9, Eg);
        for (int m = 0; m < 9; ++m) n2 = fma(Eg[m], Eg[m], n2);
        const double inv = 1.0 / sqrt(n2);
        float Ef[9];
        for (int m = 0; m < 9; ++m) Ef[m] = (float)(Eg[m] * inv);
        const double trunc = 1.5 * thr; /* :963-964 */
        const float tau2 = prm->guess_quirk ? (float)trunc : (float)(trunc * trunc);
        const uint32_t ni = pgo_mask_model(Ef, x1, y1, x2, y2, n, tau2, mask); /* :985-1009 */
        if (ni >= 5) {
            float models[PGO_MAX_MODELS][9]; /* :1013-1020: all-inlier refit */
            const uint32_t nm = pgo_npoint(x1, y1, x2, y2, mask, n, models);
            uint32_t bs = 0, bn = 0;
            int have = 0;
            for (uint32_t m = 0; m < nm; ++m) {
                uint32_t s, c;
                pgo_score_model(models[m], x1, y1, x2, y2, n, thr, &s, &c);
                if (!have || s > bs) { bs = s; bn = c; have = 1; memcpy(Ebest, models[m], sizeof Ebest); }
            }
            (void)bn;
            out->lo_runs = 1;
            if (have && ni >= prm->min_inliers) { /* :1022-1028 */
                success = 1;
                out->n_inl = ni; /* count of the guess's inliers */
                out->score = bs;
                out->used_guess = 1;
                for (int m = 0; m < 9; ++m) out->E[m] = (double)Ebest[m];
            }
        }
    }
    if (!success) { /* :1031-1055 */
        pgo_ransac_essential(x1, y1, x2, y2, n, thr, prm, seed, pair_id, out, mask);
        if (out->status != PGO_OK) return;
    }
    /* :1057-1075 */
    uint32_t votes[4], cand;
    pgo_decompose(out->E, x1, y1, x2, y2, mask, n, (int)prm->vote_all_rows, out->R, out->t,
                  votes, &cand);
    out->votes = votes[cand];
    out->cand = cand;
    out->status = (has_nan(out->R, 9) || has_nan(out->t, 3)) ? PGO_FAIL_NAN : PGO_OK;
    /* The record's E is the essential matrix OF THE RETURNED POSE: [t]x R (pose_utils.h:74-86) at unit Frobenius norm
     * (|t| = 1, so the norm of [t]x R is sqrt 2), with the sign of the fitted model.  The reference hands
     * getPoseFromEssentialMatrix a rank-2 E from cv::findEssentialMat (pose_graph_builder.h:1057-1066); a model from the
     * linear refit is only close to the manifold, and a caller that builds F from E (matcher.h:216-217) must not get it.
     * The mask refers to the fitted model (it is the model's inlier set); R, t are its decomposition. */
    if (out->status == PGO_OK) {
        double Ex[9], dot = 0.0;
        pgo_ref_essential_from_pose(out->R, out->t, Ex);
        for (int m = 0; m < 9; ++m) dot = fma(Ex[m], out->E[m], dot);
        const double sc = dot < 0.0 ? -0.70710678118654752440 : 0.70710678118654752440;
        for (int m = 0; m < 9; ++m) out->E[m] = Ex[m] * sc;
    }
}

int pgo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void pgo_estimate_pose_batch(const float* x1, const float* y1, const float* x2,
                             const float* y2, const uint64_t* off, uint32_t n_pairs,
                             const double* thr, const double* guesses,
                             const uint8_t* has_guess, const pgo_params* prm, uint64_t seed,
                             uint64_t pair_id_base, pgo_edge* out, uint8_t* masks,
                             int threads) {
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
#endif
    for (int64_t p = 0; p < (int64_t)n_pairs; ++p) {
        const uint64_t o = off[p];
        const uint32_t n = (uint32_t)(off[p + 1] - o);
        const double* g = (guesses && has_guess && has_guess[p]) ? guesses + 12 * p : NULL;
        pgo_estimate_pose(x1 + o, y1 + o, x2 + o, y2 + o, n, thr[p], g, prm, seed,
                          pair_id_base + (uint64_t)p, out + p, masks + o);
    }
    (void)threads;
}

/* ======================================================================= */
/* descriptor matching: feature_utils.h:135-202              */
/* ======================================================================= */
/* The reference runs two OpenCV brute-force kNN(2) searches (L2) and keeps query i iff
 * dist1 < 0.90*dist2 and the best match of its best match is i (:165-176), sorted by
 * dist1/dist2 (:178-180).  OpenCV is external, so the distance arithmetic is specified here:
 *   s_ij = fmaf chain over k = 0..D-1 of a_ik*b_jk (from 0)      [== v_mfma_f32_32x32x2_f32 accumulation]
 *   na_i, nb_j = the same chain of squares;  d2_ij = max(0, (na_i + nb_j) - 2*s_ij)
 * row-wise the two smallest d2 (ties: lower j), column-wise the smallest (ties: lower i);
 * dist = sqrtf(d2); test in double as the reference's `float < 0.90 * float`; ratio = dist1 / dist2 (float). */
static inline float chain_dot(const float* a, const float* b, uint32_t d) {
    float s = 0.0f;
    for (uint32_t k = 0; k < d; ++k) s = fmaf(a[k], b[k], s);
    return s;
}
uint32_t pgo_match_descriptors(const float* A, uint32_t k1, const float* B, uint32_t k2, uint32_t d,
                               uint32_t* out_i, uint32_t* out_j, double* out_ratio) {
    if (k1 < 2 || k2 < 2) return 0; /* :167-168: both directions need two neighbours */
    float* na = (float*)malloc(sizeof(float) * k1);
    float* nb = (float*)malloc(sizeof(float) * k2);
    float* r1 = (float*)malloc(sizeof(float) * k1);
    float* r2 = (float*)malloc(sizeof(float) * k1);
    uint32_t* rj = (uint32_t*)malloc(sizeof(uint32_t) * k1);
    float* c1 = (float*)malloc(sizeof(float) * k2);
    uint32_t* ci = (uint32_t*)malloc(sizeof(uint32_t) * k2);
    for (uint32_t i = 0; i < k1; ++i) na[i] = chain_dot(A + (size_t)i * d, A + (size_t)i * d, d);
    for (uint32_t j = 0; j < k2; ++j) { nb[j] = chain_dot(B + (size_t)j * d, B + (size_t)j * d, d); c1[j] = INFINITY; ci[j] = 0; }
    for (uint32_t i = 0; i < k1; ++i) {
        float b1 = INFINITY, b2 = INFINITY;
        uint32_t j1 = 0;
        for (uint32_t j = 0; j < k2; ++j) {
            const float s = chain_dot(A + (size_t)i * d, B + (size_t)j * d, d);
            const float t = na[i] + nb[j];
            float d2 = t - 2.0f * s;
            d2 = d2 > 0.0f ? d2 : 0.0f;
            if (d2 < b1) { b2 = b1; b1 = d2; j1 = j; }
            else if (d2 < b2) { b2 = d2; }
            if (d2 < c1[j]) { c1[j] = d2; ci[j] = i; }
        }
        r1[i] = b1; r2[i] = b2; rj[i] = j1;
    }
    uint32_t m = 0;
    for (uint32_t i = 0; i < k1; ++i) {
        const float dist1 = sqrtf(r1[i]), dist2 = sqrtf(r2[i]);
        if ((double)dist1 < 0.90 * (double)dist2 && ci[rj[i]] == i) {
            out_i[m] = i; out_j[m] = rj[i]; out_ratio[m] = (double)(dist1 / dist2);
            ++m;
        }
    }
    /* std::sort on (ratio, pointer) == (ratio, i): insertion sort is fine for test sizes */
    for (uint32_t a = 1; a < m; ++a) {
        const uint32_t vi = out_i[a], vj = out_j[a];
        const double vr = out_ratio[a];
        uint32_t b = a;
        while (b > 0 && (out_ratio[b - 1] > vr || (out_ratio[b - 1] == vr && out_i[b - 1] > vi))) {
            out_i[b] = out_i[b - 1]; out_j[b] = out_j[b - 1]; out_ratio[b] = out_ratio[b - 1];
            --b;
        }
        out_i[b] = vi; out_j[b] = vj; out_ratio[b] = vr;
    }
    free(na); free(nb); free(r1); free(r2); free(rj); free(c1); free(ci);
    return m;
}

/* ======================================================================= */
/* guided matching with a known pose: matcher.h:199-405      */
/* ======================================================================= */
/* F = K_dst^-T * E * K_src^-1 for pinhole K = [fx 0 cx; 0 fy cy; 0 0 1] (matcher.h:216-217), written out:
 * K^-1 = [1/fx 0 -cx/fx; 0 1/fy -cy/fy; 0 0 1].  G = E * K_src^-1 first, then F = K_dst^-T * G. */
void pgo_fundamental_from_essential(const double E[9], const double ks[4], const double kd[4], double F[9]) {
    double G[9];
    for (int r = 0; r < 3; ++r) {
        G[3 * r + 0] = E[3 * r + 0] / ks[0];
        G[3 * r + 1] = E[3 * r + 1] / ks[1];
        G[3 * r + 2] = (E[3 * r + 2] - G[3 * r + 0] * ks[2]) - G[3 * r + 1] * ks[3];
    }
    for (int c = 0; c < 3; ++c) {
        F[c] = G[c] / kd[0];
        F[3 + c] = G[3 + c] / kd[1];
        F[6 + c] = (G[6 + c] - F[c] * kd[2]) - F[3 + c] * kd[3];
    }
}

/* The candidate loop of HashingBasedMatcherWithPose::match (matcher.h:333-401).  The 45 angular bins of :258-301
 * only pre-select candidates; this restatement visits every destination keypoint in index order, which is the
 * loop the reference keeps commented at :327 -- every point the bins would offer plus the ones they miss.
 * Arithmetic follows the reference statement by statement: symmetric epipolar distance in pixels (:340-352), gate
 * at 0.75^2 (:354), f32 descriptor differences squared and summed in double (:359-365), "second best" = the best
 * before the last improvement (:367-372), the count-adapted ratio (:375-392).
 * Returns the number of matches, in source order; ratio[] = dist_ratio_sq_adapted. */
uint32_t pgo_guided_match(const double F[9], const float* kp1, uint32_t n1, const float* kp2, uint32_t n2,
                          const float* d1, const float* d2, uint32_t dim, uint32_t* out_i, uint32_t* out_j,
                          double* out_ratio) {
    const double e11 = F[0], e12 = F[1], e13 = F[2], e21 = F[3], e22 = F[4], e23 = F[5], e31 = F[6], e32 = F[7],
                 e33 = F[8];
    uint32_t m = 0;
    for (uint32_t i = 0; i < n1; ++i) {
        const double x1 = kp1[2 * i], y1 = kp1[2 * i + 1];
        double second = DBL_MAX, best = DBL_MAX;
        int best_index = -1;
        uint32_t count = 0;
        const double rx = (e11 * x1 + e12 * y1) + e13;
        const double ry = (e21 * x1 + e22 * y1) + e23;
        const double b1 = rx * rx + ry * ry;
        for (uint32_t j = 0; j < n2; ++j) {
            const double x2 = kp2[2 * j], y2 = kp2[2 * j + 1];
            const double rxc = (e11 * x2 + e21 * y2) + e31;
            const double ryc = (e12 * x2 + e22 * y2) + e32;
            const double rwc = (e13 * x2 + e23 * y2) + e33;
            const double r = (x1 * rxc + y1 * ryc) + rwc;
            const double a1 = rxc * rxc + ryc * ryc;
            const double dist = ((r * r) * (a1 + b1)) / (a1 * b1);
            if (dist >= 0.75 * 0.75) continue;
            ++count;
            double dd = 0.0;
            for (uint32_t k = 0; k < dim; ++k) {
                const float df = d1[(size_t)i * dim + k] - d2[(size_t)j * dim + k];
                const double dv = df;
                dd = dd + dv * dv;
            }
            if (dd < best) { second = best; best = dd; best_index = (int)j; }
        }
        double corr = 1.0;
        if (count < 20) corr = 0.65 * 0.65;
        if (count < 10) corr = 0.6 * 0.6;
        if (count < 5) corr = 0.5 * 0.5;
        if (count < 3) corr = 0.25 * 0.25;
        const double ratio = (best / second) / corr;
        if (ratio < 0.00001) continue;
        if (best_index > -1 && (ratio < 0.8 * 0.8 || count == 1)) {
            out_i[m] = i; out_j[m] = (uint32_t)best_index; out_ratio[m] = ratio;
            ++m;
        }
    }
    return m;
}

/* LITERAL epipolar hashing of HashingBasedMatcherWithPose::match (matcher.h:199-405, instantiated with 45 bins at
 * pose_graph_builder.h:738): destination keypoints are hashed by the angle of their epipolar line's normal in the
 * source image (:283-301), a source keypoint only meets the destination keypoints of ITS bin (:306-327), everything
 * after that is the candidate loop of pgo_guided_match.  Epipole = right null vector of F (:220-226, JacobiSVD
 * ComputeFullV; the sign cancels in the division by its third component); when it lies inside the source image the
 * angular range degenerates to min 180 / max 0 (:232-262), otherwise it comes from the destination image corners.
 * atan2 / round are libm's, as in the reference.  size_src / size_dst: cv::Size (integers) of the two images.
 * fragile[i] (optional, n1 bytes): 1 when source i's bin, or the bin of a destination keypoint that passes i's
 * epipolar gate, lies within 1e-7 of a rounding boundary -- a result that a different libm / SVD may flip. */
uint32_t pgo_ref_guided_match_binned(const double F[9], const float* kp1, uint32_t n1, const float* kp2, uint32_t n2,
                                     const float* d1, const float* d2, uint32_t dim, const int size_src[2],
                                     const int size_dst[2], int n_bins, uint32_t* out_i, uint32_t* out_j, double* out_ratio,
                                     uint8_t* fragile) {
    const double kRadianToDegree = 180.0 / 3.14159265358979323846; /* M_PI */
    const double e11 = F[0], e12 = F[1], e13 = F[2], e21 = F[3], e22 = F[4], e23 = F[5], e31 = F[6], e32 = F[7],
                 e33 = F[8];
    double V[9], sv[3];
    jacobi_right_vectors(F, 3, V, sv);
    const double ep0 = V[2] / V[8], ep1 = V[5] / V[8];
    const int in_image = ep0 >= 0 && ep0 < size_src[0] && ep1 >= 0 && ep1 < size_src[1];
    double min_angle = 180, max_angle = 0;
    if (!in_image) {
        const double corner[8] = {0, 0, (double)size_dst[0], 0, (double)size_dst[0], (double)size_dst[1], 0, (double)size_dst[1]};
        for (int c = 0; c < 8; c += 2) {
            const double x = corner[c], y = corner[c + 1];
            const double nx = e11 * x + e21 * y + e31;
            const double ny = e12 * x + e22 * y + e32;
            double angle = kRadianToDegree * atan2(ny, nx) + 180.0;
            if (angle > 180) angle -= 180;
            min_angle = angle < min_angle ? angle : min_angle;
            max_angle = angle > max_angle ? angle : max_angle;
        }
    }
    const double range = max_angle - min_angle;
    const int bins = n_bins > 0 ? n_bins : (int)range;
    int* bin_of = (int*)malloc(((size_t)n2 + 1) * sizeof(int));
    uint8_t* frag2 = (uint8_t*)calloc((size_t)n2 + 1, 1);
    for (uint32_t j = 0; j < n2; ++j) {
        const double px = kp2[2 * j], py = kp2[2 * j + 1];
        const double nx = e11 * px + e21 * py + e31;
        const double ny = e12 * px + e22 * py + e32;
        double angle = kRadianToDegree * atan2(ny, nx) + 180.0;
        if (angle > 180) angle -= 180;
        angle = (bins - 1) * (angle - min_angle) / range;
        const int b = (int)round(angle);
        bin_of[j] = b < 0 ? 0 : (b > bins - 1 ? bins - 1 : b);
        frag2[j] = !(fabs(fabs(angle - floor(angle)) - 0.5) > 1e-7);  /* also catches NaN */
    }
    uint32_t m = 0;
    for (uint32_t i = 0; i < n1; ++i) {
        const double px = kp1[2 * i], py = kp1[2 * i + 1];
        const double vx = px - ep0, vy = py - ep1;
        const double nx = -vy, ny = vx;
        double angle = kRadianToDegree * atan2(ny, nx) + 180.0;
        if (angle > 180) angle -= 180;
        angle = (bins - 1) * (angle - min_angle) / range;
        int bin = (int)round(angle);
        bin = bin < 0 ? 0 : (bin > bins - 1 ? bins - 1 : bin);
        int frag = !(fabs(fabs(angle - floor(angle)) - 0.5) > 1e-7);
        double second = DBL_MAX, best = DBL_MAX;
        int best_index = -1;
        uint32_t count = 0;
        const double x1 = px, y1 = py;
        const double rx = (e11 * x1 + e12 * y1) + e13;
        const double ry = (e21 * x1 + e22 * y1) + e23;
        const double b1 = rx * rx + ry * ry;
        for (uint32_t j = 0; j < n2; ++j) {
            if (bin_of[j] != bin && !frag2[j]) continue;
            const double x2 = kp2[2 * j], y2 = kp2[2 * j + 1];
            const double rxc = (e11 * x2 + e21 * y2) + e31;
            const double ryc = (e12 * x2 + e22 * y2) + e32;
            const double rwc = (e13 * x2 + e23 * y2) + e33;
            const double r = (x1 * rxc + y1 * ryc) + rwc;
            const double a1 = rxc * rxc + ryc * ryc;
            const double dist = ((r * r) * (a1 + b1)) / (a1 * b1);
            if (dist >= 0.75 * 0.75) continue;
            if (frag2[j]) frag = 1;       /* a borderline destination keypoint competes (or would compete) here */
            if (bin_of[j] != bin) continue;
            ++count;
            double dd = 0.0;
            for (uint32_t k = 0; k < dim; ++k) {
                const float df = d1[(size_t)i * dim + k] - d2[(size_t)j * dim + k];
                const double dv = df;
                dd = dd + dv * dv;
            }
            if (dd < best) { second = best; best = dd; best_index = (int)j; }
        }
        if (fragile) fragile[i] = (uint8_t)frag;
        double corr = 1.0;
        if (count < 20) corr = 0.65 * 0.65;
        if (count < 10) corr = 0.6 * 0.6;
        if (count < 5) corr = 0.5 * 0.5;
        if (count < 3) corr = 0.25 * 0.25;
        const double ratio = (best / second) / corr;
        if (ratio < 0.00001) continue;
        if (best_index > -1 && (ratio < 0.8 * 0.8 || count == 1)) {
            out_i[m] = i; out_j[m] = (uint32_t)best_index; out_ratio[m] = ratio;
            ++m;
        }
    }
    free(bin_of);
    free(frag2);
    return m;
}
