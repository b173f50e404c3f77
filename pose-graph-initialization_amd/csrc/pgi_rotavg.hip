// pgi_rotavg.hip -- rotation averaging (L1 + IRLS) on the pose graph.
//
// NOT in the reference: BASELINE.json:north_star adds it downstream of the
// per-edge estimator.  Algorithm: Chatterjee & Govindu (ICCV'13 / TPAMI'18), specified in
// oracle/rotavg_oracle.py and DESIGN.md §3.7.  Edge convention of the reference
// (pose.h:14, graph_traversal.h:340-344): R_rel ~ R_dst R_src^T, world->camera R_k.
//
// Data path: the problem is MB-scale (V <= ~1e4 views) and latency-bound, so it is solved by
//   rot_residual_kernel  one thread per edge   : omega = log(R_dst^T R_rel R_src), robust weight
//   rot_solve_kernel     V <= 64: ONE 1024-thread workgroup, a single launch : weighted-Laplacian normal
//                        equations assembled per vertex from a CSR adjacency (fixed order), Jacobi-
//                        preconditioned CG on the three axes at once; fixed-tree block reductions
//   cg_*_kernel          larger graphs: the same recurrences across all CUs, ONE launch per iteration (16 lanes per
//                        view), CG scalars and the convergence flag resident on the device, the host looks after a
//                        predicted number of launches; deterministic
//   rot_solve_tree_kernel  sparse, sequence-like graphs (E <= 8 V): spanning-tree preconditioner, on chip
//   cg2_*_kernel         dense band-like graphs (deep breadth-first walk): two-level preconditioner, Jacobi + a coarse
//                        space of <= 127 aggregates of neighbouring views, coarse matrix inverted on the device per step
//   rot_update_kernel    one thread per view   : R_k <- R_k exp(d_k)
// Multi-GPU: "replicas only" -- after the all-gather of the edge records every rank (or rank 0)
// runs this identical solve; an edge-partitioned CG would pay an all-reduce per iteration for a
// 3V-vector and is pure latency.
#include "pgi_internal.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstddef>
#include <cstring>
#include <numeric>
#include <vector>

namespace pgi {
// Inner solves of the rotation averaging stop at |r|_M <= 1e-4 |r0|_M (round 5; 1e-10 before).  The outer IRLS loop is a fixed-point
// iteration whose steps shrink 3-5x each, and the preconditioned systems have an effective condition of ~40 (75 iterations for ten
// orders of magnitude), so a relative residual of 1e-4 leaves ~6e-4 of a step as error -- two orders below the next step.  Measured
// (scripts/rotavg_bench.py + soak_rotavg.py, 120 random graphs x 2 solver settings): 1e-10 / 1e-6 / 1e-4 / 1e-3 -> V = 5000: 12.2 / 11.0 /
// 9.2 / 8.0 ms, band graph (config 4's shape) 19.7 / 14.6 / 12.1 / 11.1 ms; outer iteration counts and the worst difference from the
// oracle's direct solves (3.4e-7 rad) are THE SAME down to 1e-4; at 1e-3 the first iteration counts move by one.
constexpr double kInnerTolerance = 1e-4;


#define RDEV __device__ __forceinline__
constexpr uint32_t kSingleWgViews = 64;  // at or below (16 lanes per view in one pass): one-workgroup PCG in a single launch

RDEV void mat3_mul(const double* A, const double* B, double* C) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
RDEV void mat3_tmul(const double* A, const double* B, double* C) {  // A^T B
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
}

// log map through the unit quaternion (Shepperd's branch selection): robust up to pi
RDEV void so3_log(const double* R, double* w) {
    const double tr = R[0] + R[4] + R[8];
    double q0, q1, q2, q3;  // w, x, y, z
    if (tr > 0.0) {
        const double s = sqrt(tr + 1.0) * 2.0;
        q0 = 0.25 * s; q1 = (R[7] - R[5]) / s; q2 = (R[2] - R[6]) / s; q3 = (R[3] - R[1]) / s;
    } else if (R[0] > R[4] && R[0] > R[8]) {
        const double s = sqrt(1.0 + R[0] - R[4] - R[8]) * 2.0;
        q0 = (R[7] - R[5]) / s; q1 = 0.25 * s; q2 = (R[1] + R[3]) / s; q3 = (R[2] + R[6]) / s;
    } else if (R[4] > R[8]) {
        const double s = sqrt(1.0 + R[4] - R[0] - R[8]) * 2.0;
        q0 = (R[2] - R[6]) / s; q1 = (R[1] + R[3]) / s; q2 = 0.25 * s; q3 = (R[5] + R[7]) / s;
    } else {
        const double s = sqrt(1.0 + R[8] - R[0] - R[4]) * 2.0;
        q0 = (R[3] - R[1]) / s; q1 = (R[2] + R[6]) / s; q2 = (R[5] + R[7]) / s; q3 = 0.25 * s;
    }
    if (q0 < 0.0) { q0 = -q0; q1 = -q1; q2 = -q2; q3 = -q3; }
    const double nv = sqrt(q1 * q1 + q2 * q2 + q3 * q3);
    const double k = (nv < 1e-12) ? 2.0 : 2.0 * atan2(nv, q0) / nv;
    w[0] = k * q1; w[1] = k * q2; w[2] = k * q3;
}

RDEV void so3_exp(const double* w, double* R) {
    const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    const double t = sqrt(t2);
    double a, b;  // sin t / t, (1 - cos t) / t^2
    if (t < 1e-6) {
        a = 1.0 - t2 / 6.0;
        b = 0.5 - t2 / 24.0;
    } else {
        a = sin(t) / t;
        b = (1.0 - cos(t)) / t2;
    }
    const double x = w[0], y = w[1], z = w[2];
    R[0] = 1.0 - b * (y * y + z * z); R[1] = -a * z + b * x * y;        R[2] = a * y + b * x * z;
    R[3] = a * z + b * x * y;         R[4] = 1.0 - b * (x * x + z * z); R[5] = -a * x + b * y * z;
    R[6] = -a * y + b * x * z;        R[7] = a * x + b * y * z;         R[8] = 1.0 - b * (x * x + y * y);
}

struct RotEdgeDev {
    uint32_t src, dst;
    double R[9];
    double weight;
};

__global__ __launch_bounds__(256) void rot_residual_kernel(const RotEdgeDev* __restrict__ edges, uint32_t n_edges,
                                                           const double* __restrict__ R, int l1_phase, double sigma,
                                                           double* __restrict__ omega, double* __restrict__ w) {
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n_edges) return;
    const RotEdgeDev ed = edges[e];
    double Ri[9], Rj[9], T[9], D[9], om[3];
#pragma unroll
    for (int c = 0; c < 9; ++c) {
        Ri[c] = R[9 * (size_t)ed.src + c];
        Rj[c] = R[9 * (size_t)ed.dst + c];
    }
    mat3_mul(ed.R, Ri, T);   // R_rel R_src
    mat3_tmul(Rj, T, D);     // R_dst^T R_rel R_src
    so3_log(D, om);
    const double n2 = om[0] * om[0] + om[1] * om[1] + om[2] * om[2];
    double wt;
    if (l1_phase) {
        wt = ed.weight / fmax(sqrt(n2), 1e-4);
    } else {
        const double s2 = sigma * sigma, d = n2 + s2;
        wt = ed.weight * s2 / (d * d) * s2;
    }
    omega[3 * (size_t)e + 0] = om[0];
    omega[3 * (size_t)e + 1] = om[1];
    omega[3 * (size_t)e + 2] = om[2];
    w[e] = wt;
}

// block-wide sum of three doubles, fixed tree order (deterministic)
__device__ void block_sum3(double v[3], double* red /* 3*1024 */, int tid) {
    red[tid] = v[0];
    red[1024 + tid] = v[1];
    red[2048 + tid] = v[2];
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (tid < s) {
            red[tid] += red[tid + s];
            red[1024 + tid] += red[1024 + tid + s];
            red[2048 + tid] += red[2048 + tid + s];
        }
        __syncthreads();
    }
    v[0] = red[0];
    v[1] = red[1024];
    v[2] = red[2048];
    __syncthreads();
}

// adj_ptr[V+1]; adj_edge / adj_other / adj_sign per incidence (sign +1: the vertex is dst, -1: src)
__global__ __launch_bounds__(1024) void rot_solve_kernel(uint32_t n_views, const uint32_t* __restrict__ adj_ptr,
                                                         const uint32_t* __restrict__ adj_edge,
                                                         const uint32_t* __restrict__ adj_other,
                                                         const int8_t* __restrict__ adj_sign,
                                                         const uint8_t* __restrict__ is_root,
                                                         const double* __restrict__ omega, const double* __restrict__ w,
                                                         uint32_t cg_iters, double cg_tol, uint32_t row_lanes, double* __restrict__ diag,
                                                         double* __restrict__ x, double* __restrict__ r,
                                                         double* __restrict__ p, double* __restrict__ Ap,
                                                         double* __restrict__ stats /* mean|d|, cg iterations */) {
    __shared__ double red[3 * 1024];
    const int tid = threadIdx.x;
    // assemble: diag_k = sum w_e ; b_k = sum sign * w_e * omega_e ; x = 0, r = b, z = r / diag, p = z
    double rz[3] = {0, 0, 0};
    for (uint32_t k = tid; k < n_views; k += 1024) {
        double d = 0, b[3] = {0, 0, 0};
        if (!is_root[k]) {
            for (uint32_t a = adj_ptr[k]; a < adj_ptr[k + 1]; ++a) {
                const uint32_t e = adj_edge[a];
                const double we = w[e], sg = (double)adj_sign[a];
                d += we;
                b[0] += sg * we * omega[3 * (size_t)e + 0];
                b[1] += sg * we * omega[3 * (size_t)e + 1];
                b[2] += sg * we * omega[3 * (size_t)e + 2];
            }
        }
        diag[k] = d;
        const double inv = d > 0.0 ? 1.0 / d : 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            x[3 * (size_t)k + c] = 0.0;
            r[3 * (size_t)k + c] = b[c];
            const double z = b[c] * inv;
            p[3 * (size_t)k + c] = z;
            rz[c] += b[c] * z;
        }
    }
    block_sum3(rz, red, tid);
    const double rz0[3] = {rz[0], rz[1], rz[2]};
    uint32_t it = 0;
    for (; it < cg_iters; ++it) {
        bool done = true;
#pragma unroll
        for (int c = 0; c < 3; ++c) done &= !(rz[c] > cg_tol * cg_tol * rz0[c]);
        if (done) break;  // uniform: rz is identical in every thread
        double pAp[3] = {0, 0, 0};
        // `row_lanes` lanes share a view (its incidences strided over them, partial rows combined by a fixed xor tree):
        // one thread per view walks tens of neighbours as a chain of dependent gathers, the whole cost of the product
        const uint32_t sub = (uint32_t)tid & (row_lanes - 1u), per_pass = 1024u / row_lanes;
        for (uint32_t k0 = 0; k0 < n_views; k0 += per_pass) {  // uniform trip count: the shuffles below need every lane
            const uint32_t k = k0 + (uint32_t)tid / row_lanes;
            const bool live = k < n_views, free_k = live && !is_root[k];
            double y[3] = {0, 0, 0};
            if (free_k)
                for (uint32_t a = adj_ptr[k] + sub; a < adj_ptr[k + 1]; a += row_lanes) {
                    const uint32_t o = adj_other[a];
                    if (is_root[o]) continue;
                    const double we = w[adj_edge[a]];
#pragma unroll
                    for (int c = 0; c < 3; ++c) y[c] -= we * p[3 * (size_t)o + c];
                }
            for (uint32_t m = 1; m < row_lanes; m <<= 1)
#pragma unroll
                for (int c = 0; c < 3; ++c) y[c] += __shfl_xor(y[c], (int)m);
            if (live && sub == 0u) {
                const double d = diag[k];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const double pk = p[3 * (size_t)k + c], yc = free_k ? d * pk + y[c] : 0.0;
                    Ap[3 * (size_t)k + c] = yc;
                    pAp[c] += pk * yc;
                }
            }
        }
        block_sum3(pAp, red, tid);
        double alpha[3], rzn[3] = {0, 0, 0};
#pragma unroll
        for (int c = 0; c < 3; ++c) alpha[c] = pAp[c] > 0.0 ? rz[c] / pAp[c] : 0.0;
        for (uint32_t k = tid; k < n_views; k += 1024) {
            const double d = diag[k], inv = d > 0.0 ? 1.0 / d : 0.0;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const size_t i = 3 * (size_t)k + c;
                x[i] += alpha[c] * p[i];
                const double rn = r[i] - alpha[c] * Ap[i];
                r[i] = rn;
                rzn[c] += rn * (rn * inv);
            }
        }
        block_sum3(rzn, red, tid);
        for (uint32_t k = tid; k < n_views; k += 1024) {
            const double d = diag[k], inv = d > 0.0 ? 1.0 / d : 0.0;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const size_t i = 3 * (size_t)k + c;
                const double beta = rz[c] > 0.0 ? rzn[c] / rz[c] : 0.0;
                p[i] = r[i] * inv + beta * p[i];
            }
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 3; ++c) rz[c] = rzn[c];
    }
    double nd[3] = {0, 0, 0};
    for (uint32_t k = tid; k < n_views; k += 1024)
        nd[0] += sqrt(x[3 * (size_t)k] * x[3 * (size_t)k] + x[3 * (size_t)k + 1] * x[3 * (size_t)k + 1] +
                      x[3 * (size_t)k + 2] * x[3 * (size_t)k + 2]);
    block_sum3(nd, red, tid);
    if (tid == 0) {
        stats[0] = nd[0] / (double)n_views;
        stats[1] = (double)it;
    }
}

// ---- on-chip PCG with a spanning-tree preconditioner (sparse / sequence-like view graphs, up to kTreeViews views) ------
// Jacobi-preconditioned CG needs thousands of iterations on path-like view graphs (image sequences: kappa ~ (V / degree)^2;
// 2 100 iterations per solve on the config-4 surrogate) where the Laplacian of the maximum-weight spanning forest -- the
// one the initialisation already built -- as preconditioner needs ~100 (and is useless on dense graphs: 1 400 against
// Jacobi's 19), so the host switches to this kernel when the Jacobi solve runs into its iteration cap.
// A tree system L_T z = r is solved EXACTLY by two prefix sums (no elimination order, no level-by-level sweep -- the
// forest of the surrogate is 1 800 levels deep): with the views numbered in depth-first preorder a subtree is a
// contiguous range, so the flow towards the root on the edge above view k is f_k = S[k + size_k - 1] - S[k - 1]
// (S = prefix sums of r), and z_k = sum over the ancestors-or-self u of f_u / w_u is a prefix sum over the Euler tour
// with +g_u at u's entry and -g_u at its exit.  One workgroup per component of the (Laplacian x I3) system keeps its
// vectors in registers (thread t owns views [t m, t m + m)), the scans and the gathered vector live in LDS; an iteration
// is a dozen workgroup barriers.  Scans run in a fixed order: same input, same bits.
constexpr uint32_t kTreeViews = 6144;
constexpr int kTreeOwn = kTreeViews / 1024;

__device__ inline double wave_sum_fixed(double v) {  // xor tree: the same order on every call
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
// sum over the 1024 threads, returned to all of them; red: 16 doubles of LDS
__device__ inline double block_sum_1024(double v, double* red, int tid) {
    v = wave_sum_fixed(v);
    __syncthreads();  // previous use of red is over
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += red[i];
    return s;
}
// exclusive prefix (in thread order) of one value per thread; red: 16 doubles of LDS
__device__ inline double block_prefix_1024(double v, double* red, int tid) {
    const int lane = tid & 63, wv = tid >> 6;
    double x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double y = __shfl_up(x, o);
        if (lane >= o) x += y;
    }
    __syncthreads();  // previous use of red is over
    if (lane == 63) red[wv] = x;
    __syncthreads();
    double before = 0.0;
    for (int i = 0; i < wv; ++i) before += red[i];
    return before + (x - v);
}

struct TreeIncidence {  // one 16-byte load per incidence in the products
    double w;
    uint32_t other, pad;
};
// grid = 3 (component), block = 1024.  Views are in depth-first preorder ("new" numbering); new_to_old maps back.
__global__ __launch_bounds__(1024) void rot_solve_tree_kernel(uint32_t n_views, uint32_t m /* views per thread */,
                                                              const uint32_t* __restrict__ adj_ptr, const uint32_t* __restrict__ adj_edge,
                                                              const uint32_t* __restrict__ adj_other, const int8_t* __restrict__ adj_sign,
                                                              const uint32_t* __restrict__ parent_edge /* 0xFFFFFFFF: root */,
                                                              const uint32_t* __restrict__ sub_size, const uint32_t* __restrict__ tour_enter,
                                                              const uint32_t* __restrict__ tour_exit, const uint32_t* __restrict__ new_to_old,
                                                              const double* __restrict__ omega, const double* __restrict__ w,
                                                              uint32_t cg_iters, double cg_tol, TreeIncidence* __restrict__ inc,
                                                              double* __restrict__ x_out, double* __restrict__ iters_out) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x;
    const uint32_t comp = blockIdx.x, vpad = 1024u * m;
    double* tour = lds;          // 2 * vpad: prefix sums (first of r over the preorder, then over the Euler tour)
    double* pvec = lds;          // vpad: the vector being gathered by the products (the scans are idle then)
    double* xvec = tour + 2 * (size_t)vpad;  // vpad: the solution (kept out of the register file)
    double* red = xvec + vpad;   // 16
    const uint32_t k0 = (uint32_t)tid * m;

    double d[kTreeOwn], winv[kTreeOwn], r[kTreeOwn], p[kTreeOwn], q[kTreeOwn], z[kTreeOwn];
    uint32_t sz[kTreeOwn], ten[kTreeOwn], tex[kTreeOwn];
    bool fr[kTreeOwn];
#pragma unroll
    for (int j = 0; j < kTreeOwn; ++j) {
        d[j] = 0.0; winv[j] = 0.0; r[j] = 0.0; p[j] = 0.0; q[j] = 0.0; z[j] = 0.0;
        sz[j] = 1; ten[j] = 0; tex[j] = 0; fr[j] = false;
        const uint32_t k = k0 + (uint32_t)j;
        if ((uint32_t)j < m) xvec[k] = 0.0;  // only its owner ever touches an entry
        if ((uint32_t)j >= m || k >= n_views) continue;
        sz[j] = sub_size[k]; ten[j] = tour_enter[k]; tex[j] = tour_exit[k];
        const uint32_t pe = parent_edge[k];
        fr[j] = pe != 0xFFFFFFFFu;
        if (!fr[j]) continue;
        const double wp = w[pe];
        winv[j] = wp > 0.0 ? 1.0 / wp : 0.0;
        double dd = 0.0, bb = 0.0;
        for (uint32_t t = adj_ptr[k]; t < adj_ptr[k + 1]; ++t) {
            const uint32_t e = adj_edge[t];
            const double we = w[e];
            inc[t] = TreeIncidence{we, adj_other[t], 0u};  // read back by this same thread only (the three workgroups write the same values)
            dd += we;
            bb += (double)adj_sign[t] * we * omega[3 * (size_t)e + comp];
        }
        d[j] = dd;
        r[j] = bb;
    }
    // z = L_T^-1 r
    auto precondition = [&]() {
        // prefix sums of r over the preorder
        double run = 0.0, loc[kTreeOwn];
#pragma unroll
        for (int j = 0; j < kTreeOwn; ++j) { run += ((uint32_t)j < m) ? r[j] : 0.0; loc[j] = run; }
        const double before = block_prefix_1024(run, red, tid);
#pragma unroll
        for (int j = 0; j < kTreeOwn; ++j)
            if ((uint32_t)j < m) tour[k0 + (uint32_t)j] = before + loc[j];
        __syncthreads();
        double g[kTreeOwn];
#pragma unroll
        for (int j = 0; j < kTreeOwn; ++j) {
            g[j] = 0.0;
            const uint32_t k = k0 + (uint32_t)j;
            if ((uint32_t)j >= m || !fr[j]) continue;
            const double f = tour[k + sz[j] - 1u] - (k ? tour[k - 1u] : 0.0);  // what the subtree of k sends to its parent
            g[j] = f * winv[j];
        }
        __syncthreads();
        // Euler tour: +g at the entry of a view, -g at its exit (every position of the tour is written exactly once)
#pragma unroll
        for (int j = 0; j < kTreeOwn; ++j) {
            const uint32_t k = k0 + (uint32_t)j;
            if ((uint32_t)j >= m || k >= n_views) continue;
            tour[ten[j]] = g[j];
            tour[tex[j]] = -g[j];
        }
        __syncthreads();
        // inclusive prefix sums over the tour: thread t owns positions [2 t m, 2 t m + 2 m)
        const uint32_t t0 = 2u * k0, tn = 2u * n_views;
        double trun = 0.0;
        for (uint32_t i = 0; i < 2u * m; ++i) trun += (t0 + i < tn) ? tour[t0 + i] : 0.0;
        const double tbefore = block_prefix_1024(trun, red, tid);
        trun = tbefore;
        for (uint32_t i = 0; i < 2u * m; ++i)
            if (t0 + i < tn) { trun += tour[t0 + i]; tour[t0 + i] = trun; }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kTreeOwn; ++j) z[j] = ((uint32_t)j < m && fr[j]) ? tour[ten[j]] : 0.0;
        __syncthreads();  // tour is rewritten by the next call
    };
    precondition();
    double rz = 0.0;
#pragma unroll
    for (int j = 0; j < kTreeOwn; ++j) { p[j] = z[j]; rz += r[j] * z[j]; }
    rz = block_sum_1024(rz, red, tid);
    const double rz0 = rz;
    uint32_t it = 0;
    for (; it < cg_iters; ++it) {
        if (!(rz > cg_tol * cg_tol * rz0)) break;  // uniform: rz is identical in every thread
#pragma unroll
        for (int j = 0; j < kTreeOwn; ++j)
            if ((uint32_t)j < m) pvec[k0 + (uint32_t)j] = p[j];
        __syncthreads();
        double pAp = 0.0;
#pragma unroll
        for (int j = 0; j < kTreeOwn; ++j) {
            const uint32_t k = k0 + (uint32_t)j;
            double y = 0.0;
            if ((uint32_t)j < m && fr[j]) {
                y = d[j] * p[j];
                uint32_t t = adj_ptr[k];
                const uint32_t te = adj_ptr[k + 1];
                for (; t + 4 <= te; t += 4) {  // four independent loads in flight (roots hold p = 0)
                    const TreeIncidence a0 = inc[t], a1 = inc[t + 1], a2 = inc[t + 2], a3 = inc[t + 3];
                    y -= a0.w * pvec[a0.other];
                    y -= a1.w * pvec[a1.other];
                    y -= a2.w * pvec[a2.other];
                    y -= a3.w * pvec[a3.other];
                }
                for (; t < te; ++t) {
                    const TreeIncidence a0 = inc[t];
                    y -= a0.w * pvec[a0.other];
                }
            }
            q[j] = y;
            pAp += p[j] * y;
        }
        pAp = block_sum_1024(pAp, red, tid);
        const double alpha = pAp > 0.0 ? rz / pAp : 0.0;
#pragma unroll
        for (int j = 0; j < kTreeOwn; ++j) {
            if ((uint32_t)j < m) xvec[k0 + (uint32_t)j] += alpha * p[j];
            r[j] -= alpha * q[j];
        }
        precondition();
        double rzn = 0.0;
#pragma unroll
        for (int j = 0; j < kTreeOwn; ++j) rzn += r[j] * z[j];
        rzn = block_sum_1024(rzn, red, tid);
        const double beta = rz > 0.0 ? rzn / rz : 0.0;
#pragma unroll
        for (int j = 0; j < kTreeOwn; ++j) p[j] = z[j] + beta * p[j];
        rz = rzn;
    }
#pragma unroll
    for (int j = 0; j < kTreeOwn; ++j) {
        const uint32_t k = k0 + (uint32_t)j;
        if ((uint32_t)j < m && k < n_views) x_out[3 * (size_t)new_to_old[k] + comp] = xvec[k];
    }
    if (tid == 0) iters_out[comp] = (double)it;
}

// ---- multi-workgroup PCG (large graphs): scalars (gamma, alpha, beta, done) resident on the device; block partials
// are reduced in a fixed order.
struct CgState {
    double rz[3], rz0[3], alpha[3], beta[3];
    double mean_step, iters;
    int done;
    unsigned int ticket;  // workgroups that have delivered their partial sums (fused kernels)
};
constexpr int kCgBlock = 256;

__device__ void block_sum3_256(double v[3], double* red /* 3*256 */, int tid) {
    red[tid] = v[0];
    red[256 + tid] = v[1];
    red[512 + tid] = v[2];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            red[tid] += red[tid + s];
            red[256 + tid] += red[256 + tid + s];
            red[512 + tid] += red[512 + tid + s];
        }
        __syncthreads();
    }
    v[0] = red[0];
    v[1] = red[256];
    v[2] = red[512];
    __syncthreads();
}

constexpr int kRowLanes = 16;
constexpr int kRowsPerBlock = kCgBlock / kRowLanes;

// assemble diag / rhs (kRowLanes lanes share a view's incidences: one thread walking the tens of incidences of a dense scene
// graph's view took 47 us per outer step), x = 0, r = b, p = z = r / diag, q0 = s0 = 0, the PCG state records cleared
__global__ __launch_bounds__(kCgBlock) void cg_init_kernel(uint32_t n_views, const uint32_t* __restrict__ adj_ptr,
                                                           const uint32_t* __restrict__ adj_edge,
                                                           const int8_t* __restrict__ adj_sign,
                                                           const uint8_t* __restrict__ is_root,
                                                           const double* __restrict__ omega, const double* __restrict__ w,
                                                           double* __restrict__ diag, double* __restrict__ x,
                                                           double* __restrict__ r, double* __restrict__ p, double* __restrict__ q0,
                                                           double* __restrict__ s0, CgState* __restrict__ states) {
    const int tid = threadIdx.x, sub = tid & (kRowLanes - 1);
    const uint32_t k = blockIdx.x * kRowsPerBlock + (uint32_t)(tid / kRowLanes);
    if (blockIdx.x == 0 && tid < 3) states[tid] = CgState{};  // the two launch parities and the record the host reads
    double d = 0, b[3] = {0, 0, 0};
    if (k < n_views && !is_root[k])
        for (uint32_t a = adj_ptr[k] + (uint32_t)sub; a < adj_ptr[k + 1]; a += kRowLanes) {
            const uint32_t e = adj_edge[a];
            const double we = w[e], sg = (double)adj_sign[a];
            d += we;
#pragma unroll
            for (int c = 0; c < 3; ++c) b[c] += sg * we * omega[3 * (size_t)e + c];
        }
#pragma unroll
    for (int m = 1; m < kRowLanes; m <<= 1) {
        d += __shfl_xor(d, m);
#pragma unroll
        for (int c = 0; c < 3; ++c) b[c] += __shfl_xor(b[c], m);
    }
    if (k < n_views && sub == 0) {
        diag[k] = d;
        const double inv = d > 0.0 ? 1.0 / d : 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const size_t i = 3 * (size_t)k + c;
            x[i] = 0.0;
            r[i] = b[c];
            p[i] = b[c] * inv;
            q0[i] = s0[i] = 0.0;
        }
    }
}

// ---- ONE launch per PCG iteration.  The solve is bound by the ~13 us between small dependent kernels and by the depth
// of the gather chains inside them, not by arithmetic, so the iteration is arranged around a single grid-wide
// synchronisation point (the kernel boundary):
//   * Chronopoulos-Gear form of preconditioned CG: with s = A z kept next to q = A p, both inner products of an
//     iteration (gamma = r.z, delta = z.s) belong to the same vectors and are reduced together;
//       beta = gamma' / gamma,  alpha' = gamma' / (delta' - beta gamma' / alpha),
//       p = z + beta p,  q = s + beta q,  x += alpha p,  r -= alpha q,  z = r / diag,  s = A z
//   * a thread rebuilds the z entries it gathers from the PREVIOUS iteration's vectors,
//       z'[o] = (r[o] - alpha (s[o] + beta q[o])) / diag[o],
//     so the vector update needs no launch of its own (r, q, s are double-buffered: neighbours read the old ones);
//   * kRowLanes lanes share a view -- a view of a dense scene graph has tens of neighbours, and one thread walking
//     them is a chain of dependent gathers (40 us per product at V = 5000); the partial rows are combined by a fixed
//     xor tree;
//   * the fixed-order sum of the block partials is done by whichever workgroup delivers last (ticket counter).
// Everything is a fixed function of the graph: same input, same bits, on any number of ranks.
// Sum of 256 values per component (6 components) with EXACTLY the association of block_sum3_256's tree -- entry t takes
// entry t + s for s = 128, 64, ..., 1 -- but with two barriers instead of eighteen: the two upper steps go through LDS, the
// six lower ones are shuffles inside wavefront 0 (lane t + lane t + s for t < s is the same addition).  Every thread
// returns the block's sums.
__device__ void block_sum6_256(double v[6], double* red /* 6 * 256 */, int tid) {
#pragma unroll
    for (int c = 0; c < 6; ++c) red[256 * c + tid] = v[c];
    __syncthreads();
    if (tid < 128) {
#pragma unroll
        for (int c = 0; c < 6; ++c) red[256 * c + tid] += red[256 * c + tid + 128];
    }
    __syncthreads();
    if (tid < 64) {
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            double x = red[256 * c + tid] + red[256 * c + tid + 64];
#pragma unroll
            for (int sft = 32; sft >= 1; sft >>= 1) x += __shfl_down(x, sft);  // lane t (< sft) gets x_t + x_{t + sft}
            if (tid == 0) red[256 * c] = x;
        }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 6; ++c) v[c] = red[256 * c];
    __syncthreads();
}

// The scalars of one PCG iteration, derived from the block partials (gamma, delta per axis) of the PREVIOUS launch.
struct CgScalars {
    double alpha[3], beta[3], rz[3], rz0[3], iters;
    bool done;
};

// ONE launch per PCG iteration (round 4 form).  Round 3 ended every launch with a grid-wide hand-over: each workgroup
// published its partial sums behind an agent-scope fence, took a ticket, and the last one to arrive reduced all partials and
// wrote the next scalars -- a fence, an atomic and a serial tail on the critical path of a 15 us kernel.  Now a launch only
// WRITES its block's partials; the NEXT launch starts by reducing the previous launch's partials -- every workgroup
// redundantly, in the same fixed order (strided sums, then the 256-entry tree), so every workgroup holds the same bits --
// and derives alpha, beta and the stopping test from them.  Partials and the scalar state are double-buffered by launch
// parity, so nothing is read and written in the same launch.  mode 1: the init pass (alpha = beta = 0; x and r stay, p
// becomes z, q becomes 0: produces s0 = A z0 and the partials of gamma0, delta0); mode 0: an iteration; mode 2: only the
// reduction and the state record (what the host reads between chunks of launches), no vector is touched.
// Same arithmetic in the same order as the round-3 kernel: the rotations are bit-identical.
__global__ __launch_bounds__(kCgBlock) void cg_iteration_kernel(uint32_t n_views, const uint32_t* __restrict__ adj_ptr,
                                                                const uint32_t* __restrict__ adj_edge,
                                                                const uint32_t* __restrict__ adj_other,
                                                                const uint8_t* __restrict__ is_root, const double* __restrict__ w,
                                                                const double* __restrict__ diag, double* __restrict__ x,
                                                                double* __restrict__ p, const double* r_old, double* r_new,
                                                                const double* q_old, double* q_new, const double* s_old,
                                                                double* s_new, int mode, int prev_was_init, double tol,
                                                                const double* __restrict__ part_prev, double* __restrict__ part_next,
                                                                uint32_t n_blocks, const CgState* __restrict__ st_prev,
                                                                CgState* __restrict__ st_next) {
    __shared__ double red[6 * kCgBlock];
    const int tid = threadIdx.x, sub = tid & (kRowLanes - 1);
    double alpha[3] = {0, 0, 0}, beta[3] = {0, 0, 0};
    if (mode != 1) {
        if (st_prev->done) {  // (uniform) converged earlier: keep the record where the host reads it
            if (blockIdx.x == 0 && tid == 0) *st_next = *st_prev;
            return;
        }
        double v[6] = {0, 0, 0, 0, 0, 0};
        for (uint32_t b = tid; b < n_blocks; b += kCgBlock)
#pragma unroll
            for (int c = 0; c < 6; ++c) v[c] += part_prev[6 * (size_t)b + c];
        block_sum6_256(v, red, tid);
        bool done = true;
        double rz_new[3], rz0[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double g = v[c], dl = v[3 + c];
            if (prev_was_init) {
                rz0[c] = g;
                beta[c] = 0.0;
                alpha[c] = dl > 0.0 ? g / dl : 0.0;
                done &= !(g > 0.0);
            } else {
                rz0[c] = st_prev->rz0[c];
                const double a_prev = st_prev->alpha[c];
                const double bt = st_prev->rz[c] > 0.0 ? g / st_prev->rz[c] : 0.0;
                const double den = a_prev != 0.0 ? dl - bt * g / a_prev : dl;
                beta[c] = bt;
                alpha[c] = den > 0.0 ? g / den : 0.0;
                done &= !(g > tol * tol * rz0[c]);
            }
            rz_new[c] = g;
        }
        if (blockIdx.x == 0 && tid == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                st_next->rz[c] = rz_new[c];
                st_next->rz0[c] = rz0[c];
                st_next->alpha[c] = alpha[c];
                st_next->beta[c] = beta[c];
            }
            st_next->iters = prev_was_init ? 0.0 : st_prev->iters + 1.0;
            st_next->done = done;
            st_next->mean_step = 0.0;
            st_next->ticket = 0;
        }
        if (done || mode == 2) return;  // (uniform)
    }
    const uint32_t k = blockIdx.x * kRowsPerBlock + (uint32_t)(tid / kRowLanes);
    double gd[6] = {0, 0, 0, 0, 0, 0};  // gamma (3), delta (3)
    if (k < n_views) {
        const bool free_k = !is_root[k];
        // neighbours: their new z, rebuilt from the old vectors
        double y[3] = {0, 0, 0};
        if (free_k)
            for (uint32_t a = adj_ptr[k] + (uint32_t)sub; a < adj_ptr[k + 1]; a += kRowLanes) {
                const uint32_t o = adj_other[a];
                if (is_root[o]) continue;
                const double we = w[adj_edge[a]], dn = diag[o], inv = dn > 0.0 ? 1.0 / dn : 0.0;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const size_t i = 3 * (size_t)o + c;
                    const double qo = s_old[i] + beta[c] * q_old[i];
                    y[c] -= we * ((r_old[i] - alpha[c] * qo) * inv);
                }
            }
#pragma unroll
        for (int m = 1; m < kRowLanes; m <<= 1)
#pragma unroll
            for (int c = 0; c < 3; ++c) y[c] += __shfl_xor(y[c], m);
        if (sub == 0) {  // the view's own entries of every vector
            const double d = diag[k], inv = d > 0.0 ? 1.0 / d : 0.0;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const size_t i = 3 * (size_t)k + c;
                const double ro = r_old[i];
                const double pk = ro * inv + beta[c] * p[i];
                const double qk = s_old[i] + beta[c] * q_old[i];
                const double rn = ro - alpha[c] * qk, zn = rn * inv;
                const double sn = free_k ? d * zn + y[c] : 0.0;
                p[i] = pk;
                q_new[i] = qk;
                x[i] += alpha[c] * pk;
                r_new[i] = rn;
                s_new[i] = sn;
                gd[c] = rn * zn;
                gd[3 + c] = zn * sn;
            }
        }
    }
    block_sum6_256(gd, red, tid);
    if (tid == 0)
#pragma unroll
        for (int c = 0; c < 6; ++c) part_next[6 * (size_t)blockIdx.x + c] = gd[c];
}

// (gate: a record of the PCG state on the device, or null.  With a gate the kernel runs only if that solve has converged: the
// host queues it behind the reduce-only launch it is about to look at, so that a converged solve costs ONE round trip.
// veto: a device flag, or null; set = do nothing -- the two-level solve's "coarse matrix not inverted".)
__global__ __launch_bounds__(kCgBlock) void cg_step_norm_kernel(uint32_t n_views, const double* __restrict__ x,
                                                                double* __restrict__ partial, const CgState* __restrict__ gate,
                                                                const int* __restrict__ veto) {
    __shared__ double red[3 * kCgBlock];
    const int tid = threadIdx.x;
    if ((gate && !gate->done) || (veto && *veto)) return;  // (uniform)
    const uint32_t k = blockIdx.x * kCgBlock + tid;
    double nd[3] = {0, 0, 0};
    if (k < n_views)
        nd[0] = sqrt(x[3 * (size_t)k] * x[3 * (size_t)k] + x[3 * (size_t)k + 1] * x[3 * (size_t)k + 1] +
                     x[3 * (size_t)k + 2] * x[3 * (size_t)k + 2]);
    block_sum3_256(nd, red, tid);
    if (tid == 0)
        for (int c = 0; c < 3; ++c) partial[3 * blockIdx.x + c] = nd[c];
}

__global__ __launch_bounds__(256) void rot_update_kernel(uint32_t n_views, const double* __restrict__ x,
                                                         double* __restrict__ R, const CgState* __restrict__ gate,
                                                         const int* __restrict__ veto) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n_views || (gate && !gate->done) || (veto && *veto)) return;
    double Rk[9], Ex[9], Rn[9];
    const double wv[3] = {x[3 * (size_t)k], x[3 * (size_t)k + 1], x[3 * (size_t)k + 2]};
#pragma unroll
    for (int c = 0; c < 9; ++c) Rk[c] = R[9 * (size_t)k + c];
    so3_exp(wv, Ex);
    mat3_mul(Rk, Ex, Rn);
#pragma unroll
    for (int c = 0; c < 9; ++c) R[9 * (size_t)k + c] = Rn[c];
}

// ---- two-level PCG (dense, band-like view graphs: every Jacobi solve runs into the cap) ----------------------------------
// A view graph whose views see their neighbours along a walk or a ring -- tens of edges per view, so the spanning-tree kernel
// does not take it -- has a Laplacian with kappa ~ (V / reach)^2: Jacobi-preconditioned CG needs hundreds of iterations per
// solve and the cap truncates every one of them (config 4 at V = 5000, k = 40: 13 outer steps x 200 iterations).  Here the
// preconditioner gets a second level:  M^-1 = D^-1 + P Ac^-1 P^T,  P = the indicator of <= 128 AGGREGATES of neighbouring
// views (host: region growing over the adjacency), Ac = P^T A P assembled and inverted on the device for every outer step's
// weights (a dense <= 128 x 128 matrix: Gauss-Jordan in one workgroup's LDS).  The iteration keeps the one-launch form:
//   * the rows are renumbered so that an aggregate is a run of whole 16-row blocks (padding rows are inert): P^T v of a
//     vector written by the previous launch is a fixed-order sum of that launch's block partials;
//   * rho = P^T r and kappa = P^T q follow the same recurrences as r and q (kappa' = sigma + beta kappa, rho' = rho - alpha
//     kappa', sigma = P^T s from the block partials), every workgroup redundantly, workgroup 0 stores them by launch parity;
//   * every workgroup applies Ac^-1 to rho' (<= 128 x 128 doubles from L2, read through the symmetric counterpart so that
//     consecutive lanes read consecutive words) and keeps c = Ac^-1 rho' in LDS; a neighbour's rebuilt z gets c of ITS aggregate
//     (the aggregate id travels with the incidence), the view's own z the block's.
// Same input, same bits: no atomics, every sum in a fixed order.
constexpr uint32_t kMaxAggregates = 128;
constexpr uint8_t kNoAggregate = 255;

__device__ void block_sum9_256(double v[9], double* red /* 9 * 256 */, int tid) {  // the tree of block_sum6_256, nine components
#pragma unroll
    for (int c = 0; c < 9; ++c) red[256 * c + tid] = v[c];
    __syncthreads();
    if (tid < 128) {
#pragma unroll
        for (int c = 0; c < 9; ++c) red[256 * c + tid] += red[256 * c + tid + 128];
    }
    __syncthreads();
    if (tid < 64) {
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            double x = red[256 * c + tid] + red[256 * c + tid + 64];
#pragma unroll
            for (int sft = 32; sft >= 1; sft >>= 1) x += __shfl_down(x, sft);
            if (tid == 0) red[256 * c] = x;
        }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 9; ++c) v[c] = red[256 * c];
    __syncthreads();
}

// diag / rhs per row (kRowLanes lanes share a row), x = 0, r = b, p = 0; block sums of r (-> rho0 = P^T r0)
__global__ __launch_bounds__(kCgBlock) void cg2_init_kernel(uint32_t n_rows, const uint32_t* __restrict__ adj_ptr,
                                                            const uint32_t* __restrict__ adj_edge, const int8_t* __restrict__ adj_sign,
                                                            const uint8_t* __restrict__ is_root, const double* __restrict__ omega,
                                                            const double* __restrict__ w, double* __restrict__ diag,
                                                            double* __restrict__ x, double* __restrict__ r, double* __restrict__ p,
                                                            double* __restrict__ rpart, double* __restrict__ q0, double* __restrict__ s0,
                                                            CgState* __restrict__ states) {
    __shared__ double red[3 * kCgBlock];
    const int tid = threadIdx.x, sub = tid & (kRowLanes - 1);
    const uint32_t k = blockIdx.x * kRowsPerBlock + (uint32_t)(tid / kRowLanes);
    if (blockIdx.x == 0 && tid < 3) states[tid] = CgState{};
    double d = 0, b[3] = {0, 0, 0};
    if (k < n_rows && !is_root[k])
        for (uint32_t a = adj_ptr[k] + (uint32_t)sub; a < adj_ptr[k + 1]; a += kRowLanes) {
            const uint32_t e = adj_edge[a];
            const double we = w[e], sg = (double)adj_sign[a];
            d += we;
#pragma unroll
            for (int c = 0; c < 3; ++c) b[c] += sg * we * omega[3 * (size_t)e + c];
        }
#pragma unroll
    for (int m = 1; m < kRowLanes; m <<= 1) {
        d += __shfl_xor(d, m);
#pragma unroll
        for (int c = 0; c < 3; ++c) b[c] += __shfl_xor(b[c], m);
    }
    double rs[3] = {0, 0, 0};
    if (k < n_rows && sub == 0) {
        diag[k] = d;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            x[3 * (size_t)k + c] = 0.0;
            r[3 * (size_t)k + c] = b[c];
            p[3 * (size_t)k + c] = 0.0;
            q0[3 * (size_t)k + c] = s0[3 * (size_t)k + c] = 0.0;
            rs[c] = b[c];
        }
    }
    block_sum3_256(rs, red, tid);
    if (tid == 0)
        for (int c = 0; c < 3; ++c) rpart[3 * (size_t)blockIdx.x + c] = rs[c];
}

// Ac = P^T A P: workgroup a owns row a.  The incidences of the aggregate's rows are one contiguous range of the adjacency: the
// threads stage (column, weight) of a chunk of it in LDS side by side, then thread (b, segment) adds what falls into column b
// within its eighth of the chunk in index order and the eight partial sums are added in segment order -- a fixed order per
// entry, and the loads of a chunk are in flight together (walking them one after the other, every thread alike, took 300 us per
// outer step).  The diagonal entry also takes the rows' diag.
constexpr int kAsmChunk = 2048, kAsmSegments = 8;
__global__ __launch_bounds__(kMaxAggregates * kAsmSegments) void cg2_coarse_assemble_kernel(
    uint32_t n_agg, const uint32_t* __restrict__ agg_block, const uint32_t* __restrict__ adj_ptr, const uint32_t* __restrict__ adj_edge,
    const uint8_t* __restrict__ adj_agg, const uint8_t* __restrict__ is_root, const double* __restrict__ w, const double* __restrict__ diag,
    double* __restrict__ Ac) {
    __shared__ double val[kAsmChunk];
    __shared__ uint8_t col[kAsmChunk];
    __shared__ double seg_sum[kAsmSegments][kMaxAggregates];
    const uint32_t a = blockIdx.x, tid = threadIdx.x, b = tid & (kMaxAggregates - 1), seg = tid / kMaxAggregates;
    constexpr uint32_t kThreads = kMaxAggregates * kAsmSegments, kPerSeg = kAsmChunk / kAsmSegments;
    const uint32_t row0 = agg_block[a] * kRowsPerBlock, row1 = agg_block[a + 1] * kRowsPerBlock;
    double acc = 0;
    // (an inert row -- padding -- has no incidences and diag 0)
    for (uint32_t r0 = row0; r0 < row1; r0 += kAsmChunk) {
        const uint32_t n = min((uint32_t)kAsmChunk, row1 - r0);
        for (uint32_t i = tid; i < n; i += kThreads) val[i] = is_root[r0 + i] ? 0.0 : diag[r0 + i];
        __syncthreads();
        if (tid == a)  // (segment 0, column a)
            for (uint32_t i = 0; i < n; ++i) acc += val[i];
        __syncthreads();
    }
    const uint32_t i0 = adj_ptr[row0], i1 = adj_ptr[row1];
    for (uint32_t c0 = i0; c0 < i1; c0 += kAsmChunk) {
        const uint32_t n = min((uint32_t)kAsmChunk, i1 - c0);
        for (uint32_t i = tid; i < n; i += kThreads) {
            col[i] = adj_agg[c0 + i];
            val[i] = w[adj_edge[c0 + i]];
        }
        __syncthreads();
        double part = 0;
        const uint32_t e1 = min(n, (seg + 1) * kPerSeg);
        for (uint32_t i = seg * kPerSeg; i < e1; ++i)
            if (col[i] == b) part -= val[i];  // (incidences to gauge views carry kNoAggregate)
        seg_sum[seg][b] = part;
        __syncthreads();
        if (seg == 0)
#pragma unroll
            for (int g = 0; g < kAsmSegments; ++g) acc += seg_sum[g][b];
        __syncthreads();
    }
    if (seg == 0 && b < n_agg) Ac[(size_t)a * n_agg + b] = acc;
}

// Ac^-1 by Gauss-Jordan without pivoting (Ac is symmetric positive definite) in one workgroup, symmetrised on the way out;
// status = 1 when a pivot was not positive.  The matrix (padded to 128 x 128 by the identity) lives in REGISTERS: thread
// (ti, tj) of the 32 x 32 owns the 4 x 4 tile at (4 ti, 4 tj) through all steps; a step sends the pivot column and row through
// LDS (double-buffered by step parity: one barrier per step) and every thread reads its four entries of each -- 8 LDS reads
// for 16 updates (one entry per thread and LDS word, or the matrix itself in LDS, was bound by LDS bandwidth: 296 us).
// The rule per entry: (k,k) -> 1/p; row k -> M[k][j] / p; column k -> -M[i][k] / p; else M[i][j] - M[i][k] (M[k][j] / p).
__global__ __launch_bounds__(1024) void cg2_coarse_invert_kernel(uint32_t n, const double* __restrict__ Ac, double* __restrict__ Ainv,
                                                                 int* __restrict__ status) {
    __shared__ double colb[2][kMaxAggregates], rowb[2][kMaxAggregates];
    extern __shared__ double lds_inv[];  // n * n doubles: only for the symmetrisation at the end
    const uint32_t tid = threadIdx.x, ti = tid >> 5, tj = tid & 31;
    double v[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const uint32_t i = 4 * ti + a, j = 4 * tj + b;
            v[a][b] = (i < n && j < n) ? Ac[(size_t)i * n + j] : (i == j ? 1.0 : 0.0);
        }
    bool bad = false;
    for (uint32_t k = 0; k < n; ++k) {
        double* col = colb[k & 1];
        double* row = rowb[k & 1];
        const uint32_t kt = k >> 2, kr = k & 3;  // (uniform: the switches below are scalar branches, the register indices static)
        if (tj == kt) {
            switch (kr) {
                case 0: for (int a = 0; a < 4; ++a) col[4 * ti + a] = v[a][0]; break;
                case 1: for (int a = 0; a < 4; ++a) col[4 * ti + a] = v[a][1]; break;
                case 2: for (int a = 0; a < 4; ++a) col[4 * ti + a] = v[a][2]; break;
                default: for (int a = 0; a < 4; ++a) col[4 * ti + a] = v[a][3]; break;
            }
        }
        if (ti == kt) {
            switch (kr) {
                case 0: for (int b = 0; b < 4; ++b) row[4 * tj + b] = v[0][b]; break;
                case 1: for (int b = 0; b < 4; ++b) row[4 * tj + b] = v[1][b]; break;
                case 2: for (int b = 0; b < 4; ++b) row[4 * tj + b] = v[2][b]; break;
                default: for (int b = 0; b < 4; ++b) row[4 * tj + b] = v[3][b]; break;
            }
        }
        __syncthreads();
        const double piv = row[k];
        bad |= !(piv > 0.0);
        const double pinv = piv > 0.0 ? 1.0 / piv : 0.0;
        double c[4], r[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            c[a] = col[4 * ti + a];
            r[a] = row[4 * tj + a] * pinv;
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) v[a][b] = v[a][b] - c[a] * r[b];
        // the pivot row, the pivot column and the pivot itself (few lanes; which of a tile's rows / columns is uniform)
        if (ti == kt) {
            switch (kr) {
                case 0: for (int b = 0; b < 4; ++b) v[0][b] = r[b]; break;
                case 1: for (int b = 0; b < 4; ++b) v[1][b] = r[b]; break;
                case 2: for (int b = 0; b < 4; ++b) v[2][b] = r[b]; break;
                default: for (int b = 0; b < 4; ++b) v[3][b] = r[b]; break;
            }
        }
        if (tj == kt) {
            switch (kr) {
                case 0: for (int a = 0; a < 4; ++a) v[a][0] = -c[a] * pinv; break;
                case 1: for (int a = 0; a < 4; ++a) v[a][1] = -c[a] * pinv; break;
                case 2: for (int a = 0; a < 4; ++a) v[a][2] = -c[a] * pinv; break;
                default: for (int a = 0; a < 4; ++a) v[a][3] = -c[a] * pinv; break;
            }
            if (ti == kt) {
                switch (kr) {
                    case 0: v[0][0] = pinv; break;
                    case 1: v[1][1] = pinv; break;
                    case 2: v[2][2] = pinv; break;
                    default: v[3][3] = pinv; break;
                }
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const uint32_t i = 4 * ti + a, j = 4 * tj + b;
            if (i < n && j < n) lds_inv[(size_t)i * n + j] = v[a][b];
        }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const uint32_t i = 4 * ti + a, j = 4 * tj + b;
            if (i < n && j < n) Ainv[(size_t)i * n + j] = 0.5 * (v[a][b] + lds_inv[(size_t)j * n + i]);
        }
    if (tid == 0) *status = bad ? 1 : 0;
}

// One launch per iteration, as cg_iteration_kernel (modes 1 / 0 / 2, scalars from the previous launch's partials), with the
// coarse correction.  Partials per block: gamma (3), delta (3), the block's sum of s (3).  cs_*: rho | kappa | c, 3 n_agg each.
__global__ __launch_bounds__(kCgBlock) void cg2_iteration_kernel(
    uint32_t n_rows, const uint32_t* __restrict__ adj_ptr, const uint32_t* __restrict__ adj_edge, const uint32_t* __restrict__ adj_other,
    const uint8_t* __restrict__ adj_agg, const uint8_t* __restrict__ is_root, const double* __restrict__ w, const double* __restrict__ diag,
    double* __restrict__ x, double* __restrict__ p, const double* r_old, double* r_new, const double* q_old, double* q_new,
    const double* s_old, double* s_new, int mode, int prev_was_init, double tol, const double* __restrict__ part_prev,
    double* __restrict__ part_next, uint32_t n_blocks, const CgState* __restrict__ st_prev, CgState* __restrict__ st_next, uint32_t n_agg,
    const uint32_t* __restrict__ agg_block, const uint8_t* __restrict__ agg_of_block, const double* __restrict__ Ainv,
    const double* __restrict__ rpart, const double* __restrict__ cs_prev, double* __restrict__ cs_next) {
    __shared__ double red[9 * kCgBlock];
    __shared__ double l_rho[3 * kMaxAggregates], l_c[3 * kMaxAggregates], l_half[2][3 * kMaxAggregates];
    const int tid = threadIdx.x, sub = tid & (kRowLanes - 1);
    double alpha[3] = {0, 0, 0}, beta[3] = {0, 0, 0};
    if (mode != 1) {
        if (st_prev->done) {  // (uniform) converged earlier: keep the record where the host reads it
            if (blockIdx.x == 0 && tid == 0) *st_next = *st_prev;
            return;
        }
        double v[6] = {0, 0, 0, 0, 0, 0};
        for (uint32_t b = tid; b < n_blocks; b += kCgBlock)
#pragma unroll
            for (int c = 0; c < 6; ++c) v[c] += part_prev[9 * (size_t)b + c];
        block_sum6_256(v, red, tid);
        bool done = true;
        double rz_new[3], rz0[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double g = v[c], dl = v[3 + c];
            if (prev_was_init) {
                rz0[c] = g;
                beta[c] = 0.0;
                alpha[c] = dl > 0.0 ? g / dl : 0.0;
                done &= !(g > 0.0);
            } else {
                rz0[c] = st_prev->rz0[c];
                const double a_prev = st_prev->alpha[c];
                const double bt = st_prev->rz[c] > 0.0 ? g / st_prev->rz[c] : 0.0;
                const double den = a_prev != 0.0 ? dl - bt * g / a_prev : dl;
                beta[c] = bt;
                alpha[c] = den > 0.0 ? g / den : 0.0;
                done &= !(g > tol * tol * rz0[c]);
            }
            rz_new[c] = g;
        }
        if (blockIdx.x == 0 && tid == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                st_next->rz[c] = rz_new[c];
                st_next->rz0[c] = rz0[c];
                st_next->alpha[c] = alpha[c];
                st_next->beta[c] = beta[c];
            }
            st_next->iters = prev_was_init ? 0.0 : st_prev->iters + 1.0;
            st_next->done = done;
            st_next->mean_step = 0.0;
            st_next->ticket = 0;
        }
        if (done || mode == 2) return;  // (uniform)
    }
    // the coarse vectors of this iteration
    if (tid < (int)n_agg) {
        double rho[3] = {0, 0, 0}, kap[3] = {0, 0, 0};
        if (mode == 1) {
            for (uint32_t b = agg_block[tid]; b < agg_block[tid + 1]; ++b)
#pragma unroll
                for (int c = 0; c < 3; ++c) rho[c] += rpart[3 * (size_t)b + c];
        } else {
            double sig[3] = {0, 0, 0};
            for (uint32_t b = agg_block[tid]; b < agg_block[tid + 1]; ++b)
#pragma unroll
                for (int c = 0; c < 3; ++c) sig[c] += part_prev[9 * (size_t)b + 6 + c];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                kap[c] = sig[c] + beta[c] * cs_prev[3 * (size_t)n_agg + 3 * tid + c];
                rho[c] = cs_prev[3 * tid + c] - alpha[c] * kap[c];
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            l_rho[3 * tid + c] = rho[c];
            if (blockIdx.x == 0) {
                cs_next[3 * tid + c] = rho[c];
                cs_next[3 * (size_t)n_agg + 3 * tid + c] = kap[c];
            }
        }
    }
    __syncthreads();
    {   // c = Ac^-1 rho: thread (a, half) sums half of the columns
        const uint32_t a = (uint32_t)tid & (kMaxAggregates - 1), h = (uint32_t)tid >> 7, half = (n_agg + 1) / 2;
        const uint32_t b0 = h * half, b1 = min(n_agg, b0 + half);
        double acc[3] = {0, 0, 0};
        if (a < n_agg) {
            for (uint32_t b = b0; b < b1; ++b) {
                const double m = Ainv[(size_t)b * n_agg + a];
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[c] += m * l_rho[3 * b + c];
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) l_half[h][3 * a + c] = acc[c];
        }
    }
    __syncthreads();
    if (tid < (int)n_agg) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double cn = l_half[0][3 * tid + c] + l_half[1][3 * tid + c];
            l_c[3 * tid + c] = cn;
            if (blockIdx.x == 0) cs_next[6 * (size_t)n_agg + 3 * tid + c] = cn;
        }
    }
    __syncthreads();
    const uint32_t my_agg = agg_of_block[blockIdx.x];
    double c_old[3] = {0, 0, 0}, c_own[3] = {0, 0, 0};
    if (my_agg != kNoAggregate) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            c_own[c] = l_c[3 * my_agg + c];
            c_old[c] = mode == 1 ? c_own[c] : cs_prev[6 * (size_t)n_agg + 3 * my_agg + c];
        }
    }
    const uint32_t k = blockIdx.x * kRowsPerBlock + (uint32_t)(tid / kRowLanes);
    double gd[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // gamma (3), delta (3), the sum of s (3)
    if (k < n_rows) {
        const bool free_k = !is_root[k];
        double y[3] = {0, 0, 0};
        if (free_k)
            for (uint32_t a = adj_ptr[k] + (uint32_t)sub; a < adj_ptr[k + 1]; a += kRowLanes) {
                const uint32_t ag = adj_agg[a];
                if (ag == kNoAggregate) continue;  // a gauge view
                const uint32_t o = adj_other[a];
                const double we = w[adj_edge[a]], dn = diag[o], inv = dn > 0.0 ? 1.0 / dn : 0.0;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const size_t i = 3 * (size_t)o + c;
                    const double qo = s_old[i] + beta[c] * q_old[i];
                    y[c] -= we * ((r_old[i] - alpha[c] * qo) * inv + l_c[3 * ag + c]);
                }
            }
#pragma unroll
        for (int m = 1; m < kRowLanes; m <<= 1)
#pragma unroll
            for (int c = 0; c < 3; ++c) y[c] += __shfl_xor(y[c], m);
        if (sub == 0) {
            const double d = diag[k], inv = d > 0.0 ? 1.0 / d : 0.0;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const size_t i = 3 * (size_t)k + c;
                const double ro = r_old[i];
                const double zo = free_k ? ro * inv + c_old[c] : 0.0;
                const double pk = zo + beta[c] * p[i];
                const double qk = s_old[i] + beta[c] * q_old[i];
                const double rn = ro - alpha[c] * qk;
                const double zn = free_k ? rn * inv + c_own[c] : 0.0;
                const double sn = free_k ? d * zn + y[c] : 0.0;
                p[i] = pk;
                q_new[i] = qk;
                x[i] += alpha[c] * pk;
                r_new[i] = rn;
                s_new[i] = sn;
                gd[c] = rn * zn;
                gd[3 + c] = zn * sn;
                gd[6 + c] = sn;
            }
        }
    }
    block_sum9_256(gd, red, tid);
    if (tid == 0)
#pragma unroll
        for (int c = 0; c < 9; ++c) part_next[9 * (size_t)blockIdx.x + c] = gd[c];
}

// R_view <- R_view exp(x_row) for the renumbered rows (padding rows map to no view)
__global__ __launch_bounds__(256) void rot_update_rows_kernel(uint32_t n_rows, const double* __restrict__ x, const uint32_t* __restrict__ row_view,
                                                              double* __restrict__ R, const CgState* __restrict__ gate,
                                                              const int* __restrict__ veto) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n_rows || (gate && !gate->done) || (veto && *veto)) return;
    const uint32_t v = row_view[k];
    if (v == 0xFFFFFFFFu) return;
    double Rk[9], Ex[9], Rn[9];
    const double wv[3] = {x[3 * (size_t)k], x[3 * (size_t)k + 1], x[3 * (size_t)k + 2]};
#pragma unroll
    for (int c = 0; c < 9; ++c) Rk[c] = R[9 * (size_t)v + c];
    so3_exp(wv, Ex);
    mat3_mul(Rk, Ex, Rn);
#pragma unroll
    for (int c = 0; c < 9; ++c) R[9 * (size_t)v + c] = Rn[c];
}

// pgi_edge records (status OK) -> rotation-graph edges, all on the device: idx[e] = pair of edge e
__global__ __launch_bounds__(256) void rot_edges_from_table_kernel(const pgi_edge* __restrict__ table, const uint32_t* __restrict__ idx,
                                                                    const uint32_t* __restrict__ src, const uint32_t* __restrict__ dst,
                                                                    const double* __restrict__ weight, uint32_t n_edges,
                                                                    RotEdgeDev* __restrict__ out) {
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n_edges) return;
    const pgi_edge* rec = table + idx[e];
    RotEdgeDev o;
    o.src = src[e];
    o.dst = dst[e];
#pragma unroll
    for (int c = 0; c < 9; ++c) o.R[c] = rec->R[c];
    o.weight = weight[e];
    out[e] = o;
}
// rotations of selected edges (the spanning forest's) -> a dense array for the host
__global__ __launch_bounds__(256) void rot_gather_R_kernel(const RotEdgeDev* __restrict__ edges, const uint32_t* __restrict__ sel,
                                                           uint32_t n_sel, double* __restrict__ out) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n_sel) return;
    const RotEdgeDev* e = edges + sel[k];
#pragma unroll
    for (int c = 0; c < 9; ++c) out[9 * (size_t)k + c] = e->R[c];
}

// ---- host: maximum-weight spanning forest + BFS initialisation ----------------------------------------
struct ForestEdge {
    uint32_t edge, other;  // other | inv << 31
};
// Kruskal on (weight desc, edge index asc); returns the forest's edge list and its adjacency
static void spanning_forest(uint32_t V, const uint32_t* src, const uint32_t* dst, const double* weight, uint32_t nE,
                            std::vector<uint32_t>& tree, std::vector<std::vector<ForestEdge>>& adj) {
    // (weight desc, edge index asc): ONE integer key per edge -- the order-preserving bit pattern of the weight, inverted -- and a
    // stable sort by it
    std::vector<uint64_t> key[2] = {std::vector<uint64_t>(nE), std::vector<uint64_t>(nE)};
    std::vector<uint32_t> idx[2] = {std::vector<uint32_t>(nE), std::vector<uint32_t>(nE)};
    for (uint32_t e = 0; e < nE; ++e) {
        uint64_t u;
        const double wv = weight[e] == 0.0 ? 0.0 : weight[e];  // (-0.0 and +0.0 compare equal: one key)
        memcpy(&u, &wv, 8);
        u = (u >> 63) ? ~u : (u | 0x8000000000000000ull);  // ascending in the value
        key[0][e] = ~u;                                      // descending
        idx[0][e] = e;
    }
    {   // stable least-significant-digit radix sort, 11-bit digits; a digit all keys share is skipped (round 4: std::sort of the
        // pairs took 5.8 ms at 10^5 edges)
        constexpr int kBits = 11, kPasses = 6;
        std::vector<uint32_t> hist((size_t)kPasses << kBits, 0);
        for (uint32_t e = 0; e < nE; ++e)
            for (int ps = 0; ps < kPasses; ++ps) ++hist[((size_t)ps << kBits) + ((key[0][e] >> (ps * kBits)) & ((1u << kBits) - 1))];
        int cur = 0;
        for (int ps = 0; ps < kPasses; ++ps) {
            uint32_t* h = &hist[(size_t)ps << kBits];
            bool single = false;
            for (uint32_t b = 0; b < (1u << kBits); ++b) single |= h[b] == nE;
            if (single) continue;
            uint32_t run = 0;
            for (uint32_t b = 0; b < (1u << kBits); ++b) {
                const uint32_t c = h[b];
                h[b] = run;
                run += c;
            }
            const uint64_t* ks = key[cur].data();
            const uint32_t* is = idx[cur].data();
            uint64_t* kd = key[cur ^ 1].data();
            uint32_t* id = idx[cur ^ 1].data();
            for (uint32_t e = 0; e < nE; ++e) {
                const uint32_t pos = h[(ks[e] >> (ps * kBits)) & ((1u << kBits) - 1)]++;
                kd[pos] = ks[e];
                id[pos] = is[e];
            }
            cur ^= 1;
        }
        if (cur) idx[0].swap(idx[1]);
    }
    const std::vector<uint32_t>& order = idx[0];
    std::vector<uint32_t> parent(V);
    std::iota(parent.begin(), parent.end(), 0u);
    auto find = [&](uint32_t a) {
        while (parent[a] != a) {
            parent[a] = parent[parent[a]];
            a = parent[a];
        }
        return a;
    };
    adj.assign(V, {});
    tree.clear();
    for (uint32_t e : order) {
        const uint32_t a = find(src[e]), b = find(dst[e]);
        if (a == b) continue;
        parent[std::max(a, b)] = std::min(a, b);
        adj[src[e]].push_back({e, dst[e]});
        adj[dst[e]].push_back({e, src[e] | 0x80000000u});
        tree.push_back(e);
        if (tree.size() + 1 == V) break;  // spanning: the remaining edges all close cycles
    }
}
// BFS over the forest; R_of(e) = pointer to the 9 doubles of edge e's relative rotation
template <class RofE>
static void forest_bfs_init(uint32_t V, const std::vector<std::vector<ForestEdge>>& adj, RofE R_of, std::vector<double>& R,
                            std::vector<uint8_t>& is_root) {
    R.assign((size_t)V * 9, 0.0);
    is_root.assign(V, 0);
    std::vector<uint8_t> seen(V, 0);
    std::vector<uint32_t> queue;
    for (uint32_t root = 0; root < V; ++root) {
        if (seen[root]) continue;
        seen[root] = 1;
        is_root[root] = 1;
        R[9 * (size_t)root] = R[9 * (size_t)root + 4] = R[9 * (size_t)root + 8] = 1.0;
        queue.assign(1, root);
        for (size_t h = 0; h < queue.size(); ++h) {
            const uint32_t u = queue[h];
            for (const ForestEdge& pr : adj[u]) {
                const uint32_t e = pr.edge, v = pr.other & 0x7FFFFFFFu;
                const bool inv = pr.other >> 31;
                if (seen[v]) continue;
                seen[v] = 1;
                const double* Rr = R_of(e);
                const double* Ru = &R[9 * (size_t)u];
                double* Rv = &R[9 * (size_t)v];
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j < 3; ++j) {
                        double s = 0;
                        for (int k = 0; k < 3; ++k) s += (inv ? Rr[3 * k + i] : Rr[3 * i + k]) * Ru[3 * k + j];
                        Rv[3 * i + j] = s;
                    }
                queue.push_back(v);
            }
        }
    }
}

// ---- host: the aggregates of the two-level preconditioner and the renumbered adjacency ------------------------------------
struct TwoLevelPlan {
    uint32_t n_agg = 0, n_rows = 0, n_blocks = 0;
    std::vector<uint32_t> row_view;                 // row -> view, 0xFFFFFFFF = padding
    std::vector<uint32_t> ptr, aedge, aother;       // the adjacency of the rows (incidences in the views' order)
    std::vector<int8_t> asign;
    std::vector<uint8_t> aagg, root, agg_of_block;  // aggregate of the incidence's other end / of the block; row is inert
    std::vector<uint32_t> agg_block;                // n_agg + 1 block starts
};
// Breadth-first order of all views from the gauge views; returns the depth of the deepest view.  A band-like graph (views along
// a walk or a ring) is deep -- V / reach levels -- a densely connected one has a handful of levels.
static uint32_t bfs_from_gauge_views(uint32_t V, const std::vector<uint32_t>& ptr, const std::vector<uint32_t>& aother,
                                     const std::vector<uint8_t>& is_root, std::vector<uint32_t>& order) {
    order.clear();
    order.reserve(V);
    std::vector<uint32_t> level(V, 0xFFFFFFFFu);
    for (uint32_t v = 0; v < V; ++v)
        if (is_root[v]) { level[v] = 0; order.push_back(v); }
    uint32_t depth = 0;
    for (size_t h = 0; h < order.size(); ++h) {
        const uint32_t u = order[h];
        for (uint32_t a = ptr[u]; a < ptr[u + 1]; ++a) {
            const uint32_t o = aother[a];
            if (level[o] != 0xFFFFFFFFu) continue;
            level[o] = level[u] + 1;
            depth = std::max(depth, level[o]);
            order.push_back(o);
        }
    }
    for (uint32_t v = 0; v < V; ++v)
        if (level[v] == 0xFFFFFFFFu) order.push_back(v);  // (not reachable from a gauge view: cannot happen, kept for safety)
    return depth;
}
// Aggregates by region growing: seeds in breadth-first order from the gauge views (`order`), each aggregate the first `target`
// free views a breadth-first walk from its seed reaches among the unassigned ones; left-overs below half the target join
// their smallest neighbouring aggregate.  false when the graph needs more than kMaxAggregates - 1 (many components).
static bool plan_two_level(uint32_t V, const std::vector<uint32_t>& ptr, const std::vector<uint32_t>& aedge,
                           const std::vector<uint32_t>& aother, const std::vector<int8_t>& asign, const std::vector<uint8_t>& is_root,
                           const std::vector<uint32_t>& order, TwoLevelPlan& out) {
    uint32_t n_free = 0;
    for (uint32_t v = 0; v < V; ++v) n_free += is_root[v] ? 0u : 1u;
    if (n_free == 0) return false;
    const uint32_t kNone = 0xFFFFFFFFu;
    uint32_t target = std::max<uint32_t>(32u, (n_free + 111u) / 112u);
    target = (target + kRowsPerBlock - 1) / kRowsPerBlock * kRowsPerBlock;
    std::vector<uint32_t> agg, disc, size, into, final_id;  // disc: the free views in the order the walks reached them
    uint32_t n_final = 0;
    auto label = [&](uint32_t g) {  // the aggregate a merged one ended up in
        while (into[g] != g) g = into[g] = into[into[g]];
        return g;
    };
    for (int attempt = 0;; ++attempt) {
        agg.assign(V, kNone);
        disc.clear();
        disc.reserve(n_free);
        size.clear();
        for (uint32_t seed : order) {
            if (is_root[seed] || agg[seed] != kNone) continue;
            const uint32_t g = (uint32_t)size.size();
            const size_t first = disc.size();
            disc.push_back(seed);
            agg[seed] = g;
            for (size_t h = first; h < disc.size() && disc.size() - first < target; ++h)
                for (uint32_t a = ptr[disc[h]]; a < ptr[disc[h] + 1] && disc.size() - first < target; ++a) {
                    const uint32_t o = aother[a];
                    if (is_root[o] || agg[o] != kNone) continue;
                    agg[o] = g;
                    disc.push_back(o);
                }
            size.push_back((uint32_t)(disc.size() - first));
        }
        const uint32_t n0 = (uint32_t)size.size();
        into.resize(n0);
        std::iota(into.begin(), into.end(), 0u);
        // small left-overs join a neighbour (their views are contiguous in disc: walk them through a start table)
        std::vector<uint32_t> start(n0 + 1, 0);
        for (uint32_t g = 0; g < n0; ++g) start[g + 1] = start[g] + size[g];
        for (uint32_t g = 0; g < n0; ++g) {
            if (size[g] >= target / 2) continue;  // (size of the aggregate as grown; one that already took others in is large enough)
            uint32_t best = kNone;
            for (uint32_t i = start[g]; i < start[g + 1]; ++i)
                for (uint32_t a = ptr[disc[i]]; a < ptr[disc[i] + 1]; ++a) {
                    const uint32_t o = aother[a];
                    if (is_root[o]) continue;
                    const uint32_t h = label(agg[o]);
                    if (h == g) continue;
                    if (best == kNone || size[h] < size[best] || (size[h] == size[best] && h < best)) best = h;
                }
            if (best == kNone) continue;  // a component of its own
            into[g] = best;
            size[best] += size[g];
        }
        final_id.assign(n0, kNone);
        n_final = 0;
        for (uint32_t g = 0; g < n0; ++g)
            if (into[g] == g) final_id[g] = n_final++;
        if (n_final <= kMaxAggregates - 1) break;  // (ids stay below kNoAggregate and below the 128 lanes of the coarse product)
        if (attempt == 5 || target >= n_free) return false;
        target *= 2;
    }
    for (uint32_t v = 0; v < V; ++v)
        if (agg[v] != kNone) agg[v] = final_id[label(agg[v])];
    // the views of every aggregate in the order they were reached (a stable counting sort of disc by aggregate)
    std::vector<uint32_t> first_of(n_final + 1, 0);
    for (uint32_t v : disc) ++first_of[agg[v] + 1];
    for (uint32_t g = 0; g < n_final; ++g) first_of[g + 1] += first_of[g];
    std::vector<uint32_t> sorted(disc.size()), fill(first_of.begin(), first_of.end() - 1);
    for (uint32_t v : disc) sorted[fill[agg[v]]++] = v;
    // rows: the aggregates one after the other, each padded to whole blocks; the gauge views last
    std::vector<uint32_t> view_row(V, kNone);
    out = TwoLevelPlan();
    out.n_agg = n_final;
    out.agg_block.push_back(0);
    for (uint32_t g = 0; g < n_final; ++g) {
        for (uint32_t i = first_of[g]; i < first_of[g + 1]; ++i) {
            view_row[sorted[i]] = (uint32_t)out.row_view.size();
            out.row_view.push_back(sorted[i]);
        }
        while (out.row_view.size() % kRowsPerBlock) out.row_view.push_back(kNone);
        out.agg_block.push_back((uint32_t)(out.row_view.size() / kRowsPerBlock));
    }
    out.agg_of_block.assign(out.row_view.size() / kRowsPerBlock, kNoAggregate);
    for (uint32_t g = 0; g < out.n_agg; ++g)
        for (uint32_t b = out.agg_block[g]; b < out.agg_block[g + 1]; ++b) out.agg_of_block[b] = (uint8_t)g;
    for (uint32_t v = 0; v < V; ++v)
        if (is_root[v]) {
            view_row[v] = (uint32_t)out.row_view.size();
            out.row_view.push_back(v);
        }
    while (out.row_view.size() % kRowsPerBlock) out.row_view.push_back(kNone);
    out.n_rows = (uint32_t)out.row_view.size();
    out.n_blocks = out.n_rows / kRowsPerBlock;
    out.agg_of_block.resize(out.n_blocks, kNoAggregate);
    out.root.assign(out.n_rows, 1);
    out.ptr.assign(out.n_rows + 1, 0);
    out.aedge.reserve(aedge.size()); out.aother.reserve(aedge.size()); out.asign.reserve(aedge.size()); out.aagg.reserve(aedge.size());
    for (uint32_t k = 0; k < out.n_rows; ++k) {
        const uint32_t v = out.row_view[k];
        if (v != kNone) {
            out.root[k] = is_root[v];
            for (uint32_t a = ptr[v]; a < ptr[v + 1]; ++a) {
                const uint32_t o = aother[a];
                out.aedge.push_back(aedge[a]);
                out.aother.push_back(view_row[o]);
                out.asign.push_back(asign[a]);
                out.aagg.push_back(is_root[o] ? kNoAggregate : (uint8_t)agg[o]);
            }
        }
        out.ptr[k + 1] = (uint32_t)out.aedge.size();
    }
    return true;
}

// The solve proper.  The rotation-graph edges are already in HBM at d_rot (RotEdgeDev[n_edges]); the host holds only
// their endpoints, the initial rotations and the root flags.
// PGI_ROTAVG_TIMING=1: wall clock of the host-visible phases on stderr (diagnosis only)
struct PhaseClock {
    bool on = std::getenv("PGI_ROTAVG_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void mark(const char* what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[rotavg timing] %-44s %8.3f ms\n", what, 1e3 * std::chrono::duration<double>(now - t).count());
        t = now;
    }
};

static int rotavg_solve(pgi_ctx* ctx, const pgi_rotavg_params& prm_in, uint32_t n_views, uint32_t n_edges, const uint32_t* h_src,
                        const uint32_t* h_dst, const RotEdgeDev* d_rot, const std::vector<double>& R,
                        const std::vector<uint8_t>& is_root, const std::vector<std::vector<ForestEdge>>& forest, double* h_R_out,
                        uint32_t* h_iters_out) {
    PhaseClock clk;
    // CSR adjacency, incidences in edge order
    std::vector<uint32_t> ptr(n_views + 1, 0), aedge(2 * (size_t)n_edges), aother(2 * (size_t)n_edges);
    std::vector<int8_t> asign(2 * (size_t)n_edges);
    for (uint32_t e = 0; e < n_edges; ++e) {
        ++ptr[h_src[e] + 1];
        ++ptr[h_dst[e] + 1];
    }
    for (uint32_t k = 0; k < n_views; ++k) ptr[k + 1] += ptr[k];
    {
        std::vector<uint32_t> fill(ptr.begin(), ptr.end() - 1);
        for (uint32_t e = 0; e < n_edges; ++e) {
            uint32_t a = fill[h_src[e]]++;
            aedge[a] = e; aother[a] = h_dst[e]; asign[a] = -1;
            a = fill[h_dst[e]]++;
            aedge[a] = e; aother[a] = h_src[e]; asign[a] = +1;
        }
    }
    const size_t V = n_views, E = n_edges;
    // Tree path (rot_solve_tree_kernel): views renumbered in depth-first preorder of the spanning forest, with subtree
    // sizes, the edge to the parent and the Euler-tour positions; the adjacency again in that numbering.
    // Only sparse graphs qualify: the forest holds V - 1 of the E edges, and on a dense graph (E / V = 21 in the k = 20
    // benchmark) it is a far worse preconditioner than Jacobi (1 380 iterations against 19) -- there a capped first L1
    // solve is simply a hard first step, not a reason to switch.
    const bool tree_ok = n_views <= kTreeViews && (uint64_t)n_edges <= 8ull * n_views;
    const uint32_t own = tree_ok ? std::max<uint32_t>(1u, (n_views + 1023u) / 1024u) : 0u;
    std::vector<uint32_t> t_n2o, t_pedge, t_size, t_enter, t_exit, t_ptr, t_edge, t_other;
    std::vector<int8_t> t_sign;
    if (tree_ok) {
        std::vector<uint32_t> o2n(V, 0xFFFFFFFFu);
        t_n2o.reserve(V); t_pedge.assign(V, 0xFFFFFFFFu); t_size.assign(V, 1u); t_enter.assign(V, 0u); t_exit.assign(V, 0u);
        uint32_t clock = 0;
        struct Frame { uint32_t v, next; };
        std::vector<Frame> stack;
        for (uint32_t root = 0; root < n_views; ++root) {
            if (!is_root[root]) continue;
            o2n[root] = (uint32_t)t_n2o.size();
            t_n2o.push_back(root);
            t_enter[o2n[root]] = clock++;
            stack.assign(1, Frame{root, 0u});
            while (!stack.empty()) {
                Frame& f = stack.back();
                if (f.next < forest[f.v].size()) {
                    const ForestEdge& pr = forest[f.v][f.next++];
                    const uint32_t c = pr.other & 0x7FFFFFFFu;
                    if (o2n[c] != 0xFFFFFFFFu) continue;  // the parent
                    o2n[c] = (uint32_t)t_n2o.size();
                    t_n2o.push_back(c);
                    t_pedge[o2n[c]] = pr.edge;
                    t_enter[o2n[c]] = clock++;
                    stack.push_back(Frame{c, 0u});
                } else {
                    const uint32_t k = o2n[f.v];
                    t_size[k] = (uint32_t)t_n2o.size() - k;
                    t_exit[k] = clock++;
                    stack.pop_back();
                }
            }
        }
        t_ptr.assign(V + 1, 0); t_edge.resize(2 * E); t_other.resize(2 * E); t_sign.resize(2 * E);
        for (uint32_t k = 0; k < n_views; ++k) {
            const uint32_t u = t_n2o[k];
            uint32_t o = t_ptr[k];
            for (uint32_t a = ptr[u]; a < ptr[u + 1]; ++a, ++o) {
                t_edge[o] = aedge[a];
                t_other[o] = o2n[aother[a]];
                t_sign[o] = asign[a];
            }
            t_ptr[k + 1] = o;
        }
    }
    // one allocation, carved
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        const size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    };
    const size_t o_R = carve(V * 72), o_ptr = carve((V + 1) * 4),
                 o_aedge = carve(2 * E * 4), o_aother = carve(2 * E * 4), o_asign = carve(2 * E),
                 o_root = carve(V), o_omega = carve(E * 24), o_w = carve(E * 8), o_diag = carve(V * 8),
                 o_x = carve(V * 24), o_r = carve(V * 24), o_p = carve(V * 24), o_Ap = carve(V * 24), o_r2 = carve(V * 24), o_q2 = carve(V * 24), o_s = carve(V * 24),
                 o_s2 = carve(V * 24),
                 o_stats = carve(16), o_norm = carve(3 * 8 * ((V + kCgBlock - 1) / kCgBlock)), o_cg = carve(4 * sizeof(CgState)),  // two launch parities + the record the host reads
                 o_part = carve(2 * 6 * 8 * ((V + kRowsPerBlock - 1) / kRowsPerBlock + 1)),  // block partials, by launch parity
                 // tree path
                 o_tptr = carve((V + 1) * 4), o_tedge = carve(2 * E * 4), o_tother = carve(2 * E * 4), o_tsign = carve(2 * E),
                 o_tn2o = carve(V * 4), o_tpe = carve(V * 4), o_tsz = carve(V * 4), o_ten = carve(V * 4), o_tex = carve(V * 4),
                 o_aw = carve(2 * E * sizeof(TreeIncidence)), o_its = carve(32);
    char* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, off));
    struct Guard {
        char* p;
        ~Guard() { (void)hipFree(p); }
    } guard{d};
    hipStream_t st = ctx->stream;
    HIP_TRY(hipMemcpyAsync(d + o_R, R.data(), V * 72, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d + o_ptr, ptr.data(), (V + 1) * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d + o_aedge, aedge.data(), 2 * E * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d + o_aother, aother.data(), 2 * E * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d + o_asign, asign.data(), 2 * E, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d + o_root, is_root.data(), V, hipMemcpyHostToDevice, st));
    size_t tree_lds = 0;
    if (tree_ok) {
        HIP_TRY(hipMemcpyAsync(d + o_tptr, t_ptr.data(), (V + 1) * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d + o_tedge, t_edge.data(), 2 * E * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d + o_tother, t_other.data(), 2 * E * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d + o_tsign, t_sign.data(), 2 * E, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d + o_tn2o, t_n2o.data(), V * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d + o_tpe, t_pedge.data(), V * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d + o_tsz, t_size.data(), V * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d + o_ten, t_enter.data(), V * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d + o_tex, t_exit.data(), V * 4, hipMemcpyHostToDevice, st));
        tree_lds = ((size_t)3 * 1024 * own + 16) * 8;
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&rot_solve_tree_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)tree_lds));
    }
    if (clk.on) (void)hipStreamSynchronize(st);
    clk.mark("solve: adjacency, allocation, uploads");
    bool use_tree = false;  // set once the Jacobi-preconditioned solve has run into its iteration cap
    uint32_t single_wg_views = kSingleWgViews;
    if (const char* e = std::getenv("PGI_ROTAVG_SINGLE_WG_VIEWS")) single_wg_views = (uint32_t)std::max(0, std::atoi(e));  // experiments
    const bool trace = std::getenv("PGI_ROTAVG_TRACE") != nullptr;
    pgi_rotavg_params prm = prm_in;
    if (const char* e = std::getenv("PGI_ROTAVG_CG_ITERS")) prm.cg_iters = (uint32_t)std::max(1, std::atoi(e));  // experiments
    const double sigma = prm.sigma_deg * 3.14159265358979323846 / 180.0;
    uint32_t iters = 0;
    uint32_t predicted = 0;   // PCG iterations of the previous outer step's solve (multi-workgroup Jacobi branch)
    uint32_t predicted2 = 0;  // the same for the two-level solves

    // Looks at the record between chunks of launches (each look a round trip): after 16, 48, 112, 176, ... launches -- except
    // that a solve whose predecessor in the outer loop converged after n iterations first runs n + 2 + n / 8 launches in one go
    // (consecutive IRLS systems differ by their weights only and need nearly the same count; launches after convergence are
    // no-ops, so WHERE the host looks changes no bit of the result), and one whose predecessor ran into the cap runs the whole
    // cap.  Graphs that may still switch to the tree path keep every look from 112 on: the switch is decided there.
    // iterate(0): one iteration launch; iterate(2): the reduce-only launch that fills `record`; finish(record): the step-norm and
    // update launches gated on the record, and the copy of the norm partials -- queued before the host looks, so that a solve
    // found converged is already applied (one round trip per outer step).  Returns < 0 on a HIP error, else 1 / 0 = converged
    // (and applied) or not; hs = the last record.
    auto pcg_looks = [&](auto&& iterate, auto&& finish, const CgState* record, uint32_t cap, uint32_t& pred, bool slow_break, CgState& hs) -> int {
        int done = 0;
        uint32_t ci = 0, boundary = 16, step = 16;
        const uint32_t guess = pred + 2 + pred / 8;
        uint32_t jump = (pred && (!slow_break || guess <= 64)) ? std::min<uint32_t>(guess, cap) : 0;
        while (ci < cap) {
            uint32_t target;
            bool scheduled = true;
            if (jump > ci) {
                target = jump;
                jump = 0;
                scheduled = false;
            } else {
                while (boundary <= ci) {
                    step = std::min<uint32_t>(2 * step, 64);
                    boundary += step;
                }
                target = std::min<uint32_t>(boundary, cap);
            }
            for (; ci < target; ++ci) iterate(0);  // kernels no-op once converged
            iterate(2);
            {
                const int rc = finish(record);
                if (rc < 0) return rc;
            }
            HIP_TRY(hipMemcpyAsync(&hs, record, sizeof hs, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            done = hs.done;
            pred = done ? (uint32_t)hs.iters : cap;
            if (done) break;
            // hopeless for this preconditioner: after 64 iterations a well-conditioned (dense) graph has gained ten orders of
            // magnitude, a sequence-like one not even three
            if (scheduled && slow_break && ci >= 64) {
                bool slow = false;
                for (int c = 0; c < 3; ++c) slow |= hs.rz[c] > 1e-6 * hs.rz0[c];
                if (slow) break;
            }
        }
        return done;
    };

    // ---- two-level preconditioner (see cg2_iteration_kernel).  WHEN: the graph is too dense for the tree path, too large for
    // the one-workgroup solve, and DEEP -- the breadth-first walk from the gauge views needs kTwoLevelDepth levels or more, as on a
    // walk or a ring (V / reach levels), where Jacobi needs hundreds of iterations whatever the weights; a densely connected
    // graph has a handful of levels, Jacobi converges in ~20 iterations there and a capped first L1 solve is merely a hard
    // first step.  PGI_ROTAVG_TWO_LEVEL: 0 never, 1 (default) by that rule, 2 always (tests).
    constexpr uint32_t kTwoLevelDepth = 12;
    const char* two_env = std::getenv("PGI_ROTAVG_TWO_LEVEL");
    const int two_mode = two_env ? std::atoi(two_env) : 1;
    bool use_two = false;
    TwoLevelPlan plan;
    struct Guard2 {
        char* p = nullptr;
        ~Guard2() { if (p) (void)hipFree(p); }
    } guard2;
    size_t t_ptr2 = 0, t_edge2 = 0, t_other2 = 0, t_sign2 = 0, t_agg2 = 0, t_root2 = 0, t_view2 = 0, t_ablk2 = 0, t_aob2 = 0, t_diag2 = 0, t_x2 = 0,
           t_p2 = 0, t_r2[2] = {0, 0}, t_q2[2] = {0, 0}, t_s2[2] = {0, 0}, t_part2 = 0, t_rpart2 = 0, t_cs2 = 0, t_Ac2 = 0, t_Ainv2 = 0, t_cg2 = 0,
           t_status2 = 0, t_norm2 = 0;
    size_t inv_lds = 0;
    std::vector<uint32_t> bfs_order;
    auto ensure_plan = [&]() -> int {  // 0: ready; 1: this graph cannot have one; < 0: error
        if (guard2.p) return 0;
        if (!plan_two_level(n_views, ptr, aedge, aother, asign, is_root, bfs_order, plan)) return 1;
        const size_t N = plan.n_rows, NB = plan.n_blocks, NA = plan.n_agg, I = plan.aedge.size();
        size_t o2 = 0;
        auto carve2 = [&](size_t bytes) {
            const size_t o = o2;
            o2 += (bytes + 255) & ~(size_t)255;
            return o;
        };
        t_ptr2 = carve2((N + 1) * 4); t_edge2 = carve2(I * 4); t_other2 = carve2(I * 4); t_sign2 = carve2(I); t_agg2 = carve2(I);
        t_root2 = carve2(N); t_view2 = carve2(N * 4); t_ablk2 = carve2((NA + 1) * 4); t_aob2 = carve2(NB); t_diag2 = carve2(N * 8);
        t_x2 = carve2(N * 24); t_p2 = carve2(N * 24);
        for (int b = 0; b < 2; ++b) { t_r2[b] = carve2(N * 24); t_q2[b] = carve2(N * 24); t_s2[b] = carve2(N * 24); }
        t_part2 = carve2(2 * 9 * 8 * (NB + 1)); t_rpart2 = carve2(3 * 8 * NB); t_cs2 = carve2(2 * 9 * 8 * NA); t_Ac2 = carve2(NA * NA * 8);
        t_Ainv2 = carve2(NA * NA * 8); t_cg2 = carve2(4 * sizeof(CgState)); t_status2 = carve2(16);
        t_norm2 = carve2(3 * 8 * ((N + kCgBlock - 1) / kCgBlock));
        HIP_TRY(hipMalloc((void**)&guard2.p, o2));
        char* d2 = guard2.p;
        HIP_TRY(hipMemcpyAsync(d2 + t_ptr2, plan.ptr.data(), (N + 1) * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d2 + t_edge2, plan.aedge.data(), I * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d2 + t_other2, plan.aother.data(), I * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d2 + t_sign2, plan.asign.data(), I, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d2 + t_agg2, plan.aagg.data(), I, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d2 + t_root2, plan.root.data(), N, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d2 + t_view2, plan.row_view.data(), N * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d2 + t_ablk2, plan.agg_block.data(), (NA + 1) * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d2 + t_aob2, plan.agg_of_block.data(), NB, hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));  // (the plan's vectors are read by the copies)
        inv_lds = NA * NA * 8;
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&cg2_coarse_invert_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)inv_lds));
        return 0;
    };
    // relative tolerance of every inner solve (kInnerTolerance above; PGI_ROTAVG_CG_TOL overrides)
    double cg_tol = kInnerTolerance;
    if (const char* e = std::getenv("PGI_ROTAVG_CG_TOL")) cg_tol = std::min(1e-2, std::max(1e-14, std::atof(e)));
    double two_iters = 0;  // iterations of the last two-level solve (trace)
    std::vector<double> norm_host;  // block partials of the step norm (multi-workgroup, tree and two-level paths)
    // one two-level solve of the current outer step, applied to the rotations, norm_host filled; 1 when the coarse matrix could
    // not be inverted (nothing was applied, the caller falls back), < 0 on a HIP error
    auto solve_two_level = [&]() -> int {
        char* d2 = guard2.p;
        const uint32_t N = plan.n_rows, NB = plan.n_blocks, NA = plan.n_agg;
        const uint32_t* ptr2 = (const uint32_t*)(d2 + t_ptr2);
        const uint32_t* edge2 = (const uint32_t*)(d2 + t_edge2);
        const uint32_t* other2 = (const uint32_t*)(d2 + t_other2);
        const uint8_t* agg2 = (const uint8_t*)(d2 + t_agg2);
        const uint8_t* root2 = (const uint8_t*)(d2 + t_root2);
        double* diag2 = (double*)(d2 + t_diag2);
        double* x2 = (double*)(d2 + t_x2);
        double* p2 = (double*)(d2 + t_p2);
        double* rbuf[2] = {(double*)(d2 + t_r2[0]), (double*)(d2 + t_r2[1])};
        double* qbuf[2] = {(double*)(d2 + t_q2[0]), (double*)(d2 + t_q2[1])};
        double* sbuf[2] = {(double*)(d2 + t_s2[0]), (double*)(d2 + t_s2[1])};
        double* part = (double*)(d2 + t_part2);
        double* pbuf[2] = {part, part + 9 * ((size_t)NB + 1)};
        double* cs = (double*)(d2 + t_cs2);
        double* csbuf[2] = {cs, cs + 9 * (size_t)NA};
        CgState* cst = (CgState*)(d2 + t_cg2);
        CgState* record = cst + 2;
        hipLaunchKernelGGL(cg2_init_kernel, dim3(NB), dim3(kCgBlock), 0, st, N, ptr2, edge2, (const int8_t*)(d2 + t_sign2), root2,
                           (const double*)(d + o_omega), (const double*)(d + o_w), diag2, x2, rbuf[0], p2, (double*)(d2 + t_rpart2), qbuf[0], sbuf[0], cst);
        hipLaunchKernelGGL(cg2_coarse_assemble_kernel, dim3(NA), dim3(kMaxAggregates * kAsmSegments), 0, st, NA, (const uint32_t*)(d2 + t_ablk2), ptr2, edge2,
                           agg2, root2, (const double*)(d + o_w), (const double*)diag2, (double*)(d2 + t_Ac2));
        hipLaunchKernelGGL(cg2_coarse_invert_kernel, dim3(1), dim3(1024), inv_lds, st, NA, (const double*)(d2 + t_Ac2), (double*)(d2 + t_Ainv2),
                           (int*)(d2 + t_status2));
        int cur = 0;
        uint32_t launch = 0;
        auto iterate = [&](int mode) {
            const int prev = (int)((launch + 1) & 1u), next = (int)(launch & 1u);
            hipLaunchKernelGGL(cg2_iteration_kernel, dim3(mode == 2 ? 1u : NB), dim3(kCgBlock), 0, st, N, ptr2, edge2, other2, agg2, root2,
                               (const double*)(d + o_w), (const double*)diag2, x2, p2, (const double*)rbuf[cur], rbuf[cur ^ 1],
                               (const double*)qbuf[cur], qbuf[cur ^ 1], (const double*)sbuf[cur], sbuf[cur ^ 1], mode, launch == 1 ? 1 : 0, cg_tol,
                               (const double*)pbuf[prev], pbuf[next], NB, (const CgState*)(cst + prev), mode == 2 ? record : cst + next, NA,
                               (const uint32_t*)(d2 + t_ablk2), (const uint8_t*)(d2 + t_aob2), (const double*)(d2 + t_Ainv2),
                               (const double*)(d2 + t_rpart2), (const double*)csbuf[prev], csbuf[next]);
            if (mode != 2) {
                cur ^= 1;
                ++launch;
            }
        };
        int status = 0;
        const int* veto = (const int*)(d2 + t_status2);
        auto finish = [&](const CgState* gate) -> int {
            const uint32_t nbn = (N + kCgBlock - 1) / kCgBlock;
            hipLaunchKernelGGL(cg_step_norm_kernel, dim3(nbn), dim3(kCgBlock), 0, st, N, (const double*)x2, (double*)(d2 + t_norm2), gate, veto);
            hipLaunchKernelGGL(rot_update_rows_kernel, dim3((N + 255) / 256), dim3(256), 0, st, N, (const double*)x2,
                               (const uint32_t*)(d2 + t_view2), (double*)(d + o_R), gate, veto);
            norm_host.resize(3 * (size_t)nbn);
            HIP_TRY(hipMemcpyAsync(norm_host.data(), d2 + t_norm2, norm_host.size() * 8, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(&status, d2 + t_status2, sizeof status, hipMemcpyDeviceToHost, st));
            return 0;
        };
        iterate(1);
        CgState hs{};
        const int done = pcg_looks(iterate, finish, record, prm.cg_iters, predicted2, false, hs);
        if (done < 0) return done;
        if (!done) {  // ran into the cap: the truncated solution is the step
            const int rc = finish(nullptr);
            if (rc < 0) return rc;
            HIP_TRY(hipStreamSynchronize(st));
        }
        if (status) return 1;
        two_iters = hs.iters;
        return 0;
    };
    if (two_mode == 2 || (two_mode == 1 && !tree_ok && n_views > single_wg_views)) {
        const uint32_t depth = bfs_from_gauge_views(n_views, ptr, aother, is_root, bfs_order);
        if (two_mode == 2 || depth >= kTwoLevelDepth) {
            const int rc = ensure_plan();
            if (rc < 0) return rc;
            use_two = rc == 0;
        }
        if (clk.on) std::fprintf(stderr, "[rotavg timing] breadth-first depth %u, two-level %s (%u aggregates, %u rows)\n", depth, use_two ? "on" : "off",
                                 plan.n_agg, plan.n_rows);
        clk.mark("solve: two-level plan");
    }

    for (uint32_t it = 0; it < prm.l1_iters + prm.irls_iters; ++it) {
        hipLaunchKernelGGL(rot_residual_kernel, dim3((n_edges + 255) / 256), dim3(256), 0, st, d_rot, n_edges,
                           (const double*)(d + o_R), it < prm.l1_iters ? 1 : 0, sigma, (double*)(d + o_omega), (double*)(d + o_w));
        // step norm and update of a solution in view numbering (x), gated on a PCG record or not
        auto finish_views = [&](const CgState* gate) -> int {
            const uint32_t nb = (n_views + kCgBlock - 1) / kCgBlock;
            hipLaunchKernelGGL(cg_step_norm_kernel, dim3(nb), dim3(kCgBlock), 0, st, n_views, (const double*)(d + o_x), (double*)(d + o_norm), gate,
                               (const int*)nullptr);
            hipLaunchKernelGGL(rot_update_kernel, dim3((n_views + 255) / 256), dim3(256), 0, st, n_views, (const double*)(d + o_x),
                               (double*)(d + o_R), gate, (const int*)nullptr);
            norm_host.resize(3 * (size_t)nb);
            HIP_TRY(hipMemcpyAsync(norm_host.data(), d + o_norm, norm_host.size() * 8, hipMemcpyDeviceToHost, st));
            return 0;
        };
        auto solve_with_tree = [&]() {  // writes every entry of x
            hipLaunchKernelGGL(rot_solve_tree_kernel, dim3(3), dim3(1024), tree_lds, st, n_views, own, (const uint32_t*)(d + o_tptr),
                               (const uint32_t*)(d + o_tedge), (const uint32_t*)(d + o_tother), (const int8_t*)(d + o_tsign),
                               (const uint32_t*)(d + o_tpe), (const uint32_t*)(d + o_tsz), (const uint32_t*)(d + o_ten),
                               (const uint32_t*)(d + o_tex), (const uint32_t*)(d + o_tn2o), (const double*)(d + o_omega),
                               (const double*)(d + o_w), std::max<uint32_t>(prm.cg_iters, 1000u), cg_tol, (TreeIncidence*)(d + o_aw),
                               (double*)(d + o_x), (double*)(d + o_its));
        };
        // multi-workgroup Jacobi PCG: 1 = converged (and applied: finish_views ran on the record), 0 = not, < 0 on an error
        auto solve_jacobi = [&](CgState& hs) -> int {
            CgState* cst = (CgState*)(d + o_cg);
            double* part = (double*)(d + o_part);
            const uint32_t nbp = (n_views + kRowsPerBlock - 1) / kRowsPerBlock;  // kRowLanes lanes per view
            double* rbuf[2] = {(double*)(d + o_r), (double*)(d + o_r2)};
            double* qbuf[2] = {(double*)(d + o_Ap), (double*)(d + o_q2)};
            double* sbuf[2] = {(double*)(d + o_s), (double*)(d + o_s2)};
            double* pbuf[2] = {part, part + 6 * ((size_t)nbp + 1)};
            CgState* record = cst + 2;  // written by the reduce-only launch, read by the host
            hipLaunchKernelGGL(cg_init_kernel, dim3(nbp), dim3(kCgBlock), 0, st, n_views, (const uint32_t*)(d + o_ptr),
                               (const uint32_t*)(d + o_aedge), (const int8_t*)(d + o_asign), (const uint8_t*)(d + o_root),
                               (const double*)(d + o_omega), (const double*)(d + o_w), (double*)(d + o_diag),
                               (double*)(d + o_x), (double*)(d + o_r), (double*)(d + o_p), qbuf[0], sbuf[0], cst);
            int cur = 0;           // buffers holding the vectors of the last finished iteration
            uint32_t launch = 0;   // launches so far: launch j reads parity (j - 1) & 1 of partials / state, writes parity j & 1
            auto iterate = [&](int mode) {  // 1: init pass, 0: iteration, 2: reduce the last launch's partials into `record`
                const int prev = (int)((launch + 1) & 1u), next = (int)(launch & 1u);
                hipLaunchKernelGGL(cg_iteration_kernel, dim3(mode == 2 ? 1u : nbp), dim3(kCgBlock), 0, st, n_views, (const uint32_t*)(d + o_ptr),
                                   (const uint32_t*)(d + o_aedge), (const uint32_t*)(d + o_aother), (const uint8_t*)(d + o_root),
                                   (const double*)(d + o_w), (const double*)(d + o_diag), (double*)(d + o_x), (double*)(d + o_p),
                                   (const double*)rbuf[cur], rbuf[cur ^ 1], (const double*)qbuf[cur], qbuf[cur ^ 1],
                                   (const double*)sbuf[cur], sbuf[cur ^ 1], mode, launch == 1 ? 1 : 0, cg_tol, (const double*)pbuf[prev],
                                   pbuf[next], nbp, (const CgState*)(cst + prev), mode == 2 ? record : cst + next);
                if (mode != 2) {
                    cur ^= 1;
                    ++launch;
                }
            };
            iterate(1);
            return pcg_looks(iterate, finish_views, record, prm.cg_iters, predicted, tree_ok, hs);
        };
        bool used_tree_this_iter = false, used_two_this_iter = false, single_now = false;
        if (use_two) {
            const int rc = solve_two_level();
            if (rc < 0) return rc;
            if (rc == 1) use_two = false;  // a coarse matrix that could not be inverted: this graph stays with the one-level solvers
            else used_two_this_iter = true;
        }
        if (used_two_this_iter) {
        } else if (use_tree) {
            solve_with_tree();
            used_tree_this_iter = true;
        } else if (n_views <= single_wg_views) {
            uint32_t row_lanes = 16;  // as many lanes per view as keep all views in one pass of the 1024 threads
            while (row_lanes > 1 && (uint64_t)n_views * row_lanes > 1024u) row_lanes >>= 1;
            hipLaunchKernelGGL(rot_solve_kernel, dim3(1), dim3(1024), 0, st, n_views, (const uint32_t*)(d + o_ptr),
                               (const uint32_t*)(d + o_aedge), (const uint32_t*)(d + o_aother), (const int8_t*)(d + o_asign),
                               (const uint8_t*)(d + o_root), (const double*)(d + o_omega), (const double*)(d + o_w),
                               prm.cg_iters, cg_tol, row_lanes, (double*)(d + o_diag), (double*)(d + o_x), (double*)(d + o_r),
                               (double*)(d + o_p), (double*)(d + o_Ap), (double*)(d + o_stats));
            single_now = true;
            if (tree_ok) {  // a solve that ran into the cap sends this graph to the tree path (see the multi-workgroup branch)
                double probe[2] = {0, 0};
                HIP_TRY(hipMemcpyAsync(probe, d + o_stats, 16, hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                if (probe[1] >= (double)prm.cg_iters) {
                    use_tree = true;
                    solve_with_tree();
                    used_tree_this_iter = true;
                    single_now = false;
                }
            }
        } else {  // multi-workgroup PCG: launches are cheap next to a one-CU solve at this size
            CgState hs{};
            const int done = solve_jacobi(hs);
            if (done < 0) return done;
            // not converged within the cap: the graph is sparse / sequence-like and Jacobi is the wrong preconditioner for
            // it.  This step is solved again, and all following ones, by the tree-preconditioned kernel, so that the
            // iteration follows the exact-solve trajectory from the start.
            if (!done && tree_ok) {
                use_tree = true;
                solve_with_tree();
                used_tree_this_iter = true;
            } else if (!done) {  // the truncated solution is the step
                const int rc = finish_views(nullptr);
                if (rc < 0) return rc;
                HIP_TRY(hipStreamSynchronize(st));
            }
        }
        double stats[2] = {0, 0};
        const bool tree_now = used_tree_this_iter;
        if (single_now) {
            hipLaunchKernelGGL(rot_update_kernel, dim3((n_views + 255) / 256), dim3(256), 0, st, n_views, (const double*)(d + o_x),
                               (double*)(d + o_R), (const CgState*)nullptr, (const int*)nullptr);
            HIP_TRY(hipMemcpyAsync(stats, d + o_stats, 16, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
        } else {
            if (tree_now) {
                const int rc = finish_views(nullptr);
                if (rc < 0) return rc;
                HIP_TRY(hipStreamSynchronize(st));
            }
            double sum = 0;  // mean |d| = sum / V
            for (size_t b2 = 0; b2 < norm_host.size(); b2 += 3) sum += norm_host[b2];
            stats[0] = sum / (double)n_views;
        }
        iters = it + 1;
        if (trace) {  // PGI_ROTAVG_TRACE=1: one line per outer iteration
            double cg_it = stats[1];
            if (used_two_this_iter) {
                cg_it = two_iters;
            } else if (tree_now) {
                double its[3];
                HIP_TRY(hipMemcpyAsync(its, d + o_its, 24, hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                cg_it = -std::max(its[0], std::max(its[1], its[2]));  // printed negative: tree-preconditioned iterations
            } else if (n_views > single_wg_views) {
                CgState hs;
                HIP_TRY(hipMemcpyAsync(&hs, d + o_cg + 2 * sizeof(CgState), sizeof hs, hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                cg_it = hs.iters;
            }
            std::fprintf(stderr, "[rotavg] outer %2u (%s): %3.0f PCG iterations%s, mean step %.3e rad\n", it, it < prm.l1_iters ? "L1" : "IRLS", cg_it,
                         used_two_this_iter ? " (two-level)" : "", stats[0]);
        }
        if (stats[0] < prm.tol) break;
    }
    clk.mark("solve: outer iterations");
    HIP_TRY(hipMemcpyAsync(h_R_out, d + o_R, V * 72, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    clk.mark("solve: rotations back");
    if (h_iters_out) *h_iters_out = iters;
    return PGI_SUCCESS;
}

}  // namespace pgi

using namespace pgi;

extern "C" {

void pgi_default_rotavg_params(pgi_rotavg_params* p) {
    p->l1_iters = 5;
    p->irls_iters = 100;
    p->cg_iters = 200;
    p->sigma_deg = 5.0;
    p->tol = 1e-8;
}

int pgi_rotation_average(pgi_ctx* ctx, const pgi_rot_edge* h_edges, uint32_t n_edges, uint32_t n_views,
                         const pgi_rotavg_params* prm_in, double* h_R_out, uint32_t* h_iters_out) {
    if (!ctx || !h_R_out || (n_edges && !h_edges)) return fail(PGI_ERR_INVALID, "null argument");
    pgi_rotavg_params prm;
    if (prm_in) prm = *prm_in; else pgi_default_rotavg_params(&prm);
    for (uint32_t e = 0; e < n_edges; ++e)
        if (h_edges[e].src >= n_views || h_edges[e].dst >= n_views || h_edges[e].src == h_edges[e].dst)
            return fail(PGI_ERR_INVALID, "edge endpoints out of range");
    if (n_views == 0) return PGI_SUCCESS;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    PhaseClock clk;
    std::vector<uint32_t> src(n_edges), dst(n_edges), tree;
    std::vector<double> wt(n_edges);
    for (uint32_t e = 0; e < n_edges; ++e) {
        src[e] = h_edges[e].src;
        dst[e] = h_edges[e].dst;
        wt[e] = h_edges[e].weight;
    }
    std::vector<std::vector<ForestEdge>> adj;
    spanning_forest(n_views, src.data(), dst.data(), wt.data(), n_edges, tree, adj);
    clk.mark("host edges: maximum-weight spanning forest");
    std::vector<double> R;
    std::vector<uint8_t> is_root;
    forest_bfs_init(n_views, adj, [&](uint32_t e) { return h_edges[e].R; }, R, is_root);
    clk.mark("host edges: forest initialisation");
    if (h_iters_out) *h_iters_out = 0;
    if (n_edges == 0) {
        memcpy(h_R_out, R.data(), R.size() * 8);
        return PGI_SUCCESS;
    }
    static_assert(sizeof(RotEdgeDev) == sizeof(pgi_rot_edge), "edge layout");
    RotEdgeDev* d_rot = nullptr;
    HIP_TRY(hipMalloc((void**)&d_rot, (size_t)n_edges * sizeof(RotEdgeDev)));
    struct Guard {
        void* p;
        ~Guard() { (void)hipFree(p); }
    } guard{d_rot};
    HIP_TRY(hipMemcpyAsync(d_rot, h_edges, (size_t)n_edges * sizeof(pgi_rot_edge), hipMemcpyHostToDevice, ctx->stream));
    if (clk.on) (void)hipStreamSynchronize(ctx->stream);
    clk.mark("host edges: edge table to the device");
    return rotavg_solve(ctx, prm, n_views, n_edges, src.data(), dst.data(), d_rot, R, is_root, adj, h_R_out, h_iters_out);
}

int pgi_rotation_average_edges(pgi_ctx* ctx, const pgi_edge* d_edges, const uint32_t* h_src, const uint32_t* h_dst,
                               const uint32_t* h_rows, uint32_t n_pairs, uint32_t n_views, const pgi_rotavg_params* prm_in,
                               double* h_R_out, uint32_t* h_iters_out, uint32_t* h_edges_used) {
    if (!ctx || !h_R_out || (n_pairs && (!d_edges || !h_src || !h_dst))) return fail(PGI_ERR_INVALID, "null argument");
    pgi_rotavg_params prm;
    if (prm_in) prm = *prm_in; else pgi_default_rotavg_params(&prm);
    if (h_iters_out) *h_iters_out = 0;
    if (h_edges_used) *h_edges_used = 0;
    if (n_views == 0) return PGI_SUCCESS;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    PhaseClock clk;
    // (status, n_inl) of every record: an 8-byte column of the 200-byte table
    struct Meta { int32_t status; uint32_t n_inl; };
    std::vector<Meta> meta(n_pairs);
    if (n_pairs) {
        HIP_TRY(hipMemcpy2DAsync(meta.data(), sizeof(Meta), (const char*)d_edges + offsetof(pgi_edge, status), sizeof(pgi_edge),
                                 sizeof(Meta), n_pairs, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    std::vector<uint32_t> idx, src, dst;
    std::vector<double> wt;
    for (uint32_t p = 0; p < n_pairs; ++p) {
        if (meta[p].status != PGI_EDGE_OK) continue;
        if (h_src[p] >= n_views || h_dst[p] >= n_views || h_src[p] == h_dst[p])
            return fail(PGI_ERR_INVALID, "edge endpoints out of range");
        idx.push_back(p);
        src.push_back(h_src[p]);
        dst.push_back(h_dst[p]);
        wt.push_back(h_rows ? (double)meta[p].n_inl / (double)std::max<uint32_t>(1u, h_rows[p]) : 1.0);
    }
    const uint32_t nE = (uint32_t)idx.size();
    if (h_edges_used) *h_edges_used = nE;
    clk.mark("edges: status column to the host, edge list");
    std::vector<uint32_t> tree;
    std::vector<std::vector<ForestEdge>> adj;
    spanning_forest(n_views, src.data(), dst.data(), wt.data(), nE, tree, adj);
    clk.mark("edges: maximum-weight spanning forest (host)");
    std::vector<double> R;
    std::vector<uint8_t> is_root;
    if (nE == 0) {
        forest_bfs_init(n_views, adj, [&](uint32_t) { return (const double*)nullptr; }, R, is_root);
        memcpy(h_R_out, R.data(), R.size() * 8);
        return PGI_SUCCESS;
    }
    const size_t nT = tree.size();
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_rot = 0, o_idx = up((size_t)nE * sizeof(RotEdgeDev)), o_src = o_idx + up((size_t)nE * 4),
                 o_dst = o_src + up((size_t)nE * 4), o_wt = o_dst + up((size_t)nE * 4), o_tree = o_wt + up((size_t)nE * 8),
                 o_treeR = o_tree + up(nT * 4), total = o_treeR + up(nT * 72);
    char* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, total));
    struct Guard {
        void* p;
        ~Guard() { (void)hipFree(p); }
    } guard{d};
    HIP_TRY(hipMemcpyAsync(d + o_idx, idx.data(), (size_t)nE * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d + o_src, src.data(), (size_t)nE * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d + o_dst, dst.data(), (size_t)nE * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d + o_wt, wt.data(), (size_t)nE * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d + o_tree, tree.data(), nT * 4, hipMemcpyHostToDevice, st));
    RotEdgeDev* d_rot = (RotEdgeDev*)(d + o_rot);
    hipLaunchKernelGGL(rot_edges_from_table_kernel, dim3((nE + 255) / 256), dim3(256), 0, st, d_edges, (const uint32_t*)(d + o_idx),
                       (const uint32_t*)(d + o_src), (const uint32_t*)(d + o_dst), (const double*)(d + o_wt), nE, d_rot);
    hipLaunchKernelGGL(rot_gather_R_kernel, dim3(((uint32_t)nT + 255) / 256), dim3(256), 0, st, d_rot, (const uint32_t*)(d + o_tree),
                       (uint32_t)nT, (double*)(d + o_treeR));
    std::vector<double> treeR(nT * 9);
    HIP_TRY(hipMemcpyAsync(treeR.data(), d + o_treeR, nT * 72, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    std::vector<uint32_t> slot(nE, 0u);  // edge -> position in the forest list
    for (size_t k = 0; k < nT; ++k) slot[tree[k]] = (uint32_t)k;
    clk.mark("edges: device edge table, forest rotations back");
    forest_bfs_init(n_views, adj, [&](uint32_t e) { return &treeR[9 * (size_t)slot[e]]; }, R, is_root);
    clk.mark("edges: forest initialisation (host)");
    return rotavg_solve(ctx, prm, n_views, nE, src.data(), dst.data(), d_rot, R, is_root, adj, h_R_out, h_iters_out);
}

}  // extern "C"
