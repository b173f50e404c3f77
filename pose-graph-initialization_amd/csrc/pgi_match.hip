// pgi_match.hip -- descriptor matching on the exact-f32 matrix cores.
//
// Replaces feature_utils.h:135-202 of the reference (two cv::BFMatcher kNN(2) searches, Lowe ratio 0.90,
// mutual-best test, sort by ratio).  The K1 x K2 x 128 inner products are the only GEMM-shaped work next to the
// hot path; they run on v_mfma_f32_32x32x2_f32, whose accumulation is bitwise an fmaf chain over k = 0..127, so
// the matches are bit-identical to the scalar specification (oracle/pgi_oracle.c: pgo_match_descriptors).
//
//   desc_prepare_kernel   n x 128 row-major -> 128 x n_pad transposed (+ squared norms), once per image
//   desc_top2_kernel      per (128-row block, column split): distances tile by tile, running row top-2 in
//                         registers, column best through 64-bit atomicMin on (d2 bits, row)
//   match_select_kernel   per pair: merge the splits, ratio + mutual test, bitonic sort by (ratio, row)
#include <algorithm>
#include <cfloat>
#include <type_traits>
#include <utility>
#include <vector>
#include "pgi_internal.hpp"

#ifndef PGI_SCREEN_WAVES
#define PGI_SCREEN_WAVES 8
#endif
namespace {
constexpr int kD = PGI_DESC_DIM;   // 128
constexpr int kPadRows = 256;      // keypoint padding: the largest workgroup covers 8 wavefronts x one 32-row MFMA tile
constexpr int kTileJ = 64;         // columns of B staged per step: 2 MFMA tiles per wavefront
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// A pointer that arrives inside a struct (GuidedPair, ScreenPair, MatchPair) is a generic pointer to the compiler: its loads
// become flat_load, which count on BOTH wait counters -- an LDS wait (lgkmcnt) then also waits for loads that were meant to
// travel behind the arithmetic.  These say what the host knows: the pointer is device memory (global_load, vmcnt only).
#define PGI_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ f32x4 load_global_f32x4(const float* p, size_t k) { return ((const f32x4 PGI_GLOBAL*)p)[k]; }
__device__ __forceinline__ f32x2 load_global_f32x2(const float* p, size_t k) { return ((const f32x2 PGI_GLOBAL*)p)[k]; }
__device__ __forceinline__ float load_global_f32(const float* p, size_t k) { return ((const float PGI_GLOBAL*)p)[k]; }
__device__ __forceinline__ uint32_t load_global_u32(const uint32_t* p, size_t k) { return ((const uint32_t PGI_GLOBAL*)p)[k]; }

struct RowBest {
    float b1, b2;  // smallest and second-smallest squared distance of the row
    uint32_t j1;   // column of the smallest (ties: lowest column)
    uint32_t pad;
};
struct MatchPair {
    const float *at, *na, *bt, *nb;
    uint32_t n_a, n_a_pad, n_b, n_b_pad;
    uint64_t row_off;  // RowBest index of (split 0, row 0) of this pair
    uint64_t col_off;  // column-best index of column 0 of this pair
};

// C/D layout of the 32x32 MFMA: register r of lane (c = lane & 31, h = lane >> 5) holds C[row][c]
__device__ __forceinline__ uint32_t mfma_row(int r, uint32_t h) { return (uint32_t)((r & 3) + 8 * (r >> 2)) + 4u * h; }

__global__ __launch_bounds__(256) void desc_prepare_kernel(const float* __restrict__ desc, uint32_t n, uint32_t n_pad,
                                                           float* __restrict__ desc_t, float* __restrict__ norm) {
    __shared__ float tile[64 * (kD + 1)];
    const uint32_t j0 = blockIdx.x * 64u, tid = threadIdx.x;
    for (uint32_t idx = tid; idx < 64u * (kD / 4); idx += 256u) {
        const uint32_t jj = idx / (kD / 4), k4 = idx % (kD / 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j0 + jj < n) v = *reinterpret_cast<const float4*>(desc + (size_t)(j0 + jj) * kD + 4 * k4);
        float* dst = tile + jj * (kD + 1) + 4 * k4;
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    }
    __syncthreads();
    for (uint32_t idx = tid; idx < 64u * kD; idx += 256u) {
        const uint32_t k = idx >> 6, jj = idx & 63u;
        desc_t[(size_t)k * n_pad + j0 + jj] = tile[jj * (kD + 1) + k];
    }
    if (tid < 64u) {
        float s = 0.0f;
        for (int k = 0; k < kD; ++k) { const float v = tile[tid * (kD + 1) + k]; s = fmaf(v, v, s); }
        norm[j0 + tid] = s;
    }
}

// One workgroup = NW wavefronts = NW*32 rows of A against one column split of B.  B tiles (64 columns x 128 k) are
// DMA'd global -> LDS (global_load_lds_dwordx4, no staging registers) into a double buffer in MFMA operand order:
// word s*128 + sub*64 + lane holds B[k = 2s + (lane >> 5)][tile*64 + sub*32 + (lane & 31)], so one conflict-free
// ds_read2_b32 feeds both MFMA column tiles of step s.  One barrier per tile.
template <int NW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void desc_top2_kernel(const MatchPair* __restrict__ pairs,
                                                                             RowBest* __restrict__ rowpart,
                                                                             unsigned long long* __restrict__ colbest,
                                                                             uint32_t splits, uint32_t wgs_per_pair) {
    __shared__ float bt[2][kTileJ * kD];  // 2 x 32 KB
    // XCD-aware order: the workgroups an XCD runs back to back are consecutive row blocks of the same pair, so the B
    // panel they all stream stays in that XCD's L2 (bijective remap of the round-robin blockIdx -> XCD assignment)
    const uint32_t nwg = gridDim.x, xcd = blockIdx.x & 7u, qq = nwg >> 3, rr = nwg & 7u;
    const uint32_t wgid = (xcd < rr ? xcd * (qq + 1u) : rr * (qq + 1u) + (xcd - rr) * qq) + (blockIdx.x >> 3);
    const MatchPair P = pairs[wgid / wgs_per_pair];
    const uint32_t x = wgid % wgs_per_pair, rb = x / splits, split = x % splits;
    constexpr uint32_t kRows = NW * 32u;
    if (rb * kRows >= P.n_a_pad) return;
    const uint32_t tiles = P.n_b_pad / kTileJ;
    const uint32_t t0 = (uint32_t)((uint64_t)tiles * split / splits), t1 = (uint32_t)((uint64_t)tiles * (split + 1) / splits);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6, c = lane & 31u, h = lane >> 5;
    const uint32_t row_base = rb * kRows + w * 32u;

    // A fragment of the whole K range: a[s] = A[row_base + c][2s + h]
    float a[kD / 2];
#pragma unroll
    for (int s = 0; s < kD / 2; ++s) a[s] = P.at[(size_t)(2 * s + (int)h) * P.n_a_pad + row_base + c];
    float nar[16], b1[16], b2[16];
    uint32_t j1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        nar[r] = P.na[row_base + mfma_row(r, h)];
        b1[r] = INFINITY; b2[r] = INFINITY; j1[r] = 0u;
    }
    // per-lane source of chunk q (1 KiB per wavefront instruction): LDS words [ch*256 + 4*lane, +4)
    auto stage = [&](uint32_t tile, uint32_t buf) {
#pragma unroll
        for (int q = 0; q < 32 / NW; ++q) {
            const uint32_t ch = (uint32_t)q * NW + w, wd = ch * 256u + 4u * lane;
            const uint32_t s = wd >> 7, sub = (wd >> 6) & 1u, hh = (wd >> 5) & 1u, cc = wd & 31u;
            const float* src = P.bt + (size_t)(2u * s + hh) * P.n_b_pad + tile * kTileJ + sub * 32u + cc;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(&bt[buf][ch * 256u]), 16, 0, 0);
        }
    };
    // Rows past the image and columns past the image carry an infinite norm: their distances come out as +inf by the
    // same two operations as everyone else's and drop out of every minimum without a per-element select.
#pragma unroll
    for (int r = 0; r < 16; ++r)
        if (row_base + mfma_row(r, h) >= P.n_a) nar[r] = INFINITY;
    if (t0 < t1) stage(t0, 0u);
    for (uint32_t tile = t0; tile < t1; ++tile) {
        const uint32_t cur = (tile - t0) & 1u;
        __syncthreads();  // tile landed in bt[cur] (vmcnt(0) precedes the barrier); everyone is done reading bt[cur ^ 1]
        if (tile + 1 < t1) stage(tile + 1, cur ^ 1u);
        // column state of this lane's two columns: norm and the best key so far (stale reads are conservative)
        float nbj[2];
        unsigned long long seen[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const uint32_t j = tile * kTileJ + 32u * sub + c;
            nbj[sub] = j < P.n_b ? P.nb[j] : INFINITY;
            seen[sub] = __builtin_nontemporal_load(&colbest[P.col_off + j]);
        }
        const float* bcur = &bt[cur][lane];
        f32x16 acc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][r] = 0.0f; acc[1][r] = 0.0f; }
        // B operands run kAhead steps ahead of the matrix cores (ring of registers), so LDS latency never stalls an MFMA
        constexpr int kAhead = 6;
        float x0[kAhead], x1[kAhead];
#pragma unroll
        for (int s = 0; s < kAhead; ++s) { x0[s] = bcur[s * 128]; x1[s] = bcur[s * 128 + 64]; }
#pragma unroll
        for (int s = 0; s < kD / 2; ++s) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], x0[s % kAhead], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], x1[s % kAhead], acc[1], 0, 0, 0);
            if (s + kAhead < kD / 2) { x0[s % kAhead] = bcur[(s + kAhead) * 128]; x1[s % kAhead] = bcur[(s + kAhead) * 128 + 64]; }
        }
        // pin the issue order the source states: kAhead reads up front, then {2 MFMA, 1 LDS read} per step
        __builtin_amdgcn_sched_group_barrier(0x100, kAhead, 0);
#pragma unroll
        for (int s = 0; s < kD / 2; ++s) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            if (s + kAhead < kD / 2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        // branch-free epilogue, 10 VALU operations per distance.  On this chip they are NOT hidden behind the other
        // wavefront's MFMAs (f32 MFMA and f32 VALU peak at the same 256 flop/cycle/CU and, measured, add up: DESIGN.md §7),
        // so every operation removed here is kernel time
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const uint32_t j = tile * kTileJ + 32u * sub + c;
            const bool jvalid = j < P.n_b;
            float cbest = INFINITY;
            uint32_t ci = 0u;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float t = nar[r] + nbj[sub];
                float d2 = fmaf(-2.0f, acc[sub][r], t);  // == t - 2*acc: the product is exact
                d2 = d2 > 0.0f ? d2 : 0.0f;
                // columns arrive in ascending order: strict < keeps the lowest column on ties
                const bool lt1 = d2 < b1[r];
                b2[r] = __builtin_amdgcn_fmed3f(b1[r], b2[r], d2);  // second smallest of {b1 <= b2, d2}
                b1[r] = fminf(b1[r], d2);
                j1[r] = lt1 ? j : j1[r];
                const bool ltc = d2 < cbest;  // rows ascend in r: strict < keeps the lowest row on ties
                cbest = ltc ? d2 : cbest;
                ci = ltc ? row_base + mfma_row(r, h) : ci;
            }
            unsigned long long ckey = ((unsigned long long)__float_as_uint(cbest) << 32) | ci;
            const unsigned long long other = __shfl_xor(ckey, 32);
            ckey = other < ckey ? other : ckey;
            if (h == 0u && jvalid && ckey < seen[sub]) atomicMin(&colbest[P.col_off + j], ckey);
        }
    }
    // row top-2 across the 32 lanes that share a row (same h): xor butterfly inside each half-wave
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float x1 = b1[r], x2 = b2[r];
        uint32_t xj = j1[r];
#pragma unroll
        for (int m = 1; m < 32; m <<= 1) {
            const float o1 = __shfl_xor(x1, m), o2 = __shfl_xor(x2, m);
            const uint32_t oj = (uint32_t)__shfl_xor((int)xj, m);
            const bool take = o1 < x1 || (o1 == x1 && oj < xj);
            const float n2 = fminf(fmaxf(x1, o1), fminf(x2, o2));
            if (take) { x1 = o1; xj = oj; }
            x2 = n2;
        }
        if (c == 0u) {
            RowBest rbst;
            rbst.b1 = x1; rbst.b2 = x2; rbst.j1 = xj; rbst.pad = 0u;
            rowpart[P.row_off + (uint64_t)split * P.n_a_pad + row_base + mfma_row(r, h)] = rbst;
        }
    }
}

// one workgroup per pair; keys[] = dynamic LDS, np = power of two >= n_a
__global__ __launch_bounds__(1024) void match_select_kernel(const MatchPair* __restrict__ pairs, RowBest* __restrict__ rowpart,
                                                            const unsigned long long* __restrict__ colbest, uint32_t splits,
                                                            uint32_t max_matches, uint32_t* __restrict__ out_src,
                                                            uint32_t* __restrict__ out_dst, double* __restrict__ out_ratio,
                                                            uint32_t* __restrict__ out_count) {
    extern __shared__ unsigned long long keys[];
    __shared__ uint32_t s_count, s_valid;
    const MatchPair P = pairs[blockIdx.x];
    const uint32_t tid = threadIdx.x;
    uint32_t np = 2;
    while (np < P.n_a) np <<= 1;
    if (tid == 0) { s_count = 0u; s_valid = 0u; }
    __syncthreads();
    const bool both = P.n_a >= 2u && P.n_b >= 2u;  // feature_utils.h:167-168
    // Only the rows that pass the ratio and mutual tests are sorted (about a third on the bench): they are compacted to
    // the front first -- in any order, a key is unique (it carries its row) and the sort defines the output order -- and
    // the bitonic network runs over the next power of two of their number instead of over every row.
    for (uint32_t i = tid; i < np; i += 1024u) {
        unsigned long long key = ~0ull;
        if (i < P.n_a && both) {
            RowBest m = rowpart[P.row_off + i];
            for (uint32_t s = 1; s < splits; ++s) {  // later splits hold higher columns: ties keep the earlier one
                const RowBest o = rowpart[P.row_off + (uint64_t)s * P.n_a_pad + i];
                const bool take = o.b1 < m.b1;
                const float n2 = fminf(fmaxf(m.b1, o.b1), fminf(m.b2, o.b2));
                if (take) { m.b1 = o.b1; m.j1 = o.j1; }
                m.b2 = n2;
            }
            const float dist1 = sqrtf(m.b1), dist2 = sqrtf(m.b2);
            const bool mutual = (uint32_t)(colbest[P.col_off + m.j1] & 0xffffffffull) == i;
            if ((double)dist1 < 0.90 * (double)dist2 && mutual) {
                const float ratio = dist1 / dist2;
                key = ((unsigned long long)__float_as_uint(ratio) << 32) | i;
                rowpart[P.row_off + i].j1 = m.j1;  // merged best column, read back after the sort
            }
        }
        const bool valid = key != ~0ull;
        const unsigned long long bal = __ballot(valid);
        uint32_t base = 0;
        if ((tid & 63u) == 0u && bal) base = atomicAdd(&s_valid, (uint32_t)__popcll(bal));
        base = (uint32_t)__shfl((int)base, 0);
        if (valid) keys[base + (uint32_t)__popcll(bal & ((1ull << (tid & 63u)) - 1ull))] = key;
    }
    __syncthreads();
    const uint32_t n_valid = s_valid;
    np = 2;
    while (np < n_valid) np <<= 1;
    for (uint32_t i = n_valid + tid; i < np; i += 1024u) keys[i] = ~0ull;
    __syncthreads();
    for (uint32_t k = 2; k <= np; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = tid; i < np; i += 1024u) {
                const uint32_t l = i ^ j;
                if (l > i) {
                    const unsigned long long x = keys[i], y = keys[l];
                    const bool up = (i & k) == 0u;
                    if ((x > y) == up) { keys[i] = y; keys[l] = x; }
                }
            }
            __syncthreads();
        }
    }
    for (uint32_t i = tid; i < np; i += 1024u) {
        if (keys[i] != ~0ull && (i + 1 == np || keys[i + 1] == ~0ull)) s_count = i + 1;
    }
    __syncthreads();
    const uint32_t count = s_count < max_matches ? s_count : max_matches;
    for (uint32_t i = tid; i < count; i += 1024u) {
        const unsigned long long key = keys[i];
        const uint32_t src = (uint32_t)(key & 0xffffffffull);
        const size_t o = (size_t)blockIdx.x * max_matches + i;
        out_src[o] = src;
        out_dst[o] = rowpart[P.row_off + src].j1;
        out_ratio[o] = (double)__uint_as_float((uint32_t)(key >> 32));
    }
    if (tid == 0) out_count[blockIdx.x] = count;
}
// ---------------------------------------------------------------------------------------------------------------
// Screened matching: the same exact result, most of the arithmetic at the f16 matrix-core rate (16x the f32 one).
//
//   desc_screen_kernel   s~_ij from v_mfma_f32_32x32x16_f16 on f16-rounded descriptors.  |d~2 - d2| <= eps_i for
//                        every column (eps_i from the rounding analysis below), so the columns that can be among a
//                        row's two nearest are those with d~2 <= T + 2 eps_i, T = the second smallest d~2 of the row.
//                        Every lane keeps its two best columns and the VALUE of its third best: if no lane's third
//                        best is inside the window, the stored columns provably contain the exact top-2 (certificate);
//                        otherwise the row is flagged.  The bookkeeping is three VALU operations per distance: the
//                        column's position rides in the low mantissa bits of the key, so min / med3 carry it along.
//   desc_verify_kernel   exact f32 chain (the specification) for the stored candidates only -> RowBest; flagged rows
//                        go to a list.
//   desc_exact_rows_kernel  flagged rows: full exact scan, four rows per 1024-thread workgroup.
//   needed_columns_kernel   the columns the mutual test will read: the column-wise direction only visits those.
// The column-wise nearest neighbour is the row problem with the images swapped (identical bits: products and the
// norm sum commute), so the pipeline runs twice and feeds the unchanged selection kernel.
//
// eps: f16 rounding has unit roundoff u = 2^-11 (absolute 2^-25 in the subnormal range), so |sum a~b~ - sum ab| <=
// (2u + u^2) sum|a_k b_k| + 2^-25 sqrt(128) (|a| + |b|) <= (2u + u^2) sqrt(na nb) + 3.4e-7 (|a| + |b|); both f32
// accumulations add at most 128 * 2^-24 sqrt(na nb) each and the norms carry 2^-17 relative error: kEpsS = 2^-10 + 2^-15
// covers the relative part with margin; d2 = na + nb - 2s doubles everything and adds a few ulps of na + nb.  f16
// overflows above 65504: |x_k| <= sqrt(norm), so rows or images with a squared norm above 3e9 are simply flagged.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// calls f(integral_constant<int, U>) for U = 0 .. N-1 (a compile-time loop: the slot index stays a constant)
template <class F, int... U>
__device__ __forceinline__ void for_each_slot(F&& f, std::integer_sequence<int, U...>) {
    (f(std::integral_constant<int, U>{}), ...);
}
// Layout of the f16 copy of an image: tiles of 64 keypoints; inside a tile, chunk cg = k / 8 of keypoint jj sits at element
// (cg * 64 + jj) * 8 -- exactly the order the screen kernel's LDS buffer (and the MFMA B operand fetch) wants, so a tile
// is one contiguous 16 KB block.
__host__ __device__ __forceinline__ size_t f16_offset(uint32_t keypoint, uint32_t k) {
    return (size_t)(keypoint >> 6) * (64u * 128u) + ((size_t)(k >> 3) * 64u + (keypoint & 63u)) * 8u + (k & 7u);
}
constexpr float kEpsS = 0.0009765625f + 0.000030517578125f;  // 2^-10 + 2^-15
constexpr int kCand = 8;  // stored candidates per row and column split

struct ScreenPair {
    const unsigned short *abf, *bbf;  // f16 copies in MFMA operand order per 64-keypoint tile (desc_round_f16_kernel)
    const float *arm, *brm;           // row-major f32,  n_pad x 128
    const float *at, *bt;             // transposed f32, 128 x n_pad (exact fallback)
    const float *na, *nb;
    uint32_t n_a, n_a_pad, n_b, n_b_pad;
    uint64_t row_off;                 // first row slot of this pair (ScreenRow / RowBest / keys)
    // Restricted pass (the column-wise direction): only the rows listed in row_list[row_off ..] (ascending, *row_count of
    // them) are screened and verified; ScreenRow slots are then positions in the list.  nullptr: every row.
    const uint32_t* row_list;
    const uint32_t* row_count;
};
struct ScreenRow {
    uint32_t count, flagged;
    uint32_t j[kCand];
};

// TOPK = 2: a row's two nearest columns are wanted; TOPK = 1: only the nearest (the column-wise pass of the swapped
// problem).  Either way each lane keeps its two best columns and the value of its third (packed keys, see below): the
// window is T + 2 eps around the row's TOPK-th smallest approximate distance T.
template <int NW, int TOPK>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void desc_screen_kernel(const ScreenPair* __restrict__ pairs, ScreenRow* __restrict__ out,
                                                                uint32_t splits, uint32_t wgs_per_pair, uint64_t split_stride) {
    __shared__ __attribute__((aligned(16))) unsigned short bt[2][kTileJ * kD];  // 2 x 16 KB, MFMA operand order
    __shared__ float s_nbmax[NW];
    const uint32_t nwg = gridDim.x, xcd = blockIdx.x & 7u, qq = nwg >> 3, rr = nwg & 7u;
    const uint32_t wgid = (xcd < rr ? xcd * (qq + 1u) : rr * (qq + 1u) + (xcd - rr) * qq) + (blockIdx.x >> 3);
    const ScreenPair P = pairs[wgid / wgs_per_pair];
    const uint32_t x = wgid % wgs_per_pair, rb = x / splits, split = x % splits;
    constexpr uint32_t kRows = NW * 32u;
    if (rb * kRows >= P.n_a_pad) return;
    const uint32_t n_listed = P.row_list ? load_global_u32(P.row_count, 0) : 0u;
    if (P.row_list && rb * kRows >= n_listed) return;  // (workgroup-uniform)
    const uint32_t tiles = P.n_b_pad / kTileJ;
    const uint32_t t0 = (uint32_t)((uint64_t)tiles * split / splits), t1 = (uint32_t)((uint64_t)tiles * (split + 1) / splits);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6, c = lane & 31u, h = lane >> 5;
    const uint32_t row_base = rb * kRows + w * 32u;
    // slot -> row of A (slots past the end of a list repeat row 0: computed, never read)
    auto row_of = [&](uint32_t slot) { return P.row_list ? (slot < n_listed ? load_global_u32(P.row_list, P.row_off + slot) : 0u) : slot; };

    // largest column norm (for eps): every wavefront scans the norm array once
    float nbmax = 0.0f;
    for (uint32_t j = tid; j < P.n_b_pad; j += NW * 64u) nbmax = fmaxf(nbmax, load_global_f32(P.nb, j));
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) nbmax = fmaxf(nbmax, __shfl_xor(nbmax, m));
    if (lane == 0) s_nbmax[w] = nbmax;
    // A fragments: lane (c, h) holds A[row_base + c][16 ks + 8 h .. +7] for the eight k-steps
    f16x8 a[kD / 16];
    const uint32_t a_row = row_of(row_base + c);
#pragma unroll
    for (int ks = 0; ks < kD / 16; ++ks)
        a[ks] = *(const f16x8 PGI_GLOBAL*)(P.abf + f16_offset(a_row, 16u * (uint32_t)ks + 8u * h));
    // Per lane and row: the three smallest KEYS seen so far, b1 <= b2 <= b3, as unsigned integers.  A key is the bit
    // pattern of the approximate squared distance d~ = max(0, na + nb - 2 s~) -- non-negative floats order like their bit
    // patterns -- with the column's position inside this lane's stream (code = 2 * (tile - t0) + sub, 9 bits: at most
    // 256 tiles) in the low mantissa bits.  The column therefore travels with the value through plain integer min / median
    // instructions: v_med3_u32, v_med3_u32, v_min_u32 keep a top-3 where separate value and index registers needed eleven
    // compares and selects (and float min / med3 would pay a canonicalising v_max each).  The row norm enters through
    // the accumulator, which starts at -na/2 instead of 0.  The 2^-14 relative perturbation of the key is part of eps
    // below; padded columns carry a huge finite norm (an infinity with mantissa bits set would be a NaN).
    constexpr float kFar = 3.0e38f;
    constexpr uint32_t kKeyInf = 0x7F800000u;  // above every finite key
    float nar[16];
    uint32_t b1[16], b2[16], b3[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        nar[r] = load_global_f32(P.na, row_of(row_base + mfma_row(r, h)));
        b1[r] = kKeyInf; b2[r] = kKeyInf; b3[r] = kKeyInf;
    }
    auto med3u = [](uint32_t x, uint32_t y, uint32_t z) {
        uint32_t o;
        asm("v_med3_u32 %0, %1, %2, %3" : "=v"(o) : "v"(x), "v"(y), "v"(z));
        return o;
    };
    // B tiles (64 columns x 128 k, 16 KB of f16) travel global -> registers -> LDS.  A tile's MFMA work is only ~500 cycles
    // per wavefront, far less than a global load takes to come back, so a one-tile-ahead LDS-DMA double buffer left the
    // kernel waiting on memory every tile; here three tiles are in flight in registers (their loads were issued two
    // iterations before they are written to LDS), and the LDS double buffer only decouples the writer from the readers.
    // Chunk cg = (ks, h') of 64 columns x 16 bytes: LDS element ((cg * 64 + jj) * 8), global B[j0 + jj][16 ks + 8 h' ..].
    constexpr int kChunks = 16 / NW, kPre = 48 / (4 * kChunks);  // 48 VGPRs of tiles in flight: 3 tiles at NW = 4, 6 at NW = 8
    f16x8 pre[kPre][kChunks];
    auto gload = [&](uint32_t tile, f16x8 (&dst)[kChunks]) {
#pragma unroll
        for (int q = 0; q < kChunks; ++q) {
            const uint32_t cg = (uint32_t)q * NW + w;
            // operand order in HBM: a wavefront reads 1 KB contiguous (the row-major layout made every lane touch its own
            // 256-byte row, and the vector L1 then spent 64 tag cycles per load instruction)
            dst[q] = *(const f16x8 PGI_GLOBAL*)(P.bbf + (size_t)tile * (kTileJ * kD) + (cg * 64u + lane) * 8u);
        }
    };
    auto lwrite = [&](const f16x8 (&src)[kChunks], uint32_t buf) {
#pragma unroll
        for (int q = 0; q < kChunks; ++q) {
            const uint32_t cg = (uint32_t)q * NW + w;
            *reinterpret_cast<f16x8*>(&bt[buf][cg * 512u + lane * 8u]) = src[q];
        }
    };
#pragma unroll
    for (int u = 0; u < kPre; ++u)
        if (t0 + (uint32_t)u < t1) gload(t0 + (uint32_t)u, pre[u]);
    if (t0 < t1) lwrite(pre[0], 0u);
    // one tile of work; U = the tile's register slot (compile-time, so the ring never turns into selects)
    auto step = [&](uint32_t tile, auto slot) {
        constexpr int u = decltype(slot)::value;
        const uint32_t cur = (tile - t0) & 1u;
        __syncthreads();  // tile is in bt[cur] (written last iteration); everyone is done reading bt[cur ^ 1]
        if (tile + 1 < t1) lwrite(pre[(u + 1) % kPre], cur ^ 1u);      // loaded two iterations ago
        if (tile + (uint32_t)kPre < t1) gload(tile + (uint32_t)kPre, pre[u]);  // this tile's registers are free again
        float nbj[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const uint32_t j = tile * kTileJ + 32u * sub + c;
            nbj[sub] = j < P.n_b ? load_global_f32(P.nb, j) : kFar;  // padded columns: never near any window
        }
        f32x16 acc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][r] = -0.5f * nar[r]; acc[1][r] = acc[0][r]; }  // so that nb - 2 acc = na + nb - 2 s
#pragma unroll
        for (int ks = 0; ks < kD / 16; ++ks) {
            const f16x8 x0 = *reinterpret_cast<const f16x8*>(&bt[cur][((2 * ks + (int)h) * 64 + (int)c) * 8]);
            const f16x8 x1 = *reinterpret_cast<const f16x8*>(&bt[cur][((2 * ks + (int)h) * 64 + 32 + (int)c) * 8]);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks], x0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks], x1, acc[1], 0, 0, 0);
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const uint32_t code = ((tile - t0) << 1) | (uint32_t)sub;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float d = fmaxf(fmaf(-2.0f, acc[sub][r], nbj[sub]), 0.0f);
                const uint32_t k = (__float_as_uint(d) & ~0x1FFu) | (code & 0x1FFu);  // v_bfi_b32
                b3[r] = med3u(b2[r], b3[r], k);  // (old b2): third smallest of {b1, b2, b3, k}
                b2[r] = med3u(b1[r], b2[r], k);
                b1[r] = min(b1[r], k);
            }
        }
    };
    for (uint32_t base = t0; base < t1; base += (uint32_t)kPre)  // workgroup-uniform trip counts and guards
        for_each_slot([&](auto slot) {
            constexpr uint32_t u = (uint32_t)decltype(slot)::value;
            if (base + u < t1) step(base + u, slot);
        }, std::make_integer_sequence<int, kPre>{});
    __syncthreads();
    float nbm = 0.0f;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) nbm = fmaxf(nbm, s_nbmax[ww]);
    // per row: T = the TOPK-th smallest approximate distance over the 32 lanes of the row, window T + 2 eps
    auto column_of = [&](uint32_t key) {  // undo the code: the lane's own column stream is (tile, sub) -> tile*64 + 32*sub + c
        const uint32_t code = key & 0x1FFu;
        return (t0 + (code >> 1)) * (uint32_t)kTileJ + 32u * (code & 1u) + c;
    };
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        uint32_t x1 = b1[r], x2 = b2[r];
#pragma unroll
        for (int m = 1; m < 32; m <<= 1) {
            const uint32_t o1 = (uint32_t)__shfl_xor((int)x1, m), o2 = (uint32_t)__shfl_xor((int)x2, m);
            const uint32_t n2 = min(max(x1, o1), min(x2, o2));
            x1 = min(x1, o1);
            x2 = n2;
        }
        // |d~2 - d2| of the f16 products and f32 accumulations (analysis above; the accumulator starts at -na/2) + the
        // 2^-14 relative perturbation of a key by its code bits, d~ <= na + nb + 2 sqrt(na nb) <= 2 (na + nb)
        const float eps = 2.0f * (kEpsS * sqrtf(nar[r] * nbm) + 3.4e-7f * (sqrtf(nar[r]) + sqrtf(nbm))) + 1.0e-6f * (nar[r] + nbm) +
                          1.25e-4f * (nar[r] + nbm);
        const uint32_t tk = TOPK == 2 ? x2 : x1;
        // window bound as a key: every stored key <= it is a candidate (keys of valid columns are finite floats >= 0)
        const float winf = __uint_as_float(tk & 0xFFFFFE00u) + 2.0f * eps;
        const uint32_t win = tk >= kKeyInf ? 0x7F7FFFFFu : __float_as_uint(fminf(winf, 3.0e38f)) | 0x1FFu;
        const bool safe = nar[r] <= 3.0e9f && nbm <= 3.0e9f;  // f16 range; also false for NaN norms
        const uint32_t kValid = 0x7E967699u;  // bits of 1.0e38f: padded columns (norm 3e38) lie above
        const bool k1 = b1[r] <= win && b1[r] < kValid;
        const bool k2 = b2[r] <= win && b2[r] < kValid;
        const bool miss = (b3[r] <= win && b3[r] < kValid) || !safe;  // b3 = best column of this lane that is NOT stored
        const unsigned long long m1 = __ballot(k1), m2 = __ballot(k2), mm = __ballot(miss);
        const uint32_t h1 = (uint32_t)(m1 >> (32u * h)), h2 = (uint32_t)(m2 >> (32u * h)), hm = (uint32_t)(mm >> (32u * h));
        const uint32_t n1 = __popc(h1), total = n1 + __popc(h2);
        const uint32_t row = row_base + mfma_row(r, h);
        ScreenRow* o = out + (size_t)split * split_stride + P.row_off + row;
        const uint32_t below = (1u << c) - 1u;
        if (total <= (uint32_t)kCand) {
            if (k1) o->j[__popc(h1 & below)] = column_of(b1[r]);
            if (k2) o->j[n1 + __popc(h2 & below)] = column_of(b2[r]);
        }
        if (c == 0u) {
            o->count = total <= (uint32_t)kCand ? total : 0u;
            o->flagged = (hm != 0u || total > (uint32_t)kCand) ? 1u : 0u;
        }
    }
}

// exact f32 chain d2 of row i of A against row j of B (both row-major): the specification's arithmetic
__device__ __forceinline__ float exact_d2(const float* __restrict__ ai, const float* __restrict__ bj, float na, float nb) {
    float s = 0.0f;
#pragma unroll 4
    for (int k = 0; k < kD; k += 4) {
        const float4 x = *reinterpret_cast<const float4*>(ai + k), y = *reinterpret_cast<const float4*>(bj + k);
        s = fmaf(x.x, y.x, s);
        s = fmaf(x.y, y.y, s);
        s = fmaf(x.z, y.z, s);
        s = fmaf(x.w, y.w, s);
    }
    const float d2 = (na + nb) - 2.0f * s;
    return d2 > 0.0f ? d2 : 0.0f;
}

// the same chain with A read from its transposed copy (128 x n_pad): neighbouring lanes hold neighbouring rows, so the A side
// of a wavefront's load is one or two cache lines instead of 64 (the row-major read cost the vector L1 a tag per lane)
__device__ __forceinline__ float exact_d2_ta(const float* __restrict__ at_i, uint32_t n_a_pad, const float* __restrict__ bj, float na, float nb) {
    float s = 0.0f;
#pragma unroll 4
    for (int k = 0; k < kD; k += 4) {
        const float4 y = *reinterpret_cast<const float4*>(bj + k);
        s = fmaf(at_i[(size_t)k * n_a_pad], y.x, s);
        s = fmaf(at_i[(size_t)(k + 1) * n_a_pad], y.y, s);
        s = fmaf(at_i[(size_t)(k + 2) * n_a_pad], y.z, s);
        s = fmaf(at_i[(size_t)(k + 3) * n_a_pad], y.w, s);
    }
    const float d2 = (na + nb) - 2.0f * s;
    return d2 > 0.0f ? d2 : 0.0f;
}

// one thread per row: exact distances of the screened candidates -> (b1, j1, b2); flagged rows are queued.
// as_keys = 1 writes the column-best key (d2 bits, row index) of the swapped problem instead of a RowBest.
__global__ __launch_bounds__(256) void desc_verify_kernel(const ScreenPair* __restrict__ pairs, const ScreenRow* __restrict__ scr,
                                                         uint32_t splits, uint64_t split_stride, uint32_t max_rows, RowBest* __restrict__ rowbest,
                                                         unsigned long long* __restrict__ keys, uint32_t as_keys,
                                                         uint32_t* __restrict__ fb_list, uint32_t* __restrict__ fb_count) {
    const ScreenPair P = pairs[blockIdx.y];
    const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    if (slot >= P.n_a || slot >= max_rows) return;
    if (P.row_list && slot >= *P.row_count) return;
    const uint32_t i = P.row_list ? P.row_list[P.row_off + slot] : slot;
    bool flagged = false;
    float b1 = INFINITY, b2 = INFINITY;
    uint32_t j1 = 0u, seen = 0u;
    const float* at_i = P.at + i;
    const float nai = P.na[i];
    for (uint32_t s = 0; s < splits; ++s) {
        const ScreenRow* r = scr + (size_t)s * split_stride + P.row_off + slot;
        flagged |= r->flagged != 0u;
        const uint32_t cnt = r->count;
        for (uint32_t q = 0; q < cnt; ++q) {
            const uint32_t j = r->j[q];
            const float d2 = exact_d2_ta(at_i, P.n_a_pad, P.brm + (size_t)j * kD, nai, P.nb[j]);
            if (d2 < b1 || (d2 == b1 && j < j1)) { b2 = b1; b1 = d2; j1 = j; }
            else if (d2 < b2) { b2 = d2; }
            ++seen;
        }
    }
    if (as_keys ? (P.n_b >= 1u && seen < 1u) : (P.n_b >= 2u && seen < 2u)) flagged = true;
    if (flagged) {  // queue the row in its pair's list (capacity n_a_pad)
        const uint32_t slot = atomicAdd(fb_count + blockIdx.y, 1u);
        fb_list[P.row_off + slot] = i;
        return;
    }
    if (as_keys) keys[P.row_off + i] = ((unsigned long long)__float_as_uint(b1) << 32) | j1;
    else { RowBest rb; rb.b1 = b1; rb.b2 = b2; rb.j1 = j1; rb.pad = 0u; rowbest[P.row_off + i] = rb; }
}

// flagged rows: the full exact scan.  A workgroup takes four queued rows of one pair at a time; its 1024 threads own the
// columns j = tid, tid + 1024, ..., two of them per step, so every B value loaded feeds four chains and B is streamed once
// per four rows.  Few rows are flagged (a handful per pair), so the kernel is a latency chain, not a throughput problem:
// sixteen wavefronts with two independent column streams each keep 4 x 8 times more loads in flight than the first version
// (256 threads, one column at a time: 265 us for 396 rows of 56 pairs).
constexpr uint32_t kExactThreads = 1024u, kExactWaves = kExactThreads / 64u;
__global__ __launch_bounds__(1024) void desc_exact_rows_kernel(const ScreenPair* __restrict__ fwd, const ScreenPair* __restrict__ bwd, uint32_t n_pairs,
                                                              const uint32_t* __restrict__ fb_fwd, const uint32_t* __restrict__ fb_bwd,
                                                              const uint32_t* __restrict__ fb_count, RowBest* __restrict__ rowbest,
                                                              unsigned long long* __restrict__ keys, uint32_t y_base) {
    __shared__ float arow[4][kD];
    __shared__ float mb1[4][kExactWaves], mb2[4][kExactWaves];
    __shared__ uint32_t mj1[4][kExactWaves];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    const uint32_t y = blockIdx.y + y_base;      // [0, n_pairs): forward problem, [n_pairs, 2 n_pairs): the swapped one
    const uint32_t as_keys = y >= n_pairs ? 1u : 0u;  // the swapped problem writes column-best keys
    const ScreenPair P = as_keys ? bwd[y - n_pairs] : fwd[y];
    const uint32_t* fb_list = as_keys ? fb_bwd : fb_fwd;
    const uint32_t total = fb_count[y];  // forward counters, then backward
    // Only the first one or two groups of a pair have rows.  Workgroups go to the eight XCDs round robin by linear id, so
    // with group = blockIdx.x every pair's work landed on XCDs 0 and 1 (105 working groups on 64 CUs: 330 us); rotating the
    // assignment by the pair index spreads them over the chip.
    const uint32_t gx = (blockIdx.x + gridDim.x - y % gridDim.x) % gridDim.x;
    for (uint32_t e = gx * 4u; e < total; e += gridDim.x * 4u) {
        uint32_t row[4];
        float nai[4], b1[4], b2[4];
        uint32_t j1[4];
        __syncthreads();  // the previous group's readers of arow / mb* are done
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            row[q] = fb_list[P.row_off + (e + q < total ? e + q : e)];  // a short tail repeats the first row
            nai[q] = P.na[row[q]];
            b1[q] = INFINITY; b2[q] = INFINITY; j1[q] = 0u;
            if (tid < (uint32_t)kD) arow[q][tid] = P.arm[(size_t)row[q] * kD + tid];
        }
        __syncthreads();
        for (uint32_t j = tid; j < P.n_b; j += 2u * kExactThreads) {
            const uint32_t jb = j + kExactThreads;
            const bool two = jb < P.n_b;
            const uint32_t jbc = two ? jb : j;  // a lone last column is computed twice and used once
            float s[4] = {0.0f, 0.0f, 0.0f, 0.0f}, t[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 8
            for (int k = 0; k < kD; ++k) {
                const float bv = P.bt[(size_t)k * P.n_b_pad + j], bw = P.bt[(size_t)k * P.n_b_pad + jbc];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    s[q] = fmaf(arow[q][k], bv, s[q]);
                    t[q] = fmaf(arow[q][k], bw, t[q]);
                }
            }
            const float nbj = P.nb[j], nbk = P.nb[jbc];
#pragma unroll
            for (int q = 0; q < 4; ++q) {  // ascending columns: the first minimum keeps the lowest index
                float d2 = (nai[q] + nbj) - 2.0f * s[q];
                d2 = d2 > 0.0f ? d2 : 0.0f;
                if (d2 < b1[q]) { b2[q] = b1[q]; b1[q] = d2; j1[q] = j; }
                else if (d2 < b2[q]) { b2[q] = d2; }
                float e2 = (nai[q] + nbk) - 2.0f * t[q];
                e2 = e2 > 0.0f ? e2 : 0.0f;
                if (two) {
                    if (e2 < b1[q]) { b2[q] = b1[q]; b1[q] = e2; j1[q] = jb; }
                    else if (e2 < b2[q]) { b2[q] = e2; }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float x1 = b1[q], x2 = b2[q];
            uint32_t xj = j1[q];
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) {
                const float o1 = __shfl_xor(x1, m), o2 = __shfl_xor(x2, m);
                const uint32_t oj = (uint32_t)__shfl_xor((int)xj, m);
                const bool take = o1 < x1 || (o1 == x1 && oj < xj);
                const float n2 = fminf(fmaxf(x1, o1), fminf(x2, o2));
                if (take) { x1 = o1; xj = oj; }
                x2 = n2;
            }
            if (lane == 0) { mb1[q][w] = x1; mb2[q][w] = x2; mj1[q][w] = xj; }
        }
        __syncthreads();
        if (tid < 4u && e + tid < total) {
            float x1 = mb1[tid][0], x2 = mb2[tid][0];
            uint32_t xj = mj1[tid][0];
            for (uint32_t ww = 1; ww < kExactWaves; ++ww) {
                const float o1 = mb1[tid][ww], o2 = mb2[tid][ww];
                const uint32_t oj = mj1[tid][ww];
                const bool take = o1 < x1 || (o1 == x1 && oj < xj);
                const float n2 = fminf(fmaxf(x1, o1), fminf(x2, o2));
                if (take) { x1 = o1; xj = oj; }
                x2 = n2;
            }
            const uint32_t r = fb_list[P.row_off + e + tid];
            if (as_keys) keys[P.row_off + r] = ((unsigned long long)__float_as_uint(x1) << 32) | xj;
            else { RowBest rb; rb.b1 = x1; rb.b2 = x2; rb.j1 = xj; rb.pad = 0u; rowbest[P.row_off + r] = rb; }
        }
    }
}

// Which columns does the mutual test look at?  Only the best column j1 of a row that passes the ratio test
// (match_select_kernel reads colbest[j1] for nothing else), about a third of the columns on the bench and fewer on real
// images -- so the column-wise direction screens and verifies only those.  One workgroup per pair: flags in LDS, then an
// ordered compaction into row_list (ascending) and its length.
__global__ __launch_bounds__(1024) void needed_columns_kernel(const MatchPair* __restrict__ pairs, const RowBest* __restrict__ rowbest,
                                                              uint32_t* __restrict__ list, uint32_t* __restrict__ count) {
    extern __shared__ unsigned char need[];  // n_b_pad flags
    __shared__ uint32_t wsum[16];
    const MatchPair P = pairs[blockIdx.x];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    for (uint32_t j = tid; j < P.n_b_pad; j += 1024u) need[j] = 0;
    __syncthreads();
    if (P.n_a >= 2u && P.n_b >= 2u)
        for (uint32_t i = tid; i < P.n_a; i += 1024u) {
            const RowBest m = rowbest[P.row_off + i];
            if ((double)sqrtf(m.b1) < 0.90 * (double)sqrtf(m.b2)) need[m.j1] = 1;  // the test of match_select_kernel
        }
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t j0 = 0; j0 < P.n_b_pad; j0 += 1024u) {  // n_b_pad is a multiple of 64: whole wavefronts
        const uint32_t j = j0 + tid;
        const bool on = j < P.n_b && need[j] != 0;
        const unsigned long long bal = __ballot(on);
        if (lane == 0) wsum[w] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t before = 0, all = 0;
#pragma unroll
        for (uint32_t ww = 0; ww < 16u; ++ww) {
            const uint32_t v = wsum[ww];
            before += ww < w ? v : 0u;
            all += v;
        }
        if (on) list[P.col_off + base + before + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = j;
        base += all;
        __syncthreads();
    }
    if (tid == 0) count[blockIdx.x] = base;
}

__global__ __launch_bounds__(256) void desc_round_f16_kernel(const float* __restrict__ desc, uint32_t n, uint32_t n_pad,
                                                             float* __restrict__ rm, unsigned short* __restrict__ bf) {
    const size_t idx = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (idx >= (size_t)n_pad * kD) return;
    const float v = idx < (size_t)n * kD ? desc[idx] : 0.0f;
    rm[idx] = v;
    const _Float16 hv = (_Float16)v;  // v_cvt_f16_f32: round to nearest even, gradual underflow
    bf[f16_offset((uint32_t)(idx / kD), (uint32_t)(idx % kD))] = __builtin_bit_cast(unsigned short, hv);
}

// ---------------------------------------------------------------------------------------------------------------
// guided matching with a known pose (matcher.h:199-405)
struct GuidedPair {
    const float *kp1, *kp2, *d1, *d2;
    uint32_t n1, n2;
    double F[9];
    uint64_t off;  // first per-source scratch slot of this pair
    // epipolar hashing (matcher.h:218-331): bins == 0 -> every destination keypoint is a candidate
    double ep0, ep1, min_angle, range;
    int32_t bins, pad;
    uint64_t off2;  // first per-destination scratch slot of this pair (bucketed path)
};
constexpr int kGmMaxBuckets = 64;  // bins the bucketed scan handles (the reference uses 45); more -> the tiled scan below
constexpr double kRadianToDegree = 180.0 / 3.14159265358979323846;
// matcher.h:292-301 / :318-324: angle of a line normal -> bin index
__device__ __forceinline__ double epipolar_angle(double ny, double nx) {  // the folded angle, (0, 180]
    double angle = kRadianToDegree * atan2(ny, nx) + 180.0;
    if (angle > 180) angle -= 180;
    return angle;
}
__device__ __forceinline__ int32_t epipolar_bin_of(double angle, double min_angle, double range, int32_t bins) {
    angle = (double)(bins - 1) * (angle - min_angle) / range;
    const double r = round(angle);
    int32_t b = (r >= 2147483647.0) ? 2147483647 : (r > -2147483648.0 ? (int32_t)r : (int32_t)(-2147483647 - 1));  // NaN -> INT_MIN like x86
    return b < 0 ? 0 : (b > bins - 1 ? bins - 1 : b);
}
__device__ __forceinline__ int32_t epipolar_bin(double ny, double nx, double min_angle, double range, int32_t bins) {
    return epipolar_bin_of(epipolar_angle(ny, nx), min_angle, range, bins);
}
constexpr int kGmTile = 512;  // destination keypoints whose epipolar-line records are staged per step
constexpr int kGmList = 32;   // candidates a thread collects before the wavefront evaluates their descriptors

// One thread per source keypoint.  The per-destination part of the symmetric epipolar distance (F^T x2 and its
// squared norm) is computed once per tile into LDS and read as a broadcast; the gate is a handful of f64 operations
// per (i, j).  The rare survivors (~0.1 %) are queued per thread and their 128-d SSDs evaluated together, so the
// wavefront does not run a 128-step loop for one lane at a time.  Order and arithmetic of matcher.h:333-401 are
// kept: candidates in ascending j, f32 differences squared and summed in double, "second" = the best before the
// last improvement.
__global__ __launch_bounds__(256) void guided_scan_kernel(const GuidedPair* __restrict__ pairs, int32_t* __restrict__ best_out,
                                                          double* __restrict__ ratio_out) {
    __shared__ double rec[kGmTile][4];
    __shared__ int32_t rbin[kGmTile];
    __shared__ uint32_t list[kGmList][256];
    const GuidedPair P = pairs[blockIdx.y];
    if (blockIdx.x * 256u >= P.n1) return;
    const uint32_t tid = threadIdx.x, i = blockIdx.x * 256u + tid;
    const bool active = i < P.n1;
    const double e11 = P.F[0], e12 = P.F[1], e13 = P.F[2], e21 = P.F[3], e22 = P.F[4], e23 = P.F[5], e31 = P.F[6], e32 = P.F[7],
                 e33 = P.F[8];
    double x1 = 0.0, y1 = 0.0;
    if (active) { x1 = (double)P.kp1[2 * (size_t)i]; y1 = (double)P.kp1[2 * (size_t)i + 1]; }
    const double rx = (e11 * x1 + e12 * y1) + e13;
    const double ry = (e21 * x1 + e22 * y1) + e23;
    const double b1 = rx * rx + ry * ry;
    int32_t my_bin = 0;
    if (P.bins) {  // the normal of the line through the epipole and this keypoint (matcher.h:311-324)
        const double vx = x1 - P.ep0, vy = y1 - P.ep1;
        my_bin = epipolar_bin(vx, -vy, P.min_angle, P.range, P.bins);
    }
    double best = DBL_MAX, second = DBL_MAX;
    int32_t best_index = -1;
    uint32_t count = 0, nlist = 0;
    const float4* arow = reinterpret_cast<const float4*>(P.d1 + (size_t)(active ? i : 0u) * kD);
    auto flush = [&]() {
        uint32_t longest = nlist;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)longest, m); longest = o > longest ? o : longest; }
        for (uint32_t c = 0; c < longest; ++c) {
            if (c < nlist) {
                const uint32_t j = list[c][tid];
                const float4* brow = reinterpret_cast<const float4*>(P.d2 + (size_t)j * kD);
                double dd = 0.0;
                for (int k = 0; k < kD / 4; ++k) {
                    const float4 a = arow[k], b = brow[k];
                    const double v0 = (double)(a.x - b.x), v1 = (double)(a.y - b.y), v2 = (double)(a.z - b.z), v3 = (double)(a.w - b.w);
                    dd = dd + v0 * v0;
                    dd = dd + v1 * v1;
                    dd = dd + v2 * v2;
                    dd = dd + v3 * v3;
                }
                ++count;
                if (dd < best) { second = best; best = dd; best_index = (int32_t)j; }
            }
        }
        nlist = 0;
    };
    for (uint32_t j0 = 0; j0 < P.n2; j0 += (uint32_t)kGmTile) {
        __syncthreads();
        const uint32_t lim = P.n2 - j0 < (uint32_t)kGmTile ? P.n2 - j0 : (uint32_t)kGmTile;
        for (uint32_t jj = tid; jj < lim; jj += 256u) {
            const double x2 = (double)P.kp2[2 * (size_t)(j0 + jj)], y2 = (double)P.kp2[2 * (size_t)(j0 + jj) + 1];
            const double rxc = (e11 * x2 + e21 * y2) + e31;
            const double ryc = (e12 * x2 + e22 * y2) + e32;
            const double rwc = (e13 * x2 + e23 * y2) + e33;
            rec[jj][0] = rxc; rec[jj][1] = ryc; rec[jj][2] = rwc; rec[jj][3] = rxc * rxc + ryc * ryc;
            // the destination keypoint's bin: normal (nx, ny) = (rxc, ryc) of its epipolar line F^T x2 (matcher.h:283-301)
            rbin[jj] = P.bins ? epipolar_bin(ryc, rxc, P.min_angle, P.range, P.bins) : 0;
        }
        __syncthreads();
        for (uint32_t jj = 0; jj < lim; ++jj) {
            const double rxc = rec[jj][0], ryc = rec[jj][1], rwc = rec[jj][2], a1 = rec[jj][3];
            const double r = (x1 * rxc + y1 * ryc) + rwc;
            const double num = (r * r) * (a1 + b1), den = a1 * b1;
            // cheap sufficient test for "dist >= 0.75^2" (no division); everything else takes the exact path
            const bool surely_far = den > 0.0 && num >= 0.57 * den;
            if (active && !surely_far && rbin[jj] == my_bin) {
                const double dist = num / den;
                if (!(dist >= 0.75 * 0.75)) { list[nlist][tid] = j0 + jj; ++nlist; }
            }
            if (__any(nlist == (uint32_t)kGmList)) flush();
        }
    }
    flush();
    if (active) {
        double corr = 1.0;
        if (count < 20u) corr = 0.65 * 0.65;
        if (count < 10u) corr = 0.6 * 0.6;
        if (count < 5u) corr = 0.5 * 0.5;
        if (count < 3u) corr = 0.25 * 0.25;
        const double ratio = (best / second) / corr;
        const bool keep = !(ratio < 0.00001) && best_index > -1 && (ratio < 0.8 * 0.8 || count == 1u);
        best_out[P.off + i] = keep ? best_index : -1;
        ratio_out[P.off + i] = ratio;
    }
}

// one workgroup per pair: compaction in source order, then (only if more than max_n survive) rank by (ratio, position)
// ---- epipolar hashing that saves the work (bins <= kGmMaxBuckets) ------------------------------------------------------
// The tiled scan above tests every (source, destination) pair -- 64 M gates per image pair at 8000 keypoints -- and only
// then compares bins.  Here both keypoint sets are first bucketed by bin (a stable counting sort: ascending index inside
// a bucket, so candidates still arrive in ascending j and `best` / `second` / `count` come out the same), and a workgroup
// of sources of ONE bin scans the destinations of that bin only: bins x fewer gates, same matches bit for bit.

// rank of a lane among the lower lanes holding the same key, and whether it is the first of its group; n_same = group size
__device__ inline uint32_t wave_group_rank(uint32_t key, bool active, bool& leader, uint32_t& n_same) {
    uint32_t rank = 0;
    leader = false;
    n_same = 0;
    uint64_t todo = __ballot(active);
    const uint32_t lane = threadIdx.x & 63u;
    while (todo) {
        const int first = __ffsll((long long)todo) - 1;
        const uint32_t v = (uint32_t)__shfl((int)key, first);
        const uint64_t m = __ballot(active && key == v) & todo;
        if (active && key == v) {
            rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            leader = rank == 0;
            n_same = (uint32_t)__popcll(m);
        }
        todo &= ~m;
    }
    return rank;
}

// grid (pairs, 2): side 0 buckets the source keypoints, side 1 the destination keypoints (and stores their epipolar-line
// records in bucket order).  One workgroup of 1024 threads; rounds of 1024 keypoints in index order.
__global__ __launch_bounds__(1024) void guided_bucket_kernel(const GuidedPair* __restrict__ pairs, uint32_t* __restrict__ src_order,
                                                             uint32_t* __restrict__ dst_order, double* __restrict__ dst_rec,
                                                             uint32_t* __restrict__ starts /* pairs x 2 x (kGmMaxBuckets + 1) */,
                                                             uint32_t* __restrict__ chunk_start /* pairs x (kGmMaxBuckets + 1) */) {
    constexpr int kMaxRounds = PGI_DESC_MAX / 1024;
    __shared__ unsigned short cnt[kMaxRounds * 16][kGmMaxBuckets];  // per (round, wavefront) and bin
    __shared__ uint32_t bin_start[kGmMaxBuckets + 1];
    const GuidedPair P = pairs[blockIdx.x];
    const uint32_t side = blockIdx.y, tid = threadIdx.x, wv = tid >> 6;
    const uint32_t n = side ? P.n2 : P.n1, bins = (uint32_t)P.bins;
    const uint32_t rounds = (n + 1023u) >> 10;
    for (uint32_t i = tid; i < (uint32_t)kMaxRounds * 16u * kGmMaxBuckets; i += 1024u) (&cnt[0][0])[i] = 0;
    __syncthreads();
    auto bin_of = [&](uint32_t i, double rec[4]) -> uint32_t {
        if (side == 0) {
            const double x1 = (double)P.kp1[2 * (size_t)i], y1 = (double)P.kp1[2 * (size_t)i + 1];
            return (uint32_t)epipolar_bin(x1 - P.ep0, -(y1 - P.ep1), P.min_angle, P.range, P.bins);
        }
        const double x2 = (double)P.kp2[2 * (size_t)i], y2 = (double)P.kp2[2 * (size_t)i + 1];
        const double rxc = (P.F[0] * x2 + P.F[3] * y2) + P.F[6];
        const double ryc = (P.F[1] * x2 + P.F[4] * y2) + P.F[7];
        const double rwc = (P.F[2] * x2 + P.F[5] * y2) + P.F[8];
        rec[0] = rxc; rec[1] = ryc; rec[2] = rwc; rec[3] = rxc * rxc + ryc * ryc;
        return (uint32_t)epipolar_bin(ryc, rxc, P.min_angle, P.range, P.bins);
    };
    // pass 1: how many keypoints of each bin every (round, wavefront) holds
    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t i = r * 1024u + tid;
        const bool active = i < n;
        double rec[4];
        const uint32_t b = active ? bin_of(i, rec) : 0u;
        bool leader;
        uint32_t n_same;
        (void)wave_group_rank(b, active, leader, n_same);
        if (active && leader) cnt[r * 16u + wv][b] = (unsigned short)n_same;
    }
    __syncthreads();
    // exclusive prefix over (round, wavefront) for every bin, in place; totals -> bucket starts
    if (tid < bins) {
        uint32_t run = 0;
        for (uint32_t q = 0; q < rounds * 16u; ++q) {
            const uint32_t c = cnt[q][tid];
            cnt[q][tid] = (unsigned short)run;
            run += c;
        }
        bin_start[tid + 1] = run;  // count, turned into a prefix below
    }
    __syncthreads();
    if (tid == 0) {
        bin_start[0] = 0;
        uint32_t chunks = 0;
        uint32_t* cs = chunk_start + (size_t)blockIdx.x * (kGmMaxBuckets + 1);
        for (uint32_t b = 0; b < bins; ++b) {
            if (side == 0) { cs[b] = chunks; chunks += (bin_start[b + 1] + 255u) >> 8; }
            bin_start[b + 1] += bin_start[b];
        }
        if (side == 0) cs[bins] = chunks;
    }
    __syncthreads();
    uint32_t* st = starts + ((size_t)blockIdx.x * 2 + side) * (kGmMaxBuckets + 1);
    if (tid <= bins) st[tid] = bin_start[tid];
    // pass 2: place every keypoint (stable: rounds, wavefronts and lanes are all visited in index order)
    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t i = r * 1024u + tid;
        const bool active = i < n;
        double rec[4] = {0, 0, 0, 0};
        const uint32_t b = active ? bin_of(i, rec) : 0u;
        bool leader;
        uint32_t n_same;
        const uint32_t rank = wave_group_rank(b, active, leader, n_same);
        if (!active) continue;
        const uint32_t pos = bin_start[b] + cnt[r * 16u + wv][b] + rank;
        if (side == 0) {
            src_order[P.off + pos] = i;
        } else {
            dst_order[P.off2 + pos] = i;
            double* o = dst_rec + 4 * (P.off2 + pos);
            o[0] = rec[0]; o[1] = rec[1]; o[2] = rec[2]; o[3] = rec[3];
        }
    }
}

// grid (chunks, pairs): chunk = up to 256 source keypoints of one bin against the destination keypoints of that bin
__global__ __launch_bounds__(256) void guided_scan_binned_kernel(const GuidedPair* __restrict__ pairs, const uint32_t* __restrict__ src_order,
                                                                 const uint32_t* __restrict__ dst_order, const double* __restrict__ dst_rec,
                                                                 const uint32_t* __restrict__ starts, const uint32_t* __restrict__ chunk_start,
                                                                 int32_t* __restrict__ best_out, double* __restrict__ ratio_out) {
    __shared__ float4 recf[kGmTile];  // the epipolar-line records rounded to f32: the cheap first look below
    __shared__ uint32_t list[kGmList][256];  // positions (in the bin-sorted destination order) that survived it
    __shared__ uint32_t s_bin;
    const GuidedPair P = pairs[blockIdx.y];
    const uint32_t tid = threadIdx.x, bins = (uint32_t)P.bins;
    const uint32_t* cs = chunk_start + (size_t)blockIdx.y * (kGmMaxBuckets + 1);
    if (blockIdx.x >= cs[bins]) return;
    if (tid == 0) {
        uint32_t b = 0;
        while (b + 1 < bins && cs[b + 1] <= blockIdx.x) ++b;
        s_bin = b;
    }
    __syncthreads();
    const uint32_t bin = s_bin;
    const uint32_t* st1 = starts + ((size_t)blockIdx.y * 2 + 0) * (kGmMaxBuckets + 1);
    const uint32_t* st2 = starts + ((size_t)blockIdx.y * 2 + 1) * (kGmMaxBuckets + 1);
    const uint32_t s_pos = st1[bin] + (blockIdx.x - cs[bin]) * 256u + tid;
    const bool active = s_pos < st1[bin + 1];
    const uint32_t i = active ? src_order[P.off + s_pos] : 0u;
    double x1 = 0.0, y1 = 0.0;
    if (active) { x1 = (double)P.kp1[2 * (size_t)i]; y1 = (double)P.kp1[2 * (size_t)i + 1]; }
    const double rx = (P.F[0] * x1 + P.F[1] * y1) + P.F[2];
    const double ry = (P.F[3] * x1 + P.F[4] * y1) + P.F[5];
    const double b1 = rx * rx + ry * ry;
    const float x1f = (float)x1, y1f = (float)y1, b1f = (float)b1;  // (x1, y1 are f32 values: exact)
    double best = DBL_MAX, second = DBL_MAX;
    int32_t best_index = -1;
    uint32_t count = 0, nlist = 0;
    const float4* arow = reinterpret_cast<const float4*>(P.d1 + (size_t)i * kD);
    auto flush = [&]() {
        uint32_t longest = nlist;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)longest, m); longest = o > longest ? o : longest; }
        for (uint32_t c = 0; c < longest; ++c) {
            bool cand = false;
            uint32_t j = 0;
            if (c < nlist) {  // the exact gate (matcher.h:286-301 as restated in the oracle), on the survivors only
                const uint32_t pos = list[c][tid];
                const double* o = dst_rec + 4 * (P.off2 + pos);
                const double rxc = o[0], ryc = o[1], rwc = o[2], a1 = o[3];
                const double r = (x1 * rxc + y1 * ryc) + rwc;
                const double num = (r * r) * (a1 + b1), den = a1 * b1;
                const bool surely_far = den > 0.0 && num >= 0.57 * den;
                if (!surely_far) {
                    const double dist = num / den;
                    if (!(dist >= 0.75 * 0.75)) { cand = true; j = dst_order[P.off2 + pos]; }
                }
            }
            if (cand) {
                const float4* brow = reinterpret_cast<const float4*>(P.d2 + (size_t)j * kD);
                double dd = 0.0;
                for (int k = 0; k < kD / 4; ++k) {
                    const float4 a = arow[k], b = brow[k];
                    const double v0 = (double)(a.x - b.x), v1 = (double)(a.y - b.y), v2 = (double)(a.z - b.z), v3 = (double)(a.w - b.w);
                    dd = dd + v0 * v0;
                    dd = dd + v1 * v1;
                    dd = dd + v2 * v2;
                    dd = dd + v3 * v3;
                }
                ++count;
                if (dd < best) { second = best; best = dd; best_index = (int32_t)j; }
            }
        }
        nlist = 0;
    };
    const uint32_t d0 = st2[bin], d1e = st2[bin + 1];
    for (uint32_t j0 = d0; j0 < d1e; j0 += (uint32_t)kGmTile) {
        __syncthreads();
        const uint32_t lim = d1e - j0 < (uint32_t)kGmTile ? d1e - j0 : (uint32_t)kGmTile;
        for (uint32_t jj = tid; jj < lim; jj += 256u) {
            const double* o = dst_rec + 4 * (P.off2 + j0 + jj);
            recf[jj] = make_float4((float)o[0], (float)o[1], (float)o[2], (float)o[3]);
        }
        __syncthreads();
        for (uint32_t jj = 0; jj < lim; ++jj) {
            // First look in f32.  When the epipole lies outside the image nearly all keypoints share a few angular bins and a
            // source keypoint meets thousands of destination ones here, almost all of them far from its epipolar line; the
            // double-precision gate ran at half rate for every one of them.  With q the f32-rounded record,
            // |rf - r| <= 5 * 2^-24 * S for S = |x1 rxc| + |y1 ryc| + |rwc| (three rounded inputs, two fused operations), so
            // t = |rf| - 2^-20 S is a lower bound of |r| with a factor-three margin, and t^2 (a1 + b1) >= 0.60 a1 b1 in f32
            // (relative error ~1e-6) implies r^2 (a1 + b1) >= 0.57 a1 b1 in f64 -- a pair the exact gate rejects.  NaN, an
            // overflowing sum or an underflowing denominator make the test false: such a pair goes to the exact gate, which
            // runs in flush() on the survivors (a few per cent), in the same order as before.
            const float4 q = recf[jj];
            const float rf = fmaf(x1f, q.x, fmaf(y1f, q.y, q.z));
            const float sabs = fmaf(fabsf(x1f), fabsf(q.x), fmaf(fabsf(y1f), fabsf(q.y), fabsf(q.z)));
            const float tl = fabsf(rf) - 9.5367431640625e-7f * sabs;
            const float denf = q.w * b1f;
            const float sumf = q.w + b1f;
            const bool far32 = tl > 0.0f && denf > 1.0e-30f && denf < 1.0e30f && sumf < 1.0e30f && (tl * tl) * sumf >= 0.60f * denf;  // (finite right-hand side)
            if (active && !far32) { list[nlist][tid] = j0 + jj; ++nlist; }
            if (__any(nlist == (uint32_t)kGmList)) flush();
        }
    }
    flush();
    if (active) {
        double corr = 1.0;
        if (count < 20u) corr = 0.65 * 0.65;
        if (count < 10u) corr = 0.6 * 0.6;
        if (count < 5u) corr = 0.5 * 0.5;
        if (count < 3u) corr = 0.25 * 0.25;
        const double ratio = (best / second) / corr;
        const bool keep = !(ratio < 0.00001) && best_index > -1 && (ratio < 0.8 * 0.8 || count == 1u);
        best_out[P.off + i] = keep ? best_index : -1;
        ratio_out[P.off + i] = ratio;
    }
}

// ---- the same candidates found through a fine angular index (round 3) -----------------------------------------------------
// The 45 bins of the reference save little when the epipole lies far outside the image -- the usual case: nearly all
// keypoints then share two or three bins, and where a source's bin coincides with a populated destination bin the source
// still meets hundreds to thousands of destinations (config 3 from features: 51 % of the GPU time).  But the gate itself
// confines the candidates far more tightly: dist = r^2 (a1 + b1) / (a1 b1) >= r^2 / a1 = (distance of x1 from the
// destination's epipolar line l_j = F^T x2 in the SOURCE image)^2, every l_j passes through the source epipole e, so a
// destination can pass the gate of a source at distance d from e only if the angle between (x1 - e) and l_j is below
// asin(0.75 / d) -- a window of a few 1e-4 rad in the normal angle of l_j, the very angle the reference bins by.
// Destinations are counting-sorted by that angle at kGaBuckets times the resolution of... the pair's own range
// [min_angle, min_angle + range] (guided_angle_bucket_kernel); a bin is a contiguous run of fine buckets, and a source
// visits only the fine buckets that lie BOTH in its bin and in its window (+- one for rounding), applies the same bin
// equality and exact gate as before and evaluates the survivors in ascending destination index -- the order the reference
// meets them in, which `second` depends on.  Degenerate records (zero / non-finite normal or angle) live in an extra bucket
// every source visits; a source with an odd window visits its whole bin.
constexpr int kGaBuckets = 8192;
constexpr int kGaList = 16;    // candidates a thread collects per round
struct GaEntry {      // 48 B per destination keypoint, in bucket order
    double rxc, ryc, rwc, a1;
    uint32_t j;
    int32_t bin;
    uint32_t pad[2];
};
// the reference's angle of a line normal in degrees, folded to (0, 180] (matcher.h:292-296), and its place in the pair's range
__device__ __forceinline__ double ga_angle(double ny, double nx) {
    double angle = kRadianToDegree * atan2(ny, nx) + 180.0;
    if (angle > 180) angle -= 180;
    return angle;
}
// fine bucket of an angle inside [amin, amax], the span the pair's destination angles really cover (far narrower than
// the corner-derived [min_angle, min_angle + range] the bins divide); clamped at both ends; -1: not a number
__device__ __forceinline__ int ga_fine(double angle, double amin, double amax) {
    if (!(angle == angle)) return -1;
    const double f = (angle - amin) * ((double)kGaBuckets / fmax(amax - amin, 1.0e-9));
    return !(f > 0.0) ? 0 : (f >= (double)(kGaBuckets - 1) ? kGaBuckets - 1 : (int)f);
}
// centre of a source keypoint's window in the folded degrees: the normal of the line through the epipole and the keypoint
__device__ __forceinline__ double ga_source_centre(double vx, double vy) {
    double c = kRadianToDegree * atan2(vy, vx) + 90.0;
    c -= floor(c / 180.0) * 180.0;  // [0, 180)
    return c;
}
// grid (pairs), 1024 threads: records, bins and fine buckets of the destination keypoints; then the source keypoints in the
// order of their windows, so that the threads of a scanning workgroup visit the same few buckets and descriptor rows
__global__ __launch_bounds__(1024) void guided_angle_bucket_kernel(const GuidedPair* __restrict__ pairs, GaEntry* __restrict__ entries,
                                                                   uint32_t* __restrict__ keys /* per destination: its bucket */,
                                                                   uint32_t* __restrict__ starts /* pairs x (kGaBuckets + 2) */,
                                                                   double* __restrict__ spans /* pairs x {smallest, largest angle} */,
                                                                   uint32_t* __restrict__ order /* per pair: sources by window */) {
    __shared__ uint32_t cnt[kGaBuckets + 1];
    __shared__ uint32_t part[1024];
    const GuidedPair P = pairs[blockIdx.x];
    const uint32_t tid = threadIdx.x;
    __shared__ unsigned long long span[2];  // bit patterns of the smallest / largest angle (positive doubles order like integers)
    for (uint32_t i = tid; i <= (uint32_t)kGaBuckets; i += 1024u) cnt[i] = 0;
    if (tid == 0) { span[0] = ~0ull; span[1] = 0ull; }
    __syncthreads();
    for (uint32_t j = tid; j < P.n2; j += 1024u) {
        const double x2 = (double)P.kp2[2 * (size_t)j], y2 = (double)P.kp2[2 * (size_t)j + 1];
        const double rxc = (P.F[0] * x2 + P.F[3] * y2) + P.F[6];
        const double ryc = (P.F[1] * x2 + P.F[4] * y2) + P.F[7];
        const double a1 = rxc * rxc + ryc * ryc;
        if (a1 > 0.0 && a1 < 1.0e300) {
            const double ang = ga_angle(ryc, rxc);
            if (ang > 0.0) {  // (0, 180]; NaN fails
                atomicMin(&span[0], (unsigned long long)__double_as_longlong(ang));
                atomicMax(&span[1], (unsigned long long)__double_as_longlong(ang));
            }
        }
    }
    __syncthreads();
    const double amin = span[0] == ~0ull ? 0.0 : __longlong_as_double((long long)span[0]);
    const double amax = span[1] == 0ull ? 180.0 : __longlong_as_double((long long)span[1]);
    if (tid == 0) {  // the scan maps its intervals through the same span
        double* sp = reinterpret_cast<double*>(spans) + 2 * (size_t)blockIdx.x;
        sp[0] = amin;
        sp[1] = amax;
    }
    for (uint32_t j = tid; j < P.n2; j += 1024u) {
        const double x2 = (double)P.kp2[2 * (size_t)j], y2 = (double)P.kp2[2 * (size_t)j + 1];
        const double rxc = (P.F[0] * x2 + P.F[3] * y2) + P.F[6];
        const double ryc = (P.F[1] * x2 + P.F[4] * y2) + P.F[7];
        const double a1 = rxc * rxc + ryc * ryc;
        int kb = -1;
        if (a1 > 0.0 && a1 < 1.0e300) kb = ga_fine(ga_angle(ryc, rxc), amin, amax);
        const uint32_t k = kb < 0 ? (uint32_t)kGaBuckets : (uint32_t)kb;  // degenerate: the bucket every source visits
        keys[P.off2 + j] = k;
        atomicAdd(&cnt[k], 1u);
    }
    __syncthreads();
    // exclusive prefix over the kGaBuckets regular counters: 1024 threads x 8
    uint32_t local[8], sum = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { local[k] = cnt[tid * 8u + k]; sum += local[k]; }
    part[tid] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {
        const uint32_t v = tid >= d ? part[tid - d] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t run = part[tid] - sum;
    uint32_t* st = starts + (size_t)blockIdx.x * (kGaBuckets + 2);
    const uint32_t n_deg = cnt[kGaBuckets];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) { st[tid * 8u + k] = run; cnt[tid * 8u + k] = run; run += local[k]; }
    if (tid == 1023u) {
        st[kGaBuckets] = run;                         // the degenerate bucket follows the regular ones
        st[kGaBuckets + 1] = run + n_deg;
        cnt[kGaBuckets] = run;
    }
    __syncthreads();
    for (uint32_t j = tid; j < P.n2; j += 1024u) {
        const double x2 = (double)P.kp2[2 * (size_t)j], y2 = (double)P.kp2[2 * (size_t)j + 1];
        const double rxc = (P.F[0] * x2 + P.F[3] * y2) + P.F[6];
        const double ryc = (P.F[1] * x2 + P.F[4] * y2) + P.F[7];
        const double rwc = (P.F[2] * x2 + P.F[5] * y2) + P.F[8];
        const uint32_t pos = atomicAdd(&cnt[keys[P.off2 + j]], 1u);  // (order inside a bucket is arbitrary: candidates are sorted later)
        GaEntry e;
        e.rxc = rxc; e.ryc = ryc; e.rwc = rwc; e.a1 = rxc * rxc + ryc * ryc;
        e.j = j;
        e.bin = epipolar_bin(ryc, rxc, P.min_angle, P.range, P.bins);
        e.pad[0] = e.pad[1] = 0;
        entries[P.off2 + pos] = e;
    }
    // the sources, counting-sorted by the bucket of their window centre (any order is correct; this one is cache friendly)
    __syncthreads();
    for (uint32_t i = tid; i <= (uint32_t)kGaBuckets; i += 1024u) cnt[i] = 0;
    __syncthreads();
    auto source_key = [&](uint32_t i) -> uint32_t {
        const double vx = (double)P.kp1[2 * (size_t)i] - P.ep0, vy = (double)P.kp1[2 * (size_t)i + 1] - P.ep1;
        const int kb = ga_fine(ga_source_centre(vx, vy), amin, amax);
        return kb < 0 ? 0u : (uint32_t)kb;
    };
    for (uint32_t i = tid; i < P.n1; i += 1024u) atomicAdd(&cnt[source_key(i)], 1u);
    __syncthreads();
    sum = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { local[k] = cnt[tid * 8u + k]; sum += local[k]; }
    part[tid] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {
        const uint32_t v = tid >= d ? part[tid - d] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    run = part[tid] - sum;
#pragma unroll
    for (int k = 0; k < 8; ++k) { cnt[tid * 8u + k] = run; run += local[k]; }
    __syncthreads();
    for (uint32_t i = tid; i < P.n1; i += 1024u) order[P.off + atomicAdd(&cnt[source_key(i)], 1u)] = i;
}

constexpr int kGtRows = 16;        // destination descriptor rows a wavefront stages in LDS at a time
constexpr int kGtStride = kD + 4;  // floats per staged row: 16-byte aligned rows, rows 16 apart share banks
constexpr int kGtRecs = 176;       // destination records staged at a time, in the same memory (48 B each)
// grid (ceil(n1 / (64 / G)), pairs), ONE wavefront per workgroup: 64 / G sources that are neighbours in window order, G lanes
// each (lane g of a source takes the records at positions = g mod G, so the lanes of a source share its work evenly).
//   (1) every lane keeps its source's descriptor in 128 registers;
//   (2) the records of the few fine buckets the gate window and the bin leave are staged in LDS from the smallest position any
//       lane still needs; a lane lists the gate-passing destinations of its share (destination index, record position), the
//       kGaList smallest destination indices in sorted registers;
//   (3) the wavefront stages the records' descriptor rows kGtRows at a time -- neighbours in window order want the same rows,
//       so a row is fetched once per wavefront instead of once per (source, candidate) -- and each lane sums the squared
//       differences of its listed candidates that lie in the staged rows, sequentially in double (fma(v, v, dd) is
//       dd + v * v exactly: the product of two f32-valued doubles is exact), every bit of matcher.h:352-371;
//   (4) matcher.h:352-371 meets the candidates in ascending destination index and keeps `best` and `second` = the best BEFORE
//       the last improvement.  The last improvement is the first candidate that attains the overall minimum, so
//       second = min{distance of c : index of c < index of that candidate} -- which needs no particular evaluation order.
//       A source with more than its lists hold is handled in rounds of ascending index ranges.
template <int G>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void guided_scan_tile_kernel(
    const GuidedPair* __restrict__ pairs, const GaEntry* __restrict__ entries, const uint32_t* __restrict__ starts,
    const double* __restrict__ spans, const uint32_t* __restrict__ order, int32_t* __restrict__ best_out, double* __restrict__ ratio_out) {
    static_assert(G == 1 || G == 2 || G == 4, "lanes per source");
    constexpr uint32_t kSources = 64u / G;
    __shared__ uint32_t list[kGaList][64];  // (destination index << 16 | record position), ascending
    __shared__ double dist[kGaList][64];
    __shared__ __attribute__((aligned(16))) float tile[kGtRows][kGtStride];
    const GuidedPair P = pairs[blockIdx.y];
    const uint32_t lane = threadIdx.x, slot = lane / G, g = lane % G;
    if (blockIdx.x * kSources >= P.n1) return;      // (wavefront-uniform; in a live wavefront every lane stays for the shared steps)
    const bool active = blockIdx.x * kSources + slot < P.n1;
    const uint32_t i = active ? order[P.off + blockIdx.x * kSources + slot] : 0u;
    const double x1 = (double)P.kp1[2 * (size_t)i], y1 = (double)P.kp1[2 * (size_t)i + 1];
    const double rx = (P.F[0] * x1 + P.F[1] * y1) + P.F[2];
    const double ry = (P.F[3] * x1 + P.F[4] * y1) + P.F[5];
    const double b1 = rx * rx + ry * ry;
    const double vx = x1 - P.ep0, vy = y1 - P.ep1;
    const int32_t my_bin = epipolar_bin(vx, -vy, P.min_angle, P.range, P.bins);  // as guided_bucket_kernel's side 0
    const uint32_t* st = starts + (size_t)blockIdx.y * (kGaBuckets + 2);
    const double amin = spans[2 * (size_t)blockIdx.y], amax = spans[2 * (size_t)blockIdx.y + 1];
    // the fine buckets of this source's bin: bin = round((bins - 1) u), u = (angle - min_angle) / range  <=>  the angle lies
    // between min_angle + (bin -+ 0.5) range / (bins - 1) (either order: range is negative for an in-image epipole)
    int f0 = 0, f1 = kGaBuckets - 1;
    if (P.bins > 1 && P.range == P.range && P.range != 0.0) {
        const double step = P.range / (double)(P.bins - 1);
        const double ea = P.min_angle + ((double)my_bin - 0.5) * step, eb = P.min_angle + ((double)my_bin + 0.5) * step;
        const bool open_lo = my_bin == 0, open_hi = my_bin == P.bins - 1;  // the end bins also take everything beyond the range
        double lo_ang = fmin(ea, eb), hi_ang = fmax(ea, eb);
        if ((open_lo && step > 0.0) || (open_hi && step < 0.0)) lo_ang = -1.0e300;
        if ((open_hi && step > 0.0) || (open_lo && step < 0.0)) hi_ang = 1.0e300;
        f0 = max(0, ga_fine(lo_ang, amin, amax) - 1);
        f1 = min(kGaBuckets - 1, ga_fine(hi_ang, amin, amax) + 1);
    }
    // the window the gate leaves: the line through e and x1 has direction atan2(vy, vx); a gate-passing l_j deviates from it by
    // less than asin(0.75 / d), and its NORMAL is a quarter turn away.  In the reference's folded degrees:
    int g0[2] = {0, 0}, g1[2] = {kGaBuckets - 1, -1};  // up to two pieces (the window may straddle the fold at 0 / 180)
    const double d2 = vx * vx + vy * vy;
    if (d2 > 4.0 && d2 < 1.0e300 && b1 > 0.0 && b1 < 1.0e300) {
        const double c = ga_source_centre(vx, vy);
        const double half = kRadianToDegree * asin(0.75 / sqrt(d2)) * 1.0001 + 1.0e-5;  // margin for rounding
        if (2.0 * half < 179.0) {
            double lo[2] = {c - half, 0.0}, hi[2] = {c + half, -1.0};  // pieces of [c - half, c + half] folded into (0, 180]
            if (lo[0] <= 0.0) { lo[1] = lo[0] + 180.0; hi[1] = 180.0; lo[0] = 0.0; }
            else if (hi[0] > 180.0) { lo[1] = 0.0; hi[1] = hi[0] - 180.0; hi[0] = 180.0; }
            for (int k = 0; k < 2; ++k) {
                if (hi[k] < lo[k]) { g0[k] = 0; g1[k] = -1; continue; }
                const int fa = ga_fine(lo[k], amin, amax), fb = ga_fine(hi[k], amin, amax);
                if (fa < 0 || fb < 0) { g0[k] = 0; g1[k] = kGaBuckets - 1; continue; }
                g0[k] = max(0, fa - 1);
                g1[k] = min(kGaBuckets - 1, fb + 1);
            }
        }
    }
    // the fine-bucket intervals to visit: window pieces cut to the bin's interval; merged when the margins made them meet
    int v0[2], v1[2];
    for (int k = 0; k < 2; ++k) { v0[k] = max(g0[k], f0); v1[k] = min(g1[k], f1); }
    if (v1[0] >= v0[0] && v1[1] >= v0[1] && v0[1] <= v1[0] + 1 && v0[0] <= v1[1] + 1) {
        v0[0] = min(v0[0], v0[1]);
        v1[0] = max(v1[0], v1[1]);
        v1[1] = v0[1] - 1;  // empty
    }
    auto wave_sync = [] {  // LDS written by some lanes is read by others of the same wavefront
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // (1) through the tile, so that the 512-byte rows are read whole: kGtRows sources at a time, their lanes keep the row
    float a[kD];
    static_assert(kGtRows == 16, "sixteen rows in flight per pass");
    for (uint32_t part = 0; part < kSources / kGtRows; ++part) {
        {
            float2 bv[16];
#pragma unroll
            for (uint32_t u = 0; u < 16u; ++u) {
                const uint32_t ir = (uint32_t)__builtin_amdgcn_readlane((int)i, (int)((kGtRows * part + u) * G));
                bv[u] = reinterpret_cast<const float2*>(P.d1 + (size_t)ir * kD)[lane];
            }
#pragma unroll
            for (uint32_t u = 0; u < 16u; ++u) *reinterpret_cast<float2*>(&tile[u][2 * lane]) = bv[u];
        }
        wave_sync();
        if (slot / (uint32_t)kGtRows == part) {
            const float4* row = reinterpret_cast<const float4*>(&tile[slot % (uint32_t)kGtRows][0]);
#pragma unroll
            for (int q = 0; q < kD / 4; ++q) {
                const float4 t = row[q];
                a[4 * q] = t.x; a[4 * q + 1] = t.y; a[4 * q + 2] = t.z; a[4 * q + 3] = t.w;
            }
        }
        wave_sync();
    }
    // this source's record ranges: the (at most two) window pieces and the degenerate records every source visits
    uint32_t re0[3] = {0u, 0u, st[kGaBuckets]}, re1[3] = {0u, 0u, st[kGaBuckets + 1]};
    for (int k = 0; k < 2; ++k)
        if (v1[k] >= v0[k]) { re0[k] = st[v0[k]]; re1[k] = st[v1[k] + 1]; }
    if (!active) re1[0] = re0[0] = re1[1] = re0[1] = re1[2] = re0[2] = 0u;
    static_assert(sizeof(GaEntry) == 48 && kGtRecs * sizeof(GaEntry) <= sizeof(float) * kGtRows * kGtStride, "records are staged in the tile");
    GaEntry* recs = reinterpret_cast<GaEntry*>(&tile[0][0]);
    // (2) this lane's gate-passing destinations with index > after: the kGaList smallest go to `list`; returns how many in all
    auto collect = [&](int64_t after) -> uint32_t {
        uint32_t total = 0;
        uint32_t L[kGaList];  // ascending; unused places hold the largest number
#pragma unroll
        for (int p2 = 0; p2 < kGaList; ++p2) L[p2] = 0xFFFFFFFFu;
        uint32_t c0[3] = {re0[0], re0[1], re0[2]};  // cursors
        while (true) {
            uint32_t need = 0xFFFFFFFFu;
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (c0[k] < re1[k]) need = min(need, c0[k]);
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) need = min(need, (uint32_t)__shfl_xor((int)need, d));
            if (need == 0xFFFFFFFFu) break;
            const uint32_t E = need, nrec = min((uint32_t)kGtRecs, P.n2 - E);
            wave_sync();
            {   // 48-byte records as 16-byte pieces, coalesced
                const uint4* src = reinterpret_cast<const uint4*>(entries + P.off2 + E);
                uint4* dst = reinterpret_cast<uint4*>(recs);
                for (uint32_t q = lane; q < nrec * 3u; q += 64u) dst[q] = src[q];
            }
            wave_sync();
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const uint32_t lo = max(c0[k], E), hi = min(re1[k], E + nrec);
                for (uint32_t e = lo + (g + G - lo % G) % G; e < hi; e += G) {  // this lane's share: positions = g mod G
                    const GaEntry en = recs[e - E];
                    if (en.bin != my_bin || (int64_t)en.j <= after) continue;
                    const double r = (x1 * en.rxc + y1 * en.ryc) + en.rwc;
                    const double num = (r * r) * (en.a1 + b1), den = en.a1 * b1;
                    if (den > 0.0 && num >= 0.57 * den) continue;  // surely far (the pre-test of the binned scan)
                    if (!(den > 0.0 && num <= 0.56 * den)) {        // not surely near either: the reference's quotient decides
                        const double dd = num / den;
                        if (dd >= 0.75 * 0.75) continue;
                    }
                    ++total;
                    // sorted insertion, the largest dropping out when full: L'[p] = median(L[p - 1], L[p], key)
                    const uint32_t key = (en.j << 16) | e;
#pragma unroll
                    for (int p2 = kGaList - 1; p2 >= 1; --p2) L[p2] = max(L[p2 - 1], min(L[p2], key));
                    L[0] = min(L[0], key);
                }
                if (hi > c0[k]) c0[k] = hi;
            }
        }
#pragma unroll
        for (int p2 = 0; p2 < kGaList; ++p2) list[p2][lane] = L[p2];
        return total;
    };
    auto group_min_u32 = [](uint32_t v) {
#pragma unroll
        for (int d = 1; d < G; d <<= 1) v = min(v, (uint32_t)__shfl_xor((int)v, d));
        return v;
    };
    double best = DBL_MAX, second = DBL_MAX;
    int32_t best_index = -1;
    uint32_t count = 0;
    int64_t after = -1;
    bool more = true;  // (the same for the lanes of a source)
    while (__any(more)) {
        const uint32_t total = collect(more ? after : (int64_t)0x7FFFFFFF);  // (every lane takes part in the staging; a finished source lists nothing)
        // a lane that found more than it lists holds all of its share up to its largest listed index: the round covers the
        // indices up to the smallest such bound among the source's lanes; the rest waits for the next round
        const uint32_t jcap = group_min_u32(total > (uint32_t)kGaList ? (list[kGaList - 1][lane] >> 16) : 0xFFFFFFFFu);
        uint32_t nlist = 0;
        for (uint32_t c = 0; c < min(total, (uint32_t)kGaList); ++c) nlist += (list[c][lane] >> 16) <= jcap ? 1u : 0u;  // (a prefix: ascending)
        // (3) until no lane has an unevaluated candidate: stage the rows from the smallest pending record position on
        uint32_t pend = nlist ? (0xFFFFFFFFu >> (32u - nlist)) : 0u;
        while (true) {
            uint32_t emin = 0xFFFFFFFFu;
            for (uint32_t m = pend; m; m &= m - 1u) emin = min(emin, list[__builtin_ctz(m)][lane] & 0xFFFFu);
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) emin = min(emin, (uint32_t)__shfl_xor((int)emin, d));
            if (emin == 0xFFFFFFFFu) break;
            const uint32_t E = emin, nrow = min((uint32_t)kGtRows, P.n2 - E);
            const uint32_t rj = lane < nrow ? entries[P.off2 + E + lane].j : 0u;
            wave_sync();
            {
                float2 bv[16];
#pragma unroll
                for (uint32_t u = 0; u < 16u; ++u) {
                    const uint32_t j = (uint32_t)__builtin_amdgcn_readlane((int)rj, (int)u);
                    if (u < nrow) bv[u] = reinterpret_cast<const float2*>(P.d2 + (size_t)j * kD)[lane];
                }
#pragma unroll
                for (uint32_t u = 0; u < 16u; ++u)
                    if (u < nrow) *reinterpret_cast<float2*>(&tile[u][2 * lane]) = bv[u];
            }
            wave_sync();
            uint32_t mine = 0;  // this lane's pending candidates inside the staged rows
            for (uint32_t m = pend; m; m &= m - 1u) {
                const uint32_t c = (uint32_t)__builtin_ctz(m);
                if ((list[c][lane] & 0xFFFFu) - E < nrow) mine |= 1u << c;
            }
            pend &= ~mine;
            while (__any(mine != 0u)) {  // two candidates at a time: two independent chains of dependent additions
                if (mine) {
                    const uint32_t ca = (uint32_t)__builtin_ctz(mine);
                    mine &= mine - 1u;
                    const uint32_t cb = mine ? (uint32_t)__builtin_ctz(mine) : ca;
                    mine &= mine - 1u;  // (0 & anything = 0)
                    const float4* ta = reinterpret_cast<const float4*>(&tile[(list[ca][lane] & 0xFFFFu) - E][0]);
                    const float4* tb = reinterpret_cast<const float4*>(&tile[(list[cb][lane] & 0xFFFFu) - E][0]);
                    double da = 0.0, db = 0.0;
#pragma unroll
                    for (int q = 0; q < kD / 4; ++q) {
                        const float4 u = ta[q], v = tb[q];
                        // (two f32 differences per instruction: v_pk_add_f32; each is the same IEEE subtraction)
                        typedef float pk2 __attribute__((ext_vector_type(2)));
                        const pk2 a01 = {a[4 * q], a[4 * q + 1]}, a23 = {a[4 * q + 2], a[4 * q + 3]};
                        const pk2 du01 = a01 - pk2{u.x, u.y}, du23 = a23 - pk2{u.z, u.w}, dv01 = a01 - pk2{v.x, v.y}, dv23 = a23 - pk2{v.z, v.w};
                        const double u0 = (double)du01.x, u1 = (double)du01.y, u2 = (double)du23.x, u3 = (double)du23.y;
                        const double w0 = (double)dv01.x, w1 = (double)dv01.y, w2 = (double)dv23.x, w3 = (double)dv23.y;
                        da = fma(u0, u0, da); db = fma(w0, w0, db);
                        da = fma(u1, u1, da); db = fma(w1, w1, db);
                        da = fma(u2, u2, da); db = fma(w2, w2, db);
                        da = fma(u3, u3, da); db = fma(w3, w3, db);
                    }
                    dist[ca][lane] = da;
                    dist[cb][lane] = db;
                }
            }
        }
        // (4) the round's smallest distance (ties: the smallest index), then the smallest distance among smaller indices
        double m_d = DBL_MAX;
        uint32_t m_j = 0xFFFFFFFFu;
        for (uint32_t c = 0; c < nlist; ++c) {  // ascending index: the first minimum wins
            const double dc = dist[c][lane];
            if (dc < m_d) { m_d = dc; m_j = list[c][lane] >> 16; }
        }
        uint32_t cnt = nlist;
#pragma unroll
        for (int d = 1; d < G; d <<= 1) {
            const double od = __shfl_xor(m_d, d);
            const uint32_t oj = (uint32_t)__shfl_xor((int)m_j, d);
            if (od < m_d || (od == m_d && oj < m_j)) { m_d = od; m_j = oj; }
            cnt += (uint32_t)__shfl_xor((int)cnt, d);
        }
        double s_d = DBL_MAX;
        for (uint32_t c = 0; c < nlist; ++c)
            if ((list[c][lane] >> 16) < m_j) s_d = fmin(s_d, dist[c][lane]);
#pragma unroll
        for (int d = 1; d < G; d <<= 1) s_d = fmin(s_d, __shfl_xor(s_d, d));
        count += cnt;
        if (m_d < best) {  // the earlier rounds held smaller indices: their best is a predecessor of this round's first minimum
            second = fmin(best, s_d);
            best = m_d;
            best_index = (int32_t)m_j;
        }
        more = more && jcap != 0xFFFFFFFFu;
        if (more) after = (int64_t)jcap;
    }
    if (!active || g != 0u) return;
    double corr = 1.0;
    if (count < 20u) corr = 0.65 * 0.65;
    if (count < 10u) corr = 0.6 * 0.6;
    if (count < 5u) corr = 0.5 * 0.5;
    if (count < 3u) corr = 0.25 * 0.25;
    const double ratio = (best / second) / corr;
    const bool keep = !(ratio < 0.00001) && best_index > -1 && (ratio < 0.8 * 0.8 || count == 1u);
    best_out[P.off + i] = keep ? best_index : -1;
    ratio_out[P.off + i] = ratio;
}

constexpr int kGfCap = 20;  // candidates a source lists per round in guided_scan_flat_kernel (16: 7 workgroups per CU but more second
                            // rounds, 24: 5 workgroups; measured 3.04 / 3.03 / 3.33 ms per 510 pairs)
// The same scan with the descriptor sums taken off the source's own lane (round 5).  In the tile kernel above a lane sums only
// its own source's candidates among the sixteen staged rows -- one or two, while the wavefront waits for the lane that has
// five (28 % of the lanes' additions were useful).  Here, grid (8 x ceil(blocks / 8)), ONE wavefront per workgroup, one lane
// per source (64 neighbours in window order):
//   (1) every lane fetches its source's descriptor into 128 registers itself (32 loads of 16 bytes; the 64 lines a load touches
//       serve the next seven), in flight during the window arithmetic;
//   (2) as above, but a lane keeps its gate-passing destinations in its column of `list` ordered by RECORD POSITION (they are
//       met nearly in that order: an insertion that rarely moves anything); a source with more than C of them keeps the C
//       smallest destination indices and takes another round;
//   (3) per sixteen staged rows the lanes pool their candidates: (source lane, candidate a, candidate b) items -- two
//       candidates of one source, so that its descriptor travels once for two sums -- are dealt 64 at a time, one per lane,
//       whichever lane the source lives in: the source's descriptor comes through ds_bpermute, the two rows from LDS, and the
//       two sums run as independent chains, sequentially in double as above.  A lane's candidates inside a tile are a run of
//       its ordered list: sixteen entries are read at once, the run's length, the items and the position the lane waits for
//       next come from them without a loop; offsets and the next tile come over the DPP network;
//   (4) as above, without relying on the lists' order: best = the smallest (distance, index), second = the smallest distance
//       among smaller indices.
// Blocks are renumbered so that each XCD takes a contiguous run of them: neighbours in window order want the same records and
// rows, and now find them in their own L2.
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {  // every lane active; the result is wavefront-uniform
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false));  // row_half_mirror
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false));  // row_mirror: every lane holds its row's
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xA, 0xF, false));  // row_bcast:15 into rows 1 and 3
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xC, 0xF, false));  // row_bcast:31 into rows 2 and 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v) {  // every lane active
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);  // row_shr:1 (a lane without a source adds 0)
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);  // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2 and 3
    return v;
}
// PGI_GUIDED_PHASES=1 (diagnostics): PH counts the clock ticks a wavefront spends per phase into g_guided_phase
__device__ unsigned long long g_guided_phase[8];
// The scan as two kernels (the default since r05; PGI_GUIDED_SPLIT=0 keeps it in one).  What a wavefront's sums need is a LIST: per round the tiles (first record,
// items) and the items themselves.  MODE 1 of the body below produces exactly that and nothing else -- window arithmetic, gate
// pass, dealing; no descriptor in registers, no distances: 67 registers, 15 KB of LDS, twelve wavefronts per CU for the part of
// the scan that mostly waits -- into blocks of a device arena, one block per round, chained:
//     [0] tiles  [1] items  [2] next block (word offset + 1, 0: none)  [3] unused
//     [4 .. 67]  candidates listed per source, [68 .. 95] unused (the rows below start on a 128-byte line, as every block does)
//     then C x 64 destination indices as u16 (row c = every source's c-th listed candidate),
//     then kDealTiles x {first record, items}, then the items in tile order.
// guided_sum_kernel walks the chain with the source descriptors in registers: stage rows, sum, pick.  A wavefront whose
// lists do not fit (more than kDealTiles tiles in a round, arena full) raises its flag, is skipped by the sum kernel and
// redone by MODE 2 = MODE 0 for flagged wavefronts only.
constexpr uint32_t kDealTiles = 64, kDealArenas = 64, kDealHead = 96;
constexpr int kSplitCap = 24;  // candidates a source lists per round in the two-kernel form (20: 1023 + 1883 us per 512 pairs, 24: 832 + 1863,
                                // 28: 748 + 1964, 32: 787 + 1985 -- the deal kernel has fewer later rounds, the sum kernel fewer wavefronts per CU)
struct DealOut {
    uint32_t* words;       // the arena: kDealArenas parts of part_words each
    uint32_t* heads;       // words taken per part
    uint32_t* first;       // per wavefront: its first block (word offset into `words` + 1; 0: none)
    uint32_t* flags;       // per wavefront: 1 = redo in one kernel
    uint32_t part_words;
};
template <int C, bool PH, int MODE>
__device__ __forceinline__ void guided_scan_flat_body(
    const GuidedPair* __restrict__ pairs, const GaEntry* __restrict__ entries, const uint32_t* __restrict__ starts,
    const double* __restrict__ spans, const uint32_t* __restrict__ order, int32_t* __restrict__ best_out, double* __restrict__ ratio_out,
    uint32_t blocks_per_pair, uint32_t n_blocks, const DealOut out) {
    static_assert(C >= 2 && C <= 32, "a candidate's place in the list is five bits of an item");
    static_assert(kGtRows == 16, "sixteen rows in flight per pass; a row number is four bits of an item");
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    constexpr int D = 1;  // steps the sums' operands travel ahead (2 and 3 measured equal)
    __shared__ uint32_t list[C][64];  // (destination index << 16 | record position), ascending record position
    __shared__ double dist[64][C | 1];  // per source (an odd stride: the lanes that sum one source's candidates write different banks)
    __shared__ uint32_t batch[64 * kGtRows / 2];  // items: lane | a << 6 | row a << 11 | b << 15 | row b << 20
    __shared__ __attribute__((aligned(16))) float tile[kGtRows][kGtStride];
    const uint32_t per_xcd = (n_blocks + 7u) / 8u;
    const uint32_t logical = (blockIdx.x % 8u) * per_xcd + blockIdx.x / 8u;
    if (logical >= n_blocks) return;
    const uint32_t pair_id = logical / blocks_per_pair, bx = logical % blocks_per_pair;
    const GuidedPair P = pairs[pair_id];
    const uint32_t lane = threadIdx.x;
    if (bx * 64u >= P.n1) return;      // (wavefront-uniform; in a live wavefront every lane stays for the shared steps)
    if (MODE == 2 && out.flags[logical] == 0u) return;
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_ph = PH ? __builtin_amdgcn_s_memtime() : 0ull;
    auto tick = [&](int k) {
        if (PH) { const unsigned long long now = __builtin_amdgcn_s_memtime(); ph[k] += now - t_ph; t_ph = now; }
    };
    const bool active = bx * 64u + lane < P.n1;
    const uint32_t i = active ? order[P.off + bx * 64u + lane] : 0u;
    // (1)
    float a[kD];
    if (MODE != 1) {
        const float* row = P.d1 + (size_t)i * kD;
#pragma unroll
        for (int q = 0; q < kD / 4; ++q) {
            const f32x4 t = load_global_f32x4(row, (size_t)q);
            a[4 * q] = t.x; a[4 * q + 1] = t.y; a[4 * q + 2] = t.z; a[4 * q + 3] = t.w;
        }
    }
    const double x1 = (double)load_global_f32(P.kp1, 2 * (size_t)i), y1 = (double)load_global_f32(P.kp1, 2 * (size_t)i + 1);
    const double rx = (P.F[0] * x1 + P.F[1] * y1) + P.F[2];
    const double ry = (P.F[3] * x1 + P.F[4] * y1) + P.F[5];
    const double b1 = rx * rx + ry * ry;
    const double vx = x1 - P.ep0, vy = y1 - P.ep1;
    // (the folded angle of the source's epipolar line is both what its bin is cut from and the centre of its gate window:
    //  atan2(vx, -vy) = atan2(vy, vx) + a quarter turn, which ga_source_centre adds back -- one arc tangent instead of two)
    const double my_angle = epipolar_angle(vx, -vy);
    const int32_t my_bin = epipolar_bin_of(my_angle, P.min_angle, P.range, P.bins);
    const uint32_t* st = starts + (size_t)pair_id * (kGaBuckets + 2);
    const double amin = spans[2 * (size_t)pair_id], amax = spans[2 * (size_t)pair_id + 1];
    // the fine buckets of this source's bin and of its gate window, as in guided_scan_tile_kernel
    int f0 = 0, f1 = kGaBuckets - 1;
    if (P.bins > 1 && P.range == P.range && P.range != 0.0) {
        const double step = P.range / (double)(P.bins - 1);
        const double ea = P.min_angle + ((double)my_bin - 0.5) * step, eb = P.min_angle + ((double)my_bin + 0.5) * step;
        const bool open_lo = my_bin == 0, open_hi = my_bin == P.bins - 1;
        double lo_ang = fmin(ea, eb), hi_ang = fmax(ea, eb);
        if ((open_lo && step > 0.0) || (open_hi && step < 0.0)) lo_ang = -1.0e300;
        if ((open_hi && step > 0.0) || (open_lo && step < 0.0)) hi_ang = 1.0e300;
        f0 = max(0, ga_fine(lo_ang, amin, amax) - 1);
        f1 = min(kGaBuckets - 1, ga_fine(hi_ang, amin, amax) + 1);
    }
    int g0[2] = {0, 0}, g1[2] = {kGaBuckets - 1, -1};
    const double d2 = vx * vx + vy * vy;
    if (d2 > 4.0 && d2 < 1.0e300 && b1 > 0.0 && b1 < 1.0e300) {
        const double c = my_angle >= 180.0 ? my_angle - 180.0 : my_angle;  // [0, 180)
        // asin(x) <= x + 0.18 x^3 on x <= 0.375 (d2 > 4): a bound is all the window needs (the exact gate decides)
        const double xs = 0.75 / sqrt(d2);
        const double half = kRadianToDegree * (xs + 0.18 * xs * xs * xs) * 1.0001 + 1.0e-5;
        if (2.0 * half < 179.0) {
            double lo[2] = {c - half, 0.0}, hi[2] = {c + half, -1.0};
            if (lo[0] <= 0.0) { lo[1] = lo[0] + 180.0; hi[1] = 180.0; lo[0] = 0.0; }
            else if (hi[0] > 180.0) { lo[1] = 0.0; hi[1] = hi[0] - 180.0; hi[0] = 180.0; }
            for (int k = 0; k < 2; ++k) {
                if (hi[k] < lo[k]) { g0[k] = 0; g1[k] = -1; continue; }
                const int fa = ga_fine(lo[k], amin, amax), fb = ga_fine(hi[k], amin, amax);
                if (fa < 0 || fb < 0) { g0[k] = 0; g1[k] = kGaBuckets - 1; continue; }
                g0[k] = max(0, fa - 1);
                g1[k] = min(kGaBuckets - 1, fb + 1);
            }
        }
    }
    int v0[2], v1[2];
    for (int k = 0; k < 2; ++k) { v0[k] = max(g0[k], f0); v1[k] = min(g1[k], f1); }
    if (v1[0] >= v0[0] && v1[1] >= v0[1] && v0[1] <= v1[0] + 1 && v0[0] <= v1[1] + 1) {
        v0[0] = min(v0[0], v0[1]);
        v1[0] = max(v1[0], v1[1]);
        v1[1] = v0[1] - 1;
    }
    uint32_t re0[3] = {0u, 0u, st[kGaBuckets]}, re1[3] = {0u, 0u, st[kGaBuckets + 1]};
    for (int k = 0; k < 2; ++k)
        if (v1[k] >= v0[k]) { re0[k] = st[v0[k]]; re1[k] = st[v1[k] + 1]; }
    if (!active) re1[0] = re0[0] = re1[1] = re0[1] = re1[2] = re0[2] = 0u;
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    tick(0);
    static_assert(sizeof(GaEntry) == 48 && kGtRecs * sizeof(GaEntry) <= sizeof(float) * kGtRows * kGtStride, "records are staged in the tile");
    GaEntry* recs = reinterpret_cast<GaEntry*>(&tile[0][0]);
    // (2) this lane's gate-passing destinations with index > after, the C smallest indices of them in `list`; returns how many in all
    auto collect = [&](int64_t after, bool idle) -> uint32_t {  // idle: a source that is done only helps with the staging
        uint32_t total = 0;
        uint32_t c0[3] = {idle ? re1[0] : re0[0], idle ? re1[1] : re0[1], idle ? re1[2] : re0[2]};  // cursors
        while (true) {
            uint32_t need = kNone;
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (c0[k] < re1[k]) need = min(need, c0[k]);
            need = wave_min_u32(need);
            if (need == kNone) break;
            const uint32_t E = need, nrec = min((uint32_t)kGtRecs, P.n2 - E);
            wave_sync();
            {
                const uint4* src = reinterpret_cast<const uint4*>(entries + P.off2 + E);
                uint4* dst = reinterpret_cast<uint4*>(recs);
                for (uint32_t q = lane; q < nrec * 3u; q += 64u) dst[q] = src[q];
            }
            wave_sync();
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const uint32_t lo = max(c0[k], E), hi = min(re1[k], E + nrec);
                // two records per step: their reads travel together and the two gates are independent chains (the reference's tests,
                // matcher.h:340-351, evaluated without branches; only the rare value between "surely near" and "surely far" divides)
                auto gate = [&](const GaEntry& en) -> bool {
                    const bool mine = (en.bin == my_bin) & ((int64_t)en.j > after);
                    const double r = (x1 * en.rxc + y1 * en.ryc) + en.rwc;
                    const double num = (r * r) * (en.a1 + b1), den = en.a1 * b1;
                    const bool far = (den > 0.0) & (num >= 0.57 * den), near = (den > 0.0) & (num <= 0.56 * den);
                    bool pass = mine & !far;
                    if (pass && !near) pass = !(num / den >= 0.75 * 0.75);  // the reference's quotient decides
                    return pass;
                };
                auto append = [&](uint32_t j, uint32_t e) {
                    const uint32_t key = (j << 16) | e;
                    uint32_t pos = min(total, (uint32_t)C);  // where the listed ones end
                    bool put = true;
                    if (total >= (uint32_t)C) {  // full: the largest listed index makes room if this one is smaller (rare)
                        uint32_t mx = 0, mp = 0;
                        for (uint32_t p2 = 0; p2 < (uint32_t)C; ++p2) {
                            const uint32_t v = list[p2][lane];
                            if (v > mx) { mx = v; mp = p2; }
                        }
                        put = key < mx;
                        if (put) {
                            for (uint32_t p2 = mp; p2 + 1u < (uint32_t)C; ++p2) list[p2][lane] = list[p2 + 1u][lane];
                            pos = (uint32_t)C - 1u;
                        }
                    }
                    if (put) {  // into its place by record position (usually the end: records are met in ascending position per range)
                        while (pos > 0u) {
                            const uint32_t prev = list[pos - 1u][lane];
                            if ((prev & 0xFFFFu) <= e) break;
                            list[pos][lane] = prev;
                            --pos;
                        }
                        list[pos][lane] = key;
                    }
                    ++total;
                };
                for (uint32_t e = lo; e < hi; e += 2u) {
                    const bool two = e + 1u < hi;
                    const GaEntry en0 = recs[e - E], en1 = recs[(two ? e + 1u : e) - E];
                    const bool p0 = gate(en0), p1 = gate(en1) & two;
                    if (p0) append(en0.j, e);
                    if (p1) append(en1.j, e + 1u);
                }
                if (hi > c0[k]) c0[k] = hi;
            }
        }
        return total;
    };
    double best = DBL_MAX, second = DBL_MAX;
    int32_t best_index = -1;
    uint32_t count = 0;
    int64_t after = -1;
    bool more = true;
    uint32_t* deal_prev = nullptr;  // MODE 1: the block of the round before
    while (__any(more)) {
        const uint32_t total = collect(after, !more);
        tick(2);
        const uint32_t nlist = min(total, (uint32_t)C);
        uint32_t jcap = kNone;  // a source with more than it lists: this round covers the indices up to its largest listed one
        if (total > (uint32_t)C) {
            jcap = 0u;
            for (uint32_t c = 0; c < (uint32_t)C; ++c) jcap = max(jcap, list[c][lane] >> 16);
        }
        uint32_t* blk = nullptr;  // MODE 1: this round's block
        uint32_t blk_tiles = 0, blk_items = 0;
        if (MODE == 1) {
            const uint32_t listed = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_sum(nlist), 63);
            const uint32_t words = (kDealHead + (uint32_t)C * 32u + 2u * kDealTiles + listed + 31u) & ~31u;  // (an item holds one or two candidates)
            const uint32_t part = logical % kDealArenas;
            uint32_t at = 0;
            if (lane == 0u) at = atomicAdd(&out.heads[part], words);
            at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
            if (at + words > out.part_words) {  // arena full: this wavefront is redone in one kernel
                if (lane == 0u) out.flags[logical] = 1u;
                return;
            }
            const uint32_t where = part * out.part_words + at;
            blk = out.words + where;
            if (lane == 0u) {
                if (deal_prev) deal_prev[2] = where + 1u; else out.first[logical] = where + 1u;
            }
            deal_prev = blk;
            blk[4u + lane] = nlist;
            unsigned short* jt = reinterpret_cast<unsigned short*>(blk + kDealHead);
            for (uint32_t c = 0; c < nlist; ++c) jt[c * 64u + lane] = (unsigned short)(list[c][lane] >> 16);
        }
        // (3) the tiles of sixteen records that hold listed candidates, ascending
        wave_sync();  // (the lists are complete; the tile is free)
        uint32_t cur = 0;  // this lane's first candidate not yet dealt
        uint32_t E0 = wave_min_u32(nlist ? (list[0][lane] & 0xFFFFu) : kNone);
        auto tile_index = [&](uint32_t E) -> uint32_t {  // destination indices of the records E .. E + 15, one per lane
            return (E != kNone && lane < min((uint32_t)kGtRows, P.n2 - E)) ? entries[P.off2 + E + lane].j : 0u;
        };
        f32x2 bv[16];  // a tile's rows travel while the tile before it is summed
        auto fetch_rows = [&](uint32_t E, uint32_t rj) {
            const uint32_t nr = min((uint32_t)kGtRows, P.n2 - E);
#pragma unroll
            for (uint32_t u = 0; u < 16u; ++u) {
                const uint32_t j = (uint32_t)__builtin_amdgcn_readlane((int)rj, (int)u);
                if (u < nr) bv[u] = load_global_f32x2(P.d2 + (size_t)j * kD, lane);
            }
        };
        // the wavefront's first listed record position at or after `from` (kNone: none).  A lane looks at the next sixteen of its
        // ordered list; one that finds them all below `from` with more behind answers `from` itself: a valid bound, never past an entry
        auto tile_after = [&](uint32_t from) -> uint32_t {
            uint32_t m = kNone;
#pragma unroll
            for (uint32_t k = 0; k < 16u; ++k) {
                const uint32_t e = list[min(cur + k, (uint32_t)C - 1u)][lane] & 0xFFFFu;
                if (cur + k < nlist && e >= from) m = min(m, e);
            }
            if (m == kNone && cur + 16u < nlist) m = from;
            return wave_min_u32(m);
        };
        // tiles E0 (dealt and summed now), E1 (its rows travel meanwhile) and E2 (its destination indices travel): each start is
        // fixed two tiles ahead, so no load waits for the one before it
        uint32_t E1 = E0 == kNone ? kNone : tile_after(E0 + (uint32_t)kGtRows);
        uint32_t rj1 = MODE != 1 ? tile_index(E1) : 0u;
        if (MODE != 1 && E0 != kNone) fetch_rows(E0, tile_index(E0));
        while (E0 != kNone) {
            const uint32_t nrow = min((uint32_t)kGtRows, P.n2 - E0);
            // the next sixteen of this lane's ordered list: those inside the tile are a run from `cur` (positions are distinct:
            // at most sixteen)
            uint32_t rel[16];
            uint32_t n_mine = 0;
#pragma unroll
            for (uint32_t k = 0; k < 16u; ++k) {
                const uint32_t e = list[min(cur + k, (uint32_t)C - 1u)][lane] & 0xFFFFu;
                rel[k] = e - E0;
                n_mine += (cur + k < nlist && e - E0 < nrow) ? 1u : 0u;
            }
            const uint32_t n_items = (n_mine + 1u) / 2u;
            const uint32_t incl = wave_inclusive_sum(n_items);
            const uint32_t B = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            {
                const uint32_t o = incl - n_items;
#pragma unroll
                for (uint32_t k = 0; k < 16u; k += 2u)
                    if (k < n_mine) {
                        const bool two = k + 1u < n_mine;
                        const uint32_t ca = cur + k, cb = two ? ca + 1u : ca, ra = rel[k], rb = two ? rel[k + 1u] : ra;
                        batch[o + k / 2u] = lane | (ca << 6) | (ra << 11) | (cb << 15) | (rb << 20);
                    }
            }
            cur += n_mine;
            const uint32_t E2 = E1 == kNone ? kNone : tile_after(E1 + (uint32_t)kGtRows);
            const uint32_t rj2 = MODE != 1 ? tile_index(E2) : 0u;
            tick(3);
            wave_sync();  // (the previous tile's sums are done)
            if (MODE == 1) {  // the tile and its items go to the block; nothing is summed here
                if (blk_tiles == kDealTiles) {  // (wavefront-uniform) more tiles than a block describes: redone in one kernel
                    if (lane == 0u) out.flags[logical] = 1u;
                    return;
                }
                uint32_t* th = blk + kDealHead + (uint32_t)C * 32u;
                if (lane == 0u) { th[2u * blk_tiles] = E0; th[2u * blk_tiles + 1u] = B; }
                uint32_t* it = th + 2u * kDealTiles + blk_items;
                for (uint32_t k = lane; k < B; k += 64u) it[k] = batch[k];
                ++blk_tiles;
                blk_items += B;
                wave_sync();  // (the items are read before the next tile's are written)
                E0 = E1; E1 = E2;
                continue;
            }
#pragma unroll
            for (uint32_t u = 0; u < 16u; ++u)
                if (u < nrow) *reinterpret_cast<f32x2*>(&tile[u][2 * lane]) = bv[u];
            if (E1 != kNone) fetch_rows(E1, rj1);
            wave_sync();
            tick(4);
            for (uint32_t base = 0; base < B; base += 64u) {
                const bool valid = base + lane < B;
                const uint32_t item = valid ? batch[base + lane] : lane;  // (a lane without an item sums row 0 against itself and drops it:
                const uint32_t owner = item & 63u;                         //  every lane stays in, its registers are read by others)
                const uint32_t ca = (item >> 6) & 31u, cb = (item >> 15) & 31u;
                const float4* ta = reinterpret_cast<const float4*>(&tile[(item >> 11) & 15u][0]);
                const float4* tb = reinterpret_cast<const float4*>(&tile[(item >> 20) & 15u][0]);
                const int from = (int)(owner << 2);
                double da = 0.0, db = 0.0;
                typedef float pk2 __attribute__((ext_vector_type(2)));
                auto pull = [&](float s) { return __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(s))); };
                // D steps ahead: the operands of elements 4 (q + D) .. travel while 4 q .. 4 q + 3 are summed.  The empty asm
                // ties the two chains to their step: without it instruction selection hoists all 64 row reads and 128 pulls above
                // the arithmetic and the descriptor is spilled (measured: 900 dwords of scratch).
                float4 U[D + 1], V[D + 1];
                float S[D + 1][4];
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    U[d] = ta[d]; V[d] = tb[d];
                    S[d][0] = pull(a[4 * d]); S[d][1] = pull(a[4 * d + 1]); S[d][2] = pull(a[4 * d + 2]); S[d][3] = pull(a[4 * d + 3]);
                }
#pragma unroll
                for (int q = 0; q < kD / 4; ++q) {
                    if (q + D < kD / 4) {
                        const int n = (q + D) % (D + 1), e = 4 * (q + D);
                        U[n] = ta[q + D]; V[n] = tb[q + D];
                        S[n][0] = pull(a[e]); S[n][1] = pull(a[e + 1]); S[n][2] = pull(a[e + 2]); S[n][3] = pull(a[e + 3]);
                    }
                    const int c = q % (D + 1);
                    const float4 u = U[c], v = V[c];
                    const pk2 a01 = {S[c][0], S[c][1]}, a23 = {S[c][2], S[c][3]};
                    const pk2 du01 = a01 - pk2{u.x, u.y}, du23 = a23 - pk2{u.z, u.w}, dv01 = a01 - pk2{v.x, v.y}, dv23 = a23 - pk2{v.z, v.w};
                    const double u0 = (double)du01.x, u1 = (double)du01.y, u2 = (double)du23.x, u3 = (double)du23.y;
                    const double w0 = (double)dv01.x, w1 = (double)dv01.y, w2 = (double)dv23.x, w3 = (double)dv23.y;
                    da = fma(u0, u0, da); db = fma(w0, w0, db);
                    da = fma(u1, u1, da); db = fma(w1, w1, db);
                    da = fma(u2, u2, da); db = fma(w2, w2, db);
                    da = fma(u3, u3, da); db = fma(w3, w3, db);
                    asm volatile("" : "+v"(da), "+v"(db));
                }
                if (valid) {
                    dist[owner][ca] = da;
                    dist[owner][cb] = db;
                }
            }
            tick(5);
            E0 = E1; E1 = E2; rj1 = rj2;
        }
        wave_sync();
        if (MODE == 1) {
            if (lane == 0u) { blk[0] = blk_tiles; blk[1] = blk_items; blk[2] = 0u; blk[3] = 0u; }
            more = more && jcap != kNone;
            if (more) after = (int64_t)jcap;
            continue;
        }
        // (4) the round's smallest (distance, index), then the smallest distance among smaller indices
        double m_d = DBL_MAX;
        uint32_t m_j = kNone;
        for (uint32_t c = 0; c < nlist; ++c) {
            const double dc = dist[lane][c];
            const uint32_t jc = list[c][lane] >> 16;
            if (dc < m_d || (dc == m_d && jc < m_j)) { m_d = dc; m_j = jc; }
        }
        double s_d = DBL_MAX;
        for (uint32_t c = 0; c < nlist; ++c)
            if ((list[c][lane] >> 16) < m_j) s_d = fmin(s_d, dist[lane][c]);
        count += nlist;
        if (m_d < best) {  // the earlier rounds held smaller indices: their best is a predecessor of this round's first minimum
            second = fmin(best, s_d);
            best = m_d;
            best_index = (int32_t)m_j;
        }
        more = more && jcap != kNone;
        if (more) after = (int64_t)jcap;
        tick(6);
    }
    if (MODE == 1) return;
    if (PH && lane == 0u)
        for (int k = 0; k < 8; ++k) atomicAdd(&g_guided_phase[k], ph[k]);
    if (!active) return;
    double corr = 1.0;
    if (count < 20u) corr = 0.65 * 0.65;
    if (count < 10u) corr = 0.6 * 0.6;
    if (count < 5u) corr = 0.5 * 0.5;
    if (count < 3u) corr = 0.25 * 0.25;
    const double ratio = (best / second) / corr;
    const bool keep = !(ratio < 0.00001) && best_index > -1 && (ratio < 0.8 * 0.8 || count == 1u);
    best_out[P.off + i] = keep ? best_index : -1;
    ratio_out[P.off + i] = ratio;
}
template <int C, bool PH>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void guided_scan_flat_kernel(
    const GuidedPair* __restrict__ pairs, const GaEntry* __restrict__ entries, const uint32_t* __restrict__ starts,
    const double* __restrict__ spans, const uint32_t* __restrict__ order, int32_t* __restrict__ best_out, double* __restrict__ ratio_out,
    uint32_t blocks_per_pair, uint32_t n_blocks) {
    guided_scan_flat_body<C, PH, 0>(pairs, entries, starts, spans, order, best_out, ratio_out, blocks_per_pair, n_blocks, DealOut{});
}
template <int C>
__global__ __launch_bounds__(64) void guided_deal_kernel(
    const GuidedPair* __restrict__ pairs, const GaEntry* __restrict__ entries, const uint32_t* __restrict__ starts,
    const double* __restrict__ spans, const uint32_t* __restrict__ order, uint32_t blocks_per_pair, uint32_t n_blocks, const DealOut out) {
    guided_scan_flat_body<C, false, 1>(pairs, entries, starts, spans, order, nullptr, nullptr, blocks_per_pair, n_blocks, out);
}
template <int C>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void guided_redo_kernel(
    const GuidedPair* __restrict__ pairs, const GaEntry* __restrict__ entries, const uint32_t* __restrict__ starts,
    const double* __restrict__ spans, const uint32_t* __restrict__ order, int32_t* __restrict__ best_out, double* __restrict__ ratio_out,
    uint32_t blocks_per_pair, uint32_t n_blocks, const DealOut out) {
    guided_scan_flat_body<C, false, 2>(pairs, entries, starts, spans, order, best_out, ratio_out, blocks_per_pair, n_blocks, out);
}
// the second kernel of the pair: source descriptors in registers, the rounds' blocks walked in order -- stage a tile's rows, sum
// its items (as in the body above), and after a round's tiles pick the round's best and the smallest distance before it
template <int C>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void guided_sum_kernel(
    const GuidedPair* __restrict__ pairs, const GaEntry* __restrict__ entries, const uint32_t* __restrict__ order,
    int32_t* __restrict__ best_out, double* __restrict__ ratio_out, uint32_t blocks_per_pair, uint32_t n_blocks, const DealOut out) {
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    __shared__ double dist[64][C | 1];
    __shared__ __attribute__((aligned(16))) float tile[kGtRows][kGtStride];
    const uint32_t per_xcd = (n_blocks + 7u) / 8u;
    const uint32_t logical = (blockIdx.x % 8u) * per_xcd + blockIdx.x / 8u;
    if (logical >= n_blocks) return;
    const uint32_t pair_id = logical / blocks_per_pair, bx = logical % blocks_per_pair;
    const GuidedPair P = pairs[pair_id];
    const uint32_t lane = threadIdx.x;
    if (bx * 64u >= P.n1) return;
    if (out.flags[logical] != 0u) return;  // (redone by guided_redo_kernel)
    const bool active = bx * 64u + lane < P.n1;
    const uint32_t i = active ? order[P.off + bx * 64u + lane] : 0u;
    float a[kD];
    {
        const float* row = P.d1 + (size_t)i * kD;
#pragma unroll
        for (int q = 0; q < kD / 4; ++q) {
            const f32x4 t = load_global_f32x4(row, (size_t)q);
            a[4 * q] = t.x; a[4 * q + 1] = t.y; a[4 * q + 2] = t.z; a[4 * q + 3] = t.w;
        }
    }
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    double best = DBL_MAX, second = DBL_MAX;
    int32_t best_index = -1;
    uint32_t count = 0;
    uint32_t at = out.first[logical];
    while (at != 0u) {
        const uint32_t* blk = out.words + (at - 1u);
        const uint32_t n_tiles = blk[0];
        at = blk[2];
        const uint32_t nlist = blk[4u + lane];
        const unsigned short* jt = reinterpret_cast<const unsigned short*>(blk + kDealHead);
        const uint32_t* th = blk + kDealHead + (uint32_t)C * 32u;
        const uint32_t* items = th + 2u * kDealTiles;
        // this source's listed destination indices, fetched now and used after the tiles (the loads travel behind the sums)
        uint32_t jv[C];
#pragma unroll
        for (int c = 0; c < C; ++c) jv[c] = (uint32_t)c < nlist ? (uint32_t)jt[c * 64 + (int)lane] : 0u;
        // lane t holds tile t: its first record and its item count (at most kDealTiles = 64 tiles per block)
        uint32_t tE = kNone, tB = 0u;
        if (lane < n_tiles) { tE = th[2u * lane]; tB = th[2u * lane + 1u]; }
        auto tile_index = [&](uint32_t E) -> uint32_t {
            return (E != kNone && lane < min((uint32_t)kGtRows, P.n2 - E)) ? entries[P.off2 + E + lane].j : 0u;
        };
        f32x2 bv[16];
        auto fetch_rows = [&](uint32_t E, uint32_t rj) {
            const uint32_t nr = min((uint32_t)kGtRows, P.n2 - E);
#pragma unroll
            for (uint32_t u = 0; u < 16u; ++u) {
                const uint32_t j = (uint32_t)__builtin_amdgcn_readlane((int)rj, (int)u);
                if (u < nr) bv[u] = load_global_f32x2(P.d2 + (size_t)j * kD, lane);
            }
        };
        auto first_of = [&](uint32_t t) { return t < n_tiles ? (uint32_t)__builtin_amdgcn_readlane((int)tE, (int)t) : kNone; };
        wave_sync();  // (the round before is done with dist and the tile)
        uint32_t E0 = first_of(0u), E1 = first_of(1u);
        uint32_t rj1 = tile_index(E1);
        if (E0 != kNone) fetch_rows(E0, tile_index(E0));
        uint32_t item_at = 0;
        const uint32_t n_items_round = blk[1];
        uint32_t item_ahead = lane < n_items_round ? items[lane] : lane;  // the items are one stream: a batch's words are fetched a batch ahead
        for (uint32_t t = 0; t < n_tiles; ++t) {
            const uint32_t nrow = min((uint32_t)kGtRows, P.n2 - E0);
            const uint32_t B = (uint32_t)__builtin_amdgcn_readlane((int)tB, (int)t);
            const uint32_t E2 = first_of(t + 2u);
            const uint32_t rj2 = tile_index(E2);
            wave_sync();  // (the previous tile's sums are done)
#pragma unroll
            for (uint32_t u = 0; u < 16u; ++u)
                if (u < nrow) *reinterpret_cast<f32x2*>(&tile[u][2 * lane]) = bv[u];
            if (E1 != kNone) fetch_rows(E1, rj1);
            wave_sync();
            for (uint32_t base = 0; base < B; base += 64u) {
                const bool valid = base + lane < B;
                const uint32_t item = valid ? item_ahead : lane;
                {   // the next batch: the rest of this tile's items, or the start of the next tile's
                    const uint32_t nx = item_at + (base + 64u < B ? base + 64u : B) + lane;
                    item_ahead = nx < n_items_round ? items[nx] : lane;
                }
                const uint32_t owner = item & 63u;
                const uint32_t ca = (item >> 6) & 31u, cb = (item >> 15) & 31u;
                const float4* ta = reinterpret_cast<const float4*>(&tile[(item >> 11) & 15u][0]);
                const float4* tb = reinterpret_cast<const float4*>(&tile[(item >> 20) & 15u][0]);
                const int from = (int)(owner << 2);
                double da = 0.0, db = 0.0;
                typedef float pk2 __attribute__((ext_vector_type(2)));
                auto pull = [&](float s) { return __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(s))); };
                float4 u = ta[0], v = tb[0];
                float s0 = pull(a[0]), s1 = pull(a[1]), s2 = pull(a[2]), s3 = pull(a[3]);
#pragma unroll
                for (int q = 0; q < kD / 4; ++q) {
                    float4 un = u, vn = v;
                    float n0 = s0, n1 = s1, n2 = s2, n3 = s3;
                    if (q + 1 < kD / 4) {
                        un = ta[q + 1]; vn = tb[q + 1];
                        n0 = pull(a[4 * q + 4]); n1 = pull(a[4 * q + 5]); n2 = pull(a[4 * q + 6]); n3 = pull(a[4 * q + 7]);
                    }
                    const pk2 a01 = {s0, s1}, a23 = {s2, s3};
                    const pk2 du01 = a01 - pk2{u.x, u.y}, du23 = a23 - pk2{u.z, u.w}, dv01 = a01 - pk2{v.x, v.y}, dv23 = a23 - pk2{v.z, v.w};
                    const double u0 = (double)du01.x, u1 = (double)du01.y, u2 = (double)du23.x, u3 = (double)du23.y;
                    const double w0 = (double)dv01.x, w1 = (double)dv01.y, w2 = (double)dv23.x, w3 = (double)dv23.y;
                    da = fma(u0, u0, da); db = fma(w0, w0, db);
                    da = fma(u1, u1, da); db = fma(w1, w1, db);
                    da = fma(u2, u2, da); db = fma(w2, w2, db);
                    da = fma(u3, u3, da); db = fma(w3, w3, db);
                    u = un; v = vn; s0 = n0; s1 = n1; s2 = n2; s3 = n3;
                    asm volatile("" : "+v"(da), "+v"(db));  // (ties the chains to their step: see the body above)
                }
                if (valid) {
                    dist[owner][ca] = da;
                    dist[owner][cb] = db;
                }
            }
            item_at += B;
            E0 = E1; E1 = E2; rj1 = rj2;
        }
        wave_sync();
        double m_d = DBL_MAX;
        uint32_t m_j = kNone;
#pragma unroll
        for (int c = 0; c < C; ++c)
            if ((uint32_t)c < nlist) {
                const double dc = dist[lane][c];
                if (dc < m_d || (dc == m_d && jv[c] < m_j)) { m_d = dc; m_j = jv[c]; }
            }
        double s_d = DBL_MAX;
#pragma unroll
        for (int c = 0; c < C; ++c)
            if ((uint32_t)c < nlist && jv[c] < m_j) s_d = fmin(s_d, dist[lane][c]);
        count += nlist;
        if (m_d < best) {
            second = fmin(best, s_d);
            best = m_d;
            best_index = (int32_t)m_j;
        }
    }
    if (!active) return;
    double corr = 1.0;
    if (count < 20u) corr = 0.65 * 0.65;
    if (count < 10u) corr = 0.6 * 0.6;
    if (count < 5u) corr = 0.5 * 0.5;
    if (count < 3u) corr = 0.25 * 0.25;
    const double ratio = (best / second) / corr;
    const bool keep = !(ratio < 0.00001) && best_index > -1 && (ratio < 0.8 * 0.8 || count == 1u);
    best_out[P.off + i] = keep ? best_index : -1;
    ratio_out[P.off + i] = ratio;
}

__global__ __launch_bounds__(1024) void guided_select_kernel(const GuidedPair* __restrict__ pairs, const int32_t* __restrict__ best_in,
                                                             const double* __restrict__ ratio_in, uint32_t* __restrict__ ci,
                                                             uint32_t* __restrict__ cj, double* __restrict__ cr, uint32_t max_n,
                                                             uint32_t out_stride, uint32_t* __restrict__ out_src,
                                                             uint32_t* __restrict__ out_dst, double* __restrict__ out_ratio,
                                                             uint32_t* __restrict__ out_count, uint32_t* __restrict__ kept_out) {
    __shared__ uint32_t wtot[16];
    __shared__ uint32_t carry;
    const GuidedPair P = pairs[blockIdx.x];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    if (tid == 0) carry = 0u;
    __syncthreads();
    for (uint32_t base = 0; base < P.n1; base += 1024u) {  // ordered compaction of the kept matches (ballots: three barriers a round)
        const uint32_t i = base + tid;
        const int32_t bj = i < P.n1 ? best_in[P.off + i] : -1;
        const bool v = bj >= 0;
        const uint64_t mk = __ballot(v);
        if (lane == 0) wtot[wv] = (uint32_t)__popcll(mk);
        __syncthreads();
        uint32_t before = carry, all = 0;
        for (uint32_t q = 0; q < 16u; ++q) {
            if (q < wv) before += wtot[q];
            all += wtot[q];
        }
        if (v) {
            const uint32_t pos = before + (uint32_t)__popcll(mk & ((1ull << lane) - 1ull));
            ci[P.off + pos] = i; cj[P.off + pos] = (uint32_t)bj; cr[P.off + pos] = ratio_in[P.off + i];
        }
        __syncthreads();
        if (tid == 0) carry += all;
        __syncthreads();
    }
    const uint32_t m = carry;
    __threadfence_block();
    __syncthreads();
    const size_t o = (size_t)blockIdx.x * out_stride;
    if (tid == 0) kept_out[blockIdx.x] = m;
    if (max_n == 0u || m <= max_n) {
        const uint32_t lim = m < out_stride ? m : out_stride;
        for (uint32_t k = tid; k < lim; k += 1024u) { out_src[o + k] = ci[P.off + k]; out_dst[o + k] = cj[P.off + k]; out_ratio[o + k] = cr[P.off + k]; }
        if (tid == 0) out_count[blockIdx.x] = lim;
    } else if (tid == 0) {
        out_count[blockIdx.x] = max_n < out_stride ? max_n : out_stride;  // guided_rank_kernel fills the rows
    }
}

// More kept matches than the cut allows: the `lim` smallest adapted ratios (ties by position), in that order
// (pose_graph_builder.h:759-772).  One workgroup per image pair: a radix select over the ratios' bit patterns (they are
// positive doubles or +0 / +inf / NaN-free by construction of `keep`, so integer order is value order) finds the lim-th
// smallest key T in eight histogram passes; the keys below T and the first ties at T (in position order) are the answer,
// gathered into LDS and ranked among themselves.  O(m) instead of the O(m^2) rank-by-counting over all kept matches.
constexpr uint32_t kTopMax = 1024;  // the largest cut handled here (the reference cuts at 100)
__global__ __launch_bounds__(1024) void guided_topn_kernel(const GuidedPair* __restrict__ pairs, const uint32_t* __restrict__ kept,
                                                           const uint32_t* __restrict__ ci, const uint32_t* __restrict__ cj,
                                                           const double* __restrict__ cr, uint32_t max_n, uint32_t out_stride,
                                                           uint32_t* __restrict__ out_src, uint32_t* __restrict__ out_dst,
                                                           double* __restrict__ out_ratio) {
    __shared__ uint32_t hist[256];
    __shared__ unsigned long long sel_key[kTopMax];
    __shared__ uint32_t sel_pos[kTopMax];
    __shared__ uint32_t wtot[16];
    __shared__ unsigned long long s_prefix;
    __shared__ uint32_t s_need, s_count, s_carry;
    const GuidedPair P = pairs[blockIdx.x];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6, m = kept[blockIdx.x];
    if (max_n == 0u || m <= max_n) return;
    const size_t o = (size_t)blockIdx.x * out_stride;
    const uint32_t lim = max_n < out_stride ? max_n : out_stride;  // (<= kTopMax: checked by the host)
    const unsigned long long* keys = reinterpret_cast<const unsigned long long*>(cr + P.off);
    // the lim-th smallest key, a byte at a time from the top
    if (tid == 0) { s_prefix = 0ull; s_need = lim; }
    for (int shift = 56; shift >= 0; shift -= 8) {
        if (tid < 256u) hist[tid] = 0u;
        __syncthreads();
        const unsigned long long prefix = s_prefix, high = shift == 56 ? 0ull : ~0ull << (shift + 8);
        for (uint32_t k = tid; k < m; k += 1024u) {
            const unsigned long long key = keys[k];
            if ((key & high) == prefix) atomicAdd(&hist[(uint32_t)(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t need = s_need, d = 0;
            for (; d < 255u && hist[d] < need; ++d) need -= hist[d];  // the digit in which the count crosses `need`
            s_need = need;
            s_prefix = prefix | ((unsigned long long)d << shift);
        }
        __syncthreads();
    }
    const unsigned long long T = s_prefix;
    const uint32_t ties_taken = s_need;  // of the keys equal to T, the first `ties_taken` in position order belong to the answer
    if (tid == 0) { s_count = 0u; s_carry = 0u; }
    __syncthreads();
    for (uint32_t base = 0; base < m; base += 1024u) {  // ordered count of the ties (ballots), unordered gather of the answer
        const uint32_t k = base + tid;
        const unsigned long long key = k < m ? keys[k] : ~0ull;
        const bool tie = k < m && key == T;
        const uint64_t mk = __ballot(tie);
        if (lane == 0) wtot[wv] = (uint32_t)__popcll(mk);
        __syncthreads();
        uint32_t before = s_carry, all = 0;
        for (uint32_t q = 0; q < 16u; ++q) {
            if (q < wv) before += wtot[q];
            all += wtot[q];
        }
        const uint32_t tie_rank = before + (uint32_t)__popcll(mk & ((1ull << lane) - 1ull));
        if (k < m && (key < T || (tie && tie_rank < ties_taken))) {
            const uint32_t at = atomicAdd(&s_count, 1u);
            sel_key[at] = key;
            sel_pos[at] = k;
        }
        __syncthreads();
        if (tid == 0) s_carry += all;
        __syncthreads();
    }
    // exactly lim entries; rank them among themselves (ratio ascending, ties by position)
    for (uint32_t e = tid; e < lim; e += 1024u) {
        const unsigned long long key = sel_key[e];
        const uint32_t pos = sel_pos[e];
        uint32_t rank = 0;
        for (uint32_t q = 0; q < lim; ++q) rank += (sel_key[q] < key || (sel_key[q] == key && sel_pos[q] < pos)) ? 1u : 0u;
        out_src[o + rank] = ci[P.off + pos];
        out_dst[o + rank] = cj[P.off + pos];
        out_ratio[o + rank] = cr[P.off + pos];
    }
}

// the same by counting over all kept matches (cuts above kTopMax)
constexpr uint32_t kRankSplit = 4;
__global__ __launch_bounds__(1024) void guided_rank_kernel(const GuidedPair* __restrict__ pairs, const uint32_t* __restrict__ kept,
                                                           const uint32_t* __restrict__ ci, const uint32_t* __restrict__ cj,
                                                           const double* __restrict__ cr, uint32_t max_n, uint32_t out_stride,
                                                           uint32_t* __restrict__ out_src, uint32_t* __restrict__ out_dst,
                                                           double* __restrict__ out_ratio) {
    __shared__ double tile[2048];  // ratios staged 2048 at a time
    const GuidedPair P = pairs[blockIdx.y];
    const uint32_t tid = threadIdx.x, m = kept[blockIdx.y];
    if (max_n == 0u || m <= max_n) return;
    const size_t o = (size_t)blockIdx.y * out_stride;
    const uint32_t lim = max_n < out_stride ? max_n : out_stride;
    for (uint32_t k0 = blockIdx.x * 1024u; k0 < m; k0 += gridDim.x * 1024u) {  // uniform trip count: the tile loads need every thread
        const uint32_t k = k0 + tid;
        const double rk = k < m ? cr[P.off + k] : 0.0;
        uint32_t rank = 0;
        for (uint32_t q0 = 0; q0 < m; q0 += 2048u) {
            __syncthreads();
            for (uint32_t q = tid; q < 2048u && q0 + q < m; q += 1024u) tile[q] = cr[P.off + q0 + q];
            __syncthreads();
            const uint32_t qn = m - q0 < 2048u ? m - q0 : 2048u;
            uint32_t q = 0;
            for (; q + 8 <= qn; q += 8) {  // eight independent LDS reads in flight (one at a time is a latency chain)
                double rq[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) rq[u] = tile[q + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) rank += (rq[u] < rk || (rq[u] == rk && q0 + q + (uint32_t)u < k)) ? 1u : 0u;
            }
            for (; q < qn; ++q) {
                const double rq = tile[q];
                rank += (rq < rk || (rq == rk && q0 + q < k)) ? 1u : 0u;
            }
        }
        if (k < m && rank < lim) { out_src[o + rank] = ci[P.off + k]; out_dst[o + rank] = cj[P.off + k]; out_ratio[o + rank] = rk; }
    }
}

struct KpPair {
    const float *src, *dst;
    uint32_t n_src, n_dst;
    double sfx, sfy, scx, scy, dfx, dfy, dcx, dcy;
};

// exclusive scan of min(count, top_k) over the pairs: one workgroup, chunked
__global__ __launch_bounds__(1024) void corr_offsets_kernel(const uint32_t* __restrict__ counts, uint32_t n_pairs, uint32_t top_k,
                                                            unsigned long long* __restrict__ offsets) {
    __shared__ unsigned long long part[1024];
    __shared__ unsigned long long carry;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) carry = 0ull;
    __syncthreads();
    for (uint32_t base = 0; base < n_pairs; base += 1024u) {
        const uint32_t p = base + tid;
        const unsigned long long v = p < n_pairs ? (unsigned long long)(counts[p] < top_k ? counts[p] : top_k) : 0ull;
        part[tid] = v;
        __syncthreads();
        for (uint32_t d = 1; d < 1024u; d <<= 1) {
            const unsigned long long add = tid >= d ? part[tid - d] : 0ull;
            __syncthreads();
            part[tid] += add;
            __syncthreads();
        }
        if (p < n_pairs) offsets[p] = carry + part[tid] - v;
        __syncthreads();
        if (tid == 1023u) carry += part[1023];
        __syncthreads();
    }
    if (tid == 0) offsets[n_pairs] = carry;
}

__global__ __launch_bounds__(256) void corr_gather_kernel(const KpPair* __restrict__ pairs, const uint32_t* __restrict__ msrc,
                                                          const uint32_t* __restrict__ mdst, uint32_t max_matches,
                                                          const unsigned long long* __restrict__ offsets, double thr_px,
                                                          float* __restrict__ x1, float* __restrict__ y1, float* __restrict__ x2,
                                                          float* __restrict__ y2, double* __restrict__ thr) {
    const KpPair P = pairs[blockIdx.x];
    const unsigned long long o = offsets[blockIdx.x];
    const uint32_t m = (uint32_t)(offsets[blockIdx.x + 1] - o);
    for (uint32_t k = threadIdx.x; k < m; k += 256u) {
        const uint32_t i = msrc[(size_t)blockIdx.x * max_matches + k], j = mdst[(size_t)blockIdx.x * max_matches + k];
        const float2 a = i < P.n_src ? *reinterpret_cast<const float2*>(P.src + 2 * (size_t)i) : make_float2(NAN, NAN);
        const float2 b = j < P.n_dst ? *reinterpret_cast<const float2*>(P.dst + 2 * (size_t)j) : make_float2(NAN, NAN);
        x1[o + k] = (float)(((double)a.x - P.scx) / P.sfx);
        y1[o + k] = (float)(((double)a.y - P.scy) / P.sfy);
        x2[o + k] = (float)(((double)b.x - P.dcx) / P.dfx);
        y2[o + k] = (float)(((double)b.y - P.dcy) / P.dfy);
    }
    if (threadIdx.x == 0) thr[blockIdx.x] = thr_px / ((P.sfx + P.sfy + P.dfx + P.dfy) / 4.0);
}
}  // namespace

extern "C" {
uint32_t pgi_desc_padded(uint32_t n) { return (n + (uint32_t)kPadRows - 1u) / (uint32_t)kPadRows * (uint32_t)kPadRows; }

int pgi_desc_prepare(pgi_ctx* ctx, const float* d_desc, uint32_t n, float* d_desc_t, float* d_norm) {
    if (!ctx || !d_desc_t || !d_norm || (n && !d_desc)) return pgi::fail(PGI_ERR_INVALID, "pgi_desc_prepare: null argument");
    if (n > PGI_DESC_MAX) return pgi::fail(PGI_ERR_TOO_LARGE, "pgi_desc_prepare: more than PGI_DESC_MAX keypoints");
    const uint32_t n_pad = pgi_desc_padded(n);
    if (n_pad == 0) return PGI_SUCCESS;
    std::lock_guard<std::mutex> lock(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(desc_prepare_kernel, dim3(n_pad / 64), dim3(256), 0, ctx->stream, d_desc, n, n_pad, d_desc_t, d_norm);
    HIP_TRY(hipGetLastError());
    return PGI_SUCCESS;
}

int pgi_desc_prepare_screen(pgi_ctx* ctx, const float* d_desc, uint32_t n, float* d_desc_rm, uint16_t* d_desc_f16) {
    if (!ctx || !d_desc_rm || !d_desc_f16 || (n && !d_desc)) return pgi::fail(PGI_ERR_INVALID, "pgi_desc_prepare_screen: null argument");
    if (n > PGI_DESC_MAX) return pgi::fail(PGI_ERR_TOO_LARGE, "pgi_desc_prepare_screen: more than PGI_DESC_MAX keypoints");
    const uint32_t n_pad = pgi_desc_padded(n);
    if (n_pad == 0) return PGI_SUCCESS;
    std::lock_guard<std::mutex> lock(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t elems = (size_t)n_pad * kD;
    hipLaunchKernelGGL(desc_round_f16_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, ctx->stream, d_desc, n, n_pad, d_desc_rm,
                       d_desc_f16);
    HIP_TRY(hipGetLastError());
    return PGI_SUCCESS;
}

// screened path of pgi_match_descriptors_batch (views validated by the caller, ctx locked)
static int match_screened(pgi_ctx* ctx, const pgi_desc_view* h_src, const pgi_desc_view* h_dst, uint32_t n_pairs, uint32_t max_matches,
                          uint32_t* d_match_src, uint32_t* d_match_dst, double* d_ratio, uint32_t* d_counts) {
    constexpr uint32_t NWS = PGI_SCREEN_WAVES, kRowsWg = NWS * 32u;
    // the three pair tables are built in a page-locked block that mirrors their place in the workspace: one asynchronous
    // copy, no wait (three copies out of local vectors plus the synchronisation they need cost 0.1 ms per call)
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t o_hp = 0, o_fwd = o_hp + up(n_pairs * sizeof(MatchPair)), o_bwd = o_fwd + up(n_pairs * sizeof(ScreenPair)),
                 o_sf = o_bwd + up(n_pairs * sizeof(ScreenPair));
    // (two blocks in turn: the copy of a call is queued behind the previous call's kernels, so waiting for the block used
    // last time would make the host wait for the GPU on every call)
    const int sb_i = ctx->match_stage_next;
    ctx->match_stage_next ^= 1;
    if (ctx->match_stage_ev[sb_i]) HIP_TRY(hipEventSynchronize(ctx->match_stage_ev[sb_i]));  // the copy out of this block, two calls ago
    if (o_sf > ctx->match_stage_bytes[sb_i]) {
        if (ctx->h_match_stage[sb_i]) (void)hipHostFree(ctx->h_match_stage[sb_i]);
        ctx->h_match_stage[sb_i] = nullptr; ctx->match_stage_bytes[sb_i] = 0;
        HIP_TRY(hipHostMalloc(&ctx->h_match_stage[sb_i], o_sf + o_sf / 2, hipHostMallocDefault));
        ctx->match_stage_bytes[sb_i] = o_sf + o_sf / 2;
    }
    if (!ctx->match_stage_ev[sb_i]) HIP_TRY(hipEventCreateWithFlags(&ctx->match_stage_ev[sb_i], hipEventDisableTiming));
    char* hst = (char*)ctx->h_match_stage[sb_i];
    MatchPair* hp = reinterpret_cast<MatchPair*>(hst + o_hp);
    ScreenPair *fwd = reinterpret_cast<ScreenPair*>(hst + o_fwd), *bwd = reinterpret_cast<ScreenPair*>(hst + o_bwd);
    uint64_t rows_total = 0, cols_total = 0, rb_f = 0, rb_b = 0;
    uint32_t max_rb_f = 0, max_rb_b = 0, min_tiles_f = ~0u, min_tiles_b = ~0u, max_na = 0, max_nb = 0;
    for (uint32_t p = 0; p < n_pairs; ++p) {
        const pgi_desc_view &a = h_src[p], &b = h_dst[p];
        hp[p] = MatchPair{a.d_desc_t, a.d_norm, b.d_desc_t, b.d_norm, a.n, a.n_pad, b.n, b.n_pad, rows_total, cols_total};
        fwd[p] = ScreenPair{a.d_desc_f16, b.d_desc_f16, a.d_desc_rm, b.d_desc_rm, a.d_desc_t, b.d_desc_t, a.d_norm, b.d_norm,
                            a.n, a.n_pad, b.n, b.n_pad, rows_total, nullptr, nullptr};
        bwd[p] = ScreenPair{b.d_desc_f16, a.d_desc_f16, b.d_desc_rm, a.d_desc_rm, b.d_desc_t, a.d_desc_t, b.d_norm, a.d_norm,
                            b.n, b.n_pad, a.n, a.n_pad, cols_total, nullptr, nullptr};  // (row list: filled in below)
        if (a.n && b.n) {
            rb_f += a.n_pad / kRowsWg; rb_b += b.n_pad / kRowsWg;
            max_rb_f = std::max(max_rb_f, a.n_pad / kRowsWg); max_rb_b = std::max(max_rb_b, b.n_pad / kRowsWg);
            min_tiles_f = std::min(min_tiles_f, b.n_pad / (uint32_t)kTileJ); min_tiles_b = std::min(min_tiles_b, a.n_pad / (uint32_t)kTileJ);
        }
        max_na = std::max(max_na, a.n); max_nb = std::max(max_nb, b.n);
        rows_total += a.n_pad; cols_total += b.n_pad;
    }
    auto pick_splits = [](uint64_t row_blocks, uint32_t min_tiles) {
        uint32_t s = 1;
        if (row_blocks > 0 && row_blocks < 1024) s = (uint32_t)((1024 + row_blocks - 1) / row_blocks);
        if (s > 8) s = 8;
        if (min_tiles != ~0u && s > min_tiles) s = min_tiles;
        return s < 1 ? 1u : s;
    };
    // The column-wise direction only looks at the columns the mutual test will ask about (needed_columns_kernel).  Its
    // splits are chosen as for the full problem: splitting further to make up for the missing row blocks cost more in
    // candidate lists than it gained in balance (2.82 ms against 2.56 ms).  PGI_MATCH_RESTRICT=0: every column (experiments).
    static const bool restrict_cols = [] { const char* e = getenv("PGI_MATCH_RESTRICT"); return !(e && e[0] == '0'); }();
    const uint32_t sf = pick_splits(rb_f, min_tiles_f), sb = pick_splits(rb_b, min_tiles_b);
    const size_t o_sb = o_sf + up((size_t)rows_total * sf * sizeof(ScreenRow)),
                 o_rows = o_sb + up((size_t)cols_total * sb * sizeof(ScreenRow)), o_keys = o_rows + up((size_t)rows_total * sizeof(RowBest)),
                 o_fb = o_keys + up((size_t)cols_total * 8), o_cnt = o_fb + up((size_t)(rows_total + cols_total) * sizeof(uint32_t)),
                 o_list = o_cnt + up((size_t)2 * n_pairs * sizeof(uint32_t)), o_lcnt = o_list + up((size_t)cols_total * sizeof(uint32_t)),
                 bytes = o_lcnt + up((size_t)n_pairs * sizeof(uint32_t)) + 256;
    if (bytes > ctx->match_ws_bytes) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->d_match_ws) (void)hipFree(ctx->d_match_ws);
        ctx->d_match_ws = nullptr;
        ctx->match_ws_bytes = 0;
        ctx->d_match_cnt = nullptr;  // diagnostics pointer into the old workspace
        ctx->match_cnt_pairs = 0;
        HIP_TRY(hipMalloc(&ctx->d_match_ws, bytes));
        ctx->match_ws_bytes = bytes;
    }
    char* ws = (char*)ctx->d_match_ws;
    MatchPair* d_hp = (MatchPair*)(ws + o_hp);
    ScreenPair *d_fwd = (ScreenPair*)(ws + o_fwd), *d_bwd = (ScreenPair*)(ws + o_bwd);
    ScreenRow *d_sf = (ScreenRow*)(ws + o_sf), *d_sb = (ScreenRow*)(ws + o_sb);
    RowBest* d_rows = (RowBest*)(ws + o_rows);
    unsigned long long* d_keys = (unsigned long long*)(ws + o_keys);
    uint32_t* d_fb = (uint32_t*)(ws + o_fb);
    uint32_t* d_cnt = (uint32_t*)(ws + o_cnt);
    ctx->d_match_cnt = d_cnt;
    ctx->match_cnt_pairs = n_pairs;
    uint32_t* d_list = (uint32_t*)(ws + o_list);
    uint32_t* d_lcnt = (uint32_t*)(ws + o_lcnt);
    if (restrict_cols)
        for (uint32_t p = 0; p < n_pairs; ++p) { bwd[p].row_list = d_list; bwd[p].row_count = d_lcnt + p; }
    HIP_TRY(hipMemcpyAsync(ws + o_hp, hst, o_sf, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipEventRecord(ctx->match_stage_ev[sb_i], ctx->stream));
    HIP_TRY(hipMemsetAsync(d_cnt, 0, (size_t)2 * n_pairs * sizeof(uint32_t), ctx->stream));
    if (cols_total) HIP_TRY(hipMemsetAsync(d_keys, 0xFF, (size_t)cols_total * 8, ctx->stream));  // columns of empty pairs stay "no best"
    // (Running the forward direction's verification and re-scan on a second stream underneath the backward screen was tried:
    // the screen slowed down by as much as the small kernels took, 3.57 ms against 3.66 ms -- they compete for the same L2.)
    auto direction = [&](const ScreenPair* d_pairs, ScreenRow* d_scr, uint32_t splits, uint32_t max_rb, uint64_t stride, uint32_t max_rows,
                         uint32_t as_keys, uint32_t* fb, uint32_t* cnt) {
        if (max_rb == 0) return;
        const uint32_t per_pair = max_rb * splits;
        if (as_keys)  // column-wise pass: only the nearest row of each column is needed
            hipLaunchKernelGGL((desc_screen_kernel<NWS, 1>), dim3(per_pair * n_pairs), dim3(NWS * 64), 0, ctx->stream, d_pairs, d_scr, splits,
                               per_pair, stride);
        else
            hipLaunchKernelGGL((desc_screen_kernel<NWS, 2>), dim3(per_pair * n_pairs), dim3(NWS * 64), 0, ctx->stream, d_pairs, d_scr, splits,
                               per_pair, stride);
        hipLaunchKernelGGL(desc_verify_kernel, dim3((max_rows + 255) / 256, n_pairs), dim3(256), 0, ctx->stream, d_pairs, d_scr, splits, stride,
                           max_rows, d_rows, d_keys, as_keys, fb, cnt);
    };
    // exact re-scan of the flagged rows: y in [0, n_pairs) is the forward problem, [n_pairs, 2 n_pairs) the swapped one
    auto rescan = [&](uint32_t y_base, uint32_t ny) {
        hipLaunchKernelGGL(desc_exact_rows_kernel, dim3(n_pairs < 16 ? 64 : 16, ny), dim3(kExactThreads), 0, ctx->stream, d_fwd, d_bwd, n_pairs, d_fb,
                           d_fb + rows_total, d_cnt, d_rows, d_keys, y_base);
    };
    direction(d_fwd, d_sf, sf, max_rb_f, rows_total, max_na, 0u, d_fb, d_cnt);
    if (restrict_cols && max_rb_f && max_rb_b) {
        rescan(0u, n_pairs);  // the rows' results are final: which columns will the mutual test read?
        hipLaunchKernelGGL(needed_columns_kernel, dim3(n_pairs), dim3(1024), (size_t)pgi_desc_padded(max_nb), ctx->stream, d_hp, d_rows, d_list, d_lcnt);
        direction(d_bwd, d_sb, sb, max_rb_b, cols_total, max_nb, 1u, d_fb + rows_total, d_cnt + n_pairs);
        rescan(n_pairs, n_pairs);
    } else {
        direction(d_bwd, d_sb, sb, max_rb_b, cols_total, max_nb, 1u, d_fb + rows_total, d_cnt + n_pairs);
        if (max_rb_f || max_rb_b) rescan(0u, 2 * n_pairs);  // both directions in one launch
    }
    HIP_TRY(hipGetLastError());
    uint32_t np = 2;
    while (np < max_na) np <<= 1;
    HIP_TRY(hipFuncSetAttribute((const void*)match_select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    hipLaunchKernelGGL(match_select_kernel, dim3(n_pairs), dim3(1024), (size_t)np * sizeof(unsigned long long), ctx->stream, d_hp, d_rows,
                       d_keys, 1u, max_matches, d_match_src, d_match_dst, d_ratio, d_counts);
    HIP_TRY(hipGetLastError());
    return PGI_SUCCESS;
}

int pgi_match_descriptors_batch(pgi_ctx* ctx, const pgi_desc_view* h_src, const pgi_desc_view* h_dst, uint32_t n_pairs,
                                uint32_t max_matches, uint32_t* d_match_src, uint32_t* d_match_dst, double* d_ratio,
                                uint32_t* d_counts) {
    if (!ctx || !d_counts) return pgi::fail(PGI_ERR_INVALID, "pgi_match_descriptors_batch: null argument");
    if (n_pairs == 0) return PGI_SUCCESS;
    if (!h_src || !h_dst || !d_match_src || !d_match_dst || !d_ratio || max_matches == 0)
        return pgi::fail(PGI_ERR_INVALID, "pgi_match_descriptors_batch: null argument");
    std::lock_guard<std::mutex> lock(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    std::vector<MatchPair> hp(n_pairs);
    const uint32_t nw = ctx->match_waves == 8 ? 8u : 4u, rows_wg = nw * 32u;
    uint64_t row_blocks = 0, rows_total = 0, cols_total = 0;
    uint32_t max_rb = 0, min_tiles = ~0u, max_na = 0;
    for (uint32_t p = 0; p < n_pairs; ++p) {
        const pgi_desc_view &a = h_src[p], &b = h_dst[p];
        if (a.n_pad != pgi_desc_padded(a.n) || b.n_pad != pgi_desc_padded(b.n) || (a.n && (!a.d_desc_t || !a.d_norm)) ||
            (b.n && (!b.d_desc_t || !b.d_norm)))
            return pgi::fail(PGI_ERR_INVALID, "pgi_match_descriptors_batch: bad descriptor view");
        if (a.n > PGI_DESC_MAX || b.n > PGI_DESC_MAX)
            return pgi::fail(PGI_ERR_TOO_LARGE, "pgi_match_descriptors_batch: more than PGI_DESC_MAX keypoints");
        hp[p] = MatchPair{a.d_desc_t, a.d_norm, b.d_desc_t, b.d_norm, a.n, a.n_pad, b.n, b.n_pad, 0, cols_total};
        const uint32_t rb = a.n_pad / rows_wg, tiles = b.n_pad / kTileJ;
        row_blocks += (b.n ? rb : 0);
        max_rb = rb > max_rb ? rb : max_rb;
        if (a.n && b.n) min_tiles = tiles < min_tiles ? tiles : min_tiles;
        max_na = a.n > max_na ? a.n : max_na;
        rows_total += a.n_pad;
        cols_total += b.n_pad;
    }
    bool screen = ctx->match_screen != 0;
    for (uint32_t p = 0; p < n_pairs && screen; ++p)
        screen = (!h_src[p].n || (h_src[p].d_desc_rm && h_src[p].d_desc_f16)) && (!h_dst[p].n || (h_dst[p].d_desc_rm && h_dst[p].d_desc_f16));
    if (screen) return match_screened(ctx, h_src, h_dst, n_pairs, max_matches, d_match_src, d_match_dst, d_ratio, d_counts);
    // column splits: enough workgroups to cover the 256 CUs a few times over, never more splits than tiles
    uint32_t splits = 1;
    if (row_blocks > 0 && row_blocks < 1024) splits = (uint32_t)((1024 + row_blocks - 1) / row_blocks);
    if (splits > 8) splits = 8;
    if (min_tiles != ~0u && splits > min_tiles) splits = min_tiles;
    if (splits < 1) splits = 1;
    uint64_t acc = 0;
    for (uint32_t p = 0; p < n_pairs; ++p) { hp[p].row_off = acc; acc += (uint64_t)splits * hp[p].n_a_pad; }
    const size_t pair_bytes = ((size_t)n_pairs * sizeof(MatchPair) + 255) / 256 * 256;
    const size_t row_bytes = ((size_t)rows_total * splits * sizeof(RowBest) + 255) / 256 * 256;
    const size_t col_bytes = (size_t)cols_total * sizeof(unsigned long long);
    const size_t bytes = pair_bytes + row_bytes + col_bytes + 256;
    if (bytes > ctx->match_ws_bytes) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->d_match_ws) (void)hipFree(ctx->d_match_ws);
        ctx->d_match_ws = nullptr;
        ctx->match_ws_bytes = 0;
        ctx->d_match_cnt = nullptr;  // diagnostics pointer into the old workspace
        ctx->match_cnt_pairs = 0;
        HIP_TRY(hipMalloc(&ctx->d_match_ws, bytes));
        ctx->match_ws_bytes = bytes;
    }
    ctx->d_match_cnt = nullptr;  // the workspace is repurposed: the last screened match's counters are gone
    ctx->match_cnt_pairs = 0;
    char* ws = (char*)ctx->d_match_ws;
    MatchPair* d_pairs = (MatchPair*)ws;
    RowBest* d_rows = (RowBest*)(ws + pair_bytes);
    unsigned long long* d_cols = (unsigned long long*)(ws + pair_bytes + row_bytes);
    HIP_TRY(hipMemcpyAsync(d_pairs, hp.data(), (size_t)n_pairs * sizeof(MatchPair), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // hp is a local buffer
    if (col_bytes) HIP_TRY(hipMemsetAsync(d_cols, 0xFF, col_bytes, ctx->stream));
    if (max_rb > 0 && min_tiles != ~0u) {
        const uint32_t per_pair = max_rb * splits, total = per_pair * n_pairs;
        if (nw == 8)
            hipLaunchKernelGGL(desc_top2_kernel<8>, dim3(total), dim3(512), 0, ctx->stream, d_pairs, d_rows, d_cols, splits, per_pair);
        else
            hipLaunchKernelGGL(desc_top2_kernel<4>, dim3(total), dim3(256), 0, ctx->stream, d_pairs, d_rows, d_cols, splits, per_pair);
        HIP_TRY(hipGetLastError());
    }
    uint32_t np = 2;
    while (np < max_na) np <<= 1;
    const size_t lds = (size_t)np * sizeof(unsigned long long);
    static_assert(PGI_DESC_MAX * sizeof(unsigned long long) <= 128 * 1024, "selection keys must fit in LDS");
    HIP_TRY(hipFuncSetAttribute((const void*)match_select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    hipLaunchKernelGGL(match_select_kernel, dim3(n_pairs), dim3(1024), lds, ctx->stream, d_pairs, d_rows, d_cols, splits, max_matches,
                       d_match_src, d_match_dst, d_ratio, d_counts);
    HIP_TRY(hipGetLastError());
    return PGI_SUCCESS;
}

int pgi_build_correspondences(pgi_ctx* ctx, const pgi_keypoint_view* h_src, const pgi_keypoint_view* h_dst, uint32_t n_pairs,
                              uint32_t max_matches, const uint32_t* d_match_src, const uint32_t* d_match_dst,
                              const uint32_t* d_counts, uint32_t top_k, double thr_px, uint32_t dst_uses_src_intrinsics,
                              float* d_x1, float* d_y1, float* d_x2, float* d_y2, uint64_t* d_offsets, double* d_thr) {
    if (!ctx || !d_offsets) return pgi::fail(PGI_ERR_INVALID, "pgi_build_correspondences: null argument");
    std::lock_guard<std::mutex> lock(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    if (n_pairs == 0) {
        HIP_TRY(hipMemsetAsync(d_offsets, 0, sizeof(uint64_t), ctx->stream));
        return PGI_SUCCESS;
    }
    if (!h_src || !h_dst || !d_match_src || !d_match_dst || !d_counts || !d_x1 || !d_y1 || !d_x2 || !d_y2 || !d_thr ||
        max_matches == 0)
        return pgi::fail(PGI_ERR_INVALID, "pgi_build_correspondences: null argument");
    std::vector<KpPair> hp(n_pairs);
    for (uint32_t p = 0; p < n_pairs; ++p) {
        const pgi_keypoint_view &a = h_src[p], &b = h_dst[p];
        if ((a.n && !a.d_xy) || (b.n && !b.d_xy)) return pgi::fail(PGI_ERR_INVALID, "pgi_build_correspondences: bad keypoint view");
        const pgi_keypoint_view& k = dst_uses_src_intrinsics ? a : b;  // pose_graph_builder.h:908-912 when set
        hp[p] = KpPair{a.d_xy, b.d_xy, a.n, b.n, a.fx, a.fy, a.cx, a.cy, k.fx, k.fy, k.cx, k.cy};
    }
    const size_t bytes = (size_t)n_pairs * sizeof(KpPair);
    if (bytes > ctx->match_ws_bytes) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->d_match_ws) (void)hipFree(ctx->d_match_ws);
        ctx->d_match_ws = nullptr;
        ctx->match_ws_bytes = 0;
        ctx->d_match_cnt = nullptr;  // diagnostics pointer into the old workspace
        ctx->match_cnt_pairs = 0;
        HIP_TRY(hipMalloc(&ctx->d_match_ws, bytes));
        ctx->match_ws_bytes = bytes;
    }
    ctx->d_match_cnt = nullptr;  // the workspace is repurposed: the last screened match's counters are gone
    ctx->match_cnt_pairs = 0;
    KpPair* d_pairs = (KpPair*)ctx->d_match_ws;
    HIP_TRY(hipMemcpyAsync(d_pairs, hp.data(), bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // hp is a local buffer
    const uint32_t keep = top_k ? top_k : 0xffffffffu;
    hipLaunchKernelGGL(corr_offsets_kernel, dim3(1), dim3(1024), 0, ctx->stream, d_counts, n_pairs, keep,
                       (unsigned long long*)d_offsets);
    hipLaunchKernelGGL(corr_gather_kernel, dim3(n_pairs), dim3(256), 0, ctx->stream, d_pairs, d_match_src, d_match_dst, max_matches,
                       (const unsigned long long*)d_offsets, thr_px, d_x1, d_y1, d_x2, d_y2, d_thr);
    HIP_TRY(hipGetLastError());
    return PGI_SUCCESS;
}

// Right null vector of a 3x3 matrix by one-sided (Hestenes) Jacobi: the epipole of matcher.h:220-226 (JacobiSVD,
// ComputeFullV, column 2).  Host code; same operation order as the specification in oracle/pgi_oracle.c.
static void right_null_vector3(const double* A, double out[3]) {
    double B[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, sv[3];
    for (int i = 0; i < 9; ++i) B[i] = A[i];
    for (int sweep = 0; sweep < 40; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double al = 0, be = 0, ga = 0;
                for (int k = 0; k < 3; ++k) {
                    al += B[3 * k + p] * B[3 * k + p];
                    be += B[3 * k + q] * B[3 * k + q];
                    ga += B[3 * k + p] * B[3 * k + q];
                }
                if (std::fabs(ga) <= 1e-17 * std::sqrt(al * be) || ga == 0.0) continue;
                rotated = true;
                const double zeta = (be - al) / (2.0 * ga);
                const double tt = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / std::sqrt(1.0 + tt * tt), sn = c * tt;
                for (int k = 0; k < 3; ++k) {
                    const double bp = B[3 * k + p], bq = B[3 * k + q];
                    B[3 * k + p] = c * bp - sn * bq;
                    B[3 * k + q] = sn * bp + c * bq;
                    const double vp = V[3 * k + p], vq = V[3 * k + q];
                    V[3 * k + p] = c * vp - sn * vq;
                    V[3 * k + q] = sn * vp + c * vq;
                }
            }
        if (!rotated) break;
    }
    for (int j = 0; j < 3; ++j) {
        double s2 = 0;
        for (int k = 0; k < 3; ++k) s2 += B[3 * k + j] * B[3 * k + j];
        sv[j] = std::sqrt(s2);
    }
    for (int a = 0; a < 2; ++a) {  // descending singular values, columns of V along
        int m = a;
        for (int b = a + 1; b < 3; ++b)
            if (sv[b] > sv[m]) m = b;
        if (m != a) {
            std::swap(sv[a], sv[m]);
            for (int k = 0; k < 3; ++k) std::swap(V[3 * k + a], V[3 * k + m]);
        }
    }
    out[0] = V[2]; out[1] = V[5]; out[2] = V[8];
}

int pgi_guided_match_batch(pgi_ctx* ctx, const pgi_feature_view* h_src, const pgi_feature_view* h_dst, uint32_t n_pairs,
                           const double* h_pose_Rt, uint32_t n_bins, uint32_t max_n, uint32_t out_stride,
                           uint32_t* d_match_src, uint32_t* d_match_dst, double* d_ratio, uint32_t* d_counts) {
    if (!ctx || !d_counts) return pgi::fail(PGI_ERR_INVALID, "pgi_guided_match_batch: null argument");
    if (n_pairs == 0) return PGI_SUCCESS;
    if (!h_src || !h_dst || !h_pose_Rt || !d_match_src || !d_match_dst || !d_ratio || out_stride == 0)
        return pgi::fail(PGI_ERR_INVALID, "pgi_guided_match_batch: null argument");
    if (n_bins > 4096) return pgi::fail(PGI_ERR_INVALID, "pgi_guided_match_batch: n_bins out of range");
    for (uint32_t p = 0; p < n_pairs; ++p)
        if (max_n == 0 ? out_stride < h_src[p].n : out_stride < max_n)
            return pgi::fail(PGI_ERR_INVALID, "pgi_guided_match_batch: out_stride is smaller than the matches a pair may return");
    std::lock_guard<std::mutex> lock(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    std::vector<GuidedPair> hp(n_pairs);
    uint64_t total = 0, total2 = 0, active_waves = 0;  // active_waves: wavefronts of 64 sources that hold at least one source
    uint32_t max_n1 = 0, max_kp = 0;
    for (uint32_t p = 0; p < n_pairs; ++p) {
        const pgi_feature_view &a = h_src[p], &b = h_dst[p];
        if ((a.n && (!a.d_xy || !a.d_desc)) || (b.n && (!b.d_xy || !b.d_desc)))
            return pgi::fail(PGI_ERR_INVALID, "pgi_guided_match_batch: bad feature view");
        GuidedPair g{a.d_xy, b.d_xy, a.d_desc, b.d_desc, a.n, b.n, {}, total, 0.0, 0.0, 0.0, 0.0, 0, 0, total2};
        // E = [t]x R (pose.h:50, pose_utils.h), F = K_dst^-T E K_src^-1 (matcher.h:216-217), operation order as in the oracle
        const double* R = h_pose_Rt + 12 * (size_t)p;
        const double* t = R + 9;
        const double tx[9] = {0, -t[2], t[1], t[2], 0, -t[0], -t[1], t[0], 0};
        double E[9], G[9];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                double s = 0;
                for (int k = 0; k < 3; ++k) s += tx[3 * r + k] * R[3 * k + c];
                E[3 * r + c] = s;
            }
        for (int r = 0; r < 3; ++r) {
            G[3 * r + 0] = E[3 * r + 0] / a.fx;
            G[3 * r + 1] = E[3 * r + 1] / a.fy;
            G[3 * r + 2] = (E[3 * r + 2] - G[3 * r + 0] * a.cx) - G[3 * r + 1] * a.cy;
        }
        for (int c = 0; c < 3; ++c) {
            g.F[c] = G[c] / b.fx;
            g.F[3 + c] = G[3 + c] / b.fy;
            g.F[6 + c] = (G[6 + c] - g.F[c] * b.cx) - g.F[3 + c] * b.cy;
        }
        if (n_bins) {  // epipole, angular range and bin count (matcher.h:218-266)
            double nv[3];
            right_null_vector3(g.F, nv);
            g.ep0 = nv[0] / nv[2];
            g.ep1 = nv[1] / nv[2];
            const int sw = (int)a.width, sh = (int)a.height, dw = (int)b.width, dh = (int)b.height;  // cv::Size is integral
            const bool inImage = g.ep0 >= 0 && g.ep0 < sw && g.ep1 >= 0 && g.ep1 < sh;
            double minAngle = 180, maxAngle = 0;
            if (!inImage) {
                const double corner[8] = {0, 0, (double)dw, 0, (double)dw, (double)dh, 0, (double)dh};
                for (int c = 0; c < 8; c += 2) {
                    const double x = corner[c], y = corner[c + 1];
                    const double nx = g.F[0] * x + g.F[3] * y + g.F[6];
                    const double ny = g.F[1] * x + g.F[4] * y + g.F[7];
                    double angle = kRadianToDegree * std::atan2(ny, nx) + 180.0;
                    if (angle > 180) angle -= 180;
                    minAngle = angle < minAngle ? angle : minAngle;
                    maxAngle = angle > maxAngle ? angle : maxAngle;
                }
            }
            g.min_angle = minAngle;
            g.range = maxAngle - minAngle;
            g.bins = (int32_t)n_bins;
        }
        hp[p] = g;
        total += a.n;
        total2 += b.n;
        active_waves += (a.n + 63u) / 64u;
        max_n1 = a.n > max_n1 ? a.n : max_n1;
        max_kp = std::max(max_kp, std::max(a.n, b.n));
    }
    // bucketed scan (the hashing saves the work) where the bins fit its counters; otherwise the tiled scan over everything
    const bool bucketed = n_bins > 0 && n_bins <= (uint32_t)kGmMaxBuckets && max_kp <= (uint32_t)PGI_DESC_MAX;
    // default for the binned mode: the fine angular index + pooled scan (guided_scan_flat_kernel); PGI_GUIDED_ANGLE=0 keeps the
    // bin-bucketed scan, PGI_GUIDED_LANES=2 the round-3 tile scan, PGI_GUIDED_CAP = 4 | 16 the pooled scan with shorter lists
    // (read per call: tests switch them)
    const char* angle_txt = getenv("PGI_GUIDED_ANGLE");
    const bool angle_env = !angle_txt || atoi(angle_txt) != 0;
    const bool angular = n_bins > 0 && angle_env && max_kp <= 65535u;  // (record positions are packed in 16 bits)
    const size_t pair_bytes = ((size_t)n_pairs * sizeof(GuidedPair) + 255) / 256 * 256;
    const size_t slot = ((size_t)total * 8 + 255) / 256 * 256;  // one 8-byte array of `total` entries
    const size_t so_bytes = bucketed ? ((size_t)total * 4 + 255) / 256 * 256 : 0, do_bytes = bucketed ? ((size_t)total2 * 4 + 255) / 256 * 256 : 0,
                 rec_bytes = bucketed ? ((size_t)total2 * 32 + 255) / 256 * 256 : 0,
                 st_bytes = bucketed ? ((size_t)n_pairs * 3 * (kGmMaxBuckets + 1) * 4 + 255) / 256 * 256 : 0;
    const size_t kept_bytes = ((size_t)n_pairs * 4 + 255) / 256 * 256;
    const size_t ga_entry_bytes = angular ? ((size_t)total2 * sizeof(GaEntry) + 255) / 256 * 256 : 0,
                 ga_key_bytes = angular ? ((size_t)total2 * 4 + 255) / 256 * 256 : 0,
                 ga_start_bytes = angular ? ((size_t)n_pairs * (kGaBuckets + 2) * 4 + (size_t)n_pairs * 16 + 255) / 256 * 256 : 0;
    // the scan as two kernels (the default; PGI_GUIDED_SPLIT=0: one kernel): per wavefront two table words, kDealArenas heads, an arena
    const char* split_txt = getenv("PGI_GUIDED_SPLIT");
    const char* cap_txt0 = getenv("PGI_GUIDED_CAP");
    const char* lanes_txt0 = getenv("PGI_GUIDED_LANES");
    bool split = angular && !(split_txt && atoi(split_txt) == 0) && !cap_txt0 && !(lanes_txt0 && atoi(lanes_txt0) != 0);
    // words of arena per wavefront on average (a round's block reserves ~1800 at 24 candidates per source, about a sixth of the
    // wavefronts take a second one); PGI_GUIDED_ARENA_WORDS shrinks it (tests: wavefronts that do not fit are redone in one kernel)
    const char* aw_txt = getenv("PGI_GUIDED_ARENA_WORDS");
    const size_t kDealWords = aw_txt && atoi(aw_txt) > 0 ? (size_t)atoi(aw_txt) : 2560;
    const size_t deal_waves = (size_t)((max_n1 + 63u) / 64u) * n_pairs;  // the launch's wavefront index space (tables: 8 bytes each)
    // The ARENA is sized by the wavefronts that hold sources (the sum of ceil(n1 / 64) over the pairs), not by the largest view
    // times the pairs (round 5: 2048 pairs with one 20 000-keypoint view among 2 000-keypoint ones asked for 6.5 GB), and it is
    // capped -- by what the device had free when the workspace last had to grow (ctx->guided_arena_cap), and by the 32-bit word
    // offsets of a part.  A smaller arena is not an error: wavefronts whose lists do not fit are flagged and redone by the
    // one-kernel scan (guided_redo_kernel); without any arena the call runs the one-kernel scan outright.
    // (a part serves every 64th wavefront: small launches get room for a dozen blocks per part whatever the average says)
    const size_t part_words_max = 0xFFFFFFFFull / kDealArenas - 1;
    auto part_words_for = [&](size_t arena_bytes_cap) -> size_t {
        size_t w = std::max<size_t>(((size_t)active_waves * kDealWords + kDealArenas - 1) / kDealArenas, aw_txt ? 0 : 32768);
        w = std::min(w, part_words_max);
        return std::min(w, arena_bytes_cap / (4 * kDealArenas));
    };
    if (const char* e = getenv("PGI_GUIDED_ARENA_CAP_MB"))  // (tests: a device that is short of memory)
        ctx->guided_arena_cap = std::min(ctx->guided_arena_cap, (size_t)std::max(0, atoi(e)) << 20);
    size_t deal_part_words = split ? part_words_for(ctx->guided_arena_cap) : 0;
    if (split && deal_part_words < 4096 && !aw_txt) split = false;  // (a cap that leaves no useful arena)
    if (!split) deal_part_words = 0;
    const size_t deal_table_bytes_split = ((kDealArenas + 2 * deal_waves) * 4 + 255) / 256 * 256;
    size_t deal_table_bytes = split ? deal_table_bytes_split : 0;
    size_t deal_arena_bytes = split ? (deal_part_words * kDealArenas * 4 + 255) / 256 * 256 : 0;
    const size_t bytes_fixed = pair_bytes + 5 * slot + so_bytes + do_bytes + rec_bytes + st_bytes + kept_bytes + ga_entry_bytes + ga_key_bytes +
                               ga_start_bytes + 256;
    size_t bytes = bytes_fixed + deal_table_bytes + deal_arena_bytes;
    if (bytes > ctx->match_ws_bytes) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->d_match_ws) (void)hipFree(ctx->d_match_ws);
        ctx->d_match_ws = nullptr;
        ctx->match_ws_bytes = 0;
        ctx->d_match_cnt = nullptr;  // diagnostics pointer into the old workspace
        ctx->match_cnt_pairs = 0;
        if (split) {  // the arena may take half of what is free now, and never what the rest of the workspace needs
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
                const size_t room = free_b / 2 > bytes_fixed + deal_table_bytes ? free_b / 2 - bytes_fixed - deal_table_bytes : 0;
                if (deal_arena_bytes > room) {
                    ctx->guided_arena_cap = room;
                    deal_part_words = part_words_for(room);
                    if (deal_part_words < 4096) { split = false; deal_part_words = 0; deal_table_bytes = 0; }
                    deal_arena_bytes = split ? (deal_part_words * kDealArenas * 4 + 255) / 256 * 256 : 0;
                    bytes = bytes_fixed + deal_table_bytes + deal_arena_bytes;
                }
            }
        }
        hipError_t me = hipMalloc(&ctx->d_match_ws, bytes);
        if (me != hipSuccess && split) {  // no room for the lists: the one-kernel scan needs none
            (void)hipGetLastError();
            ctx->d_match_ws = nullptr;
            ctx->guided_arena_cap = 0;
            split = false;
            deal_part_words = 0; deal_table_bytes = 0; deal_arena_bytes = 0;
            bytes = bytes_fixed;
            me = hipMalloc(&ctx->d_match_ws, bytes);
        }
        if (me != hipSuccess) {
            ctx->d_match_ws = nullptr;
            return pgi::fail(PGI_ERR_DEVICE, std::string("pgi_guided_match_batch: hipMalloc of the workspace: ") + hipGetErrorString(me));
        }
        ctx->match_ws_bytes = bytes;
    }
    ctx->d_match_cnt = nullptr;  // the workspace is repurposed: the last screened match's counters are gone
    ctx->match_cnt_pairs = 0;
    char* ws = (char*)ctx->d_match_ws;
    GuidedPair* d_pairs = (GuidedPair*)ws;
    double* d_rat = (double*)(ws + pair_bytes);
    double* d_cr = (double*)(ws + pair_bytes + slot);
    int32_t* d_best = (int32_t*)(ws + pair_bytes + 2 * slot);
    uint32_t* d_ci = (uint32_t*)(ws + pair_bytes + 3 * slot);
    uint32_t* d_cj = (uint32_t*)(ws + pair_bytes + 4 * slot);
    HIP_TRY(hipMemcpyAsync(d_pairs, hp.data(), (size_t)n_pairs * sizeof(GuidedPair), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // hp is a local buffer
    if (max_n1 > 0 && angular) {
        char* q = ws + pair_bytes + 5 * slot + so_bytes + do_bytes + rec_bytes + st_bytes + kept_bytes;
        GaEntry* d_ent = (GaEntry*)q; q += ga_entry_bytes;
        uint32_t* d_key = (uint32_t*)q; q += ga_key_bytes;
        double* d_span = (double*)q;  // (8-byte aligned: first in the block)
        uint32_t* d_gst = (uint32_t*)(q + (size_t)n_pairs * 16);
        hipLaunchKernelGGL(guided_angle_bucket_kernel, dim3(n_pairs), dim3(1024), 0, ctx->stream, d_pairs, d_ent, d_key, d_gst, d_span,
                           d_cj /* the source order: guided_select_kernel overwrites it only after the scan */);
        const char* lanes_txt = getenv("PGI_GUIDED_LANES");
        const int lanes_env = lanes_txt ? atoi(lanes_txt) : 0;
        const char* cap_txt = getenv("PGI_GUIDED_CAP");
        const int cap_env = cap_txt ? atoi(cap_txt) : kGfCap;
        const uint32_t flat_per_pair = (max_n1 + 63u) / 64u, flat_blocks = flat_per_pair * n_pairs;
        const dim3 flat_grid((flat_blocks + 7u) / 8u * 8u);
        const char* ph_txt = getenv("PGI_GUIDED_PHASES");
        const bool phases = ph_txt && atoi(ph_txt) != 0;
#define PGI_FLAT(CAP, PHV) hipLaunchKernelGGL((guided_scan_flat_kernel<CAP, PHV>), flat_grid, dim3(64), 0, ctx->stream, d_pairs, d_ent, d_gst, d_span, d_cj, \
                                              d_best, d_rat, flat_per_pair, flat_blocks)
        if (lanes_env == 0 && phases) {  // diagnostics: where the wavefronts' time goes
            unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0}, got[8];
            HIP_TRY(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_guided_phase), zero, sizeof(zero), 0, hipMemcpyHostToDevice, ctx->stream));
            PGI_FLAT(kGfCap, true);
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            HIP_TRY(hipMemcpyFromSymbol(got, HIP_SYMBOL(g_guided_phase), sizeof(got)));
            double all = 0;
            for (int k = 0; k < 7; ++k) all += (double)got[k];
            fprintf(stderr, "[pgi] guided scan phases (%u pairs): setup + source rows %.1f %%, gate pass %.1f, deal %.1f, rows %.1f, sums %.1f, pick %.1f"
                    " | %.0f s_memtime ticks per wavefront, %u wavefronts launched\n",
                    n_pairs, 100 * got[0] / all, 100 * got[2] / all, 100 * got[3] / all, 100 * got[4] / all, 100 * got[5] / all, 100 * got[6] / all,
                    all / flat_blocks, flat_blocks);
        }
        else if (lanes_env == 0 && split && cap_env == kGfCap) {
            char* dq = q + ga_start_bytes;
            DealOut dout;
            dout.heads = (uint32_t*)dq;
            dout.first = dout.heads + kDealArenas;
            dout.flags = dout.first + deal_waves;
            dout.words = (uint32_t*)(dq + deal_table_bytes);
            dout.part_words = (uint32_t)deal_part_words;
            HIP_TRY(hipMemsetAsync(dq, 0, deal_table_bytes, ctx->stream));
            hipLaunchKernelGGL((guided_deal_kernel<kSplitCap>), flat_grid, dim3(64), 0, ctx->stream, d_pairs, d_ent, d_gst, d_span, d_cj, flat_per_pair,
                               flat_blocks, dout);
            hipLaunchKernelGGL((guided_sum_kernel<kSplitCap>), flat_grid, dim3(64), 0, ctx->stream, d_pairs, d_ent, d_cj, d_best, d_rat, flat_per_pair,
                               flat_blocks, dout);
            hipLaunchKernelGGL((guided_redo_kernel<kGfCap>), flat_grid, dim3(64), 0, ctx->stream, d_pairs, d_ent, d_gst, d_span, d_cj, d_best, d_rat,
                               flat_per_pair, flat_blocks, dout);
            if (getenv("PGI_GUIDED_ARENA_REPORT")) {  // diagnostics: how full the arena got, how many wavefronts were redone
                std::vector<uint32_t> h(kDealArenas + 2 * deal_waves);
                HIP_TRY(hipStreamSynchronize(ctx->stream));
                HIP_TRY(hipMemcpy(h.data(), dq, h.size() * 4, hipMemcpyDeviceToHost));
                uint64_t used = 0, redone = 0;
                uint32_t fullest = 0;
                for (uint32_t k = 0; k < kDealArenas; ++k) { used += std::min<uint32_t>(h[k], dout.part_words); fullest = std::max(fullest, h[k]); }
                for (size_t k = 0; k < deal_waves; ++k) redone += h[kDealArenas + deal_waves + k];
                fprintf(stderr, "[pgi] guided scan arena (%u pairs): %.1f MB of %.1f MB used (%.0f words per wavefront, fullest part %.0f %%), %llu of %zu wavefronts redone\n",
                        n_pairs, used * 4e-6, (double)deal_part_words * kDealArenas * 4e-6, (double)used / std::max<uint64_t>(1, active_waves),
                        100.0 * fullest / dout.part_words, (unsigned long long)redone, deal_waves);
            }
        }
        else if (lanes_env == 0 && cap_env == 4) PGI_FLAT(4, false);   // (tests: several rounds per source)
        else if (lanes_env == 0 && cap_env == 16) PGI_FLAT(16, false);
        else if (lanes_env == 0) PGI_FLAT(kGfCap, false);
#undef PGI_FLAT
        else  // PGI_GUIDED_LANES != 0: the round-3 scan, two lanes per source (kept as the cross-check of tests and soaks)
            hipLaunchKernelGGL(guided_scan_tile_kernel<2>, dim3((max_n1 + 31) / 32, n_pairs), dim3(64), 0, ctx->stream, d_pairs, d_ent, d_gst, d_span,
                               d_cj, d_best, d_rat);
        HIP_TRY(hipGetLastError());
    } else if (max_n1 > 0 && bucketed) {
        char* q = ws + pair_bytes + 5 * slot;
        uint32_t* d_so = (uint32_t*)q; q += so_bytes;
        uint32_t* d_do = (uint32_t*)q; q += do_bytes;
        double* d_rec = (double*)q; q += rec_bytes;
        uint32_t* d_starts = (uint32_t*)q;
        uint32_t* d_chunks = d_starts + (size_t)n_pairs * 2 * (kGmMaxBuckets + 1);
        hipLaunchKernelGGL(guided_bucket_kernel, dim3(n_pairs, 2), dim3(1024), 0, ctx->stream, d_pairs, d_so, d_do, d_rec, d_starts, d_chunks);
        hipLaunchKernelGGL(guided_scan_binned_kernel, dim3((max_n1 + 255) / 256 + n_bins, n_pairs), dim3(256), 0, ctx->stream, d_pairs,
                           d_so, d_do, d_rec, d_starts, d_chunks, d_best, d_rat);
        HIP_TRY(hipGetLastError());
    } else if (max_n1 > 0) {
        hipLaunchKernelGGL(guided_scan_kernel, dim3((max_n1 + 255) / 256, n_pairs), dim3(256), 0, ctx->stream, d_pairs, d_best, d_rat);
        HIP_TRY(hipGetLastError());
    }
    uint32_t* d_kept = (uint32_t*)(ws + pair_bytes + 5 * slot + so_bytes + do_bytes + rec_bytes + st_bytes);
    hipLaunchKernelGGL(guided_select_kernel, dim3(n_pairs), dim3(1024), 0, ctx->stream, d_pairs, d_best, d_rat, d_ci, d_cj, d_cr, max_n,
                       out_stride, d_match_src, d_match_dst, d_ratio, d_counts, d_kept);
    if (max_n != 0u)  // pairs with more kept matches than the cut: rank them
    {
        if (std::min(max_n, out_stride) <= kTopMax)
            hipLaunchKernelGGL(guided_topn_kernel, dim3(n_pairs), dim3(1024), 0, ctx->stream, d_pairs, d_kept, d_ci, d_cj, d_cr, max_n, out_stride,
                               d_match_src, d_match_dst, d_ratio);
        else
            hipLaunchKernelGGL(guided_rank_kernel, dim3(kRankSplit, n_pairs), dim3(1024), 0, ctx->stream, d_pairs, d_kept, d_ci, d_cj, d_cr, max_n,
                               out_stride, d_match_src, d_match_dst, d_ratio);
    }
    HIP_TRY(hipGetLastError());
    return PGI_SUCCESS;
}

// diagnostics (tests, scripts): rows the last screened match had to re-scan exactly, summed over pairs; synchronises
int pgi_internal_match_flagged(pgi_ctx* ctx, uint64_t* forward, uint64_t* backward) {
    if (!ctx || !forward || !backward) return PGI_ERR_INVALID;
    std::lock_guard<std::mutex> lock(ctx->mu);
    *forward = *backward = 0;
    if (!ctx->d_match_cnt || !ctx->match_cnt_pairs) return PGI_SUCCESS;
    std::vector<uint32_t> h((size_t)2 * ctx->match_cnt_pairs);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(h.data(), ctx->d_match_cnt, h.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (uint32_t p = 0; p < ctx->match_cnt_pairs; ++p) { *forward += h[p]; *backward += h[ctx->match_cnt_pairs + p]; }
    return PGI_SUCCESS;
}
}  // extern "C"
