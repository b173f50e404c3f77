// pgi_device.hpp -- gfx950 device building blocks of the pairwise relative-pose engine.
//
// Execution model: a wavefront (64 lanes) is split into four DPP rows of 16
// lanes; one row ("group") solves one 5-point hypothesis cooperatively (matrix
// rows live one-per-lane in registers, pivots are found with DPP row reductions,
// pivot rows travel by ds_bpermute), then the whole wavefront scores the
// resulting models over the LDS-resident correspondences with __ballot/popcount.
//
// Numerics: every floating-point statement is ONE IEEE-754 operation in a fixed
// order (compile with -ffp-contract=off); reductions are either integer
// popcounts or exact sums, so results do not depend on the lane layout.  The
// algorithm is specified in DESIGN.md §3; reference citations are relative to
// /root/reference/src/pyposegraphbuilder/include/.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PGI_DEV __device__ __forceinline__
// coarse phases are real calls: register allocation (and spilling) is per phase, and only the
// caller's few live values are saved around a call instead of inside the hot loops
#define PGI_PHASE __device__ __noinline__

namespace pgi {

constexpr int kGrid = 256;       // root-bracketing intervals
constexpr int kNewton = 10;      // safeguarded Newton iterations
constexpr int kJacobiSweeps = 6; // 9x9 tournament Jacobi sweeps
constexpr int kSvdSweeps = 4;    // 3x3 one-sided Jacobi sweeps
constexpr int kMaxModels = 10;

// ---- optional per-phase cycle accounting (-DPGI_PROFILE; scripts/profile_phases.py) ----
constexpr int kProfSlots = 32;
#ifdef PGI_PROFILE
struct Prof {
    unsigned long long t, acc[kProfSlots];
    PGI_DEV void start() {
        for (int i = 0; i < kProfSlots; ++i) acc[i] = 0;
        t = __builtin_amdgcn_s_memtime();
    }
    template <int I>
    PGI_DEV void mark() {
        const unsigned long long n = __builtin_amdgcn_s_memtime();
        acc[I] += n - t;
        t = n;
    }
    PGI_DEV void flush(unsigned long long* out, int lane) {
        if (out && lane == 0)
            for (int i = 0; i < kProfSlots; ++i) atomicAdd(out + i, acc[i]);
    }
};
#else
struct Prof {
    PGI_DEV void start() {}
    template <int I>
    PGI_DEV void mark() {}
    PGI_DEV void flush(unsigned long long*, int) {}
};
#endif

// ---- wavefront-scope LDS ordering ------------------------------------------
// DS operations of one wavefront execute in order; this only stops the
// compiler from moving LDS accesses across the exchange point.
PGI_DEV void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- DPP row (16-lane) primitives -------------------------------------------
// quad_perm [1,0,3,2] = 0xB1, quad_perm [2,3,0,1] = 0x4E, row_half_mirror = 0x141,
// row_mirror = 0x140, row_shr:n = 0x110 + n.
template <int CTRL>
PGI_DEV int dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
PGI_DEV double dpp_d(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(dpp_i<CTRL>(hi), dpp_i<CTRL>(lo));
}
PGI_DEV int row_max_i(int k) {
    k = max(k, dpp_i<0xB1>(k));
    k = max(k, dpp_i<0x4E>(k));
    k = max(k, dpp_i<0x141>(k));
    k = max(k, dpp_i<0x140>(k));
    return k;
}
// pairwise tree ((p0+p1)+(p2+p3))+((p4+p5)+(p6+p7)) + (same over lanes 8..15)
PGI_DEV double row_sum_d(double x) {
    x = x + dpp_d<0xB1>(x);
    x = x + dpp_d<0x4E>(x);
    x = x + dpp_d<0x141>(x);
    x = x + dpp_d<0x140>(x);
    return x;
}
PGI_DEV int row_scan_incl_i(int x) {
    x += dpp_i<0x111>(x);
    x += dpp_i<0x112>(x);
    x += dpp_i<0x114>(x);
    x += dpp_i<0x118>(x);
    return x;
}
// exact sums (pre-rounded summands): any order is bit-identical
PGI_DEV double wave_sum_exact(double x) {
    x = row_sum_d(x);
    x = x + __shfl_xor(x, 16);
    x = x + __shfl_xor(x, 32);
    return x;
}

// One elimination step of the row-per-lane Gauss-Jordan sweeps: columns [J0, J1) of this lane's row are scaled, the pivot
// lane's scaled values fetched (ds_bpermute, byte address `addr` = 4 x pivot lane) and the update applied --
//     v = row[j] * scale;  p = v of the pivot lane;  row[j] = fma(-f, p, v)
// element for element what the plain loop does, but in BATCHES of CH columns: all products, then all fetches in flight
// together, then all updates.  Written as one statement per column the compiler waited for every fetch on its own
// (v_mul, 2 ds_bpermute, s_waitcnt lgkmcnt(0), v_fma -- 145 exposed LDS round trips per 10x20 sweep and wavefront, the
// single largest stall of K1: round 6, profiles/r06_k1_gj_isa.txt); sched_barriers pin the three phases.
#ifndef PGI_GJ_CH
#define PGI_GJ_CH 8  // columns per batch (experiment builds: make variant NAME=ch1 DEFS=-DPGI_GJ_CH=1 is the old one-by-one form)
#endif
template <int J0, int J1, int CH = PGI_GJ_CH>
PGI_DEV void eliminate_columns(double* row, double scale, double f, int addr) {
    if constexpr (J0 < J1) {
        constexpr int N = (J1 - J0) < CH ? (J1 - J0) : CH;
        int lo[N], hi[N];
#pragma unroll
        for (int c = 0; c < N; ++c) row[J0 + c] = row[J0 + c] * scale;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < N; ++c) {
            lo[c] = __builtin_amdgcn_ds_bpermute(addr, __double2loint(row[J0 + c]));
            hi[c] = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(row[J0 + c]));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < N; ++c) row[J0 + c] = fma(-f, __hiloint2double(hi[c], lo[c]), row[J0 + c]);
        __builtin_amdgcn_sched_barrier(0);
        eliminate_columns<J0 + N, J1, CH>(row, scale, f, addr);
    }
}

// ---- counter-based RNG --------------------------------------------------------
PGI_DEV uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
PGI_DEV uint32_t fmix32(uint32_t h) {  // murmur3 finaliser: 32-bit multiplies only
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}
PGI_DEV uint32_t draw_index(uint64_t base, uint32_t hyp, uint32_t k, uint32_t n) {
    const uint32_t lo = (uint32_t)base, hi = (uint32_t)(base >> 32);
    const uint32_t h = fmix32((lo ^ (hyp * 0x9E3779B1u)) + (hi ^ (k * 0x85EBCA77u)));
    return __umulhi(h, n);
}
// Progressive sampling (pgi_params.sampler = 1): the first n(hyp) rows, n(hyp) = max(5, floor(n * (s / 64)^(1/5))) with
// s = ceil(64 * (hyp + 1) / T).  Integer arithmetic on floor(2^32 * (i / 64)^(1/5)) -- the oracle's table
// (pgo_progressive_rows); no floating point, so nothing to keep bit-compatible and no f64 registers in the round loop.
__constant__ uint32_t kFifthRoot64[65] = {
    0x00000000u, 0x6F6E336Bu, 0x80000000u, 0x8ACFF893u, 0x93088C35u, 0x99BE7209u, 0x9F741C86u, 0xA472339Du, 0xA8E5A29Du, 0xACEC422Fu, 0xB09AFB46u,
    0xB4010FB1u, 0xB729FEAEu, 0xBA1EAD01u, 0xBCE6228Cu, 0xBF860882u, 0xC203001Du, 0xC460DFEFu, 0xC6A2E032u, 0xC8CBBB8Du, 0xCADDC7B6u, 0xCCDB0842u,
    0xCEC53D43u, 0xD09DEEBCu, 0xD26675BFu, 0xD42003BEu, 0xD5CBA87Au, 0xD76A56DCu, 0xD8FCE8FAu, 0xDA842364u, 0xDC00B7F1u, 0xDD734813u, 0xDEDC66D6u,
    0xE03C9A92u, 0xE1945E57u, 0xE2E42331u, 0xE42C513Du, 0xE56D4891u, 0xE6A76213u, 0xE7DAF02Eu, 0xE9083F70u, 0xEA2F9717u, 0xEB513993u, 0xEC6D64ECu,
    0xED84532Eu, 0xEE963AB8u, 0xEFA34E8Fu, 0xF0ABBEA6u, 0xF1AFB819u, 0xF2AF6567u, 0xF3AAEEA8u, 0xF4A279B9u, 0xF5962A67u, 0xF6862294u, 0xF772825Eu,
    0xF85B6838u, 0xF940F10Eu, 0xFA23385Bu, 0xFB025844u, 0xFBDE69ABu, 0xFCB78447u, 0xFD8DBEB6u, 0xFE612E8Du, 0xFF31E869u, 0xFFFFFFFFu};
PGI_DEV uint32_t progressive_rows(uint32_t hyp, uint32_t n, uint32_t T) {
    if (n <= 5u || T == 0u) return n;
    const uint32_t step = ((hyp + 1u) * 64u + (T - 1u)) / T;  // hyp < max_iters: no overflow
    if (step >= 64u) return n;
    const uint32_t m = __umulhi(kFifthRoot64[step], n);
    return m < 5u ? 5u : m;
}

// Five distinct row indices: draws k = 0,1,2,... are accepted in order unless they repeat an
// earlier accepted index.  The first eight draws are hashed up front (independent chains).
PGI_DEV void sample5(uint64_t base, uint32_t hyp, uint32_t n, uint32_t idx[5]) {
    uint32_t cand[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) cand[k] = draw_index(base, hyp, (uint32_t)k, n);
    uint32_t got = 0;
    idx[0] = idx[1] = idx[2] = idx[3] = idx[4] = 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t j = cand[k];
        bool dup = false;
#pragma unroll
        for (int q = 0; q < 5; ++q) dup |= (q < (int)got) && (idx[q] == j);
        const bool take = (got < 5) && !dup;
#pragma unroll
        for (int q = 0; q < 5; ++q)
            if (take && q == (int)got) idx[q] = j;
        got += take ? 1u : 0u;
    }
    uint32_t k = 8;
    while (got < 5) {  // rare: three or more repeats among the first eight draws
        const uint32_t j = draw_index(base, hyp, k, n);
        ++k;
        bool dup = false;
#pragma unroll
        for (int q = 0; q < 5; ++q) dup |= (q < (int)got) && (idx[q] == j);
        if (!dup || k >= 64) {
#pragma unroll
            for (int q = 0; q < 5; ++q)
                if (q == (int)got) idx[q] = j;
            ++got;
        }
    }
}

// The same five indices, drawn cooperatively by a 16-lane group (K1's passes; round 3).  sample5 makes every lane hash
// all eight candidate draws and run the whole acceptance cascade (~320 instructions, 24 of them quarter-rate integer
// multiplies); here sub-lane k < 8 hashes draw k only, "repeats an accepted index" becomes "equals an earlier draw" (a draw
// rejected earlier equals an accepted one, so the two tests coincide; seven DPP row shifts), the accepted draws are ranked
// by a ballot and the first five travel through `slot` (five words of group-private LDS that are dead at this point).
// Wave-uniform slow path -- some group has fewer than five distinct values among its first eight draws (tiny pairs) --
// is sample5 itself, so the result is sample5's in every case.
PGI_DEV void sample5_group(uint64_t base, uint32_t hyp, uint32_t n, int s, int gbase, int lane, uint32_t* slot, uint32_t idx[5]) {
    const uint32_t cand = draw_index(base, hyp, (uint32_t)(s & 7), n);
    bool dup = false;
#define PGI_SHR_CMP(K) dup |= (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)cand, 0x110 + K, 0xF, 0xF, false) == cand;
    PGI_SHR_CMP(1) PGI_SHR_CMP(2) PGI_SHR_CMP(3) PGI_SHR_CMP(4) PGI_SHR_CMP(5) PGI_SHR_CMP(6) PGI_SHR_CMP(7)
#undef PGI_SHR_CMP
    const bool acc = s < 8 && !dup;  // (row_shr leaves the sentinel in sub-lanes below K: never equal to an index < n)
    const uint32_t gm = (uint32_t)(__ballot(acc) >> gbase) & 0xFFu;
    const bool enough = __popc(gm) >= 5;
    if (__builtin_amdgcn_readfirstlane((int)__popcll(__ballot(enough))) == 64) {
        const uint32_t rank = (uint32_t)__popc(gm & ((1u << s) - 1u));
        if (acc && rank < 5u) slot[rank] = cand;
        wave_sync();
#pragma unroll
        for (int k = 0; k < 5; ++k) idx[k] = slot[k];
        wave_sync();
    } else {
        sample5(base, hyp, n, idx);
    }
}

// ---- Sampson terms (f32; the scoring primitive) ---------------------------------
// r = p2^T E p1 and the gradient norm of graph_traversal.h:107-115, evaluated with
// one fused multiply-add per term.
PGI_DEV void sampson_terms(const float e[9], float x1, float y1, float x2, float y2, float& r2,
                           float& den) {
    const float rxc = fmaf(e[0], x2, fmaf(e[3], y2, e[6]));
    const float ryc = fmaf(e[1], x2, fmaf(e[4], y2, e[7]));
    const float rwc = fmaf(e[2], x2, fmaf(e[5], y2, e[8]));
    const float r = fmaf(x1, rxc, fmaf(y1, ryc, rwc));
    const float rx = fmaf(e[0], x1, fmaf(e[1], y1, e[2]));
    const float ry = fmaf(e[3], x1, fmaf(e[4], y1, e[5]));
    den = fmaf(rxc, rxc, fmaf(ryc, ryc, fmaf(rx, rx, ry * ry)));
    r2 = r * r;
}

// Two models at once on the packed-f32 VALU path (v_pk_fma_f32 / v_pk_mul_f32: two IEEE operations per lane and
// instruction, which is what the chip's f32 vector peak is quoted on).  Component by component the same operations in
// the same order as sampson_terms, so the bits are the same.  K1 is bound by vector-ALU issue (>= 68 % busy,
// profiles/r02_k1_pmc.json), so instructions saved are time saved.
typedef float f32x2 __attribute__((ext_vector_type(2)));
PGI_DEV f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
PGI_DEV void sampson_terms2(const f32x2 e[9], float x1, float y1, float x2, float y2, f32x2& r2, f32x2& den) {
    const f32x2 X1 = {x1, x1}, Y1 = {y1, y1}, X2 = {x2, x2}, Y2 = {y2, y2};
    const f32x2 rxc = pk_fma(e[0], X2, pk_fma(e[3], Y2, e[6]));
    const f32x2 ryc = pk_fma(e[1], X2, pk_fma(e[4], Y2, e[7]));
    const f32x2 rwc = pk_fma(e[2], X2, pk_fma(e[5], Y2, e[8]));
    const f32x2 r = pk_fma(X1, rxc, pk_fma(Y1, ryc, rwc));
    const f32x2 rx = pk_fma(e[0], X1, pk_fma(e[1], Y1, e[2]));
    const f32x2 ry = pk_fma(e[3], X1, pk_fma(e[4], Y1, e[5]));
    den = pk_fma(rxc, rxc, pk_fma(ryc, ryc, pk_fma(rx, rx, ry * ry)));
    r2 = r * r;
}

// ---- graph-cut local optimisation (pgi_params.lo_graph_cut; specification: oracle/pgi_oracle.c, pgo_gc_*) -----------------
// Integer energies: kernel level k in 0..16 (a ladder of f32 comparisons on the scoring operands), unary weight 128 per level,
// pairwise terms lambda64 * (k_p + k_q) / (32 - k_p - k_q) / 32; neighbours = the rows of a 4-D grid cell in index order.
constexpr uint32_t kGcNone = 0xFFFFFFFFu, kGcLevels = 16u, kGcCells = 4096u;
constexpr int kGcUnary = 128;
PGI_DEV uint32_t gc_cell(float4 p) {
    const int a = (int)floorf(p.x * 8.0f), b = (int)floorf(p.y * 8.0f), c = (int)floorf(p.z * 8.0f), d = (int)floorf(p.w * 8.0f);
    return (uint32_t)(a & 7) | ((uint32_t)(b & 7) << 3) | ((uint32_t)(c & 7) << 6) | ((uint32_t)(d & 7) << 9);
}
PGI_DEV uint32_t gc_kernel_level(const float e[9], float4 p, float thr2) {
    float r2, den;
    sampson_terms(e, p.x, p.y, p.z, p.w, r2, den);
    const float t = (den * thr2) * 2.25f;
    uint32_t k = 0;
#pragma unroll
    for (uint32_t j = 1; j <= kGcLevels; ++j) k += r2 < t * ((float)j * 0.0625f) ? 1u : 0u;
    return k;
}
// message of row i given its predecessor's (dp, kp): delta_i = U(in) - U(out) + [best way into "inlier" - into "outlier"]
PGI_DEV int gc_delta(uint32_t k, bool has_prev, int dp, uint32_t kp, uint32_t lambda64) {
    int d = kGcUnary * ((int)kGcLevels - 2 * (int)k);
    if (has_prev) {
        const int v00 = (int)(lambda64 * (kp + k)), v11 = (int)(lambda64 * (2u * kGcLevels - kp - k)), v01 = (int)(lambda64 * 2u * kGcLevels);
        const int in1 = dp + v11 < v01 ? dp + v11 : v01;
        const int in0 = dp + v01 < v00 ? dp + v01 : v00;
        d += in1 - in0;
    }
    return d;
}
// label of the predecessor (dp, kp) of a row with kernel level k and label `lab`
PGI_DEV uint32_t gc_prev_label(uint32_t lab, uint32_t k, int dp, uint32_t kp, uint32_t lambda64) {
    const int v00 = (int)(lambda64 * (kp + k)), v11 = (int)(lambda64 * (2u * kGcLevels - kp - k)), v01 = (int)(lambda64 * 2u * kGcLevels);
    return lab ? (dp + v11 < v01 ? 1u : 0u) : (dp + v01 < v00 ? 1u : 0u);
}
// message words (one per row, global scratch): delta << 8 | k after the forward sweep (|delta| < 2^15 with lambda64 <= 255);
// a successor in a higher 64-row block adds its decision: bit 5 = decided, bit 6 = label
PGI_DEV uint32_t gc_pack(int delta, uint32_t k) { return ((uint32_t)delta << 8) | k; }
PGI_DEV int gc_word_delta(uint32_t w) { return (int)w >> 8; }
PGI_DEV uint32_t gc_word_k(uint32_t w) { return w & 31u; }
// the words are written by one lane and read by another lane of the SAME wavefront later on: agent-scope relaxed accesses go
// to L2 (no stale L1 line, no cache maintenance); same address = same channel, so program order is memory order
PGI_DEV uint32_t gc_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
PGI_DEV void gc_store(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ---- small f64 helpers -----------------------------------------------------------
PGI_DEV int pivot_key(double a, int row) {
    const uint32_t hi = (uint32_t)__double2hiint(a) & 0x7FFFFFFFu;
    return (int)((hi & 0xFFFFFFF0u) | (uint32_t)(15 - row));
}
PGI_DEV void cross3(const double a[3], const double b[3], double c[3]) {
    c[0] = fma(a[1], b[2], -(a[2] * b[1]));
    c[1] = fma(a[2], b[0], -(a[0] * b[2]));
    c[2] = fma(a[0], b[1], -(a[1] * b[0]));
}
// element-wise select (keeps the choice in registers: v_cndmask, never an indexed reload)
PGI_DEV void sel3(bool c, const double a[3], double r[3]) {
    r[0] = c ? a[0] : r[0];
    r[1] = c ? a[1] : r[1];
    r[2] = c ? a[2] : r[2];
}
// root-bracketing grid z_j = u/(1-u^2), u uniform in (-1,1): evaluated by the compiler in IEEE
// double (identical to the run-time expression) and kept in constant memory instead of LDS
struct GridTab {
    double v[kGrid + 1];
};
constexpr GridTab make_grid() {
    GridTab g{};
    for (int j = 0; j <= kGrid; ++j) {
        const double u = (double)(2 * j - kGrid) / (double)(kGrid + 1);
        g.v[j] = u / (1.0 - u * u);
    }
    return g;
}
__device__ constexpr GridTab kGridTab = make_grid();
// (grid lookups go through kGridTab)
PGI_DEV double horner10(const double p[11], double x) {
    double v = p[10];
#pragma unroll
    for (int c = 9; c >= 0; --c) v = fma(v, x, p[c]);
    return v;
}
PGI_DEV double refine_root(const double p[11], double lo, double hi, bool slo) {
    double x = 0.5 * (lo + hi), xbest = x, vbest = __builtin_inf();
    for (int it = 0; it < kNewton; ++it) {
        double v = p[10], d = 0.0;
#pragma unroll
        for (int c = 9; c >= 0; --c) {
            d = fma(d, x, v);
            v = fma(v, x, p[c]);
        }
        const double av = fabs(v);
        if (av < vbest) {
            vbest = av;
            xbest = x;
        }
        if ((v < 0.0) == slo) lo = x; else hi = x;
        double xn = x - v / d;
        if (!(xn >= lo && xn <= hi)) xn = 0.5 * (lo + hi);
        // fixpoint: the next iteration would evaluate the same x, change neither bracket nor best --
        // leaving now returns exactly what all kNewton iterations would
        if (xn == x) break;
        x = xn;
    }
    return xbest;
}
PGI_DEV double pow_uint(double q, uint32_t k) {
    double r = 1.0, b = q;
    while (k) {
        if (k & 1u) r = r * b;
        b = b * b;
        k >>= 1;
    }
    return r;
}

// monomial index tables (lin [x,y,z,1] x lin -> quad; quad x lin -> cubic in
// Nister's column order x3 y3 x2y xy2 x2z x2 y2z y2 xyz xy | xz2 xz x yz2 yz y z3 z2 z 1)
__device__ constexpr int kQI[4][4] = {{0, 1, 2, 3}, {1, 4, 5, 6}, {2, 5, 7, 8}, {3, 6, 8, 9}};
__device__ constexpr int kCI[10][4] = {{0, 2, 4, 5},     {2, 3, 8, 9},     {4, 8, 10, 11},
                                       {5, 9, 11, 12},   {3, 1, 6, 7},     {8, 6, 13, 14},
                                       {9, 7, 14, 15},   {10, 13, 16, 17}, {11, 14, 17, 18},
                                       {12, 15, 18, 19}};
// phase-1 quads: Q0..Q5 = (E E^T)_{00,01,02,11,12,22}; Q6..Q8 = the 2x2 minors of rows 1,2.
// packed per term: entryA | entryB<<4 | sign<<8 (sign 1:+, 2:-, 0:unused)
// Looked up per LANE (index = sub-lane): a table in memory would be a vector-memory round trip in the middle of every
// pass (round 5: three dependent global_load + s_waitcnt vmcnt(0) for kQT, three more for kLAM, per pass), so the tables are
// packed into 64-bit immediates at compile time -- nibble s of kQTA[t] / kQTB[t] / kQTS[t] -- and read with a shift.
constexpr uint16_t kQT[9][3] = {
    {0x100 | 0 | (0 << 4), 0x100 | 1 | (1 << 4), 0x100 | 2 | (2 << 4)},
    {0x100 | 0 | (3 << 4), 0x100 | 1 | (4 << 4), 0x100 | 2 | (5 << 4)},
    {0x100 | 0 | (6 << 4), 0x100 | 1 | (7 << 4), 0x100 | 2 | (8 << 4)},
    {0x100 | 3 | (3 << 4), 0x100 | 4 | (4 << 4), 0x100 | 5 | (5 << 4)},
    {0x100 | 3 | (6 << 4), 0x100 | 4 | (7 << 4), 0x100 | 5 | (8 << 4)},
    {0x100 | 6 | (6 << 4), 0x100 | 7 | (7 << 4), 0x100 | 8 | (8 << 4)},
    {0x100 | 4 | (8 << 4), 0x200 | 5 | (7 << 4), 0},
    {0x100 | 3 | (8 << 4), 0x200 | 5 | (6 << 4), 0},
    {0x100 | 3 | (7 << 4), 0x200 | 4 | (6 << 4), 0},
};
constexpr uint64_t qt_pack(int t, int field) {  // field 0: entryA, 1: entryB, 2: sign
    uint64_t v = 0;
    for (int s = 0; s < 9; ++s) {
        const uint32_t w = kQT[s][t];
        const uint32_t f = field == 0 ? (w & 15u) : field == 1 ? ((w >> 4) & 15u) : (w >> 8);
        v |= (uint64_t)f << (4 * s);
    }
    return v;
}
__device__ constexpr uint64_t kQTA[3] = {qt_pack(0, 0), qt_pack(1, 0), qt_pack(2, 0)};
__device__ constexpr uint64_t kQTB[3] = {qt_pack(0, 1), qt_pack(1, 1), qt_pack(2, 1)};
__device__ constexpr uint64_t kQTS[3] = {qt_pack(0, 2), qt_pack(1, 2), qt_pack(2, 2)};
// Lambda index of (i, k): {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}}, nibble 3 i + k
constexpr uint64_t kLAMP = 0x0ull | (1ull << 4) | (2ull << 8) | (1ull << 12) | (3ull << 16) | (4ull << 20) | (2ull << 24) | (4ull << 28) | (5ull << 32);
PGI_DEV int lam_index(int i, int k) { return (int)((kLAMP >> (4 * (3 * i + k))) & 15ull); }

// ---- per-wavefront LDS scratch (doubles): [basis x4 | B(z) rows x4 | region A x4] ---------------
// group g of the wavefront uses basis[g], brow[g], rega[g].  Region A is reused in sequence:
// quads Q0..Q5 -> reduced rows (6x10) -> T[3][11] | poly[11] | rpoly[11] | brackets; the three
// minors Q6..Q8 sit in the (not yet used) B(z) area.  After the solve the model queue of the
// wavefront (40 x 9 floats) overlays the four region-A blocks.
constexpr int G_BASIS_SZ = 36, G_BROW_SZ = 39, G_REGA_SZ = 66;
constexpr int W_BASIS = 0, W_BROW = 4 * G_BASIS_SZ, W_REGA = W_BROW + 4 * G_BROW_SZ;
constexpr int W_DOUBLES = W_REGA + 4 * G_REGA_SZ;  // 564 doubles = 4512 B per wavefront
constexpr int A_T = 0, A_POLY = 33, A_RPOLY = 44, A_BRK = 56;  // offsets inside region A
struct GroupScratch {
    double* basis;  // [4][9]  X,Y,Z,W
    double* brow;   // [3][13]: bx[4] by[4] bc[5]   (minors Q6..Q8 during constraint building)
    double* rega;   // 66
};
PGI_DEV GroupScratch group_scratch(double* wave_base, int g) {
    return GroupScratch{wave_base + W_BASIS + g * G_BASIS_SZ, wave_base + W_BROW + g * G_BROW_SZ,
                        wave_base + W_REGA + g * G_REGA_SZ};
}

// Five epipolar rows -> orthonormal 4-vector null-space basis in gs[G_BASIS].
// Row r of the 5x9 system lives in sub-lane r.
PGI_DEV void nullspace5_group(const float4 pt, int s, int gbase, const GroupScratch gs) {
    double a[9];
    {
        const double x1 = pt.x, y1 = pt.y, x2 = pt.z, y2 = pt.w;
        a[0] = x2 * x1; a[1] = x2 * y1; a[2] = x2;
        a[3] = y2 * x1; a[4] = y2 * y1; a[5] = y2;
        a[6] = x1;      a[7] = y1;      a[8] = 1.0;
    }
    bool used = false;
    int mycol = -1;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int key = (s < 5 && !used) ? pivot_key(a[k], s) : -1;
        const int kmax = row_max_i(key);
        const bool is_p = (key == kmax) && (key >= 0);
        const int pl = gbase + (15 - (kmax & 15));
        if (is_p) {
            used = true;
            mycol = k;
        }
        const double scale = is_p ? 1.0 / a[k] : 1.0;  // x * 1.0 == x exactly: no per-column select
        const double f = is_p ? 0.0 : a[k];             // the pivot row runs the same update with factor 0
        switch (k) {  // (k is a compile-time constant in the unrolled loop; the column range must be one for the template)
            case 0: eliminate_columns<1, 9>(a, scale, f, pl << 2); break;
            case 1: eliminate_columns<2, 9>(a, scale, f, pl << 2); break;
            case 2: eliminate_columns<3, 9>(a, scale, f, pl << 2); break;
            case 3: eliminate_columns<4, 9>(a, scale, f, pl << 2); break;
            default: eliminate_columns<5, 9>(a, scale, f, pl << 2); break;
        }
    }
    // v_f[k] = -a[prow[k]][5+f]; v_f[5+g] = delta_fg
    if (mycol >= 0) {
#pragma unroll
        for (int f = 0; f < 4; ++f) gs.basis[9 * f + mycol] = -a[5 + f];
    }
    if (s < 4) {
#pragma unroll
        for (int g = 0; g < 4; ++g) gs.basis[9 * s + 5 + g] = (g == s) ? 1.0 : 0.0;
    }
    wave_sync();
    // modified Gram-Schmidt, element i in sub-lane i, tree dot products
    double v[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) v[f] = (s < 9) ? gs.basis[9 * f + s] : 0.0;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
#pragma unroll
        for (int g = 0; g < f; ++g) {
            const double d = row_sum_d(v[g] * v[f]);
            v[f] = fma(-d, v[g], v[f]);
        }
        const double nn = row_sum_d(v[f] * v[f]);
        const double inv = 1.0 / sqrt(nn);
        v[f] = v[f] * inv;
    }
    wave_sync();
    if (s < 9) {
#pragma unroll
        for (int f = 0; f < 4; ++f) gs.basis[9 * f + s] = v[f];
    }
    wave_sync();
}

// Debug taps of the back-end (five_point_batch only)
struct BackendDbg {
    double* cons;  // [10][20]
    double* red;   // [10][10] by pivot column
    double* poly;  // [11]
    double* roots; // [10]
    double* nroots;
};

// Nister back-end on the basis in gs[G_BASIS]: 10 cubic constraints -> Gauss-Jordan
// -> 3x3 polynomial matrix B(z) -> degree-10 determinant -> bracketed real roots ->
// E per root.  Returns true in sub-lanes that hold a valid model E32 (sub-lane =
// root index).  smp: the five sample points (orientation test) or nullptr.
// `sample(i)` returns sample row i (0..4) for the orientation test; CHECK = false skips the test.
template <bool DBG, int PB, bool CHECK, class SAMPLE>
PGI_DEV bool backend_group(const GroupScratch gs, int s, int gbase, SAMPLE sample, float E32[9],
                           const BackendDbg* dbg, Prof& prof) {
    // ---- phase 1: nine quadratic forms, one per sub-lane ------------------------------
    if (s < 9) {
        double q[10];
#pragma unroll
        for (int c = 0; c < 10; ++c) q[c] = 0.0;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int eA = (int)((kQTA[t] >> (4 * s)) & 15ull), eB = (int)((kQTB[t] >> (4 * s)) & 15ull),
                      sg = (int)((kQTS[t] >> (4 * s)) & 15ull);
            if (sg) {
                double A[4], B[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const double av = gs.basis[9 * b + eA];
                    A[b] = (sg == 1) ? av : -av;
                    B[b] = gs.basis[9 * b + eB];
                }
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) q[kQI[a][b]] = fma(A[a], B[b], q[kQI[a][b]]);
            }
        }
        double* qdst = (s < 6) ? gs.rega + 10 * s : gs.brow + 10 * (s - 6);
#pragma unroll
        for (int c = 0; c < 10; ++c) qdst[c] = q[c];
    }
    wave_sync();
    // Lambda_ii = (E E^T)_ii - tr/2, coefficient c in sub-lane c
    if (s < 10) {
        const double q0 = gs.rega[0 + s], q3 = gs.rega[30 + s], q5 = gs.rega[50 + s];
        const double tr = (q0 + q3) + q5;
        gs.rega[0 + s] = q0 - 0.5 * tr;
        gs.rega[30 + s] = q3 - 0.5 * tr;
        gs.rega[50 + s] = q5 - 0.5 * tr;
    }
    wave_sync();
    // ---- phase 2: ten cubic constraint rows, one per sub-lane ----------------------------
    double row[20];
#pragma unroll
    for (int m = 0; m < 20; ++m) row[m] = 0.0;
    if (s < 10) {
        const int i = (s > 0) ? (s - 1) / 3 : 0, j = (s > 0) ? (s - 1) % 3 : 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double* qsrc = (s == 0) ? gs.brow + 10 * k : gs.rega + 10 * lam_index(i, k);
            const int ei = (s == 0) ? k : 3 * k + j;
            const bool neg = (s == 0) && (k == 1);
            double L[4];
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                const double lv = gs.basis[9 * l + ei];
                L[l] = neg ? -lv : lv;
            }
#pragma unroll
            for (int q = 0; q < 10; ++q) {
                const double qv = qsrc[q];
#pragma unroll
                for (int l = 0; l < 4; ++l) row[kCI[q][l]] = fma(qv, L[l], row[kCI[q][l]]);
            }
        }
    }
    if constexpr (DBG) {
        if (s < 10)
            for (int m = 0; m < 20; ++m) dbg->cons[20 * s + m] = row[m];
    }
    prof.mark<PB + 0>();
    // ---- Gauss-Jordan, partial pivoting; row r in sub-lane r ---------------------------------
    bool used = false;
    int mycol = -1;
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        const int key = (s < 10 && !used) ? pivot_key(row[k], s) : -1;
        const int kmax = row_max_i(key);
        const bool is_p = (key == kmax) && (key >= 0);
        const int pl = gbase + (15 - (kmax & 15));
        if (is_p) {
            used = true;
            mycol = k;
        }
        const double scale = is_p ? 1.0 / row[k] : 1.0;  // x * 1.0 == x exactly: no per-column select
        const double f = is_p ? 0.0 : row[k];             // the pivot row runs the same update with factor 0
        switch (k) {  // (k is a compile-time constant in the unrolled loop)
            case 0: eliminate_columns<1, 20>(row, scale, f, pl << 2); break;
            case 1: eliminate_columns<2, 20>(row, scale, f, pl << 2); break;
            case 2: eliminate_columns<3, 20>(row, scale, f, pl << 2); break;
            case 3: eliminate_columns<4, 20>(row, scale, f, pl << 2); break;
            case 4: eliminate_columns<5, 20>(row, scale, f, pl << 2); break;
            case 5: eliminate_columns<6, 20>(row, scale, f, pl << 2); break;
            case 6: eliminate_columns<7, 20>(row, scale, f, pl << 2); break;
            case 7: eliminate_columns<8, 20>(row, scale, f, pl << 2); break;
            case 8: eliminate_columns<9, 20>(row, scale, f, pl << 2); break;
            default: eliminate_columns<10, 20>(row, scale, f, pl << 2); break;
        }
    }
    prof.mark<PB + 1>();
    wave_sync();  // quads are dead: region A is reused for the reduced rows
    if (mycol >= 4) {
#pragma unroll
        for (int m = 0; m < 10; ++m) gs.rega[10 * (mycol - 4) + m] = row[10 + m];
    }
    if constexpr (DBG) {
        if (mycol >= 0)
            for (int m = 0; m < 10; ++m) dbg->red[10 * mycol + m] = row[10 + m];
    }
    wave_sync();
    // ---- B(z): rows (e,f) = (x2z,x2), (y2z,y2), (xyz,xy) -> e - z f ----------------------------
    if (s < 3) {
        const double* e = gs.rega + 20 * s;
        const double* f = e + 10;
        double* b = gs.brow + 13 * s;
        b[0] = e[2]; b[1] = e[1] - f[2]; b[2] = e[0] - f[1]; b[3] = -f[0];
        b[4] = e[5]; b[5] = e[4] - f[5]; b[6] = e[3] - f[4]; b[7] = -f[3];
        b[8] = e[9]; b[9] = e[8] - f[9]; b[10] = e[7] - f[8]; b[11] = e[6] - f[7]; b[12] = -f[6];
    }
    wave_sync();
    // ---- det B(z) = sum_i (+,-,+) bc_i * (bx_r by_q - by_r bx_q) ------------------------------
    if (s < 3) {
        // the other two rows of B(z): {1, 2}, {0, 2}, {0, 1}
        const double* br = gs.brow + 13 * (s == 0 ? 1 : 0);
        const double* bq = gs.brow + 13 * (s == 2 ? 1 : 2);
        const double* bc = gs.brow + 13 * s + 8;
        double bxr[4], byr[4], bxq[4], byq[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            bxr[a] = br[a]; byr[a] = br[4 + a]; bxq[a] = bq[a]; byq[a] = bq[4 + a];
        }
        double m[7];
#pragma unroll
        for (int c = 0; c < 7; ++c) m[c] = 0.0;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) m[a + b] = fma(bxr[a], byq[b], m[a + b]);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) m[a + b] = fma(-byr[a], bxq[b], m[a + b]);
        double T[11];
#pragma unroll
        for (int c = 0; c < 11; ++c) T[c] = 0.0;
#pragma unroll
        for (int a = 0; a < 5; ++a) {
            const double bca = bc[a];
#pragma unroll
            for (int b = 0; b < 7; ++b) T[a + b] = fma(bca, m[b], T[a + b]);
        }
#pragma unroll
        for (int c = 0; c < 11; ++c) gs.rega[A_T + 11 * s + c] = T[c];  // red is dead (consumed above)
    }
    wave_sync();
    if (s < 11) {
        const double p = (gs.rega[A_T + s] - gs.rega[A_T + 11 + s]) + gs.rega[A_T + 22 + s];
        gs.rega[A_POLY + s] = p;
        gs.rega[A_RPOLY + (10 - s)] = p;
        if constexpr (DBG) dbg->poly[s] = p;
    }
    wave_sync();
    prof.mark<PB + 2>();
    // ---- bracket sign changes on the grid: sub-lane s owns intervals [16s, 16s+16) ---------------
    double p[11];
#pragma unroll
    for (int c = 0; c < 11; ++c) p[c] = gs.rega[A_POLY + c];
    uint32_t chg = 0, sgn = 0;
    {
        double gv[17];  // all 17 grid abscissae of this sub-lane up front: one constant-memory latency, not five
#pragma unroll
        for (int b = 0; b < 17; ++b) gv[b] = kGridTab.v[16 * s + b];
        sgn = horner10(p, gv[0]) < 0.0 ? 1u : 0u;
#pragma unroll
        for (int b = 0; b < 16; b += 4) {  // four independent Horner chains per step (latency)
            double v0 = p[10], v1 = p[10], v2 = p[10], v3 = p[10];
            const double g0 = gv[b + 1], g1 = gv[b + 2], g2 = gv[b + 3], g3 = gv[b + 4];
#pragma unroll
            for (int c = 9; c >= 0; --c) {
                v0 = fma(v0, g0, p[c]);
                v1 = fma(v1, g1, p[c]);
                v2 = fma(v2, g2, p[c]);
                v3 = fma(v3, g3, p[c]);
            }
            sgn |= (v0 < 0.0 ? 2u : 0u) << b;
            sgn |= (v1 < 0.0 ? 4u : 0u) << b;
            sgn |= (v2 < 0.0 ? 8u : 0u) << b;
            sgn |= (v3 < 0.0 ? 16u : 0u) << b;
        }
        chg = (sgn ^ (sgn >> 1)) & 0xFFFFu;  // bit b: sign change across interval 16 s + b
    }
    const int cnt = __popc(chg);
    const int incl = row_scan_incl_i(cnt);
    const int total = __shfl(incl, gbase + 15);
    int slot = incl - cnt;
    int* brk = reinterpret_cast<int*>(gs.rega + A_BRK);
    while (chg) {
        const int b = __ffs(chg) - 1;
        chg &= chg - 1;
        if (slot < kMaxModels) brk[slot] = (16 * s + b) | (int)(((sgn >> b) & 1u) << 16);
        ++slot;
    }
    wave_sync();
    const int nb = min(total, kMaxModels);
    if constexpr (DBG) {
        if (s == 0) dbg->nroots[0] = (double)nb;
    }
    prof.mark<PB + 3>();
    // ---- refine one root per sub-lane, back-substitute, build E -------------------------------------
    bool valid = false;
    if (s < nb) {
        const int w = brk[s];
        const int j = w & 0xFFFF;
        const bool sprev = (w >> 16) & 1;
        const double gprev = kGridTab.v[j], g = kGridTab.v[j + 1];
        // |z| > 1: solve for 1/z on the reversed polynomial (one coefficient array live either way)
        const bool tail = (gprev >= 1.0) || (g <= -1.0);
        const double* psrc = gs.rega + (tail ? A_RPOLY : A_POLY);
        double q[11];
#pragma unroll
        for (int c = 0; c < 11; ++c) q[c] = psrc[c];
        const double lo0 = tail ? 1.0 / g : gprev, hi0 = tail ? 1.0 / gprev : g;
        const double rt = refine_root(q, lo0, hi0, tail ? !sprev : sprev);
        const double z = tail ? 1.0 / rt : rt;
        if constexpr (DBG) dbg->roots[s] = z;
        prof.mark<PB + 4>();
        double rw[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double* b = gs.brow + 13 * i;
            rw[i][0] = fma(fma(fma(b[3], z, b[2]), z, b[1]), z, b[0]);
            rw[i][1] = fma(fma(fma(b[7], z, b[6]), z, b[5]), z, b[4]);
            rw[i][2] = fma(fma(fma(fma(b[12], z, b[11]), z, b[10]), z, b[9]), z, b[8]);
        }
        double c01[3], c02[3], c12[3], cb[3];
        cross3(rw[0], rw[1], c01);
        cross3(rw[0], rw[2], c02);
        cross3(rw[1], rw[2], c12);
        double wb = fabs(c01[2]);
        cb[0] = c01[0]; cb[1] = c01[1]; cb[2] = c01[2];
        const bool t02 = fabs(c02[2]) > wb;
        wb = t02 ? fabs(c02[2]) : wb;
        sel3(t02, c02, cb);
        const bool t12 = fabs(c12[2]) > wb;
        sel3(t12, c12, cb);
        const double x = cb[0] / cb[2], y = cb[1] / cb[2];
        double E[9], n2 = 0.0;
#pragma unroll
        for (int m = 0; m < 9; ++m) {
            E[m] = fma(x, gs.basis[m], fma(y, gs.basis[9 + m], fma(z, gs.basis[18 + m], gs.basis[27 + m])));
            n2 = fma(E[m], E[m], n2);
        }
        valid = (n2 > 0.0) && (n2 < 1.0e300);
        const double inv = 1.0 / sqrt(n2);
#pragma unroll
        for (int m = 0; m < 9; ++m) E[m] = E[m] * inv;
        if (CHECK && valid) {  // oriented epipolar constraint on the minimal sample
            const double c0[3] = {E[0], E[3], E[6]}, c1[3] = {E[1], E[4], E[7]}, c2[3] = {E[2], E[5], E[8]};
            double e01[3], e02[3], e12[3], ep[3];
            cross3(c0, c1, e01);
            cross3(c0, c2, e02);
            cross3(c1, c2, e12);
            const double n01 = fma(e01[0], e01[0], fma(e01[1], e01[1], e01[2] * e01[2]));
            const double n02 = fma(e02[0], e02[0], fma(e02[1], e02[1], e02[2] * e02[2]));
            const double n12 = fma(e12[0], e12[0], fma(e12[1], e12[1], e12[2] * e12[2]));
            double nbst = n01;
            ep[0] = e01[0]; ep[1] = e01[1]; ep[2] = e01[2];
            const bool u02 = n02 > nbst;
            nbst = u02 ? n02 : nbst;
            sel3(u02, e02, ep);
            const bool u12 = n12 > nbst;
            sel3(u12, e12, ep);
            int npos = 0, nneg = 0;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const float4 sp = sample(i);
                const double x1 = sp.x, y1 = sp.y, x2 = sp.z, y2 = sp.w;
                const double l0 = fma(E[0], x1, fma(E[1], y1, E[2]));
                const double l1 = fma(E[3], x1, fma(E[4], y1, E[5]));
                const double l2 = fma(E[6], x1, fma(E[7], y1, E[8]));
                const double cx = fma(ep[1], 1.0, -(ep[2] * y2));
                const double cy = fma(ep[2], x2, -(ep[0] * 1.0));
                const double cz = fma(ep[0], y2, -(ep[1] * x2));
                const double sg = fma(cx, l0, fma(cy, l1, cz * l2));
                npos += sg > 0.0;
                nneg += sg < 0.0;
            }
            valid = (npos == 5) || (nneg == 5);
        }
#pragma unroll
        for (int m = 0; m < 9; ++m) E32[m] = (float)E[m];
    }
    prof.mark<PB + 5>();
    return valid;
}

// ---- 3x3 SVD (one-sided Jacobi) and the four-candidate decomposition ----------------------------------
// pose_utils.h:144-169: R1 = U D V^T, R2 = U D^T V^T, t = U[:,2]; det fixes as :157-163.
// Executed by ONE wavefront with the matrices spread over lanes 0..2: lane l owns row l of G (= E V) and row l of V,
// three doubles each.  A rotation needs two whole columns; they travel by v_readlane (constant source lanes, results
// wave-uniform), every lane derives the same (c, s) and updates its own row.  Element for element the operations and
// their order are those of the oracle's scalar pgo_svd3 / pgo_decompose, so the bits are the same -- but the live set
// is ~50 VGPRs instead of ~130 (the scalar form, run by one thread inside K1's epilogue, was the source of r02's
// 553 spilled VGPRs), and the row updates cost a third of the instructions.
PGI_DEV double readlane_d(double v, int src) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
PGI_DEV double pick3(int o, const double a[3]) { return o == 0 ? a[0] : (o == 1 ? a[1] : a[2]); }
// E32: the model (9 floats, LDS or global); Rt (LDS or global, 21 doubles): R1[9] R2[9] t[3]; candidates
// 0:(R1,+t) 1:(R1,-t) 2:(R2,+t) 3:(R2,-t)  (pose_utils.h:182,201).  Call with the whole wavefront; no barrier inside.
template <class T>
PGI_DEV void decompose_wave(const T* Esrc, double* Rt, int lane) {
    const int l = lane < 3 ? lane : 0;  // lanes >= 3 shadow lane 0 (never read, never written out)
    double g[3], v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        g[c] = (double)Esrc[3 * l + c];
        v[c] = (c == l) ? 1.0 : 0.0;
    }
    for (int sw = 0; sw < kSvdSweeps; ++sw) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int p = (k == 2) ? 1 : 0, q = (k == 0) ? 1 : 2;
            const double p0 = readlane_d(g[p], 0), p1 = readlane_d(g[p], 1), p2 = readlane_d(g[p], 2);
            const double q0 = readlane_d(g[q], 0), q1 = readlane_d(g[q], 1), q2 = readlane_d(g[q], 2);
            const double al = fma(p0, p0, fma(p1, p1, p2 * p2));
            const double be = fma(q0, q0, fma(q1, q1, q2 * q2));
            const double ga = fma(p0, q0, fma(p1, q1, p2 * q2));
            if (ga == 0.0) continue;  // wave-uniform
            // one division + two square roots per rotation (as in the 9x9 Jacobi)
            const double da = be - al, db = 2.0 * ga;
            const double h = sqrt(fma(da, da, db * db));
            const double d = fabs(da) + h;
            const double r = sqrt(fma(d, d, db * db));
            const double inv = 1.0 / r;
            const double c = d * inv, s = (da >= 0.0 ? db : -db) * inv;
            const double gp = g[p], gq = g[q];
            g[p] = fma(c, gp, -(s * gq));
            g[q] = fma(s, gp, c * gq);
            const double vp = v[p], vq = v[q];
            v[p] = fma(c, vp, -(s * vq));
            v[q] = fma(s, vp, c * vq);
        }
    }
    double sg[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const double c0 = readlane_d(g[j], 0), c1 = readlane_d(g[j], 1), c2 = readlane_d(g[j], 2);
        sg[j] = sqrt(fma(c0, c0, fma(c1, c1, c2 * c2)));
    }
    // stable selection sort (descending) as three compare-exchanges on (a,b) = (0,1),(0,2),(1,2)
    int o0 = 0, o1 = 1, o2 = 2;
    double s0 = sg[0], s1 = sg[1], s2 = sg[2];
    if (s1 > s0) { double ts = s0; s0 = s1; s1 = ts; int to = o0; o0 = o1; o1 = to; }
    if (s2 > s0) { double ts = s0; s0 = s2; s2 = ts; int to = o0; o0 = o2; o2 = to; }
    if (s2 > s1) { double ts = s1; s1 = s2; s2 = ts; int to = o1; o1 = o2; o2 = to; }
    const double gs0 = pick3(o0, g), gs1 = pick3(o1, g);
    double vs[3] = {pick3(o0, v), pick3(o1, v), pick3(o2, v)};
    const double i0 = 1.0 / s0, i1 = 1.0 / s1;
    const double ul0 = gs0 * i0, ul1 = gs1 * i1;  // this lane's row of U, columns 0 and 1
    const double u0[3] = {readlane_d(ul0, 0), readlane_d(ul0, 1), readlane_d(ul0, 2)};
    const double u1[3] = {readlane_d(ul1, 0), readlane_d(ul1, 1), readlane_d(ul1, 2)};
    double u2[3];
    cross3(u0, u1, u2);  // third column of U, whole in every lane
    const double v0[3] = {readlane_d(vs[0], 0), readlane_d(vs[0], 1), readlane_d(vs[0], 2)};
    const double v1[3] = {readlane_d(vs[1], 0), readlane_d(vs[1], 1), readlane_d(vs[1], 2)};
    double v2[3] = {readlane_d(vs[2], 0), readlane_d(vs[2], 1), readlane_d(vs[2], 2)};
    double vc[3];
    cross3(v0, v1, vc);
    const double dv = fma(vc[0], v2[0], fma(vc[1], v2[1], vc[2] * v2[2]));
    if (dv < 0.0) { v2[0] = -v2[0]; v2[1] = -v2[1]; v2[2] = -v2[2]; }
    const double ul2 = pick3(l, u2);
    double t[3] = {u2[0], u2[1], u2[2]};
    const double tn = 1.0 / sqrt(fma(t[0], t[0], fma(t[1], t[1], t[2] * t[2])));
    t[0] = t[0] * tn; t[1] = t[1] * tn; t[2] = t[2] * tn;
    if (lane < 3) {  // row l of R1 = U D V^T and of R2 = U D^T V^T
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double a = ul1 * v0[j];
            const double b = ul0 * v1[j];
            const double c = ul2 * v2[j];
            Rt[3 * l + j] = (b - a) + c;
            Rt[9 + 3 * l + j] = (a - b) + c;
        }
        Rt[18 + l] = pick3(l, t);
    }
}

// depth signs of lambda2*x2 = lambda1*R*x1 + t: bit0 = (R,+t) in front of both, bit1 = (R,-t)
PGI_DEV uint32_t cheirality_bits(const double R[9], const double t[3], const double X1[3],
                                 const double X2[3], const double x2t[3]) {
    double a[3], nn[3], at[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) a[i] = fma(R[3 * i], X1[0], fma(R[3 * i + 1], X1[1], R[3 * i + 2]));
    cross3(a, X2, nn);
    cross3(a, t, at);
    const double d1 = fma(x2t[0], nn[0], fma(x2t[1], nn[1], x2t[2] * nn[2]));
    const double d2 = fma(at[0], nn[0], fma(at[1], nn[1], at[2] * nn[2]));
    return ((d1 > 0.0) && (d2 > 0.0) ? 1u : 0u) | ((d1 < 0.0) && (d2 < 0.0) ? 2u : 0u);
}

}  // namespace pgi
