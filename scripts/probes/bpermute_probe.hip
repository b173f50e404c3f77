// Cost of moving a source descriptor between lanes with ds_bpermute_b32 inside the guided scan's summation step (round 5):
// the 128-element step of guided_scan_flat_kernel (two rows from LDS, two sequential f64 chains) with the descriptor
//   MODE 0: in the lane's own registers (what guided_scan_tile_kernel does),
//   MODE 1: pulled from another lane with 128 ds_bpermute_b32,
//   MODE 2: read from a second LDS array with ds_read_b128 (rows 528 bytes apart).
// hipcc --offload-arch=gfx950 -O3 -o /tmp/bpermute_probe scripts/probes/bpermute_probe.hip && /tmp/bpermute_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
constexpr int kD = 128;
template <int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void step(const float* d1, const uint32_t* items, double* out, uint32_t B) {
    __shared__ __attribute__((aligned(16))) float tile[16][kD + 4];
    __shared__ __attribute__((aligned(16))) float srow[MODE == 2 ? 64 : 1][kD + 4];
    __shared__ double dist[16][64];
    const uint32_t lane = threadIdx.x;
    float a[kD];
    {
        const float4* row = reinterpret_cast<const float4*>(d1 + (size_t)((blockIdx.x * 64 + lane) % 4096) * kD);
#pragma unroll
        for (int q = 0; q < kD / 4; ++q) { const float4 t = row[q]; a[4*q] = t.x; a[4*q+1] = t.y; a[4*q+2] = t.z; a[4*q+3] = t.w; }
    }
    for (uint32_t k = lane; k < 16 * (kD + 4); k += 64) (&tile[0][0])[k] = d1[k];
    if (MODE == 2) {
#pragma unroll
        for (int q = 0; q < kD; ++q) srow[lane][q] = a[q];
    }
    __syncthreads();
    for (uint32_t base = 0; base < B; base += 64u) {
        const uint32_t item = items[(base + lane) % 4096];
        const uint32_t owner = item & 63u;
        const uint32_t ca = (item >> 6) & 15u, cb = (item >> 15) & 15u;
        const float4* ta = reinterpret_cast<const float4*>(&tile[(item >> 11) & 15u][0]);
        const float4* tb = reinterpret_cast<const float4*>(&tile[(item >> 20) & 15u][0]);
        const float4* sa = reinterpret_cast<const float4*>(&srow[MODE == 2 ? owner : 0][0]);
        const int from = (int)(owner << 2);
        double da = 0.0, db = 0.0;
        typedef float pk2 __attribute__((ext_vector_type(2)));
        auto pull4 = [&](int q, float& s0, float& s1, float& s2, float& s3) {
            if (MODE == 1) {
                s0 = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(a[4 * q])));
                s1 = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(a[4 * q + 1])));
                s2 = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(a[4 * q + 2])));
                s3 = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(a[4 * q + 3])));
            } else if (MODE == 2) {
                const float4 t = sa[q];
                s0 = t.x; s1 = t.y; s2 = t.z; s3 = t.w;
            } else {
                s0 = a[4 * q]; s1 = a[4 * q + 1]; s2 = a[4 * q + 2]; s3 = a[4 * q + 3];
            }
        };
        float4 u = ta[0], v = tb[0];
        float s0, s1, s2, s3;
        pull4(0, s0, s1, s2, s3);
#pragma unroll
        for (int q = 0; q < kD / 4; ++q) {
            float4 un = u, vn = v;
            float n0 = s0, n1 = s1, n2 = s2, n3 = s3;
            if (q + 1 < kD / 4) { un = ta[q + 1]; vn = tb[q + 1]; pull4(q + 1, n0, n1, n2, n3); }
            const pk2 a01 = {s0, s1}, a23 = {s2, s3};
            const pk2 du01 = a01 - pk2{u.x, u.y}, du23 = a23 - pk2{u.z, u.w}, dv01 = a01 - pk2{v.x, v.y}, dv23 = a23 - pk2{v.z, v.w};
            const double u0 = (double)du01.x, u1 = (double)du01.y, u2 = (double)du23.x, u3 = (double)du23.y;
            const double w0 = (double)dv01.x, w1 = (double)dv01.y, w2 = (double)dv23.x, w3 = (double)dv23.y;
            da = fma(u0, u0, da); db = fma(w0, w0, db);
            da = fma(u1, u1, da); db = fma(w1, w1, db);
            da = fma(u2, u2, da); db = fma(w2, w2, db);
            da = fma(u3, u3, da); db = fma(w3, w3, db);
            u = un; v = vn; s0 = n0; s1 = n1; s2 = n2; s3 = n3;
            asm volatile("" : "+v"(da), "+v"(db));
        }
        dist[ca][owner] = da; dist[cb][owner] = db;
    }
    __syncthreads();
    out[(size_t)blockIdx.x * 64 + lane] = dist[lane % 16][lane];
}
template <int MODE> static void run(const char* name, const float* d1, const uint32_t* items, double* out, int blocks, uint32_t B) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(step<MODE>, dim3(blocks), dim3(64), 0, 0, d1, items, out, B);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(step<MODE>, dim3(blocks), dim3(64), 0, 0, d1, items, out, B);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double steps = (double)blocks * (B / 64);
    printf("%-28s %8.3f ms: %.0f cycles of a CU per 64-item step at 2.4 GHz (256 CUs), %.2f G element pairs/s\n", name, ms,
           ms * 1e-3 * 2.4e9 * 256 / steps, steps * 64 * 2 * 128 / ms / 1e6);
}
int main() {
    std::vector<float> h(4096 * kD); for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) * 0.001f;
    std::vector<uint32_t> it(4096);
    for (int i = 0; i < 4096; ++i) { uint32_t r = (uint32_t)i * 2246822519u; r ^= r >> 15;
        const uint32_t owner = (i % 64 / 6 * 6 + (r % 13)) % 64;  // owners near the lane, a dozen lanes share a row
        it[i] = owner | ((r >> 4) % 16) << 6 | ((i % 64) / 12 % 16) << 11 | ((r >> 9) % 16) << 15 | (((i % 64) / 12 + 1) % 16) << 20; }
    float* d1; uint32_t* di; double* out; const int blocks = 256 * 7 * 4;
    hipMalloc(&d1, h.size() * 4); hipMalloc(&di, it.size() * 4); hipMalloc(&out, (size_t)blocks * 64 * 8);
    hipMemcpy(d1, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(di, it.data(), it.size() * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("own registers", d1, di, out, blocks, 64 * 32);
        run<1>("ds_bpermute_b32 x 128", d1, di, out, blocks, 64 * 32);
        run<2>("second LDS array (b128)", d1, di, out, blocks, 64 * 32);
    }
    return 0;
}
