// How fast can a kernel read page-locked host memory over PCIe (zero copy) compared with hipMemcpyAsync?
// K1 reads every row once while staging it into LDS, so a host-resident batch could be consumed in place.
//   hipcc --offload-arch=gfx950 -O2 -o build/zero_copy_probe scripts/probes/zero_copy_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// one workgroup reads `rows` floats from each of four arrays (the pair's rows), like K1's staging loop
__global__ void __launch_bounds__(256) stage_kernel(const float* x1, const float* y1, const float* x2, const float* y2, uint32_t rows, float* out, int spin) {
    __shared__ float4 lds[2048];
    const size_t o = (size_t)blockIdx.x * rows;
    for (uint32_t i = threadIdx.x; i < rows; i += 256) lds[i & 2047] = make_float4(x1[o + i], y1[o + i], x2[o + i], y2[o + i]);
    __syncthreads();
    float acc = 0.f;
    for (int k = 0; k < spin; ++k)  // stands in for the fit: keeps the workgroup resident without touching memory
        for (uint32_t i = threadIdx.x; i < 2048; i += 256) acc = fmaf(lds[(i + k) & 2047].x, 1.0001f, acc);
    if (acc == 123.456f) out[blockIdx.x] = acc;
}

int main(int argc, char** argv) {
    const uint32_t pairs = 10000, rows = 2000;
    const size_t n = (size_t)pairs * rows;
    float *h[4], *d[4], *out;
    for (int k = 0; k < 4; ++k) {
        CK(hipHostMalloc((void**)&h[k], n * 4, hipHostMallocDefault));
        for (size_t i = 0; i < n; ++i) h[k][i] = (float)(i & 1023);
        CK(hipMalloc((void**)&d[k], n * 4));
    }
    CK(hipMalloc((void**)&out, pairs * 4));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time_ms = [&](auto&& fn) {
        float best = 1e9f;
        for (int r = 0; r < 4; ++r) {
            (void)hipEventRecord(e0, s);
            fn();
            (void)hipEventRecord(e1, s);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        return best;
    };
    const double mb = 4.0 * n * 4 / 1e6;
    float t = time_ms([&] { for (int k = 0; k < 4; ++k) (void)hipMemcpyAsync(d[k], h[k], n * 4, hipMemcpyHostToDevice, s); });
    printf("hipMemcpyAsync x4          : %7.3f ms  %6.1f GB/s\n", t, mb / t);
    for (int spin : {0, 200, 800, 1600}) {
        t = time_ms([&] { hipLaunchKernelGGL(stage_kernel, dim3(pairs), dim3(256), 0, s, d[0], d[1], d[2], d[3], rows, out, spin); });
        const float td = t;
        t = time_ms([&] { hipLaunchKernelGGL(stage_kernel, dim3(pairs), dim3(256), 0, s, h[0], h[1], h[2], h[3], rows, out, spin); });
        printf("spin %4d: rows from HBM %7.3f ms | rows from host memory %7.3f ms  %6.1f GB/s\n", spin, td, t, mb / t);
    }
    return 0;
}
