// Where does the time of the FIRST upload out of a freshly allocated page-locked block go (33 MB took 8.6 ms inside
// PoseGraphBuilder::estimatePoses)?  Times allocation, host fill, first and second hipMemcpy, and a kernel reading the block.
//   hipcc --offload-arch=gfx950 -O2 -o build/first_copy_probe scripts/probes/first_copy_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
typedef std::chrono::steady_clock Clock;
static double ms_since(Clock::time_point t) { return 1e3 * std::chrono::duration<double>(Clock::now() - t).count(); }
__global__ void sum_kernel(const float* p, size_t n, float* out) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 123.25f) *out = acc;
}
int main() {
    const size_t bytes = 33u << 20;
    void* warm_h; void* warm_d;
    (void)hipHostMalloc(&warm_h, 1 << 20, hipHostMallocDefault);
    (void)hipMalloc(&warm_d, 1 << 20);
    (void)hipMemcpy(warm_d, warm_h, 1 << 20, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(sum_kernel, dim3(64), dim3(256), 0, 0, (const float*)warm_d, (size_t)1024, (float*)warm_d);
    (void)hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        void *h = nullptr, *d = nullptr;
        Clock::time_point t = Clock::now();
        (void)hipHostMalloc(&h, bytes, hipHostMallocDefault);
        const double t_hm = ms_since(t); t = Clock::now();
        memset(h, 1, bytes);
        const double t_fill = ms_since(t); t = Clock::now();
        (void)hipMalloc(&d, bytes);
        const double t_dm = ms_since(t); t = Clock::now();
        (void)hipMemcpy(d, h, bytes, hipMemcpyHostToDevice);
        const double t_c1 = ms_since(t); t = Clock::now();
        (void)hipMemcpy(d, h, bytes, hipMemcpyHostToDevice);
        const double t_c2 = ms_since(t); t = Clock::now();
        hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        t = Clock::now();
        (void)hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s);
        (void)hipStreamSynchronize(s);
        const double t_c3 = ms_since(t); t = Clock::now();
        hipLaunchKernelGGL(sum_kernel, dim3(2048), dim3(256), 0, s, (const float*)h, bytes / 4, (float*)d);
        (void)hipStreamSynchronize(s);
        const double t_k = ms_since(t);
        printf("rep %d: hipHostMalloc %.2f  fill %.2f  hipMalloc %.2f  hipMemcpy #1 %.2f  #2 %.2f  async on a stream %.2f  kernel reads host block %.2f ms\n",
               rep, t_hm, t_fill, t_dm, t_c1, t_c2, t_c3, t_k);
        (void)hipStreamDestroy(s);
        (void)hipFree(d); (void)hipHostFree(h);
    }
    return 0;
}
