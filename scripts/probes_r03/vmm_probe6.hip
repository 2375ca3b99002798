// vmm_probe6.hip -- is a buffer's class a TRANSLATION property?  For each of N hipMalloc'ed 12 GiB buffers: streaming write rate
// (the class), then the rate of uniformly random 16-byte reads over the whole buffer (every access lands on another page:
// if slow buffers are mapped with small page-table fragments, this differs by integer factors), then the same random
// reads confined to 64 MiB (translation-insensitive control).
//   hipcc --offload-arch=gfx950 -O3 -o scripts/vmm_probe6 scripts/vmm_probe6.hip ; scripts/vmm_probe6 [buffers=6]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k_write(f4 *dst, size_t n)
{
    f4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store(v, &dst[i]);
}
__global__ void k_random_read(const f4 *src, size_t n_slots, size_t per_thread, float *sink)
{
    size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    unsigned long long x = t * 0x9E3779B97F4A7C15ull + 12345;
    f4 acc = {0, 0, 0, 0};
    for (size_t k = 0; k < per_thread; k++) {
        x ^= x << 13;
        x ^= x >> 7;
        x ^= x << 17;
        acc += src[x % n_slots];
    }
    if (acc.x + acc.y == 12345.678f) *sink = acc.x;
}
static hipEvent_t e0, e1;
template <class F> static float best_ms(F launch, int reps = 3)
{
    float best = 1e9f;
    for (int r = 0; r < reps; r++) {
        hipEventRecord(e0, 0);
        launch();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
    }
    return best;
}
int main(int argc, char **argv)
{
    const int nb = argc > 1 ? atoi(argv[1]) : 6;
    const size_t bytes = 12ull << 30, n = bytes / 16;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float *sink;
    hipMalloc(&sink, 4);
    std::vector<f4 *> bufs;
    for (int k = 0; k < nb; k++) {
        f4 *p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) break;
        bufs.push_back(p);
    }
    const size_t per = 64, threads = 2048ull * 256;  // 3.4e7 random reads per launch
    for (size_t k = 0; k < bufs.size(); k++) {
        const float w = best_ms([&] { hipLaunchKernelGGL(k_write, dim3(8192), dim3(256), 0, 0, bufs[k], n); });
        const float r_all = best_ms([&] { hipLaunchKernelGGL(k_random_read, dim3(2048), dim3(256), 0, 0, bufs[k], n, per, sink); });
        const float r_loc = best_ms([&] { hipLaunchKernelGGL(k_random_read, dim3(2048), dim3(256), 0, 0, bufs[k], (size_t)(64ull << 20) / 16, per, sink); });
        printf("buffer %zu at %p: stream write %.0f GB/s | random 16-B reads over 12 GiB %.2f G/s | over 64 MiB %.2f G/s\n", k, (void *)bufs[k],
               bytes / 1e9 / (w / 1e3), threads * per / 1e9 / (r_all / 1e3), threads * per / 1e9 / (r_loc / 1e3));
    }
    return 0;
}
