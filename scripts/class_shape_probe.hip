// class_shape_probe.hip -- which property of the one-shot tile kernel keeps it clear of the write classes: the tile's size or the
// number of stores a wave issues?  N hipMalloc'ed 12 GiB buffers; per buffer (GB/s, best of 3):
//   gs2048          resident grid-stride loop (the class detector)
//   T x S           workgroups of T threads, every thread S stores of 16 bytes, tile = T * S * 16 bytes, one tile per workgroup, no loop
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o scripts/class_shape_probe scripts/class_shape_probe.hip ; scripts/class_shape_probe [buffers=8]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k_gs(f4 *dst, size_t n)
{
    f4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store(v, &dst[i]);
}
template <int T, int S> __global__ __launch_bounds__(T) void k_tile(f4 *dst, size_t n)
{
    f4 v = {1.f, 2.f, 3.f, 4.f};
    const size_t base = (size_t)blockIdx.x * (T * S) + threadIdx.x;
#pragma unroll
    for (int j = 0; j < S; j++) {
        const size_t i = base + (size_t)j * T;
        if (i < n) dst[i] = v;
    }
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
    const int nb = argc > 1 ? atoi(argv[1]) : 8;
    const size_t bytes = 12ull << 30, n = bytes / 16;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    std::vector<f4 *> bufs;
    for (int k = 0; k < nb; k++) {
        f4 *p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) break;
        bufs.push_back(p);
    }
    auto rate = [&](float ms) { return bytes / 1e9 / (ms / 1e3); };
#define TILE(T, S) rate(best_ms([&] { hipLaunchKernelGGL((k_tile<T, S>), dim3((unsigned)((n + T * S - 1) / (T * S))), dim3(T), 0, 0, bufs[k], n); }))
    printf("GB/s       gs2048 |  64x1  256x1  512x1 1024x1 | 128x2  256x2   64x4  64x16 | 64x2(2K)\n");
    for (size_t k = 0; k < bufs.size(); k++) {
        const double g = rate(best_ms([&] { hipLaunchKernelGGL(k_gs, dim3(2048), dim3(256), 0, 0, bufs[k], n); }));
        const double a = TILE(64, 1), b = TILE(256, 1), c = TILE(512, 1), d = TILE(1024, 1);
        const double e = TILE(128, 2), f = TILE(256, 2), h = TILE(64, 4), i2 = TILE(64, 16), j2 = TILE(64, 2);
        printf("buffer %zu: %6.0f | %6.0f %6.0f %6.0f %6.0f | %6.0f %6.0f %6.0f %6.0f | %6.0f\n", k, g, a, b, c, d, e, f, h, i2, j2);
        fflush(stdout);
    }
    return 0;
}
