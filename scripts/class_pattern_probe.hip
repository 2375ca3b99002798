// class_pattern_probe.hip -- which WRITE PATTERNS see a buffer's class?  torch's fill_ runs at 6.8 TB/s on every buffer on which
// the emit kernel varies by 17 % (scripts/class_emit_vs_fill.py).  N hipMalloc'ed 12 GiB buffers; per buffer, best of 3:
//   gs8192    grid-stride, 8192 resident blocks x 256 threads, 16-byte non-temporal stores (the library's k_pool_stream)
//   gs2048    the same with 2048 blocks
//   tileT     ONE tile of T bytes per block, no loop, blocks in launch order (what an elementwise torch kernel does: 8 KB),
//             plain or non-temporal stores
//   hipcc --offload-arch=gfx950 -O3 -o scripts/class_pattern_probe scripts/class_pattern_probe.hip ; scripts/class_pattern_probe [buffers=8]
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
template <int PER, bool NT> __global__ __launch_bounds__(256) void k_tile(f4 *dst, size_t n)
{
    f4 v = {1.f, 2.f, 3.f, 4.f};
    const size_t base = (size_t)blockIdx.x * (256 * PER) + threadIdx.x;
#pragma unroll
    for (int j = 0; j < PER; j++) {
        const size_t i = base + (size_t)j * 256;
        if (i < n) {
            if (NT) __builtin_nontemporal_store(v, &dst[i]); else dst[i] = v;
        }
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
#define TILE(PER, NT) rate(best_ms([&] { hipLaunchKernelGGL((k_tile<PER, NT>), dim3((unsigned)((n + 256 * PER - 1) / (256 * PER))), dim3(256), 0, 0, bufs[k], n); }))
    printf("GB/s      gs8192  gs2048 | tile4K  tile8K tile8Knt tile16K tile32K tile64K\n");
    for (int rnd = 0; rnd < 2; rnd++)
        for (size_t k = 0; k < bufs.size(); k++) {
            const double a = rate(best_ms([&] { hipLaunchKernelGGL(k_gs, dim3(8192), dim3(256), 0, 0, bufs[k], n); }));
            const double b = rate(best_ms([&] { hipLaunchKernelGGL(k_gs, dim3(2048), dim3(256), 0, 0, bufs[k], n); }));
            const double t1 = TILE(1, false), t2 = TILE(2, false), t2n = TILE(2, true), t4 = TILE(4, false), t8 = TILE(8, false), t16 = TILE(16, false);
            printf("buffer %zu: %6.0f  %6.0f | %6.0f  %6.0f  %6.0f  %6.0f  %6.0f  %6.0f\n", k, a, b, t1, t2, t2n, t4, t8, t16);
            fflush(stdout);
        }
    return 0;
}
