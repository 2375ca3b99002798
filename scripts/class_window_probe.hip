// class_window_probe.hip -- how far may a kernel's stores stray from one in-order front before a buffer's write class shows?
// One 4 KiB tile per workgroup, no loop (the pattern that writes 6.9-7.0 TB/s into every buffer, class_pattern_probe.hip), but
// the tile a workgroup writes is permuted inside windows of W consecutive tiles (stride permutation, odd multiplier):
// W = 1 is launch order; larger W scatters the in-flight stores over W x 4 KiB.
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o scripts/class_window_probe scripts/class_window_probe.hip ; scripts/class_window_probe [buffers=8]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_tile_perm(f4 *dst, size_t n_tiles, unsigned W, unsigned mult)
{
    f4 v = {1.f, 2.f, 3.f, 4.f};
    const size_t b = blockIdx.x;
    const size_t win = b / W * W;
    const size_t w_here = std::min<size_t>(W, n_tiles - win);
    size_t t = b;
    if (w_here == W) t = win + ((b - win) * (size_t)mult) % W;  // W a power of two, mult odd: a permutation of the window
    dst[t * 256 + threadIdx.x] = v;
}
__global__ void k_gs(f4 *dst, size_t n)
{
    f4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store(v, &dst[i]);
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
    const size_t bytes = 12ull << 30, n_tiles = bytes / 4096;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    std::vector<f4 *> bufs;
    for (int k = 0; k < nb; k++) {
        f4 *p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) break;
        bufs.push_back(p);
    }
    const unsigned Ws[] = {1, 512, 32768, 65536, 131072, 262144, 524288, 1048576};
    printf("GB/s, window (tiles of 4 KiB):");
    for (unsigned W : Ws) printf(" %7u", W);
    printf(" | gs2048 (the resident loop that shows the classes)\n");
    for (size_t k = 0; k < bufs.size(); k++) {
        printf("buffer %zu:                     ", k);
        for (unsigned W : Ws) {
            const unsigned mult = W > 1 ? (W / 2 + 1) | 1u : 1u;  // about half the window between consecutive workgroups
            const float ms = best_ms([&] { hipLaunchKernelGGL(k_tile_perm, dim3((unsigned)n_tiles), dim3(256), 0, 0, bufs[k], n_tiles, W, mult); });
            printf(" %7.0f", bytes / 1e9 / (ms / 1e3));
        }
        {
            const float ms = best_ms([&] { hipLaunchKernelGGL(k_gs, dim3(2048), dim3(256), 0, 0, bufs[k], bytes / 16); });
            printf(" | %7.0f", bytes / 1e9 / (ms / 1e3));
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
