// bw_calib.hip -- what this box's HBM sustains for the access mixes the emit kernel produces (hand-written
// streaming kernels, no library): pure 16-byte writes (plain / non-temporal), pure reads, copy, and a
// 72 % write / 28 % read mix (12 GB written + 4.8 GB read per launch at config 3).
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/bw_calib scripts/bw_calib.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <bool NT> __global__ void k_write(f4 *dst, size_t n)
{
    f4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (NT) __builtin_nontemporal_store(v, &dst[i]); else dst[i] = v;
    }
}
__global__ void k_read(const f4 *src, size_t n, float *sink)
{
    f4 acc = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += src[i];
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;
}
template <bool NT> __global__ void k_copy(const f4 *src, f4 *dst, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        f4 v = src[i];
        if (NT) __builtin_nontemporal_store(v, &dst[i]); else dst[i] = v;
    }
}
// WR writes per RD reads, interleaved per thread (5:2 = 71 % writes)
template <int WR, int RD> __global__ void k_mix(const f4 *src, f4 *dst, size_t n_iter, float *sink)
{
    f4 acc = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (size_t it = 0; it < n_iter; it++) {
#pragma unroll
        for (int r = 0; r < RD; r++) acc += src[(it * RD + r) * stride + t];
#pragma unroll
        for (int w = 0; w < WR; w++) __builtin_nontemporal_store(acc, &dst[(it * WR + w) * stride + t]);
    }
    if (acc.x == 12345.678f) *sink = acc.x;
}

int main()
{
    const size_t bytes = (size_t)6 << 30;  // 6 GiB per buffer: far beyond the 256 MiB Infinity Cache
    const size_t n = bytes / 16;
    f4 *a, *b;
    float *sink;
    CK(hipMalloc(&a, bytes));
    CK(hipMalloc(&b, bytes));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 0, bytes));
    CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int grids[] = {256 * 8, 256 * 16, 256 * 32};
    auto time = [&](auto launch, double gb, const char *name) {
        for (int g : grids) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; rep++) {
                (void)hipEventRecord(e0, 0);
                launch(g);
                (void)hipEventRecord(e1, 0);
                (void)hipEventSynchronize(e1);
                float ms = 0;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (rep && ms < best) best = ms;
            }
            printf("%-28s grid %5d  %7.3f ms  %7.1f GB/s\n", name, g, best, gb / (best / 1e3));
        }
    };
    const double GB = bytes / 1e9;
    time([&](int g) { hipLaunchKernelGGL(k_write<false>, dim3(g), dim3(256), 0, 0, a, n); }, GB, "write 16B plain");
    time([&](int g) { hipLaunchKernelGGL(k_write<true>, dim3(g), dim3(256), 0, 0, a, n); }, GB, "write 16B non-temporal");
    time([&](int g) { hipLaunchKernelGGL(k_read, dim3(g), dim3(256), 0, 0, a, n, sink); }, GB, "read 16B");
    time([&](int g) { hipLaunchKernelGGL(k_copy<false>, dim3(g), dim3(256), 0, 0, a, b, n); }, 2 * GB, "copy (r+w) plain");
    time([&](int g) { hipLaunchKernelGGL(k_copy<true>, dim3(g), dim3(256), 0, 0, a, b, n); }, 2 * GB, "copy (r+w) nt store");
    time([&](int g) {
        const size_t threads = (size_t)g * 256, iters = n / (5 * threads);
        hipLaunchKernelGGL((k_mix<5, 2>), dim3(g), dim3(256), 0, 0, a, b, iters, sink);
    }, 0, "mix");
    // report the mix with its true byte count per grid
    for (int g : grids) {
        const size_t threads = (size_t)g * 256, iters = n / (5 * threads);
        const double gb = (double)iters * threads * 7 * 16 / 1e9;
        float best = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL((k_mix<5, 2>), dim3(g), dim3(256), 0, 0, a, b, iters, sink);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        printf("%-28s grid %5d  %7.3f ms  %7.1f GB/s (%.1f GB: 5 writes per 2 reads)\n", "mix 71% write / 29% read", g, best, gb / (best / 1e3), gb);
    }
    return 0;
}
