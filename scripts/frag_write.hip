// frag_write.hip -- would a MIDDLE-VERTEX-major emit kernel pay?  Such a kernel would read every neighbour record once
// (reads ~ 1 GB instead of 6 GB per launch at config 3) but write the output as FRAGMENTS: the rows of pair (s, b) are a
// run of ~10 rows inside s' region, i.e. ~480 bytes of pde and ~120 bytes of ids at a 16- / 4-byte aligned offset,
// written by the wave of b, with the neighbouring runs written by other waves at unrelated times.  This measures what
// the memory system gives such writes: runs of R rows, the run slots visited in identity order (= today's contiguous
// streams) or through a random permutation (= middle-vertex-major), plain or non-temporal stores.
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/frag_write scripts/frag_write.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <numeric>
#include <random>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(4))) Id3 { uint32_t a, b, c; };
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// one wave per group of `runs_per_wave` runs (the pairs of one middle vertex); the pieces of the group flattened over the lanes
template <bool NT> __global__ __launch_bounds__(256) void k_frag_pde(d2 *dst, const uint32_t *perm, uint64_t n_runs, uint32_t R, uint32_t runs_per_wave)
{
    const unsigned lane = threadIdx.x & 63u;
    const uint64_t nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint32_t PR = R * 3;  // 16-byte pieces per run (48-byte rows)
    const d2 v = {1.0, 2.0};
    for (uint64_t w = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6; w * runs_per_wave < n_runs; w += nw) {
        const uint64_t r0 = w * runs_per_wave;
        const uint32_t nr = (uint32_t)min((uint64_t)runs_per_wave, n_runs - r0);
        for (uint32_t g = lane; g < nr * PR; g += 64) {
            const uint32_t run = g / PR, within = g % PR;
            d2 *p = dst + (uint64_t)perm[r0 + run] * PR + within;
            if (NT) __builtin_nontemporal_store(v, p); else *p = v;
        }
    }
}
template <bool NT> __global__ __launch_bounds__(256) void k_frag_ids(Id3 *dst, const uint32_t *perm, uint64_t n_runs, uint32_t R, uint32_t runs_per_wave)
{
    const unsigned lane = threadIdx.x & 63u;
    const uint64_t nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const Id3 v = {1u, 2u, 3u};
    for (uint64_t w = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6; w * runs_per_wave < n_runs; w += nw) {
        const uint64_t r0 = w * runs_per_wave;
        const uint32_t nr = (uint32_t)min((uint64_t)runs_per_wave, n_runs - r0);
        for (uint32_t g = lane; g < nr * R; g += 64) {
            const uint32_t run = g / R, within = g % R;
            Id3 *p = dst + (uint64_t)perm[r0 + run] * R + within;
            if (NT) {
                __builtin_nontemporal_store(v.a, &p->a);
                __builtin_nontemporal_store(v.b, &p->b);
                __builtin_nontemporal_store(v.c, &p->c);
            } else {
                *p = v;
            }
        }
    }
}

int main()
{
    const uint32_t R = 10, RPW = 20;        // config 3: ~10 paths per (s, b) pair, ~20 pairs per middle vertex
    const uint64_t n_runs = 20'000'000;     // 2.0e8 rows
    const uint64_t rows = n_runs * R;
    d2 *pde;
    Id3 *ids;
    uint32_t *perm_id, *perm_rnd;
    CK(hipMalloc(&pde, rows * 48));
    CK(hipMalloc(&ids, rows * 12));
    CK(hipMalloc(&perm_id, n_runs * 4));
    CK(hipMalloc(&perm_rnd, n_runs * 4));
    CK(hipMemset(pde, 0, rows * 48));
    CK(hipMemset(ids, 0, rows * 12));
    std::vector<uint32_t> h(n_runs);
    std::iota(h.begin(), h.end(), 0u);
    CK(hipMemcpy(perm_id, h.data(), n_runs * 4, hipMemcpyHostToDevice));
    std::mt19937_64 rng(2022);
    std::shuffle(h.begin(), h.end(), rng);
    CK(hipMemcpy(perm_rnd, h.data(), n_runs * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const dim3 grid(256 * 8), block(256);
    auto time = [&](auto launch, const char *name, double gb) {
        float best = 1e9f;
        for (int it = 0; it < 4; it++) {
            (void)hipEventRecord(e0, 0);
            launch();
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (it) best = std::min(best, ms);
        }
        printf("%-46s %8.3f ms  %7.1f GB/s\n", name, best, gb / (best / 1e3));
        return 0;
    };
    const double gp = rows * 48 / 1e9, gi = rows * 12 / 1e9;
    time([&] { hipLaunchKernelGGL(k_frag_pde<true>, grid, block, 0, 0, pde, perm_id, n_runs, R, RPW); }, "pde 480 B runs, in order, non-temporal", gp);
    time([&] { hipLaunchKernelGGL(k_frag_pde<false>, grid, block, 0, 0, pde, perm_id, n_runs, R, RPW); }, "pde 480 B runs, in order, plain", gp);
    time([&] { hipLaunchKernelGGL(k_frag_pde<true>, grid, block, 0, 0, pde, perm_rnd, n_runs, R, RPW); }, "pde 480 B runs, scattered, non-temporal", gp);
    time([&] { hipLaunchKernelGGL(k_frag_pde<false>, grid, block, 0, 0, pde, perm_rnd, n_runs, R, RPW); }, "pde 480 B runs, scattered, plain", gp);
    time([&] { hipLaunchKernelGGL(k_frag_ids<true>, grid, block, 0, 0, ids, perm_id, n_runs, R, RPW); }, "ids 120 B runs, in order, non-temporal", gi);
    time([&] { hipLaunchKernelGGL(k_frag_ids<false>, grid, block, 0, 0, ids, perm_id, n_runs, R, RPW); }, "ids 120 B runs, in order, plain", gi);
    time([&] { hipLaunchKernelGGL(k_frag_ids<true>, grid, block, 0, 0, ids, perm_rnd, n_runs, R, RPW); }, "ids 120 B runs, scattered, non-temporal", gi);
    time([&] { hipLaunchKernelGGL(k_frag_ids<false>, grid, block, 0, 0, ids, perm_rnd, n_runs, R, RPW); }, "ids 120 B runs, scattered, plain", gi);
    return 0;
}
