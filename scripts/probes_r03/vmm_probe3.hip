// vmm_probe3.hip -- third placement probe: is the window-to-window spread (vmm_probe2: 5.0-6.6 TB/s, a property of the
// physical backing, not additive over chunks) a property of the WRITE PATTERN?  Per 12 GiB window:
//   A  grid-stride, 8192 blocks (stride 32 MiB)          B  grid-stride, resident grid only (one moving 8 MiB window)
//   C  block-contiguous: block b writes [b S, (b+1) S)    D  wave-contiguous 4 KiB pieces dealt round-robin to waves (emit-like)
// then the same windows backed by 64 MiB physical chunks in random order (does fine-grained shuffling even it out?).
//   hipcc --offload-arch=gfx950 -O3 -o scripts/vmm_probe3 scripts/vmm_probe3.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);      \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)
__global__ void k_stride(f4 *dst, size_t n)
{
    f4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store(v, &dst[i]);
}
__global__ void k_block_contig(f4 *dst, size_t n)
{
    f4 v = {1.f, 2.f, 3.f, 4.f};
    const size_t per = (n + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) __builtin_nontemporal_store(v, &dst[i]);
}
// every wave writes pieces of `piece` f4s (contiguous), pieces dealt round-robin over all waves: the emit kernel's shape
__global__ void k_wave_pieces(f4 *dst, size_t n, unsigned piece)
{
    f4 v = {1.f, 2.f, 3.f, 4.f};
    const size_t nw = ((size_t)gridDim.x * blockDim.x) >> 6, w = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const unsigned lane = threadIdx.x & 63;
    for (size_t p = w; p * piece < n; p += nw) {
        const size_t lo = p * piece, hi = lo + piece < n ? lo + piece : n;
        for (size_t i = lo + lane; i < hi; i += 64) __builtin_nontemporal_store(v, &dst[i]);
    }
}
static hipEvent_t e0, e1;
template <class F> static double timed(size_t bytes, F launch)
{
    float best = 1e9f;
    for (int r = 0; r < 3; r++) {
        hipEventRecord(e0, 0);
        launch();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
    }
    return (double)bytes / 1e9 / (best / 1e3);
}
static void patterns(const char *tag, char *base, size_t wbytes, int nwin)
{
    const size_t n = wbytes / 16;
    const char *names[] = {"A stride 8192 blocks", "B stride 2048 blocks", "C block-contiguous 2048", "C block-contiguous 8192", "D wave pieces 4 KiB, 2048 blocks",
                           "D wave pieces 1 KiB, 2048 blocks", "D wave pieces 16 KiB, 2048 blocks"};
    for (int pat = 0; pat < 7; pat++) {
        printf("%s %-34s:", tag, names[pat]);
        for (int w = 0; w < nwin; w++) {
            f4 *p = (f4 *)(base + (size_t)w * wbytes);
            double g = 0;
            switch (pat) {
            case 0: g = timed(wbytes, [&] { hipLaunchKernelGGL(k_stride, dim3(8192), dim3(256), 0, 0, p, n); }); break;
            case 1: g = timed(wbytes, [&] { hipLaunchKernelGGL(k_stride, dim3(2048), dim3(256), 0, 0, p, n); }); break;
            case 2: g = timed(wbytes, [&] { hipLaunchKernelGGL(k_block_contig, dim3(2048), dim3(256), 0, 0, p, n); }); break;
            case 3: g = timed(wbytes, [&] { hipLaunchKernelGGL(k_block_contig, dim3(8192), dim3(256), 0, 0, p, n); }); break;
            case 4: g = timed(wbytes, [&] { hipLaunchKernelGGL(k_wave_pieces, dim3(2048), dim3(256), 0, 0, p, n, 256u); }); break;
            case 5: g = timed(wbytes, [&] { hipLaunchKernelGGL(k_wave_pieces, dim3(2048), dim3(256), 0, 0, p, n, 64u); }); break;
            case 6: g = timed(wbytes, [&] { hipLaunchKernelGGL(k_wave_pieces, dim3(2048), dim3(256), 0, 0, p, n, 1024u); }); break;
            }
            printf(" %.0f", g);
        }
        printf("\n");
    }
}
int main(int argc, char **argv)
{
    const int NW = argc > 1 ? atoi(argv[1]) : 6;  // windows
    const size_t wbytes = 12ull << 30;
    CK(hipSetDevice(0));
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    const size_t gran = 2u << 20;
    for (int mode = 0; mode < 3; mode++) {
        // mode 0: 1 GiB chunks in creation order; 1: 64 MiB chunks in creation order; 2: 64 MiB chunks, random order
        const size_t chunk = mode == 0 ? (1ull << 30) : (64ull << 20);
        const size_t K = wbytes * NW / chunk;
        void *va = nullptr;
        CK(hipMemAddressReserve(&va, chunk * K, gran, nullptr, 0));
        std::vector<hipMemGenericAllocationHandle_t> h(K);
        for (size_t k = 0; k < K; k++) CK(hipMemCreate(&h[k], chunk, &prop, 0));
        std::vector<size_t> order(K);
        for (size_t k = 0; k < K; k++) order[k] = k;
        if (mode == 2) std::shuffle(order.begin(), order.end(), std::mt19937(7));
        for (size_t k = 0; k < K; k++) CK(hipMemMap((char *)va + k * chunk, chunk, 0, h[order[k]], 0));
        CK(hipMemSetAccess(va, chunk * K, &acc, 1));
        const char *tag = mode == 0 ? "[1 GiB chunks, in order]   " : mode == 1 ? "[64 MiB chunks, in order]  " : "[64 MiB chunks, shuffled]  ";
        patterns(tag, (char *)va, wbytes, NW);
        CK(hipMemUnmap(va, chunk * K));
        for (size_t k = 0; k < K; k++) CK(hipMemRelease(h[k]));
        CK(hipMemAddressFree(va, chunk * K));
    }
    // plain hipMalloc windows for reference
    {
        std::vector<void *> bufs;
        for (int w = 0; w < NW; w++) {
            void *p = nullptr;
            if (hipMalloc(&p, wbytes) != hipSuccess) break;
            bufs.push_back(p);
        }
        printf("[hipMalloc] A / B / C2048 / D4K per buffer:");
        for (void *p : bufs) {
            const size_t n = wbytes / 16;
            printf("  %.0f/%.0f/%.0f/%.0f", timed(wbytes, [&] { hipLaunchKernelGGL(k_stride, dim3(8192), dim3(256), 0, 0, (f4 *)p, n); }),
                   timed(wbytes, [&] { hipLaunchKernelGGL(k_stride, dim3(2048), dim3(256), 0, 0, (f4 *)p, n); }),
                   timed(wbytes, [&] { hipLaunchKernelGGL(k_block_contig, dim3(2048), dim3(256), 0, 0, (f4 *)p, n); }),
                   timed(wbytes, [&] { hipLaunchKernelGGL(k_wave_pieces, dim3(2048), dim3(256), 0, 0, (f4 *)p, n, 256u); }));
        }
        printf("\n");
    }
    return 0;
}
