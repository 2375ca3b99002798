// vmm_probe.hip -- is the "slow output allocation"  a property of PHYSICAL chunks of HBM that a
// virtual-memory pool could leave out?  K physical chunks (hipMemCreate) mapped back to back into one VA range;
// every chunk written with the 16-byte non-temporal streaming kernel on its own, then windows of W chunks as one
// range (what an output buffer would be), then plain hipMalloc buffers of the same size for comparison.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/vmm_probe scripts/vmm_probe.hip
//   scripts/vmm_probe [chunks=64] [chunk MiB=1024] [window chunks=12] [hipMalloc buffers=6]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
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
__global__ void k_write(f4 *dst, size_t n)
{
    f4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store(v, &dst[i]);
}
static hipEvent_t e0, e1;
static double write_gbs(void *p, size_t bytes, int reps = 5)
{
    float best = 1e9f;
    for (int r = 0; r < reps; r++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_write, dim3(8192), dim3(256), 0, 0, (f4 *)p, bytes / 16);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
    }
    return bytes / 1e9 / (best / 1e3);
}
int main(int argc, char **argv)
{
    const int K = argc > 1 ? atoi(argv[1]) : 64;
    const size_t chunk = (size_t)(argc > 2 ? atoi(argv[2]) : 1024) << 20;
    const int W = argc > 3 ? atoi(argv[3]) : 12;
    const int NB = argc > 4 ? atoi(argv[4]) : 6;
    CK(hipSetDevice(0));
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    size_t free_b = 0, tot_b = 0;
    CK(hipMemGetInfo(&free_b, &tot_b));
    printf("granularity %zu, free %.1f GiB of %.1f GiB, %d chunks of %zu MiB\n", gran, free_b / 1073741824.0, tot_b / 1073741824.0, K, chunk >> 20);

    // (1) plain hipMalloc buffers of the window size first: the classes DESIGN section 4 describes
    const size_t wbytes = chunk * (size_t)W;
    std::vector<void *> plain;
    for (int k = 0; k < NB; k++) {
        void *p = nullptr;
        if (hipMalloc(&p, wbytes) != hipSuccess) break;
        plain.push_back(p);
    }
    printf("hipMalloc buffers of %.1f GiB, write GB/s:", wbytes / 1073741824.0);
    for (void *p : plain) printf(" %.0f", write_gbs(p, wbytes));
    printf("\n");
    // per-1/W slice of each plain buffer
    for (size_t b = 0; b < plain.size(); b++) {
        printf("  buffer %zu by chunk-sized slice:", b);
        for (int w = 0; w < W; w++) printf(" %.0f", write_gbs((char *)plain[b] + (size_t)w * chunk, chunk));
        printf("\n");
    }
    for (void *p : plain) hipFree(p);

    // (2) K physical chunks in one VA range
    void *va = nullptr;
    CK(hipMemAddressReserve(&va, chunk * (size_t)K, gran, nullptr, 0));
    std::vector<hipMemGenericAllocationHandle_t> h(K);
    int got = 0;
    for (; got < K; got++) {
        if (hipMemCreate(&h[got], chunk, &prop, 0) != hipSuccess) break;
        CK(hipMemMap((char *)va + (size_t)got * chunk, chunk, 0, h[got], 0));
    }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, chunk * (size_t)got, &acc, 1));
    printf("mapped %d chunks at %p\n", got, va);
    std::vector<double> cg(got);
    printf("per chunk write GB/s:");
    for (int k = 0; k < got; k++) {
        cg[k] = write_gbs((char *)va + (size_t)k * chunk, chunk);
        printf(" %.0f", cg[k]);
    }
    printf("\n");
    printf("windows of %d chunks (one range) write GB/s | mean of its chunks:", W);
    for (int k = 0; k + W <= got; k += W) {
        double m = 0;
        for (int j = 0; j < W; j++) m += cg[k + j];
        printf(" %.0f|%.0f", write_gbs((char *)va + (size_t)k * chunk, wbytes), m / W);
    }
    printf("\n");
    // (3) a window assembled from the FASTEST chunks: remap into a second VA range
    std::vector<int> order(got);
    for (int k = 0; k < got; k++) order[k] = k;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return cg[a] > cg[b]; });
    if (got >= 2 * W) {
        void *va2 = nullptr;
        CK(hipMemAddressReserve(&va2, wbytes * 2, gran, nullptr, 0));
        // unmap the chosen chunks from the first range, map them into the second: fastest W, then slowest W
        for (int j = 0; j < W; j++) {
            CK(hipMemUnmap((char *)va + (size_t)order[j] * chunk, chunk));
            CK(hipMemMap((char *)va2 + (size_t)j * chunk, chunk, 0, h[order[j]], 0));
            CK(hipMemUnmap((char *)va + (size_t)order[got - 1 - j] * chunk, chunk));
            CK(hipMemMap((char *)va2 + wbytes + (size_t)j * chunk, chunk, 0, h[order[got - 1 - j]], 0));
        }
        CK(hipMemSetAccess(va2, wbytes * 2, &acc, 1));
        printf("window of the %d fastest chunks: %.0f GB/s; of the %d slowest: %.0f GB/s\n", W, write_gbs(va2, wbytes), W,
               write_gbs((char *)va2 + wbytes, wbytes));
    }
    return 0;
}
