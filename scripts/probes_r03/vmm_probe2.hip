// vmm_probe2.hip -- second placement probe (see vmm_probe.hip): a 1 GiB launch is too short to tell 5.1 from 6.4 TB/s
// (launch ramp), so every physical chunk is written REPS times inside one launch.  Then: windows of W consecutive
// chunks; the slowest window's chunks remapped to a new VA range (VA or PA?); windows interleaving chunks of a fast and a
// slow window; windows of other sizes (is it the size of the streamed range?).
//   hipcc --offload-arch=gfx950 -O3 -o scripts/vmm_probe2 scripts/vmm_probe2.hip
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
__global__ void k_write(f4 *dst, size_t n, int reps)
{
    f4 v = {1.f, 2.f, 3.f, 4.f};
    for (int r = 0; r < reps; r++)
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store(v, &dst[i]);
}
static hipEvent_t e0, e1;
static double write_gbs(void *p, size_t bytes, int inner = 1, int reps = 3)
{
    float best = 1e9f;
    for (int r = 0; r < reps; r++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_write, dim3(8192), dim3(256), 0, 0, (f4 *)p, bytes / 16, inner);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
    }
    return (double)bytes * inner / 1e9 / (best / 1e3);
}
int main(int argc, char **argv)
{
    const int K = argc > 1 ? atoi(argv[1]) : 96;
    const size_t chunk = (size_t)(argc > 2 ? atoi(argv[2]) : 1024) << 20;
    const int W = argc > 3 ? atoi(argv[3]) : 12;
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
    void *va = nullptr;
    CK(hipMemAddressReserve(&va, chunk * (size_t)K, gran, nullptr, 0));
    std::vector<hipMemGenericAllocationHandle_t> h(K);
    int got = 0;
    for (; got < K; got++) {
        if (hipMemCreate(&h[got], chunk, &prop, 0) != hipSuccess) break;
        CK(hipMemMap((char *)va + (size_t)got * chunk, chunk, 0, h[got], 0));
    }
    CK(hipMemSetAccess(va, chunk * (size_t)got, &acc, 1));
    printf("mapped %d chunks of %zu MiB at %p\n", got, chunk >> 20, va);
    std::vector<double> cg(got);
    printf("per chunk, 16 x written in one launch, GB/s:");
    for (int k = 0; k < got; k++) {
        cg[k] = write_gbs((char *)va + (size_t)k * chunk, chunk, 16);
        printf(" %.0f", cg[k]);
    }
    printf("\n");
    const size_t wbytes = chunk * (size_t)W;
    std::vector<double> wg;
    printf("windows of %d chunks, GB/s:", W);
    for (int k = 0; k + W <= got; k += W) {
        wg.push_back(write_gbs((char *)va + (size_t)k * chunk, wbytes));
        printf(" %.0f", wg.back());
    }
    printf("\nsame windows, second pass:");
    for (int k = 0; k + W <= got; k += W) printf(" %.0f", write_gbs((char *)va + (size_t)k * chunk, wbytes));
    printf("\nwindows shifted by %d chunks:", W / 2);
    for (int k = W / 2; k + W <= got; k += W) printf(" %.0f", write_gbs((char *)va + (size_t)k * chunk, wbytes));
    printf("\nwindow sizes from chunk 0 (chunks: GB/s):");
    for (int w : {2, 4, 6, 8, 12, 16, 24, 32, 48}) if (w <= got) printf(" %d:%.0f", w, write_gbs(va, chunk * (size_t)w));
    const int slow = (int)(std::min_element(wg.begin(), wg.end()) - wg.begin()), fast = (int)(std::max_element(wg.begin(), wg.end()) - wg.begin());
    printf("\nwindow sizes from the slowest window's first chunk (%d):", slow * W);
    for (int w : {2, 4, 6, 8, 12, 16, 24}) if (slow * W + w <= got) printf(" %d:%.0f", w, write_gbs((char *)va + (size_t)slow * W * chunk, chunk * (size_t)w));
    printf("\nslowest window %d (%.0f), fastest %d (%.0f)\n", slow, wg[slow], fast, wg[fast]);
    // remap: slow window's chunks to a fresh VA range in the same order; then slow/fast chunks interleaved
    void *va2 = nullptr;
    CK(hipMemAddressReserve(&va2, wbytes * 3, gran, nullptr, 0));
    for (int j = 0; j < W; j++) {
        CK(hipMemUnmap((char *)va + (size_t)(slow * W + j) * chunk, chunk));
        CK(hipMemUnmap((char *)va + (size_t)(fast * W + j) * chunk, chunk));
    }
    for (int j = 0; j < W; j++) {
        CK(hipMemMap((char *)va2 + (size_t)j * chunk, chunk, 0, h[slow * W + j], 0));
        CK(hipMemMap((char *)va2 + wbytes + (size_t)j * chunk, chunk, 0, h[fast * W + j], 0));
    }
    CK(hipMemSetAccess(va2, wbytes * 2, &acc, 1));
    printf("remapped to a new VA range: slow window's chunks %.0f, fast window's chunks %.0f\n", write_gbs(va2, wbytes),
           write_gbs((char *)va2 + wbytes, wbytes));
    CK(hipMemUnmap(va2, wbytes * 2));
    for (int j = 0; j < W; j++) {  // first window: even positions slow chunks, odd positions fast; second: the rest
        const int a = j / 2;
        CK(hipMemMap((char *)va2 + (size_t)j * chunk, chunk, 0, (j & 1) ? h[fast * W + a] : h[slow * W + a], 0));
        CK(hipMemMap((char *)va2 + wbytes + (size_t)j * chunk, chunk, 0, (j & 1) ? h[fast * W + W / 2 + a] : h[slow * W + W / 2 + a], 0));
    }
    CK(hipMemSetAccess(va2, wbytes * 2, &acc, 1));
    printf("slow and fast chunks interleaved: %.0f %.0f\n", write_gbs(va2, wbytes), write_gbs((char *)va2 + wbytes, wbytes));
    // reversed order of the slow window's chunks
    CK(hipMemUnmap(va2, wbytes * 2));
    for (int j = 0; j < W; j++) CK(hipMemMap((char *)va2 + (size_t)j * chunk, chunk, 0, h[slow * W + (W - 1 - j)], 0));
    CK(hipMemSetAccess(va2, wbytes, &acc, 1));
    printf("slow window's chunks in reverse order: %.0f\n", write_gbs(va2, wbytes));
    return 0;
}
