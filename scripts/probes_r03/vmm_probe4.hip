// vmm_probe4.hip -- fourth placement probe: VIRTUAL or PHYSICAL?  (vmm_probe3: the window-to-window spread is the same
// for every write pattern, and survives shuffling the physical chunks.)  Two physical sets of 12 x 1 GiB chunks, each
// mapped in turn at the same list of virtual bases inside one reservation; if both sets show the same speed at the
// same base, the address bits that matter are virtual ones and a pool can simply pick a good base.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/vmm_probe4 scripts/vmm_probe4.hip
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
int main()
{
    const size_t chunk = 1ull << 30, W = 12, wbytes = chunk * W;
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
    const size_t span = 96ull << 30;
    void *va = nullptr;
    CK(hipMemAddressReserve(&va, span, 1ull << 30, nullptr, 0));
    printf("reservation at %p (%zu GiB)\n", va, span >> 30);
    std::vector<hipMemGenericAllocationHandle_t> h(2 * W);
    for (auto &x : h) CK(hipMemCreate(&x, chunk, &prop, 0));
    // first touch both sets somewhere so that their physical placement is fixed before the sweep
    for (int set = 0; set < 2; set++) {
        for (size_t j = 0; j < W; j++) CK(hipMemMap((char *)va + (80ull << 30) + j * chunk, chunk, 0, h[set * W + j], 0));
        CK(hipMemSetAccess((char *)va + (80ull << 30), wbytes, &acc, 1));
        hipLaunchKernelGGL(k_stride, dim3(8192), dim3(256), 0, 0, (f4 *)((char *)va + (80ull << 30)), wbytes / 16);
        CK(hipDeviceSynchronize());
        CK(hipMemUnmap((char *)va + (80ull << 30), wbytes));
    }
    const size_t offs[] = {0, 1ull << 30, 2ull << 30, 3ull << 30, 4ull << 30, 6ull << 30, 8ull << 30, 12ull << 30, 16ull << 30, 24ull << 30, 32ull << 30,
                           36ull << 30, 48ull << 30, 60ull << 30, (2ull << 20), (64ull << 20), (1ull << 30) + (2ull << 20), (12ull << 30) + (512ull << 20)};
    for (int set = 0; set < 2; set++) {
        printf("physical set %d: base offset GiB -> stride / block-contiguous GB/s\n", set);
        for (size_t o : offs) {
            char *b = (char *)va + o;
            for (size_t j = 0; j < W; j++) CK(hipMemMap(b + j * chunk, chunk, 0, h[set * W + j], 0));
            CK(hipMemSetAccess(b, wbytes, &acc, 1));
            const double a = timed(wbytes, [&] { hipLaunchKernelGGL(k_stride, dim3(8192), dim3(256), 0, 0, (f4 *)b, wbytes / 16); });
            const double c = timed(wbytes, [&] { hipLaunchKernelGGL(k_block_contig, dim3(2048), dim3(256), 0, 0, (f4 *)b, wbytes / 16); });
            printf("  %8.3f -> %.0f / %.0f\n", o / 1073741824.0, a, c);
            CK(hipMemUnmap(b, wbytes));
        }
    }
    // both sets mapped side by side at offsets 0 and 12 GiB, then swapped
    for (int swap = 0; swap < 2; swap++) {
        for (size_t j = 0; j < W; j++) {
            CK(hipMemMap((char *)va + j * chunk, chunk, 0, h[(swap ? W : 0) + j], 0));
            CK(hipMemMap((char *)va + wbytes + j * chunk, chunk, 0, h[(swap ? 0 : W) + j], 0));
        }
        CK(hipMemSetAccess(va, 2 * wbytes, &acc, 1));
        printf("side by side (%s): at 0 GiB %.0f, at 12 GiB %.0f\n", swap ? "set 1 | set 0" : "set 0 | set 1",
               timed(wbytes, [&] { hipLaunchKernelGGL(k_stride, dim3(8192), dim3(256), 0, 0, (f4 *)va, wbytes / 16); }),
               timed(wbytes, [&] { hipLaunchKernelGGL(k_stride, dim3(8192), dim3(256), 0, 0, (f4 *)((char *)va + wbytes), wbytes / 16); }));
        CK(hipMemUnmap(va, 2 * wbytes));
    }
    return 0;
}
