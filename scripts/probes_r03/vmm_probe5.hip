// vmm_probe5.hip -- one physical handle of ~11.2 GiB walked through K windows of ONE reservation (the output pool's flow),
// to find out whether large reservations are usable to their end.  Prints every window before it is touched.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/vmm_probe5 scripts/vmm_probe5.hip ; scripts/vmm_probe5 [windows=12] [separate reservations=0]
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
int main(int argc, char **argv)
{
    const int K = argc > 1 ? atoi(argv[1]) : 12;
    const bool separate = argc > 2 && atoi(argv[2]);
    const size_t bytes = 12004098048ull, stride = 12ull << 30;
    CK(hipSetDevice(0));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    hipMemGenericAllocationHandle_t h;
    CK(hipMemCreate(&h, bytes, &prop, 0));
    std::vector<char *> base(K);
    if (separate) {
        for (int k = 0; k < K; k++) {
            void *v = nullptr;
            CK(hipMemAddressReserve(&v, stride, 2u << 20, nullptr, 0));
            base[k] = (char *)v;
        }
    } else {
        void *v = nullptr;
        CK(hipMemAddressReserve(&v, stride * K, 2u << 20, nullptr, 0));
        for (int k = 0; k < K; k++) base[k] = (char *)v + k * stride;
    }
    for (int k = 0; k < K; k++) {
        printf("window %d at %p .. %p: ", k, (void *)base[k], (void *)(base[k] + bytes));
        fflush(stdout);
        CK(hipMemMap(base[k], bytes, 0, h, 0));
        CK(hipMemSetAccess(base[k], bytes, &acc, 1));
        float best = 1e9f;
        for (int r = 0; r < 3; r++) {
            hipEventRecord(e0, st);
            hipLaunchKernelGGL(k_write, dim3(8192), dim3(256), 0, st, (f4 *)base[k], bytes / 16);
            hipEventRecord(e1, st);
            CK(hipEventSynchronize(e1));
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            if (r) best = std::min(best, ms);
        }
        printf("%.0f GB/s\n", bytes / 1e9 / (best / 1e3));
        fflush(stdout);
        CK(hipMemUnmap(base[k], bytes));
    }
    printf("done\n");
    return 0;
}
