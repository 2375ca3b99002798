// vmm_probe7.hip -- do other allocation flavours have classes too?  12 GiB buffers from hipMalloc, hipExtMallocWithFlags
// (uncached, fine-grained) and hipMallocAsync (stream-ordered pool): streaming write rate of each, several of every kind.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/vmm_probe7 scripts/vmm_probe7.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k_write(f4 *dst, size_t n)
{
    f4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store(v, &dst[i]);
}
__global__ void k_write_plain(f4 *dst, size_t n)
{
    f4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = v;
}
static hipEvent_t e0, e1;
template <class F> static double gbs(size_t bytes, F launch)
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
    return bytes / 1e9 / (best / 1e3);
}
int main()
{
    const size_t bytes = 12ull << 30, n = bytes / 16;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const char *names[] = {"hipMalloc", "hipExtMallocWithFlags(uncached)", "hipExtMallocWithFlags(finegrained)", "hipMallocAsync"};
    for (int kind = 0; kind < 4; kind++) {
        std::vector<void *> bufs;
        for (int k = 0; k < 5; k++) {
            void *p = nullptr;
            hipError_t e = hipSuccess;
            if (kind == 0) e = hipMalloc(&p, bytes);
            else if (kind == 1) e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached);
            else if (kind == 2) e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
            else e = hipMallocAsync(&p, bytes, 0);
            if (e != hipSuccess) {
                printf("%s: %s\n", names[kind], hipGetErrorString(e));
                (void)hipGetLastError();
                break;
            }
            bufs.push_back(p);
        }
        hipDeviceSynchronize();
        printf("%-36s nt stores:", names[kind]);
        for (void *p : bufs) printf(" %.0f", gbs(bytes, [&] { hipLaunchKernelGGL(k_write, dim3(8192), dim3(256), 0, 0, (f4 *)p, n); }));
        printf("  | plain stores:");
        for (void *p : bufs) printf(" %.0f", gbs(bytes, [&] { hipLaunchKernelGGL(k_write_plain, dim3(8192), dim3(256), 0, 0, (f4 *)p, n); }));
        printf("\n");
        for (void *p : bufs) {
            if (kind == 3) hipFreeAsync(p, 0); else hipFree(p);
        }
        hipDeviceSynchronize();
    }
    return 0;
}
