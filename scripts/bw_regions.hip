// bw_regions.hip -- does streaming-write bandwidth depend on WHICH piece of HBM a buffer landed in?  N buffers of
// 9.6 GB allocated back to back, each written with the same 16-byte non-temporal streaming kernel; then a mixed pass
// (reads from buffer 0, writes to buffer k).  hipcc --offload-arch=gfx950 -O3 -o scripts/bw_regions scripts/bw_regions.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k_write(f4 *dst, size_t n)
{
    f4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store(v, &dst[i]);
}
__global__ void k_read(const f4 *src, size_t n, float *sink)
{
    f4 acc = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += src[i];
    if (acc.x + acc.y == 12345.678f) *sink = acc.x;
}
int main(int argc, char **argv)
{
    const int nb = argc > 1 ? atoi(argv[1]) : 16;
    const size_t bytes = (size_t)(argc > 2 ? atoi(argv[2]) : 9600) << 20, n = bytes / 16;  // MiB per buffer
    std::vector<f4 *> bufs;
    float *sink;
    hipMalloc(&sink, 4);
    for (int k = 0; k < nb; k++) {
        f4 *p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) break;
        hipMemset(p, 0, bytes);
        bufs.push_back(p);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int pass = 0; pass < 2; pass++) {
        printf("%s GB/s per buffer:", pass ? "read " : "write");
        for (size_t k = 0; k < bufs.size(); k++) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; rep++) {
                hipEventRecord(e0, 0);
                if (pass) hipLaunchKernelGGL(k_read, dim3(8192), dim3(256), 0, 0, bufs[k], n, sink);
                else hipLaunchKernelGGL(k_write, dim3(8192), dim3(256), 0, 0, bufs[k], n);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf(" %.0f", bytes / 1e9 / (best / 1e3));
        }
        printf("\n");
    }
    printf("buffer addresses:");
    for (auto p : bufs) printf(" %p", (void *)p);
    printf("\n");
    return 0;
}
