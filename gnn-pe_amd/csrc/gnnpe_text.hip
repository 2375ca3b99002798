// gnnpe_text.hip -- R7: decimal-ASCII rendering of the offline outputs on the GPU.
//
// The reference writes all_paths.txt and partition_paths.txt with `operator<<` + `endl` per line
// (GNN-PE/src/main.cpp:98-119): "<v0> <v1> <v2> \n" -- every id followed by ONE space, including the
// last -- and "<path id>\n".  Here the bytes are produced on the device (digit count -> exclusive
// scan -> tile-staged render) and the host only writes the buffer to the file.
#include <hipcub/hipcub.hpp>

#include "gnnpe_common.h"

namespace gnnpe {

__device__ __forceinline__ uint32_t ndigits32(uint32_t v)
{
    return v < 10u ? 1u : v < 100u ? 2u : v < 1000u ? 3u : v < 10000u ? 4u : v < 100000u ? 5u : v < 1000000u ? 6u
         : v < 10000000u ? 7u : v < 100000000u ? 8u : v < 1000000000u ? 9u : 10u;
}
__device__ __forceinline__ uint32_t ndigits64(uint64_t v)
{
    uint32_t n = 1;
    while (v >= 10ull) { v /= 10ull; n++; }
    return n;
}

// bytes of row r: sum(digits) + L separators + '\n'
__global__ void k_text_row_len(uint64_t n_rows, uint32_t L, const uint32_t *__restrict__ vids, uint32_t *__restrict__ len)
{
    for (uint64_t r = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; r <= n_rows; r += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t b = 0;
        if (r < n_rows) {
            b = L + 1;
            for (uint32_t k = 0; k < L; k++) b += ndigits32(vids[r * L + k]);
        }
        len[r] = b;  // len[n_rows] = 0: scan sentinel
    }
}

__global__ void k_text_id_len(uint64_t n, const uint64_t *__restrict__ ids, uint32_t *__restrict__ len)
{
    for (uint64_t r = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; r <= n; r += (uint64_t)gridDim.x * blockDim.x)
        len[r] = r < n ? ndigits64(ids[r]) + 1u : 0u;
}

constexpr int kTextRows = 512;               // rows rendered per block
constexpr int kTextLds = kTextRows * 48 + 8;  // 4 ids of 10 digits per row fit; longer rows store directly

__device__ __forceinline__ void put_u64(char *dst, uint64_t v, uint32_t nd)
{
    for (int i = (int)nd - 1; i >= 0; i--) {
        dst[i] = (char)('0' + (uint32_t)(v % 10ull));
        v /= 10ull;
    }
}

// MODE 0: rows of L vertex ids ("<v> " x L + "\n");  MODE 1: one uint64 id per line
template <int MODE>
__global__ __launch_bounds__(256) void k_text_render(uint64_t n_rows, uint32_t L, const void *__restrict__ src,
                                                     const uint64_t *__restrict__ off, char *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) char s_txt[kTextLds];
    const uint64_t r0 = (uint64_t)blockIdx.x * kTextRows;
    if (r0 >= n_rows) return;
    const uint64_t r1 = min(r0 + (uint64_t)kTextRows, n_rows);
    const uint64_t b0 = off[r0], b1 = off[r1];
    const uint32_t shift = (uint32_t)(b0 & 3ull);  // keep LDS and global 4-byte phases equal
    const uint64_t nb = b1 - b0;
    const bool staged = nb + shift <= (uint64_t)kTextLds;
    for (uint64_t r = r0 + threadIdx.x; r < r1; r += blockDim.x) {
        char *dst = staged ? (s_txt + shift + (off[r] - b0)) : (out + off[r]);
        if (MODE == 0) {
            const uint32_t *v = reinterpret_cast<const uint32_t *>(src) + r * L;
            for (uint32_t k = 0; k < L; k++) {
                const uint32_t x = v[k], nd = ndigits32(x);
                put_u64(dst, x, nd);
                dst[nd] = ' ';
                dst += nd + 1;
            }
            *dst = '\n';
        } else {
            const uint64_t x = reinterpret_cast<const uint64_t *>(src)[r];
            const uint32_t nd = ndigits64(x);
            put_u64(dst, x, nd);
            dst[nd] = '\n';
        }
    }
    if (!staged) return;
    __syncthreads();
    // aligned dwords in the middle, bytes at the ragged ends (LDS index i <-> global byte g0 + i)
    const uint64_t g0 = b0 - shift;  // multiple of 4
    const uint64_t total = nb + shift;
    const uint64_t first_dw = (shift + 3) / 4, last_dw = total / 4;  // dwords [first_dw, last_dw) are complete
    for (uint64_t w = first_dw + threadIdx.x; w < last_dw; w += blockDim.x)
        reinterpret_cast<uint32_t *>(out + g0)[w] = reinterpret_cast<const uint32_t *>(s_txt)[w];
    const uint64_t head_end = last_dw >= first_dw ? first_dw * 4 : total;
    if (threadIdx.x < 4) {
        const uint64_t i = shift + threadIdx.x;
        if (i < head_end) out[g0 + i] = s_txt[i];
    } else if (threadIdx.x < 8 && last_dw >= first_dw) {
        const uint64_t t = last_dw * 4 + (threadIdx.x - 4);
        if (t < total) out[g0 + t] = s_txt[t];
    }
}

__global__ void k_add_u64(uint64_t n, uint64_t b, uint64_t *v)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) v[i] += b;
}

struct CastU64b {
    __host__ __device__ uint64_t operator()(uint32_t v) const { return (uint64_t)v; }
};

static int scan_len(gnnpe_ctx *c, const uint32_t *len, uint64_t *off, uint64_t n)
{
    hipcub::TransformInputIterator<uint64_t, CastU64b, const uint32_t *> it(len, CastU64b());
    size_t tb = 0;
    GNNPE_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, it, off, (int64_t)n, c->stream));
    int rc = c->cub_tmp.reserve(tb);
    if (rc) return rc;
    tb = c->cub_tmp.bytes;
    GNNPE_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(c->cub_tmp.p, tb, it, off, (int64_t)n, c->stream));
    return GNNPE_OK;
}

static int render(gnnpe_ctx *c, int mode, uint64_t n_rows, uint32_t L, const void *dev_src, void *dev_text, uint64_t cap,
                  uint64_t *nbytes)
{
    GNNPE_REQUIRE(c && nbytes, GNNPE_ERR_ARG, "null argument");
    GNNPE_HIP_TRY(hipSetDevice(c->device));
    *nbytes = 0;
    if (n_rows == 0) return GNNPE_OK;
    GNNPE_REQUIRE(dev_src, GNNPE_ERR_ARG, "null source");
    int rc;
    if ((rc = c->text_len.reserve((n_rows + 1) * 4)) || (rc = c->text_off.reserve((n_rows + 1) * 8))) return rc;
    uint32_t *len = c->text_len.as<uint32_t>();
    uint64_t *off = c->text_off.as<uint64_t>();
    if (mode == 0)
        hipLaunchKernelGGL(k_text_row_len, dim3(grid_for(n_rows + 1)), dim3(kBlock), 0, c->stream, n_rows, L,
                           (const uint32_t *)dev_src, len);
    else
        hipLaunchKernelGGL(k_text_id_len, dim3(grid_for(n_rows + 1)), dim3(kBlock), 0, c->stream, n_rows,
                           (const uint64_t *)dev_src, len);
    GNNPE_HIP_TRY(hipGetLastError());
    if ((rc = scan_len(c, len, off, n_rows + 1))) return rc;
    *c->h_pinned = 0;
    GNNPE_HIP_TRY(hipMemcpyAsync(c->h_pinned, off + n_rows, 8, hipMemcpyDeviceToHost, c->stream));
    GNNPE_HIP_TRY(hipStreamSynchronize(c->stream));
    *nbytes = *c->h_pinned;
    if (!dev_text) return GNNPE_OK;  // sizing call
    GNNPE_REQUIRE(*nbytes <= cap, GNNPE_ERR_ARG, "text buffer holds %llu bytes, need %llu", (unsigned long long)cap,
                  (unsigned long long)*nbytes);
    GNNPE_REQUIRE((reinterpret_cast<uintptr_t>(dev_text) & 3u) == 0, GNNPE_ERR_ARG, "text buffer must be 4-byte aligned");
    const uint64_t nblk = (n_rows + kTextRows - 1) / kTextRows;
    GNNPE_REQUIRE(nblk < (1ull << 31), GNNPE_ERR_RANGE, "too many rows in one call; chunk the range");
    if (mode == 0)
        hipLaunchKernelGGL((k_text_render<0>), dim3((uint32_t)nblk), dim3(256), 0, c->stream, n_rows, L, dev_src, off,
                           (char *)dev_text);
    else
        hipLaunchKernelGGL((k_text_render<1>), dim3((uint32_t)nblk), dim3(256), 0, c->stream, n_rows, L, dev_src, off,
                           (char *)dev_text);
    GNNPE_HIP_TRY(hipGetLastError());
    return GNNPE_OK;
}

// flag[i] = part[i] == pid; selected output = id_base + i
struct IsPart {
    const uint32_t *part;
    uint32_t pid;
    __host__ __device__ bool operator()(uint64_t i) const { return part[i] == pid; }
};

}  // namespace gnnpe

using namespace gnnpe;

extern "C" {

int gnnpe_text_paths(gnnpe_ctx *c, uint64_t n_rows, uint32_t L, const void *dev_vids, void *dev_text, uint64_t cap,
                     uint64_t *nbytes)
{
    GNNPE_REQUIRE(L >= 1 && L <= 16, GNNPE_ERR_ARG, "bad path width %u", L);
    return render(c, 0, n_rows, L, dev_vids, dev_text, cap, nbytes);
}

int gnnpe_text_ids(gnnpe_ctx *c, uint64_t n, const void *dev_ids, void *dev_text, uint64_t cap, uint64_t *nbytes)
{
    return render(c, 1, n, 1, dev_ids, dev_text, cap, nbytes);
}

int gnnpe_select_partition(gnnpe_ctx *c, uint64_t n, const void *dev_part, uint32_t pid, uint64_t id_base, void *dev_ids,
                           uint64_t *count)
{
    GNNPE_REQUIRE(c && count, GNNPE_ERR_ARG, "null argument");
    GNNPE_HIP_TRY(hipSetDevice(c->device));
    *count = 0;
    if (n == 0) return GNNPE_OK;
    GNNPE_REQUIRE(dev_part && dev_ids, GNNPE_ERR_ARG, "null argument");
    GNNPE_REQUIRE(n < (1ull << 31), GNNPE_ERR_RANGE, "chunk the range: %llu rows", (unsigned long long)n);
    int rc;
    if ((rc = c->small.reserve(256))) return rc;
    uint64_t *d_num = c->small.as<uint64_t>() + 16;
    // select local indices i with part[i] == pid, then shift by id_base
    hipcub::CountingInputIterator<uint64_t> idx(0);
    IsPart pred{(const uint32_t *)dev_part, pid};
    size_t tb = 0;
    GNNPE_HIP_TRY(hipcub::DeviceSelect::If(nullptr, tb, idx, (uint64_t *)dev_ids, d_num, (int)n, pred, c->stream));
    if ((rc = c->cub_tmp.reserve(tb))) return rc;
    tb = c->cub_tmp.bytes;
    GNNPE_HIP_TRY(hipcub::DeviceSelect::If(c->cub_tmp.p, tb, idx, (uint64_t *)dev_ids, d_num, (int)n, pred, c->stream));
    *c->h_pinned = 0;
    GNNPE_HIP_TRY(hipMemcpyAsync(c->h_pinned, d_num, 8, hipMemcpyDeviceToHost, c->stream));
    GNNPE_HIP_TRY(hipStreamSynchronize(c->stream));
    *count = *c->h_pinned;
    if (id_base && *count) {
        hipLaunchKernelGGL(k_add_u64, dim3(grid_for(*count)), dim3(kBlock), 0, c->stream, *count, id_base, (uint64_t *)dev_ids);
        GNNPE_HIP_TRY(hipGetLastError());
    }
    return GNNPE_OK;
}

}  // extern "C"
