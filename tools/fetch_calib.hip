// tools/fetch_calib.hip -- calibrates FETCH_SIZE / WRITE_SIZE for k_decode's own access pattern.
//   hipcc --offload-arch=gfx950 -O3 -o build/fetch_calib tools/fetch_calib.hip
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -- build/fetch_calib     (WRITE_SIZE: second run)
// MI355X_MICROARCH.md calibrates FETCH_SIZE on wide coalesced streaming reads (128-byte requests
// tallied as 64 bytes: double it) and says other access widths must be calibrated on a known byte
// count.  k_decode reads 64 bytes per lane (4 x 16-byte loads) from a lane-private 128-byte line and
// comes back for the other half of the line a few rows later; it stores 16 bytes per lane into a
// lane-private 32-byte sector.  The kernels below do exactly that over buffers far larger than the
// L2 and the Infinity Cache, with known byte counts; the program prints the time of each so that the
// counter reading can be set against both the useful bytes and the time a full-line fetch would take.
// Diagnostic only, not part of the product.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// (A) wide coalesced stream: lane i of the grid reads 16 bytes at 16*i (+ grid stride)
__global__ void __launch_bounds__(256) k_stream(const uint4 *src, size_t n16, uint32_t *out)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = src[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u)
        out[0] = acc;
}

// (B) 64 bytes (4 x 16) per lane from the `half`-th half of a lane-private 128-byte line
__global__ void __launch_bounds__(256) k_half_lines(const uint4 *src, size_t n_lines, int half, uint32_t *out)
{
    uint32_t acc = 0;
    for (size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x; l < n_lines; l += (size_t)gridDim.x * blockDim.x) {
        const uint4 *p = src + l * 8 + (size_t)half * 4;
        const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc ^= a.x ^ b.y ^ c.z ^ d.w;
    }
    if (acc == 0x12345678u)
        out[0] = acc;
}

// (C) 16-byte store per lane into the `half`-th half of a lane-private 32-byte sector
__global__ void __launch_bounds__(256) k_half_sectors(uint4 *dst, size_t n_sectors, int half)
{
    for (size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x; s < n_sectors; s += (size_t)gridDim.x * blockDim.x)
        dst[s * 2 + half] = make_uint4((uint32_t)s, 1u, 2u, 3u);
}

// (E) k_decode's PCM flush: `n` consecutive 16-byte stores per lane at the start of a lane-private
//     128-byte line (n = 1: half a 32-byte sector, n = 2: one whole sector)
template <int N>
__global__ void __launch_bounds__(256) k_line_stores(uint4 *dst, size_t n_lines)
{
    for (size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x; l < n_lines; l += (size_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int i = 0; i < N; i++)
            dst[l * 8 + i] = make_uint4((uint32_t)l, 1u, 2u, (uint32_t)i);
    }
}

// (D) wide coalesced 16-byte stores
__global__ void __launch_bounds__(256) k_stream_store(uint4 *dst, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}

template <class F>
static float timed(F f)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    f();                                   // warm
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    f();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

int main()
{
    const size_t bytes = (size_t)4 << 30;          // 4 GiB: 16 x the Infinity Cache
    uint4 *buf;
    uint32_t *out;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 1, bytes));
    const int grid = 256 * 8;
    const size_t n16 = bytes / 16, n_lines = bytes / 128, n_sec = bytes / 32;
    float t;
    t = timed([&] { hipLaunchKernelGGL(k_stream, dim3(grid), dim3(256), 0, 0, buf, n16, out); });
    printf("k_stream        useful %zu B  %.3f ms  %.0f GB/s\n", bytes, t, bytes / t / 1e6);
    t = timed([&] { hipLaunchKernelGGL(k_half_lines, dim3(grid), dim3(256), 0, 0, buf, n_lines, 0, out); });
    printf("k_half_lines(0) useful %zu B  %.3f ms  %.0f GB/s useful\n", bytes / 2, t, bytes / 2 / t / 1e6);
    t = timed([&] { hipLaunchKernelGGL(k_half_lines, dim3(grid), dim3(256), 0, 0, buf, n_lines, 1, out); });
    printf("k_half_lines(1) useful %zu B  %.3f ms  %.0f GB/s useful\n", bytes / 2, t, bytes / 2 / t / 1e6);
    t = timed([&] { hipLaunchKernelGGL(k_stream_store, dim3(grid), dim3(256), 0, 0, buf, n16); });
    printf("k_stream_store  useful %zu B  %.3f ms  %.0f GB/s\n", bytes, t, bytes / t / 1e6);
    t = timed([&] { hipLaunchKernelGGL(k_half_sectors, dim3(grid), dim3(256), 0, 0, buf, n_sec, 0); });
    printf("k_half_sectors(0) useful %zu B  %.3f ms  %.0f GB/s useful\n", bytes / 2, t, bytes / 2 / t / 1e6);
    t = timed([&] { hipLaunchKernelGGL(k_half_sectors, dim3(grid), dim3(256), 0, 0, buf, n_sec, 1); });
    printf("k_half_sectors(1) useful %zu B  %.3f ms  %.0f GB/s useful\n", bytes / 2, t, bytes / 2 / t / 1e6);
    t = timed([&] { hipLaunchKernelGGL(k_line_stores<1>, dim3(grid), dim3(256), 0, 0, buf, n_lines); });
    printf("k_line_stores<1> useful %zu B  %.3f ms  %.1f Gstores/s\n", n_lines * 16, t, n_lines / t / 1e6);
    t = timed([&] { hipLaunchKernelGGL(k_line_stores<2>, dim3(grid), dim3(256), 0, 0, buf, n_lines); });
    printf("k_line_stores<2> useful %zu B  %.3f ms  %.1f Gstores/s\n", n_lines * 32, t, 2 * n_lines / t / 1e6);
    t = timed([&] { hipLaunchKernelGGL(k_line_stores<4>, dim3(grid), dim3(256), 0, 0, buf, n_lines); });
    printf("k_line_stores<4> useful %zu B  %.3f ms  %.1f Gstores/s\n", n_lines * 64, t, 4 * n_lines / t / 1e6);
    CK(hipDeviceSynchronize());
    return 0;
}
