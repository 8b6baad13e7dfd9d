// tools/probe/init_time.cpp -- what a process pays before its first decode: HIP runtime, device context, the library's
// code object (first kernel launch), pinned and device allocations.  Diagnostic; built and run by r05_init_time.sh.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <time.h>
#include "dvda_mlp_hip.h"
static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
int main()
{
    double t = now(), t1;
#define LAP(what) t1 = now(); printf("%-46s %8.1f ms\n", what, t1 - t); t = t1
    hipInit(0); LAP("hipInit");
    hipSetDevice(0); hipFree(0); LAP("hipSetDevice + hipFree(0) (device context)");
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking); LAP("hipStreamCreate");
    uint32_t out[512 * 4];
    dvda_mlp_hip_selftest_huff(0, out); LAP("first kernel of the library (code object load)");
    dvda_mlp_hip_selftest_huff(0, out); LAP("the same again");
    void *h = 0; hipHostMalloc(&h, 64u << 20, hipHostMallocDefault); LAP("hipHostMalloc 64 MB");
    void *h2 = 0; hipHostMalloc(&h2, 256u << 20, hipHostMallocDefault); LAP("hipHostMalloc 256 MB");
    void *d = 0; hipMalloc(&d, 512u << 20); LAP("hipMalloc 512 MB");
    hipMemsetAsync(d, 0, 512u << 20, s); hipStreamSynchronize(s); LAP("memset 512 MB + sync");
    hipMemcpyAsync(d, h2, 256u << 20, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); LAP("H2D 256 MB pinned");
    hipMemcpyAsync(h2, d, 256u << 20, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); LAP("D2H 256 MB pinned");
    dvda_mlp_hip_ctx *ctx = 0;
    int rc = dvda_mlp_hip_create(&ctx, 0, 64, 1u << 20); LAP("dvda_mlp_hip_create");
    printf("rc %d\n", rc);
    hipHostFree(h); hipHostFree(h2); LAP("hipHostFree both");
    hipFree(d); LAP("hipFree");
    if (ctx) dvda_mlp_hip_destroy(ctx);
    LAP("destroy ctx");
    return 0;
}
