// mlp_bounds.h -- the range-checked build (DVDA_BOUNDS): every index into a workspace the library owns goes through
// DVDA_AT(pointer, index, capacity, tag).  In the shipped library that is plain `pointer[index]`; built with
// -DDVDA_BOUNDS (libdvd-audio_amd/_build.py: build_bounds) an index at or past the capacity is counted, the first one
// is remembered (tag, index, capacity), and the access goes to a sink instead of memory the kernel does not own --
// so a kernel that would walk out of its arrays says so instead of faulting (or silently reading a neighbour's
// data).  dvda_mlp_hip_bounds_violations() reads the count; dvda_mlp_hip_destroy prints it when
// DVDA_BOUNDS_REPORT is set.  tests/test_gpu_soak.py runs the reuse soak on this build and expects zero.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mlp {

// capacities of the context's workspaces, in elements of their own type (what the kernels may index)
struct WsCaps {
    uint64_t res;          // int32 words of chain planes
    uint64_t brec;         // dwords of block records
    uint64_t frec;         // dwords of access-unit records
    uint64_t fb;           // int32 words of sequential-pass frame buffers
    uint32_t max_seg;      // segments (seg, seg_status, seg_rows, yield, lane_seg, def_list, head_list; plan/fbase: + 1)
    uint32_t max_streams;  // streams (streams, seq_list)
    uint32_t lanes;        // workspace lanes (seg_meta; fir_ws x 48, mat_ws x 30, iir_ws x 128)
    uint32_t pad;
};

#if defined(DVDA_BOUNDS)
__device__ unsigned long long g_bounds[4];      // violations, then the first one's tag / index / capacity
__device__ uint32_t g_sink[64];

__device__ __attribute__((noinline)) void bounds_hit(uint32_t tag, uint64_t i, uint64_t cap)
{
    if (atomicAdd(&g_bounds[0], 1ull) == 0) {
        g_bounds[1] = tag;
        g_bounds[2] = i;
        g_bounds[3] = cap;
    }
}

template <class T>
__device__ __forceinline__ T &bounds_at(T *p, uint64_t i, uint64_t cap, uint32_t tag)
{
    static_assert(sizeof(T) <= sizeof(g_sink), "sink too small");
    if (i >= cap) {
        bounds_hit(tag, i, cap);
        return *reinterpret_cast<T *>(reinterpret_cast<uintptr_t>(&g_sink[0]));
    }
    return p[i];
}
// a range [i, i + n) a kernel is about to touch through a raw pointer (asm stores, record writers)
__device__ __forceinline__ bool bounds_ok(uint64_t i, uint64_t n, uint64_t cap, uint32_t tag)
{
    if (i + n > cap) {
        bounds_hit(tag, i + n, cap);
        return false;
    }
    return true;
}
#define DVDA_AT(p, i, cap, tag) mlp::bounds_at((p), (uint64_t)(i), (uint64_t)(cap), (uint32_t)(tag))
#define DVDA_RANGE_OK(i, n, cap, tag) mlp::bounds_ok((uint64_t)(i), (uint64_t)(n), (uint64_t)(cap), (uint32_t)(tag))
#else
#define DVDA_AT(p, i, cap, tag) ((p)[(i)])
#define DVDA_RANGE_OK(i, n, cap, tag) true
#endif

// tags (which array): 1x k_decode, 2x chain passes, 3x index / check
enum : uint32_t {
    BT_SEG = 10, BT_STREAMS, BT_FBASE, BT_LANE_SEG, BT_LIST, BT_PLAN, BT_STATUS, BT_ROWS, BT_YIELD, BT_CHECK,
    BT_FIR, BT_META, BT_MAT, BT_IIR, BT_RES, BT_BREC, BT_FREC, BT_FB,
    BT_C_DEF = 40, BT_C_HEAD, BT_C_PLAN, BT_C_RES, BT_C_BREC, BT_C_FREC, BT_C_FIR, BT_C_META, BT_C_STATUS, BT_C_SEG,
};

} // namespace mlp
