// common.h -- device helpers shared by the gfx950 kernels of libupp_hip.so.
// CDNA4 only: 64-lane wavefronts, DPP row operations, v_readlane.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "../../../include/upp_hip.h"

#define UPP_WAVE 64

// Squared length (x*x) + (y*y) + (z*z) with the contraction nvcc applies to the
// reference sources (t = y*y; t = fma(x,x,t); t = fma(z,z,t)).  Compiled with
// -ffp-contract=off, so the fma placement is exactly what is written here.
// Must stay identical to sumsq3() in oracle/upp_oracle.c.
__device__ __forceinline__ float sumsq3(float x, float y, float z) {
    float t = y * y;
    t = __builtin_fmaf(x, x, t);
    t = __builtin_fmaf(z, z, t);
    return t;
}

// KNN_CUDA's running  ssd += tmp*tmp  over d = 0,1,2  (fma(t,t,acc) chain).
__device__ __forceinline__ float ssd3(float dx, float dy, float dz) {
    float s = dx * dx;  // fma(dx,dx,0) is exact dx*dx
    s = __builtin_fmaf(dy, dy, s);
    s = __builtin_fmaf(dz, dz, s);
    return s;
}

// ---- DPP lane permutes (all lanes active, whole rows valid) ----------------
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}
#define DPP_QUAD_XOR1 0xB1        // quad_perm:[1,0,3,2]
#define DPP_QUAD_XOR2 0x4E        // quad_perm:[2,3,0,1]
#define DPP_ROW_HALF_MIRROR 0x141 // lane i <- lane 7-i inside each 8 lanes
#define DPP_ROW_MIRROR 0x140      // lane i <- lane 15-i inside each 16 lanes
#define DPP_WAVE_SHR1 0x138       // lane i <- lane i-1 across the whole wave

__device__ __forceinline__ uint32_t readlane_u32(uint32_t v, int lane) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}

// Wave-wide max / min of a u32, result uniform (SGPR).  Four DPP steps make
// every row of 16 lanes uniform, then the four rows are combined on the SALU.
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    v = max(v, dpp_u32<DPP_QUAD_XOR1>(v));
    v = max(v, dpp_u32<DPP_QUAD_XOR2>(v));
    v = max(v, dpp_u32<DPP_ROW_HALF_MIRROR>(v));
    v = max(v, dpp_u32<DPP_ROW_MIRROR>(v));
    const uint32_t a = readlane_u32(v, 0), b = readlane_u32(v, 16);
    const uint32_t c = readlane_u32(v, 32), d = readlane_u32(v, 48);
    return max(max(a, b), max(c, d));
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    v = min(v, dpp_u32<DPP_QUAD_XOR1>(v));
    v = min(v, dpp_u32<DPP_QUAD_XOR2>(v));
    v = min(v, dpp_u32<DPP_ROW_HALF_MIRROR>(v));
    v = min(v, dpp_u32<DPP_ROW_MIRROR>(v));
    const uint32_t a = readlane_u32(v, 0), b = readlane_u32(v, 16);
    const uint32_t c = readlane_u32(v, 32), d = readlane_u32(v, 48);
    return min(min(a, b), min(c, d));
}
// wave-wide max of a float (exact under any order): order-preserving u32 image + the u32 reducer
__device__ __forceinline__ float wave_max_f32(float v) {
    const uint32_t bits = __float_as_uint(v);
    const uint32_t key = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
    const uint32_t km = wave_max_u32(key);
    return __uint_as_float((km & 0x80000000u) ? (km & 0x7FFFFFFFu) : ~km);
}
__device__ __forceinline__ float wave_sum_f32(float v) {
    // fixed combination order -> deterministic
    v += __uint_as_float(dpp_u32<DPP_QUAD_XOR1>(__float_as_uint(v)));
    v += __uint_as_float(dpp_u32<DPP_QUAD_XOR2>(__float_as_uint(v)));
    v += __uint_as_float(dpp_u32<DPP_ROW_HALF_MIRROR>(__float_as_uint(v)));
    v += __uint_as_float(dpp_u32<DPP_ROW_MIRROR>(__float_as_uint(v)));
    const float a = __uint_as_float(readlane_u32(__float_as_uint(v), 0));
    const float b = __uint_as_float(readlane_u32(__float_as_uint(v), 16));
    const float c = __uint_as_float(readlane_u32(__float_as_uint(v), 32));
    const float d = __uint_as_float(readlane_u32(__float_as_uint(v), 48));
    return (a + b) + (c + d);
}

// XCD-aware, bijective renumbering of the workgroups of a 1-D grid: workgroups b and b + 8 share an XCD (and its L2), so XCD x gets
// the contiguous run [x n/8, (x+1) n/8) of the work items.  Every row-wise kernel of a Transformer block uses it (row kernels,
// block tail, attention, and the row panels of upp_linear_f32): XCD x then works on the same ~1/8 of the token rows in every
// kernel of the block, and an activation written by one kernel is read by the next from the same L2 instead of the Infinity Cache.
// `UPP_XCD_REMAP` = 0 at build time switches it off for A/B measurements (the mapping never changes results).
#ifndef UPP_XCD_REMAP
#define UPP_XCD_REMAP 1
#endif
__device__ __forceinline__ int xcd_contiguous(int orig, int nwg) {
#if UPP_XCD_REMAP
    const int xcd = orig & 7, q8 = nwg >> 3, rem = nwg & 7;
    return (xcd < rem ? xcd * (q8 + 1) : rem * (q8 + 1) + (xcd - rem) * q8) + (orig >> 3);
#else
    return orig;
#endif
}

// 2-D grids of (tiles of a cloud, cloud): the hardware deals consecutive workgroups to the 8 XCDs in turn, so the tiles of ONE cloud -- which
// all stage the same "other" cloud -- would sit on 8 different L2s and fetch it 8 times (PMC, round 2: Chamfer 5.9 x, kNN 2.7 x its
// algorithmic bytes).  -> (tile, cloud) with every XCD owning whole clouds.
__device__ __forceinline__ void xcd_cloud_tile(int &tile, int &cloud) {
    const int nx = (int)gridDim.x;
    const int lin = xcd_contiguous((int)(blockIdx.y * gridDim.x + blockIdx.x), nx * (int)gridDim.y);
    cloud = lin / nx;
    tile = lin - cloud * nx;
}

// Cooperative global -> LDS copy of n floats by `nthreads` threads.  Each thread keeps 8 independent loads in flight
// per round: a load -> ds_write -> next load loop pays the full ~1 us memory latency on every iteration.
__device__ __forceinline__ void stage_floats(float *dst, const float *__restrict__ src, int n, int tid, int nthreads) {
    for (int i0 = tid; i0 < n; i0 += nthreads * 8) {
        float t[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { const int i = i0 + q * nthreads; t[q] = i < n ? src[i] : 0.0f; }
#pragma unroll
        for (int q = 0; q < 8; ++q) { const int i = i0 + q * nthreads; if (i < n) dst[i] = t[q]; }
    }
}

// Process-wide options (include/upp_hip.h: upp_set_option / upp_get_option) -- the library's ONLY mutable state; every other choice is an
// argument.  Defined in abi.hip (relaxed atomics); the library never reads the environment.
int upp_option(int key);

static inline int upp_launch_status(void) {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// Zero `n` floats on `stream` with a KERNEL.  Never hipMemsetAsync: on this stack (ROCm 7.2.0) a memset NODE of a captured graph works in
// the first replay and writes garbage from the second on (tools/micro/memset_graph_check.py: EMD cost -1.15e37, Chamfer gradients inf) --
// and every entry point of this library may be captured into a step graph.
__global__ __launch_bounds__(256) static void upp_zero_kernel(float *__restrict__ p, long long n) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) *reinterpret_cast<float4 *>(p + i) = make_float4(0.f, 0.f, 0.f, 0.f);          // (p is 16-byte aligned: checked by the caller)
    else for (long long j = i; j < n; ++j) p[j] = 0.0f;
}
static inline void upp_zero_async(float *p, long long n, hipStream_t stream) {
    if (n <= 0) return;
    if (reinterpret_cast<uintptr_t>(p) & 15) {            // an unaligned head (1 ... 3 floats): the scalar path of one extra launch
        long long head = (long long)((16 - (reinterpret_cast<uintptr_t>(p) & 15)) / 4);
        if (head > n) head = n;
        hipLaunchKernelGGL(upp_zero_kernel, dim3(1), dim3(256), 0, stream, p, head);
        p += head; n -= head;
        if (n <= 0) return;
    }
    hipLaunchKernelGGL(upp_zero_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, stream, p, n);
}
