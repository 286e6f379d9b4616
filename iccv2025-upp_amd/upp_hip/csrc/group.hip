// group.hip -- index gathers of the grouping stage and their scatter-add backwards.
//
//  * upp_gather_fwd/bwd: pointnet2_ops gather_points_kernel / gather_points_grad_kernel
//    (sampling_gpu.cu), behind pointnet2_utils.gather_operation, reference utils/misc.py:19.
//  * upp_group_fwd/bwd: the torch indexing of Group.forward, reference
//    models/Point_MAE_unify.py:73-88 (flat gather + centre subtraction).
// All four are pure HBM-bound byte movers: flat grid-stride kernels, one
// element per lane, coalesced on the dense side.
#include "common.h"

namespace {

constexpr int kBlock = 256;

inline int grid_for(long long total) {
    long long g = (total + kBlock - 1) / kBlock;
    if (g > 2048) g = 2048;  // 256 CUs x 8 blocks, grid-stride the rest
    if (g < 1) g = 1;
    return (int)g;
}

// out[b,c,j] = feat[b,c,idx[b,j]]
__global__ void gather_fwd_kernel(const float *__restrict__ feat, const int32_t *__restrict__ idx, float *__restrict__ out,
                                  int C, int N, int M, long long total) {
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
        const int j = (int)(i % M);
        const long long bc = i / M;
        const long long b = bc / C;
        out[i] = feat[bc * N + idx[b * M + j]];
    }
}

// grad_feat[b,c,idx[b,j]] += grad_out[b,c,j]
__global__ void gather_bwd_kernel(const float *__restrict__ grad_out, const int32_t *__restrict__ idx,
                                  float *__restrict__ grad_feat, int C, int N, int M, long long total) {
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
        const int j = (int)(i % M);
        const long long bc = i / M;
        const long long b = bc / C;
        atomicAdd(&grad_feat[bc * N + idx[b * M + j]], grad_out[i]);
    }
}

// out[b,g,k,c] = xyz[b, idx[b,g,k], c] - center[b,g,c]
__global__ void group_fwd_kernel(const float *__restrict__ xyz, const float *__restrict__ center,
                                 const int64_t *__restrict__ idx, float *__restrict__ out, int N, int G, int K,
                                 long long total) {
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
        const int c = (int)(i % 3);
        const long long e = i / 3;            // (b,g,k)
        const long long bg = e / K;           // (b,g)
        const long long b = bg / G;
        const long long r = idx[e];
        out[i] = xyz[(b * N + r) * 3 + c] - center[bg * 3 + c];
    }
}

// grad_xyz[b, idx[b,g,k], c] += grad_out[b,g,k,c]
__global__ void group_bwd_xyz_kernel(const float *__restrict__ grad_out, const int64_t *__restrict__ idx,
                                     float *__restrict__ grad_xyz, int N, int G, int K, long long total) {
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
        const int c = (int)(i % 3);
        const long long e = i / 3;
        const long long b = e / ((long long)G * K);
        const long long r = idx[e];
        atomicAdd(&grad_xyz[(b * N + r) * 3 + c], grad_out[i]);
    }
}

// grad_center[b,g,c] = -sum_k grad_out[b,g,k,c]   (k ascending: deterministic)
__global__ void group_bwd_center_kernel(const float *__restrict__ grad_out, float *__restrict__ grad_center, int K,
                                        long long total) {
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
        const int c = (int)(i % 3);
        const long long bg = i / 3;
        float s = 0.0f;
        for (int k = 0; k < K; ++k) s += grad_out[(bg * K + k) * 3 + c];
        grad_center[i] = -s;
    }
}

// Backward of the FPS centre gather (reference utils/misc.py:19 gather_operation under autograd: stage 2 and the pre-task recipe, where the
// sampled coordinates carry a gradient): g_xyz (B,N,3) = 0 everywhere except rows idx[b][j], which receive g_centers[b][j].  ONE launch, one
// workgroup per cloud: the cloud's slice is zeroed, then the M rows are added -- torch.zeros + an int32 -> int64 index copy + the scatter
// kernel were three launches per FPS call.  FPS indices of a cloud are distinct unless it has fewer distinct points than samples; equal
// indices are added atomically (two addends commute exactly; three or more depend on order in the last bit, as torch's index_add does).
__global__ __launch_bounds__(256) void fps_gather_bwd_kernel(const float *__restrict__ g_centers, const int32_t *__restrict__ idx,
                                                             float *__restrict__ g_xyz, int N, int M) {
    const int b = blockIdx.x;
    float *gx = g_xyz + (size_t)b * N * 3;
    for (int i = threadIdx.x; i < N * 3; i += 256) gx[i] = 0.0f;
    __syncthreads();                                  // (the zeros of this workgroup's own slice are visible to its own atomics)
    for (int i = threadIdx.x; i < M * 3; i += 256) {
        const int j = i / 3, c = i - 3 * j;
        const int r = idx[(size_t)b * M + j];
        if (r >= 0 && r < N) atomicAdd(&gx[(size_t)r * 3 + c], g_centers[((size_t)b * M + j) * 3 + c]);
    }
}

}  // namespace

extern "C" int upp_gather_fwd(const float *feat, const int32_t *idx, float *out, int B, int C, int N, int M, void *stream) {
    if (!feat || !idx || !out || B < 0 || C < 0 || N < 1 || M < 0) return UPP_E_BADARG;
    const long long total = (long long)B * C * M;
    if (total == 0) return 0;
    hipLaunchKernelGGL(gather_fwd_kernel, dim3(grid_for(total)), dim3(kBlock), 0, (hipStream_t)stream, feat, idx, out, C, N, M, total);
    return upp_launch_status();
}

extern "C" int upp_gather_bwd(const float *grad_out, const int32_t *idx, float *grad_feat, int B, int C, int N, int M, void *stream) {
    if (!grad_out || !idx || !grad_feat || B < 0 || C < 0 || N < 1 || M < 0) return UPP_E_BADARG;
    const long long total = (long long)B * C * M;
    if (total == 0) return 0;
    hipLaunchKernelGGL(gather_bwd_kernel, dim3(grid_for(total)), dim3(kBlock), 0, (hipStream_t)stream, grad_out, idx, grad_feat, C, N, M, total);
    return upp_launch_status();
}

extern "C" int upp_fps_gather_bwd(const float *g_centers, const int32_t *idx, float *g_xyz, int B, int N, int M, void *stream) {
    if (!g_centers || !idx || !g_xyz || B < 0 || N < 1 || M < 0) return UPP_E_BADARG;
    if (B == 0) return 0;
    hipLaunchKernelGGL(fps_gather_bwd_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, g_centers, idx, g_xyz, N, M);
    return upp_launch_status();
}

extern "C" int upp_group_fwd(const float *xyz, const float *center, const int64_t *idx, float *out, int B, int N, int G, int K, void *stream) {
    if (!xyz || !center || !idx || !out || B < 0 || N < 1 || G < 0 || K < 0) return UPP_E_BADARG;
    const long long total = (long long)B * G * K * 3;
    if (total == 0) return 0;
    hipLaunchKernelGGL(group_fwd_kernel, dim3(grid_for(total)), dim3(kBlock), 0, (hipStream_t)stream, xyz, center, idx, out, N, G, K, total);
    return upp_launch_status();
}

extern "C" int upp_group_bwd(const float *grad_out, const int64_t *idx, float *grad_xyz, float *grad_center, int B, int N, int G, int K, void *stream) {
    if (!grad_out || !idx || B < 0 || N < 1 || G < 0 || K < 0) return UPP_E_BADARG;
    const long long total = (long long)B * G * K * 3;
    if (total == 0) return 0;
    if (grad_xyz)
        hipLaunchKernelGGL(group_bwd_xyz_kernel, dim3(grid_for(total)), dim3(kBlock), 0, (hipStream_t)stream, grad_out, idx, grad_xyz, N, G, K, total);
    if (grad_center) {
        const long long tc = (long long)B * G * 3;
        hipLaunchKernelGGL(group_bwd_center_kernel, dim3(grid_for(tc)), dim3(kBlock), 0, (hipStream_t)stream, grad_out, grad_center, K, tc);
    }
    return upp_launch_status();
}
