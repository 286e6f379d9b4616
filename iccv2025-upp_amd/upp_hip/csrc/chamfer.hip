// chamfer.hip -- bidirectional nearest-neighbour squared distance + arg-min, and its
// gradient, for gfx950.  Replaces chamfer.forward / chamfer.backward of the
// reference torch extension (extensions/chamfer_dist/chamfer.cu:15-145, :173-201;
// bindings chamfer_cuda.cpp:36-39).
//
// Forward mapping: a workgroup owns 64 query points of one cloud (one per lane)
// and its 4 waves split the other cloud four ways, so B=32, n=1024 gives 2048
// waves (2 per SIMD) instead of the 512 the CUDA launch shape would.  The other
// cloud is staged through LDS as float4 and read with one broadcast ds_read_b128
// per point.  Strict '<' inside a wave (ascending index) and a (distance, index)
// lexicographic combine across waves give "lowest index among equal minima",
// which is what chamfer.cu's tiles (:47,57 strict '<', :137 strict '>') produce.
// Per-pair arithmetic is the contracted form nvcc emits for chamfer.cu:42-45.
#include "common.h"

namespace {

constexpr int kTile = 2048;  // points of the other cloud per LDS tile (32 KiB)
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void chamfer_dir_kernel(const float *__restrict__ xyz1, const float *__restrict__ xyz2,
                                                          float *__restrict__ dist, int32_t *__restrict__ idx, int n, int m) {
    // two points per 32-byte LDS record [x0 x1 y0 y1 z0 z1 - -]: the pair loop runs on packed f32 (v_pk_add / mul / fma_f32:
    // IEEE per element, the same contracted form per pair), 3 + 3 instead of 6 + 3 VALU instructions per pair
    __shared__ __attribute__((aligned(16))) float tile[kTile / 2][8];
    __shared__ float cd[4][64];
    __shared__ int ci[4][64];
    // wave-uniform wave index: the pair loop's counter, its bounds and the candidate index then live in SGPRs
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int tile_x, b;
    xcd_cloud_tile(tile_x, b);                            // whole clouds per XCD: the other cloud is fetched into ONE L2
    const int j = tile_x * 64 + lane;
    const int jc = j < n ? j : n - 1;
    const float *a = xyz1 + ((size_t)b * n + jc) * 3;
    const float x1 = a[0], y1 = a[1], z1 = a[2];
    const f32x2 X1 = {x1, x1}, Y1 = {y1, y1}, Z1 = {z1, z1};
    const float *other = xyz2 + (size_t)b * m * 3;

    float best = __builtin_inff();
    int besti = 0;
    for (int c0 = 0; c0 < m; c0 += kTile) {
        const int len = min(kTile, m - c0), len2 = (len + 1) & ~1;
        __syncthreads();
        for (int i0 = threadIdx.x; i0 < len2; i0 += 256 * 4) {   // 12 independent loads in flight per thread
            float t[4][3];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = min(i0 + q * 256, len - 1);
                const float *s = other + (size_t)(c0 + i) * 3;
                t[q][0] = s[0]; t[q][1] = s[1]; t[q][2] = s[2];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i0 + q * 256;
                if (i < len2) {
                    const bool real = i < len;            // the odd slot past the end: a point at infinity (d = inf never wins)
                    float *r = &tile[i >> 1][i & 1];
                    r[0] = real ? t[q][0] : __builtin_inff(); r[2] = real ? t[q][1] : __builtin_inff(); r[4] = real ? t[q][2] : __builtin_inff();
                }
            }
        }
        __syncthreads();
        const int seg = ((((len2 >> 1) + 3) >> 2));           // pairs per wave
        const int p0 = wave * seg;
        const int p1 = min(len2 >> 1, p0 + seg);
#pragma unroll 4
        for (int p = p0; p < p1; ++p) {
            const float4 xy = *reinterpret_cast<const float4 *>(&tile[p][0]);
            const float2 zz = *reinterpret_cast<const float2 *>(&tile[p][4]);
            const f32x2 dx = f32x2{xy.x, xy.y} - X1, dy = f32x2{xy.z, xy.w} - Y1, dz = f32x2{zz.x, zz.y} - Z1;
            f32x2 d = dy * dy;                                 // sumsq3 per element: t = dy*dy; fma(dx,dx,t); fma(dz,dz,t)
            d = __builtin_elementwise_fma(dx, dx, d);
            d = __builtin_elementwise_fma(dz, dz, d);
            const bool lt0 = d[0] < best;
            besti = lt0 ? c0 + 2 * p : besti;
            best = lt0 ? d[0] : best;
            const bool lt1 = d[1] < best;
            besti = lt1 ? c0 + 2 * p + 1 : besti;
            best = lt1 ? d[1] : best;
        }
    }
    cd[wave][lane] = best;
    ci[wave][lane] = besti;
    __syncthreads();
    if (wave == 0 && j < n) {
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float d = cd[w][lane];
            const int i = ci[w][lane];
            const bool take = (d < best) || (d == best && i < besti);
            best = take ? d : best;
            besti = take ? i : besti;
        }
        dist[(size_t)b * n + j] = best;
        idx[(size_t)b * n + j] = besti;
    }
}

// Both launches of chamfer_dist_grad_kernel (chamfer.cu:215-222) in one grid:
// element e < B*n is point j of xyz1 (direction 1), otherwise a point of xyz2.
__global__ void chamfer_grad_kernel(const float *__restrict__ xyz1, const float *__restrict__ xyz2,
                                    const int32_t *__restrict__ idx1, const int32_t *__restrict__ idx2,
                                    const float *__restrict__ gd1, const float *__restrict__ gd2, float *__restrict__ g1,
                                    float *__restrict__ g2, int B, int n, int m) {
    const long long total = (long long)B * (n + m);
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const bool dir1 = e < (long long)B * n;
        const long long r = dir1 ? e : e - (long long)B * n;
        const int na = dir1 ? n : m, nb = dir1 ? m : n;
        const long long b = r / na;
        const float *pa = (dir1 ? xyz1 : xyz2) + r * 3;
        const int j2 = (dir1 ? idx1 : idx2)[r];
        const float *pb = (dir1 ? xyz2 : xyz1) + (b * nb + j2) * 3;
        const float g = (dir1 ? gd1 : gd2)[r] * 2;
        float *ga = (dir1 ? g1 : g2) + r * 3;
        float *gb = (dir1 ? g2 : g1) + (b * nb + j2) * 3;
        const float vx = g * (pa[0] - pb[0]), vy = g * (pa[1] - pb[1]), vz = g * (pa[2] - pb[2]);
        atomicAdd(ga + 0, vx); atomicAdd(ga + 1, vy); atomicAdd(ga + 2, vz);
        atomicAdd(gb + 0, -vx); atomicAdd(gb + 1, -vy); atomicAdd(gb + 2, -vz);
    }
}

// The same sums with ONE workgroup per cloud pair and both gradient arrays in the LDS: every global operand is read once, every gradient
// written once (algorithmic bytes; the grid-wide atomics above touch each line from up to 8 L2s: PMC 5.7 x).  The accumulation order
// inside the LDS is the order the LDS unit retires the adds in -- as unordered as the reference's global atomicAdd.
__global__ __launch_bounds__(1024) void chamfer_grad_cloud_kernel(const float *__restrict__ xyz1, const float *__restrict__ xyz2,
                                                                  const int32_t *__restrict__ idx1, const int32_t *__restrict__ idx2,
                                                                  const float *__restrict__ gd1, const float *__restrict__ gd2,
                                                                  float *__restrict__ g1, float *__restrict__ g2, int n, int m) {
    extern __shared__ float acc[];                       // [n * 3 | m * 3]
    const int b = blockIdx.x, tid = threadIdx.x;
    float *a1 = acc, *a2 = acc + (size_t)n * 3;
    for (int i = tid; i < (n + m) * 3; i += 1024) acc[i] = 0.0f;
    __syncthreads();
    const float *p1 = xyz1 + (size_t)b * n * 3, *p2 = xyz2 + (size_t)b * m * 3;
    for (int r = tid; r < n + m; r += 1024) {
        const bool dir1 = r < n;
        const int j = dir1 ? r : r - n;
        const float *pa = (dir1 ? p1 : p2) + (size_t)j * 3;
        const int j2 = dir1 ? idx1[(size_t)b * n + j] : idx2[(size_t)b * m + j];
        const float *pb = (dir1 ? p2 : p1) + (size_t)j2 * 3;
        const float g = (dir1 ? gd1[(size_t)b * n + j] : gd2[(size_t)b * m + j]) * 2;
        const float vx = g * (pa[0] - pb[0]), vy = g * (pa[1] - pb[1]), vz = g * (pa[2] - pb[2]);
        float *ga = (dir1 ? a1 : a2) + j * 3, *gb = (dir1 ? a2 : a1) + j2 * 3;
        __hip_atomic_fetch_add(ga + 0, vx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(ga + 1, vy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(ga + 2, vz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(gb + 0, -vx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(gb + 1, -vy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(gb + 2, -vz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    float *o1 = g1 + (size_t)b * n * 3, *o2 = g2 + (size_t)b * m * 3;
    for (int i = tid; i < n * 3; i += 1024) o1[i] = a1[i];
    for (int i = tid; i < m * 3; i += 1024) o2[i] = a2[i];
}

// Loss reduction of the Chamfer modules (reference extensions/chamfer_dist/__init__.py:44-84): ChamferDistanceL1 = (mean sqrt d1 + mean
// sqrt d2) / 2, ChamferDistanceL2 = mean d1 + mean d2, as ONE partial-sum launch + a finalize (torch: sqrt x 2, mean x 2, add, div forward
// and the same again backward, per term).  fac1 / fac2 receive d loss / d dist element-wise (coef / (count 2 sqrt d) resp. coef / count):
// the backward of the module is upp_chamfer_bwd on upstream x fac.  Sums in a fixed order (deterministic).
constexpr int kLossBlocks = 128;
__global__ __launch_bounds__(256) void chamfer_loss_part_kernel(const float *__restrict__ d1, const float *__restrict__ d2, long long n1, long long n2,
                                                               int l1, float c1, float c2, float *__restrict__ fac1, float *__restrict__ fac2,
                                                               float *__restrict__ part) {
    __shared__ float red[2][4];
    float s1 = 0.0f, s2 = 0.0f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n1; i += (long long)kLossBlocks * 256) {
        const float d = d1[i], r = l1 ? sqrtf(d) : d;
        s1 += r;
        fac1[i] = l1 ? c1 * 0.5f / r : c1;
    }
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long long)kLossBlocks * 256) {
        const float d = d2[i], r = l1 ? sqrtf(d) : d;
        s2 += r;
        fac2[i] = l1 ? c2 * 0.5f / r : c2;
    }
    s1 = wave_sum_f32(s1); s2 = wave_sum_f32(s2);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wave] = s1; red[1][wave] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[blockIdx.x * 2 + 0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        part[blockIdx.x * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}
__global__ __launch_bounds__(64) void chamfer_loss_final_kernel(const float *__restrict__ part, float c1, float c2, float *__restrict__ out) {
    float s1 = 0.0f, s2 = 0.0f;
    for (int i = threadIdx.x; i < kLossBlocks; i += 64) { s1 += part[i * 2]; s2 += part[i * 2 + 1]; }
    s1 = wave_sum_f32(s1); s2 = wave_sum_f32(s2);
    if (threadIdx.x == 0) out[0] = c1 * s1 + c2 * s2;
}

}  // namespace

extern "C" int upp_chamfer_fwd(const float *xyz1, const float *xyz2, float *dist1, float *dist2, int32_t *idx1, int32_t *idx2,
                               int B, int n, int m, void *stream) {
    if (!xyz1 || !xyz2 || !dist1 || !dist2 || !idx1 || !idx2 || B < 0 || n < 1 || m < 1) return UPP_E_BADARG;
    if (B == 0) return 0;
    if (B > 65535) return UPP_E_RANGE;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(chamfer_dir_kernel, dim3((n + 63) / 64, B), dim3(256), 0, st, xyz1, xyz2, dist1, idx1, n, m);
    hipLaunchKernelGGL(chamfer_dir_kernel, dim3((m + 63) / 64, B), dim3(256), 0, st, xyz2, xyz1, dist2, idx2, m, n);
    return upp_launch_status();
}

extern "C" int upp_chamfer_bwd(const float *xyz1, const float *xyz2, const int32_t *idx1, const int32_t *idx2,
                               const float *grad_dist1, const float *grad_dist2, float *g1, float *g2, int B, int n, int m,
                               void *stream) {
    if (!xyz1 || !xyz2 || !idx1 || !idx2 || !grad_dist1 || !grad_dist2 || !g1 || !g2 || B < 0 || n < 1 || m < 1)
        return UPP_E_BADARG;
    if (B == 0) return 0;
    const size_t lds = (size_t)(n + m) * 3 * sizeof(float);
    if (lds <= 64 * 1024 && B <= 65535) {                // both gradient arrays of a cloud pair in the LDS (n + m <= 5,461)
        hipLaunchKernelGGL(chamfer_grad_cloud_kernel, dim3((unsigned)B), dim3(1024), lds, (hipStream_t)stream, xyz1, xyz2, idx1, idx2, grad_dist1,
                           grad_dist2, g1, g2, n, m);
        return upp_launch_status();
    }
    const long long total = (long long)B * (n + m);
    long long grid = (total + 255) / 256;
    if (grid > 2048) grid = 2048;
    upp_zero_async(g1, 3LL * B * n, (hipStream_t)stream);            // (a kernel, not a memset node: common.h)
    upp_zero_async(g2, 3LL * B * m, (hipStream_t)stream);
    hipLaunchKernelGGL(chamfer_grad_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, xyz1, xyz2, idx1, idx2,
                       grad_dist1, grad_dist2, g1, g2, B, n, m);
    return upp_launch_status();
}

extern "C" long long upp_chamfer_loss_work_floats(void) { return 2 * kLossBlocks; }

extern "C" int upp_chamfer_loss(const float *dist1, const float *dist2, int B, int n, int m, int l1, float *loss, float *fac1, float *fac2,
                                float *work, void *stream) {
    if (!dist1 || !dist2 || !loss || !fac1 || !fac2 || !work || B < 1 || n < 1 || m < 1) return UPP_E_BADARG;
    const float coef = l1 ? 0.5f : 1.0f;
    const float c1 = coef / ((float)B * (float)n), c2 = coef / ((float)B * (float)m);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(chamfer_loss_part_kernel, dim3(kLossBlocks), dim3(256), 0, st, dist1, dist2, (long long)B * n, (long long)B * m, l1 ? 1 : 0, c1, c2,
                       fac1, fac2, work);
    hipLaunchKernelGGL(chamfer_loss_final_kernel, dim3(1), dim3(64), 0, st, work, c1, c2, loss);
    return upp_launch_status();
}
