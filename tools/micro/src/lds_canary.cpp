// Does a co-resident split-bf16 Linear workgroup write into another workgroup's LDS?  A canary kernel (one workgroup per CU slot, LDS_BYTES
// of dynamic LDS filled with a pattern) re-reads its LDS for ~1 ms while upp_linear_sb_f32 runs on a second stream; mismatches are counted.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/src/lds_canary.cpp -o tools/micro/bin/lds_canary -Liccv2025-upp_amd/upp_hip/lib -lupp_hip -Wl,-rpath,$PWD/iccv2025-upp_amd/upp_hip/lib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
extern "C" long long upp_linear_sb_planes_bytes(int N, int K);
extern "C" int upp_linear_sb_prep(const float *W, long long ldw, int N, int K, int transposed, void *planes, void *stream);
extern "C" int upp_linear_sb_f32(const float *A, long long lda, const void *planes, const float *bias, float *C, long long ldc, float *aux,
                                 long long ldaux, int M, int N, int K, int epilogue, int tile, void *stream);
extern "C" int upp_linear_wgrad_grouped_sb_rows(int count, const int *M, const int *N, const int *K, int *rows);
extern "C" int upp_linear_wgrad_grouped_sb(const float *const *G, const long long *ldg, const float *const *X, const long long *ldx, float *const *partials,
                                           const int *M, const int *N, const int *K, const int *rows, int count, void *stream);
extern "C" int upp_linear_f32(const float *A, long long lda, const float *W, long long ldw, const float *bias, float *C, long long ldc, float *aux,
                              long long ldaux, int M, int N, int K, int epilogue, int tile, void *stream);

__global__ void canary(unsigned *bad, int words, int rounds) {
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < words; i += blockDim.x) lds[i] = 0xC0DE0000u ^ (unsigned)i ^ (blockIdx.x << 20);
    __syncthreads();
    unsigned n = 0;
    for (int r = 0; r < rounds; ++r) {
        for (int i = threadIdx.x; i < words; i += blockDim.x) n += lds[i] != (0xC0DE0000u ^ (unsigned)i ^ (blockIdx.x << 20));
        __builtin_amdgcn_s_sleep(8);
    }
    if (n) atomicAdd(bad, n);
}

// packed-f32 canary: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 against the scalar instructions on the same operands, many times
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void pk_canary(unsigned *bad, int rounds) {
    const float a0 = 0.3f + 1e-3f * threadIdx.x, a1 = -0.7f + 3e-3f * threadIdx.x, b0 = 0.11f + 7e-4f * blockIdx.x, b1 = 0.57f - 1e-4f * blockIdx.x;
    unsigned n = 0;
    for (int r = 0; r < rounds; ++r) {
        const float c0 = 1e-3f * r, c1 = -2e-3f * r;
        f32x2 a = {a0 + c0, a1 + c1}, b = {b0, b1}, c = {c0, c1}, d, m, s;
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(m) : "v"(a), "v"(b));
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(s) : "v"(a), "v"(b));
        float e0, e1, m0, m1, s0, s1;
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e0) : "v"(a[0]), "v"(b[0]), "v"(c[0]));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e1) : "v"(a[1]), "v"(b[1]), "v"(c[1]));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m0) : "v"(a[0]), "v"(b[0]));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m1) : "v"(a[1]), "v"(b[1]));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(s0) : "v"(a[0]), "v"(b[0]));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(s1) : "v"(a[1]), "v"(b[1]));
        n += (d[0] != e0) + (d[1] != e1) + (m[0] != m0) + (m[1] != m1) + (s[0] != s0) + (s[1] != s1);
        // the forms of csrc/fps.hip: a - {x, x} with the subtrahend broadcast from one half of a VGPR / SGPR pair
        f32x2 q0, q1, q2, q3;
        const f32x2 ub = {b0 + c0, b1 - c1};
        unsigned long long sb_ = __builtin_amdgcn_readfirstlane(__float_as_uint(ub[0])) | ((unsigned long long)__builtin_amdgcn_readfirstlane(__float_as_uint(ub[1])) << 32);
        const float u0 = __uint_as_float((unsigned)sb_), u1 = __uint_as_float((unsigned)(sb_ >> 32));
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(q0) : "v"(a), "v"(b));
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(q1) : "v"(a), "v"(b));
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(q2) : "v"(a), "s"(sb_));
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(q3) : "v"(a), "s"(sb_));
        float w[8];
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(w[0]) : "v"(a[0]), "v"(b[0]));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(w[1]) : "v"(a[1]), "v"(b[0]));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(w[2]) : "v"(a[0]), "v"(b[1]));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(w[3]) : "v"(a[1]), "v"(b[1]));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(w[4]) : "v"(a[0]), "v"(u0));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(w[5]) : "v"(a[1]), "v"(u0));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(w[6]) : "v"(a[0]), "v"(u1));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(w[7]) : "v"(a[1]), "v"(u1));
        // further forms: src0's high half into the low lane (op_sel:[1,0]), and mul / fma with op_sel:[0,1]
        f32x2 q4, q5, q6;
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(q4) : "v"(a), "v"(b));
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(q5) : "v"(a), "v"(b));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(q6) : "v"(a), "v"(b), "v"(c));
        float z[6];
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(z[0]) : "v"(a[1]), "v"(b[0]));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(z[1]) : "v"(a[1]), "v"(b[1]));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(z[2]) : "v"(a[0]), "v"(b[1]));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(z[3]) : "v"(a[1]), "v"(b[1]));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(z[4]) : "v"(a[0]), "v"(b[1]), "v"(c[0]));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(z[5]) : "v"(a[1]), "v"(b[1]), "v"(c[1]));
        const unsigned f4 = (q4[0] != z[0]) + (q4[1] != z[1]), f5 = (q5[0] != z[2]) + (q5[1] != z[3]), f6 = (q6[0] != z[4]) + (q6[1] != z[5]);
        if (f4) atomicAdd(bad + 7, f4);
        if (f5) atomicAdd(bad + 8, f5);
        if (f6) atomicAdd(bad + 9, f6);
        const unsigned f0 = (q0[0] != w[0]) + (q0[1] != w[1]), f1 = (q1[0] != w[2]) + (q1[1] != w[3]), f2 = (q2[0] != w[4]) + (q2[1] != w[5]), f3 = (q3[0] != w[6]) + (q3[1] != w[7]);
        if (f0) atomicAdd(bad + 1, f0);
        if (f1) atomicAdd(bad + 2, f1);
        if (f2) atomicAdd(bad + 3, f2);
        if (f3) atomicAdd(bad + 4, f3);
        if (f1 && atomicAdd(bad + 6, 1u) < 6) printf("VGPR op_sel: a %a %a  b %a %a  got %a %a  want %a %a  (a0-b0 %a, a1-b0 %a) lane %d round %d\n", a[0], a[1], b[0], b[1], q1[0], q1[1], w[2], w[3], w[0], w[1], threadIdx.x & 63, r);
        if ((f2 | f3) && atomicAdd(bad + 5, 1u) < 4) printf("a %a %a  s %a %a  q2 %a %a  want %a %a   q3 %a %a want %a %a\n", a[0], a[1], u0, u1, q2[0], q2[1], w[4], w[5], q3[0], q3[1], w[6], w[7]);
    }
    if (n) atomicAdd(bad, n);
}

int main(int argc, char **argv) {
    const int M = 4096, N = argc > 1 ? atoi(argv[1]) : 384, K = argc > 2 ? atoi(argv[2]) : 1536, tile = argc > 3 ? (int)strtol(argv[3], nullptr, 16) : 0;
    const int lds_bytes = argc > 4 ? atoi(argv[4]) : 29 * 1024, use_sb = argc > 5 ? atoi(argv[5]) : 1;
    float *A, *W, *C; void *planes; unsigned *bad;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&W, (size_t)N * K * 4); hipMalloc(&C, (size_t)M * N * 4);
    hipMalloc(&planes, upp_linear_sb_planes_bytes(N, K)); hipMalloc(&bad, 64);
    std::vector<float> h((size_t)M * K, 0.5f);
    hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
    hipMemset(bad, 0, 64);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    upp_linear_sb_prep(W, K, N, K, 0, planes, s2);
    hipFuncSetAttribute(reinterpret_cast<const void *>(canary), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipDeviceSynchronize();
    for (int it = 0; it < 10; ++it) {
        if (lds_bytes > 0) hipLaunchKernelGGL(canary, dim3(256), dim3(256), lds_bytes, s1, bad, lds_bytes / 4, 400);
        else hipLaunchKernelGGL(pk_canary, dim3(1024), dim3(256), 0, s1, bad, 40000);
        for (int j = 0; j < 40; ++j) {
            if (use_sb == 2) break;
            if (use_sb == 3) {              // split-bf16 weight gradient: dW (N,K) = C^T A over the M rows (bf16 MFMA, plain loads + ds_write, no LDS-DMA)
                static float *part = nullptr; static int rows = 0;
                const float *Gp = C; const float *Xp = A; long long ldg = N, ldx = K;
                if (!part) { upp_linear_wgrad_grouped_sb_rows(1, &M, &N, &K, &rows); hipMalloc(&part, (size_t)((M + rows - 1) / rows) * N * K * 4); }
                int rc3 = upp_linear_wgrad_grouped_sb(&Gp, &ldg, &Xp, &ldx, &part, &M, &N, &K, &rows, 1, s2);
                if (rc3) { printf("launch rc %d\n", rc3); return 1; }
                continue;
            }
            int rc = use_sb ? upp_linear_sb_f32(A, K, planes, nullptr, C, N, nullptr, 0, M, N, K, 0, tile, s2)
                            : upp_linear_f32(A, K, W, K, nullptr, C, N, nullptr, 0, M, N, K, 0, 0, s2);
            if (rc) { printf("launch rc %d\n", rc); return 1; }
        }
        hipDeviceSynchronize();
    }
    unsigned hb = 0; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    unsigned hf[16]; hipMemcpy(hf, bad, 64, hipMemcpyDeviceToHost);
    printf("   plain fma/mul/add %u | add VGPR op_sel_hi:[1,0] %u | add VGPR op_sel:[0,1] %u | add SGPR op_sel_hi %u | add SGPR op_sel %u | add op_sel:[1,0] %u | mul op_sel:[0,1] %u | fma op_sel:[0,1,0] %u\n",
           hf[0], hf[1], hf[2], hf[3], hf[4], hf[7], hf[8], hf[9]);
    printf("%s N=%d K=%d tile=%x %s: %u mismatches\n", use_sb == 1 ? "sb " : use_sb == 0 ? "f32" : use_sb == 3 ? "wgrad_sb" : "none", N, K, tile, lds_bytes > 0 ? "LDS canary" : "packed-f32 canary", hb);
    return 0;
}
