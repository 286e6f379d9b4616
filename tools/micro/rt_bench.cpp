// rt_bench.cpp -- stand-alone timing + bit-exactness check of the register-tiled Linear kernels (csrc/linear_rt.hip) without
// Python:   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -I iccv2025-upp_amd/upp_hip/csrc tools/micro/rt_bench.cpp -o gpurun_out/rt_bench
// Every (shape, config) pair: 3 warm-up launches, `iters` timed ones between two HIP events, then 256 sampled outputs against the
// host evaluation of the kernel's own fmaf order (oracle_linear_f32 with ks = 1).
#define UPP_RT_NO_FANCY
#define UPP_RT_CONFIGS(X) X(2, 2, 2, 2, 1, 2) X(2, 2, 2, 2, 2, 2) X(4, 2, 2, 2, 1, 2) X(4, 2, 2, 2, 1, 3) X(2, 4, 2, 2, 1, 2) X(2, 4, 2, 2, 1, 3) X(2, 2, 2, 1, 1, 2) X(2, 2, 2, 1, 1, 3) X(2, 2, 1, 2, 1, 2) X(2, 2, 4, 2, 1, 2) X(2, 2, 2, 4, 1, 2) X(2, 1, 2, 2, 1, 2) X(2, 1, 2, 2, 1, 3) X(1, 2, 2, 2, 1, 3) X(2, 2, 1, 1, 1, 2) X(2, 2, 1, 1, 1, 4) X(4, 2, 1, 1, 1, 3) X(4, 4, 1, 1, 2, 2)
#include "linear_rt.hip"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static inline float frand(uint64_t &s) {
    s = s * 6364136223846793005ULL + 1442695040888963407ULL;
    return ((int)((s >> 33) & 0xFFFFFF) - 0x800000) / (float)0x800000;
}

static float ref_elem(const float *a, const float *w, int K) {
    float acc = 0.0f;
    const int K32 = (K + 31) / 32 * 32;
    for (int k0 = 0; k0 < K32; k0 += 32)
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                const int ka = k0 + 8 * i + j, kb = ka + 4;
                acc = fmaf(ka < K ? a[ka] : 0.0f, ka < K ? w[ka] : 0.0f, acc);
                acc = fmaf(kb < K ? a[kb] : 0.0f, kb < K ? w[kb] : 0.0f, acc);
            }
    return acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static int wgrad_main(int iters) {
    // grouped weight gradients: the pre-training mix (per block: qkv, proj, fc1, fc2 at 2,080 and 864 rows), then the tall seg layers
    struct P { int M, N, K; };
    std::vector<std::vector<P>> groups;
    {
        std::vector<P> pre;
        for (int b = 0; b < 4; ++b) { pre.push_back({2080, 1152, 384}); pre.push_back({2080, 384, 384}); pre.push_back({2080, 1536, 384}); pre.push_back({2080, 384, 1536}); }
        for (int b = 0; b < 12; ++b) { pre.push_back({864, 1152, 384}); pre.push_back({864, 384, 384}); pre.push_back({864, 1536, 384}); pre.push_back({864, 384, 1536}); }
        groups.push_back(pre);
        groups.push_back({{2080, 1152, 384}});
        groups.push_back({{864, 1536, 384}});
        groups.push_back({{65536, 1024, 1536}});
        groups.push_back({{65536, 512, 1024}});
        groups.push_back({{65536, 256, 512}});
        groups.push_back({{65536, 1024, 1536}, {65536, 512, 1024}, {65536, 256, 512}, {4096, 1536, 1152}});
        groups.push_back({{1000, 200, 100}, {77, 52, 36}});
    }
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (auto &grp : groups) {
        const int n = (int)grp.size();
        std::vector<int> M(n), N(n), K(n), rows(n);
        for (int i = 0; i < n; ++i) { M[i] = grp[i].M; N[i] = grp[i].N; K[i] = grp[i].K; }
        int rc = upp_linear_wgrad_grouped_rows(n, M.data(), N.data(), K.data(), rows.data());
        if (rc) { printf("plan rc %d\n", rc); return 1; }
        // all problems of a group share one G / X pair per distinct shape (the kernels do not care); host copies kept for the check
        std::vector<std::vector<float>> hg(n), hx(n);
        std::vector<float *> dg(n), dx(n), dp(n);
        std::vector<const float *> cg(n), cx(n);
        std::vector<long long> ldg(n), ldx(n);
        double flops = 0;
        uint64_t seed = 99;
        for (int i = 0; i < n; ++i) {
            int same = -1;
            for (int j = 0; j < i; ++j) if (M[j] == M[i] && N[j] == N[i] && K[j] == K[i]) { same = j; break; }
            const size_t ng = (size_t)M[i] * N[i], nx = (size_t)M[i] * K[i];
            if (same >= 0) { dg[i] = dg[same]; dx[i] = dx[same]; }
            else {
                hg[i].resize(ng); hx[i].resize(nx);
                for (auto &v : hg[i]) v = frand(seed);
                for (auto &v : hx[i]) v = frand(seed);
                CK(hipMalloc(&dg[i], ng * 4)); CK(hipMalloc(&dx[i], nx * 4));
                CK(hipMemcpy(dg[i], hg[i].data(), ng * 4, hipMemcpyHostToDevice));
                CK(hipMemcpy(dx[i], hx[i].data(), nx * 4, hipMemcpyHostToDevice));
            }
            const int splits = (M[i] + rows[i] - 1) / rows[i];
            CK(hipMalloc(&dp[i], (size_t)splits * N[i] * K[i] * 4));
            CK(hipMemset(dp[i], 0xFF, (size_t)splits * N[i] * K[i] * 4));
            cg[i] = dg[i]; cx[i] = dx[i]; ldg[i] = N[i]; ldx[i] = K[i];
            flops += 2.0 * M[i] * N[i] * K[i];
        }
        for (int w = 0; w < 3; ++w) rc = upp_linear_wgrad_grouped_f32(cg.data(), ldg.data(), cx.data(), ldx.data(), dp.data(), M.data(), N.data(), K.data(), rows.data(), n, st);
        if (rc) { printf("launch rc %d\n", rc); return 1; }
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int w = 0; w < iters; ++w) upp_linear_wgrad_grouped_f32(cg.data(), ldg.data(), cx.data(), ldx.data(), dp.data(), M.data(), N.data(), K.data(), rows.data(), n, st);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        int bad = 0, checked = 0;
        for (int i = 0; i < n; ++i) {
            if (hg[i].empty()) continue;
            const int splits = (M[i] + rows[i] - 1) / rows[i];
            std::vector<float> hp((size_t)splits * N[i] * K[i]);
            CK(hipMemcpy(hp.data(), dp[i], hp.size() * 4, hipMemcpyDeviceToHost));
            uint64_t ps = 4242 + i;
            for (int t = 0; t < 96; ++t) {
                ps = ps * 6364136223846793005ULL + 1442695040888963407ULL;
                int sp = (int)((ps >> 40) % splits), nn = (int)((ps >> 20) % N[i]), kk = (int)((ps >> 3) % K[i]);
                if (t < 4) { sp = (t & 1) ? splits - 1 : 0; nn = (t & 2) ? N[i] - 1 : 0; kk = (t & 2) ? K[i] - 1 : 0; }
                float acc = 0.0f;
                const int m0 = sp * rows[i], m1 = std::min(M[i], m0 + rows[i]);
                for (int m = m0; m < m1; ++m) acc = fmaf(hg[i][(size_t)m * N[i] + nn], hx[i][(size_t)m * K[i] + kk], acc);
                if (memcmp(&acc, &hp[((size_t)sp * N[i] + nn) * K[i] + kk], 4)) ++bad;
                ++checked;
            }
        }
        const double us = ms * 1e3 / iters;
        printf("wgrad group of %2d (first %d x %d x %d, rows/split %d): %9.1f us  %6.1f TF  %s (%d checked)\n", n, M[0], N[0], K[0], rows[0], us, flops / (us * 1e-6) / 1e12,
               bad ? "MISMATCH" : "exact", checked);
        fflush(stdout);
        for (int i = 0; i < n; ++i) { if (!hg[i].empty()) { CK(hipFree(dg[i])); CK(hipFree(dx[i])); } CK(hipFree(dp[i])); }
    }
    return 0;
}

int main(int argc, char **argv) {
    if (argc > 2 && !strcmp(argv[2], "wgrad")) return wgrad_main(atoi(argv[1]));
    struct Shape { const char *name; int M, N, K; };
    std::vector<Shape> shapes = {{"tall_1536x1024", 65536, 1024, 1536}, {"tall_dx_1024x1536", 65536, 1536, 1024}, {"tall_1024x512", 65536, 512, 1024},
                                 {"tall_512x256", 65536, 256, 512},    {"fc1_2400", 2400, 1536, 384},         {"qkv_2400", 2400, 1152, 384},
                                 {"fc2_2400", 2400, 384, 1536},        {"proj_2400", 2400, 384, 384},         {"edge_1000x200x100", 1000, 200, 100},
                                 {"longk_2wg", 16384, 512, 16384},     {"longk_1wg", 16384, 256, 16384}};
    struct Cfg { int wm, wn, rm, rn, kc, nst; };
    std::vector<Cfg> cfgs;
#define ADD(a, b, c, d, e, f) cfgs.push_back({a, b, c, d, e, f});
    UPP_RT_CONFIGS(ADD)
    const int iters = argc > 1 ? atoi(argv[1]) : 10;
    const char *only = argc > 2 ? argv[2] : nullptr;

    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
#ifdef UPP_LIN_STAMPS
    unsigned long long *dstamps;
    CK(hipMalloc(&dstamps, (size_t)1 << 24));
    CK(hipMemset(dstamps, 0, (size_t)1 << 24));
#endif
    for (const Shape &s : shapes) {
        if (only && !strstr(s.name, only)) continue;
        const size_t na = (size_t)s.M * s.K, nw = (size_t)s.N * s.K, nc = (size_t)s.M * s.N;
        std::vector<float> ha(na), hw(nw), hc(nc);
        uint64_t seed = 12345;
        for (auto &v : ha) v = frand(seed);
        for (auto &v : hw) v = frand(seed) * 0.05f;
        float *da, *dw, *dc;
        CK(hipMalloc(&da, na * 4)); CK(hipMalloc(&dw, nw * 4)); CK(hipMalloc(&dc, nc * 4));
        CK(hipMemcpy(da, ha.data(), na * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice));
        for (const Cfg &c : cfgs) {
            LinArgs g{};
#ifdef UPP_LIN_STAMPS
            g.stamps = dstamps;
#endif
            g.A = da; g.lda = s.K; g.W = dw; g.ldw = s.K; g.C = dc; g.ldc = s.N; g.M = s.M; g.N = s.N; g.K = s.K; g.epi = LEPI_NONE;
            const int code = 0x1000000 * c.nst + 0x100000 * (4 * (c.rm - 1) + (c.rn - 1)) + 0x10000 + c.wm * 4096 + c.wn * 256 + 16 + c.kc;
            CK(hipMemsetAsync(dc, 0xFF, nc * 4, st));
            int rc = 0;
            for (int i = 0; i < 3 && !rc; ++i) rc = upp_detail_linear_rt(&g, code, st);
            if (rc) { printf("%-20s cfg %d%d%d%d/%d/%d: rc %d\n", s.name, c.wm, c.wn, c.rm, c.rn, c.kc, c.nst, rc); continue; }
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; ++i) upp_detail_linear_rt(&g, code, st);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(hc.data(), dc, nc * 4, hipMemcpyDeviceToHost));
            int bad = 0;
            uint64_t ps = 777;
            for (int t = 0; t < 256; ++t) {
                ps = ps * 6364136223846793005ULL + 1442695040888963407ULL;
                int m = (int)((ps >> 33) % s.M), n = (int)((ps >> 13) % s.N);
                if (t < 8) { m = (t & 1) ? s.M - 1 : 0; n = (t & 2) ? s.N - 1 : 0; if (t & 4) { m = s.M / 2 + 1; n = s.N / 2 + 1; } }
                const float want = ref_elem(&ha[(size_t)m * s.K], &hw[(size_t)n * s.K], s.K);
                if (memcmp(&want, &hc[(size_t)m * s.N + n], 4)) ++bad;
            }
            const double us = ms * 1e3 / iters, tf = 2.0 * s.M * s.N * s.K / (us * 1e-6) / 1e12;
            printf("%-20s cfg wm%d wn%d rm%d rn%d kc%d st%d : %9.1f us  %6.1f TF  %s", s.name, c.wm, c.wn, c.rm, c.rn, c.kc, c.nst, us, tf, bad ? "MISMATCH" : "exact");
#ifdef UPP_LIN_STAMPS
            {   // in-kernel clock and phase split of the LAST launch: medians over the workgroups
                const int BMt = c.wm * c.rm * 32, BNt = c.wn * c.rn * 32;
                const size_t nwg = (size_t)((s.M + BMt - 1) / BMt) * ((s.N + BNt - 1) / BNt);
                std::vector<unsigned long long> hs(nwg * 8);
                CK(hipMemcpy(hs.data(), dstamps, nwg * 64, hipMemcpyDeviceToHost));
                std::vector<double> clk, pro, loop, epi;
                for (size_t w = 0; w < nwg; ++w) {
                    const unsigned long long *q = &hs[w * 8];
                    if (q[5] <= q[4]) continue;
                    clk.push_back((double)(q[3] - q[0]) / (double)(q[5] - q[4]) * 0.1);      // GHz (memrealtime ticks at 100 MHz)
                    pro.push_back((double)(q[1] - q[0])); loop.push_back((double)(q[2] - q[1])); epi.push_back((double)(q[3] - q[2]));
                }
                auto med = [](std::vector<double> &v) { if (v.empty()) return 0.0; std::nth_element(v.begin(), v.begin() + v.size() / 2, v.end()); return v[v.size() / 2]; };
                const double mf = (double)c.rm * c.rn * ((s.K + 31) / 32 * 16) * 64.0;         // MFMA issue cycles of one wave
                printf("  | clk %.2f GHz  pro %.0f  loop %.0f (wave MFMA share %.2f)  epi %.0f cyc", med(clk), med(pro), med(loop), mf / med(loop), med(epi));
            }
#endif
            printf("\n");
            fflush(stdout);
        }
        CK(hipFree(da)); CK(hipFree(dw)); CK(hipFree(dc));
    }
    return 0;
}
