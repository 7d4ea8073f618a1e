// Probe (round 5): where do the ~8 us of the per-head decode attention (fa_decode_body_v2, csrc/fa_decode.cuh) go?  The kernel body is compiled here with FA_TL(i) = an
// s_memtime stamp of wave 0 / lane 0 of every workgroup, launched as a decoded token launches it (32 q heads / 8 KV heads of 128, f16 K / V of `n_kv` cells of which `n_vis`
// are visible, mask row, q8 emission) over a pool of K / V sets larger than L2 + the infinity cache, so every launch misses as a layer's first touch of its cache does.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include scripts/probes/fa_timeline_probe.hip -o scripts/probes/bin/fa_timeline_probe
//   run:   scripts/probes/bin/fa_timeline_probe [n_vis=100] [n_kv=256] [NW=4] [hot=0]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
__shared__ long long s_tl[16];        // stamps of this workgroup (LDS: a stamp must not wait for the loads in flight, as a store through a pointer loaded from memory would)
#define FA_TL(i_) do { if (threadIdx.x == 0) s_tl[i_] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#include "../../ik_llama.cpp_amd/csrc/fa_decode.cuh"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

template <int NW>
__global__ void __launch_bounds__(64 * NW) probe_kernel(TD q, TD k, TD v, TD mask, TD dst, float scale, uint8_t *q8, long long *tl) {
    __shared__ float s_m[NW], s_l[NW]; __shared__ float s_acc[NW][128];
    if (threadIdx.x == 0) s_tl[15] = (long long)__builtin_amdgcn_s_memtime();
#ifdef PIN_ARGS      // experiment: every kernel argument the body reads is requested at the entry (one batch of s_loads, one wait) instead of where a branch first needs it
#define PIN_S(x_) asm volatile("" :: "s"(x_))
    PIN_S(q.data); PIN_S(q.ne[2]); PIN_S(q.ne[3]); PIN_S(q.nb[1]); PIN_S(q.nb[2]); PIN_S(q.nb[3]);
    PIN_S(k.data); PIN_S(k.ne[1]); PIN_S(k.ne[2]); PIN_S(k.ne[3]); PIN_S(k.nb[1]); PIN_S(k.nb[2]); PIN_S(k.nb[3]);
    PIN_S(v.data); PIN_S(v.ne[2]); PIN_S(v.ne[3]); PIN_S(v.nb[1]); PIN_S(v.nb[2]); PIN_S(v.nb[3]);
    PIN_S(mask.data); PIN_S(mask.ne[2]); PIN_S(mask.ne[3]); PIN_S(mask.nb[1]); PIN_S(mask.nb[2]); PIN_S(mask.nb[3]);
    PIN_S(dst.data); PIN_S(dst.ne[1]); PIN_S(dst.ne[2]); PIN_S(dst.nb[1]); PIN_S(scale); PIN_S(q8);
#endif
    fa_decode_body_v2<false, NW>(q, k, v, mask, 1, dst, scale, 0.f, 0.f, 1.f, 1.f, 32, blockIdx.x, blockIdx.y, blockIdx.z, s_m, s_l, s_acc, q8);
    if (threadIdx.x == 0) { s_tl[10] = (long long)__builtin_amdgcn_s_memtime(); for (int i = 0; i < 16; ++i) tl[16 * blockIdx.y + i] = s_tl[i]; }
}
__global__ void empty_kernel(float *p) { if (p && threadIdx.x == 1024) p[0] = 1.f; }

static TD td(void *p, long e, long n0, long n1, long n2) { TD t; t.data = (char *)p; t.ne[0] = n0; t.ne[1] = n1; t.ne[2] = n2; t.ne[3] = 1; t.nb[0] = e; t.nb[1] = e * n0; t.nb[2] = t.nb[1] * n1; t.nb[3] = t.nb[2] * n2; return t; }

int main(int argc, char **argv) {
    const int n_vis = argc > 1 ? atoi(argv[1]) : 100, n_kv = argc > 2 ? atoi(argv[2]) : 256, NW = argc > 3 ? atoi(argv[3]) : 4, hot = argc > 4 ? atoi(argv[4]) : 0;
    const int H = 32, HK = 8, D = 128, SETS = hot ? 1 : 600;
    const size_t kv_bytes = (size_t)HK * n_kv * D * 2;
    char *kp, *vp; CK(hipMalloc(&kp, kv_bytes * SETS)); CK(hipMalloc(&vp, kv_bytes * SETS));
    std::vector<unsigned short> hk(kv_bytes / 2); for (size_t i = 0; i < hk.size(); ++i) hk[i] = 0x3000 + (unsigned short)((i * 2654435761u) >> 22);     // small positive halves
    for (int s = 0; s < SETS; ++s) { CK(hipMemcpy(kp + s * kv_bytes, hk.data(), kv_bytes, hipMemcpyHostToDevice)); CK(hipMemcpy(vp + s * kv_bytes, hk.data(), kv_bytes, hipMemcpyHostToDevice)); }
    float *qp, *dp; uint8_t *q8; __half *mp; long long *tl;
    CK(hipMalloc(&qp, H * D * 4)); CK(hipMalloc(&dp, H * D * 4)); CK(hipMalloc(&q8, H * 144)); CK(hipMalloc(&mp, 32 * n_kv * 2)); CK(hipMalloc(&tl, H * 16 * 8));
    std::vector<float> hq(H * D); for (auto &x : hq) x = 0.01f * (float)(rand() % 200 - 100); CK(hipMemcpy(qp, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
    std::vector<unsigned short> hm(32 * n_kv); for (int r = 0; r < 32; ++r) for (int j = 0; j < n_kv; ++j) hm[r * n_kv + j] = j < n_vis ? 0 : 0xfc00; CK(hipMemcpy(mp, hm.data(), hm.size() * 2, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st)); hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const TD q = td(qp, 4, D, 1, H), dst = td(dp, 4, D, H, 1), mask = td(mp, 2, n_kv, 32, 1);
    auto launch = [&](int s) {
        const TD k = td(kp + (size_t)s * kv_bytes, 2, D, n_kv, HK), v = td(vp + (size_t)s * kv_bytes, 2, D, n_kv, HK);
        const dim3 g(1, H, 1);
        if (NW == 4) hipLaunchKernelGGL(probe_kernel<4>, g, dim3(256), 0, st, q, k, v, mask, dst, 0.088f, q8, tl);
        else hipLaunchKernelGGL(probe_kernel<8>, g, dim3(512), 0, st, q, k, v, mask, dst, 0.088f, q8, tl);
    };
    for (int s = 0; s < 8; ++s) launch(s % SETS);
    CK(hipStreamSynchronize(st));
    // per-launch time in a back-to-back chain (what a graph replay sees), then the stamps of single launches
    const int N = 500;
    CK(hipEventRecord(e0, st)); for (int i = 0; i < N; ++i) launch((8 + i) % SETS); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); printf("n_vis %d n_kv %d NW %d %s: %.2f us per launch in a chain of %d\n", n_vis, n_kv, NW, hot ? "hot K/V" : "cold K/V", ms * 1e3 / N, N);
    CK(hipEventRecord(e0, st)); for (int i = 0; i < N; ++i) hipLaunchKernelGGL(empty_kernel, dim3(32), dim3(256), 0, st, (float *)nullptr); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1)); printf("empty kernel chain: %.2f us per launch\n", ms * 1e3 / N);
    // s_memtime counts shader clocks here, not 100 MHz: calibrate against the chain time (body = chain - empty launch)
    const char *names[11] = {"entry->body", "loads issued", "q.k done (K arrived)", "tile max", "P.V done (V arrived)", "loop end", "barrier 1", "barrier 2 (acc merged)", "out stored", "q8 emitted", "end"};
    std::vector<double> acc(16, 0.0); std::vector<long long> h(H * 16); int reps = 40; double span = 0;
    for (int r = 0; r < reps; ++r) {
        launch((100 + 7 * r) % SETS); CK(hipStreamSynchronize(st)); CK(hipMemcpy(h.data(), tl, h.size() * 8, hipMemcpyDeviceToHost));
        long long first = h[15], last = h[10];
        for (int w = 0; w < H; ++w) { first = std::min(first, h[16 * w + 15]); last = std::max(last, h[16 * w + 10]); for (int i = 0; i <= 10; ++i) acc[i] += (double)(h[16 * w + i] - h[16 * w + 15]); }
        span += (double)(last - first);
    }
    const double ticks_per_us = 2400.0;          // (MI355X shader clock; the stamps are for the SHAPE of the timeline)
    printf("stamps (s_memtime at ~2.4 GHz), mean over %d workgroups x %d launches, from the workgroup's entry:\n", H, reps);
    double prev = 0; for (int i = 0; i <= 10; ++i) { const double us = acc[i] / (reps * H) / ticks_per_us; printf("  %-26s %6.2f us  (+%.2f)\n", names[i], us, us - prev); prev = us; }
    (void)span;
    return 0;
}
