// Probe (round 5): can the HBM-idle window around a dependent kernel boundary (tail of launch i, the ~1.8 us boundary, ramp of launch i + 1) be used to pull the
// HEAD of launch i + 1's weight slab into the L2 of the XCD that will consume it?
//   R(i)   the decode GEMV's load shape: 512 workgroups x 256 threads, VGPR ring of 8 x 16 B per lane, workgroup b streams chunk b of slab i, checksum consumer
//   P(i)   prefetcher: 256 workgroups x 64 threads; workgroup p touches the first `pf` bytes of chunks p and p + 256 of slab i (same p % 8 => same XCD as the consumers
//          under the observed round-robin placement; both kernels record HW_REG_XCC_ID so the run says whether that held); the loads go nowhere (asm, waited at the end)
// Modes (K launches in one graph, each on its own slab of a pool larger than the 256 MB infinity cache):
//   base     R chain only
//   serial   P(i) R(i) on ONE stream: does a line fetched by one kernel survive the boundary in the consumer's L2?  (cost of P itself: mode `ponly`)
//   side     R chain on the main branch; P(i + 1) on a second branch, released by R(i - 1), sleeping `delay` x 10 ns (s_memrealtime) before it starts to load
//   tail     no second kernel: every workgroup of R(i), when its own chunk is done, touches the head of ITS chunk of slab i + 1
//   build: hipcc --offload-arch=gfx950 -O2 scripts/probes/l2_prefetch_probe.hip -o gpurun_out/l2_prefetch_probe      run: timeout 200 gpurun_out/l2_prefetch_probe [K=64] [reps=10] ["9 15 33 66"]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

struct Args { const uint4 *w; const uint4 *wnext; long n16; unsigned *out; unsigned *xcc; long pf16; long delay; };

__device__ __forceinline__ unsigned fold(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }
__device__ __forceinline__ unsigned wave_xor(unsigned v) { for (int o = 32; o > 0; o >>= 1) v ^= __shfl_xor(v, o, 64); return v; }
__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 15u; }
// touch `cnt` 16-byte units starting at base + first, `step` apart (clamped to the slab): ordinary loads, 8 in flight per lane, folded into a value that a never-true test keeps alive
// (an asm load with a dead "=v" destination lets the compiler reuse the register while the load is in flight: memory fault)
__device__ __forceinline__ unsigned touch_range(const uint4 *w, long n16, long first, long cnt, long step) {
    unsigned acc = 0;
    for (long j = first; j < cnt; j += 8 * step) {
        uint4 v[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) v[p] = w[min(j + p * step, n16 - 1)];
#pragma unroll
        for (int p = 0; p < 8; ++p) acc ^= fold(v[p]);
    }
    return acc;
}

template <bool TAIL>
__global__ void __launch_bounds__(256) stream_r(Args a) {
    constexpr int R = 8;
    const long per = (a.n16 + gridDim.x - 1) / gridDim.x, i0 = (long)blockIdx.x * per, i1 = min(a.n16, i0 + per);
    uint4 ring[R]; unsigned acc = 0;
    long i = i0 + threadIdx.x;
    auto ld = [&](long j) -> uint4 { return a.w[min(j, a.n16 - 1)]; };
#pragma unroll
    for (int p = 0; p < R; ++p) ring[p] = ld(i + 256L * p);
    for (; i < i1; i += 256L * R) {
#pragma unroll
        for (int p = 0; p < R; ++p) {
            const uint4 v = ring[p]; ring[p] = ld(i + 256L * R + 256L * p);
            if (i + 256L * p < i1) acc ^= fold(v);
        }
    }
    acc = wave_xor(acc);
    if ((threadIdx.x & 63) == 0) a.out[4 * blockIdx.x + (threadIdx.x >> 6)] = acc;
    if (a.xcc && threadIdx.x == 0) a.xcc[blockIdx.x] = xcc_id();
    if (TAIL && a.wnext) {
        const unsigned t = touch_range(a.wnext + i0, a.n16 - i0, threadIdx.x, a.pf16, 256);
        if (t == 0x9e3779b9u && a.pf16 < 0) a.out[0] = t;
    }
}

// grid 256 x 64 threads; rgrid = the consumer's workgroup count (a multiple of 256)
__global__ void __launch_bounds__(256) prefetch_p(Args a, int rgrid) {
    if (a.delay > 0) {
        const unsigned long long t0 = wall_clock64();
        while ((long)(wall_clock64() - t0) < a.delay) __builtin_amdgcn_s_sleep(8);
    }
    const long per = (a.n16 + rgrid - 1) / rgrid; unsigned acc = 0;
    for (int b = blockIdx.x; b < rgrid; b += gridDim.x) {
        const long i0 = (long)b * per;
        acc ^= touch_range(a.w + i0, a.n16 - i0, threadIdx.x, a.pf16, blockDim.x);
    }
    if (acc == 0x9e3779b9u && a.pf16 < 0) a.xcc[4095] = acc;
    if (a.xcc && threadIdx.x == 0) a.xcc[blockIdx.x] = xcc_id();
}

int main(int argc, char **argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 64, reps = argc > 2 ? atoi(argv[2]) : 10;
    const char *sizes = argc > 3 ? argv[3] : "9 15 33 66";
    hipStream_t st, side; CK(hipStreamCreate(&st)); CK(hipStreamCreate(&side));
    const size_t pool = (size_t)1280 << 20;
    uint4 *w; CK(hipMalloc(&w, pool));
    { std::vector<unsigned> h(pool / 4); unsigned x = 12345; for (auto &v : h) { x = x * 1664525u + 1013904223u; v = x; } CK(hipMemcpy(w, h.data(), pool, hipMemcpyHostToDevice)); }
    unsigned *out; CK(hipMalloc(&out, sizeof(unsigned) * K * 4096));
    unsigned *xr, *xp; CK(hipMalloc(&xr, 4096 * 4)); CK(hipMalloc(&xp, 4096 * 4));
    const int RG = 512; const int PT = argc > 4 ? atoi(argv[4]) : 128;      // prefetcher threads per workgroup
    char buf[256]; strncpy(buf, sizes, 255); buf[255] = 0;
    std::vector<hipEvent_t> ev(K + 2); for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (char *tok = strtok(buf, " "); tok; tok = strtok(nullptr, " ")) {
        const long mb = atol(tok), n16 = (mb << 20) / 16; const int nslabs = (int)(pool / ((size_t)mb << 20));
        auto slab = [&](int i) { return w + (size_t)((i % nslabs + nslabs) % nslabs) * n16; };
        std::vector<unsigned> ref;
        struct Case { const char *mode; long pf_mb; double frac; };      // pf_mb = MB of the slab's head that are prefetched (over all chunks); frac = delay as a fraction of the base launch time
        std::vector<Case> cases = {{"base", 0, 0}, {"ponly", 8, 0}, {"serial", 8, 0}, {"ponly", 16, 0}, {"serial", 16, 0}, {"serial", 32, 0},
                                   {"side", 0, 0}, {"side", 8, 0}, {"side", 8, 0.5}, {"side", 8, 0.8}, {"side", 8, 1.0}, {"side", 16, 0}, {"side", 16, 0.5}, {"side", 16, 0.8}, {"side", 16, 1.0},
                                   {"side", 24, 0.7}, {"side", 24, 1.0}, {"side", 4, 0.8}, {"side", 4, 1.0},
                                   {"tail", 4, 0}, {"tail", 8, 0}, {"tail", 16, 0}};
        double base_us = 0;
        for (const Case &c : cases) {
            const long pf16 = std::min<long>(n16 / RG, ((c.pf_mb << 20) / 16) / RG);
            const long delay = (long)(c.frac * base_us * 100.0);      // 100 MHz ticks
            const bool is_side = !strcmp(c.mode, "side"), is_serial = !strcmp(c.mode, "serial"), is_ponly = !strcmp(c.mode, "ponly"), is_tail = !strcmp(c.mode, "tail");
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            CK(hipMemsetAsync(out, 0, sizeof(unsigned) * K * 4096, st));
            CK(hipEventRecord(ev[0], st));
            for (int i = 0; i < K; ++i) {
                Args a{slab(i), is_tail && i + 1 < K ? slab(i + 1) : nullptr, n16, out + 4096 * i, i == K - 1 ? xr : nullptr, pf16, 0};
                Args p{slab(i), nullptr, n16, nullptr, i == K - 1 ? xp : nullptr, pf16, delay};
                if (is_serial || is_ponly) hipLaunchKernelGGL(prefetch_p, dim3(256), dim3(PT), 0, st, p, RG);
                if (is_side && i >= 1) {      // P(i) is released by R(i - 2) (ev[i - 1] is recorded behind R(i - 2)), i.e. it runs beside R(i - 1)
                    CK(hipStreamWaitEvent(side, ev[i - 1], 0));
                    hipLaunchKernelGGL(prefetch_p, dim3(256), dim3(PT), 0, side, p, RG);
                }
                if (!is_ponly) { if (is_tail) hipLaunchKernelGGL(stream_r<true>, dim3(RG), dim3(256), 0, st, a); else hipLaunchKernelGGL(stream_r<false>, dim3(RG), dim3(256), 0, st, a); }
                CK(hipEventRecord(ev[i + 1], st));
            }
            if (is_side) { CK(hipEventRecord(ev[K + 1], side)); CK(hipStreamWaitEvent(st, ev[K + 1], 0)); }
            CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
            bool same = true;
            if (!is_ponly) {
                std::vector<unsigned> raw((size_t)K * 4096), got(K, 0u); CK(hipMemcpy(raw.data(), out, sizeof(unsigned) * K * 4096, hipMemcpyDeviceToHost));
                for (int i = 0; i < K; ++i) for (int j = 0; j < 4096; ++j) got[i] ^= raw[(size_t)i * 4096 + j];
                if (ref.empty()) ref = got;
                same = got == ref;
            }
            float best = 1e30f; double sum = 0;
            for (int r = 0; r < reps; ++r) { CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; sum += ms; }
            const double us = best * 1e3 / K;
            if (!strcmp(c.mode, "base")) base_us = us;
            int xcc_match = -1;
            if (is_serial || is_side) {
                std::vector<unsigned> hr(RG), hp(256); CK(hipMemcpy(hr.data(), xr, RG * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hp.data(), xp, 256 * 4, hipMemcpyDeviceToHost));
                xcc_match = 0; for (int b = 0; b < RG; ++b) xcc_match += hr[b] == hp[b % 256];
            }
            printf("%3ld MB  %-7s pf %2ld MB  delay %4.2f  %7.2f us / launch (mean %7.2f)  %s  xcc match %d / %d\n", mb, c.mode, c.pf_mb, c.frac, us, sum / reps * 1e3 / K, same ? "ok" : "CHECKSUM MISMATCH", xcc_match, RG);
            fflush(stdout);
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
    }
    return 0;
}
