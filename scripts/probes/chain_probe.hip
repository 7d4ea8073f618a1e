// Probe for DESIGN 7 item 0: can the ~1.8 us kernel boundary + ~1.2 us ramp between two DEPENDENT bandwidth-bound launches be hidden by launching them as
// independent graph nodes (two alternating streams) that are ordered by a device flag instead of a graph edge?
//
// Every "layer" kernel streams its own weight slab (a stand-in for a decode GEMV: 256 CUs x 2 workgroups, 16-byte loads) and needs the previous kernel's 16 KB
// output vector.  Mode A: one stream, K dependent launches captured in a graph (today's decode pass).  Mode B: launches alternate between two streams inside
// one captured graph (fork / join with events), so launch i + 1 only has a graph edge to launch i - 1; it prefetches its first weight bytes, THEN waits
// (bounded) for launch i's "done" flag (published by i's last workgroup, agent-scope release), then reads the vector.  Both modes must produce the same output.
// At most two launches coexist (i + 2 sits behind i on the same stream), and both fit on the chip together, so the waiting workgroups cannot starve the producer.
//
// build: hipcc --offload-arch=gfx950 -O2 scripts/probes/chain_probe.hip -o gpurun_out/chain_probe ; run on the GPU box under `timeout 60`.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

constexpr int VEC = 4096;            // floats handed from launch to launch

struct Args { const uint4 *w; long n16; const float *xin; float *xout; unsigned *flags; unsigned *counters; unsigned *err; int idx; int chained; unsigned epoch; };

__global__ void __launch_bounds__(256) layer_kernel(Args a) {
    __shared__ float xs[VEC]; __shared__ int s_ok;
    const long per = (a.n16 + gridDim.x - 1) / gridDim.x, i0 = (long)blockIdx.x * per, i1 = min(a.n16, i0 + per);
    // (a) the first weight bytes do not depend on the previous launch: request them before anything else
    uint4 pre[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) { const long i = min(i0 + threadIdx.x + 256L * p, a.n16 - 1); pre[p] = a.w[i]; }
    // (b) wait for the producer (launch idx - 1 of this epoch)
    if (a.chained && a.idx > 0) {
        if (threadIdx.x == 0) {
            int ok = 1; long spins = 0;
            // bounded: ~1 s, and as soon as any wait has timed out every later one gives up after 1024 polls (the probe then reports the failure instead of keeping the GPU busy)
            while ((int)(__hip_atomic_load(a.flags + a.idx - 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - a.epoch) < 0) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > 1000000L || ((spins & 1023) == 0 && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { ok = 0; break; }
            }
            if (!ok) __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_ok = ok;
        }
        __syncthreads();
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    // (c) the dependent data
    for (int i = threadIdx.x; i < VEC; i += 256) xs[i] = __builtin_nontemporal_load(a.xin + i);
    __syncthreads();
    // (d) stream the slab
    float acc = 0.f;
#pragma unroll
    for (int p = 0; p < 4; ++p) { const long i = i0 + threadIdx.x + 256L * p; if (i < i1) acc += __uint_as_float((pre[p].x ^ pre[p].y ^ pre[p].z ^ pre[p].w) & 0x007fffffu | 0x3f800000u) * xs[(i + p) & (VEC - 1)]; }
    for (long i = i0 + threadIdx.x + 1024; i < i1; i += 256) { const uint4 v = a.w[i]; acc += __uint_as_float((v.x ^ v.y ^ v.z ^ v.w) & 0x007fffffu | 0x3f800000u) * xs[i & (VEC - 1)]; }
    // (e) every workgroup owns VEC / gridDim.x outputs (gridDim.x divides VEC): wave sums of a deterministic order
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    __shared__ float ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
    __syncthreads();
    const int outs = VEC / gridDim.x;
    if ((int)threadIdx.x < outs) a.xout[blockIdx.x * outs + threadIdx.x] = 1e-3f * (ws[0] + ws[1] + ws[2] + ws[3]) / (float)(per + 1) + 0.5f * xs[blockIdx.x * outs + threadIdx.x] + 0.25f;
    // (f) the last workgroup out publishes "done"
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(a.counters + a.idx, 1u) == gridDim.x - 1) { a.counters[a.idx] = 0; __hip_atomic_store(a.flags + a.idx, a.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
}

int main(int argc, char **argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 32;                 // launches per graph ("mat-muls per token")
    const long slab = (argc > 2 ? atol(argv[2]) : 33) << 20;      // bytes per launch
    const int reps = argc > 3 ? atoi(argv[3]) : 50;
    const int grid = 512;
    uint4 *w; float *x[2], *xa; unsigned *flags, *counters, *err;
    const int nslabs = 8;                                          // > 256 MB in rotation: HBM, not the Infinity Cache
    CK(hipMalloc(&w, slab * nslabs)); CK(hipMemset(w, 0x5a, slab * nslabs));
    CK(hipMalloc(&x[0], VEC * 4 * (K + 1))); CK(hipMalloc(&x[1], VEC * 4 * (K + 1))); CK(hipMalloc(&flags, 4 * K)); CK(hipMalloc(&counters, 4 * K)); CK(hipMalloc(&err, 4));
    CK(hipMemset(flags, 0, 4 * K)); CK(hipMemset(counters, 0, 4 * K)); CK(hipMemset(err, 0, 4));
    std::vector<float> h0(VEC); for (int i = 0; i < VEC; ++i) h0[i] = 0.001f * (i % 97);
    hipStream_t s0, s1; CK(hipStreamCreate(&s0)); CK(hipStreamCreate(&s1));
    hipEvent_t fork, join, t0, t1; CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming)); CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    double us[2] = {0, 0}; std::vector<float> out[2];
    for (int mode = 0; mode < 2; ++mode) {
        xa = x[mode]; CK(hipMemcpy(xa, h0.data(), VEC * 4, hipMemcpyHostToDevice));
        // the epoch is a kernel argument here, so every replay needs its own graph instance: build `reps + 3` graphs is too slow -- instead each replay is a fresh
        // capture-free launch sequence for the warm-up and ONE graph per epoch value would defeat the purpose.  The probe therefore resets the flags between replays
        // (a 128-byte memset node at the head of the graph) and always uses epoch 1.
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
        CK(hipMemsetAsync(flags, 0, 4 * K, s0));
        if (mode == 1) { CK(hipEventRecord(fork, s0)); CK(hipStreamWaitEvent(s1, fork, 0)); }
        for (int i = 0; i < K; ++i) {
            Args a{(const uint4 *)((const char *)w + slab * (i % nslabs)), slab / 16, xa + (long)VEC * i, xa + (long)VEC * (i + 1), flags, counters, err, i, mode, 1u};
            hipLaunchKernelGGL(layer_kernel, dim3(grid), dim3(256), 0, (mode == 1 && (i & 1)) ? s1 : s0, a);
        }
        if (mode == 1) { CK(hipEventRecord(join, s1)); CK(hipStreamWaitEvent(s0, join, 0)); }
        CK(hipStreamEndCapture(s0, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, s0));
        CK(hipStreamSynchronize(s0));
        CK(hipEventRecord(t0, s0));
        for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s0));
        CK(hipEventRecord(t1, s0)); CK(hipStreamSynchronize(s0));
        float ms = 0; CK(hipEventElapsedTime(&ms, t0, t1)); us[mode] = ms * 1e3 / reps / K;
        out[mode].resize(VEC); CK(hipMemcpy(out[mode].data(), xa + (long)VEC * K, VEC * 4, hipMemcpyDeviceToHost));
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    unsigned e = 0; CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
    bool same = true; for (int i = 0; i < VEC; ++i) same = same && out[0][i] == out[1][i];
    printf("launches/graph %d, %.1f MB per launch: graph edges %.2f us per launch (%.2f TB/s) | flag-chained, two streams %.2f us per launch (%.2f TB/s) | outputs %s | wait timeouts %s\n",
           K, slab / 1048576.0, us[0], slab / us[0] * 1e-6, us[1], slab / us[1] * 1e-6, same ? "identical" : "DIFFER", e ? "YES" : "none");
    return (same && !e) ? 0 : 1;
}
