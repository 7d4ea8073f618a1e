// Probe for DESIGN 7 item 0: can the ~1.8 us kernel boundary + ~1.2 us ramp between two DEPENDENT bandwidth-bound launches be hidden?
//
// Every "layer" streams its own weight slab (a stand-in for a decode GEMV: 256 CUs x 2 workgroups, 16-byte loads, 4 loads in flight per lane) and needs the
// previous layer's 16 KB output vector.  K layers = one "token".  Modes (all must produce the same bits):
//   A  one stream, K dependent launches captured in a graph (today's decode pass).
//   B  launches alternate between two streams inside ONE captured graph (fork / join with events): launch i + 1 only has a graph edge to launch i - 1; it requests
//      its first weight bytes, THEN waits (bounded) until launch i's arrival counter is full (every workgroup of i: stores, barrier, lane-0 agent release fence,
//      one atomic add), lane-0 agent acquire fence, then reads the vector.  At most two launches coexist and both fit on the chip together, so the waiting
//      workgroups cannot starve the producer.
//   E  the same two-stream launch sequence issued eagerly (no graph): tells whether a HIP graph keeps the two branches on separate hardware queues.
//   P  ONE persistent launch per token (512 co-resident workgroups loop over the K layers, same counters as B): no dispatch at all between layers.
// Counters are zeroed by a memset at the head of every token (graph node in A / B), so kernel arguments never change between replays.
//
// build: hipcc --offload-arch=gfx950 -O2 scripts/probes/chain_probe.hip -o gpurun_out/chain_probe ; run on the GPU box under `timeout 120`:
//   chain_probe [K=129] [reps=30] [slab MB list, default "9 33 66"]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

constexpr int VEC = 4096;            // floats handed from layer to layer
constexpr int GRID = 512;

struct Args { const uint4 *w; long n16; long slab16; int nslabs; const float *xin; float *xout; unsigned *counters; unsigned *err; int idx; int chained; int layers; };

__device__ __forceinline__ int wait_counter(const unsigned *c, unsigned target, unsigned *err) {
    // one lane polls with relaxed agent-scope loads (L2 / fabric, never the L1) and sleeps between polls; bounded (~1 s), and once any wait has timed out
    // every later one gives up after 1024 polls
    long spins = 0;
    while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(4);
        if (++spins > 2000000L || ((spins & 1023) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return 0;
        }
    }
    return 1;
}

__device__ __forceinline__ void layer_body(const Args &a, const int idx, const bool chained, float *xs, float *ws) {
    const uint4 *w = a.w + a.slab16 * (idx % a.nslabs);
    const float *xin = a.xin + (long)VEC * idx; float *xout = a.xout + (long)VEC * idx;
    const long per = (a.n16 + GRID - 1) / GRID, i0 = (long)blockIdx.x * per, i1 = min(a.n16, i0 + per);
    // (a) the first weight bytes do not depend on the previous layer: request them before anything else
    uint4 pre[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) { const long i = min(i0 + threadIdx.x + 256L * p, a.n16 - 1); pre[p] = w[i]; }
    // (b) wait for the producer
    if (chained && idx > 0) {
        if (threadIdx.x == 0) { wait_counter(a.counters + idx - 1, GRID, a.err); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
        __syncthreads();
    }
    // (c) the dependent data
    for (int i = threadIdx.x; i < VEC; i += 256) xs[i] = xin[i];
    __syncthreads();
    // (d) stream the slab
    float acc = 0.f;
#pragma unroll
    for (int p = 0; p < 4; ++p) { const long i = i0 + threadIdx.x + 256L * p; if (i < i1) acc += __uint_as_float((pre[p].x ^ pre[p].y ^ pre[p].z ^ pre[p].w) & 0x007fffffu | 0x3f800000u) * xs[(i + p) & (VEC - 1)]; }
    long i = i0 + threadIdx.x + 1024;
    for (; i + 768 < i1; i += 1024) {        // 4 loads in flight per lane
        uint4 v[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) v[p] = w[i + 256 * p];
#pragma unroll
        for (int p = 0; p < 4; ++p) acc += __uint_as_float((v[p].x ^ v[p].y ^ v[p].z ^ v[p].w) & 0x007fffffu | 0x3f800000u) * xs[(i + 256 * p) & (VEC - 1)];
    }
    for (; i < i1; i += 256) { const uint4 v = w[i]; acc += __uint_as_float((v.x ^ v.y ^ v.z ^ v.w) & 0x007fffffu | 0x3f800000u) * xs[i & (VEC - 1)]; }
    // (e) every workgroup owns VEC / GRID outputs: wave sums in a deterministic order
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
    __syncthreads();
    constexpr int outs = VEC / GRID;
    if ((int)threadIdx.x < outs) xout[VEC + blockIdx.x * outs + threadIdx.x] = 1e-3f * (ws[0] + ws[1] + ws[2] + ws[3]) / (float)(per + 1) + 0.5f * xs[blockIdx.x * outs + threadIdx.x] + 0.25f;
    // (f) arrive: stores -> barrier -> lane-0 agent release (L2 write-back) -> drained -> one atomic
    if (chained) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(a.counters + idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__global__ void __launch_bounds__(256) layer_kernel(Args a) {
    __shared__ float xs[VEC]; __shared__ float ws[4];
    layer_body(a, a.idx, a.chained != 0, xs, ws);
}

__global__ void __launch_bounds__(256) persistent_kernel(Args a) {
    __shared__ float xs[VEC]; __shared__ float ws[4];
    for (int l = 0; l < a.layers; ++l) { layer_body(a, l, true, xs, ws); __syncthreads(); }
}

int main(int argc, char **argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 129;                // layers per token ("mat-muls per token")
    const int reps = argc > 2 ? atoi(argv[2]) : 30;
    std::vector<long> slabs; for (int i = 3; i < argc; ++i) slabs.push_back(atol(argv[i]));
    if (slabs.empty()) slabs = {9, 33, 66};
    int rc = 0;
    for (long mb : slabs) {
        const long slab = mb << 20; const int nslabs = (int)((600L << 20) / slab) + 1;    // > 256 MB in rotation: HBM, not the Infinity Cache
        uint4 *w; float *x[4]; unsigned *counters, *err;
        CK(hipMalloc(&w, slab * nslabs)); CK(hipMemset(w, 0x5a, slab * nslabs));
        for (int m = 0; m < 4; ++m) CK(hipMalloc(&x[m], VEC * 4 * (K + 1)));
        CK(hipMalloc(&counters, 4 * K)); CK(hipMalloc(&err, 4)); CK(hipMemset(counters, 0, 4 * K)); CK(hipMemset(err, 0, 4));
        std::vector<float> h0(VEC); for (int i = 0; i < VEC; ++i) h0[i] = 0.001f * (i % 97);
        hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
        hipEvent_t fork, join, t0, t1; CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming)); CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
        double us[4] = {0, 0, 0, 0}; std::vector<float> out[4];
        const char *names[4] = {"A graph edges", "B two-stream flags (graph)", "E two-stream flags (eager)", "P persistent"};
        for (int mode = 0; mode < 4; ++mode) {
            float *xa = x[mode]; CK(hipMemcpy(xa, h0.data(), VEC * 4, hipMemcpyHostToDevice));
            Args base{w, slab / 16, slab / 16, nslabs, xa, xa, counters, err, 0, mode != 0, K};
            auto token = [&](bool two) -> int {      // the launch sequence of one token on s0 (/ s1)
                CK(hipMemsetAsync(counters, 0, 4 * K, s0));
                if (two) { CK(hipEventRecord(fork, s0)); CK(hipStreamWaitEvent(s1, fork, 0)); }
                for (int i = 0; i < K; ++i) { Args a = base; a.idx = i; hipLaunchKernelGGL(layer_kernel, dim3(GRID), dim3(256), 0, (two && (i & 1)) ? s1 : s0, a); }
                if (two) { CK(hipEventRecord(join, s1)); CK(hipStreamWaitEvent(s0, join, 0)); }
                return 0;
            };
            hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
            if (mode < 2) {
                CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
                if (token(mode == 1)) return 2;
                CK(hipStreamEndCapture(s0, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            }
            auto run = [&]() -> int {
                if (mode < 2) { CK(hipGraphLaunch(ge, s0)); }
                else if (mode == 2) { if (token(true)) return 2; }
                else { CK(hipMemsetAsync(counters, 0, 4 * K, s0)); hipLaunchKernelGGL(persistent_kernel, dim3(GRID), dim3(256), 0, s0, base); }
                return 0;
            };
            for (int r = 0; r < 3; ++r) if (run()) return 2;
            CK(hipStreamSynchronize(s0));
            CK(hipEventRecord(t0, s0));
            for (int r = 0; r < reps; ++r) if (run()) return 2;
            CK(hipEventRecord(t1, s0)); CK(hipStreamSynchronize(s0));
            float ms = 0; CK(hipEventElapsedTime(&ms, t0, t1)); us[mode] = ms * 1e3 / reps / K;
            out[mode].resize(VEC); CK(hipMemcpy(out[mode].data(), xa + (long)VEC * K, VEC * 4, hipMemcpyDeviceToHost));
            if (ge) { CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); }
            unsigned e = 0; CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
            bool same = true; for (int i = 0; i < VEC; ++i) same = same && out[mode][i] == out[0][i];
            printf("%5.1f MB x %d layers  %-28s %6.2f us per layer  %5.2f TB/s  outputs %s  wait timeouts %s\n", slab / 1048576.0, K, names[mode], us[mode], slab / us[mode] * 1e-6,
                   same ? "identical" : "DIFFER", e ? "YES" : "none");
            fflush(stdout);
            if (!same || e) rc = 1;
            CK(hipMemset(err, 0, 4));
        }
        CK(hipFree(w)); for (int m = 0; m < 4; ++m) CK(hipFree(x[m])); CK(hipFree(counters)); CK(hipFree(err));
        CK(hipStreamDestroy(s0)); CK(hipStreamDestroy(s1));
    }
    return rc;
}
