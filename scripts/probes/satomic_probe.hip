// Probe: does gfx950 execute scalar memory atomics (s_atomic_add ... glc)?  One ticket per wave, no VMEM op, no exec masking.
// build: hipcc --offload-arch=gfx950 -O2 scripts/probes/satomic_probe.hip -o gpurun_out/satomic_probe ; run on the GPU box under `timeout 20`.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void k(unsigned *ctr, unsigned *out, int per_wave) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    for (int i = 0; i < per_wave; ++i) {
        unsigned v = 1;
        asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(ctr) : "memory");
        if ((threadIdx.x & 63) == 0) out[wave * per_wave + i] = v;
    }
}
int main() {
    const int wgs = 1024, threads = 256, per_wave = 8, waves = wgs * threads / 64, n = waves * per_wave;
    unsigned *ctr, *out; hipMalloc(&ctr, 4); hipMalloc(&out, n * 4); hipMemset(ctr, 0, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(wgs), dim3(threads), 0, 0, ctr, out, per_wave); hipEventRecord(e1);
    hipError_t err = hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned> h(n); unsigned c = 0; hipMemcpy(h.data(), out, n * 4, hipMemcpyDeviceToHost); hipMemcpy(&c, ctr, 4, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end()); bool ok = c == (unsigned)n; for (int i = 0; i < n; ++i) ok = ok && h[i] == (unsigned)i;
    printf("sync=%s counter=%u expected=%d unique_tickets=%s time=%.1f us (%.1f ns per ticket)\n", hipGetErrorString(err), c, n, ok ? "yes" : "NO", ms * 1e3, ms * 1e6 / n);
    return ok ? 0 : 1;
}
