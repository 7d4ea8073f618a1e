// Probe for the decode target (VERDICT r03 "do this" 3b): how fast can ONE launch of a chain of dependent launches stream its weight slab on an MI355X, by load path?
//   V   today's GEMV shape: 512 workgroups x 256 threads, 16-byte loads straight into VGPRs, a ring of 8 loads in flight per lane (128 KiB per CU), checksum consumer
//   L   LDS-DMA loader: 256 workgroups x 256 threads (one per CU); wave 0 issues `global_load_lds_dwordx4` (1 KiB per instruction, up to 48 in flight: the 6-bit vmcnt)
//       into a ring of 7 x 16 KiB slots and publishes landed slots through an LDS word; waves 1-3 poll that word, read the slot with ds_read_b128 and fold it into a
//       checksum (the guide's "1 loader + 3 consumers" engine, MI355X_MICROARCH.md rows ldsdma-fill / nt-weights), default cache policy or nt (aux = 2)
//   S   LDS-DMA without roles: every wave fills and consumes its own two 16 KiB buffers
// Each mode runs K launches captured in one graph on one stream (graph edges = the dependent-launch boundary of a decode pass), every launch on its own slab of a pool
// larger than the 256 MB infinity cache; all modes must produce the same checksum per launch.
//   build: hipcc --offload-arch=gfx950 -O2 scripts/probes/stream_probe.hip -o gpurun_out/stream_probe     run: timeout 120 gpurun_out/stream_probe [K=64] [reps=20] ["9 33 66"]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;
struct Args { const uint4 *w; long n16; unsigned *out; };       // n16 = 16-byte units of this launch's slab; out[4 * workgroup + wave] = that wave's checksum (2048 words per launch; folded on the host:
                                                               // one word per launch costs 2048 serialized atomics, ~24 us -- the first version of this probe measured exactly that)

__device__ __forceinline__ unsigned fold(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }
__device__ __forceinline__ unsigned wave_xor(unsigned v) {
    for (int o = 32; o > 0; o >>= 1) v ^= __shfl_xor(v, o, 64);
    return v;
}

template <int R, bool NT>
__global__ void __launch_bounds__(256) stream_v(Args a) {
    const long per = (a.n16 + gridDim.x - 1) / gridDim.x, i0 = (long)blockIdx.x * per, i1 = min(a.n16, i0 + per);
    uint4 ring[R]; unsigned acc = 0;
    long i = i0 + threadIdx.x;
    auto ld = [&](long j) -> uint4 { const uint4 *p = a.w + min(j, a.n16 - 1); if (NT) { typedef unsigned int u4 __attribute__((ext_vector_type(4))); const u4 v = __builtin_nontemporal_load(reinterpret_cast<const u4 *>(p)); return make_uint4(v[0], v[1], v[2], v[3]); } return *p; };
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
}

template <int AUX>
__global__ void __launch_bounds__(256) stream_l(Args a) {
    constexpr int NS = 7, SLOT = 16384, PF = 3;                    // ring slots, bytes per slot, slots in flight behind the newest one (16 x 3 = 48 <= 63 outstanding DMA instructions)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    volatile unsigned *ready = reinterpret_cast<volatile unsigned *>(smem + NS * SLOT), *done = ready + 1;      // ready: slots landed; done[c]: slots consumer c has finished
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long per = ((a.n16 + gridDim.x - 1) / gridDim.x + 1023) & ~1023L, i0 = (long)blockIdx.x * per, i1 = min(a.n16, i0 + per);       // whole slots per workgroup (1024 units = 16 KiB)
    const int nslots = i1 > i0 ? (int)((i1 - i0 + 1023) >> 10) : 0;
    if (threadIdx.x < 4) ready[threadIdx.x] = 0;
    __syncthreads();
    if (wave == 0) {
        const uint32_t base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem);
        for (int s = 0; s < nslots; ++s) {
            if (s >= NS) { const int o = s - NS; while (done[o % 3] <= (unsigned)(o / 3)) __builtin_amdgcn_s_sleep(1); }      // slot free?
            const uint4 *src = a.w + i0 + 1024L * s + lane;
#pragma unroll
            for (int j = 0; j < 16; ++j)
                __builtin_amdgcn_global_load_lds((glb_void_t *)(src + min(64L * j, a.n16 - 1 - (i0 + 1024L * s + lane))), (lds_void_t *)(uintptr_t)(base + (s % NS) * SLOT + j * 1024), 16, 0, AUX);
            if (s >= PF) { asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); if (lane == 0) *ready = (unsigned)(s - PF + 1); }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) *ready = (unsigned)nslots;
    } else {
        const int c = wave - 1; unsigned acc = 0;
        for (int s = c; s < nslots; s += 3) {
            while (*ready <= (unsigned)s) __builtin_amdgcn_s_sleep(1);
            const uint4 *slot = reinterpret_cast<const uint4 *>(smem + (s % NS) * SLOT);
            const long u0 = i0 + 1024L * s;
#pragma unroll
            for (int j = 0; j < 16; ++j) { const uint4 v = slot[64 * j + lane]; if (u0 + 64 * j + lane < i1) acc ^= fold(v); }
            __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0): the slot's reads are done before it is handed back
            if (lane == 0) done[c] = (unsigned)(s / 3 + 1);
        }
        acc = wave_xor(acc);
        if (lane == 0) a.out[4 * blockIdx.x + wave] = acc;
    }
}

// Round 6 (VERDICT r05 "do this" 6): the guide's ldsdma-fill row taken literally -- ring of NS x 16 KiB, ONE loader wave issuing from inline asm (the builtin makes hipcc treat the
// wave's LGKM counter as out of order), NC consumer waves, PF slots in flight behind the newest one, the loader never waits for a counter it does not need.
template <int AUX, int NS, int PF, int NC>
__global__ void __launch_bounds__(64 * (NC + 1)) stream_l2(Args a) {
    constexpr int SLOT = 16384;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    volatile unsigned *ready = reinterpret_cast<volatile unsigned *>(smem + NS * SLOT), *done = ready + 1;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long per = ((a.n16 + gridDim.x - 1) / gridDim.x + 1023) & ~1023L, i0 = (long)blockIdx.x * per, i1 = min(a.n16, i0 + per);
    const int nslots = i1 > i0 ? (int)((i1 - i0 + 1023) >> 10) : 0;
    if (threadIdx.x < 8) ready[threadIdx.x] = 0;
    __syncthreads();
    if (wave == 0) {
        const uint32_t base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem);
        for (int s = 0; s < nslots; ++s) {
            if (s >= NS) { const int o = s - NS; while (done[o % NC] <= (unsigned)(o / NC)) __builtin_amdgcn_s_sleep(1); }
            const uint4 *src = a.w + i0 + 1024L * s + lane;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const uint4 *gp = src + min(64L * j, a.n16 - 1 - (i0 + 1024L * s + lane));
                const uint32_t l = __builtin_amdgcn_readfirstlane(base + (s % NS) * SLOT + j * 1024);
                if (AUX == 2) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" :: "v"(gp), "s"(l) : "memory", "m0");
                else          asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gp), "s"(l) : "memory", "m0");
            }
            if (s >= PF) { if (PF == 3) asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); else if (PF == 2) asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                           if (lane == 0) *ready = (unsigned)(s - PF + 1); }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) *ready = (unsigned)nslots;
    } else {
        const int c = wave - 1; unsigned acc = 0;
        for (int s = c; s < nslots; s += NC) {
            while (*ready <= (unsigned)s) __builtin_amdgcn_s_sleep(1);
            const uint4 *slot = reinterpret_cast<const uint4 *>(smem + (s % NS) * SLOT);
            const long u0 = i0 + 1024L * s;
#pragma unroll
            for (int j = 0; j < 16; ++j) { const uint4 v = slot[64 * j + lane]; if (u0 + 64 * j + lane < i1) acc ^= fold(v); }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            if (lane == 0) done[c] = (unsigned)(s / NC + 1);
        }
        acc = wave_xor(acc);
        if (lane == 0) a.out[4 * blockIdx.x + wave] = acc;
    }
}

template <int AUX>
__global__ void __launch_bounds__(256) stream_s(Args a) {
    constexpr int SLOT = 16384;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long per = ((a.n16 + gridDim.x * 4 - 1) / (gridDim.x * 4) + 1023) & ~1023L, i0 = ((long)blockIdx.x * 4 + wave) * per, i1 = min(a.n16, i0 + per);
    const int nslots = i1 > i0 ? (int)((i1 - i0 + 1023) >> 10) : 0;
    const uint32_t base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(smem + wave * 2 * SLOT));
    unsigned acc = 0;
    auto fill = [&](int s) {
        const uint4 *src = a.w + i0 + 1024L * s + lane;
#pragma unroll
        for (int j = 0; j < 16; ++j)
            __builtin_amdgcn_global_load_lds((glb_void_t *)(src + min(64L * j, a.n16 - 1 - (i0 + 1024L * s + lane))), (lds_void_t *)(uintptr_t)(base + (s & 1) * SLOT + j * 1024), 16, 0, AUX);
    };
    if (nslots > 0) fill(0);
    for (int s = 0; s < nslots; ++s) {
        if (s + 1 < nslots) { fill(s + 1); asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint4 *slot = reinterpret_cast<const uint4 *>(smem + wave * 2 * SLOT + (s & 1) * SLOT);
        const long u0 = i0 + 1024L * s;
#pragma unroll
        for (int j = 0; j < 16; ++j) { const uint4 v = slot[64 * j + lane]; if (u0 + 64 * j + lane < i1) acc ^= fold(v); }
        __builtin_amdgcn_s_waitcnt(0xc07f);
    }
    acc = wave_xor(acc);
    if (lane == 0) a.out[4 * blockIdx.x + wave] = acc;
}

int main(int argc, char **argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 64, reps = argc > 2 ? atoi(argv[2]) : 20;
    const char *sizes = argc > 3 ? argv[3] : "9 33 66";
    hipStream_t st; CK(hipStreamCreate(&st));
    const size_t pool = (size_t)640 << 20;                      // > 256 MB infinity cache: rotating slabs stay cold
    uint4 *w; CK(hipMalloc(&w, pool));
    { std::vector<unsigned> h(pool / 4); unsigned x = 12345; for (auto &v : h) { x = x * 1664525u + 1013904223u; v = x; } CK(hipMemcpy(w, h.data(), pool, hipMemcpyHostToDevice)); }
    unsigned *out; CK(hipMalloc(&out, sizeof(unsigned) * K * 4096));
    CK(hipFuncSetAttribute((const void *)stream_l<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 7 * 16384 + 64)); CK(hipFuncSetAttribute((const void *)stream_l<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 7 * 16384 + 64));
    CK(hipFuncSetAttribute((const void *)stream_l2<0, 8, 3, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16384 + 64)); CK(hipFuncSetAttribute((const void *)stream_l2<2, 8, 3, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16384 + 64));
    CK(hipFuncSetAttribute((const void *)stream_l2<2, 9, 3, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 9 * 16384 + 64)); CK(hipFuncSetAttribute((const void *)stream_l2<2, 8, 2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16384 + 64));
    CK(hipFuncSetAttribute((const void *)stream_l2<2, 8, 3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16384 + 64));
    CK(hipFuncSetAttribute((const void *)stream_s<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16384)); CK(hipFuncSetAttribute((const void *)stream_s<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16384));
    char buf[256]; strncpy(buf, sizes, 255); buf[255] = 0;
    for (char *tok = strtok(buf, " "); tok; tok = strtok(nullptr, " ")) {
        const long mb = atol(tok), n16 = (mb << 20) / 16; const int nslabs = (int)(pool / ((size_t)mb << 20));
        std::vector<unsigned> ref;
        const char *names[] = {"V  vgpr ring  8 x 16 B,  512 wg", "V  vgpr ring  8 x 16 B,  256 wg", "L  lds-dma loader, default", "L  lds-dma loader, nt", "S  lds-dma self, default", "S  lds-dma self, nt",
                               "V  vgpr ring  4 x 16 B,  512 wg", "V  vgpr ring  4 x 16 B, 1024 wg", "V  vgpr ring  8 x 16 B,  768 wg", "V  vgpr ring  8 x 16 B, 1024 wg", "V  vgpr ring 16 x 16 B,  256 wg", "V  vgpr ring 16 x 16 B,  512 wg",
                               "V  vgpr ring  8 x 16 B nt, 256 wg", "V  vgpr ring  8 x 16 B nt, 512 wg", "V  vgpr ring 16 x 16 B nt, 256 wg",
                               "L2 asm loader 8 slots pf3 3c def", "L2 asm loader 8 slots pf3 3c nt", "L2 asm loader 9 slots pf3 3c nt", "L2 asm loader 8 slots pf2 3c nt", "L2 asm loader 8 slots pf3 2c nt"};
        const int first = argc > 4 ? atoi(argv[4]) : 0, last = argc > 5 ? atoi(argv[5]) : 15;
        for (int mode = first; mode < last; ++mode) {
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            CK(hipMemsetAsync(out, 0, sizeof(unsigned) * K * 4096, st));
            for (int i = 0; i < K; ++i) {
                Args a{w + (size_t)(i % nslabs) * n16, n16, out + 4096 * i};
                switch (mode) {
                    case 0: hipLaunchKernelGGL((stream_v<8, false>), dim3(512), dim3(256), 0, st, a); break;
                    case 1: hipLaunchKernelGGL((stream_v<8, false>), dim3(256), dim3(256), 0, st, a); break;
                    case 2: hipLaunchKernelGGL(stream_l<0>, dim3(256), dim3(256), 7 * 16384 + 64, st, a); break;
                    case 3: hipLaunchKernelGGL(stream_l<2>, dim3(256), dim3(256), 7 * 16384 + 64, st, a); break;
                    case 4: hipLaunchKernelGGL(stream_s<0>, dim3(256), dim3(256), 8 * 16384, st, a); break;
                    case 5: hipLaunchKernelGGL(stream_s<2>, dim3(256), dim3(256), 8 * 16384, st, a); break;
                    case 6: hipLaunchKernelGGL((stream_v<4, false>), dim3(512), dim3(256), 0, st, a); break;
                    case 7: hipLaunchKernelGGL((stream_v<4, false>), dim3(1024), dim3(256), 0, st, a); break;
                    case 8: hipLaunchKernelGGL((stream_v<8, false>), dim3(768), dim3(256), 0, st, a); break;
                    case 9: hipLaunchKernelGGL((stream_v<8, false>), dim3(1024), dim3(256), 0, st, a); break;
                    case 10: hipLaunchKernelGGL((stream_v<16, false>), dim3(256), dim3(256), 0, st, a); break;
                    case 11: hipLaunchKernelGGL((stream_v<16, false>), dim3(512), dim3(256), 0, st, a); break;
                    case 12: hipLaunchKernelGGL((stream_v<8, true>), dim3(256), dim3(256), 0, st, a); break;
                    case 13: hipLaunchKernelGGL((stream_v<8, true>), dim3(512), dim3(256), 0, st, a); break;
                    case 14: hipLaunchKernelGGL((stream_v<16, true>), dim3(256), dim3(256), 0, st, a); break;
                    case 15: hipLaunchKernelGGL((stream_l2<0, 8, 3, 3>), dim3(256), dim3(256), 8 * 16384 + 64, st, a); break;
                    case 16: hipLaunchKernelGGL((stream_l2<2, 8, 3, 3>), dim3(256), dim3(256), 8 * 16384 + 64, st, a); break;
                    case 17: hipLaunchKernelGGL((stream_l2<2, 9, 3, 3>), dim3(256), dim3(256), 9 * 16384 + 64, st, a); break;
                    case 18: hipLaunchKernelGGL((stream_l2<2, 8, 2, 3>), dim3(256), dim3(256), 8 * 16384 + 64, st, a); break;
                    case 19: hipLaunchKernelGGL((stream_l2<2, 8, 3, 2>), dim3(256), dim3(192), 8 * 16384 + 64, st, a); break;
                }
            }
            CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
            std::vector<unsigned> raw((size_t)K * 4096), got(K, 0u); CK(hipMemcpy(raw.data(), out, sizeof(unsigned) * K * 4096, hipMemcpyDeviceToHost));
            for (int i = 0; i < K; ++i) for (int j = 0; j < 4096; ++j) got[i] ^= raw[(size_t)i * 4096 + j];
            if (mode == 0) ref = got;
            const bool same = got == ref;
            float best = 1e30f;
            for (int r = 0; r < reps; ++r) { CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; }
            const double us = best * 1e3 / K;
            printf("%3ld MB  %-34s %7.2f us / launch  %6.2f TB/s  %s\n", mb, names[mode], us, (double)mb * 1048576.0 / us / 1e6, same ? "checksums ok" : "CHECKSUM MISMATCH");
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
    }
    return 0;
}
