// gemv_attn.hip -- one decoded token: FLASH_ATTN_EXT + the attn_output MUL_MAT + its residual ADD as ONE launch.
//
// Why: the decode attention of a short context is latency-bound on 32 of 256 CUs (one workgroup per q head: ~7-9 us per layer with the chip idle), and the launch behind it
// -- the 9.4 MB attn_output matrix of an 8B model -- is all boundary, ramp and first-load latency (6 us for 1.7 us of streaming).  Here the first n_fa workgroups of the grid
// ARE the attention (fa_decode.cuh, results stored write-through), and the others are the mat-vec (gemv.cuh FX = 5): they request their weights at once -- a 4096 x 4096
// matrix is entirely in flight in the ring registers of 256 workgroups -- wait for the attention workgroups' tickets, fetch the 16 KB activation row with agent-scope loads,
// quantize it and finish.  All workgroups of the grid are co-resident (<= 2 per CU), so the wait cannot starve a producer; the spin is bounded all the same.
// Results: bit-identical to the two launches (same attention body, same mat-vec body, same order of every sum) -- tests/test_gpu_attn_fused.py.
// Reference being replaced: ggml-cuda/fattn-vec-f16.cuh + ggml-cuda/mmvq.cu + binbcast.cu for these three nodes of llm_build_kqv (src/llama-build-context.cpp).
#include "gemv_launch.cuh"
#include "fa_decode.cuh"

struct FaDecodeArgs { TD q, k, v, mask, dst; int has_mask; float scale, softcap, max_bias, m0, m1; unsigned n_head_log2; unsigned *sync; };

template <int TYPE, int VDT>
__global__ void __launch_bounds__(256) gemv_attn_kernel(const FaDecodeArgs f, const GemvArgs a, const int n_fa) {
    if ((int)blockIdx.x < n_fa) {                                    // attention of q head blockIdx.x (one token)
        __shared__ float s_m[4], s_l[4]; __shared__ float s_acc[4][128];
        fa_decode_body_v2<true>(f.q, f.k, f.v, f.mask, f.has_mask, f.dst, f.scale, f.softcap, f.max_bias, f.m0, f.m1, f.n_head_log2, 0, blockIdx.x, 0, s_m, s_l, s_acc);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's write-through stores have left the XCD
        __syncthreads();
        if (threadIdx.x == 0) (void)__hip_atomic_fetch_add(f.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    gemv_body<TYPE, 1, false, 1, VDT, GEMV_DEPTH, false, 1, 64, 5>(a, blockIdx.x - n_fa, gridDim.x - n_fa);
}

template <int TYPE>
static int launch_attn(const cdna4_context *ctx, const FaDecodeArgs &f, const GemvArgs &a, int n_fa, hipStream_t st) {
    constexpr int VDT = T_Q8_2_X4;
    const size_t lds = gemv_lds_bytes<VDT>(1, a.K, TYPE);
    long wgs; int wpw;
    gemv_grid(ctx, a.M, a.K, 1, 1, 1, lds, 1, wgs, wpw);
    if (wpw != 4 || lds > 48 * 1024) return -1;                      // (256-thread workgroups; the attention workgroups add 2.1 KiB of static LDS)
    if (wgs + n_fa > 2L * ctx->num_cu) wgs = 2L * ctx->num_cu - n_fa; // every workgroup resident from the start (<= 2 per CU)
    if (wgs < 1) return -1;
    hipLaunchKernelGGL((gemv_attn_kernel<TYPE, VDT>), dim3((unsigned)(wgs + n_fa)), dim3(256), lds, st, f, a, n_fa);
    HIP_TRY(hipGetLastError());
    return CDNA4_OK;
}

// -1: shape / type not served (the caller issues the three nodes one by one)
int cdna4_gemv_attn_launch(const cdna4_context *ctx, int type, const cdna4_tensor *q, const cdna4_tensor *k, const cdna4_tensor *v, const cdna4_tensor *mask, const cdna4_tensor *attn,
                           float scale, float max_bias, float softcap, const GemvArgs &a, unsigned *sync, hipStream_t st) {
    const long n_head = q->ne[2];
    if (q->ne[0] != 128 || q->ne[1] != 1 || q->ne[3] != 1 || n_head < 1 || n_head > 256 || (a.K >> 6) <= 32 || (a.K >> 6) > 64 || a.K != n_head * 128) return -1;
    FaDecodeArgs f; memset(&f, 0, sizeof(f));
    f.q = td_of(q); f.k = td_of(k); f.v = td_of(v); f.dst = td_of(attn); f.has_mask = mask ? 1 : 0;
    if (mask) f.mask = td_of(mask); else { f.mask.ne[2] = f.mask.ne[3] = 1; }
    if (softcap != 0.0f) scale /= softcap;
    f.scale = scale; f.softcap = softcap; f.max_bias = max_bias; f.n_head_log2 = 1u << (unsigned)floorf(log2f((float)n_head));
    f.m0 = powf(2.0f, -max_bias / f.n_head_log2); f.m1 = powf(2.0f, -(max_bias / 2.0f) / f.n_head_log2); f.sync = sync;
    GemvArgs g = a; g.fa_sync = sync; g.fa_expect = (unsigned)n_head;
    switch (type) {
        case T_Q4_K: return launch_attn<T_Q4_K>(ctx, f, g, (int)n_head, st);
        case T_Q5_K: return launch_attn<T_Q5_K>(ctx, f, g, (int)n_head, st);
        case T_Q6_K: return launch_attn<T_Q6_K>(ctx, f, g, (int)n_head, st);
        case T_IQ4_NL: return launch_attn<T_IQ4_NL>(ctx, f, g, (int)n_head, st);
    }
    return -1;
}
