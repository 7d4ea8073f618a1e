// gemv_dual.hip -- the two-type-group decode launch (Q4_K_M / Q5_K_M attention: q,k in Q4_K / Q5_K next to a Q6_K attn_v)
#include "gemv_launch.cuh"

// two type groups in one launch (gemv_dual_kernel): A = {Q4_K | Q5_K} group (possibly several matrices), B = one Q6_K matrix, N = 1, K <= 16384
template <int TA, int YITERS>
static int launch_gemv_dual_y(const cdna4_context *ctx, const GemvArgs &a, const GemvArgs &b, hipStream_t st) {
    constexpr int VA = T_Q8_2_X4, VB = T_Q8_2_X4;                      // type_vec_dot of the base types (Q6_K too: mul_mat_qY_K_q8_2_X4_T, a5)
    const size_t lds = std::max(gemv_lds_bytes<VA>(1, a.K, TA), gemv_lds_bytes<VB>(1, b.K, T_Q6_K));
    if (lds > 64 * 1024) {
        int rc = cdna4_opt_in_lds((const void *)gemv_dual_kernel<TA, VA, true, T_Q6_K, VB, YITERS, 64>); if (rc) return rc;
        rc = cdna4_opt_in_lds((const void *)gemv_dual_kernel<TA, VA, true, T_Q6_K, VB, YITERS, 0>); if (rc) return rc;
    }
    long wa, wb; int wpa, wpb;
    gemv_grid(ctx, a.M, a.K, 1, YITERS, 1, lds, 1, wa, wpa); gemv_grid(ctx, b.M, b.K, 1, YITERS, 1, lds, 1, wb, wpb);
    if (wpa != wpb) return -1;                                        // (same K => same workgroup size; defensive)
    if (a.R || b.R || (a.norm_w != b.norm_w)) return -1;
    if constexpr (YITERS == 1) {
        if (a.norm_w) {          // RMS norm of the shared activation row fused into the prologue (gemv.cuh FX = 1)
            if ((long)(a.K >> 3) > (long)xpre_for(1, YITERS, (a.K >> 6) > 32 ? 64 : 0) * 64 * wpa) return -1;
            const size_t ldn = lds + 64;
            if (a.rope_tab || b.rope_tab) {        // q,k,v epilogue (gemv.cuh FX = 4) on both groups
                if (!(a.rope_tab && b.rope_tab) || (a.K >> 6) <= 32 || a.M % 2 || b.M % 2) return -1;
                if (ldn > 64 * 1024) { const int rc = cdna4_opt_in_lds((const void *)gemv_dual_kernel<TA, VA, true, T_Q6_K, VB, YITERS, 64, 4>); if (rc) return rc; }
                hipLaunchKernelGGL((gemv_dual_kernel<TA, VA, true, T_Q6_K, VB, YITERS, 64, 4>), dim3((unsigned)(wa + wb)), dim3(64 * wpa), ldn, st, a, b, (int)wa);
                HIP_TRY(hipGetLastError());
                return CDNA4_OK;
            }
            if (ldn > 64 * 1024) { int rc = cdna4_opt_in_lds((const void *)gemv_dual_kernel<TA, VA, true, T_Q6_K, VB, YITERS, 64, 1>); if (rc) return rc; rc = cdna4_opt_in_lds((const void *)gemv_dual_kernel<TA, VA, true, T_Q6_K, VB, YITERS, 0, 1>); if (rc) return rc; }
            if ((a.K >> 6) > 32) hipLaunchKernelGGL((gemv_dual_kernel<TA, VA, true, T_Q6_K, VB, YITERS, 64, 1>), dim3((unsigned)(wa + wb)), dim3(64 * wpa), ldn, st, a, b, (int)wa);
            else                 hipLaunchKernelGGL((gemv_dual_kernel<TA, VA, true, T_Q6_K, VB, YITERS, 0, 1>), dim3((unsigned)(wa + wb)), dim3(64 * wpa), ldn, st, a, b, (int)wa);
            HIP_TRY(hipGetLastError());
            return CDNA4_OK;
        }
    }
    if (a.norm_w || a.rope_tab || b.rope_tab) return -1;
    if ((a.K >> 6) > 32) hipLaunchKernelGGL((gemv_dual_kernel<TA, VA, true, T_Q6_K, VB, YITERS, 64>), dim3((unsigned)(wa + wb)), dim3(64 * wpa), lds, st, a, b, (int)wa);
    else                 hipLaunchKernelGGL((gemv_dual_kernel<TA, VA, true, T_Q6_K, VB, YITERS, 0>), dim3((unsigned)(wa + wb)), dim3(64 * wpa), lds, st, a, b, (int)wa);
    HIP_TRY(hipGetLastError());
    return CDNA4_OK;
}
int cdna4_gemv_dual_launch(const cdna4_context *ctx, int type_a, const GemvArgs &a, const GemvArgs &b, hipStream_t st) {
    const int U = a.K >> 6, iters = U <= 64 ? 1 : (U + 63) / 64;
    if (iters > 4) return -1;
#define DUAL(TA) case TA: return iters == 1 ? launch_gemv_dual_y<TA, 1>(ctx, a, b, st) : iters == 2 ? launch_gemv_dual_y<TA, 2>(ctx, a, b, st) : launch_gemv_dual_y<TA, 4>(ctx, a, b, st);
    switch (type_a) { DUAL(T_Q4_K) DUAL(T_Q5_K) }
#undef DUAL
    return -1;
}
