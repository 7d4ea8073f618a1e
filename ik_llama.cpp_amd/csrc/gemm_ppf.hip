// gemm_ppf.hip -- the f16 x f16 prompt GEMM of large batches (gemm_ppf.cuh): type-independent, one translation unit.
#include "gemm_ppf.cuh"

// 0 = launched, -2 = HIP failure.  a.A = the f16 weight image (dequant_slab_kernel), a.A2 != nullptr only as the "fused up*gate" flag, a.pairing from the image's launcher.
int cdna4_gemm_ppf_launch(int num_cu, const GemmArgs &a_in, hipStream_t st) {
    (void)num_cu;
    GemmArgs a = a_in;
    const int rows = a.A2 ? 128 : 256;
    const long mt = (a.M + rows - 1) / rows, ntl = (a.N + WLDS_BT - 1) / WLDS_BT, wgs = mt * ntl;
    { const long budget = 4L << 20, tile_bytes = (long)WLDS_BT * a.K * 2; long G = 1;      // super-columns of token tiles whose activations fit an XCD's L2 (as launch_gemm_wlds)
      for (long d = 1; d <= ntl; ++d) if (ntl % d == 0 && d * tile_bytes <= budget) G = d;
      a.m_major = (int)G; }
    const size_t lds = 4 * WLDS_STAGE + PF_TL_BYTES;
    if (a.A2) {
        if (cdna4_opt_in_lds((const void *)gemm_ppf_kernel<true>) != 0) return -2;
        hipLaunchKernelGGL((gemm_ppf_kernel<true>), dim3((unsigned)wgs), dim3(512), lds, st, a);
    } else {
        if (cdna4_opt_in_lds((const void *)gemm_ppf_kernel<false>) != 0) return -2;
        hipLaunchKernelGGL((gemm_ppf_kernel<false>), dim3((unsigned)wgs), dim3(512), lds, st, a);
    }
    cdna4_note_launch("gemm_ppf type=1 nt=8 upgate=%d kx=64 ks=1 mw=2 xw=0 part=0 grid=%ldx1x1 ksplit=1 g=%d", a.A2 ? 1 : 0, wgs, a.m_major);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
