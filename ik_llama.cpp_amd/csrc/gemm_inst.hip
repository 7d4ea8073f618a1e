// gemm_inst.hip -- one translation unit per weight type: compiled with -DINST_TYPE=<ggml_type> (ik_llama.cpp_amd/build.py).
// Instantiates the prefill MFMA kernels of gemm_mfma.cuh for that type and exports its launcher.
#include "gemm_mfma.cuh"
#include "gemm_wlds.cuh"
#include "gemm_pp.cuh"

#ifndef INST_TYPE
#error "compile with -DINST_TYPE=<ggml_type>"
#endif
#define CAT2_(a, b) a##b
#define CAT2(a, b) CAT2_(a, b)

// grouped_nt == 0: dense / multi-matrix launch (tile shape chosen by launch_gemm_type); > 0: MUL_MAT_ID grouped form with 32 * grouped_nt token tiles.
// returns 0, or -2 on a HIP failure
int CAT2(cdna4_gemm_launch_, INST_TYPE)(int num_cu, const GemmArgs &a, int grouped_nt, hipStream_t st) {
    if (grouped_nt > 0) return launch_gemm_grouped<INST_TYPE>(grouped_nt, a, st);
    { const int rc = launch_gemm_pp<INST_TYPE>(num_cu, a, st); if (rc <= 0) return rc; }        // 256-token tiles, the two waves of a SIMD alternating between matrix and load / de-quantize intervals
    { const int rc = launch_gemm_wlds<INST_TYPE>(num_cu, a, st); if (rc <= 0) return rc; }      // 256-token tiles with the weight tile de-quantized once per workgroup, where that grid fills the chip
    return launch_gemm_type<INST_TYPE>(num_cu, a, st);
}
// large batches: the weights of this type as an f16 image for gemm_ppf_kernel (gemm_pp.cuh dequant_slab_kernel); 0, or -2 on a HIP failure
int CAT2(cdna4_dequant_slab_launch_, INST_TYPE)(const GemmArgs &a, void *w16, int *pairing, hipStream_t st) {
#if INST_TYPE == 1
    (void)a; (void)w16; (void)pairing; (void)st; return -1;      // (f16 weights: no image of their own)
#else
    return launch_dequant_slab<INST_TYPE>(a, w16, pairing, st);
#endif
}
// the runtime loads a translation unit's code object at the first launch of one of its kernels (a few ms for these: zstd-compressed, dozens of instantiations); asking for a
// kernel's attributes loads it now -- cdna4_preload_type, called when weights of this type are uploaded
int CAT2(cdna4_gemm_preload_, INST_TYPE)(void) {
    hipFuncAttributes at;
    return hipFuncGetAttributes(&at, (const void *)gemm_mfma_kernel<INST_TYPE, 4, false, 128, 1>) == hipSuccess ? 0 : -2;
}
