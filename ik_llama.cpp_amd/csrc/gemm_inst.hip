// gemm_inst.hip -- one translation unit per weight type: compiled with -DINST_TYPE=<ggml_type> (ik_llama.cpp_amd/build.py).
// Instantiates the prefill MFMA kernels of gemm_mfma.cuh for that type and exports its launcher.
#include "gemm_mfma.cuh"

#ifndef INST_TYPE
#error "compile with -DINST_TYPE=<ggml_type>"
#endif
#define CAT2_(a, b) a##b
#define CAT2(a, b) CAT2_(a, b)

// grouped_nt == 0: dense / multi-matrix launch (tile shape chosen by launch_gemm_type); > 0: MUL_MAT_ID grouped form with 32 * grouped_nt token tiles.
// returns 0, or -2 on a HIP failure
int CAT2(cdna4_gemm_launch_, INST_TYPE)(int num_cu, const GemmArgs &a, int grouped_nt, hipStream_t st) {
    if (grouped_nt > 0) return launch_gemm_grouped<INST_TYPE>(grouped_nt, a, st);
    return launch_gemm_type<INST_TYPE>(num_cu, a, st);
}
