// gemv_inst.hip -- one translation unit per (weight type, plain | fused up*gate): compiled with -DINST_TYPE=<ggml_type> -DINST_UPGATE=<0|1>
// (ik_llama.cpp_amd/build.py).  Instantiates the decode GEMV kernels of gemv.cuh for that type and exports its launcher.
#include "gemv_launch.cuh"

#ifndef INST_TYPE
#error "compile with -DINST_TYPE=<ggml_type> -DINST_UPGATE=<0|1>"
#endif
#define CAT3_(a, b, c) a##b##c
#define CAT3(a, b, c) CAT3_(a, b, c)
#if INST_UPGATE
#define FN_NAME CAT3(cdna4_gemv_launch_, INST_TYPE, _upgate)
#else
#define FN_NAME CAT3(cdna4_gemv_launch_, INST_TYPE, _plain)
#endif

// `vdt` = the activation quantization to reproduce: type_vec_dot of the tensor's type (the _R4 kernels' for weights that arrived
// row-interleaved: Q8_K32 for Q4_K/Q5_K, Q8_K for Q6_K)
int FN_NAME(const cdna4_context *ctx, int vdt, const GemvArgs &a, int ncols, unsigned grid_y, hipStream_t st) {
    constexpr int TYPE = INST_TYPE; constexpr bool UP = INST_UPGATE != 0;
    if constexpr (TYPE == T_Q4_K || TYPE == T_Q5_K)
        return vdt == T_Q8_K32 ? launch_gemv_t<TYPE, UP, T_Q8_K32>(ctx, a, ncols, grid_y, st) : launch_gemv_t<TYPE, UP, T_Q8_2_X4>(ctx, a, ncols, grid_y, st);
    else if constexpr (TYPE == T_Q6_K)
        return vdt == T_Q8_K ? launch_gemv_t<TYPE, UP, T_Q8_K>(ctx, a, ncols, grid_y, st) : launch_gemv_t<TYPE, UP, T_Q8_2_X4>(ctx, a, ncols, grid_y, st);
    else if constexpr (TYPE == T_IQ4_NL || TYPE == T_Q4_0 || TYPE == T_Q8_0 || TYPE == T_Q5_0 || TYPE == T_Q4_1 || TYPE == T_Q5_1 || TYPE == T_Q6_0 || TYPE == T_MXFP4 || type_is_kt(TYPE))
        return launch_gemv_t<TYPE, UP, T_Q8_2_X4>(ctx, a, ncols, grid_y, st);
    else
        return launch_gemv_t<TYPE, UP, T_Q8_K>(ctx, a, ncols, grid_y, st);
}
