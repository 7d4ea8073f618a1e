// gemm_mfma.cuh -- prefill (Ny > 8) dequant -> f16 MFMA GEMM.  (stub: implemented next)
#pragma once
#include "cdna4_common.cuh"
static inline bool gemm_mfma_supported(int) { return false; }
static inline int launch_gemm_mfma(int, int, long, long, long, const uint8_t *, const uint8_t *, long, const __half *, float *, long, int, const uint16_t *, hipStream_t) { return -1; }
