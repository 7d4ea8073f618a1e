// Probe (round 5): the operand / result layouts of v_mfma_f32_16x16x32_f16 and the gather of ds_read_b64_tr_b16 as csrc/fa_decode_mfma.cuh assumes them.
//   MFMA: A lane l = A[m = l % 16][k = 8 (l / 16) + 0..7], B lane l = B[k = 8 (l / 16) + 0..7][n = l % 16], D lane l, i = D[m = 4 (l / 16) + i][n = l % 16]
//   tr16: image rows of 16 halves (32 B); lane l gives the address of halves 4 (l % 16) .. + 3 of the [4][16] block its 16-lane group reads (block base + 8 (l % 16) bytes) and
//         receives column l % 16 of that block: rows 0..3
//   build: hipcc --offload-arch=gfx950 -O2 scripts/probes/mfma_layout_probe.hip -o scripts/probes/bin/mfma_layout_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void probe(const float *A, const float *B, float *D, float *T) {      // A [16][32], B [32][16], D [16][16]; T [64][4]: what each lane's tr read returned
    const int l = threadIdx.x;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)A[(l % 16) * 32 + 8 * (l / 16) + j]; b[j] = (_Float16)B[(8 * (l / 16) + j) * 16 + l % 16]; }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[(4 * (l / 16) + i) * 16 + l % 16] = c[i];
    __shared__ __attribute__((aligned(16))) _Float16 img[32 * 16];        // [32 keys][16 dims], value = 16 key + dim
    for (int i = l; i < 512; i += 64) img[i] = (_Float16)i;
    __syncthreads();
    const int g = l / 16, i16 = l % 16;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(img + (4 * g) * 16 + 4 * i16));
    f16x4 h = __builtin_bit_cast(f16x4, v);
    for (int j = 0; j < 4; ++j) T[l * 4 + j] = (float)h[j];
}
int main() {
    float hA[512], hB[512], hD[256], hT[256], *dA, *dB, *dD, *dT;
    for (int i = 0; i < 512; ++i) { hA[i] = (float)(rand() % 7 - 3); hB[i] = (float)(rand() % 5 - 2); }
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dD, 1024); hipMalloc(&dT, 1024);
    hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, dT);
    hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost); hipMemcpy(hT, dT, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { float s = 0; for (int k = 0; k < 32; ++k) s += hA[m * 32 + k] * hB[k * 16 + n]; if (s != hD[m * 16 + n]) ++bad; }
    printf("mfma_f32_16x16x32_f16 layout: %d of 256 results differ from A x B\n", bad);
    int badt = 0;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) { const float want = (float)((4 * (l / 16) + j) * 16 + l % 16); if (hT[l * 4 + j] != want) { if (badt < 8) printf("  tr lane %d elem %d: got %g want %g\n", l, j, hT[l * 4 + j], want); ++badt; } }
    printf("ds_read_b64_tr_b16 gather: %d of 256 elements differ from column (l %% 16) of the group's [4][16] block\n", badt);
    return bad || badt;
}
