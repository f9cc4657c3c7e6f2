// Probe: operand layout of v_mfma_f32_32x32x16_bf16 on gfx950.
// Hypothesis: A[i][k]: lane = i + 32*(k/8), element e = k%8;  B[k][j]: lane = j + 32*(k/8), e = k%8;
//             D[row][col]: lane = col + 32*((row>>2)&1), reg = (row&3) + 4*(row>>3).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void probe(const float* A, const float* B, float* D) {  // A[32][16], B[16][32] row-major, D[32][32]
  int l = threadIdx.x, h = l >> 5, c = l & 31;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)A[c * 16 + 8 * h + e]; b[e] = (__bf16)B[(8 * h + e) * 32 + c]; }
  f32x16 acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + c] = acc[r];
}
int main() {
  float hA[512], hB[512], hD[1024], ref[1024];
  srand(1);
  for (int i = 0; i < 512; ++i) { hA[i] = (float)((rand() % 17) - 8); hB[i] = (float)((rand() % 13) - 6); }  // exact in bf16
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int k = 0; k < 16; ++k) s += hA[i*16+k]*hB[k*32+j]; ref[i*32+j] = s; }
  float *dA, *dB, *dD; hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dD, 4096);
  hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
  double err = 0; for (int i = 0; i < 1024; ++i) err = fmax(err, fabs(hD[i] - ref[i]));
  printf("mfma_f32_32x32x16_bf16 layout probe: max err %g -> %s\n", err, err == 0 ? "HYPOTHESIS OK" : "MISMATCH");
  return err != 0;
}
