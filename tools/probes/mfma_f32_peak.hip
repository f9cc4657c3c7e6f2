// What does the fp32 matrix pipe deliver when NOTHING else is in the way?  (development probe, not part of the
// library: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_f32_peak.hip -o tools/probes/mfma_f32_peak)
// Each wave issues a long run of v_mfma_f32_32x32x2_f32 on NACC independent accumulators (no memory traffic);
// 1, 2 or 4 waves per SIMD.  Prints TFLOP/s against the 157.3 TFLOP/s data-sheet peak.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(256) probe(float *out, int iters, float a, float b) {
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = (float)(threadIdx.x + r);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run(int waves_per_simd, float *out) {
  const int blocks = 256 * waves_per_simd;  // 256 CUs x (4 waves per block = one per SIMD) x waves_per_simd
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<NACC>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f, 0.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 * iters * 8 * NACC * (2.0 * 32 * 32 * 2);
  printf("%d independent accumulators, %d wave(s) per SIMD: %.2f ms, %.1f TFLOP/s = %.1f %% of 157.3\n", NACC, waves_per_simd, ms,
         flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
}

int main() {
  float *out;
  hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  for (int w : {1, 2, 4}) run<1>(w, out);
  for (int w : {1, 2, 4}) run<4>(w, out);
  return 0;
}
