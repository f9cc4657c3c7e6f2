// mfma_power_probe.hip -- what the whole chip sustains on a pure v_mfma_f32_32x32x16_bf16 stream (1024 waves, one per
// SIMD, 4 accumulators in rotation, nothing else in the loop), with constant and with random operands: wall-clock PFLOP/s
// and the shader clock implied by the s_memtime ticks of one wave (ticks per MFMA stay at 32.x: the clock gives way).
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_pw tools/probes/mfma_power_probe.hip && /tmp/mfma_pw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) k(long long *out, float *sink, int iters, int random) {
  bf16x8 a, b;
  unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
  for (int e = 0; e < 8; ++e) {
    h = h * 1664525u + 1013904223u;
    const float ra = random ? (float)(int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f : 1.0f;
    h = h * 1664525u + 1013904223u;
    const float rb = random ? (float)(int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f : 0.5f;
    a[e] = (__bf16)ra, b[e] = (__bf16)(rb * 0.01f);
  }
  f32x16 acc[4];
  for (int d = 0; d < 4; ++d) acc[d] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 12; ++r)
#pragma unroll
      for (int d = 0; d < 4; ++d) acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[d], 0, 0, 0);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int d = 0; d < 4; ++d)
    for (int r = 0; r < 16; ++r) s += acc[d][r];
  sink[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 7) out[0] = t1 - t0;
}

int main() {
  long long *out;
  float *sink;
  hipMalloc(&out, 64);
  hipMalloc(&sink, 256 * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 40000;  // 1.92 M MFMAs per wave: ~30 ms
  for (int random = 0; random < 2; ++random)
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, out, sink, iters, random);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      long long c;
      hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost);
      const double mfmas = (double)iters * 48, flop = mfmas * 1024 * 32768.0;
      printf("%s operands: %.2f ms, %.3f PFLOP/s, %.1f ticks per MFMA, ticks / wall = %.2f GHz\n", random ? "random  " : "constant", ms,
             flop / (ms * 1e-3) / 1e15, (double)c / mfmas, (double)c / (ms * 1e-3) / 1e9);
    }
  return 0;
}
