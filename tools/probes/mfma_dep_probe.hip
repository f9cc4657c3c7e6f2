// mfma_dep_probe.hip -- cycles per v_mfma_f32_32x32x16_bf16 on ONE wave per SIMD as a function of the distance between two
// MFMAs that accumulate into the same registers (D accumulators in rotation), and of where the accumulators live.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_dep tools/probes/mfma_dep_probe.hip && /tmp/mfma_dep
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int D>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) k(long long *out, float *sink, int iters) {
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) a[e] = (__bf16)(float)(threadIdx.x + e), b[e] = (__bf16)(float)(e + 1);
  f32x16 acc[D];
  for (int d = 0; d < D; ++d) acc[d] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 48 / D; ++r)
#pragma unroll
      for (int d = 0; d < D; ++d) {
        acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[d], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int d = 0; d < D; ++d)
    for (int r = 0; r < 16; ++r) s += acc[d][r];
  sink[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int D>
void run(long long *out, float *sink) {
  const int iters = 200;
  hipLaunchKernelGGL(k<D>, dim3(256), dim3(256), 0, 0, out, sink, iters);
  hipDeviceSynchronize();
  long long c;
  hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost);
  printf("accumulators in rotation %d: %.1f cycles per MFMA\n", D, (double)c / (iters * 48));
}
int main() {
  long long *out;
  float *sink;
  hipMalloc(&out, 64);
  hipMalloc(&sink, 256 * 256 * 4);
  run<1>(out, sink);
  run<2>(out, sink);
  run<3>(out, sink);
  run<4>(out, sink);
  run<6>(out, sink);
  run<8>(out, sink);
  return 0;
}
