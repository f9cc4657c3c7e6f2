// mfma_regfile_probe.hip -- cycles per v_mfma_f32_32x32x16_bf16 (one wave per SIMD, 4 accumulators in rotation) by the
// register file each operand comes from: V = architectural VGPR, A = accumulation VGPR.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_rf tools/probes/mfma_regfile_probe.hip && /tmp/mfma_rf
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define KERNEL(NAME, CA, CB, CC)                                                                                     \
  __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) NAME(long long *out, float *sink, int iters) { \
    bf16x8 a, b;                                                                                                     \
    for (int e = 0; e < 8; ++e) a[e] = (__bf16)(float)(threadIdx.x + e), b[e] = (__bf16)(float)(e + 1);             \
    f32x16 acc[4];                                                                                                   \
    for (int d = 0; d < 4; ++d) acc[d] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};                     \
    const long long t0 = __builtin_amdgcn_s_memtime();                                                               \
    for (int it = 0; it < iters; ++it) {                                                                             \
      _Pragma("unroll") for (int r = 0; r < 12; ++r) _Pragma("unroll") for (int d = 0; d < 4; ++d)                   \
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+" CC(acc[d]) : CA(a), CB(b));                   \
    }                                                                                                                \
    const long long t1 = __builtin_amdgcn_s_memtime();                                                               \
    float s = 0;                                                                                                     \
    for (int d = 0; d < 4; ++d)                                                                                      \
      for (int r = 0; r < 16; ++r) s += acc[d][r];                                                                   \
    sink[blockIdx.x * 256 + threadIdx.x] = s;                                                                        \
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                                                       \
  }
KERNEL(k_vvv, "v", "v", "v")
KERNEL(k_vva, "v", "v", "a")
KERNEL(k_vaa, "v", "a", "a")
KERNEL(k_vav, "v", "a", "v")
KERNEL(k_aaa, "a", "a", "a")
KERNEL(k_avv, "a", "v", "v")

template <class K>
void run(const char *name, K kern, long long *out, float *sink) {
  const int iters = 200;
  hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, out, sink, iters);
  hipDeviceSynchronize();
  long long c;
  hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost);
  printf("A,B,C(=D) in %s: %.1f cycles per MFMA\n", name, (double)c / (iters * 48));
}
int main() {
  long long *out;
  float *sink;
  hipMalloc(&out, 64);
  hipMalloc(&sink, 256 * 256 * 4);
  run("V,V,V", k_vvv, out, sink);
  run("V,V,A", k_vva, out, sink);
  run("V,A,A", k_vaa, out, sink);
  run("V,A,V", k_vav, out, sink);
  run("A,A,A", k_aaa, out, sink);
  run("A,V,V", k_avv, out, sink);
  return 0;
}
