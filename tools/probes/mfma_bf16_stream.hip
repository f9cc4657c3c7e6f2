// Micro-benchmark: what one 512-register wave per SIMD can sustain on v_mfma_f32_32x32x16_bf16, chip-wide.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_bf16_stream tools/probes/mfma_bf16_stream.hip
//   ./tools/probes/mfma_bf16_stream            (on the MI355X box)
// Every variant runs `tiles` x 288 MFMAs per wave on 256 CUs x 4 waves (one per SIMD), two accumulators alternating, and
// differs in what rides in the gaps between MFMAs -- the ingredients of sa2_bf16x3_persistent_kernel's tile loop:
//   0  nothing (operands fixed in registers)
//   1  one ds_read_b128 per MFMA into a rotating operand ring (the layer-3 weights from LDS), used 6 MFMAs later
//   2  mode 1 + two plain VALU (v_cvt_pk_bf16_f32 of running values) per gap
//   3  mode 1 + four VALU per gap
//   4  mode 1 + six VALU per gap
// Reports: cycles per MFMA by s_memtime inside wave 0 of block 0, and the chip-wide rate by wall clock (= the clock the
// chip sustains under that load).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define FENCE() __builtin_amdgcn_sched_barrier(0)

template <int MODE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
    stream_kernel(int tiles, float *sink, long long *ticks, const float *seed) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 32768 / 16; i += 256) reinterpret_cast<uint4 *>(smem)[i] = make_uint4(i, i + 1, i + 2, i + 3);
  __syncthreads();
  const unsigned char *wl = smem + lane * 16;
  bf16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) a[i][e] = (__bf16)seed[(lane + e + i) & 63], b[i][e] = (__bf16)seed[(lane + 2 * e + i) & 63];
  f32x16 acc[2] = {{0}, {0}};
  bf16x8 ring[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) ring[i] = *reinterpret_cast<const bf16x8 *>(wl + 1024 * i);
  float v[8], v1 = seed[(lane + 7) & 63];  // eight independent filler chains: no VALU waits for the one before it
  unsigned pk[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = seed[(lane + i) & 63], pk[i] = 0;
  const long long t0 = (long long)__builtin_amdgcn_s_memtime();
  for (int t = 0; t < tiles; ++t) {
#pragma unroll
    for (int m = 0; m < 288; ++m) {
      bf16x8 x = a[m & 3], w = b[(m >> 1) & 3];
      if (MODE >= 1 && MODE <= 4) {
        w = ring[m % 6];
        ring[m % 6] = *reinterpret_cast<const bf16x8 *>(wl + 1024 * ((m * 7 + t) & 31));
      }
      acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, w, acc[m & 1], 0, 0, 0);
      constexpr int NV = MODE == 2 ? 2 : MODE == 3 ? 4 : MODE == 4 ? 6 : 0;
#pragma unroll
      for (int k = 0; k < NV; k += 2) {  // two VALU per round: a subtract, and a packed conversion of LAST round's value
        constexpr int c = 0;
        const int ch = (m * 3 + k / 2) & 7, prev = (ch + 5) & 7;
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        bf2 p;
        p[0] = (__bf16)v[prev];
        p[1] = (__bf16)v1;
        pk[prev] = __builtin_bit_cast(unsigned, p);
        v[ch] = v[ch] - v1;
        (void)c;
      }
      FENCE();
    }
  }
  const long long t1 = (long long)__builtin_amdgcn_s_memtime();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i] + (float)pk[i];
#pragma unroll
  for (int r = 0; r < 16; ++r) s += acc[0][r] + acc[1][r];
  sink[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}

template <int MODE>
static void run(int tiles, float *sink, long long *ticks, const float *seed) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(stream_kernel<MODE>, dim3(256), dim3(256), 32768, 0, tiles / 8, sink, ticks, seed);  // warm-up
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(stream_kernel<MODE>, dim3(256), dim3(256), 32768, 0, tiles, sink, ticks, seed);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  long long t = 0;
  hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
  const double n = (double)tiles * 288;
  const double pf = n * 1024 * 32768.0 / (ms * 1e-3) / 1e15;  // 1024 waves x 32768 FLOP per MFMA
  printf("mode %d: %7.1f ticks/MFMA (s_memtime, wave 0), %.3f ms, %.3f PFLOP/s chip-wide = %.1f ns per MFMA per SIMD "
         "(32 cycles at %.2f GHz)\n", MODE, t / n, ms, pf, ms * 1e6 / n, 32.0 / (ms * 1e6 / n));
}

int main(int argc, char **argv) {
  const int tiles = argc > 1 ? atoi(argv[1]) : 2000;
  float *sink, *seed, h[64];
  long long *ticks;
  for (int i = 0; i < 64; ++i) h[i] = 0.001f * (float)(i - 31);
  hipMalloc(&sink, 256 * 256 * 4);
  hipMalloc(&seed, 256);
  hipMalloc(&ticks, 8);
  hipMemcpy(seed, h, 256, hipMemcpyHostToDevice);
  run<0>(tiles, sink, ticks, seed);
  run<1>(tiles, sink, ticks, seed);
  run<2>(tiles, sink, ticks, seed);
  run<3>(tiles, sink, ticks, seed);
  run<4>(tiles, sink, ticks, seed);
  return 0;
}
