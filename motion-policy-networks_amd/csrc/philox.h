// philox.h -- counter RNG shared by the scene sampler and the batch assembler (Random123 Philox4x32-10;
// pinned to the published known-answer vectors in tests/test_scene_cloud.py via the oracle restatement).
#pragma once
#include <stdint.h>

// ---- Philox4x32-10 ------------------------------------------------------------------------------
struct Philox {
  uint32_t c[4];
};
__host__ __device__ __forceinline__ Philox philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                      uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return Philox{{c0, c1, c2, c3}};
}
// uniform in [0,1): top 24 bits
__host__ __device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-08f; }

