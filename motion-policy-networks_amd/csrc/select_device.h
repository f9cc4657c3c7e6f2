// select_device.h -- "n_out of the valid items, uniformly without replacement, in uniform order" for one workgroup:
// every item gets a 32-bit Philox key, the n_out smallest (key, index) pairs win and come out sorted.  Three-level
// radix select on the key (11 + 11 + 10 bits, histograms in LDS) finds the n_out-th smallest key, the survivors
// are collected and ordered by a counting sort in LDS.  Used by the depth-cloud draw (np.random.choice of
// run_inference.py:78-85) and by the scene sampler (np.random.choice of geometry.py:608).
#pragma once
#include <stdint.h>

constexpr int SEL_THREADS = 1024, SEL_SLACK = 64, SEL_MAX_OUT = 4096;
constexpr int SEL_HALF = SEL_MAX_OUT + SEL_SLACK, SEL_CAP = 2 * SEL_HALF;  // survivors | grouped copy

// item(g, key[4], valid[4]) describes the four consecutive items 4g .. 4g+3 (one Philox block keys four of them).
// sel: SEL_CAP u64 in LDS, hist: 2048 ints in LDS, s3: 3 ints in LDS.
// Returns the number of valid items; if it is >= n_out, sel[0..n_out) = (key << 32 | index) ascending.
template <class Item>
__device__ __forceinline__ int mpx_select_smallest(int total, int n_out, Item &&item, unsigned long long *sel,
                                                   int *hist, int *s3) {
  const int tid = threadIdx.x;
  uint32_t prefix = 0;  // key bits fixed so far
  int need = n_out;     // how many still to take from the keys matching the prefix
  const int shift[3] = {21, 10, 0}, bits[3] = {11, 11, 10};
  int total_valid = 0;
  for (int lvl = 0; lvl < 3; ++lvl) {
    for (int i = tid; i < 2048; i += SEL_THREADS) hist[i] = 0;
    __syncthreads();
    const uint32_t hi_mask = lvl == 0 ? 0u : (0xFFFFFFFFu << (shift[lvl] + bits[lvl]));
    for (int g = tid; 4 * g < total; g += SEL_THREADS) {
      uint32_t key[4];
      bool valid[4];
      item(g, key, valid);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (valid[u] && (key[u] & hi_mask) == prefix)
          atomicAdd(&hist[(key[u] >> shift[lvl]) & ((1u << bits[lvl]) - 1u)], 1);
    }
    __syncthreads();
    {
      // workgroup-wide prefix sum over the bins (two per thread): the bin where the running count reaches `need`
      const int nb = 1 << bits[lvl];
      const int b0 = 2 * tid, b1 = 2 * tid + 1;
      const int c0 = b0 < nb ? hist[b0] : 0, c1 = b1 < nb ? hist[b1] : 0;
      const int sum = c0 + c1, lane = tid & 63, wave = tid >> 6;
      int inc = sum;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
      }
      __syncthreads();  // everyone has read its bins: hist[0..15] is reused for the wave totals
      if (lane == 63) hist[wave] = inc;
      if (tid == 0) s3[0] = (int)(prefix | ((uint32_t)(nb - 1) << shift[lvl])), s3[1] = 1;
      __syncthreads();
      int base = 0, tot = 0;
#pragma unroll
      for (int w = 0; w < SEL_THREADS / 64; ++w) {
        const int v = hist[w];
        base += w < wave ? v : 0;
        tot += v;
      }
      if (lvl == 0 && tid == 0) s3[2] = tot;
      const int excl = base + inc - sum;
      if (excl < need && need <= excl + c0) {
        s3[0] = (int)(prefix | ((uint32_t)b0 << shift[lvl]));
        s3[1] = need - excl;
      } else if (excl + c0 < need && need <= excl + sum) {
        s3[0] = (int)(prefix | ((uint32_t)b1 << shift[lvl]));
        s3[1] = need - excl - c0;
      }
    }
    __syncthreads();
    prefix = (uint32_t)s3[0];
    need = s3[1];
    if (lvl == 0) total_valid = s3[2];
    __syncthreads();
  }
  if (total_valid < n_out) return total_valid;
  // prefix is now the n_out-th smallest key: take every key <= it (ties beyond `need` are cut after the sort)
  if (tid == 0) s3[2] = 0;
  __syncthreads();
  for (int g = tid; 4 * g < total; g += SEL_THREADS) {
    uint32_t key[4];
    bool valid[4];
    item(g, key, valid);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (valid[u] && key[u] <= prefix) {
        const int at = atomicAdd(&s3[2], 1);
        if (at < SEL_HALF) sel[at] = ((unsigned long long)key[u] << 32) | (uint32_t)(4 * g + u);
      }
  }
  __syncthreads();
  // Order the survivors by (key, index): they are ~n_out keys spread uniformly over [0, prefix], so a counting sort
  // on 2048 equal key ranges leaves ~2 per range; the rank inside a range is found by comparing its few members.
  const int m = min(s3[2], SEL_HALF);  // n_out + ties (n_out <= SEL_MAX_OUT: checked by the callers)
  unsigned long long *grp = sel + SEL_HALF;
  const unsigned long long span = (unsigned long long)prefix + 1ull;
  auto bucket = [&](unsigned long long e) { return (int)(((e >> 32) * 2048ull) / span); };
  for (int i = tid; i < 2048; i += SEL_THREADS) hist[i] = 0;
  __syncthreads();
  for (int i = tid; i < m; i += SEL_THREADS) atomicAdd(&hist[bucket(sel[i])], 1);
  __syncthreads();
  {
    const int c0 = hist[2 * tid], c1 = hist[2 * tid + 1], sum = c0 + c1, lane = tid & 63, wave = tid >> 6;
    int inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(inc, o);
      if (lane >= o) inc += t;
    }
    __shared__ int wsum[SEL_THREADS / 64];
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int w = 0; w < SEL_THREADS / 64; ++w) base += w < wave ? wsum[w] : 0;
    __syncthreads();
    const int excl = base + inc - sum;
    hist[2 * tid] = excl;          // start of the range; doubles as the scatter cursor
    hist[2 * tid + 1] = excl + c0;
  }
  __syncthreads();
  for (int i = tid; i < m; i += SEL_THREADS) {
    const unsigned long long e = sel[i];
    grp[atomicAdd(&hist[bucket(e)], 1)] = e;
  }
  __syncthreads();
  // hist[r] is now the END of range r (start = end of range r-1): rank each member inside its range
  for (int i = tid; i < m; i += SEL_THREADS) {
    const unsigned long long e = grp[i];
    const int r = bucket(e), end = hist[r], start = r ? hist[r - 1] : 0;
    int rank = 0;
    for (int j = start; j < end; ++j) rank += grp[j] < e ? 1 : 0;
    sel[start + rank] = e;
  }
  __syncthreads();
  return total_valid;
}
