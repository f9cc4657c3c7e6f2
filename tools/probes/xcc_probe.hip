// Which XCD does workgroup h of a grid run on?  Records HW_REG_XCC_ID per workgroup for a large grid of one-wave
// workgroups (one per unit: the dispatcher refills slots as they free up) and for a grid that fits the chip at once
// (the persistent launches), and prints how often xcc == blockIdx % 8.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/xcc_probe.hip -o build_ab/xcc_probe && build_ab/xcc_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void __launch_bounds__(64) probe(int *xcc, int spin, int vgprs_dummy) {
  const int id = (int)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15;
  if (threadIdx.x == 0) xcc[blockIdx.x] = id;
  float a = (float)vgprs_dummy;
  for (int i = 0; i < spin; ++i) a = a * 1.000001f + 0.5f;  // keep the slot busy for a while
  if (a == 12345.0f) xcc[0] = -1;
}
static void run(const char *what, int grid, int spin) {
  int *d;
  hipMalloc(&d, grid * sizeof(int));
  hipMemset(d, 0xff, grid * sizeof(int));
  hipLaunchKernelGGL(probe, dim3(grid), dim3(64), 0, 0, d, spin, 1);
  hipDeviceSynchronize();
  std::vector<int> h(grid);
  hipMemcpy(h.data(), d, grid * sizeof(int), hipMemcpyDeviceToHost);
  int match = 0, hist[16] = {0}, firstrow[16];
  for (int i = 0; i < grid; ++i) {
    match += h[i] == (i & 7);
    if (h[i] >= 0 && h[i] < 16) hist[h[i]]++;
    if (i < 16) firstrow[i] = h[i];
  }
  printf("%s: grid %d: xcc == blockIdx %% 8 for %d (%.1f %%); per-xcc counts:", what, grid, match, 100.0 * match / grid);
  for (int i = 0; i < 16; ++i) if (hist[i]) printf(" %d:%d", i, hist[i]);
  printf("; first 16 workgroups:");
  for (int i = 0; i < 16 && i < grid; ++i) printf(" %d", firstrow[i]);
  printf("\n");
  hipFree(d);
}
int main() {
  run("fits the chip (persistent)", 2048, 200000);
  run("fits the chip (persistent, 16 per CU)", 4096, 200000);
  run("one workgroup per unit", 131072, 2000);
  run("one 4-wave-sized grid", 256, 200000);
  return 0;
}
