// Where do the workgroups of a 1200-workgroup launch (256 lanes, 26 KB of LDS: the compositors' shape) land?  Every workgroup records
// its XCC, SE, CU and start / end clocks; the host prints how many workgroups each (XCC, CU) got and whether workgroup b of XCC x
// follows a round-robin over that XCC's CUs -- the assumption behind a load-aware tile -> workgroup mapping.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <map>
#include <algorithm>

__global__ void __launch_bounds__(256) probe(uint32_t* out, int spin) {
  __shared__ float pad[30 * 256];      // 30 KB: five workgroups per CU, like the compositors (26 KB + their register budget)
  uint32_t hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const unsigned long long t0 = wall_clock64();
  float v = threadIdx.x;
  for (int i = 0; i < spin * (1 + (int)(blockIdx.x % 7)); i++) v = v * 1.0001f + 0.5f;      // uneven work
  pad[threadIdx.x] = v;
  __syncthreads();
  const unsigned long long t1 = wall_clock64();
  if (threadIdx.x == 0) {
    out[blockIdx.x * 4 + 0] = hw; out[blockIdx.x * 4 + 1] = xcc;
    out[blockIdx.x * 4 + 2] = (uint32_t)t0; out[blockIdx.x * 4 + 3] = (uint32_t)(t1 - t0) + (pad[1] == 12345.f);
  }
}

int main() {
  const int G = 1200;
  uint32_t* d; hipMalloc(&d, G * 16);
  std::vector<uint32_t> h(G * 4);
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(probe, dim3(G), dim3(256), 0, 0, d, 2000);
    hipDeviceSynchronize();
  }
  hipMemcpy(h.data(), d, G * 16, hipMemcpyDeviceToHost);
  std::map<std::pair<int, int>, std::vector<int>> by_cu;     // (xcc, se << 8 | cu) -> workgroups
  int xcc_rr = 0;
  for (int b = 0; b < G; b++) {
    const uint32_t hw = h[b * 4], xcc = h[b * 4 + 1] & 0xf;
    const int cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;        // gfx9 HW_ID layout: cu_id [11:8], sh_id [12], se_id [15:13]
    by_cu[{(int)xcc, se << 8 | sh << 4 | cu}].push_back(b);
    if ((int)xcc == b % 8) xcc_rr++;
  }
  printf("workgroups on XCC (b %% 8): %d of %d;  distinct (XCC, SE, CU): %zu\n", xcc_rr, G, by_cu.size());
  std::map<int, int> hist;
  for (auto& kv : by_cu) hist[(int)kv.second.size()]++;
  for (auto& kv : hist) printf("  CUs with %d workgroups: %d\n", kv.first, kv.second);
  // within XCC 0: index i = b / 8 of the workgroups of each CU
  int shown = 0;
  for (auto& kv : by_cu)
    if (kv.first.first == 0 && shown++ < 40) {
      printf("  XCC 0 cu %03x:", kv.first.second);
      for (int b : kv.second) printf(" %d", b / 8);
      printf("\n");
    }
  return 0;
}
