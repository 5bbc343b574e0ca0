// Which SIMD does wave w of a 256-lane workgroup run on?  (HW_REG_HW_ID bits [5:4] = SIMD_ID on gfx9.)  If wave w -> SIMD (w + c) % 4 for
// every workgroup of a one-round launch, the compositors' per-SIMD load is a function of the tile -> workgroup table AND of which 8x8
// sub-tile each wave takes -- a second knob for the load balancing (tools/list_balance.py models it).
//   hipcc -O2 --offload-arch=gfx950 -o simd_placement.bin simd_placement.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <map>

__global__ void __launch_bounds__(256) probe(uint32_t* out, int spin) {
  __shared__ float pad[30 * 256];      // 30 KB: five workgroups per CU, like the compositors
  uint32_t hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  float v = threadIdx.x;
  for (int i = 0; i < spin; i++) v = v * 1.0001f + 0.5f;
  pad[threadIdx.x] = v;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = (hw & 0xffffu) | (xcc << 16) | ((pad[1] == 12345.f) << 31);
}

int main() {
  const int G = 1200;
  uint32_t* d; hipMalloc(&d, G * 16);
  std::vector<uint32_t> h(G * 4);
  for (int rep = 0; rep < 3; rep++) { hipLaunchKernelGGL(probe, dim3(G), dim3(256), 0, 0, d, 4000); hipDeviceSynchronize(); }
  hipMemcpy(h.data(), d, G * 16, hipMemcpyDeviceToHost);
  int distinct4 = 0, same_cu = 0;
  std::map<int, int> shift_hist, pattern;
  std::map<uint32_t, int> per_simd;      // (xcc, se, sh, cu, simd) -> waves
  for (int b = 0; b < G; b++) {
    int simd[4], cu[4];
    for (int w = 0; w < 4; w++) {
      const uint32_t v = h[b * 4 + w];
      simd[w] = (v >> 4) & 3;
      cu[w] = (int)(((v >> 8) & 0xff) | ((v >> 16) & 0xf) << 8);
      per_simd[(uint32_t)cu[w] << 2 | simd[w]]++;
    }
    const bool d4 = ((1 << simd[0]) | (1 << simd[1]) | (1 << simd[2]) | (1 << simd[3])) == 15;
    distinct4 += d4;
    same_cu += cu[0] == cu[1] && cu[1] == cu[2] && cu[2] == cu[3];
    if (d4 && simd[1] == ((simd[0] + 1) & 3) && simd[2] == ((simd[0] + 2) & 3) && simd[3] == ((simd[0] + 3) & 3)) shift_hist[simd[0]]++;
    pattern[simd[0] | simd[1] << 2 | simd[2] << 4 | simd[3] << 6]++;
  }
  printf("workgroups whose 4 waves sit on 4 distinct SIMDs: %d of %d (all on one CU: %d)\n", distinct4, G, same_cu);
  for (auto& kv : shift_hist) printf("  wave w -> SIMD (w + %d) %% 4: %d workgroups\n", kv.first, kv.second);
  printf("  distinct (SIMD of wave 0..3) patterns: %zu\n", pattern.size());
  int shown = 0;
  for (auto& kv : pattern) if (shown++ < 8) printf("    pattern %d %d %d %d: %d\n", kv.first & 3, (kv.first >> 2) & 3, (kv.first >> 4) & 3, (kv.first >> 6) & 3, kv.second);
  std::map<int, int> wh;
  for (auto& kv : per_simd) wh[kv.second]++;
  for (auto& kv : wh) printf("  SIMDs with %d waves: %d\n", kv.first, kv.second);
  // first 16 workgroups of XCC 0: SIMD of wave 0
  printf("  SIMD of wave 0, workgroups 0, 8, 16, ...:");
  for (int b = 0; b < 8 * 24; b += 8) printf(" %d", (h[b * 4] >> 4) & 3);
  printf("\n");
  return 0;
}
