// which way does DPP row_ror rotate?  prints, for lane i of row 0, the source lane of row_ror:4 / row_ror:12 / row_shr:1
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* out) {
  int l = threadIdx.x;
  out[l] = __builtin_amdgcn_update_dpp(-1, l, 0x124, 0xf, 0xf, true);
  out[64 + l] = __builtin_amdgcn_update_dpp(-1, l, 0x12C, 0xf, 0xf, true);
  out[128 + l] = __builtin_amdgcn_update_dpp(-1, l, 0x111, 0xf, 0xf, false);
}
int main() {
  int* d; hipMalloc(&d, 192 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  int h[192]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("row_ror:4  lane<-src:"); for (int i = 0; i < 16; i++) printf(" %d<-%d", i, h[i]); printf("\n");
  printf("row_ror:12 lane<-src:"); for (int i = 0; i < 16; i++) printf(" %d<-%d", i, h[64 + i]); printf("\n");
  printf("row_shr:1  lane<-src:"); for (int i = 0; i < 16; i++) printf(" %d<-%d", i, h[128 + i]); printf("\n");
  return 0;
}
