// micro-benchmark: throughput of fire-and-forget global float atomics vs plain stores in the gradient-flush pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__);exit(1);}}while(0)
__global__ void k_atomic(const uint32_t* __restrict__ ids, int n, float* __restrict__ rec, int nv) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float* d = rec + (size_t)ids[i] * 12;
  for (int v = 0; v < nv; v++) atomicAdd(&d[v], 1.0f + v);
}
__global__ void k_store(int n, float* __restrict__ pair) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4* d = (float4*)(pair + (size_t)i * 12);
  d[0] = make_float4(1, 2, 3, 4); d[1] = make_float4(1, 2, 3, 4); d[2] = make_float4(1, 2, 3, 4);
}
__global__ void k_atomic_ret(const uint32_t* __restrict__ ids, int n, uint32_t* __restrict__ cnt, uint32_t* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = atomicAdd(&cnt[ids[i]], 1u);
}
int main() {
  const int P = 150000;
  for (int mode = 0; mode < 3; mode++) {   // 0: random ids, 1: locally coherent ids (i/4 + jitter), 2: same as 1 but n=385k
    int n = mode == 2 ? 385000 : 600000;
    std::vector<uint32_t> h(n);
    srand(1);
    for (int i = 0; i < n; i++) h[i] = mode == 0 ? rand() % P : (uint32_t)(((long long)i * P / n + rand() % 700) % P);
    uint32_t* ids; float* rec; float* pair; uint32_t* cnt; uint32_t* out;
    CK(hipMalloc(&ids, n * 4)); CK(hipMalloc(&rec, (size_t)P * 48)); CK(hipMalloc(&pair, (size_t)n * 48));
    CK(hipMalloc(&cnt, 1200 * 4)); CK(hipMalloc(&out, n * 4));
    CK(hipMemcpy(ids, h.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemset(rec, 0, (size_t)P * 48));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int nv : {1, 4, 7, 12}) {
      for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k_atomic, dim3((n + 255) / 256), dim3(256), 0, 0, ids, n, rec, nv);
      hipEventRecord(e0);
      for (int r = 0; r < 20; r++) hipLaunchKernelGGL(k_atomic, dim3((n + 255) / 256), dim3(256), 0, 0, ids, n, rec, nv);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("mode %d n %d atomics nv=%2d : %.2f us/launch  (%.1f atomics/ns)\n", mode, n, nv, ms * 1000 / 20, (double)n * nv / (ms * 1e6 / 20));
    }
    hipEventRecord(e0);
    for (int r = 0; r < 20; r++) hipLaunchKernelGGL(k_store, dim3((n + 255) / 256), dim3(256), 0, 0, n, pair);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d n %d plain 48B stores : %.2f us/launch\n", mode, n, ms * 1000 / 20);
    // returning u32 atomics on 1200 counters (the old scatter pattern)
    std::vector<uint32_t> t(n); for (int i = 0; i < n; i++) t[i] = mode == 0 ? rand() % 1200 : (uint32_t)((long long)i * 1200 / n);
    CK(hipMemcpy(ids, t.data(), n * 4, hipMemcpyHostToDevice));
    hipEventRecord(e0);
    for (int r = 0; r < 20; r++) hipLaunchKernelGGL(k_atomic_ret, dim3((n + 255) / 256), dim3(256), 0, 0, ids, n, cnt, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d n %d returning u32 atomics on 1200 ctrs : %.2f us/launch\n", mode, n, ms * 1000 / 20);
    hipFree(ids); hipFree(rec); hipFree(pair); hipFree(cnt); hipFree(out);
  }
  return 0;
}
