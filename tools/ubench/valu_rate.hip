// micro-benchmark: issue cost (cycles per wave-instruction) of the instruction classes the compositors are made of.
// One workgroup per CU-ish with W waves per SIMD; each wave runs REP x 32 instructions over 8 independent registers.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__);exit(1);}}while(0)
#define REP 512

#define OP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define OP32(S) OP8(S) OP8(S) OP8(S) OP8(S)

template <int KIND>
__global__ void __launch_bounds__(1024) k(float* out, long long* cyc, float seed) {
  float r[8];
  typedef float f2v __attribute__((ext_vector_type(2)));
  f2v p[8], pseed = {seed, seed * 0.5f};
  for (int i = 0; i < 8; i++) p[i] = f2v{seed + threadIdx.x * 0.001f + i, seed - i};
  for (int i = 0; i < 8; i++) r[i] = seed + threadIdx.x * 0.001f + i;
  __shared__ float4 lds[256];
  if (threadIdx.x < 256) lds[threadIdx.x] = make_float4(seed, seed, seed, seed);
  __syncthreads();
  float4 acc4 = make_float4(0, 0, 0, 0);
  unsigned long long q[8];
  unsigned u[8];
  for (int i = 0; i < 8; i++) { q[i] = threadIdx.x + i; u[i] = threadIdx.x * 3 + i; }
  const unsigned iseed = (unsigned)(seed * 1000.f) + threadIdx.x, iseed2 = 40u;
  const unsigned lane_addr4 = (threadIdx.x & 63) * 4u, row_addr4 = (threadIdx.x & 15) * 4u, perm_addr = ((threadIdx.x ^ 4) & 63) * 4u;
  const unsigned rec_addr4 = ((threadIdx.x >> 4) & 3) * 40u + (threadIdx.x & 15) * 4u;
  unsigned lane_addr = (threadIdx.x & 63) * 16u, bc_addr = (threadIdx.x >> 6) * 16u;
  long long t0 = clock64();
  for (int it = 0; it < REP; it++) {
    if (KIND == 0) {
#define S(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(r[i]) : "v"(seed));
      OP32(S)
#undef S
    } else if (KIND == 1) {
#define S(i) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r[i]));
      OP32(S)
#undef S
    } else if (KIND == 2) {
#define S(i) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r[i]));
      OP32(S)
#undef S
    } else if (KIND == 3) {
#define S(i) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r[i]));
      OP32(S)
#undef S
    } else if (KIND == 4) {
#define S(i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
      OP32(S)
#undef S
    } else if (KIND == 5) {
#define S(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
      OP32(S)
#undef S
    } else if (KIND == 6) {
#define S(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(seed) : );
      OP32(S)
#undef S
    } else if (KIND == 7) {
#define S(i) asm volatile("ds_swizzle_b32 %0, %0 offset:swizzle(SWAP,16)\n s_waitcnt lgkmcnt(0)" : "+v"(r[i]));
      OP32(S)
#undef S
    } else if (KIND == 8) {
#define S(i) asm volatile("s_nop 1\n v_permlane32_swap_b32 %0, %1" : "+v"(r[i]), "+v"(r[(i + 1) & 7]));
      OP32(S)
#undef S
    } else if (KIND == 9) {   // broadcast LDS read, 16 B (all lanes one address)
#define S(i) asm volatile("ds_read_b128 %0, %1 offset:" #i "*16" : "=v"(acc4) : "v"(bc_addr)); 
      OP32(S)
#undef S
      asm volatile("s_waitcnt lgkmcnt(0)");
    } else if (KIND == 10) {  // per-lane LDS read, 16 B
#define S(i) asm volatile("ds_read_b128 %0, %1" : "=v"(acc4) : "v"(lane_addr));
      OP32(S)
#undef S
      asm volatile("s_waitcnt lgkmcnt(0)");
    } else if (KIND == 11) {  // dependent chain: fma -> dpp consumer (the hazard the butterfly is full of)
#define S(i) asm volatile("v_fma_f32 %0, %0, %1, %0\n v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r[i]) : "v"(seed));
      OP8(S) OP8(S)
#undef S
    } else if (KIND == 12) {  // v_mul
#define S(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(seed));
      OP32(S)
#undef S
    } else if (KIND == 14) {  // packed f32 fma (two floats per lane in a register pair)
#define S(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(pseed));
      OP32(S)
#undef S
    } else if (KIND == 15) {
#define S(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pseed));
      OP32(S)
#undef S
    } else if (KIND == 16) {
#define S(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pseed));
      OP32(S)
#undef S
    } else if (KIND == 17) {  // 32 x 32 -> 64-bit multiply-add (the compositor's record address)
#define S(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(iseed), "v"(iseed2) : "vcc");
      OP32(S)
#undef S
    } else if (KIND == 18) {
#define S(i) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(u[i]) : "v"(iseed));
      OP32(S)
#undef S
    } else if (KIND == 19) {
#define S(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(iseed));
      OP32(S)
#undef S
    } else if (KIND == 20) {  // LDS float atomic add, no return, every lane its own address
#define S(i) asm volatile("ds_add_f32 %0, %1 offset:" #i "*256" : : "v"(lane_addr4), "v"(seed) : "memory");
      OP32(S)
#undef S
      asm volatile("s_waitcnt lgkmcnt(0)");
    } else if (KIND == 21) {  // LDS float atomic add, the four 16-lane rows hit the SAME sixteen addresses
#define S(i) asm volatile("ds_add_f32 %0, %1 offset:" #i "*256" : : "v"(row_addr4), "v"(seed) : "memory");
      OP32(S)
#undef S
      asm volatile("s_waitcnt lgkmcnt(0)");
    } else if (KIND == 22) {
#define S(i) asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(r[i]) : "v"(perm_addr));
      OP32(S)
#undef S
    } else if (KIND == 23) {
#define S(i) asm volatile("s_nop 1\n v_permlane16_swap_b32 %0, %1" : "+v"(r[i]), "+v"(r[(i + 1) & 7]));
      OP32(S)
#undef S
    } else if (KIND == 24) {  // LDS float atomic add, 10 of every 16 lanes active (a gradient record), rows on different records
      if ((threadIdx.x & 15) < 10) {
#define S(i) asm volatile("ds_add_f32 %0, %1 offset:" #i "*256" : : "v"(rec_addr4), "v"(seed) : "memory");
      OP32(S)
#undef S
      }
      asm volatile("s_waitcnt lgkmcnt(0)");
    } else if (KIND == 13) {  // cndmask with an SGPR-pair mask (VOP3)
      unsigned long long m = 0x5555555555555555ull;
#define S(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r[i]) : "v"(seed), "s"(m));
      OP32(S)
#undef S
    }
  }
  long long t1 = clock64();
  float s = acc4.x + acc4.y;
  for (int i = 0; i < 8; i++) s += r[i] + p[i].x + p[i].y + (float)q[i] + (float)u[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name, float* out, long long* cyc) {
  for (int wps : {1, 2, 4}) {   // waves per SIMD (one WG of wps*256 lanes per CU; 256 WGs)
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256 * wps), 0, 0, out, cyc, 1.0001f);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256 * wps), 0, 0, out, cyc, 1.0001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[256]; CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    double avg = 0; for (int i = 0; i < 256; i++) avg += h[i]; avg /= 256;
    const double ninst = (double)REP * 32 * wps;     // wave-instructions per SIMD
    printf("%-34s waves/SIMD %d: clock64 %.2f ticks/inst/SIMD, wall %.2f ns/inst/SIMD\n", name, wps, avg / ninst, ms * 1e6 / ninst);
  }
}

int main() {
  float* out; long long* cyc;
  CK(hipMalloc(&out, 256 * 1024 * 4)); CK(hipMalloc(&cyc, 256 * 8));
  run<0>("v_fma_f32", out, cyc);
  run<12>("v_mul_f32", out, cyc);
  run<1>("v_add_f32_dpp quad_perm", out, cyc);
  run<2>("v_add_f32_dpp row_ror", out, cyc);
  run<3>("v_mov_b32_dpp quad_perm", out, cyc);
  run<4>("v_exp_f32", out, cyc);
  run<5>("v_rcp_f32", out, cyc);
  run<6>("v_cndmask_b32 vcc", out, cyc);
  run<13>("v_cndmask_b32 sgpr mask", out, cyc);
  run<7>("ds_swizzle + wait", out, cyc);
  run<8>("s_nop1 + v_permlane32_swap", out, cyc);
  run<9>("ds_read_b128 broadcast", out, cyc);
  run<10>("ds_read_b128 per-lane", out, cyc);
  run<11>("fma -> dependent dpp (pairs)", out, cyc);
  run<17>("v_mad_u64_u32", out, cyc);
  run<18>("v_lshl_add_u32", out, cyc);
  run<19>("v_mul_lo_u32", out, cyc);
  run<20>("ds_add_f32 distinct addresses", out, cyc);
  run<21>("ds_add_f32 4 rows same addresses", out, cyc);
  run<24>("ds_add_f32 10/16 lanes, 4 records", out, cyc);
  run<22>("ds_bpermute + wait", out, cyc);
  run<23>("s_nop1 + v_permlane16_swap", out, cyc);
  run<14>("v_pk_fma_f32", out, cyc);
  run<15>("v_pk_mul_f32", out, cyc);
  run<16>("v_pk_add_f32", out, cyc);
  return 0;
}
