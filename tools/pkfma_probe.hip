// Per-wave issue cost of v_fmac_f32 vs v_pk_fma_f32 at 1/2/4/8 waves per SIMD (probe).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define R8(X) X X X X X X X X
#define FM(A) "v_fmac_f32 v" #A ", v4, v9\n"
#define PK(A,B) "v_pk_fma_f32 v[" #A ":" #B "], v[4:5], v[8:9], v[" #A ":" #B "] op_sel_hi:[0,1,1]\n"
#define FM16 FM(16) FM(17) FM(18) FM(19) FM(20) FM(21) FM(22) FM(23) FM(24) FM(25) FM(26) FM(27) FM(28) FM(29) FM(30) FM(31)
#define PK8 PK(16,17) PK(18,19) PK(20,21) PK(22,23) PK(24,25) PK(26,27) PK(28,29) PK(30,31)
#define CLOB "v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31"
template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, int iters) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) asm volatile(R8(R8(FM16)) ::: CLOB);       // 1024 v_fmac
    else asm volatile(R8(R8(PK8)) ::: CLOB);                  // 512 v_pk_fma = 1024 FMAs
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x % 64 == 0) out[blockIdx.x * 16 + threadIdx.x / 64] = t1 - t0;
}
int main() {
  unsigned long long* d; CK(hipMalloc(&d, 8 * 16 * 4096));
  const int iters = 300;
  for (int cfg = 0; cfg < 4; ++cfg) {
    const int threads[] = {256, 512, 1024, 1024}; const int blocks[] = {256, 256, 256, 512};
    for (int mode = 0; mode < 2; ++mode) {
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks[cfg]), dim3(threads[cfg]), 0, 0, d, iters);
      else hipLaunchKernelGGL(k<1>, dim3(blocks[cfg]), dim3(threads[cfg]), 0, 0, d, iters);
      CK(hipGetLastError()); CK(hipDeviceSynchronize());
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0, 0));
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks[cfg]), dim3(threads[cfg]), 0, 0, d, iters);
      else hipLaunchKernelGGL(k<1>, dim3(blocks[cfg]), dim3(threads[cfg]), 0, 0, d, iters);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned long long h[16]; CK(hipMemcpy(h, d, 128, hipMemcpyDeviceToHost));
      const double per = (double)h[0] / ((mode == 0 ? 1024.0 : 512.0) * iters);
      {
        unsigned long long hh[16 * 512]; CK(hipMemcpy(hh, d, 8 * 16 * blocks[cfg], hipMemcpyDeviceToHost));
        unsigned long long mx = 0, mn = ~0ull; const int wpb = threads[cfg] / 64;
        for (int b = 0; b < blocks[cfg]; ++b) for (int w = 0; w < wpb; ++w) { auto v = hh[b * 16 + w]; mx = v > mx ? v : mx; mn = v < mn ? v : mn; }
        const double instr_per_simd = (double)blocks[cfg] * wpb / 1024.0 * (mode == 0 ? 1024.0 : 512.0) * iters;
        printf("  wall %.3f ms; ticks min %llu max %llu -> %.0f MHz if ticks=cycles; wall-derived %.3f ns per instr per SIMD\n", ms, mn, mx,
               mx / (ms * 1e3), ms * 1e6 / instr_per_simd);
      }
      printf("waves/SIMD %d  %-12s %.3f cycles per instruction per wave = %.3f cycles per FMA pair-lane\n",
             threads[cfg] * blocks[cfg] / 256 / 256, mode == 0 ? "v_fmac_f32" : "v_pk_fma_f32", per, mode == 0 ? per * 2 : per);
    }
  }
  return 0;
}
