// Does v_fmac_f32 issue slower when its VGPR operands share a register bank (index mod 4)? (probe)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// 16 independent accumulators so there is no dependency stall; REP blocks of 16 FMAs
#define R8(X) X X X X X X X X
#define FMA16(A0,A1,A2,A3,A4,A5,A6,A7,A8,A9,A10,A11,A12,A13,A14,A15,H,X) \
  "v_fmac_f32 v" #A0 ", v" #H ", v" #X "\n" "v_fmac_f32 v" #A1 ", v" #H ", v" #X "\n" \
  "v_fmac_f32 v" #A2 ", v" #H ", v" #X "\n" "v_fmac_f32 v" #A3 ", v" #H ", v" #X "\n" \
  "v_fmac_f32 v" #A4 ", v" #H ", v" #X "\n" "v_fmac_f32 v" #A5 ", v" #H ", v" #X "\n" \
  "v_fmac_f32 v" #A6 ", v" #H ", v" #X "\n" "v_fmac_f32 v" #A7 ", v" #H ", v" #X "\n" \
  "v_fmac_f32 v" #A8 ", v" #H ", v" #X "\n" "v_fmac_f32 v" #A9 ", v" #H ", v" #X "\n" \
  "v_fmac_f32 v" #A10 ", v" #H ", v" #X "\n" "v_fmac_f32 v" #A11 ", v" #H ", v" #X "\n" \
  "v_fmac_f32 v" #A12 ", v" #H ", v" #X "\n" "v_fmac_f32 v" #A13 ", v" #H ", v" #X "\n" \
  "v_fmac_f32 v" #A14 ", v" #H ", v" #X "\n" "v_fmac_f32 v" #A15 ", v" #H ", v" #X "\n"

template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, int iters) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0)   // acc banks all 0 (v16,v20,...), h = v4 (bank 0), x = v8 (bank 0): 3-way same bank
      asm volatile(R8(R8(FMA16(16,20,24,28,32,36,40,44,48,52,56,60,64,68,72,76,4,8))) ::: "v16","v20","v24","v28","v32","v36","v40","v44","v48","v52","v56","v60","v64","v68","v72","v76");
    else if (MODE == 1)  // acc bank 0, h bank 1 (v5), x bank 2 (v10): all different
      asm volatile(R8(R8(FMA16(16,20,24,28,32,36,40,44,48,52,56,60,64,68,72,76,5,10))) ::: "v16","v20","v24","v28","v32","v36","v40","v44","v48","v52","v56","v60","v64","v68","v72","v76");
    else if (MODE == 2)  // acc bank 0, h bank 0 (v4), x bank 2 (v10): acc/h pair
      asm volatile(R8(R8(FMA16(16,20,24,28,32,36,40,44,48,52,56,60,64,68,72,76,4,10))) ::: "v16","v20","v24","v28","v32","v36","v40","v44","v48","v52","v56","v60","v64","v68","v72","v76");
    else if (MODE == 3)  // acc bank 0, h bank 1 (v5), x bank 1 (v9): h/x pair
      asm volatile(R8(R8(FMA16(16,20,24,28,32,36,40,44,48,52,56,60,64,68,72,76,5,9))) ::: "v16","v20","v24","v28","v32","v36","v40","v44","v48","v52","v56","v60","v64","v68","v72","v76");
    else if (MODE == 4)  // consecutive accs (banks 0,1,2,3,...), h v4, x v9
      asm volatile(R8(R8(FMA16(16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,4,9))) ::: "v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31");
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x % 64 == 0) out[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
}
int main() {
  unsigned long long* d; CK(hipMalloc(&d, 8 * 4096));
  const int iters = 400; const char* names[] = {"acc,h,x same bank", "all different banks", "acc/h same bank", "h/x same bank", "consecutive accs"};
  for (int wpb : {256, 512, 1024}) {           // waves per SIMD: 1 block/CU of wpb threads -> wpb/256 waves per SIMD
    for (int mode = 0; mode < 5; ++mode) {
      dim3 g(256), b(wpb);
      switch (mode) {
        case 0: hipLaunchKernelGGL(k<0>, g, b, 0, 0, d, iters); break;
        case 1: hipLaunchKernelGGL(k<1>, g, b, 0, 0, d, iters); break;
        case 2: hipLaunchKernelGGL(k<2>, g, b, 0, 0, d, iters); break;
        case 3: hipLaunchKernelGGL(k<3>, g, b, 0, 0, d, iters); break;
        case 4: hipLaunchKernelGGL(k<4>, g, b, 0, 0, d, iters); break;
      }
      CK(hipGetLastError()); CK(hipDeviceSynchronize());
      unsigned long long h[4]; CK(hipMemcpy(h, d, 32, hipMemcpyDeviceToHost));
      printf("threads/block %3d  %-22s  %.3f cycles per v_fmac per wave\n", wpb, names[mode], (double)h[0] / (1024.0 * iters));
    }
  }
  return 0;
}
