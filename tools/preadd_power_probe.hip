// Would a pre-added symmetric form (s = x[n-k] + x[n-127+k]; acc += h[k]*s: one v_pk_add_f32 + one v_pk_fma_f32 per
// tap PAIR instead of two v_pk_fma_f32) be cheaper at the power cap?  Same instruction count, one multiply less per
// pair.  4 waves per SIMD on every CU, 512 VALU instructions per "tile" with 71 ds_read_b128 mixed in, as the shipped
// /4 kernel has them; reports time per launch and the in-kernel clock (tools/valu_power_probe.hip's method).
// Profiling aid: hipcc --offload-arch=gfx950 -O3 tools/preadd_power_probe.hip -o /tmp/preadd_power_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ void pks_lo(f2& acc, const f2& h, const f2& x) {
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "s"(h), "v"(x));
}
__device__ __forceinline__ void pks_hi(f2& acc, const f2& h, const f2& x) {
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(h), "v"(x));
}
__device__ __forceinline__ f2 pkadd(const f2& a, const f2& b) {
  f2 r;
  asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// MODE 0: 512 packed FMAs; 1: 256 packed adds + 256 packed FMAs; 2: 256 packed FMAs only (what is left if the adds were free)
template <int MODE>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ taps, float* __restrict__ out, int tiles,
                                             unsigned long long* stamps) {
  __shared__ f4 lds[640 * 4];
  const int lane = threadIdx.x & 63;
  f4* img = lds + 640 * (threadIdx.x >> 6);
  for (int i = lane; i < 640; i += 64) img[i] = (f4){0.001f * i, 0.002f * lane, -0.0015f * i, 0.0007f * lane};
  __syncthreads();
  f2 hs[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    const int a = __builtin_amdgcn_readfirstlane(__float_as_int(taps[2 * k]));
    const int b = __builtin_amdgcn_readfirstlane(__float_as_int(taps[2 * k + 1]));
    hs[k] = (f2){__int_as_float(a), __int_as_float(b)};
  }
  f2 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (f2){0.01f * lane + i, -0.02f * lane - i};
  f2 x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = (f2){0.37f + 0.011f * lane + i, -0.59f + 0.013f * lane - i};
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  const f4* win = img + (lane & 31) * 17;
  for (int t = 0; t < tiles; ++t) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        if (r * 9 + q < 71) {
          const f4 v = win[(r * 9 + q) % 60 + ((r * 9 + q) % 60) / 16];
          asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
        }
      }
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        if (MODE == 0) {
          pks_lo(acc[(2 * k) & 7], hs[k], x[k & 7]);
          pks_hi(acc[(2 * k + 1) & 7], hs[k], x[(k + 1) & 7]);
        } else if (MODE == 1) {
          const f2 s = pkadd(x[k & 7], x[(k + 3) & 7]);
          if (k & 1) pks_hi(acc[k & 7], hs[k], s); else pks_lo(acc[k & 7], hs[k], s);
        } else {
          if (k & 1) pks_hi(acc[k & 7], hs[k], x[k & 7]); else pks_lo(acc[k & 7], hs[k], x[k & 7]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = acc[i] * 0.5f;
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
  if (lane == 0) {
    stamps[2 * ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6))] = c1 - c0;
    stamps[2 * ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) + 1] = r1 - r0;
  }
}

int main() {
  const int blocks = 1024;
  float* taps; float* out; unsigned long long* stamps;
  CK(hipMalloc(&taps, 512)); CK(hipMalloc(&out, (size_t)blocks * 256 * 4)); CK(hipMalloc(&stamps, (size_t)blocks * 4 * 16));
  float h[128];
  for (int i = 0; i < 128; ++i) h[i] = 0.003f * (float)((i * 37) % 29 - 14) + 1e-4f * i;
  CK(hipMemcpy(taps, h, 512, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int tiles = 256;
  const char* names[3] = {"512 v_pk_fma_f32 (SGPR taps) + 71 ds_read_b128 per tile", "256 v_pk_add_f32 + 256 v_pk_fma_f32 + 71 ds_read_b128 per tile",
                          "256 v_pk_fma_f32 + 71 ds_read_b128 per tile"};
  for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 3; ++mode) {
      auto launch = [&] {
        switch (mode) {
          case 0: hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, taps, out, tiles, stamps); break;
          case 1: hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, taps, out, tiles, stamps); break;
          default: hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(256), 0, 0, taps, out, tiles, stamps); break;
        }
      };
      for (int i = 0; i < 300; ++i) launch();
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < 50; ++i) launch();
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 50;
      std::vector<unsigned long long> st((size_t)blocks * 8);
      CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
      std::vector<double> mhz;
      for (int w = 0; w < blocks * 4; ++w) if (st[2 * w + 1]) mhz.push_back(100.0 * st[2 * w] / st[2 * w + 1]);
      std::sort(mhz.begin(), mhz.end());
      printf("%-66s %.4f ms per launch | in-kernel clock %.0f MHz\n", names[mode], ms, mhz[mhz.size() / 2]);
      fflush(stdout);
    }
  return 0;
}
