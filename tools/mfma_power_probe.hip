// Would the matrix cores do the FIR's fp32 arithmetic cheaper than v_pk_fma_f32?  (They have the same fp32 peak.)
// 4 waves per SIMD on every CU run the multiply-adds of one /4 tile (512 packed FMAs = 65536 MACs per wave) as
//   0: v_pk_fma_f32 with SGPR taps (the shipped kernel's arithmetic),
//   1: v_mfma_f32_32x32x1_2b_f32 (32 per tile), 2: v_mfma_f32_16x16x1_4b_f32 (64), 3: v_mfma_f32_4x4x1_16b_f32 (256),
//   4: v_mfma_f32_32x32x2_f32 (32), 5: v_mfma_f32_16x16x4_f32 (64)
// and report time per launch and the in-kernel clock, as tools/valu_power_probe.hip does.  A FIR on the matrix cores
// is a Toeplitz product: 32 output rows waste 49 % of the MACs on zeros, 16 rows 32 %, 4 rows 9 %, so a form only
// pays if it is faster per MAC by more than that.  Also checks whether a K=1 MFMA is one IEEE fused multiply-add per
// element (the bit-exact contract would need that).
// Profiling aid: hipcc --offload-arch=gfx950 -O3 tools/mfma_power_probe.hip -o /tmp/mfma_power_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f32v __attribute__((ext_vector_type(32)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ void pks_lo(f2& acc, const f2& h, const f2& x) {
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "s"(h), "v"(x));
}
__device__ __forceinline__ void pks_hi(f2& acc, const f2& h, const f2& x) {
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(h), "v"(x));
}

template <int MODE>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ taps, float* __restrict__ out, int tiles,
                                             unsigned long long* stamps) {
  const int lane = threadIdx.x & 63;
  float a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = taps[(lane + 7 * i) & 127]; b[i] = 0.37f + 0.011f * lane + i; }
  float s = 0;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  if constexpr (MODE == 0) {
    f2 hs[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      const int u = __builtin_amdgcn_readfirstlane(__float_as_int(taps[2 * k]));
      const int v = __builtin_amdgcn_readfirstlane(__float_as_int(taps[2 * k + 1]));
      hs[k] = (f2){__int_as_float(u), __int_as_float(v)};
    }
    f2 acc[8], x[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (f2){0.01f * lane + i, -0.02f * lane - i};
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = (f2){b[i], -b[i]};
    for (int t = 0; t < tiles; ++t) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int k = 0; k < 32; ++k) {
          pks_lo(acc[(2 * k) & 7], hs[k], x[k & 3]);
          pks_hi(acc[(2 * k + 1) & 7], hs[k], x[(k + 1) & 3]);
        }
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = acc[i] * 0.5f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
  } else if constexpr (MODE == 1 || MODE == 4) {
    f32v c0v = {}, c1v = {};
    f16v d0 = {}, d1 = {}, d2 = {}, d3 = {};
    for (int t = 0; t < tiles; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if constexpr (MODE == 1) {
          c0v = __builtin_amdgcn_mfma_f32_32x32x1f32(a[r & 3], b[r & 3], c0v, 0, 0, 0);
          c1v = __builtin_amdgcn_mfma_f32_32x32x1f32(a[(r + 1) & 3], b[(r + 2) & 3], c1v, 0, 0, 0);
        } else {
          d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r & 3], b[r & 3], d0, 0, 0, 0);
          d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(r + 1) & 3], b[(r + 2) & 3], d1, 0, 0, 0);
        }
      }
      if constexpr (MODE == 1) { c0v = c0v * 0.5f; c1v = c1v * 0.5f; } else { d0 = d0 * 0.5f; d1 = d1 * 0.5f; }
    }
    if constexpr (MODE == 1) { for (int i = 0; i < 32; ++i) s += c0v[i] + c1v[i]; } else { for (int i = 0; i < 16; ++i) s += d0[i] + d1[i] + d2[i] + d3[i]; }
  } else if constexpr (MODE == 2 || MODE == 5) {
    f16v c[4] = {};
    f4 d[4] = {};
    for (int t = 0; t < tiles; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if constexpr (MODE == 2) c[q] = __builtin_amdgcn_mfma_f32_16x16x1f32(a[(r + q) & 3], b[r & 3], c[q], 0, 0, 0);
          else d[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(r + q) & 3], b[r & 3], d[q], 0, 0, 0);
        }
#pragma unroll
      for (int q = 0; q < 4; ++q) { c[q] = c[q] * 0.5f; d[q] = d[q] * 0.5f; }
    }
    for (int q = 0; q < 4; ++q) { for (int i = 0; i < 16; ++i) s += c[q][i]; s += d[q].x + d[q].y + d[q].z + d[q].w; }
  } else {
    f4 c[8] = {};
    for (int t = 0; t < tiles; ++t) {
#pragma unroll
      for (int r = 0; r < 32; ++r)
#pragma unroll
        for (int q = 0; q < 8; ++q) c[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[(r + q) & 3], b[r & 3], c[q], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 8; ++q) c[q] = c[q] * 0.5f;
    }
    for (int q = 0; q < 8; ++q) s += c[q].x + c[q].y + c[q].z + c[q].w;
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
  if (lane == 0) {
    stamps[2 * ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6))] = c1 - c0;
    stamps[2 * ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) + 1] = r1 - r0;
  }
}

// one wave: d[i][j] = mfma_4x4x1(a, b, c) for lane values given; compared with fmaf on the host
__global__ void fma_check(const float* a, const float* b, const float* c, float* d4, float* d32) {
  const int lane = threadIdx.x;
  f4 cc = {c[4 * lane], c[4 * lane + 1], c[4 * lane + 2], c[4 * lane + 3]};
  cc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[lane], b[lane], cc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) d4[4 * lane + r] = cc[r];
  f32v c32;
  for (int r = 0; r < 32; ++r) c32[r] = c[(32 * lane + r) & 255];
  c32 = __builtin_amdgcn_mfma_f32_32x32x1f32(a[lane], b[lane], c32, 0, 0, 0);
  for (int r = 0; r < 32; ++r) d32[32 * lane + r] = c32[r];
}

static int check_fma() {
  float ha[64], hb[64], hc[256], *a, *b, *c, *d4, *d32;
  unsigned seed = 12345;
  auto rnd = [&] { seed = seed * 1664525u + 1013904223u; return (float)(int)(seed >> 8) / 8388608.0f - 1.0f; };
  int bad4 = 0, bad32 = 0, total4 = 0, total32 = 0, den4 = 0, denbad4 = 0;
  CK(hipMalloc(&a, 256)); CK(hipMalloc(&b, 256)); CK(hipMalloc(&c, 1024)); CK(hipMalloc(&d4, 1024)); CK(hipMalloc(&d32, 8192));
  for (int round = 0; round < 64; ++round) {
    const float scale = round < 48 ? 1.0f : ldexpf(1.0f, -70 - round);       // later rounds: denormal products / sums
    for (int i = 0; i < 64; ++i) { ha[i] = rnd() * (round & 1 ? 1e-3f : 1.0f); hb[i] = rnd() * scale; }
    for (int i = 0; i < 256; ++i) hc[i] = rnd() * scale * (round & 2 ? 1e-4f : 1.0f);
    CK(hipMemcpy(a, ha, 256, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb, 256, hipMemcpyHostToDevice)); CK(hipMemcpy(c, hc, 1024, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(fma_check, dim3(1), dim3(64), 0, 0, a, b, c, d4, d32);
    float h4[256], h32[2048];
    CK(hipMemcpy(h4, d4, 1024, hipMemcpyDeviceToHost)); CK(hipMemcpy(h32, d32, 8192, hipMemcpyDeviceToHost));
    // 4x4x1, 16 blocks: lane l = block l/4; A row i = l%4, B column j = l%4; D[i][j] in lane (block*4 + j), register i
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 4; ++r) {
        const int blk = l / 4, j = l % 4, i = r;
        const float want = fmaf(ha[blk * 4 + i], hb[blk * 4 + j], hc[4 * l + r]);
        const bool den = std::fabs(want) < 1.17549435e-38f && want != 0.0f;
        ++total4; den4 += den;
        if (memcmp(&want, &h4[4 * l + r], 4)) { ++bad4; denbad4 += den; }
      }
    // 32x32x1, 2 blocks: lane l: A row i = l%32 of block l/32; B column j = l%32 of block l/32;
    // D: lane l holds column j = l%32; register r of block k: row i = 8*(r/4 % 4)... = (r%4) + 8*(r/4) + 4*(l/32) for r < 16 -> block 0; r >= 16 -> block 1
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 32; ++r) {
        const int blk = r / 16, rr = r % 16, j = l % 32, i = (rr % 4) + 8 * (rr / 4) + 4 * (l / 32);
        const float want = fmaf(ha[blk * 32 + i], hb[blk * 32 + j], hc[(32 * l + r) & 255]);
        ++total32;
        if (memcmp(&want, &h32[32 * l + r], 4)) ++bad32;
      }
  }
  printf("K=1 MFMA vs fmaf: 4x4x1_16b: %d of %d elements differ (%d of the %d with denormal results); 32x32x1_2b: %d of %d differ\n", bad4, total4,
         denbad4, den4, bad32, total32);
  return 0;
}

int main() {
  if (check_fma()) return 1;
  const int blocks = 1024;
  float* taps; float* out; unsigned long long* stamps;
  CK(hipMalloc(&taps, 512)); CK(hipMalloc(&out, (size_t)blocks * 256 * 4)); CK(hipMalloc(&stamps, (size_t)blocks * 4 * 16));
  float h[128];
  for (int i = 0; i < 128; ++i) h[i] = 0.003f * (float)((i * 37) % 29 - 14) + 1e-4f * i;
  CK(hipMemcpy(taps, h, 512, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int tiles = 256;
  const char* names[6] = {"v_pk_fma_f32, SGPR taps (512 per tile)", "v_mfma_f32_32x32x1_2b_f32 (32 per tile)", "v_mfma_f32_16x16x1_4b_f32 (64 per tile)",
                          "v_mfma_f32_4x4x1_16b_f32 (256 per tile)", "v_mfma_f32_32x32x2_f32 (32 per tile)", "v_mfma_f32_16x16x4_f32 (64 per tile)"};
  for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 6; ++mode) {
      auto launch = [&] {
        switch (mode) {
          case 0: hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, taps, out, tiles, stamps); break;
          case 1: hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, taps, out, tiles, stamps); break;
          case 2: hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(256), 0, 0, taps, out, tiles, stamps); break;
          case 3: hipLaunchKernelGGL(probe<3>, dim3(blocks), dim3(256), 0, 0, taps, out, tiles, stamps); break;
          case 4: hipLaunchKernelGGL(probe<4>, dim3(blocks), dim3(256), 0, 0, taps, out, tiles, stamps); break;
          default: hipLaunchKernelGGL(probe<5>, dim3(blocks), dim3(256), 0, 0, taps, out, tiles, stamps); break;
        }
      };
      for (int i = 0; i < 300; ++i) launch();
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < 50; ++i) launch();
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 50;
      std::vector<unsigned long long> st((size_t)blocks * 8);
      CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
      std::vector<double> mhz, cyc;
      for (int w = 0; w < blocks * 4; ++w) if (st[2 * w + 1]) { mhz.push_back(100.0 * st[2 * w] / st[2 * w + 1]); cyc.push_back((double)st[2 * w] / tiles); }
      std::sort(mhz.begin(), mhz.end()); std::sort(cyc.begin(), cyc.end());
      printf("%-44s %.4f ms per launch (65536 MACs x 256 tiles per wave, 4 waves/SIMD) | %.0f cycles per tile per wave | in-kernel clock %.0f MHz\n",
             names[mode], ms, cyc[cyc.size() / 2], mhz[mhz.size() / 2]);
      fflush(stdout);
    }
  return 0;
}
