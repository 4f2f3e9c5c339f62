// What does the chip sustain on the FIR's arithmetic alone, and does the operand source matter?
// 4 waves per SIMD on every CU run loops of 512 packed FMAs per "tile" on random data, as the /4 tile kernel
// does, in four forms: taps in VGPR pairs (3 VGPR operands per v_pk_fma_f32, the shipped form), taps in
// SGPR pairs (2 VGPR operands), scalar v_fmac_f32 with an SGPR tap, and both packed forms with 39 / 47 / 71
// ds_read_b128 per tile interleaved (the LDS read counts of the kernels' work splits).  Reports wall time per tile and SIMD, and the in-kernel shader clock
// (s_memtime / s_memrealtime), after a warm-up long enough for the power management to settle.
// Profiling aid: hipcc --offload-arch=gfx950 -O3 tools/valu_power_probe.hip -o /tmp/valu_power_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ void pkv_lo(f2& acc, const f2& h, const f2& x) {
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(h), "v"(x));
}
__device__ __forceinline__ void pkv_hi(f2& acc, const f2& h, const f2& x) {
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(h), "v"(x));
}
__device__ __forceinline__ void pks_lo(f2& acc, const f2& h, const f2& x) {
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "s"(h), "v"(x));
}
__device__ __forceinline__ void pks_hi(f2& acc, const f2& h, const f2& x) {
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(h), "v"(x));
}

// MODE 0: VGPR taps, 1: SGPR taps, 2: scalar fmac with SGPR taps; READS = ds_read_b128 per 512 packed FMAs (their
// values are only kept alive, no extra VALU work)
// DPPS = v_mov_b32_dpp per 512 packed FMAs (DPPK 0: wave_shl:1, 1: row_shl:1): what moving window samples between
// neighbouring lanes' registers would cost instead of re-reading them from LDS
// WRITES = ds_write_b128 per 512 packed FMAs on top (a multiple of 8); DPPK 7 = v_cvt_f32_f16 (every other one the SDWA
// form that takes the high half): the CF16 kernels' conversions
template <int MODE, int READS = 0, int DPPS = 0, int DPPK = 0, int WRITES = 0>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ taps, float* __restrict__ out, int tiles,
                                             unsigned long long* stamps) {
  __shared__ f4 lds[640 * 4];
  const int lane = threadIdx.x & 63;
  f4* img = lds + 640 * (threadIdx.x >> 6);
  for (int i = lane; i < 640; i += 64) img[i] = (f4){0.001f * i, 0.002f * lane, -0.0015f * i, 0.0007f * lane};
  __syncthreads();
  f2 hv[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) hv[k] = (f2){taps[2 * k + (lane >> 5) * 64], taps[2 * k + 1 + (lane >> 5) * 64]};
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
  f2 x[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) x[i] = (f2){0.37f + 0.011f * lane + i, -0.59f + 0.013f * lane - i};
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  const f4* win = img + (lane & 31) * 17;
  float d[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) d[i] = 0.125f * lane + i;
  for (int t = 0; t < tiles; ++t) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {          // 8 x 64 = 512 packed FMAs
      if (READS > 0) {
#pragma unroll
        for (int q = 0; q < (READS + 7) / 8; ++q) {
          if (r * ((READS + 7) / 8) + q < READS) {
            const f4 v = win[(r * 9 + q) % 60 + ((r * 9 + q) % 60) / 16];
            asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
          }
        }
      }
      if (DPPS > 0) {
#pragma unroll
        for (int q = 0; q < DPPS / 8; ++q) {
          // DPPK: which extra instruction -- 0 wave_shl:1 move, 1 row_shl:1 move, 2 v_cvt_f32_i32, 3 v_mov_b32, 4 v_pk_add_f32,
          // 5 ds_write_b128, 6 v_cndmask_b32 (price list for design decisions: time at the cap per 128 extra instructions)
          if (DPPK == 0) asm volatile("v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf" : "=v"(d[q & 7]) : "v"(d[(q + 3) & 7]));
          else if (DPPK == 1) asm volatile("v_mov_b32_dpp %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf" : "=v"(d[q & 7]) : "v"(d[(q + 3) & 7]));
          else if (DPPK == 2) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(d[q & 7]) : "v"(d[(q + 3) & 7]));
          else if (DPPK == 3) asm volatile("v_mov_b32 %0, %1" : "=v"(d[q & 7]) : "v"(d[(q + 3) & 7]));
          else if (DPPK == 4) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(acc[q & 7]) : "v"(x[q & 3]));
          else if (DPPK == 5) { if ((q & 7) == 0) img[640 - 64 + lane] = (f4){d[0], d[1], d[2], d[3]}; }
          else if (DPPK == 7) {
            if (q & 1) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(d[q & 7]) : "v"(d[(q + 3) & 7]));
            else asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(d[q & 7]) : "v"(d[(q + 3) & 7]));
          }
          else asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(d[q & 7]) : "v"(d[(q + 3) & 7]), "v"(d[(q + 5) & 7]) : );
        }
      }
      if (WRITES > 0) {
#pragma unroll
        for (int q = 0; q < WRITES / 8; ++q) img[640 - 64 + lane - 64 * (q & 1)] = (f4){d[q & 7], d[(q + 1) & 7], d[(q + 2) & 7], d[(q + 3) & 7]};
      }
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        if (MODE == 0) {
          pkv_lo(acc[(2 * k) & 7], hv[k], x[k & 3]);
          pkv_hi(acc[(2 * k + 1) & 7], hv[k], x[(k + 1) & 3]);
        } else if (MODE == 1) {
          pks_lo(acc[(2 * k) & 7], hs[k], x[k & 3]);
          pks_hi(acc[(2 * k + 1) & 7], hs[k], x[(k + 1) & 3]);
        } else {
          asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[(2 * k) & 7].x) : "s"(hs[k].x), "v"(x[k & 3].x));
          asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[(2 * k) & 7].y) : "s"(hs[k].x), "v"(x[k & 3].y));
          asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[(2 * k + 1) & 7].x) : "s"(hs[k].y), "v"(x[(k + 1) & 3].x));
          asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[(2 * k + 1) & 7].y) : "s"(hs[k].y), "v"(x[(k + 1) & 3].y));
        }
      }
    }
    // keep magnitudes bounded without changing the instruction mix much
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = acc[i] * 0.5f;
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += d[i];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s + hv[0].x;
  if (lane == 0) {
    stamps[2 * ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6))] = c1 - c0;
    stamps[2 * ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) + 1] = r1 - r0;
  }
}

// valu_power_probe                  every mix twice, time and in-kernel clock (the round 2-4 tables)
// valu_power_probe <mix> <seconds>   one mix back to back for that long (tools/price_list.py samples the board's power beside it)
int main(int argc, char** argv) {
  const int only = argc > 2 ? atoi(argv[1]) : -1;
  const double run_s = argc > 2 ? atof(argv[2]) : 0.0;
  const int blocks = 1024;                 // 4 workgroups of 4 waves per CU = 4 waves per SIMD
  float* taps; float* out; unsigned long long* stamps;
  CK(hipMalloc(&taps, 512)); CK(hipMalloc(&out, (size_t)blocks * 256 * 4)); CK(hipMalloc(&stamps, (size_t)blocks * 4 * 16));
  float h[128];
  for (int i = 0; i < 128; ++i) h[i] = 0.003f * (float)((i * 37) % 29 - 14) + 1e-4f * i;
  CK(hipMemcpy(taps, h, 512, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int tiles = 256;                   // per wave and launch: one launch = the FIR work of one bench step
  const char* names[26] = {"v_pk_fma_f32, taps in VGPR pairs", "v_pk_fma_f32, taps in SGPR pairs", "v_fmac_f32, SGPR tap",
                          "VGPR taps + 47 ds_read_b128 per tile (first-generation kernel's mix)", "SGPR taps + 47 ds_read_b128 per tile",
                          "SGPR taps + 71 ds_read_b128 per tile (shipped scalar kernel's mix)", "SGPR taps + 39 ds_read_b128 per tile",
                          "VGPR taps + 71 ds_read_b128 per tile",
                          "VGPR taps + 46 ds_read_b128 per 512 FMAs (dense /32, /8 today)",
                          "SGPR taps + 76 ds_read_b128 per 512 FMAs (16 SGPR-tap passes over the /32 tile: 38 per 256)",
                          "VGPR taps + 20 ds_read_b128 per 512 FMAs (the interpolator today: 10 per 256)",
                          "SGPR taps + 20 ds_read_b128 per 512 FMAs (the interpolator with in-lane (p, c) passes)",
                          "VGPR taps + 31 ds_read_b128 per 512 FMAs (a dense /32 kernel with 16 outputs per lane: 62 per 1024)",
                          "SGPR taps + 8 ds_read_b128 + 128 v_mov_b32_dpp wave_shl:1 per 512 FMAs (the /4 kernel sharing its window between lanes' registers)",
                          "SGPR taps + 8 ds_read_b128 + 128 v_mov_b32_dpp row_shl:1 per 512 FMAs",
                          "SGPR taps + 128 v_mov_b32_dpp wave_shl:1 per 512 FMAs, no LDS reads",
                          "SGPR taps + 24 ds_read_b128 per 512 FMAs (16 outputs per lane at /4)",
                          "SGPR taps + 128 v_cvt_f32_i32 per 512 FMAs", "SGPR taps + 128 v_mov_b32 per 512 FMAs",
                          "SGPR taps + 128 v_pk_add_f32 per 512 FMAs", "SGPR taps + 16 ds_write_b128 per 512 FMAs",
                          "SGPR taps + 128 v_cndmask_b32 per 512 FMAs",
                          "VGPR taps + 23 ds_read_b128 + 184 v_cvt_f32_f16 per 512 FMAs (CF16 /32 today: every reading lane converts)",
                          "VGPR taps + 46 ds_read_b128 + 40 v_cvt_f32_f16 + 8 ds_write_b128 per 512 FMAs (CF16 converted once on the way into a CF32 image)",
                          "VGPR taps + 46 ds_read_b128 + 40 v_cvt_f32_f16 + 16 ds_write_b128 per 512 FMAs (the same, upper bound of the staging writes: 10 needed)",
                          "VGPR taps + 23 ds_read_b128 per 512 FMAs, no conversions (what the conversions cost today)"};
  // valu_power_probe from <first>     the mixes from <first> on only (e.g. "from 22": round 5's CF16 rows)
  const int first = (argc > 2 && argv[1][0] == 'f') ? atoi(argv[2]) : 0;
  const int only_ = first ? -1 : only;
  for (int rep = 0; rep < (only_ >= 0 ? 1 : 2); ++rep)
    for (int mode = (only_ >= 0 ? only_ : first); mode < (only_ >= 0 ? only_ + 1 : 26); ++mode) {
      auto launch = [&] {
        switch (mode) {
          case 0: hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, taps, out, tiles, stamps); break;
          case 1: hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, taps, out, tiles, stamps); break;
          case 2: hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(256), 0, 0, taps, out, tiles, stamps); break;
          case 3: hipLaunchKernelGGL((probe<0, 47>), dim3(blocks), dim3(256), 0, 0, taps, out, tiles, stamps); break;
          case 4: hipLaunchKernelGGL((probe<1, 47>), dim3(blocks), dim3(256), 0, 0, taps, out, tiles, stamps); break;
          case 5: hipLaunchKernelGGL((probe<1, 71>), dim3(blocks), dim3(256), 0, 0, taps, out, tiles, stamps); break;
          case 6: hipLaunchKernelGGL((probe<1, 39>), dim3(blocks), dim3(256), 0, 0, taps, out, tiles, stamps); break;
          case 7: hipLaunchKernelGGL((probe<0, 71>), dim3(blocks), dim3(256), 0, 0, taps, out, tiles, stamps); break;
          case 8: hipLaunchKernelGGL((probe<0, 46>), dim3(blocks), dim3(256), 0, 0, taps, out, tiles, stamps); break;
          case 9: hipLaunchKernelGGL((probe<1, 76>), dim3(blocks), dim3(256), 0, 0, taps, out, tiles, stamps); break;
          case 10: hipLaunchKernelGGL((probe<0, 20>), dim3(blocks), dim3(256), 0, 0, taps, out, tiles, stamps); break;
          case 11: hipLaunchKernelGGL((probe<1, 20>), dim3(blocks), dim3(256), 0, 0, taps, out, tiles, stamps); break;
          case 12: hipLaunchKernelGGL((probe<0, 31>), dim3(blocks), dim3(256), 0, 0, taps, out, tiles, stamps); break;
          case 13: hipLaunchKernelGGL((probe<1, 8, 128, 0>), dim3(blocks), dim3(256), 0, 0, taps, out, tiles, stamps); break;
          case 14: hipLaunchKernelGGL((probe<1, 8, 128, 1>), dim3(blocks), dim3(256), 0, 0, taps, out, tiles, stamps); break;
          case 15: hipLaunchKernelGGL((probe<1, 0, 128, 0>), dim3(blocks), dim3(256), 0, 0, taps, out, tiles, stamps); break;
          case 16: hipLaunchKernelGGL((probe<1, 24>), dim3(blocks), dim3(256), 0, 0, taps, out, tiles, stamps); break;
          case 17: hipLaunchKernelGGL((probe<1, 0, 128, 2>), dim3(blocks), dim3(256), 0, 0, taps, out, tiles, stamps); break;
          case 18: hipLaunchKernelGGL((probe<1, 0, 128, 3>), dim3(blocks), dim3(256), 0, 0, taps, out, tiles, stamps); break;
          case 19: hipLaunchKernelGGL((probe<1, 0, 128, 4>), dim3(blocks), dim3(256), 0, 0, taps, out, tiles, stamps); break;
          case 20: hipLaunchKernelGGL((probe<1, 0, 128, 5>), dim3(blocks), dim3(256), 0, 0, taps, out, tiles, stamps); break;
          case 21: hipLaunchKernelGGL((probe<1, 0, 128, 6>), dim3(blocks), dim3(256), 0, 0, taps, out, tiles, stamps); break;
          case 22: hipLaunchKernelGGL((probe<0, 23, 184, 7>), dim3(blocks), dim3(256), 0, 0, taps, out, tiles, stamps); break;
          case 23: hipLaunchKernelGGL((probe<0, 46, 40, 7, 8>), dim3(blocks), dim3(256), 0, 0, taps, out, tiles, stamps); break;
          case 24: hipLaunchKernelGGL((probe<0, 46, 40, 7, 16>), dim3(blocks), dim3(256), 0, 0, taps, out, tiles, stamps); break;
          default: hipLaunchKernelGGL((probe<0, 23>), dim3(blocks), dim3(256), 0, 0, taps, out, tiles, stamps); break;
        }
      };
      for (int i = 0; i < 300; ++i) launch();          // ~100 ms: let the clocks settle on this load
      if (only_ >= 0) {                                // keep the load up while the board is sampled
        const int n = (int)(run_s / 1.2e-3);
        for (int i = 0; i < n; ++i) { launch(); if (i % 64 == 63) CK(hipStreamSynchronize(0)); }
      }
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < 50; ++i) launch();
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 50;
      std::vector<unsigned long long> st((size_t)blocks * 8);
      CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
      std::vector<double> mhz, cyc;
      for (int w = 0; w < blocks * 4; ++w) if (st[2 * w + 1]) { mhz.push_back(100.0 * st[2 * w] / st[2 * w + 1]); cyc.push_back((double)st[2 * w] / tiles); }
      std::sort(mhz.begin(), mhz.end()); std::sort(cyc.begin(), cyc.end());
      printf("%-72s %.4f ms per launch (256 tiles per wave, 4 waves/SIMD: x0.25 = one bench step's FIR) | %.0f cycles per tile per wave | in-kernel clock %.0f MHz\n",
             names[mode], ms, cyc[cyc.size() / 2], mhz[mhz.size() / 2]);
      fflush(stdout);
    }
  return 0;
}
