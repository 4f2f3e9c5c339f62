// Calibration of the HBM streaming ceiling for a 4:1 read:write mix on MI355X (profiling aid).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void copy_k(const f4* __restrict__ in, f4* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
// read 4 float4, write 1 (same ratio as decimate-by-4 CF32)
template <bool NT>
__global__ __launch_bounds__(256) void r4w1_k(const f4* __restrict__ in, f4* __restrict__ out, size_t nout) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nout; i += (size_t)gridDim.x * 256) {
    const size_t blk = i / 64, l = i % 64;
    const f4* p = in + blk * 256 + l;
    f4 a, b, c, d;
    if (NT) { a = __builtin_nontemporal_load(p); b = __builtin_nontemporal_load(p + 64); c = __builtin_nontemporal_load(p + 128); d = __builtin_nontemporal_load(p + 192); }
    else { a = p[0]; b = p[64]; c = p[128]; d = p[192]; }
    out[i] = a + b + c + d;
  }
}
// LDS-DMA staging: each wave stages 8 KiB (8 x 1 KiB), then reads LDS and writes 2 KiB
template <int AUX>
__global__ __launch_bounds__(64) void dma_r4w1_k(const f4* __restrict__ in, f4* __restrict__ out, size_t ntiles, int tiles_per_wave) {
  __shared__ f4 lds[512];
  const int lane = threadIdx.x;
  size_t t0 = (size_t)blockIdx.x * tiles_per_wave;
  for (size_t t = t0; t < t0 + tiles_per_wave && t < ntiles; ++t) {
    const f4* src = in + t * 512 + lane;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 64 * i),
                                       (__attribute__((address_space(3))) void*)(lds + 64 * i), 16, 0, AUX);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    f4 a = lds[lane] + lds[lane + 64] + lds[lane + 128] + lds[lane + 192];
    f4 b = lds[lane + 256] + lds[lane + 320] + lds[lane + 384] + lds[lane + 448];
    out[t * 128 + lane] = a;
    out[t * 128 + 64 + lane] = b;
  }
}
int main() {
  const size_t nin = (size_t)1 << 27;  // float4 count = 2 GiB
  f4 *in, *out;
  CK(hipMalloc(&in, nin * 16)); CK(hipMalloc(&out, nin * 16));
  CK(hipMemset(in, 1, nin * 16)); CK(hipMemset(out, 0, nin * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, double bytes, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0); for (int i = 0; i < 10; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("%-28s %.4f ms  %.0f GB/s\n", name, ms, bytes / (ms * 1e-3) / 1e9);
  };
  for (int g : {2048, 8192, 32768})
    timeit(g == 2048 ? "copy 1:1 grid2048" : g == 8192 ? "copy 1:1 grid8192" : "copy 1:1 grid32768", 2.0 * nin * 16, [&] { hipLaunchKernelGGL(copy_k, dim3(g), dim3(256), 0, 0, in, out, nin); });
  for (int g : {2048, 8192})
    timeit(g == 2048 ? "r4w1 plain grid2048" : "r4w1 plain grid8192", 1.25 * nin * 16, [&] { hipLaunchKernelGGL(r4w1_k<false>, dim3(g), dim3(256), 0, 0, in, out, nin / 4); });
  timeit("r4w1 nt grid8192", 1.25 * nin * 16, [&] { hipLaunchKernelGGL(r4w1_k<true>, dim3(8192), dim3(256), 0, 0, in, out, nin / 4); });
  const size_t ntiles = nin / 512;
  for (int wpc : {16, 32, 64}) {
    int waves = 256 * wpc; int tpw = (int)((ntiles + waves - 1) / waves);
    char nm[64]; snprintf(nm, 64, "dma r4w1 aux0 %dw/CU", wpc);
    timeit(nm, 1.25 * nin * 16, [&] { hipLaunchKernelGGL(dma_r4w1_k<0>, dim3(waves), dim3(64), 0, 0, in, out, ntiles, tpw); });
    snprintf(nm, 64, "dma r4w1 nt   %dw/CU", wpc);
    timeit(nm, 1.25 * nin * 16, [&] { hipLaunchKernelGGL(dma_r4w1_k<2>, dim3(waves), dim3(64), 0, 0, in, out, ntiles, tpw); });
  }
  return 0;
}
