// What costs the staging side? Variants of a 4:1 LDS-DMA stream (profiling aid).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// NLOAD DMA instructions per 8 KiB tile (8 = no halo, 9/10 = halo re-stage), STORE: 0 contiguous 1 KiB per
// instruction, 1 = 16 B per lane at a 32-B lane stride (two instructions fill 2 KiB), 2 = no stores at all
template <int NLOAD, int STORE>
__global__ __launch_bounds__(64) void dma_k(const f4* __restrict__ in, f4* __restrict__ out, int ntiles, int nwaves) {
  __shared__ f4 lds[640];
  const int lane = threadIdx.x;
  const int base = ntiles / nwaves, extra = ntiles % nwaves, w = blockIdx.x;
  const int t0 = w * base + (w < extra ? w : extra), t1 = t0 + base + (w < extra ? 1 : 0);
  f4 acc = {0, 0, 0, 0};
  for (int t = t0; t < t1; ++t) {
    const f4* src = in + (size_t)t * 512 + lane - (NLOAD > 8 ? 64 : 0);
    if (t == 0 && NLOAD > 8) src += 64;
#pragma unroll
    for (int i = 0; i < NLOAD; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 64 * i),
                                       (__attribute__((address_space(3))) void*)(lds + 64 * i), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    f4 a = lds[lane] + lds[lane + 64] + lds[lane + 128] + lds[lane + 192];
    f4 b = lds[lane + 256] + lds[lane + 320] + lds[lane + 384] + lds[lane + 448];
    if (STORE == 0) { out[(size_t)t * 128 + lane] = a; out[(size_t)t * 128 + 64 + lane] = b; }
    else if (STORE == 1) { out[(size_t)t * 128 + 2 * lane] = a; out[(size_t)t * 128 + 2 * lane + 1] = b; }
    else if (STORE == 3) { __builtin_nontemporal_store(a, &out[(size_t)t * 128 + lane]); __builtin_nontemporal_store(b, &out[(size_t)t * 128 + 64 + lane]); }
    else acc += a + b;
  }
  if (STORE == 2 && acc.x == 12345.f) out[0] = acc;
}
int main() {
  const size_t nin = (size_t)1 << 27;
  f4 *in, *out;
  CK(hipMalloc(&in, nin * 16 + (1 << 20))); CK(hipMalloc(&out, nin * 4 + (1 << 20)));
  CK(hipMemset(in, 1, nin * 16)); CK(hipMemset(out, 0, nin * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int ntiles = (int)(nin / 512);
  auto timeit = [&](const char* name, double bytes, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0); for (int i = 0; i < 10; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("%-36s %.4f ms  %.0f GB/s\n", name, ms, bytes / (ms * 1e-3) / 1e9);
  };
  const double rw = 1.25 * nin * 16, ro = 1.0 * nin * 16;
  for (int wpc : {16, 64}) {
    const int nw = 256 * wpc;
    char nm[80];
    snprintf(nm, 80, "8 loads, contiguous stores %dw/CU", wpc); timeit(nm, rw, [&] { hipLaunchKernelGGL((dma_k<8, 0>), dim3(nw), dim3(64), 0, 0, in, out, ntiles, nw); });
    snprintf(nm, 80, "8 loads, strided stores    %dw/CU", wpc); timeit(nm, rw, [&] { hipLaunchKernelGGL((dma_k<8, 1>), dim3(nw), dim3(64), 0, 0, in, out, ntiles, nw); });
    snprintf(nm, 80, "8 loads, nt stores         %dw/CU", wpc); timeit(nm, rw, [&] { hipLaunchKernelGGL((dma_k<8, 3>), dim3(nw), dim3(64), 0, 0, in, out, ntiles, nw); });
    snprintf(nm, 80, "8 loads, no stores         %dw/CU", wpc); timeit(nm, ro, [&] { hipLaunchKernelGGL((dma_k<8, 2>), dim3(nw), dim3(64), 0, 0, in, out, ntiles, nw); });
    snprintf(nm, 80, "9 loads, contiguous stores %dw/CU", wpc); timeit(nm, rw, [&] { hipLaunchKernelGGL((dma_k<9, 0>), dim3(nw), dim3(64), 0, 0, in, out, ntiles, nw); });
    snprintf(nm, 80, "10 loads, contiguous stores %dw/CU", wpc); timeit(nm, rw, [&] { hipLaunchKernelGGL((dma_k<10, 0>), dim3(nw), dim3(64), 0, 0, in, out, ntiles, nw); });
    snprintf(nm, 80, "10 loads, strided stores   %dw/CU", wpc); timeit(nm, rw, [&] { hipLaunchKernelGGL((dma_k<10, 1>), dim3(nw), dim3(64), 0, 0, in, out, ntiles, nw); });
  }
  return 0;
}
