// HBM streaming rates on MI355X with a proper warm-up (100 launches per kernel: the first ~20 launches of
// any kernel run at a transient clock): read-only, write-only, copy and the 4:1 read:write mix of a
// decimate-by-4, in "one element per thread, huge grid" and grid-stride styles (profiling aid).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void copy1(const f4* __restrict__ in, f4* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  out[i] = in[i];
}
template <bool NT>
__global__ __launch_bounds__(256) void copy4(const f4* __restrict__ in, f4* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
  const f4 a = in[i], b = in[i + 256], c = in[i + 512], d = in[i + 768];
  if (NT) { __builtin_nontemporal_store(a, out + i); __builtin_nontemporal_store(b, out + i + 256);
            __builtin_nontemporal_store(c, out + i + 512); __builtin_nontemporal_store(d, out + i + 768); }
  else { out[i] = a; out[i + 256] = b; out[i + 512] = c; out[i + 768] = d; }
}
__global__ __launch_bounds__(256) void copy_gs(const f4* __restrict__ in, f4* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
template <bool NT>
__global__ __launch_bounds__(256) void r4w1(const f4* __restrict__ in, f4* __restrict__ out) {
  const size_t o = (size_t)blockIdx.x * 256 + threadIdx.x;          // output float4
  const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
  const f4 v = in[i] + in[i + 256] + in[i + 512] + in[i + 768];
  if (NT) __builtin_nontemporal_store(v, out + o); else out[o] = v;
}
__global__ __launch_bounds__(256) void read4(const f4* __restrict__ in, f4* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
  const f4 v = in[i] + in[i + 256] + in[i + 512] + in[i + 768];
  if (v.x == 123.456f) out[0] = v;                                   // never true: keeps the loads
}
__global__ __launch_bounds__(256) void write4(f4* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
  const f4 v = {1.0f, 2.0f, 3.0f, 4.0f};
  __builtin_nontemporal_store(v, out + i); __builtin_nontemporal_store(v, out + i + 256);
  __builtin_nontemporal_store(v, out + i + 512); __builtin_nontemporal_store(v, out + i + 768);
}
// LDS-DMA staging, one 8 KiB tile per 64-thread workgroup (huge grid, short-lived waves), 2 KiB out
template <int TILES>
__global__ __launch_bounds__(64) void dma_r4w1(const f4* __restrict__ in, f4* __restrict__ out) {
  __shared__ f4 lds[512];
  const int lane = threadIdx.x;
  for (int k = 0; k < TILES; ++k) {
    const size_t t = (size_t)blockIdx.x * TILES + k;
    const f4* src = in + t * 512 + lane;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 64 * i),
                                       (__attribute__((address_space(3))) void*)(lds + 64 * i), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const f4 a = lds[lane] + lds[lane + 64] + lds[lane + 128] + lds[lane + 192];
    const f4 b = lds[lane + 256] + lds[lane + 320] + lds[lane + 384] + lds[lane + 448];
    __builtin_nontemporal_store(a, out + t * 128 + lane);
    __builtin_nontemporal_store(b, out + t * 128 + 64 + lane);
  }
}
// same, persistent: W waves, pass i covers tiles [i*W, (i+1)*W) in plain order
__global__ __launch_bounds__(64) void dma_r4w1_persist(const f4* __restrict__ in, f4* __restrict__ out, size_t ntiles) {
  __shared__ f4 lds[512];
  const int lane = threadIdx.x;
  for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const f4* src = in + t * 512 + lane;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 64 * i),
                                       (__attribute__((address_space(3))) void*)(lds + 64 * i), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const f4 a = lds[lane] + lds[lane + 64] + lds[lane + 128] + lds[lane + 192];
    const f4 b = lds[lane + 256] + lds[lane + 320] + lds[lane + 384] + lds[lane + 448];
    __builtin_nontemporal_store(a, out + t * 128 + lane);
    __builtin_nontemporal_store(b, out + t * 128 + 64 + lane);
  }
}
// persistent waves taking tiles from a global atomic queue (in request order), next index prefetched
__global__ __launch_bounds__(64) void dma_r4w1_queue(const f4* __restrict__ in, f4* __restrict__ out, size_t ntiles,
                                                     unsigned long long* counter, unsigned long long base) {
  __shared__ f4 lds[512];
  const int lane = threadIdx.x;
  unsigned long long t = 0;
  if (lane == 0) t = atomicAdd(counter, 1ull) - base;
  t = __builtin_amdgcn_readfirstlane((unsigned)t);
  while (t < ntiles) {
    const f4* src = in + t * 512 + lane;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 64 * i),
                                       (__attribute__((address_space(3))) void*)(lds + 64 * i), 16, 0, 0);
    unsigned long long tn = 0;
    if (lane == 0) tn = atomicAdd(counter, 1ull) - base;       // next tile, in flight behind the DMAs
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const f4 a = lds[lane] + lds[lane + 64] + lds[lane + 128] + lds[lane + 192];
    const f4 b = lds[lane + 256] + lds[lane + 320] + lds[lane + 384] + lds[lane + 448];
    __builtin_nontemporal_store(a, out + t * 128 + lane);
    __builtin_nontemporal_store(b, out + t * 128 + 64 + lane);
    t = __builtin_amdgcn_readfirstlane((unsigned)tn);
  }
}
// persistent waves, NQ atomic queues: a wave of class c = blockIdx % NQ takes tiles NQ*k + c in request order
template <int NQ>
__global__ __launch_bounds__(64) void dma_r4w1_queues(const f4* __restrict__ in, f4* __restrict__ out, size_t ntiles,
                                                      unsigned long long* counters, unsigned long long base) {
  __shared__ f4 lds[512];
  const int lane = threadIdx.x;
  const unsigned cls = blockIdx.x % NQ;
  unsigned long long* ctr = counters + 16 * cls;               // one 128-byte line per counter
  unsigned long long k = 0;
  if (lane == 0) k = atomicAdd(ctr, 1ull) - base;
  unsigned long long t = (unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)k) * NQ + cls;
  while (t < ntiles) {
    const f4* src = in + t * 512 + lane;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 64 * i),
                                       (__attribute__((address_space(3))) void*)(lds + 64 * i), 16, 0, 0);
    unsigned long long kn = 0;
    if (lane == 0) kn = atomicAdd(ctr, 1ull) - base;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const f4 a = lds[lane] + lds[lane + 64] + lds[lane + 128] + lds[lane + 192];
    const f4 b = lds[lane + 256] + lds[lane + 320] + lds[lane + 384] + lds[lane + 448];
    __builtin_nontemporal_store(a, out + t * 128 + lane);
    __builtin_nontemporal_store(b, out + t * 128 + 64 + lane);
    t = (unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)kn) * NQ + cls;
  }
}
int main() {
  const size_t n = (size_t)1 << 27;        // float4 count = 2 GiB
  f4 *in, *out;
  CK(hipMalloc(&in, n * 16)); CK(hipMalloc(&out, n * 16));
  CK(hipMemset(in, 1, n * 16)); CK(hipMemset(out, 0, n * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, double bytes, auto launch) {
    for (int i = 0; i < 100; ++i) launch();
    hipEventRecord(e0); for (int i = 0; i < 50; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 50;
    printf("%-44s %.4f ms  %.0f GB/s (%.3f of 8 TB/s)\n", name, ms, bytes / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 8e12);
  };
  timeit("read-only, 4 float4 per thread", 16.0 * n, [&] { hipLaunchKernelGGL(read4, dim3(n / 1024), dim3(256), 0, 0, in, out); });
  timeit("write-only nt, 4 float4 per thread", 16.0 * n, [&] { hipLaunchKernelGGL(write4, dim3(n / 1024), dim3(256), 0, 0, out); });
  timeit("copy, 1 float4 per thread", 32.0 * n, [&] { hipLaunchKernelGGL(copy1, dim3(n / 256), dim3(256), 0, 0, in, out); });
  timeit("copy, 4 float4 per thread", 32.0 * n, [&] { hipLaunchKernelGGL(copy4<false>, dim3(n / 1024), dim3(256), 0, 0, in, out); });
  timeit("copy, 4 float4 per thread, nt stores", 32.0 * n, [&] { hipLaunchKernelGGL(copy4<true>, dim3(n / 1024), dim3(256), 0, 0, in, out); });
  timeit("copy, grid-stride 32768 blocks", 32.0 * n, [&] { hipLaunchKernelGGL(copy_gs, dim3(32768), dim3(256), 0, 0, in, out, n); });
  timeit("4:1 read:write, 4 float4 in 1 out per thread", 20.0 * n, [&] { hipLaunchKernelGGL(r4w1<false>, dim3(n / 1024), dim3(256), 0, 0, in, out); });
  timeit("4:1 read:write, nt stores", 20.0 * n, [&] { hipLaunchKernelGGL(r4w1<true>, dim3(n / 1024), dim3(256), 0, 0, in, out); });
  const size_t ntiles = n / 512;
  timeit("4:1 LDS-DMA, 1 tile per 64-thread block", 20.0 * n, [&] { hipLaunchKernelGGL(dma_r4w1<1>, dim3(ntiles), dim3(64), 0, 0, in, out); });
  timeit("4:1 LDS-DMA, 4 tiles per 64-thread block", 20.0 * n, [&] { hipLaunchKernelGGL(dma_r4w1<4>, dim3(ntiles / 4), dim3(64), 0, 0, in, out); });
  timeit("4:1 LDS-DMA, 16 tiles per 64-thread block", 20.0 * n, [&] { hipLaunchKernelGGL(dma_r4w1<16>, dim3(ntiles / 16), dim3(64), 0, 0, in, out); });
  timeit("4:1 LDS-DMA persistent 4096 waves, plain strided", 20.0 * n, [&] { hipLaunchKernelGGL(dma_r4w1_persist, dim3(4096), dim3(64), 0, 0, in, out, ntiles); });
  timeit("4:1 LDS-DMA persistent 65536 waves, plain strided", 20.0 * n, [&] { hipLaunchKernelGGL(dma_r4w1_persist, dim3(65536), dim3(64), 0, 0, in, out, ntiles); });
  unsigned long long* counters; CK(hipMalloc(&counters, 128 * 256)); CK(hipMemset(counters, 0, 128 * 256));
  // every class of NQ queues is used by waves/NQ waves and ends with one out-of-range grab per wave
  {
    unsigned long long base = 0;
    timeit("4:1 LDS-DMA persistent 4096 waves, 32 atomic queues", 20.0 * n, [&] { hipLaunchKernelGGL(dma_r4w1_queues<32>, dim3(4096), dim3(64), 0, 0, in, out, ntiles, counters, base); base += ntiles / 32 + 4096 / 32; });
  }
  CK(hipMemset(counters, 0, 128 * 256));
  {
    unsigned long long base = 0;
    timeit("4:1 LDS-DMA persistent 4096 waves, 64 atomic queues", 20.0 * n, [&] { hipLaunchKernelGGL(dma_r4w1_queues<64>, dim3(4096), dim3(64), 0, 0, in, out, ntiles, counters, base); base += ntiles / 64 + 4096 / 64; });
  }
  CK(hipMemset(counters, 0, 128 * 256));
  {
    unsigned long long base = 0;
    timeit("4:1 LDS-DMA persistent 4096 waves, 16 atomic queues", 20.0 * n, [&] { hipLaunchKernelGGL(dma_r4w1_queues<16>, dim3(4096), dim3(64), 0, 0, in, out, ntiles, counters, base); base += ntiles / 16 + 4096 / 16; });
  }
  return 0;
}
