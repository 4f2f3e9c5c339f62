// Why do persistent waves stream the decimator's 4:1 read:write mix slower than short-lived ones?
// (round 1: 0.459 ms for one 8 KiB tile per short-lived 64-thread workgroup, 0.485-0.50 ms persistent).
// Hypothesis: s_waitcnt vmcnt counts loads and stores together in issue order, so a persistent wave that
// stores tile i and then stages tile i+1 waits for the store acknowledgements before it can see its DMA
// land.  Variants: stores deferred past the next tile's wait (DEFER), plain vs nt stores, no stores at all,
// double-buffered prefetch, multi-wave workgroups.  Every wave owns 10 KiB of LDS as in the real kernel.
// Profiling aid: hipcc --offload-arch=gfx950 -O3 tools/membench4.hip -o /tmp/membench4
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ void dma8(const f4* src, f4* lds) {
#pragma unroll
  for (int i = 0; i < 8; ++i)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 64 * i),
                                     (__attribute__((address_space(3))) void*)(lds + 64 * i), 16, 0, 0);
}
template <bool NT> __device__ __forceinline__ void st(f4 v, f4* p) {
  if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}
#define WAITV(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define WAITL() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// MODE 0: store right after the tile; 1: DEFER (store after the next tile's wait); 2: no stores
// WPG waves per workgroup, each with its own image; tiles strided: wave gw takes gw, gw + W, ...
template <int MODE, bool NT, int WPG>
__global__ __launch_bounds__(64 * WPG) void stream(const f4* __restrict__ in, f4* __restrict__ out, size_t ntiles) {
  __shared__ f4 lds_all[640 * WPG];
  const int lane = threadIdx.x & 63, ww = threadIdx.x >> 6;
  f4* lds = lds_all + 640 * ww;
  const size_t W = (size_t)gridDim.x * WPG;
  size_t t = (size_t)blockIdx.x * WPG + ww;
  if (t >= ntiles) return;
  f4 pa = {0, 0, 0, 0}, pb = {0, 0, 0, 0};
  size_t pt = 0;
  bool pend = false;
  while (true) {
    dma8(in + t * 512 + lane, lds);
    WAITV(0);
    if (MODE == 1 && pend) { st<NT>(pa, out + pt * 128 + lane); st<NT>(pb, out + pt * 128 + 64 + lane); }
    const f4 a = lds[lane] + lds[lane + 64] + lds[lane + 128] + lds[lane + 192];
    const f4 b = lds[lane + 256] + lds[lane + 320] + lds[lane + 384] + lds[lane + 448];
    if (MODE == 0) { st<NT>(a, out + t * 128 + lane); st<NT>(b, out + t * 128 + 64 + lane); }
    if (MODE == 1) { pa = a; pb = b; pt = t; pend = true; }
    if (MODE == 2) { if (a.x + b.y == 123.456f) out[0] = a; }
    WAITL();
    t += W;
    if (t >= ntiles) break;
  }
  if (MODE == 1 && pend) { st<NT>(pa, out + pt * 128 + lane); st<NT>(pb, out + pt * 128 + 64 + lane); }
}

// double-buffered: the next tile's DMA is issued before the current one is consumed; DEFER as above
template <bool DEFER, bool NT>
__global__ __launch_bounds__(64) void stream_db(const f4* __restrict__ in, f4* __restrict__ out, size_t ntiles) {
  __shared__ f4 lds[1280];
  const int lane = threadIdx.x;
  const size_t W = gridDim.x;
  size_t t = blockIdx.x;
  if (t >= ntiles) return;
  f4 pa = {0, 0, 0, 0}, pb = {0, 0, 0, 0};
  size_t pt = 0;
  bool pend = false;
  int cur = 0;
  dma8(in + t * 512 + lane, lds);
  while (true) {
    const size_t nx = t + W;
    if (nx < ntiles) { dma8(in + nx * 512 + lane, lds + 640 * (cur ^ 1)); WAITV(8); } else { WAITV(0); }
    if (DEFER && pend) { st<NT>(pa, out + pt * 128 + lane); st<NT>(pb, out + pt * 128 + 64 + lane); }
    const f4* l = lds + 640 * cur;
    const f4 a = l[lane] + l[lane + 64] + l[lane + 128] + l[lane + 192];
    const f4 b = l[lane + 256] + l[lane + 320] + l[lane + 384] + l[lane + 448];
    if (!DEFER) { st<NT>(a, out + t * 128 + lane); st<NT>(b, out + t * 128 + 64 + lane); }
    else { pa = a; pb = b; pt = t; pend = true; }
    WAITL();
    if (nx >= ntiles) break;
    t = nx; cur ^= 1;
  }
  if (DEFER && pend) { st<NT>(pa, out + pt * 128 + lane); st<NT>(pb, out + pt * 128 + 64 + lane); }
}

int main() {
  const size_t n = (size_t)1 << 27;        // float4 count = 2 GiB
  f4 *in, *out;
  CK(hipMalloc(&in, n * 16)); CK(hipMalloc(&out, n * 4));
  CK(hipMemset(in, 1, n * 16)); CK(hipMemset(out, 0, n * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t ntiles = n / 512;
  auto timeit = [&](const char* name, double bytes, auto launch) {
    for (int i = 0; i < 100; ++i) launch();
    float best = 1e9f, sum = 0;
    for (int r = 0; r < 3; ++r) {
      hipEventRecord(e0); for (int i = 0; i < 40; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 40; sum += ms; if (ms < best) best = ms;
    }
    printf("%-64s %.4f ms (best %.4f)  %.0f GB/s (%.3f of 8 TB/s)\n", name, sum / 3, best, bytes / (sum / 3 * 1e-3) / 1e9,
           bytes / (sum / 3 * 1e-3) / 8e12);
    fflush(stdout);
  };
  const double rw = 20.0 * n, ro = 16.0 * n;
#define RUN(label, bytes, K, grid, block) timeit(label, bytes, [&] { hipLaunchKernelGGL(K, dim3(grid), dim3(block), 0, 0, in, out, ntiles); })
  RUN("short-lived 1 tile/wave, nt", rw, (stream<0, true, 1>), ntiles, 64);
  RUN("short-lived 1 tile/wave, plain stores", rw, (stream<0, false, 1>), ntiles, 64);
  RUN("short-lived 1 tile/wave, 2 waves/WG, nt", rw, (stream<0, true, 2>), ntiles / 2, 128);
  RUN("short-lived 1 tile/wave, 4 waves/WG, nt", rw, (stream<0, true, 4>), ntiles / 4, 256);
  RUN("short-lived 1 tile/wave, 8 waves/WG, nt", rw, (stream<0, true, 8>), ntiles / 8, 512);
  RUN("short-lived 1 tile/wave, read-only", ro, (stream<2, true, 1>), ntiles, 64);
  RUN("65536 waves x 4 tiles strided, nt", rw, (stream<0, true, 1>), 65536, 64);
  RUN("65536 waves x 4 tiles strided, plain stores", rw, (stream<0, false, 1>), 65536, 64);
  RUN("65536 waves x 4 tiles strided, DEFER nt", rw, (stream<1, true, 1>), 65536, 64);
  RUN("65536 waves x 4 tiles strided, DEFER plain", rw, (stream<1, false, 1>), 65536, 64);
  RUN("65536 waves x 4 tiles strided, read-only", ro, (stream<2, true, 1>), 65536, 64);
  RUN("131072 waves x 2 tiles strided, nt", rw, (stream<0, true, 1>), 131072, 64);
  RUN("131072 waves x 2 tiles strided, DEFER nt", rw, (stream<1, true, 1>), 131072, 64);
  RUN("4096 waves x 64 tiles (persistent), nt", rw, (stream<0, true, 1>), 4096, 64);
  RUN("4096 waves x 64 tiles (persistent), plain", rw, (stream<0, false, 1>), 4096, 64);
  RUN("4096 waves x 64 tiles (persistent), DEFER nt", rw, (stream<1, true, 1>), 4096, 64);
  RUN("4096 waves x 64 tiles (persistent), DEFER plain", rw, (stream<1, false, 1>), 4096, 64);
  RUN("4096 waves x 64 tiles (persistent), read-only", ro, (stream<2, true, 1>), 4096, 64);
  RUN("16384 WG x 4 waves x 4 tiles strided, nt", rw, (stream<0, true, 4>), 16384, 256);
  RUN("16384 WG x 4 waves x 4 tiles strided, DEFER nt", rw, (stream<1, true, 4>), 16384, 256);
  RUN("2048 waves x 128 tiles double-buffered, nt", rw, (stream_db<false, true>), 2048, 64);
  RUN("2048 waves x 128 tiles double-buffered, DEFER nt", rw, (stream_db<true, true>), 2048, 64);
  RUN("32768 waves x 8 tiles double-buffered, DEFER nt", rw, (stream_db<true, true>), 32768, 64);
  RUN("32768 waves x 8 tiles double-buffered, nt", rw, (stream_db<false, true>), 32768, 64);
  return 0;
}
