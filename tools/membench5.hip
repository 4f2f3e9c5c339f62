// Round 4: why does a 4:1 read:write stream stop at 5.85 TB/s when the 1:1 copy with ONE 16-byte load per lane
// reaches 6.27 and a read-only stream 6.57 (profiles/round1g_membench3.txt)?  A sweep over
//   * the shape of a wave's life: copy1-like (one load per lane, 4 -> 1 combine, one store per 4 lanes), the
//     4-loads-per-lane form, LDS-DMA tiles of 1 .. 8 KiB per wave;
//   * how many bytes the chip keeps in flight (waves per CU capped through the dynamic LDS size);
//   * the cache policy of the loads (plain / nt / sc1 / sc0 sc1) and of the stores (plain / nt / sc1 / sc0 sc1 /
//     sc0 sc1 nt);
//   * where the output lies relative to the input (byte offset of the output base).
// Every kernel moves 2 GiB in and 0.5 GiB out (or what its name says); data is constant, nothing is checked here.
//   hipcc --offload-arch=gfx950 -O3 tools/membench5.hip -o /tmp/membench5
//   /tmp/membench5 [substring-of-name [warm [iters]]]
// Profiling aid (tools/), not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

enum { P_PLAIN = 0, P_NT = 1, P_SC1 = 2, P_SC0SC1 = 3, P_SC0SC1NT = 4 };
static const char *pname[] = {"plain", "nt", "sc1", "sc0sc1", "sc0sc1nt"};

template <int P> __device__ __forceinline__ f4 ldg(const f4 *p)
{
    f4 v;
    if constexpr (P == P_PLAIN) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    else if constexpr (P == P_NT) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    else if constexpr (P == P_SC1) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    else if constexpr (P == P_SC0SC1) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int P> __device__ __forceinline__ void stg(f4 *p, const f4 &v)
{
    if constexpr (P == P_PLAIN) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    else if constexpr (P == P_NT) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    else if constexpr (P == P_SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (P == P_SC0SC1) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
}
#define WAITV0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

extern __shared__ __attribute__((aligned(16))) f4 dyn_lds[];

// ---- A: copy1 shape.  T threads, ONE load per lane; 4 -> 1 through LDS; the first T/4 lanes store.
template <int T, int LP, int SP>
__global__ __launch_bounds__(T) void a_r4w1_lds(const f4 *__restrict__ in, f4 *__restrict__ out)
{
    const int t = threadIdx.x;
    const f4 v = ldg<LP>(in + (size_t)blockIdx.x * T + t);
    WAITV0();
    dyn_lds[t] = v;
    if constexpr (T > 64) __syncthreads();
    if (t < T / 4) {
        const f4 s = dyn_lds[4 * t] + dyn_lds[4 * t + 1] + dyn_lds[4 * t + 2] + dyn_lds[4 * t + 3];
        stg<SP>(out + (size_t)blockIdx.x * (T / 4) + t, s);
    }
}
// ---- A': the same, 4 -> 1 inside a lane quad (DPP), lanes 0, 4, 8 .. store 16 bytes each
template <int T, int LP, int SP>
__global__ __launch_bounds__(T) void a_r4w1_quad(const f4 *__restrict__ in, f4 *__restrict__ out)
{
    const int t = threadIdx.x;
    f4 v = ldg<LP>(in + (size_t)blockIdx.x * T + t);
    WAITV0();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        v[k] += __shfl_xor(v[k], 1);
        v[k] += __shfl_xor(v[k], 2);
    }
    if ((t & 3) == 0) stg<SP>(out + (size_t)blockIdx.x * (T / 4) + (t >> 2), v);
}
// ---- B: four loads per lane, one store per lane (round 1's r4w1)
template <int LP, int SP>
__global__ __launch_bounds__(256) void b_r4w1_regs(const f4 *__restrict__ in, f4 *__restrict__ out)
{
    const size_t o = (size_t)blockIdx.x * 256 + threadIdx.x, i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    const f4 a = ldg<LP>(in + i), b = ldg<LP>(in + i + 256), c = ldg<LP>(in + i + 512), d = ldg<LP>(in + i + 768);
    WAITV0();
    stg<SP>(out + o, a + b + c + d);
}
// ---- C: LDS-DMA tile of NLD KiB per 64-thread workgroup (1 tile per wave), AUX = cache policy bits of the DMA
//      (0 plain, 2 nt, 16 sc1, 17 sc0 sc1); dynamic LDS size caps the waves per CU
template <int NLD, int AUX, int SP>
__global__ __launch_bounds__(64) void c_dma_tile(const f4 *__restrict__ in, f4 *__restrict__ out)
{
    const int lane = threadIdx.x;
    const f4 *src = in + (size_t)blockIdx.x * (64 * NLD) + lane;
#pragma unroll
    for (int i = 0; i < NLD; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 64 * i),
                                         (__attribute__((address_space(3))) void *)(dyn_lds + 64 * i), 16, 0, AUX);
    WAITV0();
    constexpr int NOUT = 16 * NLD;
#pragma unroll
    for (int k = 0; k < (NOUT + 63) / 64; ++k) {
        const int j = 64 * k + lane;
        if (NOUT >= 64 || j < NOUT) {
            // lane j of the output takes chunks j, j + NOUT, j + 2 NOUT, j + 3 NOUT (conflict-free reads)
            const f4 s = dyn_lds[j] + dyn_lds[j + NOUT] + dyn_lds[j + 2 * NOUT] + dyn_lds[j + 3 * NOUT];
            stg<SP>(out + (size_t)blockIdx.x * NOUT + j, s);
        }
    }
}
// ---- C2: the decimator's real staging shape: 8 KiB tile + 2 KiB halo re-read from the previous tile (10 DMAs)
template <int AUX, int SP>
__global__ __launch_bounds__(64) void c_dma_tile_halo(const f4 *__restrict__ in, f4 *__restrict__ out)
{
    const int lane = threadIdx.x;
    const size_t t = blockIdx.x;
    const f4 *src = in + (t == 0 ? 0 : t * 512 - 128) + lane;
#pragma unroll
    for (int i = 0; i < 10; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 64 * i),
                                         (__attribute__((address_space(3))) void *)(dyn_lds + 64 * i), 16, 0, AUX);
    WAITV0();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int j = 64 * k + lane;
        const f4 s = dyn_lds[128 + j] + dyn_lds[256 + j] + dyn_lds[384 + j] + dyn_lds[512 + j] + dyn_lds[j];
        stg<SP>(out + t * 128 + j, s);
    }
}
// ---- D: one stream alone
template <int T, int LP>
__global__ __launch_bounds__(T) void d_read1(const f4 *__restrict__ in, f4 *__restrict__ out)
{
    const f4 v = ldg<LP>(in + (size_t)blockIdx.x * T + threadIdx.x);
    WAITV0();
    if (v.x == 123.456f) out[0] = v;
}
template <int T, int SP>
__global__ __launch_bounds__(T) void d_write1(const f4 *__restrict__ in, f4 *__restrict__ out)
{
    const f4 v = {1.0f, 2.0f, 3.0f, (float)threadIdx.x};
    stg<SP>(out + (size_t)blockIdx.x * T + threadIdx.x, v);
}
template <int T, int LP, int SP>
__global__ __launch_bounds__(T) void d_copy1(const f4 *__restrict__ in, f4 *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * T + threadIdx.x;
    const f4 v = ldg<LP>(in + i);
    WAITV0();
    stg<SP>(out + i, v);
}
template <int SP>
__global__ __launch_bounds__(256) void d_write4(const f4 *__restrict__ in, f4 *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    const f4 v = {1.0f, 2.0f, 3.0f, (float)threadIdx.x};
    stg<SP>(out + i, v); stg<SP>(out + i + 256, v); stg<SP>(out + i + 512, v); stg<SP>(out + i + 768, v);
}

struct Case { std::string name; double bytes; std::function<void()> launch; };

int main(int argc, char **argv)
{
    const char *filter = argc > 1 ? argv[1] : "";
    const int warm = argc > 2 ? atoi(argv[2]) : 60, iters = argc > 3 ? atoi(argv[3]) : 40;
    const size_t n = (size_t)1 << 27;                    // input float4 count = 2 GiB
    const size_t slack = (size_t)64 << 20;               // room to move the output base
    char *inb, *outb;
    CK(hipMalloc(&inb, n * 16)); CK(hipMalloc(&outb, n * 16 + slack));
    CK(hipMemset(inb, 1, n * 16)); CK(hipMemset(outb, 0, n * 16 + slack));
    const f4 *in = (const f4 *)inb;
    f4 *out = (f4 *)outb;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<Case> cases;
    const double B41 = 20.0 * n, B11 = 32.0 * n, B1 = 16.0 * n;
    char nm[256];

#define ADD(NAME, BYTES, ...) cases.push_back({NAME, BYTES, [=] { __VA_ARGS__; }})
#define A_LDS(T, LP, SP) do { snprintf(nm, sizeof nm, "A r4w1 1ld/lane LDS-combine T=%d ld=%s st=%s", T, pname[LP], pname[SP]); \
        ADD(nm, B41, hipLaunchKernelGGL((a_r4w1_lds<T, LP, SP>), dim3(n / T), dim3(T), T * 16, 0, in, out)); } while (0)
#define A_QUAD(T, LP, SP) do { snprintf(nm, sizeof nm, "A' r4w1 1ld/lane quad-combine T=%d ld=%s st=%s", T, pname[LP], pname[SP]); \
        ADD(nm, B41, hipLaunchKernelGGL((a_r4w1_quad<T, LP, SP>), dim3(n / T), dim3(T), 0, 0, in, out)); } while (0)
#define B_REGS(LP, SP) do { snprintf(nm, sizeof nm, "B r4w1 4ld/lane T=256 ld=%s st=%s", pname[LP], pname[SP]); \
        ADD(nm, B41, hipLaunchKernelGGL((b_r4w1_regs<LP, SP>), dim3(n / 1024), dim3(256), 0, 0, in, out)); } while (0)
    // LDSB = dynamic LDS bytes per workgroup; waves per CU = min(32, 160 KiB / LDSB)
#define C_DMA(NLD, AUX, SP, LDSB) do { snprintf(nm, sizeof nm, "C r4w1 LDS-DMA %d KiB/wave aux=%d st=%s lds=%d (<=%d waves/CU)", NLD, AUX, pname[SP], LDSB, (160 * 1024 / (LDSB)) > 32 ? 32 : (160 * 1024 / (LDSB))); \
        ADD(nm, B41, hipLaunchKernelGGL((c_dma_tile<NLD, AUX, SP>), dim3(n / (64 * NLD)), dim3(64), LDSB, 0, in, out)); } while (0)
#define C_HALO(AUX, SP, LDSB) do { snprintf(nm, sizeof nm, "C2 r4w1 LDS-DMA 8+2 KiB halo aux=%d st=%s lds=%d (<=%d waves/CU)", AUX, pname[SP], LDSB, (160 * 1024 / (LDSB)) > 32 ? 32 : (160 * 1024 / (LDSB))); \
        ADD(nm, B41, hipLaunchKernelGGL((c_dma_tile_halo<AUX, SP>), dim3(n / 512), dim3(64), LDSB, 0, in, out)); } while (0)

    // D: single streams
#define D_READ(T, LP) do { snprintf(nm, sizeof nm, "D read-only 1ld/lane T=%d ld=%s", T, pname[LP]); \
        ADD(nm, B1, hipLaunchKernelGGL((d_read1<T, LP>), dim3(n / T), dim3(T), 0, 0, in, out)); } while (0)
#define D_WRITE(T, SP) do { snprintf(nm, sizeof nm, "D write-only 1st/lane T=%d st=%s", T, pname[SP]); \
        ADD(nm, B1, hipLaunchKernelGGL((d_write1<T, SP>), dim3(n / T), dim3(T), 0, 0, in, out)); } while (0)
#define D_WRITE4(SP) do { snprintf(nm, sizeof nm, "D write-only 4st/lane T=256 st=%s", pname[SP]); \
        ADD(nm, B1, hipLaunchKernelGGL((d_write4<SP>), dim3(n / 1024), dim3(256), 0, 0, in, out)); } while (0)
#define D_COPY(T, LP, SP) do { snprintf(nm, sizeof nm, "D copy 1ld+1st/lane T=%d ld=%s st=%s", T, pname[LP], pname[SP]); \
        ADD(nm, B11, hipLaunchKernelGGL((d_copy1<T, LP, SP>), dim3(n / T), dim3(T), 0, 0, in, out)); } while (0)

    D_READ(256, P_PLAIN); D_READ(256, P_NT); D_READ(256, P_SC1); D_READ(256, P_SC0SC1); D_READ(64, P_PLAIN); D_READ(1024, P_PLAIN);
    D_WRITE(256, P_PLAIN); D_WRITE(256, P_NT); D_WRITE(256, P_SC1); D_WRITE(256, P_SC0SC1); D_WRITE(256, P_SC0SC1NT);
    D_WRITE(64, P_NT); D_WRITE(1024, P_NT);
    D_WRITE4(P_PLAIN); D_WRITE4(P_NT); D_WRITE4(P_SC0SC1);
    D_COPY(256, P_PLAIN, P_PLAIN); D_COPY(256, P_PLAIN, P_NT); D_COPY(256, P_NT, P_NT); D_COPY(256, P_NT, P_PLAIN);
    D_COPY(256, P_PLAIN, P_SC0SC1); D_COPY(256, P_NT, P_SC0SC1); D_COPY(256, P_SC1, P_SC1); D_COPY(256, P_NT, P_SC0SC1NT);
    D_COPY(64, P_PLAIN, P_NT); D_COPY(1024, P_PLAIN, P_NT);

    A_LDS(64, P_PLAIN, P_NT); A_LDS(128, P_PLAIN, P_NT); A_LDS(256, P_PLAIN, P_NT); A_LDS(512, P_PLAIN, P_NT); A_LDS(1024, P_PLAIN, P_NT);
    A_LDS(256, P_PLAIN, P_PLAIN); A_LDS(256, P_NT, P_NT); A_LDS(256, P_NT, P_PLAIN); A_LDS(256, P_PLAIN, P_SC0SC1); A_LDS(256, P_NT, P_SC0SC1);
    A_LDS(256, P_SC1, P_SC1); A_LDS(256, P_NT, P_SC0SC1NT); A_LDS(256, P_SC0SC1, P_NT);
    A_LDS(1024, P_NT, P_NT); A_LDS(1024, P_NT, P_SC0SC1); A_LDS(64, P_NT, P_NT);
    A_QUAD(64, P_PLAIN, P_NT); A_QUAD(256, P_PLAIN, P_NT); A_QUAD(256, P_NT, P_NT);
    B_REGS(P_PLAIN, P_PLAIN); B_REGS(P_PLAIN, P_NT); B_REGS(P_NT, P_NT); B_REGS(P_NT, P_SC0SC1); B_REGS(P_PLAIN, P_SC0SC1);

    // C: tile size x cache policy x waves per CU
    C_DMA(8, 0, P_NT, 8192); C_DMA(8, 2, P_NT, 8192); C_DMA(8, 16, P_NT, 8192); C_DMA(8, 17, P_NT, 8192);
    C_DMA(8, 0, P_PLAIN, 8192); C_DMA(8, 0, P_SC0SC1, 8192); C_DMA(8, 2, P_SC0SC1, 8192); C_DMA(8, 2, P_PLAIN, 8192);
    C_DMA(8, 0, P_NT, 10240); C_DMA(8, 0, P_NT, 13312); C_DMA(8, 0, P_NT, 16384); C_DMA(8, 0, P_NT, 20480); C_DMA(8, 0, P_NT, 27000); C_DMA(8, 0, P_NT, 40960);
    C_DMA(8, 2, P_NT, 10240); C_DMA(8, 2, P_NT, 13312); C_DMA(8, 2, P_NT, 16384); C_DMA(8, 2, P_NT, 20480); C_DMA(8, 2, P_NT, 27000); C_DMA(8, 2, P_NT, 40960);
    C_DMA(4, 0, P_NT, 4096); C_DMA(4, 0, P_NT, 5120); C_DMA(4, 0, P_NT, 6656); C_DMA(4, 0, P_NT, 8192); C_DMA(4, 0, P_NT, 10240); C_DMA(4, 0, P_NT, 16384);
    C_DMA(4, 2, P_NT, 5120); C_DMA(4, 2, P_NT, 8192); C_DMA(4, 2, P_NT, 10240);
    C_DMA(2, 0, P_NT, 5120); C_DMA(2, 2, P_NT, 5120); C_DMA(2, 0, P_NT, 10240);
    C_DMA(1, 0, P_NT, 5120); C_DMA(1, 2, P_NT, 5120); C_DMA(1, 0, P_NT, 10240);
    C_HALO(0, P_NT, 10240); C_HALO(2, P_NT, 10240); C_HALO(0, P_NT, 13312); C_HALO(0, P_NT, 16384); C_HALO(0, P_NT, 20480);
    C_HALO(2, P_NT, 13312); C_HALO(2, P_NT, 20480); C_HALO(0, P_SC0SC1, 10240); C_HALO(2, P_SC0SC1, 10240); C_HALO(0, P_PLAIN, 10240);

    // where the output lies relative to the input: the A (T=256) and C (8 KiB) forms with the output base moved
    for (size_t offs : {(size_t)256, (size_t)1024, (size_t)4096, (size_t)65536, (size_t)(1 << 20), (size_t)(3 << 20) + 8192, (size_t)(32 << 20)}) {
        f4 *o2 = (f4 *)(outb + offs);
        snprintf(nm, sizeof nm, "E out+%zu  A T=256 plain/nt", offs);
        ADD(nm, B41, hipLaunchKernelGGL((a_r4w1_lds<256, P_PLAIN, P_NT>), dim3(n / 256), dim3(256), 4096, 0, in, o2));
        snprintf(nm, sizeof nm, "E out+%zu  C 8 KiB aux=0 nt", offs);
        ADD(nm, B41, hipLaunchKernelGGL((c_dma_tile<8, 0, P_NT>), dim3(n / 512), dim3(64), 8192, 0, in, o2));
    }

    printf("# membench5: %zu cases, filter '%s', %d warm-up + %d timed launches each; 2 GiB in\n", cases.size(), filter, warm, iters);
    for (auto &c : cases) {
        if (filter[0] && !strstr(c.name.c_str(), filter)) continue;
        for (int i = 0; i < warm; ++i) c.launch();
        CK(hipGetLastError());
        float best = 1e30f, ms;
        double sum = 0;
        const int R = 4;
        for (int r = 0; r < R; ++r) {
            hipEventRecord(e0);
            for (int i = 0; i < iters / R; ++i) c.launch();
            hipEventRecord(e1);
            CK(hipEventSynchronize(e1));
            hipEventElapsedTime(&ms, e0, e1);
            ms /= (iters / R);
            sum += ms;
            if (ms < best) best = ms;
        }
        ms = (float)(sum / R);
        printf("%-78s %.4f ms (best %.4f)  %5.0f GB/s  %.3f of 8 TB/s\n", c.name.c_str(), ms, best, c.bytes / (ms * 1e-3) / 1e9,
               c.bytes / (ms * 1e-3) / 8e12);
        fflush(stdout);
    }
    return 0;
}
