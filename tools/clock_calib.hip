// Which of the clock figures is the shader clock?  (LABBOOK.md section 5.1 "Round 3", section 7)
//
// One wave per SIMD issues N independent v_fma_f32, one every 4 cycles when alone on its SIMD
// (MI355X_MICROARCH.md, "vector-instruction ISSUE cost"), so N * 4 / (wall time) is the shader clock by
// instruction count.  The same wave brackets the loop with s_memtime (claimed: shader cycles) and
// s_memrealtime (claimed: 100 MHz).  If the three agree, s_memtime / s_memrealtime is a valid in-kernel
// clock and can be trusted inside the FIR kernels, where the instruction count says nothing.
//
//   hipcc --offload-arch=gfx950 -O3 tools/clock_calib.hip -o /tmp/clock_calib && /tmp/clock_calib
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

__global__ __launch_bounds__(64) void fma_loop(unsigned long long *rec, float *sink, int iters)
{
    float a0 = threadIdx.x, a1 = 1.0f, a2 = 2.0f, a3 = 3.0f, a4 = 4.0f, a5 = 5.0f, a6 = 6.0f, a7 = 7.0f;
    const float m = 0.999f, b = 1.0e-3f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        // 32 independent-enough FMAs per iteration (8 chains, 4 rounds), kept by the asm barrier
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\t"
                         "v_fma_f32 %3, %3, %8, %9\n\tv_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\t"
                         "v_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(m), "v"(b));
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        rec[2 * blockIdx.x] = c1 - c0;
        rec[2 * blockIdx.x + 1] = r1 - r0;
    }
    sink[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int iters = 1 << 18;                       // 2^18 * 32 FMAs = 8.4 M instructions per wave
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("%s, %d CUs; per wave %d v_fma_f32\n", prop.gcnArchName, cus, iters * 32);
    printf("%-28s %10s %12s %12s %12s\n", "waves", "wall ms", "MHz by count", "MHz memtime", "realtime MHz");
    for (int per_cu : {1, 4, 8, 16}) {               // 4 per CU = one wave per SIMD: the calibration row
        const int n = cus * per_cu;
        unsigned long long *rec;
        float *sink;
        CHECK(hipMalloc(&rec, 16 * n));
        CHECK(hipMalloc(&sink, 256 * n));
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(fma_loop, dim3(n), dim3(64), 0, 0, rec, sink, iters);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
        }
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(2 * n);
        CHECK(hipMemcpy(h.data(), rec, 16 * n, hipMemcpyDeviceToHost));
        std::vector<double> mhz(n), rt(n);
        for (int i = 0; i < n; ++i) {
            mhz[i] = 100.0 * (double)h[2 * i] / (double)h[2 * i + 1];
            rt[i] = (double)h[2 * i + 1] / (ms * 1e3);     // realtime ticks per microsecond of wall time
        }
        std::sort(mhz.begin(), mhz.end());
        std::sort(rt.begin(), rt.end());
        // waves per SIMD = per_cu / 4 (at least 1): an FMA issues every 4 cycles from one wave, every 2 from two or more
        const double per_simd = per_cu < 4 ? 1.0 : per_cu / 4.0;
        const double cyc_per_fma = per_simd <= 1.0 ? 4.0 : 2.0;
        const double by_count = (double)iters * 32.0 * per_simd * cyc_per_fma / (ms * 1e3);
        char label[64];
        snprintf(label, sizeof label, "%d per CU (%g per SIMD)", per_cu, per_simd);
        printf("%-28s %10.3f %12.0f %12.0f %12.1f\n", label, ms, by_count, mhz[n / 2], rt[n / 2]);
        CHECK(hipFree(rec));
        CHECK(hipFree(sink));
    }
    printf("MHz by count assumes 4 cycles per v_fma_f32 with one wave per SIMD and 2 with two or more;\n"
           "realtime MHz = s_memrealtime ticks per microsecond of host-timed kernel (100 = the nominal 100 MHz).\n");
    return 0;
}
