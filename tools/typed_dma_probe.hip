// Can the texture path convert CF16 to CF32 on the way INTO LDS?  (round 5, the CF16 /32 kernel's conversions cost 13.6 % of
// its FIR phase: profiles/round5_valu_power_probe_cf16.txt)
//
// gfx950's LDS-DMA (`... lds`) exists for buffer_load_dword[x3|x4] and, of the typed loads, for buffer_load_format_x only.
// With a buffer descriptor whose format is {DATA_FORMAT 16, NUM_FORMAT FLOAT} a buffer_load_format_x fetches one half per
// lane and -- if the conversion applies to the LDS path as it does to the VGPR path -- writes one float per lane: one
// instruction turns 128 contiguous source bytes (32 CF16 samples) into one 256-byte row of a CF32 image, no VALU, no
// ds_write.  This probe answers two questions on the hardware:
//   1. is what lands in LDS the converted value, bit for bit (every finite half, +-0, subnormals, +-inf; NaN payloads reported)?
//   2. at what rate can a kernel stage a stream that way, against the raw 16-byte LDS-DMA of the same CF16 bytes and against
//      typed loads into VGPRs (buffer_load_format_xyzw with 16_16_16_16 FLOAT) + ds_write_b128?
// Staging shape = the dense /32 kernel's: 4-wave workgroups, a tile = 160 rows of 32 complex samples (one pad slot per 16
// rows), four workgroups per CU, nothing but the staging and a read-back of the image (checksum) per tile.
// hipcc --offload-arch=gfx950 -O3 tools/typed_dma_probe.hip -o /tmp/typed_dma_probe && /tmp/typed_dma_probe
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int ROWS = 160;                 // rows per tile (the dense /32 image has 159)
constexpr int ROW_SAMPLES = 32;           // complex samples per row: 128 B as CF16, 256 B as CF32
constexpr int ROW_SLOTS = 16;             // 16-byte slots per CF32 row
constexpr int IMG_SLOTS = ROWS * ROW_SLOTS + ROWS / 16;   // + one pad slot per 16 rows
constexpr int TILE_SAMPLES = ROWS * ROW_SAMPLES;          // 5120

// GFX9 buffer resource word 3: DST_SEL_X..W [11:0], NUM_FORMAT [14:12], DATA_FORMAT [18:15]
constexpr int W3_RAW = 0x00020000;                                                    // DATA_FORMAT 32
constexpr int W3_F16_X = 4 | (7 << 12) | (2 << 15);                                   // 16, FLOAT, X <- R
constexpr int W3_F16_XYZW = 4 | (5 << 3) | (6 << 6) | (7 << 9) | (7 << 12) | (12 << 15);   // 16_16_16_16, FLOAT

__device__ __forceinline__ v4i make_rsrc(const void *base, unsigned bytes, int word3)
{
    const unsigned long long a = (unsigned long long)base;
    v4i r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32)) & 0xffff;          // stride 0
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = word3;
    return r;
}

__device__ __forceinline__ int row_slot(int row) { return row * ROW_SLOTS + row / 16; }

// MODE 0: buffer_load_format_x ... lds with the 16/FLOAT descriptor: one row per instruction
// MODE 1: raw CF16 bytes by 16-byte LDS-DMA (global_load_lds_dwordx4): the CF16 image the multi kernel stages today (half the LDS)
// MODE 2: typed loads into VGPRs (buffer_load_format_xyzw, 16_16_16_16 FLOAT: two samples per lane) + ds_write_b128
template <int MODE>
__global__ __launch_bounds__(256) void stage(const uint32_t *__restrict__ in, float *__restrict__ sums, float *__restrict__ dump,
                                             int tiles_per_wg, unsigned in_bytes)
{
    __shared__ __attribute__((aligned(16))) f4 img[IMG_SLOTS];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform, and provably so
    const v4i rs = make_rsrc(in, in_bytes, MODE == 0 ? W3_F16_X : (MODE == 2 ? W3_F16_XYZW : W3_RAW));
    float acc = 0.0f;
    for (int t = 0; t < tiles_per_wg; ++t) {
        const long long tile = (long long)blockIdx.x * tiles_per_wg + t;
        const unsigned tile_byte = (unsigned)(tile * TILE_SAMPLES * 4);
        if (MODE == 0) {
            const unsigned voff = 2u * (unsigned)lane;                     // one half per lane
            for (int r = wave; r < ROWS; r += 4) {
                const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(img + row_slot(r)));
                const unsigned soff = __builtin_amdgcn_readfirstlane(tile_byte + 128u * (unsigned)r);
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_format_x %1, %2, %3 offen lds"
                             :: "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
            }
        } else if (MODE == 1) {
            // 20 KiB of CF16 per tile: 1280 chunks of 16 bytes, five per lane; stored as a half-size image
            for (int i = 0; i < 5; ++i) {
                const int chunk = 256 * i + 64 * wave;                     // wave-uniform base, + lane inside the instruction
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const char *)in + tile_byte + 16 * (chunk + lane)),
                                                 (__attribute__((address_space(3))) void *)(img + chunk), 16, 0, 0);
            }
        } else {
            // two samples (8 source bytes) per lane and instruction -> four floats -> one ds_write_b128: ten of each per lane and tile
            f4 v[10];
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                const unsigned voff = tile_byte + 8u * (unsigned)(256 * i + threadIdx.x);
                asm volatile("buffer_load_format_xyzw %0, %1, %2, 0 offen" : "=v"(v[i]) : "v"(voff), "s"(rs) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                const int c = 256 * i + threadIdx.x;                       // chunk = two samples = one slot
                img[c + c / 256] = v[i];
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // read the image back (the FIR would): 10 slots per lane (5 for the half-size image)
        const int nslots = MODE == 1 ? 1280 : ROWS * ROW_SLOTS;
        for (int s = threadIdx.x; s < nslots; s += 256) {
            const f4 v = MODE == 1 ? img[s] : img[s + s / 256];
            acc += v.x + v.y + v.z + v.w;
        }
        if (dump && tile == 0) {
            for (int s = threadIdx.x; s < nslots; s += 256) {
                const f4 v = MODE == 1 ? img[s] : img[s + s / 256];
                reinterpret_cast<f4 *>(dump)[s] = v;
            }
        }
        __syncthreads();
    }
    sums[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

// Second question (the S32_LE wire words of row f-3): does a {32, SSCALED} descriptor turn an int32 into (float)int32 on the way in?
// One wave, 64 words per instruction, formats tried: SSCALED (3), SINT (5: expected to pass the bits through), FLOAT (7).
__global__ void stage_s32(const int *__restrict__ in, unsigned *__restrict__ out, int nfmt)
{
    __shared__ unsigned img[64];
    const int lane = threadIdx.x;
    v4i rs = make_rsrc(in, 256, 0);
    rs.w = 4 | (nfmt << 12) | (4 << 15);                                      // DATA_FORMAT 32
    const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)img);
    const unsigned voff = 4u * (unsigned)lane;
    img[lane] = 0xdeadbeefu;
    __syncthreads();
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_format_x %1, %2, 0 offen lds\n\ts_waitcnt vmcnt(0)" :: "s"(lds_addr), "v"(voff), "s"(rs) : "memory");
    __syncthreads();
    out[lane] = img[lane];
}

static float half_to_float(uint16_t h)
{
    const uint32_t s = (uint32_t)(h >> 15) << 31, e = (h >> 10) & 31, m = h & 1023;
    uint32_t bits;
    if (e == 0) {
        if (m == 0) bits = s;
        else {
            int sh = 0;
            uint32_t mm = m;
            while (!(mm & 1024)) { mm <<= 1; ++sh; }
            bits = s | ((uint32_t)(113 - sh) << 23) | ((mm & 1023) << 13);
        }
    } else if (e == 31) bits = s | 0x7f800000u | (m << 13);
    else bits = s | ((e + 112) << 23) | (m << 13);
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

int main()
{
    const int tiles_per_wg = 48;
    const int wgs = 1024;                                 // four 4-wave workgroups per CU
    const long long samples = (long long)wgs * tiles_per_wg * TILE_SAMPLES;      // 2^27.9: ~1 GiB of CF16
    std::vector<uint32_t> h((size_t)samples);
    uint64_t st = 0x9e3779b97f4a7c15ull;
    for (auto &w : h) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; w = (uint32_t)st; }
    // the first tile: every 16-bit pattern once (65536 halfs = 32768 samples > one tile: the first 10240 halfs of it), so
    // seed the first tile with patterns spread over the whole range incl. subnormals, infinities and NaNs
    for (int i = 0; i < TILE_SAMPLES; ++i) {
        const uint16_t a = (uint16_t)(i * 13 + 1), b = (uint16_t)(0xFFFF - i * 7);
        h[i] = (uint32_t)a | ((uint32_t)b << 16);
    }
    const uint16_t special[] = {0x0000, 0x8000, 0x0001, 0x8001, 0x03FF, 0x0400, 0x7BFF, 0xFBFF, 0x7C00, 0xFC00, 0x7E00, 0x7C01, 0xFE00, 0x3C00, 0xBC00, 0x3555};
    for (int i = 0; i < 16; ++i) h[100 + i] = (uint32_t)special[i] | ((uint32_t)special[15 - i] << 16);
    uint32_t *in; float *sums, *dump;
    CK(hipMalloc(&in, samples * 4)); CK(hipMalloc(&sums, (size_t)wgs * 256 * 4)); CK(hipMalloc(&dump, (size_t)TILE_SAMPLES * 8));
    CK(hipMemcpy(in, h.data(), samples * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[3] = {"buffer_load_format_x lds, 16/FLOAT descriptor (one CF32 row per instruction)",
                            "global_load_lds_dwordx4 of the raw CF16 bytes (today's staging, half-size image)",
                            "buffer_load_format_xyzw into VGPRs (16_16_16_16 FLOAT) + ds_write_b128"};
    for (int mode = 0; mode < 3; ++mode) {
        auto launch = [&](float *d) {
            if (mode == 0) hipLaunchKernelGGL(stage<0>, dim3(wgs), dim3(256), 0, 0, in, sums, d, tiles_per_wg, (unsigned)(samples * 4));
            else if (mode == 1) hipLaunchKernelGGL(stage<1>, dim3(wgs), dim3(256), 0, 0, in, sums, d, tiles_per_wg, (unsigned)(samples * 4));
            else hipLaunchKernelGGL(stage<2>, dim3(wgs), dim3(256), 0, 0, in, sums, d, tiles_per_wg, (unsigned)(samples * 4));
        };
        CK(hipMemset(dump, 0xff, (size_t)TILE_SAMPLES * 8));
        launch(dump);
        CK(hipDeviceSynchronize());
        if (mode != 1) {
            std::vector<float> got((size_t)TILE_SAMPLES * 2);
            CK(hipMemcpy(got.data(), dump, got.size() * 4, hipMemcpyDeviceToHost));
            long bad = 0, nan_payload = 0, first = -1;
            unsigned nan_half = 0, nan_got = 0, nan_want = 0;
            for (int i = 0; i < TILE_SAMPLES * 2; ++i) {
                const uint16_t hv = (uint16_t)(h[i / 2] >> (16 * (i & 1)));
                const float want = half_to_float(hv);
                uint32_t gb, wb;
                memcpy(&gb, &got[i], 4); memcpy(&wb, &want, 4);
                if (gb != wb) {
                    if (want != want && got[i] != got[i]) {                    // both NaN, payload or quiet bit differs
                        if (!nan_payload) { nan_half = hv; nan_got = gb; nan_want = wb; }
                        ++nan_payload;
                    }
                    else { ++bad; if (first < 0) first = i; }
                }
            }
            printf("%-90s tile 0: %ld of %d floats differ from half->float (NaN payload differences: %ld)%s\n", names[mode], bad,
                   TILE_SAMPLES * 2, nan_payload, bad ? " <-- NOT the converted values" : "");
            if (nan_payload) printf("   a NaN: half 0x%04x -> 0x%08x (v_cvt_f32_f16 / this host: 0x%08x)\n", nan_half, nan_got, nan_want);
            if (bad) {
                const int i = (int)first;
                uint32_t gb; memcpy(&gb, &got[i], 4);
                printf("   first: float %d: half 0x%04x -> got 0x%08x, want %g\n", i, (unsigned)(uint16_t)(h[i / 2] >> (16 * (i & 1))), gb,
                       half_to_float((uint16_t)(h[i / 2] >> (16 * (i & 1)))));
            }
        }
        for (int i = 0; i < 30; ++i) launch(nullptr);
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 50; ++i) launch(nullptr);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 50;
        printf("%-90s %.4f ms per %.3f G samples = %.0f GB/s of CF16 source (%.2f x 2^28 samples per 0.5 ms)\n", names[mode], ms, samples / 1e9,
               samples * 4.0 / (ms * 1e-3) / 1e9, (samples / (double)(1 << 28)) * 0.5 / ms);
        fflush(stdout);
    }
    // ---- 32-bit integer formats
    {
        int hw[64];
        for (int i = 0; i < 64; ++i) hw[i] = (int)(0x9e3779b9u * (unsigned)(i + 1));
        hw[0] = 0; hw[1] = 1; hw[2] = -1; hw[3] = 0x7fffffff; hw[4] = (int)0x80000000; hw[5] = 0x7fffff80; hw[6] = 0x7fffffbf; hw[7] = 0x7fffffc0; hw[8] = 16777217; hw[9] = -16777217;
        int *din; unsigned *dout;
        CK(hipMalloc(&din, 256)); CK(hipMalloc(&dout, 256));
        CK(hipMemcpy(din, hw, 256, hipMemcpyHostToDevice));
        const int fmts[3] = {3, 5, 7};
        const char *fn[3] = {"SSCALED", "SINT", "FLOAT"};
        for (int f = 0; f < 3; ++f) {
            hipLaunchKernelGGL(stage_s32, dim3(1), dim3(64), 0, 0, din, dout, fmts[f]);
            unsigned got[64];
            CK(hipMemcpy(got, dout, 256, hipMemcpyDeviceToHost));
            int as_float = 0, raw = 0;
            for (int i = 0; i < 64; ++i) {
                const float want = (float)hw[i];
                unsigned wb; memcpy(&wb, &want, 4);
                as_float += got[i] == wb;
                raw += got[i] == (unsigned)hw[i];
            }
            printf("buffer_load_format_x lds, {32, %s}: %d of 64 words equal (float)int32, %d of 64 equal the raw word; word[1]=0x%08x word[3]=0x%08x word[6]=0x%08x\n",
                   fn[f], as_float, raw, got[1], got[3], got[6]);
        }
    }
    return 0;
}
