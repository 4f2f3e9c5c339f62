// When is what a kernel stored into HOST memory visible to the host?  (Round 3 saw, once, the last 16 bytes of a
// tile of a zero-copy pass still holding the caller's zeros when readStream returned; LABBOOK.md section 9.)
//
// A kernel writes a per-launch pattern into host memory in the decimator's store shape -- whole 1 KiB lines from
// 64 lanes x 16 bytes, then a ragged last tile written element by element with 4-byte stores, ending in the middle
// of a cache line -- the host waits, then checks EVERY word.  Cells:
//   memory:  hipHostMalloc default flags | hipHostMalloc coherent | hipHostMalloc non-coherent | hipHostRegister
//   stores:  non-temporal | plain          (whole-line stores; the ragged tail is plain 4-byte stores in both)
//   wait:    event (disable-timing, the chains' sxfir_event_create) | event created with hipEventReleaseToSystem |
//            hipStreamSynchronize
//   load:    idle chip | a streaming kernel running beside it on a second stream (the chains' read-ahead)
// A word still holding the previous launch's value (or the poison the host wrote in between) is a visibility failure.
//   hipcc --offload-arch=gfx950 -O3 tools/hostvis_probe.hip -o /tmp/hostvis_probe
//   /tmp/hostvis_probe [iterations per cell, default 100000] [cell filter substring]
// Profiling aid (tools/), not part of the product.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(2); } } while (0)

// n_words 8-byte words: word i = tag << 32 | i.  Tiles of 256 words (2 KiB) per 64-thread workgroup: two 16-byte
// stores per lane; the last, ragged tile (n_words % 256 words) goes out as 4-byte stores, like the decimator's.
template <bool NT>
__global__ __launch_bounds__(64) void store_pattern(uint64_t *dst, unsigned n_words, unsigned tag)
{
    const unsigned lane = threadIdx.x, t0 = blockIdx.x * 256u;
    if (t0 + 256u <= n_words) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const unsigned w = t0 + 128u * k + 2u * lane;           // two words = 16 bytes per lane
            f4 v;
            v.x = __uint_as_float(w); v.y = __uint_as_float(tag); v.z = __uint_as_float(w + 1u); v.w = __uint_as_float(tag);
            f4 *p = reinterpret_cast<f4 *>(dst + w);
            if (NT) __builtin_nontemporal_store(v, p); else *p = v;
        }
    } else {
        unsigned *d32 = reinterpret_cast<unsigned *>(dst);
        for (unsigned w = t0 + 4u * lane; w < t0 + 4u * lane + 4u && w < n_words; ++w) { d32[2 * w] = w; d32[2 * w + 1] = tag; }
    }
}
__global__ __launch_bounds__(256) void background(const f4 *in, f4 *out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i] * 1.0001f;
}

int main(int argc, char **argv)
{
    const long iters = argc > 1 ? atol(argv[1]) : 100000;
    const char *filter = argc > 2 ? argv[2] : "";
    // 17 whole tiles + a ragged one that ends 24 bytes into a 128-byte line
    const unsigned n_words = 17 * 256 + 83;
    const size_t bytes = (size_t)n_words * 8;
    hipStream_t st, bg;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&bg, hipStreamNonBlocking));
    hipEvent_t ev_plain, ev_sys;
    CK(hipEventCreateWithFlags(&ev_plain, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&ev_sys, hipEventDisableTiming | hipEventReleaseToSystem));
    f4 *bin, *bout;
    const size_t bn = (size_t)1 << 24;                     // 256 MiB each way per background launch
    CK(hipMalloc(&bin, bn * 16)); CK(hipMalloc(&bout, bn * 16)); CK(hipMemset(bin, 0, bn * 16));

    const char *mem_name[] = {"hipHostMalloc default", "hipHostMalloc coherent", "hipHostMalloc non-coherent", "hipHostRegister"};
    const char *wait_name[] = {"event (disable-timing)", "event (release-to-system)", "stream sync"};
    printf("# hostvis_probe: %ld launches per cell, %u words (%zu bytes) per launch, every word checked\n", iters, n_words, bytes);
    long total_bad = 0;
    for (int mem = 0; mem < 4; ++mem) {
        void *host = nullptr, *raw = nullptr;
        if (mem == 0) CK(hipHostMalloc(&host, bytes + 4096, hipHostMallocDefault));
        else if (mem == 1) CK(hipHostMalloc(&host, bytes + 4096, hipHostMallocCoherent));
        else if (mem == 2) CK(hipHostMalloc(&host, bytes + 4096, hipHostMallocNonCoherent));
        else {
            if (posix_memalign(&raw, 4096, bytes + 4096)) return 2;
            memset(raw, 0, bytes + 4096);
            CK(hipHostRegister(raw, bytes + 4096, hipHostRegisterDefault));
            host = raw;
        }
        uint64_t *dev = nullptr;
        CK(hipHostGetDevicePointer((void **)&dev, host, 0));
        uint64_t *h = (uint64_t *)host;
        for (int nt = 1; nt >= 0; --nt)
            for (int wait = 0; wait < 3; ++wait)
                for (int load = 0; load < 2; ++load) {
                    char name[256];
                    snprintf(name, sizeof name, "%-26s | %-5s stores | %-25s | %s", mem_name[mem], nt ? "nt" : "plain", wait_name[wait],
                             load ? "beside a streaming kernel" : "idle chip");
                    if (filter[0] && !strstr(name, filter)) continue;
                    long bad_launches = 0, bad_words = 0, first_launch = -1;
                    unsigned first_word = 0;
                    uint64_t first_value = 0;
                    int bg_inflight = 0;
                    for (long it = 0; it < iters; ++it) {
                        const unsigned tag = (unsigned)(it + 1) + 0x1000000u * (unsigned)(mem * 12 + nt * 6 + wait * 2 + load + 1);
                        if (load && (it % 64) == 0) {
                            if (bg_inflight >= 3) { CK(hipStreamSynchronize(bg)); bg_inflight = 0; }
                            hipLaunchKernelGGL(background, dim3(4096), dim3(256), 0, bg, bin, bout, bn);
                            ++bg_inflight;
                        }
                        if (nt) hipLaunchKernelGGL(store_pattern<true>, dim3((n_words + 255) / 256), dim3(64), 0, st, dev, n_words, tag);
                        else hipLaunchKernelGGL(store_pattern<false>, dim3((n_words + 255) / 256), dim3(64), 0, st, dev, n_words, tag);
                        if (wait == 0) { CK(hipEventRecord(ev_plain, st)); CK(hipEventSynchronize(ev_plain)); }
                        else if (wait == 1) { CK(hipEventRecord(ev_sys, st)); CK(hipEventSynchronize(ev_sys)); }
                        else CK(hipStreamSynchronize(st));
                        long bw = 0;
                        for (unsigned w = 0; w < n_words; ++w) {
                            const uint64_t want = ((uint64_t)tag << 32) | w, got = ((volatile uint64_t *)h)[w];
                            if (got != want) {
                                if (first_launch < 0) { first_launch = it; first_word = w; first_value = got; }
                                ++bw;
                            }
                        }
                        if (bw) { ++bad_launches; bad_words += bw; }
                    }
                    CK(hipStreamSynchronize(bg));
                    printf("%s : %ld of %ld launches with stale words (%ld words)", name, bad_launches, iters, bad_words);
                    if (bad_launches)
                        printf("; first: launch %ld word %u (byte %u of its 128-byte line, tile %u%s) held 0x%016llx", first_launch, first_word,
                               (first_word * 8u) % 128u, first_word / 256u, first_word >= 17u * 256u ? " = the ragged tile" : "",
                               (unsigned long long)first_value);
                    printf("\n");
                    fflush(stdout);
                    total_bad += bad_launches;
                }
        if (mem == 3) { CK(hipHostUnregister(raw)); free(raw); }
        else CK(hipHostFree(host));
    }
    printf("# total launches with stale words: %ld\n", total_bad);
    return total_bad ? 1 : 0;
}
