// C ABI of the MI355X resampling path (include/sxfir.h).  Host side of the
// "thin extern C shim": plan bookkeeping, kernel selection and launches.
#include "../../include/sxfir.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <new>
#include <utility>
#include <vector>

#include "sxfir_decim_tile.hip.h"
#include "sxfir_decim_multi.hip.h"

// Instantiated (ratio, waves per workgroup, CF16, row split) variants of the multi-column decimator: one list
// for the occupancy query in sxfir_create and the launch in launch_decim.  The production library carries
// what it launches; the profiling build (-DSXFIR_PROFILING, libsxfir_prof.so: tools/ and
// tests/test_gpu_variants.py) adds the A/B variants, the ablation modes and the environment knobs.
// (shipped: CF16 storage only -- since round 3 every CF32 / S32-word plan at ratio 8, 16, 32 runs decim_dense_kernel,
// so the multi-column kernel's CF32 and S32 instances exist in the profiling build alone, as the A/B partner)
#define SXFIR_MULTI_SHIPPED(X) \
    X(4, 1, true, 2) X(8, 4, true, 2) X(16, 4, true, 2) X(32, 4, true, 2)
#ifdef SXFIR_PROFILING
#define SXFIR_MULTI_VARIANTS(X) \
    SXFIR_MULTI_SHIPPED(X) \
    X(8, 4, false, 2) X(16, 4, false, 2) X(32, 4, false, 2) \
    X(4, 1, false, 2) X(4, 4, false, 2) X(8, 1, false, 2) X(8, 2, false, 2) X(16, 2, false, 2) X(32, 8, false, 2) \
    X(8, 1, true, 2) X(8, 2, true, 2) X(16, 2, true, 2) X(32, 8, true, 2) \
    X(4, 2, false, 4) X(8, 2, false, 4) X(8, 4, false, 4) X(16, 4, false, 4) X(16, 8, false, 4) X(32, 8, false, 4) \
    X(32, 16, false, 4) \
    X(4, 2, true, 4) X(8, 4, true, 4) X(16, 8, true, 4) X(32, 8, true, 4)
// profiling modes of decim4_tile_kernel<128> (its ABL template argument)
#define SXFIR_TILE_ABLATIONS(X) X(1) X(2) X(3) X(7) X(8) X(9) X(10) X(11) X(12) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24)
// (waves per workgroup, option bits) variants of decim4_tile2_kernel<128>
#define SXFIR_TILE2_VARIANTS(X) \
    X(1, 0) X(1, 1) X(1, 2) X(1, 3) X(1, 4) X(1, 5) X(1, 6) X(1, 7) X(1, 9) X(1, 11) \
    X(2, 0) X(2, 1) X(2, 2) X(2, 3) X(2, 6) X(2, 7) X(4, 2) X(4, 3) X(4, 7) X(8, 2) X(8, 3) \
    X(1, 16) X(1, 17) X(1, 19) X(2, 17) X(2, 19) X(1, 32) X(1, 33) X(2, 35) X(1, 49) X(2, 51) \
    X(1, 64) X(1, 65) X(1, 68) X(1, 69) X(1, 80) X(1, 81) X(2, 64) X(2, 65) X(4, 65) \
    X(16, 192) X(16, 193) X(16, 128) X(8, 192) X(4, 192) X(16, 224) X(1, 320) X(1, 576) X(1, 1088) X(1, 2112) X(1, 5184) X(1, 9280) X(1, 13376) X(1, 16448) \
    X(1, 33856) X(1, 66624) X(1, 33872) X(1, 66625) X(1, 67136) X(1, 132160) X(1, 197696) \
    X(16, 66752) X(8, 66752) X(4, 66752) X(16, 1216) X(1, 263232) X(1, 525376)
// variants that also exist with phase stamps (ABL 5)
#define SXFIR_TILE2_STAMPED(X) X(1, 33856) X(1, 66624) X(1, 525376) X(1, 9280) X(1, 5184) X(1, 1088) X(1, 0) X(1, 1) X(2, 3) X(1, 17) X(1, 5) X(1, 64) X(1, 65) X(1, 69) X(16, 192) X(16, 128)
#else
#define SXFIR_MULTI_VARIANTS(X) SXFIR_MULTI_SHIPPED(X)
#endif
#define SXFIR_MULTI_KEY(DD, WW, HH, PP) (((PP) == 4 ? 1000000 : 0) + ((HH) ? 10000 : 0) + (DD) * 100 + (WW))
#include "sxfir_decim_dense.hip.h"
#include "sxfir_interp_tile.hip.h"
#include "sxfir_interp_pass.hip.h"
#include "sxfir_decim_tile2.hip.h"
#include "sxfir_decim_wide.hip.h"               // /4, 128 symmetric taps: the shipped form since round 4
#ifdef SXFIR_PROFILING
#include "experiments/sxfir_decim_pair.hip.h"   // measured variant, not shipped
#endif
#ifdef SXFIR_PROFILING
#include "experiments/sxfir_decim_sgpr.hip.h"
#include "../../include/sxfir_prof.h"
#endif
#include "sxfir_kernels.hip.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIPCHECK(expr)                                                                        \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(SXFIR_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                        __LINE__);                                                            \
    } while (0)

inline hipStream_t S(void *s) { return reinterpret_cast<hipStream_t>(s); }

inline size_t sample_bytes(int fmt) { return fmt == SXFIR_CF16 ? 4 : 8; }   // CF32 and S32 words: 8 bytes

}  // namespace

struct sxfir_plan {
    int mode, ntaps, ratio, nchan, fmt, device;
    int kernel;            // SXFIR_KERNEL_*
    int hist_len;          // samples of history kept per channel
    int jsplit, cw;        // numeric contract
    bool tile_capable;     // decim4_tile_kernel (ratio 4, 128 or 64 taps, CF32)
    bool multi_capable;    // decim_multi_kernel (ratio 8/16/32, 32 taps per phase, CF32)
    bool itile_capable;    // interp_tile_kernel (ratio 4/8/16/32, 32 taps per phase, CF32)
    int dense_nt;          // profiling build, SXFIR_DENSE_NT = 1 / 0: decim_dense_kernel with nt / plain staging loads at every ratio
    int dense_nt_set;      // ... and whether the knob was given at all
    bool dense_hc;         // (profiling) SXFIR_DENSE_HC=1: decim_dense_kernel with halo carry (/32, /16)
    bool dense_subset;     // /8, CF32 or S32 words: the scalar-tap form of decim_dense_kernel (tap subsets on the four waves)
    bool dense32;          // decim_dense_kernel (ratio 8 / 16 / 32, 32 taps per phase, CF32 / S32): the linear-image form
    int multi_waves;       // waves per workgroup of the multi kernel
    int multi_ps;          // lanes that share the 32 tap rows of one output (2 or 4) in the multi kernel
    int occ_multi;         // resident workgroups per CU of the multi kernel
    bool tile_dbuf;        // double-buffered LDS-DMA variant of the tile kernel
    int occ_sb, occ_db;    // resident waves per CU of the two tile-kernel variants
    int oversub;           // waves launched = CUs * occupancy * oversub
    void *stamps_dev;      // diagnostic clock stamps (ABLATE 11/12 only)
    size_t stamps_n;
    float thr2;            // S32 interpolator: transmitter-keying threshold (squared magnitude)
    int sgpr_r;            // experiment: SGPR-tap variant with R outputs per lane (0 = off)
    int sched;             // tile schedule of the tile kernel (0 strided passes, 1 contiguous runs)
    int ablate;            // profiling only: 1 = memory side alone, 2 = compute side alone
    int lds_pad;           // profiling only: extra dynamic LDS bytes per workgroup of a tile2 variant (caps the waves per CU)
    int t2_wpg, t2_opt;    // profiling only: decim4_tile2_kernel variant (waves per workgroup, T2_* bits); wpg 0 = off
    bool pair;             // decim4_pair_kernel: the two tap halves on the two waves of a workgroup
    bool pair_xsep;        // ... with a separate exchange buffer (two barriers per tile instead of four)
    int occ_pair;          // its resident workgroups per CU
    bool wide8;            // product: /4 with 128 symmetric taps runs decim4_wide_kernel (8 outputs per lane, 512-output tiles)
    bool wide;             // (profiling) a non-default build of decim4_wide_kernel was asked for ("wide<nb>", "wident<nb>")
    int wide_nb;           // (profiling) its LDS read-ahead depth: 0 = default
    bool wide_nt;          // (profiling) "wident...": with non-temporal staging loads
    bool wide_pin;         // (profiling) "widentp...": and the FMA issue order pinned (volatile asm)
    int wide_pol;          // (profiling) SXFIR_WIDE_POL: cache policy of its nt loads (low byte) and stores (next byte)
    int occ_wide;
    int compute_units;
    float *taps_dev;
    float *taps_scaled_dev;   // decimators: taps * 2^-31 (exact), the scalar-tap kernels on S32 wire words; x8 interpolators: the
                              // pass-major tap table of interp8_pass_kernel (pass (c, p) at 64 (2c + p), (jj, rr) at 4 jj + rr)
    bool ipass;               // x8, 256 taps, CF32: interp8_pass_kernel (scalar taps, four passes per tile)
    int occ_ipass;
    int ipass_qi;             // inputs per lane of that kernel (2; profiling: 4)
    float taps_k[64];         // the first 64 taps (times 2^-31 for S32 plans) for kernels that take them by value
    bool symmetric;           // taps[k] == taps[ntaps-1-k] bit for bit (every linear-phase design)
    void *hist_dev;        // current history: nchan * hist_len samples
    void *hist_alt;        // the tile kernel writes the next history here, then the two swap
    long long consumed, produced;
};

extern "C" {

int sxfir_abi_version(void) { return SXFIR_ABI_VERSION; }

const char *sxfir_last_error(void) { return g_err; }

int sxfir_device_count(int *count)
{
    if (!count) return fail(SXFIR_EINVAL, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(SXFIR_ENODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return SXFIR_OK;
}

int sxfir_device_info(int device, char *name, char *arch, int *compute_units, size_t *hbm_bytes)
{
    hipDeviceProp_t p;
    HIPCHECK(hipGetDeviceProperties(&p, device));
    if (name) snprintf(name, 64, "%s", p.name);
    if (arch) {
        snprintf(arch, 32, "%s", p.gcnArchName);
        char *colon = strchr(arch, ':');
        if (colon) *colon = 0;
    }
    if (compute_units) *compute_units = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = p.totalGlobalMem;
    return SXFIR_OK;
}

int sxfir_create(sxfir_plan **out, int mode, const float *taps, int ntaps, int ratio, int nchan, int fmt,
                 int device)
{
    if (!out || !taps) return fail(SXFIR_EINVAL, "NULL argument");
    *out = nullptr;
    if (mode != SXFIR_DECIMATE && mode != SXFIR_INTERPOLATE) return fail(SXFIR_EINVAL, "bad mode %d", mode);
    if (ntaps < 1 || ntaps > 65536) return fail(SXFIR_EINVAL, "ntaps %d out of range", ntaps);
    if (ratio < 1 || ratio > 4096) return fail(SXFIR_EINVAL, "ratio %d out of range", ratio);
    if (nchan < 1 || nchan > 65535) return fail(SXFIR_EINVAL, "nchan %d out of range", nchan);
    if (fmt != SXFIR_CF32 && fmt != SXFIR_CF16 && fmt != SXFIR_S32) return fail(SXFIR_EINVAL, "bad format %d", fmt);
    if (mode == SXFIR_INTERPOLATE && ntaps % ratio)
        return fail(SXFIR_EINVAL, "interpolator needs ntaps %% ratio == 0 (%d, %d)", ntaps, ratio);

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(SXFIR_ENODEVICE, "no HIP device visible; this library has no CPU path");
    if (device < 0) HIPCHECK(hipGetDevice(&device));
    if (device >= ndev) return fail(SXFIR_EINVAL, "device %d of %d", device, ndev);
    HIPCHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHECK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(SXFIR_ENODEVICE, "device %d is %s; kernels are built for gfx950 only", device,
                    prop.gcnArchName);

    sxfir_plan *p = new (std::nothrow) sxfir_plan();
    if (!p) return fail(SXFIR_ENOMEM, "out of host memory");
    p->mode = mode;
    p->ntaps = ntaps;
    p->ratio = ratio;
    p->nchan = nchan;
    p->fmt = fmt;
    p->device = device;
    p->kernel = SXFIR_KERNEL_AUTO;
    p->compute_units = prop.multiProcessorCount;
    p->consumed = p->produced = 0;
    p->taps_dev = nullptr;
    p->taps_scaled_dev = nullptr;
    p->ipass = false;
    p->occ_ipass = 16;
    p->ipass_qi = 2;
    p->symmetric = true;
    for (int k = 0; k < ntaps / 2; ++k)
        if (memcmp(&taps[k], &taps[ntaps - 1 - k], sizeof(float)) != 0) p->symmetric = false;
    p->hist_dev = nullptr;
    p->hist_alt = nullptr;
    p->itile_capable = false;

    if (mode == SXFIR_DECIMATE) {
        p->hist_len = (ntaps + 1) & ~1;
        p->tile_capable = ((fmt == SXFIR_CF32 && ratio == 4 && (ntaps == 128 || ntaps == 64)) ||
                           (fmt == SXFIR_S32 && ratio == 4 && ntaps == 128));
        // multi-column kernel: 32 taps per phase; CF32 at ratio 8/16/32, CF16 at ratio 4/8/16/32
        p->multi_capable = (ntaps == 32 * ratio) &&
                           (((fmt == SXFIR_CF32 || fmt == SXFIR_S32) && (ratio == 8 || ratio == 16 || ratio == 32)) ||
                            (fmt == SXFIR_CF16 && (ratio == 4 || ratio == 8 || ratio == 16 || ratio == 32)));
        // Numeric contract: two row halves and column groups of 4 when the shape allows the
        // adjacent-pair trees, i.e. whole, even rows and a power-of-two number (<= 32) of column groups;
        // otherwise one chain over all taps.
        const int jt = (ntaps + ratio - 1) / ratio;
        const int ncol4 = ratio / 4;
        const bool pow2_cols = ratio % 4 == 0 && (ncol4 & (ncol4 - 1)) == 0 && ncol4 <= 32;
        if (ntaps % ratio == 0 && pow2_cols && jt % 2 == 0) {
            p->jsplit = 2;
            p->cw = 4;
        } else {
            p->jsplit = 1;
            p->cw = ratio;
        }
    } else {
        const int jt = ntaps / ratio;
        p->hist_len = (jt + 1) & ~1;
        p->tile_capable = false;
        p->multi_capable = false;
        p->itile_capable = ((fmt == SXFIR_CF32 || fmt == SXFIR_S32) && ntaps == 32 * ratio &&
                            (ratio == 4 || ratio == 8 || ratio == 16 || ratio == 32));
        p->jsplit = (jt % 2 == 0) ? 2 : 1;
        p->cw = 1;
    }

    // measured on MI355X (tools/kbench.py): single-buffered LDS-DMA at 16 waves/CU, 16 generations
    // of short-lived waves (4 tiles each at 2^28 samples), strided XCD-blocked passes; the
    // double-buffered variant at 8 waves/CU and long contiguous runs are slower
    p->tile_dbuf = false;
    p->occ_sb = p->occ_db = 8;
    p->oversub = 16;
    p->ablate = 0;
    p->sched = 0;
    p->sgpr_r = 0;
    p->thr2 = 1.0e-3f * 1.0e-3f;
    p->stamps_dev = nullptr;
    p->stamps_n = 0;
    // waves per workgroup of the multi-column kernel, measured (tools/kbench.py, KB_D, specs "w1".."w8"):
    // the choice that brings the LDS image down to 10 KiB per wave (16 waves per CU) while the 31-row
    // halo stays a small part of the staging
    p->multi_waves = ratio <= 4 ? 1 : 4;
    p->multi_ps = 2;
    // CF32 / S32 words at ratio 8, 16, 32: the linear-image form (sxfir_decim_dense.hip.h); CF16 and ratio 4 keep
    // the multi-column kernel
    p->dense32 = p->multi_capable && (ratio == 8 || ratio == 16 || ratio == 32) && fmt != SXFIR_CF16;
    // /8 CF32: the scalar-tap form of the dense kernel (tap subsets on the four waves, round 4: 4.4-5 % less time)
    p->dense_hc = false;
    p->dense_subset = p->dense32 && ratio == 8 && (fmt == SXFIR_CF32 || fmt == SXFIR_S32);
    p->t2_wpg = p->t2_opt = 0;
    p->dense_nt = 0;
    p->dense_nt_set = 0;
    p->lds_pad = 0;
    p->pair = false;
    p->pair_xsep = false;
    p->occ_pair = 8;
    p->wide8 = false;
    p->wide = false;
    p->wide_nt = false;
    p->wide_pin = false;
    p->wide_pol = 0;
    p->wide_nb = 0;
    p->occ_wide = 8;
    p->occ_multi = 2;
    // generations of workgroups per launch, measured (tools/kbench.py): the multi-column kernel's prologue
    // (64 taps and the DMA offset table per lane) is heavier than the tile kernel's, 8 beats 16; the
    // interpolator is flat between 2 and 16 (tools/ibench.py)
    if (p->multi_capable) p->oversub = 8;
    if (p->itile_capable) p->oversub = 4;
    if (p->itile_capable && ratio == 8 && (fmt == SXFIR_CF32 || fmt == SXFIR_S32)) {
        p->ipass = true;
        p->oversub = 8;                                 // measured (tools/ibench2.py): 4 / 8 / 16 generations within 0.3 %
        int nbi = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nbi, (const void *)sxfir::interp8_pass_kernel<2>, 64, 0) == hipSuccess && nbi > 0)
            p->occ_ipass = nbi;
    }
#ifdef SXFIR_PROFILING
    // A/B knobs of the profiling build.  The production library never looks at the environment.
    if (const char *v = getenv("SXFIR_TILE_VARIANT")) {
        if (strcmp(v, "mu") == 0 && mode == SXFIR_DECIMATE && fmt == SXFIR_CF32 && ratio == 4 && ntaps == 128) {
            p->multi_capable = true;        // the multi-column kernel at D = 4 instead of decim4_tile_kernel
            p->tile_capable = false;
            p->oversub = 8;
        }
    }
    if (const char *v = getenv("SXFIR_DENSE")) p->dense32 = p->dense32 && atoi(v) != 0;
    if (const char *v = getenv("SXFIR_IPASS")) {     // 0: interp_tile_kernel at x8 too (A/B); 4: four inputs per lane
        p->ipass = p->ipass && atoi(v) != 0;
        if (p->ipass && atoi(v) == 4) {
            p->ipass_qi = 4;
            int nbi = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nbi, (const void *)sxfir::interp8_pass_kernel<4>, 64, 0) == hipSuccess && nbi > 0)
                p->occ_ipass = nbi;
        }
    }
    if (const char *v = getenv("SXFIR_DENSE_NT")) { p->dense_nt = atoi(v); p->dense_nt_set = 1; }
    if (const char *v = getenv("SXFIR_DENSE_HC")) p->dense_hc = atoi(v) != 0;
    if (const char *v = getenv("SXFIR_DENSE_SUBSET")) p->dense_subset = p->dense_subset && atoi(v) != 0;     // 0: the VGPR-tap form (A/B)
    if (!p->dense32) p->dense_subset = false;
    if (getenv("SXFIR_MULTI_PS") || getenv("SXFIR_MULTI_W")) p->dense32 = false;   // those knobs belong to the multi-column kernel
    if (p->multi_capable && fmt != SXFIR_S32 && !p->dense32) {
        if (const char *v = getenv("SXFIR_MULTI_PS")) p->multi_ps = atoi(v) == 4 ? 4 : 2;
        if (p->multi_ps == 4) p->multi_waves = ratio <= 4 ? 2 : (ratio == 8 ? 4 : 8);
        if (const char *v = getenv("SXFIR_MULTI_W")) p->multi_waves = atoi(v);
        p->jsplit = p->multi_ps;
    }
    if (p->multi_capable || p->itile_capable || p->tile_capable) {
        if (const char *v = getenv("SXFIR_OVERSUB")) p->oversub = atoi(v) > 0 ? atoi(v) : 1;
    }
    if (p->multi_capable || p->tile_capable) {
        if (const char *v = getenv("SXFIR_ABLATE")) p->ablate = atoi(v);
    }
#endif
    if (p->multi_capable) {
        // resident workgroups per CU: LDS is the limiter (checked against the occupancy API below)
        const int W = p->multi_waves;
        int nb = 0;
        const void *k = nullptr;
        if (p->dense32) {
            const bool w = fmt == SXFIR_S32;
            k = ratio == 8    ? (w ? (const void *)sxfir::decim_dense_kernel<8, 0, true, 2, true> : (const void *)sxfir::decim_dense_kernel<8, 0, false, 2, true>)
                : ratio == 16 ? (w ? (const void *)sxfir::decim_dense_kernel<16, 0, true, 2> : (const void *)sxfir::decim_dense_kernel<16, 0, false, 2>)
                              : (w ? (const void *)sxfir::decim_dense_kernel<32, 0, true, 2> : (const void *)sxfir::decim_dense_kernel<32, 0, false, 2>);
#ifdef SXFIR_PROFILING
        } else if (fmt == SXFIR_S32) {   // wire-word input: one instantiation per ratio (4 waves, 2-way row split)
            k = ratio == 8    ? (const void *)sxfir::decim_multi_kernel<8, 4, false, 0, 2, true>
                : ratio == 16 ? (const void *)sxfir::decim_multi_kernel<16, 4, false, 0, 2, true>
                              : (const void *)sxfir::decim_multi_kernel<32, 4, false, 0, 2, true>;
#endif
        } else {
            switch (SXFIR_MULTI_KEY(ratio, W, fmt == SXFIR_CF16, p->multi_ps)) {
#define SXFIR_X(DD, WW, HH, PP) \
            case SXFIR_MULTI_KEY(DD, WW, HH, PP): k = (const void *)sxfir::decim_multi_kernel<DD, WW, HH, 0, PP>; break;
                SXFIR_MULTI_VARIANTS(SXFIR_X)
#undef SXFIR_X
            }
        }
        if (!k) {
            delete p;
            return fail(SXFIR_EUNSUPPORTED, "no multi-column kernel for ratio %d with %d waves per workgroup", ratio, W);
        }
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, 64 * W, 0) == hipSuccess && nb > 0) p->occ_multi = nb;
    }
    if (p->tile_capable) {
        int nb = 0;
        // the 4-outputs-per-lane kernels: the VGPR-tap form for any 128 or 64 taps -- and, in the profiling build, round 3's
        // scalar-tap form for 128 symmetric taps ("t2s"), the A/B partner of the wide kernel that replaced it
        const void *ksb = ntaps == 128 ? (const void *)sxfir::decim4_tile_kernel<128, false> : (const void *)sxfir::decim4_tile_kernel<64, false>;
#ifdef SXFIR_PROFILING
        if (ntaps == 128 && p->symmetric)
            ksb = fmt == SXFIR_S32 ? (const void *)sxfir::decim4_tile2_kernel<128, 1, sxfir::T2_SHIPPED, 0, true>
                                   : (const void *)sxfir::decim4_tile2_kernel<128, 1, sxfir::T2_SHIPPED>;
#endif
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ksb, 64, 0) == hipSuccess && nb > 0) p->occ_sb = nb;
        if (ntaps == 128 && p->symmetric) {
            // the shipped form for 128 symmetric taps: eight outputs per lane (sxfir_decim_wide.hip.h); 18.5 KB of LDS
            // per wave -> 8 waves per CU
            p->wide8 = true;
            const void *kw = fmt == SXFIR_S32 ? (const void *)sxfir::decim4_wide_kernel<0, true> : (const void *)sxfir::decim4_wide_kernel<0, false>;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kw, 64, 0) == hipSuccess && nb > 0) p->occ_wide = nb;
        }
#ifdef SXFIR_PROFILING
        if (ntaps == 128) {
            const void *kp = fmt == SXFIR_S32 ? (const void *)sxfir::decim4_pair_kernel<0, true> : (const void *)sxfir::decim4_pair_kernel<0, false>;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kp, 128, 0) == hipSuccess && nb > 0) p->occ_pair = nb;
        }
        const void *kdb = ntaps == 128 ? (const void *)sxfir::decim4_tile_kernel<128, true>
                                       : (const void *)sxfir::decim4_tile_kernel<64, true>;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kdb, 64, 0) == hipSuccess && nb > 0) p->occ_db = nb;
        if (const char *v = getenv("SXFIR_TILE_VARIANT")) {
            p->tile_dbuf = (strcmp(v, "db") == 0);
            // "sb", "db", "sg": the first-generation tile kernel (taps in VGPR pairs) also for symmetric taps
            if (strcmp(v, "sb") == 0 || strcmp(v, "db") == 0 || strncmp(v, "sg", 2) == 0) { p->symmetric = false; p->wide8 = false; }
            // "t2s": round 3's shipped form (decim4_tile2_kernel, T2_SHIPPED) as the A/B partner of the wide kernel
            if (strcmp(v, "t2s") == 0) p->wide8 = false;
            p->sgpr_r = strcmp(v, "sg") == 0 ? 8 : (strcmp(v, "sg4") == 0 ? 4 : 0);
            if (p->sgpr_r && ntaps == 128) {
                const void *k = p->sgpr_r == 8 ? (const void *)sxfir::decim4_sgpr_kernel<8>
                                               : (const void *)sxfir::decim4_sgpr_kernel<4>;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, 64, 0) == hipSuccess && nb > 0)
                    p->occ_sb = nb;
            }
            // "wide": decim4_wide_kernel (sxfir_decim_wide.hip.h), symmetric taps only
            if (strncmp(v, "wide", 4) == 0 && ntaps == 128 && p->symmetric) {
                p->wide = true;
                p->wide_nt = strncmp(v, "wident", 6) == 0;
                p->wide_pin = strncmp(v, "widentp", 7) == 0;
                p->wide_nb = atoi(v + (p->wide_pin ? 7 : (p->wide_nt ? 6 : 4)));
                // "widepol<hex>": the shipped build (wident24) with another cache policy (sxfir_decim_wide.hip.h, POL)
                if (strncmp(v, "widepol", 7) == 0) {
                    p->wide_nt = true;
                    p->wide_nb = 24;
                    p->wide_pol = (int)strtol(v + 7, nullptr, 16);
                }
            }
            // "pair": decim4_pair_kernel (sxfir_decim_pair.hip.h)
            if (strncmp(v, "pair", 4) == 0 && ntaps == 128) {
                p->pair = true;
                p->pair_xsep = strcmp(v, "pairx") == 0;
                if (p->pair_xsep) p->occ_pair = 7;
            }
            // "t2:<waves per workgroup>:<option bits>": decim4_tile2_kernel (sxfir_decim_tile2.hip.h)
            if (strncmp(v, "t2:", 3) == 0 && ntaps == 128 && fmt == SXFIR_CF32) {
                int wpg = 0, opt = 0;
                if (sscanf(v + 3, "%d:%d", &wpg, &opt) == 2) {
                    const void *k = nullptr;
                    switch (wpg * 100 + opt) {
#define SXFIR_X(WW, OO) case WW * 100 + OO: k = (const void *)sxfir::decim4_tile2_kernel<128, WW, OO>; break;
                        SXFIR_TILE2_VARIANTS(SXFIR_X)
#undef SXFIR_X
                    }
                    if (!k || ((opt & sxfir::T2_SCALAR) && !p->symmetric)) {
                        delete p;
                        return fail(SXFIR_EUNSUPPORTED, "no tile2 variant %d:%d for these taps", wpg, opt);
                    }
                    p->t2_wpg = wpg;
                    p->t2_opt = opt;
                    // SXFIR_LDS_PAD: dynamic LDS bytes on top of the kernel's own image: fewer waves fit a CU
                    if (const char *lp = getenv("SXFIR_LDS_PAD")) p->lds_pad = atoi(lp) > 0 ? atoi(lp) : 0;
                    // occ_sb = resident WAVES per CU of this variant
                    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, 64 * wpg, (size_t)p->lds_pad) == hipSuccess && nb > 0)
                        p->occ_sb = nb * wpg;
                }
            }
        }
        if (const char *v = getenv("SXFIR_SCHED")) p->sched = atoi(v);
        if (const char *v = getenv("SXFIR_OCC")) {
            if (atoi(v) > 0) p->occ_sb = p->occ_db = atoi(v);
        }
#endif
    }

    hipError_t e = hipMalloc((void **)&p->taps_dev, sizeof(float) * (size_t)ntaps);
    if (e == hipSuccess) e = hipMalloc((void **)&p->taps_scaled_dev, sizeof(float) * (size_t)ntaps);

    for (int k = 0; k < 64; ++k) p->taps_k[k] = k < ntaps ? (fmt == SXFIR_S32 ? taps[k] * 4.656612873077393e-10f : taps[k]) : 0.0f;
    if (e == hipSuccess) {
        std::vector<float> scaled(taps, taps + ntaps);
        if (mode == SXFIR_DECIMATE && ratio == 8 && ntaps == 256 && (fmt == SXFIR_CF32 || fmt == SXFIR_S32)) {
            // /8 scalar-tap form (decim_dense_kernel<8, ..., SUBSET>): subset s = 2c + p at 64 s, (jj, rr) at 4 jj + rr
            for (int c = 0; c < 2; ++c)
                for (int ph = 0; ph < 2; ++ph)
                    for (int jj = 0; jj < 16; ++jj)
                        for (int rr = 0; rr < 4; ++rr)
                            scaled[(size_t)(64 * (2 * c + ph) + 4 * jj + rr)] =
                                taps[8 * (16 * ph + jj) + 4 * c + rr] * (fmt == SXFIR_S32 ? 4.656612873077393e-10f : 1.0f);   // 2^-31: exact
        } else if (mode == SXFIR_INTERPOLATE && p->itile_capable && ratio == 8) {
            for (int c = 0; c < 2; ++c)
                for (int ph = 0; ph < 2; ++ph)
                    for (int jj = 0; jj < 16; ++jj)
                        for (int rr = 0; rr < 4; ++rr)
                            scaled[(size_t)(64 * (2 * c + ph) + 4 * jj + rr)] = taps[(16 * ph + jj) * 8 + 4 * c + rr];
        } else {
            for (float &t : scaled) t *= 4.656612873077393e-10f;      // 2^-31: exact
        }
        e = hipMemcpy(p->taps_scaled_dev, scaled.data(), sizeof(float) * (size_t)ntaps, hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipMalloc(&p->hist_dev, sample_bytes(fmt) * (size_t)p->hist_len * (size_t)nchan);
    if (e == hipSuccess) e = hipMalloc(&p->hist_alt, sample_bytes(fmt) * (size_t)p->hist_len * (size_t)nchan);
    if (e == hipSuccess) e = hipMemcpy(p->taps_dev, taps, sizeof(float) * (size_t)ntaps, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(p->hist_dev, 0, sample_bytes(fmt) * (size_t)p->hist_len * (size_t)nchan);
    if (e != hipSuccess) {
        if (p->taps_dev) (void)hipFree(p->taps_dev);
        if (p->taps_scaled_dev) (void)hipFree(p->taps_scaled_dev);
        if (p->hist_dev) (void)hipFree(p->hist_dev);
        if (p->hist_alt) (void)hipFree(p->hist_alt);
        delete p;
        return fail(SXFIR_EHIP, "plan allocation failed: %s", hipGetErrorString(e));
    }
    *out = p;
    return SXFIR_OK;
}

int sxfir_destroy(sxfir_plan *p)
{
    if (!p) return SXFIR_OK;
    (void)hipFree(p->taps_dev);
    (void)hipFree(p->taps_scaled_dev);
    (void)hipFree(p->hist_dev);
    (void)hipFree(p->hist_alt);
    delete p;
    return SXFIR_OK;
}

int sxfir_reset(sxfir_plan *p, void *stream)
{
    if (!p) return fail(SXFIR_EINVAL, "plan is NULL");
    HIPCHECK(hipMemsetAsync(p->hist_dev, 0, sample_bytes(p->fmt) * (size_t)p->hist_len * (size_t)p->nchan,
                            S(stream)));
    p->consumed = p->produced = 0;
    return SXFIR_OK;
}

int sxfir_set_history(sxfir_plan *p, const void *src_dev, size_t n, size_t stride, void *stream)
{
    if (!p || !src_dev) return fail(SXFIR_EINVAL, "NULL argument");
    if (n < (size_t)p->hist_len) return fail(SXFIR_EINVAL, "history needs %d samples per channel, %zu given", p->hist_len, n);
    if (p->nchan > 1 && stride < n) return fail(SXFIR_EINVAL, "channel stride %zu shorter than the block (%zu)", stride, n);
    const size_t sb = sample_bytes(p->fmt);
    // the LAST hist_len samples of the block, channel by channel
    const char *src = static_cast<const char *>(src_dev) + sb * (n - (size_t)p->hist_len);
    HIPCHECK(hipMemcpy2DAsync(p->hist_dev, sb * (size_t)p->hist_len, src, sb * stride, sb * (size_t)p->hist_len, (size_t)p->nchan,
                              hipMemcpyDeviceToDevice, S(stream)));
    return SXFIR_OK;
}

int sxfir_set_position(sxfir_plan *p, int64_t consumed)
{
    if (!p) return fail(SXFIR_EINVAL, "plan is NULL");
    if (consumed < 0) return fail(SXFIR_EINVAL, "negative stream position");
    p->consumed = (long long)consumed;
    p->produced = p->mode == SXFIR_DECIMATE ? ((long long)consumed + p->ratio - 1) / p->ratio
                                            : (long long)consumed * p->ratio;
    return SXFIR_OK;
}

int sxfir_set_kernel(sxfir_plan *p, int kernel)
{
    if (!p) return fail(SXFIR_EINVAL, "plan is NULL");
    if (kernel < SXFIR_KERNEL_AUTO || kernel > SXFIR_KERNEL_GENERIC) return fail(SXFIR_EINVAL, "bad kernel id");
    if (kernel == SXFIR_KERNEL_TILED && !p->tile_capable && !p->multi_capable && !p->itile_capable)
        return fail(SXFIR_EUNSUPPORTED, "no tiled kernel for ntaps=%d ratio=%d fmt=%d mode=%d", p->ntaps,
                    p->ratio, p->fmt, p->mode);
    p->kernel = kernel;
    return SXFIR_OK;
}

int sxfir_set_tx_threshold(sxfir_plan *p, float tx_threshold2)
{
    if (!p) return fail(SXFIR_EINVAL, "plan is NULL");
    p->thr2 = tx_threshold2;
    return SXFIR_OK;
}

#ifdef SXFIR_PROFILING
// Diagnostic (SXFIR_ABLATE=11/12 builds): median in-kernel shader clock in MHz of the last launch.
int sxfir_debug_clock(sxfir_plan *p, double *mhz)
{
    if (!p || !mhz || !p->stamps_dev) return fail(SXFIR_EINVAL, "no stamps recorded");
    std::vector<unsigned long long> h(2 * p->stamps_n);
    HIPCHECK(hipMemcpy(h.data(), p->stamps_dev, 16 * p->stamps_n, hipMemcpyDeviceToHost));
    std::vector<double> f;
    for (size_t i = 0; i < p->stamps_n; ++i)
        if (h[2 * i + 1] > 0) f.push_back(100.0 * (double)h[2 * i] / (double)h[2 * i + 1]);
    if (f.empty()) return fail(SXFIR_EINVAL, "no stamps recorded");
    std::sort(f.begin(), f.end());
    *mhz = f[f.size() / 2];
    return SXFIR_OK;
}

int sxfir_debug_stamps(sxfir_plan *p, unsigned long long *host, size_t capacity_records, size_t *n_records)
{
    if (!p || !host || !n_records || !p->stamps_dev || (p->ablate != 3 && p->ablate != 5)) return fail(SXFIR_EINVAL, "no stamps recorded");
    const size_t n = p->stamps_n < capacity_records ? p->stamps_n : capacity_records;
    // records: 5 x uint64 (multi-column kernel, ablate 3) or 8 x uint64 (tile2 kernel, ablate 5)
    HIPCHECK(hipMemcpy(host, p->stamps_dev, (p->ablate == 5 ? 64 : 40) * n, hipMemcpyDeviceToHost));
    *n_records = n;
    return SXFIR_OK;
}

#endif  // SXFIR_PROFILING

int sxfir_contract(const sxfir_plan *p, int *jsplit, int *cw)
{
    if (!p) return fail(SXFIR_EINVAL, "plan is NULL");
    if (jsplit) *jsplit = p->jsplit;
    if (cw) *cw = p->cw;
    return SXFIR_OK;
}

int sxfir_position(const sxfir_plan *p, int64_t *consumed, int64_t *produced)
{
    if (!p) return fail(SXFIR_EINVAL, "plan is NULL");
    if (consumed) *consumed = p->consumed;
    if (produced) *produced = p->produced;
    return SXFIR_OK;
}

static long long outputs_for(const sxfir_plan *p, long long n_in)
{
    if (p->mode == SXFIR_INTERPOLATE) return n_in * p->ratio;
    const long long D = p->ratio;
    const long long before = (p->consumed + D - 1) / D;
    const long long after = (p->consumed + n_in + D - 1) / D;
    return after - before;
}

int sxfir_outputs_for(const sxfir_plan *p, size_t n_in, size_t *n_out)
{
    if (!p || !n_out) return fail(SXFIR_EINVAL, "NULL argument");
    *n_out = (size_t)outputs_for(p, (long long)n_in);
    return SXFIR_OK;
}

// Generic path: the next call's history goes to the plan's other buffer (the caller swaps the two).
static int launch_history(sxfir_plan *p, const void *in_dev, size_t n_in, size_t in_stride, hipStream_t st)
{
    const dim3 grid((unsigned)((p->hist_len + 255) / 256), (unsigned)p->nchan);
    if (p->fmt != SXFIR_CF16)
        hipLaunchKernelGGL(sxfir::history_kernel<float2>, grid, dim3(256), 0, st, (float2 *)p->hist_alt,
                           (const float2 *)p->hist_dev, (const float2 *)in_dev, (long long)n_in, (long long)in_stride,
                           (long long)p->hist_len, p->hist_len);
    else
        hipLaunchKernelGGL(sxfir::history_kernel<uint32_t>, grid, dim3(256), 0, st, (uint32_t *)p->hist_alt,
                           (const uint32_t *)p->hist_dev, (const uint32_t *)in_dev, (long long)n_in,
                           (long long)in_stride, (long long)p->hist_len, p->hist_len);
    HIPCHECK(hipGetLastError());
    return SXFIR_OK;
}

#ifdef SXFIR_PROFILING
#include "sxfir_prof_dispatch.inc"   // the A/B variants' launch tables: 0 = not mine, 1 = launched, < 0 = error
#endif

// Launch only the resampling kernel (no history update, no position change).
static int launch_decim(sxfir_plan *p, const void *in_dev, size_t n_in, size_t in_stride, void *out_dev,
                        size_t out_stride, long long n_out, hipStream_t st, bool *history_done)
{
    *history_done = false;
    const long long D = p->ratio;
    const long long first = ((p->consumed + D - 1) / D) * D - p->consumed;
    // LDS-DMA sources need no 16-byte alignment (verified on MI355X, tools/probe_unaligned.hip): only the
    // output, written with 16-byte stores, must be aligned
    bool tiled = p->tile_capable && p->kernel != SXFIR_KERNEL_GENERIC && first == 0 &&
                 ((uintptr_t)out_dev % 16 == 0) && (p->nchan == 1 || out_stride % 2 == 0);
    const bool multi = p->multi_capable && p->kernel != SXFIR_KERNEL_GENERIC && first == 0 &&
                       ((uintptr_t)out_dev % 16 == 0) &&
                       (p->nchan == 1 || out_stride % (p->fmt == SXFIR_CF16 ? 4 : 2) == 0);
    if (multi) {
        sxfir::DecimMultiArgs a;
        a.in = in_dev;
        a.hist = p->hist_dev;
        a.hist_out = p->hist_alt;
        a.out = out_dev;
        a.taps = p->taps_dev;
        a.n_in = (long long)n_in;
        a.n_out = n_out;
        a.in_stride = (long long)in_stride;
        a.out_stride = (long long)out_stride;
        a.hist_stride = p->hist_len;
        const int W = p->multi_waves;
        const int tile_out = W * 8 * (64 / (p->multi_ps * (p->ratio / 4)));
        const long long n_tiles = (n_out + tile_out - 1) / tile_out;
        if (n_tiles > 0x7fffffffLL) return fail(SXFIR_EINVAL, "call too large");
        long long groups = ((long long)p->compute_units * p->occ_multi * p->oversub) / p->nchan;
        if (groups < 1) groups = 1;
        if (groups > n_tiles) groups = n_tiles;
        a.n_tiles = (int)n_tiles;
        a.n_groups = (int)groups;
        dim3 grid((unsigned)groups, (unsigned)p->nchan);
        a.stamps = nullptr;
        if (p->dense32) {
            // non-temporal staging loads for the image rows no other tile reads (NTLD = 2: both halos stay plain loads),
            // measured in round 4 (profiles/round4h_kbench_both_halos_plain.txt: whole kernel -0.9 % at /32, -2.8 % at /8
            // and /16 against plain loads; with only the next tile's halo plain /32 lost 1.4 %)
#define SXFIR_DENSE_LAUNCH(DD, AA, SS, NN) hipLaunchKernelGGL((sxfir::decim_dense_kernel<DD, AA, SS, NN>), grid, dim3(256), 0, st, a)
#ifdef SXFIR_PROFILING
            if (const int pr = prof_launch_dense(p, a, grid, st, groups, W)) {       // ablations, stamps, nt-load A/B
                if (pr < 0) return pr;
                *history_done = true;
                return SXFIR_OK;
            }
#endif
            if (p->dense_subset) {
                a.taps = p->taps_scaled_dev;                      // the subset-major tap table
                if (p->fmt == SXFIR_S32) hipLaunchKernelGGL((sxfir::decim_dense_kernel<8, 0, true, 2, true>), grid, dim3(256), 0, st, a);
                else hipLaunchKernelGGL((sxfir::decim_dense_kernel<8, 0, false, 2, true>), grid, dim3(256), 0, st, a);
            }
#ifdef SXFIR_PROFILING
            else if (p->ratio == 8 && p->fmt == SXFIR_S32) SXFIR_DENSE_LAUNCH(8, 0, true, 2);   // the VGPR-tap forms at /8: A/B partners only
            else if (p->ratio == 8) SXFIR_DENSE_LAUNCH(8, 0, false, 2);
#endif
            else if (p->fmt == SXFIR_S32 && p->ratio == 16) SXFIR_DENSE_LAUNCH(16, 0, true, 2);
            else if (p->fmt == SXFIR_S32) SXFIR_DENSE_LAUNCH(32, 0, true, 2);
            else if (p->ratio == 16) SXFIR_DENSE_LAUNCH(16, 0, false, 2);
            else SXFIR_DENSE_LAUNCH(32, 0, false, 2);
#undef SXFIR_DENSE_LAUNCH
            HIPCHECK(hipGetLastError());
            *history_done = true;
            return SXFIR_OK;
        }
#ifdef SXFIR_PROFILING
        if (const int pr = prof_launch_multi(p, a, grid, st, groups, W)) {           // S32 words, ablations, stamps
            if (pr < 0) return pr;
            *history_done = true;
            return SXFIR_OK;
        }
#endif
        const int key = SXFIR_MULTI_KEY(p->ratio, W, p->fmt == SXFIR_CF16, p->multi_ps);
        switch (key) {
#define SXFIR_X(DD, WW, HH, PP) \
        case SXFIR_MULTI_KEY(DD, WW, HH, PP): \
            hipLaunchKernelGGL((sxfir::decim_multi_kernel<DD, WW, HH, 0, PP>), grid, dim3(64 * WW), 0, st, a); \
            break;
            SXFIR_MULTI_VARIANTS(SXFIR_X)
#undef SXFIR_X
        default: return fail(SXFIR_EUNSUPPORTED, "no multi kernel for ratio %d with %d waves (mode %d)", p->ratio, W, key);
        }
        HIPCHECK(hipGetLastError());
        *history_done = true;
        return SXFIR_OK;
    }
    if (p->kernel == SXFIR_KERNEL_TILED && !tiled)
        return fail(SXFIR_EUNSUPPORTED,
                    "tiled kernel needs a 16-byte aligned output, an even output stride and a call that starts on "
                    "an output boundary");
    if (tiled) {
        sxfir::DecimTileArgs a;
        a.long_waves = a.long_tiles = a.long_w8 = a.short_w8 = 0;
        a.in = (const float *)in_dev;
        a.hist = (const float *)p->hist_dev;
        a.hist_out = (float *)p->hist_alt;
        a.out = (float *)out_dev;
        *history_done = true;      // caller swaps hist_dev / hist_alt when it commits the call
        a.taps = p->taps_dev;
        a.taps_scaled = p->taps_scaled_dev;
        memcpy(a.taps_k, p->taps_k, sizeof(a.taps_k));
        a.n_in = (long long)n_in;
        a.n_out = n_out;
        a.in_stride = (long long)in_stride;
        a.out_stride = (long long)out_stride;
        a.hist_stride = p->hist_len;
        int tile_out = 256;
#ifdef SXFIR_PROFILING
        if (p->sgpr_r && p->ntaps == 128) tile_out = 64 * p->sgpr_r;
#endif
        const long long n_tiles = (n_out + tile_out - 1) / tile_out;
        if (n_tiles > 0x7fffffffLL) return fail(SXFIR_EINVAL, "call too large");
        a.n_tiles = (int)n_tiles;
        a.sched = p->sched;
        a.stamps = nullptr;
#ifdef SXFIR_PROFILING
        if (const int pr = prof_launch_tile_variant(p, a, n_out, n_tiles, st)) return pr < 0 ? pr : SXFIR_OK;   // pair / wide / tile2 variants
#endif
        if (p->wide8 && p->sched != 1) {
            // 128 symmetric taps: decim4_wide_kernel, tiles of 512 outputs, one wave (= one workgroup) per tile and pass;
            // G = CUs x 8 resident waves x 16 generations waves per launch, strided XCD-blocked passes
            const long long n_tiles2 = (n_out + 511) / 512;
            long long G = ((long long)p->compute_units * p->occ_wide * p->oversub) / p->nchan;
            if (G < 1) G = 1;
            if (G > n_tiles2) G = n_tiles2;
            a.n_tiles = (int)n_tiles2;
            a.n_waves = (int)G;
            a.w8 = (G % 8 == 0) ? (int)(G / 8) : 0;
            a.run_base = a.run_extra = 0;
            {
                const int t = (int)((n_tiles2 - 1) % G);
                a.hist_wave = (p->sched == 0 && a.w8) ? (t % a.w8) * 8 + t / a.w8 : t;
            }
            dim3 grid((unsigned)G, (unsigned)p->nchan);
            if (p->fmt == SXFIR_S32) hipLaunchKernelGGL((sxfir::decim4_wide_kernel<0, true>), grid, dim3(64), 0, st, a);
            else hipLaunchKernelGGL((sxfir::decim4_wide_kernel<0, false>), grid, dim3(64), 0, st, a);
            HIPCHECK(hipGetLastError());
            return SXFIR_OK;
        }
        // Short-lived waves in generations: W = CUs * resident waves * oversub waves per launch, each covering
        // n_tiles / W tiles in strided, XCD-blocked passes (sxfir_decim_tile.hip.h).
        const bool dbuf = p->tile_dbuf;
        long long per_chan = ((long long)p->compute_units * (dbuf ? p->occ_db : p->occ_sb) * p->oversub) / p->nchan;
        if (per_chan < 1) per_chan = 1;
        if (per_chan > n_tiles) per_chan = n_tiles;
        a.n_waves = (int)per_chan;
        {
            const int W = (int)per_chan, last = (int)n_tiles - 1;
            a.w8 = (W % 8 == 0) ? W / 8 : 0;
            a.run_base = (int)(n_tiles / W);
            a.run_extra = (int)(n_tiles % W);
            if (p->sched == 1) {
                a.hist_wave = a.run_base >= 1 ? W - 1 : last;           // owner of the last contiguous run
            } else {
                const int t = last % W;                                  // first tile of the owner's sequence
                a.hist_wave = (p->sched == 0 && a.w8) ? (t % a.w8) * 8 + t / a.w8 : t;
            }
        }
#ifdef SXFIR_PROFILING
        prof_short_tail(p, a, n_tiles, &per_chan);                                    // SXFIR_SCHED=3
#endif
        dim3 grid((unsigned)per_chan, (unsigned)p->nchan);
#ifdef SXFIR_PROFILING
        if (const int pr = prof_launch_tile_first_gen(p, a, grid, per_chan, dbuf, st)) return pr < 0 ? pr : SXFIR_OK;
#endif
#ifdef SXFIR_PROFILING
        // "t2s": round 3's shipped form (with one wave per workgroup both kernels take the same schedule constants)
        if (p->ntaps == 128 && p->symmetric && p->sched != 1) {
            if (p->fmt == SXFIR_S32)
                hipLaunchKernelGGL((sxfir::decim4_tile2_kernel<128, 1, sxfir::T2_SHIPPED, 0, true>), grid, dim3(64), 0, st, a);
            else
                hipLaunchKernelGGL((sxfir::decim4_tile2_kernel<128, 1, sxfir::T2_SHIPPED>), grid, dim3(64), 0, st, a);
        } else
#endif
        if (p->fmt == SXFIR_S32)
            hipLaunchKernelGGL((sxfir::decim4_tile_kernel<128, false, 0, true>), grid, dim3(64), 0, st, a);
        else if (p->ntaps == 128)
            hipLaunchKernelGGL((sxfir::decim4_tile_kernel<128, false>), grid, dim3(64), 0, st, a);
        else
            hipLaunchKernelGGL((sxfir::decim4_tile_kernel<64, false>), grid, dim3(64), 0, st, a);
    } else {
        sxfir::GenericArgs a;
        a.in = in_dev;
        a.hist = p->hist_dev;
        a.out = out_dev;
        a.taps = p->taps_dev;
        a.n_in = (long long)n_in;
        a.n_out = n_out;
        a.in_stride = (long long)in_stride;
        a.out_stride = (long long)out_stride;
        a.hist_stride = p->hist_len;
        a.first = first;
        a.ntaps = p->ntaps;
        a.ratio = p->ratio;
        a.hist_len = p->hist_len;
        a.jsplit = p->jsplit;
        a.cw = p->cw;
        dim3 grid((unsigned)((n_out + 255) / 256), (unsigned)p->nchan);
        a.thr2 = p->thr2;
        if (p->fmt == SXFIR_CF32)
            hipLaunchKernelGGL(sxfir::decim_generic_kernel<sxfir::CF32>, grid, dim3(256), 0, st, a);
        else if (p->fmt == SXFIR_CF16)
            hipLaunchKernelGGL(sxfir::decim_generic_kernel<sxfir::CF16>, grid, dim3(256), 0, st, a);
        else
            hipLaunchKernelGGL((sxfir::decim_generic_kernel<sxfir::S32, sxfir::CF32>), grid, dim3(256), 0, st, a);
    }
    HIPCHECK(hipGetLastError());
    return SXFIR_OK;
}

static int check_io(const sxfir_plan *p, int mode, const void *in_dev, size_t n_in, size_t in_stride,
                    const void *out_dev, size_t out_stride, long long n_out)
{
    if (!p) return fail(SXFIR_EINVAL, "plan is NULL");
    if (p->mode != mode) return fail(SXFIR_EINVAL, "plan was created for the other direction");
    if ((n_in && !in_dev) || (n_out > 0 && !out_dev)) return fail(SXFIR_EINVAL, "NULL device buffer");
    if (p->nchan > 1 && (in_stride < n_in || out_stride < (size_t)n_out))
        return fail(SXFIR_EINVAL, "channel stride smaller than the block");
    if ((uintptr_t)in_dev % sample_bytes(p->fmt) || (uintptr_t)out_dev % sample_bytes(p->fmt))
        return fail(SXFIR_EINVAL, "buffers must be aligned to one complex sample");
    return SXFIR_OK;
}

int sxfir_decimate(sxfir_plan *p, const void *in_dev, size_t n_in, size_t in_stride, void *out_dev,
                   size_t out_stride, size_t *n_out_p, void *stream)
{
    if (n_out_p) *n_out_p = 0;
    if (!p) return fail(SXFIR_EINVAL, "plan is NULL");
    const long long n_out = outputs_for(p, (long long)n_in);
    int rc = check_io(p, SXFIR_DECIMATE, in_dev, n_in, in_stride, out_dev, out_stride, n_out);
    if (rc) return rc;
    if (n_in == 0) return SXFIR_OK;
    HIPCHECK(hipSetDevice(p->device));
    bool history_done = false;
    if (n_out > 0) {
        rc = launch_decim(p, in_dev, n_in, in_stride, out_dev, out_stride, n_out, S(stream), &history_done);
        if (rc) return rc;
    }
    if (!history_done) {
        rc = launch_history(p, in_dev, n_in, in_stride, S(stream));
        if (rc) return rc;
    }
    std::swap(p->hist_dev, p->hist_alt);
    p->consumed += (long long)n_in;
    p->produced += n_out;
    if (n_out_p) *n_out_p = (size_t)n_out;
    return SXFIR_OK;
}

// Launch only the interpolation kernel (no history swap, no position change).
// key: count the input samples [lo, hi) of channel 0 that reach the plan's keying threshold into *counter
struct KeyedRange { unsigned long long *counter; long long lo, hi; };
static int launch_interp(sxfir_plan *p, const void *in_dev, size_t n_in, size_t in_stride, void *out_dev,
                         size_t out_stride, long long n_out, hipStream_t st, bool *history_done,
                         const KeyedRange *key = nullptr)
{
    *history_done = false;
    const bool tiled = p->itile_capable && p->kernel != SXFIR_KERNEL_GENERIC && ((uintptr_t)out_dev % 16 == 0) &&
                       (p->nchan == 1 || out_stride % 2 == 0);
    if (p->kernel == SXFIR_KERNEL_TILED && !tiled)
        return fail(SXFIR_EUNSUPPORTED, "tiled interpolator needs a 16-byte aligned output and even strides");
    if (tiled && p->ipass) {
        // x8, 256 taps, CF32: the scalar-tap form, tiles of 128 inputs (two per lane), four (phase group, row half) passes per tile
        sxfir::InterpTileArgs t;
        t.in = (const float *)in_dev;
        t.hist = (const float *)p->hist_dev;
        t.hist_out = (float *)p->hist_alt;
        t.out = (float *)out_dev;
        t.taps = p->taps_scaled_dev;                            // the pass-major table
        t.n_in = (long long)n_in;
        t.in_stride = (long long)in_stride;
        t.out_stride = (long long)out_stride;
        t.hist_stride = p->hist_len;
        const int tile_in = 64 * p->ipass_qi;
        const long long n_tiles = ((long long)n_in + tile_in - 1) / tile_in;
        if (n_tiles > 0x7fffffffLL) return fail(SXFIR_EINVAL, "call too large");
        long long groups = ((long long)p->compute_units * p->occ_ipass * p->oversub) / p->nchan;
        if (groups < 1) groups = 1;
        if (groups > n_tiles) groups = n_tiles;
        t.n_tiles = (int)n_tiles;
        t.n_groups = (int)groups;
        t.thr2 = p->thr2;
        t.key_counter = key ? key->counter : nullptr;
        t.key_lo = key ? key->lo : 0;
        t.key_hi = key ? key->hi : 0;
        const dim3 pgrid((unsigned)groups, (unsigned)p->nchan);
#ifdef SXFIR_PROFILING
        if (p->ipass_qi == 4 && p->fmt == SXFIR_S32) return fail(SXFIR_EUNSUPPORTED, "four inputs per lane: CF32 only");
        else if (p->ipass_qi == 4 && key) hipLaunchKernelGGL((sxfir::interp8_pass_kernel<4, true>), pgrid, dim3(64), 0, st, t);
        else if (p->ipass_qi == 4) hipLaunchKernelGGL((sxfir::interp8_pass_kernel<4>), pgrid, dim3(64), 0, st, t);
        else
#endif
        if (p->fmt == SXFIR_S32 && key) hipLaunchKernelGGL((sxfir::interp8_pass_kernel<2, true, true>), pgrid, dim3(64), 0, st, t);
        else if (p->fmt == SXFIR_S32) hipLaunchKernelGGL((sxfir::interp8_pass_kernel<2, false, true>), pgrid, dim3(64), 0, st, t);
        else if (key) hipLaunchKernelGGL((sxfir::interp8_pass_kernel<2, true>), pgrid, dim3(64), 0, st, t);
        else hipLaunchKernelGGL((sxfir::interp8_pass_kernel<2>), pgrid, dim3(64), 0, st, t);
        HIPCHECK(hipGetLastError());
        *history_done = true;
        return SXFIR_OK;
    }
    if (tiled) {
        sxfir::InterpTileArgs t;
        t.in = (const float *)in_dev;
        t.hist = (const float *)p->hist_dev;
        t.hist_out = (float *)p->hist_alt;
        t.out = (float *)out_dev;
        t.taps = p->taps_dev;
        t.n_in = (long long)n_in;
        t.in_stride = (long long)in_stride;
        t.out_stride = (long long)out_stride;
        t.hist_stride = p->hist_len;
        const int qt = 4 * 4 * (32 / (p->ratio / 4));          // InterpTile<L>::TILE_IN
        const long long n_tiles = ((long long)n_in + qt - 1) / qt;
        if (n_tiles > 0x7fffffffLL) return fail(SXFIR_EINVAL, "call too large");
        long long groups = ((long long)p->compute_units * 16 * p->oversub) / p->nchan;
        if (groups < 1) groups = 1;
        if (groups > n_tiles) groups = n_tiles;
        t.n_tiles = (int)n_tiles;
        t.n_groups = (int)groups;
        t.thr2 = p->thr2;
        t.key_counter = key ? key->counter : nullptr;
        t.key_lo = key ? key->lo : 0;
        t.key_hi = key ? key->hi : 0;
        dim3 grid((unsigned)groups, (unsigned)p->nchan);
        if (key && p->fmt == SXFIR_S32) {
            switch (p->ratio) {
            case 4: hipLaunchKernelGGL((sxfir::interp_tile_kernel<4, true, true>), grid, dim3(64), 0, st, t); break;
#ifdef SXFIR_PROFILING
            case 8: hipLaunchKernelGGL((sxfir::interp_tile_kernel<8, true, true>), grid, dim3(64), 0, st, t); break;
#else
            case 8: return fail(SXFIR_EUNSUPPORTED, "x8 runs interp8_pass_kernel");   // (unreachable: p->ipass)
#endif
            case 16: hipLaunchKernelGGL((sxfir::interp_tile_kernel<16, true, true>), grid, dim3(64), 0, st, t); break;
            default: hipLaunchKernelGGL((sxfir::interp_tile_kernel<32, true, true>), grid, dim3(64), 0, st, t); break;
            }
        } else if (key) {
            switch (p->ratio) {
            case 4: hipLaunchKernelGGL((sxfir::interp_tile_kernel<4, false, true>), grid, dim3(64), 0, st, t); break;
#ifdef SXFIR_PROFILING
            case 8: hipLaunchKernelGGL((sxfir::interp_tile_kernel<8, false, true>), grid, dim3(64), 0, st, t); break;
#else
            case 8: return fail(SXFIR_EUNSUPPORTED, "x8 runs interp8_pass_kernel");   // (unreachable: p->ipass)
#endif
            case 16: hipLaunchKernelGGL((sxfir::interp_tile_kernel<16, false, true>), grid, dim3(64), 0, st, t); break;
            default: hipLaunchKernelGGL((sxfir::interp_tile_kernel<32, false, true>), grid, dim3(64), 0, st, t); break;
            }
        } else if (p->fmt == SXFIR_S32) {
            switch (p->ratio) {
            case 4: hipLaunchKernelGGL((sxfir::interp_tile_kernel<4, true>), grid, dim3(64), 0, st, t); break;
#ifdef SXFIR_PROFILING
            case 8: hipLaunchKernelGGL((sxfir::interp_tile_kernel<8, true>), grid, dim3(64), 0, st, t); break;
#else
            case 8: return fail(SXFIR_EUNSUPPORTED, "x8 runs interp8_pass_kernel");   // (unreachable: p->ipass)
#endif
            case 16: hipLaunchKernelGGL((sxfir::interp_tile_kernel<16, true>), grid, dim3(64), 0, st, t); break;
            default: hipLaunchKernelGGL((sxfir::interp_tile_kernel<32, true>), grid, dim3(64), 0, st, t); break;
            }
        } else {
            switch (p->ratio) {
            case 4: hipLaunchKernelGGL((sxfir::interp_tile_kernel<4>), grid, dim3(64), 0, st, t); break;
#ifdef SXFIR_PROFILING
            case 8: hipLaunchKernelGGL((sxfir::interp_tile_kernel<8>), grid, dim3(64), 0, st, t); break;
#else
            case 8: return fail(SXFIR_EUNSUPPORTED, "x8 runs interp8_pass_kernel");   // (unreachable: p->ipass)
#endif
            case 16: hipLaunchKernelGGL((sxfir::interp_tile_kernel<16>), grid, dim3(64), 0, st, t); break;
            default: hipLaunchKernelGGL((sxfir::interp_tile_kernel<32>), grid, dim3(64), 0, st, t); break;
            }
        }
        HIPCHECK(hipGetLastError());
        *history_done = true;
        return SXFIR_OK;
    }
    sxfir::GenericArgs a;
    a.in = in_dev;
    a.hist = p->hist_dev;
    a.out = out_dev;
    a.taps = p->taps_dev;
    a.n_in = (long long)n_in;
    a.n_out = n_out;
    a.in_stride = (long long)in_stride;
    a.out_stride = (long long)out_stride;
    a.hist_stride = p->hist_len;
    a.first = 0;
    a.ntaps = p->ntaps;
    a.ratio = p->ratio;
    a.hist_len = p->hist_len;
    a.jsplit = p->jsplit;
    a.cw = p->cw;
    dim3 grid((unsigned)((n_out + 255) / 256), (unsigned)p->nchan);
    a.thr2 = p->thr2;
    if (p->fmt == SXFIR_CF32)
        hipLaunchKernelGGL(sxfir::interp_generic_kernel<sxfir::CF32>, grid, dim3(256), 0, st, a);
    else if (p->fmt == SXFIR_CF16)
        hipLaunchKernelGGL(sxfir::interp_generic_kernel<sxfir::CF16>, grid, dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((sxfir::interp_generic_kernel<sxfir::CF32, sxfir::S32>), grid, dim3(256), 0, st, a);
    HIPCHECK(hipGetLastError());
    if (key && key->hi > key->lo) {
        // shapes the tiled kernel does not take: the count as a pass of its own (same rule, same counter)
        const long long n = key->hi - key->lo;
        unsigned g = (unsigned)std::min<long long>((n + 255) / 256, 256);
        hipLaunchKernelGGL(sxfir::count_keyed_kernel, dim3(g), dim3(256), 0, st,
                           reinterpret_cast<const float2 *>(in_dev) + key->lo, n, p->thr2, key->counter);
        HIPCHECK(hipGetLastError());
    }
    return SXFIR_OK;
}

static int interpolate_impl(sxfir_plan *p, const void *in_dev, size_t n_in, size_t in_stride, void *out_dev,
                            size_t out_stride, size_t *n_out_p, void *stream, const KeyedRange *key)
{
    if (n_out_p) *n_out_p = 0;
    if (!p) return fail(SXFIR_EINVAL, "plan is NULL");
    const long long n_out = outputs_for(p, (long long)n_in);
    int rc = check_io(p, SXFIR_INTERPOLATE, in_dev, n_in, in_stride, out_dev, out_stride, n_out);
    if (rc) return rc;
    if (n_in == 0) return SXFIR_OK;
    HIPCHECK(hipSetDevice(p->device));
    bool history_done = false;
    rc = launch_interp(p, in_dev, n_in, in_stride, out_dev, out_stride, n_out, S(stream), &history_done, key);
    if (rc) return rc;
    if (!history_done) {
        rc = launch_history(p, in_dev, n_in, in_stride, S(stream));
        if (rc) return rc;
    }
    std::swap(p->hist_dev, p->hist_alt);
    p->consumed += (long long)n_in;
    p->produced += n_out;
    if (n_out_p) *n_out_p = (size_t)n_out;
    return SXFIR_OK;
}

int sxfir_interpolate(sxfir_plan *p, const void *in_dev, size_t n_in, size_t in_stride, void *out_dev,
                      size_t out_stride, size_t *n_out_p, void *stream)
{
    return interpolate_impl(p, in_dev, n_in, in_stride, out_dev, out_stride, n_out_p, stream, nullptr);
}

int sxfir_interpolate_keyed(sxfir_plan *p, const void *in_dev, size_t n_in, size_t in_stride, void *out_dev,
                            size_t out_stride, size_t *n_out_p, size_t key_first, size_t key_count,
                            unsigned long long *counter, void *stream)
{
    if (n_out_p) *n_out_p = 0;
    if (!p) return fail(SXFIR_EINVAL, "plan is NULL");
    if (p->mode != SXFIR_INTERPOLATE) return fail(SXFIR_EINVAL, "not an interpolator plan");
    if (p->fmt == SXFIR_CF16) return fail(SXFIR_EUNSUPPORTED, "the keying count is defined on CF32 input");
    if (!counter || ((uintptr_t)counter & 7)) return fail(SXFIR_EINVAL, "counter must be an 8-byte aligned device word");
    if (key_first > n_in || key_count > n_in - key_first) return fail(SXFIR_EINVAL, "keying range outside the block");
    const KeyedRange key{counter, (long long)key_first, (long long)(key_first + key_count)};
    return interpolate_impl(p, in_dev, n_in, in_stride, out_dev, out_stride, n_out_p, stream, key_count ? &key : nullptr);
}

// Timed launches (bench.py): `iters` back-to-back passes of the resampling kernel over the same buffers and
// from the same filter state, bracketed by HIP events on the launch stream.
static int time_passes(sxfir_plan *p, int mode, const void *in_dev, size_t n_in, size_t in_stride, void *out_dev,
                       size_t out_stride, int iters, void *stream, float *ms_per_pass)
{
    if (!p || !ms_per_pass || iters < 1) return fail(SXFIR_EINVAL, "bad argument");
    const long long n_out = outputs_for(p, (long long)n_in);
    int rc = check_io(p, mode, in_dev, n_in, in_stride, out_dev, out_stride, n_out);
    if (rc) return rc;
    if (n_out < 1) return fail(SXFIR_EINVAL, "nothing to do");
    HIPCHECK(hipSetDevice(p->device));
    hipEvent_t e0, e1;
    HIPCHECK(hipEventCreate(&e0));
    HIPCHECK(hipEventCreate(&e1));
    HIPCHECK(hipEventRecord(e0, S(stream)));
    for (int i = 0; i < iters; ++i) {
        bool history_done = false;   // history buffers are not swapped: every pass filters from the same state
        rc = mode == SXFIR_DECIMATE
                 ? launch_decim(p, in_dev, n_in, in_stride, out_dev, out_stride, n_out, S(stream), &history_done)
                 : launch_interp(p, in_dev, n_in, in_stride, out_dev, out_stride, n_out, S(stream), &history_done);
        if (rc) break;
    }
    hipError_t e = hipEventRecord(e1, S(stream));
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.0f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc) return rc;
    if (e != hipSuccess) return fail(SXFIR_EHIP, "event timing failed: %s", hipGetErrorString(e));
    *ms_per_pass = ms / (float)iters;
    return SXFIR_OK;
}

int sxfir_time_decimate(sxfir_plan *p, const void *in_dev, size_t n_in, size_t in_stride, void *out_dev,
                        size_t out_stride, int iters, void *stream, float *ms_per_pass)
{
    return time_passes(p, SXFIR_DECIMATE, in_dev, n_in, in_stride, out_dev, out_stride, iters, stream, ms_per_pass);
}

int sxfir_time_interpolate(sxfir_plan *p, const void *in_dev, size_t n_in, size_t in_stride, void *out_dev,
                           size_t out_stride, int iters, void *stream, float *ms_per_pass)
{
    return time_passes(p, SXFIR_INTERPOLATE, in_dev, n_in, in_stride, out_dev, out_stride, iters, stream, ms_per_pass);
}

// In-kernel shader clock while other work runs: a few single-wave workgroups on a stream of their own spin on
// s_memtime (shader cycles) against s_memrealtime (100 MHz) for `duration_us`; sxfir_clock_probe_read waits for
// them and returns the median ratio.  They use one wave slot each and no LDS, so they sit beside a running
// resampling kernel (bench.py: roofline.shader_mhz, the clock the chip's power management holds under it).
struct sxfir_clock_probe {
    hipStream_t stream;
    unsigned long long *dev;
    int n;
};

__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long *out, unsigned long long ticks)
{
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < ticks) {
        __builtin_amdgcn_s_sleep(32);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = c1 - c0;
        out[2 * blockIdx.x + 1] = r1 - r0;
    }
}

int sxfir_clock_probe_start(sxfir_clock_probe **probe, int device, int duration_us)
{
    if (!probe || duration_us < 1 || duration_us > 10000000) return fail(SXFIR_EINVAL, "bad argument");
    *probe = nullptr;
    if (device >= 0) HIPCHECK(hipSetDevice(device));
    sxfir_clock_probe *q = new (std::nothrow) sxfir_clock_probe();
    if (!q) return fail(SXFIR_ENOMEM, "out of host memory");
    q->n = 16;
    q->dev = nullptr;
    q->stream = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&q->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void **)&q->dev, 16 * q->n);
    if (e == hipSuccess) e = hipMemsetAsync(q->dev, 0, 16 * q->n, q->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(clock_probe_kernel, dim3(q->n), dim3(64), 0, q->stream, q->dev, 100ull * (unsigned long long)duration_us);
        e = hipGetLastError();
    }
    if (e != hipSuccess) {
        if (q->dev) (void)hipFree(q->dev);
        if (q->stream) (void)hipStreamDestroy(q->stream);
        delete q;
        return fail(SXFIR_EHIP, "clock probe: %s", hipGetErrorString(e));
    }
    *probe = q;
    return SXFIR_OK;
}

int sxfir_clock_probe_read(sxfir_clock_probe *q, double *mhz)
{
    if (!q || !mhz) return fail(SXFIR_EINVAL, "NULL argument");
    std::vector<unsigned long long> h(2 * (size_t)q->n);
    hipError_t e = hipStreamSynchronize(q->stream);
    if (e == hipSuccess) e = hipMemcpy(h.data(), q->dev, 16 * q->n, hipMemcpyDeviceToHost);
    (void)hipFree(q->dev);
    (void)hipStreamDestroy(q->stream);
    const int n = q->n;
    delete q;
    if (e != hipSuccess) return fail(SXFIR_EHIP, "clock probe: %s", hipGetErrorString(e));
    std::vector<double> f;
    for (int i = 0; i < n; ++i)
        if (h[2 * i + 1] > 0) f.push_back(100.0 * (double)h[2 * i] / (double)h[2 * i + 1]);
    if (f.empty()) return fail(SXFIR_EHIP, "clock probe recorded nothing");
    std::sort(f.begin(), f.end());
    *mhz = f[f.size() / 2];
    return SXFIR_OK;
}

int sxfir_synth_fill(void *out_dev, size_t n, size_t stride, int nchan, uint64_t seed, uint32_t first_channel,
                     int64_t start, int fmt, void *stream)
{
    if (!out_dev && n) return fail(SXFIR_EINVAL, "NULL buffer");
    if (nchan < 1) return fail(SXFIR_EINVAL, "nchan < 1");
    if (n == 0) return SXFIR_OK;
    unsigned bx = (unsigned)((n + 255) / 256);
    if (bx > 16384) bx = 16384;
    dim3 grid(bx, (unsigned)nchan);
    if (fmt == SXFIR_CF32)
        hipLaunchKernelGGL(sxfir::synth_kernel<sxfir::CF32>, grid, dim3(256), 0, S(stream), out_dev, (long long)n,
                           (long long)stride, seed, first_channel, (long long)start);
    else if (fmt == SXFIR_CF16)
        hipLaunchKernelGGL(sxfir::synth_kernel<sxfir::CF16>, grid, dim3(256), 0, S(stream), out_dev, (long long)n,
                           (long long)stride, seed, first_channel, (long long)start);
    else if (fmt == SXFIR_S32)
        hipLaunchKernelGGL(sxfir::synth_s32_kernel, grid, dim3(256), 0, S(stream), (int2 *)out_dev, (long long)n,
                           (long long)stride, seed, first_channel, (long long)start);
    else
        return fail(SXFIR_EINVAL, "bad format");
    HIPCHECK(hipGetLastError());
    return SXFIR_OK;
}

static unsigned stream_grid(size_t n)
{
    size_t b = (n + 255) / 256;
    return (unsigned)(b > 8192 ? 8192 : (b ? b : 1));
}

int sxfir_convert_rx_s32(const int32_t *src, float *dst, size_t n, void *stream)
{
    if (n == 0) return SXFIR_OK;
    if (!src || !dst) return fail(SXFIR_EINVAL, "NULL buffer");
    hipLaunchKernelGGL(sxfir::convert_rx_kernel, dim3(stream_grid(n)), dim3(256), 0, S(stream), (const int2 *)src,
                       (float2 *)dst, (long long)n);
    HIPCHECK(hipGetLastError());
    return SXFIR_OK;
}

int sxfir_convert_tx_s32(const float *src, int32_t *dst, size_t n, float thr2, void *stream)
{
    if (n == 0) return SXFIR_OK;
    if (!src || !dst) return fail(SXFIR_EINVAL, "NULL buffer");
    hipLaunchKernelGGL(sxfir::convert_tx_kernel, dim3(stream_grid(n)), dim3(256), 0, S(stream),
                       (const float2 *)src, (int2 *)dst, (long long)n, thr2);
    HIPCHECK(hipGetLastError());
    return SXFIR_OK;
}

int sxfir_count_keyed(const float *src, size_t n, float thr2, unsigned long long *counter, void *stream)
{
    if (n == 0) return SXFIR_OK;
    if (!src || !counter) return fail(SXFIR_EINVAL, "NULL buffer");
    hipLaunchKernelGGL(sxfir::count_keyed_kernel, dim3(stream_grid(n) > 256 ? 256 : stream_grid(n)), dim3(256), 0, S(stream),
                       (const float2 *)src, (long long)n, thr2, counter);
    HIPCHECK(hipGetLastError());
    return SXFIR_OK;
}

int sxfir_cf32_to_cf16(const float *src, void *dst, size_t n, void *stream)
{
    if (n == 0) return SXFIR_OK;
    if (!src || !dst) return fail(SXFIR_EINVAL, "NULL buffer");
    hipLaunchKernelGGL(sxfir::cf32_to_cf16_kernel, dim3(stream_grid(n)), dim3(256), 0, S(stream),
                       (const float2 *)src, (__half2 *)dst, (long long)n);
    HIPCHECK(hipGetLastError());
    return SXFIR_OK;
}

int sxfir_cf16_to_cf32(const void *src, float *dst, size_t n, void *stream)
{
    if (n == 0) return SXFIR_OK;
    if (!src || !dst) return fail(SXFIR_EINVAL, "NULL buffer");
    hipLaunchKernelGGL(sxfir::cf16_to_cf32_kernel, dim3(stream_grid(n)), dim3(256), 0, S(stream),
                       (const __half2 *)src, (float2 *)dst, (long long)n);
    HIPCHECK(hipGetLastError());
    return SXFIR_OK;
}

// SoapySDR::ticksToTimeNs / timeNsToTicks as used by SoapySX.cpp:562-571
// (SoapySDR lib/TimeC.cpp): whole seconds in integers, remainder in double.
long long sxfir_ticks_to_time_ns(long long ticks, double rate)
{
    const long long ratell = (long long)rate;
    const long long full = ticks / ratell;
    const long long err = ticks - full * ratell;
    const double part = (double)full * (rate - (double)ratell);
    const double frac = (((double)err - part) * 1e9) / rate;
    return full * 1000000000LL + std::llround(frac);
}

long long sxfir_time_ns_to_ticks(long long time_ns, double rate)
{
    const long long ratell = (long long)rate;
    const long long full = time_ns / 1000000000LL;
    const long long err = time_ns - full * 1000000000LL;
    const double part = (double)full * (rate - (double)ratell);
    const double frac = part + ((double)err * rate) / 1e9;
    return full * ratell + std::llround(frac);
}

static double i0(double x)
{
    double sum = 1.0, term = 1.0;
    const double q = x * x / 4.0;
    for (int k = 1; k < 500; ++k) {
        term *= q / ((double)k * k);
        sum += term;
        if (term < sum * 1e-18) break;
    }
    return sum;
}

int sxfir_design_lowpass(int ntaps, int ratio, double beta, double gain, float *taps)
{
    if (ntaps < 1 || ratio < 1 || !taps) return fail(SXFIR_EINVAL, "bad argument");
    std::vector<double> h((size_t)ntaps);
    const double pi = 3.14159265358979323846;
    const double centre = (ntaps - 1) / 2.0;
    const double den = i0(beta);
    double total = 0.0;
    for (int k = 0; k < ntaps; ++k) {
        const double t = k - centre;
        const double arg = t / ratio;                       // 2*fc*t with fc = 0.5/ratio
        const double sinc = (t == 0.0) ? 1.0 : std::sin(pi * arg) / (pi * arg);
        double u = (centre > 0.0) ? t / centre : 0.0;
        u = 1.0 - u * u;
        const double win = i0(beta * std::sqrt(u > 0.0 ? u : 0.0)) / den;
        h[(size_t)k] = sinc * win / ratio;
        total += h[(size_t)k];
    }
    for (int k = 0; k < ntaps; ++k) taps[k] = (float)(h[(size_t)k] * (gain / total));
    return SXFIR_OK;
}

int sxfir_malloc(void **dev, size_t bytes)
{
    if (!dev) return fail(SXFIR_EINVAL, "NULL argument");
    *dev = nullptr;
    hipError_t e = hipMalloc(dev, bytes ? bytes : 1);
    if (e == hipErrorOutOfMemory) return fail(SXFIR_ENOMEM, "hipMalloc(%zu) out of memory", bytes);
    if (e != hipSuccess) return fail(SXFIR_EHIP, "hipMalloc: %s", hipGetErrorString(e));
    return SXFIR_OK;
}

int sxfir_free(void *dev)
{
    if (dev) HIPCHECK(hipFree(dev));
    return SXFIR_OK;
}

int sxfir_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream)
{
    HIPCHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, S(stream)));
    return SXFIR_OK;
}

int sxfir_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream)
{
    HIPCHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, S(stream)));
    return SXFIR_OK;
}

int sxfir_stream_sync(void *stream)
{
    HIPCHECK(hipStreamSynchronize(S(stream)));
    return SXFIR_OK;
}

int sxfir_set_device(int device)
{
    HIPCHECK(hipSetDevice(device));
    return SXFIR_OK;
}

int sxfir_host_alloc(void **host, size_t bytes)
{
    if (!host) return fail(SXFIR_EINVAL, "NULL argument");
    *host = nullptr;
    hipError_t e = hipHostMalloc(host, bytes ? bytes : 1, hipHostMallocDefault);
    if (e == hipErrorOutOfMemory) return fail(SXFIR_ENOMEM, "hipHostMalloc(%zu) out of memory", bytes);
    if (e != hipSuccess) return fail(SXFIR_EHIP, "hipHostMalloc: %s", hipGetErrorString(e));
    return SXFIR_OK;
}

int sxfir_host_register(void *host, size_t bytes)
{
    if (!host || !bytes) return fail(SXFIR_EINVAL, "NULL argument");
    hipError_t e = hipHostRegister(host, bytes, hipHostRegisterDefault);
    if (e != hipSuccess) return fail(SXFIR_EHIP, "hipHostRegister: %s", hipGetErrorString(e));
    return SXFIR_OK;
}

int sxfir_host_unregister(void *host)
{
    if (host) HIPCHECK(hipHostUnregister(host));
    return SXFIR_OK;
}

int sxfir_host_device_pointer(const void *host, size_t bytes, void **dev)
{
    if (!host || !dev) return fail(SXFIR_EINVAL, "NULL argument");
    *dev = nullptr;
    hipPointerAttribute_t at;
    hipError_t e = hipPointerGetAttributes(&at, host);
    if (e != hipSuccess) {
        (void)hipGetLastError();                        // plain pageable memory: not an error, just not visible
        return SXFIR_EUNSUPPORTED;
    }
    if (at.type != hipMemoryTypeHost) return SXFIR_EUNSUPPORTED;
    // the last byte must belong to the same page-locked range
    hipPointerAttribute_t end;
    if (bytes > 1 && (hipPointerGetAttributes(&end, (const char *)host + bytes - 1) != hipSuccess || end.type != hipMemoryTypeHost)) {
        (void)hipGetLastError();
        return SXFIR_EUNSUPPORTED;
    }
    void *d = nullptr;
    e = hipHostGetDevicePointer(&d, const_cast<void *>(host), 0);
    if (e != hipSuccess || !d) {
        (void)hipGetLastError();
        return SXFIR_EUNSUPPORTED;
    }
    // ... and to the same registration: inside one page-locked allocation the device view is linear, so the last
    // byte's device pointer is d + bytes - 1; a range that spans two registrations (or a pageable hole between
    // them) maps elsewhere and is refused -- kernels and DMA copies write through d across the whole range
    if (bytes > 1) {
        void *dl = nullptr;
        e = hipHostGetDevicePointer(&dl, const_cast<char *>((const char *)host + bytes - 1), 0);
        if (e != hipSuccess || dl != (char *)d + bytes - 1) {
            (void)hipGetLastError();
            return SXFIR_EUNSUPPORTED;
        }
        // and, where the runtime reports the allocation the device pointer belongs to, the range ends inside it
        void *base = nullptr;
        size_t size = 0;
        if (hipMemGetAddressRange((hipDeviceptr_t *)&base, &size, (hipDeviceptr_t)d) == hipSuccess && base && size) {
            if ((char *)d + bytes > (char *)base + size) return SXFIR_EUNSUPPORTED;
        } else {
            (void)hipGetLastError();
        }
    }
    *dev = d;
    return SXFIR_OK;
}

int sxfir_host_free(void *host)
{
    if (host) HIPCHECK(hipHostFree(host));
    return SXFIR_OK;
}

int sxfir_stream_create(void **stream)
{
    if (!stream) return fail(SXFIR_EINVAL, "NULL argument");
    hipStream_t st = nullptr;
    HIPCHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    *stream = (void *)st;
    return SXFIR_OK;
}

int sxfir_stream_destroy(void *stream)
{
    if (stream) HIPCHECK(hipStreamDestroy(S(stream)));
    return SXFIR_OK;
}

int sxfir_event_create(void **event)
{
    if (!event) return fail(SXFIR_EINVAL, "NULL argument");
    hipEvent_t e = nullptr;
    HIPCHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *event = (void *)e;
    return SXFIR_OK;
}

int sxfir_event_create_timing(void **event)
{
    if (!event) return fail(SXFIR_EINVAL, "NULL argument");
    hipEvent_t e = nullptr;
    HIPCHECK(hipEventCreate(&e));
    *event = (void *)e;
    return SXFIR_OK;
}

int sxfir_event_elapsed_ms(void *start, void *stop, float *ms)
{
    if (!start || !stop || !ms) return fail(SXFIR_EINVAL, "NULL argument");
    HIPCHECK(hipEventSynchronize((hipEvent_t)stop));
    HIPCHECK(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return SXFIR_OK;
}

int sxfir_event_destroy(void *event)
{
    if (event) HIPCHECK(hipEventDestroy((hipEvent_t)event));
    return SXFIR_OK;
}

int sxfir_event_record(void *event, void *stream)
{
    if (!event) return fail(SXFIR_EINVAL, "NULL event");
    HIPCHECK(hipEventRecord((hipEvent_t)event, S(stream)));
    return SXFIR_OK;
}

int sxfir_event_sync(void *event)
{
    if (!event) return fail(SXFIR_EINVAL, "NULL event");
    HIPCHECK(hipEventSynchronize((hipEvent_t)event));
    return SXFIR_OK;
}

int sxfir_stream_wait_event(void *stream, void *event)
{
    if (!event) return fail(SXFIR_EINVAL, "NULL event");
    HIPCHECK(hipStreamWaitEvent(S(stream), (hipEvent_t)event, 0));
    return SXFIR_OK;
}

}  // extern "C"

#include "sxfir_comm.hip.h"
