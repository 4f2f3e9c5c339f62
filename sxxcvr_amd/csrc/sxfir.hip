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
#include "sxfir_common.hip.h"
#ifdef SXFIR_PROFILING
#include "sxfir_decim_multi.hip.h"              // A/B partner of decim_dense_kernel: no instance in the production library since round 5
#endif

// Instantiated (ratio, waves per workgroup, CF16, row split) variants of the multi-column decimator: one list
// for the occupancy query in sxfir_create and the launch in launch_decim.  The production library carries
// what it launches; the profiling build (-DSXFIR_PROFILING, libsxfir_prof.so: tools/ and
// tests/test_gpu_variants.py) adds the A/B variants, the ablation modes and the environment knobs.
// (shipped: CF16 storage only -- since round 3 every CF32 / S32-word plan at ratio 8, 16, 32 runs decim_dense_kernel,
// so the multi-column kernel's CF32 and S32 instances exist in the profiling build alone, as the A/B partner)
// (Round 5: CF16 storage left the multi-column kernel altogether -- decim_dense_kernel<D, ..., HALFIN> runs /8, /16, /32 and
// decim4_wide_kernel<..., HALFIN> runs /4, 8-15 % faster through typed LDS-DMA -- so the production library carries no instance of it;
// the profiling build keeps them as A/B partners: SXFIR_DENSE=0.)
#define SXFIR_MULTI_SHIPPED(X)
#ifdef SXFIR_PROFILING
#define SXFIR_MULTI_VARIANTS(X) \
    SXFIR_MULTI_SHIPPED(X) X(4, 1, true, 2) X(8, 4, true, 2) X(16, 4, true, 2) X(32, 4, true, 2) \
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
#include "sxfir_decim_blocks.hip.h"             // /48, /96: sixteen-column blocks (whole input lines), scalar taps (round 5)
#include "sxfir_interp_tile.hip.h"
#include "sxfir_interp_pass.hip.h"
#include "sxfir_decim_wide.hip.h"               // /4, 128 symmetric taps: the shipped form since round 4
#ifdef SXFIR_PROFILING
#include "sxfir_decim_tile2.hip.h"              // round 3's /4 form and its variants: A/B partners of the wide kernel, no instance in the production library since round 4
#include "experiments/sxfir_decim_pair.hip.h"   // measured variant, not shipped (LABBOOK.md 5.1)
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

#include "sxfir_plan.hip.h"      // struct sxfir_plan, sxfir_create ... sxfir_outputs_for
#include "sxfir_launch.hip.h"    // launch tables, sxfir_decimate / sxfir_interpolate / sxfir_interpolate_keyed
#include "sxfir_timing.hip.h"    // sxfir_time_*, sxfir_clock_probe_*
#include "sxfir_runtime.hip.h"   // synthetic source, converters, time arithmetic, tap design, memory / stream / event helpers
#include "sxfir_comm.hip.h"      // sxfir_comm_*: the RCCL gather
