/*
 * sxfir_prof.h -- extra entry points of the PROFILING build of the resampling
 * library (sxxcvr_amd/lib/libsxfir_prof.so, built with -DSXFIR_PROFILING).
 *
 * The profiling build carries the kernel A/B variants and ablation modes and
 * reads its knobs from the environment when a plan is created
 * (SXFIR_TILE_VARIANT, SXFIR_OVERSUB, SXFIR_OCC, SXFIR_SCHED, SXFIR_ABLATE,
 * SXFIR_MULTI_PS, SXFIR_MULTI_W; see tools/kbench.py).  Some ablation modes
 * produce wrong results on purpose.  It is used by tools/ and
 * tests/test_gpu_variants.py only; the production library libsxfir.so has none
 * of this and never looks at the environment.
 */
#ifndef SXFIR_PROF_H
#define SXFIR_PROF_H

#include "sxfir.h"

#ifdef __cplusplus
extern "C" {
#endif

/* SXFIR_ABLATE=11/12: median in-kernel shader clock (MHz) of the last launch. */
int sxfir_debug_clock(sxfir_plan *plan, double *mhz);

/* SXFIR_ABLATE=3 on the multi-column decimator: copies the raw per-wave stamp records of the last launch
 * (5 x uint64 each: tiles, cycles in reduction + store, waiting for data, arithmetic, barrier + issuing the
 * next tile's DMAs) to `host`; returns the number of records through *n_records. */
/* SXFIR_ABLATE=5 on the tile2 /4 kernel: records of 8 x uint64 {tiles, cycles issuing DMAs, waiting for data,
 * FIR arithmetic, output transposition + stores, whole-wave cycles, whole-wave 100 MHz ticks, 0}. */
int sxfir_debug_stamps(sxfir_plan *plan, unsigned long long *host, size_t capacity_records, size_t *n_records);

#ifdef __cplusplus
}
#endif
#endif
