// Kernel dispatch of the C ABI: the launch tables of the decimators and interpolators (which kernel, which grid, which
// tap table for a plan) and the streaming entry points sxfir_decimate / sxfir_interpolate / sxfir_interpolate_keyed
// (launch, history carry-over, position commit).  Included by sxfir.hip after sxfir_plan.hip.h.
#pragma once

extern "C" {

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

// A launch that hands taps_scaled_dev to a kernel states which layout that kernel reads; sxfir_create chose the layout
// from the same plan flags, so a mismatch means the two sides were changed apart: refuse instead of filtering with
// permuted taps.
static int need_tap_table(const sxfir_plan *p, int layout, const char *kernel)
{
    if (p->tap_table == layout) return SXFIR_OK;
    return fail(SXFIR_EUNSUPPORTED, "internal: %s reads tap table layout %d, the plan carries layout %d", kernel, layout, p->tap_table);
}

#ifdef SXFIR_PROFILING
#include "sxfir_prof_dispatch.inc"   // the A/B variants' launch tables: 0 = not mine, 1 = launched, < 0 = error
#endif

// Launch geometry of a call: which kernel family, how many tiles, how many workgroups, how many of them the chip holds at
// once.  One statement of it for launch_decim / launch_interp and for sxfir_launch_geometry (tools/sizebench.py, the Device's
// log line for plans that fall to the generic kernels).
struct LaunchGeom {
    int kind;                 // GEOM_*
    const char *kernel;
    long long tile_out;       // decimator: outputs per tile; interpolator: inputs per tile
    long long n_tiles;        // per channel
    long long groups;         // workgroups per channel (times phase blocks for the CF16 interpolator tile kernel)
    long long resident;       // workgroups the chip holds at once, all channels
    int split;                // work items per tile ((tile, block) dealing of decim_blocks_kernel), else 1
};
enum { GEOM_GENERIC = 0, GEOM_MULTI = 1, GEOM_WIDE = 2, GEOM_TILE = 3, GEOM_IPASS = 4, GEOM_ITILE = 5 };

// Generations of workgroups per launch.  The plan's figure (8) was measured at 2^28 samples; the calls the API issues are 2^17 ..
// 2^25, and there the kernels whose workgroups pay a heavy prologue per launch (64 taps into VGPRs: decim_dense_kernel<16 / 32>;
// the x16 .. x96 pass kernels' window and lane constants) run 2-3 % faster when a workgroup keeps at least four tiles
// (tools/split_ab.sh: /32 at 2^25 66.7 us with 2 generations, 69.0 with 8; at 2^28 the other way round, 514 against 508): as many
// generations as leave every workgroup four tiles, at least one, at most the plan's.
static long long generations(const sxfir_plan *p, long long n_tiles, long long resident, bool heavy_prologue)
{
    if (!heavy_prologue) return p->oversub;
#ifdef SXFIR_PROFILING
    if (getenv("SXFIR_OVERSUB")) return p->oversub;              // the knob means what it says
#endif
    long long g = n_tiles * p->nchan / (4 * resident);
    if (g < 1) g = 1;
    return g > p->oversub ? p->oversub : g;
}

static long long clamp_groups(long long g, long long n_tiles)
{
    if (g < 1) g = 1;
    return g > n_tiles ? n_tiles : g;
}

// `aligned`: the output is 16-byte aligned with an even channel stride (what the tiled kernels' stores need)
static LaunchGeom decim_geom(const sxfir_plan *p, long long n_out, bool aligned, bool aligned_multi)
{
    LaunchGeom g{GEOM_GENERIC, "decim_generic_kernel", 256, (n_out + 255) / 256, (n_out + 255) / 256,
                 (long long)p->compute_units * 8, 1};
    const long long D = p->ratio;
    const long long first = ((p->consumed + D - 1) / D) * D - p->consumed;
    const bool want = p->kernel != SXFIR_KERNEL_GENERIC && first == 0;
    if (p->multi_capable && want && aligned_multi) {
        g.kind = GEOM_MULTI;
        g.kernel = p->blocks ? "decim_blocks_kernel" : p->dense32 ? "decim_dense_kernel" : "decim_multi_kernel";
        g.tile_out = p->blocks ? 512 : p->multi_waves * 8 * (64 / (p->multi_ps * (p->ratio / 4)));
        g.n_tiles = (n_out + g.tile_out - 1) / g.tile_out;
        g.resident = (long long)p->compute_units * p->occ_multi;
        g.groups = clamp_groups(g.resident * generations(p, g.n_tiles, g.resident, p->dense32 && !p->dense_subset) / p->nchan, g.n_tiles);
        // /48, /96: while a call has at most eight times as many tiles as the chip has workgroup slots, (tile, block) items are dealt,
        // one workgroup each (decim_blocks_kernel<..., SPLIT>); the plan's scratch holds that many block values
        if (p->blocks && p->join_partials && p->blocks_split && g.n_tiles * p->nchan <= p->join_tiles) {
            g.split = p->blocks;
            g.groups = g.n_tiles * p->blocks;
        }
        return g;
    }
    if (p->tile_capable && want && aligned) {
        if (p->wide8 && p->sched != 1) {
            g.kind = GEOM_WIDE;
            g.kernel = "decim4_wide_kernel";
            g.tile_out = 512;
            g.n_tiles = (n_out + 511) / 512;
            g.resident = (long long)p->compute_units * p->occ_wide;
            g.groups = clamp_groups(g.resident * p->oversub / p->nchan, g.n_tiles);
            return g;
        }
        g.kind = GEOM_TILE;
        g.kernel = "decim4_tile_kernel";
        g.tile_out = 256;
#ifdef SXFIR_PROFILING
        if (p->sgpr_r && p->ntaps == 128) g.tile_out = 64 * p->sgpr_r;
#endif
        g.n_tiles = (n_out + g.tile_out - 1) / g.tile_out;
        g.resident = (long long)p->compute_units * (p->tile_dbuf ? p->occ_db : p->occ_sb);
        g.groups = clamp_groups(g.resident * p->oversub / p->nchan, g.n_tiles);
    }
    return g;
}

static LaunchGeom interp_geom(const sxfir_plan *p, long long n_in, bool aligned)
{
    const long long n_out = n_in * p->ratio;
    LaunchGeom g{GEOM_GENERIC, "interp_generic_kernel", 256, (n_out + 255) / 256, (n_out + 255) / 256,
                 (long long)p->compute_units * 8, 1};
    if (!(p->itile_capable && p->kernel != SXFIR_KERNEL_GENERIC && aligned)) return g;
    if (p->ipass) {
        g.kind = GEOM_IPASS;
        g.kernel = "interp8_pass_kernel";
        g.tile_out = 64 * p->ipass_qi;
        g.n_tiles = (n_in + g.tile_out - 1) / g.tile_out;
        g.resident = (long long)p->compute_units * p->occ_ipass;
        // x32, x48, x96: while a call has at most four times as many tiles as the chip holds waves, (tile, phase block) items are
        // dealt (interp8_pass_kernel<..., PBSPLIT>: an interpolator's phases never meet, so nothing is joined)
        if (p->ratio > 16 && p->ipass_split && g.n_tiles * p->nchan <= 4 * g.resident) g.split = p->ratio / 16;
        // (a x48 / x96 tile is three / six blocks' work: the rule counts blocks, dealt or walked)
        g.groups = clamp_groups(g.resident * generations(p, g.n_tiles * (p->ratio >= 16 ? p->ratio / 16 : 1), g.resident, p->ratio >= 16) / p->nchan,
                                g.n_tiles * g.split);
        return g;
    }
    g.kind = GEOM_ITILE;
    g.kernel = "interp_tile_kernel";
    return g;        // (tile size and phase blocks: launch_interp, which alone knows the profiling knobs)
}

// Launch only the resampling kernel (no history update, no position change).
static int launch_decim(sxfir_plan *p, const void *in_dev, size_t n_in, size_t in_stride, void *out_dev,
                        size_t out_stride, long long n_out, hipStream_t st, bool *history_done)
{
    *history_done = false;
    const long long D = p->ratio;
    const long long first = ((p->consumed + D - 1) / D) * D - p->consumed;
    // LDS-DMA sources need no 16-byte alignment (verified on MI355X, tools/probe_unaligned.hip): only the
    // output, written with 16-byte stores, must be aligned
    const LaunchGeom geom = decim_geom(p, n_out, ((uintptr_t)out_dev % 16 == 0) && (p->nchan == 1 || out_stride % 2 == 0),
                                       ((uintptr_t)out_dev % 16 == 0) &&
                                           (p->nchan == 1 || out_stride % (p->fmt == SXFIR_CF16 ? 4 : 2) == 0));
    const bool tiled = geom.kind == GEOM_WIDE || geom.kind == GEOM_TILE;
    const bool multi = geom.kind == GEOM_MULTI;
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
        const long long n_tiles = geom.n_tiles;
        if (n_tiles > 0x7fffffffLL) return fail(SXFIR_EINVAL, "call too large");
        const long long groups = geom.groups;
        a.n_tiles = (int)n_tiles;
        a.n_groups = (int)groups;
        dim3 grid((unsigned)groups, (unsigned)p->nchan);
        a.stamps = nullptr;
        if (p->blocks) {
            // /48, /96: sixteen-column blocks, scalar taps from the block-major table of the rotated taps
            if (int rc = need_tap_table(p, TAPS_BLOCKS16, "decim_blocks_kernel")) return rc;
            a.taps = p->taps_scaled_dev;
            const sxfir::DecimBlocksJoin jn{(sxfir::f32x4 *)p->join_partials, p->join_arrived};
            // the lines no other tile reads as non-temporal loads: 2-3 % less time (profiles/round5_rates.txt)
#define SXFIR_BLOCKS_LAUNCH(NB_, S32_, NT_, HALF_) \
            do { \
                if (geom.split > 1) hipLaunchKernelGGL((sxfir::decim_blocks_kernel<NB_, S32_, NT_, HALF_, true, true>), grid, dim3(256), 0, st, a, jn); \
                else hipLaunchKernelGGL((sxfir::decim_blocks_kernel<NB_, S32_, NT_, HALF_, false, true>), grid, dim3(256), 0, st, a, jn); \
            } while (0)
#ifdef SXFIR_PROFILING
            if (p->blocks < 3 && !p->rot) {                                                                    // experiment: /16, /32 unrotated (SXFIR_BLOCKS_SMALL=2)
                if (p->blocks == 1) hipLaunchKernelGGL((sxfir::decim_blocks_kernel<1, false, true, false, false, true, false>), grid, dim3(256), 0, st, a, jn);
                else hipLaunchKernelGGL((sxfir::decim_blocks_kernel<2, false, true, false, false, true, false>), grid, dim3(256), 0, st, a, jn);
            } else
            if (p->blocks < 3) {                                                                               // experiment: /16, /32 (SXFIR_BLOCKS_SMALL=1)
                if (p->blocks == 1) hipLaunchKernelGGL((sxfir::decim_blocks_kernel<1, false, true, false, false, true>), grid, dim3(256), 0, st, a, jn);
                else hipLaunchKernelGGL((sxfir::decim_blocks_kernel<2, false, true, false, false, true>), grid, dim3(256), 0, st, a, jn);
            } else
            if (getenv("SXFIR_BLOCKS_RP") && !atoi(getenv("SXFIR_BLOCKS_RP")) && p->fmt == SXFIR_CF32) {     // A/B: round 5's form, waves by row half
                if (geom.split > 1) {
                    if (p->blocks == 3) hipLaunchKernelGGL((sxfir::decim_blocks_kernel<3, false, true, false, true, false>), grid, dim3(256), 0, st, a, jn);
                    else hipLaunchKernelGGL((sxfir::decim_blocks_kernel<6, false, true, false, true, false>), grid, dim3(256), 0, st, a, jn);
                } else {
                    if (p->blocks == 3) hipLaunchKernelGGL((sxfir::decim_blocks_kernel<3, false, true, false, false, false>), grid, dim3(256), 0, st, a, jn);
                    else hipLaunchKernelGGL((sxfir::decim_blocks_kernel<6, false, true, false, false, false>), grid, dim3(256), 0, st, a, jn);
                }
            } else
            if (getenv("SXFIR_BLOCKS_NT") && !atoi(getenv("SXFIR_BLOCKS_NT")) && p->fmt == SXFIR_CF32) {     // A/B: plain staging loads
                if (p->blocks == 3) SXFIR_BLOCKS_LAUNCH(3, false, false, false);
                else SXFIR_BLOCKS_LAUNCH(6, false, false, false);
            } else
#endif
            if (p->fmt == SXFIR_CF16) {
                // CF16 storage: the typed LDS-DMA front end
                if (p->blocks == 3) SXFIR_BLOCKS_LAUNCH(3, false, true, true);
                else SXFIR_BLOCKS_LAUNCH(6, false, true, true);
            } else if (p->blocks == 3) {
                if (p->fmt == SXFIR_S32) SXFIR_BLOCKS_LAUNCH(3, true, true, false);
                else SXFIR_BLOCKS_LAUNCH(3, false, true, false);
            } else {
                if (p->fmt == SXFIR_S32) SXFIR_BLOCKS_LAUNCH(6, true, true, false);
                else SXFIR_BLOCKS_LAUNCH(6, false, true, false);
            }
#undef SXFIR_BLOCKS_LAUNCH
            HIPCHECK(hipGetLastError());
            *history_done = true;
            return SXFIR_OK;
        }
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
            if (p->fmt == SXFIR_CF16) {
                // CF16 storage: the typed LDS-DMA front end (round 5)
                if (p->ratio == 8 && p->dense_subset) {
                    if (int rc = need_tap_table(p, TAPS_SUBSET8, "decim_dense_kernel<8, SUBSET, HALFIN>")) return rc;
                    a.taps = p->taps_scaled_dev;                  // the subset-major tap table
                    hipLaunchKernelGGL((sxfir::decim_dense_kernel<8, 0, false, 2, true, false, true>), grid, dim3(256), 0, st, a);
                }
#ifdef SXFIR_PROFILING
                else if (p->ratio == 8) hipLaunchKernelGGL((sxfir::decim_dense_kernel<8, 0, false, 0, false, false, true>), grid, dim3(256), 0, st, a);   // VGPR taps: the A/B partner
#else
                else if (p->ratio == 8) return fail(SXFIR_EUNSUPPORTED, "internal: /8 CF16 without its subset table");
#endif
                else if (p->ratio == 16) hipLaunchKernelGGL((sxfir::decim_dense_kernel<16, 0, false, 2, false, false, true>), grid, dim3(256), 0, st, a);
                else hipLaunchKernelGGL((sxfir::decim_dense_kernel<32, 0, false, 2, false, false, true>), grid, dim3(256), 0, st, a);
            } else if (p->dense_subset) {
                if (int rc = need_tap_table(p, TAPS_SUBSET8, "decim_dense_kernel<8, SUBSET>")) return rc;
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
        if (p->fmt == SXFIR_S32)                               // only wire-word plans read it (S32IN ? a.taps_scaled : a.taps)
            if (int rc = need_tap_table(p, TAPS_SCALED, "the /4 scalar-tap kernels on S32 words")) return rc;
        a.taps_scaled = p->taps_scaled_dev;
        memcpy(a.taps_k, p->taps_k, sizeof(a.taps_k));
        a.n_in = (long long)n_in;
        a.n_out = n_out;
        a.in_stride = (long long)in_stride;
        a.out_stride = (long long)out_stride;
        a.hist_stride = p->hist_len;
        // (the 4-outputs-per-lane kernels' tiles; the wide kernel's own count follows below)
        const long long n_tiles = geom.kind == GEOM_TILE ? geom.n_tiles : (n_out + 255) / 256;
        if (n_tiles > 0x7fffffffLL) return fail(SXFIR_EINVAL, "call too large");
        a.n_tiles = (int)n_tiles;
        a.sched = p->sched;
        a.stamps = nullptr;
        if (p->fmt == SXFIR_CF16 && !(p->wide8 && p->sched != 1))
            return fail(SXFIR_EUNSUPPORTED, "CF16 storage at /4 runs the wide kernel only (a profiling knob asked for another /4 variant)");
#ifdef SXFIR_PROFILING
        if (p->fmt != SXFIR_CF16)
            if (const int pr = prof_launch_tile_variant(p, a, n_out, n_tiles, st)) return pr < 0 ? pr : SXFIR_OK;   // pair / wide / tile2 variants
#endif
        if (p->wide8 && p->sched != 1) {
            // 128 symmetric taps: decim4_wide_kernel, tiles of 512 outputs, one wave (= one workgroup) per tile and pass;
            // G = CUs x 8 resident waves x 16 generations waves per launch, strided XCD-blocked passes
            const long long n_tiles2 = geom.n_tiles;
            const long long G = geom.groups;
            a.n_tiles = (int)n_tiles2;
            a.n_waves = (int)G;
            a.w8 = (G % 8 == 0) ? (int)(G / 8) : 0;
            a.run_base = a.run_extra = 0;
            {
                const int t = (int)((n_tiles2 - 1) % G);
                a.hist_wave = (p->sched == 0 && a.w8) ? (t % a.w8) * 8 + t / a.w8 : t;
            }
            dim3 grid((unsigned)G, (unsigned)p->nchan);
            if (p->symmetric) {
                if (p->fmt == SXFIR_S32) hipLaunchKernelGGL((sxfir::decim4_wide_kernel<0, true>), grid, dim3(64), 0, st, a);
                else if (p->fmt == SXFIR_CF16) hipLaunchKernelGGL((sxfir::decim4_wide_kernel<0, false, 24, true, false, 0, true>), grid, dim3(64), 0, st, a);
                else hipLaunchKernelGGL((sxfir::decim4_wide_kernel<0, false>), grid, dim3(64), 0, st, a);
            } else {
                // taps that are not bit-symmetric: the same kernel with the P0 chain's taps in VGPR pairs (ASYM); shipped for CF16 storage
                if (p->fmt == SXFIR_CF16) hipLaunchKernelGGL((sxfir::decim4_wide_kernel<0, false, 24, true, false, 0, true, true>), grid, dim3(64), 0, st, a);
#ifdef SXFIR_PROFILING
                else if (p->fmt == SXFIR_S32) hipLaunchKernelGGL((sxfir::decim4_wide_kernel<0, true, 24, true, false, 0, false, true>), grid, dim3(64), 0, st, a);
                else hipLaunchKernelGGL((sxfir::decim4_wide_kernel<0, false, 24, true, false, 0, false, true>), grid, dim3(64), 0, st, a);
#else
                else return fail(SXFIR_EUNSUPPORTED, "internal: non-symmetric taps on the wide kernel outside CF16 storage");
#endif
            }
            HIPCHECK(hipGetLastError());
            return SXFIR_OK;
        }
        // Short-lived waves in generations: W = CUs * resident waves * oversub waves per launch, each covering
        // n_tiles / W tiles in strided, XCD-blocked passes (sxfir_decim_tile.hip.h).
        long long per_chan = geom.groups;
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
        if (const int pr = prof_launch_tile_first_gen(p, a, grid, per_chan, p->tile_dbuf, st)) return pr < 0 ? pr : SXFIR_OK;
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
        a.rot = p->rot;
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
                         const KeyedRange *key = nullptr, bool *key_pending = nullptr)
{
    *history_done = false;
    const LaunchGeom geom = interp_geom(p, (long long)n_in, ((uintptr_t)out_dev % 16 == 0) && (p->nchan == 1 || out_stride % 2 == 0));
    const bool tiled = geom.kind != GEOM_GENERIC;
    if (p->kernel == SXFIR_KERNEL_TILED && !tiled)
        return fail(SXFIR_EUNSUPPORTED, "tiled interpolator needs a 16-byte aligned output and even strides");
    if (tiled && p->ipass) {
        // x8, 256 taps, CF32: the scalar-tap form, tiles of 128 inputs (two per lane), four (phase group, row half) passes per tile
        sxfir::InterpTileArgs t;
        t.in = (const float *)in_dev;
        t.hist = (const float *)p->hist_dev;
        t.hist_out = (float *)p->hist_alt;
        t.out = (float *)out_dev;
        if (int rc = need_tap_table(p, TAPS_PASS8, "interp8_pass_kernel")) return rc;
        t.taps = p->taps_scaled_dev;                            // the pass-major table
        t.n_in = (long long)n_in;
        t.in_stride = (long long)in_stride;
        t.out_stride = (long long)out_stride;
        t.hist_stride = p->hist_len;
        const long long n_tiles = geom.n_tiles * geom.split;        // (PBSPLIT: items)
        if (n_tiles > 0x7fffffffLL) return fail(SXFIR_EINVAL, "call too large");
        const long long groups = geom.groups;
        t.n_tiles = (int)n_tiles;
        t.n_groups = (int)groups;
        t.thr2 = p->thr2;
        t.key_counter = key ? key->counter : nullptr;
        t.key_lo = key ? key->lo : 0;
        t.key_hi = key ? key->hi : 0;
        const dim3 pgrid((unsigned)groups, (unsigned)p->nchan);
        if (p->ratio >= 16) {
            // x16 .. x96: two inputs per lane, ratio / 16 phase blocks of sixteen per tile
#define SXFIR_IPASS16(KK, SS) \
            switch (p->ratio + (geom.split > 1 ? 1000 : 0)) { \
            case 16: hipLaunchKernelGGL((sxfir::interp8_pass_kernel<2, KK, SS, true, 16, 16>), pgrid, dim3(64), 0, st, t); break; \
            case 32: hipLaunchKernelGGL((sxfir::interp8_pass_kernel<2, KK, SS, true, 16, 32>), pgrid, dim3(64), 0, st, t); break; \
            case 48: hipLaunchKernelGGL((sxfir::interp8_pass_kernel<2, KK, SS, true, 16, 48>), pgrid, dim3(64), 0, st, t); break; \
            case 96: hipLaunchKernelGGL((sxfir::interp8_pass_kernel<2, KK, SS, true, 16, 96>), pgrid, dim3(64), 0, st, t); break; \
            case 1032: hipLaunchKernelGGL((sxfir::interp8_pass_kernel<2, KK, SS, true, 16, 32, true>), pgrid, dim3(64), 0, st, t); break; \
            case 1048: hipLaunchKernelGGL((sxfir::interp8_pass_kernel<2, KK, SS, true, 16, 48, true>), pgrid, dim3(64), 0, st, t); break; \
            case 1096: hipLaunchKernelGGL((sxfir::interp8_pass_kernel<2, KK, SS, true, 16, 96, true>), pgrid, dim3(64), 0, st, t); break; \
            default: return fail(SXFIR_EUNSUPPORTED, "internal: no pass kernel for x%d", p->ratio); \
            }
            if (p->fmt == SXFIR_S32 && key) { SXFIR_IPASS16(true, true) }
            else if (p->fmt == SXFIR_S32) { SXFIR_IPASS16(false, true) }
            else if (key) { SXFIR_IPASS16(true, false) }
            else { SXFIR_IPASS16(false, false) }
#undef SXFIR_IPASS16
        } else
        if (p->ratio == 4) {
            // x4: four inputs per lane, two passes
            if (p->fmt == SXFIR_S32 && key) hipLaunchKernelGGL((sxfir::interp8_pass_kernel<4, true, true, true, 4>), pgrid, dim3(64), 0, st, t);
            else if (p->fmt == SXFIR_S32) hipLaunchKernelGGL((sxfir::interp8_pass_kernel<4, false, true, true, 4>), pgrid, dim3(64), 0, st, t);
            else if (key) hipLaunchKernelGGL((sxfir::interp8_pass_kernel<4, true, false, true, 4>), pgrid, dim3(64), 0, st, t);
            else hipLaunchKernelGGL((sxfir::interp8_pass_kernel<4, false, false, true, 4>), pgrid, dim3(64), 0, st, t);
        } else
#ifdef SXFIR_PROFILING
        if (p->ipass_qi == 4 && p->fmt == SXFIR_S32) return fail(SXFIR_EUNSUPPORTED, "four inputs per lane: CF32 only");
        else if (p->ipass_qi == 4 && key) hipLaunchKernelGGL((sxfir::interp8_pass_kernel<4, true>), pgrid, dim3(64), 0, st, t);
        else if (p->ipass_qi == 4) hipLaunchKernelGGL((sxfir::interp8_pass_kernel<4>), pgrid, dim3(64), 0, st, t);
        else if (p->ipass_wait0) {                          // SXFIR_IPASS_WAIT0=1: the vmcnt(0) form of every shipped instance
            if (p->fmt == SXFIR_S32 && key) hipLaunchKernelGGL((sxfir::interp8_pass_kernel<2, true, true, false>), pgrid, dim3(64), 0, st, t);
            else if (p->fmt == SXFIR_S32) hipLaunchKernelGGL((sxfir::interp8_pass_kernel<2, false, true, false>), pgrid, dim3(64), 0, st, t);
            else if (key) hipLaunchKernelGGL((sxfir::interp8_pass_kernel<2, true, false, false>), pgrid, dim3(64), 0, st, t);
            else hipLaunchKernelGGL((sxfir::interp8_pass_kernel<2, false, false, false>), pgrid, dim3(64), 0, st, t);
        } else
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
        // x48: three phase blocks of the x16 kernel; x96: three of the x32 kernel (two whole lines per input and block; six blocks
        // of the x16 kernel -- SXFIR_IBLOCK16=1 in the profiling build -- measured 5 % slower, profiles/round5_rates.txt)
        int base_l = p->ratio == 96 ? 32 : (p->ratio == 48 ? 16 : p->ratio);
#ifdef SXFIR_PROFILING
        if (p->ratio == 96 && getenv("SXFIR_IBLOCK16") && atoi(getenv("SXFIR_IBLOCK16")) && !key && p->fmt == SXFIR_CF32) base_l = 16;
#endif
        const int npb = p->ratio / base_l;
        const int qt = 4 * 4 * (32 / (base_l / 4));            // InterpTile<L>::TILE_IN
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
        dim3 grid((unsigned)(groups * npb), (unsigned)p->nchan);
        if (p->fmt == SXFIR_CF16) {
            // CF16 storage: the tile kernel with the typed LDS-DMA front end and half stores, at every ratio
            if (key) return fail(SXFIR_EUNSUPPORTED, "the keying count is defined on CF32 input");
            switch (p->ratio) {
            case 4: hipLaunchKernelGGL((sxfir::interp_tile_kernel<4, false, false, 4, true>), grid, dim3(64), 0, st, t); break;
            case 8: hipLaunchKernelGGL((sxfir::interp_tile_kernel<8, false, false, 8, true>), grid, dim3(64), 0, st, t); break;
            case 16: hipLaunchKernelGGL((sxfir::interp_tile_kernel<16, false, false, 16, true>), grid, dim3(64), 0, st, t); break;
            case 32: hipLaunchKernelGGL((sxfir::interp_tile_kernel<32, false, false, 32, true>), grid, dim3(64), 0, st, t); break;
            case 48: hipLaunchKernelGGL((sxfir::interp_tile_kernel<16, false, false, 48, true>), grid, dim3(64), 0, st, t); break;
            default: hipLaunchKernelGGL((sxfir::interp_tile_kernel<32, false, false, 96, true>), grid, dim3(64), 0, st, t); break;
            }
        }
#ifndef SXFIR_PROFILING
        // CF32 / wire-word output runs the scalar-tap pass kernels at every ratio since round 5 (p->ipass); the tile kernels'
        // CF32 instances are their A/B partners in the profiling build (SXFIR_IPASS=0)
        else return fail(SXFIR_EUNSUPPORTED, "internal: CF32 / wire-word interpolation outside the pass kernels");
#else
        else if (npb > 1) {
#define SXFIR_IBLOCKS(SS, KK) \
            do { \
                if (p->ratio == 48) hipLaunchKernelGGL((sxfir::interp_tile_kernel<16, SS, KK, 48>), grid, dim3(64), 0, st, t); \
                else hipLaunchKernelGGL((sxfir::interp_tile_kernel<32, SS, KK, 96>), grid, dim3(64), 0, st, t); \
            } while (0)
#ifdef SXFIR_PROFILING
            if (base_l == 16 && p->ratio == 96) hipLaunchKernelGGL((sxfir::interp_tile_kernel<16, false, false, 96>), grid, dim3(64), 0, st, t);
            else
#endif
            if (key && p->fmt == SXFIR_S32) SXFIR_IBLOCKS(true, true);
            else if (key) SXFIR_IBLOCKS(false, true);
            else if (p->fmt == SXFIR_S32) SXFIR_IBLOCKS(true, false);
            else SXFIR_IBLOCKS(false, false);
#undef SXFIR_IBLOCKS
        } else if (key && p->fmt == SXFIR_S32) {
            switch (p->ratio) {
#ifdef SXFIR_PROFILING
            case 4: hipLaunchKernelGGL((sxfir::interp_tile_kernel<4, true, true>), grid, dim3(64), 0, st, t); break;
#else
            case 4: return fail(SXFIR_EUNSUPPORTED, "x4 runs interp8_pass_kernel");   // (unreachable: p->ipass)
#endif
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
#ifdef SXFIR_PROFILING
            case 4: hipLaunchKernelGGL((sxfir::interp_tile_kernel<4, false, true>), grid, dim3(64), 0, st, t); break;
#else
            case 4: return fail(SXFIR_EUNSUPPORTED, "x4 runs interp8_pass_kernel");   // (unreachable: p->ipass)
#endif
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
#ifdef SXFIR_PROFILING
            case 4: hipLaunchKernelGGL((sxfir::interp_tile_kernel<4, true>), grid, dim3(64), 0, st, t); break;
#else
            case 4: return fail(SXFIR_EUNSUPPORTED, "x4 runs interp8_pass_kernel");   // (unreachable: p->ipass)
#endif
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
#ifdef SXFIR_PROFILING
            case 4: hipLaunchKernelGGL((sxfir::interp_tile_kernel<4>), grid, dim3(64), 0, st, t); break;
#else
            case 4: return fail(SXFIR_EUNSUPPORTED, "x4 runs interp8_pass_kernel");   // (unreachable: p->ipass)
#endif
#ifdef SXFIR_PROFILING
            case 8: hipLaunchKernelGGL((sxfir::interp_tile_kernel<8>), grid, dim3(64), 0, st, t); break;
#else
            case 8: return fail(SXFIR_EUNSUPPORTED, "x8 runs interp8_pass_kernel");   // (unreachable: p->ipass)
#endif
            case 16: hipLaunchKernelGGL((sxfir::interp_tile_kernel<16>), grid, dim3(64), 0, st, t); break;
            default: hipLaunchKernelGGL((sxfir::interp_tile_kernel<32>), grid, dim3(64), 0, st, t); break;
            }
        }
#endif
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
    a.rot = 0;
    dim3 grid((unsigned)((n_out + 255) / 256), (unsigned)p->nchan);
    a.thr2 = p->thr2;
    if (p->fmt == SXFIR_CF32)
        hipLaunchKernelGGL(sxfir::interp_generic_kernel<sxfir::CF32>, grid, dim3(256), 0, st, a);
    else if (p->fmt == SXFIR_CF16)
        hipLaunchKernelGGL(sxfir::interp_generic_kernel<sxfir::CF16>, grid, dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((sxfir::interp_generic_kernel<sxfir::CF32, sxfir::S32>), grid, dim3(256), 0, st, a);
    HIPCHECK(hipGetLastError());
    if (key && key->hi > key->lo && key_pending) *key_pending = true;    // counted by the caller once the call is certain to commit
    return SXFIR_OK;
}

// Shapes the tiled kernels do not take: the keying count as a pass of its own (same rule, same counter).  Queued by
// interpolate_impl AFTER the history launch has succeeded, with the position commit: a call that fails half way has not
// touched the counter, so a caller that retries the block does not count it twice (on the tiled paths the count is part of
// the one kernel launch).
static int launch_keyed_count(sxfir_plan *p, const void *in_dev, const KeyedRange *key, hipStream_t st)
{
    const long long n = key->hi - key->lo;
    unsigned g = (unsigned)std::min<long long>((n + 255) / 256, 256);
    hipLaunchKernelGGL(sxfir::count_keyed_kernel, dim3(g), dim3(256), 0, st,
                       reinterpret_cast<const float2 *>(in_dev) + key->lo, n, p->thr2, key->counter);
    HIPCHECK(hipGetLastError());
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
    bool history_done = false, key_pending = false;
    rc = launch_interp(p, in_dev, n_in, in_stride, out_dev, out_stride, n_out, S(stream), &history_done, key, &key_pending);
    if (rc) return rc;
    if (!history_done) {
        rc = launch_history(p, in_dev, n_in, in_stride, S(stream));
        if (rc) return rc;
    }
    if (key_pending) {
        rc = launch_keyed_count(p, in_dev, key, S(stream));
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

int sxfir_launch_geometry(const sxfir_plan *p, size_t n_in, sxfir_geometry *out)
{
    if (!p || !out) return fail(SXFIR_EINVAL, "NULL argument");
    memset(out, 0, sizeof(*out));
    LaunchGeom g;
    if (p->mode == SXFIR_DECIMATE) {
        g = decim_geom(p, outputs_for(p, (long long)n_in), true, true);
    } else {
        g = interp_geom(p, (long long)n_in, true);
        if (g.kind == GEOM_ITILE) {
            // CF16 storage: interp_tile_kernel, x48 / x96 as three phase blocks of its x16 / x32 form (launch_interp)
            const int base_l = p->ratio == 96 ? 32 : (p->ratio == 48 ? 16 : p->ratio);
            g.tile_out = 4 * 4 * (32 / (base_l / 4));
            g.n_tiles = ((long long)n_in + g.tile_out - 1) / g.tile_out;
            g.resident = (long long)p->compute_units * 16;
            g.groups = clamp_groups(g.resident * p->oversub / p->nchan, g.n_tiles) * (p->ratio / base_l);
        }
    }
    snprintf(out->kernel, sizeof(out->kernel), "%s", g.kernel);
    out->tiled = g.kind != GEOM_GENERIC;
    out->split = g.split;
    out->tile_samples = g.tile_out * p->ratio;       // wideband samples: a decimator's inputs, an interpolator's outputs
    if (g.kind == GEOM_GENERIC) out->tile_samples = p->mode == SXFIR_DECIMATE ? 256LL * p->ratio : 256;
    out->n_tiles = g.n_tiles;
    out->workgroups = g.groups * p->nchan;
    out->resident = g.resident;
    return SXFIR_OK;
}

}  // extern "C"
