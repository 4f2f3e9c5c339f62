// Plan bookkeeping of the C ABI (include/sxfir.h): what a plan holds, how sxfir_create chooses the kernel for a
// shape (and lays out its tap tables), and the small state entry points (reset, history, position, contract).
// Included by sxfir.hip after the kernel headers; not a stand-alone translation unit.
#pragma once

enum TapTable { TAPS_SCALED = 0, TAPS_SUBSET8 = 1, TAPS_PASS8 = 2, TAPS_BLOCKS16 = 3 };

struct sxfir_plan {
    int mode, ntaps, ratio, nchan, fmt, device;
    int kernel;            // SXFIR_KERNEL_*
    int hist_len;          // samples of history kept per channel
    int blocks;            // /48, /96: sixteen-column blocks of decim_blocks_kernel (3, 6), else 0
    int jsplit, cw;        // numeric contract
    int rot;               // ... and its rotation (0; 1 for /48, /96: sxfir_contract_rotation)
    void *join_partials;   // /48, /96: scratch of decim_blocks_kernel<..., SPLIT>: join_tiles x blocks block values of 4 KiB
    unsigned *join_arrived;   // ... and one arrival counter per (channel, tile); zero between launches
    long long join_tiles;  // tiles (over all channels) the scratch holds = the largest call the SPLIT form takes
    bool blocks_split;     // (profiling: SXFIR_BLOCKS_SPLIT=0 switches the (tile, block) dealing off)
    bool ipass_split;      // x32, x48, x96: (tile, phase block) items for small calls (profiling: SXFIR_IPASS_SPLIT=0 switches it off)
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
    float *taps_scaled_dev;   // the second tap table; its layout is one of TapTable, chosen from dense_subset / ipass in sxfir_create
    int tap_table;            // TAPS_SCALED: taps * 2^-31 (exact) in tap order, for the /4 scalar-tap kernels on S32 wire words;
                              // TAPS_SUBSET8: the subset-major table of decim_dense_kernel<8, ..., SUBSET> (times 2^-31 for S32 plans);
                              // TAPS_PASS8: the pass-major table of interp8_pass_kernel (pass (c, p) at 64 (2c + p), (jj, rr) at 4 jj + rr).
                              // Every launch that hands taps_scaled_dev to a kernel checks this first (need_tap_table).
    bool ipass;               // x8, 256 taps, CF32: interp8_pass_kernel (scalar taps, four passes per tile)
    int occ_ipass;
    int ipass_qi;             // inputs per lane of that kernel (2; profiling: 4)
    bool ipass_wait0;         // (profiling) SXFIR_IPASS_WAIT0=1: its vmcnt(0) form (A/B partner of the counted wait)
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

int sxfir_device_pci_bus_id(int device, char *bdf, size_t bdf_bytes)
{
    if (!bdf || bdf_bytes < 16) return fail(SXFIR_EINVAL, "bdf needs at least 16 bytes");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return fail(SXFIR_ENODEVICE, "no GPU visible");
    if (device < 0) HIPCHECK(hipGetDevice(&device));
    if (device >= n) return fail(SXFIR_EINVAL, "device %d of %d", device, n);
    HIPCHECK(hipDeviceGetPCIBusId(bdf, (int)bdf_bytes, device));
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
    p->ipass_wait0 = false;
    p->occ_ipass = 16;
    p->ipass_qi = 2;
    p->symmetric = true;
    for (int k = 0; k < ntaps / 2; ++k)
        if (memcmp(&taps[k], &taps[ntaps - 1 - k], sizeof(float)) != 0) p->symmetric = false;
    p->hist_dev = nullptr;
    p->hist_alt = nullptr;
    p->itile_capable = false;
    p->blocks = 0;
    p->rot = 0;
    p->join_partials = nullptr;
    p->join_arrived = nullptr;
    p->join_tiles = 0;
    p->blocks_split = true;
    p->ipass_split = true;

    if (mode == SXFIR_DECIMATE) {
        p->hist_len = (ntaps + 1) & ~1;
        // CF16 storage at /4 with 128 taps: the wide kernel with the typed-DMA front end (round 5)
        const bool half4_wide = fmt == SXFIR_CF16 && ratio == 4 && ntaps == 128;
        p->tile_capable = ((fmt == SXFIR_CF32 && ratio == 4 && (ntaps == 128 || ntaps == 64)) ||
                           (fmt == SXFIR_S32 && ratio == 4 && ntaps == 128) || half4_wide);
        // multi-column kernel: 32 taps per phase; CF32 at ratio 8/16/32, CF16 at ratio 4/8/16/32
        p->multi_capable = (ntaps == 32 * ratio) && !half4_wide &&
                           (((fmt == SXFIR_CF32 || fmt == SXFIR_S32) && (ratio == 8 || ratio == 16 || ratio == 32)) ||
                            (fmt == SXFIR_CF16 && (ratio == 4 || ratio == 8 || ratio == 16 || ratio == 32)));
        // Numeric contract: two row halves and column groups of 4 when the shape allows the
        // adjacent-pair trees, i.e. whole, even rows and a power-of-two number (<= 32) of column groups;
        // otherwise one chain over all taps.
        const int jt = (ntaps + ratio - 1) / ratio;
        const int ncol4 = ratio / 4;
        const bool pow2_cols = ratio % 4 == 0 && (ncol4 & (ncol4 - 1)) == 0 && ncol4 <= 32;
        // /48 and /96 with 32 taps per phase (the reference's rates master clock / 768 and / 1536, SoapySX.cpp:180-208):
        // decim_blocks_kernel, sixteen-column blocks of whole input lines under the ROTATED contract (slot k' holds tap
        // (k' + 1) mod ntaps: sxfir_contract_rotation); 12 / 24 column groups meet in the adjacent-pair tree whose odd element
        // at the end of a level moves up unchanged (oracle B and the generic kernel state the same tree and rotation)
        p->blocks = (ntaps == 32 * ratio && (ratio == 48 || ratio == 96)) ? ratio / 16 : 0;      // CF32, S32 words, CF16 storage
#ifdef SXFIR_PROFILING
        // experiment (round 6): /16 and /32 CF32 through the sixteen-column-block form too (one / two blocks per row, scalar taps,
        // the ROTATED contract): would config 5's shape gain what /48 and /96 gained over the VGPR-tap dense kernel?
        if (getenv("SXFIR_BLOCKS_SMALL") && atoi(getenv("SXFIR_BLOCKS_SMALL")) && fmt == SXFIR_CF32 && ntaps == 32 * ratio &&
            (ratio == 16 || ratio == 32)) {
            p->blocks = ratio / 16;
            p->blocks_split = false;
        }
#endif
        if (p->blocks) {
            p->multi_capable = true;
            p->rot = 1;
#ifdef SXFIR_PROFILING
            if (p->blocks < 3 && atoi(getenv("SXFIR_BLOCKS_SMALL")) == 2) p->rot = 0;      // ... under the unrotated contract
#endif
        }
        if (ntaps % ratio == 0 && (pow2_cols || p->blocks) && jt % 2 == 0) {
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
        // (x48, x96 -- the reference's rates master clock / 768 and / 1536 -- as three / six phase blocks of the x16 kernel)
        // (CF16 storage, round 5: interp_tile_kernel<.., HALF> at every one of these ratios)
        p->itile_capable = ((fmt == SXFIR_CF32 || fmt == SXFIR_S32 || fmt == SXFIR_CF16) && ntaps == 32 * ratio &&
                            (ratio == 4 || ratio == 8 || ratio == 16 || ratio == 32 || ratio == 48 || ratio == 96));
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
    // ... and, since round 5, CF16 storage at /32 (BASELINE config 5's fp16 leg): the dense kernel with the typed LDS-DMA front end
    // (HALFIN: the texture path converts half -> float on the way into the same CF32 image; no conversions in the FIR)
    p->dense32 = p->multi_capable && !p->blocks && (ratio == 8 || ratio == 16 || ratio == 32);
    // /8 CF32: the scalar-tap form of the dense kernel (tap subsets on the four waves, round 4: 4.4-5 % less time)
    p->dense_hc = false;
    p->dense_subset = p->dense32 && ratio == 8;      // (CF16 storage too, round 5: the typed-DMA front end under the same scalar-tap FIR)
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
    if (p->itile_capable) p->oversub = ratio > 32 ? 2 : 4;     // (x48, x96: 2 measured 1-2 % ahead of 4 .. 32, profiles/round5_rates.txt)
    if (p->itile_capable && ratio == 8 && (fmt == SXFIR_CF32 || fmt == SXFIR_S32)) {
        p->ipass = true;
        p->oversub = 8;                                 // measured (tools/ibench2.py): 4 / 8 / 16 generations within 0.3 %
        int nbi = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nbi, (const void *)sxfir::interp8_pass_kernel<2>, 64, 0) == hipSuccess && nbi > 0)
            p->occ_ipass = nbi;
    }
    // x16 .. x96 (round 5): the pass kernel over ratio / 16 phase blocks of sixteen per tile (a lane's sixteen outputs per input
    // and block are one line): 8-9 % less time than interp_tile_kernel, which keeps CF16 storage (profiles/round5_rates.txt)
    if (p->itile_capable && ratio >= 16 && (fmt == SXFIR_CF32 || fmt == SXFIR_S32)) {
        p->ipass = true;
        p->ipass_qi = 2;
        p->oversub = 8;
        int nbi = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nbi, (const void *)sxfir::interp8_pass_kernel<2, false, false, true, 16, 16>, 64, 0) == hipSuccess && nbi > 0)
            p->occ_ipass = nbi;
    }
    // x4, 128 taps (round 5): the same scalar-tap pass form with four inputs per lane (two passes; a lane's sixteen outputs are
    // one line)
    if (p->itile_capable && ratio == 4 && (fmt == SXFIR_CF32 || fmt == SXFIR_S32)) {
        p->ipass = true;
        p->ipass_qi = 4;
        p->oversub = 8;
        int nbi = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nbi, (const void *)sxfir::interp8_pass_kernel<4, false, false, true, 4>, 64, 0) == hipSuccess && nbi > 0)
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
        if (p->ipass && atoi(v) == 4 && ratio == 8) {
            p->ipass_qi = 4;
            int nbi = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nbi, (const void *)sxfir::interp8_pass_kernel<4>, 64, 0) == hipSuccess && nbi > 0)
                p->occ_ipass = nbi;
        }
    }
    if (const char *v = getenv("SXFIR_IPASS_WAIT0")) p->ipass_wait0 = atoi(v) != 0;
    if (const char *v = getenv("SXFIR_IPASS_SPLIT")) p->ipass_split = atoi(v) != 0;
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
        if (p->blocks) {
            const bool w = fmt == SXFIR_S32;
#ifdef SXFIR_PROFILING
            if (p->blocks < 3)       // (the unrotated instances have the same resources)
                k = p->blocks == 1 ? (const void *)sxfir::decim_blocks_kernel<1, false, true, false, false, true> : (const void *)sxfir::decim_blocks_kernel<2, false, true, false, false, true>;
            else
#endif
            k = p->blocks == 3 ? (w ? (const void *)sxfir::decim_blocks_kernel<3, true, true, false, false, true> : (const void *)sxfir::decim_blocks_kernel<3, false, true, false, false, true>)
                               : (w ? (const void *)sxfir::decim_blocks_kernel<6, true, true, false, false, true> : (const void *)sxfir::decim_blocks_kernel<6, false, true, false, false, true>);
            if (fmt == SXFIR_CF16 && p->blocks >= 3)
                k = p->blocks == 3 ? (const void *)sxfir::decim_blocks_kernel<3, false, true, true, false, true> : (const void *)sxfir::decim_blocks_kernel<6, false, true, true, false, true>;
        } else if (p->dense32 && fmt == SXFIR_CF16) {
            k = ratio == 8    ? (const void *)sxfir::decim_dense_kernel<8, 0, false, 2, true, false, true>
                : ratio == 16 ? (const void *)sxfir::decim_dense_kernel<16, 0, false, 2, false, false, true>
                              : (const void *)sxfir::decim_dense_kernel<32, 0, false, 2, false, false, true>;
        } else if (p->dense32) {
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
    if (p->blocks) {
        // calls of at most eight times as many tiles as the chip has workgroup slots are dealt as (tile, block) items (SPLIT).
        // Measured (tools/split_ab.sh, profiles/round6_split_ab.txt): against the walking form the dealt form takes 2.9 x less time at
        // 2^22 samples (/96), -39 % at 2^24, -15 % at 2^26 (2731 / 1366 tiles), -6 % at 5.3 x slots and +2 .. +5 % at 10.7 x slots
        // (2^28 samples at /96: the walking form keeps a tile's halo in one XCD's L2 and pays one prologue per eight tiles).
        // Scratch: 4096 tiles x blocks x 4 KiB = 48 / 96 MiB per plan.
        p->join_tiles = 8LL * p->compute_units * p->occ_multi;
#ifdef SXFIR_PROFILING
        if (const char *v = getenv("SXFIR_BLOCKS_SPLIT")) {     // 0: off; n >= 1: dealt while a call has at most n x slots tiles
            p->blocks_split = atoi(v) != 0;
            if (atoi(v) > 1) p->join_tiles = (long long)atoi(v) * p->compute_units * p->occ_multi;
        }
#endif
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
        if (ntaps == 128 && (p->symmetric || fmt == SXFIR_CF16)) {
            // the shipped form for 128 bit-symmetric taps (every linear-phase design): eight outputs per lane, all 64 distinct taps in
            // SGPR pairs (sxfir_decim_wide.hip.h); 18.5 KB of LDS per wave -> 8 waves per CU.  Its ASYM form (round 5: taps 127..64 in
            // SGPR pairs, taps 63..0 in VGPR pairs) ships for CF16 storage only, where it beats the multi-column kernel by 3 %; on CF32
            // and S32 words it measured 1.4 % slower / 0.7 % faster than decim4_tile_kernel<128> (profiles/round5_kbench_asym.txt),
            // which therefore keeps the non-symmetric 128-tap plans (the instances exist in the profiling build: SXFIR_WIDE_ASYM=1)
            p->wide8 = true;
            const void *kw;
            if (p->symmetric)
                kw = fmt == SXFIR_S32    ? (const void *)sxfir::decim4_wide_kernel<0, true>
                     : fmt == SXFIR_CF16 ? (const void *)sxfir::decim4_wide_kernel<0, false, 24, true, false, 0, true>
                                         : (const void *)sxfir::decim4_wide_kernel<0, false>;
            else
                kw = (const void *)sxfir::decim4_wide_kernel<0, false, 24, true, false, 0, true, true>;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kw, 64, 0) == hipSuccess && nb > 0) p->occ_wide = nb;
        }
#ifdef SXFIR_PROFILING
        if (ntaps == 128 && !p->symmetric && fmt != SXFIR_CF16 && getenv("SXFIR_WIDE_ASYM") && atoi(getenv("SXFIR_WIDE_ASYM"))) {
            p->wide8 = true;                                   // A/B: the wide kernel's ASYM form on CF32 / S32 words
            int nbw = 0;
            const void *kw = fmt == SXFIR_S32 ? (const void *)sxfir::decim4_wide_kernel<0, true, 24, true, false, 0, false, true>
                                              : (const void *)sxfir::decim4_wide_kernel<0, false, 24, true, false, 0, false, true>;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nbw, kw, 64, 0) == hipSuccess && nbw > 0) p->occ_wide = nbw;
        }
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
                // SXFIR_LDS_PAD: dynamic LDS bytes on top of the kernel's own image: fewer waves fit a CU (a probe: what would a
                // form with a larger tile per wave -- 16 outputs per lane, 34 KB -- have left of the latency hiding?)
                if (const char *lp = getenv("SXFIR_LDS_PAD")) {
                    p->lds_pad = atoi(lp) > 0 ? atoi(lp) : 0;
                    int nbw = 0;
                    if (p->lds_pad && hipOccupancyMaxActiveBlocksPerMultiprocessor(&nbw, (const void *)sxfir::decim4_wide_kernel<0, false, 24, true>, 64,
                                                                                     (size_t)p->lds_pad) == hipSuccess && nbw > 0)
                        p->occ_wide = nbw;
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
        // the layout follows the kernel the plan will launch (the same flags launch_decim / launch_interp branch on),
        // not the shape: a plan whose /8 scalar-tap form was switched off (profiling knobs) keeps the plain table
        p->tap_table = p->blocks ? TAPS_BLOCKS16 : p->dense_subset ? TAPS_SUBSET8 : (mode == SXFIR_INTERPOLATE && p->ipass) ? TAPS_PASS8 : TAPS_SCALED;
        if (p->tap_table == TAPS_SUBSET8) {
            // /8 scalar-tap form (decim_dense_kernel<8, ..., SUBSET>): subset s = 2c + p at 64 s, (jj, rr) at 4 jj + rr
            for (int c = 0; c < 2; ++c)
                for (int ph = 0; ph < 2; ++ph)
                    for (int jj = 0; jj < 16; ++jj)
                        for (int rr = 0; rr < 4; ++rr)
                            scaled[(size_t)(64 * (2 * c + ph) + 4 * jj + rr)] =
                                taps[8 * (16 * ph + jj) + 4 * c + rr] * (fmt == SXFIR_S32 ? 4.656612873077393e-10f : 1.0f);   // 2^-31: exact
        } else if (p->tap_table == TAPS_BLOCKS16) {
            // decim_blocks_kernel: the ROTATED taps (slot k' holds tap (k' + 1) mod ntaps); block b (columns 16 b .. 16 b + 15)
            // at 512 b, subset s = 2c + p at 64 s inside it, (jj, rr) at 4 jj + rr
            for (int b = 0; b < p->blocks; ++b)
                for (int c = 0; c < 4; ++c)
                    for (int ph = 0; ph < 2; ++ph)
                        for (int jj = 0; jj < 16; ++jj)
                            for (int rr = 0; rr < 4; ++rr) {
                                const int slot = ratio * (16 * ph + jj) + 16 * b + 4 * c + rr;
                                scaled[(size_t)(512 * b + 64 * (2 * c + ph) + 4 * jj + rr)] =
                                    taps[(slot + p->rot) % ntaps] * (fmt == SXFIR_S32 ? 4.656612873077393e-10f : 1.0f);
                            }
        } else if (p->tap_table == TAPS_PASS8) {
            const int ll = ratio >= 16 ? 16 : ratio;            // phases per (block of the) pass kernel
            for (int b = 0; b < ratio / ll; ++b)
            for (int c = 0; c < ll / 4; ++c)                    // x8: two phase groups; x4: one; x16 blocks: four
                for (int ph = 0; ph < 2; ++ph)
                    for (int jj = 0; jj < 16; ++jj)
                        for (int rr = 0; rr < 4; ++rr)
                            scaled[(size_t)(32 * ll * b + 64 * (2 * c + ph) + 4 * jj + rr)] = taps[(16 * ph + jj) * ratio + ll * b + 4 * c + rr];
        } else {
            for (float &t : scaled) t *= 4.656612873077393e-10f;      // 2^-31: exact
        }
        e = hipMemcpy(p->taps_scaled_dev, scaled.data(), sizeof(float) * (size_t)ntaps, hipMemcpyHostToDevice);
    }
    if (e == hipSuccess && p->blocks) {
        e = hipMalloc(&p->join_partials, (size_t)p->join_tiles * (size_t)p->blocks * 4096);
        if (e == hipSuccess) e = hipMalloc((void **)&p->join_arrived, sizeof(unsigned) * (size_t)p->join_tiles);
        if (e == hipSuccess) e = hipMemset(p->join_arrived, 0, sizeof(unsigned) * (size_t)p->join_tiles);
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
        if (p->join_partials) (void)hipFree(p->join_partials);
        if (p->join_arrived) (void)hipFree(p->join_arrived);
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
    if (p->join_partials) (void)hipFree(p->join_partials);
    if (p->join_arrived) (void)hipFree(p->join_arrived);
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

int sxfir_contract_rotation(const sxfir_plan *p, int *rot)
{
    if (!p || !rot) return fail(SXFIR_EINVAL, "NULL argument");
    *rot = p->rot;
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

}  // extern "C"
