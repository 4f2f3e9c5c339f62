/*
 * sx_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY; see sx_oracle.h header
 * for the parity status: "parity unpinned" w.r.t. the reference, which has no
 * software FIR, no tests and cannot be built in this image).
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off ...).
 * -ffp-contract=off is REQUIRED: every fused multiply-add below is an explicit
 * fmaf(); nothing else may be contracted, or oracle B stops being bit-exact
 * against the HIP kernels.
 */
#include "sx_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- */
/* a-T: SoapySDR::ticksToTimeNs / timeNsToTicks.                             */
/* Third-party: SoapySDR lib/TimeC.cpp (pothosware/SoapySDR, version not     */
/* pinned by the reference).  Published algorithm restated: split into whole */
/* seconds (integer) and a sub-second remainder (double), so that 64-bit     */
/* tick counts do not lose precision in a double.  Reference call sites:     */
/* SoapySX.cpp:564 (timestamp_to_samples) and :570 (samples_to_timestamp).   */
/* ------------------------------------------------------------------------- */
long long sxo_ticks_to_time_ns(long long ticks, double rate)
{
    const long long ratell = (long long)rate;
    const long long full = ticks / ratell;
    const long long err = ticks - full * ratell;
    const double part = (double)full * (rate - (double)ratell);
    const double frac = (((double)err - part) * 1000000000.0) / rate;
    return full * 1000000000LL + llround(frac);
}

long long sxo_time_ns_to_ticks(long long time_ns, double rate)
{
    const long long ratell = (long long)rate;
    const long long full = time_ns / 1000000000LL;
    const long long err = time_ns - full * 1000000000LL;
    const double part = (double)full * (rate - (double)ratell);
    const double frac = part + ((double)err * rate) / 1000000000.0;
    return full * ratell + llround(frac);
}

/* ------------------------------------------------------------------------- */
/* a-3: convert_rx_buffer, SoapySX.cpp:103-112: dst = 2^-31 * (float)src.     */
/* ------------------------------------------------------------------------- */
void sxo_convert_rx(const int32_t *src, float *dst, size_t n)
{
    const float scaling = 1.0f / 2147483648.0f;
    for (size_t i = 0; i < 2 * n; i++)
        dst[i] = scaling * (float)src[i];
}

/* convert_rx over `threads` threads (bench.py's conversion-only CPU figure: this loop is all the arithmetic
 * the reference's readStream does per sample, SoapySX.cpp:953). */
void sxo_convert_rx_mt(const int32_t *src, float *dst, size_t n, int threads)
{
    const int64_t chunk = 1 << 16;
    const int64_t nchunks = ((int64_t)n + chunk - 1) / chunk;
    if (threads < 1) threads = 1;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads)
#endif
    for (int64_t c = 0; c < nchunks; c++) {
        const int64_t b = c * chunk;
        const int64_t len = b + chunk <= (int64_t)n ? chunk : (int64_t)n - b;
        sxo_convert_rx(src + 2 * b, dst + 2 * b, (size_t)len);
    }
}

/* One rule for both corners of convert_tx_buffer: the arithmetic is the C++ SOURCE's, evaluated
 * as the abstract machine says (every product and sum rounded once, no contraction: fusing fi*fi + fq*fq into an
 * fma is a compiler licence -- gcc's -ffp-contract=fast default takes it on AArch64 -- not something the source
 * or the platform defines, and which operand it would fuse is the compiler's choice); where the abstract machine
 * leaves a result UNDEFINED, the instruction the reference's platform executes decides.  That happens once:
 * SoapySX.cpp:124-125 converts 2^31 * 1.0f to int32, which overflows; the Raspberry Pi's AArch64 FCVTZS saturates
 * and maps NaN to 0 whatever the compiler does (x86's cvttss2si yields INT32_MIN).  So: saturate, NaN -> 0. */
static int32_t sat_f32_to_i32(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return INT32_MAX;
    if (v <= -2147483648.0f) return INT32_MIN;
    return (int32_t)v;
}

/* a-4: convert_tx_buffer, SoapySX.cpp:116-137. */
void sxo_convert_tx(const float *src, int32_t *dst, size_t n, float tx_threshold2)
{
    const float scaling = 2147483648.0f; /* (float)0x7FFFFFFF rounds up to 2^31, :120 */
    for (size_t i = 0; i < 2 * n; i += 2) {
        const float fi = src[i], fq = src[i + 1];
        /* std::max(std::min(f, 1.0f), -1.0f): min(a,b) = (b<a)?b:a, max(a,b) = (a<b)?b:a */
        float ci = (1.0f < fi) ? 1.0f : fi;
        ci = (ci < -1.0f) ? -1.0f : ci;
        float cq = (1.0f < fq) ? 1.0f : fq;
        cq = (cq < -1.0f) ? -1.0f : cq;
        int32_t vi = sat_f32_to_i32(scaling * ci);
        int32_t vq = sat_f32_to_i32(scaling * cq);
        vi &= (int32_t)0xFFFFFFFC;
        vq &= (int32_t)0xFFFFFFFC;
        /* the source's arithmetic: two roundings for the products, one for the sum (rule above) */
        const float ii = fi * fi;
        const float qq = fq * fq;
        if (ii + qq >= tx_threshold2)
            vi |= 3;
        dst[i] = vi;
        dst[i + 1] = vq;
    }
}

/* ------------------------------------------------------------------------- */
/* Synthetic IQ source: counter-based (splitmix64 stream keyed by seed and   */
/* channel), so any tile is reproducible anywhere.  24-bit uniform in        */
/* [-1, 1 - 2^-23], exactly representable in fp32.                           */
/* ------------------------------------------------------------------------- */
static inline uint64_t sm64_finalize(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static inline uint64_t synth_key(uint64_t seed, uint32_t channel)
{
    return sm64_finalize(seed + 0x9E3779B97F4A7C15ULL * ((uint64_t)channel + 1));
}

void sxo_synth_iq(uint64_t seed, uint32_t channel, int64_t start, size_t n, float *out)
{
    const uint64_t key = synth_key(seed, channel);
    for (size_t i = 0; i < n; i++) {
        const int64_t idx = start + (int64_t)i;
        if (idx < 0) {
            out[2 * i] = 0.0f;
            out[2 * i + 1] = 0.0f;
            continue;
        }
        const uint64_t u = sm64_finalize(key + 0x9E3779B97F4A7C15ULL * ((uint64_t)idx + 1));
        const int32_t a = (int32_t)(u >> 40) - 8388608;
        const int32_t b = (int32_t)((u >> 16) & 0xFFFFFF) - 8388608;
        out[2 * i] = (float)a * (1.0f / 8388608.0f);
        out[2 * i + 1] = (float)b * (1.0f / 8388608.0f);
    }
}

/* The same stream produced by `threads` threads (the source is a pure function of the index). */
void sxo_synth_iq_mt(uint64_t seed, uint32_t channel, int64_t start, size_t n, float *out, int threads)
{
    const int64_t chunk = 1 << 16;
    const int64_t nchunks = ((int64_t)n + chunk - 1) / chunk;
    if (threads < 1) threads = 1;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads)
#endif
    for (int64_t c = 0; c < nchunks; c++) {
        const int64_t b = c * chunk;
        const int64_t len = b + chunk <= (int64_t)n ? chunk : (int64_t)n - b;
        sxo_synth_iq(seed, channel, start + b, (size_t)len, out + 2 * b);
    }
}

/* ------------------------------------------------------------------------- */
/* Kaiser-windowed sinc low-pass, cutoff 0.5/ratio cycles per sample.         */
/* ------------------------------------------------------------------------- */
static double bessel_i0(double x)
{
    /* power series: sum ((x/2)^k / k!)^2 */
    double sum = 1.0, term = 1.0;
    const double q = 0.25 * x * x;
    for (int k = 1; k < 200; k++) {
        term *= q / ((double)k * (double)k);
        sum += term;
        if (term < 1e-17 * sum) break;
    }
    return sum;
}

void sxo_design_lowpass(int ntaps, int ratio, double beta, double gain, float *taps)
{
    double *h = (double *)malloc(sizeof(double) * (size_t)ntaps);
    const double fc = 0.5 / (double)ratio;
    const double mid = 0.5 * (double)(ntaps - 1);
    const double i0b = bessel_i0(beta);
    double sum = 0.0;
    for (int k = 0; k < ntaps; k++) {
        const double t = (double)k - mid;
        const double a = 2.0 * fc * t;
        const double s = (t == 0.0) ? 1.0 : sin(M_PI * a) / (M_PI * a);
        const double r = t / mid;
        const double arg = 1.0 - r * r;
        const double w = bessel_i0(beta * sqrt(arg > 0.0 ? arg : 0.0)) / i0b;
        h[k] = 2.0 * fc * s * w;
        sum += h[k];
    }
    for (int k = 0; k < ntaps; k++)
        taps[k] = (float)(h[k] * gain / sum);
    free(h);
}

/* ------------------------------------------------------------------------- */
/* a-0 oracle A: fp64.                                                        */
/* ------------------------------------------------------------------------- */
int sxo_decim_f64(const float *h, int ntaps, int D, const float *x, size_t n_x,
                  int64_t m0, size_t n_out, float *y)
{
    for (size_t o = 0; o < n_out; o++) {
        const int64_t m = m0 + (int64_t)o;
        double ai = 0.0, aq = 0.0;
        if (m * D >= (int64_t)n_x) return -1;
        for (int k = 0; k < ntaps; k++) {
            const int64_t idx = m * D - k;
            if (idx < 0) break;
            ai += (double)h[k] * (double)x[2 * idx];
            aq += (double)h[k] * (double)x[2 * idx + 1];
        }
        y[2 * o] = (float)ai;
        y[2 * o + 1] = (float)aq;
    }
    return 0;
}

int sxo_interp_f64(const float *h, int ntaps, int L, const float *x, size_t n_x,
                   int64_t n0, size_t n_out, float *y)
{
    const int jt = ntaps / L;
    for (size_t o = 0; o < n_out; o++) {
        const int64_t n = n0 + (int64_t)o;
        const int64_t q = n / L;
        const int r = (int)(n % L);
        double ai = 0.0, aq = 0.0;
        if (q >= (int64_t)n_x) return -1;
        for (int j = 0; j < jt; j++) {
            const int64_t idx = q - j;
            if (idx < 0) break;
            const double c = (double)h[j * L + r];
            ai += c * (double)x[2 * idx];
            aq += c * (double)x[2 * idx + 1];
        }
        y[2 * o] = (float)ai;
        y[2 * o + 1] = (float)aq;
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* a-0 oracle B: order-matched fp32.  THE NUMERIC CONTRACT of the HIP kernels */
/* (DESIGN.md "Numeric contract"):                                           */
/*   partial_p = fmaf chain from +0.0f over the taps of group p, DESCENDING k */
/*   y = balanced adjacent-pair tree over partial_0..partial_{G-1}            */
/* Samples before the start of the stream are +0.0f and still go through the */
/* fmaf (fmaf(h, 0, acc) == acc for every acc the chain can hold).            */
/* ------------------------------------------------------------------------- */
#define SXO_MAX_GROUPS 32

/* Adjacent-pair tree: every level adds neighbours (0,1), (2,3), ...; an odd element at the end of a level moves up
 * unchanged (12 groups: 6, 3, then (a + b) and c, then their sum).  For a power of two this is the balanced tree. */
static inline float tree_sum(float *p, int g)
{
    while (g > 1) {
        const int h = g / 2;
        for (int i = 0; i < h; i++) p[i] = p[2 * i] + p[2 * i + 1];
        if (g & 1) p[h] = p[g - 1];
        g = h + (g & 1);
    }
    return p[0];
}

/* Decimator contract with polyphase rows j and phases r (tap k = j*D + r):
 *   partial[c][p] = fmaf chain from +0.0f over j DESCENDING in row range p
 *                   (jsplit contiguous ranges of ceil(ntaps/D)/jsplit rows) and,
 *                   inside a row, r DESCENDING in column group c (cw phases);
 *                   i.e. ascending sample time inside the (c, p) subset
 *   col[c]        = adjacent-pair tree over p of partial[c][p]
 *   y             = adjacent-pair tree over c of col[c]
 * (jsplit = 1, cw = D) is the plain descending-k chain.
 * rot (0 or 1; ratios 48 and 96): the rows and columns above are those of the ROTATED tap index k' = (k - rot) mod ntaps,
 *   i.e. slot k' = j*D + r stands for tap k = (k' + rot) mod ntaps and its sample x[m*D - k].  The filter is the same
 *   sum; what changes is which taps share a chain: with rot = 1 a row of the picture is the D samples that END BEFORE
 *   sample (m - j)*D -- whole 128-byte lines of the input for D = 48, 96 -- and tap 0 (the newest sample, alone in the
 *   next line) opens the chain of the last subset instead of closing the chain of the first. */
static void decim_f32_range(const float *h, int ntaps, int D, int jsplit, int cw, int rot, const float *x,
                            int64_t m_begin, int64_t m_end, int64_t m0, float *y)
{
    const int jt = (ntaps + D - 1) / D;
    const int jl = jt / jsplit;
    const int ncol = D / cw;
    /* Blocks of BLK outputs so the compiler can vectorise across outputs; the
     * per-output operation order is unchanged. */
    enum { BLK = 16 };
    int64_t m = m_begin;
    while (m < m_end) {
        const int nb = (int)((m_end - m) < BLK ? (m_end - m) : BLK);
        const int safe = (m * (int64_t)D - ((int64_t)jt * D - 1) >= 0) && nb == BLK;
        float ci[SXO_MAX_GROUPS][BLK], cq[SXO_MAX_GROUPS][BLK];
        for (int c = 0; c < ncol; c++) {
            float pi[SXO_MAX_GROUPS][BLK], pq[SXO_MAX_GROUPS][BLK];
            for (int p = 0; p < jsplit; p++) {
                float ai[BLK], aq[BLK];
                for (int b = 0; b < BLK; b++) { ai[b] = 0.0f; aq[b] = 0.0f; }
                for (int j = (p + 1) * jl - 1; j >= p * jl; j--) {
                    for (int r = (c + 1) * cw - 1; r >= c * cw; r--) {
                        int k = j * D + r;
                        if (k >= ntaps) continue;
                        k += rot;
                        if (k >= ntaps) k -= ntaps;
                        const float t = h[k];
                        if (safe) {
                            const float *xs = x + 2 * (m * (int64_t)D - k);
                            for (int b = 0; b < BLK; b++) {
                                ai[b] = fmaf(t, xs[2 * (int64_t)b * D], ai[b]);
                                aq[b] = fmaf(t, xs[2 * (int64_t)b * D + 1], aq[b]);
                            }
                        } else {
                            for (int b = 0; b < nb; b++) {
                                const int64_t idx = (m + b) * (int64_t)D - k;
                                const float xi = idx < 0 ? 0.0f : x[2 * idx];
                                const float xq = idx < 0 ? 0.0f : x[2 * idx + 1];
                                ai[b] = fmaf(t, xi, ai[b]);
                                aq[b] = fmaf(t, xq, aq[b]);
                            }
                        }
                    }
                }
                for (int b = 0; b < BLK; b++) { pi[p][b] = ai[b]; pq[p][b] = aq[b]; }
            }
            for (int b = 0; b < nb; b++) {
                float ti[SXO_MAX_GROUPS] = {0.0f}, tq[SXO_MAX_GROUPS] = {0.0f};
                for (int p = 0; p < jsplit; p++) { ti[p] = pi[p][b]; tq[p] = pq[p][b]; }
                ci[c][b] = tree_sum(ti, jsplit);
                cq[c][b] = tree_sum(tq, jsplit);
            }
        }
        for (int b = 0; b < nb; b++) {
            float ti[SXO_MAX_GROUPS] = {0.0f}, tq[SXO_MAX_GROUPS] = {0.0f};
            for (int c = 0; c < ncol; c++) { ti[c] = ci[c][b]; tq[c] = cq[c][b]; }
            y[2 * (m + b - m0)] = tree_sum(ti, ncol);
            y[2 * (m + b - m0) + 1] = tree_sum(tq, ncol);
        }
        m += nb;
    }
}

static int pow2(int v) { return v >= 1 && !(v & (v - 1)); }

static int decim_check(int ntaps, int D, int jsplit, int cw, int rot, size_t n_x, int64_t m0, size_t n_out)
{
    if (ntaps < 1 || D < 1 || m0 < 0) return -2;
    if (rot < 0 || rot >= D || (rot && ntaps % D)) return -2;
    const int jt = (ntaps + D - 1) / D;
    if (!pow2(jsplit) || jsplit > SXO_MAX_GROUPS || jt % jsplit) return -2;
    if (cw < 1 || D % cw || D / cw > SXO_MAX_GROUPS) return -2;
    if (n_out && (m0 + (int64_t)n_out - 1) * D >= (int64_t)n_x) return -1;
    return 0;
}

int sxo_decim_f32_rot(const float *h, int ntaps, int D, int jsplit, int cw, int rot, const float *x,
                      size_t n_x, int64_t m0, size_t n_out, float *y)
{
    const int rc = decim_check(ntaps, D, jsplit, cw, rot, n_x, m0, n_out);
    if (rc) return rc;
    decim_f32_range(h, ntaps, D, jsplit, cw, rot, x, m0, m0 + (int64_t)n_out, m0, y);
    return 0;
}

int sxo_decim_f32(const float *h, int ntaps, int D, int jsplit, int cw, const float *x,
                  size_t n_x, int64_t m0, size_t n_out, float *y)
{
    return sxo_decim_f32_rot(h, ntaps, D, jsplit, cw, 0, x, n_x, m0, n_out, y);
}

int sxo_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

int sxo_decim_f32_mt(const float *h, int ntaps, int D, int jsplit, int cw, const float *x,
                     size_t n_x, int64_t m0, size_t n_out, float *y, int threads)
{
    return sxo_decim_f32_rot_mt(h, ntaps, D, jsplit, cw, 0, x, n_x, m0, n_out, y, threads);
}

int sxo_decim_f32_rot_mt(const float *h, int ntaps, int D, int jsplit, int cw, int rot, const float *x,
                         size_t n_x, int64_t m0, size_t n_out, float *y, int threads)
{
    const int rc = decim_check(ntaps, D, jsplit, cw, rot, n_x, m0, n_out);
    if (rc) return rc;
    const int64_t chunk = 4096;
    const int64_t nchunks = ((int64_t)n_out + chunk - 1) / chunk;
    if (threads < 1) threads = 1;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads)
#endif
    for (int64_t c = 0; c < nchunks; c++) {
        const int64_t b = m0 + c * chunk;
        int64_t e = b + chunk;
        if (e > m0 + (int64_t)n_out) e = m0 + (int64_t)n_out;
        decim_f32_range(h, ntaps, D, jsplit, cw, rot, x, b, e, m0, y);
    }
    return 0;
}

static int interp_f32_range(const float *h, int L, int groups, int gl, const float *x, size_t n_x, int64_t n0,
                            size_t o_begin, size_t o_end, float *y)
{
    for (size_t o = o_begin; o < o_end; o++) {
        const int64_t n = n0 + (int64_t)o;
        const int64_t q = n / L;
        const int r = (int)(n % L);
        if (q >= (int64_t)n_x) return -1;
        float ti[SXO_MAX_GROUPS], tq[SXO_MAX_GROUPS];
        for (int p = 0; p < groups; p++) {
            float ai = 0.0f, aq = 0.0f;
            for (int j = (p + 1) * gl - 1; j >= p * gl; j--) {
                const int64_t idx = q - j;
                const float xi = idx < 0 ? 0.0f : x[2 * idx];
                const float xq = idx < 0 ? 0.0f : x[2 * idx + 1];
                const float c = h[j * L + r];
                ai = fmaf(c, xi, ai);
                aq = fmaf(c, xq, aq);
            }
            ti[p] = ai;
            tq[p] = aq;
        }
        y[2 * o] = tree_sum(ti, groups);
        y[2 * o + 1] = tree_sum(tq, groups);
    }
    return 0;
}

int sxo_interp_f32(const float *h, int ntaps, int L, int groups, const float *x,
                   size_t n_x, int64_t n0, size_t n_out, float *y)
{
    const int jt = ntaps / L;
    if (ntaps < 1 || L < 1 || ntaps % L || n0 < 0) return -2;
    if (!pow2(groups) || groups > SXO_MAX_GROUPS || jt % groups) return -2;
    return interp_f32_range(h, L, groups, jt / groups, x, n_x, n0, 0, n_out, y);
}

/* The same outputs, computed by `threads` threads over blocks of outputs (whole-stream parity checks). */
int sxo_interp_f32_mt(const float *h, int ntaps, int L, int groups, const float *x,
                      size_t n_x, int64_t n0, size_t n_out, float *y, int threads)
{
    const int jt = ntaps / L;
    if (ntaps < 1 || L < 1 || ntaps % L || n0 < 0) return -2;
    if (!pow2(groups) || groups > SXO_MAX_GROUPS || jt % groups) return -2;
    if (n_out && (n0 + (int64_t)n_out - 1) / L >= (int64_t)n_x) return -1;
    const int64_t chunk = 16384;
    const int64_t nchunks = ((int64_t)n_out + chunk - 1) / chunk;
    if (threads < 1) threads = 1;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads)
#endif
    for (int64_t c = 0; c < nchunks; c++) {
        const size_t b = (size_t)(c * chunk);
        size_t e = b + (size_t)chunk;
        if (e > n_out) e = n_out;
        (void)interp_f32_range(h, L, groups, jt / groups, x, n_x, n0, b, e, y);
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* IEEE binary16 <-> binary32, round-to-nearest-even, subnormals kept.        */
/* ------------------------------------------------------------------------- */
static uint16_t f32_to_f16_bits(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    const uint32_t sign = (u >> 16) & 0x8000u;
    const uint32_t absu = u & 0x7FFFFFFFu;
    if (absu >= 0x7F800000u) /* inf / nan */
        return (uint16_t)(sign | 0x7C00u | ((absu > 0x7F800000u) ? (0x0200u | ((absu >> 13) & 0x3FFu)) : 0));
    if (absu >= 0x477FF000u) /* rounds to >= 65520 -> inf */
        return (uint16_t)(sign | 0x7C00u);
    if (absu < 0x33000001u) /* <= 2^-25 -> 0 (ties to even gives 0 at exactly 2^-25) */
        return (uint16_t)sign;
    int32_t e = (int32_t)(absu >> 23) - 127;
    uint32_t m = (absu & 0x7FFFFFu) | 0x800000u;
    uint32_t shift, half_bits;
    if (e < -14) { /* subnormal half */
        shift = (uint32_t)(13 + (-14 - e));
        half_bits = 0;
    } else {
        shift = 13;
        half_bits = (uint32_t)(e + 15) << 10;
        m &= 0x7FFFFFu;
    }
    uint32_t q = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1u);
    const uint32_t halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (q & 1u))) q++;
    return (uint16_t)(sign | (half_bits + q));
}

static float f16_bits_to_f32(uint16_t hb)
{
    const uint32_t sign = ((uint32_t)hb & 0x8000u) << 16;
    const uint32_t e = (hb >> 10) & 0x1Fu;
    const uint32_t m = hb & 0x3FFu;
    uint32_t u;
    if (e == 0) {
        if (m == 0) {
            u = sign;
        } else {
            float f = (float)m * (1.0f / 16777216.0f); /* m * 2^-24 */
            memcpy(&u, &f, 4);
            u |= sign;
        }
    } else if (e == 31) {
        u = sign | 0x7F800000u | (m << 13);
    } else {
        u = sign | ((e + 112u) << 23) | (m << 13);
    }
    float f;
    memcpy(&f, &u, 4);
    return f;
}

void sxo_f32_to_f16(const float *src, uint16_t *dst, size_t n)
{
    for (size_t i = 0; i < n; i++) dst[i] = f32_to_f16_bits(src[i]);
}

void sxo_f16_to_f32(const uint16_t *src, float *dst, size_t n)
{
    for (size_t i = 0; i < n; i++) dst[i] = f16_bits_to_f32(src[i]);
}

/* ------------------------------------------------------------------------- */
/* a-1: readStream position rules, SoapySX.cpp:897-966.                       */
/* ------------------------------------------------------------------------- */
#define SXO_HAS_TIME 4

void sxo_rx_step(int64_t position, int64_t pcm_avail, uint64_t period, uint64_t buffer,
                 size_t num_elems, long timeout_us, double rate, sxo_stream_result *r)
{
    memset(r, 0, sizeof(*r));
    /* overrun: more available than the ring holds -> old samples were
     * overwritten; skip whole periods plus a 2-period margin (:910-927). */
    if (pcm_avail > (int64_t)buffer) {
        const uint64_t overwritten = (uint64_t)pcm_avail - buffer;
        uint64_t skip = (overwritten / period + 2) * period;
        /* snd_pcm_forward cannot move past what is available */
        if ((int64_t)skip > pcm_avail) skip = (uint64_t)pcm_avail;
        position += (int64_t)skip;
        pcm_avail -= (int64_t)skip;
        r->skipped = (int64_t)skip;
    }
    uint64_t length = num_elems;
    if (timeout_us <= 0) { /* non-blocking clamp, :934-942 */
        if (pcm_avail <= 0) length = 0;
        else if ((uint64_t)pcm_avail < length) length = (uint64_t)pcm_avail;
    }
    if (length > 0) { /* :947-959 */
        r->time_ns = sxo_ticks_to_time_ns(position, rate);
        r->flags |= SXO_HAS_TIME;
        position += (int64_t)length;
        r->ret = (int)length;
    }
    r->length = (int64_t)length;
    r->position = position;
}

/* a-2: writeStream position rules, SoapySX.cpp:989-1104. */
void sxo_tx_step(int64_t position, int64_t pcm_avail, int64_t pcm_delay, uint64_t period,
                 size_t num_elems, int flags, long long time_ns, long timeout_us,
                 double rate, sxo_stream_result *r)
{
    memset(r, 0, sizeof(*r));
    const int64_t playback_position = position - pcm_delay; /* :1000 */
    int64_t write_position;
    uint64_t length = num_elems;
    if (flags & SXO_HAS_TIME) { /* :1009-1023 */
        write_position = sxo_time_ns_to_ticks(time_ns, rate);
        if (playback_position - write_position > 0) {
            r->discarded = 1;
            r->ret = (int)length;
            r->length = 0;
            r->position = position;
            return;
        }
    } else { /* :1024-1038 */
        write_position = position;
        int64_t diff = playback_position - write_position;
        if (diff > 0) {
            diff = (diff / (int64_t)period + 2) * (int64_t)period;
            write_position += diff;
        }
    }
    int64_t posdiff = write_position - position; /* :1043-1073 */
    if (posdiff > 0) {
        position += posdiff;
        pcm_avail -= posdiff;
        r->skipped = posdiff;
    }
    if (timeout_us <= 0) { /* :1076-1085 */
        if (pcm_avail <= 0) length = 0;
        else if ((uint64_t)pcm_avail < length) length = (uint64_t)pcm_avail;
    }
    if (length > 0) { /* :1092-1100 */
        position += (int64_t)length;
        r->ret = (int)length;
    }
    r->length = (int64_t)length;
    r->position = position;
}

/* ------------------------------------------------------------------------- */
/* f-4: gain distribution and tuning word, SoapySX.cpp:50-63, :1236-1394.     */
/* ------------------------------------------------------------------------- */
static int range_quantize(double lo, double hi, double step, double value)
{
    double v = value < lo ? lo : value;
    v = v > hi ? hi : v;
    return (int)round((v - lo) / step);
}

static double range_value(double lo, double hi, double step, int q)
{
    double v = lo + step * (double)q;
    v = v < lo ? lo : v;
    return v > hi ? hi : v;
}

void sxo_gain_split(int direction, double value, double *coarse_db, double *fine_db)
{
    if (direction == 1) {
        /* LNA: 3-bit field, not linear: q <= 6 -> field 6 - q/2, q == 7 -> 2, q == 8 -> 1 (:1320-1327);
         * read back through map {0,8,7,6,4,2,0,0} (:1355) */
        static const int map[8] = {0, 8, 7, 6, 4, 2, 0, 0};
        const int q = range_quantize(0.0, 48.0, 6.0, value - 12.0);
        const int field = q <= 6 ? 6 - q / 2 : (q == 7 ? 2 : 1);
        const double lna = range_value(0.0, 48.0, 6.0, map[field]);
        const int pq = range_quantize(0.0, 30.0, 2.0, value - lna) & 0xF;     /* 4-bit PGA field */
        *coarse_db = lna;
        *fine_db = range_value(0.0, 30.0, 2.0, pq);
    } else {
        const int q = range_quantize(0.0, 9.0, 3.0, value - 26.0) & 0x7;      /* 3-bit DAC field */
        const double dac = range_value(0.0, 9.0, 3.0, q);
        const int mq = range_quantize(0.0, 30.0, 2.0, value - dac) & 0xF;     /* 4-bit MIXER field */
        *coarse_db = dac;
        *fine_db = range_value(0.0, 30.0, 2.0, mq);
    }
}

double sxo_quantize_frequency(double master_clock, double frequency, unsigned *word)
{
    const double step = master_clock * (1.0 / 1048576.0);
    const unsigned w = (unsigned)range_quantize(0.0, step * 16777215.0, step, frequency);
    if (word) *word = w;
    return step * (double)w;
}
