/*
 * sx_oracle.h -- CPU oracle for the sx resampling / stream path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (sxxcvr_amd/) never links, imports or
 * falls back to it.
 *
 * PARITY STATUS (read before trusting):
 *  - FIR decimator / interpolator: the reference (tejeez/sxxcvr) contains NO
 *    software FIR at all -- decimation happens inside the SX1255 chip, which
 *    SoapySX.cpp:180-208 / :1197-1208 only configures.  These functions are
 *    therefore the build's OWN definition of the filter; "parity unpinned"
 *    with respect to the reference.  They are pinned instead against an
 *    independent implementation (scipy.signal.upfirdn, fp64) by
 *    tests/golden/make_golden.py + tests/test_oracle_golden.py.
 *  - convert_rx / convert_tx: restatement of SoapySX.cpp:103-137.  PINNED BY
 *    THE REFERENCE: these two functions are the only part of the reference's
 *    translation unit that compiles in this image (three standard headers),
 *    and `make -C oracle ref` compiles them as they lie under /root/reference
 *    (with the chip tables that follow them, :139-208: one contiguous span)
 *    into oracle/_ref/libsxref.so.  tests/golden/convert_kat.npz is
 *    that library's output (make_golden.py), and test_oracle_golden.py runs
 *    the restatement against it on 2^20 fresh samples in the container.  Rows
 *    the C++ leaves undefined (a component >= 1.0f or NaN overflows the
 *    float -> int32 conversion at :124-125) follow the saturating definition
 *    below; they are marked in the fixture (tx_defined).
 *  - stream position rules: clean-room restatement of SoapySX.cpp:897-1085.
 *    The reference has no automated tests and no golden vectors, and the rest
 *    of its translation unit cannot be built in this image (it needs the
 *    SoapySDR and ALSA development headers, both absent; no stand-ins are
 *    written), so these stay "parity unpinned"; the known-answer traces in
 *    tests/golden come from hand evaluation of the reference's arithmetic
 *    and from SURVEY.md's probe notes, not from a reference run.
 *  - ticksToTimeNs / timeNsToTicks: third-party (SoapySDR lib/TimeC.cpp,
 *    version unpinned by the reference: find_package(SoapySDR CONFIG) without
 *    a version, SoapySX/CMakeLists.txt:45).  Restated from the published
 *    algorithm; anchored on the reference's call sites SoapySX.cpp:564,570.
 */
#ifndef SX_ORACLE_H
#define SX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- a-T: SoapySDR Time.hpp arithmetic (call sites SoapySX.cpp:564,570) ---- */
long long sxo_ticks_to_time_ns(long long ticks, double rate);
long long sxo_time_ns_to_ticks(long long time_ns, double rate);

/* ---- a-3 / a-4: sample conversion (SoapySX.cpp:103-112, :116-137) ---- */
/* n = number of complex samples; buffers hold 2*n scalars, I/Q interleaved. */
void sxo_convert_rx(const int32_t *src, float *dst, size_t n);
void sxo_convert_rx_mt(const int32_t *src, float *dst, size_t n, int threads);
void sxo_convert_tx(const float *src, int32_t *dst, size_t n, float tx_threshold2);

/* ---- synthetic CF32 IQ source (replaces the ALSA/I2S feed) ---- */
/* out[2*i], out[2*i+1] = I, Q of absolute sample index start+i; index < 0 -> 0. */
void sxo_synth_iq(uint64_t seed, uint32_t channel, int64_t start, size_t n, float *out);
void sxo_synth_iq_mt(uint64_t seed, uint32_t channel, int64_t start, size_t n, float *out, int threads);

/* ---- low-pass prototype: Kaiser-windowed sinc, cutoff 0.5/ratio, sum = gain ---- */
void sxo_design_lowpass(int ntaps, int ratio, double beta, double gain, float *taps);

/* ---- a-0: FIR decimator, y[m] = sum_k h[k] x[m*D - k], x[<0] = 0 ----
 * x holds absolute samples [0, n_x); outputs m in [m0, m0+n_out).
 * Returns 0, or -1 if an output needs a sample >= n_x. */
/* Oracle A: fp64 products and accumulation, ascending k, one final rounding. */
int sxo_decim_f64(const float *h, int ntaps, int D, const float *x, size_t n_x,
                  int64_t m0, size_t n_out, float *y);
/* Oracle B: order-matched fp32 (the numeric contract of the HIP kernels).
 * Tap k = j*D + r (polyphase row j, phase r).  Rows are split into `jsplit`
 * contiguous ranges, phases into column groups of `cw`; inside a (range,
 * column) subset a fmaf chain from +0.0f runs over j DESCENDING, r DESCENDING
 * (ascending sample time); partials are combined by an adjacent-pair tree
 * over the row ranges, then an adjacent-pair tree over the columns (an odd
 * element at the end of a level moves up unchanged: 12 columns -> 6 -> 3 ->
 * (a + b), c -> their sum; the ratios 48 and 96 of the reference's rate table).
 * (jsplit=1, cw=D) is the plain descending-k chain; (2, 4) at D=4 splits the
 * taps into two contiguous halves. */
int sxo_decim_f32(const float *h, int ntaps, int D, int jsplit, int cw, const float *x,
                  size_t n_x, int64_t m0, size_t n_out, float *y);
/* The same with the contract's rotation (0 or 1; 1 for the ratios 48 and 96): rows and columns are those of the slot
 * index k' = j*D + r, which stands for tap k = (k' + rot) mod ntaps and its sample x[m*D - k] -- the same filter, other
 * chains: a row is then the D samples that end BEFORE sample (m - j)*D (whole cache lines of the input), and tap 0
 * opens the last subset's chain.  Needs ntaps % D == 0. */
int sxo_decim_f32_rot(const float *h, int ntaps, int D, int jsplit, int cw, int rot, const float *x,
                      size_t n_x, int64_t m0, size_t n_out, float *y);
int sxo_decim_f32_rot_mt(const float *h, int ntaps, int D, int jsplit, int cw, int rot, const float *x,
                         size_t n_x, int64_t m0, size_t n_out, float *y, int threads);

/* ---- a-0: FIR interpolator, y[n] = sum_j h[j*L + n%L] x[n/L - j], x[<0]=0 ---- */
int sxo_interp_f64(const float *h, int ntaps, int L, const float *x, size_t n_x,
                   int64_t n0, size_t n_out, float *y);
/* groups split the per-phase taps j in contiguous groups; chain over DESCENDING j. */
int sxo_interp_f32(const float *h, int ntaps, int L, int groups, const float *x,
                   size_t n_x, int64_t n0, size_t n_out, float *y);

/* ---- CF16 storage variant (config 5): IQ stored as IEEE half, math in fp32 ---- */
void sxo_f32_to_f16(const float *src, uint16_t *dst, size_t n_scalars);
void sxo_f16_to_f32(const uint16_t *src, float *dst, size_t n_scalars);

/* ---- a-1 / a-2: stream position rules (SoapySX.cpp:897-966, :989-1104) ----
 * Pure restatement of the arithmetic between snd_pcm_avail_delay() and the
 * final snd_pcm_readi()/writei(): given what ALSA would report, say what the
 * reference does with its position counter. */
typedef struct {
    int64_t position;      /* AlsaPcm::position after the call */
    int64_t skipped;       /* samples forwarded because of overrun / underrun / timed gap */
    int64_t length;        /* samples actually read / written (after non-blocking clamp) */
    long long time_ns;     /* RX: timestamp reported; TX: unused */
    int flags;             /* RX: output flags */
    int ret;               /* return value of readStream / writeStream */
    int discarded;         /* TX: 1 if a timed write in the past was dropped */
} sxo_stream_result;

void sxo_rx_step(int64_t position, int64_t pcm_avail, uint64_t period, uint64_t buffer,
                 size_t num_elems, long timeout_us, double rate, sxo_stream_result *r);
void sxo_tx_step(int64_t position, int64_t pcm_avail, int64_t pcm_delay, uint64_t period,
                 size_t num_elems, int flags, long long time_ns, long timeout_us,
                 double rate, sxo_stream_result *r);

/* ---- f-4: overall gain -> element gains (SoapySX.cpp:1291-1394) ----
 * direction: 1 = RX (coarse "LNA" 0..48 step 6, fine "PGA" 0..30 step 2, PGA target 12 dB),
 *            0 = TX (coarse "DAC" 0..9 step 3, fine "MIXER" 0..30 step 2, MIXER target 26 dB).
 * Returns the element gains the reference ends up with, incl. the non-linear LNA field. */
void sxo_gain_split(int direction, double value, double *coarse_db, double *fine_db);
/* frequency -> 24-bit tuning word -> frequency (SoapySX.cpp:1236-1272) */
double sxo_quantize_frequency(double master_clock, double frequency, unsigned *word);

/* ---- CPU baseline helper: multi-threaded oracle B (OpenMP over output blocks) ---- */
int sxo_decim_f32_mt(const float *h, int ntaps, int D, int jsplit, int cw, const float *x,
                     size_t n_x, int64_t m0, size_t n_out, float *y, int threads);
int sxo_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
