/*
 * sxfir.h -- C ABI of the MI355X (gfx950) polyphase FIR resampling path that
 * sits behind SoapySX::readStream / writeStream.
 *
 * This is the INNER drop-in boundary: host C++ (the
 * SoapySDR Device in sxxcvr_amd/csrc/SoapySXHip.cpp, or any FFI: ctypes, cgo,
 * JNI) calls hand-written HIP kernels through these plain-C entry points.
 * Plain pointers and sizes only; no C++ or torch types; no exceptions cross
 * it.  Every function returns 0 (SXFIR_OK) or a negative SXFIR_E* code and
 * leaves a message for sxfir_last_error().
 *
 * What each entry point replaces in the reference (tejeez/sxxcvr,
 * SoapySX/SoapySX.cpp = "SX.cpp"):
 *
 *   sxfir_create ..................... the SX1255 decimator/interpolator
 *       configuration written by setSampleRate (SX.cpp:1166-1209, register
 *       table SX.cpp:180-208): the reference only programs the chip's
 *       divider; here the filter itself is built.
 *   sxfir_reset ...................... AlsaPcm::reset (SX.cpp:419-432):
 *       position = 0, stream restarted (history cleared).
 *   sxfir_decimate ................... the on-chip RX decimator + the I2S/ALSA
 *       capture that feeds snd_pcm_readi (SX.cpp:948).
 *   sxfir_interpolate ................ the on-chip TX interpolator behind
 *       snd_pcm_writei (SX.cpp:1093).
 *   sxfir_convert_rx_s32 ............. convert_rx_buffer (SX.cpp:103-112).
 *   sxfir_convert_tx_s32 ............. convert_tx_buffer (SX.cpp:116-137).
 *   sxfir_ticks_to_time_ns,
 *   sxfir_time_ns_to_ticks ........... samples_to_timestamp /
 *       timestamp_to_samples (SX.cpp:562-571), i.e. SoapySDR::ticksToTimeNs /
 *       timeNsToTicks.
 *   sxfir_synth_fill ................. the synthetic CF32 IQ source that
 *       stands in for the SX1255 ADC stream (no counterpart: hardware).
 *
 * Device buffers are caller-owned.  A plan is bound to one GPU and is not
 * thread-safe (the Device's per-stream mutex serialises calls, mirroring
 * SX.cpp:878 / :979).  `stream` is a hipStream_t passed as void* (NULL = the
 * default stream); all work is enqueued asynchronously on it.
 */
#ifndef SXFIR_H
#define SXFIR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: sxfir_stream_wait_event, sxfir_set_history, timing events, clock probe, host registration and the keying count were added
 *    (round 2).  Two diagnostic entry points of version 1, sxfir_debug_clock and sxfir_debug_stamps, moved out of
 *    libsxfir.so into the profiling build (libsxfir_prof.so, include/sxfir_prof.h): a version-1 client that referenced
 *    them no longer links against the production library; every other version-1 entry point is unchanged.
 * 3: sxfir_set_position added (round 3); nothing removed or changed.
 * 4: sxfir_interpolate_keyed and the sxfir_comm_* gather over RCCL added (round 4); nothing removed or changed.
 * 5: sxfir_device_pci_bus_id, sxfir_comm_query and sxfir_contract_rotation added (round 5); the sxfir_comm_* calls now leave the calling thread's
 *    current GPU as they found it; nothing removed.  ONE BEHAVIOUR CHANGE: for the decimators by 48 and 96 with 32 taps per phase
 *    sxfir_contract() now reports (2, 4) where ABI 4 reported (1, ratio), and the pair alone no longer states the contract for
 *    them: it holds under the rotation sxfir_contract_rotation() reports (1 for these two shapes, 0 elsewhere).  A caller that
 *    feeds only (jsplit, cw) to its own order-matched check gets other bits for /48 and /96, without an error: read the rotation.
 * 6: sxfir_launch_geometry added (round 6).  Small calls at the reference's slowest rates launch differently -- /48, /96 deal (tile,
 *    block) work items while a call has at most 8 x the chip's workgroup slots in tiles (a plan of those shapes owns 48 / 96 MiB of
 *    scratch for the hand-off), x32 .. x96 deal (tile, phase block) items up to 4 x slots -- same contract, same bits; nothing removed
 *    or changed. */
#define SXFIR_ABI_VERSION 6

enum {
    SXFIR_OK = 0,
    SXFIR_EINVAL = -1,      /* bad argument */
    SXFIR_EHIP = -2,        /* HIP runtime error (see sxfir_last_error) */
    SXFIR_ENOMEM = -3,
    SXFIR_EUNSUPPORTED = -4,
    SXFIR_ENODEVICE = -5    /* no gfx950 device visible: the product never falls back to a CPU path */
};

enum { SXFIR_DECIMATE = 0, SXFIR_INTERPOLATE = 1 };

/* IQ storage format in HBM.  Arithmetic is always fp32.
 * SXFIR_CF16: IEEE half pairs (4 bytes per sample) in and out; every input half is widened to float exactly (a NaN half is a NaN
 * float; which NaN is unspecified: the tiled decimators let the texture path widen on the way into LDS, which hands over the
 * canonical quiet NaN), the outputs are rounded to half once, to nearest even.
 * SXFIR_S32 is the reference's on-wire format (S32_LE I2S words, I then Q, 8 bytes per sample) fused
 * into the resampling kernels: a DECIMATE plan reads S32 words (value = 2^-31 * word, convert_rx_buffer,
 * SX.cpp:103-112) and writes CF32; an INTERPOLATE plan reads CF32 and writes S32 words with the
 * transmitter-keying bits of convert_tx_buffer (SX.cpp:116-137; threshold: sxfir_set_tx_threshold). */
enum { SXFIR_CF32 = 0, SXFIR_CF16 = 1, SXFIR_S32 = 2 };

/* Kernel selection (for tests / profiling).  AUTO picks the LDS-tiled kernel
 * whenever the shape allows it and the generic one-output-per-thread kernel
 * otherwise; both implement the same numeric contract. */
enum { SXFIR_KERNEL_AUTO = 0, SXFIR_KERNEL_TILED = 1, SXFIR_KERNEL_GENERIC = 2 };

typedef struct sxfir_plan sxfir_plan;

int sxfir_abi_version(void);
const char *sxfir_last_error(void);
int sxfir_device_count(int *count);
/* name: at least 64 bytes.  arch: at least 32 bytes (e.g. "gfx950"). */
int sxfir_device_info(int device, char *name, char *arch, int *compute_units, size_t *hbm_bytes);
/* PCI address of GPU `device` ("0000:05:00.0": domain:bus:device.function), bdf: at least 16 bytes.  What tells two
 * ranks' GPUs apart on a node (bench.py records one per rank). */
int sxfir_device_pci_bus_id(int device, char *bdf, size_t bdf_bytes);

/* Build a resampler for `nchan` independent channels on GPU `device`
 * (-1 = the calling thread's current HIP device).
 *   mode   SXFIR_DECIMATE: y[m] = sum_k taps[k] x[m*ratio - k]
 *          SXFIR_INTERPOLATE: y[n] = sum_j taps[j*ratio + n%ratio] x[n/ratio - j]
 *   taps   host pointer, ntaps floats (copied)
 *   fmt    SXFIR_CF32 or SXFIR_CF16 (input and output alike), or SXFIR_S32 (wire words on the
 *          hardware side, CF32 on the caller's side; see the format enum above)
 * Samples before the start of the stream are zero; the last ntaps-1 input
 * samples persist across calls (per channel) until sxfir_reset. */
int sxfir_create(sxfir_plan **plan, int mode, const float *taps, int ntaps, int ratio,
                 int nchan, int fmt, int device);
int sxfir_destroy(sxfir_plan *plan);
int sxfir_reset(sxfir_plan *plan, void *stream);
/* Seed the filter state from samples already in device memory: the plan's history becomes the LAST samples of
 * the block [src_dev, src_dev + n) of every channel (channel c at c*stride samples; n >= the plan's history
 * length: ntaps for a decimator, ntaps/ratio for an interpolator), as if that block had just been processed.
 * With it consecutive blocks of one stream can be handed to SEVERAL plans (on several HIP streams or GPUs): block
 * k+1 needs nothing from block k but the tail of its INPUT, which is in memory before either runs
 * (sxxcvr_amd.PipelinedResampler).  No speed-up on one GPU: LABBOOK.md 7. */
int sxfir_set_history(sxfir_plan *plan, const void *src_dev, size_t n, size_t stride, void *stream);
/* Place the plan at sample `consumed` of its input stream (the count sxfir_position reports as consumed; the
 * outputs produced so far follow from it: ceil(consumed / D) for a decimator, consumed * L for an interpolator).
 * A decimator's phase -- which input of the next call yields the next output, SX.cpp:950's position arithmetic --
 * depends on it, so a plan that takes over a stream in the middle (sxfir_set_history) must be told where it is
 * unless the blocks are multiples of the ratio. */
int sxfir_set_position(sxfir_plan *plan, int64_t consumed);
int sxfir_set_kernel(sxfir_plan *plan, int kernel);
/* Squared-magnitude threshold above which an SXFIR_S32 interpolator sets the two low bits of the I
 * word (tx_threshold2 of SX.cpp:540-542, :132-133).  Default 1e-6 (threshold 1e-3, SX.cpp:767). */
int sxfir_set_tx_threshold(sxfir_plan *plan, float tx_threshold2);

/* The numeric contract of this plan, for an order-matched CPU check:
 * decimator: taps split into `jsplit` contiguous ranges of polyphase rows j
 * (k = j*ratio + r) and column groups of `cw` phases r; interpolator: `jsplit`
 * contiguous ranges of j, cw = 1.  See DESIGN.md "Numeric contract". */
int sxfir_contract(const sxfir_plan *plan, int *jsplit, int *cw);
/* ... and its rotation: 0 for every shape but the decimators by 48 and 96 with 32 taps per phase (the reference's rates
 * master clock / 768 and / 1536, SoapySX.cpp:180-208), where it is 1: the rows and columns of the contract are then
 * those of the slot index k' = j*ratio + r, and slot k' holds tap k = (k' + 1) mod ntaps with its sample x[m*ratio - k].
 * The filter is the same sum; a row of the picture is then the `ratio` samples that end BEFORE sample (m - j)*ratio --
 * whole 128-byte lines of the input -- and tap 0 opens the last subset's chain. */
int sxfir_contract_rotation(const sxfir_plan *plan, int *rot);

/* Absolute count of input samples consumed / output samples produced since
 * the last reset (per channel; all channels advance together). */
int sxfir_position(const sxfir_plan *plan, int64_t *consumed, int64_t *produced);

/* Number of outputs a call with n_in new input samples would produce now. */
int sxfir_outputs_for(const sxfir_plan *plan, size_t n_in, size_t *n_out);

/* Streaming decimation.  in_dev: channel c's samples start at
 * in_dev + c*in_stride samples (complex samples of the plan's format);
 * out_dev likewise with out_stride.  Writes *n_out (may be NULL). */
int sxfir_decimate(sxfir_plan *plan, const void *in_dev, size_t n_in, size_t in_stride,
                   void *out_dev, size_t out_stride, size_t *n_out, void *stream);

/* Streaming interpolation: n_in input samples -> n_in*ratio outputs. */
int sxfir_interpolate(sxfir_plan *plan, const void *in_dev, size_t n_in, size_t in_stride,
                      void *out_dev, size_t out_stride, size_t *n_out, void *stream);

/* sxfir_interpolate with convert_tx_buffer's transmitter-keying count (SX.cpp:132-133) taken in the same pass: adds
 * to *counter (8-byte aligned word in DEVICE memory) how many of channel 0's input samples
 * [key_first, key_first + key_count) reach the plan's squared-magnitude threshold (sxfir_set_tx_threshold; the
 * rule is sxfir_count_keyed's).  The interpolator has every input sample in LDS anyway, so a pass whose input is
 * device-visible HOST memory (the Device's pinned TX slots) crosses PCIe once instead of twice.  CF32 or S32 plans. */
int sxfir_interpolate_keyed(sxfir_plan *plan, const void *in_dev, size_t n_in, size_t in_stride,
                            void *out_dev, size_t out_stride, size_t *n_out, size_t key_first, size_t key_count,
                            unsigned long long *counter, void *stream);

/* Synthetic CF32/CF16 IQ source: out_dev[c*stride + i] = sample (start+i) of
 * channel (first_channel + c); index < 0 gives zero.  Counter-based, so any
 * range can be regenerated anywhere. */
int sxfir_synth_fill(void *out_dev, size_t n, size_t stride, int nchan, uint64_t seed,
                     uint32_t first_channel, int64_t start, int fmt, void *stream);

/* S32_LE I2S wire format <-> CF32 (n complex samples, device pointers). */
int sxfir_convert_rx_s32(const int32_t *src_dev, float *dst_dev, size_t n, void *stream);
int sxfir_convert_tx_s32(const float *src_dev, int32_t *dst_dev, size_t n, float tx_threshold2,
                         void *stream);
/* Transmitter keying count of convert_tx_buffer (SX.cpp:132-133): adds to *counter the number of the n complex
 * CF32 samples at src whose squared magnitude reaches tx_threshold2.  src: device memory or device-visible
 * (pinned / registered) host memory; counter: 8-byte aligned word in DEVICE memory (sxfir_malloc; read it back
 * with sxfir_memcpy_d2h). */
int sxfir_count_keyed(const float *src, size_t n, float tx_threshold2, unsigned long long *counter, void *stream);
/* CF32 <-> CF16 storage conversion (n complex samples). */
int sxfir_cf32_to_cf16(const float *src_dev, void *dst_dev, size_t n, void *stream);
int sxfir_cf16_to_cf32(const void *src_dev, float *dst_dev, size_t n, void *stream);

/* Position <-> time at `rate` samples per second. */
long long sxfir_ticks_to_time_ns(long long ticks, double rate);
long long sxfir_time_ns_to_ticks(long long time_ns, double rate);

/* Kaiser-windowed-sinc low-pass prototype (host): cutoff 0.5/ratio, sum = gain. */
int sxfir_design_lowpass(int ntaps, int ratio, double beta, double gain, float *taps);

/* Thin device-memory helpers so that a non-HIP host language can drive the
 * library without its own HIP binding. */
int sxfir_malloc(void **dev, size_t bytes);
int sxfir_free(void *dev);
int sxfir_memcpy_h2d(void *dst_dev, const void *src, size_t bytes, void *stream);
int sxfir_memcpy_d2h(void *dst, const void *src_dev, size_t bytes, void *stream);
int sxfir_stream_sync(void *stream);
/* Pinned (page-locked) host memory: staging buffers that hipMemcpyAsync really overlaps with kernels,
 * and streams of the caller's own, so that e.g. an RX and a TX thread do not wait on each other's work
 * (the reference's two PCMs are independent in the same way, SoapySX.cpp:373). */
/* The helpers above and sxfir_synth_fill / the converters act on the calling thread's current GPU
 * (plans remember their own): a host thread that did not create the plan selects it with this. */
int sxfir_set_device(int device);
int sxfir_host_alloc(void **host, size_t bytes);
int sxfir_host_free(void *host);
/* Page-lock memory the caller already owns (e.g. the buffers it passes to readStream) so that the GPU can
 * store into it directly, and undo it.  sxfir_host_device_pointer: the device-side address of [host, host +
 * bytes) when that range is pinned or registered; SXFIR_EUNSUPPORTED (no message) for ordinary pageable
 * memory. */
int sxfir_host_register(void *host, size_t bytes);
int sxfir_host_unregister(void *host);
int sxfir_host_device_pointer(const void *host, size_t bytes, void **dev);
int sxfir_stream_create(void **stream);
int sxfir_stream_destroy(void *stream);
/* Events: completion of the work queued on a stream so far, without draining what is queued later. */
int sxfir_event_create(void **event);
int sxfir_event_destroy(void *event);
/* Events that carry a timestamp: record one before and one after a run of launches on their stream;
 * sxfir_event_elapsed_ms waits for `stop` and returns the GPU time between the two. */
int sxfir_event_create_timing(void **event);
int sxfir_event_elapsed_ms(void *start, void *stop, float *ms);
int sxfir_event_record(void *event, void *stream);
int sxfir_event_sync(void *event);
/* Work queued on `stream` after this call starts only when `event` (recorded on another stream) has fired: the
 * GPU-side ordering between a compute stream and a copy stream, without the host waiting. */
int sxfir_stream_wait_event(void *stream, void *event);

/* ---- The sharded path's one exchange step: gather of the decimated channels to a root GPU over xGMI (RCCL).
 * Channels are independent (the reference is single-channel, SX.cpp:1591-1595); BASELINE config 4 shards 64 of them
 * 8 per GPU and gathers the decimated output.  These entry points let a C / C++ host run that step itself:
 *     per piece:  ncclGroupStart;  root: ncclRecv x (N-1);  peers: ncclSend;  ncclGroupEnd      on `stream`
 * librccl is loaded on first use; without it (or without a GPU) they return SXFIR_EUNSUPPORTED / SXFIR_ENODEVICE.
 *
 * Rank-per-process form: rank 0 calls sxfir_comm_unique_id and hands the SXFIR_COMM_ID_BYTES bytes to the other
 * ranks by any means it has (a file, a socket, the launcher's store); then every rank calls sxfir_comm_init_rank
 * (collective: returns when all nranks have called it).  device -1 = the calling thread's current GPU.
 * Single-process form (one Device per shard in one process): sxfir_comm_init_all fills comms[0..ndev) for the
 * GPUs devices[0..ndev) (NULL = 0..ndev-1), rank i on devices[i]; drive them with sxfir_comm_gather_all. */
#define SXFIR_COMM_ID_BYTES 128
typedef struct sxfir_comm sxfir_comm;
int sxfir_comm_unique_id(void *id);
int sxfir_comm_init_rank(sxfir_comm **comm, const void *id, int nranks, int rank, int device);
int sxfir_comm_init_all(sxfir_comm **comms, int ndev, const int *devices);
int sxfir_comm_destroy(sxfir_comm *comm);
int sxfir_comm_rank(const sxfir_comm *comm, int *rank, int *nranks, int *device);
/* The same three numbers as RCCL itself reports them for the communicator (ncclCommUserRank, ncclCommCount,
 * ncclCommCuDevice) rather than as they were passed in: what a benchmark line should carry.  Any pointer may be NULL.
 * The sxfir_comm_* calls make the communicator's GPU current while they run and restore the caller's before returning. */
int sxfir_comm_query(const sxfir_comm *comm, int *rank, int *nranks, int *device);
/* Every rank contributes `bytes` bytes at send_dev; on the root, rank r's block lands at recv_dev +
 * r * recv_stride_bytes (recv_dev is ignored elsewhere; the root's own block is copied device-to-device on the
 * same stream unless send_dev already is its place in recv_dev).  chunk_bytes > 0 cuts the transfer into pieces of
 * that size, one RCCL group each (use a multiple of one channel's bytes: the root can then consume a step's first
 * channels while the rest are on the links, and a piece queues behind the kernel that produced it without holding
 * the whole block back); 0 = one piece.  Asynchronous on `stream`. */
int sxfir_comm_gather(sxfir_comm *comm, const void *send_dev, void *recv_dev, size_t bytes,
                      size_t recv_stride_bytes, int root, size_t chunk_bytes, void *stream);
/* The same for the single-process form, all ranks driven by the calling thread: send_dev[i] and streams[i]
 * (NULL = default streams) belong to rank i; recv_dev lives on the root's GPU. */
int sxfir_comm_gather_all(sxfir_comm *const *comms, int ndev, const void *const *send_dev, void *recv_dev,
                          size_t bytes, size_t recv_stride_bytes, int root, size_t chunk_bytes,
                          void *const *streams);

/* What a call with n_in new input samples would launch NOW (16-byte aligned output assumed): kernel family, tiles, workgroups
 * and how many of those the chip holds at once.  For callers that size their batches (the Device's chains), for
 * tools/sizebench.py, and for the Device's log line when a plan falls to the generic one-output-per-thread kernels
 * (tiled == 0: any ratio outside {4, 8, 16, 32, 48, 96} or taps_per_phase != 32 -- two orders of magnitude slower). */
typedef struct sxfir_geometry {
    char kernel[64];          /* kernel family, e.g. "decim_blocks_kernel"; "*_generic_kernel" when tiled == 0 */
    int tiled;                /* 1: an LDS-tiled kernel runs this call; 0: the generic kernel */
    int split;                /* work items per tile (decim_blocks_kernel's (tile, block) dealing), else 1 */
    long long tile_samples;   /* wideband samples per tile (a decimator's inputs, an interpolator's outputs) */
    long long n_tiles;        /* tiles per channel */
    long long workgroups;     /* workgroups launched, all channels */
    long long resident;       /* workgroups the chip holds at once (compute units x occupancy) */
} sxfir_geometry;
int sxfir_launch_geometry(const sxfir_plan *plan, size_t n_in, sxfir_geometry *geometry);

/* Timed launches for bench.py: runs `iters` back-to-back decimate passes of
 * the same buffers on `stream` bracketed by hipEvents ON THAT STREAM and
 * returns the mean milliseconds per pass of the resampling kernel. */
int sxfir_time_decimate(sxfir_plan *plan, const void *in_dev, size_t n_in, size_t in_stride,
                        void *out_dev, size_t out_stride, int iters, void *stream, float *ms_per_pass);
int sxfir_time_interpolate(sxfir_plan *plan, const void *in_dev, size_t n_in, size_t in_stride,
                           void *out_dev, size_t out_stride, int iters, void *stream, float *ms_per_pass);

/* Shader clock the chip's power management holds while other work runs: a few single-wave workgroups on a
 * stream of their own compare the shader-cycle counter with the 100 MHz real-time counter for duration_us
 * (asynchronous); sxfir_clock_probe_read waits for them, returns the median in MHz and frees the probe.
 * Start it, run the work to be measured for at least that long, then read. */
typedef struct sxfir_clock_probe sxfir_clock_probe;
int sxfir_clock_probe_start(sxfir_clock_probe **probe, int device, int duration_us);
int sxfir_clock_probe_read(sxfir_clock_probe *probe, double *mhz);

#ifdef __cplusplus
}
#endif
#endif
