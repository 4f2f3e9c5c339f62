/*
 * sx_device.h -- C ABI of the SoapySDR "driver=sx" Device (OUTER drop-in
 * boundary).  The C++ plugin itself is
 * sxxcvr_amd/csrc/SoapySXHip.cpp, a SoapySDR::Device subclass registered as
 * "sx"; this header is the flat C view of the same object that a non-C++ host
 * (Python ctypes, cgo, JNI) binds, one function per SoapySDR::Device virtual
 * that tejeez/sxxcvr overrides.  Names and argument order follow SoapySDR's
 * own C API (SoapySDRDevice_*), prefixed sx_device_ so that both can coexist.
 *
 * Reference method replaced by each entry (SoapySX/SoapySX.cpp = SX.cpp):
 *   sx_device_enumerate ............ findDevice                 SX.cpp:1629-1642
 *   sx_device_make / _unmake ....... makeDevice / ~SoapySX      SX.cpp:1647-1651, :724-734
 *   sx_device_setup_stream ......... setupStream                SX.cpp:740-794
 *   sx_device_close_stream ......... closeStream                SX.cpp:796-801
 *   sx_device_activate_stream ...... activateStream             SX.cpp:803-830
 *   sx_device_deactivate_stream .... deactivateStream           SX.cpp:832-859
 *   sx_device_get_stream_mtu ....... getStreamMTU               SX.cpp:861-866
 *   sx_device_read_stream .......... readStream                 SX.cpp:868-967
 *   sx_device_write_stream ......... writeStream                SX.cpp:969-1105
 *   sx_device_get_hardware_time .... getHardwareTime            SX.cpp:1107-1139
 *   sx_device_has_hardware_time .... hasHardwareTime            SX.cpp:1618-1623
 *   sx_device_list_sample_rates .... listSampleRates            SX.cpp:1145-1153
 *   sx_device_set/get_sample_rate .. setSampleRate/getSampleRate SX.cpp:1166-1219
 *   sx_device_get_num_channels ..... getNumChannels             SX.cpp:1591-1595
 *   sx_device_get_info ............. getDriverKey, getHardwareKey, getHardwareInfo,
 *                                    getStreamFormats, getNativeStreamFormat
 *                                                               SX.cpp:1567-1616
 *   sx_device_set/get_frequency, _gain, _antenna .............. SX.cpp:1225-1466 (register shadow)
 *   sx_device_write/read_setting ... writeSetting               SX.cpp:1472-1493
 *       As in the reference: "PA" = ON | OFF | AUTO is the one key it knows; any other key and any other PA value is IGNORED
 *       (no error); the reference does not override readSetting, so an unknown key reads as "" (SoapySDR's default).  Keys this
 *       build adds: CLOCK_ADVANCE, TX_CAPTURE_CHANNEL (write); CLOCK_NOW, RX_POSITION, TX_POSITION, TX_WRITTEN, TX_PTT_SAMPLES,
 *       RX_DIRECT_SAMPLES, TX_DIRECT_SAMPLES, RX_DECIM, TX_INTERP, RX_NTAPS, SEED, TX_CAPTURE_CHANNEL, PA (read).
 *   sx_device_setup_stream: the channel list is ignored on a one-channel device (SX.cpp:747); checks in the reference's
 *       order: format, a running stream, already set up (SX.cpp:750-764).
 *
 * Error model: where the C++ method throws (std::runtime_error in the
 * reference), the C function returns SX_DEVICE_EXCEPTION (or NULL) and the
 * message is available from sx_device_last_error().  Streaming calls return
 * the SoapySDR codes unchanged (>= 0 sample counts, negative SOAPY_SDR_*).
 */
#ifndef SX_DEVICE_H
#define SX_DEVICE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SX_DEVICE_EXCEPTION (-1000)

#define SX_SOAPY_SDR_TX 0
#define SX_SOAPY_SDR_RX 1
#define SX_SOAPY_SDR_END_BURST 2
#define SX_SOAPY_SDR_HAS_TIME 4
#define SX_SOAPY_SDR_TIMEOUT (-1)
#define SX_SOAPY_SDR_STREAM_ERROR (-2)
#define SX_SOAPY_SDR_CORRUPTION (-3)
#define SX_SOAPY_SDR_OVERFLOW (-4)
#define SX_SOAPY_SDR_NOT_SUPPORTED (-5)
#define SX_SOAPY_SDR_TIME_ERROR (-6)
#define SX_SOAPY_SDR_UNDERFLOW (-7)

typedef struct sx_device sx_device;
typedef struct sx_stream sx_stream;

const char *sx_device_last_error(void);

/* args: "key=value, key=value" markup (SoapySDR KwargsFromString). */
/* Writes the found devices as "k=v, k=v" entries separated by ';' and returns how many. */
int sx_device_enumerate(const char *args, char *out, size_t cap);
sx_device *sx_device_make(const char *args);
int sx_device_unmake(sx_device *dev);

sx_stream *sx_device_setup_stream(sx_device *dev, int direction, const char *format, const size_t *channels,
                                  size_t num_chans, const char *args);
int sx_device_close_stream(sx_device *dev, sx_stream *stream);
long sx_device_get_stream_mtu(sx_device *dev, sx_stream *stream);
int sx_device_activate_stream(sx_device *dev, sx_stream *stream, int flags, long long time_ns, size_t num_elems);
int sx_device_deactivate_stream(sx_device *dev, sx_stream *stream, int flags, long long time_ns);
int sx_device_read_stream(sx_device *dev, sx_stream *stream, void *const *buffs, size_t num_elems, int *flags,
                          long long *time_ns, long timeout_us);
int sx_device_write_stream(sx_device *dev, sx_stream *stream, const void *const *buffs, size_t num_elems, int *flags,
                           long long time_ns, long timeout_us);

int sx_device_has_hardware_time(sx_device *dev, const char *what);
int sx_device_get_hardware_time(sx_device *dev, const char *what, long long *time_ns);

int sx_device_list_sample_rates(sx_device *dev, int direction, size_t channel, double *rates, size_t cap);
int sx_device_set_sample_rate(sx_device *dev, int direction, size_t channel, double rate);
double sx_device_get_sample_rate(sx_device *dev, int direction, size_t channel);
int sx_device_get_num_channels(sx_device *dev, int direction);

/* what: "driver_key", "hardware_key", "hardware_info", "stream_formats", "native_stream_format" */
int sx_device_get_info(sx_device *dev, const char *what, int direction, char *out, size_t cap);

int sx_device_set_frequency(sx_device *dev, int direction, size_t channel, double hz);
double sx_device_get_frequency(sx_device *dev, int direction, size_t channel);
int sx_device_set_gain(sx_device *dev, int direction, size_t channel, double db);
double sx_device_get_gain(sx_device *dev, int direction, size_t channel);
int sx_device_set_antenna(sx_device *dev, int direction, size_t channel, const char *name);
int sx_device_get_antenna(sx_device *dev, int direction, size_t channel, char *out, size_t cap);

/* named gain elements ("LNA", "PGA", "DAC", "MIXER") and raw SX1255 register shadow, SX.cpp:1279-1368, :1501-1561 */
int sx_device_set_gain_element(sx_device *dev, int direction, size_t channel, const char *name, double db);
double sx_device_get_gain_element(sx_device *dev, int direction, size_t channel, const char *name);
/* getGainRange (SX.cpp:1291-1306): name == NULL or "" -> the overall range; out = {minimum, maximum, step} */
int sx_device_get_gain_range(sx_device *dev, int direction, size_t channel, const char *name, double out[3]);
int sx_device_list(sx_device *dev, const char *what, int direction, char *out, size_t cap); /* "gains", "antennas" */
int sx_device_write_registers(sx_device *dev, const char *name, unsigned addr, const unsigned *values, size_t n);
int sx_device_read_registers(sx_device *dev, const char *name, unsigned addr, unsigned *values, size_t n);

int sx_device_write_setting(sx_device *dev, const char *key, const char *value);
int sx_device_read_setting(sx_device *dev, const char *key, char *out, size_t cap);

/* Synthetic sink: copy DAC-rate CF32 samples [dac_pos, dac_pos+n) that the TX
 * interpolator produced (only the most recent ring is retained). */
int sx_device_tx_capture(sx_device *dev, long long dac_pos, size_t n, float *dst);

/* SoapySDR time helpers, as the applications call them (example/plot_rxtx_response.py:97). */
long long sx_ticks_to_time_ns(long long ticks, double rate);
long long sx_time_ns_to_ticks(long long time_ns, double rate);

/* Log capture for tests: messages at or above `level` are appended to an
 * internal buffer; sx_device_drain_log copies and clears it. */
int sx_device_set_log_level(int level);
int sx_device_drain_log(char *out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
