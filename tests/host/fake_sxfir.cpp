// TEST-ONLY stand-in for libsxfir.so on a machine without a GPU: the same C ABI (include/sxfir.h) over
// host memory, with the oracle (oracle/sx_oracle.h) doing the arithmetic.  It exists so that the host
// logic above the ABI (GpuChains.hpp: batching, read-ahead, write-behind, channel layout, ring wrap) can be
// exercised by `pytest -m "not gpu"`.  Never built into, loaded by or shipped with the product.
#include <sxfir.h>

#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

extern "C" {
#include "sx_oracle.h"
}

struct sxfir_plan {
    int mode, ntaps, ratio, nchan, fmt;
    std::vector<float> taps;
    std::vector<float> hist;      // nchan * 2 * hist_len floats
    int hist_len;
    float thr2;
};

static std::string g_err;
// "page-locked" ranges: what sxfir_host_alloc returned and what sxfir_host_register was told about
static std::vector<std::pair<const char *, size_t>> g_locked;
static int fail(const char *m) { g_err = m; return SXFIR_EINVAL; }
int g_fake_launches = 0;          // GPU passes (decimate / interpolate calls), read by the test

extern "C" {

int sxfir_abi_version(void) { return SXFIR_ABI_VERSION; }
const char *sxfir_last_error(void) { return g_err.c_str(); }
int sxfir_set_device(int) { return SXFIR_OK; }
int sxfir_malloc(void **dev, size_t bytes) { *dev = std::malloc(bytes ? bytes : 1); return *dev ? SXFIR_OK : SXFIR_ENOMEM; }
int sxfir_free(void *dev) { std::free(dev); return SXFIR_OK; }
int sxfir_host_alloc(void **host, size_t bytes)
{
    const int rc = sxfir_malloc(host, bytes);
    if (rc == SXFIR_OK) g_locked.emplace_back((const char *)*host, bytes ? bytes : 1);
    return rc;
}
static void unlock(const void *host)
{
    for (size_t i = 0; i < g_locked.size(); ++i)
        if (g_locked[i].first == (const char *)host) { g_locked.erase(g_locked.begin() + i); return; }
}
int sxfir_host_free(void *host) { unlock(host); return sxfir_free(host); }
int sxfir_host_register(void *host, size_t bytes) { g_locked.emplace_back((const char *)host, bytes); return SXFIR_OK; }
int sxfir_host_unregister(void *host) { unlock(host); return SXFIR_OK; }
int sxfir_host_device_pointer(const void *host, size_t bytes, void **dev)
{
    *dev = nullptr;
    for (const auto &r : g_locked)
        if ((const char *)host >= r.first && (const char *)host + bytes <= r.first + r.second) { *dev = const_cast<void *>(host); return SXFIR_OK; }
    return SXFIR_EUNSUPPORTED;
}
int sxfir_count_keyed(const float *src, size_t n, float thr2, unsigned long long *counter, void *)
{
    for (size_t i = 0; i < n; ++i) {
        const float ii = src[2 * i] * src[2 * i], qq = src[2 * i + 1] * src[2 * i + 1];
        *counter += (ii + qq >= thr2) ? 1u : 0u;
    }
    return SXFIR_OK;
}
int sxfir_stream_create(void **stream) { *stream = (void *)0x1; return SXFIR_OK; }
int sxfir_stream_destroy(void *) { return SXFIR_OK; }
int sxfir_stream_sync(void *) { return SXFIR_OK; }
int sxfir_event_create(void **event) { *event = (void *)0x2; return SXFIR_OK; }
int sxfir_event_destroy(void *) { return SXFIR_OK; }
int sxfir_event_record(void *, void *) { return SXFIR_OK; }
int sxfir_event_sync(void *) { return SXFIR_OK; }
int sxfir_stream_wait_event(void *, void *) { return SXFIR_OK; }
int sxfir_memcpy_h2d(void *dst, const void *src, size_t bytes, void *) { std::memcpy(dst, src, bytes); return SXFIR_OK; }
int sxfir_memcpy_d2h(void *dst, const void *src, size_t bytes, void *) { std::memcpy(dst, src, bytes); return SXFIR_OK; }

int sxfir_design_lowpass(int ntaps, int ratio, double beta, double gain, float *taps)
{
    sxo_design_lowpass(ntaps, ratio, beta, gain, taps);
    return SXFIR_OK;
}

int sxfir_create(sxfir_plan **out, int mode, const float *taps, int ntaps, int ratio, int nchan, int fmt, int)
{
    if (fmt != SXFIR_CF32) return fail("fake backend: CF32 only");
    sxfir_plan *p = new sxfir_plan();
    p->mode = mode; p->ntaps = ntaps; p->ratio = ratio; p->nchan = nchan; p->fmt = fmt;
    p->taps.assign(taps, taps + ntaps);
    p->hist_len = mode == SXFIR_DECIMATE ? ntaps : ntaps / ratio;
    p->hist.assign((size_t)nchan * 2 * p->hist_len, 0.0f);
    p->thr2 = 0.0f;
    *out = p;
    return SXFIR_OK;
}

int sxfir_destroy(sxfir_plan *p) { delete p; return SXFIR_OK; }
int sxfir_reset(sxfir_plan *p, void *) { std::fill(p->hist.begin(), p->hist.end(), 0.0f); return SXFIR_OK; }
int sxfir_set_history(sxfir_plan *, const void *, size_t, size_t, void *) { g_err = "not in the fake"; return SXFIR_EUNSUPPORTED; }
int sxfir_set_tx_threshold(sxfir_plan *p, float t) { p->thr2 = t; return SXFIR_OK; }

int sxfir_synth_fill(void *out_dev, size_t n, size_t stride, int nchan, uint64_t seed, uint32_t first_channel,
                     int64_t start, int fmt, void *)
{
    if (fmt != SXFIR_CF32) return fail("fake backend: CF32 only");
    for (int c = 0; c < nchan; ++c)
        sxo_synth_iq(seed, first_channel + (uint32_t)c, start, n, (float *)out_dev + 2 * (size_t)c * stride);
    return SXFIR_OK;
}

// one channel: (history ++ block) through the oracle, then the history moves on
static void run_channel(sxfir_plan *p, int c, const float *in, size_t n_in, float *out, size_t n_out)
{
    const int H = p->hist_len;
    std::vector<float> ext(2 * ((size_t)H + n_in));
    float *hist = p->hist.data() + 2 * (size_t)c * H;
    std::memcpy(ext.data(), hist, 8 * (size_t)H);
    std::memcpy(ext.data() + 2 * H, in, 8 * n_in);
    if (p->mode == SXFIR_DECIMATE)
        sxo_decim_f32(p->taps.data(), p->ntaps, p->ratio, 2, 4, ext.data(), (size_t)H + n_in, H / p->ratio, n_out, out);
    else
        sxo_interp_f32(p->taps.data(), p->ntaps, p->ratio, 2, ext.data(), (size_t)H + n_in, (int64_t)H * p->ratio, n_out, out);
    std::memcpy(hist, ext.data() + 2 * n_in, 8 * (size_t)H);
}

int sxfir_decimate(sxfir_plan *p, const void *in_dev, size_t n_in, size_t in_stride, void *out_dev, size_t out_stride,
                   size_t *n_out, void *)
{
    if (p->mode != SXFIR_DECIMATE || n_in % (size_t)p->ratio) return fail("fake backend: whole output blocks only");
    ++g_fake_launches;
    *n_out = n_in / (size_t)p->ratio;
    for (int c = 0; c < p->nchan; ++c)
        run_channel(p, c, (const float *)in_dev + 2 * (size_t)c * in_stride, n_in, (float *)out_dev + 2 * (size_t)c * out_stride,
                    *n_out);
    return SXFIR_OK;
}

int sxfir_interpolate(sxfir_plan *p, const void *in_dev, size_t n_in, size_t in_stride, void *out_dev, size_t out_stride,
                      size_t *n_out, void *)
{
    if (p->mode != SXFIR_INTERPOLATE) return fail("fake backend: not an interpolator");
    ++g_fake_launches;
    *n_out = n_in * (size_t)p->ratio;
    for (int c = 0; c < p->nchan; ++c)
        run_channel(p, c, (const float *)in_dev + 2 * (size_t)c * in_stride, n_in, (float *)out_dev + 2 * (size_t)c * out_stride,
                    *n_out);
    return SXFIR_OK;
}

// the interpolation pass with the keying count of channel 0's samples [key_first, key_first + key_count)
int sxfir_interpolate_keyed(sxfir_plan *p, const void *in_dev, size_t n_in, size_t in_stride, void *out_dev, size_t out_stride,
                            size_t *n_out, size_t key_first, size_t key_count, unsigned long long *counter, void *st)
{
    if (key_first > n_in || key_count > n_in - key_first || !counter) return fail("fake backend: bad keying range");
    const int rc = sxfir_interpolate(p, in_dev, n_in, in_stride, out_dev, out_stride, n_out, st);
    if (rc) return rc;
    return sxfir_count_keyed((const float *)in_dev + 2 * key_first, key_count, p->thr2, counter, st);
}

}  // extern "C"
