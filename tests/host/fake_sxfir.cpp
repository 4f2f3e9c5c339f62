// TEST-ONLY stand-in for libsxfir.so on a machine without a GPU: the same C ABI (include/sxfir.h) over
// host memory, with the oracle (oracle/sx_oracle.h) doing the arithmetic.  It exists so that the host
// logic above the ABI (GpuChains.hpp: batching, read-ahead, write-behind, channel layout, ring wrap; the Device
// module over it) can be exercised by `pytest -m "not gpu"` -- plain, under AddressSanitizer + UBSan and under
// ThreadSanitizer.  Never built into, loaded by or shipped with the product.
//
// It is ASYNCHRONOUS the way the GPU is, and more so: every stream is a queue served by a thread of its own, and
// everything the ABI says happens "on the stream" (kernels, copies, event records, cross-stream waits, the plan's
// filter state) happens there, later -- with a seeded random delay in front of each item (FAKE_SXFIR_JITTER_US,
// default 40) so that work really is still in flight when the calling thread moves on.  The calling thread sees a
// result only through what the ABI offers for that: sxfir_event_sync, sxfir_stream_sync (and sxfir_free /
// sxfir_host_free, which drain every stream first, as hipFree does).  A wait the host code forgot therefore shows
// twice: as wrong samples in the probes, and as a data race under ThreadSanitizer (the waits are the only
// happens-before edges between a stream's thread and the caller).  Stricter than the hardware in one respect: copies
// from or to ordinary (pageable) host memory are queued like any other, where hipMemcpyAsync would stage them before
// returning.
#include <sxfir.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

extern "C" {
#include "sx_oracle.h"
}

namespace {

struct Stream {
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    unsigned long long enqueued = 0, done = 0;
    bool stop = false;
    std::mt19937 rng;
    int jitter_us;
    std::thread worker;

    explicit Stream(unsigned seed) : rng(seed)
    {
        const char *e = std::getenv("FAKE_SXFIR_JITTER_US");
        jitter_us = e ? std::atoi(e) : 40;
        worker = std::thread([this] { serve(); });
    }
    ~Stream()
    {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        cv.notify_all();
        worker.join();
    }
    void serve()
    {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return stop || !q.empty(); });
            if (q.empty()) return;                       // stop, and nothing left to run
            std::function<void()> f = std::move(q.front());
            q.pop_front();
            const int wait_us = jitter_us > 0 ? (int)(rng() % (unsigned)(jitter_us + 1)) : 0;
            lk.unlock();
            if (wait_us > 0 && (rng() & 3) == 0) std::this_thread::sleep_for(std::chrono::microseconds(wait_us));
            f();
            lk.lock();
            ++done;
            cv.notify_all();
        }
    }
    void push(std::function<void()> f)
    {
        {
            std::lock_guard<std::mutex> lk(m);
            q.push_back(std::move(f));
            ++enqueued;
        }
        cv.notify_all();
    }
    void sync()
    {
        std::unique_lock<std::mutex> lk(m);
        const unsigned long long upto = enqueued;
        cv.wait(lk, [&] { return done >= upto; });
    }
};

struct Event {
    std::mutex m;
    std::condition_variable cv;
    unsigned long long recorded = 0, fired = 0;        // generations: a record is "the n-th"; fired = the latest one through
};

std::mutex g_streams_m;
std::vector<Stream *> g_streams;
Stream *g_default = nullptr;                             // the NULL stream
std::atomic<unsigned> g_stream_seed{12345};

Stream *new_stream()
{
    Stream *s = new Stream(g_stream_seed.fetch_add(7919));
    std::lock_guard<std::mutex> lk(g_streams_m);
    g_streams.push_back(s);
    return s;
}

Stream *S(void *st)
{
    if (st) return static_cast<Stream *>(st);
    std::lock_guard<std::mutex> lk(g_streams_m);
    if (!g_default) {
        g_default = new Stream(1);
        g_streams.push_back(g_default);
    }
    return g_default;
}

void drain_all()                                          // hipFree / hipHostFree: an implicit device synchronisation
{
    std::vector<Stream *> all;
    {
        std::lock_guard<std::mutex> lk(g_streams_m);
        all = g_streams;
    }
    for (Stream *s : all) s->sync();
}

thread_local std::string g_err;
// "page-locked" ranges: what sxfir_host_alloc returned and what sxfir_host_register was told about
std::mutex g_locked_m;
std::vector<std::pair<const char *, size_t>> g_locked;
int fail(const char *m) { g_err = m; return SXFIR_EINVAL; }

void unlock(const void *host)
{
    std::lock_guard<std::mutex> lk(g_locked_m);
    for (size_t i = 0; i < g_locked.size(); ++i)
        if (g_locked[i].first == (const char *)host) { g_locked.erase(g_locked.begin() + i); return; }
}

}  // namespace

struct sxfir_plan {
    int mode, ntaps, ratio, nchan, fmt;
    std::vector<float> taps;
    std::vector<float> hist;      // nchan * 2 * hist_len floats; touched on the stream's thread only, like the GPU's history buffer
    int hist_len;
    std::atomic<float> thr2;      // host-side state in the real library too: read when a pass is queued
};

std::atomic<int> g_fake_launches{0};          // GPU passes (decimate / interpolate calls), read by the test

extern "C" {

int sxfir_abi_version(void) { return SXFIR_ABI_VERSION; }
const char *sxfir_last_error(void) { return g_err.c_str(); }
int sxfir_set_device(int) { return SXFIR_OK; }
int sxfir_device_count(int *count) { *count = 1; return SXFIR_OK; }
int sxfir_device_info(int, char *name, char *arch, int *compute_units, size_t *hbm_bytes)
{
    if (name) std::strcpy(name, "fake backend (CPU, test only)");
    if (arch) std::strcpy(arch, "none");
    if (compute_units) *compute_units = 0;
    if (hbm_bytes) *hbm_bytes = 0;
    return SXFIR_OK;
}
int sxfir_malloc(void **dev, size_t bytes) { *dev = std::malloc(bytes ? bytes : 1); return *dev ? SXFIR_OK : SXFIR_ENOMEM; }
int sxfir_free(void *dev) { drain_all(); std::free(dev); return SXFIR_OK; }
int sxfir_host_alloc(void **host, size_t bytes)
{
    *host = std::malloc(bytes ? bytes : 1);
    if (!*host) return SXFIR_ENOMEM;
    std::lock_guard<std::mutex> lk(g_locked_m);
    g_locked.emplace_back((const char *)*host, bytes ? bytes : 1);
    return SXFIR_OK;
}
int sxfir_host_free(void *host) { drain_all(); unlock(host); std::free(host); return SXFIR_OK; }
int sxfir_host_register(void *host, size_t bytes)
{
    std::lock_guard<std::mutex> lk(g_locked_m);
    g_locked.emplace_back((const char *)host, bytes);
    return SXFIR_OK;
}
int sxfir_host_unregister(void *host) { unlock(host); return SXFIR_OK; }
int sxfir_host_device_pointer(const void *host, size_t bytes, void **dev)
{
    *dev = nullptr;
    std::lock_guard<std::mutex> lk(g_locked_m);
    for (const auto &r : g_locked)
        if ((const char *)host >= r.first && (const char *)host + bytes <= r.first + r.second) { *dev = const_cast<void *>(host); return SXFIR_OK; }
    return SXFIR_EUNSUPPORTED;
}

int sxfir_stream_create(void **stream) { *stream = new_stream(); return SXFIR_OK; }
int sxfir_stream_destroy(void *stream)
{
    Stream *s = static_cast<Stream *>(stream);
    if (!s) return SXFIR_OK;
    s->sync();
    {
        std::lock_guard<std::mutex> lk(g_streams_m);
        for (size_t i = 0; i < g_streams.size(); ++i)
            if (g_streams[i] == s) { g_streams.erase(g_streams.begin() + i); break; }
    }
    delete s;
    return SXFIR_OK;
}
int sxfir_stream_sync(void *stream) { S(stream)->sync(); return SXFIR_OK; }
int sxfir_event_create(void **event) { *event = new Event(); return SXFIR_OK; }
int sxfir_event_destroy(void *event) { delete static_cast<Event *>(event); return SXFIR_OK; }
int sxfir_event_record(void *event, void *stream)
{
    Event *e = static_cast<Event *>(event);
    unsigned long long gen;
    {
        std::lock_guard<std::mutex> lk(e->m);
        gen = ++e->recorded;
    }
    S(stream)->push([e, gen] {
        std::lock_guard<std::mutex> lk(e->m);
        if (e->fired < gen) e->fired = gen;
        e->cv.notify_all();
    });
    return SXFIR_OK;
}
int sxfir_event_sync(void *event)
{
    // an event that was never recorded counts as complete (hipEventSynchronize)
    Event *e = static_cast<Event *>(event);
    std::unique_lock<std::mutex> lk(e->m);
    const unsigned long long gen = e->recorded;
    e->cv.wait(lk, [&] { return e->fired >= gen; });
    return SXFIR_OK;
}
int sxfir_stream_wait_event(void *stream, void *event)
{
    Event *e = static_cast<Event *>(event);
    unsigned long long gen;
    {
        std::lock_guard<std::mutex> lk(e->m);
        gen = e->recorded;                               // the record in force when the wait is queued
    }
    S(stream)->push([e, gen] {
        std::unique_lock<std::mutex> lk(e->m);
        e->cv.wait(lk, [&] { return e->fired >= gen; });
    });
    return SXFIR_OK;
}
int sxfir_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream)
{
    S(stream)->push([dst, src, bytes] { std::memcpy(dst, src, bytes); });
    return SXFIR_OK;
}
int sxfir_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream)
{
    S(stream)->push([dst, src, bytes] { std::memcpy(dst, src, bytes); });
    return SXFIR_OK;
}

static void count_keyed_now(const float *src, size_t n, float thr2, unsigned long long *counter)
{
    unsigned long long c = 0;
    for (size_t i = 0; i < n; ++i) {
        const float ii = src[2 * i] * src[2 * i], qq = src[2 * i + 1] * src[2 * i + 1];
        c += (ii + qq >= thr2) ? 1u : 0u;
    }
    *counter += c;                                       // device memory: streams' threads and copies only
}
int sxfir_count_keyed(const float *src, size_t n, float thr2, unsigned long long *counter, void *stream)
{
    S(stream)->push([=] { count_keyed_now(src, n, thr2, counter); });
    return SXFIR_OK;
}

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

int sxfir_launch_geometry(const sxfir_plan *p, size_t n_in, sxfir_geometry *g)
{
    // the real library's rule for which shapes have an LDS-tiled kernel (sxfir_plan.hip.h); one tile per 512 outputs
    std::memset(g, 0, sizeof(*g));
    const bool table = p->ratio == 4 || p->ratio == 8 || p->ratio == 16 || p->ratio == 32 || p->ratio == 48 || p->ratio == 96;
    g->tiled = table && p->ntaps == 32 * p->ratio;
    std::snprintf(g->kernel, sizeof(g->kernel), "%s", g->tiled ? "fake_tiled_kernel"
                                                                 : (p->mode == SXFIR_DECIMATE ? "decim_generic_kernel" : "interp_generic_kernel"));
    g->split = 1;
    g->tile_samples = 512LL * p->ratio;
    const long long wide = p->mode == SXFIR_DECIMATE ? (long long)n_in : (long long)n_in * p->ratio;
    g->n_tiles = (wide + g->tile_samples - 1) / g->tile_samples;
    g->workgroups = g->n_tiles * p->nchan;
    g->resident = 512;
    return SXFIR_OK;
}

int sxfir_destroy(sxfir_plan *p) { drain_all(); delete p; return SXFIR_OK; }
int sxfir_reset(sxfir_plan *p, void *stream)
{
    S(stream)->push([p] { std::fill(p->hist.begin(), p->hist.end(), 0.0f); });
    return SXFIR_OK;
}
int sxfir_set_history(sxfir_plan *, const void *, size_t, size_t, void *) { g_err = "not in the fake"; return SXFIR_EUNSUPPORTED; }
int sxfir_set_tx_threshold(sxfir_plan *p, float t) { p->thr2 = t; return SXFIR_OK; }

int sxfir_synth_fill(void *out_dev, size_t n, size_t stride, int nchan, uint64_t seed, uint32_t first_channel,
                     int64_t start, int fmt, void *stream)
{
    if (fmt != SXFIR_CF32) return fail("fake backend: CF32 only");
    S(stream)->push([=] {
        for (int c = 0; c < nchan; ++c)
            sxo_synth_iq(seed, first_channel + (uint32_t)c, start, n, (float *)out_dev + 2 * (size_t)c * stride);
    });
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

static void run_all(sxfir_plan *p, const void *in_dev, size_t n_in, size_t in_stride, void *out_dev, size_t out_stride, size_t n_out)
{
    for (int c = 0; c < p->nchan; ++c)
        run_channel(p, c, (const float *)in_dev + 2 * (size_t)c * in_stride, n_in, (float *)out_dev + 2 * (size_t)c * out_stride, n_out);
}

int sxfir_decimate(sxfir_plan *p, const void *in_dev, size_t n_in, size_t in_stride, void *out_dev, size_t out_stride,
                   size_t *n_out, void *stream)
{
    if (p->mode != SXFIR_DECIMATE || n_in % (size_t)p->ratio) return fail("fake backend: whole output blocks only");
    ++g_fake_launches;
    const size_t no = n_in / (size_t)p->ratio;
    *n_out = no;
    S(stream)->push([=] { run_all(p, in_dev, n_in, in_stride, out_dev, out_stride, no); });
    return SXFIR_OK;
}

int sxfir_interpolate(sxfir_plan *p, const void *in_dev, size_t n_in, size_t in_stride, void *out_dev, size_t out_stride,
                      size_t *n_out, void *stream)
{
    if (p->mode != SXFIR_INTERPOLATE) return fail("fake backend: not an interpolator");
    ++g_fake_launches;
    const size_t no = n_in * (size_t)p->ratio;
    *n_out = no;
    S(stream)->push([=] { run_all(p, in_dev, n_in, in_stride, out_dev, out_stride, no); });
    return SXFIR_OK;
}

// the interpolation pass with the keying count of channel 0's samples [key_first, key_first + key_count)
int sxfir_interpolate_keyed(sxfir_plan *p, const void *in_dev, size_t n_in, size_t in_stride, void *out_dev, size_t out_stride,
                            size_t *n_out, size_t key_first, size_t key_count, unsigned long long *counter, void *stream)
{
    if (key_first > n_in || key_count > n_in - key_first || !counter) return fail("fake backend: bad keying range");
    if (p->mode != SXFIR_INTERPOLATE) return fail("fake backend: not an interpolator");
    ++g_fake_launches;
    const size_t no = n_in * (size_t)p->ratio;
    *n_out = no;
    const float thr2 = p->thr2;                          // as the real launch: the threshold in force when the pass is queued
    S(stream)->push([=] {
        run_all(p, in_dev, n_in, in_stride, out_dev, out_stride, no);
        count_keyed_now((const float *)in_dev + 2 * key_first, key_count, thr2, counter);
    });
    return SXFIR_OK;
}

}  // extern "C"
