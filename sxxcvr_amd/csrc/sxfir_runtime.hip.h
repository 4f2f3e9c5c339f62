// The rest of the C ABI a host needs around the kernels: the synthetic IQ source, the S32 wire / CF16 converters,
// SoapySDR's tick <-> ns arithmetic, the tap design, and thin device-memory / stream / event helpers (so that a C or
// C++ host never includes HIP headers).  Included by sxfir.hip.
#pragma once

extern "C" {

int sxfir_synth_fill(void *out_dev, size_t n, size_t stride, int nchan, uint64_t seed, uint32_t first_channel,
                     int64_t start, int fmt, void *stream)
{
    if (!out_dev && n) return fail(SXFIR_EINVAL, "NULL buffer");
    if (nchan < 1) return fail(SXFIR_EINVAL, "nchan < 1");
    if (n == 0) return SXFIR_OK;
    unsigned bx = (unsigned)((n + 255) / 256);
    if (bx > 16384) bx = 16384;
    dim3 grid(bx, (unsigned)nchan);
    if (fmt == SXFIR_CF32)
        hipLaunchKernelGGL(sxfir::synth_kernel<sxfir::CF32>, grid, dim3(256), 0, S(stream), out_dev, (long long)n,
                           (long long)stride, seed, first_channel, (long long)start);
    else if (fmt == SXFIR_CF16)
        hipLaunchKernelGGL(sxfir::synth_kernel<sxfir::CF16>, grid, dim3(256), 0, S(stream), out_dev, (long long)n,
                           (long long)stride, seed, first_channel, (long long)start);
    else if (fmt == SXFIR_S32)
        hipLaunchKernelGGL(sxfir::synth_s32_kernel, grid, dim3(256), 0, S(stream), (int2 *)out_dev, (long long)n,
                           (long long)stride, seed, first_channel, (long long)start);
    else
        return fail(SXFIR_EINVAL, "bad format");
    HIPCHECK(hipGetLastError());
    return SXFIR_OK;
}

static unsigned stream_grid(size_t n)
{
    size_t b = (n + 255) / 256;
    return (unsigned)(b > 8192 ? 8192 : (b ? b : 1));
}

int sxfir_convert_rx_s32(const int32_t *src, float *dst, size_t n, void *stream)
{
    if (n == 0) return SXFIR_OK;
    if (!src || !dst) return fail(SXFIR_EINVAL, "NULL buffer");
    hipLaunchKernelGGL(sxfir::convert_rx_kernel, dim3(stream_grid(n)), dim3(256), 0, S(stream), (const int2 *)src,
                       (float2 *)dst, (long long)n);
    HIPCHECK(hipGetLastError());
    return SXFIR_OK;
}

int sxfir_convert_tx_s32(const float *src, int32_t *dst, size_t n, float thr2, void *stream)
{
    if (n == 0) return SXFIR_OK;
    if (!src || !dst) return fail(SXFIR_EINVAL, "NULL buffer");
    hipLaunchKernelGGL(sxfir::convert_tx_kernel, dim3(stream_grid(n)), dim3(256), 0, S(stream),
                       (const float2 *)src, (int2 *)dst, (long long)n, thr2);
    HIPCHECK(hipGetLastError());
    return SXFIR_OK;
}

int sxfir_count_keyed(const float *src, size_t n, float thr2, unsigned long long *counter, void *stream)
{
    if (n == 0) return SXFIR_OK;
    if (!src || !counter) return fail(SXFIR_EINVAL, "NULL buffer");
    hipLaunchKernelGGL(sxfir::count_keyed_kernel, dim3(stream_grid(n) > 256 ? 256 : stream_grid(n)), dim3(256), 0, S(stream),
                       (const float2 *)src, (long long)n, thr2, counter);
    HIPCHECK(hipGetLastError());
    return SXFIR_OK;
}

int sxfir_cf32_to_cf16(const float *src, void *dst, size_t n, void *stream)
{
    if (n == 0) return SXFIR_OK;
    if (!src || !dst) return fail(SXFIR_EINVAL, "NULL buffer");
    hipLaunchKernelGGL(sxfir::cf32_to_cf16_kernel, dim3(stream_grid(n)), dim3(256), 0, S(stream),
                       (const float2 *)src, (__half2 *)dst, (long long)n);
    HIPCHECK(hipGetLastError());
    return SXFIR_OK;
}

int sxfir_cf16_to_cf32(const void *src, float *dst, size_t n, void *stream)
{
    if (n == 0) return SXFIR_OK;
    if (!src || !dst) return fail(SXFIR_EINVAL, "NULL buffer");
    hipLaunchKernelGGL(sxfir::cf16_to_cf32_kernel, dim3(stream_grid(n)), dim3(256), 0, S(stream),
                       (const __half2 *)src, (float2 *)dst, (long long)n);
    HIPCHECK(hipGetLastError());
    return SXFIR_OK;
}

// SoapySDR::ticksToTimeNs / timeNsToTicks as used by SoapySX.cpp:562-571
// (SoapySDR lib/TimeC.cpp): whole seconds in integers, remainder in double.
long long sxfir_ticks_to_time_ns(long long ticks, double rate)
{
    const long long ratell = (long long)rate;
    const long long full = ticks / ratell;
    const long long err = ticks - full * ratell;
    const double part = (double)full * (rate - (double)ratell);
    const double frac = (((double)err - part) * 1e9) / rate;
    return full * 1000000000LL + std::llround(frac);
}

long long sxfir_time_ns_to_ticks(long long time_ns, double rate)
{
    const long long ratell = (long long)rate;
    const long long full = time_ns / 1000000000LL;
    const long long err = time_ns - full * 1000000000LL;
    const double part = (double)full * (rate - (double)ratell);
    const double frac = part + ((double)err * rate) / 1e9;
    return full * ratell + std::llround(frac);
}

static double i0(double x)
{
    double sum = 1.0, term = 1.0;
    const double q = x * x / 4.0;
    for (int k = 1; k < 500; ++k) {
        term *= q / ((double)k * k);
        sum += term;
        if (term < sum * 1e-18) break;
    }
    return sum;
}

int sxfir_design_lowpass(int ntaps, int ratio, double beta, double gain, float *taps)
{
    if (ntaps < 1 || ratio < 1 || !taps) return fail(SXFIR_EINVAL, "bad argument");
    std::vector<double> h((size_t)ntaps);
    const double pi = 3.14159265358979323846;
    const double centre = (ntaps - 1) / 2.0;
    const double den = i0(beta);
    double total = 0.0;
    for (int k = 0; k < ntaps; ++k) {
        const double t = k - centre;
        const double arg = t / ratio;                       // 2*fc*t with fc = 0.5/ratio
        const double sinc = (t == 0.0) ? 1.0 : std::sin(pi * arg) / (pi * arg);
        double u = (centre > 0.0) ? t / centre : 0.0;
        u = 1.0 - u * u;
        const double win = i0(beta * std::sqrt(u > 0.0 ? u : 0.0)) / den;
        h[(size_t)k] = sinc * win / ratio;
        total += h[(size_t)k];
    }
    for (int k = 0; k < ntaps; ++k) taps[k] = (float)(h[(size_t)k] * (gain / total));
    return SXFIR_OK;
}

int sxfir_malloc(void **dev, size_t bytes)
{
    if (!dev) return fail(SXFIR_EINVAL, "NULL argument");
    *dev = nullptr;
    hipError_t e = hipMalloc(dev, bytes ? bytes : 1);
    if (e == hipErrorOutOfMemory) return fail(SXFIR_ENOMEM, "hipMalloc(%zu) out of memory", bytes);
    if (e != hipSuccess) return fail(SXFIR_EHIP, "hipMalloc: %s", hipGetErrorString(e));
    return SXFIR_OK;
}

int sxfir_free(void *dev)
{
    if (dev) HIPCHECK(hipFree(dev));
    return SXFIR_OK;
}

int sxfir_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream)
{
    HIPCHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, S(stream)));
    return SXFIR_OK;
}

int sxfir_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream)
{
    HIPCHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, S(stream)));
    return SXFIR_OK;
}

int sxfir_stream_sync(void *stream)
{
    HIPCHECK(hipStreamSynchronize(S(stream)));
    return SXFIR_OK;
}

int sxfir_set_device(int device)
{
    HIPCHECK(hipSetDevice(device));
    return SXFIR_OK;
}

int sxfir_host_alloc(void **host, size_t bytes)
{
    if (!host) return fail(SXFIR_EINVAL, "NULL argument");
    *host = nullptr;
    hipError_t e = hipHostMalloc(host, bytes ? bytes : 1, hipHostMallocDefault);
    if (e == hipErrorOutOfMemory) return fail(SXFIR_ENOMEM, "hipHostMalloc(%zu) out of memory", bytes);
    if (e != hipSuccess) return fail(SXFIR_EHIP, "hipHostMalloc: %s", hipGetErrorString(e));
    return SXFIR_OK;
}

int sxfir_host_register(void *host, size_t bytes)
{
    if (!host || !bytes) return fail(SXFIR_EINVAL, "NULL argument");
    hipError_t e = hipHostRegister(host, bytes, hipHostRegisterDefault);
    if (e != hipSuccess) return fail(SXFIR_EHIP, "hipHostRegister: %s", hipGetErrorString(e));
    return SXFIR_OK;
}

int sxfir_host_unregister(void *host)
{
    if (host) HIPCHECK(hipHostUnregister(host));
    return SXFIR_OK;
}

int sxfir_host_device_pointer(const void *host, size_t bytes, void **dev)
{
    if (!host || !dev) return fail(SXFIR_EINVAL, "NULL argument");
    *dev = nullptr;
    hipPointerAttribute_t at;
    hipError_t e = hipPointerGetAttributes(&at, host);
    if (e != hipSuccess) {
        (void)hipGetLastError();                        // plain pageable memory: not an error, just not visible
        return SXFIR_EUNSUPPORTED;
    }
    if (at.type != hipMemoryTypeHost) return SXFIR_EUNSUPPORTED;
    // the last byte must belong to the same page-locked range
    hipPointerAttribute_t end;
    if (bytes > 1 && (hipPointerGetAttributes(&end, (const char *)host + bytes - 1) != hipSuccess || end.type != hipMemoryTypeHost)) {
        (void)hipGetLastError();
        return SXFIR_EUNSUPPORTED;
    }
    void *d = nullptr;
    e = hipHostGetDevicePointer(&d, const_cast<void *>(host), 0);
    if (e != hipSuccess || !d) {
        (void)hipGetLastError();
        return SXFIR_EUNSUPPORTED;
    }
    // ... and to the same registration: inside one page-locked allocation the device view is linear, so the last
    // byte's device pointer is d + bytes - 1; a range that spans two registrations (or a pageable hole between
    // them) maps elsewhere and is refused -- kernels and DMA copies write through d across the whole range
    if (bytes > 1) {
        void *dl = nullptr;
        e = hipHostGetDevicePointer(&dl, const_cast<char *>((const char *)host + bytes - 1), 0);
        if (e != hipSuccess || dl != (char *)d + bytes - 1) {
            (void)hipGetLastError();
            return SXFIR_EUNSUPPORTED;
        }
        // and, where the runtime reports the allocation the device pointer belongs to, the range ends inside it
        void *base = nullptr;
        size_t size = 0;
        if (hipMemGetAddressRange((hipDeviceptr_t *)&base, &size, (hipDeviceptr_t)d) == hipSuccess && base && size) {
            if ((char *)d + bytes > (char *)base + size) return SXFIR_EUNSUPPORTED;
        } else {
            (void)hipGetLastError();
        }
    }
    *dev = d;
    return SXFIR_OK;
}

int sxfir_host_free(void *host)
{
    if (host) HIPCHECK(hipHostFree(host));
    return SXFIR_OK;
}

int sxfir_stream_create(void **stream)
{
    if (!stream) return fail(SXFIR_EINVAL, "NULL argument");
    hipStream_t st = nullptr;
    HIPCHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    *stream = (void *)st;
    return SXFIR_OK;
}

int sxfir_stream_destroy(void *stream)
{
    if (stream) HIPCHECK(hipStreamDestroy(S(stream)));
    return SXFIR_OK;
}

int sxfir_event_create(void **event)
{
    if (!event) return fail(SXFIR_EINVAL, "NULL argument");
    hipEvent_t e = nullptr;
    HIPCHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *event = (void *)e;
    return SXFIR_OK;
}

int sxfir_event_create_timing(void **event)
{
    if (!event) return fail(SXFIR_EINVAL, "NULL argument");
    hipEvent_t e = nullptr;
    HIPCHECK(hipEventCreate(&e));
    *event = (void *)e;
    return SXFIR_OK;
}

int sxfir_event_elapsed_ms(void *start, void *stop, float *ms)
{
    if (!start || !stop || !ms) return fail(SXFIR_EINVAL, "NULL argument");
    HIPCHECK(hipEventSynchronize((hipEvent_t)stop));
    HIPCHECK(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return SXFIR_OK;
}

int sxfir_event_destroy(void *event)
{
    if (event) HIPCHECK(hipEventDestroy((hipEvent_t)event));
    return SXFIR_OK;
}

int sxfir_event_record(void *event, void *stream)
{
    if (!event) return fail(SXFIR_EINVAL, "NULL event");
    HIPCHECK(hipEventRecord((hipEvent_t)event, S(stream)));
    return SXFIR_OK;
}

int sxfir_event_sync(void *event)
{
    if (!event) return fail(SXFIR_EINVAL, "NULL event");
    HIPCHECK(hipEventSynchronize((hipEvent_t)event));
    return SXFIR_OK;
}

int sxfir_stream_wait_event(void *stream, void *event)
{
    if (!event) return fail(SXFIR_EINVAL, "NULL event");
    HIPCHECK(hipStreamWaitEvent(S(stream), (hipEvent_t)event, 0));
    return SXFIR_OK;
}

}  // extern "C"
