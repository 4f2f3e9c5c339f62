// Measurement entry points of the C ABI: back-to-back timed passes of a plan's kernel (HIP events on the launch
// stream; bench.py's kernel_ms) and the in-kernel shader-clock probe.  Included by sxfir.hip after sxfir_launch.hip.h.
#pragma once

extern "C" {

// Timed launches (bench.py): `iters` back-to-back passes of the resampling kernel over the same buffers and
// from the same filter state, bracketed by HIP events on the launch stream.
static int time_passes(sxfir_plan *p, int mode, const void *in_dev, size_t n_in, size_t in_stride, void *out_dev,
                       size_t out_stride, int iters, void *stream, float *ms_per_pass)
{
    if (!p || !ms_per_pass || iters < 1) return fail(SXFIR_EINVAL, "bad argument");
    const long long n_out = outputs_for(p, (long long)n_in);
    int rc = check_io(p, mode, in_dev, n_in, in_stride, out_dev, out_stride, n_out);
    if (rc) return rc;
    if (n_out < 1) return fail(SXFIR_EINVAL, "nothing to do");
    HIPCHECK(hipSetDevice(p->device));
    hipEvent_t e0, e1;
    HIPCHECK(hipEventCreate(&e0));
    HIPCHECK(hipEventCreate(&e1));
    HIPCHECK(hipEventRecord(e0, S(stream)));
    for (int i = 0; i < iters; ++i) {
        bool history_done = false;   // history buffers are not swapped: every pass filters from the same state
        rc = mode == SXFIR_DECIMATE
                 ? launch_decim(p, in_dev, n_in, in_stride, out_dev, out_stride, n_out, S(stream), &history_done)
                 : launch_interp(p, in_dev, n_in, in_stride, out_dev, out_stride, n_out, S(stream), &history_done);
        if (rc) break;
    }
    hipError_t e = hipEventRecord(e1, S(stream));
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.0f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc) return rc;
    if (e != hipSuccess) return fail(SXFIR_EHIP, "event timing failed: %s", hipGetErrorString(e));
    *ms_per_pass = ms / (float)iters;
    return SXFIR_OK;
}

int sxfir_time_decimate(sxfir_plan *p, const void *in_dev, size_t n_in, size_t in_stride, void *out_dev,
                        size_t out_stride, int iters, void *stream, float *ms_per_pass)
{
    return time_passes(p, SXFIR_DECIMATE, in_dev, n_in, in_stride, out_dev, out_stride, iters, stream, ms_per_pass);
}

int sxfir_time_interpolate(sxfir_plan *p, const void *in_dev, size_t n_in, size_t in_stride, void *out_dev,
                           size_t out_stride, int iters, void *stream, float *ms_per_pass)
{
    return time_passes(p, SXFIR_INTERPOLATE, in_dev, n_in, in_stride, out_dev, out_stride, iters, stream, ms_per_pass);
}

// In-kernel shader clock while other work runs: a few single-wave workgroups on a stream of their own spin on
// s_memtime (shader cycles) against s_memrealtime (100 MHz) for `duration_us`; sxfir_clock_probe_read waits for
// them and returns the median ratio.  They use one wave slot each and no LDS, so they sit beside a running
// resampling kernel (bench.py: roofline.shader_mhz, the clock the chip's power management holds under it).
struct sxfir_clock_probe {
    hipStream_t stream;
    unsigned long long *dev;
    int n;
};

__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long *out, unsigned long long ticks)
{
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < ticks) {
        __builtin_amdgcn_s_sleep(32);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = c1 - c0;
        out[2 * blockIdx.x + 1] = r1 - r0;
    }
}

int sxfir_clock_probe_start(sxfir_clock_probe **probe, int device, int duration_us)
{
    if (!probe || duration_us < 1 || duration_us > 10000000) return fail(SXFIR_EINVAL, "bad argument");
    *probe = nullptr;
    if (device >= 0) HIPCHECK(hipSetDevice(device));
    sxfir_clock_probe *q = new (std::nothrow) sxfir_clock_probe();
    if (!q) return fail(SXFIR_ENOMEM, "out of host memory");
    q->n = 16;
    q->dev = nullptr;
    q->stream = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&q->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void **)&q->dev, 16 * q->n);
    if (e == hipSuccess) e = hipMemsetAsync(q->dev, 0, 16 * q->n, q->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(clock_probe_kernel, dim3(q->n), dim3(64), 0, q->stream, q->dev, 100ull * (unsigned long long)duration_us);
        e = hipGetLastError();
    }
    if (e != hipSuccess) {
        if (q->dev) (void)hipFree(q->dev);
        if (q->stream) (void)hipStreamDestroy(q->stream);
        delete q;
        return fail(SXFIR_EHIP, "clock probe: %s", hipGetErrorString(e));
    }
    *probe = q;
    return SXFIR_OK;
}

int sxfir_clock_probe_read(sxfir_clock_probe *q, double *mhz)
{
    if (!q || !mhz) return fail(SXFIR_EINVAL, "NULL argument");
    std::vector<unsigned long long> h(2 * (size_t)q->n);
    hipError_t e = hipStreamSynchronize(q->stream);
    if (e == hipSuccess) e = hipMemcpy(h.data(), q->dev, 16 * q->n, hipMemcpyDeviceToHost);
    (void)hipFree(q->dev);
    (void)hipStreamDestroy(q->stream);
    const int n = q->n;
    delete q;
    if (e != hipSuccess) return fail(SXFIR_EHIP, "clock probe: %s", hipGetErrorString(e));
    std::vector<double> f;
    for (int i = 0; i < n; ++i)
        if (h[2 * i + 1] > 0) f.push_back(100.0 * (double)h[2 * i] / (double)h[2 * i + 1]);
    if (f.empty()) return fail(SXFIR_EHIP, "clock probe recorded nothing");
    std::sort(f.begin(), f.end());
    *mhz = f[f.size() / 2];
    return SXFIR_OK;
}

}  // extern "C"
