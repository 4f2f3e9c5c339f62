// Support kernels of the sx resampling path (gfx950): generic resamplers for
// shapes the LDS-tiled kernels do not cover, history carry-over, the synthetic
// IQ source and the sample-format converters.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

namespace sxfir {

// Sample access for the two IQ storage formats (arithmetic is always fp32).
struct CF32 {
    typedef float2 storage;
    static __device__ __forceinline__ float2 load(const void *base, long long idx)
    {
        return reinterpret_cast<const float2 *>(base)[idx];
    }
    static __device__ __forceinline__ void store(void *base, long long idx, float2 v, float)
    {
        reinterpret_cast<float2 *>(base)[idx] = v;
    }
};

struct CF16 {
    typedef __half2 storage;
    static __device__ __forceinline__ float2 load(const void *base, long long idx)
    {
        return __half22float2(reinterpret_cast<const __half2 *>(base)[idx]);
    }
    static __device__ __forceinline__ void store(void *base, long long idx, float2 v, float)
    {
        reinterpret_cast<__half2 *>(base)[idx] = __floats2half2_rn(v.x, v.y);
    }
};

// convert_tx_buffer, SoapySX.cpp:116-137: clamp to [-1, 1], times 2^31, to int32, clear the two low bits.  The C++ source
// leaves the conversion of 2^31 * 1.0f undefined; the rule here (and in the oracle) is the ARM behaviour of the reference's
// real platform: saturate, NaN -> 0.  v_cvt_i32_f32 does exactly that, and with a saturating conversion the clamp needs
// no instruction of its own: 2^31 * f is exact (a power of two), for |f| <= 1 it is the clamped value's product, for f > 1 it
// is > 2^31 and saturates to 0x7FFFFFFF as 2^31 * 1.0f does, for f < -1 it saturates to 0x80000000 = 2^31 * -1.0f, +-inf
// likewise, NaN stays NaN through both and converts to 0.  (The x8 wire-word interpolator spends 40 % of its FIR's energy
// on this conversion when it is spelled out with comparisons: 20 instructions per sample, profiles/round4z9_price_list.txt.)
__device__ __forceinline__ int cvt_i32_sat(float v)
{
    int r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}

__device__ __forceinline__ int tx_word(float f)
{
    return cvt_i32_sat(__fmul_rn(2147483648.0f, f)) & (int)0xFFFFFFFC;
}

__device__ __forceinline__ int2 tx_words(float fi, float fq, float thr2)
{
    // both scalings in one packed multiply, both squares in another (each product rounded once, as the rule states)
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t f = {fi, fq}, k = {2147483648.0f, 2147483648.0f};
    f32x2_t v, sq;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(v) : "v"(f), "v"(k));
    asm("v_pk_mul_f32 %0, %1, %1" : "=v"(sq) : "v"(f));
    int vi = cvt_i32_sat(v.x) & (int)0xFFFFFFFC;
    const int vq = cvt_i32_sat(v.y) & (int)0xFFFFFFFC;
    if (__fadd_rn(sq.x, sq.y) >= thr2) vi |= 3;
    return make_int2(vi, vq);
}

// S32_LE I2S wire words: load = convert_rx_buffer (SoapySX.cpp:103-112), store = convert_tx_buffer.
struct S32 {
    typedef int2 storage;
    static __device__ __forceinline__ float2 load(const void *base, long long idx)
    {
        const int2 v = reinterpret_cast<const int2 *>(base)[idx];
        return make_float2(__fmul_rn(4.656612873077393e-10f, (float)v.x), __fmul_rn(4.656612873077393e-10f, (float)v.y));
    }
    static __device__ __forceinline__ void store(void *base, long long idx, float2 v, float thr2)
    {
        reinterpret_cast<int2 *>(base)[idx] = tx_words(v.x, v.y, thr2);
    }
};

struct GenericArgs {
    const void *in;
    const void *hist;       // hist_len samples preceding in[0] (per channel)
    void *out;
    const float *taps;
    long long n_in, n_out;
    long long in_stride, out_stride, hist_stride;
    long long first;        // decim: sample index (relative to in[0]) of output 0's newest sample
                            // interp: phase offset (always 0 here)
    int ntaps, ratio, hist_len;
    int jsplit, cw;         // numeric contract
    int rot;                // decimator: slot k' = j*D + r holds tap (k' + rot) mod ntaps (sxfir_contract_rotation)
    float thr2;             // S32 output only: transmitter-keying threshold (squared magnitude)
};

template <typename F>
__device__ __forceinline__ float2 sample_at(const GenericArgs &a, const void *in, const void *hist, long long idx)
{
    if (idx >= 0) return F::load(in, idx);
    if (idx + a.hist_len >= 0) return F::load(hist, idx + a.hist_len);
    return make_float2(0.0f, 0.0f);
}

// One output per thread, any ntaps / ratio / alignment.  Same contract as the
// tiled kernels: partial[c][p] over rows j descending and phases r descending,
// adjacent-pair trees over p then c.  (jsplit, cw <= 32 partials each.)
template <typename F, typename FO = F>
__global__ __launch_bounds__(256) void decim_generic_kernel(const GenericArgs a)
{
    const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= a.n_out) return;
    const int ch = blockIdx.y;
    const char *in = (const char *)a.in + sizeof(typename F::storage) * a.in_stride * ch;
    const char *hist = (const char *)a.hist + sizeof(typename F::storage) * a.hist_stride * ch;
    char *out = (char *)a.out + sizeof(typename FO::storage) * a.out_stride * ch;

    const int D = a.ratio;
    const int jt = (a.ntaps + D - 1) / D;
    const int jl = jt / a.jsplit;
    const int ncol = D / a.cw;
    const long long newest = a.first + m * D;

    float ci[32], cq[32];
    for (int c = 0; c < ncol; ++c) {
        float pi[32], pq[32];
        for (int p = 0; p < a.jsplit; ++p) {
            float si = 0.0f, sq = 0.0f;
            for (int j = (p + 1) * jl - 1; j >= p * jl; --j) {
                for (int r = (c + 1) * a.cw - 1; r >= c * a.cw; --r) {
                    int k = j * D + r;
                    if (k >= a.ntaps) continue;
                    k += a.rot;
                    if (k >= a.ntaps) k -= a.ntaps;
                    const float2 x = sample_at<F>(a, in, hist, newest - k);
                    const float t = a.taps[k];
                    si = __builtin_fmaf(t, x.x, si);
                    sq = __builtin_fmaf(t, x.y, sq);
                }
            }
            pi[p] = si;
            pq[p] = sq;
        }
        for (int n = a.jsplit; n > 1; n >>= 1)
            for (int i = 0; i < n / 2; ++i) {
                pi[i] = __fadd_rn(pi[2 * i], pi[2 * i + 1]);
                pq[i] = __fadd_rn(pq[2 * i], pq[2 * i + 1]);
            }
        ci[c] = pi[0];
        cq[c] = pq[0];
    }
    // adjacent pairs, level by level; an odd element at the end of a level moves up unchanged (12 columns: 6, 3, 2, 1)
    for (int n = ncol; n > 1; n = n / 2 + (n & 1)) {
        for (int i = 0; i < n / 2; ++i) {
            ci[i] = __fadd_rn(ci[2 * i], ci[2 * i + 1]);
            cq[i] = __fadd_rn(cq[2 * i], cq[2 * i + 1]);
        }
        if (n & 1) {
            ci[n / 2] = ci[n - 1];
            cq[n / 2] = cq[n - 1];
        }
    }
    FO::store(out, m, make_float2(ci[0], cq[0]), a.thr2);
}

// Interpolator, one output per thread: y[n] = sum_j h[j*L + n%L] x[n/L - j],
// jsplit contiguous ranges of j, chain over descending j, adjacent-pair tree.
template <typename F, typename FO = F>
__global__ __launch_bounds__(256) void interp_generic_kernel(const GenericArgs a)
{
    const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= a.n_out) return;
    const int ch = blockIdx.y;
    const char *in = (const char *)a.in + sizeof(typename F::storage) * a.in_stride * ch;
    const char *hist = (const char *)a.hist + sizeof(typename F::storage) * a.hist_stride * ch;
    char *out = (char *)a.out + sizeof(typename FO::storage) * a.out_stride * ch;

    const int L = a.ratio;
    const int jt = a.ntaps / L;
    const int jl = jt / a.jsplit;
    const long long q = n / L;
    const int r = (int)(n - q * L);
    float pi[32], pq[32];
    for (int p = 0; p < a.jsplit; ++p) {
        float si = 0.0f, sq = 0.0f;
        for (int j = (p + 1) * jl - 1; j >= p * jl; --j) {
            const float2 x = sample_at<F>(a, in, hist, q - j);
            const float t = a.taps[j * L + r];
            si = __builtin_fmaf(t, x.x, si);
            sq = __builtin_fmaf(t, x.y, sq);
        }
        pi[p] = si;
        pq[p] = sq;
    }
    for (int m = a.jsplit; m > 1; m >>= 1)
        for (int i = 0; i < m / 2; ++i) {
            pi[i] = __fadd_rn(pi[2 * i], pi[2 * i + 1]);
            pq[i] = __fadd_rn(pq[2 * i], pq[2 * i + 1]);
        }
    FO::store(out, n, make_float2(pi[0], pq[0]), a.thr2);
}

// History carry-over for the generic path: hist_out <- last hist_len samples of (hist ++ in[0, n_in)).
// Out of place (the plan keeps two history buffers and swaps them, as the tiled kernels do), so it is
// one element per thread over a grid of any size: no limit on hist_len.
template <typename S>
__global__ __launch_bounds__(256) void history_kernel(S *hist_out, const S *hist, const S *in, long long n_in,
                                                      long long in_stride, long long hist_stride, int hist_len)
{
    const int ch = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= hist_len) return;
    const long long s = n_in - hist_len + j;
    hist_out[hist_stride * ch + j] = s >= 0 ? in[in_stride * ch + s] : hist[hist_stride * ch + s + hist_len];
}

// ---- synthetic IQ source: splitmix64 stream keyed by (seed, channel) --------
__device__ __forceinline__ uint64_t sm64_finalize(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

__device__ __forceinline__ float2 synth_sample(uint64_t key, long long idx)
{
    if (idx < 0) return make_float2(0.0f, 0.0f);
    const uint64_t u = sm64_finalize(key + 0x9E3779B97F4A7C15ULL * ((uint64_t)idx + 1));
    const int a = (int)(u >> 40) - 8388608;
    const int b = (int)((u >> 16) & 0xFFFFFF) - 8388608;
    return make_float2((float)a * (1.0f / 8388608.0f), (float)b * (1.0f / 8388608.0f));
}

template <typename F>
__global__ __launch_bounds__(256) void synth_kernel(void *out, long long n, long long stride, uint64_t seed,
                                                    uint32_t first_channel, long long start)
{
    const int ch = blockIdx.y;
    const uint64_t key = sm64_finalize(seed + 0x9E3779B97F4A7C15ULL * ((uint64_t)(first_channel + ch) + 1));
    char *o = (char *)out + sizeof(typename F::storage) * stride * ch;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        F::store(o, i, synth_sample(key, start + i), 0.0f);
}

// The same source as S32_LE wire words: the 24-bit grid value v has the exact word v * 2^31
// (24-bit integer << 8), so convert_rx of these words reproduces the CF32 source bit for bit.
__global__ __launch_bounds__(256) void synth_s32_kernel(int2 *out, long long n, long long stride, uint64_t seed,
                                                        uint32_t first_channel, long long start)
{
    const int ch = blockIdx.y;
    const uint64_t key = sm64_finalize(seed + 0x9E3779B97F4A7C15ULL * ((uint64_t)(first_channel + ch) + 1));
    int2 *o = out + stride * ch;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const float2 v = synth_sample(key, start + i);
        o[i] = make_int2((int)(v.x * 8388608.0f) * 256, (int)(v.y * 8388608.0f) * 256);
    }
}

// ---- S32_LE I2S wire format (SoapySX.cpp:103-137) --------------------------
// convert_rx_buffer, SoapySX.cpp:103-112: dst = 2^-31 * (float)src
__global__ __launch_bounds__(256) void convert_rx_kernel(const int2 *src, float2 *dst, long long n)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const int2 v = src[i];
        dst[i] = make_float2(__fmul_rn(4.656612873077393e-10f, (float)v.x),
                             __fmul_rn(4.656612873077393e-10f, (float)v.y));
    }
}

__global__ __launch_bounds__(256) void convert_tx_kernel(const float2 *src, int2 *dst, long long n, float thr2)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const float2 f = src[i];
        dst[i] = tx_words(f.x, f.y, thr2);
    }
}

// Transmitter keying count (convert_tx_buffer, SoapySX.cpp:132-133): how many of n complex samples reach the
// squared-magnitude threshold.  One atomic per workgroup into a counter in device memory.
__global__ __launch_bounds__(256) void count_keyed_kernel(const float2 *src, long long n, float thr2, unsigned long long *counter)
{
    unsigned cnt = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float2 f = src[i];
        const float ii = __fmul_rn(f.x, f.x), qq = __fmul_rn(f.y, f.y);
        cnt += (__fadd_rn(ii, qq) >= thr2) ? 1u : 0u;
    }
    __shared__ unsigned block_total;
    if (threadIdx.x == 0) block_total = 0;
    __syncthreads();
    // wave total by ballot-free shuffle reduction, then one LDS add per wave
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
    if ((threadIdx.x & 63) == 0) atomicAdd(&block_total, cnt);
    __syncthreads();
    if (threadIdx.x == 0 && block_total) atomicAdd(counter, (unsigned long long)block_total);
}

__global__ __launch_bounds__(256) void cf32_to_cf16_kernel(const float2 *src, __half2 *dst, long long n)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        dst[i] = __floats2half2_rn(src[i].x, src[i].y);
}

__global__ __launch_bounds__(256) void cf16_to_cf32_kernel(const __half2 *src, float2 *dst, long long n)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        dst[i] = __half22float2(src[i]);
}

}  // namespace sxfir
