// Decimate-by-32, 1024 taps (32 per phase), CF32 or S32 wire words: the dense-image form of the
// multi-column decimator (sxfir_decim_multi.hip.h) for BASELINE config 5.  New code: the reference
// decimates inside the SX1255 (SoapySX.cpp:180-208 only programs the chip's divider).
//
// Same arithmetic, same numeric contract (2, 4) as decim_multi_kernel<32, 4>: lanes = (row half p,
// column group c of four phases, output group of 8), a 64-tap fmaf chain per lane and output, then the
// adjacent-pair tree over p and over the eight column groups.  What changes is the LDS image and
// everything that follows from it:
//
//   * The tile (128 outputs + 31 halo rows = 159 rows of 32 samples = 40 704 bytes) sits in LDS as it
//     sits in HBM: linear.  A DMA instruction (global_load_lds_dwordx4) moves 1 KiB of CONSECUTIVE bytes —
//     eight whole lines — instead of 64 pieces picked from 32 different rows, so the 40 instructions of a
//     tile are cheap to issue (the de-interleaved image needed 48 of the expensive kind: DESIGN.md 5.2).
//   * One 16-byte pad slot after every 16 rows (between DMA instructions, so the pads cost no traffic):
//     40 848 bytes per workgroup, FOUR workgroups (16 waves) per CU where the de-interleaved image
//     (48 KiB) left three.
//   * Bank conflicts: chunk k = 14 - 2c + h of a row holds column group c, half h.  With the lane bits
//     (b0..b5) = (c1, c2, g1, c0, p, g0) the 16 lanes a ds_read_b128 is served with (MI355X_MICROARCH.md,
//     LDS: {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, +32) either hit 16 different slots mod 16 or the
//     same address: c supplies the even residues, the pad count (g1 - p) the odd ones, and two lanes
//     with g1 - p equal read the same window.  g0 (an offset of 8 rows, where the pads fall at other
//     steps) is bit 5, which never varies inside a group.  Exhaustive check: tools/lds_bank_model.py.
//   * Reduction without LDS: v_permlane16_swap over p (bit 4), then DPP butterflies over c0 (row_ror:8),
//     c1 and c2 (quad_perm): the contract's tree ((c0+c1)+(c2+c3))+((c4+c5)+(c6+c7)).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sxfir_decim_multi.hip.h"

namespace sxfir {

struct DecimDense32 {
    static constexpr int D = 32;
    static constexpr int NT = 1024;
    static constexpr int W = 4;                       // waves per workgroup
    static constexpr int OW = 32;                     // outputs per wave (4 groups of 8)
    static constexpr int TILE_OUT = W * OW;           // 128
    static constexpr int NROWS = TILE_OUT + 31;       // rows q in [M0 - 31, M0 + TILE_OUT)
    static constexpr int CPR = 16;                    // 16-byte chunks per row
    static constexpr int CH = NROWS * CPR;            // 2544 chunks
    static constexpr int PADROWS = 16;                // one pad slot after every 16 rows (4 DMA instructions)
    static constexpr int NI = (CH + 63) / 64;         // 40 DMA instructions per tile
    static constexpr int NIW = NI / W;                // 10 per wave
    static constexpr int LAST_LANES = CH - 64 * (NI - 1);   // 48 lanes of the last instruction are inside the image
    static constexpr int LDS_SLOTS = CH + (NROWS - 1) / PADROWS;   // 2553 slots = 40 848 bytes
    static constexpr int WCH = 46;                    // window chunks per lane: 23 rows x 2
    static_assert(NI % W == 0, "the waves share the DMA instructions evenly");
    static_assert(LDS_SLOTS * 16 <= 160 * 1024 / 4, "four workgroups per CU");
};

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

// ABL (profiling only): 1 = staging + stores without the FIR, 2 = FIR without staging, 3 = the real
// kernel with s_memtime stamps around its phases (a.stamps, 5 counters per wave as decim_multi_kernel)
template <int ABL = 0, bool S32IN = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void
decim32_dense_kernel(const DecimMultiArgs a)
{
    using C = DecimDense32;
    __shared__ __attribute__((aligned(16))) f32x4 lds[C::LDS_SLOTS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int ww = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = ((lane >> 3) & 1) | ((lane & 1) << 1) | (((lane >> 1) & 1) << 2);   // c0 = b3, c1 = b0, c2 = b1
    const int p = (lane >> 4) & 1;
    const int g1 = (lane >> 2) & 1, g0 = lane >> 5;
    const int ch = blockIdx.y;

    const char *in = reinterpret_cast<const char *>(a.in) + 8LL * a.in_stride * ch;
    const char *hist = reinterpret_cast<const char *>(a.hist) + 8LL * a.hist_stride * ch;
    char *out = reinterpret_cast<char *>(a.out) + 8LL * a.out_stride * ch;

    // lane taps: h[kl], kl = 4*jj + rr  <->  tap 32*(16*p + jj) + 4c + rr
    f32x2 hp[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const float *t = a.taps + 32 * (16 * p + (k >> 1)) + 4 * c + 2 * (k & 1);
        float t0 = t[0], t1 = t[1];
        if constexpr (S32IN) {                        // a power of two commutes with the FMA
            t0 = __fmul_rn(t0, 4.656612873077393e-10f);
            t1 = __fmul_rn(t1, 4.656612873077393e-10f);
        }
        hp[k] = (f32x2){t0, t1};
    }

    // window: rows 8u .. 8u + 22 of the image (u = output group - 2p + 2), chunks 14 - 2c + {0, 1} of each;
    // slot = chunk + (pads before its row).  The pads of a window fall after its row 16 (g0 = 0) or 8
    // (g0 = 1): base A serves steps [0, 16), B = A + g0 steps [16, 32), A + 1 steps [32, 46).
    const int u = 4 * ww + 2 * g1 + g0 - 2 * p + 2;
    const f32x4 *winA = lds + (128 * u + 14 - 2 * c + (u >> 1));
    const f32x4 *winB = winA + g0;

    const int NG = a.n_groups;
    const int first_tile = (NG % 8 == 0) ? (int)(blockIdx.x % 8) * (NG / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    // fused history carry-over (as decim_multi_kernel): the tail of (hist ++ in) becomes the next history
    if (first_tile == (a.n_tiles - 1) % NG && ww == C::W - 1) {
        char *ho = reinterpret_cast<char *>(a.hist_out) + 8LL * a.hist_stride * ch;
        for (int j = lane; j < C::NT; j += 64) {
            const long long s = a.n_in - C::NT + j;
            const char *src = s >= 0 ? in + 8 * s : hist + 8 * (s + C::NT);
            reinterpret_cast<float2 *>(ho)[j] = *reinterpret_cast<const float2 *>(src);
        }
    }

    unsigned long long ph[5] = {0, 0, 0, 0, 0}, tk = 0;
    if constexpr (ABL == 3) tk = __builtin_amdgcn_s_memtime();
#define SXFIR_PHASE(k) \
    if constexpr (ABL == 3) { \
        const unsigned long long t_now = __builtin_amdgcn_s_memtime(); \
        ph[k] += t_now - tk; \
        tk = t_now; \
    }

    // HBM -> LDS for one tile: instruction i = ww + 4*i0 moves image chunks [64i, 64i + 64) to slots
    // 64i + i/4 onwards (i/4 = i0: the pads before it)
    auto stage = [&](int tile) __attribute__((always_inline)) {
        const long long M0 = (long long)tile * C::TILE_OUT;
        const long long s_first = 32 * (M0 - 31) - 31;                   // first sample of the image
        const bool interior = s_first >= 0 && 32 * (M0 + C::TILE_OUT - 1) <= a.n_in - 1;
        const char *base = in + 8 * s_first + 1024 * ww;
        if constexpr (ABL == 2) return;
        if (interior) {
#pragma unroll
            for (int i0 = 0; i0 < C::NIW; ++i0) {
                unsigned lo = 16u * lane;
                asm volatile("" : "+v"(lo));          // a 32-bit offset next to its use: SGPR base + VGPR offset form
                // the tile's last instruction: 48 of its lanes are inside the image
                const char *bi = base + 4096 * i0;
                asm volatile("" : "+s"(bi));          // ... and the instruction's own base stays a scalar
                if (i0 < C::NIW - 1 || ww < C::W - 1 || lane < C::LAST_LANES)
                    glds16(bi + lo, lds + (64 * ww + 257 * i0));
            }
        } else {
            // edge tiles (first / last of a call): through registers, sample by sample
            const long long last = a.n_in - 1;
#pragma nounroll
            for (int i0 = 0; i0 < C::NIW; ++i0) {
                if (i0 < C::NIW - 1 || ww < C::W - 1 || lane < C::LAST_LANES) {
                    unsigned wds[4];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const long long s = s_first + 2 * (64 * (ww + 4 * i0) + lane) + e;
                        const char *src = s >= 0 ? in + 8 * (s <= last ? s : last)
                                                 : hist + 8 * (s + C::NT >= 0 ? s + C::NT : 0);
                        wds[2 * e] = reinterpret_cast<const unsigned *>(src)[0];
                        wds[2 * e + 1] = reinterpret_cast<const unsigned *>(src)[1];
                    }
                    lds[64 * ww + 257 * i0 + lane] = (f32x4){__uint_as_float(wds[0]), __uint_as_float(wds[1]),
                                                             __uint_as_float(wds[2]), __uint_as_float(wds[3])};
                }
            }
        }
    };

    if (first_tile < a.n_tiles) stage(first_tile);
    for (int tile = first_tile; tile < a.n_tiles; tile += NG) {
        const long long M0 = (long long)tile * C::TILE_OUT;
        SXFIR_PHASE(1)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own DMAs landed ...
        __syncthreads();                                    // ... and everybody else's
        SXFIR_PHASE(2)

        // ---- compute: window sample w meets output i at local tap kl = 4*i + 63 - w
        f32x2 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = (f32x2){0.0f, 0.0f};
        if constexpr (ABL != 1)
#pragma unroll
        for (int t = 0; t < C::WCH; ++t) {
            const f32x4 *wp = t < 16 ? winA : (t < 32 ? winB : winA + 1);
            const f32x4 v = wp[16 * (t >> 1) + (t & 1)];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int w = 2 * t + s;
                f32x2 x = s ? __builtin_shufflevector(v, v, 2, 3) : __builtin_shufflevector(v, v, 0, 1);
                if constexpr (S32IN) x = (f32x2){(float)__float_as_int(x.x), (float)__float_as_int(x.y)};
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int kl = 4 * i + 63 - w;
                    if (kl >= 0 && kl < 64) {
                        if (kl & 1) pk_fma_hi(acc[i], hp[kl >> 1], x);
                        else pk_fma_lo(acc[i], hp[kl >> 1], x);
                    }
                }
            }
        }
        float ai[8], aq[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { ai[i] = acc[i].x; aq[i] = acc[i].y; }

        if constexpr (ABL == 3) asm volatile("" ::"v"(ai[0]), "v"(aq[7]));
        SXFIR_PHASE(3)
        __syncthreads();                                    // everyone is done reading this tile's image
        if (tile + NG < a.n_tiles) stage(tile + NG);
        SXFIR_PHASE(4)

        // ---- reduction in the order of the numeric contract.  p (lane bit 4): even 16-lane rows keep
        // outputs 0-3, odd rows 4-7
        float ri[4], rq[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            permlane16_swap(ai[i], ai[i + 4]);
            permlane16_swap(aq[i], aq[i + 4]);
            ri[i] = __fadd_rn(ai[i], ai[i + 4]);
            rq[i] = __fadd_rn(aq[i], aq[i + 4]);
        }
        // column groups: c0 = lane bit 3 (row_ror:8 = lane ^ 8 inside a row of 16), c1 = bit 0, c2 = bit 1
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ri[i] = __fadd_rn(ri[i], dpp_f32<0x128>(ri[i]));
            rq[i] = __fadd_rn(rq[i], dpp_f32<0x128>(rq[i]));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ri[i] = __fadd_rn(ri[i], dpp_f32<0xB1>(ri[i]));     // quad_perm [1,0,3,2]
            rq[i] = __fadd_rn(rq[i], dpp_f32<0xB1>(rq[i]));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ri[i] = __fadd_rn(ri[i], dpp_f32<0x4E>(ri[i]));     // quad_perm [2,3,0,1]
            rq[i] = __fadd_rn(rq[i], dpp_f32<0x4E>(rq[i]));
        }
        // every lane of a column octet holds outputs 4p .. 4p+3 of its group; the lanes with c0 = c2 = 0
        // store two of them each (c1 picks the pair): 16 lanes, 256 consecutive bytes per wave
        const int b0 = lane & 1;
        const long long m = M0 + 32 * ww + 16 * g1 + 8 * g0 + 4 * p + 2 * b0;
        if ((lane & 0xA) == 0) {
            const float s0 = b0 ? ri[2] : ri[0], s1 = b0 ? rq[2] : rq[0];
            const float s2 = b0 ? ri[3] : ri[1], s3 = b0 ? rq[3] : rq[1];
            char *dst = out + 8 * m;
            if (m + 2 <= a.n_out) {
                __builtin_nontemporal_store((f32x4){s0, s1, s2, s3}, reinterpret_cast<f32x4 *>(dst));
            } else if (m < a.n_out) {
                reinterpret_cast<float2 *>(dst)[0] = make_float2(s0, s1);
            }
        }
        if constexpr (ABL == 3) ph[0] += 1;
    }
    if constexpr (ABL == 3) {
        if (lane == 0 && a.stamps) {
            unsigned long long *rec = a.stamps + 5 * ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * C::W + ww);
#pragma unroll
            for (int k = 0; k < 5; ++k) rec[k] = ph[k];
        }
    }
#undef SXFIR_PHASE
}

}  // namespace sxfir
