// Variant of the decimate-by-4 tile kernel that keeps the taps in SGPRs
// (experiment; selected with SXFIR_TILE_VARIANT=sg).  Same numeric contract
// and LDS image as sxfir_decim_tile.hip.h, different work split:
//   * every lane computes R consecutive outputs over ALL taps, as two chains
//     (taps [0,64) then [64,128)) whose partials are added at the end, so no
//     cross-lane reduction is needed;
//   * a chain's 64 taps are wave-uniform, so they are scalar loads and each
//     v_fmac reads one SGPR + two VGPRs instead of three VGPRs.
// A wave's tile is 64*R outputs.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../sxfir_decim_tile.hip.h"

namespace sxfir {

template <int R>
struct DecimSgpr4 {
    static constexpr int NT = 128;
    static constexpr int TILE_OUT = 64 * R;
    static constexpr int TILE_IN = 4 * TILE_OUT;
    static constexpr int HALO = NT;
    static constexpr int PADP = 2 * R;                    // lane stride in chunks; one pad chunk after every PADP
    static constexpr int CHUNKS = (TILE_IN + HALO) / 2;
    static constexpr int SLOTS = CHUNKS + CHUNKS / PADP;
    static constexpr int NLOAD = (SLOTS + 63) / 64;
    static constexpr int WMAX = 4 * (R - 1) + 64;
    static constexpr int WCH = WMAX / 2 + 1;
    static_assert(R == 4 || R == 8, "R is 4 or 8");
};

template <int R>
__global__ __launch_bounds__(64) void decim4_sgpr_kernel(const DecimTileArgs a)
{
    using C = DecimSgpr4<R>;
    __shared__ __attribute__((aligned(16))) f32x4 lds[C::NLOAD * 64];

    const int lane = threadIdx.x;
    const int ch = blockIdx.y;
    const float *in = a.in + 2 * a.in_stride * ch;
    const float *hist = a.hist + 2 * a.hist_stride * ch;
    float *out = a.out + 2 * a.out_stride * ch;
    const long long last_chunk = (a.n_in - 1) >> 1;

    // slot -> logical chunk of every DMA this lane issues (tile invariant)
    int coff[C::NLOAD];
#pragma unroll
    for (int i = 0; i < C::NLOAD; ++i) {
        int q = 64 * i + lane;
        if (q % (C::PADP + 1) == C::PADP) q -= 1;             // pad slot: dup the left neighbour
        int c = q - q / (C::PADP + 1);
        coff[i] = c < C::CHUNKS ? c : C::CHUNKS - 1;
    }

    const int W = a.n_waves;
    const int wave = blockIdx.x;
    int tile_begin = (W % 8 == 0) ? (wave % 8) * (W / 8) + wave / 8 : wave;
    if (tile_begin >= a.n_tiles) return;

    if (tile_begin <= a.n_tiles - 1 && (a.n_tiles - 1 - tile_begin) % W == 0) {
        float *ho = a.hist_out + 2 * a.hist_stride * ch;
        for (int j = lane; j < C::NT; j += 64) {
            const long long s = a.n_in - C::NT + j;
            const float2 v = s >= 0 ? reinterpret_cast<const float2 *>(in)[s]
                                    : reinterpret_cast<const float2 *>(hist)[s + C::NT];
            reinterpret_cast<float2 *>(ho)[j] = v;
        }
    }

    for (int tile = tile_begin; tile < a.n_tiles; tile += W) {
        const long long c0 = ((long long)tile * C::TILE_IN - C::HALO) >> 1;
        const bool interior = (c0 >= 0) && (c0 + C::CHUNKS - 1 <= last_chunk);
#pragma unroll
        for (int i = 0; i < C::NLOAD; ++i) {
            long long c = c0 + coff[i];
            const f32x4 *src;
            if (interior) {
                src = reinterpret_cast<const f32x4 *>(in) + c;
            } else if (c < 0) {
                src = reinterpret_cast<const f32x4 *>(hist) + (c + C::NT / 2);
            } else {
                if (c > last_chunk) c = last_chunk;
                src = reinterpret_cast<const f32x4 *>(in) + c;
            }
            glds16(src, lds + 64 * i);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

        float si[R], sq[R];
#pragma unroll 1
        for (int p = 0; p < 2; ++p) {
            // chain p: taps [64p, 64p+64), wave-uniform -> scalar loads
            // constant address space + wave-uniform index -> s_load into SGPRs
            const __attribute__((address_space(4))) float *tp =
                (const __attribute__((address_space(4))) float *)(a.taps) + 64 * p;
            float h[64];
#pragma unroll
            for (int k = 0; k < 64; ++k) h[k] = tp[k];
            const int u0c = C::PADP * lane - 32 * p + 32;             // multiple of PADP
            const f32x4 *win = lds + (u0c + u0c / C::PADP);
            float ai[R], aq[R];
#pragma unroll
            for (int i = 0; i < R; ++i) { ai[i] = 0.0f; aq[i] = 0.0f; }
#pragma unroll
            for (int t = 0; t < C::WCH; ++t) {
                const f32x4 v = win[t + t / C::PADP];
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int w = 2 * t + s;
                    const float xi = s ? v.z : v.x;
                    const float xq = s ? v.w : v.y;
#pragma unroll
                    for (int i = 0; i < R; ++i) {
                        const int kl = 4 * i + 64 - w;
                        if (kl >= 0 && kl < 64) {
                            // one SGPR (tap) + two VGPR operands; plain asm so the compiler neither packs
                            // I/Q into v_pk_fma_f32 nor copies the tap to a VGPR
                            asm("v_fmac_f32_e32 %0, %1, %2" : "+v"(ai[i]) : "s"(h[kl]), "v"(xi));
                            asm("v_fmac_f32_e32 %0, %1, %2" : "+v"(aq[i]) : "s"(h[kl]), "v"(xq));
                        }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < R; ++i) {
                si[i] = p == 0 ? ai[i] : __fadd_rn(si[i], ai[i]);
                sq[i] = p == 0 ? aq[i] : __fadd_rn(sq[i], aq[i]);
            }
        }

        // whole-line stores through LDS (the image is dead now)
        const long long m0 = (long long)tile * C::TILE_OUT;
        if (m0 + C::TILE_OUT <= a.n_out) {
#pragma unroll
            for (int j = 0; j < R / 2; ++j)
                lds[(R / 2) * lane + j] = (f32x4){si[2 * j], sq[2 * j], si[2 * j + 1], sq[2 * j + 1]};
            f32x4 *dst = reinterpret_cast<f32x4 *>(out + 2 * m0);
#pragma unroll
            for (int k = 0; k < R / 2; ++k) {
                const f32x4 v = lds[64 * k + lane];
                __builtin_nontemporal_store(v, dst + 64 * k + lane);
            }
        } else {
            const long long m = m0 + (long long)R * lane;
#pragma unroll
            for (int i = 0; i < R; ++i)
                if (m + i < a.n_out) { out[2 * (m + i)] = si[i]; out[2 * (m + i) + 1] = sq[i]; }
        }
    }
}

}  // namespace sxfir
