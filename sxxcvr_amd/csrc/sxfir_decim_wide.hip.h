// Decimate-by-4, 128 symmetric taps: 8 outputs per lane, one wave per tile of 512 outputs (gfx950).
// The production kernel of BASELINE config 2 since round 4 (a measured variant in rounds 2-3).
//
// The /4 kernels are bound by the energy of their arithmetic at the board's power cap (LABBOOK.md 5.1), and the
// arithmetic probe ranks the instruction mixes: taps as SGPR operands, and as few LDS reads per FMA as possible.
// sxfir_decim_tile2.hip.h (T2_SCALAR) has the SGPR taps with 4 outputs per lane: 71 ds_read_b128 per 512 packed
// FMAs.  This kernel doubles the outputs per lane.  In round 2 it tied with the 4-output form: two waves per SIMD left
// its structure no faster than the power cap allowed the other.  Round 4's non-temporal staging loads (LABBOOK.md 5.1
// "Round 4") and a deeper LDS read-ahead (NB = 24 chunks; two waves per SIMD leave a wave 256 VGPRs) moved its
// structure well below the cap (0.44 ms on an all-zero input), and at the cap its lower energy per sample -- 39.5 instead
// of 71 LDS reads per 512 FMAs, a 6 % instead of a 12.5 % halo, sixteen consecutive FMAs sharing a sample pair -- now
// shows: 1.2 ... 2.4 % less time than the 4-output form on the same box (profiles/round4j_kbench_wide_nt.txt,
// round4k_kbench_wide_read_ahead.txt).
//
//   * lane l -> outputs 8l..8l+7 of the tile over ALL 128 taps: 16 accumulators (P1 = taps 127..64 and P0 = taps
//     63..0 of each output, the chains of the numeric contract, y = P0 + P1), one window of 79 chunks read ONCE:
//     chunk c feeds the P1 chains while c < 47 and the P0 chains from c = 32 on -- 79 ds_read_b128 for 1024
//     v_pk_fma_f32, 39.5 per 512;
//   * the 64 distinct taps of a symmetric filter sit in 32 SGPR pairs (h[64 + k] = h[63 - k]);
//   * tile = 512 outputs = 2048 inputs + 128-sample halo (6 % instead of 12.5 %): 1156 slots = 18 496 B of LDS per
//     wave, so 8 waves per CU (2 per SIMD) instead of 16 -- the price; the 16 independent accumulator chains keep
//     a SIMD's FMA pipe busy from two waves;
//   * one wave = one workgroup, no barriers; staging (19 LDS-DMA instructions), whole-line stores through the
//     dead image, fused history carry-over and schedule as in the other /4 kernels.
//
// New code: the reference decimates inside the SX1255 (SoapySX.cpp:180-208 only programs the divider); this
// kernel plays that role for SoapySX::readStream (SoapySX.cpp:868-967).
#pragma once

#include <utility>

#include "sxfir_decim_tile.hip.h"
#include "sxfir_common.hip.h"            // pk_fma_s_* / pk_fma_sv_*, slot_source_offset
#include "sxfir_decim_dense.hip.h"      // v4i32, half_lo_to_float / half_hi_to_float, pack_half2 (the CF16 front end)

namespace sxfir {

struct DecimWide {
    static constexpr int NT = 128, D = 4, R = 8;
    static constexpr int TILE_OUT = 64 * R;               // 512
    static constexpr int TILE_IN = TILE_OUT * D;          // 2048
    static constexpr int HALO = NT, HIST = NT;
    static constexpr int CHUNKS = (TILE_IN + HALO) / 2;   // 1088
    static constexpr int SLOTS = CHUNKS + CHUNKS / 16;    // 1156: one pad slot after every 16 chunks
    static constexpr int NI = (SLOTS + 63) / 64;          // 19 DMA instructions, the last one for LASTL lanes
    static constexpr int LASTL = SLOTS - 64 * (NI - 1);   // 4
    static constexpr int WCH = 79;                        // window chunks per lane
    static constexpr int P1CH = 47;                       // chunks [0, 47) feed P1, chunks [32, 79) feed P0
    static constexpr int P0FROM = 32;
};

// One window chunk c of lane l (window base: chunk 16l).  Sample w1 = 2c + s meets output i of the P1 chain at
// tap 64 + kl, kl = 4i + 64 - w1, which is h[63 - kl]; seen from the P0 chain the same sample is w0 = w1 - 64 and
// meets output i at tap kl0 = 4i + 64 - w0.  A function template per chunk: every tap index is a compile-time
// constant, so the taps stay in SGPRs.
// PIN: the packed FMAs as volatile asm, i.e. issued in source order -- all (up to sixteen) FMAs of a sample pair back to
// back; left to the machine scheduler only 0.28 of adjacent FMAs share their sample pair in the CF32 build
// FIRST0: a chain's first FMA takes +0 as an inline constant instead of a cleared accumulator register
// ASYM (round 5): taps that are NOT bit-symmetric.  128 distinct taps do not fit the scalar registers (64 pairs = 128 SGPRs), so the
// P1 chain's taps 127..64 take the 32 SGPR pairs (hs[m] = {h[64 + 2m], h[64 + 2m + 1]}) and the P0 chain's taps 63..0 sit in 32 VGPR
// pairs (hv[m] = {h[2m], h[2m + 1]}): half the FMAs keep the cheaper scalar operand, the window is still read once (79 chunks) and the
// tile, the staging and the stores are the symmetric form's.  Same chains, same order: the contract's bits.
template <bool S32IN, int CIDX, int NB, bool PIN = false, bool FIRST0 = true, bool ASYM = false>
__device__ __forceinline__ void fir_wide_step(const f32x4 *win, f32x4 (&buf)[NB], const f32x2 (&hs)[32], f32x2 (&a1)[8],
                                              f32x2 (&a0)[8], const f32x2 (&hv)[ASYM ? 32 : 1])
{
    // the chunk was read NB steps ago (software pipeline: two waves per SIMD do not hide an LDS round trip by
    // themselves); its register is refilled with the chunk NB steps ahead as soon as it has been consumed
    f32x4 v = buf[CIDX % NB];
    if constexpr (CIDX + NB < DecimWide::WCH) buf[CIDX % NB] = win[(CIDX + NB) + ((CIDX + NB) >> 4)];
    if constexpr (S32IN) {
        v = (f32x4){(float)__float_as_int(v.x), (float)__float_as_int(v.y), (float)__float_as_int(v.z),
                    (float)__float_as_int(v.w)};
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const f32x2 x = s ? __builtin_shufflevector(v, v, 2, 3) : __builtin_shufflevector(v, v, 0, 1);
        if constexpr (CIDX < DecimWide::P1CH) {
            const int w1 = 2 * CIDX + s;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int kl = 4 * i + 64 - w1;
                if (kl >= 0 && kl < 64) {
                    if constexpr (ASYM) {
                        // tap 64 + kl itself: hs holds h[64..127]
                        if (FIRST0 && kl == 63) pk_fma_s_hi_first(a1[i], hs[kl >> 1], x);      // the chain's first tap (127): from +0
                        else if (kl & 1) pk_fma_s_hi(a1[i], hs[kl >> 1], x);
                        else pk_fma_s_lo(a1[i], hs[kl >> 1], x);
                    } else {
                        const int j = 63 - kl;               // h[64 + kl] == h[63 - kl]
                        if constexpr (PIN) { if (j & 1) pk_fma_sv_hi(a1[i], hs[j >> 1], x); else pk_fma_sv_lo(a1[i], hs[j >> 1], x); }
                        else if (FIRST0 && kl == 63) pk_fma_s_lo_first(a1[i], hs[j >> 1], x);       // the chain's first tap (j = 0): from +0
                        else if (j & 1) pk_fma_s_hi(a1[i], hs[j >> 1], x);
                        else pk_fma_s_lo(a1[i], hs[j >> 1], x);
                    }
                }
            }
        }
        if constexpr (CIDX >= DecimWide::P0FROM) {
            const int w0 = 2 * (CIDX - DecimWide::P0FROM) + s;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int kl = 4 * i + 64 - w0;
                if (kl >= 0 && kl < 64) {
                    if constexpr (ASYM) {
                        if (FIRST0 && kl == 63) pk_fma_hi_first(a0[i], hv[kl >> 1], x);        // taps 63..0 from VGPR pairs
                        else if (kl & 1) pk_fma_hi(a0[i], hv[kl >> 1], x);
                        else pk_fma_lo(a0[i], hv[kl >> 1], x);
                    }
                    else if constexpr (PIN) { if (kl & 1) pk_fma_sv_hi(a0[i], hs[kl >> 1], x); else pk_fma_sv_lo(a0[i], hs[kl >> 1], x); }
                    else if (FIRST0 && kl == 63) pk_fma_s_hi_first(a0[i], hs[kl >> 1], x);      // the chain's first tap: from +0
                    else if (kl & 1) pk_fma_s_hi(a0[i], hs[kl >> 1], x);
                    else pk_fma_s_lo(a0[i], hs[kl >> 1], x);
                }
            }
        }
    }
}

template <bool S32IN, int NB, bool PIN, bool ASYM, int... Cs>
__device__ __forceinline__ void fir_wide_steps(std::integer_sequence<int, Cs...>, const f32x4 *win, const f32x2 (&hs)[32],
                                               f32x2 (&a1)[8], f32x2 (&a0)[8], const f32x2 (&hv)[ASYM ? 32 : 1])
{
    f32x4 buf[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) buf[c] = win[c + (c >> 4)];
    (fir_wide_step<S32IN, Cs, NB, PIN, true, ASYM>(win, buf, hs, a1, a0, hv), ...);
}

// ABL (profiling): 0 = the real kernel, 1 = staging + stores without the FIR, 5 = phase stamps (per wave 8 x uint64:
// tiles, cycles issuing DMAs, waiting for data, FIR, transposition + stores, whole wave cycles, whole wave 100 MHz
// ticks, XCC_ID | HW_ID << 8).
// NTL (round 4): DMA instructions 1..16 -- the rows no other tile reads -- are non-temporal loads; 0 (the re-read of the
// previous tile's last kilobyte) and 17, 18 (this tile's last kilobyte, the next tile's halo) stay plain.
// POL (profiling: which cache policy costs the least energy per byte at the power cap): low byte = policy bits OR-ed into
// the nt staging loads (1 = sc0, 16 = sc1), next byte = the stores: 0 nt (shipped), 1 plain, 2 sc0 sc1, 3 sc0 sc1 nt, 4 sc1.
template <int STP>
__device__ __forceinline__ void store16_policy(f32x4 v, f32x4 *dst)
{
    if constexpr (STP == 0) __builtin_nontemporal_store(v, dst);
    else if constexpr (STP == 1) *dst = v;
    else if constexpr (STP == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(v) : "memory");
    else if constexpr (STP == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(dst), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
}

// HALFIN (round 5): CF16 storage (IQ as IEEE half pairs in HBM; fp32 arithmetic; outputs rounded to half once).  The image in LDS
// is the same CF32 image: typed LDS-DMA (buffer_load_format_x ... lds with a {16, FLOAT} descriptor, sxfir_decim_dense.hip.h) lets the
// texture path convert on the way in -- instruction j turns source bytes [128 j, 128 j + 128) = chunks [16 j, 16 j + 16) into the
// slots [17 j, 17 j + 16): the image's pad slot after every 16 chunks falls between instructions.  68 typed instructions per tile
// instead of 19 one-kilobyte DMAs; the FIR below does not know the difference.
template <int ABL = 0, bool S32IN = false, int NB = 24, bool NTL = true, bool PIN = false, int POL = 0, bool HALFIN = false, bool ASYM = false>
__global__ __launch_bounds__(64) void decim4_wide_kernel(const DecimTileArgs a)
{
    using C = DecimWide;
    static_assert(!HALFIN || (!S32IN && ABL == 0 && !PIN && POL == 0), "CF16 storage: the shipped form only");
    static_assert(!ASYM || (ABL == 0 && !PIN && POL == 0), "non-symmetric taps: the shipped form only");
    static_assert(C::CHUNKS % 16 == 0, "whole 16-chunk rows");
    constexpr int SB = HALFIN ? 4 : 8;                    // bytes per complex sample in HBM
    __shared__ __attribute__((aligned(16))) f32x4 img[C::SLOTS];

    unsigned long long wave_c0 = 0, wave_r0 = 0;
    if constexpr (ABL == 5) {
        wave_c0 = __builtin_amdgcn_s_memtime();
        wave_r0 = __builtin_amdgcn_s_memrealtime();
    }
    const int lane = threadIdx.x;
    const int ch = blockIdx.y;
    const float *in = a.in + (SB / 4) * a.in_stride * ch;
    const float *hist = a.hist + (SB / 4) * a.hist_stride * ch;
    float *out = a.out + (SB / 4) * a.out_stride * ch;
    const long long last_chunk = (a.n_in - 1) >> 1;
    const int n_odd = (int)(a.n_in & 1);

    // Tile schedule: in pass i the G waves of a channel cover the G consecutive tiles [i*G, (i+1)*G), dealt so
    // that the waves of one XCD (blockIdx % 8; speed only) hold a contiguous block of the pass.
    const int G = a.n_waves;
    const int b = blockIdx.x;
    int tile = (a.sched == 0 && a.w8) ? (b & 7) * a.w8 + (b >> 3) : b;
    if (tile >= a.n_tiles) return;

    // the 64 distinct taps as SGPR pairs (scalar loads from the constant address space; S32 wire-word plans pass
    // taps already scaled by 2^-31)
    f32x2 hs[32];
    f32x2 hv[ASYM ? 32 : 1];
    {
        const __attribute__((address_space(4))) f32x2 *tq =
            (const __attribute__((address_space(4))) f32x2 *)(S32IN ? a.taps_scaled : a.taps);
#pragma unroll
        for (int m = 0; m < 32; ++m) hs[m] = tq[(ASYM ? 32 : 0) + m];     // ASYM: taps 64..127 (the P1 chain's) in the scalar registers
        if constexpr (ASYM) {
            // ... and taps 0..63 (the P0 chain's) in 32 VGPR pairs, the same value in every lane
            const f32x2 *tv = reinterpret_cast<const f32x2 *>(S32IN ? a.taps_scaled : a.taps);
#pragma unroll
            for (int m = 0; m < 32; ++m) {
                f32x2 t = tv[m];
                asm volatile("" : "+v"(t));              // keep it a vector register pair (a uniform load would land in SGPRs)
                hv[m] = t;
            }
        } else {
            hv[0] = (f32x2){0.0f, 0.0f};
        }
    }

    unsigned boff[HALFIN ? 1 : C::NI];
    if constexpr (!HALFIN) {
#pragma unroll
        for (int j = 0; j < C::NI; ++j) boff[j] = slot_source_offset(64u * j + lane, C::CHUNKS);
    }
    const unsigned img_base = HALFIN ? __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)img) : 0u;   // M0 of the typed DMA = this + a constant

    auto stage = [&](int t) __attribute__((always_inline)) {
        const long long c0 = ((long long)t * C::TILE_IN - C::HALO) >> 1;
        const bool interior = (c0 >= 0) && (c0 + C::CHUNKS - 1 <= last_chunk - n_odd);
        if constexpr (HALFIN) {
            if (interior) {
                // descriptor based at the tile's first byte (64-bit base from scalars, small constant offsets): {16, FLOAT, X <- R}
                const unsigned long long tb = (unsigned long long)(reinterpret_cast<const char *>(in) + 8 * c0);
                v4i32 rs;
                rs.x = __builtin_amdgcn_readfirstlane((int)(unsigned)tb);
                rs.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(tb >> 32)) & 0xffff;
                rs.z = 1 << 20;
                rs.w = 4 | (7 << 12) | (2 << 15);
                unsigned voff = 2u * (unsigned)lane;
                asm volatile("" : "+v"(voff));
#pragma unroll
                for (int j = 0; j < C::CHUNKS / 16; ++j) {
                    const unsigned m0v = __builtin_amdgcn_readfirstlane(img_base + 16u * 17u * (unsigned)j);
                    const unsigned soff = 128u * (unsigned)j;
                    // rows 8..59 of the 68 belong to this tile alone (nt); the first 8 re-read the previous tile's last kilobyte,
                    // the last 8 are the next tile's halo: plain, as instructions 0, 17, 18 of the CF32 form
                    if (NTL && j >= 8 && j < C::CHUNKS / 16 - 8)
                        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_format_x %1, %2, %3 offen nt lds"
                                     :: "s"(m0v), "v"(voff), "s"(rs), "s"(soff) : "memory");
                    else
                        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_format_x %1, %2, %3 offen lds"
                                     :: "s"(m0v), "v"(voff), "s"(rs), "s"(soff) : "memory");
                }
            } else {
                // edge tiles: chunk by chunk through registers (v_cvt_f32_f16: what the typed DMA does for every non-NaN half)
#pragma nounroll
                for (int j = 0; j < C::NI; ++j) {
                    if (j < C::NI - 1 || lane < C::LASTL) {
                        long long cc = c0 + (slot_source_offset(64u * j + lane, C::CHUNKS) >> 4);
                        const unsigned *src;
                        bool one = false;
                        if (cc < 0) {
                            src = reinterpret_cast<const unsigned *>(hist) + 2 * (cc + C::HIST / 2);
                        } else {
                            if (cc > last_chunk) cc = last_chunk;
                            src = reinterpret_cast<const unsigned *>(in) + 2 * cc;
                            one = n_odd && cc == last_chunk;          // the chunk's second sample lies beyond the caller's buffer
                        }
                        const unsigned w0 = src[0], w1 = one ? 0u : src[1];
                        img[64 * j + lane] = (f32x4){half_lo_to_float(w0), half_hi_to_float(w0), half_lo_to_float(w1), half_hi_to_float(w1)};
                    }
                }
            }
            return;
        }
        if (interior) {
            const char *src = reinterpret_cast<const char *>(reinterpret_cast<const f32x4 *>(in) + c0);
#pragma unroll
            for (int j = 0; j < C::NI; ++j) {
                asm volatile("" : "+v"(boff[j]));        // 32-bit offset next to its use (see stage_tile); in place: no copy
                const unsigned bo = boff[j];
                if (j < C::NI - 1 || lane < C::LASTL) {
                    if (NTL && j >= 1 && j <= 16) glds16<2 | (POL & 0xFF)>(src + bo, img + 64 * j);
                    else glds16(src + bo, img + 64 * j);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < C::NI; ++j) {
                unsigned bo = boff[j];
                asm volatile("" : "+v"(bo));
                long long cc = c0 + (bo >> 4);
                const f32x4 *src;
                if (cc < 0) {
                    src = reinterpret_cast<const f32x4 *>(hist) + (cc + C::HIST / 2);
                } else {
                    if (cc > last_chunk) cc = last_chunk;
                    src = reinterpret_cast<const f32x4 *>(in) + cc;
                }
                if (j < C::NI - 1 || lane < C::LASTL) {
                    if (n_odd && cc == last_chunk) {
                        // the chunk's second sample lies beyond the caller's buffer: 8 bytes through a register
                        const float2 v = *reinterpret_cast<const float2 *>(src);
                        img[64 * j + lane] = (f32x4){v.x, v.y, 0.0f, 0.0f};
                    } else {
                        glds16(src, img + 64 * j);
                    }
                }
            }
        }
    };

    if (b == a.hist_wave) {
        float *ho = a.hist_out + (SB / 4) * a.hist_stride * ch;
        for (int j = lane; j < C::HIST; j += 64) {
            const long long s = a.n_in - C::HIST + j;
            if constexpr (HALFIN) {
                reinterpret_cast<unsigned *>(ho)[j] = s >= 0 ? reinterpret_cast<const unsigned *>(in)[s] : reinterpret_cast<const unsigned *>(hist)[s + C::HIST];
            } else {
                const float2 v = s >= 0 ? reinterpret_cast<const float2 *>(in)[s] : reinterpret_cast<const float2 *>(hist)[s + C::HIST];
                reinterpret_cast<float2 *>(ho)[j] = v;
            }
        }
    }

    // lane l: outputs 8l..8l+7 of the tile; window from chunk 16l (lane stride 17 slots: conflict free)
    const f32x4 *win = img + 17 * lane;

    unsigned long long ph[5] = {0, 0, 0, 0, 0}, tk = 0;
    if constexpr (ABL == 5) tk = __builtin_amdgcn_s_memtime();
#define SXFIR_WIDE_PHASE(k) \
    if constexpr (ABL == 5) { \
        const unsigned long long t_now = __builtin_amdgcn_s_memtime(); \
        ph[k] += t_now - tk; \
        tk = t_now; \
    }

    const int swz_w = (lane & 1) ^ ((lane >> 1) & 3), swz_r = ((lane >> 2) & 1) ^ ((lane >> 3) & 3);   // the output transposition's swizzles
    int ntile = 0;
    for (; tile < a.n_tiles; tile += G) {
        stage(tile);
        SXFIR_WIDE_PHASE(1)
        SXFIR_WAIT_VMCNT(0);
        SXFIR_WIDE_PHASE(2)

        f32x2 a1[8], a0[8];
        if constexpr (ABL == 1 || PIN) {
#pragma unroll
            for (int i = 0; i < 8; ++i) a1[i] = a0[i] = (f32x2){0.0f, 0.0f};
        }
        if constexpr (ABL == 1) {
            const f32x4 v0 = win[0], v1 = win[17];
            a0[0] = (f32x2){v0.x + hs[0].x, v0.y};
            a0[3] = (f32x2){v0.z, v0.w + hs[31].y};
            a0[4] = (f32x2){v1.x, v1.y};
            a0[7] = (f32x2){v1.z, v1.w};
        } else {
            fir_wide_steps<S32IN, NB, PIN, ASYM>(std::make_integer_sequence<int, C::WCH>{}, win, hs, a1, a0, hv);
        }
        f32x4 y[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)                       // y = P0 + P1, one rounding each
            y[k] = (f32x4){__fadd_rn(a0[2 * k].x, a1[2 * k].x), __fadd_rn(a0[2 * k].y, a1[2 * k].y),
                           __fadd_rn(a0[2 * k + 1].x, a1[2 * k + 1].x), __fadd_rn(a0[2 * k + 1].y, a1[2 * k + 1].y)};
        if constexpr (ABL == 5) asm volatile("" ::"v"(y[0].x), "v"(y[3].w));
        SXFIR_WIDE_PHASE(3)

        const long long m0 = (long long)tile * C::TILE_OUT;
        if (m0 + C::TILE_OUT <= a.n_out) {
            // transposed through the dead image so that each store instruction writes 1 KiB of consecutive
            // addresses: output chunk c (256 per tile) sits at slot c ^ ((c >> 2) & 1) ^ ((c >> 3) & 3) -- no pad slots.
            // Lane l writes its chunks 4l + k at 4l + (k ^ tw), tw = (l & 1) ^ ((l >> 1) & 3): the 8 lanes a
            // ds_write_b128 is served with hit 8 different slots mod 8; lane l then reads chunks l + 64k' at
            // 64k' + (l ^ tr), tr = ((l >> 2) & 1) ^ ((l >> 3) & 3): the 16 lanes of a ds_read_b128 service group hit
            // 16 different slots mod 16 (searched with tools/lds_bank_model.py's groups: 0 conflicts either way; the
            // padded layout of rounds 2-4, c + (c >> 4), had 2-way write conflicts and one pair per read group).
            // (written and read by this wave only: LDS operations of one wave complete in order)
#pragma unroll
            for (int k = 0; k < 4; ++k) img[4 * lane + (k ^ swz_w)] = y[k];
            if constexpr (HALFIN) {
                // two outputs per lane and store: 8 bytes of half pairs, 512 consecutive bytes per instruction
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                u32x2 *dst = reinterpret_cast<u32x2 *>(out + m0);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f32x4 v = img[64 * k + (lane ^ swz_r)];
                    __builtin_nontemporal_store((u32x2){pack_half2(v.x, v.y), pack_half2(v.z, v.w)}, dst + 64 * k + lane);
                }
            } else {
                f32x4 *dst = reinterpret_cast<f32x4 *>(out + 2 * m0);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f32x4 v = img[64 * k + (lane ^ swz_r)];
                    store16_policy<(POL >> 8)>(v, dst + 64 * k + lane);
                }
            }
        } else {
            // ragged last tile of the call: element by element, straight from the registers
            const long long m = m0 + 8 * lane;
            if constexpr (HALFIN) {
                unsigned *dst = reinterpret_cast<unsigned *>(out + m);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (m + 2 * k < a.n_out) dst[2 * k] = pack_half2(y[k].x, y[k].y);
                    if (m + 2 * k + 1 < a.n_out) dst[2 * k + 1] = pack_half2(y[k].z, y[k].w);
                }
            } else {
                float *dst = out + 2 * m;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (m + 2 * k < a.n_out) { dst[4 * k] = y[k].x; dst[4 * k + 1] = y[k].y; }
                    if (m + 2 * k + 1 < a.n_out) { dst[4 * k + 2] = y[k].z; dst[4 * k + 3] = y[k].w; }
                }
            }
        }
        // the next tile's DMA overwrites the image only after these LDS reads have returned
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ++ntile;
        SXFIR_WIDE_PHASE(4)
    }
    if constexpr (ABL == 5) {
        ph[0] = (unsigned long long)ntile;
        const unsigned long long wave_c1 = __builtin_amdgcn_s_memtime(), wave_r1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0 && a.stamps) {
            unsigned long long *rec = a.stamps + 8 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
#pragma unroll
            for (int k = 0; k < 5; ++k) rec[k] = ph[k];
            rec[5] = wave_c1 - wave_c0;
            rec[6] = wave_r1 - wave_r0;
            unsigned xcc, hwid;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            rec[7] = (unsigned long long)(xcc & 15u) | ((unsigned long long)hwid << 8);
        }
    }
#undef SXFIR_WIDE_PHASE
}

}  // namespace sxfir
