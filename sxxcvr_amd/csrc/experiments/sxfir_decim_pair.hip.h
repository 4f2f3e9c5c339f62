// Decimate-by-4, 128 taps: the two tap halves on two WAVES (gfx950).
//
// The /4 kernels are bound by the energy of their packed FMAs (LABBOOK.md 5.1: the board sits at its power
// cap), and the cheapest FMA has its tap in an SGPR pair.  A scalar operand is wave-uniform; instead of
// giving every lane all taps (sxfir_decim_tile2.hip.h, T2_SCALAR: needs symmetric taps to fit the SGPR file and
// 71 LDS reads per 512 FMAs) this kernel gives every WAVE one tap half:
//
//   * workgroup = 2 waves sharing one tile of 512 outputs = 2048 inputs + 128-sample halo (halo 6 % instead
//     of 12.5 %), LDS image 1156 slots = 18 496 B per workgroup, 8 workgroups = 16 waves per CU;
//   * wave p holds taps [64p, 64p+64) in 32 SGPR pairs (any taps, no symmetry needed) and computes, for all
//     512 outputs, the partial sum over its half: lane l -> outputs 8l..8l+7, window of 47 ds_read_b128 for 512
//     v_pk_fma_f32 with a scalar tap operand (the first-generation kernel's LDS traffic, the scalar kernel's
//     operand traffic, no cross-lane reduction);
//   * the halves meet through LDS: each wave writes the partials of the partner's 32 lanes (4 ds_write_b128
//     per writing lane), barrier, reads the partner's partials for its own 32 lanes and adds them once
//     (y = P0 + P1, the numeric contract of DESIGN.md 3, unchanged), then transposes its 256 outputs through
//     its own part of the dead image and stores whole lines.
//   * each wave stages its own part of the image (9 / 10 LDS-DMA instructions) and keeps its exchange and
//     transposition buffers inside that part, so that a wave only ever overwrites slots it alone still needs:
//     four barriers per tile (data visible | image dead | partials written | partials read).
//
// New code: the reference decimates inside the SX1255 (SoapySX.cpp:180-208 only programs the divider); this
// kernel plays that role for SoapySX::readStream (SoapySX.cpp:868-967).
#pragma once

#include <utility>

#include "../sxfir_decim_tile.hip.h"
#include "../sxfir_decim_tile2.hip.h"

namespace sxfir {

struct DecimPair {
    static constexpr int NT = 128, D = 4, R = 8;
    static constexpr int TILE_OUT = 64 * R;               // 512
    static constexpr int TILE_IN = TILE_OUT * D;          // 2048
    static constexpr int HALO = NT, HIST = NT;
    static constexpr int CHUNKS = (TILE_IN + HALO) / 2;   // 1088
    static constexpr int SLOTS = CHUNKS + CHUNKS / 16;    // 1156: one pad slot after every 16 chunks
    static constexpr int Q1 = 576;                        // wave 0 stages slots [0, Q1), wave 1 [Q1, SLOTS)
    static constexpr int NI0 = Q1 / 64;                   // 9 DMA instructions
    static constexpr int NI1 = (SLOTS - Q1 + 63) / 64;    // 10, the last one for LAST1 lanes
    static constexpr int LAST1 = SLOTS - Q1 - 64 * (NI1 - 1);
    static constexpr int NIMAX = NI1;
    static constexpr int WCH = 47;                        // window chunks per lane: samples 1..92 of 94
    static constexpr int XSLOTS = 128 + 8;                // exchange / transposition buffer: 128 chunks + pads
    static_assert(Q1 % 64 == 0 && XSLOTS <= Q1 && XSLOTS <= SLOTS - Q1, "buffers live inside a wave's own part");
};

// One window chunk (step T of 47): sample w = 2T + s meets output i at local tap kl = 4i + 64 - w of this wave's
// half.  A function template per step: every tap index is a compile-time constant, so the taps stay in SGPRs.
template <bool S32IN, int T>
__device__ __forceinline__ void fir_half_step(const f32x4 *win, const f32x2 (&hs)[32], f32x2 (&acc)[8])
{
    f32x4 v = win[T + (T >> 4)];
    if constexpr (S32IN) {
        v = (f32x4){(float)__float_as_int(v.x), (float)__float_as_int(v.y), (float)__float_as_int(v.z),
                    (float)__float_as_int(v.w)};
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int w = 2 * T + s;
        const f32x2 x = s ? __builtin_shufflevector(v, v, 2, 3) : __builtin_shufflevector(v, v, 0, 1);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int kl = 4 * i + 64 - w;
            if (kl >= 0 && kl < 64) {
                if (kl & 1) pk_fma_s_hi(acc[i], hs[kl >> 1], x);
                else pk_fma_s_lo(acc[i], hs[kl >> 1], x);
            }
        }
    }
}

template <bool S32IN, int... Ts>
__device__ __forceinline__ void fir_half_steps(std::integer_sequence<int, Ts...>, const f32x4 *win, const f32x2 (&hs)[32],
                                               f32x2 (&acc)[8])
{
    (fir_half_step<S32IN, Ts>(win, hs, acc), ...);
}

// ABL (profiling): 0 = the real kernel, 1 = staging + stores without the FIR, 5 = phase stamps (per wave 8 x uint64:
// tiles, cycles issuing DMAs, waiting for data + barrier, FIR, exchange + transposition + stores, whole wave cycles,
// whole wave 100 MHz ticks, XCC_ID | HW_ID << 8).
// XSEP: the partials are exchanged through a buffer of their own (4 KiB more LDS per workgroup: 7 instead of 8
// workgroups per CU) instead of through the dead image, which takes two of the four barriers per tile away.
template <int ABL = 0, bool S32IN = false, bool XSEP = false>
__global__ __launch_bounds__(128) void decim4_pair_kernel(const DecimTileArgs a)
{
    using C = DecimPair;
    __shared__ __attribute__((aligned(16))) f32x4 img[C::SLOTS + (XSEP ? 2 * C::XSLOTS : 0)];

    unsigned long long wave_c0 = 0, wave_r0 = 0;
    if constexpr (ABL == 5) {
        wave_c0 = __builtin_amdgcn_s_memtime();
        wave_r0 = __builtin_amdgcn_s_memrealtime();
    }
    const int lane = threadIdx.x & 63;
    const int p = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave = tap half
    const int ch = blockIdx.y;
    const float *in = a.in + 2 * a.in_stride * ch;
    const float *hist = a.hist + 2 * a.hist_stride * ch;
    float *out = a.out + 2 * a.out_stride * ch;
    const long long last_chunk = (a.n_in - 1) >> 1;
    const int n_odd = (int)(a.n_in & 1);

    // this wave's 64 taps as SGPR pairs (scalar loads from the constant address space; S32 wire-word plans pass
    // taps already scaled by 2^-31)
    f32x2 hs[32];
    {
        const __attribute__((address_space(4))) f32x2 *tq =
            (const __attribute__((address_space(4))) f32x2 *)(S32IN ? a.taps_scaled : a.taps) + 32 * p;
#pragma unroll
        for (int m = 0; m < 32; ++m) hs[m] = tq[m];
    }

    // this wave's part of the image and the source offsets of its DMA instructions
    const int q0 = p ? C::Q1 : 0;
    unsigned boff[C::NIMAX];
#pragma unroll
    for (int j = 0; j < C::NIMAX; ++j) boff[j] = slot_source_offset((unsigned)q0 + 64u * j + lane, C::CHUNKS);

    auto stage = [&](int tile) __attribute__((always_inline)) {
        const long long c0 = ((long long)tile * C::TILE_IN - C::HALO) >> 1;
        const bool interior = (c0 >= 0) && (c0 + C::CHUNKS - 1 <= last_chunk - n_odd);
        f32x4 *part = img + q0;
        if (interior) {
            const char *src = reinterpret_cast<const char *>(reinterpret_cast<const f32x4 *>(in) + c0);
#pragma unroll
            for (int j = 0; j < C::NIMAX; ++j) {
                unsigned b = boff[j];
                asm volatile("" : "+v"(b));              // 32-bit offset next to its use (see stage_tile)
                if (j < C::NI0 || (p && (j < C::NI1 - 1 || lane < C::LAST1))) glds16(src + b, part + 64 * j);
            }
        } else {
#pragma unroll
            for (int j = 0; j < C::NIMAX; ++j) {
                unsigned b = boff[j];
                asm volatile("" : "+v"(b));
                long long cc = c0 + (b >> 4);
                const f32x4 *src;
                if (cc < 0) {
                    src = reinterpret_cast<const f32x4 *>(hist) + (cc + C::HIST / 2);
                } else {
                    if (cc > last_chunk) cc = last_chunk;
                    src = reinterpret_cast<const f32x4 *>(in) + cc;
                }
                if (j < C::NI0 || (p && (j < C::NI1 - 1 || lane < C::LAST1))) {
                    if (n_odd && cc == last_chunk) {
                        // the chunk's second sample lies beyond the caller's buffer: 8 bytes through a register
                        const float2 v = *reinterpret_cast<const float2 *>(src);
                        part[64 * j + lane] = (f32x4){v.x, v.y, 0.0f, 0.0f};
                    } else {
                        glds16(src, part + 64 * j);
                    }
                }
            }
        }
    };

    // Tile schedule: in pass i the G workgroups of a channel cover the G consecutive tiles [i*G, (i+1)*G), dealt
    // so that the workgroups of one XCD (blockIdx % 8; speed only) hold a contiguous block of the pass.
    const int G = a.n_waves;                            // workgroups per channel
    const int b = blockIdx.x;
    int tile = (a.sched == 0 && a.w8) ? (b & 7) * a.w8 + (b >> 3) : b;
    if (tile >= a.n_tiles) return;                      // (whole workgroups leave together)

    if (b == a.hist_wave && p == 0) {
        float *ho = a.hist_out + 2 * a.hist_stride * ch;
        for (int j = lane; j < C::HIST; j += 64) {
            const long long s = a.n_in - C::HIST + j;
            const float2 v = s >= 0 ? reinterpret_cast<const float2 *>(in)[s] : reinterpret_cast<const float2 *>(hist)[s + C::HIST];
            reinterpret_cast<float2 *>(ho)[j] = v;
        }
    }

    // lane l: outputs 8l..8l+7 of the tile; window from chunk 16l - 32p + 32 (a multiple of 16)
    const int u0c = 16 * lane - 32 * p + 32;
    const f32x4 *win = img + (u0c + (u0c >> 4));
    // exchange / transposition buffer at the start of this wave's part, the partner's at the start of its part
    f32x4 *xown = img + q0;                              // transposition buffer (and, without XSEP, exchange buffer)
    f32x4 *xsend = XSEP ? img + C::SLOTS + p * C::XSLOTS : xown;
    const f32x4 *xpartner = XSEP ? img + C::SLOTS + (1 - p) * C::XSLOTS : img + (p ? 0 : C::Q1);
    const bool mine = (lane >> 5) == p;                 // this lane's outputs are finished by this wave
    const int l5 = lane & 31;

    unsigned long long ph[5] = {0, 0, 0, 0, 0}, tk = 0;
    if constexpr (ABL == 5) tk = __builtin_amdgcn_s_memtime();
#define SXFIR_PAIR_PHASE(k) \
    if constexpr (ABL == 5) { \
        const unsigned long long t_now = __builtin_amdgcn_s_memtime(); \
        ph[k] += t_now - tk; \
        tk = t_now; \
    }

    int ntile = 0;
    for (; tile < a.n_tiles; tile += G) {
        stage(tile);
        SXFIR_PAIR_PHASE(1)
        SXFIR_WAIT_VMCNT(0);                            // own DMAs landed ...
        __syncthreads();                                // ... and the partner's (A)
        SXFIR_PAIR_PHASE(2)

        f32x2 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = (f32x2){0.0f, 0.0f};
        if constexpr (ABL == 1) {
            const f32x4 v0 = win[0], v1 = win[17];
            acc[0] = (f32x2){v0.x + hs[0].x, v0.y};
            acc[3] = (f32x2){v0.z, v0.w + hs[31].y};
            acc[4] = (f32x2){v1.x, v1.y};
            acc[7] = (f32x2){v1.z, v1.w};
        } else {
            fir_half_steps<S32IN>(std::make_integer_sequence<int, C::WCH>{}, win, hs, acc);
        }
        if constexpr (ABL == 5) asm volatile("" ::"v"(acc[0].x), "v"(acc[7].y));
        SXFIR_PAIR_PHASE(3)
        // (XSEP: the exchange buffer is nobody's image, and the partner's reads of the previous tile's partials lie
        // before the barrier (A) it passed with this wave)
        if constexpr (!XSEP) __syncthreads();           // both waves are done reading the image (B)

        // partials of the partner's lanes -> own buffer: lane l5 of the partner's half, chunks 4*l5 .. 4*l5+3
        if (!mine) {
            const int oc = 4 * l5;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                xsend[oc + k + ((oc + k) >> 4)] = (f32x4){acc[2 * k].x, acc[2 * k].y, acc[2 * k + 1].x, acc[2 * k + 1].y};
        }
        __syncthreads();                                // partials written, and both waves are done with the image (C)
        f32x4 y[4];
        if (mine) {
            const int oc = 4 * l5;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 o = xpartner[oc + k + ((oc + k) >> 4)];
                // y = P0 + P1 (one rounding each; which wave holds which half does not matter to the sum)
                y[k] = (f32x4){__fadd_rn(acc[2 * k].x, o.x), __fadd_rn(acc[2 * k].y, o.y), __fadd_rn(acc[2 * k + 1].x, o.z),
                               __fadd_rn(acc[2 * k + 1].y, o.w)};
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (!XSEP) __syncthreads();           // partials read: the buffers may be reused (D)

        // this wave's 256 outputs: m0 + 256p + 8*l5 + i on its own lanes
        const long long m0 = (long long)tile * C::TILE_OUT + 256 * p;
        if (m0 + 256 <= a.n_out) {
            // transposed through the wave's own buffer so that each store instruction writes 1 KiB of
            // consecutive addresses: 128 output chunks, lane l then stores chunks l and 64 + l
            if (mine) {
                const int oc = 4 * l5;
#pragma unroll
                for (int k = 0; k < 4; ++k) xown[oc + k + ((oc + k) >> 4)] = y[k];
            }
            // (written and read by this wave only: LDS operations of one wave complete in order)
            const f32x4 v0 = xown[lane + (lane >> 4)], v1 = xown[68 + lane + (lane >> 4)];
            f32x4 *dst = reinterpret_cast<f32x4 *>(out + 2 * m0);
            __builtin_nontemporal_store(v0, dst + lane);
            __builtin_nontemporal_store(v1, dst + 64 + lane);
        } else if (mine) {
            // ragged last tile of the call: element by element, straight from the registers
            const long long m = m0 + 8 * l5;
            float *dst = out + 2 * m;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (m + 2 * k < a.n_out) { dst[4 * k] = y[k].x; dst[4 * k + 1] = y[k].y; }
                if (m + 2 * k + 1 < a.n_out) { dst[4 * k + 2] = y[k].z; dst[4 * k + 3] = y[k].w; }
            }
        }
        // the next tile's DMA overwrites this wave's part only, and only after these LDS reads have returned
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ++ntile;
        SXFIR_PAIR_PHASE(4)
    }
    if constexpr (ABL == 5) {
        ph[0] = (unsigned long long)ntile;
        const unsigned long long wave_c1 = __builtin_amdgcn_s_memtime(), wave_r1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0 && a.stamps) {
            unsigned long long *rec = a.stamps + 8 * ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 2 + p);
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
#undef SXFIR_PAIR_PHASE
}

}  // namespace sxfir
