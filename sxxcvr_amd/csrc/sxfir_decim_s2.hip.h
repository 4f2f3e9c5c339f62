// Decimate-by-4, 128 taps, taps as SCALAR operands (gfx950).
//
// Measured on MI355X (tools/valu_power_probe.hip, profiles/round2c_*): the /4 tile kernel is bound by its
// packed FMAs at the clock the chip's power management allows (in-kernel 1.43 GHz with all of HBM, LDS-DMA,
// LDS reads and VALU busy), and a v_pk_fma_f32 whose tap operand is an SGPR pair instead of a VGPR pair lets
// the same arithmetic run at a 13 % higher clock (two 64-bit VGPR reads per instruction instead of three).
// A scalar operand is wave-uniform, so the work split changes: every lane computes 4 consecutive outputs over
// ALL 128 taps (no tap halves on lane pairs, no cross-lane reduction); the taps stream through SGPRs in eight
// blocks of 16 per tile (scalar loads, served by the scalar cache), the lane's window is 71 ds_read_b128.
// Same tile (256 outputs = 1024 inputs + 128-sample halo), same LDS-DMA staging, same whole-line stores.
//
// Numeric contract unchanged (jsplit = 2, cw = 4): per output and per I/Q, P1 = fmaf chain from
// +0.0f over taps 127..64, P0 over taps 63..0, y = P0 + P1 -- now inside one lane.
//
// New code: the reference decimates inside the SX1255 (SoapySX.cpp:180-208 only programs the divider).
#pragma once

#include "sxfir_decim_tile.hip.h"
#include "sxfir_decim_tile2.hip.h"

namespace sxfir {

struct DecimS2 {
    static constexpr int NT = 128, D = 4, R = 4;
    static constexpr int TILE_OUT = 64 * R;               // 256
    static constexpr int TILE_IN = TILE_OUT * D;          // 1024
    static constexpr int HALO = NT, HIST = NT;
    static constexpr int CHUNKS = (TILE_IN + HALO) / 2;   // 576
    static constexpr int PADP = 8;                        // lane stride in chunks; one pad slot after every PADP
    static constexpr int SLOTS = CHUNKS + CHUNKS / PADP - 1;   // 647: the pad after the last chunk is not needed
    static constexpr int NLOAD = (SLOTS + 63) / 64;       // 11
    static constexpr int LASTL = SLOTS - 64 * (NLOAD - 1);
    static constexpr int WCH = 71;                        // window chunks per lane: samples u = 0..141 of which 1..140 are used
};

__device__ __forceinline__ void pk_fma_s_lo(f32x2 &acc, const f32x2 &hpair, const f32x2 &x)
{
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "s"(hpair), "v"(x));
}
__device__ __forceinline__ void pk_fma_s_hi(f32x2 &acc, const f32x2 &hpair, const f32x2 &x)
{
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(hpair), "v"(x));
}

// OPT: T2_DEFER, T2_PLAINST as in decim4_tile2_kernel.  ABL 0 = real, 1 = memory side alone, 5 = phase stamps.
template <int OPT, int ABL = 0>
__global__ __launch_bounds__(64) void decim4_s2_kernel(const DecimTileArgs a)
{
    using C = DecimS2;
    constexpr bool DEFER = (OPT & T2_DEFER) != 0, PLAINST = (OPT & T2_PLAINST) != 0;
    __shared__ __attribute__((aligned(16))) f32x4 img[C::SLOTS];

    unsigned long long wave_c0 = 0, wave_r0 = 0;
    if constexpr (ABL == 5) {
        wave_c0 = __builtin_amdgcn_s_memtime();
        wave_r0 = __builtin_amdgcn_s_memrealtime();
    }
    const int lane = threadIdx.x;
    const int ch = blockIdx.y;
    const float *in = a.in + 2 * a.in_stride * ch;
    const float *hist = a.hist + 2 * a.hist_stride * ch;
    float *out = a.out + 2 * a.out_stride * ch;
    const long long last_chunk = (a.n_in - 1) >> 1;
    const int n_odd = (int)(a.n_in & 1);
    // wave-uniform taps through the scalar data path (constant address space -> s_load)
    const __attribute__((address_space(4))) f32x2 *tp = (const __attribute__((address_space(4))) f32x2 *)a.taps;

    // byte offset (from the tile's first staged chunk) of the chunk DMA instruction j brings to slot 64j + lane
    unsigned boff[C::NLOAD];
#pragma unroll
    for (int j = 0; j < C::NLOAD; ++j) {
        const unsigned q = 64u * j + lane;
        unsigned off = q - (((q + 1u) * 7282u) >> 16);                     // (q+1)/9, exact for q < 4096
        off = off < (unsigned)C::CHUNKS ? off : (unsigned)C::CHUNKS - 1u;
        boff[j] = 16u * off;
    }

    auto stage = [&](int tile) __attribute__((always_inline)) {
        const long long c0 = ((long long)tile * C::TILE_IN - C::HALO) >> 1;
        const bool interior = (c0 >= 0) && (c0 + C::CHUNKS - 1 <= last_chunk - n_odd);
        if (interior) {
            const char *src = reinterpret_cast<const char *>(reinterpret_cast<const f32x4 *>(in) + c0);
#pragma unroll
            for (int j = 0; j < C::NLOAD; ++j) {
                unsigned b = boff[j];
                asm volatile("" : "+v"(b));              // 32-bit offset next to its use (see stage_tile)
                if (j < C::NLOAD - 1 || lane < C::LASTL) glds16(src + b, img + 64 * j);
            }
        } else {
#pragma unroll
            for (int j = 0; j < C::NLOAD; ++j) {
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
                if (j < C::NLOAD - 1 || lane < C::LASTL) {
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

    const int G = a.n_waves;
    const int b = blockIdx.x;
    const int S = (a.sched == 0 && a.w8) ? (b & 7) * a.w8 + (b >> 3) : b;
    int tile = S;
    if (tile >= a.n_tiles) return;
    if (b == a.hist_wave) {
        float *ho = a.hist_out + 2 * a.hist_stride * ch;
        for (int j = lane; j < C::HIST; j += 64) {
            const long long s = a.n_in - C::HIST + j;
            const float2 v = s >= 0 ? reinterpret_cast<const float2 *>(in)[s] : reinterpret_cast<const float2 *>(hist)[s + C::HIST];
            reinterpret_cast<float2 *>(ho)[j] = v;
        }
    }

    // lane l: outputs 4l..4l+3 of the tile; window = chunks 8l .. 8l+70 (slot 9l + t + t/8)
    const f32x4 *win = img + 9 * lane;

    unsigned long long ph[5] = {0, 0, 0, 0, 0}, tk = 0;
    if constexpr (ABL == 5) tk = __builtin_amdgcn_s_memtime();
#define SXFIR_S2_PHASE(k) \
    if constexpr (ABL == 5) { \
        const unsigned long long t_now = __builtin_amdgcn_s_memtime(); \
        ph[k] += t_now - tk; \
        tk = t_now; \
    }

    f32x4 pend0 = {0, 0, 0, 0}, pend1 = {0, 0, 0, 0};
    f32x4 *pend_dst = nullptr;
    bool pending = false;
    auto flush = [&]() __attribute__((always_inline)) {
        if (pending) {
            st16<PLAINST>(pend0, pend_dst + lane);
            st16<PLAINST>(pend1, pend_dst + 64 + lane);
            pending = false;
        }
    };

    auto process = [&](int t_idx) __attribute__((always_inline)) {
        const long long m0 = (long long)t_idx * C::TILE_OUT;
        f32x2 acc[4], p1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc[i] = (f32x2){0.0f, 0.0f}; p1[i] = (f32x2){0.0f, 0.0f}; }
        if constexpr (ABL == 1) {
            const f32x4 v0 = win[0], v1 = win[9];
            acc[0] = (f32x2){v0.x, v0.y}; acc[1] = (f32x2){v0.z, v0.w}; acc[2] = (f32x2){v1.x, v1.y}; acc[3] = (f32x2){v1.z, v1.w};
        } else {
            // Tap block j holds taps 16j..16j+15.  Output i meets tap k at window sample u = 4i + 128 - k, so block
            // j covers u in [113 - 16j, 140 - 16j]; walking u upwards inside a block and the blocks downwards
            // gives every output its taps in descending order, P1 (taps 127..64) before P0 (63..0).
            const __attribute__((address_space(4))) f32x2 *tq = tp;
#pragma unroll
            for (int j = 7; j >= 0; --j) {
                // the block's 16 taps as 8 aligned SGPR pairs; the empty asm keeps the loads inside the tile loop
                // and at this point of it (128 taps do not fit the SGPR file at once)
                asm volatile("" : "+s"(tq));
                f32x2 hp[8];
#pragma unroll
                for (int m = 0; m < 8; ++m) hp[m] = tq[8 * j + m];
#pragma unroll
                for (int uu = 113 - 16 * j; uu <= 140 - 16 * j; ++uu) {
                    const f32x4 vv = win[(uu >> 1) + (uu >> 1) / C::PADP];     // chunks shared by two blocks are read once (CSE)
                    const f32x2 x = (uu & 1) ? __builtin_shufflevector(vv, vv, 2, 3) : __builtin_shufflevector(vv, vv, 0, 1);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int k = 4 * i + 128 - uu;
                        if (k >= 16 * j && k < 16 * j + 16) {
                            const int kl = k - 16 * j;
                            if (kl & 1) pk_fma_s_hi(acc[i], hp[kl >> 1], x);
                            else pk_fma_s_lo(acc[i], hp[kl >> 1], x);
                        }
                    }
                }
                if (j == 4) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { p1[i] = acc[i]; acc[i] = (f32x2){0.0f, 0.0f}; }
                }
            }
        }
        float oi[4], oq[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            oi[i] = __fadd_rn(acc[i].x, p1[i].x);
            oq[i] = __fadd_rn(acc[i].y, p1[i].y);
        }
        if constexpr (ABL == 5) asm volatile("" ::"v"(oi[0]), "v"(oq[3]));
        SXFIR_S2_PHASE(3)
        if (m0 + C::TILE_OUT <= a.n_out) {
            // lane l holds output chunks 2l and 2l+1; through the dead image so that each store instruction
            // writes 1 KiB of consecutive addresses (one pad slot after every 16 keeps both sides conflict free)
            const int oc = 2 * lane;
            img[oc + (oc >> 4)] = (f32x4){oi[0], oq[0], oi[1], oq[1]};
            img[oc + 1 + (oc >> 4)] = (f32x4){oi[2], oq[2], oi[3], oq[3]};
            const f32x4 v0 = img[lane + (lane >> 4)], v1 = img[68 + lane + (lane >> 4)];
            f32x4 *dst = reinterpret_cast<f32x4 *>(out + 2 * m0);
            if constexpr (DEFER) {
                pend0 = v0;
                pend1 = v1;
                pend_dst = dst;
                pending = true;
            } else {
                st16<PLAINST>(v0, dst + lane);
                st16<PLAINST>(v1, dst + 64 + lane);
            }
        } else {
            const long long m = m0 + 4 * lane;
            float *dst = out + 2 * m;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (m + i < a.n_out) { dst[2 * i] = oi[i]; dst[2 * i + 1] = oq[i]; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };

    int ntile = 0;
    stage(tile);
    SXFIR_S2_PHASE(1)
    SXFIR_WAIT_VMCNT(0);
    SXFIR_S2_PHASE(2)
    while (true) {
        if constexpr (DEFER) flush();
        process(tile);
        ++ntile;
        SXFIR_S2_PHASE(4)
        tile += G;
        if (tile >= a.n_tiles) break;
        stage(tile);
        SXFIR_S2_PHASE(1)
        SXFIR_WAIT_VMCNT(0);
        SXFIR_S2_PHASE(2)
    }
    if constexpr (DEFER) flush();
    if constexpr (ABL == 5) {
        ph[0] = (unsigned long long)ntile;
        const unsigned long long wave_c1 = __builtin_amdgcn_s_memtime(), wave_r1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0 && a.stamps) {
            unsigned long long *rec = a.stamps + 8 * ((size_t)(blockIdx.y * gridDim.x + blockIdx.x));
#pragma unroll
            for (int k = 0; k < 5; ++k) rec[k] = ph[k];
            rec[5] = wave_c1 - wave_c0;
            rec[6] = wave_r1 - wave_r0;
            rec[7] = 0;
        }
    }
#undef SXFIR_S2_PHASE
}

}  // namespace sxfir
