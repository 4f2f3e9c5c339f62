// Decimate-by-4 tile kernel, second generation (gfx950).  Same arithmetic, same
// LDS image and the same numeric contract as sxfir_decim_tile.hip.h (whose
// building blocks it reuses); what changes is how a wave's memory operations are
// ordered and how the work is handed out:
//
//   DEFER   a tile's two output stores are issued only AFTER the wait for the next
//           tile's LDS-DMA.  s_waitcnt vmcnt counts loads and stores together, in
//           issue order, so a persistent wave that stores tile i and then stages
//           tile i+1 cannot see its DMA land before the stores of tile i have been
//           acknowledged; deferred, a store has a whole arithmetic phase to retire
//           before anything waits on it.  Costs 8 VGPRs (the transposed outputs).
//   TAPSEP  the taps are DMA'd into their own 512 bytes of LDS at the same time as
//           the first tile instead of through the tile image before it: one memory
//           round trip per wave instead of two (matters for short-lived waves).
//   DBUF    two tile images per wave; the next tile's DMA is issued before the
//           current tile is computed and retired by a counted vmcnt.
//   WPG     waves per workgroup, each with a private tile image (no barriers): the
//           taps' LDS copy and the dispatch cost are shared.
//
// New code: the reference decimates inside the SX1255 (SoapySX.cpp:180-208 only
// programs the divider); this kernel plays that role for SoapySX::readStream
// (SoapySX.cpp:868-967).
#pragma once

#include "sxfir_decim_tile.hip.h"

namespace sxfir {

enum { T2_DEFER = 1, T2_TAPSEP = 2, T2_DBUF = 4, T2_PLAINST = 8 };

// HBM -> LDS for one tile into an image of exactly C::SLOTS slots (the last DMA
// instruction is issued for the lanes that still fall inside it).
template <int NT>
__device__ __forceinline__ void stage_tile2(const DecimTileCtx<NT> &c, int tile, f32x4 *buf)
{
    using C = DecimTile4<NT>;
    constexpr int LAST = C::SLOTS - 64 * (C::NLOAD - 1);      // lanes of the last instruction
    const long long c0 = ((long long)tile * C::TILE_IN - C::HALO) >> 1;
    const bool interior = (c0 >= 0) && (c0 + C::CHUNKS - 1 <= c.last_chunk - c.n_odd);
    if (interior) {
        const char *src = reinterpret_cast<const char *>(reinterpret_cast<const f32x4 *>(c.in) + c0);
#pragma unroll
        for (int i = 0; i < C::NLOAD; ++i) {
            unsigned b = c.boff[i];
            asm volatile("" : "+v"(b));                  // 32-bit offset next to its use (see stage_tile)
            if (i < C::NLOAD - 1 || LAST >= 64 || c.lane < LAST) glds16(src + b, buf + 64 * i);
        }
    } else {
#pragma unroll
        for (int i = 0; i < C::NLOAD; ++i) {
            unsigned b = c.boff[i];
            asm volatile("" : "+v"(b));
            long long ch = c0 + (b >> 4);
            const f32x4 *src;
            if (ch < 0) {
                src = reinterpret_cast<const f32x4 *>(c.hist) + (ch + C::HIST / 2);
            } else {
                if (ch > c.last_chunk) ch = c.last_chunk;
                src = reinterpret_cast<const f32x4 *>(c.in) + ch;
            }
            if (i < C::NLOAD - 1 || LAST >= 64 || c.lane < LAST) stage_edge_chunk(c, ch, src, buf + 64 * i);
        }
    }
}

// FIR arithmetic of one tile out of its LDS image: the 8 outputs of this lane pair,
// reduced over the two tap halves (lane l keeps outputs 0-3, lane l+32 outputs 4-7).
template <int NT, bool S32IN>
__device__ __forceinline__ void fir_tile_pk(const f32x4 *win, const f32x2 (&hp)[NT / 4], float (&oi)[4], float (&oq)[4])
{
    using C = DecimTile4<NT>;
    f32x2 acc[C::R];
#pragma unroll
    for (int i = 0; i < C::R; ++i) acc[i] = (f32x2){0.0f, 0.0f};
#pragma unroll
    for (int t = 0; t < C::WCH; ++t) {
        f32x4 v = win[t + (t >> 4)];
        if constexpr (S32IN) {
            // S32_LE wire words (convert_rx_buffer, SoapySX.cpp:103-112): int -> float here, the exact
            // 2^-31 scale is folded into the taps (a power of two commutes with the fused multiply-add)
            v = (f32x4){(float)__float_as_int(v.x), (float)__float_as_int(v.y), (float)__float_as_int(v.z),
                        (float)__float_as_int(v.w)};
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int w = 2 * t + s;
            const f32x2 x = s ? __builtin_shufflevector(v, v, 2, 3) : __builtin_shufflevector(v, v, 0, 1);
#pragma unroll
            for (int i = 0; i < C::R; ++i) {
                const int kl = 4 * i + C::TPL - w;
                if (kl >= 0 && kl < C::TPL) {
                    if (kl & 1) pk_fma_hi(acc[i], hp[kl >> 1], x);
                    else pk_fma_lo(acc[i], hp[kl >> 1], x);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float a0 = acc[i].x, a1 = acc[i + 4].x, b0 = acc[i].y, b1 = acc[i + 4].y;
        permlane32_swap(a0, a1);
        permlane32_swap(b0, b1);
        oi[i] = __fadd_rn(a0, a1);
        oq[i] = __fadd_rn(b0, b1);
    }
}

template <bool PLAIN>
__device__ __forceinline__ void st16(const f32x4 &v, f32x4 *dst)
{
    if constexpr (PLAIN) *dst = v;
    else __builtin_nontemporal_store(v, dst);
}

template <int NT, int WPG, int OPT, int ABL = 0, bool S32IN = false>
__global__ __launch_bounds__(64 * WPG) void decim4_tile2_kernel(const DecimTileArgs a)
{
    using C = DecimTile4<NT>;
    constexpr bool DEFER = (OPT & T2_DEFER) != 0, TAPSEP = (OPT & T2_TAPSEP) != 0, DBUF = (OPT & T2_DBUF) != 0;
    constexpr bool PLAINST = (OPT & T2_PLAINST) != 0;
    constexpr int IMG = C::SLOTS;                       // slots per tile image
    constexpr int NB = DBUF ? 2 : 1;
    __shared__ __attribute__((aligned(16))) f32x4 lds[WPG * NB * IMG + (TAPSEP ? NT / 4 : 0)];

    const int ww = WPG > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;
    f32x4 *img = lds + ww * (NB * IMG);
    f32x4 *tapbuf = TAPSEP ? lds + WPG * NB * IMG : img;

    DecimTileCtx<NT> c;
    c.lane = threadIdx.x & 63;
    c.g = c.lane & 31;
    c.p = c.lane >> 5;
    const int ch = blockIdx.y;
    c.in = a.in + 2 * a.in_stride * ch;
    c.hist = a.hist + 2 * a.hist_stride * ch;
    c.out = a.out + 2 * a.out_stride * ch;
    c.n_out = a.n_out;
    c.last_chunk = (a.n_in - 1) >> 1;
    c.n_odd = (int)(a.n_in & 1);
#pragma unroll
    for (int i = 0; i < C::NLOAD; ++i) {
        const unsigned q = 64u * i + c.lane;
        unsigned off = q - (((q + 1u) * 3856u) >> 16);                    // (q+1)/17, exact for q < 4096
        off = off < (unsigned)C::CHUNKS ? off : (unsigned)C::CHUNKS - 1u;
        c.boff[i] = 16u * off;
    }

    // Tile schedule: in pass i the G workgroups of a channel cover the G*WPG consecutive tiles
    // [i*G*WPG, (i+1)*G*WPG); wave ww of workgroup b takes tile (S(b) + i*G)*WPG + ww, where S deals the
    // workgroups of one XCD (blockIdx % 8 shares an XCD; speed only) a contiguous block of the pass,
    // so that halo re-reads stay in that XCD's L2.  a.sched == 2: S(b) = b (plain dispatch order).
    const int G = a.n_waves;                            // workgroups per channel
    const int b = blockIdx.x;
    const int S = (a.sched == 0 && a.w8) ? (b & 7) * a.w8 + (b >> 3) : b;
    int tile = S * WPG + ww;
    const int tile_step = G * WPG;
    if (tile >= a.n_tiles) return;

    if (b * WPG + ww == a.hist_wave) write_history<NT>(c, a.hist_out + 2 * a.hist_stride * ch, a.n_in);

    // taps of this lane's half as 64-bit pairs for the packed FMAs
    f32x2 hp[C::TPL / 2];
    auto read_taps = [&]() __attribute__((always_inline)) {
        const f32x4 *tp = tapbuf + (C::TPL / 4) * c.p;
#pragma unroll
        for (int k = 0; k < C::TPL / 4; ++k) {
            f32x4 t = tp[k];
            if constexpr (S32IN) t = t * 4.656612873077393e-10f;
            hp[2 * k] = (f32x2){t.x, t.y};
            hp[2 * k + 1] = (f32x2){t.z, t.w};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    if (c.lane < NT / 4) glds16(reinterpret_cast<const char *>(a.taps) + 16 * c.lane, tapbuf);
    if constexpr (!TAPSEP) {
        // through the (still empty) tile image: the reads must have returned before the first tile's DMA
        SXFIR_WAIT_VMCNT(0);
        read_taps();
    }

    const int u0c = 16 * c.g - (NT / 4) * c.p + NT / 4;   // this lane's first window chunk (multiple of 16)
    const int woff = u0c + (u0c >> 4);

    // outputs of the previous tile, transposed for whole-line stores, waiting to be stored (DEFER)
    f32x4 pend0 = {0, 0, 0, 0}, pend1 = {0, 0, 0, 0};
    f32x4 *pend_dst = nullptr;
    bool pending = false;

    auto flush = [&]() __attribute__((always_inline)) {
        if (pending) {
            st16<PLAINST>(pend0, pend_dst + c.lane);
            st16<PLAINST>(pend1, pend_dst + 64 + c.lane);
            pending = false;
        }
    };

    // arithmetic + output transposition of `tile` out of image `buf`
    auto process = [&](int t, f32x4 *buf) __attribute__((always_inline)) {
        const long long m0 = (long long)t * C::TILE_OUT;
        float oi[4], oq[4];
        if constexpr (ABL == 1) {
            // memory side alone: staging + stores, no FIR
            const f32x4 v0 = buf[woff], v1 = buf[woff + 17];
#pragma unroll
            for (int i = 0; i < 4; ++i) { oi[i] = v0[i] + hp[0].x; oq[i] = v1[i] + hp[31 % (C::TPL / 2)].y; }
        } else {
            fir_tile_pk<NT, S32IN>(buf + woff, hp, oi, oq);
        }
        if (m0 + C::TILE_OUT <= c.n_out) {
            // through the now dead image: chunk 4g + 2p + {0,1} of the tile's 128 output chunks, read back
            // linearly, so that each global store instruction writes 1 KiB of consecutive addresses
            const int oc = 4 * c.g + 2 * c.p;
            buf[oc + (oc >> 4)] = (f32x4){oi[0], oq[0], oi[1], oq[1]};
            buf[oc + 1 + (oc >> 4)] = (f32x4){oi[2], oq[2], oi[3], oq[3]};
            const f32x4 v0 = buf[c.lane + (c.lane >> 4)], v1 = buf[68 + c.lane + (c.lane >> 4)];
            f32x4 *dst = reinterpret_cast<f32x4 *>(c.out + 2 * m0);
            if constexpr (DEFER) {
                pend0 = v0;
                pend1 = v1;
                pend_dst = dst;
                pending = true;
            } else {
                st16<PLAINST>(v0, dst + c.lane);
                st16<PLAINST>(v1, dst + 64 + c.lane);
            }
        } else {
            // ragged last tile of the call: element by element, straight from the registers
            const long long m = m0 + 8 * c.g + 4 * c.p;
            float *dst = c.out + 2 * m;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (m + i < c.n_out) { dst[2 * i] = oi[i]; dst[2 * i + 1] = oq[i]; }
        }
        // the image may be overwritten by the next DMA only after these LDS reads have returned
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };

    if constexpr (!DBUF) {
        if constexpr (ABL != 2) stage_tile2<NT>(c, tile, img);
        SXFIR_WAIT_VMCNT(0);
        if constexpr (TAPSEP) read_taps();
        while (true) {
            if constexpr (DEFER) flush();
            process(tile, img);
            tile += tile_step;
            if (tile >= a.n_tiles) break;
            if constexpr (ABL != 2) stage_tile2<NT>(c, tile, img);
            // LDS-DMA completion is ordered for this wave's ds_reads only by its own vmcnt
            SXFIR_WAIT_VMCNT(0);
        }
        if constexpr (DEFER) flush();
    } else {
        // vmcnt bookkeeping, oldest first, at the wait of iteration k:
        //   DEFER:  DMA(k) | stores(k-2) | DMA(k+1)      -> vmcnt(NLOAD) retires DMA(k) and stores that
        //           have had the whole arithmetic phase of tile k-1 to complete
        //   else:   DMA(k) | stores(k-1) | DMA(k+1)      -> vmcnt(NLOAD) also waits for fresh stores
        static_assert(C::NLOAD == 10, "vmcnt immediate below assumes NLOAD == 10");
        if constexpr (ABL != 2) stage_tile2<NT>(c, tile, img);
        int cur = 0;
        bool first = true;
        while (true) {
            const int next = tile + tile_step;
            if (next < a.n_tiles) {
                if constexpr (ABL != 2) stage_tile2<NT>(c, next, img + (cur ^ 1) * IMG);
                SXFIR_WAIT_VMCNT(10);
            } else {
                SXFIR_WAIT_VMCNT(0);
            }
            if constexpr (TAPSEP) {
                if (first) { read_taps(); first = false; }
            }
            if constexpr (DEFER) flush();
            process(tile, img + cur * IMG);
            if (next >= a.n_tiles) break;
            tile = next;
            cur ^= 1;
        }
        if constexpr (DEFER) flush();
    }
}

}  // namespace sxfir
