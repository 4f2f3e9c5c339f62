// LDS-tiled polyphase FIR decimator for gfx950 (MI355X), decimate-by-4.
//
// This is new code: the reference (tejeez/sxxcvr) has no software FIR; the
// SX1255 chip decimates in silicon and SoapySX.cpp:180-208 / :1197-1208 only
// program its divider.  The kernel plays the role of that on-chip decimator,
// feeding the stream that SoapySX::readStream (SoapySX.cpp:868-967) hands out.
//
// Work decomposition (wave64, one wave = one workgroup, no barriers):
//   * a wave owns a few tiles of one channel (strided passes over the stream,
//     see the schedule in the kernel); a tile is 256 outputs = 1024 input
//     samples (8 KiB) plus a 128-sample halo;
//   * the tile is staged in LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per
//     wave-instruction, coalesced 16 B per lane), taps live in VGPRs (fetched
//     once per wave through an LDS broadcast); with DBUF the next tile's DMA is
//     issued before the current tile is computed and retired by a counted
//     s_waitcnt vmcnt (two LDS buffers per wave; measured slower, not shipped);
//   * lanes l and l+32 form a pair: both compute the same R = 8 consecutive
//     outputs, lane half p = l >> 5 over the tap range [NT/2*p, NT/2*(p+1));
//     each 16-byte ds_read_b128 (two complex samples) feeds up to 32
//     v_pk_fma_f32 (the I and Q fused multiply-adds of one tap in one packed
//     instruction, compute_tile_pk);
//   * the two partial dot products are combined with v_permlane32_swap
//     (gfx950) + one add, which also leaves outputs 0-3 on the low lane and
//     4-7 on the high lane; the 32 bytes a lane ends with go through LDS so
//     that every store instruction writes whole lines.
// LDS image: 16-byte chunks, one pad chunk after every 16, so the 16 lanes of
// a ds_read_b128 group (lane stride 256 B) hit 16 different 16-byte slots.
//
// Numeric contract (DESIGN.md "Numeric contract", jsplit=2, cw=4): per output
// and per I/Q, partial_p = fmaf chain from +0.0f over the taps of half p in
// DESCENDING k; y = partial_0 + partial_1.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sxfir {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct DecimTileArgs {
    const float *in;        // channel 0, sample 0 of this call (16-byte aligned)
    const float *hist;      // channel 0 history: HIST samples preceding `in`
    float *hist_out;        // where the wave of the last tile leaves the history for the next call
    float *out;             // channel 0, first output of this call
    const float *taps;      // NT floats (device)
    const float *taps_scaled;   // the same taps times 2^-31 (S32 wire-word plans of the scalar-tap kernel)
    long long n_in;         // new input samples per channel
    long long n_out;        // outputs per channel
    long long in_stride;    // samples between channels
    long long out_stride;
    long long hist_stride;
    int n_tiles;            // tiles per channel
    int n_waves;            // waves (workgroups) per channel
    int sched;              // 0 = strided passes (XCD-blocked), 1 = one contiguous run per wave, 2 = plain strided passes
    // schedule constants worked out on the host (no integer divisions in the wave's prologue)
    int w8;                 // n_waves / 8 when n_waves is a multiple of 8, else 0 (sched 0)
    int run_base, run_extra;    // n_tiles / n_waves, n_tiles % n_waves (sched 1)
    int hist_wave;          // the wave whose tiles include the last one: it carries the history over
    // short tail (scalar-tap kernel): waves [0, long_waves) make strided passes over the first long_tiles tiles,
    // waves [long_waves, n_waves) take ONE tile each of the rest; long_waves == 0: every wave makes strided passes
    int long_waves, long_tiles, long_w8, short_w8;
    unsigned long long *stamps;   // diagnostic builds only (ABL 11/12): per-wave {shader cycles, 100 MHz ticks}
    // the first 64 taps by value (scaled by 2^-31 for S32 wire-word plans): a scalar-tap kernel loads them from the
    // kernel-argument segment together with everything else, one scalar-load round trip instead of two
    float taps_k[64];
};

template <int NT>
struct DecimTile4 {
    static constexpr int D = 4;
    static constexpr int R = 8;                       // outputs per lane
    static constexpr int TPL = NT / 2;                // taps per lane
    static constexpr int TILE_OUT = 32 * R;           // 256
    static constexpr int TILE_IN = TILE_OUT * D;      // 1024
    static constexpr int HALO = NT;                   // >= NT-1, multiple of 64
    static constexpr int HIST = NT;                   // history samples kept per channel
    static constexpr int CHUNKS = (TILE_IN + HALO) / 2;
    static constexpr int SLOTS = CHUNKS + CHUNKS / 16;
    static constexpr int NLOAD = (SLOTS + 63) / 64;
    static constexpr int BUF_SLOTS = NLOAD * 64;
    static constexpr int WMAX = D * (R - 1) + TPL;    // highest window sample index used
    static constexpr int WCH = WMAX / 2 + 1;          // window chunks per lane
    static_assert(NT % 64 == 0, "tile kernel needs NT % 64 == 0");
};

// AUX = cache policy bits of the load (0 = default, 2 = nt: streaming, 1 = sc0, 16 = sc1)
template <int AUX = 0>
__device__ __forceinline__ void glds16(const void *gsrc, void *ldst)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)ldst, 16, 0, AUX);
}

// v_permlane32_swap_b32 vdst, src (gfx950): lanes 32-63 of vdst <-> lanes 0-31
// of src.  Inline asm on purpose: with two DIFFERENT operands hipcc 7.2's
// __builtin_amdgcn_permlane32_swap returns the first result register for both
// elements of its result pair (seen in the .s: "v_permlane32_swap v1, v2" then
// v1 used for r[0] and r[1]).  "s_nop 1" = the 2 wait states the ISA requires
// between a VALU write of an operand and the swap.
__device__ __forceinline__ void permlane32_swap(float &vdst, float &src)
{
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(vdst), "+v"(src));
}

template <int NT>
struct DecimTileCtx {
    const float *in, *hist;
    float *out;
    long long n_out, last_chunk;
    int n_odd;              // n_in is odd: the last chunk holds one valid sample (never fetched as 16 bytes)
    int lane, g, p;
    // byte offset (from the tile's first staged chunk) of the chunk each DMA instruction fetches for
    // this lane: tile-invariant, 32 bits, so that the DMA uses the SGPR-base + VGPR-offset form
    unsigned boff[DecimTile4<NT>::NLOAD];
};

// One 16-byte chunk of an edge tile.  With an odd n_in the last chunk holds one valid sample: its second
// half lies beyond the caller's buffer (possibly beyond the allocation) and is never touched; that lane
// fetches 8 bytes through a register instead of taking part in the DMA.
template <int NT>
__device__ __forceinline__ void stage_edge_chunk(const DecimTileCtx<NT> &c, long long ch, const f32x4 *src, f32x4 *slot0)
{
    if (c.n_odd && ch == c.last_chunk) {
        const float2 v = *reinterpret_cast<const float2 *>(src);
        slot0[c.lane] = (f32x4){v.x, v.y, 0.0f, 0.0f};
    } else {
        glds16(src, slot0);
    }
}

// HBM -> LDS for one tile, no VGPR round trip.  Slot q = 64*i + lane of the
// buffer holds logical chunk q - (q+1)/17 (a pad slot re-loads its left
// neighbour and is never read).
// AUX: cache policy bits of every DMA (profiling modes), or -1 (the default, round 4): instructions 1..7 are
// non-temporal loads; the last two -- the tile's last kilobyte, which the next tile reads again as its halo -- and
// instruction 0 -- this tile's re-read of the previous one's -- stay plain loads: whichever of the two reaches the
// XCD's L2 first allocates the line, the other hits (tools/membench5.hip, LABBOOK.md 5.1 round 4).
template <int NT, int AUX = -1>
__device__ __forceinline__ void stage_tile(const DecimTileCtx<NT> &c, int tile, f32x4 *buf)
{
    static_assert(DecimTile4<NT>::NLOAD == 10, "the halo of the next tile lies in DMA instructions 8 and 9");
    using C = DecimTile4<NT>;
    const long long c0 = ((long long)tile * C::TILE_IN - C::HALO) >> 1;   // first chunk staged (may be < 0)
    const bool interior = (c0 >= 0) && (c0 + C::CHUNKS - 1 <= c.last_chunk - c.n_odd);
    if (interior) {
        const char *src = reinterpret_cast<const char *>(reinterpret_cast<const f32x4 *>(c.in) + c0);
#pragma unroll
        for (int i = 0; i < C::NLOAD; ++i) {
            // the empty asm keeps the 32->64-bit extension next to the load (instruction selection
            // works per basic block), which is what selects the SGPR-base + 32-bit VGPR offset form
            unsigned b = c.boff[i];
            asm volatile("" : "+v"(b));
            if constexpr (AUX >= 0) glds16<AUX>(src + b, buf + 64 * i);
            else if (i >= 1 && i < 8) glds16<2>(src + b, buf + 64 * i);
            else glds16<0>(src + b, buf + 64 * i);
        }
    } else {
#pragma unroll
        for (int i = 0; i < C::NLOAD; ++i) {
            unsigned b = c.boff[i];
            asm volatile("" : "+v"(b));                  // edge tiles are rare: nothing of this hoisted out of the loop
            long long ch = c0 + (b >> 4);
            const f32x4 *src;
            if (ch < 0) {
                src = reinterpret_cast<const f32x4 *>(c.hist) + (ch + C::HIST / 2);
            } else {
                if (ch > c.last_chunk) ch = c.last_chunk;
                src = reinterpret_cast<const f32x4 *>(c.in) + ch;
            }
            stage_edge_chunk(c, ch, src, buf + 64 * i);
        }
    }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NT>
__device__ __forceinline__ void store_tile(const DecimTileCtx<NT> &c, int tile, const float (&oi)[4],
                                           const float (&oq)[4], f32x4 *xbuf)
{
    using C = DecimTile4<NT>;
    // A lane holds 4 consecutive outputs (32 bytes); stored directly that is two instructions that
    // each write 16 of every 32 bytes.  Through LDS instead (the tile image is dead by now and this
    // wave owns it): chunk 4g + 2p + {0,1} of the tile's 128 output chunks, read back linearly, so
    // that each global store instruction writes 1 KiB of consecutive addresses (whole lines).
    const long long m0 = (long long)tile * C::TILE_OUT;
    if (m0 + C::TILE_OUT <= c.n_out) {
        // one pad slot after every 16 keeps both the scattered writes (lane stride 4 chunks) and the
        // linear read-back free of bank conflicts
        const int oc = 4 * c.g + 2 * c.p;
        xbuf[oc + (oc >> 4)] = (f32x4){oi[0], oq[0], oi[1], oq[1]};
        xbuf[oc + 1 + (oc >> 4)] = (f32x4){oi[2], oq[2], oi[3], oq[3]};
        f32x4 *dst = reinterpret_cast<f32x4 *>(c.out + 2 * m0);
        const f32x4 v0 = xbuf[c.lane + (c.lane >> 4)], v1 = xbuf[68 + c.lane + (c.lane >> 4)];
        __builtin_nontemporal_store(v0, dst + c.lane);
        __builtin_nontemporal_store(v1, dst + 64 + c.lane);
    } else {
        const long long m = m0 + 8 * c.g + 4 * c.p;
        float *dst = c.out + 2 * m;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (m + i < c.n_out) { dst[2 * i] = oi[i]; dst[2 * i + 1] = oq[i]; }
    }
}

// Same arithmetic with the I and Q FMAs of one (tap, sample) pair issued as one
// v_pk_fma_f32 (two independent IEEE fused multiply-adds per instruction, so the
// results are bit-identical).  The tap is broadcast to both halves by op_sel: taps
// live in 64-bit register pairs {h[2k], h[2k+1]} and op_sel / op_sel_hi pick the
// low or the high dword for both lanes of the packed operation.  Inline asm keeps
// the register picture of the scalar loop (the compiler's own packing of this loop
// needs 178 VGPRs).
__device__ __forceinline__ void pk_fma_lo(f32x2 &acc, const f32x2 &hpair, const f32x2 &x)
{
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(hpair), "v"(x));
}
__device__ __forceinline__ void pk_fma_hi(f32x2 &acc, const f32x2 &hpair, const f32x2 &x)
{
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(hpair), "v"(x));
}
// a chain's first FMA: acc = fmaf(tap, x, +0.0f), the zero as an inline constant (no register cleared first)
__device__ __forceinline__ void pk_fma_hi_first(f32x2 &acc, const f32x2 &hpair, const f32x2 &x)
{
    asm("v_pk_fma_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,1,0]" : "=v"(acc) : "v"(hpair), "v"(x));
}
__device__ __forceinline__ void pk_fma_lo_first(f32x2 &acc, const f32x2 &hpair, const f32x2 &x)
{
    asm("v_pk_fma_f32 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "=v"(acc) : "v"(hpair), "v"(x));
}

template <int NT, bool S32IN = false>
__device__ __forceinline__ void compute_tile_pk(const DecimTileCtx<NT> &c, int tile, const f32x4 *win,
                                                const float (&h)[NT / 2], f32x4 *xbuf)
{
    using C = DecimTile4<NT>;
    f32x2 hp[C::TPL / 2];
#pragma unroll
    for (int k = 0; k < C::TPL / 2; ++k) hp[k] = (f32x2){h[2 * k], h[2 * k + 1]};
    f32x2 acc[C::R];
#pragma unroll
    for (int i = 0; i < C::R; ++i) acc[i] = (f32x2){0.0f, 0.0f};
#pragma unroll
    for (int t = 0; t < C::WCH; ++t) {
        f32x4 v = win[t + (t >> 4)];
        if constexpr (S32IN) {
            // S32_LE wire words (convert_rx_buffer, SoapySX.cpp:103-112): only the int->float conversion
            // happens here; the exact 2^-31 scale is folded into the taps by the caller, which gives the
            // same bits as scaling every sample (a power of two commutes with the fused multiply-add)
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
    float oi[4], oq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float a0 = acc[i].x, a1 = acc[i + 4].x, b0 = acc[i].y, b1 = acc[i + 4].y;
        permlane32_swap(a0, a1);
        permlane32_swap(b0, b1);
        oi[i] = __fadd_rn(a0, a1);
        oq[i] = __fadd_rn(b0, b1);
    }
    store_tile<NT>(c, tile, oi, oq, xbuf);
}

// WORK_DIV / WORK_MODE are profiling knobs (wrong results): WORK_DIV = 2 with mode 0 halves both the
// LDS reads and the FMAs, mode 1 keeps all LDS reads but halves the FMAs, mode 2 halves the LDS reads
// and keeps all FMAs (every chunk used twice).
template <int NT, bool S32IN = false, int WORK_DIV = 1, int WORK_MODE = 0>
__device__ __forceinline__ void compute_tile(const DecimTileCtx<NT> &c, int tile, const f32x4 *win,
                                             const float (&h)[NT / 2], f32x4 *xbuf)
{
    using C = DecimTile4<NT>;
    // window sample w (ascending) meets output i at local tap kl = 4*i + TPL - w;
    // ascending w = descending k, per the contract
    float ai[C::R], aq[C::R];
#pragma unroll
    for (int i = 0; i < C::R; ++i) { ai[i] = 0.0f; aq[i] = 0.0f; }

#pragma unroll
    for (int t = 0; t < (WORK_MODE == 0 ? C::WCH / WORK_DIV : C::WCH); ++t) {
        if constexpr (WORK_MODE == 1) {
            if (t % WORK_DIV != 0) {                    // read but do not use
                const f32x4 dead = win[t + (t >> 4)];
                asm volatile("" ::"v"(dead.x), "v"(dead.y), "v"(dead.z), "v"(dead.w));
                continue;
            }
        }
        f32x4 v = win[WORK_MODE == 2 ? (t / WORK_DIV) * WORK_DIV + ((t / WORK_DIV) * WORK_DIV >> 4) : t + (t >> 4)];
        if constexpr (S32IN) {
            // S32_LE wire words (convert_rx_buffer, SoapySX.cpp:103-112): only the int->float conversion
            // happens here; the exact 2^-31 scale is folded into the taps by the caller, which gives the
            // same bits as scaling every sample (a power of two commutes with the fused multiply-add)
            v = (f32x4){(float)__float_as_int(v.x), (float)__float_as_int(v.y), (float)__float_as_int(v.z),
                        (float)__float_as_int(v.w)};
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int w = 2 * t + s;
            const float xi = s ? v.z : v.x;
            const float xq = s ? v.w : v.y;
#pragma unroll
            for (int i = 0; i < C::R; ++i) {
                const int kl = 4 * i + C::TPL - w;
                if (kl >= 0 && kl < C::TPL) {
                    ai[i] = __builtin_fmaf(h[kl], xi, ai[i]);
                    aq[i] = __builtin_fmaf(h[kl], xq, aq[i]);
                }
            }
        }
    }

    // combine the two tap halves: lanes l (p=0) and l+32 (p=1).
    // swap(vdst = acc[i], src = acc[i+4]): high half of acc[i] <-> low half of
    // acc[i+4].  Afterwards acc[i] + acc[i+4] is partial_0 + partial_1 of output
    // i on the low lane and of output i+4 on the high lane.
    float oi[4], oq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        permlane32_swap(ai[i], ai[i + 4]);
        permlane32_swap(aq[i], aq[i + 4]);
        oi[i] = __fadd_rn(ai[i], ai[i + 4]);
        oq[i] = __fadd_rn(aq[i], aq[i + 4]);
    }

    store_tile<NT>(c, tile, oi, oq, xbuf);
}

// History carry-over fused into the launch (no second kernel): the wave that owns the last tile
// copies the last HIST samples of (hist ++ in) into the plan's OTHER history buffer (the current
// one is still being read by the wave of tile 0).
template <int NT>
__device__ __forceinline__ void write_history(const DecimTileCtx<NT> &c, float *hist_out, long long n_in)
{
    using C = DecimTile4<NT>;
#pragma unroll
    for (int j = c.lane; j < C::HIST; j += 64) {
        const long long s = n_in - C::HIST + j;
        const float2 v = s >= 0 ? reinterpret_cast<const float2 *>(c.in)[s]
                                : reinterpret_cast<const float2 *>(c.hist)[s + C::HIST];
        reinterpret_cast<float2 *>(hist_out)[j] = v;
    }
}

#define SXFIR_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

// ABL (profiling builds only; 3 = scalar v_fmac_f32 arithmetic instead of v_pk_fma_f32, same bits): 0 = the real kernel, 1 = stage + store but no FIR arithmetic
// (memory side alone), 2 = FIR arithmetic on whatever LDS holds, no staging (compute side alone).
// S32IN: the input (and the history) are S32_LE I2S wire words instead of CF32.
template <int NT, bool DBUF, int ABL = 0, bool S32IN = false>
__global__ __launch_bounds__(64) void decim4_tile_kernel(const DecimTileArgs a)
{
    using C = DecimTile4<NT>;
    __shared__ __attribute__((aligned(16))) f32x4 lds[(DBUF ? 2 : 1) * C::BUF_SLOTS];

    DecimTileCtx<NT> c;
    c.lane = threadIdx.x;
    c.g = c.lane & 31;
    c.p = c.lane >> 5;
    const int ch = blockIdx.y;
    c.in = a.in + 2 * a.in_stride * ch;
    c.hist = a.hist + 2 * a.hist_stride * ch;
    c.out = a.out + 2 * a.out_stride * ch;
    c.n_out = a.n_out;
    c.last_chunk = (a.n_in - 1) >> 1;                 // last input chunk holding a valid sample
    c.n_odd = (int)(a.n_in & 1);
#pragma unroll
    for (int i = 0; i < C::NLOAD; ++i) {
        const unsigned q = 64u * i + c.lane;
        unsigned off = q - (((q + 1u) * 3856u) >> 16);                    // (q+1)/17, exact for q < 4096
        off = off < (unsigned)C::CHUNKS ? off : (unsigned)C::CHUNKS - 1u;
        c.boff[i] = 16u * off;
    }

    // taps of this lane's half, h[kl] = taps[TPL*p + kl].  All lanes of a half hold the same 64 values:
    // loading them per lane costs 16 global_load_dwordx4 that each return 1 KiB (16 KiB per wave, more
    // than a tile).  Instead NT/4 lanes DMA the NT taps into the still empty tile image once and every lane
    // reads its half back with TPL/4 broadcast ds_read_b128 (2.8 % of the kernel time at 4 tiles per wave).
    float h[C::TPL];
    auto load_taps = [&]() __attribute__((always_inline)) {
        if constexpr (ABL == 23) {
#pragma unroll
            for (int k = 0; k < C::TPL; ++k) h[k] = __int_as_float(0x3a000000 + 64 * k + c.p);   // no tap traffic (wrong results)
        } else if constexpr (ABL == 24) {
#pragma unroll
            for (int k = 0; k < C::TPL; ++k) h[k] = a.taps[C::TPL * c.p + k];     // per-lane global loads (A/B)
        } else {
            if (c.lane < NT / 4) glds16(reinterpret_cast<const char *>(a.taps) + 16 * c.lane, lds);
            SXFIR_WAIT_VMCNT(0);
            const f32x4 *tp = lds + (C::TPL / 4) * c.p;
#pragma unroll
            for (int k = 0; k < C::TPL / 4; ++k) {
                const f32x4 t = tp[k];
                h[4 * k] = t.x; h[4 * k + 1] = t.y; h[4 * k + 2] = t.z; h[4 * k + 3] = t.w;
            }
            // the reads must have returned before the first tile's DMA overwrites these slots
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr (S32IN) {
#pragma unroll
                for (int k = 0; k < C::TPL; ++k) h[k] = __fmul_rn(h[k], 4.656612873077393e-10f);
            }
        }
    };

    // this lane's window chunk 0 inside a buffer
    const int u0c = 16 * c.g - (NT / 4) * c.p + NT / 4;   // logical chunk, multiple of 16
    const f32x4 *win0 = lds + (u0c + (u0c >> 4));

    // Tile schedule.  SCHED_STRIDE (default): in pass i the W waves of the grid cover the W
    // consecutive tiles [i*W, (i+1)*W), so the chip walks the stream front to back like a copy
    // kernel (compact DRAM window) instead of W far-apart cursors.  Inside a pass the tiles are
    // dealt so that the waves of one XCD (blockIdx % 8 shares an XCD; speed only) hold a
    // contiguous block, which keeps halo re-reads in that XCD's L2.
    // SCHED_RUN: one contiguous run of tiles per wave (balanced to within one tile).
    const int wave = blockIdx.x;
    const int W = a.n_waves;
    int tile_begin, tile_end, tile_step;
    if (a.sched == 0) {
        tile_begin = a.w8 ? (wave & 7) * a.w8 + (wave >> 3) : wave;
        tile_end = a.n_tiles;
        tile_step = W;
    } else if (a.sched == 2) {
        // SCHED_PLAIN: pass i covers tiles [i*W, (i+1)*W) in workgroup order
        tile_begin = wave;
        tile_end = a.n_tiles;
        tile_step = W;
    } else {
        tile_begin = wave * a.run_base + (wave < a.run_extra ? wave : a.run_extra);
        tile_end = tile_begin + a.run_base + (wave < a.run_extra ? 1 : 0);
        tile_step = 1;
    }
    if (tile_begin >= tile_end) return;

    // the wave that owns the last tile also carries the history over (before it issues any DMA, so
    // the counted vmcnt waits of the double-buffered loop are not disturbed)
    if (wave == a.hist_wave) write_history<NT>(c, a.hist_out + 2 * a.hist_stride * ch, a.n_in);

    unsigned long long st_c0 = 0, st_r0 = 0;
    if constexpr (ABL == 11 || ABL == 12) {
        st_c0 = __builtin_amdgcn_s_memtime();
        st_r0 = __builtin_amdgcn_s_memrealtime();
    }
    if constexpr (!DBUF) {
        auto stage = [&](int tile) __attribute__((always_inline)) {
            if constexpr (ABL == 20) stage_tile<NT, 2>(c, tile, lds);          // nt loads
            else if constexpr (ABL == 21) stage_tile<NT, 16>(c, tile, lds);    // sc1 loads
            else if constexpr (ABL == 22) stage_tile<NT, 1>(c, tile, lds);     // sc0 loads
            else if constexpr (ABL != 2 && ABL != 12 && (ABL < 17 || ABL > 19)) stage_tile<NT>(c, tile, lds);
        };
        load_taps();
        for (int tile = tile_begin; tile < tile_end; tile += tile_step) {
            stage(tile);
            // LDS-DMA completion is ordered for this wave's ds_reads only by its own vmcnt
            SXFIR_WAIT_VMCNT(0);
            if constexpr (ABL == 7) {
                compute_tile<NT, false, 2>(c, tile, win0, h, lds);      // half the FMAs and LDS reads
            } else if constexpr (ABL == 8) {
                compute_tile<NT, false, 4>(c, tile, win0, h, lds);      // a quarter
            } else if constexpr (ABL == 9) {
                compute_tile<NT, false, 2, 1>(c, tile, win0, h, lds);   // all LDS reads, half the FMAs
            } else if constexpr (ABL == 10) {
                compute_tile<NT, false, 2, 2>(c, tile, win0, h, lds);   // half the LDS reads, all FMAs
            } else if constexpr (ABL == 17) {
                compute_tile<NT, false, 2, 2>(c, tile, win0, h, lds);   // no staging, half the LDS reads, all FMAs
            } else if constexpr (ABL == 18) {
                compute_tile<NT, false, 2, 1>(c, tile, win0, h, lds);   // no staging, all LDS reads, half the FMAs
            } else if constexpr (ABL == 19) {
                compute_tile<NT, false, 8, 2>(c, tile, win0, h, lds);   // no staging, 1/8 of the LDS reads, all FMAs
            } else if constexpr (ABL == 3) {
                compute_tile<NT, S32IN>(c, tile, win0, h, lds);         // one v_fmac_f32 per FMA (A/B against the packed form)
            } else if constexpr (ABL != 1) {
                compute_tile_pk<NT, S32IN>(c, tile, win0, h, lds);
            } else {
                const f32x4 v0 = win0[0], v1 = win0[17];
                const long long m = (long long)tile * C::TILE_OUT + 8 * c.g + 4 * c.p;
                float *dst = c.out + 2 * m;
                if (m + 4 <= c.n_out) {
                    *reinterpret_cast<f32x4 *>(dst) = v0 + h[0];
                    *reinterpret_cast<f32x4 *>(dst + 4) = v1 + h[63 % C::TPL];
                }
            }
        }
        if constexpr (ABL == 11 || ABL == 12) {
            // in-kernel clock = shader cycles / (100 MHz ticks / 100 MHz); stamp values go only to a
            // buffer nothing else reads
            const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
            if (c.lane == 0 && a.stamps) {
                a.stamps[2 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x)] = c1 - st_c0;
                a.stamps[2 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x) + 1] = r1 - st_r0;
            }
        }
    } else {
        // vmcnt bookkeeping (in issue order): every stage_tile issues exactly NLOAD DMAs, and the
        // stores of tile-1 are older than the prefetch of tile+1.  After issuing that prefetch the
        // DMAs of `tile` have landed once at most NLOAD operations remain outstanding.
        static_assert(C::NLOAD == 10 || C::NLOAD == 6, "vmcnt immediates below assume NLOAD");
        const f32x4 *win1 = win0 + C::BUF_SLOTS;
        load_taps();
        stage_tile<NT>(c, tile_begin, lds);
        int tile = tile_begin;
        while (true) {
            // even phase: compute from buffer 0, prefetch into buffer 1
            if (tile + tile_step < tile_end) {
                stage_tile<NT>(c, tile + tile_step, lds + C::BUF_SLOTS);
                if constexpr (C::NLOAD == 10) SXFIR_WAIT_VMCNT(10); else SXFIR_WAIT_VMCNT(6);
            } else {
                SXFIR_WAIT_VMCNT(0);
            }
            compute_tile_pk<NT>(c, tile, win0, h, lds);
            if ((tile += tile_step) >= tile_end) break;
            // odd phase: compute from buffer 1, prefetch into buffer 0
            if (tile + tile_step < tile_end) {
                stage_tile<NT>(c, tile + tile_step, lds);
                if constexpr (C::NLOAD == 10) SXFIR_WAIT_VMCNT(10); else SXFIR_WAIT_VMCNT(6);
            } else {
                SXFIR_WAIT_VMCNT(0);
            }
            compute_tile_pk<NT>(c, tile, win1, h, lds + C::BUF_SLOTS);
            if ((tile += tile_step) >= tile_end) break;
        }
    }
}

}  // namespace sxfir
