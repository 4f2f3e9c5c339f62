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
//   HCARRY  a wave takes a CONTIGUOUS run of tiles and keeps the 128-sample halo in
//           LDS (one ds_read_b128 + ds_write_b128 per lane and tile) instead of
//           fetching it again: 11 % fewer bytes through the L2 -> LDS path.
//   PRIO    the wave raises its priority for the arithmetic of a tile.
//   CUQ     one workgroup of WPG = 16 waves per CU, each wave with its private image, takes the tiles of the
//           workgroup's share from a counter in LDS (ds_add_rtn, ~100 cycles) instead of a fixed tile list per
//           wave.  Measured (profiles/round2i_wave_lifetimes.txt): with a fixed list the four waves of a SIMD
//           finish 330 .. 520 us apart -- the SIMD's issue arbitration favours the older wave -- while every
//           CU's mean is within 3 %; the kernel then lasts as long as its slowest wave, or it is cut into 16
//           generations of short waves that each pay a prologue and a ~3 us hand-over.  With the queue the
//           waves of a CU stay busy until its share is done: one generation, one prologue per wave.  The
//           global tile order is the strided, XCD-blocked one (16 consecutive tiles per workgroup and pass).
//   SCALAR  symmetric filters (h[k] == h[NT-1-k] bit for bit, which every linear-phase design is): the 64
//           distinct taps live in SGPR pairs for the whole kernel and are the scalar operand of the packed
//           FMAs.  Measured (tools/valu_power_probe.hip): this kernel is bound by its FMAs at the clock the
//           chip's power management allows, and two VGPR operands per FMA instead of three let the same
//           arithmetic run at a 13 % higher clock.  A scalar operand is wave-uniform, so the work split
//           changes: every lane computes 4 consecutive outputs over ALL taps (window of 71 chunks read as
//           two streams 64 samples apart, 78 ds_read_b128; no tap halves on lane pairs, no cross-lane
//           reduction, 50 VGPRs fewer).  Same LDS image, same
//           numeric contract (P1 = chain over taps 127..64, P0 over 63..0, y = P0 + P1 inside one lane).
//
// New code: the reference decimates inside the SX1255 (SoapySX.cpp:180-208 only
// programs the divider); this kernel plays that role for SoapySX::readStream
// (SoapySX.cpp:868-967).
#pragma once

#include <utility>

#include "sxfir_decim_tile.hip.h"
#include "sxfir_common.hip.h"      // pk_fma_s_*, slot_source_offset, rgrp_table (shared with the shipped kernels)

namespace sxfir {

enum { T2_DEFER = 1, T2_TAPSEP = 2, T2_DBUF = 4, T2_PLAINST = 8, T2_HCARRY = 16, T2_PRIO = 32, T2_SCALAR = 64, T2_CUQ = 128,
       T2_MASKPAD = 256, T2_KARG = 512,
       // issue order of the 16 packed FMAs of a step (same chains, same bits): XGROUP = the four FMAs that share a
       // sample pair back to back (P1's four, then P0's four), XSTREAM = all eight of a window chunk back to back
       T2_XGROUP = 1024, T2_XSTREAM = 2048,
       // what the production library launches for 128 symmetric taps (T2_SHIPPED, below): scalar taps, the FMAs that
       // share a sample pair issued back to back (round 3: 1.3 % less time at the power cap than alternating the two
       // sample streams, profiles/round3j_kbench_fma_order.txt; the x operand of four consecutive FMAs does not
       // toggle) and, since round 4, non-temporal staging loads for everything but the two halos (NTLD7)
       // software-pipelined window reads (fir_tile_sym_pipe): the chunks of step T + PF are requested before the FMAs of
       // step T, PF = 2 (PIPE2), 3 (PIPE3 = both bits) or 4 (PIPE4); x-grouped issue order
       T2_PIPE2 = 4096, T2_PIPE4 = 8192,
       // tap-major issue order (fir_tile_sym_tapmajor): the four FMAs that share a TAP back to back -- the scalar operand,
       // broadcast to all 128 multipliers, then changes once per four FMAs and the sample operand with every FMA
       // (XGROUP is the opposite trade)
       T2_TAPMAJOR = 16384,
       // cache policy of the staging DMAs (round 4, tools/membench5.hip: a read stream runs 4.6 % faster with nt loads,
       // the decimator's 4:1 mix 6.5 %): NTLD = every DMA non-temporal; NTLD8 = DMAs 0..7 only -- the last kilobyte of
       // a tile (DMAs 8 and 9) is the next tile's halo and stays a plain load, so that it is still in the XCD's L2
       // when the neighbouring wave asks for it
       T2_NTLD = 32768, T2_NTLD8 = 65536,
       // XGROUP with the FMAs as volatile asm: the issue order is then the source order (the plain asm leaves the
       // machine scheduler free to interleave the quads: 60 % of adjacent FMAs share their sample pair in the
       // round-3 code object, 75 % when pinned)
       T2_PINNED = 131072,
       // NTLD9: DMAs 0..8 non-temporal, only the last (partial) instruction plain: most of the next tile's halo then
       // depends on nt lines staying in the L2 for the few microseconds until the neighbouring wave asks (A/B)
       T2_NTLD9 = 262144,
       // NTLD7: DMAs 1..7 non-temporal; instruction 0 -- the re-read of the previous tile's last kilobyte -- is a plain
       // load as well as 8 and 9.  With NTLD8 the counters show 5.4 % more HBM reads than bytes (traffic 1.043 x
       // algorithmic): when the halo's nt re-read reaches the L2 before the neighbour's plain load has, it streams
       // through without leaving the line there, and the plain load fetches it a second time.  Both plain: whichever
       // comes first allocates, the other hits.
       T2_NTLD7 = 524288,
       T2_SHIPPED = T2_SCALAR | T2_XGROUP | T2_NTLD7 };

// HBM -> LDS for the slots [Q0, Q0 + 64*(NI-1) + LASTL) of one tile's image: DMA instruction j fills the slots
// Q0 + 64j + lane from the per-lane byte offsets off[j] (tile-invariant, see slot_source_offset).  The last
// instruction is issued for LASTL lanes.
// lanes of the DMA instruction that starts at slot q0 whose slot is a pad slot (never read)
constexpr unsigned long long pad_lanes(int q0)
{
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l)
        if ((q0 + l + 1) % 17 == 0) m |= 1ull << l;
    return m;
}

// MASKPAD: the lanes whose slot is a pad slot sit the DMA out (6 % of the L2 -> LDS bytes)
// NTMASK: bit j set = DMA instruction j is a non-temporal load (aux = 2)
template <int NT, int Q0, int NI, int LASTL, bool MASKPAD = false, unsigned NTMASK = 0u>
__device__ __forceinline__ void stage_range(const DecimTileCtx<NT> &c, int tile, f32x4 *buf, const unsigned (&off)[NI])
{
    using C = DecimTile4<NT>;
    const long long c0 = ((long long)tile * C::TILE_IN - C::HALO) >> 1;
    constexpr int FIRSTCH = Q0 - (Q0 + 1) / 17;           // first chunk of the range
    const bool interior = (c0 + FIRSTCH >= 0) && (c0 + C::CHUNKS - 1 <= c.last_chunk - c.n_odd);
    if (interior) {
        const char *src = reinterpret_cast<const char *>(reinterpret_cast<const f32x4 *>(c.in) + c0);
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            unsigned b = off[j];
            asm volatile("" : "+v"(b));                  // 32-bit offset next to its use (see stage_tile)
            bool on = j < NI - 1 || LASTL >= 64 || c.lane < LASTL;
            if constexpr (MASKPAD) on = on && !((pad_lanes(Q0 + 64 * j) >> c.lane) & 1ull);
            if (on) {
                if ((NTMASK >> j) & 1u) glds16<2>(src + b, buf + Q0 + 64 * j);
                else glds16(src + b, buf + Q0 + 64 * j);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            unsigned b = off[j];
            asm volatile("" : "+v"(b));
            long long ch = c0 + (b >> 4);
            const f32x4 *src;
            if (ch < 0) {
                src = reinterpret_cast<const f32x4 *>(c.hist) + (ch + C::HIST / 2);
            } else {
                if (ch > c.last_chunk) ch = c.last_chunk;
                src = reinterpret_cast<const f32x4 *>(c.in) + ch;
            }
            if (j < NI - 1 || LASTL >= 64 || c.lane < LASTL) stage_edge_chunk(c, ch, src, buf + Q0 + 64 * j);
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

// FIR arithmetic of one tile for a symmetric 128-tap filter, taps hs[m] = {h[2m], h[2m+1]}, m < 32, in SGPRs.
// Lane l computes outputs 4l..4l+3 of the tile: output i meets tap k at window sample u = 4i + 128 - k
// (window = chunks 8l .. 8l+70 of the image); walking u upwards gives every output its taps in descending
// order, P1 (127..64) before P0 (63..0).  Tap k >= 64 is h[127-k].  w1 / w2: the lane's window base for
// steps whose pad count is that of an even / odd lane (the image has one pad slot after every 16 chunks
// and an odd lane's window starts in the middle of such a row).
// One step (T of 39) of fir_tile_sym: window chunk T feeds the P1 chains (taps 127..64), chunk T + 32 -- the
// samples 64 later -- the P0 chains (taps 63..0), so eight independent accumulators are in flight.  A
// function template per step instead of a loop: every tap index must be a compile-time constant for the
// taps to stay in SGPRs, and the loop form is too large for the unroller's budget (it then indexes the tap
// array dynamically, through scratch memory).
template <int WHICH, int S, int T>
__device__ __forceinline__ void fir_sym_quad(const f32x2 &x, const f32x2 (&hs)[32], f32x2 (&acc)[4])
{
    // the (up to) four FMAs of stream WHICH (1 = P1, 0 = P0) that use sample pair S of step T
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k0 = 4 * i + 64 - (2 * T + S);
        if (k0 >= 0 && k0 < 64) {
            const int k = WHICH ? 63 - k0 : k0;
            if (k & 1) pk_fma_s_hi(acc[i], hs[k >> 1], x);
            else pk_fma_s_lo(acc[i], hs[k >> 1], x);
        }
    }
}

template <int WHICH, int S, int T>
__device__ __forceinline__ void fir_sym_quad_v(const f32x2 &x, const f32x2 (&hs)[32], f32x2 (&acc)[4]);

template <bool S32IN, int T, int ORDER = 0, bool PINNED = false>
__device__ __forceinline__ void fir_sym_step(const f32x4 *w1, const f32x4 *w2, const f32x2 (&hs)[32], f32x2 (&a1)[4],
                                             f32x2 (&a0)[4])
{
    constexpr int T0 = T + 32;
    f32x4 v1 = ((T & 15) >= 8 ? w2 : w1)[T + (T >> 4)];
    f32x4 v0 = ((T0 & 15) >= 8 ? w2 : w1)[T0 + (T0 >> 4)];
    if constexpr (S32IN) {
        v1 = (f32x4){(float)__float_as_int(v1.x), (float)__float_as_int(v1.y), (float)__float_as_int(v1.z),
                     (float)__float_as_int(v1.w)};
        v0 = (f32x4){(float)__float_as_int(v0.x), (float)__float_as_int(v0.y), (float)__float_as_int(v0.z),
                     (float)__float_as_int(v0.w)};
    }
    if constexpr (ORDER != 0) {
        const f32x2 x1l = __builtin_shufflevector(v1, v1, 0, 1), x1h = __builtin_shufflevector(v1, v1, 2, 3);
        const f32x2 x0l = __builtin_shufflevector(v0, v0, 0, 1), x0h = __builtin_shufflevector(v0, v0, 2, 3);
        if constexpr (ORDER == 1 && PINNED) {
            fir_sym_quad_v<1, 0, T>(x1l, hs, a1);
            fir_sym_quad_v<0, 0, T>(x0l, hs, a0);
            fir_sym_quad_v<1, 1, T>(x1h, hs, a1);
            fir_sym_quad_v<0, 1, T>(x0h, hs, a0);
        } else if constexpr (ORDER == 1) {
            fir_sym_quad<1, 0, T>(x1l, hs, a1);
            fir_sym_quad<0, 0, T>(x0l, hs, a0);
            fir_sym_quad<1, 1, T>(x1h, hs, a1);
            fir_sym_quad<0, 1, T>(x0h, hs, a0);
        } else {
            fir_sym_quad<1, 0, T>(x1l, hs, a1);
            fir_sym_quad<1, 1, T>(x1h, hs, a1);
            fir_sym_quad<0, 0, T>(x0l, hs, a0);
            fir_sym_quad<0, 1, T>(x0h, hs, a0);
        }
        return;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int u = 2 * T + s;                       // P1 meets window sample u, P0 sample u + 64
        const f32x2 x1 = s ? __builtin_shufflevector(v1, v1, 2, 3) : __builtin_shufflevector(v1, v1, 0, 1);
        const f32x2 x0 = s ? __builtin_shufflevector(v0, v0, 2, 3) : __builtin_shufflevector(v0, v0, 0, 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k0 = 4 * i + 64 - u;             // P0's tap; P1's is k0 + 64 = h[127 - (k0 + 64)] = h[63 - k0]
            if (k0 >= 0 && k0 < 64) {
                const int k1 = 63 - k0;
                if (k1 & 1) pk_fma_s_hi(a1[i], hs[k1 >> 1], x1);
                else pk_fma_s_lo(a1[i], hs[k1 >> 1], x1);
                if (k0 & 1) pk_fma_s_hi(a0[i], hs[k0 >> 1], x0);
                else pk_fma_s_lo(a0[i], hs[k0 >> 1], x0);
            }
        }
    }
}

template <bool S32IN, int ORDER, bool PINNED, int... Ts>
__device__ __forceinline__ void fir_sym_steps(std::integer_sequence<int, Ts...>, const f32x4 *w1, const f32x4 *w2,
                                              const f32x2 (&hs)[32], f32x2 (&a1)[4], f32x2 (&a0)[4])
{
    (fir_sym_step<S32IN, Ts, ORDER, PINNED>(w1, w2, hs, a1, a0), ...);
}

template <bool S32IN, int ORDER = 0, bool PINNED = false>
__device__ __forceinline__ void fir_tile_sym(const f32x4 *w1, const f32x4 *w2, const f32x2 (&hs)[32], float (&oi)[4],
                                             float (&oq)[4])
{
    f32x2 a1[4], a0[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a1[i] = (f32x2){0.0f, 0.0f}; a0[i] = (f32x2){0.0f, 0.0f}; }
    fir_sym_steps<S32IN, ORDER, PINNED>(std::make_integer_sequence<int, 39>{}, w1, w2, hs, a1, a0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        oi[i] = __fadd_rn(a0[i].x, a1[i].x);
        oq[i] = __fadd_rn(a0[i].y, a1[i].y);
    }
}

// ---- the same arithmetic with the window reads software-pipelined.  Left to itself hipcc places every
// ds_read_b128 right before its first use and then waits for it with lgkmcnt(0): a wave exposes the whole LDS
// latency once per step, and only its neighbours on the SIMD hide it.  Here the reads are volatile loads and the
// FMAs volatile asm, which keeps them in program order: the chunks of step T + PF are requested before the FMAs
// of step T, and the compiler's own s_waitcnt becomes a counted lgkmcnt(2 * PF - ...) that finds the data there.
template <int WHICH, int S, int T>
__device__ __forceinline__ void fir_sym_quad_v(const f32x2 &x, const f32x2 (&hs)[32], f32x2 (&acc)[4])
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k0 = 4 * i + 64 - (2 * T + S);
        if (k0 >= 0 && k0 < 64) {
            const int k = WHICH ? 63 - k0 : k0;
            if (k & 1) pk_fma_sv_hi(acc[i], hs[k >> 1], x);
            else pk_fma_sv_lo(acc[i], hs[k >> 1], x);
        }
    }
}

template <int T>
__device__ __forceinline__ void fir_pipe_load(const f32x4 *w1, const f32x4 *w2, f32x4 (&q1)[39], f32x4 (&q0)[39])
{
    if constexpr (T < 39) {
        constexpr int T0 = T + 32;
        typedef const volatile __attribute__((address_space(3))) f32x4 *lds_vptr;   // volatile, and still an LDS pointer
        q1[T] = *(lds_vptr)(((T & 15) >= 8 ? w2 : w1) + (T + (T >> 4)));
        q0[T] = *(lds_vptr)(((T0 & 15) >= 8 ? w2 : w1) + (T0 + (T0 >> 4)));
    }
}

template <bool S32IN, int T>
__device__ __forceinline__ void fir_pipe_compute(const f32x4 (&q1)[39], const f32x4 (&q0)[39], const f32x2 (&hs)[32],
                                                 f32x2 (&a1)[4], f32x2 (&a0)[4])
{
    f32x4 v1 = q1[T], v0 = q0[T];
    if constexpr (S32IN) {
        v1 = (f32x4){(float)__float_as_int(v1.x), (float)__float_as_int(v1.y), (float)__float_as_int(v1.z),
                     (float)__float_as_int(v1.w)};
        v0 = (f32x4){(float)__float_as_int(v0.x), (float)__float_as_int(v0.y), (float)__float_as_int(v0.z),
                     (float)__float_as_int(v0.w)};
    }
    fir_sym_quad_v<1, 0, T>(__builtin_shufflevector(v1, v1, 0, 1), hs, a1);
    fir_sym_quad_v<0, 0, T>(__builtin_shufflevector(v0, v0, 0, 1), hs, a0);
    fir_sym_quad_v<1, 1, T>(__builtin_shufflevector(v1, v1, 2, 3), hs, a1);
    fir_sym_quad_v<0, 1, T>(__builtin_shufflevector(v0, v0, 2, 3), hs, a0);
}

template <bool S32IN, int PF, int... Ps, int... Ts>
__device__ __forceinline__ void fir_pipe_steps(std::integer_sequence<int, Ps...>, std::integer_sequence<int, Ts...>,
                                               const f32x4 *w1, const f32x4 *w2, const f32x2 (&hs)[32], f32x2 (&a1)[4],
                                               f32x2 (&a0)[4])
{
    f32x4 q1[39], q0[39];
    (fir_pipe_load<Ps>(w1, w2, q1, q0), ...);                                   // steps 0 .. PF-1
    ((fir_pipe_load<Ts + PF>(w1, w2, q1, q0), fir_pipe_compute<S32IN, Ts>(q1, q0, hs, a1, a0)), ...);
}

template <bool S32IN, int PF>
__device__ __forceinline__ void fir_tile_sym_pipe(const f32x4 *w1, const f32x4 *w2, const f32x2 (&hs)[32], float (&oi)[4],
                                                  float (&oq)[4])
{
    f32x2 a1[4], a0[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a1[i] = (f32x2){0.0f, 0.0f}; a0[i] = (f32x2){0.0f, 0.0f}; }
    fir_pipe_steps<S32IN, PF>(std::make_integer_sequence<int, PF>{}, std::make_integer_sequence<int, 39>{}, w1, w2, hs, a1, a0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        oi[i] = __fadd_rn(a0[i].x, a1[i].x);
        oq[i] = __fadd_rn(a0[i].y, a1[i].y);
    }
}

// ---- tap-major order: for tap index k0 = 63 .. 0, P1's tap h[63 - k0] meets outputs 0..3 at window samples
// u = 4i + 64 - k0 (chunks two apart), then P0's tap h[k0] meets the samples 64 later.  Every accumulator still sees
// its taps in descending order, so the bits are those of the other orders.  Eight chunks per stream are live.
template <bool S32IN, int T>
__device__ __forceinline__ void fir_tm_load(const f32x4 *w1, const f32x4 *w2, f32x4 (&q1)[39], f32x4 (&q0)[39])
{
    if constexpr (T >= 0 && T < 39) {
        constexpr int T0 = T + 32;
        f32x4 v1 = ((T & 15) >= 8 ? w2 : w1)[T + (T >> 4)];
        f32x4 v0 = ((T0 & 15) >= 8 ? w2 : w1)[T0 + (T0 >> 4)];
        if constexpr (S32IN) {
            v1 = (f32x4){(float)__float_as_int(v1.x), (float)__float_as_int(v1.y), (float)__float_as_int(v1.z), (float)__float_as_int(v1.w)};
            v0 = (f32x4){(float)__float_as_int(v0.x), (float)__float_as_int(v0.y), (float)__float_as_int(v0.z), (float)__float_as_int(v0.w)};
        }
        q1[T] = v1;
        q0[T] = v0;
    }
}

template <bool S32IN, int STEP>
__device__ __forceinline__ void fir_tm_tap(const f32x4 *w1, const f32x4 *w2, f32x4 (&q1)[39], f32x4 (&q0)[39],
                                           const f32x2 (&hs)[32], f32x2 (&a1)[4], f32x2 (&a0)[4])
{
    constexpr int K0 = 63 - STEP;                        // STEP 0 .. 63  <->  k0 63 .. 0
    constexpr int K1 = 63 - K0;
    // the chunk output 3 reaches with this tap, the first time it is reached (every second tap)
    constexpr int UTOP = 4 * 3 + 64 - K0;
    if constexpr (STEP == 0) {
        fir_tm_load<S32IN, 0>(w1, w2, q1, q0); fir_tm_load<S32IN, 2>(w1, w2, q1, q0);
        fir_tm_load<S32IN, 4>(w1, w2, q1, q0); fir_tm_load<S32IN, 6>(w1, w2, q1, q0);
    } else if constexpr (STEP == 1) {
        fir_tm_load<S32IN, 1>(w1, w2, q1, q0); fir_tm_load<S32IN, 3>(w1, w2, q1, q0);
        fir_tm_load<S32IN, 5>(w1, w2, q1, q0); fir_tm_load<S32IN, 7>(w1, w2, q1, q0);
    } else if constexpr ((UTOP & 1) == 0) {
        fir_tm_load<S32IN, (UTOP >> 1)>(w1, w2, q1, q0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int u = 4 * i + 64 - K0;
        const f32x4 v = q1[u >> 1];
        const f32x2 x = (u & 1) ? __builtin_shufflevector(v, v, 2, 3) : __builtin_shufflevector(v, v, 0, 1);
        if (K1 & 1) pk_fma_sv_hi(a1[i], hs[K1 >> 1], x);
        else pk_fma_sv_lo(a1[i], hs[K1 >> 1], x);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int u = 4 * i + 64 - K0;
        const f32x4 v = q0[u >> 1];
        const f32x2 x = (u & 1) ? __builtin_shufflevector(v, v, 2, 3) : __builtin_shufflevector(v, v, 0, 1);
        if (K0 & 1) pk_fma_sv_hi(a0[i], hs[K0 >> 1], x);
        else pk_fma_sv_lo(a0[i], hs[K0 >> 1], x);
    }
}

template <bool S32IN, int... Ss>
__device__ __forceinline__ void fir_tm_steps(std::integer_sequence<int, Ss...>, const f32x4 *w1, const f32x4 *w2,
                                             const f32x2 (&hs)[32], f32x2 (&a1)[4], f32x2 (&a0)[4])
{
    f32x4 q1[39], q0[39];
    (fir_tm_tap<S32IN, Ss>(w1, w2, q1, q0, hs, a1, a0), ...);
}

template <bool S32IN>
__device__ __forceinline__ void fir_tile_sym_tapmajor(const f32x4 *w1, const f32x4 *w2, const f32x2 (&hs)[32], float (&oi)[4],
                                                      float (&oq)[4])
{
    f32x2 a1[4], a0[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a1[i] = (f32x2){0.0f, 0.0f}; a0[i] = (f32x2){0.0f, 0.0f}; }
    fir_tm_steps<S32IN>(std::make_integer_sequence<int, 64>{}, w1, w2, hs, a1, a0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        oi[i] = __fadd_rn(a0[i].x, a1[i].x);
        oq[i] = __fadd_rn(a0[i].y, a1[i].y);
    }
}

template <bool PLAIN>
__device__ __forceinline__ void st16(const f32x4 &v, f32x4 *dst)
{
    if constexpr (PLAIN) *dst = v;
    else __builtin_nontemporal_store(v, dst);
}

// ABL (profiling): 0 = the real kernel, 1 = staging + stores without the FIR (memory side alone), 2 = FIR on
// whatever LDS holds (no staging), 5 = the real kernel with s_memtime stamps around its phases (a.stamps:
// per wave 8 x uint64 {tiles, cycles issuing DMAs, waiting for data, FIR arithmetic, output transposition +
// stores, cycles from the wave's first to its last instruction, the same span in 100 MHz ticks, XCC_ID | HW_ID << 8}).
template <int NT, int WPG, int OPT, int ABL = 0, bool S32IN = false>
__global__ __launch_bounds__(64 * WPG) void decim4_tile2_kernel(const DecimTileArgs a)
{
    using C = DecimTile4<NT>;
    constexpr bool DEFER = (OPT & T2_DEFER) != 0, TAPSEP = (OPT & T2_TAPSEP) != 0, DBUF = (OPT & T2_DBUF) != 0;
    constexpr bool PLAINST = (OPT & T2_PLAINST) != 0, HCARRY = (OPT & T2_HCARRY) != 0, PRIO = (OPT & T2_PRIO) != 0;
    constexpr bool SCALAR = (OPT & T2_SCALAR) != 0, CUQ = (OPT & T2_CUQ) != 0, KARG = (OPT & T2_KARG) != 0;
    static_assert(!KARG || (SCALAR && WPG == 1 && !CUQ && !HCARRY), "taps by value: the plain scalar-tap kernel");
    static_assert(!CUQ || (!DBUF && !HCARRY && !TAPSEP && (WPG & (WPG - 1)) == 0), "the LDS tile queue drives the single-buffered loop");
    __shared__ unsigned cuq_next;                        // CUQ: tiles of this workgroup's share handed out so far
    static_assert(!(HCARRY && DBUF), "halo carry-over is for the single-buffered loop");
    static_assert(!SCALAR || (NT == 128 && !TAPSEP), "scalar taps: 128-tap symmetric filters, no tap staging");
    constexpr int IMG = C::SLOTS;                       // slots per tile image
    constexpr int NB = DBUF ? 2 : 1;
    __shared__ __attribute__((aligned(16))) f32x4 lds[WPG * NB * IMG + (TAPSEP ? NT / 4 : 0)];

    unsigned long long wave_c0 = 0, wave_r0 = 0;
    if constexpr (ABL == 5) {
        wave_c0 = __builtin_amdgcn_s_memtime();
        wave_r0 = __builtin_amdgcn_s_memrealtime();
    }
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

    // Image split for staging: the halo (slots [0, HS): the HALO/2 chunks before the tile) and the body.
    // Without HCARRY both are staged for every tile as one range.
    constexpr int HS = C::HALO / 2 + C::HALO / 32;       // halo slots (64 chunks + 4 pads at NT = 128)
    constexpr int BODY = IMG - HS;
    constexpr int NIB = (BODY + 63) / 64, LASTB = BODY - 64 * (NIB - 1);
    constexpr int NIA = (HS + 63) / 64, LASTA = HS - 64 * (NIA - 1);
    constexpr int NIF = C::NLOAD, LASTF = IMG - 64 * (C::NLOAD - 1);
    unsigned boff[HCARRY ? NIB : NIF];
#pragma unroll
    for (int j = 0; j < (HCARRY ? NIB : NIF); ++j)
        boff[j] = slot_source_offset((HCARRY ? HS : 0) + 64u * j + c.lane, C::CHUNKS);

    auto stage_full = [&](int t, f32x4 *buf) __attribute__((always_inline)) {
        if constexpr (ABL == 2) return;
        if constexpr (HCARRY) {
            unsigned aoff[NIA];
#pragma unroll
            for (int j = 0; j < NIA; ++j) aoff[j] = slot_source_offset(64u * j + c.lane, C::CHUNKS);
            stage_range<NT, 0, NIA, LASTA, false, (OPT & T2_NTLD) ? 0xffffffffu : 0u>(c, t, buf, aoff);
            stage_range<NT, HS, NIB, LASTB, false, (OPT & T2_NTLD) ? 0xffffffffu : 0u>(c, t, buf, boff);
        } else {
            constexpr unsigned NTM = (OPT & T2_NTLD) ? 0xffffffffu : ((OPT & T2_NTLD9) ? 0x1ffu : ((OPT & T2_NTLD8) ? 0xffu : ((OPT & T2_NTLD7) ? 0xfeu : 0u)));
            stage_range<NT, 0, NIF, LASTF, (OPT & T2_MASKPAD) != 0, NTM>(c, t, buf, boff);
        }
    };

    // Tile schedule.  Strided (default): in pass i the G workgroups of a channel cover the G*WPG consecutive
    // tiles [i*G*WPG, (i+1)*G*WPG); wave ww of workgroup b takes tile (S(b) + i*G)*WPG + ww, where S deals the
    // workgroups of one XCD (blockIdx % 8 shares an XCD; speed only) a contiguous block of the pass, so that
    // halo re-reads stay in that XCD's L2; a.sched == 2: S(b) = b (plain dispatch order).
    // HCARRY: wave number b*WPG + ww takes the contiguous run of a.run_base tiles starting at its number
    // times a.run_base.
    const int G = a.n_waves;                            // workgroups per channel
    const int b = blockIdx.x;
    int tile, tile_end, tile_step;
    if constexpr (HCARRY) {
        tile = (b * WPG + ww) * a.run_base;
        tile_end = tile + a.run_base < a.n_tiles ? tile + a.run_base : a.n_tiles;
        tile_step = 1;
    } else {
        if (WPG == 1 && a.long_waves > 0) {
            // Short tail: the launch ends with one-tile waves, so that the CUs run empty over the life of a short
            // wave instead of a long one; the long waves stride over the tiles before them.
            if (b < a.long_waves) {
                const int S = a.long_w8 ? (b & 7) * a.long_w8 + (b >> 3) : b;
                tile = S;
                tile_end = a.long_tiles;
                tile_step = a.long_waves;
            } else {
                const int bs = b - a.long_waves;
                tile = a.long_tiles + (a.short_w8 ? (bs & 7) * a.short_w8 + (bs >> 3) : bs);
                tile_end = tile + 1 < a.n_tiles ? tile + 1 : a.n_tiles;
                tile_step = 1;
            }
        } else {
            const int S = (a.sched == 0 && a.w8) ? (b & 7) * a.w8 + (b >> 3) : b;
            tile = S * WPG + ww;
            tile_end = a.n_tiles;
            tile_step = G * WPG;
        }
    }
    if constexpr (CUQ) {
        if (threadIdx.x == 0) cuq_next = 0u;
        __syncthreads();
    } else {
        // (KARG: the host launches no more waves than tiles, so no wave leaves here -- and nothing stands between
        // the scalar loads of the arguments, which then go out as one batch)
        if constexpr (KARG) __builtin_assume(tile < tile_end);
        else if (tile >= tile_end) return;
        if (b * WPG + ww == a.hist_wave) write_history<NT>(c, a.hist_out + 2 * a.hist_stride * ch, a.n_in);
    }

    // taps: this lane's half as 64-bit VGPR pairs, fetched once through an LDS broadcast -- or, for a
    // symmetric filter, the 64 distinct taps as SGPR pairs (scalar loads from the constant address space;
    // an S32 plan passes taps already scaled by 2^-31 here)
    f32x2 hp[SCALAR ? 1 : C::TPL / 2];
    f32x2 hs[32];
    auto read_taps = [&]() __attribute__((always_inline)) {
        if constexpr (!SCALAR) {
            const f32x4 *tp = tapbuf + (C::TPL / 4) * c.p;
#pragma unroll
            for (int k = 0; k < C::TPL / 4; ++k) {
                f32x4 t = tp[k];
                if constexpr (S32IN) t = t * 4.656612873077393e-10f;
                hp[2 * k] = (f32x2){t.x, t.y};
                hp[2 * k + 1] = (f32x2){t.z, t.w};
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    };
    if constexpr (SCALAR && KARG) {
#pragma unroll
        for (int m = 0; m < 32; ++m) hs[m] = (f32x2){a.taps_k[2 * m], a.taps_k[2 * m + 1]};
        hp[0] = (f32x2){0.0f, 0.0f};
    } else if constexpr (SCALAR) {
        const __attribute__((address_space(4))) f32x2 *tq =
            (const __attribute__((address_space(4))) f32x2 *)(S32IN ? a.taps_scaled : a.taps);
#pragma unroll
        for (int m = 0; m < 32; ++m) hs[m] = tq[m];
        hp[0] = (f32x2){0.0f, 0.0f};
    } else {
#pragma unroll
        for (int m = 0; m < 32; ++m) hs[m] = (f32x2){0.0f, 0.0f};
        if (c.lane < NT / 4) glds16(reinterpret_cast<const char *>(a.taps) + 16 * c.lane, tapbuf);
        if constexpr (!TAPSEP) {
            // through the (still empty) tile image: the reads must have returned before the first tile's DMA
            SXFIR_WAIT_VMCNT(0);
            read_taps();
        }
    }

    // Scalar taps: lane -> output group r (outputs 4r..4r+3 of the tile, window from chunk 8r).  A
    // ds_read_b128 is served 16 lanes at a time, in the groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32), and
    // is conflict free when those lanes hit 16 different 16-byte slots mod 16.  With one pad slot per 16 chunks
    // an even r = 2m sits at slot residue m + const and an odd r = 2m+1 at m + 8 or m + 9 depending on the
    // step, so a group must not mix the two parities: the first group of each half-wave takes the even r,
    // the second the odd r, m = 0..15 in both (SQ_LDS_BANK_CONFLICT: 160 -> 32 cycles per tile and wave).
    int rgrp = c.lane;
    if constexpr (SCALAR) {
        // r of lane l5 (5 bits per lane, branch free): first group {0-3,12-15,20-27} -> 0,2,..,30 in lane order,
        // second group {4-11,16-19,28-31} -> 1,3,..,31
        const int l5 = c.lane & 31;
        rgrp = (int)((rgrp_table(l5 >> 3) >> (5 * (l5 & 7))) & 31u) + (c.lane & 32);
    }
    // this lane's first window chunk: tap halves on lane pairs -> 16g + (1 - p) * NT/4 (a multiple of 16);
    // scalar taps -> 8r, with a second base one slot on for an odd r's differently placed pads
    const int u0c = SCALAR ? 8 * rgrp : 16 * c.g - (NT / 4) * c.p + NT / 4;
    const int woff = u0c + (u0c >> 4);
    const int woff2 = woff + (rgrp & 1);
    // output transposition buffer inside the (dead) image; with HCARRY it must leave the carried halo alone
    constexpr int XB = HCARRY ? HS + 12 : 0;

    unsigned long long ph[5] = {0, 0, 0, 0, 0}, tk = 0;
    if constexpr (ABL == 5) tk = __builtin_amdgcn_s_memtime();
#define SXFIR_T2_PHASE(k) \
    if constexpr (ABL == 5) { \
        const unsigned long long t_now = __builtin_amdgcn_s_memtime(); \
        ph[k] += t_now - tk; \
        tk = t_now; \
    }

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
            for (int i = 0; i < 4; ++i) { oi[i] = v0[i] + hp[0].x + hs[3].y; oq[i] = v1[i] + hp[SCALAR ? 0 : 31 % (C::TPL / 2)].y; }
        } else {
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(2);
            constexpr int PF = ((OPT & T2_PIPE2) ? 2 : 0) + ((OPT & T2_PIPE4) ? ((OPT & T2_PIPE2) ? 1 : 4) : 0);   // 0, 2, 3, 4
            if constexpr (SCALAR && (OPT & T2_TAPMAJOR) != 0) fir_tile_sym_tapmajor<S32IN>(buf + woff, buf + woff2, hs, oi, oq);
            else if constexpr (SCALAR && PF > 0) fir_tile_sym_pipe<S32IN, PF>(buf + woff, buf + woff2, hs, oi, oq);
            else if constexpr (SCALAR) fir_tile_sym<S32IN, (OPT & T2_XGROUP) ? 1 : ((OPT & T2_XSTREAM) ? 2 : 0), (OPT & T2_PINNED) != 0>(buf + woff, buf + woff2, hs, oi, oq);
            else fir_tile_pk<NT, S32IN>(buf + woff, hp, oi, oq);
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
        }
        if constexpr (ABL == 5) asm volatile("" ::"v"(oi[0]), "v"(oq[3]));   // the arithmetic ends here
        SXFIR_T2_PHASE(3)
        if (m0 + C::TILE_OUT <= c.n_out) {
            // through the now dead image: chunk 4g + 2p + {0,1} of the tile's 128 output chunks, read back
            // linearly, so that each global store instruction writes 1 KiB of consecutive addresses
            const int oc = SCALAR ? 2 * rgrp : 4 * c.g + 2 * c.p;
            f32x4 *xb = buf + XB;
            xb[oc + (oc >> 4)] = (f32x4){oi[0], oq[0], oi[1], oq[1]};
            xb[oc + 1 + (oc >> 4)] = (f32x4){oi[2], oq[2], oi[3], oq[3]};
            const f32x4 v0 = xb[c.lane + (c.lane >> 4)], v1 = xb[68 + c.lane + (c.lane >> 4)];
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
            const long long m = m0 + (SCALAR ? 4 * rgrp : 8 * c.g + 4 * c.p);
            float *dst = c.out + 2 * m;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (m + i < c.n_out) { dst[2 * i] = oi[i]; dst[2 * i + 1] = oq[i]; }
        }
        // the image may be overwritten by the next DMA only after these LDS reads have returned
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };

    int ntile = 0;
    if constexpr (CUQ) {
        // k-th grab of this workgroup: pass k / WPG, tile k % WPG of the workgroup's WPG consecutive tiles
        const int S0 = (a.sched == 0 && a.w8) ? (b & 7) * a.w8 + (b >> 3) : b;
        auto grab = [&]() __attribute__((always_inline)) -> int {
            unsigned k = 0;
            if (c.lane == 0) k = __hip_atomic_fetch_add(&cuq_next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            k = (unsigned)__builtin_amdgcn_readfirstlane((int)k);
            const long long t = ((long long)S0 + (long long)(k / WPG) * G) * WPG + (k % WPG);
            return t < a.n_tiles ? (int)t : -1;          // monotone in k: the first miss ends the wave
        };
        int t = grab();
        while (t >= 0) {
            // the wave that gets the channel's last tile carries the history over
            if (t == a.n_tiles - 1) write_history<NT>(c, a.hist_out + 2 * a.hist_stride * ch, a.n_in);
            stage_full(t, img);
            const int tn = grab();
            SXFIR_T2_PHASE(1)
            SXFIR_WAIT_VMCNT(0);
            SXFIR_T2_PHASE(2)
            if constexpr (DEFER) flush();
            process(t, img);
            ++ntile;
            SXFIR_T2_PHASE(4)
            t = tn;
        }
        if constexpr (DEFER) flush();
    } else if constexpr (!DBUF) {
        stage_full(tile, img);
        SXFIR_T2_PHASE(1)
        SXFIR_WAIT_VMCNT(0);
        if constexpr (TAPSEP) read_taps();
        SXFIR_T2_PHASE(2)
        while (true) {
            if constexpr (DEFER) flush();
            process(tile, img);
            ++ntile;
            SXFIR_T2_PHASE(4)
            tile += tile_step;
            if (tile >= tile_end) break;
            if constexpr (HCARRY) {
                // the last HALO samples of this tile are the next tile's halo: chunk CHUNKS - HALO/2 + l -> chunk l
                constexpr int SRC0 = C::CHUNKS - C::HALO / 2;
                if (HS <= 64 || c.lane < C::HALO / 2) {
                    const int sc = SRC0 + c.lane;
                    const f32x4 v = img[sc + (sc >> 4)];
                    img[c.lane + (c.lane >> 4)] = v;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if constexpr (ABL != 2) stage_range<NT, HS, NIB, LASTB, false, (OPT & T2_NTLD) ? 0xffffffffu : 0u>(c, tile, img, boff);
            } else {
                stage_full(tile, img);
            }
            SXFIR_T2_PHASE(1)
            // LDS-DMA completion is ordered for this wave's ds_reads only by its own vmcnt
            SXFIR_WAIT_VMCNT(0);
            SXFIR_T2_PHASE(2)
        }
        if constexpr (DEFER) flush();
    } else {
        // vmcnt bookkeeping, oldest first, at the wait of iteration k:
        //   DEFER:  DMA(k) | stores(k-2) | DMA(k+1)      -> vmcnt(NLOAD) retires DMA(k) and stores that
        //           have had the whole arithmetic phase of tile k-1 to complete
        //   else:   DMA(k) | stores(k-1) | DMA(k+1)      -> vmcnt(NLOAD) also waits for fresh stores
        static_assert(C::NLOAD == 10, "vmcnt immediate below assumes NLOAD == 10");
        stage_full(tile, img);
        int cur = 0;
        bool first = true;
        while (true) {
            const int next = tile + tile_step;
            if (next < tile_end) {
                stage_full(next, img + (cur ^ 1) * IMG);
                SXFIR_T2_PHASE(1)
                SXFIR_WAIT_VMCNT(10);
            } else {
                SXFIR_WAIT_VMCNT(0);
            }
            if constexpr (TAPSEP) {
                if (first) { read_taps(); first = false; }
            }
            SXFIR_T2_PHASE(2)
            if constexpr (DEFER) flush();
            process(tile, img + cur * IMG);
            ++ntile;
            SXFIR_T2_PHASE(4)
            if (next >= tile_end) break;
            tile = next;
            cur ^= 1;
        }
        if constexpr (DEFER) flush();
    }
    if constexpr (ABL == 5) {
        SXFIR_T2_PHASE(4)
        ph[0] = (unsigned long long)ntile;
        const unsigned long long wave_c1 = __builtin_amdgcn_s_memtime(), wave_r1 = __builtin_amdgcn_s_memrealtime();
        if (c.lane == 0 && a.stamps) {
            unsigned long long *rec = a.stamps + 8 * ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * WPG + ww);
#pragma unroll
            for (int k = 0; k < 5; ++k) rec[k] = ph[k];
            rec[5] = wave_c1 - wave_c0;
            rec[6] = wave_r1 - wave_r0;
            // where the wave ran: XCC_ID in bits 0-3, HW_ID (wave, SIMD, CU, SH, SE ...) from bit 8 up
            unsigned xcc, hwid;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            rec[7] = (unsigned long long)(xcc & 15u) | ((unsigned long long)hwid << 8);
        }
    }
#undef SXFIR_T2_PHASE
}

}  // namespace sxfir
