// LDS-tiled polyphase FIR interpolator for gfx950: interpolate-by-L, L = 4*NPH
// (4, 8, 16, 32), 32 taps per phase (NT = 32*L): the TX half of BASELINE
// config 3 (256 taps, x8).  New code: in the reference the SX1255 interpolates
// what snd_pcm_writei hands it (SoapySX.cpp:1093); here it is
//     y[q*L + r] = sum_j h[j*L + r] * x[q - j].
//
// One wave = one workgroup; a tile is KT sub-tiles of QT = 128/NPH input samples
// (512 outputs each), staged by one LDS-DMA round trip and computed one after
// the other.  Lane = (p, g, c): tap-row half p (lane bit 5: j in
// [16p, 16p+16)), input group g (4 consecutive inputs), phase group c (phases
// 4c..4c+3) in the LOWEST lane bits: the lanes of one g read the same window, so
// each 16-lane service group of a ds_read_b128 touches few distinct addresses,
// all in different banks (no conflicts from x8 up; tools/lds_bank_model.py).  A lane holds its 64 taps in VGPRs and 32
// accumulators (4 inputs x 4 phases x I/Q); its window is 10 ds_read_b128
// (the tile is tiny: <= 2.3 KiB of LDS, staged by LDS-DMA) for 256 v_pk_fma_f32.
// The two row-half partials are exchanged with v_permlane32_swap and added
// once: the low lane keeps inputs 0-1, the high lane inputs 2-3, so each lane
// stores 2 x 4 consecutive outputs.
//
// Numeric contract: partial_p = fmaf chain from +0.0f over j
// DESCENDING in [16p, 16p+16); y = partial_0 + partial_1.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sxfir_decim_tile.hip.h"
#include "sxfir_kernels.hip.h"
#include "sxfir_decim_dense.hip.h"       // CF16 storage: v4i32, half_lo_to_float / half_hi_to_float, pack_half2

namespace sxfir {

struct InterpTileArgs {
    const float *in;        // channel 0, sample 0 of this call (8-byte aligned)
    const float *hist;      // 32 samples preceding `in`
    float *hist_out;
    float *out;             // 16-byte aligned
    const float *taps;
    long long n_in;         // input samples per channel (outputs = n_in * L)
    long long in_stride, out_stride, hist_stride;
    int n_tiles, n_groups;
    float thr2;             // S32 output / KEYED: transmitter-keying threshold (squared magnitude)
    // KEYED: the input samples [key_lo, key_hi) of channel 0 (indices relative to `in`) that reach thr2 are counted
    // into *key_counter (convert_tx_buffer's keying rule, SoapySX.cpp:132-133), one atomic per wave and tile
    unsigned long long *key_counter;
    long long key_lo, key_hi;
};

template <int L>
struct InterpTile {
    static constexpr int NT = 32 * L;
    static constexpr int NPH = L / 4;                     // phase groups
    static constexpr int GW = 32 / NPH;                   // input groups per wave
    static constexpr int QT = 4 * GW;                     // input samples per sub-tile
    static constexpr int KT = 4;                          // sub-tiles per staged tile
    static constexpr int TILE_IN = KT * QT;
    static constexpr int HIST = 32;
    static constexpr int CHUNKS = (TILE_IN + 32) / 2;     // staged: samples [q0 - 32, q0 + TILE_IN)
    static constexpr int NLOAD = (CHUNKS + 63) / 64;
    static constexpr int CB = NPH == 8 ? 3 : (NPH == 4 ? 2 : (NPH == 2 ? 1 : 0));   // lane bits of c
    static_assert(L % 4 == 0 && (NPH & (NPH - 1)) == 0 && NPH <= 8, "L must be 4, 8, 16 or 32");
    // Output transposition buffer: 256 chunks (512 outputs), chunk oc kept at slot oc ^ (some of its own higher
    // bits), chosen so that the 8 consecutive lanes a ds_write_b128 is served with hit 8 different slots mod 8
    // and the linear read-back stays conflict free: no pad slots.  Chunk index bits: bit 0 = which of a lane's
    // two pieces, then c, the input of the pair, p, g.
    static __device__ __forceinline__ int swz(int oc)
    {
        if (L == 4) return oc ^ ((oc >> 3) & 7);
        if (L == 8) return oc ^ (((oc >> 4) & 1) | (((oc >> 5) & 1) << 2));
        if (L == 16) return oc ^ ((oc >> 5) & 1);
        return oc ^ ((oc >> 3) & 1);
    }
};

// S32OUT: outputs leave as S32_LE I2S wire words with the keying bits (convert_tx_buffer, SoapySX.cpp:116-137)
// KEYED: the transmitter-keying count of the call's input is taken from the staged tile (every input sample is in
// LDS exactly once as a tile's own sample), so a pass that reads its input over PCIe reads it once, not twice
// LT (round 5): the plan's ratio when it is a multiple of L -- x48 and x96 (the reference's rates master clock / 768 and / 1536,
// SoapySX.cpp:180-208) as three PHASE BLOCKS of the x16 and of the x32 kernel.  An interpolator's phases never meet, so block
// pb -- phases [L pb, L pb + L) -- is the xL problem with the taps h[j LT + L pb + r] and outputs that lie LT apart per input:
// an input's L outputs of the block are one (x16) or two (x32) whole 128-byte lines, so every store instruction still writes
// whole lines.  The phase block is the fastest-varying part of the workgroup index (blocks of one tile are dispatched
// together and complete each other's DRAM pages); the tiny input tile is staged by each of them.  (x96 as six blocks of the
// x16 kernel: 5 % slower, profiles/round5_rates.txt; phase blocks of eight on the scalar-tap pass kernel -- half-line stores --
// 20-30 % slower.)
// HALF (round 5): CF16 storage on both sides (IQ as half pairs in HBM, fp32 arithmetic, outputs rounded to half once).  The
// image in LDS is the same CF32 image: interior tiles are staged by typed LDS-DMA (buffer_load_format_x ... lds with a
// {16, FLOAT} descriptor: 32 samples per instruction, the texture path converts on the way in; sxfir_decim_dense.hip.h),
// edge tiles convert in registers; a store instruction writes 512 consecutive bytes (x48: an input's sixteen outputs of a
// phase block are 64 bytes there -- half lines).
template <int L, bool S32OUT = false, bool KEYED = false, int LT = L, bool HALF = false>
__global__ __launch_bounds__(64) void interp_tile_kernel(const InterpTileArgs a)
{
    using C = InterpTile<L>;
    static_assert(!HALF || (!S32OUT && !KEYED), "CF16 storage: CF16 out, no keying count (the rule is defined on CF32 input)");
    static_assert(!HALF || (C::TILE_IN + 32) % 32 == 0, "whole typed instructions per tile");
    constexpr int SF = HALF ? 1 : 2;                    // 32-bit words per complex sample in HBM
    static_assert(LT % L == 0 && (LT == L || L == 16 || L == 32), "phase blocks of the x16 / x32 kernel");
    constexpr int NPB = LT / L;
    const int pb = NPB == 1 ? 0 : (int)(blockIdx.x % NPB);
    const int vb = NPB == 1 ? (int)blockIdx.x : (int)(blockIdx.x / NPB);          // workgroup index among a block's
    // input image, then the 512-output (4 KiB) transposition buffer
    __shared__ __attribute__((aligned(16))) f32x4 lds[C::NLOAD * 64 + 256];
    f32x4 *obuf = lds + C::NLOAD * 64;

    const int lane = threadIdx.x;
    const int p = lane >> 5;
    const int c = lane & (C::NPH - 1);                     // phase group: the lowest lane bits
    const int g = (lane >> C::CB) & (C::GW - 1);
    const int ch = blockIdx.y;

    const float *in = a.in + SF * a.in_stride * ch;
    const float *hist = a.hist + SF * a.hist_stride * ch;
    float *out = a.out + SF * a.out_stride * ch;
    // one sample of the stream as CF32, whatever the storage
    auto sample = [&](const float *base, long long s) __attribute__((always_inline)) {
        if constexpr (HALF) {
            const unsigned w = reinterpret_cast<const unsigned *>(base)[s];
            return make_float2(half_lo_to_float(w), half_hi_to_float(w));
        } else {
            return reinterpret_cast<const float2 *>(base)[s];
        }
    };

    // lane taps: h[4*jj + rr] = taps[(16p + jj)*L + 4c + rr], held as pairs hp[k] = {h[2k], h[2k+1]}
    f32x2 hp[32];          // 64-bit register pairs for the packed FMAs
#pragma unroll
    for (int k = 0; k < 32; ++k)
        hp[k] = (f32x2){a.taps[(16 * p + (k >> 1)) * LT + L * pb + 4 * c + 2 * (k & 1)],
                        a.taps[(16 * p + (k >> 1)) * LT + L * pb + 4 * c + 2 * (k & 1) + 1]};

    // window: samples q0 + 4g - 16p - 16 + w, w = 0..19  ->  LDS sample index (+32) 4g - 16p + 16 + w
    const f32x4 *win = lds + (2 * g - 8 * p + 8);

    // Tiles are dealt XCD-blocked, as in interp8_pass_kernel: in pass i the G workgroups cover tiles [iG, (i+1)G), the
    // workgroups of one XCD (blockIdx % 8; speed only) a contiguous block of them, so that a tile's 32-sample history --
    // its neighbour's tail -- is found in that XCD's L2 (dealt round robin the re-reads went to HBM: 1.014 x the bytes)
    const int NGR = a.n_groups;
    const int first_tile = (NGR % 8 == 0) ? (vb % 8) * (NGR / 8) + vb / 8 : vb;
    if (first_tile == (a.n_tiles - 1) % NGR && lane < C::HIST && pb == 0) {
        const long long s = a.n_in - C::HIST + lane;
        if constexpr (HALF) {
            const unsigned v = s >= 0 ? reinterpret_cast<const unsigned *>(in)[s] : reinterpret_cast<const unsigned *>(hist)[s + C::HIST];
            reinterpret_cast<unsigned *>(a.hist_out + SF * a.hist_stride * ch)[lane] = v;
        } else {
            const float2 v = s >= 0 ? reinterpret_cast<const float2 *>(in)[s]
                                    : reinterpret_cast<const float2 *>(hist)[s + C::HIST];
            reinterpret_cast<float2 *>(a.hist_out + 2 * a.hist_stride * ch)[lane] = v;
        }
    }

    for (int tile = first_tile; tile < a.n_tiles; tile += a.n_groups) {
        const long long q0 = (long long)tile * C::TILE_IN;
        const bool interior = (q0 >= 32) && (q0 + C::TILE_IN <= a.n_in);
        // ---- stage samples [q0 - 32, q0 + TILE_IN) ----------------------------
        if constexpr (HALF) {
            if (interior) {
                // typed LDS-DMA: instruction t moves samples [32 t, 32 t + 32) of the image: 128 source bytes -> 16 slots
                const unsigned long long wb = (unsigned long long)(reinterpret_cast<const unsigned *>(in) + (q0 - 32));
                v4i32 rs;
                rs.x = __builtin_amdgcn_readfirstlane((int)(unsigned)wb);
                rs.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(wb >> 32)) & 0xffff;    // stride 0
                rs.z = 1 << 16;
                rs.w = 4 | (7 << 12) | (2 << 15);                                             // {DATA_FORMAT 16, NUM_FORMAT FLOAT, X <- R}
                unsigned voff = 2u * (unsigned)lane;
                asm volatile("" : "+v"(voff));
                const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds);
#pragma unroll
                for (int t = 0; t < (C::TILE_IN + 32) / 32; ++t) {
                    const unsigned soff = 128u * t;
                    const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_base + 256u * t);
                    // (M0 is a reserved register to the compiler: writing it here needs, and admits, no clobber entry)
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_format_x %1, %2, %3 offen lds"
                                 :: "s"(m0v), "v"(voff), "s"(rs), "s"(soff) : "memory");
                }
            }
        }
#pragma unroll
        for (int i = 0; i < C::NLOAD; ++i) {
            if (HALF && interior) break;
            // chunk index as an opaque 32-bit value: the DMA takes the SGPR-base + VGPR-offset form and no
            // 64-bit per-lane address is kept across the tile loop
            unsigned cc = 64 * i + lane;
            cc = cc < (unsigned)C::CHUNKS ? cc : (unsigned)C::CHUNKS - 1u;
            asm volatile("" : "+v"(cc));
            const long long s = q0 - 32 + 2 * (long long)cc;
            bool dma = interior;
            if constexpr (HALF) dma = false;       // structural: a HALF instance (M0 written from inline asm above) holds no compiler-managed LDS-DMA
            if (dma) {
                if constexpr (!HALF) glds16(reinterpret_cast<const char *>(in + 2 * (q0 - 32)) + 16u * cc, lds + 64 * i);
            } else {
                float2 v0, v1;
                const long long last = a.n_in - 1;
                if (s >= 0) v0 = sample(in, s <= last ? s : last);
                else v0 = sample(hist, s + C::HIST);
                if (s + 1 >= 0) v1 = sample(in, s + 1 <= last ? s + 1 : last);
                else v1 = sample(hist, s + 1 + C::HIST);
                lds[64 * i + lane] = (f32x4){v0.x, v0.y, v1.x, v1.y};
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

        if constexpr (KEYED) {
            // the tile's own samples are chunks 16 .. 16 + TILE_IN/2 of the image (the 32 before them are history)
            if (ch == 0 && pb == 0 && q0 < a.key_hi && q0 + C::TILE_IN > a.key_lo) {
                unsigned total = 0;
#pragma unroll
                for (int k = 0; k < (C::TILE_IN / 2 + 63) / 64; ++k) {
                    const int cc = 64 * k + lane;
                    const bool in_tile = C::TILE_IN / 2 >= 64 * (k + 1) || cc < C::TILE_IN / 2;
                    const f32x4 v = lds[16 + (in_tile ? cc : 0)];
                    const long long s = q0 + 2 * cc;
                    // the rule as the reference's source states it: both products and the sum rounded once each
                    const bool k0 = in_tile && s >= a.key_lo && s < a.key_hi &&
                                    __fadd_rn(__fmul_rn(v.x, v.x), __fmul_rn(v.y, v.y)) >= a.thr2;
                    const bool k1 = in_tile && s + 1 >= a.key_lo && s + 1 < a.key_hi &&
                                    __fadd_rn(__fmul_rn(v.z, v.z), __fmul_rn(v.w, v.w)) >= a.thr2;
                    total += (unsigned)__builtin_popcountll(__ballot(k0)) + (unsigned)__builtin_popcountll(__ballot(k1));
                }
                if (lane == 0 && total) atomicAdd(a.key_counter, (unsigned long long)total);
            }
        }

#pragma unroll 1
        for (int kt = 0; kt < C::KT; ++kt) {
        const f32x4 *wk = win + kt * (C::QT / 2);
        // ---- compute: window sample w meets input qi at row jj = qi + 16 - w.  The I and Q FMAs of a
        // (tap, sample) pair are one v_pk_fma_f32 (sxfir_decim_tile.hip.h: same bits, less power).
        f32x2 acc[4][4];                                    // every chain's first FMA (tap row 15) writes it: from an inline +0
#pragma unroll
        for (int t = 0; t < 10; ++t) {
            const f32x4 v = wk[t];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int w = 2 * t + s;
                const f32x2 x = s ? __builtin_shufflevector(v, v, 2, 3) : __builtin_shufflevector(v, v, 0, 1);
#pragma unroll
                for (int qi = 0; qi < 4; ++qi) {
                    const int jj = qi + 16 - w;
                    if (jj >= 0 && jj < 16) {
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) {
                            if (jj == 15 && (rr & 1)) pk_fma_hi_first(acc[qi][rr], hp[(4 * jj + rr) >> 1], x);
                            else if (jj == 15) pk_fma_lo_first(acc[qi][rr], hp[(4 * jj + rr) >> 1], x);
                            else if (rr & 1) pk_fma_hi(acc[qi][rr], hp[(4 * jj + rr) >> 1], x);
                            else pk_fma_lo(acc[qi][rr], hp[(4 * jj + rr) >> 1], x);
                        }
                    }
                }
            }
        }
        float ai[4][4], aq[4][4];
#pragma unroll
        for (int qi = 0; qi < 4; ++qi)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) { ai[qi][rr] = acc[qi][rr].x; aq[qi][rr] = acc[qi][rr].y; }

        // ---- reduce over p: low half-wave keeps inputs 0-1, high half-wave inputs 2-3
        float oi[2][4], oq[2][4];
#pragma unroll
        for (int qi = 0; qi < 2; ++qi)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                permlane32_swap(ai[qi][rr], ai[qi + 2][rr]);
                permlane32_swap(aq[qi][rr], aq[qi + 2][rr]);
                oi[qi][rr] = __fadd_rn(ai[qi][rr], ai[qi + 2][rr]);
                oq[qi][rr] = __fadd_rn(aq[qi][rr], aq[qi + 2][rr]);
            }

        // ---- store.  A lane holds outputs (ql*L + 4c .. +3) of the sub-tile for ql = 4g + 2p + {0, 1}:
        // 16-byte pieces scattered over the 4 KiB the wave produces.  They go through LDS so that
        // every global store instruction writes 1 KiB of consecutive addresses (whole lines).
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) {
            const int oc = ((4 * g + 2 * p + qi) * L + 4 * c) >> 1;          // chunk index inside the sub-tile
            obuf[C::swz(oc)] = (f32x4){oi[qi][0], oq[qi][0], oi[qi][1], oq[qi][1]};
            obuf[C::swz(oc) ^ 1] = (f32x4){oi[qi][2], oq[qi][2], oi[qi][3], oq[qi][3]};
        }
        const long long o0 = (q0 + kt * C::QT) * LT + L * pb;               // first output of the sub-tile (of this phase block)
        const long long o_end = a.n_in * LT;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int oc = 64 * k + lane;
            f32x4 v = obuf[C::swz(oc)];
            if constexpr (S32OUT) {
                const int2 w0 = tx_words(v.x, v.y, a.thr2), w1 = tx_words(v.z, v.w, a.thr2);
                v = (f32x4){__int_as_float(w0.x), __int_as_float(w0.y), __int_as_float(w1.x), __int_as_float(w1.y)};
            }
            // chunk oc of the sub-tile: input oc / (L / 2), chunk oc % (L / 2) of its L outputs
            const long long o = NPB == 1 ? o0 + 2 * oc : o0 + (long long)(oc / (L / 2)) * LT + 2 * (oc % (L / 2));
            if constexpr (HALF) {
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                if (o + 2 <= o_end)
                    __builtin_nontemporal_store((u32x2){pack_half2(v.x, v.y), pack_half2(v.z, v.w)}, reinterpret_cast<u32x2 *>(out + o));
            } else {
                if (o + 2 <= o_end) __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(out + 2 * o));
            }
        }
        }   // kt
    }
}

}  // namespace sxfir
