// LDS-tiled polyphase FIR decimator for gfx950, any decimation D = 4*NCOL
// (4, 8, 16, 32) with 32 taps per phase (NT = 32*D): BASELINE configs 3 (256
// taps, /8) and 5 (1024 taps, /32).  New code, like sxfir_decim_tile.hip.h: the
// reference decimates inside the SX1255 (SoapySX.cpp:180-208 only programs it).
//
// Idea: a decimate-by-D polyphase filter is the sum over NCOL = D/4 "column
// groups" c of a decimate-by-4 filter: group c owns phases r in [4c, 4c+4),
//     y[m] = sum_c  sum_j sum_rr  h[D*j + 4c + rr] * x[D*(m-j) - 4c - rr].
// Per group the samples it touches form a sub-stream of rows (one row = the 4
// samples x[D*q-4c-3 .. D*q-4c] of output row q).  The LDS image holds the NCOL
// sub-streams de-interleaved, each as a dense padded row sequence, so a lane's
// window is contiguous and the inner loop is the decimate-by-4 one: 46
// ds_read_b128 feeding 512 v_pk_fma_f32 per lane.  The de-interleave costs nothing:
// LDS-DMA (global_load_lds_dwordx4) takes a per-lane SOURCE address, so each
// 16-byte piece (half a row) is fetched from wherever it lives; pieces start
// on odd sample indices (8-byte aligned sources, verified on MI355X).
//
// Workgroup = W waves sharing one tile of W*OW outputs (+ 31 halo rows); every wave owns OW of them:
// lanes = (p, c, g): tap-row range p (top lane bits), column group c (next lane bits), output group g
// (8 outputs each).  PS = 2 (shipped): two row halves, 64 taps and 512 packed FMAs per lane and tile.
// PS = 4 (SXFIR_MULTI_PS=4, measured slower: LABBOOK.md 5.2): four row quarters, 32 taps and 256 packed FMAs, so
// that twice as many, lighter waves share a tile.
// Reduction: v_permlane32_swap / v_permlane16_swap over p and c bit 0, lane xor 8 / 4 / 2 for the
// remaining column bits: the adjacent-pair trees of the numeric contract, first over the
// row ranges (jsplit = PS), then over the columns (cw = 4).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <hip/hip_fp16.h>

#include "sxfir_decim_tile.hip.h"
#include "sxfir_common.hip.h"      // DecimMultiArgs, permlane16_swap, half <-> float (shared with the shipped kernels)

namespace sxfir {

template <int D, int W, bool HALF = false, int PS = 2>
struct DecimMulti {
    // HALF: IQ stored as IEEE half pairs (CF16, 4 bytes per sample; BASELINE config 5), fp32
    // arithmetic.  A 16-byte piece is then a whole row (4 samples) instead of half a row.
    // PS: the 32 tap rows are split over PS lanes (2 or 4); PS = 4 makes a wave half as heavy (32 taps,
    // 256 packed FMAs per lane and tile), so twice as many waves share a tile and hide each other's waits.
    static constexpr int NT = 32 * D;
    static constexpr int NCOL = D / 4;
    static constexpr int JR = 32 / PS;                    // tap rows per lane
    static constexpr int TPL = 4 * JR;                    // taps per lane
    static constexpr int PB = PS == 4 ? 2 : 1;            // lane bits that select the row range
    static constexpr int GW = 64 / (PS * NCOL);           // output groups per wave
    static constexpr int R = 8;                           // outputs per lane
    static constexpr int OW = GW * R;                     // outputs per wave
    static constexpr int TILE_OUT = W * OW;
    static constexpr int NROWS = TILE_OUT + 31;           // rows q in [M0 - 31, M0 + TILE_OUT)
    static constexpr int CPR = HALF ? 1 : 2;              // 16-byte chunks per row
    static constexpr int SBYTES = HALF ? 4 : 8;           // bytes per complex sample
    static constexpr int CH = CPR * NROWS;                // chunks per sub-stream
    static constexpr int PADP = 8 * CPR;                  // lane stride in chunks; one pad chunk after every PADP
    static constexpr int SUBSL = CH + CH / PADP + 1;
    // sub-stream pitch: a whole number of DMA instructions (64 slots) with room for the bank skew
    static constexpr int IPS = (SUBSL + (NCOL * PS >= 8 ? 15 : 0) + 63) / 64;   // DMA instructions per sub-stream
    static constexpr int SUBSTRIDE = IPS * 64;
    static constexpr int NI = NCOL * IPS;                 // DMA instructions per tile (all waves together)
    static constexpr int LDS_SLOTS = NI * 64;
    // waves per SIMD the register allocator must leave room for: what the LDS image allows, at most 4
    static constexpr int LDS_WAVES = (160 * 1024 / (LDS_SLOTS * 16)) * W / 4;
    static constexpr int MIN_WAVES = LDS_WAVES < 1 ? 1 : (LDS_WAVES > 4 ? 4 : LDS_WAVES);
    static constexpr int WROWS = JR + 7;                  // window rows per lane
    static constexpr int WCH = WROWS * CPR;               // window chunks per lane
    static_assert(D % 4 == 0 && (NCOL & (NCOL - 1)) == 0 && NCOL <= 8, "D must be 4, 8, 16 or 32");
    static_assert(PS == 2 || PS == 4, "row split of 2 or 4");
    static_assert(GW >= 1, "a wave holds at least one output group");
    static_assert(SUBSL < 4000, "the multiply-shift divisions below are exact below 4000 only");
    // Bank skew per column group.  A ds_read_b128 is served 16 lanes at a time (lanes 16k..16k+15) and is
    // conflict free when those lanes hit 16 different 16-byte slots mod 16.  The output groups g (low lane
    // bits) already differ: the lane stride is PADP + 1 slots, an odd number.  Column bit k sits at lane
    // bit 5 - PB - k; those below lane bit 4 vary inside a 16-lane group and get a skew equal to their own
    // lane bit, which makes the slot residue of a lane its low four lane bits (a permutation).
    static __device__ __forceinline__ int skew(int c)
    {
        int s = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int bit = 5 - PB - k;
            if ((NCOL >> k) > 1 && bit < 4) s |= ((c >> k) & 1) << bit;
        }
        return s;
    }
    // physical slot r of a sub-stream image -> logical chunk (a pad slot repeats its left neighbour)
    static __device__ __forceinline__ int logical(int r)
    {
        return HALF ? r - (((r + 1) * 7282) >> 16)        // (r+1)/9
                    : r - (((r + 1) * 3856) >> 16);       // (r+1)/17
    }
};

// ABL (profiling only): 1 = staging + stores without the FIR, 2 = FIR without staging, 3 = the real
// kernel with s_memtime stamps around its phases (a.stamps)
// S32IN: the input (and the history) are S32_LE I2S wire words instead of CF32 (convert_rx_buffer,
// SoapySX.cpp:103-112, folded in: int->float on load, the exact 2^-31 scale in the taps).
template <int D, int W, bool HALF = false, int ABL = 0, int PS = 2, bool S32IN = false>
__global__ __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(DecimMulti<D, W, HALF, PS>::MIN_WAVES, 8))) void
decim_multi_kernel(const DecimMultiArgs a)
{
    using C = DecimMulti<D, W, HALF, PS>;
    __shared__ __attribute__((aligned(16))) f32x4 lds[C::LDS_SLOTS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int ww = __builtin_amdgcn_readfirstlane(tid >> 6);
    // row range: the top PB lane bits (PS = 4: p = 2*bit5 + bit4); column group: the bits below,
    // c bit 0 highest
    const int p = lane >> (6 - C::PB);
    int c = 0;
    if (C::NCOL >= 2) c |= (lane >> (5 - C::PB)) & 1;
    if (C::NCOL >= 4) c |= ((lane >> (4 - C::PB)) & 1) << 1;
    if (C::NCOL >= 8) c |= ((lane >> (3 - C::PB)) & 1) << 2;
    const int g = lane & (C::GW - 1);
    const int ch = blockIdx.y;

    // byte views: samples are 8 (CF32) or 4 (CF16) bytes
    const char *in = reinterpret_cast<const char *>(a.in) + (long long)C::SBYTES * a.in_stride * ch;
    const char *hist = reinterpret_cast<const char *>(a.hist) + (long long)C::SBYTES * a.hist_stride * ch;
    char *out = reinterpret_cast<char *>(a.out) + (long long)C::SBYTES * a.out_stride * ch;

    // lane taps: h[kl], kl = 4*jj + rr  <->  tap D*(JR*p + jj) + 4c + rr
    float h[C::TPL];
#pragma unroll
    for (int kl = 0; kl < C::TPL; ++kl) {
        const float t = a.taps[D * (C::JR * p + (kl >> 2)) + 4 * c + (kl & 3)];
        h[kl] = S32IN ? __fmul_rn(t, 4.656612873077393e-10f) : t;      // a power of two commutes with the FMA
    }
    f32x2 hp[C::TPL / 2];      // the same taps as 64-bit register pairs for the packed FMAs (CF32)
#pragma unroll
    for (int k = 0; k < C::TPL / 2; ++k) hp[k] = (f32x2){h[2 * k], h[2 * k + 1]};

    const int G = ww * C::GW + g;                          // output group inside the workgroup tile
    // first window row: 8G - JR*p - JR + 1, counted from the tile's first row M0 - 31
    const int cc0 = C::PADP * (G - (C::JR / 8) * p + 4 - C::JR / 8);   // first window chunk (multiple of PADP)
    const f32x4 *win = lds + (c * C::SUBSTRIDE + C::skew(c) + cc0 + cc0 / C::PADP);

    // Tile schedule: in pass i the n_groups workgroups cover the consecutive tiles [i*G, (i+1)*G), dealt so
    // that the workgroups of one XCD (blockIdx % 8 shares an XCD and its L2; speed only) hold a contiguous
    // block of them: a tile's 31 halo rows were then fetched by the same XCD moments ago and hit its L2
    // instead of going to HBM a second time (TCC_MISS: 24 % more reads than algorithmic at /16 and /32
    // before this, none after).
    const int NG = a.n_groups;
    const int first_tile = (NG % 8 == 0) ? (int)(blockIdx.x % 8) * (NG / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    // fused history carry-over: the last wave of the workgroup that owns the last tile copies the
    // tail of (hist ++ in) into the plan's other history buffer
    if (first_tile == (a.n_tiles - 1) % NG && ww == W - 1) {
        char *ho = reinterpret_cast<char *>(a.hist_out) + (long long)C::SBYTES * a.hist_stride * ch;
        for (int j = lane; j < C::NT; j += 64) {
            const long long s = a.n_in - C::NT + j;
            const char *src = s >= 0 ? in + C::SBYTES * s : hist + C::SBYTES * (s + C::NT);
            if constexpr (HALF) reinterpret_cast<unsigned *>(ho)[j] = *reinterpret_cast<const unsigned *>(src);
            else reinterpret_cast<float2 *>(ho)[j] = *reinterpret_cast<const float2 *>(src);
        }
    }

    // Tile-invariant part of the staging: sample offset (from the tile's first row, sample
    // x[D*(M0-31)]) of the 16-byte piece each of this wave's DMA instructions fetches for this lane.
    constexpr int NIW = (C::NI + W - 1) / W;
    constexpr int PBIAS = 4 * (C::NCOL - 1) + 3;
    unsigned poff[NIW];
#pragma unroll
    for (int i0 = 0; i0 < NIW; ++i0) {
        // Issue order: row block kb of every sub-stream before row block kb+1, so the NCOL pieces
        // that share a 128-byte line of the input are fetched close together (L1 hits at D = 32).
        const int o = i0 * W + ww;
        const int kb = o / C::NCOL, cs = o % C::NCOL;      // sub-stream (column group) of this instruction
        int r = 64 * kb + lane - C::skew(cs);              // slot inside the sub-stream image
        r = r < 0 ? 0 : (r >= C::SUBSL - 1 ? C::SUBSL - 2 : r);
        const int cc = C::logical(r);
        // CF32: piece = samples x[D*q - 4c - 3 + 2*half], +1 with q = M0 - 31 + (cc >> 1), half = cc & 1
        // CF16: piece = the whole row x[D*q - 4c - 3 .. D*q - 4c] with q = M0 - 31 + cc
        // biased by PBIAS so that it is never negative: the DMA takes it as an unsigned 32-bit offset
        poff[i0] = (HALF ? D * cc - 4 * cs - 3 : D * (cc >> 1) - 4 * cs - 3 + 2 * (cc & 1)) + PBIAS;
    }

    unsigned long long ph[5] = {0, 0, 0, 0, 0}, tk = 0;
    if constexpr (ABL == 3) tk = __builtin_amdgcn_s_memtime();
#define SXFIR_PHASE(k) \
    if constexpr (ABL == 3) { \
        const unsigned long long t_now = __builtin_amdgcn_s_memtime(); \
        ph[k] += t_now - tk; \
        tk = t_now; \
    }
    // HBM -> LDS for one tile: the W waves share the NI DMA instructions
    auto stage = [&](int tile) __attribute__((always_inline)) {
        const long long M0 = (long long)tile * C::TILE_OUT;
        // samples of the tile: [D*(M0-32)+1, D*(M0+TILE_OUT-1)]; interior = all inside `in`
        const bool interior = (M0 >= 32) && (D * (M0 + C::TILE_OUT - 1) <= a.n_in - 1);
        const long long s_base = D * (M0 - 31) - PBIAS;
        const char *base = in + C::SBYTES * s_base;
#pragma unroll
        for (int i0 = 0; i0 < NIW; ++i0) {
            const int o = i0 * W + ww;
            if (o < C::NI && ABL != 2) {
                const int i = (o % C::NCOL) * C::IPS + o / C::NCOL;      // instruction's place in the LDS image
                // the empty asm keeps the offset a 32-bit value next to its use: the DMA then takes the
                // SGPR-base + VGPR-offset form and nothing 64-bit is hoisted out of the tile loop
                unsigned po = poff[i0];
                asm volatile("" : "+v"(po));
                if constexpr (ABL == 4 || ABL == 5) po = 2u * ((64u * o + lane) % 2048u);   // linear sources (wrong data): TA cost probe
                if (interior) {
                    glds16(base + C::SBYTES * po, lds + 64 * i);
                } else {
                    // edge tiles (first / last of a call): through registers, sample by sample
                    const long long last = a.n_in - 1;
                    unsigned wds[4];
#pragma unroll
                    for (int e = 0; e < 16 / C::SBYTES; ++e) {
                        const long long s = s_base + po + e;
                        const char *src = s >= 0 ? in + C::SBYTES * (s <= last ? s : last)
                                                 : hist + C::SBYTES * (s + C::NT >= 0 ? s + C::NT : 0);
                        if constexpr (HALF) {
                            wds[e] = *reinterpret_cast<const unsigned *>(src);
                        } else {
                            wds[2 * e] = reinterpret_cast<const unsigned *>(src)[0];
                            wds[2 * e + 1] = reinterpret_cast<const unsigned *>(src)[1];
                        }
                    }
                    lds[64 * i + lane] = (f32x4){__uint_as_float(wds[0]), __uint_as_float(wds[1]),
                                                 __uint_as_float(wds[2]), __uint_as_float(wds[3])};
                }
            }
        }
    };

    // Per tile: wait for its data | FIR arithmetic out of LDS | barrier | issue the NEXT tile's DMAs |
    // reduce and store this tile's outputs (registers only) while those DMAs are in flight.
    if (first_tile < a.n_tiles) stage(first_tile);
    for (int tile = first_tile; tile < a.n_tiles; tile += NG) {
        const long long M0 = (long long)tile * C::TILE_OUT;
        SXFIR_PHASE(1)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // own DMAs landed ...
        __syncthreads();                                   // ... and everybody else's
        SXFIR_PHASE(2)

        // ---- compute: window sample w meets output i at local tap kl = 4*i + TPL - 1 - w ---
        // The I and Q FMAs of a (tap, sample) pair are one v_pk_fma_f32 (sxfir_decim_tile.hip.h: same
        // bits, half the instructions, measurably less power).  CF16: each sample is converted once
        // (two v_cvt_f32_f16, SDWA picks the half) into the pair the packed FMAs take.
        constexpr int SPC = HALF ? 4 : 2;                  // samples per 16-byte chunk
        f32x2 acc[8];
        if constexpr (ABL == 1 || ABL == 4) {              // (no FIR in these builds: the chains' first FMAs are absent)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = (f32x2){0.0f, 0.0f};
        }
        if constexpr (ABL != 1 && ABL != 4)
#pragma unroll
        for (int t = 0; t < C::WCH; ++t) {
            const f32x4 v = win[t + t / C::PADP];
#pragma unroll
            for (int s = 0; s < SPC; ++s) {
                const int w = SPC * t + s;
                f32x2 x;
                if constexpr (HALF) {
                    const unsigned bits = __float_as_uint(s == 0 ? v.x : (s == 1 ? v.y : (s == 2 ? v.z : v.w)));
                    x = (f32x2){half_bits_to_float(bits & 0xFFFFu), half_bits_to_float(bits >> 16)};
                } else {
                    x = s ? __builtin_shufflevector(v, v, 2, 3) : __builtin_shufflevector(v, v, 0, 1);
                    if constexpr (S32IN) x = (f32x2){(float)__float_as_int(x.x), (float)__float_as_int(x.y)};
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int kl = 4 * i + C::TPL - 1 - w;
                    if (kl >= 0 && kl < C::TPL) {
                        // kl = TPL - 1 (w = 4i) is a chain's first tap: from an inline +0, no cleared register
                        if (kl == C::TPL - 1) pk_fma_hi_first(acc[i], hp[kl >> 1], x);
                        else if (kl & 1) pk_fma_hi(acc[i], hp[kl >> 1], x);
                        else pk_fma_lo(acc[i], hp[kl >> 1], x);
                    }
                }
            }
        }
        float ai[8], aq[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { ai[i] = acc[i].x; aq[i] = acc[i].y; }

        if constexpr (ABL == 3) asm volatile("" ::"v"(ai[0]), "v"(aq[7]));   // the arithmetic ends here
        SXFIR_PHASE(3)
        __syncthreads();                                   // everyone is done reading this tile's image
        if (tile + NG < a.n_tiles) stage(tile + NG);
        SXFIR_PHASE(4)
        // ---- reductions, in the order of the numeric contract: adjacent-pair tree over the row
        // ranges p, then over the column groups c.  A swap step halves the outputs a lane holds
        // (v_permlane{16,32}_swap: no LDS); the remaining column bits are butterflies.
        float oi[4], oq[4];
        if constexpr (PS == 2) {
            // p = lane bit 5: outputs 0-3 stay on the low half-wave, 4-7 on the high one
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                permlane32_swap(ai[i], ai[i + 4]);
                permlane32_swap(aq[i], aq[i + 4]);
                oi[i] = __fadd_rn(ai[i], ai[i + 4]);
                oq[i] = __fadd_rn(aq[i], aq[i + 4]);
            }
        } else {
            // p bit 0 = lane bit 4 (ranges 0+1, 2+3): even 16-lane rows keep outputs 0-3, odd rows 4-7
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                permlane16_swap(ai[i], ai[i + 4]);
                permlane16_swap(aq[i], aq[i + 4]);
                oi[i] = __fadd_rn(ai[i], ai[i + 4]);
                oq[i] = __fadd_rn(aq[i], aq[i + 4]);
            }
        }
        const long long mw = M0 + (long long)ww * C::OW + 8 * g;
        if constexpr (C::NCOL == 1 && PS == 2) {
            const long long m = mw + 4 * p;
            char *dst = out + C::SBYTES * m;
            if (m + 4 <= a.n_out) {
                if constexpr (HALF) {
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    __builtin_nontemporal_store((u32x4){pack_half2(oi[0], oq[0]), pack_half2(oi[1], oq[1]),
                                                        pack_half2(oi[2], oq[2]), pack_half2(oi[3], oq[3])},
                                                reinterpret_cast<u32x4 *>(dst));
                } else {
                    __builtin_nontemporal_store((f32x4){oi[0], oq[0], oi[1], oq[1]}, reinterpret_cast<f32x4 *>(dst));
                    __builtin_nontemporal_store((f32x4){oi[2], oq[2], oi[3], oq[3]}, reinterpret_cast<f32x4 *>(dst + 16));
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (m + i < a.n_out) {
                        if constexpr (HALF) reinterpret_cast<unsigned *>(dst)[i] = pack_half2(oi[i], oq[i]);
                        else reinterpret_cast<float2 *>(dst)[i] = make_float2(oi[i], oq[i]);
                    }
            }
        } else {
            // ---- second swap step: PS = 2: column bit 0 (lane bit 4); PS = 4: p bit 1 (lane bit 5)
            float ri[2], rq[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if constexpr (PS == 2) {
                    permlane16_swap(oi[i], oi[i + 2]);
                    permlane16_swap(oq[i], oq[i + 2]);
                } else {
                    permlane32_swap(oi[i], oi[i + 2]);
                    permlane32_swap(oq[i], oq[i + 2]);
                }
                ri[i] = __fadd_rn(oi[i], oi[i + 2]);
                rq[i] = __fadd_rn(oq[i], oq[i + 2]);
            }
            // ---- remaining column bits: butterflies (every lane of the group ends with the sum)
            constexpr int CB0 = PS == 2 ? 1 : 0;            // first column bit still to reduce
            constexpr int CBITS = C::NCOL == 8 ? 3 : (C::NCOL == 4 ? 2 : (C::NCOL == 2 ? 1 : 0));
            int wmask = 0;                                  // lane bits that must be 0 on a writer
#pragma unroll
            for (int k = CB0; k < CBITS; ++k) {
                const int bit = 1 << (5 - C::PB - k);       // lane bit of column bit k
                wmask |= bit;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    ri[i] = __fadd_rn(ri[i], __shfl_xor(ri[i], bit));
                    rq[i] = __fadd_rn(rq[i], __shfl_xor(rq[i], bit));
                }
            }
            // outputs held: PS = 2: 4*bit5 + 2*bit4 + {0, 1};  PS = 4: 4*bit4 + 2*bit5 + {0, 1}
            const int b5 = (lane >> 5) & 1, b4 = (lane >> 4) & 1;
            const long long m = mw + (PS == 2 ? 4 * b5 + 2 * b4 : 4 * b4 + 2 * b5);
            if ((lane & wmask) == 0) {
                char *dst = out + C::SBYTES * m;
                if (m + 2 <= a.n_out) {
                    if constexpr (HALF) {
                        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                        __builtin_nontemporal_store((u32x2){pack_half2(ri[0], rq[0]), pack_half2(ri[1], rq[1])},
                                                    reinterpret_cast<u32x2 *>(dst));
                    } else {
                        __builtin_nontemporal_store((f32x4){ri[0], rq[0], ri[1], rq[1]}, reinterpret_cast<f32x4 *>(dst));
                    }
                } else if (m < a.n_out) {
                    if constexpr (HALF) reinterpret_cast<unsigned *>(dst)[0] = pack_half2(ri[0], rq[0]);
                    else reinterpret_cast<float2 *>(dst)[0] = make_float2(ri[0], rq[0]);
                }
            }
        }
        if constexpr (ABL == 3) ph[0] += 1;
    }
    if constexpr (ABL == 3) {
        if (lane == 0 && a.stamps) {
            unsigned long long *rec = a.stamps + 5 * ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * W + ww);
#pragma unroll
            for (int k = 0; k < 5; ++k) rec[k] = ph[k];
        }
    }
#undef SXFIR_PHASE
}

}  // namespace sxfir
