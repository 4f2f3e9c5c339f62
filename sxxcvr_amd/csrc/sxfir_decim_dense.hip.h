// Decimate-by-D (D = 8, 16, 32), 32 taps per phase (NT = 32*D), CF32 or S32 wire words: the dense-image
// form of the multi-column decimator (sxfir_decim_multi.hip.h) for BASELINE configs 3 (RX) and 5.  New code:
// the reference decimates inside the SX1255 (SoapySX.cpp:180-208 only programs the chip's divider).
//
// Same arithmetic, same numeric contract (2, 4) as decim_multi_kernel<D, 4>: lanes = (row half p, column
// group c of four phases, output group of 8), a 64-tap fmaf chain per lane and output, then the
// adjacent-pair tree over p and over the D/4 column groups.  What changes is the LDS image and everything
// that follows from it:
//
//   * The tile (TILE_OUT outputs + 31 halo rows of D samples) sits in LDS as it sits in HBM: linear.  A DMA
//     instruction (global_load_lds_dwordx4) moves 1 KiB of CONSECUTIVE bytes -- eight whole lines -- instead
//     of 64 pieces picked from up to 32 different rows, so it is cheap to issue and a tile needs fewer of
//     them (/32: 40 instead of 48).
//   * One 16-byte pad slot after every PADROWS rows (between DMA instructions: the pads cost no traffic).
//     /32: 40 848 bytes per workgroup, FOUR workgroups (16 waves) per CU where the de-interleaved image
//     (48 KiB) left three; /16: 37 296, /8: 35 280 bytes.
//   * Bank conflicts: chunk CPR - 2 - 2c + h of a row holds half h of column group c, so c supplies even
//     residues (mod 16 slots) and the pad count before a lane's rows the rest.  For each D a lane map
//     (which lane bit is which of c, p, g) exists under which the 16 lanes a ds_read_b128 is served with
//     (MI355X_MICROARCH.md, LDS: {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, +32) either hit 16 different
//     slots mod 16 or the very same address (lanes whose rows coincide read the same window: a broadcast).
//     Found by exhaustive search over the 720 assignments and checked by tools/lds_bank_model.py; among
//     the conflict-free maps the one with the cheapest reduction is used:
//         D = 32, pad per 16 rows: (b0..b5) = (c1, c2, g1, c0, p, g0)
//         D = 16, pad per  8 rows: (b0..b5) = (g0, g2, g1, c1, p, c0)
//         D =  8, pad per 16 rows: (b0..b5) = (g2, g3, g1, c0, p, g0)
//   * Reduction without LDS, in the contract's order: v_permlane16_swap over p (bit 4 in all three maps), then
//     the column tree with v_permlane32_swap where a column bit is lane bit 5 and DPP butterflies elsewhere
//     (row_ror:8 for bit 3, quad_perm for bits 0 and 1).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sxfir_common.hip.h"            // DecimMultiArgs, pk_fma_s_lo / pk_fma_s_hi, rgrp_table, half <-> float

namespace sxfir {

template <int D>
struct DecimDense {
    static_assert(D == 8 || D == 16 || D == 32, "dense image: D = 8, 16 or 32");
    static constexpr int NT = 32 * D;
    static constexpr int NCOL = D / 4;
    static constexpr int W = 4;                       // waves per workgroup
    static constexpr int GW = 32 / NCOL;              // output groups (of 8) per wave
    static constexpr int OW = 8 * GW;                 // outputs per wave
    static constexpr int TILE_OUT = W * OW;           // 512 / 256 / 128
    static constexpr int NROWS = TILE_OUT + 31;       // rows q in [M0 - 31, M0 + TILE_OUT)
    static constexpr int CPR = D / 2;                 // 16-byte chunks per row
    static constexpr int CH = NROWS * CPR;            // chunks of the image
    static constexpr int RPI = 64 / CPR;              // rows per DMA instruction
    static constexpr int PADROWS = D == 16 ? 8 : 16;  // one pad slot after every PADROWS rows
    static constexpr int NI = (CH + 63) / 64;         // DMA instructions per tile
    static constexpr int NIW = (NI + W - 1) / W;      // at most this many per wave
    static constexpr int LAST_LANES = CH - 64 * (NI - 1);   // lanes of the last instruction that are inside the image
    static constexpr int LDS_SLOTS = CH + (NROWS - 1) / PADROWS;
    static constexpr int WCH = 46;                    // window chunks per lane: 23 rows x 2
    static_assert(PADROWS % RPI == 0 && PADROWS % 8 == 0, "pads fall between DMA instructions and between 8-row window segments");
    static_assert(LDS_SLOTS * 16 <= 160 * 1024 / 4, "four workgroups per CU");
    // LDS slot the DMA instruction i (image chunks [64i, 64i + 64)) starts at
    static constexpr int dma_slot(int i) { return 64 * i + (i * RPI) / PADROWS; }
};

typedef int v4i32 __attribute__((ext_vector_type(4)));      // a buffer resource descriptor in four SGPRs

__device__ __forceinline__ float half_lo_to_float(unsigned w) { return half_bits_to_float(w & 0xffffu); }
__device__ __forceinline__ float half_hi_to_float(unsigned w) { return half_bits_to_float(w >> 16); }

// two outputs (four floats) leave as 16 bytes of CF32 or, CF16 storage, as two half pairs rounded once (8 bytes)
template <bool HALFOUT>
__device__ __forceinline__ void store_pair(char *dst, float i0, float q0, float i1, float q1)
{
    if constexpr (HALFOUT) {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        __builtin_nontemporal_store((u32x2){pack_half2(i0, q0), pack_half2(i1, q1)}, reinterpret_cast<u32x2 *>(dst));
    } else {
        __builtin_nontemporal_store((f32x4){i0, q0, i1, q1}, reinterpret_cast<f32x4 *>(dst));
    }
}

template <bool HALFOUT>
__device__ __forceinline__ void store_one(char *dst, float i0, float q0)
{
    if constexpr (HALFOUT) reinterpret_cast<unsigned *>(dst)[0] = pack_half2(i0, q0);
    else reinterpret_cast<float2 *>(dst)[0] = make_float2(i0, q0);
}

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

// butterfly over one lane bit (0, 1 or 3): both lanes of a pair end with the sum
template <int BIT>
__device__ __forceinline__ float butterfly_add(float v)
{
    static_assert(BIT == 0 || BIT == 1 || BIT == 3, "quad_perm covers lane bits 0 and 1, row_ror:8 bit 3");
    if constexpr (BIT == 0) return __fadd_rn(v, dpp_f32<0xB1>(v));        // quad_perm [1,0,3,2]
    else if constexpr (BIT == 1) return __fadd_rn(v, dpp_f32<0x4E>(v));   // quad_perm [2,3,0,1]
    else return __fadd_rn(v, dpp_f32<0x128>(v));                          // row_ror:8 = lane ^ 8 inside a row of 16
}

// ABL (profiling only): 1 = staging + stores without the FIR, 2 = FIR without staging, 3 = the real
// kernel with s_memtime stamps around its phases (a.stamps, 5 counters per wave as decim_multi_kernel)
// NTLD: the staging DMAs of image rows that no other tile reads are non-temporal loads (round 4: a read stream
// runs 4.6 % faster with them, tools/membench5.hip).  The image's last 31 rows are the next tile's halo: for all
// three ratios they begin exactly at DMA instruction 32 (TILE_OUT / RPI), so instructions 32.. stay plain loads whose
// lines are still in the XCD's L2 when the neighbouring workgroup asks for them.  NTLD = 1: instructions 0..31 nt;
// NTLD = 2: the image's FIRST 31 rows -- the re-read of the previous tile's halo, instructions 0 .. 31 / RPI, rounded up
// to a group of four (one per wave) -- are plain too: an nt re-read that reaches the L2 first would stream through
// without leaving the line for the neighbour's plain load, which then fetches it again (1.04 x the algorithmic bytes
// measured on the /4 kernel).
// SUBSET (/8, CF32 or S32 wire words; round 4): the scalar-tap form.  The four (row half p, column group c) tap subsets go to the four
// WAVES of the workgroup instead of to lane bits: wave ww = 2c + p holds its subset's 64 taps in 32 SGPR pairs (a.taps is
// then the subset-major table: subset s at 64 s, (jj, rr) at 4 jj + rr) and computes that subset's partial sum of ALL 512
// outputs of the tile -- lane -> output group G (8 outputs), the same 46 window reads and 512 packed FMAs per lane, one
// VGPR operand fewer per FMA.  The four partials of an output meet through LDS (the dead image): every wave writes its 512
// partials, then wave ww sums the four of outputs [128 ww, 128 ww + 128) in the contract's tree, (p0 + p1) per column
// group, then the column groups, and stores one kilobyte.  Lane -> G is the conflict-free map of the /4 scalar-tap kernel
// (even groups in the first 16-lane service group of a half-wave, odd groups in the second), which keeps the lanes of a
// service group on 16 different slots mod 16 with this image's one pad per 16 rows.
// HC (halo carry; round 4): a workgroup walks a RUN of consecutive tiles instead of every NG-th one, and the image's last 31 rows --
// the next tile's halo -- are copied inside LDS to its first 31 instead of being fetched again: at /32 the halo is 31 rows of a
// 159-row image, a fifth of everything that moves from L2 into LDS.  Each wave copies exactly the chunks its own DMA instructions
// (32 + ww + 4k) are about to overwrite -- read, then its DMAs, then the writes: program order inside the wave, no barrier added.
// HALFIN (round 5): CF16 storage (IQ as IEEE half pairs in HBM, fp32 arithmetic, outputs rounded to half once; BASELINE config 5's
// fp16 leg).  The image in LDS is the SAME CF32 image: the texture path converts on the way in.  gfx950's LDS-DMA exists for
// one typed load, buffer_load_format_x; with a buffer descriptor of format {16, FLOAT} it fetches one half per lane and writes
// one float per lane, so one instruction turns 128 consecutive source bytes (32 samples) into 16 slots of the image -- no
// v_cvt_f32_f16 (184 per 512 FMAs in decim_multi_kernel<.., CF16>, 13.6 % of its FIR phase), no ds_write, and the FIR below is the
// CF32 kernel's, conversion-free.  Bit for bit the conversion v_cvt_f32_f16 makes for every finite half, zero, subnormal and
// infinity (tools/typed_dma_probe.hip; a NaN stays a NaN with another payload).  Four typed instructions replace one 1-KiB DMA.
template <int D, int ABL = 0, bool S32IN = false, int NTLD = 0, bool SUBSET = false, bool HC = false, bool HALFIN = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void
decim_dense_kernel(const DecimMultiArgs a)
{
    using C = DecimDense<D>;
    constexpr int SB = HALFIN ? 4 : 8;                  // bytes per complex sample in HBM
    static_assert(!HALFIN || (!S32IN && !HC && ABL == 0), "CF16 storage: the shipped forms (VGPR taps; /8: scalar-tap subsets)");
    static_assert(!(HC && SUBSET), "halo carry: the VGPR-tap forms");
    static_assert(!SUBSET || (D == 8 && (ABL == 0 || (ABL == 1 && !S32IN))), "subset form: /8 (CF32 or S32 wire words: the table then holds the taps times 2^-31)");
    __shared__ __attribute__((aligned(16))) f32x4 lds[C::LDS_SLOTS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int ww = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b0 = lane & 1, b1 = (lane >> 1) & 1, b2 = (lane >> 2) & 1, b3 = (lane >> 3) & 1, b4 = (lane >> 4) & 1, b5 = lane >> 5;
    // the lane maps of the header comment
    const int p = SUBSET ? (ww & 1) : b4;
    const int c = SUBSET ? (ww >> 1) : (D == 32 ? (b3 | (b0 << 1) | (b1 << 2)) : (D == 16 ? (b5 | (b3 << 1)) : b3));
    int G = D == 32 ? (b5 | (b2 << 1)) : (D == 16 ? (b0 | (b2 << 1) | (b1 << 2)) : (b5 | (b2 << 1) | (b0 << 2) | (b1 << 3)));
    if constexpr (SUBSET) G = (int)((rgrp_table((lane & 31) >> 3) >> (5 * (lane & 7))) & 31u) + (lane & 32);
    const int ch = blockIdx.y;

    const char *in = reinterpret_cast<const char *>(a.in) + (long long)SB * a.in_stride * ch;
    char *out = reinterpret_cast<char *>(a.out) + (long long)SB * a.out_stride * ch;

    // lane taps: h[kl], kl = 4*jj + rr  <->  tap D*(16*p + jj) + 4c + rr
    f32x2 hp[SUBSET ? 1 : 32];
    f32x2 hs[32];
    if constexpr (SUBSET) {
        const __attribute__((address_space(4))) f32x2 *tq = (const __attribute__((address_space(4))) f32x2 *)a.taps;
#pragma unroll
        for (int m = 0; m < 32; ++m) hs[m] = tq[32 * ww + m];
        hp[0] = (f32x2){0.0f, 0.0f};
    } else {
#pragma unroll
        for (int m = 0; m < 32; ++m) hs[m] = (f32x2){0.0f, 0.0f};
    }
#pragma unroll
    for (int k = 0; k < (SUBSET ? 0 : 32); ++k) {
        const float *t = a.taps + D * (16 * p + (k >> 1)) + 4 * c + 2 * (k & 1);
        float t0 = t[0], t1 = t[1];
        if constexpr (S32IN) {                        // a power of two commutes with the FMA
            t0 = __fmul_rn(t0, 4.656612873077393e-10f);
            t1 = __fmul_rn(t1, 4.656612873077393e-10f);
        }
        hp[k] = (f32x2){t0, t1};
    }

    // Window: rows 8u .. 8u + 22 of the image (u = output group - 2p + 2), chunks CPR - 2 - 2c + {0, 1} of each;
    // slot = chunk + (pads before its row).  The window's three 8-row segments -- steps [0, 16), [16, 32),
    // [32, 46) -- each lie between two pad positions, so a segment has one base: win[r] = chunk 0 of the
    // window + pads before row 8u + 8r.
    const int u = (SUBSET ? 0 : C::GW * ww) + G - 2 * p + 2;
    const f32x4 *win0 = lds + (C::CPR * 8 * u + C::CPR - 2 - 2 * c + (8 * u) / C::PADROWS);
    const f32x4 *win1 = win0 + ((8 * u + 8) / C::PADROWS - (8 * u) / C::PADROWS);
    const f32x4 *win2 = win0 + ((8 * u + 16) / C::PADROWS - (8 * u) / C::PADROWS);

    const int NG = a.n_groups;
    // 32-bit tile bounds instead of 64-bit sample arithmetic per tile: tiles [1, tile_hi] lie wholly inside this call's
    // input (tile 0 reaches into the history), tiles below n_full store all TILE_OUT outputs
    constexpr int LOG_T = C::TILE_OUT == 512 ? 9 : (C::TILE_OUT == 256 ? 8 : 7);
    constexpr int LOG_D = D == 32 ? 5 : (D == 16 ? 4 : 3);
    static_assert((1 << LOG_T) == C::TILE_OUT && (1 << LOG_D) == D, "powers of two");
    const long long q_hi = a.n_in > 0 ? ((a.n_in - 1) >> LOG_D) + 1 - C::TILE_OUT : -1;
    const int tile_hi = q_hi < 0 ? -1 : (int)(q_hi >> LOG_T);
    const int n_full = (int)(a.n_out >> LOG_T);
    const int wg = (NG % 8 == 0) ? (int)(blockIdx.x % 8) * (NG / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    // tiles of this workgroup: every NG-th from wg on, or (HC) the run [wg K, wg K + K)
    const int run_len = HC ? (a.n_tiles + NG - 1) / NG : 1;
    const int first_tile = HC ? wg * run_len : wg;
    const int end_tile = HC ? (first_tile + run_len < a.n_tiles ? first_tile + run_len : a.n_tiles) : a.n_tiles;
    const int tile_step = HC ? 1 : NG;
    // fused history carry-over (as decim_multi_kernel): the tail of (hist ++ in) becomes the next history
    // (round 6: on all four waves with a thread's loads in flight before its first store -- as a loop of one wave it put up to sixteen
    // dependent round trips in front of the work of the workgroup that owns the call's LAST tile, the slowest one of a small call)
    if (HC ? (first_tile <= a.n_tiles - 1 && a.n_tiles - 1 < end_tile) : first_tile == (a.n_tiles - 1) % NG) {
        char *ho = reinterpret_cast<char *>(a.hist_out) + (long long)SB * a.hist_stride * ch;
        const char *hi = reinterpret_cast<const char *>(a.hist) + (long long)SB * a.hist_stride * ch;
        static_assert(C::NT % 256 == 0, "whole passes of the workgroup");
        constexpr int PER = C::NT / 256;
        float2 hv[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const long long s = a.n_in - C::NT + tid + 256 * k;
            const char *src = s >= 0 ? in + SB * s : hi + SB * (s + C::NT);
            if constexpr (HALFIN) hv[k].x = __uint_as_float(*reinterpret_cast<const unsigned *>(src));
            else hv[k] = *reinterpret_cast<const float2 *>(src);
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            if constexpr (HALFIN) reinterpret_cast<unsigned *>(ho)[tid + 256 * k] = __float_as_uint(hv[k].x);
            else reinterpret_cast<float2 *>(ho)[tid + 256 * k] = hv[k];
        }
    }

    // the kernel arguments as they lie in the kernarg segment, through a pointer the compiler cannot see through: a load
    // from it stays where it is written (the rare paths of the tile loop)
    auto rare_args = [&]() __attribute__((always_inline)) {
        const __attribute__((address_space(4))) DecimMultiArgs *ap =
            (const __attribute__((address_space(4))) DecimMultiArgs *)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ap));
        return ap;
    };

    unsigned long long ph[5] = {0, 0, 0, 0, 0}, tk = 0;
    if constexpr (ABL == 3) tk = __builtin_amdgcn_s_memtime();
#define SXFIR_PHASE(k) \
    if constexpr (ABL == 3) { \
        const unsigned long long t_now = __builtin_amdgcn_s_memtime(); \
        ph[k] += t_now - tk; \
        tk = t_now; \
    }

    // HBM -> LDS for one tile: instruction i = ww + 4*i0 moves image chunks [64i, 64i + 64) to the slots from
    // dma_slot(i) on.  dma_slot(ww + 4*i0) = dma_slot(4*i0) + (64 + PER_I) * ww: /32 has no pad inside a group
    // of four instructions (PER_I = 0), /16 and /8 one after every instruction (PER_I = 1).
    constexpr int PER_I = C::dma_slot(1) - 64;
    static_assert(C::dma_slot(7) == C::dma_slot(4) + 3 * (64 + PER_I), "pads inside a group of four instructions are uniform");
    // carry (HC): the image holds tile - 1: its rows TILE_OUT .. TILE_OUT + 30 are this tile's rows 0 .. 30
    constexpr int HALO_INSTR = 31 / C::RPI;                              // DMA instructions that hold halo rows only
    constexpr int CARRY_SHIFT = 64 * 32 + C::TILE_OUT / C::PADROWS;      // slots between a halo row's two places
    static_assert(C::TILE_OUT % C::PADROWS == 0, "the pads before a row and before the row TILE_OUT above it differ by a constant");
    // (HALFIN) LDS byte address of the wave's first staging slot, a scalar: M0 of the typed DMA = this + a constant
    const unsigned lds_wave_base = HALFIN ? __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds + (64 + PER_I) * ww)) : 0u;
    auto stage = [&](int tile, bool carry, bool last_of_run) __attribute__((always_inline)) {
        const long long M0 = (long long)tile * C::TILE_OUT;
        const long long s_first = D * (M0 - 31) - (D - 1);               // first sample of the image
        const bool interior = tile >= 1 && tile <= tile_hi;
        const char *base = in + SB * s_first + (128 * SB) * ww;
        if constexpr (ABL == 2) return;
        if constexpr (HALFIN) {
            if (interior) {
                // Typed LDS-DMA: the wave's old instruction i = ww + 4 i0 (64 slots = 128 samples) becomes four
                // buffer_load_format_x ... lds of 16 slots each: lane l fetches the half at byte 2 l of 128 consecutive source
                // bytes and the texture path writes its float to LDS address M0 + 4 l.  The descriptor is based at the wave's
                // first byte of THIS tile (64-bit base, rebuilt per tile from scalars), so every offset is a small constant
                // and a call may be as long as it likes; {DATA_FORMAT 16, NUM_FORMAT FLOAT, X <- R}.
                const unsigned long long wb = (unsigned long long)base;
                v4i32 rs;
                rs.x = __builtin_amdgcn_readfirstlane((int)(unsigned)wb);
                rs.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(wb >> 32)) & 0xffff;    // stride 0
                rs.z = 1 << 20;                                                               // bytes addressable from the base: a tile's share and more
                rs.w = 4 | (7 << 12) | (2 << 15);
                unsigned voff = 2u * (unsigned)lane;
                asm volatile("" : "+v"(voff));
#pragma unroll
                for (int i0 = 0; i0 < C::NIW; ++i0) {
                    const int i = ww + 4 * i0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        // slots of the image this instruction fills: 16, or fewer / none in the image's last instruction
                        const int valid = i < C::NI - 1 ? 16 : (C::LAST_LANES - 16 * j < 0 ? 0 : (C::LAST_LANES - 16 * j > 16 ? 16 : C::LAST_LANES - 16 * j));
                        if (i0 < C::NIW - 1 || i < C::NI - 1 || (i == C::NI - 1 && lane < 4 * valid)) {
                            const unsigned soff = 2048u * i0 + 128u * j;                      // bytes from the wave's base
                            const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_wave_base + 16u * (unsigned)(C::dma_slot(4 * i0) + 16 * j));
                            // (M0 is a reserved register to the compiler -- it sets it right before each of its own uses and keeps nothing
                            // in it across an asm statement -- so writing it here needs, and admits, no clobber entry)
                            // NTLD = 2 as in the CF32 form: rows no other tile reads stream through the L2 (nt), both halos stay plain
                            constexpr int FIRST_NT_H = (31 / C::RPI + 4) / 4;
                            if (NTLD == 2 && i0 >= FIRST_NT_H && i0 < 8)
                                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_format_x %1, %2, %3 offen nt lds"
                                             :: "s"(m0v), "v"(voff), "s"(rs), "s"(soff) : "memory");
                            else
                                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_format_x %1, %2, %3 offen lds"
                                             :: "s"(m0v), "v"(voff), "s"(rs), "s"(soff) : "memory");
                        }
                    }
                }
                return;
            }
        }
        // (structural, not left to the optimizer: an instance whose typed front end writes M0 from inline asm holds no
        // compiler-managed LDS-DMA at all -- LLVM may hoist or merge ITS M0 set-up across an asm statement it cannot see into)
        if constexpr (!HALFIN) if (interior) {
            f32x4 hv[C::NIW - 8];
            if constexpr (HC) {
                if (carry) {
#pragma unroll
                    for (int k = 0; k < C::NIW - 8; ++k) {
                        const int i = 32 + ww + 4 * k;
                        if (i < C::NI - 1 || (i == C::NI - 1 && lane < C::LAST_LANES))
                            hv[k] = lds[(64 + PER_I) * ww + C::dma_slot(4 * (8 + k)) + lane];
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // read before this wave's DMAs overwrite them
                }
            }
#pragma unroll
            for (int i0 = 0; i0 < C::NIW; ++i0) {
                unsigned lo = 16u * lane;
                asm volatile("" : "+v"(lo));          // a 32-bit offset next to its use: SGPR base + VGPR offset form
                const char *bi = base + 4096 * i0;
                asm volatile("" : "+s"(bi));          // ... and the instruction's own base stays a scalar
                const int i = ww + 4 * i0;
                // the image's last instruction is only partly inside it
                // (all but a wave's last instruction lie inside it whatever the wave: decided at compile time)
                static_assert(4 * (C::NIW - 2) + C::W - 1 < C::NI - 1, "only i0 = NIW - 1 can reach the end of the image");
                if (i0 < C::NIW - 1 || i < C::NI - 1 || (i == C::NI - 1 && lane < C::LAST_LANES)) {
                    static_assert(C::TILE_OUT / C::RPI == 32, "the halo rows start at DMA instruction 32");
                    constexpr int FIRST_NT = NTLD == 2 ? (31 / C::RPI + 4) / 4 : 0;   // i0 below this: the halo re-read, plain
                    if constexpr (HC) {
                        // carried tiles skip the instructions that hold halo rows only; nobody else reads a run's rows, so every
                        // load is non-temporal but the run's last 31 rows (the next run's halo) and a run's first halo
                        if (4 * i0 + C::W - 1 < HALO_INSTR) { if (carry) continue; }
                        else if (4 * i0 < HALO_INSTR) { if (carry && i < HALO_INSTR) continue; }
                        const bool plain = i0 >= 8 ? last_of_run : (!carry && i0 < FIRST_NT);
                        if (NTLD && !plain) glds16<2>(bi + lo, lds + ((64 + PER_I) * ww + C::dma_slot(4 * i0)));
                        else glds16(bi + lo, lds + ((64 + PER_I) * ww + C::dma_slot(4 * i0)));
                    } else {
                        if (NTLD && i0 >= FIRST_NT && i0 < 8) glds16<2>(bi + lo, lds + ((64 + PER_I) * ww + C::dma_slot(4 * i0)));   // i = ww + 4 i0 < 32
                        else glds16(bi + lo, lds + ((64 + PER_I) * ww + C::dma_slot(4 * i0)));
                    }
                }
            }
            if constexpr (HC) {
                if (carry) {
#pragma unroll
                    for (int k = 0; k < C::NIW - 8; ++k) {
                        const int i = 32 + ww + 4 * k;
                        if (i < C::NI - 1 || (i == C::NI - 1 && lane < C::LAST_LANES))
                            lds[(64 + PER_I) * ww + C::dma_slot(4 * (8 + k)) + lane - CARRY_SHIFT] = hv[k];
                    }
                }
            }
            return;
        }
        {
            // edge tiles (first / last of a call): through registers, sample by sample.  What only they need is read
            // from the kernel arguments here, not kept in registers across the tile loop.
            const auto *ap = rare_args();
            const long long last = ap->n_in - 1;
            const char *hist = reinterpret_cast<const char *>(ap->hist) + (long long)SB * ap->hist_stride * ch;
#pragma nounroll
            for (int i0 = 0; i0 < C::NIW; ++i0) {
                const int i = ww + 4 * i0;
                if (i < C::NI - 1 || (i == C::NI - 1 && lane < C::LAST_LANES)) {
                    unsigned wds[4];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const long long s = s_first + 2 * (64 * i + lane) + e;
                        const char *src = s >= 0 ? in + SB * (s <= last ? s : last)
                                                 : hist + SB * (s + C::NT >= 0 ? s + C::NT : 0);
                        if constexpr (HALFIN) {
                            // (v_cvt_f32_f16: what the typed DMA of the interior tiles does, for every non-NaN half)
                            const unsigned w = reinterpret_cast<const unsigned *>(src)[0];
                            wds[2 * e] = __float_as_uint(half_lo_to_float(w));
                            wds[2 * e + 1] = __float_as_uint(half_hi_to_float(w));
                        } else {
                            wds[2 * e] = reinterpret_cast<const unsigned *>(src)[0];
                            wds[2 * e + 1] = reinterpret_cast<const unsigned *>(src)[1];
                        }
                    }
                    lds[64 * i + (i * C::RPI) / C::PADROWS + lane] = (f32x4){__uint_as_float(wds[0]), __uint_as_float(wds[1]),
                                                                            __uint_as_float(wds[2]), __uint_as_float(wds[3])};
                }
            }
        }
    };

    if (first_tile < end_tile) stage(first_tile, false, first_tile + tile_step >= end_tile);
    for (int tile = first_tile; tile < end_tile; tile += tile_step) {
        const long long M0 = (long long)tile * C::TILE_OUT;
        SXFIR_PHASE(1)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own DMAs landed ...
        __syncthreads();                                    // ... and everybody else's
        SXFIR_PHASE(2)

        // ---- compute: window sample w meets output i at local tap kl = 4*i + 63 - w
        f32x2 acc[8];
        if constexpr (ABL == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = (f32x2){0.0f, 0.0f};
        }
        if constexpr (ABL != 1)
#pragma unroll
        for (int t = 0; t < C::WCH; ++t) {
            const f32x4 *wp = t < 16 ? win0 : (t < 32 ? win1 : win2);
            const f32x4 v = wp[C::CPR * (t >> 1) + (t & 1)];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int w = 2 * t + s;
                f32x2 x = s ? __builtin_shufflevector(v, v, 2, 3) : __builtin_shufflevector(v, v, 0, 1);
                if constexpr (S32IN) x = (f32x2){(float)__float_as_int(x.x), (float)__float_as_int(x.y)};
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int kl = 4 * i + 63 - w;
                    if (kl >= 0 && kl < 64) {
                        // kl = 63 (w = 4i) is a chain's first tap: from an inline +0, no cleared register
                        if constexpr (SUBSET) {
                            if (kl == 63) pk_fma_s_hi_first(acc[i], hs[kl >> 1], x);
                            else if (kl & 1) pk_fma_s_hi(acc[i], hs[kl >> 1], x);
                            else pk_fma_s_lo(acc[i], hs[kl >> 1], x);
                        } else {
                            if (kl == 63) pk_fma_hi_first(acc[i], hp[kl >> 1], x);
                            else if (kl & 1) pk_fma_hi(acc[i], hp[kl >> 1], x);
                            else pk_fma_lo(acc[i], hp[kl >> 1], x);
                        }
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
        if constexpr (SUBSET) {
            // the four subsets' partials meet in the dead image: subset s at 256 s; chunk k (two outputs) of group G at slot
            // 4G + (k ^ ((G >> 1) & 3)): the eight lanes a ds_write_b128 is served with hit eight different slots mod 8
            // (unswizzled: 96 conflict cycles per wave and tile, measured), the read-back below is linear
            if constexpr (ABL != 1) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    lds[256 * ww + 4 * G + (k ^ ((G >> 1) & 3))] = (f32x4){acc[2 * k].x, acc[2 * k].y, acc[2 * k + 1].x, acc[2 * k + 1].y};
            }
            __syncthreads();
            f32x4 y;
            {
                const f32x4 v0 = lds[64 * ww + lane], v1 = lds[256 + 64 * ww + lane], v2 = lds[512 + 64 * ww + lane],
                            v3 = lds[768 + 64 * ww + lane];
                // (p0 + p1) of column group 0, (p0 + p1) of column group 1, then the two: the contract's tree
                f32x4 s01, s23;
#pragma unroll
                for (int e = 0; e < 4; ++e) { s01[e] = __fadd_rn(v0[e], v1[e]); s23[e] = __fadd_rn(v2[e], v3[e]); }
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = __fadd_rn(s01[e], s23[e]);
            }
            __syncthreads();                                // the exchange area may be overwritten by the next DMA
            if (tile + NG < a.n_tiles) stage(tile + NG, false, false);
            // the slot this lane read holds chunk kq of group Gq: a permutation inside each 64-byte group, so the wave's
            // store still covers one kilobyte of consecutive bytes
            const int Gq = 16 * ww + (lane >> 2), kq = (lane & 3) ^ ((Gq >> 1) & 3);
            const long long m = M0 + 8 * Gq + 2 * kq;
            char *dst = out + SB * m;
            if (tile < n_full) {
                store_pair<HALFIN>(dst, y.x, y.y, y.z, y.w);
            } else {
                const long long n_out = rare_args()->n_out;     // the call's last tile
                if (m + 2 <= n_out) store_pair<HALFIN>(dst, y.x, y.y, y.z, y.w);
                else if (m < n_out) store_one<HALFIN>(dst, y.x, y.y);
            }
            continue;
        }
        if (tile + tile_step < end_tile) stage(tile + tile_step, HC, tile + 2 * tile_step >= end_tile);
        SXFIR_PHASE(4)

        // ---- reduction in the order of the numeric contract.  p (lane bit 4): even 16-lane rows keep
        // outputs 0-3 of the lane's 8, odd rows 4-7
        float ri[4], rq[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            permlane16_swap(ai[i], ai[i + 4]);
            permlane16_swap(aq[i], aq[i + 4]);
            ri[i] = __fadd_rn(ai[i], ai[i + 4]);
            rq[i] = __fadd_rn(aq[i], aq[i + 4]);
        }
        const long long mg = M0 + C::OW * ww + 8 * G + 4 * p;            // first of the 4 outputs held now
        if constexpr (D == 16) {
            // c0 = lane bit 5: the low half-wave keeps outputs 0-1 of those four, the high half 2-3; c1 = bit 3
            float si[2], sq[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                permlane32_swap(ri[i], ri[i + 2]);
                permlane32_swap(rq[i], rq[i + 2]);
                si[i] = butterfly_add<3>(__fadd_rn(ri[i], ri[i + 2]));
                sq[i] = butterfly_add<3>(__fadd_rn(rq[i], rq[i + 2]));
            }
            const long long m = mg + 2 * b5;
            if (b3 == 0) {
                char *dst = out + SB * m;
                if (tile < n_full) {
                    store_pair<HALFIN>(dst, si[0], sq[0], si[1], sq[1]);
                } else {
                    const long long n_out = rare_args()->n_out;
                    if (m + 2 <= n_out) store_pair<HALFIN>(dst, si[0], sq[0], si[1], sq[1]);
                    else if (m < n_out) store_one<HALFIN>(dst, si[0], sq[0]);
                }
            }
        } else {
            // column groups by butterflies: /32: c0 = bit 3, c1 = bit 0, c2 = bit 1; /8: c0 = bit 3
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ri[i] = butterfly_add<3>(ri[i]);
                rq[i] = butterfly_add<3>(rq[i]);
            }
            if constexpr (D == 32) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ri[i] = butterfly_add<0>(ri[i]);
                    rq[i] = butterfly_add<0>(rq[i]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ri[i] = butterfly_add<1>(ri[i]);
                    rq[i] = butterfly_add<1>(rq[i]);
                }
            }
            // every lane of a column group set holds the same four outputs; two lanes of it store a pair each:
            // /32: the lanes with c0 = c2 = 0, c1 (bit 0) picks the pair; /8: both lanes, c0 (bit 3) picks the pair
            const int sel = D == 32 ? b0 : b3;
            const bool writer = D == 32 ? (lane & 0xA) == 0 : true;
            const long long m = mg + 2 * sel;
            if (writer) {
                const float s0 = sel ? ri[2] : ri[0], s1 = sel ? rq[2] : rq[0];
                const float s2 = sel ? ri[3] : ri[1], s3 = sel ? rq[3] : rq[1];
                char *dst = out + SB * m;
                if (tile < n_full) {
                    store_pair<HALFIN>(dst, s0, s1, s2, s3);
                } else {
                    const long long n_out = rare_args()->n_out;
                    if (m + 2 <= n_out) store_pair<HALFIN>(dst, s0, s1, s2, s3);
                    else if (m < n_out) store_one<HALFIN>(dst, s0, s1);
                }
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
