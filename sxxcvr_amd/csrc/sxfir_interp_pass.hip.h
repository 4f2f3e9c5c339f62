// Interpolate-by-8, 256 taps (32 per phase), CF32: the scalar-tap form of the TX half of BASELINE config 3 (gfx950).
//
// interp_tile_kernel (sxfir_interp_tile.hip.h) spreads a tile's work over lanes (p, g, c) -- tap-row half, input group,
// phase group -- so lanes of one wave use different taps, and the taps sit in VGPRs: three vector operands per packed
// FMA.  At the board's power cap the time is energy per sample, and the arithmetic probe (tools/valu_power_probe.hip,
// profiles/round4p_valu_power_probe.txt) prices a scalar tap operand at 12 % of the FIR phase at this kernel's mix of
// LDS reads.  A scalar operand is wave-uniform, so the work split changes:
//
//   * lane g (0..63) owns QI consecutive inputs q0 + QI g + {0..QI-1} of a tile of 64 QI inputs and produces ALL their
//     8 QI outputs (the description below is for QI = 4; QI = 2 ships, see the end);
//   * the four (phase group c, row half p) tap subsets are four PASSES over the same LDS image; a pass's 64 taps
//     h[(16p + jj)*8 + 4c + rr] are 32 SGPR pairs, loaded with four s_load_dwordx16 from a pass-major copy of the taps
//     (constant address space); the lane's window -- 36 samples, 18 ds_read_b128 -- is read ONCE per tile and serves all
//     four passes (the row halves' windows overlap, the phase groups share them): 18 LDS reads per 1024 v_pk_fma_f32
//     with the tap as the scalar operand, where the other kernel has 40 and a VGPR operand more per FMA;
//   * with the window in registers the input image is free again: the NEXT tile's three LDS-DMA instructions are
//     issued before the arithmetic starts, and a counted s_waitcnt (the stores issued behind them stay outstanding)
//     finds them landed a tile later -- a wave hides its own memory latency, two per SIMD suffice;
//   * the two row halves of a phase group meet inside the lane: y = P0 + P1 (the contract's tree, no v_permlane32_swap);
//   * a lane's 32 outputs are 256 consecutive bytes; the tile's 16 KiB go through LDS (chunk k of lane g at slot
//     16g + (k ^ (g & 15)): the eight lanes a ds_write_b128 is served with hit eight different slots, the read-back is
//     linear) so that every global store instruction writes four whole 256-byte rows;
//   * 16 KiB + the 2.3 KiB input image = 18.7 KB of LDS per wave: 8 waves per CU, 2 per SIMD; sixteen accumulator chains
//     keep the FMA pipe busy from two waves (the trade the /4 wide kernel makes).
//
// Measured (tools/ibench2.py, long interleaved visits, profiles/round4q_ibench_pass.txt): QI = 4 takes 4-5 % less time
// than interp_tile_kernel -- and on an all-zero input 16 % MORE (0.43-0.46 against 0.37-0.38 ms): its energy per sample is
// lower, its structure (two waves per SIMD, sixteen stores per wave in a burst) slower, and the two walls are close.
// QI = 2 halves the tile: 8 KiB of outputs + a 1.3 KiB image = 10 KB of LDS and 125 VGPRs, i.e. 16 waves per CU, eight
// stores per wave and tile, 17 window reads per 512 FMAs (still fewer than the VGPR-tap kernel's 20): structure 0.39-0.41
// ms on zeros, **9.6 % less time than interp_tile_kernel on random IQ** (0.4636 against 0.5124 ms on the same box).  That
// form ships for x8 with CF32 or S32 wire-word output (S32OUT: tx_word fused into the stores), with or without the
// keying count (KEYED); the other ratios keep interp_tile_kernel (whose x8 instances exist in the profiling build only).
//
// Numeric contract: partial_p = fmaf chain from +0.0f over j DESCENDING in [16p, 16p+16); y = P0 + P1 --
// the same chains in the same order as interp_tile_kernel, so the outputs are bit-identical.
//
// New code: in the reference the SX1255 interpolates what snd_pcm_writei hands it (SoapySX.cpp:1093).
#pragma once

#include <utility>

#include "sxfir_interp_tile.hip.h"
#include "sxfir_common.hip.h"            // pk_fma_s_lo / pk_fma_s_hi

namespace sxfir {

// QI = inputs per lane: 4 (tile of 256 inputs, 16 KiB of outputs, 8 waves per CU -- the form described above) or 2 (tile
// of 128 inputs, 8 KiB of outputs, 10 KB of LDS: 16 waves per CU when the registers allow, at 17 instead of 18 window
// reads per HALF as many FMAs).
// L = 8 (two phase groups, four passes) or, round 5, L = 4 (one phase group, two passes; four inputs per lane: a lane's sixteen
// outputs are one 128-byte line, and 18 window reads serve 512 packed FMAs -- with two inputs per lane they would serve 256).
template <int QI, int LL = 8>
struct InterpPass8 {
    static_assert(QI == 2 || QI == 4, "inputs per lane");
    static_assert(LL == 8 || (LL == 4 && QI == 4) || (LL == 16 && QI == 2), "x8; x4 with four inputs per lane; x16 with two");
    static constexpr int L = LL;
    static constexpr int TILE_IN = 64 * QI;               // inputs per tile
    static constexpr int HIST = 32;
    static constexpr int CHUNKS = (TILE_IN + HIST) / 2;   // staged chunks: samples [q0 - 32, q0 + TILE_IN)
    static constexpr int NLOAD = (CHUNKS + 63) / 64;      // DMA instructions
    static constexpr int IMG = NLOAD * 64;
    static constexpr int CPL = QI * LL / 2;               // output chunks per lane (QI inputs x L outputs x 8 bytes / 16)
    static constexpr int OBUF = 64 * CPL;                 // output chunks per tile
    static constexpr int NW = (QI + 16) / 2;              // window chunks of one pass
    static constexpr int NWU = NW + 8;                    // ... of both row halves together (the lane reads these once)
};

// One window chunk T (0..9) of a pass: samples w = 2T, 2T+1 of the lane's window meet input qi at tap row
// jj = qi + 16 - w (when 0 <= jj < 16); hs[(4 jj + rr) >> 1] holds the pass's taps (jj, rr) pairwise.
template <int QI, int T>
__device__ __forceinline__ void interp_pass_step(const f32x4 &v, const f32x2 (&hs)[32], f32x2 (&acc)[QI][4])
{
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int w = 2 * T + s;
        const f32x2 x = s ? __builtin_shufflevector(v, v, 2, 3) : __builtin_shufflevector(v, v, 0, 1);
#pragma unroll
        for (int qi = 0; qi < QI; ++qi) {
            const int jj = qi + 16 - w;
            if (jj >= 0 && jj < 16) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    // jj = 15 is a chain's first tap (w ascends, jj descends): from an inline +0, no cleared register
                    if (jj == 15 && (rr & 1)) pk_fma_s_hi_first(acc[qi][rr], hs[(4 * jj + rr) >> 1], x);
                    else if (jj == 15) pk_fma_s_lo_first(acc[qi][rr], hs[(4 * jj + rr) >> 1], x);
                    else if (rr & 1) pk_fma_s_hi(acc[qi][rr], hs[(4 * jj + rr) >> 1], x);
                    else pk_fma_s_lo(acc[qi][rr], hs[(4 * jj + rr) >> 1], x);
                }
            }
        }
    }
}

template <int QI, int... Ts>
__device__ __forceinline__ void interp_pass_steps(std::integer_sequence<int, Ts...>, const f32x4 (&win)[(QI + 16) / 2], const f32x2 (&hs)[32],
                                                  f32x2 (&acc)[QI][4])
{
    (interp_pass_step<QI, Ts>(win[Ts], hs, acc), ...);
}

// KEYED: the transmitter-keying count of the call's input (convert_tx_buffer's rule, SoapySX.cpp:132-133) rides along as in
// interp_tile_kernel: a lane's own QI inputs are the last QI samples of the window it has just read, so the count costs two
// products, a sum and a ballot per sample and one atomic per wave and tile.  The range and the threshold are read from the
// kernel arguments where they are used (the kernel has no scalar registers to spare).
// S32OUT: outputs leave as S32_LE I2S wire words with the keying bits (convert_tx_buffer, SoapySX.cpp:116-137), converted
// between the transposition buffer and the store as in interp_tile_kernel.
// COUNTED = false (profiling build only, SXFIR_IPASS_WAIT0=1): every wait for the staged image is s_waitcnt vmcnt(0)
// instead of the counted form below -- slower, and free of the counted form's premises (exactly CPL stores behind the
// next tile's DMAs, no other VMEM instruction between them); tests/test_gpu_variants.py holds the two bit-identical,
// tests/test_abi.py::test_shipped_code_object checks the premises in the shipped disassembly.
// LL = 16 (trial, round 5): four phase groups, eight passes, a lane's 2 x 16 outputs = two whole lines, 16 KiB of outputs per tile
// (8 waves per CU, as QI = 4 at x8).  LT: the plan's ratio, a multiple of LL -- x32, x48, x96 as LT / 16 PHASE BLOCKS walked
// inside the tile loop over the SAME window (an interpolator's phases never meet; block pb is the x16 problem on the taps
// h[j LT + 16 pb + r], pass-major table, block pb at 512 pb; an input's sixteen outputs of a block are one whole line).  The next
// tile's DMAs are awaited behind the FIRST block's stores (the same counted wait, one block earlier).
// PBSPLIT (round 6; x32, x48, x96): the calls the API really issues are small (writeStream blocks of 256 .. 8192 samples, SoapySX.cpp:969-1105,
// batched by the Device's TX chain) and a tile here is 128 inputs x LT outputs walked block after block by one wave: a 2^22-output call
// at x96 is 342 tiles for 2048 wave slots.  An interpolator's phases never meet, so the PBSPLIT instance deals (tile, phase block) ITEMS:
// the launch passes n_tiles = tiles x LT / 16 items, item v is block v % NPB of tile v / NPB (a tile's blocks next to each other in the
// schedule: its 1 KiB of input is fetched from HBM once), one block per loop iteration -- the x16 kernel's loop with a block's tap table
// and output stride.  No join, same bits.  sxfir_launch chooses it while a call has at most four times as many tiles as the chip holds waves.
template <int QI, bool KEYED = false, bool S32OUT = false, bool COUNTED = true, int LL = 8, int LT = LL, bool PBSPLIT = false>
__global__ __launch_bounds__(64) void interp8_pass_kernel(const InterpTileArgs a)
{
    using C = InterpPass8<QI, LL>;
    static_assert(LT % LL == 0 && (LT == LL || LL == 16), "phase blocks of sixteen");
    static_assert(!PBSPLIT || LT > LL, "items are phase blocks");
    constexpr int NPBT = LT / LL;                 // phase blocks per tile
    constexpr int NPB = PBSPLIT ? 1 : NPBT;       // ... walked per loop iteration (per item)
    __shared__ __attribute__((aligned(16))) f32x4 lds[C::IMG + C::OBUF];
    f32x4 *obuf = lds + C::IMG;

    const int lane = threadIdx.x;
    const int ch = blockIdx.y;
    const float *in = a.in + 2 * a.in_stride * ch;
    const float *hist = a.hist + 2 * a.hist_stride * ch;
    float *out = a.out + 2 * a.out_stride * ch;
    // the pass-major tap table: pass (c, p) at 64 * (2c + p), inside it (jj, rr) at 4 jj + rr
    const __attribute__((address_space(4))) f32x2 *tq = (const __attribute__((address_space(4))) f32x2 *)a.taps;

    // Tile schedule: in pass i the G workgroups cover the G consecutive tiles [iG, (i+1)G), dealt so that the workgroups of
    // one XCD (blockIdx % 8 shares an XCD; speed only) hold a contiguous block of the pass: a tile's 32-sample history is
    // its neighbour's tail, and the re-read then finds it in that XCD's L2 (dealt round robin, every history re-read went
    // to HBM: 1.028 x the algorithmic bytes, profiles/round4_3tx_summary.json of the first build)
    const int G = a.n_groups;
    const int first_tile = (G % 8 == 0) ? (int)(blockIdx.x % 8) * (G / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    // (x16 blocks: tiles are 128 inputs x up to 96 outputs each -- few per workgroup -- so the partial LAST round is dealt plainly,
    // tile R G + blockIdx.x: XCD-blocked it falls to the first XCDs alone, x96 0.55 against 0.46 ms; the x4 / x8 instances keep
    // their schedule)
    constexpr bool PLAIN_TAIL = LL == 16;
    const int R = a.n_tiles / G, rem = a.n_tiles % G;          // G <= n_tiles: at least one whole round
    if ((PLAIN_TAIL ? (rem ? (int)blockIdx.x == rem - 1 : first_tile == G - 1) : first_tile == (a.n_tiles - 1) % G) && lane < C::HIST) {
        const long long s = a.n_in - C::HIST + lane;
        const float2 v = s >= 0 ? reinterpret_cast<const float2 *>(in)[s] : reinterpret_cast<const float2 *>(hist)[s + C::HIST];
        reinterpret_cast<float2 *>(a.hist_out + 2 * a.hist_stride * ch)[lane] = v;
    }

    // HBM -> LDS for one tile: samples [q0 - 32, q0 + 256).  The first and the last 16 chunks are shared with the
    // neighbouring tiles (plain loads), the 112 in between are this tile's alone (instruction 1 entirely: non-temporal).
    // 32-bit tile bounds instead of 64-bit sample arithmetic per tile: tiles below n_full have all their TILE_IN inputs
    // (and store all their outputs), of those all but tile 0 have their history inside this call's input
    static_assert(C::TILE_IN >= C::HIST && (C::TILE_IN & (C::TILE_IN - 1)) == 0, "tile 0 is the only one that reaches into the history");
    const int n_full = (int)(a.n_in / C::TILE_IN);
    auto stage = [&](int item) __attribute__((always_inline)) {
        const int tile = PBSPLIT ? item / NPBT : item;
        const long long q0 = (long long)tile * C::TILE_IN;
        const bool interior = tile >= 1 && tile < n_full;
        if (interior) {
#pragma unroll
            for (int i = 0; i < C::NLOAD; ++i) {
                unsigned cc = 64 * i + lane;
                cc = cc < (unsigned)C::CHUNKS ? cc : (unsigned)C::CHUNKS - 1u;
                asm volatile("" : "+v"(cc));
                const char *src = reinterpret_cast<const char *>(in + 2 * (q0 - 32)) + 16u * cc;
                // chunks 16 .. CHUNKS - 17 are this tile's alone: with QI = 4 that is all of instruction 1 (non-temporal)
                if (QI == 4 && i == 1) glds16<2>(src, lds + 64 * i);
                else glds16(src, lds + 64 * i);
            }
        } else {
            // edge tiles (first / last of a call): through registers and plain LDS writes; stage() then returns false
            // and the caller waits with vmcnt(0) instead of the counted form, which presumes NLOAD DMA instructions
#pragma unroll
            for (int i = 0; i < C::NLOAD; ++i) {
                unsigned cc = 64 * i + lane;
                cc = cc < (unsigned)C::CHUNKS ? cc : (unsigned)C::CHUNKS - 1u;
                asm volatile("" : "+v"(cc));
                const long long s = q0 - 32 + 2 * (long long)cc;
                float2 v0, v1;
                const long long last = a.n_in - 1;
                if (s >= 0) v0 = reinterpret_cast<const float2 *>(in)[s <= last ? s : last];
                else v0 = reinterpret_cast<const float2 *>(hist)[s + C::HIST];
                if (s + 1 >= 0) v1 = reinterpret_cast<const float2 *>(in)[s + 1 <= last ? s + 1 : last];
                else v1 = reinterpret_cast<const float2 *>(hist)[s + 1 + C::HIST];
                lds[64 * i + lane] = (f32x4){v0.x, v0.y, v1.x, v1.y};
            }
        }
        return interior;
    };

    // the keying threshold, read from the kernel arguments where it is used (no scalar register held across the tile loop)
    auto tx_threshold = [&]() __attribute__((always_inline)) {
        const __attribute__((address_space(4))) InterpTileArgs *ap =
            (const __attribute__((address_space(4))) InterpTileArgs *)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ap));
        return ap->thr2;
    };

    // Lane constants of the output path (round 6: 128 of the ~1200 VALU instructions per block of 1024 FMAs of the x16 .. x96 instances were
    // address arithmetic; the x4 / x8 instances carried the same per tile of 512 / 256 FMAs).  The transposition slot of chunk k = (L / 2) qi
    // + 2 c + e of lane g is CPL g + (k ^ (g & (CPL - 1))); qi, c, e occupy disjoint bits of k, so it is (((L / 2) qi + e) ^ (g & (CPL - 1))) ^
    // 2 c: 2 QI byte offsets per lane once, one XOR with 32 c per pass pair.  The store of slot 64 i + lane goes to byte STORE_STRIDE i +
    // voff[i % NV] of the tile's (block's) output: g2 = (64 i + lane) / CPL, k2 = ((64 i + lane) % CPL) ^ (g2 % CPL).  CPL = 16 (two inputs
    // x sixteen outputs; the profiling build's four inputs x eight): g2 = 4 i + lane / 16, k2 = kb ^ 4 (i & 3) with kb = (lane & 15) ^ (lane
    // >> 4), sample (2 g2 + k2 / 8) LTE + 2 (k2 % 8) of the block -- four offsets; CPL = 8 (x8: two inputs x eight outputs; x4: four x four):
    // g2 = 8 i + lane / 8, k2 = (lane & 7) ^ (lane >> 3), byte 128 g2 + 16 k2 -- one.
    // (x4 keeps the compiler's addressing: nothing to gain there -- 0.505 against 0.506 ms, same box -- and its keyed instance answered the
    // constants with 56 more scalar spills)
    constexpr bool FASTOUT = LL != 4;
    constexpr int LTE = NPBT == 1 ? 16 : LT;                            // CPL = 16: output samples between a lane's two inputs' rows
    constexpr int NV = C::CPL == 16 ? 4 : 1;
    constexpr int STORE_STRIDE = C::CPL == 16 ? 64 * LTE : 1024;        // bytes from store i to store i + 1
    static_assert(C::CPL == 16 || C::CPL == 8, "output chunks per lane");
    unsigned wslot[QI][2], voff[NV];
    if constexpr (FASTOUT) {
#pragma unroll
        for (int qi = 0; qi < QI; ++qi)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                wslot[qi][e] = 16u * (unsigned)(C::CPL * lane + (((C::L / 2) * qi + e) ^ (lane & (C::CPL - 1))));       // in bytes
                asm volatile("" : "+v"(wslot[qi][e]));
            }
        if constexpr (C::CPL == 16) {
            const unsigned kb = (unsigned)((lane & 15) ^ (lane >> 4));
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const unsigned k2 = kb ^ (unsigned)(4 * t);
                voff[t] = (unsigned)(8 * LTE) * (2u * (unsigned)(lane >> 4) + (k2 >> 3)) + 16u * (k2 & 7u);
                asm volatile("" : "+v"(voff[t]));
            }
        } else {
            voff[0] = 128u * (unsigned)(lane >> 3) + 16u * (unsigned)((lane & 7) ^ (lane >> 3));
            asm volatile("" : "+v"(voff[0]));
        }
    }

    int tile = first_tile;
    if (tile >= a.n_tiles) return;
    stage(tile);
    bool counted = false;                                       // the staged tile's DMAs sit in front of CPL stores
    bool landed = false;                                        // (phase blocks) ... and have been awaited already
    while (true) {
        // s_waitcnt vmcnt counts loads and stores together, in issue order: with the next tile's three DMAs issued
        // BEFORE this tile's sixteen stores, "at most 16 outstanding" means the DMAs have landed
        if constexpr (NPB == 1) {
            if (COUNTED && counted) {
                static_assert(C::CPL == 16 || C::CPL == 8, "stores per tile behind the next tile's DMAs");
                if constexpr (C::CPL == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        } else {
            // (phase blocks: awaited behind the previous tile's first block; the workgroup's first tile here)
            if (!landed) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const int rtile = PBSPLIT ? tile / NPBT : tile;             // (PBSPLIT: `tile` counts items)
        const int pb0 = PBSPLIT ? tile % NPBT : 0;
        const long long q0 = (long long)rtile * C::TILE_IN;

        // ---- the lane's window, once: samples q0 + QI g - 32 .. q0 + QI g + QI - 1  ->  image chunks (QI/2) g .. + NWU - 1
        // (pass p uses chunks 8 - 8p + t, t = 0 .. NW - 1, of these)
        f32x4 win[C::NWU];
        {
            const f32x4 *wp = lds + (QI / 2) * lane;
#pragma unroll
            for (int t = 0; t < C::NWU; ++t) win[t] = wp[t];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (KEYED) {
            if (ch == 0 && pb0 == 0) {                                 // (PBSPLIT: a tile's inputs are counted by its first item)
                const __attribute__((address_space(4))) InterpTileArgs *ap =
                    (const __attribute__((address_space(4))) InterpTileArgs *)__builtin_amdgcn_kernarg_segment_ptr();
                asm volatile("" : "+s"(ap));
                const long long key_lo = ap->key_lo, key_hi = ap->key_hi;      // key_hi <= n_in (checked by sxfir_interpolate_keyed)
                if (q0 < key_hi && q0 + C::TILE_IN > key_lo) {
                    const float thr2 = ap->thr2;
                    unsigned total = 0;
#pragma unroll
                    for (int e = 0; e < QI; ++e) {
                        // sample q0 + QI lane + e = window sample 32 + e: half (e & 1) of chunk 16 + e / 2
                        const f32x4 v = win[16 + e / 2];
                        const float fi = (e & 1) ? v.z : v.x, fq = (e & 1) ? v.w : v.y;
                        const long long sidx = q0 + QI * lane + e;
                        // the rule as the reference's source states it: both products and the sum rounded once each
                        const bool k = sidx >= key_lo && sidx < key_hi &&
                                       __fadd_rn(__fmul_rn(fi, fi), __fmul_rn(fq, fq)) >= thr2;
                        total += (unsigned)__builtin_popcountll(__ballot(k));
                    }
                    if (lane == 0 && total) atomicAdd(ap->key_counter, (unsigned long long)total);
                }
            }
        }
        // the image is free: fetch the next tile behind the arithmetic of this one
        int next = tile + a.n_groups;
        if constexpr (PLAIN_TAIL) {
            const int r = tile / G + 1;                          // the round after this tile's
            next = r < R ? r * G + first_tile : (r == R && (int)blockIdx.x < rem ? R * G + (int)blockIdx.x : a.n_tiles);
        }
        counted = false;
        if (next < a.n_tiles) counted = stage(next);

#pragma unroll 1
        for (int pb = pb0; pb < pb0 + NPB; ++pb) {
        // ---- four passes (x4: two, x16: eight): phase group c (outer), row half p (inner)
#pragma unroll 1
        for (int c = 0; c < C::L / 4; ++c) {
            f32x2 y[QI][4];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                f32x2 hs[32];
#pragma unroll
                for (int m = 0; m < 32; ++m) hs[m] = tq[(16 * C::L) * pb + 32 * (2 * c + p) + m];
                f32x4 wv[C::NW];
#pragma unroll
                for (int t = 0; t < C::NW; ++t) wv[t] = win[8 - 8 * p + t];
                f32x2 acc[QI][4];                               // every chain's first FMA (tap row 15) writes it
                interp_pass_steps<QI>(std::make_integer_sequence<int, C::NW>{}, wv, hs, acc);
#pragma unroll
                for (int qi = 0; qi < QI; ++qi)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        if (p == 0) y[qi][rr] = acc[qi][rr];
                        else y[qi][rr] = (f32x2){__fadd_rn(y[qi][rr].x, acc[qi][rr].x), __fadd_rn(y[qi][rr].y, acc[qi][rr].y)};
                    }
            }
            // phases 4c..4c+3 of the lane's inputs: chunks k = (L / 2) qi + 2c + {0, 1} of its CPL
#pragma unroll
            for (int qi = 0; qi < QI; ++qi) {
                if constexpr (FASTOUT) {
                    char *ob8 = reinterpret_cast<char *>(obuf);
                    *reinterpret_cast<f32x4 *>(ob8 + (wslot[qi][0] ^ (unsigned)(32 * c))) = (f32x4){y[qi][0].x, y[qi][0].y, y[qi][1].x, y[qi][1].y};
                    *reinterpret_cast<f32x4 *>(ob8 + (wslot[qi][1] ^ (unsigned)(32 * c))) = (f32x4){y[qi][2].x, y[qi][2].y, y[qi][3].x, y[qi][3].y};
                } else {
                    const int k = (C::L / 2) * qi + 2 * c;
                    obuf[C::CPL * lane + (k ^ (lane & (C::CPL - 1)))] = (f32x4){y[qi][0].x, y[qi][0].y, y[qi][1].x, y[qi][1].y};
                    obuf[C::CPL * lane + ((k + 1) ^ (lane & (C::CPL - 1)))] = (f32x4){y[qi][2].x, y[qi][2].y, y[qi][3].x, y[qi][3].y};
                }
            }
        }

        // ---- store: instruction i moves slots 64i .. 64i+63 = the rows of lanes 4i .. 4i+3, each lane the chunk its
        // slot holds (a permutation inside the 256-byte row): four whole rows per instruction.  Always sixteen store
        // instructions per tile (lanes past the end of the call sit theirs out): the counted wait relies on it.
        const long long o0 = q0 * LT + C::L * pb;               // first output sample of the tile (of this phase block)
        // chunk k2 of lane g2's row: input k2 / (L / 2), chunk k2 % (L / 2) of its L outputs of this block
        auto out_sample = [&](int g2, int k2) __attribute__((always_inline)) {
            return NPBT == 1 ? o0 + 2 * (C::CPL * g2 + k2) : o0 + (long long)(QI * g2 + k2 / (C::L / 2)) * LT + 2 * (k2 % (C::L / 2));
        };
        if (rtile < n_full) {
            // all CPL reads in flight, then CPL stores back to back (decided per wave, not per lane and store)
            f32x4 v[C::CPL];
#pragma unroll
            for (int i = 0; i < C::CPL; ++i) v[i] = obuf[64 * i + lane];
            if constexpr (S32OUT) {
                const float thr2 = tx_threshold();
#pragma unroll
                for (int i = 0; i < C::CPL; ++i) {
                    const int2 w0 = tx_words(v[i].x, v[i].y, thr2), w1 = tx_words(v[i].z, v[i].w, thr2);
                    v[i] = (f32x4){__int_as_float(w0.x), __int_as_float(w0.y), __int_as_float(w1.x), __int_as_float(w1.y)};
                }
            }
            if constexpr (FASTOUT) {
                // a scalar base per store (the tile's / block's first output byte + STORE_STRIDE i) and one of the lane's constant offsets
                // (written out: the compiler turns base + zext(offset) into 64-bit vector adds, two VALU instructions per store)
                const unsigned long long ob = (unsigned long long)(uintptr_t)(out + 2 * o0);
#pragma unroll
                for (int i = 0; i < C::CPL; ++i) {
                    const unsigned long long bi = ob + (unsigned long long)STORE_STRIDE * i;
                    asm volatile("global_store_dwordx4 %0, %1, %2 nt" :: "v"(voff[i % NV]), "v"(v[i]), "s"(bi) : "memory");
                }
            } else {
#pragma unroll
                for (int i = 0; i < C::CPL; ++i) {
                    const int slot = 64 * i + lane;
                    const int g2 = slot / C::CPL, k2 = (slot & (C::CPL - 1)) ^ (g2 & (C::CPL - 1));
                    __builtin_nontemporal_store(v[i], reinterpret_cast<f32x4 *>(out + 2 * out_sample(g2, k2)));
                }
            }
        } else {
            // the call's last tile (no counted wait follows it: the wave ends here)
            const long long o_end = a.n_in * LT;
#pragma unroll
            for (int i = 0; i < C::CPL; ++i) {
                const int slot = 64 * i + lane;
                const int g2 = slot / C::CPL, k2 = (slot & (C::CPL - 1)) ^ (g2 & (C::CPL - 1));
                f32x4 v = obuf[slot];
                if constexpr (S32OUT) {
                    const float thr2 = tx_threshold();
                    const int2 w0 = tx_words(v.x, v.y, thr2), w1 = tx_words(v.z, v.w, thr2);
                    v = (f32x4){__int_as_float(w0.x), __int_as_float(w0.y), __int_as_float(w1.x), __int_as_float(w1.y)};
                }
                const long long o = out_sample(g2, k2);              // two output samples per chunk
                if (o + 2 <= o_end) __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(out + 2 * o));
            }
        }
        // the next block's / tile's output writes reuse the buffer only after these reads have returned
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (NPB > 1) {
            if (pb == 0) {
                // the next tile's DMAs were issued before this block's CPL stores: at most CPL outstanding = they have landed
                if (COUNTED && counted) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                landed = true;
            }
        }
        }   // pb
        if (next >= a.n_tiles) break;
        tile = next;
    }
}

}  // namespace sxfir
