// Decimate-by-48 and -by-96, 32 taps per phase (1536 / 3072 taps), CF32, S32 wire words or CF16 storage: the two slowest rates of the
// reference's table (SoapySX.cpp:180-208: master clock / 768 and / 1536; the device's converters run at master clock / 16,
// so the ratio is divider / 16).  New code: the reference programs the SX1255's own decimator for these rates (:1192-1208).
//
// A row of the polyphase picture is D = 48 or 96 samples: a tile of outputs with its 31 halo rows no longer fits LDS the way
// decim_dense_kernel<32> holds it (543 rows x 96 samples = 417 KB), and 1536 taps fit no register file.  But the numeric
// contract (2, 4) never mixes columns before its tree: a (row half p, column group c) subset is a 64-tap fmaf chain over FOUR
// columns of the picture, and what meets in the tree are finished partial sums.  So the picture is cut into BLOCKS of sixteen
// columns -- four column groups, eight subsets -- and a (tile, block) step is the scalar-tap problem of
// decim_dense_kernel<8, SUBSET> with rows that lie D samples apart in HBM:
//
//   * The contract's ROTATION (sxfir_contract_rotation = 1 for these two shapes):
//     slot k' = j D + r of the picture holds tap (k' + 1) mod NT, so row j of output m is the D samples that end BEFORE
//     sample (m - j) D and a block's piece of a row is one whole, aligned 128-byte line of the input.  Unrotated, a row ends AT
//     sample (m - j) D, every piece straddles two lines, and each line is fetched by two steps: the first build of this
//     kernel (eight-column blocks, unrotated) moved 3 x the algorithmic bytes through the L2's memory side and took 1.19 ms
//     per 2^28 samples; the same with aligned 64-byte pieces 2 x and 0.77 ms (a line's two halves in two steps).  The
//     filter is the same sum; tap 0 -- the newest sample, alone at the start of the NEXT line -- opens the chain of the last
//     subset (slot NT - 1) instead of closing the chain of the first: one FMA per output with an operand from 32 rows on.
//   * The image of a step: 544 rows (31 halo rows, 512 outputs' rows, one more for tap 0) x 16 samples, linear, one pad slot
//     per 8 rows: 70 704 bytes, two workgroups per CU.  A DMA instruction moves eight whole lines (lane l: row l / 8,
//     chunk l % 8); 17 per wave and step.  Lane g owns outputs 8 g .. 8 g + 7 of the tile: its window starts 65 slots after
//     its neighbour's, so the sixteen lanes a ds_read_b128 is served with hit sixteen different slots mod 16.
//   * The eight subsets of a block go to the four waves in two passes (round 6: wave ww = column group ww, row half 1 then row
//     half 0, see RP below; round 5: row half p = ww & 1, column groups ww / 2 and ww / 2 + 2); a pass's 64 taps are 32 SGPR pairs loaded with four s_load_dwordx16 from a block-major table of the
//     rotated taps (block b at 512 b, subset 2 c + p at 64 (2 c + p), (jj, rr) at 4 jj + rr): every FMA has its tap as the
//     scalar operand, 46 window reads per 512 packed FMAs.
//   * The subsets' partials meet through the dead image in the contract's order ((p0 + p1) per column group, adjacent
//     groups, then the two pairs): the block's value, level 2 of the tree over the D / 4 column groups.  The block values of a
//     tile meet in registers the way a binary counter adds: (B0 + B1) + B2 for three blocks, ((B0 + B1) + (B2 + B3)) +
//     (B4 + B5) for six -- the adjacent-pair tree in which an odd element at the end of a level moves up unchanged, as
//     the generic kernel states it.
//   * Steps run (tile, block 0), (tile, block 1), ...; the next step's DMAs are issued when the partials have been
//     exchanged, behind the arithmetic of the CU's other workgroup.
//
// Edge tiles (first / last of a call), the fused history carry-over and the ragged last stores follow the dense kernel.
//
// SPLIT (round 6): calls the API really issues are small -- readStream / writeStream blocks are 256 .. 8192 samples
// (SoapySX.cpp:868-1105; the Device's chains batch them into passes of 2^17 .. 2^24 wideband samples) -- and a tile here is
// 512 outputs x 48 / 96 inputs: a 2^22-sample call at /96 is 86 tiles for 512 workgroup slots, each walking its six blocks one
// after the other.  The SPLIT instance deals (tile, block) ITEMS instead: one workgroup per item, NB times the workgroups, one
// step each.  The block values of a tile then meet through HBM: every item writes its block value (4 KiB) to the plan's
// scratch with device-scope stores and counts itself in; the item that finds NB - 1 others before it reads all NB values back
// (device-scope loads) and adds them in the contract's order -- (B0 + B1) + B2, ((B0 + B1) + (B2 + B3)) + (B4 + B5): the
// order is fixed by the block numbers, not by who arrives when, so the bits are those of the walking form.  sxfir_launch
// chooses SPLIT while a call has at most eight times as many tiles as the chip has slots (beyond that the walking form's rounds are
// as full, and it keeps a tile's blocks on one XCD).
#pragma once

#include <type_traits>

#include "sxfir_decim_dense.hip.h"

namespace sxfir {

struct DecimBlocks16 {
    static constexpr int W = 4;                       // waves per workgroup
    static constexpr int TILE_OUT = 512;
    static constexpr int NROWS = TILE_OUT + 31 + 1;   // aligned rows q in [M0 - 31, M0 + TILE_OUT]; the last for tap 0 alone
    static constexpr int CPR = 8;                     // 16-byte chunks per row of a block
    static constexpr int CH = NROWS * CPR;
    static constexpr int RPI = 64 / CPR;              // rows per DMA instruction
    static constexpr int PADROWS = 8;
    static constexpr int NI = CH / 64;                // 68 DMA instructions per step
    static constexpr int NIW = NI / W;                // 17 per wave
    static constexpr int LDS_SLOTS = CH + (NROWS - 1) / PADROWS;
    static constexpr int WCH = 46;                    // window chunks per lane and pass: 23 rows x 2
    static constexpr int TAP0_SLOTS = 32 * CPR + 32 / PADROWS;   // tap 0's sample: the same chunk 32 rows on
    static_assert(CH % 64 == 0 && NI % W == 0 && PADROWS == RPI, "whole instructions, a pad after each");
    static_assert(LDS_SLOTS * 16 <= 160 * 1024 / 2, "two workgroups per CU");
    static constexpr int dma_slot(int i) { return 65 * i; }
};

// NB = blocks per row (3: /48, 6: /96).  NTLD: the lines no other tile reads (image rows 32 .. 511) as non-temporal loads.
// HALFIN: CF16 storage (IQ as half pairs in HBM, fp32 arithmetic, outputs rounded to half once) through the typed LDS-DMA front
// end of decim_dense_kernel<.., HALFIN>: the image is the same CF32 image, the texture path converts on the way in.  One
// buffer_load_format_x ... lds moves two rows of the block (lanes 0-31: the 32 halves of row r, lanes 32-63: of row r + 1; 64
// floats = 16 slots): four per 128-byte-line instruction of the CF32 form.  A block's piece of a row is 64 bytes here -- half a
// line, the other half the neighbouring block's -- so each line crosses the L2's memory side twice: the bytes of the CF32 form,
// not half of them (the kernel is arithmetic-bound at either).
struct DecimBlocksJoin {
    f32x4 *partials;        // [channel][tile][block][256 lanes]: a block value, two outputs per lane
    unsigned *arrived;      // [channel][tile]: items of the tile that have written theirs; the last one resets it to 0
};

// RP (round 6; what ships -- RP = false is round 5's form, the A/B partner in the profiling build, SXFIR_BLOCKS_RP=0): the eight subsets
// of a block dealt to the waves by COLUMN GROUP -- wave c runs (p = 1, c) then (p = 0, c) -- instead of by row half.  The two windows
// are then the same two chunk columns of rows 8G .. 8G + 22 and 8G + 16 .. 8G + 38: the seven rows they share stay in registers (78
// window reads per step instead of 92), and P0 + P1 -- the first level of the contract's tree -- is an in-lane add, so four partials
// cross the exchange instead of eight.  Priced from the measured list (a ds_read_b128 = 1.65-2.2 packed FMAs, a ds_write_b128 0.56:
// profiles/round4z9_price_list.txt) at 2.8 % of a step's energy; measured on the same box, three alternations, 2^28 samples:
// /48 486 -> 467-474 us, /96 492 -> 470-473 us (-3.6 %, -4.2 %; -2 % at 2^24; profiles/round6_blocks_rp_ab.txt).
// ROT = false (profiling experiment, /16 and /32 only: rows that are contiguous in HBM): the UNROTATED contract -- the image starts one
// sample later (a block's piece is then a 128-byte stretch that starts 8 bytes into a line), slot k holds tap k, no special tap 0.
template <int NB, bool S32IN = false, bool NTLD = false, bool HALFIN = false, bool SPLIT = false, bool RP = false, bool ROT = true>
__global__ __launch_bounds__(256) void decim_blocks_kernel(const DecimMultiArgs a, const DecimBlocksJoin jn)
{
    static_assert(ROT || ((NB == 1 || NB == 2) && !HALFIN), "unrotated: contiguous rows only");
    static_assert(NB == 3 || NB == 6 || ((NB == 1 || NB == 2) && !SPLIT), "ratios 48 and 96 (profiling experiment: 16 and 32, walking form)");
    static_assert(!(HALFIN && S32IN), "one storage format");
    using C = DecimBlocks16;
    constexpr int D = 16 * NB;
    constexpr int NT = 32 * D;
    constexpr int SB = HALFIN ? 4 : 8;                  // bytes per complex sample in HBM
    __shared__ __attribute__((aligned(16))) f32x4 lds[C::LDS_SLOTS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int ww = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = RP ? 1 : (ww & 1), c0 = RP ? ww : (ww >> 1);   // the wave's subsets: (p, c0) and (p, c0 + 2); RP: (1, ww) and (0, ww)
    const int G = lane;                                 // output group of the lane
    const int ch = blockIdx.y;

    const char *in = reinterpret_cast<const char *>(a.in) + (long long)SB * a.in_stride * ch;
    char *out = reinterpret_cast<char *>(a.out) + (long long)SB * a.out_stride * ch;
    const __attribute__((address_space(4))) f32x2 *tq = (const __attribute__((address_space(4))) f32x2 *)a.taps;

    // Window of pass A: rows 8u .. 8u + 22 of the image (u = G + 2 - 2p), chunks CPR - 2 - 2 c0 + {0, 1} of each; slot = chunk
    // + (pads before its row); the window's 8-row segments each lie between two pads.  Pass B: four chunks lower.
    const int u = G - 2 * p + 2;
    const f32x4 *win0 = lds + (C::CPR * 8 * u + C::CPR - 2 - 2 * c0 + u);
    const f32x4 *win1 = win0 + 1;
    const f32x4 *win2 = win0 + 2;

    const int NG = a.n_groups;
    // tiles [1, tile_hi] lie wholly inside this call's input (tile 0 reaches into the history), tiles below n_full store
    // all 512 outputs
    const long long q_hi = a.n_in / D - C::TILE_OUT;
    const int tile_hi = q_hi < 0 ? -1 : (int)(q_hi >> 9);
    const int n_full = (int)(a.n_out >> 9);
    // Tile schedule: in round r the NG workgroups cover tiles [r NG, (r + 1) NG), the workgroups of one XCD (blockIdx % 8
    // shares an XCD; speed only) a contiguous block of the round, so that a tile's halo is found in that XCD's L2.  The LAST,
    // partial round is dealt plainly (tile = R NG + blockIdx): XCD-blocked, its tiles would all fall to the first XCDs, and
    // with tiles this large -- few per workgroup -- those XCDs would carry up to twice the work (measured: /96 at 2^28
    // samples, 5462 tiles on 4096 workgroups, 0.67 ms against 0.53 ms balanced).
    const int perm = (NG % 8 == 0) ? (int)(blockIdx.x % 8) * (NG / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    const int R = a.n_tiles / NG, rem = a.n_tiles % NG;
    const int n_rounds = SPLIT ? 1 : R + ((int)blockIdx.x < rem ? 1 : 0);
    auto tile_of = [&](int r) __attribute__((always_inline)) { return r < R ? r * NG + perm : R * NG + (int)blockIdx.x; };
    // SPLIT: NG = n_tiles * NB items, one per workgroup, in dispatch order (workgroups start in blockIdx order; HIP does not
    // promise it, and nothing but speed depends on it): FIRST the items of the call's edge tiles -- the last tile and tile 0,
    // whose images go through registers sample by sample and take the longest -- then the interior tiles block-major.  (With
    // the edge items last, the four workgroups that did not fit the first round of a 2^22-sample call at /96 -- 516 items
    // on 512 slots -- were exactly the slow ones: 22 us against 15 for one round.)
    int split_tile = 0, split_blk = 0;
    if constexpr (SPLIT) {
        const int bx = (int)blockIdx.x, ne = a.n_tiles < 2 ? a.n_tiles : 2;
        if (bx < ne * NB) {
            split_blk = bx / ne;
            split_tile = (bx % ne) ? 0 : a.n_tiles - 1;
        } else {
            const int j = bx - ne * NB, ni = a.n_tiles - ne;
            split_blk = j / ni;
            split_tile = 1 + j % ni;
        }
    }
    // fused history carry-over (by the owner of the call's last tile): the tail of (hist ++ in) becomes the next history
    if constexpr (SPLIT) {
        // (a small call's time is its slowest item's: the copy goes to the LAST workgroup -- an interior item wherever the call
        // has one, not one of the edge items in front -- on all four waves, a thread's 2 NB loads in flight before its first store;
        // as a loop of one wave it put 24 / 48 dependent round trips in front of an item's own work)
        if ((int)blockIdx.x == NG - 1) {
            char *ho = reinterpret_cast<char *>(a.hist_out) + (long long)SB * a.hist_stride * ch;
            const char *hi = reinterpret_cast<const char *>(a.hist) + (long long)SB * a.hist_stride * ch;
            constexpr int PER = NT / 256;
            float2 hv[PER];
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int j = tid + 256 * k;
                const long long s = a.n_in - NT + j;
                const char *src = s >= 0 ? in + SB * s : hi + SB * (s + NT);
                if constexpr (HALFIN) hv[k].x = __uint_as_float(*reinterpret_cast<const unsigned *>(src));
                else hv[k] = *reinterpret_cast<const float2 *>(src);
            }
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int j = tid + 256 * k;
                if constexpr (HALFIN) reinterpret_cast<unsigned *>(ho)[j] = __float_as_uint(hv[k].x);
                else reinterpret_cast<float2 *>(ho)[j] = hv[k];
            }
        }
    } else if ((rem ? (int)blockIdx.x == rem - 1 : perm == NG - 1) && ww == C::W - 1) {
        char *ho = reinterpret_cast<char *>(a.hist_out) + (long long)SB * a.hist_stride * ch;
        const char *hi = reinterpret_cast<const char *>(a.hist) + (long long)SB * a.hist_stride * ch;
        for (int j = lane; j < NT; j += 64) {
            const long long s = a.n_in - NT + j;
            const char *src = s >= 0 ? in + SB * s : hi + SB * (s + NT);
            if constexpr (HALFIN) reinterpret_cast<unsigned *>(ho)[j] = *reinterpret_cast<const unsigned *>(src);
            else reinterpret_cast<float2 *>(ho)[j] = *reinterpret_cast<const float2 *>(src);
        }
    }

    auto rare_args = [&]() __attribute__((always_inline)) {
        const __attribute__((address_space(4))) DecimMultiArgs *ap =
            (const __attribute__((address_space(4))) DecimMultiArgs *)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ap));
        return ap;
    };

    // HBM -> LDS for one (tile, block) step: instruction i = ww + 4 i0 moves image rows [8 i, 8 i + 8) -- eight lines, D
    // samples apart -- to the slots from 65 i on.  Block b's line of aligned row q is samples D q - 16 (b + 1) .. D q - 16 b - 1.
    const unsigned lane_off = (unsigned)(8 * D) * (unsigned)(lane >> 3) + 16u * (unsigned)(lane & 7);
    // (HALFIN) LDS byte address of the wave's first staging slot, a scalar: M0 of the typed DMA = this + a constant
    const unsigned lds_wave_base = HALFIN ? __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds + 65 * ww)) : 0u;
    auto stage = [&](int tile, int blk) __attribute__((always_inline)) {
        const long long M0 = (long long)tile * C::TILE_OUT;
        const long long s_first = D * (M0 - 31) - 16 * (blk + 1) + (ROT ? 0 : 1);   // first sample of the block image
        // (unrotated: the image ends one sample later, so the last tile that lies wholly inside the rotated image does not)
        const bool interior = tile >= 1 && (ROT ? tile <= tile_hi : tile < tile_hi);
        if constexpr (HALFIN) {
            if (interior) {
                // the descriptor is based at the wave's first byte of THIS step (64-bit base, rebuilt per step from scalars), so
                // every offset is a small constant and a call may be as long as it likes; {DATA_FORMAT 16, NUM_FORMAT FLOAT, X <- R}
                const unsigned long long wb = (unsigned long long)(in + SB * s_first + (SB * D * C::RPI) * ww);
                v4i32 rs;
                rs.x = __builtin_amdgcn_readfirstlane((int)(unsigned)wb);
                rs.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(wb >> 32)) & 0xffff;    // stride 0
                rs.z = 1 << 20;                                                               // bytes addressable from the base
                rs.w = 4 | (7 << 12) | (2 << 15);
                // lanes 0-31: the 32 halves of a row's piece, lanes 32-63: of the next row's
                unsigned voff = (unsigned)(SB * D) * (unsigned)(lane >> 5) + 2u * (unsigned)(lane & 31);
                asm volatile("" : "+v"(voff));
#pragma unroll
                for (int i0 = 0; i0 < C::NIW; ++i0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const unsigned soff = (unsigned)(SB * D * C::RPI * 4) * i0 + (unsigned)(2 * SB * D) * j;   // bytes from the wave's base
                        static_assert((SB * D * C::RPI * 4) * (C::NIW - 1) + (2 * SB * D) * 3 + SB * D + 64 < (1 << 20), "inside the descriptor");
                        const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_wave_base + 16u * (unsigned)(C::dma_slot(4 * i0) + 16 * j));
                        // (M0 is a reserved register to the compiler: writing it here needs, and admits, no clobber entry)
                        if (NTLD && i0 >= 1 && i0 < 16)
                            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_format_x %1, %2, %3 offen nt lds"
                                         :: "s"(m0v), "v"(voff), "s"(rs), "s"(soff) : "memory");
                        else
                            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_format_x %1, %2, %3 offen lds"
                                         :: "s"(m0v), "v"(voff), "s"(rs), "s"(soff) : "memory");
                    }
                }
                return;
            }
        }
        // (structural, not left to the optimizer: an instance whose typed front end writes M0 from inline asm holds no
        // compiler-managed LDS-DMA at all -- LLVM may hoist or merge ITS M0 set-up across an asm statement it cannot see into)
        if constexpr (!HALFIN) if (interior) {
            const char *base = in + 8 * s_first + (8 * D * C::RPI) * ww;
#pragma unroll
            for (int i0 = 0; i0 < C::NIW; ++i0) {
                unsigned lo = lane_off;
                asm volatile("" : "+v"(lo));          // a 32-bit offset next to its use: SGPR base + VGPR offset form
                const char *bi = base + (8 * D * C::RPI * 4) * i0;
                asm volatile("" : "+s"(bi));
                // image rows 32 .. 511 (instructions 4 .. 63) are this tile's alone; the others are a neighbour's halo too
                if (NTLD && i0 >= 1 && i0 < 16) glds16<2>(bi + lo, lds + (65 * ww + C::dma_slot(4 * i0)));
                else glds16(bi + lo, lds + (65 * ww + C::dma_slot(4 * i0)));
            }
            return;
        }
        {
            // edge tiles (first / last of a call): through registers, sample by sample
            const auto *ap = rare_args();
            const long long last = ap->n_in - 1;
            const char *hist = reinterpret_cast<const char *>(ap->hist) + (long long)SB * ap->hist_stride * ch;
            // SPLIT (a small call's latency IS its edge items'): every load of the step in flight before the first LDS write --
            // 34 (17 with half pairs: 8 bytes hold both samples) independent loads instead of 17 round trips to memory one
            // after the other; the walking form keeps the loop (an edge tile is one of thousands there, registers matter more)
            auto edge_chunk = [&](int i0, unsigned (&wds)[4]) __attribute__((always_inline)) {
                const int i = ww + 4 * i0;
                const int cidx = 64 * i + lane;                          // chunk of the image: row cidx / 8, chunk cidx % 8
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const long long s = s_first + (long long)D * (cidx >> 3) + 2 * (cidx & 7) + e;
                    const char *src = s >= 0 ? in + SB * (s <= last ? s : last) : hist + SB * (s + NT >= 0 ? s + NT : 0);
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
            };
            if constexpr (SPLIT) {
                unsigned wall[C::NIW][4];
#pragma unroll
                for (int i0 = 0; i0 < C::NIW; ++i0) edge_chunk(i0, wall[i0]);
#pragma unroll
                for (int i0 = 0; i0 < C::NIW; ++i0)
                    lds[C::dma_slot(ww + 4 * i0) + lane] = (f32x4){__uint_as_float(wall[i0][0]), __uint_as_float(wall[i0][1]),
                                                                   __uint_as_float(wall[i0][2]), __uint_as_float(wall[i0][3])};
            } else {
#pragma nounroll
                for (int i0 = 0; i0 < C::NIW; ++i0) {
                    unsigned wds[4];
                    edge_chunk(i0, wds);
                    lds[C::dma_slot(ww + 4 * i0) + lane] = (f32x4){__uint_as_float(wds[0]), __uint_as_float(wds[1]), __uint_as_float(wds[2]),
                                                                   __uint_as_float(wds[3])};
                }
            }
        }
    };

    auto add4 = [](const f32x4 &l, const f32x4 &r) __attribute__((always_inline)) {
        return (f32x4){__fadd_rn(l.x, r.x), __fadd_rn(l.y, r.y), __fadd_rn(l.z, r.z), __fadd_rn(l.w, r.w)};
    };

    if (n_rounds == 0) return;
    int round = 0, tile = SPLIT ? split_tile : tile_of(0), blk = SPLIT ? split_blk : 0;
    stage(tile, blk);
    f32x4 lv0, lv1, lv2;                                // waiting block sums of 1, 2, 4 blocks
    lv0 = lv1 = lv2 = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    while (true) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own DMAs landed ...
        __syncthreads();                                    // ... and everybody else's

        // ---- two passes: window sample w meets output i at local tap kl = 4*i + 63 - w
        f32x2 acc[2][8];
        f32x4 keep[RP ? 14 : 1];                            // RP: rows 16 .. 22 of the first window = rows 0 .. 6 of the second
        // (the two passes as two instantiations of a generic lambda, not a loop of two: with every inner loop unrolled the body
        // is beyond clang's full-unroll budget even under #pragma unroll, and a rolled loop turns `ps` into a run-time value)
        auto one_pass = [&](auto ps_c) __attribute__((always_inline)) {
            constexpr int ps = decltype(ps_c)::value;
            f32x2 hs[32];
            const int sub = RP ? 2 * c0 + (1 - ps) : 2 * (c0 + 2 * ps) + p;      // the pass's subset 2c + p
            constexpr int TAP0_PASS = RP ? 0 : 1;           // the pass that runs (p = 1, c = 3) on wave 3
#pragma unroll
            for (int m = 0; m < 32; ++m) hs[m] = tq[256 * blk + 32 * sub + m];
            // slot NT - 1 = (last block, p = 1, c = 3, jj = 15, rr = 3) holds tap 0, whose sample is x[m D]: the same chunk of
            // the image 32 rows further on (the first FMA of the chain; wave-uniform)
            const bool tap0 = ROT && ps == TAP0_PASS && blk == NB - 1 && ww == 3;
            f32x2 xs[8];
            if (tap0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const f32x4 v = (RP ? win0 : win0 - 4)[C::CPR * i + C::TAP0_SLOTS];
                    xs[i] = __builtin_shufflevector(v, v, 0, 1);
                    if constexpr (S32IN) xs[i] = (f32x2){(float)__float_as_int(xs[i].x), (float)__float_as_int(xs[i].y)};
                }
            }
#pragma unroll
            for (int t = 0; t < C::WCH; ++t) {
                // (RP: the second window starts 16 rows = 128 chunks + 2 pads further on)
                const f32x4 *wp = (t < 16 ? win0 : (t < 32 ? win1 : win2)) + (RP ? 130 * ps : -4 * ps);
                f32x4 v;
                if (RP && ps == 1 && t < 14) v = keep[t];
                else v = wp[C::CPR * (t >> 1) + (t & 1)];
                if (RP && ps == 0 && t >= 32) keep[t - 32] = v;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int w = 2 * t + s;
                    f32x2 x = s ? __builtin_shufflevector(v, v, 2, 3) : __builtin_shufflevector(v, v, 0, 1);
                    if constexpr (S32IN) x = (f32x2){(float)__float_as_int(x.x), (float)__float_as_int(x.y)};
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int kl = 4 * i + 63 - w;
                        if (kl >= 0 && kl < 64) {
                            if (kl == 63) {
                                if (ps == TAP0_PASS) {
                                    const f32x2 xf = tap0 ? xs[i] : x;
                                    pk_fma_s_hi_first(acc[ps][i], hs[31], xf);
                                } else {
                                    pk_fma_s_hi_first(acc[ps][i], hs[31], x);
                                }
                            } else if (kl & 1) pk_fma_s_hi(acc[ps][i], hs[kl >> 1], x);
                            else pk_fma_s_lo(acc[ps][i], hs[kl >> 1], x);
                        }
                    }
                }
            }
        };
        one_pass(std::integral_constant<int, 0>{});
        one_pass(std::integral_constant<int, 1>{});
        __syncthreads();                                    // everyone is done reading this step's image
        // the eight subsets' partials meet in the dead image: subset s = 2c + p at 256 s; chunk k (two outputs) of group G at slot
        // 4G + (k ^ ((G >> 1) & 3)): the eight lanes a ds_write_b128 is served with hit eight different slots mod 16
        if constexpr (RP) {
            // (p0 + p1) of the wave's column group in the lane, then the four column groups' values: column group c at 256 c
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 p0 = (f32x4){acc[1][2 * k].x, acc[1][2 * k].y, acc[1][2 * k + 1].x, acc[1][2 * k + 1].y};
                const f32x4 p1 = (f32x4){acc[0][2 * k].x, acc[0][2 * k].y, acc[0][2 * k + 1].x, acc[0][2 * k + 1].y};
                lds[256 * c0 + 4 * G + (k ^ ((G >> 1) & 3))] = add4(p0, p1);
            }
        } else {
#pragma unroll
        for (int ps = 0; ps < 2; ++ps)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                lds[256 * (2 * (c0 + 2 * ps) + p) + 4 * G + (k ^ ((G >> 1) & 3))] =
                    (f32x4){acc[ps][2 * k].x, acc[ps][2 * k].y, acc[ps][2 * k + 1].x, acc[ps][2 * k + 1].y};
        }
        __syncthreads();
        f32x4 y;
        {
            f32x4 col[4];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
                col[cc] = RP ? lds[256 * cc + 64 * ww + lane] : add4(lds[256 * (2 * cc) + 64 * ww + lane], lds[256 * (2 * cc + 1) + 64 * ww + lane]);
            y = add4(add4(col[0], col[1]), add4(col[2], col[3]));
        }
        if constexpr (SPLIT) {
            // ---- this item's block value to HBM; the tile's last item to arrive adds the NB values in the contract's order
            // Device-scope (sc1) stores and loads, not fences: the eight XCDs' L2s are not coherent with each other, and a release /
            // acquire fence pair at device scope writes back and INVALIDATES the issuing XCD's whole L2 -- once per workgroup that
            // cost 150 ns per item and the halo lines every neighbour was about to reuse (first build: /96 at 2^24 samples 303 us
            // against the walking form's 66).  A device-scope store goes through to memory, a device-scope load comes from there;
            // the barrier's s_waitcnt vmcnt(0) sees the stores acknowledged before lane 0 counts the item in.
            // (the form is row 1 of the hand-off table in MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup
            // visibility": sc1 stores, every storing wave's vmcnt(0), a workgroup barrier, ONE lane's device-scope atomic add, the
            // workgroup whose add came last -- told by the value the add returned -- loads sc1 behind a barrier that lane joins)
            const long long slot = ((long long)ch * a.n_tiles + tile) * NB;
            {
                const f32x4 *q = jn.partials + (slot + blk) * 256 + tid;
                asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(q), "v"(y) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            unsigned *arrived = jn.arrived + ((long long)ch * a.n_tiles + tile);
            if (tid == 0) reinterpret_cast<unsigned *>(lds)[0] = __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (reinterpret_cast<const unsigned *>(lds)[0] != (unsigned)(NB - 1)) return;
            f32x4 bv[NB < 3 ? 3 : NB];
            if constexpr (NB >= 3) {
                const f32x4 *q = jn.partials + slot * 256 + tid + 256;  // block b at q + 256 (b - 1): 4096 (b - 1) bytes on
                static_assert(NB == 3 || NB == 6, "the load lists below");
                if constexpr (NB == 3)
                    asm volatile("global_load_dwordx4 %0, %3, off offset:-4096 sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\t"
                                 "global_load_dwordx4 %2, %4, off offset:-4096 sc1\n\ts_waitcnt vmcnt(0)"
                                 : "=&v"(bv[0]), "=&v"(bv[1]), "=&v"(bv[2]) : "v"(q), "v"(q + 512) : "memory");
                else
                    asm volatile("global_load_dwordx4 %0, %6, off offset:-4096 sc1\n\tglobal_load_dwordx4 %1, %6, off sc1\n\t"
                                 "global_load_dwordx4 %2, %7, off offset:-4096 sc1\n\tglobal_load_dwordx4 %3, %7, off sc1\n\t"
                                 "global_load_dwordx4 %4, %8, off offset:-4096 sc1\n\tglobal_load_dwordx4 %5, %8, off sc1\n\t"
                                 "s_waitcnt vmcnt(0)"
                                 : "=&v"(bv[0]), "=&v"(bv[1]), "=&v"(bv[2]), "=&v"(bv[3]), "=&v"(bv[4]), "=&v"(bv[5])
                                 : "v"(q), "v"(q + 512), "v"(q + 1024) : "memory");
            }
            if (tid == 0) __hip_atomic_store(arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the next launch starts from zero
            f32x4 r = add4(add4(bv[0], bv[1]), bv[2]);      // (B0 + B1) + B2
            if constexpr (NB == 6) r = add4(add4(add4(bv[0], bv[1]), add4(bv[2], bv[3])), add4(bv[4], bv[5]));
            const long long M0 = (long long)tile * C::TILE_OUT;
            const int Gq = 16 * ww + (lane >> 2), kq = (lane & 3) ^ ((Gq >> 1) & 3);
            const long long m = M0 + 8 * Gq + 2 * kq;
            char *dst = out + SB * m;
            if (m + 2 <= a.n_out) store_pair<HALFIN>(dst, r.x, r.y, r.z, r.w);
            else if (m < a.n_out) store_one<HALFIN>(dst, r.x, r.y);
            return;
        }
        __syncthreads();                                    // the exchange area may be overwritten by the next DMA
        int nblk = blk + 1, ntile = tile;
        if (nblk == NB) {
            nblk = 0;
            ++round;
            ntile = round < n_rounds ? tile_of(round) : -1;
        }
        if (ntile >= 0) stage(ntile, nblk);

        // ---- the tree over the blocks (wave-uniform branches)
        if (blk & 1) {
            y = add4(lv0, y);
            if (blk & 2) { y = add4(lv1, y); lv2 = y; }
            else lv1 = y;
        } else lv0 = y;

        if (blk == NB - 1) {
            // 3 = 11b: levels 0 and 1 wait; 6 = 110b: levels 1 and 2
            const f32x4 r = NB == 1 ? lv0 : (NB == 2 ? lv1 : (NB == 3 ? add4(lv1, lv0) : add4(lv2, lv1)));
            const long long M0 = (long long)tile * C::TILE_OUT;
            // the slot this lane read holds chunk kq of group Gq: a permutation inside each 64-byte group, so the wave's
            // store still covers one kilobyte of consecutive bytes
            const int Gq = 16 * ww + (lane >> 2), kq = (lane & 3) ^ ((Gq >> 1) & 3);
            const long long m = M0 + 8 * Gq + 2 * kq;
            char *dst = out + SB * m;
            if (tile < n_full) {
                store_pair<HALFIN>(dst, r.x, r.y, r.z, r.w);
            } else {
                const long long n_out = rare_args()->n_out;     // the call's last tile
                if (m + 2 <= n_out) store_pair<HALFIN>(dst, r.x, r.y, r.z, r.w);
                else if (m < n_out) store_one<HALFIN>(dst, r.x, r.y);
            }
        }
        if (ntile < 0) break;
        tile = ntile;
        blk = nblk;
    }
}

}  // namespace sxfir
