// Helpers shared by the shipped kernels (decim4_wide_kernel, decim_dense_kernel, decim_blocks_kernel, interp8_pass_kernel,
// interp_tile_kernel) and by the profiling build's A/B partners (decim4_tile2_kernel, decim_multi_kernel, the experiments):
// packed FMAs with a scalar tap operand, the conflict-free lane maps' tables, the argument block of the multi-row decimators,
// half <-> float.  Round 6: these lived in sxfir_decim_tile2.hip.h and sxfir_decim_multi.hip.h, whose kernels have had no instance
// in the production library since rounds 4 / 5; the product's translation unit now includes those two headers under
// SXFIR_PROFILING only.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "sxfir_decim_tile.hip.h"       // f32x2 / f32x4, glds16, permlane32_swap

namespace sxfir {

// Argument block of the multi-row decimators (decim_dense_kernel, decim_blocks_kernel; profiling: decim_multi_kernel)
struct DecimMultiArgs {
    const void *in;         // channel 0, sample 0 of this call (aligned to one complex sample)
    const void *hist;       // NT samples preceding `in`
    void *hist_out;
    void *out;              // 16-byte aligned
    const float *taps;
    long long n_in, n_out;
    long long in_stride, out_stride, hist_stride;
    int n_tiles;            // workgroup tiles per channel
    int n_groups;           // workgroups per channel (strided passes over the tiles)
    unsigned long long *stamps;   // diagnostic builds only (ABL 3): 5 counters per wave
};

// ---- packed FMAs: acc += tap * x for two floats (I, Q) at once, the tap one half of an SGPR pair
// the same packed FMAs with the tap pair in SGPRs
__device__ __forceinline__ void pk_fma_s_lo(f32x2 &acc, const f32x2 &hpair, const f32x2 &x)
{
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "s"(hpair), "v"(x));
}
__device__ __forceinline__ void pk_fma_s_hi(f32x2 &acc, const f32x2 &hpair, const f32x2 &x)
{
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(hpair), "v"(x));
}
// the first FMA of a chain: acc = fmaf(tap, x, +0.0f) with the zero as an inline constant, so no register is cleared first
// (a cleared register costs a v_mov_b64 per chain and tile: 0.6 of a packed FMA's energy each, profiles/round4z9_price_list.txt)
__device__ __forceinline__ void pk_fma_s_lo_first(f32x2 &acc, const f32x2 &hpair, const f32x2 &x)
{
    asm("v_pk_fma_f32 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "=v"(acc) : "s"(hpair), "v"(x));
}
__device__ __forceinline__ void pk_fma_s_hi_first(f32x2 &acc, const f32x2 &hpair, const f32x2 &x)
{
    asm("v_pk_fma_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,1,0]" : "=v"(acc) : "s"(hpair), "v"(x));
}

// ... as volatile asm (issue order = source order)
__device__ __forceinline__ void pk_fma_sv_lo(f32x2 &acc, const f32x2 &hpair, const f32x2 &x)
{
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "s"(hpair), "v"(x));
}
__device__ __forceinline__ void pk_fma_sv_hi(f32x2 &acc, const f32x2 &hpair, const f32x2 &x)
{
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(hpair), "v"(x));
}


// byte offset (from the tile's first staged chunk) of the chunk that lands in slot q of the image
__device__ __forceinline__ unsigned slot_source_offset(unsigned q, unsigned chunks)
{
    unsigned off = q - (((q + 1u) * 3856u) >> 16);                        // (q+1)/17, exact for q < 4096
    off = off < chunks ? off : chunks - 1u;
    return 16u * off;
}

// output group r of lanes 8k..8k+7 of a half-wave, 5 bits each (see the kernel: conflict-free ds_read_b128 groups)
constexpr unsigned long long rgrp_word(int k)
{
    unsigned long long w = 0;
    for (int i = 0; i < 8; ++i) {
        const int l5 = 8 * k + i;
        const bool first = l5 < 4 || (l5 >= 12 && l5 < 16) || (l5 >= 20 && l5 < 28);
        const int idx = first ? (l5 < 4 ? l5 : (l5 < 16 ? l5 - 8 : l5 - 12)) : (l5 < 12 ? l5 - 4 : (l5 < 20 ? l5 - 8 : l5 - 16));
        w |= (unsigned long long)(2 * idx + (first ? 0 : 1)) << (5 * i);
    }
    return w;
}
__device__ __forceinline__ unsigned long long rgrp_table(int k)
{
    constexpr unsigned long long W0 = rgrp_word(0), W1 = rgrp_word(1), W2 = rgrp_word(2), W3 = rgrp_word(3);
    return k == 0 ? W0 : (k == 1 ? W1 : (k == 2 ? W2 : W3));
}

__device__ __forceinline__ void permlane16_swap(float &vdst, float &src)
{
    // odd 16-lane rows of vdst <-> even rows of src (inline asm for the same reason as permlane32_swap)
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(vdst), "+v"(src));
}

__device__ __forceinline__ float half_bits_to_float(unsigned bits16)
{
    return __half2float(__ushort_as_half((unsigned short)bits16));
}

__device__ __forceinline__ unsigned pack_half2(float i, float q)
{
    const __half2 h = __floats2half2_rn(i, q);
    return *reinterpret_cast<const unsigned *>(&h);
}

}  // namespace sxfir
