// xv_split8.h -- the "split8" activation / weight representation of the f16bf8 arithmetic (device code, gfx950).
//
// A value v (fp32) is carried as   hi = fp16(v)            2 bytes   (the MAIN operand of a v_mfma_f32_32x32x16_f16)
//                                  l8 = bf8(2^11 (v - hi))  1 byte    (e5m2: fp16's exponent range, 2 mantissa bits)
//                                  h8 = bf8(v)              1 byte
// and a product x*w is formed as   xh*wh  +  2^-11 (xl8*wh8 + xh8*wl8)      (fp32 accumulate)
// i.e. ONE fp16 MFMA and ONE block-scaled 8-bit MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, twice the fp16 rate; its K = 64 holds
// the two cross terms of 32 channels side by side, the 2^-11 is the instruction's E8M0 scale operand) where the bf16x3
// arithmetic of xv_kernels.hip spends three bf16 MFMAs.  hi carries 11 significant bits, hi + l8/2^11 about 14; the cross
// terms are 2^-12 of the product and are themselves good to 2-3 bits, so a product is good to ~2^-15 ... 2^-16 -- measured
// on the full network: 1.1e-5 relative L2 against the fp64 oracle (bf16x3: 5e-6; the bar is 1e-4).
//
// Range: fp16 / bf8 saturate at 57344 (largest finite e5m2, below fp16's 65504).  Encoders clamp to +-57344 and report the
// largest magnitude they saw, so that callers can flag the (absurd for a BN-normalised network) overflow and fall back.
//
// Row-slab layout (128 bytes per row and 32-channel slab -- the same geometry as the bf16 split format, so buffers, DMA
// pieces and swizzles are shared): 16-byte slot g (g = 0..3) = fp16 hi of channels 8g..8g+7; slot 4+g = [8 x l8 | 8 x h8] of
// the same channels.  Slots are XOR-swizzled with (row >> 1) & 7 (activations) / (col >> 2) & 3 (weight tiles).
// Weight tiles store [8 x h8 | 8 x l8] in slot 4+g: byte p of an activation slot always meets byte p of a weight slot, so
// l8 pairs with h8 and vice versa.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 xv_f16x8 __attribute__((ext_vector_type(8)));
typedef int xv_i32x4 __attribute__((ext_vector_type(4)));
typedef int xv_i32x8 __attribute__((ext_vector_type(8)));

constexpr float XV_SPLIT8_MAX = 57344.f;
constexpr float XV_SPLIT8_LO_SCALE = 2048.f;          // 2^11
constexpr int XV_SPLIT8_E8M0 = 127 - 11;              // E8M0 scale of the cross-term MFMA: 2^-11

// 8 channels -> hi slot + cross slot.  LO_FIRST: activations ([l8 | h8]); weights use the opposite order.
template <bool LO_FIRST>
__device__ __forceinline__ void xv_split8_encode8(const float (&v)[8], xv_f16x8 &hi, xv_i32x4 &x, float &amax)
{
    float c[8], lo[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        amax = fmaxf(amax, fabsf(v[i]));
        c[i] = __builtin_amdgcn_fmed3f(v[i], -XV_SPLIT8_MAX, XV_SPLIT8_MAX);
        hi[i] = (_Float16)c[i];
        lo[i] = c[i] - (float)hi[i];
    }
    // the "* 2^11" rides on the conversion: v_cvt_scalef32_pk_bf8_f32 converts x / scale, bit for bit what v_cvt_pk_bf8_f32 gives
    // for x * 2^11 when scale = 2^-11 (tools/experiments/cvt_scale_probe.hip) -- one VALU instruction per value less
    typedef short xv_s16x2 __attribute__((ext_vector_type(2)));
    constexpr float inv = 1.f / XV_SPLIT8_LO_SCALE;
    xv_s16x2 t0 = {0, 0}, t1 = {0, 0};
    t0 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(t0, lo[0], lo[1], inv, false);
    t0 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(t0, lo[2], lo[3], inv, true);
    t1 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(t1, lo[4], lo[5], inv, false);
    t1 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(t1, lo[6], lo[7], inv, true);
    const int l0 = __builtin_bit_cast(int, t0), l1 = __builtin_bit_cast(int, t1);
    int h0 = __builtin_amdgcn_cvt_pk_bf8_f32(c[0], c[1], 0, false);
    h0 = __builtin_amdgcn_cvt_pk_bf8_f32(c[2], c[3], h0, true);
    int h1 = __builtin_amdgcn_cvt_pk_bf8_f32(c[4], c[5], 0, false);
    h1 = __builtin_amdgcn_cvt_pk_bf8_f32(c[6], c[7], h1, true);
    if constexpr (LO_FIRST) x = (xv_i32x4){l0, l1, h0, h1};
    else x = (xv_i32x4){h0, h1, l0, l1};
}

__device__ __forceinline__ float xv_bf8_to_float(uint32_t b)        // e5m2 = the upper byte of an fp16
{
    return (float)__builtin_bit_cast(_Float16, (uint16_t)((b & 0xffu) << 8));
}
