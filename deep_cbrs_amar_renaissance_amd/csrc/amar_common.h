// Shared device/host helpers for the gfx950 kernels behind include/amar_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/amar_hip.h"

#define AMAR_WAVE 64

extern thread_local int amar_tls_hip_error;

static inline int amar_check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { amar_tls_hip_error = (int)e; return AMAR_ELAUNCH; }
    return AMAR_OK;
}

static inline bool amar_aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// hipFuncAttributeMaxDynamicSharedMemorySize is kept PER DEVICE by the runtime: a kernel that needs more than 64 KB of dynamic LDS
// gets the attribute once on every device it is launched on (`done` = the call site's own per-device flags); a refused call is
// reported instead of surfacing later as an opaque launch failure.
#define AMAR_MAX_DEVICES 64
static inline int amar_allow_lds(const void *kern, size_t bytes, bool (&done)[AMAR_MAX_DEVICES]) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= AMAR_MAX_DEVICES) { amar_tls_hip_error = (int)e; return AMAR_ELAUNCH; }
    if (!done[dev]) {
        e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) { amar_tls_hip_error = (int)e; return AMAR_ELAUNCH; }
        done[dev] = true;
    }
    return AMAR_OK;
}

__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4_fma(float a, float4 x, float4 acc) {
    acc.x = fmaf(a, x.x, acc.x); acc.y = fmaf(a, x.y, acc.y);
    acc.z = fmaf(a, x.z, acc.z); acc.w = fmaf(a, x.w, acc.w);
    return acc;
}
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) {
    return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
__device__ __forceinline__ float4 f4_shfl_xor(float4 v, int mask) {
    return make_float4(__shfl_xor(v.x, mask, 64), __shfl_xor(v.y, mask, 64),
                       __shfl_xor(v.z, mask, 64), __shfl_xor(v.w, mask, 64));
}
__device__ __forceinline__ float4 f4_shfl(float4 v, int src) {
    return make_float4(__shfl(v.x, src, 64), __shfl(v.y, src, 64), __shfl(v.z, src, 64), __shfl(v.w, src, 64));
}
__device__ __forceinline__ float f4_get(const float4 &v, int i) {
    return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w));
}

// ---- cross-lane reductions on the VALU (no LDS-pipe traffic) -----------------------------------------
// DPP row rotations cover lane offsets 1..8 inside a 16-lane row; gfx950's v_permlane16_swap /
// v_permlane32_swap cover offsets 16 and 32.  hipcc fuses `v + dpp(v)` into one v_add_f32_dpp.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// v_permlane16_swap / v_permlane32_swap exchange rows between TWO registers: with both holding v, one ends up with
// the even rows (or low half) everywhere and the other with the odd rows (high half), i.e. `own` and the xor-16
// (xor-32) partner's value.  ROCm 7.2's builtin loses the second result, so this is inline asm; hipcc adds no
// wait states inside asm (cdna_hip_programming.md 5.7), hence the s_nop on both sides.
__device__ __forceinline__ float swap16_other(float v, float &own) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    own = a;
    return b;
}
__device__ __forceinline__ float swap32_other(float v, float &own) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    own = a;
    return b;
}
// Sum over all lanes l' with l' % STRIDE == l % STRIDE; every lane receives the total. STRIDE in {1,2,4,8,16}.
template <int STRIDE>
__device__ __forceinline__ float wave_sum_stride(float v) {
    if (STRIDE <= 8) v += dpp_mov<0x128>(v);          // row_ror:8
    if (STRIDE <= 4) v += dpp_mov<0x124>(v);          // row_ror:4
    if (STRIDE <= 2) v += dpp_mov<0x122>(v);          // row_ror:2
    if (STRIDE <= 1) v += dpp_mov<0x121>(v);          // row_ror:1
    float a, b;
    b = swap16_other(v, a); v = a + b;                 // lane offset 16
    b = swap32_other(v, a); v = a + b;                 // lane offset 32
    return v;
}
__device__ __forceinline__ float wave_max_all(float v) {
    v = fmaxf(v, dpp_mov<0x128>(v)); v = fmaxf(v, dpp_mov<0x124>(v));
    v = fmaxf(v, dpp_mov<0x122>(v)); v = fmaxf(v, dpp_mov<0x121>(v));
    float a, b;
    b = swap16_other(v, a); v = fmaxf(a, b);
    b = swap32_other(v, a); v = fmaxf(a, b);
    return v;
}
template <int STRIDE>
__device__ __forceinline__ float4 f4_wave_sum_stride(float4 v) {
    return make_float4(wave_sum_stride<STRIDE>(v.x), wave_sum_stride<STRIDE>(v.y),
                       wave_sum_stride<STRIDE>(v.z), wave_sum_stride<STRIDE>(v.w));
}
