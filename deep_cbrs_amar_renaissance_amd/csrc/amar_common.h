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
