// Experimental SpMM variants (development aid; not part of the product library).
// Build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/exp_spmm.hip -o tools/libexp_spmm.so
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, uint32_t bytes) {
    // raw buffer, stride 0, num_records = bytes; third dword 0x00020000 as rocm libraries use for gfx94x/gfx950
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}

// POLICY: 0 plain global load, 1 nt, 2 sc1, 3 sc0|sc1 (buffer loads), 4 = __builtin_nontemporal_load
template <int LPN, int POLICY, int UNROLL, bool NT_STREAM>
__global__ __launch_bounds__(256) void spmm_var(const int *__restrict__ rowptr, const int *__restrict__ colidx,
                                                const float *__restrict__ vals, const float *__restrict__ X,
                                                float *__restrict__ Y, int n_rows, uint32_t x_bytes) {
    constexpr int F = LPN * 4, NS = 64 / LPN;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int q = lane % LPN, slot = lane / LPN;
    const int beg = rowptr[row], end = rowptr[row + 1];
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(X, x_bytes);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int p0 = beg + slot; p0 < end; p0 += UNROLL * NS) {
        int c[UNROLL]; float v[UNROLL]; f32x4 x[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int p = p0 + u * NS;
            const bool ok = p < end;
            if (NT_STREAM) {
                c[u] = ok ? __builtin_nontemporal_load(colidx + p) : 0;
                v[u] = ok ? __builtin_nontemporal_load(vals + p) : 0.f;
            } else {
                c[u] = ok ? colidx[p] : 0;
                v[u] = ok ? vals[p] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const uint32_t off = ((uint32_t)c[u] * F + 4 * q) * 4u;
            if (POLICY == 0) x[u] = *reinterpret_cast<const f32x4 *>(X + (size_t)c[u] * F + 4 * q);
            else if (POLICY == 1) x[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 2));
            else if (POLICY == 2) x[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16));
            else if (POLICY == 3) x[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 17));
            else x[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(X + (size_t)c[u] * F + 4 * q));
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u] * x[u];
    }
#pragma unroll
    for (int off = 32; off >= LPN; off >>= 1) {
        acc[0] += __shfl_xor(acc[0], off, 64); acc[1] += __shfl_xor(acc[1], off, 64);
        acc[2] += __shfl_xor(acc[2], off, 64); acc[3] += __shfl_xor(acc[3], off, 64);
    }
    if (slot == 0) *reinterpret_cast<f32x4 *>(Y + (size_t)row * F + 4 * q) = acc;
}

#define LAUNCH(LPN, POL, UNR, NT) \
    hipLaunchKernelGGL((spmm_var<LPN, POL, UNR, NT>), dim3((n_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, \
                       rowptr, colidx, vals, X, Y, n_rows, x_bytes)

extern "C" int exp_spmm(const int *rowptr, const int *colidx, const float *vals, const float *X, float *Y,
                        int n_rows, int n_cols, int F, int policy, int unroll, int nt_stream, void *stream) {
    const uint32_t x_bytes = (uint32_t)((size_t)n_cols * F * 4);
    if (F == 8) {
        if (nt_stream) {
            if (unroll == 2) { switch (policy) { case 0: LAUNCH(2, 0, 2, true); break; case 1: LAUNCH(2, 1, 2, true); break; case 2: LAUNCH(2, 2, 2, true); break; case 3: LAUNCH(2, 3, 2, true); break; default: LAUNCH(2, 4, 2, true); } }
            else { switch (policy) { case 0: LAUNCH(2, 0, 4, true); break; case 1: LAUNCH(2, 1, 4, true); break; case 2: LAUNCH(2, 2, 4, true); break; case 3: LAUNCH(2, 3, 4, true); break; default: LAUNCH(2, 4, 4, true); } }
        } else {
            if (unroll == 2) { switch (policy) { case 0: LAUNCH(2, 0, 2, false); break; case 1: LAUNCH(2, 1, 2, false); break; case 2: LAUNCH(2, 2, 2, false); break; case 3: LAUNCH(2, 3, 2, false); break; default: LAUNCH(2, 4, 2, false); } }
            else { switch (policy) { case 0: LAUNCH(2, 0, 4, false); break; case 1: LAUNCH(2, 1, 4, false); break; case 2: LAUNCH(2, 2, 4, false); break; case 3: LAUNCH(2, 3, 4, false); break; default: LAUNCH(2, 4, 4, false); } }
        }
    } else if (F == 32) {
        switch (policy) { case 0: LAUNCH(8, 0, 2, false); break; case 1: LAUNCH(8, 1, 2, false); break; case 2: LAUNCH(8, 2, 2, false); break; case 3: LAUNCH(8, 3, 2, false); break; default: LAUNCH(8, 4, 2, false); }
    } else return -2;
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
