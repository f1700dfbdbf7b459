// Microbenchmark: do plain VALU instructions overlap with v_mfma_f32_16x16x4_f32 on gfx950?
// Every loop iteration issues 8 independent MFMAs (8 x 32 = 256 matrix-core cycles per wave) and K independent v_fma_f32
// (K x 4 VALU cycles).  If the two pipes overlap, the time stays flat until K x 4 x waves approaches the MFMA time; if the
// MFMA holds the VALU port for its whole duration, the time grows by 4 K cycles per iteration from K = 0 on.
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_valu_overlap.hip -o tools/micro/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int K, bool MFMA>
__global__ __launch_bounds__(256) void mix(float *out, int iters, float a, float b) {
    f32x4 acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = {a, b, a, b};
    float v[K > 0 ? K : 1];
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = a + k + threadIdx.x;
    const float x = a + threadIdx.x, y = b - threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        if (MFMA) {
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[c], 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(x), "v"(y));
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
#pragma unroll
    for (int k = 0; k < K; ++k) s += v[k];
    if (s == 12345.678f) out[0] = s;
}

template <typename Kern>
static double run(Kern kern, int blocks, int iters, float *out) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3 / iters * 1e3;                            // ns per loop iteration
}

int main() {
    float *out;
    (void)hipMalloc(&out, 4);
    const int iters = 20000;
    for (int wpsimd : {1, 2, 4}) {
        const int blocks = 256 * wpsimd;
        printf("waves/SIMD %d  ns/iteration:  MFMA only %.1f | +16 VALU %.1f | +32 %.1f | +64 %.1f | +128 %.1f || VALU only 32 %.1f | 64 %.1f | 128 %.1f\n", wpsimd,
               run(mix<0, true>, blocks, iters, out), run(mix<16, true>, blocks, iters, out), run(mix<32, true>, blocks, iters, out),
               run(mix<64, true>, blocks, iters, out), run(mix<128, true>, blocks, iters, out),
               run(mix<32, false>, blocks, iters, out), run(mix<64, false>, blocks, iters, out), run(mix<128, false>, blocks, iters, out));
        fflush(stdout);
    }
    (void)hipFree(out);
    return 0;
}
