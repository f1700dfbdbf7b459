// Microbenchmark: sustained fp32 MFMA rate of v_mfma_f32_16x16x4_f32 (what csrc/amar_chain.hip runs on) and of
// v_mfma_f32_32x32x2_f32, with 1, 2, 4 or 8 independent accumulator chains per wave and 1..8 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_peak.hip -o gpurun_out/mfma_peak ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ __launch_bounds__(256) void mfma16(float *out, int iters, float a, float b) {
    f32x4 acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = {a, b, a, b};
    const float x = a + threadIdx.x, y = b - threadIdx.x;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[c], 0, 0, 0);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    if (s == 12345.678f) out[0] = s;
}

template <int CHAINS>
__global__ __launch_bounds__(256) void mfma32(float *out, int iters, float a, float b) {
    f32x16 acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = a + r;
    const float x = a + threadIdx.x, y = b - threadIdx.x;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[c], 0, 0, 0);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    if (s == 12345.678f) out[0] = s;
}

template <typename K>
static double run(K kern, int blocks, int iters, int chains, double flop_per_mfma, float *out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * chains * flop_per_mfma;
    return flops / (ms * 1e-3) / 1e12;
}

int main() {
    float *out;
    hipMalloc(&out, 4);
    const int iters = 20000;
    for (int wpsimd : {1, 2, 4, 8}) {
        const int blocks = 256 * wpsimd;                       // one 256-thread block = one wave per SIMD of a CU
        printf("waves/SIMD %d : 16x16x4 chains 1/2/4/8 = %.1f %.1f %.1f %.1f TF | 32x32x2 chains 1/2/4 = %.1f %.1f %.1f TF\n", wpsimd,
               run(mfma16<1>, blocks, iters, 1, 2048., out), run(mfma16<2>, blocks, iters, 2, 2048., out),
               run(mfma16<4>, blocks, iters, 4, 2048., out), run(mfma16<8>, blocks, iters, 8, 2048., out),
               run(mfma32<1>, blocks, iters, 1, 4096., out), run(mfma32<2>, blocks, iters, 2, 4096., out),
               run(mfma32<4>, blocks, iters, 4, 4096., out));
        fflush(stdout);
    }
    hipFree(out);
    return 0;
}
