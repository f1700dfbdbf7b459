// Microbenchmark: how do plain VALU instructions share a SIMD with v_mfma_f32_16x16x32_bf16 on gfx950?
// One loop iteration = NM MFMAs (16 matrix-pipe cycles each) and K v_fma_f32 (4 cycles each), in one of four arrangements:
//   MODE 0  clustered: the MFMAs (independent accumulators), then the VALU block         (what a compiler emits for "layer, then epilogue")
//   MODE 1  interleaved in program order: after every MFMA K / NM VALU instructions
//   MODE 2  clustered, ONE accumulator: every MFMA depends on the one before             (the split products' six-deep chains)
//   MODE 3  roles: waves 0-3 of the 512-thread workgroup issue only the MFMAs, waves 4-7 (their SIMD partners) only the VALU block
//   MODE 4  clustered, two accumulators used alternately
// If VALU work hides beside the matrix pipe, the time stays at max(16 NM, 8 NM + 4 K) per wave; if they take turns, 16 NM + 4 K.
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_bf16_overlap.hip -o tools/micro/mfma_bf16_overlap
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int K, int MODE>
__global__ __launch_bounds__(512) void mix(float *out, int iters, float a, float b) {
    constexpr int NM = 8;
    f32x4 acc[NM];
#pragma unroll
    for (int c = 0; c < NM; ++c) acc[c] = {a, b, a, b};
    float v[K > 0 ? K : 1];
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = a + k + threadIdx.x;
    const float x = a + threadIdx.x, y = b - threadIdx.x;
    const u32x4 wa = {0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f003f00u, 0x3f803f80u}, wb = {0x3f803f80u, 0x3f003f00u, 0x3f803f80u, 0x3f803f80u};
    const bf16x8 fa = __builtin_bit_cast(bf16x8, wa), fb = __builtin_bit_cast(bf16x8, wb);
    const bool mfma_role = MODE != 3 || (threadIdx.x >> 8) == 0, valu_role = MODE != 3 || (threadIdx.x >> 8) == 1;   // waves 0-3 and 4-7 share the SIMDs pairwise
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) {
#pragma unroll
            for (int c = 0; c < NM; ++c) {
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[c], 0, 0, 0);
#pragma unroll
                for (int k = c * (K / NM); k < (c + 1) * (K / NM); ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(x), "v"(y));
            }
        } else {
            if (mfma_role) {
#pragma unroll
                for (int c = 0; c < NM; ++c) {
                    if (MODE == 2) acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[0], 0, 0, 0);
                    else if (MODE == 4) acc[c & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[c & 1], 0, 0, 0);
                    else acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[c], 0, 0, 0);
                }
            }
            if (valu_role) {
#pragma unroll
                for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(x), "v"(y));
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NM; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
#pragma unroll
    for (int k = 0; k < K; ++k) s += v[k];
    if (s == 12345.678f) out[0] = s;
}

template <typename Kern>
static double run(Kern kern, int blocks, int iters, float *out) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, out, iters, 1.0f, 2.0f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, out, iters, 1.0f, 2.0f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3 / iters * 1e3;                            // ns per loop iteration
}

template <int MODE>
static void sweep(const char *name, int wpsimd, float *out) {
    const int iters = 20000, blocks = 256 * wpsimd / 2;                     // 512-thread workgroups: two waves per SIMD each
    printf("waves/SIMD %d  %-34s ns/iteration: K=0 %.1f | 16 %.1f | 32 %.1f | 64 %.1f | 128 %.1f\n", wpsimd, name,
           run(mix<0, MODE>, blocks, iters, out), run(mix<16, MODE>, blocks, iters, out), run(mix<32, MODE>, blocks, iters, out),
           run(mix<64, MODE>, blocks, iters, out), run(mix<128, MODE>, blocks, iters, out));
    fflush(stdout);
}

int main() {
    float *out;
    (void)hipMalloc(&out, 4);
    printf("8 x v_mfma_f32_16x16x32_bf16 per iteration (128 pipe cycles = 53 ns at 2.4 GHz) + K x v_fma_f32 (4 K cycles = 1.67 K ns)\n");
    for (int wpsimd : {2, 4}) {
        sweep<0>("clustered", wpsimd, out);
        sweep<1>("interleaved", wpsimd, out);
        sweep<2>("clustered, one accumulator", wpsimd, out);
        sweep<4>("clustered, two accumulators", wpsimd, out);
        sweep<3>("roles (waves 0-3 MFMA / 4-7 VALU)", wpsimd, out);
    }
    (void)hipFree(out);
    return 0;
}
