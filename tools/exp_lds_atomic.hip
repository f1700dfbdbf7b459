// Micro-benchmark (development aid): rate of full-wave ds_add_f32 into random rows of an LDS-resident Y tile
// (the open question of DESIGN.md section 10 item 1).  1024-thread workgroups, one per CU, Y tile [F=8][ROWS].
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int ROWS, int MODE>
__global__ __launch_bounds__(1024) void lds_atomic_kernel(int iters, float *out) {
    extern __shared__ float tile[];                       // [8][ROWS]
    for (int i = threadIdx.x; i < 8 * ROWS; i += 1024) tile[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int q = lane >> 5;                              // feature quad (0/1), 32 entries per wave-instruction
    unsigned h = (blockIdx.x * 1024 + threadIdx.x) * 2654435761u + 7u;
    for (int it = 0; it < iters; ++it) {
        h = h * 1664525u + 1013904223u;
        unsigned row;
        if (MODE == 0) row = (h >> 8) % ROWS;                                  // random rows
        else if (MODE == 1) row = ((threadIdx.x >> 6) * 32 + (lane & 31) + it * 64) % ROWS;   // conflict-free consecutive rows
        else row = ((h >> 8) % (ROWS / 32)) * 32 + (lane & 31);               // random 32-row groups, distinct banks inside a group
        float *dst = tile + row;
        const float v = (float)(h & 1023) * 1e-3f;
        atomicAdd(dst + (4 * q + 0) * ROWS, v); atomicAdd(dst + (4 * q + 1) * ROWS, v);
        atomicAdd(dst + (4 * q + 2) * ROWS, v); atomicAdd(dst + (4 * q + 3) * ROWS, v);
    }
    __syncthreads();
    float s = 0.f;
    for (int i = threadIdx.x; i < 8 * ROWS; i += 1024) s += tile[i];
    if (s == 123.456f) out[0] = s;
}

extern "C" float run_lds_atomic(int mode, int rows, int iters, int blocks, float *out, int reps) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto launch = [&]() {
        const size_t lds = (size_t)8 * rows * 4;
        if (rows == 2048) {
            if (mode == 0) hipLaunchKernelGGL((lds_atomic_kernel<2048, 0>), dim3(blocks), dim3(1024), lds, 0, iters, out);
            else if (mode == 1) hipLaunchKernelGGL((lds_atomic_kernel<2048, 1>), dim3(blocks), dim3(1024), lds, 0, iters, out);
            else hipLaunchKernelGGL((lds_atomic_kernel<2048, 2>), dim3(blocks), dim3(1024), lds, 0, iters, out);
        } else {
            if (mode == 0) hipLaunchKernelGGL((lds_atomic_kernel<4096, 0>), dim3(blocks), dim3(1024), lds, 0, iters, out);
            else if (mode == 1) hipLaunchKernelGGL((lds_atomic_kernel<4096, 1>), dim3(blocks), dim3(1024), lds, 0, iters, out);
            else hipLaunchKernelGGL((lds_atomic_kernel<4096, 2>), dim3(blocks), dim3(1024), lds, 0, iters, out);
        }
    };
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(lds_atomic_kernel<4096, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 4096 * 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(lds_atomic_kernel<4096, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 4096 * 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(lds_atomic_kernel<4096, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 4096 * 4);
    launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) launch();
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}
