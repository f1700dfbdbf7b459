// Micro-benchmark (development aid): what does a CU's vector-memory front end sustain for small-row gathers
// that hit in L1 / L2?  Each wave issues ITER gathers of 32-byte rows with register-generated pseudo-random rows.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

// MODE 0: lane = q*32+s, dwordx4 (current kernel)      1: lane = 2*s+q, dwordx4 (row halves in adjacent lanes)
// MODE 2: lane = 4*s+q, dwordx2 (4 lanes per row)      3: lane = 8*s+q, dword (8 lanes per row)
// MODE 4: dwordx4, every lane its own row (first 16 B only): 64 rows per instruction
template <int MODE>
__global__ __launch_bounds__(256) void gather_kernel(const float *X, unsigned row_mask, int iters, float *out) {
    const int lane = threadIdx.x & 63;
    unsigned state = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    int grp, sub;
    if (MODE == 0 || MODE >= 5) { grp = lane & 31; sub = lane >> 5; }
    else if (MODE == 1) { grp = lane >> 1; sub = lane & 1; }
    else if (MODE == 2) { grp = lane >> 2; sub = lane & 3; }
    else if (MODE == 3) { grp = lane >> 3; sub = lane & 7; }
    else { grp = lane; sub = 0; }
    const unsigned wave_seed = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 97u;
    float acc = 0.f;
    for (int it = 0; it < iters; it += 4) {
        float4 r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            // a row per group (same for all lanes of the group), different per iteration / wave
            unsigned h = (wave_seed + (unsigned)(it + u) * 64u + (unsigned)grp) * 2654435761u;
            h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
            const unsigned row = h & row_mask;
            const char *p = reinterpret_cast<const char *>(X) + row * 32u;
            if (MODE == 0 || MODE == 1) r[u] = *reinterpret_cast<const float4 *>(p + sub * 16);
            else if (MODE == 2) { const float2 t = *reinterpret_cast<const float2 *>(p + sub * 8); r[u] = make_float4(t.x, t.y, 0.f, 0.f); }
            else if (MODE == 3) { r[u] = make_float4(*reinterpret_cast<const float *>(p + sub * 4), 0.f, 0.f, 0.f); }
            else if (MODE == 4) r[u] = *reinterpret_cast<const float4 *>(p);
            else {
                const unsigned off = row * 32u + (lane >> 5) * 16u;
                if (MODE == 5) asm volatile("global_load_dwordx4 %0, %1, %2 sc0" : "=v"(r[u]) : "v"(off), "s"(X) : "memory");
                if (MODE == 6) asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=v"(r[u]) : "v"(off), "s"(X) : "memory");
                if (MODE == 7) asm volatile("global_load_dwordx4 %0, %1, %2 sc0 sc1" : "=v"(r[u]) : "v"(off), "s"(X) : "memory");
                if (MODE == 8) asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(r[u]) : "v"(off), "s"(X) : "memory");
                if (MODE == 9) asm volatile("global_load_dwordx4 %0, %1, %2 sc0 sc1 nt" : "=v"(r[u]) : "v"(off), "s"(X) : "memory");
            }
        }
        if (MODE >= 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += r[u].x + r[u].y + r[u].z + r[u].w;
    }
    if (acc == 123.456f) out[0] = acc + state;
}

extern "C" float run_gather(int mode, const float *X, unsigned row_mask, int iters, int blocks, float *out, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&]() {
        switch (mode) {
        case 0: hipLaunchKernelGGL(gather_kernel<0>, dim3(blocks), dim3(256), 0, 0, X, row_mask, iters, out); break;
        case 1: hipLaunchKernelGGL(gather_kernel<1>, dim3(blocks), dim3(256), 0, 0, X, row_mask, iters, out); break;
        case 2: hipLaunchKernelGGL(gather_kernel<2>, dim3(blocks), dim3(256), 0, 0, X, row_mask, iters, out); break;
        case 3: hipLaunchKernelGGL(gather_kernel<3>, dim3(blocks), dim3(256), 0, 0, X, row_mask, iters, out); break;
        case 4: hipLaunchKernelGGL(gather_kernel<4>, dim3(blocks), dim3(256), 0, 0, X, row_mask, iters, out); break;
        case 5: hipLaunchKernelGGL(gather_kernel<5>, dim3(blocks), dim3(256), 0, 0, X, row_mask, iters, out); break;
        case 6: hipLaunchKernelGGL(gather_kernel<6>, dim3(blocks), dim3(256), 0, 0, X, row_mask, iters, out); break;
        case 7: hipLaunchKernelGGL(gather_kernel<7>, dim3(blocks), dim3(256), 0, 0, X, row_mask, iters, out); break;
        case 8: hipLaunchKernelGGL(gather_kernel<8>, dim3(blocks), dim3(256), 0, 0, X, row_mask, iters, out); break;
        default: hipLaunchKernelGGL(gather_kernel<9>, dim3(blocks), dim3(256), 0, 0, X, row_mask, iters, out); break;
        }
    };
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}
