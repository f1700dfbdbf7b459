// Development aid (tools/exp_rank_of_n.py): a device-side delay of a given number of microseconds on a stream — the emulated wire
// of a collective.  s_memrealtime ticks at a constant 100 MHz whatever the shader clock.
//   hipcc -O2 --offload-arch=gfx950 -shared -fPIC tools/micro/spin.hip -o tools/libexp_spin.so
#include <hip/hip_runtime.h>
#include <cstdint>

__global__ void spin_kernel(long long ticks) {
    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
    while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}

extern "C" int exp_spin_us(double us, void *stream) {
    if (us < 0 || us > 1e5) return -1;                          // bounded: at most 0.1 s
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), (long long)(us * 100.0));
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
