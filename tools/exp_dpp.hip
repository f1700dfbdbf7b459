// Checks wave_sum_stride / wave_max_all (amar_common.h) against a scalar sum. Build: hipcc --offload-arch=gfx950 -shared -fPIC
#include "../deep_cbrs_amar_renaissance_amd/csrc/amar_common.h"
thread_local int amar_tls_hip_error = 0;
template <int S> __global__ void k(const float *in, float *out) { out[threadIdx.x] = wave_sum_stride<S>(in[threadIdx.x]); }
__global__ void kmax(const float *in, float *out) { out[threadIdx.x] = wave_max_all(in[threadIdx.x]); }
__global__ void kswap(const float *in, float *o16a, float *o16b, float *o32a, float *o32b) {
    float a, b;
    b = swap16_other(in[threadIdx.x], a); o16a[threadIdx.x] = a; o16b[threadIdx.x] = b;
    b = swap32_other(in[threadIdx.x], a); o32a[threadIdx.x] = a; o32b[threadIdx.x] = b;
}
extern "C" void run(int s, const float *in, float *out) {
    switch (s) {
    case 1: hipLaunchKernelGGL(k<1>, 1, 64, 0, 0, in, out); break;
    case 2: hipLaunchKernelGGL(k<2>, 1, 64, 0, 0, in, out); break;
    case 4: hipLaunchKernelGGL(k<4>, 1, 64, 0, 0, in, out); break;
    case 8: hipLaunchKernelGGL(k<8>, 1, 64, 0, 0, in, out); break;
    case 16: hipLaunchKernelGGL(k<16>, 1, 64, 0, 0, in, out); break;
    case 0: hipLaunchKernelGGL(kmax, 1, 64, 0, 0, in, out); break;
    }
}
extern "C" void run_swap(const float *in, float *a, float *b, float *c, float *d) { hipLaunchKernelGGL(kswap, 1, 64, 0, 0, in, a, b, c, d); }
