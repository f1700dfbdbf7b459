// Layout kernels: column-slice copy (Concatenate as a layout operation) and the per-layer
// sum / mean of ReductionLayer.  Reference semantics: src/layers/reduction.py:15-33,
// src/layers/fusion.py:51-53.  Pure HBM-bound element work: one float per lane, rows contiguous.
#include "amar_common.h"

namespace {

__global__ __launch_bounds__(256) void copy_columns_kernel(const float *__restrict__ src, int64_t lds,
                                                           const int32_t *__restrict__ ids, int base,
                                                           float *__restrict__ dst, int64_t ldd, int64_t n_rows,
                                                           int width) {
    const int64_t total = n_rows * width;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / width;
        const int c = (int)(i - r * width);
        const int64_t sr = ids ? (int64_t)ids[r] - base : r;
        dst[r * ldd + c] = src[sr * lds + c];
    }
}

// the same copy in 16-byte pieces (width and both leading dimensions multiples of 4 floats, 16-byte aligned bases): a quarter of
// the memory instructions — the per-element form moved the 768 k x 8 block of a partitioned step in 18 us
__global__ __launch_bounds__(256) void copy_columns_vec_kernel(const float *__restrict__ src, int64_t lds,
                                                               const int32_t *__restrict__ ids, int base,
                                                               float *__restrict__ dst, int64_t ldd, int64_t n_rows,
                                                               int quads) {
    const int64_t total = n_rows * quads;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / quads;
        const int q = (int)(i - r * quads);
        const int64_t sr = ids ? (int64_t)ids[r] - base : r;
        *reinterpret_cast<float4 *>(dst + r * ldd + 4 * q) = *reinterpret_cast<const float4 *>(src + sr * lds + 4 * q);
    }
}

// out[r, c] = (((X_0 + X_1) + X_2) + ...)[r, c]  (/ n_layers for 'mean'): tf.add_n order, then tf.divide
__global__ __launch_bounds__(256) void reduce_layers_kernel(const float *__restrict__ cat, int64_t ld, int n_layers,
                                                            int width, float *__restrict__ out, int64_t ldo,
                                                            int64_t n_rows, float div) {
    const int64_t total = n_rows * width;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / width;
        const int c = (int)(i - r * width);
        float s = cat[r * ld + c];
        for (int l = 1; l < n_layers; ++l) s += cat[r * ld + (int64_t)l * width + c];
        out[r * ldo + c] = div != 0.f ? s / div : s;
    }
}

unsigned grid_for(int64_t total) {
    int64_t b = (total + 255) / 256;
    return (unsigned)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" {

int amar_copy_columns_f32(const float *src, int64_t lds, const int32_t *ids, int32_t base,
                          float *dst, int64_t ldd, int64_t n_rows, int32_t width, amar_stream_t stream) {
    if (n_rows < 0 || width < 1 || !src || !dst || lds < width || ldd < width) return AMAR_EINVAL;
    if (n_rows == 0) return AMAR_OK;
    if ((width & 3) == 0 && (lds & 3) == 0 && (ldd & 3) == 0 && amar_aligned16(src) && amar_aligned16(dst))
        hipLaunchKernelGGL(copy_columns_vec_kernel, dim3(grid_for(n_rows * (width / 4))), dim3(256), 0,
                           static_cast<hipStream_t>(stream), src, lds, ids, base, dst, ldd, n_rows, width / 4);
    else
        hipLaunchKernelGGL(copy_columns_kernel, dim3(grid_for(n_rows * width)), dim3(256), 0,
                           static_cast<hipStream_t>(stream), src, lds, ids, base, dst, ldd, n_rows, width);
    return amar_check_launch();
}

int amar_reduce_layers_f32(const float *cat, int64_t ld, int32_t n_layers, int32_t width, float *out, int64_t ldo,
                           int64_t n_rows, int32_t mean, amar_stream_t stream) {
    if (n_rows < 0 || n_layers < 1 || width < 1 || !cat || !out || ld < (int64_t)n_layers * width || ldo < width)
        return AMAR_EINVAL;
    if (n_rows == 0) return AMAR_OK;
    hipLaunchKernelGGL(reduce_layers_kernel, dim3(grid_for(n_rows * width)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), cat, ld, n_layers, width, out, ldo, n_rows,
                       mean ? (float)n_layers : 0.f);
    return amar_check_launch();
}

}  // extern "C"
