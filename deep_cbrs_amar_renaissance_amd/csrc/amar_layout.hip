// Layout kernels: column-slice copy (Concatenate as a layout operation) and the per-layer
// sum / mean of ReductionLayer.  Reference semantics: src/layers/reduction.py:15-33,
// src/layers/fusion.py:51-53.  Pure HBM-bound element work: one float per lane, rows contiguous.
#include "amar_common.h"
#include <stdlib.h>

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

// 'w-sum' (WeightedSum, reduction.py:36-55): out = reduce_sum(multiply(w * w, [X_0 .. X_L]), axis 0) with a learnable w [L+1] on the
// device.  Products are rounded before they are added, like the two TF ops (no fused multiply-add), and added in layer order.
constexpr int WSUM_MAX_LAYERS = 8, WSUM_BWD_BLOCKS = 512;
__global__ __launch_bounds__(256) void reduce_layers_wsum_kernel(const float *__restrict__ cat, int64_t ld, int n_layers, int width,
                                                                 const float *__restrict__ w, float *__restrict__ out, int64_t ldo,
                                                                 int64_t n_rows) {
    float ww[WSUM_MAX_LAYERS];
    for (int l = 0; l < n_layers; ++l) ww[l] = __fmul_rn(w[l], w[l]);
    const int64_t total = n_rows * width;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / width;
        const int c = (int)(i - r * width);
        float s = __fmul_rn(ww[0], cat[r * ld + c]);
        for (int l = 1; l < n_layers; ++l) s = __fadd_rn(s, __fmul_rn(ww[l], cat[r * ld + (int64_t)l * width + c]));
        out[r * ldo + c] = s;
    }
}

// Reverse of it: d_cat[:, slice l] = w_l^2 d_out and, per workgroup, the partial sums of d_out . X_l (fixed grid, fixed order: the
// result does not depend on scheduling); wsum_dw_kernel adds the partials in workgroup order: dw_l = 2 w_l sum(d_out . X_l).
__global__ __launch_bounds__(256) void reduce_layers_wsum_bwd_kernel(const float *__restrict__ cat, int64_t ld, int n_layers, int width,
                                                                     const float *__restrict__ w, const float *__restrict__ d_out, int64_t ldd,
                                                                     float *__restrict__ d_cat, int64_t ldc, int64_t n_rows,
                                                                     float *__restrict__ partials) {
    __shared__ float red[4][WSUM_MAX_LAYERS];
    float ww[WSUM_MAX_LAYERS], acc[WSUM_MAX_LAYERS];
    for (int l = 0; l < WSUM_MAX_LAYERS; ++l) { ww[l] = l < n_layers ? w[l] * w[l] : 0.f; acc[l] = 0.f; }
    const int64_t total = n_rows * width;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / width;
        const int c = (int)(i - r * width);
        const float g = d_out[r * ldd + c];
#pragma unroll
        for (int l = 0; l < WSUM_MAX_LAYERS; ++l)
            if (l < n_layers) {
                acc[l] = fmaf(g, cat[r * ld + (int64_t)l * width + c], acc[l]);
                d_cat[r * ldc + (int64_t)l * width + c] = ww[l] * g;
            }
    }
#pragma unroll
    for (int l = 0; l < WSUM_MAX_LAYERS; ++l) {
        float v = acc[l];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][l] = v;
    }
    __syncthreads();
    if (threadIdx.x < WSUM_MAX_LAYERS)
        partials[(int64_t)blockIdx.x * WSUM_MAX_LAYERS + threadIdx.x] =
            ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

__global__ __launch_bounds__(64) void reduce_layers_wsum_dw_kernel(const float *__restrict__ partials, int n_blocks, int n_layers,
                                                                   const float *__restrict__ w, float *__restrict__ dw) {
    const int l = blockIdx.x;                               // one wavefront per layer weight
    float v = 0.f;
    for (int b = threadIdx.x; b < n_blocks; b += 64) v += partials[(int64_t)b * WSUM_MAX_LAYERS + l];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (threadIdx.x == 0 && l < n_layers) dw[l] = 2.f * w[l] * v;
}

unsigned grid_for(int64_t total) {
    int64_t b = (total + 255) / 256;
    return (unsigned)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

// dst[index[t] * ldd] = src[t]: the second half of a two-level permutation (models/basic.py:PairPlan).  The pair stage leaves its
// scores in `src` grouped by WINDOW of their final position (window_off[w] .. window_off[w + 1]: a narrow range of dst each), so a
// window's lines of dst are completed inside one L2 and written back whole — 12 M single-word stores at random over 48 MB are each
// a line fetched and written back (0.15 ms as a pass of its own, 0.19 ms inside the pair stage).  Workgroup b works through the
// windows w = b % 8, b % 8 + 8, ... with the other workgroups of equal b % 8 (one XCD under round-robin placement: speed only).
__global__ __launch_bounds__(256) void scatter_windows_kernel(const float *__restrict__ src, const int32_t *__restrict__ index,
                                                              float *__restrict__ dst, int64_t ldd, int64_t n,
                                                              const int32_t *__restrict__ window_off, int n_windows, int per_xcd) {
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
    for (int w = x; w < n_windows; w += 8) {
        const int64_t lo = window_off ? window_off[w] : n * w / n_windows, hi = window_off ? window_off[w + 1] : n * (w + 1) / n_windows;
        const int64_t step = (int64_t)per_xcd * 256;
        int64_t t = lo + (int64_t)j * 256 + threadIdx.x;
        for (; t + 3 * step < hi; t += 4 * step) {           // four independent loads in flight per lane
            const int32_t i0 = index[t], i1 = index[t + step], i2 = index[t + 2 * step], i3 = index[t + 3 * step];
            const float v0 = src[t], v1 = src[t + step], v2 = src[t + 2 * step], v3 = src[t + 3 * step];
            dst[(int64_t)i0 * ldd] = v0; dst[(int64_t)i1 * ldd] = v1; dst[(int64_t)i2 * ldd] = v2; dst[(int64_t)i3 * ldd] = v3;
        }
        for (; t < hi; t += step) dst[(int64_t)index[t] * ldd] = src[t];
    }
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

int amar_reduce_layers_wsum_f32(const float *cat, int64_t ld, int32_t n_layers, int32_t width, const float *w, float *out, int64_t ldo,
                                int64_t n_rows, amar_stream_t stream) {
    if (n_rows < 0 || n_layers < 1 || width < 1 || !cat || !w || !out || ld < (int64_t)n_layers * width || ldo < width)
        return AMAR_EINVAL;
    if (n_layers > WSUM_MAX_LAYERS) return AMAR_EUNSUPPORTED;
    if (n_rows == 0) return AMAR_OK;
    hipLaunchKernelGGL(reduce_layers_wsum_kernel, dim3(grid_for(n_rows * width)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), cat, ld, n_layers, width, w, out, ldo, n_rows);
    return amar_check_launch();
}

int64_t amar_reduce_layers_wsum_bwd_scratch(void) { return (int64_t)WSUM_BWD_BLOCKS * WSUM_MAX_LAYERS; }

int amar_reduce_layers_wsum_bwd_f32(const float *cat, int64_t ld, int32_t n_layers, int32_t width, const float *w,
                                    const float *d_out, int64_t ldd, float *d_cat, int64_t ld_dcat, float *dw, float *scratch,
                                    int64_t n_rows, amar_stream_t stream) {
    if (n_rows < 0 || n_layers < 1 || width < 1 || !cat || !w || !d_out || !d_cat || !dw || !scratch ||
        ld < (int64_t)n_layers * width || ld_dcat < (int64_t)n_layers * width || ldd < width)
        return AMAR_EINVAL;
    if (n_layers > WSUM_MAX_LAYERS) return AMAR_EUNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int64_t blocks = (n_rows * width + 255) / 256;
    if (blocks > WSUM_BWD_BLOCKS) blocks = WSUM_BWD_BLOCKS;
    if (blocks < 1) blocks = 1;                              // (no rows: every partial is 0, dw = 0)
    hipLaunchKernelGGL(reduce_layers_wsum_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, cat, ld, n_layers, width, w, d_out, ldd,
                       d_cat, ld_dcat, n_rows, scratch);
    hipLaunchKernelGGL(reduce_layers_wsum_dw_kernel, dim3((unsigned)n_layers), dim3(64), 0, st, scratch, (int)blocks, n_layers, w, dw);
    return amar_check_launch();
}

int amar_scatter_f32(const float *src, const int32_t *index, float *dst, int64_t ldd, int64_t n, const int32_t *window_off,
                     int32_t n_windows, amar_stream_t stream) {
    if (n < 0 || !src || !index || !dst || ldd < 1 || n_windows < 1 || n >= (1ll << 31)) return AMAR_EINVAL;
    if (n == 0) return AMAR_OK;
    static const int per_xcd_env = getenv("AMAR_SCATTER_WG") ? atoi(getenv("AMAR_SCATTER_WG")) : 0;
    const int per_xcd = per_xcd_env > 0 ? per_xcd_env : 32;   // 256 workgroups (ml1m(s=64): 0.060 ms against 0.064 at 512, 0.082 at 1 024)
    hipLaunchKernelGGL(scatter_windows_kernel, dim3(8 * per_xcd), dim3(256), 0, static_cast<hipStream_t>(stream), src, index, dst, ldd, n,
                       window_off, n_windows, per_xcd);
    return amar_check_launch();
}
}  // extern "C"
