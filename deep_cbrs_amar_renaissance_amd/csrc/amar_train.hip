// Training-step kernels: what Keras' fit() adds around the forward hot path.
// Reference semantics: binary_crossentropy + L2 regularisers + Adam as configured in
// config.yaml:50-58 and applied by experiment.py:155-188; reverse-mode derivatives of the Dense /
// GCNConv / LightGCNConv / embedding_lookup operations of src/models/{basic,gnn}.py.
// All element-wise or small-reduction work: HBM-bound, one float (or a few) per lane, fixed
// summation orders (the weight gradient is reduced in two stages, never with float atomics; the embedding-row gradient of a batch is
// added by the first position of every id, in position order: scatter_add_rows_owner_kernel).  Only an id list longer than 8 192 rows
// falls back to global float atomics (scatter_add_rows_kernel), whose last bits depend on the order of arrival; ids are range-checked
// on the host (engine.ids_to_device).
#include "amar_common.h"
#include <stdlib.h>

namespace {

// rows per partial block of the weight-gradient kernel: 128 for a batch-sized operand (a fixed 512 gave 18 blocks for a 1 024-row
// batch and a 48 x 48 layer: 22 us per call, a third of a training batch at ml1m(s=1)), growing to 512 so that node-table-sized
// operands do not multiply the partials the second stage has to add
__host__ __device__ inline int wg_rows(int64_t M) {
    const int64_t r = ((M / 64 + 63) / 64) * 64;
    return (int)(r < 128 ? 128 : r > 512 ? 512 : r);
}

__device__ __forceinline__ float act_grad(float dy, float y, int act) {
    if (act == AMAR_ACT_RELU) return y > 0.f ? dy : 0.f;
    if (act == AMAR_ACT_SIGMOID) return dy * y * (1.f - y);
    return dy;
}

__global__ __launch_bounds__(256) void act_bwd_kernel(const float *__restrict__ dY, int64_t ldd, const float *__restrict__ Y,
                                                      int64_t ldy, float *__restrict__ dZ, int64_t ldz, int64_t M, int N, int act) {
    const int64_t total = M * N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / N;
        const int c = (int)(i - r * N);
        dZ[r * ldz + c] = act_grad(dY[r * ldd + c], Y[r * ldy + c], act);
    }
}

// partial[chunk][k][n] = sum over the chunk's rows of X[m][k] * dZ[m][n]; partial_b[chunk][n] = sum dZ[m][n]
// block = 16 x 16 threads = one 16x16 tile of dW; grid = (row chunks, K tiles, N tiles)
__global__ __launch_bounds__(256) void wgrad_partial_kernel(const float *__restrict__ X, int64_t ldx, const float *__restrict__ dZ,
                                                            int64_t ldz, int64_t M, int K, int N, float *__restrict__ part_w,
                                                            float *__restrict__ part_b, int WG_ROWS) {
    __shared__ float xs[64][17], zs[64][17];
    const int tk = threadIdx.x >> 4, tn = threadIdx.x & 15;
    const int k0 = blockIdx.y * 16, n0 = blockIdx.z * 16;
    const int64_t m_beg = (int64_t)blockIdx.x * WG_ROWS, m_end = min(M, m_beg + WG_ROWS);
    float acc = 0.f, accb = 0.f;
    for (int64_t m0 = m_beg; m0 < m_end; m0 += 64) {
        __syncthreads();
        for (int e = threadIdx.x; e < 64 * 16; e += 256) {
            const int r = e >> 4, c = e & 15;
            const int64_t m = m0 + r;
            xs[r][c] = (X && m < m_end && k0 + c < K) ? X[m * ldx + k0 + c] : 0.f;
            zs[r][c] = (m < m_end && n0 + c < N) ? dZ[m * ldz + n0 + c] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int r = 0; r < 64; ++r) {
            acc = fmaf(xs[r][tk], zs[r][tn], acc);
            accb += zs[r][tn];
        }
    }
    if (part_w && k0 + tk < K && n0 + tn < N) part_w[((int64_t)blockIdx.x * K + k0 + tk) * N + n0 + tn] = acc;
    if (part_b && blockIdx.y == 0 && tk == 0 && n0 + tn < N) part_b[(int64_t)blockIdx.x * N + n0 + tn] = accb;
}

// out[e] = part[0][e] + part[1][e] + ... in chunk order (fixed order -> reproducible), optionally accumulated into out
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float *__restrict__ part, int n_chunks, int64_t size,
                                                              float *__restrict__ out) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < size; e += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int c = 0; c < n_chunks; ++c) s += part[(int64_t)c * size + e];
        out[e] = s;
    }
}

// the same for the weight and the bias partials of one wgrad call in ONE launch (elements [0, size_w) and [size_w, size_w + size_b))
__global__ __launch_bounds__(256) void reduce_partials2_kernel(const float *__restrict__ part_w, int64_t size_w, float *__restrict__ out_w,
                                                               const float *__restrict__ part_b, int64_t size_b, float *__restrict__ out_b,
                                                               int n_chunks) {
    const int64_t total = size_w + size_b;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const bool is_w = e < size_w;
        const float *part = is_w ? part_w : part_b;
        const int64_t size = is_w ? size_w : size_b, idx = is_w ? e : e - size_w;
        float s = 0.f;
        for (int c = 0; c < n_chunks; ++c) s += part[(int64_t)c * size + idx];
        (is_w ? out_w : out_b)[idx] = s;
    }
}

// Keras backend binary_crossentropy on probabilities: p clipped to [eps, 1-eps], log(p + eps); mean over the batch.
__global__ __launch_bounds__(256) void bce_grad_kernel(const float *__restrict__ p, int64_t ldp, const float *__restrict__ y,
                                                       float *__restrict__ dz, float *__restrict__ loss_terms, int64_t B) {
    const float eps = 1e-7f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < B; i += (int64_t)gridDim.x * blockDim.x) {
        const float pi = p[i * ldp], yi = y[i];
        const float pc = fminf(fmaxf(pi, eps), 1.f - eps);
        loss_terms[i] = -(yi * logf(pc + eps) + (1.f - yi) * logf(1.f - pc + eps));
        const bool inside = pi >= eps && pi <= 1.f - eps;
        const float dp = inside ? -(yi / (pc + eps) - (1.f - yi) / (1.f - pc + eps)) / (float)B : 0.f;
        dz[i] = dp * pi * (1.f - pi);                                // through the sigmoid of the last Dense layer
    }
}

__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const float *__restrict__ src, int64_t lds, const int32_t *__restrict__ ids,
                                                               int base, float *__restrict__ dst, int64_t ldd, int64_t M, int W) {
    const int64_t total = M * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / W;
        const int c = (int)(i - r * W);
        atomicAdd(dst + ((int64_t)ids[r] - base) * ldd + c, src[r * lds + c]);
    }
}

// The same sum WITHOUT atomics for batch-sized calls (M <= SCATTER_OWNER_MAX ids, held in LDS): the FIRST position of every distinct id
// owns its destination row and adds the rows of all positions with that id in ascending position order — one writer per row, a fixed
// order of additions: the embedding-row gradient no longer depends on the order in which workgroups arrive.  One wavefront per
// position: its 64 lanes compare 64 ids of the list at a time (a ballot tells whether an earlier position holds the id, later ballots
// list the positions to add), then take one column each.  (First form: one thread per position walking the list alone — a serial chain
// of 1 024 LDS reads, 85 us per call against 4.5 for the atomics; this one costs about the same as the atomics.)
constexpr int SCATTER_OWNER_MAX = 8192;
__global__ __launch_bounds__(256) void scatter_add_rows_owner_kernel(const float *__restrict__ src, int64_t lds, const int32_t *__restrict__ ids,
                                                                     int base, float *__restrict__ dst, int64_t ldd, int M, int W) {
    extern __shared__ int32_t id_lds[];
    for (int q = threadIdx.x; q < M; q += blockDim.x) id_lds[q] = ids[q];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int p = blockIdx.x * 4 + wave; p < M; p += gridDim.x * 4) {          // wave-uniform
        const int32_t id = id_lds[p];
        bool owner = true;
        for (int q0 = 0; q0 < p && owner; q0 += 64) {
            const int q = q0 + lane;
            owner = __ballot(q < p && id_lds[q] == id) == 0ull;
        }
        if (!owner) continue;
        for (int c0 = 0; c0 < W; c0 += 64) {
            const int c = c0 + lane;
            float acc = c < W ? src[(int64_t)p * lds + c] : 0.f;
            for (int q0 = (p + 1) & ~63; q0 < M; q0 += 64) {
                const int q = q0 + lane;
                unsigned long long later = __ballot(q > p && q < M && id_lds[q] == id);
                while (later) {                                                 // ascending positions
                    const int b = __builtin_ctzll(later);
                    later &= later - 1;
                    if (c < W) acc += src[(int64_t)(q0 + b) * lds + c];
                }
            }
            if (c < W) dst[((int64_t)id - base) * ldd + c] += acc;
        }
    }
}

__global__ __launch_bounds__(256) void add_inplace_kernel(float *__restrict__ dst, int64_t ldd, const float *__restrict__ src, int64_t lds,
                                                          int64_t M, int W, float scale) {
    const int64_t total = M * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / W;
        const int c = (int)(i - r * W);
        dst[r * ldd + c] += scale * src[r * lds + c];
    }
}

__global__ __launch_bounds__(256) void transpose_kernel(const float *__restrict__ src, int K, int N, float *__restrict__ dst) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < K * N; e += gridDim.x * blockDim.x) {
        const int k = e / N, n = e - k * N;
        dst[n * K + k] = src[e];
    }
}

// keras.optimizers.Adam: m, v moments; lr_t carries the bias correction; the L2 regulariser's gradient 2*l2*w is folded in
__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ w, const float *__restrict__ g, float *__restrict__ m,
                                                   float *__restrict__ v, int64_t n, float lr_t, float b1, float b2, float eps,
                                                   float l2x2) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float wi = w[i];
        const float gi = g[i] + l2x2 * wi;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        w[i] = wi - lr_t * mi / (sqrtf(vi) + eps);
    }
}

// out = (A + B) * scale[row]   (B optional): GraphSAGE's mean aggregate (s + x) / count and its reverse
__global__ __launch_bounds__(256) void row_affine_kernel(const float *__restrict__ A, int64_t lda, const float *__restrict__ B,
                                                         int64_t ldb, const float *__restrict__ scale, float *__restrict__ out,
                                                         int64_t ldo, int64_t M, int W) {
    const int64_t total = M * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / W;
        const int c = (int)(i - r * W);
        float v = A[r * lda + c];
        if (B) v += B[r * ldb + c];
        out[r * ldo + c] = v * scale[r];
    }
}

// tf.nn.l2_normalize(axis=-1) followed by the activation (GraphSageConv): inv = rsqrt(max(sum z^2, 1e-12)),
// n = z * inv (kept for the reverse pass), y = relu(n).  One thread per row: rows are a few dozen floats.
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float *__restrict__ Z, int64_t ldz, float *__restrict__ Nrm,
                                                         int64_t ldn, float *__restrict__ inv, float *__restrict__ Y,
                                                         int64_t ldy, int64_t M, int C, int relu) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < M; r += (int64_t)gridDim.x * blockDim.x) {
        const float *z = Z + r * ldz;
        float sq = 0.f;
        for (int c = 0; c < C; ++c) sq = fmaf(z[c], z[c], sq);
        const float iv = rsqrtf(fmaxf(sq, 1e-12f));
        inv[r] = iv;
        for (int c = 0; c < C; ++c) {
            const float n = z[c] * iv;
            Nrm[r * ldn + c] = n;
            Y[r * ldy + c] = relu ? fmaxf(n, 0.f) : n;
        }
    }
}

// reverse of the above: dn = dy * [n > 0];  dz = inv * (dn - n * (n . dn)); where the norm was clamped (inv = 1e6)
// z * inv is linear in z and dz = inv * dn.
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float *__restrict__ dY, int64_t ldd, const float *__restrict__ Nrm,
                                                         int64_t ldn, const float *__restrict__ inv, float *__restrict__ dZ,
                                                         int64_t ldz, int64_t M, int C, int relu) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < M; r += (int64_t)gridDim.x * blockDim.x) {
        const float *n = Nrm + r * ldn, *dy = dY + r * ldd;
        const float iv = inv[r];
        float dot = 0.f;
        for (int c = 0; c < C; ++c) {
            const float dn = (!relu || n[c] > 0.f) ? dy[c] : 0.f;
            dot = fmaf(n[c], dn, dot);
        }
        if (!(iv < 1e6f)) dot = 0.f;
        for (int c = 0; c < C; ++c) {
            const float dn = (!relu || n[c] > 0.f) ? dy[c] : 0.f;
            dZ[r * ldz + c] = iv * (dn - n[c] * dot);
        }
    }
}

// ---- GAT reverse pass (Spektral GATConv._call_single, 1 head; forward: amar_gat_layer_f32) ----------------
// With out_i = sum_j alpha_ij h_j + b, alpha = softmax_j(LeakyReLU(s_i + t_j)) (s = h.a_self, t = h.a_neigh) and
// g_i = dL/dout_i (ReLU mask applied):   d alpha_ij = g_i . h_j,   sum_k alpha_ik d alpha_ik = g_i . (out_i - b) =: c_i,
//   d e_ij = alpha_ij (d alpha_ij - c_i),  d pre_ij = d e_ij * LeakyReLU'(s_i + t_j),
//   ds_i = sum_j d pre_ij,   dt_j = sum_i d pre_ij,   dh_j = sum_i alpha_ij g_i + ds_j a_self + dt_j a_neigh.
// The edge multiset is symmetric, so "targets i that have j as a source" is CSR row j: both sums run row-wise, one
// wavefront per node, without float atomics.  Kernel 1 (node as target) recomputes the softmax statistics.
struct GatBwdArgs {
    const int32_t *rowptr; const int32_t *colidx; const float *H; int64_t ldh; const float *s_self; const float *s_neigh;
    const float *Y; int64_t ldy; const float *dY; int64_t ldd; const float *bias; const float *a_self; const float *a_neigh;
    float *dout; float *row_max; float *row_inv; float *row_c; float *ds; float *dt; float *dH; int64_t lddh;
    int self_loop; int n_rows; int C;
};

__device__ __forceinline__ float leaky02(float x) { return x > 0.f ? x : 0.2f * x; }
__device__ __forceinline__ float dot4(const float4 &a, const float4 &b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
template <int LPN>
__device__ __forceinline__ float group_sum(float v) {               // over the LPN consecutive lanes of a slot
#pragma unroll
    for (int m = 1; m < LPN; m <<= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// LPN = lanes per node = C / 4 rounded up to a power of two (the group sums are xor-shuffles); for widths in between
// (C = 24: 6 of 8 lanes) the surplus lanes carry zeros and store nothing.
template <int LPN>
__global__ __launch_bounds__(256) void gat_bwd_target_kernel(const GatBwdArgs a) {
    constexpr int NS = AMAR_WAVE / LPN;
    const int C = a.C;
    const int lane = threadIdx.x & (AMAR_WAVE - 1);
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.n_rows) return;
    const int q = lane % LPN, slot = lane / LPN;
    const bool live = 4 * q < C;
    const int beg = a.rowptr[row], end = a.rowptr[row + 1];
    const float si = a.s_self[row];
    const float4 y = live ? *reinterpret_cast<const float4 *>(a.Y + (int64_t)row * a.ldy + 4 * q) : f4_zero();
    const float4 dy = live ? *reinterpret_cast<const float4 *>(a.dY + (int64_t)row * a.ldd + 4 * q) : f4_zero();
    const float4 b = live ? *reinterpret_cast<const float4 *>(a.bias + 4 * q) : f4_zero();
    const float4 g = make_float4(y.x > 0.f ? dy.x : 0.f, y.y > 0.f ? dy.y : 0.f, y.z > 0.f ? dy.z : 0.f, y.w > 0.f ? dy.w : 0.f);
    if (slot == 0 && live) *reinterpret_cast<float4 *>(a.dout + (int64_t)row * C + 4 * q) = g;
    const float ci = wave_sum_stride<1>(slot == 0 ? dot4(g, make_float4(y.x - b.x, y.y - b.y, y.z - b.z, y.w - b.w)) : 0.f);
    // softmax statistics as in the forward kernel
    float mn = a.self_loop ? a.s_neigh[row] : -INFINITY;
    for (int p = beg + lane; p < end; p += AMAR_WAVE) mn = fmaxf(mn, a.s_neigh[a.colidx[p]]);
    mn = wave_max_all(mn);
    const float emax = leaky02(si + mn);
    float den = 0.f;
    for (int p = beg + lane; p < end; p += AMAR_WAVE) den += expf(leaky02(si + a.s_neigh[a.colidx[p]]) - emax);
    if (a.self_loop && lane == 0) den += expf(leaky02(si + a.s_neigh[row]) - emax);
    den = wave_sum_stride<1>(den);
    const float inv = 1.f / (den + 1e-9f);
    float ds = 0.f;
    auto edge = [&](int j) {
        const float pre = si + a.s_neigh[j];
        const float alpha = expf(leaky02(pre) - emax) * inv;
        const float4 h = live ? *reinterpret_cast<const float4 *>(a.H + (int64_t)j * a.ldh + 4 * q) : f4_zero();
        const float dalpha = group_sum<LPN>(dot4(g, h));
        const float dpre = alpha * (dalpha - ci) * (pre > 0.f ? 1.f : 0.2f);
        if (q == 0) ds += dpre;
    };
    // every lane of a slot group takes part in the shuffles: the loop bound is uniform per group
    for (int p = beg + slot; p < end; p += NS) edge(a.colidx[p]);
    if (a.self_loop && slot == 0) edge(row);
    ds = wave_sum_stride<1>(ds);
    if (lane == 0) { a.ds[row] = ds; a.row_max[row] = emax; a.row_inv[row] = inv; a.row_c[row] = ci; }
}

template <int LPN>
__global__ __launch_bounds__(256) void gat_bwd_source_kernel(const GatBwdArgs a) {
    constexpr int NS = AMAR_WAVE / LPN;
    const int C = a.C;
    const int lane = threadIdx.x & (AMAR_WAVE - 1);
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.n_rows) return;
    const int q = lane % LPN, slot = lane / LPN;
    const bool live = 4 * q < C;
    const int beg = a.rowptr[row], end = a.rowptr[row + 1];
    const float tj = a.s_neigh[row];
    const float4 h = live ? *reinterpret_cast<const float4 *>(a.H + (int64_t)row * a.ldh + 4 * q) : f4_zero();
    float4 acc = f4_zero();
    float dt = 0.f;
    auto edge = [&](int i) {
        const float pre = a.s_self[i] + tj;
        const float alpha = expf(leaky02(pre) - a.row_max[i]) * a.row_inv[i];
        const float4 g = live ? *reinterpret_cast<const float4 *>(a.dout + (int64_t)i * C + 4 * q) : f4_zero();
        acc = f4_fma(alpha, g, acc);
        const float dalpha = group_sum<LPN>(dot4(g, h));
        const float dpre = alpha * (dalpha - a.row_c[i]) * (pre > 0.f ? 1.f : 0.2f);
        if (q == 0) dt += dpre;
    };
    for (int p = beg + slot; p < end; p += NS) edge(a.colidx[p]);
    if (a.self_loop && slot == 0) edge(row);
    acc = f4_wave_sum_stride<LPN>(acc);
    dt = wave_sum_stride<1>(dt);
    if (slot == 0 && live) {
        const float ds = a.ds[row];
        const float4 as = *reinterpret_cast<const float4 *>(a.a_self + 4 * q);
        const float4 an = *reinterpret_cast<const float4 *>(a.a_neigh + 4 * q);
        *reinterpret_cast<float4 *>(a.dH + (int64_t)row * a.lddh + 4 * q) =
            make_float4(acc.x + ds * as.x + dt * an.x, acc.y + ds * as.y + dt * an.y,
                        acc.z + ds * as.z + dt * an.z, acc.w + ds * as.w + dt * an.w);
    }
    if (lane == 0) a.dt[row] = dt;
}

template <int LPN>
void launch_gat_bwd(const GatBwdArgs &a, hipStream_t st) {
    const dim3 grid((a.n_rows + 3) / 4), block(256);
    hipLaunchKernelGGL(gat_bwd_target_kernel<LPN>, grid, block, 0, st, a);
    hipLaunchKernelGGL(gat_bwd_source_kernel<LPN>, grid, block, 0, st, a);
}

// The same update with the step size read from device memory, and the one-thread kernel that advances it:
// state[0] = t (as float), state[1] = lr * sqrt(1 - b2^t) / (1 - b1^t).  Lets a whole training batch, optimizer
// included, replay as one hipGraph with nothing baked in that changes from step to step.
__global__ void adam_advance_kernel(float *__restrict__ state, float lr, float b1, float b2) {
    const double t = (double)state[0] + 1.0;
    state[0] = (float)t;
    state[1] = (float)((double)lr * sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t)));
}

__global__ __launch_bounds__(256) void adam_dev_kernel(float *__restrict__ w, const float *__restrict__ g, float *__restrict__ m,
                                                       float *__restrict__ v, int64_t n, const float *__restrict__ state, float b1,
                                                       float b2, float eps, float l2x2) {
    const float lr_t = state[1];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float wi = w[i];
        const float gi = g[i] + l2x2 * wi;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        w[i] = wi - lr_t * mi / (sqrtf(vi) + eps);
    }
}

// ---- FusionLayer('attention') (src/layers/fusion.py:54-68) and the residual head's sum (src/models/hybrid.py:89) ----
// With ta = a . W_att, tb = b . W_att (two Dense products done by the caller):  the softmax over the two stacked
// sources is, per feature, wa = sigmoid(tanh(ta) - tanh(tb)), wb = 1 - wa, and out = wa * a + wb * b.
__global__ __launch_bounds__(256) void attention_mix_kernel(const float *__restrict__ A, int64_t lda, const float *__restrict__ B,
                                                            int64_t ldb, const float *__restrict__ TA, int64_t ldta,
                                                            const float *__restrict__ TB, int64_t ldtb, float *__restrict__ out,
                                                            int64_t ldo, int64_t M, int D) {
    const int64_t total = M * D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / D;
        const int c = (int)(i - r * D);
        const float a = A[r * lda + c], b = B[r * ldb + c];
        const float wa = 1.f / (1.f + expf(tanhf(TB[r * ldtb + c]) - tanhf(TA[r * ldta + c])));
        out[r * ldo + c] = wa * a + (1.f - wa) * b;
    }
}

// reverse: dA = dOut * wa, dB = dOut * wb (the direct paths), dTA = g * (1 - tanh(ta)^2), dTB = -g * (1 - tanh(tb)^2)
// with g = dOut * (a - b) * wa * wb; the caller adds dTA . W^T / dTB . W^T to dA / dB and forms dW.
__global__ __launch_bounds__(256) void attention_mix_bwd_kernel(const float *__restrict__ dOut, int64_t ldd, const float *__restrict__ A,
                                                                int64_t lda, const float *__restrict__ B, int64_t ldb,
                                                                const float *__restrict__ TA, int64_t ldta, const float *__restrict__ TB,
                                                                int64_t ldtb, float *__restrict__ dA, float *__restrict__ dB,
                                                                float *__restrict__ dTA, float *__restrict__ dTB, int64_t M, int D) {
    const int64_t total = M * D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / D;
        const int c = (int)(i - r * D);
        const float a = A[r * lda + c], b = B[r * ldb + c], d = dOut[r * ldd + c];
        const float ca = tanhf(TA[r * ldta + c]), cb = tanhf(TB[r * ldtb + c]);
        const float wa = 1.f / (1.f + expf(cb - ca)), wb = 1.f - wa;
        const float g = d * (a - b) * wa * wb;
        dA[i] = d * wa; dB[i] = d * wb;                                 // contiguous [M, D] outputs
        dTA[i] = g * (1.f - ca * ca); dTB[i] = -g * (1.f - cb * cb);
    }
}

// LocalityAdaptive of DGCFConv (src/layers/dgcf_conv.py:83-102): out = X * sigmoid(w[row]), w [n, 1] trainable
__global__ __launch_bounds__(256) void locality_scale_kernel(const float *__restrict__ X, int64_t ldx, const float *__restrict__ w,
                                                             float *__restrict__ out, int64_t ldo, int64_t M, int W) {
    const int64_t total = M * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / W;
        const int c = (int)(i - r * W);
        out[r * ldo + c] = X[r * ldx + c] / (1.f + expf(-w[r]));
    }
}

// reverse: dX (+)= dOut * s, dw[row] = s (1 - s) * sum_c dOut[row, c] X[row, c], s = sigmoid(w[row]); one thread per row
__global__ __launch_bounds__(256) void locality_scale_bwd_kernel(const float *__restrict__ dOut, int64_t ldd, const float *__restrict__ X,
                                                                 int64_t ldx, const float *__restrict__ w, float *__restrict__ dX,
                                                                 int64_t lddx, float *__restrict__ dw, int64_t M, int W, int accumulate) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < M; r += (int64_t)gridDim.x * blockDim.x) {
        const float s = 1.f / (1.f + expf(-w[r]));
        float dot = 0.f;
        for (int c = 0; c < W; ++c) {
            const float d = dOut[r * ldd + c];
            dot = fmaf(d, X[r * ldx + c], dot);
            const float v = d * s;
            dX[r * lddx + c] = accumulate ? dX[r * lddx + c] + v : v;
        }
        dw[r] = s * (1.f - s) * dot;
    }
}

// out = act(a + b + c): the residual head's activation(residual(x) + x1 + x2)
__global__ __launch_bounds__(256) void add3_act_kernel(const float *__restrict__ A, int64_t lda, const float *__restrict__ B, int64_t ldb,
                                                       const float *__restrict__ C, int64_t ldc, float *__restrict__ out, int64_t ldo,
                                                       int64_t M, int W, int act) {
    const int64_t total = M * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / W;
        const int c = (int)(i - r * W);
        float v = A[r * lda + c] + B[r * ldb + c] + C[r * ldc + c];
        if (act == AMAR_ACT_RELU) v = fmaxf(v, 0.f);
        else if (act == AMAR_ACT_SIGMOID) v = 1.f / (1.f + expf(-v));
        out[r * ldo + c] = v;
    }
}

// Every parameter of a model in ONE launch: block b works on 1024 elements of the slot that owns it (slots carry their
// first block).  Also adds reg_scale * l2 * sum(w^2) of the PRE-update weights to *loss_acc (the regularisation part of the
// batch loss Keras reports), so no separate reduction launches are needed.
__global__ __launch_bounds__(256) void adam_multi_kernel(const amar_adam_slot *__restrict__ slots, int n_slots,
                                                         const float *__restrict__ state, float b1, float b2, float eps,
                                                         float reg_scale, float *__restrict__ loss_acc) {
    int sidx = 0;
    while (sidx + 1 < n_slots && (int64_t)blockIdx.x >= slots[sidx + 1].first_block) ++sidx;
    const amar_adam_slot sl = slots[sidx];
    const float lr_t = state[1], l2x2 = 2.f * sl.l2;
    const int64_t base = ((int64_t)blockIdx.x - sl.first_block) * 1024;
    // all loads of the block's four elements per thread first, then the deferred partial gradients four groups at a time (loads before
    // adds, the adds in group order): written as one load -> add chain per element, the 16 partials of a Dense kernel's gradient were 16
    // dependent memory round trips — 30 us of every training batch, whatever the model's size.  Slots whose arrays allow it (16-byte
    // aligned, n a multiple of 4: every table and kernel of the models here) move 16 bytes per lane — a thread then owns four NEIGHBOURING
    // elements instead of four 256 apart; each element's arithmetic is the same either way.
    const bool vec = (sl.n & 3) == 0 && ((reinterpret_cast<uintptr_t>(sl.w) | reinterpret_cast<uintptr_t>(sl.g) | reinterpret_cast<uintptr_t>(sl.m) |
                                          reinterpret_cast<uintptr_t>(sl.v)) & 15u) == 0;
    float wi[4], gs[4], mi[4], vi[4];
    int64_t idx[4];
    if (vec) {
        const int64_t i0 = base + 4 * threadIdx.x;
        const bool ok = i0 < sl.n;
#pragma unroll
        for (int r = 0; r < 4; ++r) idx[r] = ok ? i0 + r : sl.n;
        const float4 w4 = ok ? *reinterpret_cast<const float4 *>(sl.w + i0) : f4_zero(), g4 = ok ? *reinterpret_cast<const float4 *>(sl.g + i0) : f4_zero();
        const float4 m4 = ok ? *reinterpret_cast<const float4 *>(sl.m + i0) : f4_zero(), v4 = ok ? *reinterpret_cast<const float4 *>(sl.v + i0) : f4_zero();
        wi[0] = w4.x; wi[1] = w4.y; wi[2] = w4.z; wi[3] = w4.w;  gs[0] = g4.x; gs[1] = g4.y; gs[2] = g4.z; gs[3] = g4.w;
        mi[0] = m4.x; mi[1] = m4.y; mi[2] = m4.z; mi[3] = m4.w;  vi[0] = v4.x; vi[1] = v4.y; vi[2] = v4.z; vi[3] = v4.w;
        for (int c = 1; c < sl.g_groups; c += 16) {                  // sixteen groups in flight
            float4 part[16];
#pragma unroll
            for (int cc = 0; cc < 16; ++cc)
                part[cc] = (c + cc < sl.g_groups && ok) ? *reinterpret_cast<const float4 *>(sl.g + (int64_t)(c + cc) * sl.n + i0) : f4_zero();
#pragma unroll
            for (int cc = 0; cc < 16; ++cc)
                if (c + cc < sl.g_groups) { gs[0] += part[cc].x; gs[1] += part[cc].y; gs[2] += part[cc].z; gs[3] += part[cc].w; }
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            idx[r] = base + r * 256 + threadIdx.x;
            const bool ok = idx[r] < sl.n;
            wi[r] = ok ? sl.w[idx[r]] : 0.f;
            gs[r] = ok ? sl.g[idx[r]] : 0.f;
            mi[r] = ok ? sl.m[idx[r]] : 0.f;
            vi[r] = ok ? sl.v[idx[r]] : 0.f;
        }
        for (int c = 1; c < sl.g_groups; c += 4) {
            float part[4][4];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    part[cc][r] = (c + cc < sl.g_groups && idx[r] < sl.n) ? sl.g[(int64_t)(c + cc) * sl.n + idx[r]] : 0.f;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
                if (c + cc < sl.g_groups) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) gs[r] += part[cc][r];
                }
        }
    }
    float sq = 0.f;
    float wo[4], mo[4], vo[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float gi = gs[r] + l2x2 * wi[r];
        mo[r] = b1 * mi[r] + (1.f - b1) * gi;
        vo[r] = b2 * vi[r] + (1.f - b2) * gi * gi;
        wo[r] = wi[r] - lr_t * mo[r] / (sqrtf(vo[r]) + eps);
        if (idx[r] < sl.n) sq = fmaf(wi[r], wi[r], sq);
    }
    if (vec) {
        if (idx[0] < sl.n) {
            *reinterpret_cast<float4 *>(sl.m + idx[0]) = make_float4(mo[0], mo[1], mo[2], mo[3]);
            *reinterpret_cast<float4 *>(sl.v + idx[0]) = make_float4(vo[0], vo[1], vo[2], vo[3]);
            *reinterpret_cast<float4 *>(sl.w + idx[0]) = make_float4(wo[0], wo[1], wo[2], wo[3]);
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (idx[r] < sl.n) { sl.m[idx[r]] = mo[r]; sl.v[idx[r]] = vo[r]; sl.w[idx[r]] = wo[r]; }
    }
    if (loss_acc && sl.l2 != 0.f) {                                  // block sum, one atomic per block
        __shared__ float red[4];
        sq = wave_sum_stride<1>(sq);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(loss_acc, reg_scale * sl.l2 * (red[0] + red[1] + red[2] + red[3]));
    }
}

// *acc += scale * sum(x[0..n))  — the batch's data loss into the running loss, one block
__global__ __launch_bounds__(256) void sum_into_kernel(const float *__restrict__ x, int64_t n, float scale, float *__restrict__ acc) {
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += x[i];
    __shared__ float red[4];
    s = wave_sum_stride<1>(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *acc += scale * (red[0] + red[1] + red[2] + red[3]);
}

unsigned grid1d(int64_t total) {
    int64_t b = (total + 255) / 256;
    return (unsigned)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

// ---- the reverse pass of ONE Dense layer in two launches (round 4) -------------------------------------------------------------------
// dZ = dY * act'(Y);  dX = dZ . W^T;  dW = X^T . dZ;  db = column sums of dZ  — what act_bwd + wgrad (two launches) + dense(W^T) did in
// four launches of 5-10 us each, ~30 launches of a ~90-launch training batch at ml1m(s=1) (a batch is launch-latency, not throughput).
// A workgroup owns 64 S rows (S sub-tiles of 64, so that at most 32 workgroups leave partials): X, dZ and W tiles in LDS, both products on
// v_mfma_f32_16x16x4_f32 (round 3's attempt did them with scalar LDS-fed FMAs), dW accumulated in registers over the sub-tiles; its
// K x N partial of dW and its partial of db go to the workspace, which reduce_partials2_kernel adds in workgroup order — a fixed
// order: the gradients are reproducible bit for bit, no float atomics.  (A first version let the LAST workgroup to finish add the
// partials behind a ticket: one workgroup adding 16 x 2 352 values from other XCDs' L2s took longer than the launch it saved —
// 0.74 s per epoch against 0.50 with the separate kernels.)
constexpr int DB_ROWS = 64, DB_MAXD = 128, DB_THREADS = 256;
// partials a call leaves for its consumer (the Adam launch adds them, one thread per element, in group order, sixteen loads in flight).
// 64 = the tiles of a 4 096-row batch.  It was 256, added four at a time: the weight gradient of a convolution layer over the 9 228 nodes
// of ml1m(s=1) then reached the Adam launch as 145 partials (243 at s=64) of 64-256 elements — one workgroup adding them in dependent
// rounds: 30 us of a 37 us launch (s=1), found only when the launch was timed on a graph 64 times larger and did not get 64 times longer.
constexpr int DB_MAX_GROUPS = 64;
typedef float v4f __attribute__((ext_vector_type(4)));

// acc += A[., k] B[k, .] over k = 0 .. n - 1 (n a multiple of 16) on v_mfma_f32_16x16x4_f32, the lane's A element of step k at ap[k * as],
// its B element at bp[k * bs].  The eight LDS reads of four instructions are issued together, and those of the NEXT four before the
// current four run: written as `for (k += 4) acc = mfma(ap[k], bp[k], acc)` the compiler kept one instruction per iteration behind
// its own two reads (#pragma unroll refused: "loop not unrolled") — ~200 clocks per step, 2 400 per 16 x 16 x 48 tile, which is where
// the Dense-stack kernels of a training batch spent their time (tools/exp_ds_stamps.py).  Same order of accumulation: same bits.
__device__ __forceinline__ v4f mfma_chain16(const float *ap, const int as, const float *bp, const int bs, const int n, v4f acc) {
    float a0 = ap[0], a1 = ap[4 * as], a2 = ap[8 * as], a3 = ap[12 * as];
    float b0 = bp[0], b1 = bp[4 * bs], b2 = bp[8 * bs], b3 = bp[12 * bs];
    for (int k = 16; k < n; k += 16) {
        const float c0 = ap[k * as], c1 = ap[(k + 4) * as], c2 = ap[(k + 8) * as], c3 = ap[(k + 12) * as];
        const float d0 = bp[k * bs], d1 = bp[(k + 4) * bs], d2 = bp[(k + 8) * bs], d3 = bp[(k + 12) * bs];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b3, acc, 0, 0, 0);
        a0 = c0; a1 = c1; a2 = c2; a3 = c3; b0 = d0; b1 = d1; b2 = d2; b3 = d3;
    }
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b2, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b3, acc, 0, 0, 0);
    return acc;
}

struct DenseBwdArgs {
    const float *X; int64_t ldx; const float *Y; int64_t ldy; const float *dY; int64_t lddy; const float *W;
    float *dX; int64_t lddx; float *part_w; float *part_b; int64_t M; int K, N, act, subtiles;
    float *dZ; int64_t lddz; int accum_dx;
    int x_scalar;                                                     // (row-walking kernel: X rows by single floats — K not a multiple of 4 or an unaligned X)
};

// How a call over M rows is cut (round 4, second half).  Up to DB_MAX_GROUPS tiles of 64 rows: one workgroup and one partial per tile (the
// batch-sized calls of a training step; the Adam launch adds the partials).  Beyond that — the reverse pass of a convolution layer runs
// over every NODE of the graph, 590 592 rows at ml1m(s=64) — the launch still takes one workgroup per tile (two tiles past 8 192
// workgroups; the row-walking kernel below: `fold` workgroups per partial) and a second launch folds the raw partials, `fold` at a time
// and in workgroup order, into at most DB_MAX_GROUPS: with 256 workgroups walking 36 tiles each, one after the other behind a barrier,
// such a call took 119 us for 57 MB of operands.
constexpr int DB_MAX_LAUNCH_GROUPS = 8192;
struct DenseBwdPlan { int sub; int64_t launch_groups; int fold; int64_t out_groups; };
inline DenseBwdPlan dense_bwd_plan(int64_t M) {
    const int64_t tiles = (M + DB_ROWS - 1) / DB_ROWS;
    DenseBwdPlan p;
    p.sub = (int)((tiles + DB_MAX_LAUNCH_GROUPS - 1) / DB_MAX_LAUNCH_GROUPS);
    if (p.sub < 1) p.sub = 1;
    p.launch_groups = (tiles + p.sub - 1) / p.sub;
    p.fold = p.launch_groups > DB_MAX_GROUPS ? (int)((p.launch_groups + DB_MAX_GROUPS - 1) / DB_MAX_GROUPS) : 1;
    p.out_groups = (p.launch_groups + p.fold - 1) / p.fold;
    return p;
}

// out[g][e] = raw[g fold][e] + raw[g fold + 1][e] + ... (the last group may be short), for the weight and the bias partials in one launch
__global__ __launch_bounds__(256) void fold_partials2_kernel(const float *__restrict__ raw_w, int64_t size_w, float *__restrict__ out_w,
                                                             const float *__restrict__ raw_b, int64_t size_b, float *__restrict__ out_b,
                                                             int64_t n_raw, int fold, int64_t n_out) {
    const int64_t per = size_w + size_b, total = n_out * per;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t g = t / per, e = t - g * per;
        const bool is_w = e < size_w;
        const int64_t size = is_w ? size_w : size_b, idx = is_w ? e : e - size_w;
        const float *src = (is_w ? raw_w : raw_b) + g * fold * size + idx;
        const int64_t n = n_raw - g * fold < fold ? n_raw - g * fold : fold;
        float s = 0.f;
        int64_t c = 0;
        for (; c + 8 <= n; c += 8) {                                  // loads first, adds in order
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = src[(c + j) * size];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];
        }
        for (; c < n; ++c) s += src[c * size];
        (is_w ? out_w : out_b)[g * size + idx] = s;
    }
}

// stage a [rows x cols] tile (cols a multiple of 4 floats, 16-byte aligned source rows) into LDS with row stride `stride`: every
// thread issues ALL its 16-byte loads before the first LDS store (a loop of load -> store pairs with a run-time trip count serialised
// the tile's memory round trips: 19 us per launch for three 12 KB tiles)
template <int MAXP, typename LOAD>
__device__ __forceinline__ void stage_load(float4 (&v)[MAXP], int rows, int cols_p, int tid, LOAD load) {
    const int c4n = cols_p >> 2;                                      // float4 columns (<= 32)
    const int lg = c4n <= 4 ? 2 : c4n <= 8 ? 3 : c4n <= 16 ? 4 : 5;   // threads per row, rounded up to a power of two
    const int c4 = tid & ((1 << lg) - 1), r_in = tid >> lg, rpp = DB_THREADS >> lg;
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
        const int r = r_in + p * rpp;
        v[p] = (r < rows && c4 < c4n) ? load(r, 4 * c4) : f4_zero();
    }
}
template <int MAXP>
__device__ __forceinline__ void stage_store(const float4 (&v)[MAXP], float *lds, int stride, int rows, int cols_p, int tid) {
    const int c4n = cols_p >> 2;
    const int lg = c4n <= 4 ? 2 : c4n <= 8 ? 3 : c4n <= 16 ? 4 : 5;
    const int c4 = tid & ((1 << lg) - 1), r_in = tid >> lg, rpp = DB_THREADS >> lg;
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
        const int r = r_in + p * rpp;
        if (r < rows && c4 < c4n) {
            float *d = lds + r * stride + 4 * c4;
            d[0] = v[p].x; d[1] = v[p].y; d[2] = v[p].z; d[3] = v[p].w;
        }
    }
}
template <int MAXP, typename LOAD>
__device__ __forceinline__ void stage_tile(float *lds, int stride, int rows, int cols_p, int tid, LOAD load) {
    float4 v[MAXP];
    stage_load<MAXP>(v, rows, cols_p, tid, load);
    stage_store<MAXP>(v, lds, stride, rows, cols_p, tid);
}

template <int MAXT, bool VEC>                                         // MAXT: 16 x 16 tiles of dW per wave (4 waves): 4 covers K, N <= 64
__global__ __launch_bounds__(DB_THREADS) void dense_bwd_kernel(const DenseBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float db_lds[];
    const int Kp = (a.K + 15) & ~15, Np = (a.N + 15) & ~15;
    const int sx = Kp + 2, sz = Np + 2, sw = Np + 2;                  // row strides: (stride / 2) odd -> the column walks of the MFMA operands spread over the banks
    float *xs = db_lds;                                               // [64][sx]   X tile (zero beyond K and M)
    float *zs = xs + DB_ROWS * sx;                                    // [64][sz]   dZ tile
    float *ws = zs + DB_ROWS * sz;                                    // [Kp][sw]   W (zero beyond K, N)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, l4 = lane >> 4;
    const int kt_n = Kp >> 4, nt_n = Np >> 4;
    float4 vw[8];                                                     // the kernel's first 64 rows: requested here, stored with the first sub-tile's operands
    if (a.W) {
        if (VEC) {
            auto ldw = [&](int k, int n) { return (k < a.K && n < a.N) ? *reinterpret_cast<const float4 *>(a.W + (int64_t)k * a.N + n) : f4_zero(); };
            if (Kp > 64) stage_tile<8>(ws + 64 * sw, sw, Kp - 64, Np, tid, [&](int k, int n) { return ldw(k + 64, n); });
            stage_load<8>(vw, Kp < 64 ? Kp : 64, Np, tid, ldw);
        } else {
            for (int e = tid; e < Kp * Np; e += DB_THREADS) {
                const int k = e / Np, n = e - k * Np;
                ws[k * sw + n] = (k < a.K && n < a.N) ? a.W[(int64_t)k * a.N + n] : 0.f;
            }
        }
    }
    v4f accw[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) accw[t] = v4f{0.f, 0.f, 0.f, 0.f};
    float accb = 0.f;
    for (int sub = 0; sub < a.subtiles; ++sub) {
        const int64_t r0 = ((int64_t)blockIdx.x * a.subtiles + sub) * DB_ROWS;
        if (r0 >= a.M) break;
        __syncthreads();                                             // (the previous sub-tile's operands are no longer read)
        if (VEC) {                                                  // every 16-byte load of both tiles is issued before the first LDS store
            float4 vz[8], vy[8], vx[8];
            stage_load<8>(vz, DB_ROWS, Np, tid, [&](int r, int n) {
                const int64_t m = r0 + r;
                return (m < a.M && n < a.N) ? *reinterpret_cast<const float4 *>(a.dY + m * a.lddy + n) : f4_zero();
            });
            if (a.Y)
                stage_load<8>(vy, DB_ROWS, Np, tid, [&](int r, int n) {
                    const int64_t m = r0 + r;
                    return (m < a.M && n < a.N) ? *reinterpret_cast<const float4 *>(a.Y + m * a.ldy + n) : f4_zero();
                });
            if (a.X)
                stage_load<8>(vx, DB_ROWS, Kp, tid, [&](int r, int k) {
                    const int64_t m = r0 + r;
                    return (m < a.M && k < a.K) ? *reinterpret_cast<const float4 *>(a.X + m * a.ldx + k) : f4_zero();
                });
            if (a.Y) {
#pragma unroll
                for (int p = 0; p < 8; ++p)
                    vz[p] = make_float4(act_grad(vz[p].x, vy[p].x, a.act), act_grad(vz[p].y, vy[p].y, a.act),
                                        act_grad(vz[p].z, vy[p].z, a.act), act_grad(vz[p].w, vy[p].w, a.act));
            }
            stage_store<8>(vz, zs, sz, DB_ROWS, Np, tid);
            if (a.X) stage_store<8>(vx, xs, sx, DB_ROWS, Kp, tid);
            if (a.W && sub == 0) stage_store<8>(vw, ws, sw, Kp < 64 ? Kp : 64, Np, tid);
        } else {
            for (int e = tid; e < DB_ROWS * Np; e += DB_THREADS) {
                const int r = e / Np, n = e - r * Np;
                const int64_t m = r0 + r;
                float v = 0.f;
                if (m < a.M && n < a.N) v = act_grad(a.dY[m * a.lddy + n], a.Y ? a.Y[m * a.ldy + n] : 0.f, a.Y ? a.act : AMAR_ACT_NONE);
                zs[r * sz + n] = v;
            }
            if (a.X)
                for (int e = tid; e < DB_ROWS * Kp; e += DB_THREADS) {
                    const int r = e / Kp, k = e - r * Kp;
                    const int64_t m = r0 + r;
                    xs[r * sx + k] = (m < a.M && k < a.K) ? a.X[m * a.ldx + k] : 0.f;
                }
        }
        __syncthreads();
        if (a.dZ)                                                     // the pre-activation gradient itself (a GCN layer feeds it to A_hat)
            for (int e = tid; e < DB_ROWS * a.N; e += DB_THREADS) {
                const int r = e / a.N, n = e - r * a.N;
                if (r0 + r < a.M) a.dZ[(r0 + r) * a.lddz + n] = zs[r * sz + n];
            }
        // dX tile [64 x K] = dZ [64 x N] . W^T [N x K]: A[m][n] = dZ, B[n][k] = W[k][n]
        if (a.dX)
            for (int tile = wave; tile < 4 * kt_n; tile += DB_THREADS / 64) {
                const int mt = tile / kt_n, kt = tile - mt * kt_n;
                v4f acc = {0.f, 0.f, 0.f, 0.f};
                const float *ap = zs + (16 * mt + l16) * sz + l4, *bp = ws + (16 * kt + l16) * sw + l4;
                acc = mfma_chain16(ap, 1, bp, 1, Np, acc);
                const int col = 16 * kt + l16;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int64_t m = r0 + 16 * mt + 4 * l4 + i;
                    if (m < a.M && col < a.K) { float *o = a.dX + m * a.lddx + col; *o = a.accum_dx ? *o + acc[i] : acc[i]; }
                }
            }
        // dW [K x N] += X^T [K x 64] . dZ [64 x N]: A[k][r] = X[r][k], B[r][n] = dZ[r][n]; tile t of this wave = wave + 4 t
        if (a.part_w) {
#pragma unroll
            for (int t = 0; t < MAXT; ++t) {
                const int tile = wave + (DB_THREADS / 64) * t;
                if (tile < kt_n * nt_n) {
                    const int kt = tile / nt_n, nt = tile - kt * nt_n;
                    const float *ap = xs + l4 * sx + 16 * kt + l16, *bp = zs + l4 * sz + 16 * nt + l16;
                    v4f acc = accw[t];
                    acc = mfma_chain16(ap, sx, bp, sz, DB_ROWS, acc);
                    accw[t] = acc;
                }
            }
        }
        if (a.part_b && tid < a.N) {                               // (four independent chains; a fixed order: (r0 + r1) + (r2 + r3) per 4 rows)
            float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
#pragma unroll 4
            for (int r = 0; r < DB_ROWS; r += 4) {
                b0 += zs[r * sz + tid]; b1 += zs[(r + 1) * sz + tid]; b2 += zs[(r + 2) * sz + tid]; b3 += zs[(r + 3) * sz + tid];
            }
            accb += (b0 + b1) + (b2 + b3);
        }
    }
    if (a.part_w) {
        float *mine = a.part_w + (int64_t)blockIdx.x * a.K * a.N;
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            const int tile = wave + (DB_THREADS / 64) * t;
            if (tile < kt_n * nt_n) {
                const int kt = tile / nt_n, nt = tile - kt * nt_n, n = 16 * nt + l16;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k = 16 * kt + 4 * l4 + i;
                    if (k < a.K && n < a.N) mine[(int64_t)k * a.N + n] = accw[t][i];
                }
            }
        }
    }
    if (a.part_b && tid < a.N) a.part_b[(int64_t)blockIdx.x * a.N + tid] = accb;
}

// The same reverse pass for calls over MANY rows with NARROW operands (K, N <= 32, multiples of 4): the reverse pass of a convolution layer
// runs over every node of the graph — 590 592 rows of 8 floats at ml1m(s=64) — where the tile kernel above (256 threads staging a 64 x 8
// tile through the LDS for two 16 x 16 x 4 matrix instructions) is all latency: 85 us per call for 57-76 MB of operands.  Here NP / 4
// threads share a row: each loads the row of dZ (and of X), keeps the outer-product accumulators of its own four dW columns (K float4) in
// registers over a grid-stride walk of the rows, and computes the dX quads it owns against W read from the LDS.  A workgroup's partial =
// butterfly over the lanes of a wave, then the four waves in order: a fixed order, reproducible bit for bit.
template <int KP, int NP>
__global__ __launch_bounds__(256) void dense_bwd_rows_kernel(const DenseBwdArgs a) {
    constexpr int T = NP / 4, RPB = 256 / T, KQ = KP / 4;
    __shared__ __attribute__((aligned(16))) float wt[NP][KP];        // W transposed: wt[n][k] (zero beyond K, N)
    __shared__ float red[4][KP * NP + NP];
    const int tid = threadIdx.x, t = tid % T, rl = tid / T, lane = tid & 63, wave = tid >> 6;
    if (a.W) {
        for (int e = tid; e < KP * NP; e += 256) {
            const int k = e / NP, n = e - k * NP;
            wt[n][k] = (k < a.K && n < a.N) ? a.W[(int64_t)k * a.N + n] : 0.f;
        }
        __syncthreads();
    }
    float4 accw[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) accw[k] = f4_zero();
    float4 accb = f4_zero();
    const int n0 = 4 * t;
    for (int64_t row = (int64_t)blockIdx.x * RPB + rl; row < a.M; row += (int64_t)gridDim.x * RPB) {
        float4 g[T];
#pragma unroll
        for (int j = 0; j < T; ++j) g[j] = 4 * j < a.N ? *reinterpret_cast<const float4 *>(a.dY + row * a.lddy + 4 * j) : f4_zero();
        if (a.Y) {
#pragma unroll
            for (int j = 0; j < T; ++j) {
                if (4 * j < a.N) {
                    const float4 y = *reinterpret_cast<const float4 *>(a.Y + row * a.ldy + 4 * j);
                    g[j] = make_float4(act_grad(g[j].x, y.x, a.act), act_grad(g[j].y, y.y, a.act), act_grad(g[j].z, y.z, a.act), act_grad(g[j].w, y.w, a.act));
                }
            }
        }
        float4 gq = f4_zero();                                       // this thread's own quad of dZ (a static pick: no dynamic register index)
#pragma unroll
        for (int j = 0; j < T; ++j) if (j == t) gq = g[j];
        if (a.dZ && n0 < a.N) *reinterpret_cast<float4 *>(a.dZ + row * a.lddz + n0) = gq;
        accb.x += gq.x; accb.y += gq.y; accb.z += gq.z; accb.w += gq.w;
        if (a.X) {
            float4 x[KQ];
            if (a.x_scalar) {                                        // (K = 1: the gradient of a GAT layer's attention vector is ds^T . H)
#pragma unroll
                for (int j = 0; j < KQ; ++j) {
                    const float *xr = a.X + row * a.ldx + 4 * j;
                    x[j] = make_float4(4 * j < a.K ? xr[0] : 0.f, 4 * j + 1 < a.K ? xr[1] : 0.f, 4 * j + 2 < a.K ? xr[2] : 0.f, 4 * j + 3 < a.K ? xr[3] : 0.f);
                }
            } else {
#pragma unroll
                for (int j = 0; j < KQ; ++j) x[j] = 4 * j < a.K ? *reinterpret_cast<const float4 *>(a.X + row * a.ldx + 4 * j) : f4_zero();
            }
#pragma unroll
            for (int j = 0; j < KQ; ++j) {
                accw[4 * j + 0] = f4_fma(x[j].x, gq, accw[4 * j + 0]);
                accw[4 * j + 1] = f4_fma(x[j].y, gq, accw[4 * j + 1]);
                accw[4 * j + 2] = f4_fma(x[j].z, gq, accw[4 * j + 2]);
                accw[4 * j + 3] = f4_fma(x[j].w, gq, accw[4 * j + 3]);
            }
        }
        if (a.dX) {
            for (int q = t; q < KQ; q += T) {                        // dX[row][4 q .. 4 q + 3] = sum over n of dZ[n] * W[k][n]
                float4 acc = f4_zero();
#pragma unroll
                for (int j = 0; j < T; ++j) {
                    acc = f4_fma(g[j].x, *reinterpret_cast<const float4 *>(&wt[4 * j + 0][4 * q]), acc);
                    acc = f4_fma(g[j].y, *reinterpret_cast<const float4 *>(&wt[4 * j + 1][4 * q]), acc);
                    acc = f4_fma(g[j].z, *reinterpret_cast<const float4 *>(&wt[4 * j + 2][4 * q]), acc);
                    acc = f4_fma(g[j].w, *reinterpret_cast<const float4 *>(&wt[4 * j + 3][4 * q]), acc);
                }
                if (4 * q < a.K) {
                    float4 *o = reinterpret_cast<float4 *>(a.dX + row * a.lddx + 4 * q);
                    if (a.accum_dx) { const float4 old = *o; acc.x += old.x; acc.y += old.y; acc.z += old.z; acc.w += old.w; }
                    *o = acc;
                }
            }
        }
    }
    // the workgroup's partial: lanes that share a column quad (lane = t mod T) first, then the waves in order
    if (a.part_w) {
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            float4 v = accw[k];
#pragma unroll
            for (int off = T; off < 64; off <<= 1) {
                v.x += __shfl_xor(v.x, off, 64); v.y += __shfl_xor(v.y, off, 64); v.z += __shfl_xor(v.z, off, 64); v.w += __shfl_xor(v.w, off, 64);
            }
            if (lane < T) *reinterpret_cast<float4 *>(&red[wave][k * NP + n0]) = v;
        }
    }
    if (a.part_b) {
        float4 v = accb;
#pragma unroll
        for (int off = T; off < 64; off <<= 1) {
            v.x += __shfl_xor(v.x, off, 64); v.y += __shfl_xor(v.y, off, 64); v.z += __shfl_xor(v.z, off, 64); v.w += __shfl_xor(v.w, off, 64);
        }
        if (lane < T) *reinterpret_cast<float4 *>(&red[wave][KP * NP + n0]) = v;
    }
    __syncthreads();
    if (a.part_w)
        for (int e = tid; e < a.K * a.N; e += 256) {
            const int k = e / a.N, n = e - k * a.N;
            a.part_w[(int64_t)blockIdx.x * a.K * a.N + e] = ((red[0][k * NP + n] + red[1][k * NP + n]) + red[2][k * NP + n]) + red[3][k * NP + n];
        }
    if (a.part_b && tid < a.N)
        a.part_b[(int64_t)blockIdx.x * a.N + tid] = ((red[0][KP * NP + tid] + red[1][KP * NP + tid]) + red[2][KP * NP + tid]) + red[3][KP * NP + tid];
}


// ---- a whole Dense stack forward in one launch, every layer's output kept (round 4) ---------------------------------------------------
// What fit()'s forward pass runs per tower / classifier: y_0 = X[ids], y_{l+1} = act_l(y_l . W_l + b_l), all y_l written out (the reverse
// pass reads them).  One launch per stack instead of one per layer plus the row gather and the concat copies: a 64-row tile walks the
// layers with its activations in LDS (two buffers in turn) and the layer's kernel staged next to them; products on v_mfma_f32_16x16x4_f32
// in ascending k, bias and activation after the sum (the order of operations of amar_dense_f32).
constexpr int DS_MAX_LAYERS = 4;
// dW [K x N] = X^T . dZ (and db = column sums of dZ) for batch-sized M and WIDE layers (the 768 -> 256 -> 64 content towers of a hybrid model:
// beyond the fused reverse-pass kernels' 128 columns).  A workgroup OWNS a 32 x 32 tile of dW and walks all M rows, 256 at a stage, the
// next stage's 16-byte loads in flight while the matrix instructions of the current one run: no partial sums, no second launch, one
// fixed order of additions.  (The round-2 path — scalar FMAs over row chunks + a reduction launch — took 19.5 + 4.9 us per call at
// M = 1 024; a hybrid batch makes four.)
constexpr int WM_ROWS = 256, WM_STRIDE = 34;
__global__ __launch_bounds__(256) void wgrad_mfma_kernel(const float *__restrict__ X, int64_t ldx, const float *__restrict__ dZ, int64_t ldz,
                                                         int64_t M, int K, int N, float *__restrict__ dW, float *__restrict__ db) {
    extern __shared__ __attribute__((aligned(16))) float wm_lds[];
    float *xs = wm_lds, *zs = wm_lds + WM_ROWS * WM_STRIDE;           // [256][34] each: 32 columns of X (k0 ..) and of dZ (n0 ..)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, l4 = lane >> 4;
    const int k0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int c4 = tid & 7, r_in = tid >> 3;                          // 8 threads per row of 32 floats, 32 rows per pass, 8 passes per stage
    float4 vx[8], vz[8];
    auto request = [&](int64_t row0) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int64_t m = row0 + r_in + 32 * p;
            const int k = k0 + 4 * c4, n = n0 + 4 * c4;
            vx[p] = (m < M && k < K) ? *reinterpret_cast<const float4 *>(X + m * ldx + k) : f4_zero();
            vz[p] = (m < M && n < N) ? *reinterpret_cast<const float4 *>(dZ + m * ldz + n) : f4_zero();
        }
    };
    const int kt = wave >> 1, nt = wave & 1;
    // four accumulators, one per 64 rows of a stage (added at the end: (0 + 1) + (2 + 3)): one chain of 64 dependent matrix instructions
    // per stage ran at their latency, not at the pipe's rate
    v4f acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
    request(0);
    for (int64_t row0 = 0; row0 < M; row0 += WM_ROWS) {
        __syncthreads();                                             // (the previous stage's tiles are no longer read)
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            float *dx = xs + (r_in + 32 * p) * WM_STRIDE + 4 * c4, *dz = zs + (r_in + 32 * p) * WM_STRIDE + 4 * c4;
            dx[0] = vx[p].x; dx[1] = vx[p].y; dx[2] = vx[p].z; dx[3] = vx[p].w;
            dz[0] = vz[p].x; dz[1] = vz[p].y; dz[2] = vz[p].z; dz[3] = vz[p].w;
        }
        __syncthreads();
        if (row0 + WM_ROWS < M) request(row0 + WM_ROWS);
        {
            const float *ap = xs + l4 * WM_STRIDE + 16 * kt + l16, *bp = zs + l4 * WM_STRIDE + 16 * nt + l16;
#pragma unroll 2
            for (int r = 0; r < 64; r += 4) {
                const float a0 = ap[r * WM_STRIDE], a1 = ap[(r + 64) * WM_STRIDE], a2 = ap[(r + 128) * WM_STRIDE], a3 = ap[(r + 192) * WM_STRIDE];
                const float c0 = bp[r * WM_STRIDE], c1 = bp[(r + 64) * WM_STRIDE], c2 = bp[(r + 128) * WM_STRIDE], c3 = bp[(r + 192) * WM_STRIDE];
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, c0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, c1, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, c2, acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, c3, acc3, 0, 0, 0);
            }
        }
        if (db && blockIdx.x == 0 && tid < 32) {                     // (four independent chains, rows in order inside each: a fixed order)
#pragma unroll 4
            for (int r = 0; r < WM_ROWS; r += 4) {
                b0 += zs[r * WM_STRIDE + tid]; b1 += zs[(r + 1) * WM_STRIDE + tid]; b2 += zs[(r + 2) * WM_STRIDE + tid]; b3 += zs[(r + 3) * WM_STRIDE + tid];
            }
        }
    }
    const int n = n0 + 16 * nt + l16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = k0 + 16 * kt + 4 * l4 + i;
        if (k < K && n < N) dW[(int64_t)k * N + n] = (acc0[i] + acc1[i]) + (acc2[i] + acc3[i]);
    }
    if (db && blockIdx.x == 0 && tid < 32 && n0 + tid < N) db[n0 + tid] = (b0 + b1) + (b2 + b3);
}

#ifdef AMAR_DS_STAMPS                                  // development build only (tools/exp_ds_stamps.py): cycle stamps of the stack kernels' phases
__device__ unsigned long long ds_debug_stamps[64 * 16];
#define DS_STAMP(i) do { if (threadIdx.x == 0 && block < 64) ds_debug_stamps[16 * block + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define DS_STAMP(i) do { } while (0)
#endif
struct DenseStackArgs {
    const float *X; int64_t ldx; const int32_t *ids; float *Xcopy; int64_t ldxc;
    const float *W[DS_MAX_LAYERS]; const float *bias[DS_MAX_LAYERS]; float *Y[DS_MAX_LAYERS]; int64_t ldy[DS_MAX_LAYERS];
    int dims[DS_MAX_LAYERS + 1]; int act[DS_MAX_LAYERS]; int n_layers; int64_t M; int maxd; int vec_x; int vec_w[DS_MAX_LAYERS];
};

__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == AMAR_ACT_RELU) return fmaxf(v, 0.f);
    if (act == AMAR_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
    return v;
}

// ROWS = rows per workgroup: 64, or 16 for batch-sized operands (round 4: with 64, the four waves of a workgroup each walked 3-4 output
// tiles one after the other per layer — ~2 000 clocks of epilogue and matrix-pipe latency per tile with nothing else on the CU to hide
// it, tools/exp_ds_stamps.py; 16-row tiles make 64 workgroups of a 1 024-row batch, one tile per wave and layer)
template <int ROWS>
__device__ __forceinline__ void dense_stack_body(const DenseStackArgs &a, const int block) {
    extern __shared__ __attribute__((aligned(16))) float ds_lds[];
    const int sa = a.maxd + 2;                                        // activation row stride ((stride / 2) odd: maxd is a multiple of 16)
    float *cur = ds_lds, *nxt = cur + ROWS * sa, *ws = nxt + ROWS * sa;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, l4 = lane >> 4;
    const int64_t r0 = (int64_t)block * ROWS;
    // a layer's kernel [Kp x Np] into `ws`: by 16-byte loads where its width allows (the first 64 rows may arrive in registers,
    // requested while the previous layer's products ran)
    auto w_load = [&](int l, int k_lo, int rows, float4 (&v)[8]) {
        const int K = a.dims[l], N = a.dims[l + 1], Np = (N + 15) & ~15;
        const float *w = a.W[l];
        stage_load<8>(v, rows, Np, tid, [&](int k, int n) {
            return (k_lo + k < K && n < N) ? *reinterpret_cast<const float4 *>(w + (int64_t)(k_lo + k) * N + n) : f4_zero();
        });
    };
    auto w_scalar = [&](int l) {
        const int K = a.dims[l], N = a.dims[l + 1], Kp = (K + 15) & ~15, Np = (N + 15) & ~15, sw = Np + 2;
        for (int e = tid; e < Kp * Np; e += DB_THREADS) {
            const int k = e / Np, n = e - k * Np;
            ws[k * sw + n] = (k < K && n < N) ? a.W[l][(int64_t)k * N + n] : 0.f;
        }
    };
    float4 wreg[8];
    DS_STAMP(0);
    // every layer's bias: requested now, parked in the LDS behind the input tile's round trip (read per output tile inside the layer
    // loop, a bias value was a global load whose latency each of a wave's tiles paid again: stamps, tools/exp_ds_stamps.py)
    __shared__ float bias_s[DS_MAX_LAYERS][DB_MAXD];
    float breg[DS_MAX_LAYERS];
#pragma unroll
    for (int l = 0; l < DS_MAX_LAYERS; ++l)
        breg[l] = (l < a.n_layers && tid < a.dims[l + 1] && tid < DB_MAXD && a.bias[l]) ? a.bias[l][tid] : 0.f;
    {
        const int K = a.dims[0], Kp = (K + 15) & ~15, Np1 = (a.dims[1] + 15) & ~15;
        if (a.vec_w[0]) w_load(0, 0, Kp < 64 ? Kp : 64, wreg);       // (requested together with the input tile)
        if (a.vec_x) {
            stage_tile<8>(cur, sa, ROWS, Kp, tid, [&](int r, int k) {
                const int64_t m = r0 + r;
                if (m >= a.M || k >= K) return f4_zero();
                const float4 v = *reinterpret_cast<const float4 *>(a.X + (a.ids ? (int64_t)a.ids[m] : m) * a.ldx + k);
                if (a.Xcopy) *reinterpret_cast<float4 *>(a.Xcopy + m * a.ldxc + k) = v;
                return v;
            });
        } else {
            for (int e = tid; e < ROWS * Kp; e += DB_THREADS) {
                const int r = e / Kp, k = e - r * Kp;
                const int64_t m = r0 + r;
                float v = 0.f;
                if (m < a.M && k < K) {
                    v = a.X[(a.ids ? (int64_t)a.ids[m] : m) * a.ldx + k];
                    if (a.Xcopy) a.Xcopy[m * a.ldxc + k] = v;
                }
                cur[r * sa + k] = v;
            }
        }
        if (a.vec_w[0]) {
            stage_store<8>(wreg, ws, Np1 + 2, Kp < 64 ? Kp : 64, Np1, tid);
            if (Kp > 64) { w_load(0, 64, Kp - 64, wreg); stage_store<8>(wreg, ws + 64 * (Np1 + 2), Np1 + 2, Kp - 64, Np1, tid); }
        } else w_scalar(0);
        if (tid < DB_MAXD) {
#pragma unroll
            for (int l = 0; l < DS_MAX_LAYERS; ++l) bias_s[l][tid] = breg[l];
        }
    }
    for (int l = 0; l < a.n_layers; ++l) {
        const int K = a.dims[l], N = a.dims[l + 1], Kp = (K + 15) & ~15, Np = (N + 15) & ~15, sw = Np + 2;
        __syncthreads();                                             // `cur` and `ws` of this layer are in place
        DS_STAMP(1 + 3 * l);
        const bool more = l + 1 < a.n_layers, pre = more && a.vec_w[l + 1];
        // this layer's descriptor fields in registers: read inside the store's `if`, each was a scalar load of its own per STORE (the
        // compiler does not hoist loads out of a conditional): eight dependent ~250-clock round trips per output tile
        float *const y_out = a.Y[l];
        const int64_t ld_out = a.ldy[l], m_all = a.M;
        const int act_l = a.act[l];
        if (pre) w_load(l + 1, 0, Np < 64 ? Np : 64, wreg);          // the next layer's kernel (its K = this N) travels while the products run
        const int nt_n = Np >> 4;
        for (int tile = wave; tile < (ROWS / 16) * nt_n; tile += DB_THREADS / 64) {
            const int mt = tile / nt_n, nt = tile - mt * nt_n;
            v4f acc = {0.f, 0.f, 0.f, 0.f};
            const float *ap = cur + (16 * mt + l16) * sa + l4, *bp = ws + l4 * sw + 16 * nt + l16;
            acc = mfma_chain16(ap, 1, bp, sw, Kp, acc);
            const int n = 16 * nt + l16;
            const float b = n < N ? bias_s[l][n] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 16 * mt + 4 * l4 + i;
                const int64_t m = r0 + r;
                const float y = n < N ? act_apply(acc[i] + b, act_l) : 0.f;
                nxt[r * sa + n] = y;
                if (m < m_all && n < N) y_out[m * ld_out + n] = y;
            }
        }
        DS_STAMP(2 + 3 * l);
        __syncthreads();                                             // every wave is done with `ws`
        DS_STAMP(3 + 3 * l);
        if (more) {
            const int Np2 = (a.dims[l + 2] + 15) & ~15;
            if (pre) {
                stage_store<8>(wreg, ws, Np2 + 2, Np < 64 ? Np : 64, Np2, tid);
                if (Np > 64) { w_load(l + 1, 64, Np - 64, wreg); stage_store<8>(wreg, ws + 64 * (Np2 + 2), Np2 + 2, Np - 64, Np2, tid); }
            } else w_scalar(l + 1);
        }
        float *t = cur; cur = nxt; nxt = t;
    }
    DS_STAMP(15);
}

template <int ROWS>
__global__ __launch_bounds__(DB_THREADS) void dense_stack_kernel(const DenseStackArgs a) { dense_stack_body<ROWS>(a, (int)blockIdx.x); }

// two INDEPENDENT stacks in one launch (the user and the item tower of a training batch: 16 workgroups each, 18 us each as launches of
// their own — a batch at ml1m(s=1) is a chain of such latencies): the first `split` workgroups run the first stack
struct DenseStackPair { DenseStackArgs s0, s1; int split; };
template <int ROWS>
__global__ __launch_bounds__(DB_THREADS) void dense_stack_pair_kernel(const DenseStackPair p) {
    if ((int)blockIdx.x < p.split) dense_stack_body<ROWS>(p.s0, (int)blockIdx.x);
    else dense_stack_body<ROWS>(p.s1, (int)blockIdx.x - p.split);
}

// ---- the reverse pass of a whole Dense stack in one launch (round 4) ------------------------------------------------------------------
// A 64-row tile walks the layers from the last to the first: dZ_l in LDS, dW_l / db_l partials out, dX_l = dZ_l . W_l^T in registers,
// dZ_{l-1} = dX_l * act'(y_{l-1}) — and y_{l-1} IS layer l's input, already staged for the weight gradient.  One launch per tower /
// classifier instead of one amar_dense_bwd_f32 per layer; the partials are added by amar_adam_multi_f32 (or by one reduction per layer).
struct DenseStackBwdArgs {
    const float *dYtop; int64_t lddy; const float *Ytop; int64_t ldytop;
    const float *X[DS_MAX_LAYERS]; int64_t ldx[DS_MAX_LAYERS]; const float *W[DS_MAX_LAYERS];
    float *part_w[DS_MAX_LAYERS]; float *part_b[DS_MAX_LAYERS]; float *dX0; int64_t lddx0;
    int dims[DS_MAX_LAYERS + 1]; int act[DS_MAX_LAYERS]; int vec_x[DS_MAX_LAYERS]; int vec_w[DS_MAX_LAYERS]; int vec_top;
    int n_layers; int64_t M; int maxd;
};

template <int ROWS>                                 // rows per workgroup = rows per weight-gradient partial: 64, or 16 for batches of at most 1 024 rows
__device__ __forceinline__ void dense_stack_bwd_body(const DenseStackBwdArgs &a, const int block) {
    extern __shared__ __attribute__((aligned(16))) float sb_lds[];
    const int smax = a.maxd + 2;
    float *xs = sb_lds, *zs = xs + ROWS * smax, *ws = zs + ROWS * smax;      // X_l tile, dZ_l tile, W_l (each with its layer's stride)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, l4 = lane >> 4;
    const int64_t r0 = (int64_t)block * ROWS;
    const int L = a.n_layers;
    {   // dZ of the last layer: dYtop * act'(Ytop) (Ytop == NULL: dYtop already is dZ)
        const int N = a.dims[L], Np = (N + 15) & ~15, sz = Np + 2, actl = a.act[L - 1];
        if (a.vec_top) {
            float4 vz[8], vy[8];
            stage_load<8>(vz, ROWS, Np, tid, [&](int r, int n) {
                const int64_t m = r0 + r;
                return (m < a.M && n < N) ? *reinterpret_cast<const float4 *>(a.dYtop + m * a.lddy + n) : f4_zero();
            });
            if (a.Ytop) {
                stage_load<8>(vy, ROWS, Np, tid, [&](int r, int n) {
                    const int64_t m = r0 + r;
                    return (m < a.M && n < N) ? *reinterpret_cast<const float4 *>(a.Ytop + m * a.ldytop + n) : f4_zero();
                });
#pragma unroll
                for (int p = 0; p < 8; ++p)
                    vz[p] = make_float4(act_grad(vz[p].x, vy[p].x, actl), act_grad(vz[p].y, vy[p].y, actl),
                                        act_grad(vz[p].z, vy[p].z, actl), act_grad(vz[p].w, vy[p].w, actl));
            }
            stage_store<8>(vz, zs, sz, ROWS, Np, tid);
        } else {
            for (int e = tid; e < ROWS * Np; e += DB_THREADS) {
                const int r = e / Np, n = e - r * Np;
                const int64_t m = r0 + r;
                float v = 0.f;
                if (m < a.M && n < N) v = a.Ytop ? act_grad(a.dYtop[m * a.lddy + n], a.Ytop[m * a.ldytop + n], actl) : a.dYtop[m * a.lddy + n];
                zs[r * sz + n] = v;
            }
        }
    }
    for (int l = L - 1; l >= 0; --l) {
        const int K = a.dims[l], N = a.dims[l + 1], Kp = (K + 15) & ~15, Np = (N + 15) & ~15, sx = Kp + 2, sz = Np + 2, sw = Np + 2;
        const int kt_n = Kp >> 4, nt_n = Np >> 4;
        // stage X_l and W_l (the previous iteration's last barrier guarantees nobody reads xs / ws any more)
        if (a.vec_x[l]) {
            const float *x = a.X[l];
            const int64_t ld = a.ldx[l];
            stage_tile<8>(xs, sx, ROWS, Kp, tid, [&](int r, int k) {
                const int64_t m = r0 + r;
                return (m < a.M && k < K) ? *reinterpret_cast<const float4 *>(x + m * ld + k) : f4_zero();
            });
        } else {
            for (int e = tid; e < ROWS * Kp; e += DB_THREADS) {
                const int r = e / Kp, k = e - r * Kp;
                const int64_t m = r0 + r;
                xs[r * sx + k] = (m < a.M && k < K) ? a.X[l][m * a.ldx[l] + k] : 0.f;
            }
        }
        const bool need_dx = l > 0 || a.dX0 != nullptr;
        if (need_dx) {
            if (a.vec_w[l]) {
                const float *w = a.W[l];
                auto ldw = [&](int k, int n) { return (k < K && n < N) ? *reinterpret_cast<const float4 *>(w + (int64_t)k * N + n) : f4_zero(); };
                stage_tile<8>(ws, sw, Kp < 64 ? Kp : 64, Np, tid, ldw);
                if (Kp > 64) stage_tile<8>(ws + 64 * sw, sw, Kp - 64, Np, tid, [&](int k, int n) { return ldw(k + 64, n); });
            } else {
                for (int e = tid; e < Kp * Np; e += DB_THREADS) {
                    const int k = e / Np, n = e - k * Np;
                    ws[k * sw + n] = (k < K && n < N) ? a.W[l][(int64_t)k * N + n] : 0.f;
                }
            }
        }
        __syncthreads();
        // dW_l partial [K x N] = X_l^T . dZ_l, db_l partial
        if (a.part_w[l]) {
            float *mine = a.part_w[l] + (int64_t)block * K * N;
            for (int tile = wave; tile < kt_n * nt_n; tile += DB_THREADS / 64) {
                const int kt = tile / nt_n, nt = tile - kt * nt_n;
                const float *ap = xs + l4 * sx + 16 * kt + l16, *bp = zs + l4 * sz + 16 * nt + l16;
                v4f acc = {0.f, 0.f, 0.f, 0.f};
                acc = mfma_chain16(ap, sx, bp, sz, ROWS, acc);
                const int n = 16 * nt + l16;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k = 16 * kt + 4 * l4 + i;
                    if (k < K && n < N) mine[(int64_t)k * N + n] = acc[i];
                }
            }
        }
        if (a.part_b[l] && tid < N) {
            float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
#pragma unroll 4
            for (int r = 0; r < ROWS; r += 4) {
                b0 += zs[r * sz + tid]; b1 += zs[(r + 1) * sz + tid]; b2 += zs[(r + 2) * sz + tid]; b3 += zs[(r + 3) * sz + tid];
            }
            a.part_b[l][(int64_t)block * N + tid] = (b0 + b1) + (b2 + b3);
        }
        // dX_l [64 x K] = dZ_l . W_l^T in registers: this wave's tiles are wave, wave + 4, ... (at most 8: K <= 128)
        v4f dxa[8];
        if (need_dx) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int tile = wave + (DB_THREADS / 64) * t;
                dxa[t] = v4f{0.f, 0.f, 0.f, 0.f};
                if (tile < (ROWS / 16) * kt_n) {
                    const int mt = tile / kt_n, kt = tile - mt * kt_n;
                    const float *ap = zs + (16 * mt + l16) * sz + l4, *bp = ws + (16 * kt + l16) * sw + l4;
                    v4f acc = {0.f, 0.f, 0.f, 0.f};
                    acc = mfma_chain16(ap, 1, bp, 1, Np, acc);
                    dxa[t] = acc;
                }
            }
        }
        __syncthreads();                                             // every read of zs (and ws) of this layer is done
        if (l > 0) {                                                 // dZ_{l-1} = dX_l * act'(y_{l-1}), y_{l-1} = X_l (in xs), into zs with stride Kp + 2
            const int actp = a.act[l - 1];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int tile = wave + (DB_THREADS / 64) * t;
                if (tile < (ROWS / 16) * kt_n) {
                    const int mt = tile / kt_n, kt = tile - mt * kt_n, col = 16 * kt + l16;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = 16 * mt + 4 * l4 + i;
                        zs[r * sx + col] = col < K ? act_grad(dxa[t][i], xs[r * sx + col], actp) : 0.f;
                    }
                }
            }
            __syncthreads();                                         // zs of layer l - 1 complete; xs / ws free for the next staging
        } else if (a.dX0) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int tile = wave + (DB_THREADS / 64) * t;
                if (tile < (ROWS / 16) * kt_n) {
                    const int mt = tile / kt_n, kt = tile - mt * kt_n, col = 16 * kt + l16;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int64_t m = r0 + 16 * mt + 4 * l4 + i;
                        if (m < a.M && col < K) a.dX0[m * a.lddx0 + col] = dxa[t][i];
                    }
                }
            }
        }
    }
}

template <int ROWS>
__global__ __launch_bounds__(DB_THREADS) void dense_stack_bwd_kernel(const DenseStackBwdArgs a) { dense_stack_bwd_body<ROWS>(a, (int)blockIdx.x); }

struct DenseStackBwdPair { DenseStackBwdArgs s0, s1; int split; };
template <int ROWS>
__global__ __launch_bounds__(DB_THREADS) void dense_stack_bwd_pair_kernel(const DenseStackBwdPair p) {
    if ((int)blockIdx.x < p.split) dense_stack_bwd_body<ROWS>(p.s0, (int)blockIdx.x);
    else dense_stack_bwd_body<ROWS>(p.s1, (int)blockIdx.x - p.split);
}

}  // namespace

extern "C" {

static int build_dense_stack(const float *X, int64_t ldx, const int32_t *ids, float *Xcopy, int64_t ldxc, int32_t n_layers,
                             const float *const *W, const float *const *bias, const int32_t *dims, const int32_t *acts,
                             float *const *Y, const int64_t *ldy, int64_t M, DenseStackArgs &a, size_t &lds, int64_t &groups) {
    if (M < 0 || !X || n_layers < 1 || !W || !bias || !dims || !acts || !Y || !ldy) return AMAR_EINVAL;
    if (n_layers > DS_MAX_LAYERS) return AMAR_EUNSUPPORTED;
    a = DenseStackArgs{};
    a.X = X; a.ldx = ldx; a.ids = ids; a.Xcopy = Xcopy; a.ldxc = ldxc; a.n_layers = n_layers; a.M = M;
    int maxd = 16;
    for (int l = 0; l <= n_layers; ++l) {
        if (dims[l] < 1) return AMAR_EINVAL;
        if (dims[l] > DB_MAXD) return AMAR_EUNSUPPORTED;
        a.dims[l] = dims[l];
        const int dp = (dims[l] + 15) & ~15;
        if (dp > maxd) maxd = dp;
    }
    if (ldx < dims[0] || (Xcopy && ldxc < dims[0])) return AMAR_EINVAL;
    int maxw = 0;
    for (int l = 0; l < n_layers; ++l) {
        if (!W[l] || !Y[l] || ldy[l] < dims[l + 1]) return AMAR_EINVAL;
        if (acts[l] != AMAR_ACT_NONE && acts[l] != AMAR_ACT_RELU && acts[l] != AMAR_ACT_SIGMOID) return AMAR_EINVAL;
        a.W[l] = W[l]; a.bias[l] = bias[l]; a.Y[l] = Y[l]; a.ldy[l] = ldy[l]; a.act[l] = acts[l];
        a.vec_w[l] = ((dims[l + 1] & 3) == 0 && amar_aligned16(W[l])) ? 1 : 0;
        const int w = ((dims[l] + 15) & ~15) * (((dims[l + 1] + 15) & ~15) + 2);
        if (w > maxw) maxw = w;
    }
    a.maxd = maxd;
    a.vec_x = ((dims[0] & 3) == 0 && (ldx & 3) == 0 && amar_aligned16(X) && (!Xcopy || ((ldxc & 3) == 0 && amar_aligned16(Xcopy)))) ? 1 : 0;
    groups = (M + DB_ROWS - 1) / DB_ROWS;                             // (for 64-row workgroups; the launchers below rescale for 16)
    if (groups > 0x3fffffff / 4) return AMAR_EUNSUPPORTED;
    lds = ((size_t)2 * DB_ROWS * (maxd + 2) + (size_t)maxw) * sizeof(float);
    return AMAR_OK;
}

// batch-sized operands run in 16-row workgroups (see dense_stack_body); AMAR_DENSE_STACK_ROWS=64 keeps 64 (A/B timing)
static bool dense_stack_small_rows(int64_t M) {
    static const bool force64 = getenv("AMAR_DENSE_STACK_ROWS") && atoi(getenv("AMAR_DENSE_STACK_ROWS")) == 64;
    return M <= 4096 && !force64;
}

int amar_dense_stack_f32(const float *X, int64_t ldx, const int32_t *ids, float *Xcopy, int64_t ldxc, int32_t n_layers,
                         const float *const *W, const float *const *bias, const int32_t *dims, const int32_t *acts,
                         float *const *Y, const int64_t *ldy, int64_t M, amar_stream_t stream) {
    DenseStackArgs a;
    size_t lds = 0;
    int64_t groups = 0;
    if (const int rc = build_dense_stack(X, ldx, ids, Xcopy, ldxc, n_layers, W, bias, dims, acts, Y, ldy, M, a, lds, groups)) return rc;
    if (M == 0) return AMAR_OK;
    if (dense_stack_small_rows(M)) {
        static bool allowed16[AMAR_MAX_DEVICES] = {};
        if (const int rc = amar_allow_lds(reinterpret_cast<const void *>(dense_stack_kernel<16>), lds, allowed16)) return rc;
        hipLaunchKernelGGL(dense_stack_kernel<16>, dim3((unsigned)((M + 15) / 16)), dim3(DB_THREADS), lds, static_cast<hipStream_t>(stream), a);
        return amar_check_launch();
    }
    static bool allowed[AMAR_MAX_DEVICES] = {};
    if (const int rc = amar_allow_lds(reinterpret_cast<const void *>(dense_stack_kernel<64>), lds, allowed)) return rc;
    hipLaunchKernelGGL(dense_stack_kernel<64>, dim3((unsigned)groups), dim3(DB_THREADS), lds, static_cast<hipStream_t>(stream), a);
    return amar_check_launch();
}

int amar_dense_stack_pair_f32(const amar_dense_stack_desc *s0, const amar_dense_stack_desc *s1, amar_stream_t stream) {
    if (!s0 || !s1) return AMAR_EINVAL;
    DenseStackPair p;
    size_t lds0 = 0, lds1 = 0;
    int64_t g0 = 0, g1 = 0;
    if (const int rc = build_dense_stack(s0->X, s0->ldx, s0->ids, s0->Xcopy, s0->ldxc, s0->n_layers, s0->W, s0->bias, s0->dims, s0->acts, s0->Y, s0->ldy,
                                         s0->M, p.s0, lds0, g0)) return rc;
    if (const int rc = build_dense_stack(s1->X, s1->ldx, s1->ids, s1->Xcopy, s1->ldxc, s1->n_layers, s1->W, s1->bias, s1->dims, s1->acts, s1->Y, s1->ldy,
                                         s1->M, p.s1, lds1, g1)) return rc;
    if (g0 + g1 == 0) return AMAR_OK;
    const size_t lds = lds0 > lds1 ? lds0 : lds1;
    if (dense_stack_small_rows(s0->M) && dense_stack_small_rows(s1->M)) {
        g0 = (s0->M + 15) / 16; g1 = (s1->M + 15) / 16;
        p.split = (int)g0;
        static bool allowed16[AMAR_MAX_DEVICES] = {};
        if (const int rc = amar_allow_lds(reinterpret_cast<const void *>(dense_stack_pair_kernel<16>), lds, allowed16)) return rc;
        hipLaunchKernelGGL(dense_stack_pair_kernel<16>, dim3((unsigned)(g0 + g1)), dim3(DB_THREADS), lds, static_cast<hipStream_t>(stream), p);
        return amar_check_launch();
    }
    p.split = (int)g0;
    static bool allowed[AMAR_MAX_DEVICES] = {};
    if (const int rc = amar_allow_lds(reinterpret_cast<const void *>(dense_stack_pair_kernel<64>), lds, allowed)) return rc;
    hipLaunchKernelGGL(dense_stack_pair_kernel<64>, dim3((unsigned)(g0 + g1)), dim3(DB_THREADS), lds, static_cast<hipStream_t>(stream), p);
    return amar_check_launch();
}

// rows per workgroup (= per partial) of the stack's reverse pass: 16 up to 1 024 rows (64 workgroups of a batch instead of 16: the same
// latency argument as the forward's), else 64; AMAR_DENSE_STACK_ROWS=64 keeps 64
static int dense_stack_bwd_rows(int64_t M) {
    static const bool force64 = getenv("AMAR_DENSE_STACK_ROWS") && atoi(getenv("AMAR_DENSE_STACK_ROWS")) == 64;
    return (M <= 1024 && !force64) ? 16 : DB_ROWS;
}

int64_t amar_dense_stack_bwd_groups(int64_t M) {
    if (M < 0) return AMAR_EINVAL;
    const int rows = dense_stack_bwd_rows(M);
    return (M + rows - 1) / rows;
}

int64_t amar_dense_stack_bwd_workspace_floats(int64_t M, int32_t n_layers, const int32_t *dims) {
    if (M < 0 || n_layers < 1 || n_layers > DS_MAX_LAYERS || !dims) return AMAR_EINVAL;
    const int64_t groups = amar_dense_stack_bwd_groups(M);
    int64_t total = 4;
    for (int l = 0; l < n_layers; ++l) total += groups * ((int64_t)dims[l] * dims[l + 1] + dims[l + 1]);
    return total;
}

static int build_dense_stack_bwd(const float *dYtop, int64_t lddy, const float *Ytop, int64_t ldytop, int32_t n_layers,
                                 const float *const *X, const int64_t *ldx, const float *const *W, const int32_t *dims, const int32_t *acts,
                                 float *dX0, int64_t lddx0, float *const *dW, float *const *db, float *workspace, int64_t M,
                                 DenseStackBwdArgs &a, size_t &lds, int64_t &groups) {
    if (M < 1 || !dYtop || n_layers < 1 || !X || !ldx || !W || !dims || !acts || !dW || !db || !workspace) return AMAR_EINVAL;
    if (n_layers > DS_MAX_LAYERS) return AMAR_EUNSUPPORTED;
    if (M > 4096) return AMAR_EUNSUPPORTED;                          // (batch-sized operands: one row tile per workgroup, no sub-tiles)
    groups = amar_dense_stack_bwd_groups(M);
    a = DenseStackBwdArgs{};
    a.dYtop = dYtop; a.lddy = lddy; a.Ytop = Ytop; a.ldytop = ldytop; a.dX0 = dX0; a.lddx0 = lddx0; a.n_layers = n_layers; a.M = M;
    int maxd = 16;
    for (int l = 0; l <= n_layers; ++l) {
        if (dims[l] < 1) return AMAR_EINVAL;
        if (dims[l] > DB_MAXD) return AMAR_EUNSUPPORTED;
        a.dims[l] = dims[l];
        if (((dims[l] + 15) & ~15) > maxd) maxd = (dims[l] + 15) & ~15;
    }
    if (lddy < dims[n_layers] || (Ytop && ldytop < dims[n_layers]) || (dX0 && lddx0 < dims[0])) return AMAR_EINVAL;
    a.vec_top = ((dims[n_layers] & 3) == 0 && (lddy & 3) == 0 && amar_aligned16(dYtop) && (!Ytop || ((ldytop & 3) == 0 && amar_aligned16(Ytop)))) ? 1 : 0;
    float *p = workspace + 4;
    int maxw = 0;
    for (int l = 0; l < n_layers; ++l) {
        if (!X[l] || !W[l] || !dW[l] || !db[l] || ldx[l] < dims[l]) return AMAR_EINVAL;
        if (acts[l] != AMAR_ACT_NONE && acts[l] != AMAR_ACT_RELU && acts[l] != AMAR_ACT_SIGMOID) return AMAR_EINVAL;
        a.X[l] = X[l]; a.ldx[l] = ldx[l]; a.W[l] = W[l]; a.act[l] = acts[l];
        a.vec_x[l] = ((dims[l] & 3) == 0 && (ldx[l] & 3) == 0 && amar_aligned16(X[l])) ? 1 : 0;
        a.vec_w[l] = ((dims[l + 1] & 3) == 0 && amar_aligned16(W[l])) ? 1 : 0;
        a.part_w[l] = p; p += groups * (int64_t)dims[l] * dims[l + 1];
        a.part_b[l] = p; p += groups * (int64_t)dims[l + 1];
        const int w = ((dims[l] + 15) & ~15) * (((dims[l + 1] + 15) & ~15) + 2);
        if (w > maxw) maxw = w;
    }
    a.maxd = maxd;
    lds = ((size_t)2 * DB_ROWS * (maxd + 2) + (size_t)maxw) * sizeof(float);
    return AMAR_OK;
}

static void reduce_stack_partials(const DenseStackBwdArgs &a, const int32_t *dims, float *const *dW, float *const *db, int64_t groups, hipStream_t st) {
    for (int l = 0; l < a.n_layers; ++l)
        hipLaunchKernelGGL(reduce_partials2_kernel, dim3(grid1d((int64_t)dims[l] * dims[l + 1] + dims[l + 1])), dim3(256), 0, st, a.part_w[l],
                           (int64_t)dims[l] * dims[l + 1], dW[l], a.part_b[l], (int64_t)dims[l + 1], db[l], (int)groups);
}

int amar_dense_stack_bwd_f32(const float *dYtop, int64_t lddy, const float *Ytop, int64_t ldytop, int32_t n_layers,
                             const float *const *X, const int64_t *ldx, const float *const *W, const int32_t *dims, const int32_t *acts,
                             float *dX0, int64_t lddx0, float *const *dW, float *const *db, float *workspace, int32_t flags,
                             int64_t M, amar_stream_t stream) {
    DenseStackBwdArgs a;
    size_t lds = 0;
    int64_t groups = 0;
    if (const int rc = build_dense_stack_bwd(dYtop, lddy, Ytop, ldytop, n_layers, X, ldx, W, dims, acts, dX0, lddx0, dW, db, workspace, M, a, lds, groups)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dense_stack_bwd_rows(M) == 16) {
        static bool allowed16[AMAR_MAX_DEVICES] = {};
        if (const int rc = amar_allow_lds(reinterpret_cast<const void *>(dense_stack_bwd_kernel<16>), lds, allowed16)) return rc;
        hipLaunchKernelGGL(dense_stack_bwd_kernel<16>, dim3((unsigned)groups), dim3(DB_THREADS), lds, st, a);
    } else {
        static bool allowed[AMAR_MAX_DEVICES] = {};
        if (const int rc = amar_allow_lds(reinterpret_cast<const void *>(dense_stack_bwd_kernel<64>), lds, allowed)) return rc;
        hipLaunchKernelGGL(dense_stack_bwd_kernel<64>, dim3((unsigned)groups), dim3(DB_THREADS), lds, st, a);
    }
    if (!(flags & AMAR_DENSE_BWD_DEFER)) reduce_stack_partials(a, dims, dW, db, groups, st);
    return amar_check_launch();
}

int amar_dense_stack_bwd_pair_f32(const amar_dense_stack_bwd_desc *s0, const amar_dense_stack_bwd_desc *s1, amar_stream_t stream) {
    if (!s0 || !s1) return AMAR_EINVAL;
    DenseStackBwdPair p;
    size_t lds0 = 0, lds1 = 0;
    int64_t g0 = 0, g1 = 0;
    if (const int rc = build_dense_stack_bwd(s0->dYtop, s0->lddy, s0->Ytop, s0->ldytop, s0->n_layers, s0->X, s0->ldx, s0->W, s0->dims, s0->acts, s0->dX0,
                                             s0->lddx0, s0->dW, s0->db, s0->workspace, s0->M, p.s0, lds0, g0)) return rc;
    if (const int rc = build_dense_stack_bwd(s1->dYtop, s1->lddy, s1->Ytop, s1->ldytop, s1->n_layers, s1->X, s1->ldx, s1->W, s1->dims, s1->acts, s1->dX0,
                                             s1->lddx0, s1->dW, s1->db, s1->workspace, s1->M, p.s1, lds1, g1)) return rc;
    p.split = (int)g0;
    const size_t lds = lds0 > lds1 ? lds0 : lds1;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dense_stack_bwd_rows(s0->M) != dense_stack_bwd_rows(s1->M)) return AMAR_EUNSUPPORTED;     // (two stacks of one batch: the same row tile)
    if (dense_stack_bwd_rows(s0->M) == 16) {
        static bool allowed16[AMAR_MAX_DEVICES] = {};
        if (const int rc = amar_allow_lds(reinterpret_cast<const void *>(dense_stack_bwd_pair_kernel<16>), lds, allowed16)) return rc;
        hipLaunchKernelGGL(dense_stack_bwd_pair_kernel<16>, dim3((unsigned)(g0 + g1)), dim3(DB_THREADS), lds, st, p);
    } else {
        static bool allowed[AMAR_MAX_DEVICES] = {};
        if (const int rc = amar_allow_lds(reinterpret_cast<const void *>(dense_stack_bwd_pair_kernel<64>), lds, allowed)) return rc;
        hipLaunchKernelGGL(dense_stack_bwd_pair_kernel<64>, dim3((unsigned)(g0 + g1)), dim3(DB_THREADS), lds, st, p);
    }
    if (!(s0->flags & AMAR_DENSE_BWD_DEFER)) reduce_stack_partials(p.s0, s0->dims, s0->dW, s0->db, g0, st);
    if (!(s1->flags & AMAR_DENSE_BWD_DEFER)) reduce_stack_partials(p.s1, s1->dims, s1->dW, s1->db, g1, st);
    return amar_check_launch();
}

int64_t amar_dense_bwd_groups(int64_t M) {
    if (M < 0) return AMAR_EINVAL;
    return dense_bwd_plan(M).out_groups;
}

int64_t amar_dense_bwd_workspace_floats(int64_t M, int32_t K, int32_t N) {
    if (M < 0 || K < 0 || N < 1) return AMAR_EINVAL;
    const DenseBwdPlan p = dense_bwd_plan(M);
    return 4 + (p.out_groups + (p.fold > 1 ? p.out_groups * p.fold : 0)) * ((int64_t)K * N + N);   // the partials a caller sees first, the raw ones behind them
}

int amar_dense_bwd_f32(const float *X, int64_t ldx, const float *Y, int64_t ldy, const float *dY, int64_t lddy, const float *W,
                       int32_t act, float *dX, int64_t lddx, float *dW, float *db, float *dZ, int64_t lddz, float *workspace,
                       int64_t M, int32_t K, int32_t N, amar_stream_t stream) {
    const bool defer = (act & AMAR_DENSE_BWD_DEFER) != 0, accum = (act & AMAR_DENSE_BWD_ACCUM_DX) != 0;
    act &= ~(AMAR_DENSE_BWD_DEFER | AMAR_DENSE_BWD_ACCUM_DX);
    if (M < 0 || K < 1 || N < 1 || !dY || lddy < N || (!dX && !dW && !db && !dZ)) return AMAR_EINVAL;
    if (dZ && lddz < N) return AMAR_EINVAL;
    if (Y && ldy < N) return AMAR_EINVAL;
    if (act != AMAR_ACT_NONE && act != AMAR_ACT_RELU && act != AMAR_ACT_SIGMOID) return AMAR_EINVAL;
    if (act != AMAR_ACT_NONE && !Y) return AMAR_EINVAL;
    if (dX && (!W || lddx < K)) return AMAR_EINVAL;
    if (dW && (!X || ldx < K)) return AMAR_EINVAL;
    if ((dW || db) && !workspace) return AMAR_EINVAL;
    if (K > DB_MAXD || N > DB_MAXD) return AMAR_EUNSUPPORTED;
    if (M == 0) return AMAR_EUNSUPPORTED;                             // (an empty batch: the separate kernels define the zero gradients)
    const DenseBwdPlan plan = dense_bwd_plan(M);
    const int sub = plan.sub;
    const int64_t groups = plan.out_groups;
    const int64_t size_w = dW ? (int64_t)K * N : 0, size_b = db ? (int64_t)N : 0;
    const int Kp = (K + 15) & ~15, Np = (N + 15) & ~15;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // 16-byte loads where every operand allows them (K, N and the leading dimensions multiples of 4 floats, 16-byte aligned bases)
    const bool use_x = dW != nullptr, use_w = dX != nullptr, use_y = act != AMAR_ACT_NONE;
    const bool vec_rest = (N & 3) == 0 && (lddy & 3) == 0 && amar_aligned16(dY) && (!use_y || ((ldy & 3) == 0 && amar_aligned16(Y))) &&
                          (!use_w || ((K & 3) == 0 && amar_aligned16(W)));
    const bool vec_x = !use_x || ((K & 3) == 0 && (ldx & 3) == 0 && amar_aligned16(X));
    const bool vec = vec_rest && vec_x;
    // many rows of narrow operands (a convolution layer's reverse pass over every node): the row-walking kernel, `fold` of its workgroups
    // per partial the caller sees (AMAR_DENSE_BWD_ROWS_OFF: the tile kernel, for A/B timing)
    static const bool rows_off = getenv("AMAR_DENSE_BWD_ROWS_OFF") != nullptr;
    const bool rows_form = plan.fold > 1 && vec_rest && K <= 32 && N <= 32 && (!dX || ((lddx & 3) == 0 && amar_aligned16(dX))) &&
                           (!dZ || ((lddz & 3) == 0 && amar_aligned16(dZ))) && !rows_off;      // (X may be read by single floats there)
    // the row-walking kernel up to 32 768 rows: one workgroup per partial the caller sees, no fold launch (9 228 rows at ml1m(s=1): 49
    // workgroups of three passes); beyond: up to 64 workgroups per partial
    const bool need_fold = plan.fold > 1 && !(rows_form && M <= 32768);
    const int fold = !need_fold ? 1 : rows_form ? (plan.fold < 64 ? plan.fold : 64) : plan.fold;
    const int64_t n_raw = rows_form ? groups * fold : plan.launch_groups;      // workgroups of the main launch = raw partials
    float *part_w = dW ? workspace + 4 : nullptr;                    // [groups][K N], then [groups][N]: what the caller (or the Adam launch) adds
    float *part_b = db ? workspace + 4 + groups * size_w : nullptr;
    float *raw_w = part_w, *raw_b = part_b;                          // where the workgroups write: the same, unless a fold launch follows
    if (need_fold) {
        float *raw = workspace + 4 + groups * ((int64_t)K * N + N);
        raw_w = dW ? raw : nullptr;
        raw_b = db ? raw + n_raw * size_w : nullptr;
    }
    DenseBwdArgs a{use_x ? X : nullptr, ldx, use_y ? Y : nullptr, ldy, dY, lddy, use_w ? W : nullptr, dX, lddx, raw_w, raw_b, M, K, N, act, sub,
                   dZ, lddz, accum ? 1 : 0, vec_x ? 0 : 1};
    if (rows_form) {
#define AMAR_DBR_LAUNCH(KK, NN) hipLaunchKernelGGL((dense_bwd_rows_kernel<KK, NN>), dim3((unsigned)n_raw), dim3(256), 0, st, a)
        const int kc = K <= 8 ? 0 : (K <= 16 ? 1 : 2), nc = N <= 8 ? 0 : (N <= 16 ? 1 : 2);
        switch (3 * kc + nc) {
        case 0: AMAR_DBR_LAUNCH(8, 8); break;
        case 1: AMAR_DBR_LAUNCH(8, 16); break;
        case 2: AMAR_DBR_LAUNCH(8, 32); break;
        case 3: AMAR_DBR_LAUNCH(16, 8); break;
        case 4: AMAR_DBR_LAUNCH(16, 16); break;
        case 5: AMAR_DBR_LAUNCH(16, 32); break;
        case 6: AMAR_DBR_LAUNCH(32, 8); break;
        case 7: AMAR_DBR_LAUNCH(32, 16); break;
        default: AMAR_DBR_LAUNCH(32, 32); break;
        }
#undef AMAR_DBR_LAUNCH
    } else {
        const size_t lds = ((size_t)DB_ROWS * (Kp + 2) + (size_t)DB_ROWS * (Np + 2) + (size_t)Kp * (Np + 2)) * sizeof(float);
        const bool small = (Kp >> 4) * (Np >> 4) <= 16;
#define AMAR_DB_LAUNCH(MT, VV)                                                                                           \
    do {                                                                                                                 \
        static bool allowed[AMAR_MAX_DEVICES] = {};                                                                      \
        if (const int rc = amar_allow_lds(reinterpret_cast<const void *>(dense_bwd_kernel<MT, VV>), lds, allowed)) return rc;   \
        hipLaunchKernelGGL((dense_bwd_kernel<MT, VV>), dim3((unsigned)n_raw), dim3(DB_THREADS), lds, st, a);             \
    } while (0)
        if (small) { if (vec) AMAR_DB_LAUNCH(4, true); else AMAR_DB_LAUNCH(4, false); }
        else { if (vec) AMAR_DB_LAUNCH(16, true); else AMAR_DB_LAUNCH(16, false); }
#undef AMAR_DB_LAUNCH
    }
    if (need_fold && (dW || db))
        hipLaunchKernelGGL(fold_partials2_kernel, dim3(grid1d(groups * (size_w + size_b))), dim3(256), 0, st, raw_w, size_w, part_w, raw_b, size_b, part_b,
                           n_raw, fold, groups);
    if (defer) return amar_check_launch();                          // the partials stay in the workspace (amar_adam_multi_f32 adds them)
    if (dW && db) hipLaunchKernelGGL(reduce_partials2_kernel, dim3(grid1d((int64_t)K * N + N)), dim3(256), 0, st, part_w, (int64_t)K * N, dW,
                                     part_b, (int64_t)N, db, (int)groups);
    else if (dW) hipLaunchKernelGGL(reduce_partials_kernel, dim3(grid1d((int64_t)K * N)), dim3(256), 0, st, part_w, (int)groups, (int64_t)K * N, dW);
    else if (db) hipLaunchKernelGGL(reduce_partials_kernel, dim3(grid1d((int64_t)N)), dim3(256), 0, st, part_b, (int)groups, (int64_t)N, db);
    return amar_check_launch();
}

int amar_act_bwd_f32(const float *dY, int64_t ldd, const float *Y, int64_t ldy, float *dZ, int64_t ldz,
                     int64_t M, int32_t N, int32_t act, amar_stream_t stream) {
    if (M < 0 || N < 1 || !dY || !Y || !dZ || ldd < N || ldy < N || ldz < N) return AMAR_EINVAL;
    if (act != AMAR_ACT_NONE && act != AMAR_ACT_RELU && act != AMAR_ACT_SIGMOID) return AMAR_EINVAL;
    if (M == 0) return AMAR_OK;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid1d(M * N)), dim3(256), 0, static_cast<hipStream_t>(stream), dY, ldd, Y, ldy, dZ, ldz, M, N, act);
    return amar_check_launch();
}

int64_t amar_wgrad_scratch_floats(int64_t M, int32_t K, int32_t N) {
    if (M < 0 || K < 0 || N < 1) return AMAR_EINVAL;
    const int64_t chunks = (M + wg_rows(M) - 1) / wg_rows(M);
    return chunks * ((int64_t)K * N + N);
}

int amar_wgrad_f32(const float *X, int64_t ldx, const float *dZ, int64_t ldz, int64_t M, int32_t K, int32_t N,
                   float *dW, float *db, float *scratch, amar_stream_t stream) {
    if (M < 1 || N < 1 || !dZ || ldz < N || !scratch || (!dW && !db)) return AMAR_EINVAL;
    if (dW && (!X || K < 1 || ldx < K)) return AMAR_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // wide layers over batch-sized operands: every 32 x 32 tile of dW by one workgroup on the matrix instruction, nothing to reduce
    static const bool no_mfma = getenv("AMAR_WGRAD_MFMA") && atoi(getenv("AMAR_WGRAD_MFMA")) == 0;      // development switch (A/B timing)
    if (dW && !no_mfma && M <= 16384 && (K & 3) == 0 && (N & 3) == 0 && (ldx & 3) == 0 && (ldz & 3) == 0 && amar_aligned16(X) && amar_aligned16(dZ) &&
        (int64_t)((K + 31) / 32) * ((N + 31) / 32) >= 16) {
        const size_t lds = (size_t)2 * WM_ROWS * WM_STRIDE * sizeof(float);
        static bool allowed[AMAR_MAX_DEVICES] = {};
        if (const int rc = amar_allow_lds(reinterpret_cast<const void *>(wgrad_mfma_kernel), lds, allowed)) return rc;
        hipLaunchKernelGGL(wgrad_mfma_kernel, dim3((unsigned)((K + 31) / 32), (unsigned)((N + 31) / 32)), dim3(256), lds, st, X, ldx, dZ, ldz, M, K, N, dW, db);
        return amar_check_launch();
    }
    const int Kk = dW ? K : 0;
    const int WG_ROWS = wg_rows(M);
    const int64_t chunks = (M + WG_ROWS - 1) / WG_ROWS;
    if (chunks > 0x7fffffff) return AMAR_EUNSUPPORTED;
    float *part_w = dW ? scratch : nullptr;
    float *part_b = db ? scratch + chunks * (int64_t)Kk * N : nullptr;
    const dim3 grid((unsigned)chunks, (unsigned)(dW ? (K + 15) / 16 : 1), (unsigned)((N + 15) / 16));
    hipLaunchKernelGGL(wgrad_partial_kernel, grid, dim3(256), 0, st, dW ? X : nullptr, ldx, dZ, ldz, M, Kk ? Kk : 1, N, part_w, part_b, WG_ROWS);
    if (dW && db) hipLaunchKernelGGL(reduce_partials2_kernel, dim3(grid1d((int64_t)K * N + N)), dim3(256), 0, st, part_w, (int64_t)K * N, dW,
                                     part_b, (int64_t)N, db, (int)chunks);
    else if (dW) hipLaunchKernelGGL(reduce_partials_kernel, dim3(grid1d((int64_t)K * N)), dim3(256), 0, st, part_w, (int)chunks, (int64_t)K * N, dW);
    else hipLaunchKernelGGL(reduce_partials_kernel, dim3(grid1d(N)), dim3(256), 0, st, part_b, (int)chunks, (int64_t)N, db);
    return amar_check_launch();
}

int amar_bce_grad_f32(const float *p, int64_t ldp, const float *y, float *dz, float *loss_terms, int64_t B, amar_stream_t stream) {
    if (B < 1 || !p || !y || !dz || !loss_terms || ldp < 1) return AMAR_EINVAL;
    hipLaunchKernelGGL(bce_grad_kernel, dim3(grid1d(B)), dim3(256), 0, static_cast<hipStream_t>(stream), p, ldp, y, dz, loss_terms, B);
    return amar_check_launch();
}

int amar_scatter_add_rows_f32(const float *src, int64_t lds, const int32_t *ids, int32_t base, float *dst, int64_t ldd,
                              int64_t M, int32_t W, amar_stream_t stream) {
    if (M < 0 || W < 1 || !src || !ids || !dst || lds < W || ldd < W) return AMAR_EINVAL;
    if (M == 0) return AMAR_OK;
    static const bool atomics = getenv("AMAR_SCATTER_ATOMIC") && atoi(getenv("AMAR_SCATTER_ATOMIC")) == 1;   // development switch (A/B)
    if (M <= SCATTER_OWNER_MAX && !atomics) {
        int64_t blocks = (M + 3) / 4;                                 // one wavefront per position, every workgroup holds the id list
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(scatter_add_rows_owner_kernel, dim3((unsigned)blocks), dim3(256), (size_t)M * sizeof(int32_t), static_cast<hipStream_t>(stream),
                           src, lds, ids, base, dst, ldd, (int)M, W);
        return amar_check_launch();
    }
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(grid1d(M * W)), dim3(256), 0, static_cast<hipStream_t>(stream), src, lds, ids, base, dst, ldd, M, W);
    return amar_check_launch();
}

int amar_add_inplace_f32(float *dst, int64_t ldd, const float *src, int64_t lds, int64_t M, int32_t W, float scale, amar_stream_t stream) {
    if (M < 0 || W < 1 || !src || !dst || lds < W || ldd < W) return AMAR_EINVAL;
    if (M == 0) return AMAR_OK;
    hipLaunchKernelGGL(add_inplace_kernel, dim3(grid1d(M * W)), dim3(256), 0, static_cast<hipStream_t>(stream), dst, ldd, src, lds, M, W, scale);
    return amar_check_launch();
}

int amar_row_affine_f32(const float *A, int64_t lda, const float *B, int64_t ldb, const float *scale, float *out, int64_t ldo,
                        int64_t M, int32_t W, amar_stream_t stream) {
    if (M < 0 || W < 1 || !A || !scale || !out || lda < W || ldo < W || (B && ldb < W)) return AMAR_EINVAL;
    if (M == 0) return AMAR_OK;
    hipLaunchKernelGGL(row_affine_kernel, dim3(grid1d(M * W)), dim3(256), 0, static_cast<hipStream_t>(stream), A, lda, B, ldb, scale, out, ldo, M, W);
    return amar_check_launch();
}

int amar_l2norm_fwd_f32(const float *Z, int64_t ldz, float *Nrm, int64_t ldn, float *inv, float *Y, int64_t ldy,
                        int64_t M, int32_t C, int32_t act, amar_stream_t stream) {
    if (M < 0 || C < 1 || !Z || !Nrm || !inv || !Y || ldz < C || ldn < C || ldy < C) return AMAR_EINVAL;
    if (act != AMAR_ACT_NONE && act != AMAR_ACT_RELU) return AMAR_EUNSUPPORTED;
    if (M == 0) return AMAR_OK;
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(grid1d(M)), dim3(256), 0, static_cast<hipStream_t>(stream), Z, ldz, Nrm, ldn, inv, Y, ldy, M, C, act == AMAR_ACT_RELU);
    return amar_check_launch();
}

int amar_l2norm_bwd_f32(const float *dY, int64_t ldd, const float *Nrm, int64_t ldn, const float *inv, float *dZ, int64_t ldz,
                        int64_t M, int32_t C, int32_t act, amar_stream_t stream) {
    if (M < 0 || C < 1 || !dY || !Nrm || !inv || !dZ || ldd < C || ldn < C || ldz < C) return AMAR_EINVAL;
    if (act != AMAR_ACT_NONE && act != AMAR_ACT_RELU) return AMAR_EUNSUPPORTED;
    if (M == 0) return AMAR_OK;
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(grid1d(M)), dim3(256), 0, static_cast<hipStream_t>(stream), dY, ldd, Nrm, ldn, inv, dZ, ldz, M, C, act == AMAR_ACT_RELU);
    return amar_check_launch();
}

int amar_gat_bwd_f32(const int32_t *rowptr, const int32_t *colidx, const float *H, int64_t ldh, int32_t C,
                     const float *s_self, const float *s_neigh, const float *Y, int64_t ldy, const float *dY, int64_t ldd,
                     const float *bias, const float *a_self, const float *a_neigh,
                     float *dout, float *row_scratch, float *ds, float *dt, float *dH, int64_t lddh,
                     int32_t self_loop, int32_t n_rows, amar_stream_t stream) {
    if (n_rows < 0 || !rowptr || !H || !s_self || !s_neigh || !Y || !dY || !bias || !a_self || !a_neigh || !dout ||
        !row_scratch || !ds || !dt || !dH) return AMAR_EINVAL;
    if (ldh < C || ldy < C || ldd < C || lddh < C || (ldh & 3) || (ldy & 3) || (ldd & 3) || (lddh & 3)) return AMAR_EINVAL;
    if (!amar_aligned16(H) || !amar_aligned16(Y) || !amar_aligned16(dY) || !amar_aligned16(bias) || !amar_aligned16(a_self) ||
        !amar_aligned16(a_neigh) || !amar_aligned16(dout) || !amar_aligned16(dH)) return AMAR_EINVAL;
    if (n_rows == 0) return AMAR_OK;
    if (!colidx) return AMAR_EINVAL;
    GatBwdArgs a{rowptr, colidx, H, ldh, s_self, s_neigh, Y, ldy, dY, ldd, bias, a_self, a_neigh, dout,
                 row_scratch, row_scratch + n_rows, row_scratch + 2 * (int64_t)n_rows, ds, dt, dH, lddh, self_loop ? 1 : 0, n_rows, C};
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (C < 4 || C > 64 || (C & 3)) return AMAR_EUNSUPPORTED;
    if (C <= 4) launch_gat_bwd<1>(a, st);
    else if (C <= 8) launch_gat_bwd<2>(a, st);
    else if (C <= 16) launch_gat_bwd<4>(a, st);
    else if (C <= 32) launch_gat_bwd<8>(a, st);
    else launch_gat_bwd<16>(a, st);
    return amar_check_launch();
}

int amar_attention_mix_f32(const float *A, int64_t lda, const float *B, int64_t ldb, const float *TA, int64_t ldta,
                           const float *TB, int64_t ldtb, float *out, int64_t ldo, int64_t M, int32_t D, amar_stream_t stream) {
    if (M < 0 || D < 1 || !A || !B || !TA || !TB || !out || lda < D || ldb < D || ldta < D || ldtb < D || ldo < D) return AMAR_EINVAL;
    if (M == 0) return AMAR_OK;
    hipLaunchKernelGGL(attention_mix_kernel, dim3(grid1d(M * D)), dim3(256), 0, static_cast<hipStream_t>(stream), A, lda, B, ldb, TA, ldta, TB, ldtb, out, ldo, M, D);
    return amar_check_launch();
}

int amar_attention_mix_bwd_f32(const float *dOut, int64_t ldd, const float *A, int64_t lda, const float *B, int64_t ldb,
                               const float *TA, int64_t ldta, const float *TB, int64_t ldtb,
                               float *dA, float *dB, float *dTA, float *dTB, int64_t M, int32_t D, amar_stream_t stream) {
    if (M < 0 || D < 1 || !dOut || !A || !B || !TA || !TB || !dA || !dB || !dTA || !dTB || ldd < D || lda < D || ldb < D ||
        ldta < D || ldtb < D) return AMAR_EINVAL;
    if (M == 0) return AMAR_OK;
    hipLaunchKernelGGL(attention_mix_bwd_kernel, dim3(grid1d(M * D)), dim3(256), 0, static_cast<hipStream_t>(stream), dOut, ldd, A, lda, B, ldb, TA, ldta, TB, ldtb, dA, dB, dTA, dTB, M, D);
    return amar_check_launch();
}

int amar_locality_scale_f32(const float *X, int64_t ldx, const float *w, float *out, int64_t ldo, int64_t M, int32_t W,
                            amar_stream_t stream) {
    if (M < 0 || W < 1 || !X || !w || !out || ldx < W || ldo < W) return AMAR_EINVAL;
    if (M == 0) return AMAR_OK;
    hipLaunchKernelGGL(locality_scale_kernel, dim3(grid1d(M * W)), dim3(256), 0, static_cast<hipStream_t>(stream), X, ldx, w, out, ldo, M, W);
    return amar_check_launch();
}

int amar_locality_scale_bwd_f32(const float *dOut, int64_t ldd, const float *X, int64_t ldx, const float *w, float *dX, int64_t lddx,
                                float *dw, int64_t M, int32_t W, int32_t accumulate, amar_stream_t stream) {
    if (M < 0 || W < 1 || !dOut || !X || !w || !dX || !dw || ldd < W || ldx < W || lddx < W) return AMAR_EINVAL;
    if (M == 0) return AMAR_OK;
    hipLaunchKernelGGL(locality_scale_bwd_kernel, dim3(grid1d(M)), dim3(256), 0, static_cast<hipStream_t>(stream), dOut, ldd, X, ldx, w, dX, lddx, dw, M, W, accumulate ? 1 : 0);
    return amar_check_launch();
}

int amar_add3_act_f32(const float *A, int64_t lda, const float *B, int64_t ldb, const float *C, int64_t ldc, float *out, int64_t ldo,
                      int64_t M, int32_t W, int32_t act, amar_stream_t stream) {
    if (M < 0 || W < 1 || !A || !B || !C || !out || lda < W || ldb < W || ldc < W || ldo < W) return AMAR_EINVAL;
    if (act != AMAR_ACT_NONE && act != AMAR_ACT_RELU && act != AMAR_ACT_SIGMOID) return AMAR_EINVAL;
    if (M == 0) return AMAR_OK;
    hipLaunchKernelGGL(add3_act_kernel, dim3(grid1d(M * W)), dim3(256), 0, static_cast<hipStream_t>(stream), A, lda, B, ldb, C, ldc, out, ldo, M, W, act);
    return amar_check_launch();
}

int amar_transpose_f32(const float *src, int32_t K, int32_t N, float *dst, amar_stream_t stream) {
    if (K < 1 || N < 1 || !src || !dst) return AMAR_EINVAL;
    hipLaunchKernelGGL(transpose_kernel, dim3(grid1d((int64_t)K * N)), dim3(256), 0, static_cast<hipStream_t>(stream), src, K, N, dst);
    return amar_check_launch();
}

int amar_adam_f32(float *w, const float *g, float *m, float *v, int64_t n, float lr_t, float beta_1, float beta_2,
                  float epsilon, float l2, amar_stream_t stream) {
    if (n < 0 || !w || !g || !m || !v) return AMAR_EINVAL;
    if (n == 0) return AMAR_OK;
    hipLaunchKernelGGL(adam_kernel, dim3(grid1d(n)), dim3(256), 0, static_cast<hipStream_t>(stream), w, g, m, v, n, lr_t, beta_1, beta_2, epsilon, 2.f * l2);
    return amar_check_launch();
}

int amar_adam_advance_f32(float *state, float learning_rate, float beta_1, float beta_2, amar_stream_t stream) {
    if (!state) return AMAR_EINVAL;
    hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), state, learning_rate, beta_1, beta_2);
    return amar_check_launch();
}

int amar_adam_dev_f32(float *w, const float *g, float *m, float *v, int64_t n, const float *state, float beta_1, float beta_2,
                      float epsilon, float l2, amar_stream_t stream) {
    if (n < 0 || !w || !g || !m || !v || !state) return AMAR_EINVAL;
    if (n == 0) return AMAR_OK;
    hipLaunchKernelGGL(adam_dev_kernel, dim3(grid1d(n)), dim3(256), 0, static_cast<hipStream_t>(stream), w, g, m, v, n, state, beta_1, beta_2, epsilon, 2.f * l2);
    return amar_check_launch();
}

int amar_adam_multi_f32(const amar_adam_slot *slots, int32_t n_slots, int64_t total_blocks, const float *state, float beta_1,
                        float beta_2, float epsilon, float reg_scale, float *loss_acc, amar_stream_t stream) {
    if (!slots || n_slots < 1 || total_blocks < 1 || total_blocks > 0x7fffffff || !state) return AMAR_EINVAL;
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, static_cast<hipStream_t>(stream), slots, n_slots, state,
                       beta_1, beta_2, epsilon, reg_scale, loss_acc);
    return amar_check_launch();
}

int amar_sum_into_f32(const float *x, int64_t n, float scale, float *acc, amar_stream_t stream) {
    if (n < 0 || !acc || (n > 0 && !x)) return AMAR_EINVAL;
    if (n == 0) return AMAR_OK;
    hipLaunchKernelGGL(sum_into_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), x, n, scale, acc);
    return amar_check_launch();
}

#ifdef AMAR_DS_STAMPS
int amar_ds_debug_copy(unsigned long long *host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ds_debug_stamps), (size_t)n * sizeof(unsigned long long));
}
#endif

}  // extern "C"
