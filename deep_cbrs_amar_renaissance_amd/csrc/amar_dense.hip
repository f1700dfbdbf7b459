// Scoring-head kernels for gfx950: fp32 MFMA dense layer with fused row gather, bias and
// activation.  Reference semantics: Keras Dense / Concatenate / tf.nn.embedding_lookup as cited
// in include/amar_hip.h.
//
// v_mfma_f32_32x32x2_f32 is exact fp32 (a k-ordered fmaf chain), so results match a plain fp32
// dot product bit for bit in k order; there is no reduced-precision path on gfx950.
//
// Tile: 128 (rows) x 64 (cols) per 256-thread workgroup, K stepped by 16 through LDS.
//   wave w owns rows [32w, 32w+32) and both 32-column halves -> 2 accumulators of 16 VGPRs.
//   LDS images are k-major so that the MFMA operand fetch (lane l: row/col l&31, k = l>>5)
//   is a conflict-free ds_read_b32 for A and B alike.
#include "amar_common.h"
#include <stdlib.h>
#include <string.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 64, BK = 16;   // BK = 32 (half the barriers, 25 KB of LDS per workgroup) measured 12 % slower (tools/exp_dense_ab.py)
constexpr int A_LD = BM + 4;     // +4 floats: the transposing store is at most 2-way conflicted
constexpr int B_LD = BN;

struct DenseArgs {
    const float *X; int64_t ldx; const int32_t *ids;
    const float *W; const float *bias; float *Y; int64_t ldy;
    int64_t M; int K; int N; int act;
    int w_trans;                                   // W holds the transpose: element (k, n) of the product's B sits at W[n * K + k]
    int n_col_blocks;
};

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == AMAR_ACT_RELU) return fmaxf(v, 0.f);
    if (act == AMAR_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
    return v;
}

// FULL: K a multiple of BK, N of BN, vector-aligned operands, W not transposed — the staging loads carry no guards (rows past
// M re-read row M - 1 and are dropped by the epilogue): the k loop is straight-line code.  On gfx950 an fp32 MFMA holds the
// SIMD's VALU port, so the ~100 scalar / vector instructions and the dozen divergent branches the guarded staging costs per
// k-tile came straight out of the matrix pipe's time (768 -> 256 -> 64 towers: 57 % of the fp32 MFMA peak with them).
template <bool VEC_X, bool VEC_W, bool FULL = false>
__global__ __launch_bounds__(256) void dense_mfma_kernel(const DenseArgs a) {
    __shared__ float As[BK * A_LD];
    __shared__ float Bs[BK * B_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Workgroup -> tile with XCD affinity: workgroup L runs on XCD L % 8; the column blocks of one row tile are consecutive
    // in ONE XCD's queue, so its X rows (128 x K) are fetched from HBM once and re-read from that XCD's L2 — with (row tile,
    // column block) on (blockIdx.x, blockIdx.y) every column block swept all of X again (4 passes over 1.2 GB for the
    // 768 -> 256 BERT tower).
    const int64_t L = blockIdx.x;
    const int64_t j = L >> 3;
    const int64_t m_blk = (j / a.n_col_blocks) * 8 + (L & 7);
    if (m_blk * BM >= a.M) return;
    const int64_t m0 = m_blk * BM;
    const int n0 = (int)(j % a.n_col_blocks) * BN;

    // staging assignment: X tile = 128 rows x 4 float4 -> 2 rows per thread; W tile = 16 x 16 float4 -> 1 per thread
    const int xr = tid >> 2, xq = tid & 3;
    int64_t src_row[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int64_t m = m0 + xr + 64 * h;
        if (FULL) { const int64_t mc = m < a.M ? m : a.M - 1; src_row[h] = a.ids ? (int64_t)a.ids[mc] : mc; }
        else src_row[h] = m < a.M ? (a.ids ? (int64_t)a.ids[m] : m) : -1;
    }
    const int wk = tid >> 4, wq = tid & 15;
    const float *xp0 = a.X + src_row[0] * a.ldx + 4 * xq, *xp1 = a.X + src_row[1] * a.ldx + 4 * xq;   // FULL: this thread's staging sources
    const float *wp = a.W + (int64_t)wk * a.N + n0 + 4 * wq;

    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }

    // global -> registers for the k-tile that starts at k0
    float xa[2][4], wb[4];
    auto fetch = [&](int k0) {
        if (FULL) {
            const float4 v0 = *reinterpret_cast<const float4 *>(xp0 + k0), v1 = *reinterpret_cast<const float4 *>(xp1 + k0);
            const float4 w4 = *reinterpret_cast<const float4 *>(wp + (int64_t)k0 * a.N);
            xa[0][0] = v0.x; xa[0][1] = v0.y; xa[0][2] = v0.z; xa[0][3] = v0.w;
            xa[1][0] = v1.x; xa[1][1] = v1.y; xa[1][2] = v1.z; xa[1][3] = v1.w;
            wb[0] = w4.x; wb[1] = w4.y; wb[2] = w4.z; wb[3] = w4.w;
            return;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = k0 + 4 * xq;
            if (src_row[h] >= 0 && VEC_X && k + 3 < a.K) {
                const float4 v = *reinterpret_cast<const float4 *>(a.X + src_row[h] * a.ldx + k);
                xa[h][0] = v.x; xa[h][1] = v.y; xa[h][2] = v.z; xa[h][3] = v.w;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    xa[h][i] = (src_row[h] >= 0 && k + i < a.K) ? a.X[src_row[h] * a.ldx + k + i] : 0.f;
            }
        }
        const int k = k0 + wk, n = n0 + 4 * wq;
        if (a.w_trans) {
#pragma unroll
            for (int i = 0; i < 4; ++i) wb[i] = (k < a.K && n + i < a.N) ? a.W[(int64_t)(n + i) * a.K + k] : 0.f;
        } else if (k < a.K && VEC_W && n + 3 < a.N) {
            const float4 v = *reinterpret_cast<const float4 *>(a.W + (int64_t)k * a.N + n);
            wb[0] = v.x; wb[1] = v.y; wb[2] = v.z; wb[3] = v.w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) wb[i] = (k < a.K && n + i < a.N) ? a.W[(int64_t)k * a.N + n + i] : 0.f;
        }
    };

    fetch(0);
    for (int k0 = 0; k0 < a.K; k0 += BK) {
        __syncthreads();                                // previous tile fully consumed
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i) As[(4 * xq + i) * A_LD + xr + 64 * h] = xa[h][i];
        *reinterpret_cast<float4 *>(&Bs[wk * B_LD + 4 * wq]) = make_float4(wb[0], wb[1], wb[2], wb[3]);
        __syncthreads();
        // the next k-tile travels from memory while this one is multiplied (on gfx950 nothing else overlaps an fp32 MFMA)
        if (k0 + BK < a.K) fetch(k0 + BK);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const int k = kk + (lane >> 5);
            const float av = As[k * A_LD + 32 * wave + (lane & 31)];
            const float b0 = Bs[k * B_LD + (lane & 31)];
            const float b1 = Bs[k * B_LD + 32 + (lane & 31)];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc1, 0, 0, 0);
        }
    }

    // epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int col = lane & 31;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int n = n0 + 32 * half + col;
        if (n >= a.N) continue;
        const float b = a.bias ? a.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (m < a.M) {
                const float v = (half == 0 ? acc0[r] : acc1[r]) + b;
                a.Y[m * a.ldy + n] = apply_act(v, a.act);
            }
        }
    }
}

// Batch-sized products of wide layers (1 024 rows x 768 -> 256: the first layer of a content tower inside model.fit or a per-batch
// predict): the tiles above make 32 workgroups of such a product, each walking K alone on its CU with one k-tile in flight — 38 us
// for 0.4 GFLOP.  Here a workgroup takes 64 rows x 64 columns (one 32 x 32 accumulator per wave: four times the workgroups), k-tiles of
// 32 with TWO of them in flight ahead of the one being multiplied.  Guard-free like FULL (K a multiple of 32, N of 64, aligned
// operands); the same instruction and the same (k, k + 1) pairing per step as the kernels above: bit-identical results.
constexpr int SB = 64, SK = 32, SA_LD = SB + 4, SB_LD = SB;
__global__ __launch_bounds__(256) void dense_mfma_small_kernel(const DenseArgs a) {
    __shared__ float As[SK * SA_LD];                                 // [k][row] (transposed on the way in)
    __shared__ float Bs[SK * SB_LD];                                 // [k][col]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t m0 = (int64_t)blockIdx.x * SB;
    const int n0 = blockIdx.y * SB;
    // staging: X tile = 64 rows x 8 float4 -> 2 per thread (rows xr, xr + 32); W tile = 32 k x 16 float4 -> 2 per thread (k rows wk, wk + 16)
    const int xr = tid >> 3, xq = tid & 7, wk = tid >> 4, wq = tid & 15;
    const float *xp[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int64_t m = m0 + xr + 32 * h, mc = m < a.M ? m : a.M - 1;          // (rows past M re-read row M - 1: dropped by the epilogue)
        xp[h] = a.X + (a.ids ? (int64_t)a.ids[mc] : mc) * a.ldx + 4 * xq;
    }
    const float *wp = a.W + (int64_t)wk * a.N + n0 + 4 * wq;
    float4 xa[2][2], wb[2][2];                                      // [slot][piece]: two k-tiles in flight
    auto fetch = [&](int slot, int k0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            xa[slot][h] = *reinterpret_cast<const float4 *>(xp[h] + k0);
            wb[slot][h] = *reinterpret_cast<const float4 *>(wp + (int64_t)(k0 + 16 * h) * a.N);
        }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int row_half = wave >> 1, col_half = wave & 1;
    auto multiply = [&](int slot) {
        __syncthreads();                                             // previous tile fully consumed
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float4 v = xa[slot][h];
            float *d = &As[(4 * xq) * SA_LD + xr + 32 * h];
            d[0] = v.x; d[SA_LD] = v.y; d[2 * SA_LD] = v.z; d[3 * SA_LD] = v.w;
            *reinterpret_cast<float4 *>(&Bs[(wk + 16 * h) * SB_LD + 4 * wq]) = wb[slot][h];
        }
        __syncthreads();
    };
    auto products = [&]() {
#pragma unroll
        for (int kk = 0; kk < SK; kk += 2) {
            const int k = kk + (lane >> 5);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[k * SA_LD + 32 * row_half + (lane & 31)], Bs[k * SB_LD + 32 * col_half + (lane & 31)], acc, 0, 0, 0);
        }
    };
    fetch(0, 0);
    if (SK < a.K) fetch(1, SK);
    for (int k0 = 0; k0 < a.K; k0 += 2 * SK) {                         // two k-tiles per trip: static register slots
        multiply(0);
        if (k0 + 2 * SK < a.K) fetch(0, k0 + 2 * SK);
        products();
        if (k0 + SK < a.K) {
            multiply(1);
            if (k0 + 3 * SK < a.K) fetch(1, k0 + 3 * SK);
            products();
        }
    }
    const int n = n0 + 32 * col_half + (lane & 31);
    const float b = a.bias ? a.bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t m = m0 + 32 * row_half + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < a.M) a.Y[m * a.ldy + n] = apply_act(acc[r] + b, a.act);
    }
}

// The guard-free form on a 128 x 128 tile (N a multiple of 128: the 768 -> 256 BERT layers): every wave keeps its 32 rows against all
// 128 columns in FOUR accumulators, so one A fragment read feeds four MFMAs (three LDS reads per four MFMAs instead of three per two)
// and a k-tile's two barriers are paid once per 32 MFMAs per wave instead of once per 16.  Same k order per output: bit-identical.
constexpr int BN2 = 128, B2_LD = BN2;
template <int BKT>                                                  // k-tile depth: 16 or 32
__global__ __launch_bounds__(256) void dense_mfma128_kernel(const DenseArgs a) {
    __shared__ float As[BKT * A_LD];
    __shared__ float Bs[BKT * B2_LD];
    constexpr int XQ = BKT / 4;                                     // float4 per X row and k-tile
    constexpr int XH = 128 * XQ / 256;                              // X float4 per thread (2 or 4)
    constexpr int WH = BKT * 32 / 256;                              // W float4 per thread (2 or 4)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t L = blockIdx.x;                                  // XCD-affine tile order, as in dense_mfma_kernel
    const int64_t j = L >> 3;
    const int64_t m_blk = (j / a.n_col_blocks) * 8 + (L & 7);
    if (m_blk * BM >= a.M) return;
    const int64_t m0 = m_blk * BM;
    const int n0 = (int)(j % a.n_col_blocks) * BN2;

    const int xr = tid / XQ, xq = tid % XQ;                         // X tile: thread's rows xr + (256 / XQ) h
    auto row_ptr = [&](int h) {
        const int64_t m = m0 + xr + (256 / XQ) * h;
        const int64_t mc = m < a.M ? m : a.M - 1;
        return a.X + (a.ids ? (int64_t)a.ids[mc] : mc) * a.ldx + 4 * xq;
    };
    const float *xp0 = row_ptr(0), *xp1 = row_ptr(1), *xp2 = XH > 2 ? row_ptr(2) : xp0, *xp3 = XH > 2 ? row_ptr(3) : xp0;
    const int wk = tid >> 5, wq = tid & 31;                        // W tile: k rows wk + 8 h, float4 column wq
    const float *wp = a.W + (int64_t)wk * a.N + n0 + 4 * wq;

    f32x16 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

    float4 xa0, xa1, xa2, xa3, wb0, wb1, wb2, wb3;
    xa2 = xa3 = wb2 = wb3 = make_float4(0.f, 0.f, 0.f, 0.f);
#define AMAR_D128_FETCH(k0_)                                                                                     \
    do {                                                                                                         \
        xa0 = *reinterpret_cast<const float4 *>(xp0 + (k0_));                                                    \
        xa1 = *reinterpret_cast<const float4 *>(xp1 + (k0_));                                                    \
        if (XH > 2) { xa2 = *reinterpret_cast<const float4 *>(xp2 + (k0_)); xa3 = *reinterpret_cast<const float4 *>(xp3 + (k0_)); } \
        wb0 = *reinterpret_cast<const float4 *>(wp + (int64_t)(k0_) * a.N);                                      \
        wb1 = *reinterpret_cast<const float4 *>(wp + (int64_t)((k0_) + 8) * a.N);                                \
        if (WH > 2) { wb2 = *reinterpret_cast<const float4 *>(wp + (int64_t)((k0_) + 16) * a.N);                 \
                      wb3 = *reinterpret_cast<const float4 *>(wp + (int64_t)((k0_) + 24) * a.N); }               \
    } while (0)
#define AMAR_D128_STAGE_X(v_, h_)                                                                                \
    do {                                                                                                         \
        const int r_ = xr + (256 / XQ) * (h_);                                                                   \
        As[(4 * xq + 0) * A_LD + r_] = (v_).x; As[(4 * xq + 1) * A_LD + r_] = (v_).y;                            \
        As[(4 * xq + 2) * A_LD + r_] = (v_).z; As[(4 * xq + 3) * A_LD + r_] = (v_).w;                            \
    } while (0)
    AMAR_D128_FETCH(0);
    for (int k0 = 0; k0 < a.K; k0 += BKT) {
        __syncthreads();                                            // previous tile fully consumed
        AMAR_D128_STAGE_X(xa0, 0);
        AMAR_D128_STAGE_X(xa1, 1);
        if (XH > 2) { AMAR_D128_STAGE_X(xa2, 2); AMAR_D128_STAGE_X(xa3, 3); }
        *reinterpret_cast<float4 *>(&Bs[wk * B2_LD + 4 * wq]) = wb0;
        *reinterpret_cast<float4 *>(&Bs[(wk + 8) * B2_LD + 4 * wq]) = wb1;
        if (WH > 2) {
            *reinterpret_cast<float4 *>(&Bs[(wk + 16) * B2_LD + 4 * wq]) = wb2;
            *reinterpret_cast<float4 *>(&Bs[(wk + 24) * B2_LD + 4 * wq]) = wb3;
        }
        __syncthreads();
        if (k0 + BKT < a.K) AMAR_D128_FETCH(k0 + BKT);
#pragma unroll
        for (int kk = 0; kk < BKT; kk += 2) {
            const int k = kk + (lane >> 5);
            const float av = As[k * A_LD + 32 * wave + (lane & 31)];
            const float *bp = &Bs[k * B2_LD + (lane & 31)];
            const float b0 = bp[0], b1 = bp[32], b2 = bp[64], b3 = bp[96];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b2, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b3, acc[3], 0, 0, 0);
        }
    }
    const int col = lane & 31;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int n = n0 + 32 * c + col;
        const float b = a.bias ? a.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (m < a.M) a.Y[m * a.ldy + n] = apply_act(acc[c][r] + b, a.act);
        }
    }
}

#undef AMAR_D128_FETCH
#undef AMAR_D128_STAGE_X

// ---- the same 128 x 128 tile with its products on the bf16 matrix instruction, both operands split three ways -----------------
// A finite f32 is exactly hi + mid + lo with each part its next 8 significand bits; bf16 x bf16 products are exact in f32 and the
// matrix pipe accumulates in f32; the six part products of weight >= 2^-16 are taken, small ones first, so a term x.w is off by at
// most 3 . 2^-24 |x.w| — the size of its own f32 rounding (csrc/amar_chain.hip has the same form for the scoring heads).  Six
// v_mfma_f32_32x32x16_bf16 (32 cycles each) cover k = 16 where the f32 instruction needs eight of 64 cycles, and a bf16 MFMA holds
// the SIMD's vector issue for 8 of its 32 cycles only: the splitting of the X tile (5 VALU instructions per value, once per value
// for all 128 columns) runs beside the matrix pipe.  W arrives pre-split (amar_dense_split_pack_f32: per (column block, k-tile of
// 32) one 24 KB image in the order of the LDS tile); X is split on its way from the staging registers to the LDS.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

struct DenseSplitArgs {
    const float *X; int64_t ldx; const int32_t *ids;
    const u32x4 *Wq; const float *bias; float *Y; int64_t ldy;
    int64_t M; int K; int N; int act; int n_col_blocks;
};

struct Parts { uint32_t h, m, l; };
__device__ __forceinline__ Parts split2(float a, float b) {       // (a, b) -> bf16 pairs (a in the low half) of the three parts
    const f32x2 x = {a, b};
    const u32x2 xb = __builtin_bit_cast(u32x2, x);
    Parts q;
    q.h = __builtin_amdgcn_perm(xb[1], xb[0], 0x07060302u);
    const f32x2 r1 = x - __builtin_bit_cast(f32x2, xb & 0xffff0000u);
    const u32x2 r1b = __builtin_bit_cast(u32x2, r1);
    q.m = __builtin_amdgcn_perm(r1b[1], r1b[0], 0x07060302u);
    const f32x2 r2 = r1 - __builtin_bit_cast(f32x2, r1b & 0xffff0000u);
    const u32x2 r2b = __builtin_bit_cast(u32x2, r2);
    q.l = __builtin_amdgcn_perm(r2b[1], r2b[0], 0x07060302u);
    return q;
}

constexpr int SPLIT_IMG = 3 * 2 * 2 * 128;                          // 16-byte entries of one operand image: [part][k-step][k half][row or column]

__global__ __launch_bounds__(256) void dense_split128_kernel(const DenseSplitArgs a) {
    __shared__ u32x4 As[SPLIT_IMG];
    __shared__ u32x4 Bs[SPLIT_IMG];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t L = blockIdx.x;                                  // XCD-affine tile order, as in dense_mfma_kernel
    const int64_t j = L >> 3;
    const int64_t m_blk = (j / a.n_col_blocks) * 8 + (L & 7);
    if (m_blk * BM >= a.M) return;
    const int64_t m0 = m_blk * BM;
    const int n_blk = (int)(j % a.n_col_blocks), n0 = n_blk * BN2;
    const int n_kt = a.K / 32;

    // X staging: thread -> row xr, the 16 floats of k-step xs of the k-tile (both halves of 8)
    const int xr = tid & 127, xs = tid >> 7;
    const int64_t m = m0 + xr, mc = m < a.M ? m : a.M - 1;
    const float *xp = a.X + (a.ids ? (int64_t)a.ids[mc] : mc) * a.ldx + 16 * xs;
    const u32x4 *wimg = a.Wq + (size_t)n_blk * n_kt * SPLIT_IMG + tid;

    f32x16 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

    float4 xa[4];
    u32x4 wb[6];
    auto fetch = [&](int kt) {
#pragma unroll
        for (int q = 0; q < 4; ++q) xa[q] = *reinterpret_cast<const float4 *>(xp + 32 * kt + 4 * q);
#pragma unroll
        for (int q = 0; q < 6; ++q) wb[q] = wimg[(size_t)kt * SPLIT_IMG + 256 * q];
    };
    fetch(0);
    const int frag = (lane >> 5) * 128 + (lane & 31);               // this lane's entry inside a [k half][row or column] plane
    for (int kt = 0; kt < n_kt; ++kt) {
        __syncthreads();                                            // previous tile fully consumed
        // (splitting behind the MFMAs of the tile before, outside the two barriers, holds the parts in 24 more registers: 190 VGPRs,
        // two waves per SIMD instead of three, 1.07 against 1.03 ms for the 768 -> 256 layer)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const float4 v0 = xa[2 * hf], v1 = xa[2 * hf + 1];
            const Parts q0 = split2(v0.x, v0.y), q1 = split2(v0.z, v0.w), q2 = split2(v1.x, v1.y), q3 = split2(v1.z, v1.w);
            u32x4 h, mm, lo;
            h[0] = q0.h; h[1] = q1.h; h[2] = q2.h; h[3] = q3.h;
            mm[0] = q0.m; mm[1] = q1.m; mm[2] = q2.m; mm[3] = q3.m;
            lo[0] = q0.l; lo[1] = q1.l; lo[2] = q2.l; lo[3] = q3.l;
            As[((0 * 2 + xs) * 2 + hf) * 128 + xr] = h;
            As[((1 * 2 + xs) * 2 + hf) * 128 + xr] = mm;
            As[((2 * 2 + xs) * 2 + hf) * 128 + xr] = lo;
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) Bs[tid + 256 * q] = wb[q];
        __syncthreads();
        if (kt + 1 < n_kt) fetch(kt + 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            // wave (wr, wc) owns rows 64 wr .. +63 against columns 64 wc .. +63: 6 + 6 fragment reads feed 24 MFMAs (32 rows against all
            // 128 columns would read 3 + 12: the LDS, not the matrix pipe, set the pace of this loop)
            bf16x8 af[2][3], bf[2][3];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    af[q][p] = __builtin_bit_cast(bf16x8, As[p * 512 + ks * 256 + frag + 64 * (wave >> 1) + 32 * q]);
                    bf[q][p] = __builtin_bit_cast(bf16x8, Bs[p * 512 + ks * 256 + frag + 64 * (wave & 1) + 32 * q]);
                }
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    f32x16 v = acc[2 * rb + cb];
                    v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rb][2], bf[cb][0], v, 0, 0, 0);
                    v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rb][0], bf[cb][2], v, 0, 0, 0);
                    v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rb][1], bf[cb][1], v, 0, 0, 0);
                    v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rb][1], bf[cb][0], v, 0, 0, 0);
                    v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rb][0], bf[cb][1], v, 0, 0, 0);
                    v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rb][0], bf[cb][0], v, 0, 0, 0);
                    acc[2 * rb + cb] = v;
                }
        }
    }
    const int col = lane & 31;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int n = n0 + 64 * (wave & 1) + 32 * (c & 1) + col;
        const float b = a.bias ? a.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t mr = m0 + 64 * (wave >> 1) + 32 * (c >> 1) + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (mr < a.M) a.Y[mr * a.ldy + n] = apply_act(acc[c][r] + b, a.act);
        }
    }
}

// ---- per-user top-k --------------------------------------------------------------------------
// One wave per user.  Round t picks the best pair that comes strictly after round t-1's winner in
// the order (score descending, item id ascending); items are distinct within a user.
__global__ __launch_bounds__(256) void topk_segmented_kernel(const int32_t *__restrict__ seg_ptr,
                                                              const int32_t *__restrict__ item_ids,
                                                              const float *__restrict__ scores, int n_users, int k,
                                                              int32_t *__restrict__ out_items,
                                                              float *__restrict__ out_scores) {
    const int lane = threadIdx.x & 63;
    const int u = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (u >= n_users) return;
    const int beg = seg_ptr[u], end = seg_ptr[u + 1];
    float prev_s = INFINITY;
    int prev_i = -1;
    for (int t = 0; t < k; ++t) {
        float best_s = -INFINITY;
        int best_i = 0x7fffffff;
        for (int p = beg + lane; p < end; p += 64) {
            const float s = scores[p];
            const int it = item_ids[p];
            const bool after_prev = (s < prev_s) || (s == prev_s && it > prev_i);
            const bool better = (s > best_s) || (s == best_s && it < best_i);
            if (after_prev && better) { best_s = s; best_i = it; }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float os = __shfl_xor(best_s, off, 64);
            const int oi = __shfl_xor(best_i, off, 64);
            if (os > best_s || (os == best_s && oi < best_i)) { best_s = os; best_i = oi; }
        }
        const bool found = best_i != 0x7fffffff;
        if (lane == 0) {
            out_items[(int64_t)u * k + t] = found ? best_i : -1;
            out_scores[(int64_t)u * k + t] = found ? best_s : -INFINITY;
        }
        if (!found) {
            if (lane == 0)
                for (int r = t + 1; r < k; ++r) { out_items[(int64_t)u * k + r] = -1; out_scores[(int64_t)u * k + r] = -INFINITY; }
            break;
        }
        prev_s = best_s; prev_i = best_i;
    }
}

}  // namespace

extern "C" {

int amar_dense_f32(const float *X, int64_t ldx, const int32_t *ids,
                   const float *W, const float *bias, float *Y, int64_t ldy,
                   int64_t M, int32_t K, int32_t N, int32_t act, amar_stream_t stream) {
    if (M < 0 || K < 1 || N < 1 || !X || !W || !Y || ldx < K || ldy < N) return AMAR_EINVAL;
    const int w_trans = (act & AMAR_DENSE_WT) ? 1 : 0;
    act &= ~AMAR_DENSE_WT;
    if (act != AMAR_ACT_NONE && act != AMAR_ACT_RELU && act != AMAR_ACT_SIGMOID) return AMAR_EINVAL;
    if (M == 0) return AMAR_OK;
    const int64_t gx = (M + BM - 1) / BM;
    if (gx > 0x7fffffffLL) return AMAR_EUNSUPPORTED;
    const int ny = (N + BN - 1) / BN;
    DenseArgs a{X, ldx, ids, W, bias, Y, ldy, M, K, N, act, w_trans, ny};
    const int64_t total = ((gx + 7) / 8) * 8 * ny;
    if (total > 0x7fffffffLL) return AMAR_EUNSUPPORTED;
    const dim3 grid((unsigned)total), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool vx = (ldx & 3) == 0 && amar_aligned16(X);
    const bool vw = (N & 3) == 0 && amar_aligned16(W);
    static const bool no_full = getenv("AMAR_DENSE_FULL") && atoi(getenv("AMAR_DENSE_FULL")) == 0;     // development switch (A/B timing)
    static const bool no_small = getenv("AMAR_DENSE_SMALL") && atoi(getenv("AMAR_DENSE_SMALL")) == 0;  // ... the batch-sized form
    if (vx && vw && !w_trans && K % SK == 0 && N % SB == 0 && M <= 4096 && !no_full && !no_small) {
        hipLaunchKernelGGL(dense_mfma_small_kernel, dim3((unsigned)((M + SB - 1) / SB), (unsigned)(N / SB)), block, 0, st, a);
        return amar_check_launch();
    }
    static const bool no_128 = getenv("AMAR_DENSE_128") && atoi(getenv("AMAR_DENSE_128")) == 0;        // ... the 128-column tile
    // the 128-column tile only where it still fills the chip: a batch-sized product (1 024 rows x 768 -> 256: the first layer of a content
    // tower inside model.fit) is 32 workgroups of it, each walking K alone on its CU — 64 us against 38 with the 64-column tile's 64 workgroups
    if (vx && vw && !w_trans && K % BK == 0 && N % BN2 == 0 && !no_full && !no_128 && ((gx + 7) / 8) * 8 * (N / BN2) >= 256) {
        DenseArgs a2 = a;
        a2.n_col_blocks = N / BN2;
        const int64_t total2 = ((gx + 7) / 8) * 8 * a2.n_col_blocks;
        static const int bk2 = getenv("AMAR_DENSE_BK") ? atoi(getenv("AMAR_DENSE_BK")) : 16;            // development switch
        if (bk2 == 32 && K % 32 == 0) hipLaunchKernelGGL(dense_mfma128_kernel<32>, dim3((unsigned)total2), block, 0, st, a2);
        else hipLaunchKernelGGL(dense_mfma128_kernel<16>, dim3((unsigned)total2), block, 0, st, a2);
        return amar_check_launch();
    }
    if (vx && vw && !w_trans && K % BK == 0 && N % BN == 0 && !no_full) hipLaunchKernelGGL((dense_mfma_kernel<true, true, true>), grid, block, 0, st, a);
    else if (vx && vw) hipLaunchKernelGGL((dense_mfma_kernel<true, true>), grid, block, 0, st, a);
    else if (vx) hipLaunchKernelGGL((dense_mfma_kernel<true, false>), grid, block, 0, st, a);
    else if (vw) hipLaunchKernelGGL((dense_mfma_kernel<false, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((dense_mfma_kernel<false, false>), grid, block, 0, st, a);
    return amar_check_launch();
}

// W [K, N] (row-major f32) -> the pre-split image dense_split128_kernel stages: for every (column block of 128, k-tile of 32) one
// [part][k-step of 16][k half of 8][column] array of 8 bf16 (k ascending); K % 32 == 0, N % 128 == 0.  HOST arrays.
int64_t amar_dense_split_bytes(int32_t K, int32_t N) {
    if (K < 32 || N < 128 || (K & 31) || (N & 127)) return AMAR_EUNSUPPORTED;
    return (int64_t)K * N * 6;
}

int amar_dense_split_pack_f32(const float *W, int32_t K, int32_t N, void *out) {
    if (!W || !out) return AMAR_EINVAL;
    if (K < 32 || N < 128 || (K & 31) || (N & 127)) return AMAR_EUNSUPPORTED;
    uint16_t *o = static_cast<uint16_t *>(out);
    const int n_kt = K / 32;
    for (int nb = 0; nb < N / 128; ++nb)
        for (int kt = 0; kt < n_kt; ++kt) {
            uint16_t *img = o + ((size_t)nb * n_kt + kt) * SPLIT_IMG * 8;
            for (int ks = 0; ks < 2; ++ks)
                for (int hf = 0; hf < 2; ++hf)
                    for (int col = 0; col < 128; ++col)
                        for (int jj = 0; jj < 8; ++jj) {
                            const float w = W[(size_t)(32 * kt + 16 * ks + 8 * hf + jj) * N + 128 * nb + col];
                            uint32_t wbits; memcpy(&wbits, &w, 4);
                            const uint32_t hb = wbits & 0xffff0000u;
                            float hf32; memcpy(&hf32, &hb, 4);
                            const float r1 = w - hf32;
                            uint32_t r1b; memcpy(&r1b, &r1, 4);
                            const uint32_t mb = r1b & 0xffff0000u;
                            float mf32; memcpy(&mf32, &mb, 4);
                            const float r2 = r1 - mf32;
                            uint32_t r2b; memcpy(&r2b, &r2, 4);
                            const size_t e = ((size_t)(ks * 2 + hf) * 128 + col) * 8 + jj;
                            img[e] = (uint16_t)(hb >> 16);
                            img[(size_t)1 * 2 * 2 * 128 * 8 + e] = (uint16_t)(mb >> 16);
                            img[(size_t)2 * 2 * 2 * 128 * 8 + e] = (uint16_t)(r2b >> 16);
                        }
        }
    return AMAR_OK;
}

// amar_dense_f32 on the split products (see dense_split128_kernel): Wq = amar_dense_split_pack_f32's image on the DEVICE.
int amar_dense_split_f32(const float *X, int64_t ldx, const int32_t *ids, const void *Wq, const float *bias, float *Y, int64_t ldy,
                         int64_t M, int32_t K, int32_t N, int32_t act, amar_stream_t stream) {
    if (M < 0 || K < 1 || N < 1 || !X || !Wq || !Y || ldx < K || ldy < N) return AMAR_EINVAL;
    if (act != AMAR_ACT_NONE && act != AMAR_ACT_RELU && act != AMAR_ACT_SIGMOID) return AMAR_EINVAL;
    if ((K & 31) || (N & 127) || (ldx & 3) || !amar_aligned16(X) || !amar_aligned16(Wq)) return AMAR_EUNSUPPORTED;
    if (M == 0) return AMAR_OK;
    const int64_t gx = (M + BM - 1) / BM;
    const int64_t total = ((gx + 7) / 8) * 8 * (N / BN2);
    if (total > 0x7fffffffLL) return AMAR_EUNSUPPORTED;
    DenseSplitArgs a{X, ldx, ids, static_cast<const u32x4 *>(Wq), bias, Y, ldy, M, K, N, act, N / BN2};
    hipLaunchKernelGGL(dense_split128_kernel, dim3((unsigned)total), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return amar_check_launch();
}

int amar_topk_segmented_f32(const int32_t *seg_ptr, const int32_t *item_ids, const float *scores,
                            int32_t n_users, int32_t k, int32_t *out_items, float *out_scores,
                            amar_stream_t stream) {
    if (n_users < 0 || k < 1 || !seg_ptr) return AMAR_EINVAL;
    if (k > 64) return AMAR_EUNSUPPORTED;
    if (n_users == 0) return AMAR_OK;                                    // nothing to rank: the outputs may be empty (NULL)
    if (!item_ids || !scores || !out_items || !out_scores) return AMAR_EINVAL;
    hipLaunchKernelGGL(topk_segmented_kernel, dim3((n_users + 3) / 4), dim3(256), 0,
                       static_cast<hipStream_t>(stream), seg_ptr, item_ids, scores, n_users, k, out_items, out_scores);
    return amar_check_launch();
}

}  // extern "C"
