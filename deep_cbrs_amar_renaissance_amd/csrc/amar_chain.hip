// Fused gather + dense chain for the scoring heads (gfx950, fp32 MFMA 16x16x4).
//
// Computes, for every row p of a batch (a (user, item) pair, or an entity row):
//     x   = [ A[ida(p), 0:Da] || B[idb(p), 0:Db] ]                     gather + Concatenate
//     x   = act_l( x . W_l + b_l )   for each layer l                    Keras Dense stack
// and writes either the last layer's [P, N] block or, when the last layer has one unit, the
// [P] scores.  Reference semantics: tf.nn.embedding_lookup + Concatenate + Dense stacks at
// src/models/basic.py:31-37,72-75, src/models/hybrid.py:72-89, src/models/dense.py:4-17.
//
// Formulation (transposed): Out^T[feature, pair] = W^T . X^T on v_mfma_f32_16x16x4_f32 with the
// pair on the MFMA column (lane & 15) and features on rows.  The C/D layout of that instruction
// (lane l, register r <-> row 4*(l>>4)+r, column l&15) is exactly its B-operand layout for
// k = 4*(l>>4)+r, so a layer's accumulators — after bias (pre-loaded as the initial accumulator)
// and ReLU — feed the next layer's MFMAs as B operands with NO cross-lane movement and no LDS
// round trip: activations never leave registers.  The gathered input rows arrive in the same
// layout for free: lane l loads the float4 at feature offset 16t + 4*(l>>4) of pair l&15.
// Weights are pre-packed on the host into A-operand fragment order (one ds_read_b128 per lane
// feeds 4 MFMAs x PT pair tiles) and live in LDS for the whole launch.
//
// f32 MFMA is an exact k-ordered fmaf chain, so results do not depend on tiling; the k order
// differs from a plain dot product only by the fixed interleave (4g + r), well inside 1e-6.
#include "amar_common.h"
#include <stdlib.h>
#include <string.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CHAIN_MAX_LAYERS = 8;
constexpr int CHAIN_MAX_SEG = 8;

struct ChainArgs {
    const float *A; int64_t lda; int Da; const int32_t *ids_a; int base_a;
    // n_seg > 0 (chain_rows_kernel<.., SEG = true> only): row p of the A input is [seg[0][p] || seg[1][p] || ...], the concatenation of
    // n_seg tables with the same row numbering — a 'concatenation' reduction (reduction.py:15-17) read in place from the per-layer
    // tables, never materialised.  seg_off[s] = first feature of table s (multiples of 4), seg_off[n_seg] = Da.
    int n_seg; const float *seg[CHAIN_MAX_SEG]; int64_t seg_ld[CHAIN_MAX_SEG]; int seg_off[CHAIN_MAX_SEG + 1];
    const float *B; int64_t ldb; int Db; const int32_t *ids_b; int base_b;
    const float *wpack; int wpack_floats;
    int sum_inputs, in_act;             // x = in_act(A[ida] + B[idb]) instead of [A[ida] || B[idb]]
    int n_layers;                       // MFMA layers (a trailing 1-unit layer is the VALU `dot` stage)
    int kt[CHAIN_MAX_LAYERS], nt[CHAIN_MAX_LAYERS], act[CHAIN_MAX_LAYERS];
    int w_off[CHAIN_MAX_LAYERS], b_off[CHAIN_MAX_LAYERS];
    int has_dot, dot_off, dot_bias_off, dot_act, dot_kt;
    int n_out;                          // width of the written block (last MFMA layer) when !has_dot
    float *out; int64_t ldo; int64_t P;
    const int32_t *out_index;           // row p is written to out row out_index[p] (a pair list kept in another order), or NULL
};

__device__ __forceinline__ int64_t chain_out_row(const ChainArgs &a, int64_t p) { return a.out_index ? (int64_t)a.out_index[p] : p; }

__device__ __forceinline__ float chain_act(float v, int act) {
    if (act == AMAR_ACT_RELU) return fmaxf(v, 0.f);
    if (act == AMAR_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
    return v;
}

// RELU (template flag of the kernels below): the summed-input activation and every MFMA layer's activation are ReLU — true
// for every head of the reference's configs (dense.py:4-17 applies one `activation` throughout, 'relu' in config.yaml:27).
// With the activation a run-time value every one of the 48 activated registers per 32 pairs went through a chain of
// scalar branches (3 000 instructions in the loop, the VALU as busy as the matrix pipe); known at compile time the loop
// body is straight-line code.
template <bool RELU>
__device__ __forceinline__ float chain_act_t(float v, int act) { return RELU ? fmaxf(v, 0.f) : chain_act(v, act); }
// AM (activation mode of chain_kernel): 0 = run-time values, 1 = ReLU everywhere, 2 = ReLU everywhere but the LAST MFMA layer, which is
// linear — the per-entity towers that end in the folded half of the classifier's first layer (models/basic.py:_split_plan).

// FULL: every MFMA layer has exactly MAXT input and MAXT output tiles (e.g. grid1's 48 -> 48 -> 48 classifier with
// MAXT = 3): the tile guards fold away and the layer bodies become straight-line MFMA code.
template <int MAXT, int PT, bool FULL, int AM>
__global__ __launch_bounds__(256) void chain_kernel(const ChainArgs a) {
    constexpr bool RELU = AM != 0;
    extern __shared__ __attribute__((aligned(16))) float w_lds[];
    for (int i = threadIdx.x * 4; i < a.wpack_floats; i += blockDim.x * 4)
        *reinterpret_cast<float4 *>(&w_lds[i]) = *reinterpret_cast<const float4 *>(a.wpack + i);
    __syncthreads();

    const int lane = threadIdx.x & 63, g = lane >> 4, col = lane & 15;
    const int64_t pairs_per_wave = 16 * PT;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t stride = (int64_t)gridDim.x * 4 * pairs_per_wave;

    for (int64_t base = wave0 * pairs_per_wave; base < a.P; base += stride) {
        f32x4 x[MAXT][PT];
        // ---- gather + concatenate: lane (g, col) holds features 16t+4g .. +3 of pair base + 16*pt + col
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const int64_t p = base + 16 * pt + col;
            const bool ok = p < a.P;
            const int64_t ra = ok ? (a.ids_a ? (int64_t)a.ids_a[p] - a.base_a : p) : 0;
            const int64_t rb = (ok && a.Db) ? (a.ids_b ? (int64_t)a.ids_b[p] - a.base_b : p) : 0;
#pragma unroll
            for (int t = 0; t < MAXT; ++t) {
                const int f = 16 * t + 4 * g;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if ((FULL || t < a.kt[0]) && ok) {
                    if (a.sum_inputs) {
                        if (f < a.Da) {
                            const f32x4 va = *reinterpret_cast<const f32x4 *>(a.A + ra * a.lda + f);
                            const f32x4 vb = *reinterpret_cast<const f32x4 *>(a.B + rb * a.ldb + f);
                            v = va + vb;
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = chain_act_t<RELU>(v[r], a.in_act);
                        }
                    } else if (f < a.Da) v = *reinterpret_cast<const f32x4 *>(a.A + ra * a.lda + f);
                    else if (f < a.Da + a.Db) v = *reinterpret_cast<const f32x4 *>(a.B + rb * a.ldb + (f - a.Da));
                }
                x[t][pt] = v;
            }
        }
        // ---- dense layers on MFMA, activations stay in registers
        for (int l = 0; l < a.n_layers; ++l) {
            const int KT = FULL ? MAXT : a.kt[l], NT = FULL ? MAXT : a.nt[l];
            const float *wl = w_lds + a.w_off[l];
            const float *bl = w_lds + a.b_off[l];
            f32x4 y[MAXT][PT];
#pragma unroll
            for (int m = 0; m < MAXT; ++m) {
                if (FULL || m < NT) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bl + 16 * m + 4 * g);
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) y[m][pt] = b4;
#pragma unroll
                    for (int t = 0; t < MAXT; ++t) {
                        if (FULL || t < KT) {
                            const f32x4 w4 = *reinterpret_cast<const f32x4 *>(wl + ((m * KT + t) * 64 + lane) * 4);
#pragma unroll
                            for (int r = 0; r < 4; ++r)
#pragma unroll
                                for (int pt = 0; pt < PT; ++pt)
                                    y[m][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[r], x[t][pt][r], y[m][pt], 0, 0, 0);
                        }
                    }
                }
            }
            const int act = a.act[l];
            if (AM == 2 && l == a.n_layers - 1) {                              // the linear last layer
#pragma unroll
                for (int m = 0; m < MAXT; ++m)
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) {
                        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                        x[m][pt] = (FULL || m < NT) ? y[m][pt] : zero;
                    }
            } else {
#pragma unroll
                for (int m = 0; m < MAXT; ++m)
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) {
                        f32x4 v = {0.f, 0.f, 0.f, 0.f};
                        if (FULL || m < NT) {
                            v = y[m][pt];
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = chain_act_t<RELU>(v[r], act);
                        }
                        x[m][pt] = v;
                    }
            }
        }
        // ---- output
        if (a.has_dot) {
            const float *wd = w_lds + a.dot_off;
            const float bd = w_lds[a.dot_bias_off];
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                float s = 0.f;
#pragma unroll
                for (int t = 0; t < MAXT; ++t) {
                    if (FULL || t < a.dot_kt) {
                        const f32x4 w4 = *reinterpret_cast<const f32x4 *>(wd + 16 * t + 4 * g);
#pragma unroll
                        for (int r = 0; r < 4; ++r) s = fmaf(x[t][pt][r], w4[r], s);
                    }
                }
                s += __shfl_xor(s, 16, 64);
                s += __shfl_xor(s, 32, 64);
                const int64_t p = base + 16 * pt + col;
                if (g == 0 && p < a.P) a.out[chain_out_row(a, p) * a.ldo] = chain_act(s + bd, a.dot_act);
            }
        } else {
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                const int64_t p = base + 16 * pt + col;
#pragma unroll
                for (int m = 0; m < MAXT; ++m) {
                    const int f = 16 * m + 4 * g;
                    if (p < a.P && f < a.n_out) *reinterpret_cast<f32x4 *>(a.out + chain_out_row(a, p) * a.ldo + f) = x[m][pt];
                }
            }
        }
    }
}

// ReLU as one integer max on the float's bits (no NaN-quieting pre-pass).
// Identical to fmaxf(v, 0) for every finite v (and -0, -NaN -> 0).  A positive-sign NaN stays a NaN here while fmaxf(NaN, 0) = 0
// in the generic kernel: not reachable from finite weights and rows, so the two kernels agree bit for bit on real inputs.
__device__ __forceinline__ float relu_bits(float v) { return __int_as_float(max(__float_as_int(v), 0)); }

// Entity-tower form of the same chain: ONE input table (rows themselves or ids), no trailing dot, ReLU layers with an optionally linear
// last one, and the tile counts of every layer known at COMPILE time (SHAPE = layer count | tiles of dims[0] << 3 | tiles of
// dims[1] << 6 | ...).  The generic kernel above walks its layers with run-time tile guards — a scalar branch around every weight
// fragment — and asks for its rows at the top of an iteration; here the layer bodies are straight-line MFMA code and the rows of
// iteration k+1 are requested right after layer 0 of iteration k has consumed the current ones (unconditionally: positions past the
// end re-read the last row).  Same arithmetic in the same order: bit-identical to the generic kernel.
constexpr int chain_shape(int nl, int t0, int t1, int t2 = 0, int t3 = 0) { return nl | t0 << 3 | t1 << 6 | t2 << 9 | t3 << 12; }
constexpr int shape_nl(int s) { return s & 7; }
constexpr int shape_t(int s, int j) { return (s >> (3 * (j + 1))) & 7; }
constexpr int shape_maxt(int s) {
    int m = 0;
    for (int j = 0; j <= shape_nl(s); ++j) m = shape_t(s, j) > m ? shape_t(s, j) : m;
    return m;
}

template <int SHAPE, int PT, bool LASTLIN, bool SEG>
__global__ __launch_bounds__(256) void chain_rows_kernel(const ChainArgs a) {
    constexpr int NL = shape_nl(SHAPE), T0 = shape_t(SHAPE, 0), TN = shape_t(SHAPE, NL), MAXT = shape_maxt(SHAPE);
    extern __shared__ __attribute__((aligned(16))) float w_lds[];
    for (int i = threadIdx.x * 4; i < a.wpack_floats; i += blockDim.x * 4)
        *reinterpret_cast<float4 *>(&w_lds[i]) = *reinterpret_cast<const float4 *>(a.wpack + i);
    __syncthreads();

    const int lane = threadIdx.x & 63, g = lane >> 4, col = lane & 15;
    constexpr uint32_t rows_per_wave = 16 * PT;
    const uint32_t stride = gridDim.x * 4 * rows_per_wave;
    const uint32_t P = (uint32_t)a.P, last = P - 1;                 // (the launcher checks P + 2 strides < 2^30 and the row bytes)
    const uint32_t lda = (uint32_t)a.lda * 4u;
    const int f_last = 16 * (T0 - 1) + 4 * g;                      // this lane's features of the last input tile: past Da -> zeros
    const bool tail_ok = f_last < a.Da;
    const char *Ag = reinterpret_cast<const char *>(a.A) + 16 * g;
    const int tail_off = tail_ok ? 64 * (T0 - 1) : -16 * g;        // (a lane past Da in the last tile re-reads the row's first features)
    // SEG: this lane's float4 of input tile t (features 16t + 4g .. +3) lies in ONE of the concatenated tables (their widths are
    // multiples of 4): its column-0 address and row bytes, fixed for the launch (a lane past Da takes table 0's first features)
    const char *seg_base[SEG ? T0 : 1];
    uint32_t seg_ldb[SEG ? T0 : 1];
    if constexpr (SEG) {
#pragma unroll
        for (int t = 0; t < T0; ++t) {
            const int f = 16 * t + 4 * g;
            int sidx = 0;
            for (int j = 1; j < a.n_seg; ++j) sidx = (f < a.Da && f >= a.seg_off[j]) ? j : sidx;
            seg_base[t] = reinterpret_cast<const char *>(a.seg[sidx] + (f < a.Da ? f - a.seg_off[sidx] : 0));
            seg_ldb[t] = (uint32_t)a.seg_ld[sidx] * 4u;
        }
    }

    f32x4 xin[MAXT][PT];
    auto gather = [&](uint32_t b) {
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const uint32_t p = min(b + 16 * pt + col, last);
            const uint32_t row = a.ids_a ? (uint32_t)(a.ids_a[p] - a.base_a) : p;
            const char *pa = Ag + (uint64_t)row * lda;
#pragma unroll
            for (int t = 0; t < T0; ++t) {
                if constexpr (SEG) xin[t][pt] = *reinterpret_cast<const f32x4 *>(seg_base[t] + (uint64_t)row * seg_ldb[t]);
                else xin[t][pt] = *reinterpret_cast<const f32x4 *>(pa + (t == T0 - 1 ? tail_off : 64 * t));
            }
        }
    };

    uint32_t base = (blockIdx.x * 4 + (threadIdx.x >> 6)) * rows_per_wave;
    if (base >= P) return;
    gather(base);
    for (; base < P; base += stride) {
        f32x4 x[MAXT][PT];
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const int KT = shape_t(SHAPE, l), NT = shape_t(SHAPE, l + 1);
            const float *wl = w_lds + a.w_off[l];
            const float *bl = w_lds + a.b_off[l];
            f32x4 y[MAXT][PT];
#pragma unroll
            for (int m = 0; m < MAXT; ++m) {
                if (m < NT) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bl + 16 * m + 4 * g);
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) y[m][pt] = b4;
#pragma unroll
                    for (int t = 0; t < MAXT; ++t) {
                        if (t < KT) {
                            const f32x4 w4 = *reinterpret_cast<const f32x4 *>(wl + ((m * KT + t) * 64 + lane) * 4);
#pragma unroll
                            for (int r = 0; r < 4; ++r)
#pragma unroll
                                for (int pt = 0; pt < PT; ++pt) {
                                    float xv = l == 0 ? xin[t][pt][r] : x[t][pt][r];
                                    if (l == 0 && t == T0 - 1) xv = tail_ok ? xv : 0.f;
                                    y[m][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[r], xv, y[m][pt], 0, 0, 0);
                                }
                        }
                    }
                }
            }
            if (l == 0) gather(base + stride);                      // the current rows are consumed: request the next iteration's
#pragma unroll
            for (int m = 0; m < MAXT; ++m)
#pragma unroll
                for (int pt = 0; pt < PT; ++pt)
                    if (m < NT) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) x[m][pt][r] = (LASTLIN && l == NL - 1) ? y[m][pt][r] : relu_bits(y[m][pt][r]);
                    }
        }
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const uint32_t p = base + 16 * pt + col;
            float *po = a.out + (int64_t)p * a.ldo + 4 * g;
#pragma unroll
            for (int m = 0; m < TN; ++m)
                if (p < P && 16 * m + 4 * g < a.n_out) *reinterpret_cast<f32x4 *>(po + 16 * m) = x[m][pt];
        }
    }
}

// Pair-stage form of the same chain (x = relu(A[ida] + B[idb]), every layer MAXT x MAXT tiles with ReLU, Da = 16 MAXT).
//
// What bounds this stage (tools/micro/mfma_valu_overlap.hip, tools/exp_chain_bound.py): on gfx950 an fp32 MFMA holds the
// SIMD's VALU port for its whole duration — MFMA cycles and plain VALU cycles ADD UP, also across the waves of a SIMD —
// so the launch time is (72 MFMAs x 32 cycles) + 4 x (VALU instructions) per 32 pairs, plus whatever memory latency is
// left exposed.  The generic kernel above spends ~400 VALU instructions per 32 pairs (0.72 ms for 12.1 M pairs with the
// memory system taken out of the picture, against 0.35 ms of MFMA work) and waits for its gathers six times per
// iteration.  This kernel therefore
//   * issues the 12 gathers of iteration k+1 before the MFMA block of iteration k (raw A and B rows in their own
//     registers: they are consumed — summed, ReLU — right before the next ones are requested) and keeps the ids two
//     iterations ahead;
//   * takes ReLU as one integer max on the float's bits (no NaN-quieting pre-pass), forms addresses from 32-bit row
//     numbers, and evaluates the final activation once per 32 pairs (lane group g finishes pair tile g).
// Same arithmetic in the same order as the generic kernel: the scores are bit-identical.

// SCATTER: scores go to out[out_index[p]] (a pair list prepared in XCD-affine order writes back in the caller's order); the index is
// requested at the top of the iteration, unconditionally like every other load here, and has the whole MFMA block to arrive.
//
// SPLIT (the default since round 3; AMAR_PAIR_MFMA=f32 keeps the f32 instruction): the layers' products run on
// v_mfma_f32_16x16x32_bf16 with BOTH operands split into three bf16 parts.  A finite f32 is EXACTLY hi + mid + lo with each part
// its next 8 significand bits (truncation splits: x - hi is exact, and so on), every bf16 x bf16 product is exact in f32, and the matrix
// pipe accumulates in f32; of the nine part products the six with weight >= 2^-16 are taken (lo.hi, hi.lo, mid.mid, mid.hi, hi.mid,
// hi.hi — small ones first), so a term x.w is off by at most 3 . 2^-24 |x.w|: the size of the f32 rounding of that product itself.
// Not bit-identical to the f32 MFMA chain (tests/test_kernels_gpu.py bounds the difference); what it buys: 6 MFMAs of 16 cycles cover
// k = 32 where the f32 instruction needs 8 of 32 cycles, and a bf16 MFMA holds the SIMD's vector issue for 8 of its 16 cycles only,
// so the splitting (4.5 VALU instructions per activation) and the rest of the loop run beside the matrix pipe instead of behind it.
// k-slot j of k-step s at lane group g is feature 16 (2s + j/4) + 4g + j%4: the two f32 tiles 2s, 2s+1 the lane already holds, so the
// gathers, the register-resident hand-over between layers and the host-side weight pack are those of the f32 form — the workgroup
// splits its LDS copy of the packed weights into bf16 fragments once, at its start.  With an odd tile count (48 = 3 tiles) the spare
// half k-step carries the bias (activation 1.0 at feature 16 MAXT, the bias in the weight's row), so accumulators start at 0.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// a, b -> (hi, mid, lo) bf16 pairs, a in the low half.  The two remainders of a level come from ONE packed subtraction.
struct Split3 { uint32_t h, m, l; };
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ Split3 split_pair(float a, float b) {
    const f32x2 x = {a, b};
    const u32x2 xb = __builtin_bit_cast(u32x2, x);
    Split3 q;
    q.h = __builtin_amdgcn_perm(xb[1], xb[0], 0x07060302u);
    const f32x2 r1 = x - __builtin_bit_cast(f32x2, xb & 0xffff0000u);
    const u32x2 r1b = __builtin_bit_cast(u32x2, r1);
    q.m = __builtin_amdgcn_perm(r1b[1], r1b[0], 0x07060302u);
    const f32x2 r2 = r1 - __builtin_bit_cast(f32x2, r1b & 0xffff0000u);
    const u32x2 r2b = __builtin_bit_cast(u32x2, r2);
    q.l = __builtin_amdgcn_perm(r2b[1], r2b[0], 0x07060302u);
    return q;
}

// two f32 tiles of a lane (k-slots 0..3 and 4..7 of one bf16 k-step) -> the three bf16 operands
struct Split3x4 { u32x4 h, m, l; };
__device__ __forceinline__ Split3x4 split_tiles(const f32x4 &t0, const f32x4 &t1) {
    const Split3 q0 = split_pair(t0[0], t0[1]), q1 = split_pair(t0[2], t0[3]), q2 = split_pair(t1[0], t1[1]), q3 = split_pair(t1[2], t1[3]);
    Split3x4 o;
    o.h[0] = q0.h; o.h[1] = q1.h; o.h[2] = q2.h; o.h[3] = q3.h;
    o.m[0] = q0.m; o.m[1] = q1.m; o.m[2] = q2.m; o.m[3] = q3.m;
    o.l[0] = q0.l; o.l[1] = q1.l; o.l[2] = q2.l; o.l[3] = q3.l;
    return o;
}
// acc += W . X over one k-step of 32 with both operands split: the six part products of weight >= 2^-16, small ones first
__device__ __forceinline__ f32x4 split_mfma(const u32x4 *w, const Split3x4 &x, f32x4 acc) {
    const bf16x8 wh = __builtin_bit_cast(bf16x8, w[0]), wm = __builtin_bit_cast(bf16x8, w[64]), wl = __builtin_bit_cast(bf16x8, w[128]);
    const bf16x8 xh = __builtin_bit_cast(bf16x8, x.h), xm = __builtin_bit_cast(bf16x8, x.m), xl = __builtin_bit_cast(bf16x8, x.l);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh, acc, 0, 0, 0);
    return acc;
}

template <int MAXT, int PT, bool SCATTER, bool SPLIT>
__global__ __launch_bounds__(256) void chain_pipe_kernel(const ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float w_lds[];
    for (int i = threadIdx.x * 4; i < a.wpack_floats; i += blockDim.x * 4)
        *reinterpret_cast<float4 *>(&w_lds[i]) = *reinterpret_cast<const float4 *>(a.wpack + i);
    __syncthreads();

    const int lane = threadIdx.x & 63, g = lane >> 4, col = lane & 15;
    constexpr int KS = (MAXT + 1) / 2;                       // bf16 k-steps of 32 (two f32 tiles each)
    constexpr bool BIASK = SPLIT && (MAXT & 1);              // the bias rides in the spare half k-step
    u32x4 *frag = reinterpret_cast<u32x4 *>(w_lds + ((a.wpack_floats + 3) & ~3));   // bf16 fragments [layer][m][s][part][lane]
    if (SPLIT) {
        for (int idx = threadIdx.x >> 6; idx < a.n_layers * MAXT * KS; idx += 4) {
            const int l = idx / (MAXT * KS), m = (idx % (MAXT * KS)) / KS, sk = idx % KS;
            const float *wl = w_lds + a.w_off[l];
            const f32x4 wa = *reinterpret_cast<const f32x4 *>(wl + ((m * MAXT + 2 * sk) * 64 + lane) * 4);
            f32x4 wb = {0.f, 0.f, 0.f, 0.f};
            if (2 * sk + 1 < MAXT) wb = *reinterpret_cast<const f32x4 *>(wl + ((m * MAXT + 2 * sk + 1) * 64 + lane) * 4);
            else if (BIASK && g == 0) wb[0] = w_lds[a.b_off[l] + 16 * m + col];
            u32x4 *dst = frag + idx * 3 * 64 + lane;
            const Split3 q0 = split_pair(wa[0], wa[1]), q1 = split_pair(wa[2], wa[3]), q2 = split_pair(wb[0], wb[1]), q3 = split_pair(wb[2], wb[3]);
            u32x4 h, mm, lo;
            h[0] = q0.h; h[1] = q1.h; h[2] = q2.h; h[3] = q3.h;
            mm[0] = q0.m; mm[1] = q1.m; mm[2] = q2.m; mm[3] = q3.m;
            lo[0] = q0.l; lo[1] = q1.l; lo[2] = q2.l; lo[3] = q3.l;
            dst[0] = h; dst[64] = mm; dst[128] = lo;
        }
        __syncthreads();
    }
    // Positions are 32-bit (the launcher checks P + 3 strides < 2^30): the ids' byte offsets stay 32-bit next to a scalar base
    // and a row address is ONE v_mad_u64_u32 (row x row bytes + base) — a dozen 64-bit VALU operations per iteration gone.
    constexpr uint32_t pairs_per_wave = 16 * PT;
    const uint32_t wave0 = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t stride = gridDim.x * 4 * pairs_per_wave;
    const uint32_t P = (uint32_t)a.P, last = P - 1;
    const char *Ag = reinterpret_cast<const char *>(a.A + 4 * g), *Bg = reinterpret_cast<const char *>(a.B + 4 * g);
    const uint32_t lda = (uint32_t)a.lda * 4u, ldb = (uint32_t)a.ldb * 4u;   // row bytes (the launcher checks that both fit 32 bits)
    const char *ids_a = reinterpret_cast<const char *>(a.ids_a), *ids_b = reinterpret_cast<const char *>(a.ids_b);

    // Ids and gathers are requested UNCONDITIONALLY (positions past the end re-read the last pair, whose rows exist): an
    // exec-masked load is followed by its own s_waitcnt, which would also wait for every gather still in flight, and a
    // conditionally overwritten register set costs a copy of all 48 registers per iteration.
    auto load_ids = [&](uint32_t b, int32_t (&ra)[PT], int32_t (&rb)[PT]) {
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const uint32_t off = min(b + 16 * pt + col, last) << 2;
            ra[pt] = *reinterpret_cast<const int32_t *>(ids_a + off);
            rb[pt] = *reinterpret_cast<const int32_t *>(ids_b + off);
        }
    };
    auto issue = [&](const int32_t (&ra)[PT], const int32_t (&rb)[PT], f32x4 (&va)[MAXT][PT], f32x4 (&vb)[MAXT][PT]) {
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const char *pa = Ag + (uint64_t)(uint32_t)(ra[pt] - a.base_a) * lda;   // one v_mad_u64_u32 each
            const char *pb = Bg + (uint64_t)(uint32_t)(rb[pt] - a.base_b) * ldb;
#pragma unroll
            for (int t = 0; t < MAXT; ++t) {
                va[t][pt] = *reinterpret_cast<const f32x4 *>(pa + 64 * t);
                vb[t][pt] = *reinterpret_cast<const f32x4 *>(pb + 64 * t);
            }
        }
    };
#define AMAR_SPLIT_MFMA(W, X)                                                                                                   \
    _Pragma("unroll") for (int pt = 0; pt < PT; ++pt)                                                                           \
        y[m][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W, __builtin_bit_cast(bf16x8, X[pt]), y[m][pt], 0, 0, 0)
    // one iteration: consume (va, vb), start the next iteration's gathers into (na, nb), run the layers, store
    auto step = [&](uint32_t base, int32_t (&ra)[PT], int32_t (&rb)[PT], f32x4 (&va)[MAXT][PT], f32x4 (&vb)[MAXT][PT],
                    f32x4 (&na)[MAXT][PT], f32x4 (&nb)[MAXT][PT]) {
        int32_t out_row = 0;
        if (SCATTER) out_row = *reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(a.out_index) +
                                                                  (min(base + 16 * min(g, PT - 1) + col, last) << 2));
        f32x4 x[MAXT][PT];
#pragma unroll
        for (int t = 0; t < MAXT; ++t)
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                f32x4 v = va[t][pt] + vb[t][pt];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = relu_bits(v[r]);
                x[t][pt] = v;
            }
        issue(ra, rb, na, nb);                                     // next iteration's rows (the last pair's again past the end)
        load_ids(base + 2 * stride, ra, rb);
        for (int l = 0; SPLIT && l < a.n_layers; ++l) {
            // activations -> bf16 parts, in B-operand order: dwords 0,1 = tile 2s, dwords 2,3 = tile 2s + 1
            u32x4 xq[3][KS][PT];
#pragma unroll
            for (int sk = 0; sk < KS; ++sk)
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) {
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        const int t = 2 * sk + hf;
                        if (t < MAXT) {
                            const Split3 q0 = split_pair(x[t][pt][0], x[t][pt][1]), q1 = split_pair(x[t][pt][2], x[t][pt][3]);
                            xq[0][sk][pt][2 * hf] = q0.h; xq[1][sk][pt][2 * hf] = q0.m; xq[2][sk][pt][2 * hf] = q0.l;
                            xq[0][sk][pt][2 * hf + 1] = q1.h; xq[1][sk][pt][2 * hf + 1] = q1.m; xq[2][sk][pt][2 * hf + 1] = q1.l;
                        } else {
                            xq[0][sk][pt][2] = (BIASK && g == 0) ? 0x00003f80u : 0u;       // bf16 1.0 at feature 16 MAXT
                            xq[0][sk][pt][3] = 0u;
                            xq[1][sk][pt][2] = xq[1][sk][pt][3] = xq[2][sk][pt][2] = xq[2][sk][pt][3] = 0u;
                        }
                    }
                }
            const u32x4 *fl = frag + (size_t)l * MAXT * KS * 3 * 64 + lane;
            const float *bl = w_lds + a.b_off[l];
            f32x4 y[MAXT][PT];
#pragma unroll
            for (int m = 0; m < MAXT; ++m) {
                f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
                if (!BIASK) b4 = *reinterpret_cast<const f32x4 *>(bl + 16 * m + 4 * g);
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) y[m][pt] = b4;
#pragma unroll
                for (int sk = 0; sk < KS; ++sk) {
                    const bf16x8 wh = __builtin_bit_cast(bf16x8, fl[((m * KS + sk) * 3 + 0) * 64]);
                    const bf16x8 wm = __builtin_bit_cast(bf16x8, fl[((m * KS + sk) * 3 + 1) * 64]);
                    const bf16x8 wo = __builtin_bit_cast(bf16x8, fl[((m * KS + sk) * 3 + 2) * 64]);
                    AMAR_SPLIT_MFMA(wo, xq[0][sk]);
                    AMAR_SPLIT_MFMA(wh, xq[2][sk]);
                    AMAR_SPLIT_MFMA(wm, xq[1][sk]);
                    AMAR_SPLIT_MFMA(wm, xq[0][sk]);
                    AMAR_SPLIT_MFMA(wh, xq[1][sk]);
                    AMAR_SPLIT_MFMA(wh, xq[0][sk]);
                }
            }
#pragma unroll
            for (int m = 0; m < MAXT; ++m)
#pragma unroll
                for (int pt = 0; pt < PT; ++pt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[m][pt][r] = relu_bits(y[m][pt][r]);
        }
        for (int l = 0; !SPLIT && l < a.n_layers; ++l) {
            const float *wl = w_lds + a.w_off[l] + lane * 4;
            const float *bl = w_lds + a.b_off[l];
            f32x4 y[MAXT][PT];
            // the weight fragment of tile (m, t) + 1 is requested BEFORE the eight MFMAs of tile (m, t) are issued (left to itself the
            // compiler reads a fragment right before its own MFMAs: the LDS round trip sat between every two MFMA groups of a wave)
            f32x4 wnext = *reinterpret_cast<const f32x4 *>(wl);
#pragma unroll
            for (int m = 0; m < MAXT; ++m) {
                const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bl + 16 * m + 4 * g);
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) y[m][pt] = b4;
#pragma unroll
                for (int t = 0; t < MAXT; ++t) {
                    const f32x4 w4 = wnext;
                    if (m * MAXT + t + 1 < MAXT * MAXT) {
                        wnext = *reinterpret_cast<const f32x4 *>(wl + (m * MAXT + t + 1) * 256);
                        __builtin_amdgcn_sched_barrier(0);                  // keep the read above this tile's MFMAs
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int pt = 0; pt < PT; ++pt)
                            y[m][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[r], x[t][pt][r], y[m][pt], 0, 0, 0);
                }
            }
#pragma unroll
            for (int m = 0; m < MAXT; ++m)
#pragma unroll
                for (int pt = 0; pt < PT; ++pt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[m][pt][r] = relu_bits(y[m][pt][r]);
        }
        if (a.has_dot) {
            const float *wd = w_lds + a.dot_off;
            const float bd = w_lds[a.dot_bias_off];
            f32x4 w4[MAXT];
#pragma unroll
            for (int t = 0; t < MAXT; ++t) w4[t] = *reinterpret_cast<const f32x4 *>(wd + 16 * t + 4 * g);
            float s[PT];
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                float acc = 0.f;
#pragma unroll
                for (int t = 0; t < MAXT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc = fmaf(x[t][pt][r], w4[t][r], acc);
                acc += __shfl_xor(acc, 16, 64);
                acc += __shfl_xor(acc, 32, 64);                    // every lane of a column now holds the pair's sum
                s[pt] = acc;
            }
            // lane group g finishes pair tile g: one activation pass for all PT tiles
            float z = s[0];
#pragma unroll
            for (int pt = 1; pt < PT; ++pt) z = g == pt ? s[pt] : z;
            z += bd;
            z = chain_act(z, a.dot_act);                            // same expression as the generic kernel: bit-identical scores
            const uint32_t p = base + 16 * g + col;
            if (g < PT && p < P) a.out[(SCATTER ? (int64_t)out_row : (int64_t)p) * a.ldo] = z;
        } else {
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                const uint32_t p = base + 16 * pt + col;
#pragma unroll
                for (int m = 0; m < MAXT; ++m) {
                    const int f = 16 * m + 4 * g;
                    if (p < P && f < a.n_out) *reinterpret_cast<f32x4 *>(a.out + chain_out_row(a, p) * a.ldo + f) = x[m][pt];
                }
            }
        }
    };
#undef AMAR_SPLIT_MFMA

    uint32_t base = wave0 * pairs_per_wave;
    if (base >= P) return;
    f32x4 va[MAXT][PT], vb[MAXT][PT];
    int32_t ra[PT], rb[PT];
    load_ids(base, ra, rb);
    issue(ra, rb, va, vb);
    load_ids(base + stride, ra, rb);
    for (; base < P; base += stride) step(base, ra, rb, va, vb, va, vb);   // the raw rows are consumed before the next ones are requested
}

// Two-branch form for the hybrid head (src/models/hybrid.py:72-89 with the first layers of dense3a / dense3b folded
// into the per-entity tables):  xa = act(A1[.] + B1[.]) -> branch-1 layers;  xb = act(A2[.] + B2[.]) -> branch-2 layers;
// trunk = classifier over [xa || xb].  Everything stays in registers; up to 4 tiles (64 features) per branch and
// per trunk layer.  Tile counts are runtime values <= 4 (guards fold for the common 4/4 case only at run time).
struct DualArgs {
    const float *A[2]; int64_t lda[2]; const int32_t *ida[2]; int base_a[2];
    const float *B[2]; int64_t ldb[2]; const int32_t *idb[2]; int base_b[2];
    int D, tb, in_act;                    // branch width, its tile count, activation of the summed inputs
    int n_branch; int bw_off[2][CHAIN_MAX_LAYERS], bb_off[2][CHAIN_MAX_LAYERS], b_act[CHAIN_MAX_LAYERS];
    int n_trunk, tt; int tw_off[CHAIN_MAX_LAYERS], tbias_off[CHAIN_MAX_LAYERS], t_act[CHAIN_MAX_LAYERS];
    int dot_off, dot_bias_off, dot_act;
    const float *wpack; int wpack_floats;
    float *out; int64_t ldo; int64_t P;
    const int32_t *out_index;             // pair p is written to out row out_index[p] (a pair list kept in another order), or NULL
    // split-product form (dual_chain_split_kernel): LDS holds bf16 fragments [layer][m][s][part][lane] (16 bytes each) followed by an f32
    // tail with the biases and the 1-unit layer; *_frag = first fragment triple of a layer, *_tail = float offset inside the tail
    int n_frag, tail_floats;
    int bw_frag[2][CHAIN_MAX_LAYERS], bb_tail[2][CHAIN_MAX_LAYERS], tw_frag[CHAIN_MAX_LAYERS], tb_tail[CHAIN_MAX_LAYERS], dot_tail, dot_bias_tail;
};

template <int PT>
__global__ __launch_bounds__(1024) void dual_chain_kernel(const DualArgs a) {
    constexpr int T = 4;
    extern __shared__ __attribute__((aligned(16))) float w_lds[];
    for (int i = threadIdx.x * 4; i < a.wpack_floats; i += blockDim.x * 4)
        *reinterpret_cast<float4 *>(&w_lds[i]) = *reinterpret_cast<const float4 *>(a.wpack + i);
    __syncthreads();
    const int lane = threadIdx.x & 63, g = lane >> 4, col = lane & 15;
    const int64_t pairs_per_wave = 16 * PT;
    const int wpb = blockDim.x >> 6;                      // 16 waves share one LDS copy of the weights
    const int64_t wave0 = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6);
    const int64_t stride = (int64_t)gridDim.x * wpb * pairs_per_wave;
    const int TB = a.tb, TT = a.tt;

    for (int64_t base = wave0 * pairs_per_wave; base < a.P; base += stride) {
        f32x4 xb[2][T][PT];                               // branch activations
#pragma unroll
        for (int br = 0; br < 2; ++br) {
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                const int64_t p = base + 16 * pt + col;
                const bool ok = p < a.P;
                const int64_t ra = ok ? (a.ida[br] ? (int64_t)a.ida[br][p] - a.base_a[br] : p) : 0;
                const int64_t rb = ok ? (a.idb[br] ? (int64_t)a.idb[br][p] - a.base_b[br] : p) : 0;
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const int f = 16 * t + 4 * g;
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (t < TB && ok && f < a.D) {
                        const f32x4 va = *reinterpret_cast<const f32x4 *>(a.A[br] + ra * a.lda[br] + f);
                        const f32x4 vb = *reinterpret_cast<const f32x4 *>(a.B[br] + rb * a.ldb[br] + f);
                        v = va + vb;
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = chain_act(v[r], a.in_act);
                    }
                    xb[br][t][pt] = v;
                }
            }
            for (int l = 0; l < a.n_branch; ++l) {
                const float *wl = w_lds + a.bw_off[br][l], *bl = w_lds + a.bb_off[br][l];
                f32x4 y[T][PT];
#pragma unroll
                for (int m = 0; m < T; ++m) {
                    if (m < TB) {
                        const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bl + 16 * m + 4 * g);
#pragma unroll
                        for (int pt = 0; pt < PT; ++pt) y[m][pt] = b4;
#pragma unroll
                        for (int t = 0; t < T; ++t)
                            if (t < TB) {
                                const f32x4 w4 = *reinterpret_cast<const f32x4 *>(wl + ((m * TB + t) * 64 + lane) * 4);
#pragma unroll
                                for (int r = 0; r < 4; ++r)
#pragma unroll
                                    for (int pt = 0; pt < PT; ++pt)
                                        y[m][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[r], xb[br][t][pt][r], y[m][pt], 0, 0, 0);
                            }
                    }
                }
#pragma unroll
                for (int m = 0; m < T; ++m)
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) {
                        f32x4 v = {0.f, 0.f, 0.f, 0.f};
                        if (m < TB) {
                            v = y[m][pt];
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = chain_act(v[r], a.b_act[l]);
                        }
                        xb[br][m][pt] = v;
                    }
            }
        }
        // ---- trunk: first layer reads [xa || xb] (2*TB k-tiles), later layers TT x TT
        f32x4 x[T][PT];
        for (int l = 0; l < a.n_trunk; ++l) {
            const float *wl = w_lds + a.tw_off[l], *bl = w_lds + a.tbias_off[l];
            const int KT = l == 0 ? 2 * TB : TT;
            f32x4 y[T][PT];
#pragma unroll
            for (int m = 0; m < T; ++m) {
                if (m < TT) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bl + 16 * m + 4 * g);
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) y[m][pt] = b4;
#pragma unroll
                    for (int t = 0; t < 2 * T; ++t) {
                        const bool first = l == 0;
                        // k-tile t of the first trunk layer: branch 0 tiles 0..TB-1, then branch 1 tiles
                        const int br = t >= T ? 1 : 0, tl = t >= T ? t - T : t;
                        const bool active = first ? (tl < TB) : (t < TT);
                        if (active) {
                            const int kt_idx = first ? (br * TB + tl) : t;
                            const f32x4 w4 = *reinterpret_cast<const f32x4 *>(wl + ((m * KT + kt_idx) * 64 + lane) * 4);
#pragma unroll
                            for (int r = 0; r < 4; ++r)
#pragma unroll
                                for (int pt = 0; pt < PT; ++pt) {
                                    const float bv = first ? xb[br][tl][pt][r] : (t < T ? x[t < T ? t : 0][pt][r] : 0.f);
                                    y[m][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[r], bv, y[m][pt], 0, 0, 0);
                                }
                        }
                    }
                }
            }
#pragma unroll
            for (int m = 0; m < T; ++m)
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) {
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (m < TT) {
                        v = y[m][pt];
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = chain_act(v[r], a.t_act[l]);
                    }
                    x[m][pt] = v;
                }
        }
        const float *wd = w_lds + a.dot_off;
        const float bd = w_lds[a.dot_bias_off];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            float sacc = 0.f;
#pragma unroll
            for (int t = 0; t < T; ++t)
                if (t < TT) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4 *>(wd + 16 * t + 4 * g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) sacc = fmaf(x[t][pt][r], w4[r], sacc);
                }
            sacc += __shfl_xor(sacc, 16, 64);
            sacc += __shfl_xor(sacc, 32, 64);
            const int64_t p = base + 16 * pt + col;
            if (g == 0 && p < a.P) a.out[(a.out_index ? (int64_t)a.out_index[p] : p) * a.ldo] = chain_act(sacc + bd, a.dot_act);
        }
    }
}

// The hybrid head's common shape — 64-wide branches and trunk (4 tiles everywhere), ids on all four tables, ReLU
// throughout (econfigs/hybrid-gnn*.yaml) — without tile guards or run-time activation selection, and with ALL sixteen
// row gathers of an iteration (2 branches x 4 tiles x {A, B}) issued back to back before anything waits on them.  The
// generic kernel above guards every tile load with the exec mask, which keeps the compiler from batching them: a wave
// then pays eight memory round trips (~2 us each, the tables live in the Infinity Cache) per 16 pairs against 4.4 us of
// MFMA work — with four waves per SIMD that alone explains its 63 % MFMA utilisation.  Same arithmetic in the same order:
// bit-identical scores.
__global__ __launch_bounds__(1024) void dual_chain_full_kernel(const DualArgs a) {
    constexpr int T = 4;
    extern __shared__ __attribute__((aligned(16))) float w_lds[];
    for (int i = threadIdx.x * 4; i < a.wpack_floats; i += blockDim.x * 4)
        *reinterpret_cast<float4 *>(&w_lds[i]) = *reinterpret_cast<const float4 *>(a.wpack + i);
    __syncthreads();
    const int lane = threadIdx.x & 63, g = lane >> 4, col = lane & 15;
    const int wpb = blockDim.x >> 6;
    const int64_t wave0 = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6);
    const int64_t stride = (int64_t)gridDim.x * wpb * 16;

    for (int64_t base = wave0 * 16; base < a.P; base += stride) {
        const int64_t p = base + col;
        const bool ok = p < a.P;                               // pairs past the end read row 0 and are never stored
        f32x4 va[2][T], vb[2][T];
#pragma unroll
        for (int br = 0; br < 2; ++br) {
            const uint32_t ra = ok ? (uint32_t)(a.ida[br][p] - a.base_a[br]) : 0u;
            const uint32_t rb = ok ? (uint32_t)(a.idb[br][p] - a.base_b[br]) : 0u;
            const float *pa = a.A[br] + (uint64_t)ra * (uint32_t)a.lda[br] + 4 * g;
            const float *pb = a.B[br] + (uint64_t)rb * (uint32_t)a.ldb[br] + 4 * g;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                va[br][t] = *reinterpret_cast<const f32x4 *>(pa + 16 * t);
                vb[br][t] = *reinterpret_cast<const f32x4 *>(pb + 16 * t);
            }
        }
        f32x4 xb[2][T];
#pragma unroll
        for (int br = 0; br < 2; ++br) {
#pragma unroll
            for (int t = 0; t < T; ++t) {
                f32x4 v = va[br][t] + vb[br][t];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = relu_bits(v[r]);
                xb[br][t] = v;
            }
            for (int l = 0; l < a.n_branch; ++l) {
                const float *wl = w_lds + a.bw_off[br][l], *bl = w_lds + a.bb_off[br][l];
                f32x4 y[T];
#pragma unroll
                for (int m = 0; m < T; ++m) {
                    y[m] = *reinterpret_cast<const f32x4 *>(bl + 16 * m + 4 * g);
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        const f32x4 w4 = *reinterpret_cast<const f32x4 *>(wl + ((m * T + t) * 64 + lane) * 4);
#pragma unroll
                        for (int r = 0; r < 4; ++r) y[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[r], xb[br][t][r], y[m], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int m = 0; m < T; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) xb[br][m][r] = relu_bits(y[m][r]);
            }
        }
        // ---- trunk: the first layer reads [xa || xb] (8 k-tiles), later layers 4 x 4
        f32x4 x[T];
        for (int l = 0; l < a.n_trunk; ++l) {
            const float *wl = w_lds + a.tw_off[l], *bl = w_lds + a.tbias_off[l];
            f32x4 y[T];
            if (l == 0) {
#pragma unroll
                for (int m = 0; m < T; ++m) {
                    y[m] = *reinterpret_cast<const f32x4 *>(bl + 16 * m + 4 * g);
#pragma unroll
                    for (int t = 0; t < 2 * T; ++t) {
                        const f32x4 w4 = *reinterpret_cast<const f32x4 *>(wl + ((m * 2 * T + t) * 64 + lane) * 4);
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            y[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[r], xb[t / T][t % T][r], y[m], 0, 0, 0);
                    }
                }
            } else {
#pragma unroll
                for (int m = 0; m < T; ++m) {
                    y[m] = *reinterpret_cast<const f32x4 *>(bl + 16 * m + 4 * g);
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        const f32x4 w4 = *reinterpret_cast<const f32x4 *>(wl + ((m * T + t) * 64 + lane) * 4);
#pragma unroll
                        for (int r = 0; r < 4; ++r) y[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[r], x[t][r], y[m], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int m = 0; m < T; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) x[m][r] = relu_bits(y[m][r]);
        }
        const float *wd = w_lds + a.dot_off;
        float sacc = 0.f;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const f32x4 w4 = *reinterpret_cast<const f32x4 *>(wd + 16 * t + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) sacc = fmaf(x[t][r], w4[r], sacc);
        }
        sacc += __shfl_xor(sacc, 16, 64);
        sacc += __shfl_xor(sacc, 32, 64);
        if (g == 0 && ok) a.out[(a.out_index ? (int64_t)a.out_index[p] : p) * a.ldo] = chain_act(sacc + w_lds[a.dot_bias_off], a.dot_act);
    }
}

// The same head with its products on the bf16 matrix instruction, both operands split three ways (the pair-stage kernel's SPLIT form
// above: f32-accurate, 6 MFMAs of 16 cycles per k-step of 32 instead of 8 of 32 cycles).  The f32 weight blob (84 KB for the 64-wide
// stacks) is split by the workgroup into 126 KB of bf16 fragments at its start, straight from global memory — blob and fragments
// would not fit the LDS together; biases and the 1-unit layer stay f32 in a small tail.  AMAR_PAIR_MFMA=f32 keeps the f32 kernel.
// PT pair tiles of 16 per wave and iteration, THREADS per workgroup (the 126 KB of fragments allow one workgroup per CU, so THREADS sets
// the waves per SIMD and the registers a lane may hold: 768 -> 3 waves, 168 registers, PT = 1 — the default, see the launcher).
template <int PT, int THREADS>
__global__ __launch_bounds__(THREADS) void dual_chain_split_kernel(const DualArgs a) {
    constexpr int T = 4, KS = 2;
    extern __shared__ __attribute__((aligned(16))) float w_lds[];
    u32x4 *frag = reinterpret_cast<u32x4 *>(w_lds);
    float *tail = w_lds + (size_t)a.n_frag * 3 * 64 * 4;
    const int lane = threadIdx.x & 63, g = lane >> 4, col = lane & 15;
    constexpr int wpb = THREADS / 64;
    const int wave = threadIdx.x >> 6;
    {
        // fragment triple f of a layer with KT input tiles: (m, s) = (f / (KT/2), f % (KT/2)); its two f32 tiles are 2s and 2s + 1
        auto build = [&](int w_off, int kt, int frag0) {
            const int ks = kt / 2;
            for (int f = wave; f < T * ks; f += wpb) {
                const int m = f / ks, sk = f % ks;
                const float *wl = a.wpack + w_off + ((size_t)(m * kt + 2 * sk) * 64 + lane) * 4;
                const Split3x4 q = split_tiles(*reinterpret_cast<const f32x4 *>(wl), *reinterpret_cast<const f32x4 *>(wl + 256));
                u32x4 *dst = frag + (size_t)(frag0 + f) * 3 * 64 + lane;
                dst[0] = q.h; dst[64] = q.m; dst[128] = q.l;
            }
        };
        for (int br = 0; br < 2; ++br)
            for (int l = 0; l < a.n_branch; ++l) build(a.bw_off[br][l], T, a.bw_frag[br][l]);
        for (int l = 0; l < a.n_trunk; ++l) build(a.tw_off[l], l == 0 ? 2 * T : T, a.tw_frag[l]);
        for (int i = threadIdx.x; i < 16 * T; i += THREADS) {
            for (int br = 0; br < 2; ++br)
                for (int l = 0; l < a.n_branch; ++l) tail[a.bb_tail[br][l] + i] = a.wpack[a.bb_off[br][l] + i];
            for (int l = 0; l < a.n_trunk; ++l) tail[a.tb_tail[l] + i] = a.wpack[a.tbias_off[l] + i];
            tail[a.dot_tail + i] = a.wpack[a.dot_off + i];
        }
        if (threadIdx.x == 0) tail[a.dot_bias_tail] = a.wpack[a.dot_bias_off];
    }
    __syncthreads();
    const int64_t wave0 = (int64_t)blockIdx.x * wpb + wave;
    const int64_t stride = (int64_t)gridDim.x * wpb * 16 * PT;

    // y[m][pt] += W(m, k-step) . x[pt]: one fragment triple from the LDS, the six part products of every tile (small ones first)
    auto layer_step = [&](const u32x4 *w, const Split3x4 (&x)[PT], f32x4 (&y)[PT]) {
        const bf16x8 wh = __builtin_bit_cast(bf16x8, w[0]), wm = __builtin_bit_cast(bf16x8, w[64]), wl = __builtin_bit_cast(bf16x8, w[128]);
#define AMAR_DUAL_MFMA(W, PART)                                                                                                  \
    _Pragma("unroll") for (int pt = 0; pt < PT; ++pt)                                                                            \
        y[pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W, __builtin_bit_cast(bf16x8, x[pt].PART), y[pt], 0, 0, 0)
        AMAR_DUAL_MFMA(wl, h); AMAR_DUAL_MFMA(wh, l); AMAR_DUAL_MFMA(wm, m); AMAR_DUAL_MFMA(wm, h); AMAR_DUAL_MFMA(wh, m); AMAR_DUAL_MFMA(wh, h);
#undef AMAR_DUAL_MFMA
    };

    for (int64_t base = wave0 * 16 * PT; base < a.P; base += stride) {
        Split3x4 xq[2][KS][PT];                                // the branches' outputs, split: the trunk's first layer reads all four k-steps
#pragma unroll
        for (int br = 0; br < 2; ++br) {
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                const int64_t p = base + 16 * pt + col;
                const bool ok = p < a.P;                       // pairs past the end read row 0 and are never stored
                const uint32_t ra = ok ? (uint32_t)(a.ida[br][p] - a.base_a[br]) : 0u;
                const uint32_t rb = ok ? (uint32_t)(a.idb[br][p] - a.base_b[br]) : 0u;
                const float *pa = a.A[br] + (uint64_t)ra * (uint32_t)a.lda[br] + 4 * g;
                const float *pb = a.B[br] + (uint64_t)rb * (uint32_t)a.ldb[br] + 4 * g;
                f32x4 xb[T];
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    f32x4 v = *reinterpret_cast<const f32x4 *>(pa + 16 * t) + *reinterpret_cast<const f32x4 *>(pb + 16 * t);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = relu_bits(v[r]);
                    xb[t] = v;
                }
#pragma unroll
                for (int sk = 0; sk < KS; ++sk) xq[br][sk][pt] = split_tiles(xb[2 * sk], xb[2 * sk + 1]);
            }
            for (int l = 0; l < a.n_branch; ++l) {
                const u32x4 *fl = frag + (size_t)a.bw_frag[br][l] * 3 * 64 + lane;
                const float *bl = tail + a.bb_tail[br][l];
                f32x4 y[T][PT];
#pragma unroll
                for (int m = 0; m < T; ++m) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bl + 16 * m + 4 * g);
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) y[m][pt] = b4;
#pragma unroll
                    for (int sk = 0; sk < KS; ++sk) layer_step(fl + (m * KS + sk) * 3 * 64, xq[br][sk], y[m]);
                }
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) {
#pragma unroll
                    for (int m = 0; m < T; ++m)
#pragma unroll
                        for (int r = 0; r < 4; ++r) y[m][pt][r] = relu_bits(y[m][pt][r]);
#pragma unroll
                    for (int sk = 0; sk < KS; ++sk) xq[br][sk][pt] = split_tiles(y[2 * sk][pt], y[2 * sk + 1][pt]);
                }
            }
        }
        // ---- trunk: the first layer reads [xa || xb] (four k-steps), later layers two
        f32x4 x[T][PT];
        Split3x4 xt[KS][PT];
        for (int l = 0; l < a.n_trunk; ++l) {
            const u32x4 *fl = frag + (size_t)a.tw_frag[l] * 3 * 64 + lane;
            const float *bl = tail + a.tb_tail[l];
            f32x4 y[T][PT];
#pragma unroll
            for (int m = 0; m < T; ++m) {
                const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bl + 16 * m + 4 * g);
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) y[m][pt] = b4;
                if (l == 0) {
#pragma unroll
                    for (int sk = 0; sk < 2 * KS; ++sk) layer_step(fl + (m * 2 * KS + sk) * 3 * 64, xq[sk / KS][sk % KS], y[m]);
                } else {
#pragma unroll
                    for (int sk = 0; sk < KS; ++sk) layer_step(fl + (m * KS + sk) * 3 * 64, xt[sk], y[m]);
                }
            }
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
#pragma unroll
                for (int m = 0; m < T; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[m][pt][r] = relu_bits(y[m][pt][r]);
                if (l + 1 < a.n_trunk) {
#pragma unroll
                    for (int sk = 0; sk < KS; ++sk) xt[sk][pt] = split_tiles(x[2 * sk][pt], x[2 * sk + 1][pt]);
                }
            }
        }
        const float *wd = tail + a.dot_tail;
        f32x4 w4[T];
#pragma unroll
        for (int t = 0; t < T; ++t) w4[t] = *reinterpret_cast<const f32x4 *>(wd + 16 * t + 4 * g);
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            float sacc = 0.f;
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) sacc = fmaf(x[t][pt][r], w4[t][r], sacc);
            sacc += __shfl_xor(sacc, 16, 64);
            sacc += __shfl_xor(sacc, 32, 64);
            const int64_t p = base + 16 * pt + col;
            if (g == 0 && p < a.P) a.out[(a.out_index ? (int64_t)a.out_index[p] : p) * a.ldo] = chain_act(sacc + tail[a.dot_bias_tail], a.dot_act);
        }
    }
}

inline int tiles16(int n) { return (n + 15) / 16; }

}  // namespace

extern "C" {

// Number of floats of the packed blob for layer widths dims[0..n_layers] (dims[0] = input width).
int64_t amar_chain_pack_floats(const int32_t *dims, int32_t n_layers) {
    if (!dims || n_layers < 1 || n_layers > CHAIN_MAX_LAYERS) return AMAR_EINVAL;
    int64_t total = 0;
    for (int l = 0; l < n_layers; ++l) {
        const int K = dims[l], N = dims[l + 1];
        if (K < 1 || N < 1) return AMAR_EINVAL;
        if (l == n_layers - 1 && N == 1 && n_layers > 1) total += 16 * tiles16(K) + 4;         // dot weights + bias
        else total += (int64_t)tiles16(N) * tiles16(K) * 256 + 16 * tiles16(N);                // fragments + bias
    }
    return total;
}

// HOST-side packing: kernels[l] is row-major [dims[l], dims[l+1]] (Keras Dense kernel), biases[l] is [dims[l+1]].
// Fragment order: W_p[m][t][lane][r] = W[16t + 4(lane>>4) + r][16m + (lane&15)], zero outside the matrix.
int amar_chain_pack_f32(const float *const *kernels, const float *const *biases, const int32_t *dims,
                        int32_t n_layers, float *out) {
    if (!kernels || !biases || !out) return AMAR_EINVAL;
    const int64_t total = amar_chain_pack_floats(dims, n_layers);
    if (total < 0) return (int)total;
    int64_t off = 0;
    for (int l = 0; l < n_layers; ++l) {
        const int K = dims[l], N = dims[l + 1], KT = tiles16(K), NT = tiles16(N);
        const float *W = kernels[l], *b = biases[l];
        if (l == n_layers - 1 && N == 1 && n_layers > 1) {
            for (int k = 0; k < 16 * KT; ++k) out[off + k] = k < K ? W[k] : 0.f;
            off += 16 * KT;
            out[off] = b[0]; out[off + 1] = out[off + 2] = out[off + 3] = 0.f;
            off += 4;
        } else {
            for (int m = 0; m < NT; ++m)
                for (int t = 0; t < KT; ++t)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int r = 0; r < 4; ++r) {
                            const int k = 16 * t + 4 * (lane >> 4) + r, n = 16 * m + (lane & 15);
                            out[off++] = (k < K && n < N) ? W[(int64_t)k * N + n] : 0.f;
                        }
            for (int n = 0; n < 16 * NT; ++n) out[off++] = n < N ? b[n] : 0.f;
        }
    }
    return off == total ? AMAR_OK : AMAR_EINVAL;
}

int amar_chain_f32(const float *A, int64_t lda, int32_t Da, const int32_t *ids_a, int32_t base_a,
                   const float *B, int64_t ldb, int32_t Db, const int32_t *ids_b, int32_t base_b,
                   int32_t sum_inputs, int32_t in_act,
                   const float *wpack, const int32_t *dims, const int32_t *acts, int32_t n_layers,
                   float *out, int64_t ldo, int64_t P, amar_stream_t stream) {
    return amar_chain_indexed_f32(A, lda, Da, ids_a, base_a, B, ldb, Db, ids_b, base_b, sum_inputs, in_act, wpack, dims, acts, n_layers,
                                  out, ldo, nullptr, P, stream);
}

static int chain_run(const float *A, int64_t lda, int32_t Da, const int32_t *ids_a, int32_t base_a,
                     const float *B, int64_t ldb, int32_t Db, const int32_t *ids_b, int32_t base_b,
                     int32_t sum_inputs, int32_t in_act,
                     const float *wpack, const int32_t *dims, const int32_t *acts, int32_t n_layers,
                     float *out, int64_t ldo, const int32_t *out_index, int64_t P, amar_stream_t stream,
                     int32_t n_seg, const float *const *seg, const int64_t *seg_ld, const int32_t *seg_width);

int amar_chain_indexed_f32(const float *A, int64_t lda, int32_t Da, const int32_t *ids_a, int32_t base_a,
                           const float *B, int64_t ldb, int32_t Db, const int32_t *ids_b, int32_t base_b,
                           int32_t sum_inputs, int32_t in_act,
                           const float *wpack, const int32_t *dims, const int32_t *acts, int32_t n_layers,
                           float *out, int64_t ldo, const int32_t *out_index, int64_t P, amar_stream_t stream) {
    return chain_run(A, lda, Da, ids_a, base_a, B, ldb, Db, ids_b, base_b, sum_inputs, in_act, wpack, dims, acts, n_layers, out, ldo, out_index, P,
                     stream, 0, nullptr, nullptr, nullptr);
}

int amar_chain_segments_f32(const float *const *seg, const int64_t *seg_ld, const int32_t *seg_width, int32_t n_seg,
                            const int32_t *ids, int32_t base,
                            const float *wpack, const int32_t *dims, const int32_t *acts, int32_t n_layers,
                            float *out, int64_t ldo, int64_t P, amar_stream_t stream) {
    if (!seg || !seg_ld || !seg_width || n_seg < 1 || n_seg > CHAIN_MAX_SEG) return AMAR_EINVAL;
    int64_t da = 0;
    for (int j = 0; j < n_seg; ++j) {
        if (!seg[j] || seg_width[j] < 4 || (seg_width[j] & 3) || seg_ld[j] < seg_width[j] || (seg_ld[j] & 3) || !amar_aligned16(seg[j]) ||
            seg_ld[j] >= (1ll << 30))
            return AMAR_EINVAL;
        da += seg_width[j];
    }
    if (da > 128) return AMAR_EUNSUPPORTED;
    return chain_run(seg[0], seg_ld[0], (int32_t)da, ids, base, nullptr, 0, 0, nullptr, 0, 0, AMAR_ACT_NONE, wpack, dims, acts, n_layers, out, ldo,
                     nullptr, P, stream, n_seg, seg, seg_ld, seg_width);
}

static int chain_run(const float *A, int64_t lda, int32_t Da, const int32_t *ids_a, int32_t base_a,
                     const float *B, int64_t ldb, int32_t Db, const int32_t *ids_b, int32_t base_b,
                     int32_t sum_inputs, int32_t in_act,
                     const float *wpack, const int32_t *dims, const int32_t *acts, int32_t n_layers,
                     float *out, int64_t ldo, const int32_t *out_index, int64_t P, amar_stream_t stream,
                     int32_t n_seg, const float *const *seg, const int64_t *seg_ld, const int32_t *seg_width) {
    if (P < 0 || !A || !wpack || !dims || !acts || !out || Da < 4 || Db < 0) return AMAR_EINVAL;
    if (sum_inputs && (Db != Da || !B)) return AMAR_EINVAL;
    if (in_act != AMAR_ACT_NONE && in_act != AMAR_ACT_RELU && in_act != AMAR_ACT_SIGMOID) return AMAR_EINVAL;
    if ((Da & 3) || (Db & 3) || (lda & 3) || (n_seg == 0 && lda < Da) || !amar_aligned16(A) || !amar_aligned16(wpack)) return AMAR_EINVAL;
    if (Db && (!B || (ldb & 3) || ldb < Db || !amar_aligned16(B))) return AMAR_EINVAL;
    if (n_layers < 1 || n_layers > CHAIN_MAX_LAYERS || dims[0] != (sum_inputs ? Da : Da + Db)) return AMAR_EINVAL;
    ChainArgs a{};
    a.A = A; a.lda = lda; a.Da = Da; a.ids_a = ids_a; a.base_a = base_a;
    a.B = B; a.ldb = ldb; a.Db = Db; a.ids_b = ids_b; a.base_b = base_b;
    a.wpack = wpack; a.out = out; a.ldo = ldo; a.P = P; a.sum_inputs = sum_inputs ? 1 : 0; a.in_act = in_act;
    a.out_index = out_index;
    a.n_seg = n_seg;
    for (int j = 0; j < n_seg; ++j) {
        a.seg[j] = seg[j]; a.seg_ld[j] = seg_ld[j];
        a.seg_off[j + 1] = a.seg_off[j] + seg_width[j];
    }
    int maxw = 0, off = 0;
    for (int l = 0; l < n_layers; ++l) {
        const int K = dims[l], N = dims[l + 1], act = acts[l];
        if (K < 1 || N < 1 || (act != AMAR_ACT_NONE && act != AMAR_ACT_RELU && act != AMAR_ACT_SIGMOID)) return AMAR_EINVAL;
        maxw = K > maxw ? K : maxw;
        if (l == n_layers - 1 && N == 1 && n_layers > 1) {
            a.has_dot = 1; a.dot_off = off; a.dot_kt = tiles16(K); a.dot_act = act;
            off += 16 * tiles16(K);
            a.dot_bias_off = off;
            off += 4;
        } else {
            maxw = N > maxw ? N : maxw;
            a.kt[a.n_layers] = tiles16(K); a.nt[a.n_layers] = tiles16(N); a.act[a.n_layers] = act;
            a.w_off[a.n_layers] = off;
            off += tiles16(N) * tiles16(K) * 256;
            a.b_off[a.n_layers] = off;
            off += 16 * tiles16(N);
            a.n_layers++;
            a.n_out = N;
        }
    }
    a.wpack_floats = off;
    if (a.n_layers < 1) return AMAR_EINVAL;
    if (!a.has_dot && ((ldo & 3) || ldo < a.n_out || (a.n_out & 3) || !amar_aligned16(out))) return AMAR_EINVAL;
    if (a.has_dot && ldo < 1) return AMAR_EINVAL;
    if (maxw > 128 || (size_t)off * sizeof(float) > 150 * 1024) return AMAR_EUNSUPPORTED;
    if (P == 0) return AMAR_OK;
    const size_t lds_bytes = (size_t)off * sizeof(float);
    hipStream_t st = static_cast<hipStream_t>(stream);
    // tile budget: MAXT = widest layer / 16 rounded up to {3, 4, 8}; PT pair tiles per wave (register budget ~ MAXT * PT)
    static const int force_pt = getenv("AMAR_CHAIN_PT") ? atoi(getenv("AMAR_CHAIN_PT")) : 0;
    const int maxt = maxw <= 48 ? 3 : (maxw <= 64 ? 4 : 8);
    bool full = (dims[0] + 15) / 16 == maxt && (!a.has_dot || a.dot_kt == maxt);
    for (int l = 0; l < a.n_layers; ++l) full = full && a.kt[l] == maxt && a.nt[l] == maxt;
    bool relu = !a.sum_inputs || a.in_act == AMAR_ACT_RELU, relu_but_last = relu && !a.has_dot;
    for (int l = 0; l < a.n_layers; ++l) {
        relu = relu && a.act[l] == AMAR_ACT_RELU;
        relu_but_last = relu_but_last && a.act[l] == (l == a.n_layers - 1 ? AMAR_ACT_NONE : AMAR_ACT_RELU);
    }
    static const bool no_am2 = getenv("AMAR_CHAIN_AM2") && atoi(getenv("AMAR_CHAIN_AM2")) == 0;   // A/B: run-time activations for the towers
    const int am = relu ? 1 : (relu_but_last && !no_am2 ? 2 : 0);
    int pt = 2;                                                   // measured best on grid1/grid2/grid6 shapes (tools/exp_chain.py)
    if (force_pt == 1 || force_pt == 2 || (force_pt == 4 && maxt != 8)) pt = force_pt;
    static const int gen_cap = getenv("AMAR_CHAIN_GRID") ? atoi(getenv("AMAR_CHAIN_GRID")) : 4096;
#define AMAR_CHAIN_LAUNCH(MT, PTT)                                                                                      \
    do {                                                                                                                \
        int64_t blocks = (P + 4 * 16 * PTT - 1) / (4 * 16 * PTT);                                                       \
        if (blocks > gen_cap) blocks = gen_cap;                                                                         \
        auto kern = full ? (am == 1 ? chain_kernel<MT, PTT, true, 1> : am == 2 ? chain_kernel<MT, PTT, true, 2>         \
                                                                               : chain_kernel<MT, PTT, true, 0>)        \
                         : (am == 1 ? chain_kernel<MT, PTT, false, 1> : am == 2 ? chain_kernel<MT, PTT, false, 2>       \
                                                                                : chain_kernel<MT, PTT, false, 0>);     \
        if (lds_bytes > 64 * 1024 &&                                                                                    \
            hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,       \
                                (int)lds_bytes) != hipSuccess)                                                          \
            return AMAR_ELAUNCH;                                                                                        \
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds_bytes, st, a);                                  \
    } while (0)
    // pair stage (ReLU of the sum of two gathered rows, square ReLU layers): the pipelined kernel; AMAR_CHAIN_PIPE=0 keeps the generic one
    static const bool no_pipe = getenv("AMAR_CHAIN_PIPE") && atoi(getenv("AMAR_CHAIN_PIPE")) == 0;
    const bool a_dot_ok = !a.has_dot || a.dot_kt == maxt;
    const bool small_tables = lda < (1ll << 30) && ldb < (1ll << 30) && P < (1ll << 30) - (4ll << 20);   // 32-bit row bytes and positions (+ 3 strides of <= 8 192 workgroups)
    if (!no_pipe && relu && a.sum_inputs && a.in_act == AMAR_ACT_RELU && full && a_dot_ok && a.ids_a && a.ids_b && a.Da == 16 * maxt &&
        a.Db == a.Da && maxt <= 4 && pt == 2 && lds_bytes <= 64 * 1024 && small_tables) {
        int64_t blocks = (P + 4 * 16 * 2 - 1) / (4 * 16 * 2);
        // 1 536 = 256 CUs x 3 resident workgroups x 2: whole rounds of workgroups, no tail (ml1m(s=64): 0.639 ms against 0.647 at 4 096,
        // 0.658 at 8 192; AMAR_CHAIN_BLOCKS overrides — keep it a multiple of 8: PairPlan's XCD affinity)
        static const int cap = getenv("AMAR_CHAIN_BLOCKS") ? atoi(getenv("AMAR_CHAIN_BLOCKS")) : 1536;
        if (blocks > cap) blocks = cap;
        if (blocks > 8192) blocks = 8192;
        const dim3 grid((unsigned)blocks), block(256);
        if (a.out_index && !a.has_dot) return AMAR_EUNSUPPORTED;
        // products on the bf16 matrix instruction with three-way split operands (see the kernel's header); AMAR_PAIR_MFMA=f32: the f32 one
        static const bool f32_only = getenv("AMAR_PAIR_MFMA") && !strcmp(getenv("AMAR_PAIR_MFMA"), "f32");
        const size_t split_bytes = (((size_t)off + 3) & ~(size_t)3) * sizeof(float) + (size_t)a.n_layers * maxt * ((maxt + 1) / 2) * 3 * 1024;
        const bool split = !f32_only && split_bytes <= 64 * 1024;
        static const size_t lds_pad = getenv("AMAR_CHAIN_LDS_PAD") ? (size_t)atoi(getenv("AMAR_CHAIN_LDS_PAD")) : 0;   // dev: fewer resident workgroups
        const size_t lds = (split ? split_bytes : lds_bytes) + ((split ? split_bytes : lds_bytes) + lds_pad <= 64 * 1024 ? lds_pad : 0);
#define AMAR_PIPE_LAUNCH(MT, SC)                                                                                          \
        do {                                                                                                              \
            if (split) hipLaunchKernelGGL((chain_pipe_kernel<MT, 2, SC, true>), grid, block, lds, st, a);                 \
            else hipLaunchKernelGGL((chain_pipe_kernel<MT, 2, SC, false>), grid, block, lds, st, a);                      \
        } while (0)
        if (a.out_index) {
            if (maxt == 3) AMAR_PIPE_LAUNCH(3, true); else AMAR_PIPE_LAUNCH(4, true);
        } else {
            if (maxt == 3) AMAR_PIPE_LAUNCH(3, false); else AMAR_PIPE_LAUNCH(4, false);
        }
#undef AMAR_PIPE_LAUNCH
        return amar_check_launch();
    }
    // entity towers (one table, no dot, ReLU with an optionally linear last layer) in a shape with a compile-time kernel; AMAR_CHAIN_ROWS=0
    // keeps the generic one
    static const bool no_rows = getenv("AMAR_CHAIN_ROWS") && atoi(getenv("AMAR_CHAIN_ROWS")) == 0;
    if ((!no_rows || n_seg > 0) && (am == 1 || am == 2) && !a.sum_inputs && a.Db == 0 && !a.has_dot && !a.out_index && (pt == 2 || n_seg > 0) && a.n_layers <= 3 &&
        maxt <= 4 && lds_bytes <= 64 * 1024 && lda < (1ll << 30) && P < (1ll << 30) - (4ll << 20)) {
        const int shape = chain_shape(a.n_layers, a.kt[0], a.nt[0], a.n_layers > 1 ? a.nt[1] : 0, a.n_layers > 2 ? a.nt[2] : 0);
        // 1 024 workgroups = 4 per CU, every wave a few iterations deep in its prefetch (ml1m(s=64) towers: 0.059 ms against 0.063 at
        // 4 096 and 0.068 for the generic kernel; AMAR_CHAIN_GRID overrides)
        int64_t blocks = (P + 4 * 16 * 2 - 1) / (4 * 16 * 2);
        const int64_t rows_cap = getenv("AMAR_CHAIN_GRID") ? gen_cap : 1024;
        if (blocks > rows_cap) blocks = rows_cap;
        if (blocks > 8192) blocks = 8192;
        const dim3 grid((unsigned)blocks), block(256);
        bool done = true;
#define AMAR_ROWS_CASE(NLL, A0, A1, A2, A3)                                                                                      \
        case chain_shape(NLL, A0, A1, A2, A3):                                                                                   \
            if (n_seg > 0) {                                                                                                     \
                if (am == 2) hipLaunchKernelGGL((chain_rows_kernel<chain_shape(NLL, A0, A1, A2, A3), 2, true, true>), grid, block, lds_bytes, st, a);   \
                else hipLaunchKernelGGL((chain_rows_kernel<chain_shape(NLL, A0, A1, A2, A3), 2, false, true>), grid, block, lds_bytes, st, a);          \
            } else if (am == 2) hipLaunchKernelGGL((chain_rows_kernel<chain_shape(NLL, A0, A1, A2, A3), 2, true, false>), grid, block, lds_bytes, st, a);   \
            else hipLaunchKernelGGL((chain_rows_kernel<chain_shape(NLL, A0, A1, A2, A3), 2, false, false>), grid, block, lds_bytes, st, a);          \
            break
        switch (shape) {
        AMAR_ROWS_CASE(3, 2, 2, 2, 3);      // 24 -> 24 -> 24 -> 48: basic-gnn grid1's towers with the classifier's first layer folded in
        AMAR_ROWS_CASE(3, 3, 3, 3, 4);      // 48 -> 48 -> 48 -> 64: grid2
        AMAR_ROWS_CASE(2, 2, 2, 2, 0);      // 24 -> 24 -> 24: the graph towers of the hybrid head
        AMAR_ROWS_CASE(3, 1, 2, 2, 3);      //  8 -> 24 -> 24 -> 48: grid1's towers after a 'mean' / 'sum' / 'w-sum' reduction (LightGCN, DGCF)
        AMAR_ROWS_CASE(3, 1, 3, 3, 4);      // 16 -> 48 -> 48 -> 64: grid2's
        AMAR_ROWS_CASE(2, 3, 3, 3, 0);      // 48 -> 48 -> 48
        default: done = false;
        }
#undef AMAR_ROWS_CASE
        if (done) return amar_check_launch();
    }
    if (n_seg > 0) return AMAR_EUNSUPPORTED;       // segments are read by the compile-time tower shapes only: the caller concatenates for the others
    if (maxt == 3) { if (pt == 1) AMAR_CHAIN_LAUNCH(3, 1); else if (pt == 2) AMAR_CHAIN_LAUNCH(3, 2); else AMAR_CHAIN_LAUNCH(3, 4); }
    else if (maxt == 4) { if (pt == 1) AMAR_CHAIN_LAUNCH(4, 1); else if (pt == 2) AMAR_CHAIN_LAUNCH(4, 2); else AMAR_CHAIN_LAUNCH(4, 4); }
    else { if (pt == 1) AMAR_CHAIN_LAUNCH(8, 1); else AMAR_CHAIN_LAUNCH(8, 2); }
#undef AMAR_CHAIN_LAUNCH
    return amar_check_launch();
}

// Fused two-branch head.  Each branch b in {0,1}: x_b = in_act(A_b[ida_b] + B_b[idb_b]) ([P, D]), then n_branch Dense
// layers D -> D (packed with amar_chain_pack_f32, dims [D, D, ...], one blob per branch); trunk: Dense stack over
// [x_0 || x_1] with dims trunk_dims[0] = 2D, equal hidden widths <= 64 and a final 1-unit layer (packed the same way).
// D and the trunk widths must be multiples of 4 and <= 64.  wpack = branch-0 blob, branch-1 blob, trunk blob, contiguous.
int amar_dual_chain_f32(const float *const *A, const int64_t *lda, const int32_t *const *ida, const int32_t *base_a,
                        const float *const *B, const int64_t *ldb, const int32_t *const *idb, const int32_t *base_b,
                        int32_t D, int32_t in_act, int32_t n_branch, const int32_t *branch_acts,
                        const int32_t *trunk_dims, const int32_t *trunk_acts, int32_t n_trunk,
                        const float *wpack, float *out, int64_t ldo, int64_t P, amar_stream_t stream) {
    return amar_dual_chain_indexed_f32(A, lda, ida, base_a, B, ldb, idb, base_b, D, in_act, n_branch, branch_acts, trunk_dims, trunk_acts,
                                       n_trunk, wpack, out, ldo, nullptr, P, stream);
}

// The same head on a pair list kept in another order (models/basic.py:PairPlan): pair p is written to out[out_index[p] * ldo].
int amar_dual_chain_indexed_f32(const float *const *A, const int64_t *lda, const int32_t *const *ida, const int32_t *base_a,
                                const float *const *B, const int64_t *ldb, const int32_t *const *idb, const int32_t *base_b,
                                int32_t D, int32_t in_act, int32_t n_branch, const int32_t *branch_acts,
                                const int32_t *trunk_dims, const int32_t *trunk_acts, int32_t n_trunk,
                                const float *wpack, float *out, int64_t ldo, const int32_t *out_index, int64_t P, amar_stream_t stream) {
    if (P < 0 || !A || !B || !lda || !ldb || !ida || !idb || !base_a || !base_b || !wpack || !out || !trunk_dims || !trunk_acts)
        return AMAR_EINVAL;
    if (D < 4 || (D & 3) || n_branch < 0 || n_branch > CHAIN_MAX_LAYERS || (n_branch && !branch_acts)) return AMAR_EINVAL;
    if (n_trunk < 2 || n_trunk > CHAIN_MAX_LAYERS || trunk_dims[0] != 2 * D || trunk_dims[n_trunk] != 1 || ldo < 1) return AMAR_EINVAL;
    if (D > 64) return AMAR_EUNSUPPORTED;
    DualArgs a{};
    for (int b = 0; b < 2; ++b) {
        if (!A[b] || !B[b] || lda[b] < D || ldb[b] < D || (lda[b] & 3) || (ldb[b] & 3) || !amar_aligned16(A[b]) || !amar_aligned16(B[b]))
            return AMAR_EINVAL;
        a.A[b] = A[b]; a.lda[b] = lda[b]; a.ida[b] = ida[b]; a.base_a[b] = base_a[b];
        a.B[b] = B[b]; a.ldb[b] = ldb[b]; a.idb[b] = idb[b]; a.base_b[b] = base_b[b];
    }
    a.D = D; a.tb = tiles16(D); a.in_act = in_act; a.n_branch = n_branch;
    int off = 0;
    for (int b = 0; b < 2; ++b)
        for (int l = 0; l < n_branch; ++l) {
            a.bw_off[b][l] = off; off += a.tb * a.tb * 256;
            a.bb_off[b][l] = off; off += 16 * a.tb;
            a.b_act[l] = branch_acts[l];
        }
    const int W = trunk_dims[1];
    if (W < 4 || (W & 3) || W > 64) return AMAR_EUNSUPPORTED;
    a.tt = tiles16(W); a.n_trunk = n_trunk - 1;
    for (int l = 0; l < n_trunk - 1; ++l) {
        if (trunk_dims[l + 1] != W) return AMAR_EUNSUPPORTED;
        const int KT = l == 0 ? 2 * a.tb : a.tt;
        // the trunk's first layer is packed for a 2D-wide input: its k-tiles are those of [x_0 || x_1] only if D % 16 == 0
        if (l == 0 && (D & 15)) return AMAR_EUNSUPPORTED;
        a.tw_off[l] = off; off += a.tt * KT * 256;
        a.tbias_off[l] = off; off += 16 * a.tt;
        a.t_act[l] = trunk_acts[l];
    }
    a.dot_off = off; off += 16 * a.tt;
    a.dot_bias_off = off; off += 4;
    a.dot_act = trunk_acts[n_trunk - 1];
    a.wpack = wpack; a.wpack_floats = off; a.out = out; a.ldo = ldo; a.P = P; a.out_index = out_index;
    if ((size_t)off * sizeof(float) > 150 * 1024) return AMAR_EUNSUPPORTED;
    if (P == 0) return AMAR_OK;
    const size_t lds_bytes = (size_t)off * sizeof(float);
    constexpr int PT = 1;
    constexpr int THREADS = 1024;                          // the blob (~84 KB for 64-wide stacks) allows one workgroup per CU
    int64_t blocks = (P + (THREADS / 64) * 16 * PT - 1) / ((THREADS / 64) * 16 * PT);
    if (blocks > 1024) blocks = 1024;
    // the common shape of the hybrid head (64-wide everywhere, ids on every table, ReLU throughout): guard-free kernel
    static const bool no_full = getenv("AMAR_DUAL_FULL") && atoi(getenv("AMAR_DUAL_FULL")) == 0;
    bool full = !no_full && D == 64 && W == 64 && in_act == AMAR_ACT_RELU && n_branch >= 1 && ida[0] && ida[1] && idb[0] && idb[1];
    for (int b = 0; b < 2; ++b) full = full && lda[b] < (1ll << 32) && ldb[b] < (1ll << 32);
    for (int l = 0; l < n_branch; ++l) full = full && branch_acts[l] == AMAR_ACT_RELU;
    for (int l = 0; l < n_trunk - 1; ++l) full = full && trunk_acts[l] == AMAR_ACT_RELU;
    // the 64-wide head on the split products (see dual_chain_split_kernel); AMAR_PAIR_MFMA=f32: the f32 instruction
    static const bool f32_only = getenv("AMAR_PAIR_MFMA") && !strcmp(getenv("AMAR_PAIR_MFMA"), "f32");
    if (full && !f32_only && n_branch >= 1) {
        int nf = 0, tl = 0;
        for (int b = 0; b < 2; ++b)
            for (int l = 0; l < n_branch; ++l) { a.bw_frag[b][l] = nf; nf += 4 * 2; a.bb_tail[b][l] = tl; tl += 64; }
        for (int l = 0; l < a.n_trunk; ++l) { a.tw_frag[l] = nf; nf += 4 * (l == 0 ? 4 : 2); a.tb_tail[l] = tl; tl += 64; }
        a.dot_tail = tl; tl += 64; a.dot_bias_tail = tl; tl += 4;
        a.n_frag = nf; a.tail_floats = tl;
        const size_t bytes = (size_t)nf * 3 * 1024 + (size_t)tl * sizeof(float);
        if (bytes <= 160 * 1024) {
            static bool lds_ok[AMAR_MAX_DEVICES];
            // one pair tile per wave in 768-thread workgroups: three waves per SIMD with up to 168 registers each (164 used, no spills) —
            // 1 024 threads leave a lane 128 registers (25 spilled), two tiles per wave in 512 threads need 256 (two waves per SIMD, which do
            // not cover the un-prefetched gathers at the top of an iteration: 35 % of a wave's time in s_waitcnt).  ml1m(s=64), prepared
            // list: 2.70 ms against 2.82 (two tiles, 512 threads) and 2.93 (1 024 threads).  AMAR_DUAL_PT=1 / 2: those forms.
            static const int dual_pt = getenv("AMAR_DUAL_PT") ? atoi(getenv("AMAR_DUAL_PT")) : 0;
            static bool lds_ok2[AMAR_MAX_DEVICES], lds_ok3[AMAR_MAX_DEVICES];
            hipStream_t st = static_cast<hipStream_t>(stream);
            if (dual_pt == 1) {
                if (bytes > 64 * 1024)                                // (allowed once per device, for the largest image)
                    if (const int rc = amar_allow_lds(reinterpret_cast<const void *>(dual_chain_split_kernel<1, 1024>), 160 * 1024, lds_ok)) return rc;
                hipLaunchKernelGGL((dual_chain_split_kernel<1, 1024>), dim3((unsigned)blocks), dim3(1024), bytes, st, a);
            } else if (dual_pt == 2) {
                if (bytes > 64 * 1024)
                    if (const int rc = amar_allow_lds(reinterpret_cast<const void *>(dual_chain_split_kernel<2, 512>), 160 * 1024, lds_ok2)) return rc;
                int64_t blocks2 = (P + 8 * 32 - 1) / (8 * 32);
                if (blocks2 > 1024) blocks2 = 1024;
                hipLaunchKernelGGL((dual_chain_split_kernel<2, 512>), dim3((unsigned)blocks2), dim3(512), bytes, st, a);
            } else {
                if (bytes > 64 * 1024)
                    if (const int rc = amar_allow_lds(reinterpret_cast<const void *>(dual_chain_split_kernel<1, 768>), 160 * 1024, lds_ok3)) return rc;
                int64_t blocks3 = (P + 12 * 16 - 1) / (12 * 16);
                if (blocks3 > 1024) blocks3 = 1024;
                hipLaunchKernelGGL((dual_chain_split_kernel<1, 768>), dim3((unsigned)blocks3), dim3(768), bytes, st, a);
            }
            return amar_check_launch();
        }
    }
    auto kern = full ? dual_chain_full_kernel : dual_chain_kernel<PT>;
    if (lds_bytes > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
        return AMAR_ELAUNCH;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(THREADS), lds_bytes, static_cast<hipStream_t>(stream), a);
    return amar_check_launch();
}

}  // extern "C"
