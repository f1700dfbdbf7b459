// GNN propagation kernels for gfx950: CSR row gather-reduce with fused layer epilogues.
//
// Work decomposition (all four convolutions share it):
//   one 64-lane wavefront per CSR row; a row of width F floats is F/4 float4 "quads", so
//   LPN = F/4 lanes cooperate on one non-zero (one coalesced 4F-byte read of the source row)
//   and NS = 64/LPN non-zeros are in flight per wave-instruction.  Partial sums stay in
//   registers and are combined with a fixed xor-butterfly of wavefront shuffles, so results are
//   bitwise reproducible (no atomics).  The epilogue (bias, ReLU, running layer sum, the next
//   layer's tiny dense product, GraphSAGE's concat-dense-normalise, GAT's softmax weights) runs
//   in the same kernel on the reduced row, and the result is written once, straight into the
//   layer's column slice of the concatenation buffer.
//
// Reference semantics: see include/amar_hip.h (each entry point cites the reference file:line).
#include "amar_common.h"
#include <stdlib.h>

namespace {

constexpr int WAVES_PER_BLOCK = 4;

// Sum over the row's non-zeros of w_p * X[col_p, 4q:4q+4]; every lane returns the row total of its quad.
template <int LPN, bool HAS_VALS>
__device__ __forceinline__ float4 row_gather_sum(const int32_t *__restrict__ colidx, const float *__restrict__ vals,
                                                 const float *__restrict__ X, int64_t ldx, int beg, int end,
                                                 int slot, int q) {
    constexpr int NS = AMAR_WAVE / LPN;
    float4 acc = f4_zero();
    int p = beg + slot;
    // two non-zeros per lane in flight: both index loads are issued before either gather
    for (; p + NS < end; p += 2 * NS) {
        const int c0 = colidx[p], c1 = colidx[p + NS];
        const float v0 = HAS_VALS ? vals[p] : 1.f, v1 = HAS_VALS ? vals[p + NS] : 1.f;
        const float4 x0 = *reinterpret_cast<const float4 *>(X + (int64_t)c0 * ldx + 4 * q);
        const float4 x1 = *reinterpret_cast<const float4 *>(X + (int64_t)c1 * ldx + 4 * q);
        acc = f4_fma(v0, x0, acc);
        acc = f4_fma(v1, x1, acc);
    }
    if (p < end) {
        const int c0 = colidx[p];
        const float v0 = HAS_VALS ? vals[p] : 1.f;
        const float4 x0 = *reinterpret_cast<const float4 *>(X + (int64_t)c0 * ldx + 4 * q);
        acc = f4_fma(v0, x0, acc);
    }
    return f4_wave_sum_stride<LPN>(acc);
}

// All lanes receive the full reduced row y[0:F] (lane qq of slot 0 holds quad qq).
template <int F>
__device__ __forceinline__ void broadcast_row(const float4 yq, float (&full)[F]) {
#pragma unroll
    for (int qq = 0; qq < F / 4; ++qq) {                  // v_readlane: the value becomes a scalar operand
        full[4 * qq + 0] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, yq.x), qq));
        full[4 * qq + 1] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, yq.y), qq));
        full[4 * qq + 2] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, yq.z), qq));
        full[4 * qq + 3] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, yq.w), qq));
    }
}

struct SpmmArgs {
    const int32_t *rowptr; const int32_t *colidx; const float *vals;
    const float *X; int64_t ldx;
    float *Y; int64_t ldy;
    const float *bias; int relu;
    const float *acc_in; int64_t ld_acc_in; float *acc_out; int64_t ld_acc_out; float acc_div; int accum; int accum_div;
    const float *Wn; int Cn; float *Hn; int64_t ldhn;
    int n_rows;
    const float *next_scale;                     // Hn rows are multiplied by next_scale[row] (value-free XS image), or NULL
};

// Bias / ReLU / store / running layer sum / next layer's dense product for one reduced row.
template <int F, bool FUSE_NEXT>
__device__ __forceinline__ void spmm_epilogue(const SpmmArgs &a, int row, float4 y, int q, int slot, int lane,
                                              const float (&wn)[F]) {
    if (a.bias) {
        const float4 b = *reinterpret_cast<const float4 *>(a.bias + 4 * q);
        y = f4_add(y, b);
    }
    if (a.relu) { y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f); }
    if (slot == 0) {
        if (a.Y) *reinterpret_cast<float4 *>(a.Y + (int64_t)row * a.ldy + 4 * q) = y;
        if (a.accum) {
            float4 s = *reinterpret_cast<const float4 *>(a.acc_in + (int64_t)row * a.ld_acc_in + 4 * q);
            s = f4_add(s, y);
            if (a.accum_div) { s.x /= a.acc_div; s.y /= a.acc_div; s.z /= a.acc_div; s.w /= a.acc_div; }
            *reinterpret_cast<float4 *>(a.acc_out + (int64_t)row * a.ld_acc_out + 4 * q) = s;
        }
    }
    if (FUSE_NEXT) {
        float full[F];
        broadcast_row<F>(y, full);
        float h = 0.f;
#pragma unroll
        for (int k = 0; k < F; ++k) h = fmaf(full[k], wn[k], h);
        if (lane < a.Cn) a.Hn[(int64_t)row * a.ldhn + lane] = h;
    }
}

// v1: one wavefront per row (kept for A/B timing: AMAR_SPMM_V1=1).
template <int F, bool HAS_VALS, bool FUSE_NEXT>
__global__ __launch_bounds__(WAVES_PER_BLOCK * AMAR_WAVE) void spmm_row_kernel(const SpmmArgs a) {
    constexpr int LPN = F / 4;
    const int lane = threadIdx.x & (AMAR_WAVE - 1);
    const int row = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= a.n_rows) return;                       // wave-uniform
    const int q = lane % LPN, slot = lane / LPN;
    float wn[F];                                       // column `lane` of the next layer's kernel
    if (FUSE_NEXT) {
#pragma unroll
        for (int k = 0; k < F; ++k) wn[k] = lane < a.Cn ? a.Wn[k * a.Cn + lane] : 0.f;
    }
    const int beg = a.rowptr[row], end = a.rowptr[row + 1];
    const float4 y = row_gather_sum<LPN, HAS_VALS>(a.colidx, a.vals, a.X, a.ldx, beg, end, slot, q);
    spmm_epilogue<F, FUSE_NEXT>(a, row, y, q, slot, lane, wn);
}

// v2: streaming form.  A wavefront owns `rpw` consecutive rows, i.e. ONE contiguous range of the
// colidx/vals arrays.  It streams that range in coalesced 64-entry register tiles (lane l holds
// entry pt + l), always one tile ahead of use, so the HBM latency of the CSR stream is paid once
// per wave instead of once per row; row boundaries come from one coalesced rowptr load kept in a
// register and read with v_readlane.  Gather slots pick their (column, value) out of the tile
// registers with a wavefront shuffle (ds_bpermute), gather the source row quad, and accumulate;
// rows are finished in order with the same xor-butterfly and epilogue as v1, so v1 and v2 agree
// bit for bit whenever a row's non-zeros fall in the same slot order (they do: slot = index mod NS).
template <int F, bool HAS_VALS, bool FUSE_NEXT>
__global__ __launch_bounds__(WAVES_PER_BLOCK * AMAR_WAVE) void spmm_stream_kernel(const SpmmArgs a, const int rpw) {
    constexpr int LPN = F / 4, NS = AMAR_WAVE / LPN;
    const int lane = threadIdx.x & (AMAR_WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6));
    const int r0 = wave * rpw;
    if (r0 >= a.n_rows) return;
    const int nr = min(rpw, a.n_rows - r0);
    const int q = lane % LPN, slot = lane / LPN;
    float wn[F];
    if (FUSE_NEXT) {
#pragma unroll
        for (int k = 0; k < F; ++k) wn[k] = lane < a.Cn ? a.Wn[k * a.Cn + lane] : 0.f;
    }
    const int rp = a.rowptr[r0 + min(lane, nr)];
    const int p1 = __builtin_amdgcn_readlane(rp, nr);
    int pt = __builtin_amdgcn_readlane(rp, 0);          // base of the current tile
    int ccur = 0, cnxt = 0;
    float vcur = 1.f, vnxt = 1.f;
    if (pt + lane < p1) { ccur = a.colidx[pt + lane]; if (HAS_VALS) vcur = a.vals[pt + lane]; }
    if (pt + 64 + lane < p1) { cnxt = a.colidx[pt + 64 + lane]; if (HAS_VALS) vnxt = a.vals[pt + 64 + lane]; }

    int row_beg = pt;
    for (int k = 0; k < nr; ++k) {
        const int row_end = __builtin_amdgcn_readlane(rp, k + 1);
        float4 acc = f4_zero();
        int pos = row_beg;
        // slot s of the row takes the row's non-zeros s, s+NS, s+2NS, ... exactly like v1
        while (pos < row_end) {
            if (pos >= pt + 64) {                       // advance one tile, keep one tile in flight
                ccur = cnxt; vcur = vnxt; pt += 64;
                cnxt = 0; vnxt = 1.f;
                if (pt + 64 + lane < p1) { cnxt = a.colidx[pt + 64 + lane]; if (HAS_VALS) vnxt = a.vals[pt + 64 + lane]; }
            }
            const int hi = min(row_end, pt + 64);
            // GPI gathers in flight per pass: a whole 64-entry tile from F = 16 on (round 4: with two, a wave walking the heaviest row
            // of ml1m(s=1) — 1 400 entries, 16 per gather at F = 16 — paid 44 dependent memory round trips: the 15 us of every
            // propagation launch of a training batch); each slot still adds its entries in the same order (s, s + NS, s + 2 NS, ...)
            constexpr int GPI = LPN <= 2 ? 2 : (LPN >= 8 ? 8 : LPN);
            for (int base = pos; base < hi; base += GPI * NS) {
                // map tile positions back to the row-relative slot order: position p belongs to slot (p - row_beg) % NS
                const int i0 = base + ((slot - (base - row_beg)) & (NS - 1));
                float4 xg[GPI];
                float vg[GPI];
#pragma unroll
                for (int j = 0; j < GPI; ++j) {
                    const int ij = i0 + j * NS;
                    const int cj = __shfl(ccur, (ij - pt) & 63, 64);
                    vg[j] = HAS_VALS ? __shfl(vcur, (ij - pt) & 63, 64) : 1.f;
                    xg[j] = ij < hi ? *reinterpret_cast<const float4 *>(a.X + (int64_t)cj * a.ldx + 4 * q) : f4_zero();
                }
#pragma unroll
                for (int j = 0; j < GPI; ++j)
                    if (i0 + j * NS < hi) acc = f4_fma(vg[j], xg[j], acc);
            }
            pos = hi;
        }
        acc = f4_wave_sum_stride<LPN>(acc);
        spmm_epilogue<F, FUSE_NEXT>(a, r0 + k, acc, q, slot, lane, wn);
        row_beg = row_end;
    }
}

int spmm_rows_per_wave(int n_rows) {
    // enough waves to fill 256 CUs x 32 wave slots a few times over, at most 16 rows per wave
    int rpw = n_rows / (256 * 32 * 4);
    return rpw < 1 ? 1 : (rpw > 16 ? 16 : rpw);
}

template <bool HAS_VALS, bool FUSE_NEXT>
int launch_spmm(const SpmmArgs &a, int F, hipStream_t st) {
    static const bool use_v1 = getenv("AMAR_SPMM_V1") != nullptr;
    const dim3 block(WAVES_PER_BLOCK * AMAR_WAVE);
    if (use_v1) {
        const dim3 grid((a.n_rows + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK);
        switch (F) {
        case 4:  hipLaunchKernelGGL((spmm_row_kernel<4, HAS_VALS, FUSE_NEXT>), grid, block, 0, st, a); break;
        case 8:  hipLaunchKernelGGL((spmm_row_kernel<8, HAS_VALS, FUSE_NEXT>), grid, block, 0, st, a); break;
        case 16: hipLaunchKernelGGL((spmm_row_kernel<16, HAS_VALS, FUSE_NEXT>), grid, block, 0, st, a); break;
        case 32: hipLaunchKernelGGL((spmm_row_kernel<32, HAS_VALS, FUSE_NEXT>), grid, block, 0, st, a); break;
        case 64: hipLaunchKernelGGL((spmm_row_kernel<64, HAS_VALS, FUSE_NEXT>), grid, block, 0, st, a); break;
        default: return AMAR_EUNSUPPORTED;
        }
        return amar_check_launch();
    }
    const int rpw = spmm_rows_per_wave(a.n_rows);
    const int waves = (a.n_rows + rpw - 1) / rpw;
    const dim3 grid((waves + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK);
    switch (F) {
    case 4:  hipLaunchKernelGGL((spmm_stream_kernel<4, HAS_VALS, FUSE_NEXT>), grid, block, 0, st, a, rpw); break;
    case 8:  hipLaunchKernelGGL((spmm_stream_kernel<8, HAS_VALS, FUSE_NEXT>), grid, block, 0, st, a, rpw); break;
    case 16: hipLaunchKernelGGL((spmm_stream_kernel<16, HAS_VALS, FUSE_NEXT>), grid, block, 0, st, a, rpw); break;
    case 32: hipLaunchKernelGGL((spmm_stream_kernel<32, HAS_VALS, FUSE_NEXT>), grid, block, 0, st, a, rpw); break;
    case 64: hipLaunchKernelGGL((spmm_stream_kernel<64, HAS_VALS, FUSE_NEXT>), grid, block, 0, st, a, rpw); break;
    default: return AMAR_EUNSUPPORTED;
    }
    return amar_check_launch();
}

// Epilogue for kernels where ONE LANE holds a whole reduced row (acc[q] = features 4q..4q+3):
// bias, ReLU, store into the concat slice, running layer sum, and the next layer's X.W with
// wave-uniform (scalar) reads of Wnext.
template <int F, bool FUSE_NEXT>
__device__ __forceinline__ void lane_row_epilogue(const SpmmArgs &e, int row, float4 (&acc)[F / 4], const float *wn_lds = nullptr) {
    constexpr int LPN = F / 4;
#pragma unroll
    for (int q = 0; q < LPN; ++q) {
        float4 y = acc[q];
        if (e.bias) y = f4_add(y, *reinterpret_cast<const float4 *>(e.bias + 4 * q));
        if (e.relu) { y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f); }
        acc[q] = y;
        if (e.Y) *reinterpret_cast<float4 *>(e.Y + (int64_t)row * e.ldy + 4 * q) = y;
        if (e.accum) {
            float4 s = *reinterpret_cast<const float4 *>(e.acc_in + (int64_t)row * e.ld_acc_in + 4 * q);
            s = f4_add(s, y);
            if (e.accum_div) { s.x /= e.acc_div; s.y /= e.acc_div; s.z /= e.acc_div; s.w /= e.acc_div; }
            *reinterpret_cast<float4 *>(e.acc_out + (int64_t)row * e.ld_acc_out + 4 * q) = s;
        }
    }
    if (FUSE_NEXT && wn_lds && (e.Cn & 3) == 0) {
        // the kernel staged in LDS ([F][Cn], read as broadcasts); same order of operations per output as the scalar form below
        for (int j0 = 0; j0 < e.Cn; j0 += 4) {
            float4 h = f4_zero();
#pragma unroll
            for (int q = 0; q < LPN; ++q)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    h = f4_fma(f4_get(acc[q], c), *reinterpret_cast<const float4 *>(wn_lds + (4 * q + c) * e.Cn + j0), h);
            if (e.next_scale) { const float sc = e.next_scale[row]; h.x *= sc; h.y *= sc; h.z *= sc; h.w *= sc; }
            float *dst = e.Hn + (int64_t)row * e.ldhn + j0;
            if ((e.ldhn & 3) == 0) *reinterpret_cast<float4 *>(dst) = h;
            else { dst[0] = h.x; dst[1] = h.y; dst[2] = h.z; dst[3] = h.w; }
        }
    } else if (FUSE_NEXT) {
        for (int j0 = 0; j0 < e.Cn; j0 += 4) {
            float h[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < LPN; ++q)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float yk = f4_get(acc[q], c);
                    const float *w = e.Wn + (int64_t)(4 * q + c) * e.Cn + j0;
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (j0 + t < e.Cn) h[t] = fmaf(yk, w[t], h[t]);
                }
            if (e.next_scale) { const float sc = e.next_scale[row]; h[0] *= sc; h[1] *= sc; h[2] *= sc; h[3] *= sc; }
            float *dst = e.Hn + (int64_t)row * e.ldhn + j0;
            if (j0 + 3 < e.Cn && (e.ldhn & 3) == 0) *reinterpret_cast<float4 *>(dst) = make_float4(h[0], h[1], h[2], h[3]);
            else for (int t = 0; t < 4 && j0 + t < e.Cn; ++t) dst[t] = h[t];
        }
    }
}

// ---- v3: sliced-jagged (SJ) SpMM ----------------------------------------------------------------
// Why: on ml1m(s=64) the row-gather kernels above move 4.3 GB through the fabric per launch for
// 0.49 GB of algorithmic bytes (rocprofv3: every L2 miss is a 128-B line fill for a 32-B gather,
// L2 hit rate 45 %).  Here (format: utilities/math.py:SlicedJagged, include/amar_hip.h)
//   * columns are cut into slices whose part of X fits the 4 MB per-XCD L2 and every wave sweeps
//     the slices in the same order, so gathers hit L2 and each XCD fetches each slice once;
//   * ONE LANE OWNS ONE ROW: its F partial sums stay in registers over all slices — no cross-lane
//     reduction, no shuffles, no LDS; the sum runs in ascending column order (a scalar CSR loop);
//   * inside a (wave, slice) block the non-zeros are in jagged-diagonal order, so at step j the
//     active lanes (ballot + mbcnt) read consecutive 8-byte (col, val) entries: coalesced, unpadded.
struct SjArgs {
    const int2 *entries; const int16_t *counts; const int32_t *wave_start; int n_slices; int n_waves;
    SpmmArgs e;                                        // X/ldx, outputs and epilogue options (rowptr/colidx/vals unused)
};

__device__ __forceinline__ int lanes_below(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

template <int F, bool FUSE_NEXT>
__global__ __launch_bounds__(WAVES_PER_BLOCK * AMAR_WAVE) void spmm_sj_kernel(const SjArgs a) {
    constexpr int LPN = F / 4;
    const int lane = threadIdx.x & (AMAR_WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6));
    if (wave >= a.n_waves) return;
    const int row = wave * AMAR_WAVE + lane;
    const SpmmArgs &e = a.e;
    int base = a.wave_start[wave];
    const int16_t *cptr = a.counts + (int64_t)wave * a.n_slices * AMAR_WAVE + lane;
    float4 acc[LPN];
#pragma unroll
    for (int q = 0; q < LPN; ++q) acc[q] = f4_zero();

    int cnt_next = cptr[0];
    for (int k = 0; k < a.n_slices; ++k) {
        const int cnt = cnt_next;
        if (k + 1 < a.n_slices) cnt_next = cptr[(k + 1) * AMAR_WAVE];
        for (int j = 0;; j += 2) {
            const bool a0 = cnt > j, a1 = cnt > j + 1;
            const unsigned long long b0 = __ballot(a0);
            if (b0 == 0ull) break;
            const unsigned long long b1 = __ballot(a1);
            const int n0 = __popcll(b0);
            const int i0 = base + lanes_below(b0), i1 = base + n0 + lanes_below(b1);
            base += n0 + __popcll(b1);
            int2 e0 = make_int2(0, 0), e1 = make_int2(0, 0);
            if (a0) e0 = a.entries[i0];
            if (a1) e1 = a.entries[i1];
            float4 x0[LPN], x1[LPN];
#pragma unroll
            for (int q = 0; q < LPN; ++q) {
                x0[q] = f4_zero(); x1[q] = f4_zero();
                if (a0) x0[q] = *reinterpret_cast<const float4 *>(e.X + (int64_t)e0.x * e.ldx + 4 * q);
                if (a1) x1[q] = *reinterpret_cast<const float4 *>(e.X + (int64_t)e1.x * e.ldx + 4 * q);
            }
            const float v0 = __int_as_float(e0.y), v1 = __int_as_float(e1.y);
#pragma unroll
            for (int q = 0; q < LPN; ++q) {
                if (a0) acc[q] = f4_fma(v0, x0[q], acc[q]);
                if (a1) acc[q] = f4_fma(v1, x1[q], acc[q]);
            }
        }
    }
    if (row >= e.n_rows) return;
    lane_row_epilogue<F, FUSE_NEXT>(e, row, acc);
}

template <bool FUSE_NEXT>
int launch_spmm_sj(const SjArgs &a, int F, hipStream_t st) {
    const dim3 grid((a.n_waves + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK), block(WAVES_PER_BLOCK * AMAR_WAVE);
    switch (F) {
    case 4:  hipLaunchKernelGGL((spmm_sj_kernel<4, FUSE_NEXT>), grid, block, 0, st, a); break;
    case 8:  hipLaunchKernelGGL((spmm_sj_kernel<8, FUSE_NEXT>), grid, block, 0, st, a); break;
    case 16: hipLaunchKernelGGL((spmm_sj_kernel<16, FUSE_NEXT>), grid, block, 0, st, a); break;
    case 32: hipLaunchKernelGGL((spmm_sj_kernel<32, FUSE_NEXT>), grid, block, 0, st, a); break;
    case 64: hipLaunchKernelGGL((spmm_sj_kernel<64, FUSE_NEXT>), grid, block, 0, st, a); break;
    default: return AMAR_EUNSUPPORTED;
    }
    return amar_check_launch();
}

// ---- v4: XCD-sliced (XS) SpMM = per-slice partial products + combine ------------------------------
// Format and rationale: utilities/math.py:XcdSliced, include/amar_hip.h.  Workgroup b works on column
// slice b % S only, so (with the round-robin dispatch of workgroups over the 8 XCDs) each XCD's L2
// holds one slice of X and nothing else of it.  The partial rows of a (64-row block, slice) pair go to
// P[slice][row]; a second kernel adds diag . X and the non-empty partials in slice order (a fixed order:
// results are bitwise reproducible) and applies the fused layer epilogue.
struct XsArgs {
    const int32_t *rowptr; const int32_t *colidx; const float *vals;     // XS image
    const float *X; int64_t ldx; float *P; int n_rows; int n_slices; int blocks_per_slice;
    bool off32;                                                          // every byte offset into X < 2^32
};

// Lanes take CONSECUTIVE entries of the wave's (64-row block, slice) range: lane = q * EPS + s handles feature
// quad q of entry s of the current step (EPS = 64 / (F/4) entries per step), so colidx/vals loads are coalesced
// and every lane gathers.  The row of an entry rides in bits 26..31 of its column word (row & 63); since entries
// are sorted by row, a DPP segmented inclusive scan over s (add the value d lanes back iff its key is equal)
// leaves each row-run's sum in the run's last lane, which adds it to the wave's LDS accumulator [64 rows][F].
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_pull(float v) {     // lanes outside ROW_MASK / without a source get 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_pull_key(int k) {     // ... and key -1 (matches no row)
    return __builtin_amdgcn_update_dpp(-1, k, CTRL, ROW_MASK, 0xF, false);
}

template <int CTRL, int ROW_MASK, int DIST, int EPS>
__device__ __forceinline__ void seg_scan_step(float4 &p, int key, int s) {
    const int kprev = dpp_pull_key<CTRL, ROW_MASK>(key);
    const float m = (kprev == key && (DIST == 0 || s >= DIST)) ? 1.f : 0.f;
    p.x = fmaf(m, dpp_pull<CTRL, ROW_MASK>(p.x), p.x); p.y = fmaf(m, dpp_pull<CTRL, ROW_MASK>(p.y), p.y);
    p.z = fmaf(m, dpp_pull<CTRL, ROW_MASK>(p.z), p.z); p.w = fmaf(m, dpp_pull<CTRL, ROW_MASK>(p.w), p.w);
}

// The same scan for EPS >= 16 in hand-scheduled ISA.  hipcc turns `fmaf(m, dpp(p), p)` into v_mov 0 + v_mov_dpp +
// v_pk_fma (VOP3P has no DPP form): 14 VALU per level.  v_fmac_f32_dpp pulls and accumulates in one instruction and
// leaves a lane without a DPP source untouched, which is exactly the scan's "no left neighbour" case: 7 per level.
// Hazards (cdna_hip_programming.md 5.7): a DPP read needs 2 wait states after the VALU write of its source: the
// leading s_nop covers whatever the compiler put in front; inside the block 3+ instructions separate every pair.
#define AMAR_SEG_LEVEL(SHIFT)                                              \
    "v_mov_b32_dpp %4, %6 " SHIFT " bank_mask:0xf bound_ctrl:1\n\t"        \
    "v_cmp_eq_u32_e32 vcc, %4, %6\n\t"                                     \
    "v_cndmask_b32_e64 %5, 0, 1.0, vcc\n\t"                                \
    "v_fmac_f32_dpp %0, %0, %5 " SHIFT " bank_mask:0xf\n\t"                \
    "v_fmac_f32_dpp %1, %1, %5 " SHIFT " bank_mask:0xf\n\t"                \
    "v_fmac_f32_dpp %2, %2, %5 " SHIFT " bank_mask:0xf\n\t"                \
    "v_fmac_f32_dpp %3, %3, %5 " SHIFT " bank_mask:0xf\n\t"
template <int EPS>
__device__ __forceinline__ void seg_scan_isa(float4 &p, int key) {
    static_assert(EPS >= 16, "a 16-lane DPP row must not span several feature groups");
    int kprev; float m;
    asm volatile("s_nop 1\n\t"
                 AMAR_SEG_LEVEL("row_shr:1 row_mask:0xf") AMAR_SEG_LEVEL("row_shr:2 row_mask:0xf")
                 AMAR_SEG_LEVEL("row_shr:4 row_mask:0xf") AMAR_SEG_LEVEL("row_shr:8 row_mask:0xf")
                 : "+v"(p.x), "+v"(p.y), "+v"(p.z), "+v"(p.w), "=&v"(kprev), "=&v"(m) : "v"(key) : "vcc");
    if (EPS >= 32)      // lane 15 of rows 0 / 2 into every lane of rows 1 / 3 (rows 0 / 2 are not written: their m is moot)
        asm volatile("s_nop 1\n\t" AMAR_SEG_LEVEL("row_bcast:15 row_mask:0xa")
                     : "+v"(p.x), "+v"(p.y), "+v"(p.z), "+v"(p.w), "=&v"(kprev), "=&v"(m) : "v"(key) : "vcc");
    if (EPS >= 64)
        asm volatile("s_nop 1\n\t" AMAR_SEG_LEVEL("row_bcast:31 row_mask:0xc")
                     : "+v"(p.x), "+v"(p.y), "+v"(p.z), "+v"(p.w), "=&v"(kprev), "=&v"(m) : "v"(key) : "vcc");
}
#undef AMAR_SEG_LEVEL

// Workgroup -> (row chunk, slice) with XCD affinity.  n_slices = 8 * phases: workgroup b lands on XCD b % 8 (round-robin
// dispatch) and works, during phase p = b / (8 * blocks_per_slice), on slice 8 p + b % 8 — so at any time an XCD's L2 holds ONE
// slice of X, also when the table is several times larger than the aggregate L2.  (n_slices not a multiple of 8: plain b % S.)
__device__ __forceinline__ void xs_block_to_tile(int b, int n_slices, int blocks_per_slice, int &chunk, int &slice) {
    if ((n_slices & 7) == 0) {
        const int per_phase = 8 * blocks_per_slice;
        const int phase = b / per_phase, r = b - phase * per_phase;
        chunk = r >> 3;
        slice = phase * 8 + (r & 7);
    } else {
        slice = b % n_slices;
        chunk = b / n_slices;
    }
}

// OFF32: every byte offset into X fits 32 bits, so the gathers take the `saddr + voffset` form and the per-lane
// address arithmetic is one multiply-add instead of a 64-bit chain.
//
// Work split inside a wave (v4.2).  A "super-step" is EPS * EPL consecutive entries of the tile; lane = q * EPS + s
// (q = feature quad, EPS = 64 / (F/4) lanes per quad) owns the EPL CONSECUTIVE entries s*EPL .. s*EPL+EPL-1:
//   1. 16-byte non-temporal loads of its EPL column words and values (the read-once stream), then EPL 16-byte gathers;
//   2. an in-lane serial segmented sum over those entries: whenever the row key changes, the finished run goes to
//      the wave's LDS accumulator with ds_add_f32; the lane keeps its LAST run;
//   3. ONE cross-lane segmented scan (seg_scan_isa) over the lanes' last runs, keyed by the last key: since the entries
//      are sorted by row, "key of lane s-d == key of lane s" still means every entry in between belongs to that row;
//   4. the last lane of every cross-lane run adds the result to the LDS accumulator.
// What bounds it (tools/exp_xs_floor.py, tools/exp_gather.py): a 32-byte row costs a whole 128-byte
// line fill into the CU's L1, and a CU sustains 0.43-0.45 line fills per clock from its XCD's L2 whatever the lane
// layout (= 0.21 ms for the 55.9 M gathers of ml1m(s=64)); the index stream adds its HBM time on top.  VALU work
// (3.5x less than the one-entry-per-lane v4.1) and the LDS adds are hidden behind that.

constexpr int XS_WAVES = 4;                     // waves per workgroup of the partial kernel: one tile each (1 or 4: same time)
template <int F, bool OFF32, int EPL, bool VALS>
__global__ __launch_bounds__(XS_WAVES * AMAR_WAVE) void spmm_xs_partial_kernel(const XsArgs a) {
    constexpr int LPN = F / 4, EPS = AMAR_WAVE / LPN, SUPER = EPS * EPL;
    constexpr int PAD_KEY = AMAR_WAVE;                               // key of an entry past the end: equals no row, never flushed
    __shared__ float lds_acc[XS_WAVES][AMAR_WAVE * F];
    const int lane = threadIdx.x & (AMAR_WAVE - 1);
    int k, chunk;                                                    // slice <-> XCD affinity
    xs_block_to_tile(blockIdx.x, a.n_slices, a.blocks_per_slice, chunk, k);
    const int r0 = __builtin_amdgcn_readfirstlane((chunk * XS_WAVES + (threadIdx.x >> 6)) * AMAR_WAVE);
    if (r0 >= a.n_rows) return;
    const int nr = min(AMAR_WAVE, a.n_rows - r0);
    const int32_t *rp = a.rowptr + (int64_t)k * a.n_rows + r0;
    const int beg = rp[0], end = rp[nr];                             // wave-uniform (scalar loads)
    if (beg == end) return;                                          // empty (block, slice): the combine skips it too
    float *acc = lds_acc[threadIdx.x >> 6];
#pragma unroll
    for (int c = 0; c < F; ++c) acc[c * AMAR_WAVE + lane] = 0.f;     // wave-private region; LDS ops of a wave stay in order
    const int q = lane / EPS, s = lane % EPS;
    const int n_tile = end - beg;
    const char *cbase = reinterpret_cast<const char *>(a.colidx + beg);   // scalar bases + 32-bit lane offsets
    const char *vbase = VALS ? reinterpret_cast<const char *>(a.vals + beg) : nullptr;   // value-free image: every entry weighs 1

    auto flush = [&](int key, const float4 &p) {                     // acc[feature][row]: lanes of one instruction hit distinct rows
        float *dst = acc + key;
        atomicAdd(dst + (4 * q + 0) * AMAR_WAVE, p.x); atomicAdd(dst + (4 * q + 1) * AMAR_WAVE, p.y);
        atomicAdd(dst + (4 * q + 2) * AMAR_WAVE, p.z); atomicAdd(dst + (4 * q + 3) * AMAR_WAVE, p.w);
    };
    auto gather = [&](int cw) {
        const unsigned col = (unsigned)cw & 0x3ffffffu;
        if (OFF32) return *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(a.X) + (col * (unsigned)a.ldx + 4u * q) * 4u);
        return *reinterpret_cast<const float4 *>(a.X + (int64_t)col * a.ldx + 4 * q);
    };
    // the lane's EPL column words and values of the super-step that starts at t0 (entries past the end re-read the last one)
    auto load_words = [&](int t0, int (&cw)[EPL], float (&v)[EPL]) {
        const int first = t0 + s * EPL;
        if (t0 + SUPER <= n_tile) {                                  // wave-uniform: a full super-step, 16-byte loads
#pragma unroll
            for (int j4 = 0; j4 < EPL; j4 += 4) {
                // read-once stream: non-temporal, so that it does not push the X slice out of the XCD's L2
                // (hipcc merges the four dword loads into one global_load_dwordx4 nt; 4-byte alignment is enough)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    cw[j4 + j] = __builtin_nontemporal_load(reinterpret_cast<const int32_t *>(cbase + (unsigned)(first + j4 + j) * 4u));
                    v[j4 + j] = VALS ? __builtin_nontemporal_load(reinterpret_cast<const float *>(vbase + (unsigned)(first + j4 + j) * 4u)) : 1.f;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                const unsigned off = (unsigned)max(min(first + j, n_tile - 1), 0) * 4u;
                cw[j] = *reinterpret_cast<const int32_t *>(cbase + off);
                v[j] = VALS ? *reinterpret_cast<const float *>(vbase + off) : 1.f;
            }
        }
    };

    int cw[EPL];
    float v[EPL];
    load_words(0, cw, v);
    for (int t0 = 0; t0 < n_tile; t0 += SUPER) {
        const int first = t0 + s * EPL;                              // the lane's first entry, relative to the tile
        float4 x[EPL];
        int key[EPL];
        float vv[EPL];
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            x[j] = gather(cw[j]);
            key[j] = first + j < n_tile ? (int)((unsigned)cw[j] >> 26) : PAD_KEY;
            vv[j] = v[j];
        }
        // the next super-step's words are requested as soon as this one's gathers are out: a tile is only ~3 super-steps
        // long, so the stream's latency would otherwise be paid once per step, in series with the gathers'
        if (t0 + SUPER < n_tile) load_words(t0 + SUPER, cw, v);

        float4 cur = make_float4(vv[0] * x[0].x, vv[0] * x[0].y, vv[0] * x[0].z, vv[0] * x[0].w);
#pragma unroll
        for (int j = 1; j < EPL; ++j) {
            const bool same = key[j] == key[j - 1];
            if (!same) flush(key[j - 1], cur);                       // key[j-1] is a real row: padding only follows padding
            const float keep = same ? 1.f : 0.f;
            cur = make_float4(fmaf(vv[j], x[j].x, keep * cur.x), fmaf(vv[j], x[j].y, keep * cur.y),
                              fmaf(vv[j], x[j].z, keep * cur.z), fmaf(vv[j], x[j].w, keep * cur.w));
        }
        const int kl = key[EPL - 1];
        if constexpr (EPS >= 16) {
            seg_scan_isa<EPS>(cur, kl);
        } else {                                                     // a 16-lane DPP row spans several feature groups: guarded C++ form
            seg_scan_step<0x111, 0xF, 1, EPS>(cur, kl, s);
            if (EPS >= 4) seg_scan_step<0x112, 0xF, 2, EPS>(cur, kl, s);
            if (EPS >= 8) seg_scan_step<0x114, 0xF, 4, EPS>(cur, kl, s);
        }
        // a cross-lane run ends where the next lane's last key differs (or the group ends); wave_shl:1 = 0x130: lane l reads l + 1
        const int knext = __builtin_amdgcn_mov_dpp(kl, 0x130, 0xF, 0xF, true);
        if (kl != PAD_KEY && (s == EPS - 1 || knext != kl)) flush(kl, cur);
    }
    if (lane < nr) {
        float *out = a.P + ((int64_t)k * a.n_rows + r0 + lane) * F;
#pragma unroll
        for (int c4 = 0; c4 < LPN; ++c4)
            *reinterpret_cast<float4 *>(out + 4 * c4) =
                make_float4(acc[(4 * c4 + 0) * AMAR_WAVE + lane], acc[(4 * c4 + 1) * AMAR_WAVE + lane],
                            acc[(4 * c4 + 2) * AMAR_WAVE + lane], acc[(4 * c4 + 3) * AMAR_WAVE + lane]);
    }
}

struct XsCombineArgs { const float *diag; const float *P; const int32_t *rowptr; int n_slices; const float *row_scale; const float *Xself; SpmmArgs e; };

template <int F, bool FUSE_NEXT>
__global__ __launch_bounds__(256) void spmm_xs_combine_kernel(const XsCombineArgs a) {
    constexpr int LPN = F / 4;
    const int row = blockIdx.x * 256 + threadIdx.x;
    const SpmmArgs &e = a.e;
    if (row >= e.n_rows) return;
    float4 acc[LPN];
    const float d = a.diag[row];
#pragma unroll
    for (int q = 0; q < LPN; ++q) {
        const float4 x = *reinterpret_cast<const float4 *>(a.Xself + (int64_t)row * e.ldx + 4 * q);   // the row's own features
        acc[q] = make_float4(d * x.x, d * x.y, d * x.z, d * x.w);
    }
    const int w0 = __builtin_amdgcn_readfirstlane(row & ~(AMAR_WAVE - 1));      // the partial kernel's 64-row block
    const int w1 = min(w0 + AMAR_WAVE, e.n_rows);
    for (int k = 0; k < a.n_slices; ++k) {
        const int32_t *rp = a.rowptr + (int64_t)k * e.n_rows;
        if (rp[w0] == rp[w1]) continue;                              // nothing was written for this (block, slice)
        const float *p = a.P + ((int64_t)k * e.n_rows + row) * F;
#pragma unroll
        for (int q = 0; q < LPN; ++q) acc[q] = f4_add(acc[q], *reinterpret_cast<const float4 *>(p + 4 * q));
    }
    if (a.row_scale) {                                               // value-free image: y_i = d_i^-1/2 * sum_j c_ij x'_j
        const float sc = a.row_scale[row];
#pragma unroll
        for (int q = 0; q < LPN; ++q) { acc[q].x *= sc; acc[q].y *= sc; acc[q].z *= sc; acc[q].w *= sc; }
    }
    lane_row_epilogue<F, FUSE_NEXT>(e, row, acc);
}

template <int F>
int launch_spmm_xs(const XsArgs &pa, const XsCombineArgs &ca, bool fuse, hipStream_t st) {
    const dim3 block(XS_WAVES * AMAR_WAVE);
    const dim3 pgrid((unsigned)(pa.blocks_per_slice * pa.n_slices));
    constexpr int EPL = F >= 8 ? 8 : 4;
    if (pa.vals) {
        if (pa.off32) hipLaunchKernelGGL((spmm_xs_partial_kernel<F, true, EPL, true>), pgrid, block, 0, st, pa);
        else hipLaunchKernelGGL((spmm_xs_partial_kernel<F, false, EPL, true>), pgrid, block, 0, st, pa);
    } else {
        if (pa.off32) hipLaunchKernelGGL((spmm_xs_partial_kernel<F, true, EPL, false>), pgrid, block, 0, st, pa);
        else hipLaunchKernelGGL((spmm_xs_partial_kernel<F, false, EPL, false>), pgrid, block, 0, st, pa);
    }
    const dim3 cgrid((ca.e.n_rows + 255) / 256);
    if (fuse) hipLaunchKernelGGL((spmm_xs_combine_kernel<F, true>), cgrid, dim3(256), 0, st, ca);
    else hipLaunchKernelGGL((spmm_xs_combine_kernel<F, false>), cgrid, dim3(256), 0, st, ca);
    return amar_check_launch();
}

// ---- v5: LDS-tiled (LT) SpMM: column-ordered windows, the Y tile in LDS, one launch -------------------------------
// Format and rationale: utilities/lds_tiled.py, include/amar_hip.h.  One workgroup (16 waves, one per CU: the tile takes
// 128 KB of LDS) owns a tile of consecutive rows and walks its entries in COLUMN order, a window of a few hundred
// neighbouring columns at a time, so that the window's rows of X are fetched into the CU's L1 once and every further
// entry of the window hits them (2-4 entries share a 128-byte line on ml1m(s=64)) — instead of the one L2 request per
// gathered 32-byte row that bounds the XS kernels.  The unit of ownership is the virtual row (a long row is cut into
// several): virtual row v belongs to wave v % 16, whose LDS rows are private, so the accumulation is a plain
// ds_read_b128 / add / ds_write_b128 (LDS float atomics run at ~3 clocks per LANE on gfx950).  The image spreads a
// wave's entries so that one step's EPS entries hit distinct LDS rows; a repeat in the very next slot is folded into its
// neighbour's registers with one DPP shift, any other repeat is flagged and added with ds_add_f32 after the step.  The
// waves keep G steps of gathers in flight and meet at one s_barrier per window (pacing only: it is what keeps the
// window L1-resident, not a data dependency).
struct LtArgs {
    const int32_t *words; const int32_t *stream_start; const int32_t *wsteps; const int32_t *tile_row0; const int32_t *n_win;
    const int32_t *vstart; const int32_t *vcount;
    int maxwin1; int cbits;
    int pace_mask;                               // the waves meet at a barrier after every window w with (w & pace_mask) == pace_mask
    const float *X; int64_t ldx; const float *Xself; const float *diag; const float *row_scale;
    SpmmArgs e;
    // GAT mode (amar_gat_lt_f32): per-node attention scalars over the image's columns, the upper bound of s_neigh, the rows' own
    // scalars (row i of the block = column row_offset + i), the block's CSR for the exact per-row fall-back
    const float *s_neigh; const float *bmax; const float *s_self_rows; const float *s_neigh_rows;
    const int32_t *csr_rowptr; const int32_t *csr_colidx; int self_loop;
};

#ifdef AMAR_LT_STAMPS                                  // development build only (tools/exp_lt_stamps.py): per-tile cycle stamps
__device__ unsigned long long lt_debug_stamps[4 * 8192];
#define LT_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192) lt_debug_stamps[4 * blockIdx.x + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define LT_STAMP(i) do { } while (0)
#endif
constexpr int LT_WAVES = 16;
#ifndef AMAR_LT_TILE_BYTES                          // development builds (tools/exp_lt_half.py): a smaller Y tile, several workgroups per CU
#define AMAR_LT_TILE_BYTES (128 << 10)
#endif
#ifndef AMAR_LT_MIN_WAVES                           // ... and the waves per SIMD the register allocation must leave room for
#define AMAR_LT_MIN_WAVES 4
#endif
constexpr int LT_TILE_BYTES = AMAR_LT_TILE_BYTES;
// GAT mode keeps (sum of weights, s_self) next to every LDS row: 4F + 8 bytes per virtual row, fewer rows per wave
// (utilities/lds_tiled.py:GAT_ROWS_PER_WAVE holds the same table)
constexpr int lt_gat_rw(int F) { return F == 8 ? 216 : F == 16 ? 124 : F == 32 ? 64 : 384; }
constexpr int lt_bits(int n) { int b = 0; while ((1 << b) < n) ++b; return b; }
constexpr int LT_CHUNK = 256;                      // entries per index chunk: 64 lanes x one 16-byte load
#ifndef LT_PREFETCH
#define LT_PREFETCH 2                              // index chunks in flight per wave ahead of the one being issued
#endif

// Virtual row v of a tile belongs to wave v % 16; its LDS row is WAVE-MAJOR (round 4): wave * RW + v / 16.  A step's address is then
// one v_bfe_u32 + one v_lshl_add_u32 on a per-lane constant; rounds 2-3 interleaved the waves' rows with a rotation
// (l * 16 + (w + l) % 16: five instructions per step).  Bank-wise the two are alike: a wave's rows of a step are random either way.
#ifndef AMAR_LT_WAVE_MAJOR
#define AMAR_LT_WAVE_MAJOR 1
#endif
template <int RW>
__device__ __forceinline__ int lt_lds_row(int v) {                    // virtual row -> row of the LDS tile (see lds_tiled.py)
    const int w = v & (LT_WAVES - 1), l = v / LT_WAVES;
    return AMAR_LT_WAVE_MAJOR ? w * RW + l : l * LT_WAVES + ((w + l) & (LT_WAVES - 1));
}

// ABL (development, tools/exp_lt.py): 1 = no atomic path, 2 = no LDS read-add-write, 4 = no gathers (timing only: wrong sums)
// PACE: 0 = waves run free, 1 = one s_barrier per (pace_mask + 1) windows, 2 = arrival counters in LDS: a wave leaves window w
// once every wave has left window w - 1 (one window of slack: measured slower than the barrier)
// OFF32: 0 = 64-bit gather addresses, 1 = 32-bit byte offsets, 2 = 32-bit offsets into a dense table (ldx == F: a shift, no multiply)
// GAT: the attention layer on the same walk (amar_gat_lt_f32).  An entry (r, c) weighs w = exp(e_rc - M_r), e_rc =
// LeakyReLU_0.2(s_self[r] + s_neigh[c]), against the row's fixed bound M_r = LeakyReLU(s_self[r] + max_j s_neigh[j]) >= max_c e_rc:
// weights are then ADDITIVE (no running maximum), so virtual rows, in-register pairs and flagged repeats work as for the plain
// sum; the LDS row carries (sum w.h, sum w, s_self[r]).  The softmax is invariant to the choice of M_r as long as nothing
// underflows: a row whose weight sum stays below e^-60 is recomputed in the epilogue from the block's CSR with its true maximum.
// SAGE: GraphSAGE's tail in the epilogue (AMAR_SPMM_SAGE_TAIL): the tile's sums are the mean aggregate; the row leaves as
// relu(l2_normalize([x_i || agg_i] . W + b)) with W = e.Wn [2F, F], in sage_tail_kernel's order of operations.
// PAIRS = false (AMAR_SPMM_LT_NOPAIRS: the image flags EVERY repeat of a row inside a step, utilities/lds_tiled.py `pairs=False`):
// the implicit-pair logic — one DPP read of the previous slot's row, two compares, four selects and four DPP adds per step — is
// compiled out.  Wide rows are few entries per wave-instruction (F = 32: 8), so a step's fixed instructions weigh 4x what they
// do at F = 8 (rocprofv3, ml1m(s=64): 32 VALU instructions per 8 entries, the VALU 70 % busy) while repeats inside a step are
// rare (0.14 % pairs at F = 32): they go to the atomic path instead.
// (Tried on top of it and not kept, profiles/r3_exp_lt_two_quads.txt: TWO float4 per lane — F / 8 lanes per entry, twice the entries
// per step — 0.45 / 0.61 ms per ml1m(s=64) product at F = 16 / 32 against 0.30 / 0.485: the LDS read-add-write then moves 32
// contiguous bytes per entry and instruction instead of 64 / 128 and conflicts more than the saved index arithmetic is worth.)
template <int F, int OFF32, bool FUSE_NEXT, int U, int PACE, int ABL = 0, bool GAT = false, bool SAGE = false, bool PAIRS = true>
__global__ __launch_bounds__(LT_WAVES * AMAR_WAVE, AMAR_LT_MIN_WAVES) void spmm_lt_kernel(const LtArgs a) {
    constexpr int LPN = F / 4, EPS = AMAR_WAVE / LPN, RW = GAT ? lt_gat_rw(F) : LT_TILE_BYTES / (4 * F * LT_WAVES), CS = LT_CHUNK / EPS;
    constexpr int G = U - 1;                                          // steps of gathers in flight ahead of the accumulation
    static_assert(CS % U == 0 && G < CS, "register slots of the in-flight steps must be static inside a chunk");
    extern __shared__ __attribute__((aligned(16))) float lt_lds[];
    float *ytile = lt_lds;                                            // [RW * LT_WAVES][F]
    float2 *side = reinterpret_cast<float2 *>(lt_lds + LT_WAVES * RW * F);          // GAT: [RW * LT_WAVES] (sum of weights, s_self)
    int32_t *ring_all = reinterpret_cast<int32_t *>(lt_lds + LT_WAVES * RW * (GAT ? F + 2 : F));   // [LT_WAVES][LT_CHUNK]
    unsigned *arrived = reinterpret_cast<unsigned *>(ring_all + LT_WAVES * LT_CHUNK);   // [8] arrival counters (PACE 2)
    const int t = blockIdx.x;
    const int lane = threadIdx.x & (AMAR_WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    LT_STAMP(0);
    const int r0 = a.tile_row0[t], nr = a.tile_row0[t + 1] - r0;
    for (int i = threadIdx.x; i < LT_WAVES * RW * (GAT ? F + 2 : F) / 4; i += LT_WAVES * AMAR_WAVE)
        reinterpret_cast<float4 *>(ytile)[i] = f4_zero();
    if (threadIdx.x < 8) arrived[threadIdx.x] = 0;
    __syncthreads();
    float bmax = 0.f;
    if (GAT) {                                                        // every virtual row carries its row's s_self
        bmax = *a.bmax;
        const int vt = a.vcount[t];
        for (int lr = threadIdx.x; lr < nr; lr += LT_WAVES * AMAR_WAVE) {
            const int row = r0 + lr;
            const float as = a.s_self_rows[row];
            const int v0 = a.vstart[row], v1 = lr + 1 < nr ? a.vstart[row + 1] : vt;
            for (int v = v0; v < v1; ++v) side[lt_lds_row<RW>(v)].y = as;
        }
        __syncthreads();
    }

    const int s = lane / LPN, q = lane % LPN;                         // LPN adjacent lanes share an entry
    const int32_t *stream = a.words + a.stream_start[t * LT_WAVES + wave];
    const int32_t *ws = a.wsteps + ((int64_t)t * LT_WAVES + wave) * a.maxwin1;     // the wave's window table [maxwin1]
    const int nwin = a.n_win[t];
    // entry n_win of the window table: the steps that hold entries of the stream (round 4: exact; the stream itself stays padded to
    // whole chunks, but the steps of the padding are not walked — a 190-row tile of an 8-rank block has ~860 entries per wave, 1 024 padded)
    const int n_steps = __builtin_amdgcn_readfirstlane(ws[nwin]);
    const int n_chunks = (n_steps + CS - 1) / CS;
    int32_t *ring = ring_all + wave * LT_CHUNK;
    const unsigned cmask = (1u << a.cbits) - 1u;

    typedef int v4i __attribute__((ext_vector_type(4)));
    auto load_chunk = [&](int c) {                                    // read-once stream: non-temporal
        return __builtin_nontemporal_load(reinterpret_cast<const v4i *>(stream + (int64_t)c * LT_CHUNK) + lane);
    };
    int wd[U], wn[CS];                                                // words of the steps in flight / of the chunk being issued
    float4 x[U];
    float bs[GAT ? U : 1];                                            // GAT: s_neigh of the steps in flight
    auto refill = [&](const v4i &pre) {                               // chunk registers -> LDS -> one word per (step, entry slot)
        *reinterpret_cast<v4i *>(ring + 4 * lane) = pre;
#pragma unroll
        for (int j = 0; j < CS; ++j) wn[j] = ring[j * EPS + s];
    };
    auto issue = [&](int ks, int slot) {                              // ks, slot: compile-time after unrolling
        const int w = wn[ks];
        wd[slot] = w;
        const unsigned col = (unsigned)w & cmask;
        if (ABL & 4) { const float v = __builtin_bit_cast(float, col); x[slot] = make_float4(v, v, v, v); }
        else if (OFF32 == 2) x[slot] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(a.X) + ((col * (unsigned)F + 4u * q) * 4u));
        else if (OFF32 == 1) x[slot] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(a.X) + (col * (unsigned)a.ldx + 4u * q) * 4u);
        else x[slot] = *reinterpret_cast<const float4 *>(a.X + (int64_t)col * a.ldx + 4 * q);
        if (GAT) bs[slot] = (ABL & 8) ? 0.25f : *reinterpret_cast<const float *>(reinterpret_cast<const char *>(a.s_neigh) + (uint64_t)col * 4u);
    };
    auto accumulate = [&](int slot) {
        const int w = wd[slot];
        const int lrow = (int)__builtin_amdgcn_ubfe((unsigned)w, (unsigned)a.cbits, (unsigned)lt_bits(RW));   // (w >> cbits) & LMASK as one v_bfe_u32
        const int lds_row = AMAR_LT_WAVE_MAJOR ? wave * RW + lrow : lrow * LT_WAVES + ((wave + lrow) & (LT_WAVES - 1));
        float *yp = ytile + lds_row * F + 4 * q;
        float4 xv = x[slot];
        if (ABL & 2) {                                                // keep the gathers alive without LDS traffic
            if (xv.x == 123.f && xv.y == 4.f) *reinterpret_cast<float4 *>(yp) = xv;
            return;
        }
        float wgt = 0.f, lsum = 0.f;
        if (GAT) {
            const float2 sd = (ABL & 16) ? make_float2(0.f, 0.5f) : side[lds_row];   // (sum of weights so far, s_self of the row)
            lsum = sd.x;
            const float z = sd.y + bs[slot], zz = sd.y + bmax;
            wgt = (ABL & 32) ? z * 0.01f : __expf(fmaxf(z, 0.2f * z) - fmaxf(zz, 0.2f * zz));
            xv.x *= wgt; xv.y *= wgt; xv.z *= wgt; xv.w *= wgt;
        }
        // implicit pair: same virtual row as the previous slot (row_shr: lane l reads l - LPN inside its 16-lane DPP row;
        // the first slot of a DPP row keeps -1) and not flagged -> its values go to that slot's registers, no LDS update
        bool paired = false;
        if constexpr (PAIRS) {
            const int prev = __builtin_amdgcn_update_dpp(-1, lrow, 0x110 + LPN, 0xF, 0xF, false);
            paired = prev == lrow && w >= 0;
            const float4 give = paired ? xv : f4_zero();
            xv.x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, give.x), 0x100 + LPN, 0xF, 0xF, true));
            xv.y += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, give.y), 0x100 + LPN, 0xF, 0xF, true));
            xv.z += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, give.z), 0x100 + LPN, 0xF, 0xF, true));
            xv.w += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, give.w), 0x100 + LPN, 0xF, 0xF, true));
            if (GAT) {
                const float gw = paired ? wgt : 0.f;
                wgt += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, gw), 0x100 + LPN, 0xF, 0xF, true));
            }
        }
        if (w >= 0 && !paired) {
            float4 y = *reinterpret_cast<float4 *>(yp);
            y = f4_add(y, xv);
            *reinterpret_cast<float4 *>(yp) = y;
            if (GAT && !(ABL & 16)) side[lds_row].x = lsum + wgt;     // the LPN lanes of the entry store the same value
        }
        if (!(ABL & 1) && w < 0) {                                    // the row occurs earlier in this step: after its plain add
            atomicAdd(yp + 0, xv.x); atomicAdd(yp + 1, xv.y); atomicAdd(yp + 2, xv.z); atomicAdd(yp + 3, xv.w);
            if (GAT && q == 0) atomicAdd(&side[lds_row].x, wgt);
        }
    };

    // window ends: lane l of `wtab` holds ws[wbase + l]; the end of window `win` is entry win + 1.  The reload (once per
    // 64 windows) is inline asm with its own wait: a compiler-visible load in this rarely taken branch would make every
    // later v_readlane wait for vmcnt(0), i.e. drain the gathers in flight at every window.
    int wbase = 0;
    int wtab;
    auto load_wtab = [&]() {
        const int32_t *p = ws + min(wbase + lane, nwin);
        asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(wtab) : "v"(p) : "memory");
    };
    load_wtab();
    int win = 0;
    auto window_end = [&](int w) {                                    // w < nwin, wave-uniform
        if (w + 1 >= wbase + AMAR_WAVE) { wbase = w + 1; load_wtab(); }
        return __builtin_amdgcn_readlane(wtab, w + 1 - wbase);
    };
    int wend = nwin > 0 ? window_end(0) : 0x7fffffff;
    auto pace = [&](int done) {                                       // `done` steps finished: leave every window that ends here
        while (wend <= done) {
            if (PACE == 1 && (win & a.pace_mask) == a.pace_mask) __builtin_amdgcn_s_barrier();
            if (PACE == 2) {
                if (lane == 0) atomicAdd(arrived + (win & 7), 1u);
                if (win > 0) {
                    const unsigned need = (unsigned)LT_WAVES * ((unsigned)(win - 1) / 8u + 1u);
                    volatile unsigned *slot = arrived + ((win - 1) & 7);
                    while (__builtin_amdgcn_readfirstlane((int)*slot) < (int)need) __builtin_amdgcn_s_sleep(1);
                }
            }
            ++win;
            wend = win < nwin ? window_end(win) : 0x7fffffff;
        }
    };
    LT_STAMP(1);
    pace(0);
    // index chunks are requested PF chunks ahead, in turn into PF register sets: `next` holds chunk c + 1 when chunk c reaches
    // the point where it starts issuing that chunk's steps, and is then reloaded with chunk c + 1 + PF
    constexpr int PF = LT_PREFETCH;
    v4i pre[PF];
#pragma unroll
    for (int j = 0; j < PF; ++j) pre[j] = v4i{0, 0, 0, 0};
    auto chunk_body = [&](int c, v4i &next) {
#pragma unroll
        for (int js = 0; js < CS; ++js) {
            if (js + G == CS) {                                       // the steps issued from here on belong to the next chunk
                refill(next);                                         // (past the last chunk: stale words, valid columns, never added)
                if (c + 1 + PF < n_chunks) next = load_chunk(c + 1 + PF);
            }
            issue((js + G) % CS, (js + G) % U);                       // (unconditional: the counted waits rely on a static load sequence)
            if (c * CS + js < n_steps) accumulate(js % U);
            pace(c * CS + js + 1);
        }
    };
    if (n_chunks > 0) {
        pre[0] = load_chunk(0);
        refill(pre[0]);
#pragma unroll
        for (int j = 1; j <= PF; ++j)                                 // chunk j waits in set j % PF
            if (j < n_chunks) pre[j % PF] = load_chunk(j);
#pragma unroll
        for (int j = 0; j < G; ++j) issue(j, j);
    }
    for (int c = 0; c < n_chunks; c += PF) {
#pragma unroll
        for (int j = 0; j < PF; ++j)
            if (c + j < n_chunks) chunk_body(c + j, pre[(j + 1) % PF]);      // chunk k waits in set k % PF (c is a multiple of PF)
    }
    __syncthreads();

    // epilogue: y_i = row_scale_i . (diag_i . x_i + the row's virtual rows), then bias / ReLU / store / running sum / next X.W
    LT_STAMP(2);
    // Wide layers keep the next layer's kernel in the (now idle) index ring: one thread finishes one row, and F x Cn weights by
    // wave-uniform scalar loads are F x Cn / 4 dependent s_load round trips per wave with nothing left to hide them — the fused
    // next X.W cost 0.27 ms per ml1m(s=64) layer at F = 32 (0.57 -> 0.84) and 0.05 at F = 16; broadcast ds_read_b128 instead.
    constexpr bool WN_LDS = FUSE_NEXT && !SAGE && !GAT && F >= 16;
    float *wn_lds = reinterpret_cast<float *>(ring_all);              // [F][Cn], Cn <= 64: at most 8 KB of the 16 KB ring
    if constexpr (WN_LDS) {
        for (int i = threadIdx.x; i < F * a.e.Cn; i += LT_WAVES * AMAR_WAVE) wn_lds[i] = a.e.Wn[i];
        __syncthreads();
    }
    // A thread finishes up to EK rows per pass; their global operands (virtual-row range, diag, scale, the row's own X) are all
    // requested before the first one is used: one memory round trip per pass instead of one per row (a 4 080-row tile of
    // four-entry rows spent a quarter of its time in these).
    const int vtile = a.vcount[t];
    constexpr int EK = SAGE ? 1 : F <= 8 ? 4 : F == 16 ? 2 : 1, ETH = LT_WAVES * AMAR_WAVE;   // (the SAGE tail needs the registers itself)
    // Short tiles (round 4: the per-type row blocks of an 8-rank partition are 100-190 rows of up to ten virtual rows each): with one
    // thread per row nine tenths of the workgroup idle while a hundred threads walk their virtual rows one LDS round trip at a time —
    // 14 k of a 70 k-cycle tile.  2, 4, 8 or 16 lanes share a row instead (as many as the workgroup has for the tile's rows), each
    // summing every G-th virtual row, folded with a butterfly; the group's first lane finishes the row.  (The sums of a row are then
    // taken in another order than in a tall tile: fixed by the tile's height, so a launch stays reproducible bit for bit.)
    if constexpr (!GAT && !SAGE) {
        const int lg = nr * 16 <= ETH ? 4 : nr * 8 <= ETH ? 3 : nr * 4 <= ETH ? 2 : nr * 2 <= ETH ? 1 : 0;
        if (lg > 0) {
            const int lr = (int)threadIdx.x >> lg, g = (int)threadIdx.x & ((1 << lg) - 1);
            const bool live = lr < nr;
            const int row = r0 + (live ? lr : 0);
            const int v0 = a.vstart[row], v1 = !live ? v0 : (lr + 1 < nr ? a.vstart[row + 1] : vtile);
            const float d = a.diag[row], sc = a.row_scale[row];
            float4 xs[LPN], acc[LPN];
#pragma unroll
            for (int qq = 0; qq < LPN; ++qq) {
                xs[qq] = *reinterpret_cast<const float4 *>(a.Xself + (int64_t)row * a.e.ldx + 4 * qq);
                acc[qq] = f4_zero();
            }
            for (int v = v0 + g; v < v1; v += 1 << lg) {
                const float *yp = ytile + lt_lds_row<RW>(v) * F;
#pragma unroll
                for (int qq = 0; qq < LPN; ++qq) acc[qq] = f4_add(acc[qq], *reinterpret_cast<const float4 *>(yp + 4 * qq));
            }
            for (int m = 1; m < (1 << lg); m <<= 1)
#pragma unroll
                for (int qq = 0; qq < LPN; ++qq) {
                    acc[qq].x += __shfl_xor(acc[qq].x, m, AMAR_WAVE); acc[qq].y += __shfl_xor(acc[qq].y, m, AMAR_WAVE);
                    acc[qq].z += __shfl_xor(acc[qq].z, m, AMAR_WAVE); acc[qq].w += __shfl_xor(acc[qq].w, m, AMAR_WAVE);
                }
            if (live && g == 0) {
#pragma unroll
                for (int qq = 0; qq < LPN; ++qq) {
                    acc[qq].x = sc * fmaf(d, xs[qq].x, acc[qq].x); acc[qq].y = sc * fmaf(d, xs[qq].y, acc[qq].y);
                    acc[qq].z = sc * fmaf(d, xs[qq].z, acc[qq].z); acc[qq].w = sc * fmaf(d, xs[qq].w, acc[qq].w);
                }
                lane_row_epilogue<F, FUSE_NEXT>(a.e, row, acc, WN_LDS ? wn_lds : nullptr);
            }
#ifdef AMAR_LT_STAMPS
            __syncthreads();
            LT_STAMP(3);
#endif
            return;
        }
    }
    for (int base = threadIdx.x; base < nr; base += EK * ETH) {
    int ev0[EK], ev1[EK];
    float ed[EK], esc[EK], eas[GAT ? EK : 1], ebs[GAT ? EK : 1];
    float4 exs[EK][LPN];
#pragma unroll
    for (int k = 0; k < EK; ++k) {
        const int lr = base + k * ETH;
        if (lr < nr) {
            const int row = r0 + lr;
            ev0[k] = a.vstart[row]; ev1[k] = lr + 1 < nr ? a.vstart[row + 1] : vtile;
            ed[k] = a.diag[row];
            esc[k] = GAT ? 0.f : a.row_scale[row];
            if (GAT) { eas[k] = a.s_self_rows[row]; ebs[k] = a.s_neigh_rows[row]; }
#pragma unroll
            for (int qq = 0; qq < LPN; ++qq) exs[k][qq] = *reinterpret_cast<const float4 *>(a.Xself + (int64_t)row * a.e.ldx + 4 * qq);
        }
    }
#pragma unroll
    for (int k = 0; k < EK; ++k) {
        const int lr = base + k * ETH;
        if (lr >= nr) break;
        const int row = r0 + lr;
        const int v0 = ev0[k], v1 = ev1[k];
        const float d = ed[k];
        float4 acc[LPN];
        if (GAT) {
            // self term: the added self loop and any (i, i) edges of A itself, with the same bound as the walk
            const float as = eas[k], bself = ebs[k];
            const float zz = as + bmax, zs = as + bself;
            const float nself = d + (a.self_loop ? 1.f : 0.f);
            const float ws = nself > 0.f ? nself * __expf(fmaxf(zs, 0.2f * zs) - fmaxf(zz, 0.2f * zz)) : 0.f;
            float l = ws;
#pragma unroll
            for (int qq = 0; qq < LPN; ++qq) {
                const float4 xs = exs[k][qq];
                acc[qq] = make_float4(ws * xs.x, ws * xs.y, ws * xs.z, ws * xs.w);
            }
            for (int v = v0; v < v1; ++v) {
                const int lv = lt_lds_row<RW>(v);
                const float *yp = ytile + lv * F;
#pragma unroll
                for (int qq = 0; qq < LPN; ++qq) acc[qq] = f4_add(acc[qq], *reinterpret_cast<const float4 *>(yp + 4 * qq));
                l += side[lv].x;
            }
            float inv;
            if (l >= 8.7e-27f) inv = 1.f / l;                          // e^-60: the row's largest weight is a normal number
            else {
                // the bound is more than 60 above this row's own maximum (or the row is empty): Spektral's arithmetic on the
                // block's CSR row, with the row's true maximum (gat_row_kernel's two passes, one lane)
                const int beg = a.csr_rowptr[row], end = a.csr_rowptr[row + 1];
                float mn = a.self_loop ? bself : -INFINITY;
                for (int p = beg; p < end; ++p) mn = fmaxf(mn, a.s_neigh[a.csr_colidx[p]]);
                const float zm = as + mn, emax = fmaxf(zm, 0.2f * zm);
                l = 0.f;
#pragma unroll
                for (int qq = 0; qq < LPN; ++qq) acc[qq] = f4_zero();
                for (int p = beg; p <= end; ++p) {
                    if (p == end && !a.self_loop) break;
                    const int64_t c = p < end ? (int64_t)a.csr_colidx[p] : (int64_t)(a.s_neigh_rows - a.s_neigh) + row;
                    const float z = as + a.s_neigh[c];
                    const float wv = __expf(fmaxf(z, 0.2f * z) - emax);
                    l += wv;
#pragma unroll
                    for (int qq = 0; qq < LPN; ++qq) acc[qq] = f4_fma(wv, *reinterpret_cast<const float4 *>(a.X + c * a.ldx + 4 * qq), acc[qq]);
                }
                inv = 1.f / (l + 1e-9f);
            }
#pragma unroll
            for (int qq = 0; qq < LPN; ++qq) { acc[qq].x *= inv; acc[qq].y *= inv; acc[qq].z *= inv; acc[qq].w *= inv; }
        } else {
            const float sc = esc[k];
            float4 xself[SAGE ? LPN : 1];
#pragma unroll
            for (int qq = 0; qq < LPN; ++qq) {
                const float4 xs = exs[k][qq];
                if (SAGE) xself[qq] = xs;
                acc[qq] = make_float4(d * xs.x, d * xs.y, d * xs.z, d * xs.w);
            }
            for (int v = v0; v < v1; ++v) {
                const float *yp = ytile + lt_lds_row<RW>(v) * F;
#pragma unroll
                for (int qq = 0; qq < LPN; ++qq) acc[qq] = f4_add(acc[qq], *reinterpret_cast<const float4 *>(yp + 4 * qq));
            }
#pragma unroll
            for (int qq = 0; qq < LPN; ++qq) { acc[qq].x *= sc; acc[qq].y *= sc; acc[qq].z *= sc; acc[qq].w *= sc; }
            if (SAGE) {
                float z[F];
#pragma unroll
                for (int c = 0; c < F; ++c) z[c] = 0.f;
#pragma unroll
                for (int half = 0; half < 2; ++half)
#pragma unroll
                    for (int qq = 0; qq < LPN; ++qq)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float v = f4_get(half == 0 ? xself[qq] : acc[qq], j);
                            const float *w = a.e.Wn + (int64_t)(half * F + 4 * qq + j) * F;      // wave-uniform: scalar loads (staged in LDS: slower)
#pragma unroll
                            for (int c = 0; c < F; ++c) z[c] = fmaf(v, w[c], z[c]);
                        }
                float sq = 0.f;
#pragma unroll
                for (int c = 0; c < F; ++c) { z[c] += a.e.bias[c]; sq = fmaf(z[c], z[c], sq); }
                const float iv = rsqrtf(fmaxf(sq, 1e-12f));          // tf.nn.l2_normalize
                float *y = a.e.Y + (int64_t)row * a.e.ldy;
                float *y2 = a.e.Hn ? a.e.Hn + (int64_t)row * a.e.ldhn : nullptr;     // a dense copy for the next layer's gathers
#pragma unroll
                for (int c4 = 0; c4 < F; c4 += 4) {
                    const float4 o = make_float4(fmaxf(z[c4] * iv, 0.f), fmaxf(z[c4 + 1] * iv, 0.f), fmaxf(z[c4 + 2] * iv, 0.f), fmaxf(z[c4 + 3] * iv, 0.f));
                    *reinterpret_cast<float4 *>(y + c4) = o;
                    if (y2) *reinterpret_cast<float4 *>(y2 + c4) = o;
                }
                continue;
            }
        }
        lane_row_epilogue<F, FUSE_NEXT>(a.e, row, acc, WN_LDS ? wn_lds : nullptr);
    }
    }
#ifdef AMAR_LT_STAMPS
    __syncthreads();
    LT_STAMP(3);
#endif
}

template <int F>
int launch_spmm_lt(const LtArgs &a, int n_tiles, int off32, bool fuse, int variant, bool sage, bool nopairs, hipStream_t st) {
    constexpr int RW = LT_TILE_BYTES / (4 * F * LT_WAVES);
    const size_t lds = (size_t)LT_WAVES * RW * F * 4 + (size_t)LT_WAVES * LT_CHUNK * 4 + 32;
    const dim3 grid((unsigned)n_tiles), block(LT_WAVES * AMAR_WAVE);
#define AMAR_LT_LAUNCH_S(OFF, FUSE, UU, PP, AA, SS) AMAR_LT_LAUNCH_P(OFF, FUSE, UU, PP, AA, SS, true)
#define AMAR_LT_LAUNCH_P(OFF, FUSE, UU, PP, AA, SS, PR)                                                                 \
    do {                                                                                                                 \
        auto kern = spmm_lt_kernel<F, OFF, FUSE, UU, PP, AA, false, SS, PR>;                                             \
        static bool allowed[AMAR_MAX_DEVICES] = {};                                                                      \
        if (const int rc = amar_allow_lds(reinterpret_cast<const void *>(kern), lds, allowed)) return rc;               \
        hipLaunchKernelGGL(kern, grid, block, lds, st, a);                                                               \
    } while (0)
#define AMAR_LT_LAUNCH(OFF, FUSE, UU, PP, AA) AMAR_LT_LAUNCH_S(OFF, FUSE, UU, PP, AA, false)
    if constexpr (F >= 8) {
        if (sage) {                                                   // GraphSAGE tail fused (the layer's input is a dense table or a concat slice)
            if (off32 == 2) AMAR_LT_LAUNCH_S(2, false, 4, 1, 0, true); else if (off32 == 1) AMAR_LT_LAUNCH_S(1, false, 4, 1, 0, true);
            else return AMAR_EUNSUPPORTED;
            return amar_check_launch();
        }
    } else if (sage) return AMAR_EUNSUPPORTED;
    // variant (development, AMAR_LT_VARIANT; F = 8, dense table, no fused next layer only): see tools/exp_lt.py
    if constexpr (F == 8) {
        if (variant && off32 == 2 && !fuse) {
            switch (variant) {
            case 1:  AMAR_LT_LAUNCH(2, false, 2, 1, 0); break;      // 1 step ahead
            case 2:  AMAR_LT_LAUNCH(2, false, 4, 0, 0); break;      // unpaced
            case 3:  AMAR_LT_LAUNCH(2, false, 8, 1, 0); break;      // 7 steps ahead
            case 5:  AMAR_LT_LAUNCH(2, false, 4, 2, 0); break;      // counter pacing, one window of slack
            case 8:  AMAR_LT_LAUNCH(1, false, 4, 1, 0); break;      // multiply-add addressing
            // ablations (wrong sums; ABL bits: 1 no atomic path, 2 no LDS update, 4 no gathers): 20 + ABL paced, 30 + ABL unpaced
            case 21: AMAR_LT_LAUNCH(2, false, 4, 1, 1); break;
            case 22: AMAR_LT_LAUNCH(2, false, 4, 1, 2); break;
            case 23: AMAR_LT_LAUNCH(2, false, 4, 1, 3); break;
            case 24: AMAR_LT_LAUNCH(2, false, 4, 1, 4); break;
            case 25: AMAR_LT_LAUNCH(2, false, 4, 1, 5); break;
            case 27: AMAR_LT_LAUNCH(2, false, 4, 1, 7); break;
            case 31: AMAR_LT_LAUNCH(2, false, 4, 0, 1); break;
            case 33: AMAR_LT_LAUNCH(2, false, 4, 0, 3); break;
            case 35: AMAR_LT_LAUNCH(2, false, 4, 0, 5); break;
            case 37: AMAR_LT_LAUNCH(2, false, 4, 0, 7); break;
            default: return AMAR_EINVAL;
            }
            return amar_check_launch();
        }
    }
    if constexpr (F >= 8) {
        if (nopairs && off32 != 0) {                                  // an image without implicit pairs: the lean step (see PAIRS above)
            if constexpr (F == 8) {                                   // development variants of the pair-free F = 8 kernel (tools/exp_lt8.py)
                if (variant && off32 == 2 && !fuse) {
                    switch (variant) {
                    case 2:  AMAR_LT_LAUNCH_P(2, false, 4, 0, 0, false, false); break;      // unpaced
                    case 3:  AMAR_LT_LAUNCH_P(2, false, 8, 1, 0, false, false); break;      // 7 steps ahead
                    case 22: AMAR_LT_LAUNCH_P(2, false, 4, 1, 2, false, false); break;      // no LDS update (wrong sums)
                    case 24: AMAR_LT_LAUNCH_P(2, false, 4, 1, 4, false, false); break;      // no gathers, paced (wrong sums)
                    case 35: AMAR_LT_LAUNCH_P(2, false, 4, 0, 5, false, false); break;      // no gathers, no atomics, unpaced
                    default: return AMAR_EINVAL;
                    }
                    return amar_check_launch();
                }
            }
            static const int uenv = getenv("AMAR_LT_U") ? atoi(getenv("AMAR_LT_U")) : 0;      // development: steps of gathers in flight
            if (uenv == 8 && off32 == 2) {
                if (fuse) AMAR_LT_LAUNCH_P(2, true, 8, 1, 0, false, false); else AMAR_LT_LAUNCH_P(2, false, 8, 1, 0, false, false);
                return amar_check_launch();
            }
            if (off32 == 2) { if (fuse) AMAR_LT_LAUNCH_P(2, true, 4, 1, 0, false, false); else AMAR_LT_LAUNCH_P(2, false, 4, 1, 0, false, false); }
            else { if (fuse) AMAR_LT_LAUNCH_P(1, true, 4, 1, 0, false, false); else AMAR_LT_LAUNCH_P(1, false, 4, 1, 0, false, false); }
            return amar_check_launch();
        }
    }
    if (off32 == 2) { if (fuse) AMAR_LT_LAUNCH(2, true, 4, 1, 0); else AMAR_LT_LAUNCH(2, false, 4, 1, 0); }
    else if (off32 == 1) { if (fuse) AMAR_LT_LAUNCH(1, true, 4, 1, 0); else AMAR_LT_LAUNCH(1, false, 4, 1, 0); }
    else { if (fuse) AMAR_LT_LAUNCH(0, true, 4, 1, 0); else AMAR_LT_LAUNCH(0, false, 4, 1, 0); }
#undef AMAR_LT_LAUNCH
#undef AMAR_LT_LAUNCH_S
#undef AMAR_LT_LAUNCH_P
    return amar_check_launch();
}

template <int F>
int launch_gat_lt(const LtArgs &a, int n_tiles, int off32, hipStream_t st) {
    constexpr int RW = lt_gat_rw(F);
    const size_t lds = (size_t)LT_WAVES * RW * (F + 2) * 4 + (size_t)LT_WAVES * LT_CHUNK * 4 + 32;
    static_assert((size_t)LT_WAVES * RW * (F + 2) * 4 + (size_t)LT_WAVES * LT_CHUNK * 4 + 32 <= (160u << 10), "one workgroup's LDS");
    const dim3 grid((unsigned)n_tiles), block(LT_WAVES * AMAR_WAVE);
#define AMAR_GAT_LT_LAUNCH_A(OFF, AA)                                                                                    \
    do {                                                                                                                 \
        auto kern = spmm_lt_kernel<F, OFF, false, 4, 1, AA, true>;                                                       \
        static bool allowed[AMAR_MAX_DEVICES] = {};                                                                      \
        if (const int rc = amar_allow_lds(reinterpret_cast<const void *>(kern), lds, allowed)) return rc;               \
        hipLaunchKernelGGL(kern, grid, block, lds, st, a);                                                               \
    } while (0)
#define AMAR_GAT_LT_LAUNCH(OFF) AMAR_GAT_LT_LAUNCH_A(OFF, 0)
    if constexpr (F == 8) {                                           // development ablations (wrong results): tools/exp_gat_lt.py
        static const int abl = getenv("AMAR_GAT_ABL") ? atoi(getenv("AMAR_GAT_ABL")) : 0;
        if (abl && off32 == 2) {
            switch (abl) {
            case 8:  AMAR_GAT_LT_LAUNCH_A(2, 8); break;               // no s_neigh gather
            case 16: AMAR_GAT_LT_LAUNCH_A(2, 16); break;              // no side-array traffic
            case 24: AMAR_GAT_LT_LAUNCH_A(2, 24); break;
            case 32: AMAR_GAT_LT_LAUNCH_A(2, 32); break;              // no exp
            case 56: AMAR_GAT_LT_LAUNCH_A(2, 56); break;
            default: return AMAR_EINVAL;
            }
            return amar_check_launch();
        }
    }
    if (off32 == 2) AMAR_GAT_LT_LAUNCH(2); else if (off32 == 1) AMAR_GAT_LT_LAUNCH(1); else AMAR_GAT_LT_LAUNCH(0);
#undef AMAR_GAT_LT_LAUNCH
#undef AMAR_GAT_LT_LAUNCH_A
    return amar_check_launch();
}

// max over a vector, for the GAT bound: out is reset to -inf in-stream, blocks fold with the ordered-integer form of float max
__global__ __launch_bounds__(256) void colmax_kernel(const float *__restrict__ x, int64_t n, float *out) {
    float m = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, x[i]);
    m = wave_max_all(m);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
        if (m >= 0.f) atomicMax(reinterpret_cast<int *>(out), __builtin_bit_cast(int, m));      // non-negative floats order like ints,
        else atomicMin(reinterpret_cast<unsigned *>(out), __builtin_bit_cast(unsigned, m));     // negative ones like reversed unsigneds
    }
}

bool ld_ok(int64_t ld, int F) { return ld >= F && (ld & 3) == 0; }

// The row kernels are instantiated for widths 4, 8, 16, 32, 64.  Every operation below that is separable by feature
// column (SpMM and its bias / ReLU / running-sum epilogue, the GAT aggregation once the per-node attention scalars
// exist) runs any other multiple of 4 as a sequence of column chunks of those widths on strided views: 24 = 16 + 8,
// 48 = 32 + 16, 96 = 64 + 32 (TwoStep / TwoWay stacks with a 'concatenation' hand-over, tsgnn.py:65-75).
bool native_width(int F) { return F == 4 || F == 8 || F == 16 || F == 32 || F == 64; }
int chunk_width(int remaining) { return remaining >= 64 ? 64 : remaining >= 32 ? 32 : remaining >= 16 ? 16 : remaining >= 8 ? 8 : 4; }
const float *at(const float *p, int o) { return p ? p + o : nullptr; }
float *at(float *p, int o) { return p ? p + o : nullptr; }

// ---- row-wise X.W prologue -------------------------------------------------------------------
struct XwArgs {
    const float *X; int64_t ldx; int F; const float *W; int C;
    float *H; int64_t ldh; float *copy_to; int64_t ld_copy;
    const float *a_self; const float *a_neigh; float *s_self; float *s_neigh;
    int n_rows;
    const float *row_scale;                      // H rows are multiplied by row_scale[row] (value-free XS image), or NULL
    const int32_t *row_ids;                      // output row p reads X[row_ids[p]] (a negative id: a zero row), or NULL: X[p]
};

// One wave handles 64/CP rows at a time (CP = C rounded up to a power of two): lane = (row slot, out column).
__global__ __launch_bounds__(256) void rowwise_xw_kernel(const XwArgs a, int CP) {
    extern __shared__ float w_lds[];                   // W[F][C]
    for (int i = threadIdx.x; i < a.F * a.C; i += blockDim.x) w_lds[i] = a.W[i];
    __syncthreads();
    const int rows_per_block = blockDim.x / CP;
    const int c = threadIdx.x % CP, rslot = threadIdx.x / CP;
    for (int64_t row = (int64_t)blockIdx.x * rows_per_block + rslot; row < a.n_rows;
         row += (int64_t)gridDim.x * rows_per_block) {
        const int64_t src = a.row_ids ? (int64_t)a.row_ids[row] : row;
        const float *x = a.X + (src < 0 ? 0 : src) * a.ldx;
        float h = 0.f;
        if (c < a.C) {
            for (int k = 0; k < a.F; ++k) h = fmaf(x[k], w_lds[k * a.C + c], h);
            if (src < 0) h = 0.f;
            a.H[row * a.ldh + c] = a.row_scale ? h * a.row_scale[row] : h;
        }
        if (a.copy_to) for (int k = c; k < a.F; k += CP) a.copy_to[row * a.ld_copy + k] = x[k];
        if (a.s_self) {
            float ps = c < a.C ? h * a.a_self[c] : 0.f, pn = c < a.C ? h * a.a_neigh[c] : 0.f;
            for (int off = CP / 2; off >= 1; off >>= 1) { ps += __shfl_xor(ps, off, 64); pn += __shfl_xor(pn, off, 64); }
            if (c == 0) { a.s_self[row] = ps; a.s_neigh[row] = pn; }
        }
    }
}

// One thread per row for the square widths the GNN stacks use (F = C = 8 or 16, float4-aligned operands): the row arrives in
// F/4 16-byte loads and leaves in as many 16-byte stores per destination (H, the slice copy), the weights sit in LDS and are read
// as broadcasts.  The generic kernel above moves every element with its own dword instruction — C lanes per row re-reading the
// row — and ran at 2.2 TB/s of traffic (33 us over the 768 k padded rows of an 8-rank partition, 22 us at ml1m(s=64)).
template <int F>
__global__ __launch_bounds__(256) void rowwise_xw_vec_kernel(const XwArgs a) {
    __shared__ __attribute__((aligned(16))) float w_lds[F * F + 2 * F];
    for (int i = threadIdx.x; i < F * F; i += 256) w_lds[i] = a.W[i];
    if (a.s_self) for (int i = threadIdx.x; i < F; i += 256) { w_lds[F * F + i] = a.a_self[i]; w_lds[F * F + F + i] = a.a_neigh[i]; }
    __syncthreads();
    // R rows per thread, all of their loads requested before the first FMA (F = 8: one row per thread left 9 228 waves of 2 KB for
    // 8 192 wave slots at ml1m(s=64), a second, nearly empty round — 20.5 -> 15.9 us; F = 16: two rows per thread are slower, 37 -> 46)
    constexpr int R = F == 8 ? 2 : 1;
    for (int64_t row0 = (int64_t)blockIdx.x * 256 * R + threadIdx.x; row0 < a.n_rows; row0 += (int64_t)gridDim.x * 256 * R) {
        float x[R][F], sc[R];
        bool pad[R];
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int64_t row = row0 + 256 * j < a.n_rows ? row0 + 256 * j : row0;        // (past the end: the first row again, not stored)
            const int64_t src = a.row_ids ? (int64_t)a.row_ids[row] : row;
            pad[j] = src < 0;
            const int64_t xrow = pad[j] ? 0 : src;
#pragma unroll
            for (int q = 0; q < F / 4; ++q) {
                const float4 v = *reinterpret_cast<const float4 *>(a.X + xrow * a.ldx + 4 * q);
                x[j][4 * q] = v.x; x[j][4 * q + 1] = v.y; x[j][4 * q + 2] = v.z; x[j][4 * q + 3] = v.w;
            }
            sc[j] = pad[j] ? 0.f : (a.row_scale ? a.row_scale[row] : 1.f);
        }
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int64_t row = row0 + 256 * j;
            if (row >= a.n_rows) break;
            float h[F];
            if (a.copy_to) {
#pragma unroll
                for (int q = 0; q < F / 4; ++q)
                    *reinterpret_cast<float4 *>(a.copy_to + row * a.ld_copy + 4 * q) = make_float4(x[j][4 * q], x[j][4 * q + 1], x[j][4 * q + 2], x[j][4 * q + 3]);
            }
#pragma unroll
            for (int c = 0; c < F; ++c) h[c] = 0.f;
#pragma unroll
            for (int k = 0; k < F; ++k)                               // same order of operations per output as the generic kernel: k ascending
#pragma unroll
                for (int c4 = 0; c4 < F; c4 += 4) {
                    // F = 8: the 64 weights by wave-uniform (scalar) loads straight into the FMAs' operands — 16 LDS broadcasts per
                    // row made this form slower than the generic one on a dense 590 k-row table (25 us against 22)
                    // (read through the constant address space: the kernel never writes W, and only then does the compiler keep the
                    // loads scalar across the loop's stores)
                    float4 w;
                    if (F == 8) {
                        const __attribute__((address_space(4))) float *wc = (const __attribute__((address_space(4))) float *)a.W;
                        w = make_float4(wc[k * F + c4], wc[k * F + c4 + 1], wc[k * F + c4 + 2], wc[k * F + c4 + 3]);
                    } else w = *reinterpret_cast<const float4 *>(&w_lds[k * F + c4]);
                    h[c4] = fmaf(x[j][k], w.x, h[c4]); h[c4 + 1] = fmaf(x[j][k], w.y, h[c4 + 1]);
                    h[c4 + 2] = fmaf(x[j][k], w.z, h[c4 + 2]); h[c4 + 3] = fmaf(x[j][k], w.w, h[c4 + 3]);
                }
#pragma unroll
            for (int q = 0; q < F / 4; ++q)
                *reinterpret_cast<float4 *>(a.H + row * a.ldh + 4 * q) = (a.row_scale || a.row_ids)
                    ? make_float4(h[4 * q] * sc[j], h[4 * q + 1] * sc[j], h[4 * q + 2] * sc[j], h[4 * q + 3] * sc[j])
                    : make_float4(h[4 * q], h[4 * q + 1], h[4 * q + 2], h[4 * q + 3]);
            if (a.s_self) {                                           // (the generic kernel adds these by an xor butterfly: last-bit differences)
                float ps = 0.f, pn = 0.f;
#pragma unroll
                for (int c = 0; c < F; ++c) { ps = fmaf(h[c], w_lds[F * F + c], ps); pn = fmaf(h[c], w_lds[F * F + F + c], pn); }
                a.s_self[row] = ps; a.s_neigh[row] = pn;
            }
        }
    }
}

// ---- GraphSAGE (mean) ------------------------------------------------------------------------
struct SageArgs {
    const int32_t *rowptr; const int32_t *colidx; const float *X; int64_t ldx;
    const float *W; const float *bias; int C; float *Y; int64_t ldy; int self_loop; int n_rows;
};

template <int F>
__global__ __launch_bounds__(WAVES_PER_BLOCK * AMAR_WAVE) void sage_row_kernel(const SageArgs a) {
    constexpr int LPN = F / 4;
    const int lane = threadIdx.x & (AMAR_WAVE - 1);
    const int row = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= a.n_rows) return;
    const int q = lane % LPN, slot = lane / LPN;

    float wx[F], wa[F];                                // column `lane` of W[0:F] (self) and W[F:2F] (aggregate)
#pragma unroll
    for (int k = 0; k < F; ++k) {
        wx[k] = lane < a.C ? a.W[k * a.C + lane] : 0.f;
        wa[k] = lane < a.C ? a.W[(F + k) * a.C + lane] : 0.f;
    }
    const int beg = a.rowptr[row], end = a.rowptr[row + 1];
    float4 agg = row_gather_sum<LPN, false>(a.colidx, nullptr, a.X, a.ldx, beg, end, slot, q);
    const float4 xs = *reinterpret_cast<const float4 *>(a.X + (int64_t)row * a.ldx + 4 * q);
    float cnt = (float)(end - beg);
    if (a.self_loop) { agg = f4_add(agg, xs); cnt += 1.f; }
    cnt = fmaxf(cnt, 1.f);                             // unsorted_segment_mean of an empty segment is 0
    agg.x /= cnt; agg.y /= cnt; agg.z /= cnt; agg.w /= cnt;

    float fx[F], fa[F];
    broadcast_row<F>(xs, fx);
    broadcast_row<F>(agg, fa);
    float o = 0.f;
#pragma unroll
    for (int k = 0; k < F; ++k) o = fmaf(fx[k], wx[k], o);
#pragma unroll
    for (int k = 0; k < F; ++k) o = fmaf(fa[k], wa[k], o);
    o = lane < a.C ? o + a.bias[lane] : 0.f;
    float ss = o * o;
    ss = wave_sum_stride<1>(ss);
    o *= rsqrtf(fmaxf(ss, 1e-12f));                    // tf.nn.l2_normalize(axis=-1), before the activation
    o = fmaxf(o, 0.f);
    if (lane < a.C) a.Y[(int64_t)row * a.ldy + lane] = o;
}

// ---- GAT (1 head) ----------------------------------------------------------------------------
struct GatArgs {
    const int32_t *rowptr; const int32_t *colidx; const float *H; int64_t ldh;
    const float *s_self; const float *s_neigh; const float *bias; float *Y; int64_t ldy; int self_loop; int n_rows;
};

__device__ __forceinline__ float leaky02(float x) { return x > 0.f ? x : 0.2f * x; }

template <int C>
__global__ __launch_bounds__(WAVES_PER_BLOCK * AMAR_WAVE) void gat_row_kernel(const GatArgs a) {
    constexpr int LPN = C / 4, NS = AMAR_WAVE / LPN;
    const int lane = threadIdx.x & (AMAR_WAVE - 1);
    const int row = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= a.n_rows) return;
    const int q = lane % LPN, slot = lane / LPN;
    const int beg = a.rowptr[row], end = a.rowptr[row + 1];
    const float si = a.s_self[row];

    // pass 1: LeakyReLU is monotone, so max_j e_ij = LeakyReLU(s_i + max_j n_j)
    float mn = a.self_loop ? a.s_neigh[row] : -INFINITY;
    for (int p = beg + lane; p < end; p += AMAR_WAVE) mn = fmaxf(mn, a.s_neigh[a.colidx[p]]);
    mn = wave_max_all(mn);
    const float emax = leaky02(si + mn);

    // pass 2: un-normalised softmax weights and the weighted sum of source rows
    float4 acc = f4_zero();
    float den = 0.f;
    for (int p = beg + slot; p < end; p += NS) {
        const int c = a.colidx[p];
        const float w = expf(leaky02(si + a.s_neigh[c]) - emax);
        const float4 h = *reinterpret_cast<const float4 *>(a.H + (int64_t)c * a.ldh + 4 * q);
        acc = f4_fma(w, h, acc);
        if (q == 0) den += w;
    }
    if (a.self_loop && slot == 0) {
        const float w = expf(leaky02(si + a.s_neigh[row]) - emax);
        const float4 h = *reinterpret_cast<const float4 *>(a.H + (int64_t)row * a.ldh + 4 * q);
        acc = f4_fma(w, h, acc);
        if (q == 0) den += w;
    }
    acc = f4_wave_sum_stride<LPN>(acc);
    den = wave_sum_stride<1>(den);
    const float inv = 1.f / (den + 1e-9f);
    if (slot == 0) {
        const float4 b = *reinterpret_cast<const float4 *>(a.bias + 4 * q);
        float4 y = make_float4(fmaxf(acc.x * inv + b.x, 0.f), fmaxf(acc.y * inv + b.y, 0.f),
                               fmaxf(acc.z * inv + b.z, 0.f), fmaxf(acc.w * inv + b.w, 0.f));
        *reinterpret_cast<float4 *>(a.Y + (int64_t)row * a.ldy + 4 * q) = y;
    }
}


// ---- GAT on the XCD-sliced image (large graphs) ----------------------------------------------------------------
// Same tiling as spmm_xs_*: workgroup b works on column slice b % S, a wave on one (64-row block, slice) tile, a lane on
// EPL consecutive entries.  The gathered table is G = 2C floats wide per node: [ h_j (C) | s_neigh_j | 0 .. ] (64-byte rows
// for C = 8: never straddling a 128-byte line; 48-byte rows fit the L2s better but a quarter of them straddle: measured
// 0.55-0.59 ms against 0.54 ms per layer), so one L2 request serves the row AND the neighbour scalar.  The softmax is kept exact with the "online" form: every partial
// result is a triple (m, l, o) = (running max of e, sum exp(e - m), sum exp(e - m) h_j), and two triples of one row merge as
//     M = max(m1, m2);  l = l1 exp(m1 - M) + l2 exp(m2 - M);  o = o1 exp(m1 - M) + o2 exp(m2 - M).
// In-lane serial merge over the lane's entries, ONE cross-lane segmented scan with that operator (EPS = 16: one DPP row),
// then the run-end lane merges into the wave's LDS accumulator (plain read-modify-write: a row is touched by at most one
// lane group per LDS instruction, and LDS instructions of a wave stay in order).  Per (slice, row) the kernel leaves
// (o, m, l); the combine kernel merges the slices with the self loop, divides by (l + 1e-9), adds the bias, applies ReLU.
struct GatXsArgs {
    const int32_t *rowptr; const int32_t *colidx; const float *HT; const float *s_self; float *P;
    int n_rows; int n_slices; int blocks_per_slice; bool off32;
    int row_offset;                              // row i of this block is node row_offset + i of s_self / the packed table
};

struct Soft4 { float m; float l; float4 o; };

// One of the two rescale factors is always exp(0) = 1: a single exp(-|m1 - m2|) serves the merge (v_exp_f32 form; both
// maxima -inf means both sides are empty and the factor is moot).
__device__ __forceinline__ void soft_merge(Soft4 &a, float m2, float l2, const float4 &o2) {    // a <- a (+) (m2, l2, o2)
    const float M = fmaxf(a.m, m2);
    const float t = (a.m == m2) ? 1.f : __expf(-fabsf(a.m - m2));
    const float ea = a.m >= m2 ? 1.f : t, eb = a.m >= m2 ? t : 1.f;
    a.l = a.l * ea + l2 * eb;
    a.o = make_float4(a.o.x * ea + o2.x * eb, a.o.y * ea + o2.y * eb, a.o.z * ea + o2.z * eb, a.o.w * ea + o2.w * eb);
    a.m = M;
}

// DIST / EPS: with fewer than 16 entries per step several feature quads share a 16-lane DPP row, so a lane only merges with
// the lane DIST to its left when that lane belongs to the same quad (s >= DIST); with EPS = 16 a DPP row is one quad.
template <int CTRL, int DIST, int EPS>
__device__ __forceinline__ void soft_scan_level(Soft4 &st, int key, int s) {
    const int kprev = __builtin_amdgcn_update_dpp(-1, key, CTRL, 0xF, 0xF, false);
    const float mp = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, st.m), CTRL, 0xF, 0xF, false));
    const float lp = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, st.l), CTRL, 0xF, 0xF, false));
    float4 op;
    op.x = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, st.o.x), CTRL, 0xF, 0xF, false));
    op.y = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, st.o.y), CTRL, 0xF, 0xF, false));
    op.z = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, st.o.z), CTRL, 0xF, 0xF, false));
    op.w = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, st.o.w), CTRL, 0xF, 0xF, false));
    if (kprev == key && (EPS >= 16 || s >= DIST)) soft_merge(st, mp, lp, op);          // lanes without a source got key -1: no merge
}

template <int C, bool OFF32>
__global__ __launch_bounds__(XS_WAVES * AMAR_WAVE) void gat_xs_partial_kernel(const GatXsArgs a) {
    // a packed row is G = 2C floats: QH = C/4 quads of h, then one quad whose first float is s_neigh, the rest padding; an
    // entry takes 2 QH lanes (quads above QH idle), so a step covers EPS = 32 / QH entries: 16 (C = 8), 8 (C = 16), 4 (C = 32)
    constexpr int G = 2 * C, QH = C / 4, EPS = AMAR_WAVE / (2 * QH), EPL = 8, SUPER = EPS * EPL;
    static_assert(C == 8 || C == 16 || C == 32, "EPS must stay inside one 16-lane DPP row");
    constexpr int PAD_KEY = AMAR_WAVE;
    __shared__ float lds_o[XS_WAVES][C * AMAR_WAVE];
    __shared__ float lds_ml[XS_WAVES][3 * AMAR_WAVE];              // m, l, and the tile's s_self
    const int lane = threadIdx.x & (AMAR_WAVE - 1), wv = threadIdx.x >> 6;
    int k, chunk;
    xs_block_to_tile(blockIdx.x, a.n_slices, a.blocks_per_slice, chunk, k);
    const int r0 = __builtin_amdgcn_readfirstlane((chunk * XS_WAVES + wv) * AMAR_WAVE);
    if (r0 >= a.n_rows) return;
    const int nr = min(AMAR_WAVE, a.n_rows - r0);
    const int32_t *rp = a.rowptr + (int64_t)k * a.n_rows + r0;
    const int beg = rp[0], end = rp[nr];
    if (beg == end) return;
    float *acc_o = lds_o[wv], *acc_m = lds_ml[wv], *acc_l = lds_ml[wv] + AMAR_WAVE, *srow = lds_ml[wv] + 2 * AMAR_WAVE;
#pragma unroll
    for (int c = 0; c < C; ++c) acc_o[c * AMAR_WAVE + lane] = 0.f;
    acc_m[lane] = -INFINITY; acc_l[lane] = 0.f;
    srow[lane] = lane < nr ? a.s_self[a.row_offset + r0 + lane] : 0.f;
    const int q = lane / EPS, s = lane % EPS;
    const int n_tile = end - beg;
    const char *cbase = reinterpret_cast<const char *>(a.colidx + beg);

    auto flush = [&](int key, const Soft4 &st) {                     // merge a finished run into the wave's accumulator row
        if (q >= QH) return;                                         // the scalar / padding quads carry no output features
        Soft4 cur;
        cur.m = acc_m[key]; cur.l = acc_l[key];
        cur.o = make_float4(acc_o[(4 * q + 0) * AMAR_WAVE + key], acc_o[(4 * q + 1) * AMAR_WAVE + key],
                            acc_o[(4 * q + 2) * AMAR_WAVE + key], acc_o[(4 * q + 3) * AMAR_WAVE + key]);
        soft_merge(cur, st.m, st.l, st.o);
        acc_o[(4 * q + 0) * AMAR_WAVE + key] = cur.o.x; acc_o[(4 * q + 1) * AMAR_WAVE + key] = cur.o.y;
        acc_o[(4 * q + 2) * AMAR_WAVE + key] = cur.o.z; acc_o[(4 * q + 3) * AMAR_WAVE + key] = cur.o.w;
        if (q == QH - 1) { acc_m[key] = cur.m; acc_l[key] = cur.l; }  // after every quad of this entry read the old (m, l): same instruction order for all
    };

    for (int t0 = 0; t0 < n_tile; t0 += SUPER) {
        const int first = t0 + s * EPL;
        int cw[EPL], key[EPL];
        float4 x[EPL];
        if (t0 + SUPER <= n_tile) {                                  // wave-uniform: a full super-step, the lane's words as 16-byte loads
#pragma unroll                                                       // (8 separate clamped dword loads cost as much TA time as the gathers)
            for (int j = 0; j < EPL; ++j) {
                cw[j] = __builtin_nontemporal_load(reinterpret_cast<const int32_t *>(cbase + (unsigned)(first + j) * 4u));
                key[j] = (int)((unsigned)cw[j] >> 26);
            }
        } else {
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                const unsigned off = (unsigned)max(min(first + j, n_tile - 1), 0) * 4u;
                cw[j] = __builtin_nontemporal_load(reinterpret_cast<const int32_t *>(cbase + off));
                key[j] = first + j < n_tile ? (int)((unsigned)cw[j] >> 26) : PAD_KEY;
            }
        }
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const unsigned col = (unsigned)cw[j] & 0x3ffffffu;
            x[j] = f4_zero();
            if (q <= QH) {
                if (OFF32) x[j] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(a.HT) + (col * (unsigned)G + 4u * q) * 4u);
                else x[j] = *reinterpret_cast<const float4 *>(a.HT + (int64_t)col * G + 4 * q);
            }
        }
        Soft4 st;
        int kcur = PAD_KEY;
        st.m = -INFINITY; st.l = 0.f; st.o = f4_zero();
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            // the neighbour scalar sits in quad QH's first float: lanes (q, s) read it from lane (QH, s)
            const float tj = __shfl(x[j].x, QH * EPS + s, AMAR_WAVE);
            const int kj = key[j];
            if (j > 0 && kj != kcur && kcur != PAD_KEY) flush(kcur, st);
            if (kj != kcur) { st.m = -INFINITY; st.l = 0.f; st.o = f4_zero(); kcur = kj; }
            if (kj != PAD_KEY) {
                const float pre = srow[kj] + tj;
                const float e = pre > 0.f ? pre : 0.2f * pre;
                soft_merge(st, e, 1.f, x[j]);
            }
        }
        // one segmented scan over the EPS lanes of the quad inside its DPP row (row_shr:1, 2, 4, 8), keyed by the lane's last key
        soft_scan_level<0x111, 1, EPS>(st, kcur, s); soft_scan_level<0x112, 2, EPS>(st, kcur, s);
        if (EPS >= 8) soft_scan_level<0x114, 4, EPS>(st, kcur, s);
        if (EPS >= 16) soft_scan_level<0x118, 8, EPS>(st, kcur, s);
        const int knext = __builtin_amdgcn_mov_dpp(kcur, 0x130, 0xF, 0xF, true);
        if (kcur != PAD_KEY && (s == EPS - 1 || knext != kcur)) flush(kcur, st);
    }
    if (lane < nr) {                                                  // P[slice][row] = (o[0..C), m, l, unused...)
        float *out = a.P + ((int64_t)k * a.n_rows + r0 + lane) * G;
#pragma unroll
        for (int c = 0; c < C; ++c) out[c] = acc_o[c * AMAR_WAVE + lane];
        out[C] = acc_m[lane]; out[C + 1] = acc_l[lane];
    }
}

struct GatXsCombineArgs {
    const float *P; const int32_t *rowptr; const float *HT; const float *s_self; const float *bias; float *Y; int64_t ldy;
    int n_rows; int n_slices; int self_loop; int row_offset;
};

template <int C>
__global__ __launch_bounds__(256) void gat_xs_combine_kernel(const GatXsCombineArgs a) {
    constexpr int G = 2 * C;
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= a.n_rows) return;
    float m = -INFINITY, l = 0.f, o[C];
#pragma unroll
    for (int c = 0; c < C; ++c) o[c] = 0.f;
    auto merge = [&](float m2, float l2, const float *o2) {
        const float M = fmaxf(m, m2);
        const float t = (m == m2) ? 1.f : __expf(-fabsf(m - m2));
        const float ea = m >= m2 ? 1.f : t, eb = m >= m2 ? t : 1.f;
        l = l * ea + l2 * eb;
#pragma unroll
        for (int c = 0; c < C; ++c) o[c] = o[c] * ea + o2[c] * eb;
        m = M;
    };
    const float *own = a.HT + (int64_t)(a.row_offset + row) * G;
    if (a.self_loop) {
        const float pre = a.s_self[a.row_offset + row] + own[C];
        merge(pre > 0.f ? pre : 0.2f * pre, 1.f, own);
    }
    const int w0 = __builtin_amdgcn_readfirstlane(row & ~(AMAR_WAVE - 1));
    const int w1 = min(w0 + AMAR_WAVE, a.n_rows);
    for (int k = 0; k < a.n_slices; ++k) {
        const int32_t *rp = a.rowptr + (int64_t)k * a.n_rows;
        if (rp[w0] == rp[w1]) continue;                              // the partial kernel wrote nothing for this (block, slice)
        const float *p = a.P + ((int64_t)k * a.n_rows + row) * G;
        if (p[C + 1] > 0.f) merge(p[C], p[C + 1], p);
    }
    const float inv = 1.f / (l + 1e-9f);
    float *y = a.Y + (int64_t)row * a.ldy;
#pragma unroll
    for (int c = 0; c < C; ++c) y[c] = fmaxf(o[c] * inv + a.bias[c], 0.f);
}

// [ H (C) | s_neigh | 0 .. ] rows of 2C floats: what the partial kernel gathers
__global__ __launch_bounds__(256) void gat_pack_kernel(const float *__restrict__ H, int64_t ldh, const float *__restrict__ s_neigh,
                                                       float *__restrict__ HT, int n, int C) {
    const int G = 2 * C;
    const int64_t total = (int64_t)n * G;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / G;
        const int c = (int)(i - r * G);
        HT[i] = c < C ? H[r * ldh + c] : (c == C ? s_neigh[r] : 0.f);
    }
}


// ---- GraphSAGE tail: Y = relu(l2_normalize([X || AGG] . W + b)) ------------------------------------------------------------
// The part of Spektral's GraphSageConv.call after the aggregation (layer built at src/models/gnn.py:354-361), for the
// routes that produce the mean aggregate with an SpMM (XCD-sliced image on large graphs, column-chunked widths): one
// thread per row, the [2F, C] kernel and the bias in LDS (every lane reads the same element: broadcast), the C outputs in
// registers.  Replaces the copy into [x || agg] + amar_dense_f32 + amar_l2norm_fwd_f32 of the inference path: 96 B per row
// instead of ~350.
struct SageTailArgs { const float *X; int64_t ldx; const float *G; int64_t ldg; int F; const float *W; const float *bias; int C;
                      float *Y; int64_t ldy; int64_t M; };

template <int CP>
__global__ __launch_bounds__(256) void sage_tail_kernel(const SageTailArgs a) {
    extern __shared__ __attribute__((aligned(16))) float w_lds[];
    const int nw = 2 * a.F * a.C;
    for (int i = threadIdx.x; i < nw; i += blockDim.x) w_lds[i] = a.W[i];
    for (int i = threadIdx.x; i < a.C; i += blockDim.x) w_lds[nw + i] = a.bias[i];
    __syncthreads();
    const int C = a.C;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < a.M; r += (int64_t)gridDim.x * blockDim.x) {
        float acc[CP];
#pragma unroll
        for (int c = 0; c < CP; ++c) acc[c] = 0.f;
        for (int half = 0; half < 2; ++half) {
            const float *src = half == 0 ? a.X + r * a.ldx : a.G + r * a.ldg;
            const float *w = w_lds + half * a.F * C;
            for (int k4 = 0; k4 < a.F; k4 += 4) {
                const float4 v = *reinterpret_cast<const float4 *>(src + k4);
                const float xs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int c = 0; c < CP; ++c)
                        if (c < C) acc[c] = fmaf(xs[j], w[(k4 + j) * C + c], acc[c]);
            }
        }
        float sq = 0.f;
#pragma unroll
        for (int c = 0; c < CP; ++c)
            if (c < C) { acc[c] += w_lds[nw + c]; sq = fmaf(acc[c], acc[c], sq); }
        const float iv = rsqrtf(fmaxf(sq, 1e-12f));              // tf.nn.l2_normalize: x * rsqrt(max(sum x^2, 1e-12))
        float *y = a.Y + r * a.ldy;
#pragma unroll
        for (int c4 = 0; c4 < CP; c4 += 4)
            if (c4 < C)
                *reinterpret_cast<float4 *>(y + c4) = make_float4(fmaxf(acc[c4] * iv, 0.f), fmaxf(acc[c4 + 1] * iv, 0.f),
                                                                  fmaxf(acc[c4 + 2] * iv, 0.f), fmaxf(acc[c4 + 3] * iv, 0.f));
    }
}

}  // namespace

extern "C" {

int amar_spmm_csr_f32(const int32_t *rowptr, const int32_t *colidx, const float *vals,
                      const float *X, int64_t ldx, float *Y, int64_t ldy,
                      int32_t n_rows, int32_t F, uint32_t flags, const float *bias,
                      const float *acc_in, int64_t ld_acc_in, float *acc_out, int64_t ld_acc_out,
                      float acc_div, amar_stream_t stream) {
    if (n_rows < 0 || !rowptr || !X) return AMAR_EINVAL;
    if (n_rows == 0) return AMAR_OK;
    if (!colidx) return AMAR_EINVAL;
    const bool accum = flags & AMAR_SPMM_ACCUM;
    if (!Y && !accum) return AMAR_EINVAL;
    if (!ld_ok(ldx, F) || !amar_aligned16(X)) return AMAR_EINVAL;
    if (Y && (!ld_ok(ldy, F) || !amar_aligned16(Y))) return AMAR_EINVAL;
    if ((flags & AMAR_SPMM_BIAS) && (!bias || !amar_aligned16(bias))) return AMAR_EINVAL;
    if (accum && (!acc_in || !acc_out || !ld_ok(ld_acc_in, F) || !ld_ok(ld_acc_out, F) ||
                  !amar_aligned16(acc_in) || !amar_aligned16(acc_out))) return AMAR_EINVAL;
    if ((flags & AMAR_SPMM_ACCUM_DIV) && !(acc_div != 0.f)) return AMAR_EINVAL;
    if (!native_width(F)) {
        if (F < 4 || (F & 3)) return AMAR_EUNSUPPORTED;
        for (int o = 0, w; o < F; o += w) {
            w = chunk_width(F - o);
            const int rc = amar_spmm_csr_f32(rowptr, colidx, vals, X + o, ldx, at(Y, o), ldy, n_rows, w, flags, at(bias, o),
                                             at(acc_in, o), ld_acc_in, at(acc_out, o), ld_acc_out, acc_div, stream);
            if (rc != AMAR_OK) return rc;
        }
        return AMAR_OK;
    }
    SpmmArgs a{};
    a.rowptr = rowptr; a.colidx = colidx; a.vals = vals; a.X = X; a.ldx = ldx; a.Y = Y; a.ldy = ldy;
    a.bias = (flags & AMAR_SPMM_BIAS) ? bias : nullptr; a.relu = (flags & AMAR_SPMM_RELU) ? 1 : 0;
    a.acc_in = acc_in; a.ld_acc_in = ld_acc_in; a.acc_out = acc_out; a.ld_acc_out = ld_acc_out;
    a.acc_div = acc_div; a.accum = accum ? 1 : 0; a.accum_div = (flags & AMAR_SPMM_ACCUM_DIV) ? 1 : 0;
    a.n_rows = n_rows;
    hipStream_t st = static_cast<hipStream_t>(stream);
    return vals ? launch_spmm<true, false>(a, F, st) : launch_spmm<false, false>(a, F, st);
}

int amar_gcn_layer_f32(const int32_t *rowptr, const int32_t *colidx, const float *vals,
                       const float *H, int64_t ldh, int32_t C, const float *bias,
                       float *Y, int64_t ldy,
                       const float *Wnext, int32_t Cn, float *Hnext, int64_t ldhn,
                       int32_t n_rows, amar_stream_t stream) {
    if (n_rows < 0 || !rowptr || !H || !Y || !bias) return AMAR_EINVAL;
    if (n_rows == 0) return AMAR_OK;
    if (!colidx) return AMAR_EINVAL;
    if (!ld_ok(ldh, C) || !ld_ok(ldy, C) || !amar_aligned16(H) || !amar_aligned16(Y) || !amar_aligned16(bias))
        return AMAR_EINVAL;
    if (Wnext && (!Hnext || Cn < 1 || ldhn < Cn)) return AMAR_EINVAL;
    if (Wnext && Cn > 64) return AMAR_EUNSUPPORTED;
    if (!native_width(C)) {
        // column chunks; the fused next-layer product needs the whole output row in one lane, so it is not offered here
        if (C < 4 || (C & 3) || Wnext) return AMAR_EUNSUPPORTED;
        for (int o = 0, w; o < C; o += w) {
            w = chunk_width(C - o);
            const int rc = amar_gcn_layer_f32(rowptr, colidx, vals, H + o, ldh, w, bias + o, Y + o, ldy, nullptr, 0, nullptr, 0,
                                              n_rows, stream);
            if (rc != AMAR_OK) return rc;
        }
        return AMAR_OK;
    }
    SpmmArgs a{};
    a.rowptr = rowptr; a.colidx = colidx; a.vals = vals; a.X = H; a.ldx = ldh; a.Y = Y; a.ldy = ldy;
    a.bias = bias; a.relu = 1; a.Wn = Wnext; a.Cn = Cn; a.Hn = Hnext; a.ldhn = ldhn; a.n_rows = n_rows;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (Wnext) return vals ? launch_spmm<true, true>(a, C, st) : launch_spmm<false, true>(a, C, st);
    return vals ? launch_spmm<true, false>(a, C, st) : launch_spmm<false, false>(a, C, st);
}

static int rowwise_xw_run(const float *X, int64_t ldx, int32_t F, const float *W, int32_t C,
                          float *H, int64_t ldh, float *copy_to, int64_t ld_copy,
                          const float *a_self, const float *a_neigh, float *s_self, float *s_neigh,
                          const float *row_scale, const int32_t *row_ids, int32_t n_rows, amar_stream_t stream);

int amar_rowwise_xw_f32(const float *X, int64_t ldx, int32_t F, const float *W, int32_t C,
                        float *H, int64_t ldh, float *copy_to, int64_t ld_copy,
                        const float *a_self, const float *a_neigh, float *s_self, float *s_neigh,
                        const float *row_scale, int32_t n_rows, amar_stream_t stream) {
    return rowwise_xw_run(X, ldx, F, W, C, H, ldh, copy_to, ld_copy, a_self, a_neigh, s_self, s_neigh, row_scale, nullptr, n_rows, stream);
}

int amar_rowwise_xw_gather_f32(const float *X, int64_t ldx, int32_t F, const int32_t *row_ids, const float *W, int32_t C,
                               float *H, int64_t ldh, const float *row_scale, int32_t n_rows, amar_stream_t stream) {
    if (!row_ids) return AMAR_EINVAL;
    return rowwise_xw_run(X, ldx, F, W, C, H, ldh, nullptr, 0, nullptr, nullptr, nullptr, nullptr, row_scale, row_ids, n_rows, stream);
}

static int rowwise_xw_run(const float *X, int64_t ldx, int32_t F, const float *W, int32_t C,
                          float *H, int64_t ldh, float *copy_to, int64_t ld_copy,
                          const float *a_self, const float *a_neigh, float *s_self, float *s_neigh,
                          const float *row_scale, const int32_t *row_ids, int32_t n_rows, amar_stream_t stream) {
    if (n_rows < 0 || !X || !W || !H || F < 1 || C < 1 || ldx < F || ldh < C) return AMAR_EINVAL;
    if (F > 64 || C > 64) return AMAR_EUNSUPPORTED;
    if (copy_to && ld_copy < F) return AMAR_EINVAL;
    const bool attn = a_self || a_neigh || s_self || s_neigh;
    if (attn && !(a_self && a_neigh && s_self && s_neigh)) return AMAR_EINVAL;
    if (n_rows == 0) return AMAR_OK;
    int CP = 1;
    while (CP < C) CP <<= 1;
    if (attn && row_scale) return AMAR_EINVAL;                       // the attention scalars are defined on the un-scaled product
    XwArgs a{X, ldx, F, W, C, H, ldh, copy_to, ld_copy, a_self, a_neigh, attn ? s_self : nullptr, s_neigh, n_rows, row_scale, row_ids};
    static const bool no_vec = getenv("AMAR_XW_VEC") && atoi(getenv("AMAR_XW_VEC")) == 0;            // development switch (A/B timing)
    if (!no_vec && F == C && (F == 8 || F == 16) && (ldx & 3) == 0 && (ldh & 3) == 0 && amar_aligned16(X) && amar_aligned16(H) &&
        (!copy_to || ((ld_copy & 3) == 0 && amar_aligned16(copy_to)))) {
        const int rows_per_block = F == 8 ? 512 : 256;              // (two rows per thread at F = 8)
        int64_t vblocks = ((int64_t)n_rows + rows_per_block - 1) / rows_per_block;
        if (vblocks > 8192) vblocks = 8192;
        if (F == 8) hipLaunchKernelGGL(rowwise_xw_vec_kernel<8>, dim3((unsigned)vblocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
        else hipLaunchKernelGGL(rowwise_xw_vec_kernel<16>, dim3((unsigned)vblocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
        return amar_check_launch();
    }
    const int rows_per_block = 256 / CP;
    int64_t blocks = ((int64_t)n_rows + rows_per_block - 1) / rows_per_block;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(rowwise_xw_kernel, dim3((unsigned)blocks), dim3(256), (size_t)F * C * sizeof(float),
                       static_cast<hipStream_t>(stream), a, CP);
    return amar_check_launch();
}

int amar_sage_layer_f32(const int32_t *rowptr, const int32_t *colidx,
                        const float *X, int64_t ldx, int32_t F,
                        const float *W, const float *bias, int32_t C,
                        float *Y, int64_t ldy, int32_t self_loop,
                        int32_t n_rows, amar_stream_t stream) {
    if (n_rows < 0 || !rowptr || !X || !W || !bias || !Y || C < 1 || ldy < C) return AMAR_EINVAL;
    if (!ld_ok(ldx, F) || !amar_aligned16(X)) return AMAR_EINVAL;
    if (C > 64) return AMAR_EUNSUPPORTED;
    if (n_rows == 0) return AMAR_OK;
    if (!colidx) return AMAR_EINVAL;
    SageArgs a{rowptr, colidx, X, ldx, W, bias, C, Y, ldy, self_loop ? 1 : 0, n_rows};
    const dim3 grid((n_rows + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK), block(WAVES_PER_BLOCK * AMAR_WAVE);
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (F) {
    case 4:  hipLaunchKernelGGL(sage_row_kernel<4>, grid, block, 0, st, a); break;
    case 8:  hipLaunchKernelGGL(sage_row_kernel<8>, grid, block, 0, st, a); break;
    case 16: hipLaunchKernelGGL(sage_row_kernel<16>, grid, block, 0, st, a); break;
    case 32: hipLaunchKernelGGL(sage_row_kernel<32>, grid, block, 0, st, a); break;
    default: return AMAR_EUNSUPPORTED;
    }
    return amar_check_launch();
}

int amar_sage_tail_f32(const float *X, int64_t ldx, const float *AGG, int64_t lda, int32_t F,
                       const float *W, const float *bias, int32_t C, float *Y, int64_t ldy,
                       int64_t n_rows, amar_stream_t stream) {
    if (n_rows < 0 || !X || !AGG || !W || !bias || !Y || F < 4 || C < 4 || (F & 3) || (C & 3)) return AMAR_EINVAL;
    if (!ld_ok(ldx, F) || !ld_ok(lda, F) || !ld_ok(ldy, C) || !amar_aligned16(X) || !amar_aligned16(AGG) || !amar_aligned16(Y))
        return AMAR_EINVAL;
    if (F > 64 || C > 64) return AMAR_EUNSUPPORTED;
    if (n_rows == 0) return AMAR_OK;
    SageTailArgs a{X, ldx, AGG, lda, F, W, bias, C, Y, ldy, n_rows};
    const size_t lds = (size_t)(2 * F * C + C) * sizeof(float);   // <= 33 KB
    int64_t blocks = (n_rows + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    const dim3 grid((unsigned)blocks), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (C <= 4) hipLaunchKernelGGL(sage_tail_kernel<4>, grid, block, lds, st, a);
    else if (C <= 8) hipLaunchKernelGGL(sage_tail_kernel<8>, grid, block, lds, st, a);
    else if (C <= 16) hipLaunchKernelGGL(sage_tail_kernel<16>, grid, block, lds, st, a);
    else if (C <= 32) hipLaunchKernelGGL(sage_tail_kernel<32>, grid, block, lds, st, a);
    else hipLaunchKernelGGL(sage_tail_kernel<64>, grid, block, lds, st, a);
    return amar_check_launch();
}

int amar_gat_layer_f32(const int32_t *rowptr, const int32_t *colidx,
                       const float *H, int64_t ldh, int32_t C,
                       const float *s_self, const float *s_neigh, const float *bias,
                       float *Y, int64_t ldy, int32_t self_loop,
                       int32_t n_rows, amar_stream_t stream) {
    if (n_rows < 0 || !rowptr || !H || !s_self || !s_neigh || !bias || !Y) return AMAR_EINVAL;
    if (!ld_ok(ldh, C) || !ld_ok(ldy, C) || !amar_aligned16(H) || !amar_aligned16(Y) || !amar_aligned16(bias))
        return AMAR_EINVAL;
    if (n_rows == 0) return AMAR_OK;
    if (!colidx) return AMAR_EINVAL;
    if (!native_width(C)) {
        // the attention coefficients only depend on the per-node scalars: each column chunk recomputes the same softmax
        if (C < 4 || (C & 3)) return AMAR_EUNSUPPORTED;
        for (int o = 0, w; o < C; o += w) {
            w = chunk_width(C - o);
            const int rc = amar_gat_layer_f32(rowptr, colidx, H + o, ldh, w, s_self, s_neigh, bias + o, Y + o, ldy, self_loop,
                                              n_rows, stream);
            if (rc != AMAR_OK) return rc;
        }
        return AMAR_OK;
    }
    GatArgs a{rowptr, colidx, H, ldh, s_self, s_neigh, bias, Y, ldy, self_loop ? 1 : 0, n_rows};
    const dim3 grid((n_rows + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK), block(WAVES_PER_BLOCK * AMAR_WAVE);
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (C) {
    case 4:  hipLaunchKernelGGL(gat_row_kernel<4>, grid, block, 0, st, a); break;
    case 8:  hipLaunchKernelGGL(gat_row_kernel<8>, grid, block, 0, st, a); break;
    case 16: hipLaunchKernelGGL(gat_row_kernel<16>, grid, block, 0, st, a); break;
    case 32: hipLaunchKernelGGL(gat_row_kernel<32>, grid, block, 0, st, a); break;
    case 64: hipLaunchKernelGGL(gat_row_kernel<64>, grid, block, 0, st, a); break;
    default: return AMAR_EUNSUPPORTED;
    }
    return amar_check_launch();
}

int amar_gat_xs_f32(const int32_t *rowptr, const int32_t *colidx, int32_t n_slices,
                    const float *H, int64_t ldh, int32_t C, const float *s_self, const float *s_neigh, const float *bias,
                    float *packed, float *partials, float *Y, int64_t ldy, int32_t self_loop, int32_t n_rows,
                    int32_t n_cols, int32_t row_offset, amar_stream_t stream) {
    if (n_rows < 0 || n_slices < 1 || !rowptr || !H || !s_self || !s_neigh || !bias || !packed || !partials || !Y) return AMAR_EINVAL;
    if (n_cols < n_rows || row_offset < 0 || row_offset + n_rows > n_cols) return AMAR_EINVAL;
    if (ldh < C || ldy < C || !amar_aligned16(packed) || !amar_aligned16(partials)) return AMAR_EINVAL;
    if (C != 8 && C != 16 && C != 32) return AMAR_EUNSUPPORTED;
    if (n_rows == 0) return AMAR_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t total = (int64_t)n_cols * (2 * C);
    hipLaunchKernelGGL(gat_pack_kernel, dim3((unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256)), dim3(256), 0, st,
                       H, ldh, s_neigh, packed, n_cols, C);
    GatXsArgs pa{rowptr, colidx, packed, s_self, partials, n_rows, n_slices,
                 (n_rows + XS_WAVES * AMAR_WAVE - 1) / (XS_WAVES * AMAR_WAVE), (int64_t)n_cols * (2 * C) * 4 < (int64_t(1) << 32), row_offset};
    const dim3 pgrid((unsigned)(pa.blocks_per_slice * pa.n_slices)), block(XS_WAVES * AMAR_WAVE);
    GatXsCombineArgs ca{partials, rowptr, packed, s_self, bias, Y, ldy, n_rows, n_slices, self_loop ? 1 : 0, row_offset};
    const dim3 cgrid((n_rows + 255) / 256);
#define AMAR_GAT_XS(CC)                                                                                     \
    do {                                                                                                     \
        if (pa.off32) hipLaunchKernelGGL((gat_xs_partial_kernel<CC, true>), pgrid, block, 0, st, pa);        \
        else hipLaunchKernelGGL((gat_xs_partial_kernel<CC, false>), pgrid, block, 0, st, pa);                \
        hipLaunchKernelGGL((gat_xs_combine_kernel<CC>), cgrid, dim3(256), 0, st, ca);                        \
    } while (0)
    if (C == 8) AMAR_GAT_XS(8); else if (C == 16) AMAR_GAT_XS(16); else AMAR_GAT_XS(32);
#undef AMAR_GAT_XS
    return amar_check_launch();
}

int amar_spmm_sj_f32(const int32_t *entries, const int16_t *counts, const int32_t *wave_start, int32_t n_slices,
                     const float *X, int64_t ldx, float *Y, int64_t ldy,
                     int32_t n_rows, int32_t F, uint32_t flags, const float *bias,
                     const float *acc_in, int64_t ld_acc_in, float *acc_out, int64_t ld_acc_out, float acc_div,
                     const float *Wnext, int32_t Cn, float *Hnext, int64_t ldhn, amar_stream_t stream) {
    if (n_rows < 0 || n_slices < 1 || !counts || !wave_start || !X) return AMAR_EINVAL;
    if (n_rows == 0) return AMAR_OK;
    if (!entries) return AMAR_EINVAL;
    const bool accum = flags & AMAR_SPMM_ACCUM;
    if (!Y && !accum) return AMAR_EINVAL;
    if (!ld_ok(ldx, F) || !amar_aligned16(X)) return AMAR_EINVAL;
    if (Y && (!ld_ok(ldy, F) || !amar_aligned16(Y))) return AMAR_EINVAL;
    if ((flags & AMAR_SPMM_BIAS) && (!bias || !amar_aligned16(bias))) return AMAR_EINVAL;
    if (accum && (!acc_in || !acc_out || !ld_ok(ld_acc_in, F) || !ld_ok(ld_acc_out, F) ||
                  !amar_aligned16(acc_in) || !amar_aligned16(acc_out))) return AMAR_EINVAL;
    if ((flags & AMAR_SPMM_ACCUM_DIV) && !(acc_div != 0.f)) return AMAR_EINVAL;
    if (Wnext && (!Hnext || Cn < 1 || ldhn < Cn)) return AMAR_EINVAL;
    if (Wnext && Cn > 64) return AMAR_EUNSUPPORTED;
    SjArgs a{};
    a.entries = reinterpret_cast<const int2 *>(entries); a.counts = counts; a.wave_start = wave_start;
    a.n_slices = n_slices; a.n_waves = (n_rows + AMAR_WAVE - 1) / AMAR_WAVE;
    a.e.X = X; a.e.ldx = ldx; a.e.Y = Y; a.e.ldy = ldy;
    a.e.bias = (flags & AMAR_SPMM_BIAS) ? bias : nullptr; a.e.relu = (flags & AMAR_SPMM_RELU) ? 1 : 0;
    a.e.acc_in = acc_in; a.e.ld_acc_in = ld_acc_in; a.e.acc_out = acc_out; a.e.ld_acc_out = ld_acc_out;
    a.e.acc_div = acc_div; a.e.accum = accum ? 1 : 0; a.e.accum_div = (flags & AMAR_SPMM_ACCUM_DIV) ? 1 : 0;
    a.e.Wn = Wnext; a.e.Cn = Cn; a.e.Hn = Hnext; a.e.ldhn = ldhn; a.e.n_rows = n_rows;
    hipStream_t st = static_cast<hipStream_t>(stream);
    return Wnext ? launch_spmm_sj<true>(a, F, st) : launch_spmm_sj<false>(a, F, st);
}

int amar_spmm_xs_f32(const float *diag, const int32_t *rowptr, const int32_t *colidx, const float *vals, const float *row_scale,
                     int32_t n_slices, const float *X, int64_t ldx, int32_t n_cols, const float *Xself, float *partials,
                     float *Y, int64_t ldy, int32_t n_rows, int32_t F, uint32_t flags, const float *bias,
                     const float *acc_in, int64_t ld_acc_in, float *acc_out, int64_t ld_acc_out, float acc_div,
                     const float *Wnext, int32_t Cn, float *Hnext, int64_t ldhn, amar_stream_t stream) {
    if (n_rows < 0 || n_cols < 0 || n_slices < 1 || !diag || !rowptr || !X || !partials) return AMAR_EINVAL;
    if (n_rows == 0) return AMAR_OK;
    if (!Xself) { if (n_cols < n_rows) return AMAR_EINVAL; Xself = X; }   // square: row i's own features are X[i]
    if (!amar_aligned16(Xself)) return AMAR_EINVAL;
    // colidx / vals may be NULL when the matrix has no off-diagonal entry (every segment is then empty)
    const bool accum = flags & AMAR_SPMM_ACCUM;
    if (!Y && !accum) return AMAR_EINVAL;
    if (!ld_ok(ldx, F) || !amar_aligned16(X) || !amar_aligned16(partials)) return AMAR_EINVAL;
    if (Y && (!ld_ok(ldy, F) || !amar_aligned16(Y))) return AMAR_EINVAL;
    if ((flags & AMAR_SPMM_BIAS) && (!bias || !amar_aligned16(bias))) return AMAR_EINVAL;
    if (accum && (!acc_in || !acc_out || !ld_ok(ld_acc_in, F) || !ld_ok(ld_acc_out, F) ||
                  !amar_aligned16(acc_in) || !amar_aligned16(acc_out))) return AMAR_EINVAL;
    if ((flags & AMAR_SPMM_ACCUM_DIV) && !(acc_div != 0.f)) return AMAR_EINVAL;
    if (Wnext && (!Hnext || Cn < 1 || ldhn < Cn)) return AMAR_EINVAL;
    if (Wnext && Cn > 64) return AMAR_EUNSUPPORTED;
    if ((flags & AMAR_SPMM_SCALE_NEXT) && (!row_scale || !Wnext)) return AMAR_EINVAL;
    if ((vals != nullptr) == (row_scale != nullptr) && colidx) return AMAR_EINVAL;   // exactly one of them (unless there are no entries)
    XsArgs pa{rowptr, colidx, vals, X, ldx, partials, n_rows, n_slices,
              (n_rows + XS_WAVES * AMAR_WAVE - 1) / (XS_WAVES * AMAR_WAVE),
              (int64_t)n_cols * ldx * 4 < (int64_t(1) << 32)};
    XsCombineArgs ca{};
    ca.diag = diag; ca.P = partials; ca.rowptr = rowptr; ca.n_slices = n_slices; ca.row_scale = row_scale; ca.Xself = Xself;
    ca.e.next_scale = (flags & AMAR_SPMM_SCALE_NEXT) ? row_scale : nullptr;
    ca.e.X = X; ca.e.ldx = ldx; ca.e.Y = Y; ca.e.ldy = ldy;
    ca.e.bias = (flags & AMAR_SPMM_BIAS) ? bias : nullptr; ca.e.relu = (flags & AMAR_SPMM_RELU) ? 1 : 0;
    ca.e.acc_in = acc_in; ca.e.ld_acc_in = ld_acc_in; ca.e.acc_out = acc_out; ca.e.ld_acc_out = ld_acc_out;
    ca.e.acc_div = acc_div; ca.e.accum = accum ? 1 : 0; ca.e.accum_div = (flags & AMAR_SPMM_ACCUM_DIV) ? 1 : 0;
    ca.e.Wn = Wnext; ca.e.Cn = Cn; ca.e.Hn = Hnext; ca.e.ldhn = ldhn; ca.e.n_rows = n_rows;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (F) {
    case 4:  return launch_spmm_xs<4>(pa, ca, Wnext != nullptr, st);
    case 8:  return launch_spmm_xs<8>(pa, ca, Wnext != nullptr, st);
    case 16: return launch_spmm_xs<16>(pa, ca, Wnext != nullptr, st);
    case 32: return launch_spmm_xs<32>(pa, ca, Wnext != nullptr, st);
    case 64: return launch_spmm_xs<64>(pa, ca, Wnext != nullptr, st);
    default: return AMAR_EUNSUPPORTED;
    }
}

int amar_spmm_lt_f32(const int32_t *words, const int32_t *stream_start, const int32_t *wsteps, const int32_t *tile_row0,
                     const int32_t *n_win, const int32_t *vstart, const int32_t *vcount, int32_t n_tiles, int32_t maxwin1, int32_t pace_every,
                     const float *diag, const float *row_scale,
                     const float *X, int64_t ldx, int32_t n_cols, const float *Xself,
                     float *Y, int64_t ldy, int32_t n_rows, int32_t F, uint32_t flags, const float *bias,
                     const float *acc_in, int64_t ld_acc_in, float *acc_out, int64_t ld_acc_out, float acc_div,
                     const float *Wnext, int32_t Cn, float *Hnext, int64_t ldhn, amar_stream_t stream) {
    if (n_rows < 0 || n_cols < 0 || n_tiles < 0 || maxwin1 < 1 || pace_every < 1 || (pace_every & (pace_every - 1))) return AMAR_EINVAL;
    if (n_rows == 0 || n_tiles == 0) return n_rows == 0 ? AMAR_OK : AMAR_EINVAL;
    if (!words || !stream_start || !wsteps || !tile_row0 || !n_win || !vstart || !vcount || !diag || !row_scale || !X) return AMAR_EINVAL;
    if (F != 4 && F != 8 && F != 16 && F != 32) return AMAR_EUNSUPPORTED;
    const int rw = LT_TILE_BYTES / (4 * F * LT_WAVES);
    int lbits = 0;
    while ((1 << lbits) < rw) ++lbits;
    const int cbits = 31 - lbits;
    if ((int64_t)n_cols > (int64_t(1) << cbits)) return AMAR_EUNSUPPORTED;
    if (!Xself) { if (n_cols < n_rows) return AMAR_EINVAL; Xself = X; }
    if (!amar_aligned16(Xself) || !amar_aligned16(words)) return AMAR_EINVAL;
    const bool accum = flags & AMAR_SPMM_ACCUM;
    if (!Y && !accum) return AMAR_EINVAL;
    if (!ld_ok(ldx, F) || !amar_aligned16(X)) return AMAR_EINVAL;
    if (Y && (!ld_ok(ldy, F) || !amar_aligned16(Y))) return AMAR_EINVAL;
    if ((flags & AMAR_SPMM_BIAS) && (!bias || !amar_aligned16(bias))) return AMAR_EINVAL;
    if (accum && (!acc_in || !acc_out || !ld_ok(ld_acc_in, F) || !ld_ok(ld_acc_out, F) ||
                  !amar_aligned16(acc_in) || !amar_aligned16(acc_out))) return AMAR_EINVAL;
    if ((flags & AMAR_SPMM_ACCUM_DIV) && !(acc_div != 0.f)) return AMAR_EINVAL;
    const bool sage = flags & AMAR_SPMM_SAGE_TAIL;
    if (sage) {                                                       // Wnext = the layer's kernel [2F, F], bias its bias, Y the layer's output
        if (!Wnext || !bias || !Y || accum || (flags & (AMAR_SPMM_RELU | AMAR_SPMM_SCALE_NEXT))) return AMAR_EINVAL;
        if (Cn != F || F < 8) return AMAR_EUNSUPPORTED;
        if (Hnext && (!ld_ok(ldhn, F) || !amar_aligned16(Hnext))) return AMAR_EINVAL;
    } else if (Wnext && (!Hnext || Cn < 1 || ldhn < Cn)) return AMAR_EINVAL;
    if (Wnext && Cn > 64) return AMAR_EUNSUPPORTED;
    if ((flags & AMAR_SPMM_SCALE_NEXT) && !Wnext) return AMAR_EINVAL;
    LtArgs a{};
    a.words = words; a.stream_start = stream_start; a.wsteps = wsteps; a.tile_row0 = tile_row0; a.n_win = n_win;
    a.vstart = vstart; a.vcount = vcount;
    a.maxwin1 = maxwin1; a.cbits = cbits; a.pace_mask = pace_every - 1;
    a.X = X; a.ldx = ldx; a.Xself = Xself; a.diag = diag; a.row_scale = row_scale;
    a.e.next_scale = (flags & AMAR_SPMM_SCALE_NEXT) ? row_scale : nullptr;
    a.e.X = X; a.e.ldx = ldx; a.e.Y = Y; a.e.ldy = ldy;
    a.e.bias = ((flags & AMAR_SPMM_BIAS) || sage) ? bias : nullptr; a.e.relu = (flags & AMAR_SPMM_RELU) ? 1 : 0;
    a.e.acc_in = acc_in; a.e.ld_acc_in = ld_acc_in; a.e.acc_out = acc_out; a.e.ld_acc_out = ld_acc_out;
    a.e.acc_div = acc_div; a.e.accum = accum ? 1 : 0; a.e.accum_div = (flags & AMAR_SPMM_ACCUM_DIV) ? 1 : 0;
    a.e.Wn = Wnext; a.e.Cn = Cn; a.e.Hn = Hnext; a.e.ldhn = ldhn; a.e.n_rows = n_rows;
    const int off32 = (int64_t)n_cols * ldx * 4 < (int64_t(1) << 32) ? (ldx == F ? 2 : 1) : 0;
    const bool nopairs = (flags & AMAR_SPMM_LT_NOPAIRS) != 0;
    const char *venv = getenv("AMAR_LT_VARIANT");               // development switch (tools/exp_lt.py)
    const int variant = venv ? atoi(venv) : 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (F) {
    case 4:  return launch_spmm_lt<4>(a, n_tiles, off32, Wnext != nullptr && !sage, variant, sage, nopairs, st);
    case 8:  return launch_spmm_lt<8>(a, n_tiles, off32, Wnext != nullptr && !sage, variant, sage, nopairs, st);
    case 16: return launch_spmm_lt<16>(a, n_tiles, off32, Wnext != nullptr && !sage, variant, sage, nopairs, st);
    default: return launch_spmm_lt<32>(a, n_tiles, off32, Wnext != nullptr && !sage, variant, sage, nopairs, st);
    }
}

int amar_colmax_f32(const float *x, int64_t n, float *out, amar_stream_t stream) {
    if (n < 0 || !out || (n > 0 && !x)) return AMAR_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(out), 0xFF800000u, 1, st) != hipSuccess) return amar_check_launch();   // -inf
    if (n == 0) return AMAR_OK;
    const int64_t blocks = (n + 255) / 256;
    // few blocks: every block ends in one atomic on the same word, and 1 024 of them took 13 us for 590 k floats
    hipLaunchKernelGGL(colmax_kernel, dim3((unsigned)(blocks < 96 ? blocks : 96)), dim3(256), 0, st, x, n, out);
    return amar_check_launch();
}

int amar_gat_lt_rows_per_wave(int32_t C) { return (C == 8 || C == 16 || C == 32) ? lt_gat_rw(C) : 0; }

int amar_gat_lt_f32(const int32_t *words, const int32_t *stream_start, const int32_t *wsteps, const int32_t *tile_row0,
                    const int32_t *n_win, const int32_t *vstart, const int32_t *vcount, int32_t n_tiles, int32_t maxwin1, int32_t pace_every,
                    int32_t rows_per_wave, const float *diag, const int32_t *rowptr, const int32_t *colidx,
                    const float *H, int64_t ldh, int32_t C, const float *s_self, const float *s_neigh, const float *s_neigh_max,
                    const float *bias, float *Y, int64_t ldy, int32_t self_loop, int32_t n_rows, int32_t n_cols, int32_t row_offset,
                    amar_stream_t stream) {
    // the image's geometry is part of the contract: an image cut for taller tiles would index past the workgroup's LDS rows
    if (rows_per_wave != amar_gat_lt_rows_per_wave(C)) return (C == 8 || C == 16 || C == 32) ? AMAR_EINVAL : AMAR_EUNSUPPORTED;
    if (n_rows < 0 || n_cols < 0 || n_tiles < 0 || maxwin1 < 1 || pace_every < 1 || (pace_every & (pace_every - 1)) || row_offset < 0) return AMAR_EINVAL;
    if (n_rows == 0 || n_tiles == 0) return n_rows == 0 ? AMAR_OK : AMAR_EINVAL;
    if (!words || !stream_start || !wsteps || !tile_row0 || !n_win || !vstart || !vcount || !diag || !rowptr || !colidx ||
        !H || !s_self || !s_neigh || !s_neigh_max || !bias || !Y) return AMAR_EINVAL;
    if (C != 8 && C != 16 && C != 32) return AMAR_EUNSUPPORTED;
    if ((int64_t)row_offset + n_rows > n_cols) return AMAR_EINVAL;
    const int cbits = 31 - lt_bits(lt_gat_rw(C));
    if ((int64_t)n_cols > (int64_t(1) << cbits)) return AMAR_EUNSUPPORTED;
    if (!ld_ok(ldh, C) || !amar_aligned16(H) || !ld_ok(ldy, C) || !amar_aligned16(Y) || !amar_aligned16(bias) || !amar_aligned16(words)) return AMAR_EINVAL;
    LtArgs a{};
    a.words = words; a.stream_start = stream_start; a.wsteps = wsteps; a.tile_row0 = tile_row0; a.n_win = n_win;
    a.vstart = vstart; a.vcount = vcount;
    a.maxwin1 = maxwin1; a.cbits = cbits; a.pace_mask = pace_every - 1;
    a.X = H; a.ldx = ldh; a.Xself = H + (int64_t)row_offset * ldh; a.diag = diag; a.row_scale = nullptr;
    a.s_neigh = s_neigh; a.bmax = s_neigh_max; a.s_self_rows = s_self + row_offset; a.s_neigh_rows = s_neigh + row_offset;
    a.csr_rowptr = rowptr; a.csr_colidx = colidx; a.self_loop = self_loop ? 1 : 0;
    a.e.X = H; a.e.ldx = ldh; a.e.Y = Y; a.e.ldy = ldy; a.e.bias = bias; a.e.relu = 1; a.e.n_rows = n_rows;
    const int off32 = (int64_t)n_cols * ldh * 4 < (int64_t(1) << 32) ? (ldh == C ? 2 : 1) : 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (C) {
    case 8:  return launch_gat_lt<8>(a, n_tiles, off32, st);
    case 16: return launch_gat_lt<16>(a, n_tiles, off32, st);
    default: return launch_gat_lt<32>(a, n_tiles, off32, st);
    }
}

#ifdef AMAR_LT_STAMPS
int amar_lt_debug_copy(unsigned long long *host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(lt_debug_stamps), (size_t)n * sizeof(unsigned long long));
}
#endif

}  // extern "C"

