/*
 * amar_hip.h — C-ABI of the MI355X (gfx950) GNN-propagation + hybrid-scoring hot path.
 *
 * The reference (swapUniba/Deep_CBRS_Amar_Renaissance) is pure Python on TensorFlow/Keras/
 * Spektral and has no FFI of its own; each entry point below replaces the framework call the
 * reference makes at the cited file:line (paths relative to the reference root).  A maintainer
 * binds these with ctypes — see INTEGRATION.md.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller; nothing here allocates, frees or
 *     synchronises; work is enqueued on `stream` (a hipStream_t passed as void*, NULL = default)
 *   - matrices are row-major fp32 with an explicit leading dimension (in elements), so a layer
 *     can write straight into its column slice of the [N, d(L+1)] concatenation buffer
 *     (ReductionLayer 'concatenation', src/layers/reduction.py:15-16)
 *   - CSR is canonical: int32 rowptr[n_rows+1], int32 colidx[nnz] ascending within a row
 *   - return value: 0 = ok, <0 = AMAR_E* below; never throws, never aborts
 *   - re-entrant for distinct streams; no global mutable state
 */
#ifndef AMAR_HIP_H
#define AMAR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AMAR_OK            0
#define AMAR_EINVAL       -1   /* bad argument (null pointer, negative size, misaligned ld)   */
#define AMAR_EUNSUPPORTED -2   /* shape outside what the kernels are built for                */
#define AMAR_ELAUNCH      -3   /* HIP reported an error at launch (see amar_last_hip_error)   */

#define AMAR_ACT_NONE     0
#define AMAR_ACT_RELU     1
#define AMAR_ACT_SIGMOID  2
#define AMAR_DENSE_WT 0x100    /* amar_dense_f32, OR-ed into `act`: W is given transposed ([N, K] row-major), for dX = dZ . W^T */

/* flags of amar_spmm_csr_f32 */
#define AMAR_SPMM_BIAS      1u   /* y += bias[F]                                              */
#define AMAR_SPMM_RELU      2u   /* y = max(y, 0) after bias                                  */
#define AMAR_SPMM_ACCUM     4u   /* acc_out = acc_in + y (LightGCN running layer sum)         */
#define AMAR_SPMM_ACCUM_DIV 8u   /* ... and acc_out /= acc_div (ReductionLayer 'mean')        */
#define AMAR_SPMM_SCALE_NEXT 16u /* amar_spmm_xs_f32, value-free image: Hnext[i] *= row_scale[i] */
#define AMAR_SPMM_SAGE_TAIL 32u  /* amar_spmm_lt_f32 on GraphSAGE's mean-aggregate image (diag = self loop, row_scale = 1/count, X un-scaled):
                                    Y[i] = relu(l2_normalize([X_i || mean_i] . Wnext + bias)), Wnext [2F, F] (Cn = F), the tail of
                                    spektral GraphSageConv (src/models/gnn.py:354-361) in the same launch; Hnext (optional, ld ldhn) receives
                                    a second copy of Y: a dense table for the next layer's gathers when Y is a concat slice */
#define AMAR_SPMM_LT_NOPAIRS 64u /* amar_spmm_lt_f32: the image holds no implicit pairs — every repeat of a virtual row inside a step is
                                    flagged (utilities/lds_tiled.py, pairs=False: the default for F >= 16) — so the kernel may skip
                                    the pair logic of a step; an image WITH implicit pairs must not carry this flag */

typedef void *amar_stream_t;

int amar_version(void);
const char *amar_error_string(int code);
int amar_last_hip_error(void);          /* last hipError_t seen by this thread, 0 if none */

/* ---- propagation --------------------------------------------------------------------------
 * Y[n_rows, F] = A . X   (+ bias, ReLU, running sum)        fp32, CSR, F in {4, 8, 16, 32, 64};
 * any other multiple of 4 runs as column chunks of those widths (24 = 16 + 8), same results column by column
 * Replaces spektral.layers.ops.modal_dot -> tf.sparse.sparse_dense_matmul at
 * src/layers/lightgcn_conv.py:51-54 and inside GCNConv.call (built at src/models/gnn.py:289-295,
 * invoked at src/models/gnn.py:78).  vals == NULL means an all-ones (binary) matrix.
 * ldx, ldy (and ld_acc) must be multiples of 4 and the bases 16-byte aligned.
 * Y may be NULL when only the running sum is wanted (last LightGCN layer).
 */
int amar_spmm_csr_f32(const int32_t *rowptr, const int32_t *colidx, const float *vals,
                      const float *X, int64_t ldx, float *Y, int64_t ldy,
                      int32_t n_rows, int32_t F, uint32_t flags, const float *bias,
                      const float *acc_in, int64_t ld_acc_in, float *acc_out, int64_t ld_acc_out,
                      float acc_div, amar_stream_t stream);

/* The same product on the sliced-jagged (SJ) image of A, for graphs whose node table exceeds the
 * per-XCD L2 (built by utilities/math.py:SlicedJagged.from_csr; any producer may build it):
 *   rows are taken 64 at a time (wave w = rows [64w, 64w+64), lane = row & 63);
 *   columns are cut into n_slices slices of 2^cbits columns (slice of col = col >> cbits);
 *   for w = 0.., for slice = 0..n_slices-1, for j = 0,1,..: the j-th non-zero (ascending column) of
 *   every row of w that has more than j non-zeros in that slice, in ascending lane order, is the
 *   next entry:  entries[e] = (int32 global column, fp32 value bits)               8 bytes per non-zero
 *   counts[(w * n_slices + slice) * 64 + lane] = non-zeros of that row in that slice (int16)
 *   wave_start[w] = index of wave w's first entry; wave_start[n_waves] = nnz.
 * flags / bias / acc_* as amar_spmm_csr_f32 (AMAR_SPMM_RELU with AMAR_SPMM_BIAS gives the GCN
 * epilogue); Wnext != NULL additionally writes Hnext[i, 0:Cn] = Y[i, :] . Wnext (Cn <= 64) like
 * amar_gcn_layer_f32.  Each row is summed in ascending column order by a single lane.
 */
int amar_spmm_sj_f32(const int32_t *entries, const int16_t *counts, const int32_t *wave_start, int32_t n_slices,
                     const float *X, int64_t ldx, float *Y, int64_t ldy,
                     int32_t n_rows, int32_t F, uint32_t flags, const float *bias,
                     const float *acc_in, int64_t ld_acc_in, float *acc_out, int64_t ld_acc_out, float acc_div,
                     const float *Wnext, int32_t Cn, float *Hnext, int64_t ldhn, amar_stream_t stream);

/* The same product on the XCD-sliced (XS) image of a square A — the form used when the node table does
 * not fit one 4 MB per-XCD L2 (utilities/math.py:XcdSliced.from_csr builds it; any producer may):
 *   diag[n]                  the diagonal of A (duplicates summed)
 *   the off-diagonal non-zeros sorted by (slice, row, column), columns cut into n_slices contiguous slices of
 *   equal non-zero count:  colidx (= (row & 63) << 26 | column; n <= 2^26) / vals, and
 *   rowptr[k * n + r] = first entry of (slice k, row r)  (int32 [n_slices*n + 1])
 * Two launches: per-slice partial rows -> partials[n_slices, n, F] (caller-provided scratch), workgroup b touching
 * slice b % n_slices only (XCD <-> L2 affinity under round-robin dispatch); then
 *   Y[i] = epilogue( diag[i] . X[i] + sum_k partials[k][i] )   in slice order, epilogue as amar_spmm_sj_f32.
 * X has n_cols rows (the columns of A).  Xself[i] is row i's own feature row for the diag term: NULL means X (square A,
 * the rows index the same table); a row block of a larger matrix (multi-GPU node-range partition) passes X + offset.
 *
 * Value-free form (vals == NULL, row_scale != NULL) for A = S (C) S with S = diag(row_scale) and C a matrix of small
 * non-negative integers — exactly gcn_filter's D^-1/2 (A + I) D^-1/2 (Spektral, called at src/models/gnn.py:283,381):
 * every entry weighs 1 (an entry of C equal to c is stored c times), diag[i] = C_ii, X must already hold S . X
 * (row i scaled by row_scale[i]) and Y[i] = epilogue( row_scale[i] * (diag[i] . X[i] + sum_k partials[k][i]) ).
 * This halves the read-once index stream.  AMAR_SPMM_SCALE_NEXT also scales Hnext[i] by row_scale[i], so that a
 * chain of GCN layers stays in the pre-scaled form.  With vals != NULL row_scale must be NULL.
 */
int amar_spmm_xs_f32(const float *diag, const int32_t *rowptr, const int32_t *colidx, const float *vals, const float *row_scale,
                     int32_t n_slices, const float *X, int64_t ldx, int32_t n_cols, const float *Xself, float *partials, float *Y, int64_t ldy,
                     int32_t n_rows, int32_t F, uint32_t flags, const float *bias,
                     const float *acc_in, int64_t ld_acc_in, float *acc_out, int64_t ld_acc_out, float acc_div,
                     const float *Wnext, int32_t Cn, float *Hnext, int64_t ldhn, amar_stream_t stream);

/* The same value-free product on the LDS-tiled (LT) image (utilities/lds_tiled.py:LdsTiled builds it; any producer
 * may) — the form the 2-layer basic-gnn propagation (src/models/gnn.py:74-84 with GCNConv / LightGCNConv layers,
 * gnn.py:289-295, src/layers/lightgcn_conv.py:51-54) runs on when the node table exceeds the per-XCD L2s:
 * ONE launch, one 1024-thread workgroup per tile of consecutive rows, the tile's fp32 sums in 128 KB of LDS, the
 * tile's entries walked in column order so that neighbouring entries share L1 lines.  W = 16 waves,
 * RW = 128 KB / (4.F.W) LDS rows per wave, cbits = 31 - log2(RW) (n_cols <= 2^cbits, else AMAR_EUNSUPPORTED).
 *   tile_row0[n_tiles+1]   row range of every tile
 *   vstart[n_rows+1], vcount[n_tiles]   a row owns the virtual rows [vstart[i], vstart[i+1]) of its tile (the last row of
 *                          tile t up to vcount[t]); a tile has at most W.(RW-1) virtual rows.  Virtual row v belongs to
 *                          wave v % W and accumulates in LDS row (v/W).W + (v%W + v/W) % W.
 *   words                  one int32 per unit entry: flag << 31 | (v / W) << cbits | column; every (tile, wave) stream is
 *                          contiguous, 256-entry aligned and padded with (RW-1) << cbits.  Within a step (64/(F/4)
 *                          consecutive entries) a virtual row is read-modified-written once: a repeat in the next slot
 *                          (same 16-lane DPP row, flag 0) is folded into its neighbour in registers; any other repeat
 *                          carries flag = 1 and is added with an LDS atomic after the step.
 *   stream_start[n_tiles*W], wsteps[n_tiles][W][maxwin1], n_win[n_tiles]: see utilities/lds_tiled.py
 *   pace_every             the tile's waves meet at a barrier after every pace_every-th window (a power of two >= 1)
 * X (n_cols rows) must hold S.X; Y[i] = epilogue( row_scale[i] . (diag[i] . Xself[i] + sum of the row's entries) ),
 * flags / bias / acc_* / Wnext / AMAR_SPMM_SCALE_NEXT as amar_spmm_xs_f32.  The summation order of a row is fixed by
 * the image, so results are bitwise reproducible run to run (they differ from the XS / CSR forms in the last bits).
 */
int amar_spmm_lt_f32(const int32_t *words, const int32_t *stream_start, const int32_t *wsteps, const int32_t *tile_row0,
                     const int32_t *n_win, const int32_t *vstart, const int32_t *vcount, int32_t n_tiles, int32_t maxwin1, int32_t pace_every,
                     const float *diag, const float *row_scale,
                     const float *X, int64_t ldx, int32_t n_cols, const float *Xself,
                     float *Y, int64_t ldy, int32_t n_rows, int32_t F, uint32_t flags, const float *bias,
                     const float *acc_in, int64_t ld_acc_in, float *acc_out, int64_t ld_acc_out, float acc_div,
                     const float *Wnext, int32_t Cn, float *Hnext, int64_t ldhn, amar_stream_t stream);

/* One fused GCN layer (src/models/gnn.py:289-295 + gnn.py:78, Spektral GCNConv.call):
 *     Y[i, 0:C]      = ReLU( sum_j A_hat[i,j] . H[j, 0:C] + bias )      H = X_prev . W  (pre-multiplied)
 *     Hnext[i, 0:Cn] = Y[i, :] . Wnext[C, Cn]                            (only if Wnext != NULL)
 * so that layer l's epilogue performs layer l+1's dense product and each layer is ONE kernel.
 * C in {4,8,16,32,64}; Cn <= 64.  Other multiples of 4 run as column chunks, without the fused next product
 * (Wnext must be NULL for them: AMAR_EUNSUPPORTED otherwise).
 */
int amar_gcn_layer_f32(const int32_t *rowptr, const int32_t *colidx, const float *vals,
                       const float *H, int64_t ldh, int32_t C, const float *bias,
                       float *Y, int64_t ldy,
                       const float *Wnext, int32_t Cn, float *Hnext, int64_t ldhn,
                       int32_t n_rows, amar_stream_t stream);

/* Row-wise small dense product used as the GNN prologue (Keras `K.dot(x, kernel)` inside
 * GCNConv / GATConv; reference call site src/models/gnn.py:78):
 *     H[i, 0:C] = X[i, 0:F] . W[F, C]                     F, C <= 64
 *     copy_to != NULL:  copy_to[i, 0:F] = X[i, 0:F]       (X_0 slice of the concat buffer)
 *     a_self/a_neigh != NULL (GAT):  s_self[i] = H[i,:].a_self,  s_neigh[i] = H[i,:].a_neigh
  * row_scale != NULL (not together with the attention scalars): H[i] is multiplied by row_scale[i] — the pre-scaled table
 * the value-free form of amar_spmm_xs_f32 gathers from.
 */
int amar_rowwise_xw_f32(const float *X, int64_t ldx, int32_t F, const float *W, int32_t C,
                        float *H, int64_t ldh, float *copy_to, int64_t ld_copy,
                        const float *a_self, const float *a_neigh, float *s_self, float *s_neigh,
                        const float *row_scale, int32_t n_rows, amar_stream_t stream);

/* The same product with a row gather:  H[p, 0:C] = row_scale[p] . ( X[row_ids[p], 0:F] . W[F, C] ),  p < n_rows
 * (row_scale NULL: 1; row_ids[p] < 0: a zero row).  The X_0 . W_1 prologue of the SAME call site (src/models/gnn.py:78) when
 * the gathered table is kept in the rank-major block layout of a node-range partition (parallel.py:TypedPartition): X is the
 * trainable node table in the reference's id order (users | items [| properties], src/data/loaders.py:43-68), row_ids maps
 * every row of the block layout to its node (-1 for the padding rows at the end of a type's last block). */
int amar_rowwise_xw_gather_f32(const float *X, int64_t ldx, int32_t F, const int32_t *row_ids, const float *W, int32_t C,
                               float *H, int64_t ldh, const float *row_scale, int32_t n_rows, amar_stream_t stream);

/* One GraphSAGE-mean layer (Spektral 1.x GraphSageConv, built at src/models/gnn.py:354-361):
 *     agg_i = ( [self_loop] X_i + sum_{j in N(i)} X_j ) / ( [self_loop] 1 + |N(i)| )
 *     Y_i   = ReLU( l2_normalize( [X_i || agg_i] . W[2F, C] + bias ) )
 * rowptr/colidx hold the raw symmetric adjacency with duplicate edges kept and no diagonal;
 * edge values are ignored, as in the reference.  F in {4,8,16,32}; C <= 64.
 */
int amar_sage_layer_f32(const int32_t *rowptr, const int32_t *colidx,
                        const float *X, int64_t ldx, int32_t F,
                        const float *W, const float *bias, int32_t C,
                        float *Y, int64_t ldy, int32_t self_loop,
                        int32_t n_rows, amar_stream_t stream);

/* The tail of the same layer when the mean aggregate AGG[n_rows, F] was produced by an SpMM (amar_spmm_xs_f32 on the
 * XCD-sliced mean image for large graphs; amar_spmm_csr_f32 + amar_row_affine_f32 for widths the fused kernel is not
 * instantiated for):  Y = relu(l2_normalize([X || AGG] . W + bias)), W [2F, C] row-major.  F, C multiples of 4, <= 64. */
int amar_sage_tail_f32(const float *X, int64_t ldx, const float *AGG, int64_t lda, int32_t F,
                       const float *W, const float *bias, int32_t C, float *Y, int64_t ldy,
                       int64_t n_rows, amar_stream_t stream);

/* One GAT layer, 1 head (Spektral 1.x GATConv._call_single, built at src/models/gnn.py:321-328):
 *     e_ij  = LeakyReLU_0.2( s_self[i] + s_neigh[j] ),  j in N(i) (+ i itself if self_loop)
 *     alpha = exp(e_ij - max_j e_ij) / ( sum_j exp(e_ij - max_j e_ij) + 1e-9 )
 *     Y_i   = ReLU( sum_j alpha_ij H_j + bias )
 * H, s_self, s_neigh come from amar_rowwise_xw_f32.  C in {4,8,16,32,64}; other multiples of 4 run as column chunks
 * (the attention coefficients only depend on the per-node scalars).
 */
int amar_gat_layer_f32(const int32_t *rowptr, const int32_t *colidx,
                       const float *H, int64_t ldh, int32_t C,
                       const float *s_self, const float *s_neigh, const float *bias,
                       float *Y, int64_t ldy, int32_t self_loop,
                       int32_t n_rows, amar_stream_t stream);

/* ---- scoring head -------------------------------------------------------------------------
 * Y[M, N] = act( X[M, K] . W[K, N] + bias[N] )   fp32 MFMA GEMM (Keras Dense; src/models/dense.py:4-17)
 * ids != NULL gathers the input rows first: row m of the product reads X[ids[m], :]
 * (tf.nn.embedding_lookup at src/models/basic.py:73-74, src/models/hybrid.py:138-139).
 * Y is written at column offset 0 of a matrix with leading dimension ldy, so two calls with
 * Y and Y + N realise `Concatenate` (src/models/basic.py:35, src/layers/fusion.py:51-53).
 */
int amar_dense_f32(const float *X, int64_t ldx, const int32_t *ids,
                   const float *W, const float *bias, float *Y, int64_t ldy,
                   int64_t M, int32_t K, int32_t N, int32_t act, amar_stream_t stream);

/* The same layer with its products on the bf16 matrix instruction and BOTH operands split three ways (x = hi + mid + lo
 * exactly, six part products accumulated in f32: as accurate as the f32 instruction, 2.7 times fewer matrix-pipe cycles) for
 * the wide layers of the content towers (768 -> 256 of src/models/hybrid.py:52-58).  K % 32 == 0, N % 128 == 0, ldx % 4 == 0,
 * 16-byte aligned X; other shapes return AMAR_EUNSUPPORTED (use amar_dense_f32).  Wq is the layer's kernel pre-split by
 * amar_dense_split_pack_f32 (HOST in, HOST out of amar_dense_split_bytes(K, N) bytes — once per weight update), copied to
 * the device by the caller.
 * PRECONDITION: finite inputs.  The split x = hi + mid + lo is exact for finite x only; an infinite activation or weight gives
 * mid = Inf - Inf = NaN, so a product that the f32 instruction would return as +-Inf (or saturate) comes back NaN here.  NaN
 * inputs propagate as NaN in both forms.  The same holds for the split-product forms of amar_chain_f32 / amar_dual_chain_f32
 * (the default of the pair stages; AMAR_PAIR_MFMA=f32 / AMAR_DENSE_SPLIT=0 select the f32 instruction). */
int64_t amar_dense_split_bytes(int32_t K, int32_t N);
int amar_dense_split_pack_f32(const float *W, int32_t K, int32_t N, void *out);
int amar_dense_split_f32(const float *X, int64_t ldx, const int32_t *ids, const void *Wq, const float *bias,
                         float *Y, int64_t ldy, int64_t M, int32_t K, int32_t N, int32_t act, amar_stream_t stream);

/* Fused gather + Concatenate + Dense stack (BasicRS / HybridCBRS towers and classifiers:
 * src/models/basic.py:31-37,72-75, src/models/hybrid.py:72-89, src/models/dense.py:4-17).
 * For every row p < P:
 *     x = [ A[ida(p), 0:Da] || B[idb(p), 0:Db] ],  ida(p) = ids_a ? ids_a[p] - base_a : p   (same for B)
 *         or, with sum_inputs != 0 (Da == Db):  x = in_act( A[ida(p)] + B[idb(p)] ) — the form a Dense layer over a
 *         concatenation takes once its two halves have been applied per entity (x.W = u.W[:d] + i.W[d:])
 *     x = act_l( x . W_l + b_l )  for l = 0 .. n_layers-1      dims[0] = Da + Db (Da if sum_inputs), dims[l+1] = units of layer l
 * out[p, 0:dims[n_layers]] = x.  A trailing 1-unit layer (the sigmoid scorer) is evaluated as a
 * dot product and written to out[p * ldo].  Activations stay in registers between layers
 * (fp32 MFMA 16x16x4); weights come pre-packed in fragment order from amar_chain_pack_f32 and
 * stay in LDS.  Limits: every width <= 128, Da and Db multiples of 4, <= 8 layers; shapes
 * outside them return AMAR_EUNSUPPORTED (use amar_dense_f32 layer by layer instead).
 * amar_chain_pack_floats / amar_chain_pack_f32 run on the HOST (host pointers): kernels[l] is the
 * row-major [dims[l], dims[l+1]] Keras kernel, biases[l] its bias; the blob is then copied to
 * the device by the caller.
 * Arithmetic: f32 values and f32 sums throughout.  The pair-stage form (sum_inputs with both id lists, ReLU, equal
 * layer widths of 48 or 64, a trailing 1-unit layer) and amar_dual_chain_f32's 64-wide form take their PRODUCTS on
 * the bf16 matrix instruction with both operands split into three bf16 parts (x = hi + mid + lo exactly; six part
 * products accumulated in f32): a term x.w is off by at most 3 * 2^-24 |x.w| — as close to a float64 evaluation as the
 * f32 instruction, not bit-identical to it.  Environment AMAR_PAIR_MFMA=f32 keeps v_mfma_f32_16x16x4_f32 everywhere.
 */
int64_t amar_chain_pack_floats(const int32_t *dims, int32_t n_layers);
int amar_chain_pack_f32(const float *const *kernels, const float *const *biases, const int32_t *dims,
                        int32_t n_layers, float *out);
int amar_chain_f32(const float *A, int64_t lda, int32_t Da, const int32_t *ids_a, int32_t base_a,
                   const float *B, int64_t ldb, int32_t Db, const int32_t *ids_b, int32_t base_b,
                   int32_t sum_inputs, int32_t in_act,
                   const float *wpack, const int32_t *dims, const int32_t *acts, int32_t n_layers,
                   float *out, int64_t ldo, int64_t P, amar_stream_t stream);

/* The same with an output index: row p of the chain is written to out row out_index[p] (int32 [P], a permutation of
 * 0..P-1, or NULL = amar_chain_f32).  For a pair list the caller keeps in another order than the one the scores are
 * wanted in — the pair stage of a predict pass (src/experiment.py:197, the test Sequence's order, datasets.py:199-203)
 * bucketed ONCE per dataset by item range so that the workgroups of one XCD (pairs p with (p >> 7) % 8 equal, under
 * round-robin dispatch) gather item-tower rows of one eighth of the items and find them in that XCD's L2: the list is
 * constant across steps and epochs, the scores still land in the Sequence's order.
 */
int amar_chain_indexed_f32(const float *A, int64_t lda, int32_t Da, const int32_t *ids_a, int32_t base_a,
                           const float *B, int64_t ldb, int32_t Db, const int32_t *ids_b, int32_t base_b,
                           int32_t sum_inputs, int32_t in_act,
                           const float *wpack, const int32_t *dims, const int32_t *acts, int32_t n_layers,
                           float *out, int64_t ldo, const int32_t *out_index, int64_t P, amar_stream_t stream);

/* The per-entity tower form of the chain (one input table, no second table, no trailing 1-unit layer) reading the
 * `Concatenate` of ReductionLayer('concatenation') (src/layers/reduction.py:15-17; src/models/gnn.py:84) IN PLACE:
 *     x = [ seg[0][r, 0:w_0] || seg[1][r, 0:w_1] || ... ],   r = ids ? ids[p] - base : p       (HOST arrays of n_seg entries)
 * so that the [N, d(L+1)] table never has to be assembled — each X_l stays where its layer (or the all-gather of a
 * node-range partition) left it.  Widths are multiples of 4, their sum <= 128, n_seg <= 8.  Only stacks with a
 * compile-time tower shape run this way (ReLU layers with an optionally linear last one, every width <= 64, at most three
 * layers: the towers of every econfig of the reference); other shapes return AMAR_EUNSUPPORTED — the caller then copies the
 * columns together (amar_copy_columns_f32) and calls amar_chain_f32.  Same arithmetic as amar_chain_f32 on the assembled
 * table, bit for bit. */
int amar_chain_segments_f32(const float *const *seg, const int64_t *seg_ld, const int32_t *seg_width, int32_t n_seg,
                            const int32_t *ids, int32_t base,
                            const float *wpack, const int32_t *dims, const int32_t *acts, int32_t n_layers,
                            float *out, int64_t ldo, int64_t P, amar_stream_t stream);

/* Fused two-branch scorer for the hybrid head (src/models/hybrid.py:72-89) once the first Dense layers of
 * dense3a / dense3b have been folded into the per-entity tables:
 *     x_b  = in_act( A[b][ida_b(p)] + B[b][idb_b(p)] )            b = 0, 1;  [P, D]
 *     x_b  = act_l( x_b . W_bl + b_bl )                            n_branch layers D -> D per branch
 *     out  = trunk( [x_0 || x_1] )                                 Dense stack trunk_dims[0] = 2D, equal hidden widths, last = 1 unit
 * A, lda, ida, base_a (and B, ...) are HOST arrays of two entries (device pointers inside).  wpack = the
 * amar_chain_pack_f32 blobs of branch 0 (dims [D, D, ..]), branch 1 and the trunk, concatenated on the device.
 * D % 16 == 0, D <= 64, trunk hidden widths <= 64; other shapes return AMAR_EUNSUPPORTED (use amar_chain_f32).
 */
int amar_dual_chain_f32(const float *const *A, const int64_t *lda, const int32_t *const *ida, const int32_t *base_a,
                        const float *const *B, const int64_t *ldb, const int32_t *const *idb, const int32_t *base_b,
                        int32_t D, int32_t in_act, int32_t n_branch, const int32_t *branch_acts,
                        const int32_t *trunk_dims, const int32_t *trunk_acts, int32_t n_trunk,
                        const float *wpack, float *out, int64_t ldo, int64_t P, amar_stream_t stream);
/* The same on a pair list kept in another order (pair p is written to out[out_index[p] * ldo]; out_index == NULL: in place),
 * like amar_chain_indexed_f32. */
int amar_dual_chain_indexed_f32(const float *const *A, const int64_t *lda, const int32_t *const *ida, const int32_t *base_a,
                                const float *const *B, const int64_t *ldb, const int32_t *const *idb, const int32_t *base_b,
                                int32_t D, int32_t in_act, int32_t n_branch, const int32_t *branch_acts,
                                const int32_t *trunk_dims, const int32_t *trunk_acts, int32_t n_trunk,
                                const float *wpack, float *out, int64_t ldo, const int32_t *out_index, int64_t P,
                                amar_stream_t stream);

/* Concatenate / ReductionLayer as layout operations (src/layers/reduction.py:15-33,
 * src/layers/fusion.py:51-53): copy a [n_rows, width] block between two strided matrices
 * (row r of dst reads row ids[r] - base of src when ids != NULL: tf.nn.embedding_lookup), and
 * out = X_0 + X_1 + ... (+ division by n_layers for 'mean') over the n_layers equal-width column
 * blocks of a concatenation buffer, added in layer order like tf.add_n.
 */
int amar_copy_columns_f32(const float *src, int64_t lds, const int32_t *ids, int32_t base,
                          float *dst, int64_t ldd, int64_t n_rows, int32_t width, amar_stream_t stream);
int amar_reduce_layers_f32(const float *cat, int64_t ld, int32_t n_layers, int32_t width, float *out, int64_t ldo,
                           int64_t n_rows, int32_t mean, amar_stream_t stream);

/* dst[index[t] * ldd] = src[t] for t in [0, n): Keras `predict` returns the scores in the order of the Sequence's pairs
 * (src/data/datasets.py:199-213 — the test Sequence is not shuffled, its order is the file's), while the pair stage walks
 * a list prepared for its gathers; this is the second half of the way back.  The positions are visited window by window
 * (window w = [window_off[w], window_off[w + 1]), device array of n_windows + 1 entries; NULL: n_windows equal slices),
 * all workgroups of one XCD in the same window: prepared so that a window's destinations are a narrow range of dst, its
 * lines are completed in that XCD's L2 and leave it whole.  index must be a partial permutation (no two t with the same
 * destination); n < 2^31. */
int amar_scatter_f32(const float *src, const int32_t *index, float *dst, int64_t ldd, int64_t n,
                     const int32_t *window_off, int32_t n_windows, amar_stream_t stream);

/* ReductionLayer('w-sum') = WeightedSum (src/layers/reduction.py:36-55): out = sum over the n_layers column blocks of
 * (w[l] * w[l]) * X_l with a learnable device vector w [n_layers] (initialised to ones); products rounded, then added in layer order.
 * n_layers <= 8.  The reverse pass writes d_cat[:, block l] = w[l]^2 d_out and dw[l] = 2 w[l] sum(d_out . X_l), the sums formed per
 * workgroup of a fixed grid and added in workgroup order (no float atomics); `scratch` holds amar_reduce_layers_wsum_bwd_scratch()
 * floats.
 */
int amar_reduce_layers_wsum_f32(const float *cat, int64_t ld, int32_t n_layers, int32_t width, const float *w, float *out, int64_t ldo,
                                int64_t n_rows, amar_stream_t stream);
int64_t amar_reduce_layers_wsum_bwd_scratch(void);
int amar_reduce_layers_wsum_bwd_f32(const float *cat, int64_t ld, int32_t n_layers, int32_t width, const float *w,
                                    const float *d_out, int64_t ldd, float *d_cat, int64_t ld_dcat, float *dw, float *scratch,
                                    int64_t n_rows, amar_stream_t stream);

/* The same GAT layer on the XCD-sliced image of the (square) edge-list adjacency, for graphs whose node table exceeds the
 * per-XCD L2s: rowptr / colidx as in amar_spmm_xs_f32 (values unused).  `packed` is scratch [n, 2C] floats that the call
 * fills with [ H | s_neigh | 0 .. ] rows (one L2 request then serves the neighbour's features and its scalar), `partials`
 * scratch [n_slices, n, 2C].  The segment softmax stays exact through (max, sum, weighted sum) triples merged per
 * (row, slice) and across slices.  C = 8.  H, s_self, s_neigh and `packed` have n_cols rows (the columns of the image);
 * a row block of a larger graph (multi-GPU partition) has n_rows < n_cols and its row i is node row_offset + i.
 */
int amar_gat_xs_f32(const int32_t *rowptr, const int32_t *colidx, int32_t n_slices,
                    const float *H, int64_t ldh, int32_t C, const float *s_self, const float *s_neigh, const float *bias,
                    float *packed, float *partials, float *Y, int64_t ldy, int32_t self_loop, int32_t n_rows,
                    int32_t n_cols, int32_t row_offset, amar_stream_t stream);

/* The same GAT layer (Spektral GATConv as instantiated at src/models/gnn.py:321-328) on the LDS-tiled image of the
 * edge-list adjacency (layout as amar_spmm_lt_f32; every edge a unit entry, duplicates repeated), C = 8, 16 or 32.
 * The LDS row of a virtual row holds (sum w.h [C], sum w, s_self of its row): RW = amar_gat_lt_rows_per_wave(C) rows per
 * wave (216 / 124 / 64), cbits = 31 - ceil(log2(RW)); `rows_per_wave` states the RW the image was cut for and must equal that
 * value (AMAR_EINVAL otherwise: a taller tile would index past the workgroup's LDS).  An entry (i, j) weighs
 *     w_ij = exp( LeakyReLU_0.2(s_self[i] + s_neigh[j]) - M_i ),   M_i = LeakyReLU_0.2(s_self[i] + *s_neigh_max)
 * with *s_neigh_max >= every s_neigh (amar_colmax_f32 writes it): M_i bounds the row's maximum, so no running maximum is
 * needed and the weights add up like the plain sum's entries; out_i = ReLU( (sum_j w_ij H_j) / (sum_j w_ij) + bias ) is
 * Spektral's max-subtracted softmax up to rounding (its +1e-9 in the denominator, >= 1 there, is below fp32 resolution).
 * A row whose weight sum stays below e^-60 (its own maximum lies more than 60 under the bound), and every empty row, is
 * recomputed in the same launch from rowptr / colidx (the CSR of the same rows and columns, duplicates kept) with the row's true
 * maximum and the 1e-9, exactly as amar_gat_layer_f32 does.  diag[i] = number of (i, i) edges in the list (they, and the added
 * self loop when self_loop != 0, enter as the row's self term).  H, s_self, s_neigh cover the n_cols columns; row i of a row
 * block (multi-GPU partition) is node row_offset + i.
 */
int amar_gat_lt_f32(const int32_t *words, const int32_t *stream_start, const int32_t *wsteps, const int32_t *tile_row0,
                    const int32_t *n_win, const int32_t *vstart, const int32_t *vcount, int32_t n_tiles, int32_t maxwin1, int32_t pace_every,
                    int32_t rows_per_wave, const float *diag, const int32_t *rowptr, const int32_t *colidx,
                    const float *H, int64_t ldh, int32_t C, const float *s_self, const float *s_neigh, const float *s_neigh_max,
                    const float *bias, float *Y, int64_t ldy, int32_t self_loop, int32_t n_rows, int32_t n_cols, int32_t row_offset,
                    amar_stream_t stream);
int amar_gat_lt_rows_per_wave(int32_t C);
/* out[0] = max(x[0..n)) (-inf for n = 0), reset and folded in-stream: the bound amar_gat_lt_f32 takes. */
int amar_colmax_f32(const float *x, int64_t n, float *out, amar_stream_t stream);

/* ---- hybrid-head variants of econfigs/hybrid-gnn-tweaks*.yaml -----------------------
 * amar_attention_mix_f32      FusionLayer('attention') (src/layers/fusion.py:54-68) after the two products
 *                             TA = A . att_weight, TB = B . att_weight (amar_dense_f32, no bias): the softmax over the two
 *                             stacked sources is per feature wa = sigmoid(tanh(TA) - tanh(TB)); out = wa*A + (1-wa)*B
 * amar_attention_mix_bwd_f32  its reverse: dA, dB = the direct paths, dTA, dTB = gradients of the two products
 *                             (all four contiguous [M, D])
 * amar_add3_act_f32           out = act(A + B + C): the residual head, activation(residual(x) + x1 + x2)
 *                             (src/models/hybrid.py:86-89)
 * amar_locality_scale_f32     DGCFConv's LocalityAdaptive (src/layers/dgcf_conv.py:83-102): out = X * sigmoid(w[row]);
 *                             the layer is then amar_spmm_* on the DGCF adjacency (dgcf_conv.py:32-36)
 * amar_locality_scale_bwd_f32 dX (+)= dOut * sigmoid(w), dw[row] = sigmoid'(w[row]) * (dOut[row] . X[row])
 */
int amar_attention_mix_f32(const float *A, int64_t lda, const float *B, int64_t ldb, const float *TA, int64_t ldta,
                           const float *TB, int64_t ldtb, float *out, int64_t ldo, int64_t M, int32_t D, amar_stream_t stream);
int amar_attention_mix_bwd_f32(const float *dOut, int64_t ldd, const float *A, int64_t lda, const float *B, int64_t ldb,
                               const float *TA, int64_t ldta, const float *TB, int64_t ldtb,
                               float *dA, float *dB, float *dTA, float *dTB, int64_t M, int32_t D, amar_stream_t stream);
int amar_add3_act_f32(const float *A, int64_t lda, const float *B, int64_t ldb, const float *C, int64_t ldc, float *out, int64_t ldo,
                      int64_t M, int32_t W, int32_t act, amar_stream_t stream);
int amar_locality_scale_f32(const float *X, int64_t ldx, const float *w, float *out, int64_t ldo, int64_t M, int32_t W,
                            amar_stream_t stream);
int amar_locality_scale_bwd_f32(const float *dOut, int64_t ldd, const float *X, int64_t ldx, const float *w, float *dX, int64_t lddx,
                                float *dw, int64_t M, int32_t W, int32_t accumulate, amar_stream_t stream);

/* ---- training step -------------------------------------------------------
 * What Keras' fit() adds around the forward path for one batch (src/experiment.py:155-188, config.yaml:50-58):
 * reverse-mode derivatives of Dense / GCNConv / LightGCNConv / embedding_lookup, binary cross-entropy,
 * L2 regularisers (src/models/gnn.py:45,293-294) and the Adam update.  The forward kernels above are reused
 * (A_hat is symmetric, so the SpMM is its own transpose; dX = dZ . W^T is amar_dense_f32 on the transposed kernel).
 *
 * amar_act_bwd_f32          dZ = dY * act'(Y)        (Y = the layer's OUTPUT; relu / sigmoid / none)
 * amar_wgrad_f32            dW[K,N] = X^T . dZ and/or db[N] = column sums of dZ, reduced in two stages in a fixed
 *                           order (no float atomics); scratch must hold amar_wgrad_scratch_floats(M, K, N) floats
 * amar_bce_grad_f32         Keras backend binary_crossentropy on probabilities (epsilon 1e-7, mean over B):
 *                           loss_terms[i] and dz[i] = dL/dlogit_i through the final sigmoid
 * amar_scatter_add_rows_f32 dst[ids[m] - base, :] += src[m, :]   (gradient of embedding_lookup).  Up to 8 192 ids: without atomics —
 *                           the first position of an id adds the rows of all its positions in position order (one writer per row,
 *                           reproducible bit for bit); longer lists: global float atomics (order-dependent last bits)
 * amar_add_inplace_f32      dst += scale * src on strided [M, W] blocks
 * amar_row_affine_f32       out = (A + B) * scale[row], B optional: GraphSAGE's mean aggregate (sum + self) / count
 *                           (Spektral GraphSageConv, built at src/models/gnn.py:354-361) and its reverse
 * amar_l2norm_fwd_f32       tf.nn.l2_normalize(axis=-1) + activation as GraphSageConv applies them: inv[r] =
 *                           rsqrt(max(sum z^2, 1e-12)), Nrm = z * inv (kept for the reverse pass), Y = act(Nrm)
 * amar_l2norm_bwd_f32       dZ = inv * (dn - Nrm * (Nrm . dn)) with dn = dY * act'(Nrm); dZ = inv * dn where the norm
 *                           was clamped
 * amar_gat_bwd_f32          reverse of amar_gat_layer_f32 (Spektral GATConv, src/models/gnn.py:321-328) given dY = dL/dY:
 *                           dout = dY * [Y > 0] ([n, C] contiguous, also the source of the bias gradient), ds / dt = the
 *                           gradients of the two attention scalars per node, dH = dL/dH including their ds (x) a_self +
 *                           dt (x) a_neigh terms.  row_scratch: 3 * n_rows floats.  Row-wise sums only (no float atomics):
 *                           relies on the edge multiset being symmetric, as build_adjacency_matrix + symmetrize_matrix
 *                           produce it (src/data/preprocess.py:44-170, src/utilities/math.py:6-21).  C in {4,8,16,32,64}.
 * amar_transpose_f32        dst[N,K] = src[K,N]^T
 * amar_adam_f32             keras.optimizers.Adam on a flat parameter: g' = g + 2*l2*w; m, v moments; lr_t = the
 *                           bias-corrected step lr * sqrt(1 - b2^t) / (1 - b1^t);  w -= lr_t * m / (sqrt(v) + epsilon)
 * amar_adam_advance_f32     state[0] = t + 1, state[1] = lr_t for that t (device memory, 2 floats): the step counter of
 *                           keras.optimizers.Adam (`iterations`) kept on the device ...
 * amar_adam_dev_f32         ... and the same update as amar_adam_f32 reading lr_t from state[1], so that a whole batch
 *                           (forward, reverse pass, optimizer) can be captured once as a hipGraph and replayed
 */
int amar_act_bwd_f32(const float *dY, int64_t ldd, const float *Y, int64_t ldy, float *dZ, int64_t ldz,
                     int64_t M, int32_t N, int32_t act, amar_stream_t stream);
int64_t amar_wgrad_scratch_floats(int64_t M, int32_t K, int32_t N);
/* A whole Dense stack forward in ONE launch with every layer's output kept (what model.fit's forward pass needs of a tower / classifier:
 * src/models/dense.py:4-17 called from src/models/basic.py:31-37): y_0 = X[ids] (ids == NULL: X), y_{l+1} = act_l(y_l . W_l + b_l) written to
 * Y[l] (leading dimension ldy[l]: a column slice of a wider buffer realises `Concatenate`), l < n_layers <= 4, every width <= 128
 * (else AMAR_EUNSUPPORTED: layer by layer with amar_dense_f32).  W, bias, Y, ldy, dims (n_layers + 1 widths), acts: HOST arrays of device
 * pointers / values.  Xcopy != NULL: the gathered input rows are also written there (the reverse pass multiplies by them). */
int amar_dense_stack_f32(const float *X, int64_t ldx, const int32_t *ids, float *Xcopy, int64_t ldxc, int32_t n_layers,
                         const float *const *W, const float *const *bias, const int32_t *dims, const int32_t *acts,
                         float *const *Y, const int64_t *ldy, int64_t M, amar_stream_t stream);
/* Two INDEPENDENT stacks in one launch — the user and the item tower of src/models/basic.py:31-35 inside model.fit: each is 16 workgroups
 * of a 1 024-pair batch and ~18 us as a launch of its own, and a training batch at ML-1M size is a chain of such latencies.  A descriptor
 * holds the arguments of amar_dense_stack_f32 (same meaning, same limits, same error codes); the results are those of two separate calls. */
typedef struct amar_dense_stack_desc {
    const float *X; int64_t ldx; const int32_t *ids; float *Xcopy; int64_t ldxc; int32_t n_layers;
    const float *const *W; const float *const *bias; const int32_t *dims; const int32_t *acts; float *const *Y; const int64_t *ldy; int64_t M;
} amar_dense_stack_desc;
int amar_dense_stack_pair_f32(const amar_dense_stack_desc *s0, const amar_dense_stack_desc *s1, amar_stream_t stream);
/* The reverse pass of a whole Dense stack (a tower / the classifier of src/models/basic.py:11-37 inside model.fit) in ONE launch: from
 * dYtop = d(loss)/d(last output) (Ytop = that output; Ytop == NULL: dYtop is already the last pre-activation's gradient) down to
 * dX0 = d(loss)/d(stack input) (or NULL), leaving every layer's dW[l] [K_l, N_l] and db[l] [N_l].  X[l] = layer l's input (X[l+1] is layer
 * l's output), W, dims (n_layers + 1 widths), acts as in amar_dense_stack_f32; n_layers <= 4, widths <= 128, M <= 4 096 rows (else
 * AMAR_EUNSUPPORTED: layer by layer).  workspace: amar_dense_stack_bwd_workspace_floats(M, n_layers, dims) floats of scratch.
 * flags & AMAR_DENSE_BWD_DEFER: dW / db are not written; layer l's partials stay at  workspace + 4 + sum_{j<l} G (K_j N_j + N_j):
 * [G][K_l N_l] then [G][N_l],  G = amar_dense_stack_bwd_groups(M)  (amar_adam_multi_f32 with g_groups = G adds them). */
int64_t amar_dense_stack_bwd_groups(int64_t M);      /* G: the partials per layer a deferred call leaves (one per workgroup: 16 rows each up to M = 1 024, else 64) */
int64_t amar_dense_stack_bwd_workspace_floats(int64_t M, int32_t n_layers, const int32_t *dims);
int amar_dense_stack_bwd_f32(const float *dYtop, int64_t lddy, const float *Ytop, int64_t ldytop, int32_t n_layers,
                             const float *const *X, const int64_t *ldx, const float *const *W, const int32_t *dims, const int32_t *acts,
                             float *dX0, int64_t lddx0, float *const *dW, float *const *db, float *workspace, int32_t flags,
                             int64_t M, amar_stream_t stream);
/* ... and the reverse passes of two independent stacks in one launch (descriptor = the arguments of amar_dense_stack_bwd_f32; each stack
 * with its own workspace; the partial sums of a stack without AMAR_DENSE_BWD_DEFER in `flags` are added by launches behind the shared one). */
typedef struct amar_dense_stack_bwd_desc {
    const float *dYtop; int64_t lddy; const float *Ytop; int64_t ldytop; int32_t n_layers;
    const float *const *X; const int64_t *ldx; const float *const *W; const int32_t *dims; const int32_t *acts;
    float *dX0; int64_t lddx0; float *const *dW; float *const *db; float *workspace; int32_t flags; int64_t M;
} amar_dense_stack_bwd_desc;
int amar_dense_stack_bwd_pair_f32(const amar_dense_stack_bwd_desc *s0, const amar_dense_stack_bwd_desc *s1, amar_stream_t stream);
/* The reverse pass of ONE Dense layer (Keras Dense inside model.fit: src/models/dense.py:4-17, src/experiment.py:183-188) in two launches
 * instead of four:
 *     dZ = dY * act'(Y)   (Y = the layer's OUTPUT; act == AMAR_ACT_NONE or Y == NULL: dY already is dZ)
 *     dX[M, K] = dZ . W^T (dX == NULL: skipped)      dW[K, N] = X^T . dZ (dW == NULL: skipped)      db[N] = column sums of dZ (or NULL)
 * Both products on the f32 matrix instruction; the workgroups (one per 64 rows) leave partial weight / bias gradients in the workspace, which
 * the second launch adds in workgroup order (a FIXED order: no float atomics, results reproducible bit for bit).  Past 64 workgroups (M >
 * 4 096: the reverse pass of a convolution layer runs over every node of the graph) operands of at most 32 columns take a row-walking
 * kernel instead of the tile kernel, and a launch in between folds the raw partials, in workgroup order, into at most 64 (G below).
 * K, N <= 128 (wider layers: AMAR_EUNSUPPORTED — use amar_act_bwd_f32 + amar_wgrad_f32 + amar_dense_f32 with AMAR_DENSE_WT).
 * workspace: amar_dense_bwd_workspace_floats(M, K, N) floats owned by the caller (scratch: any contents); two calls in flight on
 * different streams must not share one.
 * act | AMAR_DENSE_BWD_DEFER: the second launch is left out — dW / db (still non-NULL to request them) are NOT written; the partials stay
 * in the workspace as  workspace + 4: [G][K * N] (if dW)  then [G][N] (if db),  G = amar_dense_bwd_groups(M), for a consumer that adds them
 * itself (amar_adam_multi_f32 with g_groups = G). */
#define AMAR_DENSE_BWD_DEFER 0x100
/* act | AMAR_DENSE_BWD_ACCUM_DX: dX += dZ . W^T instead of dX = (a layer whose input already carries a gradient: the concat slices of a
 * convolution stack).  dZ != NULL: the pre-activation gradient dZ itself is also written ([M, N], leading dimension lddz) — a GCN layer
 * multiplies it by A_hat before the weight gradient — so that act', its bias gradient and dZ are one launch (X, W, dX, dW all NULL). */
#define AMAR_DENSE_BWD_ACCUM_DX 0x200
int64_t amar_dense_bwd_groups(int64_t M);
int64_t amar_dense_bwd_workspace_floats(int64_t M, int32_t K, int32_t N);
int amar_dense_bwd_f32(const float *X, int64_t ldx, const float *Y, int64_t ldy, const float *dY, int64_t lddy, const float *W,
                       int32_t act, float *dX, int64_t lddx, float *dW, float *db, float *dZ, int64_t lddz, float *workspace,
                       int64_t M, int32_t K, int32_t N, amar_stream_t stream);
int amar_wgrad_f32(const float *X, int64_t ldx, const float *dZ, int64_t ldz, int64_t M, int32_t K, int32_t N,
                   float *dW, float *db, float *scratch, amar_stream_t stream);
int amar_bce_grad_f32(const float *p, int64_t ldp, const float *y, float *dz, float *loss_terms, int64_t B, amar_stream_t stream);
int amar_scatter_add_rows_f32(const float *src, int64_t lds, const int32_t *ids, int32_t base, float *dst, int64_t ldd,
                              int64_t M, int32_t W, amar_stream_t stream);
int amar_add_inplace_f32(float *dst, int64_t ldd, const float *src, int64_t lds, int64_t M, int32_t W, float scale, amar_stream_t stream);
int amar_row_affine_f32(const float *A, int64_t lda, const float *B, int64_t ldb, const float *scale, float *out, int64_t ldo,
                        int64_t M, int32_t W, amar_stream_t stream);
int amar_l2norm_fwd_f32(const float *Z, int64_t ldz, float *Nrm, int64_t ldn, float *inv, float *Y, int64_t ldy,
                        int64_t M, int32_t C, int32_t act, amar_stream_t stream);
int amar_l2norm_bwd_f32(const float *dY, int64_t ldd, const float *Nrm, int64_t ldn, const float *inv, float *dZ, int64_t ldz,
                        int64_t M, int32_t C, int32_t act, amar_stream_t stream);
int amar_gat_bwd_f32(const int32_t *rowptr, const int32_t *colidx, const float *H, int64_t ldh, int32_t C,
                     const float *s_self, const float *s_neigh, const float *Y, int64_t ldy, const float *dY, int64_t ldd,
                     const float *bias, const float *a_self, const float *a_neigh,
                     float *dout, float *row_scratch, float *ds, float *dt, float *dH, int64_t lddh,
                     int32_t self_loop, int32_t n_rows, amar_stream_t stream);
int amar_transpose_f32(const float *src, int32_t K, int32_t N, float *dst, amar_stream_t stream);
int amar_adam_f32(float *w, const float *g, float *m, float *v, int64_t n, float lr_t, float beta_1, float beta_2,
                  float epsilon, float l2, amar_stream_t stream);
int amar_adam_advance_f32(float *state, float learning_rate, float beta_1, float beta_2, amar_stream_t stream);
/* All parameters of a model in one launch (a table of slots in device memory; slot k owns blocks [first_block_k,
 * first_block_{k+1}) of 1024 elements each, first_block_0 = 0, total_blocks = sum of ceil(n / 1024)); the update of
 * amar_adam_dev_f32.  If loss_acc != NULL, reg_scale * l2 * sum(w^2) of the pre-update weights is added to *loss_acc (the
 * regularisation part of the loss Keras reports).  amar_sum_into_f32: *acc += scale * sum(x) (the data part). */
typedef struct amar_adam_slot { float *w; const float *g; float *m; float *v; int64_t n; int64_t first_block; float l2; int32_t g_groups; } amar_adam_slot;
/* g_groups == 0: g[n] is the gradient.  g_groups = G > 0: g holds G partial gradients [G][n] (what amar_dense_bwd_f32 leaves in its
 * workspace with AMAR_DENSE_BWD_DEFER) and the gradient is their sum in the order 0 .. G-1 — the reduction launch of every layer folded
 * into the one Adam launch (same order of additions as the explicit reduction: the same bits). */
int amar_adam_multi_f32(const amar_adam_slot *slots, int32_t n_slots, int64_t total_blocks, const float *state, float beta_1,
                        float beta_2, float epsilon, float reg_scale, float *loss_acc, amar_stream_t stream);
int amar_sum_into_f32(const float *x, int64_t n, float scale, float *acc, amar_stream_t stream);
int amar_adam_dev_f32(float *w, const float *g, float *m, float *v, int64_t n, const float *state, float beta_1, float beta_2,
                      float epsilon, float l2, amar_stream_t stream);

/* ---- ranking ------------------------------------------------------------------------------
 * Per-user top-k over that user's own test pairs (src/utilities/metrics.py:11-34):
 * pairs are grouped by user (seg_ptr[n_users+1] into item_ids/scores); for each user the k
 * best (score desc, item id asc on ties) are written to out_items/out_scores [n_users, k],
 * padded with -1 / -inf when the user has fewer than k pairs.  k <= 64.
 */
int amar_topk_segmented_f32(const int32_t *seg_ptr, const int32_t *item_ids, const float *scores,
                            int32_t n_users, int32_t k, int32_t *out_items, float *out_scores,
                            amar_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* AMAR_HIP_H */
