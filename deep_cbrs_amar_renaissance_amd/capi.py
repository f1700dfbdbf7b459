"""ctypes binding of the C-ABI in ``include/amar_hip.h`` (``libamar_hip.so``).

torch is only the carrier here: tensors provide device memory (``data_ptr()``) and the current
HIP stream; every function below checks shapes on the host, hands raw pointers to the library
and raises on a non-zero return code.  There is NO fallback: if the shared library is missing,
or a tensor is not on a GPU, the call fails loudly.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libamar_hip.so')

ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 2
ACT_CODES = {None: ACT_NONE, 'linear': ACT_NONE, 'relu': ACT_RELU, 'sigmoid': ACT_SIGMOID}
SPMM_BIAS, SPMM_RELU, SPMM_ACCUM, SPMM_ACCUM_DIV, SPMM_SCALE_NEXT, SPMM_SAGE_TAIL, SPMM_LT_NOPAIRS = 1, 2, 4, 8, 16, 32, 64

_P = ctypes.c_void_p
_I32, _I64, _U32, _F32 = ctypes.c_int32, ctypes.c_int64, ctypes.c_uint32, ctypes.c_float

# symbol -> (restype, argtypes); kept in the order of include/amar_hip.h
SIGNATURES = {
    'amar_version': (ctypes.c_int, []),
    'amar_error_string': (ctypes.c_char_p, [ctypes.c_int]),
    'amar_last_hip_error': (ctypes.c_int, []),
    'amar_spmm_csr_f32': (ctypes.c_int, [_P, _P, _P, _P, _I64, _P, _I64, _I32, _I32, _U32, _P,
                                         _P, _I64, _P, _I64, _F32, _P]),
    'amar_spmm_sj_f32': (ctypes.c_int, [_P, _P, _P, _I32, _P, _I64, _P, _I64, _I32, _I32, _U32, _P,
                                        _P, _I64, _P, _I64, _F32, _P, _I32, _P, _I64, _P]),
    'amar_spmm_xs_f32': (ctypes.c_int, [_P, _P, _P, _P, _P, _I32, _P, _I64, _I32, _P, _P, _P, _I64, _I32, _I32, _U32, _P,
                                        _P, _I64, _P, _I64, _F32, _P, _I32, _P, _I64, _P]),
    'amar_gat_lt_f32': (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _I64, _I32, _P, _P, _P, _P, _P, _I64,
                                       _I32, _I32, _I32, _I32, _P]),
    'amar_gat_lt_rows_per_wave': (ctypes.c_int, [_I32]),
    'amar_colmax_f32': (ctypes.c_int, [_P, _I64, _P, _P]),
    'amar_spmm_lt_f32': (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _P, _P, _P, _I64, _I32, _P, _P, _I64, _I32, _I32, _U32, _P,
                                        _P, _I64, _P, _I64, _F32, _P, _I32, _P, _I64, _P]),
    'amar_gat_xs_f32': (ctypes.c_int, [_P, _P, _I32, _P, _I64, _I32, _P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _P]),
    'amar_gcn_layer_f32': (ctypes.c_int, [_P, _P, _P, _P, _I64, _I32, _P, _P, _I64, _P, _I32, _P, _I64, _I32, _P]),
    'amar_rowwise_xw_f32': (ctypes.c_int, [_P, _I64, _I32, _P, _I32, _P, _I64, _P, _I64, _P, _P, _P, _P, _P, _I32, _P]),
    'amar_rowwise_xw_gather_f32': (ctypes.c_int, [_P, _I64, _I32, _P, _P, _I32, _P, _I64, _P, _I32, _P]),
    'amar_sage_layer_f32': (ctypes.c_int, [_P, _P, _P, _I64, _I32, _P, _P, _I32, _P, _I64, _I32, _I32, _P]),
    'amar_sage_tail_f32': (ctypes.c_int, [_P, _I64, _P, _I64, _I32, _P, _P, _I32, _P, _I64, _I64, _P]),
    'amar_gat_layer_f32': (ctypes.c_int, [_P, _P, _P, _I64, _I32, _P, _P, _P, _P, _I64, _I32, _I32, _P]),
    'amar_dense_f32': (ctypes.c_int, [_P, _I64, _P, _P, _P, _P, _I64, _I64, _I32, _I32, _I32, _P]),
    'amar_dense_split_bytes': (ctypes.c_int64, [_I32, _I32]),
    'amar_dense_split_pack_f32': (ctypes.c_int, [_P, _I32, _I32, _P]),
    'amar_dense_split_f32': (ctypes.c_int, [_P, _I64, _P, _P, _P, _P, _I64, _I64, _I32, _I32, _I32, _P]),
    'amar_chain_pack_floats': (ctypes.c_int64, [_P, _I32]),
    'amar_chain_pack_f32': (ctypes.c_int, [_P, _P, _P, _I32, _P]),
    'amar_chain_f32': (ctypes.c_int, [_P, _I64, _I32, _P, _I32, _P, _I64, _I32, _P, _I32, _I32, _I32, _P, _P, _P, _I32, _P, _I64, _I64, _P]),
    'amar_chain_indexed_f32': (ctypes.c_int, [_P, _I64, _I32, _P, _I32, _P, _I64, _I32, _P, _I32, _I32, _I32, _P, _P, _P, _I32, _P, _I64, _P, _I64, _P]),
    'amar_chain_segments_f32': (ctypes.c_int, [_P, _P, _P, _I32, _P, _I32, _P, _P, _P, _I32, _P, _I64, _I64, _P]),
    'amar_dual_chain_f32': (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _P, _P, _P, _I32, _P, _P, _I64, _I64, _P]),
    'amar_dual_chain_indexed_f32': (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _P, _P, _P, _I32, _P, _P, _I64, _P, _I64, _P]),
    'amar_copy_columns_f32': (ctypes.c_int, [_P, _I64, _P, _I32, _P, _I64, _I64, _I32, _P]),
    'amar_scatter_f32': (ctypes.c_int, [_P, _P, _P, _I64, _I64, _P, _I32, _P]),
    'amar_reduce_layers_f32': (ctypes.c_int, [_P, _I64, _I32, _I32, _P, _I64, _I64, _I32, _P]),
    'amar_reduce_layers_wsum_f32': (ctypes.c_int, [_P, _I64, _I32, _I32, _P, _P, _I64, _I64, _P]),
    'amar_reduce_layers_wsum_bwd_scratch': (ctypes.c_int64, []),
    'amar_reduce_layers_wsum_bwd_f32': (ctypes.c_int, [_P, _I64, _I32, _I32, _P, _P, _I64, _P, _I64, _P, _P, _I64, _P]),
    'amar_act_bwd_f32': (ctypes.c_int, [_P, _I64, _P, _I64, _P, _I64, _I64, _I32, _I32, _P]),
    'amar_wgrad_scratch_floats': (ctypes.c_int64, [_I64, _I32, _I32]),
    'amar_wgrad_f32': (ctypes.c_int, [_P, _I64, _P, _I64, _I64, _I32, _I32, _P, _P, _P, _P]),
    'amar_dense_stack_f32': (ctypes.c_int, [_P, _I64, _P, _P, _I64, _I32, _P, _P, _P, _P, _P, _P, _I64, _P]),
    'amar_dense_stack_pair_f32': (ctypes.c_int, [_P, _P, _P]),
    'amar_dense_stack_bwd_groups': (ctypes.c_int64, [_I64]),
    'amar_dense_stack_bwd_workspace_floats': (ctypes.c_int64, [_I64, _I32, _P]),
    'amar_dense_stack_bwd_pair_f32': (ctypes.c_int, [_P, _P, _P]),
    'amar_dense_stack_bwd_f32': (ctypes.c_int, [_P, _I64, _P, _I64, _I32, _P, _P, _P, _P, _P, _P, _I64, _P, _P, _P, _I32, _I64, _P]),
    'amar_dense_bwd_workspace_floats': (ctypes.c_int64, [_I64, _I32, _I32]),
    'amar_dense_bwd_groups': (ctypes.c_int64, [_I64]),
    'amar_dense_bwd_f32': (ctypes.c_int, [_P, _I64, _P, _I64, _P, _I64, _P, _I32, _P, _I64, _P, _P, _P, _I64, _P, _I64, _I32, _I32, _P]),
    'amar_bce_grad_f32': (ctypes.c_int, [_P, _I64, _P, _P, _P, _I64, _P]),
    'amar_scatter_add_rows_f32': (ctypes.c_int, [_P, _I64, _P, _I32, _P, _I64, _I64, _I32, _P]),
    'amar_add_inplace_f32': (ctypes.c_int, [_P, _I64, _P, _I64, _I64, _I32, _F32, _P]),
    'amar_row_affine_f32': (ctypes.c_int, [_P, _I64, _P, _I64, _P, _P, _I64, _I64, _I32, _P]),
    'amar_l2norm_fwd_f32': (ctypes.c_int, [_P, _I64, _P, _I64, _P, _P, _I64, _I64, _I32, _I32, _P]),
    'amar_l2norm_bwd_f32': (ctypes.c_int, [_P, _I64, _P, _I64, _P, _P, _I64, _I64, _I32, _I32, _P]),
    'amar_gat_bwd_f32': (ctypes.c_int, [_P, _P, _P, _I64, _I32, _P, _P, _P, _I64, _P, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _I64,
                                        _I32, _I32, _P]),
    'amar_attention_mix_f32': (ctypes.c_int, [_P, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _I64, _I64, _I32, _P]),
    'amar_attention_mix_bwd_f32': (ctypes.c_int, [_P, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _P, _P, _P, _I64, _I32, _P]),
    'amar_add3_act_f32': (ctypes.c_int, [_P, _I64, _P, _I64, _P, _I64, _P, _I64, _I64, _I32, _I32, _P]),
    'amar_locality_scale_f32': (ctypes.c_int, [_P, _I64, _P, _P, _I64, _I64, _I32, _P]),
    'amar_locality_scale_bwd_f32': (ctypes.c_int, [_P, _I64, _P, _I64, _P, _P, _I64, _P, _I64, _I32, _I32, _P]),
    'amar_transpose_f32': (ctypes.c_int, [_P, _I32, _I32, _P, _P]),
    'amar_adam_f32': (ctypes.c_int, [_P, _P, _P, _P, _I64, _F32, _F32, _F32, _F32, _F32, _P]),
    'amar_adam_advance_f32': (ctypes.c_int, [_P, _F32, _F32, _F32, _P]),
    'amar_adam_dev_f32': (ctypes.c_int, [_P, _P, _P, _P, _I64, _P, _F32, _F32, _F32, _F32, _P]),
    'amar_adam_multi_f32': (ctypes.c_int, [_P, _I32, _I64, _P, _F32, _F32, _F32, _F32, _P, _P]),
    'amar_sum_into_f32': (ctypes.c_int, [_P, _I64, _F32, _P, _P]),
    'amar_topk_segmented_f32': (ctypes.c_int, [_P, _P, _P, _I32, _I32, _P, _P, _P]),
}

_lib = None


class AmarError(RuntimeError):
    pass


def load():
    """Load libamar_hip.so (once). Raises if it has not been built: there is no CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AmarError(
                "{} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C deep_cbrs_amar_renaissance_amd/csrc`. The HIP path has no fallback.".format(LIB_PATH))
        lib = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = restype, argtypes
        _lib = lib
    return _lib


def _check(code, what):
    if code != 0:
        lib = load()
        msg = lib.amar_error_string(code).decode()
        raise (ValueError if code == -1 else AmarError)(
            "{} failed: {} (code {}, hipError {})".format(what, msg, code, lib.amar_last_hip_error()))


def _ptr(t, dtype=None, name='tensor'):
    if t is None:
        return None
    if not t.is_cuda:
        raise AmarError("{} must live on the GPU (got {}); the HIP path has no CPU fallback".format(name, t.device))
    if dtype is not None and t.dtype != dtype:
        raise ValueError("{} must be {} (got {})".format(name, dtype, t.dtype))
    return t.data_ptr()


_EMPTY_STANDINS = {}


def _ptr_entries(t, dtype, name):
    """Pointer of a per-non-zero array (colidx / vals).  A matrix without any non-zero has empty arrays whose data pointer
    is NULL; the C entry points reject NULL (a NULL with non-zeros behind it would fault the GPU), so an empty array is
    replaced by a one-element stand-in that the kernels never read (every row range is empty)."""
    if t is None or t.numel() > 0:
        return _ptr(t, dtype, name)
    _ptr(t, dtype, name)                                             # same device / dtype checks
    key = (t.device, dtype)
    if key not in _EMPTY_STANDINS:
        _EMPTY_STANDINS[key] = torch.zeros(4, dtype=dtype, device=t.device)
    return _EMPTY_STANDINS[key].data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ld(t, name):
    """Leading dimension of a 2-D row-major view (last dim contiguous)."""
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise ValueError("{} must be a 2-D tensor whose last dimension is contiguous".format(name))
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


def spmm_csr(rowptr, colidx, vals, X, Y=None, bias=None, relu=False, acc_in=None, acc_out=None, acc_div=None):
    """Y = A.X (+bias, ReLU); optionally acc_out = (acc_in + Y) [/ acc_div]. Tensors are views into device memory."""
    n_rows = rowptr.numel() - 1
    F = X.shape[1]
    if colidx.numel() and int(X.shape[0]) < 1:
        raise ValueError("X has no rows")
    flags = (SPMM_BIAS if bias is not None else 0) | (SPMM_RELU if relu else 0)
    if acc_out is not None:
        flags |= SPMM_ACCUM | (SPMM_ACCUM_DIV if acc_div is not None else 0)
        if acc_in is None or acc_in.shape != (n_rows, F) or acc_out.shape != (n_rows, F):
            raise ValueError("acc_in/acc_out must be [n_rows, F]")
    if Y is not None and tuple(Y.shape) != (n_rows, F):
        raise ValueError("Y must be [n_rows, F]")
    if vals is not None and vals.numel() != colidx.numel():
        raise ValueError("vals and colidx differ in length")
    code = load().amar_spmm_csr_f32(
        _ptr(rowptr, torch.int32, 'rowptr'), _ptr_entries(colidx, torch.int32, 'colidx'), _ptr_entries(vals, torch.float32, 'vals'),
        _ptr(X, torch.float32, 'X'), _ld(X, 'X'), _ptr(Y, torch.float32, 'Y'), _ld(Y, 'Y') if Y is not None else 0,
        n_rows, F, flags, _ptr(bias, torch.float32, 'bias'),
        _ptr(acc_in, torch.float32, 'acc_in'), _ld(acc_in, 'acc_in') if acc_in is not None else 0,
        _ptr(acc_out, torch.float32, 'acc_out'), _ld(acc_out, 'acc_out') if acc_out is not None else 0,
        float(acc_div) if acc_div is not None else 1.0, _stream())
    _check(code, 'amar_spmm_csr_f32')


def spmm_sj(sj, X, Y=None, bias=None, relu=False, acc_in=None, acc_out=None, acc_div=None, Wnext=None, Hnext=None):
    """Y = A.X on the sliced-jagged image `sj` of A (utilities.math.SlicedJagged), with the epilogues of
    spmm_csr / gcn_layer.  Used when the node table does not fit the per-XCD L2."""
    n_rows = sj.shape[0]
    F = X.shape[1]
    flags = (SPMM_BIAS if bias is not None else 0) | (SPMM_RELU if relu else 0)
    if acc_out is not None:
        flags |= SPMM_ACCUM | (SPMM_ACCUM_DIV if acc_div is not None else 0)
        if acc_in is None or tuple(acc_in.shape) != (n_rows, F) or tuple(acc_out.shape) != (n_rows, F):
            raise ValueError("acc_in/acc_out must be [n_rows, F]")
    if Y is not None and tuple(Y.shape) != (n_rows, F):
        raise ValueError("Y must be [n_rows, F]")
    if X.shape[0] < sj.shape[1]:
        raise ValueError("X has fewer rows than the matrix has columns")
    Cn = 0
    if Wnext is not None:
        if Wnext.shape[0] != F or not Wnext.is_contiguous() or Hnext is None or tuple(Hnext.shape) != (n_rows, Wnext.shape[1]):
            raise ValueError("spmm_sj: Wnext [F, Cn] contiguous and Hnext [n_rows, Cn] expected")
        Cn = Wnext.shape[1]
    code = load().amar_spmm_sj_f32(
        _ptr(sj.entries, torch.int32, 'entries'), _ptr(sj.counts, torch.int16, 'counts'),
        _ptr(sj.wave_start, torch.int32, 'wave_start'), sj.n_slices,
        _ptr(X, torch.float32, 'X'), _ld(X, 'X'), _ptr(Y, torch.float32, 'Y'), _ld(Y, 'Y') if Y is not None else 0,
        n_rows, F, flags, _ptr(bias, torch.float32, 'bias'),
        _ptr(acc_in, torch.float32, 'acc_in'), _ld(acc_in, 'acc_in') if acc_in is not None else 0,
        _ptr(acc_out, torch.float32, 'acc_out'), _ld(acc_out, 'acc_out') if acc_out is not None else 0,
        float(acc_div) if acc_div is not None else 1.0,
        _ptr(Wnext, torch.float32, 'Wnext'), Cn, _ptr(Hnext, torch.float32, 'Hnext'),
        _ld(Hnext, 'Hnext') if Hnext is not None else 0, _stream())
    _check(code, 'amar_spmm_sj_f32')


def spmm_xs(xs, X, Y=None, bias=None, relu=False, acc_in=None, acc_out=None, acc_div=None, Wnext=None, Hnext=None,
            prescaled=False, scale_next=False, xself=None, rows_needed=None):
    """Y = A.X on the XCD-sliced image `xs` of A (utilities.math.XcdSliced): per-slice partial products with
    XCD <-> L2 affinity, then the combine kernel with the epilogues of spmm_csr / gcn_layer.

    A value-free image (xs.row_scale is set: A = S C S) gathers from S.X: `prescaled=True` says X already is that
    table (the fused GCN chain keeps it so with `scale_next`), otherwise one row_affine pass makes it here.

    `xs` may also be an LDS-tiled image (utilities.lds_tiled.LdsTiled, what DeviceCSR.tiled_image returns where the
    column-ordered form pays): the call is then amar_spmm_lt_f32, same keywords.

    `xself` ([n_rows, F], same leading dimension as X): where the rows' OWN entries of X are read from, instead of
    X[diag_offset:] — a row block of a partition whose own rows of the gathered table may not have landed yet (parallel.py)."""
    if hasattr(xs, 'words'):
        return spmm_lt(xs, X, Y, bias=bias, relu=relu, acc_in=acc_in, acc_out=acc_out, acc_div=acc_div, Wnext=Wnext, Hnext=Hnext,
                       prescaled=prescaled, scale_next=scale_next, xself=xself, rows_needed=rows_needed)
    n_rows = xs.shape[0]
    F = X.shape[1]
    flags = (SPMM_BIAS if bias is not None else 0) | (SPMM_RELU if relu else 0)
    if acc_out is not None:
        flags |= SPMM_ACCUM | (SPMM_ACCUM_DIV if acc_div is not None else 0)
        if acc_in is None or tuple(acc_in.shape) != (n_rows, F) or tuple(acc_out.shape) != (n_rows, F):
            raise ValueError("acc_in/acc_out must be [n_rows, F]")
    if Y is not None and tuple(Y.shape) != (n_rows, F):
        raise ValueError("Y must be [n_rows, F]")
    diag_offset = int(getattr(xs, 'diag_offset', 0))
    if X.shape[0] != xs.shape[1] or diag_offset + n_rows > X.shape[0]:
        raise ValueError("X must have one row per column of the matrix")
    Cn = 0
    if Wnext is not None:
        if Wnext.shape[0] != F or not Wnext.is_contiguous() or Hnext is None or tuple(Hnext.shape) != (n_rows, Wnext.shape[1]):
            raise ValueError("spmm_xs: Wnext [F, Cn] contiguous and Hnext [n_rows, Cn] expected")
        Cn = Wnext.shape[1]
    P = xs.partials(F)
    row_scale = getattr(xs, 'row_scale', None)
    if row_scale is None and (prescaled or scale_next):
        raise ValueError("spmm_xs: prescaled / scale_next need a value-free image")
    if row_scale is not None and not prescaled:
        col_scale = getattr(xs, 'col_scale', None)
        Xs = torch.empty((X.shape[0], F), dtype=torch.float32, device=X.device)
        row_affine(X, col_scale if col_scale is not None else row_scale, Xs)
        X = Xs
    if scale_next:
        flags |= SPMM_SCALE_NEXT
    code = load().amar_spmm_xs_f32(
        _ptr(xs.diag, torch.float32, 'diag'), _ptr(xs.rowptr, torch.int32, 'rowptr'), _ptr(xs.colidx, torch.int32, 'colidx'),
        _ptr(xs.vals, torch.float32, 'vals'), _ptr(row_scale, torch.float32, 'row_scale'), xs.n_slices,
        _ptr(X, torch.float32, 'X'), _ld(X, 'X'), X.shape[0], _xself_ptr(xself, X, diag_offset, n_rows, prescaled or row_scale is None),
        _ptr(P, torch.float32, 'partials'),
        _ptr(Y, torch.float32, 'Y'), _ld(Y, 'Y') if Y is not None else 0,
        n_rows, F, flags, _ptr(bias, torch.float32, 'bias'),
        _ptr(acc_in, torch.float32, 'acc_in'), _ld(acc_in, 'acc_in') if acc_in is not None else 0,
        _ptr(acc_out, torch.float32, 'acc_out'), _ld(acc_out, 'acc_out') if acc_out is not None else 0,
        float(acc_div) if acc_div is not None else 1.0,
        _ptr(Wnext, torch.float32, 'Wnext'), Cn, _ptr(Hnext, torch.float32, 'Hnext'),
        _ld(Hnext, 'Hnext') if Hnext is not None else 0, _stream())
    _check(code, 'amar_spmm_xs_f32')


def _xself_ptr(xself, X, diag_offset, n_rows, usable):
    """The pointer the SpMM kernels read the rows' own entries of X through: `xself` when given, else X[diag_offset:] (None: X)."""
    if xself is None:
        return _ptr(X[diag_offset:], torch.float32, 'X') if diag_offset else None
    if not usable:
        raise ValueError("xself needs the table as the kernel gathers it (prescaled=True on a value-free image)")
    if tuple(xself.shape) != (n_rows, X.shape[1]) or _ld(xself, 'xself') != _ld(X, 'X'):
        raise ValueError("xself must be [n_rows, F] with the leading dimension of X")
    return _ptr(xself, torch.float32, 'xself')


def spmm_lt(lt, X, Y=None, bias=None, relu=False, acc_in=None, acc_out=None, acc_div=None, Wnext=None, Hnext=None,
            prescaled=False, scale_next=False, sage_tail=None, xself=None, rows_needed=None):
    """Y = A.X on the LDS-tiled image `lt` of a value-free A = S C S (utilities.lds_tiled.LdsTiled): one launch, the
    row tile's sums in LDS, gathers in column order.  Keywords as spmm_xs (`prescaled`: X already holds S.X).
    sage_tail = (kernel [2F, F], bias [F]) on GraphSAGE's mean-aggregate image: Y = relu(l2_normalize([X || mean] . kernel + bias))
    in the same launch (AMAR_SPMM_SAGE_TAIL)."""
    n_rows, n_cols = lt.shape
    F = X.shape[1]
    if sage_tail is not None:
        if bias is not None or relu or acc_out is not None or Wnext is not None or not prescaled or Y is None:
            raise ValueError("spmm_lt: sage_tail takes the un-scaled table (prescaled=True), Y, and no other epilogue")
        if Hnext is not None and tuple(Hnext.shape) != (n_rows, F):
            raise ValueError("spmm_lt: with sage_tail, Hnext is a second [n_rows, F] copy of Y")
        Wnext, bias = sage_tail
        if tuple(Wnext.shape) != (2 * F, F) or not Wnext.is_contiguous() or bias.numel() != F:
            raise ValueError("spmm_lt: sage_tail = (kernel [2F, F] contiguous, bias [F]) expected")
    if F != lt.F:
        raise ValueError("spmm_lt: the image was built for width {}, X is {} wide".format(lt.F, F))
    from deep_cbrs_amar_renaissance_amd.utilities import lds_tiled
    if lt.rw > lds_tiled.geometry(F)[1]:
        raise ValueError("spmm_lt: the image's tiles hold {} rows per wave, the kernel's {}".format(lt.rw, lds_tiled.geometry(F)[1]))
    flags = (SPMM_SAGE_TAIL if sage_tail is not None else (SPMM_BIAS if bias is not None else 0) | (SPMM_RELU if relu else 0))
    if acc_out is not None:
        flags |= SPMM_ACCUM | (SPMM_ACCUM_DIV if acc_div is not None else 0)
        if acc_in is None or tuple(acc_in.shape) != (n_rows, F) or tuple(acc_out.shape) != (n_rows, F):
            raise ValueError("acc_in/acc_out must be [n_rows, F]")
    if Y is not None and tuple(Y.shape) != (n_rows, F):
        raise ValueError("Y must be [n_rows, F]")
    if X.shape[0] != n_cols or lt.diag_offset + n_rows > X.shape[0]:
        raise ValueError("X must have one row per column of the matrix")
    Cn = 0
    if sage_tail is not None:
        Cn = F
    elif Wnext is not None:
        if Wnext.shape[0] != F or not Wnext.is_contiguous() or Hnext is None or tuple(Hnext.shape) != (n_rows, Wnext.shape[1]):
            raise ValueError("spmm_lt: Wnext [F, Cn] contiguous and Hnext [n_rows, Cn] expected")
        Cn = Wnext.shape[1]
    if not prescaled:
        Xs = torch.empty((X.shape[0], F), dtype=torch.float32, device=X.device)
        row_affine(X, lt.col_scale if lt.col_scale is not None else lt.row_scale, Xs)
        X = Xs
    if scale_next:
        flags |= SPMM_SCALE_NEXT
    if not getattr(lt, 'pairs', True):
        flags |= SPMM_LT_NOPAIRS
    off = lt.diag_offset
    n_tiles = lt.n_tiles
    if rows_needed is not None and rows_needed < n_rows:
        # only the tiles that hold rows [0, rows_needed): the rows of the others are left as they are (the last layer of a
        # user-item-property graph: no tower reads its property rows — models/gnn.py `rows_needed`)
        cache = lt.__dict__.setdefault('_tiles_for_rows', {})
        if rows_needed not in cache:
            cache[rows_needed] = int((lt.tile_row0[:-1] < int(rows_needed)).sum())
        n_tiles = cache[rows_needed]
        if n_tiles == 0:
            return
    code = load().amar_spmm_lt_f32(
        _ptr(lt.words, torch.int32, 'words'), _ptr(lt.stream_start, torch.int32, 'stream_start'),
        _ptr(lt.wsteps, torch.int32, 'wsteps'), _ptr(lt.tile_row0, torch.int32, 'tile_row0'), _ptr(lt.n_win, torch.int32, 'n_win'),
        _ptr(lt.vstart, torch.int32, 'vstart'), _ptr(lt.vcount, torch.int32, 'vcount'), n_tiles, lt.maxwin1, lt.pace_every, _ptr(lt.diag, torch.float32, 'diag'), _ptr(lt.row_scale, torch.float32, 'row_scale'),
        _ptr(X, torch.float32, 'X'), _ld(X, 'X'), X.shape[0], _xself_ptr(xself, X, off, n_rows, prescaled),
        _ptr(Y, torch.float32, 'Y'), _ld(Y, 'Y') if Y is not None else 0,
        n_rows, F, flags, _ptr(bias, torch.float32, 'bias'),
        _ptr(acc_in, torch.float32, 'acc_in'), _ld(acc_in, 'acc_in') if acc_in is not None else 0,
        _ptr(acc_out, torch.float32, 'acc_out'), _ld(acc_out, 'acc_out') if acc_out is not None else 0,
        float(acc_div) if acc_div is not None else 1.0,
        _ptr(Wnext, torch.float32, 'Wnext'), Cn, _ptr(Hnext, torch.float32, 'Hnext'),
        _ld(Hnext, 'Hnext') if Hnext is not None else 0, _stream())
    _check(code, 'amar_spmm_lt_f32')


def gcn_layer(rowptr, colidx, vals, H, bias, Y, Wnext=None, Hnext=None):
    n_rows = rowptr.numel() - 1
    C = H.shape[1]
    if tuple(Y.shape) != (n_rows, C) or bias.numel() != C or H.shape[0] < n_rows:
        raise ValueError("gcn_layer: H [>=n_rows, C], bias [C], Y [n_rows, C] expected")
    Cn = 0
    if Wnext is not None:
        if Wnext.shape[0] != C or not Wnext.is_contiguous() or Hnext is None or tuple(Hnext.shape) != (n_rows, Wnext.shape[1]):
            raise ValueError("gcn_layer: Wnext [C, Cn] contiguous and Hnext [n_rows, Cn] expected")
        Cn = Wnext.shape[1]
    code = load().amar_gcn_layer_f32(
        _ptr(rowptr, torch.int32, 'rowptr'), _ptr_entries(colidx, torch.int32, 'colidx'), _ptr_entries(vals, torch.float32, 'vals'),
        _ptr(H, torch.float32, 'H'), _ld(H, 'H'), C, _ptr(bias, torch.float32, 'bias'),
        _ptr(Y, torch.float32, 'Y'), _ld(Y, 'Y'),
        _ptr(Wnext, torch.float32, 'Wnext'), Cn, _ptr(Hnext, torch.float32, 'Hnext'),
        _ld(Hnext, 'Hnext') if Hnext is not None else 0, n_rows, _stream())
    _check(code, 'amar_gcn_layer_f32')


def rowwise_xw(X, W, H, copy_to=None, a_self=None, a_neigh=None, s_self=None, s_neigh=None, row_scale=None, row_ids=None):
    """H = X . W row by row (+ the X_0 slice copy, GAT's attention scalars, a row scale).  row_ids (int32 [rows of H]): output row p
    reads X[row_ids[p]] and a negative id leaves a zero row (amar_rowwise_xw_gather_f32: the block layout of a node-range partition)."""
    if row_ids is not None:
        n_out, F = int(row_ids.numel()), X.shape[1]
        if copy_to is not None or a_self is not None or s_self is not None:
            raise ValueError("rowwise_xw: the row-gather form computes H only")
        if W.shape[0] != F or not W.is_contiguous() or tuple(H.shape) != (n_out, W.shape[1]) or not row_ids.is_contiguous():
            raise ValueError("rowwise_xw: W [F, C] contiguous, row_ids contiguous and H [len(row_ids), C] expected")
        if row_scale is not None and (row_scale.numel() != n_out or not row_scale.is_contiguous()):
            raise ValueError("rowwise_xw: row_scale must be a contiguous [len(row_ids)] vector")
        code = load().amar_rowwise_xw_gather_f32(
            _ptr(X, torch.float32, 'X'), _ld(X, 'X'), F, _ptr(row_ids, torch.int32, 'row_ids'), _ptr(W, torch.float32, 'W'), W.shape[1],
            _ptr(H, torch.float32, 'H'), _ld(H, 'H'), _ptr(row_scale, torch.float32, 'row_scale'), n_out, _stream())
        _check(code, 'amar_rowwise_xw_gather_f32')
        return
    n_rows, F = X.shape
    if W.shape[0] != F or not W.is_contiguous() or tuple(H.shape) != (n_rows, W.shape[1]):
        raise ValueError("rowwise_xw: W [F, C] contiguous and H [n_rows, C] expected")
    if copy_to is not None and tuple(copy_to.shape) != (n_rows, F):
        raise ValueError("rowwise_xw: copy_to must be [n_rows, F]")
    if row_scale is not None and (row_scale.numel() != n_rows or not row_scale.is_contiguous()):
        raise ValueError("rowwise_xw: row_scale must be a contiguous [n_rows] vector")
    for v, nm in ((s_self, 's_self'), (s_neigh, 's_neigh')):
        if v is not None and (v.numel() != n_rows or not v.is_contiguous()):
            raise ValueError("rowwise_xw: {} must be a contiguous [n_rows] vector".format(nm))
    code = load().amar_rowwise_xw_f32(
        _ptr(X, torch.float32, 'X'), _ld(X, 'X'), F, _ptr(W, torch.float32, 'W'), W.shape[1],
        _ptr(H, torch.float32, 'H'), _ld(H, 'H'),
        _ptr(copy_to, torch.float32, 'copy_to'), _ld(copy_to, 'copy_to') if copy_to is not None else 0,
        _ptr(a_self, torch.float32, 'a_self'), _ptr(a_neigh, torch.float32, 'a_neigh'),
        _ptr(s_self, torch.float32, 's_self'), _ptr(s_neigh, torch.float32, 's_neigh'),
        _ptr(row_scale, torch.float32, 'row_scale'), n_rows, _stream())
    _check(code, 'amar_rowwise_xw_f32')


def sage_layer(rowptr, colidx, X, W, bias, Y, self_loop=True):
    n_rows = rowptr.numel() - 1
    F = X.shape[1]
    if W.shape[0] != 2 * F or not W.is_contiguous() or bias.numel() != W.shape[1] or \
            tuple(Y.shape) != (n_rows, W.shape[1]) or X.shape[0] < n_rows:
        raise ValueError("sage_layer: W [2F, C] contiguous, bias [C], Y [n_rows, C] expected")
    code = load().amar_sage_layer_f32(
        _ptr(rowptr, torch.int32, 'rowptr'), _ptr_entries(colidx, torch.int32, 'colidx'),
        _ptr(X, torch.float32, 'X'), _ld(X, 'X'), F, _ptr(W, torch.float32, 'W'), _ptr(bias, torch.float32, 'bias'),
        W.shape[1], _ptr(Y, torch.float32, 'Y'), _ld(Y, 'Y'), 1 if self_loop else 0, n_rows, _stream())
    _check(code, 'amar_sage_layer_f32')


def sage_tail(X, agg, W, bias, Y):
    """Y = relu(l2_normalize([X || agg] . W + bias)): GraphSageConv after its mean aggregate (see amar_sage_tail_f32)."""
    n_rows, F = agg.shape
    if X.shape[1] != F or X.shape[0] < n_rows or W.shape[0] != 2 * F or not W.is_contiguous() or bias.numel() != W.shape[1] or \
            tuple(Y.shape) != (n_rows, W.shape[1]):
        raise ValueError("sage_tail: X [>=n_rows, F], agg [n_rows, F], W [2F, C] contiguous, bias [C], Y [n_rows, C] expected")
    code = load().amar_sage_tail_f32(_ptr(X, torch.float32, 'X'), _ld(X, 'X'), _ptr(agg, torch.float32, 'agg'), _ld(agg, 'agg'), F,
                                     _ptr(W, torch.float32, 'W'), _ptr(bias, torch.float32, 'bias'), W.shape[1],
                                     _ptr(Y, torch.float32, 'Y'), _ld(Y, 'Y'), n_rows, _stream())
    _check(code, 'amar_sage_tail_f32')


def sage_tail_supported(F, C):
    return F % 4 == 0 and C % 4 == 0 and 4 <= F <= 64 and 4 <= C <= 64


def gat_layer(rowptr, colidx, H, s_self, s_neigh, bias, Y, self_loop=True):
    n_rows = rowptr.numel() - 1
    C = H.shape[1]
    if tuple(Y.shape) != (n_rows, C) or bias.numel() != C or s_self.numel() < n_rows or s_neigh.numel() < H.shape[0]:
        raise ValueError("gat_layer: bias [C], Y [n_rows, C], s_self/s_neigh [n] expected")
    code = load().amar_gat_layer_f32(
        _ptr(rowptr, torch.int32, 'rowptr'), _ptr_entries(colidx, torch.int32, 'colidx'),
        _ptr(H, torch.float32, 'H'), _ld(H, 'H'), C, _ptr(s_self, torch.float32, 's_self'),
        _ptr(s_neigh, torch.float32, 's_neigh'), _ptr(bias, torch.float32, 'bias'),
        _ptr(Y, torch.float32, 'Y'), _ld(Y, 'Y'), 1 if self_loop else 0, n_rows, _stream())
    _check(code, 'amar_gat_layer_f32')


def gat_xs(xs, H, s_self, s_neigh, bias, Y, self_loop=True):
    """gat_layer on the XCD-sliced image `xs` of the edge-list adjacency (utilities.math.XcdSliced; values unused).
    `xs` may be a row block (multi-GPU partition): H / s_self / s_neigh then cover all of its columns."""
    n, n_cols = xs.shape
    row_offset = int(getattr(xs, 'diag_offset', 0))
    C = H.shape[1]
    if tuple(Y.shape) != (n, C) or H.shape[0] != n_cols or bias.numel() != C or s_self.numel() != n_cols or s_neigh.numel() != n_cols:
        raise ValueError("gat_xs: H [n_cols, C], Y [n_rows, C], bias [C], s_self / s_neigh [n_cols] expected")
    scratch = xs.__dict__.setdefault('_gat_scratch', {})
    if C not in scratch:
        scratch[C] = (torch.empty((n_cols, 2 * C), dtype=torch.float32, device=H.device),
                      torch.empty((xs.n_slices, n, 2 * C), dtype=torch.float32, device=H.device))
    packed, partials = scratch[C]
    code = load().amar_gat_xs_f32(
        _ptr(xs.rowptr, torch.int32, 'rowptr'), _ptr(xs.colidx, torch.int32, 'colidx'), xs.n_slices,
        _ptr(H, torch.float32, 'H'), _ld(H, 'H'), C, _ptr(s_self, torch.float32, 's_self'), _ptr(s_neigh, torch.float32, 's_neigh'),
        _ptr(bias, torch.float32, 'bias'), _ptr(packed), _ptr(partials), _ptr(Y, torch.float32, 'Y'), _ld(Y, 'Y'),
        1 if self_loop else 0, n, n_cols, row_offset, _stream())
    _check(code, 'amar_gat_xs_f32')


def colmax(x, out):
    """out[0] = max(x) (a 1-element float32 device tensor), in-stream."""
    if not x.is_contiguous() or out.numel() < 1:
        raise ValueError("colmax: contiguous x and a 1-element out expected")
    _check(load().amar_colmax_f32(_ptr(x, torch.float32, 'x'), x.numel(), _ptr(out, torch.float32, 'out'), _stream()), 'amar_colmax_f32')


def gat_lt(lt, csr, H, s_self, s_neigh, bias, Y, self_loop=True):
    """gat_layer on the LDS-tiled image `lt` (utilities.lds_tiled.LdsTiled built with the GAT geometry, DeviceCSR.tiled_gat_image)
    of the edge-list adjacency `csr` (a DeviceCSR, or a row block carrying diag_offset): one launch + the bound's reduction."""
    n, n_cols = lt.shape
    row_offset = int(getattr(lt, 'diag_offset', 0))
    C = H.shape[1]
    if C != lt.F or tuple(Y.shape) != (n, C) or H.shape[0] != n_cols or bias.numel() != C or s_self.numel() != n_cols or s_neigh.numel() != n_cols:
        raise ValueError("gat_lt: H [n_cols, C], Y [n_rows, C], bias [C], s_self / s_neigh [n_cols] and an image built for C expected")
    if tuple(csr.shape) != (n, n_cols):
        raise ValueError("gat_lt: the CSR and the image must describe the same block")
    bmax = lt.__dict__.setdefault('_gat_bound', torch.empty(1, dtype=torch.float32, device=H.device))
    colmax(s_neigh, bmax)
    code = load().amar_gat_lt_f32(
        _ptr(lt.words, torch.int32, 'words'), _ptr(lt.stream_start, torch.int32, 'stream_start'),
        _ptr(lt.wsteps, torch.int32, 'wsteps'), _ptr(lt.tile_row0, torch.int32, 'tile_row0'), _ptr(lt.n_win, torch.int32, 'n_win'),
        _ptr(lt.vstart, torch.int32, 'vstart'), _ptr(lt.vcount, torch.int32, 'vcount'), lt.n_tiles, lt.maxwin1, lt.pace_every,
        int(lt.rw), _ptr(lt.diag, torch.float32, 'diag'), _ptr(csr.rowptr, torch.int32, 'rowptr'), _ptr_entries(csr.colidx, torch.int32, 'colidx'),
        _ptr(H, torch.float32, 'H'), _ld(H, 'H'), C, _ptr(s_self, torch.float32, 's_self'), _ptr(s_neigh, torch.float32, 's_neigh'),
        _ptr(bmax, torch.float32, 'bmax'), _ptr(bias, torch.float32, 'bias'), _ptr(Y, torch.float32, 'Y'), _ld(Y, 'Y'),
        1 if self_loop else 0, n, n_cols, row_offset, _stream())
    _check(code, 'amar_gat_lt_f32')


DENSE_WT = 0x100


def dense(X, W, bias, Y, act='relu', ids=None, w_transposed=False):
    """Y = act(X[ids] . W + bias). Y may be a column slice of a wider buffer (concatenation).
    w_transposed: W holds the transpose ([N, K]), i.e. Y = X . W^T — the reverse pass' dX = dZ . W^T without a transpose launch."""
    K, N = (W.shape[1], W.shape[0]) if w_transposed else W.shape
    M = ids.numel() if ids is not None else X.shape[0]
    if X.shape[1] != K or not W.is_contiguous() or tuple(Y.shape) != (M, N):
        raise ValueError("dense: X [*, K], W [K, N] contiguous, Y [M, N] expected")
    if bias is not None and (bias.numel() != N or not bias.is_contiguous()):
        raise ValueError("dense: bias must be a contiguous [N] vector")
    code = load().amar_dense_f32(
        _ptr(X, torch.float32, 'X'), _ld(X, 'X'), _ptr(ids, torch.int32, 'ids'),
        _ptr(W, torch.float32, 'W'), _ptr(bias, torch.float32, 'bias'), _ptr(Y, torch.float32, 'Y'), _ld(Y, 'Y'),
        M, K, N, ACT_CODES[act] | (DENSE_WT if w_transposed else 0), _stream())
    _check(code, 'amar_dense_f32')


def dense_split_supported(K, N):
    """amar_dense_split_f32 takes this layer (the wide layers of the content towers); AMAR_DENSE_SPLIT=0 keeps the f32 instruction."""
    return K % 32 == 0 and N % 128 == 0 and K >= 32 and os.environ.get('AMAR_DENSE_SPLIT', '1') != '0'


def dense_split_pack(W):
    """Kernel [K, N] (host numpy float32) -> the pre-split image of amar_dense_split_pack_f32 as a uint8 numpy array."""
    import numpy as np
    W = np.ascontiguousarray(W, dtype=np.float32)
    K, N = W.shape
    lib = load()
    nbytes = int(lib.amar_dense_split_bytes(K, N))
    if nbytes < 0:
        _check(nbytes, 'amar_dense_split_bytes')
    out = np.empty(nbytes, dtype=np.uint8)
    _check(lib.amar_dense_split_pack_f32(W.ctypes.data, K, N, out.ctypes.data), 'amar_dense_split_pack_f32')
    return out


def dense_split(X, Wq, K, N, bias, Y, act='relu', ids=None):
    """Y = act(X[ids] . W + bias) with W pre-split (dense_split_pack, on the device as uint8): amar_dense_split_f32."""
    M = ids.numel() if ids is not None else X.shape[0]
    if X.shape[1] != K or tuple(Y.shape) != (M, N) or Wq.numel() != K * N * 6 or Wq.dtype != torch.uint8 or not Wq.is_contiguous():
        raise ValueError("dense_split: X [*, K], Wq of K N 6 bytes, Y [M, N] expected")
    if bias is not None and (bias.numel() != N or not bias.is_contiguous()):
        raise ValueError("dense_split: bias must be a contiguous [N] vector")
    code = load().amar_dense_split_f32(
        _ptr(X, torch.float32, 'X'), _ld(X, 'X'), _ptr(ids, torch.int32, 'ids'), _ptr(Wq, torch.uint8, 'Wq'),
        _ptr(bias, torch.float32, 'bias'), _ptr(Y, torch.float32, 'Y'), _ld(Y, 'Y'), M, K, N, ACT_CODES[act], _stream())
    _check(code, 'amar_dense_split_f32')


CHAIN_MAX_WIDTH, CHAIN_MAX_LAYERS = 128, 8


def chain_supported(dims, in_a, in_b=0, sum_inputs=False):
    """True when amar_chain_f32 can run a dense stack with these widths (else use dense() per layer)."""
    width_in = in_a if sum_inputs else in_a + in_b
    return (1 <= len(dims) - 1 <= CHAIN_MAX_LAYERS and max(dims) <= CHAIN_MAX_WIDTH and in_a % 4 == 0 and in_a >= 4
            and in_b % 4 == 0 and dims[0] == width_in and (not sum_inputs or in_a == in_b)
            and (dims[-1] % 4 == 0 or (dims[-1] == 1 and len(dims) > 2)))


def chain_pack(kernels, biases):
    """Host-side packing of a Dense stack into MFMA fragment order; returns a device float32 blob + dims."""
    import numpy as np
    ks = [np.ascontiguousarray(k, dtype=np.float32) for k in kernels]
    bs = [np.ascontiguousarray(b, dtype=np.float32) for b in biases]
    dims = [ks[0].shape[0]] + [k.shape[1] for k in ks]
    for k, b, kin, n in zip(ks, bs, dims[:-1], dims[1:]):
        if k.shape != (kin, n) or b.shape != (n,):
            raise ValueError("chain_pack: inconsistent layer shapes")
    lib = load()
    dims_c = (ctypes.c_int32 * len(dims))(*dims)
    total = lib.amar_chain_pack_floats(dims_c, len(ks))
    if total < 0:
        _check(int(total), 'amar_chain_pack_floats')
    out = np.empty(total, dtype=np.float32)
    kp = (ctypes.c_void_p * len(ks))(*[k.ctypes.data for k in ks])
    bp = (ctypes.c_void_p * len(bs))(*[b.ctypes.data for b in bs])
    _check(lib.amar_chain_pack_f32(kp, bp, dims_c, len(ks), out.ctypes.data), 'amar_chain_pack_f32')
    return out, dims


class ConcatTable:
    """The column-wise concatenation of several [rows, w_j] device tables with the same row numbering, NOT materialised: what
    ReductionLayer('concatenation') hands to the towers when every layer's output stays where its producer (or the all-gather of a
    node-range partition) left it.  `chain` reads it in place (amar_chain_segments_f32) where the stack has a compile-time tower
    shape and assembles it once otherwise (`materialize`)."""

    def __init__(self, tables):
        tables = list(tables)
        if not tables:
            raise ValueError("ConcatTable: at least one table expected")
        rows = int(tables[0].shape[0])
        if any(t.dim() != 2 or int(t.shape[0]) != rows or t.dtype != torch.float32 for t in tables):
            raise ValueError("ConcatTable: float32 [rows, w] tables with equal row counts expected")
        # the segment-reading kernel moves float4s: tables whose widths are not multiples of 4 are assembled once instead (`chain`)
        self.in_place = all(int(t.shape[1]) % 4 == 0 for t in tables)
        self.segments = tables
        self.shape = (rows, sum(int(t.shape[1]) for t in tables))
        self.device, self.dtype = tables[0].device, torch.float32

    def __getitem__(self, key):
        if not isinstance(key, slice) or key.step not in (None, 1):
            raise TypeError("ConcatTable supports contiguous row slices only")
        return ConcatTable([t[key] for t in self.segments])

    def materialize(self):
        out = torch.empty(self.shape, dtype=torch.float32, device=self.device)
        off = 0
        for t in self.segments:
            copy_columns(t, out[:, off:off + t.shape[1]])
            off += t.shape[1]
        return out


def chain_segments(table, wpack, dims, acts, out, ids=None, base=0):
    """`chain` on a ConcatTable read in place; returns False when the stack's shape has no segment-reading kernel."""
    P = out.shape[0]
    segs = table.segments
    if ids is not None and ids.numel() != P:
        raise ValueError("chain: ids must have one id per output row")
    if ids is None and table.shape[0] < P:
        raise ValueError("chain: input blocks have fewer rows than the output")
    n = len(segs)
    ptrs = (ctypes.c_void_p * n)(*[_ptr(t, torch.float32, 'segment') for t in segs])
    lds = (ctypes.c_int64 * n)(*[_ld(t, 'segment') for t in segs])
    ws = (ctypes.c_int32 * n)(*[int(t.shape[1]) for t in segs])
    dims_c = (ctypes.c_int32 * len(dims))(*dims)
    acts_c = (ctypes.c_int32 * len(acts))(*[ACT_CODES[a] for a in acts])
    code = load().amar_chain_segments_f32(ptrs, lds, ws, n, _ptr(ids, torch.int32, 'ids'), int(base),
                                          _ptr(wpack, torch.float32, 'wpack'), dims_c, acts_c, len(acts),
                                          _ptr(out, torch.float32, 'out'), _ld(out, 'out'), P, _stream())
    if code == -2:                                                   # AMAR_EUNSUPPORTED: no compile-time tower shape for this stack
        return False
    _check(code, 'amar_chain_segments_f32')
    return True


def chain(A, wpack, dims, acts, out, ids_a=None, base_a=0, B=None, ids_b=None, base_b=0, sum_inputs=False, in_act=None, out_index=None):
    """out = DenseStack([A[ids_a - base_a] || B[ids_b - base_b]]), or DenseStack(in_act(A[..] + B[..])) with
    sum_inputs; see amar_chain_f32 in include/amar_hip.h.  out_index (int32 [P]): row p goes to out[out_index[p]]
    (amar_chain_indexed_f32: a pair list kept in XCD-affine order, scores back in the caller's order).
    A may be a ConcatTable (per-layer tables read in place: amar_chain_segments_f32)."""
    if isinstance(A, ConcatTable):
        if A.in_place and B is None and not sum_inputs and out_index is None and chain_segments(A, wpack, dims, acts, out, ids=ids_a, base=base_a):
            return
        A = A.materialize()
    P = out.shape[0]
    Da, Db = A.shape[1], (B.shape[1] if B is not None else 0)
    for ids, nm in ((ids_a, 'ids_a'), (ids_b, 'ids_b')):
        if ids is not None and ids.numel() != P:
            raise ValueError("chain: {} must have one id per output row".format(nm))
    if ids_a is None and A.shape[0] < P or (B is not None and ids_b is None and B.shape[0] < P):
        raise ValueError("chain: input blocks have fewer rows than the output")
    dims_c = (ctypes.c_int32 * len(dims))(*dims)
    acts_c = (ctypes.c_int32 * len(acts))(*[ACT_CODES[a] for a in acts])
    if out_index is not None and out_index.numel() != P:
        raise ValueError("chain: out_index must have one entry per output row")
    code = load().amar_chain_indexed_f32(
        _ptr(A, torch.float32, 'A'), _ld(A, 'A'), Da, _ptr(ids_a, torch.int32, 'ids_a'), int(base_a),
        _ptr(B, torch.float32, 'B'), _ld(B, 'B') if B is not None else 0, Db, _ptr(ids_b, torch.int32, 'ids_b'), int(base_b),
        1 if sum_inputs else 0, ACT_CODES[in_act],
        _ptr(wpack, torch.float32, 'wpack'), dims_c, acts_c, len(acts),
        _ptr(out, torch.float32, 'out'), _ld(out, 'out'), _ptr(out_index, torch.int32, 'out_index'), P, _stream())
    _check(code, 'amar_chain_indexed_f32' if out_index is not None else 'amar_chain_f32')


def dual_chain_supported(D, n_branch_dims_equal, trunk_dims):
    return (D % 16 == 0 and D <= 64 and n_branch_dims_equal and len(trunk_dims) >= 3 and trunk_dims[0] == 2 * D
            and trunk_dims[-1] == 1 and len(set(trunk_dims[1:-1])) == 1 and trunk_dims[1] <= 64 and trunk_dims[1] % 4 == 0)


def dual_chain(tables_a, tables_b, ids_a, ids_b, bases_a, bases_b, D, in_act, branch_acts, trunk_dims, trunk_acts, wpack, out,
               out_index=None):
    """Fused two-branch head: see amar_dual_chain_f32. tables_* / ids_* / bases_* are 2-element sequences; out_index (int32 [P]):
    pair p goes to out[out_index[p]] (amar_dual_chain_indexed_f32)."""
    P = out.shape[0]
    arr = lambda vals, ctype: (ctype * 2)(*vals)
    A = arr([_ptr(t, torch.float32, 'A') for t in tables_a], ctypes.c_void_p)
    B = arr([_ptr(t, torch.float32, 'B') for t in tables_b], ctypes.c_void_p)
    IA = arr([_ptr(t, torch.int32, 'ida') for t in ids_a], ctypes.c_void_p)
    IB = arr([_ptr(t, torch.int32, 'idb') for t in ids_b], ctypes.c_void_p)
    for ids in list(ids_a) + list(ids_b):
        if ids is not None and ids.numel() != P:
            raise ValueError("dual_chain: one id per output row expected")
    lda = arr([_ld(t, 'A') for t in tables_a], ctypes.c_int64)
    ldb = arr([_ld(t, 'B') for t in tables_b], ctypes.c_int64)
    ba, bb = arr([int(b) for b in bases_a], ctypes.c_int32), arr([int(b) for b in bases_b], ctypes.c_int32)
    bacts = (ctypes.c_int32 * max(1, len(branch_acts)))(*[ACT_CODES[a] for a in branch_acts])
    tdims = (ctypes.c_int32 * len(trunk_dims))(*trunk_dims)
    tacts = (ctypes.c_int32 * len(trunk_acts))(*[ACT_CODES[a] for a in trunk_acts])
    if out_index is not None and out_index.numel() != P:
        raise ValueError("dual_chain: out_index must have one entry per output row")
    code = load().amar_dual_chain_indexed_f32(A, lda, IA, ba, B, ldb, IB, bb, D, ACT_CODES[in_act], len(branch_acts), bacts,
                                              tdims, tacts, len(trunk_acts), _ptr(wpack, torch.float32, 'wpack'),
                                              _ptr(out, torch.float32, 'out'), _ld(out, 'out'), _ptr(out_index, torch.int32, 'out_index'),
                                              P, _stream())
    _check(code, 'amar_dual_chain_f32' if out_index is None else 'amar_dual_chain_indexed_f32')


def scatter(src, index, dst, window_off=None, n_windows=1):
    """dst[index[t]] = src[t] (rows of a [n, 1] / [n] float32 destination), window by window: see amar_scatter_f32."""
    n = int(index.numel())
    if src.numel() < n or (window_off is not None and window_off.numel() != n_windows + 1):
        raise ValueError("scatter: one source value per index and n_windows + 1 window offsets expected")
    if n == 0:
        return
    _check(load().amar_scatter_f32(_ptr(src, torch.float32, 'src'), _ptr(index, torch.int32, 'index'), _ptr(dst, torch.float32, 'dst'),
                                   _ld(dst, 'dst') if dst.dim() == 2 else 1, n, _ptr(window_off, torch.int32, 'window_off'),
                                   int(n_windows), _stream()), 'amar_scatter_f32')


def copy_columns(src, dst, ids=None, base=0):
    """dst[r, :] = src[ids[r] - base, :] (or src[r, :] without ids); dst may be a column slice."""
    n = ids.numel() if ids is not None else src.shape[0]
    if src.shape[1] != dst.shape[1] or dst.shape[0] != n:
        raise ValueError("copy_columns: shapes differ")
    code = load().amar_copy_columns_f32(_ptr(src, torch.float32, 'src'), _ld(src, 'src'),
                                        _ptr(ids, torch.int32, 'ids'), int(base),
                                        _ptr(dst, torch.float32, 'dst'), _ld(dst, 'dst'),
                                        n, src.shape[1], _stream())
    _check(code, 'amar_copy_columns_f32')


def reduce_layers(cat, n_layers, width, out, mean=False):
    if cat.shape[1] != n_layers * width or tuple(out.shape) != (cat.shape[0], width):
        raise ValueError("reduce_layers: cat [n, n_layers*width], out [n, width] expected")
    code = load().amar_reduce_layers_f32(_ptr(cat, torch.float32, 'cat'), _ld(cat, 'cat'), n_layers, width,
                                         _ptr(out, torch.float32, 'out'), _ld(out, 'out'), cat.shape[0],
                                         1 if mean else 0, _stream())
    _check(code, 'amar_reduce_layers_f32')


def reduce_layers_wsum(cat, n_layers, width, w, out):
    """out = sum_l (w_l * w_l) * cat[:, block l]  (ReductionLayer('w-sum'), reduction.py:36-55); w: device vector of n_layers floats."""
    if cat.shape[1] != n_layers * width or tuple(out.shape) != (cat.shape[0], width) or w.numel() != n_layers or not w.is_contiguous():
        raise ValueError("reduce_layers_wsum: cat [n, n_layers*width], w [n_layers], out [n, width] expected")
    code = load().amar_reduce_layers_wsum_f32(_ptr(cat, torch.float32, 'cat'), _ld(cat, 'cat'), n_layers, width, _ptr(w, torch.float32, 'w'),
                                              _ptr(out, torch.float32, 'out'), _ld(out, 'out'), cat.shape[0], _stream())
    _check(code, 'amar_reduce_layers_wsum_f32')


def reduce_layers_wsum_bwd(cat, n_layers, width, w, d_out, d_cat, dw):
    """Reverse of reduce_layers_wsum: d_cat[:, block l] = w_l^2 d_out, dw[l] = 2 w_l sum(d_out . cat[:, block l]) (deterministic)."""
    n = cat.shape[0]
    if (cat.shape[1] != n_layers * width or tuple(d_cat.shape) != tuple(cat.shape) or tuple(d_out.shape) != (n, width)
            or w.numel() != n_layers or dw.numel() != n_layers or not w.is_contiguous() or not dw.is_contiguous()):
        raise ValueError("reduce_layers_wsum_bwd: cat / d_cat [n, n_layers*width], d_out [n, width], w / dw [n_layers] expected")
    scratch = torch.empty(int(load().amar_reduce_layers_wsum_bwd_scratch()), dtype=torch.float32, device=cat.device)
    code = load().amar_reduce_layers_wsum_bwd_f32(_ptr(cat, torch.float32, 'cat'), _ld(cat, 'cat'), n_layers, width, _ptr(w, torch.float32, 'w'),
                                                  _ptr(d_out, torch.float32, 'd_out'), _ld(d_out, 'd_out'),
                                                  _ptr(d_cat, torch.float32, 'd_cat'), _ld(d_cat, 'd_cat'), _ptr(dw, torch.float32, 'dw'),
                                                  _ptr(scratch, torch.float32, 'scratch'), n, _stream())
    _check(code, 'amar_reduce_layers_wsum_bwd_f32')


def adam_advance(state, learning_rate, beta_1, beta_2):
    if state.numel() != 2 or not state.is_contiguous():
        raise ValueError("adam_advance: state must be 2 contiguous floats (t, lr_t)")
    _check(load().amar_adam_advance_f32(_ptr(state, torch.float32, 'state'), float(learning_rate), float(beta_1), float(beta_2),
                                        _stream()), 'amar_adam_advance_f32')


def adam_dev(w, g, m, v, state, beta_1, beta_2, epsilon, l2=0.0):
    if not (w.is_contiguous() and g.is_contiguous() and m.is_contiguous() and v.is_contiguous()) or \
            not (w.numel() == g.numel() == m.numel() == v.numel()):
        raise ValueError("adam_dev: contiguous tensors of equal size expected")
    code = load().amar_adam_dev_f32(_ptr(w, torch.float32, 'w'), _ptr(g, torch.float32, 'g'), _ptr(m, torch.float32, 'm'),
                                    _ptr(v, torch.float32, 'v'), w.numel(), _ptr(state, torch.float32, 'state'), float(beta_1),
                                    float(beta_2), float(epsilon), float(l2), _stream())
    _check(code, 'amar_adam_dev_f32')


class DeferredGradient:
    """A weight / bias gradient still in partial sums: `partials` [groups, n] (a view of a dense_bwd workspace), to be added in group
    order by its consumer (adam_slot_table -> amar_adam_multi_f32's g_groups).  `materialize()` adds them on the spot."""

    def __init__(self, partials, groups, shape):
        self.partials, self.groups, self.shape = partials, int(groups), tuple(shape)

    def materialize(self):
        return self.partials.view(self.groups, -1).sum(0).view(self.shape)


class AdamSlot(ctypes.Structure):
    """include/amar_hip.h: amar_adam_slot"""
    _fields_ = [('w', ctypes.c_void_p), ('g', ctypes.c_void_p), ('m', ctypes.c_void_p), ('v', ctypes.c_void_p),
                ('n', ctypes.c_int64), ('first_block', ctypes.c_int64), ('l2', ctypes.c_float), ('g_groups', ctypes.c_int32)]


def adam_slot_table(entries):
    """entries: [(w, g, m, v, l2)] contiguous fp32 tensors -> (host uint8 tensor holding the slot table, total blocks)."""
    table = (AdamSlot * len(entries))()
    block = 0
    for k, (w, g, m, v, l2) in enumerate(entries):
        groups = 0
        if isinstance(g, DeferredGradient):                          # partial gradients [groups][n] left by dense_bwd(defer=True)
            g, groups = g.partials, g.groups
            if g.numel() != groups * w.numel():
                raise ValueError("adam_slot_table: deferred gradient of the wrong size")
        elif g.numel() != w.numel():
            raise ValueError("adam_slot_table: gradient of the wrong size")
        if not (w.is_contiguous() and g.is_contiguous() and m.is_contiguous() and v.is_contiguous()) or \
                not (w.numel() == m.numel() == v.numel()):
            raise ValueError("adam_slot_table: contiguous tensors of equal size expected")
        table[k] = AdamSlot(w.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), w.numel(), block, float(l2), int(groups))
        block += (w.numel() + 1023) // 1024
    host = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).clone()
    return host, block


def adam_multi(table_dev, n_slots, total_blocks, state, beta_1, beta_2, epsilon, reg_scale=0.0, loss_acc=None):
    code = load().amar_adam_multi_f32(ctypes.c_void_p(table_dev.data_ptr()), n_slots, total_blocks, _ptr(state, torch.float32, 'state'),
                                      float(beta_1), float(beta_2), float(epsilon), float(reg_scale),
                                      _ptr(loss_acc, torch.float32, 'loss_acc'), _stream())
    _check(code, 'amar_adam_multi_f32')


def sum_into(x, acc, scale=1.0):
    _check(load().amar_sum_into_f32(_ptr(x, torch.float32, 'x'), x.numel(), float(scale), _ptr(acc, torch.float32, 'acc'), _stream()),
           'amar_sum_into_f32')


def topk_segmented(seg_ptr, item_ids, scores, k):
    n_users = seg_ptr.numel() - 1
    out_items = torch.empty((n_users, k), dtype=torch.int32, device=scores.device)
    out_scores = torch.empty((n_users, k), dtype=torch.float32, device=scores.device)
    code = load().amar_topk_segmented_f32(
        _ptr(seg_ptr, torch.int32, 'seg_ptr'), _ptr(item_ids, torch.int32, 'item_ids'),
        _ptr(scores, torch.float32, 'scores'), n_users, k, _ptr(out_items), _ptr(out_scores), _stream())
    _check(code, 'amar_topk_segmented_f32')
    return out_items, out_scores


# ---- training step ----------------------------------------------------------------------------------------------
def act_bwd(dY, Y, dZ, act):
    """dZ = dY * act'(Y) with Y the layer output (dZ may alias dY)."""
    if tuple(dY.shape) != tuple(Y.shape) or tuple(dZ.shape) != tuple(Y.shape):
        raise ValueError("act_bwd: shapes differ")
    code = load().amar_act_bwd_f32(_ptr(dY, torch.float32, 'dY'), _ld(dY, 'dY'), _ptr(Y, torch.float32, 'Y'), _ld(Y, 'Y'),
                                   _ptr(dZ, torch.float32, 'dZ'), _ld(dZ, 'dZ'), Y.shape[0], Y.shape[1], ACT_CODES[act], _stream())
    _check(code, 'amar_act_bwd_f32')


def wgrad(X, dZ, dW=None, db=None):
    """dW = X^T . dZ and / or db = column sums of dZ (deterministic two-stage reduction)."""
    M, N = dZ.shape
    K = X.shape[1] if X is not None else 0
    if dW is not None and (X is None or X.shape[0] != M or tuple(dW.shape) != (K, N) or not dW.is_contiguous()):
        raise ValueError("wgrad: X [M, K], dZ [M, N], dW [K, N] contiguous expected")
    if db is not None and (db.numel() != N or not db.is_contiguous()):
        raise ValueError("wgrad: db must be a contiguous [N] vector")
    lib = load()
    n_scratch = lib.amar_wgrad_scratch_floats(M, K if dW is not None else 0, N)
    scratch = torch.empty(max(int(n_scratch), 1), dtype=torch.float32, device=dZ.device)
    code = lib.amar_wgrad_f32(_ptr(X if dW is not None else None, torch.float32, 'X'), _ld(X, 'X') if dW is not None else 0,
                              _ptr(dZ, torch.float32, 'dZ'), _ld(dZ, 'dZ'), M, K, N,
                              _ptr(dW, torch.float32, 'dW'), _ptr(db, torch.float32, 'db'), _ptr(scratch), _stream())
    _check(code, 'amar_wgrad_f32')


def dense_stack_supported(dims):
    """amar_dense_stack_f32 takes a stack of these widths (at most 4 layers, every width <= 128)."""
    return 2 <= len(dims) <= 5 and all(1 <= int(d) <= 128 for d in dims)


def _dense_stack_args(X, weights, biases, acts, outs, ids=None, xcopy=None):
    """The argument list of amar_dense_stack_f32 (without the stream) for one stack, checked."""
    n = len(weights)
    M = int(ids.numel()) if ids is not None else int(X.shape[0])
    dims = [int(weights[0].shape[0])] + [int(w.shape[1]) for w in weights]
    if X.shape[1] != dims[0] or len(biases) != n or len(acts) != n or len(outs) != n:
        raise ValueError("dense_stack: X [*, K_0] and one kernel / bias / activation / output per layer expected")
    for l, (w, b, y) in enumerate(zip(weights, biases, outs)):
        if tuple(w.shape) != (dims[l], dims[l + 1]) or not w.is_contiguous() or tuple(y.shape) != (M, dims[l + 1]):
            raise ValueError("dense_stack: layer {}: W [K, N] contiguous and Y [M, N] expected".format(l))
        if b is not None and (b.numel() != dims[l + 1] or not b.is_contiguous()):
            raise ValueError("dense_stack: layer {}: bias must be a contiguous [N] vector".format(l))
    if xcopy is not None and tuple(xcopy.shape) != (M, dims[0]):
        raise ValueError("dense_stack: xcopy must be [M, K_0]")
    arr_p = ctypes.c_void_p * n
    wp = arr_p(*[_ptr(w, torch.float32, 'W') for w in weights])
    bp = arr_p(*[_ptr(b, torch.float32, 'bias') for b in biases])
    yp = arr_p(*[_ptr(y, torch.float32, 'Y') for y in outs])
    ld = (ctypes.c_int64 * n)(*[_ld(y, 'Y') for y in outs])
    dm = (ctypes.c_int32 * (n + 1))(*dims)
    ac = (ctypes.c_int32 * n)(*[ACT_CODES[a] for a in acts])
    return [_ptr(X, torch.float32, 'X'), _ld(X, 'X'), _ptr(ids, torch.int32, 'ids'), _ptr(xcopy, torch.float32, 'xcopy'),
            _ld(xcopy, 'xcopy') if xcopy is not None else 0, n, wp, bp, dm, ac, yp, ld, M]


def dense_stack(X, weights, biases, acts, outs, ids=None, xcopy=None):
    """A Dense stack forward in one launch with every layer's output kept (amar_dense_stack_f32): y_0 = X[ids], outs[l] = act_l(y_l . W_l + b_l).
    outs[l] may be column slices of wider buffers; xcopy ([M, K_0]): also receives the gathered input rows."""
    args = _dense_stack_args(X, weights, biases, acts, outs, ids=ids, xcopy=xcopy)
    _check(load().amar_dense_stack_f32(*args, _stream()), 'amar_dense_stack_f32')


class DenseStackDesc(ctypes.Structure):
    """include/amar_hip.h: amar_dense_stack_desc"""
    _fields_ = [('X', ctypes.c_void_p), ('ldx', ctypes.c_int64), ('ids', ctypes.c_void_p), ('Xcopy', ctypes.c_void_p), ('ldxc', ctypes.c_int64),
                ('n_layers', ctypes.c_int32), ('W', ctypes.c_void_p), ('bias', ctypes.c_void_p), ('dims', ctypes.c_void_p), ('acts', ctypes.c_void_p),
                ('Y', ctypes.c_void_p), ('ldy', ctypes.c_void_p), ('M', ctypes.c_int64)]


class DenseStackBwdDesc(ctypes.Structure):
    """include/amar_hip.h: amar_dense_stack_bwd_desc"""
    _fields_ = [('dYtop', ctypes.c_void_p), ('lddy', ctypes.c_int64), ('Ytop', ctypes.c_void_p), ('ldytop', ctypes.c_int64), ('n_layers', ctypes.c_int32),
                ('X', ctypes.c_void_p), ('ldx', ctypes.c_void_p), ('W', ctypes.c_void_p), ('dims', ctypes.c_void_p), ('acts', ctypes.c_void_p),
                ('dX0', ctypes.c_void_p), ('lddx0', ctypes.c_int64), ('dW', ctypes.c_void_p), ('db', ctypes.c_void_p), ('workspace', ctypes.c_void_p),
                ('flags', ctypes.c_int32), ('M', ctypes.c_int64)]


def _as_pointer(v):
    """ctypes arrays (host arrays of device pointers / values) and c_void_p values as plain addresses for a Structure field."""
    if isinstance(v, ctypes.Array):
        return ctypes.cast(v, ctypes.c_void_p).value
    return v.value if isinstance(v, ctypes.c_void_p) else v


def dense_stack_pair(first, second):
    """Two independent Dense stacks forward in ONE launch (amar_dense_stack_pair_f32).  first / second: dicts of dense_stack's arguments
    (X, weights, biases, acts, outs, ids, xcopy).  Same results as two dense_stack calls."""
    a0, a1 = _dense_stack_args(**first), _dense_stack_args(**second)
    d0, d1 = DenseStackDesc(*[_as_pointer(v) for v in a0]), DenseStackDesc(*[_as_pointer(v) for v in a1])
    _check(load().amar_dense_stack_pair_f32(ctypes.byref(d0), ctypes.byref(d1), _stream()), 'amar_dense_stack_pair_f32')


def dense_stack_bwd_supported(dims, M):
    """amar_dense_stack_bwd_f32 takes this stack (at most 4 layers no wider than 128, batch-sized operands: M <= 4 096 rows)."""
    return dense_stack_supported(dims) and 1 <= int(M) <= 4096 and os.environ.get('AMAR_DENSE_STACK_BWD', '1') != '0'


def dense_stack_bwd_workspace(M, dims, device):
    n = len(dims) - 1
    dm = (ctypes.c_int32 * (n + 1))(*[int(d) for d in dims])
    floats = int(load().amar_dense_stack_bwd_workspace_floats(int(M), n, dm))
    return torch.empty(max(floats, 4), dtype=torch.float32, device=device)


def _dense_stack_bwd_args(dYtop, Ytop, inputs, weights, acts, workspace, dWs, dbs, dX0=None, defer=False):
    """The argument list of amar_dense_stack_bwd_f32 (without the stream) for one stack, checked, and its (n, M, dims)."""
    n = len(weights)
    M = int(dYtop.shape[0])
    dims = [int(weights[0].shape[0])] + [int(w.shape[1]) for w in weights]
    if len(inputs) != n or len(acts) != n or len(dWs) != n or len(dbs) != n or dYtop.shape[1] != dims[-1]:
        raise ValueError("dense_stack_bwd: one input / kernel / activation / dW / db per layer and dYtop [M, N_last] expected")
    for l in range(n):
        if tuple(inputs[l].shape) != (M, dims[l]) or tuple(weights[l].shape) != (dims[l], dims[l + 1]) or not weights[l].is_contiguous() or \
                tuple(dWs[l].shape) != (dims[l], dims[l + 1]) or not dWs[l].is_contiguous() or dbs[l].numel() != dims[l + 1]:
            raise ValueError("dense_stack_bwd: layer {}: shapes".format(l))
    if dX0 is not None and tuple(dX0.shape) != (M, dims[0]):
        raise ValueError("dense_stack_bwd: dX0 must be [M, K_0]")
    arr_p = ctypes.c_void_p * n
    xp = arr_p(*[_ptr(x, torch.float32, 'X') for x in inputs])
    lx = (ctypes.c_int64 * n)(*[_ld(x, 'X') for x in inputs])
    wp = arr_p(*[_ptr(w, torch.float32, 'W') for w in weights])
    dwp = arr_p(*[_ptr(w, torch.float32, 'dW') for w in dWs])
    dbp = arr_p(*[_ptr(b, torch.float32, 'db') for b in dbs])
    dm = (ctypes.c_int32 * (n + 1))(*dims)
    ac = (ctypes.c_int32 * n)(*[ACT_CODES[a] for a in acts])
    if workspace.numel() < load().amar_dense_stack_bwd_workspace_floats(M, n, dm):
        raise ValueError("dense_stack_bwd: workspace too small (capi.dense_stack_bwd_workspace)")
    args = [_ptr(dYtop, torch.float32, 'dYtop'), _ld(dYtop, 'dYtop'), _ptr(Ytop, torch.float32, 'Ytop'), _ld(Ytop, 'Ytop') if Ytop is not None else 0,
            n, xp, lx, wp, dm, ac, _ptr(dX0, torch.float32, 'dX0'), _ld(dX0, 'dX0') if dX0 is not None else 0, dwp, dbp,
            _ptr(workspace, torch.float32, 'workspace'), DENSE_BWD_DEFER if defer else 0, M]
    return args, (n, M, dims)


def _deferred_stack_gradients(workspace, n, M, dims):
    g = int(load().amar_dense_stack_bwd_groups(int(M)))
    out, off = [], 4
    for l in range(n):
        kn, nn = dims[l] * dims[l + 1], dims[l + 1]
        out.append((DeferredGradient(workspace[off:off + g * kn], g, (dims[l], dims[l + 1])),
                    DeferredGradient(workspace[off + g * kn:off + g * (kn + nn)], g, (nn,))))
        off += g * (kn + nn)
    return out


def dense_stack_bwd(dYtop, Ytop, inputs, weights, acts, workspace, dWs, dbs, dX0=None, defer=False):
    """The reverse pass of a Dense stack in one launch (amar_dense_stack_bwd_f32).  inputs[l] = layer l's input, Ytop = the last layer's
    output (None: dYtop already is the last pre-activation's gradient).  defer=True: dWs / dbs are not written; returns one
    (DeferredGradient dW, DeferredGradient db) per layer."""
    args, (n, M, dims) = _dense_stack_bwd_args(dYtop, Ytop, inputs, weights, acts, workspace, dWs, dbs, dX0=dX0, defer=defer)
    _check(load().amar_dense_stack_bwd_f32(*args, _stream()), 'amar_dense_stack_bwd_f32')
    return _deferred_stack_gradients(workspace, n, M, dims) if defer else None


def dense_stack_bwd_pair(first, second):
    """The reverse passes of two independent Dense stacks in ONE launch (amar_dense_stack_bwd_pair_f32).  first / second: dicts of
    dense_stack_bwd's arguments.  Returns the two results dense_stack_bwd would have returned."""
    (a0, m0), (a1, m1) = _dense_stack_bwd_args(**first), _dense_stack_bwd_args(**second)
    d0, d1 = DenseStackBwdDesc(*[_as_pointer(v) for v in a0]), DenseStackBwdDesc(*[_as_pointer(v) for v in a1])
    _check(load().amar_dense_stack_bwd_pair_f32(ctypes.byref(d0), ctypes.byref(d1), _stream()), 'amar_dense_stack_bwd_pair_f32')
    return (_deferred_stack_gradients(first['workspace'], *m0) if first.get('defer') else None,
            _deferred_stack_gradients(second['workspace'], *m1) if second.get('defer') else None)


def dense_bwd_supported(K, N):
    """amar_dense_bwd_f32 can take this layer (act', dX, dW, db: two launches instead of four)."""
    return 1 <= K <= 128 and 1 <= N <= 128


def dense_bwd_enabled():
    """Whether the training tapes use amar_dense_bwd_f32 (AMAR_DENSE_BWD=0: the separate kernels).  Measured at ml1m(s=1), BasicGCN
    16 x 2, batch 1 024 (tools/profile_train.sh): 703 ms of kernel time per 1 482 batches against 768."""
    return os.environ.get('AMAR_DENSE_BWD', '1') != '0'


def dense_stack_enabled():
    """Whether the training tapes run a Dense stack's forward as ONE launch (amar_dense_stack_f32; AMAR_DENSE_STACK=0: layer by layer)."""
    return os.environ.get('AMAR_DENSE_STACK', '1') != '0'


def dense_bwd_workspace(M, K, N, device):
    """A workspace for dense_bwd calls of this shape (allocate once, reuse every batch: it holds the workgroups' partial gradients)."""
    n = int(load().amar_dense_bwd_workspace_floats(int(M), int(K), int(N)))
    return torch.zeros(max(n, 4), dtype=torch.float32, device=device)


DENSE_BWD_DEFER = 0x100


DENSE_BWD_ACCUM_DX = 0x200


def dense_bwd(X, Y, dY, W, act, workspace, dX=None, dW=None, db=None, defer=False, dZ=None, accumulate_dx=False, K=None):
    """The reverse pass of one Dense layer in two launches instead of four (amar_dense_bwd_f32): dZ = dY * act'(Y); dX = dZ . W^T; dW = X^T . dZ;
    db = column sums of dZ.  Y is the layer's OUTPUT (None with act None: dY already is dZ); any of dX / dW / db may be None."""
    M, N = dY.shape
    K = W.shape[0] if W is not None else (X.shape[1] if X is not None else int(K or 1))    # (K only sizes the workspace when neither is given)
    if dX is not None and (W is None or tuple(W.shape) != (K, N) or not W.is_contiguous() or tuple(dX.shape) != (M, K)):
        raise ValueError("dense_bwd: W [K, N] contiguous and dX [M, K] expected")
    if dW is not None and (X is None or tuple(X.shape) != (M, K) or tuple(dW.shape) != (K, N) or not dW.is_contiguous()):
        raise ValueError("dense_bwd: X [M, K] and dW [K, N] contiguous expected")
    if db is not None and (db.numel() != N or not db.is_contiguous()):
        raise ValueError("dense_bwd: db must be a contiguous [N] vector")
    if Y is not None and tuple(Y.shape) != (M, N):
        raise ValueError("dense_bwd: Y [M, N] expected")
    if dZ is not None and tuple(dZ.shape) != (M, N):
        raise ValueError("dense_bwd: dZ [M, N] expected")
    lib = load()
    if workspace is None or workspace.numel() < lib.amar_dense_bwd_workspace_floats(M, K, N):
        raise ValueError("dense_bwd: workspace too small (capi.dense_bwd_workspace)")
    code = lib.amar_dense_bwd_f32(
        _ptr(X, torch.float32, 'X'), _ld(X, 'X') if X is not None else 0, _ptr(Y, torch.float32, 'Y'), _ld(Y, 'Y') if Y is not None else 0,
        _ptr(dY, torch.float32, 'dY'), _ld(dY, 'dY'), _ptr(W, torch.float32, 'W'),
        ACT_CODES[act] | (DENSE_BWD_DEFER if defer else 0) | (DENSE_BWD_ACCUM_DX if accumulate_dx else 0),
        _ptr(dX, torch.float32, 'dX'), _ld(dX, 'dX') if dX is not None else 0, _ptr(dW, torch.float32, 'dW'), _ptr(db, torch.float32, 'db'),
        _ptr(dZ, torch.float32, 'dZ'), _ld(dZ, 'dZ') if dZ is not None else 0,
        _ptr(workspace, torch.float32, 'workspace'), M, K, N, _stream())
    _check(code, 'amar_dense_bwd_f32')
    if defer:
        # defer=True: dW / db are NOT written; the partial sums stay in the workspace — returned as DeferredGradients (dW's, db's)
        g = int(lib.amar_dense_bwd_groups(M))
        nw = g * K * N if dW is not None else 0
        return (DeferredGradient(workspace[4:4 + nw], g, (K, N)) if dW is not None else None,
                DeferredGradient(workspace[4 + nw:4 + nw + g * N], g, (N,)) if db is not None else None)
    return None


def bce_grad(p, y, dz, loss_terms):
    B = y.numel()
    code = load().amar_bce_grad_f32(_ptr(p, torch.float32, 'p'), _ld(p, 'p') if p.dim() == 2 else 1, _ptr(y, torch.float32, 'y'),
                                    _ptr(dz, torch.float32, 'dz'), _ptr(loss_terms, torch.float32, 'loss_terms'), B, _stream())
    _check(code, 'amar_bce_grad_f32')


def scatter_add_rows(src, ids, dst, base=0):
    if src.shape[0] != ids.numel() or src.shape[1] != dst.shape[1]:
        raise ValueError("scatter_add_rows: src [M, W], ids [M], dst [*, W] expected")
    code = load().amar_scatter_add_rows_f32(_ptr(src, torch.float32, 'src'), _ld(src, 'src'), _ptr(ids, torch.int32, 'ids'), int(base),
                                            _ptr(dst, torch.float32, 'dst'), _ld(dst, 'dst'), src.shape[0], src.shape[1], _stream())
    _check(code, 'amar_scatter_add_rows_f32')


def add_inplace(dst, src, scale=1.0):
    if tuple(dst.shape) != tuple(src.shape):
        raise ValueError("add_inplace: shapes differ")
    code = load().amar_add_inplace_f32(_ptr(dst, torch.float32, 'dst'), _ld(dst, 'dst'), _ptr(src, torch.float32, 'src'), _ld(src, 'src'),
                                       dst.shape[0], dst.shape[1], float(scale), _stream())
    _check(code, 'amar_add_inplace_f32')


def row_affine(a, scale, out, b=None):
    """out = (a + b) * scale[row] on strided [M, W] blocks (b optional)."""
    M, W = a.shape
    if tuple(out.shape) != (M, W) or scale.numel() != M or (b is not None and tuple(b.shape) != (M, W)):
        raise ValueError("row_affine: a, b, out [M, W] and scale [M] expected")
    code = load().amar_row_affine_f32(_ptr(a, torch.float32, 'a'), _ld(a, 'a'), _ptr(b, torch.float32, 'b'),
                                      _ld(b, 'b') if b is not None else 0, _ptr(scale, torch.float32, 'scale'),
                                      _ptr(out, torch.float32, 'out'), _ld(out, 'out'), M, W, _stream())
    _check(code, 'amar_row_affine_f32')


def l2norm_fwd(z, nrm, inv, y, act='relu'):
    """nrm = l2_normalize(z) per row, inv = the row scale, y = act(nrm)  (GraphSageConv's tail)."""
    M, C = z.shape
    if tuple(nrm.shape) != (M, C) or tuple(y.shape) != (M, C) or inv.numel() != M:
        raise ValueError("l2norm_fwd: z, nrm, y [M, C] and inv [M] expected")
    code = load().amar_l2norm_fwd_f32(_ptr(z, torch.float32, 'z'), _ld(z, 'z'), _ptr(nrm, torch.float32, 'nrm'), _ld(nrm, 'nrm'),
                                      _ptr(inv, torch.float32, 'inv'), _ptr(y, torch.float32, 'y'), _ld(y, 'y'), M, C,
                                      ACT_CODES[act], _stream())
    _check(code, 'amar_l2norm_fwd_f32')


def l2norm_bwd(dy, nrm, inv, dz, act='relu'):
    M, C = dy.shape
    if tuple(nrm.shape) != (M, C) or tuple(dz.shape) != (M, C) or inv.numel() != M:
        raise ValueError("l2norm_bwd: dy, nrm, dz [M, C] and inv [M] expected")
    code = load().amar_l2norm_bwd_f32(_ptr(dy, torch.float32, 'dy'), _ld(dy, 'dy'), _ptr(nrm, torch.float32, 'nrm'), _ld(nrm, 'nrm'),
                                      _ptr(inv, torch.float32, 'inv'), _ptr(dz, torch.float32, 'dz'), _ld(dz, 'dz'), M, C,
                                      ACT_CODES[act], _stream())
    _check(code, 'amar_l2norm_bwd_f32')


def gat_bwd(rowptr, colidx, H, s_self, s_neigh, Y, dY, bias, a_self, a_neigh, self_loop=True):
    """Reverse of gat_layer. Returns (dout [n, C], ds [n], dt [n], dH [n, C])."""
    n = rowptr.numel() - 1
    C = H.shape[1]
    if tuple(Y.shape) != (n, C) or tuple(dY.shape) != (n, C) or H.shape[0] != n or bias.numel() != C or \
            a_self.numel() != C or a_neigh.numel() != C:
        raise ValueError("gat_bwd: H, Y, dY [n, C]; bias, a_self, a_neigh [C] expected")
    dev = H.device
    dout = torch.empty((n, C), dtype=torch.float32, device=dev)
    scratch = torch.empty(3 * n, dtype=torch.float32, device=dev)
    ds, dt = torch.empty(n, dtype=torch.float32, device=dev), torch.empty(n, dtype=torch.float32, device=dev)
    dH = torch.empty((n, C), dtype=torch.float32, device=dev)
    code = load().amar_gat_bwd_f32(
        _ptr(rowptr, torch.int32, 'rowptr'), _ptr_entries(colidx, torch.int32, 'colidx'), _ptr(H, torch.float32, 'H'), _ld(H, 'H'), C,
        _ptr(s_self, torch.float32, 's_self'), _ptr(s_neigh, torch.float32, 's_neigh'), _ptr(Y, torch.float32, 'Y'), _ld(Y, 'Y'),
        _ptr(dY, torch.float32, 'dY'), _ld(dY, 'dY'), _ptr(bias, torch.float32, 'bias'), _ptr(a_self, torch.float32, 'a_self'),
        _ptr(a_neigh, torch.float32, 'a_neigh'), _ptr(dout), _ptr(scratch), _ptr(ds), _ptr(dt), _ptr(dH), C,
        1 if self_loop else 0, n, _stream())
    _check(code, 'amar_gat_bwd_f32')
    return dout, ds, dt, dH


def attention_mix(a, b, ta, tb, out):
    """out = wa * a + (1 - wa) * b with wa = sigmoid(tanh(ta) - tanh(tb)) per feature (FusionLayer 'attention')."""
    M, D = a.shape
    if any(tuple(t.shape) != (M, D) for t in (b, ta, tb, out)):
        raise ValueError("attention_mix: five [M, D] blocks expected")
    code = load().amar_attention_mix_f32(_ptr(a, torch.float32, 'a'), _ld(a, 'a'), _ptr(b, torch.float32, 'b'), _ld(b, 'b'),
                                         _ptr(ta, torch.float32, 'ta'), _ld(ta, 'ta'), _ptr(tb, torch.float32, 'tb'), _ld(tb, 'tb'),
                                         _ptr(out, torch.float32, 'out'), _ld(out, 'out'), M, D, _stream())
    _check(code, 'amar_attention_mix_f32')


def attention_mix_bwd(dout, a, b, ta, tb):
    """Returns (dA, dB, dTA, dTB), contiguous [M, D]."""
    M, D = a.shape
    if any(tuple(t.shape) != (M, D) for t in (dout, b, ta, tb)):
        raise ValueError("attention_mix_bwd: five [M, D] blocks expected")
    outs = [torch.empty((M, D), dtype=torch.float32, device=a.device) for _ in range(4)]
    code = load().amar_attention_mix_bwd_f32(_ptr(dout, torch.float32, 'dout'), _ld(dout, 'dout'), _ptr(a, torch.float32, 'a'), _ld(a, 'a'),
                                             _ptr(b, torch.float32, 'b'), _ld(b, 'b'), _ptr(ta, torch.float32, 'ta'), _ld(ta, 'ta'),
                                             _ptr(tb, torch.float32, 'tb'), _ld(tb, 'tb'), _ptr(outs[0]), _ptr(outs[1]), _ptr(outs[2]),
                                             _ptr(outs[3]), M, D, _stream())
    _check(code, 'amar_attention_mix_bwd_f32')
    return outs


def add3_act(a, b, c, out, act='relu'):
    M, W = a.shape
    if any(tuple(t.shape) != (M, W) for t in (b, c, out)):
        raise ValueError("add3_act: four [M, W] blocks expected")
    code = load().amar_add3_act_f32(_ptr(a, torch.float32, 'a'), _ld(a, 'a'), _ptr(b, torch.float32, 'b'), _ld(b, 'b'),
                                    _ptr(c, torch.float32, 'c'), _ld(c, 'c'), _ptr(out, torch.float32, 'out'), _ld(out, 'out'),
                                    M, W, ACT_CODES[act], _stream())
    _check(code, 'amar_add3_act_f32')


def locality_scale(x, w, out):
    """out = x * sigmoid(w[row])  (DGCFConv's LocalityAdaptive)."""
    M, W = x.shape
    if tuple(out.shape) != (M, W) or w.numel() != M or not w.is_contiguous():
        raise ValueError("locality_scale: x, out [M, W] and contiguous w [M] expected")
    code = load().amar_locality_scale_f32(_ptr(x, torch.float32, 'x'), _ld(x, 'x'), _ptr(w, torch.float32, 'w'),
                                          _ptr(out, torch.float32, 'out'), _ld(out, 'out'), M, W, _stream())
    _check(code, 'amar_locality_scale_f32')


def locality_scale_bwd(dout, x, w, dx, dw, accumulate=False):
    M, W = x.shape
    if tuple(dout.shape) != (M, W) or tuple(dx.shape) != (M, W) or w.numel() != M or dw.numel() != M:
        raise ValueError("locality_scale_bwd: dout, x, dx [M, W]; w, dw [M] expected")
    code = load().amar_locality_scale_bwd_f32(_ptr(dout, torch.float32, 'dout'), _ld(dout, 'dout'), _ptr(x, torch.float32, 'x'), _ld(x, 'x'),
                                              _ptr(w, torch.float32, 'w'), _ptr(dx, torch.float32, 'dx'), _ld(dx, 'dx'),
                                              _ptr(dw, torch.float32, 'dw'), M, W, 1 if accumulate else 0, _stream())
    _check(code, 'amar_locality_scale_bwd_f32')


def transpose(src):
    K, N = src.shape
    if not src.is_contiguous():
        raise ValueError("transpose: contiguous [K, N] expected")
    dst = torch.empty((N, K), dtype=torch.float32, device=src.device)
    _check(load().amar_transpose_f32(_ptr(src, torch.float32, 'src'), K, N, _ptr(dst), _stream()), 'amar_transpose_f32')
    return dst


def adam(w, g, m, v, lr_t, beta_1, beta_2, epsilon, l2=0.0):
    if not (w.is_contiguous() and g.is_contiguous() and m.is_contiguous() and v.is_contiguous()) or \
            not (w.numel() == g.numel() == m.numel() == v.numel()):
        raise ValueError("adam: contiguous tensors of equal size expected")
    code = load().amar_adam_f32(_ptr(w, torch.float32, 'w'), _ptr(g, torch.float32, 'g'), _ptr(m, torch.float32, 'm'),
                                _ptr(v, torch.float32, 'v'), w.numel(), float(lr_t), float(beta_1), float(beta_2), float(epsilon),
                                float(l2), _stream())
    _check(code, 'amar_adam_f32')
