"""ctypes binding of libptgnn_amd.so (the C ABI declared in include/ptgnn_amd.h).

The product path has NO CPU fallback: if the HIP library is missing or an entry point fails,
callers get a RuntimeError (never a silent eager/PyTorch substitute)."""
import ctypes
import os
import threading

import torch  # noqa: F401  (must be imported first: the library binds to torch's HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PTGNN_AMD_LIB") or os.path.join(_HERE, "csrc", "libptgnn_amd.so")

_c = ctypes
_i64, _i32, _vp, _f32 = _c.c_int64, _c.c_int32, _c.c_void_p, _c.c_float

# name -> (restype, argtypes); mirrors include/ptgnn_amd.h one to one
SIGNATURES = {
    "ptgnn_amd_version": (_c.c_int, []),
    "ptgnn_amd_last_error": (_c.c_char_p, []),
    "ptgnn_amd_set_gemm_mode": (_c.c_int, [_c.c_int]),
    "ptgnn_amd_get_gemm_mode": (_c.c_int, []),
    "ptgnn_amd_launch_count": (_i64, [_c.c_int]),
    "ptgnn_amd_launch_name": (_c.c_char_p, [_c.c_int]),
    "ptgnn_amd_csr_workspace_bytes": (_c.c_size_t, [_i64, _i64]),
    "ptgnn_amd_csr_control_bytes": (_c.c_size_t, []),
    "ptgnn_amd_set_plan_path": (_c.c_int, [_c.c_int]),
    "ptgnn_amd_type_bits": (_c.c_int, [_i32]),
    "ptgnn_amd_csr_build": (_c.c_int, [_vp, _vp, _vp, _i32, _i64, _i64, _c.c_int, _vp, _vp, _vp, _vp,
                                       _i32, _vp, _vp, _vp, _vp, _vp, _c.c_size_t, _vp]),
    "ptgnn_amd_hub_workspace_bytes": (_c.c_size_t, [_i64, _i32, _c.c_int]),
    "ptgnn_amd_hub_ticket_count": (_i64, [_i64, _i32]),
    "ptgnn_amd_hub_list": (_c.c_int, [_vp, _i64, _i32, _vp, _vp, _vp]),
    "ptgnn_amd_validate_indices": (_c.c_int, [_vp, _i64, _i64, _vp, _vp]),
    "ptgnn_amd_shard_index_workspace_bytes": (_c.c_size_t, [_i64]),
    "ptgnn_amd_shard_index": (_c.c_int, [_vp, _vp, _vp, _i32, _i64, _i64, _vp, _i32, _i64, _vp, _vp, _vp, _i64, _vp,
                                         _vp, _vp, _c.c_size_t, _vp]),
    "ptgnn_amd_unique_sources_workspace_bytes": (_c.c_size_t, [_i64, _i32]),
    "ptgnn_amd_edge_table_bytes": (_c.c_size_t, []),
    "ptgnn_amd_unique_sources": (_c.c_int, [_vp, _i64, _i32, _i32, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _c.c_size_t,
                                            _vp]),
    "ptgnn_amd_edge_linear_shared_supported": (_c.c_int, [_i32, _i32, _i32]),
    "ptgnn_amd_edge_linear_shared_f32": (_c.c_int, [_vp, _i64, _i64, _i32, _vp, _vp, _i32, _i32, _c.c_int, _vp, _i64,
                                                    _vp]),
    "ptgnn_amd_gather_reduce_f32": (_c.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i32, _i64, _i32, _c.c_int,
                                               _c.c_int, _vp, _vp, _f32, _vp, _i64, _vp, _i64, _i32, _vp,
                                               _vp, _vp, _c.c_size_t, _vp, _vp]),
    "ptgnn_amd_gather_reduce_rows_f32": (_c.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i32, _i64, _i32, _c.c_int,
                                                    _c.c_int, _vp, _vp, _f32, _vp, _i64, _vp, _i64, _i32, _vp,
                                                    _vp, _vp, _c.c_size_t, _vp, _i64, _i64, _vp]),
    "ptgnn_amd_gather_update_supported": (_c.c_int, [_i32, _i32]),
    "ptgnn_amd_gather_update_f32": (_c.c_int, [_vp, _i64, _vp, _vp, _i32, _i64, _i32, _c.c_int, _c.c_int, _vp, _vp, _f32,
                                               _vp, _vp, _i32, _c.c_int, _vp, _i64, _vp]),
    "ptgnn_amd_gather_reduce_masked_f32": (_c.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _i64,
                                                      _i64, _i32, _vp, _vp, _vp, _c.c_size_t, _vp, _vp]),
    "ptgnn_amd_segment_mul_f32": (_c.c_int, [_vp, _i64, _vp, _vp, _i64, _i64, _i32, _vp, _i64, _vp]),
    "ptgnn_amd_linear_f32": (_c.c_int, [_vp, _i64, _i32, _i64, _vp, _i32, _vp, _c.c_int, _vp, _i64,
                                        _vp]),
    "ptgnn_amd_linear_add_f32": (_c.c_int, [_vp, _i64, _i32, _i64, _vp, _i32, _vp, _c.c_int, _vp, _i64, _vp, _i64,
                                            _vp]),
    "ptgnn_amd_batch_offsets_i64": (_c.c_int, [_vp, _i64, _vp, _vp, _i32, _i64, _vp, _vp]),
    "ptgnn_amd_edge_linear_f32": (_c.c_int, [_vp, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _c.c_int, _vp,
                                             _i64, _vp]),
    "ptgnn_amd_edge_linear_feat_f32": (_c.c_int, [_vp, _i64, _i64, _i32, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _i32, _i32,
                                                  _c.c_int, _vp, _i64, _vp]),
    "ptgnn_amd_edge_linear_dropout_f32": (_c.c_int, [_vp, _i64, _i64, _i32, _vp, _vp, _vp, _i32, _i32, _vp, _i64,
                                                     _c.c_int, _c.c_float, _c.c_uint64, _vp]),
    "ptgnn_amd_dropout_bitmask_bytes": (_c.c_size_t, [_i64, _i32]),
    "ptgnn_amd_dropout_bitmask": (_c.c_int, [_i64, _i32, _c.c_float, _c.c_uint64, _vp, _vp]),
    "ptgnn_amd_edge_linear_masked_supported": (_c.c_int, [_i32, _i32, _c.c_int]),
    "ptgnn_amd_edge_linear_masked_f32": (_c.c_int, [_vp, _i64, _i64, _i32, _vp, _vp, _vp, _i32, _i32, _vp, _i64,
                                                    _c.c_int, _c.c_float, _vp, _vp]),
    "ptgnn_amd_edge_weight_grad_masked_supported": (_c.c_int, [_i32, _i32]),
    "ptgnn_amd_edge_weight_grad_masked_f32": (_c.c_int, [_vp, _i64, _i64, _i32, _vp, _vp, _vp, _i64, _i32, _i32,
                                                         _c.c_float, _vp, _vp, _vp, _c.c_size_t, _vp]),
    "ptgnn_amd_edge_wgrad_workspace_bytes": (_c.c_size_t, [_i64, _i32, _i32, _i32]),
    "ptgnn_amd_edge_weight_grad_f32": (_c.c_int, [_vp, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _i64, _i32, _i32,
                                                  _c.c_float, _c.c_uint64, _vp, _vp, _c.c_size_t, _vp]),
    "ptgnn_amd_linear_weight_grad_f32": (_c.c_int, [_vp, _i64, _i32, _vp, _i64, _i64, _i32, _vp, _vp, _vp,
                                                    _c.c_size_t, _vp]),
    "ptgnn_amd_segment_spread_f32": (_c.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _i32, _vp, _i64, _vp]),
    "ptgnn_amd_row_epilogue_f32": (_c.c_int, [_vp, _i64, _i64, _i32, _i32, _vp, _vp, _c.c_float, _vp, _i64, _vp]),
    "ptgnn_amd_row_epilogue_workspace_bytes": (_c.c_size_t, [_i64, _i32]),
    "ptgnn_amd_act_dropout_backward_f32": (_c.c_int, [_vp, _vp, _vp, _c.c_float, _c.c_int, _i64, _vp, _vp]),
    "ptgnn_amd_row_epilogue_backward_f32": (_c.c_int, [_vp, _i64, _vp, _i64, _i64, _i32, _i32, _vp, _c.c_float, _vp,
                                                       _i64, _vp, _vp, _vp, _c.c_size_t, _vp]),
    "ptgnn_amd_gru_cell_f32": (_c.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i32, _i32,
                                          _vp, _i64, _vp]),
    "ptgnn_amd_gru_cell_train_f32": (_c.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i32, _i32,
                                                _vp, _i64, _vp, _vp]),
    "ptgnn_amd_gru_cell_backward_gates_f32": (_c.c_int, [_vp, _i64, _vp, _vp, _i64, _i64, _i32, _vp, _vp, _vp,
                                                         _vp]),
    "ptgnn_amd_gather_rows_f32": (_c.c_int, [_vp, _i64, _vp, _i64, _i32, _vp, _i64, _vp]),
    "ptgnn_amd_weighted_pool_workspace_bytes": (_c.c_size_t, [_i64, _i64, _i32]),
    "ptgnn_amd_weighted_pool_f32": (_c.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _i64, _i32, _vp, _i64, _vp, _c.c_size_t,
                                               _vp]),
    "ptgnn_amd_weighted_pool_backward_workspace_bytes": (_c.c_size_t, [_i64, _i32]),
    "ptgnn_amd_weighted_pool_backward_f32": (_c.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _i64, _i32, _vp, _i64, _vp, _vp,
                                                        _c.c_size_t, _vp]),
}

# The header version this host was written against (include/ptgnn_amd.h: PTGNN_AMD_VERSION).  A stale .so with an
# older ABI (e.g. ptgnn_amd_shard_index before `bad_index_count` joined its signature) is refused at load time.
MIN_VERSION = 102

_lock = threading.Lock()
_lib = None


EUNSUPPORTED = -2    # PTGNN_AMD_EUNSUPPORTED of include/ptgnn_amd.h


class PtgnnAmdError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes library.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise PtgnnAmdError(
                    f"{LIB_PATH} is missing: build it with `python -m ptgnn_amd.build` "
                    "(ptgnn_amd has no CPU/eager fallback by design)")
            lib = ctypes.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)  # AttributeError if the .so is stale
                fn.restype, fn.argtypes = res, args
            have = lib.ptgnn_amd_version()
            if have < MIN_VERSION:
                raise PtgnnAmdError(f"{LIB_PATH} reports ABI version {have}, this host needs >= {MIN_VERSION}: "
                                    "rebuild it with `python -m ptgnn_amd.build --force`")
            _lib = lib
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().ptgnn_amd_last_error()
        raise PtgnnAmdError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")
