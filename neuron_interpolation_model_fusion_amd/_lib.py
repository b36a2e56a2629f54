"""ctypes binding of libnif.so (the C ABI declared in include/nif.h).

The library is built in-tree (``csrc/libnif.so``, see ``__graft_entry__.build``).
There is no fallback: if the shared object is missing or a symbol is absent the
import fails loudly, and every op raises on a non-zero status.
"""
import ctypes
import os

# torch first: libnif.so must bind to the HIP runtime torch has already loaded.  Loading the
# system libamdhip64 before torch's own copy puts two runtimes in one process and the second
# one reports "no ROCm-capable device".
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libnif.so')

NIF_F32, NIF_F64 = 0, 1
NIF_RELU_A, NIF_RELU_B, NIF_SYMMETRIC, NIF_TRANS_A, NIF_TRANS_B, NIF_XTY_SPLIT_BF16 = 1, 2, 4, 8, 16, 32
NIF_LAP_MAX_N = 8192
NIF_LAP_STATIC_MAX_N = 2048
NIF_LAP_SCIPY_ORDER = 1
NIF_LAP_INIT_ONLY = 2
NIF_LAP_RESUME = 4
NIF_LAP_BUDGET_SHIFT = 8
NIF_EMD_MAX_N = 8192
NIF_EIGH_MAX_N = 6144
NIF_LSQ_NORMALIZE, NIF_LSQ_PRIMAL = 1, 2
NIF_EMD_EDGES_GLOBAL = 1
NIF_EMD_COLD_START = 2
NIF_EMD_ARC_POOL = 4
NIF_EMD_ONE_LANE_PATHS = 8
NIF_ERR_WORKSPACE = -2
NIF_ERR_UNSUPPORTED = -3

_c = ctypes
_vp, _i32, _i64, _sz = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_size_t

# name -> (restype, argtypes); mirrors include/nif.h one to one
SIGNATURES = {
    'nif_version': (_i32, []),
    'nif_error_string': (_c.c_char_p, [_i32]),
    'nif_device_count': (_i32, []),
    'nif_profile_enable': (None, [_i32]),
    'nif_profile_collect': (_i32, [_i32, _c.POINTER(_c.c_double), _c.POINTER(_c.c_int64)]),
    'nif_profile_collect_spans': (_i32, [_i32, _c.POINTER(_c.c_float), _c.c_int64, _c.POINTER(_c.c_int64)]),
    'nif_profile_work': (_i32, [_i32, _c.POINTER(_c.c_double)]),
    'nif_xty_workspace_bytes': (_sz, [_i64, _i32, _i32, _i32, _i32]),
    'nif_xty': (_i32, [_vp, _i64, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _i64, _vp, _vp, _vp, _sz, _vp]),
    'nif_xty_sym_gf32': (_i32, [_vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _i64, _vp, _vp, _sz, _vp]),
    'nif_packed_upper_len': (_sz, [_i32, _i32]),
    'nif_xty_sym_packed': (_i32, [_vp, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _sz, _vp]),
    'nif_unpack_upper': (_i32, [_vp, _i32, _i32, _vp, _i64, _i32, _vp, _vp]),
    'nif_cost_dist': (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _i64, _vp]),
    'nif_cost_hf': (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _i64, _vp]),
    'nif_lap_workspace_bytes': (_sz, [_i32]),
    'nif_lap_solve_f32': (_i32, [_vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'nif_lap_solve_f32_static': (_i32, [_vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    'nif_lap_batch_workspace_bytes': (_sz, [_i32, _i32]),
    'nif_lap_solve_batch_f32': (_i32, [_vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'nif_kmeans_workspace_bytes': (_sz, [_i32, _i32]),
    'nif_kmeans_farthest_init': (_i32, [_vp, _i64, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    'nif_kmeans_assign_to_neurons': (_i32, [_vp, _i64, _vp, _i32, _i32, _i32, _vp, _vp]),
    'nif_kmeans_lloyd_step': (_i32, [_vp, _i64, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'nif_kmeans_farthest_init_gf32': (_i32, [_vp, _i64, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    'nif_kmeans_farthest_init_batch': (_i32, [_vp, _i64, _vp, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _vp]),
    'nif_kmeans_farthest_init_batch_gf32': (_i32, [_vp, _i64, _vp, _vp, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _vp]),
    'nif_kmeans_assign_to_neurons_gf32': (_i32, [_vp, _i64, _vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    'nif_kmeans_lloyd_step_gf32': (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'nif_kmeans_lloyd_step_batch': (_i32, [_vp, _i64, _vp, _vp, _vp, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'nif_kmeans_lloyd_step_batch_gf32': (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _sz,
                                                _vp]),
    'nif_kmeans_lloyd_step_range': (_i32, [_vp, _i64, _vp, _vp, _vp, _i32, _i32, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp,
                                           _sz, _vp]),
    'nif_kmeans_lloyd_step_range_gf32': (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _i32, _i32, _i32, _vp, _vp, _vp,
                                                _vp, _vp, _sz, _vp]),
    'nif_kmeans_cost_from_partials': (_i32, [_vp, _i32, _vp, _i32, _vp, _vp]),
    'nif_kmeans_repair_kf_batch': (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _i32, _vp, _vp, _sz, _vp]),
    'nif_kmeans_repair_kf': (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _sz, _vp]),
    'nif_kmeans_centroid_weights': (_i32, [_vp, _i64, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _sz, _vp]),
    'nif_kmeans_targets': (_i32, [_vp, _i64, _i64, _vp, _vp, _i32, _i32, _vp, _i64, _vp, _vp, _sz, _vp]),
    'nif_refit_pairs': (_i32, [_vp, _i64, _vp, _vp, _vp, _i32, _i32, _vp, _vp]),
    'nif_refit_gather': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _c.c_float, _c.c_float, _i32, _i32, _i32, _vp, _vp]),
    'nif_sinkhorn_workspace_bytes': (_sz, [_i32, _i32]),
    'nif_sinkhorn_f32': (_i32, [_vp, _i64, _i32, _i32, _vp, _vp, _c.c_double, _i32, _c.c_double, _vp, _i64, _vp, _vp, _vp, _vp,
                                _vp, _sz, _vp]),
    'nif_emd_workspace_bytes': (_sz, [_i32, _i32]),
    'nif_emd_batch_workspace_bytes': (_sz, [_i32, _i32]),
    'nif_emd_solve_batch_f32': (_i32, [_vp, _i64, _i64, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp,
                                       _vp, _sz, _vp]),
    'nif_emd_solve_f32': (_i32, [_vp, _i64, _i32, _i32, _vp, _vp, _c.c_double, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp,
                                 _sz, _vp]),
    'nif_refit_sparse': (_i32, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _c.c_float, _c.c_float, _i32, _i32, _i32,
                                _vp, _vp]),
    'nif_wmse_workspace_bytes': (_sz, []),
    'nif_wmse_forward': (_i32, [_vp, _vp, _vp, _i64, _i32, _i64, _vp, _vp, _sz, _vp]),
    'nif_wmse_backward': (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i64, _vp, _vp]),
    'nif_lap_count_tight_f32': (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _c.c_double, _vp, _vp]),
    'nif_lap_certify_workspace_bytes': (_sz, [_i32]),
    'nif_lap_certify_unique_f32': (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _c.c_double, _vp, _vp, _sz, _vp]),
    'nif_im2row_f32': (_i32, [_vp, _i64] + [_i32] * 10 + [_vp, _i64, _vp]),
    'nif_lsq_refit_workspace_bytes': (_sz, [_i64, _i32, _i32, _i32]),
    'nif_lsq_refit_f32': (_i32, [_vp, _i64, _i64, _i32, _vp, _i64, _i32, _i32, _vp, _i64, _vp, _i64, _vp,
                                 _c.POINTER(_c.c_int32), _vp, _sz, _vp]),
    'nif_lsq_solve_workspace_bytes': (_sz, [_i32, _i32]),
    'nif_lsq_solve_f64': (_i32, [_vp, _i64, _vp, _i64, _i32, _i32, _c.c_double, _vp, _i64, _c.POINTER(_c.c_int32), _vp, _sz,
                                 _vp]),
    'nif_kmeans_farthest_init_redo_count': (_i32, [_c.POINTER(_c.c_uint32), _vp]),
    'nif_eigh_trd_launch_counts': (_i32, [_c.POINTER(_c.c_uint64)]),
    'nif_eigh_kept_workspace_bytes': (_sz, [_i32]),
    'nif_eigh_kept_f64': (_i32, [_vp, _i64, _i32, _c.c_double, _vp, _vp, _i64, _c.POINTER(_c.c_int32), _vp, _sz, _vp]),
}


class NifError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            f'or `make -C {os.path.dirname(LIB_PATH)}`. There is no CPU fallback.')
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is absent
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(status, what):
    if status != 0:
        msg = lib.nif_error_string(status).decode()
        raise NifError(f'{what} failed with status {status}: {msg}')
