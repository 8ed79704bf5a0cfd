"""ctypes binding of libcna_hip.so (ABI: include/cna_hip.h).

There is deliberately no fallback: if the shared library has not been built, or no GPU is
visible, every call raises.  Build with ``python -c "import __graft_entry__ as g; g.build()"``
(or ``make -C cna_amd/csrc``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libcna_hip.so')

c_ctx = C.c_void_p
c_i64p = C.POINTER(C.c_int64)
c_i32p = C.POINTER(C.c_int32)
c_f64p = C.POINTER(C.c_double)
c_u8p = C.POINTER(C.c_uint8)

# name -> (restype, argtypes); one entry per function declared in include/cna_hip.h
SIGNATURES = {
    'cna_last_error': (C.c_char_p, []),
    'cna_abi_version': (C.c_int, []),
    'cna_device_count': (C.c_int, [C.POINTER(C.c_int)]),
    'cna_ctx_create': (C.c_int, [C.c_int, C.POINTER(c_ctx)]),
    'cna_ctx_destroy': (C.c_int, [c_ctx]),
    'cna_ctx_sync': (C.c_int, [c_ctx]),
    'cna_ctx_device_bytes': (C.c_int, [c_ctx, c_i64p]),
    'cna_set_state_f32': (C.c_int, [c_ctx, C.c_int]),
    'cna_comm_unique_id': (C.c_int, [C.c_void_p]),
    'cna_comm_init': (C.c_int, [c_ctx, C.c_int, C.c_int, C.c_void_p]),
    'cna_graph_upload': (C.c_int, [c_ctx, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_int]),
    'cna_comm_info': (C.c_int, [c_ctx, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'cna_comm_selftest': (C.c_int, [c_ctx, C.c_double, C.POINTER(C.c_int)]),
    'cna_comm_init_shm': (C.c_int, [c_ctx, C.c_int, C.c_int, C.c_char_p, C.c_int64]),
    'cna_set_halo': (C.c_int, [c_ctx, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'cna_set_cell_order': (C.c_int, [c_ctx, C.c_void_p]),
    'cna_graph_reorder': (C.c_int, [c_ctx, C.c_void_p]),
    'cna_set_local_view': (C.c_int, [c_ctx, C.c_int]),
    'cna_colsums': (C.c_int, [c_ctx, C.c_double]),
    'cna_fetch_colsums': (C.c_int, [c_ctx, C.c_void_p]),
    'cna_set_samples': (C.c_int, [c_ctx, C.c_void_p, C.c_int, C.c_void_p]),
    'cna_restart_nam': (C.c_int, [c_ctx]),
    'cna_nam_step': (C.c_int, [c_ctx, C.c_int, C.c_int, C.c_int]),
    'cna_nam_select_hint': (C.c_int, [c_ctx, C.c_void_p, C.c_int]),
    'cna_nam_auto': (C.c_int, [c_ctx, C.c_int, C.POINTER(C.c_int), C.c_void_p]),
    'cna_nam_auto_launch': (C.c_int, [c_ctx, C.c_int]),
    'cna_nam_auto_finish': (C.c_int, [c_ctx, C.POINTER(C.c_int), C.c_void_p]),
    'cna_stat_qc': (C.c_int, [c_ctx, c_f64p, c_f64p, c_i64p]),
    'cna_fetch_cell_stat': (C.c_int, [c_ctx, C.c_void_p, C.c_int64]),
    'cna_dense_load': (C.c_int, [c_ctx, C.c_void_p, C.c_int]),
    'cna_dense_step': (C.c_int, [c_ctx]),
    'cna_dense_fetch': (C.c_int, [c_ctx, C.c_void_p]),
    'cna_batch_kurtosis': (C.c_int, [c_ctx, C.c_int, C.c_void_p, C.c_int]),
    'cna_zero_variance': (C.c_int, [c_ctx, C.c_void_p, C.c_int, C.c_void_p, c_i64p]),
    'cna_select': (C.c_int, [c_ctx, C.c_void_p, C.c_int64, C.c_void_p, C.c_int]),
    'cna_set_resid_factors': (C.c_int, [c_ctx, C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    'cna_select_checked': (C.c_int, [c_ctx, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.POINTER(C.c_int64)]),
    'cna_select_standardized': (C.c_int, [c_ctx, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.POINTER(C.c_int64),
                                          C.c_void_p, C.POINTER(C.c_double)]),
    'cna_upload_x': (C.c_int, [c_ctx, C.c_void_p, C.c_int64, C.c_int]),
    'cna_resid_apply': (C.c_int, [c_ctx, C.c_void_p, C.c_int]),
    'cna_resid_lowrank': (C.c_int, [c_ctx, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, c_f64p]),
    'cna_resid_lowrank_bk': (C.c_int, [c_ctx, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, c_f64p, C.c_void_p, C.c_int, c_f64p]),
    'cna_select_resid_bk': (C.c_int, [c_ctx, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, c_f64p, C.c_void_p, C.c_int, c_f64p,
                                      c_i64p, c_i64p, C.POINTER(C.c_int)]),
    'cna_standardize': (C.c_int, [c_ctx, C.c_int]),
    'cna_gram': (C.c_int, [c_ctx, C.c_void_p]),
    'cna_gram_launch': (C.c_int, [c_ctx]),
    'cna_gram_fetch': (C.c_int, [c_ctx, C.c_void_p]),
    'cna_project': (C.c_int, [c_ctx, C.c_void_p, C.c_int, C.c_void_p]),
    'cna_x_identity': (C.c_int, [c_ctx, C.POINTER(C.c_int)]),
    'cna_ncorrs': (C.c_int, [c_ctx, C.c_void_p, C.c_void_p, c_f64p]),
    'cna_null_local': (C.c_int, [c_ctx, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    'cna_condition_phenotypes': (C.c_int, [c_ctx, C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    'cna_null_local_resident': (C.c_int, [c_ctx, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'cna_null_local_launch': (C.c_int, [c_ctx, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    'cna_null_local_prepare': (C.c_int, [c_ctx, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    'cna_null_local_fetch': (C.c_int, [c_ctx, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'cna_null_local_discard': (C.c_int, [c_ctx]),
    'cna_gram_pcs_tests': (C.c_int, [c_ctx, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p,
                                     C.c_void_p, C.POINTER(C.c_int)]),
    'cna_host_top_eig': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, c_f64p, c_f64p]),
    'cna_host_eig_check': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, c_f64p, c_f64p]),
    'cna_null_local_i8_stats': (C.c_int, [c_ctx, C.POINTER(C.c_int), c_i64p, C.POINTER(C.c_int)]),
    'cna_percell_fdr_pinned': (C.c_int, [c_ctx, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    'cna_nam_steps': (C.c_int, [c_ctx, C.c_int]),
    'cna_host_legacy_randn': (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_int64, C.c_void_p]),
    'cna_knn_graph': (C.c_int, [c_ctx, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, c_i64p]),
    'cna_host_hash64': (C.c_uint64, [C.c_void_p, C.c_int64, C.c_int]),
    'cna_host_copy': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int]),
    'cna_host_expand_u16': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int]),
    'cna_host_permuted_nnz': (C.c_int64, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    'cna_host_permute_rows': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    'cna_host_cluster_order': (C.c_int64, [C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    'cna_host_cluster_order_mt': (C.c_int64, [C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    'cna_host_cluster_graph': (C.c_int64, [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'cna_host_set_threads': (None, [C.c_int]),
    'cna_host_argsort_gather': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    'cna_host_draw_start': (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_int64, C.c_int]),
    'cna_host_draw_start_idx': (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    'cna_host_gather_rows': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_int]),
    'cna_host_draw_wait': (C.c_int, []),
    'cna_host_draw_join': (C.c_int, []),
    'cna_host_draw_times': (None, [C.POINTER(C.c_double)]),
    'cna_assoc_begin': (C.c_int, [c_ctx, C.c_int, C.c_void_p, C.c_int]),
    'cna_assoc_begin_part': (C.c_int, [c_ctx, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    'cna_assoc_finish': (C.c_int, [c_ctx, C.c_void_p, C.c_void_p]),
    'cna_assoc_run': (C.c_int, [c_ctx, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'cna_host_draw_then_condition': (C.c_int, [c_ctx, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    'cna_global_test_launch': (C.c_int, [c_ctx, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]),
    'cna_global_test_fetch': (C.c_int, [c_ctx, C.c_void_p, C.c_void_p, C.c_void_p]),
    'cna_select_standardized_fused': (C.c_int, [c_ctx, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_void_p, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int), C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_void_p, C.POINTER(C.c_int)]),
    'cna_reference_thresholds': (C.c_int, [C.c_double, C.c_int, C.c_void_p, C.c_void_p]),
    'cna_percell_coef_launch': (C.c_int, [c_ctx]),
    'cna_percell_coef_wait': (C.c_int, [c_ctx, C.POINTER(C.c_void_p)]),
    'cna_percell_fdr_copy_early': (C.c_int, [c_ctx, C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_int)]),
    'cna_percell_fdr_copied_early': (C.c_int, [c_ctx, C.POINTER(C.c_int)]),
    'cna_fetch_rows': (C.c_int, [c_ctx, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    'cna_project_keep': (C.c_int, [c_ctx, C.c_void_p, C.c_int]),
    'cna_stat_median': (C.c_int, [c_ctx, C.POINTER(C.c_double)]),
    'cna_global_test': (C.c_int, [c_ctx, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_void_p]),
    'cna_obs_counts': (C.c_int, [c_ctx, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'cna_percell_fdr': (C.c_int, [c_ctx, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'cna_matrix_shape': (C.c_int, [c_ctx, C.c_int, c_i64p, C.POINTER(C.c_int)]),
    'cna_fetch_matrix': (C.c_int, [c_ctx, C.c_int, C.c_void_p, C.c_int]),
    'cna_allgather_host': (C.c_int, [c_ctx, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]),
    'cna_prof_enable': (C.c_int, [c_ctx, C.c_int]),
    'cna_prof_reset': (C.c_int, [c_ctx]),
    'cna_prof_get': (C.c_int, [c_ctx, C.c_int, c_f64p, c_i64p]),
    'cna_kernel_name': (C.c_char_p, [C.c_int]),
}

MAT_NAM, MAT_X, MAT_PROJ = 0, 1, 2
ASSOC_DONE, ASSOC_GENERAL, ASSOC_NEED_PCS, ASSOC_STALE = 0, 1, 2, 3
ASSOC_MAXT = 512


class AssocArgs(C.Structure):
    """struct cna_assoc_args (include/cna_hip.h)"""
    _fields_ = [('colmap', C.c_void_p), ('n_sel', C.c_int32), ('r', C.c_int32), ('y', C.c_void_p), ('M', C.c_void_p),
                ('resid_C', C.c_void_p), ('resid_W', C.c_void_p), ('ks', C.c_void_p), ('K', C.c_int32), ('Nnull', C.c_int32),
                ('table', C.c_void_p), ('draw_pending', C.c_int32), ('conditioned', C.c_int32), ('use_native_eig', C.c_int32),
                ('coef_first', C.c_int32), ('resid_tol', C.c_double), ('gap_tol', C.c_double), ('coef_dst', C.c_void_p),
                ('fdr_dst', C.c_void_p), ('n_dst', C.c_int64), ('copy_threads', C.c_int32), ('n_verify', C.c_int32),
                ('verify_ptr', C.c_void_p * 4), ('verify_bytes', C.c_int64 * 4), ('verify_hash', C.c_uint64 * 4),
                ('verify_threads', C.c_int32), ('reserved', C.c_int32),
                ('G', C.c_void_p), ('U', C.c_void_p), ('minp', C.c_void_p), ('r2', C.c_void_p), ('kidx', C.c_void_p)]


class AssocOut(C.Structure):
    """struct cna_assoc_out (include/cna_hip.h)"""
    _fields_ = [('status', C.c_int32), ('T', C.c_int32), ('eig_accepted', C.c_int32), ('null_fused', C.c_int32),
                ('coef_in_dst', C.c_int32), ('fdr_in_dst', C.c_int32), ('n_zero', C.c_int64), ('max_abs', C.c_double),
                ('coef_ptr', C.c_void_p), ('fdr_ptr', C.c_void_p), ('t_ms', C.c_double * 16),
                ('thr', C.c_double * ASSOC_MAXT), ('fdr', C.c_double * ASSOC_MAXT), ('runmin', C.c_double * ASSOC_MAXT),
                ('tail_sums', C.c_int64 * ASSOC_MAXT), ('ranks', C.c_int64 * ASSOC_MAXT), ('num_detected', C.c_int64 * ASSOC_MAXT)]

KERNELS = ['colsum', 'nam_first', 'nam_step', 'batch_kurtosis', 'zero_variance', 'select', 'resid_xb',
           'standardize', 'gram', 'gram_reduce', 'ncorrs', 'null_local', 'obs_counts', 'percell_fdr',
           'project_xb', 'transpose', 'rccl', 'condition', 'global_test', 'nam_step_sparse', 'halo_exchange', 'halo_wait']

_lib = None


class CnaHipError(RuntimeError):
    """A libcna_hip.so entry point returned a non-zero status."""


def load():
    """dlopen the library and attach prototypes.  Raises if it is missing -- never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CnaHipError(
            'libcna_hip.so is not built (%s).  cna_amd has no CPU path; build the HIP library with '
            '`python -c "import __graft_entry__ as g; g.build()"` or `make -C cna_amd/csrc`.' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        msg = load().cna_last_error()
        raise CnaHipError('%s failed (status %d): %s' % (what, status, msg.decode() if msg else '?'))


def ptr(a):
    """Raw data pointer of a C-contiguous numpy array (or None)."""
    return None if a is None else a.ctypes.data_as(C.c_void_p)
