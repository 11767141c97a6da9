"""ctypes bindings of include/mxg.h and include/mxh.h (loud failure if a library is missing)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBDIR = os.path.join(_HERE, "lib")

c_i64p = C.POINTER(C.c_int64)
c_i32p = C.POINTER(C.c_int32)
c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)
vp = C.c_void_p


class MxError(RuntimeError):
    pass


class HaloPlan(C.Structure):
    _fields_ = [("nneigh", C.c_int), ("neigh", c_ip), ("send_off", c_i64p),
                ("send_idx", c_i64p), ("recv_off", c_i64p)]


class SolveOpts(C.Structure):
    _fields_ = [("tol", C.c_double), ("restart", C.c_int), ("max_iters", C.c_int),
                ("precond", C.c_int), ("flags", C.c_int),
                ("lmin", C.c_double), ("lmax", C.c_double), ("mg", C.c_void_p)]


class SolveInfo(C.Structure):
    _fields_ = [("iters", C.c_int), ("converged", C.c_int), ("true_res", C.c_double)]


class SolverParamsC(C.Structure):
    _fields_ = [("nev", C.c_int), ("basis", C.c_int), ("block_size", C.c_int),
                ("max_restarts", C.c_int), ("tol", C.c_double), ("shift", C.c_double),
                ("lin_type", C.c_char_p), ("lin_basis", C.c_int), ("lin_tol", C.c_double),
                ("prec_type", C.c_char_p), ("verbose", C.c_int), ("seed", C.c_uint64),
                ("amg_smoother", C.c_char_p), ("amg_sweeps", C.c_int), ("eig_type", C.c_char_p)]


# name -> (restype, argtypes); the list doubles as the export check of the CPU test-suite
MXG_SIGNATURES = {
    "mxg_last_error": (C.c_char_p, []),
    "mxg_get_unique_id": (C.c_int, [vp]),
    "mxg_ctx_create": (C.c_int, [C.c_int, C.c_int, C.c_int, vp, C.POINTER(vp)]),
    "mxg_ctx_destroy": (C.c_int, [vp]),
    "mxg_ctx_sync": (C.c_int, [vp]),
    "mxg_ctx_rank": (C.c_int, [vp]),
    "mxg_ctx_nranks": (C.c_int, [vp]),
    "mxg_ctx_comm_ranks": (C.c_int, [vp]),
    "mxg_timer_start": (C.c_int, [vp]),
    "mxg_timer_stop": (C.c_int, [vp, C.POINTER(C.c_float)]),
    "mxg_barrier": (C.c_int, [vp]),
    "mxg_local_world_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
    "mxg_local_world_destroy": (C.c_int, [vp]),
    "mxg_ctx_create_local": (C.c_int, [C.c_int, C.c_int, C.c_int, vp, C.POINTER(vp)]),
    "mxg_comm_allgather_i64": (C.c_int, [vp, C.c_int64, c_i64p]),
    "mxg_comm_allgatherv": (C.c_int, [vp, vp, C.c_int64, c_i64p, vp]),
    "mxg_comm_allreduce_host": (C.c_int, [vp, c_dp, C.c_int]),
    "mxg_csr_create": (C.c_int, [vp, C.c_int64, C.c_int64, C.c_int64, vp, vp, vp, C.c_int,
                                 C.POINTER(HaloPlan), C.POINTER(vp)]),
    "mxg_csr_destroy": (C.c_int, [vp]),
    "mxg_csr_multiply": (C.c_int, [vp, vp, vp, C.POINTER(vp)]),
    "mxg_csr_add": (C.c_int, [vp, vp, C.c_double, vp, C.c_double, C.POINTER(vp)]),
    "mxg_csr_add_z": (C.c_int, [vp, vp, c_dp, vp, c_dp, C.POINTER(vp)]),
    "mxg_csr_purge": (C.c_int, [vp, vp, C.c_double, C.POINTER(vp)]),
    "mxg_csr_download": (C.c_int, [vp, vp, vp, vp]),
    "mxg_csr_info": (C.c_int, [vp, c_i64p, c_i64p, c_i64p, c_i64p, c_ip]),
    "mxg_csr_apply": (C.c_int, [vp, vp, vp, vp]),
    "mxg_csr_apply_bytes": (C.c_int64, [vp, C.c_int]),
    "mxg_csr_compress": (C.c_int, [vp, c_i64p, c_i64p, c_i64p]),
    "mxg_csr_compress_from": (C.c_int, [vp, vp, vp, vp, c_i64p, c_i64p, c_i64p]),
    "mxg_csr_set_kernel": (C.c_int, [vp, C.c_int]),
    "mxg_csr_kernel_name": (C.c_char_p, [vp]),
    "mxg_bench_stream": (C.c_int, [vp, C.c_int64, C.c_int64, C.c_int, C.POINTER(C.c_float)]),
    "mxg_debug_tile_plan_apply": (C.c_int, [C.c_int64, C.c_int64, vp, vp, vp, C.c_int, vp, vp, vp, c_i64p]),
    "mxg_mv_create": (C.c_int, [vp, C.c_int64, C.c_int, C.c_int, C.POINTER(vp)]),
    "mxg_mv_view": (C.c_int, [vp, c_ip, C.c_int, C.POINTER(vp)]),
    "mxg_mv_clone": (C.c_int, [vp, c_ip, C.c_int, C.POINTER(vp)]),
    "mxg_mv_destroy": (C.c_int, [vp]),
    "mxg_mv_info": (C.c_int, [vp, c_i64p, c_ip, c_ip]),
    "mxg_mv_put": (C.c_int, [vp, C.c_int, C.c_int64, C.c_int64, vp]),
    "mxg_mv_fetch": (C.c_int, [vp, C.c_int, C.c_int64, C.c_int64, vp]),
    "mxg_mv_copy": (C.c_int, [vp, vp]),
    "mxg_mv_fill": (C.c_int, [vp, C.c_double, C.c_double]),
    "mxg_mv_random": (C.c_int, [vp, C.c_uint64]),
    "mxg_mv_conj": (C.c_int, [vp]),
    "mxg_mv_scale": (C.c_int, [vp, C.c_double, C.c_double]),
    "mxg_mv_scale_cols": (C.c_int, [vp, vp]),
    "mxg_mv_mask_rows": (C.c_int, [vp, vp]),
    "mxg_mv_axpby": (C.c_int, [C.c_double, C.c_double, vp, C.c_double, C.c_double, vp]),
    "mxg_mv_add": (C.c_int, [C.c_double, C.c_double, vp, C.c_double, C.c_double, vp, vp]),
    "mxg_mv_dot": (C.c_int, [vp, vp, vp]),
    "mxg_mv_norm2": (C.c_int, [vp, vp]),
    "mxg_mv_set_block": (C.c_int, [vp, vp, c_ip, C.c_int]),
    "mxg_mv_times_mat": (C.c_int, [C.c_double, C.c_double, vp, vp, C.c_int, C.c_int, C.c_int,
                                   C.c_double, C.c_double, vp]),
    "mxg_mv_trans_mv": (C.c_int, [C.c_double, C.c_double, vp, vp, vp, C.c_int]),
    "mxg_csr_product_form": (C.c_int, [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "mxg_mv_trans_mv2": (C.c_int, [C.c_double, C.c_double, vp, vp, vp, vp, C.c_int]),
    "mxg_mg_create": (C.c_int, [vp, C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.c_int, C.c_int, C.c_int,
                                C.c_double, C.c_double, C.POINTER(vp)]),
    "mxg_mg_destroy": (C.c_int, [vp]),
    "mxg_mg_apply": (C.c_int, [vp, vp, vp, vp]),
    "mxg_mg_set_tail": (C.c_int, [vp, vp, c_i64p]),
    "mxg_mg_set_chebyshev": (C.c_int, [vp, C.c_double]),
    "mxg_solve_cg": (C.c_int, [vp, vp, vp, vp, C.POINTER(SolveOpts), C.POINTER(SolveInfo)]),
    "mxg_solve_gmres": (C.c_int, [vp, vp, vp, vp, C.POINTER(SolveOpts), C.POINTER(SolveInfo)]),
    "mxg_solve_bicgstab": (C.c_int, [vp, vp, vp, vp, C.POINTER(SolveOpts), C.POINTER(SolveInfo)]),
}

MXH_SIGNATURES = {
    "mxh_last_error": (C.c_char_p, []),
    "mxh_sim_create": (C.c_int, [C.POINTER(C.c_int), c_dp, c_dp, C.POINTER(C.c_char_p),
                                 C.POINTER(C.c_char_p), C.c_char_p, C.c_int, C.POINTER(vp)]),
    "mxh_sim_destroy": (C.c_int, [vp]),
    "mxh_partition": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                C.POINTER(C.c_int)]),
    "mxh_owner_of_cell": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mxh_sim_create_box": (C.c_int, [C.POINTER(C.c_int), c_dp, c_dp, C.POINTER(C.c_char_p),
                                     C.POINTER(C.c_char_p), C.c_char_p, C.c_int, C.POINTER(C.c_int),
                                     C.POINTER(C.c_int), C.POINTER(vp)]),
    "mxh_sim_create_ex": (C.c_int, [C.POINTER(C.c_int), c_dp, c_dp, C.POINTER(C.c_char_p),
                                    C.POINTER(C.c_char_p), C.c_char_p, C.c_char_p, C.c_double, C.c_int,
                                    C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(vp)]),
    "mxh_dist_ops_create": (C.c_int, [vp, vp, C.POINTER(vp)]),
    "mxh_dist_ops_matrix": (C.c_int, [vp, C.c_char_p, C.c_double, C.POINTER(vp), c_i64p, C.POINTER(c_i64p), c_i64p,
                                      C.POINTER(c_i64p), c_i64p]),
    "mxh_dist_ops_destroy": (C.c_int, [vp]),
    "mxh_sim_device_chain": (C.c_int, [vp, vp]),
    "mxh_sim_dist_op": (C.c_int, [vp, C.c_char_p, C.c_double, c_i64p, c_i64p, C.POINTER(c_i64p),
                                  C.POINTER(c_i64p), C.POINTER(c_dp), C.POINTER(c_i64p), c_i64p,
                                  C.POINTER(c_i64p), c_ip]),
    "mxh_sim_op": (C.c_int, [vp, C.c_char_p, c_i64p, c_i64p, c_i64p, C.POINTER(c_i64p),
                             C.POINTER(c_i32p), C.POINTER(c_dp)]),
    "mxh_sim_shifted": (C.c_int, [vp, C.c_double, c_i64p, c_i64p, C.POINTER(c_i64p),
                                  C.POINTER(c_i32p), C.POINTER(c_dp)]),
    "mxh_sim_map": (C.c_int, [vp, C.c_char_p, c_i64p, C.POINTER(c_i64p)]),
    "mxh_sim_used_mask": (C.c_int, [vp, c_i64p, C.POINTER(c_dp)]),
    "mxh_sim_setup_seconds": (C.c_double, [vp]),
    "mxh_mg_create": (C.c_int, [vp, vp, C.c_char_p, C.c_double, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_double,
                                C.c_double, C.POINTER(vp)]),
    "mxh_mg_destroy": (C.c_int, [vp]),
    "mxh_mg_handle": (vp, [vp]),
    "mxh_mg_fine_matrix": (vp, [vp]),
    "mxh_mg_level_matrix": (vp, [vp, C.c_int, C.c_char]),
    "mxh_mg_level_sizes": (C.c_int, [vp, c_i64p, c_i64p, c_ip]),
    "mxh_solver_default_params": (None, [C.POINTER(SolverParamsC)]),
    "mxh_solver_create": (C.c_int, [vp, vp, C.POINTER(SolverParamsC), C.POINTER(vp)]),
    "mxh_solver_destroy": (C.c_int, [vp]),
    "mxh_solver_solve": (C.c_int, [vp, c_ip]),
    "mxh_solver_result": (C.c_int, [vp, C.c_int, c_dp, c_dp, c_dp, c_dp, c_dp]),
    "mxh_solver_stats": (C.c_int, [vp, c_ip, c_ip, c_ip, c_dp, c_dp, c_dp]),
    "mxh_solver_eigvec": (C.c_int, [vp, C.c_int, vp]),
    "mxh_hdf5_available": (C.c_int, []),
    "mxh_field_write": (C.c_int, [C.c_char_p, C.c_char_p, c_i64p, c_dp, C.c_char_p, C.c_size_t]),
    "mxh_field_append": (C.c_int, [C.c_char_p, C.c_char_p, c_i64p, c_dp, C.c_char_p, C.c_size_t]),
    "mxh_field_read_h5": (C.c_int, [C.c_char_p, C.c_char_p, c_i64p, c_dp, C.c_int64]),
    "mxh_problem_parse": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(vp)]),
    "mxh_problem_load": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(vp)]),
    "mxh_problem_destroy": (C.c_int, [vp]),
    "mxh_problem_param": (C.c_int, [vp, C.c_char_p, C.c_char_p, C.c_size_t]),
    "mxh_problem_grid": (C.c_int, [vp, C.POINTER(C.c_int), c_dp, c_dp, C.POINTER(C.c_int),
                                   C.POINTER(C.c_int), c_dp, c_ip]),
    "mxh_problem_pec_func": (C.c_int, [vp, c_dp, c_dp]),
    "mxh_problem_is_complex": (C.c_int, [vp]),
    "mxh_problem_op": (C.c_int, [vp, C.c_char_p, c_i64p, c_i64p, c_i64p, C.POINTER(c_i64p), C.POINTER(c_i32p),
                                 C.POINTER(c_dp), c_ip]),
    "mxh_problem_sim": (C.c_int, [vp, C.POINTER(vp)]),
    "mxh_problem_solver": (C.c_int, [vp, vp, C.c_int, C.POINTER(vp)]),
}

_cache = {}


def _load(name, signatures):
    if name in _cache:
        return _cache[name]
    path = os.path.join(_LIBDIR, name)
    if not os.path.exists(path):
        raise MxError("%s is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(there is no CPU fallback for the hot path)" % path)
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    for fn, (res, args) in signatures.items():
        f = getattr(lib, fn)   # AttributeError here = the library misses a declared symbol
        f.restype = res
        f.argtypes = args
    _cache[name] = lib
    return lib


def load_mxg():
    return _load("libmxg.so", MXG_SIGNATURES)


def load_mxhost():
    from . import _env
    _env.set_omp_threads()
    load_mxg()
    return _load("libmxhost.so", MXH_SIGNATURES)
