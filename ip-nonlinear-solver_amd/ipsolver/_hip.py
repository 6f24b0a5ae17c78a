"""ctypes binding of ``libipx.so`` (include/ipx.h) -- the only way this package
reaches the GPU.  There is no CPU fallback: if the library is missing, or no
HIP device is visible when a kernel is needed, the import / call fails loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libipx.so")

WS_DOUBLES = 65536          # IPX_WS_DOUBLES
VEC_GRID_CAP = 1024         # IPX_VEC_GRID_CAP (tests/test_abi.py checks it against ipx_reduce_grid)
SPMV_TILE_NNZ = 2048        # IPX_SPMV_TILE_NNZ

_c = ctypes
_P, _I64, _I32, _F64 = _c.c_void_p, _c.c_int64, _c.c_int32, _c.c_double

# name -> argtypes; every function returns int unless listed in _RESTYPES.
_SIGNATURES = {
    "ipx_device_info": [_c.POINTER(_c.c_int), _c.POINTER(_c.c_int), _c.c_char_p, _c.c_int],
    "ipx_read_doubles": [_P, _c.c_int, _P, _P],
    "ipx_read_folded": [_c.c_int, _P, _P, _P],
    "ipx_fold_combine": [_c.c_int, _P, _P, _P, _P],
    "ipx_reduce_grid": [_I64],
    "ipx_dot_partials": [_I64, _P, _P, _P, _P],
    "ipx_norms_partials": [_I64, _P, _P, _P],
    "ipx_axpby": [_I64, _F64, _P, _F64, _P, _P, _P],
    "ipx_mul": [_I64, _P, _P, _P, _P],
    "ipx_fill": [_I64, _F64, _P, _P],
    "ipx_clip": [_I64, _P, _P, _P, _P, _P],
    "ipx_affine": [_I64, _F64, _P, _F64, _P, _P],
    "ipx_gather": [_I64, _P, _P, _P, _P, _P, _P],
    "ipx_scatter": [_I64, _P, _P, _P, _P],
    "ipx_scatter_add": [_I64, _P, _P, _P, _P],
    "ipx_max_scalar": [_I64, _P, _F64, _P, _P],
    "ipx_where_positive": [_I64, _P, _P, _F64, _P, _P],
    "ipx_assign_negated_where": [_I64, _P, _P, _P, _P],
    "ipx_sum_log": [_I64, _P, _P, _P, _P],
    "ipx_dense_gemv": [_I64, _I64, _P, _I64, _P, _F64, _P, _F64, _P, _P, _P, _P, _P],
    "ipx_gram_f64_mfma": [_I64, _I64, _P, _I64, _P, _P],
    "ipx_gram_splits": [_I64, _I64],
    "ipx_gram_f64_mfma_split": [_I64, _I64, _P, _I64, _P, _P, _I32, _P],
    "ipx_aat_dense": [_I64, _P, _P, _P, _P, _P],
    "ipx_chol_factor": [_I64, _P, _P, _P, _P],
    "ipx_chol_inverse": [_I64, _P, _P, _P],
    "ipx_dot": [_I64, _P, _P, _P, _P, _P],
    "ipx_norms": [_I64, _P, _P, _P, _P],
    "ipx_box_inside": [_I64, _P, _P, _P, _P, _P, _P],
    "ipx_box_sphere_reduce": [_I64, _P, _P, _F64, _P, _P, _P, _P, _P],
    "ipx_csr_tiles_host": [_I64, _P, _I32, _I32, _P, _I64],
    "ipx_csr_spmv": [_I64, _I64, _P, _P, _P, _P, _I32, _P, _F64, _P, _F64, _P, _P,
                     _c.c_int, _P, _P, _P],
    "ipx_csr_spmv_ex": [_I64, _I64, _P, _P, _P, _P, _I32, _P, _F64, _P, _F64, _P, _P, _P, _P, _P, _P],
    "ipx_fold2": [_P, _I32, _P, _P, _P],
    "ipx_cg_step1": [_I64, _P, _I32, _P, _I32, _P, _P, _P, _P, _P, _P, _P, _I32, _P],
    "ipx_cg_step2": [_I64, _P, _I32, _I32, _P, _I32, _P, _I32, _P, _I32, _P, _P, _P, _I32, _P],
    "ipx_cg_state_size": [],
    "ipx_cg_vec_grid": [_I64],
    "ipx_cg_hp": [_P, _P],
    "ipx_cg_resume": [_P, _I32, _I32, _P],
    "ipx_cg_iterate": [_P, _I32, _I32, _P],
    "ipx_cg_resident_ok": [_P],
    "ipx_cg_prime_state": [_P, _P, _P, _F64, _F64, _F64, _F64, _F64, _P],
    "ipx_cg_prime": [_P, _P, _I32, _P, _P, _P, _P, _F64, _F64, _F64, _F64, _F64, _I32, _I32, _P],
    "ipx_cg_iterate_timed": [_P, _I32, _I32, _P, _P],
    "ipx_banded_kmax": [],
    "ipx_banded_levels": [_P],
    "ipx_banded_decoupled": [_P],
    "ipx_banded_pcr_level": [_P],
    "ipx_banded_refine_steps": [_P, _P],
    "ipx_banded_set_decoupling": [_P, _c.c_int],
    "ipx_banded_factor": [_P, _P, _P],
    "ipx_banded_status": [_P, _P],
    "ipx_banded_solve": [_P, _P, _P, _P],
    "ipx_banded_solve_multilaunch": [_P, _P, _P, _P],
    "ipx_banded_solve_guarded_c": [_P, _P, _P, _P, _P],
    "ipx_banded_solve_resid": [_P, _P, _P, _P, _P, _P, _P],
    "ipx_banded_decoupled_geometry": [_P, _P],
    "ipx_cg_step2_hp": [_P, _I32, _I32, _P],
    "ipx_pcg_state_size": [],
    "ipx_pcg_iterate": [_P, _I32, _I32, _P],
    "ipx_blockjacobi_build": [_I64, _P, _P, _P, _P, _P, _P, _P],
    "ipx_blockjacobi_apply": [_I64, _I64, _P, _P, _P, _P, _P, _P, _P],
    "ipx_cg_shard2_segment": [_P, _P, _I32, _I32, _I32, _P],
    "ipx_cg_shard2_fold_hp": [_P, _P, _P],
    "ipx_cg_shard2_iterate": [_P, _P, _I32, _I32, _P],
    "ipx_cg_shard2_fusable": [_P, _P],
    "ipx_cg_shard2_resident_ok": [_P, _P],
    "ipx_cg_shard2_resident": [_P, _P, _I32, _I32, _P],
    "ipx_cg_save_pb": [_P, _P],
    "ipx_peer_pingpong": [_P, _I32, _I32, _P, _P],
    "ipx_peer_attach_resident": [_P, _I64],
    "ipx_peer_export_resident": [_P, _P],
    "ipx_peer_import_resident": [_P, _I32, _P],
    "ipx_peer_resident_ready": [_P],
    "ipx_cg_resident_max_global": [],
    "ipx_peer_set_timeout": [_P, _F64],
    "ipx_peer_handle_bytes": [],
    "ipx_peer_export": [_P, _P],
    "ipx_peer_import": [_P, _I32, _P],
    "ipx_peer_ready": [_P],
    "ipx_peer_sequence": [_P, _P],
    "ipx_peer_allreduce": [_P, _I32, _P, _P, _P, _I32, _P],
    "ipx_peer_allgather": [_P, _I32, _P, _P, _P, _P],
    "ipx_peer_exchange": [_P, _P, _I32, _P, _P, _P],
    "ipx_aat_band": [_I64, _I32, _P, _P, _P, _P, _P, _P],
    "ipx_aat_band_w": [_I64, _I32, _P, _P, _P, _P, _P, _P, _P],
    "ipx_pairs_factor": [_I32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "ipx_pairs_tsolve": [_I32, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "ipx_pairs_vsolve": [_I32, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "ipx_boxschur_solve": [_P, _P, _P, _P, _P, _P, _P],
    "ipx_boxschur_project": [_P, _P, _P, _P, _P, _P, _P, _P, _P],
    "ipx_boxschur_project_count": [_P],
    "ipx_banded_status_deferred": [_P, _P, _P],
    "ipx_banded_refactor": [_P, _I64, _I32, _P, _P, _P, _P, _P, _P, _P],
    "ipx_sqp_block_size": [],
    "ipx_sqp_front": [_P, _c.c_int, _c.c_int, _c.c_int, _F64, _F64, _F64, _F64, _F64, _F64, _F64,
                      _F64, _I32, _P],
    "ipx_sqp_model": [_P, _F64, _F64, _F64, _c.c_int, _P],
    "ipx_sqp_judge": [_P, _P, _F64, _P, _P],
    "ipx_sqp_refresh": [_P, _P],
    "ipx_sqp_cg_timing": [_c.c_int, _P, _P],
}
_RESTYPES = {"ipx_version": _c.c_char_p, "ipx_last_error": _c.c_char_p,
             "ipx_launch_count": _c.c_longlong, "ipx_read_count": _c.c_longlong,
             "ipx_banded_create": _P, "ipx_banded_destroy": None,
             "ipx_dense_padded": _I64, "ipx_gram_ws_doubles": _I64, "ipx_peer_create": _P, "ipx_peer_destroy": None,
             "ipx_peer_halo_capacity": _I64, "ipx_peer_fused_launches": _I64,
             "ipx_cg_resident_ll_words": _I64, "ipx_cg_prime_ws_doubles": _I64,
             "ipx_peer_resident_launches": _I64, "ipx_cg_resident_limits": None,
             "ipx_sqp_part_doubles": _I64, "ipx_sqp_model_host": None, "ipx_sqp_ratio_host": None,
             "ipx_sqp_radius_host": None, "ipx_sqp_box_sphere_host": None}
_EXTRA_ARGTYPES = {"ipx_banded_create": [_I64, _I32, _I32], "ipx_banded_destroy": [_P],
                   "ipx_dense_padded": [_I64], "ipx_gram_ws_doubles": [_I64, _I32],
                   "ipx_peer_create": [_I32, _I32, _I64],
                   "ipx_peer_destroy": [_P], "ipx_peer_halo_capacity": [_P],
                   "ipx_peer_fused_launches": [_P], "ipx_cg_resident_ll_words": [_I32, _I32],
                   "ipx_cg_prime_ws_doubles": [_P, _I32], "ipx_peer_resident_launches": [_P],
                   "ipx_cg_resident_limits": [_P], "ipx_sqp_part_doubles": [_P],
                   "ipx_sqp_model_host": [_P], "ipx_sqp_ratio_host": [_P],
                   "ipx_sqp_radius_host": [_P], "ipx_sqp_box_sphere_host": [_P, _F64, _c.c_int, _P]}

_lib = None


class IpxError(RuntimeError):
    pass


_ERRORS = {-1: "invalid argument", -2: "HIP launch/runtime error",
           -3: "matrix is not positive definite", -4: "out of memory",
           -5: "no path in this solver for the matrix at hand",
           -6: "factorization complete but numerically rank deficient"}


def load():
    """Load libipx.so (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise IpxError(
            "libipx.so not found at %s: build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C ip-nonlinear-solver_amd/csrc`. This package has no CPU "
            "fallback." % LIB_PATH)
    # torch bundles its own HIP runtime under the same SONAME as /opt/rocm's;
    # import it first so libipx.so binds to the runtime torch's allocator and
    # streams live in (two runtimes in one process see no device).
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, args in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is absent
        fn.argtypes = args
        fn.restype = _c.c_int
    for name, res in _RESTYPES.items():
        getattr(lib, name).restype = res
    for name, args in _EXTRA_ARGTYPES.items():
        getattr(lib, name).argtypes = args
    _lib = lib
    return lib


# ---- the ONE debug switch of the library -------------------------------------------------
# IPX_DEBUG_FORMS = comma-separated list of kernel forms to turn OFF, for A/B measurements and
# for the parity tests that compare each form with the plain one (tests/test_gpu_qp.py):
#   no-fuse            step1 / step2 / g = r - A'v as launches of their own (no fused SpMVs)
#   no-resident        small problems on the separate launches (no csrc/resident.hip)
#   no-compact-groups  box-Schur group tables in their general form (44 B per variable)
#   no-affine-groups   ... with the column table read instead of computed
#   keep-xn2           ||x + alpha p||^2 formed even for an infinite trust radius
#   pack-comm          sharded loop: the collectives in pack kernels of their own (5 launches)
#   no-post-tail       barrier problems: the per-item back substitution as a launch of its own
#                      (k_pairs_post) instead of as the tail of the Schur solve's kernel
# Anything else that selects behaviour is an argument (``options={'shard': True}``) or a
# deployment setting (IPX_SHARD, IPX_SHARD_TRANSPORT=dist).
# (round 5 removed "no-c16" and "no-diag-merge": their A/Bs are settled -- profiles/r03*, r04* --
# and the 16-bit index tables / the merged diagonal are what every qualifying pattern gets)
#   no-step-chain      the outer iteration's stages host-driven (sqp.HostStages) instead of as
#                      device chains (csrc/sqp.hip)
DEBUG_FORMS = ("no-fuse", "no-resident", "no-compact-groups", "no-affine-groups", "keep-xn2",
               "pack-comm", "no-post-tail", "no-step-chain")


def debug_form(name):
    assert name in DEBUG_FORMS, name
    val = os.environ.get("IPX_DEBUG_FORMS", "")
    if not val:
        return False
    toks = [t.strip() for t in val.split(",") if t.strip()]
    for t in toks:
        if t not in DEBUG_FORMS:
            raise IpxError("IPX_DEBUG_FORMS: unknown form %r (known: %s)" % (t, ", ".join(DEBUG_FORMS)))
    return name in toks


def exported_symbols():
    return sorted(list(_SIGNATURES) + list(_RESTYPES))


def check(code, what=""):
    if code < 0:
        detail = load().ipx_last_error().decode() if code == -2 else ""
        raise IpxError("%s failed: %s (code %d) %s" % (what or "ipx call",
                                                       _ERRORS.get(code, "?"), code, detail))
    return code


_BOUND = {}


def call(name, *args):
    """Call an int-returning entry point and raise on a negative status.  (The bound function is
    looked up once per name: ``getattr`` on the CDLL and two Python frames per launch were a
    third of its host cost.)"""
    fn = _BOUND.get(name)
    if fn is None:
        fn = _BOUND[name] = getattr(load(), name)
    rc = fn(*args)
    return rc if rc >= 0 else check(rc, name)
