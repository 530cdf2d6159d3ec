"""ctypes binding of libcup3d_hip.so — the C ABI declared in include/cup3d_hip.h.

There is NO CPU fallback: every compute entry point runs hand-written HIP kernels on
an MI355X (gfx950) and raises Cup3dError when the library or a GPU is missing.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# Two flavours are built from the same sources (csrc/Makefile): the release library, and libcup3d_hip_testing.so (-DCUP3D_TESTING) with
# the debug-option map, the in-process "virtual" communicator and the A/B kernel variants.  The test-suite and the tuning scans ask
# for the second one (tests/conftest.py sets CUP3D_HIP_FLAVOUR=testing); bench.py and __graft_entry__.smoke() run the release build.
FLAVOUR = os.environ.get("CUP3D_HIP_FLAVOUR", "release")
if FLAVOUR not in ("release", "testing"):
    raise ImportError(f"CUP3D_HIP_FLAVOUR={FLAVOUR!r}: expected 'release' or 'testing'")
LIB_PATH = os.path.join(HERE, "libcup3d_hip.so" if FLAVOUR == "release" else "libcup3d_hip_testing.so")

FIELD_CHI, FIELD_PRES, FIELD_VEL, FIELD_TMPV, FIELD_LHS = 0, 1, 2, 3, 4
FIELD_NCOMP = {FIELD_CHI: 1, FIELD_PRES: 1, FIELD_VEL: 3, FIELD_TMPV: 3, FIELD_LHS: 1}
BC = {"freespace": 0, "periodic": 1, "wall": 2}
NBR_HALO = 0x40000000
NBR_COARSER = 0x20000000


class Cup3dError(RuntimeError):
    pass


class PoissonParams(C.Structure):
    _fields_ = [("tol", C.c_double), ("tol_rel", C.c_double), ("mean_constraint", C.c_int),
                ("max_iter", C.c_int), ("max_restarts", C.c_int), ("block_solver", C.c_int)]


class PoissonResult(C.Structure):
    _fields_ = [("iterations", C.c_int), ("restarts", C.c_int), ("norm0", C.c_double), ("norm", C.c_double),
                ("used_xopt", C.c_int)]


class RunStats(C.Structure):
    """cup3d_run_stats"""
    _fields_ = [("halo_exchanges", C.c_long), ("halo_bytes_sent", C.c_double), ("allreduces", C.c_long), ("host_waits", C.c_long),
                ("host_wait_seconds", C.c_double), ("solver_iterations", C.c_long), ("field_bytes_uploaded", C.c_double), ("field_bytes_downloaded", C.c_double)]


class ProfileEntry(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_long), ("total_ms", C.c_double)]


_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_lp = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_vp = C.c_void_p

class Obstacle(C.Structure):
    """cup3d_obstacle"""
    _fields_ = [("nblocks", C.c_long), ("slots", C.c_void_p), ("chi", C.c_void_p), ("udef", C.c_void_p),
                ("cm", C.c_double * 3), ("vel", C.c_double * 3), ("omega", C.c_double * 3),
                ("force", C.c_double * 3), ("torque", C.c_double * 3)]


# name -> (restype, argtypes); must list every symbol of include/cup3d_hip.h
SIGNATURES = {
    "cup3d_last_error": (C.c_char_p, []),
    "cup3d_version": (C.c_char_p, []),
    "cup3d_sfc_create": (C.c_int, [C.c_int] * 4 + [C.POINTER(_vp)]),
    "cup3d_sfc_destroy": (None, [_vp]),
    "cup3d_sfc_forward": (C.c_longlong, [_vp] + [C.c_int] * 4),
    "cup3d_sfc_inverse": (None, [_vp, C.c_longlong, C.c_int, _ip]),
    "cup3d_sfc_encode": (C.c_longlong, [_vp, C.c_int, _ip]),
    "cup3d_sfc_info": (None, [_vp, C.c_int, _ip, _lp, _lp, _lp]),
    "cup3d_grid_create_uniform": (C.c_int, [_ip, C.c_int, C.c_int, C.c_double, _ip, C.c_int, C.c_int, C.POINTER(_vp)]),
    "cup3d_grid_create_mesh": (C.c_int, [_ip, C.c_int, C.c_double, _ip, C.c_long, _ip, _lp, C.POINTER(_vp)]),
    "cup3d_grid_ninterface_faces": (C.c_long, [_vp]),
    "cup3d_grid_interface": (C.c_int, [_vp, _vp, _vp, _vp]),
    "cup3d_grid_valid_states": (C.c_int, [_vp, np.ctypeslib.ndpointer(dtype=np.int8, flags="C_CONTIGUOUS")]),
    "cup3d_grid_adapted": (C.c_int, [_vp, np.ctypeslib.ndpointer(dtype=np.int8, flags="C_CONTIGUOUS"), C.POINTER(_vp)]),
    "cup3d_adapt_transfer": (C.c_int, [_vp, _vp, C.c_int]),
    "cup3d_adapt_migrate": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int]),
    "cup3d_grid_destroy": (None, [_vp]),
    "cup3d_grid_nblocks": (C.c_long, [_vp]),
    "cup3d_grid_nblocks_global": (C.c_long, [_vp]),
    "cup3d_grid_nhalo_faces": (C.c_long, [_vp]),
    "cup3d_grid_nsend_faces": (C.c_long, [_vp]),
    "cup3d_grid_ninner": (C.c_long, [_vp]),
    "cup3d_grid_tables": (C.c_int, [_vp, _lp, _dp]),
    "cup3d_grid_neighbours": (C.c_int, [_vp, _ip]),
    "cup3d_grid_halo_plan": (C.c_int, [_vp, _lp, _lp, _vp]),
    "cup3d_calc_max_timestep": (C.c_double, [C.c_double] * 4 + [C.c_int, C.c_int, C.c_double, _dp]),
    "cup3d_calc_max_timestep2": (C.c_double, [C.c_double] * 4 + [C.c_int, C.c_int, C.c_double, _dp, C.c_int]),
    "cup3d_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "cup3d_device_init": (C.c_int, [C.c_int]),
    "cup3d_set_stream": (C.c_int, [_vp]),
    "cup3d_device_synchronize": (C.c_int, []),
    "cup3d_comm_unique_id": (C.c_int, [_vp]),
    "cup3d_comm_init": (C.c_int, [C.c_int, C.c_int, _vp]),
    "cup3d_comm_finalize": (C.c_int, []),
    "cup3d_sim_create": (C.c_int, [_vp, C.POINTER(_vp)]),
    "cup3d_sim_destroy": (None, [_vp]),
    "cup3d_sim_device_bytes": (C.c_size_t, [_vp]),
    "cup3d_sim_upload_blocks": (C.c_int, [_vp, C.c_int, C.POINTER(_vp)]),
    "cup3d_sim_download_blocks": (C.c_int, [_vp, C.c_int, C.POINTER(_vp)]),
    "cup3d_sim_upload": (C.c_int, [_vp, C.c_int, _dp]),
    "cup3d_sim_download": (C.c_int, [_vp, C.c_int, _dp]),
    "cup3d_sim_fill": (C.c_int, [_vp, C.c_int, C.c_double]),
    "cup3d_sim_device_ptr": (C.c_int, [_vp, C.c_int, C.POINTER(_vp)]),
    "cup3d_sim_mark_written": (C.c_int, [_vp, C.c_int]),
    "cup3d_sim_set_obstacles": (C.c_int, [_vp, C.c_int]),
    "cup3d_sim_checksum": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_ulonglong)]),
    "cup3d_stats_reset": (C.c_int, []),
    "cup3d_stats_read": (C.c_int, [C.POINTER(RunStats)]),
    "cup3d_advect_diffuse": (C.c_int, [_vp, C.c_double, C.c_double, _dp]),
    "cup3d_max_u": (C.c_int, [_vp, _dp, C.POINTER(C.c_double)]),
    "cup3d_external_forcing": (C.c_int, [_vp] + [C.c_double] * 4),
    "cup3d_poisson_default_params": (None, [C.POINTER(PoissonParams)]),
    "cup3d_compute_lhs": (C.c_int, [_vp, C.c_int]),
    "cup3d_preconditioner": (C.c_int, [_vp, C.c_int]),
    "cup3d_poisson_solve": (C.c_int, [_vp, C.POINTER(PoissonParams), C.POINTER(PoissonResult)]),
    "cup3d_pressure_rhs": (C.c_int, [_vp, C.c_double]),
    "cup3d_div_pressure": (C.c_int, [_vp]),
    "cup3d_grad_p": (C.c_int, [_vp, C.c_double]),
    "cup3d_pressure_project": (C.c_int, [_vp, C.c_double, C.c_int, C.POINTER(PoissonParams), C.POINTER(PoissonResult)]),
    "cup3d_restrict": (C.c_int, [_vp, _vp, C.c_int]),
    "cup3d_prolong": (C.c_int, [_vp, _vp, C.c_int]),
    "cup3d_tag_blocks": (C.c_int, [_vp, C.c_int, C.c_double, C.c_double, np.ctypeslib.ndpointer(dtype=np.int8, flags="C_CONTIGUOUS")]),
    "cup3d_compute_vorticity": (C.c_int, [_vp]),
    "cup3d_grad_chi_on_tmp": (C.c_int, [_vp, C.c_double, C.c_double, C.c_int]),
    "cup3d_grad_chi_on_tmp_over_ranks": (C.c_int, [_vp, _vp, _vp, C.c_double, C.c_double, C.c_int]),
    "cup3d_grid_adapted_owners": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, _vp]),
    "cup3d_grid_rank_view": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "cup3d_grid_view_sizes": (C.c_int, [_vp, _vp]),
    "cup3d_grid_view_plan": (C.c_int, [_vp] * 9),
    "cup3d_grid_view_boxes": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp]),
    "cup3d_sim_upload_block_list": (C.c_int, [_vp, C.c_int, C.c_long, _vp, _vp]),
    "cup3d_sim_download_block_list": (C.c_int, [_vp, C.c_int, C.c_long, _vp, _vp]),
    "cup3d_advect_diffuse_implicit": (C.c_int, [_vp, C.c_double, C.c_double, _dp, C.POINTER(PoissonParams), C.POINTER(PoissonResult)]),
    "cup3d_advect_implicit": (C.c_int, [_vp, C.c_double, C.c_double, _dp]),
    "cup3d_diffusion_rhs": (C.c_int, [_vp]),
    "cup3d_diffusion_lhs": (C.c_int, [_vp, C.c_int, C.c_double, C.c_double]),
    "cup3d_diffusion_preconditioner": (C.c_int, [_vp, C.c_double, C.c_double]),
    "cup3d_diffusion_solve": (C.c_int, [_vp, C.c_int, C.c_double, C.c_double, C.POINTER(PoissonParams), C.POINTER(PoissonResult)]),
    "cup3d_penalization": (C.c_int, [_vp, C.c_double, C.c_double, C.c_int, C.c_int, C.POINTER(Obstacle)]),
    "cup3d_update_tmpv": (C.c_int, [_vp, C.c_int, C.POINTER(Obstacle)]),
    "cup3d_profile_enable": (C.c_int, [C.c_int]),
    "cup3d_profile_reset": (C.c_int, []),
    "cup3d_profile_read": (C.c_int, [C.POINTER(ProfileEntry), C.c_int, C.POINTER(C.c_int)]),
    "cup3d_poisson_path_checksum": (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(C.c_ulonglong)]),
    "cup3d_profile_block_cg_iterations": (C.c_int, [_vp, C.POINTER(C.c_long), C.POINTER(C.c_long)]),
}
# test-support symbols (not part of the drop-in surface)
DEBUG_SIGNATURES = {
    "cup3d_debug_virtual_ranks": (C.c_int, [C.c_int]),
    "cup3d_debug_halo_pull": (C.c_int, [_vp, C.POINTER(_vp), C.c_int, C.c_int, C.c_int, C.c_int]),
    "cup3d_debug_advdiff_stage": (C.c_int, [_vp, C.c_int, C.c_double, C.c_double, _dp]),
    "cup3d_debug_set_option": (C.c_int, [C.c_char_p, C.c_int]),
    "cup3d_debug_amr_slabs": (C.c_int, [_vp, C.c_int, C.c_int, _dp]),
    "cup3d_debug_virtual_comm": (C.c_int, [C.c_int]),
    "cup3d_debug_host_transport": (C.c_int, [C.c_int, C.c_int, _vp]),
    "cup3d_debug_wave_sum": (C.c_int, [_dp, _dp]),
    "cup3d_debug_ctl_step": (C.c_int, [C.c_int, _dp, _dp]),
    "cup3d_debug_mg_plan_check": (C.c_int, [_vp, _vp, C.c_int]),
    "cup3d_debug_grid_inner_blocks": (C.c_int, [_vp, _vp]),
    "cup3d_debug_grid_rank_view_tensorial": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
}

_lib = None


def lib():
    """Load libcup3d_hip.so (built by __graft_entry__.build() / make -C cup3d_amd/csrc)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Cup3dError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        # the test-support names exist in libcup3d_hip_testing.so only; the release library does not export them
        for table in (SIGNATURES, DEBUG_SIGNATURES) if os.path.basename(LIB_PATH).endswith("_testing.so") else (SIGNATURES,):
            for name, (res, args) in table.items():
                fn = getattr(L, name)
                fn.restype = res
                fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise Cup3dError(f"cup3d error {rc}: {lib().cup3d_last_error().decode()}")


_device_ready = False


def device_init(device=0):
    global _device_ready
    check(lib().cup3d_device_init(device))
    _device_ready = True


def device_count():
    n = C.c_int(0)
    rc = lib().cup3d_device_count(C.byref(n))
    return n.value if rc == 0 else 0
