"""ctypes binding of libuammd_hip.so — the C ABI declared in include/uammd_hip.h.

There is deliberately NO fallback: if the HIP library is missing or a call fails, an exception is
raised (UammdHipError carries uammd_hip_last_error()).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UAMMD_HIP_LIB") or os.path.join(_HERE, "lib", "libuammd_hip.so")  # (override: A/B timing of two builds)


class UammdHipError(RuntimeError):
    pass


class LJPairParameters(C.Structure):
    _fields_ = [("cutOff2", C.c_float), ("sigma2", C.c_float), ("epsilonDivSigma2", C.c_float), ("shift", C.c_float)]


class CellListData(C.Structure):
    _fields_ = [("d_cellStart", C.c_void_p), ("d_cellEnd", C.c_void_p), ("d_sortPos", C.c_void_p),
                ("d_groupIndex", C.c_void_p), ("d_sortHash", C.c_void_p), ("cellDim", C.c_int * 3),
                ("boxSize", C.c_float * 3), ("periodic", C.c_int * 3), ("VALID_CELL", C.c_uint),
                ("numberParticles", C.c_int)]


class VerletListData(C.Structure):
    _fields_ = [("d_neighbourList", C.c_void_p), ("d_numberNeighbours", C.c_void_p), ("d_sortPos", C.c_void_p),
                ("d_groupIndex", C.c_void_p), ("particleStride", C.c_int), ("maxNeighboursPerParticle", C.c_int),
                ("numberParticles", C.c_int)]


class IBMKernel(C.Structure):
    _fields_ = [("kind", C.c_int), ("support", C.c_int * 3), ("prefactor", C.c_float), ("tau", C.c_float),
                ("rmax", C.c_float), ("invh", C.c_float * 3)]


class FCMParameters(C.Structure):
    _fields_ = [("boxSize", C.c_float * 3), ("cells", C.c_int * 3), ("viscosity", C.c_float), ("seed", C.c_uint),
                ("kernel", IBMKernel), ("hydrodynamicRadius", C.c_float)]


class ICMParameters(C.Structure):
    _fields_ = [("boxSize", C.c_float * 3), ("temperature", C.c_float), ("viscosity", C.c_float), ("density", C.c_float),
                ("hydrodynamicRadius", C.c_float), ("dt", C.c_float), ("cells", C.c_int * 3), ("sumThermalDrift", C.c_int),
                ("removeTotalMomentum", C.c_int), ("seed", C.c_uint)]


class FIBParameters(C.Structure):
    _fields_ = [("boxSize", C.c_float * 3), ("temperature", C.c_float), ("viscosity", C.c_float), ("hydrodynamicRadius", C.c_float),
                ("dt", C.c_float), ("cells", C.c_int * 3), ("scheme", C.c_int), ("seed", C.c_uint)]


class BDHI2DParameters(C.Structure):
    _fields_ = [("boxSize", C.c_float * 2), ("hydrodynamicRadius", C.c_float), ("viscosity", C.c_float), ("temperature", C.c_float),
                ("dt", C.c_float), ("cells", C.c_int * 2), ("seed", C.c_uint), ("kernel", C.c_int)]


class BDHI2DParametersF64(C.Structure):
    _fields_ = [("boxSize", C.c_double * 2), ("hydrodynamicRadius", C.c_double), ("viscosity", C.c_double), ("temperature", C.c_double),
                ("dt", C.c_double), ("cells", C.c_int * 2), ("seed", C.c_uint), ("kernel", C.c_int)]


class PoissonParameters(C.Structure):
    _fields_ = [("boxSize", C.c_float * 3), ("epsilon", C.c_float), ("tolerance", C.c_float), ("gw", C.c_float),
                ("split", C.c_float), ("upsampling", C.c_float)]


class PoissonParametersF64(C.Structure):
    _fields_ = [("boxSize", C.c_double * 3), ("epsilon", C.c_double), ("tolerance", C.c_double), ("gw", C.c_double),
                ("split", C.c_double), ("upsampling", C.c_double)]


class PoissonInfoF64(C.Structure):
    _fields_ = [("cells", C.c_int * 3), ("support", C.c_int), ("nearFieldCutOff", C.c_double), ("nTable", C.c_int), ("h", C.c_double)]


class PoissonInfo(C.Structure):
    _fields_ = [("cells", C.c_int * 3), ("support", C.c_int), ("nearFieldCutOff", C.c_float), ("nTable", C.c_int),
                ("h", C.c_float)]


class IBMKernel64(C.Structure):
    _fields_ = [("kind", C.c_int), ("support", C.c_int * 3), ("prefactor", C.c_double), ("tau", C.c_double), ("rmax", C.c_double),
                ("invh", C.c_double * 3)]


class FCMParameters64(C.Structure):
    _fields_ = [("boxSize", C.c_double * 3), ("cells", C.c_int * 3), ("viscosity", C.c_double), ("kernel", IBMKernel64)]


_f3 = C.c_float * 3
_i3 = C.c_int * 3
_vp = C.c_void_p
_f = C.c_float
_i = C.c_int
_u = C.c_uint

# name -> (restype, argtypes).  Every symbol include/uammd_hip.h declares must be listed here:
# tests/test_abi.py checks header, library and this table against each other.
_d = C.c_double
_d3 = C.c_double * 3
MATVEC64 = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p)

SIGNATURES = {
    "uammd_hip_abi_version": (_i, []),
    # DOUBLE_PRECISION build (f64.hip, lanczos.hip)
    "uammd_fcm_gaussian_kernel_f64": (_i, [_d, _d, C.POINTER(IBMKernel64), C.POINTER(_d)]),
    "uammd_fcm_advise_grid_size_f64": (_d, [_d, _d]),
    "uammd_ibm_spread_f64": (_i, [_vp, _i, _vp, _i, _i, _d3, _i3, _i3, _i, C.POINTER(IBMKernel64), _vp, _vp]),
    "uammd_ibm_gather_f64": (_i, [_vp, _i, _vp, _i, _i, _d3, _i3, _i3, _i, C.POINTER(IBMKernel64), _vp, _vp]),
    "uammd_fcm_create_f64": (_i, [C.POINTER(FCMParameters64), C.POINTER(_vp)]),
    "uammd_fcm_destroy_f64": (_i, [_vp]),
    "uammd_fcm_displacements_f64": (_i, [_vp, _vp, _vp, _i, _vp, _vp]),
    "uammd_pse_far_raw_cells_f64": (_i, [_d3, _d, _d, _i3]),
    "uammd_pse_far_create_f64": (_i, [_d3, _i3, _d, _d, _d, _d, _d, C.POINTER(_vp), C.POINTER(_i), C.POINTER(_d)]),
    "uammd_pse_near_create_f64": (_i, [_d3, _d, _d, _d, _d, C.POINTER(_vp), C.POINTER(_d), C.POINTER(_i)]),
    "uammd_pse_near_destroy_f64": (_i, [_vp]),
    "uammd_pse_near_mdot_f64": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    "uammd_lanczos_create_f64": (_i, [C.POINTER(_vp)]),
    "uammd_lanczos_destroy_f64": (_i, [_vp]),
    "uammd_lanczos_run_f64": (_i, [_vp, MATVEC64, _vp, _vp, _vp, _d, _i, _vp, C.POINTER(_i)]),
    "uammd_lanczos_run_iterations_f64": (_i, [_vp, MATVEC64, _vp, _vp, _vp, _i, _i, _vp, C.POINTER(_d)]),
    "uammd_fcm_displacements_thermal_f64": (_i, [_vp, _vp, _vp, _i, _d, _d, _u, _u, _vp, _vp]),
    "uammd_pse_near_stochastic_f64": (_i, [_vp, _vp, _vp, _i, _d, _d, _u, _u, _d, _vp, _vp, C.POINTER(_i)]),
    "uammd_convert_f64_to_f32": (_i, [_vp, _vp, C.c_size_t, _vp]),
    "uammd_bdhi_euler_maruyama_f64": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _d, _d, _i, _vp]),
    "uammd_lanczos_set_iteration_hard_limit_f64": (_i, [_vp, _i]),
    "uammd_lanczos_get_last_run_required_steps_f64": (_i, [_vp, C.POINTER(_i)]),
    "uammd_hip_last_error": (C.c_char_p, []),
    "uammd_hip_device_count": (_i, [C.POINTER(_i)]),
    "uammd_hip_set_device": (_i, [_i]),
    "uammd_hip_set_tunable": (_i, [C.c_char_p, _i]),
    "uammd_celllist_create": (_i, [C.POINTER(_vp)]),
    "uammd_celllist_destroy": (_i, [_vp]),
    "uammd_celllist_create_grid": (_i, [_f3, _i3, _f3, _i3, _f3, _i3]),
    "uammd_celllist_update": (_i, [_vp, _vp, _i, _f3, _i3, _i3, _vp]),
    "uammd_celllist_get": (_i, [_vp, C.POINTER(CellListData)]),
    "uammd_celllist_check_errors": (_i, [_vp, _vp]),
    "uammd_comm_unique_id": (_i, [C.c_char_p]),
    "uammd_comm_rccl_version": (_i, [C.POINTER(C.c_int)]),
    "uammd_comm_init": (_i, [C.POINTER(C.c_void_p), _i, _i, C.c_char_p]),
    "uammd_comm_destroy": (_i, [_vp]),
    "uammd_comm_rank": (_i, [_vp]),
    "uammd_comm_world": (_i, [_vp]),
    "uammd_comm_halo_exchange": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _vp]),
    "uammd_comm_exchange_counts": (_i, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), _vp]),
    "uammd_comm_exchange_counts_device": (_i, [_vp, _vp, C.POINTER(C.c_int), _vp]),
    "uammd_comm_alltoall": (_i, [_vp, _vp, _vp, C.c_size_t, _vp]),
    "uammd_comm_allreduce_sum": (_i, [_vp, _vp, _i, _vp]),
    "uammd_halo_pack": (_i, [_vp, _vp, _i, _vp, _i, _f, _f, _vp, _vp, _vp]),
    "uammd_lj_tile_stats": (_i, [_vp, _i, C.POINTER(C.c_uint), _vp]),
    "uammd_lj_profile_enable": (_i, [_vp, _i]),
    "uammd_lj_profile_read": (_i, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
    "uammd_celllist_set_option": (_i, [_vp, C.c_char_p, _i]),
    "uammd_sort_pairs": (_i, [_vp, _vp, _i, _i, _vp]),
    "uammd_gather": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "uammd_scatter": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "uammd_lj_table_changed": (_i, []),
    "uammd_slab_refresh_lj": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "uammd_halo_pack_gj1": (_i, [_vp, _vp, _vp, _vp, _f, _vp, _vp, _i, _vp, _i, _f, _f, _vp, _vp, _f, _f, _i, _f, C.c_uint, C.c_uint, _vp]),
    "uammd_celllist_update_gj1": (_i, [_vp, _vp, _i, _f3, _i3, _i3, _vp, _vp, _vp, _f, _vp, _vp, _i, _f, _f, _i, _f, C.c_uint, C.c_uint,
                                      _vp]),
    "uammd_lj_process_pair_parameters": (_i, [_f, _f, _f, _i, C.POINTER(LJPairParameters)]),
    "uammd_lj_transverse_celllist": (_i, [_vp, _vp, _i, _f3, _i3, _vp, _vp, _vp, _vp, _i, _vp]),
    "uammd_lj_transverse_celllist_gj2": (_i, [_vp, _vp, _i, _f3, _i3, _vp, _vp, _vp, _f, _f, _i, _i, _vp]),
    "uammd_lj_transverse_nbody": (_i, [_vp, _i, _vp, _i, _f3, _i3, _vp, _vp, _vp, _vp, _vp]),
    "uammd_verletlist_create": (_i, [C.POINTER(_vp)]),
    "uammd_verletlist_destroy": (_i, [_vp]),
    "uammd_verletlist_update": (_i, [_vp, _vp, _i, _f3, _i3, _f, _vp, C.POINTER(_i)]),
    "uammd_verletlist_force_next_update": (_i, [_vp]),
    "uammd_verletlist_set_cutoff_multiplier": (_i, [_vp, _f]),
    "uammd_slab_select_workspace": (_i, [_i, C.POINTER(C.c_size_t)]),
    "uammd_slab_select": (_i, [_vp, _i, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "uammd_slab_pack_rows": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _i, _f, _f, _vp, _vp, _vp]),
    "uammd_slab_unpack_rows": (_i, [_vp, _vp, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp]),
    "uammd_slab_max_displacement": (_i, [_vp, _vp, _i, _vp, _vp]),
    "uammd_slab_add2": (_i, [_vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "uammd_slab_copy2": (_i, [_vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "uammd_verletlist_get_steps_since_last_update": (_i, [_vp, C.POINTER(_i)]),
    "uammd_verletlist_get": (_i, [_vp, C.POINTER(VerletListData)]),
    "uammd_lj_transverse_verletlist": (_i, [_vp, _vp, _i, _f3, _i3, _vp, _vp, _vp, _vp, _vp]),
    "uammd_verletnvt_gj": (_i, [_i, _vp, _vp, _vp, _vp, _f, _vp, _i, _f, _f, _i, _f, _u, _u, _vp]),
    "uammd_verletnvt_gj_keyed": (_i, [_i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _i, _f, _f, _i, _f, _u, _u, _vp]),
    "uammd_verletnvt_gj_lj_step": (_i, [_vp, _vp, _vp, _vp, _vp, _f, _i, _f3, _i3, _f3, _i3, _i3, _vp, _i, _f, _f, _i, _f, _u, _u, _i, _vp]),
    "uammd_verletnvt_basic": (_i, [_i, _vp, _vp, _vp, _vp, _f, _vp, _i, _f, _f, _i, _f, _u, _u, _vp]),
    "uammd_verletnvt_initial_velocities": (_i, [_vp, _vp, _f, _i, _i, _u, _vp]),
    "uammd_sum_kinetic_energy": (_i, [_vp, _vp, _vp, _f, _vp, _i, _vp]),
    "uammd_bd_euler_maruyama": (_i, [_vp, _vp, _vp, C.POINTER(_f), _f, _vp, _f, _i, _f, _i, _u, _u, _vp]),
    "uammd_fcm_euler_maruyama": (_i, [_vp, _vp, _vp, _i, _f, _vp]),
    "uammd_bd_scheme_step": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _f, _i, _f, _i, _u, _u, _vp]),
    "uammd_bd_scheme_step_f64": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _d, _vp, _d, _i, _d, _i, _u, _u, _vp]),
    "uammd_bdhi_euler_maruyama": (_i, [_vp, _vp, _vp, _vp, C.POINTER(_f), _i, _f, _f, _i, _vp]),
    "uammd_fill_zero": (_i, [_vp, C.c_size_t, _vp]),
    "uammd_fill_zero_indexed": (_i, [_vp, _vp, _i, _i, _vp]),
    "uammd_fcm_gaussian_kernel": (_i, [_f, _f, C.POINTER(IBMKernel), C.POINTER(_f)]),
    "uammd_fcm_advise_grid_size": (_f, [_f, _f]),
    "uammd_ibm_barnett_magland_kernel": (_i, [_f, _f, _i, _f, C.POINTER(IBMKernel)]),
    "uammd_ibm_spread": (_i, [_vp, _i, _vp, _i, _i, _f3, _i3, _i3, _i, C.POINTER(IBMKernel), _vp, _vp]),
    "uammd_ibm_gather": (_i, [_vp, _i, _vp, _i, _i, _f3, _i3, _i3, _i, C.POINTER(IBMKernel), _vp, _vp]),
    "uammd_bdhi_cholesky_create": (_i, [_i, _f, _f, C.POINTER(_vp)]),
    "uammd_bdhi_cholesky_destroy": (_i, [_vp]),
    "uammd_bdhi_cholesky_setup_step": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "uammd_bdhi_cholesky_mf": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "uammd_bdhi_cholesky_bdw": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "uammd_icm_create": (_i, [C.POINTER(ICMParameters), C.POINTER(_vp), C.POINTER(_i3), C.POINTER(_f)]),
    "uammd_icm_destroy": (_i, [_vp]),
    "uammd_icm_predictor": (_i, [_vp, _vp, _i, _vp]),
    "uammd_icm_fluid_and_corrector": (_i, [_vp, _vp, _vp, _i, _vp]),
    "uammd_icm_get_fluid_velocity": (_i, [_vp, _vp, _i, _vp]),
    "uammd_icm_set_fluid_velocity": (_i, [_vp, _vp, _vp]),
    "uammd_icm_set_noise": (_i, [_vp, _vp]),
    "uammd_fib_create": (_i, [C.POINTER(FIBParameters), C.POINTER(_vp), C.POINTER(_i3), C.POINTER(_f)]),
    "uammd_fib_destroy": (_i, [_vp]),
    "uammd_fib_forward": (_i, [_vp, _vp, _vp, _i, _vp]),
    "uammd_fib_set_noise": (_i, [_vp, _vp]),
    "uammd_fib_self_mobility": (_f, [_f, _f, _f]),
    "uammd_bdhi2d_create": (_i, [C.POINTER(BDHI2DParameters), C.POINTER(_vp), C.POINTER(C.c_int * 2), C.POINTER(_i)]),
    "uammd_bdhi2d_destroy": (_i, [_vp]),
    "uammd_bdhi2d_velocities": (_i, [_vp, _vp, _vp, _i, _vp, _vp]),
    "uammd_bdhi2d_update_positions": (_i, [_vp, _vp, _i, _f, _vp]),
    "uammd_bdhi2d_get_counter": (_i, [_vp, C.POINTER(_u)]),
    "uammd_poisson_create_f64": (_i, [C.POINTER(PoissonParametersF64), C.POINTER(_vp), C.POINTER(PoissonInfoF64)]),
    "uammd_poisson_destroy_f64": (_i, [_vp]),
    "uammd_poisson_sum_f64": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _i, _vp]),
    "uammd_poisson_field_potential_f64": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "uammd_bdhi2d_create_f64": (_i, [C.POINTER(BDHI2DParametersF64), C.POINTER(_vp), C.POINTER(C.c_int * 2), C.POINTER(_i)]),
    "uammd_bdhi2d_destroy_f64": (_i, [_vp]),
    "uammd_bdhi2d_velocities_f64": (_i, [_vp, _vp, _vp, _i, _vp, _vp]),
    "uammd_bdhi2d_update_positions_f64": (_i, [_vp, _vp, _i, _d, _vp]),
    "uammd_poisson_create": (_i, [C.POINTER(PoissonParameters), C.POINTER(_vp), C.POINTER(PoissonInfo)]),
    "uammd_poisson_destroy": (_i, [_vp]),
    "uammd_poisson_set_option": (_i, [_vp, C.c_char_p, _i]),
    "uammd_poisson_sum": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _i, _vp]),
    "uammd_poisson_field_potential": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "uammd_fcm_create": (_i, [C.POINTER(FCMParameters), C.POINTER(_vp)]),
    "uammd_fcm_destroy": (_i, [_vp]),
    "uammd_fcm_displacements": (_i, [_vp, _vp, _vp, _i, _f, _f, _vp, _vp]),
    "uammd_fcm_displacements_staged": (_i, [_vp, _vp, _vp, _i, _f, _f, _vp, _i, _vp]),
    "uammd_fcm_step_euler_maruyama": (_i, [_vp, _vp, _vp, _i, _f, _f, _f, _vp, _i, _vp]),
    "uammd_fcm_export_fourier": (_i, [_vp, _vp, _vp]),
    "uammd_fcm_self_mobility": (C.c_double, [C.c_double, C.c_double, C.c_double]),
    "uammd_fcm_get_seed2": (_i, [_vp, C.POINTER(_u)]),
    "uammd_fcm_set_seed2": (_i, [_vp, _u]),
    "uammd_fcm_set_option": (_i, [_vp, C.c_char_p, _i]),
    "uammd_fcm_torque_gaussian_kernel": (_i, [_f, _f, _f, C.POINTER(IBMKernel)]),
    "uammd_fcm_set_torque_kernel": (_i, [_vp, C.POINTER(IBMKernel)]),
    "uammd_fcm_displacements_torque": (_i, [_vp, _vp, _vp, _vp, _i, _f, _f, _vp, _vp, _vp]),
    "uammd_fcm_euler_maruyama_dir": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _f, _vp]),
    "uammd_fcm_slab_create": (_i, [C.POINTER(FCMParameters), _i, _i, _i, _i, _i, C.POINTER(_vp)]),
    "uammd_fcm_slab_destroy": (_i, [_vp]),
    "uammd_fcm_slab_set_option": (_i, [_vp, C.c_char_p, _i]),
    "uammd_fcm_slab_spread": (_i, [_vp, _vp, _vp, _i, _vp, _vp]),
    "uammd_fcm_slab_gather": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "uammd_fcm_slab_forward_xy": (_i, [_vp, _vp, _vp]),
    "uammd_fcm_slab_transpose_pack": (_i, [_vp, _vp, _vp, _vp]),
    "uammd_fcm_slab_transpose_unpack": (_i, [_vp, _vp, _vp, _vp]),
    "uammd_fcm_slab_forward_xy_fold": (_i, [_vp, _vp, _vp, _vp, _i, _vp]),
    "uammd_fcm_slab_inverse_xy": (_i, [_vp, _vp, _vp]),
    "uammd_fcm_slab_inverse_xy_inter": (_i, [_vp, _vp, _vp, _vp]),
    "uammd_fcm_slab_inverse_xy_inter_wrap": (_i, [_vp, _vp, _vp, _i, _vp]),
    "uammd_fcm_slab_gather_inter": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "uammd_fcm_slab_fft_z": (_i, [_vp, _vp, _i, _vp]),
    "uammd_fcm_slab_kspace": (_i, [_vp, _vp, _i, _f, _f, _u, _vp]),
    "uammd_fcm_slab_z_fused": (_i, [_vp, _vp, _i, _f, _f, _u, _vp]),
    "uammd_pse_near_create": (_i, [_f3, _f, _f, _f, _f, _f, _u, C.POINTER(_vp), C.POINTER(_f), C.POINTER(_i)]),
    "uammd_pse_near_destroy": (_i, [_vp]),
    "uammd_pse_near_set_shear_strain": (_i, [_vp, _f]),
    "uammd_pse_near_set_option": (_i, [_vp, C.c_char_p, _i]),
    "uammd_pse_near_positions_changed": (_i, [_vp]),
    "uammd_pse_near_prepare": (_i, [_vp, _vp, _i, _vp]),
    "uammd_lanczos_get_schedule": (_i, [_vp, _vp]),
    "uammd_lanczos_set_schedule": (_i, [_vp, _vp]),
    "uammd_lanczos_set_interleave": (_i, [_vp, _vp, _vp]),
    "uammd_pse_near_set_interleave": (_i, [_vp, _vp, _vp]),
    "uammd_pse_near_set_interleave_early": (_i, [_vp, _vp, _vp]),
    "uammd_lanczos_set_interleave_early": (_i, [_vp, _vp, _vp]),
    "uammd_pse_near_pair_records": (_i, [_vp, _vp, _vp]),
    "uammd_pse_near_list_stats": (_i, [_vp, _vp]),
    "uammd_pse_near_set_mdot_rider": (_i, [_vp, _vp, _vp]),
    "uammd_pse_near_mdot": (_i, [_vp, _vp, _vp, _i, _vp, _vp]),
    "uammd_pse_near_stochastic": (_i, [_vp, _vp, _i, _f, _f, _u, _vp, _vp, C.POINTER(_i)]),
    "uammd_pse_near_noise": (_i, [_vp, _i, _f, _u, _vp, _vp]),
    "uammd_pse_near_dot": (_i, [_vp, _vp, _vp, _i, _vp, _vp]),
    "uammd_pse_far_raw_cells": (_i, [_f3, _f, _f, _i3]),
    "uammd_pse_far_create": (_i, [_f3, _i3, _f, _f, _f, _f, _f, _u, C.POINTER(_vp), C.POINTER(_i), C.POINTER(_f)]),
    "uammd_pse_far_set_shear_strain": (_i, [_vp, _f]),
    "uammd_pse_far_displacements": (_i, [_vp, _vp, _vp, _i, _f, _f, _u, _vp, _vp]),
    "uammd_pse_far_displacements_half": (_i, [_vp, _vp, _vp, _i, _f, _f, _u, _vp, _i, _vp]),
    "uammd_lanczos_create": (_i, [C.POINTER(_vp)]),
    "uammd_lanczos_destroy": (_i, [_vp]),
    "uammd_lanczos_run": (_i, [_vp, _vp, _vp, _vp, _vp, _f, _i, _vp, C.POINTER(_i)]),
    "uammd_lanczos_run_iterations": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, C.POINTER(_f)]),
    "uammd_lanczos_set_iteration_hard_limit": (_i, [_vp, _i]),
    "uammd_lanczos_set_option": (_i, [_vp, C.c_char_p, _i]),
    "uammd_lanczos_set_allreduce": (_i, [_vp, _vp, _vp, _i]),
    "uammd_lanczos_get_last_run_required_steps": (_i, [_vp, C.POINTER(_i)]),
    "uammd_rpy_nbody_mdot": (_i, [_vp, _vp, _i, _vp, _f, _f, _i, _vp, _vp]),
    "uammd_rpy_lanczos_bdw": (_i, [_vp, _vp, _vp, _f, _f, _i, _vp, _f, _vp, _vp, C.POINTER(_i)]),
    "uammd_rpy_nbody_mdot_f64": (_i, [_vp, _vp, _i, _vp, _d, _d, _i, _vp, _vp]),
    "uammd_rpy_lanczos_bdw_f64": (_i, [_vp, _vp, _vp, _d, _d, _i, _vp, _d, _vp, _vp, C.POINTER(_i)]),
    "uammd_bdhi_cholesky_create_f64": (_i, [_i, _d, _d, C.POINTER(_vp)]),
    "uammd_bdhi_cholesky_destroy_f64": (_i, [_vp]),
    "uammd_bdhi_cholesky_setup_step_f64": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "uammd_bdhi_cholesky_mf_f64": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "uammd_bdhi_cholesky_bdw_f64": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
}

MATVEC_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p)
INTERLEAVE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)

_lib = None


def load():
    """Loads the shared library (no GPU needed to load it) and installs the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise UammdHipError(f"{LIB_PATH} is missing: build it with `python -m uammd_amd.build` "
                            "(there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.uammd_hip_abi_version() != 1:
        raise UammdHipError("libuammd_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().uammd_hip_last_error()
        raise UammdHipError(f"libuammd_hip error {rc}: {msg.decode() if msg else '?'}")


def f3(v):
    import numpy as np
    a = np.broadcast_to(np.asarray(v, dtype=np.float32), (3,))
    return _f3(float(a[0]), float(a[1]), float(a[2]))


def i3(v):
    import numpy as np
    a = np.broadcast_to(np.asarray(v), (3,))
    return _i3(int(a[0]), int(a[1]), int(a[2]))
