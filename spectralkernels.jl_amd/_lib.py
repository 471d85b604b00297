"""ctypes binding of include/spectralkernels_hip.h.  Fails loudly if the HIP library is missing."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPECTRALKERNELS_HIP_LIB", os.path.join(HERE, "libspectralkernels_hip.so"))

SK_OK, SK_EINVAL, SK_ESUBDIVIDE, SK_ENAN, SK_ESTACK, SK_EHIP, SK_ENODEVICE, SK_EUNSUPPORTED = range(8)
SK_WARN_QUADSPEC = 100
SDF_CALLBACK, SDF_MATERN, SDF_EXPONENTIAL, SDF_GAUSSIAN = 0, 1, 2, 3

SDF_BATCH_FN = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_size_t, C.c_void_p)
SHARD_FN = C.CFUNCTYPE(C.c_int, C.c_int, C.POINTER(C.c_double), C.c_size_t, C.c_double, C.c_int, C.POINTER(C.c_double),
                       C.POINTER(C.c_double), C.c_void_p)


class ConfigArgs(C.Structure):
    _fields_ = [("sdf_kind", C.c_int), ("sdf_params", C.c_double * 8), ("f", SDF_BATCH_FN), ("f_ctx", C.c_void_p),
                ("df", SDF_BATCH_FN), ("df_ctx", C.c_void_p), ("dim", C.c_int), ("alpha", C.c_double),
                ("tol", C.c_double), ("derivative", C.c_int), ("logw", C.c_int), ("criteria", C.c_int),
                ("has_tail", C.c_int), ("tail", C.c_double), ("m", C.c_int64), ("k", C.c_int64),
                ("device", C.c_int), ("nufft_width", C.c_int), ("hankel_min_lags", C.c_int), ("sdf_dparam", C.c_int),
                ("batch_panels", C.c_int)]


class Stats(C.Structure):
    _fields_ = [("n_unique", C.c_int64), ("n_outer", C.c_int64), ("n_panels", C.c_int64), ("n_accepted", C.c_int64),
                ("sum_active", C.c_int64), ("n_steps_launched", C.c_int64), ("n_syncs", C.c_int64),
                ("last_b", C.c_double), ("n_steps", C.c_int64), ("n_rerun", C.c_int64),
                ("sum_active_steps", C.c_int64), ("n_fix", C.c_int64)]


K_NAMES = ["gen", "spread", "fft", "interp", "advance", "near"]


class KernelTimes(C.Structure):
    _fields_ = [("total_ms", C.c_double * 6), ("launches", C.c_int64 * 6), ("active_items", C.c_int64 * 6)]


_lib = None


def load():
    """Loads the in-tree HIP library.  There is no fallback: a missing library is an error."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libspectralkernels_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `python spectralkernels.jl_amd/build.py`. The HIP path has no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, dp = C.c_void_p, C.POINTER(C.c_double)
    L.sk_config_args_default.argtypes = [C.POINTER(ConfigArgs)]
    L.sk_config_args_default.restype = None
    L.sk_config_create.argtypes = [C.POINTER(vp), C.POINTER(ConfigArgs)]
    L.sk_config_destroy.argtypes = [vp]
    L.sk_config_destroy.restype = None
    L.sk_config_info.argtypes = [vp, dp, dp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int)]
    L.sk_config_rules.argtypes = [vp, C.c_int, vp, vp, vp, vp]
    L.sk_compute_k0.argtypes = [vp, dp]
    L.sk_kernel_values.argtypes = [vp, vp, C.c_size_t, C.c_double, C.c_int, C.c_int, vp, vp]
    L.sk_kernel_values_device.argtypes = [vp, vp, C.c_size_t, C.c_double, C.c_int, vp, vp, vp]
    L.sk_kernel_values_multi.argtypes = [vp, C.c_size_t, vp, C.c_size_t, vp, vp, vp, vp]
    L.sk_kernel_values_joint.argtypes = [vp, C.c_size_t, vp, C.c_size_t, C.c_double, vp, vp, vp]
    L.sk_kernel_jacobian_pairs.argtypes = [vp, C.c_size_t, vp, vp, vp, C.c_size_t, C.c_int, vp, C.c_size_t, C.c_double, vp,
                                           C.c_int, vp, C.c_int, vp, vp]
    L.sk_kernel_values_pairs.argtypes = [vp, vp, C.c_size_t, C.c_int, vp, C.c_size_t, C.c_double, C.c_int, C.c_int, vp, vp]
    L.sk_kernel_values_pairs_device.argtypes = [vp, vp, C.c_size_t, C.c_int, vp, C.c_size_t, C.c_double, C.c_int, vp, vp, vp]
    L.sk_dense_cov_matrix.argtypes = [vp, vp, C.c_size_t, C.c_int, vp]
    L.sk_dense_cov_matrix_device.argtypes = [vp, vp, C.c_size_t, C.c_int, vp, vp]
    L.sk_panel_sums.argtypes = [vp, vp, vp, C.c_size_t, C.c_double, C.c_double, vp, C.c_size_t, C.c_int, vp, vp]
    L.sk_last_host_timing.argtypes = [vp, dp]
    L.sk_last_step_kinds.argtypes = [vp, C.POINTER(C.c_longlong)]
    L.sk_host_threads.argtypes = [C.c_int]
    L.sk_device_count.argtypes = [C.POINTER(C.c_int)]
    L.sk_device_malloc.argtypes = [C.POINTER(vp), C.c_size_t, C.c_int]
    L.sk_device_free.argtypes = [vp]
    L.sk_memcpy_h2d.argtypes = [vp, vp, C.c_size_t]
    L.sk_memcpy_d2h.argtypes = [vp, vp, C.c_size_t]
    L.sk_device_synchronize.argtypes = [C.c_int]
    L.sk_device_mem_info.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.sk_host_alloc.argtypes = [C.POINTER(vp), C.c_size_t]
    L.sk_host_free.argtypes = [vp]
    L.sk_sharded_create.argtypes = [C.POINTER(vp), C.POINTER(ConfigArgs), C.POINTER(C.c_int), C.c_int]
    L.sk_sharded_create_injected.argtypes = [C.POINTER(vp), C.c_int, SHARD_FN, vp]
    L.sk_sharded_destroy.argtypes = [vp]
    L.sk_sharded_destroy.restype = None
    L.sk_sharded_count.argtypes = [vp]
    L.sk_kernel_values_sharded.argtypes = [vp, vp, C.c_size_t, C.c_double, C.c_int, C.c_int, vp, vp]
    L.sk_sharded_set_split.argtypes = [vp, C.c_int]
    L.sk_kernel_values_pairs_sharded.argtypes = [vp, vp, C.c_size_t, C.c_int, vp, C.c_size_t, C.c_double, C.c_int, C.c_int, vp, vp]
    L.sk_sharded_last.argtypes = [vp, C.c_int, C.POINTER(Stats), dp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.sk_last_stats.argtypes = [vp, C.POINTER(Stats)]
    L.sk_last_trace.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.sk_launch_floor.argtypes = [vp, C.c_int, dp]
    L.sk_profile_enable.argtypes = [vp, C.c_int]
    L.sk_profile_read.argtypes = [vp, C.POINTER(KernelTimes)]
    L.sk_host_gauss_rule.argtypes = [C.c_int64, C.c_double, vp, vp]
    L.sk_host_compute_k0.argtypes = [C.POINTER(ConfigArgs), dp]
    L.sk_host_compute_k0_part.argtypes = [C.POINTER(ConfigArgs), C.c_double, C.c_double, dp]
    L.sk_host_validate.argtypes = [C.POINTER(ConfigArgs), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.sk_host_es_eval.argtypes = [C.c_int, C.c_double, dp, dp]
    L.sk_abi_layout.argtypes = [C.c_int, C.POINTER(C.c_size_t), C.c_size_t, C.POINTER(C.c_size_t)]
    L.sk_host_stage_scan.argtypes = [vp, C.c_size_t, vp, C.c_int, C.POINTER(C.c_int)]
    L.sk_last_error.restype = C.c_char_p
    L.sk_version.restype = C.c_char_p
    L.sk_device_available.restype = C.c_int
    _lib = L
    return L


def abi_layout(which):
    """[sizeof, offset of every field] of struct `which` (0 sk_config_args, 1 sk_stats, 2 sk_kernel_times) as the library
    was compiled (sk_abi_layout)."""
    out = (C.c_size_t * 64)()
    n = C.c_size_t()
    if load().sk_abi_layout(which, out, 64, C.byref(n)) != 0:
        raise RuntimeError(last_error())
    return [int(out[i]) for i in range(n.value)]


def mirror_layout(cls):
    """The same description of a ctypes mirror."""
    return [C.sizeof(cls)] + [getattr(cls, name).offset for name, _ in cls._fields_]


def last_error():
    return load().sk_last_error().decode()
