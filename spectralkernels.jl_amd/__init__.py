"""MI355X-native kernel_values hot path of SpectralKernels.jl (HIP kernels behind a C ABI)."""
from .api import (SDF, AdaptiveKernelConfig, DeviceArray, ShardedKernelConfig, SpectralKernelsError, build_dense_cov_matrix,
                  compute_k0, device_available, device_count, device_mem_info, gen_derivative_config, gen_new_sdf_config, host_timing, step_kinds, kernel_jacobian_pairs, kernel_values, kernel_values_device, kernel_values_into,
                  kernel_values_joint, kernel_values_multi, kernel_values_pairs, panel_sums, pinned_empty, recognize_sdf, synchronize)

__all__ = ["SDF", "AdaptiveKernelConfig", "SpectralKernelsError", "compute_k0", "kernel_values",
           "kernel_values_device", "kernel_values_into", "kernel_values_multi", "kernel_values_joint", "kernel_jacobian_pairs", "kernel_values_pairs", "build_dense_cov_matrix", "panel_sums",
           "device_available", "device_count", "device_mem_info", "DeviceArray", "pinned_empty", "synchronize", "host_timing", "step_kinds", "gen_derivative_config", "gen_new_sdf_config", "recognize_sdf",
           "ShardedKernelConfig"]
