"""libff_amd -- MI355X-native drop-in for libff's multi_exp (Pippenger/BDLO12) hot path.

Only what the path needs lives here:
  csrc/            hand-written HIP kernels (gfx950) + the C-ABI engine (include/amdmsm.h)
  engine.py        host-side mirror of libff's multi_exp interface over the C ABI (ctypes)
  distributed.py   range-sharded multi-GPU MSM (one process per GPU, RCCL all-gather of partials)
  build.py         hipcc build of libamdmsm.so (in-tree)

There is no CPU implementation in this package: importing is cheap, but every compute
entry raises if libamdmsm.so is missing or no GPU is visible.
"""
from .engine import (  # noqa: F401
    ALT_BN128, BLS12_377, BLS12_381, BW6_761, G1, G2, OUT_AFFINE, OUT_JACOBIAN, OUT_LIBFF, AmdMsmError, Engine,
    bdlo12_signed_optimal_c, load_library, multi_exp_base_form_normal, multi_exp_base_form_special,
    multi_exp_method_BDLO12, multi_exp_method_BDLO12_signed, multi_exp_method_bos_coster,
    multi_exp_method_naive, multi_exp_method_naive_plain, multi_exp_multi, multi_exp_filter_one_zero_multi, msm_device_multi, pippenger_optimal_c, plan, precompute_num_digits, sizes,
    endomorphism_info)

__all__ = [
    "ALT_BN128", "BLS12_377", "BLS12_381", "BW6_761", "G1", "G2", "OUT_AFFINE", "OUT_JACOBIAN", "OUT_LIBFF",
    "AmdMsmError", "Engine", "bdlo12_signed_optimal_c", "load_library", "multi_exp_base_form_normal",
    "multi_exp_base_form_special", "multi_exp_method_BDLO12", "multi_exp_method_BDLO12_signed",
    "multi_exp_method_bos_coster", "multi_exp_method_naive", "multi_exp_method_naive_plain", "multi_exp_multi",
    "multi_exp_filter_one_zero_multi", "msm_device_multi",
    "pippenger_optimal_c", "plan", "precompute_num_digits", "sizes", "endomorphism_info",
]
