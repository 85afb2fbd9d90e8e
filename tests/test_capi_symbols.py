"""The C-ABI shared library must load on a CPU-only host and export every symbol the
public headers declare (no compute calls here -- those need a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADERS = [os.path.join(ROOT, "include", h) for h in ("amdmsm.h", "libff_amd_ffi.h")]


def declared_symbols(path):
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    return sorted(set(re.findall(r"\b((?:amdmsm|alt_bn128|bls12_377|bw6_761)_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    import libff_amd
    from libff_amd import build

    if not os.path.exists(libff_amd.engine.SO_PATH):
        build.build()
    return ctypes.CDLL(libff_amd.engine.SO_PATH)


@pytest.mark.parametrize("header", HEADERS)
def test_every_declared_symbol_is_exported(lib, header):
    if not os.path.exists(header):
        pytest.skip(f"{os.path.basename(header)} not present yet")
    names = declared_symbols(header)
    assert len(names) >= 4
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in {os.path.basename(header)} but not exported: {missing}"


def test_python_binding_lists_the_same_symbols():
    import libff_amd.engine as e

    assert sorted(e.EXPORTED_SYMBOLS) == declared_symbols(HEADERS[0])


def test_sizes_and_plan_without_gpu(lib):
    """Pure host-side queries work without a device; element sizes = libff's sizeof."""
    import libff_amd
    from common import GROUPS, literal

    for name, curve, group in GROUPS:
        s = libff_amd.sizes(curve, group)
        lit = literal()["groups"][name]
        assert s["fr_bytes"] == lit["fr_bytes"] and s["g_bytes"] == lit["g_bytes"]
        assert s["affine_bytes"] == 2 * lit["coord_bytes"] and s["fr_bits"] == lit["fr_bits"]
        p = libff_amd.plan(curve, group, 1 << 20)
        assert p["num_buckets"] == 1 << (p["c"] - 1)
        # (with the endomorphism split -- the default where the curve group has prime order -- the
        # windows cover the half-length scalars instead: tests/test_endomorphism.py)
        p = libff_amd.plan(curve, group, 1 << 20, endomorphism=-1)
        assert p["num_windows"] * p["c"] >= lit["fr_bits"] + 2
    for n, c in literal()["bdlo12_signed_optimal_c"].items():
        assert libff_amd.bdlo12_signed_optimal_c(int(n)) == c
    for n, c in literal()["pippenger_optimal_c"].items():
        assert libff_amd.pippenger_optimal_c(int(n)) == c


def test_planner_window_rule_for_small_and_medium_inputs(lib):
    """The planner's measured window sizes (engine.cpp choose_c, profiles/r04_experiments.txt): c = 10 below 2^10 points,
    c = 13 (12 for bw6_761 below 2^15) up to 2^15 / 2^17 / 2^18 points depending on group and split, c = 16 beyond, and
    the configurations BASELINE.json names keep the choices the numbers in DESIGN.md were measured with."""
    import libff_amd

    def c_of(curve, group, lg, endo=0):
        p = libff_amd.plan(curve, group, 1 << lg, endomorphism=endo)
        return p["c"], p["endomorphism"]

    assert c_of(0, 1, 2) == (10, True) and c_of(0, 1, 9) == (10, True)
    assert c_of(0, 1, 10) == (13, True) and c_of(0, 1, 14) == (13, True)
    assert c_of(0, 1, 15) == (16, True) and c_of(0, 1, 20) == (16, True)
    assert c_of(0, 2, 4) == (10, False) and c_of(0, 2, 16) == (13, False) and c_of(0, 2, 18) == (16, False)
    assert c_of(0, 2, 16, endo=1) == (13, True) and c_of(0, 2, 17, endo=1) == (16, True)
    assert c_of(2, 1, 12, endo=1) == (12, True) and c_of(2, 1, 16, endo=1) == (13, True)
    # BASELINE configs: [1] 2^20, the 2^23 shard of [3], 2^26; [2] bls12_377 G1 2^22; [4] shards
    assert c_of(0, 1, 23) == (19, True) and c_of(0, 1, 26) == (20, False)
    assert c_of(1, 1, 22) == (17, False)
    assert c_of(2, 1, 21, endo=1) == (16, True) and c_of(1, 2, 21, endo=1) == (16, True)


def test_compute_fails_loudly_without_gpu(lib):
    """No CPU fallback: creating an engine on a GPU-less host must raise."""
    import libff_amd

    if lib.amdmsm_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(libff_amd.AmdMsmError):
        libff_amd.Engine(0)
