"""Run the drop-in template-shim check (oracle/shim_check.cpp): the reference's own
multi_exp<> call sites compiled against include/libff_amd/multiexp.hpp, GPU result compared
with libff's CPU naive_plain using libff's operator==.  The binary links the reference, so it
is built only where /root/reference is mounted (oracle/build_ref.sh) and travels to the GPU
box prebuilt under oracle/_ref/."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "shim_check")


@pytest.mark.gpu
def test_template_shim_with_real_libff_types():
    if not os.path.exists(BIN):
        pytest.skip("oracle/_ref/shim_check not built (needs the reference sources at build time)")
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=900)
    print(r.stdout[-3000:])
    print(r.stderr[-2000:])
    assert r.returncode == 0 and "SHIM CHECK PASSED" in r.stdout
    assert r.stdout.count("small-input routing n=1..64: ok") == 3   # both routes of the small-input threshold, three groups
    # the 2^20-point call with chunks = 16 (one MSM, not 16) and the registered-bases call ran
    assert "chunks=16" in r.stdout and "filter_one_zero ok" in r.stdout


@pytest.mark.gpu
def test_template_shim_multi_device_route():
    """The same program with two engine contexts configured (AMDMSM_DEVICES=0,0: both on the one GPU
    of the test box) and a split threshold low enough that every case with >= 200 points goes
    through amdmsm_multi_exp_multi (multiexp.tcc:655-687 with chunk = context)."""
    if not os.path.exists(BIN):
        pytest.skip("oracle/_ref/shim_check not built (needs the reference sources at build time)")
    env = dict(os.environ, AMDMSM_DEVICES="0,0", SHIM_CHECK_MIN_SPLIT="100", SHIM_CHECK_SKIP_LARGE="1", SHIM_CHECK_SKIP_SMALL="1")
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=900, env=env)
    print(r.stdout[-3000:])
    print(r.stderr[-2000:])
    assert r.returncode == 0 and "SHIM CHECK PASSED" in r.stdout


@pytest.mark.gpu
def test_template_shim_with_endomorphism_permitted():
    """The same program with libff_amd::endomorphism_mode() = 1: G2 and the bls12_377 / bw6_761 groups
    split their scalars too (their bases are libff group elements: in the order-r subgroup)."""
    if not os.path.exists(BIN):
        pytest.skip("oracle/_ref/shim_check not built (needs the reference sources at build time)")
    env = dict(os.environ, SHIM_CHECK_ENDOMORPHISM="1", SHIM_CHECK_SKIP_LARGE="1", SHIM_CHECK_SKIP_SMALL="1")
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=900, env=env)
    print(r.stdout[-3000:])
    print(r.stderr[-2000:])
    assert r.returncode == 0 and "SHIM CHECK PASSED" in r.stdout and "endomorphism mode 1" in r.stdout


def test_shim_header_compiles_against_reference():
    """CPU-side: the header is valid C++11 against the reference headers (syntax + template
    selection); needs /root/reference and gmp.h, so it is skipped on the GPU box."""
    ref = "/root/reference"
    gmpinc = os.path.join(ROOT, "oracle", "_ref", "gmpinc")
    if not os.path.isdir(os.path.join(ref, "libff")) or not os.path.isdir(gmpinc):
        pytest.skip("reference headers not available here")
    src = os.path.join(ROOT, "oracle", "shim_check.cpp")
    cmd = ["g++", "-std=c++11", "-fsyntax-only", "-DNDEBUG", "-DCURVE_ALT_BN128", "-DNO_PROCPS", "-DBINARY_OUTPUT",
           "-DMONTGOMERY_OUTPUT", "-DUSE_ASM", "-w", "-I" + ref, "-I" + gmpinc, "-I" + os.path.join(ROOT, "include"), src]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]


def test_shim_small_inputs_run_the_callers_libff_without_a_device():
    """Below libff_amd::small_input_threshold() the routed libff::multi_exp runs the reference's own generic body in the
    caller's translation unit (multiexp.tcc:655-661 is what the reference does with a handful of points): n = 1 .. 64 for
    three groups against naive_plain, in a process that never touches a GPU.  Needs the binary built against the
    reference, i.e. this container."""
    import torch

    if not os.path.exists(BIN) or torch.cuda.is_available():
        pytest.skip("oracle/_ref/shim_check not built, or a GPU box (the device tests above cover both routes there)")
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    r = subprocess.run([BIN, "--cpu-route-only"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "SHIM CPU ROUTE PASSED" in r.stdout and r.stdout.count("routing n=1..64: ok") == 3, r.stdout[-2000:]
