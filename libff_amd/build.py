"""Build libff_amd/libamdmsm.so for gfx950 with hipcc (in-tree, no JIT cache).

One translation unit per (curve, group) pair -- msm_group.hip compiled with
-DAMDMSM_GROUP=... -- plus the host engine; the eight device TUs build in parallel.
Objects are cached under libff_amd/csrc/build/ keyed by source mtimes.
"""
import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
SO_PATH = os.path.join(HERE, "libamdmsm.so")

GROUPS = ["alt_bn128_g1", "alt_bn128_g2", "bls12_377_g1", "bls12_377_g2", "bw6_761_g1", "bw6_761_g2",
          "bls12_381_g1", "bls12_381_g2"]
# per-group code-generation policy (see fp.cuh Fp<P, INL> and msm_group.hip):
#   AMDMSM_HOT_INLINE  inline the Montgomery product inside the bucket-accumulation loop
#   AMDMSM_BENCH_BOTH  also build the inline variants of the throughput probes
GROUP_FLAGS = {g: ["-DAMDMSM_HOT_INLINE=1", "-DAMDMSM_BENCH_BOTH=1"] for g in GROUPS}
# register budget of the bucket-accumulation kernel: 4 waves per SIMD (128 VGPRs, 3 dwords of
# scratch) instead of the 137 VGPRs / 3 waves the compiler picks unconstrained: -4 % at 2^20
# (superseded by the reduced-radix loop below, which is built for three)
# overlap mode (several MSMs in flight: the tail of one under the sort and accumulation of the next, engine.cpp
# amdmsm_ctx::bulk_stream): the tail kernels get the 128 registers three accumulation waves leave of a SIMD
# (round 4: the tail kernels' serial sums run on reduced-radix limbs and want ~195 registers; capped at 128 for the overlap
# mode -- which stays off by default, it measured slower -- they spilled 292 B per lane: reduction phase 0.327 / 1.108 / 3.19 ms
# at 2^20 / 2^23 / 2^26 capped against 0.310 / 1.025 / 3.11 uncapped, profiles/r04_experiments.txt.  No cap any more.)
GROUP_FLAGS["alt_bn128_g1"] = GROUP_FLAGS["alt_bn128_g1"] + ["-DAMDMSM_OVERLAP_OK=1"]
# wide fields: unconstrained, the kernel takes 256 VGPRs plus 50..125 AGPRs as spill space and
# runs ONE wave per SIMD; capped at 256 registers (two waves per SIMD, a little scratch) it is
# 23-25 % faster (bls12_377 G2 2^21: 28.1 -> 22.9 ms, bw6_761 G1 2^21: 43.8 -> 35.0 ms);
# 12-limb G1: 168 registers / three waves, +4 %.  alt_bn128 G2 is fastest unconstrained
# (249 VGPRs, two waves).
for _g in ("bls12_377_g2", "bls12_381_g2", "bw6_761_g1", "bw6_761_g2"):
    GROUP_FLAGS[_g] = GROUP_FLAGS[_g] + ["-DAMDMSM_ACC_WAVES=2"]
# 8- and 12-limb G1: Montgomery products inlined in the cold kernels too (fix-up, bucket reduction,
# butterfly; no call / operand moves around the ~1 us product of a lone wave): reduction phase
# 0.72 -> 0.66 ms (alt_bn128 2^20), 2.74 -> 2.10 ms (bls12_377 2^22).  No gain for Fq2 / 24 limbs.
for _g in ("alt_bn128_g1", "bls12_377_g1", "bls12_381_g1"):
    GROUP_FLAGS[_g] = GROUP_FLAGS[_g] + ["-DAMDMSM_COLD_INLINE=1"]
# Fq2 groups: every element split over a pair of lanes in k_accumulate (fp2h.cuh): same
# multiply-accumulate count, half the registers per lane -- bls12_377 G2 256 VGPRs + 16 B of scratch
# at two waves instead of + 440 B: 22.4 -> 18.1 ms at 2^21 (1.86 G madd/s, 0.90 of the MAC bound);
# alt_bn128 G2 168 VGPRs at three waves: 4.67 -> 4.57 ms at 2^20
for _g in ("bls12_377_g2", "bls12_381_g2", "alt_bn128_g2"):
    GROUP_FLAGS[_g] = GROUP_FLAGS[_g] + ["-DAMDMSM_ACC_SPLIT=1"]
GROUP_FLAGS["alt_bn128_g2"] = GROUP_FLAGS["alt_bn128_g2"] + ["-DAMDMSM_ACC_WAVES=3"]
for _g in ("bls12_377_g1", "bls12_381_g1"):
    GROUP_FLAGS[_g] = GROUP_FLAGS[_g] + ["-DAMDMSM_ACC_WAVES=3"]
# k_accumulate on reduced-radix limbs for the prime-field groups (rr.cuh: 28 / 29-bit signed limbs, one
# v_mad_i64_i32 per limb product and no carry instruction behind it, limb-wise linear operations): 1.3x the
# mixed additions per second of the 32-bit loop on every field (tools/proto_rr.hip).  Register budget: three
# waves per SIMD for the 9-limb field (134 registers), two for 14 and 28 limbs.
for _g, _w in (("alt_bn128_g1", 3), ("bls12_377_g1", 2), ("bls12_381_g1", 2), ("bw6_761_g1", 2), ("bw6_761_g2", 2),
               ("alt_bn128_g2", 3), ("bls12_377_g2", 2), ("bls12_381_g2", 2)):
    GROUP_FLAGS[_g] = [f for f in GROUP_FLAGS[_g] if not f.startswith("-DAMDMSM_ACC_WAVES=")] + ["-DAMDMSM_ACC_RR=1", f"-DAMDMSM_ACC_WAVES={_w}"]
ARCH = "gfx950"
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-I" + INCLUDE, "-I" + CSRC,
          "-Wno-unused-result"]
DEVICE_DEPS = ["msm_group.hip", "rr.cuh", "rr_chain.inc", "fp.cuh", "fp2.cuh", "fp2h.cuh", "ec.cuh", "wide.cuh", "wide28.cuh", "mac_chain.inc", "curve_params.h", "group_vtable.h", os.path.join(HERE, "build.py")]
HOST_DEPS = ["engine.cpp", "ffi.cpp", "engine_internal.h", "group_vtable.h", os.path.join(INCLUDE, "amdmsm.h"),
             os.path.join(INCLUDE, "libff_amd_ffi.h")]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the amdmsm engine is HIP-only and cannot be built without ROCm")
    return exe


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d if os.path.isabs(d) else os.path.join(CSRC, d)) > t for d in deps)


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout[-4000:], r.stderr[-8000:]))
    return r


def build(force=False, verbose=True, jobs=None):
    os.makedirs(OBJ, exist_ok=True)
    cc = hipcc()
    tasks = []
    objs = []
    groups = [g for g in os.environ.get("AMDMSM_GROUPS", ",".join(GROUPS)).split(",") if g]
    # AMDMSM_EXTRA_FLAGS="-DAMDMSM_ACC_WAVES=4 ...": experiment knobs appended to every group TU
    extra = os.environ.get("AMDMSM_EXTRA_FLAGS", "").split()
    for g in groups:
        if g not in GROUPS:
            raise RuntimeError(f"unknown group {g!r}")
        o = os.path.join(OBJ, f"group_{g}.o")
        objs.append(o)
        if force or _newer(o, DEVICE_DEPS):
            # an experiment flag replaces the group's own setting of the same macro
            names = {f.split("=")[0] for f in extra}
            own = [f for f in GROUP_FLAGS.get(g, []) if f.split("=")[0] not in names]
            tasks.append([cc, *COMMON, "-c", os.path.join(CSRC, "msm_group.hip"),
                          f"-DAMDMSM_GROUP={g}", f"-DAMDMSM_VT=vt_{g}", *own, *extra, "-o", o])
    # AMDMSM_FFI_NO_REFERENCE_SYMBOLS=1: leave out the reference's own FFI names (<curve>_init / _g1_add /
    # _g1_mul, include/libff_amd_ffi.h) so that the library can sit next to libff-ffi
    no_ref = os.environ.get("AMDMSM_FFI_NO_REFERENCE_SYMBOLS", "0") not in ("", "0")
    for src in ("engine", "ffi"):
        eo = os.path.join(OBJ, f"{src}{'_noref' if no_ref and src == 'ffi' else ''}.o")
        objs.append(eo)
        if force or _newer(eo, HOST_DEPS):
            tasks.append([cc, *COMMON, *(["-DAMDMSM_FFI_NO_REFERENCE_SYMBOLS=1"] if no_ref and src == "ffi" else []),
                          "-x", "hip", "-c", os.path.join(CSRC, f"{src}.cpp"), "-o", eo])
    if tasks:
        if verbose:
            print(f"[libff_amd.build] compiling {len(tasks)} translation unit(s) for {ARCH} ...", flush=True)
        jobs = jobs or min(len(tasks), max(1, (os.cpu_count() or 2) - 1))
        with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as ex:
            list(ex.map(_run, tasks))
    if tasks or not os.path.exists(SO_PATH) or no_ref != os.path.exists(SO_PATH + ".noref"):
        _run([cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-Wl,-soname,libamdmsm.so", "-o", SO_PATH, *objs])
        if no_ref:
            open(SO_PATH + ".noref", "w").close()   # marker: this link carries no reference FFI names
        elif os.path.exists(SO_PATH + ".noref"):
            os.remove(SO_PATH + ".noref")
        if verbose:
            print(f"[libff_amd.build] linked {SO_PATH}", flush=True)
    return SO_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
