#!/usr/bin/env python3
"""A table shaped like the reference profiler's own (profile_multiexp.cpp:275-399, 401-414: G1 and G2, one row per size,
32 distinct bases repeated, SHA512_rng scalars) with the GPU beside the reference on the same box:

  group, log2n, gpu_device_ms, gpu_host_entry_ms, ref_chunks1_ms, ref_best_ms, ref_best_chunks, gpu_result_equals_ref

  gpu_device_ms      amdmsm_msm_device, inputs resident in HBM (HIP events around the MSM's kernels, best of 3)
  gpu_host_entry_ms  amdmsm_multi_exp on host vectors -- what libff::multi_exp costs through the header shim: bases and
                     scalars over PCIe, import, MSM, result back (wall clock, best of 3 after a warm-up call)
  ref_chunks1_ms     the reference's multi_exp<BDLO12_signed, special> ("djb_signed_mixed"), chunks = 1: one core, what
                     the reference profiler measures (oracle/_ref = libff itself); up to --ref1-max
  ref_best_ms        the same with the best of chunks in {cores/4, cores/2, cores} (OpenMP over ranges); up to --ref-max
The crossover of gpu_host_entry_ms and ref_chunks1_ms is where libff_amd::small_input_threshold() comes from.

  python tools/profile_sweep.py [--out profiles/r04_profile_sweep.csv]
"""
import argparse
import csv
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import libff_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "profile_sweep.csv"))
    ap.add_argument("--sizes", type=int, nargs="+", default=list(range(1, 8)) + list(range(8, 21)) + [22, 24, 26])
    ap.add_argument("--g2-max", type=int, default=20)
    ap.add_argument("--ref1-max", type=int, default=20)
    ap.add_argument("--ref-max", type=int, default=24)
    ap.add_argument("--curve", type=int, default=0)
    args = ap.parse_args()
    from oracle import ref   # the reference itself: the timed CPU side of the table

    have_ref = ref.available()
    if have_ref:
        ref.lib()
    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    eng = libff_amd.Engine(0)
    eng.set_timing(True)
    curve = args.curve
    rows = []
    for group in (1, 2):
        sz = libff_amd.sizes(curve, group)
        gen = ref if have_ref else None
        if gen is None:
            from oracle import port as gen
            gen.build()
        p32 = gen.bases_r32(curve, group, 32)
        nmax = max(s for s in args.sizes if group == 1 or s <= args.g2_max)
        scal_all = None
        for lg in args.sizes:
            if group == 2 and lg > args.g2_max:
                continue
            n = 1 << lg
            bases = np.ascontiguousarray(np.tile(p32, ((n + 31) // 32, 1))[:n])
            # SHA512_rng scalars up to 2^22; above that the first 2^22 values repeated (the CPU generator is a second per 2^20)
            if scal_all is None:
                scal_all = gen.scalars_sha512(curve, 0, 1 << min(nmax, 22))
            sc = scal_all[:n] if n <= scal_all.shape[0] else np.ascontiguousarray(np.tile(scal_all, (n // scal_all.shape[0], 1)))
            # GPU, host entry
            got = eng.multi_exp(curve, group, bases, sc, base_form=libff_amd.multi_exp_base_form_special)
            host_ms = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                got = eng.multi_exp(curve, group, bases, sc, base_form=libff_amd.multi_exp_base_form_special)
                host_ms = min(host_ms, (time.perf_counter() - t0) * 1e3)
            # GPU, device-resident
            aff = np.ascontiguousarray(bases[:, : sz["affine_bytes"] // 8])
            d_b, d_s, d_o = eng.malloc(aff.nbytes), eng.malloc(sc.nbytes), eng.malloc(sz["g_bytes"])
            eng.h2d(d_b, aff)
            eng.h2d(d_s, sc)
            dev_ms = 1e9
            for _ in range(4):
                eng.msm_device(curve, group, d_b.value, d_s.value, n, d_o.value, out_form=libff_amd.OUT_AFFINE)
                eng.synchronize()
                dev_ms = min(dev_ms, eng.get_timings()["total_ms"])
            for q in (d_b, d_s, d_o):
                eng.free(q)
            del aff
            r1 = rbest = rchunks = eq = ""
            if have_ref and lg <= args.ref_max:
                if lg <= args.ref1_max:
                    want, secs = ref.multi_exp(curve, group, bases, sc, ref.BDLO12_SIGNED, ref.FORM_SPECIAL, chunks=1, want_time=True)
                    r1 = f"{secs * 1e3:.3f}"
                    eq = bool((ref.group_op(curve, group, 4, want) == got).all())
                if lg >= 10:
                    best = None
                    ref.multi_exp(curve, group, bases[:1 << 10], sc[:1 << 10], ref.BDLO12_SIGNED, ref.FORM_SPECIAL, chunks=cores)   # thread pool up
                    for ch in sorted({max(1, cores // 4), max(1, cores // 2), cores}):
                        if ch > n:
                            continue
                        want, secs = ref.multi_exp(curve, group, bases, sc, ref.BDLO12_SIGNED, ref.FORM_SPECIAL, chunks=ch, want_time=True)
                        if best is None or secs < best[0]:
                            best = (secs, ch)
                        if eq == "":
                            eq = bool((ref.group_op(curve, group, 4, want) == got).all())
                    if best:
                        rbest, rchunks = f"{best[0] * 1e3:.3f}", best[1]
            row = [f"{libff_amd.engine.CURVE_NAMES[curve]}_g{group}", lg, f"{dev_ms:.3f}", f"{host_ms:.3f}", r1, rbest, rchunks, eq]
            rows.append(row)
            print(",".join(str(x) for x in row), flush=True)
            del bases, sc
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["group", "log2n", "gpu_device_ms", "gpu_host_entry_ms", "ref_chunks1_ms", "ref_best_ms", "ref_best_chunks",
                    "gpu_result_equals_ref"])
        w.writerows(rows)
        f.write(f"# bases: 32 distinct points repeated (profile_multiexp.cpp:14-15,24-50), scalars SHA512_rng; reference = libff's own "
                f"multi_exp<BDLO12_signed, special> (oracle/_ref) on this box: {cores} hardware threads\n")


if __name__ == "__main__":
    main()
