#!/usr/bin/env python3
"""Benchmark of the hot path: G1 multi-scalar multiplication on MI355X.

  python bench.py --gpus N --steps K --warmup W [--log2n L] [--curve alt_bn128]

One "step" = one full MSM (libff::multi_exp) over this rank's shard of synthetic
(scalar, base) pairs that are already resident in HBM.  With N > 1 (launched by
torch.distributed.run, one rank per GPU) every rank owns a contiguous range of the
input (weak scaling: 2^L points per GPU), reduces it to one partial point and the
partials are exchanged with one RCCL all-gather and summed on every rank
(multiexp.tcc:663-687 with rank == chunk).  value = total scalar-muls/s of the job.

The JSON line also carries
  roofline      bucket-accumulation kernel (dominant): algorithmic bytes per launch
                (96 B per alt_bn128 G1 point x points per launch) over its
                mean duration from HIP events on the launch stream, against the 8 TB/s
                HBM peak.  The path is integer-ALU bound; the fraction is small by nature.
  cpu_baseline  libff's own multi_exp (oracle/_ref, "reference") or the C restatement
                ("port") timed on the host cores of this box on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import libff_amd  # noqa: E402
from libff_amd.distributed import ShardedMsm  # noqa: E402

CURVES = {"alt_bn128": 0, "bls12_377": 1, "bw6_761": 2, "bls12_381": 3}
# one scalar + one affine base per scalar-mul
ALGO_BYTES = {(0, 1): 96, (1, 1): 128, (1, 2): 224, (2, 1): 240, (0, 2): 160, (2, 2): 240, (3, 1): 128, (3, 2): 224}
HBM_PEAK_GBS = 8000.0
# What bounds k_accumulate is the issue rate of its multiply-accumulate pairs (v_mad_u64_u32 +
# v_addc_co_u32, both half rate): 33.15 T lane-instructions/s measured with 4 waves per SIMD
# (profiles/r01_ubench_instruction_rates.txt).  One Fq product = 2 N^2 pairs on N 32-bit limbs; one
# mixed addition = 10 coordinate products (8M + 2S), a coordinate product in Fq2 = 3 (M) or 2 (S)
# Fq products.
MAC_PAIR_PEAK = 33.15e12
FQ_LIMBS = {0: 8, 1: 12, 2: 24, 3: 12}
FQ_PRODUCTS_PER_MADD = {1: 10, 2: 28}
FR_MODULUS = {
    0: 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001,
    1: 0x12AB655E9A2CA55660B44D1E5C37B00159AA76FED00000010A11800000000001,
    2: 0x1AE3A4617C510EAC63B05C06CA1493B1A22D9F300F5138F1EF3622FBA094800170B5D44300000008508C00000000001,
    3: 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001,
}


def random_scalars(curve, n, device, seed):
    """Uniform residues in [0, r) as (n, limbs) int64 = libff Fr layout (rejection sampling,
    like SHA512_rng: mask to the modulus bit length, redraw rows >= r; rng.tcc:49-66)."""
    r = FR_MODULUS[curve]
    limbs = (r.bit_length() + 63) // 64
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    r_limbs = [(r >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(limbs)]
    top_mask = (1 << (r.bit_length() - 64 * (limbs - 1))) - 1

    def draw(m):
        x = torch.randint(-(1 << 63), (1 << 63) - 1, (m, limbs), dtype=torch.int64, device=device, generator=gen)
        x[:, limbs - 1] &= top_mask
        return x

    def ge_r(x):
        # lexicographic compare from the top limb, unsigned: flip the sign bit for ordering
        sb = -(1 << 63)
        ge = torch.ones(x.shape[0], dtype=torch.bool, device=device)
        decided = torch.zeros_like(ge)
        for i in range(limbs - 1, -1, -1):
            ri = r_limbs[i] - (1 << 64) if r_limbs[i] >= (1 << 63) else r_limbs[i]
            xi = x[:, i] ^ sb
            rv = ri ^ sb
            gt, lt = xi > rv, xi < rv
            ge = torch.where(~decided & lt, torch.zeros_like(ge), ge)
            decided = decided | gt | lt
        return ge   # equal rows stay True (x == r is rejected)

    x = draw(n)
    bad = ge_r(x)
    while bool(bad.any()):
        idx = bad.nonzero().squeeze(1)
        x[idx] = draw(idx.numel())
        bad = ge_r(x)
    return x


def pmc_traffic(curve_name, group, log2n, window_bits):
    """HBM bytes per k_accumulate launch from the committed rocprofv3 PMC passes
    (profiles/r*_pmc_*.json: separate FETCH_SIZE / WRITE_SIZE runs of this very command,
    gfx950 correction 2*FETCH + WRITE, see the file), or None when no pass matches the workload."""
    import glob

    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*pmc*.json"))):
        try:
            d = json.load(open(path))
            wl = d["workload"]
            if (wl["curve"], wl["group"], wl["log2n"], wl["window_bits"]) != (curve_name, group, log2n, window_bits):
                continue
            k = d["kernels"]["k_accumulate"]
            best = (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0
        except Exception:
            continue
    return best


def cpu_baseline(curve, group, log2n_sample):
    """Time the CPU path on this box: the reference's multi_exp<BDLO12_signed, special> when
    oracle/_ref/libff_ref.so is present, else the C restatement; all host cores, one range
    per core (multiexp.tcc:667-679)."""
    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    n = 1 << log2n_sample
    from oracle import ref

    if ref.available():
        kind, be = "reference", ref
        be.lib()
        bases = be.bases_seq(curve, group, n)
        scalars = be.scalars_sha512(curve, 0, n)
        _, secs = be.multi_exp(curve, group, bases, scalars, be.BDLO12_SIGNED, be.FORM_SPECIAL, chunks=cores,
                               want_time=True)
    else:
        from oracle import port

        kind, be = "port", port
        be.build()
        bases = be.bases_seq(curve, group, n)
        scalars = be.scalars_sha512(curve, 0, n)
        t0 = time.perf_counter()
        be.multi_exp(curve, group, bases, scalars, be.BDLO12_SIGNED, be.FORM_SPECIAL, chunks=cores, omp=True)
        secs = time.perf_counter() - t0
    return {"value": n / secs, "unit": "scalar-muls/s", "cores": cores, "kind": kind,
            "sample": f"first 2^{log2n_sample} (scalar, base) pairs of the workload family (SHA512_rng scalars, "
                      f"bases (i+1)G), multi_exp<BDLO12_signed, special>, chunks={cores} (OpenMP), {secs:.2f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log2n", type=int, default=int(os.environ.get("AMDMSM_BENCH_LOG2N", "20")),
                    help="points per GPU = 2^log2n")
    ap.add_argument("--curve", default="alt_bn128", choices=sorted(CURVES))
    ap.add_argument("--group", type=int, default=1, choices=(1, 2))
    ap.add_argument("--window-bits", type=int, default=0)
    ap.add_argument("--cpu-sample-log2n", type=int, default=18)
    ap.add_argument("--extra-log2n", type=int, default=int(os.environ.get("AMDMSM_BENCH_EXTRA_LOG2N", "26")),
                    help="also time this size after the main region (reported under 'also'); 0 disables")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pipeline", type=int, default=int(os.environ.get("AMDMSM_BENCH_PIPELINE", "1")),
                    help="MSMs in flight in the timed region (1 = strictly one after the other, which is "
                         "what `value` and the roofline kernel timings are quoted on)")
    ap.add_argument("--also-pipelined", type=int, default=3,
                    help="after the timed region, also measure the same workload with this many MSMs in flight "
                         "(reported under 'pipelined'); 0 disables")
    ap.add_argument("--precomputed-c", type=int, default=16,
                    help="also time the precomputed-multiples MSM (multi_exp_stream_with_precompute's algorithm on an "
                         "HBM-resident table of [2^(jc)]P) with this window size (reported under 'precomputed'); "
                         "0 disables")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                             "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    curve, group = CURVES[args.curve], args.group
    n = 1 << args.log2n
    sz = libff_amd.sizes(curve, group)
    eng = libff_amd.Engine(local_rank)
    plan = libff_amd.plan(curve, group, n, args.window_bits)

    # ---- synthetic inputs, resident in HBM before the timed region ----------
    stream = torch.cuda.current_stream().cuda_stream
    bases = torch.empty((n, sz["affine_bytes"] // 8), dtype=torch.int64, device=dev)
    eng.gen_bases_seq_device(curve, group, rank * n, n, bases.data_ptr(), stream=stream)   # (rank*n + i + 1) * G
    scalars = random_scalars(curve, n, dev, seed=1234 + rank)
    torch.cuda.synchronize()

    msm = ShardedMsm(eng, curve, group, depth=max(1, args.pipeline))
    eng.set_timing(True)

    def step():
        return msm.run(bases, scalars, n, libff_amd.OUT_LIBFF, window_bits=args.window_bits)

    for _ in range(max(args.warmup, msm.depth if args.warmup else 0)):
        step()
    msm.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    acc_ms = []
    phases = []
    pending = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        _, slot = step()
        pending.append(slot)
        # read the HIP-event timings of a step only once a later step is already enqueued
        while len(pending) >= msm.depth:
            t = eng.get_timings(pending.pop(0))
            acc_ms.append(t["accumulate_ms"])
            phases.append(t)
    while pending:
        t = eng.get_timings(pending.pop(0))
        acc_ms.append(t["accumulate_ms"])
        phases.append(t)
    msm.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- same workload with several MSMs in flight (throughput of back-to-back MSMs) ----
    pipelined = None
    if args.also_pipelined > 1 and msm.depth == 1:
        pm = ShardedMsm(eng, curve, group, depth=args.also_pipelined)
        for _ in range(pm.depth):
            pm.run(bases, scalars, n, libff_amd.OUT_LIBFF, window_bits=args.window_bits)
        pm.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        kp = max(args.steps, 2 * pm.depth)
        tp = time.perf_counter()
        for _ in range(kp):
            pm.run(bases, scalars, n, libff_amd.OUT_LIBFF, window_bits=args.window_bits)
        pm.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        ep = time.perf_counter() - tp
        if world > 1:
            tt = torch.tensor([ep], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            ep = float(tt.item())
        pipelined = {"msms_in_flight": pm.depth, "steps": kp, "value": n * world * kp / ep, "unit": "scalar-muls/s",
                     "ms_per_step": ep / kp * 1e3,
                     "note": "consecutive MSMs issued on alternating streams / workspace slots so the few-wave "
                             "tail of one overlaps the bulk kernels of the next; per-kernel timings are not "
                             "comparable with the un-overlapped ones, so `value` stays the depth-1 figure"}
        msm = ShardedMsm(eng, curve, group, depth=1)

    # ---- fixed bases with precomputed multiples (multi_exp_stream_with_precompute, HBM-resident) ----
    precomputed = None
    if args.precomputed_c:
        pc = args.precomputed_c
        D = libff_amd.precompute_num_digits(curve, pc)
        table = torch.empty((n * D, sz["affine_bytes"] // 8), dtype=torch.int64, device=dev)
        tb = time.perf_counter()
        eng.precompute_bases_device(curve, group, bases.data_ptr(), n, pc, D, table.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        build_s = time.perf_counter() - tb
        out_pt = torch.zeros(sz["g_bytes"] // 8, dtype=torch.int64, device=dev)
        for _ in range(2):
            eng.msm_precomputed_device(curve, group, table.data_ptr(), scalars.data_ptr(), n, pc, D, out_pt.data_ptr(),
                                       stream=stream)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        kq = max(args.steps, 4)
        tq = time.perf_counter()
        for _ in range(kq):
            eng.msm_precomputed_device(curve, group, table.data_ptr(), scalars.data_ptr(), n, pc, D, out_pt.data_ptr(),
                                       stream=stream)
        phq = eng.get_timings()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        eq = time.perf_counter() - tq
        if world > 1:
            tt = torch.tensor([eq], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            eq = float(tt.item())
        precomputed = {"window_bits": pc, "multiples_per_base": D, "table_gib": table.numel() * 8 / 2**30,
                       "table_build_ms": build_s * 1e3, "steps": kq, "value": n * world * kq / eq,
                       "unit": "scalar-muls/s", "ms_per_step": eq / kq * 1e3, "phases_ms_last": phq,
                       "note": "libff's multi_exp_stream_with_precompute algorithm (one bucket set over the table "
                               "[2^(jc)]P_i, no doublings) with the table resident in HBM; per-rank results are not "
                               "combined in this leg; a different reference entry point than `value`'s multi_exp"}
        del table

    # ---- second size of the metric (2^26 by default), outside the main timed region ----
    also = None
    if args.extra_log2n and args.extra_log2n != args.log2n:
        n2 = 1 << args.extra_log2n
        del bases, scalars
        torch.cuda.empty_cache()
        bases2 = torch.empty((n2, sz["affine_bytes"] // 8), dtype=torch.int64, device=dev)
        eng.gen_bases_seq_device(curve, group, rank * n2, n2, bases2.data_ptr(), stream=stream)
        scalars2 = random_scalars(curve, n2, dev, seed=4321 + rank)
        torch.cuda.synchronize()
        for _ in range(msm.depth):
            msm.run(bases2, scalars2, n2, libff_amd.OUT_LIBFF)
        msm.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        k2 = 4
        t1 = time.perf_counter()
        for _ in range(k2):
            _, slot2 = msm.run(bases2, scalars2, n2, libff_amd.OUT_LIBFF)
        ph2 = eng.get_timings(slot2)
        msm.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        e2 = time.perf_counter() - t1
        if world > 1:
            tt = torch.tensor([e2], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            e2 = float(tt.item())
        p2 = libff_amd.plan(curve, group, n2)
        also = {"workload": f"{args.curve} G{group} MSM, 2^{args.extra_log2n} points per GPU", "steps": k2,
                "value": n2 * world * k2 / e2, "unit": "scalar-muls/s", "ms_per_step": e2 / k2 * 1e3,
                "window_bits": p2["c"], "num_windows": p2["num_windows"], "phases_ms_last": ph2}
        del bases2, scalars2

    if rank == 0:
        total_points = n * world * args.steps
        value = total_points / elapsed
        acc = float(np.mean(acc_ms))
        algo_bytes = ALGO_BYTES[(curve, group)] * n
        achieved = algo_bytes / (acc * 1e-3) / 1e9
        mean_phase = {k: float(np.mean([p[k] for p in phases])) for k in phases[0]}
        fq_products = FQ_PRODUCTS_PER_MADD[group if not (curve == 2) else 1]
        lane_instr = float(n) * plan["num_windows"] * fq_products * 4 * FQ_LIMBS[curve] ** 2
        mac_rate = lane_instr / (acc * 1e-3)
        mac_issue = {"achieved": mac_rate / 1e12, "peak": MAC_PAIR_PEAK / 1e12, "unit": "T lane-instr/s",
                     "frac": mac_rate / MAC_PAIR_PEAK,
                     "what": "v_mad_u64_u32 + v_addc_co_u32 issues of the Montgomery products in k_accumulate "
                             "(upper bound: every list entry counted as a full mixed addition) against the pair's "
                             "measured issue rate at 4 waves/SIMD"}
        out = {
            "metric": "G1 MSM throughput (scalar-muls/sec)" if group == 1 else "G2 MSM throughput (scalar-muls/sec)",
            "value": value,
            "unit": "scalar-muls/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.curve} G{group} MSM, 2^{args.log2n} points per GPU, bases (i+1)G affine resident "
                            f"in HBM, uniform random scalars in [0,r) (Montgomery residues as libff holds them)",
                "points_per_gpu": n,
                "total_points": n * world,
                "window_bits": plan["c"],
                "num_windows": plan["num_windows"],
                "parallelism": f"range-sharded x{world}, all-gather of partial points + local sum",
                "msms_in_flight": msm.depth,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k_accumulate (bucket accumulation)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_traffic(args.curve, group, args.log2n, plan["c"]),
                "algorithmic_bytes_per_launch": algo_bytes,
                "kernel_ms": acc,
                "note": "integer-ALU bound path (no MFMA); HBM fraction is small by construction",
                "mac_issue": mac_issue,
            },
            "phases_ms": mean_phase,
        }
        if pipelined is not None:
            out["pipelined"] = pipelined
        if precomputed is not None:
            out["precomputed"] = precomputed
        if also is not None:
            out["also"] = also
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(curve, group, min(args.cpu_sample_log2n, args.log2n))
            except Exception as e:  # the baseline is a reported extra; never lose the GPU line to it
                out["cpu_baseline"] = {"value": None, "unit": "scalar-muls/s", "cores": 0, "kind": "port",
                                       "sample": f"failed: {e!r}"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
