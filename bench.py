#!/usr/bin/env python3
"""Benchmark of the hot path: G1 multi-scalar multiplication on MI355X.

  python bench.py --gpus N --steps K --warmup W [--log2n L] [--curve alt_bn128]

One "step" = one full MSM (libff::multi_exp) over synthetic (scalar, base) pairs that are
already resident in HBM.

N = 1   `value` = BASELINE configs[1]: alt_bn128 G1, 2^20 points, one MSM after the other.
        Inside `config.legs` (the driver keeps `config`): the 2^26-point size of the metric,
        the end-to-end (scalars over PCIe, resident bases) and host-entry (everything from host
        memory) timings of SURVEY.md §8(d), back-to-back MSMs in flight, precomputed multiples.
N > 1   one rank per GPU (started by torch.distributed.run -- by the caller, or by this script
        itself when WORLD_SIZE is unset).  `value` = BASELINE configs[3]: alt_bn128 G1, 2^26 points
        in TOTAL, range-sharded (multiexp.tcc:663-687 with rank == chunk): every rank reduces
        its 2^26/N points to one partial point, one RCCL all-gather of the partials, local sum
        -> "scaling": "strong".  `config.legs` adds configs[4] (bw6_761 G1 + bls12_377 G2, 2^24
        points in total each, issued together on two streams per rank), the weak-scaling figure
        (2^20 points per GPU) and the same 2^26 MSM on rank 0 alone.

The driver's record keeps the scalar keys of `config` and `roofline` and drops nested objects, so every figure the metric is
quoted on is ALSO a scalar key of `config`: ms_per_step_2p26 / value_2p26 / sort_ms_2p26 / accumulate_ms_2p26 / mac_issue_frac_2p26
(the 2^26 half of the metric), ms_per_step_<leg> / value_<leg> for the R32 bases (the reference profiler's input shape), configs[2],
the configs[4] shards and totals, the per-rank shards of configs[3] (2^23 / 2^24 / 2^25) and predicted_efficiency_{2,4,8}gpu (whole
input / (N x (shard + 0.05 ms exchange)): a prediction from one GPU, never a measurement of N).

The JSON line also carries
  roofline      bucket-accumulation kernel (dominant): algorithmic bytes per launch
                (96 B per alt_bn128 G1 point x points per launch) over its
                mean duration from HIP events on the launch stream, against the 8 TB/s
                HBM peak.  The path is integer-ALU bound; the fraction is small by nature.
  cpu_baseline  libff's own multi_exp (oracle/_ref, "reference") or the C restatement
                ("port") timed on the host cores of this box.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CURVES = {"alt_bn128": 0, "bls12_377": 1, "bw6_761": 2, "bls12_381": 3}
# one scalar + one affine base per scalar-mul
ALGO_BYTES = {(0, 1): 96, (1, 1): 128, (1, 2): 224, (2, 1): 240, (0, 2): 160, (2, 2): 240, (3, 1): 128, (3, 2): 224}
HBM_PEAK_GBS = 8000.0
# What bounds k_accumulate is the issue rate of its integer multiply-accumulates (tools/ubench.hip,
# profiles/r03_ubench_instruction_rates.txt, 4 waves per SIMD):
#   * prime-field groups (rr.cuh, reduced radix): one v_mad_i64_i32 per limb product, 30.27 T lane-instructions/s.
#     A mixed addition is 7 products of 2 L^2, 2 squarings of L (L + 1) / 2 + L^2 and one fused sum of two products
#     with one reduction (3 L^2) on L limbs of 28 / 29 bits.
#   * Fq2 groups: the same on lane pairs (rr.cuh Rr2H), see rr_mads_per_madd.
#   (The 32-bit loop of rounds 1-2 paid a v_mad_u64_u32 + v_addc_co_u32 pair per limb product, 33.15 T lane-instr/s.)
MAD_I64_PEAK = 30.27e12
FQ_LIMBS = {0: 8, 1: 12, 2: 24, 3: 12}
RR_LIMBS = {0: 9, 1: 14, 2: 28, 3: 14}   # rr_shape<Fq>::L


def rr_mads_per_madd(curve, group):
    """v_mad_i64_i32 issues of one mixed addition in k_accumulate (rr.cuh), summed over the lanes that share it"""
    L = RR_LIMBS[curve]
    if group == 1 or curve == 2:   # coordinates in Fq: 7 products, 2 squarings, one fused sum of two products
        return 7 * 2 * L * L + 2 * (L * (L + 1) // 2 + L * L) + 3 * L * L
    # Fq2 over a lane pair, per lane: a product is a fused sum of two Fq products (3 L^2), a complex squaring (u^2 = -1)
    # one Fq product (2 L^2), Y3 one fused sum of four (5 L^2) where a column holds it, else two products
    per_lane = {0: 6 * 3 + 2 * 2 + 2 * 3, 1: 8 * 3 + 2 * 3, 3: 6 * 3 + 2 * 2 + 5}[curve] * L * L
    return 2 * per_lane


FR_MODULUS = {
    0: 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001,
    1: 0x12AB655E9A2CA55660B44D1E5C37B00159AA76FED00000010A11800000000001,
    2: 0x1AE3A4617C510EAC63B05C06CA1493B1A22D9F300F5138F1EF3622FBA094800170B5D44300000008508C00000000001,
    3: 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001,
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--log2n", type=int, default=int(os.environ.get("AMDMSM_BENCH_LOG2N", "0")),
                    help="points of the `value` workload: per GPU at N = 1 (default 20), in TOTAL at N > 1 (default 26)")
    ap.add_argument("--curve", default="alt_bn128", choices=sorted(CURVES))
    ap.add_argument("--group", type=int, default=1, choices=(1, 2))
    ap.add_argument("--window-bits", type=int, default=0)
    ap.add_argument("--endomorphism", type=int, default=0, choices=(-1, 0, 1, 2),
                    help="amdmsm_opts.endomorphism of the `value` workload: 0 = the split only where the whole curve group "
                         "has order r (alt_bn128 G1), 1 = permitted (the synthetic bases are multiples of the generator)")
    ap.add_argument("--cpu-log2n", type=int, default=20, help="size of the cpu_baseline workload")
    ap.add_argument("--extra-log2n", type=int, default=int(os.environ.get("AMDMSM_BENCH_EXTRA_LOG2N", "26")),
                    help="N = 1: also time this size (config.legs.points_2pXX); 0 disables")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="only the `value` workload (profiling runs)")
    ap.add_argument("--pipeline", type=int, default=int(os.environ.get("AMDMSM_BENCH_PIPELINE", "1")),
                    help="MSMs in flight in the timed region (1 = strictly one after the other, which is "
                         "what `value` and the roofline kernel timings are quoted on)")
    ap.add_argument("--also-pipelined", type=int, default=3,
                    help="N = 1: also measure the same workload with this many MSMs in flight; 0 disables")
    ap.add_argument("--batch", type=int, default=4,
                    help="N = 1: also time this many MSMs handed over as one batch (amdmsm_msm_device_batch); 0 / 1 disables")
    ap.add_argument("--precomputed-c", type=int, default=16,
                    help="N = 1: also time the precomputed-multiples MSM (multi_exp_stream_with_precompute's algorithm "
                         "on an HBM-resident table of [2^(jc)]P) with this window size; 0 disables")
    ap.add_argument("--no-next-rows", action="store_true",
                    help="N = 1: skip the batch_exp / multi_exp_stream / FFI legs (SURVEY.md section 8(f) rows)")
    ap.add_argument("--no-r32", action="store_true", help="N = 1: skip the leg with the reference profiler's 32 repeated bases")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="N = 1: skip the configs[2] / configs[4] legs (bls12_377 G1 2^22; bw6_761 G1 + bls12_377 G2 at 2^21 and 2^24)")
    ap.add_argument("--config4-log2n", type=int, default=24, help="N > 1: total points of the configs[4] leg; 0 disables")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks as fresh child
    processes (this parent never touches the GPU) and pass rank 0's JSON line through."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    r = subprocess.run(cmd, env=env)
    sys.exit(r.returncode)


ARGS = None
if __name__ == "__main__":
    ARGS = parse_args()
    if ARGS.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(ARGS)   # does not return

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import libff_amd  # noqa: E402
from libff_amd.distributed import ShardedMsm, shard_range  # noqa: E402


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
    (profiles/r*_pmc*.json: separate FETCH_SIZE / WRITE_SIZE runs of this very command,
    gfx950 correction 2*FETCH + WRITE, see the file), or None when no pass matches the workload."""
    import glob

    best, src = None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*pmc*.json"))):
        try:
            d = json.load(open(path))
            wl = d["workload"]
            if (wl["curve"], wl["group"], wl["log2n"], wl["window_bits"]) != (curve_name, group, log2n, window_bits):
                continue
            k = d["kernels"]["k_accumulate"]
            best = (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0
            src = os.path.relpath(path, ROOT)
        except Exception:
            continue
    return best, src


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(curve, group, log2n):
    """libff's CPU path on this box, on the headline workload (2^log2n points, SHA512_rng scalars,
    bases (i+1)G in special form): multi_exp<BDLO12_signed, special> with chunks = 1 (what the
    reference's own profiler measures, profile_multiexp.cpp:184-207) and the best of
    chunks in {cores/4, cores/2, cores} (OpenMP over ranges, multiexp.tcc:667-679) after a
    warm-up call that spins the thread pool up.  oracle/_ref/libff_ref.so = the reference
    itself ("reference"); the C restatement otherwise ("port")."""
    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    n = 1 << log2n
    from oracle import ref

    if ref.available():
        kind, be = "reference", ref
        be.lib()

        def run(b, s, chunks):
            _, secs = be.multi_exp(curve, group, b, s, be.BDLO12_SIGNED, be.FORM_SPECIAL, chunks=chunks, want_time=True)
            return secs
    else:
        from oracle import port

        kind, be = "port", port
        be.build()

        def run(b, s, chunks):
            t0 = time.perf_counter()
            be.multi_exp(curve, group, b, s, be.BDLO12_SIGNED, be.FORM_SPECIAL, chunks=chunks, omp=chunks > 1)
            return time.perf_counter() - t0

    bases = be.bases_seq(curve, group, n)
    scalars = be.scalars_sha512(curve, 0, n)
    run(bases[:1 << 14], scalars[:1 << 14], cores)   # OpenMP warm-up
    multi = {}
    for chunks in sorted({max(1, cores // 4), max(1, cores // 2), cores}):
        multi[chunks] = min(run(bases, scalars, chunks) for _ in range(2))
    best = min(multi, key=multi.get)
    # chunks = 1: one core, the whole workload (about 7 s for 2^20 points on this class of host)
    n1 = n if n <= (1 << 20) else (1 << 20)
    t1 = run(bases[:n1], scalars[:n1], 1)
    return {"value": n / multi[best], "unit": "scalar-muls/s", "cores": min(best, cores), "kind": kind,
            "cpu": cpu_model(), "host_cores": cores,
            "sample": f"2^{log2n} (scalar, base) pairs of the workload family (SHA512_rng scalars, bases (i+1)G, special "
                      f"form), multi_exp<BDLO12_signed, special>; chunks={best} (OpenMP), best of "
                      f"{ {c: round(t, 3) for c, t in multi.items()} } s after a warm-up call",
            "single_core": {"value": n1 / t1, "unit": "scalar-muls/s", "cores": 1,
                            "sample": f"chunks=1 (profile_multiexp.cpp:184-207) on the first {n1} pairs, {t1:.2f} s"}}


class Timer:
    """barrier + synchronize on both sides, max over ranks (the bench contract)."""

    def __init__(self, world, dev):
        self.world, self.dev = world, dev

    def fence(self):
        torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(self, seconds):
        if self.world > 1:
            tt = torch.tensor([seconds], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else self.dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt.item())
        return seconds


def gen_inputs(eng, curve, group, first, n, dev, seed):
    sz = libff_amd.sizes(curve, group)
    bases = torch.empty((max(n, 1), sz["affine_bytes"] // 8), dtype=torch.int64, device=dev)
    eng.gen_bases_seq_device(curve, group, first, n, bases.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    scalars = random_scalars(curve, max(n, 1), dev, seed)
    torch.cuda.synchronize()
    return bases, scalars


def timed_msm(tm, eng, msm, bases, scalars, n, steps, warmup, window_bits=0, want_phases=True):
    """W warm-up steps, then exactly K timed steps; returns (seconds, per-step phase dicts)."""
    for _ in range(max(warmup, msm.depth if warmup else 0)):
        msm.run(bases, scalars, n, libff_amd.OUT_LIBFF, window_bits=window_bits)
    msm.synchronize()
    tm.fence()
    # Steps are enqueued back to back (depth 1: on ONE stream, so the MSMs still run strictly one
    # after the other); each timed call gets a ticket and the HIP-event phase times of the timed
    # region are read after it, so no host synchronisation sits between two steps.
    phases, tickets = [], []
    t0 = time.perf_counter()
    for _ in range(steps):
        msm.run(bases, scalars, n, libff_amd.OUT_LIBFF, window_bits=window_bits)
        if want_phases:
            tickets.append(eng.last_timing_ticket())
    msm.synchronize()
    tm.fence()
    elapsed = tm.max_over_ranks(time.perf_counter() - t0)
    for tk in tickets[-60:]:   # the engine keeps the last 64 tickets
        phases.append(eng.get_timings(ticket=tk))
    return elapsed, phases


def roofline_of(curve_name, curve, group, n_launch, plan, acc_ms, log2n_for_pmc):
    algo_bytes = ALGO_BYTES[(curve, group)] * n_launch
    achieved = algo_bytes / (acc_ms * 1e-3) / 1e9
    # list entries: one per (digit column, window); the endomorphism split has two half-length columns per point
    columns = 2 if plan.get("endomorphism") else 1
    entries = float(n_launch) * columns * plan["num_windows"]
    lane_instr = entries * rr_mads_per_madd(curve, group)
    mac_peak = MAD_I64_PEAK
    mac_what = ("v_mad_i64_i32 issues of the reduced-radix Montgomery products in k_accumulate (every list entry counted as "
                f"a full mixed addition: {rr_mads_per_madd(curve, group)} multiply-accumulates on limbs of {RR_LIMBS[curve]} x "
                f"{29 if RR_LIMBS[curve] == 9 else 28} bits) against the instruction's measured issue rate at 4 waves/SIMD; the "
                "rest of the loop's issue time is its shifts, masks and limb-wise additions")
    mac_rate = lane_instr / (acc_ms * 1e-3)
    traffic, traffic_src = pmc_traffic(curve_name, group, log2n_for_pmc, plan["c"]) if log2n_for_pmc else (None, None)
    return {
        "bound": "hbm",
        "kernel": "k_accumulate (bucket accumulation)",
        "achieved": achieved,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS,
        "traffic": traffic,
        "traffic_source": (f"{traffic_src}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload committed with the "
                           "build, (2*FETCH + WRITE) * 1024 per k_accumulate launch; not re-measured in this run")
        if traffic_src else None,
        "algorithmic_bytes_per_launch": algo_bytes,
        "kernel_ms": acc_ms,
        "note": "integer-ALU bound path (no MFMA); HBM fraction is small by construction",
        "mac_issue": {"achieved": mac_rate / 1e12, "peak": mac_peak / 1e12, "unit": "T lane-instr/s",
                      "frac": mac_rate / mac_peak, "what": mac_what},
    }


def baseline_tag(curve_name, group, log2n):
    return {("alt_bn128", 1, 20): " (BASELINE configs[1])", ("bls12_377", 1, 22): " (BASELINE configs[2])",
            ("alt_bn128", 1, 26): " (the north-star target size)", ("bw6_761", 1, 21): " (per-rank shard of BASELINE configs[4])",
            ("bls12_377", 2, 21): " (per-rank shard of BASELINE configs[4])"}.get((curve_name, group, log2n), "")


def mean_phases(phases):
    return {k: float(np.mean([p[k] for p in phases])) for k in phases[0]} if phases else {}


def other_config_legs(args, tm, dev, device_index):
    """The other single-GPU-sized BASELINE configurations, each with its phase times and its own
    roofline (the reference's profiler runs G1 and G2 in one pass, profile_multiexp.cpp:401-414):
      configs[2]   bls12_377 G1, 2^22 points;
      configs[4]   bw6_761 G1 and bls12_377 G2: the per-rank shard of the 8-GPU run (2^21 points each),
                   alone and issued together on two contexts (what a rank of `--gpus 8` does), and the
                   2^24-point totals alone on this one GPU.
    Bases (i+1)G are multiples of the generator, so the endomorphism split is permitted explicitly for
    configs[4] (amdmsm_opts.endomorphism = 1), as in the N > 1 leg; configs[2] runs with the default."""
    legs = {}
    k = max(2, min(args.steps, 3))

    def alone(eng, cname, group, log2n, endo, tag, what):
        curve = CURVES[cname]
        n = 1 << log2n
        eng.endomorphism = endo
        eng.set_timing(True)
        plan = libff_amd.plan(curve, group, n, endomorphism=endo)
        b, s = gen_inputs(eng, curve, group, 0, n, dev, 31 + log2n)
        msm = ShardedMsm(eng, curve, group, depth=1)
        e, ph = timed_msm(tm, eng, msm, b, s, n, k, 1)
        acc = float(np.mean([p["accumulate_ms"] for p in ph]))
        legs[tag] = {"workload": what, "steps": k, "value": n * k / e, "unit": "scalar-muls/s", "ms_per_step": e / k * 1e3,
                     "window_bits": plan["c"], "num_windows": plan["num_windows"], "endomorphism_split": plan["endomorphism"],
                     "phases_ms": mean_phases(ph), "roofline": roofline_of(cname, curve, group, n, plan, acc, log2n)}
        return b, s, n

    eng = libff_amd.Engine(device_index)
    for lg in (23, 24, 25):   # configs[3]: what one rank of the 8 / 4 / 2-GPU run does
        b, s, n = alone(eng, "alt_bn128", 1, lg, 0, f"shard_alt_bn128_g1_2p{lg}",
                        f"BASELINE configs[3], per-rank shard alone: alt_bn128 G1 MSM, 2^{lg} points (2^26 over {1 << (26 - lg)} GPUs)")
        del b, s
        torch.cuda.empty_cache()
    b, s, n = alone(eng, "bls12_377", 1, 22, 0, "config2_bls12_377_g1_2p22",
                    "BASELINE configs[2]: bls12_377 G1 MSM (384-bit field), 2^22 points, 1 GPU")
    del b, s
    torch.cuda.empty_cache()
    # configs[4], per-rank shard
    eng2 = libff_amd.Engine(device_index)
    b1, s1, n1 = alone(eng, "bw6_761", 1, 21, 1, "config4_shard_bw6_761_g1_2p21",
                       "BASELINE configs[4], per-rank shard alone: bw6_761 G1 MSM, 2^21 points (2^24 over 8 GPUs)")
    b2, s2, n2 = alone(eng2, "bls12_377", 2, 21, 1, "config4_shard_bls12_377_g2_2p21",
                       "BASELINE configs[4], per-rank shard alone: bls12_377 G2 MSM, 2^21 points (2^24 over 8 GPUs)")
    jobs = [(ShardedMsm(eng, 2, 1, depth=1), b1, s1, n1), (ShardedMsm(eng2, 1, 2, depth=1), b2, s2, n2)]

    def both():
        for m, bb, ss, nn in jobs:   # asynchronous: the two MSMs are in flight together
            m.run(bb, ss, nn, libff_amd.OUT_LIBFF)

    both()
    for m, _, _, _ in jobs:
        m.synchronize()
    tm.fence()
    t0 = time.perf_counter()
    for _ in range(k):
        both()
    for m, _, _, _ in jobs:
        m.synchronize()
    tm.fence()
    e = time.perf_counter() - t0
    alone_ms = legs["config4_shard_bw6_761_g1_2p21"]["ms_per_step"] + legs["config4_shard_bls12_377_g2_2p21"]["ms_per_step"]
    legs["config4_shard_pair_together"] = {
        "workload": "BASELINE configs[4], what one rank of the 8-GPU run does: bw6_761 G1 2^21 + bls12_377 G2 2^21 issued together "
                    "on two contexts (two streams); a step = both MSMs", "steps": k, "value": (n1 + n2) * k / e,
        "unit": "scalar-muls/s", "ms_per_step": e / k * 1e3, "one_after_the_other_ms": alone_ms}
    del jobs, b1, s1, b2, s2
    torch.cuda.empty_cache()
    # configs[4], the stated totals on one GPU
    b, s, n = alone(eng, "bw6_761", 1, 24, 1, "config4_total_bw6_761_g1_2p24",
                    "BASELINE configs[4] at its stated total on ONE GPU: bw6_761 G1 MSM, 2^24 points")
    del b, s
    torch.cuda.empty_cache()
    b, s, n = alone(eng2, "bls12_377", 2, 24, 1, "config4_total_bls12_377_g2_2p24",
                    "BASELINE configs[4] at its stated total on ONE GPU: bls12_377 G2 MSM, 2^24 points")
    del b, s
    torch.cuda.empty_cache()
    eng.close()
    eng2.close()
    return legs


def next_row_legs(args, eng, dev):
    """SURVEY.md section 8(f) rows through the same C ABI, one figure each (tools/bench_next.py has the long form with
    the reference beside it): fixed-base batch_exp, multi_exp_stream from a file, the FFI entry with its validation."""
    import ctypes
    import tempfile

    legs = {}
    curve, group = 0, 1
    sz = libff_amd.sizes(curve, group)
    n = 1 << 20
    # batch_exp / get_window_table (multiexp.tcc:809-947), window 17 = alt_bn128_G1::fixed_base_exp_window_table at 2^20
    g = eng.gen_bases_seq(curve, group, 1, first=0)[0]
    v = np.ascontiguousarray(random_scalars(curve, n, dev, 77).cpu().numpy()).view(np.uint64)
    res = np.zeros((n, sz["g_bytes"] // 8), dtype=np.uint64)
    eng.batch_exp(curve, group, sz["fr_bits"], 17, g, v, out=res)
    first = eng.batch_exp_timings()
    t0 = time.perf_counter()
    eng.batch_exp(curve, group, sz["fr_bits"], 17, g, v, out=res)
    dt = time.perf_counter() - t0
    del res
    again = eng.batch_exp_timings()
    legs["batch_exp"] = {"workload": "alt_bn128 G1 batch_exp, 2^20 scalars, window 17 (15 x 2^17-entry table), host vectors in and out",
                         "value": n / dt, "unit": "exponentiations/s", "ms_per_call": dt * 1e3,
                         "device_ms": {"window_table_first_call": first["table_ms"], "exponentiations": again["exp_ms"],
                                       "scalars_h2d": again["h2d_ms"], "results_d2h": again["d2h_ms"]},
                         "exponentiations_per_s_device": n / (again["exp_ms"] * 1e-3)}
    # multi_exp_stream (multiexp_stream.tcc:164-191): 2^20 on-disk records (binary, Montgomery, uncompressed) from the page cache
    aff = torch.from_numpy(eng.gen_bases_seq(curve, group, n, first=0)[:, : sz["affine_bytes"] // 8].astype(np.int64)).contiguous()
    cl = sz["affine_bytes"] // 16
    rec = aff.view(n, 2, cl).flip(2).contiguous().view(torch.uint8).view(n, 2, cl, 8).flip(3).contiguous()   # big-endian coordinates
    with tempfile.NamedTemporaryFile(dir=os.environ.get("TMPDIR", "/tmp"), suffix=".bases", delete=False) as f:
        f.write(rec.numpy().tobytes())
        path = f.name
    eng.multi_exp_stream_file(curve, group, path, v)
    t0 = time.perf_counter()
    eng.multi_exp_stream_file(curve, group, path, v)
    dt = time.perf_counter() - t0
    os.unlink(path)
    legs["multi_exp_stream_file"] = {"workload": "alt_bn128 G1 multi_exp_stream, 2^20 records of 64 B read from a file (page cache), "
                                                 "scalars from host memory", "value": n / dt, "unit": "scalar-muls/s",
                                     "ms_per_call": dt * 1e3, "file_GB_per_s": n * sz["affine_bytes"] / dt / 1e9}
    # <curve>_g1_multiexp (include/libff_amd_ffi.h): big-endian plain inputs, every element validated on the device
    for cname, cv, m in (("bls12_377", 1, 1 << 20), ("bw6_761", 2, 1 << 17)):
        s1 = libff_amd.sizes(cv, 1)
        fl = s1["affine_bytes"] // 16
        am = np.ascontiguousarray(eng.gen_bases_seq(cv, 1, m, first=5)[:, : 2 * fl]).reshape(2 * m, fl)
        one = np.zeros_like(am)
        one[:, 0] = 1
        plain = eng.field_op(cv, 1, 0, am, one)                       # x * 1 * R^-1: out of Montgomery form
        bb = np.ascontiguousarray(np.ascontiguousarray(plain[:, ::-1]).view(np.uint8).reshape(2 * m, fl, 8)[..., ::-1]).reshape(-1)
        sp = np.ascontiguousarray(random_scalars(cv, m, dev, 78).cpu().numpy()).view(np.uint64)   # read as plain integers < r
        sb = np.ascontiguousarray(np.ascontiguousarray(sp[:, ::-1]).view(np.uint8).reshape(m, -1, 8)[..., ::-1]).reshape(-1)
        o = np.zeros(s1["affine_bytes"], dtype=np.uint8)
        fn = getattr(eng.lib, f"{cname}_g1_multiexp")
        fn.restype = ctypes.c_bool
        eng.lib.amdmsm_ffi_last_timings.restype = ctypes.c_bool

        def call():
            t0 = time.perf_counter()
            ok = fn(bb.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(bb.size), sb.ctypes.data_as(ctypes.c_void_p),
                    ctypes.c_size_t(sb.size), o.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(o.size))
            return bool(ok), time.perf_counter() - t0

        ok, _ = call()
        ok2, dt = call()
        ms = (ctypes.c_float * 3)()
        if not (ok and ok2 and eng.lib.amdmsm_ffi_last_timings(ms)):
            legs[f"ffi_{cname}_g1_multiexp"] = {"value": None, "note": "call failed"}
            continue
        legs[f"ffi_{cname}_g1_multiexp"] = {
            "workload": f"{cname}_g1_multiexp, {m} big-endian plain (point, scalar) pairs from host memory, every point checked "
                        "(range, curve equation, subgroup) on the device before the MSM", "value": m / dt, "unit": "scalar-muls/s",
            "ms_per_call": dt * 1e3, "device_ms": {"inputs_h2d": ms[0], "decode_and_validate": ms[1], "msm_and_encode": ms[2]}}
    return legs


def single_gpu(args, tm, eng, dev, curve, group):
    log2n = args.log2n or 20
    n = 1 << log2n
    plan = libff_amd.plan(curve, group, n, args.window_bits, endomorphism=args.endomorphism)
    sz = libff_amd.sizes(curve, group)
    bases, scalars = gen_inputs(eng, curve, group, 0, n, dev, 1234)
    msm = ShardedMsm(eng, curve, group, depth=max(1, args.pipeline))
    eng.set_timing(True)
    elapsed, phases = timed_msm(tm, eng, msm, bases, scalars, n, args.steps, args.warmup, args.window_bits)
    acc = float(np.mean([p["accumulate_ms"] for p in phases]))
    legs = {}

    if not args.no_legs and args.also_pipelined > 1 and msm.depth == 1:
        pm = ShardedMsm(eng, curve, group, depth=args.also_pipelined)
        kp = max(args.steps, 2 * pm.depth)
        ep, _ = timed_msm(tm, eng, pm, bases, scalars, n, kp, pm.depth, args.window_bits, want_phases=False)
        legs["pipelined"] = {"msms_in_flight": pm.depth, "steps": kp, "value": n * kp / ep, "unit": "scalar-muls/s",
                             "ms_per_step": ep / kp * 1e3,
                             "note": "consecutive MSMs on alternating streams / workspace slots: the few-wave tail of one "
                                     "overlaps the bulk kernels of the next; `value` stays the one-at-a-time figure"}
        msm = ShardedMsm(eng, curve, group, depth=1)

    # ---- the reference profiler's own input shape: 32 distinct points repeated (profile_multiexp.cpp:14-15, 24-50) ----
    if not args.no_legs and not args.no_r32:
        first = torch.randint(0, 1 << 62, (32,), generator=torch.Generator().manual_seed(32)).tolist()
        p32 = torch.empty((32, sz["affine_bytes"] // 8), dtype=torch.int64, device=dev)
        for j, f in enumerate(first):   # P_j = (f_j + 1) G: 32 unrelated multiples of the generator
            eng.gen_bases_seq_device(curve, group, f, 1, p32[j].data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        bases_r32 = p32.repeat(n // 32, 1).contiguous()
        er, phr = timed_msm(tm, eng, msm, bases_r32, scalars, n, args.steps, args.warmup, args.window_bits)
        acc_r = float(np.mean([p["accumulate_ms"] for p in phr]))
        legs["r32_bases"] = {
            "workload": f"{args.curve} G{group} MSM, 2^{log2n} points, bases = 32 distinct points repeated (the reference "
                        "profiler's own input shape, profile_multiexp.cpp:14-15,24-50): equal and opposite points meet in "
                        "the buckets all the time", "steps": args.steps, "value": n * args.steps / er, "unit": "scalar-muls/s",
            "ms_per_step": er / args.steps * 1e3, "phases_ms": mean_phases(phr),
            "roofline": roofline_of(args.curve, curve, group, n, plan, acc_r, 0)}
        del bases_r32, p32

    # ---- k MSMs handed over as ONE batch (amdmsm_msm_device_batch): the tails of all of them run as one set of kernels ----
    if not args.no_legs and args.batch > 1:
        kb = args.batch
        outs = [torch.zeros(sz["g_bytes"] // 8, dtype=torch.int64, device=dev) for _ in range(kb)]
        scs = [scalars] + [random_scalars(curve, n, dev, 2000 + j) for j in range(1, kb)]
        bs = [bases] + [gen_inputs(eng, curve, group, 77 * j, n, dev, 1)[0] for j in range(1, kb)]
        st_b = torch.cuda.Stream(dev)
        torch.cuda.synchronize()

        def batch_call():
            eng.msm_device_batch(curve, group, [b.data_ptr() for b in bs], [x.data_ptr() for x in scs], n,
                                 [o.data_ptr() for o in outs], out_form=libff_amd.OUT_LIBFF, window_bits=args.window_bits,
                                 stream=st_b.cuda_stream)

        for _ in range(2):
            batch_call()
        st_b.synchronize()
        tm.fence()
        reps = max(args.steps // kb, 3)
        t0 = time.perf_counter()
        for _ in range(reps):
            batch_call()
        phb = eng.get_timings()
        st_b.synchronize()
        tm.fence()
        eb = time.perf_counter() - t0
        legs["batched"] = {"msms_per_call": kb, "calls": reps, "value": n * kb * reps / eb, "unit": "scalar-muls/s",
                           "ms_per_msm": eb / (kb * reps) * 1e3, "ms_per_call": eb / reps * 1e3, "phases_ms_last_call": phb,
                           "note": f"{kb} MSMs of 2^{log2n} points (different bases and scalars) in one amdmsm_msm_device_batch call: "
                                   "sort and accumulation MSM after MSM, fix-up / bucket reduction / Horner once over the windows "
                                   "of all of them; `value` stays the one-at-a-time figure"}
        del outs, scs, bs

    # ---- SURVEY §8(d) "end-to-end": scalars cross PCIe inside the timed region, bases resident
    #      (amdmsm_register_bases), result back on the host; and the plain host entry where the
    #      bases (libff (X, Y, Z) records, 96 B) cross PCIe and are imported on every call
    if not args.no_legs:
        h_bases = eng.gen_bases_seq(curve, group, n, first=0)     # libff special-form records on the host
        h_scalars = np.ascontiguousarray(scalars.cpu().numpy()).view(np.uint64)

        def host_calls(k):
            for _ in range(2):
                eng.multi_exp(curve, group, h_bases, h_scalars, base_form=libff_amd.multi_exp_base_form_special,
                              out_form=libff_amd.OUT_LIBFF)
            t0 = time.perf_counter()
            for _ in range(k):
                eng.multi_exp(curve, group, h_bases, h_scalars, base_form=libff_amd.multi_exp_base_form_special,
                              out_form=libff_amd.OUT_LIBFF)
            return (time.perf_counter() - t0) / k

        kh = max(args.steps, 5)
        t_host = host_calls(kh)
        handle = eng.register_bases(curve, group, h_bases, libff_amd.multi_exp_base_form_special)
        t_e2e = host_calls(kh)
        eng.unregister_bases(handle)
        legs["end_to_end"] = {"value": n / t_e2e, "unit": "scalar-muls/s", "ms_per_step": t_e2e * 1e3, "steps": kh,
                              "what": "amdmsm_multi_exp on host vectors with the bases registered (resident in HBM): "
                                      f"{n * sz['fr_bytes'] >> 20} MiB of scalars H2D from pageable memory + MSM + result D2H "
                                      "per step -- the proving-key use case"}
        legs["host_entry"] = {"value": n / t_host, "unit": "scalar-muls/s", "ms_per_step": t_host * 1e3, "steps": kh,
                              "what": "amdmsm_multi_exp, everything from pageable host memory on every call: "
                                      f"{n * sz['g_bytes'] >> 20} MiB of (X, Y, Z) bases + {n * sz['fr_bytes'] >> 20} MiB of "
                                      "scalars H2D, base import, MSM (bases travel while the scalars are sorted)"}
        del h_bases, h_scalars

    # ---- fixed bases with precomputed multiples (multi_exp_stream_with_precompute, HBM-resident) ----
    if not args.no_legs and args.precomputed_c:
        pc = args.precomputed_c
        D = libff_amd.precompute_num_digits(curve, pc)
        stream = torch.cuda.current_stream().cuda_stream
        table = torch.empty((n * D, sz["affine_bytes"] // 8), dtype=torch.int64, device=dev)
        tb = time.perf_counter()
        eng.precompute_bases_device(curve, group, bases.data_ptr(), n, pc, D, table.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        build_s = time.perf_counter() - tb
        out_pt = torch.zeros(sz["g_bytes"] // 8, dtype=torch.int64, device=dev)
        for _ in range(2):
            eng.msm_precomputed_device(curve, group, table.data_ptr(), scalars.data_ptr(), n, pc, D, out_pt.data_ptr())
        eng.synchronize()
        tm.fence()
        kq = max(args.steps, 4)
        tq = time.perf_counter()
        for _ in range(kq):
            eng.msm_precomputed_device(curve, group, table.data_ptr(), scalars.data_ptr(), n, pc, D, out_pt.data_ptr())
        phq = eng.get_timings()
        eng.synchronize()
        tm.fence()
        eq = time.perf_counter() - tq
        legs["precomputed"] = {"window_bits": pc, "multiples_per_base": D, "table_gib": table.numel() * 8 / 2**30,
                               "table_build_ms": build_s * 1e3, "steps": kq, "value": n * kq / eq,
                               "unit": "scalar-muls/s", "ms_per_step": eq / kq * 1e3, "phases_ms_last": phq,
                               "note": "libff's multi_exp_stream_with_precompute algorithm (one bucket set over the table "
                                       "[2^(jc)]P_i, no doublings) with the table resident in HBM; a different reference "
                                       "entry point than `value`'s multi_exp"}
        del table

    # ---- the other size of the metric (2^26), outside the main timed region ----
    roof2 = None
    if not args.no_legs and args.extra_log2n and args.extra_log2n != log2n:
        n2 = 1 << args.extra_log2n
        del bases, scalars
        torch.cuda.empty_cache()
        bases2, scalars2 = gen_inputs(eng, curve, group, 0, n2, dev, 4321)
        k2 = 4
        e2, ph2 = timed_msm(tm, eng, msm, bases2, scalars2, n2, k2, 1)
        p2 = libff_amd.plan(curve, group, n2, endomorphism=args.endomorphism)
        acc2 = float(np.mean([p["accumulate_ms"] for p in ph2]))
        legs[f"points_2p{args.extra_log2n}"] = {
            "workload": f"{args.curve} G{group} MSM, 2^{args.extra_log2n} points on one GPU (the north-star target size)",
            "steps": k2, "value": n2 * k2 / e2, "unit": "scalar-muls/s", "ms_per_step": e2 / k2 * 1e3,
            "window_bits": p2["c"], "num_windows": p2["num_windows"], "endomorphism_split": p2["endomorphism"],
            "phases_ms": mean_phases(ph2)}
        roof2 = roofline_of(args.curve, curve, group, n2, p2, acc2, args.extra_log2n)
        del bases2, scalars2
        torch.cuda.empty_cache()

    # ---- the adjacent entry points ----
    if not args.no_legs and not args.no_next_rows and (args.curve, group) == ("alt_bn128", 1):
        try:
            legs["next_rows"] = next_row_legs(args, eng, dev)
        except Exception as e:   # reported extras: never lose the headline to them
            legs["next_rows"] = {"failed": repr(e)}

    # ---- the other BASELINE configurations that fit one GPU (configs[2], configs[4]) ----
    if not args.no_legs and not args.no_other_configs and (args.curve, group) == ("alt_bn128", 1):
        legs.update(other_config_legs(args, tm, dev, eng.device))

    roof = roofline_of(args.curve, curve, group, n, plan, acc, log2n)
    # The driver's record keeps scalar keys of `config` / `roofline` only (nested objects are dropped): the other half of
    # the metric (2^26), the phase times and the other BASELINE configurations are repeated here as scalars.
    mp = mean_phases(phases)
    flat = {"sort_ms": mp.get("scatter_ms"), "accumulate_ms": mp.get("accumulate_ms"), "reduce_ms": mp.get("reduce_ms"),
            "final_ms": mp.get("final_ms"), "device_total_ms": mp.get("total_ms"), "mac_issue_frac": roof["mac_issue"]["frac"]}
    roof["mac_issue_frac"] = roof["mac_issue"]["frac"]
    xl = legs.get(f"points_2p{args.extra_log2n}")
    if xl and roof2 is not None:
        t = f"2p{args.extra_log2n}"
        flat.update({f"ms_per_step_{t}": xl["ms_per_step"], f"value_{t}": xl["value"], f"window_bits_{t}": xl["window_bits"],
                     f"num_windows_{t}": xl["num_windows"], f"sort_ms_{t}": xl["phases_ms"].get("scatter_ms"),
                     f"accumulate_ms_{t}": xl["phases_ms"].get("accumulate_ms"), f"reduce_ms_{t}": xl["phases_ms"].get("reduce_ms"),
                     f"final_ms_{t}": xl["phases_ms"].get("final_ms"), f"mac_issue_frac_{t}": roof2["mac_issue"]["frac"],
                     f"hbm_frac_{t}": roof2["frac"]})
        roof.update({f"kernel_ms_{t}": roof2["kernel_ms"], f"achieved_{t}": roof2["achieved"], f"frac_{t}": roof2["frac"],
                     f"mac_issue_frac_{t}": roof2["mac_issue"]["frac"], f"traffic_{t}": roof2["traffic"]})
    for key, tag in (("r32_bases", "r32"), ("pipelined", "pipelined"), ("end_to_end", "end_to_end"), ("host_entry", "host_entry"),
                     ("precomputed", "precomputed"), ("config2_bls12_377_g1_2p22", "config2_bls12_377_g1_2p22"),
                     ("config4_shard_bw6_761_g1_2p21", "config4_bw6_761_g1_2p21"),
                     ("config4_shard_bls12_377_g2_2p21", "config4_bls12_377_g2_2p21"),
                     ("config4_shard_pair_together", "config4_pair_together"),
                     ("config4_total_bw6_761_g1_2p24", "config4_bw6_761_g1_2p24"),
                     ("config4_total_bls12_377_g2_2p24", "config4_bls12_377_g2_2p24"),
                     ("shard_alt_bn128_g1_2p23", "shard_2p23")):
        leg = legs.get(key)
        if leg and leg.get("ms_per_step") is not None:
            flat[f"ms_per_step_{tag}"] = leg["ms_per_step"]
            flat[f"value_{tag}"] = leg["value"]
            if "roofline" in leg:
                flat[f"mac_issue_frac_{tag}"] = leg["roofline"]["mac_issue"]["frac"]
                flat[f"accumulate_ms_{tag}"] = leg["roofline"]["kernel_ms"]
    for lg in (24, 25):
        if f"shard_alt_bn128_g1_2p{lg}" in legs:
            flat[f"ms_per_step_shard_2p{lg}"] = legs[f"shard_alt_bn128_g1_2p{lg}"]["ms_per_step"]
    if xl and "shard_alt_bn128_g1_2p23" in legs and (args.curve, group, args.extra_log2n) == ("alt_bn128", 1, 26):
        # strong scaling of configs[3] as ONE GPU can predict it: whole input / (N x (shard + exchange)); the exchange
        # (all-gather of N partial points + k_sum_points) is taken as 0.05 ms -- not a measurement of N GPUs
        for lg, ng in ((25, 2), (24, 4), (23, 8)):
            leg = legs.get(f"shard_alt_bn128_g1_2p{lg}")
            if leg:
                flat[f"predicted_efficiency_{ng}gpu"] = xl["ms_per_step"] / (ng * (leg["ms_per_step"] + 0.05))
    if "batched" in legs:
        flat["ms_per_msm_batched"] = legs["batched"]["ms_per_msm"]
        flat["value_batched"] = legs["batched"]["value"]
    if roof2 is not None:
        roof[f"at_2p{args.extra_log2n}"] = {k: roof2[k] for k in ("achieved", "frac", "traffic", "traffic_source",
                                                                  "algorithmic_bytes_per_launch", "kernel_ms", "mac_issue")}
    out = {
        "metric": "G1 MSM throughput (scalar-muls/sec)" if group == 1 else "G2 MSM throughput (scalar-muls/sec)",
        "value": n * args.steps / elapsed,
        "unit": "scalar-muls/s",
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "config": {
            "workload": f"{args.curve} G{group} MSM, 2^{log2n} points, 1 GPU{baseline_tag(args.curve, group, log2n)}; bases (i+1)G affine "
                        f"resident in HBM, uniform random scalars in [0,r) (Montgomery residues as libff holds them)",
            "points_per_gpu": n,
            "total_points": n,
            "window_bits": plan["c"],
            "num_windows": plan["num_windows"],
            "endomorphism_split": plan["endomorphism"],
            "parallelism": "1 GPU",
            "msms_in_flight": msm.depth,
            **flat,
            "phases_ms": mean_phases(phases),
            "legs": legs,
        },
        "roofline": roof,
    }
    return out


def multi_gpu(args, tm, eng, dev, rank, world, curve, group):
    """configs[3]: 2^26 points in total, range-sharded; plus configs[4] and the weak-scaling leg."""
    log2n = args.log2n or 26
    total = 1 << log2n
    lo, hi = shard_range(total, world, rank)
    n = hi - lo
    plan = libff_amd.plan(curve, group, n, args.window_bits, endomorphism=args.endomorphism)
    bases, scalars = gen_inputs(eng, curve, group, lo, n, dev, 1234 + rank)   # bases (lo + i + 1) * G
    msm = ShardedMsm(eng, curve, group, depth=1)
    eng.set_timing(True)
    elapsed, phases = timed_msm(tm, eng, msm, bases, scalars, n, args.steps, args.warmup, args.window_bits)
    acc = float(np.mean([p["accumulate_ms"] for p in phases]))
    legs = {}
    # What the step should cost: this rank's shard MSM alone (HIP events around its kernels, max over
    # ranks) + the exchange (all-gather of the partial points + k_sum_points), timed on its own.
    shard_ms = tm.max_over_ranks(float(np.mean([p["total_ms"] for p in phases])) * 1e-3) * 1e3
    kx = 20
    for _ in range(3):
        msm.exchange_only(libff_amd.OUT_LIBFF)
    msm.synchronize()
    tm.fence()
    tx = time.perf_counter()
    for _ in range(kx):
        msm.exchange_only(libff_amd.OUT_LIBFF)
    msm.synchronize()
    tm.fence()
    exchange_ms = tm.max_over_ranks(time.perf_counter() - tx) / kx * 1e3

    # the same total on rank 0 alone (what one GPU does with the whole input), the others wait
    if not args.no_legs:
        del bases, scalars
        torch.cuda.empty_cache()
        if rank == 0:
            b1, s1 = gen_inputs(eng, curve, group, 0, total, dev, 99)
            st1 = torch.cuda.Stream(dev)
            o1 = torch.zeros(libff_amd.sizes(curve, group)["g_bytes"] // 8, dtype=torch.int64, device=dev)
            k1 = 3
            eng.msm_device(curve, group, b1.data_ptr(), s1.data_ptr(), total, o1.data_ptr(),
                           out_form=libff_amd.OUT_LIBFF, stream=st1.cuda_stream)
            st1.synchronize()
            t0 = time.perf_counter()
            for _ in range(k1):
                eng.msm_device(curve, group, b1.data_ptr(), s1.data_ptr(), total, o1.data_ptr(),
                               out_form=libff_amd.OUT_LIBFF, stream=st1.cuda_stream)
            st1.synchronize()
            e1 = time.perf_counter() - t0
            legs["same_total_on_one_gpu"] = {"value": total * k1 / e1, "unit": "scalar-muls/s", "ms_per_step": e1 / k1 * 1e3,
                                             "steps": k1, "what": f"the whole 2^{log2n}-point MSM on rank 0's GPU alone "
                                             "(strong-scaling reference for `value`)"}
            del b1, s1
            torch.cuda.empty_cache()
        tm.fence()

        # weak scaling: 2^20 points per GPU
        nw = 1 << 20
        bw, sw = gen_inputs(eng, curve, group, rank * nw, nw, dev, 555 + rank)
        kw = max(args.steps, 10)
        ew, _ = timed_msm(tm, eng, msm, bw, sw, nw, kw, 2, want_phases=False)
        legs["weak_2p20_per_gpu"] = {"value": nw * world * kw / ew, "unit": "scalar-muls/s", "ms_per_step": ew / kw * 1e3,
                                     "steps": kw, "scaling": "weak", "points_per_gpu": nw}
        del bw, sw

        # configs[4]: bw6_761 G1 + bls12_377 G2, 2^24 points in total each, issued together
        if args.config4_log2n:
            t4 = 1 << args.config4_log2n
            lo4, hi4 = shard_range(t4, world, rank)
            n4 = hi4 - lo4
            # (both groups have a cofactor: the bases here are multiples of the generator, so the
            # endomorphism split is permitted explicitly -- amdmsm_opts.endomorphism = 1)
            eng2 = libff_amd.Engine(eng.device, endomorphism=1)   # second context: its own stream and workspace
            eng.endomorphism = 1
            jobs = []
            for e, (cv, gp) in ((eng, (2, 1)), (eng2, (1, 2))):
                b4, s4 = gen_inputs(e, cv, gp, lo4, n4, dev, 777 + rank)
                jobs.append((ShardedMsm(e, cv, gp, depth=1), b4, s4))

            def both():
                for m, b4, s4 in jobs:   # asynchronous: the two MSMs (and their all-gathers) are in flight together
                    m.run(b4, s4, n4, libff_amd.OUT_LIBFF)

            both()
            for m, _, _ in jobs:
                m.synchronize()
            tm.fence()
            k4 = max(2, min(args.steps, 5))
            t0 = time.perf_counter()
            for _ in range(k4):
                both()
            for m, _, _ in jobs:
                m.synchronize()
            tm.fence()
            e4 = tm.max_over_ranks(time.perf_counter() - t0)
            legs["config4_bw6_761_g1_plus_bls12_377_g2"] = {
                "value": 2 * t4 * k4 / e4, "unit": "scalar-muls/s", "ms_per_step": e4 / k4 * 1e3, "steps": k4,
                "what": f"BASELINE configs[4]: bw6_761 G1 MSM + bls12_377 G2 MSM, 2^{args.config4_log2n} points in total each "
                        f"({n4} per rank), both issued together on two streams per rank; a step = both MSMs incl. their "
                        "all-gathers; value counts the scalar-muls of both; endomorphism split permitted (bases in the order-r subgroup)"}
            del jobs
            eng2.close()
            eng.endomorphism = args.endomorphism

    out = {
        "metric": "G1 MSM throughput (scalar-muls/sec)" if group == 1 else "G2 MSM throughput (scalar-muls/sec)",
        "value": total * args.steps / elapsed,
        "unit": "scalar-muls/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "config": {
            "workload": f"{args.curve} G{group} MSM, 2^{log2n} points in total over {world} GPUs (BASELINE configs[3]): "
                        f"rank r owns the contiguous range [r*n/N, (r+1)*n/N) of bases (i+1)G (affine, resident in its HBM) "
                        "and uniform random scalars; one partial point per rank, RCCL all-gather, local sum",
            "points_per_gpu": n,
            "total_points": total,
            "window_bits": plan["c"],
            "num_windows": plan["num_windows"],
            "endomorphism_split": plan["endomorphism"],
            "parallelism": f"range-sharded x{world} (multiexp.tcc:663-687 with rank = chunk), all-gather of partial points "
                           "+ local sum",
            "msms_in_flight": 1,
            "phases_ms": mean_phases(phases),
            "shard_ms": shard_ms,
            "exchange_ms": exchange_ms,
            "predicted_ms": shard_ms + exchange_ms,
            "predicted_note": "ms_per_step should equal shard_ms (the slowest rank's MSM over its 2^log2n / N points, HIP events) "
                              "+ exchange_ms (RCCL all-gather of N partial points + k_sum_points, timed back to back on its own); "
                              "DESIGN.md section 5 states the expected strong-scaling curve",
            "legs": legs,
        },
        "roofline": roofline_of(args.curve, curve, group, n, plan, acc, 0),
    }
    return out


def main():
    args = ARGS
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    # AMDMSM_BENCH_REHEARSAL=1: every rank uses GPU 0 and gloo carries the collectives -- the multi-rank
    # code path on a one-GPU box (RCCL refuses several ranks on one device).  Never a measurement.
    rehearsal = os.environ.get("AMDMSM_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    curve, group = CURVES[args.curve], args.group
    eng = libff_amd.Engine(local_rank, endomorphism=args.endomorphism)
    tm = Timer(world, dev)
    if world == 1:
        out = single_gpu(args, tm, eng, dev, curve, group)
    else:
        out = multi_gpu(args, tm, eng, dev, rank, world, curve, group)
    if rank == 0:
        if rehearsal:
            out["data"] = "synthetic; REHEARSAL (all ranks share GPU 0, gloo): not a measurement"
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(curve, group, args.cpu_log2n)
            except Exception as e:  # the baseline is a reported extra; never lose the GPU line to it
                out["cpu_baseline"] = {"value": None, "unit": "scalar-muls/s", "cores": 0, "kind": "port",
                                       "sample": f"failed: {e!r}"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
