"""-m gpu: the multi-GPU shapes of BASELINE configs[3] / configs[4] and the boundary features
around them, on the one GPU a test box has.

  * per-rank shard sizes: alt_bn128 G1 2^23 (2^26 over 8 GPUs), bls12_377 G1 2^22 (configs[2]),
    bw6_761 G1 2^21 and bls12_377 G2 2^21 issued together (2^24 over 8 GPUs) -- closed form
    (the reference's own test pattern, test_multiexp.cpp:205-256) and sharded == unsharded;
  * the C-ABI multi-device entries (amdmsm_multi_exp_multi / amdmsm_msm_device_multi,
    multiexp.tcc:655-687 with chunk = context) with ndev = 1 and with two contexts on one device;
  * ShardedMsm (the torch.distributed layer bench.py uses) through the nccl backend at world
    size 1, with inputs that change every step;
  * resident base vectors (amdmsm_register_bases) and the device-side filter_one_zero counts.
"""
import os
import sys

import numpy as np
import pytest

import libff_amd
from common import golden, small_scalars_mont, to_int
from libff_amd import multi_exp_base_form_normal, multi_exp_base_form_special

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _closed_form_point(port, curve, group, scalars_mont, first):
    """(sum_i s_i * (first + i + 1) mod r) * G::one(), affine -- test_multiexp.cpp:205-256."""
    plain = port.fr_as_bigint(curve, scalars_mont)
    n, fl = plain.shape
    r = to_int(golden()[f"{libff_amd.engine.CURVE_NAMES[curve]}_g1/fr_modulus"])
    lo32 = (plain & np.uint64(0xFFFFFFFF)).astype(np.uint64)
    hi32 = (plain >> np.uint64(32)).astype(np.uint64)
    total = 0
    blk = 256   # 2^32 * 2^24 * 2^8 = 2^64 would wrap: weights stay below 2^24 + blk here
    for b0 in range(0, n, blk):
        w = np.arange(first + b0 + 1, first + min(b0 + blk, n) + 1, dtype=np.uint64)[:, None]
        lo = (lo32[b0:b0 + blk] * w).sum(axis=0, dtype=np.uint64)
        hi = (hi32[b0:b0 + blk] * w).sum(axis=0, dtype=np.uint64)
        for j in range(fl):
            total += (int(lo[j]) << (64 * j)) + (int(hi[j]) << (64 * j + 32))
    k = total % r
    k_plain = np.array([[(k >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(fl)]], dtype=np.uint64)
    k_mont = port.fr_from_bigint(curve, k_plain)[0]
    one, _ = port.group_consts(curve, group)
    return port.group_op(curve, group, 4, port.scalar_mul(curve, group, one, k_mont))


class _Resident:
    """(first + i + 1) * G bases generated in HBM + uploaded scalars, for one engine context."""

    def __init__(self, eng, curve, group, n, first, scalars):
        self.eng, self.n = eng, n
        s = libff_amd.sizes(curve, group)
        self.s = s
        self.d_bases = eng.malloc(n * s["affine_bytes"])
        self.d_sc = eng.malloc(scalars.nbytes)
        self.d_out = eng.malloc(s["g_bytes"])
        eng.gen_bases_seq_device(curve, group, first, n, self.d_bases.value)
        eng.h2d(self.d_sc, scalars)

    def result(self):
        self.eng.synchronize()
        out = np.zeros(self.s["g_bytes"] // 8, dtype=np.uint64)
        self.eng.d2h(out, self.d_out)
        return out

    def free(self):
        for p in (self.d_bases, self.d_sc, self.d_out):
            self.eng.free(p)


@pytest.mark.parametrize("name,curve,group,log2n", [("alt_bn128_g1", 0, 1, 23), ("bls12_377_g1", 1, 1, 22)])
def test_per_rank_shard_sizes_closed_form_and_sharding(engine, port, name, curve, group, log2n):
    """configs[3]'s per-rank shard (alt_bn128 G1 2^23 = 2^26 / 8) and configs[2] (bls12_377 G1 2^22)
    at full size: closed form, and the same input as two ranges on two contexts
    (amdmsm_msm_device_multi; odd split so the last range takes a remainder) == unsharded."""
    n = 1 << log2n
    sc = port.scalars_sha512(curve, 0, n)
    want = _closed_form_point(port, curve, group, sc, 0)
    r = _Resident(engine, curve, group, n, 0, sc)
    e2 = libff_amd.Engine(0)
    try:
        engine.msm_device(curve, group, r.d_bases.value, r.d_sc.value, n, r.d_out.value, out_form=libff_amd.OUT_AFFINE)
        assert (r.result() == want).all()
        # two ranges [0, one) and [one, n) with one = (n - 5) // 2: pointers into the same arrays
        s = r.s
        one = (n - 5) // 2
        bases = [r.d_bases.value, r.d_bases.value + one * s["affine_bytes"]]
        scal = [r.d_sc.value, r.d_sc.value + one * s["fr_bytes"]]
        engine.h2d(r.d_out, np.zeros(s["g_bytes"] // 8, dtype=np.uint64))
        libff_amd.msm_device_multi([engine, e2], curve, group, bases, scal, [one, n - one], r.d_out.value,
                                   out_form=libff_amd.OUT_AFFINE)
        assert (r.result() == want).all()
    finally:
        r.free()
        e2.close()


def test_north_star_size_2p26_closed_form(engine, port):
    """The north-star size on one GPU: alt_bn128 G1, 2^26 points, bit-exact against a closed form.
    Scalars s_i = pool[i mod 4096] (4096 SHA512_rng values), bases (i+1)G generated on the device, so
    sum_i s_i (i+1) G = (sum_j pool_j * sum_{i = j mod 4096} (i+1)) G needs 4096 terms on the CPU
    while the device sorts and accumulates all 13 x 2^26 (point, window) entries.  Then the same
    input as two uneven ranges on two contexts (amdmsm_msm_device_multi).  Device-resident: the
    6 GiB of inputs stay in HBM."""
    curve, group, log2n, m = 0, 1, 26, 4096
    n = 1 << log2n
    pool = port.scalars_sha512(curve, 4242, m)
    plain = port.fr_as_bigint(curve, pool)
    r = to_int(golden()["alt_bn128_g1/fr_modulus"])
    reps = n // m
    k = 0
    for j in range(m):
        # i = j + t*m, t < reps: sum (i + 1) = reps*(j + 1) + m * reps*(reps-1)/2
        k += to_int(plain[j]) * (reps * (j + 1) + m * (reps * (reps - 1) // 2))
    k %= r
    fl = pool.shape[1]
    k_mont = port.fr_from_bigint(curve, np.array([[(k >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(fl)]], dtype=np.uint64))[0]
    one, _ = port.group_consts(curve, group)
    want = port.group_op(curve, group, 4, port.scalar_mul(curve, group, one, k_mont))
    sc = np.ascontiguousarray(np.tile(pool, (reps, 1)))
    res = _Resident(engine, curve, group, n, 0, sc)
    del sc
    e2 = libff_amd.Engine(0)
    try:
        engine.msm_device(curve, group, res.d_bases.value, res.d_sc.value, n, res.d_out.value, out_form=libff_amd.OUT_AFFINE)
        assert (res.result() == want).all()
        s = res.s
        one_r = n // 2 - 3 * m - 1
        engine.h2d(res.d_out, np.zeros(s["g_bytes"] // 8, dtype=np.uint64))
        libff_amd.msm_device_multi([engine, e2], curve, group,
                                   [res.d_bases.value, res.d_bases.value + one_r * s["affine_bytes"]],
                                   [res.d_sc.value, res.d_sc.value + one_r * s["fr_bytes"]], [one_r, n - one_r],
                                   res.d_out.value, out_form=libff_amd.OUT_AFFINE)
        assert (res.result() == want).all()
    finally:
        res.free()
        e2.close()


def _tiled_pool_closed_form(port, curve, group, n, m, seed):
    """scalars s_i = pool[i mod m], bases (i+1)G: sum_i s_i (i+1) G = (sum_j pool_j * sum_{i = j mod m} (i+1)) G,
    m terms on the CPU (test_multiexp.cpp:205-256's pattern); returns (pool, expected affine point)."""
    pool = port.scalars_sha512(curve, seed, m)
    plain = port.fr_as_bigint(curve, pool)
    r = to_int(golden()[f"{libff_amd.engine.CURVE_NAMES[curve]}_g1/fr_modulus"])
    reps = n // m
    k = 0
    for j in range(m):
        k += to_int(plain[j]) * (reps * (j + 1) + m * (reps * (reps - 1) // 2))
    k %= r
    fl = pool.shape[1]
    k_mont = port.fr_from_bigint(curve, np.array([[(k >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(fl)]], dtype=np.uint64))[0]
    one, _ = port.group_consts(curve, group)
    return pool, port.group_op(curve, group, 4, port.scalar_mul(curve, group, one, k_mont))


@pytest.mark.parametrize("name,curve,group", [("bw6_761_g1", 2, 1), ("bls12_377_g2", 1, 2)])
@pytest.mark.parametrize("endo", [0, 1])
def test_config4_stated_totals_2p24_closed_form(port, name, curve, group, endo):
    """BASELINE configs[4] at its stated total on one GPU: bw6_761 G1 and bls12_377 G2 MSMs over 2^24
    points (3 GiB of affine bases each, generated in HBM), bit-exact against the closed form -- on the
    default path and with the endomorphism split permitted (the bases are multiples of the generator),
    which is how bench.py's configs[4] legs run."""
    n, m = 1 << 24, 4096
    pool, want = _tiled_pool_closed_form(port, curve, group, n, m, 2424 + group)
    eng = libff_amd.Engine(0, endomorphism=endo)
    sc = np.ascontiguousarray(np.tile(pool, (n // m, 1)))
    res = _Resident(eng, curve, group, n, 0, sc)
    del sc
    try:
        eng.msm_device(curve, group, res.d_bases.value, res.d_sc.value, n, res.d_out.value, out_form=libff_amd.OUT_AFFINE)
        assert (res.result() == want).all()
    finally:
        res.free()
        eng.close()


def test_config4_two_msms_issued_together(engine, port):
    """configs[4] per-rank shard: bw6_761 G1 2^21 and bls12_377 G2 2^21 enqueued back to back on two
    contexts (two streams) with no synchronisation in between, as a BW6/BLS12 prover issues
    them; both checked against the closed form, and again after swapping the issue order."""
    n = 1 << 21
    e2 = libff_amd.Engine(0)
    jobs = []
    try:
        for eng, (curve, group) in ((engine, (2, 1)), (e2, (1, 2))):
            sc = port.scalars_sha512(curve, 12345, n)
            jobs.append((eng, curve, group, _Resident(eng, curve, group, n, 7, sc), _closed_form_point(port, curve, group, sc, 7)))
        for order in (jobs, jobs[::-1]):
            for eng, curve, group, r, _ in order:
                eng.h2d(r.d_out, np.zeros(r.s["g_bytes"] // 8, dtype=np.uint64))
            for eng, curve, group, r, _ in order:   # asynchronous launches: both MSMs are in flight together
                eng.msm_device(curve, group, r.d_bases.value, r.d_sc.value, n, r.d_out.value, out_form=libff_amd.OUT_AFFINE)
            for eng, curve, group, r, want in order:
                assert (r.result() == want).all(), (curve, group)
    finally:
        for j in jobs:
            j[3].free()
        e2.close()


@pytest.mark.parametrize("name,curve,group,n", [("alt_bn128_g1", 0, 1, 70001), ("bls12_377_g2", 1, 2, 4099),
                                                 ("bw6_761_g1", 2, 1, 3001)])
def test_multi_exp_multi_c_abi(engine, port, name, curve, group, n):
    """amdmsm_multi_exp_multi from host vectors: ndev = 1, two and three contexts on one device
    (remainder in the last range), normal- and special-form bases, vs the oracle; n < ndev."""
    bases = port.bases_seq(curve, group, n, first=21)
    sc = port.scalars_sha512(curve, 4242, n)
    want = port.multi_exp(curve, group, bases, sc, port.BDLO12_SIGNED, 1, chunks=8, omp=True)
    extra = [libff_amd.Engine(0), libff_amd.Engine(0)]
    try:
        for engines in ([engine], [engine, extra[0]], [engine] + extra):
            got = libff_amd.multi_exp_multi(engines, curve, group, bases, sc, base_form=multi_exp_base_form_special)
            assert (got == want).all(), len(engines)
        # normal-form bases (3 * P_i, Z != 1): shared inversions per range
        nb = engine.group_op(curve, group, 0, engine.group_op(curve, group, 2, bases[:500]), bases[:500])
        want_n = port.multi_exp(curve, group, nb, sc[:500], port.BDLO12_SIGNED, 0)
        got = libff_amd.multi_exp_multi([engine, extra[0]], curve, group, nb, sc[:500], base_form=multi_exp_base_form_normal)
        assert (got == want_n).all()
        # fewer points than contexts: falls back to one call (multiexp.tcc:656)
        got = libff_amd.multi_exp_multi([engine] + extra, curve, group, bases[:2], sc[:2], base_form=multi_exp_base_form_special)
        assert (got == port.multi_exp(curve, group, bases[:2], sc[:2], port.BDLO12_SIGNED, 1)).all()
        with pytest.raises(libff_amd.AmdMsmError):
            libff_amd.multi_exp_multi([engine, engine], curve, group, bases, sc)
    finally:
        for e in extra:
            e.close()


def test_resident_bases_registry(engine, port):
    """amdmsm_register_bases: later calls on the registered vector (or on row ranges of it) read the
    copy in HBM -- proven by changing the host array after registration: the result keeps following
    the registered content until amdmsm_invalidate_bases, then follows the host array again."""
    curve, group, n = 0, 1, 5000
    bases = np.ascontiguousarray(port.bases_seq(curve, group, n, first=0))
    other = port.bases_seq(curve, group, n, first=100000)
    sc1, sc2 = port.scalars_sha512(curve, 1, n), port.scalars_sha512(curve, 2, n)
    w1 = port.multi_exp(curve, group, bases, sc1, port.BDLO12_SIGNED, 1, chunks=4, omp=True)
    w2 = port.multi_exp(curve, group, bases, sc2, port.BDLO12_SIGNED, 1, chunks=4, omp=True)
    h = engine.register_bases(curve, group, bases, multi_exp_base_form_special)
    try:
        assert (engine.multi_exp(curve, group, bases, sc1, base_form=multi_exp_base_form_special) == w1).all()
        assert (engine.multi_exp(curve, group, bases, sc2, base_form=multi_exp_base_form_special) == w2).all()
        # a row range of the registered vector (what a chunked / multi-GPU caller passes)
        sub = bases[1000:3500]
        wsub = port.multi_exp(curve, group, sub, sc1[1000:3500], port.BDLO12_SIGNED, 1)
        assert (engine.multi_exp(curve, group, sub, sc1[1000:3500], base_form=multi_exp_base_form_special) == wsub).all()
        # a different form is a different import: not served from the registration
        assert (engine.multi_exp(curve, group, bases, sc1, base_form=multi_exp_base_form_normal) == w1).all()
        keep = bases.copy()
        bases[:] = other     # the registration still holds the old content
        assert (engine.multi_exp(curve, group, bases, sc1, base_form=multi_exp_base_form_special) == w1).all()
        engine.invalidate_bases(bases)
        w_other = port.multi_exp(curve, group, other, sc1, port.BDLO12_SIGNED, 1, chunks=4, omp=True)
        assert (engine.multi_exp(curve, group, bases, sc1, base_form=multi_exp_base_form_special) == w_other).all()
        bases[:] = keep
        with pytest.raises(libff_amd.AmdMsmError):
            engine.unregister_bases(h)       # dropped by the invalidation
        h = engine.register_bases(curve, group, bases, multi_exp_base_form_special)
        assert (engine.multi_exp(curve, group, bases, sc2, base_form=multi_exp_base_form_special) == w2).all()
    finally:
        engine.invalidate_bases(None)


def test_filter_one_zero_counts_on_device_large(engine, port):
    """multi_exp_filter_one_zero's three counts (multiexp.tcc:713-757) come from device counters;
    2^20 witness-like scalars (most of them 0 / 1), Montgomery and plain representation."""
    curve, group, n = 0, 1, 1 << 20
    rng = np.random.default_rng(3)
    sc = port.scalars_sha512(curve, 0, n)
    zo = small_scalars_mont(port, curve, [0, 1])
    pick = rng.integers(0, 5, size=n)
    sc[pick == 0] = zo[0]
    sc[(pick == 1) | (pick == 2)] = zo[1]
    bases = engine.gen_bases_seq(curve, group, n, first=0)
    want = port.multi_exp(curve, group, bases, sc, port.BDLO12_SIGNED, 1, chunks=16, omp=True)
    got, stats = engine.multi_exp_filter_one_zero(curve, group, bases, sc, base_form=multi_exp_base_form_special)
    assert (got == want).all()
    assert stats == {"skipped": int((pick == 0).sum()), "ones": int(((pick == 1) | (pick == 2)).sum()),
                     "other": int((pick >= 3).sum())}
    plain = port.fr_as_bigint(curve, sc)
    got, stats2 = engine.multi_exp_filter_one_zero(curve, group, bases, plain, base_form=multi_exp_base_form_special,
                                                   scalars_plain=True)
    assert (got == want).all() and stats2 == stats


def test_precomputed_rejects_unservable_num_digits(engine, port):
    """(c, num_digits) pairs the kernels cannot serve are refused up front, and a non-default
    num_digits (one more than ceil(bits / c): keeps the final carry) is served."""
    curve, group, n, c = 0, 1, 600, 11
    D = libff_amd.precompute_num_digits(curve, c)
    b = port.bases_seq(curve, group, n, first=3)
    s = port.scalars_sha512(curve, 5, n)
    z = libff_amd.sizes(curve, group)
    tab = engine.precompute_table(curve, group, b, c, num_digits=D + 1)
    d_src, d_tab = engine.malloc(tab.nbytes), engine.malloc(tab.shape[0] * z["affine_bytes"])
    d_sc, d_out = engine.malloc(s.nbytes), engine.malloc(z["g_bytes"])
    try:
        engine.h2d(d_src, tab)
        engine.h2d(d_sc, s)
        engine.import_bases_device(curve, group, d_src, tab.strides[0], 1, tab.shape[0], d_tab)
        engine.msm_precomputed_device(curve, group, d_tab, d_sc, n, c, D + 1, d_out, out_form=libff_amd.OUT_AFFINE)
        engine.synchronize()
        out = np.zeros(z["g_bytes"] // 8, dtype=np.uint64)
        engine.d2h(out, d_out)
        assert (out == port.multi_exp(curve, group, b, s, port.BDLO12_SIGNED, 1, chunks=4, omp=True)).all()
        for bad in (D + 2, 200, 512):
            with pytest.raises(libff_amd.AmdMsmError):
                engine.msm_precomputed_device(curve, group, d_tab, d_sc, n, c, bad, d_out)
    finally:
        for p in (d_src, d_tab, d_sc, d_out):
            engine.free(p)


def test_sharded_msm_nccl_world_size_one():
    """ShardedMsm (libff_amd/distributed.py, what bench.py --gpus N runs) through the nccl backend
    at world size 1, depth 1 and 2, with scalars that change every step: each result against the
    oracle -- without and with force_exchange (the all-gather over RCCL on the device tensor and the
    combining kernel executed even though there is one rank).  Child process: it initialises
    torch.distributed."""
    import subprocess

    code = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
import libff_amd
from libff_amd.distributed import ShardedMsm, numpy_words
from oracle import port
port.build()
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
curve, group, n = 0, 1, 20000
eng = libff_amd.Engine(0)
sz = libff_amd.sizes(curve, group)
bases_h = port.bases_seq(curve, group, n, first=0)
bases = torch.empty((n, sz["affine_bytes"] // 8), dtype=torch.int64, device=dev)
eng.gen_bases_seq_device(curve, group, 0, n, bases.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
for depth, force in ((1, False), (2, False), (1, True), (2, True)):
    # force_exchange: the RCCL all-gather on the device tensor and k_sum_points really run at world size 1
    msm = ShardedMsm(eng, curve, group, depth=depth, force_exchange=force)
    outs, wants = [], []
    for step in range(4):
        sc_h = port.scalars_sha512(curve, 1000 * step + depth, n)
        sc = torch.from_numpy(sc_h.view(np.int64)).to(dev)        # produced on the current stream
        res, slot = msm.run(bases, sc, n, libff_amd.OUT_AFFINE)
        msm.streams[slot %% depth].synchronize()
        outs.append(numpy_words(res).copy())
        wants.append(port.multi_exp(curve, group, bases_h, sc_h, port.BDLO12_SIGNED, 1, chunks=8, omp=True))
    msm.synchronize()
    for o, w in zip(outs, wants):
        assert (o == w).all()
dist.destroy_process_group()
print("sharded-nccl-ok")
''' % (REPO, os.path.join(REPO, "tests"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "sharded-nccl-ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def _sharded_rank(rank, world, port_no, n, steps, q):
    """one rank of test_sharded_msm_two_ranks_one_gpu (spawned: fresh process, own engine context)"""
    import torch
    import torch.distributed as dist

    from libff_amd.distributed import ShardedMsm, numpy_words, shard_range
    from oracle import port

    port.build()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port_no}", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    curve, group = 0, 1
    eng = libff_amd.Engine(0)
    sz = libff_amd.sizes(curve, group)
    lo, hi = shard_range(n, world, rank)
    bases = torch.empty((hi - lo, sz["affine_bytes"] // 8), dtype=torch.int64, device=dev)
    eng.gen_bases_seq_device(curve, group, lo, hi - lo, bases.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    msm = ShardedMsm(eng, curve, group, depth=1)
    outs = []
    for step in range(steps):
        sc_all = port.scalars_sha512(curve, 5000 * step, n)          # every rank draws the same vector ...
        sc = torch.from_numpy(sc_all[lo:hi].view(np.int64).copy()).to(dev)   # ... and uploads its own range
        res, _ = msm.run(bases, sc, hi - lo, libff_amd.OUT_AFFINE)
        msm.synchronize()
        outs.append(numpy_words(res).copy())
    q.put((rank, outs))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_msm_two_ranks_one_gpu(port):
    """The multi-rank path of bench.py --gpus N (ShardedMsm: range shard, all-gather of the partial
    points, local sum) with TWO ranks, both on the one GPU of the test box (gloo carries the
    exchange: RCCL refuses two ranks on one device), scalars changing every step: every rank's
    result at every step == the oracle's MSM over the whole input (odd n: the last range is longer)."""
    import socket

    import torch.multiprocessing as mp

    n, steps, world = 30001, 3, 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port_no = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_rank, args=(r, world, port_no, n, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    bases = port.bases_seq(0, 1, n, first=0)
    for step in range(steps):
        want = port.multi_exp(0, 1, bases, port.scalars_sha512(0, 5000 * step, n), port.BDLO12_SIGNED, 1, chunks=8, omp=True)
        for r in range(world):
            assert (got[r][step] == want).all(), (r, step)


def test_bench_multi_rank_path_rehearsal():
    """bench.py --gpus 2 end to end on the one-GPU box: the script starts its two ranks itself, both
    use GPU 0 and gloo carries the collectives (AMDMSM_BENCH_REHEARSAL=1; RCCL refuses two ranks on
    one device).  Checks the contract of the printed line: strong scaling on a fixed total, the
    configs[4] leg with both MSMs in flight, the weak-scaling leg, rank 0's solo reference."""
    import json
    import subprocess

    env = dict(os.environ, AMDMSM_BENCH_REHEARSAL="1")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--log2n", "21", "--config4-log2n", "15"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [x for x in r.stdout.splitlines() if x.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 2 and d["warmup"] == 1
    assert d["config"]["total_points"] == 1 << 21 and d["config"]["points_per_gpu"] == 1 << 20
    assert d["value"] > 0 and abs(d["value"] - (1 << 21) / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    legs = d["config"]["legs"]
    assert set(legs) == {"same_total_on_one_gpu", "weak_2p20_per_gpu", "config4_bw6_761_g1_plus_bls12_377_g2"}
    assert all(v["value"] > 0 for v in legs.values())
    assert d["roofline"]["frac"] > 0 and "cpu_baseline" not in d
    cfg = d["config"]
    assert cfg["shard_ms"] > 0 and cfg["exchange_ms"] > 0 and abs(cfg["predicted_ms"] - cfg["shard_ms"] - cfg["exchange_ms"]) < 1e-9


def test_range_split_peer_copy_and_filter_multi():
    """Inputs longer than one MSM handles are cut into contiguous ranges whose partial results are summed
    (the reference's chunk loop, multiexp.tcc:655-687) -- AMDMSM_MAX_RANGE_POINTS lowers the limit from
    2^28 to 5000 points here so that every entry point takes that path: the device entry, the host entry
    (with its 0 / 1 statistics accumulated over the ranges), the two multi-device entries.
    AMDMSM_FORCE_PEER_COPY=1 makes the multi-device entries call hipMemcpyPeerAsync although both
    contexts sit on the one GPU of the box.  Everything against the oracle.  Child process: both knobs
    are read once per process."""
    import subprocess

    code = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
import libff_amd
from oracle import port
port.build()
e1, e2 = libff_amd.Engine(0), libff_amd.Engine(0)
for curve, group, n in ((0, 1, 20001), (1, 2, 12007)):
    s = libff_amd.sizes(curve, group)
    bases = port.bases_r32(curve, group, n) if curve else port.bases_seq(curve, group, n)
    sc = port.scalars_sha512(curve, 77, n)
    one = port.fr_from_bigint(curve, np.array([[1] + [0] * (sc.shape[1] - 1)], dtype=np.uint64))[0]
    sc[5::7] = 0
    sc[3::11] = one
    ones, zeros = len(range(3, n, 11)), len([i for i in range(5, n, 7) if i %% 11 != 3])   # the ones overwrite some zeros
    want = port.multi_exp(curve, group, bases, sc, port.BDLO12_SIGNED, port.FORM_SPECIAL, chunks=8, omp=True)
    # host entry, 5 / 3 ranges
    got = e1.multi_exp(curve, group, bases, sc, base_form=libff_amd.multi_exp_base_form_special)
    assert (got == want).all(), ("host", curve, group)
    got, st = e1.multi_exp_filter_one_zero(curve, group, bases, sc, base_form=libff_amd.multi_exp_base_form_special)
    assert (got == want).all() and (st["skipped"], st["ones"], st["other"]) == (zeros, ones, n - zeros - ones), st
    # two contexts, each range split again; partials through hipMemcpyPeerAsync
    got = libff_amd.multi_exp_multi([e1, e2], curve, group, bases, sc, base_form=libff_amd.multi_exp_base_form_special)
    assert (got == want).all(), ("multi", curve, group)
    got, st = libff_amd.multi_exp_filter_one_zero_multi([e1, e2], curve, group, bases, sc,
                                                        base_form=libff_amd.multi_exp_base_form_special)
    assert (got == want).all() and (st["skipped"], st["ones"], st["other"]) == (zeros, ones, n - zeros - ones), st
    # device entry
    aff = np.ascontiguousarray(bases[:, : s["affine_bytes"] // 8])
    d_b, d_s, d_o = e1.malloc(aff.nbytes), e1.malloc(sc.nbytes), e1.malloc(s["g_bytes"])
    e1.h2d(d_b, aff); e1.h2d(d_s, sc)
    e1.msm_device(curve, group, d_b.value, d_s.value, n, d_o.value, out_form=libff_amd.OUT_AFFINE)
    e1.synchronize()
    out = np.zeros(s["g_bytes"] // 8, dtype=np.uint64)
    e1.d2h(out, d_o)
    assert (out == want).all(), ("device", curve, group)
    half = n // 2 - 1
    libff_amd.msm_device_multi([e1, e2], curve, group, [d_b.value, d_b.value + half * s["affine_bytes"]],
                               [d_s.value, d_s.value + half * s["fr_bytes"]], [half, n - half], d_o.value,
                               out_form=libff_amd.OUT_AFFINE)
    e1.d2h(out, d_o)
    assert (out == want).all(), ("device multi", curve, group)
print("ranges-ok")
''' % (REPO, os.path.join(REPO, "tests"))
    env = dict(os.environ, AMDMSM_MAX_RANGE_POINTS="5000", AMDMSM_FORCE_PEER_COPY="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "ranges-ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_opts_struct_size_is_checked(engine):
    """amdmsm_opts carries its own size (AMDMSM_ABI_VERSION 3): a caller compiled against another layout
    is refused with AMDMSM_ERR_BAD_ARG instead of having its fields misread."""
    import ctypes

    from libff_amd.engine import _Opts

    assert engine.lib.amdmsm_abi_version() == libff_amd.engine.ABI_VERSION
    o = engine._opts()
    o.struct_size = ctypes.sizeof(_Opts) - 8   # e.g. the 24-byte layout of the first ABI
    out = np.zeros(12, dtype=np.uint64)
    rc = engine.lib.amdmsm_multi_exp(engine.h, 0, 1, None, ctypes.c_size_t(0), 0, None, ctypes.c_size_t(0),
                                     out.ctypes.data_as(ctypes.c_void_p), ctypes.byref(o))
    assert rc == -2 and b"struct_size" in engine.lib.amdmsm_last_error(engine.h)
    o.struct_size = ctypes.sizeof(_Opts)
    rc = engine.lib.amdmsm_multi_exp(engine.h, 0, 1, None, ctypes.c_size_t(0), 0, None, ctypes.c_size_t(0),
                                     out.ctypes.data_as(ctypes.c_void_p), ctypes.byref(o))
    assert rc == 0


def test_overlap_mode_parity():
    """AMDMSM_OVERLAP=1 (experimental, off by default: DESIGN.md section 7): with several MSMs in flight the bulk of every MSM
    runs on one engine stream in call order and its tail on a second one, the accumulation kernel one workgroup per CU
    short.  Three contexts' worth of back-to-back MSMs with scalars that change every step, every result against the
    oracle; also a group without overlap support (falls back to the plain path) and the one-at-a-time depth."""
    import subprocess

    code = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, torch
import libff_amd
from libff_amd.distributed import ShardedMsm, numpy_words
from oracle import port
port.build()
dev = torch.device("cuda", 0)
eng = libff_amd.Engine(0)
for curve, group, n in ((0, 1, 50001), (1, 1, 9001)):
    sz = libff_amd.sizes(curve, group)
    bases_h = port.bases_seq(curve, group, n, first=0)
    bases = torch.empty((n, sz["affine_bytes"] // 8), dtype=torch.int64, device=dev)
    eng.gen_bases_seq_device(curve, group, 0, n, bases.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for depth in (3, 1, 2):
        msm = ShardedMsm(eng, curve, group, depth=depth)
        outs, wants, keep = [], [], []
        for step in range(2 * depth + 1):
            sc_h = port.scalars_sha512(curve, 300 * step + depth, n)
            sc = torch.from_numpy(sc_h.view(np.int64)).to(dev)
            keep.append(sc)
            res, slot = msm.run(bases, sc, n, libff_amd.OUT_AFFINE)
            outs.append((res, slot))
            wants.append(port.multi_exp(curve, group, bases_h, sc_h, port.BDLO12_SIGNED, 1, chunks=8, omp=True))
            if len(outs) >= depth:   # read a result before its slot is reused
                r, sl = outs[-depth]
                msm.streams[sl %% depth].synchronize()
                assert (numpy_words(r) == wants[-depth]).all(), (curve, depth, step)
        msm.synchronize()
print("overlap-ok")
''' % (REPO, os.path.join(REPO, "tests"))
    env = dict(os.environ, AMDMSM_OVERLAP="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "overlap-ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.parametrize("name,curve,group,n,k", [("alt_bn128_g1", 0, 1, 30001, 3), ("bls12_377_g2", 1, 2, 5003, 2),
                                                   ("bw6_761_g1", 2, 1, 2500, 4), ("alt_bn128_g1", 0, 1, 700, 8)])
def test_msm_device_batch(engine, port, name, curve, group, n, k):
    """amdmsm_msm_device_batch: k MSMs of one group and length in one call -- different bases (R32 sets shifted against each
    other, (i+1)G, repeated points) and different scalars per MSM (one with 0 / 1 / r - 1 and a heavy hitter) -- every result
    against the oracle's multi_exp of that pair; once more with a forced window size; and k = 1."""
    s = libff_amd.sizes(curve, group)
    aw = s["affine_bytes"] // 8
    sets, wants = [], []
    for j in range(k):
        if j % 3 == 0:
            b = port.bases_seq(curve, group, n, first=1000 * j)
        elif j % 3 == 1:
            b = np.roll(port.bases_r32(curve, group, n), 5 * j, axis=0)
        else:
            b = np.repeat(port.bases_seq(curve, group, (n + 6) // 7, first=3), 7, axis=0)[:n]
        sc = port.scalars_sha512(curve, 40 * j + 1, n)
        if j == 1:
            sc[0:n:9] = 0
            sc[1:n:13] = small_scalars_mont(port, curve, [1])[0]
            sc[2:n:5] = sc[2]
        sets.append((np.ascontiguousarray(b[:, :aw]), sc))
        wants.append(port.multi_exp(curve, group, b, sc, port.BDLO12_SIGNED, port.FORM_SPECIAL, chunks=8, omp=True))
    ptrs = []
    try:
        for b, sc in sets:
            d_b, d_s, d_o = engine.malloc(b.nbytes), engine.malloc(sc.nbytes), engine.malloc(s["g_bytes"])
            engine.h2d(d_b, b)
            engine.h2d(d_s, sc)
            ptrs.append((d_b, d_s, d_o))
        for wb, kk in ((0, k), (12, k), (0, 1)):
            for _, _, d_o in ptrs:
                engine.h2d(d_o, np.zeros(s["g_bytes"] // 8, dtype=np.uint64))
            engine.msm_device_batch(curve, group, [p[0].value for p in ptrs[:kk]], [p[1].value for p in ptrs[:kk]], n,
                                    [p[2].value for p in ptrs[:kk]], out_form=libff_amd.OUT_AFFINE, window_bits=wb)
            engine.synchronize()
            for j in range(kk):
                out = np.zeros(s["g_bytes"] // 8, dtype=np.uint64)
                engine.d2h(out, ptrs[j][2])
                assert (out == wants[j]).all(), (wb, kk, j)
    finally:
        for t in ptrs:
            for q in t:
                engine.free(q)


def test_multi_exp_batch_host_entry(engine, port):
    """amdmsm_multi_exp_batch (host vectors; what libff_amd::multi_exp_batch of the C++ shim calls): three pairs, one of the
    base vectors registered (resident) and two uploaded; each result against the oracle; then k = 1."""
    curve, group, n = 0, 1, 20011
    bases = [port.bases_seq(curve, group, n, first=11 * j) for j in range(3)]
    scs = [port.scalars_sha512(curve, 900 + j, n) for j in range(3)]
    wants = [port.multi_exp(curve, group, b, s_, port.BDLO12_SIGNED, port.FORM_NORMAL, chunks=8, omp=True) for b, s_ in zip(bases, scs)]
    h = engine.register_bases(curve, group, bases[2], libff_amd.multi_exp_base_form_normal)
    try:
        got = engine.multi_exp_batch(curve, group, bases, scs, base_form=libff_amd.multi_exp_base_form_normal)
        for j in range(3):
            assert (got[j] == wants[j]).all(), j
        got = engine.multi_exp_batch(curve, group, bases[:1], scs[:1], base_form=libff_amd.multi_exp_base_form_normal)
        assert (got[0] == wants[0]).all()
    finally:
        engine.unregister_bases(h)


def test_multi_exp_batch_with_base_cache_smaller_than_the_batch():
    """AMDMSM_BASE_CACHE_MB registers the base vectors a host call sees and evicts least-recently-used ones when the cap is
    reached.  A batch of four vectors against a cap that holds two of them (an entry is charged twice its affine bytes):
    registering the later vectors must not free the copies the batch already resolved for the earlier ones -- they are
    pinned for the duration of the call and the vectors that do not fit are uploaded like unregistered ones.  Run twice
    (the second call finds some vectors cached, in a different LRU order), each result against the oracle.  Child process:
    the cap is read once per process."""
    import subprocess

    code = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
import libff_amd
from oracle import port
port.build()
e = libff_amd.Engine(0)
curve, group, n, k = 0, 1, 1 << 14, 4          # 4 x 2 x 2^14 x 64 B = 8 MiB wanted, cap 5 MiB
bases = [port.bases_seq(curve, group, n, first=1000 * j) for j in range(k)]
scs = [port.scalars_sha512(curve, 300 + j, n) for j in range(k)]
wants = [port.multi_exp(curve, group, b, s, port.BDLO12_SIGNED, port.FORM_SPECIAL, chunks=8, omp=True) for b, s in zip(bases, scs)]
for rep in range(3):
    order = list(range(k)) if rep != 1 else [2, 0, 3, 1]
    got = e.multi_exp_batch(curve, group, [bases[j] for j in order], [scs[j] for j in order],
                            base_form=libff_amd.multi_exp_base_form_special)
    for pos, j in enumerate(order):
        assert (got[pos] == wants[j]).all(), (rep, j)
    # single calls in between reshuffle the LRU order
    assert (e.multi_exp(curve, group, bases[3], scs[3], base_form=libff_amd.multi_exp_base_form_special) == wants[3]).all()
print("cache-batch-ok")
''' % (REPO, os.path.join(REPO, "tests"))
    env = dict(os.environ, AMDMSM_BASE_CACHE_MB="5")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "cache-batch-ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
