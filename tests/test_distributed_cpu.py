"""World-size-2 (and 3) rehearsal of the multi-GPU exchange on CPU with the gloo backend.

The exchange is libff's own chunking (multiexp.tcc:663-687): rank r reduces the
contiguous range [r*one, (r+1)*one) (last rank takes the remainder) to one partial point,
one all-gather moves the partials, every rank sums them.  On a GPU box the two callables
are the HIP engine; here they are backed by the oracle so the plumbing (range split,
all-gather layout, combine order) is checked without a device."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port_no, curve, group, n, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import port
    from libff_amd.distributed import shard_range, sharded_multi_exp

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port_no)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_range(n, world, rank)
        bases = port.bases_seq(curve, group, hi - lo, first=lo)        # this rank's shard only
        scalars = port.scalars_sha512(curve, lo, hi - lo)
        zero = port.group_consts(curve, group)[1]

        def local_msm():
            # partial in (X, Y, Z); Jacobian/projective as the oracle produces it
            r = port.multi_exp(curve, group, bases, scalars, port.BDLO12_SIGNED, 1) if hi > lo else zero
            return torch.from_numpy(r.view(np.int64).copy())

        def combine(stacked):
            acc = zero
            for k in range(stacked.shape[0]):
                acc = port.group_op(curve, group, 0, acc, stacked[k].numpy().view(np.uint64))
            return port.group_op(curve, group, 4, acc)

        res = sharded_multi_exp(local_msm, combine)
        # the form ShardedMsm uses: one all_gather_into_tensor into a buffer the caller owns (no per-step allocation)
        from libff_amd.distributed import all_gather_partials

        mine = local_msm()
        buf = torch.full((world, mine.numel()), -1, dtype=mine.dtype)
        got = all_gather_partials(mine, out=buf)
        assert got.data_ptr() == buf.data_ptr() and (buf[rank] == mine).all() and (combine(buf) == res).all()
        q.put((rank, lo, hi, res.tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,curve,group,n", [(2, 0, 1, 301), (3, 1, 1, 100), (2, 2, 1, 1), (8, 0, 1, 203)])   # 8: the node the bench targets
def test_sharded_msm_gloo(port, world, curve, group, n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port_no = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port_no, curve, group, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = port.multi_exp(curve, group, port.bases_seq(curve, group, n), port.scalars_sha512(curve, 0, n),
                          port.BDLO12_SIGNED, 1)
    ranges = sorted((lo, hi) for _, lo, hi, _ in results)
    assert ranges[0][0] == 0 and ranges[-1][1] == n
    assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
    for _, _, _, res in results:
        assert (np.array(res, dtype=np.uint64) == want).all()   # every rank holds the full result


def test_shard_range_matches_libff_chunking():
    from libff_amd.distributed import shard_range

    for total in (0, 1, 5, 8, 257, 1 << 20):
        for world in (1, 2, 3, 4, 8):
            one = total // world
            for r in range(world):
                lo, hi = shard_range(total, world, r)
                assert lo == r * one
                assert hi == (total if r == world - 1 else (r + 1) * one)


def test_bench_self_launch_propagates_child_failure():
    """`python bench.py --gpus 2` with no launcher around it starts its two ranks itself
    (torch.distributed.run children; the parent never touches a GPU).  On this GPU-less host
    both ranks stop with the "needs an MI355X" message and the parent must exit non-zero."""
    import subprocess
    import sys

    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the ranks would run the benchmark")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert "needs an MI355X" in (r.stdout + r.stderr)
