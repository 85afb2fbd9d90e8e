"""Range-sharded MSM across the GPUs of one node.

libff::multi_exp already shards by contiguous input range and sums the partial
results serially (multiexp.tcc:663-687: ``one = total / chunks``, the last chunk takes
the remainder).  Here a chunk is a rank: one process per GPU, each rank reduces its
range to ONE engine-Jacobian point on its own device, the partials are exchanged with
a single all-gather (RCCL over xGMI; 3*coord bytes per rank, latency-bound) and every
rank sums the world_size partials locally -- all-gather + local reduce = the
"all-reduce of partial sums" (EC addition is not an RCCL reduction operator).

The two callables make the exchange testable on CPU with the gloo backend
(tests/test_distributed_cpu.py supplies oracle-backed ones); on a GPU box they default
to the HIP engine.
"""
import numpy as np


def shard_range(total, world_size, rank):
    """Contiguous range of rank ``rank``: multiexp.tcc:663, 675-678."""
    one = total // world_size
    lo = rank * one
    hi = total if rank == world_size - 1 else (rank + 1) * one
    return lo, hi


def all_gather_partials(partial_words, group=None, out=None):
    """all-gather one partial point (1-D integer tensor) from every rank -> (world, words).

    ``out``: a preallocated contiguous (world, words) tensor on the partial's device that receives the points
    (one ``all_gather_into_tensor``: no per-step allocation, no list of N tensors, no ``torch.stack`` -- the payload
    is 96-288 bytes per rank, so those were most of the step's exchange time)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if partial_words.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal of the multi-rank path with gloo (several ranks sharing one GPU, where RCCL
        # refuses duplicate devices): stage through the host.  .cpu() waits for the producing stream.
        host = partial_words.cpu()
        gathered = torch.empty((world, host.numel()), dtype=host.dtype)
        dist.all_gather_into_tensor(gathered.view(-1), host, group=group)   # flat: gloo accepts only the 1-D form
        if out is not None:
            out.copy_(gathered)
            return out
        return gathered.to(partial_words.device, non_blocking=False)
    if out is None:
        out = torch.empty((world, partial_words.numel()), dtype=partial_words.dtype, device=partial_words.device)
    dist.all_gather_into_tensor(out.view(-1), partial_words, group=group)
    return out


def sharded_multi_exp(local_msm, combine, group=None):
    """Run ``local_msm()`` (this rank's partial, engine-Jacobian words as a torch tensor on
    the rank's device), exchange, and return ``combine(stacked_partials)``."""
    partial = local_msm()
    stacked = all_gather_partials(partial, group=group)
    return combine(stacked)


class ShardedMsm:
    """Device-resident sharded MSM for one (curve, group): each rank holds its own range of
    compact-affine bases and Montgomery scalars in HBM as torch tensors.

    ``depth`` > 1 keeps that many MSMs in flight: step k runs on stream ``k % depth`` with its own
    workspace slot and result buffers, so the few-wave tail (bucket reduction, Horner) of one
    step overlaps the bulk kernels of the next -- the way a prover issues its back-to-back MSMs.
    """

    def __init__(self, engine, curve, group_id, process_group=None, depth=1, force_exchange=False):
        """force_exchange: run the exchange step (all-gather of the partial point on the device tensor +
        k_sum_points) even at world size 1, where it is otherwise skipped -- so that a one-GPU box
        executes the RCCL call and the combining kernel of the multi-rank path."""
        import torch

        self.torch = torch
        self.engine = engine
        self.curve = curve
        self.group_id = group_id
        self.pg = process_group
        self.depth = depth
        self.force_exchange = force_exchange
        from .engine import sizes

        self.sz = sizes(curve, group_id)
        dev = torch.device("cuda", engine.device)
        words = self.sz["g_bytes"] // 8
        self.partial = [torch.zeros(words, dtype=torch.int64, device=dev) for _ in range(depth)]
        self.result = [torch.zeros(words, dtype=torch.int64, device=dev) for _ in range(depth)]
        # receive buffers of the exchange, one per step in flight, owned here: (world, words), filled by ONE
        # all_gather_into_tensor per step (allocated at first use, when the process group is known to be up)
        self.gathered = [None] * depth
        # Always explicit, non-default streams: torch's default stream has handle 0, which the C ABI
        # reads as "the context's own stream" -- a non-blocking stream that does not synchronise
        # with torch's legacy null stream, so the all-gather (ordered by torch's current stream)
        # could read the partial before the MSM has written it.
        self.streams = [torch.cuda.Stream(dev) for _ in range(depth)]
        self.k = 0
        engine.set_pipeline_depth(depth)

    def run(self, bases_affine, scalars, n, out_form, window_bits=0):
        """bases_affine / scalars: this rank's shard (torch tensors on its GPU).  Returns
        (result tensor, workspace slot).  The MSM, the all-gather (RCCL orders itself on torch's
        current stream, which is the step's stream inside the ``with`` block) and the final sum all
        run on the step's own stream; the result is ready once that stream has been synchronised
        (``synchronize()``) or waited for (``streams[slot]``)."""
        from .engine import OUT_JACOBIAN
        import torch.distributed as dist

        torch = self.torch
        i = self.k % self.depth
        self.k += 1
        stream = self.streams[i]
        assert stream.cuda_stream != 0
        partial, result = self.partial[i], self.result[i]
        # inputs were produced on the caller's current stream
        stream.wait_stream(torch.cuda.current_stream(stream.device))
        multi = dist.is_available() and dist.is_initialized() and (dist.get_world_size(self.pg) > 1 or self.force_exchange)
        with torch.cuda.stream(stream):
            if not multi:
                # one rank: the MSM writes the requested form itself, nothing to exchange or sum
                self.engine.msm_device(self.curve, self.group_id, bases_affine.data_ptr(), scalars.data_ptr(), n,
                                       result.data_ptr(), out_form=out_form, window_bits=window_bits,
                                       stream=stream.cuda_stream)
                return result, self.engine.last_slot()
            self.engine.msm_device(self.curve, self.group_id, bases_affine.data_ptr(), scalars.data_ptr(), n,
                                   partial.data_ptr(), out_form=OUT_JACOBIAN, window_bits=window_bits,
                                   stream=stream.cuda_stream)
            slot = self.engine.last_slot()
            stacked = all_gather_partials(partial, group=self.pg, out=self._gather_buf(i))
            self.engine.sum_points_device(self.curve, self.group_id, stacked.data_ptr(), stacked.shape[0], out_form,
                                          result.data_ptr(), stream=stream.cuda_stream)
        return result, slot

    def _gather_buf(self, i):
        import torch.distributed as dist

        if self.gathered[i] is None:
            world = dist.get_world_size(self.pg)
            self.gathered[i] = self.torch.zeros((world, self.partial[i].numel()), dtype=self.partial[i].dtype,
                                                device=self.partial[i].device)
        return self.gathered[i]

    def exchange_only(self, out_form):
        """The exchange step alone on slot 0's buffers (all-gather of the current partial + local sum):
        what bench.py times to split a sharded step into shard time and exchange latency."""
        torch = self.torch
        stream = self.streams[0]
        with torch.cuda.stream(stream):
            stacked = all_gather_partials(self.partial[0], group=self.pg, out=self._gather_buf(0))
            self.engine.sum_points_device(self.curve, self.group_id, stacked.data_ptr(), stacked.shape[0], out_form,
                                          self.result[0].data_ptr(), stream=stream.cuda_stream)
        return self.result[0]

    def synchronize(self):
        for s in self.streams:
            s.synchronize()


def numpy_words(t):
    return np.ascontiguousarray(t.detach().cpu().numpy()).view(np.uint64)
