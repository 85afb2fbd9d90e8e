#!/usr/bin/env python3
"""Back-to-back MSMs with several in flight (the `pipelined` leg of bench.py alone), for kernel traces:
  rocprofv3 --kernel-trace --output-format csv -d out -- python3 tools/pipelined_probe.py [log2n] [depth] [steps]
and tools/pipelined_timeline.py out/**/*kernel_trace.csv prints who ran beside whom."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libff_amd  # noqa: E402
from bench import gen_inputs  # noqa: E402
from libff_amd.distributed import ShardedMsm  # noqa: E402


def main():
    log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    depth = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    curve, group, n = 0, 1, 1 << log2n
    dev = torch.device("cuda", 0)
    eng = libff_amd.Engine(0)
    bases, scalars = gen_inputs(eng, curve, group, 0, n, dev, 1)
    msm = ShardedMsm(eng, curve, group, depth=depth)
    for _ in range(2 * depth):
        msm.run(bases, scalars, n, libff_amd.OUT_LIBFF)
    msm.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        msm.run(bases, scalars, n, libff_amd.OUT_LIBFF)
    t_host = time.perf_counter() - t0
    msm.synchronize()
    dt = time.perf_counter() - t0
    print(f"2^{log2n} points, {depth} in flight: {dt / steps * 1e3:.3f} ms per MSM over {steps} steps "
          f"(host enqueue {t_host / steps * 1e3:.3f} ms per MSM)")


if __name__ == "__main__":
    main()
