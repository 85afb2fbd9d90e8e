#!/bin/bash
# one-box timing round of the current build: headline sizes, shard, wide groups (tools/sweep_c.py, tools/bench_configs.py)
out=gpurun_out/${1:-ab}.log
python tools/sweep_c.py --log2n 20 23 26 --c 0 > $out 2>&1
python tools/sweep_c.py --log2n 23 --c 17 --endo -1 >> $out 2>&1
python tools/bench_configs.py bw6_761:1:21 bls12_377:2:21 bls12_377:1:22 alt_bn128:2:20 >> $out 2>&1
grep -v amdgpu.ids $out
