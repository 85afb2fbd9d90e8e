#!/bin/bash
# kernel-trace statistics of one tools/sweep_c.py invocation:  tools/ktrace.sh <tag> <sweep_c.py arguments ...>
# -> gpurun_out/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats) and the sweep's own phase line in gpurun_out/<tag>.log
set -e
tag=$1; shift
repo=$(cd "$(dirname "$0")/.." && pwd)
out=$repo/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o $tag --output-format csv -- python3 $repo/tools/sweep_c.py "$@" > $out/$tag.log 2>&1
find /tmp/prof_$tag -name "*kernel_stats.csv" -exec cp {} $out/${tag}_kernel_stats.csv \;
head -25 $out/${tag}_kernel_stats.csv | cut -c1-150
tail -3 $out/$tag.log
