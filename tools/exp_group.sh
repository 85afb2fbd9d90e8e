#!/usr/bin/env bash
# Rebuild ONE group with extra -D flags and time one configuration of tools/bench_configs.py.
#   tools/exp_group.sh <tag> <group> "<flags>" <curve:group:log2n> [more configs]
set -euo pipefail
tag="$1"; group="$2"; flags="$3"; shift 3
export AMDMSM_GROUPS="$group" AMDMSM_EXTRA_FLAGS="$flags"
python -m libff_amd.build --force > /dev/null
echo "== $tag ($group, $flags)"
python tools/bench_configs.py "$@" 2>/dev/null
