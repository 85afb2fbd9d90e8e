#!/usr/bin/env bash
# Run on the MI355X box (through gpurun): rocprofv3 kernel-trace statistics and the two PMC passes
# (FETCH_SIZE, WRITE_SIZE -- separate runs, --kernel-trace only) of `python3 bench.py` at the two sizes
# of the metric; raw outputs under gpurun_out/prof_<tag>/, tracked summaries under profiles/.
#   tools/collect_profiles.sh r02
set -uo pipefail
tag="${1:-r02}"
R="${GRAFT_REPO_ROOT:-$(pwd)}"
cd /tmp && export TMPDIR=/tmp
for L in 20 26; do
  steps=5; [ "$L" = 26 ] && steps=3
  args="--steps $steps --warmup 1 --no-cpu-baseline --no-legs --log2n $L"
  out="$R/gpurun_out/prof_${tag}_2p$L"
  rm -rf "$out"; mkdir -p "$out"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 "$R/bench.py" $args > "$out/bench_under_rocprof.json" 2> "$out/trace.err"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/fetch" -- python3 "$R/bench.py" $args > /dev/null 2> "$out/fetch.err"
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/write" -- python3 "$R/bench.py" $args > /dev/null 2> "$out/write.err"
  echo "2^$L collected: $(ls "$out")"
done
cd "$R"
for L in 20 26; do
  out="gpurun_out/prof_${tag}_2p$L"
  stats=$(find "$out/trace" -name "*kernel_stats.csv" | head -1)
  fetch=$(find "$out/fetch" -name "*counter_collection.csv" | head -1)
  write=$(find "$out/write" -name "*counter_collection.csv" | head -1)
  c=$(python3 -c "import json;print(json.load(open('$out/bench_under_rocprof.json'))['config']['window_bits'])")
  python3 tools/summarize_profiles.py "${tag}_alt_bn128_g1_2p$L" "$stats" "$fetch" "$write" alt_bn128 1 $L "$c"
  cp "$out/bench_under_rocprof.json" "profiles/${tag}_bench_2p${L}_under_rocprof.json"
  mkdir -p "gpurun_out/profiles_${tag}"
  cp "profiles/${tag}_alt_bn128_g1_2p${L}_kernel_stats.csv" "profiles/${tag}_alt_bn128_g1_2p${L}_pmc.json" "profiles/${tag}_bench_2p${L}_under_rocprof.json" "gpurun_out/profiles_${tag}/"
done
