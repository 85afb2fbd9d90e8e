#!/usr/bin/env bash
# Run on the MI355X box (through gpurun): rocprofv3 kernel-trace statistics and the two PMC passes
# (FETCH_SIZE, WRITE_SIZE -- separate runs, --kernel-trace only, the program directly after `--`)
# of `python3 bench.py --no-legs` for every BASELINE configuration that fits one GPU; raw outputs
# under gpurun_out/prof_<tag>_<cfg>/, tracked summaries under profiles/.
#   tools/collect_profiles.sh r03 [curve:group:log2n:steps[:endomorphism] ...]
# (endomorphism = 1: the split is permitted, as the configs[4] legs of bench.py run it)
set -uo pipefail
tag="${1:-r03}"
shift || true
cfgs=("$@")
if [ "${#cfgs[@]}" = 0 ]; then
  cfgs=(alt_bn128:1:20:5 alt_bn128:1:26:3 bls12_377:1:22:3 bw6_761:1:21:3:1 bls12_377:2:21:3:1)
fi
R="${GRAFT_REPO_ROOT:-$(pwd)}"
cd /tmp && export TMPDIR=/tmp
for cfg in "${cfgs[@]}"; do
  IFS=: read -r curve group L steps endo <<< "$cfg"
  name="${curve}_g${group}_2p$L"
  args="--endomorphism ${endo:-0} "
  args+="--steps $steps --warmup 1 --no-cpu-baseline --no-legs --curve $curve --group $group --log2n $L"
  out="$R/gpurun_out/prof_${tag}_$name"
  rm -rf "$out"; mkdir -p "$out"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 "$R/bench.py" $args > "$out/bench_under_rocprof.json" 2> "$out/trace.err"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/fetch" -- python3 "$R/bench.py" $args > /dev/null 2> "$out/fetch.err"
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/write" -- python3 "$R/bench.py" $args > /dev/null 2> "$out/write.err"
  echo "$name collected: $(ls "$out")"
done
cd "$R"
mkdir -p "gpurun_out/profiles_${tag}"
for cfg in "${cfgs[@]}"; do
  IFS=: read -r curve group L steps endo <<< "$cfg"
  name="${curve}_g${group}_2p$L"
  out="gpurun_out/prof_${tag}_$name"
  stats=$(find "$out/trace" -name "*kernel_stats.csv" | head -1)
  fetch=$(find "$out/fetch" -name "*counter_collection.csv" | head -1)
  write=$(find "$out/write" -name "*counter_collection.csv" | head -1)
  c=$(python3 -c "import json;print(json.load(open('$out/bench_under_rocprof.json'))['config']['window_bits'])")
  python3 tools/summarize_profiles.py "${tag}_$name" "$stats" "$fetch" "$write" "$curve" "$group" "$L" "$c"
  cp "$out/bench_under_rocprof.json" "profiles/${tag}_${name}_bench_under_rocprof.json"
  cp "profiles/${tag}_${name}_kernel_stats.csv" "profiles/${tag}_${name}_pmc.json" "profiles/${tag}_${name}_bench_under_rocprof.json" "gpurun_out/profiles_${tag}/"
done
