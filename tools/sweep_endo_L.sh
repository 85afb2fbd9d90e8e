set -e
for spec in "alt_bn128 1 20" "alt_bn128 1 16" "bls12_377 1 20" "bw6_761 1 20" "alt_bn128 2 20" "bls12_377 2 20"; do
  set -- $spec
  for L in 2 4 8 16 32; do
    python tools/sweep_c.py --curve $1 --group $2 --log2n $3 --c 0 0 --endo 1 --segment-len $L 2>/dev/null | tail -1 | sed "s/^/$1 G$2 L=$L /"
  done
done
