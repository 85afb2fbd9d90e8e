#!/bin/bash
# window sizes 9..16 for small and medium inputs (round 2 excluded 9..15 below 2^16 points; the reduction kernels have changed since)
out=gpurun_out/exp_e.log; : > $out
for cfg in "alt_bn128 1 0" "alt_bn128 2 0" "alt_bn128 2 1" "bls12_377 1 0" "bls12_377 1 1" "bw6_761 1 1" "bls12_377 2 1"; do
  set -- $cfg
  echo "== $1 G$2 endo=$3" >> $out
  python tools/sweep_c.py --curve $1 --group $2 --endo $3 --log2n 2 6 10 12 13 14 15 16 17 --c 0 9 10 11 12 13 14 15 16 2>/dev/null | cut -c1-60 >> $out
done
python - <<'EOF'
import re
cur=None; data={}
for ln in open('gpurun_out/exp_e.log'):
    if ln.startswith('=='): cur=ln.strip('= \n'); data[cur]={}; continue
    m=re.match(r'n=2\^(\d+) (\w+) c=\s*(\d+) W=\s*(\d+) total=\s*([\d.]+)',ln)
    if m: data[cur].setdefault(int(m[1]),[]).append((int(m[3]),int(m[4]),float(m[5])))
for k,v in data.items():
    print(k)
    for lg,rows in v.items():
        print('  2^%d: planner c=%d (%.3f) | '%(lg,rows[0][0],rows[0][2]) + ' '.join('c%d:%.3f'%(c,t) for c,W,t in rows[1:]))
EOF
