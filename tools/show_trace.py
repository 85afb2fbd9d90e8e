#!/usr/bin/env python3
"""Steady-state slice of a rocprofv3 kernel trace of tools/pipelined_probe.py: every kernel (memsets included) between the
fifth-last and the third-last k_accumulate with start / end / duration in us and its hardware queue.

  python3 tools/show_trace.py <directory given to rocprofv3 -d>
"""
import csv,re,glob,sys
f=sorted(glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True))[-1]
rows=[]
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Queue_Id","?")))
rows.sort()
acc=[i for i,r in enumerate(rows) if "k_accumulate(" in r[2]]
i0=acc[-5]
t0=rows[i0][0]
for r in rows[i0:acc[-3]+2]:
    print(f"{(r[0]-t0)/1e3:9.1f} {(r[1]-t0)/1e3:9.1f} {(r[1]-r[0])/1e3:8.1f} q{r[3]} {r[2].replace('amdmsm::(anonymous namespace)::','')[:40]}")
