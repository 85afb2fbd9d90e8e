#!/usr/bin/env python3
"""Randomised parity soak: random group, size, scalar pattern, window size, base form and
chunking; the engine (through the C ABI) against the CPU oracle.  Test infrastructure.

  python tools/fuzz_parity.py --seconds 240 --seed 1
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import libff_amd  # noqa: E402
from common import GROUPS, small_scalars_mont  # noqa: E402
from oracle import port  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-n", type=int, default=4000)
    ap.add_argument("--mode", choices=("multi_exp", "precomputed", "batch"), default="multi_exp")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    eng = libff_amd.Engine(0)
    t_end = time.time() + args.seconds
    it = 0
    fails = 0
    import tempfile
    tmpdir = tempfile.mkdtemp(prefix="amdmsm_fuzz_")
    while args.mode == "precomputed" and time.time() < t_end:
        # multi_exp_stream_with_precompute: device-built table == oracle table, file entry == oracle
        name, curve, group = GROUPS[rng.integers(len(GROUPS))]
        heavy = group == 2 or curve == 2
        n = int(rng.integers(1, (300 if heavy else 1500) + 1))
        c = int(rng.integers(3, 14))
        D = libff_amd.precompute_num_digits(curve, c) + int(rng.integers(2))
        sc = port.scalars_sha512(curve, int(rng.integers(1 << 30)), n)
        if rng.random() < 0.3:
            sc[rng.random(n) < 0.7] = sc[0]
        bases = port.bases_seq(curve, group, n, first=int(rng.integers(0, 1 << 20)))
        if n >= 3 and rng.random() < 0.3:
            bases[2] = port.group_consts(curve, group)[1]
        tab = eng.precompute_table(curve, group, bases, c, num_digits=D)
        ok = (tab == port.precompute_table(curve, group, bases, c, num_digits=D)).all()
        want = port.multi_exp_precompute(curve, group, tab, sc, c, num_digits=D)
        if D == libff_amd.precompute_num_digits(curve, c):
            path = os.path.join(tmpdir, "t.bin")
            open(path, "wb").write(port.disk_write(curve, group, tab).tobytes())
            got = eng.multi_exp_stream_with_precompute_file(curve, group, path, sc, c,
                                                            chunk_points=int(rng.choice([0, 0, 17, 256])))
            ok = ok and (got == want).all()
        it += 1
        if not ok:
            fails += 1
            print(f"MISMATCH it={it} {name} n={n} c={c} D={D}", flush=True)
        if it % 25 == 0:
            print(f"[fuzz precomputed] {it} cases, {fails} mismatches", flush=True)
    while args.mode == "batch" and time.time() < t_end:
        # amdmsm_multi_exp_batch: k pairs of one group and length, every result against the oracle's single multi_exp
        name, curve, group = GROUPS[rng.integers(len(GROUPS))]
        heavy = group == 2 or curve == 2
        n = int(rng.integers(1, (args.max_n // 4 if heavy else args.max_n) + 1))
        k = int(rng.integers(1, 9))
        form = int(rng.integers(2))
        bl, sl = [], []
        for j in range(k):
            sc = port.scalars_sha512(curve, int(rng.integers(1 << 30)), n)
            r = rng.random()
            if r < 0.2:
                sc[rng.random(n) < 0.7] = sc[0]
            elif r < 0.4:
                sm = small_scalars_mont(port, curve, [0, 1, 2])
                pick = rng.integers(0, 6, size=n)
                for v in range(3):
                    sc[pick == v] = sm[v]
            b = port.bases_seq(curve, group, n, first=int(rng.integers(0, 1 << 20))) if rng.random() < 0.6 else \
                np.roll(port.bases_r32(curve, group, n), int(rng.integers(32)), axis=0)
            if n >= 3 and rng.random() < 0.2:
                b[2] = port.group_consts(curve, group)[1]
            bl.append(b)
            sl.append(sc)
        c = int(rng.choice([0, 0, 0, 3, 7, 10, 11, 13, 16]))
        eng.endomorphism = int(rng.choice([0, 1, 2, -1]))
        if rng.random() < 0.3 and n >= 1:
            h = eng.register_bases(curve, group, bl[0], form)
        else:
            h = None
        got = eng.multi_exp_batch(curve, group, bl, sl, base_form=form, window_bits=c)
        if h is not None:
            eng.unregister_bases(h)
        it += 1
        for j in range(k):
            want = port.multi_exp(curve, group, bl[j], sl[j], port.BDLO12_SIGNED, 1, chunks=4, omp=True)
            if not (got[j] == want).all():
                fails += 1
                print(f"MISMATCH it={it} {name} n={n} k={k} j={j} c={c} form={form} endomorphism={eng.endomorphism}", flush=True)
        if it % 25 == 0:
            print(f"[fuzz batch] {it} cases, {fails} mismatches", flush=True)
    while args.mode == "multi_exp" and time.time() < t_end:
        name, curve, group = GROUPS[rng.integers(len(GROUPS))]
        heavy = group == 2 or curve == 2
        n = int(rng.integers(1, (args.max_n // 4 if heavy else args.max_n) + 1))
        pattern = int(rng.integers(6))
        sc = port.scalars_sha512(curve, int(rng.integers(1 << 30)), n)
        if pattern == 1:      # witness-like: 0 / 1 / small / random
            pick = rng.integers(0, 5, size=n)
            sm = small_scalars_mont(port, curve, [0, 1, 2, 3])
            for v in range(4):
                sc[pick == v] = sm[v]
        elif pattern == 2:    # one value repeated
            sc[rng.random(n) < 0.8] = sc[0]
        elif pattern == 3:    # all equal
            sc[:] = sc[0]
        elif pattern == 4:    # small scalars only
            vals = [int(x) for x in rng.integers(0, 1 << 20, size=n)]
            sc = small_scalars_mont(port, curve, vals)
        first = int(rng.integers(0, 1 << 20))
        bases = port.bases_seq(curve, group, n, first=first) if rng.random() < 0.6 else port.bases_r32(curve, group, n)
        if pattern == 5 and n >= 4:   # equal and opposite bases, an infinite base
            bases[1] = bases[0]
            bases[2] = port.group_op(curve, group, 3, bases[0])
            bases[3] = port.group_consts(curve, group)[1]
        c = int(rng.choice([0, 0, 0, 2, 3, 5, 7, 9, 11, 13, 16]))
        # amdmsm_opts.endomorphism: every base here is a multiple of the generator, so all modes must agree
        eng.endomorphism = int(rng.choice([0, 1, 2, 2, -1]))
        form = int(rng.integers(2))
        chunks = int(rng.choice([1, 1, 2, 3, 8]))
        want = port.multi_exp(curve, group, bases, sc, port.BDLO12_SIGNED, 1, chunks=4, omp=True)
        got = eng.multi_exp(curve, group, bases, sc, base_form=form, window_bits=c, chunks=chunks, split_chunks=True)
        it += 1
        if not (got == want).all():
            fails += 1
            print(f"MISMATCH it={it} {name} n={n} pattern={pattern} c={c} form={form} chunks={chunks} first={first} "
                  f"endomorphism={eng.endomorphism}", flush=True)
        if it % 50 == 0:
            print(f"[fuzz] {it} cases, {fails} mismatches", flush=True)
    print(f"[fuzz] done: {it} cases, {fails} mismatches")
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
