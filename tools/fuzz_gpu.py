#!/usr/bin/env python3
"""One-off wider sweeps of the seeded GPU parity checks (same checkers as tests/test_gpu_parity.py, other seeds and
larger shape ranges):  python tools/fuzz_gpu.py [seed] [count] [seconds]
Prints one line per failure and a summary; exit status 1 if anything failed.

Known degenerate class (seen with seeds 31, 2028, 31337; not a kernel error): whole-step shapes with latent_size 1
can fail on 'be' -- the encoder bias gradient is then ONE number, a sum over all rows that nearly cancels, and the
check's max|a - b| / max|b| is the relative error of that single sum (the strict fp32 path shows the same figure).
Found by this tool and fixed: seed 4242, row blocks of whole workgroups out of a padded narrow point set (mmd_n16.hip)."""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T  # noqa: E402
import test_gpu_r3 as R3  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 240.0
rs = np.random.RandomState(seed)
t_end = time.time() + budget
fails, ran = [], 0


def attempt(label, fn, *args):
    global ran
    ran += 1
    try:
        fn(*args)
    except Exception as e:      # AssertionError included
        fails.append((label, args, repr(e)[:300]))
        print("FAIL", label, args, repr(e)[:300], flush=True)
        traceback.print_exc(limit=2)


for i in range(count):
    if time.time() > t_end:
        break
    # whole steps: wider ranges than the committed sweep (larger batches, wide hidden sizes, n > S, L == d)
    B = int(rs.randint(1, 200))
    S = int(rs.randint(1, 40))
    d = int(rs.choice([8, 24, 33, 64, 100, 256, 515, 768, 1024, 1030]))
    n = int(rs.randint(1, 34))
    L = int(rs.choice([1, 2, 3, 7, 16, 33, 77, 100, 130, 200, 300, 512, 768])) if rs.rand() > 0.15 else d
    dec = str(rs.choice(["", "no_norm"]))
    attempt("step", T.test_random_shapes_whole_step_against_fp64_oracle, B, S, d, n, L, dec)
    # row blocks: every kernel family, unaligned ranges
    N = int(rs.randint(2, 6000))
    D = int(rs.choice([1, 4, 15, 17, 31, 64, 77, 81, 96, 129, 160, 192, 193, 255, 256, 257, 400, 513, 640, 769, 1000, 1024, 1030]))
    lo = int(rs.randint(0, N))
    hi = int(rs.randint(lo + 1, N + 1))
    attempt("block", T.test_random_row_blocks_against_fp64_oracle, N, D, lo, hi)
    # the three-stage (row-sharded) entry points against the one-call form and fp64: odd rank counts, ragged row counts
    world = int(rs.choice([2, 3, 4, 5, 8]))
    n_loc = int(rs.randint(1, 900))
    Ds = int(rs.choice([3, 16, 64, 77, 130, 192, 193, 256, 300, 384, 400, 512, 640, 768, 769, 1000, 1024]))
    kind = str(rs.choice(["f32", "f32", "strict", "bf16"]))
    attempt("staged", R3.test_staged_entry_points_equal_the_one_call_form, (world, world * n_loc, Ds, kind))
    if i % 10 == 0:
        print(f"[{i}] ran {ran} failures {len(fails)}", flush=True)
print(f"done: ran {ran}, failures {len(fails)}")
sys.exit(1 if fails else 0)
