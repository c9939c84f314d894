"""Histogram of tip-set groups per read (a CLS_PROFILE_HOOKS build run with CLS_PROFILE_STOP=2 writes the count into `one`)."""
import sys
import numpy as np

rec = np.load(sys.argv[1])
st = rec["status"]
n = rec["one"][st == 0xFE]
print("reads", len(rec), "with groups", len(n))
for q in (1, 10, 25, 50, 75, 90, 99):
    print(f"p{q}: {np.percentile(n, q):.0f}", end="  ")
print()
for t in (64, 96, 112, 128, 139, 192):
    print(f"<= {t}: {np.mean(n <= t):.3f}", end="  ")
print()
