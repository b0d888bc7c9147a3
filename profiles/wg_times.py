#!/usr/bin/env python3
"""profiles/wg_times.py <file>: per-workgroup start / end records of scan_gemm_f16x (bench.py --debug 16
with EXPANN_WG_TIMES=<file>): how many workgroups are resident over the launch, how long they run by
round, and what happens between one workgroup's end and the next one's start on the same CU."""
import sys
import numpy as np
a = np.loadtxt(sys.argv[1], dtype=np.uint64)
b, t0, t1, place = a[:, 0], a[:, 1].astype(np.int64), a[:, 2].astype(np.int64), a[:, 3]
ok = t1 > 0
b, t0, t1, place = b[ok], t0[ok], t1[ok], place[ok]
base = t0.min()
t0 = (t0 - base) / 100.0
t1 = (t1 - base) / 100.0   # us
dur = t1 - t0
print(f"{len(b)} workgroups, launch span {t1.max():.1f} us, mean duration {dur.mean():.1f} us "
      f"(min {dur.min():.1f}, max {dur.max():.1f}); sum(dur)/span = {dur.sum() / t1.max():.1f} resident on average")
order = np.argsort(t0)
for lo in range(0, len(b), 512):
    sel = order[lo:lo + 512]
    print(f"  start rank {lo:5d}..{lo + len(sel) - 1:5d}: starts {t0[sel].min():8.1f}..{t0[sel].max():8.1f} us, "
          f"duration mean {dur[sel].mean():7.1f} (p10 {np.percentile(dur[sel], 10):7.1f}, p90 {np.percentile(dur[sel], 90):7.1f})")
ts = np.linspace(0, t1.max(), 21)[1:-1]
print("  resident at", " ".join(f"{t:.0f}us:{int(((t0 <= t) & (t1 > t)).sum())}" for t in ts))
hw = (place & np.uint64(0xFFFFFFFF)).astype(np.int64)
xcc = (place >> np.uint64(32)).astype(np.int64) & 0xF
cu = (hw >> 8) & 0xF
sh = (hw >> 12) & 0x1
se = (hw >> 13) & 0x7
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
print(f"  distinct (xcc, se, sh, cu) places: {len(np.unique(key))}; workgroups per place: min {np.bincount(key)[np.unique(key)].min()}, "
      f"max {np.bincount(key)[np.unique(key)].max()}; per xcc: {np.bincount(xcc).tolist()}")
gaps = []
for k in np.unique(key):
    sel = np.where(key == k)[0]
    s = sel[np.argsort(t0[sel])]
    # two slots per place: a start is matched with the latest end before it
    ends = []
    for i in s:
        prev = [e for e in ends if e <= t0[i] + 1e-9]
        if prev and len(ends) >= 2:
            e = max(prev)
            gaps.append(t0[i] - e)
            ends.remove(e)
        ends.append(t1[i])
gaps = np.array(gaps)
if len(gaps):
    print(f"  gap between an end and the next start on the same CU: mean {gaps.mean():.1f} us, median {np.median(gaps):.1f}, "
          f"p90 {np.percentile(gaps, 90):.1f}, max {gaps.max():.1f} ({len(gaps)} gaps)")
