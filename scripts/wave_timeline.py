"""Reads a PT_WAVE_TRACE dump and prints, per extend launch, how many waves were resident over time."""
import sys
import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.uint64)
rounds, waves = int(raw[0]), int(raw[1])
rec = raw[2:].reshape(rounds, waves, 4)
for r in range(rounds):
    a = rec[r]
    a = a[a[:, 0] > 0]
    if len(a) == 0:
        continue
    t0 = a[:, 0].min()
    start = (a[:, 0] - t0) * 1e-2          # us
    exh = (a[:, 1] - t0) * 1e-2
    end = (a[:, 2] - t0) * 1e-2
    longest = (a[:, 3] & 0xffffffff).astype(np.int64)
    span = end.max()
    grid = np.linspace(0, span, 11)
    resident = [(int(((start <= t) & (end > t)).sum())) for t in grid[:-1]]
    print(f"launch {r}: waves {len(a)}, span {span:.0f} us, first exhausted at {exh.min():.0f} us, "
          f"median end {np.median(end):.0f} us, p90 end {np.percentile(end, 90):.0f}, p99 end {np.percentile(end, 99):.0f}, "
          f"started late (>10us) {int((start > 10).sum())}, longest ray {longest.max()} steps")
    print("   resident at 0%..90% of span:", resident)
