"""Per-workgroup clock stamps of the IVF score-segment scan (diagnostic build MRAG_IVFS_DIAG & 128, MRAG_IVFS_STAMPS=<file>):
phase durations in core clocks, medians over the workgroups."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.int64).reshape(-1, 8)
a = a[a[:, 0] != 0]
steps = a[:, 6]
ph = {"prologue (desc, gq, first loads issued)": a[:, 1] - a[:, 0], "first stage lands": a[:, 2] - a[:, 1],
      "K loop (to the last tile's last MFMA)": a[:, 3] - a[:, 2], "store issue": a[:, 4] - a[:, 3], "store drain (vmcnt(0))": a[:, 5] - a[:, 4],
      "whole": a[:, 5] - a[:, 0]}
print("workgroups %d, K steps median %d" % (len(a), np.median(steps)))
for k, v in ph.items():
    print("  %-45s median %8.0f  mean %8.0f  p90 %8.0f" % (k, np.median(v), v.mean(), np.percentile(v, 90)))
print("  K loop cycles per step: median %.0f" % np.median((a[:, 3] - a[:, 2]) / np.maximum(steps, 1)))
print("  span of all stamps: %.0f cycles" % (a[:, 5].max() - a[:, 0].min()))
