"""EM step time of the 784-leaf model at a small batch size (median of 5 repetitions after 2 warm-ups).

    python tools/small_batch.py 100 [steps]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import bench_matrix  # noqa: E402

n = int(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
r = bench_matrix.em_throughput(n, steps)
print(f"N={n} ms_per_step={r['ms_per_step']:.4f} samples_per_s={r['samples_per_s']:.0f}")
