import sys, os, json, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/tools")
import bench_matrix as bm
from bayesiannetwork_amd.torch_settings import TorchSettings
bm.ts = TorchSettings(device="cuda", dtype="float64")
n = 1 << 20
for name, loops in (("c1_docs1_f64", False), ("c2_docs2_f64", False), ("c4_loops_f64", True)):
    r = bm.resampled(name, n, loops=loops)
    print(name, round(r["ms"], 3), "ms", round(r["samples_per_s"] / 1e6, 1), "M/s")
