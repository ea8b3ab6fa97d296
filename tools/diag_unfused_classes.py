"""Kernel table of one EM step of Q(classes) -> 784 x Y(2 | Q) at 60 000 observations (for class counts off the fused path)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import latent_classes as L
from bayesiannetwork_amd.inference_machines.hip_spa.hip_spa_inference_machine import enable_profiling, profile_report

classes = int(sys.argv[1])
orig = L.profile_report
captured = {}
def spy():
    r = orig()
    captured["r"] = r
    return r
L.profile_report = spy
print(json.dumps(L.run(classes, steps=3)))
for r in sorted(captured["r"], key=lambda r: -r["total_ms"]):
    print(r["kernel"], r["launches"], round(r["total_ms"], 3))
