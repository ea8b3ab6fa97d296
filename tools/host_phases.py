"""Host-side time of each phase of an EM step on a tiny batch (no synchronisation inside the loop,
so these are the times the host spends issuing work), averaged over 300 steps."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings
from bayesiannetwork_amd.optimizers.em_step import FlatModel
from bayesiannetwork_amd.torch_settings import TorchSettings

ts = TorchSettings(device="cuda", dtype="float32")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
network, ys = bench.build_model(ts, "c3a")
ev = [bench.synthetic_evidence(ts, n, len(ys), 1 + b, torch.device("cuda", 0)) for b in range(2)]
settings = SpaInferenceMachineSettings(ts, True)
model = FlatModel.of(network)
acc = dict(factory=0.0, enter=0.0, stats=0.0, gather=0.0, m_step=0.0, assign=0.0, drop=0.0)
steps = 320
for i in range(steps):
    if i == 20:
        for k in acc: acc[k] = 0.0
        torch.cuda.synchronize(); t_all = time.perf_counter()
    t0 = time.perf_counter(); machine = HipSpaInferenceMachine(settings, network, ys, n)
    t1 = time.perf_counter()
    t2 = time.perf_counter(); ll, counts = machine.em_statistics(ev[i % 2])
    t3 = time.perf_counter(); flat = model.gather()
    t4 = time.perf_counter(); new = machine.m_step(counts, flat, 0.1, 0.01)
    t5 = time.perf_counter(); model.assign(new)
    t6 = time.perf_counter(); del machine
    t7 = time.perf_counter()
    for k, d in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t7 - t6)):
        acc[k] += d
issue = time.perf_counter() - t_all
torch.cuda.synchronize()
total = time.perf_counter() - t_all
print({k: round(v / (steps - 20) * 1e6, 1) for k, v in acc.items()}, "us per step; issue", round(issue / (steps - 20) * 1e6, 1), "total", round(total / (steps - 20) * 1e6, 1))
