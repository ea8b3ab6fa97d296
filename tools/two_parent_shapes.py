"""Time one statistics pass (T = 10 iterations + log-likelihood + expected counts) of a loopy model
Q(10) -> F1(d0), F2(d1) -> 784 x Y(s | F1, F2) at 60 000 observations for several (d0, d1, s).
Shapes with a fused two-parent kernel run on it, the others on the unfused kernels:
    python tools/two_parent_shapes.py 2,2,2 3,3,3 3,3,4
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bayesiannetwork_amd.bayesian_network import BayesianNetworkBuilder, Node
from bayesiannetwork_amd.inference_machines.evidence import Evidence
from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings
from bayesiannetwork_amd.inference_machines.hip_spa.hip_spa_inference_machine import enable_profiling, profile_report
from bayesiannetwork_amd.torch_settings import TorchSettings


def rand_cpt(gen, shape, ts):
    p = torch.rand(shape, generator=gen, dtype=torch.float64) + 0.05
    return (p / p.sum(dim=-1, keepdim=True)).to(ts.dtype).to(ts.device)


def run(d0, d1, s, n=60000, leaves=784, iterations=10, steps=10):
    ts = TorchSettings(device="cuda", dtype="float32")
    gen = torch.Generator().manual_seed(7)
    q = Node(rand_cpt(gen, (10,), ts), name="Q")
    f1 = Node(rand_cpt(gen, (10, d0), ts), name="F1")
    f2 = Node(rand_cpt(gen, (10, d1), ts), name="F2")
    builder = BayesianNetworkBuilder()
    builder.add_node(q, parents=[])
    builder.add_node(f1, parents=[q])
    builder.add_node(f2, parents=[q])
    ys = []
    for i in range(leaves):
        y = Node(rand_cpt(gen, (d0, d1, s), ts), name=f"Y{i}")
        builder.add_node(y, parents=[f1, f2])
        ys.append(y)
    network = builder.build()
    states = torch.randint(0, s, (leaves, n), generator=gen)
    evidence = [torch.nn.functional.one_hot(states[i], s).to(ts.dtype).cuda() * (1 - 1e-6) + 5e-7 for i in range(leaves)]
    machine = HipSpaInferenceMachine(
        SpaInferenceMachineSettings(ts, False, num_iterations=iterations, allow_loops=True), network, ys, n)
    times = []
    for step in range(steps + 2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        machine.enter_evidence(Evidence(evidence, ts))
        machine.em_statistics()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    enable_profiling(True)
    profile_report()
    machine.enter_evidence(Evidence(evidence, ts))
    machine.em_statistics()
    report = profile_report()
    enable_profiling(False)
    fused = any(r["kernel"].startswith(f"leaf_family[{d0}x{d1}x{s}") for r in report)
    ms = sorted(times[2:])[len(times[2:]) // 2] * 1e3
    return {"shape": [d0, d1, s], "fused": fused, "ms_per_pass": ms, "samples_per_s": n / ms * 1e3}


if __name__ == "__main__":
    shapes = [tuple(int(x) for x in arg.split(",")) for arg in sys.argv[1:]] or [(2, 2, 2), (3, 3, 3), (3, 3, 4)]
    for shape in shapes:
        print(json.dumps(run(*shape)), flush=True)
