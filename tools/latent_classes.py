"""EM step of Q(classes) -> 784 x Y(2 | Q) at 60 000 resident observations for several numbers of
latent classes.  The wide-parent / leaf-family kernels serve the class counts they are instantiated
for (the row says so: "fused"), the unfused kernels any other:
    python tools/latent_classes.py 5 8 10 12 [--n 100]
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
from bayesiannetwork_amd.optimizers.em_step import FlatModel
from bayesiannetwork_amd.torch_settings import TorchSettings


def rand_cpt(gen, shape, ts):
    p = torch.rand(shape, generator=gen, dtype=torch.float64) + 0.05
    return (p / p.sum(dim=-1, keepdim=True)).to(ts.dtype).to(ts.device)


def run(classes, n=60000, leaves=784, steps=20):
    ts = TorchSettings(device="cuda", dtype="float32")
    gen = torch.Generator().manual_seed(7)
    q = Node(rand_cpt(gen, (classes,), ts), name="Q")
    builder = BayesianNetworkBuilder()
    builder.add_node(q, parents=[])
    ys = []
    for i in range(leaves):
        y = Node(rand_cpt(gen, (classes, 2), ts), name=f"Y{i}")
        builder.add_node(y, parents=[q])
        ys.append(y)
    network = builder.build()
    model = FlatModel.of(network)
    model.assign(model.gather())
    states = torch.randint(0, 2, (leaves, n), generator=gen)
    whole = (torch.nn.functional.one_hot(states, 2).to(ts.dtype) * (1 - 1e-6) + 5e-7).cuda()
    evidence = Evidence(list(whole.unbind(0)), ts)
    machine = HipSpaInferenceMachine(SpaInferenceMachineSettings(ts, True), network, ys, n)
    for _ in range(3):
        machine.em_step(evidence, 0.1, 0.01)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        machine.em_step(evidence, 0.1, 0.01)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    enable_profiling(True)
    profile_report()
    machine.em_step(evidence, 0.1, 0.01)
    report = profile_report()
    enable_profiling(False)
    fused = any(r["kernel"].startswith(f"leaf_family[{classes}x2") for r in report)
    return {"classes": classes, "fused": fused, "ms_per_step": ms, "samples_per_s": n / ms * 1e3,
            "launches": sum(r["launches"] for r in report),
            "kernels": sorted(({"kernel": r["kernel"], "launches": r["launches"], "ms": round(r["total_ms"], 4)}
                               for r in report), key=lambda r: -r["ms"])[:6]}


if __name__ == "__main__":
    args = sys.argv[1:]
    n = 60000
    if "--n" in args:  # observations per step (small ones show the one-launch step of star networks)
        at = args.index("--n")
        n = int(args[at + 1])
        del args[at:at + 2]
    for arg in args or ["5", "8", "10", "12"]:
        row = run(int(arg), n=n, steps=200 if n <= 2000 else 20)
        row["batch"] = n
        print(json.dumps(row), flush=True)
