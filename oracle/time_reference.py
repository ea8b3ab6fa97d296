"""Time the REAL reference package beside this repository's CPU oracle on the benchmark's EM step.

Runs only in the build container (it imports ``/root/reference``; see oracle/make_fixtures.py for
the PYTHONPATH recipe).  Workload: the 784-leaf latent-class model Q(10) -> 784 x Y(2), float32,
one mini-batch EM step (lr 0.1, regularisation 0.01) on N observations of Bernoulli(0.2) pixels,
the benchmark's inputs.  Protocol for both: 2 untimed steps, then the median of 5 timed steps, the
same torch thread count.  The reference is run as its optimizer runs it (a new machine every
step, ``optimizers/em_batch_optimizer.py:37-59``); the oracle both ways (machine rebuilt every
step / plan kept).  Writes profiles/r2_reference_vs_oracle.json: the ratio says how much faster
the oracle (bench.py's ``cpu_baseline``, the only CPU baseline that can travel to the GPU box) is
than the code it stands for.

    python3 oracle/time_reference.py [N] [threads]
"""

from __future__ import annotations

import json
import os
import platform
import statistics
import sys
import time

import torch

from bayesian_network.bayesian_network import BayesianNetworkBuilder, Node
from bayesian_network.inference_machines.evidence import Evidence
from bayesian_network.inference_machines.spa_v3.spa_inference_machine import (
    SpaInferenceMachine,
    SpaInferenceMachineSettings,
)
from bayesian_network.optimizers.em_batch_optimizer import EmBatchOptimizer, EmBatchOptimizerSettings
from common.torch_settings import TorchSettings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.spa_oracle import SpaOracle, m_step_batch  # noqa: E402


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 7500
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1)
    torch.set_num_threads(threads)
    ts = TorchSettings(device="cpu", dtype="float32")
    gen = torch.Generator().manual_seed(0)

    def rand_cpt(shape):
        p = torch.rand(shape, generator=gen, dtype=torch.float64)
        p = (p / p.sum(dim=-1, keepdim=True)).to(torch.float32)
        return p / p.sum(dim=-1, keepdim=True)

    q = Node(torch.full((10,), 0.1), name="Q")
    builder = BayesianNetworkBuilder().add_node(q)
    ys = []
    for i in range(784):
        y = Node(rand_cpt((10, 2)), name=f"Y{i}")
        builder.add_node(y, parents=q)
        ys.append(y)
    network = builder.build()
    x = (torch.rand((784, n), generator=torch.Generator().manual_seed(1)) < 0.2).float() * 0.999999 + 5e-7
    data = [torch.stack([1 - x[i], x[i]], dim=1).contiguous() for i in range(784)]
    evidence = Evidence(data, ts)

    class OneBatch:
        num_observations = n

        def __len__(self):
            return 1

        def __iter__(self):
            return iter([evidence])

    settings = SpaInferenceMachineSettings(torch_settings=ts, average_log_likelihood=True)
    optimizer = EmBatchOptimizer(
        network, lambda net: SpaInferenceMachine(settings, net, ys, n),
        EmBatchOptimizerSettings(learning_rate=0.1, num_epochs=1, regularization=0.01))

    def timed(step, warmup=2, steps=5):
        for _ in range(warmup):
            step()
        out = []
        for _ in range(steps):
            t0 = time.perf_counter()
            step()
            out.append(time.perf_counter() - t0)
        return statistics.median(out)

    t_reference = timed(lambda: optimizer.optimize(OneBatch()))

    state = {"cpts": [node.cpt.clone() for node in network.nodes]}

    def oracle_rebuilt():
        oracle = SpaOracle(network, ys, n, dtype=torch.float32, cpts=state["cpts"])
        oracle.enter_evidence(data)
        oracle.log_likelihood(True)
        state["cpts"] = m_step_batch(state["cpts"], oracle.expected_counts(), 0.1, 0.01)

    kept = SpaOracle(network, ys, n, dtype=torch.float32, cpts=state["cpts"])

    def oracle_cached():
        kept.reset_messages()
        kept.set_cpts(state["cpts"])
        kept.enter_evidence(data)
        kept.log_likelihood(True)
        state["cpts"] = m_step_batch(state["cpts"], kept.expected_counts(), 0.1, 0.01)

    t_rebuilt = timed(oracle_rebuilt)
    t_cached = timed(oracle_cached)
    out = dict(
        workload=f"Q(10)->784xY(2), float32, one EmBatchOptimizer step, N={n}, T=4",
        cpu_model=cpu_model(), threads=threads, torch=torch.__version__,
        protocol="2 warm-up steps, median of 5",
        reference_s_per_step=t_reference, reference_samples_per_s=n / t_reference,
        oracle_rebuilt_s_per_step=t_rebuilt, oracle_rebuilt_samples_per_s=n / t_rebuilt,
        oracle_plan_cached_s_per_step=t_cached, oracle_plan_cached_samples_per_s=n / t_cached,
        oracle_over_reference=t_reference / t_rebuilt,
        oracle_plan_cached_over_reference=t_reference / t_cached,
    )
    path = os.path.join(ROOT, "profiles", "r2_reference_vs_oracle.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
