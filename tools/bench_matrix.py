"""Secondary measurements on one GPU (the headline line is bench.py's):

  * EM step throughput of the 784-leaf model at the batch sizes the reference's experiments use
    (100, 1 000, 7 500 = 60 000 / 8, 60 000) and sizes in between, on float evidence and on one state
    byte per leaf and observation;
  * inference-only throughput (enter_evidence + T iterations + log-likelihood, no E/M-step) on
    the 784-leaf model, the two docs networks and the loops network at 2^20 observations.

Networks other than the 784-leaf ones are rebuilt from the golden fixtures (CPTs as the reference's
docs / experiments define them); their evidence rows are the fixture's rows resampled to N.
Each figure is the median of 5 timed repetitions after 2 warm-ups.  Prints one JSON object.

    python tools/bench_matrix.py > gpurun_out/bench_matrix.json
"""
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import bench  # noqa: E402
from conftest import Fixture  # noqa: E402
from bayesiannetwork_amd.inference_machines.evidence import Evidence  # noqa: E402
from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings  # noqa: E402
from bayesiannetwork_amd.optimizers import EmBatchOptimizer, EmBatchOptimizerSettings, OptimizerLogger  # noqa: E402
from bayesiannetwork_amd.torch_settings import TorchSettings  # noqa: E402

ts = TorchSettings(device="cuda", dtype="float32")
device = torch.device("cuda", 0)


def median_time(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    return statistics.median(times)


def em_throughput(n, steps, states=False):
    from bayesiannetwork_amd.inference_machines.evidence import HardEvidence

    network, ys = bench.build_model(ts, "c3a")
    batches = [bench.synthetic_evidence(ts, n, len(ys), 1 + b, device) for b in range(2)]
    if states:  # the same evidence as one state byte per leaf and observation, resident on the device
        batches = [HardEvidence(torch.stack([d.argmax(dim=1) for d in b.data]), [2] * len(ys), ts,
                                on_value=1.0 - 1e-6, off_value=1e-6) for b in batches]

    def factory(net):
        return HipSpaInferenceMachine(SpaInferenceMachineSettings(ts, True), net, ys, n)

    def run():
        opt = EmBatchOptimizer(network, factory,
                               EmBatchOptimizerSettings(learning_rate=0.1, num_epochs=1, regularization=0.01,
                                                        deferred_logging=True), OptimizerLogger())
        opt.optimize(bench._Batches([batches[i % 2] for i in range(steps)], n * steps))

    t = median_time(run)
    return dict(workload="C3a EM step" + (" on resident state bytes" if states else ""), batch=n,
                steps_per_repetition=steps, ms_per_step=t / steps * 1e3, samples_per_s=n * steps / t)


def inference_throughput(label, network, observed, evidence, n, iterations=None, loops=False):
    machine = HipSpaInferenceMachine(
        SpaInferenceMachineSettings(ts, True, num_iterations=iterations, allow_loops=loops), network, observed, n)

    def run():
        machine.reset_messages()
        machine.enter_evidence(evidence)
        machine.log_likelihood_tensor()

    t = median_time(run)
    return dict(workload=label + " inference (enter_evidence + T iterations + log-likelihood)", batch=n,
                iterations=machine.num_iterations, ms=t * 1e3, samples_per_s=n / t,
                device_bytes=machine.device_bytes)


def resampled(fixture_name, n, seed=1, iterations=None, loops=False):
    fx = Fixture(fixture_name)
    network, observed, _ = fx.build(ts)
    rows = fx.evidence_tensors()
    gen = torch.Generator().manual_seed(seed)
    pick = torch.randint(0, rows[0].shape[0], (n,), generator=gen)
    evidence = Evidence([r[pick].to(device).contiguous() for r in rows], ts)
    return inference_throughput(fixture_name, network, observed, evidence, n,
                                iterations if iterations is not None else fx.meta["num_iterations"], loops)


def main():
    out = dict(device=torch.cuda.get_device_name(0), dtype="float32", results=[])
    for n, steps in ((100, 200), (1000, 200), (2048, 200), (3072, 200), (4096, 100), (7500, 100), (16384, 50),
                     (30000, 50), (60000, 50)):
        out["results"].append(em_throughput(n, steps))
    for n, steps in ((4096, 100), (7500, 100), (16384, 50), (60000, 50)):
        out["results"].append(em_throughput(n, steps, states=True))
    network, ys = bench.build_model(ts, "c3a")
    out["results"].append(inference_throughput("C3a 784-leaf", network, ys,
                                               bench.synthetic_evidence(ts, 60000, len(ys), 1, device), 60000))
    n_big = 1 << 20
    out["results"].append(resampled("c1_docs1_f32", n_big))
    out["results"].append(resampled("c2_docs2_f32", n_big))
    out["results"].append(resampled("c4_loops_f32", n_big, loops=True))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
