"""Error statistics of the loopy 784-leaf model in float32: HIP machine and float32 oracle against the float64 oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import test_hip_headline_models as H
from bayesiannetwork_amd.inference_machines.evidence import Evidence
from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings

n = 1025
for iters in (2, 4, 6, 10):
    ts, network, ys = H._leaf_model("c3b", "float32", "cuda")
    ev = H._mixed_evidence(n, len(ys), 5, ts.dtype)
    o64 = H._oracle_for("c3b", network, ys, "float32", n, iters, True, ev)
    o32 = H._oracle_for("c3b", network, ys, "float32", n, iters, True, ev, torch.float32)
    m = HipSpaInferenceMachine(SpaInferenceMachineSettings(ts, False, num_iterations=iters, allow_loops=True), network, ys, n)
    m.enter_evidence(Evidence([e.cuda() for e in ev], ts))
    nodes = list(network.nodes)
    for j in (0, 1, 2, 5):
        want = o64.infer_single(j).numpy()
        a = np.abs(m.infer_single_nodes([nodes[j]])[0].cpu().double().numpy() - want)
        b = np.abs(o32.infer_single(j).double().numpy() - want)
        print(f"T={iters} {nodes[j].name}: ours max {a.max():.3g} median {np.median(a):.3g} p99 {np.quantile(a, .99):.3g} | oracle32 max {b.max():.3g} median {np.median(b):.3g} p99 {np.quantile(b, .99):.3g}")
