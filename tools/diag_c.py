"""Where do the float32 normalisers of a fixture differ most from the float64 truth?"""
import sys
import numpy as np
import torch

sys.path.insert(0, "tests")
from conftest import Fixture
from test_hip_parity import _machine
from bayesiannetwork_amd.inference_machines.evidence import Evidence

fx = Fixture(sys.argv[1] if len(sys.argv) > 1 else "c3a_784_f32")
ts, network, observed, by_name, machine = _machine(fx)
machine.enter_evidence(Evidence(fx.evidence_tensors(), ts))
nodes = list(network.nodes)
got = torch.stack([machine.local_log_likelihoods(n) for n in nodes]).cpu().numpy().astype(np.float64)
truth = fx["truth/c"].astype(np.float64)
ref32 = fx["out/c"].astype(np.float64)
bound = 1e-5 * np.abs(truth) + 2e-6
for name, arr in (("ours", got), ("ref32", ref32)):
    e = np.abs(arr - truth) / bound
    k, n = np.unravel_index(np.argmax(e), e.shape)
    print(name, "worst", e.max(), "node", nodes[k].name, "n", n, "truth", truth[k, n], "value", arr[k, n], "abs err", abs(arr[k, n] - truth[k, n]))
    print("   node0 worst", e[0].max(), "abs", np.abs(arr[0] - truth[0]).max(), "leaf worst", e[1:].max(), "abs", np.abs(arr[1:] - truth[1:]).max())
    print("   mean signed err node0", (arr[0] - truth[0]).mean(), "leaves", (arr[1:] - truth[1:]).mean())
