"""Runs one draw of tests/test_hip_random_networks.py and prints where non-finite values first appear."""
import sys

import torch

sys.path.insert(0, ".")
from tests.test_hip_random_networks import _draw  # noqa: E402

from bayesiannetwork_amd.bayesian_network import BayesianNetworkBuilder, Node  # noqa: E402
from bayesiannetwork_amd.inference_machines.evidence import Evidence  # noqa: E402
from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings  # noqa: E402
from bayesiannetwork_amd.torch_settings import TorchSettings  # noqa: E402

seed = int(sys.argv[1])
dtype = sys.argv[2]
spec, observed_idx, evidence, n, loopy = _draw(seed)
ts = TorchSettings(device="cuda", dtype=dtype)
nodes = []
for name, parents, cpt in spec:
    c = cpt.to(ts.dtype)
    nodes.append(Node((c / c.sum(-1, keepdim=True)).to(ts.device), name=name))
builder = BayesianNetworkBuilder()
for node, (_, parents, _) in zip(nodes, spec):
    builder.add_node(node, parents=[nodes[p] for p in parents])
network = builder.build()
print("n", n, "nodes", len(spec), "observed", observed_idx)
for j, (name, parents, cpt) in enumerate(spec):
    print(j, name, parents, tuple(cpt.shape), "obs" if j in observed_idx else "")
for iters in (None, 7):
    try:
        machine = HipSpaInferenceMachine(SpaInferenceMachineSettings(ts, False, num_iterations=iters, allow_loops=True),
                                         network, [nodes[j] for j in observed_idx], n)
        break
    except ValueError as e:
        print("needs iterations:", e)
print("iterations", machine.num_iterations, "unmapped", machine.unmapped_bytes, "held", machine.device_bytes)
ev = Evidence([e.to(ts.dtype).cuda() for e in evidence], ts)
machine.enter_evidence(ev)
print("one call  ll", machine.log_likelihood(), "unmapped", machine.unmapped_bytes)
bad = [node.name for node in network.nodes if not torch.isfinite(machine.local_log_likelihoods(node)).all()]
print("  non-finite c at", bad, "unmapped", machine.unmapped_bytes)
machine.reset_messages()
machine.enter_evidence(ev)
machine._iterate()
print("two calls ll", machine.log_likelihood(), "unmapped", machine.unmapped_bytes)
bad = [node.name for node in network.nodes if not torch.isfinite(machine.local_log_likelihoods(node)).all()]
print("  non-finite c at", bad)

from bayesiannetwork_amd.inference_machines.hip_spa import _native
fresh_machine = HipSpaInferenceMachine(machine.settings, network, [nodes[j] for j in observed_idx], n)
fresh_machine.enter_evidence(ev)
for t in range(3):
    _native.check(fresh_machine._lib.bn_iterate(fresh_machine._plan, 1, fresh_machine._stream()))
    fresh_machine.must_iterate = False
    print("iteration", t + 1)
    for node in network.nodes:
        c = fresh_machine.local_log_likelihoods(node)
        if not torch.isfinite(c).all():
            print("   c", node.name, "bad rows", int((~torch.isfinite(c)).sum()))
        for par in list(network.parents_of(node)) + [node]:
            for d in ("v2f", "f2v"):
                m = fresh_machine.message(par, node, d)
                if not torch.isfinite(m).all():
                    print("   message", d, par.name, "<->factor", node.name, "rows bad", int((~torch.isfinite(m)).any(-1).sum()), m[0].tolist())
