"""Deviation of the root's log-normaliser c from the float64 oracle, float32 plan, wide trees of several class counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import test_hip_generated_networks as G
from oracle.spa_oracle import SpaOracle
from bayesiannetwork_amd.inference_machines.evidence import Evidence
from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings

for name in sys.argv[1:]:
    for n in (300, 2000):
        spec, observed_names, iterations, loops, cpts, evidence, make = G._build(name, "float32", n)
        ts, net, by = make("cuda", "float32")
        ev = [evidence[o].float() for o in observed_names]
        ts64, net64, by64 = make("cpu", "float64")
        for node, _, _ in spec:
            by64[node].cpt = by[node].cpt.cpu().double()
        o64 = SpaOracle(net64, [by64[o] for o in observed_names], n, dtype=torch.float64, num_iterations=iterations, allow_loops=loops)
        o64.enter_evidence([e.double() for e in ev])
        o64.log_likelihood(False)
        m = HipSpaInferenceMachine(SpaInferenceMachineSettings(ts, False), net, [by[o] for o in observed_names], n)
        m.enter_evidence(Evidence([e.cuda() for e in ev], ts))
        q = list(net.nodes)[0]
        g, k = o64.v_of[0]
        want = o64.vgroups[g]["c"][k].numpy()
        got = m.local_log_likelihoods(q).cpu().double().numpy()
        err = np.abs(got - want)
        marg = np.abs(m.infer_single_nodes([q])[0].cpu().double().numpy() - o64.infer_single(0).numpy()).max()
        print(name, n, "c Q max abs err %.3g  mean %.3g  marginal max abs err %.3g" % (err.max(), err.mean(), marg), flush=True)
