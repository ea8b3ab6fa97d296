"""Replication check at a large batch: evidence rows repeated R times must give the same mean
log-likelihood and the same mean expected counts as the base batch (any index overflow or
tile-boundary mistake at large N breaks this).  python tools/check_large_n.py [base] [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from bayesiannetwork_amd.inference_machines.evidence import Evidence  # noqa: E402
from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings  # noqa: E402
from bayesiannetwork_amd.torch_settings import TorchSettings  # noqa: E402

base = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 256
ts = TorchSettings(device="cuda", dtype="float32")
network, ys = bench.build_model(ts, "c3a")
device = torch.device("cuda", 0)
small_ev = bench.synthetic_evidence(ts, base, len(ys), 1, device)
big_ev = Evidence([t.repeat(reps, 1) for t in small_ev.data], ts)
settings = SpaInferenceMachineSettings(ts, True)
small = HipSpaInferenceMachine(settings, network, ys, base)
ll_s, c_s = small.em_statistics(small_ev)
big = HipSpaInferenceMachine(settings, network, ys, base * reps)
ll_b, c_b = big.em_statistics(big_ev)
torch.cuda.synchronize()
print("N", base * reps, "ll small", ll_s.item(), "ll big", ll_b.item())
print("plan memory after the statistics call: %.2f GB held, %.2f GB of message arrays not backed yet"
      % (big.device_bytes / 1e9, big.unmapped_bytes / 1e9))
held_for_statistics = big.device_bytes
rel = ((c_b / reps - c_s).abs() / (c_s.abs() + 1e-9)).max().item()
print("max relative difference of mean counts", rel)
del big_ev
two = HipSpaInferenceMachine(settings, network, ys, base * reps)
big_ev = Evidence([t.repeat(reps, 1) for t in small_ev.data], ts)
two.enter_evidence(big_ev)
ll_t, c_t = two.em_statistics()
print("two-call route ll", ll_t.item(), "counts equal", torch.equal(c_t, c_b))
q = list(network.nodes)[0]
a = big.infer_single_nodes([q])[0]
b = small.infer_single_nodes([q])[0]
print("Q marginals replicate bit-exactly:", torch.equal(a.view(reps, base, -1), b.expand(reps, -1, -1)))
assert abs(ll_s.item() - ll_b.item()) < 1e-6 * abs(ll_s.item()) and rel < 1e-4
print("plan memory after the query: %.2f GB" % (big.device_bytes / 1e9))
assert held_for_statistics < 16e9 or base * reps < (1 << 20), "an EM-only plan at 2^20 observations must stay under 16 GB"
