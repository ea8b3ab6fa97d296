"""Two ranks on one device over gloo, the sharded EM step of bench.py, with a synchronisation and a
progress line before the collective: tells a kernel that does not finish from a collective that
does not.   python3 tools/two_rank_probe.py [batch]   (starts its two ranks itself)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if "RANK" not in os.environ:
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                              env=dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                                       MASTER_PORT=str(port))) for r in range(2)]
    sys.exit(max(abs(p.wait()) for p in procs))

import faulthandler

faulthandler.dump_traceback_later(50, exit=True)
import torch

import bench
from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings
from bayesiannetwork_amd.optimizers import EmBatchOptimizer, EmBatchOptimizerSettings, OptimizerLogger
from bayesiannetwork_amd.optimizers import em_batch_optimizer, em_step
from bayesiannetwork_amd.torch_settings import TorchSettings

rank = int(os.environ["RANK"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
torch.cuda.set_device(0)
torch.distributed.init_process_group("gloo")
print(f"rank {rank}: group up", file=sys.stderr, flush=True)
t = torch.ones(10, dtype=torch.float64, device="cuda")
torch.distributed.all_reduce(t)
print(f"rank {rank}: first all_reduce done {t[0].item()}", file=sys.stderr, flush=True)
ts = TorchSettings(device="cuda", dtype="float32")
network, ys = bench.build_model(ts, "c3a")
batches = [bench.synthetic_evidence(ts, n, len(ys), 1 + 17 * rank + b, torch.device("cuda", 0)) for b in range(2)]
original = em_step.reduce_packed


def traced(machine, packed, sharded):
    t0 = time.perf_counter()
    if not os.environ.get("PROBE_NO_SYNC"):
        torch.cuda.synchronize()
    print(f"rank {rank}: statistics done after {time.perf_counter() - t0:.3f} s, entering all_reduce", file=sys.stderr, flush=True)
    out = original(machine, packed, sharded)
    torch.cuda.synchronize()
    print(f"rank {rank}: all_reduce done", file=sys.stderr, flush=True)
    return out


em_batch_optimizer.reduce_packed = traced
opt = EmBatchOptimizer(network, lambda net: HipSpaInferenceMachine(SpaInferenceMachineSettings(ts, True), net, ys, n),
                       EmBatchOptimizerSettings(learning_rate=0.1, num_epochs=1, regularization=0.01, sharded=True,
                                                deferred_logging=True), OptimizerLogger())
opt.optimize(bench._Batches([batches[i % 2] for i in range(steps)], n * steps))
torch.cuda.synchronize()
print(f"rank {rank}: done", file=sys.stderr, flush=True)
torch.distributed.destroy_process_group()
