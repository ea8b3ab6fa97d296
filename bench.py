#!/usr/bin/env python3
"""Headline benchmark: EM steps per second on the MNIST-style 784-leaf latent model.

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong]

Workload (the latent-class model of the reference's dev_experiments/mnist, single-hidden-node
form): Q(10) -> 784 x Y(2), random CPTs (seed 0), synthetic
evidence x ~ Bernoulli(0.2) soft-clamped to [5e-7, 1-5e-7] (seed 1 + rank), float32.
One step = one EM step of the mini-batch optimizer on one batch per GPU: fresh machine
(pooled plan), enter_evidence, T=4 flooding iterations, log-likelihood, expected counts,
[all-reduce of the counts over ranks], M-step (lr 0.1, regularisation 0.01).  Evidence is
resident in HBM before the timed region.

N > 1: one process per GPU.  Started without a launcher (no RANK in the environment) the script
starts its N ranks itself, as fresh child processes, before it has touched a GPU, and passes rank
0's line on; under torch.distributed.run it takes RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from
the environment.  BN_BENCH_SHARE_DEVICE=1 puts every rank on device 0 and lets them meet over gloo
(rehearsal of the sharded path on a one-GPU box).  Two modes, both measured in an N > 1 run:
weak scaling (every rank its own batch of --batch observations, 60 000 by default; the line's
`value`) and strong scaling (the 60 000 images of the configuration split over the ranks,
60 000 / N each; `strong`), together with the all-reduce's payload and duration.
--scaling strong makes the strong-scaling figure the line's `value`.

Rank 0 prints one JSON line: metric / value / unit, the step time, the workload in `config`,
`roofline` for the dominant kernel (HBM bytes moved per launch, from the committed counter pass of
this command, over the launch time measured live, against the HBM peak; beside it what else could
bind: the share of SIMD cycles issuing vector instructions, and the launch's counted arithmetic
against the fp32 peak), per-kernel durations in `kernels`, and `cpu_baseline` (the CPU oracle on a
bounded sample of the same workload).
"""

from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
FP32_PEAK_TFLOPS = 157.3  # vector and matrix fp32 alike


# profile label -> kernel-name stems it covers (a label can stand for several kernels: the loop launches of
# the leaf families run on leaf_loop_p1, their full / reduce launches on leaf_family_p1)
_KERNEL_STEMS = {"leaf_family": ("bn::leaf_family_p", "bn::leaf_loop_p", "bn::hard_loop_", "bn::hard_reduce_"),
                 "tree_iteration": ("bn::factor_tree_mfma",), "tree_iteration_fresh": ("bn::factor_tree_mfma",),
                 "tree_iteration_exact": ("bn::factor_tree_mfma",), "em_finish": ("bn::em_finish",),
                 "factor_pass": ("bn::factor_p",),
                 "em_counts": ("bn::em_p",), "variable_pass": ("bn::variable_small_fast",),
                 "wide_combine": ("bn::wide_combine_fast", "bn::wide_combine_states"),
                 "evidence_scatter": ("bn::evidence_scatter",), "log_likelihood": ("bn::ll_partial",)}


def _latest_profile(suffix: str):
    """profiles/r<N>_<suffix> of the highest round N that has one."""
    import glob
    import re

    best = None
    for path in glob.glob(os.path.join(ROOT, "profiles", f"r*_{suffix}")):
        m = re.match(r"r(\d+)_", os.path.basename(path))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), path)
    return best[1] if best else None


def pmc_counters(kernel_label: str, model: str, signature: str):
    """What the committed rocprofv3 --pmc summaries of this same command say about the kernel behind a
    profile label: ``(HBM bytes per launch, VALU-busy share, provenance)``.  The summaries
    (tools/pmc_traffic.sh: FETCH_SIZE / WRITE_SIZE passes; tools/pmc_valu.sh: SQ_ACTIVE_INST_VALU
    against GRBM_GUI_ACTIVE) carry, under "_meta", the commit they were taken at and the
    register / LDS / spill signature of every profile label's kernels at that time; a figure is
    handed out only while the kernels running now still have that signature - otherwise the code
    has changed since the pass and the figure is None.  A label can cover several instantiations
    (loop / reduce modes): the launch-weighted mean."""
    stems = _KERNEL_STEMS.get(kernel_label.split("[")[0])
    tag = "" if model == "c3a" else f"_{model}"  # (model "c3a_u8": the headline model on resident state bytes)
    out, provenance = [], {}
    for suffix, key in ((f"pmc_traffic{tag}.json", "hbm_bytes_per_launch"), (f"pmc_valu{tag}.json", "valu_busy")):
        path = _latest_profile(suffix)
        value = None
        if path and stems:
            data = json.load(open(path))
            meta = data.get("_meta", {})
            recorded = meta.get("signatures", {}).get(kernel_label)
            match = recorded is not None and recorded == signature
            provenance[key] = dict(file=os.path.relpath(path, ROOT), git=meta.get("git"), signature_match=match)
            hits = [v for k, v in data.items() if k != "_meta" and k.startswith(stems) and v.get(key) is not None]
            if match and hits:
                value = sum(h[key] * h["launches"] for h in hits) / sum(h["launches"] for h in hits)
        out.append(value)
    return out[0], out[1], provenance


def build_model(ts, model: str, num_pixels: int = 784, classes: int = 10):
    from bayesiannetwork_amd.bayesian_network import BayesianNetworkBuilder, Node

    gen = torch.Generator().manual_seed(0)

    def rand_cpt(shape):
        p = torch.rand(shape, generator=gen, dtype=torch.float64)
        p = (p / p.sum(dim=-1, keepdim=True)).to(ts.dtype)
        return (p / p.sum(dim=-1, keepdim=True)).to(ts.device)

    q = Node(torch.full((classes,), 1.0 / classes, dtype=ts.dtype, device=ts.device), name="Q")
    builder = BayesianNetworkBuilder().add_node(q)
    ys = []
    if model == "c3b":
        f1 = Node(rand_cpt((10, 2)), name="F1")
        f2 = Node(rand_cpt((10, 2)), name="F2")
        builder.add_node(f1, parents=q).add_node(f2, parents=q)
        for i in range(num_pixels):
            y = Node(rand_cpt((2, 2, 2)), name=f"Y{i}")
            builder.add_node(y, parents=[f1, f2])
            ys.append(y)
    else:
        for i in range(num_pixels):
            y = Node(rand_cpt((classes, 2)), name=f"Y{i}")
            builder.add_node(y, parents=q)
            ys.append(y)
    return builder.build(), ys


def build_deep_tree(ts, depth: int = 6, states: int = 64):
    """Deep-tree workload: balanced binary tree of 64-state hidden nodes (2^depth - 1 of
    them) with one observed 64-state leaf under every hidden leaf; CPTs [64] / [64, 64]."""
    from bayesiannetwork_amd.bayesian_network import BayesianNetworkBuilder, Node

    gen = torch.Generator().manual_seed(0)

    def rand_cpt(shape):
        p = torch.rand(shape, generator=gen, dtype=torch.float64)
        p = (p / p.sum(dim=-1, keepdim=True)).to(ts.dtype)
        return (p / p.sum(dim=-1, keepdim=True)).to(ts.device)

    root = Node(rand_cpt((states,)), name="H0")
    builder = BayesianNetworkBuilder().add_node(root)
    frontier, count = [root], 1
    for _ in range(depth - 1):
        nxt = []
        for parent in frontier:
            for _ in range(2):
                node = Node(rand_cpt((states, states)), name=f"H{count}")
                count += 1
                builder.add_node(node, parents=parent)
                nxt.append(node)
        frontier = nxt
    leaves = []
    for i, parent in enumerate(frontier):
        leaf = Node(rand_cpt((states, states)), name=f"Y{i}")
        builder.add_node(leaf, parents=parent)
        leaves.append(leaf)
    return builder.build(), leaves


def one_hot_evidence(ts, n, nodes, seed, device):
    from bayesiannetwork_amd.inference_machines.evidence import Evidence

    gen = torch.Generator().manual_seed(seed)
    data = []
    for node in nodes:
        idx = torch.randint(0, node.num_states, (n,), generator=gen)
        e = torch.nn.functional.one_hot(idx, node.num_states).to(ts.dtype) * (1 - 1e-6) + 5e-7
        data.append(e.to(device).contiguous())
    return Evidence(data, ts)


def synthetic_evidence(ts, n, num_pixels, seed, device):
    from bayesiannetwork_amd.inference_machines.evidence import Evidence

    gen = torch.Generator().manual_seed(seed)
    x = (torch.rand((num_pixels, n), generator=gen) < 0.2).to(ts.dtype) * 0.999999 + 5e-7
    x = x.to(device)
    stacked = torch.stack([1 - x, x], dim=2).contiguous()  # [pixels, n, 2]
    return Evidence(list(stacked.unbind(0)), ts)


class _Batches:
    def __init__(self, batches, total):
        self._batches, self.num_observations = batches, total

    def __len__(self):
        return len(self._batches)

    def __iter__(self):
        return iter(self._batches)


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform

    return platform.processor() or "unknown"


def cpu_baseline(model: str, n: int, iters, warmup: int = 2, steps: int = 5):
    """The CPU oracle (torch-on-CPU restatement of the reference's pass structure) doing the same
    EM step on a bounded sample: ``n`` observations (60 000 / 8, the per-GPU batch of the 8-GPU
    job; the oracle's temporaries do not fit 60 000), ``warmup`` untimed steps, then the median of
    ``steps`` timed ones, both with the machine rebuilt every step (what the reference's optimizer
    does, optimizers/em_batch_optimizer.py:41) and with the plan kept.  kind = "port"; `value` is
    the faster of the two.  The real reference timed beside this oracle in the build container
    (oracle/time_reference.py) is in profiles/r2_reference_vs_oracle.json."""
    import statistics

    from bayesiannetwork_amd.torch_settings import TorchSettings
    from oracle.spa_oracle import SpaOracle, m_step_batch

    threads = int(os.environ.get("OMP_NUM_THREADS") or 0) or min(os.cpu_count() or 1, 16)
    torch.set_num_threads(threads)
    ts = TorchSettings(device="cpu", dtype="float32")
    network, ys = build_model(ts, model)
    ev = synthetic_evidence(ts, n, len(ys), 1, "cpu").data
    state = {"cpts": [node.cpt for node in network.nodes]}

    def rebuilt():
        oracle = SpaOracle(network, ys, n, dtype=torch.float32, num_iterations=iters, allow_loops=True,
                           cpts=state["cpts"])
        oracle.enter_evidence(ev)
        oracle.log_likelihood(True)
        state["cpts"] = m_step_batch(state["cpts"], oracle.expected_counts(), 0.1, 0.01)

    kept = SpaOracle(network, ys, n, dtype=torch.float32, num_iterations=iters, allow_loops=True, cpts=state["cpts"])

    def cached():
        kept.reset_messages()
        kept.set_cpts(state["cpts"])
        kept.enter_evidence(ev)
        kept.log_likelihood(True)
        state["cpts"] = m_step_batch(state["cpts"], kept.expected_counts(), 0.1, 0.01)

    def timed(step):
        for _ in range(warmup):
            step()
        times = []
        for _ in range(steps):
            t0 = time.perf_counter()
            step()
            times.append(time.perf_counter() - t0)
        return statistics.median(times)

    t_rebuilt, t_cached = timed(rebuilt), timed(cached)
    out = dict(value=n / min(t_rebuilt, t_cached), unit="samples/s", cores=threads, kind="port",
               sample=f"one EM step on {n} observations, float32, torch CPU oracle; {warmup} warm-up steps, "
                      f"median of {steps}",
               cpu_model=cpu_model(), rebuilt_every_step=n / t_rebuilt, plan_cached=n / t_cached)
    ref = os.path.join(ROOT, "profiles", "r2_reference_vs_oracle.json")
    if os.path.exists(ref):
        r = json.load(open(ref))
        out["reference_fidelity"] = dict(
            note="real reference vs this oracle, build container, same N and threads (oracle/time_reference.py)",
            cpu_model=r["cpu_model"], threads=r["threads"], reference_samples_per_s=r["reference_samples_per_s"],
            oracle_over_reference=r["oracle_over_reference"])
    return out


def roofline_block(records, model, default_workload):
    """The `roofline` object for the dominant kernel of ``records`` (bn_profile_report entries of a
    timed region): HBM bytes per launch from the committed counter pass over the live launch time."""
    if not records:
        return None
    dom = max(records, key=lambda k: k["total_ms"])
    avg_s = dom["total_ms"] / dom["launches"] / 1e3
    traffic, valu_busy, provenance = pmc_counters(dom["kernel"], model, dom.get("signature", ""))
    if not default_workload:  # the counter passes ran the default command of each model
        traffic = valu_busy = None
    achieved = traffic / avg_s / 1e9 if traffic is not None else None
    flops = dom.get("alg_flops") or None
    useful = flops / avg_s / 1e12 / FP32_PEAK_TFLOPS if flops else None
    fracs = {"hbm": achieved / HBM_PEAK_GBS if achieved is not None else None, "valu": valu_busy}
    live = {k: v for k, v in fracs.items() if v is not None}
    return dict(
        bound="hbm", kernel=dom["kernel"], achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
        frac=fracs["hbm"], traffic=traffic, avg_launch_ms=avg_s * 1e3,
        # what the launch is closest to being bound by, of the figures at hand
        binds=max(live, key=live.get) if live else None,
        valu_busy=valu_busy, useful_flop_frac=useful,
        useful_flops_per_launch=flops, fp32_peak_tflops=FP32_PEAK_TFLOPS,
        counters=provenance, signature=dom.get("signature"),
        # message-level bytes (every message read and written once); the fused kernels keep an iteration's
        # messages in registers, so this is not a bound on them
        alg_equivalent=dict(bytes_per_launch=dom["alg_bytes"], GBps=dom["alg_bytes"] / avg_s / 1e9,
                            x_hbm_peak=dom["alg_bytes"] / avg_s / 1e9 / HBM_PEAK_GBS))


def secondary_em_step(ts, device, model, n, steps, warmup):
    """EM-step throughput of another workload on this GPU (resident evidence, same optimizer
    settings as the headline run): one dict for the ``secondary`` list, with its own `roofline`."""
    from bayesiannetwork_amd.inference_machines.hip_spa import (HipSpaInferenceMachine, SpaInferenceMachineSettings,
                                                                enable_profiling, profile_report)
    from bayesiannetwork_amd.optimizers import EmBatchOptimizer, EmBatchOptimizerSettings, OptimizerLogger

    iters = 10 if model == "c3b" else None
    if model == "c5":
        network, ys = build_deep_tree(ts)
        batches = [one_hot_evidence(ts, n, ys, 1 + b, device) for b in range(2)]
    elif model == "c3a_u8":
        # the headline model on the same evidence given as one state byte per leaf and observation
        from bayesiannetwork_amd.inference_machines.evidence import HardEvidence
        network, ys = build_model(ts, "c3a")
        batches = [HardEvidence(torch.stack([d.argmax(dim=1) for d in b.data]).to(torch.uint8), [2] * len(ys), ts,
                                on_value=float(b.data[0].max()), off_value=float(b.data[0].min()))
                   for b in (synthetic_evidence(ts, n, len(ys), 1 + i, device) for i in range(2))]
    else:
        network, ys = build_model(ts, model)
        batches = [synthetic_evidence(ts, n, len(ys), 1 + b, device) for b in range(2)]
    settings = SpaInferenceMachineSettings(ts, True, num_iterations=iters, allow_loops=model == "c3b")

    def factory(net):
        return HipSpaInferenceMachine(settings, net, ys, n)

    def run(count):
        opt = EmBatchOptimizer(network, factory,
                               EmBatchOptimizerSettings(learning_rate=0.1, num_epochs=1, regularization=0.01,
                                                        deferred_logging=True), OptimizerLogger())
        opt.optimize(_Batches([batches[i % 2] for i in range(count)], n * count))

    run(warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # kernel durations from one more, untimed step
    enable_profiling(True)
    profile_report()
    run(1)
    torch.cuda.synchronize()
    records = profile_report()
    enable_profiling(False)
    t_used = factory(network).num_iterations
    workload = {"c3b": "Q(10)->F1,F2(2)->784xY(2|F1,F2) loopy model",
                "c3a_u8": "Q(10)->784xY(2) latent-class model on resident uint8 states (HardEvidence)",
                "c5": "depth-6 binary tree of 64-state hidden nodes + 32 observed 64-state leaves"}[model]
    return dict(workload=f"{workload}, EM step (T={t_used})", batch=n, steps=steps, warmup=warmup,
                value=n * steps / elapsed, unit="samples/s", ms_per_step=elapsed / steps * 1e3,
                roofline=roofline_block(records, model, True))


def launch_ranks(n_ranks: int) -> int:
    """Starts the N ranks of an N-GPU run as child processes of their own and passes rank 0's output
    on.  Called before this process has made any GPU call (the children initialise the GPU, the
    parent never does)."""
    import socket
    import subprocess

    share = bool(os.environ.get("BN_BENCH_SHARE_DEVICE"))
    devices = torch.cuda.device_count()  # does not initialise the GPU
    if not share and devices < n_ranks:
        print(f"bench.py: --gpus {n_ranks} but {devices} GPU(s) visible (BN_BENCH_SHARE_DEVICE=1 puts all ranks on "
              "device 0 over gloo)", file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for rank in range(n_ranks):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(0 if share else rank), WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        if share:
            env["BN_BENCH_BACKEND"] = "gloo"
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    # the contract is one JSON line on stdout; anything else a library printed there goes to stderr
    for text in out.decode().splitlines():
        print(text, file=sys.stdout if text.startswith("{") else sys.stderr)
    sys.stdout.flush()
    return max(abs(c) for c in codes)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="observations per GPU per step (weak scaling)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="which figure of an N > 1 run is the line's `value`: weak = --batch observations per GPU, "
                         "strong = 60 000 observations split over the GPUs (both are measured and printed)")
    ap.add_argument("--model", choices=["c3a", "c3b", "c5"], default="c3a")
    ap.add_argument("--dtype", choices=["float32", "float64"], default="float32")
    ap.add_argument("--evidence", choices=["resident", "resident-uint8", "host-float", "host-uint8"],
                    default="resident",
                    help="resident: batches already in HBM as the reference's float tensors (the contract's figure).  "
                         "resident-uint8: in HBM as uint8 state indices (HardEvidence).  host-*: batches start in "
                         "host memory every step and cross PCIe through PrefetchingLoader (PCIe-inclusive rate)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="leave out the loopy and deep-tree EM steps")
    ap.add_argument("--no-kernel-timing", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    if os.environ.get("BN_BENCH_WATCHDOG"):
        # where every thread is, should the run still be going after so many seconds
        import faulthandler

        faulthandler.dump_traceback_later(float(os.environ["BN_BENCH_WATCHDOG"]), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    backend = os.environ.get("BN_BENCH_BACKEND", "nccl")
    if world > 1 or "RANK" in os.environ:
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            torch.distributed.init_process_group(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from bayesiannetwork_amd.inference_machines.hip_spa import (
        HipSpaInferenceMachine,
        SpaInferenceMachineSettings,
        enable_profiling,
        profile_report,
    )
    from bayesiannetwork_amd.optimizers import EmBatchOptimizer, EmBatchOptimizerSettings, OptimizerLogger
    from bayesiannetwork_amd.torch_settings import TorchSettings

    ts = TorchSettings(device="cuda", dtype=args.dtype)
    device = torch.device("cuda", local_rank)
    default_size = args.batch is None
    n_weak = args.batch or (131072 if args.model == "c5" else 60000)
    iters = 10 if args.model == "c3b" else None
    sharded = world > 1 or bool(os.environ.get("BN_BENCH_FORCE_SHARDED"))
    if args.model == "c5":
        network, ys = build_deep_tree(ts)
    else:
        network, ys = build_model(ts, args.model)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    def measure(n, with_kernel_timing):
        """W warm-up steps, then exactly K timed ones on batches of n observations per rank; the
        slowest rank's time."""
        # two distinct resident batches, alternated, so no step re-reads the evidence it just used
        if args.model == "c5":
            batches = [one_hot_evidence(ts, n, ys, 1 + 17 * rank + b, device) for b in range(2)]
        else:
            batches = [synthetic_evidence(ts, n, len(ys), 1 + 17 * rank + b, device) for b in range(2)]

        def factory(net):
            return HipSpaInferenceMachine(
                SpaInferenceMachineSettings(ts, True, num_iterations=iters, allow_loops=args.model == "c3b"),
                net, ys, n)

        logger = OptimizerLogger()
        host_batches = None
        if args.evidence == "resident-uint8":
            from bayesiannetwork_amd.inference_machines.evidence import HardEvidence
            batches = [HardEvidence(torch.stack([d.argmax(dim=1) for d in b.data]).to(torch.uint8),
                                    [d.shape[1] for d in b.data], ts, on_value=float(b.data[0].max()),
                                    off_value=float(b.data[0].min())) for b in batches]
        elif args.evidence != "resident":
            from bayesiannetwork_amd.inference_machines.evidence import Evidence, HardEvidence, PrefetchingLoader
            host_ts = TorchSettings(device="cpu", dtype=args.dtype)
            host_batches = []
            for b in batches:
                if args.evidence == "host-uint8":
                    # the synthetic evidence is hard: state = argmax, on / off = the two values it takes
                    states = torch.stack([d.argmax(dim=1) for d in b.data]).cpu()
                    hard = HardEvidence(states, [d.shape[1] for d in b.data], host_ts,
                                        on_value=float(b.data[0].max()), off_value=float(b.data[0].min()))
                    hard.states = hard.states.pin_memory()
                    host_batches.append(hard)
                else:
                    # one pinned buffer [observed nodes, n, states], the tensors are its slices
                    host_batches.append(Evidence(list(torch.stack(b.data).cpu().pin_memory().unbind(0)), host_ts))

        def run(steps):
            opt = EmBatchOptimizer(
                network, factory,
                EmBatchOptimizerSettings(learning_rate=0.1, num_epochs=1, regularization=0.01, sharded=sharded,
                                         deferred_logging=True), logger)
            if host_batches is not None:
                opt.optimize(PrefetchingLoader(_Batches([host_batches[i % 2] for i in range(steps)], n * steps), device))
            else:
                opt.optimize(_Batches([batches[i % 2] for i in range(steps)], n * steps))

        run(args.warmup)
        kernels = []
        if with_kernel_timing:
            # every kernel's durations from a short untimed pass; inside the timed region only the
            # dominant kernel carries events (a timed dispatch costs a few microseconds extra)
            enable_profiling(True)
            profile_report()
            run(min(args.steps, 5))
            torch.cuda.synchronize()
            kernels = profile_report()
            top = max(kernels, key=lambda k: k["total_ms"])
            # a timed dispatch is fenced off from its neighbours (a few microseconds each): inside the
            # timed region every stride-th launch of the dominant kernel is timed, with a stride coprime
            # to its launches per step so that all of its modes are sampled evenly
            per_step = max(1, round(top["launches"] / min(args.steps, 5)))
            stride = next(s for s in range(5, 64) if math.gcd(s, per_step) == 1)
            enable_profiling(True, only=top["kernel"], stride=stride)
        fence()
        t0 = time.perf_counter()
        run(args.steps)
        fence()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            elapsed = t.item()
        timed = []
        if with_kernel_timing:
            timed = profile_report()
            enable_profiling(False)
        machine = factory(network)
        lls = [log.ll for log in logger.logs]
        return dict(n=n, elapsed=elapsed, kernels=kernels, timed=timed, t_used=machine.num_iterations,
                    cpt_elements=sum(node.cpt.numel() for node in network.nodes),
                    ll_first_last=[lls[-args.steps], lls[-1]] if len(lls) >= args.steps else None)

    def all_reduce_cost(elements):
        """Payload and mean duration of the step's one collective, measured on its own: the statistics
        buffer [expected-count sums, log-likelihood, observation count] in float64."""
        if world == 1:
            return None
        from bayesiannetwork_amd.optimizers.em_step import all_reduce_sum

        buf = torch.zeros(elements, dtype=torch.float64, device=device)
        for _ in range(5):
            all_reduce_sum(buf)
        fence()
        t0 = time.perf_counter()
        reps = 50
        for _ in range(reps):
            all_reduce_sum(buf)
        torch.cuda.synchronize()
        return dict(payload_bytes=elements * 8, mean_us=(time.perf_counter() - t0) / reps * 1e6, backend=backend)

    timing = not args.no_kernel_timing
    weak = measure(n_weak, timing)
    strong = None
    if world > 1:
        total = 60000 if args.model != "c5" else 131072  # the job's observations, split over the ranks
        strong = measure(max(2, (total // world) & ~1), timing and args.scaling == "strong")
    collective = all_reduce_cost(weak["cpt_elements"] + 2)

    if rank == 0:
        primary = strong if (args.scaling == "strong" and strong is not None) else weak
        n = primary["n"]
        # the committed counter passes ran the default command: float32, 60 000 resident observations
        default_workload = (args.dtype == "float32" and args.evidence in ("resident", "resident-uint8") and
                            n == (131072 if args.model == "c5" else 60000))
        counters_of = args.model + ("_u8" if args.evidence == "resident-uint8" else "")
        roofline = roofline_block(primary["timed"], counters_of, default_workload) if primary["timed"] else None
        kernels = primary["kernels"]

        def figure(m):
            return dict(value=world * m["n"] * args.steps / m["elapsed"], unit="samples/s", batch_per_gpu=m["n"],
                        global_batch=m["n"] * world, ms_per_step=m["elapsed"] / args.steps * 1e3)

        out = {
            "metric": "evidence samples/sec (sum-product infer + EM step), MNIST 784-leaf latent model",
            "value": world * n * args.steps / primary["elapsed"],
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": primary["elapsed"] / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if primary is strong else "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.dtype == "float32" else "f64",
            "data": "synthetic",
            "config": {
                "workload": {"c3a": "Q(10)->784xY(2) latent-class model",
                             "c3b": "Q(10)->F1,F2(2)->784xY(2|F1,F2) loopy model",
                             "c5": "depth-6 binary tree of 64-state hidden nodes + 32 observed 64-state leaves"}[args.model]
                + f", EM step (T={primary['t_used']} flooding iterations + ll + expected counts + M-step)",
                "batch_per_gpu": n,
                "evidence": args.evidence,
                "global_batch": n * world,
                "learning_rate": 0.1,
                "regularization": 0.01,
                "parallelism": f"observations sharded over {world} GPU(s), counts all-reduced" if world > 1 else "single GPU",
            },
            "ll_first_last": primary["ll_first_last"],
            "roofline": roofline,
            "kernels": [
                dict(kernel=k["kernel"], launches=k["launches"], avg_ms=k["total_ms"] / k["launches"],
                     alg_GBps=(k["alg_bytes"] / (k["total_ms"] / k["launches"] / 1e3) / 1e9) if k["total_ms"] > 0 else None,
                     signature=k.get("signature"))
                for k in kernels
            ],
        }
        if world > 1:
            # both modes of the N-GPU job, whichever is the line's `value`
            out["weak"] = figure(weak)
            out["strong"] = figure(strong)
            out["all_reduce"] = collective
        if world == 1 and not args.no_cpu_baseline and args.model != "c5":
            # the CPU rate is flat in the batch size (it is bound per observation); the oracle's temporaries
            # do not fit 60 000 observations, so it steps on 7 500 (the per-GPU share of the 8-GPU job)
            out["cpu_baseline"] = cpu_baseline(args.model, 7500, iters)
            out["cpu_baseline"]["sample_batch"] = 7500
            out["cpu_baseline"]["gpu_batch"] = n
            out["cpu_baseline"]["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        if world == 1 and not args.no_secondary and args.model == "c3a" and default_size:
            # the other two EM workloads of the config list, a few steps each, same optimizer settings
            out["secondary"] = [secondary_em_step(ts, device, "c3a_u8", 60000, steps=20, warmup=5),
                                secondary_em_step(ts, device, "c3b", 60000, steps=10, warmup=3),
                                secondary_em_step(ts, device, "c5", 131072, steps=2, warmup=1)]
        print(json.dumps(out))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
