#!/usr/bin/env python3
"""Headline benchmark: EM steps per second on the MNIST-style 784-leaf latent model.

    python bench.py --gpus N --steps K --warmup W

Workload (the latent-class model of the reference's dev_experiments/mnist, single-hidden-node
form): Q(10) -> 784 x Y(2), random CPTs (seed 0), synthetic
evidence x ~ Bernoulli(0.2) soft-clamped to [5e-7, 1-5e-7] (seed 1 + rank), float32.
One step = one EM step of the mini-batch optimizer on one batch per GPU: fresh machine
(pooled plan), enter_evidence, T=4 flooding iterations, log-likelihood, expected counts,
[all-reduce of the counts over ranks], M-step (lr 0.1, regularisation 0.01).  Evidence is
resident in HBM before the timed region.  Weak scaling: every rank gets its own batch of
--batch observations; value = all ranks' observations / slowest rank's time.

Rank 0 prints one JSON line: metric / value / unit, the step time, the workload in `config`,
`roofline` for the dominant kernel (algorithmic bytes per launch / measured launch time against the
HBM peak, plus the profiler-measured HBM bytes per launch in `traffic`), per-kernel durations in
`kernels`, and `cpu_baseline` (the CPU oracle on a bounded sample of the same workload).
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


# profile label -> kernel-name stems it covers (a label can stand for several kernels: the loop launches of
# the leaf families run on leaf_loop_p1, their full / reduce launches on leaf_family_p1)
_KERNEL_STEMS = {"leaf_family": ("bn::leaf_family_p", "bn::leaf_loop_p"), "factor_pass": ("bn::factor_p",),
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


def pmc_counters(kernel_label: str, model: str = "c3a"):
    """(HBM bytes per launch, VALU-busy share of the launch's SIMD cycles) of the kernel behind a
    profile label, from the committed rocprofv3 --pmc summaries of this same command
    (tools/pmc_traffic.sh: FETCH_SIZE / WRITE_SIZE passes; tools/pmc_valu.sh: SQ_ACTIVE_INST_VALU
    against GRBM_GUI_ACTIVE).  A label can cover several instantiations (loop / reduce modes): the
    launch-weighted mean.  None where no summary covers the kernel."""
    stems = _KERNEL_STEMS.get(kernel_label.split("[")[0])
    if model != "c3a" or stems is None:  # the committed passes ran the default (784-leaf) command
        return None, None
    out = []
    for suffix, key in (("pmc_traffic.json", "hbm_bytes_per_launch"), ("pmc_valu.json", "valu_busy")):
        path = _latest_profile(suffix)
        hits = [v for k, v in json.load(open(path)).items() if k.startswith(stems)] if path else []
        hits = [h for h in hits if h.get(key) is not None]
        out.append(sum(h[key] * h["launches"] for h in hits) / sum(h["launches"] for h in hits) if hits else None)
    return out[0], out[1]


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


def secondary_em_step(ts, device, model, n, steps, warmup):
    """EM-step throughput of another workload on this GPU (resident evidence, same optimizer
    settings as the headline run): one dict for the ``secondary`` list."""
    from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings
    from bayesiannetwork_amd.optimizers import EmBatchOptimizer, EmBatchOptimizerSettings, OptimizerLogger

    iters = 10 if model == "c3b" else None
    if model == "c5":
        network, ys = build_deep_tree(ts)
        batches = [one_hot_evidence(ts, n, ys, 1 + b, device) for b in range(2)]
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
    t_used = factory(network).num_iterations
    workload = {"c3b": "Q(10)->F1,F2(2)->784xY(2|F1,F2) loopy model",
                "c5": "depth-6 binary tree of 64-state hidden nodes + 32 observed 64-state leaves"}[model]
    return dict(workload=f"{workload}, EM step (T={t_used})", batch=n, steps=steps, warmup=warmup,
                value=n * steps / elapsed, unit="samples/s", ms_per_step=elapsed / steps * 1e3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="observations per GPU per step")
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

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1 or "RANK" in os.environ:
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
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
    n = args.batch or (131072 if args.model == "c5" else 60000)
    iters = 10 if args.model == "c3b" else None
    # two distinct resident batches, alternated, so no step re-reads the evidence it just used
    if args.model == "c5":
        network, ys = build_deep_tree(ts)
        batches = [one_hot_evidence(ts, n, ys, 1 + 17 * rank + b, device) for b in range(2)]
    else:
        network, ys = build_model(ts, args.model)
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
        if host_batches is not None:
            opt = EmBatchOptimizer(
                network, factory,
                EmBatchOptimizerSettings(learning_rate=0.1, num_epochs=1, regularization=0.01,
                                         sharded=world > 1, deferred_logging=True), logger)
            opt.optimize(PrefetchingLoader(_Batches([host_batches[i % 2] for i in range(steps)], n * steps), device))
            return
        opt = EmBatchOptimizer(
            network, factory,
            EmBatchOptimizerSettings(learning_rate=0.1, num_epochs=1, regularization=0.01,
                                     sharded=world > 1 or bool(os.environ.get("BN_BENCH_FORCE_SHARDED")),
                                     deferred_logging=True),
            logger)
        opt.optimize(_Batches([batches[i % 2] for i in range(steps)], n * steps))

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    run(args.warmup)
    kernels = []
    if not args.no_kernel_timing:
        # every kernel's durations from a short untimed pass; inside the timed region only the
        # dominant kernel carries events (a timed dispatch costs a few microseconds extra)
        enable_profiling(True)
        profile_report()
        run(min(args.steps, 5))
        torch.cuda.synchronize()
        kernels = profile_report()
        top = max(kernels, key=lambda k: k["total_ms"])
        dominant = top["kernel"]
        # a timed dispatch is fenced off from its neighbours (a few microseconds each): inside the
        # timed region every stride-th launch of the dominant kernel is timed, with a stride coprime
        # to its launches per step so that all of its modes are sampled evenly
        per_step = max(1, round(top["launches"] / min(args.steps, 5)))
        stride = next(s for s in range(5, 64) if math.gcd(s, per_step) == 1)
        enable_profiling(True, only=dominant, stride=stride)
    fence()
    t0 = time.perf_counter()
    run(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = t.item()

    timed = []
    if not args.no_kernel_timing:
        timed = profile_report()
        enable_profiling(False)

    if rank == 0:
        machine = factory(network)
        t_used = machine.num_iterations
        lls = [log.ll for log in logger.logs]
        roofline = None
        if timed:
            dom = max(timed, key=lambda k: k["total_ms"])
            avg_s = dom["total_ms"] / dom["launches"] / 1e3
            # the committed counter passes ran the default command: float32, 60 000 resident observations
            default_workload = args.dtype == "float32" and args.evidence == "resident" and n == 60000
            # The leaf-family kernel keeps the messages of an iteration in registers, so the
            # message-level byte count (every message read and written once) is not a bound on it; it
            # is reported as `alg_equivalent`.  The fraction is taken against what the kernel can be
            # bound by: HBM bytes actually moved (counter passes) per live launch time, or the share of
            # SIMD cycles in which a vector instruction executes, whichever is larger.
            traffic, valu_busy = pmc_counters(dom["kernel"], args.model if default_workload else "other")
            alg = dict(bytes_per_launch=dom["alg_bytes"], GBps=dom["alg_bytes"] / avg_s / 1e9,
                       x_hbm_peak=dom["alg_bytes"] / avg_s / 1e9 / HBM_PEAK_GBS)
            hbm = None
            if traffic is not None:
                hbm = dict(achieved=traffic / avg_s / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                           frac=traffic / avg_s / 1e9 / HBM_PEAK_GBS)
            if hbm is not None and (valu_busy is None or hbm["frac"] >= valu_busy):
                roofline = dict(bound="hbm", kernel=dom["kernel"], **hbm)
            elif valu_busy is not None:
                roofline = dict(bound="valu", kernel=dom["kernel"], achieved=valu_busy, peak=1.0,
                                unit="share of SIMD cycles issuing vector instructions", frac=valu_busy)
            else:
                roofline = dict(bound="hbm", kernel=dom["kernel"], achieved=None, peak=HBM_PEAK_GBS, unit="GB/s",
                                frac=None)
            roofline.update(traffic=traffic, hbm=hbm, valu_busy=valu_busy, avg_launch_ms=avg_s * 1e3,
                            alg_equivalent=alg)
        out = {
            "metric": "evidence samples/sec (sum-product infer + EM step), MNIST 784-leaf latent model",
            "value": world * n * args.steps / elapsed,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.dtype == "float32" else "f64",
            "data": "synthetic",
            "config": {
                "workload": {"c3a": "Q(10)->784xY(2) latent-class model",
                             "c3b": "Q(10)->F1,F2(2)->784xY(2|F1,F2) loopy model",
                             "c5": "depth-6 binary tree of 64-state hidden nodes + 32 observed 64-state leaves"}[args.model]
                + f", EM step (T={t_used} flooding iterations + ll + expected counts + M-step)",
                "batch_per_gpu": n,
                "evidence": args.evidence,
                "global_batch": n * world,
                "learning_rate": 0.1,
                "regularization": 0.01,
                "parallelism": f"observations sharded over {world} GPU(s), counts all-reduced" if world > 1 else "single GPU",
            },
            "ll_first_last": [lls[-args.steps], lls[-1]] if len(lls) >= args.steps else None,
            "roofline": roofline,
            "kernels": [
                dict(kernel=k["kernel"], launches=k["launches"], avg_ms=k["total_ms"] / k["launches"],
                     alg_GBps=(k["alg_bytes"] / (k["total_ms"] / k["launches"] / 1e3) / 1e9) if k["total_ms"] > 0 else None)
                for k in kernels
            ],
        }
        if world == 1 and not args.no_cpu_baseline and args.model != "c5":
            out["cpu_baseline"] = cpu_baseline(args.model, 7500, iters)
        if world == 1 and not args.no_secondary and args.model == "c3a" and default_size:
            # the other two EM workloads of the config list, a few steps each, same optimizer settings
            del machine
            out["secondary"] = [secondary_em_step(ts, device, "c3b", 60000, steps=10, warmup=3),
                                secondary_em_step(ts, device, "c5", 131072, steps=2, warmup=1)]
        print(json.dumps(out))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
