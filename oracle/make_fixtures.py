"""Generate the golden fixtures in ``tests/golden/`` from the REAL reference package.

Runs only in the build container (it imports ``/root/reference``); the GPU box
never sees it run.  Usage, from the repository root::

    mkdir -p /tmp/oracle_stubs
    printf 'import pydantic\\nclass BaseSettings(pydantic.BaseModel):\\n    model_config = pydantic.ConfigDict(arbitrary_types_allowed=True)\\n' \\
        > /tmp/oracle_stubs/pydantic_settings.py
    R=/root/reference/packages PYTHONDONTWRITEBYTECODE=1 \\
    PYTHONPATH=/tmp/oracle_stubs:$R/bayesian_network/src:$R/common/src:. \\
        python3 oracle/make_fixtures.py

(``pydantic_settings`` is not installed here; the three-line stand-in above only
gives the reference's two-field settings object a base class.)

Every fixture is one ``.npz`` (no pickles) holding

* ``meta``   JSON: node list in construction order with parent names, observed
             node names, N, iteration settings, dtype, network node order;
* ``cpt/<node>``, ``ev/<node>``      the inputs, in the fixture's dtype;
* ``out/...``   what the reference's ``spa_v3`` machine returned in that dtype;
* ``truth/...`` (float32 fixtures) the same queries answered by the reference in
             float64 on the up-cast inputs;
* for EM fixtures ``em/...``: CPTs after each optimizer step and the logged ll.

While writing, the script also runs this repository's CPU oracle on the same
reference network objects and aborts if they disagree, so a fixture file never
exists without the oracle having matched the reference on it.
"""

from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

from bayesian_network.bayesian_network import BayesianNetwork, Node
from bayesian_network.inference_machines.evidence import Evidence
from bayesian_network.inference_machines.naive.naive_inference_machine import NaiveInferenceMachine
from bayesian_network.inference_machines.common import InferenceMachineSettings
from bayesian_network.inference_machines.spa_v3.spa_inference_machine import (
    SpaInferenceMachine,
    SpaInferenceMachineSettings,
)
from bayesian_network.optimizers.em_batch_optimizer import EmBatchOptimizer, EmBatchOptimizerSettings
from bayesian_network.optimizers.em_optimizer import EmOptimizer, EmOptimizerSettings
from bayesian_network.optimizers.common import OptimizerLogger
from common.torch_settings import TorchSettings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.spa_oracle import SpaOracle  # noqa: E402

OUT_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
TDT = {"float64": torch.float64, "float32": torch.float32}


# ----------------------------------------------------------------- helpers
def rand_cpt(gen, shape):
    p = torch.rand(shape, generator=gen, dtype=torch.float64)
    return p / p.sum(dim=-1, keepdim=True)


def soft_rows(gen, n, s):
    p = torch.rand((n, s), generator=gen, dtype=torch.float64) + 1e-3
    return p / p.sum(dim=-1, keepdim=True)


def one_hot_rows(gen, n, s, gamma=None):
    idx = torch.randint(0, s, (n,), generator=gen)
    e = torch.nn.functional.one_hot(idx, s).to(torch.float64)
    if gamma is not None:
        e = e * gamma + (1 - gamma) / 2
    return e


def to_dtype_cpt(cpt64, dtype):
    c = cpt64.to(dtype)
    return c / c.sum(dim=-1, keepdim=True)


def build_reference(spec, cpts, dtype_name):
    ts = TorchSettings(device="cpu", dtype=dtype_name)
    by_name = {}
    nodes = []
    parents = {}
    for name, parent_names in spec["nodes"]:
        node = Node(cpts[name].to(TDT[dtype_name]), name=name)
        by_name[name] = node
        nodes.append(node)
        parents[node] = None
    for name, parent_names in spec["nodes"]:
        parents[by_name[name]] = [by_name[p] for p in parent_names]
    network = BayesianNetwork(nodes, parents)
    observed = [by_name[o] for o in spec["observed"]]
    return ts, network, observed, by_name


def make_machine(ts, network, observed, N, spec, average):
    return SpaInferenceMachine(
        settings=SpaInferenceMachineSettings(
            torch_settings=ts,
            average_log_likelihood=average,
            num_iterations=spec.get("num_iterations"),
            allow_loops=spec.get("allow_loops", False),
        ),
        bayesian_network=network,
        observed_nodes=observed,
        num_observations=N,
    )


def query_reference(spec, cpts, evidence, dtype_name, want):
    """Run the reference machine; returns dict of numpy arrays."""
    ts, network, observed, by_name = build_reference(spec, cpts, dtype_name)
    N = spec["N"]
    im = make_machine(ts, network, observed, N, spec, average=False)
    if observed:
        im.enter_evidence(Evidence([evidence[o].to(TDT[dtype_name]) for o in spec["observed"]], ts))
    order = [n.name for n in network.nodes]
    res = {}
    singles = im.infer_single_nodes(list(network.nodes))
    families = im.infer_nodes_with_parents(list(network.nodes))
    for name, s, f in zip(order, singles, families):
        if want["single"] == "all" or name in want["single"]:
            res[f"single/{name}"] = s.numpy().copy()
        if want["family"] == "all" or name in want["family"]:
            res[f"family/{name}"] = f.numpy().copy()
        if want.get("counts", False):
            res[f"counts/{name}"] = f.mean(dim=0).numpy().copy()
    if observed:
        res["ll_sum"] = np.array(im.log_likelihood())
        lls = torch.cat([g.local_log_likelihoods for g in im.factor_graph.variable_node_groups])
        res["ll_avg"] = np.array(lls.sum(dim=0).mean().item())
        # per-node normalisers c[node, n], network order
        c = {}
        for g in im.factor_graph.variable_node_groups:
            for k, node in enumerate(g.nodes):
                c[node.name] = g.local_log_likelihoods[k].numpy().copy()
        res["c"] = np.stack([c[name] for name in order])
    if want.get("messages", False):
        fg = im.factor_graph
        for node in network.nodes:
            vgroup = fg.get_variable_node_group(node)
            fgroup = fg.get_factor_node_group(node)
            # local edge
            res[f"f2v/{node.name}/{node.name}"] = vgroup.get_input_tensor(node, node).numpy().copy()
            res[f"v2f/{node.name}/{node.name}"] = fgroup.get_input_tensor(node, node).numpy().copy()
            for parent in network.parents_of(node):
                pg = fg.get_variable_node_group(parent)
                res[f"f2v/{parent.name}/{node.name}"] = pg.get_input_tensor(parent, node).numpy().copy()
                res[f"v2f/{parent.name}/{node.name}"] = fgroup.get_input_tensor(node, parent).numpy().copy()
    res["_T_used"] = np.array(im._num_iterations)
    res["_order"] = order
    return res, (ts, network, observed, im)


def check_oracle(spec, ts, network, observed, evidence, dtype_name, ref, tol):
    oracle = SpaOracle(
        network,
        observed,
        spec["N"],
        dtype=TDT[dtype_name],
        num_iterations=spec.get("num_iterations"),
        allow_loops=spec.get("allow_loops", False),
    )
    assert oracle.num_iterations == int(ref["_T_used"]), (oracle.num_iterations, ref["_T_used"])
    if observed:
        oracle.enter_evidence([evidence[o].to(TDT[dtype_name]) for o in spec["observed"]])
    order = ref["_order"]
    worst = 0.0
    for j, name in enumerate(order):
        for kind, fn in (("single", oracle.infer_single), ("family", oracle.infer_family)):
            key = f"{kind}/{name}"
            if key in ref:
                got = fn(j).numpy()
                err = np.max(np.abs(got - ref[key]) / (np.abs(ref[key]) + 1e-300))
                worst = max(worst, float(err))
    if observed:
        for avg, key in ((False, "ll_sum"), (True, "ll_avg")):
            got = oracle.log_likelihood(avg)
            worst = max(worst, abs(got - float(ref[key])) / abs(float(ref[key])))
        got_c = oracle.local_log_likelihoods().numpy()
        worst = max(worst, float(np.max(np.abs(got_c - ref["c"]) / (np.abs(ref["c"]) + 1e-300))))
    for key in ref:
        if key.startswith("f2v/") or key.startswith("v2f/"):
            kind, var, owner = key.split("/")
            iv, io = order.index(var), order.index(owner)
            got = (oracle.message_f2v if kind == "f2v" else oracle.message_v2f)(iv, io).numpy()
            worst = max(worst, float(np.max(np.abs(got - ref[key]) / (np.abs(ref[key]) + 1e-300))))
    assert worst < tol, f"oracle disagrees with the reference on {spec['name']}/{dtype_name}: {worst}"
    return worst


def save(spec, dtype_name, cpts, evidence, ref, truth=None, extra=None):
    arrays = {}
    meta = dict(
        name=spec["name"],
        dtype=dtype_name,
        nodes=[[n, list(p)] for n, p in spec["nodes"]],
        observed=list(spec["observed"]),
        N=spec["N"],
        num_iterations=spec.get("num_iterations"),
        allow_loops=spec.get("allow_loops", False),
        T_used=int(ref["_T_used"]),
        network_order=list(ref["_order"]),
        source=spec.get("source", ""),
    )
    np_dt = np.float64 if dtype_name == "float64" else np.float32
    for name, c in cpts.items():
        arrays[f"cpt/{name}"] = c.numpy().astype(np_dt)
    for name, e in evidence.items():
        arrays[f"ev/{name}"] = e.numpy().astype(np_dt)
    for key, val in ref.items():
        if not key.startswith("_"):
            arrays[f"out/{key}"] = val
    if truth is not None:
        for key, val in truth.items():
            if not key.startswith("_"):
                arrays[f"truth/{key}"] = val
    if extra:
        arrays.update(extra)
    arrays["meta"] = np.array(json.dumps(meta))
    path = os.path.join(OUT_DIR, f"{spec['name']}_{'f64' if dtype_name == 'float64' else 'f32'}.npz")
    np.savez_compressed(path, **arrays)
    size = os.path.getsize(path)
    print(f"  wrote {os.path.basename(path)}  {size / 1024:.1f} KiB")


def emit(spec, cpts64, evidence64, want, dtypes=("float64", "float32"), naive_check=False):
    print(spec["name"])
    for dtype_name in dtypes:
        cpts = {k: to_dtype_cpt(v, TDT[dtype_name]) for k, v in cpts64.items()}
        evidence = {k: v.to(TDT[dtype_name]) for k, v in evidence64.items()}
        ref, (ts, network, observed, im) = query_reference(spec, cpts, evidence, dtype_name, want)
        tol = 1e-11 if dtype_name == "float64" else 2e-3
        worst = check_oracle(spec, ts, network, observed, evidence, dtype_name, ref, tol)
        print(f"  {dtype_name}: oracle vs reference worst rel err {worst:.2e}")
        truth = None
        if dtype_name == "float32":
            cpts_up = {k: v.to(torch.float64) for k, v in cpts.items()}
            cpts_up = {k: v / v.sum(dim=-1, keepdim=True) for k, v in cpts_up.items()}
            ev_up = {k: v.to(torch.float64) for k, v in evidence.items()}
            truth, _ = query_reference(spec, cpts_up, ev_up, "float64", want)
        if naive_check and dtype_name == "float64":
            naive = NaiveInferenceMachine(
                settings=InferenceMachineSettings(ts, average_log_likelihood=False),
                bayesian_network=network,
                observed_nodes=observed,
            )
            naive.enter_evidence(Evidence([evidence[o] for o in spec["observed"]], ts))
            exact = naive.infer_single_nodes(list(network.nodes))
            for name, p in zip(ref["_order"], exact):
                if f"single/{name}" in ref:
                    assert np.allclose(p.numpy(), ref[f"single/{name}"], rtol=1e-9, atol=1e-12), name
            # The log-likelihood is NOT compared: with T = diameter the reference's sum of
            # per-node normalisers still mixes normalisers from different iterations and
            # differs from the exact value (docs-1 net: -3.30525 vs exact -3.30443).  Parity
            # target is the reference's own number, which the oracle reproduces above.
            print(f"  exact (brute-force) machine agrees on marginals; ll exact {naive.log_likelihood():.10f} "
                  f"vs spa {float(ref['ll_sum']):.10f}")
        save(spec, dtype_name, cpts, evidence, ref, truth)


ALL = dict(single="all", family="all", messages=True)


# ----------------------------------------------------------------- configs
def config_docs1():
    spec = dict(
        name="c1_docs1",
        nodes=[("Q", []), ("Y", ["Q"])],
        observed=["Y"],
        N=3,
        source="docs/1. Simple Bayesian network.md:56-57 (CPTs), :70-75 (posteriors)",
    )
    cpts = {
        "Q": torch.tensor([0.4, 0.6], dtype=torch.float64),
        "Y": torch.tensor([[0.1, 0.6, 0.3], [0.5, 0.1, 0.4]], dtype=torch.float64),
    }
    evidence = {"Y": torch.eye(3, dtype=torch.float64)}
    emit(spec, cpts, evidence, ALL, naive_check=True)


def docs2_cpts():
    return {
        "Q1": torch.tensor([0.4, 0.6], dtype=torch.float64),
        "Q2": torch.tensor([[0.3, 0.7], [0.2, 0.8]], dtype=torch.float64),
        "Y1": torch.tensor([[0.1, 0.6, 0.3], [0.5, 0.1, 0.4]], dtype=torch.float64),
        "Y2": torch.tensor([[0.5, 0.3, 0.2], [0.4, 0.2, 0.4]], dtype=torch.float64),
        "Y3": torch.tensor([[0.4, 0.6], [0.9, 0.1]], dtype=torch.float64),
    }


DOCS2_NODES = [("Q1", []), ("Q2", ["Q1"]), ("Y1", ["Q1"]), ("Y2", ["Q2"]), ("Y3", ["Q2"])]


def config_docs2():
    gen = torch.Generator().manual_seed(2)
    n_soft = 64
    spec = dict(
        name="c2_docs2",
        nodes=DOCS2_NODES,
        observed=["Y1", "Y2", "Y3"],
        N=1 + n_soft,
        source="docs/2. Complex Bayesian Network.md:51-60 (CPTs, evidence), :909-914 (posteriors, ll)",
    )
    evidence = {
        "Y1": torch.cat([torch.tensor([[0.0, 1.0, 0.0]], dtype=torch.float64), soft_rows(gen, n_soft, 3)]),
        "Y2": torch.cat([torch.tensor([[0.0, 0.0, 1.0]], dtype=torch.float64), soft_rows(gen, n_soft, 3)]),
        "Y3": torch.cat([torch.tensor([[1.0, 0.0]], dtype=torch.float64), soft_rows(gen, n_soft, 2)]),
    }
    emit(spec, docs2_cpts(), evidence, ALL, naive_check=True)


def config_chain():
    # the reference's generic suite, "NetworkWithSingleParents": Q1 -> Q2 -> Y
    gen = torch.Generator().manual_seed(3)
    spec = dict(name="chain3", nodes=[("Q1", []), ("Q2", ["Q1"]), ("Y", ["Q2"])], observed=["Y"], N=7)
    cpts = {"Q1": rand_cpt(gen, (2,)), "Q2": rand_cpt(gen, (2, 3)), "Y": rand_cpt(gen, (3, 4))}
    evidence = {"Y": one_hot_rows(gen, 7, 4, gamma=1 - 1e-9)}
    emit(spec, cpts, evidence, ALL, naive_check=True)
    # all nodes observed, and nothing observed (N=0 -> 1)
    spec2 = dict(spec, name="chain3_allobs", observed=["Q1", "Q2", "Y"], N=5)
    ev2 = {
        "Q1": one_hot_rows(gen, 5, 2, gamma=1 - 1e-9),
        "Q2": one_hot_rows(gen, 5, 3, gamma=1 - 1e-9),
        "Y": one_hot_rows(gen, 5, 4, gamma=1 - 1e-9),
    }
    emit(spec2, cpts, ev2, ALL, naive_check=True)
    spec3 = dict(spec, name="chain3_noobs", observed=[], N=0)
    emit(spec3, cpts, {}, dict(single="all", family="all", messages=True))


def config_tree8():
    # the generic suite's "ComplexNetworkWithSingleParents": 8-node tree, cardinalities 2/3/4
    gen = torch.Generator().manual_seed(4)
    nodes = [
        ("Q1", []),
        ("Q2", ["Q1"]),
        ("Q3", ["Q2"]),
        ("Y1", ["Q1"]),
        ("Y2", ["Q2"]),
        ("Y3", ["Q3"]),
        ("Y4", ["Q3"]),
        ("Y5", ["Q3"]),
    ]
    card = dict(Q1=2, Q2=3, Q3=4, Y1=2, Y2=3, Y3=4, Y4=2, Y5=3)
    cpts = {}
    for name, ps in nodes:
        cpts[name] = rand_cpt(gen, tuple(card[p] for p in ps) + (card[name],))
    obs = ["Y1", "Y2", "Y3", "Y4", "Y5"]
    N = 33
    evidence = {o: one_hot_rows(gen, N, card[o], gamma=1 - 1e-9) for o in obs}
    emit(dict(name="tree8", nodes=nodes, observed=obs, N=N), cpts, evidence, ALL, naive_check=True)


def config_mixed_group():
    # observed 1-child node and unobserved 2-child node share a (3 inputs, 2 states) group;
    # child listed before its parent in construction order so network order != list order
    gen = torch.Generator().manual_seed(5)
    nodes = [("C", ["B"]), ("A", []), ("B", ["A"]), ("D", ["A"]), ("E", ["D"])]
    card = dict(A=2, B=2, C=3, D=2, E=2)
    cpts = {}
    for name, ps in nodes:
        cpts[name] = rand_cpt(gen, tuple(card[p] for p in ps) + (card[name],))
    obs = ["E", "B", "C"]
    N = 21
    evidence = {o: soft_rows(gen, N, card[o]) for o in obs}
    # no brute-force check here: the reference's exact machine reshapes CPTs assuming every
    # parent precedes its child in network order, which this construction order violates
    emit(dict(name="mixed_group", nodes=nodes, observed=obs, N=N), cpts, evidence, ALL)
    topo = [("A", []), ("B", ["A"]), ("C", ["B"]), ("D", ["A"]), ("E", ["D"])]
    emit(dict(name="mixed_group_topo", nodes=topo, observed=obs, N=N), cpts, evidence, ALL, naive_check=True)


def config_polytree():
    gen = torch.Generator().manual_seed(6)
    nodes = [("A", []), ("B", []), ("C", ["A", "B"]), ("D", ["C"])]
    card = dict(A=2, B=3, C=4, D=3)
    cpts = {}
    for name, ps in nodes:
        cpts[name] = rand_cpt(gen, tuple(card[p] for p in ps) + (card[name],))
    N = 19
    evidence = {"D": soft_rows(gen, N, 3), "A": soft_rows(gen, N, 2)}
    emit(dict(name="polytree", nodes=nodes, observed=["D", "A"], N=N), cpts, evidence, ALL, naive_check=True)


def config_loopy3():
    # the generic suite's "NetworkWithMultipleParents" (loopy), spa_v3 with explicit iterations
    gen = torch.Generator().manual_seed(7)
    nodes = [("Q1", []), ("Q2", ["Q1"]), ("Y", ["Q1", "Q2"])]
    cpts = {"Q1": rand_cpt(gen, (2,)), "Q2": rand_cpt(gen, (2, 3)), "Y": rand_cpt(gen, (2, 3, 4))}
    N = 9
    evidence = {"Y": one_hot_rows(gen, N, 4, gamma=1 - 1e-9)}
    emit(
        dict(name="loopy3", nodes=nodes, observed=["Y"], N=N, num_iterations=10, allow_loops=True),
        cpts,
        evidence,
        ALL,
    )


def config_loops_c4():
    gen = torch.Generator().manual_seed(0)
    nodes = [("Q0", []), ("Q1", ["Q0"]), ("Q2", ["Q0", "Q1"]), ("Y", ["Q0", "Q2"])]
    cpts = {
        "Q0": rand_cpt(gen, (2,)),
        "Q1": rand_cpt(gen, (2, 3)),
        "Q2": rand_cpt(gen, (2, 3, 4)),
        "Y": rand_cpt(gen, (2, 4, 5)),
    }
    gen = torch.Generator().manual_seed(1)
    N = 5
    evidence = {"Y": one_hot_rows(gen, N, 5)}
    emit(
        dict(
            name="c4_loops",
            nodes=nodes,
            observed=["Y"],
            N=N,
            num_iterations=10,
            allow_loops=True,
            source="dev_experiments/loops/loops.py:34-58 (shapes), :69-75 (one-hot evidence)",
        ),
        cpts,
        evidence,
        ALL,
    )


def mnist_like(num_pixels, literal, N, name, want, seed_cpt=0, seed_ev=1):
    gen = torch.Generator().manual_seed(seed_cpt)
    nodes = [("Q", [])]
    cpts = {"Q": torch.full((10,), 0.1, dtype=torch.float64)}
    if literal:
        nodes += [("F1", ["Q"]), ("F2", ["Q"])]
        cpts["F1"] = rand_cpt(gen, (10, 2))
        cpts["F2"] = rand_cpt(gen, (10, 2))
    obs = []
    for i in range(num_pixels):
        y = f"Y{i}"
        if literal:
            nodes.append((y, ["F1", "F2"]))
            cpts[y] = rand_cpt(gen, (2, 2, 2))
        else:
            nodes.append((y, ["Q"]))
            cpts[y] = rand_cpt(gen, (10, 2))
        obs.append(y)
    gen = torch.Generator().manual_seed(seed_ev)
    x = (torch.rand((N, num_pixels), generator=gen, dtype=torch.float64) < 0.2).to(torch.float64)
    x = x * 0.999999 + 5e-7
    evidence = {f"Y{i}": torch.stack([1 - x[:, i], x[:, i]], dim=1) for i in range(num_pixels)}
    spec = dict(
        name=name,
        nodes=nodes,
        observed=obs,
        N=N,
        source="dev_experiments/mnist/mnist.py:63-91 (model), :97-110 (evidence transform)",
    )
    if literal:
        spec.update(num_iterations=10, allow_loops=True)
    emit(spec, cpts, evidence, want)


def config_c5_mini():
    # depth-3 binary tree of 64-state hidden nodes, one observed 64-state leaf under each hidden leaf
    gen = torch.Generator().manual_seed(0)
    S = 64
    nodes = [("H0", [])]
    cpts = {"H0": rand_cpt(gen, (S,))}
    hidden_leaves = []
    frontier = ["H0"]
    count = 1
    for depth in range(2):
        nxt = []
        for parent in frontier:
            for _ in range(2):
                name = f"H{count}"
                count += 1
                nodes.append((name, [parent]))
                cpts[name] = rand_cpt(gen, (S, S))
                nxt.append(name)
        frontier = nxt
    hidden_leaves = frontier
    obs = []
    for i, h in enumerate(hidden_leaves):
        name = f"Y{i}"
        nodes.append((name, [h]))
        cpts[name] = rand_cpt(gen, (S, S))
        obs.append(name)
    gen = torch.Generator().manual_seed(1)
    N = 256
    evidence = {o: one_hot_rows(gen, N, S, gamma=1 - 1e-6) for o in obs}
    want = dict(single="all", family=[], counts=True, messages=False)
    emit(dict(name="c5_mini", nodes=nodes, observed=obs, N=N), cpts, evidence, want)


# ---------------------------------------------------------------------- EM
def em_fixture(name, spec, cpts64, evidence64, dtype_name, kind):
    print(name, dtype_name, kind)
    cpts = {k: to_dtype_cpt(v, TDT[dtype_name]) for k, v in cpts64.items()}
    evidence = {k: v.to(TDT[dtype_name]) for k, v in evidence64.items()}
    ts, network, observed, by_name = build_reference(spec, cpts, dtype_name)
    logger = OptimizerLogger()
    extra = {}
    order = [n.name for n in network.nodes]
    full = Evidence([evidence[o] for o in spec["observed"]], ts)

    def factory_for(n):
        def factory(net):
            return make_machine(ts, net, observed, n, spec, average=True)

        return factory

    if kind == "full":
        steps = 3
        snapshots = []

        class Snap(OptimizerLogger):
            def log(self, epoch, iteration, ll):
                super().log(epoch, iteration, ll)
                snapshots.append({n.name: n.cpt.numpy().copy() for n in network.nodes})

        logger = Snap()
        EmOptimizer(network, factory_for(spec["N"]), EmOptimizerSettings(num_iterations=steps), logger).optimize(full)
        meta_em = dict(kind="full", steps=steps)
    else:
        batch = spec["N"] // 2
        snapshots = []

        class Snap(OptimizerLogger):
            def log(self, epoch, iteration, ll):
                super().log(epoch, iteration, ll)
                snapshots.append({n.name: n.cpt.numpy().copy() for n in network.nodes})

        logger = Snap()

        class Loader:
            num_observations = spec["N"]

            def __len__(self):
                return 2

            def __iter__(self):
                yield full[0:batch]
                yield full[batch : 2 * batch]

        settings = EmBatchOptimizerSettings(learning_rate=0.1, num_epochs=2, regularization=0.01)
        EmBatchOptimizer(network, factory_for(batch), settings, logger).optimize(Loader())
        meta_em = dict(kind="batch", batch=batch, learning_rate=0.1, num_epochs=2, regularization=0.01)

    for step, snap in enumerate(snapshots):
        for nname, arr in snap.items():
            extra[f"em/step{step}/{nname}"] = arr
    extra["em/ll"] = np.array([log.ll for log in logger.logs])
    extra["em/meta"] = np.array(json.dumps(meta_em))
    ref = {"_T_used": np.array(spec.get("num_iterations") or 0), "_order": order}
    # T_used: take it from a fresh machine
    ref["_T_used"] = np.array(factory_for(spec["N"])(network)._num_iterations)
    spec2 = dict(spec, name=name)
    save(spec2, dtype_name, cpts, evidence, ref, extra=extra)


def config_em():
    gen = torch.Generator().manual_seed(11)
    N = 40
    spec = dict(name="em_docs2", nodes=DOCS2_NODES, observed=["Y1", "Y2", "Y3"], N=N)
    evidence = {
        "Y1": one_hot_rows(gen, N, 3, gamma=1 - 1e-6),
        "Y2": one_hot_rows(gen, N, 3, gamma=1 - 1e-6),
        "Y3": one_hot_rows(gen, N, 2, gamma=1 - 1e-6),
    }
    for dt in ("float64", "float32"):
        em_fixture("em_full_docs2", spec, docs2_cpts(), evidence, dt, "full")
        em_fixture("em_batch_docs2", spec, docs2_cpts(), evidence, dt, "batch")

    # 16-pixel literal mnist.py model, loopy, T=10
    gen = torch.Generator().manual_seed(0)
    nodes = [("Q", []), ("F1", ["Q"]), ("F2", ["Q"])]
    cpts = {"Q": torch.full((10,), 0.1, dtype=torch.float64), "F1": rand_cpt(gen, (10, 2)), "F2": rand_cpt(gen, (10, 2))}
    obs = []
    for i in range(16):
        nodes.append((f"Y{i}", ["F1", "F2"]))
        cpts[f"Y{i}"] = rand_cpt(gen, (2, 2, 2))
        obs.append(f"Y{i}")
    gen = torch.Generator().manual_seed(1)
    N = 32
    x = (torch.rand((N, 16), generator=gen, dtype=torch.float64) < 0.2).to(torch.float64) * 0.999999 + 5e-7
    evidence = {f"Y{i}": torch.stack([1 - x[:, i], x[:, i]], dim=1) for i in range(16)}
    spec = dict(name="em_c3b", nodes=nodes, observed=obs, N=N, num_iterations=10, allow_loops=True)
    for dt in ("float64", "float32"):
        em_fixture("em_full_c3b_mini16", spec, cpts, evidence, dt, "full")
        em_fixture("em_batch_c3b_mini16", spec, cpts, evidence, dt, "batch")

    # 16-pixel star (the 784-leaf model in miniature), tree, T=4
    gen = torch.Generator().manual_seed(0)
    nodes = [("Q", [])]
    cpts = {"Q": torch.full((10,), 0.1, dtype=torch.float64)}
    for i in range(16):
        nodes.append((f"Y{i}", ["Q"]))
        cpts[f"Y{i}"] = rand_cpt(gen, (10, 2))
    spec = dict(name="em_c3a", nodes=nodes, observed=obs, N=N)
    for dt in ("float64", "float32"):
        em_fixture("em_batch_c3a_mini16", spec, cpts, evidence, dt, "batch")


def main():
    os.makedirs(OUT_DIR, exist_ok=True)
    torch.set_num_threads(4)
    config_docs1()
    config_docs2()
    config_chain()
    config_tree8()
    config_mixed_group()
    config_polytree()
    config_loopy3()
    config_loops_c4()
    mnist_like(16, False, 32, "c3a_mini16", ALL)
    mnist_like(16, True, 32, "c3b_mini16", ALL)
    slim = dict(single=["Q", "F1", "F2", "Y0", "Y391", "Y783"], family=["Q", "Y0", "Y783"], counts=True, messages=False)
    mnist_like(784, False, 64, "c3a_784", slim)
    mnist_like(784, True, 64, "c3b_784", slim)
    config_c5_mini()
    config_em()


if __name__ == "__main__":
    main()
