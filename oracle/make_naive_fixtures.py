"""Generate ``tests/golden/naive_*.npz`` from the REAL reference's brute-force machine.

Runs only in the build container (imports ``/root/reference``), with the same environment as
``oracle/make_fixtures.py`` (see its docstring)::

    R=/root/reference/packages PYTHONDONTWRITEBYTECODE=1 \\
    PYTHONPATH=/tmp/oracle_stubs:$R/bayesian_network/src:$R/common/src:. \\
        python3 oracle/make_naive_fixtures.py

Inputs are those of existing fixtures (same CPTs, same evidence) whose construction order lists
every parent before its children — the only order the reference's machine handles.  Each output
file holds ``meta`` (as the input fixture's), the inputs again, and

* ``naive/single/<node>``  ``[N, S]``                          infer_single_nodes
* ``naive/family/<node>``  ``[N, ...]`` axes in network order   infer_nodes_with_parents
* ``naive/log_likelihoods`` ``[N]``, ``naive/ll_sum``, ``naive/ll_avg``

While writing, this repository's CPU restatement (``oracle/naive_oracle.py``) is run on the same
inputs and the script aborts if the two disagree.
"""

from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

from bayesian_network.inference_machines.common import InferenceMachineSettings
from bayesian_network.inference_machines.evidence import Evidence
from bayesian_network.inference_machines.naive.naive_inference_machine import NaiveInferenceMachine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.make_fixtures import OUT_DIR, TDT, build_reference  # noqa: E402
from oracle.naive_oracle import NaiveOracle  # noqa: E402

SOURCES = ["c1_docs1", "c2_docs2", "chain3", "chain3_allobs", "tree8", "mixed_group_topo", "polytree", "loopy3",
           "c4_loops"]


def main():
    for source in SOURCES:
        for tag, dtype_name in (("f64", "float64"), ("f32", "float32")):
            z = np.load(os.path.join(OUT_DIR, f"{source}_{tag}.npz"), allow_pickle=False)
            meta = json.loads(str(z["meta"]))
            spec = dict(name=meta["name"], nodes=[(n, p) for n, p in meta["nodes"]], observed=meta["observed"])
            cpts = {n: torch.from_numpy(z["cpt/" + n]) for n, _ in spec["nodes"]}
            evidence = {o: torch.from_numpy(z["ev/" + o]) for o in spec["observed"]}
            ts, network, observed, by_name = build_reference(spec, cpts, dtype_name)
            order = [n.name for n in network.nodes]
            assert order == meta["network_order"]
            for node in network.nodes:  # the reference's reshape needs ascending parent order
                idx = [order.index(p.name) for p in network.parents_of(node)]
                assert idx == sorted(idx) and all(i < order.index(node.name) for i in idx), source
            arrays = {}
            for average in (False, True):
                machine = NaiveInferenceMachine(InferenceMachineSettings(ts, average_log_likelihood=average), network,
                                                observed)
                machine.enter_evidence(Evidence([evidence[o].to(TDT[dtype_name]) for o in spec["observed"]], ts))
                arrays["naive/ll_avg" if average else "naive/ll_sum"] = np.array(machine.log_likelihood())
            singles = machine.infer_single_nodes(list(network.nodes))
            families = machine.infer_nodes_with_parents(list(network.nodes))
            for name, s, f in zip(order, singles, families):
                arrays[f"naive/single/{name}"] = s.numpy().copy()
                arrays[f"naive/family/{name}"] = f.numpy().copy()
            arrays["naive/log_likelihoods"] = machine._log_likelihoods.reshape(-1).numpy().copy()

            parents = [[order.index(p.name) for p in network.parents_of(n)] for n in network.nodes]
            oracle = NaiveOracle([n.cpt for n in network.nodes], parents, [order.index(o) for o in spec["observed"]],
                                 dtype=TDT[dtype_name])
            oracle.enter_evidence([evidence[o].to(TDT[dtype_name]) for o in spec["observed"]])
            tol = 1e-12 if dtype_name == "float64" else 1e-5
            worst = 0.0
            for j, name in enumerate(order):
                worst = max(worst, float((oracle.infer([j]) - singles[j]).abs().max()))
                fam = parents[j] + [j]
                worst = max(worst, float((oracle.infer(fam) - families[j]).abs().max()))
            worst = max(worst, float((oracle.log_likelihoods - machine._log_likelihoods.reshape(-1)).abs().max()))
            assert worst < tol, (source, dtype_name, worst)

            for key in z.files:
                if key.startswith("cpt/") or key.startswith("ev/"):
                    arrays[key] = z[key]
            meta["source"] = "inference_machines/naive/naive_inference_machine.py run on the inputs of " + source
            arrays["meta"] = np.array(json.dumps(meta))
            path = os.path.join(OUT_DIR, f"naive_{source}_{tag}.npz")
            np.savez_compressed(path, **arrays)
            print(f"wrote {os.path.basename(path)} {os.path.getsize(path) / 1024:.1f} KiB; oracle worst abs err {worst:.1e}")


if __name__ == "__main__":
    main()
