import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Fixture:
    """One golden ``.npz``: inputs plus what the reference returned for them."""

    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
        self.meta = json.loads(str(self.z["meta"]))
        self.dtype = torch.float64 if self.meta["dtype"] == "float64" else torch.float32

    def __contains__(self, key):
        return key in self.z.files

    def __getitem__(self, key):
        return self.z[key]

    def keys(self, prefix):
        return [k for k in self.z.files if k.startswith(prefix)]

    def build(self, torch_settings, cpt_override=None):
        """Network, observed nodes, name->node map, built with this repository's classes."""
        from bayesiannetwork_amd.bayesian_network import BayesianNetwork, Node

        by_name, nodes, parents = {}, [], {}
        for name, _ in self.meta["nodes"]:
            cpt = torch.from_numpy(self.z["cpt/" + name]).to(torch_settings.device, torch_settings.dtype)
            if cpt_override and name in cpt_override:
                cpt = cpt_override[name]
            node = Node(cpt, name=name)
            by_name[name] = node
            nodes.append(node)
        for name, parent_names in self.meta["nodes"]:
            parents[by_name[name]] = [by_name[p] for p in parent_names]
        network = BayesianNetwork(nodes, parents)
        observed = [by_name[o] for o in self.meta["observed"]]
        return network, observed, by_name

    def evidence_tensors(self, device="cpu"):
        return [torch.from_numpy(self.z["ev/" + o]).to(device) for o in self.meta["observed"]]


def fixture_names(prefix_exclude=("em_", "naive_")):
    names = sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz"))
    return [n for n in names if not n.startswith(prefix_exclude)]


def naive_fixture_names():
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz") and f.startswith("naive_"))


def em_fixture_names():
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz") and f.startswith("em_"))


@pytest.fixture
def load_fixture():
    return Fixture
