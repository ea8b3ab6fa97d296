"""ctypes binding of ``libbn_hip.so`` (C ABI in ``include/bn_hip.h``).

The library is built in-tree by ``make`` / ``__graft_entry__.build()`` and is the
only compute path: there is no CPU or eager fallback, and a missing library is
an ``ImportError`` at first use.  ``torch`` is imported first so that the
library binds to the HIP runtime PyTorch-ROCm already loaded (same SONAME),
which makes ``tensor.data_ptr()`` addresses and torch streams directly usable.
"""

from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_int, c_int32, c_int64, c_uint64, c_void_p

import torch  # noqa: F401  (loads the HIP runtime the library must share)

ABI_VERSION = 1
_LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "libbn_hip.so")
_lib = None


class BnError(RuntimeError):
    pass


def lib_path() -> str:
    return _LIB_PATH


def _declare(lib):
    i32p = POINTER(c_int32)
    lib.bn_abi_version.restype = c_int
    lib.bn_abi_version.argtypes = []
    lib.bn_last_error.restype = c_char_p
    lib.bn_last_error.argtypes = []
    lib.bn_plan_create.restype = c_int
    lib.bn_plan_create.argtypes = [c_int, c_int64, c_int32, i32p, i32p, i32p, i32p, i32p, i32p, c_int32,
                                   POINTER(c_void_p)]
    lib.bn_plan_destroy.restype = c_int
    lib.bn_plan_destroy.argtypes = [c_void_p]
    lib.bn_plan_topology.restype = c_int
    lib.bn_plan_topology.argtypes = [c_void_p, i32p, i32p]
    lib.bn_plan_cpt_elements.restype = c_int64
    lib.bn_plan_cpt_elements.argtypes = [c_void_p]
    lib.bn_plan_device_bytes.restype = c_int64
    lib.bn_plan_device_bytes.argtypes = [c_void_p]
    lib.bn_plan_unmapped_bytes.restype = c_int64
    lib.bn_plan_unmapped_bytes.argtypes = [c_void_p]
    lib.bn_plan_map_message_arrays.restype = c_int
    lib.bn_plan_map_message_arrays.argtypes = [c_void_p]
    lib.bn_em_step.restype = c_int
    lib.bn_em_step.argtypes = [c_void_p, c_void_p, c_void_p, c_double, c_double, c_int32, c_int32, c_double, c_double,
                               c_void_p, c_void_p, c_void_p]
    lib.bn_em_step_statistics.restype = c_int
    lib.bn_em_step_statistics.argtypes = [c_void_p, c_void_p, c_void_p, c_double, c_double, c_int32, c_int32, c_void_p,
                                          c_void_p, c_void_p]
    lib.bn_plan_enable_step_graphs.restype = c_int
    lib.bn_plan_enable_step_graphs.argtypes = [c_void_p, c_int32]
    lib.bn_plan_enable_star_step.restype = c_int
    lib.bn_plan_enable_star_step.argtypes = [c_void_p, c_int32]
    lib.bn_plan_enable_evidence_compaction.restype = c_int
    lib.bn_plan_enable_evidence_compaction.argtypes = [c_void_p, c_int32]
    lib.bn_plan_evidence_compactions.restype = c_int64
    lib.bn_plan_evidence_compactions.argtypes = [c_void_p]
    lib.bn_plan_star_step_launches.restype = c_int64
    lib.bn_plan_star_step_launches.argtypes = [c_void_p]
    lib.bn_plan_star_step_failures.restype = c_int64
    lib.bn_plan_star_step_failures.argtypes = [c_void_p]
    lib.bn_star_barrier_verdict.restype = c_int32
    lib.bn_star_barrier_verdict.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
    lib.bn_cardinality_served.restype = c_int32
    lib.bn_cardinality_served.argtypes = [c_int32]
    lib.bn_two_parent_shape_served.restype = c_int32
    lib.bn_two_parent_shape_served.argtypes = [c_int32, c_int32, c_int32]
    lib.bn_cardinality_pack_directory.restype = ctypes.c_char_p
    lib.bn_cardinality_pack_directory.argtypes = []
    lib.bn_plan_step_graph_launches.restype = c_int64
    lib.bn_plan_step_graph_launches.argtypes = [c_void_p]
    lib.bn_set_cpts.restype = c_int
    lib.bn_set_cpts.argtypes = [c_void_p, c_void_p, c_void_p]
    lib.bn_reset_messages.restype = c_int
    lib.bn_reset_messages.argtypes = [c_void_p, c_void_p]
    lib.bn_enter_evidence.restype = c_int
    lib.bn_enter_evidence.argtypes = [c_void_p, POINTER(c_void_p), c_void_p]
    lib.bn_enter_evidence_states.restype = c_int
    lib.bn_enter_evidence_states.argtypes = [c_void_p, c_void_p, c_double, c_double, c_void_p]
    lib.bn_iterate.restype = c_int
    lib.bn_iterate.argtypes = [c_void_p, c_int32, c_void_p]
    lib.bn_log_likelihood.restype = c_int
    lib.bn_log_likelihood.argtypes = [c_void_p, c_int32, c_void_p, c_void_p]
    lib.bn_local_log_likelihoods.restype = c_int
    lib.bn_local_log_likelihoods.argtypes = [c_void_p, c_int32, c_void_p, c_void_p]
    lib.bn_marginal_single.restype = c_int
    lib.bn_marginal_single.argtypes = [c_void_p, c_int32, c_void_p, c_void_p]
    lib.bn_marginal_family.restype = c_int
    lib.bn_marginal_family.argtypes = [c_void_p, c_int32, c_void_p, c_void_p]
    lib.bn_em_accumulate.restype = c_int
    lib.bn_em_accumulate.argtypes = [c_void_p, c_void_p, c_void_p]
    lib.bn_em_statistics.restype = c_int
    lib.bn_em_statistics.argtypes = [c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p]
    lib.bn_em_statistics_static.restype = c_int
    lib.bn_em_statistics_static.argtypes = [c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p]
    lib.bn_em_statistics_from.restype = c_int
    lib.bn_em_statistics_from.argtypes = [c_void_p, POINTER(c_void_p), c_int32, c_int32, c_void_p, c_void_p, c_void_p]
    lib.bn_em_statistics_from_states.restype = c_int
    lib.bn_em_statistics_from_states.argtypes = [c_void_p, c_void_p, c_double, c_double, c_int32, c_int32, c_void_p,
                                                 c_void_p, c_void_p]
    lib.bn_m_step.restype = c_int
    lib.bn_m_step.argtypes = [c_void_p, c_void_p, c_void_p, c_double, c_double, c_void_p, c_void_p]
    lib.bn_normaliser_change.restype = c_int
    lib.bn_normaliser_change.argtypes = [c_void_p, c_void_p, c_void_p]
    lib.bn_get_message.restype = c_int
    lib.bn_get_message.argtypes = [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p]
    lib.bn_exact_create.restype = c_int
    lib.bn_exact_create.argtypes = [c_int, c_int32, i32p, i32p, i32p, i32p, c_int32, POINTER(c_void_p)]
    lib.bn_exact_destroy.restype = c_int
    lib.bn_exact_destroy.argtypes = [c_void_p]
    lib.bn_exact_set_cpts.restype = c_int
    lib.bn_exact_set_cpts.argtypes = [c_void_p, c_void_p, c_void_p]
    lib.bn_exact_enter_evidence.restype = c_int
    lib.bn_exact_enter_evidence.argtypes = [c_void_p, c_int64, POINTER(c_void_p), c_void_p]
    lib.bn_exact_log_likelihood.restype = c_int
    lib.bn_exact_log_likelihood.argtypes = [c_void_p, c_int32, c_void_p, c_void_p, c_void_p]
    lib.bn_exact_marginal.restype = c_int
    lib.bn_exact_marginal.argtypes = [c_void_p, i32p, c_int32, c_void_p, c_void_p]
    lib.bn_sampler_create.restype = c_int
    lib.bn_sampler_create.argtypes = [c_int, c_int32, i32p, i32p, i32p, POINTER(c_void_p)]
    lib.bn_sampler_destroy.restype = c_int
    lib.bn_sampler_destroy.argtypes = [c_void_p]
    lib.bn_sampler_cpt_elements.restype = c_int64
    lib.bn_sampler_cpt_elements.argtypes = [c_void_p]
    lib.bn_sampler_workspace_bytes.restype = c_int64
    lib.bn_sampler_workspace_bytes.argtypes = [c_void_p, i32p, c_int32, c_int64]
    lib.bn_sampler_walk_order.restype = c_int32
    lib.bn_sampler_walk_order.argtypes = [c_int32, i32p, i32p, i32p, c_int32, i32p]
    lib.bn_sampler_sample.restype = c_int
    lib.bn_sampler_sample.argtypes = [c_void_p, c_void_p, i32p, c_int32, c_int64, c_uint64, c_void_p, c_void_p, c_void_p]
    lib.bn_profile_enable.restype = c_int
    lib.bn_profile_enable.argtypes = [c_int32]
    lib.bn_profile_only.restype = c_int
    lib.bn_profile_only.argtypes = [c_char_p]
    lib.bn_profile_stride.restype = c_int
    lib.bn_profile_stride.argtypes = [c_int32]
    lib.bn_profile_report.restype = c_int
    lib.bn_profile_report.argtypes = [c_char_p, c_int64]


EXPORTS = (
    "bn_abi_version", "bn_last_error", "bn_plan_create", "bn_plan_destroy", "bn_plan_topology",
    "bn_plan_cpt_elements", "bn_plan_device_bytes", "bn_plan_unmapped_bytes", "bn_plan_map_message_arrays", "bn_plan_enable_step_graphs",
    "bn_plan_step_graph_launches", "bn_plan_enable_star_step", "bn_plan_enable_evidence_compaction", "bn_plan_evidence_compactions", "bn_plan_star_step_launches", "bn_plan_star_step_failures", "bn_star_barrier_verdict", "bn_cardinality_served", "bn_two_parent_shape_served", "bn_cardinality_pack_directory", "bn_em_step", "bn_em_step_statistics", "bn_set_cpts", "bn_reset_messages",
    "bn_enter_evidence", "bn_enter_evidence_states", "bn_iterate", "bn_log_likelihood", "bn_local_log_likelihoods",
    "bn_marginal_single", "bn_marginal_family", "bn_em_accumulate", "bn_em_statistics", "bn_em_statistics_static", "bn_em_statistics_from", "bn_em_statistics_from_states", "bn_m_step", "bn_normaliser_change", "bn_get_message",
    "bn_exact_create", "bn_exact_destroy", "bn_exact_set_cpts", "bn_exact_enter_evidence",
    "bn_exact_log_likelihood", "bn_exact_marginal",
    "bn_sampler_create", "bn_sampler_destroy", "bn_sampler_cpt_elements", "bn_sampler_workspace_bytes",
    "bn_sampler_walk_order", "bn_sampler_sample",
    "bn_profile_enable", "bn_profile_report", "bn_profile_only", "bn_profile_stride",
)


def load():
    """The loaded library (cached).  Raises ImportError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise ImportError(
                f"{_LIB_PATH} is missing: build it with `make` (or __graft_entry__.build()); "
                "there is no fallback path"
            )
        lib = ctypes.CDLL(_LIB_PATH)
        _declare(lib)
        if lib.bn_abi_version() != ABI_VERSION:
            raise ImportError(f"libbn_hip.so ABI {lib.bn_abi_version()} != expected {ABI_VERSION}; rebuild it")
        _lib = lib
    return _lib


def build_cardinality_pack(states: int) -> str:
    """Builds the cardinality pack for wide parents of ``states`` states (``make pack D0=states`` in
    the source tree: about a minute of hipcc, no GPU needed) unless it is there already, and returns
    its path.  Set-up code calls this before the first plan with such a node is built; plans built
    earlier keep the route they have.  The library itself never compiles anything."""
    import subprocess

    path = os.path.join(os.path.dirname(_LIB_PATH), f"libbn_hip_d{int(states)}.so")
    if not os.path.exists(path):
        root = os.path.dirname(os.path.dirname(os.path.abspath(_LIB_PATH)))
        subprocess.run(["make", "-C", root, "pack", f"D0={int(states)}"], check=True, capture_output=True)
    return path


def check(rc: int):
    if rc != 0:
        raise BnError(load().bn_last_error().decode("utf-8", "replace"))


def i32_array(values):
    arr = (c_int32 * max(len(values), 1))(*values)
    return arr
