"""oracle/ref_import.py -- TEST INFRASTRUCTURE (golden-vector generation only).

Imports the *reference's own Python* from /root/reference so that golden vectors can be
captured from it in the build container.  /root/reference does not exist on the GPU box, so
nothing under tests/, bench.py or __graft_entry__.py imports this module at run time; only
oracle/gen_golden.py does (the committed fixtures under tests/golden/ are its output).

Two obstacles, both off the hot path, are worked around without touching the reference:

* gnngls/__init__.py:1-5 imports `concorde.tsp`, `lkh`, `tsplib95` at top level (used only
  by optimal_tour / fixed_edge_tour, __init__.py:47-52,63-75).  Empty stub modules are put
  in sys.modules.
* gnngls/models.py:1 and gnngls/datasets.py:5 import `dgl` (pinned dgl-cu111==0.6.1,
  Pipfile.lock:316-328, not installed, no network).  A shim `dgl` module is injected whose
  `nn.GATConv` is the oracle's restatement (oracle/model_oracle.py) and whose
  `nn.utils.Sequential` is torch's nn.Sequential (models.py:59 uses it only as a container;
  the forward iterates it manually at models.py:67-68).  The *wiring* of models.py therefore
  runs verbatim; only GATConv's arithmetic is restated.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("GNNGLS_REFERENCE", "/root/reference")


def reference_available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "gnngls", "operators.py"))


def _stub(name):
    m = types.ModuleType(name)
    sys.modules[name] = m
    return m


def import_reference(with_models=False):
    """Returns the imported reference package `gnngls` (algorithms, operators always;
    models when with_models=True, through the dgl shim)."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    for name in ("concorde", "concorde.tsp", "lkh", "tsplib95"):
        if name not in sys.modules:
            _stub(name)
    sys.modules["concorde"].tsp = sys.modules["concorde.tsp"]

    if with_models and "dgl" not in sys.modules:
        import torch.nn as nn
        from . import model_oracle

        dgl = _stub("dgl")
        dgl_nn = _stub("dgl.nn")
        dgl_nn_utils = _stub("dgl.nn.utils")
        dgl.nn = dgl_nn
        dgl_nn.utils = dgl_nn_utils
        dgl_nn.GATConv = model_oracle.GATConvOracle
        dgl_nn_utils.Sequential = nn.Sequential

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import gnngls  # noqa: E402  (the reference package)
    from gnngls import algorithms, operators  # noqa: F401
    if with_models:
        from gnngls import models  # noqa: F401
    return gnngls
