"""gnngls_amd -- MI355X-native hot path of proroklab/gnngls (edge-regret GNN forward + guided local
search), hand-written HIP kernels behind the reference's Python call surface.

Host helpers that the reference keeps in gnngls/__init__.py are mirrored in gnngls_amd.host.
"""
from .host import (is_equivalent_tour, is_valid_tour, optimal_cost, tour_cost,  # noqa: F401
                   tour_to_edge_attribute)

__all__ = ["tour_cost", "optimal_cost", "is_valid_tour", "is_equivalent_tour", "tour_to_edge_attribute"]
