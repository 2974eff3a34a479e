"""Host-side helpers with the names and semantics of the reference's gnngls/__init__.py (lines cited per
function).  Plain Python over the caller's networkx graph: input/output plumbing, not the hot path (the
batched device versions live in gnngls_amd.ops)."""
import functools
import operator


def _tour_edges(tour):
    return zip(tour[:-1], tour[1:])


def tour_to_edge_attribute(G, tour):
    """{edge: is it on the tour (either orientation)} for every edge of G (reference __init__.py:9-14)."""
    on_tour = {frozenset(e) for e in _tour_edges(tour)}
    return {e: frozenset(e) in on_tour for e in G.edges}


def tour_cost(G, tour, weight="weight"):
    """Sum of the edge attribute along the tour, accumulated left to right starting from the integer 0 exactly
    like the reference's `c = 0; c += w` loop (reference __init__.py:17-21) -- the fp64 summation order is part of
    the contract (the device kernels reproduce it bit for bit)."""
    return functools.reduce(operator.add, (G.edges[e][weight] for e in _tour_edges(tour)), 0)


def is_equivalent_tour(tour_a, tour_b):
    """Same cycle in either direction (reference __init__.py:24-29)."""
    return tour_a in (tour_b, tour_b[::-1])


def is_valid_tour(G, tour):
    """Starts and ends at the depot 0, visits the depot exactly twice and every other node exactly once
    (reference __init__.py:32-44)."""
    if tour[0] != 0 or tour[-1] != 0:
        return False
    return all(tour.count(v) == (2 if v == 0 else 1) for v in G.nodes)


def optimal_cost(G, weight="weight"):
    """Total weight of the edges flagged `in_solution`, in edge insertion order (reference __init__.py:55-60)."""
    return functools.reduce(operator.add, (d[weight] for _, _, d in G.edges(data=True) if d["in_solution"]), 0)
