"""Host-side helpers mirroring gnngls/__init__.py (reference lines cited per function).
Pure Python on the caller's networkx graph -- these are input/output plumbing, not the hot path."""


def tour_to_edge_attribute(G, tour):
    """gnngls/__init__.py:9-14"""
    in_tour = {}
    tour_edges = set(zip(tour[:-1], tour[1:]))
    for e in G.edges:
        in_tour[e] = e in tour_edges or tuple(reversed(e)) in tour_edges
    return in_tour


def tour_cost(G, tour, weight="weight"):
    """gnngls/__init__.py:17-21: c = 0; c += w(e) left to right (fp64, order matters)."""
    c = 0
    for e in zip(tour[:-1], tour[1:]):
        c += G.edges[e][weight]
    return c


def is_equivalent_tour(tour_a, tour_b):
    """gnngls/__init__.py:24-29"""
    return tour_a == tour_b[::-1] or tour_a == tour_b


def is_valid_tour(G, tour):
    """gnngls/__init__.py:32-44"""
    if tour[0] != 0 or tour[-1] != 0:
        return False
    for n in G.nodes:
        c = tour.count(n)
        if n == 0:
            if c != 2:
                return False
        elif c != 1:
            return False
    return True


def optimal_cost(G, weight="weight"):
    """gnngls/__init__.py:55-60"""
    c = 0
    for e in G.edges:
        if G.edges[e]["in_solution"]:
            c += G.edges[e][weight]
    return c
