"""learning/ecc/utils.py -- only the name the SPG path's callers import."""


def get_edge_shards(degs, edge_mem_limit):
    """API shim for reference learning/ecc/utils.py:56-69.  The reference splits the node range into shards of at
    most ~edge_mem_limit edges to bound the size of its per-edge temporaries; the HIP kernels never materialise
    per-edge products, so the whole graph is always ONE shard: [(number of nodes, number of edges)].  (The sharding
    rule itself is restated, and checked against the reference, in oracle/spg_oracle.py.)"""
    return [(len(degs), int(sum(int(d) for d in degs)))]
