"""learning/ecc/utils.py -- only what the SPG path touches."""
import numpy as np


def get_edge_shards(degs, edge_mem_limit):
    """Reference learning/ecc/utils.py:56-69: splits the node range into shards of at most ~edge_mem_limit
    edges.  The HIP kernels process all edges in one launch (the result does not depend on the
    sharding), so this is kept for API compatibility and for tests of that invariance."""
    d = degs if isinstance(degs, np.ndarray) else degs.cpu().numpy()
    cs = np.cumsum(d)
    cse = cs // edge_mem_limit
    _, cse_i, cse_c = np.unique(cse, return_index=True, return_counts=True)
    shards = []
    for b in range(len(cse_i)):
        numd = cse_c[b]
        nume = (cs[-1] if b == len(cse_i) - 1 else cs[cse_i[b + 1] - 1]) - cs[cse_i[b]] + d[cse_i[b]]
        shards.append((int(numd), int(nume)))
    return shards
