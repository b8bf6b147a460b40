"""Drop-in for the two `libply_c` functions of the reference's partition pipeline that run on the GPU here
(partition/partition.py:124-150: `libply_c.prune`, `libply_c.compute_geof`); see graphs.py."""
from .graphs import compute_geof, prune  # noqa: F401
