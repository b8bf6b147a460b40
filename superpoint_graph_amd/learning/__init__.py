"""Host-side mirror of the reference's `learning` package for the hot path (same module / class / argument
names as loicland/superpoint_graph `learning/`), backed by the HIP kernels of libspg_hip.so:

    learning.pointnet   STNkD, PointNet, CloudEmbedder         (reference learning/pointnet.py)
    learning.graphnet   GraphNetwork, create_fnet              (reference learning/graphnet.py)
    learning.modules    RNNGraphConvModule, GRUCellEx          (reference learning/modules.py)
    learning.ecc        GraphConvInfo, GraphConvFunction, ...  (reference learning/ecc/)
    learning.spg        eccpc_collate and the loader contract  (reference learning/spg.py)
"""
