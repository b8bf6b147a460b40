"""CPU time of the superpoint-graph construction after the triangulation: the oracle restatement (numpy, the reference's own
Python loops over components and superedges) and, when /root/reference is present, the imported reference function itself.
    python oracle/devtools/spgraph_cpu.py [n_points] [n_components]"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from scipy.spatial import Delaunay
from oracle import spg_partition_oracle as P

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
n_com = int(sys.argv[2]) if len(sys.argv) > 2 else 300
rng = np.random.default_rng(0)
centers = rng.uniform(-30, 30, (n_com, 3))
which = rng.integers(0, n_com, n)
xyz = (centers[which] + rng.normal(size=(n, 3)) * 1.5).astype(np.float32)
_, comp = np.unique(which, return_inverse=True)
components = [np.flatnonzero(comp == c) for c in range(comp.max() + 1)]
lab = rng.integers(0, 13, n)
t0 = time.perf_counter(); tets = Delaunay(xyz).simplices; t_del = time.perf_counter() - t0
t0 = time.perf_counter(); g = P.sp_graph_after_triangulation(xyz, 3.0, comp, components, lab, 13, tets); t_or = time.perf_counter() - t0
print(f'{n} points, {len(components)} components, {len(tets)} tetrahedra, {len(g["source"])} superedges: Delaunay {t_del:.2f} s, '
      f'oracle after the triangulation {t_or:.2f} s = {n / t_or / 1e6:.3f} M points/s')
REF = os.environ.get('SPG_REFERENCE', '/root/reference')
if os.path.isdir(REF):
    import scipy.spatial
    sys.path.insert(0, os.path.join(REF, 'partition'))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        import graphs

    class _D(scipy.spatial.Delaunay):
        @property
        def vertices(self):
            return self.simplices
    graphs.Delaunay = _D
    t0 = time.perf_counter(); graphs.compute_sp_graph(xyz, 3.0, comp, components, lab, 13); t_ref = time.perf_counter() - t0
    print(f'imported reference compute_sp_graph (incl. its own Delaunay): {t_ref:.2f} s -> after the triangulation {t_ref - t_del:.2f} s '
          f'= {n / max(t_ref - t_del, 1e-9) / 1e6:.3f} M points/s')
