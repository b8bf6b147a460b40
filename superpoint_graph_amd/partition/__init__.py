"""Device-side pieces of the reference's `partition/` preprocessing that feed the learning hot path (SURVEY.md section 8, row
f4 tail): `graphs.compute_sp_graph` after the triangulation and ply_c's `compute_geof`."""
