"""Drop-in stand-in for the ONE pytorch3d entry point the reference uses on the path
(``pytorch3d.ops.knn_points``, README.md:34 pins v0.7.6), backed by the HIP spatial-hash KNN."""
