"""Drop-in package name the reference imports (scene/gaussian_model.py:28:
``from simple_knn._C import distCUDA2``)."""
