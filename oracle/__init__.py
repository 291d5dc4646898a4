"""TEST INFRASTRUCTURE ONLY.

CPU restatement (torch float64 + autograd) of the rasterizer hot path of
yunjinli/TRASE.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this package, and only as the
checker.  Nothing under ``trase_amd/`` or the drop-in shims imports it.

PARITY UNPINNED for the rasterizer core: the reference's CUDA source
(``submodules/diff-gaussian-rasterization``) is an empty, un-vendored submodule
and the reference ships no tests or golden vectors.  The
sub-steps the reference *does* implement in Python (SH evaluation, covariance
construction, camera matrices, the deformation MLP, the losses) are pinned by
fixtures generated from the imported reference (tests/golden/, see
tests/golden/make_golden.py).
"""
