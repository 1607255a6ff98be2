"""CPU oracle for the colvarsfinder training hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch (CPU, fp32 or fp64) restatement of the reference
algorithm for the per-batch training step of ``EigenFunctionTask`` /
``AutoEncoderTask`` (reference ``colvarsfinder/core.py``) and of the third-party
``molann`` alignment + feature layer the reference's dipeptide example plugs in
as ``pp_layer`` (``examples/dipeptide/main.ipynb:31-32,333-348``).

Only ``tests/``, ``__graft_entry__.smoke()``, ``tools/gen_golden.py`` and the
``cpu_baseline`` leg of ``bench.py`` may import it - and only as the checker /
the timed CPU baseline.  The shipped package (``colvars-finder_amd/colvarsfinder``)
never imports anything from here; its compute goes through the HIP extension and
fails loudly when that is missing.

Pinning status
--------------
* ``losses.py`` / ``train.py`` / ``nnref.py`` (everything that restates code under
  ``/root/reference``): pinned against outputs of the reference itself, imported in
  the build container by ``tools/gen_golden.py`` and committed as fixtures under
  ``tests/golden/`` (``tests/test_oracle_golden.py``).
* ``pp.py`` (Kabsch alignment + position/bond/angle/dihedral features): the
  arithmetic lives in ``molann`` (PyPI ``molann``, GitHub zwpku/molann), which is
  neither vendored in ``/root/reference`` nor pinned to a version anywhere in it
  (absent from ``setup.cfg:21-25`` and ``docs/environment.yml``) nor installed here.
  The reference holds no test or golden vector at that boundary, so for this file
  alone: **parity unpinned**.  It restates the published algorithm (Kabsch 1976/78
  with the det-sign fix, row-vector convention ``x_aligned = (x - c) @ R``) and is
  pinned by invariance / known-answer / finite-difference tests of our own, plus
  the one piece of evidence the reference prints (``main.ipynb:304-327``: the
  reference coordinates are the align-group positions minus their unweighted
  centroid).
"""
