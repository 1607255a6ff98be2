"""colvarsfinder on MI355X: the training hot path of zwpku/colvars-finder as hand-written
gfx950 HIP kernels behind the reference's ``colvarsfinder.core`` / ``colvarsfinder.nn`` API.

Importing the package does not need a GPU; constructing a task does (and needs
``libcvf_hip.so`` built next to this file: ``python __graft_entry__.py``)."""

__version__ = "0.1.0"
