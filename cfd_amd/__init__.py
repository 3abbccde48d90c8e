"""cfd_amd -- MI355X-native SIMPLE/PISO outer step behind the reference's solver API.

    from cfd_amd import solver, mesh
    fields, step, time = solver.run_case(CONFIG)

The numerical work is done by hand-written HIP kernels (cfd_amd/csrc, libcfd_hip.so,
C ABI in include/cfd_hip.h).  Importing the package does not need a GPU; any compute
call does, and fails loudly without one.
"""
from . import mesh  # noqa: F401

__all__ = ["mesh", "solver", "bc", "device", "numba_kernels", "post_processing"]
__version__ = "0.1.0"
