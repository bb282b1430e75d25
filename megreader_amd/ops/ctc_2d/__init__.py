"""Extension-level boundary of the 2D-CTC op: `ops.ctc_2d.ctc_2d_csrc` is the pybind11 module the
reference's OWN `ops/ctc_2d/ctc_loss_2d.py:3,15,30` binds.  `megreader_amd.dropin.install(level="extension")` registers
`ctc_2d_csrc` below under that dotted name, so the reference's Function file runs unchanged on the HIP kernels."""
from . import ctc_2d_csrc  # noqa: F401
