"""Dense Jacobian path (BASELINE config 2): device matrix + Gram/Cholesky
normal-equation solver.  Kernels: csrc/dense.hip."""
import numpy as np
import torch

from . import _hip
from . import device as dv
from .device import DVec, _p, stream_ptr, ctx


class DeviceDense:
    def __init__(self, t):
        raise NotImplementedError("dense Jacobian path: kernels not built yet")

    @staticmethod
    def from_host(a):
        return DeviceDense(None)


class DenseNormalSolver:
    def __init__(self, A):
        raise NotImplementedError("dense Jacobian path: kernels not built yet")
