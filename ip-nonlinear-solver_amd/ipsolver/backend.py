"""Backend selection for the outer loops.

The product has exactly one backend, ``backend_hip`` (GPU only -- importing it
without libipx.so or using it without a HIP device fails loudly; there is no
CPU fallback).  ``use(module)`` lets the test-suite inject the CPU oracle's
backend to check the host logic on a machine without a GPU.
"""
_active = None


def get():
    global _active
    if _active is None:
        from . import backend_hip
        _active = backend_hip
    return _active


class use:
    """Context manager: ``with backend.use(oracle.numpy_backend): ...``"""

    def __init__(self, module):
        self.module = module

    def __enter__(self):
        global _active
        self.prev, _active = _active, self.module
        return self.module

    def __exit__(self, *exc):
        global _active
        _active = self.prev
