"""Backends.  `HIP` (alias `GPU`) is the product: every method runs in libsdm_hip.so on the GPU.
There is deliberately no CPU fallback here (the CPU restatement lives in oracle/, test-only)."""


def __getattr__(name):
    if name in ("HIP", "GPU"):
        from .hip import HIP  # pylint: disable=import-outside-toplevel

        return HIP
    raise AttributeError(name)
