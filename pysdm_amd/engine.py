"""`Engine`: arrays + calls for one implementation of include/sdm_hip.h.

The host layer of this package (population, runners, PySDM-shaped backend) is written against this
small interface only: allocate / upload / download arrays where the library expects them and call
header symbols by name.  `HipEngine` is the product: torch CUDA tensors (torch is used for device
memory and streams, nothing else) and libsdm_hip.so.  The test suite supplies a second engine over
the CPU oracle (oracle/engine.py, numpy arrays); nothing in this package refers to it.
"""
import ctypes

import numpy as np

from . import abi

FLOAT, INT, BOOL = np.float64, np.int64, np.bool_


class Engine:
    """interface; see HipEngine"""

    name = "abstract"
    library = None
    handle = None

    # ---- arrays -----------------------------------------------------------------------------
    def empty(self, shape, dtype):
        raise NotImplementedError

    def upload(self, array):
        """a new library-side array holding a copy of the numpy array"""
        raise NotImplementedError

    def download(self, array):
        """a numpy copy"""
        raise NotImplementedError

    def zeros(self, shape, dtype):
        out = self.empty(shape, dtype)
        self.fill(out, 0)
        return out

    def full(self, shape, dtype, value):
        out = self.empty(shape, dtype)
        self.fill(out, value)
        return out

    @staticmethod
    def fill(array, value):
        array[...] = value

    @staticmethod
    def assign(dst, src):
        dst[...] = src

    @staticmethod
    def size(array):
        return int(np.prod(tuple(array.shape), dtype=np.int64))

    # ---- calls ------------------------------------------------------------------------------
    def call(self, symbol, *args):
        self._before_call()
        self.library.invoke(symbol, self.handle, args)

    def _before_call(self):
        pass

    def synchronize(self):
        self.call("sdm_ctx_synchronize")

    def scalar_out(self, symbol, ctype, *args):
        """for symbols whose last parameter is a host out-pointer: returns its value"""
        out = ctype()
        self.call(symbol, *args, out)
        return out.value


class HipEngine(Engine):
    """one sdm_ctx per process and device; follows torch's current stream"""

    name = "hip"
    _instances = {}

    def __init__(self, device_index):
        import torch  # pylint: disable=import-outside-toplevel

        self.torch = torch
        self.library = abi.hip_library()
        self.handle = abi.c_ptr()
        self.library.check(self.library.cdll.sdm_ctx_create(ctypes.byref(self.handle),
                                                            abi.c_int(device_index)))
        self.device = torch.device("cuda", device_index)
        self._stream = None
        self._dtype = {FLOAT: torch.float64, INT: torch.int64, BOOL: torch.bool,
                       np.uint8: torch.uint8}

    @classmethod
    def get(cls, device_index=None):
        import torch  # pylint: disable=import-outside-toplevel

        if not torch.cuda.is_available():
            raise RuntimeError("pysdm_amd needs a GPU (torch.cuda.is_available() is False); "
                               "there is no CPU fallback")
        if device_index is None:
            device_index = torch.cuda.current_device()
        if device_index not in cls._instances:
            cls._instances[device_index] = cls(device_index)
        return cls._instances[device_index]

    def _before_call(self):
        stream = self.torch.cuda.current_stream(self.device).cuda_stream
        if stream != self._stream:
            self.library.invoke("sdm_ctx_set_stream", self.handle, (stream,))
            self._stream = stream

    def empty(self, shape, dtype):
        return self.torch.empty(shape, dtype=self._dtype[np.dtype(dtype).type],
                                device=self.device)

    def upload(self, array):
        # (a writable, contiguous host copy: torch refuses to wrap read-only arrays quietly - the
        # arrays of an .npz file are read-only)
        return self.torch.from_numpy(np.require(array, requirements=["C", "W"])).to(self.device)

    @staticmethod
    def download(array):
        return array.detach().cpu().numpy()
