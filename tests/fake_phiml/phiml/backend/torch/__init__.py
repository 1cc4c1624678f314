""" TEST DOUBLE of phiml.backend.torch: the TORCH singleton whose type a 'hip' backend subclasses """
from .. import BACKENDS, Backend


class TorchBackend(Backend):
    def __init__(self):
        super().__init__('torch')


TORCH = TorchBackend()
BACKENDS.append(TORCH)
