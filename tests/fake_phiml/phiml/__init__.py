""" TEST DOUBLE of the `phiml` package (see tests/fake_phiml/README.md) """
from . import math, backend   # noqa: F401
__version__ = "0.0-test-double"
