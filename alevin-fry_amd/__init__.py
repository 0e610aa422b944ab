"""alevin-fry_amd — MI355X-native `alevin-fry quant` hot path.

The directory name carries a hyphen (it mirrors the reference's name), so import it
with `importlib.import_module("alevin-fry_amd")`.  Contents:
  csrc/       hand-written HIP kernels for gfx950 + the C ABI of include/afquant.h
  afquant.py  host-side mirror of the reference's per-cell quant interface (ctypes)
  rad.py      collated-RAD chunk codec (host)
  synth.py    seeded synthetic inputs
"""
from . import _abi, rad, synth  # noqa: F401
from .afquant import AfqError, Quantifier, QuantResult, WorkerConfig, load_library  # noqa: F401
