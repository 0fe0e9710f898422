"""reze-engine_amd — MI355X-native per-frame PMX morph + skin path behind the Reze Engine API.

Layout (only what the hot path needs):
  csrc/   hand-written HIP kernels (gfx950) + the C ABI (include/reze_deform.h) + raw N-API shim
  host/   JavaScript host side mirroring the reference's Engine / Model / loaders (Node)
  capi.py ctypes binding of the C ABI (bench.py, tests)
  synth.py deterministic synthetic PMX-shaped workloads
The directory name carries a hyphen (it is the name the build contract fixes); the importable
Python package name is `reze_engine_amd`, provided by the shim module at the repo root.
"""
from . import capi, shard, synth  # noqa: F401
from .capi import DeformContext, RzError, device_count, shard_range  # noqa: F401

__all__ = ["capi", "shard", "synth", "DeformContext", "RzError", "device_count", "shard_range"]
