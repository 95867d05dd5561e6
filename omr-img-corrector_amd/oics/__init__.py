"""`oics` -- host-side mirror of the reference crate's public modules (packages/lib/src/lib.rs:1-12)
over libomrdeskew.so.  Module names follow the crate: calculate, fft, hough, omr, projection, transfer, types.
"""
from . import calculate, fft, hough, omr, projection, synth, transfer, types  # noqa: F401
from ._lib import LIB_PATH, OmrError, lib  # noqa: F401
