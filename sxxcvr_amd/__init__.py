"""sxxcvr_amd -- MI355X-native polyphase FIR resampling path behind the
SoapySDR ``driver=sx`` Device/Stream surface of tejeez/sxxcvr.

The product is native: hand-written HIP kernels (gfx950) behind a C ABI
(include/sxfir.h, include/sx_device.h).  This package only loads those
libraries and mirrors the reference's Python-facing call pattern; there is no
CPU implementation and nothing here imports the oracle.
"""
from ._native import NativeError, load_sxfir  # noqa: F401
from .resampler import PipelinedResampler, Resampler, design_lowpass, pin_array, synth_fill, unpin_array  # noqa: F401

__all__ = ["NativeError", "load_sxfir", "Resampler", "PipelinedResampler", "design_lowpass", "synth_fill", "pin_array", "unpin_array"]
