"""flydog_sdr_gps_amd -- MI355X (gfx950) implementation of the FlyDog_SDR_GPS DSP hot path.

The product is the C-ABI library ``libkiwigpu.so`` (include/kiwigpu.h, sources in
``csrc/``).  This package is the thin host side above it: a ctypes binding and a
mirror of the reference's entry points for the path (gps/search.cpp
SearchInit / Sample / Correlate).  There is no CPU fallback: without the built
library and a gfx950 device, constructing a context raises.
"""
from ._lib import KiwiGpuError, Context, load_library, library_path  # noqa: F401
from .acq import Searcher, AcqResult  # noqa: F401
from .wf import Waterfall, WfParams  # noqa: F401
from .ddc import Ddc, RxDdc  # noqa: F401
from .snd import FastFir  # noqa: F401
from .post import Post  # noqa: F401
from .wire import Adpcm  # noqa: F401
from .handoff import Aperture, chan_start  # noqa: F401
from . import sats, prn, synth, shard, wf, snd, post, wire, handoff  # noqa: F401

__all__ = ["KiwiGpuError", "Context", "Searcher", "AcqResult", "Waterfall", "WfParams", "Ddc", "RxDdc", "FastFir", "Post", "Adpcm", "Aperture", "chan_start",
           "load_library", "library_path", "sats", "prn", "synth", "shard", "wf"]
