"""tal_asrd_amd -- MI355X-native acoustic hot path of calclavia/tal-asrd.

log-mel front-end -> TDS encoder -> diarization / decoder heads as hand-written
HIP kernels (gfx950) behind the reference's own module API.  See DESIGN.md.
"""
from . import synth  # noqa: F401
from ._native import NativeError, LIB_PATH  # noqa: F401
from .modules import PositionalEncoding, weight_init  # noqa: F401
from .models import (ASRModel, LogMelSpec, ModRZTXDecoderLayer, SDModel, TDS, TDSBlock,  # noqa: F401
                     padding_mask)

__version__ = "0.1.0"
