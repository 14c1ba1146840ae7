"""channelcoding_amd -- MI355X-native BCH / Reed-Solomon decoding (min-sum belief
propagation and the algebraic syndrome / Berlekamp-Massey / Chien / error-value
chain) behind the API surface of hannesweisbach/channelcoding.

The compute lives in libchannelcoding_amd.so (hand-written HIP for gfx950,
include/channelcoding_amd.h).  This package is the host-side mirror of the
reference's classes plus the torch.distributed Monte-Carlo harness.
"""
from . import _capi as capi
from ._capi import CcError
from .codes import (berlekamp_massey_tag, cyclic, decoding_failure, dmin, errors, euklid_tag, min_sum, min_sum_decoder, min_sum_tag,
                    normalized_2d_min_sum_tag, normalized_min_sum_tag, offset_min_sum_tag,
                    peterson_gorenstein_zierler_tag, primitive_bch, rs, self_correcting_1_min_sum_tag,
                    self_correcting_2_min_sum_tag)

__all__ = [
    "capi", "CcError", "cyclic", "primitive_bch", "rs", "errors", "dmin", "decoding_failure",
    "peterson_gorenstein_zierler_tag", "berlekamp_massey_tag", "euklid_tag", "min_sum_tag",
    "normalized_min_sum_tag", "offset_min_sum_tag", "self_correcting_1_min_sum_tag",
    "self_correcting_2_min_sum_tag", "normalized_2d_min_sum_tag", "min_sum", "min_sum_decoder",
]
