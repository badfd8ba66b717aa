"""MI355X-native drop-in for rofl_crypto's ZK norm-bound hot path.

Host-side mirror of the reference interface (same module names, argument meaning and error
behaviour as rofl_crypto::{range_proof_vec, l2_range_proof_vec, pedersen_ops, conversion32}),
bound to the C ABI in include/rofl_zk.h via ctypes.  There is NO CPU fallback: if librofl_zk.so is
missing or no HIP device is present the calls fail loudly.
"""
from . import api  # noqa: F401
from .api import (  # noqa: F401
    RoflError, Nonce, lib, range_proof_vec, l2_range_proof_vec, pedersen_ops, conversion32, rand_proof_vec, square_rand_proof_vec, square_proof_vec, compressed_rand_proof,
    set_device, get_device, last_timing, last_kernel_times, set_timing, bench_femul, set_fp, get_fp, set_option, get_option,
)
from . import params  # noqa: F401,E402
from .params import EncParamsRange, EncParamsRangeCompressed, EncParamsL2, EncParamsL2Compressed, EncModelParamsAccumulator, wire  # noqa: F401,E402
