"""MI355X-native scan-to-submap registration hot path (package root).

Holds only what the path needs: ``csrc/`` (hand-written HIP kernels + the C ABI
of ``include/pcm_amd.h``, built into ``libpcm_amd.so``), the host-side mirror
of the reference's registration interface (``registration.py``), the synthetic
Livox-shaped input generator (``synth.py``) and the batch sharding helper
(``sharding.py``).  There is no CPU fallback: loading fails loudly when the HIP
library is missing, and contexts fail when no HIP device is present.
"""
from .capi import PcmError, build_library, library_path, load_library  # noqa: F401
from . import sharding  # noqa: F401
from .registration import (GicpRegistration, NdtRegistration, P2PlaneRegistration, PclNdtRegistration, Registration, VgicpCudaRegistration, VgicpRegistration,  # noqa: F401
                           RegistrationResult, align_batch)
