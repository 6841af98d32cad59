"""omni_avsr_amd — MI355X-native inference hot path for Omni-AVSR.

Python here is host glue only (the reference's boundary is the Python class `AVSR_LLMs`,
Omni_AVSR/modeling_OmniAVSR.py:27); every tensor op runs in libomniavsr_hip.so (csrc/, C ABI in
include/omni_avsr.h).  There is no CPU or PyTorch fallback: importing `omni_avsr_amd.lib` raises if
the shared library is missing.
"""
__version__ = "0.1.0"
