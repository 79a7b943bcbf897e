"""libreasr_amd: MI355X-native streaming RNN-Transducer inference path behind the LibreASR API.

Only the hot path lives here: csrc/ (hand-written gfx950 HIP + the C ABI of
include/lasr.h), engine.py (ctypes owner of one lasr_ctx) and lib/ (the host-side mirror of the
reference's Python inference surface)."""
__version__ = "0.1.0"
