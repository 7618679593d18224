"""skred_amd -- MI355X-native render path for skred's per-voice synth loop.

Only what the hot path needs lives here: ``csrc/`` (HIP kernels + the C ABI declared in
``include/``), and thin Python mirrors of that ABI (``bank``, ``device``, ``banks``) used by
tests and bench.py.  The product path has no CPU fallback: ``device`` raises if the HIP
library is missing or no GPU is usable.
"""
__version__ = "0.1.0"
