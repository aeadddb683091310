"""zkvm_amd — MI355X-native PLONK prover core (Fr NTT, KZG10 G1 MSM, quotient evaluation) behind a C boundary.

`libzkgpu.so` (hand-written HIP, gfx950) does all the work; this package is the thin host-side mirror of the
reference's interface for that path.
"""
from ._lib import LIB_PATH, ZkGpuError, build  # noqa: F401
from .api import (COSET_FFT, COSET_IFFT, FFT, IFFT, CommitKey, Context, EvaluationDomain,  # noqa: F401
                  InvalidEvalDomainSize, PolynomialDegreeIsZero, PolynomialDegreeTooLarge, g1_compress,
                  g1_sum_to_affine)
