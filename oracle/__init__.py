"""
CPU oracle for the block-tridiagonal Gauss-Markov hot path.

TEST INFRASTRUCTURE ONLY.  This package is a NumPy (and, in ``oracle/csrc``, plain C)
restatement of the reference algorithm (AaltoML/vi-diffusion-processes, a Markovflow fork).
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and there only as the checker -- never as the product.  The product path
(``vi-diffusion-processes_amd``) never imports this package and fails loudly when the HIP
library is missing.

Pinning: the restatement is pinned by golden vectors generated from the reference's own
TF-free test tools (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``) and by the dense
``numpy.linalg`` identities the reference's unit tests use (KA1..KA11 in SURVEY.md section 8c).
Parts the reference itself never tests (CVI-DP on a non-linear drift, VDP, any d>1 SDE term)
are marked "parity unpinned" where they are defined.
"""
