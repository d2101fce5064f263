"""CPU oracle for the QIDDM quantum-layer hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and there only as the checker.  The product path
(``qiddm_amd``) never imports this package and fails loudly when its HIP
extension is missing.

Pinning status (see DESIGN.md "Oracle"):

* ``oracle.diffusion`` (``add_normal_noise_multiple`` + the ``Diffusion``
  training step / sampler): **pinned** against outputs of the reference's own
  ``src/noise.py`` / ``src/models.py`` run in the build container; the vectors
  live in ``tests/golden/diffusion_*.npz`` and were produced by
  ``tests/golden/make_diffusion_golden.py``.
* ``oracle.statevector`` / ``oracle.dense`` / ``oracle.circuits`` (the
  statevector arithmetic of PennyLane 0.29.0 ``default.qubit``, PennyLane-
  Lightning 0.30.0 and qW-Map 0.1.2, which are pinned third-party dependencies
  of the reference -- ``requirements.txt:44-45,66`` -- and are neither vendored
  under ``/root/reference`` nor installable offline): **parity unpinned**.
  The reference holds no tests, golden vectors or recorded circuit outputs for
  this path.  The restatement follows PennyLane's
  published operator definitions and is cross-checked by two independent
  implementations (gate-by-gate strided update vs. dense Kronecker unitaries)
  plus the analytic known-answer identities KA1-KA12 of SURVEY.md section 8c.
"""
