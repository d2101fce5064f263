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
* ``oracle.statevector`` / ``oracle.circuits`` / ``oracle.pca``, the **RZ-encoding / Rot / SEL-range /
  CZ-ring / <Z_i> family** (rows A1, A2 of SURVEY.md section 8a; the shared ``Rot``, wire-order and range code
  also serves A3): **pinned at 8-bit resolution to outputs of the reference itself.**  The reference ships
  five trained ``QIDDM_PL_noise(784, 8, 6, 2)`` checkpoints together with the PNG trajectories its authors'
  PennyLane-Lightning run wrote for them (``results_rebuttal_complex_dataset/{medmnist,logo2kplus}.zip``,
  ``src/bloodmnist.py:231-277, 374-411``).  ``tests/golden/make_reference_runs.py`` extracts them as data
  (``tests/golden/reference_runs/``); ``tests/test_oracle_reference_runs.py`` reproduces all
  5 folders x 10 images x 6 steps within one grey level (identical level on > 99.99 % of the pixels) and shows
  that each single convention slip (RZ sign, phi<->omega, SEL ranges, wire order of the encoding or of the
  measurement, PCA sign rule) misses by tens of levels; ``tests/test_gpu_reference_runs.py`` does the same for the
  HIP path through the harness.
* The **AmplitudeEmbedding / SEL-CNOT / probs / ``qw_map.tanh`` family** (rows A4, A5): **parity unpinned** --
  definition + known-answer tests only.  The reference holds no outputs for it: the
  ``QDenseUndirected_old60_*.pt`` files in the same zips are not the model behind the PNGs (the PL model ran last
  and overwrote the images; one of them is a mis-named classical UNet), and ``QConv2d.forward`` never calls its
  circuit as checked in (finding F3).  PennyLane 0.29.0 / PennyLane-Lightning 0.30.0 / qW-Map 0.1.2
  (``requirements.txt:44-45,66``) are neither vendored under ``/root/reference`` nor installable offline.  This
  half follows PennyLane's published operator definitions, shares the pinned ``Rot`` / wire-order / range code,
  and is cross-checked by two independent implementations (gate-by-gate strided update vs. dense Kronecker
  unitaries) plus the known-answer identities KA1-KA12 of SURVEY.md section 8c.
* ``oracle.density`` (Kraus channels) and ``oracle.metrics`` (SSIM/PSNR): **parity unpinned** (definitions only).
"""
