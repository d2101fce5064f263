"""qiddm_amd -- MI355X-native engine for the QIDDM quantum-layer denoise hot path.

    from qiddm_amd import nn, models, noise, qml

``nn`` mirrors the reference's ``nn`` namespace (qdense / qconv / unet classes),
``models.Diffusion`` and ``noise.add_normal_noise_multiple`` mirror ``src/models.py`` and
``src/noise.py``; ``qml`` is the slice of the PennyLane front-end those layers use, bound
to hand-written HIP statevector kernels for gfx950 through the C ABI of
``include/qiddm_hip.h``.  There is no CPU execution path.
"""
from . import circuit, models, nn, noise, qml  # noqa: F401
from .circuit import Circuit, get_default_precision, set_default_precision  # noqa: F401

__version__ = "0.1.0"
