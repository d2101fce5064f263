"""Oracle restatement of ``sklearn.decomposition.PCA(n).fit_transform``.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

The PCA nets of the reference re-fit on every forward call
(``self.pca.fit_transform(batch)``, ``nn/qdense.py:456, 544, 725, 1429``; finding F4 of
SURVEY.md).  For the batches the shipped trajectories were produced on (10 images of 784
pixels, 8 components) scikit-learn's ``svd_solver="auto"`` picks the exact LAPACK path
("full": ``8 >= 0.8 * min(10, 784)``), so the restatement is: centre, thin SVD, sign
convention, scores ``U * S``.

**Pinned** (``tests/test_oracle_reference_runs.py``): only the V-based sign convention
(largest-magnitude entry of every *component* positive -- scikit-learn >= 1.5,
``svd_flip(u_based_decision=False)``) reproduces the PNG trajectories the reference ships
in ``results_rebuttal_complex_dataset/*.zip``; the U-based one of the 1.1.3 pinned in
``requirements.txt:76`` misses them by up to 109/255, so the authors' runs used the newer
rule and that is what the oracle implements.
"""
from __future__ import annotations

import numpy as np


def pca_fit_transform(x, n_components: int, u_based_decision: bool = False) -> np.ndarray:
    x = np.asarray(x, dtype=np.float64)
    xc = x - x.mean(axis=0)
    u, s, vt = np.linalg.svd(xc, full_matrices=False)
    if u_based_decision:
        idx = np.argmax(np.abs(u), axis=0)
        signs = np.sign(u[idx, np.arange(u.shape[1])])
    else:
        idx = np.argmax(np.abs(vt), axis=1)
        signs = np.sign(vt[np.arange(vt.shape[0]), idx])
    signs = np.where(signs == 0, 1.0, signs)
    return (u * signs * s)[:, :n_components]
