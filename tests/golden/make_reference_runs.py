"""Extract the reference-held QIDDM_PL_noise sampling trajectories (data only).

The reference ships, inside ``results_rebuttal_complex_dataset/medmnist.zip``
and ``logo2kplus.zip``, five label folders each holding

  * ``QIDDM_PL_noise=8_L=6_N=2_<k>.pt`` -- the trained checkpoint written by
    ``QIDDM_PL_noise.save_model`` (``nn/qdense.py:1456-1462``) from
    ``src/bloodmnist.py:199-201``;
  * ``image_{1..10}/step_{1..6}.png`` -- ``plt.imsave(..., cmap="gray")`` of the
    six rows of ``diff.sample(first_x, n_iters=5)`` (``src/bloodmnist.py:231-277``),
    i.e. outputs of PennyLane-Lightning run by the reference's authors.

This script copies the checkpoints and decodes the PNGs into ONE small npz
(``uint8 (folder, image, step, 28, 28)``; the gray colour map makes R=G=B, so
the red channel is the value).  It runs in the build container only (it reads
``/root/reference``); the outputs under ``tests/golden/reference_runs/`` are
data fixtures and travel with the repo.  No reference source text is copied.

    python tests/golden/make_reference_runs.py
"""
import io
import pathlib
import zipfile

import numpy as np

REF = pathlib.Path("/root/reference/results_rebuttal_complex_dataset")
OUT = pathlib.Path(__file__).resolve().parent / "reference_runs"

# (zip, folder inside the zip, label suffix of the checkpoint)
RUNS = [
    ("medmnist.zip", "medmnist/bloodmnist", 0),
    ("medmnist.zip", "medmnist/PneumoniaMNIST", 0),
    ("logo2kplus.zip", "logo2kplus/Ascari", 1),
    ("logo2kplus.zip", "logo2kplus/Phillips 66", 4),
    ("logo2kplus.zip", "logo2kplus/Sanyo", 5),
]


def _decode_png(raw: bytes) -> np.ndarray:
    import matplotlib.image as mpimg  # the writer was matplotlib's imsave
    img = mpimg.imread(io.BytesIO(raw), format="png")          # (H, W, 4) float32 in [0,1]
    rgb = np.rint(img[..., :3] * 255.0).astype(np.uint8)
    assert (rgb[..., 0] == rgb[..., 1]).all() and (rgb[..., 0] == rgb[..., 2]).all()
    return rgb[..., 0]


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    steps = np.zeros((len(RUNS), 10, 6, 28, 28), dtype=np.uint8)
    names = []
    for f, (zname, folder, label) in enumerate(RUNS):
        zf = zipfile.ZipFile(REF / zname)
        ck = f"{folder}/QIDDM_PL_noise=8_L=6_N=2_{label}.pt"
        tag = folder.split("/")[-1].replace(" ", "_")
        (OUT / f"{tag}__QIDDM_PL_noise=8_L=6_N=2_{label}.pt").write_bytes(zf.read(ck))
        names.append(f"{tag}__QIDDM_PL_noise=8_L=6_N=2_{label}.pt")
        for i in range(10):
            for s in range(6):
                steps[f, i, s] = _decode_png(zf.read(f"{folder}/image_{i + 1}/step_{s + 1}.png"))
    np.savez_compressed(OUT / "steps.npz", steps=steps, checkpoints=np.array(names))
    print("wrote", OUT, steps.shape, names)


if __name__ == "__main__":
    main()
