"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  CPU restatement of the per-scene standardisation the
reference applies to node / edge feature frames: sklearn StandardScaler (population variance, zero scale -> 1)
fitted on columns [c_first:] in float64, then cast to float32 (reference processing/data.py:471-506, 512-519).
Pinned by tests/golden/ingest_small.npz, which the reference's own dataLoader produced (tests/golden/make_golden.py)."""
import numpy as np


def standardize(cols64: np.ndarray, c_first: int) -> np.ndarray:
    x = np.asarray(cols64, np.float64)
    out = x.copy()
    body = x[:, c_first:]
    mean = body.mean(axis=0)
    scale = np.sqrt(((body - mean) ** 2).mean(axis=0))
    scale[scale < 10 * np.finfo(np.float64).eps] = 1.0
    out[:, c_first:] = (body - mean) / scale
    return out.astype(np.float32)
