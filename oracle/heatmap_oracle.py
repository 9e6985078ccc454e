"""TEST INFRASTRUCTURE -- CPU restatement of the attention heat-map post-processing the reference applies before
drawing (reference demo.py:533-552), in the reference's own arithmetic: float32 numpy, numpy's percentile.
The reference's demo.py cannot be imported here (it needs torchvision), so this follows its text line by line;
numpy itself is the third-party arithmetic (np.power, np.percentile with the default linear interpolation).
Only tests may import this module."""
import numpy as np


def heatmap_postprocess(attn) -> np.ndarray:
    a = np.asarray(attn, dtype=np.float32).ravel()          # demo.py:530 attn_map.cpu().numpy() of an fp32 tensor
    a = np.power(a, 3)                                       # demo.py:533
    thr = np.percentile(a, 70)                               # demo.py:536
    a = np.where(a > thr, a, a * 0.3)                        # demo.py:537
    a = (a - a.min()) / (a.max() - a.min() + 1e-8)           # demo.py:540
    n = len(a)                                               # demo.py:543-552
    g = int(np.sqrt(n))
    if g * g == n:
        return a.reshape(g, g)
    g = int(np.ceil(np.sqrt(n)))
    padded = np.zeros(g * g)
    padded[:n] = a
    return padded.reshape(g, g)
