#!/usr/bin/env python3
"""BASELINE configs[4] at its size: |score - float64 oracle| of the HIP step and of the REFERENCE'S fp32 op sequence (NumPy,
four matrix products), by score magnitude.  The numbers quoted in tests/test_token_pooled_full_size.py and DESIGN.md section 2."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_token_pooled_full_size as T  # noqa: E402


def main():
    z = T.problem.__wrapped__() if hasattr(T.problem, "__wrapped__") else T.problem.__pytest_wrapped__.obj()
    ref = T._oracle(z)
    ref32 = T._reference_fp32_scores(z)
    st = T._step(z)
    scores = torch.empty((T.B, T.N_CAND), device="cuda:0")
    st.forward_backward(T._batch(z), scores=scores)
    torch.cuda.synchronize()
    x = scores.cpu().numpy()
    err, err32, mag = np.abs(x - ref["outputs"]), np.abs(ref32.astype(np.float64) - ref["outputs"]), np.abs(ref["outputs"])
    print(f"{'|x| band':>14s} {'elements':>10s} {'HIP max':>10s} {'HIP rms':>10s} {'HIP >1e-4':>10s} {'ref32 max':>10s} {'ref32 rms':>10s} {'ref32 >1e-4':>11s}")
    for lo, hi in ((0, 10), (10, 30), (30, 60), (60, 100), (100, 130), (130, 1e9)):
        b = (mag > lo) & (mag <= hi)
        if b.any():
            print(f"{f'({lo}, {hi if hi < 1e9 else chr(8734)}]':>14s} {int(b.sum()):10d} {err[b].max():10.2e} {np.sqrt((err[b] ** 2).mean()):10.2e} {int((err[b] > 1e-4).sum()):10d} "
                  f"{err32[b].max():10.2e} {np.sqrt((err32[b] ** 2).mean()):10.2e} {int((err32[b] > 1e-4).sum()):11d}")
    print("HIP vs ref32 directly: max", np.abs(x - ref32).max(), " elements > 1e-4:", int((np.abs(x - ref32) > 1e-4).sum()), "of", x.size)


if __name__ == "__main__":
    main()
