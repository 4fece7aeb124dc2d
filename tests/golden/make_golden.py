"""Generates tests/golden/*.npz by running the REAL reference modules (imported from
/root/reference through oracle/ref_shim.py) on seeded weights/inputs.  Build-container only.

    python tests/golden/make_golden.py [case ...]      (no names = every case)

The fixtures pin oracle/tts_oracle.py on machines without the reference tree (the GPU box).
Cases/inputs are defined in tests/golden/cases.py and shared with tests/test_oracle_pin.py.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from tests.golden import cases  # noqa: E402


def main():
    torch.set_num_threads(1)
    only = set(sys.argv[1:])
    for name, fn in cases.CASES.items():
        if only and name not in only:
            continue
        out = fn("ref")
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **{k: v.detach().cpu().numpy() for k, v in out.items()})
        print(name, {k: tuple(v.shape) for k, v in out.items()}, "%.1f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
