"""Generates tests/golden/beta_schedules.npz from the REFERENCE (run in the build container only: /root/reference is
absent on the GPU box).  Every 50th beta of each schedule name the reference's make_beta_schedule accepts
(MToV/losses/ddpm.py:78-99) at the shipped (timesteps, linear_start, linear_end) = (1000, .0015, .0195)."""
import os
import sys
import types

import numpy as np


def main():
    def stub(name, **a):
        m = types.ModuleType(name)
        for k, v in a.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    tv = stub("torchvision")                      # ddpm.py:19,25-27,31 import these and never use them on this path
    tv.utils = stub("torchvision.utils", make_grid=None)
    tv.transforms = stub("torchvision.transforms", ToTensor=object, ToPILImage=object)
    stub("cv2")
    sys.path.insert(0, "/root/reference/MToV")
    from losses.ddpm import make_beta_schedule

    out = {n: np.asarray(make_beta_schedule(n, 1000, 0.0015, 0.0195))[::50].astype(np.float64) for n in ("linear", "cosine", "sqrt_linear", "sqrt")}
    np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "beta_schedules.npz"), **out)


if __name__ == "__main__":
    main()
