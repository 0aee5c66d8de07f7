"""Time of one delay calibration (fxc_estimate_delay) for device-resident streams of the reference's size."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit('/', 2)[0])
from effex_amd import plan as plan_mod  # noqa: E402


def main():
    rate = 2.4e6
    for n in (4096, 1 << 18, 1 << 22):
        rng = np.random.default_rng(1)
        a = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
        b = np.roll(a, 3)
        da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        with plan_mod.FxPlan(2, 4096, 4, 4096 * 4) as p:
            est = p.estimate_delay(da, db, rate)
            t0 = time.perf_counter()
            reps = 20
            for _ in range(reps):
                p.estimate_delay(da, db, rate)
            dt = (time.perf_counter() - t0) / reps
        print(json.dumps({"n": n, "padded": 1 << int(np.ceil(np.log2(2 * n))), "ms_per_calibration": round(dt * 1e3, 4),
                          "estimate_samples": round(est * rate, 6)}))


if __name__ == '__main__':
    main()
