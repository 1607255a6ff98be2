#!/usr/bin/env python3
"""Host -> HBM upload rate of a trajectory shard: pageable tensor.to(device) vs _hip.upload_f32
(page-locked in place for fp32 sources, pinned double-buffered conversion for fp64 ones).  One JSON line per case."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "colvars-finder_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from colvarsfinder import _hip  # noqa: E402

dev = torch.device("cuda")
torch.zeros(1, device=dev)
for name, dtype, n in (("fp32 1 GiB", np.float32, 1 << 28), ("fp64 1 GiB", np.float64, 1 << 27)):
    a = np.random.default_rng(0).random(n, dtype=np.float64).astype(dtype).reshape(-1, 22, 3) if n % 66 == 0 else \
        np.random.default_rng(0).random(n // 66 * 66, dtype=np.float64).astype(dtype).reshape(-1, 22, 3)
    gib = a.nbytes / 2**30
    for tag, fn in (("pageable .to(device)", lambda: torch.as_tensor(a).to(device=dev, dtype=torch.float32)),
                    ("upload_f32", lambda: _hip.upload_f32(a, dev))):
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = fn()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        ok = bool(torch.equal(out[:1000].cpu(), torch.as_tensor(a[:1000]).to(torch.float32)) and
                  torch.equal(out[-1000:].cpu(), torch.as_tensor(a[-1000:]).to(torch.float32)))
        print(json.dumps(dict(case=name, path=tag, seconds=best, source_GiB=gib, source_GiB_per_s=gib / best, matches=ok)))
        del out
