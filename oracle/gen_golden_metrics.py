"""Golden vectors for the step metrics from the reference's own libs/metrics.py + libs/utils.py
(numpy only, importable here).  Usage: python oracle/gen_golden_metrics.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")
from libs.metrics import pose_accuracy            # noqa: E402
from libs.utils import get_max_preds              # noqa: E402
import mtn_metrics_oracle as M                    # noqa: E402
import mtn_oracle as O                            # noqa: E402

rng = np.random.default_rng(7)
B, J, S = 6, 21, 48
_, _, tgt, _ = O.synthetic_batch(B, 4 * S, seed=77)
tgt = tgt.numpy()
out = (tgt + 0.35 * rng.standard_normal(tgt.shape)).astype(np.float32)
out[0, 3] = -1.0                      # all-negative map -> masked prediction
tgt[1, 5] = 0.0                       # empty target -> joint ignored
out = out.astype(np.float16).astype(np.float32)        # the fixture stores fp16: compute on those values
tgt = tgt.astype(np.float16).astype(np.float32)
preds, maxvals = get_max_preds(out)
acc, avg, cnt, pred = pose_accuracy(out, tgt)
a2, v2, c2, p2 = M.pose_accuracy(out, tgt)
assert np.array_equal(M.get_max_preds(out)[0], preds) and np.allclose(a2, acc) and c2 == cnt and np.array_equal(p2, pred)
path = os.path.join(os.path.dirname(HERE), "tests", "golden", "metrics.npz")
np.savez_compressed(path, out=out.astype(np.float16), tgt=tgt.astype(np.float16), preds=preds, maxvals=maxvals,
                    acc=acc, avg=np.float64(avg), cnt=np.int64(cnt))
print("oracle == reference; wrote", path, os.path.getsize(path))
