import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PSRS_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz"))
                    if not os.path.basename(p).startswith(("rng", "enc_", "td_", "td2_", "queue_", "exo_", "h5_", "philox_")))


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def is_f32_case(d, key):
    return d[key].dtype == np.float32 and d["in_p_log"].dtype == np.float32
