#!/opt/conda/bin/python3.9
"""Write the HDF5 fixtures under tests/golden/hdf5/ by RUNNING THE REFERENCE'S OWN `OfflineDataset.save_hdf5`.

Run from the repo root:   /opt/conda/bin/python3.9 tests/golden/make_hdf5_golden.py

The image's main interpreter (/usr/bin/python3, 3.10) has no h5py -- which is why the product reads HDF5 with its own reader
(rl-offline-simulation_amd/hdf5.py) -- but the side interpreter /opt/conda/bin/python3.9 carries a real h5py 3.3.0 on libhdf5 1.10.6 and
NumPy 1.26.  The files below are therefore genuine h5py output of the reference's writer (offsim4rl/data.py:85-98: one gzip dataset per
experience key, the spaces and the ProbDistribution pickled into opaque attributes), produced by importing offsim4rl/data.py from
/root/reference BY PATH; nothing of the reference is copied, a fixture is the file it wrote plus the arrays it was given (.npz).

What is NOT real here, and why: gym and torch are absent from that interpreter too, and offsim4rl/data.py imports both at module level
(`import gym`, `from torch.utils.data import Dataset`).  The script registers, in sys.modules only, (a) an empty `torch.utils.data.Dataset`
and `offsim4rl.core` (neither takes part in save_hdf5), and (b) `gym.spaces.box.Box` / `gym.spaces.discrete.Discrete` classes that carry
the attribute set gym 0.21 - 0.26 instances have (`_shape`, `dtype`, `low`, `high`, `bounded_below`, `bounded_above`, `low_repr`, `high_repr`,
`_np_random`; `n`, `start`), so that the pickled attributes name the modules and carry the state a gym installation would have written.
The HDF5 container -- superblock, object headers, symbol-table groups, chunk B-trees, deflate chunks, opaque attributes, the enum h5py
stores NumPy bools as -- is what these fixtures pin; the pickle side was already pinned by test_restricted_loads_* on byte streams.

Also written, straight through h5py (no reference code involved; they exist to exercise the reader's format coverage): files with many
small chunks (a two-level chunk B-tree), shuffle + fletcher32 filters, contiguous and compact layouts, a scalar, big-endian and 16-bit
types, enough keys for a multi-node group B-tree, and the same content under libver='latest' (object header v2 + link messages).
"""
import importlib.util
import os
import sys
import types

import numpy as np
import h5py

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden", "hdf5")


# ---- sys.modules stand-ins for what offsim4rl/data.py imports and this interpreter lacks (see the docstring) ----
def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Space:
    def __init__(self, shape, dtype):
        self._shape = None if shape is None else tuple(shape)
        self.dtype = None if dtype is None else np.dtype(dtype)
        self._np_random = None


class Box(_Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        shape = tuple(shape) if shape is not None else np.shape(low)
        super().__init__(shape, dtype)
        self.low = np.full(shape, low, dtype=dtype) if np.isscalar(low) else np.asarray(low, dtype)
        self.high = np.full(shape, high, dtype=dtype) if np.isscalar(high) else np.asarray(high, dtype)
        self.low_repr = str(low)
        self.high_repr = str(high)
        self.bounded_below = -np.inf < self.low
        self.bounded_above = np.inf > self.high


class Discrete(_Space):
    def __init__(self, n, start=0):
        super().__init__((), np.int64)
        self.n = int(n)
        self.start = int(start)


Box.__module__ = "gym.spaces.box"
Discrete.__module__ = "gym.spaces.discrete"
_Space.__module__ = "gym.spaces.space"
_Space.__name__ = _Space.__qualname__ = "Space"
gym = _module("gym", Space=_Space)
gym.spaces = _module("gym.spaces", Box=Box, Discrete=Discrete, Space=_Space)
_module("gym.spaces.box", Box=Box)
_module("gym.spaces.discrete", Discrete=Discrete)
_module("gym.spaces.space", Space=_Space)
torch = _module("torch")
torch.utils = _module("torch.utils")
torch.utils.data = _module("torch.utils.data", Dataset=type("Dataset", (), {}))
pkg = _module("offsim4rl")
pkg.__path__ = [os.path.join(REF, "offsim4rl")]
pkg.core = _module("offsim4rl.core")

spec = importlib.util.spec_from_file_location("offsim4rl.data", os.path.join(REF, "offsim4rl/data.py"))
ref_data = importlib.util.module_from_spec(spec)
sys.modules["offsim4rl.data"] = ref_data
spec.loader.exec_module(ref_data)
OfflineDataset, ProbDistribution = ref_data.OfflineDataset, ref_data.ProbDistribution


def _save(name, ds, group_name=None, nested=False):
    path = os.path.join(OUT, name + ".hdf5")
    ds.save_hdf5(path, group_name)  # the reference's writer
    if not nested:
        # read back by the reference's own loader, so that the expected arrays are what the REFERENCE sees in its file
        back = OfflineDataset.load_hdf5(path, group_name)
        exp = {k: np.asarray(back.experience[k]) for k in back.experience}
        back.close()
    else:
        # HDF5Dataset (data.py:120-133) iterates the top-level names only and fails its own length validation on the "infos" GROUP
        # ("Length of infos (2) does not match ..."); files with infos/* are read in the reference by utils/dataset_utils.py:13-34
        # (visititems + [:]), which cannot be imported here (agents, gym envs): the same two h5py calls, spelled out
        exp = {}
        with h5py.File(path, "r") as f:
            f.visititems(lambda n, item: exp.__setitem__(n.replace("/", "__"), item[:]) if isinstance(item, h5py.Dataset) else None)
    np.savez_compressed(os.path.join(OUT, name + ".expected.npz"), **exp)
    print(name, os.path.getsize(path), "bytes", sorted(exp))


def cartpole_like(n, seed, with_infos):
    """A log with the schema record_dataset_in_memory writes (utils/dataset_utils.py:103-113): float32 observations, int64 actions,
    float32 action distributions, float64 rewards, bool terminals, int64 episode ids / steps, and flattened infos/<key> columns."""
    rng = np.random.default_rng(seed)
    obs = rng.normal(size=(n + 1, 4)).astype(np.float32)
    term = rng.random(n) < 0.02
    steps = np.zeros(n, np.int64)
    ep = np.zeros(n, np.int64)
    for i in range(1, n):
        steps[i] = 0 if term[i - 1] else steps[i - 1] + 1
        ep[i] = ep[i - 1] + (1 if term[i - 1] else 0)
    p = rng.dirichlet(np.ones(2), size=n).astype(np.float32)
    exp = dict(episode_ids=ep, steps=steps, observations=obs[:-1], actions=(rng.random(n) < p[:, 1]).astype(np.int64),
               action_distributions=p, rewards=np.ones(n, np.float64), next_observations=obs[1:], terminals=term)
    if with_infos:
        # dataset_utils.py:83-100: info values of supported scalar types become "infos/<key>" columns; h5py creates the "infos" group
        exp["infos/z"] = rng.integers(0, 162, size=n).astype(np.int64)
        exp["infos/TimeLimit.truncated"] = rng.random(n) < 0.01
    return exp


def main():
    os.makedirs(OUT, exist_ok=True)
    # 1. the two datasets of the reference's tests/test_data.py:11-62, through the reference's writer
    two = dict(observations=np.array([[0.0], [0.5]], dtype=np.float32), actions=np.array([0, 1], dtype=np.int64),
               rewards=np.array([0.0, 1.0], dtype=np.float32), next_observations=np.array([[0.5], [0.0]], dtype=np.float32),
               terminals=np.array([False, True], dtype=bool))
    box1 = gym.spaces.Box(low=0, high=1, shape=(1,), dtype=np.float32)
    _save("ref_test_data_discrete", OfflineDataset(box1, gym.spaces.Discrete(4), ProbDistribution.Discrete,
                                                    action_distributions=np.full((2, 4), 0.25, dtype=np.float32), **two))
    _save("ref_test_data_no_dist", OfflineDataset(box1, gym.spaces.Discrete(4), ProbDistribution.NoProbability, **two))
    # 2. a CartPole-shaped log with infos/*, in the root group and in a named group
    high = np.array([4.8, np.finfo(np.float32).max, 0.42, np.finfo(np.float32).max], dtype=np.float32)
    cart = gym.spaces.Box(-high, high, dtype=np.float32)
    _save("cartpole_like_3k_infos", OfflineDataset(cart, gym.spaces.Discrete(2), ProbDistribution.Discrete, **cartpole_like(3000, 7, True)), nested=True)
    _save("cartpole_like_10k_group", OfflineDataset(cart, gym.spaces.Discrete(2), ProbDistribution.Discrete, **cartpole_like(10000, 8, False)),
          group_name="train")

    # 3. straight h5py: format coverage of the reader (no reference code)
    rng = np.random.default_rng(3)
    cover = {
        "many_chunks_i32": (np.arange(40000, dtype=np.int32) % 977, dict(chunks=(64,), compression="gzip")),       # 625 chunks: 2-level B-tree
        "many_chunks_2d": ((np.arange(3000 * 5) % 251).reshape(3000, 5).astype(np.float32), dict(chunks=(7, 2), compression="gzip", shuffle=True)),
        "fletcher": (rng.integers(0, 9, size=(500, 3)).astype(np.int16), dict(chunks=(128, 3), fletcher32=True, compression="gzip")),
        "chunked_plain": (rng.integers(0, 9, size=1000).astype(np.uint8), dict(chunks=(300,))),
        "contiguous_f64": (np.linspace(0, 1, 777), {}),
        "big_endian_i4": (np.arange(50).astype(">i4"), {}),
        "half": (np.linspace(-2, 2, 33).astype(np.float16), {}),
        "bools_2d": (rng.random((40, 3)) < 0.5, dict(compression="gzip")),
        "scalar_f32": (np.float32(2.5), {}),
        "empty_1d": (np.zeros((0,), np.float32), {}),
        "fixed_str": (np.array([b"ab", b"cde", b""], dtype="S3"), {}),
    }
    for libver, fname in (("earliest", "coverage_earliest.hdf5"), ("latest", "coverage_latest.hdf5")):
        with h5py.File(os.path.join(OUT, fname), "w", libver=libver) as f:
            for k, (v, kw) in cover.items():
                if libver == "latest" and kw:
                    continue  # layout-v4 chunk indexes (fixed / extensible arrays) are outside what save_hdf5 writes: the reader refuses them
                f.create_dataset(k, data=v, **kw)
            f.attrs["i"] = np.int64(7)
            f.attrs["arr"] = np.arange(5, dtype=np.float32)
            f.attrs.create("blob", np.void(b"\x00\x01\x02payload\xff"))
            g = f.create_group("deep/er")
            g.create_dataset("x", data=np.arange(6).reshape(2, 3))
            g.attrs["note"] = np.bytes_(b"fixed")
            if libver == "earliest":
                many = f.create_group("many_keys")  # more entries than one symbol-table node holds (2 * leaf K = 8): a B-tree with several leaves
                for i in range(60):
                    many.create_dataset(f"k{i:03d}", data=np.full(3, i, np.int64))
        print(fname, os.path.getsize(os.path.join(OUT, fname)), "bytes")
    with h5py.File(os.path.join(OUT, "coverage_latest_chunked.hdf5"), "w", libver="latest") as f:
        f.create_dataset("x", data=np.arange(100), chunks=(10,))  # the reader must refuse this with a clear message
    np.savez_compressed(os.path.join(OUT, "coverage.expected.npz"), **{k: np.asarray(v) for k, (v, _) in cover.items()})


if __name__ == "__main__":
    main()
