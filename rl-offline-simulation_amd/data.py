"""Logged-experience container, same schema as the reference's OfflineDataset
(offsim4rl/data.py:16-118).  HDF5 I/O and SAS_Dataset are storage / encoder-training helpers
outside the replay-loop path and are not reproduced."""
import enum
import json
import logging
from collections import namedtuple

import numpy as np


class ProbDistribution(enum.Enum):
    """Type of probability distribution used to describe actions (data.py:16-30)."""
    NoProbability = 0
    LoggedActionOnly = 1
    Discrete = 2
    TorchDistribution = 3


Transition = namedtuple(
    "Transition",
    ["episode_id", "step", "observation", "action", "action_distribution", "reward", "next_observation", "terminal", "info"])

REQUIRED_KEYS = ("observations", "actions", "rewards", "next_observations", "terminals")


class OfflineDataset:
    """Dict of equal-length arrays + spaces (data.py:38-66)."""

    def __init__(self, observation_space, action_space, action_dist_type, **experience):
        self._validate_experience(experience)
        self.observation_space = observation_space
        self.action_space = action_space
        self.action_dist_type = action_dist_type
        self.experience = experience

    def iterate_row_tuples(self):
        """data.py:68-79 (missing `steps` => step 0 for every row)."""
        e = self.experience
        for i in range(e["observations"].shape[0]):
            yield Transition(
                e["episode_ids"][i] if "episode_ids" in e else None,
                e["steps"][i] if "steps" in e else 0,
                e["observations"][i], e["actions"][i],
                e["action_distributions"][i] if "action_distributions" in e else None,
                e["rewards"][i], e["next_observations"][i], e["terminals"][i],
                e["infos"][i] if "infos" in e else {})

    @staticmethod
    def _validate_experience(experience):
        """data.py:100-118: required keys, equal lengths, matching observation shapes."""
        for k in REQUIRED_KEYS:
            if k not in experience:
                raise ValueError(f"Missing required key {k} in experience")
        if experience["observations"].shape != experience["next_observations"].shape:
            raise ValueError("Shapes in observations and next_observations do not match")
        n = experience["observations"].shape[0]
        for k in experience:
            if len(experience[k]) != n:
                raise ValueError(f"Length of {k} ({len(experience[k])}) does not match length of observations ({n})")
        if "steps" not in experience:
            logging.warning("Missing steps in experience. Algorithms may need to assume all states can be initial states...")
        if "episode_ids" not in experience:
            logging.warning("Missing episode_ids in experience. Some algorithms may not be compatible with this dataset.")

    def __len__(self):
        return int(np.shape(self.experience["observations"])[0])

    # ---- ingestion formats ----
    # The reference stores datasets as HDF5: one gzip dataset per experience key, the spaces and the
    # ProbDistribution pickled into group attributes (data.py:85-98, 120-146).  h5py is not part of this stack, so
    # the container this package WRITES is .npz with the same keys and the spaces described in JSON (nothing is
    # unpickled on load); HDF5 files written by the reference load through load_hdf5 -- with h5py where it is
    # importable, with the package's own reader of that subset of the format (hdf5.py) where it is not.
    def save_npz(self, path):
        meta = json.dumps({"observation_space": _space_to_json(self.observation_space), "action_space": _space_to_json(self.action_space),
                           "action_dist_type": ProbDistribution(self.action_dist_type).name})
        np.savez_compressed(path, __spaces__=np.frombuffer(meta.encode(), dtype=np.uint8), **{k: np.asarray(v) for k, v in self.experience.items()})

    @classmethod
    def load_npz(cls, path):
        with np.load(path, allow_pickle=False) as z:
            meta = json.loads(z["__spaces__"].tobytes().decode())
            return cls(_space_from_json(meta["observation_space"]), _space_from_json(meta["action_space"]),
                       ProbDistribution[meta["action_dist_type"]], **{k: z[k] for k in z.files if k != "__spaces__"})

    @classmethod
    def load_hdf5(cls, path, group_name=None, reader=None):
        """data.py:81-83 / HDF5Dataset (data.py:120-146): arrays are read into memory (the device table copies them anyway).
        The `infos/<key>` datasets written by record_dataset_in_memory (utils/dataset_utils.py:83-113) come back as experience
        keys "infos/<key>" (the reference's own HDF5Dataset trips over that group -- it validates `len(group)` against the number of
        rows -- and reads such files through utils/dataset_utils.py:26-34 instead; here one loader serves both).  The pickled
        gym.spaces attributes are decoded by a restricted unpickler onto spaces.Discrete / Box.
        `reader`: "h5py", "native" (this package's own reader, hdf5.py: the subset of the format such files use, NumPy + zlib), or
        None = h5py where it is importable, the native reader otherwise.  As in the reference (data.py:123) a `group_name` the file
        does not hold falls back on the root group."""
        if reader not in (None, "h5py", "native"):
            raise ValueError(f"unknown HDF5 reader {reader!r}")
        mod = None
        if reader != "native":
            try:
                import h5py as mod
            except ImportError:
                if reader == "h5py":
                    raise
        if mod is None:
            from . import hdf5 as mod
        with mod.File(path, "r") as fin:
            group = fin.get(group_name, default=fin) if group_name else fin
            return cls.from_hdf5_group(group)

    @classmethod
    def from_hdf5_group(cls, group):
        """Any h5py-like group: `.attrs` mapping with the pickled spaces, datasets by key, sub-groups flattened to "a/b"."""
        def attr(name, default=None):
            b = group.attrs.get(name, None)
            return default if b is None else restricted_loads(b.tobytes() if hasattr(b, "tobytes") else bytes(b))

        exp = {}

        def walk(g, prefix):
            for k in g:
                v = g[k]
                if hasattr(v, "keys") and not hasattr(v, "shape"):
                    walk(v, prefix + k + "/")
                else:
                    exp[prefix + k] = np.asarray(v)
        walk(group, "")
        return cls(attr("observation_space"), attr("action_space"), attr("action_dist_type", ProbDistribution.NoProbability), **exp)


def _space_to_json(sp):
    if sp is None:
        return None
    if hasattr(sp, "n") and not hasattr(sp, "low"):
        return {"kind": "Discrete", "n": int(sp.n)}
    if hasattr(sp, "low") and hasattr(sp, "high"):
        return {"kind": "Box", "shape": [int(x) for x in sp.shape], "dtype": np.dtype(sp.dtype).name,
                "low": np.asarray(sp.low, np.float64).reshape(-1).tolist(), "high": np.asarray(sp.high, np.float64).reshape(-1).tolist()}
    raise ValueError(f"cannot describe space {sp!r}: only Discrete and Box are part of the OfflineDataset schema")


def _space_from_json(d):
    from . import spaces
    if d is None:
        return None
    if d["kind"] == "Discrete":
        return spaces.Discrete(d["n"])
    shape = tuple(d["shape"])
    dt = np.dtype(d["dtype"])
    return spaces.Box(np.array(d["low"], np.float64).reshape(shape).astype(dt), np.array(d["high"], np.float64).reshape(shape).astype(dt), shape, dt.type)


class _PickledSpace:
    """Receives the state of a pickled gym.spaces.Discrete / Box (their __reduce_ex__ is object.__reduce_ex__: class + __dict__)."""

    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else state[0] or {})

    def resolve(self):
        from . import spaces
        d = self.__dict__
        if "n" in d:
            return spaces.Discrete(int(d["n"]))
        shape = tuple(d.get("_shape", d.get("shape", np.shape(d["low"]))))
        return spaces.Box(d["low"], d["high"], shape, np.dtype(d.get("dtype", np.float32)).type)


class _Inert:
    """Stands in for numpy.random's pickle helpers (__generator_ctor, __bit_generator_ctor, PCG64, SeedSequence, ...): callable,
    accepts any state, does nothing."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Inert()

    def __setstate__(self, state):
        pass


def restricted_loads(data):
    """pickle.loads for the attributes the reference stores in its HDF5 files (data.py:88-91): gym.spaces.Discrete / Box map
    onto this package's spaces, offsim4rl.data.ProbDistribution onto the local enum, NumPy array / dtype reconstruction is
    allowed, everything else is refused -- a data file cannot name arbitrary callables."""
    import io
    import pickle

    allowed_numpy = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"), ("numpy", "ndarray"),
                     ("numpy", "dtype"), ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
                     ("numpy.core.numeric", "_frombuffer"), ("numpy._core.numeric", "_frombuffer")}

    class U(pickle.Unpickler):
        def find_class(self, module, name):
            if module.split(".")[0] in ("gym", "gymnasium") and name in ("Discrete", "Box"):
                return _PickledSpace
            if name == "ProbDistribution":
                return ProbDistribution
            if (module, name) in allowed_numpy:
                import importlib
                return getattr(importlib.import_module(module), name)
            if module.startswith("numpy") and name in ("float32", "float64", "int64", "int32", "uint8", "bool_"):
                return getattr(np, name)
            if (module, name) == ("copyreg", "_reconstructor") or (module, name) == ("builtins", "object"):
                import copyreg
                return copyreg._reconstructor if name == "_reconstructor" else object
            # a space that has been seeded or sampled carries its np_random Generator (gym.spaces.Space._np_random); nothing of it
            # is needed: its constructors and bit generators unpickle into an inert placeholder instead of being imported
            if module.startswith("numpy.random"):
                return _Inert
            raise pickle.UnpicklingError(f"refusing to load {module}.{name} from a dataset file")

    obj = U(io.BytesIO(data)).load()
    return obj.resolve() if isinstance(obj, _PickledSpace) else obj
