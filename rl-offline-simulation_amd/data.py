"""Logged-experience container, same schema as the reference's OfflineDataset
(offsim4rl/data.py:16-118).  HDF5 I/O and SAS_Dataset are storage / encoder-training helpers
outside the replay-loop path and are not reproduced."""
import enum
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
    # the native container here is .npz with the same keys; HDF5 files written by the reference load through
    # load_hdf5 wherever h5py is importable.
    def save_npz(self, path):
        import pickle
        meta = np.frombuffer(pickle.dumps((self.observation_space, self.action_space, self.action_dist_type)), dtype=np.uint8)
        np.savez_compressed(path, __spaces__=meta, **{k: np.asarray(v) for k, v in self.experience.items()})

    @classmethod
    def load_npz(cls, path):
        import pickle
        with np.load(path, allow_pickle=False) as z:
            obs_space, act_space, dist_type = pickle.loads(z["__spaces__"].tobytes())
            return cls(obs_space, act_space, dist_type, **{k: z[k] for k in z.files if k != "__spaces__"})

    @classmethod
    def load_hdf5(cls, path, group_name=None):
        """data.py:81-83 / HDF5Dataset (data.py:120-146): arrays are read into memory (the device table copies them anyway)."""
        try:
            import h5py
        except ImportError as e:  # pragma: no cover - h5py is absent from the build image
            raise ImportError("load_hdf5 needs h5py; convert with the reference's tools or use save_npz/load_npz") from e
        import pickle
        with h5py.File(path, "r") as fin:
            group = fin.get(group_name, default=fin) if group_name else fin

            def attr(name, default=None):
                b = group.attrs.get(name, default=None)
                return pickle.loads(b.tobytes()) if b is not None else default
            return cls(attr("observation_space"), attr("action_space"), attr("action_dist_type", ProbDistribution.NoProbability),
                       **{k: np.asarray(group[k]) for k in group})
