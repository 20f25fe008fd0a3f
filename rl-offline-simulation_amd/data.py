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
