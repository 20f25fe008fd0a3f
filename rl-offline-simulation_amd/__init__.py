"""MI355X-native Per-State Rejection Sampling engine (offsim4rl's replay-loop hot path).

Import as `rl_offline_simulation_amd` (the repo-root shim maps the hyphenated directory name)."""
from . import _lib  # noqa: F401
from .data import OfflineDataset, ProbDistribution, Transition  # noqa: F401
from .core import RevealedRandomnessEnv  # noqa: F401
from . import spaces, synth  # noqa: F401

__all__ = ["OfflineDataset", "ProbDistribution", "Transition", "RevealedRandomnessEnv", "spaces", "synth"]
