"""RevealedRandomnessEnv: an environment whose agent reveals its action distribution
(offsim4rl/core.py:7-45).  gym is absent from this stack; the class is a plain base with the
same method surface (reset / step / step_dist)."""
import numpy as np

try:  # torch is only needed to recognise Distribution objects
    from torch.distributions import Distribution
except Exception:  # pragma: no cover
    class Distribution:  # type: ignore
        pass


class RevealedRandomnessEnv:
    def reset(self, seed=None):
        raise NotImplementedError

    def step(self, action):
        raise NotImplementedError

    def step_dist(self, action_dist):
        """core.py:12-45: sample an action from the revealed distribution, then step()."""
        if isinstance(action_dist, Distribution):
            orig_action = action_dist.sample()
            numpy_action = orig_action.cpu().numpy()
        elif isinstance(action_dist, np.ndarray):
            orig_action = np.random.choice(a=np.arange(len(action_dist)), p=action_dist)
            numpy_action = orig_action
        else:
            raise ValueError("action_dist must be a torch.distributions.Distribution or numpy array")
        return (orig_action, *self.step(numpy_action))
