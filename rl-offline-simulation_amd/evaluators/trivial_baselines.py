"""Trivial baselines (offsim4rl/evaluators/trivial_baselines.py:6-29): two switches on the same kernels --
`always accept` (no RNG draw) and `single latent state`."""
from .. import _lib as L
from .per_state_rejection import PerStateRejectionSampling


class _DummyEncoder:
    def encode(self, observation):
        return [0] * observation.shape[0]


class FollowObservationOnly(PerStateRejectionSampling):
    """Follow observation queues, but accept the transition irrespective of the action probabilities."""
    _device_reject_mode = L.REJECT_NEVER

    def _reject(self, p_new, p_log, a) -> bool:
        return False


class FollowActionOnly(PerStateRejectionSampling):
    """Reject based on the action probabilities, but treat all observations the same."""

    def __init__(self, dataset, **kwargs):
        super().__init__(dataset, num_states=1, encoder=_DummyEncoder(), **kwargs)


class ServeRandomTransitions(PerStateRejectionSampling):
    """Just serve random transitions."""
    _device_reject_mode = L.REJECT_NEVER

    def __init__(self, dataset, **kwargs):
        super().__init__(dataset, num_states=1, encoder=_DummyEncoder(), **kwargs)

    def _reject(self, p_new, p_log, a) -> bool:
        return False
