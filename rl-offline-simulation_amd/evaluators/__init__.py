from .psrs import PSRS, BatchedPSRS, evalMC_psrs, evalmc_rollouts, qlearn_psrs, expSARSA_psrs, SHUFFLE_PER_ROLLOUT, SHUFFLE_SHARED, SHUFFLE_NONE  # noqa: F401
from .per_state_rejection import PerStateRejectionSampling  # noqa: F401
from .trivial_baselines import FollowObservationOnly, FollowActionOnly, ServeRandomTransitions  # noqa: F401
from .queue_evaluator import QueueEvaluator, BatchedQueueEvaluator  # noqa: F401
from .psrs_exo import PSRS_Exo  # noqa: F401
from .vector_env import VectorPSRS  # noqa: F401
