from .heuristic import CartpoleBoxEncoder  # noqa: F401
from .homer import HOMEREncoder  # noqa: F401
