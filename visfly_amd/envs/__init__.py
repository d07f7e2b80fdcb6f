from .base import DroneGymEnvsBase  # noqa: F401
from .tasks import HoverEnv, NavigationEnv, RacingEnv  # noqa: F401
