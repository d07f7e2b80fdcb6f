from .base import DroneGymEnvsBase  # noqa: F401
from .tasks import HoverEnv, HoverEnv2, NavigationEnv, NavigationEnv2, RacingEnv, RacingEnv2  # noqa: F401
