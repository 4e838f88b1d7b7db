from .directed import *  # noqa: F401,F403
from .general import *  # noqa: F401,F403
from .signed import *  # noqa: F401,F403
