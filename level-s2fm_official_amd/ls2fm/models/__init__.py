from .SDF import SDF            # noqa: F401
from .RadF import RadF          # noqa: F401
from .Renderer import Renderer  # noqa: F401
