"""Import alias for the product package.

The package directory is named `vision-textile-inspection_amd/` (not a valid Python
identifier), so this stub maps the importable name `vti_amd` onto it.
"""
import os as _os

__path__.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                                 "vision-textile-inspection_amd"))

from .api import *  # noqa: E402,F401,F403
from .api import __all__  # noqa: E402,F401
