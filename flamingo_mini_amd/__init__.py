"""Import name of the package.  The sources live in ../flamingo-mini_amd/ (a hyphen is not importable), so this stub
only extends the package search path; `import flamingo_mini_amd` then behaves like a normal package."""
import os as _os

__path__.append(_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "flamingo-mini_amd"))

from ._exports import *  # noqa: E402,F401,F403
from ._exports import __all__  # noqa: E402,F401
