"""planeverb_amd -- MI355X-native (HIP, gfx950) implementation of Planeverb's FDTD + impulse-response-analysis path
behind Planeverb's own C-ABI.  See DESIGN.md and include/planeverb_amd.h."""
from . import api  # noqa: F401
from .build import build, LIB_PATH  # noqa: F401
