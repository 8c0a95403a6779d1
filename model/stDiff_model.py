from stdiff_amd.model.stDiff_model import *  # noqa: F401,F403  (reference: model/stDiff_model.py)
from stdiff_amd.model.stDiff_model import DiT_stDiff  # noqa: F401
