"""MI355X-native Drost point-pair-feature registration path (drop-in for the
`pcl/alignment` path of nicolasavru/objective-slam).

The directory name contains a hyphen, so import it with
``importlib.import_module("objective-slam_amd")`` (tests/conftest.py does).

  ppf    -- Model / Scene / ppf_registration over the C-ABI (include/oslam.h)
  synth  -- deterministic synthetic clouds for tests and bench
  dist   -- the multi-GPU exchange step (all-reduce of maxima + all-gather of peaks)
  evaluate (imported on demand) -- recall vs occlusion, the protocol of the reference's analyze_mian.py
"""
from . import dist, ppf, synth  # noqa: F401
from .ppf import Model, Scene, ppf_registration, ht_dist  # noqa: F401
