"""Parity constants of the reference's NCuts path (values, not code).

Each value cites where the reference defines it (``/root/reference/pipeline/config.py``).
The reference star-imports these as module globals (``ncuts_utils.py:25``); here they
are plain keyword defaults so nothing is read at import time.
"""
from __future__ import annotations

PROXIMITY_THRESHOLD = 1.0   # config.py:65  radius of the affinity graph (metres, inclusive <=)
SPLIT_LIM = 0.01            # config.py:61  segments <= 1 % of the original chunk are never split
MAJOR_VOXEL_SIZE = 0.35     # config.py:56
CHUNK_SIZE = (25.0, 25.0, 25.0)  # config.py:57 (metres)
TARL_NORM = False           # config.py:64
NUM_DINO_FEATURES = 384     # config.py:67
NUM_TARL_FEATURES = 96      # tarl_extractor.py:84-89 (96-d MinkUNet features)
NUM_CUTS = 10               # normalized_cut.py:54  get_min_ncut(ev, D, w, 10)
EIGSH_SIGMA = 1e-10         # normalized_cut.py:49
MIN_POINTS_METRIC = 200     # metrics_class.py:17

# config.py:6-37 -- the three shipped NCuts configurations
CONFIG_SPATIAL = dict(name="spatial_1.0_t_0.075", alpha=1.0, theta=0.0, gamma=0.0, beta=0.0, T=0.075)
CONFIG_TARL_SPATIAL = dict(name="spatial_1.0_tarl_0.5_t_0.03", alpha=1.0, theta=0.5, gamma=0.0, beta=0.0, T=0.03)
CONFIG_TARL_SPATIAL_DINO = dict(name="spatial_1.0_tarl_0.5_dino_0.1_t_0.005", alpha=1.0, theta=0.5, gamma=0.1, beta=0.0, T=0.005)
CONFIG = CONFIG_TARL_SPATIAL  # config.py:87
