"""autoinst_amd -- MI355X-native NCuts hot path of artonson/autoinst.

Affinity build + recursive normalized cut as hand-written HIP kernels (gfx950) behind the
reference's own Python call surface.  See DESIGN.md and include/autoinst_hip.h.
"""
from .config import CONFIG, CONFIG_SPATIAL, CONFIG_TARL_SPATIAL, CONFIG_TARL_SPATIAL_DINO, PROXIMITY_THRESHOLD, SPLIT_LIM  # noqa: F401


def __getattr__(name):
    # compute entry points load the HIP library on first use and raise if it is missing
    if name in ("normalized_cut", "ncuts", "get_affinity_matrix", "build_affinity", "ncuts_chunk", "ncuts_labels", "ncuts_labels_batch",
                "Context", "DeviceGraph", "default_context", "last_stats", "fiedler", "sweep", "lsym_apply", "bench_spmv", "eigs_smallest"):
        from . import ncuts_api as _n
        return getattr(_n, name)
    if name in ("Metrics", "score", "label_pairs", "merge_chunks_unite_instances2", "merge_associate", "unique_points"):
        from . import labels_api as _l
        return getattr(_l, name)
    raise AttributeError(name)
