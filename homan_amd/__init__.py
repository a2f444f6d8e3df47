"""homan_amd -- MI355X-native implementation of HOMan's joint hand-object optimisation hot path.

Drop-in for `homan.homan.HOMan` / `homan.jointopt.optimize_hand_object` of hassony2/homan; every leaf
(silhouette rasteriser, MANO LBS, rigid transforms, SDF collision, contact, small losses, Adam) is a hand-written
HIP kernel for gfx950 behind the C ABI of include/homan_amd.h.  No CPU fallback.
"""
from .homan import HOMan  # noqa: F401
from .jointopt import optimize_hand_object  # noqa: F401
