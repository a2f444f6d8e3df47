"""Constants of the hot path (reference homan/constants.py:32-33, homan/losses.py:90,95)."""
REND_SIZE = 256                 # size of the target masks / silhouette raster
BBOX_EXPANSION_FACTOR = 0.3     # ROI padding used when the target masks are cut
INTERACTION_Z_THRESH = 3        # Losses.thresh
INTERACTION_BBOX_EXPANSION = 0.2   # Losses.expansion
CONTACT_THRESH = 0.010          # contactloss.compute_contact_loss defaults
COLLISION_THRESH = 0.020
SDF_SCALE_FACTOR = 0.2          # SDFSceneLoss.forward(scale_factor=0.2)
INTERACTION_MAPPING = {"default": ["lhand", "rhand"]}
