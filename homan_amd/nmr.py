"""The `nr.renderer.Renderer` surface the reference's visualisation code holds (reference homan/homan.py:168-176,
510-613; homan/visualize.py:44-128), on the HIP rasteriser.

This is the object the reference passes around as `model.renderer`: it carries the camera (`K`), the image size and the
light / background attributes the reference assigns, and renders

    renderer.render(vertices, faces, textures, K=None) -> (rgb (B,3,S,S), depth (B,S,S), alpha (B,S,S))
    renderer(vertices, faces, mode="silhouettes", K=None) -> (B,S,S)

for texture_size-1 textures (B,F,1,1,1,3) (homan/meshutils.py:7-51).  R = identity, t = 0, orig_size = 1, no
distortion - the only configuration the reference constructs.  Visualisation only: outputs carry no gradient (the
optimisation renders through homan_amd.ops.silhouette_loss / depth_render).
"""
import torch

from . import ops


class Renderer:
    def __init__(self, image_size=256, K=None, R=None, t=None, orig_size=1, anti_aliasing=True, fill_back=True, **_unused):
        if not anti_aliasing or not fill_back:
            raise NotImplementedError("the reference only builds Renderer(anti_aliasing=True, fill_back=True)")
        self.image_size, self.K, self.orig_size = image_size, K, orig_size
        # upstream defaults (UNVERIFIED recollection, see oracle/nmr.py); the reference overrides them (homan.py:173-176)
        self.light_direction = [0, 1, 0]
        self.light_intensity_direction = 0.5
        self.light_intensity_ambient = 0.5
        self.background_color = [0, 0, 0]
        self._ctx = {}

    def _context(self, faces, num_verts):
        key = (faces.data_ptr(), tuple(faces.shape), num_verts)
        if key not in self._ctx:
            if len(self._ctx) > 8:
                self._ctx.clear()
            self._ctx[key] = ops.SilhouetteContext(faces, num_verts, faces.shape[0], self.image_size, faces.device)
        return self._ctx[key]

    def _K(self, K, batch):
        K = self.K if K is None else K
        return K.repeat(batch, 1, 1) if K.shape[0] == 1 and batch > 1 else K

    def render(self, vertices, faces, textures=None, K=None):
        B = vertices.shape[0]
        sctx = self._context(faces, vertices.shape[1])
        if textures is None:
            textures = torch.ones(B, faces.shape[1], 1, 1, 1, 3, device=vertices.device)
        return ops.render_rgbd(vertices, self._K(K, B).contiguous(), sctx, textures, self.light_direction,
                               self.light_intensity_ambient, self.light_intensity_direction, self.background_color,
                               float(self.orig_size))

    def __call__(self, vertices, faces, textures=None, mode=None, K=None):
        if mode == "silhouettes":
            return self.render(vertices, faces, None, K)[2]
        if mode is None:
            return self.render(vertices, faces, textures, K)
        raise ValueError(f"mode {mode} is not used by the reference")
