"""CPU restatement of the `neural_renderer` surface the reference touches.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  PARITY UNPINNED (third-party
package hassony2/multiperson `neural_renderer` @ HEAD, not in /root/reference).

Reference call sites restated here:
  nr.projection(vertices, K=, R=, t=, dist_coeffs=, orig_size=)   homan/losses.py:34-41
  nr.renderer.Renderer(image_size=, K=, R=, t=, orig_size=)       homan/losses.py:73-77, homan/homan.py:168-172
  renderer(verts, faces, K=, mode="silhouettes") -> (B,S,S)       homan/losses.py:187
  renderer.render(verts, faces, textures[, K=]) -> (rgb, depth, alpha)   homan/homan.py:391,406,535
  (rgb: flat per-face colours under nr.lighting, ambient + directional; homan/homan.py:173-176 sets the light)
Published algorithm: Kato et al., "Neural 3D Mesh Renderer", CVPR 2018
(projection with OpenCV-style distortion, fill_back face doubling, hard
z-buffer rasteriser at 2x supersampling, vertical flip, 2x2 average pool,
edge-sweep pseudo-gradient).  The rasteriser core is oracle/csrc/nmr_raster.c.
"""
import numpy as np
import torch

from . import clib

DEFAULT_NEAR = 0.1
DEFAULT_FAR = 100.0
DEFAULT_EPS = 1e-3


def projection(vertices, K, R, t, dist_coeffs, orig_size, eps=1e-9):
    """[X,Y,Z] -> [u,v,z]: u,v in [-1,1] (v up), z = camera depth.

    The two 3x3 products are written out as ((a*m0 + b*m1) + c*m2) with one IEEE fp32 operation per product / sum
    (upstream: torch.matmul = cuBLAS, whose rounding no CPU reproduces; torch's CPU matmul is an FMA chain or a plain loop
    depending on the size).  A fixed order makes the projected vertices - and the coverage decisions of the hard rasteriser
    that depend on their last bit - a function of the inputs alone; the HIP kernel (csrc/raster_setup.hip project_vertex) follows
    the same order, bit for bit."""
    Rt = R.expand(vertices.shape[0], 3, 3) if R.shape[0] != vertices.shape[0] else R
    X, Y, Z = vertices[:, :, 0], vertices[:, :, 1], vertices[:, :, 2]
    tt = t.reshape(-1, 1, 3)
    x = ((X * Rt[:, None, 0, 0] + Y * Rt[:, None, 0, 1]) + Z * Rt[:, None, 0, 2]) + tt[:, :, 0]
    y = ((X * Rt[:, None, 1, 0] + Y * Rt[:, None, 1, 1]) + Z * Rt[:, None, 1, 2]) + tt[:, :, 1]
    z = ((X * Rt[:, None, 2, 0] + Y * Rt[:, None, 2, 1]) + Z * Rt[:, None, 2, 2]) + tt[:, :, 2]
    x_ = x / (z + eps)
    y_ = y / (z + eps)
    k1 = dist_coeffs[:, None, 0]
    k2 = dist_coeffs[:, None, 1]
    p1 = dist_coeffs[:, None, 2]
    p2 = dist_coeffs[:, None, 3]
    k3 = dist_coeffs[:, None, 4]
    r2 = x_ ** 2 + y_ ** 2          # r**2 without the sqrt (avoids 0/0 in autograd at the axis)
    radial = 1 + k1 * r2 + k2 * r2 ** 2 + k3 * r2 ** 3
    x__ = x_ * radial + 2 * p1 * x_ * y_ + p2 * (r2 + 2 * x_ ** 2)
    y__ = y_ * radial + p1 * (r2 + 2 * y_ ** 2) + 2 * p2 * x_ * y_
    u = (x__ * K[:, None, 0, 0] + y__ * K[:, None, 0, 1]) + K[:, None, 0, 2]
    v = (x__ * K[:, None, 1, 0] + y__ * K[:, None, 1, 1]) + K[:, None, 1, 2]
    v = orig_size - v
    u = 2 * (u - orig_size / 2.0) / orig_size
    v = 2 * (v - orig_size / 2.0) / orig_size
    return torch.stack([u, v, z], dim=-1)


def vertices_to_faces(vertices, faces):
    """(B,V,3),(B,F,3) int -> (B,F,3,3)."""
    bs, nv = vertices.shape[:2]
    faces = faces.long() + (torch.arange(bs, device=vertices.device) * nv)[:, None, None]
    return vertices.reshape(bs * nv, 3)[faces]


class _RasterizeAlphaDepth(torch.autograd.Function):
    """Hard rasterisation of NDC faces -> (alpha, depth) on an `is` grid (no flip, no pool)."""

    @staticmethod
    def forward(ctx, faces, image_size, near, far, eps):
        f = np.ascontiguousarray(faces.detach().cpu().numpy(), dtype=np.float32)
        B, NF = f.shape[:2]
        idx = np.empty((B, image_size, image_size), np.int32)
        dep = np.empty((B, image_size, image_size), np.float32)
        clib.lib().orc_nmr_face_index_map(clib.fptr(f), B, NF, image_size, near, far,
                                          clib.iptr(idx), clib.fptr(dep))
        ctx.meta = (B, NF, image_size, eps)
        ctx.f = f
        ctx.idx = idx
        alpha = torch.from_numpy((idx >= 0).astype(np.float32))
        return alpha, torch.from_numpy(dep), torch.from_numpy(idx)

    @staticmethod
    def backward(ctx, grad_alpha, grad_depth, _gi):
        B, NF, image_size, eps = ctx.meta
        ga = np.ascontiguousarray(grad_alpha.detach().cpu().numpy(), dtype=np.float32)
        gf = np.zeros((B, NF, 9), np.float32)
        clib.lib().orc_nmr_grad_faces_alpha(clib.fptr(ctx.f), clib.iptr(ctx.idx), clib.fptr(ga),
                                            B, NF, image_size, eps, clib.fptr(gf))
        if grad_depth is not None and bool((grad_depth != 0).any()):
            gd = np.ascontiguousarray(grad_depth.detach().cpu().numpy(), dtype=np.float32)
            clib.lib().orc_nmr_grad_faces_depth(clib.fptr(ctx.f), clib.iptr(ctx.idx), clib.fptr(gd), B, NF,
                                                image_size, clib.fptr(gf))
        return torch.from_numpy(gf).view(B, NF, 3, 3), None, None, None, None


def rasterize_alpha_depth(faces, image_size, anti_aliasing=True, near=DEFAULT_NEAR,
                          far=DEFAULT_FAR, eps=DEFAULT_EPS):
    """(B,NF,3,3) NDC faces -> alpha, depth (B,S,S): 2x SSAA, vertical flip, 2x2 avg-pool."""
    s = image_size * 2 if anti_aliasing else image_size
    alpha, depth, idx = _RasterizeAlphaDepth.apply(faces, s, near, far, eps)
    alpha = alpha.flip(1)
    depth = depth.flip(1)
    if anti_aliasing:
        alpha = torch.nn.functional.avg_pool2d(alpha[:, None], kernel_size=(2, 2))[:, 0]
        depth = torch.nn.functional.avg_pool2d(depth[:, None], kernel_size=(2, 2))[:, 0]
    return alpha, depth, idx


def lighting(faces, textures, intensity_ambient=0.5, intensity_directional=0.5, color_ambient=(1, 1, 1),
             color_directional=(1, 1, 1), direction=(0, 1, 0)):
    """nr.lighting (UNVERIFIED recollection of the upstream package): per-face flat shading of the textures.
    faces (B,NF,3,3) camera-space face vertices, textures (B,NF,t,t,t,3).  light = ambient + directional *
    relu(<normal, direction>), normal = normalize((v0 - v1) x (v2 - v1), eps 1e-5); `direction` is used as given
    (not normalised)."""
    bs, nf = faces.shape[:2]
    light = torch.zeros(bs, nf, 3, dtype=torch.float32)
    if intensity_ambient != 0:
        light = light + intensity_ambient * torch.tensor(color_ambient, dtype=torch.float32)[None, None, :]
    if intensity_directional != 0:
        f = faces.reshape(bs * nf, 3, 3)
        v10 = f[:, 0] - f[:, 1]
        v12 = f[:, 2] - f[:, 1]
        normals = torch.nn.functional.normalize(torch.cross(v10, v12, dim=1), eps=1e-5).reshape(bs, nf, 3)
        d = torch.tensor(direction, dtype=torch.float32)[None, None, :]
        cos = torch.relu((normals * d).sum(2))
        light = light + intensity_directional * torch.tensor(color_directional, dtype=torch.float32)[None, None, :] \
            * cos[:, :, None]
    return textures * light[:, :, None, None, None, :]


def shade_index_map(idx, textures, background_color, anti_aliasing=True):
    """rgb image of a face index map (B,is,is) (no flip yet) under per-face colours: texture_size 1, so every sample
    of a face reads the face's single texel; empty samples read the background; then the vertical flip and the 2x2
    average pool of the other outputs.  -> (B,3,S,S)."""
    B = idx.shape[0]
    col = textures.reshape(B, -1, 3)
    bg = torch.tensor(background_color, dtype=torch.float32)
    ii = torch.as_tensor(idx).long()
    rgb = torch.where((ii >= 0)[..., None], torch.gather(col, 1, ii.clamp(min=0).reshape(B, -1, 1).expand(-1, -1, 3))
                      .reshape(*ii.shape, 3), bg.expand(*ii.shape, 3))
    rgb = rgb.flip(1).permute(0, 3, 1, 2)
    if anti_aliasing:
        rgb = torch.nn.functional.avg_pool2d(rgb, kernel_size=(2, 2))
    return rgb


class Renderer:
    """Subset of nr.renderer.Renderer used by the reference (camera_mode='projection')."""

    def __init__(self, image_size=256, anti_aliasing=True, fill_back=True, K=None, R=None, t=None,
                 dist_coeffs=None, orig_size=1024, near=DEFAULT_NEAR, far=DEFAULT_FAR, **_unused):
        self.image_size = image_size
        self.anti_aliasing = anti_aliasing
        self.fill_back = fill_back
        self.K, self.R, self.t = K, R, t
        if dist_coeffs is None:
            dist_coeffs = torch.zeros(1, 5)
        self.dist_coeffs = dist_coeffs
        self.orig_size = orig_size
        self.near, self.far = near, far
        self.rasterizer_eps = DEFAULT_EPS
        # lighting/background attributes the reference assigns (homan/homan.py:173-176)
        self.light_direction = [0, 1, 0]
        self.light_intensity_direction = 0.5
        self.light_intensity_ambient = 0.5
        self.background_color = [0, 0, 0]

    def _ndc_faces(self, vertices, faces, K, R, t, dist_coeffs, orig_size):
        if self.fill_back:
            faces = torch.cat((faces, faces[:, :, [2, 1, 0]]), dim=1)
        K = self.K if K is None else K
        R = self.R if R is None else R
        t = self.t if t is None else t
        dist_coeffs = self.dist_coeffs if dist_coeffs is None else dist_coeffs
        orig_size = self.orig_size if orig_size is None else orig_size
        v = projection(vertices, K, R, t, dist_coeffs, orig_size)
        return vertices_to_faces(v, faces)

    def render_silhouettes(self, vertices, faces, K=None, R=None, t=None, dist_coeffs=None, orig_size=None):
        f = self._ndc_faces(vertices, faces, K, R, t, dist_coeffs, orig_size)
        alpha, _, _ = rasterize_alpha_depth(f, self.image_size, self.anti_aliasing, self.near, self.far,
                                            self.rasterizer_eps)
        return alpha

    def render(self, vertices, faces, textures=None, K=None, R=None, t=None, dist_coeffs=None, orig_size=None):
        """-> (rgb (B,3,S,S), depth (B,S,S), alpha (B,S,S)).  textures (B,F,1,1,1,3) (the only texture size the
        reference uses, homan/meshutils.py:7-51); None renders white faces (the hot path never reads rgb).  Order as
        upstream (UNVERIFIED): fill_back doubles faces and textures, lighting on the camera-space faces, projection,
        rasterisation."""
        f = self._ndc_faces(vertices, faces, K, R, t, dist_coeffs, orig_size)
        alpha, depth, idx = rasterize_alpha_depth(f, self.image_size, self.anti_aliasing, self.near, self.far,
                                                  self.rasterizer_eps)
        if textures is None:
            textures = torch.ones(faces.shape[0], faces.shape[1], 1, 1, 1, 3)
        faces_l = torch.cat((faces, faces[:, :, [2, 1, 0]]), dim=1) if self.fill_back else faces
        tex = torch.cat((textures, textures), dim=1) if self.fill_back else textures
        tex = lighting(vertices_to_faces(vertices.detach(), faces_l), tex.float(), self.light_intensity_ambient,
                       self.light_intensity_direction, (1, 1, 1), (1, 1, 1), self.light_direction)
        rgb = shade_index_map(idx, tex, self.background_color, self.anti_aliasing)
        return rgb, depth, alpha

    def __call__(self, vertices, faces, textures=None, mode=None, K=None, R=None, t=None,
                 dist_coeffs=None, orig_size=None):
        if mode == "silhouettes":
            return self.render_silhouettes(vertices, faces, K, R, t, dist_coeffs, orig_size)
        if mode is None:
            return self.render(vertices, faces, textures, K, R, t, dist_coeffs, orig_size)
        raise ValueError(f"mode {mode} not supported by the oracle renderer")
