"""Combined scene topology + per-face colours for the visualisation renders.

`get_faces_and_textures` has the call surface of reference homan/meshutils.py:7-51 (it is what `HOMan.__init__` calls to
build `model.faces` / `model.textures`, homan.py:199-219); the palette holds the colours of reference
homan/utils/nmr_renderer.py:7-23 (8-bit RGB triplets, "pink" excepted)."""
import torch

_PALETTE_8BIT = {"blue": (166, 189, 219), "mint": (166, 229, 204), "mint2": (202, 229, 223), "green": (153, 216, 201),
                 "green2": (171, 221, 164), "red": (251, 128, 114), "orange": (253, 174, 97), "yellow": (210, 200, 124),
                 "white": (255, 255, 255), "gold": (240, 200, 0), "grey": (204, 204, 204)}
COLORS = {name: [c / 255.0 for c in rgb] for name, rgb in _PALETTE_8BIT.items()}
COLORS["pink"] = [0.9, 0.7, 0.7]


def _resolve_colours(n_meshes, color_names, colors_list):
    if colors_list is not None:
        return colors_list
    if len(color_names) != n_meshes:       # (color_names=None fails here with a TypeError, as in the reference)
        raise ValueError(f"Invalid number of colors {len(color_names)} for {n_meshes} verts")
    return [COLORS[name] for name in color_names]


def get_faces_and_textures(verts_list, faces_list, color_names=None, colors_list=None):
    """verts_list: [(B_k, V_k, 3)], faces_list: [(1 | B_k, f_k, 3)] -> (faces (1, F, 3) int64, textures (1, F, 1, 1, 1, 3)).
    Mesh k contributes B_k copies of its topology, copy i indexing vertices [offset_k + i * V_k, offset_k + (i + 1) * V_k)
    of the concatenated vertex array, in one flat colour."""
    colours = _resolve_colours(len(verts_list), color_names, colors_list)
    faces_out, tex_out, offset = [], [], 0
    for verts, faces, rgb in zip(verts_list, faces_list, colours):
        copies, nv = verts.shape[0], verts.shape[1]
        shift = offset + nv * torch.arange(copies, device=verts.device)
        topo = (faces.repeat(copies, 1, 1) + shift[:, None, None]).reshape(-1, 3).long()
        faces_out.append(topo)
        tex_out.append(torch.tensor(rgb, dtype=torch.float32, device=verts.device).expand(topo.shape[0], 1, 1, 1, 3))
        offset += nv * copies
    return torch.cat(faces_out)[None], torch.cat(tex_out)[None].contiguous()
