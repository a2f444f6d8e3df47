"""Combined scene topology + per-face colours for the visualisation renders (reference homan/meshutils.py:7-51; colour
table of reference homan/utils/nmr_renderer.py:7-23)."""
import torch

COLORS = {
    "blue": [0.65098039, 0.74117647, 0.85882353], "pink": [0.9, 0.7, 0.7], "mint": [166 / 255.0, 229 / 255.0, 204 / 255.0],
    "mint2": [202 / 255.0, 229 / 255.0, 223 / 255.0], "green": [153 / 255.0, 216 / 255.0, 201 / 255.0],
    "green2": [171 / 255.0, 221 / 255.0, 164 / 255.0], "red": [251 / 255.0, 128 / 255.0, 114 / 255.0],
    "orange": [253 / 255.0, 174 / 255.0, 97 / 255.0], "yellow": [210 / 255.0, 200 / 255.0, 124 / 255.0],
    "white": [1, 1, 1], "gold": [240 / 255, 200 / 255, 0], "grey": [204 / 255, 204 / 255, 204 / 255],
}


def get_faces_and_textures(verts_list, faces_list, color_names=None, colors_list=None):
    """verts_list: [(B,V,3)], faces_list: [(B|1,f,3)] -> faces (1,F,3) long with the meshes' vertex offsets applied,
    textures (1,F,1,1,1,3).  As the reference: a missing colour list with `color_names=None` raises at len(None)
    (:22-25 checks the length first)."""
    if colors_list is None:
        if len(color_names) != len(verts_list):
            raise ValueError(f"Invalid number of colors {len(color_names)} for {len(verts_list)} verts")
        colors_list = [COLORS[name] for name in color_names]
    all_faces, all_textures, offset = [], [], 0
    for verts, faces, colors in zip(verts_list, faces_list, colors_list):
        B = len(verts)
        index_offset = torch.arange(B, device=verts.device) * verts.shape[1] + offset
        offset += verts.shape[1] * B
        faces_repeat = (faces.clone().repeat(B, 1, 1) + index_offset.view(-1, 1, 1)).reshape(-1, 3)
        all_faces.append(faces_repeat.long())
        textures = torch.tensor(colors, dtype=torch.float32, device=verts.device)
        all_textures.append(textures.repeat(faces_repeat.shape[0], 1, 1, 1, 1))
    return torch.cat(all_faces).unsqueeze(0), torch.cat(all_textures).unsqueeze(0)
