"""`joint_fit.pt` read / write with the reference's on-disk contract (reference fit_vid_dataset.py:365-372 save,
:322-338 resume; the keys downstream consumers read are pinned by homan/postprocess.py:16-77).

The file is `torch.save({"state_dict": {...}})` of the model's state_dict on CPU, contiguous, without the MANO layer's
own buffers ("mano_model" in the key).  A file written by the reference loads here and vice versa: parameter and buffer
names are the reference's (tests/test_oracle_golden.py::test_state_dict_keys_cover_reference), and resuming goes through
`load_state_dict(strict=False)` exactly as reference homan/jointopt.py:126-127.
"""
import torch


def joint_fit_state(model):
    """:366-371: the dict stored under "state_dict"."""
    return {key: val.detach().contiguous().cpu() for key, val in model.state_dict().items() if "mano_model" not in key}


def save_joint_fit(model, path):
    """:365-372."""
    torch.save({"state_dict": joint_fit_state(model)}, path)


def load_joint_fit(path, device="cuda"):
    """:331-336: the state_dict of a previous fit, moved to the device, ready for
    `optimize_hand_object(..., state_dict=...)` / `build_model(..., state_dict=...)`."""
    state_dict = torch.load(path, map_location="cpu")["state_dict"]
    return {key: val.to(device) for key, val in state_dict.items()}
