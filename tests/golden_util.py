"""Load a golden fixture (tests/golden/*.npz) into torch tensors grouped by prefix."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(case):
    z = np.load(os.path.join(GOLDEN, f"{case}.npz"))
    out = {"in": {}, "sd": {}, "train": {}, "grad": {}, "sd1": {}, "eval": {}, "adam1": {}, "sd2": {}}
    out["meta"] = json.loads(bytes(z["meta_json"]).decode())
    out["loss"] = torch.from_numpy(np.asarray(z["loss"]))
    out["loss2"] = torch.from_numpy(np.asarray(z["loss2"]))
    out["gradnorm"] = torch.from_numpy(np.asarray(z["gradnorm"])) if "gradnorm" in z.files else None
    for k in z.files:
        if "." not in k:
            continue
        grp, name = k.split(".", 1)
        if grp in out and grp != "meta":
            out[grp][name] = torch.from_numpy(np.asarray(z[k]))
    return out


ALL_CASES = ["unet2d_f4", "unet2d_f4_o2_dil2", "unet3d_f4", "unet3d_f4_interp", "siam_f4_concat", "siam_f4_max",
             "siam_f4_corr", "siam_f4_control", "mo3d_f4_interp", "mo3d_f4_convT", "attention_f4", "unet_v0_f4", "baby_f4",
             "mo3d_f4_trainer_convT", "mo3d_f4_trainer_interp"]
