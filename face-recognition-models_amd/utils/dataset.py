"""Dataset classes with the reference's names and item contracts (main_code/utils/dataset.py:74-136,
335-360).  They define the INPUT LAYOUT the kernels see -- 112x112 RGB -> ToTensor -> Normalize(0.5, 0.5)
= fp32 CHW in [-1, 1] -- and are deliberately thin: decoding JPEGs is outside the hot path."""
import os

import torch
from torch.utils.data import Dataset


def _load_rgb(path):
    from PIL import Image
    with Image.open(path) as im:
        return im.convert("RGB")


def default_transform(img):
    """ToTensor + Normalize(mean 0.5, std 0.5) (model_utils.py:539-547) without torchvision."""
    import numpy as np
    a = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float()
    return (a / 255.0 - 0.5) / 0.5


def uint8_hwc(img):
    """The image as the decoder left it: uint8 [H, W, 3].  The native train path takes this layout directly -- a quarter
    of the host->device bytes of the fp32 transform; ToTensor + Normalize(0.5, 0.5) then run on the GPU, fused with the
    stem's layout change (frx_input_prep), with bit-identical results (tests/test_gpu_conv.py::test_stem)."""
    import numpy as np
    return torch.from_numpy(np.asarray(img, dtype=np.uint8).copy())


class CASIAwebfaceDataset(Dataset):
    """root_dir/<split>/<identity>/<image>; label = index of the identity folder.  Folders are SORTED
    (upstream uses os.listdir order, which is not reproducible -- SURVEY M9)."""

    def __init__(self, root_dir, split="train", transform=None):
        self.transform = transform or default_transform
        base = os.path.join(root_dir, split)
        if not os.path.exists(base):
            raise FileNotFoundError(f"Directory {base} does not exist")
        self.samples = []
        self.identities = [x for x in sorted(os.listdir(base)) if os.path.isdir(os.path.join(base, x))]
        self.class_to_idx = {name: i for i, name in enumerate(self.identities)}
        self.idx_to_class = {i: name for name, i in self.class_to_idx.items()}
        self.num_of_identities = len(self.identities)
        for ident, label in self.class_to_idx.items():
            d = os.path.join(base, ident)
            self.samples += [(os.path.join(d, f), label) for f in sorted(os.listdir(d))
                             if f.lower().endswith((".jpg", ".jpeg", ".png"))]

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, i):
        path, label = self.samples[i]
        try:
            return self.transform(_load_rgb(path)), label
        except Exception:
            return None          # unreadable image: dropped by custom_collate_fn (dataset.py:127-131 upstream)


class FlatPairDataset(Dataset):
    """pairs of integer image ids (a, b, same) -> ({a}.jpg, {b}.jpg, same)"""

    def __init__(self, pairs, img_dir, transform=None):
        self.pairs, self.img_dir, self.transform = pairs, img_dir, transform or default_transform

    def __len__(self):
        return len(self.pairs)

    def load_id(self, idx):
        return self.transform(_load_rgb(os.path.join(self.img_dir, f"{int(idx)}.jpg")))

    def __getitem__(self, i):
        a, b, same = self.pairs[i]
        return self.load_id(a), self.load_id(b), int(same)
