"""Oracle: the reference's input transforms (get_seg_datasets.py:49-86) on the CPU.

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.

The reference composes torchvision transforms, and torchvision is not installed here; on PIL images those
transforms are thin wrappers over **Pillow**, which IS installed, so the resampling itself is pinned by Pillow's
own output.  What is restated from torchvision's published behaviour (unpinned) is only the glue:
``Resize(int)`` scales the SHORTER side to `size` and the other to ``int(size * long / short)``;
``Resize((h, w))`` resizes to exactly that; ``CenterCrop`` starts at ``int(round((H - h) / 2.))``;
``ToTensor`` is ``uint8 -> float32 / 255`` in CHW; ``Normalize`` is ``(x - mean) / std``.
"""
import numpy as np
import torch
from PIL import Image

MEAN = [.485, .456, .406]      # get_seg_datasets.py:41-42
STD = [.229, .224, .225]


def resized_hw(h, w, size):
    if isinstance(size, (tuple, list)):
        if len(size) == 2:
            return int(size[0]), int(size[1])
        size = size[0]
    if (w <= h and w == size) or (h <= w and h == size):
        return h, w
    if w < h:
        return int(size * h / w), size
    return size, int(size * w / h)


def crop_hw(size):
    return (int(size[0]), int(size[1])) if isinstance(size, (tuple, list)) and len(size) == 2 else (int(size if not isinstance(size, (tuple, list)) else size[0]),) * 2


def crop_origin(H, W, ch, cw):
    return int(round((H - ch) / 2.)), int(round((W - cw) / 2.))


def image_chain(img_u8_hwc, input_dim):
    """transformations_test (:72-77): Resize -> CenterCrop -> ToTensor -> Normalize.  -> float32 [3,h,w]."""
    H, W = img_u8_hwc.shape[:2]
    rh, rw = resized_hw(H, W, input_dim)
    im = Image.fromarray(np.ascontiguousarray(img_u8_hwc)).resize((rw, rh), Image.BILINEAR)
    ch, cw = crop_hw(input_dim)
    top, left = crop_origin(rh, rw, ch, cw)
    a = np.asarray(im)[top:top + ch, left:left + cw]
    t = torch.from_numpy(np.ascontiguousarray(a)).permute(2, 0, 1).to(torch.float32).div(255)
    mean = torch.tensor(MEAN, dtype=torch.float32).view(3, 1, 1)
    std = torch.tensor(STD, dtype=torch.float32).view(3, 1, 1)
    return t.sub(mean).div(std)


def label_lut(void_index):
    """:82-85 applied to every byte value: ToTensor (/255, float32) -> *255 -> long (TRUNCATION: some values come
    back as v-1) -> 255 becomes the void index."""
    v = torch.arange(256, dtype=torch.uint8).to(torch.float32).div(255)
    x = (v * 255).type(torch.long)
    return torch.where(x == 255, torch.tensor(void_index), x)


def target_chain(lbl_u8_hw, target_dim, void_index=21):
    """transformations_target (:79-86) on a palette ('P') image: Pillow resizes 'P' images with NEAREST whatever
    filter is asked for.  -> int64 [1,h,w]."""
    H, W = lbl_u8_hw.shape
    rh, rw = resized_hw(H, W, target_dim)
    im = Image.fromarray(np.ascontiguousarray(lbl_u8_hw), mode="P").resize((rw, rh), Image.BILINEAR)
    ch, cw = crop_hw(target_dim)
    top, left = crop_origin(rh, rw, ch, cw)
    a = np.asarray(im)[top:top + ch, left:left + cw]
    t = torch.from_numpy(np.ascontiguousarray(a)).unsqueeze(0).to(torch.float32).div(255)
    x = (t * 255).type(torch.long)
    return torch.where(x == 255, torch.tensor(void_index), x)
