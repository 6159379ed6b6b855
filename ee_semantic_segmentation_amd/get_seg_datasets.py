"""Datasets for the entry points.  The reference loads PASCAL-VOC through torchvision
(get_seg_datasets.py:33-146, network download) - the download / file IO is out of scope here.
What IS kept: the label convention (void 255 -> index C, get_seg_datasets.py:85), a seeded synthetic
dataset of the same shapes so the train / eval entry points run offline, and the transform chain itself
(:49-86) as ``DevicePreprocess``: decoded uint8 pixels in, normalised fp32 image / int64 target out, computed
on the GPU bit-exactly as Pillow + torchvision do on the CPU."""
import ctypes as C

import torch
from torch.utils.data import Dataset

from . import kernels as K
from ._lib import check, lib


def remap_void(labels, num_classes, void_value=255):
    """get_seg_datasets.py:85: pixels labelled 255 become class index `num_classes`."""
    labels = labels.clone()
    labels[labels == void_value] = num_classes
    return labels


class SyntheticSeg(Dataset):
    """randn images, piecewise-constant labels (32x32 blocks), ~5% void."""

    def __init__(self, n, img_dim, num_classes=21, seed=0):
        self.n, self.dim, self.C, self.seed = n, img_dim, num_classes, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        H = W = self.dim
        X = torch.randn(3, H, W, generator=g)
        blocks = torch.randint(0, self.C, (1, 1, (H + 31) // 32, (W + 31) // 32), generator=g).float()
        y = torch.nn.functional.interpolate(blocks, size=(H, W), mode="nearest")[0].long()
        y[torch.rand(1, H, W, generator=g) < 0.05] = self.C
        return X, y


class LoadDataset:
    """Signature of the reference class (get_seg_datasets.py:19-31, :148-158); returns
    synthetic (train, val, test) sets - VOC loading needs torchvision + a download."""

    def __init__(self, input_dim, target_dim=None, *_, num_classes=21, sizes=(64, 10, 10)):
        self.input_dim, self.C, self.sizes = input_dim, num_classes, sizes

    def get_dataset(self, data_path=None, dataset="voc_seg"):
        return tuple(SyntheticSeg(n, self.input_dim, self.C, seed=s) for s, n in enumerate(self.sizes))


class DevicePreprocess:
    """The reference's test-time transforms (get_seg_datasets.py:72-86) on the device.

    ``image(u8 [H,W,3])  -> float32 [3,h,w]``: Resize (Pillow bilinear, antialiased) -> CenterCrop -> ToTensor ->
    Normalize(mean, std);  ``target(u8 [H,W]) -> int64 [1,h,w]``: Resize (NEAREST, as Pillow does for palette
    images) -> CenterCrop -> the ``ToTensor()*255 -> long -> 255 -> void`` chain as a 256-entry table, truncation
    hazard included.  ``batch(images, targets)`` fills ``[B,3,h,w]`` / ``[B,1,h,w]`` tensors.  Inputs may live on
    the host (pinned or not) or on the device; Pillow's coefficient / index tables are built per source size by the
    library's host helpers and cached.  JPEG/PNG decoding stays with whoever produces the uint8 arrays."""

    MEAN = [.485, .456, .406]
    STD = [.229, .224, .225]

    def __init__(self, input_dim, target_dim=None, num_classes=21, device="cuda"):
        self.input_dim, self.target_dim = input_dim, target_dim or input_dim
        self.C, self.device = num_classes, torch.device(device)
        self.mean = torch.tensor(self.MEAN, dtype=torch.float32, device=self.device)
        self.std = torch.tensor(self.STD, dtype=torch.float32, device=self.device)
        v = torch.arange(256, dtype=torch.uint8).to(torch.float32).div(255)       # CPU arithmetic, as the reference's loader
        x = (v * 255).type(torch.long)
        self.lut = torch.where(x == 255, torch.tensor(num_classes), x).to(self.device)
        self._bil, self._near = {}, {}

    # ---- torchvision glue ---------------------------------------------------------------------------
    @staticmethod
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

    @staticmethod
    def crop_hw(size):
        if isinstance(size, (tuple, list)) and len(size) == 2:
            return int(size[0]), int(size[1])
        s = int(size[0] if isinstance(size, (tuple, list)) else size)
        return s, s

    # ---- Pillow tables (host helpers of libeeseg, cached per (in, out)) --------------------------------
    def _bilinear(self, n_in, n_out):
        key = (n_in, n_out)
        if key not in self._bil:
            ks = lib().eeseg_pil_bilinear_coeffs(n_in, n_out, None, None, 0)
            check(min(ks, 0), "eeseg_pil_bilinear_coeffs")
            bounds = torch.empty((n_out, 2), dtype=torch.int32)
            kk = torch.empty((n_out, ks), dtype=torch.int32)
            check(min(lib().eeseg_pil_bilinear_coeffs(n_in, n_out, C.c_void_p(bounds.data_ptr()), C.c_void_p(kk.data_ptr()), ks), 0),
                  "eeseg_pil_bilinear_coeffs")
            self._bil[key] = (bounds.to(self.device), kk.to(self.device), ks)
        return self._bil[key]

    def _nearest(self, n_in, n_out):
        key = (n_in, n_out)
        if key not in self._near:
            idx = torch.empty(n_out, dtype=torch.int32)
            check(lib().eeseg_pil_nearest_index(n_in, n_out, C.c_void_p(idx.data_ptr())), "eeseg_pil_nearest_index")
            self._near[key] = idx.to(self.device)
        return self._near[key]

    # ---- transforms -----------------------------------------------------------------------------------
    def image(self, img, out=None):
        img = img.to(self.device, non_blocking=True).contiguous()
        assert img.dtype == torch.uint8 and img.dim() == 3
        H, W, ch = img.shape
        rh, rw = self.resized_hw(H, W, self.input_dim)
        dh, dw = self.crop_hw(self.input_dim)
        top, left = int(round((rh - dh) / 2.)), int(round((rw - dw) / 2.))
        hb, hk, hks = self._bilinear(W, rw)
        vb, vk, vks = self._bilinear(H, rh)
        tmp = torch.empty((H, rw, ch), dtype=torch.uint8, device=self.device)
        if out is None:
            out = torch.empty((ch, dh, dw), dtype=torch.float32, device=self.device)
        assert out.shape == (ch, dh, dw) and out.is_contiguous() and out.dtype == torch.float32
        check(lib().eeseg_preprocess_image_u8(K._p(img), H, W, ch, rh, rw, K._p(hb), K._p(hk), hks, K._p(vb), K._p(vk), vks,
                                              top, left, dh, dw, K._p(self.mean), K._p(self.std), K._p(tmp), K._p(out),
                                              K._stream()), "eeseg_preprocess_image_u8")
        return out

    def target(self, lbl, out=None):
        lbl = lbl.to(self.device, non_blocking=True).contiguous()
        assert lbl.dtype == torch.uint8 and lbl.dim() == 2
        H, W = lbl.shape
        rh, rw = self.resized_hw(H, W, self.target_dim)
        dh, dw = self.crop_hw(self.target_dim)
        top, left = int(round((rh - dh) / 2.)), int(round((rw - dw) / 2.))
        if out is None:
            out = torch.empty((1, dh, dw), dtype=torch.int64, device=self.device)
        assert out.shape == (1, dh, dw) and out.is_contiguous() and out.dtype == torch.int64
        check(lib().eeseg_preprocess_label_u8(K._p(lbl), H, W, K._p(self._nearest(H, rh)), K._p(self._nearest(W, rw)), rh, rw,
                                              top, left, dh, dw, K._p(self.lut), K._p(out), K._stream()),
              "eeseg_preprocess_label_u8")
        return out

    def batch(self, images, targets):
        dh, dw = self.crop_hw(self.input_dim)
        th, tw = self.crop_hw(self.target_dim)
        X = torch.empty((len(images), 3, dh, dw), dtype=torch.float32, device=self.device)
        y = torch.empty((len(targets), 1, th, tw), dtype=torch.int64, device=self.device)
        for i, (im, lb) in enumerate(zip(images, targets)):
            self.image(im, X[i])
            self.target(lb, y[i])
        return X, y
