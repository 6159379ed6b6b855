"""Datasets for the entry points.  The reference loads PASCAL-VOC through torchvision
(get_seg_datasets.py:33-146, network download) - out of scope here (dataset IO).
What IS kept: the label convention (void 255 -> index C, get_seg_datasets.py:85) and a
seeded synthetic dataset of the same shapes so the train / eval entry points run offline."""
import torch
from torch.utils.data import Dataset


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
