"""Input pipeline.  CPU part: the library's host helpers reproduce Pillow's resampling tables
bit-exactly (checked against Pillow itself on random images, both directions, up- and down-scaling).  GPU part:
the device chain equals the oracle chain (Pillow + torch CPU ops) bit for bit."""
import ctypes as C

import numpy as np
import pytest
import torch
from PIL import Image

SIZES = [(37, 53, 16, 23), (50, 41, 61, 50), (375, 500, 256, 341), (128, 128, 128, 128), (33, 100, 8, 24), (20, 20, 33, 33),
         (281, 500, 513, 912)]


def _tables(n_in, n_out):
    from ee_semantic_segmentation_amd._lib import lib
    ks = lib().eeseg_pil_bilinear_coeffs(n_in, n_out, None, None, 0)
    b = np.zeros((n_out, 2), np.int32)
    k = np.zeros((n_out, ks), np.int32)
    assert lib().eeseg_pil_bilinear_coeffs(n_in, n_out, C.c_void_p(b.ctypes.data), C.c_void_p(k.ctypes.data), ks) == ks
    return b, k


def _apply(a, b, k, axis):
    a = np.moveaxis(a.astype(np.int64), axis, 0)
    out = np.zeros((b.shape[0],) + a.shape[1:], np.uint8)
    for i in range(b.shape[0]):
        x0, n = b[i]
        acc = (1 << 21) + np.tensordot(k[i, :n].astype(np.int64), a[x0:x0 + n], axes=(0, 0))
        out[i] = np.clip(acc >> 22, 0, 255)
    return np.moveaxis(out, 0, axis)


@pytest.mark.parametrize("H,W,oh,ow", SIZES)
def test_host_tables_reproduce_pillow_bilinear(H, W, oh, ow):
    rng = np.random.default_rng(H * 1000 + W)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
    tmp = _apply(img, *_tables(W, ow), axis=1)            # horizontal pass first, 8-bit intermediate
    got = _apply(tmp, *_tables(H, oh), axis=0)
    assert np.array_equal(got, ref)


def test_host_nearest_index_reproduces_pillow_on_palette_images():
    from ee_semantic_segmentation_amd._lib import lib
    rng = np.random.default_rng(5)
    cases = [(50, 61), (500, 912), (375, 513), (281, 513), (1024, 769), (97, 513), (513, 97)]
    cases += [(int(a), int(b)) for a, b in rng.integers(5, 700, (60, 2))]
    for n_in, n_out in cases:
        idx = np.zeros(n_out, np.int32)
        assert lib().eeseg_pil_nearest_index(n_in, n_out, C.c_void_p(idx.ctypes.data)) == 0
        vals = (np.arange(n_in) * 7919 % 251).astype(np.uint8)
        ref = np.asarray(Image.fromarray(vals[None, :].repeat(2, 0), mode="P").resize((n_out, 2), Image.BILINEAR))[0]
        assert np.array_equal(vals[idx], ref), (n_in, n_out)


def test_label_table_follows_the_reference_float_chain():
    """ToTensor()*255 -> long (get_seg_datasets.py:82-84) truncates; the table is built with the SAME float32 ops
    on the CPU so whatever they do to a byte is reproduced.  With IEEE division (torch CPU) v/255*255 comes back
    exactly for all 256 bytes, so here the table is the identity with 255 -> void."""
    from oracle.preprocess_ref import label_lut
    lut = label_lut(21)
    want = torch.arange(256)
    want[255] = 21
    assert torch.equal(lut, want)


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,dim", [(375, 500, 256), (500, 333, 513), (281, 500, (200, 300)), (64, 48, 97)])
def test_device_chain_is_bit_exact_with_pillow_chain(H, W, dim):
    from ee_semantic_segmentation_amd.get_seg_datasets import DevicePreprocess
    from oracle.preprocess_ref import image_chain, target_chain
    rng = np.random.default_rng(H + W)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    lbl = rng.integers(0, 21, (H, W), dtype=np.uint8)
    lbl[rng.random((H, W)) < 0.05] = 255
    pre = DevicePreprocess(dim, num_classes=21)
    X = pre.image(torch.from_numpy(img)).cpu()
    y = pre.target(torch.from_numpy(lbl)).cpu()
    Xr, yr = image_chain(img, dim), target_chain(lbl, dim, 21)
    assert X.shape == Xr.shape and torch.equal(X, Xr)
    assert y.shape == yr.shape and torch.equal(y, yr)
    Xb, yb = pre.batch([torch.from_numpy(img)] * 2, [torch.from_numpy(lbl)] * 2)
    assert torch.equal(Xb[1].cpu(), Xr) and torch.equal(yb[1].cpu(), yr)
