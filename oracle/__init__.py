"""CPU oracle for the early-exit DeepLabV3 hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it, and only as the checker.  The product
package ``ee_semantic_segmentation_amd`` never imports this package and fails
loudly when its HIP library is missing.

What is restated here and what pins it:

* ``losses_ref``   - BrXEntropyLoss (``my_pixelwise_xentropy.py:11-46``) and the
  raw-logit multi-exit Lovasz (``branchy_seg_losses.py:133-159`` ->
  ``lovaszsoftmax.py:19-31,154-219``).  PINNED: golden vectors generated in this
  container by importing those reference files (``scripts/make_golden.py`` ->
  ``tests/golden/*.npz``).
* ``metrics_ref``  - mIoU / _compute_basics (``compute_mIoU.py:7-36``,
  ``seg_metrics.py:13-28``) PINNED by the reference self-check value
  0.9513888955116272 and generated vectors; img_norm_entropy
  (``eval_br_ent.py:19-36``) PINNED against ``scipy.stats.entropy`` vectors for
  the un-pooled gate; the pooled variants use ``skimage.measure.block_reduce``
  which is absent here -> restated from its documented semantics, PARITY
  UNPINNED for pooled variants.
* ``deeplab_ref``  - ``branchyDeepv3`` (``from_deepv3_new.py:56-155``) over a
  pure ``torch.nn`` restatement of the un-vendored torchvision
  ResNet/DeepLabHead/ASPP.  torchvision is not installed and
  the reference model files are not importable, so the ARCHITECTURE PARITY IS
  UNPINNED beyond torchvision's published parameter counts (42,004,074 /
  60,996,202 incl. aux head), which ``tests/test_oracle_model.py`` reproduces.
  The arithmetic itself is torch CPU fp32 (F.conv2d / F.batch_norm / ...),
  i.e. exactly the ops the reference reaches through torchvision.
"""
