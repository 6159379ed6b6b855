"""``from_deepv3`` - the older module name the evaluation scripts import
(ee_dnn_op.py:3, ee_dnn_op_ne.py:3).  Its branchyDeepv3 (from_deepv3.py:28-125) has the
same forward as from_deepv3_new.py with hard-coded 21-class DeepLabHead branches."""
from .from_deepv3_new import ExitLogits, branchyDeepv3, get_base_model, upsample_logits  # noqa: F401
