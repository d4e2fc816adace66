"""CPU oracle for the PyLC segmentation hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch (CPU, fp32) restatement of the reference's
training / inference step (U-Net and DeepLabV3+ forward, MultiLoss, backward,
clip + AdamW).  It is written functionally over a flat ``{name: tensor}`` state
dict instead of ``nn.Module`` trees so that it shares no structure with the
product code in ``pylc_amd/``.

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- as the checker / reported baseline,
never as the thing measured or shipped.  Nothing under ``pylc_amd/`` imports it.

Parity pinning: the reference ships no tests or golden vectors for this path
(SURVEY.md section 4).  The oracle is therefore pinned against outputs of the
reference itself, generated in the build container by
``tests/golden/make_golden.py`` (imports /root/reference, asserts
oracle == reference, writes the fixtures under ``tests/golden/``).
"""
from .nets import (deeplab_forward, unet_forward, init_state, formula_state,  # noqa: F401
                   state_spec)
from .loss import multiloss, ce_loss, dice_loss, focal_loss  # noqa: F401
from .step import normalize_image, train_step, eval_step, test_step, make_optimizer, calibrate_bn, StepConfig  # noqa: F401
from .metrics import weighted_jaccard  # noqa: F401
from .stitch import split_tiles, stitch_scores, stitch_classes, colourize_resize  # noqa: F401
from .driver import run_training  # noqa: F401
