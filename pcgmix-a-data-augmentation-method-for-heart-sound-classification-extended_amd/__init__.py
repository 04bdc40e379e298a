"""MI355X-native PCGmix hot path (augmentation -> log-mel -> train step).

Drop-in for the reference's per-batch augmentation call
``augmentations.augment(args, data, target_ohe, frames, wav, step_counter, model,
device, RESULTS_ARGS)`` (reference augmentations.py:698, called at
train_model.py:504-507).  The O(B*C*T) work runs in hand-written HIP kernels for
gfx950 behind the C ABI declared in ``include/pcgmix_hip.h``.
"""
from . import (_lib, augmentations, augmentations2d, dataloader_physionet, frontend, hostprep, models, models2d,  # noqa: F401
               saliency, synthetic, train_model)
