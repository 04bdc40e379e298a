"""Drop-in replacement for the PCGmix branch of the reference's ``augmentations2d.augment``
(augmentations2d.py:267, ``durratiomixup`` branch :397-427; called from train_model.py:505).

Also the mask variants ``durmixcutout(t,f)``, ``durmixtimemask(t)``, ``durmixfreqmask(f)``
(augmentations2d.py:286-395): the same splice followed by a zeroed rectangle, fused into the same
kernel launch as one extra predicate.

Spectrogram batches are (B, 1, F, W); the four heart states are ranges of the last (time)
axis and ``frames`` holds their boundaries in spectrogram columns.  The splice is the 1D one
applied to every frequency row, so the same kernel runs with C = F rows and T = W columns
(augmentations2d.py:206-221).  alpha is fixed to 1 and only same-label partners exist in 2D
(augmentations2d.py:410-411).  ``(saloptenv)durratiomixup`` / ``(saloptsum)durratiomixup`` place the
shorter state inside the longer one by saliency (augmentations2d.py:125-204, 416-423): maps from the
frozen ResNet9-2D 'base' checkpoint through ``saliency.get_saliency_maps(dim=2)``, then the 1D
displacement search and offset splice.
"""
from __future__ import annotations

from . import hostprep
from .augmentations import (_as_numpy_frames, _check_data, apply_plan, gate_passes,
                            labels_from_ohe, splice_plain)


def augment(args, data, target_ohe, frames, wav, step_counter, model, device, RESULTS_ARGS,
            host_labels=None):
    method = args.method
    step = int(step_counter.count)
    if hostprep.select_method(method, is2d=True) is None:
        return data, target_ohe, [], None
    _check_data(data, 4)
    B, Cc, F, W = data.shape
    recipe = hostprep.plain_recipe(method, True)
    if recipe is not None and B > 0:              # durratiomixup: one library call
        if not gate_passes(recipe, method, step, data.device.index):
            return data, target_ohe, [], None
        out, mix = splice_plain(recipe, data.view(B, Cc * F, W), host_labels, frames, step,
                                target_ohe=target_ohe)
        return out.view(B, Cc, F, W), target_ohe, mix, None
    frames_np = _as_numpy_frames(frames)
    labels = (lambda: labels_from_ohe(target_ohe)) if host_labels is None else host_labels
    plan = hostprep.make_plan(method, labels, frames_np, wav, step, B, Cc * F, is2d=True, n_cols=W)
    if not plan.fired:
        return data, target_ohe, [], None
    sal = None
    if plan.salopt_mode is not None:
        # '(saloptenv)durratiomixup' / '(saloptsum)…' on spectrograms (augmentations2d.py:416-423):
        # the frozen ResNet9-2D's input gradient -> (B, W) maps (saliency.get_saliency_maps, dim=2),
        # then the same displacement search and offset splice as in 1D, with C = F rows, T = W columns
        from . import saliency
        sal = saliency.get_saliency_maps(args, device, data, target_ohe, frames_np, dim=2)
    out = apply_plan(plan, data.view(B, Cc * F, W), frames_np, sal).view(B, Cc, F, W)
    return out, target_ohe, plan.mix, None
