"""Capture, views and importance dispatch shared by the fusion classes
(reference: utils/fusion_utils.py:7-111)."""
import os
import weakref

import torch

from .model_utils import is_conv_weight


def _common_count(models, attr, what):
    counts = [len(getattr(m, attr)) for m in models]
    for idx, c in enumerate(counts[1:]):
        if c != counts[0]:
            raise ValueError(f'The number of {what} of model 1 is different compared to model {idx + 1}:'
                             f' {counts[0]} != {c}')
    return counts[0]


def get_num_layers(models):
    """Common layer count; ValueError on mismatch (fusion_utils.py:7-15)."""
    return _common_count(models, 'get_ordered_trainable_named_layers', 'layers')


def get_num_levels(models):
    """Common level count; ValueError on mismatch (fusion_utils.py:18-26)."""
    return _common_count(models, 'get_ordered_trainable_named_levels', 'levels')


def initialize_activations(model, activation_dataset):
    """One hooked forward of all fusion samples (fusion_utils.py:29-35)."""
    model.remove_hooks()
    model.register_hooks()
    with torch.no_grad():
        model(activation_dataset)


def normalize_activations(activations):
    """Unit L2 norm per neuron (column) (fusion_utils.py:38-40)."""
    return activations / torch.norm(activations, dim=0, keepdim=True)


def normalize_activations_per_sample(activations):
    """Unit L2 norm per sample (row) (fusion_utils.py:43-45)."""
    return activations / torch.norm(activations, dim=1, keepdim=True)


def get_layer_weight(model, layer):
    """Detached copy of a layer's weight; conv weights as (out, in, kh*kw) (fusion_utils.py:48-57)."""
    weight = model.get_ordered_trainable_named_layers[layer].weight.data.detach().clone()
    conv = is_conv_weight(weight)
    return (weight.flatten(2) if conv else weight), conv


def get_layer_width(model, layer_idx):
    return model.get_ordered_trainable_named_layers[layer_idx].weight.data.shape[0]


def get_preactivations_of_layer(model, target_model, layer):
    """(rows, width) pre-activations of both models; conv rows are (b, h, w) (fusion_utils.py:68-82)."""
    _, a = model.get_inputs_and_preactivations(layer_idx=layer)
    _, b = target_model.get_inputs_and_preactivations(layer_idx=layer)
    if a.dim() == 4:
        a = a.permute(1, 0, 2, 3).flatten(1).T
        b = b.permute(1, 0, 2, 3).flatten(1).T
    return a, b


_SCORE_CACHE = weakref.WeakKeyDictionary()  # model -> (key, [score per layer / level])


def _score_cache_key(method, model, X, y, fusion_logic, batch_size):
    """Everything the scores of a model depend on; in-place edits of a tensor bump its _version."""
    tensors = [X] + ([y] if torch.is_tensor(y) else []) + list(model.parameters()) + list(model.buffers())
    return (method, fusion_logic, batch_size, model.training,
            tuple((t.data_ptr(), t._version, tuple(t.shape), t.dtype) for t in tensors))


def compute_neuron_importance_scores(method, model, idx, X, y, fusion_logic, batch_size=50):
    """Per-neuron importance of a layer or level, returned on the CPU (fusion_utils.py:85-111).

    'uniform' is ``ones(width)``; 'conductance' / 'deeplift' / anything else (feature ablation, as in the
    reference's final ``else``) are the attributions restated in utils/neuron_importance.py.

    A fusion asks for every layer (level) of a model in turn (hf_kf.py:527-536, otfusion.py:215-226, alf.py:604-613), and
    the reference runs the model forward and backward once per layer and batch.  Conductance and DeepLIFT scores of
    ALL layers come out of one pass per batch (neuron_importance.get_scores), so the first request computes them all and
    the following ones are served from a per-model record keyed by the data, the weights' versions and the settings
    -- the same numbers (NIF_IMPORTANCE_PER_LAYER=1 keeps one pass per layer).
    """
    assert fusion_logic in ['layer', 'level'], f'Invalid fusion logic: {fusion_logic}'
    if fusion_logic == 'layer':
        reps = model.get_ordered_trainable_named_layers
        layers = [rep.layer for rep in reps]
        widths = [rep.weight.data.shape[0] for rep in reps]
    else:
        reps = model.get_ordered_trainable_named_levels
        layers = [rep.network for rep in reps]
        widths = [rep.output_width for rep in reps]
    layer, width = layers[idx], widths[idx]
    if method == 'uniform':
        return torch.ones(width)
    from .neuron_importance import get_score, get_scores
    name = method if method in ('conductance', 'deeplift') else 'featureablation'
    if name == 'featureablation' or os.environ.get('NIF_IMPORTANCE_PER_LAYER', '0') == '1':
        return get_score(name, model, layer, width, X, y, batch_size=batch_size).cpu()
    key = _score_cache_key(name, model, X, y, fusion_logic, batch_size)
    hit = _SCORE_CACHE.get(model)
    if hit is None or hit[0] != key:
        hit = (key, [s.cpu() for s in get_scores(name, model, layers, widths, X, y, batch_size=batch_size)])
        _SCORE_CACHE[model] = hit
    return hit[1][idx].clone()
