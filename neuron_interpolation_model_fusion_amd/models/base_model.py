"""Model abstraction the fusion classes sit on (reference: models/base_model.py:10-133).

The API is kept verbatim because it is the boundary of the hot path: the fusion
algorithms only ever touch a model through the members defined here.
"""
import contextlib
import os
from abc import ABC, abstractmethod
from typing import NamedTuple

import torch
import torch.nn as nn

from ..utils.model_utils import is_conv_weight


class WeightRepresentation(NamedTuple):
    """A trainable layer: (name, weight tensor, module).  reference base_model.py:10-14"""
    name: str
    weight: torch.Tensor
    layer: nn.Module


class LevelRepresentation(NamedTuple):
    """A group of layers fused as one unit by ALF.  reference base_model.py:17-21"""
    name: str
    output_width: int
    network: nn.Module


# copy_model() builds a fresh instance and loads the weights into it (models/vgg.py:120-125 ...): the random
# initialisation of that instance is thrown away, yet it is most of the host time of a small fusion (27 of 69 ms for two
# 784-8192-10 MLPs).  The matching algorithms construct their copies under `twin_copy()`.  By DEFAULT that context does
# nothing, so torch's generator is consumed exactly as the reference consumes it and whatever is seeded downstream in
# the same process (an ALF run, fine-tuning, the next CLI loop iteration) stays draw-for-draw identical.  The fast mode
# (`NIF_FAST_TWIN_COPY=1` in the environment, or `twin_copy.replay_init_draws = False`) skips the discarded
# initialisation: same fused weights, different generator state afterwards.
_RANDOM_INITS = ('kaiming_uniform_', 'kaiming_normal_', 'uniform_', 'normal_', 'xavier_uniform_', 'xavier_normal_',
                 'trunc_normal_')


@contextlib.contextmanager
def twin_copy():
    """Context for copy_model() calls whose random initialisation is overwritten at once.  In the fast mode the
    initialisers leave the (uninitialised) storage alone and torch's generator is not advanced; it swaps functions of
    torch.nn.init for the duration of the block, so it is not for use while another thread constructs modules."""
    if twin_copy.replay_init_draws:
        yield
        return
    saved = {name: getattr(nn.init, name) for name in _RANDOM_INITS}
    try:
        for name in _RANDOM_INITS:
            setattr(nn.init, name, lambda tensor, *args, **kwargs: tensor)
        yield
    finally:
        for name, fn in saved.items():
            setattr(nn.init, name, fn)


twin_copy.replay_init_draws = os.environ.get('NIF_FAST_TWIN_COPY', '0') != '1'


class BaseModel(nn.Module, ABC):
    # True on classes whose forward never writes into a tensor a hooked layer has seen (see register_hooks)
    captures_may_alias = False

    def __init__(self, train_indices=None, val_indices=None):
        super().__init__()
        # layer name -> captured input / output of the last hooked forward
        self.layer_inputs = {}
        self.layer_outputs = {}
        self.handles = {}
        self.train_indices = train_indices
        self.val_indices = val_indices

    def _adopt_weights(self, twin):
        """What `twin.load_state_dict(self.state_dict()); twin.to(self.device)` leaves behind (models/vgg.py:120-125,
        mlp.py, vit.py copy_model), without its two trips over PCIe: the freshly built twin lives on the host, so the
        reference's sequence copies every weight device -> host (synchronising) and back.  Here every parameter and
        buffer of the twin is re-pointed at a device clone of ours."""
        mine = dict(self.named_parameters())
        mine.update(dict(self.named_buffers()))
        theirs = dict(twin.named_parameters())
        theirs.update(dict(twin.named_buffers()))
        if mine.keys() != theirs.keys():
            raise RuntimeError(f'copy_model: state mismatch {sorted(mine.keys() ^ theirs.keys())}')
        with torch.no_grad():
            for name, t in theirs.items():
                src = mine[name]
                if src.shape != t.shape:
                    raise RuntimeError(f'copy_model: size mismatch for {name}: {tuple(src.shape)} vs {tuple(t.shape)}')
                t.data = src.detach().clone()
        return twin.to(self.device)  # anything that is not a parameter or buffer follows; a no-op otherwise

    # -- structure ---------------------------------------------------------------
    @property
    @abstractmethod
    def get_ordered_trainable_named_layers(self):
        """list[WeightRepresentation] in application order (base_model.py:42-46)."""

    @property
    @abstractmethod
    def get_ordered_trainable_named_levels(self):
        """list[LevelRepresentation] in application order (base_model.py:54-58)."""

    @abstractmethod
    def forward_until_level(self, x, level_idx=-1):
        """Output of level `level_idx` (-1: the whole model) (base_model.py:48-52)."""

    @abstractmethod
    def copy_model(self):
        """Same architecture, same weights, no hooks (base_model.py:60-64)."""

    @property
    def device(self):
        return next(self.parameters()).device

    # -- weight write-back (base_model.py:66-79) ----------------------------------
    def forward_level(self, acts, level_idx):
        """Output of level ``level_idx`` given the output of level ``level_idx - 1`` (the model input for level
        0): ``forward_until_level(x, l) == forward_level(forward_until_level(x, l - 1), l)``.  Not part of the
        reference's API; lets callers walk the levels once instead of re-running the prefix (alf.py:190 does
        O(levels^2) forwards).  Models that cannot offer it leave this unimplemented."""
        raise NotImplementedError

    def set_weight_at_layer(self, trainable_layer_idx, new_weight):
        layer = self.get_ordered_trainable_named_layers[trainable_layer_idx].layer
        current = layer.weight.data
        if is_conv_weight(current):
            new_weight = new_weight.view_as(current)
        with torch.no_grad():
            layer.weight.data = new_weight.to(self.device)

    def set_bias_at_layer(self, trainable_layer_idx, new_bias):
        layer = self.get_ordered_trainable_named_layers[trainable_layer_idx].layer
        with torch.no_grad():
            layer.bias.data = new_bias.to(self.device)

    # -- activation capture (base_model.py:103-133) --------------------------------
    def register_hooks(self):
        names = {rep.layer: rep.name for rep in self.get_ordered_trainable_named_layers}
        # The reference clones every captured tensor (base_model.py:110-118), and so does this class by default.  A
        # clone only matters if something writes into the tensor after the hook has seen it (an in-place module, or a
        # functional in-place op in forward()).  A model class whose forward is known to do neither sets
        # `captures_may_alias = True` (VGG, MLPNet, ViT here): its captures then alias the forward's own tensors -- no
        # copy pass -- and the tensors' version counters are kept, so that a write that happens anyway is reported
        # when the capture is read instead of silently corrupting costs and refits.
        alias = bool(getattr(type(self), 'captures_may_alias', False)) and \
            not any(getattr(m, 'inplace', False) for m in self.modules())
        self._capture_versions = {}

        def capture(module, inputs, output):
            name = names[module]
            x, y = inputs[0].detach(), output.detach()
            if alias:
                self._capture_versions[name] = (x._version, y._version)
            self.layer_inputs[name] = x if alias else x.clone()
            self.layer_outputs[name] = y if alias else y.clone()

        for rep in self.get_ordered_trainable_named_layers:
            self.handles[rep.name] = rep.layer.register_forward_hook(capture)

    def remove_hooks(self):
        for h in self.handles.values():
            h.remove()
        self.handles.clear()
        self.layer_inputs.clear()
        self.layer_outputs.clear()
        getattr(self, '_capture_versions', {}).clear()

    def get_inputs_and_preactivations(self, layer_idx):
        name = self.get_ordered_trainable_named_layers[layer_idx].name
        x, y = self.layer_inputs[name], self.layer_outputs[name]
        seen = getattr(self, '_capture_versions', {}).get(name)
        if seen is not None and seen != (x._version, y._version):
            raise RuntimeError(f'{type(self).__name__}: the captured activations of layer {name!r} were modified in place '
                               f'after the hook saw them; set captures_may_alias = False on the class')
        return x, y
