"""Model abstraction the fusion classes sit on (reference: models/base_model.py:10-133).

The API is kept verbatim because it is the boundary of the hot path: the fusion
algorithms only ever touch a model through the members defined here.
"""
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


class BaseModel(nn.Module, ABC):

    def __init__(self, train_indices=None, val_indices=None):
        super().__init__()
        # layer name -> captured input / output of the last hooked forward
        self.layer_inputs = {}
        self.layer_outputs = {}
        self.handles = {}
        self.train_indices = train_indices
        self.val_indices = val_indices

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

        def capture(module, inputs, output):
            name = names[module]
            self.layer_inputs[name] = inputs[0].detach().clone()
            self.layer_outputs[name] = output.detach().clone()

        for rep in self.get_ordered_trainable_named_layers:
            self.handles[rep.name] = rep.layer.register_forward_hook(capture)

    def remove_hooks(self):
        for h in self.handles.values():
            h.remove()
        self.handles.clear()
        self.layer_inputs.clear()
        self.layer_outputs.clear()

    def get_inputs_and_preactivations(self, layer_idx):
        name = self.get_ordered_trainable_named_layers[layer_idx].name
        return self.layer_inputs[name], self.layer_outputs[name]
