"""Shape predicates and parameter resets (reference: utils/model_utils.py:6-63)."""
import copy

import torch
import torch.nn as nn


def check_if_models_have_same_architecture(model1, model2):
    if model1.__class__ != model2.__class__:
        return False
    for (n1, p1), (n2, p2) in zip(model1.named_parameters(), model2.named_parameters()):
        if n1 != n2 or p1.shape != p2.shape:
            return False
    return True


def module_is_trainable(module):
    return len(list(module.parameters())) > 0


def is_conv_weight(weight):
    return weight.dim() == 4


def is_transformer_activation(activation):
    return activation.dim() == 3


def is_conv_activation(activation):
    return activation.dim() == 4


def reset_parameters(net):
    for child in net.children():
        if hasattr(child, 'reset_parameters'):
            child.reset_parameters()


def soft_reset_parameters(net, eps):
    """param <- (1 - eps) * param + eps * freshly initialised param (model_utils.py:41-56)."""
    fresh = copy.deepcopy(net)
    reset_parameters(fresh)
    fresh_params = dict(fresh.named_parameters())
    with torch.no_grad():
        for name, param in net.named_parameters():
            param.data.mul_(1 - eps).add_(eps * fresh_params[name].data)


def disable_dropout(net):
    for mod in net.modules():
        if isinstance(mod, (nn.Dropout, nn.Dropout1d, nn.Dropout2d, nn.Dropout3d)):
            mod.eval()
