"""Checkpoint and evaluation compatibility (reference: utils/run_utils.py:17-72).

The reference saves whole modules (``torch.save(model, path)``) whose pickles name its classes by
their module path (``models.mlp.MLPNet`` ...).  ``add_safe_globals`` registers this package's
classes under those paths as well as their own, so ``torch.load(path, weights_only=True)`` turns
a reference checkpoint into one of our models (same attribute and parameter names) and models
saved from here load back the same way.  ``eval_model`` reports the reference's metrics without
torchmetrics (the confusion matrix is a bincount).
"""
import warnings

import torch
import torch.nn as nn

from ..models.mlp import MLPNet
from ..models.vgg import VGG
from ..models.conv import GemmConv2d
from ..models.vit import MultiHeadSelfAttention, InputEmbedder, TransformerEncoder, ViT

# reference module path -> classes it pickles under that path (run_utils.py:9-14)
_REFERENCE_PATHS = {
    'models.mlp': [MLPNet],
    'models.vgg': [VGG],
    'models.vit': [MultiHeadSelfAttention, InputEmbedder, TransformerEncoder, ViT],
}

_TORCH_CLASSES = [
    nn.Linear, nn.Dropout, nn.Conv2d, nn.MaxPool2d, nn.AvgPool2d, nn.Dropout2d, nn.BatchNorm2d, nn.Parameter,
    nn.TransformerEncoderLayer, nn.MultiheadAttention, nn.modules.linear.NonDynamicallyQuantizableLinear,
    nn.LayerNorm, nn.Sequential, nn.ModuleList, nn.ReLU, nn.GELU, nn.Flatten, nn.AdaptiveAvgPool2d,
    nn.AdaptiveMaxPool2d, set, GemmConv2d,
]


def add_safe_globals():
    """Allow-list for ``torch.load(..., weights_only=True)`` of whole-module checkpoints."""
    entries = list(_TORCH_CLASSES)
    for module_path, classes in _REFERENCE_PATHS.items():
        for cls in classes:
            entries.append(cls)                                        # saved from this package
            entries.append((cls, f'{module_path}.{cls.__qualname__}'))  # saved by the reference
    torch.serialization.add_safe_globals(entries)


def load_model(path, device='cpu'):
    """``torch.load`` of a whole-module checkpoint with the safe loader only."""
    add_safe_globals()
    return torch.load(path, weights_only=True, map_location=device).to(device)


def confusion_matrix(predictions, targets, num_classes):
    """rows = true class, columns = predicted class (torchmetrics' multiclass convention)."""
    flat = targets.long() * num_classes + predictions.long()
    return torch.bincount(flat, minlength=num_classes * num_classes).reshape(num_classes, num_classes)


def eval_model(model, test_dataloader, device, num_classes=10):
    """loss / accuracy / macro-F1 / weighted one-vs-rest ROC-AUC / confusion matrix over a dataloader
    (run_utils.py:33-72)."""
    from sklearn.exceptions import UndefinedMetricWarning
    from sklearn.metrics import accuracy_score, f1_score, roc_auc_score
    criterion = torch.nn.CrossEntropyLoss()
    model.eval()
    outputs, targets = [], []
    with torch.no_grad():
        for data, target in test_dataloader:
            outputs.append(model(data.to(device)).float().cpu())
            targets.append(torch.as_tensor(target).reshape(-1))
    outputs, targets = torch.cat(outputs), torch.cat(targets).long()
    predictions = outputs.argmax(dim=1)
    onehot = torch.zeros_like(outputs).scatter_(1, targets.view(-1, 1), 1)
    with warnings.catch_warnings():
        warnings.filterwarnings('ignore', category=UndefinedMetricWarning)
        roc_auc = roc_auc_score(onehot.numpy(), torch.softmax(outputs, dim=1).numpy(), average='weighted',
                                multi_class='ovr')
    return {
        'loss': criterion(outputs, targets).item(),
        'accuracy': accuracy_score(targets, predictions),
        'f1': f1_score(targets, predictions, average='macro'),
        'roc_auc': roc_auc,
        'confusion_matrix': confusion_matrix(predictions, targets, num_classes),
    }
