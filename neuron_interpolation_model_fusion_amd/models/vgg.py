"""VGG for 32x32 inputs, organised in levels (reference: models/vgg.py:7-170).

One level = everything up to and including one conv (+BN) (+ReLU); pooling and -- with
use_activations=False -- the previous conv's ReLU open the next level, so that level
outputs are pre-activations.  Module order inside each level matches the reference so
its state_dicts load unchanged.
"""
import torch
import torch.nn as nn

from .base_model import BaseModel, WeightRepresentation, LevelRepresentation
from .conv import GemmConv2d


class VGG(BaseModel):
    captures_may_alias = True  # forward() never writes into a tensor a hooked layer has seen
    cfg = {
        'VGG11': [64, 'M', 128, 'M', 256, 256, 'M', 512, 512, 'M', 512, 512, 'M'],
        'VGG11_3/2': [64, 'M', 192, 'M', 384, 384, 'M', 768, 768, 'M', 768, 768, 'M'],
        'VGG11_quad': [64, 'M', 512, 'M', 1024, 1024, 'M', 2048, 2048, 'M', 2048, 512, 'M'],
        'VGG11_doub': [64, 'M', 256, 'M', 512, 512, 'M', 1024, 1024, 'M', 1024, 512, 'M'],
        'VGG11_half': [64, 'M', 64, 'M', 128, 128, 'M', 256, 256, 'M', 256, 512, 'M'],
        'VGG13': [64, 64, 'M', 128, 128, 'M', 256, 256, 'M', 512, 512, 'M', 512, 512, 'M'],
        'VGG16': [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M'],
        'VGG19': [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M',
                  512, 512, 512, 512, 'M'],
    }

    def __init__(self, vgg_name, num_classes, input_shape, batch_norm=False, use_bias=True, use_activations=True,
                 train_indices=None, val_indices=None):
        super().__init__(train_indices, val_indices)
        if vgg_name not in self.cfg:
            raise ValueError(f"VGG configuration '{vgg_name}' not found.")
        self.vgg_name, self.num_classes, self.input_shape = vgg_name, num_classes, input_shape
        self.use_batch_norm, self.use_bias, self.use_activations = batch_norm, use_bias, use_activations

        levels, pending = nn.ModuleList(), []
        channels, relu_owed = input_shape[0], False
        for item in self.cfg[vgg_name]:
            if relu_owed:
                pending.append(nn.ReLU())
                relu_owed = False
            if item == 'M':
                pending.append(nn.MaxPool2d(kernel_size=2, stride=2))
                continue
            pending.append(GemmConv2d(channels, item, kernel_size=3, padding=1, bias=use_bias))
            if batch_norm:
                pending.append(nn.BatchNorm2d(item))
            if use_activations:
                pending.append(nn.ReLU())
            else:
                relu_owed = True
            levels.append(nn.Sequential(*pending))
            pending, channels = [], item
        if relu_owed:
            pending.append(nn.ReLU())
        self.feature_levels = levels

        # classifier input size from a dry run (the reference draws the dummy with torch.randn too,
        # vgg.py:42-48, so seeded construction consumes the same amount of RNG state)
        with torch.no_grad():
            probe = torch.randn((1, *input_shape))
            for level in self.feature_levels:
                probe = level(probe)
            for mod in pending:
                probe = mod(probe)
            flat = probe.flatten(1).shape[1]
        self.classifier = nn.Sequential(*pending, nn.Flatten(start_dim=1),
                                        nn.Linear(flat, num_classes, bias=use_bias))

    def forward_features(self, x):
        for level in self.feature_levels:
            x = level(x)
        return x

    def forward(self, x):
        return self.classifier(self.forward_features(x))

    def forward_until_level(self, x, level_idx=-1):
        if level_idx == -1:
            return self.forward(x)
        if level_idx > len(self.get_ordered_trainable_named_levels):
            raise IndexError(f'level_idx {level_idx} is out of bounds. Model has '
                             f'{len(self.get_ordered_trainable_named_levels)} levels (including classifier).')
        for idx, level in enumerate(self.feature_levels):
            x = level(x)
            if idx == level_idx:
                return x
        return self.classifier(x)

    def forward_level(self, acts, level_idx):
        if level_idx < len(self.feature_levels):
            return self.feature_levels[level_idx](acts)
        return self.classifier(acts)

    def copy_model(self):
        twin = VGG(self.vgg_name, self.num_classes, self.input_shape, batch_norm=self.use_batch_norm,
                   use_bias=self.use_bias, use_activations=self.use_activations)
        return self._adopt_weights(twin)

    @property
    def get_ordered_trainable_named_layers(self):
        reps, convs = [], 0
        for level in self.feature_levels:
            for mod in level:
                if isinstance(mod, nn.Conv2d):
                    reps.append(WeightRepresentation(f'features.conv{convs}', mod.weight, mod))
                    convs += 1
                if self.use_batch_norm and isinstance(mod, nn.BatchNorm2d):
                    reps.append(WeightRepresentation(f'features.bn{convs}', mod.weight, mod))
        head = self.classifier[-1]
        reps.append(WeightRepresentation('classifier.fc', head.weight, head))
        return reps

    @property
    def get_ordered_trainable_named_levels(self):
        reps = []
        for idx, level in enumerate(self.feature_levels):
            conv = [mod for mod in level if isinstance(mod, nn.Conv2d)][-1]
            reps.append(LevelRepresentation(f'level_{idx}', conv.out_channels, level))
        reps.append(LevelRepresentation('classifier', self.num_classes, self.classifier))
        return reps
