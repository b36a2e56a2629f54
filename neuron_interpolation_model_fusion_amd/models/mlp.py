"""Two-layer MLP (reference: models/mlp.py:7-73). Parameter names match the reference's
checkpoints: level1.1.{weight,bias}, level2.{0|1}.{weight,bias}."""
import torch.nn as nn

from .base_model import BaseModel, WeightRepresentation, LevelRepresentation


class MLPNet(BaseModel):
    captures_may_alias = True  # forward() never writes into a tensor a hooked layer has seen

    def __init__(self, input_size, hidden_size, output_size, use_bias=True, use_activations=True,
                 train_indices=None, val_indices=None):
        super().__init__(train_indices, val_indices)
        self.input_size, self.hidden_size, self.output_size = input_size, hidden_size, output_size
        self.use_bias, self.use_activations = use_bias, use_activations
        first = [nn.Flatten(start_dim=1), nn.Linear(input_size, hidden_size, bias=use_bias)]
        second = [nn.Linear(hidden_size, output_size, bias=use_bias)]
        # with use_activations=False the ReLU opens the NEXT level, so level outputs are pre-activations
        if use_activations:
            first.append(nn.ReLU())
        else:
            second.insert(0, nn.ReLU())
        self.level1 = nn.Sequential(*first)
        self.level2 = nn.Sequential(*second)

    def forward(self, x):
        return self.level2(self.level1(x))

    def forward_until_level(self, x, level_idx=-1):
        x = self.level1(x.view(x.shape[0], -1))
        if level_idx == 0:
            return x
        return self.level2(x)

    def forward_level(self, acts, level_idx):
        return self.level1(acts.view(acts.shape[0], -1)) if level_idx == 0 else self.level2(acts)

    def copy_model(self):
        twin = MLPNet(self.input_size, self.hidden_size, self.output_size, self.use_bias, self.use_activations,
                      self.train_indices, self.val_indices)
        return self._adopt_weights(twin)

    @property
    def get_ordered_trainable_named_layers(self):
        fc1, fc2 = self.level1[1], self.level2[-1]
        return [WeightRepresentation('fc1', fc1.weight, fc1), WeightRepresentation('fc2', fc2.weight, fc2)]

    @property
    def get_ordered_trainable_named_levels(self):
        n1 = 'fc1' + ('_with_relu' if self.use_activations else '')
        n2 = ('relu_with_' if self.use_activations else '') + 'fc2'
        return [LevelRepresentation(n1, self.hidden_size, self.level1),
                LevelRepresentation(n2, self.output_size, self.level2)]
