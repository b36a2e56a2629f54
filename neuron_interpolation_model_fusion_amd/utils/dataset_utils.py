"""Fusion-sample selection (reference: utils/dataset_utils.py:212-282).

The reference's loaders download MNIST / CIFAR / Tiny-ImageNet through torchvision and Hugging Face; neither is
available offline, and they are not part of the matching path.  What the fusion CLIs need from them is a dataset
object with ``targets`` and integer indexing; ``TensorVisionDataset`` gives that shape to tensors that are already on
disk (``torch.save({'data': X, 'targets': y}, path)``) or to seeded synthetic inputs.
"""
import numpy as np
import torch


class TensorVisionDataset(torch.utils.data.Dataset):
    """(data[i], int(targets[i])) with a ``targets`` list, like a torchvision dataset after its transform."""

    def __init__(self, data, targets):
        self.data = data
        self.targets = [int(t) for t in targets]

    def __len__(self):
        return self.data.shape[0]

    def __getitem__(self, i):
        return self.data[i], self.targets[i]

    @staticmethod
    def load(path):
        blob = torch.load(path, weights_only=True, map_location='cpu')
        return TensorVisionDataset(blob['data'].float(), blob['targets'])

    @staticmethod
    def synthetic(input_shape, num_classes, size, seed=0):
        g = torch.Generator().manual_seed(seed)
        return TensorVisionDataset(torch.randn(size, *input_shape, generator=g), torch.arange(size) % num_classes)


def sample_balanced_subset(original_dataset, indices_to_use=None, batch_size=400):
    """Class-balanced subset of ``batch_size`` samples (dataset_utils.py:212-270): per class a numpy shuffle of its
    positions (global numpy stream, class order ascending), ``batch_size // classes`` each, the remainder dealt
    round-robin.  The positions index ``original_dataset`` itself, as in the reference."""
    if indices_to_use is None:
        indices_to_use = np.arange(len(original_dataset))
    if batch_size > len(indices_to_use):
        raise ValueError(f"Batch size {batch_size} is larger than the number of available indices {len(indices_to_use)}.")
    if type(original_dataset.targets) is list:
        targets = np.array(original_dataset.targets)[indices_to_use]
    else:
        targets = original_dataset.targets.numpy()[indices_to_use]
    classes = np.unique(targets).tolist()
    per_class = {c: np.where(targets == c)[0].tolist() for c in classes}
    chosen, quota = [], batch_size // len(classes)
    for c in classes:
        np.random.shuffle(per_class[c])
        chosen.extend(per_class[c][:quota])
        per_class[c] = per_class[c][quota:]
    missing = batch_size - len(chosen)
    while missing > 0:
        dealt = False
        for c in classes:
            if missing == 0:
                break
            if per_class[c]:
                chosen.append(per_class[c].pop(0))
                missing -= 1
                dealt = True
        if not dealt:
            raise ValueError('not enough samples to fill the subset')  # the reference loops forever here
    first = original_dataset[0][0]
    if len(first.shape) == 3:
        x = torch.stack([original_dataset[i][0] for i in chosen])
    else:
        x = torch.cat([original_dataset[i][0] for i in chosen], dim=0)
    if len(x.shape) == 3:
        x = x.unsqueeze(1)
    y = torch.tensor([original_dataset[i][1] for i in chosen], dtype=torch.int64)
    return x, y


def get_classifier_head_weights(models, original_dataset, num_classes):
    """Share of each class's samples every model was trained on (dataset_utils.py:273-282)."""
    counts = np.zeros((len(models), num_classes))
    for i, model in enumerate(models):
        targets = np.array(original_dataset.targets)[model.train_indices + model.val_indices]
        for c in range(num_classes):
            counts[i][c] = np.sum(targets == c)
    return counts / np.sum(counts, axis=0, keepdims=True)
