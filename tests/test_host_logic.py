"""CPU tests of host-side logic that has to reproduce torch's own bookkeeping draw for draw."""
import torch
from torch.utils.data import DataLoader, TensorDataset, random_split

from neuron_interpolation_model_fusion_amd.fusion_algorithms.alf import ALFusion


def _reference_orders(n, batch_size, epochs, seed):
    """What the reference's loaders (alf.py:517-520) visit: per epoch the shuffled train pass, then the val pass."""
    torch.manual_seed(seed)
    train, val = random_split(TensorDataset(torch.arange(n)), [0.8, 0.2])
    train_dl = DataLoader(train, batch_size=batch_size, shuffle=True)
    val_dl = DataLoader(val, batch_size=batch_size, shuffle=False)
    out = []
    for _ in range(epochs):
        out.append(torch.cat([b[0] for b in train_dl]))
        out.append(torch.cat([b[0] for b in val_dl]))
    return out, torch.rand(3)  # the trailing draw shows the generator ended in the same state


def test_on_device_batching_visits_samples_in_the_loaders_order():
    for n, bs, seed in ((37, 8, 5), (10, 32, 0), (101, 7, 123)):
        ref, ref_tail = _reference_orders(n, bs, 3, seed)
        torch.manual_seed(seed)
        train_idx, val_idx = ALFusion._split_indices(n)
        got = []
        for _ in range(3):
            got.append(train_idx[ALFusion._loader_order(train_idx.shape[0], True)])
            ALFusion._loader_order(val_idx.shape[0], False)
            got.append(val_idx)
        tail = torch.rand(3)
        assert len(ref) == len(got)
        for r, g in zip(ref, got):
            assert torch.equal(r, g)
        assert torch.equal(ref_tail, tail)
