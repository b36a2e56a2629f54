"""Shared plumbing of the four fusion CLIs (reference: main_scripts/run_individual_algorithms/run_*.py).

Flags, output file names and the order of fusions follow the reference scripts.  Two things differ, both because
this machine has no torchvision and no network: the fusion samples come from ``--fusion_data`` (a file written with
``torch.save({'data': X, 'targets': y}, path)``) or from ``--synthetic_data`` (seeded N(0,1) inputs of the models'
input shape), and evaluation prints the metrics of ``utils.run_utils.eval_model`` (``--eval_data``) instead of
drawing bar plots.
"""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))

from neuron_interpolation_model_fusion_amd.utils.dataset_utils import TensorVisionDataset, sample_balanced_subset  # noqa: E402
from neuron_interpolation_model_fusion_amd.utils.general_utils import set_seed  # noqa: E402,F401
from neuron_interpolation_model_fusion_amd.utils.run_utils import add_safe_globals, eval_model  # noqa: E402,F401


def add_data_args(parser):
    parser.add_argument('--fusion_data', type=str, default=None,
                        help="File with {'data': X, 'targets': y} tensors to draw the fusion samples from")
    parser.add_argument('--synthetic_data', action='store_true',
                        help='Draw the fusion samples from seeded N(0,1) inputs (no dataset on this machine)')
    parser.add_argument('--eval_data', type=str, default=None,
                        help="File with {'data': X, 'targets': y} test tensors; metrics are printed for every model")


def num_classes_of(savedir):
    path = Path(savedir).as_posix()
    if 'MNIST' in path:
        return 10
    if 'Tiny-ImageNet' in path:
        return 200
    if 'CIFAR-100' in path:
        return 100
    return 10


def load_models(paths, device):
    models = []
    for p in paths:
        m = torch.load(p, weights_only=True, map_location=device).to(device)
        print(f'Model loaded from {p}. Model class: {m.__class__.__name__}.')
        models.append(m)
    return models


def input_shape_of(model):
    if hasattr(model, 'input_shape'):
        return tuple(model.input_shape)
    if hasattr(model, 'img_size') and hasattr(model, 'in_c'):
        return (model.in_c, model.img_size, model.img_size)
    if hasattr(model, 'input_size'):
        side = int(round(model.input_size ** 0.5))
        return (1, side, side) if side * side == model.input_size else (model.input_size,)
    raise ValueError(f'cannot infer the input shape of {model.__class__.__name__}; pass --fusion_data')


def fusion_dataset(args, models, savedir):
    if args.fusion_data:
        return TensorVisionDataset.load(args.fusion_data)
    if args.synthetic_data:
        size = max(4 * args.num_fusion_samples, 64 * num_classes_of(savedir))
        return TensorVisionDataset.synthetic(input_shape_of(models[0]), num_classes_of(savedir), size, seed=args.seed)
    raise SystemExit('no dataset on this machine: pass --fusion_data FILE or --synthetic_data')


def fusion_samples(args, dataset, model, device, use_model_indices=True):
    indices = None
    if use_model_indices and not args.synthetic_data and model.train_indices is not None:
        indices = list(model.train_indices) + list(model.val_indices or [])
    X, y = sample_balanced_subset(dataset, indices, batch_size=args.num_fusion_samples)
    return X.to(device), y.to(device)


def report(args, savedir, patterns, device):
    if not args.eval_data:
        return
    test = TensorVisionDataset.load(args.eval_data)
    dl = torch.utils.data.DataLoader(test, batch_size=128)
    paths = [p for pat in patterns for p in sorted(Path(savedir).glob(pat))]
    for p in paths:
        model = torch.load(p, weights_only=True, map_location=device).to(device)
        stats = eval_model(model, dl, device=device, num_classes=num_classes_of(savedir))
        print(f"{p.stem}: loss {stats['loss']:.4f}  accuracy {stats['accuracy']:.4f}  f1 {stats['f1']:.4f}  "
              f"roc_auc {stats['roc_auc']:.4f}")
