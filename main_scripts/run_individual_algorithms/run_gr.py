"""Git Re-Basin activation matching of two models (reference: main_scripts/run_individual_algorithms/run_gr.py)."""
import argparse
from pathlib import Path

import torch

import _common as c
from neuron_interpolation_model_fusion_amd.fusion_algorithms import GitRebasin


def parse_args():
    parser = argparse.ArgumentParser(description='Run GitRebasin fusion with optional plotting.')
    parser.add_argument('--model_savedir', type=str, required=True, help='Directory containing base models')
    parser.add_argument('--num_fusion_samples', type=int, default=400, help='Number of samples to use for fusion (default: 400)')
    parser.add_argument('--iid_split', action='store_true', help='Use IID split (default: non-IID)')
    parser.add_argument('--seed', type=int, default=42, help='Random seed (default: 42)')
    parser.add_argument('--device', type=str, default='cuda', help='Device to use (a HIP device; there is no CPU path)')
    parser.add_argument('--plot_path', type=str, default=None, help='Accepted for compatibility; metrics are printed instead')
    c.add_data_args(parser)
    return parser.parse_args()


def main():
    args = parse_args()
    c.set_seed(args.seed)
    device = torch.device(args.device)
    savedir = Path(args.model_savedir)
    print(f'Using device: {device}')
    models = [m.eval() for m in c.load_models(sorted(savedir.glob('model*.pt')), device)]
    if len(models) > 2:
        print(f'Warning: More than 2 models found ({len(models)}); using the first two...')
    models = models[:2]
    print('-' * 50)
    print('Fusing models with GitRebasin (Hungarian Matching)...')
    dataset = c.fusion_dataset(args, models, savedir)
    X, y = c.fusion_samples(args, dataset, models[1], device)
    for method in ['uniform', 'conductance', 'deeplift']:
        print(f'Fusing with {method.title()} Scores')
        fused = GitRebasin(models[0], models[1]).fuse(X, y, neuron_importance_method=method, handle_skip=False)
        torch.save(fused, savedir / f'gr-{method}.pt')
        print(f"Fused model saved to {savedir / f'gr-{method}.pt'}")
        torch.cuda.empty_cache()
    c.report(args, savedir, ['model*.pt', 'gr-*.pt'], device)


if __name__ == '__main__':
    c.add_safe_globals()
    main()
