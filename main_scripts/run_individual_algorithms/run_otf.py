"""OTFusion in activation and weight mode (reference: main_scripts/run_individual_algorithms/run_otf.py)."""
import argparse
from pathlib import Path

import torch

import _common as c
from neuron_interpolation_model_fusion_amd.fusion_algorithms import OTFusion


def parse_args():
    parser = argparse.ArgumentParser(description='Run OTFusion with CLI arguments.')
    parser.add_argument('--model_savedir', type=str, required=True, help='Directory containing base models')
    parser.add_argument('--num_fusion_samples', type=int, default=400, help='Number of samples to use for fusion (default: 400)')
    parser.add_argument('--num_models', type=int, help='Number of models to fuse')
    parser.add_argument('--iid_split', action='store_true', help='Use IID split (default: non-IID)')
    parser.add_argument('--seed', type=int, default=3409, help='Random seed (default: 3409)')
    parser.add_argument('--device', type=str, default='cuda', help='Device to use (a HIP device; there is no CPU path)')
    parser.add_argument('--normalize_activations', action='store_true', help='Normalize activations before fusion (default: False)')
    parser.add_argument('--plot_path', type=str, default=None, help='Accepted for compatibility; metrics are printed instead')
    c.add_data_args(parser)
    return parser.parse_args()


def main():
    args = parse_args()
    c.set_seed(args.seed)
    device = torch.device(args.device)
    savedir = Path(args.model_savedir)
    if args.num_models is None:
        args.num_models = len(sorted(savedir.glob('model_*.pt')))
        print(f'No num_models specified, using all {args.num_models} models found in {savedir}')
    print(f'Using device: {device}')
    if args.iid_split:
        paths = list(savedir.glob('**/model*.pt'))[:args.num_models]
    else:
        paths = [savedir / f'model_{i}.pt' for i in range(args.num_models)]
    models = c.load_models(paths, device)
    dataset = c.fusion_dataset(args, models, savedir)
    X, y = c.fusion_samples(args, dataset, models[1], device)

    print('-' * 50)
    print('Running OT Fusion...')
    models_to_align, model_to_align_with = models[:-1], models[-1]

    def run_otfusion(mode, method, rescale=None):
        print(f'{mode.upper()} mode - {method.title()} Scores', f'(rescale min={rescale})' if rescale else '')
        strategy = {'mode': mode,
                    mode: {'activations_dataset': X, 'activation_labels': y, 'neuron_importance_method': method,
                           'normalize_activations': args.normalize_activations, 'rescale_min_importance': rescale}}
        fused = OTFusion(models_to_align, model_to_align_with).fuse_models(strategy, handle_skip=(mode == 'acts'))
        name = f'ot-{method}-{mode}' + (f'-{rescale}' if rescale else '') + '.pt'
        torch.save(fused, savedir / name)
        print(f'Saved to {savedir / name}')
        torch.cuda.empty_cache()

    for mode in ['acts', 'wts']:
        for method in ['uniform', 'conductance', 'deeplift']:
            run_otfusion(mode, method)
        for rescale_val in [0.1, 0.5]:
            run_otfusion(mode, 'conductance', rescale=rescale_val)
    print(f'All fused models saved to {savedir}')
    c.report(args, savedir, ['model*.pt', 'ot-*.pt'], device)


if __name__ == '__main__':
    c.add_safe_globals()
    main()
