"""ALF / KF gradient fusion (reference: main_scripts/run_individual_algorithms/run_alf.py; same flags)."""
import argparse
from pathlib import Path

import torch

import _common as c
from neuron_interpolation_model_fusion_amd.fusion_algorithms import ALFusion, ALFusionConfig
from neuron_interpolation_model_fusion_amd.utils.dataset_utils import get_classifier_head_weights
from neuron_interpolation_model_fusion_amd.utils.fusion_utils import get_num_levels


def parse_args():
    p = argparse.ArgumentParser(description='Run ALFusion with full CLI config support.')
    p.add_argument('--model_savedir', type=str, required=True, help='Directory containing base models')
    p.add_argument('--num_models', type=int, required=False, help='Number of models to fuse (default: all in the directory)')
    p.add_argument('--device', type=str, default='cuda', help='Device to use (a HIP device; there is no CPU path)')
    p.add_argument('--seed', type=int, default=42, help='Random seed (default: 42)')
    p.add_argument('--num_fusion_samples', type=int, required=True, help='Number of samples to use for fusion.')
    p.add_argument('--save_dir', type=str, help='Directory to save fused models (default: same as model_savedir)')
    p.add_argument('--plot_path', type=str, help='Accepted for compatibility; metrics are printed instead')
    p.add_argument('--optimizer_per_level', type=str, default='adam', help='One optimizer or a comma-separated list')
    p.add_argument('--lr_per_level', type=str, default='0.001', help='One learning rate or a comma-separated list')
    p.add_argument('--weight_decay_per_level', type=str, default='0.0001', help='One weight decay or a comma-separated list')
    p.add_argument('--epochs_per_level', type=str, default='100', help='One epoch count or a comma-separated list')
    p.add_argument('--weight_perturbation_per_level', type=str, default='1',
                   help='(1 - eps) * original_weights + eps * random_weights; one eps or a comma-separated list')
    p.add_argument('--use_scheduler', action='store_true')
    p.add_argument('--normalize_neuron_importances', action='store_true')
    p.add_argument('--align_skip_connections', action='store_true')
    p.add_argument('--train_batch_size', type=int, default=32)
    p.add_argument('--max_patience_epochs', type=int, default=15)
    p.add_argument('--normalize_acts_for_kmeans', action='store_true')
    p.add_argument('--num_kmeans_runs', type=int, default=1)
    p.add_argument('--init_kmeans_with_2nd_acts', action='store_true')
    p.add_argument('--use_weighted_loss', action='store_true')
    p.add_argument('--run_last_layer_kd', action='store_true')
    p.add_argument('--use_kl_kd_on_head', action='store_true')
    p.add_argument('--kd_temperature', type=float, default=2.0)
    p.add_argument('--verbose', action='store_true')
    p.add_argument('--head_weights_per_dataset_class', action='store_true',
                   help='Not a reference flag. The reference computes the classifier-head shares with '
                        'num_classes=args.num_models (run_alf.py:114), which only broadcasts against the logits when '
                        'the number of models equals the number of classes; this flag counts the dataset classes instead.')
    c.add_data_args(p)
    return p.parse_args()


def build_fusion_config(args, models, dataset, device, num_classes):
    num_levels = get_num_levels(models)
    # run_alf.py:114-115 as written: the shares are always computed, with num_classes=args.num_models, kept in
    # float64, and used only under --use_weighted_loss.  --head_weights_per_dataset_class (ours) swaps in the
    # dataset's class count; nothing else changes.
    head_classes = num_classes if args.head_weights_per_dataset_class else args.num_models
    cls_head_weights = get_classifier_head_weights(models, dataset, num_classes=head_classes)
    cls_head_weights = torch.from_numpy(cls_head_weights).to(device) if args.use_weighted_loss else None

    def expand(name, cast):
        vals = [cast(x.strip()) for x in getattr(args, name).split(',')]
        return vals * num_levels if len(vals) == 1 else vals

    return ALFusionConfig(
        X=None, y=None,
        optimizer_per_level=expand('optimizer_per_level', str), lr_per_level=expand('lr_per_level', float),
        weight_decay_per_level=expand('weight_decay_per_level', float), epochs_per_level=expand('epochs_per_level', int),
        weight_perturbation_per_level=expand('weight_perturbation_per_level', float),
        use_scheduler=args.use_scheduler, neuron_importance_method='uniform',
        normalize_neuron_importances=args.normalize_neuron_importances,
        align_skip_connections=args.align_skip_connections, train_batch_size=args.train_batch_size,
        max_patience_epochs=args.max_patience_epochs, normalize_acts_for_kmeans=args.normalize_acts_for_kmeans,
        num_kmeans_runs=args.num_kmeans_runs, init_kmeans_with_2nd_acts=args.init_kmeans_with_2nd_acts,
        use_weighted_loss=args.use_weighted_loss, cls_head_weights=cls_head_weights, use_kd_on_head=False,
        kd_temperature=-1, verbose=args.verbose)


def main():
    args = parse_args()
    c.set_seed(args.seed)
    device = torch.device(args.device)
    savedir = Path(args.model_savedir)
    save_dir = Path(args.save_dir) if args.save_dir else savedir
    paths = sorted(savedir.glob('**/model*.pt'))
    if args.num_models is None:
        args.num_models = len(paths)
    models = c.load_models(paths[:args.num_models], device)
    dataset = c.fusion_dataset(args, models, savedir)
    X, y = c.fusion_samples(args, dataset, models[0], device, use_model_indices=False)
    cfg = build_fusion_config(args, models, dataset, device, c.num_classes_of(savedir))
    cfg.X, cfg.y = X, y

    for method in ['uniform', 'conductance', 'deeplift']:
        print(f'Running ALFusion with {method}')
        cfg.neuron_importance_method = method
        fused = ALFusion(models).fuse(cfg)
        torch.save(fused, save_dir / f'alf-{args.num_models}-{method}.pt')
        print(f"Saved fused model to {save_dir / f'alf-{args.num_models}-{method}.pt'}")
        del fused
        torch.cuda.empty_cache()

    if args.run_last_layer_kd:
        print('Running ALFusion with last layer knowledge distillation')
        cfg.use_kd_on_head = args.use_kl_kd_on_head
        cfg.kd_temperature = args.kd_temperature
        cfg.skip_levels = list(range(get_num_levels(models) - 1))
        fused = ALFusion(models).fuse(cfg)
        torch.save(fused, save_dir / f'alf-{args.num_models}-logit-kd.pt')
        print(f"Saved last layer KD model to {save_dir / f'alf-{args.num_models}-logit-kd.pt'}")
    c.report(args, save_dir, ['alf-*.pt'], device)


if __name__ == '__main__':
    c.add_safe_globals()
    main()
