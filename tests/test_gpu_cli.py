"""-m gpu: the four fusion CLIs end to end on small models saved as whole-module checkpoints (child processes)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPTS = os.path.join(ROOT, 'main_scripts', 'run_individual_algorithms')


@pytest.fixture(scope='module')
def savedir(tmp_path_factory):
    from neuron_interpolation_model_fusion_amd.models import MLPNet
    d = tmp_path_factory.mktemp('MLPs-MNIST')
    torch.manual_seed(0)
    for k in range(2):
        m = MLPNet(64, 24, 10, use_bias=True, use_activations=False, train_indices=list(range(200)), val_indices=[])
        torch.save(m, d / f'model_{k}.pt')
    X = torch.randn(400, 1, 8, 8)
    y = torch.arange(400) % 10
    torch.save({'data': X, 'targets': y}, d / 'data.pt')
    return d


def _run(script, savedir, *extra):
    cmd = [sys.executable, os.path.join(SCRIPTS, script), '--model_savedir', str(savedir), '--num_fusion_samples', '100',
           '--fusion_data', str(savedir / 'data.pt'), '--eval_data', str(savedir / 'data.pt'), *extra]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    return out.stdout


def _check_saved(savedir, names):
    from neuron_interpolation_model_fusion_amd.models import MLPNet
    from neuron_interpolation_model_fusion_amd.utils.run_utils import load_model
    for n in names:
        m = load_model(savedir / n)
        assert type(m) is MLPNet and all(torch.isfinite(p).all() for p in m.parameters()), n


def test_run_hf_kf(savedir):
    out = _run('run_hf_kf.py', savedir, '--normalize_neuron_importance')
    _check_saved(savedir, [f'{a}-2-{m}.pt' for a in ('hf', 'kf') for m in ('uniform', 'conductance', 'deeplift')])
    assert 'kf-2-deeplift: loss' in out


def test_run_gr(savedir):
    _run('run_gr.py', savedir)
    _check_saved(savedir, [f'gr-{m}.pt' for m in ('uniform', 'conductance', 'deeplift')])


def test_run_otf(savedir):
    _run('run_otf.py', savedir, '--num_models', '2')
    _check_saved(savedir, ['ot-uniform-acts.pt', 'ot-deeplift-wts.pt', 'ot-conductance-acts-0.1.pt', 'ot-conductance-wts-0.5.pt'])


def test_run_alf(savedir):
    # 2 models, 10 classes: with the reference's own head-weight call (num_classes=args.num_models, run_alf.py:114)
    # --use_weighted_loss cannot broadcast the (2,) shares against the (B, 10) logits; --head_weights_per_dataset_class
    # (ours) counts the dataset's classes
    _run('run_alf.py', savedir, '--epochs_per_level', '2', '--use_weighted_loss', '--head_weights_per_dataset_class',
         '--normalize_acts_for_kmeans', '--run_last_layer_kd', '--use_kl_kd_on_head', '--lr_per_level', '0.001,0.0005')
    _check_saved(savedir, ['alf-2-uniform.pt', 'alf-2-deeplift.pt', 'alf-2-logit-kd.pt'])


def test_run_alf_default_flags_behave_like_the_reference(savedir):
    """Same flags, same behaviour: without --use_weighted_loss the shares are computed and dropped (run_alf.py:114-115)
    and the run succeeds; with it, the reference's shape error at alf.py:405 comes back unchanged."""
    _run('run_alf.py', savedir, '--epochs_per_level', '1', '--run_last_layer_kd')
    _check_saved(savedir, ['alf-2-uniform.pt', 'alf-2-logit-kd.pt'])
    cmd = [sys.executable, os.path.join(SCRIPTS, 'run_alf.py'), '--model_savedir', str(savedir), '--num_fusion_samples',
           '100', '--fusion_data', str(savedir / 'data.pt'), '--eval_data', str(savedir / 'data.pt'),
           '--epochs_per_level', '1', '--use_weighted_loss']
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and 'must match the size of tensor b (2)' in out.stderr
