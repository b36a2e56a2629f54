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
    _run('run_alf.py', savedir, '--epochs_per_level', '2', '--use_weighted_loss', '--normalize_acts_for_kmeans',
         '--run_last_layer_kd', '--lr_per_level', '0.001,0.0005')
    _check_saved(savedir, ['alf-2-uniform.pt', 'alf-2-deeplift.pt', 'alf-2-logit-kd.pt'])
