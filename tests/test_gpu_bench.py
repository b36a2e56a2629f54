"""-m gpu: the end-to-end workloads of bench.py (--workload c1 | c3 | c4) run and report what they claim, at
reduced sizes where the full one takes seconds."""
import json
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope='module')
def ops():
    from neuron_interpolation_model_fusion_amd import ops
    return ops


@pytest.fixture(autouse=True)
def _restore_twin_copy_mode():
    """The runners skip copy_model()'s discarded initialisation (a process-wide switch): leave it as it was."""
    from neuron_interpolation_model_fusion_amd.models.base_model import twin_copy
    saved = twin_copy.replay_init_draws
    yield
    twin_copy.replay_init_draws = saved


def test_c3_workload_recovers_the_planted_permutation(ops):
    import bench_e2e
    r = bench_e2e.run_c3(ops, torch.device('cuda', 0), 1, 1, hidden=512, batch=256)
    assert r['extra']['planted_permutation_recovered']
    assert r['roofline']['achieved'] > 0 and r['seconds'] > 0
    json.dumps(r)  # the bench line must serialise


def test_c4_workload_runs_at_reduced_batch(ops):
    import bench_e2e
    r = bench_e2e.run_c4(ops, torch.device('cuda', 0), 1, 1, batch=64, depth=2)
    assert r['roofline']['launches'] > 0 and r['seconds'] > 0
    json.dumps(r)


def test_c1_workload_reports_the_tridiagonalisation(ops):
    import bench_e2e
    r = bench_e2e.run(  # full C1: a VGG11 fusion takes ~0.4 s
        'c1', ops, torch.device('cuda', 0), 1, 1)
    assert r['roofline']['eigensolver_calls'] > 0 and r['roofline']['achieved'] > 0
    assert r['config']['hf2_seconds'] > 0
    json.dumps(r)


def test_sinkhorn_workload_runs_at_reduced_size(ops):
    import bench_e2e
    r = bench_e2e.run_sk(ops, torch.device('cuda', 0), 1, 1, n=512, iters=30)
    assert r['extra']['iterations'] == 30 and r['roofline']['achieved'] > 0
    json.dumps(r)
