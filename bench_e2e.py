"""End-to-end workloads of bench.py (`--workload c1|c3|c4`): whole fusions on the BASELINE.json configurations that
fit one GPU, timed wall-clock, each with the roofline of the library kernel that dominates it.

  c1  HF/KF linear fusion at the VGG11 / CIFAR shape (config 1): KF of 4 random-init VGG11s (the only method the
      reference runs on 4 models, hf_kf.py:381-390) and HF of the first two, 400 fusion samples.  Dominant library
      kernel: the Householder tridiagonalisation pass of the LS refit's eigensolver (HBM-bound).
  c3  Git Re-Basin of two MLPs with hidden width 8192 (config 3), 4096 samples, planted permutation: Gram 8192 x 8192
      over 4096 rows (f32 MFMA) + the 8192 x 8192 assignment.
  c4  ALF neuron clustering + targets on every level of two ViTs at the CIFAR geometry (config 4: hidden 384, 65
      tokens, 5000 samples => m = 325 000 rows, N = 768, k = 384); the SGD fit is torch autograd and is not part of
      the timed region.  Dominant library kernel: the symmetric Gram (f32 MFMA).

Weights are random (the reference's VGG / ViT checkpoints are missing blobs) and inputs synthetic.  The
discarded random initialisation of copy_model() is skipped here (NIF_FAST_TWIN_COPY semantics, models/base_model.py):
same fused weights, 30 ms less host time per fusion."""
import contextlib
import io
import time

import numpy as np
import torch

F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md "Peak FP32 (matrix)"
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md "HBM3E peak BW" (spec)


def _timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    times = []
    for _ in range(steps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    return times


def _quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        return fn(*a, **k)


def run_c1(ops, device, steps, warmup):
    from neuron_interpolation_model_fusion_amd.fusion_algorithms import LSFusion
    from neuron_interpolation_model_fusion_amd.models import VGG
    from neuron_interpolation_model_fusion_amd.models.base_model import twin_copy
    twin_copy.replay_init_draws = False
    models = []
    for k in range(4):
        torch.manual_seed(100 + k)
        models.append(VGG('VGG11', 10, (3, 32, 32), batch_norm=False, use_bias=True, use_activations=False).eval().to(device))
    torch.manual_seed(0)
    np.random.seed(0)
    X = torch.randn(400, 3, 32, 32, device=device)

    def kf():
        return _quiet(LSFusion(models).fuse, X, None, 'k_means')

    def hf():
        return _quiet(LSFusion(models[:2]).fuse, X, None, 'hungarian')

    _timed(kf, 0, max(warmup, 1))
    ops.profile_collect(ops.PROF_EIGH_TRD)
    ops.profile_work(ops.PROF_EIGH_TRD)
    ops.profile_enable(True, slots=[ops.PROF_EIGH_TRD])
    t_kf = _timed(kf, steps, 0)
    ops.profile_enable(False)
    trd_ms, calls = ops.profile_collect(ops.PROF_EIGH_TRD)
    trd_bytes = ops.profile_work(ops.PROF_EIGH_TRD)
    t_hf = _timed(hf, steps, 1)
    achieved = trd_bytes / (trd_ms * 1e-3) / 1e9 if trd_ms > 0 else None
    return dict(
        seconds=float(np.mean(t_kf)),
        workload='C1 KF linear fusion of 4 random-init VGG11 (CIFAR geometry), 400 fusion samples, 10 k-means tries per '
                 'layer, 9 layers; whole LSFusion.fuse() wall-clock',
        extra={'kf4_seconds_each': [round(t, 4) for t in t_kf], 'hf2_seconds': float(np.mean(t_hf)),
               'hf2_seconds_each': [round(t, 4) for t in t_hf], 'levels_per_fusion': 9,
               'kf4_levels_per_s': 9 / float(np.mean(t_kf)), 'hf2_levels_per_s': 9 / float(np.mean(t_hf)),
               'reference_hf2_seconds_8_cpu_cores': 291.8},
        roofline={'kernel': 'nif::trd_step_kernel (Householder tridiagonalisation of the normal matrix: fused rank-2 '
                            'update + matrix-vector pass, one launch per column)',
                  'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                  'frac': achieved / HBM_PEAK_GBS if achieved else None, 'traffic': None,
                  'algorithmic_bytes': trd_bytes, 'kernel_ms_total': trd_ms, 'eigensolver_calls': calls,
                  'share_of_wall_clock': trd_ms * 1e-3 / sum(t_kf) if t_kf else None,
                  'note': '8 (n-j-1)^2 bytes read per column j plus 8 (n-j-1)^2 written by the steps that store the trailing matrix (every step up to a trailing size of 1024, every other step above); the pass is launch/latency-bound below n ~ 2500'})


def run_c3(ops, device, steps, warmup, hidden=8192, batch=4096, hard=None):
    from neuron_interpolation_model_fusion_amd.fusion_algorithms import GitRebasin
    from neuron_interpolation_model_fusion_amd.models import MLPNet
    from neuron_interpolation_model_fusion_amd.models.base_model import twin_copy
    twin_copy.replay_init_draws = False
    torch.manual_seed(0)
    m0 = MLPNet(784, hidden, 10, use_bias=True, use_activations=False).to(device)
    m1 = m0.copy_model()
    perm = torch.randperm(hidden, device=device)
    with torch.no_grad():
        fc1, fc2 = m1.level1[1], m1.level2[-1]
        fc1.weight.copy_(fc1.weight[perm] + 0.02 * torch.randn_like(fc1.weight))
        fc1.bias.copy_(fc1.bias[perm])
        fc2.weight.copy_(fc2.weight[:, perm])
    X = torch.randn(batch, 1, 28, 28, device=device)
    state = {}

    def gr():
        g = GitRebasin(m0, m1)
        _quiet(g.fuse, X, None, 'uniform')
        state['gr'] = g

    _timed(gr, 0, max(warmup, 1))
    ops.profile_collect(ops.PROF_XTY)
    ops.profile_enable(True, slots=[ops.PROF_XTY])
    times = _timed(gr, steps, 0)
    ops.profile_enable(False)
    xty_ms, launches = ops.profile_collect(ops.PROF_XTY)
    ok = bool((state['gr'].permutations[0].long() == torch.argsort(perm)).all())
    paths = list(getattr(ops.lap_solve_many, 'last_paths', []))
    hard_leg = None
    if hard if hard is not None else hidden >= 4096:
        # the same assignment size on a HARD instance -- the cost between UNRELATED post-ReLU activations, where the
        # dual initialisation leaves a third of the rows free -- next to scipy on this host's CPU (the planted
        # instance above is solved by the initialisation: its 0.4 ms says nothing about the solver)
        from scipy.optimize import linear_sum_assignment
        g = torch.Generator(device=device).manual_seed(3)
        A0 = torch.randn(batch, hidden, device=device, generator=g).clamp_min(0)
        A1 = torch.randn(batch, hidden, device=device, generator=g).clamp_min(0)
        G, na, nb = ops.xty(A0, A1)
        C = ops.cost_dist(G, na, nb)
        del A0, A1, G
        ops.lap_solve(C)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        col = ops.lap_solve(C)
        torch.cuda.synchronize()
        t_gpu = time.perf_counter() - t0
        Cn = C.cpu().numpy()
        t0 = time.perf_counter()
        r_ref, c_ref = linear_sum_assignment(Cn)
        t_cpu = time.perf_counter() - t0
        got = float(Cn[np.arange(hidden), col.cpu().numpy()].astype(np.float64).sum())
        ref = float(Cn[r_ref, c_ref].astype(np.float64).sum())
        hard_leg = {'n': hidden, 'gpu_fast_solver_s': t_gpu, 'scipy_cpu_s': t_cpu, 'speedup': t_cpu / t_gpu,
                    'same_optimal_cost': abs(got - ref) <= 1e-9 * abs(ref),
                    'instance': 'cost = distances between unrelated post-ReLU activations (no planted matching)'}
    flops = steps * (2.0 * batch * hidden * hidden + 2.0 * batch * 10 * 10)
    achieved = flops / (xty_ms * 1e-3) / 1e12 if xty_ms > 0 else None
    return dict(
        seconds=float(np.mean(times)),
        workload=f'C3 Git Re-Basin of two MLPs 784-{hidden}-10, {batch} samples, planted hidden-unit permutation: '
                 f'{hidden} x {hidden} activation Gram + assignment + refit; whole GitRebasin.fuse() wall-clock',
        extra={'seconds_each': [round(t, 4) for t in times], 'planted_permutation_recovered': ok,
               'assignment_paths': paths, 'hard_assignment': hard_leg, 'levels_per_fusion': 2,
               'levels_per_s': 2 / float(np.mean(times))},
        roofline={'kernel': 'nif::xty_kernel<float, MmaF32> (v_mfma_f32_32x32x2_f32)', 'bound': 'mfma',
                  'achieved': achieved, 'peak': F32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                  'frac': achieved / F32_MFMA_PEAK_TFLOPS if achieved else None, 'traffic': None,
                  'algorithmic_flops': flops, 'kernel_ms_total': xty_ms, 'launches': launches,
                  'share_of_wall_clock': xty_ms * 1e-3 / sum(times) if times else None})


def run_c4(ops, device, steps, warmup, batch=5000, depth=7):
    from neuron_interpolation_model_fusion_amd.fusion_algorithms import ALFusion, ALFusionConfig
    from neuron_interpolation_model_fusion_amd.models import ViT
    models = []
    for k in range(2):
        torch.manual_seed(100 + k)
        models.append(ViT(in_c=3, num_classes=100, img_size=32, patch=8, depth=depth, hidden=384, mlp_hidden=384,
                          head=8).eval().to(device))
    torch.manual_seed(0)
    X = torch.randn(batch, 3, 32, 32, device=device)
    nlev = len(models[0].get_ordered_trainable_named_levels)
    alf = ALFusion(models)
    alf.ni_scores = [{l: torch.ones(m.get_ordered_trainable_named_levels[l].output_width) for l in range(nlev)}
                     for m in models]
    cfg = ALFusionConfig(X=X, y=None, optimizer_per_level=['adam'] * nlev, lr_per_level=[1e-3] * nlev,
                         weight_decay_per_level=[0.0] * nlev, epochs_per_level=[0] * nlev, verbose=False,
                         normalize_acts_for_kmeans=True, init_kmeans_with_2nd_acts=False, use_weighted_loss=False)
    widths = [models[0].get_ordered_trainable_named_levels[l].output_width for l in range(nlev - 1)]

    def clustering():
        level_X = X
        for level in range(nlev - 1):
            with torch.no_grad():
                level_X = X if level == 0 else models[0].forward_level(level_X, level - 1)
            _quiet(alf._k_means_v2, level, widths[level], cfg, level_X)

    _timed(clustering, 0, max(warmup, 1))
    ops.profile_collect(ops.PROF_XTY)
    ops.profile_enable(True, slots=[ops.PROF_XTY])
    times = _timed(clustering, steps, 0)
    ops.profile_enable(False)
    xty_ms, launches = ops.profile_collect(ops.PROF_XTY)
    m = batch * 65
    flops = steps * sum(float(m) * (2 * w) * (2 * w + 1) for w in widths)   # symmetric Gram: m N (N + 1), N = 2 w
    achieved = flops / (xty_ms * 1e-3) / 1e12 if xty_ms > 0 else None
    return dict(
        seconds=float(np.mean(times)),
        workload=f'C4 ALF neuron clustering + targets on the {nlev - 1} hidden levels of two ViTs (hidden 384, 65 tokens, '
                 f'depth {depth}), {batch} samples => m = {m} rows, N = 768, k = 384, per-sample-normalised activations; '
                 f'capture + Gram + Lloyd + targets wall-clock (the SGD level fit is torch autograd, not timed)',
        extra={'seconds_each': [round(t, 4) for t in times], 'levels_per_fusion': nlev - 1,
               'levels_per_s': (nlev - 1) / float(np.mean(times))},
        roofline={'kernel': 'nif::xty_kernel<float, MmaF32> symmetric Gram (v_mfma_f32_32x32x2_f32)', 'bound': 'mfma',
                  'achieved': achieved, 'peak': F32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                  'frac': achieved / F32_MFMA_PEAK_TFLOPS if achieved else None, 'traffic': None,
                  'algorithmic_flops': flops, 'kernel_ms_total': xty_ms, 'launches': launches,
                  'share_of_wall_clock': xty_ms * 1e-3 / sum(times) if times else None})


def run_sk(ops, device, steps, warmup, n=8192, iters=200):
    """The additional Sinkhorn mode (not a reference path): `iters` iterations on an n x n cost, uniform marginals."""
    g = torch.Generator(device=device).manual_seed(0)
    C = torch.rand(n, n, device=device, generator=g)
    a = torch.full((n,), 1.0 / n, dtype=torch.float64, device=device)
    state = {}

    def solve():
        state['r'] = ops.sinkhorn(C, a, a, 0.05, max_iter=iters, tol=0.0)

    times = _timed(solve, steps, max(warmup, 1))
    sec = float(np.mean(times))
    achieved = 8.0 * n * n * iters / sec / 1e9
    return dict(
        seconds=sec,
        workload=f'Sinkhorn mode (additional, approximate; the reference uses exact ot.emd): {iters} iterations on a '
                 f'{n} x {n} cost, uniform marginals, reg 0.05; whole ops.sinkhorn() wall-clock incl. K build and plan',
        extra={'seconds_each': [round(t, 5) for t in times], 'iterations': state['r']['iterations'],
               'marginal_error': state['r']['err'], 'us_per_iteration': sec / iters * 1e6},
        roofline={'kernel': 'nif::sk_colpass_kernel + sk_colfinish_kernel + sk_rowpass_kernel (one Sinkhorn iteration)',
                  'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                  'frac': achieved / HBM_PEAK_GBS, 'traffic': None, 'algorithmic_bytes': 8.0 * n * n * iters,
                  'note': '8 n^2 bytes per iteration (K f32 streamed twice); wall-clock of the whole call, so the '
                          'K build, the plan and the host round trips every 50 iterations are inside'})


def run_c5(ops, device, steps, warmup, dist=None, world=1, rank=0, width=3072, models=8, rows=200_000, lloyd_iters=3,
           host_sync=False):
    """BASELINE.json config 5: synthetic 8-model fusion at ViT-B width -- one level of neuron clustering on N = 8 * width
    neuron columns over 200 000 fusion samples whose ROWS are sharded over the ranks (strong scaling: rows per GPU =
    rows / world).  One step = this rank's partial symmetric Gram (f32 MFMA) -> ONE all-reduce of the packed block upper
    triangle in f64 (written by the reduction kernel, expanded by one kernel into the f32 k-means storage + f64
    diagonal; skipped with one rank) -> ALF's first assignment + `lloyd_iters` Lloyd steps on the replicated N x N Gram
    (alf.py:221-272).  Returns the dict bench.py prints; time is the max over ranks."""
    import contextlib as _ctx
    from neuron_interpolation_model_fusion_amd import parallel
    N, k = models * width, width
    lo, hi = rows * rank // world, rows * (rank + 1) // world
    m_local = hi - lo
    g = torch.Generator(device=device).manual_seed(500 + rank)
    A = torch.empty((m_local, N), dtype=torch.float32, device=device)
    perms = [torch.randperm(width, generator=torch.Generator().manual_seed(77 + j)).to(device) for j in range(models)]
    for r0 in range(0, m_local, 8192):   # model j = a noisy column permutation of a shared signal, chunked to bound temporaries
        r1 = min(m_local, r0 + 8192)
        base = torch.randn(r1 - r0, width, device=device, generator=g)
        for j in range(models):
            A[r0:r1, j * width:(j + 1) * width] = 0.9 * base[:, perms[j]] + 0.3 * torch.randn(r1 - r0, width, device=device, generator=g)
    s = torch.ones(N, device=device)
    sharded = dist is not None and world > 1
    state = {}

    def level():
        ctx = parallel.SampleSharding(None) if sharded else _ctx.nullcontext()
        with ctx:
            if sharded and host_sync:
                torch.cuda.synchronize()   # gloo rehearsal on a shared GPU only (see bench.py level_front)
            G, gdiag = ops.kmeans_gram(A, rows_sharded=True)
        km = ops.KMeans(G, gdiag, s, k, gdiag=gdiag)
        assign = km.assign_to_neurons(0)
        for _ in range(lloyd_iters):
            assign, nearest, cost, kprime = km.lloyd_step(assign)
        state['assign'], state['cost'], state['kprime'] = assign, cost, kprime

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(warmup, 1)):
        level()
    sync()
    ops.profile_collect(ops.PROF_XTY)
    ops.profile_enable(True, slots=[ops.PROF_XTY])
    t0 = time.perf_counter()
    for _ in range(steps):
        level()
    sync()
    elapsed = time.perf_counter() - t0
    ops.profile_enable(False)
    xty_ms, launches = ops.profile_collect(ops.PROF_XTY)
    if dist is not None:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # every model's neuron j' = perms[j]^-1 ... lands in the cluster of the model-0 neuron it copies: clusters of `models`
    assign = state['assign'].view(models, width)
    want = torch.stack([torch.argsort(perms[0])[perms[j]] for j in range(models)])   # cluster label = model-0 column of the same signal
    planted = bool((assign.long() == want).all())
    flops = float(m_local) * N * (N + 1)
    achieved = flops * launches / (xty_ms * 1e-3) / 1e12 if xty_ms > 0 else None
    return dict(
        elapsed=elapsed, steps=steps, m_local=m_local, N=N, k=k,
        workload=f'C5 synthetic {models}-model fusion at width {width}: N = {N} neuron columns, k = {k} clusters, {rows} '
                 f'fusion samples with rows sharded over {world} GPU(s) ({m_local} per GPU); per level: partial symmetric '
                 f'Gram (f32 MFMA) + one f64 all-reduce of the packed block upper triangle + first assignment + '
                 f'{lloyd_iters} Lloyd steps on the replicated Gram',
        extra={'planted_clusters_recovered': planted, 'kprime': int(state['kprime']), 'lloyd_cost': float(state['cost']),
               'allreduce_bytes_per_level': 8 * (ops.lib.nif_packed_upper_len(N, parallel.TRI_BLOCK)) if sharded else 0},
        roofline={'kernel': 'nif::xty_kernel<float, MmaF32> symmetric Gram (v_mfma_f32_32x32x2_f32)', 'bound': 'mfma',
                  'achieved': achieved, 'peak': F32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                  'frac': achieved / F32_MFMA_PEAK_TFLOPS if achieved else None, 'traffic': None,
                  'algorithmic_flops_per_launch': flops, 'avg_kernel_ms': xty_ms / max(launches, 1), 'launches': launches,
                  'share_of_wall_clock': xty_ms * 1e-3 / elapsed if elapsed > 0 else None})


RUNNERS = {'c1': run_c1, 'c3': run_c3, 'c4': run_c4, 'sk': run_sk}


def run(workload, ops, device, steps, warmup, **kw):
    from neuron_interpolation_model_fusion_amd.models.base_model import twin_copy
    saved = twin_copy.replay_init_draws  # the runners switch the discarded initialisation off: put it back
    try:
        r = RUNNERS[workload](ops, device, steps, warmup, **kw)
    finally:
        twin_copy.replay_init_draws = saved
    return {
        'metric': 'sinkhorn solve wall-clock' if workload == 'sk' else 'end-to-end fusion wall-clock', 'value': r['seconds'], 'unit': 's', 'n_gpus': 1, 'steps': steps,
        'warmup': max(warmup, 1), 'ms_per_step': r['seconds'] * 1e3, 'higher_is_better': False, 'scaling': 'strong',
        'vs_baseline': None, 'dtype': 'f32 inputs; f64 normal equations / duals', 'data': 'synthetic',
        'config': dict(workload=r['workload'], **r['extra']), 'roofline': r['roofline'],
    }


def summary(ops, device, steps=2):
    """End-to-end wall-clock of the BASELINE.json configurations that fit one GPU, for bench.py's default line
    (config.e2e): seconds per whole fusion, mean of `steps` runs after one warm-up run each."""
    out = {}
    r = run('c1', ops, device, steps, 1)
    out['c1_kf_s'] = r['value']
    out['c1_hf_s'] = r['config']['hf2_seconds']
    out['c1_trd_hbm_frac'] = r['roofline']['frac']
    torch.cuda.empty_cache()
    r = run('c3', ops, device, steps, 1, hard=False)
    out['c3_s'] = r['value']
    out['c3_planted_permutation_recovered'] = r['config']['planted_permutation_recovered']
    torch.cuda.empty_cache()
    r = run('c4', ops, device, steps, 1)
    out['c4_s'] = r['value']
    torch.cuda.empty_cache()
    out['what'] = ('c1: LSFusion.fuse of random-init VGG11s, 400 samples (KF 4 models / HF 2 models); c3: GitRebasin.fuse '
                   'of two 784-8192-10 MLPs, 4096 samples, planted permutation; c4: ALF clustering + targets on the 7 hidden '
                   'levels of two ViTs, 5000 samples (SGD fit not timed); `python bench.py --workload c1|c3|c4` prints '
                   'each with its roofline')
    return out
