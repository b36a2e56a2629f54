"""HF / KF linear fusion on MI355X (reference: fusion_algorithms/hf_kf.py).

Same class, attributes and ``fuse`` signature.  What changed underneath, per layer:

* LS refit (hf_kf.py:213,233-234 / 377,397: ``torch.linalg.pinv(X) @ acts``): normal equations
  built with the f64 MFMA Gram kernel and solved through an eigendecomposition truncated at
  pinv's own cutoff (ops.ls_solve) -- X is never inverted and never leaves the device.
* HF cost (hf_kf.py:93-124: n0*n1/64 chunked (m x d) @ (d x 64) matmuls): closed form on the
  PROJECTED activations.  With P = X pinv(X), p = P a:  pred - a0_i = b (p1_j - p0_i)
  - (a0_i - p0_i), the two parts orthogonal, so C_ij = s0_i sqrt(b^2 D_ij + r0_i) + s1_j sqrt(a^2
  D_ij + r1_j) with D = ||p0_i - p1_j||^2 and r the squared LS residuals.  All of it comes from
  R = X^T A and W: P0^T P1 = R0^T W1 (a d-contraction), ||p_i||^2 = <R_i, W_i>, r = ||a||^2 - ||p||^2.
* assignment (hf_kf.py:251): device LAP kernel in scipy order (identical matching under ties).
* KF k-means (hf_kf.py:405-490): Gram form on G = R^T W, every step one pass over that N x N
  matrix (ops.KMeans); the farthest-point initialisation reads one Gram row per step instead of
  recomputing a c x N x m cdist.
"""
import os
import time

import numpy as np
import torch

from .. import ops
from ..models.base_model import twin_copy
from ..utils.fusion_utils import (
    get_num_layers, initialize_activations, get_layer_width, compute_neuron_importance_scores)
from ..utils.model_utils import is_conv_weight

COST_MATRIX_VECTORIZED_CHUNK_SIZE = 8  # kept for source compatibility (hf_kf.py:19); unused by the closed form


def _levels_follow_layers(model, num_layers):
    """True when the model can be walked level by level in place of layer by layer: it implements forward_level, has
    as many levels as trainable layers, and level i holds trainable layer i and no other."""
    from ..models.base_model import BaseModel
    if type(model).forward_level is BaseModel.forward_level:
        return False
    try:
        levels = model.get_ordered_trainable_named_levels
        layers = model.get_ordered_trainable_named_layers
    except Exception:  # a model without levels
        return False
    if len(levels) != num_layers or len(layers) != num_layers:
        return False
    trainable = {id(rep.layer) for rep in layers}
    for level, rep in zip(levels, layers):
        inside = {id(mod) for mod in level.network.modules()} & trainable
        if inside != {id(rep.layer)}:
            return False
    return True


class LSFusion:
    fusion_methods = ['hungarian', 'k_means']
    neuron_importance_methods = ['uniform', 'conductance', 'deeplift', 'featureablation']

    def __init__(self, models):
        self.models = models
        self.neuron_importance_scores = None
        self.last_layer_info = []  # per layer of the last fuse(): dict(cost, col) or dict(assign)

    @property
    def device(self):
        return self.models[0].device

    def _config_check(self, fusion_method, neuron_importance_method):
        if fusion_method not in self.fusion_methods:
            raise ValueError(f'Invalid fusion importance method: {fusion_method}')
        if neuron_importance_method not in self.neuron_importance_methods:
            raise ValueError(f'Invalid neuron importance method: {neuron_importance_method}')

    # -- shared per-layer set-up --------------------------------------------------------------------
    @staticmethod
    def _design_matrix(fused_hat, layer_idx):
        """Layer input as (m, d): im2col rows for conv, ones column for the bias
        (hf_kf.py:196-208); row stride padded for 16-byte loads."""
        rep = fused_hat.get_ordered_trainable_named_layers[layer_idx]
        X = fused_hat.get_inputs_and_preactivations(layer_idx)[0]
        in_filters = None
        use_bias = rep.layer.bias is not None
        if is_conv_weight(rep.weight):
            conv = rep.layer
            in_filters = conv.in_channels
            if tuple(conv.dilation) != (1, 1):
                raise ValueError('dilated convolutions are not supported by the LS refit')
            Xp = ops.design_matrix(X, conv.kernel_size, conv.stride, conv.padding, ones=use_bias)
        else:
            Xp = ops.design_matrix(X, ones=use_bias)
        return Xp, in_filters, use_bias

    def _layer_acts(self, models, layer_idx, rows, is_conv):
        """(rows, sum of widths) f32: every model's pre-activations side by side, conv maps as (B, H, W, C) rows
        (hf_kf.py:224-225) -- each model is copied once, straight into its column slice."""
        parts = [m.get_inputs_and_preactivations(layer_idx)[1] for m in models]
        widths = [p.shape[1] for p in parts]
        out = torch.empty((rows, sum(widths)), dtype=torch.float32, device=parts[0].device)
        off = 0
        for p, w in zip(parts, widths):
            if is_conv:
                B, _, H, W = p.shape
                out.view(B, H, W, -1)[..., off:off + w].copy_(p.permute(0, 2, 3, 1))
            else:
                out[:, off:off + w].copy_(p.reshape(rows, w))
            off += w
        return out

    @staticmethod
    def _split_rows(rows, fused_hat, layer_idx, use_bias, is_conv, in_filters):
        """(n, d) fused neurons -> (new_weight, new_bias) in the layer's own shapes (hf_kf.py:266-279);
        the bias lives on the CPU like the reference's (:260,402)."""
        n = rows.shape[0]
        new_bias = None
        if use_bias:
            new_bias = rows[:, -1].clone().cpu()
            rows = rows[:, :-1]
        new_weight = rows.reshape(n, in_filters, -1) if is_conv else rows
        return new_weight.contiguous(), new_bias

    @staticmethod
    def _get_cost_matrix_vectorized_chunked(X, acts0, acts1, w0, w1, s0, s1, chunk_size=COST_MATRIX_VECTORIZED_CHUNK_SIZE):
        """HF cost matrix for given LS solutions w0 (d, n0), w1 (d, n1); same arguments and numpy
        result as the reference's static method (hf_kf.py:93-124), closed form on the device."""
        R, _, na = ops.xty(X, torch.cat([acts0, acts1], dim=1).contiguous(), acc='f64')
        W = torch.cat([w0, w1], dim=1).double().contiguous()
        n0 = w0.shape[1]
        G01, _, _ = ops.xty(R[:, :n0], W[:, n0:], acc='f64', norms=False)
        pp = (R * W).sum(0)
        r = (na - pp).clamp_min(0)
        return ops.cost_hf(G01, pp[:n0], pp[n0:], r[:n0], r[n0:], s0, s1).cpu().numpy()

    # -- HF ---------------------------------------------------------------------------------------------
    def _fuse_layer_hungarian(self, layer_idx, fused_hat, norm_acts):
        """hf_kf.py:186-281."""
        layers = fused_hat.get_ordered_trainable_named_layers
        is_conv = is_conv_weight(layers[layer_idx].weight)
        X, in_filters, use_bias = self._design_matrix(fused_hat, layer_idx)
        dev = X.device
        start = time.time()
        acts = self._layer_acts(self.models[:2], layer_idx, X.shape[0], is_conv)
        sol = ops.ls_solve(X, acts, normalize=norm_acts)
        print(f'Pseudo-inverse time: {time.time() - start}')
        s0 = self.neuron_importance_scores[0][layer_idx].clone().to(dev)
        s1 = self.neuron_importance_scores[1][layer_idx].clone().to(dev)
        n0 = get_layer_width(self.models[0], layer_idx)
        W, R, na = sol['W'], sol['R'], sol['na']

        start = time.time()
        G01, _, _ = ops.xty(R[:, :n0], W[:, n0:], acc='f64', norms=False)  # P0^T P1
        pp = (R * W).sum(0)                                                  # ||p_i||^2
        r = (na - pp).clamp_min(0)                                           # squared LS residuals
        cost = ops.cost_hf(G01, pp[:n0], pp[n0:], r[:n0], r[n0:], s0, s1)
        print(f'Cost matrix time: {time.time() - start}')
        print()
        col = ops.lap_solve(cost, scipy_order=True)
        self.last_layer_info.append(dict(cost=cost, col=col, W=W))
        if layer_idx == len(layers) - 1:
            col = torch.arange(n0, device=dev, dtype=torch.int32)  # identity on the output layer (:254-255)
        rows = ops.refit_pairs(W, col, s0, s1, n0)
        return self._split_rows(rows, fused_hat, layer_idx, use_bias, is_conv, in_filters)

    # -- KF ---------------------------------------------------------------------------------------------
    def _fuse_layer_k_means(self, layer, fused_hat, norm_acts, num_k_means_tries=10,
                            k_means_early_stop_minimum_improvement=0.05, k_means_early_stop_steps_check=3):
        """hf_kf.py:345-510."""
        layers = fused_hat.get_ordered_trainable_named_layers
        is_conv = is_conv_weight(layers[layer].weight)
        is_last_layer = layer == len(layers) - 1
        output_size = layers[layer].weight.shape[0]
        X, in_filters, use_bias = self._design_matrix(fused_hat, layer)
        dev = X.device
        start = time.time()
        acts = self._layer_acts(self.models, layer, X.shape[0], is_conv)
        sol = ops.ls_solve(X, acts, normalize=norm_acts)
        print(f'Pseudo-inverse time: {time.time() - start}')
        s = torch.cat([self.neuron_importance_scores[k][layer].clone() for k in range(len(self.models))], dim=0).to(dev)
        W, R, na = sol['W'], sol['R'], sol['na']

        identity = torch.cat([torch.arange(output_size, dtype=torch.int32) for _ in self.models]).to(dev)
        if is_last_layer:
            km = ops.KMeans(None, na, s, output_size)  # grouping only: one cluster per output unit (:405-406)
            best_assignment = identity
        else:
            GP, _, _ = ops.xty(R, W, acc='f64', norms=False)  # (XW)^T (XW) = R^T W
            GP, gdiag = ops.kmeans_gram_storage(GP)
            T = num_k_means_tries
            best_assignment, best_total_cost = identity, np.inf
            start = time.time()
            # The tries are independent and share every matrix, so they advance in lock step as ONE batch: a round is
            # one launch per kernel for all the tries still alive (nif_kmeans_lloyd_step_batch / repair_kf_batch) and
            # ONE read-back of their costs / change flags, where the reference's loop (and a try-by-try port) launches
            # and synchronises per try and step.  Each try keeps the exact control flow of hf_kf.py:448-475; the
            # first-centroid draws come from np.random in try order.
            t0 = time.time()
            firsts = [int(np.random.randint(low=0, high=output_size)) for _ in range(T)]  # (:424)
            finals = [None] * T
            logs = [[] for _ in range(T)]
            it_init = T * (output_size - 1)
            t_init, t_conv, it_conv = 0.0, 0.0, 0
            for lo in range(0, T, 16):  # 16 tries per batch
                ids = list(range(lo, min(T, lo + 16)))
                km = ops.KMeans(GP, na, s, output_size, gdiag=gdiag, tries=len(ids))
                _, A = km.farthest_init_many([firsts[t] for t in ids])
                torch.cuda.synchronize(dev)
                t_init += time.time() - t0
                t0 = time.time()
                live = list(range(len(ids)))
                while live:
                    it_conv += len(live)
                    out, nearest, cost, _ = km.lloyd_step_many(A, live)
                    unrepaired = out.clone()  # kept for a try whose step turns out to be its early stop
                    changed = km.repair_kf_many(out, nearest, A, live)
                    flags = torch.stack([cost, changed.double()], dim=1).tolist()
                    still = []
                    for e in live:
                        current_cost, chg = flags[e]
                        log = logs[ids[e]]
                        if len(log) >= k_means_early_stop_steps_check and current_cost >= \
                                log[-k_means_early_stop_steps_check] * (1.0 - k_means_early_stop_minimum_improvement):
                            print("K-means Early Stopping at ", len(log), " iterations since ", current_cost, " >= ",
                                  log[-k_means_early_stop_steps_check], "*",
                                  (1.0 - k_means_early_stop_minimum_improvement))
                            finals[ids[e]] = unrepaired[e].clone()  # before the repair (:460-462)
                            continue
                        log.append(current_cost)
                        if chg:
                            still.append(e)
                        else:
                            finals[ids[e]] = out[e].clone()
                    A, live = out, still  # (rows of tries that have ended are never read again)
                F = torch.stack([finals[t] for t in ids])
                totals = km.lloyd_step_many(F)[2].tolist()
                for t, total_cost in zip(ids, totals):
                    if total_cost < best_total_cost:
                        best_assignment, best_total_cost = finals[t], total_cost
                t_conv += time.time() - t0
                t0 = time.time()
            print(f'K-means time: {(time.time() - start):.2f}, Centroid Initialization Time: {t_init:.2f}, '
                  f'Centroid Initialization Iterations: {it_init}, Centroid Convergence Time: {t_conv:.2f}, '
                  f'Centroid Convergence Iterations: {it_conv}')
        print()
        self.last_layer_info.append(dict(assign=best_assignment, W=W))
        rows, _ = km.centroid_weights(W, best_assignment)
        return self._split_rows(rows, fused_hat, layer, use_bias, is_conv, in_filters)

    def fuse(self, activation_dataset, activation_labels, fusion_method='hungarian', neuron_importance_method='uniform',
             norm_acts=False, norm_neuron_importance=True):
        """hf_kf.py:512-569."""
        self._config_check(fusion_method, neuron_importance_method)
        num_layers = get_num_layers(self.models)
        start_time = time.time()
        self.neuron_importance_scores = [[] for _ in range(len(self.models))]
        for k, model in enumerate(self.models):
            for layer_idx in range(num_layers):
                scores = compute_neuron_importance_scores(neuron_importance_method, model, layer_idx,
                                                          activation_dataset, activation_labels, fusion_logic='layer')
                if norm_neuron_importance:
                    scores = scores / torch.sum(scores)
                self.neuron_importance_scores[k].append(scores)
        print(f'Neuron Importance Score Time: {time.time() - start_time}')

        for m in self.models:
            m.to(activation_dataset.device)
            initialize_activations(m, activation_dataset)
        with twin_copy():
            fused_hat = self.models[0].copy_model().eval()
        self.last_layer_info = []
        # The fused model's input of layer l changes with every layer fused before it; the reference re-runs the whole
        # fused model for each layer (hf_kf.py:549: L^2 / 2 layer evaluations that matter, L^2 that run).  Where level l
        # of the model is exactly trainable layer l plus parameter-free modules (VGG without batch norm, MLP), the
        # levels are walked once instead: level l runs on the output of level l - 1 to capture the layer's input, and
        # again after its weights are set -- the same modules on the same tensors, so the same numbers.
        walk = os.environ.get('NIF_HF_FULL_FORWARD', '0') != '1' and _levels_follow_layers(fused_hat, num_layers)
        cur = activation_dataset
        if walk:
            fused_hat.remove_hooks()
            fused_hat.register_hooks()
        for layer_idx in range(num_layers):
            print(f'Fusing layer {layer_idx + 1}/{num_layers}')
            if walk:
                with torch.no_grad():
                    fused_hat.forward_level(cur, layer_idx)  # the hook of layer l keeps its input
            else:
                initialize_activations(fused_hat, activation_dataset)  # the fused model's input changes every layer
            if fusion_method == 'hungarian':
                new_W, new_b = self._fuse_layer_hungarian(layer_idx, fused_hat, norm_acts)
            else:
                new_W, new_b = self._fuse_layer_k_means(layer_idx, fused_hat, norm_acts)
            fused_hat.set_weight_at_layer(layer_idx, new_W)
            if new_b is not None:
                fused_hat.set_bias_at_layer(layer_idx, new_b)
            if walk and layer_idx + 1 < num_layers:
                with torch.no_grad():
                    cur = fused_hat.forward_level(cur, layer_idx)

        for m in self.models:
            m.remove_hooks()
        fused_hat.remove_hooks()
        with twin_copy():
            return fused_hat.copy_model()
