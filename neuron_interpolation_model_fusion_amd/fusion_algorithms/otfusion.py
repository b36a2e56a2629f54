"""OTFusion on MI355X (reference: fusion_algorithms/otfusion.py).

Same class and ``fuse_models`` signature / alignment_strategy schema (run_otf.py:100-109).
The reference computes the ground metric with torch.cdist on the CPU (otfusion.py:150) and the
plan with POT's network simplex ``ot.emd`` on the CPU (:32).  With uniform marginals on a
square problem -- what 'uniform' importance produces for equal-width models (:242,258) -- an
optimal plan is a vertex P/n of the Birkhoff polytope, so the exact plan comes from the device
assignment kernel; plans stay index vectors and the two T-matrix products of a layer (edge
alignment :261-306, barycentric projection :313-335) collapse into one gather fused with the
running sum.  Non-uniform importances or unequal widths go through the exact transport kernel
(nif_emd_solve_f32); its plans have at most n_k + n_f - 1 entries and stay sparse, and the same
two products become one sparse gather (nif_refit_sparse).
"""
import os
import time

import torch

from .. import ops
from ..models.base_model import twin_copy
from ..utils.fusion_utils import (
    get_num_layers, initialize_activations, get_layer_weight, get_layer_width, get_preactivations_of_layer,
    compute_neuron_importance_scores)
from ..utils.model_utils import is_conv_weight


def get_ot_matrix(cost, mu, nu):
    """Exact OT plan for a device cost matrix (dim1, dim2) with marginals mu, nu (otfusion.py:18-32).
    Returns the dense plan T (device, f32)."""
    perm, plan = _solve_plan(cost, mu, nu)
    if plan is None:
        plan = ops.SparsePlan.from_permutation(perm, float(mu[0]))
    return plan.dense()


def _solve_plan(cost, mu, nu):
    """(row4col, None) when the problem is an assignment (uniform marginals, equal widths: T = P / n),
    else (None, SparsePlan) from the exact transport kernel."""
    n0, n1 = cost.shape
    uniform = n0 == n1 and bool(torch.equal(mu, mu[:1].expand_as(mu)) and torch.equal(nu, nu[:1].expand_as(nu)))
    if uniform:
        _, row4col, _, _ = ops.lap_solve(cost, duals=True)
        return row4col, None
    return None, ops.emd_solve(cost, mu, nu)


def _as_sparse(perm, plan, value):
    return plan if plan is not None else ops.SparsePlan.from_permutation(perm, value)


class OTFusion:
    """Fuses models into a target model with optimal transport."""

    def __init__(self, models, target_model):
        self.models, self.target_model = models, target_model
        self.device = self.target_model.device
        self.alignment_strategy = None
        self.neuron_importance_scores = None
        self.target_neuron_importance_scores = None
        self.plans = []  # (layer, model, row4col) of the last fuse_models(), for inspection

    def _check_num_layers(self):
        num_layers = get_num_layers(self.models)
        target_num_layers = len(self.target_model.get_ordered_trainable_named_layers)
        if num_layers != target_num_layers:
            raise ValueError(f'The number of layers of the models to be fused is different '
                             f'compared to the target model: {num_layers} != {target_num_layers}')
        return target_num_layers

    def _rescaled_scores(self, method, model, layer_idx, dataset, labels, rescale):
        scores = compute_neuron_importance_scores(method, model, layer_idx, dataset, labels, fusion_logic='layer')
        if rescale is not None:  # otfusion.py:110-112
            scores = scores - torch.min(scores)
            scores = (scores / torch.max(scores)) * (1 - rescale) + rescale
        return scores

    def _prepare_acts_based_alignment(self, mode='acts'):
        """Importance scores of every model and the target; hooked forwards in 'acts' mode
        (otfusion.py:95-128)."""
        num_layers = self._check_num_layers()
        cfg = self.alignment_strategy[mode]
        dataset, labels, method = cfg['activations_dataset'], cfg['activation_labels'], cfg['neuron_importance_method']
        rescale = cfg['rescale_min_importance'] if 'rescale_min_importance' in cfg else None
        start_time = time.time()
        self.neuron_importance_scores = [
            [self._rescaled_scores(method, model, l, dataset, labels, rescale) for l in range(num_layers)]
            for model in self.models]
        self.target_neuron_importance_scores = [
            self._rescaled_scores(method, self.target_model, l, dataset, labels, rescale) for l in range(num_layers)]
        print(f'Neuron Importance Score Time: {time.time() - start_time}')
        if mode == 'acts':
            for model in self.models:
                self._initialize_activations(model)

    def _initialize_activations(self, model):
        initialize_activations(model, self.alignment_strategy['acts']['activations_dataset'])

    @staticmethod
    def _get_preactivation_cost(act1, act2, normalize=False):
        """Euclidean distance between neuron activation columns, act: (rows, width) on the device
        (otfusion.py:135-150; the reference moves both to the CPU first)."""
        G, na, nb = ops.xty(act1, act2, rows_sharded=True)
        if normalize:  # columns scaled to unit norm (fusion_utils.py:38-40) == cosine form of the Gram
            G = G / (na.sqrt()[:, None] * nb.sqrt()[None, :])
            na, nb = torch.ones_like(na), torch.ones_like(nb)
        return ops.cost_dist(G, na, nb)

    def _get_preactivation_ground_metric(self, model, fused_model_hat, layer):
        act1, act2 = get_preactivations_of_layer(model, fused_model_hat, layer)
        return self._get_preactivation_cost(act1.to(self.device), act2.to(self.device),
                                            self.alignment_strategy['acts']['normalize_activations'])

    @staticmethod
    def _get_weight_ground_metric(aligned_weight, fused, layer):
        """Distance between the aligned incoming weights of model k's neurons and the fused model's
        (otfusion.py:163-182): rows are neurons, the contraction runs over the fan-in."""
        fused_weight = fused.get_ordered_trainable_named_layers[layer].weight.detach()
        a = aligned_weight.flatten(1).contiguous()
        b = fused_weight.flatten(1).contiguous()
        G, na, nb = ops.xty(a, b, trans_a=True, trans_b=True)
        return ops.cost_dist(G, na, nb)

    def _get_ground_metric(self, current_model, fused_model_hat, layer_idx, model_k_aligned_weight):
        if self.alignment_strategy['mode'] not in ['acts', 'wts']:
            raise ValueError('Only `acts` and `wts` modes are currently supported')
        if self.alignment_strategy['mode'] == 'acts':
            return self._get_preactivation_ground_metric(current_model, fused_model_hat, layer_idx)
        return self._get_weight_ground_metric(model_k_aligned_weight, fused_model_hat, layer_idx)

    def _get_input_dim(self):
        first_layer = self.target_model.get_ordered_trainable_named_layers[0].layer
        if is_conv_weight(first_layer.weight.data):
            return first_layer.in_channels
        return first_layer.in_features

    def fuse_models(self, alignment_strategy, handle_skip=False, handle_bias=True):
        """Aligns self.models to self.target_model layer by layer and averages (otfusion.py:193-361)."""
        [model.eval() for model in self.models]
        self.target_model.eval()
        self.alignment_strategy = alignment_strategy
        mode = alignment_strategy['mode']
        num_layers = self._check_num_layers()
        input_dim = self._get_input_dim()
        with twin_copy():
            fused_hat = self.target_model.copy_model().eval()
        dev = self.device
        use_scores = mode == 'acts' or alignment_strategy['wts']['neuron_importance_method'] != 'uniform'
        if use_scores:
            self._prepare_acts_based_alignment(mode)
        if mode == 'acts':
            self._initialize_activations(fused_hat)  # target activations are captured once (otfusion.py:223-224)
        if alignment_strategy.get('solver', 'emd') == 'sinkhorn':
            return self._fuse_models_entropic(fused_hat, num_layers, input_dim, use_scores, handle_bias)
        if alignment_strategy.get('solver', 'emd') != 'emd':
            raise ValueError("alignment_strategy['solver'] must be 'emd' (the reference's exact plan) or 'sinkhorn'")

        # plans of the previous layer per model, as (row4col, None) -- row4col[j'] = model-k unit transported
        # to fused unit j' -- or (None, SparsePlan); (None, None) = the identity coupling diag(nu_prev) of the
        # input (otfusion.py:216-217)
        # Activation mode: the ground metric of (layer, model) comes from activations captured once, so no plan depends
        # on another one: all are solved up front, side by side on their own streams (one workgroup each) -- the
        # assignment-shaped ones (uniform marginals, equal widths) by the assignment kernel, the rest by the exact
        # transport kernel.  Weight mode goes through _solve_plan layer by layer.
        presolved = {}
        if mode == 'acts' and os.environ.get('NIF_OTF_SERIAL_PLANS') != '1':  # (A/B aid: plans one by one in the loop)
            keys, costs, ekeys, eprobs = [], [], [], []
            for layer in range(num_layers):
                n_f = get_layer_width(fused_hat, layer)
                for k, model in enumerate(self.models):
                    n_k = get_layer_width(model, layer)
                    mu_, nu_ = self.neuron_importance_scores[k][layer], self.target_neuron_importance_scores[layer]
                    C = self._get_ground_metric(model, fused_hat, layer, None)
                    if n_k == n_f and bool(torch.equal(mu_, mu_[:1].expand_as(mu_))) and \
                            bool(torch.equal(nu_, nu_[:1].expand_as(nu_))):
                        keys.append((layer, k))
                        costs.append(C)
                    else:  # the marginals _solve_plan would be handed in the loop below
                        ekeys.append((layer, k))
                        mu_, nu_ = mu_.to(dev), nu_.to(dev)
                        eprobs.append((C, mu_ / mu_.sum(), nu_ / nu_.sum()))
            if costs:
                _, rows_ = ops.lap_solve_many(costs, scipy_order=False, rows=True)
                presolved = {key: (C, r, None) for key, C, r in zip(keys, costs, rows_)}
            if eprobs:
                plans_ = ops.emd_solve_many(eprobs)
                presolved.update({key: (pr[0], None, plan) for key, pr, plan in zip(ekeys, eprobs, plans_)})
        prev_plan = [(None, None) for _ in self.models]
        prev_width = [input_dim for _ in self.models]
        nu_prev = None
        prev_is_conv = False
        self.plans = []
        for layer in range(num_layers):
            print(f'Fusing layer {layer+1}/{num_layers}')
            total_weight, is_conv = get_layer_weight(fused_hat, layer)
            total_weight = total_weight.to(dev)
            n_f = total_weight.shape[0]
            flat_shape = total_weight.shape
            total_weight = total_weight.reshape(n_f, -1).contiguous()
            total_bias = None
            if handle_bias:
                total_bias = fused_hat.get_ordered_trainable_named_layers[layer].layer.bias.data.clone().detach()
                total_bias = total_bias.reshape(n_f, 1).to(dev)
            if use_scores:
                nu = self.target_neuron_importance_scores[layer].to(dev)
                nu = nu / nu.sum()
            else:
                nu = torch.ones(get_layer_width(fused_hat, layer), device=dev) / n_f

            for k, model in enumerate(self.models):
                model_k_weight, _ = get_layer_weight(model, layer)
                model_k_weight = model_k_weight.to(dev)
                n_k = get_layer_width(model, layer)
                if use_scores:
                    mu = self.neuron_importance_scores[k][layer].to(dev)
                    mu = mu / mu.sum()
                else:
                    mu = torch.ones(n_k, device=dev) / n_k

                # handle_skip only does ResNet bookkeeping in the reference (otfusion.py:266-281): on VGG / MLP
                # it stores a skip plan that is never read, so it changes nothing here.
                # sp = positions per transported input unit: kh*kw (conv), H*W of the last conv map (first
                # linear layer after the conv stack, otfusion.py:290-303), else 1
                if is_conv:
                    sp = model_k_weight.shape[2]
                elif prev_is_conv:
                    sp = model_k_weight.shape[1] // prev_width[k]
                else:
                    sp = 1
                wk_flat = model_k_weight.reshape(n_k, -1).contiguous()
                p_perm, p_sparse = prev_plan[k]
                # column side of the alignment, T_prev diag(1/nu_prev): an index vector, or a CSC triplet
                cols = p_sparse.scaled(col_scale=1.0 / nu_prev) if p_sparse is not None else None

                # ground metric; 'wts' mode needs the edge-aligned weight W_k T_prev diag(1/nu_prev) first
                if (layer, k) in presolved:
                    C = presolved[(layer, k)][0]
                elif mode == 'acts':
                    C = self._get_ground_metric(model, fused_hat, layer, None)
                else:
                    if cols is None:
                        edges = ops.refit_gather(None, wk_flat, colsrc=p_perm, alpha=0.0, beta=1.0, sp=sp)
                    else:
                        edges = ops.refit_sparse(None, wk_flat, cols=cols, sp=sp)
                    C = self._get_ground_metric(model, fused_hat, layer, edges)
                perm, sparse = presolved[(layer, k)][1:] if (layer, k) in presolved else _solve_plan(C, mu, nu)
                self.plans.append((layer, k, perm if sparse is None else sparse))
                last = layer == num_layers - 1  # T_corrected = I on the output layer (otfusion.py:315-316)
                if handle_bias:
                    bk = model.get_ordered_trainable_named_layers[layer].layer.bias.detach().reshape(n_k, 1).to(dev)

                # total += T^T diag(1/mu) (W_k T_prev diag(1/nu_prev)), accumulated in place by one gather:
                # index vectors when both plans are permutations, the sparse form otherwise
                if (last or sparse is None) and cols is None:
                    rowsrc = torch.arange(n_k, device=dev, dtype=torch.int32) if last else perm
                    total_weight = ops.refit_gather(total_weight, wk_flat, rowsrc=rowsrc, colsrc=p_perm,
                                                    alpha=1.0, beta=1.0, sp=sp)
                    if handle_bias:
                        total_bias = ops.refit_gather(total_bias, bk, rowsrc=rowsrc, alpha=1.0, beta=1.0)
                else:
                    rows = None if last else _as_sparse(perm, sparse, float(mu[0])).scaled(row_scale=1.0 / mu)
                    if cols is None and p_perm is not None:
                        cols = _as_sparse(p_perm, None, 1.0).scaled()  # P diag(1/nu_prev) of a permutation plan = P
                    total_weight = ops.refit_sparse(total_weight, wk_flat, rows=rows, cols=cols, sp=sp)
                    if handle_bias:
                        total_bias = ops.refit_sparse(total_bias, bk, rows=rows, cols=None, sp=1)
                prev_plan[k] = (perm, sparse)
                prev_width[k] = n_k

            scale = 1.0 / (len(self.models) + 1)
            fused_hat.set_weight_at_layer(layer, (total_weight * scale).reshape(flat_shape))
            if handle_bias:
                fused_hat.set_bias_at_layer(layer, (total_bias * scale).reshape(n_f))
            prev_is_conv = is_conv
            nu_prev = nu

        if mode == 'acts':
            for model in self.models:
                model.remove_hooks()
            self.target_model.remove_hooks()
            fused_hat.remove_hooks()
            with twin_copy():
                fused_hat = fused_hat.copy_model()
        return fused_hat

    def _fuse_models_entropic(self, fused_hat, num_layers, input_dim, use_scores, handle_bias):
        """The loop of otfusion.py:226-361 with DENSE plans from the library's Sinkhorn mode
        (alignment_strategy['solver'] = 'sinkhorn', ['sinkhorn_reg'] = POT's `reg`, optional ['sinkhorn_iters'],
        ['sinkhorn_tol']).  Not a reference path: the reference's plan is the exact ot.emd.  Dense plans make the
        alignment plain matrix products (library GEMMs), as the reference writes them."""
        st = self.alignment_strategy
        mode, dev = st['mode'], self.device
        reg = float(st['sinkhorn_reg'])
        iters, tol = int(st.get('sinkhorn_iters', 1000)), float(st.get('sinkhorn_tol', 1e-9))
        nu_prev = torch.ones(input_dim, device=dev) / input_dim
        T_prev = [torch.diag(nu_prev) for _ in self.models]
        prev_is_conv = False
        self.plans, self.sinkhorn_log = [], []
        for layer in range(num_layers):
            print(f'Fusing layer {layer+1}/{num_layers}')
            total_weight, is_conv = get_layer_weight(fused_hat, layer)
            total_weight = total_weight.to(dev).clone()
            n_f = total_weight.shape[0]
            total_bias = None
            if handle_bias:
                total_bias = fused_hat.get_ordered_trainable_named_layers[layer].layer.bias.data.clone().detach().to(dev)
            if use_scores:
                nu = self.target_neuron_importance_scores[layer].to(dev)
                nu = nu / nu.sum()
            else:
                nu = torch.ones(n_f, device=dev) / n_f
            for k, model in enumerate(self.models):
                wk, _ = get_layer_weight(model, layer)
                wk = wk.to(dev)
                n_k = get_layer_width(model, layer)
                if use_scores:
                    mu = self.neuron_importance_scores[k][layer].to(dev)
                    mu = mu / mu.sum()
                else:
                    mu = torch.ones(n_k, device=dev) / n_k
                Tp = T_prev[k] @ torch.diag(1.0 / nu_prev)  # (otfusion.py:283-303)
                if is_conv:
                    edges = torch.einsum('ois,ij->ojs', wk, Tp)
                elif prev_is_conv:
                    edges = torch.einsum('ois,ij->ojs', wk.view(n_k, Tp.shape[0], -1), Tp).flatten(1)
                else:
                    edges = wk @ Tp
                C = self._get_ground_metric(model, fused_hat, layer, edges if mode == 'wts' else None)
                res = ops.sinkhorn(C, mu, nu, reg, max_iter=iters, tol=tol)
                self.sinkhorn_log.append(dict(layer=layer, model=k, state=res['state'], iterations=res['iterations'],
                                              err=res['err']))
                T = res['T']
                self.plans.append((layer, k, T))
                T_prev[k] = T
                Tc = T.T @ torch.diag(1.0 / mu) if layer < num_layers - 1 else torch.eye(n_k, device=dev)
                total_weight += (Tc @ edges.flatten(1)).view_as(total_weight)
                if handle_bias:
                    bk = model.get_ordered_trainable_named_layers[layer].layer.bias.detach().to(dev)
                    total_bias += (Tc @ bk).view_as(total_bias)
            scale = 1.0 / (len(self.models) + 1)
            fused_hat.set_weight_at_layer(layer, total_weight * scale)
            if handle_bias:
                fused_hat.set_bias_at_layer(layer, total_bias * scale)
            nu_prev, prev_is_conv = nu, is_conv
        if mode == 'acts':
            for model in self.models:
                model.remove_hooks()
            self.target_model.remove_hooks()
            fused_hat.remove_hooks()
            with twin_copy():
                fused_hat = fused_hat.copy_model()
        return fused_hat
