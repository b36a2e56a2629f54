"""Git Re-Basin activation matching on MI355X (reference: fusion_algorithms/git_rebasin.py).

Same class, same ``fuse`` signature.  Per layer the reference builds an (m, n0, n1) broadcast
for the cost (git_rebasin.py:60-78), solves the assignment with scipy on the host (:112) and
multiplies dense 0/1 permutation matrices into the weights (:163-213).  Here the cost is the
closed form 2 s0 s1/(s0+s1) * ||relu(a0_i) - relu(a1_j)|| on top of one MFMA cross-Gram with
the ReLU applied while loading (ops.xty + ops.cost_dist), the assignment is the device LAP
kernel (ops.lap_solve), and the permutations stay index vectors: the weight write-back is one
gather fused with the importance-weighted average (ops.refit_gather).
"""
import torch

from .. import ops
from ..models.base_model import twin_copy
from ..utils.fusion_utils import (
    get_num_layers, initialize_activations, get_layer_weight, compute_neuron_importance_scores)


class GitRebasin:

    neuron_importance_methods = ['uniform', 'conductance', 'deeplift', 'featureablation']

    def __init__(self, model0, model1):
        self.model0 = model0
        self.model1 = model1
        self.neuron_importance_scores0 = None
        self.neuron_importance_scores1 = None
        self.permutations = []  # col_ind per layer of the last fuse() (device int32), for inspection
        self.cost_matrices = None  # set to [] before fuse() to keep every layer's cost matrix

    def _prepare_acts_based_alignment(self, dataset, labels, method):
        """Importance scores and one hooked forward of both ORIGINAL models (git_rebasin.py:29-46)."""
        num_layers = get_num_layers([self.model0, self.model1])
        self.neuron_importance_scores0, self.neuron_importance_scores1 = [], []
        for layer_idx in range(num_layers):
            self.neuron_importance_scores0.append(
                compute_neuron_importance_scores(method, self.model0, layer_idx, dataset, labels, fusion_logic='layer'))
            self.neuron_importance_scores1.append(
                compute_neuron_importance_scores(method, self.model1, layer_idx, dataset, labels, fusion_logic='layer'))
        initialize_activations(self.model0, dataset)
        initialize_activations(self.model1, dataset)

    @staticmethod
    def _rows(pre):
        """Pre-activations as (rows, width); the ReLU of git_rebasin.py:97-98 is applied by the
        Gram kernel while loading.  The Gram is a sum over rows, so any fixed row order works;
        (b, h, w) is the reference's (:99-101)."""
        if pre.dim() == 4:
            return pre.permute(0, 2, 3, 1).reshape(-1, pre.shape[1])
        return pre

    @staticmethod
    def _get_cost_matrix_vectorized(acts0, acts1, s0, s1):
        """Cost matrix of post-ReLU activations acts0 (m, n0), acts1 (m, n1); same arguments as
        the reference's static method (git_rebasin.py:60-78), result on the CPU."""
        G, na, nb = ops.xty(acts0, acts1)
        return ops.cost_dist(G, na, nb, s0, s1).cpu()

    def _layer_cost(self, layer, device):
        """Cost matrix of one layer (git_rebasin.py:92-109) from the activations captured once of both ORIGINAL models."""
        pre0 = self._rows(self.model0.get_inputs_and_preactivations(layer)[1]).to(device)
        pre1 = self._rows(self.model1.get_inputs_and_preactivations(layer)[1]).to(device)
        s0 = self.neuron_importance_scores0[layer].to(device)
        s1 = self.neuron_importance_scores1[layer].to(device)
        G, na, nb = ops.xty(pre0, pre1, relu_a=True, relu_b=True, rows_sharded=True)
        cost = ops.cost_dist(G, na, nb, s0, s1)
        if self.cost_matrices is not None:
            self.cost_matrices.append(cost)
        return cost

    def _match_all_layers(self, num_layers, device):
        """col_ind of every layer's min-cost matching of model1's neurons to model0's (git_rebasin.py:80-123).  The
        reference solves them inside its layer loop, but no layer's cost depends on another layer's matching (the
        activations are those of the un-fused models), so all costs are built first and the assignment problems --
        one workgroup each -- are solved side by side (ops.lap_solve_many), in scipy's order under exact ties."""
        costs = [self._layer_cost(layer, device) for layer in range(num_layers)]
        cols = ops.lap_solve_many(costs, scipy_order=True)
        n_last = costs[-1].shape[0]
        cols[-1] = torch.arange(n_last, device=device, dtype=torch.int32)  # identity on the output layer (:115-116)
        return cols

    def fuse(self, activations_dataset, activations_labels, neuron_importance_method, handle_skip=False):
        if neuron_importance_method not in self.neuron_importance_methods:
            raise ValueError(f'Invalid neuron importance method: {neuron_importance_method}')
        num_layers = get_num_layers([self.model0, self.model1])
        device = activations_dataset.device
        self._prepare_acts_based_alignment(activations_dataset, activations_labels, neuron_importance_method)
        with twin_copy():
            fused_hat = self.model0.copy_model().eval()
        self.permutations = []

        cols = self._match_all_layers(num_layers, device)
        prev_is_conv, prev_col = False, None
        skip_col, residual_col, residual_idx = None, None, None
        for layer in range(num_layers):
            print(f'Fusing layer {layer + 1}/{num_layers}')
            w0, is_conv = get_layer_weight(self.model0, layer)
            w1, _ = get_layer_weight(self.model1, layer)
            b0 = self.model0.get_ordered_trainable_named_layers[layer].layer.bias
            b1 = self.model1.get_ordered_trainable_named_layers[layer].layer.bias
            s0 = self.neuron_importance_scores0[layer].to(device)
            s1 = self.neuron_importance_scores1[layer].to(device)
            col = cols[layer]

            in_col = prev_col
            if layer > 0 and is_conv and handle_skip:
                # ResNet bookkeeping of git_rebasin.py:172-186 with permutations kept as index vectors
                if w1.shape[1] != w1.shape[0]:
                    if w1.shape[2] != 1:
                        skip_col = prev_col.clone()
                    else:
                        residual_col, residual_idx = prev_col.clone(), layer
                        in_col = skip_col
                elif residual_col is not None and residual_idx == layer - 1:
                    if not torch.equal(prev_col, residual_col):
                        raise NotImplementedError(
                            'averaging two different permutations over a residual connection '
                            '(git_rebasin.py:185-186) does not give a permutation; only models without '
                            'skip connections are covered (the reference ResNet is not constructible)')

            # spatial positions per permuted input unit: kh*kw for conv, H*W of the last conv map for the
            # first linear layer after the conv stack, 1 otherwise (git_rebasin.py:189-201)
            if is_conv:
                sp = w1.shape[2]
            elif layer > 0 and prev_is_conv:
                sp = w1.shape[1] // in_col.shape[0]
            else:
                sp = 1
            s1p = s1[col.long()]  # P @ s1 (git_rebasin.py:204)
            a, b = s0 / (s0 + s1p), s1p / (s0 + s1p)
            n_out = w0.shape[0]
            new_W = ops.refit_gather(w0.reshape(n_out, -1).to(device), w1.reshape(n_out, -1).to(device), rowsrc=col,
                                     colsrc=in_col if layer > 0 else None, alpha=a, beta=b, sp=sp)
            fused_hat.set_weight_at_layer(layer, new_W)
            if b0 is not None:
                new_b = ops.refit_gather(b0.detach().reshape(n_out, 1).to(device),
                                         b1.detach().reshape(n_out, 1).to(device), rowsrc=col, alpha=a, beta=b)
                fused_hat.set_bias_at_layer(layer, new_b.reshape(n_out))
            self.permutations.append(col)
            prev_is_conv, prev_col = is_conv, col

        self.model0.remove_hooks()
        self.model1.remove_hooks()
        fused_hat.remove_hooks()
        with twin_copy():
            return fused_hat.copy_model()
