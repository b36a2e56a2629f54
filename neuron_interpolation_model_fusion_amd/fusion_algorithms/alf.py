"""ALF ("KF gradient fusion") on MI355X (reference: fusion_algorithms/alf.py).

Same ``ALFusionConfig`` fields, same ``ALFusion`` methods and signatures.  Per level the
reference (1) captures every model's level output, (2) clusters the neurons of all models with
a hand-rolled k-means over their activation vectors (alf.py:171-298: a k x N cdist over
m = samples x tokens rows plus a Python centroid loop per iteration), (3) optionally re-aligns
the targets to the level input with a Hungarian solve (:126-143), and (4) regresses the level
onto the targets with SGD/Adam (:493-594).

Here (2) runs in Gram form: one symmetric MFMA Gram of the (per-sample-normalised) activations,
then Lloyd steps that only stream that N x N matrix (ops.KMeans); the targets T are one
gather-reduce pass over the activations; (3) is a cross-Gram + the device LAP kernel in scipy
order.  (4) stays the reference's PyTorch loop: it is autograd training, not a
kernel of this path.
"""
import time
from dataclasses import dataclass

import torch
import torch.nn as nn
from torch.utils.data import Dataset, DataLoader
from tqdm import trange

from .. import ops
from ..models.base_model import BaseModel
from ..utils.fusion_utils import get_num_levels, normalize_activations_per_sample, compute_neuron_importance_scores
from ..utils.model_utils import (
    is_conv_activation, is_transformer_activation, reset_parameters, soft_reset_parameters, disable_dropout)


@dataclass
class ALFusionConfig:
    """Configuration of ALFusion.fuse (alf.py:29-55; same fields, same defaults)."""

    X: torch.Tensor
    y: torch.Tensor | None
    optimizer_per_level: list[int, str]
    lr_per_level: list[float]
    weight_decay_per_level: list[float]
    epochs_per_level: list[int]
    use_scheduler: bool = False
    neuron_importance_method: str = 'uniform'
    normalize_neuron_importances: bool = False
    align_skip_connections: bool = False
    train_batch_size: int = 32
    max_patience_epochs: int = 15
    normalize_acts_for_kmeans: bool = False
    num_kmeans_runs: int = 1
    init_kmeans_with_2nd_acts: bool = True
    use_weighted_loss: bool = True
    cls_head_weights: torch.Tensor | None = None
    use_kd_on_head: bool = False
    kd_temperature: float = 2.0
    skip_levels: list[int] = None
    reset_level_parameters: list[int] = None
    weight_perturbation_per_level: list[float] | None = None
    verbose: bool = True


class ActivationDataset(Dataset):
    """(level input, level target) pairs (alf.py:58-72)."""

    def __init__(self, X, T):
        self.X, self.T = X, T

    def __len__(self):
        return self.X.shape[0]

    def __getitem__(self, idx):
        return self.X[idx], self.T[idx]


class ALFusion:
    neuron_importance_methods = ['uniform', 'conductance', 'deeplift', 'featureablation']

    def __init__(self, models):
        self.models = models
        self.ni_scores = []
        self.weight_change_per_level = []

    @property
    def device(self):
        return self.models[0].device

    def _config_check(self, fusion_config):
        method = fusion_config.neuron_importance_method
        if method not in self.neuron_importance_methods:
            raise ValueError(f'Invalid neuron importance method: {method}')

    # -- layouts (alf.py:104-124) ------------------------------------------------------------------------
    def _flatten_acts(self, acts):
        if acts.dim() == 2:
            return acts
        if acts.dim() == 3:  # ViT: [B, hidden, seq] -> (B*seq, hidden)
            return acts.permute(0, 2, 1).reshape(-1, acts.shape[1])
        if acts.dim() == 4:  # conv: [B, C, H, W] -> (B*W*H, C), W-major
            return acts.permute(0, 3, 2, 1).reshape(-1, acts.shape[1])
        raise ValueError(f'Invalid shape for activations: {acts.shape}. Expected a 2D, 3D or 4D tensor.')

    def _unflatten_target(self, T, target_width, height, width, seqlen):
        if height != 0 and width != 0:
            return T.reshape(-1, width, height, target_width).permute(0, 3, 2, 1)
        if seqlen != 0:
            return T.reshape(-1, seqlen, target_width).permute(0, 2, 1)
        return T

    # -- target / input matching (alf.py:126-143) ----------------------------------------------------------
    def _match_T_to_X(self, X, T, W, chunk_size=64):
        """aligned_T[:, r] = T[:, c] for the min-cost matching of cost[i][j] = ||X[:, i] - T[:, j]|| * W[j]."""
        X = self._flatten_acts(X).contiguous()
        T = T.contiguous()
        G, na, nb = ops.xty(X, T, rows_sharded=True)
        cost = ops.cost_dist(G, na, nb, colscale=W)
        col = ops.lap_solve(cost, scipy_order=True)
        aligned = torch.zeros_like(T)
        aligned[:, :col.shape[0]] = T[:, col.long()]
        return aligned

    @staticmethod
    def _calculate_centroids(output_size, assignment, acts, weights=None):
        """Rows = neurons: (weighted) mean of every non-empty cluster's rows, label order (alf.py:145-169).
        Kept for API compatibility (small inputs: the class-wise logit mean of the last level)."""
        assignment = torch.as_tensor(assignment, device=acts.device).long()
        w = torch.ones(acts.shape[0], device=acts.device) if weights is None else weights.to(acts.device).float()
        sums = torch.zeros(output_size, acts.shape[1], device=acts.device).index_add_(0, assignment, acts * w[:, None])
        tot = torch.zeros(output_size, device=acts.device).index_add_(0, assignment, w)
        live = tot > 0
        return sums[live] / tot[live][:, None]

    # -- neuron clustering (alf.py:171-298) ------------------------------------------------------------------
    @staticmethod
    def _repair_on_host(assignment, nearest, target_width):
        """alf.py:247-262 verbatim on host copies: the order in which empty clusters are served is the
        pop order of a Python set, which only Python reproduces.  Rare path (a cluster went empty)."""
        assignment = assignment.cpu().long()
        nearest = nearest.cpu().clone()
        counts = torch.bincount(assignment, minlength=target_width)
        unassigned_clusters = set(range(target_width)) - set(assignment.tolist())
        while unassigned_clusters:
            i = unassigned_clusters.pop()
            far = torch.max(nearest, dim=0)[1]
            if counts[assignment[far].item()] == 1:
                nearest[far] = float('-inf')
                unassigned_clusters.add(i)
                continue
            counts[assignment[far].item()] -= 1
            assignment[far] = i
            nearest[far] = float('-inf')
        nearest = torch.where(nearest == float('-inf'), torch.tensor(0.0, dtype=nearest.dtype), nearest)
        return assignment, nearest

    def _k_means_v2(self, level_idx, target_width, fusion_config, level_X):
        X = fusion_config.X
        dev = self.device
        seqlen, height, width = 0, 0, 0
        with torch.no_grad():
            acts = [m.forward_until_level(X, level_idx) for m in self.models]
        match_T_to_level_X = fusion_config.align_skip_connections and (acts[0].shape == level_X.shape)
        if is_conv_activation(acts[0]):
            height, width = acts[0].shape[2], acts[0].shape[3]
        elif is_transformer_activation(acts[0]):
            seqlen = acts[0].shape[2]
        acts = [self._flatten_acts(a) for a in acts]
        if fusion_config.normalize_acts_for_kmeans:
            kmeans_acts = torch.cat(tuple(normalize_activations_per_sample(a) for a in acts), dim=1).contiguous()
        else:
            kmeans_acts = None
        acts = torch.cat(tuple(acts), dim=1).contiguous()
        if kmeans_acts is None:
            kmeans_acts = acts

        ni_scores_of_level = [self.ni_scores[k][level_idx] for k in range(len(self.models))]
        sample_weight = torch.cat(ni_scores_of_level).to(dev)
        sw64 = sample_weight.double()

        # one symmetric Gram of the clustering vectors; everything below streams it
        G, _, _ = ops.xty(kmeans_acts, None, acc='f32', norms=False, rows_sharded=True)
        km = ops.KMeans(G, torch.diagonal(G).contiguous(), sample_weight, target_width)

        time_cc, iters = 0.0, 0
        best_total_cost, best_assignment = float('inf'), None
        for _ in range(fusion_config.num_kmeans_runs):
            c0 = target_width if fusion_config.init_kmeans_with_2nd_acts else 0
            assignment = km.assign_to_neurons(c0)
            start = time.time()
            costs_log = []
            changed = True
            while changed:
                iters += 1
                prev = assignment
                assignment, nearest, cost, _ = km.lloyd_step(prev)
                n_empty = target_width - torch.unique(assignment).numel()
                if n_empty > 0:
                    a_host, d_host = self._repair_on_host(assignment, nearest, target_width)
                    assignment = a_host.to(device=dev, dtype=torch.int32)
                    current_cost = float(torch.sum(d_host.to(dev) * sw64).item())
                else:
                    current_cost = float(cost.item())
                changed = bool((assignment != prev).any().item())
                if len(costs_log) >= 3 and current_cost >= costs_log[-3]:
                    print("K-means Early Stopping at ", len(costs_log), " iterations since ", current_cost, " >= ",
                          costs_log[-3], "*", 1.0)
                    break
                costs_log.append(current_cost)
            _, _, total, _ = km.lloyd_step(assignment)
            total_cost = float(total.item())
            if total_cost < best_total_cost:
                best_assignment, best_total_cost = assignment, total_cost
            time_cc += time.time() - start

        # targets from the UN-normalised activations (alf.py:288-289)
        T, kprime = km.targets(acts, best_assignment)
        T = T[:, :int(kprime.item())]
        if match_T_to_level_X:
            total_importances = torch.zeros(T.shape[1], device=dev).index_add_(
                0, best_assignment.long(), sample_weight.float())
            T = self._match_T_to_X(level_X, T, total_importances)
        T = self._unflatten_target(T, target_width, height, width, seqlen)
        return T, best_assignment.long()

    # -- last level, losses, optimiser (alf.py:398-491) ----------------------------------------------------
    def _last_level_grouping(self, num_classes, fusion_config):
        all_model_logits = []
        with torch.no_grad():
            for i, m in enumerate(self.models):
                logits = m(fusion_config.X)
                if fusion_config.cls_head_weights is not None:
                    logits = logits * fusion_config.cls_head_weights[i]
                if fusion_config.use_kd_on_head:
                    logits = (logits / fusion_config.kd_temperature).softmax(dim=1)
                all_model_logits.append(logits)
        all_model_logits = torch.cat(tuple(all_model_logits), dim=1)
        assignment = torch.cat([torch.arange(start=0, end=num_classes) for _ in range(len(self.models))])
        T = self._calculate_centroids(num_classes, assignment, all_model_logits.T).T
        return T, assignment

    def _criterion_vectorized(self, act_pred, act_true, cluster_weights):
        return (((act_pred - act_true) ** 2) * cluster_weights).mean()

    def _get_criterion(self, level_idx, target_width, assignment, fusion_config, is_last_level):
        if not fusion_config.use_weighted_loss or (is_last_level and fusion_config.cls_head_weights is None):
            return nn.MSELoss().to(self.device)
        if is_last_level and fusion_config.use_kd_on_head:
            return nn.KLDivLoss(reduction='batchmean').to(self.device)
        ni_scores = torch.cat([self.ni_scores[i][level_idx] for i in range(len(self.models))], dim=0)
        ni_scores = ni_scores.detach().to(self.device)
        cluster_weights = torch.zeros(target_width, device=self.device).index_add_(
            0, torch.as_tensor(assignment, device=self.device).long(), ni_scores.float())

        def weighted_loss(y_pred, y_true):
            return self._criterion_vectorized(y_pred, y_true, cluster_weights)
        return weighted_loss

    def _get_optimizer(self, level_idx, level_network, fusion_config):
        lr = fusion_config.lr_per_level[level_idx]
        weight_decay = fusion_config.weight_decay_per_level[level_idx]
        optim_name = fusion_config.optimizer_per_level[level_idx].lower()
        if fusion_config.verbose:
            print(f'Optimizer for level {level_idx}: {optim_name}. Learning rate: {lr}. Weight decay: {weight_decay}.')
        if optim_name == 'adam':
            return torch.optim.Adam(level_network.parameters(), lr=lr, weight_decay=weight_decay)
        if optim_name == 'sgd':
            return torch.optim.SGD(level_network.parameters(), lr=lr, weight_decay=weight_decay)
        raise ValueError(f'Invalid optimizer: {optim_name}')

    def _get_level_input(self, level_idx, fused_hat, fusion_config):
        if level_idx == 0:
            return fusion_config.X
        fused_hat.eval()
        with torch.no_grad():
            return fused_hat.forward_until_level(fusion_config.X, level_idx - 1)

    # -- per-level fit (alf.py:493-594): the reference's training loop, unchanged in behaviour ---------------
    def _fuse_level_sgd(self, level_idx, fused_hat, fusion_config):
        level_X = self._get_level_input(level_idx, fused_hat, fusion_config)
        level = fused_hat.get_ordered_trainable_named_levels[level_idx]
        target_width, network = level.output_width, level.network
        weights_before = nn.utils.parameters_to_vector(network.parameters()).detach().clone()

        num_levels = get_num_levels(self.models)
        is_last_level = level_idx == num_levels - 1
        if not is_last_level:
            T, assignment = self._k_means_v2(level_idx, target_width, fusion_config, level_X)
        else:
            T, assignment = self._last_level_grouping(target_width, fusion_config)

        dataset = ActivationDataset(level_X, T)
        train_dataset, val_dataset = torch.utils.data.random_split(dataset, [0.8, 0.2])
        train_dl = DataLoader(train_dataset, batch_size=fusion_config.train_batch_size, shuffle=True)
        val_dl = DataLoader(val_dataset, batch_size=fusion_config.train_batch_size, shuffle=False)

        if level_idx in (fusion_config.reset_level_parameters or []):
            reset_parameters(network)
        elif fusion_config.weight_perturbation_per_level is not None \
                and fusion_config.weight_perturbation_per_level[level_idx] > 0:
            soft_reset_parameters(network, fusion_config.weight_perturbation_per_level[level_idx])

        network.train().float()
        disable_dropout(network)
        optimizer = self._get_optimizer(level_idx, network, fusion_config)
        scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=50)
        criterion = self._get_criterion(level_idx, target_width, assignment, fusion_config, is_last_level)

        patience = fusion_config.max_patience_epochs
        best_val_loss = float('inf')
        num_epochs = fusion_config.epochs_per_level[level_idx]
        for epoch in range(num_epochs):
            train_loss = 0.0
            network.train(True)
            for x, y in train_dl:
                x, y = x.to(self.device).float(), self._flatten_acts(y).to(self.device).float()
                optimizer.zero_grad()
                loss = criterion(self._flatten_acts(network(x)), y)
                loss.backward()
                train_loss += loss.item()
                optimizer.step()
            if fusion_config.use_scheduler:
                scheduler.step(train_loss)  # the reference passes the loss as the epoch argument (alf.py:559)

            val_loss = 0.0
            network.eval()
            with torch.no_grad():
                for x, y in val_dl:
                    x, y = x.to(self.device).float(), self._flatten_acts(y).to(self.device).float()
                    val_loss += criterion(self._flatten_acts(network(x)), y).item()
            avg_train_loss = train_loss / len(train_dl)
            avg_val_loss = val_loss / len(val_dl)
            if fusion_config.verbose:
                print(f'Epoch: {epoch + 1}/{num_epochs}.    Avg Train Loss: {avg_train_loss:.4f}.    '
                      f'Avg Val Loss: {avg_val_loss:.4f}.')
            if avg_val_loss >= best_val_loss:
                patience -= 1
                if patience == 0:
                    if fusion_config.verbose:
                        print('Early stopping')
                    break
            else:
                best_val_loss = avg_val_loss
                patience = fusion_config.max_patience_epochs

        weights_after = nn.utils.parameters_to_vector(network.parameters()).detach().clone()
        weight_change = torch.norm(weights_after - weights_before).item()
        self.weight_change_per_level.append(weight_change)
        if fusion_config.verbose:
            print(f'Weight change for level {level_idx}: {weight_change:.4f}')

    def fuse(self, fusion_config, fused_hat=None):
        """alf.py:596-638; fuses IN PLACE into fused_hat (a copy of models[0] when None)."""
        self._config_check(fusion_config)
        num_levels = get_num_levels(self.models)
        start_time = time.time()
        self.ni_scores = [{} for _ in range(len(self.models))]
        self.weight_change_per_level = []
        for k, model in enumerate(self.models):
            for level_idx in range(num_levels):
                if level_idx in (fusion_config.skip_levels or []):
                    continue
                scores = compute_neuron_importance_scores(fusion_config.neuron_importance_method, model, idx=level_idx,
                                                          X=fusion_config.X, y=fusion_config.y, fusion_logic='level')
                if fusion_config.normalize_neuron_importances:
                    scores = scores / torch.sum(scores)
                self.ni_scores[k][level_idx] = scores
        if fusion_config.verbose:
            print(f'Neuron Importance Score Time: {time.time() - start_time:.2f} seconds')
        if fused_hat is None:
            fused_hat = self.models[0].copy_model()
        for level_idx in trange(num_levels, desc='Fusing levels'):
            if level_idx in (fusion_config.skip_levels or []):
                continue
            self._fuse_level_sgd(level_idx, fused_hat, fusion_config)
        return fused_hat
