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
order.  (4) is autograd training and stays PyTorch, but its data path is on the device:
batches are index gathers from resident tensors in the loader's exact sample order, and the
(cluster-weighted) squared-error loss is one fused forward and one fused backward pass.
"""
import math
import time
from dataclasses import dataclass

import torch
import torch.nn as nn
from torch.utils.data import Dataset
from tqdm import trange

from .. import ops
from ..models.base_model import BaseModel
from ..parallel import SampleSharding, allgather_rows, broadcast_parameters
from ..utils.fusion_utils import get_num_levels, compute_neuron_importance_scores
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


class _GraphedBatch:
    """Forward + loss (+ backward) of one full-size batch of a level fit, captured once in a HIP graph and
    replayed per batch: the fit of a narrow level is launch-bound (a few dozen small kernels per step).
    The optimiser step stays outside the graph, so Adam / SGD behave exactly as in eager mode.  The first
    ``warmup`` full batches run eagerly (library handles, kernel selection) before the capture."""

    def __init__(self, level_X, T, batch_size, train, warmup=2):
        self.x = torch.empty((batch_size,) + tuple(level_X.shape[1:]), dtype=level_X.dtype, device=level_X.device)
        self.y = torch.empty((batch_size,) + tuple(T.shape[1:]), dtype=T.dtype, device=T.device)
        self.loss = torch.zeros((), dtype=torch.float64, device=level_X.device)
        self.train = train
        self.warmup = warmup
        self.graph = None
        self.failed = False
        self.params, self.grads = [], []

    def ready(self, host, network, criterion):
        if self.failed:
            return False
        if self.graph is not None:
            return True
        if self.warmup > 0:
            self.warmup -= 1
            return False
        try:
            self._capture(host, network, criterion)
        except RuntimeError as e:  # e.g. a level whose forward synchronises: keep the eager loop
            print(f'[ALFusion] HIP graph capture of the level fit failed, continuing eagerly: {e}')
            self.graph, self.failed = None, True
        return self.graph is not None

    def _capture(self, host, network, criterion):
        self.params = [p for p in network.parameters() if p.requires_grad]
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            if self.train:
                for p in self.params:
                    p.grad = None
                loss = host._batch_loss(network, criterion, self.x, self.y)
                loss.backward()
            else:
                with torch.no_grad():
                    loss = host._batch_loss(network, criterion, self.x, self.y)
            self.loss.copy_(loss.detach().double())
        self.grads = [p.grad for p in self.params]
        self.graph = graph

    def replay(self, level_X, T, idx, total):
        torch.index_select(level_X, 0, idx, out=self.x)
        torch.index_select(T, 0, idx, out=self.y)
        self.graph.replay()
        if self.train:
            for p, g in zip(self.params, self.grads):
                p.grad = g  # an eager batch in between re-pointed .grad; the graph writes to these
        total += self.loss


class ALFusion:
    neuron_importance_methods = ['uniform', 'conductance', 'deeplift', 'featureablation']
    use_hip_graphs = True  # replay the per-batch forward/backward of the level fit as a HIP graph

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

    def _flatten_and_join(self, acts, normalize):
        """``torch.cat([_flatten_acts(a) for a in acts], dim=1)`` and, with normalize, the same of the per-sample
        normalised blocks (alf.py:181-189) -- each model's activations are transposed straight into their columns of
        the joined matrix (one pass instead of a flattening copy plus a concatenating one), and the normalised
        block is written into place by the division itself.  Same values as the separate steps."""
        rows = acts[0].numel() // acts[0].shape[1]
        N = sum(a.shape[1] for a in acts)
        joined = torch.empty((rows, N), dtype=acts[0].dtype, device=acts[0].device)
        normed = torch.empty_like(joined) if normalize else None
        off = 0
        for a in acts:
            h = a.shape[1]
            if a.dim() == 2:
                joined[:, off:off + h].copy_(a)
            elif a.dim() == 3:  # [B, hidden, seq] -> rows (b, s)
                joined.view(a.shape[0], a.shape[2], N)[:, :, off:off + h].copy_(a.permute(0, 2, 1))
            elif a.dim() == 4:  # [B, C, H, W] -> rows (b, w, h)
                joined.view(a.shape[0], a.shape[3], a.shape[2], N)[..., off:off + h].copy_(a.permute(0, 3, 2, 1))
            else:
                raise ValueError(f'Invalid shape for activations: {a.shape}. Expected a 2D, 3D or 4D tensor.')
            if normalize:
                block = joined[:, off:off + h]
                torch.div(block, torch.norm(block, dim=1, keepdim=True), out=normed[:, off:off + h])
            off += h
        return joined, (normed if normalize else joined)

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
        w = w.to(acts.dtype)  # float64 head weights promote the logits (run_alf.py:114-115): keep their precision
        sums = torch.zeros(output_size, acts.shape[1], device=acts.device, dtype=acts.dtype).index_add_(
            0, assignment, acts * w[:, None])
        tot = torch.zeros(output_size, device=acts.device, dtype=acts.dtype).index_add_(0, assignment, w)
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

    def _level_outputs(self, X, level_idx):
        """Every model's output of level ``level_idx`` on X.  The reference re-runs each model from the input
        for every level (alf.py:190); the base models do not change during a fusion, so the outputs of the
        level visited last are kept and the next level is one step further (BaseModel.forward_level)."""
        key = (X.data_ptr(), tuple(X.shape), tuple(id(m) for m in self.models))
        cache = getattr(self, '_level_cache', None)
        acts = None
        with torch.no_grad():
            if cache is not None and cache[0] == key:
                try:
                    if cache[1] == level_idx:
                        acts = cache[2]
                    elif cache[1] == level_idx - 1:
                        acts = [m.forward_level(a, level_idx) for m, a in zip(self.models, cache[2])]
                except NotImplementedError:
                    acts = None
            if acts is None:
                acts = [m.forward_until_level(X, level_idx) for m in self.models]
        self._level_cache = (key, level_idx, acts)
        return acts

    def _k_means_v2(self, level_idx, target_width, fusion_config, level_X):
        X = fusion_config.X
        dev = self.device
        seqlen, height, width = 0, 0, 0
        acts = self._level_outputs(X, level_idx)
        match_T_to_level_X = fusion_config.align_skip_connections and (acts[0].shape == level_X.shape)
        if is_conv_activation(acts[0]):
            height, width = acts[0].shape[2], acts[0].shape[3]
        elif is_transformer_activation(acts[0]):
            seqlen = acts[0].shape[2]
        acts, kmeans_acts = self._flatten_and_join(acts, fusion_config.normalize_acts_for_kmeans)

        ni_scores_of_level = [self.ni_scores[k][level_idx] for k in range(len(self.models))]
        sample_weight = torch.cat(ni_scores_of_level).to(dev)
        sw64 = sample_weight.double()

        # one symmetric Gram of the clustering vectors; everything below streams it
        Gs, gdiag = ops.kmeans_gram(kmeans_acts, rows_sharded=True)
        km = ops.KMeans(Gs, gdiag, sample_weight, target_width, gdiag=gdiag)

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
                # ONE read-back per iteration: (distinct labels, cost, labels changed) -- the reference's loop asks
                # the device three separate questions here (alf.py:245-247,264)
                distinct = (torch.bincount(assignment.long(), minlength=target_width) > 0).sum()
                n_distinct, current_cost, chg = torch.stack(
                    [distinct.double(), cost, (assignment != prev).any().double()]).tolist()
                changed = chg != 0.0
                if int(n_distinct) < target_width:  # a cluster went empty: the rare path, on the host (see _repair_on_host)
                    a_host, d_host = self._repair_on_host(assignment, nearest, target_width)
                    assignment = a_host.to(device=dev, dtype=torch.int32)
                    current_cost, chg = torch.stack([torch.sum(d_host.to(dev) * sw64),
                                                     (assignment != prev).any().double()]).tolist()
                    changed = chg != 0.0
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
        """mean(cluster_weights[c] * (pred - true)^2) (alf.py:414-422), one fused pass on the device."""
        return ops.weighted_mse(act_pred, act_true, cluster_weights)

    def _get_criterion(self, level_idx, target_width, assignment, fusion_config, is_last_level):
        """alf.py:424-454.  The squared-error criteria accept the level's own [batch, channel, ...] layout as
        well as the flattened (rows, channel) one (the mean runs over every element either way) and are
        tagged ``native_layout`` so the fit loop skips the flatten copies."""
        if not fusion_config.use_weighted_loss or (is_last_level and fusion_config.cls_head_weights is None):
            def mse_loss(y_pred, y_true):
                return ops.weighted_mse(y_pred, y_true, None)
            mse_loss.native_layout = True
            return mse_loss
        if is_last_level and fusion_config.use_kd_on_head:
            return nn.KLDivLoss(reduction='batchmean').to(self.device)
        ni_scores = torch.cat([self.ni_scores[i][level_idx] for i in range(len(self.models))], dim=0)
        ni_scores = ni_scores.detach().to(self.device)
        cluster_weights = torch.zeros(target_width, device=self.device).index_add_(
            0, torch.as_tensor(assignment, device=self.device).long(), ni_scores.float())

        def weighted_loss(y_pred, y_true):
            return self._criterion_vectorized(y_pred, y_true, cluster_weights)
        weighted_loss.native_layout = True
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
        """Input of level ``level_idx`` of the model being fused (alf.py:476-491).  Levels are fitted in order and
        the levels below are final, so the previous level's input is carried one level further when it is at hand."""
        if level_idx == 0:
            return fusion_config.X
        fused_hat.eval()
        key = (fusion_config.X.data_ptr(), tuple(fusion_config.X.shape), id(fused_hat))
        cache = getattr(self, '_fused_cache', None)
        out = None
        with torch.no_grad():
            if cache is not None and cache[0] == key and cache[1] == level_idx - 1:
                try:
                    out = fused_hat.forward_level(cache[2], level_idx - 1)
                except NotImplementedError:
                    out = None
            if out is None:
                out = fused_hat.forward_until_level(fusion_config.X, level_idx - 1)
        self._fused_cache = (key, level_idx, out)
        return out

    # -- per-level fit (alf.py:493-594) ------------------------------------------------------------------
    @staticmethod
    def _split_indices(n):
        """Index sets of ``random_split(dataset, [0.8, 0.2])`` (alf.py:518): same lengths, same single
        ``randperm`` draw from the default CPU generator, without materialising per-sample datasets."""
        lengths = [int(math.floor(n * f)) for f in (0.8, 0.2)]
        for i in range(n - sum(lengths)):
            lengths[i % 2] += 1
        perm = torch.randperm(n)
        return perm[:lengths[0]], perm[lengths[0]:]

    @staticmethod
    def _loader_order(n, shuffle):
        """Sample order of one pass of ``DataLoader(..., shuffle=shuffle)`` (alf.py:519-520), consuming the
        default generator exactly as the loader does: one draw per iterator, and for a shuffled pass one
        more that seeds the private generator of the permutation."""
        torch.empty((), dtype=torch.int64).random_()
        if not shuffle:
            return torch.arange(n)
        g = torch.Generator()
        g.manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()))
        return torch.randperm(n, generator=g)

    def _batch_loss(self, network, criterion, x, y):
        y_hat = network(x)
        if not getattr(criterion, 'native_layout', False):
            y_hat, y = self._flatten_acts(y_hat), self._flatten_acts(y)
        return criterion(y_hat, y)

    def _run_batches(self, network, criterion, level_X, T, order, batch_size, optimizer=None, graphed=None):
        """One pass over ``order`` in batches sliced on the device; returns (sum of batch losses, batches).
        The losses are accumulated on the device in f64 (the order and values the reference's running
        Python float sum sees) and read back once.  ``graphed`` (a _GraphedBatch) replays full-size batches
        as one HIP graph launch each; ragged tails and the warm-up batches run eagerly."""
        total = torch.zeros((), dtype=torch.float64, device=self.device)
        num_batches = 0
        for lo in range(0, order.shape[0], batch_size):
            idx = order[lo:lo + batch_size]
            if graphed is not None and idx.shape[0] == batch_size and graphed.ready(self, network, criterion):
                graphed.replay(level_X, T, idx, total)
                if optimizer is not None:
                    optimizer.step()
            else:
                x, y = level_X.index_select(0, idx), T.index_select(0, idx)
                if optimizer is not None:
                    optimizer.zero_grad()
                loss = self._batch_loss(network, criterion, x, y)
                if optimizer is not None:
                    loss.backward()
                    optimizer.step()
                total += loss.detach().double()
                del loss  # a live autograd graph would pin its AccumulateGrad nodes to this stream and break a later capture
            num_batches += 1
        return total.item(), num_batches

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

        # inputs and targets stay resident on the device in the network's own layout; batches are index
        # gathers (the reference builds per-sample Dataset / DataLoader objects over the same tensors)
        level_X = level_X.to(self.device).float().contiguous()
        T = T.to(self.device).float().contiguous()
        sharding = SampleSharding._active
        if sharding is not None:
            # The regression is not a row sum: it visits the samples in the loaders' order.  Its inputs and targets
            # (rows = this rank's samples) are gathered in rank order, so every rank holds the rows of the
            # un-sharded run and fits the level on all of them -- the replicated tail of the sharded path.  The
            # fit draws from torch's CPU generator (split, shuffles, re-initialisation): ranks seeded alike walk
            # the same trajectory, and the level is broadcast from rank 0 afterwards so that none can drift.
            level_X = allgather_rows(level_X, sharding.group)
            T = allgather_rows(T, sharding.group)
        train_idx, val_idx = self._split_indices(level_X.shape[0])
        val_order = val_idx.to(self.device)
        batch_size = fusion_config.train_batch_size

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

        # launch-bound inner loop -> one HIP graph per phase once a level has enough full batches to pay for it
        use_graphs = self.use_hip_graphs and level_X.is_cuda
        train_graph = _GraphedBatch(level_X, T, batch_size, True) \
            if use_graphs and train_idx.shape[0] // batch_size >= 8 else None
        val_graph = _GraphedBatch(level_X, T, batch_size, False) \
            if use_graphs and val_idx.shape[0] // batch_size >= 8 else None

        patience = fusion_config.max_patience_epochs
        best_val_loss = float('inf')
        num_epochs = fusion_config.epochs_per_level[level_idx]
        for epoch in range(num_epochs):
            network.train(True)
            train_order = train_idx[self._loader_order(train_idx.shape[0], True)].to(self.device)
            train_loss, train_batches = self._run_batches(network, criterion, level_X, T, train_order, batch_size,
                                                          optimizer, train_graph)
            if fusion_config.use_scheduler:
                scheduler.step(train_loss)  # the reference passes the loss as the epoch argument (alf.py:559)

            network.eval()
            self._loader_order(val_idx.shape[0], False)
            with torch.no_grad():
                val_loss, val_batches = self._run_batches(network, criterion, level_X, T, val_order, batch_size,
                                                          None, val_graph)
            avg_train_loss = train_loss / train_batches
            avg_val_loss = val_loss / val_batches
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

        if train_graph is not None and train_graph.graph is not None:
            optimizer.zero_grad(set_to_none=True)  # the gradients lived in the graph's private pool
        if sharding is not None:
            broadcast_parameters(network, 0, sharding.group)
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
        self._level_cache = self._fused_cache = None
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
