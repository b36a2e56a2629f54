"""Small ViT organised in levels (reference: models/vit.py:10-223).

Levels only: `get_ordered_trainable_named_layers` is empty, so of the four algorithms only ALF
(level-wise) can fuse it -- exactly as in the reference.  Level outputs are kept as
[batch, hidden, seq] (the neuron axis first after the batch, vit.py:57,88-95); the last encoder
returns the CLS row [batch, hidden].  Module and parameter names match the reference so its
state_dicts load unchanged.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .base_model import BaseModel, LevelRepresentation


class MultiHeadSelfAttention(nn.Module):
    def __init__(self, hidden_dim, head, dropout):
        super().__init__()
        self.head, self.hidden_dim = head, hidden_dim
        self.sqrt_d = hidden_dim ** 0.5  # the reference scales by sqrt(hidden), not sqrt(head dim) (vit.py:15)
        self.q = nn.Linear(hidden_dim, hidden_dim)
        self.k = nn.Linear(hidden_dim, hidden_dim)
        self.v = nn.Linear(hidden_dim, hidden_dim)
        self.o = nn.Linear(hidden_dim, hidden_dim)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x):
        b, n, _ = x.size()
        split = lambda t: t.view(b, n, self.head, self.hidden_dim // self.head).transpose(1, 2)
        q, k, v = split(self.q(x)), split(self.k(x)), split(self.v(x))
        score = F.softmax(torch.einsum('bhif, bhjf->bhij', q, k) / self.sqrt_d, dim=-1)
        attn = torch.einsum('bhij, bhjf->bihf', score, v)
        return self.dropout(self.o(attn.flatten(2)))


class InputEmbedder(nn.Module):
    """Image -> patch words -> linear embedding + CLS token + positions, output [batch, hidden, seq]."""

    def __init__(self, img_size, patch, in_c, hidden_dim):
        super().__init__()
        self.img_size, self.patch, self.in_c, self.hidden_dim = img_size, patch, in_c, hidden_dim
        self.patch_size = img_size // patch
        self.emb = nn.Linear(self.patch_size ** 2 * in_c, hidden_dim)
        self.cls_token = nn.Parameter(torch.randn(1, 1, hidden_dim))
        self.pos_emb = nn.Parameter(torch.randn(1, patch ** 2 + 1, hidden_dim))

    def _to_words(self, x):
        ps = self.patch_size
        out = x.unfold(2, ps, ps).unfold(3, ps, ps).permute(0, 2, 3, 4, 5, 1)
        return out.reshape(x.size(0), self.patch ** 2, -1)

    def forward(self, x):
        out = self.emb(self._to_words(x))
        out = torch.cat([self.cls_token.repeat(out.size(0), 1, 1), out], dim=1) + self.pos_emb
        return out.permute(0, 2, 1)


class TransformerEncoder(nn.Module):
    def __init__(self, hidden_dim, mlp_hidden, head, is_last_encoder, dropout=0.0):
        super().__init__()
        self.la1 = nn.LayerNorm(hidden_dim)
        self.msa = MultiHeadSelfAttention(hidden_dim, head=head, dropout=dropout)
        self.la2 = nn.LayerNorm(hidden_dim)
        self.mlp = nn.Sequential(nn.Linear(hidden_dim, mlp_hidden), nn.GELU(), nn.Dropout(dropout),
                                 nn.Linear(mlp_hidden, hidden_dim), nn.GELU(), nn.Dropout(dropout))
        self.is_last_encoder = is_last_encoder

    def forward(self, x):
        x = x.permute(0, 2, 1)  # [batch, seq, hidden]
        out = self.msa(self.la1(x)) + x
        out = self.mlp(self.la2(out)) + out
        return out[:, 0] if self.is_last_encoder else out.permute(0, 2, 1)


class ViT(BaseModel):
    captures_may_alias = True  # forward() never writes into a tensor a hooked layer has seen

    def __init__(self, in_c, num_classes=10, img_size=32, patch=8, dropout=0.0, depth=7, hidden=384,
                 mlp_hidden=384 * 4, head=8, embedder_separare_level=False, train_indices=None, val_indices=None):
        super().__init__(train_indices, val_indices)
        self.in_c, self.num_classes, self.img_size, self.patch = in_c, num_classes, img_size, patch
        self.dropout_p, self.depth, self.hidden, self.mlp_hidden, self.head = dropout, depth, hidden, mlp_hidden, head
        self.embedder_separare_level = embedder_separare_level  # (sic) the reference's keyword
        embedder = InputEmbedder(img_size, patch, in_c, hidden)
        encoders = [TransformerEncoder(hidden, mlp_hidden=mlp_hidden, dropout=dropout, head=head,
                                       is_last_encoder=(i == depth - 1)) for i in range(depth)]
        if embedder_separare_level:
            blocks = [embedder] + encoders
        else:
            blocks = [nn.Sequential(embedder, encoders[0])] + encoders[1:]
        self.blocks = nn.ModuleList(blocks)
        self.fc = nn.Sequential(nn.LayerNorm(hidden), nn.Linear(hidden, num_classes))

    def forward(self, x):
        for blk in self.blocks:
            x = blk(x)
        return self.fc(x)

    def forward_until_level(self, x, level_idx=-1):
        if level_idx == -1:
            return self.forward(x)
        for i, blk in enumerate(self.blocks):
            x = blk(x)
            if i == level_idx:
                return x
        return self.fc(x)

    def forward_level(self, acts, level_idx):
        return self.blocks[level_idx](acts) if level_idx < len(self.blocks) else self.fc(acts)

    def copy_model(self):
        twin = ViT(in_c=self.in_c, num_classes=self.num_classes, img_size=self.img_size, patch=self.patch,
                   dropout=self.dropout_p, depth=self.depth, hidden=self.hidden, mlp_hidden=self.mlp_hidden,
                   head=self.head, embedder_separare_level=self.embedder_separare_level)
        return self._adopt_weights(twin)

    @property
    def get_ordered_trainable_named_layers(self):
        return []

    @property
    def get_ordered_trainable_named_levels(self):
        levels, start = [], 0
        if self.embedder_separare_level:
            levels.append(LevelRepresentation('embedder', self.hidden, self.blocks[0]))
            start = 1
        for i, blk in enumerate(self.blocks[start:]):
            levels.append(LevelRepresentation(f'block_{i}', self.hidden, blk))
        levels.append(LevelRepresentation('classifier', self.num_classes, self.fc))
        return levels
