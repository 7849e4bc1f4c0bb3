"""Restatement of the third-party `vit-pytorch` ViT (lucidrains, PyPI, UNPINNED in the reference's
requirements.txt:8 and absent from this container), >=1.2 module layout, written from its published
structure so that the reference's `from vit_pytorch import ViT` resolves here.  This is the layout the
reference depends on structurally (trainer.py:671-673 indexes transformer.layers[i][0].dropout).
Used ONLY by tests/golden/make_goldens.py to run the imported reference; goldens therefore pin this
restatement (ViT parity is unpinned upstream — see SURVEY.md §8(c))."""
import torch
from torch import nn


class FeedForward(nn.Module):
    def __init__(self, dim, hidden_dim, dropout=0.0):
        super().__init__()
        self.net = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, hidden_dim), nn.GELU(), nn.Dropout(dropout),
                                 nn.Linear(hidden_dim, dim), nn.Dropout(dropout))

    def forward(self, x):
        return self.net(x)


class Attention(nn.Module):
    def __init__(self, dim, heads=8, dim_head=64, dropout=0.0):
        super().__init__()
        inner = dim_head * heads
        self.heads, self.scale = heads, dim_head ** -0.5
        self.norm = nn.LayerNorm(dim)
        self.attend = nn.Softmax(dim=-1)
        self.dropout = nn.Dropout(dropout)
        self.to_qkv = nn.Linear(dim, inner * 3, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, dim), nn.Dropout(dropout))

    def forward(self, x):
        x = self.norm(x)
        b, n, _ = x.shape
        q, k, v = (t.reshape(b, n, self.heads, -1).transpose(1, 2) for t in self.to_qkv(x).chunk(3, dim=-1))
        attn = self.dropout(self.attend(torch.matmul(q, k.transpose(-1, -2)) * self.scale))
        out = torch.matmul(attn, v).transpose(1, 2).reshape(b, n, -1)
        return self.to_out(out)


class Transformer(nn.Module):
    def __init__(self, dim, depth, heads, dim_head, mlp_dim, dropout=0.0):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.layers = nn.ModuleList([nn.ModuleList([Attention(dim, heads, dim_head, dropout),
                                                    FeedForward(dim, mlp_dim, dropout)]) for _ in range(depth)])

    def forward(self, x):
        for attn, ff in self.layers:
            x = attn(x) + x
            x = ff(x) + x
        return self.norm(x)


class _Patchify(nn.Module):
    """einops Rearrange('b c (h p1) (w p2) -> b (h w) (p1 p2 c)') without einops' module (keeps index 0)."""
    def __init__(self, p):
        super().__init__()
        self.p = p

    def forward(self, img):
        b, c, H, W = img.shape
        p = self.p
        return img.reshape(b, c, H // p, p, W // p, p).permute(0, 2, 4, 3, 5, 1).reshape(b, (H // p) * (W // p), p * p * c)


class ViT(nn.Module):
    def __init__(self, *, image_size, patch_size, num_classes, dim, depth, heads, mlp_dim, pool="cls",
                 channels=3, dim_head=64, dropout=0.0, emb_dropout=0.0):
        super().__init__()
        n = (image_size // patch_size) ** 2
        pd = channels * patch_size * patch_size
        self.to_patch_embedding = nn.Sequential(_Patchify(patch_size), nn.LayerNorm(pd), nn.Linear(pd, dim), nn.LayerNorm(dim))
        self.pos_embedding = nn.Parameter(torch.randn(1, n + 1, dim))
        self.cls_token = nn.Parameter(torch.randn(1, 1, dim))
        self.dropout = nn.Dropout(emb_dropout)
        self.transformer = Transformer(dim, depth, heads, dim_head, mlp_dim, dropout)
        self.pool = pool
        self.to_latent = nn.Identity()
        self.mlp_head = nn.Linear(dim, num_classes)

    def forward(self, img):
        x = self.to_patch_embedding(img)
        b, n, _ = x.shape
        x = torch.cat((self.cls_token.expand(b, -1, -1), x), dim=1)
        x = x + self.pos_embedding[:, : n + 1]
        x = self.dropout(x)
        x = self.transformer(x)
        x = x.mean(dim=1) if self.pool == "mean" else x[:, 0]
        return self.mlp_head(self.to_latent(x))
