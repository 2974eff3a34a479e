"""Mirror of gnngls/models.py (reference models.py:5-70) on the MI355X HIP path.

`EdgePropertyPredictionModel` keeps the reference's constructor signature, module tree and
state-dict key layout (so `model.load_state_dict(checkpoint['model_state_dict'])`, `.to(device)`,
`.eval()` from scripts/test.py:43-54 work unchanged), but `forward` does not run torch ops: it
packs the parameters once into the fp32 weight image declared in include/gnngls_hip.h and calls
`gnngls_regret_forward` (hand-written HIP: f32-MFMA node MLPs + LDS-tiled line-graph attention).

In training mode (`model.train()`, scripts/train.py:20-32) `forward` is one autograd node: the HIP training forward
(`gnngls_regret_train_forward`, BatchNorm on batch statistics, running statistics updated like torch does) whose backward
is `gnngls_regret_train_backward`; `loss.backward()` / `optimizer.step()` of train.py work unchanged on the module's
parameters.

The graph argument `G` is a `LineGraph` (this package's stand-in for the DGL graph built at
datasets.py:56-60); because the line graph of the complete graph K_n has closed-form structure it
only carries n and the node->TSP-edge map.
"""
import ctypes
import itertools

import torch
import torch.nn as nn

from . import _lib

EMBED_DIM = 128
N_HEADS = 8
HIDDEN_DIM = 512


class LineGraph:
    """Line graph of K_n for `batch` stacked instances (cf. datasets.py:56-60, train.py:118 dgl.batch).
    Node r of an instance is the TSP edge (i<j) of rank r in itertools.combinations(range(n), 2)."""

    def __init__(self, n, batch=1, device=None):
        self.n = int(n)
        self.batch = int(batch)
        e = torch.tensor(list(itertools.combinations(range(self.n), 2)), dtype=torch.int64).reshape(-1, 2)
        self.ndata = {"e": e.repeat(self.batch, 1) if self.batch > 1 else e}
        self.device = torch.device("cpu")
        if device is not None:
            self.to(device)

    @property
    def nodes_per_instance(self):
        return self.n * (self.n - 1) // 2

    def number_of_nodes(self):
        return self.nodes_per_instance * self.batch

    def to(self, device):
        g = LineGraph.__new__(LineGraph)
        g.n, g.batch, g.device = self.n, self.batch, torch.device(device)
        g.ndata = {k: v.to(device) for k, v in self.ndata.items()}
        return g


def batch(graphs):
    """dgl.batch (train.py:118-121, the DataLoader's collate_fn) for LineGraphs of one size: the disjoint union is again
    a LineGraph (batch = total instance count); every ndata entry is concatenated along the node axis."""
    graphs = list(graphs)
    if not graphs or any(g.n != graphs[0].n for g in graphs):
        raise ValueError("batch() needs a non-empty list of line graphs of one instance size (homogeneous dataset)")
    out = LineGraph.__new__(LineGraph)
    out.n, out.batch, out.device = graphs[0].n, sum(g.batch for g in graphs), graphs[0].device
    out.ndata = {k: torch.cat([g.ndata[k] for g in graphs]) for k in graphs[0].ndata}
    return out


class SkipConnection(nn.Module):
    """models.py:5-15 (parameter container; the sum x + y is fused into the HIP epilogues)."""

    def __init__(self, module):
        super().__init__()
        self.module = module


class GATConvParams(nn.Module):
    """Parameters of dgl.nn.GATConv(in_feats, out_feats, num_heads) (models.py:23): fc.weight
    [H*F, in], attn_l / attn_r [1,H,F]; DGL 0.6.1 initialisation (xavier_normal_, gain sqrt(2)).
    A `bias` [H*F] parameter (DGL >= 0.7 checkpoints: `rst + bias` after the aggregation) appears when a checkpoint carries
    one: folded into BatchNorm-1's shift at pack time (inference); in training mode BatchNorm-1 normalises with the batch
    mean, which absorbs a per-column constant -- the prediction does not depend on it, its gradient is exactly zero, and
    only the running mean moves by it (see _TrainStep)."""

    def __init__(self, in_feats, out_feats, num_heads):
        super().__init__()
        self.fc = nn.Linear(in_feats, out_feats * num_heads, bias=False)
        self.attn_l = nn.Parameter(torch.empty(1, num_heads, out_feats))
        self.attn_r = nn.Parameter(torch.empty(1, num_heads, out_feats))
        self.register_buffer("bias", None)
        gain = nn.init.calculate_gain("relu")
        nn.init.xavier_normal_(self.fc.weight, gain=gain)
        nn.init.xavier_normal_(self.attn_l, gain=gain)
        nn.init.xavier_normal_(self.attn_r, gain=gain)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        key = prefix + "bias"
        if key in state_dict and self.bias is None:
            self._buffers.pop("bias", None)
            self.register_parameter("bias", nn.Parameter(torch.zeros_like(state_dict[key], dtype=torch.float32)))   # then loaded
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)


class AttentionLayer(nn.Module):
    """models.py:18-41"""

    def __init__(self, embed_dim, n_heads, hidden_dim):
        super().__init__()
        self.message_passing = SkipConnection(GATConvParams(embed_dim, embed_dim // n_heads, n_heads))
        self.feed_forward = nn.Sequential(
            nn.BatchNorm1d(embed_dim),
            SkipConnection(nn.Sequential(nn.Linear(embed_dim, hidden_dim), nn.ReLU(), nn.Linear(hidden_dim, embed_dim))),
            nn.BatchNorm1d(embed_dim),
        )


class EdgePropertyPredictionModel(nn.Module):
    """models.py:44-70.  As in the reference the number of AttentionLayers is `n_heads`
    (models.py:59-61 iterates range(n_heads)); `n_layers` is accepted and ignored."""

    def __init__(self, in_dim, embed_dim, out_dim, n_layers, n_heads=1):
        super().__init__()
        self.embed_dim = embed_dim
        self.in_dim = in_dim
        self.out_dim = out_dim
        self.n_heads = n_heads
        self.embed_layer = nn.Linear(in_dim, embed_dim)
        self.message_passing_layers = nn.Sequential(
            *(AttentionLayer(embed_dim, n_heads, HIDDEN_DIM) for _ in range(n_heads)))
        self.decision_layer = nn.Linear(embed_dim, out_dim)
        self._packed = None
        self._prepared = None       # device image of the split feed-forward weights (gnngls_regret_prepare), made with _packed
        self._workspace = None

    # -- weight image -----------------------------------------------------------------------------
    def _check_supported(self):
        if self.embed_dim != EMBED_DIM or self.n_heads != N_HEADS or self.out_dim != 1:
            raise NotImplementedError(
                "the HIP forward is specialised to the reference architecture (embed_dim=128, n_heads=8, "
                f"out_dim=1; got embed_dim={self.embed_dim}, n_heads={self.n_heads}, out_dim={self.out_dim})")

    @staticmethod
    def _bn_affine(bn):
        scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
        shift = bn.bias.detach().float() - bn.running_mean.detach().float() * scale
        return scale, shift

    def pack_weights(self, device):
        """fp32 weight image in the order documented in include/gnngls_hip.h."""
        self._check_supported()
        blocks = [self.embed_layer.weight.detach().float().reshape(-1), self.embed_layer.bias.detach().float()]
        for layer in self.message_passing_layers:
            gat = layer.message_passing.module
            bn1, ff, bn2 = layer.feed_forward[0], layer.feed_forward[1].module, layer.feed_forward[2]
            s1, t1 = self._bn_affine(bn1)
            if gat.bias is not None:          # DGL >= 0.7: rst + bias, folded into the BN1 shift
                t1 = t1 + gat.bias.to(s1.device).reshape(-1) * s1
            s2, t2 = self._bn_affine(bn2)
            blocks += [gat.fc.weight.detach().float().reshape(-1), gat.attn_l.detach().float().reshape(-1),
                       gat.attn_r.detach().float().reshape(-1), s1, t1,
                       ff[0].weight.detach().float().reshape(-1), ff[0].bias.detach().float(),
                       ff[2].weight.detach().float().reshape(-1), ff[2].bias.detach().float(), s2, t2]
        blocks += [self.decision_layer.weight.detach().float().reshape(-1), self.decision_layer.bias.detach().float(),
                   torch.zeros(3)]
        packed = torch.cat([b.to("cpu") for b in blocks]).contiguous()
        expect = _lib.load().gnngls_model_packed_floats(self.in_dim, len(self.message_passing_layers))
        assert packed.numel() == expect, (packed.numel(), expect)
        return packed.to(device)

    def invalidate(self):
        self._packed = None

    def _state_version(self):
        """Changes whenever a parameter or buffer is written in place (optimizer.step(), running statistics)."""
        return tuple(t._version for t in itertools.chain(self.parameters(), self.buffers()))

    def train_parameters(self):
        """Parameters in the order of the raw training image (include/gnngls_hip.h, N4)."""
        out = [self.embed_layer.weight, self.embed_layer.bias]
        for layer in self.message_passing_layers:
            gat = layer.message_passing.module
            bn1, ff, bn2 = layer.feed_forward[0], layer.feed_forward[1].module, layer.feed_forward[2]
            out += [gat.fc.weight, gat.attn_l, gat.attn_r, bn1.weight, bn1.bias, ff[0].weight, ff[0].bias,
                    ff[2].weight, ff[2].bias, bn2.weight, bn2.bias]
        return out + [self.decision_layer.weight, self.decision_layer.bias]

    def gat_biases(self):
        """[(layer index, bias parameter)] of the layers whose GATConv carries a bias (DGL >= 0.7 checkpoints)."""
        return [(k, layer.message_passing.module.bias) for k, layer in enumerate(self.message_passing_layers)
                if layer.message_passing.module.bias is not None]

    def batch_norms(self):
        return [bn for layer in self.message_passing_layers for bn in (layer.feed_forward[0], layer.feed_forward[2])]

    def load_state_dict(self, *args, **kwargs):
        self._packed = None
        return super().load_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        self._packed = None
        return super()._apply(fn, *args, **kwargs)

    # -- forward ------------------------------------------------------------------------------------
    def forward(self, G, x):
        """x [B*N, in_dim] fp32 on the GPU, G a LineGraph -> [B*N, 1] fp32 (models.py:65-70)."""
        if not x.is_cuda:
            raise _lib.GnnglsHipError("EdgePropertyPredictionModel.forward needs CUDA/HIP tensors (no CPU fallback)")
        n = G.n
        N = n * (n - 1) // 2
        x = x.contiguous().float()
        total, in_dim = x.shape
        assert in_dim == self.in_dim and total % N == 0 and total == G.number_of_nodes()
        if self.training:                                   # train.py:20-32
            self._check_supported()
            biases = [b for _, b in self.gat_biases()]
            return _TrainStep.apply(self, x, total // N, n, len(biases), *self.train_parameters(), *biases)
        return regret_forward(self, x, total // N, n).reshape(total, 1)


class _TrainStep(torch.autograd.Function):
    """y_pred = model(batch, x) in training mode as ONE autograd node over the HIP training kernels."""

    @staticmethod
    def forward(ctx, model, x, B, n, n_bias, *params):
        L = _lib.load()
        dev = x.device
        params, biases = (params[:-n_bias], params[-n_bias:]) if n_bias else (params, ())
        n_layers = len(model.message_passing_layers)
        image = torch.cat([p.detach().reshape(-1).float() for p in params] + [torch.zeros(3, device=dev)]).contiguous()
        assert image.numel() == L.gnngls_model_packed_floats(model.in_dim, n_layers)
        N = n * (n - 1) // 2
        # the workspace holds the activations the backward needs: one per forward call, owned by the autograd node
        ws = torch.empty(int(L.gnngls_regret_train_workspace_bytes(B, n, n_layers)), dtype=torch.uint8, device=dev)
        y = torch.empty((B * N, 1), dtype=torch.float32, device=dev)
        stats = torch.empty((n_layers, 2, 2, EMBED_DIM), dtype=torch.float32, device=dev)
        bns = model.batch_norms()
        eps = bns[0].eps if bns else 1e-5
        _lib.check(L.gnngls_regret_train_forward(_lib.ptr(x), _lib.ptr(image), B, n, model.in_dim, n_layers, eps,
                                                 _lib.ptr(y), _lib.ptr(stats), _lib.ptr(ws), ctypes.c_int64(ws.numel()),
                                                 _lib.current_stream()), "regret_train_forward")
        # GATConv bias (DGL >= 0.7): h1 = h + GATConv(h) + bias feeds BatchNorm-1 only; in training mode its batch mean
        # takes the constant up (variance, prediction and every other gradient unchanged), so the kernels run without it
        for (k, _), bias in zip(model.gat_biases(), biases):
            stats[k, 0, 0] += bias.detach().reshape(-1)
        _update_running_stats(bns, stats.reshape(-1, 2, EMBED_DIM))
        ctx.model_dims = (B, n, model.in_dim, n_layers)
        ctx.shapes = [p.shape for p in params]
        ctx.bias_shapes = [b.shape for b in biases]
        ctx.save_for_backward(x, image)
        ctx.ws = ws
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.load()
        if ctx.ws is None:
            raise RuntimeError("the activations of this training forward were consumed by its first backward "
                               "(the hidden activations are overwritten in place): retain_graph is not supported")
        x, image = ctx.saved_tensors
        B, n, in_dim, n_layers = ctx.model_dims
        grads = torch.empty_like(image)
        dy = dy.contiguous().float()
        _lib.check(L.gnngls_regret_train_backward(_lib.ptr(x), _lib.ptr(image), _lib.ptr(dy), B, n, in_dim, n_layers,
                                                  _lib.ptr(grads), _lib.ptr(ctx.ws), ctypes.c_int64(ctx.ws.numel()),
                                                  _lib.current_stream()), "regret_train_backward")
        ctx.ws = None
        out, off = [], 0
        for shape in ctx.shapes:
            k = shape.numel()
            out.append(grads[off:off + k].view(shape))
            off += k
        # d loss / d bias = column sums of BatchNorm-1's input gradient = 0 exactly (batch statistics)
        zeros = [torch.zeros(shape, dtype=torch.float32, device=grads.device) for shape in ctx.bias_shapes]
        return (None, None, None, None, None, *out, *zeros)


@torch.no_grad()
def _update_running_stats(bns, stats):
    """nn.BatchNorm1d's training-mode bookkeeping (models.py:27,35): running <- (1-f)*running + f*batch with the
    UNBIASED batch variance, f = momentum (or 1/num_batches_tracked when momentum is None)."""
    tracked = [(bn, stats[k]) for k, bn in enumerate(bns) if bn.track_running_stats and bn.running_mean is not None]
    if not tracked:
        return
    torch._foreach_add_([bn.num_batches_tracked for bn, _ in tracked], 1)
    momenta = {bn.momentum for bn, _ in tracked}
    if len(momenta) == 1 and None not in momenta:
        f = momenta.pop()
        torch._foreach_lerp_([bn.running_mean for bn, _ in tracked], [s[0] for _, s in tracked], f)
        torch._foreach_lerp_([bn.running_var for bn, _ in tracked], [s[1] for _, s in tracked], f)
        return
    for bn, s in tracked:
        f = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
        bn.running_mean.lerp_(s[0], f)
        bn.running_var.lerp_(s[1], f)


def regret_forward(model, feat, B, n, max_workspace_bytes=48 << 30):
    """feat [B*N, in_dim] (or [B,N]) fp32 cuda -> y [B,N] fp32 cuda via gnngls_regret_forward."""
    L = _lib.load()
    dev = feat.device
    version = model._state_version()
    n_layers = len(model.message_passing_layers)
    if model._packed is None or model._packed.device != dev or getattr(model, "_packed_version", None) != version:
        model._packed = model.pack_weights(dev)
        model._packed_version = version
        # the weight-only part of the forward (bf16 pieces of the feed-forward / fc weights in MFMA fragment order), once per
        # weight image instead of once per call (test.py:43-54 loads the checkpoint once, test.py:72-77 calls per instance)
        if not hasattr(L, "gnngls_regret_forward_prepared"):                  # (an ABI-3 build named by GNNGLS_HIP_SO: A/B runs)
            model._prepared = None
        else:
            model._prepared = torch.empty(int(L.gnngls_regret_prepared_bytes(n_layers)), dtype=torch.uint8, device=dev)
            _lib.check(L.gnngls_regret_prepare(_lib.ptr(model._packed), model.in_dim, n_layers, _lib.ptr(model._prepared),
                                               ctypes.c_int64(model._prepared.numel()), _lib.current_stream()), "regret_prepare")
    N = n * (n - 1) // 2
    need = int(L.gnngls_regret_forward_workspace_bytes(B, n))
    ws_bytes = min(need, max(int(L.gnngls_regret_forward_workspace_bytes(1, n)), max_workspace_bytes))
    ws = model._workspace
    if ws is None or ws.device != dev or ws.numel() < ws_bytes:
        model._workspace = None
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        model._workspace = ws
    y = torch.empty((B, N), dtype=torch.float32, device=dev)
    if model._prepared is None:
        _lib.check(L.gnngls_regret_forward(_lib.ptr(feat), _lib.ptr(model._packed), B, n, model.in_dim, n_layers, _lib.ptr(y),
                                           _lib.ptr(ws), ctypes.c_int64(ws.numel()), _lib.current_stream()), "regret_forward")
        return y
    _lib.check(L.gnngls_regret_forward_prepared(_lib.ptr(feat), _lib.ptr(model._packed), _lib.ptr(model._prepared),
                                                ctypes.c_int64(model._prepared.numel()), B, n, model.in_dim, n_layers,
                                                _lib.ptr(y), _lib.ptr(ws), ctypes.c_int64(ws.numel()), _lib.current_stream()),
               "regret_forward")
    return y


def pack_features(D, scale, min_):
    """D [B,n,n] fp64 cuda -> scaled features [B,N] fp32 (datasets.py:73-95 for features=[weight])."""
    B, n, _ = D.shape
    feat = torch.empty((B, n * (n - 1) // 2), dtype=torch.float32, device=D.device)
    L = _lib.load()
    _lib.check(L.gnngls_pack_features(_lib.ptr(D), B, n, float(scale), float(min_), _lib.ptr(feat),
                                      _lib.current_stream()), "pack_features")
    return feat


def unpack_regret(y, n, scale, min_):
    """y [B,N] fp32 cuda -> 'regret_pred' guide matrix [B,n,n] fp64 (test.py:79-83)."""
    B = y.shape[0]
    out = torch.empty((B, n, n), dtype=torch.float64, device=y.device)
    L = _lib.load()
    _lib.check(L.gnngls_unpack_regret(_lib.ptr(y.contiguous()), B, n, float(scale), float(min_), _lib.ptr(out),
                                      _lib.current_stream()), "unpack_regret")
    return out
