"""CPU oracle for the ERFNet hot path.  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import it.  The shipped package
(erfnet_pytorch_amd) never imports anything from oracle/ and has no CPU fallback.

It restates, as pure functions over a state_dict (no nn.Module classes), the
algorithm of the reference's Python model, loss, metric and optimiser:

  downsampler      /root/reference/train/erfnet.py:11-22
  nb1d             /root/reference/train/erfnet.py:39-74
  upsampler        /root/reference/train/erfnet.py:146-155
  encoder          /root/reference/train/erfnet.py:77-109
  decoder          /root/reference/train/erfnet.py:157-181
  erfnet           /root/reference/train/erfnet.py:271-286
  imagenet head    /root/reference/train/erfnet_imagenet.py:64-129
  multi_models     /root/reference/train/multi_models.py:23-36,75-221 (eval forward)
  weighted CE      /root/reference/train/main.py:72-80,121,202
  class weights    /root/reference/train/main.py:92-111
  iou counts       /root/reference/train/iouEval.py:8-75
  Adam + poly lr   /root/reference/train/main.py:140,156-157,165

Parity pin: the reference holds no tests, golden vectors or pretrained weights
(see tests/golden/README.md), so this oracle is pinned against outputs of the
reference's own modules executed in the build container; the generator is
tests/golden/make_golden.py and its outputs are the committed tests/golden/*.npz.
tests/test_oracle_golden.py re-checks the oracle against them on every run.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
BN_EPS = 1e-3          # every BatchNorm2d in the model: train/erfnet.py:17,47,53,150
BN_MOMENTUM = 0.1      # torch default, unchanged by the reference

# train/main.py:92-111 -- the 20-class Cityscapes recipe; class 19 is "ignore" (weight 0)
CITYSCAPES_WEIGHTS = [
    2.8149201869965, 6.9850029945374, 3.7890393733978, 9.9428062438965, 9.7702074050903,
    9.5110931396484, 10.311357498169, 10.026463508606, 4.6323022842407, 9.5608062744141,
    7.8698215484619, 9.5168733596802, 10.373730659485, 6.6616044044495, 10.260489463806,
    10.287888526917, 10.289801597595, 10.405355453491, 10.138095855713, 0.0,
]

# (kind, in, out, dropout p, dilation): train/erfnet.py:80-95 and :163-169
ENCODER_LAYERS = ([("down", 16, 64, 0.0, 1)] + [("nb", 64, 64, 0.03, 1)] * 5 + [("down", 64, 128, 0.0, 1)]
                  + [("nb", 128, 128, 0.3, d) for _ in range(2) for d in (2, 4, 8, 16)])
DECODER_LAYERS = [("up", 128, 64, 0.0, 1), ("nb", 64, 64, 0.0, 1), ("nb", 64, 64, 0.0, 1),
                  ("up", 64, 16, 0.0, 1), ("nb", 16, 16, 0.0, 1), ("nb", 16, 16, 0.0, 1)]


class _RoundSTE(torch.autograd.Function):
    """Storage rounding of an activation: the value is rounded to `dtype` on the way forward and the gradient that flows back
    through the same tensor is rounded too (an engine that keeps its feature maps in bf16 keeps their gradients in bf16)."""

    @staticmethod
    def forward(ctx, t, dtype, fwd, bwd):
        ctx.dtype, ctx.bwd = dtype, bwd
        return t.to(dtype).to(t.dtype) if fwd else t.clone()

    @staticmethod
    def backward(ctx, g):
        return (g.to(ctx.dtype).to(g.dtype) if ctx.bwd else g), None, None, None


class RunState:
    """Mode switches and side channels of one forward pass.

    `storage_dtype` (default None = the reference's arithmetic, which is what the golden vectors pin): when set to
    torch.bfloat16 the SAME graph is evaluated with every stored feature map, every convolution weight and every feature-map
    gradient rounded to that type at the point an engine with bf16 storage and fp32 accumulation rounds them -- the yardstick
    for the bf16 kernels.  `record_gates`: collect, per ReLU / max-pool decision tensor, the smallest distance of a
    pre-activation from its decision boundary relative to the tensor's rms (`gate_margins`), so that a test can tell whether
    its input has a decision that rounding noise could flip."""

    def __init__(self, training: bool = False, dropout_masks: Optional[Dict[str, torch.Tensor]] = None,
                 update_running: bool = True, encoder_dropout: Optional[Dict[int, float]] = None,
                 storage_dtype: Optional[torch.dtype] = None, record_gates: bool = False):
        self.training = training
        self.storage_dtype = storage_dtype
        self.record_gates = record_gates
        self.gate_margins: List[Tuple[str, float]] = []
        # prefix -> [N,C,1,1] multiplier (already scaled by 1/(1-p)); when absent in
        # training mode the oracle draws a mask with torch's RNG like nn.Dropout2d
        self.dropout_masks = dropout_masks if dropout_masks is not None else {}
        self.update_running = update_running
        self.new_running: Dict[str, torch.Tensor] = {}
        self.used_masks: Dict[str, torch.Tensor] = {}

    # -- storage-rounding hooks (identity unless storage_dtype is set) ----------------------------------------------------
    def r(self, t):
        """a stored feature map (value and gradient rounded)"""
        return t if self.storage_dtype is None else _RoundSTE.apply(t, self.storage_dtype, True, True)

    def w(self, t):
        """a convolution weight as the matrix units see it (rounded copy; the gradient reaches the fp32 master unchanged)"""
        return t if self.storage_dtype is None else _RoundSTE.apply(t, self.storage_dtype, True, False)

    def g(self, t):
        """a tensor kept in fp32 whose GRADIENT is stored rounded (the logits)"""
        return t if self.storage_dtype is None else _RoundSTE.apply(t, self.storage_dtype, False, True)

    def relu(self, name, t):
        if self.record_gates:
            rms = float(t.detach().double().pow(2).mean().sqrt())
            a = t.detach().abs()
            a = a[a > 0]                       # an exact zero (zero input under a dropped channel) is a reproducible decision
            self.gate_margins.append((name, (float(a.min()) if a.numel() else float("inf")) / max(rms, 1e-30)))
        return F.relu(t)

    def note_pool(self, name, x):
        """2x2 max-pool routing: margin = smallest gap between the two largest values of a window"""
        if self.record_gates:
            n, c, h, w = x.shape
            win = x.detach().reshape(n, c, h // 2, 2, w // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, c, h // 2, w // 2, 4)
            top = win.topk(2, dim=-1).values
            rms = float(x.detach().double().pow(2).mean().sqrt())
            gap = top[..., 0] - top[..., 1]
            gap = gap[gap > 0]                 # an exact tie goes to the first maximum in every implementation
            self.gate_margins.append((name, (float(gap.min()) if gap.numel() else float("inf")) / max(rms, 1e-30)))


def _bn(sd: SD, p: str, x: torch.Tensor, st: RunState) -> torch.Tensor:
    g, b = sd[p + ".weight"], sd[p + ".bias"]
    rm, rv = sd[p + ".running_mean"], sd[p + ".running_var"]
    if not st.training:
        return F.batch_norm(x, rm, rv, g, b, False, BN_MOMENTUM, BN_EPS)
    # batch statistics: biased variance normalises, the unbiased one feeds running_var
    rm2, rv2 = rm.detach().clone(), rv.detach().clone()
    y = F.batch_norm(x, rm2, rv2, g, b, True, BN_MOMENTUM, BN_EPS)
    if st.update_running:
        st.new_running[p + ".running_mean"] = rm2
        st.new_running[p + ".running_var"] = rv2
        st.new_running[p + ".num_batches_tracked"] = sd[p + ".num_batches_tracked"] + 1
    return y


def downsampler(sd: SD, p: str, x: torch.Tensor, st: RunState) -> torch.Tensor:
    # the 3-channel initial block reads the fp32 image with fp32 weights; every other layer's weights are stored rounded
    wt = sd[p + ".conv.weight"] if x.shape[1] == 3 else st.w(sd[p + ".conv.weight"])
    conv = F.conv2d(x, wt, sd[p + ".conv.bias"], stride=2, padding=1)
    st.note_pool(p + ".pool", x)
    pool = F.max_pool2d(x, 2, stride=2)
    z = st.r(torch.cat([conv, pool], 1))
    return st.r(st.relu(p + ".relu", _bn(sd, p + ".bn", z, st)))


def nb1d(sd: SD, p: str, x: torch.Tensor, dilation: int, dropprob: float, st: RunState) -> torch.Tensor:
    d = dilation
    y = st.r(st.relu(p + ".relu1", F.conv2d(x, st.w(sd[p + ".conv3x1_1.weight"]), sd[p + ".conv3x1_1.bias"], padding=(1, 0))))
    y = st.r(F.conv2d(y, st.w(sd[p + ".conv1x3_1.weight"]), sd[p + ".conv1x3_1.bias"], padding=(0, 1)))
    y = st.r(st.relu(p + ".relu2", _bn(sd, p + ".bn1", y, st)))
    y = st.r(st.relu(p + ".relu3", F.conv2d(y, st.w(sd[p + ".conv3x1_2.weight"]), sd[p + ".conv3x1_2.bias"], padding=(d, 0),
                                            dilation=(d, 1))))
    y = st.r(F.conv2d(y, st.w(sd[p + ".conv1x3_2.weight"]), sd[p + ".conv1x3_2.bias"], padding=(0, d), dilation=(1, d)))
    y = _bn(sd, p + ".bn2", y, st)
    if dropprob != 0 and st.training:
        if p in st.dropout_masks:
            m = st.dropout_masks[p]
        else:
            keep = (torch.rand(y.shape[0], y.shape[1], 1, 1) >= dropprob).to(y.dtype)
            m = keep / (1.0 - dropprob)
        st.used_masks[p] = m
        y = y * m
    return st.r(st.relu(p + ".relu4", y + x))


def upsampler(sd: SD, p: str, x: torch.Tensor, st: RunState) -> torch.Tensor:
    y = st.r(F.conv_transpose2d(x, st.w(sd[p + ".conv.weight"]), sd[p + ".conv.bias"], stride=2, padding=1, output_padding=1))
    return st.r(st.relu(p + ".relu", _bn(sd, p + ".bn", y, st)))


def _run_layers(sd: SD, prefix: str, layers, x: torch.Tensor, st: RunState, dropout_override=None) -> torch.Tensor:
    for i, (kind, _cin, _cout, p, d) in enumerate(layers):
        name = f"{prefix}.layers.{i}"
        if kind == "down":
            x = downsampler(sd, name, x, st)
        elif kind == "up":
            x = upsampler(sd, name, x, st)
        else:
            x = nb1d(sd, name, x, d, p if dropout_override is None else dropout_override, st)
    return x


def encoder(sd: SD, x: torch.Tensor, st: RunState, predict: bool = False, prefix: str = "encoder",
            dropout_override=None) -> torch.Tensor:
    x = downsampler(sd, prefix + ".initial_block", x, st)
    x = _run_layers(sd, prefix, ENCODER_LAYERS, x, st, dropout_override)
    if predict:
        x = st.g(F.conv2d(x, st.w(sd[prefix + ".output_conv.weight"]), sd[prefix + ".output_conv.bias"]))
    return x


def decoder(sd: SD, x: torch.Tensor, st: RunState, prefix: str = "decoder") -> torch.Tensor:
    x = _run_layers(sd, prefix, DECODER_LAYERS, x, st)
    return st.g(F.conv_transpose2d(x, st.w(sd[prefix + ".output_conv.weight"]), sd[prefix + ".output_conv.bias"], stride=2))


def erfnet(sd: SD, x: torch.Tensor, st: Optional[RunState] = None, only_encode: bool = False) -> torch.Tensor:
    st = st or RunState()
    if only_encode:
        return encoder(sd, x, st, predict=True)
    return decoder(sd, encoder(sd, x, st), st)


def downsampler_simplified(sd: SD, p: str, x: torch.Tensor, st: RunState) -> torch.Tensor:
    """train/multi_models.py:23-36: conv3x3 stride 1 -> maxpool 2x2 -> BatchNorm -> ReLU."""
    y = F.conv2d(x, sd[p + ".conv.weight"], sd[p + ".conv.bias"], stride=1, padding=1)
    return F.relu(_bn(sd, p + ".bn", F.max_pool2d(y, 2, stride=2), st))


def _skip_decoder(sd: SD, p: str, x, s1, s2, st: RunState) -> torch.Tensor:
    """train/multi_models.py:116-178 (both regression decoders share this graph)."""
    x = upsampler(sd, p + ".upblock_1.0", x, st)
    x = nb1d(sd, p + ".upblock_1.1", x, 1, 0.0, st)
    x = nb1d(sd, p + ".upblock_1.2", x, 1, 0.0, st)
    x = torch.cat([x, s2], dim=1)
    x = upsampler(sd, p + ".upblock_2.0", x, st)
    x = nb1d(sd, p + ".upblock_2.1", x, 1, 0.0, st)
    x = nb1d(sd, p + ".upblock_2.2", x, 1, 0.0, st)
    x = torch.cat([x, s1], dim=1)
    return F.conv_transpose2d(x, sd[p + ".upblock_3.0.weight"], sd[p + ".upblock_3.0.bias"], stride=2)


def multi_models(sd: SD, x: torch.Tensor, st: Optional[RunState] = None):
    """train/multi_models.py:205-221, eval mode: (traversability map, depth map, class logits)."""
    st = st or RunState()
    if st.training:
        raise NotImplementedError("the multi-task oracle restates the eval-mode forward only")
    s1 = downsampler_simplified(sd, "encoder.downblock_1", x, st)
    y = downsampler_simplified(sd, "encoder.downblock_2.0", s1, st)
    for i in range(1, 4):
        y = nb1d(sd, f"encoder.downblock_2.{i}", y, 1, 0.03, st)
    s2 = y
    y = downsampler_simplified(sd, "encoder.downblock_3.0", y, st)
    for i, d in enumerate((2, 4, 8, 16)):
        y = nb1d(sd, f"encoder.downblock_3.{i + 1}", y, d, 0.3, st)
    trav = _skip_decoder(sd, "decoder_traversability", y, s1, s2, st)
    depth = _skip_decoder(sd, "decoder_depth", y, s1, s2, st)
    c = F.conv2d(y, sd["classifier.classifier.0.weight"], sd["classifier.classifier.0.bias"], stride=2, padding=1)
    c = F.relu(_bn(sd, "classifier.classifier.1", c, st))            # Dropout2d(0.3) is the identity in eval mode
    c = c.mean(3).mean(2)
    return trav, depth, F.linear(c, sd["classifier.linear.weight"], sd["classifier.linear.bias"])


def erfnet_imagenet(sd: SD, x: torch.Tensor, st: Optional[RunState] = None) -> torch.Tensor:
    """train/erfnet_imagenet.py:64-129: encoder (dropout 0.1, no output_conv) -> maxpool2 -> avgpool14 -> linear."""
    st = st or RunState()
    x = encoder(sd, x, st, prefix="features.encoder", dropout_override=0.1)
    x = F.avg_pool2d(F.max_pool2d(x, 2, stride=2), 14, 1, 0)
    return F.linear(x.reshape(x.shape[0], 128), sd["classifier.linear.weight"], sd["classifier.linear.bias"])


# ----------------------------------------------------------------------------- loss / metric

def weighted_ce_parts(logits: torch.Tensor, target: torch.Tensor, weight: Optional[torch.Tensor]):
    """(sum_p w[t_p] * nll_p, sum_p w[t_p]) -- the numerator and denominator of
    torch.nn.CrossEntropyLoss(weight) with reduction='mean' (train/main.py:121)."""
    logp = F.log_softmax(logits if logits.dtype == torch.float64 else logits.float(), dim=1)
    nll = -logp.gather(1, target[:, None]).squeeze(1)
    w = weight[target] if weight is not None else torch.ones_like(nll)
    return (w * nll).sum(), w.sum()


def weighted_ce(logits, target, weight=None):
    num, den = weighted_ce_parts(logits, target, weight)
    return num / den


def argmax_labels(logits: torch.Tensor) -> torch.Tensor:
    """outputs.max(1)[1].unsqueeze(1) (eval/eval_iou.py:93): first maximal index wins."""
    return logits.max(1)[1].unsqueeze(1)


def iou_counts(pred: torch.Tensor, label: torch.Tensor, n_classes: int, ignore_index: int = 19):
    """tp/fp/fn per class as float64 (train/iouEval.py:21-69), from [N,1,H,W] int64 maps.

    Integer restatement of the one-hot products: tp_c = #(pred=c & label=c),
    fp_c = #(pred=c & label!=c & label!=ignore), fn_c = #(pred!=c & label=c).
    """
    ign = ignore_index if n_classes > ignore_index else -1
    classes = n_classes if ign == -1 else n_classes - 1
    p = pred.reshape(-1).long()
    g = label.reshape(-1).long()
    hist = torch.bincount(g * n_classes + p, minlength=n_classes * n_classes).reshape(n_classes, n_classes).double()
    tp = torch.zeros(classes, dtype=torch.float64)
    fp = torch.zeros(classes, dtype=torch.float64)
    fn = torch.zeros(classes, dtype=torch.float64)
    for c in range(classes):
        tp[c] = hist[c, c]
        fn[c] = hist[c, :].sum() - hist[c, c]
        col = hist[:, c].sum() - hist[c, c]
        if ign != -1:
            col = col - hist[ign, c]
        fp[c] = col
    return tp, fp, fn


def iou_plane_sums(x: torch.Tensor, y: torch.Tensor, n_classes: int, ignore_index: int = 19):
    """tp/fp/fn per class (float64) when x and/or y come as N x nClasses x H x W planes (train/iouEval.py:32-65): label maps
    are expanded to one-hot planes, planes are taken as they are; products in float32 as in the reference, sums in float64."""
    ign = ignore_index if n_classes > ignore_index else -1

    def planes(t):
        if t.size(1) == 1:
            return torch.zeros(t.size(0), n_classes, t.size(2), t.size(3)).scatter_(1, t.long(), 1)
        return t.float()
    xo, yo = planes(x), planes(y)
    ignores = 0
    if ign != -1:
        ignores = yo[:, ign].unsqueeze(1)
        xo, yo = xo[:, :ign], yo[:, :ign]
    tp = (xo * yo).double().sum(dim=(0, 2, 3))
    fp = (xo * (1 - yo - ignores)).double().sum(dim=(0, 2, 3))
    fn = ((1 - xo) * yo).double().sum(dim=(0, 2, 3))
    return tp, fp, fn


def iou_from_counts(tp, fp, fn):
    iou = tp / (tp + fp + fn + 1e-15)        # train/iouEval.py:71-75
    return iou.mean(), iou


# ----------------------------------------------------------------------------- optimiser

def poly_lr(base_lr: float, epoch: int, num_epochs: int) -> float:
    """train/main.py:156-157 with scheduler.step(epoch) at :165."""
    return base_lr * math.pow(1 - ((epoch - 1) / num_epochs), 0.9)


def adam_step(params: List[torch.Tensor], grads: List[Optional[torch.Tensor]], exp_avg, exp_avg_sq, step: int,
              lr=5e-4, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-4):
    """torch.optim.Adam semantics as configured at train/main.py:140 (coupled L2
    decay; tensors whose grad is None are skipped entirely).  In place; step is 1-based."""
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    for p, g, m, v in zip(params, grads, exp_avg, exp_avg_sq):
        if g is None:
            continue
        g = g + weight_decay * p
        m.mul_(beta1).add_(g, alpha=1 - beta1)
        v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m, denom, value=-lr / bc1)


def sgd_step(params, grads, bufs, lr=0.1, momentum=0.9, weight_decay=1e-4):
    """torch.optim.SGD as configured at imagenet/main.py:83-85."""
    for i, (p, g) in enumerate(zip(params, grads)):
        if g is None:
            continue
        g = g + weight_decay * p
        if bufs[i] is None:
            bufs[i] = g.clone()
        else:
            bufs[i].mul_(momentum).add_(g)
        p.add_(bufs[i], alpha=-lr)


# ----------------------------------------------------------------------------- training step helpers

def train_step_grads(sd: SD, images, target, weight, masks=None, num_classes=20, only_encode=False, storage_dtype=None):
    """One forward/backward in training mode.  Returns (loss, grads dict, new running stats, used masks).
    `only_encode`: the encoder stage of the recipe (train/main.py:203, `model(inputs, only_encode=enc)`)."""
    leaf = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v)
            for k, v in sd.items()}
    st = RunState(training=True, dropout_masks=masks, storage_dtype=storage_dtype)
    logits = erfnet(leaf, images, st, only_encode=only_encode)
    loss = weighted_ce(logits, target, weight)
    loss.backward()
    grads = {k: v.grad for k, v in leaf.items() if isinstance(v, torch.Tensor) and v.requires_grad}
    return loss.detach(), grads, st.new_running, st.used_masks, logits.detach()


# ----------------------------------------------------------------------------- key/shape templates

def _bn_t(p, c):
    return {p + ".weight": (c,), p + ".bias": (c,), p + ".running_mean": (c,), p + ".running_var": (c,),
            p + ".num_batches_tracked": ()}


def down_template(p, cin, cout):
    t = {p + ".conv.weight": (cout - cin, cin, 3, 3), p + ".conv.bias": (cout - cin,)}
    t.update(_bn_t(p + ".bn", cout))
    return t


def nb_template(p, c, bn_last=False):
    """bn_last: registration order of train/erfnet_imagenet.py:30-40 (both BNs after the four convs)."""
    t = {}
    for name, k in (("conv3x1_1", (3, 1)), ("conv1x3_1", (1, 3))):
        t[f"{p}.{name}.weight"] = (c, c) + k
        t[f"{p}.{name}.bias"] = (c,)
    if not bn_last:
        t.update(_bn_t(p + ".bn1", c))
    for name, k in (("conv3x1_2", (3, 1)), ("conv1x3_2", (1, 3))):
        t[f"{p}.{name}.weight"] = (c, c) + k
        t[f"{p}.{name}.bias"] = (c,)
    if bn_last:
        t.update(_bn_t(p + ".bn1", c))
    t.update(_bn_t(p + ".bn2", c))
    return t


def up_template(p, cin, cout):
    t = {p + ".conv.weight": (cin, cout, 3, 3), p + ".conv.bias": (cout,)}
    t.update(_bn_t(p + ".bn", cout))
    return t


def _layers_template(prefix, layers, bn_last=False):
    t = {}
    for i, (kind, cin, cout, _p, _d) in enumerate(layers):
        name = f"{prefix}.layers.{i}"
        t.update(down_template(name, cin, cout) if kind == "down" else
                 up_template(name, cin, cout) if kind == "up" else nb_template(name, cin, bn_last))
    return t


def encoder_template(num_classes, prefix="encoder", with_output_conv=True, bn_last=False):
    t = down_template(prefix + ".initial_block", 3, 16)
    t.update(_layers_template(prefix, ENCODER_LAYERS, bn_last))
    if with_output_conv:
        t[prefix + ".output_conv.weight"] = (num_classes, 128, 1, 1)
        t[prefix + ".output_conv.bias"] = (num_classes,)
    return t


def erfnet_template(num_classes):
    """Keys and shapes of ERFNet(num_classes).state_dict(), in registration order."""
    t = encoder_template(num_classes)
    t.update(_layers_template("decoder", DECODER_LAYERS))
    t["decoder.output_conv.weight"] = (16, num_classes, 2, 2)
    t["decoder.output_conv.bias"] = (num_classes,)
    return t


def imagenet_template(num_classes=1000):
    t = encoder_template(0, prefix="features.encoder", with_output_conv=False, bn_last=True)
    t["classifier.linear.weight"] = (num_classes, 128)
    t["classifier.linear.bias"] = (num_classes,)
    return t


def multi_models_template(num_classes=14):
    """Keys and shapes of Multi_models(num_classes).state_dict() (train/multi_models.py:205-213), in registration order."""
    def simp(p, cin, cout):
        t = {p + ".conv.weight": (cout, cin, 3, 3), p + ".conv.bias": (cout,)}
        t.update(_bn_t(p + ".bn", cout))
        return t
    t = simp("encoder.downblock_1", 3, 16)
    t.update(simp("encoder.downblock_2.0", 16, 64))
    for i in range(1, 4):
        t.update(nb_template(f"encoder.downblock_2.{i}", 64))
    t.update(simp("encoder.downblock_3.0", 64, 128))
    for i in range(1, 5):
        t.update(nb_template(f"encoder.downblock_3.{i}", 128))
    for dec in ("decoder_traversability", "decoder_depth"):
        t.update(up_template(dec + ".upblock_1.0", 128, 64))
        t.update(nb_template(dec + ".upblock_1.1", 64))
        t.update(nb_template(dec + ".upblock_1.2", 64))
        t.update(up_template(dec + ".upblock_2.0", 128, 16))
        t.update(nb_template(dec + ".upblock_2.1", 16))
        t.update(nb_template(dec + ".upblock_2.2", 16))
        t[dec + ".upblock_3.0.weight"] = (32, 1, 2, 2)
        t[dec + ".upblock_3.0.bias"] = (1,)
    t["classifier.classifier.0.weight"] = (64, 128, 3, 3)
    t["classifier.classifier.0.bias"] = (64,)
    t.update(_bn_t("classifier.classifier.1", 64))
    t["classifier.linear.weight"] = (num_classes, 64)
    t["classifier.linear.bias"] = (num_classes,)
    return t


def synthetic_state(template, salt=0):
    from erfnet_pytorch_amd import synthetic as S
    return {k: S.fill_tensor(k, shp, salt) for k, shp in template.items()}
