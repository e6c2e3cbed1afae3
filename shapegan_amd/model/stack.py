"""Executor that runs an nn.Sequential of *parameter containers* on the HIP kernels.

The model classes keep torch.nn layer objects (nn.Conv3d, nn.BatchNorm3d, nn.Linear, ...) purely as owners of
parameters and buffers: that gives the reference's state_dict keys, default initialisation (same RNG draws under
the same seed) and .to()/.cuda() behaviour for free.  Their ATen forward is never called — `run_stack` walks the
container, groups [conv|linear] -> [batchnorm] -> [activation] runs and dispatches each group to one fused
shapegan_amd.ops call (bias and activation in the GEMM epilogue, LeakyReLU folded into the batch-norm apply pass).
"""
import torch.nn as nn

from .. import ops
from ..lib import ACT_LEAKY, ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH
from . import Lambda


def _act_of(m):
    if isinstance(m, nn.LeakyReLU):
        return ACT_LEAKY, float(m.negative_slope)
    if isinstance(m, nn.ReLU):
        return ACT_RELU, 0.0
    if isinstance(m, nn.Tanh):
        return ACT_TANH, 0.0
    if isinstance(m, nn.Sigmoid):
        return ACT_SIGMOID, 0.0
    return None


def _is_k4(m, stride, padding):
    return tuple(m.kernel_size) == (4, 4, 4) and tuple(m.stride) == (stride,) * 3 and tuple(m.padding) == (padding,) * 3


def _producer(m, x, act, slope, out=None):
    """One conv / linear layer with `act` fused into its epilogue (out: see ops.conv_transpose3d_k4s2p1)."""
    if isinstance(m, nn.Conv3d):
        if _is_k4(m, 2, 1):
            return ops.conv3d_k4s2p1(x, m.weight, m.bias, act, slope)
        if _is_k4(m, 1, 0) and tuple(x.shape[2:]) == (4, 4, 4):
            # 4^3 -> 1^3: a GEMM over (ci, tap)  (model/gan.py:55, model/autoencoder.py:28)
            y = ops.LinearAct.apply(x.reshape(x.shape[0], -1), m.weight.reshape(m.out_channels, -1), m.bias, act, slope,
                                    False, 0)
            return y.reshape(x.shape[0], m.out_channels, 1, 1, 1)
    elif isinstance(m, nn.ConvTranspose3d):
        if _is_k4(m, 2, 1):
            return ops.conv_transpose3d_k4s2p1(x, m.weight, m.bias, act, slope, out)
        if _is_k4(m, 1, 0) and tuple(x.shape[2:]) == (1, 1, 1):
            # 1^3 -> 4^3: y[b, co*64+tap] = x[b,:] @ W[:, co*64+tap] + bias[co]  (model/gan.py:9, autoencoder.py:51)
            y = ops.LinearAct.apply(x.reshape(x.shape[0], -1), m.weight.reshape(m.in_channels, -1), m.bias, act, slope,
                                    True, 6)
            return y.reshape(x.shape[0], m.out_channels, 4, 4, 4)
    elif isinstance(m, nn.Linear):
        return ops.LinearAct.apply(x, m.weight, m.bias, act, slope, False, 0)
    raise NotImplementedError("shapegan_amd has no HIP kernel for %r on input %s" % (m, tuple(x.shape)))


def _batchnorm(m, x, act, slope, training):
    use_batch_stats = training or not m.track_running_stats
    if m.momentum is None:
        # torch switches to a cumulative moving average (factor 1/num_batches_tracked) here; no reference model uses it
        raise NotImplementedError("shapegan_amd BatchNorm: momentum=None (cumulative average) is not implemented")
    momentum = m.momentum
    return ops.BatchNormAct.apply(x, m.weight, m.bias, m.running_mean, m.running_var,
                                  m.num_batches_tracked if use_batch_stats else None, use_batch_stats, m.eps, momentum,
                                  act, slope)


# Switch for the inference-mode fusion below (A/B, and for tests that replay a RECORDED fp32 trajectory: a different rounding of
# the generator's samples may flip a LeakyReLU pre-activation that sits within 1e-7 of zero in the critic, DESIGN.md 3.3 "kinks")
FUSE_BN_INTO_LAST_CONV_TRANSPOSE = True


def _bn_into_last_conv_transpose(mods, i, raw, training, out):
    """[.., BatchNorm3d, LeakyReLU / ReLU, ConvTranspose3d(C -> 1, k4 s2 p1) (, activation)] at the end of a generator / decoder
    (model/gan.py:18-22, model/autoencoder.py:60-63) WITHOUT grad mode and with batch statistics: the statistics are taken from
    the producer's raw output `raw` (running buffers updated as torch does), and normalisation + activation are applied inside the
    last transposed convolution's loads — the normalised tensor (67 MB at batch 64) is neither written nor read back.  Returns
    (result, next index) or None when the pattern / mode does not apply (the unfused path is taken)."""
    import torch
    bn = mods[i + 1]
    if not FUSE_BN_INTO_LAST_CONV_TRANSPOSE or torch.is_grad_enabled() or not isinstance(bn, nn.BatchNorm3d) or i + 3 >= len(mods):
        return None
    if not (training or not bn.track_running_stats) or bn.momentum is None or not bn.affine:
        return None
    a = _act_of(mods[i + 2])
    last = mods[i + 3]
    if a is None or a[0] not in (ACT_LEAKY, ACT_RELU) or not (0.0 <= a[1] <= 1.0):
        return None
    if not (isinstance(last, nn.ConvTranspose3d) and _is_k4(last, 2, 1) and last.out_channels == 1
            and ops.convT_to1_pre_served(raw, last.weight)):
        return None
    nxt_i = i + 4
    a2 = _act_of(mods[nxt_i]) if nxt_i < len(mods) else None
    if a2 is not None:
        nxt_i += 1
    scale, shift = ops.bn_train_stats_affine(raw, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                             bn.num_batches_tracked if bn.track_running_stats else None, bn.eps, bn.momentum)
    y = ops.conv_transpose3d_to1_pre_raw(raw, scale, shift, a[0], a[1], last.weight, last.bias,
                                         a2[0] if a2 is not None else ACT_NONE, a2[1] if a2 is not None else 0.0,
                                         out if nxt_i == len(mods) else None)
    return y, nxt_i


def run_stack(modules, x, training, out=None):
    """out: optional destination of the LAST layer's result when that is a k4 s2 p1 transposed convolution (+ activation) and grad
    mode is off; ignored otherwise (the caller checks what it got back)."""
    mods = list(modules)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, (nn.Conv3d, nn.ConvTranspose3d, nn.Linear)):
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            if isinstance(nxt, (nn.BatchNorm3d, nn.BatchNorm1d)):
                x = _producer(m, x, ACT_NONE, 0.0)
                fused = _bn_into_last_conv_transpose(mods, i, x, training, out)
                if fused is not None:
                    x, i = fused
                    continue
                a = _act_of(mods[i + 2]) if i + 2 < len(mods) else None
                if a is not None:
                    x = _batchnorm(nxt, x, a[0], a[1], training)
                    i += 3
                else:
                    x = _batchnorm(nxt, x, ACT_NONE, 0.0, training)
                    i += 2
                continue
            a = _act_of(nxt) if nxt is not None else None
            head = mods[i + 2] if a is not None and i + 2 < len(mods) else None
            if (isinstance(m, nn.Conv3d) and _is_k4(m, 2, 1) and isinstance(head, nn.Conv3d) and _is_k4(head, 1, 0)
                    and x.dim() == 5 and ops.head_dot_served(x, head.weight, a[0])):
                # Conv3d(k4 s2 p1) -> activation -> Conv3d(C -> 1, k4 s1) on the 4^3 grid (model/gan.py:53-55): one node; the
                # activation rides in the head's loads, its backward and both bias gradients in the head's one backward pass
                y = ops.conv_head(x, m.weight, m.bias, a[0], a[1], head.weight, head.bias)
                x = y.reshape(x.shape[0], 1, 1, 1, 1)
                i += 3
                continue
            if a is not None:
                x = _producer(m, x, a[0], a[1], out if i + 2 == len(mods) else None)
                i += 2
            else:
                x = _producer(m, x, ACT_NONE, 0.0, out if i + 1 == len(mods) else None)
                i += 1
            continue
        if isinstance(m, (nn.BatchNorm3d, nn.BatchNorm1d)):
            a = _act_of(mods[i + 1]) if i + 1 < len(mods) else None
            if a is not None:
                x = _batchnorm(m, x, a[0], a[1], training)
                i += 2
            else:
                x = _batchnorm(m, x, ACT_NONE, 0.0, training)
                i += 1
            continue
        a = _act_of(m)
        if a is not None:
            x = ops.Act.apply(x, a[0], a[1])
        elif isinstance(m, Lambda):
            x = m.function(x)
        else:
            raise NotImplementedError("shapegan_amd has no HIP kernel for %r" % (m,))
        i += 1
    return x


def _shape_after(plan, shape):
    """Shape of the tensor that leaves the last producer of `plan` ([(producer, BatchNorm3d, act), ...]; BatchNorm and activations
    keep shapes) for an input of `shape`; None if a producer cannot take it."""
    for m, _, _ in plan:
        if isinstance(m, nn.Linear):
            if shape[-1] != m.in_features:
                return None
            shape = shape[:-1] + (m.out_features,)
            continue
        if len(shape) != 5 or shape[1] != m.in_channels:
            return None
        k, st, pd = m.kernel_size, m.stride, m.padding
        if isinstance(m, nn.ConvTranspose3d):
            if tuple(m.output_padding) != (0, 0, 0) or tuple(m.dilation) != (1, 1, 1):
                return None
            sp = tuple((shape[2 + i] - 1) * st[i] - 2 * pd[i] + k[i] for i in range(3))
        else:
            if tuple(m.dilation) != (1, 1, 1):
                return None
            sp = tuple((shape[2 + i] + 2 * pd[i] - k[i]) // st[i] + 1 for i in range(3))
        if min(sp) < 1:
            return None
        shape = (shape[0], m.out_channels) + sp
    return shape


def run_stack_groups(modules, x, groups, outs):
    """`groups` independent evaluations of a generator / decoder stack in ONE pass, WITHOUT grad mode and with batch statistics:
    x [groups * B, ...] holds the groups' inputs one after the other; every BatchNorm3d normalises each group with that group's own
    statistics and updates its running buffers group after group (sg_bn_train_fwd_grouped), so the result equals `groups`
    separate calls — but every convolution runs once on groups * B samples and the BatchNorm passes are one launch pair for all
    groups.  The stack must end in [BatchNorm3d, LeakyReLU / ReLU, ConvTranspose3d(C -> 1, k4 s2 p1) (, activation)] (model/gan.py
    :18-22); group g's samples are written to outs[g].  Returns False when the stack does not have that shape (the caller then
    evaluates the groups one by one)."""
    import torch
    mods = list(modules)
    if torch.is_grad_enabled() or not FUSE_BN_INTO_LAST_CONV_TRANSPOSE or groups < 2:
        return False
    # shape check first: [producer, BN3d, act] * k, then ConvTranspose3d(C -> 1) (+ act)
    i, plan = 0, []
    while i + 2 < len(mods) and isinstance(mods[i], (nn.Conv3d, nn.ConvTranspose3d, nn.Linear)) \
            and isinstance(mods[i + 1], nn.BatchNorm3d) and _act_of(mods[i + 2]) is not None:
        bn = mods[i + 1]
        if bn.momentum is None or not bn.affine or not bn.track_running_stats:
            return False
        plan.append((mods[i], bn, _act_of(mods[i + 2])))
        i += 3
    if not plan or i >= len(mods):
        return False
    last = mods[i]
    a_out = _act_of(mods[i + 1]) if i + 1 < len(mods) else None
    if i + 1 + (a_out is not None) != len(mods):
        return False
    if not (isinstance(last, nn.ConvTranspose3d) and _is_k4(last, 2, 1) and last.out_channels == 1):
        return False
    a_in = plan[-1][2]
    if a_in[0] not in (ACT_LEAKY, ACT_RELU) or not (0.0 <= a_in[1] <= 1.0):
        return False
    # the last layer must be served by sg_convT3d_k4s2p1_to1_pre at the shape its input WILL have: decided from the plan before
    # anything is launched — a refusal behind the grouped BatchNorms would come with their running buffers already updated for
    # every group, while refusing here lets the caller fall back to one evaluation per group
    shape = _shape_after(plan, tuple(x.shape))
    if shape is None or len(shape) != 5 or last.weight.shape[1] != 1 or not ops.convT_to1_pre_eligible(*shape):
        return False
    for k, (m, bn, a) in enumerate(plan):
        x = _producer(m, x, ACT_NONE, 0.0)
        if k + 1 < len(plan):
            x = ops.bn_train_fwd_grouped_raw(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.eps,
                                             bn.momentum, a[0], a[1], groups)
    if tuple(x.shape) != shape:
        raise RuntimeError("run_stack_groups: planned %s, got %s" % (shape, tuple(x.shape)))
    bn = plan[-1][1]
    scale, shift = ops.bn_train_stats_affine(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.eps,
                                             bn.momentum, groups)
    ops.conv_transpose3d_to1_pre_raw(x, scale, shift, a_in[0], a_in[1], last.weight, last.bias,
                                     a_out[0] if a_out is not None else ACT_NONE, a_out[1] if a_out is not None else 0.0, outs=outs)
    return True
