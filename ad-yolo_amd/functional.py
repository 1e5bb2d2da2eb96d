"""Block-granular autograd nodes: each forward/backward is a hand-ordered sequence of gfx950 kernel
launches (ops.py -> libadyolo_hip.so).  torch.autograd only chains the nodes and accumulates the
parameter gradients, so ``loss.backward()`` works exactly as in the reference's train loop
(/root/reference/src/train.py:49-55).

Activations are channels-last float32: [N][T][F][C].
"""
import os

import torch

from . import ops

# Fusion switches (all on by default; the env override exists so that A/B timings can be taken in one process)
FUSE_AFFINE = os.environ.get("ADYOLO_FUSE_AFFINE", "1") != "0"   # BN1 affine applied while conv2 stages its input
FUSE_STATS = os.environ.get("ADYOLO_FUSE_STATS", "1") != "0"     # BN statistics from the conv epilogue
FUSE_DR = os.environ.get("ADYOLO_FUSE_DR", "1") != "0"           # identity-shortcut gradient formed in the dgrad epilogue
# BN1 backward sums from the dgrad(conv2) epilogue (no separate read pass over dy and the BatchNorm input).  With the direct
# kernel's scalar epilogue and a one-stage tile reduction this was 7 % slower; with the Winograd kernel's float4 epilogue and
# the two-stage adyolo_bn_bwd_tiles it measures 173.5 -> 171.0 ms per step, so it is on (ADYOLO_FUSE_BNBWD=0 switches it off)
FUSE_BNBWD = os.environ.get("ADYOLO_FUSE_BNBWD", "1") != "0"
# SE / BN2 backward sums of block A from the dgrad(conv1) epilogue of the identity-shortcut block B that follows it: the
# launch that produces dA = conv1_dgrad(da_B) + de_B * (e_B > 0) also sums dA * (e_A > 0) and dA * (e_A > 0) * xhat(c_A) per
# patch (stat_mask = e_A = B's input, stat_aux = c_A), so A's backward skips its three-tensor reduction pass.  Active for 12
# of the 16 blocks, bit-compatible with the unfused path in the golden tests, but the two extra tensors the epilogue
# reads cost what the removed pass saved (168.5 vs 168.3 ms per step) in round 1.  Round 2: with the ReLU masks read as
# BITS (stat_mask and addend_mask of the epilogue move 1/32 of the bytes) the fusion wins 0.3-0.5 ms per step
# (160.45 / 160.72 -> 160.18 ms in one session) and is on; ADYOLO_FUSE_SEBWD=0 switches it off.
FUSE_SEBWD = os.environ.get("ADYOLO_FUSE_SEBWD", "1") != "0"
# The block's final ReLU mask (e > 0) is written as bits by se_tail_fwd (1/32 of the bytes of e) and the two backward passes
# of the SE tail read the bits instead of e: 7 -> 5.06 tensor passes for se_tail_bwd.
FUSE_MASKBITS = os.environ.get("ADYOLO_FUSE_MASKBITS", "1") != "0"
# The stem's BatchNorm output is never written: the stem hands relu(conv(x)) and (scale, shift) to the first block, which
# applies the affine while conv1 / its weight-gradient stage the tensor and while the SE tail reads the identity shortcut.
FUSE_STEM_AFFINE = os.environ.get("ADYOLO_FUSE_STEM_AFFINE", "1") != "0"
# the tail of the last block in front of a pooled stage boundary writes avgpool2(e) + the mask bits of e, never e (round 6)
FUSE_POOL = os.environ.get("ADYOLO_FUSE_POOL", "1") != "0"
FUSE_POOL_BWD = os.environ.get("ADYOLO_FUSE_POOL_BWD", "1") != "0"      # ... and its backward never reads a full-size de


class BlockLink:
    """Side channel between two consecutive SE blocks (A feeds only B, B has an identity shortcut and no pooling):
    A.forward publishes (c_A, mean2_A, invstd2_A); B.backward leaves the per-patch sums for A.backward."""

    def __init__(self):
        self.cc = self.mean2 = self.invstd2 = self.tiles = self.ebits = None
        self.affine = self.stem_bn = None        # (stem hand-over: see StemFn / FUSE_STEM_AFFINE)
        self.prepooled = False                   # A wrote avgpool2(e) instead of e (FUSE_POOL): B must not pool again


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


_CONST = {}


def _const(value, shape, device):
    """A cached read-only float32 tensor filled with ``value`` (identity affines of ReluFn / residual BatchNormFn: two fill
    launches per call otherwise, ~100 per ResNet-Conformer step)."""
    key = (float(value), tuple(shape), str(device))
    t = _CONST.get(key)
    if t is None:
        t = torch.full(tuple(shape), float(value), dtype=torch.float32, device=device)
        _CONST[key] = t
    return t


class GradSink:
    """While a ``TrainStep`` runs backward, the big gradient producers write STRAIGHT into the parameter's slice of the flat
    gradient buffer (``dist.FlatParameters``) and return None to autograd for that input: no ``grad += new`` launch per
    parameter (~170 per step), and a data-parallel bucket is complete exactly when its last producer has run
    (``done`` calls the reducer's hook by hand, since autograd's post-accumulate hook does not fire for a None gradient).
    Inactive (views is None) everywhere else: plain autograd semantics, e.g. for ``torch.optim.Adam`` loops and tests."""

    def __init__(self):
        self.views = None          # {param data_ptr: (index in flat.params, grad view)}
        self.notify = None         # index -> None, or None when there is no active reducer
        self.step = 4              # element size

    def begin(self, flat, reducer=None):
        self.views = {p.data_ptr(): (i, p.grad) for i, p in enumerate(flat.params)}
        self.notify = reducer.notify if (reducer is not None and reducer.active) else None

    def end(self):
        self.views = self.notify = None

    def view(self, ptr, shape=None):
        """Gradient slice of the parameter whose storage starts at ``ptr`` (or None when inactive / unknown)."""
        if self.views is None:
            return None
        ent = self.views.get(ptr)
        return None if ent is None else (ent[1] if shape is None else ent[1].view(shape))

    def span(self, ptrs):
        """ONE contiguous gradient view covering the parameters ``ptrs`` if they lie back to back, in this order, in the
        flat buffer; else None."""
        if self.views is None:
            return None
        ents = [self.views.get(q) for q in ptrs]
        if any(e is None for e in ents):
            return None
        base = ents[0][1]
        off = base.data_ptr()
        for _, g in ents:
            if g.data_ptr() != off:
                return None
            off += g.numel() * 4
        n = (off - base.data_ptr()) // 4
        return torch.as_strided(base, (n,), (1,))

    def done(self, *ptrs):
        if self.notify is not None:
            for q in ptrs:
                self.notify(self.views[q][0])

    def params(self, *tensors):
        """Gradient views of ALL the given parameter tensors (None entries allowed and passed through), each checked to be
        the whole parameter (same storage start and element count: a slice or a copy of a parameter must keep going
        through autograd); None when inactive or when any of them is not a sinkable parameter."""
        if self.views is None:
            return None
        out = []
        for t in tensors:
            if t is None:
                out.append(None)
                continue
            ent = self.views.get(t.data_ptr())
            if ent is None or ent[1].numel() != t.numel() or not t.is_contiguous():
                return None
            out.append(ent[1].view(t.shape))
        return out

    def done_params(self, *tensors):
        self.done(*[t.data_ptr() for t in tensors if t is not None])


def saved(grad_fn):
    """{name: tensor} of what a StemFn / SEBlockFn node saved for backward -- the supported way for tests to reach e.g. the
    ReLU output ``a`` (= relu(conv1)), the block output ``e`` or the ReLU-mask bits ``ebits`` of a block (the positional
    layout of ``saved_tensors`` changes whenever a fusion adds or drops an operand)."""
    return dict(zip(grad_fn.saved_names, grad_fn.saved_tensors))


SINK = GradSink()
_COUNTER_SCOPE = []           # innermost active bn_counter_scope's list (empty: counters are bumped immediately)


class bn_counter_scope:
    """``with bn_counter_scope():`` around an encoder forward: the ``num_batches_tracked += 1`` of every BatchNorm that takes
    a training-mode step inside is deferred and applied as ONE multi-tensor launch on exit (36 single-element launches per
    step otherwise).  The list belongs to the scope: a forward that raises drops its pending bumps instead of leaving
    them for another model, and a training-mode BatchNorm used outside any scope is bumped on the spot."""

    def __enter__(self):
        self.pending = []
        _COUNTER_SCOPE.append(self.pending)
        return self

    def __exit__(self, exc_type, exc, tb):
        _COUNTER_SCOPE.pop()
        if exc_type is None and self.pending:
            torch._foreach_add_(self.pending, 1)
        self.pending = []
        return False


def _bump_counter(t):
    if _COUNTER_SCOPE:
        _COUNTER_SCOPE[-1].append(t)
    else:
        t.add_(1)


class _BNState:
    """Batch-norm running buffers of one layer (updated in place by the stats kernel)."""

    def __init__(self, mod):
        self.mod = mod

    def stats(self, x, training):
        m = self.mod
        if training:
            ssum, mean, invstd = ops.bn_stats(x, m.running_mean, m.running_var, m.momentum, m.eps)
            _bump_counter(m.num_batches_tracked)
            ops.params_changed()                 # the running buffers moved (kernels write them without a version bump)
        else:
            ssum = None
            mean, invstd = ops.bn_eval_stats(m.running_mean, m.running_var, m.eps)
        return ssum, mean, invstd

    def eval_affine(self, gamma, beta):
        """Evaluation mode: (mean, invstd, scale, shift) from the running buffers, computed once and kept on the module until a
        parameter or buffer changes (72 four-microsecond launches per SE-ResNet34 forward otherwise: 9 % of a one-clip pass).
        Validity: ``ops.PARAMS_EPOCH`` (bumped by every kernel that writes parameters or buffers in place: Adam, the
        training-mode statistics) and the tensors' own version counters (``load_state_dict`` / ``copy_``)."""
        m = self.mod
        key = (ops.PARAMS_EPOCH[0], m.running_mean.data_ptr(), m.running_mean._version, m.running_var._version,
               gamma.data_ptr(), gamma._version, beta.data_ptr(), beta._version)
        c = m.__dict__.get("_adyolo_eval_affine")
        if c is None or c[0] != key:
            mean, invstd = ops.bn_eval_stats(m.running_mean, m.running_var, m.eps)
            scale, shift = ops.bn_scale_shift(gamma, beta, mean, invstd)
            c = (key, mean, invstd, scale, shift)
            m.__dict__["_adyolo_eval_affine"] = c
        return c[1:]

    def stats_tiles(self, tile_stats, x, update=True, affine=None):
        """Training-mode statistics from the per-patch sums a conv epilogue produced (no extra read pass).
        affine=(gamma, beta): the same finishing launch also returns (scale, shift)."""
        m = self.mod
        n, c = x.shape[0], x.shape[-1]
        hw = x.numel() // (n * c)
        g, b = affine if affine is not None else (None, None)
        if update:
            ops.params_changed()
            out = ops.bn_stats_tiles(tile_stats, n, hw, m.running_mean, m.running_var, m.momentum, m.eps, g, b)
            _bump_counter(m.num_batches_tracked)
            return out
        return ops.bn_stats_tiles(tile_stats, n, hw, None, None, m.momentum, m.eps, g, b)


class StemFn(torch.autograd.Function):
    """conv3x3(7->32, bias) -> ReLU -> BatchNorm   (reference resnet.py:183-185; ReLU before BN)."""

    @staticmethod
    def forward(ctx, x8, w, b, gamma, beta, bn, training, holder=None):
        """holder (a BlockLink-like object) given: the BatchNorm affine is NOT applied here -- the output is relu(conv(x)) and
        holder.affine = (scale, shift) for the consumer (SEBlockFn's ``p_affine``); the incoming gradient is then the
        gradient w.r.t. the affine's output, exactly what the consumer returns for its input."""
        wpk, _ = ops.pack_w3x3(w, x8.shape[-1], want_dgrad=False)      # 8-channel pixels (FOA: 7 features) or 32 (MIC: 10)
        if training and FUSE_STATS:
            a, st = ops.conv3x3(x8, wpk, w.shape[0], bias=b, relu=True, want_stats=True)
            _, mean, invstd, scale, shift = _BNState(bn).stats_tiles(st, a, affine=(gamma, beta))
        else:
            a = ops.conv3x3(x8, wpk, w.shape[0], bias=b, relu=True)
            if training:
                _, mean, invstd = _BNState(bn).stats(a, True)
                scale, shift = ops.bn_scale_shift(gamma, beta, mean, invstd)
            else:
                mean, invstd, scale, shift = _BNState(bn).eval_affine(gamma, beta)
        ctx.holder = None
        if holder is not None:
            holder.affine = (scale, shift)
            if training and FUSE_BNBWD:            # the consumer's dgrad epilogue can sum this BatchNorm's backward statistics
                # (a detached alias: `a` itself is this node's OUTPUT, and node -> holder -> a -> grad_fn would be a
                #  reference cycle that keeps the node and its AccumulateGrad edges alive until the garbage collector
                #  runs -- stale AccumulateGrad nodes from an earlier stream break hipGraph capture of the next step)
                holder.stem_bn = (a.detach(), mean, invstd)
                ctx.holder = holder
            out = a
        else:
            out = ops.affine(a, scale, shift)
        ctx.training = training
        ctx.cin_real = w.shape[1]
        ctx.ptrs = (w.data_ptr(), gamma.data_ptr(), beta.data_ptr())
        ctx.wshape = tuple(w.shape)
        ctx.saved_names = ["x8", "a", "gamma", "mean", "invstd"]
        ctx.save_for_backward(x8, a, gamma, mean, invstd)
        return out

    @staticmethod
    def backward(ctx, dout):
        if not ctx.training:
            raise NotImplementedError("backward through eval-mode BatchNorm is not part of the hot path")
        x8, a, gamma, mean, invstd = ctx.saved_tensors
        pw, pg, pb = ctx.ptrs
        vg, vb, vw = SINK.view(pg), SINK.view(pb), SINK.view(pw, ctx.wshape)
        tiles = None
        if ctx.holder is not None and ctx.holder.tiles is not None:      # left by the first block's dgrad epilogue
            tiles, ctx.holder.tiles = ctx.holder.tiles, None
        da, dgamma, dbeta, db = ops.bn_bwd(_c(dout), a, gamma, mean, invstd, relu_mask=True, tile_stats=tiles,
                                           out_dgamma=vg, out_dbeta=vb, want_dx_colsum=True)   # db = sum of da: the conv bias gradient
        dw = ops.conv3x3_wgrad(x8, da, ctx.cin_real, out=vw)
        if vw is not None:
            SINK.done(pw, pg, pb)
            return None, None, db, None, None, None, None, None
        return None, dw, db, dgamma, dbeta, None, None, None


class SEBlockFn(torch.autograd.Function):
    """SEBasicBlock (reference resnet.py:25-47):
    [AvgPool2d(2,2)] -> conv3x3 -> ReLU -> BN1 -> conv3x3 -> BN2 -> SE -> (+ x | BN(conv1x1(x))) -> ReLU.

    BN2 is never materialised: the SE squeeze comes from the per-sample channel sums the BN statistics
    pass produces anyway (mean_hw(bn2(c)) = scale*mean_hw(c)+shift) and BN2's affine, the SE scale, the
    residual add and the ReLU are one fused pass.
    """

    @staticmethod
    def forward(ctx, x, training, pool, bns, w1, g1, b1, w2, g2, b2, fw1, fb1, fw2, fb2, wd=None, gd=None, bd=None):
        bn1, bn2, bnd, link_in, link_out = bns[:5]
        p_aff = bns[5] if len(bns) > 5 else None      # (scale, shift): the input is seen through this per-channel affine
        ctx.stem_holder = bns[6] if len(bns) > 6 else None    # BlockLink of the stem: its BatchNorm backward sums come from our dgrad
        packs = bns[7] if len(bns) > 7 else None              # (u_fwd1, u_dgrad1, u_fwd2, u_dgrad2) from ops.WinoPackSet, or None
        pool_next = bool(bns[8]) if len(bns) > 8 else False   # the NEXT block starts with AvgPool2d(2, 2): see FUSE_POOL below
        prepooled = pool and link_in is not None and link_in.prepooled   # ... and the block BELOW handed us avgpool2(its output)
        if p_aff is not None and (pool or wd is not None):
            raise NotImplementedError("p_affine is only supported for identity-shortcut blocks without pooling")
        p = ops.avgpool2(x) if (pool and not prepooled) else x
        n, h, w_, cin = p.shape
        c = w1.shape[0]
        if packs is not None:
            # (ops.DualPack -> the form this launch size runs on: F(4x4) when the grid fills the chip, else F(2x2))
            # (the data-gradient of conv1 always carries an addend -- the shortcut's gradient -- in its epilogue)
            # (conv1's data-gradient pack is picked at the end of this forward pass, when its operand combination is known: with
            #  32-channel output blocks only the persistent F(4x4) kernel exists, and it is built for certain combinations)
            wpk1, wpk1d, wpk2, wpk2d = [pk.pick(n, h, w_, co, ad) if (isinstance(pk, ops.DualPack) and i != 1) else pk
                                        for i, (pk, co, ad) in enumerate(zip(packs, (c, cin, c, c), (False, True, False, False)))]
        else:
            wpk1, wpk1d = ops.pack_w3x3(w1, cin)
            wpk2, wpk2d = ops.pack_w3x3(w2, c)
        if training and FUSE_STATS:
            a, st1 = ops.conv3x3(p, wpk1, c, relu=True, want_stats=True, in_affine=p_aff)
            _, mean1, invstd1, scale1, shift1 = _BNState(bn1).stats_tiles(st1, a, affine=(g1, b1))
        else:
            a = ops.conv3x3(p, wpk1, c, relu=True, in_affine=p_aff)
            if training:
                _, mean1, invstd1 = _BNState(bn1).stats(a, True)
                scale1, shift1 = ops.bn_scale_shift(g1, b1, mean1, invstd1)
            else:
                mean1, invstd1, scale1, shift1 = _BNState(bn1).eval_affine(g1, b1)
        if FUSE_AFFINE:
            # BN1's affine is applied while conv2 stages its input: bn1(a) is never written to HBM
            src, aff = a, (scale1, shift1)
        else:
            src, aff = ops.affine(a, scale1, shift1), None
        if FUSE_STATS:
            cc, st2 = ops.conv3x3(src, wpk2, c, in_affine=aff, want_stats=True)
            scale2 = None
            if training:
                ssum2, mean2, invstd2, scale2, shift2 = _BNState(bn2).stats_tiles(st2, cc, affine=(g2, b2))
            else:
                ssum2, _, _ = _BNState(bn2).stats_tiles(st2, cc, update=False)
                mean2, invstd2, scale2, shift2 = _BNState(bn2).eval_affine(g2, b2)
        else:
            scale2 = None
            cc = ops.conv3x3(src, wpk2, c, in_affine=aff)
            if training:
                ssum2, mean2, invstd2 = _BNState(bn2).stats(cc, True)
            else:
                ssum2, _, _ = ops.bn_stats(cc, None, None)
                mean2, invstd2, scale2, shift2 = _BNState(bn2).eval_affine(g2, b2)
        if scale2 is None:
            scale2, shift2 = ops.bn_scale_shift(g2, b2, mean2, invstd2)
        pooled, hid, s = ops.se_fc_fwd(ssum2, scale2, shift2, fw1, fb1, fw2, fb2, h * w_)
        q = None
        meand = invstdd = None
        if wd is not None:
            q = ops.gemm(p, wd, n * h * w_, c, cin, cin, cin).view(n, h, w_, c)
            if training:
                _, meand, invstdd = _BNState(bnd).stats(q, True)
                scaled, shiftd = ops.bn_scale_shift(gd, bd, meand, invstdd)
            else:
                meand, invstdd, scaled, shiftd = _BNState(bnd).eval_affine(gd, bd)
            r, raff = q, (scaled, shiftd)            # the downsample BatchNorm is applied while the tail reads q
        else:
            r, raff = p, p_aff
        # FUSE_POOL: in front of a pooled stage boundary the tail writes avgpool2(e) and the ReLU-mask bits of e; e itself -- read
        # only by that pooling in the forward pass, through its bits in the backward pass -- never goes to HBM (round 6)
        pool_out = FUSE_POOL and pool_next and link_out is not None and ops.se_tail_pool_ok(h, w_, c) and \
            (FUSE_MASKBITS or not training)
        if pool_out:
            e, ebits = ops.se_tail_fwd(cc, r, scale2, shift2, s, want_mask=training, r_affine=raff, pool_hw=(h, w_))
            link_out.prepooled = True
        elif training and FUSE_MASKBITS:
            e, ebits = ops.se_tail_fwd(cc, r, scale2, shift2, s, want_mask=True, r_affine=raff)
        else:
            e, ebits = ops.se_tail_fwd(cc, r, scale2, shift2, s, r_affine=raff), None
        ctx.link_in = link_in if (FUSE_SEBWD and FUSE_DR and training and link_in is not None and not pool
                                  and link_in.cc is not None) else None
        ctx.link_out = link_out if (FUSE_SEBWD and training) else None
        if ctx.link_out is not None:
            link_out.cc, link_out.mean2, link_out.invstd2, link_out.tiles = cc, mean2, invstd2, None
            link_out.ebits = ebits
        ctx.training, ctx.pool, ctx.has_down = training, pool and not prepooled, wd is not None
        ctx.pool_out = (h, w_) if pool_out else None
        ctx.in_hw = (x.shape[1], x.shape[2])
        ctx.fused_affine = aff is not None
        ctx.a_unfused = None if aff is not None else a      # (A/B switch only; keeps `a` alive for BN1's backward)
        ctx.has_bits = ebits is not None
        ctx.p_aff = p_aff
        # storage addresses of the parameters whose gradients can be written straight into the flat buffer (GradSink)
        ctx.ptrs = (w1.data_ptr(), g1.data_ptr(), b1.data_ptr(), w2.data_ptr(),
                    (fb2.data_ptr(), fw2.data_ptr(), fb1.data_ptr(), fw1.data_ptr(), b2.data_ptr(), g2.data_ptr()))
        ctx.wshapes = (tuple(w1.shape), tuple(w2.shape))
        # ... and of the shortcut projection (1x1 convolution + BatchNorm; round 6: three AccumulateGrad adds fewer per such block)
        ctx.down_ptrs = (wd.data_ptr(), gd.data_ptr(), bd.data_ptr(), tuple(wd.shape)) if wd is not None else None
        if isinstance(wpk1d, ops.DualPack):
            # the operand combination backward() will launch conv1's data-gradient with (see there): projection shortcut -> addend
            # (+ statistics against the block above's BatchNorm input with its ReLU-mask bits); identity -> addend + mask bits +
            # statistics / mask bits of the block above; the very first block -> addend + mask bits + statistics against the stem's
            # BatchNorm input (combination 15, persistent since round 6); the other combinations (the unfused ones) have no
            # persistent F(4x4) form
            lk = ctx.link_in
            if wd is not None:
                p_ok = lk is None or lk.ebits is not None
            elif lk is not None:
                p_ok = ebits is not None and lk.ebits is not None
            else:
                stem15 = FUSE_DR and ebits is not None and ctx.stem_holder is not None and \
                    getattr(ctx.stem_holder, "stem_bn", None) is not None
                p_ok = (not FUSE_DR) or stem15
            wpk1d = wpk1d.pick(n, h, w_, cin, True, p_ok)
        tensors = [p, src, scale1, cc, None if pool_out else e, g1, mean1, invstd1, g2, b2, mean2, invstd2, ssum2, pooled, hid, s, fw1, fw2,
                   wpk1d, wpk2d, shift1]
        names = ["p", "a", "scale1", "cc", "e", "g1", "mean1", "invstd1", "g2", "b2", "mean2", "invstd2", "ssum2", "pooled",
                 "hid", "s", "fw1", "fw2", "wpk1d", "wpk2d", "shift1"]
        if ebits is not None:
            tensors.append(ebits)
            names.append("ebits")
        if wd is not None:
            tensors += [q, wd, gd, meand, invstdd]
            names += ["q", "wd", "gd", "meand", "invstdd"]
        ctx.saved_names = names                 # (saved(grad_fn) below: tests address saved tensors by NAME, not by position)
        ctx.save_for_backward(*tensors)
        return e

    @staticmethod
    def backward(ctx, de):
        if not ctx.training:
            raise NotImplementedError("backward through eval-mode BatchNorm is not part of the hot path")
        t = ctx.saved_tensors
        (p, src, scale1, cc, e, g1, mean1, invstd1, g2, b2, mean2, invstd2, ssum2, pooled, hid, s, fw1, fw2, wpk1d,
         wpk2d, shift1) = t[:21]
        n, h, w_, cin = p.shape
        c = cc.shape[-1]
        de = _c(de)
        ebits_early = t[21] if ctx.has_bits else None
        pooled_hw = None
        if ctx.pool_out is not None and ebits_early is not None and FUSE_POOL_BWD:
            # our output was avgpool2(e): both passes of the tail's backward spread the pooled gradient on the fly, and the
            # apply pass writes the gradient of e itself where the identity shortcut needs it (conv1's data-gradient addend)
            pooled_hw = ctx.pool_out
        elif ctx.pool_out is not None:
            de = ops.avgpool2_bwd(de, ctx.pool_out[0], ctx.pool_out[1])
        tiles = None
        if ctx.link_out is not None and ctx.link_out.tiles is not None:
            tiles, ctx.link_out.tiles = ctx.link_out.tiles, None      # left by the block above (its dgrad produced `de`)
        nb = 21
        ebits = None
        if ctx.has_bits:
            ebits, nb = t[21], 22
        pw1, pg1, pb1, pw2, pse = ctx.ptrs
        vse = SINK.span(pse)                  # [fc.2.bias | fc.2.weight | fc.0.bias | fc.0.weight | bn2.bias | bn2.weight] or None
        vw1, vw2 = SINK.view(pw1, ctx.wshapes[0]), SINK.view(pw2, ctx.wshapes[1])
        vg1, vb1 = SINK.view(pg1), SINK.view(pb1)
        sunk = vse is not None and vw1 is not None and vw2 is not None and vg1 is not None and vb1 is not None
        if not sunk:
            vse = vw1 = vw2 = vg1 = vb1 = None
        de_full = None
        if pooled_hw is not None and not ctx.has_down and FUSE_DR:
            de_full = torch.empty(n, pooled_hw[0], pooled_hw[1], c, dtype=torch.float32, device=de.device)
        dc, dr, dg2, db2, dfw1, dfb1, dfw2, dfb2 = ops.se_tail_bwd(de, e, cc, g2, b2, mean2, invstd2, ssum2, pooled,
                                                                   hid, s, fw1, fw2,
                                                                   want_dr=ctx.has_down or not FUSE_DR, tile_stats=tiles,
                                                                   mask=ebits, packed_out=vse, pooled_hw=pooled_hw,
                                                                   de_out=de_full)
        if pooled_hw is not None:
            de = de_full                          # (None when nothing below reads it)
        if ctx.fused_affine:
            a = src
            dw2 = ops.conv3x3_wgrad(a, dc, c, in_affine=(scale1, shift1), out=vw2)
        else:
            dw2 = ops.conv3x3_wgrad(src, dc, c, out=vw2)
            a = ctx.a_unfused
        if FUSE_BNBWD:
            dbb, st = ops.conv3x3(dc, wpk2d, c, want_stats=True, stat_bn=(a, mean1, invstd1))
            da, dg1, db1 = ops.bn_bwd(dbb, a, g1, mean1, invstd1, relu_mask=True, tile_stats=st, out_dgamma=vg1,
                                      out_dbeta=vb1)
        else:
            dbb = ops.conv3x3(dc, wpk2d, c)
            da, dg1, db1 = ops.bn_bwd(dbb, a, g1, mean1, invstd1, relu_mask=True, out_dgamma=vg1, out_dbeta=vb1)
        dw1 = ops.conv3x3_wgrad(p, da, cin, in_affine=ctx.p_aff, out=vw1)
        if sunk:
            SINK.done(*pse)
            SINK.done(pw2, pg1, pb1, pw1)
            dw1 = dg1 = db1 = dw2 = dg2 = db2 = dfw1 = dfb1 = dfw2 = dfb2 = None
        dwd = dgd = dbd = None
        if ctx.has_down:
            q, wd, gd, meand, invstdd = t[nb:nb + 5]
            pwd, pgd, pbd, wdshape = ctx.down_ptrs
            vwd, vgd, vbd = SINK.view(pwd, (wdshape[0], wdshape[1])), SINK.view(pgd), SINK.view(pbd)
            dsunk = sunk and vwd is not None and vgd is not None and vbd is not None
            if not dsunk:
                vwd = vgd = vbd = None
            dq, dgd, dbd = ops.bn_bwd(dr, q, gd, meand, invstdd, relu_mask=False, out_dgamma=vgd, out_dbeta=vbd)
            rows = n * h * w_
            dwd = ops.gemm(dq, p, c, cin, rows, c, cin, trans_a=True, trans_b=True,
                           splits=ops.wgrad_splits(c, cin, rows), out=vwd).view(c, cin, 1, 1)
            if dsunk:
                SINK.done(pwd, pgd, pbd)
                dwd = dgd = dbd = None
            dp_res = ops.gemm(dq, wd, rows, cin, c, c, cin, trans_b=True).view(n, h, w_, cin)
            if ctx.link_in is not None:       # un-pooled stage boundary (stage 4): dp is the gradient of the block above
                lk = ctx.link_in
                dp, lk.tiles = ops.conv3x3(da, wpk1d, cin, addend=dp_res, want_stats=True,
                                           stat_bn=(lk.cc, lk.mean2, lk.invstd2),
                                           stat_mask=lk.ebits if lk.ebits is not None else p)
            else:
                dp = ops.conv3x3(da, wpk1d, cin, addend=dp_res)
        else:
            # identity shortcut: its gradient de * (e > 0) is formed inside the dgrad epilogue
            emask = ebits if ebits is not None else e          # this block's ReLU mask: bits when the forward stored them
            if ctx.link_in is not None:
                lk = ctx.link_in
                dp, lk.tiles = ops.conv3x3(da, wpk1d, cin, addend=de, addend_mask=emask, want_stats=True,
                                           stat_bn=(lk.cc, lk.mean2, lk.invstd2),
                                           stat_mask=lk.ebits if lk.ebits is not None else p)
            elif FUSE_DR and ctx.stem_holder is not None and getattr(ctx.stem_holder, "stem_bn", None) is not None:
                # first block: dp is the gradient w.r.t. the stem's BatchNorm output -- sum dp and dp * xhat(a_stem) per patch
                # here, so the stem's backward skips its 2.5 GB reduction pass
                dp, ctx.stem_holder.tiles = ops.conv3x3(da, wpk1d, cin, addend=de, addend_mask=emask, want_stats=True,
                                                        stat_bn=ctx.stem_holder.stem_bn)
            elif FUSE_DR:
                dp = ops.conv3x3(da, wpk1d, cin, addend=de, addend_mask=emask)
            else:
                dp = ops.conv3x3(da, wpk1d, cin, addend=dr)
        dx = ops.avgpool2_bwd(dp, ctx.in_hw[0], ctx.in_hw[1]) if ctx.pool else dp
        return (dx, None, None, None, dw1, dg1, db1, dw2, dg2, db2, dfw1, dfb1, dfw2, dfb2, dwd, dgd, dbd)


class SAPFn(torch.autograd.Function):
    """SelfAttentionPooling over the F axis (reference resnet.py:115-123): x [B][T][F][C] -> [B][T][C]."""

    @staticmethod
    def forward(ctx, x, w, b):
        bsz, t, f, c = x.shape
        y, attn = ops.sap_fwd(x.view(bsz * t, f, c), w.view(-1), b)
        ctx.bias = b                           # (only its storage address and shape are used: GradSink)
        ctx.save_for_backward(x, w, attn)
        return y.view(bsz, t, c)

    @staticmethod
    def backward(ctx, dy):
        x, w, attn = ctx.saved_tensors
        bsz, t, f, c = x.shape
        # (GradSink: the two accumulators the kernel sums into ARE the parameters' slices of the flat gradient buffer, zero since
        #  zero_grad -- no zero-fill launches, no AccumulateGrad adds; round 6)
        sunk = SINK.params(w, ctx.bias) if (ctx.needs_input_grad[1] and ctx.needs_input_grad[2]) else None
        dx, dw, db = ops.sap_bwd(_c(dy).view(bsz * t, c), x.view(bsz * t, f, c), w.view(-1), attn,
                                 out_dw=sunk[0].view(-1) if sunk is not None else None, out_db=sunk[1] if sunk is not None else None)
        if sunk is not None:
            SINK.done_params(w, ctx.bias)
            return dx.view(bsz, t, f, c), None, None
        return dx.view(bsz, t, f, c), dw.view_as(w), db


class BiGRULayerFn(torch.autograd.Function):
    """One bidirectional GRU layer, hidden 128 (half of nn.GRU(num_layers=2) at reference resnet.py:153)."""

    @staticmethod
    def forward(ctx, x, wih_f, whh_f, bih_f, bhh_f, wih_r, whh_r, bih_r, bhh_r, save):
        bsz, t, cin = x.shape
        rows = bsz * t
        x2 = x.view(rows, cin)
        gx = torch.empty(rows, 768, dtype=torch.float32, device=x.device)
        ops.gemm(x2, wih_f, rows, 384, cin, cin, cin, bias=bih_f, out=gx, ldc=768)
        ops.gemm(x2, wih_r, rows, 384, cin, cin, cin, bias=bih_r, out=gx[:, 384:], ldc=768)
        whh = torch.stack([whh_f, whh_r], 0).contiguous()
        bhh = torch.stack([bhh_f, bhh_r], 0).contiguous()
        out, gates, hprev = ops.gru_fwd(gx.view(bsz, t, 2, 384), whh, bhh, save)
        if save:
            ctx.save_for_backward(x, wih_f, wih_r, whh, gates, hprev)
        ctx.saved = save
        ctx.params = (wih_f, whh_f, bih_f, bhh_f, wih_r, whh_r, bih_r, bhh_r)     # (storage addresses and shapes: GradSink)
        return out

    @staticmethod
    def backward(ctx, dout):
        if not ctx.saved:
            raise NotImplementedError("GRU backward needs the training-mode forward (gates were not saved)")
        x, wih_f, wih_r, whh, gates, hprev = ctx.saved_tensors
        bsz, t, cin = x.shape
        rows = bsz * t
        x2 = x.view(rows, cin)
        dgx, dgh = ops.gru_bwd(_c(dout), gates, hprev, whh)
        dgx2, dgh2, hp2 = dgx.view(rows, 768), dgh.view(rows, 768), hprev.view(rows, 256)
        splits = ops.wgrad_splits(384, 128, rows)
        grads = []
        dx = torch.empty_like(x2)
        # GradSink: the eight weight / bias gradients of the layer go straight into their slices of the flat gradient buffer
        # (16 AccumulateGrad adds fewer per step over the two layers; round 6)
        sunk = SINK.params(*ctx.params) if all(ctx.needs_input_grad[1:9]) else None
        for d, wih in enumerate((wih_f, wih_r)):
            gxd, ghd, hpd = dgx2[:, d * 384:], dgh2[:, d * 384:], hp2[:, d * 128:]
            o = sunk[4 * d:4 * d + 4] if sunk is not None else (None, None, None, None)
            dwih = ops.gemm(gxd, x2, 384, cin, rows, 768, cin, trans_a=True, trans_b=True, splits=splits, out=o[0])
            dwhh = ops.gemm(ghd, hpd, 384, 128, rows, 768, 256, trans_a=True, trans_b=True, splits=splits, out=o[1])
            dbih = ops.colsum(gxd[:, :384], out=o[2])
            dbhh = ops.colsum(ghd[:, :384], out=o[3])
            ops.gemm(gxd, wih, rows, cin, 384, 768, cin, trans_b=True, out=dx, ldc=cin, accumulate=(d == 1))
            grads += [dwih, dwhh, dbih, dbhh]
        if sunk is not None:
            SINK.done_params(*ctx.params)
            grads = [None] * 8
        return (dx.view(bsz, t, cin), *grads, None)


class LNTanhFn(torch.autograd.Function):
    """tanh(LayerNorm(x))  (reference resnet.py:196-197)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        c = x.shape[-1]
        y = ops.ln_tanh_fwd(x.view(-1, c), gamma, beta, eps)
        ctx.eps = eps
        ctx.beta = beta                        # (only its storage address and shape are used: GradSink)
        ctx.save_for_backward(x, y, gamma)
        return y.view_as(x)

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma = ctx.saved_tensors
        c = x.shape[-1]
        sunk = SINK.params(gamma, ctx.beta) if (ctx.needs_input_grad[1] and ctx.needs_input_grad[2]) else None
        dx, dgamma, dbeta = ops.ln_tanh_bwd(_c(dy).view(-1, c), x.view(-1, c), y, gamma, ctx.eps,
                                            out_dgamma=sunk[0] if sunk is not None else None,
                                            out_dbeta=sunk[1] if sunk is not None else None)
        if sunk is not None:
            SINK.done_params(gamma, ctx.beta)
            return dx.view_as(x), None, None, None
        return dx.view_as(x), dgamma, dbeta, None


class LinearFn(torch.autograd.Function):
    """nn.Linear on the last axis (reference linearheads.py:95-98)."""

    @staticmethod
    def forward(ctx, x, w, b):
        k = x.shape[-1]
        x = _c(x)
        y = ops.linear(x.view(-1, k), w, b)
        ctx.bias = b                           # (only its storage address and shape are used: GradSink)
        ctx.save_for_backward(x, w)
        return y.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        k = x.shape[-1]
        b = ctx.bias
        sunk = SINK.params(w, b) if (ctx.needs_input_grad[1] and (b is None or ctx.needs_input_grad[2])) else None
        if sunk is not None:          # weight / bias gradients written straight into the flat gradient buffer
            dx, _, _ = ops.linear_bwd(x.view(-1, k), w, _c(dy).view(-1, w.shape[0]), need_dx=ctx.needs_input_grad[0],
                                      out_dw=sunk[0], out_db=sunk[1], need_db=b is not None)
            SINK.done_params(w, b)
            return (dx.view_as(x) if dx is not None else None), None, None
        dx, dw, db = ops.linear_bwd(x.view(-1, k), w, _c(dy).view(-1, w.shape[0]), need_dx=ctx.needs_input_grad[0])
        return (dx.view_as(x) if dx is not None else None), dw, (db if ctx.needs_input_grad[2] else None)


class DropoutFn(torch.autograd.Function):
    """Inter-layer GRU dropout (p = 0.3, training only): y = x * mask, mask in {0, 1/(1-p)}."""

    @staticmethod
    def forward(ctx, x, mask):
        ctx.save_for_backward(mask)
        return ops.mul(x, mask)

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        return ops.mul(_c(dy), mask), None


class DropoutHashFn(torch.autograd.Function):
    """Dropout whose mask is regenerated from (seed, offset) of a ``DropoutStream`` in the forward and in the backward
    kernel -- the same mask values ``DropoutFn`` would read from a tensor, without the tensor."""

    @staticmethod
    def forward(ctx, x, p, seed, offset, offset_dev=None):
        ctx.key = (p, seed, offset, offset_dev)
        return ops.dropout_apply(_c(x), p, seed, offset, offset_dev)

    @staticmethod
    def backward(ctx, dy):
        p, seed, offset, offset_dev = ctx.key
        return ops.dropout_apply(_c(dy), p, seed, offset, offset_dev), None, None, None, None


class ADYOLOLossFn(torch.autograd.Function):
    """AD-YOLO loss (reference loss.py:189-251); the gradient w.r.t. the logits is produced by the same
    launch that computes the loss and only rescaled by the incoming gradient in backward."""

    @staticmethod
    def forward(ctx, logit, target, cfg):
        loss, dlogit, _ = ops.adyolo_loss(logit, target, cfg["nb_classes"], cfg["grid"], cfg["anchors"], cfg["thr"],
                                          cfg["gains"], cfg["grid_size"], cfg["g_overlap"],
                                          need_grad=ctx.needs_input_grad[0])
        if dlogit is not None:
            ctx.save_for_backward(dlogit)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        (dlogit,) = ctx.saved_tensors
        return ops.scale_dev(dlogit, _c(dloss).view(1)), None, None


class ActFn(torch.autograd.Function):
    """Head activation: sigmoid on the first ``n_sig`` columns, tanh on the rest (reference linearheads.py:44-47,65,83)."""

    @staticmethod
    def forward(ctx, x, n_sig):
        k = x.shape[-1]
        y = ops.act_fwd(_c(x).view(-1, k), n_sig)
        ctx.n_sig = n_sig
        ctx.save_for_backward(y)
        return y.view_as(x)

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return ops.act_bwd(_c(dy).view(-1, y.shape[-1]), y, ctx.n_sig).view_as(dy), None


class _FusedLossFn(torch.autograd.Function):
    """Losses whose gradient w.r.t. the network output is produced by the same launch as the value."""

    @staticmethod
    def forward(ctx, output, target, kind, cfg):
        k = output.shape[-1]
        need = ctx.needs_input_grad[0]
        out2d = _c(output).view(-1, k)
        if kind == "adpit":
            loss, dout = ops.adpit_loss(out2d, _c(target), cfg["nb_classes"], need)
        else:
            loss, dout = ops.seddoa_loss(out2d, _c(target).view(-1, k), cfg["nsed"], cfg["masked"], cfg["w_bce"],
                                         cfg["w_mse"], need)
        if dout is not None:
            ctx.save_for_backward(dout)
        ctx.shape = output.shape
        return loss.view(())

    @staticmethod
    def backward(ctx, dloss):
        (dout,) = ctx.saved_tensors
        return ops.scale_dev(dout, _c(dloss).view(1)).view(ctx.shape), None, None, None


# ============================================================================================== conformer nodes
class ConvFn(torch.autograd.Function):
    """General strided convolution (channels-last) as an implicit GEMM on the fp32 MFMA: the column matrix is never built,
    the GEMM gathers its operand from the activation tensor while staging it (``adyolo_conv_gemm``).
    x [N][H][W][Cin], w [Cout][Cin][KH][KW]  (reference resnet_conformer.py:347 7x7 s(1,2); torchvision BasicBlock
    3x3 / 1x1 s(1,2) at :353-393)."""

    @staticmethod
    def forward(ctx, x, w, stride, padding):
        n, h, ww, cin = x.shape
        cout, _, kh, kw = w.shape
        ctx.geom = (n, h, ww, cin, cout, kh, kw, stride[0], stride[1], padding[0], padding[1])
        ctx.save_for_backward(x, w)
        return ops.conv_gemm(0, x, ops.pack_wk(w), *ctx.geom)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        n, h, ww, cin, cout, kh, kw = ctx.geom[:7]
        dy = _c(dy)
        sunk = SINK.params(w)
        dw = ops.unpack_wk(ops.conv_gemm(2, x, dy, *ctx.geom), cout, cin, kh, kw, out=sunk[0] if sunk is not None else None)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.conv_gemm(1, dy, ops.pack_wk(w.transpose(0, 1).contiguous()), *ctx.geom)
        if sunk is not None:
            SINK.done_params(w)
            dw = None
        return dx, dw, None, None


class Conv3x1WinoFn(torch.autograd.Function):
    """3 x 1 convolution along H (stride 1, padding 1) on a map ONE bin wide -- x [N][H][1][Cin], w [Cout][Cin][3][1] -- as a 1-D
    Winograd F(4, 3): half the matrix FLOPs of the implicit-GEMM form (``csrc/wino1d.hip``; the deep stages of the
    ResNet-Conformer, reference resnet_conformer.py:353-393).  Keeps the transformed input V for the weight gradient."""

    @staticmethod
    def forward(ctx, x, w):
        n, h, _, cin = x.shape
        cout = w.shape[0]
        w3 = _c(w).view(cout, cin, 3)
        y, v = ops.wino1d_conv(_c(x).view(n, h, cin), ops.wino1d_filter(w3, cout, cin, 0), n, h, cin, cout)
        ctx.save_for_backward(v, w3)
        ctx.geom = (n, h, cin, cout)
        ctx.w = w                              # (storage address / shape for GradSink)
        return y.view(n, h, 1, cout)

    @staticmethod
    def backward(ctx, dy):
        v, w3 = ctx.saved_tensors
        n, h, cin, cout = ctx.geom
        dy3 = _c(dy).view(n, h, cout)
        sunk = SINK.params(ctx.w)
        dw = ops.wino1d_wgrad(v, dy3, n, h, cin, cout, out=sunk[0].view(cout, cin, 3) if sunk is not None else None)
        dx = None
        if ctx.needs_input_grad[0]:
            dx, _ = ops.wino1d_conv(dy3, ops.wino1d_filter(w3, cout, cin, 1), n, h, cout, cin)
            dx = dx.view(n, h, 1, cin)
        if sunk is not None:
            SINK.done_params(ctx.w)
            dw = None
        else:
            dw = dw.view(cout, cin, 3, 1)
        return dx, dw


class Conv3x3NarrowFn(torch.autograd.Function):
    """Stride-1 3x3 convolution on a map 3 or 4 bins wide (the ResNet-Conformer's 128-channel stage, reference
    resnet_conformer.py:353-393): forward and data gradient on the persistent F(4x4) kernel with patches ONE tile wide (128 x 4
    pixels: no padded tiles, a quarter of the implicit GEMM's matrix work), weight gradient in the F(4x4) domain with runs made
    of four samples side by side (``csrc/wino4w.hip``), or on the implicit GEMM below its dispatch threshold."""

    @staticmethod
    def forward(ctx, x, w):
        n, h, ww, cin = x.shape
        cout = w.shape[0]
        wpk, wpkd = ops.pack_w3x3(w, cin)
        ctx.geom = (n, h, ww, cin, cout, 3, 3, 1, 1, 1, 1)
        ctx.save_for_backward(x, wpkd)
        ctx.w = w                              # (storage address / shape for GradSink)
        return ops.conv3x3(x, wpk, cout)

    @staticmethod
    def backward(ctx, dy):
        x, wpkd = ctx.saved_tensors
        n, h, ww, cin, cout = ctx.geom[:5]
        dy = _c(dy)
        sunk = SINK.params(ctx.w)
        if ops.wgrad_form(cin, cout, None, (n, h, ww))[0] == "wino4_wgrad_kernel":
            # (round 5: the F(4x4)-domain kernel builds its 4-tile runs from 4 / (W / 4) samples side by side on these maps)
            dw = ops.conv3x3_wgrad(x, dy, cin, out=sunk[0] if sunk is not None else None)
        else:
            dw = ops.unpack_wk(ops.conv_gemm(2, x, dy, *ctx.geom), cout, cin, 3, 3, out=sunk[0] if sunk is not None else None)
        dx = ops.conv3x3(dy, wpkd, cin) if ctx.needs_input_grad[0] else None
        if sunk is not None:
            SINK.done_params(ctx.w)
            dw = None
        return dx, dw


class Conv3x3S1Fn(torch.autograd.Function):
    """Stride-1 3x3 convolution on the implicit-GEMM kernel (K2), no fusion."""

    @staticmethod
    def forward(ctx, x, w):
        cin = x.shape[-1]
        wpk, wpkd = ops.pack_w3x3(w, cin)
        ctx.save_for_backward(x, wpkd)
        ctx.cin = w.shape[1]
        ctx.w = w                              # (storage address / shape for GradSink)
        return ops.conv3x3(x, wpk, w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x, wpkd = ctx.saved_tensors
        dy = _c(dy)
        sunk = SINK.params(ctx.w)
        dw = ops.conv3x3_wgrad(x, dy, ctx.cin, out=sunk[0] if sunk is not None else None)
        dx = ops.conv3x3(dy, wpkd, x.shape[-1]) if ctx.needs_input_grad[0] else None
        if sunk is not None:
            SINK.done_params(ctx.w)
            dw = None
        return dx, dw


class ReluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        c = x.shape[-1]
        y = ops.affine_relu(x, _const(1.0, (c,), x.device), _const(0.0, (c,), x.device))
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return ops.relu_bwd(_c(dy), y)


class BatchNormFn(torch.autograd.Function):
    """BatchNorm over the last (channel) axis of a channels-last tensor [N][...][C], optional fused ReLU on the output,
    optional residual: y = act(bn(x) + r)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, bn, training, relu, residual=None):
        if training:
            _, mean, invstd = _BNState(bn).stats(x, True)
            scale, shift = ops.bn_scale_shift(gamma, beta, mean, invstd)
        else:
            mean, invstd, scale, shift = _BNState(bn).eval_affine(gamma, beta)
        if residual is not None:
            n, c = x.shape[0], x.shape[-1]
            ones = _const(1.0, (n, c), x.device)
            y = ops.se_tail_fwd(x, residual, scale, shift, ones)            # relu(bn(x) * 1 + r)
        elif relu:
            y = ops.affine_relu(x, scale, shift)
        else:
            y = ops.affine(x, scale, shift)
        ctx.training, ctx.relu, ctx.has_res = training, relu or residual is not None, residual is not None
        ctx.beta = beta                        # (storage address / shape for GradSink)
        ctx.save_for_backward(x, y, gamma, mean, invstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        if not ctx.training:
            raise NotImplementedError("backward through eval-mode BatchNorm is not part of the hot path")
        x, y, gamma, mean, invstd = ctx.saved_tensors
        g = ops.relu_bwd(_c(dy), y) if ctx.relu else _c(dy)
        sunk = SINK.params(gamma, ctx.beta)
        if sunk is not None:
            dx, _, _ = ops.bn_bwd(g, x, gamma, mean, invstd, relu_mask=False, out_dgamma=sunk[0], out_dbeta=sunk[1])
            SINK.done_params(gamma, ctx.beta)
            return dx, None, None, None, None, None, (g if ctx.has_res else None)
        dx, dgamma, dbeta = ops.bn_bwd(g, x, gamma, mean, invstd, relu_mask=False)
        return dx, dgamma, dbeta, None, None, None, (g if ctx.has_res else None)


class MaxPool3Fn(torch.autograd.Function):
    """MaxPool2d(3, stride (1,2), padding 1) (reference resnet_conformer.py:350)."""

    @staticmethod
    def forward(ctx, x):
        y, arg = ops.maxpool3_fwd(x)
        ctx.w = x.shape[2]
        ctx.save_for_backward(arg)
        return y

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        return ops.maxpool3_bwd(_c(dy), arg, ctx.w)


class LNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        c = x.shape[-1]
        y = ops.ln_fwd(_c(x).view(-1, c), gamma, beta, eps)
        ctx.eps = eps
        ctx.beta = beta                        # (storage address / shape for GradSink)
        ctx.save_for_backward(x, gamma)
        return y.view_as(x)

    @staticmethod
    def backward(ctx, dy):
        x, gamma = ctx.saved_tensors
        c = x.shape[-1]
        sunk = SINK.params(gamma, ctx.beta)
        if sunk is not None:                   # the kernel adds into the (zeroed) gradient slices
            dx, _, _ = ops.ln_bwd(_c(dy).view(-1, c), _c(x).view(-1, c), gamma, ctx.eps, acc_dgamma=sunk[0], acc_dbeta=sunk[1])
            SINK.done_params(gamma, ctx.beta)
            return dx.view_as(x), None, None, None
        dx, dg, db = ops.ln_bwd(_c(dy).view(-1, c), _c(x).view(-1, c), gamma, ctx.eps)
        return dx.view_as(x), dg, db, None


class SwishFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.swish_fwd(_c(x))

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.swish_bwd(_c(dy), _c(x))


class GLUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        c2 = x.shape[-1]
        ctx.save_for_backward(x)
        return ops.glu_fwd(_c(x).view(-1, c2)).view(*x.shape[:-1], c2 // 2)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        c2 = x.shape[-1]
        return ops.glu_bwd(_c(dy).view(-1, c2 // 2), _c(x).view(-1, c2)).view_as(x)


class DWConv3Fn(torch.autograd.Function):
    """Depthwise Conv1d(k=3, dilation d, padding d) over time, channels-last (reference resnet_conformer.py:169)."""

    @staticmethod
    def forward(ctx, x, w, bias, dilation):
        ctx.dilation = dilation
        ctx.bias = bias                        # (storage address / shape for GradSink)
        ctx.save_for_backward(x, w)
        return ops.dwconv3(_c(x), w.view(-1, 3), bias, dilation)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _c(dy)
        dx = ops.dwconv3(dy, w.view(-1, 3), None, ctx.dilation, flip=True)
        sunk = SINK.params(w, ctx.bias) if ctx.bias is not None else None
        if sunk is not None:
            ops.dwconv3_wgrad(dy, _c(x), ctx.dilation, out_dw=sunk[0].view(-1, 3), out_db=sunk[1])
            SINK.done_params(w, ctx.bias)
            return dx, None, None, None
        dw, db = ops.dwconv3_wgrad(dy, _c(x), ctx.dilation)
        return dx, dw.view_as(w), db, None


class AxpbyFn(torch.autograd.Function):
    """a * x + b * z  (ResidualConnectionModule, reference resnet_conformer.py:98)."""

    @staticmethod
    def forward(ctx, x, z, a, b):
        ctx.ab = (a, b)
        return ops.axpby(_c(x), _c(z), a, b)

    @staticmethod
    def backward(ctx, dy):
        a, b = ctx.ab
        dy = _c(dy)
        # a factor of exactly 1 hands the incoming gradient on (no copy launch: autograd does not write into it)
        return (dy if a == 1.0 else ops.axpby(dy, dy, a, 0.0)), (dy if b == 1.0 else ops.axpby(dy, dy, b, 0.0)), None, None


class DropoutAxpbyFn(torch.autograd.Function):
    """a * dropout(x) + b * z in ONE pass, mask regenerated from (seed, offset) like ``DropoutHashFn``: the residual mix of a
    Conformer sub-module that ends in nn.Dropout (reference resnet_conformer.py:98 over :178 / :209 / :272-274).  Same
    values as ``AxpbyFn(DropoutHashFn(x), z, a, b)``; backward: a * dropout-mask * dy and b * dy."""

    @staticmethod
    def forward(ctx, x, z, a, b, p, seed, offset, offset_dev=None):
        ctx.key = (a, b, p, seed, offset, offset_dev)
        return ops.dropout_axpby(_c(x), _c(z), a, b, p, seed, offset, offset_dev)

    @staticmethod
    def backward(ctx, dy):
        a, b, p, seed, offset, offset_dev = ctx.key
        dy = _c(dy)
        dx = ops.dropout_axpby(dy, None, a, 0.0, p, seed, offset, offset_dev)
        dz = dy if b == 1.0 else ops.axpby(dy, dy, b, 0.0)
        return dx, dz, None, None, None, None, None, None


class AvgPool1dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, fac):
        ctx.meta = (x.shape[1], k, fac)
        return ops.avgpool1d(_c(x), k, fac)

    @staticmethod
    def backward(ctx, dy):
        t, k, fac = ctx.meta
        return ops.avgpool1d_bwd(_c(dy), t, k, fac), None, None


class AttentionCoreFn(torch.autograd.Function):
    """dropout(softmax(scale * Q K^T), p) V per head; q, k, v: [B][T][heads*64] (reference resnet_conformer.py:57-85).
    Flash style on the fp32 matrix cores (csrc/attention.hip): the T x T scores never reach HBM; backward recomputes the
    probabilities from Q, K and the per-row log-sum-exp.  ``drop`` = (p, seed32) or None."""

    @staticmethod
    def forward(ctx, q, k, v, heads, scale, drop):
        q, k, v = _c(q), _c(k), _c(v)
        p, seed = drop if drop is not None else (0.0, 0)
        need = any(ctx.needs_input_grad[:3])
        out, lse = ops.attn_fwd(q, k, v, heads, scale, p, seed, want_lse=need)
        ctx.meta = (heads, scale, p, seed)
        if need:
            ctx.save_for_backward(q, k, v, out, lse)
        return out

    @staticmethod
    def backward(ctx, dctx):
        q, k, v, out, lse = ctx.saved_tensors
        heads, scale, p, seed = ctx.meta
        dq, dk, dv = ops.attn_bwd(q, k, v, out, _c(dctx), lse, heads, scale, p, seed)
        return dq, dk, dv, None, None, None
