"""Tensor-level wrappers over the C ABI (include/adyolo_hip.h).

PyTorch is used here only as plumbing: device memory (caching allocator), the current HIP stream and,
in ``dist.py``, torch.distributed/RCCL.  Every function launches hand-written gfx950 kernels through
``libadyolo_hip.so``; nothing in this module computes with ATen ops.
"""
import ctypes
import os

import torch

from . import _lib

_c = _lib.call
NULL = ctypes.c_void_p(0)


def _p(t):
    if t is None:
        return NULL
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.AdyoloHipError("adyolo ops need tensors on a HIP device (got %s); there is no CPU path" % t.device)
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise _lib.AdyoloHipError("adyolo ops need contiguous float32 tensors (got %s, contiguous=%s)"
                                      % (t.dtype, t.is_contiguous()))


def _new(like, *shape):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


def _zeros(like, *shape):
    return torch.zeros(shape, dtype=torch.float32, device=like.device)


# ---------------------------------------------------------------------------------------------- exact data parallelism
class ExactDP:
    """Switch + collectives of EXACT data parallelism (``train.TrainStep(exact=True)`` / ADYOLO_DP_EXACT=1): every quantity
    the reference normalises over the batch is formed over the batch of ALL ranks, so N ranks on N equal shards compute what
    one device computes on the concatenated batch (SURVEY.md section 8e "optional"; the reference itself is single-device,
    src/train.py:40-62).  Three places need an exchange, all tiny:
      * BatchNorm forward (36 layers): the per-sample sums (sum, sum of squares) [B][C] of every rank are all-gathered in rank
        order and finished by the same kernel as on one device -> bit-identical batch statistics and running statistics;
      * BatchNorm backward (36 layers, incl. the BatchNorm folded into the SE tail): the two per-channel sums are all-reduced
        for the dx formula; the parameter gradients written to the flat buffer stay local (the bucket all-reduce sums them);
      * the AD-YOLO loss: the four counts (distinct positives per threshold, responsible pairs) are all-reduced between the
        assignment and the pass over the logits; the returned loss is the sum of the ranks' shares.
    Gradients are then SUMMED over the ranks (not averaged).  Off (world 1, or not enabled): everything is local, the
    DDP-conventional semantics of ``dist.py``."""

    def __init__(self):
        self.on, self.world, self.group = False, 1, None

    def enable(self, group=None):
        import torch.distributed as dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.on = self.world > 1
        return self.on

    def disable(self):
        self.on = False

    def all_reduce(self, t):
        import torch.distributed as dist
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def gather_rows(self, t):
        """[n][c] of every rank -> [world * n][c] in rank order (== the row order of the concatenated batch)."""
        import torch.distributed as dist
        parts = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(parts, t.contiguous(), group=self.group)
        return torch.cat(parts, 0)


EXACT = ExactDP()


def EXACT_world_mean(t, group=None):
    """``t`` averaged over the ranks (in place; a no-op without a process group)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        t /= dist.get_world_size(group)
    return t


# Device -> host through page-locked staging buffers (one per shape and dtype, reused): a pageable ``tensor.cpu()`` of the decoded
# predictions (5.8 MB per 60 s clip) runs at a fifth of the PCIe rate and was a third of an eight-clip evaluation pass.
_PINNED = {}


def to_host(t):
    """``t`` copied into a reused page-locked host tensor of its shape (valid until the next ``to_host`` of that shape)."""
    key = (tuple(t.shape), t.dtype)
    buf = _PINNED.get(key)
    if buf is None:
        if len(_PINNED) >= 16:                       # clips of many different lengths: keep the page-locked pool bounded
            _PINNED.pop(next(iter(_PINNED)))
        buf = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        _PINNED[key] = buf
    buf.copy_(t, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return buf


# Parameter / buffer epoch: bumped by everything that writes parameters or BatchNorm buffers IN PLACE through a kernel (no
# torch version bump): the optimizer step, the training-mode BatchNorm statistics, the start-up broadcast.  Evaluation-mode
# caches (BatchNorm affines, ``functional._BNState.eval_affine``; recorded forward graphs, ``graph.ForwardGraphs``) are keyed on it.
PARAMS_EPOCH = [0]


def params_changed():
    PARAMS_EPOCH[0] += 1


# ---------------------------------------------------------------------------------------------- conv
def conv_algo():
    """'winograd4' (default since round 4: F(4x4,3x3) on the fp32 MFMA where ``_w4_eligible`` says so -- 4x fewer matrix FLOPs
    than the direct form -- and F(2x2,3x3) elsewhere), 'winograd' (F(2x2,3x3) everywhere, 2.25x fewer) or 'direct' (implicit
    GEMM); chosen per call of ``pack_w3x3`` / ``WinoPackSet.refresh`` from ADYOLO_CONV_ALGO."""
    a = os.environ.get("ADYOLO_CONV_ALGO", "winograd4").lower()
    if a not in ("winograd4", "winograd", "direct"):
        raise _lib.AdyoloHipError("ADYOLO_CONV_ALGO must be 'winograd4', 'winograd' or 'direct' (got %r)" % a)
    return a


# Dispatch switches of the 3x3 convolutions, read from the environment ONCE -- at import, and again by ``reload_thresholds()``
# (tests and A/B tools that move one call it; README "Switches") -- into ONE table that every launch consults: the numeric
# thresholds AND the on / off switches (round 5 ADVICE: ADYOLO_W4_PERSIST / _NARROW / ADYOLO_WINO1D used to be read per launch,
# three getenv calls inside every adyolo_wino4_fwd, while the thresholds were frozen).  The library keeps its own copy of the two
# switches it consults (csrc/wino4.hip, read at its first launch), refreshed by the same call.  ADYOLO_CONV_ALGO stays a
# per-call read (``conv_algo``): it selects the algorithm, is validated where it is read and is part of the graph stamp on its own.
W4_THRESHOLDS = {}
_THRESHOLD_KEYS = ("min_k", "min_k_addend", "min_wgs", "min_wgrad_work", "min_k_32")


def reload_thresholds():
    """ADYOLO_W4_MIN_K (32; 64 until round 6 -- with the persistent kernel the 32 -> 64 stage transition runs 0.56 -> 0.42 ms on
    F(4x4)): smallest contraction that gets the F(4x4) form packed; ADYOLO_W4_MIN_K_ADDEND (32; 128 until the
    persistent kernel of round 5): the same for launches that add a tensor in their epilogue; ADYOLO_W4_MIN_WGS (200): below
    that many 64-channel x patch work items a launch stays on the F(2x2) kernel (two workgroups per CU); ADYOLO_W4W_MIN_WORK
    (6 M): smallest (tile rows of 16-column runs) x Cin x Cout = N W/16 H/4 Cin Cout for the F(4x4)-domain weight gradient
    (``wgrad_form``; measured crossover with the F(2x2)-domain kernel at 3.3-6.5 M at every stage since the kernel stages between
    its MFMA groups -- 16 x 20 s: 1.18-1.41 x at every block shape, 8 x 20 s: 0.8-0.9 x at the 3.3 M stage transitions, 1.0-1.3 x
    from 6.5 M on: profiles/r05_w4w_small_shapes.txt; 2^24 with the first version of the kernel);
    ADYOLO_W4_MIN_K_32 (32): smallest contraction for the F(4x4) form with 32-channel OUTPUT blocks (stage 1's 32 -> 32 layers
    and the 64 -> 32 data-gradient: 1.04-1.14 x the F(2x2) kernel per launch, profiles/r05_w4p_nb1_ab.txt).
    On / off switches in the same table: ADYOLO_W4_PERSIST (persistent F(4x4) kernel; 0 = the one-patch form), ADYOLO_W4_NARROW
    (patches one / two tiles wide on maps up to 8 bins wide), ADYOLO_WINO1D (1-D Winograd along time on one-bin-wide maps),
    ADYOLO_WGRAD_ALGO (weight-gradient algorithm when it differs from ADYOLO_CONV_ALGO).
    Returns the numeric thresholds (``switch_table()``: everything)."""
    env = os.environ.get
    W4_THRESHOLDS.update(min_k=int(env("ADYOLO_W4_MIN_K", "32")),
                         min_k_addend=int(env("ADYOLO_W4_MIN_K_ADDEND", "32")),
                         min_wgs=int(env("ADYOLO_W4_MIN_WGS", "200")),
                         min_wgrad_work=int(env("ADYOLO_W4W_MIN_WORK", "6000000")),
                         min_k_32=int(env("ADYOLO_W4_MIN_K_32", "32")),
                         persist=env("ADYOLO_W4_PERSIST", "1") != "0",
                         narrow=env("ADYOLO_W4_NARROW", "1") != "0",
                         wino1d=env("ADYOLO_WINO1D", "1") != "0",
                         wgrad_algo=(env("ADYOLO_WGRAD_ALGO") or "").lower() or None)
    if _lib.loaded():                                  # (not loaded yet: the library reads the environment at its first launch)
        _lib.load().adyolo_reload_switches()
    return {k: W4_THRESHOLDS[k] for k in _THRESHOLD_KEYS}


def switch_table():
    """Every cached dispatch switch + the algorithm: what a recorded graph's kernel choice depends on besides its input shape
    (``graph.ForwardGraphs._stamp`` / ``StepGraphs`` key on it) and what bench.py echoes in ``dispatch``."""
    t = dict(W4_THRESHOLDS)
    t["conv_algo"] = conv_algo()
    return t


def switch_stamp():
    return tuple(sorted((k, str(v)) for k, v in switch_table().items()))


reload_thresholds()

# When set to a dict, ``conv3x3`` counts its launches in it: key = (kernel name, Cin, Cout, operand bits) -- the table bench.py prints as
# ``dispatch`` and tests/test_gpu_parity_scale.py asserts at the bench shape.
DISPATCH_LOG = None


W4P_EPIS = (0, 1, 2, 9, 15, 27, 31)            # operand combinations the persistent F(4x4) kernel is built for (csrc/wino4p_launch.hpp)


def w4_narrow_ok(cin, cout):
    """A stride-1 3x3 convolution on a map at most 8 bins wide gets the persistent F(4x4) kernel with patches one / two tiles
    wide (``adyolo_wino4_fwd``, plain launches, 64-channel output blocks in both directions) -- the ResNet-Conformer's middle
    stages: 2.9 x the implicit GEMM at 32 x 800 x 4 x 128 (tools/wino4/narrow_check.py)."""
    return (conv_algo() == "winograd4" and cin % 64 == 0 and cout % 64 == 0 and _w4_eligible(cin, cout) and _w4_eligible(cout, cin)
            and W4_THRESHOLDS["narrow"] and W4_THRESHOLDS["persist"])


def _w4_eligible(k_gemm, n_gemm, allow32=False):
    """Does this GEMM direction (contraction over k_gemm channels, n_gemm output channels) get the F(4x4,3x3) form
    (csrc/wino4.hip, wino4p.hpp) packed beside the F(2x2) one?  The kernels take 16-channel pairs of the contraction and 64
    output channels per workgroup -- or 32 (round 5; the persistent kernel only, which deals 1 / 2 / 4 / 8 channel blocks to
    the XCDs: ``DualPack.pick`` keeps such a launch on F(2x2) when its operand combination has no persistent form).  One
    workgroup per CU: it wins by more the longer the contraction -- per launch at the bench shapes (round 5, persistent form)
    1.4-1.6 x the F(2x2) kernel at 256 channels, 1.3-1.4 x at 128, 1.2-1.4 x at 64; DESIGN.md, "F(4x4,3x3), round 5".
    ADYOLO_W4_MIN_K moves the threshold (``reload_thresholds``).  allow32: also 32-channel output blocks (``WinoPackSet``, whose
    ``DualPack`` can fall back; plain ``pack_w3x3`` packs them only on request)."""
    if k_gemm % 32 or k_gemm > 512:
        return False
    if n_gemm % 64 == 0:
        return k_gemm >= W4_THRESHOLDS["min_k"]
    return allow32 and n_gemm % 32 == 0 and n_gemm // 32 in (1, 2, 4, 8) and k_gemm >= W4_THRESHOLDS["min_k_32"]


def pack_w3x3(w, cin_pad, want_dgrad=True, algo=None, allow32=False):
    """w [Cout][Cin][3][3] -> (fwd pack, dgrad pack or None).

    direct:   wpk_fwd [Cout][9][cin_pad], wpk_dgrad [cin_pad][9][Cout]
    winograd: u_fwd [16][Cout/32][cin_pad/8][256], u_dgrad [16][cin_pad/32][Cout/8][256]  (needs both channel
              counts to be multiples of 32; the 8-channel stem stays direct).  ``conv3x3`` tells them apart by rank.
    winograd4: per direction the F(4x4,3x3) form [36][N/32][K/8][256] where ``_w4_eligible`` says so, else the F(2x2) form."""
    _chk(w)
    cout, cin = w.shape[0], w.shape[1]
    algo = algo or conv_algo()
    if algo == "winograd4" and cin_pad % 32 == 0 and cout % 32 == 0:
        # per direction: the F(4x4) form [36][N/32][K/8][256] where it applies, else the F(2x2) form
        f4, d4 = _w4_eligible(cin_pad, cout, allow32), want_dgrad and _w4_eligible(cout, cin_pad, allow32)
        uf = _new(w, 36 if f4 else 16, cout // 32, cin_pad // 8, 256)
        ud = _new(w, 36 if d4 else 16, cin_pad // 32, cout // 8, 256) if want_dgrad else None
        if f4 or d4:
            _c("adyolo_wino4_pack_w", _p(w), _p(uf if f4 else None), _p(ud if d4 else None), cout, cin, cin_pad, _stream())
        if not f4 or (want_dgrad and not d4):
            _c("adyolo_wino_pack_w", _p(w), _p(None if f4 else uf), _p(None if (d4 or not want_dgrad) else ud), cout, cin,
               cin_pad, _stream())
        return uf, ud
    if algo == "winograd" and cin_pad % 32 == 0 and cout % 32 == 0:
        uf = _new(w, 16, cout // 32, cin_pad // 8, 256)
        ud = _new(w, 16, cin_pad // 32, cout // 8, 256) if want_dgrad else None
        _c("adyolo_wino_pack_w", _p(w), _p(uf), _p(ud), cout, cin, cin_pad, _stream())
        return uf, ud
    wf = _new(w, cout, 9, cin_pad)
    wd = _new(w, cin_pad, 9, cout) if want_dgrad else None
    _c("adyolo_pack_w3x3", _p(w), _p(wf), _p(wd), cout, cin, cin_pad, _stream())
    return wf, wd


class DualPack:
    """Both Winograd forms of one filter direction: ``f4`` [36][N/32][K/8][256] for the F(4x4,3x3) kernel, ``f2`` [16][...] for
    the F(2x2) kernel.  ``conv3x3`` takes the F(4x4) form when its launch fills the chip and the F(2x2) form for small grids
    (one workgroup per CU and 32 x 16 / 16 x 32-pixel patches: a single 60 s clip is 76 workgroups at stage 4)."""
    __slots__ = ("f4", "f2")

    def __init__(self, f4, f2):
        self.f4, self.f2 = f4, f2

    def pick(self, n, h, w, cout, addend=False, persistent_ok=True):
        """addend: the launch adds a tensor in its epilogue (the data-gradient of a block's first convolution); until round 5 the
        F(2x2) kernel was the faster one there below 128 channels (ADYOLO_W4_MIN_K_ADDEND, now 32: no effect by default).  persistent_ok: the launch's
        operand combination has a persistent form (``W4P_EPIS``, no bias, masks as bits) -- with 32-channel output blocks
        (cout % 64 != 0) there is no other F(4x4) kernel."""
        if cout % 64 and not (persistent_ok and W4_THRESHOLDS["persist"]):
            return self.f2
        wgs = _lib.load().adyolo_wino4_tiles(n, h, w) * ((cout + 63) // 64)
        if wgs < W4_THRESHOLDS["min_wgs"]:
            return self.f2
        if addend and self.f4.shape[2] * 8 < W4_THRESHOLDS["min_k_addend"]:
            return self.f2
        return self.f4


class WinoPackSet:
    """The Winograd-packed forms (forward + data-gradient) of a fixed list of 3x3 filters, refreshed by ONE launch
    (``adyolo_wino_pack_many``) instead of two per filter: ``refresh()`` at the start of a forward pass, ``get(i)`` ->
    (u_fwd, u_dgrad) of filter i.  The buffers and the descriptor table are allocated once and rebuilt only when a filter's
    storage moves (``.to()``, re-homing into a flat parameter buffer)."""

    def __init__(self):
        self.key = self.stamp = None

    def refresh(self, weights, frozen=False):
        """frozen (evaluation mode): skip the launch when nothing was written to the filters since the last one
        (``PARAMS_EPOCH`` for in-place kernels, the tensors' version counters for ``copy_`` / ``load_state_dict``)."""
        w4 = conv_algo() == "winograd4"
        key = (w4, W4_THRESHOLDS["min_k"], W4_THRESHOLDS["min_k_32"]) + tuple(w.data_ptr() for w in weights)
        stamp = (PARAMS_EPOCH[0],) + tuple(w._version for w in weights)
        if frozen and key == self.key and stamp == self.stamp:
            return self
        self.stamp = stamp
        if key != self.key:
            dev = weights[0].device
            self.packs, rows, rows4 = [], [], []
            for w in weights:
                _chk(w)
                cout, cin = w.shape[0], w.shape[1]
                if cout % 32 or cin % 32:
                    raise _lib.AdyoloHipError("WinoPackSet: channel counts must be multiples of 32")
                # per direction: the F(2x2) form always, the F(4x4) form [36][N/32][K/8][256] beside it where it applies
                # (winograd4): ``conv3x3`` picks by the size of the launch (``DualPack``)
                f4, d4 = w4 and _w4_eligible(cin, cout, True), w4 and _w4_eligible(cout, cin, True)
                uf = _new(w, 16, cout // 32, cin // 8, 256)
                ud = _new(w, 16, cin // 32, cout // 8, 256)
                rows.append([w.data_ptr(), uf.data_ptr(), ud.data_ptr(), cout, cin, cin, 0, 0])
                if f4 or d4:
                    uf4 = _new(w, 36, cout // 32, cin // 8, 256) if f4 else None
                    ud4 = _new(w, 36, cin // 32, cout // 8, 256) if d4 else None
                    rows4.append([w.data_ptr(), uf4.data_ptr() if f4 else 0, ud4.data_ptr() if d4 else 0, cout, cin, cin, 0, 0])
                    uf = DualPack(uf4, uf) if f4 else uf
                    ud = DualPack(ud4, ud) if d4 else ud
                self.packs.append((uf, ud))
            self.table = torch.tensor(rows, dtype=torch.int64, device=dev) if rows else None
            self.table4 = torch.tensor(rows4, dtype=torch.int64, device=dev) if rows4 else None
            self.nrows, self.nrows4 = len(rows), len(rows4)
            self.max_cout = max(w.shape[0] for w in weights)
            self.max_cin = max(w.shape[1] for w in weights)
            self.key = key
        if self.table is not None:
            _c("adyolo_wino_pack_many", _p(self.table), self.nrows, self.max_cout, self.max_cin, _stream())
        if self.table4 is not None:
            _c("adyolo_wino4_pack_many", _p(self.table4), self.nrows4, self.max_cout, self.max_cin, _stream())
        return self

    def get(self, i):
        return self.packs[i]


def conv3x3(x, wpk, cout, bias=None, addend=None, relu=False, addend_mask=None, in_affine=None, want_stats=False,
            stat_bn=None, stat_mask=None):
    """x [N][H][W][Cin] -> [N][H][W][cout];  wpk from ``pack_w3x3`` (direct [cout][9][Cin] or Winograd, rank 4).

    in_affine=(scale, shift): the producer's BatchNorm affine is applied while staging x (padding stays 0);
    addend_mask: addend is multiplied by (mask > 0); want_stats: also return the per-patch channel sums
    [2][tiles][cout] of the output for ``bn_stats_tiles``; stat_bn=(aux, mean, invstd): the second per-patch sum
    becomes sum(y * xhat(aux)) (the output is a gradient, aux the BatchNorm input) for ``bn_bwd(..., tile_stats=)``;
    stat_mask: both sums are taken of y * (stat_mask > 0) (for ``se_tail_bwd(..., tile_stats=)`` of the block below).
    addend_mask / stat_mask may also be the int64 ReLU-mask BITS ``se_tail_fwd(..., want_mask=True)`` returned."""
    mbits = 0
    if addend_mask is not None and addend_mask.dtype == torch.int64:
        mbits |= 1
    if stat_mask is not None and stat_mask.dtype == torch.int64:
        mbits |= 2
    n, h, w, cin = x.shape
    # operand combination, as the persistent F(4x4) kernel's EPI bits: 1 statistics, 2 addend, 4 addend mask, 8 stat_bn, 16 stat mask
    epi = (1 if want_stats else 0) | (2 if addend is not None else 0) | (4 if addend_mask is not None else 0) | \
          (8 if stat_bn is not None else 0) | (16 if stat_mask is not None else 0)
    p_ok = bias is None and epi in W4P_EPIS and (addend_mask is None or mbits & 1) and (stat_mask is None or mbits & 2)
    if isinstance(wpk, DualPack):
        wpk = wpk.pick(n, h, w, cout, addend is not None, p_ok)
    elif wpk.dim() == 4 and wpk.shape[0] == 36 and cout % 64 and not p_ok:
        raise _lib.AdyoloHipError("conv3x3: a F(4x4) pack with %d output channels needs an operand combination the persistent kernel "
                                  "is built for (pack with ops.DualPack / WinoPackSet to fall back to F(2x2))" % cout)
    _chk(x, wpk, bias, addend, None if mbits & 1 else addend_mask, None if mbits & 2 else stat_mask)
    wino = wpk.dim() == 4
    wino4 = wino and wpk.shape[0] == 36
    y = _new(x, n, h, w, cout)
    stats = None
    if want_stats:
        lib = _lib.load()
        tiles = (lib.adyolo_wino4_tiles if wino4 else lib.adyolo_wino_tiles if wino else lib.adyolo_conv3x3_tiles)(n, h, w)
        stats = _new(x, 2, tiles, cout)
    sc, sh = in_affine if in_affine is not None else (None, None)
    sa, sm, si = stat_bn if stat_bn is not None else (None, None, None)
    if wino4:
        fn = "adyolo_wino4_fwd"
    else:
        fn = "adyolo_wino_fwd" if wino else "adyolo_conv3x3_fwd"
    _c(fn, _p(x), _p(wpk), _p(bias), _p(addend), _p(addend_mask),
       _p(sc), _p(sh), _p(y), _p(stats), _p(sa), _p(sm), _p(si), _p(stat_mask), n, h, w, cin, cout, int(relu), mbits,
       _stream())
    if DISPATCH_LOG is not None:
        if wino4:
            name = "wino4p_fwd_kernel" if _lib.load().adyolo_wino4_last_form() == 2 else "wino4_fwd_kernel"
        else:
            name = "wino_fwd_kernel" if wino else "conv3x3_fwd_kernel"
        k = (name, cin, cout, epi)
        DISPATCH_LOG[k] = DISPATCH_LOG.get(k, 0) + 1
    return (y, stats) if want_stats else y


def wgrad_form(cin, cout, algo=None, shape=None):
    """-> (kernel name, matrix FLOPs issued / direct-convolution FLOPs) of the weight-gradient ``conv3x3_wgrad`` launches for these
    channel counts (and, with ``shape`` = (N, H, W), this launch): the Winograd F(4x4,3x3) domain (csrc/wino4w.hip, round 5: 9
    multiplies per 36) under 'winograd4' when the kernel takes the shape (Cout % 32, W % 16, H % 4) and the launch is large
    enough (ADYOLO_W4W_MIN_WORK; one workgroup per CU: small launches stay on the two-per-CU kernel); else
    the F(2x2,3x3) domain (16 per 36) when both channel counts are multiples of 32; else the direct implicit GEMM.  The ONE
    place that decides it -- ``conv3x3_wgrad`` and bench.py both ask here."""
    algo = algo or W4_THRESHOLDS["wgrad_algo"] or conv_algo()
    if algo == "winograd4" and shape is not None and cin % 32 == 0 and cout % 32 == 0:
        n, h, w = shape
        if (w % 16 == 0 or w in (4, 8)) and h % 4 == 0 and n * w // 16 * (h // 4) * cin * cout >= W4_THRESHOLDS["min_wgrad_work"] and \
                _lib.load().adyolo_wino4_wgrad_slabs(n, h, w, cin, cout) > 0:
            return "wino4_wgrad_kernel", 9.0 / 36.0
    if algo in ("winograd", "winograd4") and cin % 32 == 0 and cout % 32 == 0:
        return "wino_wgrad_kernel", 16.0 / 36.0
    return "conv3x3_wgrad_kernel", 1.0


def conv3x3_wgrad(x, dy, cin_real, in_affine=None, algo=None, out=None):
    """x [N][H][W][Cin], dy [N][H][W][Cout] -> dw [Cout][cin_real][3][3] (x optionally seen through an affine).

    Winograd form (default) when both channel counts are multiples of 32, else the direct implicit GEMM."""
    _chk(x, dy)
    n, h, w, cin = x.shape
    cout = dy.shape[3]
    sc, sh = in_affine if in_affine is not None else (None, None)
    dw = out if out is not None else _new(x, cout, cin_real, 3, 3)      # out: e.g. the parameter's slice of the flat gradient buffer
    form = wgrad_form(cin, cout, algo, (n, h, w))[0]
    if DISPATCH_LOG is not None:
        k = (form, cin, cout, 1 if in_affine is not None else 0)
        DISPATCH_LOG[k] = DISPATCH_LOG.get(k, 0) + 1
    if form == "wino4_wgrad_kernel":
        nslab = _lib.load().adyolo_wino4_wgrad_slabs(n, h, w, cin, cout)
        slabs = _new(x, nslab, 36, cin, cout)
        du = _new(x, 36, cin, cout)
        _c("adyolo_wino4_wgrad", _p(x), _p(dy), _p(sc), _p(sh), _p(slabs), _p(du), _p(dw), n, h, w, cin, cin_real, cout, _stream())
        return dw
    if form == "wino_wgrad_kernel":
        nslab = _lib.load().adyolo_wino_wgrad_slabs(n, h, w, cin, cout)
        if nslab <= 0:
            raise _lib.AdyoloHipError("wino_wgrad_slabs rejected the shape")
        slabs = _new(x, nslab, 16, cin, cout)
        du = _new(x, 16, cin, cout)
        _c("adyolo_wino_wgrad", _p(x), _p(dy), _p(sc), _p(sh), _p(slabs), _p(du), _p(dw), n, h, w, cin, cin_real, cout,
           _stream())
        return dw
    nslab = _lib.load().adyolo_conv3x3_wgrad_slabs(n, h, w, cin, cout)
    if nslab <= 0:
        raise _lib.AdyoloHipError("conv3x3_wgrad_slabs rejected the shape")
    cinp = ((cin + 31) // 32) * 32
    slabs = _new(x, nslab, cout, 9, cinp)
    _c("adyolo_conv3x3_wgrad", _p(x), _p(dy), _p(sc), _p(sh), _p(slabs), _p(dw), n, h, w, cin, cin_real, cout, _stream())
    return dw


# ---------------------------------------------------------------------------------------------- gemm
def gemm(a, b, m, n, k, lda, ldb, trans_a=False, trans_b=False, bias=None, out=None, ldc=None, accumulate=False,
         splits=1):
    """C[m][n] = sum_k opA(m,k) opB(n,k) (+bias).  ``a``/``b``/``out`` may be views with an offset."""
    if ldc is None:
        ldc = n
    if out is None:
        out = _new(a, m, n)
        ldc = n
    slabs = None
    if splits > 1:
        slabs = _new(a, splits, m, n)
    _c("adyolo_gemm", _p(a), _p(b), _p(bias), _p(out), _p(slabs), m, n, k, lda, ldb, ldc, int(trans_a), int(trans_b),
       splits, int(accumulate), _stream())
    return out


def wgrad_splits(m, n, k):
    """Split-K factor for a weight-gradient GEMM (m x n output, contraction k = rows of the batch): enough 128 x 64 output
    tiles x splits to put ~2 workgroups on each of the 256 CUs, at least 512 contraction rows per split, at most 256
    (a 64 x 32 output over 2.5 M rows -- the 1x1 shortcut convolution of stage 2 -- ran on 64 workgroups with the old cap
    of 64: 1.54 ms for a 0.94 GB read)."""
    tiles = ((m + 127) // 128) * ((n + 63) // 64)
    want = (512 + tiles - 1) // tiles
    return max(1, min(256, want, k // 512))


def linear(x2d, w, bias=None):
    """x2d [R][K] @ w[N][K]^T + bias."""
    _chk(x2d, w, bias)
    r, k = x2d.shape
    return gemm(x2d, w, r, w.shape[0], k, k, k, bias=bias)


def linear_bwd(x2d, w, dy2d, need_dx=True, out_dw=None, out_db=None, need_db=True):
    """-> dx [R][K], dw [N][K], db [N].  out_dw / out_db: write there instead (e.g. the parameters' slices of the flat
    gradient buffer, ``functional.GradSink``)."""
    _chk(x2d, w, dy2d)
    r, k = x2d.shape
    n = w.shape[0]
    dx = gemm(dy2d, w, r, k, n, n, k, trans_b=True) if need_dx else None
    dw = gemm(dy2d, x2d, n, k, r, n, k, trans_a=True, trans_b=True, splits=wgrad_splits(n, k, r), out=out_dw)
    db = colsum(dy2d, out=out_db) if need_db else None
    return dx, dw, db


def colsum(a2d, out=None, accumulate=False):
    r, c = a2d.shape
    if out is None:
        out = _new(a2d, c)
        accumulate = False
    partial = _new(a2d, 1024, c)
    _c("adyolo_colsum", _p(a2d), _p(out), _p(partial), r, c, a2d.stride(0), int(accumulate), _stream())
    return out


def add(a, b):
    _chk(a, b)
    y = torch.empty_like(a)
    _c("adyolo_add", _p(a), _p(b), _p(y), a.numel(), _stream())
    return y


def scale_dev(a, scalar_dev):
    """a * scalar_dev[0], the scalar staying on the device (no host sync)."""
    _chk(a, scalar_dev)
    y = torch.empty_like(a)
    _c("adyolo_scale_dev", _p(a), _p(scalar_dev), _p(y), a.numel(), _stream())
    return y


def mul(a, b):
    _chk(a, b)
    y = torch.empty_like(a)
    _c("adyolo_mul", _p(a), _p(b), _p(y), a.numel(), _stream())
    return y


# ---------------------------------------------------------------------------------------------- norm
def bn_stats(x, running_mean=None, running_var=None, momentum=0.1, eps=1e-5):
    """x [N][H][W][C] -> per-sample sums [N][C], mean [C], invstd [C]; updates running stats in place."""
    _chk(x, running_mean, running_var)
    n, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (n * c)
    ssum, mean, invstd = _new(x, n, c), _new(x, c), _new(x, c)
    partial = _new(x, 4 * 1024 * c)
    if EXACT.on and running_mean is not None:         # training statistics over the batch of all ranks
        _c("adyolo_bn_stats", _p(x), _p(ssum), _p(mean), _p(invstd), NULL, NULL, _p(partial), n, hw, c, momentum, eps, _stream())
        ps1 = partial[3 * 1024 * c:3 * 1024 * c + n * c].view(n, c)          # per-sample sums of squares (layout of the C side)
        g0, g1 = EXACT.gather_rows(ssum), EXACT.gather_rows(ps1)
        _c("adyolo_bn_finish", _p(g0), _p(g1), _p(mean), _p(invstd), _p(running_mean), _p(running_var), NULL, NULL, NULL, NULL,
           g0.shape[0], hw, c, momentum, eps, _stream())
        return ssum, mean, invstd
    _c("adyolo_bn_stats", _p(x), _p(ssum), _p(mean), _p(invstd), _p(running_mean), _p(running_var), _p(partial), n,
       hw, c, momentum, eps, _stream())
    return ssum, mean, invstd


def bn_stats_tiles(tile_stats, n, hw, running_mean=None, running_var=None, momentum=0.1, eps=1e-5, gamma=None,
                   beta=None):
    """BatchNorm statistics from the per-patch sums of a convolution epilogue ([2][tiles][C]) -> ssum, mean, invstd
    (+ scale = gamma*invstd, shift = beta - mean*scale from the same finishing launch when gamma / beta are given)."""
    _chk(tile_stats, running_mean, running_var, gamma, beta)
    tiles, c = tile_stats.shape[1], tile_stats.shape[2]
    ssum, mean, invstd = _new(tile_stats, n, c), _new(tile_stats, c), _new(tile_stats, c)
    scale, shift = (_new(tile_stats, c), _new(tile_stats, c)) if gamma is not None else (None, None)
    if EXACT.on and running_mean is not None:         # training statistics over the batch of all ranks
        ps1 = _new(tile_stats, n, c)
        _c("adyolo_bn_persample", _p(tile_stats), _p(ssum), _p(ps1), n, tiles // n, c, _stream())
        g0, g1 = EXACT.gather_rows(ssum), EXACT.gather_rows(ps1)
        _c("adyolo_bn_finish", _p(g0), _p(g1), _p(mean), _p(invstd), _p(running_mean), _p(running_var), _p(gamma), _p(beta),
           _p(scale), _p(shift), g0.shape[0], hw, c, momentum, eps, _stream())
    else:
        partial = _new(tile_stats, 2 * 1024 * c)
        _c("adyolo_bn_stats_tiles", _p(tile_stats), _p(ssum), _p(mean), _p(invstd), _p(running_mean), _p(running_var),
           _p(gamma), _p(beta), _p(scale), _p(shift), _p(partial), n, tiles // n, hw, c, momentum, eps, _stream())
    if gamma is not None:
        return ssum, mean, invstd, scale, shift
    return ssum, mean, invstd


def bn_eval_stats(running_mean, running_var, eps=1e-5):
    _chk(running_mean, running_var)
    c = running_mean.numel()
    mean, invstd = _new(running_mean, c), _new(running_mean, c)
    _c("adyolo_bn_eval_stats", _p(running_mean), _p(running_var), _p(mean), _p(invstd), c, eps, _stream())
    return mean, invstd


def bn_scale_shift(gamma, beta, mean, invstd):
    _chk(gamma, beta, mean, invstd)
    c = gamma.numel()
    scale, shift = _new(gamma, c), _new(gamma, c)
    _c("adyolo_bn_scale_shift", _p(gamma), _p(beta), _p(mean), _p(invstd), _p(scale), _p(shift), c, _stream())
    return scale, shift


def affine(x, scale, shift):
    _chk(x, scale, shift)
    c = x.shape[-1]
    y = torch.empty_like(x)
    _c("adyolo_affine_nhwc", _p(x), _p(scale), _p(shift), _p(y), x.numel() // c, c, _stream())
    return y


def bn_bwd(dy, x, gamma, mean, invstd, relu_mask=False, tile_stats=None, out_dgamma=None, out_dbeta=None,
           want_dx_colsum=False):
    """-> dx, dgamma, dbeta.  relu_mask: additionally multiply dx by (x > 0) (x is a ReLU output).
    tile_stats: per-patch (sum dy, sum dy*xhat) written by the convolution that produced dy (skips the reduce pass)."""
    _chk(dy, x, gamma, mean, invstd)
    c = x.shape[-1]
    rows = x.numel() // c
    sdy = out_dbeta if out_dbeta is not None else _new(x, c)          # = dbeta
    sdyx = out_dgamma if out_dgamma is not None else _new(x, c)       # = dgamma
    if tile_stats is not None:
        _c("adyolo_bn_bwd_tiles", _p(tile_stats), _p(sdy), _p(sdyx), _p(_new(tile_stats, 2, 256, c)), tile_stats.shape[1], c,
           _stream())
    else:
        partial = _new(x, 2 * 1024 * c)
        _c("adyolo_bn_bwd_reduce", _p(dy), _p(x), _p(mean), _p(invstd), _p(sdy), _p(sdyx), _p(partial), rows, c,
           _stream())
    dx = torch.empty_like(x)
    colsum = part = None
    if want_dx_colsum:          # channel sums of dx from the same pass (e.g. the bias gradient of the producing convolution)
        colsum, part = _new(x, c), _new(x, 8192, c)
    a_sdy, a_sdyx, cs = sdy, sdyx, 1.0
    if EXACT.on:                # the dx formula needs the sums over the batch of ALL ranks; sdy / sdyx (= the parameter
        glob = EXACT.all_reduce(torch.stack([sdy, sdyx]))      # gradients, possibly slices of the flat buffer) stay local
        a_sdy, a_sdyx, cs = glob[0], glob[1], float(EXACT.world)
    _c("adyolo_bn_bwd_apply", _p(dy), _p(x), _p(gamma), _p(mean), _p(invstd), _p(a_sdy), _p(a_sdyx), _p(dx), NULL, NULL,
       _p(colsum), _p(part), rows, c, int(relu_mask), cs, _stream())
    if want_dx_colsum:
        return dx, sdyx, sdy, colsum
    return dx, sdyx, sdy


def se_fc_fwd(ssum, scale, shift, w1, b1, w2, b2, hw):
    _chk(ssum, scale, shift, w1, b1, w2, b2)
    n, c = ssum.shape
    cr = w1.shape[0]
    pooled, hid, s = _new(ssum, n, c), _new(ssum, n, cr), _new(ssum, n, c)
    _c("adyolo_se_fc_fwd", _p(ssum), _p(scale), _p(shift), _p(w1), _p(b1), _p(w2), _p(b2), _p(pooled), _p(hid), _p(s),
       n, hw, c, cr, _stream())
    return pooled, hid, s


def se_tail_pool_ok(h, w, ch):
    """True when ``se_tail_fwd(..., pool_hw=(h, w))`` takes the shape (adyolo_se_tail_fwd_pool_ok)."""
    return bool(_lib.load().adyolo_se_tail_fwd_pool_ok(int(h), int(w), int(ch)))


def se_tail_fwd(c_t, r_t, scale, shift, s, want_mask=False, r_affine=None, pool_hw=None):
    """e = relu((c*scale+shift)*s + r), r seen through r_affine = (scale, shift) per channel when given.  want_mask: also return the ReLU mask (e > 0) as bits (int64 words, 1/32 of the
    bytes of e) for ``se_tail_bwd(..., mask=)``; None when the shape does not support it (HW*C/4 % 64 != 0).
    pool_hw = (H, W): return avgpool2(e) [N][H/2][W/2][C] INSTEAD of e (the block in front of a pooled stage boundary; e is
    never written), always as a pair (pooled, mask or None); the caller checks ``se_tail_pool_ok`` first."""
    _chk(c_t, r_t, scale, shift, s)
    n, ch = c_t.shape[0], c_t.shape[-1]
    hw = c_t.numel() // (n * ch)
    mask = None
    if want_mask:
        words = _lib.load().adyolo_relu_mask_words(n, hw, ch)
        if words > 0:
            mask = torch.empty(words, dtype=torch.int64, device=c_t.device)
    rs, rt = r_affine if r_affine is not None else (None, None)
    if pool_hw is not None:
        h, w = pool_hw
        if want_mask and mask is None:
            raise _lib.AdyoloHipError("se_tail_fwd(pool_hw=): the shape has no mask bits")
        pooled = _new(c_t, n, h // 2, w // 2, ch)
        _c("adyolo_se_tail_fwd_pool", _p(c_t), _p(r_t), _p(scale), _p(shift), _p(s), _p(rs), _p(rt), _p(pooled), _p(mask), n, h, w,
           ch, _stream())
        return pooled, mask
    e = torch.empty_like(c_t)
    _c("adyolo_se_tail_fwd", _p(c_t), _p(r_t), _p(scale), _p(shift), _p(s), _p(rs), _p(rt), _p(e), _p(mask), n, hw, ch,
       _stream())
    return (e, mask) if want_mask else e


def se_tail_bwd(de, e, c_t, gamma, beta, mean, invstd, ssum, pooled, hid, s, w1, w2, want_dr=True, tile_stats=None,
                mask=None, packed_out=None, pooled_hw=None, de_out=None):
    """Backward of  e = relu(bn2(c) * s + r)  incl. the SE FCs.
    -> dc, dr, dgamma, dbeta, dw1, db1, dw2, db2
    tile_stats: per-patch sums [2][tiles][C] of de * (e > 0) and de * (e > 0) * xhat(c) from the convolution epilogue
    that produced ``de`` (``conv3x3(..., stat_bn=(c, mean, invstd), stat_mask=e)``): the reduction pass is skipped.
    mask: the bits ``se_tail_fwd(..., want_mask=True)`` returned; both passes then read them instead of e.
    pooled_hw = (H, W): the block's output was avgpool2(e) (``se_tail_fwd(pool_hw=)``) and ``de`` is the gradient of THAT,
    [N][H/2][W/2][C]; needs ``mask``.  de_out (optional, [N][H][W][C]) receives the gradient of e itself (avgpool2's backward),
    which no pass here reads."""
    _chk(de, e, c_t, gamma, beta, mean, invstd, ssum, pooled, hid, s, w1, w2, de_out)
    if pooled_hw is not None and mask is None:
        raise _lib.AdyoloHipError("se_tail_bwd(pooled_hw=) needs the ReLU-mask bits")
    n, ch = c_t.shape[0], c_t.shape[-1]
    hw = c_t.numel() // (n * ch)
    cr = w1.shape[0]
    sg, sgx = _new(c_t, n, ch), _new(c_t, n, ch)
    if tile_stats is not None:
        _c("adyolo_se_tail_bwd_tiles", _p(tile_stats), _p(sg), _p(sgx), n, tile_stats.shape[1] // n, ch, _stream())
    else:
        partial = _new(c_t, 2 * 1024 * ch)
        if pooled_hw is not None:
            _c("adyolo_se_tail_bwd_reduce_pooled", _p(de), _p(mask), _p(c_t), _p(mean), _p(invstd), _p(sg), _p(sgx), _p(partial),
               n, pooled_hw[0], pooled_hw[1], ch, _stream())
        else:
            _c("adyolo_se_tail_bwd_reduce", _p(de), _p(e), _p(mask), _p(c_t), _p(mean), _p(invstd), _p(sg), _p(sgx), _p(partial),
               n, hw, ch, _stream())
    pw = 2 * ch * cr + cr + 3 * ch
    part, cws = _new(c_t, n, pw), _new(c_t, 1024, pw)
    packed = packed_out if packed_out is not None else _new(c_t, pw)    # packed_out: the six gradients' slice of the flat buffer
    dpool = _new(c_t, n, ch)
    _c("adyolo_se_fc_bwd", _p(sg), _p(sgx), _p(ssum), _p(gamma), _p(beta), _p(mean), _p(invstd), _p(pooled), _p(hid),
       _p(s), _p(w1), _p(w2), _p(dpool), _p(part), _p(packed), _p(cws), n, hw, ch, cr, _stream())
    o = 0
    db2 = packed[o:o + ch]; o += ch
    dw2 = packed[o:o + ch * cr].view(ch, cr); o += ch * cr
    db1 = packed[o:o + cr]; o += cr
    dw1 = packed[o:o + cr * ch].view(cr, ch); o += cr * ch
    sdd = packed[o:o + ch]; o += ch
    sddx = packed[o:o + ch]
    dgamma, dbeta = sddx, sdd
    dc = torch.empty_like(c_t)
    dr = torch.empty_like(c_t) if want_dr else None
    a_sdd, a_sddx, cs = sdd, sddx, 1.0
    if EXACT.on:                # BN2's batch sums over all ranks for the dc formula; the packed parameter gradients stay local
        glob = EXACT.all_reduce(torch.stack([sdd, sddx]))
        a_sdd, a_sddx, cs = glob[0], glob[1], float(EXACT.world)
    if pooled_hw is not None:
        _c("adyolo_se_tail_bwd_apply_pooled", _p(de), _p(mask), _p(c_t), _p(gamma), _p(mean), _p(invstd), _p(s), _p(dpool),
           _p(a_sdd), _p(a_sddx), _p(dc), _p(dr), _p(de_out), n, pooled_hw[0], pooled_hw[1], ch, cs, _stream())
    else:
        _c("adyolo_se_tail_bwd_apply", _p(de), _p(e), _p(mask), _p(c_t), _p(gamma), _p(mean), _p(invstd), _p(s), _p(dpool),
           _p(a_sdd), _p(a_sddx), _p(dc), _p(dr), n, hw, ch, cs, _stream())
    return dc, dr, dgamma, dbeta, dw1, db1, dw2, db2


def avgpool2(x):
    _chk(x)
    n, h, w, c = x.shape
    y = _new(x, n, h // 2, w // 2, c)
    _c("adyolo_avgpool2_fwd", _p(x), _p(y), n, h, w, c, _stream())
    return y


def avgpool2_bwd(dy, h, w):
    _chk(dy)
    n, c = dy.shape[0], dy.shape[3]
    dx = _new(dy, n, h, w, c)
    _c("adyolo_avgpool2_bwd", _p(dy), _p(dx), n, h, w, c, _stream())
    return dx


# ---------------------------------------------------------------------------------------------- sequence
def sap_fwd(x, w, b):
    """x [R][F][256] -> y [R][256], attn [R][F]."""
    _chk(x, w, b)
    r, f, c = x.shape
    y, attn = _new(x, r, c), _new(x, r, f)
    _c("adyolo_sap_fwd", _p(x), _p(w), _p(b), _p(y), _p(attn), r, f, c, _stream())
    return y, attn


def sap_bwd(dy, x, w, attn, out_dw=None, out_db=None):
    """out_dw [C] / out_db [1]: ZEROED accumulators to sum into instead of fresh ones (``functional.GradSink``: the parameters' slices
    of the flat gradient buffer, zero since ``zero_grad``)."""
    _chk(dy, x, w, attn)
    r, f, c = x.shape
    dx = torch.empty_like(x)
    dw = out_dw if out_dw is not None else _zeros(x, c)
    db = out_db if out_db is not None else _zeros(x, 1)
    partial = _new(x, 1024 * 260)
    _c("adyolo_sap_bwd", _p(dy), _p(x), _p(w), _p(attn), _p(dx), _p(dw), _p(db), _p(partial), r, f, c, _stream())
    return dx, dw, db


def gru_fwd(gx, whh, bhh, save):
    """gx [B][T][2][384], whh [2][384][128], bhh [2][384] -> out [B][T][256] (+ gates, hprev when ``save``)."""
    _chk(gx, whh, bhh)
    b, t = gx.shape[0], gx.shape[1]
    out = _new(gx, b, t, 256)
    gates = _new(gx, b, t, 2, 4, 128) if save else None
    hprev = _new(gx, b, t, 2, 128) if save else None
    _c("adyolo_gru_fwd", _p(gx), _p(whh), _p(bhh), _p(out), _p(gates), _p(hprev), b, t, _stream())
    return out, gates, hprev


def gru_bwd(dout, gates, hprev, whh):
    _chk(dout, gates, hprev, whh)
    b, t = dout.shape[0], dout.shape[1]
    dgx, dgh = _new(dout, b, t, 2, 384), _new(dout, b, t, 2, 384)
    _c("adyolo_gru_bwd", _p(dout), _p(gates), _p(hprev), _p(whh), _p(dgx), _p(dgh), b, t, _stream())
    return dgx, dgh


def ln_tanh_fwd(x2d, gamma, beta, eps=1e-5):
    _chk(x2d, gamma, beta)
    r, c = x2d.shape
    y = torch.empty_like(x2d)
    _c("adyolo_ln_tanh_fwd", _p(x2d), _p(gamma), _p(beta), _p(y), r, c, eps, _stream())
    return y


def ln_tanh_bwd(dy2d, x2d, y2d, gamma, eps=1e-5, out_dgamma=None, out_dbeta=None):
    """out_dgamma / out_dbeta [C]: ZEROED accumulators to sum into instead of fresh ones (see ``sap_bwd``)."""
    _chk(dy2d, x2d, y2d, gamma)
    r, c = x2d.shape
    dx = torch.empty_like(x2d)
    dgamma = out_dgamma if out_dgamma is not None else _zeros(x2d, c)
    dbeta = out_dbeta if out_dbeta is not None else _zeros(x2d, c)
    partial = _new(x2d, 1024 * 512)
    _c("adyolo_ln_tanh_bwd", _p(dy2d), _p(x2d), _p(y2d), _p(gamma), _p(dx), _p(dgamma), _p(dbeta), _p(partial), r, c,
       eps, _stream())
    return dx, dgamma, dbeta


def dropout_mask(like, p, seed, offset):
    mask = torch.empty_like(like)
    _c("adyolo_dropout_mask", _p(mask), mask.numel(), float(p), ctypes.c_uint64(seed), ctypes.c_uint64(offset), _stream())
    return mask


def dropout_apply(x, p, seed, offset, offset_dev=None):
    """x * mask, mask = dropout_mask(x, p, seed, offset) generated on the fly (no mask tensor).
    offset_dev (int64 tensor of one element on the device, or None): its value is added to ``offset`` inside the kernel --
    the form a launch recorded in a hipGraph needs (``rng.DropoutStream`` while a step is being captured)."""
    _chk(x)
    y = torch.empty_like(x)
    if offset_dev is None:
        _c("adyolo_dropout_apply", _p(x), _p(y), x.numel(), float(p), ctypes.c_uint64(seed), ctypes.c_uint64(offset), _stream())
    else:
        _c("adyolo_dropout_apply_dev", _p(x), _p(y), x.numel(), float(p), ctypes.c_uint64(seed),
           ctypes.c_uint64(offset & 0xFFFFFFFFFFFFFFFF), _p(offset_dev), _stream())
    return y


def counter_add_(counter, inc):
    """counter[0] += inc on the device (int64 tensor of one element): advances a device-side dropout offset."""
    if not counter.is_cuda or counter.dtype != torch.int64 or counter.numel() != 1:
        raise _lib.AdyoloHipError("counter_add_ needs a one-element int64 tensor on the device")
    _c("adyolo_counter_add", _p(counter), ctypes.c_uint64(int(inc) & 0xFFFFFFFFFFFFFFFF), _stream())
    return counter


# ---------------------------------------------------------------------------------------------- loss / optim
def adyolo_loss(logit, target, nb_classes, grid=(8, 4), anchors=5, thr=(45.0, 25.0, 10.0),
                gains=(5.0, 1.0, 5.0, 3.0), grid_size=(45.0, 45.0), g_overlap=0.5, need_grad=True, grad_scale=1.0,
                want_dist=False):
    """logit [B][T][G*A*(C+3)], target [M][7] (device) -> loss (1,), dlogit (or None), dist (or None)."""
    _chk(logit, target)
    b, t = logit.shape[0], logit.shape[1]
    m = target.shape[0]
    words = _lib.load().adyolo_loss_workspace_words(b * t, grid[0] * grid[1], anchors, m)
    ws = _new(logit, words)
    loss = _new(logit, 1)
    dlogit = torch.empty_like(logit) if need_grad else None
    dist = _new(logit, m, anchors) if want_dist else None
    thr_h = (ctypes.c_float * 3)(*[float(v) for v in thr])
    gains_h = (ctypes.c_float * 4)(*[float(v) for v in gains])
    args = (_p(logit), _p(target), _p(ws), _p(loss), _p(dlogit), _p(dist), b, t, grid[0], grid[1],
            anchors, nb_classes, m, ctypes.cast(thr_h, ctypes.c_void_p), ctypes.cast(gains_h, ctypes.c_void_p),
            float(grid_size[0]), float(grid_size[1]), float(g_overlap), float(grad_scale))
    if EXACT.on:                # counts over the batch of all ranks between the assignment and the pass over the logits
        _c("adyolo_loss_phase", *args, 1, 0, _stream())
        EXACT.all_reduce(ws[:4].view(torch.int32))
        na = b * t * grid[0] * grid[1] * anchors
        _c("adyolo_loss_phase", *args, 2, na * EXACT.world, _stream())
        EXACT.all_reduce(loss)                          # the ranks' shares add up to the loss of the concatenated batch
    else:
        _c("adyolo_loss_fwd_bwd", *args, _stream())
    return loss, dlogit, dist


def act_fwd(x2d, n_sigmoid_cols):
    """columns [0, n_sigmoid_cols) -> sigmoid, the rest -> tanh."""
    _chk(x2d)
    y = torch.empty_like(x2d)
    _c("adyolo_act_fwd", _p(x2d), _p(y), x2d.shape[0], x2d.shape[1], int(n_sigmoid_cols), _stream())
    return y


def act_bwd(dy2d, y2d, n_sigmoid_cols):
    _chk(dy2d, y2d)
    dx = torch.empty_like(y2d)
    _c("adyolo_act_bwd", _p(dy2d), _p(y2d), _p(dx), y2d.shape[0], y2d.shape[1], int(n_sigmoid_cols), _stream())
    return dx


def seddoa_loss(out2d, tgt2d, nsed, masked, w_bce, w_mse, need_grad=True):
    _chk(out2d, tgt2d)
    loss = _new(out2d, 1)
    dout = torch.empty_like(out2d) if need_grad else None
    partial = _new(out2d, 2 * 1024)
    _c("adyolo_seddoa_loss", _p(out2d), _p(tgt2d), _p(loss), _p(dout), _p(partial), out2d.shape[0], out2d.shape[1],
       int(nsed), int(masked), float(w_bce), float(w_mse), _stream())
    return loss, dout


def adpit_loss(out2d, tgt, nb_classes, need_grad=True):
    """out2d [rows][9*C], tgt [rows][6][4][C]."""
    _chk(out2d, tgt)
    loss = _new(out2d, 1)
    dout = torch.empty_like(out2d) if need_grad else None
    partial = _new(out2d, 1024)
    _c("adyolo_adpit_loss", _p(out2d), _p(tgt), _p(loss), _p(dout), _p(partial), out2d.shape[0], int(nb_classes), _stream())
    return loss, dout


def yolo_decode(logit, nb_classes, grid=(8, 4), anchors=5, grid_size=(45.0, 45.0), g_overlap=0.5):
    """logit [...][G*A*(C+3)] -> decoded [frames][Gaz][Gel][A][C+3] (conf, class-confidence scores, U, V)."""
    _chk(logit)
    ch = nb_classes + 3
    frames = logit.numel() // (grid[0] * grid[1] * anchors * ch)
    out = _new(logit, frames, grid[0], grid[1], anchors, ch)
    _c("adyolo_yolo_decode", _p(logit), _p(out), frames, grid[0], grid[1], anchors, nb_classes, float(grid_size[0]),
       float(grid_size[1]), float(g_overlap), _stream())
    return out


def adam_step(param, grad, exp_avg, exp_avg_sq, step, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
              grad_scale=1.0):
    _chk(param, grad, exp_avg, exp_avg_sq)
    _c("adyolo_adam_step", _p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), lr, betas[0], betas[1],
       eps, weight_decay, int(step), grad_scale, _stream())


def adam_step_dev(param, grad, exp_avg, exp_avg_sq, step_dev, bc_dev, lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                  weight_decay=0.0, grad_scale=1.0):
    """``adam_step`` with the step counter on the device: step_dev (int64, one element) is incremented by the call, bc_dev
    (2 floats) receives the bias corrections.  No argument changes from step to step (hipGraph-replayable)."""
    _chk(param, grad, exp_avg, exp_avg_sq, bc_dev)
    if not step_dev.is_cuda or step_dev.dtype != torch.int64 or step_dev.numel() != 1 or bc_dev.numel() < 2:
        raise _lib.AdyoloHipError("adam_step_dev needs a one-element int64 step counter and 2 floats of scratch on the device")
    _c("adyolo_adam_step_dev", _p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), lr, betas[0], betas[1],
       eps, weight_decay, _p(step_dev), _p(bc_dev), grad_scale, _stream())


def nchw_to_nhwc8(x):
    _chk(x)
    b, c, h, w = x.shape
    y = _new(x, b, h, w, 8)
    _c("adyolo_nchw_to_nhwc8", _p(x), _p(y), b, c, h, w, _stream())
    return y


# ---------------------------------------------------------------------------------------------- conformer pieces
def _kp(kh, kw, c):
    return (kh * kw * c + 3) // 4 * 4


def conv_out_hw(h, w, kh, kw, sh, sw, ph, pw):
    return (h + 2 * ph - kh) // sh + 1, (w + 2 * pw - kw) // sw + 1


# ---------------------------------------------------------------------------------------------- 1-D Winograd F(4, 3) (K9w)
def wino1d_ok(n, h, cin, cout):
    """The 3 x 1 convolution along H (x [N][H][1][Cin]) takes the 1-D Winograd form: whole output tiles, GEMM-friendly channel
    counts and enough rows to fill the chip with the six position GEMMs (``ADYOLO_WINO1D=0`` switches it off)."""
    return (W4_THRESHOLDS["wino1d"] and h % 4 == 0 and cin % 64 == 0 and cout % 64 == 0
            and n * (h // 4) >= 2048)


def wino1d_conv(x3, u, n, h, cin, cout):
    """x3 [N][H][Cin], u [6][Cout][Cin] (``wino1d_filter`` mode 0 / 1) -> (y [N][H][Cout], V [6][N T][Cin])."""
    rows = n * (h // 4)
    v = _new(x3, 6, rows, cin)
    _c("adyolo_wino1d_in", _p(x3), _p(v), n, h, cin, _stream())
    m = _new(x3, 6, rows, cout)
    gemm_batched(v, u, m, rows, cout, cin, cin, cin, cout, False, False, 6, 1, rows * cin, 0, cout * cin, 0, rows * cout, 0)
    y = _new(x3, n, h, cout)
    _c("adyolo_wino1d_out", _p(m), _p(y), n, h, cout, _stream())
    return y, v


def wino1d_filter(w3, cout, cin, mode):
    """w3 [Cout][Cin][3] -> U [6][Cout][Cin] (mode 0) or [6][Cin][Cout] (mode 1: data gradient)."""
    u = _new(w3, 6, cin if mode else cout, cout if mode else cin)
    _c("adyolo_wino1d_filter", _p(w3), _p(u), cout, cin, mode, _stream())
    return u


def wino1d_wgrad(v, dy3, n, h, cin, cout, out=None):
    """v [6][N T][Cin] (the forward's transformed input), dy3 [N][H][Cout] -> dw [Cout][Cin][3] (into ``out`` when given)."""
    rows = n * (h // 4)
    e = _new(dy3, 6, rows, cout)
    _c("adyolo_wino1d_dy", _p(dy3), _p(e), n, h, cout, _stream())
    du = _new(dy3, 6, cout, cin)
    gemm_batched(e, v, du, cout, cin, rows, cout, cin, cin, True, True, 6, 1, rows * cout, 0, rows * cin, 0, cout * cin, 0)
    dw = out if out is not None else _new(dy3, cout, cin, 3)
    _c("adyolo_wino1d_filter", _p(dw), _p(du), cout, cin, 2, _stream())
    return dw


def conv_gemm(mode, src, other, n, h, w, cin, cout, kh, kw, sh, sw, ph, pw):
    """General strided convolution as an implicit GEMM (no column buffer), channels-last.
    mode 0: src x [N][H][W][Cin], other wk = pack_wk(w) -> y [N][Ho][Wo][Cout]
    mode 1: src dy [N][Ho][Wo][Cout], other wkT = pack_wk(w.transpose(0, 1)) -> dx [N][H][W][Cin]
    mode 2: src x, other dy -> dwk [Cout][Kp] (split-K over the output pixels, summed in a fixed order)."""
    _chk(src, other)
    ho, wo = conv_out_hw(h, w, kh, kw, sh, sw, ph, pw)
    splits, slabs = 1, None
    if mode == 0:
        out = _new(src, n, ho, wo, cout)
    elif mode == 1:
        out = _new(src, n, h, w, cin)
    else:
        kp = _kp(kh, kw, cin)
        out = _new(src, cout, kp)
        splits = wgrad_splits(cout, kp, n * ho * wo)
        if splits > 1:
            slabs = _new(src, splits, cout, kp)
    _c("adyolo_conv_gemm", _p(src), _p(other), _p(out), _p(slabs), mode, n, h, w, cin, cout, kh, kw, sh, sw, ph, pw, splits,
       _stream())
    return out


def pack_wk(w):
    """w [Cout][Cin][KH][KW] -> [Cout][Kp] with k = (kh*KW+kw)*Cin + ci."""
    _chk(w)
    cout, cin, kh, kw = w.shape
    wk = _new(w, cout, _kp(kh, kw, cin))
    _c("adyolo_pack_wk", _p(w), _p(wk), cout, cin, kh, kw, 1, _stream())
    return wk


def unpack_wk(wk, cout, cin, kh, kw, out=None):
    w = out if out is not None else _new(wk, cout, cin, kh, kw)
    _c("adyolo_pack_wk", _p(w), _p(wk), cout, cin, kh, kw, 0, _stream())
    return w


def maxpool3_fwd(x):
    _chk(x)
    n, h, w, c = x.shape
    wo = (w + 2 - 3) // 2 + 1
    y = _new(x, n, h, wo, c)
    arg = torch.empty((n, h, wo, c), dtype=torch.uint8, device=x.device)
    _c("adyolo_maxpool3_fwd", _p(x), _p(y), _p(arg), n, h, w, c, _stream())
    return y, arg


def maxpool3_bwd(dy, arg, w):
    _chk(dy)
    n, h, _, c = dy.shape
    dx = _new(dy, n, h, w, c)                   # (gather form: every element is written)
    _c("adyolo_maxpool3_bwd", _p(dy), _p(arg), _p(dx), n, h, w, c, _stream())
    return dx


def affine_relu(x, scale, shift):
    _chk(x, scale, shift)
    c = x.shape[-1]
    y = torch.empty_like(x)
    _c("adyolo_affine_relu_nhwc", _p(x), _p(scale), _p(shift), _p(y), x.numel() // c, c, _stream())
    return y


def relu_bwd(dy, y):
    _chk(dy, y)
    dx = torch.empty_like(y)
    _c("adyolo_relu_bwd", _p(dy), _p(y), _p(dx), y.numel(), _stream())
    return dx


def dropout_axpby(x, z, a, b, p, seed, offset, offset_dev=None):
    """a * dropout(x) + b * z in one pass (z None: a * dropout(x)); the mask of ``dropout_apply(x, p, seed, offset, offset_dev)``."""
    _chk(x, z)
    y = torch.empty_like(x)
    _c("adyolo_dropout_axpby", _p(x), _p(z), _p(y), x.numel(), float(p), ctypes.c_uint64(seed),
       ctypes.c_uint64(offset & 0xFFFFFFFFFFFFFFFF), _p(offset_dev), float(a), float(b), _stream())
    return y


def axpby(x, z, a, b):
    _chk(x, z)
    y = torch.empty_like(x)
    _c("adyolo_axpby", _p(x), _p(z), _p(y), float(a), float(b), x.numel(), _stream())
    return y


def swish_fwd(x):
    _chk(x)
    y = torch.empty_like(x)
    _c("adyolo_swish_fwd", _p(x), _p(y), x.numel(), _stream())
    return y


def swish_bwd(dy, x):
    _chk(dy, x)
    dx = torch.empty_like(x)
    _c("adyolo_swish_bwd", _p(dy), _p(x), _p(dx), x.numel(), _stream())
    return dx


def glu_fwd(x2d):
    _chk(x2d)
    r, c2 = x2d.shape
    y = _new(x2d, r, c2 // 2)
    _c("adyolo_glu_fwd", _p(x2d), _p(y), r, c2 // 2, _stream())
    return y


def glu_bwd(dy2d, x2d):
    _chk(dy2d, x2d)
    dx = torch.empty_like(x2d)
    _c("adyolo_glu_bwd", _p(dy2d), _p(x2d), _p(dx), x2d.shape[0], x2d.shape[1] // 2, _stream())
    return dx


def dwconv3(x, w, bias, dilation, flip=False):
    """x [B][T][C], w [C][3] (or [C][1][3]), bias [C] or None."""
    _chk(x, w, bias)
    b, t, c = x.shape
    y = torch.empty_like(x)
    _c("adyolo_dwconv3_fwd", _p(x), _p(w), _p(bias), _p(y), b, t, c, int(dilation), int(flip), _stream())
    return y


def dwconv3_wgrad(dy, x, dilation, out_dw=None, out_db=None):
    _chk(dy, x)
    b, t, c = x.shape
    dw = out_dw if out_dw is not None else _new(x, c, 3)
    db = out_db if out_db is not None else _new(x, c)
    partial, cws = _new(x, 1024, 4 * c), _new(x, 1024, 3 * c)
    _c("adyolo_dwconv3_wgrad", _p(dy), _p(x), _p(dw), _p(db), _p(partial), _p(cws), b, t, c, int(dilation), _stream())
    return dw, db


def softmax_fwd(s2d, scale):
    _chk(s2d)
    p = torch.empty_like(s2d)
    _c("adyolo_softmax_fwd", _p(s2d), _p(p), s2d.shape[0], s2d.shape[1], float(scale), _stream())
    return p


def softmax_bwd(dp2d, p2d, scale):
    _chk(dp2d, p2d)
    ds = torch.empty_like(p2d)
    _c("adyolo_softmax_bwd", _p(dp2d), _p(p2d), _p(ds), p2d.shape[0], p2d.shape[1], float(scale), _stream())
    return ds


def avgpool1d(x, k, fac):
    _chk(x)
    b, t, c = x.shape
    y = _new(x, b, t // k, c)
    _c("adyolo_avgpool1d_fwd", _p(x), _p(y), b, t, c, k, float(fac), _stream())
    return y


def avgpool1d_bwd(dy, t, k, fac):
    _chk(dy)
    b, _, c = dy.shape
    dx = _new(dy, b, t, c)
    _c("adyolo_avgpool1d_bwd", _p(dy), _p(dx), b, t, c, k, float(fac), _stream())
    return dx


def ln_fwd(x2d, gamma, beta, eps=1e-5):
    _chk(x2d, gamma, beta)
    y = torch.empty_like(x2d)
    _c("adyolo_ln_fwd", _p(x2d), _p(gamma), _p(beta), _p(y), x2d.shape[0], x2d.shape[1], eps, _stream())
    return y


def ln_bwd(dy2d, x2d, gamma, eps=1e-5, acc_dgamma=None, acc_dbeta=None):
    """acc_dgamma / acc_dbeta: the kernel ADDS the two gradients into these (the parameters' slices of the flat gradient
    buffer, zeroed at the start of the step) instead of into fresh zeros."""
    _chk(dy2d, x2d, gamma)
    r, c = x2d.shape
    dx = torch.empty_like(x2d)
    dgamma = acc_dgamma if acc_dgamma is not None else _zeros(x2d, c)
    dbeta = acc_dbeta if acc_dbeta is not None else _zeros(x2d, c)
    partial = _new(x2d, 1024 * 512)
    _c("adyolo_ln_bwd", _p(dy2d), _p(x2d), _p(gamma), _p(dx), _p(dgamma), _p(dbeta), _p(partial), r, c, eps, _stream())
    return dx, dgamma, dbeta


def gemm_batched(a, b, c, m, n, k, lda, ldb, ldc, trans_a, trans_b, outer, inner, oa, ia, ob, ib, oc, ic, alpha=1.0,
                 accumulate=False):
    _c("adyolo_gemm_batched", _p(a), _p(b), _p(c), m, n, k, lda, ldb, ldc, int(trans_a), int(trans_b), outer, inner,
       oa, ia, ob, ib, oc, ic, float(alpha), int(accumulate), _stream())
    return c


# ---------------------------------------------------------------------------------------------- attention (flash style)
def _seed_args(seed):
    """seed: an int, or a one-element int32 tensor on the device holding the 32-bit seed (``seed32_dev``: hipGraph replays)"""
    if isinstance(seed, torch.Tensor):
        return ctypes.c_uint32(0), _p(seed)
    return ctypes.c_uint32(seed & 0xFFFFFFFF), None


def seed32_dev(seed, offset, offset_dev):
    """The value ``rng.DropoutStream.seed32`` computes on the host, computed on the device from (seed, offset + *offset_dev) into a
    one-element int32 tensor: a recorded step replays it with the stream's current offset."""
    out = torch.empty(1, dtype=torch.int32, device=offset_dev.device)
    _c("adyolo_seed32_dev", ctypes.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), ctypes.c_uint64(offset & 0xFFFFFFFFFFFFFFFF), _p(offset_dev),
       _p(out), _stream())
    return out


def attn_fwd(q, k, v, heads, scale, dropout_p=0.0, seed=0, want_lse=True):
    """q, k, v [B][T][heads*64] -> ctx [B][T][heads*64], lse2 [B][heads][T] (or None); scores never reach HBM."""
    _chk(q, k, v)
    b, t, e = q.shape
    ctxv = torch.empty_like(q)
    lse = _new(q, b, heads, t) if want_lse else None
    sv, sp = _seed_args(seed)
    _c("adyolo_attn_fwd", _p(q), _p(k), _p(v), _p(ctxv), _p(lse), b, t, heads, e // heads, float(scale), float(dropout_p),
       sv, sp, _stream())
    return ctxv, lse


def attn_bwd(q, k, v, ctxv, dctx, lse, heads, scale, dropout_p=0.0, seed=0):
    _chk(q, k, v, ctxv, dctx, lse)
    b, t, e = q.shape
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    delta = _new(q, b, heads, t)
    sv, sp = _seed_args(seed)
    _c("adyolo_attn_bwd", _p(q), _p(k), _p(v), _p(ctxv), _p(dctx), _p(lse), _p(delta), _p(dq), _p(dk), _p(dv), b, t, heads,
       e // heads, float(scale), float(dropout_p), sv, sp, _stream())
    return dq, dk, dv


def attn_dropout_mask(like, b, t, heads, dropout_p, seed):
    """The (B, heads, T, T) mask (0 or 1/(1-p)) that attn_fwd / attn_bwd apply for (dropout_p, seed) -- tests only."""
    mask = torch.empty((b, heads, t, t), dtype=torch.float32, device=like.device)
    _c("adyolo_attn_dropout_mask", _p(mask), b, t, heads, float(dropout_p), ctypes.c_uint32(seed & 0xFFFFFFFF), _stream())
    return mask


# ---------------------------------------------------------------------------------------------- input pipeline (8f rows 2-3)
def pcm16_to_f32(pcm, out=None):
    """int16 samples (any shape, contiguous, on the GPU) -> float32 ``x / 32768 + 1e-8`` (datasets.py:105)."""
    if not pcm.is_cuda or pcm.dtype != torch.int16 or not pcm.is_contiguous():
        raise _lib.AdyoloHipError("pcm16_to_f32 needs a contiguous int16 tensor on a HIP device")
    if out is None:
        out = torch.empty(pcm.shape, dtype=torch.float32, device=pcm.device)
    _c("adyolo_pcm16_to_f32", _p(pcm), _p(out), pcm.numel(), _stream())
    return out


def mask_ranges_(feat, ranges):
    """In-place SpecAug masking of feat [B][T][F][C]: ranges int32 [B][4] = {t0, t1, f0, f1} per sample."""
    _chk(feat)
    if ranges.dtype != torch.int32 or not ranges.is_cuda or not ranges.is_contiguous():
        raise _lib.AdyoloHipError("mask_ranges_ needs contiguous int32 ranges on the device")
    b, t, f, c = feat.shape
    _c("adyolo_mask_ranges", _p(feat), _p(ranges), b, t, f, c, _stream())
    return feat


def colstats(a2d):
    """[rows][cols] fp32 -> float64 [4][cols]: column sum, sum of squares, max, min."""
    _chk(a2d)
    rows, cols = a2d.shape
    out = torch.empty((4, cols), dtype=torch.float64, device=a2d.device)
    partial = _new(a2d, 4, 1024, cols)
    _c("adyolo_colstats", _p(a2d), _p(partial), _p(out), rows, cols, _stream())
    return out
