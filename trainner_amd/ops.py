"""Thin Python wrappers over the C ABI: NHWC views, packed weights, per-device workspaces.

Nothing here computes with torch; torch only allocates the buffers the kernels read and write.
"""
import ctypes as C

import os

import torch

from . import hip
from .hip import (ACT_LRELU, ACT_NONE, ACT_RELU, CONV_1x1, CONV_3x3, CONV_3x3_C4, CONV_3x3_UP2, CONV_4x4_S2, CONV_7x7_C4, DGRAD_4x4_S2,  # noqa: F401
                  PACK_C4_DGRAD3, PACK_C4_FWD, PACK_COL_DGRAD3, PACK_COL_FWD, PACK_DENSE_DGRAD, PACK_DGRAD_3x3, PACK_DGRAD_S2, PACK_FWD, PACK_FWD_S2D, CView,
                  ConvDesc,
                  DensePackItem, PackItem, WgradDesc)


def round_up(a, b):
    return (a + b - 1) // b * b


class View:
    """Channels [coff, coff+C) of an fp32 NHWC buffer [N,H,W,Ctot]."""
    __slots__ = ("buf", "coff", "C")

    def __init__(self, buf, coff=0, C=None):
        assert buf.dim() == 4 and buf.dtype == torch.float32 and buf.is_contiguous()
        self.buf, self.coff = buf, coff
        self.C = buf.shape[3] - coff if C is None else C
        assert 0 <= coff and coff + self.C <= buf.shape[3]

    N = property(lambda s: s.buf.shape[0])
    H = property(lambda s: s.buf.shape[1])
    W = property(lambda s: s.buf.shape[2])
    ctot = property(lambda s: s.buf.shape[3])
    pixels = property(lambda s: s.buf.shape[0] * s.buf.shape[1] * s.buf.shape[2])

    def sub(self, off, C):
        return View(self.buf, self.coff + off, C)

    def c(self):
        return CView(self.buf.data_ptr(), self.buf.shape[3], self.coff)

    def dense(self):
        """torch view [N,H,W,C] of this window (debug / tests only)."""
        return self.buf[..., self.coff:self.coff + self.C]


def new_act(N, H, W, C, device):
    """Uninitialised NHWC activation buffer."""
    return torch.empty((N, H, W, C), dtype=torch.float32, device=device)


def cv(v):
    return hip.NULLVIEW if v is None else v.c()


# ----------------------------------------------------------------------------------------------
# workspaces (per device, grown on demand; all kernels of one engine run on one stream)
# ----------------------------------------------------------------------------------------------
class _Workspaces:
    def __init__(self):
        self.bufs = {}

    def get(self, name, nbytes, device):
        key = (name, str(device))
        t = self.bufs.get(key)
        if t is None or t.numel() * 8 < nbytes:
            n = max(int(nbytes * 1.25) // 8 + 1, 1024)
            t = torch.empty(n, dtype=torch.float64, device=device)
            self.bufs[key] = t
        return t


WS = _Workspaces()


# ----------------------------------------------------------------------------------------------
# weight packing
# ----------------------------------------------------------------------------------------------
class Packed:
    """A packed weight slab; `owner` = the packer whose run() rewrites it (owner.gen counts those runs), None for one-off packs."""
    __slots__ = ("t", "KoutP", "KinP", "kind", "owner")

    def __init__(self, t, KoutP, KinP, kind, owner=None):
        self.t, self.KoutP, self.KinP, self.kind, self.owner = t, KoutP, KinP, kind, owner


def pack_dims(Cout, Cin, kh, kw, kind):
    lib = hip.load()
    ko, ki, n = hip.c_i(), hip.c_i(), hip.c_l()
    hip.check(lib.tnr_pack_dims(Cout, Cin, kh, kw, kind, C.byref(ko), C.byref(ki), C.byref(n)), "pack_dims")
    return ko.value, ki.value, n.value


class WeightPacker:
    """One device table of pack jobs for a whole network; `run()` re-lays out every weight with a
    single launch (weights change every optimiser step)."""

    def __init__(self, device):
        self.device = device
        self.jobs = []          # (weight tensor, kind)
        self.packed = []
        self.table = None
        self.max_out = 0
        self.flat = None
        self.gen = 0            # completed run() calls (consumers of derived layouts -- the sweep images -- compare it)

    def add(self, w, kind):
        Cout, Cin, kh, kw = w.shape
        ko, ki, n = pack_dims(Cout, Cin, kh, kw, kind)
        self.jobs.append((w, kind, ko, ki, n))
        self.packed.append(None)
        self.table = None
        return len(self.jobs) - 1

    def _finalize(self):
        self.__dict__.pop("_sweep_images", None)       # derived layouts of the old slabs
        self.__dict__.pop("_sweep_batch", None)
        self.__dict__.pop("_wq_images", None)
        total = sum(round_up(j[4], 64) for j in self.jobs)
        self.flat = torch.empty(total, dtype=torch.float32, device=self.device)
        items = (PackItem * len(self.jobs))()
        off = 0
        for i, (w, kind, ko, ki, n) in enumerate(self.jobs):
            seg = self.flat[off:off + n]
            off += round_up(n, 64)
            Cout, Cin, kh, kw = w.shape
            items[i] = PackItem(w.data_ptr(), seg.data_ptr(), Cout, Cin, kh, kw, kind, ko, ki, n)
            self.packed[i] = Packed(seg, ko, ki, kind, self)
            self.max_out = max(self.max_out, n)
        raw = bytes(items)
        self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)
        self._ptrs = [j[0].data_ptr() for j in self.jobs]

    def get(self, idx):
        if self.table is None:
            self._finalize()
        return self.packed[idx]

    def run(self):
        if not self.jobs:
            return
        if self.table is None or any(j[0].data_ptr() != p for j, p in zip(self.jobs, self._ptrs)):
            self._finalize()
        hip.check(hip.load().tnr_pack_weights(self.table.data_ptr(), len(self.jobs), self.max_out, hip.stream()),
                  "pack_weights")
        self.gen += 1
        _repack_sweep_images(self)


class DensePacker:
    """Packs the 'gradient dense block' slabs (tnr_pack_dense_dgrad) of every dense block of a network
    with one launch.  add() takes the five conv weights of a block and the block's residual scale."""

    def __init__(self, device):
        self.device = device
        self.jobs = []          # (weights[5], t, nf, gc, scale5, ko, ki, n)
        self.packed = []
        self.table = None
        self.max_out = 0
        self.gen = 0

    def add_block(self, weights, nf, gc, scale5):
        """-> list of 5 job indices (t = 0..4)."""
        lib = hip.load()
        idx = []
        for t in range(5):
            ko, ki, n = hip.c_i(), hip.c_i(), hip.c_l()
            hip.check(lib.tnr_pack_dense_dims(nf, gc, t, C.byref(ko), C.byref(ki), C.byref(n)), "pack_dense_dims")
            self.jobs.append((list(weights), t, nf, gc, float(scale5), ko.value, ki.value, n.value))
            self.packed.append(None)
            idx.append(len(self.jobs) - 1)
        self.table = None
        return idx

    def _finalize(self):
        self.__dict__.pop("_sweep_images", None)
        self.__dict__.pop("_sweep_batch", None)
        total = sum(round_up(j[7], 64) for j in self.jobs)
        self.flat = torch.empty(total, dtype=torch.float32, device=self.device)
        items = (DensePackItem * len(self.jobs))()
        off = 0
        for i, (ws, t, nf, gc, sc, ko, ki, n) in enumerate(self.jobs):
            seg = self.flat[off:off + n]
            off += round_up(n, 64)
            it = DensePackItem()
            for k in range(5):
                it.w[k] = ws[k].data_ptr()
            it.wp, it.nf, it.gc, it.t, it.KoutP, it.KinP, it.scale5, it.n_out = seg.data_ptr(), nf, gc, t, ko, ki, sc, n
            items[i] = it
            self.packed[i] = Packed(seg, ko, ki, PACK_DENSE_DGRAD, self)
            self.max_out = max(self.max_out, n)
        self.table = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(self.device)
        self._ptrs = [w.data_ptr() for j in self.jobs for w in j[0]]

    def get(self, idx):
        if self.table is None:
            self._finalize()
        return self.packed[idx]

    def run(self):
        if not self.jobs:
            return
        if self.table is None or [w.data_ptr() for j in self.jobs for w in j[0]] != self._ptrs:
            self._finalize()
        hip.check(hip.load().tnr_pack_dense_dgrad(self.table.data_ptr(), len(self.jobs), self.max_out, hip.stream()),
                  "pack_dense_dgrad")
        self.gen += 1
        _repack_sweep_images(self)


# ----------------------------------------------------------------------------------------------
# convolution
# ----------------------------------------------------------------------------------------------
class ConvProfile:
    """Optional per-launch HIP-event timing of the implicit-GEMM kernels (bench.py's roofline leg).
    Events are recorded on the launch stream around every launch; nothing is synchronised until
    `summary()`."""

    def __init__(self):
        self.records = []      # (family, flops, start_event, end_event)

    def begin(self):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        return ev

    def end(self, family, flops, start, shape=None):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        self.records.append((family, flops, start, ev, shape))

    def summary(self, by_shape=False):
        torch.cuda.synchronize()
        out = {}
        for fam, flops, a, b, shape in self.records:
            key = (fam,) + tuple(shape or ()) if by_shape else fam
            d = out.setdefault(key, {"launches": 0, "flops": 0.0, "ms": 0.0})
            d["launches"] += 1
            d["flops"] += flops
            d["ms"] += a.elapsed_time(b)
        return out


PROFILE = None      # set to a ConvProfile() to time launches
# matrix-core arithmetic of every MFMA launch below.  hip.MMA_F32: v_mfma_f32_32x32x2_f32; hip.MMA_BF16X3: fp32 operands split
# exactly into three bf16 values, six partial products on the bf16 matrix core, fp32 accumulate (fp32-level accuracy at 6/16 of the
# matrix-core cycles; TNR_MMA=bf16x3); hip.MMA_BF16: operands ROUNDED to bf16 (`use_amp: true`, set by BaseModel.setup_amp)
FP32_MMA = {"f32": hip.MMA_F32, "bf16x3": hip.MMA_BF16X3}[os.environ.get("TNR_MMA", "bf16x3").lower()]
MMA = FP32_MMA


def _conv_desc(d, x, wp, y, mode=CONV_3x3, bias=None, act=ACT_NONE, slope=0.2, alpha=1.0, r1=None, r1_ch=None,
               beta1=1.0, r2=None, alpha2=1.0, mask=None, m_lo=0, m_hi=None, m_slope=0.2, reflect=False, noise=None):
    d.x = x.c()
    d.N, d.H, d.W, d.Cin = x.N, x.H, x.W, x.C
    d.wp, d.KinP, d.KoutP = wp.t.data_ptr(), wp.KinP, wp.KoutP
    d.y = y.c()
    d.Ho, d.Wo, d.Cout = y.H, y.W, y.C
    d.mode = mode
    d.bias = hip.ptr(bias)
    d.act, d.slope, d.alpha = act, slope, alpha
    d.r1 = cv(r1)
    d.r1_ch = (r1.C if r1_ch is None else r1_ch) if r1 is not None else 0
    d.beta1 = beta1
    d.r2 = cv(r2)
    d.alpha2 = alpha2
    d.m = cv(mask)
    d.m_lo = m_lo
    d.m_hi = (y.C if m_hi is None else m_hi)
    d.m_slope = m_slope
    d.mma = MMA
    d.pad_mode = 1 if reflect else 0          # TNR_CONV_3x3 only: ReflectionPad2d(1) borders instead of zeros
    if noise is not None:                      # Noise: the ESRGAN+ multiplier of tnr_conv_desc.noise_* (None: the fields stay zero)
        d.noise_sigma, d.noise_pos, d.noise_key0, d.noise_key1, d.noise_pix0 = noise.sigma, noise.pos, noise.key0, noise.key1, noise.pix0


class Noise:
    """ESRGAN+ GaussianNoise multiplier of one dense block (block.py:587-600): m = 1 + sigma * n(key, element).  pos: where a
    convolution epilogue applies it (1 after the r1 step = forward, 2 after the r2 step = backward); pix0: first pixel of this
    rank's shard in the global batch."""
    __slots__ = ("sigma", "pos", "key0", "key1", "pix0")

    def __init__(self, sigma, key, pos=1, pix0=0):
        self.sigma, self.pos, self.pix0 = float(sigma), pos, pix0 & 0xFFFFFFFF
        self.key0, self.key1 = key & 0xFFFFFFFF, (key >> 32) & 0xFFFFFFFF

    def at(self, pos):
        n = Noise(self.sigma, 0, pos, self.pix0)
        n.key0, n.key1 = self.key0, self.key1
        return n


def noise_key(seed, call, block):
    """64-bit key of (seed, training forward `call`, dense block): splitmix64 over the three words, so that neighbouring calls /
    blocks share no structure the device-side counter hash (csrc/gauss_noise.h) would have to undo."""
    M = 0xFFFFFFFFFFFFFFFF

    def mix(z):
        z = (z + 0x9E3779B97F4A7C15) & M
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        return z ^ (z >> 31)

    return mix(mix(mix(seed & M) ^ (call & M)) ^ (block & M))


def gauss_mult(dst, src, noise):
    """dst = (src or 1) * the noise multiplier field (tnr_gauss_mult); dst / src: views of `dst.C` channels."""
    hip.check(hip.load().tnr_gauss_mult(dst.c(), cv(src), dst.pixels, dst.C, noise.sigma, noise.key0, noise.key1, noise.pix0,
                                        hip.stream()), "gauss_mult")


IMAGE_C4 = os.environ.get("TNR_IMAGE_C4", "1") != "0"       # taps-in-K kernel for <= 4-channel image layers (A/B switch)
SMALL_GEMM = os.environ.get("TNR_SMALL_GEMM", "1") != "0"   # im2col + split-K GEMM for <= 4096-pixel layers (A/B switch)


S2_D4 = os.environ.get("TNR_S2_D4", "1") != "0"      # the four-tap layers (4x4 stride 2 and its data-gradient) on the weight-stream machinery too (A/B switch)
X3_D4 = os.environ.get("TNR_X3_D4", "1") != "0"      # TNR_MMA=bf16x3: 64-cout 3x3 layers take their weights as a pre-split stream (A/B switch)
_wq_oneoff = {}


def _wq_image(lib, d, wp, dev, tag=None):
    """The pre-split weight stream of a launch (tnr_conv_desc.wq; None: the launch cannot use one).  Kept on the packer that owns the
    packed weights and rebuilt (one small launch) when that packer has run since -- once per optimiser step, once ever for the VGG.
    tag: a second stream of the same weights in another order (the pixel-shuffle store's) is cached under its own key."""
    need = lib.tnr_conv_wq_bytes(C.byref(d))
    if need <= 0:
        return None
    owner = wp.owner
    cache, gen = (_wq_oneoff, None) if owner is None else (owner.__dict__.setdefault("_wq_images", {}), owner.gen)
    # one-off packs (no owning packer): an image per (packed weights, stream) -- it is re-packed on every call on the CURRENT stream,
    # so two streams must not share one -- and at most 64 of them (least recently used out: the addresses change as tensors come and go)
    key = wp.t.data_ptr() if owner is not None else (wp.t.data_ptr(), hip.stream())
    if tag is not None:
        key = (tag, key)
    ent = cache.get(key)
    if owner is None and ent is not None:
        cache[key] = cache.pop(key)          # most recently used last
    if ent is None or ent[0].numel() * 4 < need:
        ent = cache[key] = [torch.empty(need // 4, dtype=torch.float32, device=dev), None]
        if owner is None:
            while len(cache) > 64:
                cache.pop(next(iter(cache)))
    if gen is None or ent[1] != gen:
        hip.check(lib.tnr_conv_wq_pack(C.byref(d), ent[0].data_ptr(), need, hip.stream()), "conv_wq_pack")
        ent[1] = gen
    return ent[0]


# TNR_MMA=bf16x3: 64-cout 3x3 layers in the Winograd F(2x2, 3x3) form (csrc/conv_wino.hip): 2.25 x fewer matrix-core instructions for the
# same convolution, a few more fp32 roundings per element (error vs fp64 <= 3 x the fp32 matrix core's in the tests; NOT bit-identical
# to the direct kernels).  The transform + operand split of an input chunk is vector-ALU work per PIXEL that only 64 output channels per
# workgroup amortise (registers: 16 transform positions x 2 x 2 accumulator tiles), so the kernel is bound by it, not by the matrix
# core: x 1.19 (128 ch) / 1.26 (256) / 1.23-1.31 (512) over the direct weight-stream kernel, x 1.04-1.11 on the 64-channel layers
# (profiles/r09o_wino_blate_ab.txt; analysis DESIGN.md 3.9).  Dense blocks never use it (their per-layer fallback must stay
# bit-identical to the one-launch forms).  TNR_WINO=0: off.
WINO = os.environ.get("TNR_WINO", "1") != "0"
WINO_MIN_CIN = int(os.environ.get("TNR_WINO_MIN_CIN", "64"))
WINO_MIN_PIXELS = int(os.environ.get("TNR_WINO_MIN_PIXELS", "4096"))


def _wino_image(lib, d, wp, dev):
    """The transform-domain weight stream of a launch (tnr_conv_desc.wq with wq_form = 1; None: the launch cannot run in the Winograd
    form).  Cached and refreshed like _wq_image: per owning packer and its generation, or per (weights, stream) for one-off packs."""
    need = lib.tnr_conv_wino_bytes(C.byref(d))
    if need <= 0:
        return None
    owner = wp.owner
    cache, gen = (_wq_oneoff, None) if owner is None else (owner.__dict__.setdefault("_wq_images", {}), owner.gen)
    key = ("wino", wp.t.data_ptr()) if owner is not None else ("wino", wp.t.data_ptr(), hip.stream())
    ent = cache.get(key)
    if owner is None and ent is not None:
        cache[key] = cache.pop(key)
    if ent is None or ent[0].numel() * 4 < need:
        ent = cache[key] = [torch.empty(need // 4, dtype=torch.float32, device=dev), None]
        if owner is None:
            while len(cache) > 64:
                cache.pop(next(iter(cache)))
    if gen is None or ent[1] != gen:
        hip.check(lib.tnr_conv_wino_pack(C.byref(d), ent[0].data_ptr(), need, hip.stream()), "conv_wino_pack")
        ent[1] = gen
    return ent[0]


SHUFFLE_FOLD = os.environ.get("TNR_SHUFFLE_FOLD", "1") != "0"      # nn.PixelShuffle(2) folded into the convolution's store (A/B switch)


def conv_shuffle2(x, wp, y, **epi):
    """conv (nf -> 4 nf, 3x3) + nn.PixelShuffle(2) + the epilogue's activation in ONE launch (block.pixelshuffle_block, block.py:374-387):
    y is the SHUFFLED tensor [N, 2 H, 2 W, nf]; the [N, H, W, 4 nf] intermediate and the depth-to-space pass do not exist
    (tnr_conv_desc.shuffle).  Returns False when the launch cannot run that way (fp32-matrix-core arithmetic, shapes the weight-stream
    kernel does not take): the caller then runs conv + depth_to_space."""
    if not (SHUFFLE_FOLD and X3_D4 and MMA in (hip.MMA_BF16X3, hip.MMA_BF16) and x.buf.is_cuda):
        return False
    assert y.H == 2 * x.H and y.W == 2 * x.W and wp.KoutP == 4 * y.C
    lib = hip.load()
    d = ConvDesc()
    _conv_desc(d, x, wp, y, CONV_3x3, **epi)
    d.Ho, d.Wo, d.Cout, d.shuffle = x.H, x.W, 4 * y.C, 2
    d.m_hi = 0
    img = _wq_image(lib, d, wp, x.buf.device, tag="shuffle2")
    if img is None:
        return False
    d.wq, d.wq_bytes = img.data_ptr(), img.numel() * 4
    t0 = PROFILE.begin() if PROFILE is not None else None
    hip.check(lib.tnr_conv_forward(C.byref(d), hip.stream()), "conv_forward (pixel-shuffle store)")
    if PROFILE is not None:
        PROFILE.end("conv_tile_3x3", 2.0 * x.pixels * 9 * x.C * 4 * y.C, t0, (x.C, 4 * y.C, x.H, wp.kind))
    return True


def conv(x, wp, y, mode=CONV_3x3, wino=None, **epi):
    """wino: None = the process policy (WINO and the layer's size), True / False = force / forbid the Winograd form for this launch."""
    d = ConvDesc()
    _conv_desc(d, x, wp, y, mode, **epi)
    if y.pixels <= 16384 and wp.KinP >= 512:      # candidates for split-K (tnr_conv_workspace_bytes decides)
        need = hip.load().tnr_conv_workspace_bytes(C.byref(d))
        if need > 0:
            ws = WS.get("splitk@%x" % hip.stream(), need, x.buf.device)
            d.ws, d.ws_bytes = ws.data_ptr(), ws.numel() * 8
    use_wino = (WINO and x.C >= WINO_MIN_CIN and y.pixels >= WINO_MIN_PIXELS) if wino is None else wino
    if use_wino and mode == CONV_3x3 and d.mma == hip.MMA_BF16X3 and not d.ws and y.C % 64 == 0:
        img = _wino_image(hip.load(), d, wp, x.buf.device)
        if img is not None:
            d.wq, d.wq_bytes, d.wq_form = img.data_ptr(), img.numel() * 4, 1
        else:
            assert wino is not True, "this launch cannot run in the Winograd form"
    if not d.wq and X3_D4 and (mode == CONV_3x3 or (S2_D4 and mode in (CONV_4x4_S2, DGRAD_4x4_S2))) and d.mma in (hip.MMA_BF16X3, hip.MMA_BF16) \
            and not d.ws and y.C % 64 == 0:
        img = _wq_image(hip.load(), d, wp, x.buf.device)
        if img is not None:
            d.wq, d.wq_bytes = img.data_ptr(), img.numel() * 4
    if PROFILE is None:
        hip.check(hip.load().tnr_conv_forward(C.byref(d), hip.stream()), "conv_forward")
        return
    t0 = PROFILE.begin()
    hip.check(hip.load().tnr_conv_forward(C.byref(d), hip.stream()), "conv_forward")
    taps = {CONV_3x3: 9, CONV_3x3_UP2: 9, CONV_1x1: 1, CONV_3x3_C4: 9, CONV_7x7_C4: 49}.get(mode, 16)
    opix = y.pixels if mode != DGRAD_4x4_S2 else y.pixels // 4     # each input-grad pixel sees 4 of the 16 taps
    fam = {CONV_3x3: "conv_tile_3x3", CONV_3x3_UP2: "conv_tile_3x3_up2", CONV_4x4_S2: "conv_tile_4x4s2",
           DGRAD_4x4_S2: "conv_tile_dgrad4x4s2", CONV_1x1: "conv_tile_1x1", CONV_3x3_C4: "conv_tile_3x3_c4", CONV_7x7_C4: "conv_tile_7x7_c4"}[mode]
    if d.wq_form == 1:
        fam = "conv_wino_3x3"          # (algorithmic FLOP of the convolution it computes: 9 taps)
    PROFILE.end(fam, 2.0 * opix * taps * min(x.C, wp.KinP) * y.C, t0, (x.C, y.C, y.H, wp.kind))


CHAIN_MAX = 6
CHAIN_X3 = os.environ.get("TNR_CHAIN_X3", "1") == "1"   # TNR_MMA=bf16x3 also inside tnr_conv_chain (A/B switch)
CONV_CHAIN = os.environ.get("TNR_CONV_CHAIN", "1") != "0"     # 0: one launch per layer (A/B switch)
CONV_SWEEP = os.environ.get("TNR_CONV_SWEEP", "1") != "0"     # TNR_MMA=bf16x3: dense blocks through tnr_conv_sweep (0: tnr_conv_chain; A/B switch)
COLLECTIVES_IN_FLIGHT = False   # True (dp.py) from the first gradient bucket handed to RCCL on the side stream until the compute
                                # stream has waited for all of them: a chain launch needs every workgroup of its grid
                                # co-resident, which RCCL kernels sharing the CUs could delay -> one launch per layer meanwhile
# TNR_CHAIN_WITH_COLLECTIVES=1: keep the one-launch forms while gradient buckets are on the wire.  Measured next to a 4 x 64 MB RCCL
# all-reduce on a side stream (1-rank group, tools/chain_stress.py --rccl, profiles/r03j): bit-identical results, no wait ever timed out,
# +0.10 ms per dense-block launch (sweep 0.61 -> 0.71 ms, chain 1.01 -> 1.14 ms).  Off by default: with N > 1 ranks an RCCL kernel waits
# for its peers while it holds CUs the launch wants all of -- never exercised on hardware here.
CHAIN_WITH_COLLECTIVES = os.environ.get("TNR_CHAIN_WITH_COLLECTIVES", "0") == "1"
AMP_SWEEP = os.environ.get("TNR_AMP_SWEEP", "1") != "0"      # use_amp (bf16 operands): dense blocks through the sweep's bf16-operand form (0: tnr_conv_chain; A/B switch)
SWEEP_DISPENSED = os.environ.get("TNR_SWEEP_WAVES", "4") != "8"      # the four-wave forms take their tiles from an atomic counter (conv_sweep.hip)
COUNTERS = {"one_launch_next_to_collectives": 0, "per_layer_next_to_collectives": 0}      # dense blocks launched while gradient buckets were in flight


def dense_blocks_overlap_collectives():
    """True when the dense blocks stay one launch each next to in-flight gradient buckets in the CURRENT arithmetic (the dispensed
    sweep of TNR_MMA_BF16X3, or TNR_CHAIN_WITH_COLLECTIVES=1): the generator's all-reduce can then overlap its backward for free."""
    return CHAIN_WITH_COLLECTIVES or (CONV_CHAIN and CONV_SWEEP and SWEEP_DISPENSED and
                                      ((CHAIN_X3 and MMA == hip.MMA_BF16X3) or (AMP_SWEEP and MMA == hip.MMA_BF16)))


def g_buckets_leave_in_backward():
    """TNR_DP_OVERLAP_G: 0 (default) = the generator's gradient buckets go out at its optimizer step; auto = from inside its backward
    whenever the dense blocks stay one launch next to them; 1 = from inside its backward regardless (per-layer launches where needed)."""
    want = os.environ.get("TNR_DP_OVERLAP_G", "0")
    return want == "1" or (want == "auto" and dense_blocks_overlap_collectives())


# every dense block's sweep image rebuilt in ONE launch per packer run (tnr_conv_sweep_pack_batch: 2 launches per optimiser step instead of 138).
# OFF by default: measured 0.6-2.1 ms per step SLOWER than the per-block 5 us launches, which sit in launch gaps the stream pays anyway
# (profiles/r10g_sweep_pack_batch_ab.txt); bit-identical either way.
SWEEP_PACK_BATCH = os.environ.get("TNR_SWEEP_PACK_BATCH", "0") != "0"


def _repack_sweep_images(owner):
    """After `owner` (a WeightPacker / DensePacker) re-packed its weights: rebuild the sweep images of ALL its dense blocks in one launch
    (tnr_conv_sweep_pack_batch) instead of one small launch per block at its first use -- 2 launches per optimiser step instead of
    2 x 69 for RRDBNet-23.  The batch is the set of images _sweep_image has built for this owner so far."""
    b = owner.__dict__.get("_sweep_batch")
    if not b or not SWEEP_PACK_BATCH:
        return
    if b["table"] is None:
        raw = b"".join(bytes(it) for it in b["items"])
        b["table"] = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(b["ents"][0][0].device)
    hip.check(hip.load().tnr_conv_sweep_pack_batch(b["table"].data_ptr(), len(b["items"]), b["max_units"], hip.stream()), "conv_sweep_pack_batch")
    for ent in b["ents"]:
        ent[1] = owner.gen


_chain_epoch = {}
_sweep_images = {}              # sweep images of one-off packs (no owning packer): (packed-weight pointers) -> [image, None]


def _sweep_image(lib, descs, n, stages, dev):
    """The pre-split weight stream of a dense block for tnr_conv_sweep (None: not sweepable).  Kept on the packer that owns the
    block's packed weights and rebuilt (one small launch) when that packer has run since -- once per optimiser step."""
    need = lib.tnr_conv_sweep_image_bytes(descs, n)
    if need <= 0:
        return None
    owner = stages[0]["wp"].owner
    if owner is None or any(st["wp"].owner is not owner for st in stages):
        owner = None
        cache, gen = _sweep_images, None               # one-off packs: rebuilt on every call
    else:
        cache = owner.__dict__.setdefault("_sweep_images", {})
        gen = owner.gen
    key = tuple(st["wp"].t.data_ptr() for st in stages) + (() if owner is not None else (hip.stream(),))
    ent = cache.get(key)
    if owner is None and ent is not None:
        cache[key] = cache.pop(key)          # one-off packs: per stream, least recently used out (see _wq_image)
    if ent is None or ent[0].numel() * 4 < need:
        ent = cache[key] = [torch.empty(need // 4, dtype=torch.float32, device=dev), None]
        if owner is None:
            while len(cache) > 64:
                cache.pop(next(iter(cache)))
    if gen is None or ent[1] != gen:
        hip.check(lib.tnr_conv_sweep_pack(descs, n, ent[0].data_ptr(), need, hip.stream()), "conv_sweep_pack")
        if owner is not None and SWEEP_PACK_BATCH and ent[1] is None:
            # first build of this block's image: from the owner's next run() on it is rebuilt with all the others in one launch
            item = hip.SweepPackItem()
            hip.check(lib.tnr_conv_sweep_pack_item(descs, n, ent[0].data_ptr(), need, C.byref(item)), "conv_sweep_pack_item")
            b = owner.__dict__.setdefault("_sweep_batch", {"items": [], "ents": [], "table": None, "max_units": 0})
            b["items"].append(item)
            b["ents"].append(ent)
            b["table"] = None
            b["max_units"] = max(b["max_units"], item.units)
        ent[1] = gen
    return ent[0]


# Per-box choice between the one-launch dense block and five per-layer launches in TNR_MMA_BF16X3 (bit-identical results either way).
# The sweep hands tiles over between workgroups through system-coherent stores / loads and agent-scope progress words: fabric traffic
# that never touches L2.  On two of ~25 boxes met in round 5 that path was slow -- the sweep alone ran 1.6 x slower (935-941 vs 587-606 us
# per launch at batch 16), every other kernel at its usual rate -- while the per-layer path (plain loads and stores,
# 686 us on a normal box) does not use it.  The choice is made EXPLICITLY, once, when a training model is set up
# (`calibrate_dense_block_form`, called by SRModel with a scratch block of the TRAINING shape -- never from inside a forward, whose first
# full-size call could be a validation or tiled-inference shape): both forms are timed (3 launches each, ~10 ms) and the per-layer path
# is taken if the sweep is more than 10 % SLOWER than it.  Data-parallel ranks agree on ONE form (all-reduce MAX: if any rank's sweep is
# slow every rank runs per-layer launches -- a rank that kept the slow sweep would be the step's straggler) and every rank's own
# measurement is kept in SWEEP_AUTO_STATE["per_rank"] (bench.py prints it).  Not calibrated (choice None): the sweep.  TNR_SWEEP_AUTO=0: off.
SWEEP_AUTO = os.environ.get("TNR_SWEEP_AUTO", "1") != "0"
SWEEP_AUTO_STATE = {"choice": None, "sweep_us": None, "layers_us": None}     # choice: None (not calibrated) | "sweep" | "layers"


def _stage_views(st):
    return [st[k] for k in ("x", "y", "r1", "r2", "mask") if st.get(k) is not None]


def _calibrate_dense_block(stages):
    """Time the one-launch form and the per-layer form of `stages` on the current stream and record this process's own choice.  The block
    is executed several times, which is only idempotent when no stage writes what another launch of the block reads: the LAST stage's
    output must not share a buffer with any input (the four inner stages write channel groups of the dense buffer that the block itself
    produces) -- asserted, not assumed."""
    global PROFILE
    out = stages[-1]["y"]
    for st in stages:
        for v in _stage_views(st):
            if v is not out and v.buf.data_ptr() == out.buf.data_ptr():
                lo, hi = max(v.coff, out.coff), min(v.coff + v.C, out.coff + out.C)
                assert hi <= lo, "calibration re-executes the block: its output must not alias an input (channels [%d, %d))" % (lo, hi)
    prof, PROFILE = PROFILE, None
    prev = SWEEP_AUTO_STATE["choice"]
    try:
        def timed(fn):
            fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(3):
                fn()
            b.record()
            torch.cuda.synchronize()
            return 1e3 * a.elapsed_time(b) / 3.0

        def layers():
            for st in stages:
                conv(wino=False, **{k: v for k, v in st.items() if k != "fresh_from"})      # (the direct kernels: bit-identical to the one-launch forms)

        SWEEP_AUTO_STATE["choice"] = "sweep"                # (the timed calls below go through conv_chain itself)
        t_sweep, t_layers = timed(lambda: conv_chain(stages)), timed(layers)
        t_sweep = min(t_sweep, timed(lambda: conv_chain(stages)))          # (the clock may still be ramping at a process's first launches)
        prev = "layers" if t_sweep > 1.10 * t_layers else "sweep"
        SWEEP_AUTO_STATE.update(sweep_us=round(t_sweep, 1), layers_us=round(t_layers, 1))
    finally:
        SWEEP_AUTO_STATE["choice"] = prev
        PROFILE = prof


def dense_block_form_applies(stages):
    return (SWEEP_AUTO and CONV_CHAIN and CONV_SWEEP and CHAIN_X3 and len(stages) == 5 and FP32_MMA == hip.MMA_BF16X3 and
            stages[0]["x"].buf.is_cuda and
            all(st.get("mode", CONV_3x3) == CONV_3x3 and st["y"].C % 32 == 0 and st["wp"].KoutP == st["y"].C for st in stages))


def calibrate_dense_block_form(stages, dp=None):
    """The explicit per-box calibration (see above).  stages: a dense block over SCRATCH buffers of the training shape (None: nothing to
    time here -- a CPU stand-in run -- but the ranks still exchange their records).  dp: the
    data-parallel group -- every rank must call this at the same point (two small collectives).  Returns SWEEP_AUTO_STATE."""
    applies = bool(stages) and dense_block_form_applies(stages)
    if applies:
        prev, globals()["MMA"] = MMA, FP32_MMA               # (the fp32 arithmetic of the step, also when called inside an amp region)
        try:
            _calibrate_dense_block(stages)
        finally:
            globals()["MMA"] = prev
    SWEEP_AUTO_STATE["own_choice"] = SWEEP_AUTO_STATE["choice"]
    if dp is not None and dp.active:
        import torch.distributed as dist
        dev = stages[0]["x"].buf.device if (stages and stages[0]["x"].buf.is_cuda and dist.get_backend(dp.group) == "nccl") else torch.device("cpu")
        mine = torch.tensor([1.0 if SWEEP_AUTO_STATE["choice"] == "layers" else 0.0, float(SWEEP_AUTO_STATE["sweep_us"] or 0.0),
                             float(SWEEP_AUTO_STATE["layers_us"] or 0.0), 1.0 if applies else 0.0], dtype=torch.float32, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(dp.world_size)]
        dist.all_gather(allr, mine, group=dp.group)
        rows = [t.cpu().tolist() for t in allr]
        SWEEP_AUTO_STATE["per_rank"] = [{"rank": r, "choice": (("layers" if v[0] else "sweep") if v[3] else None),
                                         "sweep_us": round(v[1], 1) or None, "layers_us": round(v[2], 1) or None} for r, v in enumerate(rows)]
        if any(v[3] for v in rows):
            SWEEP_AUTO_STATE["choice"] = "layers" if any(v[0] and v[3] for v in rows) else "sweep"       # all-reduce MAX, by hand
    if SWEEP_AUTO_STATE["choice"] is not None:
        import logging
        logging.getLogger("base").info("dense-block form: %s (this rank: sweep %s us, per-layer %s us%s)", SWEEP_AUTO_STATE["choice"],
                                       SWEEP_AUTO_STATE["sweep_us"], SWEEP_AUTO_STATE["layers_us"],
                                       "; ranks: %s" % SWEEP_AUTO_STATE["per_rank"] if "per_rank" in SWEEP_AUTO_STATE else "")
    return SWEEP_AUTO_STATE


def conv_chain(stages):
    """Dependent 3x3 convolutions over one pixel grid in one launch (tnr_conv_chain).  stages: dicts with the
    arguments of conv() (x, wp, y, bias, act, ..., mask) plus fresh_from: first input channel produced by the
    previous stage of this chain (None for the first stage).  Same results as calling conv() per stage.
    In TNR_MMA_BF16X3 a residual dense block (or its gradient mirror) goes through tnr_conv_sweep instead: every input channel
    chunk read once per phase for all the stages that consume it, weights streamed pre-split (csrc/conv_sweep.hip)."""
    n = len(stages)
    assert 1 <= n <= CHAIN_MAX
    eligible = all(st.get("mode", CONV_3x3) == CONV_3x3 and st["y"].C % 32 == 0 and st["wp"].KoutP == st["y"].C for st in stages)
    # gradient buckets on the wire: tnr_conv_chain (and the eight-wave sweep) need their whole grid co-resident, which RCCL's
    # kernels on the same CUs could delay -> one launch per layer meanwhile.  The four-wave sweep DISPENSES its tiles in order (a
    # waited-for tile was either taken by a running workgroup or is the next to be dispensed).  Stage s of tile T reads stage s - 1 of
    # T + tiles_x + 1, so T finishes its five stages once tiles up to T + 4 (tiles_x + 1) are held: 4 (tiles_x + 1) + 1 resident
    # workgroups guarantee progress (21 on the 128-wide trunk, never more than the tiles of one image): it may
    # stay one launch -- opt-in (TNR_DP_OVERLAP_G=1, models/sr_model.py) until a multi-GPU run has exercised it.
    crowded = COLLECTIVES_IN_FLIGHT and not CHAIN_WITH_COLLECTIVES
    sweep_ok = CONV_SWEEP and n == 5 and ((MMA == hip.MMA_BF16X3 and CHAIN_X3) or (MMA == hip.MMA_BF16 and AMP_SWEEP))
    auto = SWEEP_AUTO and CONV_CHAIN and eligible and sweep_ok and MMA == hip.MMA_BF16X3 and stages[0]["x"].buf.is_cuda
    if auto and SWEEP_AUTO_STATE["choice"] == "layers":      # (calibrate_dense_block_form chose it at model set-up: never timed here)
        for st in stages:
            conv(wino=False, **{k: v for k, v in st.items() if k != "fresh_from"})      # (the direct kernels: bit-identical to the one-launch forms)
        return
    if not CONV_CHAIN or not eligible or (crowded and not (sweep_ok and SWEEP_DISPENSED)):
        if crowded and CONV_CHAIN and eligible:
            COUNTERS["per_layer_next_to_collectives"] += 1
        for st in stages:
            conv(wino=False, **{k: v for k, v in st.items() if k != "fresh_from"})      # (the direct kernels: bit-identical to the one-launch forms)
        return
    lib = hip.load()
    descs = (ConvDesc * n)()
    fresh = (C.c_int32 * n)()
    flops = 0.0
    for i, st in enumerate(stages):
        kw = {k: v for k, v in st.items() if k != "fresh_from"}
        _conv_desc(descs[i], **kw)
        if descs[i].mma == hip.MMA_BF16X3 and not CHAIN_X3:
            descs[i].mma = hip.MMA_F32
        ff = st.get("fresh_from")
        fresh[i] = -1 if ff is None else ff
        flops += 2.0 * st["y"].pixels * 9 * min(st["x"].C, st["wp"].KinP) * st["y"].C
    dev = stages[0]["x"].buf.device
    fault_word(dev)
    need = lib.tnr_conv_chain_workspace_bytes(C.byref(descs[0]))
    key = ("chain", str(dev), hip.stream(), need)    # one counter set per (stream, tile grid)
    ws = WS.bufs.get(key)
    if ws is None:
        ws = torch.zeros(need // 4, dtype=torch.int32, device=dev)     # progress counters start at 0; last word = error flag
        WS.bufs[key] = ws
        _chain_epoch[key] = 0
    _chain_epoch[key] += 1
    if _chain_epoch[key] > 0x0FFFFFF0:                 # epoch wrap: stale counters would compare as already satisfied
        fill(ws[:-1].view(torch.float32), 0.0)         # (stream-ordered after every earlier launch on this stream)
        _chain_epoch[key] = 1
    image = None
    if CONV_SWEEP and n == 5 and (descs[0].mma == hip.MMA_BF16X3 or (descs[0].mma == hip.MMA_BF16 and AMP_SWEEP)):
        image = _sweep_image(lib, descs, n, stages, dev)
    if crowded and image is None:                      # (a 5-stage block the sweep does not cover: shapes, tiles per image)
        COUNTERS["per_layer_next_to_collectives"] += 1
        for st in stages:
            conv(wino=False, **{k: v for k, v in st.items() if k != "fresh_from"})      # (the direct kernels: bit-identical to the one-launch forms)
        return
    if COLLECTIVES_IN_FLIGHT:
        COUNTERS["one_launch_next_to_collectives"] += 1
    t0 = PROFILE.begin() if PROFILE is not None else None
    if image is not None:
        hip.check(lib.tnr_conv_sweep(descs, n, image.data_ptr(), ws.data_ptr(), ws.numel() * 4, _chain_epoch[key], hip.stream()), "conv_sweep")
    else:
        hip.check(lib.tnr_conv_chain(descs, fresh, n, ws.data_ptr(), ws.numel() * 4, _chain_epoch[key], hip.stream()), "conv_chain")
    if PROFILE is not None:
        x0, yl = stages[0]["x"], stages[-1]["y"]
        PROFILE.end("conv_chain", flops, t0, (x0.C, yl.C, yl.H, stages[0]["wp"].kind))


_FAULT = None


def fault_word(device=None):
    """The engine's fault latch: one zero-initialised device word, registered with the library (tnr_set_fault_word) before the first
    one-launch dense block runs.  A bounded tile hand-off wait that gives up sets it; it is never cleared.  adam_step() launches
    behind it (a faulted step is not applied) and check_engine_errors() raises on it at the next host synchronisation."""
    global _FAULT
    if _FAULT is None:
        _FAULT = torch.zeros(1, dtype=torch.int32, device=device if device is not None else hip.engine_device())
        hip.check(hip.load().tnr_set_fault_word(_FAULT.data_ptr()), "set_fault_word")
    return _FAULT


def chain_error_flag():
    """Nonzero if a tile hand-off wait of tnr_conv_chain / tnr_conv_sweep ever gave up in this process (one host sync)."""
    bad = 0 if _FAULT is None else int(_FAULT.item() != 0)
    for key, ws in WS.bufs.items():       # (workspace tail words: only written while no latch was registered)
        if isinstance(key, tuple) and key[0] == "chain":
            bad += int(ws[-1].item() != 0)
    return bad


def conv_thin(x, w, y, bias=None, alpha=1.0, dgrad=False):
    """3x3 s1 p1 convolution with <= 4 output channels on the vector ALUs (tnr_conv_thin).  w is the layer's OIHW
    weight; dgrad=True computes the data-gradient of a layer with <= 4 INPUT channels (x = gradient of its output).
    The [tap][channel][4] weight layout is rebuilt on every call (a 2 304-element launch) so it is never stale."""
    lib = hip.load()
    Cout, Cin = w.shape[0], w.shape[1]
    red = Cout if dgrad else Cin
    n = lib.tnr_conv_thin_pack_floats(red)
    wp = WS.get("thin_w@%x" % hip.stream(), n * 4, x.buf.device)
    hip.check(lib.tnr_conv_thin_pack(w.data_ptr(), wp.data_ptr(), Cout, Cin, int(dgrad), hip.stream()), "conv_thin_pack")
    t0 = PROFILE.begin() if PROFILE is not None else None
    hip.check(lib.tnr_conv_thin(x.c(), x.N, x.H, x.W, x.C, wp.data_ptr(), y.c(), y.C, hip.ptr(bias), alpha, hip.stream()), "conv_thin")
    if PROFILE is not None:
        PROFILE.end("conv_thin", 2.0 * y.pixels * 9 * x.C * y.C, t0, (x.C, y.C, y.H, int(dgrad)))


def conv_thin7(x, w, y, pad=3, reflect=True, bias=None, alpha=1.0, dgrad=False):
    """7x7 stride-1 convolution with <= 4 output channels in one launch (tnr_conv_thin7): y[p] = sum_t w[t] x[p + t - pad] over y's own
    grid; reflect: ReflectionPad2d(pad) borders (y the size of x), else zeros outside x.  dgrad=True: the data-gradient of a layer with
    <= 4 INPUT channels (x = gradient of its output; pad = 6 and a (H + 6) x (W + 6) y give the gradient of its reflection-padded input)."""
    lib = hip.load()
    Cout, Cin = w.shape[0], w.shape[1]
    n = lib.tnr_conv_thin7_pack_floats(Cout if dgrad else Cin)
    wp = WS.get("thin7_w@%x" % hip.stream(), n * 4, x.buf.device)
    hip.check(lib.tnr_conv_thin7_pack(w.data_ptr(), wp.data_ptr(), Cout, Cin, int(dgrad), hip.stream()), "conv_thin7_pack")
    t0 = PROFILE.begin() if PROFILE is not None else None
    hip.check(lib.tnr_conv_thin7(x.c(), x.N, x.H, x.W, x.C, wp.data_ptr(), y.c(), y.H, y.W, y.C, pad, int(reflect), hip.ptr(bias), alpha,
                                 hip.stream()), "conv_thin7")
    if PROFILE is not None:
        PROFILE.end("conv_thin", 2.0 * y.pixels * 49 * x.C * y.C, t0, (x.C, y.C, y.H, 70 + int(dgrad)))


def wgrad_thin7(big, small4, dw, db, flip, rpad=0, off=0, alpha=1.0, beta=1.0):
    """Weight gradient of a 7x7 layer with <= 3 channels on one side in one launch (tnr_wgrad_thin7):
    flip=False: an image -> C layer (big = gradient of its output, small4 = its reflection-padded NHWC4 input, off = 0);
    flip=True: a C -> image layer (big = its input, read through ReflectionPad2d(rpad = 3); small4 = NHWC4 gradient of its output, off = -6)."""
    lib = hip.load()
    cs = dw.shape[0] if flip else dw.shape[1]
    need = lib.tnr_wgrad_thin7_workspace_bytes(big.N, big.H + 2 * rpad, big.C)
    ws = WS.get("wgrad_thin7@%x" % hip.stream(), need, big.buf.device)
    t0 = PROFILE.begin() if PROFILE is not None else None
    hip.check(lib.tnr_wgrad_thin7(big.c(), big.N, big.H, big.W, rpad, small4.c(), small4.H, small4.W, off, big.C, cs, int(flip),
                                  dw.data_ptr(), hip.ptr(db), alpha, beta, ws.data_ptr(), ws.numel() * 8, hip.stream()), "wgrad_thin7")
    if PROFILE is not None:
        PROFILE.end("wgrad_thin", 2.0 * big.N * (big.H + 2 * rpad) * (big.W + 2 * rpad) * 49 * big.C * cs, t0, (big.C, cs, big.H, 70 + int(flip)))


def small_gemm_ok(x, y, k, stride, epi):
    """A k x k convolution runs as im2col + split-K GEMM when it has <= 4096 output pixels (tiles of the direct
    kernel would be mostly padding and too few to fill the chip) and a plain epilogue."""
    if not SMALL_GEMM or y.pixels > 4096 or (y.pixels % 8) != 0 or (x.C % 4) != 0:
        return False
    return not any(epi.get(n) is not None for n in ("r1", "r2", "mask")) and not epi.get("reflect")


def conv_small(x, wp_col, y, k, stride, pad=1, **epi):
    """y = conv_kxk(x) through the patch matrix: tnr_im2col, then TNR_CONV_1x1 over [pixels][k*k*C] (split-K)."""
    M, K = y.pixels, k * k * x.C
    dev = x.buf.device
    col = WS.get("im2col@%x" % hip.stream(), M * K * 4, dev)
    hip.check(hip.load().tnr_im2col(x.c(), x.N, x.H, x.W, x.C, k, stride, pad, y.H, y.W, col.data_ptr(), hip.stream()), "im2col")
    wimg = 32 if M % 32 == 0 else (16 if M % 16 == 0 else 8)
    cimg = col.view(torch.float32)[:M * K].view(1, M // wimg, wimg, K)
    yimg = View(y.buf.view(1, M // wimg, wimg, y.ctot), y.coff, y.C)
    conv(View(cimg), wp_col, yimg, mode=CONV_1x1, **epi)


def conv_col(x, wp_col, y, k, stride=1, pad=1, **epi):
    """y = conv_kxk(x) (TNR_PACK_COL_FWD weights) or the stride-1 data-gradient conv_kxk(g, pad = k - 1 - p) with flipped /
    transposed weights (TNR_PACK_COL_DGRAD3) for kernel geometries without a direct MFMA tile (the PatchGAN's 4x4 stride-1
    layers): tnr_im2col into [N, Ho, Wo, k*k*C], then the 1x1 implicit-GEMM kernel over that image."""
    K = k * k * x.C
    if K % 16 or K != wp_col.KinP:
        raise hip.HipEngineError("conv_col: k*k*C = %d must be a multiple of 16 and match the packed weights (%d)" % (K, wp_col.KinP))
    M = y.pixels
    col = WS.get("im2col@%x" % hip.stream(), M * K * 4, x.buf.device)
    hip.check(hip.load().tnr_im2col(x.c(), x.N, x.H, x.W, x.C, k, stride, pad, y.H, y.W, col.data_ptr(), hip.stream()), "im2col")
    conv(View(col.view(torch.float32)[:M * K].view(y.N, y.H, y.W, K)), wp_col, y, mode=CONV_1x1, **epi)


WGRAD_GROUP_MAX = 12
WGRAD_PAIR = os.environ.get("TNR_WGRAD_PAIR", "1") != "0"   # dense blocks: conv4+conv3 and conv2+conv1 as 64-cout pairs (A/B switch)
WGRAD_X3 = os.environ.get("TNR_WGRAD_X3", "1") == "1"   # TNR_MMA=bf16x3 also in the weight-gradient kernel (A/B switch)


def _wgrad_desc(d, x, g, dw, db, mode, cin_begin, alpha, beta, reflect=False, pair=None):
    d.x = x.c()
    d.N, d.H, d.W, d.Cin = x.N, x.H, x.W, x.C
    d.g = g.c()
    d.Ho, d.Wo, d.Cout = g.H, g.W, g.C
    d.mode = mode
    d.dw, d.cin_total, d.cin_begin = dw.data_ptr(), dw.shape[1], cin_begin
    d.db = hip.ptr(db)
    d.alpha, d.beta = alpha, beta
    d.mma = hip.MMA_F32 if (MMA == hip.MMA_BF16X3 and not WGRAD_X3) else MMA
    d.pad_mode = 1 if reflect else 0
    if pair is not None:           # cout pair: g covers two layers' gradients, channels >= split belong to (dw2, db2)
        dw2, db2, split = pair
        d.dw2, d.cin_total2, d.cout_split, d.db2 = dw2.data_ptr(), dw2.shape[1], split, hip.ptr(db2)


def wgrad(x, g, dw, db=None, mode=CONV_3x3, cin_begin=0, alpha=1.0, beta=1.0, reflect=False):
    """dw (OIHW, full tensor) += alpha * sum g (x) x over input channels [cin_begin, cin_begin+x.C)."""
    wgrad_group([dict(x=x, g=g, dw=dw, db=db, cin_begin=cin_begin, alpha=alpha, beta=beta, reflect=reflect)], mode=mode)


def wgrad_group(items, mode=CONV_3x3):
    """Several weight gradients of one pixel geometry and one workgroup tile class in a single launch
    (tnr_conv_wgrad_group).  items: dicts with x, g, dw and optional db, cin_begin, alpha, beta, pair = (dw2, db2, cout_split)."""
    lib = hip.load()
    n = len(items)
    assert 1 <= n <= WGRAD_GROUP_MAX
    descs = (WgradDesc * n)()
    for d, it in zip(descs, items):
        _wgrad_desc(d, it["x"], it["g"], it["dw"], it.get("db"), mode, it.get("cin_begin", 0),
                    it.get("alpha", 1.0), it.get("beta", 1.0), it.get("reflect", False), it.get("pair"))
    dev = items[0]["x"].buf.device
    for i, d in enumerate(descs):
        need = lib.tnr_wgrad_workspace_bytes(C.byref(d))
        ws = WS.get("wgrad%d" % i, need, dev)
        d.ws, d.ws_bytes = ws.data_ptr(), ws.numel() * 8
    if PROFILE is None:
        hip.check(lib.tnr_conv_wgrad_group(descs, n, hip.stream()), "conv_wgrad_group")
        return
    t0 = PROFILE.begin()
    hip.check(lib.tnr_conv_wgrad_group(descs, n, hip.stream()), "conv_wgrad_group")
    taps = 16 if mode == CONV_4x4_S2 else 9
    x0, g0 = items[0]["x"], items[0]["g"]
    flops = sum(2.0 * it["g"].pixels * taps * it["x"].C * it["g"].C for it in items)
    PROFILE.end("wgrad_tile", flops, t0, (sum(it["x"].C for it in items) if n > 1 else x0.C, g0.C, g0.H, mode + 100 * (n - 1)))


def wgrad_thin(big, small4, dw, db, flip, alpha=1.0, beta=1.0):
    """Weight gradient of a 3x3 layer with <= 3 channels on one side on the vector ALUs (tnr_wgrad_thin): flip=False
    for a 3 -> C layer (big = gradient of its output, small4 = NHWC4 input image), flip=True for a C -> 3 layer
    (big = its input, small4 = NHWC4 gradient of its output)."""
    lib = hip.load()
    cs = dw.shape[0] if flip else dw.shape[1]
    need = lib.tnr_wgrad_thin_workspace_bytes(big.N, big.H, big.C)
    ws = WS.get("wgrad_thin@%x" % hip.stream(), need, big.buf.device)
    t0 = PROFILE.begin() if PROFILE is not None else None
    hip.check(lib.tnr_wgrad_thin(big.c(), small4.c(), big.N, big.H, big.W, big.C, cs, int(flip), dw.data_ptr(), hip.ptr(db),
                                 alpha, beta, ws.data_ptr(), ws.numel() * 8, hip.stream()), "wgrad_thin")
    if PROFILE is not None:
        PROFILE.end("wgrad_thin", 2.0 * big.pixels * 9 * big.C * cs, t0, (big.C, cs, big.H, int(flip)))


# ----------------------------------------------------------------------------------------------
# layout / resampling / elementwise
# ----------------------------------------------------------------------------------------------
def nchw_to_nhwc(src, dst, Cpad=None, scale=None, shift=None):
    N, Cc, H, W = src.shape
    assert src.is_contiguous() and src.dtype == torch.float32
    hip.check(hip.load().tnr_nchw_to_nhwc(src.data_ptr(), N, Cc, H, W, dst.c(), dst.C if Cpad is None else Cpad,
                                          hip.ptr(scale), hip.ptr(shift), hip.stream()), "nchw_to_nhwc")


def nhwc_to_nchw(src, dst, scale=None, accumulate=False):
    N, Cc, H, W = dst.shape
    assert dst.is_contiguous() and dst.dtype == torch.float32
    hip.check(hip.load().tnr_nhwc_to_nchw(src.c(), N, Cc, H, W, dst.data_ptr(), hip.ptr(scale), int(accumulate),
                                          hip.stream()), "nhwc_to_nchw")


def upsample2x_bwd(gup, gx, mask=None, mslope=0.2):
    hip.check(hip.load().tnr_upsample2x_bwd(gup.c(), gx.c(), gx.N, gx.H, gx.W, gx.C, cv(mask), mslope, hip.stream()),
              "upsample2x_bwd")


def depth_to_space(x, y):
    hip.check(hip.load().tnr_depth_to_space(x.c(), y.c(), x.N, x.H, x.W, y.C, hip.stream()), "depth_to_space")


def space_to_depth_bwd(gy, gx, mask=None, mslope=0.2):
    hip.check(hip.load().tnr_space_to_depth_bwd(gy.c(), gx.c(), gx.N, gx.H, gx.W, gy.C, cv(mask), mslope, hip.stream()),
              "space_to_depth_bwd")


def maxpool2_fwd(x, y):
    hip.check(hip.load().tnr_maxpool2_fwd(x.c(), y.c(), x.N, x.H, x.W, x.C, hip.stream()), "maxpool2_fwd")


def maxpool2_bwd(gy, x, gx):
    hip.check(hip.load().tnr_maxpool2_bwd(gy.c(), x.c(), gx.c(), x.N, x.H, x.W, x.C, hip.stream()), "maxpool2_bwd")


def bilinear2x_fwd(x, y):
    """y [N,2H,2W,C] = bilinear x2 of x [N,H,W,C], align_corners=False."""
    hip.check(hip.load().tnr_bilinear2x_fwd(x.c(), y.c(), x.N, x.H, x.W, x.C, hip.stream()), "bilinear2x_fwd")


def bilinear2x_bwd(gy, gx=None, gz=None, mask=None, mslope=0.2):
    """adjoint of bilinear2x_fwd: gx = plain gradient and / or gz = gradient * LeakyReLU'(mask)."""
    ref = gx if gx is not None else gz
    hip.check(hip.load().tnr_bilinear2x_bwd(gy.c(), cv(gx), cv(gz), cv(mask), mslope, ref.N, ref.H, ref.W, ref.C, hip.stream()),
              "bilinear2x_bwd")


def add2(dst, a, b):
    hip.check(hip.load().tnr_add2(dst.c(), a.c(), b.c(), dst.pixels, dst.C, hip.stream()), "add2")


def mask_copy(dst, src, y, mslope=0.2):
    hip.check(hip.load().tnr_mask_copy(dst.c(), src.c(), y.c(), dst.pixels, dst.C, mslope, hip.stream()), "mask_copy")


# ----------------------------------------------------------------------------------------------
# image-to-image family: generic convolution (vector ALUs), padding helpers, tanh, GAN loss
# ----------------------------------------------------------------------------------------------
def gconv_fwd(x, w, y, bias=None, stride=1, pad=0, reflect=False, act=ACT_NONE, slope=0.2):
    """y = act(conv(x, w) + bias): any square kernel / stride, zero or reflection padding (tnr_gconv_fwd); w OIHW."""
    Cout, Cin, k, _ = w.shape
    t0 = PROFILE.begin() if PROFILE is not None else None
    hip.check(hip.load().tnr_gconv_fwd(x.c(), x.N, x.H, x.W, Cin, w.data_ptr(), hip.ptr(bias), y.c(), y.H, y.W, Cout, k, stride, pad,
                                       int(reflect), act, slope, hip.stream()), "gconv_fwd")
    if PROFILE is not None:
        PROFILE.end("gconv", 2.0 * y.pixels * k * k * Cin * Cout, t0, (Cin, Cout, y.H, k))


def gconv_dgrad(g, w, gx, stride=1, pad=0, reflect=False):
    Cout, Cin, k, _ = w.shape
    t0 = PROFILE.begin() if PROFILE is not None else None
    hip.check(hip.load().tnr_gconv_dgrad(g.c(), gx.N, gx.H, gx.W, Cin, w.data_ptr(), gx.c(), g.H, g.W, Cout, k, stride, pad, int(reflect),
                                         hip.stream()), "gconv_dgrad")
    if PROFILE is not None:
        PROFILE.end("gconv", 2.0 * g.pixels * k * k * Cin * Cout, t0, (Cin, Cout, g.H, k + 100))


def gconv_wgrad(x, g, dw, db=None, stride=1, pad=0, reflect=False, alpha=1.0, beta=1.0):
    Cout, Cin, k, _ = dw.shape
    lib = hip.load()
    ws = WS.get("gconv_wgrad@%x" % hip.stream(), lib.tnr_gconv_wgrad_workspace_bytes(Cout, Cin, k), x.buf.device)
    t0 = PROFILE.begin() if PROFILE is not None else None
    hip.check(lib.tnr_gconv_wgrad(x.c(), x.N, x.H, x.W, Cin, g.c(), g.H, g.W, Cout, k, stride, pad, int(reflect), dw.data_ptr(), hip.ptr(db),
                                  alpha, beta, ws.data_ptr(), ws.numel() * 8, hip.stream()), "gconv_wgrad")
    if PROFILE is not None:
        PROFILE.end("gconv", 2.0 * g.pixels * k * k * Cin * Cout, t0, (Cin, Cout, g.H, k + 200))


def bias_grad(g, db, alpha=1.0, beta=1.0):
    """db = beta db + alpha * sum over the pixels of g (per channel)."""
    ws = WS.get("bias_grad@%x" % hip.stream(), 512 * g.C * 8, g.buf.device)
    hip.check(hip.load().tnr_bias_grad(g.c(), g.pixels, g.C, db.data_ptr(), alpha, beta, ws.data_ptr(), ws.numel() * 8, hip.stream()), "bias_grad")


def pad2d(x, y, pad, reflect):
    """y [N, H + 2 pad, W + 2 pad, C] = x with a zero (reflect False) or reflected border."""
    hip.check(hip.load().tnr_pad2d(x.c(), y.c(), x.N, x.H, x.W, x.C, pad, int(reflect), hip.stream()), "pad2d")


def unpad2d(xp, y, pad, fold):
    """y [N,H,W,C] = centre crop of xp (fold False) or the adjoint of the reflection padding (fold True)."""
    hip.check(hip.load().tnr_unpad2d(xp.c(), y.c(), y.N, y.H, y.W, y.C, pad, int(fold), hip.stream()), "unpad2d")


def window2d(src, dst, oy, ox, acc=False):
    """dst[n, y, x, :] (+)= src[n, y + oy, x + ox, :] inside src, 0 outside (same N and C; zero-embedding at an offset / offset
    crop; acc: added to dst instead of replacing it)."""
    hip.check(hip.load().tnr_window2d(src.c(), src.H, src.W, dst.c(), dst.N, dst.H, dst.W, dst.C, oy, ox, int(acc), hip.stream()), "window2d")


def tanh_fwd(x, y):
    hip.check(hip.load().tnr_tanh_fwd(x.data_ptr(), y.data_ptr(), x.numel(), hip.stream()), "tanh_fwd")


def tanh_bwd(g, y, gx):
    hip.check(hip.load().tnr_tanh_bwd(g.data_ptr(), y.data_ptr(), gx.data_ptr(), g.numel(), hip.stream()), "tanh_bwd")


def gan_loss(pred, kind, target, out, grad=None):
    """GANLoss against a constant label: kind 0 vanilla (BCE with logits), 1 lsgan (MSE), 2 mean(pred); out[0] = mean loss."""
    hip.check(hip.load().tnr_gan_loss(pred.data_ptr(), pred.numel(), kind, float(target), out.data_ptr(), hip.ptr(grad), hip.stream()), "gan_loss")


def axpby(dst, src, a=1.0, b=1.0):
    """dst = a*src + b*dst"""
    hip.check(hip.load().tnr_axpby(dst.c(), src.c(), dst.pixels, dst.C, a, b, hip.stream()), "axpby")


def mask_mul(g, y, mslope=0.2):
    hip.check(hip.load().tnr_mask_mul(g.c(), y.c(), g.pixels, g.C, mslope, hip.stream()), "mask_mul")


def fill(t, value=0.0):
    hip.check(hip.load().tnr_fill(t.data_ptr(), t.numel(), value, hip.stream()), "fill")


# ----------------------------------------------------------------------------------------------
# batch norm / linear
# ----------------------------------------------------------------------------------------------
def bn_train_fwd(z, y, gamma, beta, running_mean, running_var, num_batches, save_mean, save_invstd,
                 momentum=0.1, eps=1e-5, act=ACT_LRELU, slope=0.2, stat64=None):
    """stat64: optional fp64 [2 C] tensor that receives the batch mean / unbiased variance of the running-statistics update
    (bn_replay_running re-applies it)."""
    lib = hip.load()
    ws = WS.get("bn", lib.tnr_bn_workspace_bytes(z.C), z.buf.device)
    hip.check(lib.tnr_bn_train_fwd_stats(z.c(), y.c(), z.pixels, z.C, gamma.data_ptr(), beta.data_ptr(),
                                         hip.ptr(running_mean), hip.ptr(running_var), hip.ptr(num_batches), momentum, eps,
                                         save_mean.data_ptr(), save_invstd.data_ptr(), hip.ptr(stat64), act, slope, ws.data_ptr(),
                                         hip.stream()), "bn_train_fwd")


def bn_replay_running(running_mean, running_var, num_batches, stat64, momentum=0.1):
    """The side effects of one more training-mode forward over the same batch (running statistics, batch counter)."""
    hip.check(hip.load().tnr_bn_replay_running(running_mean.data_ptr(), running_var.data_ptr(), hip.ptr(num_batches), stat64.data_ptr(),
                                               running_mean.numel(), momentum, hip.stream()), "bn_replay_running")


BN_MASK_FROM_Z = os.environ.get("TNR_BN_MASK_FROM_Z", "1") != "0"      # A/B switch


def bn_train_bwd(gy, y, z, gz, gamma, save_mean, save_invstd, dgamma=None, dbeta=None, acc_beta=1.0, mslope=0.2, beta=None):
    """Backward of y = act(BatchNorm_train(z)).  beta (the forward's shift) given: the activation mask is recomputed from z
    (tnr_bn_train_bwd_z: bit-identical, y is not read)."""
    lib = hip.load()
    ws = WS.get("bn", lib.tnr_bn_workspace_bytes(z.C), z.buf.device)
    if beta is not None and BN_MASK_FROM_Z:
        hip.check(lib.tnr_bn_train_bwd_z(gy.c(), z.c(), gz.c(), z.pixels, z.C, gamma.data_ptr(), beta.data_ptr(), save_mean.data_ptr(),
                                         save_invstd.data_ptr(), mslope, hip.ptr(dgamma), hip.ptr(dbeta), acc_beta, ws.data_ptr(),
                                         hip.stream()), "bn_train_bwd_z")
        return
    hip.check(lib.tnr_bn_train_bwd(gy.c(), y.c(), z.c(), gz.c(), z.pixels, z.C, gamma.data_ptr(),
                                   save_mean.data_ptr(), save_invstd.data_ptr(), mslope, hip.ptr(dgamma),
                                   hip.ptr(dbeta), acc_beta, ws.data_ptr(), hip.stream()), "bn_train_bwd")


def instnorm_fwd(z, y, save_mean, save_invstd, eps=1e-5, act=ACT_NONE, slope=0.0):
    """y = act(InstanceNorm2d(z)) (no affine, per image and channel statistics); save_mean / save_invstd: [N * C]."""
    lib = hip.load()
    ws = WS.get("instnorm", lib.tnr_instnorm_workspace_bytes(z.N, z.C), z.buf.device)
    hip.check(lib.tnr_instnorm_fwd(z.c(), y.c(), z.N, z.H * z.W, z.C, eps, save_mean.data_ptr(), save_invstd.data_ptr(), act, slope,
                                   ws.data_ptr(), hip.stream()), "instnorm_fwd")


def instnorm_bwd(gy, y, z, gz, save_mean, save_invstd, mslope=1.0):
    """gz = d loss / d z of instnorm_fwd from gy, the gradient of its (activated) output; mslope: the activation's negative slope
    (0 ReLU, 1 no activation), the gate is read from y."""
    lib = hip.load()
    ws = WS.get("instnorm", lib.tnr_instnorm_workspace_bytes(z.N, z.C), z.buf.device)
    hip.check(lib.tnr_instnorm_bwd(gy.c(), y.c(), z.c(), gz.c(), z.N, z.H * z.W, z.C, save_mean.data_ptr(), save_invstd.data_ptr(),
                                   mslope, ws.data_ptr(), hip.stream()), "instnorm_bwd")


def linear_fwd(x, w, b, y, act=ACT_NONE, slope=0.2):
    N, In = x.shape
    Out = w.shape[0]
    hip.check(hip.load().tnr_linear_fwd(x.data_ptr(), w.data_ptr(), hip.ptr(b), y.data_ptr(), N, In, Out, act, slope,
                                        hip.stream()), "linear_fwd")


def linear_bwd(x, w, gy, yact=None, gx=None, dw=None, db=None, acc_beta=1.0, mslope=0.2):
    N, In = x.shape
    Out = w.shape[0]
    gpre = WS.get("lin", N * Out * 4, x.device)
    hip.check(hip.load().tnr_linear_bwd(x.data_ptr(), w.data_ptr(), gy.data_ptr(), hip.ptr(yact), mslope, hip.ptr(gx),
                                        hip.ptr(dw), hip.ptr(db), N, In, Out, acc_beta, gpre.data_ptr(), hip.stream()),
              "linear_bwd")


# ----------------------------------------------------------------------------------------------
# losses / optimiser
# ----------------------------------------------------------------------------------------------
def _red_ws(device):
    lib = hip.load()
    return WS.get("red", lib.tnr_reduce_workspace_bytes(), device)


def l1_mean_fwd(a, b, scale, out):
    hip.check(hip.load().tnr_l1_mean_fwd(a.data_ptr(), b.data_ptr(), a.numel(), scale, out.data_ptr(),
                                         _red_ws(a.device).data_ptr(), hip.stream()), "l1_mean_fwd")


def l1_mean_bwd(a, b, scale, gscale, ga, accumulate=False):
    hip.check(hip.load().tnr_l1_mean_bwd(a.data_ptr(), b.data_ptr(), a.numel(), scale, hip.ptr(gscale), ga.data_ptr(),
                                         int(accumulate), hip.stream()), "l1_mean_bwd")


def _reduce_ws(dev):
    return WS.get("reduce@%x" % hip.stream(), hip.load().tnr_reduce_workspace_bytes(), dev)


def ragan_phase_a(pf, pr, sums):
    hip.check(hip.load().tnr_ragan_phase_a(pf.data_ptr(), pr.data_ptr(), pf.numel(), sums.data_ptr(), _reduce_ws(pf.device).data_ptr(), hip.stream()), "ragan_a")


def ragan_phase_b(pf, pr, stage, sums):
    hip.check(hip.load().tnr_ragan_phase_b(pf.data_ptr(), pr.data_ptr(), pf.numel(), stage, sums.data_ptr(), _reduce_ws(pf.device).data_ptr(), hip.stream()),
              "ragan_b")


def ragan_phase_c(pf, pr, stage, weight, sums, out, gf, gr):
    hip.check(hip.load().tnr_ragan_phase_c(pf.data_ptr(), pr.data_ptr(), pf.numel(), stage, float(weight), sums.data_ptr(),
                                           out.data_ptr(), hip.ptr(gf), hip.ptr(gr), hip.stream()), "ragan_c")


def scale_by(dst, src, gscale):
    hip.check(hip.load().tnr_scale_by(dst.data_ptr(), src.data_ptr(), src.numel(), gscale.data_ptr(), hip.stream()),
              "scale_by")


def sumsq(g, out):
    hip.check(hip.load().tnr_sumsq(g.data_ptr(), g.numel(), out.data_ptr(), _red_ws(g.device).data_ptr(), hip.stream()),
              "sumsq")


def clip_by_norm(g, sumsq_t, max_norm):
    hip.check(hip.load().tnr_clip_by_norm(g.data_ptr(), g.numel(), sumsq_t.data_ptr(), max_norm, hip.stream()),
              "clip_by_norm")


def adam_step(p, g, m, v, step_size, b1, b2, bc2_sqrt, eps, wd=0.0):
    """One Adam launch over a flat buffer, behind the fault latch: if a tile hand-off of this process ever timed out, the update is
    NOT applied (the weights stay those of the last healthy step until check_engine_errors() raises)."""
    hip.check(hip.load().tnr_adam_step_guarded(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), step_size, b1,
                                               b2, bc2_sqrt, eps, wd, fault_word(p.device).data_ptr(), hip.stream()), "adam_step")
