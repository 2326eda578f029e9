"""Multi-scale deformable attention op: the two functions of the reference's pybind module
`MultiScaleDeformableAttention` (models/ops/src/vision.cpp:13-16, ms_deform_attn.h:20-62) and the
autograd Function built on them (models/ops/functions/ms_deform_attn_func.py:21-38), on top of the
C ABI grit_msda_{fwd,bwd}_{f32,f64}.

Differences from the reference, on purpose:
  * `im2col_step` is accepted and ignored (no `batch % im2col_step == 0` failure);
  * bf16/fp16 inputs are computed in float32 and cast back (the reference dispatches float/double only);
  * launch failures raise (the reference only printf's them, ms_deform_im2col_cuda.cuh:948-952).
"""
import ctypes
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from grit_amd import lib as _lib


# bench.py sets this to a list to collect (kind, start_event, end_event, algorithmic_bytes) per launch: HIP events
# recorded on the launch stream right around the kernel, so the roofline figure is measured inside the real step
PROFILE_EVENTS = None

# bench.py --points spread: a callable loc -> loc that replaces the sampling locations right before the launch (diagnostic:
# a randomly initialised decoder keeps every point near the image centre; a trained one spreads them -- SURVEY 8d config 2)
LOC_OVERRIDE = None
# bench.py sets this to have forward launches also record (loc, spatial_shapes, level_start_index, B, S, M, value element size),
# from which the cache lines the gather touches are counted afterwards (unique_lines_touched)
PROFILE_RECORD_GEOMETRY = False

# bf16 value maps: grad_value is accumulated in f32 (float atomics into a staging map, ONE rounding to bf16 per touched cell:
# the precision of the reference's atomicAdd, ms_deform_im2col_cuda.cuh:125-152) -- grit_msda_bwd_bf16_staged.
# GRIT_MSDA_BWD_F32ACC=0 opts into accumulation IN bf16 by packed atomics (grit_msda_bwd_bf16acc*: twice the atomic rate, but
# every add rounds to 8 mantissa bits and the result depends on the arrival order more strongly).
F32_ACCUMULATE = os.environ.get("GRIT_MSDA_BWD_F32ACC", "1") != "0"
# How the f32 accumulation is done.  "sorted" (default): gather form -- contributions binned by cell in LDS, every cell summed in
# f32 registers, all inside the LDS of one CU per (image, head): no atomics on memory, dense output (grit_msda_bwd_bf16_sorted).  "staged": f32 atomics into a staging map + flush
# (grit_msda_bwd_bf16_staged; also the fallback where the sorted path does not apply: S or Lq*L*P beyond one workgroup's LDS).
F32_METHOD = os.environ.get("GRIT_MSDA_BWD_METHOD", "sorted")
if F32_METHOD not in ("sorted", "staged"):
    raise ValueError("GRIT_MSDA_BWD_METHOD must be 'sorted' or 'staged'")

_STAGE = {}  # (device, B, S, M) -> [stage f32 [B,S,M,64], cell flags u8 [B,S,M], dirty]: zero on entry AND exit of every call


def _staging(device, B, S, M):
    key = (str(device), B, S, M)
    ent = _STAGE.get(key)
    if ent is None:
        if len(_STAGE) >= 4:  # shapes change with the batch / image size: keep the scratch of the last few only
            _STAGE.clear()
        ent = _STAGE[key] = [torch.zeros((B, S, M, 64), dtype=torch.float32, device=device),
                             torch.zeros((B, S, M), dtype=torch.uint8, device=device), False]
    elif ent[2]:  # a previous call died between its two launches: the invariant (all zero) has to be re-established
        ent[0].zero_()
        ent[1].zero_()
        ent[2] = False
    return ent


def _bwd_staged(value_ptr, pixel_stride, shapes, lsi, loc, aw, go, B, S, M, D, L, Lq, P, gv_ptr, gl, ga, device):
    ent = _staging(device, B, S, M)
    ent[2] = True
    with _lib.device_guard(device), _Timed("bwd_bf16", _algorithmic_bytes("bwd", B, S, M, D, L, Lq, P, 2, 4, 4)):
        st = _lib.load().grit_msda_bwd_bf16_staged(value_ptr, pixel_stride, _ptr(shapes), _ptr(lsi), _ptr(loc), _ptr(aw), _ptr(go),
                                                   B, S, M, D, L, Lq, P, _ptr(ent[0]), _ptr(ent[1]), gv_ptr, _ptr(gl), _ptr(ga),
                                                   _lib.current_stream_ptr())
    _lib.check(st, "grit_msda_bwd_bf16_staged")
    ent[2] = False


def sorted_applies(B, S, M, L, Lq, P):
    """True when the gather-form backward handles the shape (one (image, head) problem must fit the LDS of a CU)."""
    return F32_METHOD == "sorted" and _lib.load().grit_msda_bwd_sorted_supported(B, S, M, L, Lq, P) == 0


LAST_BWD_SHAPE = None  # (B, S, M, L, Lq, P) of the last gather-form backward (bench.py prices its bytes with it)


def _bwd_sorted(value_ptr, pixel_stride, shapes, lsi, loc, aw, go, B, S, M, D, L, Lq, P, gv_ptr, gl, ga, device):
    """Value gradient in gather form; writes EVERY cell of the [B, S, M, 64] slice behind gv_ptr."""
    global LAST_BWD_SHAPE
    LAST_BWD_SHAPE = (B, S, M, L, Lq, P)
    with _lib.device_guard(device), _Timed("bwd_bf16", _algorithmic_bytes("bwd", B, S, M, D, L, Lq, P, 2, 4, 2)):
        st = _lib.load().grit_msda_bwd_bf16_sorted(value_ptr, pixel_stride, _ptr(shapes), _ptr(lsi), _ptr(loc), _ptr(aw), _ptr(go),
                                                   B, S, M, D, L, Lq, P, gv_ptr, _ptr(gl), _ptr(ga), _lib.current_stream_ptr())
    _lib.check(st, "grit_msda_bwd_bf16_sorted")


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _algorithmic_bytes(kind, B, S, M, D, L, Lq, P, esize, geom_esize=None, grad_esize=None):
    """every tensor once (forward); value-sized traffic x3 + loc/weights read+written (backward).
    esize: bytes per element of the value map / output rows as the kernel sees them; geom_esize: of sampling
    locations and weights (fp32 on the bf16 path); grad_esize: of grad_value (fp32 atomics on the bf16 path)."""
    geom_esize = geom_esize or esize
    grad_esize = grad_esize or esize
    pts = B * Lq * M * L * P
    if kind == "fwd":
        return esize * (B * S * M * D + B * Lq * M * D) + geom_esize * 3 * pts
    # read value, read-modify-write grad_value, read grad_output; read loc + weights, write their gradients
    return esize * (B * S * M * D + B * Lq * M * D) + grad_esize * 2 * B * S * M * D + geom_esize * 2 * 3 * pts


def unique_lines_touched(loc, shapes, lsi, S, M, line_pixels=1):
    """Number of distinct (image, pixel, head) cells the bilinear gather of one forward launch reads: for D = 64 a cell is one
    128-byte line of a bf16 map (two lines of an fp32 map).  Same in-range rule as the kernel
    (ms_deform_im2col_cuda.cuh:288, 56-78): a point counts if -1 < h < H and -1 < w < W, a corner if it lies inside the map."""
    B, Lq, _, L, P, _ = loc.shape
    loc = loc.float()
    keys = []
    for l in range(L):
        H, W = int(shapes[l, 0]), int(shapes[l, 1])
        x = loc[:, :, :, l, :, 0] * W - 0.5
        y = loc[:, :, :, l, :, 1] * H - 0.5
        ok = (y > -1) & (x > -1) & (y < H) & (x < W)
        x0, y0 = torch.floor(x).long(), torch.floor(y).long()
        b = torch.arange(B, device=loc.device).view(B, 1, 1, 1)
        m = torch.arange(M, device=loc.device).view(1, 1, M, 1)
        for dy in (0, 1):
            for dx in (0, 1):
                xx, yy = x0 + dx, y0 + dy
                inside = ok & (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
                cell = ((b * S + int(lsi[l]) + yy * W + xx) * M + m)
                keys.append(cell[inside])
    keys = torch.cat(keys)
    return int(torch.unique(keys).numel()) if keys.numel() else 0


class _Timed(object):

    def __init__(self, kind, nbytes, geometry=None):
        self.kind, self.nbytes, self.geometry = kind, nbytes, geometry

    def __enter__(self):
        self.on = PROFILE_EVENTS is not None and not torch.cuda.is_current_stream_capturing()
        if self.on:
            self.a, self.b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.a.record()
        return self

    def __exit__(self, *exc):
        if self.on and PROFILE_EVENTS is not None:
            self.b.record()
            PROFILE_EVENTS.append((self.kind, self.a, self.b, self.nbytes, self.geometry if PROFILE_RECORD_GEOMETRY else None))
        return False


def _check_inputs(value, shapes, lsi, loc, aw):
    _lib.require_device(value, shapes, lsi, loc, aw)
    if value.dim() != 4 or loc.dim() != 6 or aw.dim() != 5:
        raise RuntimeError("ms_deform_attn: value [B,S,M,D], sampling_loc [B,Lq,M,L,P,2], attn_weight [B,Lq,M,L,P] expected")
    for name, t in (("value", value), ("spatial_shapes", shapes), ("level_start_index", lsi),
                    ("sampling_loc", loc), ("attn_weight", aw)):
        if not t.is_contiguous():  # reference: AT_ASSERTM(x.is_contiguous()), ms_deform_attn_cuda.cu:28-32
            raise RuntimeError("%s tensor has to be contiguous" % name)
    if shapes.dtype != torch.int64 or lsi.dtype != torch.int64:
        raise RuntimeError("spatial_shapes / level_start_index must be int64")
    B, S, M, D = value.shape
    _, Lq, M2, L, P, two = loc.shape
    if M2 != M or two != 2 or tuple(aw.shape) != (B, Lq, M, L, P) or shapes.shape[0] != L or loc.shape[0] != B:
        raise RuntimeError("ms_deform_attn: inconsistent shapes")
    return B, S, M, D, L, Lq, P


def _compute_dtype(t):
    return torch.float64 if t.dtype == torch.float64 else torch.float32


def _bf16_fast_path(value, D, L, P):
    return value.dtype == torch.bfloat16 and D == 64 and L * P <= 16


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step=64):
    B, S, M, D, L, Lq, P = _check_inputs(value, spatial_shapes, level_start_index, sampling_loc, attn_weight)
    if LOC_OVERRIDE is not None:
        sampling_loc = LOC_OVERRIDE(sampling_loc).to(sampling_loc.dtype).contiguous()
    if _bf16_fast_path(value, D, L, P):  # value map stays bf16: no fp32 staging copy
        loc, aw = sampling_loc.float(), attn_weight.float()
        out = torch.empty((B, Lq, M * D), dtype=torch.bfloat16, device=value.device)
        geometry = (loc, spatial_shapes, level_start_index, B, S, M, 2) if PROFILE_RECORD_GEOMETRY else None
        with _lib.device_guard(value.device), _Timed("fwd_bf16", _algorithmic_bytes("fwd", B, S, M, D, L, Lq, P, 2, 4), geometry):
            st = _lib.load().grit_msda_fwd_bf16(_ptr(value), _ptr(spatial_shapes), _ptr(level_start_index), _ptr(loc),
                                                _ptr(aw), B, S, M, D, L, Lq, P, _ptr(out), _lib.current_stream_ptr())
        _lib.check(st, "grit_msda_fwd_bf16")
        return out
    cdt = _compute_dtype(value)
    v, loc, aw = value.to(cdt), sampling_loc.to(cdt), attn_weight.to(cdt)
    out = torch.empty((B, Lq, M * D), dtype=cdt, device=value.device)
    fn = _lib.load().grit_msda_fwd_f64 if cdt == torch.float64 else _lib.load().grit_msda_fwd_f32
    geometry = (loc, spatial_shapes, level_start_index, B, S, M, v.element_size()) if PROFILE_RECORD_GEOMETRY else None
    with _lib.device_guard(value.device), _Timed("fwd", _algorithmic_bytes("fwd", B, S, M, D, L, Lq, P, v.element_size()), geometry):
        st = fn(_ptr(v), _ptr(spatial_shapes), _ptr(level_start_index), _ptr(loc), _ptr(aw),
                B, S, M, D, L, Lq, P, _ptr(out), _lib.current_stream_ptr())
    _lib.check(st, "grit_msda_fwd")
    return out.to(value.dtype)


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output,
                            im2col_step=64):
    B, S, M, D, L, Lq, P = _check_inputs(value, spatial_shapes, level_start_index, sampling_loc, attn_weight)
    _lib.require_device(grad_output)
    if _bf16_fast_path(value, D, L, P) and not F32_ACCUMULATE:
        # value gradient accumulated in bf16 by packed atomics (two channels per memory-side atomic: the f32-atomic rate
        # is what bounds this kernel) -- the rounding behaviour of torch's own bf16 scatter / grid_sample backward
        loc, aw = sampling_loc.float(), attn_weight.float()
        go = grad_output.to(torch.bfloat16).contiguous()
        gv = torch.zeros(value.shape, dtype=torch.bfloat16, device=value.device)
        gl, ga = torch.empty_like(loc), torch.empty_like(aw)
        with _lib.device_guard(value.device), _Timed("bwd_bf16", _algorithmic_bytes("bwd", B, S, M, D, L, Lq, P, 2, 4, 2)):
            st = _lib.load().grit_msda_bwd_bf16acc(_ptr(value), _ptr(spatial_shapes), _ptr(level_start_index), _ptr(loc),
                                                   _ptr(aw), _ptr(go), B, S, M, D, L, Lq, P, _ptr(gv), _ptr(gl), _ptr(ga),
                                                   _lib.current_stream_ptr())
        _lib.check(st, "grit_msda_bwd_bf16acc")
        return [gv, gl.to(sampling_loc.dtype), ga.to(attn_weight.dtype)]
    if _bf16_fast_path(value, D, L, P):
        loc, aw = sampling_loc.float(), attn_weight.float()
        go = grad_output.to(torch.bfloat16).contiguous()
        gl, ga = torch.empty_like(loc), torch.empty_like(aw)
        if sorted_applies(B, S, M, L, Lq, P):
            gv = torch.empty(value.shape, dtype=torch.bfloat16, device=value.device)  # every cell is written
            _bwd_sorted(_ptr(value), M * D, spatial_shapes, level_start_index, loc.contiguous(), aw.contiguous(), go,
                        B, S, M, D, L, Lq, P, _ptr(gv), gl, ga, value.device)
        else:
            gv = torch.zeros(value.shape, dtype=torch.bfloat16, device=value.device)
            _bwd_staged(_ptr(value), M * D, spatial_shapes, level_start_index, loc, aw, go, B, S, M, D, L, Lq, P, _ptr(gv), gl,
                        ga, value.device)
        return [gv, gl.to(sampling_loc.dtype), ga.to(attn_weight.dtype)]
    cdt = _compute_dtype(value)
    v, loc, aw = value.to(cdt), sampling_loc.to(cdt), attn_weight.to(cdt)
    go = grad_output.to(cdt).contiguous()
    gv = torch.zeros_like(v)  # atomics accumulate into it (ms_deform_attn_cuda.cu:121)
    gl = torch.empty_like(loc)
    ga = torch.empty_like(aw)
    fn = _lib.load().grit_msda_bwd_f64 if cdt == torch.float64 else _lib.load().grit_msda_bwd_f32
    with _lib.device_guard(value.device), _Timed("bwd", _algorithmic_bytes("bwd", B, S, M, D, L, Lq, P, v.element_size())):
        st = fn(_ptr(v), _ptr(spatial_shapes), _ptr(level_start_index), _ptr(loc), _ptr(aw), _ptr(go),
                B, S, M, D, L, Lq, P, _ptr(gv), _ptr(gl), _ptr(ga), _lib.current_stream_ptr())
    _lib.check(st, "grit_msda_bwd")
    return [gv.to(value.dtype), gl.to(sampling_loc.dtype), ga.to(attn_weight.dtype)]


class StackedValueMaps(object):
    """The value maps of all decoder layers in ONE tensor [B, S, layers, M, D] (bf16) -- the output of a single projection
    GEMM -- plus the bookkeeping that lets the layers' backward kernels accumulate into ONE gradient buffer of the same
    layout.  Autograd sees the stacked tensor as an input of every layer's sampling node; only the node that runs last in
    the backward pass (the first layer's) hands the shared buffer back as the gradient, the others return None, so the
    engine neither materialises per-layer gradients nor adds them up."""

    def __init__(self, stacked, layers):
        assert stacked.dim() == 5 and stacked.is_contiguous() and stacked.dtype == torch.bfloat16
        self.stacked, self.layers = stacked, layers
        self.pending, self.grad = 0, None
        self.layers_seen = set()

    def layer_ptr(self, tensor, layer):
        B, S, n, M, D = self.stacked.shape
        return ctypes.c_void_p(tensor.data_ptr() + layer * M * D * tensor.element_size())


class _StackedMSDAFn(Function):
    """ms_deform_attn on slice `layer` of a StackedValueMaps (bf16, D = 64, L*P <= 16: the strided kernels)."""

    @staticmethod
    def forward(ctx, stacked, maps, layer, shapes, lsi, loc, aw):
        B, S, n, M, D = stacked.shape
        _, Lq, _, L, P, _ = loc.shape
        loc, aw = loc.float().contiguous(), aw.float().contiguous()
        if LOC_OVERRIDE is not None:
            loc = LOC_OVERRIDE(loc).float().contiguous()
        out = torch.empty((B, Lq, M * D), dtype=torch.bfloat16, device=stacked.device)
        geometry = (loc, shapes, lsi, B, S, M, 2) if PROFILE_RECORD_GEOMETRY else None
        with _lib.device_guard(stacked.device), _Timed("fwd_bf16", _algorithmic_bytes("fwd", B, S, M, D, L, Lq, P, 2, 4), geometry):
            st = _lib.load().grit_msda_fwd_bf16_strided(maps.layer_ptr(stacked, layer), n * M * D, _ptr(shapes), _ptr(lsi),
                                                        _ptr(loc), _ptr(aw), B, S, M, D, L, Lq, P, _ptr(out),
                                                        _lib.current_stream_ptr())
        _lib.check(st, "grit_msda_fwd_bf16_strided")
        ctx.save_for_backward(stacked, shapes, lsi, loc, aw)
        ctx.maps, ctx.layer = maps, layer
        maps.pending += 1
        maps.layers_seen.add(layer)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        stacked, shapes, lsi, loc, aw = ctx.saved_tensors
        maps, layer = ctx.maps, ctx.layer
        B, S, n, M, D = stacked.shape
        _, Lq, _, L, P, _ = loc.shape
        use_sorted = F32_ACCUMULATE and sorted_applies(B, S, M, L, Lq, P)
        if maps.grad is None:
            # sorted path: every layer's kernel writes its whole slice, so the buffer needs no fill -- provided every slice has
            # a sampling node waiting (all layers of the decoder ran); otherwise one fill for all layers
            every_slice = use_sorted and maps.layers_seen == set(range(n)) and maps.pending == n
            maps.grad = torch.empty_like(stacked) if every_slice else torch.zeros_like(stacked)
        go = grad_output.to(torch.bfloat16).contiguous()
        gl, ga = torch.empty_like(loc), torch.empty_like(aw)
        if use_sorted:
            _bwd_sorted(maps.layer_ptr(stacked, layer), n * M * D, shapes, lsi, loc, aw, go, B, S, M, D, L, Lq, P,
                        maps.layer_ptr(maps.grad, layer), gl, ga, stacked.device)
        elif F32_ACCUMULATE:
            _bwd_staged(maps.layer_ptr(stacked, layer), n * M * D, shapes, lsi, loc, aw, go, B, S, M, D, L, Lq, P,
                        maps.layer_ptr(maps.grad, layer), gl, ga, stacked.device)
        else:
            with _lib.device_guard(stacked.device), _Timed("bwd_bf16", _algorithmic_bytes("bwd", B, S, M, D, L, Lq, P, 2, 4, 2)):
                st = _lib.load().grit_msda_bwd_bf16acc_strided(maps.layer_ptr(stacked, layer), n * M * D, _ptr(shapes), _ptr(lsi),
                                                               _ptr(loc), _ptr(aw), _ptr(go), B, S, M, D, L, Lq, P,
                                                               maps.layer_ptr(maps.grad, layer), _ptr(gl), _ptr(ga),
                                                               _lib.current_stream_ptr())
            _lib.check(st, "grit_msda_bwd_bf16acc_strided")
        maps.pending -= 1
        gstacked = None
        if maps.pending == 0:  # every layer has added its part
            gstacked, maps.grad = maps.grad, None
        return gstacked, None, None, None, None, gl, ga


def stacked_fast_path(stacked, L, P):
    return (stacked.is_cuda and stacked.dtype == torch.bfloat16 and stacked.shape[-1] == 64 and L * P <= 16
            and stacked.numel() * 2 < (1 << 32))


def ms_deform_attn_stacked(maps, layer, shapes, lsi, loc, aw):
    return _StackedMSDAFn.apply(maps.stacked, maps, layer, shapes, lsi, loc, aw)


class MSDeformAttnFunction(Function):
    """Same call signature and gradient tuple as the reference Function."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
                im2col_step=64):
        ctx.im2col_step = im2col_step
        value, sampling_locations, attention_weights = (value.contiguous(), sampling_locations.contiguous(),
                                                        attention_weights.contiguous())
        output = ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                                        attention_weights, im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                              attention_weights)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, lsi, loc, aw = ctx.saved_tensors
        gv, gl, ga = ms_deform_attn_backward(value, shapes, lsi, loc, aw, grad_output.contiguous(), ctx.im2col_step)
        return gv, None, None, gl, ga, None
