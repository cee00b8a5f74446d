"""Per-call options of the operator (include/gsrast.h `gsr_options`, ABI v6; NEW, not in the reference's interface).

The reference's settings tuple (GaussianRasterizationSettings) is a fixed 12-field contract, so options that exist
only here -- the tile band of a tile-grid-sharded view, `fast_exp`, the A/B switches of the kernels -- travel beside
it: a thread-local stack, read when `GaussianRasterizer.forward` / `FusedGaussianRasterizer.forward` is called and
stored WITH THE GRAPH, so that the backward of that call (which autograd runs on another thread, possibly after the
`with` block has been left, possibly while another thread renders with other options) uses exactly the options of its
own forward.  Nothing process-global is touched:

    with gaustudio_amd.options(tile_band=(0, 34)):            # this thread renders tile rows [0, 34) only
        out = rasterizer(...)
    with gaustudio_amd.options(fast_exp=True):                # v_exp_f32 in both compositing kernels
        out = rasterizer(...); loss(out).backward()

A field that is not set falls back to the enclosing `with`, then to the process default
(`_C.set_option` / environment variables of libgsrast.so).
"""
import threading

FIELDS = ("tight_binning", "cull", "fwd_variant", "bwd_variant", "speculative", "tile_row_lo", "tile_row_hi", "fast_exp", "forward_only")
_tls = threading.local()


def current():
    """The options a rasterizer call made NOW on this thread would run with: a tuple of len(FIELDS) ints, -1 = process
    default."""
    stack = getattr(_tls, "stack", None)
    return stack[-1] if stack else (-1,) * len(FIELDS)


class options:
    def __init__(self, tile_band=None, fast_exp=None, tight_binning=None, cull=None, fwd_variant=None, bwd_variant=None,
                 speculative=None):
        self._set = {}
        if tile_band is not None:
            lo, hi = int(tile_band[0]), int(tile_band[1])
            if lo < 0:
                raise ValueError("tile_band: lo must be >= 0")
            self._set["tile_row_lo"], self._set["tile_row_hi"] = lo, (hi if hi > 0 else 0)
        for k, v in (("fast_exp", fast_exp), ("tight_binning", tight_binning), ("cull", cull), ("fwd_variant", fwd_variant),
                     ("bwd_variant", bwd_variant), ("speculative", speculative)):
            if v is not None:
                self._set[k] = int(v)

    def __enter__(self):
        cur = list(current())
        for k, v in self._set.items():
            cur[FIELDS.index(k)] = v
        if not hasattr(_tls, "stack"):
            _tls.stack = []
        _tls.stack.append(tuple(cur))
        return self

    def __exit__(self, *exc):
        _tls.stack.pop()
        return False


def resolved():
    """current(), with `fast_exp` resolved against the process default NOW.  The autograd Functions store THIS with the graph:
    the backward must evaluate exp() exactly as its forward did (the alpha >= 1/255 decisions depend on it), whatever a
    `_C.set_option("fast_exp", ...)` -- or another thread -- does to the process default between the two calls.  The other
    fields change no bit of the result and may stay deferred."""
    cur = list(current())
    i = FIELDS.index("fast_exp")
    if cur[i] < 0:
        from . import _C
        cur[i] = int(_C.get_option("fast_exp"))
        # fast_exp merely inherited from the process default (on since round 4), together with an A/B kernel variant that has
        # no v_exp_f32 form (per-wave lists: fwd_variant 1, bwd_variant bit 1): the variant wins, the call runs in the
        # reproducible mode -- asking for BOTH explicitly stays an error of the library
        if cur[i]:
            fv, bv = cur[FIELDS.index("fwd_variant")], cur[FIELDS.index("bwd_variant")]
            fv = int(_C.get_option("fwd_variant")) if fv < 0 else fv
            bv = int(_C.get_option("bwd_variant")) if bv < 0 else bv
            if fv == 1 or (bv >= 0 and (bv & 2)):
                cur[i] = 0
    return tuple(cur)


def note_grad_mode(enabled: bool):
    """Called by the module wrappers right before Function.apply: inside Function.forward grad mode is always off and
    ctx.needs_input_grad ignores torch.no_grad(), so the caller's grad mode has to be captured outside."""
    _tls.grad_enabled = bool(enabled)


def clear_grad_mode():
    """The wrappers' `finally`: a forward that raised before reading the note must not leave it behind for the next direct
    Function.apply of this thread (a stale False would make that training forward forward_only)."""
    _tls.grad_enabled = None


def take_grad_mode():
    """The grad mode noted for THIS call (True when the Function was applied directly, without the wrapper)."""
    g = getattr(_tls, "grad_enabled", None)
    _tls.grad_enabled = None
    return True if g is None else g


def for_forward(needs_backward: bool):
    """resolved(), plus `forward_only` = 1 when nothing of this call can be differentiated (torch.no_grad(), or no input requires
    grad): the forward then skips what only a backward would read."""
    needs_backward = bool(needs_backward) and take_grad_mode()
    cur = list(resolved())
    cur[FIELDS.index("forward_only")] = 0 if needs_backward else 1
    return tuple(cur)
