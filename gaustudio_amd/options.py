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
    return tuple(cur)


def for_forward(needs_backward: bool):
    """resolved(), plus `forward_only` = 1 when nothing of this call can be differentiated (torch.no_grad(), or no input requires
    grad): the forward then skips what only a backward would read."""
    cur = list(resolved())
    cur[FIELDS.index("forward_only")] = 0 if needs_backward else 1
    return tuple(cur)
