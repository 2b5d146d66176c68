"""Test-only stand-in for `numba` (absent from this image).

`njit`/`jit` become identity decorators, so the reference's @njit functions run as the
plain NumPy float64 Python they are written as (same expression order, no fastmath).
Build-authored test tooling: contains no reference code. Used only by
oracle/refshim/ref_loader.py inside the build container.
"""


def _identity_decorator(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]

    def wrap(fn):
        return fn

    return wrap


njit = _identity_decorator
jit = _identity_decorator
