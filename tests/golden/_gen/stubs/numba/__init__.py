"""Identity stand-in for numba, used ONLY by tests/golden/_gen/generate_golden.py.

numba is not installable in this image (no network).
``@njit`` promises the semantics of the decorated Python function, so the
reference's own source is executed by CPython to produce golden vectors.
Known deltas vs a real numba run are listed in tests/golden/README.md.
"""


def njit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]

    def deco(fn):
        return fn

    return deco


jit = njit
prange = range


def set_num_threads(n):
    return None
