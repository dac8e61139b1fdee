"""Name-only stand-in for geometricus (un-vendored dependency of the reference).

The golden generator builds Protein objects directly from synthetic tensors and
never touches these names.
"""


class _Absent:
    def __getattr__(self, name):
        raise RuntimeError("geometricus is not available in this image")


moment_invariants = _Absent()
Geometricus = _Absent()
ShapemerLearn = _Absent()
