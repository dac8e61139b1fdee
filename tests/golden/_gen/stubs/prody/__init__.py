"""Name-only stand-in (annotations in the reference mention prody.AtomGroup)."""


class AtomGroup:  # noqa: D401 - placeholder type
    pass
