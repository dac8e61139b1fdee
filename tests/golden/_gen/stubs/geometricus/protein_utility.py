"""Name-only stand-in."""


def get_structure_files(*a, **k):
    raise RuntimeError("geometricus is not available in this image")


parse_structure_file = get_structure_files
