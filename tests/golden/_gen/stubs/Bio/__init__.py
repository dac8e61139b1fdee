"""Name-only stand-in for Bio (never called by the golden generator)."""
