"""Name-only stand-in for Bio.PDB."""
