"""Deterministic synthetic structure families.

The reference ships no benchmark inputs, so every configuration in
BASELINE.json is exercised on a *two-level clade family*: a random-walk
C-alpha trace, K clades derived from it by coordinate/tensor noise plus an
indel, and P/K members per clade derived the same way, each finally moved by a
random rigid motion.  Indels keep the length fixed (delete g residues, append g
fresh ones) so that every pair needs gaps.  All arrays are float64 C-contiguous,
as the reference's ``Protein`` expects (``multiple_alignment.py:312-319,486``).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

import numpy as np

CA_STEP = 3.8  # Angstrom between consecutive C-alpha atoms


@dataclass
class Structure:
    name: str
    tensors: np.ndarray  # (L, d) float64
    coordinates: np.ndarray  # (L, 3) float64
    sequence: str


def _walk(rng: np.random.Generator, length: int, start=None) -> np.ndarray:
    steps = rng.normal(size=(length, 3))
    steps /= np.linalg.norm(steps, axis=1, keepdims=True)
    steps *= CA_STEP
    trace = np.cumsum(steps, axis=0)
    if start is not None:
        trace = trace + start
    return trace


def _indel(rng, coords, tensors, gmax, append=True):
    g = int(rng.integers(0, gmax + 1)) if gmax > 0 else 0
    if g == 0:
        return coords, tensors
    length = coords.shape[0]
    start = int(rng.integers(0, length - g + 1))
    keep = np.r_[0:start, start + g:length]
    coords, tensors = coords[keep], tensors[keep]
    if append:
        tail = _walk(rng, g, start=coords[-1])
        coords = np.vstack([coords, tail])
        tensors = np.vstack([tensors, rng.uniform(size=(g, tensors.shape[1]))])
    return coords, tensors


def _random_rotation(rng) -> np.ndarray:
    q, r = np.linalg.qr(rng.normal(size=(3, 3)))
    q = q * np.sign(np.diag(r))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q


def make_family(num: int, length: int, dim: int = 10, seed: int = 20240,
                ragged: bool = False, clades: int | None = None) -> List[Structure]:
    """P=``num`` structures of ``length`` residues with ``dim``-wide tensors."""
    rng = np.random.default_rng(seed)
    if clades is None:
        clades = 8 if num >= 32 else 4
    clades = max(1, min(clades, num))
    base = _walk(rng, length)
    t0 = rng.uniform(size=(length, dim))
    per = [num // clades + (1 if c < num % clades else 0) for c in range(clades)]
    out: List[Structure] = []
    for c in range(clades):
        xc = base + rng.normal(scale=2.0, size=base.shape)
        tc = t0 + rng.normal(scale=0.08, size=t0.shape)
        xc, tc = _indel(rng, xc, tc, length // 10)
        for _ in range(per[c]):
            x = xc + rng.normal(scale=1.0, size=xc.shape)
            t = tc + rng.normal(scale=0.03, size=tc.shape)
            x, t = _indel(rng, x, t, length // 20)
            if ragged:
                target = int(rng.integers(int(0.8 * length), length + 1))
                cut = x.shape[0] - target
                if cut > 0:
                    s = int(rng.integers(0, x.shape[0] - cut + 1))
                    keep = np.r_[0:s, s + cut:x.shape[0]]
                    x, t = x[keep], t[keep]
            rot = _random_rotation(rng)
            x = x @ rot + rng.uniform(-50.0, 50.0, size=3)
            idx = len(out)
            out.append(Structure(
                name=f"s{idx:04d}",
                tensors=np.ascontiguousarray(t, dtype=np.float64),
                coordinates=np.ascontiguousarray(x, dtype=np.float64),
                sequence="A" * x.shape[0],
            ))
    return out


def make_ragged_family(num: int, shortest: int, longest: int, dim: int = 10, seed: int = 20250, clades: int | None = None) -> List[Structure]:
    """A family whose members keep a random contiguous window of ``shortest`` .. ``longest`` residues of a ``longest``-residue
    family member each: ragged lengths as every real input has them (the reference's own example is 85 / 79 / 80 residues),
    still one family (related structures, alignments with gaps)."""
    fam = make_family(num, longest, dim=dim, seed=seed, clades=clades)
    rng = np.random.default_rng(seed + 7)
    for s in fam:
        keep = int(rng.integers(shortest, longest + 1))
        start = int(rng.integers(0, s.coordinates.shape[0] - keep + 1))
        s.coordinates = np.ascontiguousarray(s.coordinates[start:start + keep])
        s.tensors = np.ascontiguousarray(s.tensors[start:start + keep])
        s.sequence = s.sequence[start:start + keep]
    return fam


def make_mixed_family(num_short: int, short_len: int, num_long: int, long_len: int, dim: int = 10, seed: int = 20251) -> List[Structure]:
    """``num_short`` structures of ``short_len`` residues (windows of the long ones' family) followed by ``num_long`` members of
    ``long_len`` residues: a family of domains with a few full-length chains in it."""
    fam = make_family(num_short + num_long, long_len, dim=dim, seed=seed)
    rng = np.random.default_rng(seed + 11)
    for s in fam[:num_short]:
        start = int(rng.integers(0, long_len - short_len + 1))
        s.coordinates = np.ascontiguousarray(s.coordinates[start:start + short_len])
        s.tensors = np.ascontiguousarray(s.tensors[start:start + short_len])
        s.sequence = s.sequence[start:start + short_len]
    return fam


def pack(structures: List[Structure]):
    """Concatenate a family into the packed layout the C-ABI takes.

    Returns ``coords (sum L, 3)``, ``tensors (sum L, d)``, ``offsets int64 (P+1)``.
    """
    lens = [s.coordinates.shape[0] for s in structures]
    offsets = np.zeros(len(structures) + 1, dtype=np.int64)
    offsets[1:] = np.cumsum(lens)
    coords = np.ascontiguousarray(np.vstack([s.coordinates for s in structures]), dtype=np.float64)
    tensors = np.ascontiguousarray(np.vstack([s.tensors for s in structures]), dtype=np.float64)
    return coords, tensors, offsets
