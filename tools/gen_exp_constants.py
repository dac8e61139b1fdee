#!/usr/bin/env python3
"""Print the constants of the shared FP64 exp algorithm (hex floats).

exp(x) = 2^(k/16) * exp(r),  k = RN(x*16/ln2),  r = x - k*ln2/16 (|r| <= ln2/32)
Table entries 2^(j/16) are split hi + lo (hi = RN(value), lo = RN(value - hi)).
Computed with 60-digit decimal arithmetic; float(Decimal) rounds correctly.
"""
from decimal import Decimal, getcontext

getcontext().prec = 60
LN2 = Decimal(2).ln()


def split(v):
    hi = float(v)
    lo = float(v - Decimal(hi))
    return hi, lo


print("INV_LN2_16 =", float(Decimal(16) / LN2).hex())
hi, lo = split(LN2 / Decimal(16))
print("LN2_16_HI  =", hi.hex())
print("LN2_16_LO  =", lo.hex())
for j in range(16):
    v = (LN2 * Decimal(j) / Decimal(16)).exp()
    hi, lo = split(v)
    print(f"  {{ {hi.hex()}, {lo.hex()} }}, /* 2^({j}/16) */")
for n in range(2, 8):
    f = 1
    for i in range(2, n + 1):
        f *= i
    print(f"C{n} = {float(Decimal(1) / Decimal(f)).hex()}  /* 1/{n}! */")
