"""Exhaustive check (exact rational arithmetic) that  q' = fma(fma(-255, q, p), r, q),  q = RN(p * r),  r = RN(1/255)
equals the IEEE float32 quotient RN(p / 255) for every integer p in 0..255 (aux_kernels.hip::div255_exact)."""
import math
from fractions import Fraction as F

import numpy as np


def rn32(x):
    """Round a Fraction to the nearest float32 (ties to even); returns a Fraction."""
    if x == 0:
        return F(0)
    s, a = (1 if x > 0 else -1), abs(x)
    e = math.floor(math.log2(float(a)))
    while F(2) ** e > a:
        e -= 1
    while F(2) ** (e + 1) <= a:
        e += 1
    ulp = F(2) ** (e - 23)
    q = a / ulp
    n = q.numerator // q.denominator
    rem = q - n
    if rem > F(1, 2) or (rem == F(1, 2) and n % 2 == 1):
        n += 1
    return s * n * ulp


def main():
    r = rn32(F(1, 255))
    bad_mul = bad = 0
    for p in range(256):
        exact = rn32(F(p, 255))
        assert float(exact) == float(np.float32(p) / np.float32(255))
        q = rn32(F(p) * r)
        e = rn32(F(p) - 255 * q)
        q2 = rn32(q + e * r)
        bad_mul += q != exact
        bad += q2 != exact
    print(f"p * r alone differs from p / 255 for {bad_mul} of 256 inputs; with the fma correction: {bad}")
    assert bad == 0


if __name__ == "__main__":
    main()
