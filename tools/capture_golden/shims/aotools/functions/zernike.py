"""Stand-in for aotools.functions.zernike: Noll index -> (n, m)."""
import numpy


def zernIndex(j):
    n = int((-1.0 + numpy.sqrt(8 * (j - 1) + 1)) / 2.0)
    p = j - (n * (n + 1)) / 2.0
    k = n % 2
    m = int((p + k) / 2.0) * 2 - k
    if m != 0:
        m *= 1 if j % 2 == 0 else -1
    return [n, m]
