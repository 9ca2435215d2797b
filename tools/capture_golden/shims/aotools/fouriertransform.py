"""Stand-in for aotools.fouriertransform: centred, scaled DFT wrappers."""
from numpy import fft


def ft(data, delta):
    return fft.fftshift(fft.fft(fft.fftshift(data, axes=(-1)), axis=-1), axes=(-1)) * delta


def ift(DATA, delta_f):
    return fft.ifftshift(fft.ifft(fft.ifftshift(DATA, axes=(-1)), axis=-1), axes=(-1)) * len(DATA) * delta_f


def ft2(data, delta):
    return fft.fftshift(fft.fft2(fft.fftshift(data, axes=(-1, -2))), axes=(-1, -2)) * delta ** 2


def ift2(DATA, delta_f):
    N = DATA.shape[0]
    return fft.ifftshift(fft.ifft2(fft.ifftshift(DATA))) * (N * delta_f) ** 2
