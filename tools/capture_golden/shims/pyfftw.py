"""Stand-in for pyfftw: FFTW(IN, OUT, axes) == unnormalised FORWARD DFT over `axes`."""
import numpy


def empty_aligned(shape, dtype="complex128", **k):
    return numpy.empty(shape, dtype=dtype)


class FFTW:
    def __init__(self, input_array, output_array, axes=(-1,), direction="FFTW_FORWARD",
                 flags=(), threads=1):
        assert direction == "FFTW_FORWARD"
        self._in, self._out, self._axes = input_array, output_array, tuple(axes)

    def __call__(self):
        self._out[:] = numpy.fft.fftn(self._in, axes=self._axes)
        return self._out
