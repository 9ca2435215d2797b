"""Stand-in for `aotools` (absent here): only what /root/reference/fast imports."""
import numpy
from . import fouriertransform
from . import functions
from . import turbulence


def circle(radius, size, circle_centre=(0, 0), origin="middle"):
    # pixel-centre coordinates 0.5, 1.5, ...; disc centred on size/2 for "middle"
    c = numpy.arange(0.5, size, 1.0)
    x, y = numpy.meshgrid(c, c)
    if origin == "middle":
        x = x - size / 2.0
        y = y - size / 2.0
    x = x - circle_centre[0]
    y = y - circle_centre[1]
    return (x * x + y * y <= radius * radius).astype(float)


def gaussian2d(size, width, amplitude=1.0, cent=None):
    try:
        ys, xs = size[0], size[1]          # (rows, columns): the image has shape `size`
    except (TypeError, IndexError):
        xs = ys = size
    try:
        yw, xw = float(width[0]), float(width[1])
    except (TypeError, IndexError):
        xw = yw = float(width)
    if not cent:
        xc, yc = xs / 2.0, ys / 2.0
    else:
        yc, xc = cent[0], cent[1]
    X, Y = numpy.meshgrid(range(0, xs), range(0, ys))
    return amplitude * numpy.exp(-(((xc - X) / xw) ** 2 + ((yc - Y) / yw) ** 2) / 2)


def cn2_to_r0(cn2, lamda=500.0e-9):
    return (0.423 * (2 * numpy.pi / lamda) ** 2 * cn2) ** (-3.0 / 5.0)


def isoplanaticAngle(cn2, h, lamda=500.0e-9):
    # aotools returns ARCSECONDS (atmos_conversions.isoplanaticAngle: "... * 180. * 3600. / numpy.pi")
    Jh = numpy.sum(cn2 * h ** (5.0 / 3.0))
    return 0.057 * lamda ** (6.0 / 5.0) * Jh ** (-3.0 / 5.0) * 180.0 * 3600.0 / numpy.pi


def coherenceTime(cn2, v, lamda=500.0e-9):
    Jv = numpy.sum(cn2 * v ** (5.0 / 3.0))
    return float(0.057 * lamda ** (6.0 / 5.0) * Jv ** (-3.0 / 5.0))


def rytov_variance(cn2, h, lamda=500.0e-9):
    # plane-wave Rytov variance 2.25 k^(7/6) sum cn2 h^(5/6)
    k = 2 * numpy.pi / lamda
    return float(2.25 * k ** (7.0 / 6.0) * numpy.sum(cn2 * h ** (5.0 / 6.0)))
