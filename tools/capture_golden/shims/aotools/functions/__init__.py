from . import zernike
