from . import fits
