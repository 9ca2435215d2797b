"""Stand-in: FITS I/O is out of scope; names only."""
class Header(dict):
    pass
def writeto(*a, **k):
    raise NotImplementedError
def getheader(*a, **k):
    raise NotImplementedError
def getdata(*a, **k):
    raise NotImplementedError
