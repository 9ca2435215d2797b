"""Stand-in: orbit mechanics are out of scope; names only."""
class _L:
    def timescale(self):
        return None
    def tle_file(self, *a, **k):
        raise NotImplementedError
load = _L()
def wgs84(*a, **k):
    raise NotImplementedError
class EarthSatellite:
    pass
class Topos:
    pass
