from . import profile_compression
