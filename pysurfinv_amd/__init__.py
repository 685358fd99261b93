"""pysurfinv_amd -- MI355X-native drop-in for pySurfInv's fast_surf() forward path."""
__version__ = "0.1.0"
