__version__ = "0.2.4"  # tracks the reference release whose `adapted detect` surface is mirrored
