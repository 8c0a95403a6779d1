"""Import shim: the reference driver's `from process.data import *` (test-stDiff.py:20) over stdiff_amd.process."""
