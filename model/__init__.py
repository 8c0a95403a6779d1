"""Import shim: the reference driver's module names (test-stDiff.py:11-14) over stdiff_amd.model."""
