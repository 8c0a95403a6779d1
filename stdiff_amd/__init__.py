"""stdiff_amd: the stDiff hot path (DiT_stDiff denoiser, NoiseScheduler, sample_stDiff, normal_train_stDiff)
on MI355X / gfx950 through hand-written HIP kernels (libstdiff_hip.so, C ABI in include/stdiff_hip.h).

Drop-in import surface (reference: test-stDiff.py:11-14,20):
    from stdiff_amd.model.stDiff_model import DiT_stDiff
    from stdiff_amd.model.stDiff_scheduler import NoiseScheduler
    from stdiff_amd.model.stDiff_train import normal_train_stDiff
    from stdiff_amd.model.sample import sample_stDiff
    from stdiff_amd.process.data import get_data_loader
"""
__version__ = "0.1.0"
