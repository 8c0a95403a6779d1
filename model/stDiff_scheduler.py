from stdiff_amd.model.stDiff_scheduler import NoiseScheduler, betas_for_alpha_bar  # noqa: F401  (reference: model/stDiff_scheduler.py)
